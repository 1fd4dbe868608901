"""Cross-wave repeat rate of leaf states in steady state: what an unbounded on-device evaluation cache could save."""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET
from grok_alpha_zero_amd.net import Connect4Net
G = 4096
net = Connect4Net(6).eval()
eng = SelfPlayEngine("Connect4", G, 200, 42, 8, 7, 2.5, 0.5, seed=1234, evaluator=EVAL_RESNET, net_blocks=6, ring_capacity=0)
eng.load_weights(net.export_engine_weights())
eng.run_waves(3000)
seen = set(); tot = hit = 0; by_stones_tot = np.zeros(43, np.int64); by_stones_hit = np.zeros(43, np.int64)
for w in range(600):
    eng.run_waves(1)
    x, pend = eng.read_batch()
    rows = x.reshape(G, -1)[np.asarray(pend) != 0]
    stones = (x.reshape(G, 42, 4)[np.asarray(pend) != 0][:, :, 1] != 0).sum(1) if False else None
    keys = [r.tobytes() for r in rows]
    nst = (rows.reshape(len(rows), 42, 4)[:, :, 0] != 0).sum(1)
    for k, s in zip(keys, nst):
        tot += 1; by_stones_tot[s] += 1
        if k in seen:
            hit += 1; by_stones_hit[s] += 1
        else:
            seen.add(k)
    if w in (99, 299, 599):
        print(f"waves {w + 1}: evaluations {tot}, repeats {hit} ({hit / tot:.3f}), distinct {len(seen)}", flush=True)
print("repeat rate by stones on the leaf board:", [(int(s), round(by_stones_hit[s] / max(1, by_stones_tot[s]), 2), int(by_stones_tot[s])) for s in range(0, 43, 3)])
