"""How many leaf states of one wave are duplicates across games?  (on-device evaluation cache.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET
from grok_alpha_zero_amd.net import Connect4Net
G = 4096
net = Connect4Net(6).eval()
eng = SelfPlayEngine("Connect4", G, 200, 42, 8, 7, 2.5, 0.5, seed=1234, evaluator=EVAL_RESNET, net_blocks=6, ring_capacity=0)
eng.load_weights(net.export_engine_weights())
for warm in (50, 400, 2000, 4000):
    eng.run_waves(warm)
    fr = []
    for _ in range(5):
        eng.run_waves(7)
        x, pend = eng.read_batch()
        rows = x.reshape(G, -1)[np.asarray(pend) != 0]
        uniq = np.unique(rows, axis=0).shape[0]
        fr.append((rows.shape[0], uniq))
    print(f"after +{warm} waves: pending/unique per wave: {fr}  dup fraction {1 - sum(u for _, u in fr) / max(1, sum(n for n, _ in fr)):.3f}", flush=True)
