#!/usr/bin/env python3
"""Length distribution of self-play games at the headline configuration (Connect4, 200 simulations per move, 6-block random-init network) and
the stationary distribution of the ply a running game is at, P(p) ~ #{games longer than p} — what bench.py's staggered start samples from.
GPU; usage: python tools/game_length_hist.py [config] [n_games]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET, SEARCH_GUMBEL, SEARCH_PUCT
from grok_alpha_zero_amd.net import NETS
cfg = sys.argv[1] if len(sys.argv) > 1 else "connect4"
game, G, sims, blocks, max_actions, ef, es, cpuct, alpha, search, gm = bench.CONFIGS[cfg]
G = int(sys.argv[2]) if len(sys.argv) > 2 else G
gumbel = search == "gumbel"
net = NETS[game](blocks, seed=0, policy_head="linear" if gumbel else "softmax").eval()
e = SelfPlayEngine(game, G, sims, max_actions, ef, es, cpuct, alpha, seed=1234, evaluator=EVAL_RESNET, net_blocks=blocks, ring_capacity=4 * G,
                   search=SEARCH_GUMBEL if gumbel else SEARCH_PUCT, gumbel_m=gm, c_visit=50.0, c_scale=1.0, policy_is_logits=gumbel, games_budget=3 * G)
e.load_weights(net.export_engine_weights())
T = []
for _ in range(400):
    e.run_waves(400)
    T += [r["T"] for r in e.drain_finished(4 * G)]
    if len(T) >= 3 * G:
        break
T = np.array(T)
surv = np.array([(T > p).sum() for p in range(max_actions)], float)
print(json.dumps(dict(config=cfg, games=len(T), mean_length=float(T.mean()), hist=np.bincount(T, minlength=max_actions + 1).tolist(),
                      stationary_ply=(surv / surv.sum()).round(5).tolist())))
