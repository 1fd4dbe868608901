#!/usr/bin/env python3
"""Time series of the headline workload after bench.py's equilibrium start: plies, evaluator calls and finished games per 400 waves.  GPU."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET
from grok_alpha_zero_amd.net import NETS
G, sims = 4096, 200
net = NETS["Connect4"](6, seed=0).eval(); w = net.export_engine_weights()
def mk(**kw):
    e = SelfPlayEngine("Connect4", G, sims, 42, 8, 7, 2.5, 0.5, seed=kw.pop("seed", 1234), evaluator=EVAL_RESNET, net_blocks=6, **kw); e.load_weights(w); return e
mode = sys.argv[1] if len(sys.argv) > 1 else "equilibrium"
rng = np.random.default_rng(977)
e = mk(ring_capacity=0, seed=int(os.environ.get("PROBE_SEED", "1234")))
if mode == "equilibrium":
    hs = bench.selfplay_histories(lambda: mk(seed=4321, ring_capacity=0), G, int(42 * 1.1 * sims), rng)
    print("start plies: mean %.1f  hist %s" % (np.mean([len(h) for h in hs]), np.bincount([len(h) for h in hs], minlength=42).tolist()))
    for slot, h in enumerate(hs):
        if h: e.set_position(slot, h)
e.synchronize()
if mode == "equilibrium":
    back = e.read_positions()
    bad = sum(1 for a, b in zip(hs, back) if a != b)
    print("set_position check: %d of %d slots differ from what was requested" % (bad, G))
    e.run_waves(1); e.synchronize()
    back = e.read_positions()
    bad = sum(1 for a, b in zip(hs, back) if a != b[:len(a)])
    print("after one wave: %d slots whose history does not start with the requested one" % bad)
s0 = e.stats()
for step in range(int(sys.argv[2]) if len(sys.argv) > 2 else 50):
    e.run_waves(400); e.synchronize(); s1 = e.stats()
    dp, de, dg = int(s1["plies"] - s0["plies"]), int(s1["evals"] - s0["evals"]), int(s1["game_stats"][2] - s0["game_stats"][2])
    n = np.array([len(h) for h in e.read_positions()])
    hist = np.bincount(np.minimum(n // 6, 6), minlength=7)
    print("step %2d: plies %6d  evals %8d  evals/ply %6.1f  games finished %5d   ply of the games: mean %4.1f  [0-5 6-11 .. 36-41]: %s" % (step, dp, de, de / max(dp, 1), dg, n.mean(), hist.tolist()))
    s0 = s1
