"""Probe: K engines of G / K games each, free-running on their own streams, against ONE engine of G games (evaluator calls per second).
Games are independent of how they are grouped (rows of a batch are bit-independent), so K phase-shifted fused launches in flight could hide
the heads and the start of each other's tree step.  usage: python tools/two_engines_probe.py [config] [G] [K] [waves] [chunk]"""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET, SEARCH_GUMBEL, SEARCH_PUCT
from grok_alpha_zero_amd.net import NETS

config = sys.argv[1] if len(sys.argv) > 1 else "connect4"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = sys.argv[3] if len(sys.argv) > 3 else "2"          # number of equal groups, or explicit sizes "2560,1536"
waves = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 50
burn = int(sys.argv[6]) if len(sys.argv) > 6 else 200
sizes = [int(v) for v in K.split(",")] if "," in K else [G // int(K)] * int(K)
assert sum(sizes) <= G
gumbel = config == "gumbel"
game, blocks, sims = ("Gomoku", 10, 400) if config == "gomoku" else ("Connect4", 6, 32 if gumbel else 200)
net = NETS[game](blocks, seed=0, policy_head="linear" if gumbel else "softmax").eval()
w = net.export_engine_weights()


def mk(n, off):
    e = SelfPlayEngine(game, n, sims, 225 if game == "Gomoku" else 42, 8, 7, 4.5 if game == "Gomoku" else 2.5, 0.05 if game == "Gomoku" else 0.5,
                       seed=1234, slot_offset=off, evaluator=EVAL_RESNET, net_blocks=blocks, ring_capacity=0,
                       search=SEARCH_GUMBEL if gumbel else SEARCH_PUCT, gumbel_m=7, c_visit=50.0, c_scale=1.0, policy_is_logits=gumbel)
    e.load_weights(w)
    return e


def run(engs, label):
    for i in range(0, burn, 200):
        for e in engs:
            e.run_waves(200)
        for e in engs:
            e.synchronize()
    s0 = [e.stats() for e in engs]
    t0 = time.perf_counter()
    for i in range(0, waves, chunk):                      # enqueue in small chunks so that the engines' launches interleave in time
        for e in engs:
            e.run_waves(chunk)
    for e in engs:
        e.synchronize()
    dt = time.perf_counter() - t0
    s1 = [e.stats() for e in engs]
    ev = sum(b["evals"] - a["evals"] for a, b in zip(s0, s1))
    pos = sum(b["plies"] - a["plies"] for a, b in zip(s0, s1))
    fa = sum(b["fused_faults"] for b in s1)
    print(f"{label}: {ev / dt / 1e6:.3f} M evals/s  {pos / dt:.0f} pos/s  {dt / waves * 1e6:.1f} us per wave of all engines  faults {fa}", flush=True)


one = [mk(G, 0)]
run(one, f"1 x {G}")
one[0].close()
many = [mk(n, sum(sizes[:i])) for i, n in enumerate(sizes)]
run(many, "groups " + " | ".join(str(n) for n in sizes))
for e in many:
    e.close()
