"""Evaluator-only probe for rocprofv3 passes: builds the Connect4 net, loads it into the engine and runs the
forward pass `repeats` times on a full batch (Compute_Speed.py-style, /root/reference/Compute_Speed.py:40-63)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=4096)
ap.add_argument("--blocks", type=int, default=6)
ap.add_argument("--repeats", type=int, default=20)
a = ap.parse_args()
from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET
from grok_alpha_zero_amd.net import Connect4Net, flops_per_position
net = Connect4Net(a.blocks).eval()
eng = SelfPlayEngine("Connect4", a.games, 200, 42, 8, 7, 2.5, 0.5, seed=1, evaluator=EVAL_RESNET, net_blocks=a.blocks, ring_capacity=0)
eng.load_weights(net.export_engine_weights())
x = np.random.default_rng(0).integers(-1, 2, size=(a.games, 6, 7, 4)).astype(np.int8)
p, v, ms = eng.evaluate(x, repeats=a.repeats)
fl = flops_per_position(a.blocks)["total"] * a.games
print(f"forward {ms:.3f} ms/batch of {a.games}  = {fl / ms / 1e9:.1f} TFLOP/s  ({a.games / ms * 1e3:.0f} positions/s)")
