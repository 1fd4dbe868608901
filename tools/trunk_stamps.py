"""Summarise the phase stamps k_trunk writes under GAZ_TRUNK_STAMPS=<file> (csrc/trunk.hpp: wall clock in 100 MHz ticks and
shader clock of wave 0 of every workgroup).  usage: trunk_stamps.py <file> [blocks]"""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 128).astype(np.int64)
nb = min(int(sys.argv[2]) if len(sys.argv) > 2 else 6, 10)
st, cyc = raw[:, :64], raw[:, 64:]
t0 = st[:, 0].min()
us = (st - t0) / 100.0
start, end = us[:, 0], us[:, 63]
dur = end - start
print(f"workgroups {len(st)}  kernel span {end.max():.1f} us   per-WG duration mean {dur.mean():.1f} us (min {dur.min():.1f}, max {dur.max():.1f})")
dc = (cyc[:, 63] - cyc[:, 0]).astype(float)
print(f"shader clock while the workgroups run: {np.mean(dc / dur):.0f} MHz (p10 {np.percentile(dc / dur, 10):.0f}, p90 {np.percentile(dc / dur, 90):.0f})")
order = np.argsort(start)
rounds = [order[:512], order[512:1024], order[1024:]]
for i, r in enumerate(rounds):
    if len(r):
        print(f"  round {i}: {len(r)} workgroups, start {start[r].mean():.1f} us, duration {dur[r].mean():.1f} us, "
              f"{np.mean(dc[r] / dur[r]):.0f} MHz, {dc[r].mean():.0f} cycles")


def phase(name, i0, i1):
    d = (cyc[:, i1] - cyc[:, i0]).astype(float)
    w = us[:, i1] - us[:, i0]
    print(f"  {name:<28s} {d.mean():8.0f} cycles  (p10 {np.percentile(d, 10):7.0f}  p90 {np.percentile(d, 90):7.0f})  {w.mean():6.2f} us")
    return d.mean()


phase("image DMA wait", 0, 1)
phase("block 0 operand", 1, 2)
if (cyc[:, 58] > 0).all():                                 # stem inside the launch: its sub-phases (wave 0)
    phase("  stem: planes + MFMA tile 0", 1, 56)
    phase("  stem: planes + MFMA tile 1", 56, 57)
    phase("  stem: GELU epilogue", 57, 58)
    phase("  stem: barrier", 58, 2)
names = ["conv1 taps", "barrier", "h write + barrier", "conv2 taps", "barrier", "epilogue + barrier"]
tot = np.zeros(6)
for b in range(nb):
    prev = 2 if b == 0 else 8 + 6 * (b - 1)
    for k in range(6):
        cur = 3 + 6 * b + k
        tot[k] += (cyc[:, cur] - cyc[:, prev]).astype(float).mean()
        prev = cur
print("per block (mean over blocks and workgroups), shader cycles; pure MFMA of one conv = 9 taps x 8 k-steps x 2 TM x 2 x 32 = 9216 at TM = 2:")
for k in range(6):
    print(f"  {names[k]:<28s} {tot[k] / nb:8.0f}")
print(f"  {'block total':<28s} {tot.sum() / nb:8.0f}")
phase("copy out", 8 + 6 * (nb - 1), 63)
