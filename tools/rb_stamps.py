"""Summarise the phase stamps k_resblock writes under GAZ_RB_STAMPS=<file> (wall clock, 100 MHz ticks, wave 0 of each workgroup)."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 64).astype(np.int64)
st, cyc = raw[:, :32], raw[:, 32:]
t0 = st[:, 0].min()
us = (st - t0) / 100.0
names = ["start", "dma landed", "transform"] + [f"tap{t}" for t in range(18)] + ["h written", "Ct written", "end"]
start, end = us[:, 0], us[:, 23]
print(f"workgroups {len(st)}  kernel span {end.max():.1f} us   per-WG duration mean {np.mean(end - start):.1f} us (min {np.min(end - start):.1f}, max {np.max(end - start):.1f})")
order = [0, 1, 2] + list(range(3, 12)) + [21] + list(range(12, 21)) + [22, 23]
first = start < 2.0                                  # workgroups of the first round (all start together)
prev = None
for i in order:
    if prev is not None:
        d = us[:, i] - us[:, prev]
        print(f"  {names[prev]:>11s} -> {names[i]:<11s} mean {d.mean():6.2f} us   p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}"
              f"   | first round {d[first].mean():6.2f}   later {d[~first].mean() if (~first).any() else 0:6.2f}")
    prev = i
print(f"per-WG duration: first round {np.mean((end - start)[first]):.1f} us, later {np.mean((end - start)[~first]) if (~first).any() else 0:.1f} us")
h, _ = np.histogram(start, bins=np.arange(0, end.max() + 5, 5.0))
print("WG starts per 5 us:", h.tolist())
dc = (cyc[:, 23] - cyc[:, 0]).astype(float); dt = (end - start)
print(f"shader clock while the workgroups run: {np.mean(dc / dt):.0f} MHz (p10 {np.percentile(dc / dt, 10):.0f}, p90 {np.percentile(dc / dt, 90):.0f})")
tapc = (cyc[:, 20] - cyc[:, 19]).astype(float)
print(f"last tap (no weight DMA): {tapc.mean():.0f} cycles; a full tap (tap13->14): {(cyc[:, 17] - cyc[:, 16]).mean():.0f} cycles; pure MFMA = 2048")
