#!/bin/bash
# Gomoku: wall-clock stamps of one fused launch (one batch) -> timeline + per-phase cycles of the trunk workgroups
out=gpurun_out/${1:-gmkst}; mkdir -p $out
GAZ_FUSED_STAMPS=$out/fused.bin:300 timeout -k 10 250 python bench.py --config gomoku --steps 1 --warmup 0 --game-groups 1 --other-configs 0 --no-cpu-baseline --cache-leg 0 > /dev/null 2> $out/err.log || { tail -3 $out/err.log; exit 1; }
python tools/fused_timeline.py $out/fused.bin 32
python - <<PY
import numpy as np
raw = np.fromfile("$out/fused.bin", dtype=np.uint64).reshape(-1, 128).astype(np.int64)
tr = raw[32:]; tr = tr[tr[:, 0] > 0]
cyc = tr[:, 64:]; st = tr[:, :64]
def ph(name, a, b):
    ok = (cyc[:, a] > 0) & (cyc[:, b] > 0)
    d = (cyc[ok, b] - cyc[ok, a]); w = (st[ok, b] - st[ok, a]) / 100.0
    print(f"  {name:<26s} {d.mean():9.0f} cycles  {w.mean():7.2f} us  (n={ok.sum()})")
ph("wait (0 -> 1)", 0, 1); ph("stem + block-0 operand (1 -> 2)", 1, 2)
for b in range(10):
    base = 3 + 6 * b
    if (cyc[:, base + 5] > 0).any(): ph(f"block {b} (prev -> +5)", 2 if b == 0 else base - 1, base + 5)
ph("heads / copy out (56 -> 63)", 56, 63)
for a_, b_, n_ in ((56, 57, "gh: setup"), (57, 58, "gh: barrier"), (58, 59, "gh: preact + barrier"), (59, 60, "gh: policy taps"), (60, 61, "gh: policy out + value taps"), (61, 63, "gh: value out")):
    if (cyc[:, b_] > 0).all() and (cyc[:, a_] > 0).all(): ph(n_, a_, b_); ph("whole (0 -> 63)", 0, 63)
PY
