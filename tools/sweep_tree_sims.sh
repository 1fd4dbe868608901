#!/bin/bash
# headline config, fused launch: evaluation-free simulations per game and launch (--max-tree-sims) and the staggered start, back to back on one box
out=gpurun_out/${1:-sweep}; mkdir -p $out
for st in 1 0; do
for m in 1 2 3 4 6; do
  timeout -k 10 200 python bench.py --steps 8 --warmup 3 --stagger $st --max-tree-sims $m --other-configs 0 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 > $out/s${st}_m$m.json 2> $out/s${st}_m$m.err || exit 1
  python - <<PY
import json
d=json.loads(open("$out/s${st}_m$m.json").read().strip().splitlines()[-1])
r=d["roofline"]; t=d["detail"]
print("stagger $st max_tree_sims $m: %.0f pos/s  %.2fM evals/s  evals/pos %.1f  wave %.1f us  fused %.1f us  trunk %.1f us  tree(sep) %.1f us" % (d["value"], t["evals_per_s"]/1e6, t["evals_per_position"], d["ms_per_step"]/d["config"]["waves_per_step"]*1e3, r["fused_launch"]["avg_launch_us"], r["avg_launch_us"], t["ms_tree_kernel_per_wave"]*1e3))
PY
done
done
