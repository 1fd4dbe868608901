#!/bin/bash
# Gumbel config (BASELINE configs[4]) on one box: fused launch with four games per wavefront / one game per wavefront / separate launches
out=gpurun_out/${1:-abg}; mkdir -p $out
run() {  # name, env...
  env "${@:2}" timeout -k 10 200 python bench.py --config gumbel --steps 8 --warmup 3 --no-cpu-baseline --cache-leg 0 > $out/$1.json 2> $out/$1.err || { tail -5 $out/$1.err; return 1; }
  python - <<PY
import json
d=json.loads(open("$out/$1.json").read().strip().splitlines()[-1])
r=d["roofline"]; t=d["detail"]
f=r.get("fused_launch") or {}
print("$1: %.0f pos/s  %.2fM evals/s  evals/pos %.1f  wave %.1f us  fused %.1f us  trunk %.1f us  tree(sep) %.1f us  faults %d" % (d["value"], t["evals_per_s"]/1e6, t["evals_per_position"], d["ms_per_step"]/d["config"]["waves_per_step"]*1e3, f.get("avg_launch_us", 0), r["avg_launch_us"], t["ms_tree_kernel_per_wave"]*1e3, t["fused_launch_faults"]))
PY
}
run fused_teams GAZ_X=1 && run fused_single GAZ_FUSE_GUMBEL_TEAMS=0 && run separate GAZ_FUSE_GUMBEL=0 && run fused_teams2 GAZ_X=1 && run separate2 GAZ_FUSE_GUMBEL=0
