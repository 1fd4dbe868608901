#!/bin/bash
# games per tree block of the fused launch (GAZ_FUSE_TREE_ROUNDS x 16), headline and Gumbel configs, one box
out=gpurun_out/${1:-rounds}; mkdir -p $out
run() {  # name, config, env...
  env "${@:3}" timeout -k 10 200 python bench.py --config $2 --steps 8 --warmup 3 --other-configs 0 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 > $out/$1.json 2> $out/$1.err || { tail -5 $out/$1.err; return 1; }
  python - <<PY
import json
d=json.loads(open("$out/$1.json").read().strip().splitlines()[-1])
r=d["roofline"]; t=d["detail"]
f=r.get("fused_launch") or {}
print("$1: %.0f pos/s  %.2fM evals/s  evals/pos %.1f  wave %.1f us  fused %.1f us  trunk %.1f us  faults %d" % (d["value"], t["evals_per_s"]/1e6, t["evals_per_position"], d["ms_per_step"]/d["config"]["waves_per_step"]*1e3, f.get("avg_launch_us", 0), r["avg_launch_us"], t["fused_launch_faults"]))
PY
}
for r in 1 2 4 8 1; do run c4_r$r connect4 GAZ_FUSE_TREE_ROUNDS=$r || exit 1; done
for r in 1 2 4 8 1; do run gum_r$r gumbel GAZ_FUSE_TREE_ROUNDS=$r || exit 1; done
