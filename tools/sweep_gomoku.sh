#!/bin/bash
out=gpurun_out/${1:-gmk}; mkdir -p $out
run() {  # name, args...
  timeout -k 10 250 python bench.py --config gomoku --steps 3 --warmup 1 --no-cpu-baseline --cache-leg 0 "${@:2}" > $out/$1.json 2> $out/$1.err || { tail -5 $out/$1.err; return 1; }
  python - <<PY
import json
d=json.loads(open("$out/$1.json").read().strip().splitlines()[-1])
r=d["roofline"]; t=d["detail"]
print("$1: %.0f pos/s  %.3fM evals/s  evals/pos %.1f  wave %.1f us  trunk %.1f us (frac %.3f)  tree %.1f us  evaluator %.1f us" % (d["value"], t["evals_per_s"]/1e6, t["evals_per_position"], d["ms_per_step"]/d["config"]["waves_per_step"]*1e3, r["avg_launch_us"], r["frac"], t["ms_tree_kernel_per_wave"]*1e3, t["ms_evaluator_per_wave"]*1e3))
PY
}
run s1_m4 --stagger 1 && run s1_m1 --stagger 1 --max-tree-sims 1 && run s1_m2 --stagger 1 --max-tree-sims 2 && run s0_m4 --stagger 0 && run s0_m1 --stagger 0 --max-tree-sims 1
