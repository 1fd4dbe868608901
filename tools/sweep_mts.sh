#!/bin/bash
# evaluation-free simulations per game and launch (--max-tree-sims), current fused launches, one box: headline, Gumbel, Gomoku
out=gpurun_out/${1:-mts}; mkdir -p $out
run() {  # name, config, args...
  timeout -k 10 250 python bench.py --config $2 --other-configs 0 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 "${@:3}" > $out/$1.json 2> $out/$1.err || { tail -5 $out/$1.err; return 1; }
  python - <<PY
import json
d=json.loads(open("$out/$1.json").read().strip().splitlines()[-1])
r=d["roofline"]; t=d["detail"]; f=r.get("fused_launch") or {}
print("$1: %.0f pos/s  %.3fM evals/s  evals/pos %.1f  wave %.1f us  fused %.1f us" % (d["value"], t["evals_per_s"]/1e6, t["evals_per_position"], d["ms_per_step"]/d["config"]["waves_per_step"]*1e3, f.get("avg_launch_us", 0)))
PY
}
C4=${C4:-"4 3 6 8 12"}; GUM=${GUM:-"4 2 8"}; GMK=${GMK:-"4 2 8"}
for m in $C4; do run c4_m$m connect4 --steps 8 --warmup 2 --max-tree-sims $m || exit 1; done
for m in $GUM; do run gum_m$m gumbel --steps 6 --warmup 2 --max-tree-sims $m || exit 1; done
for m in $GMK; do run gmk_m$m gomoku --steps 3 --warmup 1 --max-tree-sims $m || exit 1; done
