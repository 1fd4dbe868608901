#!/bin/bash
# headline value against the length of the timed region + the Gomoku config fused vs separate, one box
out=gpurun_out/${1:-stab}; mkdir -p $out
short="--other-configs 0 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0"
for k in 8 20 40; do
  timeout -k 10 300 python bench.py --steps $k --warmup 5 $short > $out/steps$k.json 2> $out/steps$k.err || { tail -3 $out/steps$k.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$out/steps$k.json").read().strip().splitlines()[-1]); t=d["detail"]
print("steps $k: %.0f pos/s  %.2fM evals/s  evals/pos %.1f  games finished %d" % (d["value"], t["evals_per_s"]/1e6, t["evals_per_position"], t["games_finished"]))
PY
done
for f in 1 0; do
  GAZ_FUSE_GOMOKU=$f timeout -k 10 300 python bench.py --config gomoku --steps 3 --warmup 1 --no-cpu-baseline --cache-leg 0 > $out/gmk_f$f.json 2> $out/gmk_f$f.err || { tail -3 $out/gmk_f$f.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$out/gmk_f$f.json").read().strip().splitlines()[-1]); t=d["detail"]; r=d["roofline"]
print("gomoku fused=$f: %.0f pos/s  %.3fM evals/s  evals/pos %.1f  wave %.1f us  dominant %.1f us  fused %s" % (d["value"], t["evals_per_s"]/1e6, t["evals_per_position"], d["ms_per_step"]/d["config"]["waves_per_step"]*1e3, r["avg_launch_us"], (r.get("fused_launch") or {}).get("avg_launch_us")))
PY
done
