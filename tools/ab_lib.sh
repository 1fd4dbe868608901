#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_lib.sh <tag> <libA> <libB> [config ...]
out=gpurun_out/$1; mkdir -p $out; A=$2; B=$3; shift 3
run() {  # name, lib, config
  GAZ_ENGINE_LIB=$2 timeout -k 10 250 python bench.py --config $3 --steps 8 --warmup 2 --other-configs 0 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 > $out/$1.json 2> $out/$1.err || { tail -5 $out/$1.err; return 1; }
  python - <<PY
import json
d=json.loads(open("$out/$1.json").read().strip().splitlines()[-1])
r=d["roofline"]; t=d["detail"]; f=r.get("fused_launch") or {}
print("$1: %.0f pos/s  %.3fM evals/s  wave %.1f us  fused %.1f us  dominant %.1f us" % (d["value"], t["evals_per_s"]/1e6, d["ms_per_step"]/d["config"]["waves_per_step"]*1e3, f.get("avg_launch_us", 0), r["avg_launch_us"]))
PY
}
for c in "${@:-connect4}"; do
  run ${c}_A $PWD/$A $c && run ${c}_B $PWD/$B $c && run ${c}_A2 $PWD/$A $c && run ${c}_B2 $PWD/$B $c || exit 1
done
