#!/bin/bash
# A/B on one box, headline config: the fused launch with the completion queue (default) vs a flag per board (GAZ_FUSE_QUEUE=0), each with a
# stamp timeline of one launch (tools/fused_timeline.py).  usage: tools/ab_fused.sh <tag> [extra bench args]
out=gpurun_out/${1:-ab}; mkdir -p $out; shift
run() {  # name, queue
  GAZ_FUSE_QUEUE=$2 GAZ_FUSED_STAMPS=$out/$1.stamps:2500 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --other-configs 0 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 "${@:3}" > $out/$1.json 2> $out/$1.err || { tail -5 $out/$1.err; return 1; }
  python - <<PY
import json
d=json.loads(open("$out/$1.json").read().strip().splitlines()[-1])
r=d["roofline"]; t=d["detail"]
print("$1: %.0f pos/s  %.2fM evals/s  evals/pos %.1f  wave %.1f us  fused %.1f us  trunk %.1f us  tree(sep) %.1f us  faults %d" % (d["value"], t["evals_per_s"]/1e6, t["evals_per_position"], d["ms_per_step"]/d["config"]["waves_per_step"]*1e3, r["fused_launch"]["avg_launch_us"], r["avg_launch_us"], t["ms_tree_kernel_per_wave"]*1e3, t["fused_launch_faults"]))
PY
  python tools/fused_timeline.py $out/$1.stamps 256 | grep -v "^   compute\|dispatch round"
}
run queue 1 "$@" && run flags 0 "$@" && run queue2 1 "$@" && run flags2 0 "$@"
