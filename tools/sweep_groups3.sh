#!/bin/bash
# Gomoku: game groups x heads on high-priority streams (GAZ_HEADS_PRIORITY), one box
out=gpurun_out/${1:-groups3}; mkdir -p $out
run() {  # name, config, env..., -- args...
  local name=$1 cfg=$2; shift 2
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 250 python bench.py --config $cfg --other-configs 0 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 "$@" > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; return 1; }
  python - <<PY
import json
d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1])
t=d["detail"]
print("$name: %.0f pos/s  %.3fM evals/s  evals/pos %.1f  wave %.1f us  groups %s fused %s" % (d["value"], t["evals_per_s"]/1e6, t["evals_per_position"], d["ms_per_step"]/d["config"]["waves_per_step"]*1e3, t.get("game_groups"), t.get("fused_tree_and_trunk_launch")))
PY
}
for g in 1 2; do for p in 0 1; do
  run gmk_g${g}_p$p gomoku GAZ_HEADS_PRIORITY=$p -- --steps 3 --warmup 1 --game-groups $g || exit 1
done; done
run gmk_g4_p1 gomoku GAZ_HEADS_PRIORITY=1 -- --steps 3 --warmup 1 --game-groups 4 || exit 1
