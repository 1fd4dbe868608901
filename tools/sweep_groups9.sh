#!/bin/bash
# Gumbel, two game groups (two tree rounds per block, 3-board tiles): evaluation-free simulations per launch; Connect4 PUCT the same; one box
out=gpurun_out/${1:-groups9}; mkdir -p $out
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
for m in 6 8 12 16 24; do run gum_m$m gumbel X=1 -- --steps 6 --warmup 2 --max-tree-sims $m || exit 1; done
for m in 8 12 16 24; do run c4_m$m connect4 X=1 -- --steps 8 --warmup 2 --max-tree-sims $m || exit 1; done
