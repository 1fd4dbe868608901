#!/bin/bash
# Gumbel (8192 games) with two game groups: tree rounds per block, evaluation-free simulations per launch; one box
out=gpurun_out/${1:-groups6}; mkdir -p $out
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
run gum_g1 gumbel X=1 -- --steps 6 --warmup 2 --game-groups 1 || exit 1
for r in 1 2 8; do run gum_g2_r$r gumbel GAZ_FUSE_TREE_ROUNDS=$r -- --steps 6 --warmup 2 --game-groups 2 || exit 1; done
for m in 8 32; do run gum_g2_m$m gumbel X=1 -- --steps 6 --warmup 2 --game-groups 2 --max-tree-sims $m || exit 1; done
run gum_g2_noteams gumbel GAZ_FUSE_GUMBEL_TEAMS=0 -- --steps 6 --warmup 2 --game-groups 2 || exit 1
run gmk_auto gomoku X=1 -- --steps 3 --warmup 1 || exit 1
