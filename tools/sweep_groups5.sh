#!/bin/bash
# Gomoku with two game groups: tree rounds per block; and the all-empty-boards start (the opening of a generation) with one and two groups
out=gpurun_out/${1:-groups5}; mkdir -p $out
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
for r in 2 3 4 6; do run gmk_g2_r$r gomoku GAZ_FUSE_TREE_ROUNDS=$r -- --steps 3 --warmup 1 --game-groups 2 || exit 1; done
run gmk_g1_r4 gomoku GAZ_FUSE_TREE_ROUNDS=4 -- --steps 3 --warmup 1 --game-groups 1 || exit 1
run gmk_empty_g1 gomoku X=1 -- --steps 3 --warmup 1 --game-groups 1 --stagger 0 || exit 1
run gmk_empty_g2_r4 gomoku GAZ_FUSE_TREE_ROUNDS=4 -- --steps 3 --warmup 1 --game-groups 2 --stagger 0 || exit 1
run gmk_empty_g2_r8 gomoku X=1 -- --steps 3 --warmup 1 --game-groups 2 --stagger 0 || exit 1
