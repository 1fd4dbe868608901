#!/bin/bash
# Gomoku: game groups 1 / 2 / 3 and evaluation-free simulations per launch with two groups, one box (bench start: random plies 0 .. 60)
out=gpurun_out/${1:-groups4}; mkdir -p $out
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
for g in 1 2 3 1 2; do run gmk_g${g}_$RANDOM gomoku X=1 -- --steps 3 --warmup 1 --game-groups $g || exit 1; done
for m in 16 64; do run gmk_g2_m$m gomoku X=1 -- --steps 3 --warmup 1 --game-groups 2 --max-tree-sims $m || exit 1; done
run gmk_g2_r4 gomoku GAZ_FUSE_TREE_ROUNDS=4 -- --steps 3 --warmup 1 --game-groups 2 || exit 1
run gmk_g2_r16 gomoku GAZ_FUSE_TREE_ROUNDS=16 -- --steps 3 --warmup 1 --game-groups 2 || exit 1
