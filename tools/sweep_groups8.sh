#!/bin/bash
# number of game groups with the all-3-board-tile plan of a shared chip (Connect4 PUCT and Gumbel), one box
out=gpurun_out/${1:-groups8}; mkdir -p $out
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
for g in 2 3 4; do run c4_g$g connect4 X=1 -- --steps 8 --warmup 2 --game-groups $g || exit 1; done
for g in 2 3 4; do run gum_g$g gumbel X=1 -- --steps 6 --warmup 2 --game-groups $g || exit 1; done
run gum_g3_r1 gumbel GAZ_FUSE_TREE_ROUNDS=1 -- --steps 6 --warmup 2 --game-groups 3 || exit 1
run gum_g2_r1 gumbel GAZ_FUSE_TREE_ROUNDS=1 -- --steps 6 --warmup 2 --game-groups 2 || exit 1
