#!/bin/bash
# game groups, follow-ups on one box: separate tree / trunk launches inside the groups, group count with separate launches
out=gpurun_out/${1:-groups2}; mkdir -p $out
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
run c4_g2_fused connect4 X=1 -- --steps 8 --warmup 2 || exit 1
run c4_g2_sep connect4 GAZ_FUSE_WAVE=0 -- --steps 8 --warmup 2 || exit 1
run c4_g3_sep connect4 GAZ_FUSE_WAVE=0 -- --steps 8 --warmup 2 --game-groups 3 || exit 1
run c4_g2_sep_m4 connect4 GAZ_FUSE_WAVE=0 -- --steps 8 --warmup 2 --max-tree-sims 4 || exit 1
run gum_g2_sep gumbel GAZ_FUSE_WAVE=0 -- --steps 6 --warmup 2 --game-groups 2 || exit 1
run gum_g1 gumbel X=1 -- --steps 6 --warmup 2 || exit 1
