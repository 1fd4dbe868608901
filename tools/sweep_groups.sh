#!/bin/bash
# game groups (gaz_engine_config::game_groups) on one box: headline config, max_tree_sims x tree rounds per block with two groups; the cache leg and Gumbel grouped
out=gpurun_out/${1:-groups}; mkdir -p $out
run() {  # name, config, env..., -- args...
  local name=$1 cfg=$2; shift 2
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 250 python bench.py --config $cfg --other-configs 0 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 "$@" > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; return 1; }
  python - <<PY
import json
d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1])
t=d["detail"]
print("$name: %.0f pos/s  %.3fM evals/s  evals/pos %.1f  wave %.1f us  groups %s" % (d["value"], t["evals_per_s"]/1e6, t["evals_per_position"], d["ms_per_step"]/d["config"]["waves_per_step"]*1e3, t.get("game_groups")))
PY
}
for m in ${MTS:-8 12 16}; do for r in ${ROUNDS:-1 2}; do
  run c4_m${m}_r$r connect4 GAZ_FUSE_TREE_ROUNDS=$r -- --steps 8 --warmup 2 --max-tree-sims $m || exit 1
done; done
run c4_cache_g1 connect4 X=1 -- --steps 8 --warmup 2 --eval-cache 24 --game-groups 1 || exit 1
run c4_cache_g2 connect4 X=1 -- --steps 8 --warmup 2 --eval-cache 24 --game-groups 2 || exit 1
run gum_g1 gumbel X=1 -- --steps 6 --warmup 2 --game-groups 1 || exit 1
run gum_g2 gumbel X=1 -- --steps 6 --warmup 2 --game-groups 2 || exit 1
