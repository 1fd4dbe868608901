#!/bin/bash
# evaluation cache: table size and evaluation-free-simulation cap, one box session
for cfg in "$@"; do
  read -r c cap conf <<< "$cfg"
  out=$(timeout -k 10 400 python bench.py --config ${conf:-connect4} --steps 4 --warmup 3 --no-cpu-baseline --cache-leg 0 --eval-cache $c --max-tree-sims $cap 2>/dev/null | tail -1)
  echo "cache=$c cap=$cap ${conf:-connect4} $(echo "$out" | python -c "import sys,json; j=json.loads(sys.stdin.read()); d=j['detail']; print(round(j['value']), 'pos/s hit', round(d['eval_cache_hits']/max(1,d['evaluator_calls']),3), 'tree', round(d['ms_tree_kernel_per_wave'],4), 'eval', round(d['ms_evaluator_per_wave'],4))")"
done
