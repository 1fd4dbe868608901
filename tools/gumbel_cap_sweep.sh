#!/bin/bash
for cap in 1 2 4 8; do
  out=$(timeout -k 10 300 python bench.py --config gumbel --steps 2 --warmup 1 --no-cpu-baseline --max-tree-sims $cap 2>/dev/null | tail -1)
  echo "cap=$cap $(echo "$out" | python -c "import sys,json; j=json.loads(sys.stdin.read()); d=j['detail']; print(round(j['value']), 'pos/s tree', round(d['ms_tree_kernel_per_wave'],4), 'eval', round(d['ms_evaluator_per_wave'],4), 'evals/pos', round(d['evals_per_position'],2), 'sims/s', round(d['sims_per_s']))")"
done
