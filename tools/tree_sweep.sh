#!/bin/bash
# tree-kernel register / occupancy variants (build_variants/libgaz_wpeN.so), both searches, one box session
for rep in 1 2; do
  for w in 2 3 4; do
    for cfg in connect4 gumbel; do
      out=$(GAZ_ENGINE_LIB=build_variants/libgaz_wpe$w.so timeout -k 10 300 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
      echo "wpe=$w $cfg $(echo "$out" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']), 'pos/s tree', round(j['detail']['ms_tree_kernel_per_wave'],4), 'eval', round(j['detail']['ms_evaluator_per_wave'],4))")"
    done
  done
done
