#!/bin/bash
# HBM traffic of the trunk kernel per launch: the two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of the driver's
# command in its short form, summarised by tools/pmc_traffic.py.   usage (GPU box, repo root): tools/pmc_trunk_traffic.sh <tag>
tag=${1:-traffic}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
pm="--steps 1 --warmup 0 --waves-per-step 40 --burn-in-waves 400 --other-configs 0 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0"
for c in "f FETCH_SIZE" "w WRITE_SIZE"; do
  set -- $c; d=$1; shift
  timeout -k 10 300 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $out/pmc_$d -o run -- python bench.py $pm > $out/pmc_$d.log 2>&1 || exit 1
done
python tools/pmc_traffic.py $out/pmc_f/run_counter_collection.csv $out/pmc_w/run_counter_collection.csv k_trunk_mix > $out/trunk_traffic.json && grep "bytes" $out/trunk_traffic.json
