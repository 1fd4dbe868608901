#!/bin/bash
# Round 3: one GPU session's worth of evidence, all from the DRIVER's command line (python bench.py --gpus 1 --steps 20 --warmup 5):
#   bench line (headline + gumbel + gomoku legs), value stability over --steps 8 / 20 / 40, rocprofv3 kernel statistics of the same command,
#   and the PMC passes (separate runs, --kernel-trace only): FETCH_SIZE, WRITE_SIZE, MFMA busy, LDS bank conflicts.
# usage (GPU box, repo root): tools/profile_round3.sh <tag>      -> gpurun_out/<tag>/
set -o pipefail
tag=${1:-r03}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
short="--other-configs 0 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0"
echo "== bench (driver command)"; timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || exit 1
for k in 8 40; do echo "== bench --steps $k"; timeout -k 10 300 python bench.py --steps $k --warmup 5 $short > $out/bench_steps$k.json 2> $out/bench_steps$k.err || exit 1; done
echo "== rocprofv3 --kernel-trace --stats (headline + legs, 3 steps)"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 > $out/stats.log 2>&1 || exit 1
pm="--steps 1 --warmup 0 --waves-per-step 40 --burn-in-waves 400 $short"
for c in "f FETCH_SIZE" "w WRITE_SIZE" "m SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "l SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  set -- $c; d=$1; shift
  echo "== rocprofv3 --pmc $*"
  timeout -k 10 300 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $out/pmc_$d -o run -- python bench.py $pm > $out/pmc_$d.log 2>&1 || exit 1
done
find $out -name "*.csv" | head -30
echo done
