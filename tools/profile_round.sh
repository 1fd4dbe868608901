#!/bin/bash
# One GPU session's worth of measurements for a round: bench lines of the three configs, the rocprofv3 kernel statistics of the
# headline command and the two PMC passes (FETCH_SIZE / WRITE_SIZE in separate passes, kernel trace only) behind bench.py's
# `roofline.traffic`.  usage (on the GPU box, from the repo root): tools/profile_round.sh <tag> [prof]   -> gpurun_out/<tag>/   (prof: profiler passes only)
set -o pipefail
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ "$2" != "prof" ]; then
echo "== bench connect4"; timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
tail -c 600 $out/bench.json; echo
echo "== bench gumbel"; timeout -k 10 300 python bench.py --config gumbel --no-cpu-baseline > $out/bench_gumbel.json 2> $out/bench_gumbel.err || exit 1
echo "== bench gomoku"; timeout -k 10 300 python bench.py --config gomoku --no-cpu-baseline > $out/bench_gomoku.json 2> $out/bench_gomoku.err || exit 1
fi
echo "== rocprofv3 --kernel-trace --stats (headline command, 2 steps)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 > $out/stats.log 2>&1 || exit 1
echo "== rocprofv3 --pmc FETCH_SIZE (evaluator probe + 40 waves of the engine)"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_f -o run -- python bench.py --steps 1 --warmup 0 --waves-per-step 40 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 > $out/pmc_f.log 2>&1 || exit 1
echo "== rocprofv3 --pmc WRITE_SIZE"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_w -o run -- python bench.py --steps 1 --warmup 0 --waves-per-step 40 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 > $out/pmc_w.log 2>&1 || exit 1
find $out -name "*.csv" | head -20
echo done
echo "== rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (MFMA utilisation)"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_m -o run -- python bench.py --steps 1 --warmup 0 --waves-per-step 40 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 > $out/pmc_m.log 2>&1 || exit 1
echo "== rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/pmc_l -o run -- python bench.py --steps 1 --warmup 0 --waves-per-step 40 --no-cpu-baseline --cache-leg 0 --ref-convention-leg 0 > $out/pmc_l.log 2>&1 || exit 1
echo done2
