#!/bin/bash
# Collect the round's bench lines and rocprofv3 summaries on the GPU box (run via gpurun from the
# repo root; results land in gpurun_out/prof/ and are copied into profiles/ by hand).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "default bench done"
timeout -k 10 300 python3 bench.py --batch 64 --steps 10 --warmup 2 --pool 2 --streams 2 --no-cpu-baseline > $O/bench_batch64.json 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --channels grad_hist_4_u1 --no-cpu-baseline > $O/bench_gh4u1.json 2>> $O/bench_default.err || exit 1
echo "bench lines done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_default --output-format csv -- python3 bench.py --no-cpu-baseline > $O/stats_default.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_streams1 --output-format csv -- python3 bench.py --no-cpu-baseline --streams 1 > $O/stats_streams1.log 2>&1 || exit 1
echo "kernel stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-graph --streams 1 > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-graph --streams 1 > $O/pmc_write.log 2>&1 || exit 1
echo "traffic passes done"
find $O -name "*.csv" | head -30
