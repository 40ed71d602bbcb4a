#!/bin/bash
# rocprofv3 kernel-trace stats of the three bench shapes (B=1 one stream, B=64, configs[4]) -> OUTDIR/*_kernel_stats.csv
# usage (GPU box, repo root): tools/quick_stats.sh OUTDIR
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$1; mkdir -p $O
S="--no-cpu-baseline --no-through-api --no-config3 --repeats 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_streams1 --output-format csv -- python3 bench.py $S --streams 1 > $O/stats_streams1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_batch64 --output-format csv -- python3 bench.py $S --batch 64 --steps 10 --warmup 2 --pool 2 --streams 1 > $O/stats_batch64.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_cfg5 --output-format csv -- python3 bench.py $S --config 5 --steps 10 --warmup 2 --pool 2 --streams 1 > $O/stats_cfg5.log 2>&1 || exit 1
for d in stats_streams1 stats_batch64 stats_cfg5; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
  rm -rf $O/$d
done
head -6 $O/*_kernel_stats.csv
