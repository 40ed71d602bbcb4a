#!/bin/bash
# Collect the round's bench lines and rocprofv3 summaries on the GPU box (run via gpurun from the repo root):
#   tools/collect_profiles.sh gpurun_out/prof    -> copy the summaries you want judged into profiles/rNN/ by hand.
# Counter passes use --kernel-trace + --pmc only (no other trace domains).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=${1:-gpurun_out/prof}
PART=${PART:-12}
mkdir -p $O
if [[ $PART == *1* ]]; then
timeout -k 10 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
echo "default bench done"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --batch 64 --steps 10 --warmup 2 --pool 2 --streams 2 --no-cpu-baseline --no-config3 > $O/bench_batch64.json 2>> $O/bench_default.err || exit 1
timeout -k 10 400 python3 bench.py --config 5 --steps 20 --warmup 5 > $O/bench_cfg5.json 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --config 5 --batch 16 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_batch16.json 2>> $O/bench_default.err || exit 1
WB_FORCE_RANK16=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-config3 --no-through-api > $O/bench_default_rank16.json 2>> $O/bench_default.err || exit 1
WB_FORCE_RANK16=1 timeout -k 10 300 python3 bench.py --batch 64 --steps 10 --warmup 2 --pool 2 --streams 2 --no-cpu-baseline --no-config3 --no-through-api > $O/bench_batch64_rank16.json 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --channels grad_hist_4_u1 --no-cpu-baseline > $O/bench_grad_hist_4_u1.json 2>> $O/bench_default.err || exit 1
WB_NO_RANKS=1 timeout -k 10 300 python3 bench.py --batch 64 --steps 10 --warmup 2 --pool 2 --streams 2 --no-cpu-baseline --no-config3 > $O/bench_batch64_float_channels.json 2>> $O/bench_default.err || exit 1
WB_NO_RANKS=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-config3 > $O/bench_default_float_channels.json 2>> $O/bench_default.err || exit 1
WB_CASC_JIT=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-config3 --no-jit > $O/bench_default_generic_cascade.json 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --no-cpu-baseline --steps 20 --warmup 5 2>> $O/bench_default.err | grep "^{" > $O/bench_2ranks_gloo_selflaunch.json || exit 1   # (gloo prints a connection banner on stdout)
timeout -k 10 300 python3 tools/bench_next_rows.py > $O/bench_next_rows.txt 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 tools/detect_breakdown.py > $O/model_detect_host_timeline.txt 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 tools/pinned_probe.py > $O/pinned_probe.txt 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --force-collective --no-cpu-baseline --no-config3 --steps 20 --warmup 5 > $O/bench_1rank_rccl_collective.json 2>> $O/bench_default.err || exit 1
echo "bench lines done"
fi
if [[ $PART == *2* ]]; then
S="--no-cpu-baseline --no-through-api --no-config3 --repeats 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_default --output-format csv -- python3 bench.py $S > $O/stats_default.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_streams1 --output-format csv -- python3 bench.py $S --streams 1 > $O/stats_streams1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_batch64 --output-format csv -- python3 bench.py $S --batch 64 --steps 10 --warmup 2 --pool 2 --streams 1 > $O/stats_batch64.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_cfg5 --output-format csv -- python3 bench.py $S --config 5 --steps 10 --warmup 2 --pool 2 --streams 1 > $O/stats_cfg5.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_model_detect --output-format csv -- python3 tools/detect_breakdown.py > $O/stats_model_detect.log 2>&1 || exit 1
echo "kernel stats done"
T="--no-cpu-baseline --no-through-api --no-config3 --repeats 1 --steps 8 --warmup 2 --no-graph --streams 1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 bench.py $T > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 bench.py $T > $O/pmc_write.log 2>&1 || exit 1
python3 tools/traffic_summary.py $O/pmc_fetch $O/pmc_write > $O/traffic_pmc.json
rm -rf $O/pmc_fetch $O/pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 bench.py $T --config 5 --pool 1 > $O/pmc_fetch5.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 bench.py $T --config 5 --pool 1 > $O/pmc_write5.log 2>&1 || exit 1
python3 tools/traffic_summary.py $O/pmc_fetch $O/pmc_write | sed 's/traffic_bytes_per_launch_b1/traffic_bytes_per_launch/' > $O/traffic_pmc_cfg5.json
echo "traffic passes done"
for d in stats_default stats_streams1 stats_batch64 stats_cfg5 stats_model_detect; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
done
rm -rf $O/stats_default $O/stats_streams1 $O/stats_batch64 $O/stats_cfg5 $O/stats_model_detect $O/pmc_fetch $O/pmc_write
fi
if [[ $PART == *3* ]]; then
tools/collect_sq.sh $O/sq > /dev/null 2>&1 && cp $O/sq/sq_counters.txt $O/sq/sq_counters.json $O/ && rm -rf $O/sq
tools/collect_sq.sh $O/sq5 --config 5 > /dev/null 2>&1 && cp $O/sq5/sq_counters.txt $O/sq_counters_cfg5.txt && rm -rf $O/sq5
tools/phase_insts.sh $O/ph > $O/phase_instruction_counts.txt 2>&1; rm -rf $O/ph
fi
ls -la $O
