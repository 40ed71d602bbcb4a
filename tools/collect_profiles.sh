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
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_flags.json 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --batch 64 --steps 10 --warmup 2 --pool 2 --streams 2 --no-cpu-baseline > $O/bench_batch64.json 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --channels grad_hist_4_u1 --no-cpu-baseline > $O/bench_grad_hist_4_u1.json 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --channels grad_hist_4_u1 --batch 64 --steps 10 --warmup 2 --pool 2 --streams 2 --no-cpu-baseline > $O/bench_grad_hist_4_u1_batch64.json 2>> $O/bench_default.err || exit 1
WB_NO_RANKS=1 timeout -k 10 300 python3 bench.py --batch 64 --steps 10 --warmup 2 --pool 2 --streams 2 --no-cpu-baseline > $O/bench_batch64_float_channels.json 2>> $O/bench_default.err || exit 1
WB_NO_RANKS=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench_default_float_channels.json 2>> $O/bench_default.err || exit 1
WB_CASC_JIT=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-jit > $O/bench_default_generic_cascade.json 2>> $O/bench_default.err || exit 1
WB_CASC_JIT=0 timeout -k 10 300 python3 bench.py --batch 64 --steps 10 --warmup 2 --pool 2 --streams 2 --no-cpu-baseline --no-jit > $O/bench_batch64_generic_cascade.json 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --no-cpu-baseline --steps 20 --warmup 5 2>> $O/bench_default.err | grep "^{" > $O/bench_2ranks_gloo_selflaunch.json || exit 1   # (gloo prints a connection banner on stdout)
timeout -k 10 300 python3 tools/bench_cfg5.py 4 > $O/bench_cfg5.txt 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 tools/bench_cfg5.py 16 >> $O/bench_cfg5.txt 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 tools/bench_next_rows.py > $O/bench_next_rows.txt 2>> $O/bench_default.err || exit 1
# (the two-model call once more with a read-back that holds the synthetic image's 4000+ detections per model)
WB_FETCH_ROWS=8192 timeout -k 10 300 python3 tools/bench_next_rows.py 2>> $O/bench_default.err | grep "waldboost.detect" >> $O/bench_next_rows.txt || exit 1
timeout -k 10 300 python3 tools/detect_breakdown.py > $O/model_detect_host_timeline.txt 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 tools/stream_profile.py 3 > $O/detect_stream_host_profile.txt 2>> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --force-collective --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_1rank_rccl_collective.json 2>> $O/bench_default.err || exit 1
echo "bench lines done"
fi
if [[ $PART == *2* ]]; then
S="--no-cpu-baseline --no-through-api --no-config3 --repeats 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_default --output-format csv -- python3 bench.py $S > $O/stats_default.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_streams1 --output-format csv -- python3 bench.py $S --streams 1 > $O/stats_streams1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_batch64 --output-format csv -- python3 bench.py $S --batch 64 --steps 10 --warmup 2 --pool 2 --streams 1 > $O/stats_batch64.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_cfg5 --output-format csv -- python3 tools/bench_cfg5.py > $O/stats_cfg5.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_model_detect --output-format csv -- python3 tools/detect_breakdown.py > $O/stats_model_detect.log 2>&1 || exit 1
echo "kernel stats done"
T="--no-cpu-baseline --no-through-api --no-config3 --repeats 1 --steps 8 --warmup 2 --no-graph --streams 1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 bench.py $T > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 bench.py $T > $O/pmc_write.log 2>&1 || exit 1
echo "traffic passes done"
python3 tools/traffic_summary.py $O/pmc_fetch $O/pmc_write > $O/traffic_pmc.json
for d in stats_default stats_streams1 stats_batch64 stats_cfg5 stats_model_detect; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
done
rm -rf $O/stats_default $O/stats_streams1 $O/stats_batch64 $O/stats_cfg5 $O/stats_model_detect $O/pmc_fetch $O/pmc_write
tools/collect_sq.sh $O/sq > /dev/null 2>&1 && cp $O/sq/sq_counters.txt $O/sq/sq_counters.json $O/ && rm -rf $O/sq
tools/phase_insts.sh $O/ph > $O/phase_instruction_counts.txt 2>&1; rm -rf $O/ph
fi
ls -la $O
