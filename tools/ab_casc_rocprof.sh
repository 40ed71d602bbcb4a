#!/bin/bash
# the cascade kernel's round-3 queue / ending against round 4's, rebuilt on one box, by rocprofv3 kernel stats (B=64 and B=1, one stream)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=${1:-gpurun_out/ab_rocprof}; mkdir -p $O
S="--no-cpu-baseline --no-through-api --no-config3 --repeats 2"
run() {
  make -C waldboost_amd/csrc clean > /dev/null 2>&1
  make -C waldboost_amd/csrc -j8 DEFS="$2" > $O/make_$1.log 2>&1 || { tail -5 $O/make_$1.log; exit 1; }
  for rep in 1 2; do
  WB_JIT_CACHE=/tmp/jit_$1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/s64_$1_$rep --output-format csv -- python3 bench.py $S --batch 64 --steps 10 --warmup 2 --pool 2 --streams 1 > $O/log_$1.txt 2>&1 || exit 1
  WB_JIT_CACHE=/tmp/jit_$1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/s1_$1_$rep --output-format csv -- python3 bench.py $S --streams 1 > $O/log_$1.txt 2>&1 || exit 1
  for d in s64_$1_$rep s1_$1_$rep; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "$d casc $(grep wb_casc_jit $f | cut -d, -f2-4) | chan $(grep 'channels_kernel' $f | head -1 | awk -F'",' '{print $2}' | cut -d, -f1-3)"; rm -rf $O/$d; done
  done
}
run qfull_barrier "-DWB_CASC_QFULL=1 -DWB_CASC_END_BARRIER=1"
run qcap_exit ""
run qfull_barrier2 "-DWB_CASC_QFULL=1 -DWB_CASC_END_BARRIER=1"
run qcap_exit2 ""
