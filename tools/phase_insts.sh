#!/bin/bash
# Instructions per wave, cumulative by phase: the kernels return early (WB_CASC_DBG / WB_CHAN_DBG) under one rocprofv3 --pmc pass.
# usage (GPU box, repo root): tools/phase_insts.sh OUTDIR
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$1; mkdir -p $O
ARGS="--no-cpu-baseline --no-through-api --no-config3 --steps 3 --warmup 1 --repeats 1 --no-graph --streams 1 --batch 8 --pool 1 --stages 127"
P="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_CVT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES"
for v in "WB_CASC_DBG=2" "WB_CASC_DBG=4" "WB_CASC_DBG=8" "WB_CASC_DBG=16" "WB_CASC_DBG=32" "WB_CHAN_DBG=1" "WB_CHAN_DBG=2" "WB_CHAN_DBG=4" "X=0"; do
  export $v
  rm -rf $O/p
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $P -d $O/p --output-format csv -- python3 bench.py $ARGS > $O/log.txt 2>&1 || { tail -3 $O/log.txt; exit 1; }
  echo "== $v"
  python3 tools/pmc_summary.py --batch 8 $O/p | grep -E "^(channels|cascade)_|per wave"
  unset ${v%%=*}
done
rm -rf $O/p
