#!/bin/bash
# SQ counters of the hot kernels (three rocprofv3 --pmc passes of 8 counters; no trace domains besides the kernel trace).
# usage (on the GPU box, repo root): tools/collect_sq.sh OUTDIR [extra bench.py args]   -> OUTDIR/sq_counters.{txt,json}
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
B=8
ARGS="--no-cpu-baseline --no-through-api --no-config3 --steps 4 --warmup 1 --repeats 1 --no-graph --streams 1 --batch $B --pool 1 $*"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P -d $O/pmc_sq$i --output-format csv -- python3 bench.py $ARGS > $O/pmc_sq$i.log 2>&1 || { tail -5 $O/pmc_sq$i.log; exit 1; }
  echo "pass $i done"
done
python3 tools/pmc_summary.py --batch $B --json $O/sq_counters.json $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 > $O/sq_counters.txt
rm -rf $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3
cat $O/sq_counters.txt
