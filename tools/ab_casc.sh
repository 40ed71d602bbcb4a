set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=${1:-gpurun_out/r4l}; mkdir -p $O
run() {  # name defs
  make -C waldboost_amd/csrc clean > /dev/null 2>&1
  make -C waldboost_amd/csrc -j8 DEFS="$2" > $O/make_$1.log 2>&1 || { tail -5 $O/make_$1.log; exit 1; }
  WB_JIT_CACHE=/tmp/jit_$1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-through-api > $O/b1_$1.json 2> $O/err_$1.log
  WB_JIT_CACHE=/tmp/jit_$1 python bench.py --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-through-api > $O/b64_$1.json 2>> $O/err_$1.log
  WB_JIT_CACHE=/tmp/jit_$1 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg5_$1.json 2>> $O/err_$1.log
  echo "$1 done"
}
run qfull_barrier "-DWB_CASC_QFULL=1 -DWB_CASC_END_BARRIER=1"
run qcap_barrier "-DWB_CASC_END_BARRIER=1"
run qcap_exit ""
