cd $GRAFT_REPO_ROOT
WB_JIT_VERBOSE=1 timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_jit.py tests/test_gpu_ranks.py tests/test_gpu_fuzz.py -q -s 2>&1 | grep -v amdgpu.ids > gpurun_out/t_jit.txt; tail -8 gpurun_out/t_jit.txt; grep "compiler 0" gpurun_out/t_jit.txt | cut -c1-160 | sort | uniq -c
