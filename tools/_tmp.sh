cd $GRAFT_REPO_ROOT
WB_JIT_VERBOSE=1 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -x -q -s 2>&1 | grep -v amdgpu.ids > gpurun_out/fuzz_verbose.txt
tail -3 gpurun_out/fuzz_verbose.txt
echo "self-test lines:"; grep -c "self-test:" gpurun_out/fuzz_verbose.txt; grep "self-test:" gpurun_out/fuzz_verbose.txt | grep -v "self-test: 0 of" | head; grep "compiler [01]:" gpurun_out/fuzz_verbose.txt | cut -c1-200 | sort | uniq -c | head
