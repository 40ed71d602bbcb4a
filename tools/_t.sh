cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_jit.py tests/test_gpu_graph.py tests/test_gpu_stream.py -x -q 2>&1 | tail -15
