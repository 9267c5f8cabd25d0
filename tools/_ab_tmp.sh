cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "2x2_map" 2>&1 | tail -15
