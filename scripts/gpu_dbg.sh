cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "pixel_losses or l1" 2>&1 | tail -8
