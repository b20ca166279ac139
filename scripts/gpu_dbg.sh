cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "data_parallel" 2>&1 | tail -30
