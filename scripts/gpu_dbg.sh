cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -k "test_block_vs_reference_golden" 2>&1 | tail -40
