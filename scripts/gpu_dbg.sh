cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "avgpool or conv_transpose" 2>&1 | tail -25
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -s -k "test_generator_vs_reference_golden and deconv" 2>&1 | tail -40
