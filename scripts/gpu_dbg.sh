cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -s -k "test_generator_vs_reference_golden and deconv1x and dtype0" 2>&1 | grep -E "generator|assert|Error|passed|failed" | head -20
