set -e
python -m pytest tests/test_parity_gpu.py -q -s > gpurun_out/p1.log 2>&1 || true
grep -E "passed|failed|FAILED|max-rel|iteration|Error" gpurun_out/p1.log | head -60
