#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py -q -s -k "teacher_forced" > gpurun_out/t_tf.log 2>&1
grep -v "^Constructing\|^Number\|^Output" gpurun_out/t_tf.log | grep -v "^block[0-9]* .*rms-rel.*max-rel" | tail -40
timeout -k 10 300 python -m pytest tests/test_fp8_gpu.py -x -q > gpurun_out/t_fp8.log 2>&1; tail -3 gpurun_out/t_fp8.log
B=4 bash scripts/gpu_c5.sh > gpurun_out/c5.log 2>&1; grep -v "^  bg_\|^{'bound" gpurun_out/c5.log | tail; grep "bg_quant\|fp8" gpurun_out/c5.log | head -12
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k c5 > gpurun_out/c5_test.log 2>&1; grep "c5 2304\|passed\|failed\|Error" gpurun_out/c5_test.log | tail
