#!/bin/bash
# element-wise / reduce micro-bench at the step's activation shapes + the kernel parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 ./build_tmp/ew_patterns > gpurun_out/ew_patterns.log 2>&1; grep "lib\|hipMemcpy" gpurun_out/ew_patterns.log
timeout -k 10 300 python scripts/bench_ew.py 2>&1 | grep -v amdgpu > gpurun_out/ew.log
cat gpurun_out/ew.log | grep "72x48\|144x96" 
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -3
