#!/bin/bash
# element-wise / reduce micro-bench at the step's activation shapes + the kernel parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "dwconv" 2>&1 | tail -5 || exit 1
timeout -k 10 300 python scripts/bench_ew.py ${EW_CASES:-copy dw_} 2>&1 | grep -v amdgpu > gpurun_out/ew.log
cat gpurun_out/ew.log
