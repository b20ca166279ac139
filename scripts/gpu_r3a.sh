#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "bwd_fused" > gpurun_out/t_fused.log 2>&1; tail -5 gpurun_out/t_fused.log
for cfg in "256" "512"; do echo "== BLOCKS $cfg"; BGAMD_FB_BLOCKS=$cfg timeout -k 10 200 python scripts/bench_dwfused.py 8 16 2>&1 | grep -v amdgpu.ids; done > gpurun_out/bench_dwfused.log 2>&1
cat gpurun_out/bench_dwfused.log
