#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad_grouped" > gpurun_out/wg_tests.log 2>&1
rc=$?
tail -n 3 gpurun_out/wg_tests.log
[ $rc -ne 0 ] && exit $rc
echo "== 5 stages"; timeout -k 10 300 python scripts/bench_wgrad.py 2>&1 | grep -v amdgpu | head -4 | tee gpurun_out/wg_bench.log
echo "== 4 stages"; BGAMD_WGG_NBUF=4 timeout -k 10 300 python scripts/bench_wgrad.py 2>&1 | grep -v amdgpu | head -4 | tee -a gpurun_out/wg_bench.log
