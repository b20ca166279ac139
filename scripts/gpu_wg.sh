#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad_grouped" > gpurun_out/wg_tests.log 2>&1
rc=$?
tail -n 3 gpurun_out/wg_tests.log
[ $rc -ne 0 ] && exit $rc
echo "== XCD-local gangs"; timeout -k 10 300 python scripts/bench_wgrad.py 2>&1 | grep -v amdgpu | tee gpurun_out/wg_bench.log
echo "== spread"; BGAMD_WGG_SPREAD=1 timeout -k 10 300 python scripts/bench_wgrad.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/wg_bench.log
