#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv or split" > gpurun_out/fat_tests.log 2>&1
rc=$?
tail -n 5 gpurun_out/fat_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/bench_fat.py 8 2>&1 | grep -v amdgpu | tee gpurun_out/fat_bench.log
