#!/bin/bash
# fat-tile GEMM: parity tests, A/B micro-benchmark
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "fat or conv2d_fwd_bwd or wide_tile or fused_stat or 2gib" > gpurun_out/fat_tests.log 2>&1
rc=$?
tail -n 5 gpurun_out/fat_tests.log
[ $rc -ne 0 ] && exit $rc
echo "== default" | tee gpurun_out/fat_bench.log
timeout -k 10 300 python scripts/bench_fat.py 8 16 2>&1 | grep -v amdgpu | tee -a gpurun_out/fat_bench.log
echo "== FAT128" | tee -a gpurun_out/fat_bench.log
SHAPES=small BGAMD_FAT128=1 timeout -k 10 300 python scripts/bench_fat.py 8 16 2>&1 | grep -v amdgpu | tee -a gpurun_out/fat_bench.log
