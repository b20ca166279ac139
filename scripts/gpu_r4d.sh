#!/bin/bash
# round 4, call D: the image-chunked schedule -- does the Infinity Cache serve a fresh chunk (micro-benchmark), its tests, then
# the headline step with it off / forward only / forward + backward
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out/r4d
timeout -k 10 300 python scripts/bench_mall.py > gpurun_out/r4d/mall.txt 2>&1; grep -v amdgpu gpurun_out/r4d/mall.txt
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -q -k "chunked or fused_fork" > gpurun_out/r4d/t1.log 2>&1; tail -30 gpurun_out/r4d/t1.log | cut -c1-600
bash scripts/gpu_q.sh "BGAMD_CHUNK_MB=0" "BGAMD_CHUNK_MB=96 BGAMD_CHUNK_BWD=0" "BGAMD_CHUNK_MB=96" "BGAMD_CHUNK_MB=0" "BGAMD_CHUNK_MB=96 BGAMD_CHUNK_BWD=0" "BGAMD_CHUNK_MB=96" || exit 1
