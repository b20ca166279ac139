#!/bin/bash
# round 4, call D: the image-chunked schedule -- its tests, then the headline step with it off / on at several chunk sizes
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out/r4d
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -q -k "chunked or fused_fork" > gpurun_out/r4d/t1.log 2>&1; tail -30 gpurun_out/r4d/t1.log | cut -c1-600
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "block or one_iteration or reuse_g_forward or failed_step or data_parallel or whole_step" > gpurun_out/r4d/t2.log 2>&1 || { tail -40 gpurun_out/r4d/t2.log; exit 1; }
tail -3 gpurun_out/r4d/t2.log
bash scripts/gpu_q.sh "BGAMD_CHUNK_MB=0" "BGAMD_CHUNK_MB=96 BGAMD_CHUNK_BWD=0" "BGAMD_CHUNK_MB=96" "BGAMD_CHUNK_MB=48" "BGAMD_CHUNK_MB=0" "BGAMD_CHUNK_MB=96 BGAMD_CHUNK_BWD=0" "BGAMD_CHUNK_MB=96" || exit 1
BENCH_ARGS="" bash scripts/gpu_fam.sh || exit 1
