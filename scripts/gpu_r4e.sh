#!/bin/bash
# round 4, call E: block 1's fork behind bn2 -- tests, headline A/B (fork kernel off / on), family table
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out/r4e
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -x -q -k "fused_fork or block or one_iteration or c3_full_size or failed_step or data_parallel" > gpurun_out/r4e/t1.log 2>&1 || { tail -40 gpurun_out/r4e/t1.log | cut -c1-400; exit 1; }
tail -3 gpurun_out/r4e/t1.log
bash scripts/gpu_q.sh "BGAMD_FORK_FUSED=0" "BGAMD_FORK_FUSED=1" "BGAMD_FORK_FUSED=0" "BGAMD_FORK_FUSED=1" || exit 1
BENCH_ARGS="" bash scripts/gpu_fam.sh || exit 1
