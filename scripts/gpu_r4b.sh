#!/bin/bash
# round 4, call B: the fork kernel -- its kernel test, the tests of this round's host changes, the micro-benchmark against the
# launches it replaces, the headline step with the kernel off / on (same call), then the whole -m gpu suite
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out/r4b
set -o pipefail
true
true
timeout -k 10 300 python scripts/dbg_capture_fail.py > gpurun_out/r4b/dbg_capture.txt 2>&1; tail -25 gpurun_out/r4b/dbg_capture.txt
true
timeout -k 10 300 python scripts/bench_dwfork.py 8 16 > gpurun_out/r4b/bench_dwfork.txt 2>&1 || { tail -20 gpurun_out/r4b/bench_dwfork.txt; exit 1; }
cat gpurun_out/r4b/bench_dwfork.txt
bash scripts/gpu_q.sh "BGAMD_FORK_FUSED=0 BGAMD_NO_HANDOVER=1" "BGAMD_FORK_FUSED=1 BGAMD_NO_HANDOVER=1" "BGAMD_FORK_FUSED=1" "BGAMD_FORK_FUSED=0 BGAMD_NO_HANDOVER=1" "BGAMD_FORK_FUSED=1 BGAMD_NO_HANDOVER=1" "BGAMD_FORK_FUSED=1" || exit 1
BENCH_ARGS="" bash scripts/gpu_fam.sh || exit 1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4b/t_all.log 2>&1; rc=$?
tail -5 gpurun_out/r4b/t_all.log
exit $rc
