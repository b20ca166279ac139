#!/bin/bash
# round 4, call A: VALU issue-rate micro-benchmark + the headline family table of the unchanged tree (reference for A/B)
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out/r4a
timeout -k 10 120 scripts/micro/valu_rate > gpurun_out/r4a/valu_rate.txt 2>&1 || { tail -5 gpurun_out/r4a/valu_rate.txt; exit 1; }
cat gpurun_out/r4a/valu_rate.txt
bash scripts/gpu_fam.sh
