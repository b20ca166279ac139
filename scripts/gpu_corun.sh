#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out/corun
timeout -k 10 200 python scripts/bench_corun.py > gpurun_out/corun/shipped.txt 2>&1; grep -v amdgpu gpurun_out/corun/shipped.txt
BGAMD_FAT_NO384=1 timeout -k 10 200 python scripts/bench_corun.py > gpurun_out/corun/no384.txt 2>&1; grep -v amdgpu gpurun_out/corun/no384.txt
