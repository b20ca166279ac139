#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
echo "== shipped"; SHAPES=small timeout -k 10 200 python scripts/bench_fat.py 8 2>&1 | grep -v amdgpu | grep "728-> 728\|1536\|2048"
echo "== no A traffic (ablation)"; BGAMD_LIB=$PWD/abl_build/libbgamd_FAT_NO_A.so SHAPES=small timeout -k 10 200 python scripts/bench_fat.py 8 2>&1 | grep -v amdgpu | grep "728-> 728\|1536\|2048"
