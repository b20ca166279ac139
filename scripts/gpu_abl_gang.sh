#!/bin/bash
# grouped weight gradient, ablation builds (scripts/ablate_gang.sh), spread gangs and XCD-local gangs
cd ${GRAFT_REPO_ROOT:-$PWD}
export CASES_ONLY=${CASES_ONLY:-1}
for mode in spread local; do
  [ $mode = local ] && export BGAMD_WGG_XCD_LOCAL=1
  echo "== $mode shipped"; timeout -k 10 120 python scripts/bench_wgrad.py 2>&1 | grep -v amdgpu
  echo "== $mode 5 stages"; BGAMD_WGG_NBUF=5 timeout -k 10 120 python scripts/bench_wgrad.py 2>&1 | grep -v amdgpu
  for v in G_NO_DMA G_HOT G_NO_FRAG G_NO_MFMA; do
    echo "== $mode $v"; BGAMD_LIB=$PWD/abl_build/libbgamd_$v.so timeout -k 10 120 python scripts/bench_wgrad.py 2>&1 | grep -v amdgpu
  done
done
