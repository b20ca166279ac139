#!/bin/bash
# grouped weight gradient: spread / XCD-local gangs x 4 / 5 ring stages
cd ${GRAFT_REPO_ROOT:-$PWD}
export CASES_ONLY=${CASES_ONLY:-4}
for mode in spread local; do
  [ $mode = local ] && export BGAMD_WGG_XCD_LOCAL=1
  for nb in 4 5; do
    echo "== $mode, $nb stages"; BGAMD_WGG_NBUF=$nb timeout -k 10 120 python scripts/bench_wgrad.py 2>&1 | grep -v amdgpu | sed 's/per-layer.*| grouped/grouped/'
  done
done
