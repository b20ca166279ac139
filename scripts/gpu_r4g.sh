#!/bin/bash
# round 4, call G: the whole -m gpu suite and smoke() on the round's tree, then the evidence run
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out/r4g
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4g/t_all.log 2>&1; rc=$?
tail -5 gpurun_out/r4g/t_all.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_evidence.sh
