#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -q -x -s -k "validation or two_ranks or oracle_average or step_schedules" > gpurun_out/t.log 2>&1
rc=$?
grep -E "^validation|passed|failed|tell the" gpurun_out/t.log | tail; [ $rc -ne 0 ] && tail -30 gpurun_out/t.log
exit $rc
