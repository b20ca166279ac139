#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "bwd_fused" > gpurun_out/t_fused.log 2>&1; tail -5 gpurun_out/t_fused.log
for cfg in "512 256" "384 256" "512 512"; do set -- $cfg; echo "== PIX $1 BLOCKS $2"; BGAMD_FB_PIX=$1 BGAMD_FB_BLOCKS=$2 timeout -k 10 200 python scripts/bench_dwfused.py 8 16 2>&1 | grep -v amdgpu.ids; done > gpurun_out/bench_dwfused.log 2>&1
cat gpurun_out/bench_dwfused.log
timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -x -q -s -k "teacher_forced" > gpurun_out/t_tf.log 2>&1
grep -v "^Constructing\|^Number\|^Output" gpurun_out/t_tf.log | tail -60
