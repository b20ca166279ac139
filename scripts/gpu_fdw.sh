#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "dwconv or norm" 2>&1 | tail -15 || exit 1
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "block" 2>&1 | tail -15 || exit 1
timeout -k 10 300 python scripts/bench_ew.py dw_bwd_data bwd_reduce 2>&1 | grep -v amdgpu | grep "72x48x  728\|144x96\|1536"
bash scripts/gpu_q.sh "A=1" "BGAMD_NO_FUSED_DW_REDUCE=1"
