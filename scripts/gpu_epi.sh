#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv" 2>&1 | tail -3 || exit 1
SHAPES=small timeout -k 10 300 python scripts/bench_fat.py 8 2>&1 | grep "72x 48"
