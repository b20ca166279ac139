#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "dwconv" 2>&1 | tail -8 || exit 1
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "block or generator or discriminator or iteration" 2>&1 | tail -8 || exit 1
