#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "A=1" "BGAMD_NO_FORK_DW=1" "BGAMD_NO_FOLD_FINALIZE=1" "BGAMD_NO_FUSED_DW=1" "BGAMD_DW_RING=0" "BGAMD_RED_BLOCKS=1024 BGAMD_EWB_BLOCKS=1024"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "checkpoint_matches" 2>&1 | grep "AssertionError: (\|passed\|failed" | cut -c1-260
done
