#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "A=default" "BGAMD_SMALL=0" "BGAMD_WGRAD_GROUP=0" "BGAMD_FORK_FUSED=0" "BGAMD_NO_FUSED_DW=1" "BGAMD_NO_SPLITK=1" "BGAMD_WGRAD_WS=1" "BGAMD_NO_FUSED_STATS=1"; do
  env $cfg timeout -k 10 200 python scripts/check_determinism.py 256 4 d 2>/dev/null | tail -1 | sed "s/^/[$cfg] /"
done
timeout -k 10 200 python scripts/check_determinism.py 128 4 d 2>/dev/null | tail -1
timeout -k 10 200 python scripts/check_determinism.py 256 4 g 2>/dev/null | tail -1
