#!/bin/bash
# partial-conv GAN iteration under the round's switches (one box)
cd $GRAFT_REPO_ROOT
for cfg in "A=default" "BGAMD_SMALL=0" "BGAMD_EWS_ROWS=16" "BGAMD_WGRAD_STREAM=1" "BGAMD_WGRAD_STREAM=1 BGAMD_WGRAD_STREAM_MIN=0" "A=default"; do
  env $cfg timeout -k 10 300 python scripts/bench_infill3d.py 2>/dev/null | grep "ms/step" | sed "s/^/[$cfg] /"
done
