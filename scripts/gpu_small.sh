#!/bin/bash
# A/B of the few-tile GEMM path (BGAMD_SMALL=0 switches it off): the 256 x 256 configuration's shapes as graph-replayed
# launches with hot and with cold weights (COLD=n copies, beyond the Infinity Cache), then the configuration's step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/small
for cfg in "BGAMD_SMALL=0" "BGAMD_SMALL=1"; do
  for cold in 1 400; do
    for shape in "728 728 16 16" "1024 1536 16 16"; do
      c=$cold; [ "$cold" = 400 ] && [ "$shape" != "728 728 16 16" ] && c=150
      env $cfg COLD=$c timeout -k 10 200 python scripts/bench_conv_one.py $shape 8 2>/dev/null | sed "s/^/[$cfg COLD=$c] /" || exit 1
    done
  done
done > gpurun_out/small/micro.txt
cat gpurun_out/small/micro.txt
bash scripts/gpu_c2.sh BGAMD_SMALL=0 BGAMD_SMALL=1 BGAMD_SMALL=0 BGAMD_SMALL=1
