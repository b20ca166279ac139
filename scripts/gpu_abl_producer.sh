#!/bin/bash
# the synthetic-producer ablation of the fat GEMM on the dominant shapes: shipped library, then each build, with 128-byte
# rows (two stages) and with 64-byte rows (four stages: the variant a fused kernel's LDS budget allows)
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out/ablp
for bkb in 128 64; do
  echo "== shipped, ${bkb}-byte rows"; BGAMD_FAT_BKB=$bkb SHAPES=small timeout -k 10 200 python scripts/bench_fat.py 8 2>&1 | grep "728-> 728.*72x 48\|1536->1536"
  for so in abl_build/libbgamd_P*.so; do
    echo "== $(basename $so .so | sed 's/libbgamd_//'), ${bkb}-byte rows"
    BGAMD_FAT_BKB=$bkb BGAMD_LIB=$PWD/$so SHAPES=small timeout -k 10 200 python scripts/bench_fat.py 8 2>&1 | grep "728-> 728.*72x 48\|1536->1536"
  done
done 2>&1 | tee gpurun_out/ablp/producer.txt
