#!/bin/bash
cd $GRAFT_REPO_ROOT
export BGAMD_LIB=$GRAFT_REPO_ROOT/abl_build/libbgamd_stamps.so
for args in "728 728 72 48 8" "728 728 72 48 16" "1536 1536 72 48 8" "728 728 144 96 8"; do
  echo "=== $args (BKB 128)"; timeout -k 10 120 python scripts/stamps_fat.py $args 2>&1 | grep -v amdgpu
done
echo "=== 728 728 72 48 8 (BKB 64)"; BGAMD_FAT_BKB=64 timeout -k 10 120 python scripts/stamps_fat.py 728 728 72 48 8 2>&1 | grep -v amdgpu
