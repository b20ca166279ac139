#!/bin/bash
# 3-D depthwise kernels with all taps in flight (BGAMD_DW3_TAPS=0: the pointer kernels): parity tests, then the 3-D GAN step both ways
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dw3
timeout -k 10 900 python -m pytest tests/test_volume_gpu.py tests/test_infill_gpu.py -x -q -m gpu > gpurun_out/dw3/pytest.txt 2>&1 || { tail -30 gpurun_out/dw3/pytest.txt; exit 1; }
tail -2 gpurun_out/dw3/pytest.txt
for m in 0 1; do
  BGAMD_DW3_TAPS=$m timeout -k 10 300 python scripts/bench_gan3d.py > gpurun_out/dw3/gan3d_taps$m.txt 2>&1 || { tail gpurun_out/dw3/gan3d_taps$m.txt; exit 1; }
  grep -A4 "ms/step" gpurun_out/dw3/gan3d_taps$m.txt | grep "ms/step\|dwconv3" | sed "s/^/[taps=$m] /"
done
