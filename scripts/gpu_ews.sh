#!/bin/bash
# rows per thread of the normalisation kernels that derive their coefficients in-kernel (BGAMD_EWS_ROWS, BGAMD_RED_ROWS): c2 and c3 steps
cd $GRAFT_REPO_ROOT
for cfg in "BGAMD_EWS_ROWS=16" "BGAMD_EWS_ROWS=4" "BGAMD_EWS_ROWS=8" "BGAMD_EWS_ROWS=4 BGAMD_RED_ROWS=4" "BGAMD_EWS_ROWS=16"; do
  env $cfg timeout -k 10 400 python bench.py --height 256 --width 256 --steps 30 --warmup 6 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/ews.json 2> gpurun_out/ews.err || { tail -5 gpurun_out/ews.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/ews.json'));print('[c2 $cfg]',d['value'],d['ms_per_step'])"
  env $cfg timeout -k 10 400 python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/ews.json 2> gpurun_out/ews.err || { tail -5 gpurun_out/ews.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/ews.json'));print('[c3 $cfg]',d['value'],d['ms_per_step'])"
done
