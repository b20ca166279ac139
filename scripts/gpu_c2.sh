#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -k 10 400 python bench.py --height 256 --width 256 --steps 30 --warmup 6 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/c2_$i.json 2> gpurun_out/c2_$i.err || { tail -5 gpurun_out/c2_$i.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/c2_$i.json'));print('[$cfg]',d['value'],d['ms_per_step'],d['ms_per_step_median'])"
done
