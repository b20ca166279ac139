#!/bin/bash
# one-box sweep of scheduling / tiling switches on the headline step (each line: samples/s, ms/step); the first and last lines are the default
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  env $cfg timeout -k 10 400 python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/knobs.json 2> gpurun_out/knobs.err || { echo "[$cfg] failed"; tail -3 gpurun_out/knobs.err; continue; }
  python -c "import json;d=json.load(open('gpurun_out/knobs.json'));print('[$cfg]',round(d['value'],3),round(d['ms_per_step'],3))"
done
