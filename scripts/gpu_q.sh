#!/bin/bash
# quick A/B of the headline step under a few environment settings: scripts/gpu_q.sh "ENV1=.. ENV2=.." "..."
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/q_$i.json 2> gpurun_out/q_$i.err || { tail -5 gpurun_out/q_$i.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/q_$i.json'));print('[$cfg]',d['value'],d['ms_per_step'],d['ms_per_step_median'])"
done
