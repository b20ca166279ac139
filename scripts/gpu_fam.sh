#!/bin/bash
# headline bench with the per-family kernel profile printed
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-floor > gpurun_out/fam.json 2> gpurun_out/fam.err || { tail -20 gpurun_out/fam.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/fam.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median')})
r=d['roofline']; print({k:r[k] for k in r if k!='families'})
tot=sum(v['total_ms'] for v in r['families'].values())
for k,v in r['families'].items(): print(f"  {k:40s} {v['launches']:5d} {v['total_ms']:9.2f} ms {100*v['total_ms']/tot:5.1f}%  {v['tflops']:7.1f} TF {v['alg_gbps']:7.0f} GB/s")
print('sum', tot)
PY
