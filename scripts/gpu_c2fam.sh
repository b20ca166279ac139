#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --height 256 --width 256 --steps 30 --warmup 6 --no-cpu-baseline --no-host-floor > gpurun_out/c2fam.json 2> gpurun_out/c2fam.err || { tail -5 gpurun_out/c2fam.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c2fam.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median')})
r=d['roofline']
tot=sum(v['total_ms'] for v in r['families'].values()); n=sum(v['launches'] for v in r['families'].values())
print('kernel sum ms', tot, 'launches', n)
for k,v in list(r['families'].items())[:16]: print(f"  {k:36s} {v['launches']:5d} {v['total_ms']:8.2f} ms  avg {1e3*v['total_ms']/v['launches']:7.1f} us {v['tflops']:7.1f} TF {v['alg_gbps']:7.0f} GB/s")
PY
