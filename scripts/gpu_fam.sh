#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out/fam
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-floor ${BENCH_ARGS} --dump-launches gpurun_out/fam/launches.txt > gpurun_out/fam/bench.json 2> gpurun_out/fam/bench.err || { tail -5 gpurun_out/fam/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/fam/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median')})
r=d['roofline']; print({k:r[k] for k in r if k not in ('families','traffic_source')})
tot=0
for k,v in r['families'].items():
    tot+=v['total_ms']; print(f"  {k:34s} {v['launches']:5d} {v['total_ms']:9.2f} ms  {v['tflops']:7.1f} TF {v['alg_gbps']:7.0f} GB/s")
print('sum', tot)
PY
