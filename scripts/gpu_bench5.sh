set -e
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err || (tail -20 gpurun_out/bench_q.err; exit 1)
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_q.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})
r=d['roofline']; print({k:r[k] for k in r if k!='families'})
for k,v in list(r['families'].items())[:24]: print(f"  {k:30s} {v['launches']:5d} {v['total_ms']:9.2f} ms  {v['tflops']:7.1f} TF {v['alg_gbps']:8.0f} GB/s")
PY
