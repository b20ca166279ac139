set -e
python -m pytest tests/test_kernels_gpu.py -x -q > gpurun_out/k4.log 2>&1 || (tail -30 gpurun_out/k4.log; exit 1)
tail -1 gpurun_out/k4.log
python -m pytest tests/test_parity_gpu.py -x -q > gpurun_out/p3.log 2>&1 || (tail -30 gpurun_out/p3.log; exit 1)
tail -1 gpurun_out/p3.log
timeout -k 10 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err || (tail -20 gpurun_out/bench_full.err; exit 1)
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_full.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})
r=d['roofline']; print({k:r[k] for k in r if k!='families'})
for k,v in list(r['families'].items())[:14]: print(f"  {k:28s} {v['launches']:5d} {v['total_ms']:9.2f} ms  {v['tflops']:7.1f} TF")
PY
