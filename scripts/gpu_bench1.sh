set -e
timeout -k 10 300 python bench.py --height 256 --width 256 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_256.json 2> gpurun_out/bench_256.err || (tail -20 gpurun_out/bench_256.err; exit 1)
cat gpurun_out/bench_256.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}); print(json.dumps(d['roofline'])[:1500])"
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err || (tail -20 gpurun_out/bench_full.err; exit 1)
cat gpurun_out/bench_full.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}); print(json.dumps(d['roofline'])[:3000])"
