set -e
R=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 500 python bench.py --steps 8 --warmup 3 > gpurun_out/bench_full.json 2> >(tee gpurun_out/bench_full.err >&2) || (tail -20 gpurun_out/bench_full.err; exit 1)
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_full.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('cpu_baseline'))
r=d['roofline']; print({k:r[k] for k in r if k!='families'})
for k,v in list(r['families'].items())[:14]: print(f"  {k:28s} {v['launches']:5d} {v['total_ms']:9.2f} ms  {v['tflops']:7.1f} TF")
PY
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err || (tail -5 $R/gpurun_out/prof_bench.err; exit 1)
echo kernel-trace done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile > /dev/null 2> $R/gpurun_out/pmc_fetch.err || (tail -5 $R/gpurun_out/pmc_fetch.err; exit 1)
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile > /dev/null 2> $R/gpurun_out/pmc_write.err || (tail -5 $R/gpurun_out/pmc_write.err; exit 1)
echo pmc done
