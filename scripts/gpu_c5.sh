#!/bin/bash
# c5 (2304x1536x32): bf16 against fp8 operands in one call, plus the headline as a regression check.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/c5; mkdir -p $O; cd $R
B=${B:-4}
timeout -k 10 500 python bench.py --height 2304 --width 1536 --channels 32 --batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-host-floor > $O/bf16.json 2> $O/bf16.err || { tail -20 $O/bf16.err; exit 1; }
timeout -k 10 500 python bench.py --height 2304 --width 1536 --channels 32 --batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-host-floor --dtype fp8 > $O/fp8.json 2> $O/fp8.err || { tail -20 $O/fp8.err; exit 1; }
python - <<'PY'
import json
for t in ("bf16","fp8"):
    d=json.loads(open(f'gpurun_out/c5/{t}.json').read().strip().splitlines()[-1])
    print(t,{k:d[k] for k in ('value','ms_per_step','ms_per_step_median')}, d['config']['last_d_loss'], d['config']['last_g_loss'])
    r=d['roofline']; print({k:r[k] for k in r if k!='families'})
    for k,v in list(r['families'].items())[:18]: print(f"  {k:34s} {v['launches']:5d} {v['total_ms']:9.2f} ms  {v['tflops']:7.1f} TF {v['alg_gbps']:7.0f} GB/s")
PY
