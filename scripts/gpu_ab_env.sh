#!/bin/bash
# same-call A/B of an environment switch on the headline workload: gpu_ab_env.sh VAR v1 v2 ...   (interleaved twice)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/ab; mkdir -p $O
VAR=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "unset" ]; then unset $VAR; else export $VAR=$v; fi
    timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-host-floor --no-kernel-profile ${BENCH_ARGS} > $O/ab_${VAR}_${v}_$rep.json 2> $O/ab_${VAR}_${v}_$rep.err || { tail -5 $O/ab_${VAR}_${v}_$rep.err; exit 1; }
    python -c "import json;d=json.load(open('$O/ab_${VAR}_${v}_$rep.json'));print('$VAR=$v rep $rep: %.2f ms/step (median %.2f)  %.2f samples/s' % (d['ms_per_step'], d['ms_per_step_median'], d['value']))"
  done
done
