#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-floor --dump-launches gpurun_out/launches_$2.txt 2>/dev/null > gpurun_out/bench_$2.json; python -c "import json;d=json.load(open('gpurun_out/bench_$2.json'));print('$1',round(d['ms_per_step'],2),round(d['ms_per_step_median'],2), round(d['roofline']['frac'],3))"; }
BGAMD_ROW_ALIGN=64 run "row align  64" a64
BGAMD_ROW_ALIGN=128 run "row align 128" a128
BGAMD_ROW_ALIGN=256 run "row align 256" a256
