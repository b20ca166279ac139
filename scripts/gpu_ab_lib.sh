#!/bin/bash
# same-box A/B of two builds of the library: bias-gan_amd/libbgamd_old.so (built from the previous commit) against the tree's
cd $GRAFT_REPO_ROOT
P=bias-gan_amd
cp $P/libbgamd.so $P/libbgamd_new.so
run() {
  cp $P/libbgamd_$1.so $P/libbgamd.so
  timeout -k 10 400 python bench.py --height 256 --width 256 --steps 30 --warmup 6 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/ab_c2_$1_$2.json 2> gpurun_out/ab_$1.err || { tail -5 gpurun_out/ab_$1.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/ab_c2_$1_$2.json'));print('[c2 $1]',d['value'],d['ms_per_step'],d['ms_per_step_median'])"
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/ab_c3_$1_$2.json 2> gpurun_out/ab_$1.err || { tail -5 gpurun_out/ab_$1.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/ab_c3_$1_$2.json'));print('[c3 $1]',d['value'],d['ms_per_step'],d['ms_per_step_median'])"
}
run old 1 && run new 1 && run old 2 && run new 2
cp $P/libbgamd_new.so $P/libbgamd.so
