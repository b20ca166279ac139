#!/bin/bash
# weight gradients on their own stream or on the caller's (BGAMD_NO_WGRAD_STREAM=1): c3 alternating, then c4 (wgan-gp) and c5
cd $GRAFT_REPO_ROOT
one() {  # tag, env, bench args
  env $2 timeout -k 10 600 python bench.py $3 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/wgs.json 2> gpurun_out/wgs.err || { tail -5 gpurun_out/wgs.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/wgs.json'));print('[$1 $2]',round(d['value'],3),round(d['ms_per_step'],3))"
}
for i in 1 2 3; do
  one c3 A=default "--steps 15 --warmup 5" && one c3 BGAMD_NO_WGRAD_STREAM=1 "--steps 15 --warmup 5" || exit 1
done
one c4 A=default "--loss wgan-gp --steps 8 --warmup 3" && one c4 BGAMD_NO_WGRAD_STREAM=1 "--loss wgan-gp --steps 8 --warmup 3" || exit 1
one c5 A=default "--height 2304 --width 1536 --channels 32 --batch 4 --steps 6 --warmup 2" && one c5 BGAMD_NO_WGRAD_STREAM=1 "--height 2304 --width 1536 --channels 32 --batch 4 --steps 6 --warmup 2"
