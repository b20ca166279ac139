#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-host-floor 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',round(d['ms_per_step'],2),round(d['ms_per_step_median'],2))"; }
BGAMD_NO_WGRAD_STREAM=1 BGAMD_WGRAD_GROUP=1 run "one-stream wgrad, grouped  "
BGAMD_NO_WGRAD_STREAM=1 BGAMD_WGRAD_GROUP=0 run "one-stream wgrad, per-layer"
BGAMD_WGRAD_GROUP=1 run "side-stream wgrad, grouped  "
BGAMD_WGRAD_GROUP=0 run "side-stream wgrad, per-layer"
BGAMD_NO_WGRAD_STREAM=1 BGAMD_NO_SIDE_STREAM=1 BGAMD_NO_G_PREFETCH=1 BGAMD_WGRAD_GROUP=1 run "everything on one stream, grouped  "
BGAMD_NO_WGRAD_STREAM=1 BGAMD_NO_SIDE_STREAM=1 BGAMD_NO_G_PREFETCH=1 BGAMD_WGRAD_GROUP=0 run "everything on one stream, per-layer"
