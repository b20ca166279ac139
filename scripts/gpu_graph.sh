#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { python bench.py --height 256 --width 256 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-host-floor 2> gpurun_out/c2_tmp.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',round(d['value'],1),round(d['ms_per_step'],2))"; grep "host enqueue" gpurun_out/c2_tmp.err; }
run "c2 graph, all streams      "
BGAMD_NO_WGRAD_STREAM=1 run "c2 graph, no wgrad stream   "
BGAMD_NO_WGRAD_STREAM=1 BGAMD_NO_SIDE_STREAM=1 BGAMD_NO_G_PREFETCH=1 run "c2 graph, single stream     "
BGAMD_STEP_GRAPH=0 BGAMD_NO_WGRAD_STREAM=1 BGAMD_NO_SIDE_STREAM=1 BGAMD_NO_G_PREFETCH=1 run "c2 eager, single stream     "
