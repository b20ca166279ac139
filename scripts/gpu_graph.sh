#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== c2 256x256x16 batch 8: eager vs captured step"
BGAMD_STEP_GRAPH=0 timeout -k 10 300 python bench.py --height 256 --width 256 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-host-floor 2> gpurun_out/c2_eager.err | tee gpurun_out/c2_eager.json | python -c "import json,sys;d=json.loads(sys.stdin.read());print('eager',d['value'],d['ms_per_step'])"
grep "host enqueue" gpurun_out/c2_eager.err
BGAMD_STEP_GRAPH=1 timeout -k 10 300 python bench.py --height 256 --width 256 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-host-floor 2> gpurun_out/c2_graph.err | tee gpurun_out/c2_graph.json | python -c "import json,sys;d=json.loads(sys.stdin.read());print('graph',d['value'],d['ms_per_step'])"
grep "host enqueue" gpurun_out/c2_graph.err
echo "== 64x64x16 batch 8"
BGAMD_STEP_GRAPH=0 timeout -k 10 300 python bench.py --height 64 --width 64 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-host-floor 2> gpurun_out/c1_eager.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print('eager',d['value'],d['ms_per_step'])"
grep "host enqueue" gpurun_out/c1_eager.err
BGAMD_STEP_GRAPH=1 timeout -k 10 300 python bench.py --height 64 --width 64 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-host-floor 2> gpurun_out/c1_graph.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print('graph',d['value'],d['ms_per_step'])"
grep "host enqueue" gpurun_out/c1_graph.err
