#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_reader_gpu.py -q -x > gpurun_out/reader_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/reader_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/bench_resident.json 2> gpurun_out/bench_resident.err
python -c "import json;d=json.load(open('gpurun_out/bench_resident.json'));print('resident',d['value'],d['ms_per_step'],d['ms_per_step_median'])"
timeout -k 10 500 python bench.py --data ring --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/bench_ring.json 2> gpurun_out/bench_ring.err
rc=$?; tail -n 3 gpurun_out/bench_ring.err
python -c "import json;d=json.load(open('gpurun_out/bench_ring.json'));print('ring',d['value'],d['ms_per_step'],d['ms_per_step_median'],d['staging'])"
