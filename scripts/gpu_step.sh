#!/bin/bash
# GPU test suite (the full-size file separately: its prints are the measured deviations), then the headline bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$1" != "quick" ]; then
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s > gpurun_out/tests_fullsize.log 2>&1
grep -E "^c3|^c4|^c5|running stat|passed|failed" gpurun_out/tests_fullsize.log
fi
timeout -k 10 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_fullsize_gpu.py > gpurun_out/tests_all.log 2>&1
rc=$?
tail -n 8 gpurun_out/tests_all.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-launches gpurun_out/launches_fat.txt > gpurun_out/bench_fat.json 2> gpurun_out/bench_fat.err
tail -n 4 gpurun_out/bench_fat.err; python -c "import json;d=json.load(open('gpurun_out/bench_fat.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median'],d['roofline']['frac'],d.get('host_ms_per_step'))"
BGAMD_WGRAD_GROUP=0 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/bench_nogroup.json 2> gpurun_out/bench_nogroup.err
python -c "import json;d=json.load(open('gpurun_out/bench_nogroup.json'));print(d['value'],d['ms_per_step'])"
