#!/bin/bash
# fork kernel under fp8 operands: fp8 tests, the full-size c5 parity test, then c5 bf16 / fp8 in one call
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/fp8fork_pytest.txt 2>&1 || { tail -30 gpurun_out/fp8fork_pytest.txt; exit 1; }
tail -2 gpurun_out/fp8fork_pytest.txt
for i in 1 2; do
for dt in bf16 fp8; do
  timeout -k 10 500 python bench.py --height 2304 --width 1536 --channels 32 --batch 4 --steps 6 --warmup 2 --no-cpu-baseline --no-host-floor --no-kernel-profile --dtype $dt > gpurun_out/fp8fork_$dt.json 2> gpurun_out/fp8fork.err || { tail -5 gpurun_out/fp8fork.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/fp8fork_$dt.json'));print('[c5 $dt]',round(d['value'],3),round(d['ms_per_step'],2))"
done
done
