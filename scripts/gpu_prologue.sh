#!/bin/bash
# after the prologue reorder (data requests before coefficient tables): kernel + parity tests, then the c2 and c3 steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prologue
timeout -k 10 1000 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/prologue/pytest.txt 2>&1 || { tail -30 gpurun_out/prologue/pytest.txt; exit 1; }
tail -2 gpurun_out/prologue/pytest.txt
bash scripts/gpu_c2.sh A=1 A=2 || exit 1
for i in 1 2; do
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-host-floor > gpurun_out/prologue/c3_$i.json 2> gpurun_out/prologue/c3_$i.err || { tail -5 gpurun_out/prologue/c3_$i.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/prologue/c3_$i.json'));print('[c3]',d['value'],d['ms_per_step'],d['ms_per_step_median'])"
done
