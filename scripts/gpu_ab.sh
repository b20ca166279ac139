set -e
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --dump-launches gpurun_out/launches.txt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
python scripts/launch_table.py gpurun_out/launches.txt 6
