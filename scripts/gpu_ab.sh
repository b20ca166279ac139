set -e
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or norm" 2>&1 | tail -4
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -3
for v in 1 8 16; do
  echo "== bench BGAMD_STAT_COPIES=$v"
  BGAMD_STAT_COPIES=$v timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
