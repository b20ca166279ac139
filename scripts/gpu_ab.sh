set -e
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -4
for v in 1 ""; do
  echo "== BGAMD_NO_G_PREFETCH=$v"
  BGAMD_NO_G_PREFETCH=$v timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['last_d_loss'], d['config']['last_g_loss'])"
done
