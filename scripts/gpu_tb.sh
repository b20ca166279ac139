set -e
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
bash scripts/gpu_bench5.sh
