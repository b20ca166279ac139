cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -x -q -k "dw" 2>&1 | tail -3
echo "== old"; BGAMD_DW_OLD=1 timeout -k 10 100 python scripts/bench_ew.py dw_bwd_weight 2>/dev/null
for rb in 2 4 8 16; do echo "== DWW_RB=$rb"; BGAMD_DWW_RB=$rb timeout -k 10 100 python scripts/bench_ew.py dw_bwd_weight 2>/dev/null; done
for rb in 2 3 6; do echo "== DW_RB=$rb"; BGAMD_DW_RB=$rb timeout -k 10 100 python scripts/bench_ew.py dw_fwd 2>/dev/null; done
