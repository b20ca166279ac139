cd ${GRAFT_REPO_ROOT:-$PWD}
BGAMD_NO_FUSED_STATS=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --dump-launches gpurun_out/launches_nofs.txt > /dev/null 2>&1
timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --dump-launches gpurun_out/launches.txt > /dev/null 2>&1
wc -l gpurun_out/launches*.txt
