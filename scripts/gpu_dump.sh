cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --dump-launches gpurun_out/launches.txt > /dev/null 2>&1
wc -l gpurun_out/launches.txt
