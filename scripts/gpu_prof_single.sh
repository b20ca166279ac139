# rocprofv3 kernel trace of bench.py with every stream switch off (weight-gradient stream, side stream, early G forward):
# kernels run back to back, so their durations are comparable with the HIP-event figures of bench.py's single-stream
# profile step.  Output: gpurun_out/prof_single/bench_kernel_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp BGAMD_NO_WGRAD_STREAM=1 BGAMD_NO_SIDE_STREAM=1 BGAMD_NO_G_PREFETCH=1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_single -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile > $R/gpurun_out/prof_single.json 2> $R/gpurun_out/prof_single.err || (tail -5 $R/gpurun_out/prof_single.err; exit 1)
tail -n 1 $R/gpurun_out/prof_single.json | cut -c1-200
