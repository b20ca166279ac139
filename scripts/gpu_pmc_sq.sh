# One SQ counter pass over bench.py (separate from the kernel trace and the TCC passes, as gpurun requires):
# wave cycles, wait buckets, MFMA busy cycles, LDS activity / bank conflicts per dispatch.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_sq -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile > /dev/null 2> $R/gpurun_out/pmc_sq.err || (tail -5 $R/gpurun_out/pmc_sq.err; exit 1)
echo sq done; ls -la $R/gpurun_out/pmc_sq | head
