#!/bin/bash
# The round's evidence run on the GPU box: bench.py (the driver's command, with the CPU baseline), rocprofv3
# --kernel-trace --stats of the same command (default streams, and every stream switch off), separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE, one SQ set).  Outputs under gpurun_out/evidence/; the summaries are copied into profiles/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/evidence
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/evidence/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median','host_ms_per_step')}, d.get('cpu_baseline'))
r=d['roofline']; print({k:r[k] for k in r if k!='families'})
for k,v in list(r['families'].items())[:16]: print(f"  {k:34s} {v['launches']:5d} {v['total_ms']:9.2f} ms  {v['tflops']:7.1f} TF {v['alg_gbps']:7.0f} GB/s")
PY
timeout -k 10 400 python bench.py --loss wgan-gp --steps 10 --warmup 3 --no-cpu-baseline --no-host-floor > $O/bench_wgan_gp.json 2> $O/bench_wgan_gp.err || { tail -20 $O/bench_wgan_gp.err; exit 1; }
python -c "import json;d=json.load(open('$O/bench_wgan_gp.json'));print('wgan-gp',d['value'],d['ms_per_step'],d['roofline']['step_conv_stack_frac'])"
timeout -k 10 400 python bench.py --height 256 --width 256 --steps 30 --warmup 6 --no-cpu-baseline --no-kernel-profile > $O/bench_c2.json 2> $O/bench_c2.err || { tail -20 $O/bench_c2.err; exit 1; }
python -c "import json;d=json.load(open('$O/bench_c2.json'));print('c2',d['value'],d['ms_per_step'],d.get('host_ms_per_step'))"
# c5 (2304x1536x32, batch 4): bf16 and fp8 operands in the same call, and the rocprofv3 kernel statistics of the fp8 command
timeout -k 10 500 python bench.py --height 2304 --width 1536 --channels 32 --batch 4 --steps 6 --warmup 2 --no-cpu-baseline --no-host-floor > $O/bench_c5_bf16.json 2> $O/bench_c5_bf16.err || { tail -20 $O/bench_c5_bf16.err; exit 1; }
timeout -k 10 500 python bench.py --height 2304 --width 1536 --channels 32 --batch 4 --steps 6 --warmup 2 --no-cpu-baseline --no-host-floor --dtype fp8 > $O/bench_c5_fp8.json 2> $O/bench_c5_fp8.err || { tail -20 $O/bench_c5_fp8.err; exit 1; }
python -c "import json;a=json.load(open('$O/bench_c5_bf16.json'));b=json.load(open('$O/bench_c5_fp8.json'));print('c5 bf16',a['value'],a['ms_per_step'],'fp8',b['value'],b['ms_per_step'],b['roofline']['kernel'],b['roofline']['frac'])"
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o bench -- python3 $R/bench.py --height 2304 --width 1536 --channels 32 --batch 4 --dtype fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-host-floor > $O/prof_c5.json 2> $O/prof_c5.err || { tail -5 $O/prof_c5.err; exit 1; }
echo c5 fp8 kernel-trace done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-host-floor > $O/prof_bench.json 2> $O/prof_bench.err || { tail -5 $O/prof_bench.err; exit 1; }
echo kernel-trace done
BGAMD_NO_SIDE_STREAM=1 BGAMD_NO_G_PREFETCH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -o bench -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-host-floor > $O/prof_single.json 2> $O/prof_single.err || { tail -5 $O/prof_single.err; exit 1; }
echo single-stream kernel-trace done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o bench -- python3 $R/bench.py --height 256 --width 256 --steps 10 --warmup 6 --no-cpu-baseline --no-kernel-profile --no-host-floor > $O/prof_c2.json 2> $O/prof_c2.err || { tail -5 $O/prof_c2.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o bench -- python3 $R/bench.py --loss wgan-gp --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-host-floor > $O/prof_c4.json 2> $O/prof_c4.err || { tail -5 $O/prof_c4.err; exit 1; }
echo c2 / c4 kernel-trace done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-host-floor > /dev/null 2> $O/pmc_fetch.err || { tail -5 $O/pmc_fetch.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-host-floor > /dev/null 2> $O/pmc_write.err || { tail -5 $O/pmc_write.err; exit 1; }
echo pmc traffic done
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-host-floor > /dev/null 2> $O/pmc_sq.err || { tail -5 $O/pmc_sq.err; exit 1; }
echo sq done
cd $R
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
# the per-dispatch counter CSVs are large: fold them here, keep the summaries
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1); S=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1)
python scripts/pmc_summary.py $F $W $O/pmc_hbm_traffic.json | head -12
python scripts/pmc_sq_summary.py $S $O/pmc_sq.json
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/prof_single -name "*kernel_stats.csv" | head -1) $O/kernel_stats_single_stream.csv
cp $(find $O/prof_c5 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_c5_fp8.csv
cp $(find $O/prof_c2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_c2.csv
cp $(find $O/prof_c4 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_c4.csv
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/prof $O/prof_single $O/prof_c5 $O/prof_c2 $O/prof_c4
ls -la $O
