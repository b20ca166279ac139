#!/bin/bash
# rocprofv3 on the element-wise pattern micro-benchmark: kernel durations and a few counters per kernel
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/ewprof
mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o ew -- $R/build_tmp/ew_patterns > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TCC_EA_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA_RDREQ_DRAM_sum TCC_EA_WRREQ_DRAM_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$n -o ew -- $R/build_tmp/ew_patterns > $O/pmc_$n.log 2>&1 || { tail -5 $O/pmc_$n.log; }
done
cd $R
python - <<'PY'
import csv, glob, collections, os
O='gpurun_out/ewprof'
keep=('copyBuffer','v3<4>','v2<4>','norm_act_fwd','dw_ring','dw_s1')
f=glob.glob(O+'/kt/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(k in r['Name'] for k in keep): print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
for d in sorted(glob.glob(O+'/pmc_*')):
    if not os.path.isdir(d): continue
    fs=glob.glob(d+'/**/*counter_collection.csv', recursive=True)
    if not fs: print(d,'no csv'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if not any(q in k for q in keep): continue
        # first half of the dispatches = 41 MB case, second half = 163 MB
        print(k, {c:(round(sum(x[:len(x)//2])/max(1,len(x)//2)), round(sum(x[len(x)//2:])/max(1,len(x)-len(x)//2))) for c,x in v.items()})
PY
rm -rf $O/kt $O/pmc_*
