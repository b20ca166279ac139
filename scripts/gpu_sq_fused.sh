#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/sqf; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
ONE=1 timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc -o b -- python3 $R/scripts/bench_dwfused.py 8 > $O/out.log 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
cd $R
S=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python scripts/pmc_sq_summary.py $S $O/sq.json; python - <<'PY'
import json; d=json.load(open('gpurun_out/sqf/sq.json'))
for k,v in d.items():
    if isinstance(v, dict): print(k, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
PY
python - <<'PY'
import csv, collections, glob
f = glob.glob('gpurun_out/sqf/pmc/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(f)):
    n = row['Kernel_Name']
    k = 'fused<1>' if 'dw_bwd_fused_kernelILb1' in n else 'fused<0>' if 'dw_bwd_fused_kernelILb0' in n else None
    if k: agg[k][row['Counter_Name']] += float(row['Counter_Value'])
for k, c in agg.items(): print(k, dict(c))
PY
rm -rf $O/pmc
