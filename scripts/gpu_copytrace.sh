#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/copytrace
mkdir -p $O
export TMPDIR=/tmp; cd /tmp
BGAMD_STEP_GRAPH=0 BGAMD_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o t -- python3 $R/bench.py --height 128 --width 128 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-host-floor > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
cd $R
python - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/copytrace/kt/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
agg=collections.Counter()
def short(n):
    n=n.replace('(anonymous namespace)::','').replace('void ','')
    return n[:48]
for i,n in enumerate(names):
    if 'copyBuffer' in n:
        agg[(short(names[i-1]) if i else '-', short(names[i+1]) if i+1<len(names) else '-')]+=1
print('copyBuffer total', sum(agg.values()), 'of', len(names))
for k,v in agg.most_common(15): print(v, k)
PY
rm -rf $O/kt
