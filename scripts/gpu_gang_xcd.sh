#!/bin/bash
# grouped weight gradient: XCD-local placement (default) against spread gangs -- time and L2-miss bytes (FETCH_SIZE: KiB units x 2 on gfx950)
cd ${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
O=$PWD/gpurun_out/gang_xcd; mkdir -p $O
for mode in local spread; do
  [ $mode = spread ] && export BGAMD_WGG_SPREAD=1
  echo "== $mode"; CASES_ONLY=${CASES_ONLY:-11} timeout -k 10 120 python scripts/bench_wgrad.py 2>&1 | grep -v amdgpu | sed 's/per-layer.*| grouped/grouped/'
  (cd /tmp && CASES_ONLY=1 timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$mode -o w -- python3 $OLDPWD/scripts/bench_wgrad.py > /dev/null 2> $O/$mode.err) || { tail -5 $O/$mode.err; exit 1; }
  F=$(find $O/$mode -name "*counter_collection.csv" | head -1)
  python - $F <<'P'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "wgrad_gang" in k: print(f"{k:60s} x{len(v):3d}  {2 * 1024 * sum(v) / len(v) / 1e6:9.1f} MB fetched per launch (operands: 3864.6 MB)")
P
done
rm -rf $O/spread $O/local
