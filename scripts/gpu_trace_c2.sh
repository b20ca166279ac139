#!/bin/bash
# per-dispatch kernel trace of the 256x256x16 step, folded by (kernel, grid, workgroup): which launches of the small maps are slow for their size
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/trace_c2; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/raw -o bench -- python3 $R/bench.py --height 256 --width 256 --steps 6 --warmup 6 --no-cpu-baseline --no-kernel-profile --no-host-floor > $O/c2.json 2> $O/c2.err || { tail -5 $O/c2.err; exit 1; }
cd $R
f=$(find $O/raw -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/by_shape.txt <<'PY'
import csv, sys, collections, re
g = collections.defaultdict(lambda: [0, 0.0])
rows = list(csv.DictReader(open(sys.argv[1])))
# keep the last third of the dispatches (the timed replays)
rows = rows[len(rows) * 2 // 3:]
for r in rows:
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:60]
    k = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    g[k][0] += 1
    g[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in g.values())
print("total us", tot, "dispatches", sum(v[0] for v in g.values()))
for k, (c, t) in sorted(g.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{k[0]:60s} grid {k[1]:>8s} wg {k[2]:>5s} x{c:5d} {t:10.1f} us avg {t / c:7.1f}")
PY
rm -rf $O/raw
head -60 $O/by_shape.txt
