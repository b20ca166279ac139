#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rb in ${RBS:-0 3 4 5 6 8 9 10 12 14 16 18 24}; do
  echo "== rb=$rb"
  BGAMD_DW_RB=$rb timeout -k 10 300 python scripts/bench_ew.py dw_fwd dw_bwd_data 2>&1 | grep -v amdgpu | grep -v "dw_fwd_pre" 
done > gpurun_out/rb_sweep.log 2>&1
python - <<'PY'
import re, collections
t=collections.defaultdict(dict); rb=None
for l in open('gpurun_out/rb_sweep.log'):
    m=re.match(r'== rb=(\d+)', l)
    if m: rb=int(m.group(1)); continue
    m=re.match(r'(\S+)\s+(\d+)\s+(dw_\w+)\s+([\d.]+) us', l)
    if m: t[(m.group(1)+m.group(2), m.group(3))][rb]=float(m.group(4))
for k,v in t.items():
    print(k, ' '.join(f"{r}:{v[r]:.1f}" for r in sorted(v)))
PY
