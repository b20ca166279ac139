#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 256 512 768 1024 2048; do
  echo "== blocks=$b"
  BGAMD_EWS_BLOCKS=$b BGAMD_RED_BLOCKS=$b timeout -k 10 300 python scripts/bench_ew.py norm_act_fwd_stats bwd_reduce bwd_apply 2>&1 | grep -v amdgpu
done > gpurun_out/ewb_sweep.log 2>&1
python - <<'PY'
import re, collections
t=collections.defaultdict(dict); rb=None
for l in open('gpurun_out/ewb_sweep.log'):
    m=re.match(r'== blocks=(\d+)', l)
    if m: rb=int(m.group(1)); continue
    m=re.match(r'(\S+)\s+(\d+)\s+(\S+)\s+([\d.]+) us', l)
    if m: t[(m.group(1)+m.group(2), m.group(3))][rb]=float(m.group(4))
for k,v in t.items():
    print(k, ' '.join(f"{r}:{v[r]:.1f}" for r in sorted(v)))
PY
