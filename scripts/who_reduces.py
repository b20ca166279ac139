"""Which call sites still launch the stand-alone BatchNorm-backward reduce pass (bg_norm_act_bwd_reduce) in one eager step of the
headline configuration: histogram by (ops.py line, rows, channels)."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BGAMD_STEP_GRAPH"] = "0"
import torch
import bench
from bias_gan_amd import _lib as L

hist = collections.Counter()
orig = L.call
def spy(name, *a):
    if name == "bg_norm_act_bwd_reduce":
        fr = [f for f in traceback.extract_stack() if f.filename.endswith("ops.py")]
        hist[(fr[-1].lineno if fr else -1, "rows", int(a[11]), "C", int(a[12]), "groups", int(a[13]), "act", int(a[14]), "y" if a[3] else "-")] += 1
    return orig(name, *a)
L.call = spy
import bias_gan_amd.ops as ops
ops.L.call = spy
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-profile", "--no-host-floor"]
try:
    bench.main()
except SystemExit:
    pass
for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
    print(v, k)
