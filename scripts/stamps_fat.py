"""Where a fat-tile GEMM launch spends its time: per-workgroup s_memrealtime stamps of a -DBG_STAMPS diagnostic build
(BGAMD_LIB=abl_build/libbgamd_stamps.so).  usage: stamps_fat.py CIN COUT H W BATCH"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

cin, cout, H, W, N = (int(a) for a in sys.argv[1:6])
lib = L.load()
x = torch.randn(N, H, W, cin, device="cuda").bfloat16()
cp = (cin + 63) // 64 * 64
w = torch.zeros(cout, 1, 1, cp, device="cuda", dtype=torch.bfloat16)
w[..., :cin] = (torch.randn(cout, 1, 1, cin, device="cuda") * 0.05).bfloat16()
y = torch.empty(N, H, W, cout, device="cuda", dtype=torch.bfloat16)
st = torch.zeros(2, cout, device="cuda", dtype=torch.float64)
desc = L.ConvDesc(L.BF16, N, H, W, cin, H, W, cout, 1, 1, 1, 0, 1, cin, cout)
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
fn = lambda: L.call("bg_conv2d_fwd_stats", desc, x.data_ptr(), w.data_ptr(), y.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 1)
for _ in range(20): fn()   # warm clocks and caches
torch.cuda.synchronize()
lib.bg_conv_debug_stamps.argtypes = [C.c_void_p]
lib.bg_conv_debug_stamps(dbg.data_ptr())
fn(); torch.cuda.synchronize()
lib.bg_conv_debug_stamps(None)
d = dbg.view(-1, 8).cpu().numpy()
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
us = lambda a: (a - t0) * 0.01      # 100 MHz
import numpy as np
print(f"{len(d)} workgroups; all times in us from the first workgroup's start")
for name, col in (("start", 0), ("prologue issued", 1), ("first step landed+computed", 2), ("K loop done", 3), ("epilogue issued", 4), ("stores retired", 5)):
    v = us(d[:, col].astype(np.float64))
    print(f"  {name:28s} min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f}")
kl = (d[:, 3] - d[:, 2]) * 0.01
cyc = (d[:, 7] - d[:, 6]).astype(np.float64)
print(f"  K loop after its first step: median {np.median(kl):.2f} us, {np.median(cyc):.0f} cycles -> clock {np.median(cyc / np.maximum(kl, 1e-9)) * 1e-3:.2f} GHz")
print(f"  per phase (median): prologue {np.median((d[:,1]-d[:,0])*0.01):.2f}  first step {np.median((d[:,2]-d[:,1])*0.01):.2f}  rest of K loop {np.median(kl):.2f}  "
      f"epilogue {np.median((d[:,4]-d[:,3])*0.01):.2f}  store drain {np.median((d[:,5]-d[:,4])*0.01):.2f}")
