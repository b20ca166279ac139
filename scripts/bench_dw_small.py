"""Depthwise 3x3 forward on small maps, graph-replayed: plain, with BatchNorm + LeakyReLU from precomputed scale / shift tables,
and with the tables derived in-kernel from the producer's fp64 sums (what the step runs).  What does the in-kernel table cost?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

for n, h, w, c in ((8, 16, 16, 728), (8, 32, 32, 728), (8, 72, 48, 728), (16, 16, 16, 728)):
    nb = 8
    xs = [torch.randn(n, h, w, c, device="cuda").bfloat16() for _ in range(nb)]
    ys = [torch.empty(n, h, w, c, device="cuda", dtype=torch.bfloat16) for _ in range(nb)]
    wdw = (torch.randn(3, 3, c, device="cuda") * 0.3).bfloat16()
    f32 = lambda *s: torch.rand(*s, device="cuda") + 0.5
    gamma, beta, scale, shift = f32(c), f32(c), f32(1, c), f32(1, c)
    mean, rstd, so, sh = f32(1, c), f32(1, c), f32(1, c), f32(1, c)
    rows = n * h * w
    sums = torch.stack([torch.randn(c, device="cuda", dtype=torch.float64) * rows * 0.1, (torch.rand(c, device="cuda", dtype=torch.float64) + 1.0) * rows])
    desc = L.DwDesc(L.BF16, n, h, w, c, h, w, 1, 1, c, c)
    it = [0]
    def nx():
        it[0] += 1
        return it[0] % nb
    def plain():
        i = nx(); L.call("bg_dwconv3x3_fwd", desc, xs[i].data_ptr(), wdw.data_ptr(), ys[i].data_ptr())
    def pre():
        i = nx(); L.call("bg_dwconv3x3_fwd_pre", desc, xs[i].data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, 1, wdw.data_ptr(), ys[i].data_ptr())
    def pre_stats():
        i = nx(); L.call("bg_dwconv3x3_fwd_pre_stats", desc, xs[i].data_ptr(), sums[0].data_ptr(), sums[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                         1e-5, 0.1, None, None, mean.data_ptr(), rstd.data_ptr(), so.data_ptr(), sh.data_ptr(), 1, 1, wdw.data_ptr(), ys[i].data_ptr())
    out = []
    for name, fn in (("plain", plain), ("scale/shift tables", pre), ("in-kernel from sums", pre_stats)):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(48): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): g.replay()
        e1.record(); torch.cuda.synchronize()
        out.append(f"{name} {e0.elapsed_time(e1) / 192 * 1e3:6.1f} us")
    print(f"{n:2d} x {h:3d} x {w:3d} x {c}: " + " | ".join(out), flush=True)
