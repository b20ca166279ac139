"""The fused depthwise/BatchNorm backward kernel against the three kernels it replaces, interleaved in one process.
usage: bench_dwfused.py [BATCH ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

SHAPES = [(728, 72, 48, 1), (728, 144, 96, 1), (256, 288, 192, 1), (128, 576, 384, 1), (1536, 72, 48, 2), (1024, 72, 48, 1)]
batches = [int(a) for a in sys.argv[1:]] or [8, 16]
if os.environ.get("ONE"):
    SHAPES = SHAPES[:1]


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


for N in batches:
    for c, h, w, d in SHAPES:
        ld = (c * 2 + 63) // 64 * 32
        rows = N * h * w
        if rows * ld * 2 > 4e9:
            continue
        x = torch.randn(N, h, w, ld, device="cuda").bfloat16()[..., :c]
        g = torch.randn(N, h, w, ld, device="cuda").bfloat16()[..., :c]
        da = torch.empty(N, h, w, ld, device="cuda", dtype=torch.bfloat16)[..., :c]
        wk = (torch.randn(3, 3, c, device="cuda") * 0.3).bfloat16()
        f32 = lambda *s: torch.randn(*s, device="cuda").abs() + 0.5
        mean, rstd, scale, shift = f32(1, c), f32(1, c), f32(1, c), torch.randn(1, c, device="cuda")
        gamma, beta = f32(c), f32(c)
        dw = torch.zeros(3, 3, c, device="cuda")
        s = torch.zeros(2, 1, c, device="cuda", dtype=torch.float64)
        desc_d = L.DwDesc(L.BF16, N, h, w, c, h, w, 1, d, ld, ld)
        t1 = timeit(lambda: L.call("bg_dwconv3x3_bwd_data", desc_d, g.data_ptr(), wk.data_ptr(), da.data_ptr()))
        t2 = timeit(lambda: L.call("bg_dwconv3x3_bwd_weight_pre", desc_d, x.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, 1, g.data_ptr(), dw.data_ptr()))
        t3 = timeit(lambda: L.call("bg_norm_act_bwd_reduce", L.BF16, da.data_ptr(), ld, None, 0, x.data_ptr(), ld, mean.data_ptr(), rstd.data_ptr(),
                                   gamma.data_ptr(), beta.data_ptr(), rows, c, 1, 1, s[0].data_ptr(), s[1].data_ptr()))
        fused = lambda dwp: L.call("bg_dwconv3x3_bwd_fused", desc_d, g.data_ptr(), wk.data_ptr(), x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                   mean.data_ptr(), rstd.data_ptr(), 1, 1, da.data_ptr(), ld, dwp, s[0].data_ptr(), s[1].data_ptr())
        tf = timeit(lambda: fused(dw.data_ptr()))
        tn = timeit(lambda: fused(None))
        mb = rows * c * 2e-6
        print(f"b{N:2d} C{c:5d} {h:3d}x{w:3d} d{d} ({mb:6.1f} MB/tensor): dgrad {t1:6.1f} + wgrad {t2:6.1f} + reduce {t3:6.1f} = {t1 + t2 + t3:6.1f} us | "
              f"fused {tf:6.1f} us ({3 * mb / tf * 1e-3:5.2f} TB/s) x{(t1 + t2 + t3) / tf:.2f} | without dW {tn:6.1f} us vs {t1 + t3:6.1f} x{(t1 + t3) / tn:.2f}", flush=True)
