"""bg_dwconv3x3_bwd_fork against the launches it replaces (data_add + depthwise weight gradient + the producer's reduce pass,
and the apply pass with / without its residual-gradient output), interleaved in one process.  usage: bench_dwfork.py [BATCH ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

SHAPES = [(728, 72, 48), (728, 144, 96), (256, 288, 192), (128, 576, 384), (1024, 72, 48)]
batches = [int(a) for a in sys.argv[1:]] or [8, 16]


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
    for c, h, w in SHAPES:
        ld = (c * 2 + 63) // 64 * 32
        rows = N * h * w
        if rows * ld * 2 > 4e9:
            continue
        mk = lambda: torch.randn(N, h, w, ld, device="cuda").bfloat16()[..., :c]
        a0, g, sk, z, t, dz, dres = mk(), mk(), mk(), mk(), mk(), mk(), mk()
        wk = (torch.randn(3, 3, c, device="cuda") * 0.3).bfloat16()
        f32 = lambda *s: torch.randn(*s, device="cuda").abs() + 0.5
        mean, rstd, gamma, beta = f32(1, c), f32(1, c), f32(c), f32(c)
        dw = torch.zeros(3, 3, c, device="cuda")
        s = torch.zeros(2, 1, c, device="cuda", dtype=torch.float64)
        d = L.DwDesc(L.BF16, N, h, w, c, h, w, 1, 1, ld, ld)
        t1 = timeit(lambda: L.call("bg_dwconv3x3_bwd_data_add", d, g.data_ptr(), wk.data_ptr(), sk.data_ptr(), ld, t.data_ptr()))
        t2 = timeit(lambda: L.call("bg_dwconv3x3_bwd_weight", d, a0.data_ptr(), g.data_ptr(), dw.data_ptr()))
        t3 = timeit(lambda: L.call("bg_norm_act_bwd_reduce", L.BF16, t.data_ptr(), ld, a0.data_ptr(), ld, z.data_ptr(), ld, mean.data_ptr(),
                                   rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, c, 1, 1, s[0].data_ptr(), s[1].data_ptr()))
        ap = lambda y, dr, act: L.call("bg_norm_act_bwd_apply_stats", L.BF16, t.data_ptr(), ld, y, ld, z.data_ptr(), ld, s[0].data_ptr(),
                                       s[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1, None, None,
                                       dz.data_ptr(), ld, dr, ld, rows, c, 1, act)
        t4 = timeit(lambda: ap(a0.data_ptr(), dres.data_ptr(), 1))     # round 3: sign from y, writes dz and the residual gradient
        t5 = timeit(lambda: ap(None, None, 0))                         # behind the fork kernel: g already activated, one output
        fk = lambda dwp: L.call("bg_dwconv3x3_bwd_fork", d, g.data_ptr(), wk.data_ptr(), a0.data_ptr(), sk.data_ptr(), ld, z.data_ptr(), ld,
                                mean.data_ptr(), rstd.data_ptr(), 1, 1, t.data_ptr(), ld, dwp, s[0].data_ptr(), s[1].data_ptr())
        tf = timeit(lambda: fk(dw.data_ptr()))
        tn = timeit(lambda: fk(None))
        mb = rows * c * 2e-6
        print(f"b{N:2d} C{c:5d} {h:3d}x{w:3d} ({mb:6.1f} MB/tensor): data_add {t1:6.1f} + wgrad {t2:6.1f} + reduce {t3:6.1f} + apply {t4:6.1f} = "
              f"{t1 + t2 + t3 + t4:6.1f} us | fork {tf:6.1f} ({5 * mb / tf:5.2f} TB/s) + apply {t5:6.1f} = {tf + t5:6.1f} us x{(t1 + t2 + t3 + t4) / (tf + t5):.2f} | "
              f"frozen weights: {t1 + t3 + t4:6.1f} vs {tn + t5:6.1f} (fork {tn:6.1f}) x{(t1 + t3 + t4) / (tn + t5):.2f}", flush=True)
