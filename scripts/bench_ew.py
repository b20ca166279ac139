"""Micro-benchmark of the HBM-bound entry points at the step's main activation shapes (bf16, N=8).
Prints algorithmic GB/s per (entry, shape).  Tuning knobs are read by the library from the environment."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bias_gan_amd  # noqa: E402,F401
from bias_gan_amd import _lib as L  # noqa: E402

DEV = torch.device("cuda", 0)
SHAPES = [(8, 576, 384, 128), (8, 288, 192, 256), (8, 144, 96, 728), (8, 72, 48, 728), (16, 72, 48, 728), (8, 72, 48, 1536)]
only = sys.argv[1:] or None


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    dt = L.BF16
    for (n, h, w, c) in SHAPES:
        cp = (c + 7) // 8 * 8
        rows = n * h * w
        nbytes = rows * cp * 2
        # rotate over several buffers so that nothing is served from the 256 MB Infinity Cache
        nb = max(2, int(1.5e9 // nbytes))
        xs = [torch.randn(n, h, w, cp, device=DEV).to(torch.bfloat16) for _ in range(nb)]
        gs = [torch.randn(n, h, w, cp, device=DEV).to(torch.bfloat16) for _ in range(nb)]
        ys = [torch.empty(n, h, w, cp, device=DEV, dtype=torch.bfloat16) for _ in range(nb)]
        f64 = lambda *s: torch.zeros(*s, device=DEV, dtype=torch.float64)  # noqa: E731
        f32 = lambda *s: torch.zeros(*s, device=DEV)  # noqa: E731
        gamma, beta = torch.rand(cp, device=DEV) + 0.5, torch.randn(cp, device=DEV) * 0.1
        s, ss = f64(cp), f64(cp)
        L.call("bg_norm_stats", dt, xs[0].data_ptr(), rows, cp, cp, 1, s.data_ptr(), ss.data_ptr())
        mean, rstd = f32(cp), f32(cp)
        s1, s2 = f64(cp), f64(cp)
        wdw = torch.randn(3, 3, cp, device=DEV).to(torch.bfloat16)
        dwg = f32(3, 3, cp)
        ctr = [0]

        def nxt():
            ctr[0] = (ctr[0] + 1) % nb
            return ctr[0]

        def fwd_stats():
            i = nxt()
            L.call("bg_norm_act_fwd_stats", dt, xs[i].data_ptr(), cp, s.data_ptr(), ss.data_ptr(), gamma.data_ptr(),
                   beta.data_ptr(), 1e-5, 0.1, None, None, mean.data_ptr(), rstd.data_ptr(), None, 0, ys[i].data_ptr(), cp,
                   rows, cp, 1, 1)

        scale, shift = torch.rand(cp, device=DEV) + 0.5, torch.randn(cp, device=DEV) * 0.1

        def fwd_plain():
            i = nxt()
            L.call("bg_norm_act_fwd", dt, xs[i].data_ptr(), cp, scale.data_ptr(), shift.data_ptr(), None, 0, ys[i].data_ptr(), cp,
                   rows, cp, 1, 1)

        def finalize():
            L.call("bg_norm_finalize", s.data_ptr(), ss.data_ptr(), rows, 1, cp, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, None, None,
                   mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())

        def stats():
            i = nxt()
            L.call("bg_norm_stats", dt, xs[i].data_ptr(), rows, cp, cp, 1, s1.data_ptr(), s2.data_ptr())

        def bwd_reduce():
            i = nxt()
            L.call("bg_norm_act_bwd_reduce", dt, gs[i].data_ptr(), cp, None, cp, xs[i].data_ptr(), cp, mean.data_ptr(),
                   rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, cp, 1, 1, s1.data_ptr(), s2.data_ptr())

        def bwd_apply():
            i = nxt()
            L.call("bg_norm_act_bwd_apply_stats", dt, gs[i].data_ptr(), cp, None, cp, xs[i].data_ptr(), cp, s1.data_ptr(),
                   s2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1, None, None,
                   ys[i].data_ptr(), cp, None, 0, rows, cp, 1, 1)

        desc = L.DwDesc(dt, n, h, w, cp, h, w, 1, 1, cp, cp)

        def dw_fwd():
            i = nxt()
            L.call("bg_dwconv3x3_fwd", desc, xs[i].data_ptr(), wdw.data_ptr(), ys[i].data_ptr())

        def dw_bwd_data():
            i = nxt()
            L.call("bg_dwconv3x3_bwd_data", desc, gs[i].data_ptr(), wdw.data_ptr(), ys[i].data_ptr())

        def dw_bwd_weight():
            i = nxt()
            L.call("bg_dwconv3x3_bwd_weight", desc, xs[i].data_ptr(), gs[i].data_ptr(), dwg.data_ptr())

        def dw_fwd_pre():
            i = nxt()
            L.call("bg_dwconv3x3_fwd_pre", desc, xs[i].data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, 1, wdw.data_ptr(),
                   ys[i].data_ptr())

        def dw_bwd_weight_pre():
            i = nxt()
            L.call("bg_dwconv3x3_bwd_weight_pre", desc, xs[i].data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, 1,
                   gs[i].data_ptr(), dwg.data_ptr())

        def copy():
            i = nxt()
            ys[i].copy_(xs[i])

        cases = [("copy(torch)", copy, 2), ("norm_stats", stats, 1), ("norm_act_fwd_stats", fwd_stats, 2), ("norm_act_fwd(plain)", fwd_plain, 2),
                 ("norm_finalize", finalize, 0),
                 ("bwd_reduce", bwd_reduce, 2), ("bwd_apply_stats", bwd_apply, 3), ("dw_fwd", dw_fwd, 2),
                 ("dw_bwd_data", dw_bwd_data, 2), ("dw_bwd_weight", dw_bwd_weight, 2), ("dw_fwd_pre", dw_fwd_pre, 2),
                 ("dw_bwd_weight_pre", dw_bwd_weight_pre, 2)]
        for name, fn, k in cases:
            if only and not any(o in name for o in only):
                continue
            t = timeit(fn)
            print(f"{n}x{h}x{w}x{c:5d} {name:20s} {t * 1e6:8.1f} us  {k * nbytes / t * 1e-9:7.0f} GB/s", flush=True)
        del xs, gs, ys
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
