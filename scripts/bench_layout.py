"""bg_nchw_to_nhwc / bg_nhwc_to_nchw on the step's boundary tensors (fp32 NCHW fields <-> bf16 NHWC): microseconds and TB/s.
BGAMD_LAYOUT_LDS=0: the four-pixel register forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

for n, c, h, w in ((8, 16, 1152, 768), (16, 16, 1152, 768), (8, 16, 256, 256), (4, 32, 2304, 1536)):
    x = torch.randn(n, c, h, w, device="cuda")
    nh = torch.empty(n, h, w, c, device="cuda", dtype=torch.bfloat16)
    back = torch.empty_like(x)
    byt = x.numel() * 6.0
    res = []
    for name, fn in (("nchw->nhwc", lambda: L.call("bg_nchw_to_nhwc", L.BF16, x.data_ptr(), nh.data_ptr(), n, c, h * w, c, c)),
                     ("nhwc->nchw", lambda: L.call("bg_nhwc_to_nchw", L.BF16, nh.data_ptr(), c, back.data_ptr(), n, c, h * w))):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        res.append(f"{name} {best:7.1f} us {byt / best * 1e-6:5.2f} TB/s")
    assert torch.equal(back, x.bfloat16().float())
    print(f"{n:2d} x {c} x {h} x {w}: " + " | ".join(res), flush=True)
