"""Micro-benchmark of the MFMA conv kernels on the unique layer shapes of the
1152x768x16 workload, batch 8, bf16.  Prints TFLOP/s per
shape for forward / data-gradient / weight-gradient and the FLOP-weighted mean."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd
from bias_gan_amd import _lib as L

N = int(os.environ.get("BATCH", "8"))
# (cin, cout, k, stride, dil, H, W, count_in_G+D_forward_units)
SHAPES = [
    (16, 128, 3, 2, 1, 1152, 768, 1),
    (128, 128, 3, 1, 1, 576, 384, 1),
    (128, 128, 1, 1, 1, 576, 384, 2),
    (128, 128, 1, 1, 1, 288, 192, 1),
    (128, 256, 1, 1, 1, 288, 192, 1),
    (256, 256, 1, 1, 1, 288, 192, 1),
    (256, 728, 1, 1, 1, 144, 96, 1),
    (728, 728, 1, 1, 1, 144, 96, 1),
    (728, 728, 1, 1, 1, 72, 48, 50),
    (728, 1024, 1, 1, 1, 72, 48, 2),
    (1024, 1536, 1, 1, 1, 72, 48, 1),
    (1536, 1536, 1, 1, 1, 72, 48, 1),
    (1536, 2048, 1, 1, 1, 72, 48, 1),
    (2048, 256, 3, 1, 12, 72, 48, 3),
    (304, 256, 3, 1, 1, 288, 192, 1),
    (256, 256, 3, 1, 1, 288, 192, 1),
]
dev = "cuda"
tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
reps = 5
for cin, cout, k, s, d, H, W, cnt in SHAPES:
    pad = d * (k - 1) // 2
    Ho = (H + 2 * pad - d * (k - 1) - 1) // s + 1
    Wo = (W + 2 * pad - d * (k - 1) - 1) // s + 1
    LDP = int(os.environ.get("LDPAD", "0"))
    ldx = (cin + LDP - 1) // LDP * LDP if LDP else cin
    ldy = (cout + LDP - 1) // LDP * LDP if LDP else cout
    x = torch.randn(N, H, W, ldx, device=dev).bfloat16()[..., :cin]
    cp, kp = (cin + 63) // 64 * 64, (cout + 63) // 64 * 64
    w = torch.zeros(cout, k, k, cp, device=dev, dtype=torch.bfloat16)
    w[..., :cin] = (torch.randn(cout, k, k, cin, device=dev) * 0.05).bfloat16()
    wt = torch.zeros(cin, k, k, kp, device=dev, dtype=torch.bfloat16)
    wt[..., :cout] = w[..., :cin].permute(3, 1, 2, 0)
    y = torch.empty(N, Ho, Wo, ldy, device=dev, dtype=torch.bfloat16)[..., :cout]
    dy = torch.randn(N, Ho, Wo, ldy, device=dev).bfloat16()[..., :cout]
    dx = torch.empty(N, H, W, ldx, device=dev, dtype=torch.bfloat16)[..., :cin]
    dw = torch.zeros(cout, k, k, cin, device=dev)
    desc = L.ConvDesc(L.BF16, N, H, W, cin, Ho, Wo, cout, k, k, s, pad, d, ldx, ldy)
    flops = 2.0 * N * Ho * Wo * cout * cin * k * k
    res = []
    for name, fn in (("fwd", lambda: L.call("bg_conv2d_fwd", desc, x.data_ptr(), w.data_ptr(), None, y.data_ptr())),
                     ("dgrad", lambda: L.call("bg_conv2d_bwd_data", desc, dy.data_ptr(), wt.data_ptr(), dx.data_ptr())),
                     ("wgrad", lambda: L.call("bg_conv2d_bwd_weight", desc, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append(flops / ms * 1e-9)
        tot[name][0] += flops * cnt; tot[name][1] += ms * cnt
    print(f"{cin:5d}->{cout:5d} k{k} s{s} d{d:2d} {H:4d}x{W:4d} x{cnt:2d}  {flops*1e-9:8.1f} GF  fwd {res[0]:7.1f}  dgrad {res[1]:7.1f}  wgrad {res[2]:7.1f} TF/s")
for k_, (f, ms) in tot.items():
    print(f"weighted {k_}: {f / ms * 1e-9:7.1f} TF/s, {ms:8.2f} ms per (G-ish) forward unit")
