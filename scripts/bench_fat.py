"""A/B of the GEMM tile families on the workload's pointwise / dense shapes (bf16): classic 64x64-per-wave tiles
(variant 0) against the library's choice (fat tiles where planned), interleaved in one process.
usage: bench_fat.py [BATCH ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

SHAPES = [  # cin, cout, k, dil, H, W
    (728, 728, 1, 1, 72, 48), (728, 1024, 1, 1, 72, 48), (1024, 1536, 1, 1, 72, 48), (1536, 1536, 1, 1, 72, 48),
    (1536, 2048, 1, 1, 72, 48), (2048, 256, 1, 1, 72, 48), (2048, 256, 3, 12, 72, 48), (1280, 256, 1, 1, 72, 48),
    (256, 728, 1, 1, 144, 96), (728, 728, 1, 1, 144, 96), (256, 256, 1, 1, 288, 192), (304, 256, 3, 1, 288, 192),
    (256, 256, 3, 1, 288, 192),
    (128, 128, 1, 1, 576, 384), (128, 128, 3, 1, 576, 384), (128, 128, 1, 1, 288, 192), (128, 256, 1, 1, 288, 192),
    (128, 48, 1, 1, 288, 192),
]
if os.environ.get("SHAPES") == "hbm":
    SHAPES = [s_ for s_ in SHAPES if s_[0] <= 256 and s_[2] == 1]
if os.environ.get("SHAPES") == "small":
    SHAPES = [s_ for s_ in SHAPES if s_[1] <= 128 or s_[0] == 728 and s_[4] == 72]
batches = [int(a) for a in sys.argv[1:]] or [8, 16]
for N in batches:
    for cin, cout, k, d, H, W in SHAPES:
        pad = d * (k - 1) // 2
        x = torch.randn(N, H, W, cin, device="cuda").bfloat16()
        cp, kp = (cin + 63) // 64 * 64, (cout + 63) // 64 * 64
        w = torch.zeros(cout, k, k, cp, device="cuda", dtype=torch.bfloat16)
        w[..., :cin] = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).bfloat16()
        wt = torch.zeros(cin, k, k, kp, device="cuda", dtype=torch.bfloat16)
        wt[..., :cout] = w[..., :cin].permute(3, 1, 2, 0)
        y = torch.empty(N, H, W, cout, device="cuda", dtype=torch.bfloat16)
        dy = torch.randn(N, H, W, cout, device="cuda").bfloat16()
        dx = torch.empty(N, H, W, cin, device="cuda", dtype=torch.bfloat16)
        st = torch.zeros(2, cout, device="cuda", dtype=torch.float64)
        desc = L.ConvDesc(L.BF16, N, H, W, cin, H, W, cout, k, k, 1, pad, d, cin, cout)
        flops = 2.0 * N * H * W * cout * cin * k * k
        fns = (("fwd_stats", lambda: L.call("bg_conv2d_fwd_stats", desc, x.data_ptr(), w.data_ptr(), y.data_ptr(),
                                            st[0].data_ptr(), st[1].data_ptr(), 1)),
               ("dgrad    ", lambda: L.call("bg_conv2d_bwd_data", desc, dy.data_ptr(), wt.data_ptr(), dx.data_ptr())))
        if os.environ.get("WITH_PLAIN_FWD"):     # the forward pass without its statistics epilogue
            fns += (("fwd      ", lambda: L.call("bg_conv2d_fwd", desc, x.data_ptr(), w.data_ptr(), None, y.data_ptr())),)
        for name, fn in fns:
            res = {}
            for rnd in range(3):
                for variant in (0, -1):
                    L.conv_variant(variant)
                    fn(); torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10): fn()
                    e1.record(); torch.cuda.synchronize()
                    res.setdefault(variant, []).append(e0.elapsed_time(e1) / 10 * 1e3)
            L.conv_variant(-1)
            a, b = min(res[0]), min(res[-1])
            print(f"b{N:2d} {cin:4d}->{cout:4d} k{k} d{d:2d} {H:3d}x{W:3d} {name}: classic {a:7.1f} us {flops / a * 1e-6:6.0f} TF | "
                  f"auto {b:7.1f} us {flops / b * 1e-6:6.0f} TF | x{a / b:.2f}", flush=True)
