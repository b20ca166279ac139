import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/bias-gan_amd") else os.getcwd())
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L
SH = [(128, 128, 3, 1, 576, 384, 8), (128, 128, 3, 1, 576, 384, 16), (128, 128, 1, 1, 576, 384, 8), (16, 128, 3, 1, 576, 384, 8), (128, 256, 1, 1, 288, 192, 8), (256, 256, 1, 1, 288, 192, 8), (256, 256, 3, 1, 288, 192, 8)]
for cin, cout, k, d, H, W, N in SH:
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
    fns = (("fwd_stats", lambda: L.call("bg_conv2d_fwd_stats", desc, x.data_ptr(), w.data_ptr(), y.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 1)),
           ("dgrad    ", lambda: L.call("bg_conv2d_bwd_data", desc, dy.data_ptr(), wt.data_ptr(), dx.data_ptr())))
    for name, fn in fns:
        res = {}
        for rnd in range(3):
            for variant in (0, -1, 2):
                L.conv_variant(variant)
                try:
                    fn(); torch.cuda.synchronize()
                except RuntimeError as e:
                    res.setdefault(variant, []).append(float("nan")); continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): fn()
                e1.record(); torch.cuda.synchronize()
                res.setdefault(variant, []).append(e0.elapsed_time(e1) / 5 * 1e3)
        L.conv_variant(-1)
        a, b, c = min(res[0]), min(res[-1]), min(res[2])
        print(f"b{N:2d} {cin:4d}->{cout:4d} k{k} {H:3d}x{W:3d} {name}: classic {a:7.1f} us {flops / a * 1e-6:6.0f} TF | auto {b:7.1f} | fat forced {c:7.1f} us {flops / c * 1e-6:6.0f} TF", flush=True)
