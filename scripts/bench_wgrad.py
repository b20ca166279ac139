"""Weight-gradient A/B: per-layer kernel (float-atomic pixel splits) against the grouped gang launch, bf16.
usage: bench_wgrad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

CASES = [(48, 8 * 72 * 48, 728, 728), (48, 16 * 72 * 48, 728, 728), (1, 8 * 72 * 48, 1536, 1536), (1, 8 * 72 * 48, 1536, 2048),
         (1, 8 * 72 * 48, 728, 1024), (1, 8 * 144 * 96, 728, 728), (1, 8 * 288 * 192, 256, 256), (1, 8 * 576 * 384, 128, 128),
         (1, 16 * 576 * 384, 128, 128), (1, 8 * 72 * 48, 2048, 256), (1, 8 * 72 * 48, 1280, 256)]
for nl, m, cin, cout in CASES[:int(os.environ.get('CASES_ONLY', len(CASES)))]:
    nbuf = min(nl, 6)   # operands of the group's layers: a few distinct buffers reused (memory)
    xs = [torch.randn(m, cin, device="cuda").bfloat16() for _ in range(nbuf)]
    gs = [(torch.randn(m, cout, device="cuda") * 0.01).bfloat16() for _ in range(nbuf)]
    dws = [torch.zeros(cout, cin, device="cuda") for _ in range(nl)]
    tbl = torch.tensor([[xs[l % nbuf].data_ptr(), gs[l % nbuf].data_ptr(), dws[l].data_ptr(), 0] for l in range(nl)],
                       dtype=torch.int64)
    desc = L.ConvDesc(L.BF16, 1, 1, m, cin, 1, m, cout, 1, 1, 1, 0, 1, cin, cout)
    flops = 2.0 * nl * m * cin * cout

    def per_layer():
        for l in range(nl):
            L.call("bg_conv2d_bwd_weight", desc, xs[l % nbuf].data_ptr(), gs[l % nbuf].data_ptr(), dws[l].data_ptr(), None)

    def grouped():
        L.call("bg_conv2d_bwd_weight_grouped", L.BF16, tbl.data_ptr(), nl, m, cin, cout, cin, cout)
    res = {}
    for rnd in range(3):
        for name, fn in (("per-layer", per_layer), ("grouped", grouped)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 3 * 1e3)
    a, b = min(res["per-layer"]), min(res["grouped"])
    print(f"{nl:2d} x [{m:7d} px, {cin:4d} -> {cout:4d}]: per-layer {a:9.1f} us {flops / a * 1e-6:6.0f} TF | grouped {b:9.1f} us "
          f"{flops / b * 1e-6:6.0f} TF | x{a / b:.2f}", flush=True)


# k x k layers: per-layer kernel against the gang kernel with the taps as gang members / layers
KK = [(8, 288, 192, 256, 256, 3, 1), (8, 288, 192, 304, 256, 3, 1), (16, 288, 192, 256, 256, 3, 1), (8, 72, 48, 2048, 256, 3, 12),
      (8, 72, 48, 2048, 256, 3, 6), (8, 144, 96, 256, 256, 3, 1), (8, 72, 48, 728, 728, 3, 1), (2, 64, 64, 256, 256, 3, 1)]
for n, h, w, cin, cout, k, dil in (KK if not os.environ.get("CASES_ONLY") else KK[:int(os.environ["CASES_ONLY"])]):
    x = torch.randn(n, h, w, cin, device="cuda").bfloat16()
    g = (torch.randn(n, h, w, cout, device="cuda") * 0.01).bfloat16()
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    desc = L.ConvDesc(L.BF16, n, h, w, cin, h, w, cout, k, k, 1, dil * (k - 1) // 2, dil, cin, cout)
    tbl = torch.tensor([[x.data_ptr(), g.data_ptr(), dw.data_ptr(), 0]], dtype=torch.int64)
    flops = 2.0 * n * h * w * cin * cout * k * k
    fns = (("per-layer", lambda: L.call("bg_conv2d_bwd_weight", desc, x.data_ptr(), g.data_ptr(), dw.data_ptr(), None)),
           ("gang", lambda: L.call("bg_conv2d_bwd_weight_grouped_taps", desc, tbl.data_ptr(), 1)))
    res = {}
    for rnd in range(3):
        for name, fn in fns:
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 3 * 1e3)
    a, b = min(res["per-layer"]), min(res["gang"])
    print(f"{k}x{k} d{dil:2d} [{n:2d} x {h:3d} x {w:3d}, {cin:4d} -> {cout:4d}]: per-layer {a:9.1f} us {flops / a * 1e-6:6.0f} TF | gang {b:9.1f} us "
          f"{flops / b * 1e-6:6.0f} TF | x{a / b:.2f}", flush=True)
