"""Per-layer weight gradient: the float-atomic kernel (bg_conv2d_bwd_weight) against the workspace form
(bg_conv2d_bwd_weight_ws: plain stores of the split tiles + an ordered second pass) on the layers of the step that the gang
kernel does not take.  Prints microseconds, TFLOP/s or TB/s, the largest difference between the two results and whether
two runs of the workspace form are bit-identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

CASES = [(8, 576, 384, 128, 128, 3, 1, 1), (16, 576, 384, 128, 128, 3, 1, 1), (8, 576, 384, 128, 128, 1, 1, 1), (16, 576, 384, 128, 128, 1, 1, 1),
         (8, 1152, 768, 16, 128, 3, 2, 1), (16, 288, 192, 256, 256, 1, 1, 1), (16, 144, 96, 256, 728, 1, 1, 1), (8, 72, 48, 1024, 1024, 1, 1, 1),
         (8, 16, 16, 728, 728, 1, 1, 1)]
for n, h, w, cin, cout, k, stride, dil in CASES:
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    x = torch.randn(n, h, w, cin, device="cuda").bfloat16()
    g = (torch.randn(n, ho, wo, cout, device="cuda") * 0.01).bfloat16()
    desc = L.ConvDesc(L.BF16, n, h, w, cin, ho, wo, cout, k, k, stride, dil * (k - 1) // 2, dil, cin, cout)
    nb = L.wgrad_ws_bytes(desc)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    outs = {}
    def atomic(dw): L.call("bg_conv2d_bwd_weight", desc, x.data_ptr(), g.data_ptr(), dw.data_ptr(), None)
    def wsf(dw): L.call("bg_conv2d_bwd_weight_ws", desc, x.data_ptr(), g.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb)
    res = {}
    for name, fn in (("atomic", atomic), ("ws", wsf), ("ws2", wsf)):
        dw = torch.zeros(cout, k, k, cin, device="cuda")
        fn(dw); torch.cuda.synchronize()
        outs[name] = dw.clone()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn(dw)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
        res[name] = best
    flops = 2.0 * n * ho * wo * cin * cout * k * k
    byt = 2.0 * n * (h * w * cin + ho * wo * cout)
    diff = (outs["ws"] - outs["atomic"]).abs().max().item() / (outs["atomic"].abs().max().item() + 1e-30)
    print(f"{k}x{k} s{stride} [{n:2d} x {h:4d} x {w:3d}, {cin:4d} -> {cout:4d}] ws {nb / 2**20:6.1f} MiB: atomic {res['atomic']:8.1f} us "
          f"({flops / res['atomic'] * 1e-6:5.0f} TF, {byt / res['atomic'] * 1e-6:4.2f} TB/s) | workspace {res['ws']:8.1f} us "
          f"({flops / res['ws'] * 1e-6:5.0f} TF, {byt / res['ws'] * 1e-6:4.2f} TB/s) x{res['atomic'] / res['ws']:.2f} | rel diff {diff:.1e} | "
          f"two runs identical: {torch.equal(outs['ws'], outs['ws2'])}", flush=True)
