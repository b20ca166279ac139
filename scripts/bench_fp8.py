"""bf16 fat-tile GEMM against the fp8 one (block-scaled MFMA 16x16x128) on the pointwise / dense shapes of the
2304x1536x32 workload (c5), interleaved in one process; also the quantiser alone.  usage: bench_fp8.py [BATCH ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

SHAPES = [  # cin, cout, k, dil, H, W   (2304x1536 fields: /16 = 144x96, /8 = 288x192, /4 = 576x384)
    (728, 728, 1, 1, 144, 96), (728, 1024, 1, 1, 144, 96), (1536, 1536, 1, 1, 144, 96), (1536, 2048, 1, 1, 144, 96),
    (2048, 256, 3, 12, 144, 96), (1280, 256, 1, 1, 144, 96), (256, 728, 1, 1, 288, 192), (728, 728, 1, 1, 288, 192),
    (256, 256, 1, 1, 576, 384), (304, 256, 3, 1, 576, 384), (728, 728, 1, 1, 72, 48),
]
up = lambda c, g: (c + g - 1) // g * g
batches = [int(a) for a in sys.argv[1:]] or [2, 8]


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
    for cin, cout, k, d, H, W in SHAPES:
        pad = d * (k - 1) // 2
        M = N * H * W
        if M * max(cin, cout) * 2 > 6e9:
            continue
        x = torch.randn(N, H, W, cin, device="cuda").bfloat16()
        cp, kp = up(cin, 64), up(cout, 64)
        w = torch.zeros(cout, k, k, cp, device="cuda", dtype=torch.bfloat16)
        w[..., :cin] = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).bfloat16()
        wt = torch.zeros(cin, k, k, kp, device="cuda", dtype=torch.bfloat16)
        wt[..., :cout] = w[..., :cin].permute(3, 1, 2, 0)
        y = torch.empty(N, H, W, cout, device="cuda", dtype=torch.bfloat16)
        dy = (torch.randn(N, H, W, cout, device="cuda") * 1e-3).bfloat16()
        dx = torch.empty(N, H, W, cin, device="cuda", dtype=torch.bfloat16)
        st = torch.zeros(2, cout, device="cuda", dtype=torch.float64)
        desc = L.ConvDesc(L.BF16, N, H, W, cin, H, W, cout, k, k, 1, pad, d, cin, cout)
        # fp8 operands
        cq, kq = up(cin, 16), up(cout, 16)
        ldq, ldkq = up(cq, 64), up(kq, 64)
        xq = torch.zeros(M, ldq, dtype=torch.uint8, device="cuda")
        dyq = torch.zeros(M, ldkq, dtype=torch.uint8, device="cuda")
        ex = torch.zeros(4, dtype=torch.int32, device="cuda"); ex[1] = 8
        wm = torch.zeros(up(cout, 16), k, k, cq, device="cuda"); wm[:cout, ..., :cin] = w[..., :cin].float()
        dk = torch.zeros(wm.shape[0] * k * k * up(cq, 128), dtype=torch.uint8, device="cuda")
        dt_ = torch.zeros(cq * k * k * up(wm.shape[0], 128), dtype=torch.uint8, device="cuda")
        tbl = torch.tensor([[0, 0, 0, wm.shape[0], k * k, cq, up(cq, 128), up(wm.shape[0], 128)]], dtype=torch.int64, device="cuda")
        ew, ws = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
        L.call("bg_pack_conv_weights_fp8", wm.data_ptr(), dk.data_ptr(), dt_.data_ptr(), tbl.data_ptr(), 1, dk.numel() + dt_.numel(), ew.data_ptr(), ws.data_ptr())
        qx = lambda: L.call("bg_quant_fp8", L.BF16, x.data_ptr(), cin, M, cin, xq.data_ptr(), ldq, cq, 0, ex.data_ptr(), ex[2:].data_ptr())
        qdy = lambda: L.call("bg_quant_fp8", L.BF16, dy.data_ptr(), cout, M, cout, dyq.data_ptr(), ldkq, kq, 1, ex[1:].data_ptr(), ex[3:].data_ptr())
        qx(); qdy()
        d8f = L.ConvDesc(L.BF16, N, H, W, cq, H, W, cout, k, k, 1, pad, d, ldq, cout)
        d8b = L.ConvDesc(L.BF16, N, H, W, cin, H, W, kq, k, k, 1, pad, d, cin, ldkq)
        flops = 2.0 * M * cout * cin * k * k
        t = {}
        t["f16"] = timeit(lambda: L.call("bg_conv2d_fwd_stats", desc, x.data_ptr(), w.data_ptr(), y.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 1))
        t["f8"] = timeit(lambda: L.call("bg_conv2d_fwd_fp8", d8f, xq.data_ptr(), dk.data_ptr(), ex.data_ptr(), ew.data_ptr(), None, y.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 1))
        t["d16"] = timeit(lambda: L.call("bg_conv2d_bwd_data", desc, dy.data_ptr(), wt.data_ptr(), dx.data_ptr()))
        t["d8"] = timeit(lambda: L.call("bg_conv2d_bwd_data_fp8", d8b, dyq.data_ptr(), 1, dt_.data_ptr(), ex[1:].data_ptr(), ew.data_ptr(), dx.data_ptr()))
        t["qx"], t["qdy"] = timeit(qx), timeit(qdy)
        gb = lambda n, c: M * c * 3 / n * 1e-3   # bf16 in + fp8 out
        print(f"b{N:2d} {cin:4d}->{cout:4d} k{k} d{d:2d} {H:3d}x{W:3d} fwd bf16 {t['f16']:7.1f} us {flops / t['f16'] * 1e-6:5.0f} TF | fp8 {t['f8']:7.1f} us "
              f"{flops / t['f8'] * 1e-6:5.0f} TF x{t['f16'] / t['f8']:.2f} | dgrad bf16 {t['d16']:7.1f} {flops / t['d16'] * 1e-6:5.0f} TF | fp8 {t['d8']:7.1f} "
              f"{flops / t['d8'] * 1e-6:5.0f} TF x{t['d16'] / t['d8']:.2f} | quant x {t['qx']:6.1f} us {gb(t['qx'], cin):5.0f} GB/s dy {t['qdy']:6.1f} us", flush=True)
