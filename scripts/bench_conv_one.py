"""One conv shape, forward with and without the statistics epilogue: microseconds per launch and TFLOP/s.
usage: bench_conv_one.py CIN COUT H W [BATCH ...]   (1x1 convolutions, bf16)
COLD=n: every launch of the replayed graph reads another of n copies of the weights (n x their size beyond the 256 MB
Infinity Cache: the weights come from HBM, as a layer's do in a training step); activations stay hot either way."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

cin, cout, H, W = (int(a) for a in sys.argv[1:5])
for N in [int(a) for a in sys.argv[5:]] or [8, 16]:
    x = torch.randn(N, H, W, cin, device="cuda").bfloat16()
    cp = (cin + 63) // 64 * 64
    w = torch.zeros(cout, 1, 1, cp, device="cuda", dtype=torch.bfloat16)
    w[..., :cin] = (torch.randn(cout, 1, 1, cin, device="cuda") * 0.05).bfloat16()
    y = torch.empty(N, H, W, cout, device="cuda", dtype=torch.bfloat16)
    st = torch.zeros(2, 2 * cout, device="cuda", dtype=torch.float64)
    op = (cout + 63) // 64 * 64
    wt = torch.zeros(cin, 1, 1, op, device="cuda", dtype=torch.bfloat16)   # the data gradient's copy: [Cin][taps][Cout padded]
    wt[..., :cout] = w[..., :cin].permute(3, 1, 2, 0)
    dx = torch.empty_like(x)
    cold = int(os.environ.get("COLD", "1"))
    ws = [w] + [w.clone() for _ in range(cold - 1)]
    wts = [wt] + [wt.clone() for _ in range(cold - 1)]
    it = [0]
    def nxt(lst):
        it[0] += 1
        return lst[it[0] % len(lst)].data_ptr()
    nl = max(50, cold)
    desc = L.ConvDesc(L.BF16, N, H, W, cin, H, W, cout, 1, 1, 1, 0, 1, cin, cout)
    flops = 2.0 * N * H * W * cout * cin
    for name, fn in (("fwd      ", lambda: L.call("bg_conv2d_fwd", desc, x.data_ptr(), nxt(ws), None, y.data_ptr())),
                     ("fwd_stats", lambda: L.call("bg_conv2d_fwd_stats", desc, x.data_ptr(), nxt(ws), y.data_ptr(),
                                                  st[0].data_ptr(), st[1].data_ptr(), 1)),
                     ("bwd_data ", lambda: L.call("bg_conv2d_bwd_data", desc, y.data_ptr(), nxt(wts), dx.data_ptr()))):
        fn(); torch.cuda.synchronize()
        # >= 50 launches replayed as one graph: the host cannot issue a 10-us kernel back to back (about 12 us per call from Python)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(nl): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (4 * nl) * 1e3
        print(f"{cin}->{cout} {H}x{W} batch {N:2d} {name}: {us:7.1f} us  {flops / us * 1e-6:7.1f} TF/s")
