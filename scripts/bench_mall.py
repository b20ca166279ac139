"""Does a consumer find a freshly written chunk in the 256 MB Infinity Cache?  Depthwise forward (writes d) followed by the
pointwise GEMM that reads d, on 128-channel 576 x 384 maps: whole batch (8 images, 453 MB per tensor), image by image
(57 MB), and image by image with 512 MB of unrelated traffic between producer and consumer (the cache flushed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

N, H, W, C = 8, 576, 384, 128
x = torch.randn(N, H, W, C, device="cuda").bfloat16()
d = torch.empty_like(x); z = torch.empty_like(x)
wdw = (torch.randn(3, 3, C, device="cuda") * 0.3).bfloat16()
wpw = (torch.randn(C, 1, 1, C, device="cuda") * 0.05).bfloat16()
st = torch.zeros(2, C, device="cuda", dtype=torch.float64)
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
es = 2
img = H * W * C * es


def dw(n0, n1):
    L.call("bg_dwconv3x3_fwd", L.DwDesc(L.BF16, n1 - n0, H, W, C, H, W, 1, 1, C, C), x.data_ptr() + n0 * img, wdw.data_ptr(), d.data_ptr() + n0 * img)


def pw(n0, n1):
    L.call("bg_conv2d_fwd_stats", L.ConvDesc(L.BF16, n1 - n0, H, W, C, H, W, C, 1, 1, 1, 0, 1, C, C), d.data_ptr() + n0 * img,
           wpw.data_ptr(), z.data_ptr() + n0 * img, st[0].data_ptr(), st[1].data_ptr(), 1)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def whole():
    dw(0, N); pw(0, N)


def chunked(c):
    for n0 in range(0, N, c):
        dw(n0, n0 + c); pw(n0, n0 + c)


def pw_only_after_flush(c):     # consumer alone, its input evicted: the per-chunk cost without cache residency
    for n0 in range(0, N, c):
        junk.zero_(); pw(n0, n0 + c)


def flush_only(c):
    for n0 in range(0, N, c):
        junk.zero_()


def pw_only_warm(c):            # consumer alone right after the producer of the same chunk (timed: producer + consumer - producer)
    for n0 in range(0, N, c):
        dw(n0, n0 + c); pw(n0, n0 + c)


def dw_only(c):
    for n0 in range(0, N, c):
        dw(n0, n0 + c)


print(f"whole batch  dw+pw: {timed(whole):8.1f} us   (dw alone {timed(lambda: dw(0, N)):7.1f}, pw alone {timed(lambda: pw(0, N)):7.1f})")
for c in (1, 2, 4):
    t_all, t_dw = timed(lambda: chunked(c)), timed(lambda: dw_only(c))
    t_cold = timed(lambda: pw_only_after_flush(c)) - timed(lambda: flush_only(c))
    print(f"chunks of {c}: dw+pw {t_all:8.1f} us  (dw launches alone {t_dw:7.1f} -> pw behind its producer {t_all - t_dw:7.1f}; pw with its input evicted {t_cold:7.1f})")
