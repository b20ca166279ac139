"""Can an HBM-bound element-wise kernel run BESIDE an MFMA-bound GEMM on the same CUs?  Two HIP streams: a loop of fat-tile
GEMM launches on one, a loop of normalisation / depthwise launches on the other; wall time of both together against each
alone.  The shipped 384 x 224 tile owns the CU (512 threads x 256 VGPRs, 160 KB LDS): nothing can be co-resident.
BGAMD_FAT_NO384=1 plans 256 x 224 tiles (192 VGPRs, 120 KB LDS): a 256-thread element-wise workgroup fits beside them.
usage: bench_corun.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L

N, H, W = 8, 72, 48
dev = "cuda"


def gemm_setup(cin, cout):
    x = torch.randn(N, H, W, cin, device=dev).bfloat16()
    cp = (cin + 63) // 64 * 64
    w = torch.zeros(cout, 1, 1, cp, device=dev, dtype=torch.bfloat16)
    w[..., :cin] = (torch.randn(cout, 1, 1, cin, device=dev) * 0.05).bfloat16()
    y = torch.empty(N, H, W, cout, device=dev, dtype=torch.bfloat16)
    st = torch.zeros(2, cout, device=dev, dtype=torch.float64)
    desc = L.ConvDesc(L.BF16, N, H, W, cin, H, W, cout, 1, 1, 1, 0, 1, cin, cout)
    return lambda s: L.call_on(s, "bg_conv2d_fwd_stats", desc, x.data_ptr(), w.data_ptr(), y.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 1), (x, w, y, st)


def ew_setup(c):
    x = torch.randn(2 * N, H, W, c, device=dev).bfloat16()
    y = torch.empty_like(x)
    sc, sh = torch.rand(1, c, device=dev) + 0.5, torch.randn(1, c, device=dev)
    rows = 2 * N * H * W
    return lambda s: L.call_on(s, "bg_norm_act_fwd", L.BF16, x.data_ptr(), c, sc.data_ptr(), sh.data_ptr(), None, 0, y.data_ptr(), c, rows, c, 1, 1), (x, y, sc, sh)


def run(fa, fb, na, nb):
    sa, sb = L.side_stream(torch.device(dev), "corun-a"), L.side_stream(torch.device(dev), "corun-b")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    best = 1e9
    for _ in range(3):
        e0.record(cur)
        sa.wait_stream(cur); sb.wait_stream(cur)
        for i in range(max(na, nb)):
            if i < na: fa(sa.cuda_stream)
            if i < nb: fb(sb.cuda_stream)
        cur.wait_stream(sa); cur.wait_stream(sb)
        e1.record(cur)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


for cin, cout in ((1536, 1536), (728, 728)):
    g, keep1 = gemm_setup(cin, cout)
    e, keep2 = ew_setup(728)
    L.conv_variant(-1)
    reps = 20
    ta = run(g, e, reps, 0)
    tb = run(g, e, 0, reps)
    for ratio in (1, 2):
        tab = run(g, e, reps, reps * ratio)
        print(f"{cin}->{cout} [{'256-row tiles' if os.environ.get('BGAMD_FAT_NO384') else 'shipped tiles'}]: {reps} GEMMs alone {ta:8.1f} us ({ta / reps:6.1f} each), "
              f"{reps} element-wise (80 MB in, 80 MB out) alone {tb:8.1f} us ({tb / reps:6.1f} each); {reps} + {reps * ratio} on two streams {tab:8.1f} us "
              f"= {tab / (ta + tb * ratio):.2f} of the sum", flush=True)
