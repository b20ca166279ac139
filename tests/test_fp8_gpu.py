"""fp8 operand path (BASELINE.json configs[4]; include/bgamd.h "fp8 operand path"): the quantiser, the delayed-scaling
roll, the e4m3 weight pack and the block-scaled-MFMA convolution kernels, each against a plain PyTorch fp32 CPU
evaluation.  Runs on the MI355X.

Tolerances: the quantiser and the pack are BIT-EXACT against torch's own float8 casts of the same scaled values (both
round to nearest even; the kernels clamp to the format's largest finite value first).  The convolutions are compared
with an fp32 convolution of the SAME fp8-rounded operands (decoded from the bytes the kernels read): what remains is
fp32 accumulation order and the bf16 rounding of the output, 1e-2 of the output's max magnitude like every bf16 kernel.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import bias_gan_amd  # noqa: E402,F401
from bias_gan_amd import _lib as L  # noqa: E402

DEV = "cuda"
F8 = {L.FP8_E4M3: (torch.float8_e4m3fn, 448.0, 9), L.FP8_E5M2: (torch.float8_e5m2, 57344.0, 16)}


def up(c, g):
    return (c + g - 1) // g * g


def decode(q, fmt):
    """uint8 tensor of fp8 bytes -> fp32 (CPU)."""
    return q.cpu().view(F8[fmt][0]).float()


def torch_quant(x, e, fmt):
    """The specification of bg_quant_fp8: clamp(x * 2^e) to the largest finite value, round to nearest even."""
    dt, lim, _ = F8[fmt]
    return (x.float() * 2.0 ** e).clamp(-lim, lim).to(dt).view(torch.uint8)


def exp_for(amax, top, margin=0):
    """Largest e with amax * 2^e <= 0.875 * 2^top, minus the margin."""
    if amax <= 0:
        return 0
    m, k = math.frexp(amax)
    return (top if m <= 0.875 else top - 1) - k - margin


def quant(x2d, c, e, fmt, amax=None):
    """x2d: [rows, ld] bf16/fp32 cuda -> ([rows, ldq] uint8, Cq)."""
    rows, ld = x2d.shape
    cq = up(c, 16)
    ldq = up(cq, 64)
    xq = torch.full((rows, ldq), 0x55, dtype=torch.uint8, device=DEV)
    ex = torch.tensor([e], dtype=torch.int32, device=DEV)
    L.call("bg_quant_fp8", L.dt(x2d.dtype), x2d.data_ptr(), ld, rows, c, xq.data_ptr(), ldq, cq, fmt, ex.data_ptr(),
           None if amax is None else amax.data_ptr())
    return xq, cq, ex


@pytest.mark.parametrize("fmt", [L.FP8_E4M3, L.FP8_E5M2])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("c", [728, 16, 256, 40])
def test_quant_fp8_bit_exact(fmt, dtype, c):
    g = torch.Generator().manual_seed(c + fmt)
    rows, ld = 301, up(c, 32) + 32
    x = torch.randn(rows, ld, generator=g) * torch.logspace(-4, 3, ld)[None, :]     # subnormals, normals and overflow
    x[0, 0], x[1, 1] = 1e9, -1e9
    xd = x.to(dtype).to(DEV)
    amax = torch.zeros(1, dtype=torch.int32, device=DEV)
    for e in (0, 3, -2):
        amax.zero_()
        xq, cq, _ = quant(xd, c, e, fmt, amax)
        torch.cuda.synchronize()
        ref = torch_quant(xd[:, :c].cpu(), e, fmt)
        got = xq.cpu()
        assert torch.equal(got[:, :c], ref), f"fmt {fmt} e {e}: {(got[:, :c] != ref).sum().item()} bytes differ"
        assert (got[:, c:cq] == 0).all() and (got[:, cq:] == 0x55).all()       # pad lanes zero, the rest untouched
        assert amax.view(torch.float32).item() == xd[:, :c].float().abs().max().item()


def test_fp8_roll_delayed_scaling():
    amaxs = [0.0, 1.0, 448.0, 449.0, 3.1e-5, 6.0e4, 0.874 * 2 ** -20, 0.876 * 2 ** -20]
    fmt = torch.tensor([0, 0, 0, 0, 1, 1, 0, 1], dtype=torch.int32, device=DEV)
    amax = torch.tensor(amaxs, dtype=torch.float32, device=DEV).view(torch.int32)
    ex = torch.full((8,), 77, dtype=torch.int32, device=DEV)
    L.call("bg_fp8_roll", ex.data_ptr(), amax.data_ptr(), fmt.data_ptr(), 8, 1)
    torch.cuda.synchronize()
    got = ex.cpu().tolist()
    want = [77] + [exp_for(a, F8[int(f)][2], 1) for a, f in zip(amaxs[1:], fmt.cpu().tolist()[1:])]   # unvisited: kept
    assert got == want, (got, want)
    assert (amax == 0).all()
    for a, f, e in zip(amaxs[1:], fmt.cpu().tolist()[1:], got[1:]):    # one bit of head-room, never more than two
        assert a * 2.0 ** (e + 1) <= F8[int(f)][1] < a * 2.0 ** (e + 2)


def pack_fp8(wk):
    """wk: dense fp32 [Kp, KH, KW, Cp] cuda -> (KRSC e4m3, CRSK e4m3, exponent tensor)."""
    kp, kh, kw, cp = wk.shape
    cpp, kpp = up(cp, 128), up(kp, 128)
    dk = torch.full((kp * kh * kw * cpp,), 0x55, dtype=torch.uint8, device=DEV)
    dtt = torch.full((cp * kh * kw * kpp,), 0x55, dtype=torch.uint8, device=DEV)
    tbl = torch.tensor([[0, 0, 0, kp, kh * kw, cp, cpp, kpp]], dtype=torch.int64, device=DEV)
    ex = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = torch.zeros(1, dtype=torch.int32, device=DEV)
    L.call("bg_pack_conv_weights_fp8", wk.data_ptr(), dk.data_ptr(), dtt.data_ptr(), tbl.data_ptr(), 1, dk.numel() + dtt.numel(),
           ex.data_ptr(), ws.data_ptr())
    return dk, dtt, ex


@pytest.mark.parametrize("shape", [(728, 1, 1, 728), (256, 3, 3, 304), (40, 1, 1, 24)])
def test_pack_conv_weights_fp8(shape):
    kp, kh, kw, cp = shape
    g = torch.Generator().manual_seed(kp)
    wk = (torch.randn(shape, generator=g) / math.sqrt(cp * kh * kw)).to(DEV)
    dk, dtt, ex = pack_fp8(wk)
    torch.cuda.synchronize()
    e = ex.item()
    assert e == exp_for(wk.abs().max().item(), 9)
    assert 224.0 < wk.abs().max().item() * 2.0 ** e <= 448.0
    cpp, kpp = up(cp, 128), up(kp, 128)
    q = torch_quant(wk.cpu(), e, L.FP8_E4M3).view(kp, kh * kw, cp)
    ref_k = torch.zeros(kp, kh * kw, cpp, dtype=torch.uint8)
    ref_k[:, :, :cp] = q
    ref_t = torch.zeros(cp, kh * kw, kpp, dtype=torch.uint8)
    ref_t[:, :, :kp] = q.permute(2, 1, 0)
    assert torch.equal(dk.cpu().view_as(ref_k), ref_k) and torch.equal(dtt.cpu().view_as(ref_t), ref_t)


FP8_CONV_CASES = [
    # n, h, w, cin, cout, k, stride, pad, dil
    (2, 24, 20, 728, 728, 1, 1, 0, 1),       # the middle flow's pointwise layer: 256 x 224 tiles
    (2, 9, 7, 24, 40, 1, 1, 0, 1),           # Cout <= 128: the 128-row tile, one K-step with a tail
    (1, 16, 16, 304, 256, 3, 1, 1, 1),       # decoder 3 x 3
    (3, 5, 6, 2048, 256, 3, 1, 2, 2),        # ASPP-like: dilated taps mostly in the padding, 256 x 112 tiles
    (2, 11, 9, 128, 256, 1, 2, 0, 1),        # strided skip convolution
    (2, 21, 19, 16, 128, 3, 2, 1, 1),
    (1, 40, 36, 1536, 1536, 1, 1, 0, 1),
]


@pytest.mark.parametrize("case", FP8_CONV_CASES)
@pytest.mark.parametrize("with_stats", [False, True])
def test_conv2d_fwd_fp8(case, with_stats):
    n, h, w, cin, cout, k, s, p, d = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, h, w, cin, generator=g) * 1.5
    wt = torch.randn(cout, k, k, cin, generator=g) / math.sqrt(cin * k * k)
    coutp = up(cout, 8)
    xd = x.to(torch.bfloat16).to(DEV).view(n * h * w, cin)
    ex_val = exp_for(xd.float().abs().max().item(), 9, 1)
    xq, cq, ex = quant(xd, cin, ex_val, L.FP8_E4M3)
    wk = torch.zeros(coutp, k, k, cq, device=DEV)
    wk[:cout, :, :, :cin] = wt.to(DEV)
    dk, _, ew = pack_fp8(wk)
    ho = (h + 2 * p - d * (k - 1) - 1) // s + 1
    wo = (w + 2 * p - d * (k - 1) - 1) // s + 1
    ldy = coutp + 24
    y = torch.full((n, ho, wo, ldy), 7.0, dtype=torch.bfloat16, device=DEV)
    desc = L.ConvDesc(L.BF16, n, h, w, cq, ho, wo, coutp, k, k, s, p, d, xq.shape[1], ldy)
    groups = 2 if (with_stats and n % 2 == 0) else 1
    st = torch.zeros(2, groups, coutp, dtype=torch.float64, device=DEV) if with_stats else None
    bias = None if with_stats else (torch.randn(coutp, generator=g)).to(DEV)
    L.call("bg_conv2d_fwd_fp8", desc, xq.data_ptr(), dk.data_ptr(), ex.data_ptr(), ew.data_ptr(), L.ptr(bias), y.data_ptr(),
           None if st is None else st[0].data_ptr(), None if st is None else st[1].data_ptr(), groups)
    torch.cuda.synchronize()
    # fp32 evaluation of the same fp8-rounded operands
    xr = decode(xq[:, :cin], L.FP8_E4M3).view(n, h, w, cin) * 2.0 ** -ex.item()
    wr = decode(dk.view(coutp, k * k, -1)[:cout, :, :cin], L.FP8_E4M3).view(cout, k, k, cin) * 2.0 ** -ew.item()
    ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.permute(0, 3, 1, 2), None if bias is None else bias[:cout].cpu(), s, p, d)
    got = y[..., :cout].float().cpu().permute(0, 3, 1, 2)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert err <= 1e-2 * scale, f"fp8 conv fwd {case}: max err {err:.3e} of {scale:.3e}"
    assert (y[..., coutp:].float() == 7.0).all(), "lanes beyond Cout were written"
    # the rounded operands differ from the originals by fp8's 2^-4 relative step: a sanity bound against the unquantised
    # bf16 convolution (catches a wrong exponent or a lost K-slab, which the same-operand check cannot see if the
    # decode went wrong the same way)
    ref0 = F.conv2d(xd.float().cpu().view(n, h, w, cin).permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), None if bias is None else bias[:cout].cpu(), s, p, d)
    rms = ((got - ref0).pow(2).mean().sqrt() / ref0.pow(2).mean().sqrt()).item()
    assert rms <= 6e-2, f"fp8 conv vs unquantised: rms-rel {rms:.3e}"
    if with_stats:
        yy = y[..., :cout].float().cpu().view(groups, -1, cout).double()
        assert torch.allclose(st[0, :, :cout].cpu(), yy.sum(1), rtol=1e-6, atol=1e-6 * scale * yy.shape[1])
        assert torch.allclose(st[1, :, :cout].cpu(), (yy * yy).sum(1), rtol=1e-6, atol=1e-6 * scale * scale * yy.shape[1])


@pytest.mark.parametrize("case", FP8_CONV_CASES)
@pytest.mark.parametrize("fmt", [L.FP8_E5M2, L.FP8_E4M3])
def test_conv2d_bwd_data_fp8(case, fmt):
    n, h, w, cin, cout, k, s, p, d = case
    g = torch.Generator().manual_seed(12)
    ho = (h + 2 * p - d * (k - 1) - 1) // s + 1
    wo = (w + 2 * p - d * (k - 1) - 1) // s + 1
    dy = torch.randn(n, ho, wo, cout, generator=g) * 3e-4            # gradient-sized values: the exponent does the work
    wt = torch.randn(cout, k, k, cin, generator=g) / math.sqrt(cin * k * k)
    cinp = up(cin, 8)
    dyd = dy.to(torch.bfloat16).to(DEV).view(n * ho * wo, cout)
    e_val = exp_for(dyd.float().abs().max().item(), F8[fmt][2], 1)
    dyq, cq, edy = quant(dyd, cout, e_val, fmt)
    wk = torch.zeros(cq, k, k, cinp, device=DEV)
    wk[:cout, :, :, :cin] = wt.to(DEV)
    _, dtt, ew = pack_fp8(wk)
    ldx = cinp + 8
    dx = torch.full((n, h, w, ldx), 7.0, dtype=torch.bfloat16, device=DEV)
    desc = L.ConvDesc(L.BF16, n, h, w, cinp, ho, wo, cq, k, k, s, p, d, ldx, dyq.shape[1])
    L.call("bg_conv2d_bwd_data_fp8", desc, dyq.data_ptr(), fmt, dtt.data_ptr(), edy.data_ptr(), ew.data_ptr(), dx.data_ptr())
    torch.cuda.synchronize()
    dyr = decode(dyq[:, :cout], fmt).view(n, ho, wo, cout) * 2.0 ** -edy.item()
    wr = decode(dtt.view(cinp, k * k, -1)[:cin, :, :cout], L.FP8_E4M3).view(cin, k, k, cout) * 2.0 ** -ew.item()   # CRSK
    w_oihw = wr.permute(3, 0, 1, 2)
    ref = torch.nn.grad.conv2d_input((n, cin, h, w), w_oihw, dyr.permute(0, 3, 1, 2), s, p, d)
    got = dx[..., :cin].float().cpu().permute(0, 3, 1, 2)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert err <= 1e-2 * scale, f"fp8 conv dgrad {case} fmt {fmt}: max err {err:.3e} of {scale:.3e}"
    assert (dx[..., cinp:].float() == 7.0).all()
    ref0 = torch.nn.grad.conv2d_input((n, cin, h, w), wt.permute(0, 3, 1, 2), dyd.float().cpu().view(n, ho, wo, cout).permute(0, 3, 1, 2), s, p, d)
    rms = ((got - ref0).pow(2).mean().sqrt() / ref0.pow(2).mean().sqrt()).item()
    assert rms <= (1.2e-1 if fmt == L.FP8_E5M2 else 6e-2), f"fp8 dgrad vs unquantised: rms-rel {rms:.3e}"


def test_fp8_mfma_operand_map_with_exact_integers():
    """Exact-integer check of the operand mapping (asymmetric operands): small integers are exact in e4m3 and their
    dot products exact in fp32, so the kernel must reproduce the integer convolution bit for bit."""
    n, h, w, cin, cout = 1, 16, 14, 256, 256
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-4, 5, (n * h * w, cin), generator=g).float()
    wt = torch.randint(-3, 4, (cout, 1, 1, cin), generator=g).float()
    wt[:, 0, 0, :] += (torch.arange(cout)[:, None] % 3 == 0).float() * (torch.arange(cin)[None, :] % 5 == 0).float()   # asymmetric
    xq, cq, ex = quant(x.to(torch.bfloat16).to(DEV), cin, 0, L.FP8_E4M3)
    dk, _, ew = pack_fp8(wt.to(DEV))
    y = torch.zeros(n, h, w, cout, dtype=torch.bfloat16, device=DEV)
    desc = L.ConvDesc(L.BF16, n, h, w, cq, h, w, cout, 1, 1, 1, 0, 1, xq.shape[1], cout)
    L.call("bg_conv2d_fwd_fp8", desc, xq.data_ptr(), dk.data_ptr(), ex.data_ptr(), ew.data_ptr(), None, y.data_ptr(), None, None, 1)
    torch.cuda.synchronize()
    ref = (x @ wt.view(cout, cin).t()).to(torch.bfloat16).float()      # integers < 2^8 in magnitude survive bf16 or round alike
    assert torch.equal(y.float().cpu().view(-1, cout), ref)
