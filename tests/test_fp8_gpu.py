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


# ----------------------------------------------------------------------------------------------------------------------
# model level: compute_dtype = torch.float8_e4m3fn (bf16 storage, fp8 GEMM operands with delayed scaling)
import numpy as np  # noqa: E402
import torch.nn as nn  # noqa: E402

from bias_gan_amd import ops  # noqa: E402
from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg  # noqa: E402
from bias_gan_amd.gpsro_train.train_gan import GANTrainer  # noqa: E402
from bias_gan_amd.runtime import pad_to, vec_of  # noqa: E402
from bias_gan_amd.utils import losses  # noqa: E402
from bias_gan_amd.utils import parsing_helpers as ph  # noqa: E402
from oracle import gan_oracle as orc  # noqa: E402

FP8 = torch.float8_e4m3fn


def rms_err(got, ref):
    got, ref = torch.as_tensor(got).double(), torch.as_tensor(ref).double()
    return ((got - ref).pow(2).mean().sqrt() / (ref.pow(2).mean().sqrt() + 1e-30)).item()


def build(c, h, w, dtype, seeds=(21, 22)):
    gspec, dspec = orc.generator_spec(c, c, 0, "batch"), orc.discriminator_spec(c, h, w, "batch")
    G = dxg.Generator(c, c, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d, compute_dtype=dtype)
    D = dxg.Discriminator(c, normalizer=nn.BatchNorm2d, input_size=(h, w), compute_dtype=dtype)
    G.load_state_dict(orc.fill_state(gspec, seeds[0])), D.load_state_dict(orc.fill_state(dspec, seeds[1]))
    return G.to(DEV).train(), D.to(DEV).train(), gspec, dspec


def trainer(G, D, n, mode="ModifiedMinMax"):
    crit = losses.GANLoss(mode, n, torch.device(DEV))
    return GANTrainer(G, D, ph.get_optimizer(G.parameters(), "Adam", 1e-4, 1e-8, 1e-5),
                      ph.get_optimizer(D.parameters(), "Adam", 1e-4, 1e-8, 1e-5), crit, losses.L1Loss(), loss_type_gan=mode,
                      loss_weight_gp=10.0), crit


def test_fp8_mode_uses_the_fp8_kernels_after_calibration(monkeypatch):
    """The first step of an fp8-mode trainer is the calibration (bf16 GEMMs, every site records its maximum): same
    launches as the bf16 path plus the quantiser.  From the second step on the eligible convolutions launch
    bg_conv2d_fwd_fp8 / bg_conv2d_bwd_data_fp8.  Counted through the library's launch profile hook."""
    monkeypatch.setenv("BGAMD_STEP_GRAPH", "0")
    monkeypatch.setattr(ops, "_FP8_MIN_WORK", 0)      # every eligible layer (the default keeps the narrow ones on bf16 operands)
    c, h, w, n = 4, 64, 64, 2
    G, D, _, _ = build(c, h, w, FP8)
    tr, _ = trainer(G, D, n)
    x, y = (t.to(DEV) for t in orc.synthetic_fields(n, c, h, w, 1000))
    counts = []
    for _ in range(2):
        L.PROFILE = []
        tr.step(x, y)
        torch.cuda.synchronize()
        names = [p[0] for p in L.PROFILE]
        L.PROFILE = None
        counts.append({k: names.count(k) for k in ("bg_conv2d_fwd_fp8", "bg_conv2d_bwd_data_fp8", "bg_quant_fp8", "bg_conv2d_fwd_stats",
                                                    "bg_conv2d_bwd_data", "bg_fp8_roll")})
    print(counts)
    assert counts[0]["bg_conv2d_fwd_fp8"] == 0 and counts[0]["bg_conv2d_bwd_data_fp8"] == 0 and counts[0]["bg_quant_fp8"] > 100
    assert counts[1]["bg_conv2d_fwd_fp8"] > 100 and counts[1]["bg_conv2d_bwd_data_fp8"] > 100
    assert counts[1]["bg_conv2d_fwd_stats"] < counts[0]["bg_conv2d_fwd_stats"] // 4
    assert counts[0]["bg_fp8_roll"] == 2 and counts[1]["bg_fp8_roll"] == 2
    a = G.arena()
    ex = a.site_exp.cpu()
    assert a.sites_ready and (ex[1::2] > 4).sum() > 50, "gradient sites should have picked large exponents"


def test_calibrate_fp8_leaves_the_training_state_untouched(monkeypatch):
    monkeypatch.setattr(ops, "_FP8_MIN_WORK", 0)
    c, h, w, n = 4, 64, 64, 2
    G, D, _, _ = build(c, h, w, FP8)
    tr, _ = trainer(G, D, n)
    x, y = (t.to(DEV) for t in orc.synthetic_fields(n, c, h, w, 1000))
    G(x), D(y)                                   # arenas exist
    before = {k: v.clone() for net in (G, D) for k, v in net.state_dict().items()}
    torch.manual_seed(5)
    r0 = torch.get_rng_state()
    tr.calibrate_fp8(x, y)
    assert torch.equal(torch.get_rng_state(), r0) and tr.step_count == 0 and tr.g_opt._t == 0 and tr.d_opt._t == 0
    after = {k: v for net in (G, D) for k, v in net.state_dict().items()}
    assert all(torch.equal(before[k], after[k]) for k in before)
    assert G.arena().sites_ready and D.arena().sites_ready and int((G.arena().site_exp != 0).sum()) > 50


def test_blocks_teacher_forced_fp8(monkeypatch):
    """Per-Block error of the fp8 operand path without the cascade (the bf16 version is test_parity_gpu.py's
    test_blocks_teacher_forced_bf16): each Xception block of the generator gets the fp32 oracle's input and its output
    is compared with the oracle's.  e4m3 carries 3 mantissa bits (relative step 2^-4 .. 2^-3, rms ~ 2.5 % per operand);
    a block chains 3-4 pointwise GEMMs on such operands.  Bounds = 2 x the largest value measured (printed with -s)."""
    monkeypatch.setattr(ops, "_FP8_MIN_WORK", 0)      # every eligible layer on fp8 operands, the middle flow included
    c, h, w, n = 16, 64, 96, 2
    spec = orc.generator_spec(c, c, 0, "batch")
    G = dxg.Generator(c, c, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d, compute_dtype=FP8)
    G.load_state_dict(orc.fill_state(spec, 7))
    G.to(DEV).train()
    P = orc.fill_state(spec, 7)
    x, y0 = orc.synthetic_fields(n, c, h, w, 3)
    losses.L1Loss()(G(x.to(DEV)), y0.to(DEV)).backward()       # calibration pass: every site sees its tensors
    G.arena().roll_fp8()
    ctx = orc.NormCtx("batch", True, update_stats=False)
    pre = "model.xception_features."
    xf = G.model.xception_features
    worst = 0.0
    BF16 = torch.bfloat16
    with torch.no_grad():
        t = orc.lrelu(orc.norm(P, pre + "bn1", F.conv2d(x, P[pre + "conv1.weight"], None, 2, 1), ctx))
        t = orc.lrelu(orc.norm(P, pre + "bn2", F.conv2d(t, P[pre + "conv2.weight"], None, 1, 1), ctx))
        for cfg in orc.xception_block_table(16):
            yb = orc.block(P, pre + cfg["name"] + ".", cfg, t, ctx)
            xi = ops.ToInternal.apply(t.to(DEV), pad_to(t.shape[1], vec_of(BF16)), BF16)
            got = ops.FromInternal.apply(getattr(xf, cfg["name"])(xi), cfg["cout"]).float().cpu()
            e = rms_err(got, yb)
            print(f"{cfg['name']:8s} {tuple(yb.shape)} fp8 rms-rel {e:.2e}")
            worst = max(worst, e)
            t = yb
    print(f"worst over blocks: rms-rel {worst:.2e}")
    assert worst <= 1.31e-1, worst      # measured 6.54e-2 (block20; the middle flow 2.1e-2 .. 6.2e-2)


@pytest.mark.parametrize("mode", ["ModifiedMinMax", "Wasserstein"])
def test_fp8_first_iteration_vs_oracle(mode, monkeypatch):
    """One whole loop iteration (both Adam updates) of the fp8 path after calibrate_fp8 against oracle.GANStep (fp32, CPU)
    on the same seeded inputs and labels, next to the bf16 path's distance from the same oracle.  The randomly filled
    140-layer nets amplify storage rounding (bf16 end-to-end: ~1e-1); fp8 operands add to it."""
    monkeypatch.setenv("BGAMD_STEP_GRAPH", "0")
    monkeypatch.setattr(ops, "_FP8_MIN_WORK", 0)
    c, h, w, n = 4, 64, 64, 2
    x, y = orc.synthetic_fields(n, c, h, w, 1000)
    res = {}
    for dtype in (torch.bfloat16, FP8):
        G, D, gspec, dspec = build(c, h, w, dtype)
        tr, crit = trainer(G, D, n, mode)
        tr.calibrate_fp8(x.to(DEV), y.to(DEV))
        torch.manual_seed(333)
        labels = crit.draw_labels() if mode == "ModifiedMinMax" else None
        eta = torch.rand(n, 1, 1, 1) if mode == "Wasserstein" else None
        d_loss, g_loss = tr.step(x.to(DEV), y.to(DEV), labels=labels, eta=eta)
        res[dtype] = (float(d_loss), float(g_loss))
    st = orc.GANStep(orc.fill_state(gspec, 21), orc.fill_state(dspec, 22), orc.trainable_keys(gspec), orc.trainable_keys(dspec),
                     "batch", mode, w_gp=10.0)
    d_ref, g_ref = st.step(x, y, labels=labels, eta=eta)
    for dtype, (d, g) in res.items():
        print(f"{mode} {dtype}: d_loss {d:.5f} (oracle {d_ref:.5f}, rel {abs(d - d_ref) / abs(d_ref):.2e})  g_loss {g:.5f} "
              f"(oracle {g_ref:.5f}, rel {abs(g - g_ref) / abs(g_ref):.2e})")
    d, g = res[FP8]
    assert np.isfinite(d) and np.isfinite(g)
    # measured: d_loss 5.7e-2 / 3.9e-2, g_loss 7.1e-2 / 7.8e-2 (bf16 path on the same inputs: 3.0e-2 / 2.6e-2, 1.7e-2 / 1.4e-2)
    assert abs(d - d_ref) <= 1.2e-1 * abs(d_ref) and abs(g - g_ref) <= 1.6e-1 * abs(g_ref)


def test_fp8_whole_step_graph(monkeypatch):
    """The captured step (third call onwards) with fp8 operands: quantisers, fp8 GEMMs and the exponent roll are graph
    nodes, the exponents live in device memory.  With the learning rate at 0 a replayed step must equal the eager one."""
    c, h, w, n = 4, 64, 64, 2

    def run(flag):
        monkeypatch.setenv("BGAMD_STEP_GRAPH", flag)
        G, D, _, _ = build(c, h, w, FP8, seeds=(41, 42))
        crit = losses.GANLoss("ModifiedMinMax", n, torch.device(DEV))
        tr = GANTrainer(G, D, ph.get_optimizer(G.parameters(), "Adam", 0.0, 1e-8, 0.0), ph.get_optimizer(D.parameters(), "Adam", 0.0, 1e-8, 0.0),
                        crit, losses.L1Loss())
        out = []
        for s_ in range(5):
            torch.manual_seed(300 + s_)
            x, y = (t.to(DEV) for t in orc.synthetic_fields(n, c, h, w, 700 + s_))
            d_loss, g_loss = tr.step(x, y)
            out.append((float(d_loss), float(g_loss)))
        return out, len(getattr(tr, "_graphs", {})), G.arena().site_exp.cpu()

    (e, ge, xe), (g, gg, xg) = run("0"), run("1")
    assert ge == 0 and gg == 1
    assert torch.equal(xe, xg), "the exponent history of the replayed run differs from the eager one"
    for i, (a, b) in enumerate(zip(e, g)):
        print(f"step {i}: eager d {a[0]:.6f} g {a[1]:.6f} | graph d {b[0]:.6f} g {b[1]:.6f}")
        assert abs(a[0] - b[0]) <= 2e-5 * abs(a[0]) + 1e-6 and abs(a[1] - b[1]) <= 2e-5 * abs(a[1]), (i, a, b)


def test_fp8_sites_become_ready_one_by_one(monkeypatch):
    """A quantisation site runs its GEMM on fp8 operands only after a roll has given it an exponent from data it saw
    (runtime.Arena.site_ready): forward-only passes calibrate the forward sites; the first backward pass after that still
    runs its data gradients on bf16 operands (an uncalibrated e5m2 site with exponent 0 would flush 1e-6-sized
    gradients to zero), the next one on fp8."""
    monkeypatch.setattr(ops, "_FP8_MIN_WORK", 0)
    c, h, w, n = 4, 64, 64, 2
    G, _, _, _ = build(c, h, w, FP8)
    x, y = (t.to(DEV) for t in orc.synthetic_fields(n, c, h, w, 1000))

    def run(backward):
        L.PROFILE = []
        out = G(x)
        if backward:
            losses.L1Loss()(out, y).backward()
        torch.cuda.synchronize()
        names = [p[0] for p in L.PROFILE]
        L.PROFILE = None
        G.arena().roll_fp8()
        return names.count("bg_conv2d_fwd_fp8"), names.count("bg_conv2d_bwd_data_fp8"), names.count("bg_conv2d_bwd_data")

    a = run(False)      # calibration of the forward sites
    b = run(True)       # forward on fp8; backward sites seen for the first time: bf16 data gradients
    c_ = run(True)      # everything on fp8
    print(a, b, c_)
    assert a[0] == 0 and b[0] > 50 and b[1] == 0 and b[2] > 50 and c_[0] == b[0] and c_[1] > 50 and c_[2] < b[2]


@pytest.mark.parametrize("c", [728, 64, 40])
@pytest.mark.parametrize("with_res", [False, True])
def test_norm_bwd_apply_writes_the_e5m2_copy(c, with_res):
    """bg_norm_act_bwd_apply_stats_q8: dx identical to bg_norm_act_bwd_apply_stats', its e5m2 copy BIT-EXACT against
    torch's cast of the stored bf16 dx scaled by the site's exponent (pad lanes zero), the site's max |dx| recorded."""
    n, h, w, groups, act = 2, 9, 11, 1, 1
    rows = n * h * w
    g_ = torch.Generator().manual_seed(c)
    ld = up(c, 32)
    dt = torch.bfloat16
    x = (torch.randn(rows, ld, generator=g_) * 2 + 0.3).to(dt).to(DEV)
    gy = (torch.randn(rows, ld, generator=g_) * 1e-3).to(dt).to(DEV)
    y = (torch.randn(rows, ld, generator=g_)).to(dt).to(DEV) if with_res else None      # stored output: the activation's branch
    gamma, beta = (torch.rand(c, generator=g_) + 0.5).to(DEV), torch.randn(c, generator=g_).to(DEV)
    s, ss = torch.zeros(groups, c, dtype=torch.float64, device=DEV), torch.zeros(groups, c, dtype=torch.float64, device=DEV)
    L.call("bg_norm_stats", L.BF16, x.data_ptr(), rows, c, ld, groups, s.data_ptr(), ss.data_ptr())
    mean, rstd, scale, shift = (torch.zeros(groups, c, device=DEV) for _ in range(4))
    L.call("bg_norm_finalize_affine", s.data_ptr(), ss.data_ptr(), rows // groups, groups, c, gamma.data_ptr(), beta.data_ptr(),
           1e-5, 0.1, None, None, mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
    s1, s2 = torch.zeros(groups, c, dtype=torch.float64, device=DEV), torch.zeros(groups, c, dtype=torch.float64, device=DEV)
    L.call("bg_norm_act_bwd_reduce", L.BF16, gy.data_ptr(), ld, L.ptr(y), ld, x.data_ptr(), ld, mean.data_ptr(), rstd.data_ptr(),
           gamma.data_ptr(), beta.data_ptr(), rows, c, groups, act, s1.data_ptr(), s2.data_ptr())
    outs = []
    cq, ldq = up(c, 16), up(up(c, 16), 64)
    e_val = 12
    for q8 in (False, True):
        dx = torch.full((rows, ld), 5.0, dtype=dt, device=DEV)
        dres = torch.full((rows, ld), 5.0, dtype=dt, device=DEV) if with_res else None
        dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        args = [L.BF16, gy.data_ptr(), ld, L.ptr(y), ld, x.data_ptr(), ld, s1.data_ptr(), s2.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                mean.data_ptr(), rstd.data_ptr(), 1, dg.data_ptr(), db.data_ptr(), dx.data_ptr(), ld, L.ptr(dres), ld, rows, c, groups, act]
        if q8:
            dxq = torch.full((rows, ldq), 0x55, dtype=torch.uint8, device=DEV)
            ex = torch.tensor([e_val], dtype=torch.int32, device=DEV)
            am = torch.zeros(1, dtype=torch.int32, device=DEV)
            L.call("bg_norm_act_bwd_apply_stats_q8", *args, dxq.data_ptr(), ldq, ex.data_ptr(), am.data_ptr())
        else:
            L.call("bg_norm_act_bwd_apply_stats", *args)
        torch.cuda.synchronize()
        outs.append((dx.clone(), None if dres is None else dres.clone(), dg.clone(), db.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])
    if with_res:
        assert torch.equal(outs[0][1], outs[1][1])
    dxs = outs[1][0][:, :c]
    ref = torch_quant(dxs.cpu(), e_val, L.FP8_E5M2)
    got = dxq.cpu()
    assert torch.equal(got[:, :c], ref), f"{(got[:, :c] != ref).sum().item()} bytes differ"
    assert (got[:, c:cq] == 0).all() and (got[:, cq:] == 0x55).all()
    assert am.view(torch.float32).item() == dxs.float().abs().max().item()
    assert (outs[1][0][:, c:].float() == 5.0).all()


def test_fp8_backward_uses_the_producer_written_copy(monkeypatch):
    """With the sites calibrated, the BatchNorm backward writes the e5m2 copy of the gradient its pointwise convolution's
    data-gradient GEMM reads (bg_norm_act_bwd_apply_stats_q8): those layers need no bg_quant_fp8 pass, including the
    narrow ones the default layer rule would otherwise leave on bf16 operands."""
    monkeypatch.setenv("BGAMD_STEP_GRAPH", "0")
    c, h, w, n = 4, 64, 64, 2
    G, D, _, _ = build(c, h, w, FP8)
    tr, _ = trainer(G, D, n)
    x, y = (t.to(DEV) for t in orc.synthetic_fields(n, c, h, w, 1000))
    counts = []
    for _ in range(3):
        L.PROFILE = []
        tr.step(x, y)
        torch.cuda.synchronize()
        names = [p[0] for p in L.PROFILE]
        L.PROFILE = None
        counts.append({k: names.count(k) for k in ("bg_norm_act_bwd_apply_stats_q8", "bg_norm_act_bwd_apply_stats", "bg_quant_fp8",
                                                    "bg_conv2d_bwd_data_fp8", "bg_conv2d_bwd_data")})
    print(counts)
    assert counts[0]["bg_norm_act_bwd_apply_stats_q8"] == 0                      # calibration step
    assert counts[2]["bg_norm_act_bwd_apply_stats_q8"] > 100 and counts[2]["bg_conv2d_bwd_data_fp8"] > 100
    assert counts[2]["bg_quant_fp8"] < counts[0]["bg_quant_fp8"]
    # every site the fused producer feeds was calibrated from data (gradients of 1e-3 .. 1e-7 need large exponents; an
    # exponent left at 0 would flush them to zero in e5m2)
    for net in (G, D):
        a = net.arena()
        ready_grad = sorted(i for i in a._site_ready if i & 1)
        ex = a.site_exp.cpu()
        assert len(ready_grad) > 50 and all(int(ex[i]) >= 5 for i in ready_grad), [(i, int(ex[i])) for i in ready_grad if int(ex[i]) < 5]
    d_loss, g_loss = tr.step(x, y)
    assert np.isfinite(float(d_loss)) and np.isfinite(float(g_loss))
