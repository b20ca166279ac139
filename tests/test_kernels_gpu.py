"""Every C-ABI entry point of libbgamd.so against a plain PyTorch fp32 CPU
evaluation of the same op on the same (seeded) inputs.  Runs on the MI355X.

Tolerances (stated per test): BG_F32 kernels <= 1e-4 of the output's max
magnitude (different reduction order only); BG_BF16 kernels are fed inputs that
are already bf16-representable, so the only differences are fp32 accumulation
order and the final bf16 rounding of the output (2^-8 relative): 1e-2 of max.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import bias_gan_amd  # noqa: E402
from bias_gan_amd import _lib as L  # noqa: E402
from bias_gan_amd import ops  # noqa: E402

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return 1e-4 if dtype == torch.float32 else 1e-2


def rnd(shape, seed, dtype, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(shape, generator=g) * scale
    return t.to(dtype).to(torch.float32)  # representable in `dtype`


def up(c, dtype):
    v = 8 if dtype == torch.bfloat16 else 4
    return (c + v - 1) // v * v


def to_nhwc(x, dtype, ld=None, coff=0):
    """NCHW fp32 cpu -> NHWC `dtype` cuda buffer [N,H,W,ld]; returns (buffer, view of the C-slice)."""
    n, c, h, w = x.shape
    cp = up(c, dtype)
    ld = ld or cp
    buf = torch.zeros(n, h, w, ld, dtype=dtype, device=DEV)
    buf[..., coff:coff + c] = x.permute(0, 2, 3, 1).to(dtype).to(DEV)
    return buf, buf[..., coff:]


def from_nhwc(view, c):
    return view[..., :c].permute(0, 3, 1, 2).float().cpu()


def krsc(w, dtype):
    """Dense [Coutp][KH][KW][Cinp] copy (the master layout bg_pack_conv_weights reads)."""
    co, ci, kh, kw = w.shape
    out = torch.zeros(up(co, dtype), kh, kw, up(ci, dtype), dtype=dtype, device=DEV)
    out[:co, :, :, :ci] = w.permute(0, 2, 3, 1).to(dtype).to(DEV)
    return out


def pack(wk, dtype):
    """(K-padded KRSC, K-padded CRSK) operand copies through bg_pack_conv_weights."""
    kp, kh, kw, cp = wk.shape
    g = L.kpad(dtype)
    cpp, kpp = (cp + g - 1) // g * g, (kp + g - 1) // g * g
    dk = torch.full((kp * kh * kw * cpp,), 9.0, dtype=dtype, device=DEV)
    dtt = torch.full((cp * kh * kw * kpp,), 9.0, dtype=dtype, device=DEV)
    tbl = torch.tensor([[0, 0, 0, kp, kh * kw, cp, cpp, kpp]], dtype=torch.int64, device=DEV)
    L.call("bg_pack_conv_weights", L.dt(dtype), wk.data_ptr(), dk.data_ptr(), dtt.data_ptr(), tbl.data_ptr(), 1,
           dk.numel() + dtt.numel())
    ref_k = torch.zeros(kp, kh * kw, cpp, dtype=dtype, device=DEV)
    ref_k[:, :, :cp] = wk.view(kp, kh * kw, cp)
    ref_t = torch.zeros(cp, kh * kw, kpp, dtype=dtype, device=DEV)
    ref_t[:, :, :kp] = wk.view(kp, kh * kw, cp).permute(2, 1, 0)
    assert torch.equal(dk.view_as(ref_k), ref_k) and torch.equal(dtt.view_as(ref_t), ref_t)
    return dk, dtt


def assert_close(got, ref, rel, what=""):
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e} > {rel})"


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, pad, dil
    (2, 9, 7, 24, 40, 1, 1, 0, 1),
    (2, 24, 20, 728, 728, 1, 1, 0, 1),
    (2, 20, 18, 16, 128, 3, 1, 1, 1),
    (2, 21, 19, 8, 128, 3, 2, 1, 1),
    (1, 13, 11, 64, 32, 3, 1, 6, 6),
    (2, 11, 9, 128, 256, 1, 2, 0, 1),
    (1, 16, 16, 304, 256, 3, 1, 1, 1),
    (3, 5, 6, 2048, 256, 3, 1, 2, 2),
    (2, 8, 8, 4, 4, 1, 1, 0, 1),
    (2, 40, 70, 16, 128, 3, 2, 1, 1),     # first-layer shape class: the small-Cin stride-2 data-gradient kernel (bf16)
    (1, 37, 35, 12, 64, 3, 2, 1, 1),
]


def conv_desc(dtype, n, h, w, cin, cout, k, s, p, d, ldx, ldy):
    ho = (h + 2 * p - d * (k - 1) - 1) // s + 1
    wo = (w + 2 * p - d * (k - 1) - 1) // s + 1
    return L.ConvDesc(L.dt(dtype), n, h, w, up(cin, dtype), ho, wo, up(cout, dtype), k, k, s, p, d, ldx, ldy), ho, wo


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("sliced", [False, True])
def test_conv2d_fwd_bwd(case, dtype, sliced):
    n, h, w, cin, cout, k, s, p, d = case
    x = rnd((n, cin, h, w), 1, dtype)
    wt = rnd((cout, cin, k, k), 2, dtype, 1.0 / math.sqrt(cin * k * k))
    bias = rnd((cout,), 3, torch.float32)
    cinp, coutp = up(cin, dtype), up(cout, dtype)
    ldx = cinp + (16 if sliced else 0)
    ldy = coutp + (24 if sliced else 0)
    xoff, yoff = (8 if sliced else 0), (16 if sliced else 0)
    desc, ho, wo = conv_desc(dtype, n, h, w, cin, cout, k, s, p, d, ldx, ldy)
    xb, xv = to_nhwc(x, dtype, ldx, xoff)
    wk = krsc(wt, dtype)
    wpk, wpt = pack(wk, dtype)
    bpad = torch.zeros(coutp, device=DEV)
    bpad[:cout] = bias.to(DEV)
    yb = torch.full((n, ho, wo, ldy), 7.0, dtype=dtype, device=DEV)
    yv = yb[..., yoff:]
    L.call("bg_conv2d_fwd", desc, xv.data_ptr(), wpk.data_ptr(), bpad.data_ptr(), yv.data_ptr())
    ref = F.conv2d(x, wt, bias, s, p, d)
    assert_close(from_nhwc(yv, cout), ref, tol(dtype), "fwd")
    if sliced:  # nothing outside the slice was touched
        assert (yb[..., :yoff] == 7.0).all() and (yb[..., yoff + coutp:] == 7.0).all()
    if coutp > cout:
        assert (yv[..., cout:coutp] == 0).all()

    # backward
    go = rnd((n, cout, ho, wo), 4, dtype)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, s, p, d).backward(go)
    gb, gv = to_nhwc(go, dtype, ldy, yoff)
    dxb = torch.full((n, h, w, ldx), 5.0, dtype=dtype, device=DEV)
    dxv = dxb[..., xoff:]
    L.call("bg_conv2d_bwd_data", desc, gv.data_ptr(), wpt.data_ptr(), dxv.data_ptr())
    assert_close(from_nhwc(dxv, cin), xr.grad, tol(dtype), "bwd_data")
    dw = torch.zeros(coutp, k, k, cinp, device=DEV)
    db = torch.zeros(coutp, device=DEV)
    L.call("bg_conv2d_bwd_weight", desc, xv.data_ptr(), gv.data_ptr(), dw.data_ptr(), db.data_ptr())
    got_dw = dw[:cout, :, :, :cin].permute(0, 3, 1, 2).cpu()
    assert_close(got_dw, wr.grad, 1e-4 if dtype == torch.float32 else 2e-3, "bwd_weight")
    assert_close(db[:cout].cpu(), go.sum((0, 2, 3)), 1e-4 if dtype == torch.float32 else 2e-3, "dbias")
    # accumulation semantics: a second call adds
    L.call("bg_conv2d_bwd_weight", desc, xv.data_ptr(), gv.data_ptr(), dw.data_ptr(), None)
    assert_close(dw[:cout, :, :, :cin].permute(0, 3, 1, 2).cpu(), 2 * wr.grad, 2e-3, "bwd_weight accumulate")


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_mfma_layout_asymmetric(dtype):
    """A = identity-like weights with an asymmetric input: catches a swapped
    row/column map of the MFMA accumulator (cdna guide section 3)."""
    n, h, w, c = 1, 16, 16, 128
    x = torch.arange(n * c * h * w, dtype=torch.float32).reshape(n, c, h, w) % 251
    wt = torch.zeros(c, c, 1, 1)
    for i in range(c):
        wt[i, (i * 7 + 3) % c, 0, 0] = 1.0  # permutation, not symmetric
    desc, ho, wo = conv_desc(dtype, n, h, w, c, c, 1, 1, 0, 1, c, c)
    xb, xv = to_nhwc(x, dtype)
    yb = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
    wpk, _ = pack(krsc(wt, dtype), dtype)
    L.call("bg_conv2d_fwd", desc, xv.data_ptr(), wpk.data_ptr(), None, yb.data_ptr())
    assert torch.equal(from_nhwc(yb, c), F.conv2d(x, wt))  # small integers: exact in bf16 too


DW_CASES = [(2, 13, 18, 16, 1, 1), (2, 12, 10, 24, 2, 1), (1, 9, 7, 728, 1, 1), (2, 9, 7, 40, 1, 2), (1, 7, 5, 8, 2, 1),
            (2, 16, 12, 24, 1, 2), (2, 11, 13, 40, 2, 1), (1, 20, 18, 264, 2, 1),
            # the tiny maps at the bottom of a 64 x 64 discriminator: one- and two-row bands, fewer rows than taps
            (4, 4, 4, 728, 1, 1), (4, 4, 4, 728, 1, 2), (2, 2, 3, 24, 1, 1), (3, 1, 5, 16, 1, 1), (2, 3, 2, 16, 1, 2)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", DW_CASES)
def test_dwconv(case, dtype):
    n, h, w, c, s, d = case
    x = rnd((n, c, h, w), 5, dtype)
    wt = rnd((c, 1, 3, 3), 6, dtype, 0.3)
    ho, wo = -(-h // s), -(-w // s)
    ld = up(c, dtype) + 8
    xb, xv = to_nhwc(x, dtype, ld, 8)
    wk = torch.zeros(3, 3, up(c, dtype), dtype=dtype, device=DEV)
    wk[:, :, :c] = wt[:, 0].permute(1, 2, 0).to(dtype).to(DEV)
    desc = L.DwDesc(L.dt(dtype), n, h, w, up(c, dtype), ho, wo, s, d, ld, ld)
    yb = torch.zeros(n, ho, wo, ld, dtype=dtype, device=DEV)
    yv = yb[..., 8:]
    L.call("bg_dwconv3x3_fwd", desc, xv.data_ptr(), wk.data_ptr(), yv.data_ptr())
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    ref = F.conv2d(F.pad(xr, (d, d, d, d)), wr, None, s, 0, d, groups=c)  # fixed_padding for k=3: (d, d)
    assert ref.shape[2:] == (ho, wo)
    assert_close(from_nhwc(yv, c), ref.detach(), tol(dtype), "dw fwd")
    go = rnd(tuple(ref.shape), 7, dtype)
    ref.backward(go)
    gb, gv = to_nhwc(go, dtype, ld, 8)
    dxb = torch.zeros(n, h, w, ld, dtype=dtype, device=DEV)
    L.call("bg_dwconv3x3_bwd_data", desc, gv.data_ptr(), wk.data_ptr(), dxb[..., 8:].data_ptr())
    assert_close(from_nhwc(dxb[..., 8:], c), xr.grad, tol(dtype), "dw bwd_data")
    dw = torch.zeros(3, 3, up(c, dtype), device=DEV)
    L.call("bg_dwconv3x3_bwd_weight", desc, xv.data_ptr(), gv.data_ptr(), dw.data_ptr())
    assert_close(dw[:, :, :c].permute(2, 0, 1).cpu(), wr.grad[:, 0], 1e-4 if dtype == torch.float32 else 2e-3, "dw wgrad")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["batch", "instance", "eval", "identity"])
@pytest.mark.parametrize("act,with_res", [(1, False), (0, True), (1, True)])
def test_norm_act(dtype, mode, act, with_res):
    n, h, w, c = 3, 7, 9, 24
    x = (rnd((n, c, h, w), 8, dtype, 2.0) + 0.5).to(dtype).float()  # keep it representable in `dtype`
    res = rnd((n, c, h, w), 9, dtype)
    gamma = torch.rand(c, generator=torch.Generator().manual_seed(1)) + 0.5
    beta = torch.randn(c, generator=torch.Generator().manual_seed(2)) * 0.2
    rmean0 = torch.randn(c, generator=torch.Generator().manual_seed(3)) * 0.1
    rvar0 = torch.rand(c, generator=torch.Generator().manual_seed(4)) + 0.5
    rows, groups = n * h * w, (n if mode == "instance" else 1)
    xb, xv = to_nhwc(x, dtype)
    rb, rv = to_nhwc(res, dtype)
    yb = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
    dtc = L.dt(dtype)
    f32 = lambda *s: torch.zeros(*s, device=DEV)  # noqa: E731
    f64 = lambda *s: torch.zeros(*s, device=DEV, dtype=torch.float64)  # noqa: E731
    g_d, b_d = gamma.to(DEV), beta.to(DEV)
    rm_d, rv_d = rmean0.to(DEV), rvar0.to(DEV)
    mean, rstd, scale, shift = f32(groups, c), f32(groups, c), f32(groups, c), f32(groups, c)
    # ---- reference
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rmean0.clone(), rvar0.clone()
    if mode == "batch":
        z = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    elif mode == "eval":
        z = F.batch_norm(xr, rm_ref, rv_ref, gr, br, False, 0.1, 1e-5)
    elif mode == "instance":
        z = F.instance_norm(xr, eps=1e-5)
    else:
        z = xr
    if with_res:
        z = z + rr
    yref = F.leaky_relu(z, 0.2) if act else z
    # ---- kernels
    if mode in ("batch", "instance"):
        s, ss = f64(groups, c), f64(groups, c)
        L.call("bg_norm_stats", dtc, xv.data_ptr(), rows, c, c, groups, s.data_ptr(), ss.data_ptr())
        aff = mode == "batch"
        L.call("bg_norm_finalize", s.data_ptr(), ss.data_ptr(), rows // groups, groups, c,
               g_d.data_ptr() if aff else None, b_d.data_ptr() if aff else None, 1e-5, 0.1,
               rm_d.data_ptr() if aff else None, rv_d.data_ptr() if aff else None, mean.data_ptr(), rstd.data_ptr(),
               scale.data_ptr(), shift.data_ptr())
    elif mode == "eval":
        L.call("bg_norm_eval_affine", c, g_d.data_ptr(), b_d.data_ptr(), rm_d.data_ptr(), rv_d.data_ptr(), 1e-5,
               scale.data_ptr(), shift.data_ptr())
        mean.copy_(rm_d), rstd.copy_(1.0 / torch.sqrt(rv_d + 1e-5))
    ident = mode == "identity"
    L.call("bg_norm_act_fwd", dtc, xv.data_ptr(), c, None if ident else scale.data_ptr(),
           None if ident else shift.data_ptr(), rv.data_ptr() if with_res else None, c, yb.data_ptr(), c, rows, c, groups,
           act)
    assert_close(from_nhwc(yb, c), yref.detach(), tol(dtype), "fwd")
    if mode == "batch":
        assert_close(rm_d.cpu(), rm_ref, 1e-5, "running_mean")
        assert_close(rv_d.cpu(), rv_ref, 1e-5, "running_var")
    # ---- backward
    go = rnd(tuple(yref.shape), 10, dtype)
    yref.backward(go)
    gb, gv = to_nhwc(go, dtype)
    # the saved output the kernels see is the (possibly bf16-rounded) kernel output
    A, B, Cc = f32(groups, c), f32(groups, c), f32(groups, c)
    dgamma, dbeta = f32(c), f32(c)
    dxb = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
    dresb = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
    if ident:
        L.call("bg_norm_act_bwd_apply", dtc, gv.data_ptr(), c, yb.data_ptr(), c, None, 0, None, None, None,
               dxb.data_ptr(), c, dresb.data_ptr() if with_res else None, c, rows, c, groups, act)
    else:
        s1, s2 = f64(groups, c), f64(groups, c)
        L.call("bg_norm_act_bwd_reduce", dtc, gv.data_ptr(), c, yb.data_ptr(), c, xv.data_ptr(), c, mean.data_ptr(),
               rstd.data_ptr(), None, None, rows, c, groups, act, s1.data_ptr(), s2.data_ptr())
        aff = mode in ("batch", "eval")
        L.call("bg_norm_bwd_finalize", s1.data_ptr(), s2.data_ptr(), rows // groups, groups, c,
               g_d.data_ptr() if aff else None, mean.data_ptr(), rstd.data_ptr(), 0 if mode == "eval" else 1,
               A.data_ptr(), B.data_ptr(), Cc.data_ptr(), dgamma.data_ptr() if aff else None,
               dbeta.data_ptr() if aff else None)
        L.call("bg_norm_act_bwd_apply", dtc, gv.data_ptr(), c, yb.data_ptr(), c, xv.data_ptr(), c, A.data_ptr(),
               B.data_ptr(), Cc.data_ptr(), dxb.data_ptr(), c, dresb.data_ptr() if with_res else None, c, rows, c, groups,
               act)
        if aff:
            assert_close(dgamma.cpu(), gr.grad, 5e-4 if dtype == torch.float32 else 2e-2, "dgamma")
            assert_close(dbeta.cpu(), br.grad, 5e-4 if dtype == torch.float32 else 2e-2, "dbeta")
    assert_close(from_nhwc(dxb, c), xr.grad, 5e-4 if dtype == torch.float32 else 2e-2, "dx")
    if with_res:
        assert_close(from_nhwc(dresb, c), rr.grad, tol(dtype), "dres")


def test_norm_single_element_instance():
    """InstanceNorm on a 1x1 map (global-pool branch): mean = x, var = 0 -> output 0."""
    n, c = 2, 16
    x = torch.randn(n, 1, 1, c, device=DEV)
    s = torch.zeros(n, c, device=DEV, dtype=torch.float64)
    ss = torch.zeros_like(s)
    L.call("bg_norm_stats", L.F32, x.data_ptr(), n, c, c, n, s.data_ptr(), ss.data_ptr())
    mean, rstd, scale, shift = (torch.zeros(n, c, device=DEV) for _ in range(4))
    L.call("bg_norm_finalize", s.data_ptr(), ss.data_ptr(), 1, n, c, None, None, 1e-5, 0.1, None, None, mean.data_ptr(),
           rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
    y = torch.empty_like(x)
    L.call("bg_norm_act_fwd", L.F32, x.data_ptr(), c, scale.data_ptr(), shift.data_ptr(), None, 0, y.data_ptr(), c, n, c,
           n, 1)
    assert y.abs().max().item() < 1e-3 * x.abs().max().item()


RESIZE_CASES = [(2, 5, 4, 17, 13, 8), (1, 1, 1, 6, 5, 16), (2, 6, 7, 6, 7, 4), (1, 9, 12, 33, 47, 12), (2, 8, 8, 3, 5, 4)]


@pytest.mark.parametrize("dtypes", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                    (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("case", RESIZE_CASES)
def test_resize_bilinear(case, dtypes):
    n, hi, wi, ho, wo, c = case
    din, dout = dtypes
    x = rnd((n, c, hi, wi), 11, din)
    xb = x.permute(0, 2, 3, 1).contiguous().to(din).to(DEV)
    yb = torch.zeros(n, ho, wo, c, dtype=dout, device=DEV)
    L.call("bg_resize_bilinear_fwd", L.dt(din), L.dt(dout), xb.data_ptr(), c, yb.data_ptr(), c, n, hi, wi, ho, wo, c)
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr, size=(ho, wo), mode="bilinear", align_corners=True)
    assert_close(from_nhwc(yb, c), ref.detach(), 1e-5 if dout == torch.float32 else 1e-2, "resize fwd")
    go = rnd((n, c, ho, wo), 12, dout)
    ref.backward(go)
    gb = go.permute(0, 2, 3, 1).contiguous().to(dout).to(DEV)
    dxb = torch.zeros(n, hi, wi, c, dtype=din, device=DEV)
    L.call("bg_resize_bilinear_bwd", L.dt(dout), L.dt(din), gb.data_ptr(), c, dxb.data_ptr(), c, n, hi, wi, ho, wo, c)
    assert_close(from_nhwc(dxb, c), xr.grad, 1e-5 if din == torch.float32 else 1e-2, "resize bwd")


@pytest.mark.parametrize("dtype", DTYPES)
def test_colsum_broadcast_cast_layout(dtype):
    n, h, w, c = 3, 6, 5, 24
    x = rnd((n, c, h, w), 13, dtype)
    xb, xv = to_nhwc(x, dtype)
    out = torch.zeros(n, c, device=DEV)
    L.call("bg_colsum", L.dt(dtype), xv.data_ptr(), c, n * h * w, c, n, 1.0 / (h * w), out.data_ptr())
    assert_close(out.cpu(), x.mean((2, 3)), 1e-5, "colsum")
    yb = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
    L.call("bg_broadcast_rows", L.dt(dtype), out.data_ptr(), 2.0, yb.data_ptr(), c, n * h * w, c, n)
    assert_close(from_nhwc(yb, c), (2.0 * x.mean((2, 3)))[:, :, None, None].expand(n, c, h, w), tol(dtype), "broadcast")
    # cast with strides
    dst = torch.zeros(n * h * w, c + 8, device=DEV)
    L.call("bg_cast_rows", L.dt(dtype), L.F32, xv.data_ptr(), c, dst.data_ptr(), c + 8, n * h * w, c)
    assert torch.equal(dst[:, :c].cpu(), x.permute(0, 2, 3, 1).reshape(-1, c))
    # layout
    xs = rnd((n, 5, h, w), 14, torch.float32).to(DEV)
    nh = torch.full((n, h, w, 8), 3.0, dtype=dtype, device=DEV)
    L.call("bg_nchw_to_nhwc", L.dt(dtype), xs.data_ptr(), nh.data_ptr(), n, 5, h * w, 8, 8)
    assert_close(nh[..., :5].permute(0, 3, 1, 2).float().cpu(), xs.cpu(), tol(dtype), "nchw->nhwc")
    assert (nh[..., 5:] == 0).all()
    back = torch.zeros(n, 5, h, w, device=DEV)
    L.call("bg_nhwc_to_nchw", L.dt(dtype), nh.data_ptr(), 8, back.data_ptr(), n, 5, h * w)
    assert torch.equal(back.cpu(), nh[..., :5].permute(0, 3, 1, 2).float().cpu())
    a = rnd((n * h * w, c), 15, dtype)
    ab, bb = a.to(dtype).to(DEV), xv.reshape(-1, c).clone()
    L.call("bg_axpy_rows", L.dt(dtype), ab.data_ptr(), c, bb.data_ptr(), c, n * h * w, c)
    assert_close(bb.float().cpu(), a + x.permute(0, 2, 3, 1).reshape(-1, c), tol(dtype), "axpy")


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_head(dtype):
    n, h, w, c = 3, 4, 5, 64
    x = rnd((n, c, h, w), 16, dtype)
    wl = rnd((1, c * h * w), 17, torch.float32, 0.05)
    bl = torch.tensor([0.3])
    xb, xv = to_nhwc(x, dtype)
    w_d, b_d = wl.to(DEV), bl.to(DEV)
    logits = torch.zeros(n, 1, device=DEV)
    L.call("bg_linear_head_fwd", L.dt(dtype), xv.data_ptr(), c, w_d.data_ptr(), b_d.data_ptr(), logits.data_ptr(), n,
           h * w, c)
    xr, wr, br = x.clone().requires_grad_(True), wl.clone().requires_grad_(True), bl.clone().requires_grad_(True)
    ref = F.linear(xr.reshape(n, -1), wr, br)
    assert_close(logits.cpu(), ref.detach(), 1e-5, "head fwd")
    dl = torch.tensor([[0.5], [-1.0], [0.25]])
    ref.backward(dl)
    dxb = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
    dw, db = torch.zeros(1, c * h * w, device=DEV), torch.zeros(1, device=DEV)
    dl_d = dl.to(DEV)
    L.call("bg_linear_head_bwd", L.dt(dtype), xv.data_ptr(), c, w_d.data_ptr(), dl_d.data_ptr(), dxb.data_ptr(), c,
           dw.data_ptr(), db.data_ptr(), n, h * w, c)
    assert_close(from_nhwc(dxb, c), xr.grad, tol(dtype), "head dx")
    assert_close(dw.cpu(), wr.grad, 1e-5, "head dw")
    assert_close(db.cpu(), br.grad, 1e-6, "head db")


def test_losses():
    n = 5
    x = torch.randn(n, 1) * 3
    y = torch.rand(n, 1)
    loss, dx = torch.zeros(1, device=DEV), torch.zeros(n, 1, device=DEV)
    xd, yd = x.to(DEV), y.to(DEV)
    L.call("bg_bce_logits", xd.data_ptr(), yd.data_ptr(), n, loss.data_ptr(), dx.data_ptr())
    xr = x.clone().requires_grad_(True)
    ref = F.binary_cross_entropy_with_logits(xr, y)
    ref.backward()
    assert_close(loss.cpu(), ref.detach().reshape(1), 1e-6, "bce")
    assert_close(dx.cpu(), xr.grad, 1e-6, "bce grad")
    p, t, w = torch.randn(2, 3, 9, 7), torch.randn(2, 3, 9, 7), torch.rand(2, 3, 9, 7)
    p[0, 0, 0, 0] = t[0, 0, 0, 0]  # sign(0) = 0
    for wt in (None, w):
        l1 = torch.zeros(1, device=DEV)
        pd, td = p.to(DEV), t.to(DEV)
        wd = None if wt is None else wt.to(DEV)
        L.call("bg_l1_loss_fwd", pd.data_ptr(), td.data_ptr(), L.ptr(wd), p.numel(), 1.0 / p.numel(), l1.data_ptr())
        pr = p.clone().requires_grad_(True)
        ref = ((pr - t).abs() * (1 if wt is None else wt)).mean()
        ref.backward()
        assert_close(l1.cpu(), ref.detach().reshape(1), 1e-5, "l1")
        dp = torch.zeros_like(pd)
        coef = torch.tensor([1.0], device=DEV)
        L.call("bg_l1_loss_bwd", pd.data_ptr(), td.data_ptr(), L.ptr(wd), p.numel(), 1.0 / p.numel(), coef.data_ptr(),
               dp.data_ptr())
        assert_close(dp.cpu(), pr.grad, 1e-6, "l1 grad")
    g = torch.randn(2, 6, 5, 4)
    gp = torch.zeros(1, device=DEV)
    gd = g.to(DEV)
    L.call("bg_gp_penalty", gd.data_ptr(), 2, 6, 20, 1.0 / 40, gp.data_ptr())
    assert_close(gp.cpu(), ((g.norm(2, dim=1) - 1) ** 2).mean().reshape(1), 1e-5, "gp")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,dil", [(2, 40, 36, 128, 128, 3, 1, 1), (3, 33, 20, 24, 40, 1, 1, 1), (2, 64, 48, 16, 128, 3, 2, 1),
                                                         (2, 24, 24, 136, 264, 3, 1, 2), (1, 16, 16, 728, 728, 1, 1, 1)])
def test_conv_wgrad_workspace_form_is_the_atomic_one_and_reproducible(dtype, n, h, w, cin, cout, k, stride, dil):
    """bg_conv2d_bwd_weight_ws (split tiles in a workspace, added in split order) == bg_conv2d_bwd_weight (float atomics) to
    summation order, accumulates into dW like it, and two runs are bit-identical -- taps as tiles, the folded few-channel
    layout (Cin 16, stride 2), channel counts off the tile size, a dilated layer, the 16 x 16 maps."""
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    pad = dil * (k - 1) // 2
    g_ = torch.Generator().manual_seed(5)
    x = torch.randn(n, h, w, cin, generator=g_).to(dtype).to(DEV)
    gy = (torch.randn(n, ho, wo, cout, generator=g_) * 0.1).to(dtype).to(DEV)
    desc = L.ConvDesc(L.dt(dtype), n, h, w, cin, ho, wo, cout, k, k, stride, pad, dil, cin, cout)
    nb = L.wgrad_ws_bytes(desc)
    ws = torch.full((nb,), 0x7f, dtype=torch.uint8, device=DEV)      # stale bytes: every slice the second pass reads must have been written
    base = torch.randn(cout, k, k, cin, generator=g_).to(DEV)
    outs = []
    for form in ("atomic", "ws", "ws"):
        dw = base.clone()
        if form == "atomic":
            L.call("bg_conv2d_bwd_weight", desc, x.data_ptr(), gy.data_ptr(), dw.data_ptr(), None)
        else:
            L.call("bg_conv2d_bwd_weight_ws", desc, x.data_ptr(), gy.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb)
        outs.append(dw)
    torch.cuda.synchronize()
    assert torch.equal(outs[1], outs[2])
    assert_close((outs[1] - base).cpu(), (outs[0] - base).cpu(), 5e-6 if dtype == torch.float32 else 2e-5, "workspace vs atomic")
    with pytest.raises(RuntimeError, match="workspace"):
        L.call("bg_conv2d_bwd_weight_ws", desc, x.data_ptr(), gy.data_ptr(), outs[0].data_ptr(), None, ws.data_ptr(), nb - 16)


def test_adam_matches_torch():
    n = 1000
    p0, g1, g2 = torch.randn(n), torch.randn(n) * 1e-3, torch.randn(n) * 1e-3
    for decoupled in (0, 1):
        pr = p0.clone().requires_grad_(True)
        opt = (torch.optim.AdamW if decoupled else torch.optim.Adam)([pr], lr=1e-3, eps=1e-8, weight_decay=1e-2)
        p, m, v = p0.to(DEV).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        plp = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
        for t, g in enumerate((g1, g2), 1):
            pr.grad = g.clone()
            opt.step()
            gdev = (2 * g).to(DEV)
            L.call("bg_adam_step", p.data_ptr(), gdev.data_ptr(), m.data_ptr(), v.data_ptr(), plp.data_ptr(), n,
                   1e-3, 0.9, 0.999, 1e-8, 1e-2, decoupled, 1 - 0.9 ** t, 1 - 0.999 ** t, 0.5)
        assert_close(p.cpu(), pr.detach(), 1e-6, "adam")
        assert torch.equal(plp.cpu(), p.cpu().to(torch.bfloat16))


def test_adam_with_device_scalars_matches_launch_arguments():
    """bg_adam_step_dev (lr / bias corrections / gradient scale read from four device floats written by bg_set_floats: the
    form a captured training step replays) == bg_adam_step with the same numbers as launch arguments, bit for bit."""
    n = 4099
    g_ = torch.Generator().manual_seed(1)
    p0, gr = torch.randn(n, generator=g_), torch.randn(n, generator=g_) * 1e-2
    for decoupled in (0, 1):
        ref = [t.to(DEV).clone() for t in (p0, torch.zeros(n), torch.zeros(n))]
        dev = [t.to(DEV).clone() for t in (p0, torch.zeros(n), torch.zeros(n))]
        lpa, lpb = (torch.zeros(n, dtype=torch.bfloat16, device=DEV) for _ in range(2))
        hyper = torch.zeros(4, device=DEV)
        gdev = gr.to(DEV)
        for t in (1, 2, 3):
            lr, bc1, bc2, sc = 1e-3 / t, 1 - 0.9 ** t, 1 - 0.999 ** t, 0.5
            L.call("bg_adam_step", ref[0].data_ptr(), gdev.data_ptr(), ref[1].data_ptr(), ref[2].data_ptr(), lpa.data_ptr(), n, lr, 0.9,
                   0.999, 1e-8, 1e-2, decoupled, bc1, bc2, sc)
            L.call("bg_set_floats", hyper.data_ptr(), 4, lr, bc1, bc2, sc)
            L.call("bg_adam_step_dev", dev[0].data_ptr(), gdev.data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), lpb.data_ptr(), n,
                   hyper.data_ptr(), 0.9, 0.999, 1e-8, 1e-2, decoupled)
        for a, b in zip(ref, dev):
            assert torch.equal(a, b)
        assert torch.equal(lpa, lpb)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,cp,ld", [(16, 16, 32), (5, 8, 8), (20, 24, 40), (16, 16, 16), (13, 16, 16), (32, 32, 32), (27, 32, 32), (8, 8, 8)])
def test_layout_four_pixel_form(dtype, c, cp, ld):
    """HW % 4 == 0: the float4 forms of bg_nchw_to_nhwc / bg_nhwc_to_nchw (pad lanes zeroed, the row tail beyond Cp
    untouched, exact round trip); dense bf16 rows of 8 / 16 / 32 channels take the LDS-staged form (1 024-pixel tiles: the
    12 x 172 image is two tiles and a tail of 16 pixels)."""
    n, h, w = 3, 12, 43 * 4
    xs = rnd((n, c, h, w), 41, torch.float32).to(DEV)
    nh = torch.full((n, h, w, ld), 3.0, dtype=dtype, device=DEV)
    L.call("bg_nchw_to_nhwc", L.dt(dtype), xs.data_ptr(), nh.data_ptr(), n, c, h * w, cp, ld)
    assert torch.equal(nh[..., :c].permute(0, 3, 1, 2).float(), xs.to(dtype).float())
    assert (nh[..., c:cp] == 0).all() and (nh[..., cp:] == 3.0).all()
    back = torch.zeros(n, c, h, w, device=DEV)
    L.call("bg_nhwc_to_nchw", L.dt(dtype), nh.data_ptr(), ld, back.data_ptr(), n, c, h * w)
    assert torch.equal(back, nh[..., :c].permute(0, 3, 1, 2).float())


def test_bad_arguments_raise():
    d = L.ConvDesc(L.BF16, 1, 8, 8, 12, 8, 8, 16, 1, 1, 1, 0, 1, 12, 16)  # Cin not a multiple of 8
    z = torch.zeros(8, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        L.call("bg_conv2d_fwd", d, z.data_ptr(), z.data_ptr(), None, z.data_ptr())
    d = L.ConvDesc(L.BF16, 1, 8, 8, 16, 9, 8, 16, 1, 1, 1, 0, 1, 16, 16)  # wrong Ho
    with pytest.raises(RuntimeError, match="conv arithmetic"):
        L.call("bg_conv2d_fwd", d, z.data_ptr(), z.data_ptr(), None, z.data_ptr())


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_fwd_fused_statistics(dtype):
    """bg_conv2d_fwd_stats: per-channel sum / sum of squares of the STORED outputs from the epilogue."""
    n, h, w, cin, cout, k = 3, 13, 11, 24, 200, 3
    x = rnd((n, cin, h, w), 21, dtype)
    wt = rnd((cout, cin, k, k), 22, dtype, 1.0 / math.sqrt(cin * k * k))
    desc, ho, wo = conv_desc(dtype, n, h, w, cin, cout, k, 1, 1, 1, up(cin, dtype), up(cout, dtype))
    xb, xv = to_nhwc(x, dtype)
    wpk, _ = pack(krsc(wt, dtype), dtype)
    yb = torch.zeros(n, ho, wo, up(cout, dtype), dtype=dtype, device=DEV)
    st = torch.zeros(2, 1, up(cout, dtype), dtype=torch.float64, device=DEV)
    L.call("bg_conv2d_fwd_stats", desc, xv.data_ptr(), wpk.data_ptr(), yb.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 1)
    y = yb.double()
    assert_close(from_nhwc(yb, cout), F.conv2d(x, wt, None, 1, 1, 1), tol(dtype), "fwd")
    assert_close(st[0, 0].cpu(), y.sum((0, 1, 2)).cpu(), 1e-5, "sum")
    assert_close(st[1, 0].cpu(), (y * y).sum((0, 1, 2)).cpu(), 1e-5, "sumsq")
    # a statistic group must own whole 128-pixel tiles
    with pytest.raises(RuntimeError, match="statistic group"):
        L.call("bg_conv2d_fwd_stats", desc, xv.data_ptr(), wpk.data_ptr(), yb.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_wide_tile_launch(dtype):
    """A launch the round-count model sends to the 256 x 256 tile (2 x 1024 tiles of 256 x 128 against 4 rounds of
    256 x 256, 16 K-steps): outputs, per-group statistics from the 4-pixel-wave epilogue, rows beyond M and the
    partial last channel tile."""
    n, h, w, cin, cout = 2, 256, 255, 512, 264          # M = 130 560: the last pixel tile is partial
    x = rnd((n, cin, h, w), 31, dtype)
    wt = rnd((cout, cin, 1, 1), 32, dtype, 1.0 / math.sqrt(cin))
    desc, ho, wo = conv_desc(dtype, n, h, w, cin, cout, 1, 1, 0, 1, up(cin, dtype), up(cout, dtype))
    xb, xv = to_nhwc(x, dtype)
    wpk, _ = pack(krsc(wt, dtype), dtype)
    yb = torch.zeros(n, ho, wo, up(cout, dtype), dtype=dtype, device=DEV)
    st = torch.zeros(2, 1, up(cout, dtype), dtype=torch.float64, device=DEV)
    L.call("bg_conv2d_fwd_stats", desc, xv.data_ptr(), wpk.data_ptr(), yb.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 1)
    ref = F.conv2d(x, wt)
    assert_close(from_nhwc(yb, cout), ref, tol(dtype), "fwd")
    y = yb.double()
    assert_close(st[0, 0].cpu(), y.sum((0, 1, 2)).cpu(), 1e-5, "sum")
    assert_close(st[1, 0].cpu(), (y * y).sum((0, 1, 2)).cpu(), 1e-5, "sumsq")
    # two statistic groups of 65 536 pixels (whole 256-pixel tiles)
    x2, desc2 = x[:, :, :, :128].contiguous(), conv_desc(dtype, n, h, 128, cin, cout, 1, 1, 0, 1, up(cin, dtype), up(cout, dtype))[0]
    xb2, xv2 = to_nhwc(torch.cat([x2, x2 + 1.0], 0), dtype)
    desc2 = conv_desc(dtype, 2 * n, h, 128, cin, cout, 1, 1, 0, 1, up(cin, dtype), up(cout, dtype))[0]
    yb2 = torch.zeros(2 * n, h, 128, up(cout, dtype), dtype=dtype, device=DEV)
    st2 = torch.zeros(2, 2, up(cout, dtype), dtype=torch.float64, device=DEV)
    L.call("bg_conv2d_fwd_stats", desc2, xv2.data_ptr(), wpk.data_ptr(), yb2.data_ptr(), st2[0].data_ptr(), st2[1].data_ptr(), 2)
    y2 = yb2.double()
    for g in range(2):
        yg = y2[n * g:n * g + n]
        assert_close(st2[0, g].cpu(), yg.sum((0, 1, 2)).cpu(), 1e-5, f"sum[{g}]")
        assert_close(st2[1, g].cpu(), (yg * yg).sum((0, 1, 2)).cpu(), 1e-5, f"sumsq[{g}]")
    assert_close(from_nhwc(yb2[:n], cout), F.conv2d(x2, wt), tol(dtype), "fwd, group 0")


FAT_CASES = [
    # n, h, w, cin, cout, k, stride, pad, dil, groups
    (2, 24, 20, 728, 728, 1, 1, 0, 1, 1),      # 384-row tiles, two channel tiles (the second partial), tiny pixel tiles
    (2, 9, 7, 24, 136, 1, 1, 0, 1, 1),         # one-pixel tiles
    (4, 16, 16, 40, 264, 1, 1, 0, 1, 2),       # statistic groups
    (2, 11, 9, 128, 256, 1, 2, 0, 1, 1),       # strided 1x1 (the Block skip) and its lattice data gradient
    (1, 16, 16, 304, 256, 3, 1, 1, 1, 1),      # 3x3: the tap-walking variant (256-row tiles)
    (3, 5, 6, 520, 256, 3, 1, 2, 2, 1),        # dilated 3x3, mostly padding taps
    (2, 21, 19, 72, 200, 3, 2, 1, 1, 1),       # strided 3x3 + strided data gradient
    (6, 72, 48, 728, 728, 1, 1, 0, 1, 2),      # the workload's tile: 2 x 96 tiles of 216 pixels, two groups
    (1, 150, 120, 64, 1032, 1, 1, 0, 1, 1),    # several rounds of 256-row tiles
    (2, 130, 120, 24, 256, 3, 1, 1, 1, 1),     # 256 x 224 tiles with taps (one round against two of 256 x 112)
    (2, 33, 31, 128, 128, 1, 1, 0, 1, 1),      # 128-row tiles (the configuration of activations beyond 2 GiB)
    (2, 20, 18, 16, 128, 3, 1, 1, 1, 1),
    (2, 21, 19, 24, 72, 3, 2, 1, 1, 1),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", FAT_CASES)
def test_conv_fat_tile_variant(case, dtype):
    """The fat-tile GEMM kernel (one 384/256-row tile per CU, run-time pixel width) forced onto every legal launch:
    forward with the statistics epilogue and the data gradient against torch, and against the classic tiles."""
    n, h, w, cin, cout, k, s, p, d, groups = case
    x = rnd((n, cin, h, w), 41, dtype)
    wt = rnd((cout, cin, k, k), 42, dtype, 1.0 / math.sqrt(cin * k * k))
    cinp, coutp = up(cin, dtype), up(cout, dtype)
    ldx, ldy = cinp + 16, coutp + 8
    desc, ho, wo = conv_desc(dtype, n, h, w, cin, cout, k, s, p, d, ldx, ldy)
    xb, xv = to_nhwc(x, dtype, ldx, 8)
    wpk, wpt = pack(krsc(wt, dtype), dtype)
    go = rnd((n, cout, ho, wo), 44, dtype)
    gb, gv = to_nhwc(go, dtype, ldy, 8)
    ref = F.conv2d(x, wt, None, s, p, d)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, wt, None, s, p, d).backward(go)
    outs = {}
    try:
        for variant in (2, 0):
            L.conv_variant(variant)
            yb = torch.full((n, ho, wo, ldy), 7.0, dtype=dtype, device=DEV)
            yv = yb[..., 8:]
            st = torch.zeros(2, groups, coutp, dtype=torch.float64, device=DEV)
            L.call("bg_conv2d_fwd_stats", desc, xv.data_ptr(), wpk.data_ptr(), yv.data_ptr(), st[0].data_ptr(),
                   st[1].data_ptr(), groups)
            assert_close(from_nhwc(yv, cout), ref, tol(dtype), f"fwd (variant {variant})")
            assert (yb[..., :8] == 7.0).all() and (yb[..., 8 + coutp:] == 7.0).all(), "wrote outside its channel slice"
            if coutp > cout:
                assert (yv[..., cout:coutp] == 0).all()
            y = yv[..., :coutp].double()
            per = n // groups
            for g in range(groups):
                yg = y[per * g:per * (g + 1)]
                assert_close(st[0, g].cpu(), yg.sum((0, 1, 2)).cpu(), 1e-5, f"sum[{g}] (variant {variant})")
                assert_close(st[1, g].cpu(), (yg * yg).sum((0, 1, 2)).cpu(), 1e-5, f"sumsq[{g}] (variant {variant})")
            dxb = torch.full((n, h, w, ldx), 5.0, dtype=dtype, device=DEV)
            dxv = dxb[..., 8:]
            L.call("bg_conv2d_bwd_data", desc, gv.data_ptr(), wpt.data_ptr(), dxv.data_ptr())
            assert_close(from_nhwc(dxv, cin), xr.grad, tol(dtype), f"bwd_data (variant {variant})")
            assert (dxb[..., :8] == 5.0).all() and (dxb[..., 8 + cinp:] == 5.0).all()
            outs[variant] = (yv.float().cpu(), dxv.float().cpu())
    finally:
        L.conv_variant(-1)
    # same products, same fp32 accumulation per output element up to the order of the K-steps: the two tile families
    # agree far inside the test tolerance
    assert_close(outs[2][0][..., :cout], outs[0][0][..., :cout], 2e-6 if dtype == torch.float32 else 8e-3, "fat vs classic fwd")


@pytest.mark.parametrize("case", [(3, 1000, 72, 200, 0), (5, 2 * 24 * 20, 728, 728, 16), (1, 4099, 264, 520, 8), (2, 77, 8, 8, 0),
                                  (50, 6 * 12 * 8, 728, 728, 0)])
def test_conv_wgrad_grouped(case):
    """bg_conv2d_bwd_weight_grouped: dW_l += dy_l^T x_l for a group of same-shape pointwise layers in one launch (gangs of
    one workgroup per 256 x 256 output tile walking the (layer, pixel) space), against fp32 matmuls of the same
    bf16-representable operands and against the per-layer kernel; accumulation semantics; bit-reproducibility when the
    group spans more layers than ranges (every tile then receives at most two commutative adds)."""
    nl, m, cin, cout, slack = case
    dtype = torch.bfloat16
    g_ = torch.Generator(device=DEV).manual_seed(7)
    ldx, ldy = cin + slack, cout + slack
    xs = [torch.randn((m, ldx), generator=g_, device=DEV).to(dtype) for _ in range(nl)]
    gs = [(torch.randn((m, ldy), generator=g_, device=DEV) / math.sqrt(m)).to(dtype) for _ in range(nl)]
    dws = [torch.full((cout, cin), 0.5, device=DEV) for _ in range(nl)]
    tbl = torch.tensor([[x.data_ptr(), g.data_ptr(), d.data_ptr(), 0] for x, g, d in zip(xs, gs, dws)], dtype=torch.int64)
    L.call("bg_conv2d_bwd_weight_grouped", L.BF16, tbl.data_ptr(), nl, m, cin, cout, ldx, ldy)
    for l in range(nl):
        ref = gs[l][:, :cout].float().t() @ xs[l][:, :cin].float()
        assert_close((dws[l] - 0.5).cpu(), ref.cpu(), 2e-3, f"layer {l}")
    first = [d.clone() for d in dws]
    # per-layer kernel on the same operands
    desc = L.ConvDesc(L.BF16, 1, 1, m, cin, 1, m, cout, 1, 1, 1, 0, 1, ldx, ldy)
    dw1 = torch.zeros(cout, cin, device=DEV)
    L.call("bg_conv2d_bwd_weight", desc, xs[0].data_ptr(), gs[0].data_ptr(), dw1.data_ptr(), None)
    assert_close((first[0] - 0.5).cpu(), dw1.cpu(), 1e-4, "grouped vs per-layer")
    # a second call accumulates; a repeated run on fresh buffers is bit-identical when nl >= 2 (ranges >= one layer)
    L.call("bg_conv2d_bwd_weight_grouped", L.BF16, tbl.data_ptr(), nl, m, cin, cout, ldx, ldy)
    assert_close((dws[0] - 0.5).cpu(), 2 * (first[0] - 0.5).cpu(), 1e-5, "accumulate")
    if nl >= 256 // ((-(-cin // 256)) * (-(-cout // 256))):     # at least as many layers as gangs
        again = [torch.full((cout, cin), 0.5, device=DEV) for _ in range(nl)]
        tbl2 = torch.tensor([[x.data_ptr(), g.data_ptr(), d.data_ptr(), 0] for x, g, d in zip(xs, gs, again)], dtype=torch.int64)
        L.call("bg_conv2d_bwd_weight_grouped", L.BF16, tbl2.data_ptr(), nl, m, cin, cout, ldx, ldy)
        for a, b in zip(again, first):
            assert torch.equal(a, b), "grouped weight gradient is not bit-reproducible"


@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout, k, dil, slack, layers
    (2, 24, 40, 256, 256, 3, 1, 0, 1),       # nine taps as members of one gang (T = 9)
    (1, 37, 33, 304, 256, 3, 1, 8, 1),       # two ci tiles (T = 18), ragged map: 1221 pixels, W not a multiple of anything
    (2, 20, 24, 512, 256, 3, 6, 0, 1),       # dilation 6: most of the border masked
    (1, 16, 20, 1024, 256, 3, 12, 0, 2),     # 4 tiles x 9 taps > 32: taps as layers; dilation beyond half the map; 2 layers
    (3, 12, 20, 256, 512, 3, 2, 0, 1),       # W < 32: several rows per K-step; images change inside a K-step
    (1, 40, 48, 256, 256, 5, 1, 0, 1),       # 5 x 5: 25 taps
    (1, 10, 12, 1024, 256, 3, 1, 0, 9),      # taps as layers, nine caller layers: two launches of at most 7 x 9 kernel layers
])
def test_conv_wgrad_grouped_taps(case):
    """bg_conv2d_bwd_weight_grouped_taps: the weight gradient of k x k stride-1 'same' convolutions through the gang
    kernel (a tap = a pointwise weight gradient against x shifted by the tap's rows / columns, border masked), against
    fp32 matmuls of the same bf16-representable operands with the zero-padded, shifted input and against the per-layer kernel;
    accumulation semantics."""
    n, h, w, cin, cout, k, dil, slack, nl = case
    dtype = torch.bfloat16
    g_ = torch.Generator(device=DEV).manual_seed(11)
    ldx, ldy = cin + slack, cout + slack
    pad = dil * (k - 1) // 2
    xs = [torch.randn((n, h, w, ldx), generator=g_, device=DEV).to(dtype) for _ in range(nl)]
    gs = [(torch.randn((n, h, w, ldy), generator=g_, device=DEV) / math.sqrt(n * h * w)).to(dtype) for _ in range(nl)]
    dws = [torch.full((cout, k, k, cin), 0.25, device=DEV) for _ in range(nl)]
    desc = L.ConvDesc(L.BF16, n, h, w, cin, h, w, cout, k, k, 1, pad, dil, ldx, ldy)
    tbl = torch.tensor([[x.data_ptr(), g.data_ptr(), d.data_ptr(), 0] for x, g, d in zip(xs, gs, dws)], dtype=torch.int64)
    L.call("bg_conv2d_bwd_weight_grouped_taps", desc, tbl.data_ptr(), nl)
    for l in range(nl):
        # reference: one fp32 matmul per tap against the zero-padded, shifted input (no MIOpen backward-weights solver involved)
        xp = torch.nn.functional.pad(xs[l][..., :cin].float(), (0, 0, pad, pad, pad, pad))
        g2 = gs[l][..., :cout].float().reshape(-1, cout)
        ref = torch.empty(cout, k, k, cin, device=DEV)
        for r in range(k):
            for c in range(k):
                ref[:, r, c, :] = g2.t() @ xp[:, r * dil:r * dil + h, c * dil:c * dil + w, :].reshape(-1, cin)
        assert_close((dws[l] - 0.25).cpu(), ref.cpu(), 2e-3, f"layer {l}")
    dw1 = torch.zeros(cout, k, k, cin, device=DEV)
    L.call("bg_conv2d_bwd_weight", desc, xs[0].data_ptr(), gs[0].data_ptr(), dw1.data_ptr(), None)
    assert_close((dws[0] - 0.25).cpu(), dw1.cpu(), 1e-4, "gang vs per-layer")
    first = dws[0].clone()
    L.call("bg_conv2d_bwd_weight_grouped_taps", desc, tbl.data_ptr(), nl)
    assert_close((dws[0] - 0.25).cpu(), 2 * (first - 0.25).cpu(), 1e-5, "accumulate")


def test_conv_wgrad_grouped_taps_rejects_other_geometries():
    x = torch.zeros(1, 8, 8, 256, device=DEV, dtype=torch.bfloat16)
    dw = torch.zeros(256, 3, 3, 256, device=DEV)
    tbl = torch.tensor([[x.data_ptr(), x.data_ptr(), dw.data_ptr(), 0]], dtype=torch.int64)
    for kw in (dict(stride=2, pad=1, ho=4), dict(stride=1, pad=0, ho=6)):
        d = L.ConvDesc(L.BF16, 1, 8, 8, 256, kw["ho"], kw["ho"], 256, 3, 3, kw["stride"], kw["pad"], 1, 256, 256)
        with pytest.raises(RuntimeError, match="stride-1 odd-kernel"):
            L.call("bg_conv2d_bwd_weight_grouped_taps", d, tbl.data_ptr(), 1)


def test_conv_operands_beyond_2gib():
    """Activations larger than 2 GiB (the 2304x1536x32 configuration's entry flow at batch 16): the fat-tile kernel rebases
    its buffer descriptors per tile, so 32-bit offsets never span the tensor.  Sampled pixels (first, around the 2 GiB
    boundary, last) against a matmul; statistics against the stored tensor."""
    dtype = torch.bfloat16
    n, h, w, c = 1, 2912, 2900, 128                      # 8.44 M pixels x 256 B = 2.16 GB per tensor
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((n, h, w, c), generator=g, device=DEV, dtype=torch.float32).to(dtype)
    assert x.numel() * 2 > (1 << 31)
    wt = rnd((c, c, 1, 1), 52, dtype, 1.0 / math.sqrt(c))
    wpk, wpt = pack(krsc(wt, dtype), dtype)
    y = torch.empty_like(x)
    st = torch.zeros(2, 1, c, dtype=torch.float64, device=DEV)
    desc = L.ConvDesc(L.BF16, n, h, w, c, h, w, c, 1, 1, 1, 0, 1, c, c)
    L.call("bg_conv2d_fwd_stats", desc, x.data_ptr(), wpk.data_ptr(), y.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 1)
    m = h * w
    rows = torch.tensor([0, 1, 4_194_303, 4_194_304, 4_194_305, 8_388_607, 8_388_608, 8_388_609, m - 2, m - 1], device=DEV)
    xf, yf = x.view(m, c), y.view(m, c)
    ref = xf[rows].float() @ wt.view(c, c).t().to(DEV)
    assert_close(yf[rows].float().cpu(), ref.cpu(), 1e-2, "fwd rows")
    ssum = torch.zeros(c, dtype=torch.float64, device=DEV)
    for a in range(0, m, 1 << 20):
        ssum += yf[a:a + (1 << 20)].double().sum(0)
    assert_close(st[0, 0].cpu(), ssum.cpu(), 1e-5, "sum")
    dx = torch.empty_like(x)
    L.call("bg_conv2d_bwd_data", desc, y.data_ptr(), wpt.data_ptr(), dx.data_ptr())
    ref2 = yf[rows].float() @ wt.view(c, c).to(DEV)
    assert_close(dx.view(m, c)[rows].float().cpu(), ref2.cpu(), 1e-2, "bwd_data rows")


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_fwd_fused_statistics_per_group(dtype):
    """groups = 2: sums of the two halves of the batch land in separate rows."""
    n, h, w, cin, cout, k = 4, 16, 16, 24, 264, 1       # 2 images per group = 512 pixels = 4 tiles
    x = rnd((n, cin, h, w), 23, dtype)
    wt = rnd((cout, cin, k, k), 24, dtype, 1.0 / math.sqrt(cin))
    desc, ho, wo = conv_desc(dtype, n, h, w, cin, cout, k, 1, 0, 1, up(cin, dtype), up(cout, dtype))
    xb, xv = to_nhwc(x, dtype)
    wpk, _ = pack(krsc(wt, dtype), dtype)
    yb = torch.zeros(n, ho, wo, up(cout, dtype), dtype=dtype, device=DEV)
    st = torch.zeros(2, 2, up(cout, dtype), dtype=torch.float64, device=DEV)
    L.call("bg_conv2d_fwd_stats", desc, xv.data_ptr(), wpk.data_ptr(), yb.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 2)
    y = yb.double()
    for g in range(2):
        yg = y[2 * g:2 * g + 2]
        assert_close(st[0, g].cpu(), yg.sum((0, 1, 2)).cpu(), 1e-5, f"sum[{g}]")
        assert_close(st[1, g].cpu(), (yg * yg).sum((0, 1, 2)).cpu(), 1e-5, f"sumsq[{g}]")


@pytest.mark.parametrize("dtype", DTYPES)
def test_batchnorm_over_statistic_groups(dtype):
    """ops.batch_groups(2): one pass over the concatenated batch == two separate BatchNorm calls
    (outputs, input gradients, parameter gradients summed, running statistics updated twice in order)."""
    from bias_gan_amd.architecture.gpsro import deeplab as dl
    import torch.nn as nn
    n, c, h, w = 2, 24, 9, 7

    class Net(dl.BGModule):
        def __init__(self):
            super().__init__()
            self.bn = nn.BatchNorm2d(c)

        def forward(self, x):
            return dl.apply_norm(self, self.bn, x, act=True)

    def run(batched):
        torch.manual_seed(1)
        net = Net().set_compute_dtype(dtype)
        net.bn.weight.data.uniform_(0.5, 1.5)
        net.bn.bias.data.normal_(0, 0.2)
        net = net.to(DEV).train()
        xs = [rnd((n, c, h, w), 70 + i, dtype, 2.0).to(DEV).requires_grad_(True) for i in range(2)]
        gos = [rnd((n, c, h, w), 80 + i, dtype).to(DEV) for i in range(2)]
        if batched:
            xin = torch.cat(xs)
            with ops.batch_groups(2):
                y = ops.FromInternal.apply(net(ops.ToInternal.apply(xin, up(c, dtype), dtype)), c)
            y.backward(torch.cat(gos))
            ys = [y[:n], y[n:]]
        else:
            ys = []
            for x, go in zip(xs, gos):
                y = ops.FromInternal.apply(net(ops.ToInternal.apply(x, up(c, dtype), dtype)), c)
                y.backward(go)
                ys.append(y)
        torch.cuda.synchronize()
        sd = net.state_dict()
        return ([y.detach().cpu() for y in ys], [x.grad.cpu() for x in xs], net.bn.weight.grad.cpu(), net.bn.bias.grad.cpu(),
                sd["bn.running_mean"].cpu(), sd["bn.running_var"].cpu(), int(sd["bn.num_batches_tracked"]))

    a, b = run(False), run(True)
    for i in range(2):
        assert_close(b[0][i], a[0][i], 1e-6 if dtype == torch.float32 else 1e-2, f"y[{i}]")
        assert_close(b[1][i], a[1][i], 1e-5 if dtype == torch.float32 else 1e-2, f"dx[{i}]")
    assert_close(b[2], a[2], 1e-5 if dtype == torch.float32 else 1e-2, "dgamma")
    assert_close(b[3], a[3], 1e-5 if dtype == torch.float32 else 1e-2, "dbeta")
    assert_close(b[4], a[4], 1e-6, "running_mean")
    assert_close(b[5], a[5], 1e-6, "running_var")
    assert a[6] == b[6] == 2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["batch", "instance"])
def test_norm_act_fused_finalize_matches_split(dtype, mode):
    """bg_norm_act_fwd_stats / bg_norm_act_bwd_apply_stats (finalize folded into the apply kernels)
    against the three-launch sequence they replace."""
    n, h, w, c = 3, 11, 9, 40
    x = (rnd((n, c, h, w), 31, dtype, 2.0) + 0.5).to(dtype).float()
    res, go = rnd((n, c, h, w), 32, dtype), rnd((n, c, h, w), 33, dtype)
    rows, groups, dtc = n * h * w, (n if mode == "instance" else 1), L.dt(dtype)
    (xb, xv), (rb, rv), (gb, gv) = to_nhwc(x, dtype), to_nhwc(res, dtype), to_nhwc(go, dtype)
    aff = mode == "batch"
    gamma = (torch.rand(c) + 0.5).to(DEV) if aff else None
    beta = (torch.randn(c) * 0.2).to(DEV) if aff else None
    f32 = lambda *s: torch.zeros(*s, device=DEV)  # noqa: E731
    f64 = lambda *s: torch.zeros(*s, device=DEV, dtype=torch.float64)  # noqa: E731
    s, ss = f64(groups, c), f64(groups, c)
    L.call("bg_norm_stats", dtc, xv.data_ptr(), rows, c, c, groups, s.data_ptr(), ss.data_ptr())
    out = {}
    for fused in (False, True):
        rm, rvv = (torch.full((c,), 0.1, device=DEV), torch.full((c,), 0.9, device=DEV)) if aff else (None, None)
        mean, rstd, scale, shift = f32(groups, c), f32(groups, c), f32(groups, c), f32(groups, c)
        y = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
        if fused:
            L.call("bg_norm_act_fwd_stats", dtc, xv.data_ptr(), c, s.data_ptr(), ss.data_ptr(), L.ptr(gamma), L.ptr(beta),
                   1e-5, 0.1, L.ptr(rm), L.ptr(rvv), mean.data_ptr(), rstd.data_ptr(), rv.data_ptr(), c, y.data_ptr(), c,
                   rows, c, groups, 1)
        else:
            L.call("bg_norm_finalize", s.data_ptr(), ss.data_ptr(), rows // groups, groups, c, L.ptr(gamma), L.ptr(beta),
                   1e-5, 0.1, L.ptr(rm), L.ptr(rvv), mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
            L.call("bg_norm_act_fwd", dtc, xv.data_ptr(), c, scale.data_ptr(), shift.data_ptr(), rv.data_ptr(), c,
                   y.data_ptr(), c, rows, c, groups, 1)
        s1, s2 = f64(groups, c), f64(groups, c)
        L.call("bg_norm_act_bwd_reduce", dtc, gv.data_ptr(), c, y.data_ptr(), c, xv.data_ptr(), c, mean.data_ptr(),
               rstd.data_ptr(), None, None, rows, c, groups, 1, s1.data_ptr(), s2.data_ptr())
        dx = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
        dres = torch.zeros_like(dx)
        dg, db = (f32(c), f32(c)) if aff else (None, None)
        if fused:
            L.call("bg_norm_act_bwd_apply_stats", dtc, gv.data_ptr(), c, y.data_ptr(), c, xv.data_ptr(), c, s1.data_ptr(),
                   s2.data_ptr(), L.ptr(gamma), L.ptr(beta), mean.data_ptr(), rstd.data_ptr(), 1, L.ptr(dg), L.ptr(db),
                   dx.data_ptr(), c, dres.data_ptr(), c, rows, c, groups, 1)
        else:
            A, B, Cc = f32(groups, c), f32(groups, c), f32(groups, c)
            L.call("bg_norm_bwd_finalize", s1.data_ptr(), s2.data_ptr(), rows // groups, groups, c, L.ptr(gamma),
                   mean.data_ptr(), rstd.data_ptr(), 1, A.data_ptr(), B.data_ptr(), Cc.data_ptr(), L.ptr(dg), L.ptr(db))
            L.call("bg_norm_act_bwd_apply", dtc, gv.data_ptr(), c, y.data_ptr(), c, xv.data_ptr(), c, A.data_ptr(),
                   B.data_ptr(), Cc.data_ptr(), dx.data_ptr(), c, dres.data_ptr(), c, rows, c, groups, 1)
        out[fused] = dict(y=y.float().cpu(), dx=dx.float().cpu(), dres=dres.float().cpu(), mean=mean.cpu(), rstd=rstd.cpu(),
                          rm=None if rm is None else rm.cpu(), rv=None if rvv is None else rvv.cpu(),
                          dg=None if dg is None else dg.cpu(), db=None if db is None else db.cpu())
    a, b = out[False], out[True]
    for k in a:
        if a[k] is not None:
            assert_close(b[k], a[k], 2e-5 if dtype == torch.float32 else 1e-2, k)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["batch", "instance"])
def test_norm_act_bwd_sign_recomputed_from_x(dtype, mode):
    """Layers without a residual: bg_norm_act_bwd_reduce / _apply_stats called with y == NULL take
    the LeakyReLU branch from the recomputed pre-activation; results must be IDENTICAL (bit for bit)
    to the calls that read the stored activation."""
    n, h, w, c = 4, 13, 10, 48
    x = (rnd((n, c, h, w), 41, dtype, 2.0) + 0.3).to(dtype).float()
    go = rnd((n, c, h, w), 43, dtype)
    rows, groups, dtc = n * h * w, (n if mode == "instance" else 1), L.dt(dtype)
    (xb, xv), (gb, gv) = to_nhwc(x, dtype), to_nhwc(go, dtype)
    aff = mode == "batch"
    gamma = (torch.rand(c) - 0.3).to(DEV) if aff else None      # negative gammas included
    beta = (torch.randn(c) * 0.2).to(DEV) if aff else None
    f32 = lambda *s: torch.zeros(*s, device=DEV)  # noqa: E731
    f64 = lambda *s: torch.zeros(*s, device=DEV, dtype=torch.float64)  # noqa: E731
    s, ss = f64(groups, c), f64(groups, c)
    L.call("bg_norm_stats", dtc, xv.data_ptr(), rows, c, c, groups, s.data_ptr(), ss.data_ptr())
    mean, rstd = f32(groups, c), f32(groups, c)
    y = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
    L.call("bg_norm_act_fwd_stats", dtc, xv.data_ptr(), c, s.data_ptr(), ss.data_ptr(), L.ptr(gamma), L.ptr(beta), 1e-5, 0.1,
           None, None, mean.data_ptr(), rstd.data_ptr(), None, 0, y.data_ptr(), c, rows, c, groups, 1)
    out = {}
    for with_y in (True, False):
        yp = y.data_ptr() if with_y else None
        s1, s2 = f64(groups, c), f64(groups, c)
        L.call("bg_norm_act_bwd_reduce", dtc, gv.data_ptr(), c, yp, c, xv.data_ptr(), c, mean.data_ptr(), rstd.data_ptr(),
               L.ptr(gamma), L.ptr(beta), rows, c, groups, 1, s1.data_ptr(), s2.data_ptr())
        dx = torch.zeros(n, h, w, c, dtype=dtype, device=DEV)
        dg, db = (f32(c), f32(c)) if aff else (None, None)
        L.call("bg_norm_act_bwd_apply_stats", dtc, gv.data_ptr(), c, yp, c, xv.data_ptr(), c, s1.data_ptr(), s2.data_ptr(),
               L.ptr(gamma), L.ptr(beta), mean.data_ptr(), rstd.data_ptr(), 1, L.ptr(dg), L.ptr(db), dx.data_ptr(), c, None, 0,
               rows, c, groups, 1)
        out[with_y] = (dx.float().cpu(), None if dg is None else dg.cpu(), None if db is None else db.cpu())
    # block-reduction order is fixed, only the fp64 atomics' arrival order varies: exact for dx up to that
    assert_close(out[False][0], out[True][0], 1e-6, "dx")
    assert (y.float() > 0).sum().item() not in (0, y.numel())
    if aff:
        assert_close(out[False][1], out[True][1], 1e-6, "dgamma")
        assert_close(out[False][2], out[True][2], 1e-6, "dbeta")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(4, 13, 18, 16, 1, 2, 1), (2, 9, 7, 728, 1, 1, 1), (4, 16, 12, 24, 2, 2, 1), (2, 6, 5, 40, 1, 1, 0),
                                  (2, 11, 50, 264, 1, 1, 2)])
def test_dwconv_with_norm_act_prologue(case, dtype):
    """bg_dwconv3x3_fwd_pre / bg_dwconv3x3_bwd_weight_pre (the producer's BatchNorm affine + activation applied to every
    loaded chunk, the activated tensor never stored) against bg_norm_finalize_affine + bg_norm_act_fwd followed by the
    plain depthwise kernels: the forward output must be IDENTICAL bit for bit, the weight gradient equal up to the
    arrival order of its fp32 atomics.  Statistic groups, dilation 2 and all three activation codes are covered; the
    mean / rstd / running statistics of bg_norm_finalize_affine are checked against bg_norm_act_fwd_stats' fused finalize."""
    n, h, w, c, d, groups, act = case
    cp = up(c, dtype)
    x = (rnd((n, c, h, w), 61, dtype, 2.0) + 0.4).to(dtype).float()
    wt = rnd((c, 1, 3, 3), 62, dtype, 0.3)
    go = rnd((n, c, h, w), 63, dtype)
    ld = cp + 8
    (xb, xv), (gb, gv) = to_nhwc(x, dtype, ld, 8), to_nhwc(go, dtype, ld, 8)
    wk = torch.zeros(3, 3, cp, dtype=dtype, device=DEV)
    wk[:, :, :c] = wt[:, 0].permute(1, 2, 0).to(dtype).to(DEV)
    rows, dtc = n * h * w, L.dt(dtype)
    gamma = (torch.rand(cp) - 0.3).to(DEV)      # negative gammas included
    beta = (torch.randn(cp) * 0.2).to(DEV)
    f32 = lambda *s: torch.zeros(*s, device=DEV)  # noqa: E731
    f64 = lambda *s: torch.zeros(*s, device=DEV, dtype=torch.float64)  # noqa: E731
    s, ss = f64(groups, cp), f64(groups, cp)
    L.call("bg_norm_stats", dtc, xv.data_ptr(), rows, cp, ld, groups, s.data_ptr(), ss.data_ptr())
    # reference pipeline: fused-finalize apply kernel, then the plain depthwise kernels
    rm0, rv0 = torch.full((cp,), 0.1, device=DEV), torch.full((cp,), 0.9, device=DEV)
    mean0, rstd0 = f32(groups, cp), f32(groups, cp)
    a = torch.zeros(n, h, w, ld, dtype=dtype, device=DEV)
    L.call("bg_norm_act_fwd_stats", dtc, xv.data_ptr(), ld, s.data_ptr(), ss.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
           1e-5, 0.1, rm0.data_ptr(), rv0.data_ptr(), mean0.data_ptr(), rstd0.data_ptr(), None, 0, a[..., 8:].data_ptr(), ld,
           rows, cp, groups, act)
    desc = L.DwDesc(dtc, n, h, w, cp, h, w, 1, d, ld, ld)
    y0 = torch.zeros(n, h, w, ld, dtype=dtype, device=DEV)
    L.call("bg_dwconv3x3_fwd", desc, a[..., 8:].data_ptr(), wk.data_ptr(), y0[..., 8:].data_ptr())
    dw0 = f32(3, 3, cp)
    L.call("bg_dwconv3x3_bwd_weight", desc, a[..., 8:].data_ptr(), gv.data_ptr(), dw0.data_ptr())
    # fused pipeline
    rm1, rv1 = torch.full((cp,), 0.1, device=DEV), torch.full((cp,), 0.9, device=DEV)
    mean1, rstd1, scale, shift = f32(groups, cp), f32(groups, cp), f32(groups, cp), f32(groups, cp)
    L.call("bg_norm_finalize_affine", s.data_ptr(), ss.data_ptr(), rows // groups, groups, cp, gamma.data_ptr(), beta.data_ptr(),
           1e-5, 0.1, rm1.data_ptr(), rv1.data_ptr(), mean1.data_ptr(), rstd1.data_ptr(), scale.data_ptr(), shift.data_ptr())
    y1 = torch.zeros(n, h, w, ld, dtype=dtype, device=DEV)
    L.call("bg_dwconv3x3_fwd_pre", desc, xv.data_ptr(), scale.data_ptr(), shift.data_ptr(), groups, act, wk.data_ptr(),
           y1[..., 8:].data_ptr())
    dw1 = f32(3, 3, cp)
    L.call("bg_dwconv3x3_bwd_weight_pre", desc, xv.data_ptr(), scale.data_ptr(), shift.data_ptr(), groups, act, gv.data_ptr(),
           dw1.data_ptr())
    assert torch.equal(mean0, mean1) and torch.equal(rstd0, rstd1)
    assert_close(rm1.cpu(), rm0.cpu(), 1e-6, "running_mean")  # same formula; the compiler contracts the two copies differently
    assert_close(rv1.cpu(), rv0.cpu(), 1e-6, "running_var")
    # ... and the one-launch form (finalize folded into the depthwise kernel's prologue)
    rm2, rv2 = torch.full((cp,), 0.1, device=DEV), torch.full((cp,), 0.9, device=DEV)
    mean2, rstd2, scale2, shift2 = f32(groups, cp), f32(groups, cp), f32(groups, cp), f32(groups, cp)
    y2 = torch.zeros(n, h, w, ld, dtype=dtype, device=DEV)
    L.call("bg_dwconv3x3_fwd_pre_stats", desc, xv.data_ptr(), s.data_ptr(), ss.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
           1e-5, 0.1, rm2.data_ptr(), rv2.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), scale2.data_ptr(), shift2.data_ptr(),
           groups, act, wk.data_ptr(), y2[..., 8:].data_ptr())
    assert torch.equal(mean2, mean1) and torch.equal(rstd2, rstd1)
    assert torch.equal(scale2, scale) and torch.equal(shift2, shift)
    assert_close(rm2.cpu(), rm1.cpu(), 1e-6, "running_mean (folded finalize)")
    assert_close(rv2.cpu(), rv1.cpu(), 1e-6, "running_var (folded finalize)")
    assert torch.equal(y2, y1), (y2.float() - y1.float()).abs().max().item()
    assert torch.equal(y0, y1), (y0.float() - y1.float()).abs().max().item()
    assert y0[..., 8:].float().abs().sum().item() > 0
    assert_close(dw1.cpu(), dw0.cpu(), 1e-5, "dw wgrad (fused prologue)")
    # and against torch in fp32 on the bf16-rounded activation the reference pipeline stored
    ar = from_nhwc(a[..., 8:], c)
    ref = F.conv2d(F.pad(ar, (d, d, d, d)), wt, None, 1, 0, d, groups=c)
    assert_close(from_nhwc(y1[..., 8:], c), ref, tol(dtype), "dw fwd (fused prologue) vs torch")
    # unsupported geometry is refused, not silently mis-computed
    bad = L.DwDesc(dtc, n, h, w, cp, -(-h // 2), -(-w // 2), 2, 1, ld, ld)
    with pytest.raises(RuntimeError):
        L.call("bg_dwconv3x3_fwd_pre", bad, xv.data_ptr(), scale.data_ptr(), shift.data_ptr(), groups, act, wk.data_ptr(),
               y1[..., 8:].data_ptr())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 13, 18, 16, 1), (1, 9, 7, 728, 1), (2, 16, 12, 24, 2), (2, 11, 50, 264, 1), (8, 30, 20, 40, 1)])
def test_dwconv_bwd_data_plus_addend(case, dtype):
    """bg_dwconv3x3_bwd_data_add (data gradient + the skip path's gradient in one pass) against bg_dwconv3x3_bwd_data
    followed by an fp32 add of the stored result and the addend, both against torch; in place (dx aliasing the addend)
    gives the same bits."""
    n, h, w, c, d = case
    cp = up(c, dtype)
    wt = rnd((c, 1, 3, 3), 82, dtype, 0.3)
    go, ad = rnd((n, c, h, w), 83, dtype), rnd((n, c, h, w), 84, dtype)
    ld = cp + 8
    (gb, gv), (ab, av) = to_nhwc(go, dtype, ld, 8), to_nhwc(ad, dtype, ld, 8)
    wk = torch.zeros(3, 3, cp, dtype=dtype, device=DEV)
    wk[:, :, :c] = wt[:, 0].permute(1, 2, 0).to(dtype).to(DEV)
    desc = L.DwDesc(L.dt(dtype), n, h, w, cp, h, w, 1, d, ld, ld)
    dx1 = torch.zeros(n, h, w, ld, dtype=dtype, device=DEV)
    L.call("bg_dwconv3x3_bwd_data_add", desc, gv.data_ptr(), wk.data_ptr(), av.data_ptr(), ld, dx1[..., 8:].data_ptr())
    xr = torch.zeros(n, c, h, w, requires_grad=True)
    F.conv2d(F.pad(xr, (d, d, d, d)), wt, None, 1, 0, d, groups=c).backward(go)
    assert_close(from_nhwc(dx1[..., 8:], c), xr.grad + ad, tol(dtype), "dw bwd_data + addend vs torch")
    dx0 = torch.zeros(n, h, w, ld, dtype=dtype, device=DEV)
    L.call("bg_dwconv3x3_bwd_data", desc, gv.data_ptr(), wk.data_ptr(), dx0[..., 8:].data_ptr())
    two_pass = (dx0[..., 8:].float() + av.float()).to(dtype)
    # the fused kernel adds before rounding: at most one unit in the last place of the storage type apart
    assert_close(dx1[..., 8:].float().cpu(), two_pass.float().cpu(), 1e-6 if dtype == torch.float32 else 8e-3, "fused vs two passes")
    ab2 = ab.clone()
    L.call("bg_dwconv3x3_bwd_data_add", desc, gv.data_ptr(), wk.data_ptr(), ab2[..., 8:].data_ptr(), ld, ab2[..., 8:].data_ptr())
    assert torch.equal(ab2[..., 8:], dx1[..., 8:])
    assert torch.equal(ab2[..., :8], ab[..., :8])   # the neighbouring channel slice is untouched


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("p", [0, 1])
def test_avgpool2x2(dtype, p):
    """bg_avgpool2x2 forward and adjoint vs F.avg_pool2d(2, 1, p) (count_include_pad default)."""
    n, c, h, w = 2, 24, 7, 10
    x = rnd((n, c, h, w), 51, dtype).requires_grad_(True)
    ref = F.avg_pool2d(x, 2, 1, p)
    go = rnd(tuple(ref.shape), 52, dtype)
    ref.backward(go)
    ho, wo = h + 2 * p - 1, w + 2 * p - 1
    (xb, xv), (gb, gv) = to_nhwc(x.detach(), dtype), to_nhwc(go, dtype)
    cp = xv.shape[-1]
    y = torch.zeros(n, ho, wo, cp, dtype=dtype, device=DEV)
    L.call("bg_avgpool2x2", L.dt(dtype), xv.data_ptr(), cp, y.data_ptr(), cp, n, h, w, ho, wo, cp, -p)
    assert_close(from_nhwc(y, c), ref.detach(), tol(dtype), "y")
    dx = torch.zeros(n, h, w, cp, dtype=dtype, device=DEV)
    L.call("bg_avgpool2x2", L.dt(dtype), gv.data_ptr(), cp, dx.data_ptr(), cp, n, ho, wo, h, w, cp, p - 1)
    assert_close(from_nhwc(dx, c), x.grad, tol(dtype), "dx")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("pad,op", [(1, (0, 1)), (0, (1, 0)), (1, (1, 0)), (1, (1, 1))])
def test_conv_transpose_via_conv_entry_points(dtype, pad, op):
    """nn.ConvTranspose2d(3, stride 2, output_padding as in deeplab.py:406-431) through ops.ConvTranspose2dFn:
    forward = bg_conv2d_bwd_data, input gradient = bg_conv2d_fwd, weight gradient = bg_conv2d_bwd_weight."""
    from bias_gan_amd.architecture.gpsro import deeplab as dl
    n, cin, cout, h, w = 2, 24, 16, 5, 6
    m = dl.ConvTranspose2d(cin, cout, 3, stride=2, padding=pad, output_padding=op).set_compute_dtype(dtype)
    wt = rnd((cin, cout, 3, 3), 61, dtype, 0.2)
    m.weight.data.copy_(wt)
    m.to(DEV)
    x = rnd((n, cin, h, w), 62, dtype)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, None, 2, pad, op)
    go = rnd(tuple(ref.shape), 63, dtype)
    ref.backward(go)
    xd = x.to(DEV).requires_grad_(True)
    xi = ops.ToInternal.apply(xd, up(cin, dtype), dtype)
    y = ops.FromInternal.apply(m(xi), cout)
    assert tuple(y.shape) == tuple(ref.shape)
    assert_close(y.detach().cpu(), ref.detach(), tol(dtype), "y")
    y.backward(go.to(DEV))
    torch.cuda.synchronize()
    assert_close(xd.grad.cpu(), xr.grad, tol(dtype) * 2, "dx")
    assert_close(m.weight.grad.cpu(), wr.grad, 5e-4 if dtype == torch.float32 else 2e-2, "dw")


@pytest.mark.parametrize("kind", ["l1", "smooth_l1", "l2"])
@pytest.mark.parametrize("weighted", [False, True])
def test_pixel_losses(kind, weighted):
    """nn.L1Loss / nn.SmoothL1Loss / nn.MSELoss and the weighted forms (train_gan.py:142-152, losses.py:101-126)."""
    from bias_gan_amd.utils import losses as bl
    torch.manual_seed(3)
    p = (torch.randn(2, 5, 9, 11) * 1.5).requires_grad_(True)
    t, w = torch.randn(2, 5, 9, 11), torch.rand(2, 5, 9, 11)
    f = {"l1": F.l1_loss, "smooth_l1": F.smooth_l1_loss, "l2": F.mse_loss}[kind]
    ref = (f(p, t, reduction="none") * w).mean() if weighted else f(p, t)
    ref.backward()
    pd = p.detach().to(DEV).requires_grad_(True)
    if weighted:
        crit = bl.L2LossWeighted() if kind == "l2" else bl.L1LossWeighted(smooth=kind == "smooth_l1")
        got = crit(pd, t.to(DEV), w.to(DEV))
    else:
        got = {"l1": bl.L1Loss, "smooth_l1": bl.SmoothL1Loss, "l2": bl.MSELoss}[kind]()(pd, t.to(DEV))
    got.backward()
    assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert_close(pd.grad.cpu(), p.grad, 1e-5, "dp")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 3, 4, 512, 264, 3, 1, 1, 1), (4, 1, 1, 2048, 256, 1, 1, 0, 1), (1, 5, 6, 96, 40, 3, 2, 1, 1),
                                  (2, 4, 3, 1032, 136, 3, 1, 2, 2)])
def test_conv_split_k(case, dtype):
    """bg_conv2d_fwd_splitk / bg_conv2d_bwd_data_splitk + bg_splitk_reduce (few tiles, long reduction): partial tiles
    of K ranges in per-split workspace slices, summed in index order == the one-launch convolution, bit-identical
    from run to run; a split starting in the middle of a tap, strided and dilated taps."""
    n, h, w, cin, cout, k, s, p, d = case
    x = rnd((n, cin, h, w), 61, dtype)
    wt = rnd((cout, cin, k, k), 62, dtype, 1.0 / math.sqrt(cin * k * k))
    cinp, coutp = up(cin, dtype), up(cout, dtype)
    desc, ho, wo = conv_desc(dtype, n, h, w, cin, cout, k, s, p, d, cinp, coutp)
    xb, xv = to_nhwc(x, dtype)
    wpk, wpt = pack(krsc(wt, dtype), dtype)
    ref = F.conv2d(x, wt, None, s, p, d)
    gran = 32 if dtype == torch.bfloat16 else 16

    def legal(ksteps, want):       # the split counts the entry points accept: non-empty ranges of ceil(ksteps / splits)
        per = -(-ksteps // want)
        return -(-ksteps // per)

    kf, kb = k * k * -(-cinp // gran), k * k * -(-coutp // gran)
    for want in (2, 5, 7):
        splits = legal(kf, want)
        outs = []
        for _ in range(2):
            ws = torch.full((splits, n * ho * wo, coutp), float("nan"), device=DEV)
            yb = torch.zeros(n, ho, wo, coutp + 8, dtype=dtype, device=DEV)
            L.call("bg_conv2d_fwd_splitk", desc, xv.data_ptr(), wpk.data_ptr(), ws.data_ptr(), splits)
            L.call("bg_splitk_reduce", L.dt(dtype), ws.data_ptr(), splits, n * ho * wo, coutp, yb.data_ptr(), coutp + 8)
            outs.append(yb)
        assert torch.equal(outs[0], outs[1])
        assert_close(from_nhwc(outs[0][..., :coutp], cout), ref, tol(dtype), f"fwd, {splits} splits")
        assert (outs[0][..., cout:] == 0).all()
    go = rnd((n, cout, ho, wo), 63, dtype)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, wt, None, s, p, d).backward(go)
    gb, gv = to_nhwc(go, dtype)
    for want in (3, 6):
        splits = legal(kb, want)
        ws = torch.full((splits, n * h * w, cinp), float("nan"), device=DEV)
        dxb = torch.zeros(n, h, w, cinp, dtype=dtype, device=DEV)
        L.call("bg_conv2d_bwd_data_splitk", desc, gv.data_ptr(), wpt.data_ptr(), ws.data_ptr(), splits)
        L.call("bg_splitk_reduce", L.dt(dtype), ws.data_ptr(), splits, n * h * w, cinp, dxb.data_ptr(), cinp)
        assert_close(from_nhwc(dxb, cin), xr.grad, tol(dtype), f"bwd_data, {splits} splits")
    bad = next((c_ for c_ in range(2, kf) if legal(kf, c_) != c_), None)   # a count that would leave a split empty
    if bad is not None:
        with pytest.raises(RuntimeError, match="non-empty"):
            L.call("bg_conv2d_fwd_splitk", desc, xv.data_ptr(), wpk.data_ptr(), ws.data_ptr(), bad)


@pytest.mark.parametrize("with_dw", [True, False])
@pytest.mark.parametrize("case", [(4, 13, 18, 16, 1, 2, 1), (2, 9, 7, 728, 1, 1, 1), (4, 16, 12, 24, 2, 2, 1), (2, 6, 5, 40, 1, 1, 0),
                                  (2, 11, 50, 264, 1, 1, 2), (2, 72, 48, 728, 1, 1, 1), (2, 37, 101, 136, 2, 1, 1), (3, 5, 130, 72, 1, 3, 1),
                                  (4, 72, 48, 728, 1, 2, 1)])     # last: 72 tiles per slab on 21 workgroups -> a workgroup's tiles cross the statistic groups
def test_dwconv_bwd_fused_matches_the_three_kernels(case, with_dw):
    """bg_dwconv3x3_bwd_fused (one pass over dy and x: data gradient, weight gradient on the recomputed activation and
    the two BatchNorm-backward statistics) against the three round-2 kernels it replaces -- bg_dwconv3x3_bwd_data,
    bg_dwconv3x3_bwd_weight_pre, bg_norm_act_bwd_reduce -- on the same bf16 tensors.  The data gradient is the same nine
    fp32 products per element summed in another order, then rounded to bf16: equal to bf16 rounding (1 ulp = 2^-8
    relative, on a few elements); weight gradient and statistics sum the same terms over the pixels in another order:
    2e-5 of their max.  Also against torch in fp32 (1e-2 of max, the bf16 kernel bound)."""
    n, h, w, c, d, groups, act = case
    dtype = torch.bfloat16
    cp = up(c, dtype)
    x = (rnd((n, c, h, w), 71, dtype, 2.0) + 0.4).to(dtype).float()
    wt = rnd((c, 1, 3, 3), 72, dtype, 0.3)
    go = rnd((n, c, h, w), 73, dtype)
    ld = cp + 8
    (xb, xv), (gb, gv) = to_nhwc(x, dtype, ld, 8), to_nhwc(go, dtype, ld, 8)
    wk = torch.zeros(3, 3, cp, dtype=dtype, device=DEV)
    wk[:, :, :c] = wt[:, 0].permute(1, 2, 0).to(dtype).to(DEV)
    rows, dtc = n * h * w, L.dt(dtype)
    gamma = (torch.rand(cp) - 0.3).to(DEV)
    beta = (torch.randn(cp) * 0.2).to(DEV)
    f32 = lambda *s: torch.zeros(*s, device=DEV)  # noqa: E731
    f64 = lambda *s: torch.zeros(*s, device=DEV, dtype=torch.float64)  # noqa: E731
    s, ss = f64(groups, cp), f64(groups, cp)
    L.call("bg_norm_stats", dtc, xv.data_ptr(), rows, cp, ld, groups, s.data_ptr(), ss.data_ptr())
    mean, rstd, scale, shift = f32(groups, cp), f32(groups, cp), f32(groups, cp), f32(groups, cp)
    L.call("bg_norm_finalize_affine", s.data_ptr(), ss.data_ptr(), rows // groups, groups, cp, gamma.data_ptr(), beta.data_ptr(),
           1e-5, 0.1, None, None, mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
    # the three separate kernels
    ldd = cp + 16
    da0 = torch.full((n, h, w, ldd), 3.0, dtype=dtype, device=DEV)
    L.call("bg_dwconv3x3_bwd_data", L.DwDesc(dtc, n, h, w, cp, h, w, 1, d, ldd, ld), gv.data_ptr(), wk.data_ptr(), da0.data_ptr())
    dw0 = f32(3, 3, cp)
    L.call("bg_dwconv3x3_bwd_weight_pre", L.DwDesc(dtc, n, h, w, cp, h, w, 1, d, ld, ld), xv.data_ptr(), scale.data_ptr(),
           shift.data_ptr(), groups, act, gv.data_ptr(), dw0.data_ptr())
    s1a, s2a = f64(groups, cp), f64(groups, cp)
    L.call("bg_norm_act_bwd_reduce", dtc, da0.data_ptr(), ldd, None, 0, xv.data_ptr(), ld, mean.data_ptr(), rstd.data_ptr(),
           gamma.data_ptr(), beta.data_ptr(), rows, cp, groups, act, s1a.data_ptr(), s2a.data_ptr())
    # the fused kernel
    da1 = torch.full((n, h, w, ldd), 3.0, dtype=dtype, device=DEV)
    dw1 = f32(3, 3, cp)
    s1b, s2b = f64(groups, cp), f64(groups, cp)
    L.call("bg_dwconv3x3_bwd_fused", L.DwDesc(dtc, n, h, w, cp, h, w, 1, d, ld, ld), gv.data_ptr(), wk.data_ptr(), xv.data_ptr(),
           scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), groups, act, da1.data_ptr(), ldd,
           dw1.data_ptr() if with_dw else None, s1b.data_ptr(), s2b.data_ptr())
    torch.cuda.synchronize()
    assert (da1[..., cp:].float() == 3.0).all(), "lanes beyond C were written"
    a, b = da0[..., :cp].float().cpu(), da1[..., :cp].float().cpu()
    assert (a - b).abs().max().item() <= 2.0 ** -7 * a.abs().max().item(), (a - b).abs().max().item()
    assert ((a != b).float().mean().item()) < 2e-2, "more than 2 % of the elements round differently"
    ref = torch.nn.grad.conv2d_input((n, c, h + 2 * d, w + 2 * d), wt, go, 1, 0, d, groups=c)[:, :, d:d + h, d:d + w]
    assert_close(from_nhwc(da1, c), ref, 1e-2, "fused data gradient vs torch")
    if with_dw:
        assert_close(dw1.cpu(), dw0.cpu(), 2e-5, "fused depthwise weight gradient")
    else:
        assert (dw1 == 0).all()
    # statistics: the fused kernel sums g from ITS stored da; where da rounds differently (<= 1 bf16 ulp on < 2 % of the
    # elements) the sums move by that much
    assert_close(s1b.cpu(), s1a.cpu(), 2e-4, "sum g")
    assert_close(s2b.cpu(), s2a.cpu(), 2e-4, "sum g * xhat")
    # ... and exactly (2e-5) against the separate reduction run on the fused kernel's own da
    s1c, s2c = f64(groups, cp), f64(groups, cp)
    L.call("bg_norm_act_bwd_reduce", dtc, da1.data_ptr(), ldd, None, 0, xv.data_ptr(), ld, mean.data_ptr(), rstd.data_ptr(),
           gamma.data_ptr(), beta.data_ptr(), rows, cp, groups, act, s1c.data_ptr(), s2c.data_ptr())
    assert_close(s1b.cpu(), s1c.cpu(), 2e-5, "sum g vs the reduction of the same da")
    assert_close(s2b.cpu(), s2c.cpu(), 2e-5, "sum g * xhat vs the reduction of the same da")


@pytest.mark.parametrize("with_dw", [True, False])
@pytest.mark.parametrize("case", [(2, 9, 7, 728, 1, 1), (4, 13, 18, 16, 2, 1), (2, 6, 5, 40, 1, 0), (2, 11, 50, 264, 1, 2), (2, 72, 48, 728, 1, 1),
                                  (2, 37, 101, 136, 2, 1), (3, 5, 130, 72, 3, 1), (4, 72, 48, 728, 2, 1), (1, 3, 5, 8, 1, 1)])
def test_dwconv_bwd_fork_matches_the_kernels_it_replaces(case, with_dw):
    """bg_dwconv3x3_bwd_fork (the fork at a Block's input, backward, in one pass: deeplab.py:134-141) against the launches
    it replaces on the same bf16 tensors -- bg_dwconv3x3_bwd_data_add (t = dwT(dy) + skip), bg_dwconv3x3_bwd_weight (a0 against
    dy), bg_norm_act_bwd_reduce with y = a0 (sum g, sum g * xhat, g = t * act'(a0)) and the residual-gradient output of
    bg_norm_act_bwd_apply (gout = t * act'(a0)) -- and against torch in fp32.  t rounds like the separate kernel's (the same
    fp32 terms in another order: <= 1 bf16 ulp on a few elements); the sums are taken from the UNROUNDED product like the
    reduce kernel's.  Last case: a map smaller than one tile (every range of the staging descriptors starts or ends outside)."""
    n, h, w, c, groups, act = case
    dtype = torch.bfloat16
    cp = up(c, dtype)
    a0 = rnd((n, c, h, w), 81, dtype, 1.5)
    z = (rnd((n, c, h, w), 82, dtype, 2.0) + 0.3).to(dtype).float()
    wt = rnd((c, 1, 3, 3), 83, dtype, 0.3)
    go = rnd((n, c, h, w), 84, dtype)
    sk = rnd((n, c, h, w), 85, dtype, 0.7)
    ld, lds_, ldz, ldo = cp + 8, cp + 16, cp + 24, cp + 32
    (ab, av), (gb, gv) = to_nhwc(a0, dtype, ld, 8), to_nhwc(go, dtype, ld, 8)
    (sb, sv), (zb, zv) = to_nhwc(sk, dtype, lds_, 8), to_nhwc(z, dtype, ldz, 8)
    wk = torch.zeros(3, 3, cp, dtype=dtype, device=DEV)
    wk[:, :, :c] = wt[:, 0].permute(1, 2, 0).to(dtype).to(DEV)
    rows, dtc = n * h * w, L.dt(dtype)
    f32 = lambda *s_: torch.zeros(*s_, device=DEV)  # noqa: E731
    f64 = lambda *s_: torch.zeros(*s_, device=DEV, dtype=torch.float64)  # noqa: E731
    s, ss = f64(groups, cp), f64(groups, cp)
    L.call("bg_norm_stats", dtc, zv.data_ptr(), rows, cp, ldz, groups, s.data_ptr(), ss.data_ptr())
    gamma, beta = torch.ones(cp, device=DEV), torch.zeros(cp, device=DEV)
    mean, rstd, scale, shift = f32(groups, cp), f32(groups, cp), f32(groups, cp), f32(groups, cp)
    L.call("bg_norm_finalize_affine", s.data_ptr(), ss.data_ptr(), rows // groups, groups, cp, gamma.data_ptr(), beta.data_ptr(),
           1e-5, 0.1, None, None, mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
    # ---- the separate launches
    t0 = torch.full((n, h, w, ldo), 3.0, dtype=dtype, device=DEV)
    L.call("bg_dwconv3x3_bwd_data_add", L.DwDesc(dtc, n, h, w, cp, h, w, 1, 1, ldo, ld), gv.data_ptr(), wk.data_ptr(), sv.data_ptr(),
           lds_, t0.data_ptr())
    dw0 = f32(3, 3, cp)
    L.call("bg_dwconv3x3_bwd_weight", L.DwDesc(dtc, n, h, w, cp, h, w, 1, 1, ld, ld), av.data_ptr(), gv.data_ptr(), dw0.data_ptr())
    s1a, s2a = f64(groups, cp), f64(groups, cp)
    L.call("bg_norm_act_bwd_reduce", dtc, t0.data_ptr(), ldo, av.data_ptr() if act else None, ld, zv.data_ptr(), ldz, mean.data_ptr(),
           rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, cp, groups, act, s1a.data_ptr(), s2a.data_ptr())
    # ---- the fused kernel
    g1 = torch.full((n, h, w, ldo), 3.0, dtype=dtype, device=DEV)
    dw1 = f32(3, 3, cp)
    s1b, s2b = f64(groups, cp), f64(groups, cp)
    L.call("bg_dwconv3x3_bwd_fork", L.DwDesc(dtc, n, h, w, cp, h, w, 1, 1, ld, ld), gv.data_ptr(), wk.data_ptr(), av.data_ptr(),
           sv.data_ptr(), lds_, zv.data_ptr(), ldz, mean.data_ptr(), rstd.data_ptr(), groups, act, g1.data_ptr(), ldo,
           dw1.data_ptr() if with_dw else None, s1b.data_ptr(), s2b.data_ptr())
    torch.cuda.synchronize()
    assert (g1[..., cp:].float() == 3.0).all(), "lanes beyond C were written"
    slope = 1.0 if act == 0 else 0.0 if act == 2 else 0.2
    fac = torch.where(av[..., :cp].float() > 0, 1.0, slope)
    want = (t0[..., :cp].float() * fac).to(dtype).float().cpu()          # what the apply pass would have written as the residual gradient
    got = g1[..., :cp].float().cpu()
    assert (want - got).abs().max().item() <= 2.0 ** -7 * want.abs().max().item(), (want - got).abs().max().item()
    assert ((want != got).float().mean().item()) < 2e-2, "more than 2 % of the elements round differently"
    ref_t = torch.nn.grad.conv2d_input((n, c, h + 2, w + 2), wt, go, 1, 0, 1, groups=c)[:, :, 1:1 + h, 1:1 + w] + sk
    ref = ref_t * torch.where(a0 > 0, 1.0, slope)
    assert_close(from_nhwc(g1, c), ref, 1e-2, "fork gradient vs torch")
    if with_dw:
        assert_close(dw1.cpu(), dw0.cpu(), 2e-5, "fork depthwise weight gradient")
        ref_dw = torch.nn.grad.conv2d_weight(torch.nn.functional.pad(a0, (1, 1, 1, 1)), (c, 1, 3, 3), go, 1, 0, 1, groups=c)
        assert_close(dw1[:, :, :c].permute(2, 0, 1).cpu(), ref_dw[:, 0], 1e-2, "fork depthwise weight gradient vs torch")
    else:
        assert (dw1 == 0).all()
    assert_close(s1b.cpu(), s1a.cpu(), 2e-4, "sum g")
    assert_close(s2b.cpu(), s2a.cpu(), 2e-4, "sum g * xhat")
