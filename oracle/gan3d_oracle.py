"""CPU restatement of the reference's 3-D DeepLab GAN path -- TEST INFRASTRUCTURE ONLY.

Same conventions as oracle/gan_oracle.py (functional fp32 evaluation over a {state_dict key: tensor} map,
every function citing the reference lines it restates).  Reference files, under /root/reference/src/deepCam/:
architecture/gpsro/deeplab3d.py, architecture/gpsro/deeplab3d_gan.py, gpsro_train/train_gan3d.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned against the reference by tests/golden/*3d*.npz (tests/golden/make_golden.py, `gan3d` target).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F

from .gan_oracle import (NormCtx, State, _norm_entries, block_units, ceil_div, fill_state, lrelu, norm,  # noqa: F401
                         trainable_keys)


# ---------------------------------------------------------------------------- topology
def xception3d_block_table(os: int = 16) -> List[dict]:
    """Block3d hyper-parameters of Xception3d (deeplab3d.py:107-163): the 2-D table with a 64-wide stem."""
    if os == 16:
        b3_stride, mid_rate, exit_rates = 2, 1, (1, 2)
    elif os == 8:
        b3_stride, mid_rate, exit_rates = 1, 2, (2, 4)
    else:
        raise NotImplementedError
    tbl = [
        dict(name="block1", cin=64, cout=128, reps=2, stride=2, dil=1, start_relu=False, grow_first=True, is_last=False),
        dict(name="block2", cin=128, cout=256, reps=2, stride=2, dil=1, start_relu=True, grow_first=True, is_last=False),
        dict(name="block3", cin=256, cout=728, reps=2, stride=b3_stride, dil=1, start_relu=True, grow_first=True, is_last=True),
    ]
    for i in range(4, 20):
        tbl.append(dict(name=f"block{i}", cin=728, cout=728, reps=3, stride=1, dil=mid_rate, start_relu=True,
                        grow_first=True, is_last=False))
    tbl.append(dict(name="block20", cin=728, cout=1024, reps=2, stride=1, dil=exit_rates[0], start_relu=True,
                    grow_first=False, is_last=True))
    return tbl


def out_dhw16(d: int, h: int, w: int) -> Tuple[int, int, int]:
    for _ in range(4):
        d, h, w = ceil_div(d, 2), ceil_div(h, 2), ceil_div(w, 2)
    return d, h, w


def xception3d_spec(prefix: str, cin: int, norm_kind: str, os: int = 16):
    """(key, shape, role) of every Xception3d state_dict entry in registration order (deeplab3d.py:120-163)."""
    k3 = (3, 3, 3)
    s = [(prefix + "conv1.weight", (32, cin) + k3, "conv")]
    s += _norm_entries(prefix + "bn1", 32, norm_kind)
    s.append((prefix + "conv2.weight", (64, 32) + k3, "conv"))
    s += _norm_entries(prefix + "bn2", 64, norm_kind)
    for cfg in xception3d_block_table(os):
        bp = prefix + cfg["name"] + "."
        if cfg["cin"] != cfg["cout"] or cfg["stride"] != 1:
            s.append((bp + "skip.weight", (cfg["cout"], cfg["cin"], 1, 1, 1), "conv"))
            s += _norm_entries(bp + "skipbn", cfg["cout"], norm_kind)
        for u in block_units(cfg):
            up = bp + f"rep.{u['idx']}"
            if u["kind"] == "sep":
                s.append((up + ".conv1.weight", (u["cin"], 1) + k3, "conv"))
                s.append((up + ".pointwise.weight", (u["cout"], u["cin"], 1, 1, 1), "conv"))
            elif u["kind"] == "norm":
                s += _norm_entries(up, u["c"], norm_kind)
    for name, ci, co in (("3", 1024, 1536), ("4", 1536, 1536), ("5", 1536, 2048)):
        s.append((prefix + f"conv{name}.conv1.weight", (ci, 1) + k3, "conv"))
        s.append((prefix + f"conv{name}.pointwise.weight", (co, ci, 1, 1, 1), "conv"))
        s += _norm_entries(prefix + f"bn{name}", co, norm_kind)
    return s


def deeplab3d_spec(prefix: str, cin: int, cout: int, norm_kind: str, os: int = 16, upsampler: str = "Interpolate"):
    """DeepLab3d (deeplab3d.py:468-532) with the Interpolate (:301-312), Deconv or Deconv1x (:342-444) upsampler."""
    s = xception3d_spec(prefix + "xception_features.", cin, norm_kind, os)
    for i in (1, 2, 3, 4):
        k = 1 if i == 1 else 3
        s.append((prefix + f"aspp{i}.atrous_convolution.weight", (256, 2048, k, k, k), "conv"))
        s += _norm_entries(prefix + f"aspp{i}.bn", 256, norm_kind)
    s.append((prefix + "global_avg_pool.1.weight", (256, 2048, 1, 1, 1), "conv"))
    s += _norm_entries(prefix + "global_avg_pool.2", 256, norm_kind)
    s.append((prefix + "conv1.weight", (256, 1280, 1, 1, 1), "conv"))
    s += _norm_entries(prefix + "bn1", 256, norm_kind)
    s.append((prefix + "conv2.weight", (48, 128, 1, 1, 1), "conv"))
    s += _norm_entries(prefix + "bn2", 48, norm_kind)
    if upsampler == "Interpolate":
        up = prefix + "upsample.last_conv."
        s.append((up + "0.weight", (256, 304, 3, 3, 3), "conv"))
        s += _norm_entries(up + "1", 256, norm_kind)
        s.append((up + "3.weight", (256, 256, 3, 3, 3), "conv"))
        s += _norm_entries(up + "4", 256, norm_kind)
        s.append((up + "6.weight", (cout, 256, 1, 1, 1), "conv"))
        s.append((up + "6.bias", (cout,), "bias"))
        return s
    assert upsampler in ("Deconv", "Deconv1x"), upsampler
    n_up = 128 if upsampler == "Deconv1x" else cout
    up, k3 = prefix + "upsample.", (3, 3, 3)
    # nn.ConvTranspose3d weights are [Cin, Cout, k, k, k]
    s.append((up + "deconv1.0.weight", (256, 256) + k3, "conv"))
    s += _norm_entries(up + "deconv1.1", 256, norm_kind)
    s.append((up + "deconv2.0.weight", (256, 256) + k3, "conv"))
    s += _norm_entries(up + "deconv2.1", 256, norm_kind)
    s.append((up + "conv1.0.weight", (256, 304) + k3, "conv"))
    s += _norm_entries(up + "conv1.1", 256, norm_kind)
    s.append((up + "conv1.3.weight", (256, 256) + k3, "conv"))
    s += _norm_entries(up + "conv1.4", 256, norm_kind)
    s.append((up + "conv1.6.weight", (256, 256, 1, 1, 1), "conv"))
    s.append((up + "conv1.6.bias", (256,), "bias"))
    s.append((up + "deconv3.0.weight", (256, 128) + k3, "conv"))
    s += _norm_entries(up + "deconv3.1", 128, norm_kind)
    s.append((up + "last_deconv.0.weight", (128, n_up) + k3, "conv"))
    if upsampler == "Deconv1x":
        ex = prefix + "upsample_extension."
        s += _norm_entries(ex + "init_norm.0", 128, norm_kind)
        s.append((ex + "conv1.0.weight", (64, cin) + k3, "conv"))
        s += _norm_entries(ex + "conv1.1", 64, norm_kind)
        s.append((ex + "conv1.3.weight", (128, 64) + k3, "conv"))
        s += _norm_entries(ex + "conv1.4", 128, norm_kind)
        s.append((ex + "conv2.0.weight", (64, 256) + k3, "conv"))
        s += _norm_entries(ex + "conv2.1", 64, norm_kind)
        s.append((ex + "conv2.3.weight", (cout, 64) + k3, "conv"))
    return s


def generator3d_spec(cin: int, cout: int, noise_dims: int, norm_kind: str, os: int = 16, upsampler: str = "Interpolate"):
    """Generator = noise concat + DeepLab3d under 'model.' (deeplab3d_gan.py:69-100)."""
    return deeplab3d_spec("model.", cin + noise_dims, cout, norm_kind, os, upsampler)


def discriminator3d_spec(cin: int, norm_kind: str, os: int = 16):
    """Discriminator = Xception3d + mean over (D,H,W) + Linear(2048, 1) (deeplab3d_gan.py:12-44)."""
    s = xception3d_spec("xception_features.", cin, norm_kind, os)
    s.append(("linear.weight", (1, 2048), "linear"))
    s.append(("linear.bias", (1,), "bias"))
    return s


# ---------------------------------------------------------------------------- forward
def sepconv3d_same(P: State, key: str, x: torch.Tensor, stride: int, dil: int, ctx: Optional[NormCtx] = None):
    """SeparableConv3d_same: fixed_padding on all three dims, depthwise 3x3x3, pointwise 1x1x1 (deeplab3d.py:22-43)."""
    q = ctx.q if ctx is not None else (lambda t: t)
    wd, wp = P[key + ".conv1.weight"], P[key + ".pointwise.weight"]
    k = wd.shape[-1]
    tot = k + (k - 1) * (dil - 1) - 1
    beg, end = tot // 2, tot - tot // 2
    x = F.pad(x, (beg, end, beg, end, beg, end))
    x = q(F.conv3d(x, q(wd), None, stride, 0, dil, groups=wd.shape[0]))
    return q(F.conv3d(x, q(wp)))


def block3d(P: State, bp: str, cfg: dict, x: torch.Tensor, ctx: NormCtx) -> torch.Tensor:
    """Block3d (deeplab3d.py:46-99); in-place first LeakyReLU made explicit as in gan_oracle.block."""
    q = ctx.q
    a = q(lrelu(x)) if cfg["start_relu"] else x
    h = a
    units = block_units(cfg)
    i = 1 if cfg["start_relu"] else 0
    while i < len(units):
        u = units[i]
        if u["kind"] == "sep":
            h = sepconv3d_same(P, bp + f"rep.{u['idx']}", h, u["stride"], u["dil"], ctx)
            i += 1
        elif u["kind"] == "relu":
            h = q(lrelu(h))
            i += 1
        else:
            h = norm(P, bp + f"rep.{u['idx']}", h, ctx)
            if i == len(units) - 1:
                break
            if units[i + 1]["kind"] == "relu":
                h = q(lrelu(h))
                i += 2
            else:
                h = q(h)
                i += 1
    if cfg["cin"] != cfg["cout"] or cfg["stride"] != 1:
        s = q(F.conv3d(a, q(P[bp + "skip.weight"]), None, cfg["stride"]))
        s = q(norm(P, bp + "skipbn", s, ctx))
    else:
        s = a
    return h + s


def xception3d(P: State, prefix: str, x: torch.Tensor, ctx: NormCtx, os: int = 16):
    """Xception3d.forward (deeplab3d.py:173-221) -> (features, low_level_feat)."""
    q = ctx.q
    x = q(F.conv3d(q(x), q(P[prefix + "conv1.weight"]), None, 2, 1))
    x = q(lrelu(norm(P, prefix + "bn1", x, ctx)))
    x = q(F.conv3d(x, q(P[prefix + "conv2.weight"]), None, 1, 1))
    x = q(lrelu(norm(P, prefix + "bn2", x, ctx)))
    low = None
    for cfg in xception3d_block_table(os):
        x = block3d(P, prefix + cfg["name"] + ".", cfg, x, ctx)
        if cfg["name"] == "block1":
            low = q(lrelu(x))      # aliased and later activated in place by block2 (deeplab3d.py:184-185)
    x = q(x)
    rate = 2 if os == 16 else 4
    for name in ("3", "4", "5"):
        x = sepconv3d_same(P, prefix + f"conv{name}", x, 1, rate, ctx)
        x = q(lrelu(norm(P, prefix + f"bn{name}", x, ctx)))
    return x, low


def trilinear_ac(x: torch.Tensor, size) -> torch.Tensor:
    return F.interpolate(x, size=tuple(size), mode="trilinear", align_corners=True)


def _deconv3d_unit(P: State, key: str, x: torch.Tensor, ctx: NormCtx, out_pad):
    """ConvTranspose3d(3, stride 2, pad 1, no bias) -> normaliser -> AvgPool3d(2, 1, 0) -> LeakyReLU (one nn.Sequential
    of the 3-D DeconvUpsampler, deeplab3d.py:351-354)."""
    q = ctx.q
    x = q(F.conv_transpose3d(x, q(P[key + ".0.weight"]), None, 2, 1, out_pad))
    x = q(norm(P, key + ".1", x, ctx))
    return q(lrelu(q(F.avg_pool3d(x, 2, 1, 0))))


def deconv_upsampler3d(P: State, up: str, x: torch.Tensor, low: torch.Tensor, ctx: NormCtx) -> torch.Tensor:
    """DeconvUpsampler.forward in 3-D (deeplab3d.py:383-395)."""
    q = ctx.q
    x = _deconv3d_unit(P, up + "deconv1", x, ctx, (1, 1, 1))
    x = _deconv3d_unit(P, up + "deconv2", x, ctx, (1, 1, 1))
    x = q(trilinear_ac(x, low.shape[2:]))
    x = torch.cat((x, low), dim=1)
    x = q(lrelu(norm(P, up + "conv1.1", q(F.conv3d(x, q(P[up + "conv1.0.weight"]), None, 1, 1)), ctx)))
    x = q(lrelu(norm(P, up + "conv1.4", q(F.conv3d(x, q(P[up + "conv1.3.weight"]), None, 1, 1)), ctx)))
    x = q(F.conv3d(x, q(P[up + "conv1.6.weight"]), P[up + "conv1.6.bias"]))
    x = _deconv3d_unit(P, up + "deconv3", x, ctx, (0, 1, 0))
    return q(F.conv_transpose3d(x, q(P[up + "last_deconv.0.weight"]), None, 2, 1, (1, 1, 1)))


def upsampler_extension3d(P: State, ex: str, x_in: torch.Tensor, x: torch.Tensor, ctx: NormCtx) -> torch.Tensor:
    """DeconvUpsamplerExtension.forward (deeplab3d.py:436-444)."""
    q = ctx.q
    skip = q(lrelu(norm(P, ex + "conv1.1", q(F.conv3d(x_in, q(P[ex + "conv1.0.weight"]), None, 1, 1)), ctx)))
    skip = q(lrelu(norm(P, ex + "conv1.4", q(F.conv3d(skip, q(P[ex + "conv1.3.weight"]), None, 1, 1)), ctx)))
    x = q(norm(P, ex + "init_norm.0", x, ctx))
    x = q(lrelu(q(F.avg_pool3d(x, 2, 1, 1))))
    x = torch.cat((x, skip), dim=1)
    x = q(lrelu(norm(P, ex + "conv2.1", q(F.conv3d(x, q(P[ex + "conv2.0.weight"]), None, 1, 1)), ctx)))
    return q(F.conv3d(x, q(P[ex + "conv2.3.weight"]), None, 1, 1))


def deeplab3d(P: State, prefix: str, x_in: torch.Tensor, ctx: NormCtx, os: int = 16) -> torch.Tensor:
    """DeepLab3d.forward (deeplab3d.py:534-566); the upsampler in use is read off the state's keys: Interpolate
    (:314-322), Deconv (+ final AvgPool3d(2,1,1), :562-563) or Deconv1x (+ extension)."""
    q = ctx.q
    rates = [1, 6, 12, 18] if os == 16 else [1, 12, 24, 36]
    x, low = xception3d(P, prefix + "xception_features.", x_in, ctx, os)
    branches = []
    for i, r in zip((1, 2, 3, 4), rates):
        w = q(P[prefix + f"aspp{i}.atrous_convolution.weight"])
        b = q(F.conv3d(x, w, None, 1, 0 if r == 1 else r, r))
        branches.append(q(lrelu(norm(P, prefix + f"aspp{i}.bn", b, ctx))))
    g = q(x.mean(dim=(2, 3, 4), keepdim=True))
    g = q(F.conv3d(g, q(P[prefix + "global_avg_pool.1.weight"])))
    g = q(lrelu(norm(P, prefix + "global_avg_pool.2", g, ctx)))
    branches.append(q(trilinear_ac(g, branches[-1].shape[2:])))
    x = torch.cat(branches, dim=1)
    x = q(lrelu(norm(P, prefix + "bn1", q(F.conv3d(x, q(P[prefix + "conv1.weight"]))), ctx)))
    low = q(lrelu(norm(P, prefix + "bn2", q(F.conv3d(low, q(P[prefix + "conv2.weight"]))), ctx)))
    if prefix + "upsample.deconv1.0.weight" in P:
        x = deconv_upsampler3d(P, prefix + "upsample.", x, low, ctx)
        if prefix + "upsample_extension.conv1.0.weight" in P:
            return upsampler_extension3d(P, prefix + "upsample_extension.", x_in, x, ctx)
        return F.avg_pool3d(x, 2, 1, 1)
    D, H, W = x_in.shape[2:]
    x = q(trilinear_ac(x, (ceil_div(D, 4), ceil_div(H, 4), ceil_div(W, 4))))
    x = torch.cat((x, low), dim=1)
    up = prefix + "upsample.last_conv."
    x = q(lrelu(norm(P, up + "1", q(F.conv3d(x, q(P[up + "0.weight"]), None, 1, 1)), ctx)))
    x = q(lrelu(norm(P, up + "4", q(F.conv3d(x, q(P[up + "3.weight"]), None, 1, 1)), ctx)))
    x = q(F.conv3d(x, q(P[up + "6.weight"]), P[up + "6.bias"]))
    return trilinear_ac(x, (D, H, W))


def generator3d(P: State, x: torch.Tensor, ctx: NormCtx, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Generator.forward (deeplab3d_gan.py:88-100)."""
    if noise is not None:
        x = torch.cat((x, noise), dim=1)
    return deeplab3d(P, "model.", x, ctx)


def discriminator3d(P: State, x: torch.Tensor, ctx: NormCtx):
    """Discriminator.forward (deeplab3d_gan.py:28-44): mean over (D,H,W), Linear(2048,1) -> (logits, sigmoid)."""
    f, _ = xception3d(P, "xception_features.", x, ctx)
    logits = F.linear(f.mean(dim=(2, 3, 4)), P["linear.weight"], P["linear.bias"])
    return logits, torch.sigmoid(logits)


def gradient_penalty3d(P: State, fake: torch.Tensor, real: torch.Tensor, eta: torch.Tensor, ctx: NormCtx):
    """deeplab3d_gan.py:103-125: per-SAMPLE flattened gradient norm (unlike the 2-D per-pixel one); first order
    only, so the value is a constant w.r.t. the critic's parameters."""
    xi = (eta * fake + (1.0 - eta) * real).detach().requires_grad_(True)
    logits, _ = discriminator3d(P, xi, ctx)
    (g,) = torch.autograd.grad(logits, xi, torch.ones_like(logits))
    gn = g.reshape(g.shape[0], -1).norm(2, dim=1)
    return ((gn - 1.0) ** 2).mean().detach()


def synthetic_volumes(n: int, c: int, d: int, h: int, w: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, c, d, h, w), generator=g)
    return x, x + 0.1 * torch.randn((n, c, d, h, w), generator=g)


def update_schedule(kind: str, step: int, warmup: int, d_acc_avg: float = 0.0, freq_g: int = 1, freq_d: int = 1,
                    acc_min: float = 0.0, acc_max: float = 1.0, wasserstein: bool = False):
    """(train_generator, train_discriminator) of train_gan3d.py:270-293 (the script compares the dict against
    the strings "static" / "adaptive", which never matches a dict -- the intent, a `type` key, is restated)."""
    if kind == "static":
        return (step < warmup) or (step % freq_g == 0), (step >= warmup) and (step % freq_d == 0)
    if kind == "adaptive" and not wasserstein:
        if step < warmup:
            return True, False
        if d_acc_avg > acc_max:
            return True, False
        if d_acc_avg < acc_min:
            return False, True
        return True, True
    return True, True
