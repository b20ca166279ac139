"""CPU oracle for the Bias-GAN conv-GAN training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``bias-gan_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / timed baseline.

This is a *functional* restatement (plain fp32 PyTorch CPU ops over a flat
``{state_dict key: tensor}`` mapping) of the arithmetic of the reference's hot
path.  It does not contain reference source; each function cites the reference
lines whose behaviour it restates (paths relative to
``/root/reference/src/deepCam``).  It is pinned against outputs of the
reference itself: ``tests/golden/make_golden.py`` imports the reference
modules in the build container, runs them on seeded inputs, and stores the
vectors under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
this file against those vectors.

Semantics that parity depends on:
  * every ``Block`` that starts with its in-place LeakyReLU activates the
    tensor it was handed, so the residual / skip-conv path sees the activated
    input and so does whoever else holds that tensor (architecture/gpsro/
    deeplab.py:100-121,132-143);
  * the low-level skip feature is therefore ``leaky_relu(block1_out)``
    (deeplab.py:241-243);
  * ``gradient_penalty`` is a constant (no graph) (deeplab_gan.py:98-114);
  * ``GANLoss.d_loss`` draws fake labels, real labels, then one swap uniform
    (utils/losses.py:150-159).
"""
from __future__ import annotations

import math
import zlib
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SLOPE = 0.2
BN_EPS = 1e-5
BN_MOMENTUM = 0.1

State = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------
# Topology tables (what the reference constructors build) -- deeplab.py:146-222
# ----------------------------------------------------------------------------
def xception_block_table(os: int = 16) -> List[dict]:
    """Per-block hyper-parameters of the modified aligned Xception
    (deeplab.py:153-213)."""
    if os == 16:
        b3_stride, mid_rate, exit_rates = 2, 1, (1, 2)
    elif os == 8:
        b3_stride, mid_rate, exit_rates = 1, 2, (2, 4)
    else:
        raise NotImplementedError
    tbl = [
        dict(name="block1", cin=128, cout=128, reps=2, stride=2, dil=1, start_relu=False, grow_first=True, is_last=False),
        dict(name="block2", cin=128, cout=256, reps=2, stride=2, dil=1, start_relu=True, grow_first=True, is_last=False),
        dict(name="block3", cin=256, cout=728, reps=2, stride=b3_stride, dil=1, start_relu=True, grow_first=True, is_last=True),
    ]
    for i in range(4, 20):
        tbl.append(dict(name=f"block{i}", cin=728, cout=728, reps=3, stride=1, dil=mid_rate, start_relu=True,
                        grow_first=True, is_last=False))
    tbl.append(dict(name="block20", cin=728, cout=1024, reps=2, stride=1, dil=exit_rates[0], start_relu=True,
                    grow_first=False, is_last=True))
    return tbl


def block_units(cfg: dict) -> List[dict]:
    """The ``rep`` Sequential of one Block as a list of units, with the index
    each sub-module has inside ``rep`` (deeplab.py:101-130).  Unit kinds:
    'relu', 'sep' (SeparableConv2d_same), 'norm'."""
    cin, cout, reps = cfg["cin"], cfg["cout"], cfg["reps"]
    units = []
    filters = cin
    if cfg["grow_first"]:
        units += [dict(kind="relu"), dict(kind="sep", cin=cin, cout=cout, stride=1, dil=cfg["dil"]),
                  dict(kind="norm", c=cout)]
        filters = cout
    for _ in range(reps - 1):
        units += [dict(kind="relu"), dict(kind="sep", cin=filters, cout=filters, stride=1, dil=cfg["dil"]),
                  dict(kind="norm", c=filters)]
    if not cfg["grow_first"]:
        units += [dict(kind="relu"), dict(kind="sep", cin=cin, cout=cout, stride=1, dil=cfg["dil"]),
                  dict(kind="norm", c=cout)]
    if not cfg["start_relu"]:
        units = units[1:]
    if cfg["stride"] != 1:
        units.append(dict(kind="sep", cin=cout, cout=cout, stride=2, dil=1))
    if cfg["stride"] == 1 and cfg["is_last"]:
        units.append(dict(kind="sep", cin=cout, cout=cout, stride=1, dil=1))
    for i, u in enumerate(units):
        u["idx"] = i
    return units


def ceil_div(a: int, b: int) -> int:
    return -(-a // b)


def out_hw16(h: int, w: int) -> Tuple[int, int]:
    """Spatial size of the os=16 bottleneck: four ceil-halvings."""
    for _ in range(4):
        h, w = ceil_div(h, 2), ceil_div(w, 2)
    return h, w


# ----------------------------------------------------------------------------
# Parameter specs / deterministic fills
# ----------------------------------------------------------------------------
def _norm_entries(prefix: str, c: int, norm: str) -> List[Tuple[str, Tuple[int, ...], str]]:
    if norm == "batch":
        return [(prefix + ".weight", (c,), "gamma"), (prefix + ".bias", (c,), "beta"),
                (prefix + ".running_mean", (c,), "rmean"), (prefix + ".running_var", (c,), "rvar"),
                (prefix + ".num_batches_tracked", (), "nbt")]
    return []


def xception_spec(prefix: str, cin: int, norm: str, os: int = 16) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, role) for every state_dict entry of Xception, in the
    reference's registration order (deeplab.py:178-222)."""
    s: List[Tuple[str, Tuple[int, ...], str]] = []
    s.append((prefix + "conv1.weight", (128, cin, 3, 3), "conv"))
    s += _norm_entries(prefix + "bn1", 128, norm)
    s.append((prefix + "conv2.weight", (128, 128, 3, 3), "conv"))
    s += _norm_entries(prefix + "bn2", 128, norm)
    for cfg in xception_block_table(os):
        bp = prefix + cfg["name"] + "."
        if cfg["cin"] != cfg["cout"] or cfg["stride"] != 1:
            s.append((bp + "skip.weight", (cfg["cout"], cfg["cin"], 1, 1), "conv"))
            s += _norm_entries(bp + "skipbn", cfg["cout"], norm)
        for u in block_units(cfg):
            up = bp + f"rep.{u['idx']}"
            if u["kind"] == "sep":
                s.append((up + ".conv1.weight", (u["cin"], 1, 3, 3), "conv"))
                s.append((up + ".pointwise.weight", (u["cout"], u["cin"], 1, 1), "conv"))
            elif u["kind"] == "norm":
                s += _norm_entries(up, u["c"], norm)
    for name, ci, co in (("3", 1024, 1536), ("4", 1536, 1536), ("5", 1536, 2048)):
        s.append((prefix + f"conv{name}.conv1.weight", (ci, 1, 3, 3), "conv"))
        s.append((prefix + f"conv{name}.pointwise.weight", (co, ci, 1, 1), "conv"))
        s += _norm_entries(prefix + f"bn{name}", co, norm)
    return s


def deeplab_spec(prefix: str, cin: int, cout: int, norm: str, os: int = 16, upsampler: str = "Interpolate"):
    """DeepLabv3_plus (deeplab.py:604-651) with the Interpolate (:363-369), Deconv or Deconv1x
    (:398-431, :465-487) upsampler; entries in the reference's state_dict order."""
    s = xception_spec(prefix + "xception_features.", cin, norm, os)
    for i in (1, 2, 3, 4):
        k = 1 if i == 1 else 3
        s.append((prefix + f"aspp{i}.atrous_convolution.weight", (256, 2048, k, k), "conv"))
        s += _norm_entries(prefix + f"aspp{i}.bn", 256, norm)
    s.append((prefix + "global_avg_pool.1.weight", (256, 2048, 1, 1), "conv"))
    s += _norm_entries(prefix + "global_avg_pool.2", 256, norm)
    s.append((prefix + "conv1.weight", (256, 1280, 1, 1), "conv"))
    s += _norm_entries(prefix + "bn1", 256, norm)
    s.append((prefix + "conv2.weight", (48, 128, 1, 1), "conv"))
    s += _norm_entries(prefix + "bn2", 48, norm)
    if upsampler == "Interpolate":
        up = prefix + "upsample.last_conv."
        s.append((up + "0.weight", (256, 304, 3, 3), "conv"))
        s += _norm_entries(up + "1", 256, norm)
        s.append((up + "3.weight", (256, 256, 3, 3), "conv"))
        s += _norm_entries(up + "4", 256, norm)
        s.append((up + "6.weight", (cout, 256, 1, 1), "conv"))
        s.append((up + "6.bias", (cout,), "bias"))
        return s
    assert upsampler in ("Deconv", "Deconv1x"), upsampler
    n_up = 128 if upsampler == "Deconv1x" else cout
    up = prefix + "upsample."
    # nn.ConvTranspose2d weights are [Cin, Cout, k, k]
    s.append((up + "deconv1.0.weight", (256, 256, 3, 3), "conv"))
    s += _norm_entries(up + "deconv1.1", 256, norm)
    s.append((up + "deconv2.0.weight", (256, 256, 3, 3), "conv"))
    s += _norm_entries(up + "deconv2.1", 256, norm)
    s.append((up + "conv1.0.weight", (256, 304, 3, 3), "conv"))
    s += _norm_entries(up + "conv1.1", 256, norm)
    s.append((up + "conv1.3.weight", (256, 256, 3, 3), "conv"))
    s += _norm_entries(up + "conv1.4", 256, norm)
    s.append((up + "conv1.6.weight", (256, 256, 1, 1), "conv"))
    s.append((up + "conv1.6.bias", (256,), "bias"))
    s.append((up + "deconv3.0.weight", (256, 128, 3, 3), "conv"))
    s += _norm_entries(up + "deconv3.1", 128, norm)
    s.append((up + "last_deconv.0.weight", (128, n_up, 3, 3), "conv"))
    if upsampler == "Deconv1x":
        ex = prefix + "upsample_extension."
        s += _norm_entries(ex + "init_norm.0", 128, norm)
        s.append((ex + "conv1.0.weight", (64, cin, 3, 3), "conv"))
        s += _norm_entries(ex + "conv1.1", 64, norm)
        s.append((ex + "conv1.3.weight", (128, 64, 3, 3), "conv"))
        s += _norm_entries(ex + "conv1.4", 128, norm)
        s.append((ex + "conv2.0.weight", (64, 256, 3, 3), "conv"))
        s += _norm_entries(ex + "conv2.1", 64, norm)
        s.append((ex + "conv2.3.weight", (cout, 64, 3, 3), "conv"))
    return s


def generator_spec(cin: int, cout: int, noise_dims: int, norm: str, os: int = 16, upsampler: str = "Interpolate"):
    """Generator = noise concat + DeepLabv3_plus under key prefix 'model.'
    (deeplab_gan.py:64-94)."""
    return deeplab_spec("model.", cin + noise_dims, cout, norm, os, upsampler)


def discriminator_spec(cin: int, h: int, w: int, norm: str, os: int = 16):
    """Discriminator = Xception + Linear(2048*h16*w16, 1) (deeplab_gan.py:12-39;
    the reference hard-codes 12288 = 2048*2*3 for the 19x37 grid, :21)."""
    s = xception_spec("xception_features.", cin, norm, os)
    h16, w16 = out_hw16(h, w)
    s.append(("linear.weight", (1, 2048 * h16 * w16), "linear"))
    s.append(("linear.bias", (1,), "bias"))
    return s


def fill_state(spec, seed: int = 0) -> State:
    """Deterministic fill keyed by the entry name (crc32(name) ^ seed), so any
    implementation can rebuild the same state without sharing an RNG stream.
    Conv/linear weights ~ N(0, 1/fan_in) * 1.2 (keeps activations O(1) through
    65 layers), gamma ~ U(0.8, 1.2), beta ~ N(0, 0.1), running stats (0.1 N, U(0.5,1.5))."""
    st: State = {}
    for key, shape, role in spec:
        rng = np.random.default_rng((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0xFFFFFFFF)
        if role in ("conv", "linear"):
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            v = rng.standard_normal(shape, dtype=np.float32) * np.float32(1.2 / math.sqrt(fan_in))
        elif role == "gamma":
            v = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        elif role in ("beta", "bias"):
            v = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        elif role == "rmean":
            v = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        elif role == "rvar":
            v = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif role == "nbt":
            st[key] = torch.zeros((), dtype=torch.long)
            continue
        else:
            raise ValueError(role)
        st[key] = torch.from_numpy(np.ascontiguousarray(v))
    return st


def trainable_keys(spec) -> List[str]:
    return [k for k, _, role in spec if role in ("conv", "linear", "gamma", "beta", "bias")]


def synthetic_fields(n: int, c: int, h: int, w: int, seed: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """inputs ~ N(0,1), label = input + 0.1 N(0,1); numpy PCG64
    keyed by seed so every implementation regenerates identical fields."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, c, h, w), dtype=np.float32)
    y = x + np.float32(0.1) * rng.standard_normal((n, c, h, w), dtype=np.float32)
    return torch.from_numpy(x), torch.from_numpy(y)


# ----------------------------------------------------------------------------
# Operators
# ----------------------------------------------------------------------------
def lrelu(x: torch.Tensor) -> torch.Tensor:
    """nn.LeakyReLU(0.2) (deeplab.py:100,180,334)."""
    return F.leaky_relu(x, SLOPE)


class NormCtx:
    """Which normaliser the net was built with and whether it trains.
    'batch' = nn.BatchNorm2d defaults, 'instance' = nn.InstanceNorm2d defaults
    (no affine, no running stats), 'identity' = pass-through."""

    def __init__(self, kind: str = "batch", training: bool = True, update_stats: bool = True, bf16: bool = False):
        assert kind in ("batch", "instance", "identity")
        self.kind, self.training, self.update_stats = kind, training, update_stats
        # bf16=True emulates the storage roundings of the BG_BF16 kernel path (activations
        # and conv weights held in bfloat16, all arithmetic and statistics in fp32) so the
        # GPU bf16 forward can be checked against a CPU evaluation with the same rounding
        # points.  Rounding is straight-through for autograd.
        self.bf16 = bf16

    def q(self, x: torch.Tensor) -> torch.Tensor:
        if not self.bf16:
            return x
        return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def norm(P: State, key: str, x: torch.Tensor, ctx: NormCtx) -> torch.Tensor:
    """Normaliser WITHOUT storage rounding: callers round after the fused
    norm(+residual)(+activation), like the kernels do."""
    if ctx.kind == "identity":
        return x
    if ctx.kind == "instance":
        # InstanceNorm2d defaults; a single-element instance normalises to 0
        # (mean = x, var = 0): the unchecked arithmetic of the reference's
        # torch 1.8 era.  torch >= 1.9 raises instead, so
        # compute it directly.
        sp = tuple(range(2, x.dim()))  # spatial dims: (2,3) for 2-D nets, (2,3,4) for the 3-D ones
        m = x.mean(dim=sp, keepdim=True)
        v = x.var(dim=sp, unbiased=False, keepdim=True)
        return (x - m) / torch.sqrt(v + BN_EPS)
    g, b = P[key + ".weight"], P[key + ".bias"]
    rm, rv = P[key + ".running_mean"], P[key + ".running_var"]
    if ctx.training:
        if ctx.update_stats:
            P[key + ".num_batches_tracked"] += 1
            return F.batch_norm(x, rm, rv, g, b, True, BN_MOMENTUM, BN_EPS)
        return F.batch_norm(x, None, None, g, b, True, BN_MOMENTUM, BN_EPS)
    return F.batch_norm(x, rm, rv, g, b, False, BN_MOMENTUM, BN_EPS)


def sepconv_same(P: State, key: str, x: torch.Tensor, stride: int, dil: int, ctx: Optional[NormCtx] = None) -> torch.Tensor:
    """SeparableConv2d_same: zero-pad by k_eff-1 split (beg=total//2, end=rest),
    depthwise 3x3 (no bias), pointwise 1x1 (no bias), nothing in between
    (deeplab.py:66-87)."""
    q = ctx.q if ctx is not None else (lambda t: t)
    wd, wp = P[key + ".conv1.weight"], P[key + ".pointwise.weight"]
    k = wd.shape[-1]
    keff = k + (k - 1) * (dil - 1)
    tot = keff - 1
    beg = tot // 2
    end = tot - beg
    x = F.pad(x, (beg, end, beg, end))
    x = q(F.conv2d(x, q(wd), None, stride, 0, dil, groups=wd.shape[0]))
    return q(F.conv2d(x, q(wp)))


def block(P: State, bp: str, cfg: dict, x: torch.Tensor, ctx: NormCtx) -> torch.Tensor:
    """One Block (deeplab.py:90-143).  ``x`` is NOT mutated here; instead the
    aliasing of the reference's in-place first LeakyReLU is made explicit: when
    the block starts with it, both the unit chain and the skip path consume
    ``leaky_relu(x)``.  The returned sum is not storage-rounded (ctx.bf16): its
    consumer rounds after its own fused activation, as the kernels do."""
    q = ctx.q
    a = q(lrelu(x)) if cfg["start_relu"] else x
    h = a
    units = block_units(cfg)
    i = 1 if cfg["start_relu"] else 0
    tail = None
    while i < len(units):
        u = units[i]
        if u["kind"] == "sep":
            h = sepconv_same(P, bp + f"rep.{u['idx']}", h, u["stride"], u["dil"], ctx)
            i += 1
        elif u["kind"] == "relu":
            h = q(lrelu(h))
            i += 1
        else:
            h = norm(P, bp + f"rep.{u['idx']}", h, ctx)
            if i == len(units) - 1:
                tail = h
                break
            if units[i + 1]["kind"] == "relu":  # fused normalise + activate, one rounding
                h = q(lrelu(h))
                i += 2
            else:
                h = q(h)
                i += 1
    if cfg["cin"] != cfg["cout"] or cfg["stride"] != 1:
        s = q(F.conv2d(a, q(P[bp + "skip.weight"]), None, cfg["stride"]))
        s = q(norm(P, bp + "skipbn", s, ctx))
    else:
        s = a
    return h + s


def xception(P: State, prefix: str, x: torch.Tensor, ctx: NormCtx, os: int = 16):
    """Xception.forward (deeplab.py:231-278) -> (features, low_level_feat)."""
    q = ctx.q
    x = q(F.conv2d(q(x), q(P[prefix + "conv1.weight"]), None, 2, 1))
    x = q(lrelu(norm(P, prefix + "bn1", x, ctx)))
    x = q(F.conv2d(x, q(P[prefix + "conv2.weight"]), None, 1, 1))
    x = q(lrelu(norm(P, prefix + "bn2", x, ctx)))
    low = None
    for cfg in xception_block_table(os):
        x = block(P, prefix + cfg["name"] + ".", cfg, x, ctx)
        if cfg["name"] == "block1":
            # block2's in-place LeakyReLU later activates this very tensor
            # (deeplab.py:242 aliases it), so the skip feature is activated.
            low = q(lrelu(x))
    x = q(x)  # block20's sum feeds conv3 without an activation in between
    rate = 2 if os == 16 else 4
    for name in ("3", "4", "5"):
        x = sepconv_same(P, prefix + f"conv{name}", x, 1, rate, ctx)
        x = q(lrelu(norm(P, prefix + f"bn{name}", x, ctx)))
    return x, low


def bilinear_ac(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """F.interpolate(mode='bilinear', align_corners=True) (deeplab.py:375,379,663)."""
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


def _deconv_unit(P: State, key: str, x: torch.Tensor, ctx: NormCtx, pad: int, out_pad, pool: bool = True):
    """ConvTranspose2d(3, stride 2, no bias) -> normaliser -> AvgPool2d(2, 1, 0) -> LeakyReLU
    (one nn.Sequential of DeconvUpsampler, deeplab.py:406-430)."""
    q = ctx.q
    x = q(F.conv_transpose2d(x, q(P[key + ".0.weight"]), None, 2, pad, out_pad))
    x = q(norm(P, key + ".1", x, ctx))
    if pool:
        x = q(F.avg_pool2d(x, 2, 1, 0))
    return q(lrelu(x))


def deconv_upsampler(P: State, up: str, x: torch.Tensor, low: torch.Tensor, ctx: NormCtx) -> torch.Tensor:
    """DeconvUpsampler.forward (deeplab.py:436-444)."""
    q = ctx.q
    x = _deconv_unit(P, up + "deconv1", x, ctx, 1, (0, 1))
    x = _deconv_unit(P, up + "deconv2", x, ctx, 0, (1, 0))
    x = torch.cat((x, low), dim=1)
    x = q(lrelu(norm(P, up + "conv1.1", q(F.conv2d(x, q(P[up + "conv1.0.weight"]), None, 1, 1)), ctx)))
    x = q(lrelu(norm(P, up + "conv1.4", q(F.conv2d(x, q(P[up + "conv1.3.weight"]), None, 1, 1)), ctx)))
    x = q(F.conv2d(x, q(P[up + "conv1.6.weight"]), P[up + "conv1.6.bias"]))
    x = _deconv_unit(P, up + "deconv3", x, ctx, 1, (1, 0))
    return q(F.conv_transpose2d(x, q(P[up + "last_deconv.0.weight"]), None, 2, 1, (1, 1)))


def upsampler_extension(P: State, ex: str, x_in: torch.Tensor, x: torch.Tensor, ctx: NormCtx) -> torch.Tensor:
    """UpsamplerExtension.forward (deeplab.py:489-498)."""
    q = ctx.q
    skip = q(lrelu(norm(P, ex + "conv1.1", q(F.conv2d(x_in, q(P[ex + "conv1.0.weight"]), None, 1, 1)), ctx)))
    skip = q(lrelu(norm(P, ex + "conv1.4", q(F.conv2d(skip, q(P[ex + "conv1.3.weight"]), None, 1, 1)), ctx)))
    x = q(norm(P, ex + "init_norm.0", x, ctx))
    x = q(lrelu(q(F.avg_pool2d(x, 2, 1, 1))))
    x = torch.cat((x, skip), dim=1)
    x = q(lrelu(norm(P, ex + "conv2.1", q(F.conv2d(x, q(P[ex + "conv2.0.weight"]), None, 1, 1)), ctx)))
    return q(F.conv2d(x, q(P[ex + "conv2.3.weight"]), None, 1, 1))


def deeplab(P: State, prefix: str, x_in: torch.Tensor, ctx: NormCtx, os: int = 16) -> torch.Tensor:
    """DeepLabv3_plus.forward (deeplab.py:654-684); the upsampler in use is read off the state's keys:
    Interpolate (:374-381), Deconv (+ final AvgPool2d(2,1,1), :681-682) or Deconv1x (+ extension)."""
    q = ctx.q
    rates = [1, 6, 12, 18] if os == 16 else [1, 12, 24, 36]
    x, low = xception(P, prefix + "xception_features.", x_in, ctx, os)
    branches = []
    for i, r in zip((1, 2, 3, 4), rates):
        w = q(P[prefix + f"aspp{i}.atrous_convolution.weight"])
        b = q(F.conv2d(x, w, None, 1, 0 if r == 1 else r, r))
        branches.append(q(lrelu(norm(P, prefix + f"aspp{i}.bn", b, ctx))))
    g = q(x.mean(dim=(2, 3), keepdim=True))
    g = q(F.conv2d(g, q(P[prefix + "global_avg_pool.1.weight"])))
    g = q(lrelu(norm(P, prefix + "global_avg_pool.2", g, ctx)))
    branches.append(q(bilinear_ac(g, branches[-1].shape[2:])))
    x = torch.cat(branches, dim=1)
    x = q(lrelu(norm(P, prefix + "bn1", q(F.conv2d(x, q(P[prefix + "conv1.weight"]))), ctx)))
    low = q(lrelu(norm(P, prefix + "bn2", q(F.conv2d(low, q(P[prefix + "conv2.weight"]))), ctx)))
    H, W = x_in.shape[2], x_in.shape[3]
    if prefix + "upsample.deconv1.0.weight" in P:
        x = deconv_upsampler(P, prefix + "upsample.", x, low, ctx)
        if prefix + "upsample_extension.conv1.0.weight" in P:
            return upsampler_extension(P, prefix + "upsample_extension.", x_in, x, ctx)
        return q(F.avg_pool2d(x, 2, 1, 1))
    x = q(bilinear_ac(x, (ceil_div(H, 4), ceil_div(W, 4))))
    x = torch.cat((x, low), dim=1)
    up = prefix + "upsample.last_conv."
    x = q(lrelu(norm(P, up + "1", q(F.conv2d(x, q(P[up + "0.weight"]), None, 1, 1)), ctx)))
    x = q(lrelu(norm(P, up + "4", q(F.conv2d(x, q(P[up + "3.weight"]), None, 1, 1)), ctx)))
    x = q(F.conv2d(x, q(P[up + "6.weight"]), P[up + "6.bias"]))
    return bilinear_ac(x, (H, W))  # the last resize writes fp32


def generator(P: State, x: torch.Tensor, ctx: NormCtx, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Generator.forward (deeplab_gan.py:85-94); ``noise`` is the already drawn
    [N, nd, H, W] tensor (None when noise_dimensions == 0)."""
    if noise is not None:
        x = torch.cat((x, noise), dim=1)
    return deeplab(P, "model.", x, ctx)


def discriminator(P: State, x: torch.Tensor, ctx: NormCtx):
    """Discriminator.forward (deeplab_gan.py:28-39) -> (logits, sigmoid(logits))."""
    f, _ = xception(P, "xception_features.", x, ctx)
    logits = F.linear(f.reshape(f.shape[0], -1), P["linear.weight"], P["linear.bias"])
    return logits, torch.sigmoid(logits)


# ----------------------------------------------------------------------------
# Losses
# ----------------------------------------------------------------------------
def bce_logits(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """nn.BCEWithLogitsLoss(): mean of max(x,0) - x*y + log(1+exp(-|x|)) (losses.py:138)."""
    return (x.clamp(min=0) - x * y + torch.log1p(torch.exp(-x.abs()))).mean()


def draw_d_labels(batch: int, gen: Optional[torch.Generator] = None):
    """The three host draws of GANLoss.d_loss in the reference's order:
    fake ~ U(0,.2)[N,1], real ~ U(.8,1)[N,1], swap ~ U(0,1) scalar
    (losses.py:152-156).  torch.distributions.Uniform.rsample is
    low + rand(shape)*(high-low) on the global generator."""
    kw = {} if gen is None else {"generator": gen}
    t = lambda v: torch.tensor(v, dtype=torch.float32)  # Uniform keeps low/high as fp32 tensors

    def u(lo, hi, shape):
        return t(lo) + torch.rand(shape, **kw) * (t(hi) - t(lo))

    fake = u(0.0, 0.2, (batch, 1))
    real = u(0.8, 1.0, (batch, 1))
    swap = bool(u(0.0, 1.0, ()) < 0.05)
    return fake, real, swap


def gan_d_loss(mode: str, logits_real, logits_fake, label_fake=None, label_real=None, swap=False):
    """GANLoss.d_loss (losses.py:150-164)."""
    if mode == "ModifiedMinMax":
        if swap:
            return 0.5 * (bce_logits(logits_fake, label_real) + bce_logits(logits_real, label_fake))
        return 0.5 * (bce_logits(logits_fake, label_fake) + bce_logits(logits_real, label_real))
    if mode == "Wasserstein":
        return (logits_fake - logits_real).mean()
    raise NotImplementedError(mode)


def gan_g_loss(mode: str, logits_fake):
    """GANLoss.g_loss (losses.py:167-172)."""
    if mode == "ModifiedMinMax":
        return bce_logits(logits_fake, torch.ones_like(logits_fake))
    if mode == "Wasserstein":
        return -1.0 * logits_fake.mean()
    raise NotImplementedError(mode)


def l1_weighted(pred, target, weights, normalize=False, eps=1e-8):
    """L1LossWeighted (losses.py:101-112)."""
    a = (pred - target).abs() * weights
    return a.sum() / (weights.sum() + eps) if normalize else a.mean()


def gradient_penalty(P: State, fake: torch.Tensor, real: torch.Tensor, eta: torch.Tensor, ctx: NormCtx):
    """gradient_penalty (deeplab_gan.py:98-114): x^ = eta*fake + (1-eta)*real,
    grad of sum(logits) w.r.t. x^, mean over N,H,W of (||grad||_2 over C - 1)^2.
    Returned detached: the reference builds no graph (create_graph is commented
    out), so it shifts the loss value and contributes no gradient."""
    xi = (eta * fake.detach() + (1.0 - eta) * real.detach()).requires_grad_(True)
    logits, _ = discriminator(P, xi, ctx)
    (g,) = torch.autograd.grad(logits, xi, torch.ones_like(logits))
    return ((g.norm(2, dim=1) - 1.0) ** 2).mean().detach()


# ----------------------------------------------------------------------------
# Optimiser (torch.optim.Adam / AdamW semantics, parsing_helpers.py:8-12)
# ----------------------------------------------------------------------------
class Adam:
    def __init__(self, keys: List[str], lr=1e-4, eps=1e-8, weight_decay=1e-5, betas=(0.9, 0.999), decoupled=False):
        self.keys, self.lr, self.eps, self.wd, self.betas, self.decoupled = keys, lr, eps, weight_decay, betas, decoupled
        self.t = 0
        self.m: State = {}
        self.v: State = {}

    @torch.no_grad()
    def step(self, P: State, grads: State):
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for k in self.keys:
            g = grads.get(k)
            if g is None:
                continue
            p = P[k]
            if self.decoupled:
                p.mul_(1 - self.lr * self.wd)
            elif self.wd != 0:
                g = g + self.wd * p
            if k not in self.m:
                self.m[k] = torch.zeros_like(p)
                self.v[k] = torch.zeros_like(p)
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


class Lamb:
    """apex.optimizers.FusedLAMB as utils/parsing_helpers.py:13-14 constructs it (lr, eps, weight_decay; apex's other
    defaults).  apex is a dependency that is NOT in the reference tree (NVIDIA apex of nvcr.io/nvidia/pytorch:20.12-py3,
    docker/Dockerfile:1), so this restates its published algorithm -- apex/optimizers/fused_lamb.py (global gradient norm
    over all parameters, one step count per group) and csrc/multi_tensor_lamb.cu (LAMBStage1Functor / LAMBStage2Functor):
    PARITY UNPINNED against apex itself; pinned where it can be: without weight decay and below the clipping norm the step
    IS torch.optim.Adam's (tests/test_oracle_golden.py::test_lamb_*)."""

    def __init__(self, keys: List[str], lr=1e-3, eps=1e-6, weight_decay=0.01, betas=(0.9, 0.999), bias_correction=True,
                 adam_w_mode=True, grad_averaging=True, max_grad_norm=1.0, use_nvlamb=False):
        self.keys, self.lr, self.eps, self.wd, self.betas = keys, lr, eps, weight_decay, betas
        self.bias_correction, self.adam_w_mode, self.grad_averaging = bias_correction, adam_w_mode, grad_averaging
        self.max_grad_norm, self.use_nvlamb = max_grad_norm, use_nvlamb
        self.t = 0
        self.m: State = {}
        self.v: State = {}

    @torch.no_grad()
    def step(self, P: State, grads: State):
        self.t += 1
        b1, b2 = self.betas
        b3 = 1 - b1 if self.grad_averaging else 1.0
        bc1, bc2 = (1 - b1 ** self.t, 1 - b2 ** self.t) if self.bias_correction else (1.0, 1.0)
        keys = [k for k in self.keys if grads.get(k) is not None]
        gnorm = math.sqrt(sum(float((grads[k].double() ** 2).sum()) for k in keys))
        clip = gnorm / self.max_grad_norm if (self.max_grad_norm > 0 and gnorm > self.max_grad_norm) else 1.0
        for k in keys:
            p = P[k]
            sg = grads[k] / clip
            if not self.adam_w_mode:
                sg = sg + self.wd * p
            if k not in self.m:
                self.m[k] = torch.zeros_like(p)
                self.v[k] = torch.zeros_like(p)
            self.m[k].mul_(b1).add_(sg, alpha=b3)
            self.v[k].mul_(b2).addcmul_(sg, sg, value=1 - b2)
            u = (self.m[k] / bc1) / ((self.v[k] / bc2).sqrt() + self.eps)
            if self.adam_w_mode:
                u = u + self.wd * p
            ratio = self.lr
            if self.use_nvlamb or self.wd != 0:
                pn, un = float(p.double().norm()), float(u.double().norm())
                if pn != 0.0 and un != 0.0:
                    ratio = self.lr * pn / un
            p.add_(u, alpha=-ratio)



# ----------------------------------------------------------------------------
# The training step (train_gan.py:244-298)
# ----------------------------------------------------------------------------
class GANStep:
    """D-step then G-step on one batch, as the reference loop executes them
    (train_gan.py:250-298), with the results-neutral waste removed: the D-step
    does not back-propagate into G and the G-step does not keep D weight
    gradients.  Both G forwards and all three D forwards run
    in train mode, so BatchNorm running statistics receive the same number of
    momentum updates as in the reference."""

    def __init__(self, PG: State, PD: State, g_keys, d_keys, norm_kind="batch", loss_mode="ModifiedMinMax",
                 lr_g=1e-4, lr_d=1e-4, eps=1e-8, weight_decay=1e-5, w_gan=1.0, w_reg=1.0, w_gp=10.0,
                 noise_dims=0, noise_type="Uniform"):
        self.PG, self.PD, self.g_keys, self.d_keys = PG, PD, g_keys, d_keys
        self.ctx = NormCtx(norm_kind, True)
        self.mode = loss_mode
        self.g_opt = Adam(g_keys, lr_g, eps, weight_decay)
        self.d_opt = Adam(d_keys, lr_d, eps, weight_decay)
        self.w_gan, self.w_reg, self.w_gp = w_gan, w_reg, w_gp
        self.noise_dims, self.noise_type = noise_dims, noise_type

    def _noise(self, x):
        if self.noise_dims == 0:
            return None
        shape = (x.shape[0], self.noise_dims, x.shape[2], x.shape[3])
        return torch.rand(shape) if self.noise_type == "Uniform" else torch.randn(shape)

    def _leaves(self, P, keys):
        Q = dict(P)
        for k in keys:
            Q[k] = P[k].detach().requires_grad_(True)
        return Q

    def d_step(self, inputs, real, labels=None, eta=None):
        with torch.no_grad():
            fake = generator(self.PG, inputs, self.ctx, self._noise(inputs))
        Q = self._leaves(self.PD, self.d_keys)
        lr_, _ = discriminator(Q, real, self.ctx)
        lf_, _ = discriminator(Q, fake, self.ctx)
        if self.mode == "ModifiedMinMax":
            lab_f, lab_r, swap = labels if labels is not None else draw_d_labels(real.shape[0])
            loss = gan_d_loss(self.mode, lr_, lf_, lab_f, lab_r, swap)
        else:
            loss = gan_d_loss(self.mode, lr_, lf_)
            if eta is None:
                eta = torch.rand((real.shape[0], 1, 1, 1))
            loss = loss + self.w_gp * gradient_penalty(Q, fake, real, eta, self.ctx)
        grads = torch.autograd.grad(loss, [Q[k] for k in self.d_keys], allow_unused=True)
        self._sync_buffers(Q, self.PD)
        self.d_opt.step(self.PD, {k: g for k, g in zip(self.d_keys, grads) if g is not None})
        return float(loss.detach())

    def g_step(self, inputs, real, masks=None, warmup=False):
        Q = self._leaves(self.PG, self.g_keys)
        fake = generator(Q, inputs, self.ctx, self._noise(inputs))
        lf_, _ = discriminator(self.PD, fake, self.ctx)
        gan = gan_g_loss(self.mode, lf_)
        reg = (fake - real).abs().mean() if masks is None else l1_weighted(fake, real, masks)
        loss = reg if warmup else self.w_gan * gan + self.w_reg * reg
        grads = torch.autograd.grad(loss, [Q[k] for k in self.g_keys], allow_unused=True)
        self._sync_buffers(Q, self.PG)
        self.g_opt.step(self.PG, {k: g for k, g in zip(self.g_keys, grads) if g is not None})
        return float(loss.detach()), fake.detach()

    @staticmethod
    def _sync_buffers(Q, P):
        # running stats were updated in place on the shared buffer tensors
        for k in P:
            if Q[k] is not P[k] and not Q[k].requires_grad:
                P[k] = Q[k]

    def step(self, inputs, real, labels=None, eta=None, masks=None):
        d = self.d_step(inputs, real, labels, eta)
        g, _ = self.g_step(inputs, real, masks)
        return d, g
