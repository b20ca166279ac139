"""DeepLabv3+ / modified aligned Xception on the MI355X kernels.

Host-side mirror of the reference's ``architecture/gpsro/deeplab.py``: same
class names, constructor arguments and state_dict keys (so checkpoints and call
sites interchange), but ``forward`` is a fused NHWC schedule over the C ABI
(bias_gan_amd.ops) instead of torch.nn ops:

  * conv -> norm -> LeakyReLU chains run as  conv kernel + (stats, fused
    normalise+activate) ;
  * the Block residual ``x += skip`` and the NEXT block's leading in-place
    LeakyReLU are one fused normalise+add+activate pass.  The reference's
    in-place activation also rewrites the tensor its caller still holds
    (deeplab.py:100-121,141, :242), so here every block hands its successor an
    already-activated tensor and the skip paths / low-level feature read that
    same tensor -- identical values, no aliasing tricks;
  * ``fixed_padding`` (deeplab.py:66-72) is index arithmetic inside the depthwise
    kernel, no padded copy;
  * torch.cat is a strided copy into one NHWC buffer.

Module boundaries stay NCHW fp32 like the reference; inside, activations are
NHWC in the compute dtype (bf16 by default, fp32 for parity runs).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ... import ops
from ...runtime import BGModule, StatsPool, pad_to, vec_of


class Identity(nn.Module):
    """Pass-through normaliser (architecture/era/deeplab_xception.py:9-14), used
    for --disable_batchnorm."""

    def __init__(self, channels=None):
        super().__init__()

    def forward(self, x):
        return x


# ------------------------------------------------------------------ parameter holders
def _conv_layout(p, vec):
    k, c, r, s = p.shape
    return "conv", (pad_to(k, vec), r, s, pad_to(c, vec))


def _dw_layout(p, vec):
    c, _, r, s = p.shape
    return "dw", (r, s, pad_to(c, vec))


def _vec_layout(p, vec):
    return "vec", (pad_to(p.shape[0], vec),)


class Conv2d(BGModule):
    """Parameter container + launcher with nn.Conv2d's attribute names
    (weight [Cout, Cin/groups, k, k], optional bias) and default initialisation."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        assert groups in (1, in_channels), "only dense and depthwise convolutions occur on this path"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size, kernel_size)
        self.stride, self.padding, self.dilation, self.groups = (stride,) * 2, (padding,) * 2, (dilation,) * 2, groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))  # nn.Conv2d.reset_parameters
        if bias:
            bound = 1.0 / math.sqrt(self.weight[0].numel())
            nn.init.uniform_(self.bias, -bound, bound)
        self._bg_param_layout = {"weight": _dw_layout if groups > 1 else _conv_layout, "bias": _vec_layout}

    def forward(self, x, stats=None, with_bias=True):
        """x: NHWC activation (padded channels).  stats: see ops.Conv2dFn.  with_bias=False leaves the bias to the
        caller (the partial convolution adds it after its mask ratio)."""
        a = self.arena()
        ws = a.by_param[id(self.weight)]
        if self.groups > 1:
            return ops.DwConv3x3Fn.apply(x, self.weight, a, ws, self.stride[0], self.dilation[0])
        if self.bias is None or not with_bias:
            y = ops.Conv2dFn.apply(x, self.weight, None, a, ws, None, self.stride[0], self.padding[0], self.dilation[0], stats)
            if getattr(a, "fp8", False):
                ops.fp8_tag_output(y, a, ws, *self.kernel_size)
            return y
        return ops.Conv2dFn.apply(x, self.weight, self.bias, a, ws, a.by_param[id(self.bias)], self.stride[0], self.padding[0],
                                  self.dilation[0], stats)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size[0]}, s={self.stride[0]}, groups={self.groups}"


class ConvTranspose2d(BGModule):
    """nn.ConvTranspose2d(groups=1, bias=False) parameter container (weight [Cin, Cout, k, k]).  The
    arena stores it as the dense convolution it is the adjoint of ([K = Cin, C = Cout])."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=(0, 0), bias=False):
        super().__init__()
        assert not bias, "the transposed convolutions on this path have no bias"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size, kernel_size)
        self.stride, self.padding = (stride,) * 2, (padding,) * 2
        self.output_padding = tuple(output_padding)
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, kernel_size, kernel_size))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self.bias = None
        self._bg_param_layout = {"weight": _conv_layout}

    def forward(self, x):
        a = self.arena()
        return ops.ConvTranspose2dFn.apply(x, self.weight, a, a.by_param[id(self.weight)], self.stride[0], self.padding[0],
                                           self.output_padding)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size[0]}, s={self.stride[0]}, op={self.output_padding}"


def _norm_kind(m: nn.Module) -> str:
    if isinstance(m, nn.BatchNorm2d):
        if not (m.affine and m.track_running_stats):
            raise NotImplementedError("BatchNorm2d must keep its defaults (affine, running stats)")
        return "batch"
    if isinstance(m, nn.InstanceNorm2d):
        if m.affine or m.track_running_stats:
            raise NotImplementedError("InstanceNorm2d must keep its defaults (no affine, no running stats)")
        return "instance"
    if isinstance(m, (Identity, nn.Identity)):
        return "identity"
    raise NotImplementedError(f"normalizer {type(m).__name__} is not supported on the HIP path "
                              "(BatchNorm2d, InstanceNorm2d and Identity are)")


import os as _os
_FUSED_STATS = _os.environ.get("BGAMD_NO_FUSED_STATS") is None  # A/B switch


_FUSED_DW = _os.environ.get("BGAMD_NO_FUSED_DW") is None  # A/B switch: BatchNorm + LeakyReLU inside the next depthwise conv


def norm_dw_fusable(m: nn.Module, nxt) -> bool:
    """[norm m -> LeakyReLU -> SeparableConv2d_same nxt]: can the depthwise kernel of `nxt` apply m's affine and the
    activation to its input itself (ops.NormActDwConvFn)?  Training-mode BatchNorm2d, stride 1, dilation 1 or 2."""
    return (_FUSED_DW and isinstance(m, nn.BatchNorm2d) and m.training and m.affine and m.track_running_stats
            and isinstance(nxt, SeparableConv2d_same) and nxt.conv1.stride[0] == 1 and nxt.conv1.dilation[0] in (1, 2))


def norm_act_dw(owner: BGModule, m: nn.BatchNorm2d, x, stats, dw: "Conv2d"):
    """dw( LeakyReLU( m(x) ) ) in one pass over x (see norm_dw_fusable)."""
    a = owner.arena()
    groups = ops.current_bn_groups()
    m.__dict__["_bg_nbt_pending"] = m.__dict__.get("_bg_nbt_pending", 0) + groups * ops.current_bn_repeat()
    mom = 0.1 if m.momentum is None else float(m.momentum)
    if ops.current_bn_repeat() > 1:
        mom = 1.0 - (1.0 - mom) ** ops.current_bn_repeat()
    return ops.NormActDwConvFn.apply(x, m.weight, m.bias, dw.weight, a, a.by_param[id(m.weight)], a.by_param[id(m.bias)],
                                     a.by_param[id(dw.weight)], m.running_mean, m.running_var, True, float(m.eps), mom,
                                     stats, groups, dw.dilation[0])


def conv_norm(owner: BGModule, conv, m: nn.Module, x, res=None, act=False, skip_dw=False, defer_norm=False, offer_tail=False):
    """conv (Conv2d or SeparableConv2d_same) -> norm (+ residual) (+ LeakyReLU).  When the
    normaliser is a training-mode BatchNorm2d its batch statistics come out of the
    convolution's epilogue instead of a separate pass over the conv output.
    skip_dw: x is already the depthwise output of `conv` (norm_act_dw ran it); defer_norm: return
    (raw conv output, its statistics) for norm_act_dw instead of applying the norm."""
    dense = conv.pointwise if isinstance(conv, SeparableConv2d_same) else conv
    stats = None
    if isinstance(m, nn.BatchNorm2d) and m.training and dense.bias is None and _FUSED_STATS:
        groups = ops.current_bn_groups()
        n, h, w, _ = x.shape
        if isinstance(conv, SeparableConv2d_same):
            s = conv.conv1.stride[0]
            ho, wo = -(-h // s), -(-w // s)
        else:
            k, s, p, d = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
            ho, wo = (h + 2 * p - d * (k - 1) - 1) // s + 1, (w + 2 * p - d * (k - 1) - 1) // s + 1
        # the epilogue attributes whole 128-pixel tiles to a statistic group
        if groups == 1 or (n % groups == 0 and (n // groups * ho * wo) % 128 == 0):
            kp = owner.arena().by_param[id(dense.weight)].phys_shape[0]
            stats = StatsPool.get(x.device).take(2, groups, kp)
    if skip_dw:
        y = conv.pointwise(x, stats)
    else:
        y = conv(x, stats) if stats is not None else conv(x)
    if defer_norm:
        return y, stats
    return apply_norm(owner, m, y, res=res, act=act, stats=stats, offer_tail=offer_tail)


def apply_norm(owner: BGModule, m: nn.Module, x, res=None, act=False, stats=None, offer_tail=False):
    """norm (+ residual) (+ LeakyReLU(0.2)) in one pass over the activation.  offer_tail: the caller guarantees that the
    result has exactly ONE consumer (the next Block): a training-mode BatchNorm then offers that consumer the first half of
    its backward (ops.NormTail)."""
    kind = _norm_kind(m)
    if kind == "identity":
        if res is None and not act:
            return x
        return ops.NormActFn.apply(x, res, None, None, None, None, None, None, None, "identity", False, act, 0.0, 0.0)
    if kind == "instance":
        return ops.NormActFn.apply(x, res, None, None, None, None, None, None, None, "instance", m.training, act,
                                   float(m.eps), 0.0)
    a = owner.arena()
    gs, bs = a.by_param[id(m.weight)], a.by_param[id(m.bias)]
    groups = ops.current_bn_groups() if m.training else 1
    if m.training:
        # counted on the host and folded into the buffer when state_dict() is taken
        # (74 one-element device increments per forward would only cost launches)
        m.__dict__["_bg_nbt_pending"] = m.__dict__.get("_bg_nbt_pending", 0) + groups * ops.current_bn_repeat()
    mom = 0.1 if m.momentum is None else float(m.momentum)
    if ops.current_bn_repeat() > 1:      # k identical forwards in one: the closed form of k momentum updates
        mom = 1.0 - (1.0 - mom) ** ops.current_bn_repeat()
    return ops.NormActFn.apply(x, res, m.weight, m.bias, a, gs, bs, m.running_mean, m.running_var, "batch", m.training,
                               act, float(m.eps), mom, stats, groups, None, bool(offer_tail))


def fixed_padding_extents(kernel_size, rate):
    """(pad_beg, pad_end) of the reference's fixed_padding (deeplab.py:66-72)."""
    k_eff = kernel_size + (kernel_size - 1) * (rate - 1)
    total = k_eff - 1
    return total // 2, total - total // 2


class SeparableConv2d_same(BGModule):
    """depthwise 3x3 ("same" padding folded in) -> pointwise 1x1, nothing between
    (deeplab.py:75-87)."""

    def __init__(self, inplanes, planes, kernel_size=3, stride=1, dilation=1, bias=False):
        super().__init__()
        assert kernel_size == 3 and not bias
        self.conv1 = Conv2d(inplanes, inplanes, kernel_size, stride, 0, dilation, groups=inplanes, bias=bias)
        self.pointwise = Conv2d(inplanes, planes, 1, 1, 0, 1, 1, bias=bias)

    def forward(self, x, stats=None):
        return self.pointwise(self.conv1(x), stats)


class Block(BGModule):
    """Xception residual unit (deeplab.py:90-143)."""

    def __init__(self, inplanes, planes, reps, stride=1, dilation=1, start_with_relu=True, grow_first=True,
                 is_last=False, normalizer=nn.BatchNorm2d):
        super().__init__()
        if planes != inplanes or stride != 1:
            self.skip = Conv2d(inplanes, planes, 1, stride=stride, bias=False)
            self.skipbn = normalizer(planes)
        else:
            self.skip = None
        self.relu = nn.LeakyReLU(0.2, inplace=True)
        self.start_with_relu = start_with_relu
        rep = []
        filters = inplanes
        if grow_first:
            rep += [self.relu, SeparableConv2d_same(inplanes, planes, 3, stride=1, dilation=dilation), normalizer(planes)]
            filters = planes
        for _ in range(reps - 1):
            rep += [self.relu, SeparableConv2d_same(filters, filters, 3, stride=1, dilation=dilation), normalizer(filters)]
        if not grow_first:
            rep += [self.relu, SeparableConv2d_same(inplanes, planes, 3, stride=1, dilation=dilation), normalizer(planes)]
        if not start_with_relu:
            rep = rep[1:]
        if stride != 1:
            rep.append(SeparableConv2d_same(planes, planes, 3, stride=2))
        if stride == 1 and is_last:
            rep.append(SeparableConv2d_same(planes, planes, 3, stride=1))
        self.rep = nn.Sequential(*rep)

    def forward(self, inp, pre_activated=False, activate_output=False, sole_consumer=False):
        """inp: NHWC.  pre_activated: the caller already applied this block's
        leading LeakyReLU (fused into the producer).  activate_output: also apply
        the NEXT block's leading LeakyReLU to the sum.  sole_consumer: the caller passes the result to exactly one
        consumer, the next Block (whose fork backward may then take over the first half of this Block's final BatchNorm
        backward: ops.NormTail)."""
        a = inp
        if self.start_with_relu and not pre_activated:
            a = ops.leaky_relu(inp)
        units = list(self.rep)
        i = 1 if self.start_with_relu else 0
        dw_done = False  # h is already the depthwise output of the unit about to run (norm_act_dw / the fork below)
        first = units[i] if i < len(units) else None
        if isinstance(first, SeparableConv2d_same) and ops.fork_dw_ok(first.conv1):
            # the Block's input feeds its first depthwise convolution and its skip path: one autograd node whose
            # backward adds the skip path's gradient inside the depthwise data-gradient kernel
            arena = self.arena()
            h, a_skip = ops.ForkDwConv3x3Fn.apply(a, first.conv1.weight, arena, arena.by_param[id(first.conv1.weight)],
                                                  first.conv1.dilation[0], getattr(a, "_bg_tail", None))
            dw_done = True
        else:
            a_main, a_skip = ops.fork(a, 2)
            h = a_main
        last_norm = None
        last_sep = None
        while i < len(units):
            u = units[i]
            if isinstance(u, SeparableConv2d_same):
                nxt_norm = i + 1 < len(units) and not isinstance(units[i + 1], (SeparableConv2d_same, nn.LeakyReLU))
                if not nxt_norm:
                    h = u.pointwise(h) if dw_done else u(h)
                    dw_done = False
                    i += 1
                    continue
                m = units[i + 1]
                if i + 1 == len(units) - 1:      # block-final norm: fused with the residual below
                    last_norm, last_sep = m, u
                    break
                nxt_relu = isinstance(units[i + 2], nn.LeakyReLU)
                if nxt_relu and i + 3 < len(units) and norm_dw_fusable(m, units[i + 3]):
                    # [u -> m -> LeakyReLU -> next unit]: m and the activation run inside the next unit's depthwise kernel
                    z, stats = conv_norm(self, u, m, h, skip_dw=dw_done, defer_norm=True)
                    h = norm_act_dw(self, m, z, stats, units[i + 3].conv1)
                    dw_done = True
                    i += 3
                    continue
                h = conv_norm(self, u, m, h, act=nxt_relu, skip_dw=dw_done)
                dw_done = False
                i += 3 if nxt_relu else 2
            elif isinstance(u, nn.LeakyReLU):
                h = ops.leaky_relu(h)
                i += 1
            else:  # a norm not preceded by a conv (not produced by the reference's constructor)
                h = apply_norm(self, u, h)
                i += 1
        if self.skip is not None:
            s = conv_norm(self, self.skip, self.skipbn, a_skip)
        else:
            s = a_skip
        if last_norm is not None:
            return conv_norm(self, last_sep, last_norm, h, res=s, act=activate_output, skip_dw=dw_done, offer_tail=sole_consumer)
        return ops.add(h, s, act=activate_output)


class Xception(BGModule):
    """Modified aligned Xception-65 feature extractor (deeplab.py:146-291)."""

    def __init__(self, inplanes=3, os=16, pretrained=False, normalizer=nn.BatchNorm2d):
        super().__init__()
        if os == 16:
            entry_block3_stride, middle_block_rate, exit_block_rates = 2, 1, (1, 2)
        elif os == 8:
            entry_block3_stride, middle_block_rate, exit_block_rates = 1, 2, (2, 4)
        else:
            raise NotImplementedError
        if pretrained:
            raise NotImplementedError("pretrained weights are a remote download in the reference (deeplab.py:295); "
                                      "load a state_dict instead")
        self.inplanes = inplanes
        self.conv1 = Conv2d(inplanes, 128, 3, stride=2, padding=1, bias=False)
        self.bn1 = normalizer(128)
        self.relu = nn.LeakyReLU(0.2, inplace=True)
        self.conv2 = Conv2d(128, 128, 3, stride=1, padding=1, bias=False)
        self.bn2 = normalizer(128)
        self.block1 = Block(128, 128, reps=2, stride=2, start_with_relu=False, normalizer=normalizer)
        self.block2 = Block(128, 256, reps=2, stride=2, start_with_relu=True, grow_first=True, normalizer=normalizer)
        self.block3 = Block(256, 728, reps=2, stride=entry_block3_stride, start_with_relu=True, grow_first=True,
                            is_last=True, normalizer=normalizer)
        for i in range(4, 20):
            setattr(self, f"block{i}", Block(728, 728, reps=3, stride=1, dilation=middle_block_rate, start_with_relu=True,
                                             grow_first=True, normalizer=normalizer))
        self.block20 = Block(728, 1024, reps=2, stride=1, dilation=exit_block_rates[0], start_with_relu=True,
                             grow_first=False, is_last=True, normalizer=normalizer)
        self.conv3 = SeparableConv2d_same(1024, 1536, 3, stride=1, dilation=exit_block_rates[1])
        self.bn3 = normalizer(1536)
        self.conv4 = SeparableConv2d_same(1536, 1536, 3, stride=1, dilation=exit_block_rates[1])
        self.bn4 = normalizer(1536)
        self.conv5 = SeparableConv2d_same(1536, 2048, 3, stride=1, dilation=exit_block_rates[1])
        self.bn5 = normalizer(2048)
        self._init_weight()

    def _init_weight(self):
        # deeplab.py:280-291: kaiming_normal_ on every conv, norm affine = (1, 0)
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward_nhwc(self, x, want_low=True):
        x = conv_norm(self, self.conv1, self.bn1, x, act=True)
        # (block1 is the only reader of bn2's output and starts with its fork: it may take over the first half of bn2's
        # backward -- ops.NormTail, bg_dwconv3x3_bwd_fork -- on the step's largest tensors, 450 - 900 MB at 1152 x 768)
        x = conv_norm(self, self.conv2, self.bn2, x, act=True, offer_tail=True)
        # every block output below already carries the next block's leading LeakyReLU
        x = self.block1(x, activate_output=True)
        low = None
        if want_low:
            # = leaky_relu(block1 output): the tensor block2 activates in place (deeplab.py:241-243)
            low, x = ops.fork(x, 2)
        for i in range(2, 20):   # each of these outputs goes to the next Block and nowhere else
            x = getattr(self, f"block{i}")(x, pre_activated=True, activate_output=True, sole_consumer=True)
        # data parallel: once backward has come back to here, the gradients of the exit flow and of everything registered
        # after this Xception (ASPP, decoder / the critic's head: ~48 % of the generator's bytes) are final and their
        # all-reduce starts under the middle and entry flow's backward (ops.GradMilestoneFn; no-op on one GPU)
        x = ops.grad_milestone(x, self.arena(), list(self.block20.parameters()))
        x = self.block20(x, pre_activated=True, activate_output=False)
        x = conv_norm(self, self.conv3, self.bn3, x, act=True)
        x = conv_norm(self, self.conv4, self.bn4, x, act=True)
        x = conv_norm(self, self.conv5, self.bn5, x, act=True)
        return x, low

    def forward(self, x):
        """NCHW fp32 in, (features, low_level_feat) NCHW fp32 out, as the reference."""
        dt = self.compute_dtype()
        xi = ops.ToInternal.apply(x, pad_to(self.inplanes, vec_of(dt)), dt)
        f, low = self.forward_nhwc(xi)
        return ops.FromInternal.apply(f, 2048), ops.FromInternal.apply(low, 128)


class ASPP_module(BGModule):
    """1x1 (rate 1) or dilated 3x3 conv -> norm -> LeakyReLU (deeplab.py:322-355)."""

    def __init__(self, inplanes, planes, rate, normalizer=nn.BatchNorm2d):
        super().__init__()
        k, pad = (1, 0) if rate == 1 else (3, rate)
        self.atrous_convolution = Conv2d(inplanes, planes, k, stride=1, padding=pad, dilation=rate, bias=False)
        self.bn = normalizer(planes)
        self.relu = nn.LeakyReLU(0.2)
        nn.init.kaiming_normal_(self.atrous_convolution.weight)

    def forward(self, x):
        return conv_norm(self, self.atrous_convolution, self.bn, x, act=True)


def _gan_conv_init(m: Conv2d):
    # normal(0, gain/sqrt(k*k*Cout)), gain = calculate_gain('leaky_relu', 0.2) (deeplab.py:383-388)
    gain = nn.init.calculate_gain("leaky_relu", 0.2)
    n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
    nn.init.normal_(m.weight, mean=0.0, std=gain / math.sqrt(n))


class InterpolationUpsampler(BGModule):
    """bilinear to H/4, concat with the 48-channel skip, two 3x3 conv+norm+LeakyReLU,
    1x1 (with bias) to n_output, bilinear to HxW (deeplab.py:358-395)."""

    def __init__(self, n_output, normalizer=nn.BatchNorm2d):
        super().__init__()
        self.n_output = n_output
        self.last_conv = nn.Sequential(Conv2d(304, 256, 3, stride=1, padding=1, bias=False), normalizer(256),
                                       nn.LeakyReLU(0.2, inplace=True),
                                       Conv2d(256, 256, 3, stride=1, padding=1, bias=False), normalizer(256),
                                       nn.LeakyReLU(0.2, inplace=True), Conv2d(256, n_output, 1, stride=1))
        for m in self.modules():
            if isinstance(m, Conv2d):
                _gan_conv_init(m)

    def forward(self, x, low_level_features, input_size):
        H, W = int(input_size[-2]), int(input_size[-1])
        x = ops.ResizeBilinearFn.apply(x, -(-H // 4), -(-W // 4), None)
        x = ops.concat(x, low_level_features)
        lc = self.last_conv
        x = conv_norm(self, lc[0], lc[1], x, act=True)
        x = conv_norm(self, lc[3], lc[4], x, act=True)
        x = lc[6](x)
        # final resize writes fp32: it is the generator output / loss input
        return ops.ResizeBilinearFn.apply(x, H, W, torch.float32)


def _norm_pool_act(owner: BGModule, m: nn.Module, pool: nn.Module, x):
    """normalizer -> nnpooler -> LeakyReLU(0.2), the tail of every Deconv unit (deeplab.py:406-409)."""
    if isinstance(pool, nn.AvgPool2d):
        p = pool.padding if isinstance(pool.padding, int) else pool.padding[0]
        return ops.leaky_relu(ops.avgpool2x2(apply_norm(owner, m, x, act=False), p))
    return apply_norm(owner, m, x, act=True)  # nn_pooling=False: nn.Identity in its place


def _deconv_init(mod: nn.Module):
    # DeconvUpsampler/UpsamplerExtension.__init_weight (deeplab.py:446-462,500-516): the same normal
    # fill for Conv2d and ConvTranspose2d, n = k*k*out_channels
    gain = nn.init.calculate_gain("leaky_relu", 0.2)
    for m in mod.modules():
        if isinstance(m, (Conv2d, ConvTranspose2d)):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            nn.init.normal_(m.weight, mean=0.0, std=gain / math.sqrt(n))


class DeconvUpsampler(BGModule):
    """Four stride-2 transposed convolutions with hand-tuned output paddings, each followed by
    normaliser -> AvgPool2d(2, 1) (anti-checkerboard) -> LeakyReLU, around the skip concat and the
    3x3 / 3x3 / 1x1 stack (deeplab.py:398-444).  Shape-locked by those paddings to H = 16a-13,
    W = 16b-11 inputs (19x37: the 10-degree GPS-RO grid); other sizes fail at the concat like the
    reference does."""

    def __init__(self, n_output, normalizer=nn.BatchNorm2d, nn_pooling=True):
        super().__init__()
        pooler = nn.AvgPool2d if nn_pooling else (lambda *a, **k: nn.Identity())

        def unit(cin, cout, pad, op):
            return nn.Sequential(ConvTranspose2d(cin, cout, 3, stride=2, padding=pad, output_padding=op, bias=False),
                                 normalizer(cout), pooler(2, stride=1, padding=0), nn.LeakyReLU(0.2, inplace=True))
        self.deconv1 = unit(256, 256, 1, (0, 1))
        self.deconv2 = unit(256, 256, 0, (1, 0))
        self.conv1 = nn.Sequential(Conv2d(304, 256, 3, stride=1, padding=1, bias=False), normalizer(256),
                                   nn.LeakyReLU(0.2, inplace=True),
                                   Conv2d(256, 256, 3, stride=1, padding=1, bias=False), normalizer(256),
                                   nn.LeakyReLU(0.2, inplace=True), Conv2d(256, 256, 1, stride=1))
        self.deconv3 = unit(256, 128, 1, (1, 0))
        self.last_deconv = nn.Sequential(ConvTranspose2d(128, n_output, 3, stride=2, padding=1, output_padding=(1, 1),
                                                         bias=False))
        _deconv_init(self)

    def _unit(self, seq, x):
        return _norm_pool_act(self, seq[1], seq[2], seq[0](x))

    def forward(self, x, low_level_features, input_size):
        x = self._unit(self.deconv1, x)
        x = self._unit(self.deconv2, x)
        if tuple(x.shape[1:3]) != tuple(low_level_features.shape[1:3]):
            raise RuntimeError("Sizes of tensors must match except in dimension 1: the Deconv upsampler needs "
                               "H = 16a-13, W = 16b-11 (got a {}x{} decoder map for a {}x{} skip)".format(
                                   x.shape[1], x.shape[2], low_level_features.shape[1], low_level_features.shape[2]))
        x = ops.concat(x, low_level_features)
        c1 = self.conv1
        x = conv_norm(self, c1[0], c1[1], x, act=True)
        x = conv_norm(self, c1[3], c1[4], x, act=True)
        x = c1[6](x)
        x = self._unit(self.deconv3, x)
        return self.last_deconv[0](x)


class UpsamplerExtension(BGModule):
    """Full-resolution stem on the raw input + normalised decoder output -> 3x3 / 3x3 to n_output
    (deeplab.py:465-498)."""

    def __init__(self, n_input, n_output, normalizer=nn.BatchNorm2d, nn_pooling=True):
        super().__init__()
        pooler = nn.AvgPool2d if nn_pooling else (lambda *a, **k: nn.Identity())
        self.init_norm = nn.Sequential(normalizer(128), pooler(2, stride=1, padding=1), nn.LeakyReLU(0.2, inplace=True))
        self.conv1 = nn.Sequential(Conv2d(n_input, 64, 3, stride=1, padding=1, bias=False), normalizer(64),
                                   nn.LeakyReLU(0.2, inplace=True),
                                   Conv2d(64, 128, 3, stride=1, padding=1, bias=False), normalizer(128),
                                   nn.LeakyReLU(0.2, inplace=True))
        self.conv2 = nn.Sequential(Conv2d(256, 64, 3, stride=1, padding=1, bias=False), normalizer(64),
                                   nn.LeakyReLU(0.2, inplace=True),
                                   Conv2d(64, n_output, 3, stride=1, padding=1, bias=False))
        _deconv_init(self)

    def forward(self, input, x):
        c1, c2 = self.conv1, self.conv2
        skip = conv_norm(self, c1[0], c1[1], input, act=True)
        skip = conv_norm(self, c1[3], c1[4], skip, act=True)
        x = _norm_pool_act(self, self.init_norm[0], self.init_norm[1], x)
        x = ops.concat(x, skip)
        x = conv_norm(self, c2[0], c2[1], x, act=True)
        return c2[3](x)


class DeepLabv3_plus(BGModule):
    """Encoder-ASPP-decoder (deeplab.py:585-684)."""

    def __init__(self, n_input=3, n_output=21, os=16, upsampler_type="Deconv", pretrained=False, _print=True,
                 normalizer=nn.BatchNorm2d, nn_pooling=True):
        super().__init__()
        if _print:
            print("Constructing DeepLabv3+ model...")
            print("Number of output channels: {}".format(n_output))
            print("Output stride: {}".format(os))
            print("Number of Input Channels: {}".format(n_input))
        self.upsampler_type, self.nn_pooling = upsampler_type, nn_pooling
        self.n_input, self.n_output = n_input, n_output
        self.xception_features = Xception(n_input, os, pretrained, normalizer)
        if os == 16:
            rates = [1, 6, 12, 18]
        elif os == 8:
            rates = [1, 12, 24, 36]
        else:
            raise NotImplementedError
        self.aspp1 = ASPP_module(2048, 256, rate=rates[0], normalizer=normalizer)
        self.aspp2 = ASPP_module(2048, 256, rate=rates[1], normalizer=normalizer)
        self.aspp3 = ASPP_module(2048, 256, rate=rates[2], normalizer=normalizer)
        self.aspp4 = ASPP_module(2048, 256, rate=rates[3], normalizer=normalizer)
        self.relu = nn.LeakyReLU(0.2)
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), Conv2d(2048, 256, 1, stride=1, bias=False),
                                             normalizer(256), nn.LeakyReLU(0.2))
        self.conv1 = Conv2d(1280, 256, 1, bias=False)
        self.bn1 = normalizer(256)
        self.conv2 = Conv2d(128, 48, 1, bias=False)
        self.bn2 = normalizer(48)
        if self.upsampler_type == "Interpolate":
            self.upsample = InterpolationUpsampler(n_output, normalizer)
        elif self.upsampler_type.startswith("Deconv"):
            self.upsample = DeconvUpsampler(n_output=128 if self.upsampler_type == "Deconv1x" else n_output,
                                            normalizer=normalizer, nn_pooling=nn_pooling)
        else:
            raise NotImplementedError("Error, upsampler {} not implemented.".format(upsampler_type))
        if self.upsampler_type in ("Deconv1x", "Interpolate1x"):
            self.upsample_extension = UpsamplerExtension(n_input, n_output, normalizer, nn_pooling)
        elif self.upsampler_type == "Deconv":
            self.final_pool = nn.AvgPool2d(2, stride=1, padding=1) if nn_pooling else nn.Identity()

    def forward_nhwc(self, xi, H, W):
        if self.upsampler_type == "Deconv1x":
            xi, x_raw = ops.fork(xi, 2)              # the extension reads the raw input again
        x, low = self.xception_features.forward_nhwc(xi)
        x1i, x2i, x3i, x4i, x5i = ops.fork(x, 5)
        x1, x2, x3, x4 = self.aspp1(x1i), self.aspp2(x2i), self.aspp3(x3i), self.aspp4(x4i)
        gp = self.global_avg_pool
        x5 = conv_norm(self, gp[1], gp[2], ops.GlobalAvgPoolFn.apply(x5i), act=True)
        x5 = ops.ResizeBilinearFn.apply(x5, x4.shape[1], x4.shape[2], None)
        x = ops.concat(x1, x2, x3, x4, x5)
        x = conv_norm(self, self.conv1, self.bn1, x, act=True)
        low = conv_norm(self, self.conv2, self.bn2, low, act=True)
        x = self.upsample(x, low, (H, W))
        if self.upsampler_type == "Deconv1x":        # deeplab.py:679-682
            x = self.upsample_extension(x_raw, x)
        elif self.upsampler_type == "Deconv" and isinstance(self.final_pool, nn.AvgPool2d):
            x = ops.avgpool2x2(x, 1)
        return x

    def forward(self, input):
        """NCHW fp32 [N, n_input, H, W] -> NCHW fp32 [N, n_output, H, W]."""
        dt = self.compute_dtype()
        n, c, H, W = input.shape
        assert c == self.n_input, f"expected {self.n_input} input channels, got {c}"
        xi = ops.ToInternal.apply(input, pad_to(c, vec_of(dt)), dt)
        y = self.forward_nhwc(xi, H, W)
        return ops.FromInternal.apply(y, self.n_output)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()
