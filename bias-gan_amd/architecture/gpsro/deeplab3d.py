"""3-D DeepLab / Xception3d on the MI355X kernels.

Host-side mirror of the reference's ``architecture/gpsro/deeplab3d.py``: same class names, constructor
arguments and state_dict keys.  A volume [N, C, D, H, W] is held inside the network as the NHWC tensor
[N*D, H, W, C] (depth folded into the batch), so the normalisation, activation, concat and 1x1x1-convolution
kernels of the 2-D path apply as they are; the third dimension is handled by

  * ``Conv3d``: depth unfold (ops.DepthUnfoldFn) + the 2-D GEMM convolution over KD*C channels;
  * ``SeparableConv3d_same``: depthwise 3x3x3 kernel with the "same" padding folded in + pointwise 1x1x1;
  * trilinear interpolation = linear along depth, then the 2-D bilinear kernel (separable).

Forward functions take and return ``(x, n)``-style arguments: ``x`` the folded tensor, ``n`` the sample count.
Only the Interpolate upsampler is built (the Deconv ones are shape-locked, deeplab3d.py:342-466).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ... import ops
from ...runtime import BGModule, pad_to, vec_of
from .deeplab import Identity, _norm_kind, _vec_layout  # noqa: F401


# ------------------------------------------------------------------ parameter holders
def _conv3d_layout(p, vec):
    k, c, kd, r, s = p.shape
    return "conv3d", (pad_to(k, vec), r, s, kd * pad_to(c, vec))


def _dw3d_layout(p, vec):
    c, _, kd, r, s = p.shape
    return "dw3d", (kd, r, s, pad_to(c, vec))


class Conv3d(BGModule):
    """nn.Conv3d parameter container + launcher (weight [Cout, Cin/groups, k, k, k], optional bias)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        assert groups in (1, in_channels), "only dense and depthwise convolutions occur on this path"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size,) * 3
        self.stride, self.padding, self.dilation, self.groups = (stride,) * 3, (padding,) * 3, (dilation,) * 3, groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, kernel_size, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(self.weight[0].numel())
            nn.init.uniform_(self.bias, -bound, bound)
        self._bg_param_layout = {"weight": _dw3d_layout if groups > 1 else _conv3d_layout, "bias": _vec_layout}

    def forward(self, x, n, stats=None, with_bias=True):
        """x: [N*D, H, W, Cin] folded volume; returns [N*Do, Ho, Wo, Cout].  with_bias=False leaves the bias to the
        caller (the partial convolution adds it after its mask ratio)."""
        a = self.arena()
        ws = a.by_param[id(self.weight)]
        k, s, p, d = self.kernel_size[0], self.stride[0], self.padding[0], self.dilation[0]
        if self.groups > 1:
            assert k == 3 and p == 0, "depthwise convolutions on this path are the 'same'-padded 3x3x3 ones"
            return ops.DwConv3dFn.apply(x, self.weight, a, ws, n, s, d)
        if k > 1 or s > 1:   # gather the depth taps (k = 1, stride 2: depth subsampling) next to the channels
            x = ops.DepthUnfoldFn.apply(x, n, k, s, p, d)
        if self.bias is None or not with_bias:
            return ops.Conv2dFn.apply(x, self.weight, None, a, ws, None, s, p, d, stats)
        return ops.Conv2dFn.apply(x, self.weight, self.bias, a, ws, a.by_param[id(self.bias)], s, p, d, stats)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size[0]}, s={self.stride[0]}, groups={self.groups}"


def apply_norm3d(owner: BGModule, m: nn.Module, x, n, res=None, act=False, stats=None):
    """normalizer (+ residual) (+ LeakyReLU) on a folded volume; instance statistics are per SAMPLE (n groups)."""
    kind = _norm_kind3d(m)
    if kind == "identity":
        if res is None and not act:
            return x
        return ops.NormActFn.apply(x, res, None, None, None, None, None, None, None, "identity", False, act, 0.0, 0.0)
    if kind == "instance":
        return ops.NormActFn.apply(x, res, None, None, None, None, None, None, None, "instance", m.training, act,
                                   float(m.eps), 0.0, None, 1, n)
    a = owner.arena()
    gs, bs = a.by_param[id(m.weight)], a.by_param[id(m.bias)]
    groups = ops.current_bn_groups() if m.training else 1
    if m.training:
        m.__dict__["_bg_nbt_pending"] = m.__dict__.get("_bg_nbt_pending", 0) + groups
    mom = 0.1 if m.momentum is None else float(m.momentum)
    return ops.NormActFn.apply(x, res, m.weight, m.bias, a, gs, bs, m.running_mean, m.running_var, "batch", m.training,
                               act, float(m.eps), mom, stats, groups)


def _norm_kind3d(m: nn.Module) -> str:
    if isinstance(m, nn.BatchNorm3d):
        if not (m.affine and m.track_running_stats):
            raise NotImplementedError("BatchNorm3d must keep its defaults (affine, running stats)")
        return "batch"
    if isinstance(m, nn.InstanceNorm3d):
        if m.affine or m.track_running_stats:
            raise NotImplementedError("InstanceNorm3d must keep its defaults (no affine, no running stats)")
        return "instance"
    if isinstance(m, (Identity, nn.Identity)):
        return "identity"
    raise NotImplementedError(f"normalizer {type(m).__name__} is not supported on the HIP path "
                              "(BatchNorm3d, InstanceNorm3d and Identity are)")


def conv_norm3d(owner: BGModule, conv, m: nn.Module, x, n, res=None, act=False):
    """conv (Conv3d or SeparableConv3d_same) -> norm (+ residual) (+ LeakyReLU); BatchNorm3d training statistics
    come out of the GEMM epilogue when the layer's pixels split evenly over the statistic groups."""
    dense = conv.pointwise if isinstance(conv, SeparableConv3d_same) else conv
    stats = None
    if isinstance(m, nn.BatchNorm3d) and m.training and dense.bias is None:
        groups = ops.current_bn_groups()
        if groups == 1:
            from ...runtime import StatsPool
            kp = owner.arena().by_param[id(dense.weight)].phys_shape[0]
            stats = StatsPool.get(x.device).take(2, 1, kp)
    y = conv(x, n, stats) if stats is not None else conv(x, n)
    return apply_norm3d(owner, m, y, n, res=res, act=act, stats=stats)


class SeparableConv3d_same(BGModule):
    """depthwise 3x3x3 ("same" padding folded in) -> pointwise 1x1x1 (deeplab3d.py:31-43)."""

    def __init__(self, inplanes, planes, kernel_size=3, stride=1, dilation=1, bias=False):
        super().__init__()
        assert kernel_size == 3 and not bias
        self.conv1 = Conv3d(inplanes, inplanes, kernel_size, stride, 0, dilation, groups=inplanes, bias=bias)
        self.pointwise = Conv3d(inplanes, planes, 1, 1, 0, 1, 1, bias=bias)

    def forward(self, x, n, stats=None):
        return self.pointwise(self.conv1(x, n), n, stats)


class Block3d(BGModule):
    """Xception residual unit in 3-D (deeplab3d.py:46-99)."""

    def __init__(self, inplanes, planes, reps, stride=1, dilation=1, start_with_relu=True, grow_first=True,
                 is_last=False, normalizer=nn.BatchNorm3d):
        super().__init__()
        if planes != inplanes or stride != 1:
            self.skip = Conv3d(inplanes, planes, 1, stride=stride, bias=False)
            self.skipbn = normalizer(planes)
        else:
            self.skip = None
        self.relu = nn.LeakyReLU(0.2, inplace=True)
        self.start_with_relu = start_with_relu
        rep = []
        filters = inplanes
        if grow_first:
            rep += [self.relu, SeparableConv3d_same(inplanes, planes, 3, stride=1, dilation=dilation), normalizer(planes)]
            filters = planes
        for _ in range(reps - 1):
            rep += [self.relu, SeparableConv3d_same(filters, filters, 3, stride=1, dilation=dilation), normalizer(filters)]
        if not grow_first:
            rep += [self.relu, SeparableConv3d_same(inplanes, planes, 3, stride=1, dilation=dilation), normalizer(planes)]
        if not start_with_relu:
            rep = rep[1:]
        if stride != 1:
            rep.append(SeparableConv3d_same(planes, planes, 3, stride=2))
        if stride == 1 and is_last:
            rep.append(SeparableConv3d_same(planes, planes, 3, stride=1))
        self.rep = nn.Sequential(*rep)

    def forward(self, inp, n, pre_activated=False, activate_output=False):
        """As deeplab.Block.forward: the in-place leading LeakyReLU of the reference is made explicit (the skip
        path and the caller's tensor see the activated value), the residual add is fused with the last norm."""
        a = inp
        if self.start_with_relu and not pre_activated:
            a = ops.leaky_relu(inp)
        a_main, a_skip = ops.fork(a, 2)
        units = list(self.rep)
        i = 1 if self.start_with_relu else 0
        h = a_main
        last_norm = last_sep = None
        while i < len(units):
            u = units[i]
            if isinstance(u, SeparableConv3d_same):
                nxt_norm = i + 1 < len(units) and not isinstance(units[i + 1], (SeparableConv3d_same, nn.LeakyReLU))
                if not nxt_norm:
                    h = u(h, n)
                    i += 1
                    continue
                m = units[i + 1]
                if i + 1 == len(units) - 1:
                    last_norm, last_sep = m, u
                    break
                nxt_relu = isinstance(units[i + 2], nn.LeakyReLU)
                h = conv_norm3d(self, u, m, h, n, act=nxt_relu)
                i += 3 if nxt_relu else 2
            elif isinstance(u, nn.LeakyReLU):
                h = ops.leaky_relu(h)
                i += 1
            else:
                h = apply_norm3d(self, u, h, n)
                i += 1
        s = conv_norm3d(self, self.skip, self.skipbn, a_skip, n) if self.skip is not None else a_skip
        if last_norm is not None:
            return conv_norm3d(self, last_sep, last_norm, h, n, res=s, act=activate_output)
        return ops.add(h, s, act=activate_output)


class Xception3d(BGModule):
    """Modified aligned Xception in 3-D (deeplab3d.py:102-221)."""

    def __init__(self, inplanes=3, os=16, pretrained=False, normalizer=nn.BatchNorm3d):
        super().__init__()
        if os == 16:
            entry_block3_stride, middle_block_rate, exit_block_rates = 2, 1, (1, 2)
        elif os == 8:
            entry_block3_stride, middle_block_rate, exit_block_rates = 1, 2, (2, 4)
        else:
            raise NotImplementedError
        if pretrained:
            raise NotImplementedError("pretrained weights are a remote download in the reference (deeplab3d.py:239)")
        self.inplanes = inplanes
        self.conv1 = Conv3d(inplanes, 32, 3, stride=2, padding=1, bias=False)
        self.bn1 = normalizer(32)
        self.relu = nn.LeakyReLU(0.2, inplace=True)
        self.conv2 = Conv3d(32, 64, 3, stride=1, padding=1, bias=False)
        self.bn2 = normalizer(64)
        self.block1 = Block3d(64, 128, reps=2, stride=2, start_with_relu=False, normalizer=normalizer)
        self.block2 = Block3d(128, 256, reps=2, stride=2, start_with_relu=True, grow_first=True, normalizer=normalizer)
        self.block3 = Block3d(256, 728, reps=2, stride=entry_block3_stride, start_with_relu=True, grow_first=True,
                              is_last=True, normalizer=normalizer)
        for i in range(4, 20):
            setattr(self, f"block{i}", Block3d(728, 728, reps=3, stride=1, dilation=middle_block_rate, start_with_relu=True,
                                               grow_first=True, normalizer=normalizer))
        self.block20 = Block3d(728, 1024, reps=2, stride=1, dilation=exit_block_rates[0], start_with_relu=True,
                               grow_first=False, is_last=True, normalizer=normalizer)
        self.conv3 = SeparableConv3d_same(1024, 1536, 3, stride=1, dilation=exit_block_rates[1])
        self.bn3 = normalizer(1536)
        self.conv4 = SeparableConv3d_same(1536, 1536, 3, stride=1, dilation=exit_block_rates[1])
        self.bn4 = normalizer(1536)
        self.conv5 = SeparableConv3d_same(1536, 2048, 3, stride=1, dilation=exit_block_rates[1])
        self.bn5 = normalizer(2048)
        for m in self.modules():           # Xception3d.__init_weight: kaiming_normal_ on every Conv3d (deeplab3d.py:223-234)
            if isinstance(m, Conv3d):
                nn.init.kaiming_normal_(m.weight)

    def forward_folded(self, x, n, want_low=True):
        x = conv_norm3d(self, self.conv1, self.bn1, x, n, act=True)
        x = conv_norm3d(self, self.conv2, self.bn2, x, n, act=True)
        x = self.block1(x, n, activate_output=True)
        low = None
        if want_low:
            low, x = ops.fork(x, 2)     # = leaky_relu(block1 output): aliased and activated in place by block2
        for i in range(2, 20):
            x = getattr(self, f"block{i}")(x, n, pre_activated=True, activate_output=True)
        x = self.block20(x, n, pre_activated=True, activate_output=False)
        x = conv_norm3d(self, self.conv3, self.bn3, x, n, act=True)
        x = conv_norm3d(self, self.conv4, self.bn4, x, n, act=True)
        x = conv_norm3d(self, self.conv5, self.bn5, x, n, act=True)
        return x, low


def to_folded(x5: torch.Tensor, cp: int, dtype: torch.dtype):
    """NCDHW fp32 -> folded NHWC [N*D, H, W, cp]: the same memory walk as NCHW -> NHWC with H := D*H."""
    n, c, d, h, w = x5.shape
    return ops.ToInternal.apply(x5.reshape(n, c, d * h, w), cp, dtype).view(n * d, h, w, cp)


def from_folded(x: torch.Tensor, n: int, c: int):
    """folded NHWC [N*D, H, W, Cp] -> NCDHW fp32 [N, c, D, H, W]."""
    nd, h, w, cp = x.shape
    d = nd // n
    y = ops.FromInternal.apply(ops.nhwc(x).reshape(n, d * h, w, cp), c)
    return y.view(n, c, d, h, w)


class ASPP_module(BGModule):
    """1x1x1 (rate 1) or dilated 3x3x3 conv -> norm -> LeakyReLU (deeplab3d.py:265-298)."""

    def __init__(self, inplanes, planes, rate, normalizer=nn.BatchNorm3d):
        super().__init__()
        k, pad = (1, 0) if rate == 1 else (3, rate)
        self.atrous_convolution = Conv3d(inplanes, planes, k, stride=1, padding=pad, dilation=rate, bias=False)
        self.bn = normalizer(planes)
        self.relu = nn.LeakyReLU(0.2)
        nn.init.kaiming_normal_(self.atrous_convolution.weight)

    def forward(self, x, n):
        return conv_norm3d(self, self.atrous_convolution, self.bn, x, n, act=True)


def _gan_conv3d_init(m: Conv3d):
    # normal(0, gain/sqrt(k*k*Cout)) -- the reference multiplies only two of the three kernel extents (deeplab3d.py:327-329)
    gain = nn.init.calculate_gain("leaky_relu", 0.2)
    nn.init.normal_(m.weight, mean=0.0, std=gain / math.sqrt(m.kernel_size[0] * m.kernel_size[1] * m.out_channels))


class InterpolationUpsampler(BGModule):
    """trilinear to /4, concat with the 48-channel skip, 3x3x3 / 3x3x3 / 1x1x1, trilinear to full size
    (deeplab3d.py:301-340)."""

    def __init__(self, n_output, normalizer=nn.BatchNorm3d):
        super().__init__()
        self.n_output = n_output
        self.last_conv = nn.Sequential(Conv3d(304, 256, 3, stride=1, padding=1, bias=False), normalizer(256),
                                       nn.LeakyReLU(0.2, inplace=True),
                                       Conv3d(256, 256, 3, stride=1, padding=1, bias=False), normalizer(256),
                                       nn.LeakyReLU(0.2, inplace=True), Conv3d(256, n_output, 1, stride=1))
        for m in self.modules():
            if isinstance(m, Conv3d):
                _gan_conv3d_init(m)

    def forward(self, x, low_level_features, n, size):
        D, H, W = size
        x = ops.resize_trilinear(x, n, -(-D // 4), -(-H // 4), -(-W // 4))
        x = ops.concat(x, low_level_features)
        lc = self.last_conv
        x = conv_norm3d(self, lc[0], lc[1], x, n, act=True)
        x = conv_norm3d(self, lc[3], lc[4], x, n, act=True)
        x = lc[6](x, n)
        return ops.resize_trilinear(x, n, D, H, W, torch.float32)   # fp32: generator output / loss input


class ConvTranspose3d(BGModule):
    """nn.ConvTranspose3d(groups=1, bias=False) parameter container (weight [Cin, Cout, k, k, k]).  The arena stores it
    as the dense 3-D convolution it is the adjoint of; forward = that convolution's data gradient: the 2-D
    data-gradient GEMM over the KD*C channel blocks, then the depth fold."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=(0, 0, 0), bias=False):
        super().__init__()
        assert not bias, "the transposed convolutions on this path have no bias"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size,) * 3
        self.stride, self.padding = (stride,) * 3, (padding,) * 3
        self.output_padding = tuple(output_padding) if not isinstance(output_padding, int) else (output_padding,) * 3
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, kernel_size, kernel_size, kernel_size))
        self.bias = None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self._bg_param_layout = {"weight": _conv3d_layout}

    def forward(self, x, n):
        a = self.arena()
        ws = a.by_param[id(self.weight)]
        k, s, p = self.kernel_size[0], self.stride[0], self.padding[0]
        y = ops.ConvTranspose2dFn.apply(x, self.weight, a, ws, s, p, self.output_padding[1:])
        d = x.shape[0] // n
        return ops.DepthFoldFn.apply(y, n, k, s, p, 1, (d - 1) * s - 2 * p + k + self.output_padding[0])

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size[0]}, s={self.stride[0]}, op={self.output_padding}"


def _norm_pool_act3d(owner: BGModule, m: nn.Module, pool: nn.Module, x, n):
    """normalizer -> nnpooler -> LeakyReLU(0.2), the tail of every 3-D Deconv unit (deeplab3d.py:351-354)."""
    if isinstance(pool, nn.AvgPool3d):
        p = pool.padding if isinstance(pool.padding, int) else pool.padding[0]
        return ops.leaky_relu(ops.avgpool3d_2(apply_norm3d(owner, m, x, n, act=False), n, p))
    return apply_norm3d(owner, m, x, n, act=True)   # nn_pooling=False: nn.Identity in its place


def _deconv3d_init(mod: nn.Module):
    # DeconvUpsampler / DeconvUpsamplerExtension.__init_weight (deeplab3d.py:397-414,446-463): the same normal fill for
    # Conv3d and ConvTranspose3d, n = k0 * k1 * out_channels
    gain = nn.init.calculate_gain("leaky_relu", 0.2)
    for m in mod.modules():
        if isinstance(m, (Conv3d, ConvTranspose3d)):
            nn.init.normal_(m.weight, mean=0.0, std=gain / math.sqrt(m.kernel_size[0] * m.kernel_size[1] * m.out_channels))


class DeconvUpsampler(BGModule):
    """Four stride-2 transposed 3-D convolutions, each followed by normaliser -> AvgPool3d(2, 1) -> LeakyReLU, around
    the trilinear match to the skip, the concat and the 3x3x3 / 3x3x3 / 1x1x1 stack (deeplab3d.py:342-395).
    Shape-locked by the output paddings to D = 4a-3, H = 4b-1, W = 4c-3 once DeepLab3d's final pool is applied
    (45 x 19 x 37: the GPS-RO grid)."""

    def __init__(self, n_output, normalizer=nn.BatchNorm3d, nn_pooling=True):
        super().__init__()
        pooler = nn.AvgPool3d if nn_pooling else (lambda *a, **k: nn.Identity())

        def unit(cin, cout, op):
            return nn.Sequential(ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=op, bias=False),
                                 normalizer(cout), pooler(2, stride=1, padding=0), nn.LeakyReLU(0.2, inplace=True))
        self.deconv1 = unit(256, 256, (1, 1, 1))
        self.deconv2 = unit(256, 256, (1, 1, 1))
        self.conv1 = nn.Sequential(Conv3d(304, 256, 3, stride=1, padding=1, bias=False), normalizer(256),
                                   nn.LeakyReLU(0.2, inplace=True),
                                   Conv3d(256, 256, 3, stride=1, padding=1, bias=False), normalizer(256),
                                   nn.LeakyReLU(0.2, inplace=True), Conv3d(256, 256, 1, stride=1))
        self.deconv3 = unit(256, 128, (0, 1, 0))
        self.last_deconv = nn.Sequential(ConvTranspose3d(128, n_output, 3, stride=2, padding=1, output_padding=(1, 1, 1),
                                                         bias=False))
        _deconv3d_init(self)

    def _unit(self, seq, x, n):
        return _norm_pool_act3d(self, seq[1], seq[2], seq[0](x, n), n)

    def forward(self, x, low_level_features, n, size):
        x = self._unit(self.deconv1, x, n)
        x = self._unit(self.deconv2, x, n)
        low = low_level_features
        x = ops.resize_trilinear(x, n, low.shape[0] // n, low.shape[1], low.shape[2])    # "add a matching layer"
        x = ops.concat(x, low)
        c1 = self.conv1
        x = conv_norm3d(self, c1[0], c1[1], x, n, act=True)
        x = conv_norm3d(self, c1[3], c1[4], x, n, act=True)
        x = c1[6](x, n)
        x = self._unit(self.deconv3, x, n)
        return self.last_deconv[0](x, n)


class DeconvUpsamplerExtension(BGModule):
    """Full-resolution stem on the raw input + normalised decoder output -> 3x3x3 / 3x3x3 to n_output
    (deeplab3d.py:417-444)."""

    def __init__(self, n_input, n_output, normalizer=nn.BatchNorm3d, nn_pooling=True):
        super().__init__()
        pooler = nn.AvgPool3d if nn_pooling else (lambda *a, **k: nn.Identity())
        self.init_norm = nn.Sequential(normalizer(128), pooler(2, stride=1, padding=1), nn.LeakyReLU(0.2, inplace=True))
        self.conv1 = nn.Sequential(Conv3d(n_input, 64, 3, stride=1, padding=1, bias=False), normalizer(64),
                                   nn.LeakyReLU(0.2, inplace=True),
                                   Conv3d(64, 128, 3, stride=1, padding=1, bias=False), normalizer(128),
                                   nn.LeakyReLU(0.2, inplace=True))
        self.conv2 = nn.Sequential(Conv3d(256, 64, 3, stride=1, padding=1, bias=False), normalizer(64),
                                   nn.LeakyReLU(0.2, inplace=True),
                                   Conv3d(64, n_output, 3, stride=1, padding=1, bias=False))
        _deconv3d_init(self)

    def forward(self, input, x, n):
        c1, c2 = self.conv1, self.conv2
        skip = conv_norm3d(self, c1[0], c1[1], input, n, act=True)
        skip = conv_norm3d(self, c1[3], c1[4], skip, n, act=True)
        x = _norm_pool_act3d(self, self.init_norm[0], self.init_norm[1], x, n)
        if tuple(x.shape[:3]) != tuple(skip.shape[:3]):
            raise RuntimeError("Sizes of tensors must match except in dimension 1: the 3-D Deconv upsampler needs "
                               "D = 4a-3, H = 4b-1, W = 4c-3 inputs")
        x = ops.concat(x, skip)
        x = conv_norm3d(self, c2[0], c2[1], x, n, act=True)
        return c2[3](x, n)


class DeepLab3d(BGModule):
    """Encoder-ASPP-decoder in 3-D (deeplab3d.py:468-566)."""

    def __init__(self, n_input=3, n_output=21, os=16, upsampler_type="Deconv", pretrained=False, _print=True,
                 normalizer=nn.BatchNorm3d, nn_pooling=True):
        super().__init__()
        if _print:
            print("Constructing DeepLabv3+ model...")
            print("Number of output channels: {}".format(n_output))
            print("Output stride: {}".format(os))
            print("Number of Input Channels: {}".format(n_input))
        self.upsampler_type, self.nn_pooling = upsampler_type, nn_pooling
        self.n_input, self.n_output = n_input, n_output
        self.xception_features = Xception3d(n_input, os, pretrained, normalizer)
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
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool3d((1, 1, 1)), Conv3d(2048, 256, 1, stride=1, bias=False),
                                             normalizer(256), nn.LeakyReLU(0.2))
        self.conv1 = Conv3d(1280, 256, 1, bias=False)
        self.bn1 = normalizer(256)
        self.conv2 = Conv3d(128, 48, 1, bias=False)
        self.bn2 = normalizer(48)
        if self.upsampler_type == "Interpolate":
            self.upsample = InterpolationUpsampler(n_output, normalizer)
        elif self.upsampler_type.startswith("Deconv"):
            self.upsample = DeconvUpsampler(n_output=128 if self.upsampler_type == "Deconv1x" else n_output,
                                            normalizer=normalizer, nn_pooling=nn_pooling)
        else:
            raise NotImplementedError("Error, upsampler {} not implemented.".format(upsampler_type))
        if self.upsampler_type == "Deconv1x":
            self.upsample_extension = DeconvUpsamplerExtension(n_input, n_output, normalizer, nn_pooling)
        elif self.upsampler_type == "Deconv":
            self.final_pool = nn.AvgPool3d(2, stride=1, padding=1) if nn_pooling else nn.Identity()

    def forward(self, input):
        """NCDHW fp32 [N, n_input, D, H, W] -> NCDHW fp32 [N, n_output, D, H, W]."""
        dt = self.compute_dtype()
        n, c, D, H, W = input.shape
        assert c == self.n_input, f"expected {self.n_input} input channels, got {c}"
        xi = to_folded(input, pad_to(c, vec_of(dt)), dt)
        x, low = self.xception_features.forward_folded(xi, n)
        x1i, x2i, x3i, x4i, x5i = ops.fork(x, 5)
        x1, x2, x3, x4 = self.aspp1(x1i, n), self.aspp2(x2i, n), self.aspp3(x3i, n), self.aspp4(x4i, n)
        gp = self.global_avg_pool
        x5 = conv_norm3d(self, gp[1], gp[2], ops.GlobalAvgPoolFn.apply(x5i, n), n, act=True)
        x5 = ops.resize_trilinear(x5, n, x4.shape[0] // n, x4.shape[1], x4.shape[2])
        x = ops.concat(x1, x2, x3, x4, x5)
        x = conv_norm3d(self, self.conv1, self.bn1, x, n, act=True)
        low = conv_norm3d(self, self.conv2, self.bn2, low, n, act=True)
        y = self.upsample(x, low, n, (D, H, W))
        if self.upsampler_type == "Deconv1x":
            y = self.upsample_extension(xi, y, n)
        elif self.upsampler_type == "Deconv" and isinstance(self.final_pool, nn.AvgPool3d):
            y = ops.avgpool3d_2(y, n, 1)
        return from_folded(y, n, self.n_output)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm3d):
                m.eval()
