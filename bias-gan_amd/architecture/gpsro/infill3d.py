"""Partial-convolution U-Net in 3-D on the MI355X kernels.

Host-side mirror of the reference's ``architecture/gpsro/infill3d.py`` (PCBActiv3d, PConvUNet3d) and
``architecture/common/partialconv3d.py`` (PartialConv3d with multi_channel=True, return_mask=True): same class
names, constructor arguments and state_dict keys.  Volumes and masks live as folded NHWC tensors [N*D,H,W,C]
(see deeplab3d.py); masks hold exact 0/1 values.

PartialConv3d = mask window sum (bg_mask_window) + input*mask written straight into the concatenated layer input
(ops.MaskedConcatFn) + the dense convolution (depth unfold + 2-D GEMM kernels, deeplab3d.Conv3d) +
raw_out*mask_ratio (bg_scale_rows).  The updated mask is clamp(window sum, 0, 1) with Cout EQUAL channels, so it is
kept as one fp32 value per pixel (ops.RowsMask) and only the network's final mask is expanded to the tensor the
reference returns: no Cout-channel mask tensor is written, resized, concatenated or read inside the network.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import ops
from ...runtime import BGModule, pad_to, vec_of
from .deeplab3d import Conv3d, apply_norm3d, from_folded, to_folded

ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2


def partial_forward(conv, run_conv, xs, masks, n, planar):
    """The partial convolution around a dense convolution `run_conv(x)` (bias left out) of the container `conv`:
    mask window sum -> input * mask written into the concatenated layer input -> convolution -> * mask_ratio
    (+ bias * update_mask).  xs: folded tensors [N*D,H,W,C_i] whose channel concatenation is the layer input; masks:
    one per segment, a folded per-channel 0/1 tensor or an ops.RowsMask (one value per pixel standing for C_i equal
    channels, which every update_mask is).  Returns (output, update_mask as RowsMask over Cout channels).
    partialconv3d.py:79-84 / partialconv2d.py:79-84: ((conv + b - b) * ratio + b) * update_mask
    = conv * ratio + b * update_mask, since ratio already carries the 0/1 update_mask factor."""
    if not isinstance(xs, (list, tuple)):
        xs, masks = [xs], [masks]
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    dims = (xs[0].shape[0] // n, xs[0].shape[1], xs[0].shape[2])
    rows = [m for m in masks if isinstance(m, ops.RowsMask)]
    full = [m for m in masks if not isinstance(m, ops.RowsMask)]
    assert sum(m.channels for m in rows) <= conv.in_channels
    full_channels = conv.in_channels - sum(m.channels for m in rows)
    if len(full) > 1:      # per-channel masks on both halves of a concatenation (after PCDropout3d): one tensor for the window sum
        widths = [x.shape[3] for x, m in zip(xs, masks) if not isinstance(m, ops.RowsMask)]
        full = [ops.concat_full_masks(full, widths[:-1] + [full_channels - sum(widths[:-1])])]
    upd, ratio, _ = ops.mask_window(n, dims, k, s, p, conv.eps, full[0] if full else None, full_channels, rows, planar=planar)
    raw = run_conv(ops.MaskedConcatFn.apply(tuple(masks), *xs))
    a = conv.arena()
    bs = None if conv.bias is None else a.by_param[id(conv.bias)]
    out = ops.ScaleRowsFn.apply(raw, ratio, conv.bias, upd if bs is not None else None, a, bs)
    return out, ops.RowsMask(upd, conv.out_channels)


class PartialConv3d(Conv3d):
    """nn.Conv3d subclass of the reference (partialconv3d.py:14-92) for multi_channel=True, return_mask=True."""

    def __init__(self, *args, multi_channel=True, return_mask=True, eps=1.e-8, **kwargs):
        if not (multi_channel and return_mask):
            raise NotImplementedError("only PartialConv3d(multi_channel=True, return_mask=True) occurs on this path")
        super().__init__(*args, **kwargs)
        self.multi_channel, self.return_mask, self.eps = True, True, eps

    def forward(self, xs, masks, n):
        """See partial_forward; volumes folded as [N*D,H,W,C]."""
        return partial_forward(self, lambda x: Conv3d.forward(self, x, n, with_bias=False), xs, masks, n, planar=False)


def mask_tensor(m, n, dims, channels, dtype):
    """RowsMask -> folded tensor [N*D,H,W,pad(channels)] with equal channels (what the reference returns as mask)."""
    if not isinstance(m, ops.RowsMask):
        return m
    return ops.rows_from_scalar(m.rows, n * dims[0], dims[1], dims[2], pad_to(channels, vec_of(dtype)), dtype)


class PCBActiv3d(BGModule):
    """PartialConv3d -> normalizer -> ReLU | LeakyReLU(0.2) | none (infill3d.py:82-114)."""

    def __init__(self, in_ch, out_ch, normalizer=nn.BatchNorm3d, sample='none-3', activ='relu', conv_bias=False):
        super().__init__()
        k, s, p = {'down-5': (5, 2, 2), 'down-7': (7, 2, 3), 'down-3': (3, 2, 1), 'point-1': (1, 1, 0)}.get(sample, (3, 1, 1))
        self.conv = PartialConv3d(in_ch, out_ch, k, s, p, bias=conv_bias, multi_channel=True, return_mask=True, eps=1e-6)
        if normalizer is not None:
            self.bn = normalizer(out_ch)
        if activ == 'relu':
            self.activation = nn.ReLU()
        elif activ == 'leaky':
            self.activation = nn.LeakyReLU(negative_slope=0.2)

    def forward(self, x, mask, n):
        h, m = self.conv(x, mask, n)
        act = ACT_NONE
        if hasattr(self, 'activation'):
            act = ACT_RELU if isinstance(self.activation, nn.ReLU) else ACT_LEAKY
        if hasattr(self, 'bn'):
            h = apply_norm3d(self, self.bn, h, n, act=act)
        elif act:
            h = ops.NormActFn.apply(h, None, None, None, None, None, None, None, None, "identity", False, act, 0.0, 0.0)
        return h, m


class PCDropout3d(nn.Module):
    """infill3d.py:115-135: nn.Dropout3d acts on the MASK (whole (sample, channel) maps with probability p); what it
    dropped is zeroed in the input where the mask was valid and the input is rescaled by 1/(1-p).  Evaluation mode is
    the identity.  The draw is keep[n][c] ~ Bernoulli(1-p) from torch's generator (the reference's own stream of
    nn.Dropout3d cannot be reproduced); `inject` (a list of [N,C] 0/1 tensors, consumed call by call) replaces the
    draw -- that is how the parity tests feed the reference's recorded draws."""

    def __init__(self, p):
        super().__init__()
        self.p, self.scale = float(p), 1. - float(p)
        self.inject = None

    def draw(self, n, c, cp, device):
        if self.inject:
            k = self.inject.pop(0).to(device=device, dtype=torch.float32)
            assert tuple(k.shape) == (n, c), f"injected keep mask {tuple(k.shape)} for a layer of {(n, c)}"
        else:
            k = torch.bernoulli(torch.full((n, c), self.scale, dtype=torch.float32, device=device))
        if cp > c:
            k = torch.cat([k, torch.ones(n, cp - c, dtype=torch.float32, device=device)], dim=1)
        return k.contiguous()

    def forward(self, h, mask, n, channels):
        """h: folded features [N*D,H,W,Cp]; mask: RowsMask or per-channel folded tensor.  Returns (h_d, mask_d)."""
        if not self.training:
            return h, mask
        keep = self.draw(n, channels, h.shape[3], h.device)
        return ops.PCDropoutFn.apply(h, mask, keep, n, self.scale)


class PConvUNet3d(BGModule):
    """Partial-convolution U-Net (infill3d.py:135-239): upsampling_mode 'nearest' or 'trilinear', optional PCDropout3d."""

    def __init__(self, layer_size=7, input_channels=3, output_channels=3, upsampling_mode='nearest',
                 normalizer=nn.BatchNorm3d, dropout_p=0., compute_dtype=None):
        super().__init__()
        if upsampling_mode not in ('nearest', 'trilinear'):
            raise NotImplementedError("the HIP path builds upsampling_mode 'nearest' and 'trilinear'")
        self.freeze_enc_bn = False
        self.upsampling_mode, self.layer_size = upsampling_mode, layer_size
        self.input_channels, self.output_channels = input_channels, output_channels
        self.enc_1 = PCBActiv3d(input_channels, 64, sample='down-3', normalizer=None)
        self.enc_2 = PCBActiv3d(64, 128, sample='down-3', normalizer=normalizer)
        self.enc_3 = PCBActiv3d(128, 256, sample='down-3', normalizer=normalizer)
        self.enc_4 = PCBActiv3d(256, 512, sample='down-3', normalizer=normalizer)
        for i in range(4, self.layer_size):
            setattr(self, 'enc_{:d}'.format(i + 1), PCBActiv3d(512, 512, sample='down-3', normalizer=normalizer))
        for i in range(4, self.layer_size):
            setattr(self, 'dec_{:d}'.format(i + 1), PCBActiv3d(512 + 512, 512, activ='leaky', normalizer=normalizer))
        self.dec_4 = PCBActiv3d(512 + 256, 256, activ='leaky', normalizer=normalizer)
        self.dec_3 = PCBActiv3d(256 + 128, 128, activ='leaky', normalizer=normalizer)
        self.dec_2 = PCBActiv3d(128 + 64, 64, activ='leaky', normalizer=normalizer)
        self.dec_1 = PCBActiv3d(64 + input_channels, 32, activ='leaky', normalizer=normalizer)
        self.dropout = PCDropout3d(p=dropout_p) if dropout_p > 0. else None
        self.last_conv = PCBActiv3d(32, output_channels, activ=None, normalizer=None, sample='point-1', conv_bias=True)
        for m in self.modules():    # __init_weights (infill3d.py:166-175): kaiming_normal_ convs, zero biases
            if isinstance(m, Conv3d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        if compute_dtype is not None:
            self.set_compute_dtype(compute_dtype)

    def forward(self, input, input_mask):
        """NCDHW fp32 input and 0/1 mask [N,Cin,D,H,W] -> (output [N,Cout,D,H,W] fp32, its mask)."""
        dt = self.compute_dtype()
        n, c = input.shape[0], input.shape[1]
        cp = pad_to(c, vec_of(dt))
        hs, ms = {0: to_folded(input, cp, dt)}, {0: to_folded(input_mask, cp, dt)}
        for i in range(1, self.layer_size + 1):
            enc = getattr(self, 'enc_{:d}'.format(i))
            hs[i], ms[i] = enc(hs[i - 1], ms[i - 1], n)
            if self.dropout is not None:
                hs[i], ms[i] = self.dropout(hs[i], ms[i], n, enc.conv.out_channels)
        h, m = hs[self.layer_size], ms[self.layer_size]
        resize = ops.NearestResize3dFn if self.upsampling_mode == 'nearest' else ops.TrilinearResize3dFn
        for i in range(self.layer_size, 0, -1):
            e, em = hs[i - 1], ms[i - 1]
            src, size = (h.shape[0] // n, h.shape[1], h.shape[2]), (e.shape[0] // n, e.shape[1], e.shape[2])
            h = resize.apply(h, n, *size)
            if isinstance(m, ops.RowsMask):        # the mask is resized with mode='nearest' whatever the features use (infill3d.py:221-222)
                m = ops.nearest_rows(m, n, src, size)
            else:
                with torch.no_grad():
                    m = ops.NearestResize3dFn.apply(m, n, *size)
            # torch.cat of the features and of the masks (infill3d.py:224-225) happens inside the layer
            dec = getattr(self, 'dec_{:d}'.format(i))
            h, m = dec([h, e], [m, em], n)
            if self.dropout is not None:
                h, m = self.dropout(h, m, n, dec.conv.out_channels)
        dims = (h.shape[0] // n, h.shape[1], h.shape[2])
        h, m = self.last_conv(h, m, n)
        return from_folded(h, n, self.output_channels), from_folded(mask_tensor(m, n, dims, self.output_channels, dt), n,
                                                                    self.output_channels)
