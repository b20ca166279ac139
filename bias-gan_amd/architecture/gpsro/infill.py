"""Partial-convolution U-Net in 2-D on the MI355X kernels ("2-D shapes").

Host-side mirror of the reference's ``architecture/gpsro/infill.py`` (PCBActiv, PConvUNet) and
``architecture/common/partialconv2d.py`` (PartialConv2d with multi_channel=True, return_mask=True): same class names,
constructor arguments and state_dict keys.  Built from the pieces of infill3d.py: an image batch is the folded volume
with depth 1 per sample, the mask window is planar (bg_mask_window, planar = 1), masks past the first layer are one
value per pixel (ops.RowsMask).
"""
from __future__ import annotations

import torch.nn as nn

from ... import ops
from ...runtime import BGModule, pad_to, vec_of
from .deeplab import Conv2d, apply_norm
from .infill3d import ACT_LEAKY, ACT_NONE, ACT_RELU, mask_tensor, partial_forward


class PartialConv2d(Conv2d):
    """nn.Conv2d subclass of the reference (partialconv2d.py:14-89) for multi_channel=True, return_mask=True."""

    def __init__(self, *args, multi_channel=True, return_mask=True, eps=1.e-8, **kwargs):
        if not (multi_channel and return_mask):
            raise NotImplementedError("only PartialConv2d(multi_channel=True, return_mask=True) occurs on this path")
        super().__init__(*args, **kwargs)
        self.multi_channel, self.return_mask, self.eps = True, True, eps

    def forward(self, xs, masks):
        """xs, masks as in infill3d.partial_forward with NHWC images [N,H,W,C]."""
        n = (xs[0] if isinstance(xs, (list, tuple)) else xs).shape[0]
        return partial_forward(self, lambda x: Conv2d.forward(self, x, with_bias=False), xs, masks, n, planar=True)


class PCBActiv(BGModule):
    """PartialConv2d -> normalizer -> ReLU | LeakyReLU(0.2) | none (infill.py:100-136)."""

    def __init__(self, in_ch, out_ch, normalizer=nn.BatchNorm2d, sample='none-3', activ='relu', conv_bias=False):
        super().__init__()
        k, s, p = {'down-5': (5, 2, 2), 'down-7': (7, 2, 3), 'down-3': (3, 2, 1), 'point-1': (1, 1, 0)}.get(sample, (3, 1, 1))
        self.conv = PartialConv2d(in_ch, out_ch, k, s, p, bias=conv_bias, multi_channel=True, return_mask=True, eps=1e-6)
        if normalizer is not None:
            self.bn = normalizer(out_ch)
        if activ == 'relu':
            self.activation = nn.ReLU()
        elif activ == 'leaky':
            self.activation = nn.LeakyReLU(negative_slope=0.2)

    def forward(self, x, mask):
        h, m = self.conv(x, mask)
        act = ACT_NONE
        if hasattr(self, 'activation'):
            act = ACT_RELU if isinstance(self.activation, nn.ReLU) else ACT_LEAKY
        if hasattr(self, 'bn'):
            h = apply_norm(self, self.bn, h, act=act)
        elif act:
            h = ops.NormActFn.apply(h, None, None, None, None, None, None, None, None, "identity", False, act, 0.0, 0.0)
        return h, m


class PConvUNet(BGModule):
    """Partial-convolution U-Net (infill.py:139-210); needs input_channels == output_channels like the reference
    (dec_1 is built for 64 + output_channels and fed the network input)."""

    def __init__(self, layer_size=7, input_channels=3, output_channels=3, upsampling_mode='nearest', normalizer=nn.BatchNorm2d,
                 compute_dtype=None):
        super().__init__()
        if upsampling_mode not in ('nearest', 'bilinear'):
            raise NotImplementedError("the HIP path builds upsampling_mode 'nearest' and 'bilinear'")
        self.freeze_enc_bn = False
        self.upsampling_mode, self.layer_size = upsampling_mode, layer_size
        self.input_channels, self.output_channels = input_channels, output_channels
        self.enc_1 = PCBActiv(input_channels, 64, sample='down-3', normalizer=None)
        self.enc_2 = PCBActiv(64, 128, sample='down-3', normalizer=normalizer)
        self.enc_3 = PCBActiv(128, 256, sample='down-3', normalizer=normalizer)
        self.enc_4 = PCBActiv(256, 512, sample='down-3', normalizer=normalizer)
        for i in range(4, self.layer_size):
            setattr(self, 'enc_{:d}'.format(i + 1), PCBActiv(512, 512, sample='down-3', normalizer=normalizer))
        for i in range(4, self.layer_size):
            setattr(self, 'dec_{:d}'.format(i + 1), PCBActiv(512 + 512, 512, activ='leaky', normalizer=normalizer))
        self.dec_4 = PCBActiv(512 + 256, 256, activ='leaky', normalizer=normalizer)
        self.dec_3 = PCBActiv(256 + 128, 128, activ='leaky', normalizer=normalizer)
        self.dec_2 = PCBActiv(128 + 64, 64, activ='leaky', normalizer=normalizer)
        self.dec_1 = PCBActiv(64 + output_channels, 32, activ='leaky', normalizer=normalizer)
        # "for 1x1 resolution" (infill.py:160-163): a pointwise encoder of the input joins the last layer
        self.input_enc_1 = PCBActiv(input_channels, 64, activ='leaky', sample='point-1', normalizer=normalizer)
        self.last_conv = PCBActiv(64 + 32, output_channels, normalizer=None, activ=None, sample='point-1', conv_bias=True)
        if compute_dtype is not None:
            self.set_compute_dtype(compute_dtype)

    def forward(self, input, input_mask):
        """NCHW fp32 input and 0/1 mask [N,C,H,W] -> (output [N,Cout,H,W] fp32, its mask)."""
        dt = self.compute_dtype()
        n, c = input.shape[0], input.shape[1]
        cp = pad_to(c, vec_of(dt))
        x0, m0 = ops.ToInternal.apply(input, cp, dt), ops.ToInternal.apply(input_mask, cp, dt)
        hs, ms = {0: x0}, {0: m0}
        for i in range(1, self.layer_size + 1):
            hs[i], ms[i] = getattr(self, 'enc_{:d}'.format(i))(hs[i - 1], ms[i - 1])
        h, m = hs[self.layer_size], ms[self.layer_size]
        for i in range(self.layer_size, 0, -1):
            e, em = hs[i - 1], ms[i - 1]
            src, size = (1, h.shape[1], h.shape[2]), (1, e.shape[1], e.shape[2])
            # 'bilinear' (align_corners unset, infill.py:193-195) = the trilinear kernel on volumes of depth 1
            h = (ops.NearestResize3dFn if self.upsampling_mode == 'nearest' else ops.TrilinearResize3dFn).apply(h, n, *size)
            m = ops.nearest_rows(m, n, src, size)
            h, m = getattr(self, 'dec_{:d}'.format(i))([h, e], [m, em])
        hin, hin_mask = self.input_enc_1(x0, m0)
        h, m = self.last_conv([h, hin], [m, hin_mask])
        dims = (1, h.shape[1], h.shape[2])
        return (ops.FromInternal.apply(h, self.output_channels),
                ops.FromInternal.apply(mask_tensor(m, n, dims, self.output_channels, dt), self.output_channels))
