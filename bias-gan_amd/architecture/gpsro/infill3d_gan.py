"""Partial-convolution GAN in 3-D with the reference's API (architecture/gpsro/infill3d_gan.py) on the MI355X kernels."""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import ops
from ... import _lib as L
from ...runtime import BGModule, pad_to, vec_of
from .deeplab3d import Conv3d, to_folded
from .infill3d import PCBActiv3d, PConvUNet3d as Generator  # noqa: F401  (the reference re-exports it under this name)


class _LinearNoBias(BGModule):
    def __init__(self, in_features, out_features):
        super().__init__()
        assert out_features == 1
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = None


class Discriminator(BGModule):
    """PCBActiv3d encoder -> mean over (D,H,W) -> Linear(512, 1, bias=False) (infill3d_gan.py:16-63).  Quirk of the
    reference kept: all `layer_size` layers run, the logits come from the output of layer layer_size-1."""

    def __init__(self, layer_size=7, input_channels=3, normalizer=nn.BatchNorm3d, compute_dtype=None):
        super().__init__()
        self.layer_size, self.input_channels = layer_size, input_channels
        self.enc_1 = PCBActiv3d(input_channels, 64, sample='down-3', normalizer=None)
        self.enc_2 = PCBActiv3d(64, 128, sample='down-3', normalizer=normalizer)
        self.enc_3 = PCBActiv3d(128, 256, sample='down-3', normalizer=normalizer)
        self.enc_4 = PCBActiv3d(256, 512, sample='down-3', normalizer=normalizer)
        for i in range(4, self.layer_size):
            setattr(self, 'enc_{:d}'.format(i + 1), PCBActiv3d(512, 512, sample='down-3', normalizer=normalizer))
        self.linear = _LinearNoBias(512, 1)
        self.sigmoid = nn.Sigmoid()
        for m in self.modules():
            if isinstance(m, Conv3d):
                nn.init.kaiming_normal_(m.weight)
        nn.init.kaiming_uniform_(self.linear.weight, a=5 ** 0.5)     # nn.Linear's default
        self._zero_bias = None
        if compute_dtype is not None:
            self.set_compute_dtype(compute_dtype)

    def forward(self, input, input_mask):
        dt = self.compute_dtype()
        n, c = input.shape[0], input_mask.shape[1]
        if input.shape[1] != c:        # `input * mask` broadcasts a 1-channel volume over the mask's channels
            input = input.expand(-1, c, -1, -1, -1)
        cp = pad_to(c, vec_of(dt))
        h, m = to_folded(input, cp, dt), to_folded(input_mask, cp, dt)
        feat = None
        for i in range(1, self.layer_size + 1):
            if i == self.layer_size:
                feat = h                       # h_{layer_size-1} feeds the head (infill3d_gan.py:58-60)
            h, m = getattr(self, 'enc_{:d}'.format(i))(h, m, n)
        pooled = ops.GlobalAvgPoolFn.apply(feat, n)
        a, lin = self.arena(), self.linear
        if self._zero_bias is None or self._zero_bias.device != pooled.device:
            self._zero_bias = torch.zeros(1, dtype=torch.float32, device=pooled.device)
        logits = ops.LinearHeadNoBiasFn.apply(pooled, lin.weight, a, a.by_param[id(lin.weight)], self._zero_bias)
        return logits, torch.sigmoid(logits)


class GAN(object):
    """infill3d_gan.py:66-85."""

    def __init__(self, input_channels=3, output_channels=3, gen_layer_size=7, disc_layer_size=7, upsampling_mode='nearest',
                 gen_normalizer=nn.BatchNorm3d, disc_normalizer=nn.BatchNorm3d):
        self.generator = Generator(layer_size=gen_layer_size, input_channels=input_channels, output_channels=output_channels,
                                   upsampling_mode='nearest', normalizer=gen_normalizer)
        self.discriminator = Discriminator(layer_size=disc_layer_size, input_channels=input_channels, normalizer=disc_normalizer)

    def generate(self, inp, mask):
        x, _ = self.generator(inp, mask)
        return x

    def discriminate(self, inp, mask):
        return self.discriminator(inp, mask)


class InpaintingLoss:
    """utils/losses.py:47-98 without a feature extractor: {'hole', 'valid', 'tv'}.  The masked distances use the
    pixel-loss kernels with the mask as weight (f(m*d) = m*f(d) for a 0/1 mask and f(0) = 0)."""

    def __init__(self, loss_type="smooth-l1", extractor=None):
        if extractor is not None:
            raise NotImplementedError("the perceptual / style terms need a VGG extractor; not on this path")
        try:
            self.kind = {"l1": 0, "smooth-l1": 1, "l2": 2}[loss_type]
        except KeyError:
            raise NotImplementedError(f"Error: loss_type {loss_type} not implemented.")

    def __call__(self, input, output, gt, mask):
        inv = 1.0 / output.numel()
        hole = ops.L1LossFn.apply(output, gt, 1.0 - mask, inv, self.kind)
        valid = ops.L1LossFn.apply(output, gt, mask, inv, self.kind)
        return {"hole": hole, "valid": valid, "tv": ops.TVLossCompFn.apply(output, input, mask)}
