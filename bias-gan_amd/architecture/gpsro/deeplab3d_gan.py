"""3-D Generator / Discriminator / gradient_penalty with the reference's API
(architecture/gpsro/deeplab3d_gan.py) on the MI355X kernels."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ... import ops
from ... import _lib as L
from ...runtime import BGModule, pad_to, vec_of
from .deeplab3d import *  # noqa: F401,F403
from .deeplab3d import Conv3d, DeepLab3d, Xception3d, to_folded
from .deeplab_gan import Linear


class Discriminator(BGModule):
    """Xception3d features -> mean over (D,H,W) -> Linear(2048, 1) -> (logits, sigmoid) (deeplab3d_gan.py:12-44)."""

    def __init__(self, n_input=3, os=16, pretrained=False, normalizer=nn.LayerNorm, compute_dtype=None):
        super().__init__()
        self.n_input = n_input
        self.xception_features = Xception3d(n_input, os, pretrained, normalizer)
        self.linear = Linear(2048, 1)
        self.sigmoid = nn.Sigmoid()
        self._init_weight()
        if compute_dtype is not None:
            self.set_compute_dtype(compute_dtype)

    def _init_weight(self):
        # deeplab3d_gan.py:51-67: every Conv3d re-initialised (n = k*k*Cout, two kernel extents), xavier Linear
        gain = nn.init.calculate_gain("leaky_relu", 0.2)
        for m in self.modules():
            if isinstance(m, Conv3d):
                nn.init.normal_(m.weight, mean=0.0, std=gain / math.sqrt(m.kernel_size[0] * m.kernel_size[1] * m.out_channels))
            elif isinstance(m, Linear):
                nn.init.xavier_uniform_(m.weight, 1.0)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, input):
        dt = self.compute_dtype()
        n, c = input.shape[0], input.shape[1]
        assert c == self.n_input, f"expected {self.n_input} input channels, got {c}"
        xi = to_folded(input, pad_to(c, vec_of(dt)), dt)
        f, _ = self.xception_features.forward_folded(xi, n, want_low=False)
        pooled = ops.GlobalAvgPoolFn.apply(f, n)                       # [N,1,1,2048]: torch.mean(x, dim=[2,3,4])
        a, lin = self.arena(), self.linear
        logits = ops.LinearHeadFn.apply(pooled, lin.weight, lin.bias, a, a.by_param[id(lin.weight)], a.by_param[id(lin.bias)])
        return logits, torch.sigmoid(logits)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm3d):
                m.eval()


class Generator(BGModule):
    """noise concat + DeepLab3d (deeplab3d_gan.py:69-100)."""

    def __init__(self, n_input, n_output, upsampler_type, noise_type, noise_dimensions, os=16, pretrained=False,
                 normalizer=nn.BatchNorm3d, compute_dtype=None):
        super().__init__()
        self.noise_dimensions = noise_dimensions
        if noise_type == "Uniform":
            self.dist = torch.distributions.uniform.Uniform(0., 1.)
        elif noise_type == "Normal":
            self.dist = torch.distributions.normal.Normal(0., 1.)
        else:
            raise NotImplementedError("Error, noise type {} not supported.".format(noise_type))
        self.model = DeepLab3d(n_input=(n_input + noise_dimensions), n_output=n_output, os=os,
                               upsampler_type=upsampler_type, pretrained=pretrained, normalizer=normalizer)
        if compute_dtype is not None:
            self.set_compute_dtype(compute_dtype)

    def forward(self, input_raw, noise=None):
        if self.noise_dimensions > 0:
            if noise is None:   # drawn on the host RNG stream and copied, like the reference (deeplab3d_gan.py:92)
                noise = self.dist.rsample((input_raw.shape[0], self.noise_dimensions) + tuple(input_raw.shape[2:])).to(input_raw.device)
            input = torch.cat((input_raw, noise), dim=1)
        else:
            input = input_raw
        return self.model(input)


def gradient_penalty(critic, images_fake, images_real, eta=None):
    """WGAN-GP term as the reference's 3-D version computes it (deeplab3d_gan.py:103-125): the gradient is
    flattened per SAMPLE before the 2-norm (the 2-D one norms per pixel over channels); first order only."""
    if eta is None:
        eta = torch.distributions.uniform.Uniform(0., 1.).rsample((images_fake.shape[0], 1, 1, 1, 1))
    eta = eta.to(images_fake.device)
    images_interpol = (eta * images_fake.detach() + (1. - eta) * images_real.detach()).requires_grad_(True)
    params = [p for p in critic.parameters() if p.requires_grad]
    for p in params:
        p.requires_grad_(False)
    try:
        logits_interpol, _ = critic(images_interpol)
        gradients = torch.autograd.grad(outputs=logits_interpol, inputs=images_interpol,
                                        grad_outputs=torch.ones_like(logits_interpol))[0]
    finally:
        for p in params:
            p.requires_grad_(True)
    g = gradients.contiguous().float()
    n = g.shape[0]
    out = torch.zeros(1, dtype=torch.float32, device=g.device)
    # bg_gp_penalty norms over "channels" per (sample, pixel): with one pixel per sample and every element of the
    # sample as a channel it is the per-sample flattened norm
    L.call("bg_gp_penalty", g.data_ptr(), n, g[0].numel(), 1, 1.0 / n, out.data_ptr())
    return out.view(())
