"""CPU restatement of the reference's partial-convolution U-Net GAN in 3-D --
TEST INFRASTRUCTURE ONLY.  Reference files under /root/reference/src/deepCam/: architecture/common/partialconv3d.py,
architecture/gpsro/infill3d.py (PCBActiv3d, PConvUNet3d), architecture/gpsro/infill3d_gan.py (Discriminator),
utils/losses.py:40-98 (InpaintingLoss, total_variation_loss), gpsro_train/infill3d_gan_module.py:290-375 (loop).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned against the reference by tests/golden/infill3d_*.npz (tests/golden/make_golden.py, `infill3d` target).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from .gan_oracle import NormCtx, State, _norm_entries, fill_state, norm, trainable_keys  # noqa: F401

PC_EPS = 1e-6   # PCBActiv3d builds every PartialConv3d with eps=1e-6 (infill3d.py:86-99)


def _enc_channels(i: int, cin: int):
    """(in, out) channels of enc_i (infill3d.py:143-150)."""
    table = {1: (cin, 64), 2: (64, 128), 3: (128, 256), 4: (256, 512)}
    return table.get(i, (512, 512))


def _dec_channels(i: int, cin: int):
    """(in, out) channels of dec_i (infill3d.py:152-158)."""
    table = {4: (512 + 256, 256), 3: (256 + 128, 128), 2: (128 + 64, 64), 1: (64 + cin, 32)}
    return table.get(i, (512 + 512, 512))


def _pcb_entries(prefix, cin, cout, k, norm_kind, bias=False):
    s = [(prefix + ".conv.weight", (cout, cin, k, k, k), "conv")]
    if bias:
        s.append((prefix + ".conv.bias", (cout,), "bias"))
    if norm_kind is not None:
        s += _norm_entries(prefix + ".bn", cout, norm_kind)
    return s


def unet3d_spec(cin: int, cout: int, layer_size: int = 7, norm_kind: str = "batch"):
    """state_dict entries of PConvUNet3d in registration order (infill3d.py:139-165)."""
    s = []
    for i in range(1, max(layer_size, 4) + 1):
        ci, co = _enc_channels(i, cin)
        s += _pcb_entries(f"enc_{i}", ci, co, 3, None if i == 1 else norm_kind)
    for i in range(5, layer_size + 1):
        ci, co = _dec_channels(i, cin)
        s += _pcb_entries(f"dec_{i}", ci, co, 3, norm_kind)
    for i in (4, 3, 2, 1):
        ci, co = _dec_channels(i, cin)
        s += _pcb_entries(f"dec_{i}", ci, co, 3, norm_kind)
    s += _pcb_entries("last_conv", 32, cout, 1, None, bias=True)
    return s


def unet2d_spec(cin: int, cout: int, layer_size: int = 7, norm_kind: str = "batch"):
    """state_dict entries of the 2-D PConvUNet in registration order (infill.py:140-163); kernels are [.,.,k,k]."""
    s = []
    for key, shape, role in unet3d_spec(cin, cout, layer_size, norm_kind):
        if key.startswith("last_conv"):
            continue
        if key == "dec_1.conv.weight":          # 64 + output_channels (infill.py:158)
            shape = (32, 64 + cout) + shape[2:]
        s.append((key, shape[:-1] if role == "conv" else shape, role))
    s += [(k_, sh[:-1] if r == "conv" else sh, r) for k_, sh, r in _pcb_entries("input_enc_1", cin, 64, 1, norm_kind)]
    s += [(k_, sh[:-1] if r == "conv" else sh, r) for k_, sh, r in _pcb_entries("last_conv", 64 + 32, cout, 1, None, bias=True)]
    return s


def unet2d(P: State, x, mask, layer_size: int, ctx: NormCtx, upsampling_mode: str = "nearest"):
    """2-D PConvUNet.forward (infill.py:165-210); upsampling_mode 'nearest' | 'bilinear' for the features (:193-195), the
    mask always 'nearest' (:196-198)."""
    hs, ms = {0: x}, {0: mask}
    for i in range(1, layer_size + 1):
        hs[i], ms[i] = pcb_activ(P, f"enc_{i}", hs[i - 1], ms[i - 1], 3, 2, 1, i != 1, "relu", ctx)
    h, m = hs[layer_size], ms[layer_size]
    for i in range(layer_size, 0, -1):
        size = tuple(hs[i - 1].shape[2:])
        h = F.interpolate(h, size=size, mode=upsampling_mode)
        m = F.interpolate(m, size=size, mode="nearest")
        h, m = pcb_activ(P, f"dec_{i}", torch.cat([h, hs[i - 1]], dim=1), torch.cat([m, ms[i - 1]], dim=1), 3, 1, 1, True,
                         "leaky", ctx)
    hin, hin_mask = pcb_activ(P, "input_enc_1", x, mask, 1, 1, 0, True, "leaky", ctx)
    return pcb_activ(P, "last_conv", torch.cat([h, hin], dim=1), torch.cat([m, hin_mask], dim=1), 1, 1, 0, False, None, ctx)


def disc3d_spec(cin: int, layer_size: int = 7, norm_kind: str = "batch"):
    """state_dict entries of infill3d_gan.Discriminator (infill3d_gan.py:17-30)."""
    s = []
    for i in range(1, max(layer_size, 4) + 1):
        ci, co = _enc_channels(i, cin)
        s += _pcb_entries(f"enc_{i}", ci, co, 3, None if i == 1 else norm_kind)
    s.append(("linear.weight", (1, 512), "linear"))
    return s


def partial_conv3d(x, mask, w, b, stride, pad):
    """PartialConv3d.forward with multi_channel=True, return_mask=True (partialconv3d.py:49-92); with a 4-D weight
    the same function is PartialConv2d.forward (partialconv2d.py:49-89), which differs in the rank only."""
    cout, cin, k, nd = w.shape[0], w.shape[1], w.shape[2], w.dim() - 2
    conv = F.conv3d if nd == 3 else F.conv2d
    with torch.no_grad():
        ones = torch.ones((1, cin) + (k,) * nd, dtype=mask.dtype)
        upd = conv(mask, ones, None, stride, pad).expand(-1, cout, *([-1] * nd))   # identical for every output channel
        ratio = float(cin * k ** nd) / (upd + PC_EPS)
        upd = torch.clamp(upd, 0, 1)
        ratio = ratio * upd
    raw = conv(x * mask, w, b, stride, pad)
    if b is not None:
        bv = b.view((1, -1) + (1,) * nd)
        out = ((raw - bv) * ratio + bv) * upd
    else:
        out = raw * ratio
    return out, upd


def pcb_activ(P: State, key: str, x, mask, k, stride, pad, has_norm, act, ctx: NormCtx):
    """PCBActiv3d.forward (infill3d.py:108-114): partial conv -> normalizer -> ReLU | LeakyReLU(0.2) | none."""
    h, m = partial_conv3d(x, mask, P[key + ".conv.weight"], P.get(key + ".conv.bias"), stride, pad)
    if has_norm:
        h = norm(P, key + ".bn", h, ctx)
    if act == "relu":
        h = F.relu(h)
    elif act == "leaky":
        h = F.leaky_relu(h, 0.2)
    return h, m


def pc_dropout3d(x, mask, keep, p: float):
    """PCDropout3d.forward in training mode (infill3d.py:119-131) for a given draw of its nn.Dropout3d: keep [N,C] in {0,1}
    (Dropout3d zeroes whole (sample, channel) maps of the MASK with probability p and scales the rest by 1/(1-p))."""
    scale = 1.0 - p
    dropped = mask * keep.view(keep.shape[0], keep.shape[1], 1, 1, 1) / scale      # self.dropout(mask)
    mask_d = torch.round(dropped * scale)
    drop_vals = mask - mask_d
    return x * (1.0 - drop_vals) / scale, mask_d


def unet3d(P: State, x, mask, layer_size: int, ctx: NormCtx, upsampling_mode: str = "nearest", dropout_p: float = 0.0, keeps=None):
    """PConvUNet3d.forward (infill3d.py:177-239).  upsampling_mode: 'nearest' | 'trilinear' (the mask is always resized
    with 'nearest', :221-222).  dropout_p > 0 (training mode): `keeps` lists the Dropout3d draws [N,C] in call order --
    one per encoder layer, then one per decoder layer."""
    keeps = list(keeps) if keeps is not None else None
    hs, ms = {0: x}, {0: mask}
    for i in range(1, layer_size + 1):
        hs[i], ms[i] = pcb_activ(P, f"enc_{i}", hs[i - 1], ms[i - 1], 3, 2, 1, i != 1, "relu", ctx)
        if dropout_p > 0.0:
            hs[i], ms[i] = pc_dropout3d(hs[i], ms[i], keeps.pop(0), dropout_p)
    h, m = hs[layer_size], ms[layer_size]
    for i in range(layer_size, 0, -1):
        size = tuple(hs[i - 1].shape[2:])
        h = F.interpolate(h, size=size, mode=upsampling_mode)
        m = F.interpolate(m, size=size, mode="nearest")
        h = torch.cat([h, hs[i - 1]], dim=1)
        m = torch.cat([m, ms[i - 1]], dim=1)
        h, m = pcb_activ(P, f"dec_{i}", h, m, 3, 1, 1, True, "leaky", ctx)
        if dropout_p > 0.0:
            h, m = pc_dropout3d(h, m, keeps.pop(0), dropout_p)
    return pcb_activ(P, "last_conv", h, m, 1, 1, 0, False, None, ctx)


def disc3d(P: State, x, mask, layer_size: int, ctx: NormCtx):
    """infill3d_gan.Discriminator.forward (infill3d_gan.py:43-63).  Quirk kept: every enc layer runs, but the
    logits come from the mean of the output of layer layer_size-1 (`enc_h_key = 'h_{i-1}'` after the loop)."""
    hs, ms = {0: x}, {0: mask}
    for i in range(1, layer_size + 1):
        hs[i], ms[i] = pcb_activ(P, f"enc_{i}", hs[i - 1], ms[i - 1], 3, 2, 1, i != 1, "relu", ctx)
    feat = hs[layer_size - 1].mean(dim=(2, 3, 4))
    logits = F.linear(feat, P["linear.weight"])
    return logits, torch.sigmoid(logits)


def total_variation_loss(image):
    """utils/losses.py:40-44, applied as written: shifts along dims 3 and 2 (W and H of a 4-D image batch; H and D of
    a 5-D volume batch)."""
    return (image[:, :, :, :-1] - image[:, :, :, 1:]).abs().mean() + (image[:, :, :-1, :] - image[:, :, 1:, :]).abs().mean()


def inpainting_loss(inp, out, gt, mask, loss_type: str = "smooth-l1"):
    """InpaintingLoss.forward without a feature extractor (utils/losses.py:62-98) -> {'hole','valid','tv'}."""
    dist = {"l1": F.l1_loss, "smooth-l1": F.smooth_l1_loss, "l2": F.mse_loss}[loss_type]
    comp = mask * inp + (1 - mask) * out
    return {"hole": dist((1. - mask) * out, (1. - mask) * gt), "valid": dist(mask * out, mask * gt),
            "tv": total_variation_loss(comp)}


def synthetic_infill(n, c, d, h, w, seed, hole=0.3):
    """(input with holes zeroed, ground truth, 0/1 mask per channel)."""
    g = torch.Generator().manual_seed(seed)
    gt = torch.randn((n, c, d, h, w), generator=g)
    mask = (torch.rand((n, c, d, h, w), generator=g) > hole).float()
    return gt * mask, gt, mask


def infill_update_flags(step: int, warmup: int, d_acc_avg: float, acc_min: float, acc_max: float):
    """(train_generator, train_discriminator) of infill3d_gan_module.py:294-311."""
    if step < warmup:
        return True, False
    if d_acc_avg > acc_max:       # discriminator too good
        return True, False
    if d_acc_avg < acc_min:       # discriminator too bad
        return False, True
    return True, True


class InfillGANStep:
    """One iteration of infill3d_gan_module.py:272-375 with the results-neutral waste removed (the D part does not
    back-propagate into the generator; the G part keeps no discriminator weight gradients).  Every forward runs in
    train mode -- two generator and three discriminator passes per iteration, whatever the update flags say -- so
    the BatchNorm running statistics move exactly as in the reference.  The discriminator sees the 1-channel
    volume against the (1 + noise)-channel mask: `input * mask` broadcasts it (partialconv3d.py:77)."""

    def __init__(self, PG: State, PD: State, g_keys, d_keys, g_layers: int, d_layers: int, loss_weights: dict,
                 loss_type="smooth-l1", warmup=0, acc_min=0.0, acc_max=1.0, lr_g=1e-4, lr_d=1e-4, eps=1e-8,
                 weight_decay=1e-5, norm_kind="batch", decoupled=False):
        from .gan_oracle import Adam
        self.PG, self.PD, self.g_keys, self.d_keys = PG, PD, g_keys, d_keys
        self.g_layers, self.d_layers = g_layers, d_layers
        self.w, self.loss_type = dict(loss_weights), loss_type
        self.warmup, self.acc_min, self.acc_max = warmup, acc_min, acc_max
        self.ctx = NormCtx(norm_kind, True)
        self.g_opt = Adam(g_keys, lr_g, eps, weight_decay, decoupled=decoupled)     # decoupled: torch.optim.AdamW
        self.d_opt = Adam(d_keys, lr_d, eps, weight_decay, decoupled=decoupled)
        self.step_count, self.d_acc_avg = 0, 0.5
        self.last_flags = (True, True)
        self.last_terms = {}

    @staticmethod
    def _leaves(P, keys):
        Q = dict(P)
        for k in keys:
            Q[k] = P[k].detach().requires_grad_(True)
        return Q

    @staticmethod
    def _sync_buffers(Q, P):
        for k in P:
            if Q[k] is not P[k] and not Q[k].requires_grad:
                P[k] = Q[k]

    def step(self, inputs_raw, outputs_real, masks_raw, noise, labels):
        """inputs_raw, outputs_real, masks_raw: [N,1,D,H,W]; noise [N,nd,D,H,W]; labels = draw_d_labels(N)."""
        from .gan_oracle import gan_d_loss, gan_g_loss
        inputs = torch.cat((inputs_raw, noise), dim=1)
        masks = torch.cat((masks_raw, torch.ones_like(noise)), dim=1)
        train_g, train_d = infill_update_flags(self.step_count, self.warmup, self.d_acc_avg, self.acc_min, self.acc_max)
        self.last_flags = (train_g, train_d)
        # ---- discriminator part (:314-341)
        with torch.no_grad():
            fake, _ = unet3d(self.PG, inputs, masks, self.g_layers, self.ctx)
        Q = self._leaves(self.PD, self.d_keys)
        with torch.set_grad_enabled(train_d):
            lr_, _ = disc3d(Q, outputs_real, masks, self.d_layers, self.ctx)
            lf_, _ = disc3d(Q, fake, masks, self.d_layers, self.ctx)
            lab_f, lab_r, swap = labels
            d_loss = gan_d_loss("ModifiedMinMax", lr_, lf_, lab_f, lab_r, swap) * self.w["adv"]
        self.d_acc_avg = float(0.5 * ((lr_ > 0).float().mean() + (lf_ <= 0).float().mean()))
        self._sync_buffers(Q, self.PD)
        if train_d:
            grads = torch.autograd.grad(d_loss, [Q[k] for k in self.d_keys], allow_unused=True)
            self.d_opt.step(self.PD, {k: g for k, g in zip(self.d_keys, grads) if g is not None})
        # ---- generator part (:344-373)
        Q = self._leaves(self.PG, self.g_keys)
        with torch.set_grad_enabled(train_g):
            fake, _ = unet3d(Q, inputs, masks, self.g_layers, self.ctx)
            lf_, _ = disc3d(self.PD, fake, masks, self.d_layers, self.ctx)
            terms = inpainting_loss(inputs_raw, fake, outputs_real, masks_raw, self.loss_type)
            if self.step_count >= self.warmup:
                terms["adv"] = gan_g_loss("ModifiedMinMax", lf_)
            g_loss = sum(terms[k] * self.w[k] for k in terms)
        self._sync_buffers(Q, self.PG)
        if train_g:
            grads = torch.autograd.grad(g_loss, [Q[k] for k in self.g_keys], allow_unused=True)
            self.g_opt.step(self.PG, {k: g for k, g in zip(self.g_keys, grads) if g is not None})
        self.last_terms = {k: float(v.detach()) for k, v in terms.items()}
        self.step_count += 1
        return float(d_loss.detach()), float(g_loss.detach())
