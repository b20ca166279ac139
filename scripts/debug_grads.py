"""Debug aid: per-parameter gradient error of the HIP path vs the fp64 CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn as nn
import bias_gan_amd
from bias_gan_amd import ops
from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg
from oracle import gan_oracle as orc

which = sys.argv[1] if len(sys.argv) > 1 else "D"
dtype = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else torch.bfloat16
c, h, w, n = 4, 64, 64, 2
torch.set_num_threads(16)
x, y = orc.synthetic_fields(n, c, h, w, 103)
if which == "D":
    spec = orc.discriminator_spec(c, h, w, "batch")
    P = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in orc.fill_state(spec, 3).items()}
    keys = orc.trainable_keys(spec)
    for k in keys: P[k].requires_grad_(True)
    xr = x.double().requires_grad_(True)
    logits, _ = orc.discriminator(P, xr, orc.NormCtx("batch", True))
    tgt = torch.linspace(0.1, 0.9, n).reshape(-1, 1)
    orc.bce_logits(logits, tgt.double()).backward()
    M = dxg.Discriminator(c, normalizer=nn.BatchNorm2d, input_size=(h, w), compute_dtype=dtype)
    M.load_state_dict(orc.fill_state(spec, 3)); M.to("cuda").train()
    xg = x.cuda().requires_grad_(True)
    lg, _ = M(xg)
    ops.BCEWithLogitsFn.apply(lg, tgt.cuda()).backward()
    print("logits", lg.detach().cpu().numpy().ravel(), logits.detach().numpy().ravel())
else:
    spec = orc.generator_spec(c, c, 0, "batch")
    P = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in orc.fill_state(spec, 3).items()}
    keys = orc.trainable_keys(spec)
    for k in keys: P[k].requires_grad_(True)
    xr = x.double().requires_grad_(True)
    out = orc.generator(P, xr, orc.NormCtx("batch", True))
    (out - y.double()).abs().mean().backward()
    M = dxg.Generator(c, c, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d, compute_dtype=dtype)
    M.load_state_dict(orc.fill_state(spec, 3)); M.to("cuda").train()
    xg = x.cuda().requires_grad_(True)
    og = M(xg)
    from bias_gan_amd.utils import losses
    losses.L1Loss()(og, y.cuda()).backward()
    print("out err", (og.detach().cpu().double() - out.detach()).abs().max().item() / out.detach().abs().max().item())
print("dx err", (xg.grad.cpu().double() - xr.grad).abs().max().item() / xr.grad.abs().max().item())
dd = (xg.grad.cpu().double() - xr.grad)
print("dx rms err", (dd.pow(2).mean().sqrt() / xr.grad.pow(2).mean().sqrt()).item(), "n elems with err > 1e-3 max:", int((dd.abs() > 1e-3 * xr.grad.abs().max()).sum()), "of", dd.numel())
named = dict(M.named_parameters())
rows = []
for k in keys:
    g, r = named[k].grad.cpu().double(), P[k].grad
    rows.append(((g - r).abs().max().item() / (r.abs().max().item() + 1e-30), k, ((g - r).pow(2).mean().sqrt() / (r.pow(2).mean().sqrt() + 1e-30)).item()))
rows.sort(reverse=True)
for e, k, s in rows[:25]:
    print(f"{e:10.3e}  {s:10.3e}  {k}")
print("median max-rel", np.median([r[0] for r in rows]), "worst rms-rel", max(r[2] for r in rows))
