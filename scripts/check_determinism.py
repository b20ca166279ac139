"""Run-to-run spread of one network's forward / backward in one process (fp32): logits and the gradient arena of three
identical passes.  Atomics give ~1e-6; anything larger is worth a look.  usage: check_determinism.py [HW] [N] [g|d]"""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.nn as nn
import bias_gan_amd  # noqa
from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg
from bias_gan_amd.runtime import StatsPool
from oracle import gan_oracle as orc

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
which = sys.argv[3] if len(sys.argv) > 3 else "d"
c = 4
with contextlib.redirect_stdout(io.StringIO()):
    if which == "d":
        net = dxg.Discriminator(c, normalizer=nn.BatchNorm2d, input_size=(hw, hw), compute_dtype=torch.float32)
        net.load_state_dict(orc.fill_state(orc.discriminator_spec(c, hw, hw, "batch"), 32))
    else:
        net = dxg.Generator(c, c, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d, compute_dtype=torch.float32)
        net.load_state_dict(orc.fill_state(orc.generator_spec(c, c, 0, "batch", upsampler="Interpolate"), 31))
net.to("cuda:0").train()
x, y = (t.to("cuda:0") for t in orc.synthetic_fields(n, c, hw, hw, 77))
outs, grads = [], []
for rep in range(3):
    StatsPool.reset_all()
    net.zero_grad(set_to_none=True)
    a = net.arena()
    a.zero_grad(); a.attach_grads()
    o = net(x)
    o = o[0] if isinstance(o, tuple) else o
    (o.float() * torch.linspace(0.5, 1.5, o.numel(), device=o.device).view_as(o)).mean().backward()
    torch.cuda.synchronize()
    outs.append(o.detach().double().cpu().numpy().copy())
    grads.append(a.grad.double().cpu().numpy().copy())
def rel(p, q):
    return np.abs(p - q).max() / (np.abs(q).max() + 1e-30), np.sqrt(((p - q) ** 2).mean()) / (np.sqrt((q ** 2).mean()) + 1e-30)
for i in (1, 2):
    print(f"[{which} {hw}x{hw} n={n}] run {i} vs 0: output max-rel %.2e rms-rel %.2e | gradient arena max-rel %.2e rms-rel %.2e" % (*rel(outs[i], outs[0]), *rel(grads[i], grads[0])))
