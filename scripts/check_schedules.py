"""Gradient arenas of the first iteration under the trainer's scheduling options against the plain sequential schedule
(fp32, D's learning rate 0 so that the generator's gradient is comparable too): max-relative and rms-relative differences."""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BGAMD_STEP_GRAPH"] = "0"
import numpy as np
import torch, torch.nn as nn
import bias_gan_amd  # noqa
from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg
from bias_gan_amd.gpsro_train.train_gan import GANTrainer
from bias_gan_amd.utils import losses, parsing_helpers as ph
from oracle import gan_oracle as orc

DEV = "cuda:0"
c, h, w, n = 4, int(os.environ.get("HW", "64")), int(os.environ.get("HW", "64")), int(os.environ.get("NB", "2"))

def run(batched, ahead):
    with contextlib.redirect_stdout(io.StringIO()):
        G = dxg.Generator(c, c, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d, compute_dtype=torch.float32)
        G.load_state_dict(orc.fill_state(orc.generator_spec(c, c, 0, "batch", upsampler="Interpolate"), 31)); G.to(DEV).train()
        D = dxg.Discriminator(c, normalizer=nn.BatchNorm2d, input_size=(h, w), compute_dtype=torch.float32)
        D.load_state_dict(orc.fill_state(orc.discriminator_spec(c, h, w, "batch"), 32)); D.to(DEV).train()
    crit = losses.GANLoss("ModifiedMinMax", n, torch.device(DEV))
    g_opt = ph.get_optimizer(G.parameters(), "Adam", 1e-4, 1e-8, 1e-5)
    d_opt = ph.get_optimizer(D.parameters(), "Adam", 0.0, 1e-8, 0.0)
    tr = GANTrainer(G, D, g_opt, d_opt, crit, losses.L1Loss())
    tr._batched_d = batched
    tr._g_ahead_ok = ahead
    if not ahead and not batched:
        tr._side = None
    seen = {}
    for tag, net, opt in (("g", G, g_opt), ("d", D, d_opt)):
        orig = opt.step
        def rec(*a, _o=orig, _t=tag, _n=net, **k):
            torch.cuda.synchronize()
            seen.setdefault(_t, _n.arena().grad.double().clone())
            return _o(*a, **k)
        opt.step = rec
    torch.manual_seed(3)
    x, y = (t.to(DEV) for t in orc.synthetic_fields(n, c, h, w, 77))
    labels = crit.draw_labels()
    dl, gl = tr.step(x, y, labels=labels)
    torch.cuda.synchronize()
    return seen, float(dl), float(gl)

def rel(a, b):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30), np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30)

base, dl0, gl0 = run(False, False)
again, dl1, gl1 = run(False, False)
print("plain twice        : d %.2e / %.2e   g %.2e / %.2e   d_loss %.7f %.7f  g_loss %.6f %.6f" % (*rel(again["d"], base["d"]), *rel(again["g"], base["g"]), dl0, dl1, gl0, gl1))
for name, b_, a_ in (("batched D only    ", True, False), ("G ahead only      ", False, True), ("batched D + ahead ", True, True)):
    s_, dl, gl = run(b_, a_)
    print("%s: d %.2e / %.2e   g %.2e / %.2e   d_loss %.7f  g_loss %.6f" % (name, *rel(s_["d"], base["d"]), *rel(s_["g"], base["g"]), dl, gl))
