"""How far ahead of the GPU is the host at the end of each half-step?  (host-bound phases show lag ~ 0)"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import bias_gan_amd  # noqa
from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg
from bias_gan_amd.gpsro_train.train_gan import GANTrainer
from bias_gan_amd.runtime import StatsPool
from bias_gan_amd.utils import losses, parsing_helpers as ph

dev = torch.device("cuda", 0)
c, h, w, n = 16, 1152, 768, 8
with contextlib.redirect_stdout(io.StringIO()):
    G = dxg.Generator(c, c, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d, compute_dtype=torch.bfloat16).to(dev)
    D = dxg.Discriminator(c, normalizer=nn.BatchNorm2d, input_size=(h, w), compute_dtype=torch.bfloat16).to(dev)
G.train(), D.train()
tr = GANTrainer(G, D, ph.get_optimizer(G.parameters(), "Adam", 1e-4, 1e-8, 1e-5), ph.get_optimizer(D.parameters(), "Adam", 1e-4, 1e-8, 1e-5),
                losses.GANLoss("ModifiedMinMax", n, dev), losses.L1Loss())
x = torch.randn(n, c, h, w, device=dev); y = x + 0.1 * torch.randn_like(x)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
rows = []
for it in range(4):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0 = time.perf_counter(); e0.record()
    StatsPool.reset_all()
    tr._want_g_ahead = True
    tr.d_step(x, y)
    tr._want_g_ahead = False
    td = time.perf_counter(); e1.record()
    tr.g_step(x, y)
    tr._finish_d(); tr.step_count += 1
    tg = time.perf_counter(); e2.record()
    torch.cuda.synchronize()
    rows.append((1e3 * (td - t0), e0.elapsed_time(e1), 1e3 * (tg - t0), e0.elapsed_time(e2)))
for r in rows:
    print("host d_step done %.1f ms | gpu d_step done %.1f ms || host step done %.1f ms | gpu step done %.1f ms" % r)
