"""cProfile of the host side of the 3-D GAN step at the GPS-RO grid (launch-bound: where the enqueue time goes)."""
import cProfile, os, pstats, sys, io, contextlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import bias_gan_amd  # noqa
from bias_gan_amd.architecture.gpsro import deeplab3d_gan as g3
from bias_gan_amd.gpsro_train.train_gan3d import GANTrainer3d
from bias_gan_amd.utils import losses, parsing_helpers as ph

dev = torch.device("cuda", 0)
n, d, h, w = 16, 45, 19, 37
with contextlib.redirect_stdout(io.StringIO()):
    G = g3.Generator(1, 1, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm3d, compute_dtype=torch.bfloat16).to(dev)
    D = g3.Discriminator(1, normalizer=nn.BatchNorm3d, compute_dtype=torch.bfloat16).to(dev)
G.train(), D.train()
tr = GANTrainer3d(G, D, ph.get_optimizer(G.parameters(), "Adam", 1e-4, 1e-8, 1e-5), ph.get_optimizer(D.parameters(), "Adam", 1e-4, 1e-8, 1e-5),
                  losses.GANLoss("ModifiedMinMax", n, dev), losses.L1Loss())
x = torch.randn(n, 1, d, h, w, device=dev); y = x + 0.1 * torch.randn_like(x)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    tr.step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 5:.1f} ms/step, with device {1e3 * (t2 - t0) / 5:.1f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(x, y)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:7000])
