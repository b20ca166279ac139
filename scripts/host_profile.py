"""cProfile of the host side of GANTrainer.step() at the bench configuration (where does the enqueue time go)."""
import cProfile, os, pstats, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import bias_gan_amd  # noqa
from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg
from bias_gan_amd.gpsro_train.train_gan import GANTrainer
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
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(x, y)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
