"""Throughput of the 3-D GAN step (row (f)-3) on synthetic volumes: GPS-RO sized 45x19x37 (the grid the reference's
train_gan3d.py was run on, one channel) and a larger 64x96x96 volume.  Prints samples/s and the per-entry-point
time table of one profiled step.  Not the headline benchmark (bench.py)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L
from bias_gan_amd.architecture.gpsro import deeplab3d_gan as g3
from bias_gan_amd.gpsro_train.train_gan3d import GANTrainer3d
from bias_gan_amd.utils import losses, parsing_helpers as ph

dev = torch.device("cuda", 0)
for (n, d, h, w) in ((16, 45, 19, 37), (4, 64, 96, 96)):
    with contextlib.redirect_stdout(io.StringIO()):
        G = g3.Generator(1, 1, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm3d, compute_dtype=torch.bfloat16).to(dev)
        D = g3.Discriminator(1, normalizer=nn.BatchNorm3d, compute_dtype=torch.bfloat16).to(dev)
    G.train(), D.train()
    tr = GANTrainer3d(G, D, ph.get_optimizer(G.parameters(), "Adam", 1e-4, 1e-8, 1e-5),
                      ph.get_optimizer(D.parameters(), "Adam", 1e-4, 1e-8, 1e-5), losses.GANLoss("ModifiedMinMax", n, dev),
                      losses.L1Loss())
    x = torch.randn(n, 1, d, h, w, device=dev); y = x + 0.1 * torch.randn_like(x)
    for _ in range(4):        # two eager steps, the capture, one replay
        tr.step(x, y)
    torch.cuda.synchronize()
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
        dl, gl = tr.step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"[step graph {os.environ.get('BGAMD_STEP_GRAPH', 'auto')}: {len(getattr(tr, '_graphs', {}))} captured] volume {d}x{h}x{w}, batch {n}: {1e3 * dt:.1f} ms/step, {n / dt:.1f} samples/s  (d_loss {float(dl):.3f}, g_loss {float(gl):.3f})", flush=True)
    L.PROFILE = []
    tr.step(x, y)
    torch.cuda.synchronize()
    recs, L.PROFILE = L.PROFILE, None
    fam = {}
    for name, flops, e0, e1, nb in recs:
        f = fam.setdefault(name, [0, 0.0, 0.0])
        f[0] += 1; f[1] += e0.elapsed_time(e1); f[2] += flops
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"   {k:28s} x{v[0]:4d} {v[1]:8.2f} ms" + (f"  {v[2] / v[1] * 1e-9:7.1f} TF/s" if v[2] else ""))
    del G, D, tr
    torch.cuda.empty_cache()
