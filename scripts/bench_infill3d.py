"""Throughput of the partial-convolution GAN iteration (row (f)-4) on synthetic volumes: the GPS-RO grid 45x19x37 the
reference's infill3d_gan_module.py ran on (layer sizes 6/6, one noise channel, batch 4 as in infill3d_gan_1.yaml) and a
64x96x96 volume.  Prints samples/s and the per-entry-point time table of one profiled iteration.  Not the headline
benchmark (bench.py)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import bias_gan_amd  # noqa
from bias_gan_amd import _lib as L
from bias_gan_amd.architecture.gpsro import infill3d_gan as ig
from bias_gan_amd.gpsro_train.train_infill3d_gan import InfillGANTrainer
from bias_gan_amd.utils import losses, parsing_helpers as ph

dev = torch.device("cuda", 0)
W = {"valid": 1.0, "hole": 0.5, "tv": 0.1, "adv": 0.5}
for (n, d, h, w) in ((4, 45, 19, 37), (16, 45, 19, 37), (4, 64, 96, 96), (4, 45, 19, 37)):
    with contextlib.redirect_stdout(io.StringIO()):
        net = ig.GAN(input_channels=2, output_channels=1, gen_layer_size=6, disc_layer_size=6)
        G, D = net.generator.set_compute_dtype(torch.bfloat16).to(dev), net.discriminator.set_compute_dtype(torch.bfloat16).to(dev)
    G.train(), D.train()
    tr = InfillGANTrainer(G, D, ph.get_optimizer(G.parameters(), "AdamW", 1e-4, 1e-8, 0.01),
                          ph.get_optimizer(D.parameters(), "AdamW", 1e-4, 1e-8, 0.01), losses.GANLoss("ModifiedMinMax", n, dev),
                          ig.InpaintingLoss("l2"), W, 0, 0.0, 1.0)
    gt = torch.randn(n, 1, d, h, w, device=dev)
    mask = (torch.rand(n, 1, d, h, w, device=dev) > 0.3).float()
    x, noise = gt * mask, torch.randn(n, 1, d, h, w, device=dev)
    for _ in range(4):      # two eager steps, the capture, one replay
        tr.step(x, gt, mask, noise)
    torch.cuda.synchronize()
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
        dl, gl = tr.step(x, gt, mask, noise)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"volume {d}x{h}x{w}, batch {n}: {1e3 * dt:.1f} ms/step, {n / dt:.1f} samples/s  (d_loss {float(dl):.3f}, g_loss {float(gl):.3f})", flush=True)
    L.PROFILE = []
    tr.step(x, gt, mask, noise)
    torch.cuda.synchronize()
    recs, L.PROFILE = L.PROFILE, None
    fam = {}
    for name, flops, e0, e1, nb in recs:
        f = fam.setdefault(name, [0, 0.0, 0.0])
        f[0] += 1; f[1] += e0.elapsed_time(e1); f[2] += flops
    tot = sum(v[1] for v in fam.values())
    print(f"   {len(recs)} launches, {tot:.1f} ms of kernel time")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {k:28s} x{v[0]:4d} {v[1]:8.2f} ms" + (f"  {v[2] / v[1] * 1e-9:7.1f} TF/s" if v[2] else ""))
    del G, D, tr, net
    torch.cuda.empty_cache()
