"""One small G+D training iteration on the GPU, checked against the CPU oracle.
Called by __graft_entry__.smoke().  Lives under tests/ (not in the package): it imports
the oracle, which only tests, smoke() and bench.py's cpu_baseline leg may do."""
import os
import sys

import torch
import torch.nn as nn


def run(device="cuda:0", c=4, h=64, w=64, n=2):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bias_gan_amd  # noqa: F401  (the import shim of the hyphenated package directory)
    from oracle import gan_oracle as orc
    from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg
    from bias_gan_amd.gpsro_train.train_gan import GANTrainer
    from bias_gan_amd.utils import losses
    from bias_gan_amd.utils import parsing_helpers as ph

    results = {}
    # fp32 path: the oracle's own tolerance.  bf16 path: storage rounding through the randomly filled 140-layer
    # nets reaches 10-20 % at the outputs (tests/test_parity_gpu.py header), and which side of a bf16 rounding
    # boundary a value lands on changes with the kernels' summation order, so the loss is only pinned loosely.
    for dtype, tol_d, tol_g in ((torch.float32, 1e-3, 2e-2), (torch.bfloat16, 2e-1, 3e-1)):
        gspec = orc.generator_spec(c, c, 0, "batch")
        dspec = orc.discriminator_spec(c, h, w, "batch")
        sg, sd = orc.fill_state(gspec, 21), orc.fill_state(dspec, 22)
        G = dxg.Generator(c, c, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d, compute_dtype=dtype)
        D = dxg.Discriminator(c, normalizer=nn.BatchNorm2d, input_size=(h, w), compute_dtype=dtype)
        G.load_state_dict(sg), D.load_state_dict(sd)
        G.to(device).train(), D.to(device).train()
        crit = losses.GANLoss("ModifiedMinMax", n, torch.device(device))
        tr = GANTrainer(G, D, ph.get_optimizer(G.parameters(), "Adam", 1e-4, 1e-8, 1e-5),
                        ph.get_optimizer(D.parameters(), "Adam", 1e-4, 1e-8, 1e-5), crit, losses.L1Loss())
        x, y = orc.synthetic_fields(n, c, h, w, 1000)
        torch.manual_seed(333)
        labels = crit.draw_labels()
        d_loss, g_loss = tr.step(x.to(device), y.to(device), labels=labels)
        torch.cuda.synchronize()
        st = orc.GANStep(orc.fill_state(gspec, 21), orc.fill_state(dspec, 22), orc.trainable_keys(gspec),
                         orc.trainable_keys(dspec), "batch", "ModifiedMinMax")
        d_ref, g_ref = st.step(x, y, labels=labels)
        d_loss, g_loss = float(d_loss), float(g_loss)
        results[str(dtype)] = (d_loss, d_ref, g_loss, g_ref)
        assert abs(d_loss - d_ref) <= tol_d * abs(d_ref), f"{dtype}: d_loss {d_loss} vs oracle {d_ref}"
        assert abs(g_loss - g_ref) <= tol_g * abs(g_ref), f"{dtype}: g_loss {g_loss} vs oracle {g_ref}"
    print("smoke ok:", results)
    return results
