"""Which Python lines issue aten::copy_ / clone / fill during one training step (they are graph nodes and launches too)."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench  # noqa: F401  (reuses its trainer construction)

def main():
    import argparse
    from torch.utils._python_dispatch import TorchDispatchMode
    hits = collections.Counter()

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if any(k in name for k in ("copy_", "clone", "fill_", "zero_", "_to_copy", "cat", "add", "mul", "sum", "mean", "div", "sub", "neg")):
                fr = [f for f in traceback.extract_stack()[:-1] if "bias-gan_amd" in f.filename or "bias_gan_amd" in f.filename or f.filename.endswith("bench.py")]
                where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
                hits[(name, where)] += 1
            return func(*args, **(kwargs or {}))

    import torch.nn as nn
    from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg
    from bias_gan_amd.gpsro_train.train_gan import GANTrainer
    from bias_gan_amd.utils import losses
    from bias_gan_amd.utils import parsing_helpers as ph
    os.environ["BGAMD_STEP_GRAPH"] = "0"
    c, n, device, dtype, mode = 16, 8, torch.device("cuda", 0), torch.bfloat16, "ModifiedMinMax"
    G = dxg.Generator(c, c, "Interpolate", "Uniform", 0, os=16, pretrained=False, normalizer=nn.BatchNorm2d, compute_dtype=dtype).to(device)
    D = dxg.Discriminator(n_input=c, os=16, pretrained=False, normalizer=nn.BatchNorm2d, input_size=(64, 64), compute_dtype=dtype).to(device)
    G.train(), D.train()
    trainer = GANTrainer(G, D, ph.get_optimizer(G.parameters(), "Adam", 1e-4, 1e-8, 1e-5),
                         ph.get_optimizer(D.parameters(), "Adam", 1e-4, 1e-8, 1e-5), losses.GANLoss(mode, n, device),
                         losses.L1Loss(), loss_type_gan=mode, loss_weight_gp=10.0)
    batches = [bench.synthetic_batch(n, c, 64, 64, 1, device)]
    for _ in range(3):
        trainer.step(*batches[0])
    torch.cuda.synchronize()
    with Spy():
        trainer.step(*batches[0])
    torch.cuda.synchronize()
    for (name, where), n in hits.most_common(12):
        print(f"{n:5d}  {name:40s} {where}")
    # the backward pass runs in autograd's own threads (dispatch modes are thread-local): profile those with stacks
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        trainer.step(*batches[0])
        torch.cuda.synchronize()
    agg = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::zeros", "aten::fill_", "aten::zero_"):
            st = [f for f in (ev.stack or []) if "bias" in f or "train_gan" in f]
            agg[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"), str(ev.input_shapes)[:60])] += 1
    for (name, where, shp), n in agg.most_common(30):
        print(f"{n:5d}  {name:18s} {where[-90:]}  {shp}")


if __name__ == "__main__":
    main()
