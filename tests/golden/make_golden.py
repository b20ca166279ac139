#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference).  It imports the
reference's own PyTorch modules (architecture/gpsro/deeplab_gan.py, deeplab.py,
utils/losses.py) on CPU, feeds them deterministic states and seeded synthetic
fields, and stores inputs' seeds + expected outputs as small .npz files.  The
reference cannot travel to the GPU box; these vectors can.

The only shim is an empty module object registered as ``conv2d_local`` (an
absent third-party extension that deeplab.py imports at :7 and only dead code
uses).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src/deepCam")
sys.modules.setdefault("conv2d_local", types.ModuleType("conv2d_local"))
# architecture/gpsro/infill3d.py:8 and infill3d_gan.py:8 import `models` from torchvision (absent here) and never
# use it: an empty module object in its place, as for conv2d_local
if "torchvision" not in sys.modules:
    _tv = types.ModuleType("torchvision")
    _tv.models = types.ModuleType("torchvision.models")
    sys.modules["torchvision"], sys.modules["torchvision.models"] = _tv, _tv.models
sys.dont_write_bytecode = True

from architecture.gpsro import deeplab as ref_dl  # noqa: E402
from architecture.gpsro import deeplab_gan as ref_gan  # noqa: E402
from utils import losses as ref_losses  # noqa: E402
from architecture.gpsro import deeplab3d_gan as ref_gan3d  # noqa: E402

from oracle import gan_oracle as orc  # noqa: E402  (only for specs / deterministic fills / fields)
from oracle import gan3d_oracle as orc3  # noqa: E402
from oracle import infill3d_oracle as orci  # noqa: E402
sys.path.insert(0, "/root/reference/src/deepCam/architecture/gpsro")
from architecture.gpsro import infill3d_gan as ref_infill  # noqa: E402
from architecture.gpsro import infill as ref_infill2d  # noqa: E402

torch.set_num_threads(8)


def build_ref_generator(c, norm_cls, seed, upsampler="Interpolate"):
    g = ref_gan.Generator(c, c, upsampler, "Uniform", 0, os=16, pretrained=False, normalizer=norm_cls)
    kind = "batch" if norm_cls is nn.BatchNorm2d else "instance"
    spec = orc.generator_spec(c, c, 0, kind, upsampler=upsampler)
    sd = g.state_dict()
    assert [k for k, _, _ in spec] == list(sd.keys()), "generator key order differs from the reference"
    for k, shape, _ in spec:
        assert tuple(sd[k].shape) == tuple(shape), (k, sd[k].shape, shape)
    g.load_state_dict(orc.fill_state(spec, seed))
    return g, spec


def build_ref_discriminator(c, h, w, norm_cls, seed):
    d = ref_gan.Discriminator(n_input=c, os=16, pretrained=False, normalizer=norm_cls)
    h16, w16 = orc.out_hw16(h, w)
    d.linear = nn.Linear(2048 * h16 * w16, 1)  # the reference hard-codes 12288 (19x37 grid)
    kind = "batch" if norm_cls is nn.BatchNorm2d else "instance"
    spec = orc.discriminator_spec(c, h, w, kind)
    sd = d.state_dict()
    assert [k for k, _, _ in spec] == list(sd.keys()), "discriminator key order differs from the reference"
    for k, shape, _ in spec:
        assert tuple(sd[k].shape) == tuple(shape), (k, sd[k].shape, shape)
    d.load_state_dict(orc.fill_state(spec, seed))
    return d, spec


def checksums(named):
    out = {}
    for k, v in named:
        v = v.detach().double()
        out[k] = np.array([v.sum().item(), v.abs().sum().item(), (v * v).sum().item()])
    return out


GRAD_FULL_G = ["model.xception_features.bn1.weight", "model.xception_features.block1.skipbn.bias",
               "model.upsample.last_conv.6.bias", "model.upsample.last_conv.6.weight",
               "model.xception_features.block7.rep.4.conv1.weight", "model.bn2.weight",
               "model.aspp3.bn.weight", "model.global_avg_pool.2.bias"]
GRAD_FULL_D = ["xception_features.bn1.weight", "xception_features.block1.skipbn.bias", "linear.bias",
               "xception_features.block7.rep.4.conv1.weight", "xception_features.bn5.weight"]


def golden_generator(tag, c, h, w, n, seed, upsampler="Interpolate", grad_full=None, bufs=None):
    g, spec = build_ref_generator(c, nn.BatchNorm2d, seed, upsampler)
    g.train()
    x, y = orc.synthetic_fields(n, c, h, w, seed + 100)
    out = g(x)
    loss = (out - y).abs().mean()
    loss.backward()
    res = {"out": out.detach().numpy(), "loss": np.array(loss.item())}
    cs = checksums((k, p.grad) for k, p in g.named_parameters())
    res["grad_keys"] = np.array(list(cs.keys()))
    res["grad_cs"] = np.stack(list(cs.values()))
    named = dict(g.named_parameters())
    for k in (grad_full or GRAD_FULL_G):
        res["grad::" + k] = named[k].grad.numpy()
    sd = g.state_dict()
    for k in (bufs or ("model.xception_features.bn1.running_mean", "model.xception_features.bn1.running_var",
                       "model.global_avg_pool.2.running_var", "model.upsample.last_conv.4.running_mean")):
        res["buf::" + k] = sd[k].numpy()
    # eval-mode forward with the now-updated running stats
    g.eval()
    with torch.no_grad():
        res["out_eval"] = g(x).numpy()
    np.savez_compressed(os.path.join(HERE, f"generator_{tag}.npz"), meta=json.dumps(
        dict(c=c, h=h, w=w, n=n, seed=seed, field_seed=seed + 100, upsampler=upsampler)), **res)
    print("generator", tag, "loss", loss.item())


def golden_generator_crop(tag, c, h, w, n, seed, stride=8):
    """The generator at a realistic field size (SURVEY section 7 stage 0: 256 x 256 x 16, N = 4), stored as a strided crop
    of the output plus checksums of the full output and of every parameter gradient (the full tensors would be 4 MB +
    220 MB)."""
    g, spec = build_ref_generator(c, nn.BatchNorm2d, seed)
    g.train()
    x, y = orc.synthetic_fields(n, c, h, w, seed + 100)
    out = g(x)
    loss = (out - y).abs().mean()
    loss.backward()
    res = {"out_crop": out.detach()[:, :, ::stride, ::stride].contiguous().numpy(), "loss": np.array(loss.item()),
           "out_cs": checksums([("out", out)])["out"]}
    cs = checksums((k, p.grad) for k, p in g.named_parameters())
    res["grad_keys"] = np.array(list(cs.keys()))
    res["grad_cs"] = np.stack(list(cs.values()))
    named = dict(g.named_parameters())
    for k in GRAD_FULL_G:
        res["grad::" + k] = named[k].grad.numpy()
    sd = g.state_dict()
    for k in ("model.xception_features.bn1.running_mean", "model.xception_features.bn1.running_var",
              "model.global_avg_pool.2.running_var", "model.upsample.last_conv.4.running_mean"):
        res["buf::" + k] = sd[k].numpy()
    np.savez_compressed(os.path.join(HERE, f"generator_{tag}.npz"), meta=json.dumps(
        dict(c=c, h=h, w=w, n=n, seed=seed, field_seed=seed + 100, stride=stride, upsampler="Interpolate")), **res)
    print("generator (crop)", tag, "loss", loss.item())


def golden_generator_noise(tag, c, nd, h, w, n, seed, noise_type="Uniform", noise_seed=77):
    """Generator with noise_dimensions > 0 -- the reference's default (train_gan.py:460): the noise is drawn on the HOST
    RNG stream by torch.distributions...rsample inside forward (deeplab_gan.py:85-90), so it is a function of
    torch.manual_seed alone.  Stored: the draw itself (pins draw order / shape), output, gradient checksums."""
    g = ref_gan.Generator(c, c, "Interpolate", noise_type, nd, os=16, pretrained=False, normalizer=nn.BatchNorm2d)
    spec = orc.generator_spec(c, c, nd, "batch")
    sd = g.state_dict()
    assert [k for k, _, _ in spec] == list(sd.keys())
    g.load_state_dict(orc.fill_state(spec, seed))
    g.train()
    x, y = orc.synthetic_fields(n, c, h, w, seed + 100)
    torch.manual_seed(noise_seed)
    noise = g.dist.rsample((n, nd, h, w))
    torch.manual_seed(noise_seed)
    out = g(x)
    loss = (out - y).abs().mean()
    loss.backward()
    res = {"out": out.detach().numpy(), "loss": np.array(loss.item()), "noise": noise.numpy()}
    cs = checksums((k, p.grad) for k, p in g.named_parameters())
    res["grad_keys"] = np.array(list(cs.keys()))
    res["grad_cs"] = np.stack(list(cs.values()))
    res["grad::model.xception_features.conv1.weight"] = dict(g.named_parameters())["model.xception_features.conv1.weight"].grad.numpy()
    np.savez_compressed(os.path.join(HERE, f"generator_{tag}.npz"), meta=json.dumps(
        dict(c=c, nd=nd, h=h, w=w, n=n, seed=seed, field_seed=seed + 100, noise_type=noise_type, noise_seed=noise_seed)), **res)
    print("generator", tag, "loss", loss.item())


def golden_checkpoint_structure(c=4, h=64, w=64, n=2, seed=21):
    """What the reference puts into a .cpt (train_gan.py:401-411): one real loop iteration of the reference modules
    under torch.optim.Adam, then the STRUCTURE of the checkpoint dictionary -- keys in order, shapes, dtypes, the
    optimiser's param_groups, per-parameter step counts -- plus three checksums per tensor.  (The tensors themselves
    are 1.1 GB; the structure and the sums are what a loader has to agree with.)"""
    g, gspec = build_ref_generator(c, nn.BatchNorm2d, seed)
    d, dspec = build_ref_discriminator(c, h, w, nn.BatchNorm2d, seed + 1)
    g.train(), d.train()
    g_opt = torch.optim.Adam(g.parameters(), lr=1e-4, eps=1e-8, weight_decay=1e-5)
    d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, eps=1e-8, weight_decay=1e-5)
    g_opt.param_groups[0]["initial_lr"] = 1e-4          # utils/parsing_helpers.py:19
    d_opt.param_groups[0]["initial_lr"] = 1e-4
    crit = ref_losses.GANLoss("ModifiedMinMax", n, torch.device("cpu"))
    x, y = orc.synthetic_fields(n, c, h, w, 1000)
    torch.manual_seed(seed)
    fake = g(x)                                           # train_gan.py:250-271
    lr_, _ = d(y)
    lf_, _ = d(fake)
    d_loss = crit.d_loss(lr_, lf_)
    d_opt.zero_grad()
    d_loss.backward()
    d_opt.step()
    fake = g(x)                                           # train_gan.py:273-298
    lf_, _ = d(fake)
    g_loss = crit.g_loss(lf_) + (fake - y).abs().mean()
    g_opt.zero_grad()
    g_loss.backward()
    g_opt.step()
    ck = {"step": 1, "epoch": 0, "generator": g.state_dict(), "discriminator": d.state_dict(),
          "g_opt": g_opt.state_dict(), "d_opt": d_opt.state_dict(), "amp": None}

    def tens(t):
        t64 = t.detach().double()
        return {"shape": list(t.shape), "dtype": str(t.dtype).replace("torch.", ""),
                "cs": [t64.sum().item(), t64.abs().sum().item(), (t64 * t64).sum().item()]}

    def opt(sd):
        return {"param_groups": [{k: (list(v) if isinstance(v, tuple) else v) for k, v in pg.items()} for pg in sd["param_groups"]],
                "state": {str(i): {"step": float(st["step"]), "step_is_tensor": torch.is_tensor(st["step"]),
                                   "exp_avg": tens(st["exp_avg"]), "exp_avg_sq": tens(st["exp_avg_sq"])}
                          for i, st in sd["state"].items()}}
    out = {"keys": list(ck.keys()), "step": ck["step"], "epoch": ck["epoch"],
           "generator": [[k, tens(v)] for k, v in ck["generator"].items()],
           "discriminator": [[k, tens(v)] for k, v in ck["discriminator"].items()],
           "g_opt": opt(ck["g_opt"]), "d_opt": opt(ck["d_opt"]),
           "meta": dict(c=c, h=h, w=w, n=n, seed=seed, field_seed=1000, lr=1e-4, eps=1e-8, weight_decay=1e-5,
                        d_loss=d_loss.item(), g_loss=g_loss.item())}
    with open(os.path.join(HERE, "checkpoint_structure.json"), "w") as f:
        json.dump(out, f)
    print("checkpoint structure: d_loss", d_loss.item(), "g_loss", g_loss.item())


def golden_discriminator(tag, c, h, w, n, seed, norm_cls):
    d, spec = build_ref_discriminator(c, h, w, norm_cls, seed)
    d.train()
    x, _ = orc.synthetic_fields(n, c, h, w, seed + 100)
    x.requires_grad_(True)
    logits, pred = d(x)
    tgt = torch.linspace(0.1, 0.9, n).reshape(n, 1)
    loss = nn.BCEWithLogitsLoss()(logits, tgt)
    loss.backward()
    res = {"logits": logits.detach().numpy(), "pred": pred.detach().numpy(), "loss": np.array(loss.item()),
           "dx": x.grad.numpy()}
    cs = checksums((k, p.grad) for k, p in d.named_parameters())
    res["grad_keys"] = np.array(list(cs.keys()))
    res["grad_cs"] = np.stack(list(cs.values()))
    named = dict(d.named_parameters())
    for k in GRAD_FULL_D:
        if k in named:
            res["grad::" + k] = named[k].grad.numpy()
    np.savez_compressed(os.path.join(HERE, f"discriminator_{tag}.npz"), meta=json.dumps(
        dict(c=c, h=h, w=w, n=n, seed=seed, field_seed=seed + 100,
             norm="batch" if norm_cls is nn.BatchNorm2d else "instance")), **res)
    print("discriminator", tag, "loss", loss.item())


def golden_modules(seed=5):
    """Small building blocks on their own: Block variants, SeparableConv2d_same,
    ASPP_module, InterpolationUpsampler."""
    res = {}
    rng = np.random.default_rng(seed)

    def fill(mod):
        for k, v in mod.state_dict().items():
            if v.dtype.is_floating_point:
                r = np.random.default_rng(sum(map(ord, k)) + seed)
                if k.endswith("running_var"):
                    v.copy_(torch.from_numpy(r.uniform(0.5, 1.5, v.shape).astype(np.float32)))
                elif k.endswith("weight") and v.dim() == 1:
                    v.copy_(torch.from_numpy(r.uniform(0.8, 1.2, v.shape).astype(np.float32)))
                else:
                    fan = max(1, int(np.prod(v.shape[1:]))) if v.dim() > 1 else 4
                    v.copy_(torch.from_numpy((r.standard_normal(v.shape) / np.sqrt(fan)).astype(np.float32)))

    cases = [
        ("blk_a", dict(inplanes=16, planes=24, reps=2, stride=2, dilation=1, start_with_relu=False, grow_first=True, is_last=False), (2, 16, 13, 18)),
        ("blk_b", dict(inplanes=24, planes=24, reps=3, stride=1, dilation=1, start_with_relu=True, grow_first=True, is_last=False), (2, 24, 9, 7)),
        ("blk_c", dict(inplanes=24, planes=40, reps=2, stride=1, dilation=2, start_with_relu=True, grow_first=False, is_last=True), (2, 24, 9, 7)),
        ("blk_d", dict(inplanes=16, planes=32, reps=2, stride=2, dilation=1, start_with_relu=True, grow_first=True, is_last=True), (2, 16, 12, 10)),
    ]
    for tag, kw, shape in cases:
        m = ref_dl.Block(normalizer=nn.BatchNorm2d, **kw)
        fill(m)
        m.train()
        x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
        x_in = x.clone().requires_grad_(True)
        xx = x_in * 1.0  # non-leaf so the in-place relu is legal
        y = m(xx)
        go = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        y.backward(go)
        res[tag + "::x"] = x.numpy()
        res[tag + "::x_after"] = xx.detach().numpy()  # shows the in-place activation of the input
        res[tag + "::y"] = y.detach().numpy()
        res[tag + "::go"] = go.numpy()
        res[tag + "::dx"] = x_in.grad.numpy()
        res[tag + "::cfg"] = np.array(json.dumps(kw))
        for k, v in m.state_dict().items():
            res[tag + "::sd::" + k] = v.numpy()
        for k, p in m.named_parameters():
            res[tag + "::grad::" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "modules.npz"), **res)
    print("modules ok")


def golden_losses():
    res = {}
    for seed in (0, 7, 123, 999):
        for mode in ("ModifiedMinMax", "Wasserstein"):
            n = 4
            crit = ref_losses.GANLoss(mode, n, torch.device("cpu"))
            g = torch.Generator().manual_seed(seed + 1)
            lr_ = torch.randn(n, 1, generator=g)
            lf_ = torch.randn(n, 1, generator=g)
            torch.manual_seed(seed)
            d = crit.d_loss(lr_, lf_)
            gl = crit.g_loss(lf_)
            # what the three draws were (same seed, same order)
            torch.manual_seed(seed)
            if mode == "ModifiedMinMax":
                lab_f = crit.dist_fake.rsample((n, 1))
                lab_r = crit.dist_real.rsample((n, 1))
                sw = crit.dist_swap.sample()
                res[f"{mode}_{seed}::label_fake"] = lab_f.numpy()
                res[f"{mode}_{seed}::label_real"] = lab_r.numpy()
                res[f"{mode}_{seed}::swap_u"] = sw.numpy()
            res[f"{mode}_{seed}::logits_real"] = lr_.numpy()
            res[f"{mode}_{seed}::logits_fake"] = lf_.numpy()
            res[f"{mode}_{seed}::d_loss"] = np.array(d.item())
            res[f"{mode}_{seed}::g_loss"] = np.array(gl.item())
    # a seed that triggers the 5 % label swap
    for seed in range(2000):
        torch.manual_seed(seed)
        torch.rand(4, 1); torch.rand(4, 1)
        if torch.rand(()) < 0.05:
            res["swap_seed"] = np.array(seed)
            crit = ref_losses.GANLoss("ModifiedMinMax", 4, torch.device("cpu"))
            g = torch.Generator().manual_seed(1)
            lr_ = torch.randn(4, 1, generator=g); lf_ = torch.randn(4, 1, generator=g)
            torch.manual_seed(seed)
            res["swap::d_loss"] = np.array(crit.d_loss(lr_, lf_).item())
            res["swap::logits_real"] = lr_.numpy(); res["swap::logits_fake"] = lf_.numpy()
            break
    # weighted L1
    rng = np.random.default_rng(3)
    p = torch.from_numpy(rng.standard_normal((2, 3, 5, 4)).astype(np.float32))
    t = torch.from_numpy(rng.standard_normal((2, 3, 5, 4)).astype(np.float32))
    wt = torch.from_numpy(rng.uniform(0, 1, (2, 3, 5, 4)).astype(np.float32))
    res["l1w::p"], res["l1w::t"], res["l1w::w"] = p.numpy(), t.numpy(), wt.numpy()
    res["l1w::plain"] = np.array(ref_losses.L1LossWeighted()(p, t, wt).item())
    res["l1w::normalized"] = np.array(ref_losses.L1LossWeighted(normalize=True)(p, t, wt).item())
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **res)
    print("losses ok")


def golden_gradient_penalty(c=4, h=32, w=32, n=2, seed=11):
    d, spec = build_ref_discriminator(c, h, w, nn.BatchNorm2d, seed)
    d.train()
    fake, real = orc.synthetic_fields(n, c, h, w, seed + 100)
    torch.manual_seed(seed)
    gp = ref_gan.gradient_penalty(d, fake, real)
    torch.manual_seed(seed)
    eta = torch.rand((n, 1, 1, 1))
    res = {"gp": np.array(gp.item()), "eta": eta.numpy(), "has_graph": np.array(gp.grad_fn is not None),
           "bn1_rm_after": d.state_dict()["xception_features.bn1.running_mean"].numpy()}
    np.savez_compressed(os.path.join(HERE, "gradient_penalty.npz"), meta=json.dumps(
        dict(c=c, h=h, w=w, n=n, seed=seed, field_seed=seed + 100)), **res)
    print("gp", gp.item(), "graph", gp.grad_fn)


def golden_trajectory(tag, mode, c=4, h=64, w=64, n=2, seed=21, steps=3, adam_eps=1e-8):
    """Three iterations of the loop body train_gan.py:244-298 driven on the
    reference modules, torch.optim.Adam (lr 1e-4, eps 1e-8, wd 1e-5: launcher
    values), GANLoss and nn.L1Loss, weights 1/1, GP weight 10."""
    g, gspec = build_ref_generator(c, nn.BatchNorm2d, seed)
    d, dspec = build_ref_discriminator(c, h, w, nn.BatchNorm2d, seed + 1)
    g.train(); d.train()
    g_opt = torch.optim.Adam(g.parameters(), lr=1e-4, eps=adam_eps, weight_decay=1e-5)
    d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, eps=adam_eps, weight_decay=1e-5)
    crit = ref_losses.GANLoss(mode, n, torch.device("cpu"))
    l1 = nn.L1Loss()
    torch.manual_seed(333)
    res = {"d_loss": [], "g_loss": [], "labels_fake": [], "labels_real": [], "swap": [], "eta": []}
    watch_g = ["model.xception_features.conv1.weight", "model.xception_features.bn1.weight",
               "model.upsample.last_conv.6.bias", "model.xception_features.block10.rep.1.pointwise.weight"]
    watch_d = ["xception_features.conv1.weight", "linear.weight", "linear.bias",
               "xception_features.block10.rep.1.pointwise.weight"]
    for s in range(steps):
        inputs, real = orc.synthetic_fields(n, c, h, w, 1000 + s)
        # record the host draws this step will make (same generator state)
        st = torch.get_rng_state()
        if mode == "ModifiedMinMax":
            lf, lr_, sw = orc.draw_d_labels(n)
            res["labels_fake"].append(lf.numpy()); res["labels_real"].append(lr_.numpy()); res["swap"].append(sw)
        else:
            res["eta"].append(torch.rand((n, 1, 1, 1)).numpy())
        torch.set_rng_state(st)
        # ---- D-step (train_gan.py:250-266)
        fake = g(inputs)
        logits_real, _ = d(real)
        logits_fake, _ = d(fake)
        d_loss = crit.d_loss(logits_real, logits_fake)
        if mode == "Wasserstein":
            d_loss = d_loss + 10.0 * ref_gan.gradient_penalty(d, fake, real)
        d_opt.zero_grad()
        d_loss.backward()
        d_opt.step()
        # ---- G-step (train_gan.py:273-293)
        fake = g(inputs)
        logits_fake, _ = d(fake)
        gan_loss = crit.g_loss(logits_fake)
        reg = l1(fake, real)
        g_loss = 1.0 * gan_loss + 1.0 * reg
        g_opt.zero_grad()
        g_loss.backward()
        g_opt.step()
        res["d_loss"].append(d_loss.item()); res["g_loss"].append(g_loss.item())
        gsd, dsd = g.state_dict(), d.state_dict()
        for k in watch_g:
            res.setdefault("G::" + k, []).append(checksums([(k, gsd[k])])[k])
        for k in watch_d:
            res.setdefault("D::" + k, []).append(checksums([(k, dsd[k])])[k])
        res.setdefault("G::bn1.running_mean", []).append(gsd["model.xception_features.bn1.running_mean"].numpy().copy())
        res.setdefault("G::bn1.running_var", []).append(gsd["model.xception_features.bn1.running_var"].numpy().copy())
        res.setdefault("D::bn1.running_mean", []).append(dsd["xception_features.bn1.running_mean"].numpy().copy())
        res.setdefault("G::bn1.nbt", []).append(int(gsd["model.xception_features.bn1.num_batches_tracked"]))
        res.setdefault("D::bn1.nbt", []).append(int(dsd["xception_features.bn1.num_batches_tracked"]))
        print(tag, "step", s, d_loss.item(), g_loss.item())
    out = {k: np.array(v) for k, v in res.items() if len(v)}
    np.savez_compressed(os.path.join(HERE, f"trajectory_{tag}.npz"), meta=json.dumps(
        dict(c=c, h=h, w=w, n=n, seed=seed, steps=steps, mode=mode, torch_seed=333, field_seed0=1000,
             adam_eps=adam_eps)), **out)


def golden_c1_plumbing(c=4, h=64, w=64, n=2, steps=3):
    """BASELINE.json configs[0]: 1-layer G (conv3x3 -> BN -> LeakyReLU, the stem
    pattern deeplab.py:178-180) + 1-layer D (same stem -> flatten -> Linear),
    assembled from torch.nn, driven through the reference GANLoss."""
    rng = np.random.default_rng(77)
    g = nn.Sequential(nn.Conv2d(c, c, 3, padding=1, bias=False), nn.BatchNorm2d(c), nn.LeakyReLU(0.2))
    dstem = nn.Sequential(nn.Conv2d(c, 8, 3, stride=2, padding=1, bias=False), nn.BatchNorm2d(8), nn.LeakyReLU(0.2))
    dlin = nn.Linear(8 * (h // 2) * (w // 2), 1)
    res = {}
    with torch.no_grad():
        for name, mod in (("g", g), ("dstem", dstem), ("dlin", dlin)):
            for k, v in mod.state_dict().items():
                if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
                    fan = max(1, int(np.prod(v.shape[1:]))) if v.dim() > 1 else 1
                    if v.dim() == 1 and k.endswith("weight"):
                        v.copy_(torch.from_numpy(rng.uniform(0.8, 1.2, v.shape).astype(np.float32)))
                    else:
                        v.copy_(torch.from_numpy((rng.standard_normal(v.shape) / np.sqrt(fan)).astype(np.float32)))
                res[f"init::{name}.{k}"] = v.numpy().copy()
    params_d = list(dstem.parameters()) + list(dlin.parameters())
    g_opt = torch.optim.Adam(g.parameters(), lr=1e-4, eps=1e-8, weight_decay=1e-5)
    d_opt = torch.optim.Adam(params_d, lr=1e-4, eps=1e-8, weight_decay=1e-5)
    crit = ref_losses.GANLoss("ModifiedMinMax", n, torch.device("cpu"))
    D = lambda x: dlin(dstem(x).reshape(x.shape[0], -1))
    torch.manual_seed(333)
    dl, gl = [], []
    for s in range(steps):
        inputs, real = orc.synthetic_fields(n, c, h, w, 2000 + s)
        fake = g(inputs)
        d_loss = crit.d_loss(D(real), D(fake))
        d_opt.zero_grad(); d_loss.backward(); d_opt.step()
        fake = g(inputs)
        g_loss = crit.g_loss(D(fake)) + nn.L1Loss()(fake, real)
        g_opt.zero_grad(); g_loss.backward(); g_opt.step()
        dl.append(d_loss.item()); gl.append(g_loss.item())
    res["d_loss"], res["g_loss"] = np.array(dl), np.array(gl)
    res["final::g.0.weight"] = g[0].weight.detach().numpy()
    res["final::dlin.weight_cs"] = checksums([("w", dlin.weight)])["w"]
    np.savez_compressed(os.path.join(HERE, "c1_plumbing.npz"), meta=json.dumps(
        dict(c=c, h=h, w=w, n=n, steps=steps, torch_seed=333, field_seed0=2000)), **res)
    print("c1", dl, gl)


def _load_checked(mod, spec, seed):
    sd = mod.state_dict()
    assert [k for k, _, _ in spec] == list(sd.keys()), "key order differs from the reference"
    for k, shape, _ in spec:
        assert tuple(sd[k].shape) == tuple(shape), (k, sd[k].shape, shape)
    mod.load_state_dict(orc.fill_state(spec, seed))


def golden_gan3d(c=1, d=16, h=24, w=24, n=2):
    """3-D DeepLab GAN: Generator (Interpolate upsampler, BatchNorm3d), Discriminator with
    BatchNorm3d and InstanceNorm3d (the Wasserstein critic, train_gan3d.py:140), per-sample gradient penalty."""
    res = {}
    g = ref_gan3d.Generator(c, c, "Interpolate", "Uniform", 0, os=16, pretrained=False, normalizer=nn.BatchNorm3d)
    gspec = orc3.generator3d_spec(c, c, 0, "batch")
    _load_checked(g, gspec, 31)
    g.train()
    x, y = orc3.synthetic_volumes(n, c, d, h, w, 131)
    out = g(x)
    loss = (out - y).abs().mean()
    loss.backward()
    res["g::out"], res["g::loss"] = out.detach().numpy(), np.array(loss.item())
    cs = checksums((k, p.grad) for k, p in g.named_parameters())
    res["g::grad_keys"], res["g::grad_cs"] = np.array(list(cs.keys())), np.stack(list(cs.values()))
    named = dict(g.named_parameters())
    for k in ("model.xception_features.conv1.weight", "model.xception_features.bn1.weight",
              "model.xception_features.block1.rep.0.conv1.weight", "model.upsample.last_conv.6.weight",
              "model.aspp3.bn.bias", "model.global_avg_pool.2.weight"):
        res["g::grad::" + k] = named[k].grad.numpy()
    sd = g.state_dict()
    for k in ("model.xception_features.bn1.running_mean", "model.xception_features.bn2.running_var",
              "model.upsample.last_conv.4.running_mean"):
        res["g::buf::" + k] = sd[k].numpy()
    g.eval()
    with torch.no_grad():
        res["g::out_eval"] = g(x).numpy()
    # the InstanceNorm3d critic gets a larger volume: on 16x24x24 its deepest maps hold 4 values per channel and
    # rstd = 1/sqrt(var + 1e-5) blows rounding noise up to 10 % of the logits (two fp32 evaluations disagree)
    d2, h2, w2 = 2 * d, 2 * h, 2 * w
    x_in, _ = orc3.synthetic_volumes(n, c, d2, h2, w2, 133)
    for tag, norm_cls, kind, seed in (("d_bn", nn.BatchNorm3d, "batch", 32), ("d_in", nn.InstanceNorm3d, "instance", 33)):
        dm = ref_gan3d.Discriminator(n_input=c, os=16, pretrained=False, normalizer=norm_cls)
        dspec = orc3.discriminator3d_spec(c, kind)
        _load_checked(dm, dspec, seed)
        dm.train()
        xd = (x if tag == "d_bn" else x_in).clone().requires_grad_(True)
        logits, pred = dm(xd)
        tgt = torch.linspace(0.1, 0.9, n).reshape(-1, 1)
        dl = nn.functional.binary_cross_entropy_with_logits(logits, tgt)
        dl.backward()
        res[f"{tag}::logits"], res[f"{tag}::pred"] = logits.detach().numpy(), pred.detach().numpy()
        res[f"{tag}::loss"], res[f"{tag}::dx"] = np.array(dl.item()), xd.grad.numpy()
        cs = checksums((k, p.grad) for k, p in dm.named_parameters())
        res[f"{tag}::grad_keys"], res[f"{tag}::grad_cs"] = np.array(list(cs.keys())), np.stack(list(cs.values()))
        res[f"{tag}::grad::linear.weight"] = dict(dm.named_parameters())["linear.weight"].grad.numpy()
        if tag == "d_in":
            # gradient penalty with the critic above: eta comes from the host RNG (deeplab3d_gan.py:105-106)
            fake, real = orc3.synthetic_volumes(n, c, d2, h2, w2, 132)
            torch.manual_seed(77)
            gp = ref_gan3d.gradient_penalty(dm, fake, real)
            torch.manual_seed(77)
            eta = torch.distributions.uniform.Uniform(0., 1.).rsample((n, 1, 1, 1, 1))
            res["gp::value"], res["gp::eta"] = np.array(gp.item()), eta.numpy()
            assert not gp.requires_grad
    np.savez_compressed(os.path.join(HERE, "gan3d_c1_16x24x24.npz"),
                        meta=json.dumps(dict(c=c, d=d, h=h, w=w, n=n, g_seed=31, d_bn_seed=32, d_in_seed=33,
                                             field_seed=131, gp_field_seed=132, in_dhw=[d2, h2, w2], in_field_seed=133)),
                        **res)
    with open(os.path.join(HERE, "state_dict_keys_3d.json"), "w") as f:
        json.dump({"generator3d_c1": [[k, list(v.shape)] for k, v in g.state_dict().items()],
                   "discriminator3d_c1_bn": [[k, s_] for k, s_ in
                                             [(k, list(sh)) for k, sh, _ in orc3.discriminator3d_spec(c, "batch")]]}, f)
    print("gan3d goldens written: g loss", float(res["g::loss"]), "gp", float(res["gp::value"]))


def golden_gan3d_deconv(c=2, d=21, h=19, w=21, n=2):
    """3-D generators with the Deconv (the default of train_gan3d.py:575) and Deconv1x upsamplers (deeplab3d.py:342-444,
    518-532) on a volume their output paddings fit (D = 4a-3, H = 4b-1, W = 4c-3)."""
    res, keys = {}, {}
    x, y = orc3.synthetic_volumes(n, c, d, h, w, 141)
    for tag, up, seed in (("deconv", "Deconv", 41), ("deconv1x", "Deconv1x", 42)):
        g = ref_gan3d.Generator(c, c, up, "Uniform", 0, os=16, pretrained=False, normalizer=nn.BatchNorm3d)
        gspec = orc3.generator3d_spec(c, c, 0, "batch", upsampler=up)
        _load_checked(g, gspec, seed)
        g.train()
        out = g(x)
        assert out.shape == y.shape, (out.shape, y.shape)
        loss = (out - y).abs().mean()
        loss.backward()
        res[f"{tag}::out"], res[f"{tag}::loss"] = out.detach().numpy(), np.array(loss.item())
        cs = checksums((k, p.grad) for k, p in g.named_parameters())
        res[f"{tag}::grad_keys"], res[f"{tag}::grad_cs"] = np.array(list(cs.keys())), np.stack(list(cs.values()))
        named = dict(g.named_parameters())
        watch = ["model.upsample.conv1.6.bias", "model.upsample.deconv2.1.weight", "model.xception_features.conv1.weight"]
        if up == "Deconv1x":
            watch += ["model.upsample_extension.conv1.0.weight", "model.upsample_extension.init_norm.0.bias",
                      "model.upsample_extension.conv2.3.weight"]
        else:
            watch += ["model.upsample.last_deconv.0.weight"]
        for k in watch:
            res[f"{tag}::grad::" + k] = named[k].grad.numpy()
        for k in ("model.upsample.deconv1.0.weight", "model.upsample.deconv3.0.weight"):   # every 8th channel pair
            res[f"{tag}::grad8::" + k] = named[k].grad[::8, ::8].contiguous().numpy()
        sd = g.state_dict()
        for k in ("model.upsample.deconv1.1.running_mean", "model.upsample.deconv3.1.running_var"):
            res[f"{tag}::buf::" + k] = sd[k].numpy()
        keys["generator3d_" + tag] = [[k, list(v.shape)] for k, v in sd.items()]
    np.savez_compressed(os.path.join(HERE, f"gan3d_deconv_c{c}_{d}x{h}x{w}.npz"),
                        meta=json.dumps(dict(c=c, d=d, h=h, w=w, n=n, seeds={"deconv": 41, "deconv1x": 42}, field_seed=141)),
                        **res)
    with open(os.path.join(HERE, "state_dict_keys_3d_deconv.json"), "w") as f:
        json.dump(keys, f)
    print("gan3d deconv goldens written:", {t: float(res[t + "::loss"]) for t in ("deconv", "deconv1x")})


def golden_infill3d(cin=2, cout=1, d=32, h=24, w=40, n=2, g_layers=4, d_layers=5):
    """Partial-convolution U-Net generator and discriminator in 3-D + the inpainting loss."""
    res = {}
    g = ref_infill.Generator(layer_size=g_layers, input_channels=cin, output_channels=cout, upsampling_mode="nearest",
                             normalizer=nn.BatchNorm3d)
    gspec = orci.unet3d_spec(cin, cout, g_layers)
    _load_checked(g, gspec, 51)
    g.train()
    x, gt, mask = orci.synthetic_infill(n, cin, d, h, w, 151)
    out, out_mask = g(x, mask)
    ld = ref_losses.InpaintingLoss(loss_type="smooth-l1")(x[:, :cout], out, gt[:, :cout], mask[:, :cout])
    loss = 6.0 * ld["hole"] + 1.0 * ld["valid"] + 0.1 * ld["tv"]
    loss.backward()
    res["g::out"], res["g::out_mask"] = out.detach().numpy(), out_mask.detach().numpy()
    for k_ in ("hole", "valid", "tv"):
        res["g::loss_" + k_] = np.array(ld[k_].item())
    cs = checksums((k, p.grad) for k, p in g.named_parameters())
    res["g::grad_keys"], res["g::grad_cs"] = np.array(list(cs.keys())), np.stack(list(cs.values()))
    named = dict(g.named_parameters())
    for k in ("enc_1.conv.weight", "enc_2.bn.weight", "dec_1.conv.weight", "dec_3.bn.bias", "last_conv.conv.weight",
              "last_conv.conv.bias"):
        res["g::grad::" + k] = named[k].grad.numpy()
    sd = g.state_dict()
    for k in ("enc_2.bn.running_mean", "dec_1.bn.running_var"):
        res["g::buf::" + k] = sd[k].numpy()
    # intermediate masks of the encoder (bit-exact index/mask arithmetic)
    with torch.no_grad():
        m_ = mask
        for i in range(1, g_layers + 1):
            _, m_ = getattr(g, f"enc_{i}").conv(torch.zeros_like(m_), m_)
            res[f"g::enc_mask_{i}"] = m_[:, :1].numpy()
    dm = ref_infill.Discriminator(layer_size=d_layers, input_channels=cout, normalizer=nn.BatchNorm3d)
    dspec = orci.disc3d_spec(cout, d_layers)
    _load_checked(dm, dspec, 52)
    dm.train()
    xd = gt[:, :cout].clone().requires_grad_(True)
    logits, pred = dm(xd, mask[:, :cout])
    tgt = torch.linspace(0.1, 0.9, n).reshape(-1, 1)
    dl = nn.functional.binary_cross_entropy_with_logits(logits, tgt)
    dl.backward()
    res["d::logits"], res["d::loss"], res["d::dx"] = logits.detach().numpy(), np.array(dl.item()), xd.grad.numpy()
    cs = checksums((k, p.grad) for k, p in dm.named_parameters() if p.grad is not None)
    res["d::grad_keys"], res["d::grad_cs"] = np.array(list(cs.keys())), np.stack(list(cs.values()))
    np.savez_compressed(os.path.join(HERE, f"infill3d_c{cin}_{d}x{h}x{w}.npz"),
                        meta=json.dumps(dict(cin=cin, cout=cout, d=d, h=h, w=w, n=n, g_layers=g_layers, d_layers=d_layers,
                                             g_seed=51, d_seed=52, field_seed=151)), **res)
    print("infill3d goldens written:", {k_: float(res["g::loss_" + k_]) for k_ in ("hole", "valid", "tv")})


def golden_infill3d_options(cin=2, cout=1, d=18, h=10, w=14, n=2, g_layers=4, p_drop=0.25):
    """PConvUNet3d's two options off the GAN path (infill3d.py:139-142): upsampling_mode='trilinear' (odd sizes: source
    maps with scales 2/3, 3/5, 5/9, 4/7 besides 1/2) and dropout_p > 0 (PCDropout3d in training mode).  The draws of
    the reference's nn.Dropout3d are RECORDED by a forward hook -- keep[n][c] = the mask map survived -- so that the
    oracle and the HIP path can be handed the same draw; maps whose mask is all zero have no observable draw (and none
    matters: mask * keep = 0 either way) and are recorded as kept."""
    res = {}
    x, gt, mask = orci.synthetic_infill(n, cin, d, h, w, 161)
    for tag, kw in (("tri", dict(upsampling_mode="trilinear")), ("drop", dict(upsampling_mode="nearest", dropout_p=p_drop)),
                    ("tridrop", dict(upsampling_mode="trilinear", dropout_p=p_drop))):
        g = ref_infill.Generator(layer_size=g_layers, input_channels=cin, output_channels=cout, normalizer=nn.BatchNorm3d, **kw)
        _load_checked(g, orci.unet3d_spec(cin, cout, g_layers), 53)
        g.train()
        keeps = []
        if g.dropout is not None:
            def hook(_m, inp, out):
                alive = inp[0].flatten(2).amax(dim=2) > 0
                keeps.append(torch.where(alive, (out.flatten(2).amax(dim=2) > 0).float(), torch.ones_like(alive, dtype=torch.float32)))
            g.dropout.dropout.register_forward_hook(hook)
        torch.manual_seed(1234)
        out, out_mask = g(x, mask)
        ld = ref_losses.InpaintingLoss(loss_type="smooth-l1")(x[:, :cout], out, gt[:, :cout], mask[:, :cout])
        (6.0 * ld["hole"] + 1.0 * ld["valid"] + 0.1 * ld["tv"]).backward()
        res[tag + "::out"], res[tag + "::out_mask"] = out.detach().numpy(), out_mask.detach().numpy()
        for i, k_ in enumerate(keeps):
            res[f"{tag}::keep_{i}"] = k_.numpy()
        res[tag + "::n_keeps"] = np.array(len(keeps))
        cs = checksums((k, p_.grad) for k, p_ in g.named_parameters())
        res[tag + "::grad_keys"], res[tag + "::grad_cs"] = np.array(list(cs.keys())), np.stack(list(cs.values()))
        named = dict(g.named_parameters())
        for k in ("enc_1.conv.weight", "enc_2.bn.weight", "dec_1.conv.weight", "dec_2.bn.bias", "last_conv.conv.bias"):
            res[f"{tag}::grad::" + k] = named[k].grad.numpy()
    np.savez_compressed(os.path.join(HERE, f"infill3d_options_c{cin}_{d}x{h}x{w}.npz"),
                        meta=json.dumps(dict(cin=cin, cout=cout, d=d, h=h, w=w, n=n, g_layers=g_layers, p_drop=p_drop, g_seed=53,
                                             field_seed=161)), **res)
    print("infill3d option goldens written:", {t: int(res[t + "::n_keeps"]) for t in ("tri", "drop", "tridrop")})


def golden_infill_trajectory(d=32, h=24, w=40, n=2, g_layers=4, d_layers=5, steps=7, warmup=1, acc_min=0.55, acc_max=0.8,
                             lr=1e-3, wd=0.01, adam_eps=1e-4):
    """Seven iterations of the loop body infill3d_gan_module.py:272-375 driven on the reference's Generator,
    Discriminator, InpaintingLoss, GANLoss and metrics.accuracy with torch.optim.AdamW (the optimizer / weights /
    thresholds of gpsro_configs/infill3d_gan_1.yaml, lr 1e-3, tv weight 0.1 so that every term acts)."""
    from utils import metrics as ref_metrics
    weights = {"valid": 1.0, "hole": 0.5, "tv": 0.1, "adv": 0.5}
    nd = 1
    g = ref_infill.Generator(layer_size=g_layers, input_channels=1 + nd, output_channels=1, upsampling_mode="nearest",
                             normalizer=nn.BatchNorm3d)
    dm = ref_infill.Discriminator(layer_size=d_layers, input_channels=1 + nd, normalizer=nn.BatchNorm3d)
    gspec, dspec = orci.unet3d_spec(1 + nd, 1, g_layers), orci.disc3d_spec(1 + nd, d_layers)
    _load_checked(g, gspec, 61)
    _load_checked(dm, dspec, 62)
    g.train(); dm.train()
    g_opt = torch.optim.AdamW(g.parameters(), lr=lr, eps=adam_eps, weight_decay=wd)
    d_opt = torch.optim.AdamW(dm.parameters(), lr=lr, eps=adam_eps, weight_decay=wd)
    crit = ref_losses.GANLoss("ModifiedMinMax", n, torch.device("cpu"))
    rec = ref_losses.InpaintingLoss(loss_type="l2")
    torch.manual_seed(444)
    res = {k: [] for k in ("d_loss", "g_loss", "d_acc", "train_g", "train_d", "labels_fake", "labels_real", "swap",
                           "hole", "valid", "tv", "adv")}
    watch_g = ["enc_1.conv.weight", "enc_3.bn.weight", "dec_2.conv.weight", "last_conv.conv.bias"]
    watch_d = ["enc_1.conv.weight", "enc_4.bn.bias", "enc_5.conv.weight", "linear.weight"]
    d_acc_avg = 0.5
    for s in range(steps):
        x, gt, mask = orci.synthetic_infill(n, 1, d, h, w, 2000 + s)
        noise = torch.randn((n, nd, d, h, w), generator=torch.Generator().manual_seed(3000 + s))
        inputs, masks = torch.cat((x, noise), dim=1), torch.cat((mask, torch.ones_like(noise)), dim=1)
        st = torch.get_rng_state()
        lf, lr_, sw = orc.draw_d_labels(n)
        res["labels_fake"].append(lf.numpy()); res["labels_real"].append(lr_.numpy()); res["swap"].append(sw)
        torch.set_rng_state(st)
        if s < warmup:
            tg, td = True, False
        elif d_acc_avg > acc_max:
            tg, td = True, False
        elif d_acc_avg < acc_min:
            tg, td = False, True
        else:
            tg, td = True, True
        fake, _ = g(inputs, masks)
        logits_real, pred_real = dm(gt, masks)
        logits_fake, pred_fake = dm(fake, masks)
        d_loss = crit.d_loss(logits_real, logits_fake) * weights["adv"]
        d_acc_avg = float(0.5 * (ref_metrics.accuracy(pred_real, crit.label_real) + ref_metrics.accuracy(pred_fake, crit.label_fake)))
        if td:
            d_opt.zero_grad()
            d_loss.backward()
            d_opt.step()
        fake, _ = g(inputs, masks)
        logits_fake, _ = dm(fake, masks)
        terms = rec(x, fake, gt, mask)
        if s >= warmup:
            terms["adv"] = crit.g_loss(logits_fake)
        g_loss = 0.
        for k in terms:
            g_loss = g_loss + terms[k] * weights[k]
        if tg:
            g_opt.zero_grad()
            g_loss.backward()
            g_opt.step()
        res["d_loss"].append(d_loss.item()); res["g_loss"].append(g_loss.item()); res["d_acc"].append(d_acc_avg)
        res["train_g"].append(tg); res["train_d"].append(td)
        for k in ("hole", "valid", "tv", "adv"):
            res[k].append(terms[k].item() if k in terms else np.nan)
        gsd, dsd = g.state_dict(), dm.state_dict()
        for k in watch_g:
            res.setdefault("G::" + k, []).append(checksums([(k, gsd[k])])[k])
        for k in watch_d:
            res.setdefault("D::" + k, []).append(checksums([(k, dsd[k])])[k])
        res.setdefault("G::enc_2.bn.running_mean", []).append(gsd["enc_2.bn.running_mean"].numpy().copy())
        res.setdefault("D::enc_2.bn.running_var", []).append(dsd["enc_2.bn.running_var"].numpy().copy())
        res.setdefault("G::nbt", []).append(int(gsd["enc_2.bn.num_batches_tracked"]))
        res.setdefault("D::nbt", []).append(int(dsd["enc_2.bn.num_batches_tracked"]))
        print("infill trajectory step", s, (tg, td), d_acc_avg, d_loss.item(), g_loss.item())
    out = {k: np.array(v) for k, v in res.items() if len(v)}
    np.savez_compressed(os.path.join(HERE, "trajectory_infill3d.npz"), meta=json.dumps(
        dict(d=d, h=h, w=w, n=n, g_layers=g_layers, d_layers=d_layers, steps=steps, warmup=warmup, acc_min=acc_min,
             acc_max=acc_max, lr=lr, wd=wd, adam_eps=adam_eps, weights=weights, noise_dims=nd, g_seed=61, d_seed=62, torch_seed=444,
             field_seed0=2000, noise_seed0=3000, loss_type="l2")), **out)


def golden_infill2d(c=2, h=40, w=56, n=2, layers=4, mode="nearest"):
    """2-D partial-convolution U-Net (infill.py, partialconv2d.py; SURVEY 8(f)-4 "2-D shapes") + the inpainting loss on
    4-D tensors (its total-variation term then shifts along W and H).  mode: upsampling_mode of the features
    ('bilinear': a second file, on sizes whose halvings are odd -- 22 x 26 -> 11 x 13 -> 6 x 7 -> 3 x 4 -> 2 x 2)."""
    res = {}
    g = ref_infill2d.PConvUNet(layer_size=layers, input_channels=c, output_channels=c, upsampling_mode=mode,
                               normalizer=nn.BatchNorm2d)
    spec = orci.unet2d_spec(c, c, layers)
    _load_checked(g, spec, 71)
    g.train()
    gen = torch.Generator().manual_seed(171)
    gt = torch.randn((n, c, h, w), generator=gen)
    mask = (torch.rand((n, c, h, w), generator=gen) > 0.3).float()
    x = gt * mask
    out, out_mask = g(x, mask)
    ld = ref_losses.InpaintingLoss(loss_type="l1")(x, out, gt, mask)
    (6.0 * ld["hole"] + 1.0 * ld["valid"] + 0.1 * ld["tv"]).backward()
    res["out"], res["out_mask"] = out.detach().numpy(), out_mask.detach().numpy()
    for k_ in ("hole", "valid", "tv"):
        res["loss_" + k_] = np.array(ld[k_].item())
    cs = checksums((k, p.grad) for k, p in g.named_parameters())
    res["grad_keys"], res["grad_cs"] = np.array(list(cs.keys())), np.stack(list(cs.values()))
    named = dict(g.named_parameters())
    for k in ("enc_1.conv.weight", "dec_1.conv.weight", "input_enc_1.conv.weight", "input_enc_1.bn.weight",
              "last_conv.conv.weight", "last_conv.conv.bias"):
        res["grad::" + k] = named[k].grad.numpy()
    sd = g.state_dict()
    for k in ("enc_2.bn.running_mean", "input_enc_1.bn.running_var"):
        res["buf::" + k] = sd[k].numpy()
    np.savez_compressed(os.path.join(HERE, f"infill2d_c{c}_{h}x{w}.npz" if mode == "nearest" else f"infill2d_{mode}_c{c}_{h}x{w}.npz"),
                        meta=json.dumps(dict(c=c, h=h, w=w, n=n, layers=layers, seed=71, field_seed=171, mode=mode)), **res)
    print("infill2d goldens written:", {k_: float(res["loss_" + k_]) for k_ in ("hole", "valid", "tv")})


if __name__ == "__main__":
    which = sys.argv[1:] or ["all"]
    if "all" in which or "infill3d" in which:
        golden_infill3d()
    if "all" in which or "infill3d_options" in which:
        golden_infill3d_options()
    if "all" in which or "gan3d_deconv" in which:
        golden_gan3d_deconv()
    if "all" in which or "infill2d" in which:
        golden_infill2d()
    if "all" in which or "infill2d_bilinear" in which:
        golden_infill2d(h=22, w=26, mode="bilinear")
    if "all" in which or "infill3d_trajectory" in which:
        golden_infill_trajectory()
    if "all" in which or "gan3d" in which:
        golden_gan3d()
    if "all" in which or "keys" in which:
        g, gs = build_ref_generator(16, nn.BatchNorm2d, 0)
        d, ds = build_ref_discriminator(16, 64, 64, nn.BatchNorm2d, 0)
        with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
            json.dump({"generator_c16": [[k, list(v.shape)] for k, v in g.state_dict().items()],
                       "discriminator_c16_64x64": [[k, list(v.shape)] for k, v in d.state_dict().items()]}, f)
    if "all" in which or "modules" in which:
        golden_modules()
    if "all" in which or "losses" in which:
        golden_losses()
    if "all" in which or "gen" in which:
        golden_generator("c4_64x64", 4, 64, 64, 2, seed=1)
        golden_generator("c8_40x56", 8, 40, 56, 2, seed=2)
    if "all" in which or "gen256" in which:
        golden_generator_crop("c16_256x256", 16, 256, 256, 4, seed=12)
    if "all" in which or "gen_noise" in which:
        golden_generator_noise("nd1_c4_40x56", 4, 1, 40, 56, 2, seed=8)
        golden_generator_noise("nd2n_c4_40x56", 4, 2, 40, 56, 2, seed=9, noise_type="Normal")
    if "all" in which or "cpt" in which:
        golden_checkpoint_structure()
    if "all" in which or "deconv" in which:
        # the Deconv upsamplers only exist on H = 16a-13, W = 16b-11 grids (19x37 is the GPS-RO grid)
        golden_generator("deconv_c4_19x37", 4, 19, 37, 2, seed=6, upsampler="Deconv",
                         grad_full=["model.upsample.deconv1.1.bias", "model.upsample.deconv2.1.weight",
                                    "model.upsample.conv1.6.bias", "model.upsample.deconv3.1.weight",
                                    "model.upsample.last_deconv.0.weight", "model.xception_features.bn1.weight"],
                         bufs=["model.upsample.deconv1.1.running_mean", "model.upsample.deconv3.1.running_var"])
        golden_generator("deconv1x_c4_19x37", 4, 19, 37, 2, seed=7, upsampler="Deconv1x",
                         grad_full=["model.upsample.last_deconv.0.weight", "model.upsample_extension.conv1.0.weight",
                                    "model.upsample_extension.init_norm.0.weight",
                                    "model.upsample_extension.conv2.3.weight", "model.upsample.deconv2.1.bias"],
                         bufs=["model.upsample_extension.init_norm.0.running_mean",
                               "model.upsample_extension.conv2.1.running_var"])
    if "all" in which or "disc" in which:
        golden_discriminator("c4_64x64_bn", 4, 64, 64, 2, 3, nn.BatchNorm2d)
        golden_discriminator("c8_40x56_bn", 8, 40, 56, 3, 4, nn.BatchNorm2d)
        golden_discriminator("c4_64x64_in", 4, 64, 64, 2, 5, nn.InstanceNorm2d)
    if "all" in which or "gp" in which:
        golden_gradient_penalty()
    if "all" in which or "traj" in which:
        # One full loop iteration (D-step then G-step) with the launcher's Adam settings.
        # Longer trajectories of the full nets at N=2, 64x64 are chaotic: an fp32 and an fp64
        # evaluation of the same graph differ by 4-25 % in d_loss at the second iteration
        # (Adam's first steps are sign-like and the 140-layer chain amplifies rounding noise),
        # so they cannot pin anything; the multi-step loop semantics are pinned on the
        # well-conditioned one-layer nets of configs[0] (c1_plumbing) instead.
        golden_trajectory("mmm", "ModifiedMinMax", steps=1)
        golden_trajectory("wgp", "Wasserstein", steps=1)
    if "all" in which or "c1" in which:
        golden_c1_plumbing()
