"""Checkpoint interchange with the reference (train_gan.py:401-411, comm/distributed.py:130-157): the optimiser entries
of a .cpt are torch.optim.Adam.state_dict() objects.  FusedAdam (one fused launch over the flat parameter arena) must
emit and accept exactly that layout, and a resumed run must continue the LR schedule where it stopped."""
import copy
import os

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

import bias_gan_amd  # noqa: E402,F401
from bias_gan_amd.architecture.gpsro import deeplab as dl  # noqa: E402
from bias_gan_amd.utils import parsing_helpers as ph  # noqa: E402

DEV = "cuda"


class _Net(dl.BGModule):
    """conv 3x3 -> BatchNorm -> LeakyReLU -> separable conv -> 1x1 conv with bias: every parameter layout of the arena
    (dense KRSC, depthwise, vectors, padded channel counts)."""

    def __init__(self):
        super().__init__()
        self.c1 = dl.Conv2d(12, 40, 3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(40)
        self.sep = dl.SeparableConv2d_same(40, 24)
        self.c2 = dl.Conv2d(24, 4, 1)

    def forward(self, x):
        from bias_gan_amd import ops
        from bias_gan_amd.runtime import pad_to, vec_of
        dt = self.compute_dtype()
        h = ops.ToInternal.apply(x, pad_to(12, vec_of(dt)), dt)
        h = dl.conv_norm(self, self.c1, self.bn, h, act=True)
        h = self.c2(self.sep(h))
        return ops.FromInternal.apply(h, 4)


def _twin(net):
    """CPU parameters with the reference's shapes, in module.parameters() order, + torch.optim.Adam."""
    ps = [nn.Parameter(p.detach().cpu().clone().contiguous()) for p in net.parameters()]
    return ps


def _run(net, x):
    net.zero_grad()
    net(x).square().mean().backward()


@pytest.mark.parametrize("name", ["Adam", "AdamW"])
def test_optimizer_state_is_torch_adam_layout(name):
    torch.manual_seed(3)
    net = _Net().to(DEV)
    net.set_compute_dtype(torch.float32)
    net.train()
    opt = ph.get_optimizer(net.parameters(), name, 1e-2, 1e-8, 1e-3)
    ps = _twin(net)
    ref = (torch.optim.Adam if name == "Adam" else torch.optim.AdamW)(ps, lr=1e-2, eps=1e-8, weight_decay=1e-3)
    ref.param_groups[0]["initial_lr"] = 1e-2
    xs = [torch.randn(2, 12, 9, 7, device=DEV) for _ in range(4)]
    assert opt.state_dict()["state"] == {}                      # nothing before the first step, like torch
    for x in xs[:2]:
        _run(net, x)
        for p, q in zip(net.parameters(), ps):
            q.grad = p.grad.detach().cpu().clone().contiguous()
        opt.step()
        ref.step()
    for p, q in zip(net.parameters(), ps):
        assert torch.allclose(p.detach().cpu(), q.detach(), rtol=1e-5, atol=1e-6)
    ours, theirs = opt.state_dict(), ref.state_dict()
    # structure: the same keys, parameter indices and step counts
    assert set(ours.keys()) == {"state", "param_groups"}
    assert list(ours["state"].keys()) == list(theirs["state"].keys()) == list(range(len(ps)))
    assert ours["param_groups"][0]["params"] == theirs["param_groups"][0]["params"]
    for k in ("lr", "betas", "eps", "weight_decay", "amsgrad", "initial_lr"):
        assert ours["param_groups"][0][k] == theirs["param_groups"][0][k], k
    for i in theirs["state"]:
        a, b = ours["state"][i], theirs["state"][i]
        assert set(a.keys()) == set(b.keys()) == {"step", "exp_avg", "exp_avg_sq"}
        assert float(a["step"]) == float(b["step"]) == 2.0
        assert a["exp_avg"].shape == b["exp_avg"].shape == ps[i].shape and a["exp_avg"].is_contiguous()
        assert torch.allclose(a["exp_avg"].cpu(), b["exp_avg"], rtol=2e-6, atol=1e-10)
        assert torch.allclose(a["exp_avg_sq"].cpu(), b["exp_avg_sq"], rtol=2e-6, atol=1e-14)

    # (1) torch's state into a FRESH FusedAdam (the reference's .cpt read by this package) ...
    net2 = _Net().to(DEV)
    net2.set_compute_dtype(torch.float32)
    net2.load_state_dict(net.state_dict())
    net2.train()
    net2.arena()
    opt2 = ph.get_optimizer(net2.parameters(), name, 5e-1, 1e-3, 0.0)      # hyper-parameters come from the checkpoint
    opt2.load_state_dict(copy.deepcopy(theirs))
    assert opt2.param_groups[0]["lr"] == 1e-2 and opt2.param_groups[0]["eps"] == 1e-8
    # ... and this package's state into torch.optim.Adam (its .cpt read by the reference)
    ps3 = [nn.Parameter(q.detach().clone()) for q in ps]
    ref3 = (torch.optim.Adam if name == "Adam" else torch.optim.AdamW)(ps3, lr=5e-1)
    ref3.load_state_dict({"state": {i: {k: v.cpu() for k, v in st.items()} for i, st in ours["state"].items()},
                          "param_groups": ours["param_groups"]})
    # the next update is the same in all four optimisers
    _run(net, xs[2])
    grads = [p.grad.detach().cpu().clone().contiguous() for p in net.parameters()]
    _run(net2, xs[2])
    for q, q3, g in zip(ps, ps3, grads):
        q.grad, q3.grad = g.clone(), g.clone()
    opt.step(), opt2.step(), ref.step(), ref3.step()
    for p, p2, q, q3 in zip(net.parameters(), net2.parameters(), ps, ps3):
        assert torch.allclose(p.detach().cpu(), q.detach(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(p2.detach().cpu(), q.detach(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(q3.detach(), q.detach(), rtol=1e-5, atol=1e-6)
    # the moments follow the parameters into a rebuilt arena (module.to() / set_compute_dtype())
    before = opt.state_dict()
    net.set_compute_dtype(torch.bfloat16)
    net.arena()
    opt._moments(net.arena())
    after = opt.state_dict()
    for i in before["state"]:
        assert torch.equal(before["state"][i]["exp_avg"], after["state"][i]["exp_avg"])
        assert torch.equal(before["state"][i]["exp_avg_sq"], after["state"][i]["exp_avg_sq"])


def test_resume_continues_lr_schedule(tmp_path):
    """--checkpoint + --lr_schedule_*: restore first, then build the schedules with last_step = the restored step
    (train_gan.py:163-173).  Milestones 1 and 2, saved at step 3: a schedule restarted at -1 would decay twice more."""
    from bias_gan_amd.gpsro_train import train_gan as tg
    out = str(tmp_path)
    base = ["--channels", "0", "1", "2", "3", "--synthetic_size", "64", "64", "--local_batch_size", "2", "--amp_opt_level", "O0",
            "--logging_frequency", "100", "--output_dir", out, "--model_prefix", "net", "--noise_dimensions", "0",
            "--start_lr_generator", "1e-3", "--start_lr_discriminator", "2e-3",
            "--lr_schedule_generator", "type=multistep,milestones=1 2,decay_rate=0.5",
            "--lr_schedule_discriminator", "type=cosine_annealing,t_max=8,eta_min=0.0"]
    p = tg.build_parser()
    args = p.parse_args(base + ["--max_steps", "5", "--save_frequency", "3"])
    full = tg.main(args)
    ck = os.path.join(out, "net_step_3.cpt")                    # <model_prefix>_step_<N>.cpt (train_gan.py:411)
    assert os.path.isfile(ck)
    sd = torch.load(ck)
    assert sd["step"] == 3 and set(sd) == {"step", "epoch", "generator", "discriminator", "g_opt", "d_opt", "amp"}
    assert abs(sd["g_opt"]["param_groups"][0]["lr"] - 0.25e-3) < 1e-12     # both milestones passed
    args2 = p.parse_args(base + ["--max_steps", "5", "--checkpoint", ck])
    resumed = tg.main(args2)
    assert resumed.step_count == full.step_count == 5
    for a, b in ((full.g_opt, resumed.g_opt), (full.d_opt, resumed.d_opt)):
        assert abs(a.param_groups[0]["lr"] - b.param_groups[0]["lr"]) <= 1e-9 * a.param_groups[0]["lr"]
    assert abs(full.g_opt.param_groups[0]["lr"] - 0.25e-3) < 1e-12
    assert resumed.g_scheduler.last_epoch == full.g_scheduler.last_epoch
