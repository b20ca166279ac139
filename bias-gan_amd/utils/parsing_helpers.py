"""Optimiser / LR-schedule factories with the reference's signatures
(utils/parsing_helpers.py:8-43).  Adam and AdamW run as ONE fused HIP launch over
the network's flat parameter arena (runtime.Arena)."""
from __future__ import annotations

import torch
import torch.optim as optim

from .. import _lib as L
from .. import runtime


class FusedAdam(optim.Optimizer):
    """torch.optim.Adam / AdamW arithmetic (betas 0.9/0.999) in bg_adam_step.

    If the parameters are all the trainable tensors of one arena the whole update
    is a single kernel over the flat buffers, which also refreshes the bf16 weight
    copy; otherwise it falls back to one launch per parameter segment."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled))
        self._m = {}
        self._v = {}
        self._t = 0

    def _arenas(self):
        out = {}
        for g in self.param_groups:
            for p in g["params"]:
                a = runtime.arena_of(p)
                if a is None:
                    raise RuntimeError("FusedAdam: parameter is not in an arena yet (run a forward pass on the GPU first)")
                out.setdefault(id(a), (a, g, []))[2].append(a.by_param[id(p)])
        return list(out.values())

    def zero_grad(self, set_to_none: bool = False):
        for a, _, _ in self._arenas_or_empty():
            a.zero_grad()
            a.attach_grads()

    def _arenas_or_empty(self):
        try:
            return self._arenas()
        except RuntimeError:
            return []

    @torch.no_grad()
    def step(self, closure=None):
        self._t += 1
        for a, g, slots in self._arenas():
            b1, b2 = g["betas"]
            bc1, bc2 = 1.0 - b1 ** self._t, 1.0 - b2 ** self._t
            if id(a) not in self._m:
                self._m[id(a)] = torch.zeros_like(a.master)
                self._v[id(a)] = torch.zeros_like(a.master)
            m, v = self._m[id(a)], self._v[id(a)]
            scale = 1.0
            ddp = getattr(a, "ddp", None)
            if ddp is not None:
                ddp.finish()           # gradient all-reduce (SUM) must have landed
                scale = 1.0 / ddp.world_size
            full = len(slots) == len(a.slots)   # the parameter list covers the whole arena
            segs = [(0, a.numel)] if full else [(s.off, s.numel) for s in slots]
            for off, n in segs:
                L.call("bg_adam_step", a.master.data_ptr() + 4 * off, a.grad.data_ptr() + 4 * off, m.data_ptr() + 4 * off,
                       v.data_ptr() + 4 * off, None if a.lp is None else a.lp.data_ptr() + 2 * off, n, float(g["lr"]),
                       float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), int(bool(g["decoupled"])), bc1,
                       bc2, scale)
            a.refresh_copies(cast=False)   # CRSK copies for the data-gradient GEMMs
            a._synced_version = a.master._version

    def state_dict(self):
        sd = {"t": self._t, "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}
        for i, (a, _, _) in enumerate(self._arenas_or_empty()):
            if id(a) in self._m:
                sd[f"exp_avg_{i}"] = self._m[id(a)].clone()
                sd[f"exp_avg_sq_{i}"] = self._v[id(a)].clone()
        return sd

    def load_state_dict(self, sd):
        self._t = sd["t"]
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
        arenas = self._arenas_or_empty()
        if not arenas and any(k.startswith("exp_avg_") for k in sd):
            raise RuntimeError("FusedAdam.load_state_dict: the parameter arenas do not exist yet; call module.arena() "
                               "(module on the GPU) before restoring optimiser moments")
        for i, (a, _, _) in enumerate(arenas):
            if f"exp_avg_{i}" in sd:
                self._m[id(a)] = sd[f"exp_avg_{i}"].to(a.device).clone()
                self._v[id(a)] = sd[f"exp_avg_sq_{i}"].to(a.device).clone()


def get_optimizer(parameters, optimizer_name, start_lr, adam_eps, weight_decay):
    if isinstance(parameters, torch.nn.Module):   # the reference script hands the module over (train_gan.py:155-156)
        parameters = parameters.parameters()
    if optimizer_name == "Adam":
        optimizer = FusedAdam(parameters, lr=start_lr, eps=adam_eps, weight_decay=weight_decay)
    elif optimizer_name == "AdamW":
        optimizer = FusedAdam(parameters, lr=start_lr, eps=adam_eps, weight_decay=weight_decay, decoupled=True)
    else:
        raise NotImplementedError("Error, optimizer {} not supported".format(optimizer_name))
    optimizer.param_groups[0]["initial_lr"] = start_lr
    return optimizer


def get_lr_schedule(start_lr, scheduler_arg, optimizer, last_step=-1):
    init_step = last_step if last_step > 0 else -1
    if scheduler_arg["type"] == "multistep":
        if isinstance(scheduler_arg["milestones"], str):
            milestones = [int(x) for x in scheduler_arg["milestones"].split()]
        elif isinstance(scheduler_arg["milestones"], list):
            milestones = [int(x) for x in scheduler_arg["milestones"]]
        else:
            raise NotImplementedError("milestones variable has to be either a string or a list")
        gamma = float(scheduler_arg["decay_rate"])
        return optim.lr_scheduler.MultiStepLR(optimizer, milestones=milestones, gamma=gamma, last_epoch=init_step)
    elif scheduler_arg["type"] == "cosine_annealing":
        return optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=scheduler_arg["t_max"],
                                                    eta_min=scheduler_arg["eta_min"], last_epoch=init_step)
    else:
        raise ValueError("Error, scheduler type {} not supported.".format(scheduler_arg["type"]))
