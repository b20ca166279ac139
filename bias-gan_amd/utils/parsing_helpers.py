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
        self._mom = {}     # id(arena) -> (arena, exp_avg flat, exp_avg_sq flat): the arena reference keeps the key valid
        self._t = 0
        self._hyper = {}   # id(arena) -> device [4] floats {lr, bias_corr1, bias_corr2, grad_scale} (captured steps)
        self.capturing = False   # True while a training step is captured into a hipGraph (gpsro_train.train_gan)

    # -- captured (hipGraph) steps -------------------------------------------------------------------
    # A replayed step cannot take lr / bias corrections from launch arguments (they are baked in at capture), so the
    # captured launch is bg_adam_step_dev, which reads them from four device floats; prepare_replay() advances the step
    # count and rewrites those floats before the capture and before every replay.
    def _hyper_of(self, a):
        h = self._hyper.get(id(a))
        if h is None or h.device != a.master.device:
            h = self._hyper[id(a)] = torch.zeros(4, dtype=torch.float32, device=a.master.device)
        return h

    def prepare_replay(self):
        self._t += 1
        for a, g, _ in self._arenas():
            b1, b2 = g["betas"]
            ddp = getattr(a, "ddp", None)
            L.call("bg_set_floats", self._hyper_of(a).data_ptr(), 4, float(g["lr"]), 1.0 - b1 ** self._t, 1.0 - b2 ** self._t,
                   1.0 if ddp is None else 1.0 / ddp.world_size)

    def _moments(self, a):
        """Flat first/second moment buffers over arena `a`.  If the parameters moved to a new arena since the moments
        were created (module.to(), set_compute_dtype()), the values migrate parameter by parameter."""
        ent = self._mom.get(id(a))
        if ent is not None and ent[0] is a:
            return ent[1], ent[2]
        m, v = torch.zeros_like(a.master), torch.zeros_like(a.master)
        for key, (old, om, ov) in list(self._mom.items()):
            moved = False
            for s_new in a.slots:
                s_old = old.by_param.get(id(s_new.param))
                if s_old is not None and s_old.param is s_new.param:
                    a._view(m, s_new).copy_(old._view(om, s_old).to(m.device))
                    a._view(v, s_new).copy_(old._view(ov, s_old).to(v.device))
                    moved = True
            if moved:
                del self._mom[key]
        self._mom[id(a)] = (a, m, v)
        return m, v

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
        if self.capturing:
            return self._step_captured()
        self._t += 1
        for a, g, slots in self._arenas():
            b1, b2 = g["betas"]
            bc1, bc2 = 1.0 - b1 ** self._t, 1.0 - b2 ** self._t
            m, v = self._moments(a)
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

    def _step_captured(self):
        for a, g, slots in self._arenas():
            b1, b2 = g["betas"]
            m, v = self._moments(a)
            if getattr(a, "ddp", None) is not None:
                raise RuntimeError("FusedAdam: captured steps are single-process (the gradient all-reduce is not captured)")
            full = len(slots) == len(a.slots)
            segs = [(0, a.numel)] if full else [(s.off, s.numel) for s in slots]
            for off, n in segs:
                L.call("bg_adam_step_dev", a.master.data_ptr() + 4 * off, a.grad.data_ptr() + 4 * off, m.data_ptr() + 4 * off,
                       v.data_ptr() + 4 * off, None if a.lp is None else a.lp.data_ptr() + 2 * off, n, self._hyper_of(a).data_ptr(),
                       float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), int(bool(g["decoupled"])))
            a.refresh_copies(cast=False)
            a._synced_version = a.master._version

    # -- checkpoint interchange ----------------------------------------------------------------------
    # The reference stores torch.optim.Adam.state_dict() under 'g_opt' / 'd_opt' (train_gan.py:401-411) and
    # restores it with load_state_dict (comm/distributed.py:137-139):
    #   {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{'lr', 'betas', 'eps', 'weight_decay',
    #    'amsgrad', ..., 'params': [i, ...]}]},  i = position of the parameter in module.parameters().
    # state_dict() emits exactly that (per-parameter tensors in the reference's [Cout,Cin,kh,kw] shapes, cut from the
    # flat arena moments); load_state_dict() accepts it -- and the flat layout of this package's round-1 checkpoints.
    def state_dict(self):
        groups, state, idx = [], {}, 0
        have = bool(self._mom) and self._t > 0
        for g in self.param_groups:
            pg = {k: v for k, v in g.items() if k not in ("params", "decoupled")}
            pg.setdefault("amsgrad", False)
            pg.setdefault("maximize", False)
            pg.setdefault("foreach", None)
            pg.setdefault("capturable", False)
            pg.setdefault("differentiable", False)
            pg.setdefault("fused", None)
            pg.setdefault("decoupled_weight_decay", bool(g.get("decoupled", False)))   # torch >= 2.7 keeps Adam/AdamW apart by this
            pg["params"] = list(range(idx, idx + len(g["params"])))
            for p in g["params"]:
                a = runtime.arena_of(p)
                if have and a is not None and id(a) in self._mom and self._mom[id(a)][0] is a and p.requires_grad:
                    _, m, v = self._mom[id(a)]
                    sl = a.by_param[id(p)]
                    state[idx] = {"step": torch.tensor(float(self._t)),
                                  "exp_avg": a._view(m, sl).detach().clone().contiguous(),
                                  "exp_avg_sq": a._view(v, sl).detach().clone().contiguous()}
                idx += 1
            groups.append(pg)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "state" not in sd:                      # flat layout of this package's earlier checkpoints
            return self._load_flat(sd)
        saved = sd["param_groups"]
        if len(saved) != len(self.param_groups):
            raise ValueError("FusedAdam.load_state_dict: the checkpoint has a different number of parameter groups")
        order = []
        for g, sg in zip(self.param_groups, saved):
            if len(sg["params"]) != len(g["params"]):
                raise ValueError("FusedAdam.load_state_dict: a parameter group's size does not match the checkpoint's")
            for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
                if k in sg:
                    g[k] = tuple(sg[k]) if k == "betas" else sg[k]
            order += list(zip(sg["params"], g["params"]))
        state = sd["state"]
        if not state:
            self._t = 0
            self._mom = {}
            return
        steps = set()
        for i, p in order:
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            a = runtime.arena_of(p)
            if a is None:
                raise RuntimeError("FusedAdam.load_state_dict: the parameter arenas do not exist yet; call module.arena() "
                                   "(module on the GPU) before restoring optimiser moments")
            m, v = self._moments(a)
            sl = a.by_param[id(p)]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"FusedAdam.load_state_dict: moment of parameter {i} has shape {tuple(st['exp_avg'].shape)}, "
                                 f"the parameter {tuple(p.shape)}")
            a._view(m, sl).copy_(st["exp_avg"].to(device=m.device, dtype=torch.float32))
            a._view(v, sl).copy_(st["exp_avg_sq"].to(device=v.device, dtype=torch.float32))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            # torch keeps one step count per parameter (a parameter without a gradient skips its update); the fused
            # update has one bias correction for the whole arena
            raise NotImplementedError(f"FusedAdam.load_state_dict: per-parameter step counts differ ({sorted(steps)})")
        self._t = steps.pop() if steps else 0

    def _load_flat(self, sd):
        self._t = sd["t"]
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
        arenas = self._arenas_or_empty()
        if not arenas and any(k.startswith("exp_avg_") for k in sd):
            raise RuntimeError("FusedAdam.load_state_dict: the parameter arenas do not exist yet; call module.arena() "
                               "(module on the GPU) before restoring optimiser moments")
        for i, (a, _, _) in enumerate(arenas):
            if f"exp_avg_{i}" in sd:
                self._mom[id(a)] = (a, sd[f"exp_avg_{i}"].to(a.device).clone(), sd[f"exp_avg_sq_{i}"].to(a.device).clone())


class FusedLAMB(FusedAdam):
    """apex.optimizers.FusedLAMB as the reference constructs it (parsing_helpers.py:13-14: lr, eps, weight_decay; apex's
    other defaults: betas (0.9, 0.999), bias_correction, adam_w_mode, grad_averaging, max_grad_norm 1.0, no nvlamb) over the
    flat parameter arena -- bg_sumsq_f32 (global gradient norm), bg_lamb_stage1, bg_lamb_stage2; a "tensor" of the trust
    ratio is a parameter's arena slot.  apex is not part of the reference tree: the arithmetic follows its published
    two-stage kernel (see include/bgamd.h); moments, checkpoints and the DDP hand-over are FusedAdam's.  Steps run eagerly
    (the trainers capture FusedAdam steps only)."""

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, adam_w_mode=True,
                 grad_averaging=True, max_grad_norm=1.0, use_nvlamb=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=adam_w_mode)
        for g in self.param_groups:
            g.update(bias_correction=bool(bias_correction), grad_averaging=bool(grad_averaging), max_grad_norm=float(max_grad_norm),
                     use_nvlamb=bool(use_nvlamb))
        self._seg = {}     # id(arena) -> (arena, device int64 offsets [nseg + 1] of the slots of this optimiser, scratch)

    def _segments(self, a, slots):
        ent = self._seg.get(id(a))
        if ent is None or ent[0] is not a or ent[1].device != a.master.device:
            segs = [(s_.off, s_.off + s_.numel) for s_ in sorted(slots, key=lambda s_: s_.off)]
            # the kernels take consecutive [seg[t], seg[t+1]) ranges: the alignment gaps between slots become filler tensors
            # (arena padding: p = g = m = v = 0 there, so update 0 and norms 0)
            cuts = []
            pos = segs[0][0]
            for lo, hi in segs:
                if lo > pos:
                    cuts.append(pos)
                cuts.append(lo)
                pos = hi
            cuts.append(pos)
            seg = torch.tensor(cuts, dtype=torch.int64, device=a.master.device)
            ent = self._seg[id(a)] = (a, seg, torch.zeros(2 * (len(cuts) - 1) + 2, dtype=torch.float32, device=a.master.device),
                                      torch.zeros(1, dtype=torch.float64, device=a.master.device))
        return ent[1], ent[2], ent[3]

    def prepare_replay(self):
        raise RuntimeError("FusedLAMB: steps are not captured into hipGraphs")

    @torch.no_grad()
    def step(self, closure=None):
        if self.capturing:
            raise RuntimeError("FusedLAMB: steps are not captured into hipGraphs")
        self._t += 1
        for a, g, slots in self._arenas():
            b1, b2 = g["betas"]
            bc1, bc2 = (1.0 - b1 ** self._t, 1.0 - b2 ** self._t) if g["bias_correction"] else (1.0, 1.0)
            m, v = self._moments(a)
            scale = 1.0
            ddp = getattr(a, "ddp", None)
            if ddp is not None:
                ddp.finish()
                scale = 1.0 / ddp.world_size
            seg, norms, gsq = self._segments(a, slots)
            nseg = seg.numel() - 1
            norms.zero_()
            gsq.zero_()
            # the global norm runs over the gradients of THIS optimiser's parameters (apex: all param groups of the instance);
            # gaps hold zeros, so a parameter list that covers the arena is one launch
            spans = [(0, a.numel)] if len(slots) == len(a.slots) else [(s_.off, s_.numel) for s_ in slots]
            for off, cnt in spans:
                L.call("bg_sumsq_f32", a.grad.data_ptr() + 4 * off, cnt, scale, gsq.data_ptr())
            pn, un = norms.data_ptr(), norms.data_ptr() + 4 * (nseg + 1)
            L.call("bg_lamb_stage1", a.master.data_ptr(), a.grad.data_ptr(), m.data_ptr(), v.data_ptr(), seg.data_ptr(), nseg,
                   gsq.data_ptr(), float(g["max_grad_norm"]), float(b1), float(b2), int(g["grad_averaging"]), float(g["eps"]),
                   float(g["weight_decay"]), int(bool(g["decoupled"])), bc1, bc2, scale, pn, un)
            L.call("bg_lamb_stage2", a.master.data_ptr(), a.grad.data_ptr(), None if a.lp is None else a.lp.data_ptr(), seg.data_ptr(),
                   nseg, pn, un, float(g["lr"]), float(g["weight_decay"]), int(g["use_nvlamb"]))
            a.refresh_copies(cast=False)
            a._synced_version = a.master._version


def get_optimizer(parameters, optimizer_name, start_lr, adam_eps, weight_decay):
    if isinstance(parameters, torch.nn.Module):   # the reference script hands the module over (train_gan.py:155-156)
        parameters = parameters.parameters()
    if optimizer_name == "Adam":
        optimizer = FusedAdam(parameters, lr=start_lr, eps=adam_eps, weight_decay=weight_decay)
    elif optimizer_name == "AdamW":
        optimizer = FusedAdam(parameters, lr=start_lr, eps=adam_eps, weight_decay=weight_decay, decoupled=True)
    elif optimizer_name == "LAMB":
        optimizer = FusedLAMB(parameters, lr=start_lr, eps=adam_eps, weight_decay=weight_decay)
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
        # the command line delivers strings (StoreDictKeyPair, train_gan.py:41-47)
        return optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=int(scheduler_arg["t_max"]),
                                                    eta_min=float(scheduler_arg["eta_min"]), last_epoch=init_step)
    else:
        raise ValueError("Error, scheduler type {} not supported.".format(scheduler_arg["type"]))


def resume_lr_schedule(start_lr, scheduler_arg, optimizer, start_step):
    """Schedule for a run restored at `start_step`, continuing exactly where the saved run stopped.

    The reference builds it as get_lr_schedule(..., last_step=start_step) (train_gan.py:170-173); torch's scheduler
    constructor then performs one step of its own, so the reference's resumed run is one scheduler step AHEAD of the
    run that wrote the checkpoint (a milestone at start_step + 1 fires before the first resumed update).  Here the
    schedule is built from step 0 and moved to `start_step` with its closed form: same LR at every step as an
    uninterrupted run WHEN each scheduler stepped once per loop iteration of the saved run (--update_frequency_* 1 and
    no generator warm-up: the launcher's values).  With other update frequencies a scheduler steps only on its own
    network's updates, so `start_step` overcounts its position; the checkpoint carries no scheduler counters (the
    reference's does not either), and the resumed LR is then that of loop step `start_step`.  INTEGRATION.md records
    this deviation from the reference."""
    for g in optimizer.param_groups:
        g["lr"] = g.get("initial_lr", start_lr)
    sched = get_lr_schedule(start_lr, scheduler_arg, optimizer, last_step=-1)
    if start_step > 0:
        sched.last_epoch = int(start_step)
        sched._step_count = int(start_step) + 1
        for g, lr in zip(optimizer.param_groups, sched._get_closed_form_lr()):
            g["lr"] = lr
        sched._last_lr = [g["lr"] for g in optimizer.param_groups]
    return sched
