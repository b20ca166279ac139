"""hipGraph replay of a module's forward pass under torch.no_grad() (the D-step's generator forward).

On small fields / volumes the step is bound by the host: one training step is ~2 700 launches of 10-30 us and Python
needs ~20 us to issue each.  The D-step's generator forward builds no autograd graph, its shapes are fixed and it
depends on the host only through its input, so it is captured once (torch.cuda.CUDAGraph over the same C-ABI launches:
they go to torch's current stream, which is the capturing stream) and replayed afterwards with one launch.

What a replay must redo on the host, because the Python code that did it during capture does not run again:
  * the slots of the statistics pool (runtime.StatsPool) the captured launches accumulate into stay reserved;
  * BatchNorm's num_batches_tracked is counted on the host (folded into the buffer by state_dict()).
The parameter arena's low-precision / packed copies are refreshed eagerly BEFORE capture and before every replay, so
the captured launches never contain (or miss) a repack.  The first call with a new input signature runs eagerly (lazy
initialisation, DistributedModel's broadcast); the second captures.  BGAMD_GRAPH: 0 never, 1 always, unset: inputs of at
most 2^22 elements (the host-bound regime; the 1152x768x16 workload is GPU-bound and stays eager)."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib as L
from .runtime import StatsPool

_MODE = os.environ.get("BGAMD_GRAPH")
AUTO_MAX_ELEMS = 1 << 22


def _unwrap(m):
    return getattr(m, "module", m) if m.__class__.__name__ == "DistributedModel" else m


class NoGradGraph:
    def __init__(self, module: nn.Module):
        self.module = module
        self.entries = {}
        self.seen = set()

    def enabled_for(self, args) -> bool:
        if _MODE == "0" or L.PROFILE is not None or not all(a.is_cuda for a in args):
            return False
        if torch.cuda.is_current_stream_capturing():     # inside a captured training step: no nested graph
            return False
        root = _unwrap(self.module)
        if getattr(root, "noise_dimensions", 0):       # host RNG draws inside forward(): not capturable
            return False
        return _MODE == "1" or sum(a.numel() for a in args) <= AUTO_MAX_ELEMS

    def __call__(self, *args):
        if not self.enabled_for(args):
            with torch.no_grad():
                return self.module(*args)
        root = _unwrap(self.module)
        key = (tuple((tuple(a.shape), a.dtype, a.device.index) for a in args), root.training,
               id(getattr(root, "_bg_arena", None)))
        e = self.entries.get(key)
        if e is None and key not in self.seen:
            self.seen.add(key)
            with torch.no_grad():
                return self.module(*args)
        arena = root.arena() if hasattr(root, "arena") else None
        if arena is not None:
            arena.sync()                                # outside the graph: see the module docstring
        if e is None:
            e = self.entries[key] = self._capture(args)
            e["arena"] = arena      # the key holds id(arena): keeping the arena alive keeps that id from being reused
            return e["out"]
        pool = StatsPool.get(args[0].device)
        if pool.used != e["pool_before"]:               # someone took statistic slots earlier in this step: stay correct
            with torch.no_grad():
                return self.module(*args)
        for s, a in zip(e["static_in"], args):
            s.copy_(a, non_blocking=True)
        pool.used = e["pool_after"]
        for m, k in e["nbt"]:
            m.__dict__["_bg_nbt_pending"] = m.__dict__.get("_bg_nbt_pending", 0) + k
        e["graph"].replay()
        return e["out"]

    def _capture(self, args):
        dev = args[0].device
        pool = StatsPool.get(dev)
        static_in = [a.clone() for a in args]
        bns = [m for m in _unwrap(self.module).modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        before = [m.__dict__.get("_bg_nbt_pending", 0) for m in bns]
        pool_before = pool.used
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        with torch.no_grad(), torch.cuda.graph(g):
            out = self.module(*static_in)
        nbt = [(m, m.__dict__.get("_bg_nbt_pending", 0) - b) for m, b in zip(bns, before)]
        e = {"graph": g, "static_in": static_in, "out": out, "pool_before": pool_before, "pool_after": pool.used,
             "nbt": [(m, k) for m, k in nbt if k]}
        g.replay()                                      # capture records, it does not execute
        return e


def roll_fp8_sites(*nets):
    """End of a training iteration on the fp8 operand path: next step's exponents from this step's recorded maxima, for
    every network that is in fp8 mode (no-op otherwise).  Every trainer's _eager_step ends with it -- without the roll no
    quantisation site ever becomes ready and the GEMMs stay on bf16 operands while paying for the amax passes."""
    for net in nets:
        a = getattr(_unwrap(net), "_bg_arena", None)
        if a is not None and a.fp8:
            a.roll_fp8()


class HostStepState:
    """Everything a training step advances on the HOST, so that a step that died half-way (a failed whole-step capture:
    GANTrainer._graph_step) can be taken back before the eager re-run -- otherwise the LR schedule runs one step ahead
    for the rest of the run, num_batches_tracked over-counts and fp8 sites become 'ready' on data they never saw.

    Snapshot: the optimisers' step counts and learning rates, the schedulers' state, every BatchNorm's pending forward
    count, the trainer's step counter and pending flags, the statistics pool's slot cursor and the fp8 site sets of the
    parameter arenas.  Device memory is not part of it: a capture records launches, it does not run them."""

    def __init__(self, trainer, nets, opts, scheds, extra_attrs=()):
        import copy
        self.trainer, self.opts, self.scheds = trainer, list(opts), [s for s in scheds if s is not None]
        self.t = [o._t for o in self.opts]
        self.lrs = [[g["lr"] for g in o.param_groups] for o in self.opts]
        self.sched_sd = [copy.deepcopy(s.state_dict()) for s in self.scheds]
        self.mods = [m for n in nets for m in _unwrap(n).modules() if "_bg_nbt_pending" in m.__dict__]
        self.nbt = [m.__dict__["_bg_nbt_pending"] for m in self.mods]
        self.nets = [_unwrap(n) for n in nets]
        self.arenas = [a for a in (getattr(n, "_bg_arena", None) for n in self.nets) if a is not None and getattr(a, "fp8", False)]
        self.sites = [(set(a._site_seen), set(a._site_ready), a.sites_ready) for a in self.arenas]
        self.pools = [(p, p.used) for p in StatsPool.all()]
        self.attrs = {k: getattr(trainer, k) for k in ("step_count",) + tuple(extra_attrs) if hasattr(trainer, k)}

    def restore(self):
        for o, t, lrs in zip(self.opts, self.t, self.lrs):
            o._t = t
            for g, lr in zip(o.param_groups, lrs):
                g["lr"] = lr
        for s, sd in zip(self.scheds, self.sched_sd):
            s.load_state_dict(sd)
        seen = set(map(id, self.mods))
        for m, k in zip(self.mods, self.nbt):
            m.__dict__["_bg_nbt_pending"] = k
        for n in self.nets:                      # counters that did not exist at snapshot time
            for m in n.modules():
                if id(m) not in seen and m.__dict__.get("_bg_nbt_pending", 0):
                    m.__dict__["_bg_nbt_pending"] = 0
        for a, (seen_, ready, flag) in zip(self.arenas, self.sites):
            a._site_seen, a._site_ready, a.sites_ready = set(seen_), set(ready), flag
        for p, used in self.pools:
            p.used = used
        for k, v in self.attrs.items():
            setattr(self.trainer, k, v)
