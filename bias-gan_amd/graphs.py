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
