"""`comm` helper class with the reference's API (comm/distributed.py:10-204) for
one process per GPU over RCCL (torch.distributed backend "nccl" on ROCm).

Gradient synchronisation is NOT a per-parameter hook storm: a network's
gradients live in one flat fp32 arena (runtime.Arena), so data parallelism is a
handful of large all-reduces over that buffer on a side HIP stream
(FlatAllReduce), launched when the backward pass ends and waited for only
when the fused Adam needs the result -- the next forward pass runs meanwhile.
xGMI is point-to-point (7 links/GPU), so few large messages beat many small ones.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import _lib as _L


class FlatAllReduce:
    """SUM all-reduce of one flat tensor in `bucket_elems`-sized pieces on a side
    stream.  Works on CPU tensors too (gloo), where it is simply asynchronous.

    The buffer may be reduced in two instalments: launch_range(a, b) starts the reduction of a part that is already
    final (the tail of the gradient arena when the backward pass has left the layers that live there:
    ops.GradMilestoneFn), launch() the rest; finish() waits for both."""

    def __init__(self, flat: torch.Tensor, group=None, bucket_elems: int = 32 * 1024 * 1024):
        self.flat, self.group = flat, group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = max(1, int(bucket_elems))
        self.cuda = flat.is_cuda
        # one reduction stream per device (library-owned, never a PyTorch pool stream): every FlatAllReduce of a process
        # shares it, so the number of streams does not grow with the number of networks / arena rebuilds
        self.stream = _L.side_stream(flat.device, "allreduce") if self.cuda else None
        self.works = []
        self.launched = False
        self.done = []          # element ranges already launched in this round (launch_range)
        self.ms_pending = 0     # ops.GradMilestoneFn nodes created since the last finish() whose backward has not run
        self.log = None         # tests: a list that receives ("launch", a, b) / ("finish",) in call order

    def buckets(self, a=0, b=None):
        n = self.flat.numel() if b is None else b
        return [(o, min(o + self.bucket_elems, n)) for o in range(a, n, self.bucket_elems)]

    def _issue(self, a, b, after=()):
        if b <= a:
            return
        if self.log is not None:
            self.log.append(("launch", a, b))
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            for s in after:
                self.stream.wait_stream(s)
            with torch.cuda.stream(self.stream):
                for lo, hi in self.buckets(a, b):
                    self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            for lo, hi in self.buckets(a, b):
                self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def launch_range(self, a, b, after=()):
        """Start reducing flat[a:b] now: its values are final on the current stream (and on the streams in `after`)."""
        if self.launched or self.world_size == 1:
            return
        a, b = max(0, int(a)), min(int(b), self.flat.numel())
        for lo, hi in self.done:
            if not (b <= lo or a >= hi):
                raise RuntimeError("FlatAllReduce.launch_range: overlapping ranges in one round")
        self._issue(a, b, after)
        self.done.append((a, b))

    def launch(self):
        """Start the reduction of the buffer's current contents (call after backward): everything launch_range() has not
        taken yet."""
        if self.launched or self.world_size == 1:
            self.launched = True
            return
        pos = 0
        for lo, hi in sorted(self.done):
            self._issue(pos, lo)
            pos = max(pos, hi)
        self._issue(pos, self.flat.numel())
        self.launched = True

    def finish(self):
        """Make the reduced values visible to the current stream (launches first if nobody did)."""
        if not self.launched:
            self.launch()
        if self.log is not None:
            self.log.append(("finish",))
        for w in self.works:
            w.wait()
        self.works = []
        if self.cuda and self.world_size > 1:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        self.launched = False
        self.done = []
        self.ms_pending = 0


class DistributedModel(nn.Module):
    """Stand-in for apex.parallel.DistributedDataParallel (comm/distributed.py:195-199):
    parameters broadcast from rank 0, gradients summed over ranks (the optimiser
    divides by world size).  Keys gain the 'module.' prefix like DDP's."""

    def __init__(self, module: nn.Module, group=None):
        super().__init__()
        self.module = module
        self.group = group
        self._ready = None      # the arena the broadcast / reducer were set up for

    def _prepare(self):
        arena = self.module.arena()
        if self._ready is not arena:      # first call, or the arena was rebuilt (module.to(), set_compute_dtype())
            dist.broadcast(arena.master, src=0, group=self.group)
            arena.weights_changed()
            arena.sync()
            arena.ddp = FlatAllReduce(arena.grad, self.group)
            self._ready = arena
        return arena

    def forward(self, *a, **kw):
        self._prepare()
        return self.module(*a, **kw)

    def launch_grad_allreduce(self):
        self._prepare().ddp.launch()


class comm(object):

    def __init__(self, mode="openmpi"):
        port = "29500"
        os.environ.setdefault("MASTER_PORT", port)
        if "RANK" in os.environ and "WORLD_SIZE" in os.environ and mode != "dummy":
            # launched by torch.distributed.run: one process per GPU, RCCL over xGMI
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("BGAMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if torch.cuda.is_available():
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
            if not dist.is_initialized():
                dist.init_process_group(backend=backend)
            return
        if mode in ("openmpi", "openmpi-nccl"):
            # The reference's default "openmpi" hands the collectives to MPI (comm/distributed.py:45-47); here every
            # multi-process mode runs over RCCL, and the launcher's environment only supplies rank / size / address:
            # mpirun (OMPI_COMM_WORLD_*, PMIX_SERVER_URI2 for the address, single node like the reference's
            # "openmpi-nccl", :49-56).  Without any launcher it is one process -- MPI's singleton init -- and needs
            # no process group.
            if "OMPI_COMM_WORLD_SIZE" not in os.environ:
                os.environ.setdefault("MASTER_ADDR", "localhost")
                return
            uri = os.getenv("PMIX_SERVER_URI2")
            if uri and "//" in uri:
                os.environ["MASTER_ADDR"] = uri.split("//")[1].split(":")[0]
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            comm_rank = int(os.getenv('OMPI_COMM_WORLD_RANK', 0))
            comm_size = int(os.getenv("OMPI_COMM_WORLD_SIZE", 1))
            backend = os.environ.get("BGAMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if torch.cuda.is_available():
                torch.cuda.set_device(int(os.getenv("OMPI_COMM_WORLD_LOCAL_RANK", comm_rank)) % max(1, torch.cuda.device_count()))
            if not dist.is_initialized():
                dist.init_process_group(backend=backend, rank=comm_rank, world_size=comm_size)
        elif mode == "dummy":
            os.environ.setdefault("MASTER_ADDR", "localhost")
        else:
            raise ValueError("comm: unknown mode {!r} (openmpi, openmpi-nccl, dummy)".format(mode))

    def metric_average(self, val, name=None, op_name=None, device=None):
        """SUM over ranks (times 1/size when op_name == 'average'), returned as a
        Python float -- the reference's semantics, host sync included
        (comm/distributed.py:12-34: the default op_name=None returns the SUM)."""
        if dist.is_available() and dist.is_initialized():
            fact = 1. / float(self.size()) if op_name == "average" else 1.
            tensor = val.clone().detach().requires_grad_(False) if isinstance(val, torch.Tensor) else torch.tensor(val)
            if device is not None:
                tensor = tensor.to(device)
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
            return fact * tensor.item()
        return val.item() if isinstance(val, torch.Tensor) else val

    def printr(self, msg, rank=0):
        if self.rank() == rank:
            print(msg)

    def size(self):
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def rank(self):
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    def local_rank(self):
        if dist.is_available() and dist.is_initialized() and torch.cuda.is_available():
            return dist.get_rank() % torch.cuda.device_count()
        return 0

    def broadcast(self, tensor, root_rank, name=None):
        if dist.is_available() and dist.is_initialized():
            dist.broadcast(tensor, src=root_rank)
        return tensor

    @staticmethod
    def _strip(sd):
        return {k.replace("module.", ""): v for k, v in sd.items()}

    def init_training_state(self, model, optimizer, checkpoint_name, device_id):
        if (checkpoint_name is not None) and (os.path.isfile(checkpoint_name)):
            checkpoint = torch.load(checkpoint_name, map_location=device_id)
            optimizer.load_state_dict(checkpoint['optimizer'])
            model.load_state_dict(self._strip(checkpoint['model']))
            return checkpoint['step'], checkpoint['epoch']
        return 0, 0

    def init_gan_training_state(self, gmodel, dmodel, gopt, dopt, checkpoint_name, device_id):
        """Restore {step, epoch, generator, discriminator, g_opt, d_opt} (comm/distributed.py:130-157)."""
        if (checkpoint_name is not None) and (os.path.isfile(checkpoint_name)):
            checkpoint = torch.load(checkpoint_name, map_location=device_id)
            gopt.load_state_dict(checkpoint['g_opt'])
            dopt.load_state_dict(checkpoint['d_opt'])
            gmodel.load_state_dict(self._strip(checkpoint['generator']))
            dmodel.load_state_dict(self._strip(checkpoint['discriminator']))
            return checkpoint['step'], checkpoint['epoch']
        return 0, 0

    def DistributedModel(self, model):
        if dist.is_available() and dist.is_initialized():
            return DistributedModel(model)
        return model

    def DistributedOptimizer(self, optimizer, named_parameters, compression_name, op_name):
        return optimizer
