"""RCCL under the reducer: the product's FlatAllReduce / DistributedModel over torch.distributed's "nccl" backend
(= RCCL on ROCm).  One MI355X is all a test box has, so the communicator has ONE rank: the collectives are real RCCL calls on the
reducer's own stream (library-owned HIP stream, async work handles, the two-instalment launch), their result is the identity.
Runs in a child process: a process group must not leak into the other tests of this session.  What this does NOT show is a
transfer between GPUs -- that needs the driver's 8-GPU node."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import bias_gan_amd  # noqa
from bias_gan_amd.comm.distributed import FlatAllReduce, DistributedModel
from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", world_size=1, rank=0)
    assert dist.get_backend() == "nccl"
    probe = torch.ones(4, device="cuda")
    dist.all_reduce(probe)                       # the communicator is built lazily: its set-up belongs to "initialisation"
    torch.cuda.synchronize()
except Exception as e:                           # the box's RCCL / network set-up, not the product
    print("RCCL_INIT_FAILED", repr(e))
    sys.exit(3)
n = 3 * 1024 * 1024 + 17
flat = torch.randn(n, device="cuda")
want = flat.clone()
far = FlatAllReduce(flat, bucket_elems=1 << 20)
far.world_size = 2                      # one rank, but drive the reducer as a data-parallel run would
far.log = []
far.launch_range(n // 2, n)             # the arena's tail, while "backward" continues
flat[: n // 2].mul_(2.0)                # ... and the front is still being written on the main stream
want[: n // 2].mul_(2.0)
far.launch()
far.finish()
torch.cuda.synchronize()
assert torch.equal(flat, want), (flat - want).abs().max().item()
assert far.log == [("launch", n // 2, n), ("launch", 0, n // 2), ("finish",)], far.log
# the wrapper the trainers use: broadcast of the parameters from rank 0, then a gradient round over RCCL
G = dxg.Generator(4, 4, "Interpolate", "Uniform", 0, normalizer=torch.nn.BatchNorm2d, compute_dtype=torch.bfloat16).cuda()
M = DistributedModel(G)
x = torch.randn(2, 4, 64, 64, device="cuda")
y = M(x)
y.float().abs().mean().backward()
a = G.arena()
g0 = a.grad.clone()
a.ddp.world_size = 2
M.launch_grad_allreduce()
a.ddp.finish()
torch.cuda.synchronize()
assert torch.equal(a.grad, g0) and g0.abs().sum().item() > 0
dist.destroy_process_group()
print("RCCL_OK", n)
'''


def test_reducer_runs_over_rccl_with_one_rank():
    port = 29600 + os.getpid() % 300
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", NCCL_SOCKET_IFNAME="lo")
    try:
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "port": port}], capture_output=True, text=True, timeout=240, env=env)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL did not come up within 240 s on this box (communicator set-up, not the reducer)")
    if r.returncode == 3 and "RCCL_INIT_FAILED" in r.stdout:
        pytest.skip("RCCL could not be initialised on this box: " + r.stdout.strip()[-300:])
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
