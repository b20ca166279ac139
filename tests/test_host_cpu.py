"""CPU-only tests of the host logic: module/state_dict mirror, host RNG label
draws, optimiser/schedule plumbing, and the N>1 path (world_size 2 over gloo)."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import bias_gan_amd  # noqa: F401
from bias_gan_amd.architecture.gpsro import deeplab as dl
from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg
from bias_gan_amd.comm.distributed import FlatAllReduce, comm
from bias_gan_amd.utils import losses
from bias_gan_amd.utils import parsing_helpers as ph


def test_state_dict_keys_match_reference(golden_dir, capsys):
    ref = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    G = dxg.Generator(16, 16, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d)
    D = dxg.Discriminator(16, normalizer=nn.BatchNorm2d, input_size=(64, 64))
    assert [[k, list(v.shape)] for k, v in G.state_dict().items()] == ref["generator_c16"]
    assert [[k, list(v.shape)] for k, v in D.state_dict().items()] == ref["discriminator_c16_64x64"]
    # the reference's default head (19x37 grid) is 12288 wide (deeplab_gan.py:21)
    assert dxg.Discriminator(4, normalizer=nn.BatchNorm2d).linear.in_features == 12288
    # InstanceNorm contributes no entries (defaults: no affine, no running stats)
    Di = dxg.Discriminator(4, normalizer=nn.InstanceNorm2d, input_size=(64, 64))
    assert all("running" not in k for k in Di.state_dict())


@pytest.mark.parametrize("upsampler", ["Deconv", "Deconv1x"])
def test_deconv_state_dict_keys_match_reference(upsampler):
    """Key order and shapes of the Deconv variants: the oracle's spec was asserted equal to the reference's
    state_dict when the goldens were generated (tests/golden/make_golden.py: build_ref_generator)."""
    from oracle import gan_oracle as orc
    G = dxg.Generator(4, 4, upsampler, "Uniform", 0, normalizer=nn.BatchNorm2d)
    spec = orc.generator_spec(4, 4, 0, "batch", upsampler=upsampler)
    assert [(k, tuple(v.shape)) for k, v in G.state_dict().items()] == [(k, tuple(sh)) for k, sh, _ in spec]
    # nn_pooling=False swaps the AvgPool2d for nn.Identity without shifting any key
    from bias_gan_amd.architecture.gpsro import deeplab as dl
    Gn = dl.DeepLabv3_plus(4, 4, 16, upsampler, False, False, nn.BatchNorm2d, nn_pooling=False)
    assert ["model." + k for k in Gn.state_dict()] == [k for k, _, _ in spec]


def test_init_distributions_follow_reference():
    torch.manual_seed(0)
    D = dxg.Discriminator(16, normalizer=nn.BatchNorm2d, input_size=(64, 64))
    w = D.xception_features.block5.rep[1].pointwise.weight  # 728 -> 728, k=1
    gain = nn.init.calculate_gain("leaky_relu", 0.2)
    assert abs(w.std().item() - gain / np.sqrt(728)) < 0.05 * gain / np.sqrt(728)   # deeplab_gan.py:46-52
    assert float(D.linear.bias.abs().sum()) == 0.0
    G = dxg.Generator(16, 16, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d)
    w = G.model.xception_features.block5.rep[1].pointwise.weight
    assert abs(w.std().item() - np.sqrt(2.0 / 728)) < 0.05 * np.sqrt(2.0 / 728)    # kaiming_normal_, deeplab.py:285


def test_no_cpu_fallback():
    G = dxg.Generator(4, 4, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G(torch.zeros(2, 4, 64, 64))
    Gd = dxg.Generator(4, 4, "Deconv1x", "Uniform", 0, normalizer=nn.BatchNorm2d)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Gd(torch.zeros(2, 4, 19, 37))
    with pytest.raises(NotImplementedError):   # like the reference: no upsampler of that name (deeplab.py:642)
        dxg.Generator(4, 4, "Interpolate1x", "Uniform", 0, normalizer=nn.BatchNorm2d)
    with pytest.raises(NotImplementedError):
        dxg.Generator(4, 4, "Interpolate", "Cauchy", 0)


def test_block_unit_layout():
    b = dl.Block(128, 128, reps=2, stride=2, start_with_relu=False, normalizer=nn.BatchNorm2d)
    kinds = [type(m).__name__ for m in b.rep]
    assert kinds == ["SeparableConv2d_same", "BatchNorm2d", "LeakyReLU", "SeparableConv2d_same", "BatchNorm2d",
                     "SeparableConv2d_same"]
    assert b.skip is not None and dl.fixed_padding_extents(3, 2) == (2, 2) and dl.fixed_padding_extents(3, 1) == (1, 1)


def test_ganloss_host_draws_bit_exact(golden_dir):
    z = np.load(os.path.join(golden_dir, "losses.npz"))
    crit = losses.GANLoss("ModifiedMinMax", 4, torch.device("cpu"))
    for seed in (0, 7, 123, 999):
        torch.manual_seed(seed)
        lf, lr, swap = crit.draw_labels()
        np.testing.assert_array_equal(lf.numpy(), z[f"ModifiedMinMax_{seed}::label_fake"])
        np.testing.assert_array_equal(lr.numpy(), z[f"ModifiedMinMax_{seed}::label_real"])
        assert swap == bool(z[f"ModifiedMinMax_{seed}::swap_u"] < 0.05)
    with pytest.raises(NotImplementedError):
        losses.GANLoss("Hinge", 4, torch.device("cpu"))


def test_optimizer_factory_and_schedules():
    p = [nn.Parameter(torch.zeros(3))]
    opt = ph.get_optimizer(p, "Adam", 1e-3, 1e-8, 1e-4)
    assert opt.param_groups[0]["initial_lr"] == 1e-3 and not opt.param_groups[0]["decoupled"]
    assert ph.get_optimizer(p, "AdamW", 1e-3, 1e-8, 1e-4).param_groups[0]["decoupled"]
    lamb = ph.get_optimizer(p, "LAMB", 1e-3, 1e-8, 1e-4)     # apex FusedLAMB's defaults beside the three the reference passes
    g0 = lamb.param_groups[0]
    assert (type(lamb).__name__, g0["lr"], g0["eps"], g0["weight_decay"], g0["max_grad_norm"], g0["decoupled"], g0["grad_averaging"],
            g0["use_nvlamb"]) == ("FusedLAMB", 1e-3, 1e-8, 1e-4, 1.0, True, True, False)
    with pytest.raises(NotImplementedError):
        ph.get_optimizer(p, "SGD", 1e-3, 1e-8, 1e-4)
    sch = ph.get_lr_schedule(1e-3, {"type": "multistep", "milestones": "2 4", "decay_rate": "0.1"}, opt)
    lrs = []
    for _ in range(5):
        lrs.append(opt.param_groups[0]["lr"])
        sch.step()
    assert np.allclose(lrs, [1e-3, 1e-3, 1e-4, 1e-4, 1e-5])
    ph.get_lr_schedule(1e-3, {"type": "cosine_annealing", "t_max": 10, "eta_min": 0.0}, opt)
    with pytest.raises(ValueError):
        ph.get_lr_schedule(1e-3, {"type": "nope"}, opt)
    with pytest.raises(RuntimeError, match="not in an arena"):
        opt.step()   # parameters that never went through a GPU forward


def test_comm_dummy_single_process():
    c = comm(mode="dummy")
    assert c.size() == 1 and c.rank() == 0 and c.local_rank() == 0
    assert c.metric_average(torch.tensor(2.5), "x") == 2.5 and c.metric_average(3, "x") == 3
    m = nn.Linear(2, 2)
    assert c.DistributedModel(m) is m and c.DistributedOptimizer("o", None, None, "average") == "o"
    assert c.init_gan_training_state(None, None, None, None, None, "cpu") == (0, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    c = comm(mode="torchrun")                    # picks gloo on a CPU-only box
    assert c.size() == world and c.rank() == rank
    # metric_average: SUM by default, mean with op_name="average" (comm/distributed.py:15-17)
    s = c.metric_average(torch.tensor(float(rank + 1)), "loss")
    a = c.metric_average(torch.tensor(float(rank + 1)), "loss", op_name="average")
    # flat gradient all-reduce in several buckets, asynchronous launch then finish
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    far = FlatAllReduce(flat, bucket_elems=300)
    assert len(far.buckets()) == 4
    far.launch()
    far.finish()
    far2 = FlatAllReduce(torch.ones(7) * (rank + 1))
    far2.finish()                                # finish() alone also launches
    t = torch.full((3,), float(rank))
    c.broadcast(t, 0)
    # dataset-style sharding contract of the reference (gpsro_dataset.py:24-33): contiguous slices of one shuffle
    q.put((rank, s, a, flat.sum().item(), far2.flat.tolist(), t.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_world_size_2_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tot = world * (world + 1) / 2.0      # sum over ranks of (rank + 1)
    for rank, s, a, fsum, f2, t in res:
        assert s == tot and a == tot / world
        assert fsum == float(np.arange(1000).sum() * tot)
        assert f2 == [tot] * 7
        assert t == [0.0, 0.0, 0.0]


def test_resume_lr_schedule_continues_exactly():
    """parsing_helpers.resume_lr_schedule: a schedule restored at step s gives the LRs of an uninterrupted run from s on
    (multistep with a milestone right at / after the restore point, cosine annealing)."""
    import torch
    from bias_gan_amd.utils import parsing_helpers as ph
    for arg in ({"type": "multistep", "milestones": "2 4", "decay_rate": "0.5"},
                {"type": "cosine_annealing", "t_max": "8", "eta_min": "0.0"}):
        p = [torch.nn.Parameter(torch.zeros(3))]
        o = torch.optim.Adam(p, lr=1e-3)
        o.param_groups[0]["initial_lr"] = 1e-3
        s = ph.get_lr_schedule(1e-3, arg, o)
        lrs = []
        for _ in range(8):
            lrs.append(o.param_groups[0]["lr"])
            o.step()
            s.step()
        for start in (1, 3, 4, 5):
            o2 = torch.optim.Adam(p, lr=lrs[start])          # what the checkpoint's optimiser state carries
            o2.param_groups[0]["initial_lr"] = 1e-3
            s2 = ph.resume_lr_schedule(1e-3, arg, o2, start)
            for k in range(start, 8):
                assert abs(o2.param_groups[0]["lr"] - lrs[k]) < 1e-12, (arg["type"], start, k)
                o2.step()
                s2.step()


def test_bench_plain_multi_gpu_launch_starts_one_process_per_gpu():
    """`python bench.py --gpus N` without a launcher environment must not exit: the parent (which never touches the GPU)
    starts torch.distributed.run with N ranks on 127.0.0.1 and relays their output."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BGAMD_BENCH_LAUNCH_ECHO"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
