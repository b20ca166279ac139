#!/usr/bin/env python3
"""bench.py -- climate-field samples/s for one full G+D training step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric): 1152x768x16 synthetic fields, batch 8 per GPU,
Generator = noise-free DeepLabv3+/Xception-65 with the Interpolate upsampler,
Discriminator = Xception-65 + Linear head, ModifiedMinMax + L1 (weights 1/1),
Adam(lr 1e-4, eps 1e-8, wd 1e-5), bf16 activations/weights with fp32
accumulation, fp32 master weights / statistics / optimiser.  One "step" = one
loop iteration of the reference (train_gan.py:244-298): D-step (G forward, two D
forwards, D backward, Adam) then G-step (G forward, D forward, backward through D
and G, Adam).  Data parallel over N GPUs: one process per GPU, gradients
all-reduced over RCCL on a side stream, weak scaling.

Rank 0 prints ONE JSON line; see README/DESIGN.md for the extra objects:
  roofline     per-kernel durations measured with HIP events (one extra,
               un-timed profiled step) for the dominant MFMA kernel family
  cpu_baseline the CPU oracle (a port of the reference path on PyTorch CPU ops)
               timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402

# forward GMAC per sample at 1152x768x16 (G 313.9, D 190.2);
# a step needs 4 G-units + 8 D-units  ->  2776.9 GMAC = 5.5538 TFLOP per sample.
FWD_GMAC = {(1152, 768, 16): (313.9, 190.2), (256, 256, 16): (23.25, 14.09), (64, 64, 4): (1.439, 0.866),
            (2304, 1536, 32): (1272.9, 776.9)}


def w_alg_tflop(h, w, c, loss, g_units=4):
    """4 G-units + 8 D-units per sample and step; the gradient penalty adds one D forward and one data-gradient-only D
    backward = 2 D-units (c4 = c3 + 2 D = 6.31 TFLOP)."""
    if (h, w, c) not in FWD_GMAC:
        return None
    g, d = FWD_GMAC[(h, w, c)]
    return 2.0 * (g_units * g + (10 if loss == "wgan-gp" else 8) * d) * 1e-3
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 (the block-scaled f8f6f4 MFMA forms; same table)
PEAK_F32_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=1152)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--channels", type=int, default=16)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="bf16: the headline; f32: the parity path; fp8: bf16 storage with the dense convolutions' forward and "
                         "data-gradient GEMMs on fp8 operands (block-scaled MFMA 16x16x128, fp32 accumulation) -- BASELINE.json "
                         "configs[4] at --height 2304 --width 1536 --channels 32")
    ap.add_argument("--loss", default="mmm", choices=["mmm", "wgan-gp"],
                    help="mmm: ModifiedMinMax + L1 (configs[2], the headline); wgan-gp: Wasserstein + gradient penalty "
                         "(configs[3]: one more D forward and a data-gradient-only D backward per step)")
    ap.add_argument("--data", default="resident", choices=["resident", "ring"],
                    help="resident: two synthetic batches in HBM (the headline: inputs resident when the timed region starts); "
                         "ring: every step's batch comes from .npy files (HWC fp32, one sample per file, the CAM layout) "
                         "through the pinned-host -> HBM staging ring while the previous step runs; prints the file -> HBM "
                         "rate and is compared with the resident figure (never the headline value)")
    ap.add_argument("--reuse-g-forward", action="store_true",
                    help="OPT-IN, not the headline: one generator forward per step instead of the reference's two identical ones "
                         "(GANTrainer.reuse_g_forward); the algorithmic work of the step is then 3 G-units + 8 D-units")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-floor", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write (entry, flops, bytes, ms) of every launch of the profile step")
    return ap.parse_args()


def synthetic_batch(n, c, h, w, seed, device):
    """inputs ~ N(0,1), label = input + 0.1 N(0,1), generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn((n, c, h, w), generator=g, device=device)
    y = x + 0.1 * torch.randn((n, c, h, w), generator=g, device=device)
    return x, y


def cpu_baseline(c, h, w):
    """Time the CPU oracle (a functional port of the reference path on PyTorch CPU
    ops, same oneDNN convolutions the reference would hit) on a bounded sample of
    the SAME workload: ONE full G+D iteration at the full field size, batch 2 (the
    smallest batch BatchNorm accepts on the 1x1 global-pool branch), on the box's
    host cores.  About 20 s."""
    from oracle import gan_oracle as orc
    # the box's CPU share, not the host's core count (a 1-GPU box gets 16 cores)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("BG_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    hs, ws, n = h, w, 2
    gspec = orc.generator_spec(c, c, 0, "batch")
    dspec = orc.discriminator_spec(c, hs, ws, "batch")
    st = orc.GANStep(orc.fill_state(gspec, 1), orc.fill_state(dspec, 2), orc.trainable_keys(gspec),
                     orc.trainable_keys(dspec), "batch", "ModifiedMinMax")
    x, y = orc.synthetic_fields(n, c, hs, ws, 333)
    torch.manual_seed(333)
    print(f"[bench] cpu_baseline: timing the oracle on {cores} threads ...", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    st.step(x, y)
    dt = time.perf_counter() - t0
    print(f"[bench] cpu_baseline: {dt:.1f} s", file=sys.stderr, flush=True)
    frac = (hs * ws) / float(h * w)
    return {"value": n / dt * frac, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"one G+D iteration of oracle/gan_oracle.py (fp32, PyTorch CPU ops) at {hs}x{ws}x{c}, batch {n}: "
                      f"{dt:.1f} s"}


class RingFeed:
    """The measured loop fed from files: sample k of a batch is data_in_<k>.npy / data_out_<k>.npy ([H, W, C] fp32, the
    layout of the CAM files, cam_numpy_singlefile_dataset.py:90-99), written once from the synthetic batches.  next()
    collects the batch whose reads were submitted one step ago and submits the following one: file -> pinned slot by the
    ring's reader threads, slot -> HBM on the ring's copy stream, all under the running step."""

    def __init__(self, batches, device_index, rank):
        import tempfile
        import numpy as np
        from bias_gan_amd.data import numpy_reader as nr
        self.dir = tempfile.mkdtemp(prefix=f"bgamd_ring_{rank}_")
        self.sets = []
        for b, (x, y) in enumerate(batches):
            names = []
            for k in range(x.shape[0]):
                fi, fo = (os.path.join(self.dir, f"data_{t}_{b}_{k}.npy") for t in ("in", "out"))
                np.save(fi, x[k].permute(1, 2, 0).contiguous().cpu().numpy())
                np.save(fo, y[k].permute(1, 2, 0).contiguous().cpu().numpy())
                names += [fi, fo]
            self.sets.append(names)
        self.n = batches[0][0].shape[0]
        self.slots = 4 * self.n + 2                       # two batches of (in, out) samples in flight
        self.threads = int(os.environ.get("BG_RING_THREADS", "8"))
        self.reader = nr.numpy_reader(False, device_index, ring_slots=self.slots)
        self.reader.num_intra_threads = self.threads
        self.reader.parse(self.sets[0][0])
        self.sample_bytes = os.path.getsize(self.sets[0][0])
        self.turn = 0
        self.bytes = 0
        import atexit
        atexit.register(self.close)      # ~1.8 GB of .npy files per rank at 1152x768x16, batch 8
        self._submit()

    def close(self):
        import shutil
        d, self.dir = self.dir, None
        if d:
            self.reader = None
            shutil.rmtree(d, ignore_errors=True)

    def _submit(self):
        for f in self.sets[self.turn % len(self.sets)]:
            self.reader.prefetch(f)
        self.turn += 1

    def next(self):
        shape = (self.n, *self.reader.shape)                  # [N, H, W, C]: samples land in the rows of a batch buffer
        dev = torch.device("cuda", torch.cuda.current_device())
        xb, yb = torch.empty(shape, device=dev), torch.empty(shape, device=dev)
        for k in range(self.n):                               # submit order: in_0, out_0, in_1, ...
            self.reader.get_prefetched(out=xb[k])
            self.reader.get_prefetched(out=yb[k])
        self._submit()
        self.bytes += 2 * self.n * self.sample_bytes
        # NCHW views of channels-last memory: the module boundary converts without a transpose (ops.ToInternal)
        return xb.permute(0, 3, 1, 2), yb.permute(0, 3, 1, 2)


def host_floor(c, n, dtype, device, mode):
    """Host time to enqueue one step, measured where the device cannot push back: the same nets, batch and launch
    sequence on 64x64 fields (the GPU work per launch is a few microseconds, the Python / ctypes / autograd work per
    launch is what it is at full size).  hipGraph replay of the no-grad generator forward is switched off so that the
    count of launches matches the full-size step."""
    import contextlib
    import torch.nn as nn
    from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg
    from bias_gan_amd.gpsro_train.train_gan import GANTrainer
    from bias_gan_amd.utils import losses
    from bias_gan_amd.utils import parsing_helpers as ph
    from bias_gan_amd import graphs
    old, graphs._MODE = graphs._MODE, "0"
    old_env = os.environ.get("BGAMD_STEP_GRAPH")
    os.environ["BGAMD_STEP_GRAPH"] = "0"          # the eager launch path is what is being timed
    try:
        with contextlib.redirect_stdout(sys.stderr):
            G = dxg.Generator(c, c, "Interpolate", "Uniform", 0, os=16, pretrained=False, normalizer=nn.BatchNorm2d,
                              compute_dtype=dtype).to(device)
            D = dxg.Discriminator(n_input=c, os=16, pretrained=False, normalizer=nn.BatchNorm2d, input_size=(64, 64),
                                  compute_dtype=dtype).to(device)
        G.train(), D.train()
        tr = GANTrainer(G, D, ph.get_optimizer(G.parameters(), "Adam", 1e-4, 1e-8, 1e-5),
                        ph.get_optimizer(D.parameters(), "Adam", 1e-4, 1e-8, 1e-5), losses.GANLoss(mode, n, device),
                        losses.L1Loss(), loss_type_gan=mode, loss_weight_gp=10.0)
        x, y = synthetic_batch(n, c, 64, 64, 1, device)
        for _ in range(3):
            tr.step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            tr.step(x, y)
        dt = (time.perf_counter() - t0) / 5
        torch.cuda.synchronize()
        return 1e3 * dt
    finally:
        graphs._MODE = old
        if old_env is None:
            os.environ.pop("BGAMD_STEP_GRAPH", None)
        else:
            os.environ["BGAMD_STEP_GRAPH"] = old_env


def _half(fam, pick, mfma_peak_tflops):
    """Sum of the per-entry-point records `pick` selects: launches, single-stream kernel time, algorithmic FLOP / bytes and
    the fraction of the bound that applies (MFMA peak for the GEMM half, 8 TB/s for the element-wise half)."""
    sel = [v for k, v in fam.items() if pick(k)]
    n, secs = sum(v[0] for v in sel), sum(v[1] for v in sel)
    flops, nbytes = sum(v[2] for v in sel), sum(v[3] for v in sel)
    out = {"launches": n, "total_ms": 1e3 * secs}
    if mfma_peak_tflops is not None:
        tf = flops / secs * 1e-12 if secs > 0 else 0.0
        out.update(alg_tflop=flops * 1e-12, tflops=tf, frac=tf / mfma_peak_tflops, peak=mfma_peak_tflops, unit="TFLOP/s")
    else:
        tb = nbytes / secs * 1e-12 if secs > 0 else 0.0
        out.update(alg_gb=nbytes * 1e-9, tbps=tb, frac=tb / 8.0, peak=8.0, unit="TB/s")
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Launched plainly (`python bench.py --gpus N`): this parent never touches the GPU; it starts the N ranks as
        # children of torch.distributed.run (one process per GPU over RCCL) and relays rank 0's JSON line and the exit
        # code -- the reference's launcher contract (comm/distributed.py:195-199 expects one process per device).
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("[bench] no launcher environment: starting " + " ".join(cmd), file=sys.stderr, flush=True)
        if os.environ.get("BGAMD_BENCH_LAUNCH_ECHO"):      # tests/test_host_cpu.py: show the child command, start nothing
            print(json.dumps(cmd))
            raise SystemExit(0)
        raise SystemExit(subprocess.run(cmd).returncode)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    if os.environ.get("BGAMD_REHEARSE_ONE_GPU"):   # rehearsal of the N > 1 path on a one-GPU box: every rank on
        local_rank = 0                              # device 0, gloo instead of RCCL (which refuses duplicate devices)
        os.environ["LOCAL_RANK"] = "0"
        os.environ.setdefault("BGAMD_DIST_BACKEND", "gloo")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import bias_gan_amd  # noqa: F401
    from bias_gan_amd import _lib as L
    from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg
    from bias_gan_amd.comm.distributed import comm as distcomm
    from bias_gan_amd.gpsro_train.train_gan import GANTrainer
    from bias_gan_amd.utils import losses
    from bias_gan_amd.utils import parsing_helpers as ph
    L.load()  # fail loudly if the HIP library is missing

    comm = distcomm(mode="dummy" if world == 1 else "torchrun")
    seed = 333 + 7 * rank                      # the reference's seed rule (train_gan.py:56)
    torch.manual_seed(seed)
    c, h, w, n = args.channels, args.height, args.width, args.batch
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "fp8": torch.float8_e4m3fn}[args.dtype]

    torch.manual_seed(333)                     # identical initial weights on every rank (DDP also broadcasts)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):   # the constructors print like the reference's; stdout is for the JSON line
        G = dxg.Generator(c, c, "Interpolate", "Uniform", 0, os=16, pretrained=False, normalizer=nn.BatchNorm2d,
                          compute_dtype=dtype).to(device)
        D = dxg.Discriminator(n_input=c, os=16, pretrained=False, normalizer=nn.BatchNorm2d, input_size=(h, w),
                              compute_dtype=dtype).to(device)
    torch.manual_seed(seed)
    G.train(), D.train()
    g_opt = ph.get_optimizer(G.parameters(), "Adam", 1e-4, 1e-8, 1e-5)
    d_opt = ph.get_optimizer(D.parameters(), "Adam", 1e-4, 1e-8, 1e-5)
    Gd, Dd = comm.DistributedModel(G), comm.DistributedModel(D)
    mode = "Wasserstein" if args.loss == "wgan-gp" else "ModifiedMinMax"
    crit = losses.GANLoss(mode, n, device)
    trainer = GANTrainer(Gd, Dd, g_opt, d_opt, crit, losses.L1Loss(), loss_type_gan=mode, loss_weight_gp=10.0)
    trainer.reuse_g_forward = bool(args.reuse_g_forward)
    if world > 1:   # what actually carries the gradients: the driver's torchrun line must show RCCL with N ranks
        devs = [None] * world
        dist.all_gather_object(devs, f"rank {rank}: cuda:{torch.cuda.current_device()} ({torch.cuda.get_device_name()})")
        if rank == 0:
            print(f"[bench] torch.distributed backend {dist.get_backend()}" + (" (= RCCL on ROCm)" if dist.get_backend() == "nccl" else " (rehearsal)")
                  + f", world size {dist.get_world_size()}: "
                  + "; ".join(devs), file=sys.stderr, flush=True)

    batches = [synthetic_batch(n, c, h, w, seed + 1000 * i, device) for i in range(2)]
    feed = None
    if args.data == "ring":
        feed = RingFeed(batches, local_rank, rank)
        batches = None

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"models built, {args.warmup} warm-up steps")
    nxt = (lambda i: feed.next()) if feed is not None else (lambda i: batches[i % 2])
    if args.dtype == "fp8":          # first exponents of the quantisation sites (state-neutral; un-timed)
        trainer.calibrate_fp8(*(batches[0] if batches is not None else feed.next()))
    for i in range(args.warmup):
        trainer.step(*nxt(i))
    sync_all()
    if feed is not None:
        feed.bytes = 0
    note(f"timing {args.steps} steps")
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step durations (no host sync)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        d_loss, g_loss = trainer.step(*nxt(i))
        marks[i + 1].record()
    host_enqueue = time.perf_counter() - t0      # host time to ENQUEUE the steps (no device sync in the loop)
    sync_all()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    note(f"host enqueue {1e3 * host_enqueue / args.steps:.1f} ms/step")
    note(f"done: {1e3 * elapsed / args.steps:.1f} ms/step")
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    d_loss, g_loss = float(d_loss), float(g_loss)
    assert d_loss == d_loss and g_loss == g_loss, "training diverged to NaN"

    value = args.steps * n * world / elapsed
    out = {
        "metric": "climate-field samples/sec (G+D step) at 1152x768x16" if (h, w, c) == (1152, 768, 16)
        else f"climate-field samples/sec (G+D step) at {h}x{w}x{c}",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_step_median": median_ms,   # median of per-step device times (HIP events)
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic" if feed is None else "synthetic, staged from .npy files through the ring every step",
        "config": {"workload": f"{h}x{w}x{c} synthetic fields, batch {n}/GPU, DeepLabv3+/Xception-65 generator "
                               "(Interpolate upsampler, noise_dimensions 0) + Xception-65/Linear discriminator, "
                               "BatchNorm, " + ("Wasserstein + gradient penalty (weight 10) + L1" if args.loss == "wgan-gp" else "ModifiedMinMax + L1")
                               + ", Adam(1e-4, eps 1e-8, wd 1e-5), D-step + G-step per step"
                               + (" [OPT-IN --reuse-g-forward: one generator forward per step]" if args.reuse_g_forward else ""),
                   "global_batch": n * world, "parallelism": f"dp{world}", "last_d_loss": d_loss, "last_g_loss": g_loss,
                   "collective_backend": (dist.get_backend() if world > 1 else None)},
    }

    if feed is not None:
        out["staging"] = {"file_to_hbm_GBps": feed.bytes / elapsed * 1e-9, "bytes_per_step": feed.bytes / args.steps,
                          "files_per_step": 2 * n, "reader_threads": feed.threads, "ring_slots": feed.slots,
                          "note": "files read from the page cache by the ring's threads into pinned slots, H2D on the ring's copy "
                                  "stream under the previous step; the step's stream waits on the slot events only"}
        batches = [feed.next(), feed.next()]
    # ---- roofline: one extra (un-timed) step with HIP events around every C-ABI launch
    # (every rank runs the step -- it contains the gradient all-reduces -- rank 0 records)
    # Two such steps, every launch keeps the shorter of its two measurements: a bracket also contains whatever the host
    # did between recording the first event and launching (an allocator trim, a page fault), which once put 25 ms on a
    # 1 ms kernel and moved the whole family's figure.
    prof = []
    if not args.no_kernel_profile:
        side, trainer._side = trainer._side, None   # one stream for this step: events then bracket ONE kernel each
        for _ in range(2):
            if rank == 0:
                L.PROFILE = []
                L.PROFILE_SHAPES = [] if args.dump_launches else None
            trainer.step(*batches[0])
            torch.cuda.synchronize()
            if rank == 0:
                prof.append([(nm, fl, e0.elapsed_time(e1), nb) for nm, fl, e0, e1, nb in L.PROFILE])
                shapes, L.PROFILE, L.PROFILE_SHAPES = L.PROFILE_SHAPES, None, None
        trainer._side = side
    if rank == 0 and not args.no_kernel_profile:
        a, b = prof
        if len(a) == len(b) and all(x[0] == y[0] for x, y in zip(a, b)):
            a = [(x[0], x[1], min(x[2], y[2]), x[3]) for x, y in zip(a, b)]

        class _Ev:   # (elapsed already taken)
            def __init__(self, ms): self.ms = ms
            def elapsed_time(self, other): return other.ms
        recs = [(nm, fl, _Ev(0.0), _Ev(ms), nb) for nm, fl, ms, nb in a]
        if args.dump_launches:
            with open(args.dump_launches, "w") as f:
                for i, (name, flops, e0, e1, nbytes) in enumerate(recs):
                    f.write(f"{name} {flops:.0f} {nbytes:.0f} {e0.elapsed_time(e1):.5f} {shapes[i] if shapes and len(shapes) == len(recs) else '-'}\n")
        fam = {}
        for name, flops, e0, e1, nbytes in recs:
            f = fam.setdefault(name, [0, 0.0, 0.0, 0.0])
            f[0] += 1
            f[1] += e0.elapsed_time(e1) * 1e-3
            f[2] += flops
            f[3] += nbytes
        # A GEMM launch is MFMA-bound when its arithmetic intensity (algorithmic FLOP per algorithmic byte, every
        # operand moved once) is above the machine's ridge, peak MFMA / peak HBM = 2.5e15 / 8e12 = 312 FLOP/B (bf16);
        # below it the same kernel is an HBM-bound copy with some arithmetic attached (the 128-channel 1x1 layers of the
        # entry flow: 64 FLOP/B) and belongs under the HBM roofline: those launches are listed as '<entry point>[hbm]'
        # with their algorithmic GB/s and are not part of the MFMA figure.
        def peak_of(name):   # the MFMA peak of the operand type an entry point computes on
            return PEAK_FP8_TFLOPS if name.endswith("_fp8") else PEAK_F32_TFLOPS if args.dtype == "f32" else PEAK_BF16_TFLOPS
        ridge = peak_of("") * 1e12 / 8e12
        fam = {}
        for name, flops, e0, e1, nbytes in recs:
            if name.startswith("bg_conv2d") and nbytes > 0 and flops / nbytes < peak_of(name) * 1e12 / 8e12:
                name += "[hbm]"
            f = fam.setdefault(name, [0, 0.0, 0.0, 0.0])
            f[0] += 1
            f[1] += e0.elapsed_time(e1) * 1e-3
            f[2] += flops
            f[3] += nbytes
        mfma = {k: v for k, v in fam.items() if k.startswith("bg_conv2d") and not k.endswith("[hbm]")}
        dom = max(mfma, key=lambda k: mfma[k][1])
        # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes of this same
        # command (profiles/*_pmc_hbm_traffic.json, FETCH_SIZE corrected per MI355X_MICROARCH.md);
        # PMC collection needs the profiler, so it is not re-measured inside this process.
        traffic = None
        try:
            import glob
            pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")))[-1]
            for fam_name, rec in json.load(open(pj))["kernels"].items():
                if "fat tiles" in fam_name and dom.replace("_stats", "") in fam_name.split(" ")[0].split("+"):
                    traffic = rec["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        cnt, secs, flops, _ = mfma[dom]
        achieved = flops / secs * 1e-12
        walg = w_alg_tflop(h, w, c, args.loss, 3 if args.reuse_g_forward else 4)   # the work actually required in that mode
        out["roofline"] = {
            "bound": "mfma", "kernel": dom, "ridge_flop_per_byte": peak_of(dom) * 1e12 / 8e12, "achieved": achieved, "peak": peak_of(dom),
            "unit": "TFLOP/s", "frac": achieved / peak_of(dom), "traffic": traffic,
            "traffic_source": None if traffic is None else "committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                              "command (profiles/*_pmc_hbm_traffic.json); not re-measured in this run",
            "launches": cnt, "avg_launch_ms": 1e3 * secs / cnt, "alg_gflop_per_launch": flops / cnt * 1e-9,
            "step_conv_stack_tflops": None if walg is None else value / world * walg,
            # whole conv stack against the bf16 peak (the north_star's figure; an fp8 run is quoted against it too -- its
            # weight gradients and the layers below the fp8 threshold compute in bf16)
            "step_conv_stack_frac": None if walg is None else value / world * walg / PEAK_BF16_TFLOPS,
            # per entry point: MFMA-bound ones in TFLOP/s, HBM-bound ones in algorithmic GB/s (peak 8000)
            # the two halves of the step, so that the driver-run line tells the whole story: every GEMM
            # entry point together (MFMA-bound and HBM-bound launches: all of the step's conv FLOPs over all of their time)
            # against the MFMA peak, and every other kernel together against the HBM peak
            "all_gemm": _half(fam, lambda k: k.startswith("bg_conv2d"), peak_of("")),
            "elementwise": _half(fam, lambda k: not k.startswith("bg_conv2d"), None),
            "families": {k: {"launches": v[0], "total_ms": 1e3 * v[1], "tflops": (v[2] / v[1] * 1e-12 if v[1] > 0 else 0.0),
                             "alg_gbps": (v[3] / v[1] * 1e-9 if v[1] > 0 else 0.0)}
                         for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])},
        }
    if rank == 0 and world == 1 and not args.no_host_floor:
        out["host_ms_per_step"] = host_floor(c, n, dtype, device, mode)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(c, h, w)
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
