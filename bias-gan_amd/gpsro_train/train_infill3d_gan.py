"""Training iteration of the partial-convolution GAN in 3-D (reference: gpsro_train/infill3d_gan_module.py:272-375)
on the HIP path.

What the reference's loop body does, restated here:
  * the update flags come from the step counter and LAST iteration's discriminator accuracy (:294-311): warm-up
    trains G alone; accuracy above disc_acc_max trains G alone, below disc_acc_min D alone, otherwise both;
  * both losses are evaluated every iteration, in train mode (two generator and three discriminator passes, so the
    BatchNorm statistics move five times whatever the flags say);
  * the noise channels join the input, all-ones channels join the mask (:279-291); the discriminator receives the
    1-channel volume with the (1 + noise)-channel mask, which `input * mask` broadcasts (partialconv3d.py:77);
  * d_loss = GANLoss.d_loss * loss_weights["adv"]; g_loss = sum_k loss_weights[k] * term_k over hole / valid / tv
    (+ adv once the warm-up is over).
Results-neutral waste is not reproduced: the D part does not back-propagate into the generator and the G part keeps
no discriminator weight gradients (the reference zeroes both before they are used)."""
from __future__ import annotations

import torch

from ..comm.distributed import DistributedModel
from ..graphs import HostStepState, NoGradGraph, roll_fp8_sites
from ..runtime import StatsPool


def _unwrap(m):
    return m.module if isinstance(m, DistributedModel) else m


class InfillGANTrainer:
    def __init__(self, generator, discriminator, g_opt, d_opt, criterion_gan, criterion_reconst, loss_weights,
                 gen_warmup_steps=0, disc_acc_min=0.0, disc_acc_max=1.0, g_scheduler=None, d_scheduler=None):
        self.generator, self.discriminator = generator, discriminator
        self.g_opt, self.d_opt = g_opt, d_opt
        self.criterion_gan, self.criterion_reconst = criterion_gan, criterion_reconst
        self.loss_weights = {k: float(v) for k, v in loss_weights.items()}
        self.warmup, self.acc_min, self.acc_max = gen_warmup_steps, disc_acc_min, disc_acc_max
        self.g_scheduler, self.d_scheduler = g_scheduler, d_scheduler
        self.step_count = 0
        self._d_acc = 0.5                          # infill3d_gan_module.py:268
        self._acc_pending = None                   # (pinned host scalar, event) of the accuracy still in flight
        self.last_flags = (True, True)
        self.last_terms = {}
        self._d_params = [p for p in _unwrap(discriminator).parameters()]
        self._g_nograd = NoGradGraph(generator)   # hipGraph replay of the D part's generator forward (graphs.py)

    @property
    def d_acc_avg(self):
        """The discriminator accuracy of the last iteration (the reference's `d_acc_avg`).  It is only needed when the
        NEXT iteration picks its update flags, so its read-back is asynchronous: a copy into pinned memory behind the
        D part, waited for here -- the host keeps enqueueing the G part instead of stalling in the middle of a step."""
        if self._acc_pending is not None:
            host, ev, scale = self._acc_pending
            ev.synchronize()
            self._d_acc = float(host) * scale
            self._acc_pending = None
        return self._d_acc

    def update_flags(self):
        """(train_generator, train_discriminator), infill3d_gan_module.py:294-311."""
        if self.step_count < self.warmup:
            return True, False
        if self.d_acc_avg > self.acc_max:
            return True, False
        if self.d_acc_avg < self.acc_min:
            return False, True
        return True, True

    def step(self, inputs_raw, outputs_real, masks_raw, noise, labels=None, comm=None):
        """inputs_raw, outputs_real, masks_raw: [N,1,D,H,W] fp32 on the device; noise [N,nd,D,H,W] (the reference draws
        it on the host and copies it over).  Returns (d_loss, g_loss) as device scalars.

        Where the iteration qualifies (_graph_ok) it is captured into ONE hipGraph per (shapes, update flags, warm-up)
        configuration on its third call and replayed afterwards: its ~580 launches against ~12 ms of kernels made it
        host-bound.  The update flags are picked on the host, from last iteration's accuracy, BEFORE the graph is chosen."""
        if self._graph_ok(inputs_raw, comm):
            return self._graph_step(inputs_raw, outputs_real, masks_raw, noise, labels)
        return self._eager_step(inputs_raw, outputs_real, masks_raw, noise, labels, comm)

    # -- whole-step hipGraph (the machinery of gpsro_train.train_gan.GANTrainer._graph_step) ---------------------------
    def _graph_ok(self, inputs_raw, comm):
        import os
        from .. import _lib as L
        if os.environ.get("BGAMD_STEP_GRAPH") == "0" or L.PROFILE is not None or not inputs_raw.is_cuda:
            return False
        if torch.cuda.is_current_stream_capturing() or (comm is not None and comm.size() > 1):
            return False
        if isinstance(self.generator, DistributedModel) or isinstance(self.discriminator, DistributedModel):
            return False
        if getattr(self.criterion_gan, "mode", None) != "ModifiedMinMax":
            return False
        return type(self.g_opt).__name__ == "FusedAdam" and type(self.d_opt).__name__ == "FusedAdam"

    def _graph_step(self, inputs_raw, outputs_real, masks_raw, noise, labels):
        train_g, train_d = self.update_flags()
        def _opt_key(o):
            return tuple((float(g["eps"]), float(g["weight_decay"]), tuple(float(b) for b in g["betas"]),
                          bool(g.get("amsgrad", False)), type(o).__name__ + str(g.get("decoupled", ""))) for g in o.param_groups)
        key = (tuple(inputs_raw.shape), tuple(noise.shape), inputs_raw.dtype, train_g, train_d, self.step_count < self.warmup,
               id(getattr(_unwrap(self.generator), "_bg_arena", None)), id(getattr(_unwrap(self.discriminator), "_bg_arena", None)),
               tuple(sorted(self.loss_weights.items())), id(self.criterion_gan), id(self.criterion_reconst),
               _opt_key(self.d_opt), _opt_key(self.g_opt))
        if not hasattr(self, "_graphs"):
            self._graphs, self._graph_seen, self._graph_failed = {}, {}, set()
        e = self._graphs.get(key)
        if key in self._graph_failed or (e is None and self._graph_seen.get(key, 0) < 2):   # two eager steps first
            self._graph_seen[key] = self._graph_seen.get(key, 0) + 1
            return self._eager_step(inputs_raw, outputs_real, masks_raw, noise, labels, None)
        # the host's random draws, in the eager order; a swap exchanges the two label tensors -- no branch in the graph
        lf, lr_, swap = labels if labels is not None else self.criterion_gan.draw_labels()
        lab = (lr_, lf) if swap else (lf, lr_)
        dev = inputs_raw.device
        srcs = (inputs_raw, outputs_real, masks_raw, noise)
        if e is None:
            e = {"in": tuple(t.clone() for t in srcs), "lab": tuple(t.to(dev).clone() for t in lab)}
        else:
            for dst, src in zip(e["in"], srcs):
                dst.copy_(src, non_blocking=True)
            for dst, src in zip(e["lab"], lab):
                dst.copy_(src if src.is_cuda else src.pin_memory(), non_blocking=True)
        opts = ([self.d_opt] if train_d else []) + ([self.g_opt] if train_g else [])
        pool = StatsPool.get(dev)
        if "graph" not in e:
            snap = HostStepState(self, (self.generator, self.discriminator), (self.d_opt, self.g_opt),
                                 (self.d_scheduler, self.g_scheduler),
                                 extra_attrs=("last_flags", "last_terms", "_d_acc", "_acc_pending"))
            for o in opts:
                o.prepare_replay()
            try:
                self._capture(e, key, opts, pool)
            except Exception as err:   # noqa: BLE001 -- whatever the capture raised, the eager path is the fallback
                snap.restore()         # step counts, LR schedules, BatchNorm counters, pool cursor: graphs.HostStepState
                self._graph_failed.add(key)
                self._graphs.pop(key, None)
                import warnings
                warnings.warn(f"whole-step hipGraph capture failed ({type(err).__name__}: {err}); this configuration stays eager")
                torch.cuda.synchronize()
                return self._eager_step(inputs_raw, outputs_real, masks_raw, noise, labels if labels is not None else (lf, lr_, swap), None)
        else:
            for o in opts:
                o.prepare_replay()       # step count, lr and bias corrections of THIS step -> device
            for net in (self.generator, self.discriminator):
                _unwrap(net).arena().sync()     # weights written outside the graph since the last step
            pool.used = e["pool_after"]
            for m_, k in e["nbt"]:
                m_.__dict__["_bg_nbt_pending"] = m_.__dict__.get("_bg_nbt_pending", 0) + k
            e["graph"].replay()
            if train_d and self.d_scheduler is not None:
                self.d_scheduler.step()
            if train_g and self.g_scheduler is not None:
                self.g_scheduler.step()
            self.step_count += 1
        self.last_flags = (train_g, train_d)
        self.last_terms = e["terms"]
        self._publish_accuracy(e["acc"])
        return e["d_loss"], e["g_loss"]

    def _capture(self, e, key, opts, pool):
        import torch.nn as nn
        bns = [m for net in (self.generator, self.discriminator) for m in _unwrap(net).modules()
               if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        before = [m.__dict__.get("_bg_nbt_pending", 0) for m in bns]
        for net in (self.generator, self.discriminator):       # packed weight copies current before the graph starts
            _unwrap(net).arena().sync()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        for o in opts:
            o.capturing = True
        self._capturing = True
        try:
            with torch.cuda.graph(g):
                d_loss, g_loss = self._eager_step(*e["in"], (e["lab"][0], e["lab"][1], False), None)
        finally:
            self._capturing = False
            for o in opts:
                o.capturing = False
        e.update(graph=g, d_loss=d_loss, g_loss=g_loss, acc=self._acc_dev, terms=self.last_terms, pool_after=pool.used,
                 nbt=[(m, m.__dict__.get("_bg_nbt_pending", 0) - b) for m, b in zip(bns, before) if m.__dict__.get("_bg_nbt_pending", 0) != b])
        self._graphs[key] = e
        g.replay()                      # capture records, it does not execute

    def _publish_accuracy(self, acc):
        """The accuracy steers the NEXT iteration: asynchronous read-back into pinned memory (see d_acc_avg)."""
        if acc.is_cuda:
            host = torch.empty((), dtype=torch.float32, pin_memory=True)
            host.copy_(acc, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._acc_pending = (host, ev, 1.0)
        else:
            self._d_acc = float(acc)

    def _eager_step(self, inputs_raw, outputs_real, masks_raw, noise, labels=None, comm=None):
        StatsPool.reset_all()
        inputs = torch.cat((inputs_raw, noise), dim=1)
        masks = torch.cat((masks_raw, torch.ones_like(noise)), dim=1)
        train_g, train_d = self.last_flags = self.update_flags()
        # ---- discriminator part (:314-341)
        outputs_fake, _ = self._g_nograd(inputs, masks)
        with torch.set_grad_enabled(train_d):
            logits_real, _ = self.discriminator(outputs_real, masks)
            logits_fake, _ = self.discriminator(outputs_fake, masks)
            d_loss = self.criterion_gan.d_loss(logits_real, logits_fake, labels) * self.loss_weights["adv"]
        with torch.no_grad():   # utils/metrics.py:18-32 against labels 1 / 0: sigmoid(logit) > 0.5  <=>  logit > 0
            acc = 0.5 * ((logits_real > 0).float().mean() + (logits_fake <= 0).float().mean())
        # the accuracy steers the NEXT iteration (a host value in the reference: metric_average's SUM over ranks)
        if comm is not None and comm.size() > 1:
            import torch.distributed as dist
            acc = acc.clone()
            dist.all_reduce(acc)
        if getattr(self, "_capturing", False):
            self._acc_dev = acc          # read back after every replay (_graph_step)
        else:
            self._publish_accuracy(acc)
        if train_d:
            self.d_opt.zero_grad()
            d_loss.backward()
            if isinstance(self.discriminator, DistributedModel):
                self.discriminator.launch_grad_allreduce()
            self.d_opt.step()
            if self.d_scheduler is not None:
                self.d_scheduler.step()
        # ---- generator part (:344-373)
        for p in self._d_params:
            p.requires_grad_(False)
        try:
            with torch.set_grad_enabled(train_g):
                outputs_fake, _ = self.generator(inputs, masks)
                logits_fake, _ = self.discriminator(outputs_fake, masks)
                terms = self.criterion_reconst(inputs_raw, outputs_fake, outputs_real, masks_raw)
                if self.step_count >= self.warmup:
                    terms["adv"] = self.criterion_gan.g_loss(logits_fake)
                g_loss = 0.
                for key in terms:
                    g_loss = g_loss + terms[key] * self.loss_weights[key]
            if train_g:
                self.g_opt.zero_grad()
                g_loss.backward()
                if isinstance(self.generator, DistributedModel):
                    self.generator.launch_grad_allreduce()
                self.g_opt.step()
                if self.g_scheduler is not None:
                    self.g_scheduler.step()
        finally:
            for p in self._d_params:
                p.requires_grad_(True)
        self.last_terms = {k: v.detach() for k, v in terms.items()}
        roll_fp8_sites(self.generator, self.discriminator)
        self.step_count += 1
        return d_loss.detach(), g_loss.detach()


class Infill3dGAN:
    """The reference's training module (gpsro_train/infill3d_gan_module.py:40-245): built from a config dict with the
    keys of gpsro_configs/infill3d_gan_1.yaml, `.train()` runs the loop.  wandb / plotting / the validation harness are
    not part of this build; `synthetic_size: [D, H, W]` in the config trains on synthetic volumes instead of
    `root_dir`'s GPSRODataset (masks = True)."""

    def __init__(self, config, comm=None):
        import os

        import torch.nn as nn

        from ..architecture.gpsro import infill3d_gan as dxi
        from ..comm.distributed import comm as distcomm
        from ..utils import losses
        from ..utils import parsing_helpers as ph

        self.config = dict(config)
        c = self.config
        self.comm = comm or distcomm(mode="dummy" if "RANK" not in os.environ else "torchrun")   # :48: mode="dummy"
        self.device = torch.device("cuda", self.comm.local_rank())
        torch.cuda.set_device(self.device)
        torch.manual_seed(333)

        def norm_of(key):                                             # :96-112
            kind = c.get(key, "batch_norm")
            if kind == "batch_norm":
                return nn.BatchNorm3d
            if kind == "instance_norm":
                return nn.InstanceNorm3d
            raise NotImplementedError("Error, " + str(kind) + " not supported")

        if c.get("noise_dimensions", 1) <= 0:                         # :147-148
            raise NotImplementedError("Error, please use at least one noise dimension.")
        self.noise_type = c.get("noise_type", "Normal")
        if self.noise_type not in ("Uniform", "Normal"):
            raise NotImplementedError("Error, noise type {} not supported.".format(self.noise_type))
        cdt = torch.bfloat16 if c.get("enable_amp", True) else torch.float32     # autocast -> bf16 storage, fp32 accumulate
        n_in = 1 + c["noise_dimensions"]
        net = dxi.GAN(input_channels=n_in, output_channels=1, gen_normalizer=norm_of("gen_layer_normalization"),
                      disc_normalizer=norm_of("disc_layer_normalization"), gen_layer_size=6, disc_layer_size=6)   # :115-119
        self.generator = net.generator.set_compute_dtype(cdt).to(self.device)
        self.discriminator = net.discriminator.set_compute_dtype(cdt).to(self.device)
        lw = c.get("loss_weights", {"valid": 1., "hole": 0.8, "tv": 0.0, "adv": 0.5})
        if "loss_weights.valid" in c:                                 # the flattened sweep form (:131-132)
            lw = {k.split(".")[1]: float(v) for k, v in c.items() if k.startswith("loss_weights")}
        self.loss_weights = {k: float(v) for k, v in lw.items()}
        g_opt = ph.get_optimizer(self.generator.parameters(), c.get("gen_optimizer", "AdamW"), c.get("gen_start_lr", 1e-3),
                                 c.get("gen_adam_eps", 1e-8), c.get("gen_weight_decay", 0.01))
        d_opt = ph.get_optimizer(self.discriminator.parameters(), c.get("disc_optimizer", "AdamW"), c.get("disc_start_lr", 1e-3),
                                 c.get("disc_adam_eps", 1e-8), c.get("disc_weight_decay", 0.01))
        g_sched = ph.get_lr_schedule(c["gen_start_lr"], c["gen_lr_schedule"], g_opt) if c.get("gen_lr_schedule") else None
        d_sched = ph.get_lr_schedule(c["disc_start_lr"], c["disc_lr_schedule"], d_opt) if c.get("disc_lr_schedule") else None
        self.generator.train(), self.discriminator.train()
        self.trainer = InfillGANTrainer(
            self.comm.DistributedModel(self.generator), self.comm.DistributedModel(self.discriminator), g_opt, d_opt,
            losses.GANLoss("ModifiedMinMax", c["local_batch_size"], self.device),
            dxi.InpaintingLoss(loss_type=c.get("loss_type", "smooth-l1")), self.loss_weights,
            gen_warmup_steps=c.get("gen_warmup_steps", 0), disc_acc_min=c.get("disc_acc_min", 0.0),
            disc_acc_max=c.get("disc_acc_max", 1.0), g_scheduler=g_sched, d_scheduler=d_sched)

    def _batches(self):
        import os

        from torch.utils.data import DataLoader

        from ..data import gpsro_dataset as gpsro
        c, dev = self.config, self.device
        n = c["local_batch_size"]
        if c.get("synthetic_size"):
            d, h, w = c["synthetic_size"]
            g = torch.Generator(device=dev).manual_seed(333 + 7 * self.comm.rank())
            while True:
                gt = torch.randn((n, d, h, w), generator=g, device=dev)
                m = (torch.rand((n, d, h, w), generator=g, device=dev) > 0.3).float()
                yield gt * m, gt, m, None
        root = os.path.join(c["root_dir"], "train")
        ds = gpsro.GPSRODataset(root, statsfile=os.path.join(c["root_dir"], "stats3d.npz"), channels=c.get("channels", list(range(45))),
                                normalization_type="MinMax" if self.noise_type == "Uniform" else "MeanVariance", shuffle=True,
                                masks=True, shard_idx=self.comm.rank(), shard_num=self.comm.size(),
                                num_intra_threads=c.get("max_intra_threads", 1), read_device=dev, send_device=dev)
        while True:
            yield from DataLoader(ds, n, drop_last=True)

    def train(self):
        import datetime as dt
        c, tr = self.config, self.trainer
        self.comm.printr('{:14.4f} REPORT: starting training'.format(dt.datetime.now().timestamp()), 0)
        for inputs_raw, outputs_real, masks_raw, _ in self._batches():
            inputs_raw, outputs_real, masks_raw = (t.unsqueeze(1) for t in (inputs_raw, outputs_real, masks_raw))   # :273-276
            shape = (inputs_raw.shape[0], c["noise_dimensions"]) + tuple(inputs_raw.shape[2:])
            noise = (torch.rand(shape) if self.noise_type == "Uniform" else torch.randn(shape)).to(self.device)      # host draw (:279-283)
            d_loss, g_loss = tr.step(inputs_raw, outputs_real, masks_raw, noise, comm=self.comm)
            if tr.step_count % c.get("logging_frequency", 50) == 0 or tr.step_count >= c.get("max_steps", 100):
                self.comm.printr('{:14.4f} REPORT training: step {} d_loss {} g_loss {} flags {}'.format(
                    dt.datetime.now().timestamp(), tr.step_count, float(d_loss), float(g_loss), tr.last_flags), 0)
            if tr.step_count >= c.get("max_steps", 100):
                break
        return tr


def main(pargs):
    import yaml
    if pargs.config_file is not None:
        with open(pargs.config_file) as f:
            doc = yaml.safe_load(f)
        config = dict(doc.get("default", doc))                        # utils/yparams.py: the "default" section
        config.setdefault("channels", list(range(0, 45)))
        config["checkpoint"] = pargs.checkpoint
    else:
        raise SystemExit("--config_file is required (the reference's wandb-sweep default is not part of this build)")
    if pargs.run_tag is not None:
        config["run_tag"] = pargs.run_tag
    for kv in pargs.set or []:                                        # small overrides: --set max_steps=4 synthetic_size=[16,16,16]
        k, v = kv.split("=", 1)
        config[k] = yaml.safe_load(v)
    return Infill3dGAN(config).train()


def build_parser():
    import argparse as ap
    AP = ap.ArgumentParser()
    AP.add_argument("--checkpoint", type=str, default=None, help="Checkpoint file to restart training from.")
    AP.add_argument("--config_file", type=str, default=None, help="YAML file to read config data from")
    AP.add_argument("--run_tag", type=str, default=None, help="A tag to identify the run")
    AP.add_argument("--set", type=str, nargs="*", default=None, help="key=value overrides of the config (YAML values)")
    return AP


if __name__ == "__main__":
    main(build_parser().parse_known_args()[0])
