"""Training loop of the 3-D GAN (reference: gpsro_train/train_gan3d.py:250-360) on the 3-D HIP path.

Differences from the 2-D loop that this class restates: both losses are evaluated EVERY iteration (forward passes
in train mode, so BatchNorm statistics move) while the updates follow a schedule -- static frequencies, or
"adaptive" on the discriminator's accuracy (train_gan3d.py:270-293); d_loss is multiplied by loss_weight_gan
(:306); the gradient penalty norms the input gradient per sample (deeplab3d_gan.py:103-125)."""
from __future__ import annotations

from ..architecture.gpsro import deeplab3d_gan as dxg3
from ..graphs import roll_fp8_sites
from ..runtime import StatsPool
from .train_gan import GANTrainer


class GANTrainer3d(GANTrainer):
    def __init__(self, *args, relative_update_schedule=None, **kw):
        """relative_update_schedule: {"type": "static", "update_frequency_generator": g, "update_frequency_discriminator": d}
        or {"type": "adaptive", "acc_min": a, "acc_max": b} (train_gan3d.py:592; the reference compares the dict
        itself with the strings "static"/"adaptive" -- a latent bug -- the intended key is restated as "type")."""
        kw.setdefault("gradient_penalty_fn", dxg3.gradient_penalty)
        super().__init__(*args, **kw)
        self.schedule = dict(relative_update_schedule or {"type": "static", "update_frequency_generator": 1,
                                                          "update_frequency_discriminator": 1})
        self.d_loss_scale = self.w_gan
        self.d_acc_avg = 0.0
        self._batched_d = False     # the 3-D layers' pixel counts rarely split into whole 128-pixel tiles per half
        self._always_eval = True    # both losses every iteration; the schedule's flags gate the two updates

    def update_flags(self):
        s, sch = self.step_count, self.schedule
        kind = sch.get("type", "static")
        if kind == "static":
            fg, fd = int(sch.get("update_frequency_generator", 1)), int(sch.get("update_frequency_discriminator", 1))
            return (s < self.warmup) or (s % fg == 0), (s >= self.warmup) and (s % fd == 0)
        if kind == "adaptive" and self.loss_type_gan != "Wasserstein":
            if s < self.warmup:
                return True, False
            if self.d_acc_avg > float(sch["acc_max"]):     # discriminator too good
                return True, False
            if self.d_acc_avg < float(sch["acc_min"]):     # discriminator too bad
                return False, True
            return True, True
        return True, True

    def _step_flags(self):
        return self.update_flags()

    def _eager_step(self, inputs, outputs_real, masks=None, labels=None, eta=None):
        StatsPool.reset_all()
        self._train_g, self._train_d = self.update_flags()
        self._want_g_ahead = False
        d_loss = self.d_step(inputs, outputs_real, labels, eta)
        g_loss = self.g_step(inputs, outputs_real, masks)
        self._finish_d()
        roll_fp8_sites(self.generator, self.discriminator)
        self.step_count += 1
        return d_loss, g_loss

    def step(self, inputs, outputs_real, masks=None, labels=None, eta=None, comm=None):
        """One iteration of train_gan3d.py:250-360.  Where the step qualifies it is one captured hipGraph per flag
        combination (GANTrainer._graph_step): at the 45x19x37 GPS-RO grid the ~2 700 launches of 10-30 us are bound by the
        host, not by the GPU."""
        self._train_g, self._train_d = self.update_flags()
        if self._graph_ok(inputs, outputs_real, masks):
            d_loss, g_loss = self._graph_step(inputs, outputs_real, masks, labels, eta)
        else:
            d_loss, g_loss = self._eager_step(inputs, outputs_real, masks, labels, eta)
        if self.loss_type_gan == "ModifiedMinMax":
            # the accuracy steers the NEXT iteration's schedule: a host value, like the reference's metric_average (read
            # after the iteration has been enqueued: it is only needed by the next update_flags())
            acc = self.last_d_acc
            self.d_acc_avg = comm.metric_average(acc, "train_accuracy_discriminator", device=acc.device) if comm is not None \
                else float(acc)
        return d_loss, g_loss


def main(pargs):
    """Command line of the reference's train_gan3d.py (:90-360) on the 3-D HIP path: one-channel volumes whose depth
    axis is the selected level range (`inputs.unsqueeze(1)`, :267-268), BatchNorm3d generator, InstanceNorm3d critic
    under the Wasserstein loss (:140), the static / adaptive relative update schedule."""
    import datetime as dt
    import os

    import torch
    import torch.nn as nn
    from torch.utils.data import DataLoader

    from ..architecture.gpsro import deeplab as dxc
    from ..comm.distributed import comm as distcomm
    from ..data import gpsro_dataset as gpsro
    from ..utils import losses
    from ..utils import parsing_helpers as ph

    comm = distcomm(mode="dummy" if "RANK" not in os.environ else "torchrun")
    seed = 333 + 7 * comm.rank()
    torch.manual_seed(seed)
    device = torch.device("cuda", comm.local_rank())
    torch.cuda.set_device(device)
    if pargs.synthetic_size is not None:
        d, h, w = pargs.synthetic_size
        g = torch.Generator(device=device).manual_seed(seed)

        def batches():
            while True:
                x = torch.randn((pargs.local_batch_size, d, h, w), generator=g, device=device)
                m = (torch.rand(x.shape, generator=g, device=device) > 0.1).float() if pargs.enable_masks else None
                yield x, x + 0.1 * torch.randn(x.shape, generator=g, device=device), m, None
        train_loader = batches()
    else:
        root = pargs.data_dir_prefix
        train_set = gpsro.GPSRODataset(os.path.join(root, "train"), statsfile=os.path.join(root, "stats3d.npz"),
                                       channels=pargs.channels,
                                       normalization_type="MinMax" if pargs.noise_type == "Uniform" else "MeanVariance",
                                       shuffle=True, masks=pargs.enable_masks, shard_idx=comm.rank(), shard_num=comm.size(),
                                       num_intra_threads=pargs.max_intra_threads, read_device=device, send_device=device)
        train_loader = DataLoader(train_set, pargs.local_batch_size, drop_last=True)

    cdt = torch.float32 if pargs.amp_opt_level == "O0" else torch.bfloat16
    g_norm = dxc.Identity if pargs.disable_batchnorm else nn.BatchNorm3d
    d_norm = dxc.Identity if pargs.disable_batchnorm else \
        (nn.InstanceNorm3d if pargs.loss_type_gan == "Wasserstein" else nn.BatchNorm3d)
    generator = dxg3.Generator(1, 1, pargs.upsampler_type, pargs.noise_type, pargs.noise_dimensions, os=16, pretrained=False,
                               normalizer=g_norm, compute_dtype=cdt).to(device)
    discriminator = dxg3.Discriminator(n_input=1, os=16, pretrained=False, normalizer=d_norm, compute_dtype=cdt).to(device)
    criterion_gan = losses.GANLoss(pargs.loss_type_gan, pargs.local_batch_size, device)
    if pargs.loss_type_regression == "l1":
        criterion_regression = losses.L1LossWeighted() if pargs.enable_masks else losses.L1Loss()
    elif pargs.loss_type_regression == "smooth_l1":
        criterion_regression = losses.SmoothL1Loss()
    elif pargs.loss_type_regression == "l2":
        criterion_regression = losses.MSELoss()
    else:
        raise NotImplementedError("Error, loss {} not implemented.".format(pargs.loss_type_regression))
    g_opt = ph.get_optimizer(generator.parameters(), pargs.optimizer_generator, pargs.start_lr_generator, pargs.adam_eps,
                             pargs.weight_decay)
    d_opt = ph.get_optimizer(discriminator.parameters(), pargs.optimizer_discriminator, pargs.start_lr_discriminator,
                             pargs.adam_eps, pargs.weight_decay)
    generator.train(), discriminator.train()
    g_sched = ph.get_lr_schedule(pargs.start_lr_generator, pargs.lr_schedule_generator, g_opt) \
        if pargs.lr_schedule_generator else None
    d_sched = ph.get_lr_schedule(pargs.start_lr_discriminator, pargs.lr_schedule_discriminator, d_opt) \
        if pargs.lr_schedule_discriminator else None
    trainer = GANTrainer3d(comm.DistributedModel(generator), comm.DistributedModel(discriminator), g_opt, d_opt, criterion_gan,
                           criterion_regression, loss_type_gan=pargs.loss_type_gan, loss_weight_gan=pargs.loss_weight_gan,
                           loss_weight_regression=pargs.loss_weight_regression, loss_weight_gp=pargs.loss_weight_gp,
                           enable_masks=pargs.enable_masks, generator_warmup_steps=pargs.generator_warmup_steps,
                           g_scheduler=g_sched, d_scheduler=d_sched,
                           relative_update_schedule=pargs.relative_update_schedule or None)
    if pargs.checkpoint:
        trainer.load_checkpoint(pargs.checkpoint, comm, device)
    comm.printr('{:14.4f} REPORT: starting training'.format(dt.datetime.now().timestamp()), 0)
    epoch, d_avg, g_avg = 0, 0., 0.
    while trainer.step_count < pargs.max_steps:
        for batch in train_loader:
            inputs, outputs_real = batch[0].unsqueeze(1), batch[1].unsqueeze(1)
            masks = batch[2].unsqueeze(1) if pargs.enable_masks else None
            d_loss, g_loss = trainer.step(inputs, outputs_real, masks, comm=comm)
            if trainer.step_count % pargs.logging_frequency == 0 or trainer.step_count == pargs.max_steps:
                d_avg = comm.metric_average(d_loss, "train_loss_discriminator", device=device)
                g_avg = comm.metric_average(g_loss, "train_loss_generator", device=device)
                comm.printr('{:14.4f} REPORT training: step {} d_loss {} g_loss {} d_acc {}'.format(
                    dt.datetime.now().timestamp(), trainer.step_count, d_avg, g_avg, trainer.d_acc_avg), 0)
            if pargs.save_frequency > 0 and trainer.step_count % pargs.save_frequency == 0 and comm.rank() == 0:
                trainer.save_checkpoint(os.path.join(pargs.output_dir, "gan3d_step_{}.cpt".format(trainer.step_count)), epoch)
            if trainer.step_count >= pargs.max_steps:
                break
        epoch += 1
    return trainer


def build_parser():
    from .train_gan import build_parser as base
    import argparse as ap

    class StoreDictKeyPair(ap.Action):
        def __call__(self, parser, namespace, values, option_string=None):
            setattr(namespace, self.dest, dict(kv.split("=") for kv in values.split(",")))

    AP = base()
    for a in list(AP._actions):                       # the 3-D script's differences from train_gan.py (:575,592,596)
        if a.dest in ("synthetic_size", "update_frequency_generator", "update_frequency_discriminator"):
            AP._remove_action(a)
            for s in a.option_strings:
                AP._option_string_actions.pop(s, None)
    AP.set_defaults(upsampler_type="Deconv", amp_opt_level="O0")
    AP.add_argument("--relative_update_schedule", action=StoreDictKeyPair, default={},
                    help="type=static,update_frequency_generator=1,update_frequency_discriminator=1 | "
                         "type=adaptive,acc_min=0.2,acc_max=0.8")
    AP.add_argument("--synthetic_size", type=int, nargs=3, default=None, metavar=("D", "H", "W"),
                    help="train on synthetic N(0,1) volumes of this size instead of a dataset")
    return AP


if __name__ == "__main__":
    main(build_parser().parse_args())
