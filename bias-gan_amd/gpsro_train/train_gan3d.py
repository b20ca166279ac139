"""Training loop of the 3-D GAN (reference: gpsro_train/train_gan3d.py:250-360) on the 3-D HIP path.

Differences from the 2-D loop that this class restates: both losses are evaluated EVERY iteration (forward passes
in train mode, so BatchNorm statistics move) while the updates follow a schedule -- static frequencies, or
"adaptive" on the discriminator's accuracy (train_gan3d.py:270-293); d_loss is multiplied by loss_weight_gan
(:306); the gradient penalty norms the input gradient per sample (deeplab3d_gan.py:103-125)."""
from __future__ import annotations

from ..architecture.gpsro import deeplab3d_gan as dxg3
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

    def step(self, inputs, outputs_real, masks=None, labels=None, eta=None, comm=None):
        StatsPool.reset_all()
        self._train_g, self._train_d = self.update_flags()
        self._want_g_ahead = False
        d_loss = self.d_step(inputs, outputs_real, labels, eta)
        if self.loss_type_gan == "ModifiedMinMax":
            # the accuracy steers the NEXT iteration's schedule: a host value, like the reference's metric_average
            acc = self.last_d_acc
            self.d_acc_avg = comm.metric_average(acc, "train_accuracy_discriminator", device=acc.device) if comm is not None \
                else float(acc)
        g_loss = self.g_step(inputs, outputs_real, masks)
        self._finish_d()
        self.step_count += 1
        return d_loss, g_loss
