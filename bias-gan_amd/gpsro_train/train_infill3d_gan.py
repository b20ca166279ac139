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
from ..graphs import NoGradGraph
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
        it on the host and copies it over).  Returns (d_loss, g_loss) as device scalars."""
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
        if acc.is_cuda:
            host = torch.empty((), dtype=torch.float32, pin_memory=True)
            host.copy_(acc, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._acc_pending = (host, ev, 1.0)
        else:
            self._d_acc = float(acc)
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
        self.step_count += 1
        return d_loss.detach(), g_loss.detach()
