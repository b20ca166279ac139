"""The GAN training step of the reference loop (gpsro_train/train_gan.py:244-298)
as a reusable object.

Results-neutral work of the reference is not executed: the
D-step does not back-propagate into the generator (its forward runs without a
graph) and the G-step computes no discriminator weight gradients.  Everything
that changes results is kept: both generator forwards and all discriminator
forwards run in train mode (BatchNorm running statistics get the same number of
momentum updates), losses and label draws follow the reference's order.
"""
from __future__ import annotations

import torch

from ..architecture.gpsro import deeplab_gan as dxg
from ..comm.distributed import DistributedModel
from ..runtime import StatsPool


def _unwrap(m):
    return m.module if isinstance(m, DistributedModel) else m


class GANTrainer:
    def __init__(self, generator, discriminator, g_opt, d_opt, criterion_gan, criterion_regression,
                 loss_type_gan="ModifiedMinMax", loss_weight_gan=1.0, loss_weight_regression=1.0, loss_weight_gp=10.0,
                 enable_masks=False, generator_warmup_steps=0, g_scheduler=None, d_scheduler=None,
                 update_frequency_generator=1, update_frequency_discriminator=1):
        self.generator, self.discriminator = generator, discriminator
        self.g_opt, self.d_opt = g_opt, d_opt
        self.criterion_gan, self.criterion_regression = criterion_gan, criterion_regression
        self.loss_type_gan = loss_type_gan
        self.w_gan, self.w_reg, self.w_gp = loss_weight_gan, loss_weight_regression, loss_weight_gp
        self.enable_masks = enable_masks
        self.warmup = generator_warmup_steps
        self.g_scheduler, self.d_scheduler = g_scheduler, d_scheduler
        self.freq_g, self.freq_d = update_frequency_generator, update_frequency_discriminator
        self.step_count = 0
        self._d_pending = False
        # the generator forward of the D-step has no data dependence on D(real): run them on two HIP
        # streams so the tails of one network's kernels overlap the other's (BGAMD_NO_SIDE_STREAM=1 disables)
        import os
        self._side = None if os.environ.get("BGAMD_NO_SIDE_STREAM") else "auto"
        self._d_params = [p for p in _unwrap(discriminator).parameters()]

    # -- train_gan.py:250-271 -------------------------------------------------------------
    def d_step(self, inputs, outputs_real, labels=None, eta=None):
        if self._side == "auto":
            self._side = torch.cuda.Stream(device=inputs.device) if inputs.is_cuda else None
        if self._side is not None:
            main = torch.cuda.current_stream(inputs.device)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side), torch.no_grad():
                outputs_fake = self.generator(inputs)
            logits_real, _ = self.discriminator(outputs_real)    # concurrently on the main stream
            main.wait_stream(self._side)
            outputs_fake.record_stream(main)
        else:
            with torch.no_grad():                   # no graph through G: D's update cannot use it
                outputs_fake = self.generator(inputs)
            logits_real, _ = self.discriminator(outputs_real)
        logits_fake, _ = self.discriminator(outputs_fake)
        if labels is not None:
            d_loss = self.criterion_gan.d_loss(logits_real, logits_fake, labels)
        else:
            d_loss = self.criterion_gan.d_loss(logits_real, logits_fake)
        if self.loss_type_gan == "Wasserstein":
            d_loss = d_loss + self.w_gp * dxg.gradient_penalty(_unwrap(self.discriminator), outputs_fake, outputs_real, eta)
        self.d_opt.zero_grad()
        d_loss.backward()
        if isinstance(self.discriminator, DistributedModel):
            self.discriminator.launch_grad_allreduce()   # overlaps with the G-step's generator forward
        self._d_pending = True                            # Adam for D is issued by _finish_d()
        return d_loss.detach()

    def _finish_d(self):
        if self._d_pending:
            self.d_opt.step()
            if self.d_scheduler is not None:
                self.d_scheduler.step()
            self._d_pending = False

    # -- train_gan.py:273-298 -------------------------------------------------------------
    def g_step(self, inputs, outputs_real, masks=None):
        for p in self._d_params:                    # D is only differentiated w.r.t. its input here
            p.requires_grad_(False)
        try:
            outputs_fake = self.generator(inputs)
            # D's Adam update (and its gradient all-reduce) must land before D is used again
            self._finish_d()
            logits_fake, _ = self.discriminator(outputs_fake)
            gan_loss = self.criterion_gan.g_loss(logits_fake)
            if self.enable_masks:
                regression_loss = self.criterion_regression(outputs_fake, outputs_real, masks)
            else:
                regression_loss = self.criterion_regression(outputs_fake, outputs_real)
            if self.step_count < self.warmup:
                g_loss = regression_loss
            else:
                g_loss = self.w_gan * gan_loss + self.w_reg * regression_loss
            self.g_opt.zero_grad()
            g_loss.backward()
        finally:
            for p in self._d_params:
                p.requires_grad_(True)
        if isinstance(self.generator, DistributedModel):
            self.generator.launch_grad_allreduce()
        self.g_opt.step()
        if self.g_scheduler is not None:
            self.g_scheduler.step()
        return g_loss.detach()

    def step(self, inputs, outputs_real, masks=None, labels=None, eta=None):
        """One loop iteration: D-step then G-step.  Returns device scalars (no host sync).

        D's optimiser step is issued inside g_step after the generator forward has
        been enqueued: the generator forward does not read D, so under data
        parallelism D's gradient all-reduce runs on the RCCL stream beneath it."""
        StatsPool.reset_all()   # one fill clears every statistic accumulator of the previous step
        s = self.step_count
        train_generator = (s < self.warmup) or (s % self.freq_g == 0)            # train_gan.py:247
        train_discriminator = (s >= self.warmup) and (s % self.freq_d == 0)      # train_gan.py:248
        d_loss = g_loss = None
        if train_discriminator:
            d_loss = self.d_step(inputs, outputs_real, labels, eta)
        if train_generator:
            g_loss = self.g_step(inputs, outputs_real, masks)
        self._finish_d()
        self.step_count += 1
        return d_loss, g_loss


    # -- train_gan.py:330-398 -----------------------------------------------------------
    @torch.no_grad()
    def validate(self, loader, comm=None):
        """Eval-mode pass over a validation loader: returns (d_loss, g_loss) averaged over
        all samples of all ranks (three summed all-reduces, like train_gan.py:377-386)."""
        G, D = self.generator, self.discriminator
        G.eval(), D.eval()
        dev = next(_unwrap(G).parameters()).device
        count = torch.zeros((), device=dev)
        d_sum = torch.zeros((), device=dev)
        g_sum = torch.zeros((), device=dev)
        try:
            for batch in loader:
                inputs, outputs_real = batch[0], batch[1]
                masks = batch[2] if self.enable_masks else None
                outputs_fake = G(inputs)
                logits_real, _ = D(outputs_real)
                logits_fake, _ = D(outputs_fake)
                d_loss = self.criterion_gan.d_loss(logits_real, logits_fake)
                gan_loss = self.criterion_gan.g_loss(logits_fake)
                if self.enable_masks:
                    reg = self.criterion_regression(outputs_fake, outputs_real, masks)
                else:
                    reg = self.criterion_regression(outputs_fake, outputs_real)
                n = float(inputs.shape[0])
                count += n
                d_sum += n * d_loss
                g_sum += n * (self.w_gan * gan_loss + self.w_reg * reg)
        finally:
            G.train(), D.train()
        if comm is not None and comm.size() > 1:
            import torch.distributed as dist
            for t in (count, d_sum, g_sum):
                dist.all_reduce(t)
        c = max(float(count), 1.0)
        return float(d_sum) / c, float(g_sum) / c

    # -- train_gan.py:401-431 / comm/distributed.py:130-157 -------------------------------
    def save_checkpoint(self, path, epoch=0):
        """{step, epoch, generator, discriminator, g_opt, d_opt, amp}: the reference's dictionary, with
        the reference's state_dict key names (DistributedModel adds the 'module.' prefix like DDP)."""
        torch.save({"step": self.step_count, "epoch": epoch,
                    "generator": {k: v.detach().clone().contiguous() for k, v in self.generator.state_dict().items()},
                    "discriminator": {k: v.detach().clone().contiguous()
                                      for k, v in self.discriminator.state_dict().items()},
                    "g_opt": self.g_opt.state_dict(), "d_opt": self.d_opt.state_dict(), "amp": None}, path)

    def load_checkpoint(self, path, comm, device):
        self.step_count, epoch = comm.init_gan_training_state(_unwrap(self.generator), _unwrap(self.discriminator),
                                                              self.g_opt, self.d_opt, path, device)
        return self.step_count, epoch
