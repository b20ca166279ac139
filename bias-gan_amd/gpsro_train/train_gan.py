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
from .. import ops
from .._lib import side_stream as _side_stream
from ..graphs import HostStepState, NoGradGraph, roll_fp8_sites
from ..runtime import StatsPool


def _unwrap(m):
    return m.module if isinstance(m, DistributedModel) else m


class GANTrainer:
    def __init__(self, generator, discriminator, g_opt, d_opt, criterion_gan, criterion_regression,
                 loss_type_gan="ModifiedMinMax", loss_weight_gan=1.0, loss_weight_regression=1.0, loss_weight_gp=10.0,
                 enable_masks=False, generator_warmup_steps=0, g_scheduler=None, d_scheduler=None,
                 update_frequency_generator=1, update_frequency_discriminator=1, gradient_penalty_fn=None):
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
        self._gp = gradient_penalty_fn or dxg.gradient_penalty
        self._train_d = self._train_g = True   # False: evaluate the losses only (train_gan3d.py's update schedule)
        self._always_eval = False              # True (3-D loop): both losses are evaluated every iteration, the flags gate the updates
        self.d_loss_scale = 1.0                # train_gan3d.py:306 multiplies d_loss by loss_weight_gan
        self.last_d_acc = None                 # 0.5 * (acc(real) + acc(fake)) of the last D-step (device scalar)
        # the generator forward of the D-step has no data dependence on D(real): run them on two HIP
        # streams so the tails of one network's kernels overlap the other's (BGAMD_NO_SIDE_STREAM=1 disables)
        import os
        self._side = None if os.environ.get("BGAMD_NO_SIDE_STREAM") else "auto"
        # D(real) and D(fake) of the D-step as ONE pass over the concatenated batch with per-half BatchNorm
        # statistics (ops.batch_groups): same results, kernels twice as large (BGAMD_NO_BATCHED_D=1 disables)
        self._batched_d = not os.environ.get("BGAMD_NO_BATCHED_D")
        # the G-step's generator forward reads nothing the D-step writes: it is issued on the side stream as
        # soon as the D-step's own (graph-free) generator forward is enqueued and runs under D's forward and
        # backward.  Only without host-drawn noise (the reference's RNG draw order is noise, labels, noise).
        self._g_ahead_ok = not os.environ.get("BGAMD_NO_G_PREFETCH")
        # OPT-IN (off by default; not the measured headline): the reference evaluates the generator twice per iteration on
        # the same input with unchanged weights (train_gan.py:252 and :275 -- G is not updated in between), so without
        # host-drawn noise the second forward reproduces the first.  With reuse_g_forward the D-step's forward is run
        # WITH its autograd graph and serves the G-step too; BatchNorm running statistics receive the two momentum
        # updates in closed form (ops.bn_repeat) and num_batches_tracked counts two.  Saves one generator forward per step.
        self.reuse_g_forward = bool(os.environ.get("BGAMD_REUSE_G_FORWARD"))
        self._g_ahead = None
        self._want_g_ahead = False
        self._d_params = [p for p in _unwrap(discriminator).parameters()]
        self._g_nograd = NoGradGraph(generator)  # the D-step's generator forward: hipGraph replay where the host is the limiter

    # -- train_gan.py:250-271 -------------------------------------------------------------
    def d_step(self, inputs, outputs_real, labels=None, eta=None):
        if not self._train_d:                       # losses / accuracy only: no graph
            outputs_fake = self._g_nograd(inputs)
            with torch.no_grad():
                logits_real, _ = self.discriminator(outputs_real)
                logits_fake, _ = self.discriminator(outputs_fake)
            return self._d_update(logits_real, logits_fake, outputs_fake, outputs_real, labels, eta)
        if self._batched_d and inputs.is_cuda:
            n = outputs_real.shape[0]
            if self.reuse_g_forward and self._want_g_ahead:
                with ops.bn_repeat(2):
                    fake_g = self.generator(inputs)          # ONE forward, with graph: this step's only one
                self._g_ahead = (inputs, fake_g, False)
                outputs_fake = fake_g.detach()
                note = getattr(fake_g, "_bg_internal", None)      # detach() is a new tensor object over the same bytes and version:
                if note is not None:                               # carry the hand-over note (ops._internal_of) along
                    outputs_fake._bg_internal = note
                with ops.batch_groups(2):
                    logits, _ = self.discriminator((outputs_real, outputs_fake))
                return self._d_update(logits[:n], logits[n:], outputs_fake, outputs_real, labels, eta)
            outputs_fake = self._g_nograd(inputs)   # no autograd graph through G: D's update cannot use it
            if self._want_g_ahead:
                if self._side == "auto":
                    self._side = _side_stream(inputs.device, "generator-ahead")
                if self._side is not None:
                    self._side.wait_stream(torch.cuda.current_stream(inputs.device))
                    with torch.cuda.stream(self._side):
                        self._g_ahead = (inputs, self.generator(inputs), True)
            with ops.batch_groups(2):               # group 0 = real, group 1 = fake: the reference's call order
                logits, _ = self.discriminator((outputs_real, outputs_fake))
            return self._d_update(logits[:n], logits[n:], outputs_fake, outputs_real, labels, eta)
        if self._side == "auto":
            self._side = _side_stream(inputs.device, "generator-ahead") if inputs.is_cuda else None
        if self._side is not None:
            main = torch.cuda.current_stream(inputs.device)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side), torch.no_grad():
                outputs_fake = self.generator(inputs)
            logits_real, _ = self.discriminator(outputs_real)    # concurrently on the main stream
            main.wait_stream(self._side)
            outputs_fake.record_stream(main)
        else:
            outputs_fake = self._g_nograd(inputs)   # no autograd graph through G: D's update cannot use it
            logits_real, _ = self.discriminator(outputs_real)
        logits_fake, _ = self.discriminator(outputs_fake)
        return self._d_update(logits_real, logits_fake, outputs_fake, outputs_real, labels, eta)

    def _d_update(self, logits_real, logits_fake, outputs_fake, outputs_real, labels, eta):
        if labels is not None:
            d_loss = self.criterion_gan.d_loss(logits_real, logits_fake, labels)
        else:
            d_loss = self.criterion_gan.d_loss(logits_real, logits_fake)
        if self.loss_type_gan == "Wasserstein":
            d_loss = d_loss + self.w_gp * self._gp(_unwrap(self.discriminator), outputs_fake, outputs_real, eta)
        if self.d_loss_scale != 1.0:
            d_loss = d_loss * self.d_loss_scale
        with torch.no_grad():   # utils/metrics.py:18-32 with labels 1 / 0: sigmoid(logit) > 0.5  <=>  logit > 0
            self.last_d_acc = 0.5 * ((logits_real > 0).float().mean() + (logits_fake <= 0).float().mean())
        if not self._train_d:
            return d_loss.detach()
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
        if not self._train_g:                       # loss only
            self._g_ahead = None
            self._finish_d()
            with torch.no_grad():
                outputs_fake = self.generator(inputs)
                logits_fake, _ = self.discriminator(outputs_fake)
                gan_loss = self.criterion_gan.g_loss(logits_fake)
                reg = (self.criterion_regression(outputs_fake, outputs_real, masks) if self.enable_masks
                       else self.criterion_regression(outputs_fake, outputs_real))
            return reg if self.step_count < self.warmup else self.w_gan * gan_loss + self.w_reg * reg
        for p in self._d_params:                    # D is only differentiated w.r.t. its input here
            p.requires_grad_(False)
        try:
            ahead, self._g_ahead = self._g_ahead, None
            if ahead is not None and ahead[0] is inputs:
                outputs_fake = ahead[1]
                if ahead[2]:      # computed on the side stream
                    main = torch.cuda.current_stream(inputs.device)
                    main.wait_stream(self._side)
                    outputs_fake.record_stream(main)
            else:
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
            if ahead is not None and ahead[0] is inputs and ahead[2]:
                # the early generator forward ran on the side stream, so autograd runs its backward nodes there too;
                # with the weight-gradient stream switched off nothing else joins that stream before Adam reads the gradients
                torch.cuda.current_stream(inputs.device).wait_stream(self._side)
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
        parallelism D's gradient all-reduce runs on the RCCL stream beneath it.

        Where the step qualifies (_graph_ok) it is captured into ONE hipGraph on its third call and replayed
        afterwards: ~2 400 launches become one, the host's ~20 us per launch (Python, ctypes, autograd) disappears
        from the step and the gaps between dependent kernels shrink to the hardware's."""
        if self._graph_ok(inputs, outputs_real, masks):
            return self._graph_step(inputs, outputs_real, masks, labels, eta)
        return self._eager_step(inputs, outputs_real, masks, labels, eta)

    # -- whole-step hipGraph ---------------------------------------------------------------------------
    def _graph_ok(self, inputs, outputs_real, masks):
        import os
        from .. import _lib as L
        mode = os.environ.get("BGAMD_STEP_GRAPH")
        if mode == "0" or L.PROFILE is not None or not inputs.is_cuda or torch.cuda.is_current_stream_capturing():
            return False
        if isinstance(self.generator, DistributedModel) or isinstance(self.discriminator, DistributedModel):
            return False        # the gradient all-reduces are not captured: data-parallel runs stay eager
        g = _unwrap(self.generator)
        if getattr(g, "noise_dimensions", 0) and not getattr(g, "noise_on_device", False):
            return False        # host RNG draws inside forward()
        if self.loss_type_gan not in ("ModifiedMinMax", "Wasserstein"):
            return False
        if not self._always_eval and not (self._train_d and self._train_g):
            return False
        if type(self.g_opt).__name__ != "FusedAdam" or type(self.d_opt).__name__ != "FusedAdam":
            return False
        # unset: the host-bound regime (batches of at most 2^24 field elements: 256x256x16 x 8 is 2^23); 1: every size
        return mode == "1" or inputs.numel() <= (1 << 24)

    def _graph_step(self, inputs, outputs_real, masks, labels, eta):
        from ..runtime import StatsPool as _SP
        s = self.step_count
        train_g, train_d = self._step_flags()
        eval_d = train_d or self._always_eval      # the 3-D loop evaluates both losses every iteration (train_gan3d.py:250-360)
        # everything the capture bakes into launch arguments is part of the key: a changed loss weight, criterion or
        # optimiser hyper-parameter captures a new graph instead of silently replaying the old numbers
        def _opt_key(o):
            return tuple((float(g["eps"]), float(g["weight_decay"]), tuple(float(b) for b in g["betas"]),
                          bool(g.get("amsgrad", False)), type(o).__name__ + str(g.get("decoupled", ""))) for g in o.param_groups)
        key = (tuple(inputs.shape), inputs.dtype, tuple(outputs_real.shape), masks is not None, train_g, train_d, s < self.warmup,
               id(getattr(_unwrap(self.generator), "_bg_arena", None)), id(getattr(_unwrap(self.discriminator), "_bg_arena", None)),
               float(self.w_gan), float(self.w_reg), float(self.w_gp), float(self.d_loss_scale), bool(self.enable_masks),
               id(self.criterion_regression), id(self.criterion_gan), self.loss_type_gan, id(self._gp),
               _opt_key(self.d_opt), _opt_key(self.g_opt))
        if not hasattr(self, "_graphs"):
            self._graphs, self._graph_seen, self._graph_failed = {}, {}, set()
        if key in self._graph_failed:                            # a capture of this configuration failed once: stay eager
            return self._eager_step(inputs, outputs_real, masks, labels, eta)
        e = self._graphs.get(key)
        if e is None and self._graph_seen.get(key, 0) < 2:       # two eager steps first: arenas, allocator, lazy state
            self._graph_seen[key] = self._graph_seen.get(key, 0) + 1
            return self._eager_step(inputs, outputs_real, masks, labels, eta)
        # the host's random draws, in the eager order (GANLoss.d_loss draws fake, real, swap: utils/losses.py; the
        # gradient penalty draws eta: deeplab_gan.py:99); the swap exchanges the two label tensors -- no branch in the graph
        n = outputs_real.shape[0]
        lab = None
        if eval_d and self.loss_type_gan == "ModifiedMinMax":
            lf, lr_, swap = labels if labels is not None else self.criterion_gan.draw_labels()
            lab = (lr_, lf) if swap else (lf, lr_)
        if eval_d and self.loss_type_gan == "Wasserstein" and eta is None:
            eta = torch.distributions.uniform.Uniform(0., 1.).rsample((n, 1, 1, 1))
        dev = inputs.device
        if e is None:
            e = {"x": inputs.clone(), "y": outputs_real.clone(), "m": None if masks is None else masks.clone(),
                 "lab": None if lab is None else tuple(t.to(dev).clone() for t in lab),
                 "eta": None if eta is None or self.loss_type_gan != "Wasserstein" else eta.to(dev).clone()}
        else:
            e["x"].copy_(inputs, non_blocking=True)
            e["y"].copy_(outputs_real, non_blocking=True)
            if masks is not None:
                e["m"].copy_(masks, non_blocking=True)
            if lab is not None:
                for dst, src in zip(e["lab"], lab):
                    dst.copy_(src.pin_memory() if not src.is_cuda else src, non_blocking=True)
            if e["eta"] is not None:
                e["eta"].copy_(eta.pin_memory() if not eta.is_cuda else eta, non_blocking=True)
        opts = ([self.d_opt] if train_d else []) + ([self.g_opt] if train_g else [])
        pool = _SP.get(dev)
        if "graph" not in e:
            # A capture can fail where the eager step works (a host sync inside a user-supplied loss or gradient-penalty
            # function): the optimisers' step counts are rolled back, the configuration is marked and runs eagerly for good.
            # The attempted capture runs _eager_step on the host: whatever that advanced before it died (optimiser step counts,
            # LR schedules -- _finish_d() steps D's -- BatchNorm forward counts, statistics-pool cursor, fp8 site sets, the
            # step counter) is put back, so that the eager re-run below is this iteration's ONLY step (graphs.HostStepState).
            snap = HostStepState(self, (self.generator, self.discriminator), (self.d_opt, self.g_opt),
                                 (self.d_scheduler, self.g_scheduler))
            for o in opts:
                o.prepare_replay()
            try:
                self._capture(e, key, opts, pool)
            except Exception as err:   # noqa: BLE001 -- whatever the capture raised, the eager path is the fallback
                snap.restore()
                self._graph_failed.add(key)
                self._graphs.pop(key, None)
                self._d_pending, self._g_ahead, self._want_g_ahead = False, None, False
                import warnings
                warnings.warn(f"whole-step hipGraph capture failed ({type(err).__name__}: {err}); this configuration stays eager")
                torch.cuda.synchronize()
                # the host RNG stream advances ONCE per iteration: the eager re-run takes the labels / eta drawn above
                if labels is None and lab is not None:
                    labels = (lf, lr_, swap)
                return self._eager_step(inputs, outputs_real, masks, labels, eta)
        else:
            for o in opts:
                o.prepare_replay()       # step count, lr and bias corrections of THIS step -> device
            # weights written outside the graph since the last step (load_checkpoint / load_state_dict, manual edits)
            # bump the master arena's version: refresh the bf16 / packed copies the captured kernels read (a no-op check
            # otherwise -- the captured Adam rewrites them itself)
            for net in (self.generator, self.discriminator):
                _unwrap(net).arena().sync()
            pool.used = e["pool_after"]
            for m_, k in e["nbt"]:
                m_.__dict__["_bg_nbt_pending"] = m_.__dict__.get("_bg_nbt_pending", 0) + k
            e["graph"].replay()
            if train_d and self.d_scheduler is not None:
                self.d_scheduler.step()
            if train_g and self.g_scheduler is not None:
                self.g_scheduler.step()
            self.step_count += 1
        self.last_d_acc = e["acc"]
        return e["d_loss"], e["g_loss"]

    def _capture(self, e, key, opts, pool):
        import torch.nn as nn
        bns = [m for net in (self.generator, self.discriminator) for m in _unwrap(net).modules()
               if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        before = [m.__dict__.get("_bg_nbt_pending", 0) for m in bns]
        for net in (self.generator, self.discriminator):       # packed weight copies current before the graph starts
            _unwrap(net).arena().sync()
        labels = None if e["lab"] is None else (e["lab"][0], e["lab"][1], False)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        for o in opts:
            o.capturing = True
        try:
            with torch.cuda.graph(g):
                d_loss, g_loss = self._eager_step(e["x"], e["y"], e["m"], labels, e["eta"])
        finally:
            for o in opts:
                o.capturing = False
        e.update(graph=g, d_loss=d_loss, g_loss=g_loss, acc=self.last_d_acc, pool_after=pool.used,
                 nbt=[(m, m.__dict__.get("_bg_nbt_pending", 0) - b) for m, b in zip(bns, before) if m.__dict__.get("_bg_nbt_pending", 0) != b])
        self._graphs[key] = e
        g.replay()                      # capture records, it does not execute

    def _step_flags(self):
        """(train generator, train discriminator) of the current loop step (train_gan.py:247-248)."""
        s = self.step_count
        return (s < self.warmup) or (s % self.freq_g == 0), (s >= self.warmup) and (s % self.freq_d == 0)

    def _eager_step(self, inputs, outputs_real, masks=None, labels=None, eta=None):
        StatsPool.reset_all()   # one fill clears every statistic accumulator of the previous step
        train_generator, train_discriminator = self._step_flags()
        d_loss = g_loss = None
        g = _unwrap(self.generator)
        self._want_g_ahead = (self._g_ahead_ok and train_discriminator and train_generator and inputs.is_cuda
                              and (getattr(g, "noise_dimensions", 0) == 0 or getattr(g, "noise_on_device", False)))
        if train_discriminator:
            d_loss = self.d_step(inputs, outputs_real, labels, eta)
        self._want_g_ahead = False
        if train_generator:
            g_loss = self.g_step(inputs, outputs_real, masks)
        self._finish_d()
        roll_fp8_sites(self.generator, self.discriminator)   # fp8 operand path: next step's exponents from this step's maxima
        self.step_count += 1
        return d_loss, g_loss


    # -- fp8 operand path: calibration ------------------------------------------------------------------
    def calibrate_fp8(self, inputs, outputs_real, masks=None):
        """Give every fp8 quantisation site its first exponent without changing the training state.

        The fp8 path scales each quantised tensor by a power of two taken from the max |value| the site saw ONE STEP
        EARLIER (delayed scaling); before a site has seen anything its GEMM runs on the bf16 operands.  This runs one
        loop iteration on (inputs, outputs_real) -- every site records its maximum -- and then puts parameters,
        BatchNorm buffers, optimiser moments, schedules, the step count and the host RNG back, so that the NEXT call of
        step() is the run's first iteration, on fp8 operands.  Without it the first iteration is the calibration (bf16
        GEMMs).  No-op for networks that are not in fp8 mode."""
        import copy
        nets = [_unwrap(self.generator), _unwrap(self.discriminator)]
        if not any(getattr(n, "_bg_fp8", False) for n in nets):
            return
        snap_net = [{k: v.detach().clone() for k, v in n.state_dict().items()} for n in nets]
        opts = [self.g_opt, self.d_opt]
        snap_opt = [copy.deepcopy(o.state_dict()) for o in opts]
        scheds = [self.g_scheduler, self.d_scheduler]
        snap_sched = [None if sc is None else copy.deepcopy(sc.state_dict()) for sc in scheds]
        rng, step = torch.get_rng_state(), self.step_count
        dev = next(nets[0].parameters()).device
        rng_dev = torch.cuda.get_rng_state(dev) if dev.type == "cuda" else None   # noise_on_device generators draw from it
        try:
            self._eager_step(inputs, outputs_real, masks)
        finally:
            for n, sd in zip(nets, snap_net):
                n.load_state_dict(sd)
                for m in n.modules():
                    if m.__dict__.get("_bg_nbt_pending", 0):
                        m.__dict__["_bg_nbt_pending"] = 0
            for o, sd in zip(opts, snap_opt):
                o.load_state_dict(sd)
            for sc, sd in zip(scheds, snap_sched):
                if sc is not None:
                    sc.load_state_dict(sd)
            torch.set_rng_state(rng)
            if rng_dev is not None:
                torch.cuda.set_rng_state(rng_dev, dev)
            self.step_count = step
            self._d_pending, self._g_ahead = False, None

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
        ck = {"step": self.step_count, "epoch": epoch,
              "generator": {k: v.detach().clone().contiguous() for k, v in self.generator.state_dict().items()},
              "discriminator": {k: v.detach().clone().contiguous()
                                for k, v in self.discriminator.state_dict().items()},
              "g_opt": self.g_opt.state_dict(), "d_opt": self.d_opt.state_dict(), "amp": None}
        # fp8 operand path only (an extra key the reference's loader never reads): the quantisation sites' exponents and
        # which of them are calibrated, so that a resumed run does not repeat a bf16 calibration step
        fp8 = {name: st for name, st in (("generator", getattr(_unwrap(self.generator), "_bg_arena", None)),
                                          ("discriminator", getattr(_unwrap(self.discriminator), "_bg_arena", None)))
               if st is not None and st.fp8_state() is not None}
        if fp8:
            ck["bgamd_fp8"] = {name: a.fp8_state() for name, a in fp8.items()}
        torch.save(ck, path)

    def load_checkpoint(self, path, comm, device):
        # the optimiser moments are flat buffers over the parameter arenas: build those first
        _unwrap(self.generator).arena(), _unwrap(self.discriminator).arena()
        self.step_count, epoch = comm.init_gan_training_state(_unwrap(self.generator), _unwrap(self.discriminator),
                                                              self.g_opt, self.d_opt, path, device)
        import os
        if path is not None and os.path.isfile(path):
            st = torch.load(path, map_location="cpu").get("bgamd_fp8")
            if st:
                _unwrap(self.generator).arena().load_fp8_state(st.get("generator"))
                _unwrap(self.discriminator).arena().load_fp8_state(st.get("discriminator"))
        return self.step_count, epoch


# ----------------------------------------------------------------------------------------
# Command line with the reference's flag names (train_gan.py:440-486).  wandb logging and the
# matplotlib visualiser are harness and left out; everything on the step path is here.
def main(pargs):
    import datetime as dt
    import os

    import torch.nn as nn
    from torch.utils.data import DataLoader

    from ..architecture.gpsro import deeplab as dxc
    from ..comm.distributed import comm as distcomm
    from ..data import gpsro_dataset as gpsro
    from ..utils import losses
    from ..utils import parsing_helpers as ph

    comm = distcomm(mode="dummy" if "RANK" not in os.environ else "torchrun")
    seed = 333 + 7 * comm.rank()                                   # train_gan.py:56
    torch.manual_seed(seed)
    device = torch.device("cuda", comm.local_rank())
    torch.cuda.set_device(device)
    normalizer = dxc.Identity if pargs.disable_batchnorm else nn.BatchNorm2d
    nch = len(pargs.channels)

    if pargs.synthetic_size is not None:
        h, w = pargs.synthetic_size
        g = torch.Generator(device=device).manual_seed(seed)

        def batches():
            while True:
                x = torch.randn((pargs.local_batch_size, nch, h, w), generator=g, device=device)
                yield x, x + 0.1 * torch.randn(x.shape, generator=g, device=device), None, None
        train_loader, validation_loader, field = batches(), None, (h, w)
    else:
        root = pargs.data_dir_prefix
        kw = dict(statsfile=os.path.join(root, "stats.npz"), channels=pargs.channels,
                  normalization_type="MinMax" if pargs.noise_type == "Uniform" else "MeanVariance", shuffle=True,
                  masks=pargs.enable_masks, shard_idx=comm.rank(), shard_num=comm.size(),
                  num_intra_threads=pargs.max_intra_threads, read_device=device, send_device=device)
        train_set = gpsro.GPSRODataset(os.path.join(root, "train"), **kw)
        train_loader = DataLoader(train_set, pargs.local_batch_size, drop_last=True)
        validation_loader = DataLoader(gpsro.GPSRODataset(os.path.join(root, "validation"), **kw), pargs.local_batch_size,
                                       drop_last=True)
        field = tuple(train_set.shapes[0][-2:])

    # apex.amp levels (train_gan.py:159-160) map onto the compute type: O0 is fp32 storage and
    # arithmetic, O1..O3 are bf16 storage with fp32 accumulation and fp32 master weights.
    cdt = torch.float32 if pargs.amp_opt_level == "O0" else torch.bfloat16
    generator = dxg.Generator(nch, nch, pargs.upsampler_type, pargs.noise_type, pargs.noise_dimensions, os=16,
                              pretrained=False, normalizer=normalizer, compute_dtype=cdt).to(device)
    discriminator = dxg.Discriminator(n_input=nch, os=16, pretrained=False, normalizer=normalizer, input_size=field,
                                      compute_dtype=cdt).to(device)
    criterion_gan = losses.GANLoss(pargs.loss_type_gan, pargs.local_batch_size, device)
    if pargs.loss_type_regression == "l1":                         # train_gan.py:142-152
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
    # The reference's order (train_gan.py:163-173): restore nets and optimisers FIRST, then build the schedules at the
    # restored step -- the optimiser state carries the already decayed 'lr', and a schedule started at -1 would count
    # the passed milestones (or the cosine phase) a second time.
    start_step, start_epoch = 0, 0
    if pargs.checkpoint:
        generator.arena(), discriminator.arena()   # the optimiser moments are flat buffers over the parameter arenas
        start_step, start_epoch = comm.init_gan_training_state(generator, discriminator, g_opt, d_opt, pargs.checkpoint, device)
    g_sched = ph.resume_lr_schedule(pargs.start_lr_generator, pargs.lr_schedule_generator, g_opt, start_step) \
        if pargs.lr_schedule_generator else None
    d_sched = ph.resume_lr_schedule(pargs.start_lr_discriminator, pargs.lr_schedule_discriminator, d_opt, start_step) \
        if pargs.lr_schedule_discriminator else None
    trainer = GANTrainer(comm.DistributedModel(generator), comm.DistributedModel(discriminator), g_opt, d_opt, criterion_gan,
                         criterion_regression, loss_type_gan=pargs.loss_type_gan, loss_weight_gan=pargs.loss_weight_gan,
                         loss_weight_regression=pargs.loss_weight_regression, loss_weight_gp=pargs.loss_weight_gp,
                         enable_masks=pargs.enable_masks, generator_warmup_steps=pargs.generator_warmup_steps,
                         g_scheduler=g_sched, d_scheduler=d_sched,
                         update_frequency_generator=pargs.update_frequency_generator,
                         update_frequency_discriminator=pargs.update_frequency_discriminator)
    trainer.step_count = start_step
    comm.printr('{:14.4f} REPORT: starting training'.format(dt.datetime.now().timestamp()), 0)
    epoch, d_avg, g_avg = start_epoch, 0., 0.
    while trainer.step_count < pargs.max_steps:
        for batch in train_loader:
            inputs, outputs_real = batch[0], batch[1]
            masks = batch[2] if pargs.enable_masks else None
            d_loss, g_loss = trainer.step(inputs, outputs_real, masks)
            if trainer.step_count % pargs.logging_frequency == 0 or trainer.step_count == pargs.max_steps:
                if d_loss is not None:
                    d_avg = comm.metric_average(d_loss, "train_loss_discriminator", device=device)
                if g_loss is not None:
                    g_avg = comm.metric_average(g_loss, "train_loss_generator", device=device)
                comm.printr('{:14.4f} REPORT training: step {} d_loss {} g_loss {}'.format(
                    dt.datetime.now().timestamp(), trainer.step_count, d_avg, g_avg), 0)
            if validation_loader is not None and trainer.step_count % pargs.validation_frequency == 0:
                vd, vg = trainer.validate(validation_loader, comm)
                comm.printr('{:14.4f} REPORT validation: step {} d_loss {} g_loss {}'.format(
                    dt.datetime.now().timestamp(), trainer.step_count, vd, vg), 0)
            if pargs.save_frequency > 0 and trainer.step_count % pargs.save_frequency == 0 and comm.rank() == 0:
                trainer.save_checkpoint(os.path.join(pargs.output_dir, pargs.model_prefix + "_step_" +
                                                     str(trainer.step_count) + ".cpt"), epoch)     # train_gan.py:411
            if trainer.step_count >= pargs.max_steps:
                break
        epoch += 1
        if comm.rank() == 0:   # every epoch, whatever --save_frequency says                    # train_gan.py:419-431
            trainer.save_checkpoint(os.path.join(pargs.output_dir, pargs.model_prefix + "_epoch_" + str(epoch) + ".cpt"), epoch)
    return trainer


def build_parser():
    import argparse as ap

    class StoreDictKeyPair(ap.Action):                            # train_gan.py:41-47
        def __call__(self, parser, namespace, values, option_string=None):
            setattr(namespace, self.dest, dict(kv.split("=") for kv in values.split(",")))

    AP = ap.ArgumentParser()
    AP.add_argument("--data_dir_prefix", type=str, default="/", help="dataset root with train/, validation/, stats.npz")
    AP.add_argument("--output_dir", type=str, default=".")
    AP.add_argument("--checkpoint", type=str, default=None)
    AP.add_argument("--channels", type=int, nargs='+', default=list(range(45)))
    AP.add_argument("--upsampler_type", type=str, default="Interpolate", choices=["Interpolate", "Deconv", "Deconv1x"])
    AP.add_argument("--noise_type", type=str, default="Uniform", choices=["Uniform", "Normal"])
    AP.add_argument("--noise_dimensions", type=int, default=1)
    AP.add_argument("--local_batch_size", type=int, default=1)
    AP.add_argument("--max_steps", type=int, default=100)
    AP.add_argument("--generator_warmup_steps", type=int, default=0)
    AP.add_argument("--update_frequency_generator", type=int, default=1)
    AP.add_argument("--update_frequency_discriminator", type=int, default=1)
    AP.add_argument("--optimizer_generator", type=str, default="Adam", choices=["Adam", "AdamW"])
    AP.add_argument("--optimizer_discriminator", type=str, default="Adam", choices=["Adam", "AdamW"])
    AP.add_argument("--start_lr_generator", type=float, default=1e-3)
    AP.add_argument("--start_lr_discriminator", type=float, default=1e-3)
    AP.add_argument("--adam_eps", type=float, default=1e-8)
    AP.add_argument("--weight_decay", type=float, default=1e-4)
    AP.add_argument("--loss_type_gan", type=str, default="ModifiedMinMax", choices=["ModifiedMinMax", "Wasserstein"])
    AP.add_argument("--loss_type_regression", type=str, default="l1", choices=["l1", "smooth_l1", "l2"])
    AP.add_argument("--loss_weight_gan", type=float, default=1.)
    AP.add_argument("--loss_weight_regression", type=float, default=1.)
    AP.add_argument("--loss_weight_gp", type=float, default=10.)
    AP.add_argument("--lr_schedule_generator", action=StoreDictKeyPair, default={})
    AP.add_argument("--lr_schedule_discriminator", action=StoreDictKeyPair, default={})
    AP.add_argument("--logging_frequency", type=int, default=10)
    AP.add_argument("--validation_frequency", type=int, default=100)
    AP.add_argument("--save_frequency", type=int, default=0)
    AP.add_argument("--max_intra_threads", type=int, default=8)
    AP.add_argument("--enable_masks", action='store_true')
    AP.add_argument("--disable_batchnorm", action='store_true')
    AP.add_argument("--amp_opt_level", type=str, default="O1", help="O0: fp32; O1..O3: bf16 storage, fp32 accumulate")
    # accepted for command-line compatibility; logging/visualisation harness is not part of this build
    AP.add_argument("--run_tag", type=str, default="run")
    AP.add_argument("--model_prefix", type=str, default="model")
    AP.add_argument("--max_inter_threads", type=int, default=1)
    AP.add_argument("--training_visualization_frequency", type=int, default=50)
    AP.add_argument("--validation_visualization_frequency", type=int, default=5)
    AP.add_argument("--disable_gds", action='store_true')
    AP.add_argument("--resume_logging", action='store_true')
    AP.add_argument("--synthetic_size", type=int, nargs=2, default=None, metavar=("H", "W"),
                    help="train on synthetic N(0,1) fields of this size instead of a dataset")
    return AP


if __name__ == "__main__":
    main(build_parser().parse_args())
