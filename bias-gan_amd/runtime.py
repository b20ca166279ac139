"""Parameter arenas and the module base class.

MI355X-first memory layout for the model state: every trainable tensor of a
network lives in ONE flat fp32 arena (plus one flat gradient arena and, for the
bf16 path, one flat bf16 copy and one flat CRSK-transposed copy for the
data-gradient GEMMs).  That gives
  * one fused Adam launch per network instead of ~500,
  * one contiguous buffer to all-reduce over RCCL in a few large messages,
  * conv weights stored the way the MFMA kernels read them (KRSC), while the
    nn.Parameter the user sees keeps the reference's [Cout,Cin,KH,KW] shape as a
    permuted VIEW of the arena (so state_dict keys/shapes interchange with
    reference checkpoints; deeplab.py / deeplab_gan.py define the names).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L


def default_compute_dtype() -> torch.dtype:
    v = os.environ.get("BGAMD_DTYPE", "bf16").lower()
    return torch.float32 if v in ("f32", "fp32", "float32") else torch.bfloat16


def vec_of(dtype: torch.dtype) -> int:
    return 8 if dtype == torch.bfloat16 else 4


def pad_to(c: int, v: int) -> int:
    return (c + v - 1) // v * v


class ParamSlot:
    """Where one nn.Parameter lives inside the arenas."""
    __slots__ = ("name", "param", "kind", "off", "numel", "phys_shape", "t_off", "k_off", "krsc", "f8", "k8_off", "t8_off")

    def __init__(self, name, param, kind, phys_shape):
        self.name, self.param, self.kind, self.phys_shape = name, param, kind, tuple(phys_shape)
        self.numel = 1
        for s in phys_shape:
            self.numel *= s
        self.off = -1
        self.t_off = -1   # offset inside the packed CRSK arena, dense convs only
        self.k_off = -1   # offset inside the packed KRSC arena, dense convs only
        self.krsc = None  # (K, RS, C, Cp, Kp) for dense convs
        self.f8 = None    # index into the arena's fp8 tables (dense convs of an fp8 arena)
        self.k8_off = self.t8_off = -1


class StatsPool:
    """Zero-initialised fp64 scratch for per-channel statistic sums, handed out as slices of one
    buffer that is cleared with ONE fill per training step (GANTrainer.step calls reset()),
    instead of one torch.zeros launch per normalisation layer and direction (~600 per step)."""
    _pools = {}
    epoch = 0     # bumped by reset_all(): one per training iteration.  ops' hand-over notes are only valid inside one epoch.

    def __init__(self, device, n=1 << 22):
        self.buf = torch.zeros(n, dtype=torch.float64, device=device)
        self.used = 0

    @classmethod
    def get(cls, device) -> "StatsPool":
        key = (device.type, device.index)
        if key not in cls._pools:
            cls._pools[key] = StatsPool(device)
        return cls._pools[key]

    def take(self, *shape) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        n_al = (n + 31) // 32 * 32
        if self.used + n_al > self.buf.numel():   # exhausted (nobody resets): plain allocation
            return torch.zeros(shape, dtype=torch.float64, device=self.buf.device)
        out = self.buf[self.used:self.used + n].view(shape)
        self.used += n_al
        return out

    def reset(self):
        if self.used:
            self.buf[:self.used].zero_()
            self.used = 0

    @classmethod
    def reset_all(cls):
        cls.epoch += 1
        for p in cls._pools.values():
            p.reset()

    @classmethod
    def all(cls):
        return list(cls._pools.values())


_ARENA_OF = {}  # id(param) -> weakref to its arena


def arena_of(p):
    """The arena a Parameter currently lives in (None before the first GPU forward)."""
    r = _ARENA_OF.get(id(p))
    a = None if r is None else r()
    if a is not None and a.by_param.get(id(p)) is not None and a.by_param[id(p)].param is p:
        return a
    return None


class Arena:
    """Flat storage for the parameters of one network (see module docstring)."""

    def __init__(self, root: nn.Module, compute_dtype: torch.dtype, fp8: bool = False):
        self.compute_dtype = compute_dtype
        self.fp8 = bool(fp8) and compute_dtype == torch.bfloat16
        self.slots: List[ParamSlot] = []
        self.by_param: Dict[int, ParamSlot] = {}
        dev = None
        for name, mod in root.named_modules():
            spec = getattr(mod, "_bg_param_layout", None)
            for pname, p in mod._parameters.items():
                if p is None:
                    continue
                dev = p.device if dev is None else dev
                if p.device != dev:
                    raise RuntimeError("bias_gan_amd: all parameters of a network must live on one device")
                full = f"{name}.{pname}" if name else pname
                kind, phys = ("plain", tuple(p.shape))
                if spec is not None and pname in spec:
                    kind, phys = spec[pname](p, vec_of(compute_dtype))
                slot = ParamSlot(full, p, kind, phys)
                self.slots.append(slot)
                self.by_param[id(p)] = slot
        if dev is None:
            raise RuntimeError("bias_gan_amd: module has no parameters")
        self.device = dev
        off = 0
        t_off = k_off = 0
        g = L.kpad(compute_dtype)
        for s in self.slots:
            s.off = off
            off += pad_to(s.numel, 64)  # 256-byte aligned segments
            if s.kind in ("conv", "conv3d"):    # conv3d: [Kp, KH, KW, KD*Cp] -- a 2-D conv over the depth-unfolded input
                k, r, q, c = s.phys_shape
                cp, kp = pad_to(c, g), pad_to(k, g)
                s.krsc = (k, r * q, c, cp, kp)
                s.k_off, s.t_off = k_off, t_off
                k_off += pad_to(k * r * q * cp, 64)
                t_off += pad_to(c * r * q * kp, 64)
        self.numel = off
        self.master = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.lp = torch.zeros(off, dtype=torch.bfloat16, device=dev) if compute_dtype == torch.bfloat16 else None
        # packed operand copies of the dense conv weights (reduction dim zero-padded)
        self.wk = torch.zeros(max(k_off, 1), dtype=compute_dtype, device=dev)
        self.wt = torch.zeros(max(t_off, 1), dtype=compute_dtype, device=dev)
        tbl = [[s.off, s.k_off, s.t_off, s.krsc[0], s.krsc[1], s.krsc[2], s.krsc[3], s.krsc[4]]
               for s in self.slots if s.krsc is not None]
        self.tr_tbl = torch.tensor(tbl if tbl else [[0] * 8], dtype=torch.int64, device=dev)
        self.tr_layers = len(tbl)
        self.tr_max = max([s.krsc[0] * s.krsc[1] * s.krsc[3] + s.krsc[2] * s.krsc[1] * s.krsc[4]
                           for s in self.slots if s.krsc is not None] + [1])
        # move the current values in and re-point the Parameters at arena views
        with torch.no_grad():
            for s in self.slots:
                view = self._view(self.master, s)
                view.copy_(s.param.data)
                s.param.data = view
                s.param.grad = None
        self._synced_version = -1
        self.ddp = None
        self.attach_grads()
        if self.fp8:
            self._build_fp8()
        import weakref
        for s in self.slots:
            _ARENA_OF[id(s.param)] = weakref.ref(self)

    # -- fp8 operand path (BASELINE.json configs[4]) ---------------------------------------
    def _build_fp8(self):
        """e4m3 copies of the dense conv weights (reduction dim zero-padded to 128) with one power-of-two exponent per
        layer, and the quantisation sites of the activations: two per layer (forward input: e4m3; output gradient:
        e5m2) with delayed scaling -- a site's exponent comes from the max |value| it saw in the previous step
        (bg_quant_fp8 records it, roll_fp8() turns it into the next step's exponent)."""
        dev = self.device
        layers = [s for s in self.slots if s.krsc is not None]
        k_off = t_off = 0
        rows = []
        for i, s in enumerate(layers):
            k, rs, c, _, _ = s.krsc
            c8, k8 = pad_to(pad_to(c, 16), 128), pad_to(pad_to(k, 16), 128)
            s.f8 = i
            s.k8_off, s.t8_off = k_off, t_off
            rows.append([s.off, k_off, t_off, k, rs, c, c8, k8])
            k_off += pad_to(k * rs * c8, 64)
            t_off += pad_to(c * rs * k8, 64)
        n = max(len(layers), 1)
        self.wk8 = torch.zeros(max(k_off, 16), dtype=torch.uint8, device=dev)
        self.wt8 = torch.zeros(max(t_off, 16), dtype=torch.uint8, device=dev)
        self.tbl8 = torch.tensor(rows if rows else [[0] * 8], dtype=torch.int64, device=dev)
        self.max8 = max([r[3] * r[4] * r[6] + r[5] * r[4] * r[7] for r in rows] + [1])
        self.w_exp = torch.zeros(n, dtype=torch.int32, device=dev)
        self._w_amax = torch.zeros(n, dtype=torch.int32, device=dev)
        # sites: 2 * layer + 0 = forward input (e4m3), 2 * layer + 1 = output gradient (e5m2)
        self.site_exp = torch.zeros(2 * n, dtype=torch.int32, device=dev)
        self.site_amax = torch.zeros(2 * n, dtype=torch.int32, device=dev)
        self.site_fmt = torch.tensor([L.FP8_E4M3, L.FP8_E5M2] * n, dtype=torch.int32, device=dev)
        # a site is READY once a roll has turned a recorded maximum into its exponent; until then its GEMM runs on the bf16
        # operand and the quantiser only records max |value| (the first step of a run, or the first time a schedule -- warm-up,
        # update frequencies -- reaches a layer's backward pass).  Tracked on the host: it knows which sites it launched.
        self._site_seen, self._site_ready = set(), set()
        self.sites_ready = False       # any roll so far (informational)
        self.n_fp8_layers = len(layers)

    def roll_fp8(self):
        """End of a training step: exponents of the next step from this step's recorded maxima (one bit of head-room)."""
        if self.fp8 and self.n_fp8_layers:
            L.call("bg_fp8_roll", self.site_exp.data_ptr(), self.site_amax.data_ptr(), self.site_fmt.data_ptr(),
                   2 * self.n_fp8_layers, 1)
            self._site_ready |= self._site_seen
            self._site_seen.clear()
            self.sites_ready = True

    def fp8_state(self):
        """The delayed-scaling state a checkpoint must carry for a resumed fp8 run to continue like an uninterrupted one:
        every site's exponent and which sites are calibrated (None outside fp8 mode)."""
        if not (self.fp8 and self.n_fp8_layers):
            return None
        return {"site_exp": self.site_exp.detach().cpu().clone(), "site_ready": sorted(self._site_ready),
                "names": [s.name for s in self.slots if s.krsc is not None]}

    def load_fp8_state(self, st):
        """Inverse of fp8_state(); a state from a different network layout is ignored (the run recalibrates)."""
        if st is None or not (self.fp8 and self.n_fp8_layers):
            return False
        if list(st.get("names", [])) != [s.name for s in self.slots if s.krsc is not None]:
            return False
        self.site_exp.copy_(st["site_exp"].to(self.site_exp.device))
        self._site_ready = set(int(i) for i in st["site_ready"])
        self._site_seen = set()
        self.sites_ready = bool(self._site_ready)
        return True

    def site_ready(self, s: ParamSlot, grad: bool) -> bool:
        """Has this site's exponent been derived from data it saw?"""
        return 2 * s.f8 + (1 if grad else 0) in self._site_ready

    def site_seen(self, s: ParamSlot, grad: bool):
        """A launch that records this site's max |value| has been issued in this step (the next roll makes it ready)."""
        self._site_seen.add(2 * s.f8 + (1 if grad else 0))

    def weight8_ptr(self, s: ParamSlot) -> int:
        return self.wk8.data_ptr() + s.k8_off

    def weight8_t_ptr(self, s: ParamSlot) -> int:
        return self.wt8.data_ptr() + s.t8_off

    def w_exp_ptr(self, s: ParamSlot) -> int:
        return self.w_exp.data_ptr() + 4 * s.f8

    def site_ptrs(self, s: ParamSlot, grad: bool):
        """(exponent pointer, amax pointer) of a layer's forward-input / output-gradient site."""
        i = 2 * s.f8 + (1 if grad else 0)
        return self.site_exp.data_ptr() + 4 * i, self.site_amax.data_ptr() + 4 * i

    # -- views ---------------------------------------------------------------------
    def _view(self, flat: torch.Tensor, s: ParamSlot) -> torch.Tensor:
        seg = flat[s.off:s.off + s.numel]
        p = s.param
        if s.kind == "conv":       # arena [Kp,R,S,Cp] -> logical [K,C,R,S]
            k, c = p.shape[0], p.shape[1]
            return seg.view(s.phys_shape).permute(0, 3, 1, 2)[:k, :c]
        if s.kind == "dw":         # arena [R,S,Cp] -> logical [C,1,R,S]
            c = p.shape[0]
            return seg.view(s.phys_shape).permute(2, 0, 1)[:c].unsqueeze(1)
        if s.kind == "conv3d":     # arena [Kp,R,S,KD,Cp] -> logical [K,C,KD,R,S]
            k, c, kd = p.shape[0], p.shape[1], p.shape[2]
            kp, r, q, kdc = s.phys_shape
            return seg.view(kp, r, q, kd, kdc // kd).permute(0, 4, 3, 1, 2)[:k, :c]
        if s.kind == "dw3d":       # arena [KD,R,S,Cp] -> logical [C,1,KD,R,S]
            c = p.shape[0]
            return seg.view(s.phys_shape).permute(3, 0, 1, 2)[:c].unsqueeze(1)
        if s.kind == "vec":        # arena [Cp] -> logical [C]
            return seg[:p.shape[0]]
        return seg.view(p.shape)

    def attach_grads(self):
        """Make every Parameter's .grad a view of the gradient arena."""
        for s in self.slots:
            if s.param.requires_grad:
                s.param.grad = self._view(self.grad, s)

    def zero_grad(self):
        L.call("bg_fill_f32", self.grad.data_ptr(), 0.0, self.numel)

    def ensure_grad(self, s: ParamSlot):
        """Called from a backward before it accumulates into the arena.  If a
        foreign optimiser set .grad to None (zero_grad(set_to_none=True)), zero
        this parameter's segment and re-attach the view."""
        if arena_of(s.param) is not self:
            raise RuntimeError("bias_gan_amd: backward through a graph whose parameter arena was rebuilt "
                               "(module moved / dtype changed / sub-module arena replaced after the forward pass)")
        g = s.param.grad
        if g is None or g.data_ptr() != self.grad.data_ptr() + 4 * s.off:
            L.call("bg_fill_f32", self.grad.data_ptr() + 4 * s.off, 0.0, s.numel)
            s.param.grad = self._view(self.grad, s)

    # -- low-precision / transposed copies ---------------------------------------------
    def weights_changed(self):
        self._synced_version = -1

    def sync(self):
        """Refresh the bf16 copy and the CRSK copies if the master arena changed
        (torch-side writes bump the tensor version; our Adam calls refresh_copies())."""
        v = self.master._version
        if v != self._synced_version:
            self.refresh_copies(cast=True)
            self._synced_version = v

    def refresh_copies(self, cast: bool):
        if self.lp is not None and cast:
            L.call("bg_cast_f32_to_bf16", self.master.data_ptr(), self.lp.data_ptr(), self.numel)
        if self.tr_layers:
            src = self.lp if self.lp is not None else self.master
            L.call("bg_pack_conv_weights", L.dt(self.compute_dtype), src.data_ptr(), self.wk.data_ptr(),
                   self.wt.data_ptr(), self.tr_tbl.data_ptr(), self.tr_layers, self.tr_max)
            if self.fp8:   # from the fp32 master: one rounding, per-layer exponent from the layer's own max |w|
                L.call("bg_pack_conv_weights_fp8", self.master.data_ptr(), self.wk8.data_ptr(), self.wt8.data_ptr(),
                       self.tbl8.data_ptr(), self.n_fp8_layers, self.max8, self.w_exp.data_ptr(), self._w_amax.data_ptr())

    def weight_ptr(self, s: ParamSlot) -> int:
        """Device pointer of the compute-dtype operand copy of a parameter: packed KRSC for
        dense convs, the flat [R,S,C] copy for depthwise weights."""
        if s.krsc is not None:
            return self.wk.data_ptr() + self.wk.element_size() * s.k_off
        if self.lp is not None:
            return self.lp.data_ptr() + 2 * s.off
        return self.master.data_ptr() + 4 * s.off

    def weight_t_ptr(self, s: ParamSlot) -> int:
        return self.wt.data_ptr() + self.wt.element_size() * s.t_off

    def master_ptr(self, s: ParamSlot) -> int:
        return self.master.data_ptr() + 4 * s.off

    def grad_ptr(self, s: ParamSlot) -> int:
        return self.grad.data_ptr() + 4 * s.off


class BGModule(nn.Module):
    """Base of every module in this package: lazily (re)builds the arena of the
    root it is called through, and invalidates it when .to()/.cuda() moves data."""

    _bg_arena: Optional[Arena] = None
    _bg_dtype: Optional[torch.dtype] = None
    _bg_fp8: bool = False

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        # The OUTERMOST module that is called builds the arena for its whole
        # subtree before any child runs (children called through it find it ready).
        self.register_forward_pre_hook(BGModule._ensure_arena_hook)

    @staticmethod
    def _ensure_arena_hook(mod, inputs):
        mod.arena()

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        for m in self.modules():
            if isinstance(m, BGModule):
                object.__setattr__(m, "_bg_arena", None)
        return out

    def compute_dtype(self) -> torch.dtype:
        return self._bg_dtype or default_compute_dtype()

    def set_compute_dtype(self, dtype: torch.dtype):
        """torch.float32: the parity path; torch.bfloat16: the measured path; torch.float8_e4m3fn: bf16 storage with the
        dense convolutions' forward and data-gradient GEMMs on fp8 operands (e4m3 weights and inputs, e5m2 output
        gradients, fp32 accumulation: BASELINE.json configs[4])."""
        fp8 = dtype == torch.float8_e4m3fn
        if fp8:
            dtype = torch.bfloat16
        for m in self.modules():
            if isinstance(m, BGModule):
                object.__setattr__(m, "_bg_dtype", dtype)
                object.__setattr__(m, "_bg_fp8", fp8)
                object.__setattr__(m, "_bg_arena", None)
        return self

    def arena(self) -> Arena:
        """The arena this module's parameters live in (built on first use, with
        this module as root if no ancestor built one)."""
        a = self._bg_arena
        if a is None:
            first = next(self.parameters(), None)
            if first is None or first.device.type != "cuda":
                raise RuntimeError("bias_gan_amd: the HIP path needs the module on a GPU (call .to('cuda')); "
                                   "there is no CPU fallback")
            a = Arena(self, self.compute_dtype(), self._bg_fp8)
            for m in self.modules():
                if isinstance(m, BGModule):
                    object.__setattr__(m, "_bg_arena", a)
        a.sync()
        return a

    def slot(self, p: nn.Parameter) -> ParamSlot:
        return self.arena().by_param[id(p)]

    def flush_counters(self):
        """Fold the host-side BatchNorm forward counts into num_batches_tracked."""
        for m in self.modules():
            pend = m.__dict__.get("_bg_nbt_pending", 0)
            if pend and getattr(m, "num_batches_tracked", None) is not None:
                m.num_batches_tracked += pend
                m.__dict__["_bg_nbt_pending"] = 0

    def state_dict(self, *a, **kw):
        self.flush_counters()
        return super().state_dict(*a, **kw)

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        if self._bg_arena is not None:
            self._bg_arena.weights_changed()
        return out
