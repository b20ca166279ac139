"""torch.autograd.Function wrappers over the C ABI (include/bgamd.h).

Same shape as the reference's own native-op pattern (Conv2dLocalFunction,
architecture/gpsro/deeplab.py:9-22): forward() calls the extension and saves
tensors, backward() calls the extension's backward entry points.  Activations
inside a network are NHWC tensors [N,H,W,C] (C padded to a 16-byte multiple,
pad lanes zero) whose pixel stride may exceed C (channel slices of wider
buffers).  Parameter gradients are accumulated by the kernels straight into the
network's flat gradient arena (runtime.Arena); the Functions return None for
them.
"""
from __future__ import annotations

from typing import Optional, Tuple

import os as _os

import torch

from . import _lib as L
from .runtime import Arena, ParamSlot, StatsPool, vec_of


# --------------------------------------------------------------------------- helpers
def _ld(t: torch.Tensor) -> int:
    """Pixel stride of an NHWC tensor (strides of size-1 dims are meaningless, so
    take it from the innermost spatial dim that has extent > 1)."""
    n, h, w, c = t.shape
    if w > 1:
        return t.stride(2)
    if h > 1:
        return t.stride(1)
    if n > 1:
        return t.stride(0)
    return c


def _is_nhwc(t: torch.Tensor, vec: int) -> bool:
    if t.dim() != 4 or t.data_ptr() % 16 != 0:
        return False
    n, h, w, c = t.shape
    if c % vec != 0 or (c > 1 and t.stride(3) != 1):
        return False
    ld = _ld(t)
    if ld < c or ld % vec != 0:
        return False
    if w > 1 and h > 1 and t.stride(1) != w * ld:
        return False
    if n > 1 and h * w > 1 and t.stride(0) != h * w * ld:
        return False
    return True


def nhwc(t: torch.Tensor) -> torch.Tensor:
    """Return `t` if the kernels can address it as NHWC rows, else a packed copy."""
    vec = 8 if t.dtype == torch.bfloat16 else 4
    if _is_nhwc(t, vec):
        return t
    return t.contiguous()


def ld_of(t: torch.Tensor) -> int:
    return _ld(t)


def rows_of(t: torch.Tensor) -> int:
    return t.shape[0] * t.shape[1] * t.shape[2]


ROW_ALIGN_BYTES = int(_os.environ.get("BGAMD_ROW_ALIGN", "64"))   # pixel rows start on 64-byte boundaries (0: packed)


def new_act(n, h, w, c, dtype, device) -> torch.Tensor:
    """Fresh NHWC activation.  The pixel stride is rounded up to a 64-byte multiple (C = 728 bf16:
    1456 -> 1472 bytes) so that the 64-byte K-slabs the GEMM kernels fetch per pixel, and the row
    segments of the element-wise kernels, do not straddle memory sectors; the pad lanes are never
    read or written (every kernel takes the pixel stride separately from C).  Rows of at most half that (the 16
    field channels in bf16: 32 bytes) stay packed: they tile the 64-byte sectors exactly, and padding them would
    double the traffic of the layout conversions and of the first convolution's operand."""
    es = 2 if dtype == torch.bfloat16 else 4
    q = ROW_ALIGN_BYTES // es
    ld = (c + q - 1) // q * q if q and (2 * c * es > ROW_ALIGN_BYTES or ROW_ALIGN_BYTES % (c * es)) else c
    if ld == c:
        return torch.empty((n, h, w, c), dtype=dtype, device=device)
    return torch.empty((n, h, w, ld), dtype=dtype, device=device)[..., :c]


def _f32(*shape, device):
    return torch.zeros(shape, dtype=torch.float32, device=device)


def _f64(*shape, device):
    """Zeroed fp64 statistic accumulators (slices of the per-device StatsPool)."""
    return StatsPool.get(device).take(*shape)


def _e32(*shape, device):
    """Uninitialised fp32 scratch for buffers a kernel writes completely (no fill launch)."""
    return torch.empty(shape, dtype=torch.float32, device=device)


# ------------------------------------------------------- weight gradients on a second stream (opt-in)
# In a backward pass the data-gradient chain is the critical path; the weight gradients hang off it
# and nothing downstream reads them before the optimiser.  They CAN be issued on a second HIP stream so
# that their workgroups fill the tails and ramps of the (short) kernels of the chain; the stream is
# joined back into the caller's stream when the autograd engine finishes the pass (queue_callback),
# so .grad is ordered like any other result of backward().  Off by default since round 4 (BGAMD_WGRAD_STREAM=1
# enables): with most weight gradients deferred to the grouped launches at the end of the pass there is little
# left to overlap, and what is left competes with the chain's own kernels for the chip -- same-box A/B
# (scripts/gpu_wgs.sh): 1152 x 768 114.2 -> 112.7 ms, wgan-gp 135.6 -> 133.2, 2304 x 1536 x 32 232.7 -> 229.2,
# 256 x 256 22.0 -> 21.7 without it.
_WG_STREAMS = {}
_WG_PENDING = {}     # device -> graph task id of the backward pass whose callback is registered
_WG_ENABLED = _os.environ.get("BGAMD_WGRAD_STREAM", "0") != "0" and not _os.environ.get("BGAMD_NO_WGRAD_STREAM")
# ... and then only for layers whose activation has at least this many elements: below it (the 16 x 16 and 32 x 32 maps
# of the 256 x 256 configuration) a launch is a few microseconds of latency and the fork / join costs more than it hides
_WG_MIN_ELEMS = int(_os.environ.get("BGAMD_WGRAD_STREAM_MIN", str(8 << 20)))


# Per-layer weight gradients (the layers the gang kernel does not take) CAN go through the workspace form: no float atomics,
# bit-reproducible (BGAMD_WGRAD_WS=1).  Not the default: the atomics are not what these launches wait for -- 3 x 3 128 -> 128
# at 576 x 384 runs 866 -> 836 us, but the HBM-bound 1 x 1 layers pay for writing and re-reading 32 MB of split tiles
# (168 -> 277 us; scripts/bench_wgrad_ws.py, profiles/r04_bench_wgrad_ws.txt).
_WG_WS = _os.environ.get("BGAMD_WGRAD_WS", "0") != "0"


def _wgrad_layer(desc, x, g, dw_ptr, dbias=None, swap=False):
    """(entry point, arguments, tensors to keep alive) of one layer's weight gradient; swap: transposed convolutions hand
    their operands over with the roles exchanged."""
    a, b = (g, x) if swap else (x, g)
    if not _WG_WS:
        return "bg_conv2d_bwd_weight", (desc, a.data_ptr(), b.data_ptr(), dw_ptr, dbias), (x, g)
    nbytes = L.wgrad_ws_bytes(desc)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    return "bg_conv2d_bwd_weight_ws", (desc, a.data_ptr(), b.data_ptr(), dw_ptr, dbias, ws.data_ptr(), nbytes), (x, g, ws)


def _wg_register(key):
    """One end-of-backward callback per device and backward pass.  A pass is identified by the autograd engine's graph
    task id: if an earlier pass raised, its callback never ran -- its entry here is stale, the groups it queued are
    dropped (their gradients belong to the failed pass) and this pass registers its own callback."""
    task = torch._C._current_graph_task_id()   # >= 0: every caller checked that a backward pass is running
    if _WG_PENDING.get(key, None) != task:
        if key in _WG_PENDING:
            _WG_GROUPS.pop(key, None)
        _WG_PENDING[key] = task
        torch.autograd.Variable._execution_engine.queue_callback(lambda: wgrad_join(key))


def wgrad_call(dev, tensors, name, *args):
    """Launch a weight-gradient entry point on the second stream (after everything enqueued so far on the
    caller's stream); `tensors` are the operands whose memory must outlive that launch."""
    if (not _WG_ENABLED or L.PROFILE is not None or torch._C._current_graph_task_id() < 0
            or max(t.numel() for t in tensors[:2]) < _WG_MIN_ELEMS):
        L.call(name, *args)   # the profile step times ONE kernel per event pair; outside a backward pass nothing would join
        return
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ent = _WG_STREAMS.get(key)
    if ent is None:
        side = L.side_stream(dev, "wgrad")
        ent = _WG_STREAMS[key] = (side, side.cuda_stream)
    side, raw = ent
    side.wait_stream(torch.cuda.current_stream(dev))
    _wg_register(key)
    for t in tensors:
        t.record_stream(side)
    L.call_on(raw, name, *args)   # the kernel only accumulates into the gradient arena: no allocation on that stream


def wgrad_join(key=None):
    """Make the current stream wait for the weight-gradient stream(s) (after issuing the grouped launches that were
    collected during this backward pass)."""
    for k in ([key] if key is not None else list(set(_WG_STREAMS) | set(_WG_GROUPS))):
        wgrad_group_flush(k)
        ent = _WG_STREAMS.get(k)
        if ent is not None:
            torch.cuda.current_stream(ent[0].device).wait_stream(ent[0])
        _WG_PENDING.pop(k, None)


# Pointwise (1x1) weight gradients are not launched layer by layer: a single 728 x 728 layer has 9 output tiles, so it can
# only fill the chip by splitting its pixels over many workgroups that all add a copy of dW with float atomics.  They are
# collected per shape during the backward pass and issued as ONE launch per shape when the pass ends
# (bg_conv2d_bwd_weight_grouped: the 48 + 2 identical middle-flow layers are one launch whose workgroups each keep a
# dW tile in registers over a whole layer).  BGAMD_WGRAD_GROUP=0 restores the per-layer launches.
_WG_GROUPS = {}
_WG_GROUP_ENABLED = _os.environ.get("BGAMD_WGRAD_GROUP", "1") != "0"
_WG_CHUNK = int(_os.environ.get("BGAMD_WGG_CHUNK", "0"))   # layers per grouped launch issued DURING the pass (0: all at its end)


def wgrad_group_ok(dtype, kh, kw, stride, pad, dil, has_bias) -> bool:
    return (_WG_GROUP_ENABLED and dtype == torch.bfloat16 and kh == 1 and kw == 1 and stride == 1 and pad == 0 and dil == 1
            and not has_bias)


_WG_TAPS_ENABLED = _os.environ.get("BGAMD_WGRAD_TAPS", "1") != "0"


def wgrad_taps_ok(dtype, n, h, w, cin, cout, kh, kw, stride, pad, dil, has_bias, ldx, ldy) -> bool:
    """k x k stride-1 'same' convolutions whose weight gradient goes through the gang kernel (bg_conv2d_bwd_weight_grouped_taps):
    measured (scripts/bench_wgrad.py) 1.3-1.9x the per-layer kernel on the decoder's and the ASPP's 3 x 3 layers; layers
    whose 256 x 256 tiles would be mostly padding, or whose maps are too small to give every gang a long range, stay."""
    if not (_WG_GROUP_ENABLED and _WG_TAPS_ENABLED and dtype == torch.bfloat16 and kh == kw and kh in (3, 5) and stride == 1
            and pad == dil * (kh - 1) // 2 and not has_bias and _gang_shape_ok(cin, cout, 0.55)):
        return False
    rows = n * h * w
    if (rows + dil * (kh // 2) * (w + 1)) * max(ldx, ldy) * 2 >= (1 << 31) or cout * kh * kw * cin * 4 >= (1 << 31):
        return False            # the gang kernel's 32-bit operand offsets
    return rows * kh * kw >= 64 * 256           # >= 64 K-steps (of 32 pixels) per workgroup


def wgrad_group_add(dev, x, g, dw_ptr, rows, cin, cout, desc):
    """Queue dW += g^T x of one pointwise layer for the end-of-backward grouped launch."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if (not _gang_shape_ok(cin, cout) or torch._C._current_graph_task_id() < 0
            or rows * max(ld_of(x), ld_of(g)) * 2 >= (1 << 31)):    # the gang kernel's 32-bit operand offsets
        # shapes the gang kernel never takes (mostly padding in 256 x 256 tiles: the 128-channel entry-flow layers, 450 MB
        # of gradient each at 1152 x 768, batch 8) go out now, on the side stream under the data-gradient chain, instead
        # of holding their operands until the pass ends; so does a call from outside a backward pass (no callback to flush)
        name, args, keep = _wgrad_layer(desc, x, g, dw_ptr)
        wgrad_call(dev, keep, name, *args)
        return
    _wg_register(key)
    sig = (rows, cin, cout, ld_of(x), ld_of(g))
    jobs = _WG_GROUPS.setdefault(key, {}).setdefault(sig, [])
    jobs.append((x, g, dw_ptr, desc))
    if _WG_CHUNK and len(jobs) >= _WG_CHUNK:
        # issue the group's launch now, on the side stream: its (MFMA / LDS-bound, one workgroup per CU) workgroups run
        # under the HBM-bound normalisation / depthwise kernels of the data-gradient chain instead of after the pass
        wgrad_group_flush(key, sig)


def _gang_pays(n_layers, cin, cout, rows=0) -> bool:
    """Measured (scripts/bench_wgrad.py): the gang launch beats the per-layer kernel for groups of layers (3.0x on the 48
    middle-flow layers), for single layers with many output tiles (1536 x 1536 and up: 1.3-1.6x) and for single layers
    whose pixel count gives every gang a long range (728 -> 728 at 144 x 96: 1.5x); a single layer of up to 12 tiles
    on few pixels is 0.9x (one range per gang is too short against its flush)."""
    tiles = (-(-cin // 256)) * (-(-cout // 256))
    long_ranges = rows // 32 >= 96 * max(1, 256 // tiles) and tiles >= 2
    return _gang_shape_ok(cin, cout) and (n_layers >= 2 or tiles >= 24 or long_ranges)


def _gang_shape_ok(cin, cout, fill=0.7) -> bool:
    """256 x 256 tiles mostly padding (128-channel layers): 0.3-0.5x of the per-layer kernel, whatever the group size.
    (k x k layers: 304 -> 256, 59 % filled, is still 1.45x.)"""
    tiles = (-(-cin // 256)) * (-(-cout // 256))
    return cin * cout >= fill * tiles * 65536


def wgrad_group_flush(key, only_sig=None):
    if only_sig is None:
        groups = _WG_GROUPS.pop(key, None)
    else:
        jobs_ = _WG_GROUPS.get(key, {}).pop(only_sig, None)
        groups = {only_sig: jobs_} if jobs_ else None
    if not groups:
        return
    dev = torch.device("cuda", key)
    any_side = _WG_ENABLED and L.PROFILE is None and any(sig[0] * max(sig[1], sig[2]) >= _WG_MIN_ELEMS for sig in groups)
    if any_side:
        ent = _WG_STREAMS.get(key)
        if ent is None:
            side = L.side_stream(dev, "wgrad")
            ent = _WG_STREAMS[key] = (side, side.cuda_stream)
        side, raw = ent
        side.wait_stream(torch.cuda.current_stream(dev))
    for (rows, cin, cout, ldx, ldy), jobs in groups.items():
        use_side = any_side and rows * max(cin, cout) >= _WG_MIN_ELEMS
        if use_side:
            for x, g, _, _ in jobs:
                x.record_stream(side)
                g.record_stream(side)
        if not _gang_pays(len(jobs), cin, cout, rows):
            for x, g, dwp, desc in jobs:
                name, args, keep = _wgrad_layer(desc, x, g, dwp)
                if use_side:
                    keep[-1].record_stream(side)
                    L.call_on(raw, name, *args)
                else:
                    L.call(name, *args)
            continue
        # the addresses are read on the host during the call and travel in the kernel arguments: no table in device memory
        tbl = torch.tensor([[x.data_ptr(), g.data_ptr(), dwp, 0] for x, g, dwp, _ in jobs], dtype=torch.int64)
        if use_side:
            L.call_on(raw, "bg_conv2d_bwd_weight_grouped", L.BF16, tbl.data_ptr(), len(jobs), rows, cin, cout, ldx, ldy)
        else:
            L.call("bg_conv2d_bwd_weight_grouped", L.BF16, tbl.data_ptr(), len(jobs), rows, cin, cout, ldx, ldy)


# ------------------------------------------------------------------ data parallel: early reduction of the arena's tail
# Parameters sit in the gradient arena in module order; backward visits them in reverse.  When the backward pass crosses
# a milestone placed in front of module M in forward order, every gradient of M and of everything registered after it is
# final (queued on this stream or on the weight-gradient stream) -- provided the arena's tail is exactly those modules.
# The milestone's backward then flushes the grouped weight gradients collected so far and starts the all-reduce of the
# tail on the reduction stream (comm.distributed.FlatAllReduce.launch_range) while the rest of the pass runs: the
# reference's apex DDP overlaps its buckets with backward the same way (comm/distributed.py:195-199).
_DDP_EARLY = _os.environ.get("BGAMD_DDP_EARLY", "1") != "0"


def arena_tail_offset(arena: Arena, first_tail_params) -> Optional[int]:
    """Arena offset from which on every slot belongs to a module registered at or after `first_tail_params`' owner, or
    None when the layout does not have that shape (then nothing is reduced early)."""
    ids = {id(p) for p in first_tail_params}
    offs = [s.off for s in arena.slots if id(s.param) in ids]
    if not offs:
        return None
    off0 = min(offs)
    seen_tail = False
    for s in arena.slots:                       # slots are in registration order with increasing offsets
        if s.off >= off0:
            seen_tail = True
        elif seen_tail:
            return None
    return off0


class GradMilestoneFn(torch.autograd.Function):
    """Identity.  backward: the gradient arena from `off` on is final -> start its all-reduce (data parallel only)."""

    @staticmethod
    def forward(ctx, x, probe, arena: Arena, off: int):
        # `probe` is one of the tail's parameters: with frozen weights (the G-step's critic; the gradient penalty, which
        # freezes the critic around its autograd.grad) grad_milestone() inserts no node, and needs_input_grad[1] is False
        # for a node created while the probe did not require a gradient
        ctx.arena, ctx.off = arena, off
        # a network may be called more than once before one backward pass (D(real) and D(fake)): the tail is final when the
        # LAST of these nodes has run, so they are counted (reset by FlatAllReduce.finish())
        arena.ddp.ms_pending += 1
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ddp = getattr(ctx.arena, "ddp", None)
        if ddp is not None:
            ddp.ms_pending -= 1
        if (ctx.needs_input_grad[1] and ddp is not None and ddp.world_size > 1 and ddp.ms_pending == 0
                and not (g.is_cuda and torch.cuda.is_current_stream_capturing())):
            after = ()
            if g.is_cuda:
                key = g.device.index if g.device.index is not None else torch.cuda.current_device()
                wgrad_group_flush(key)                        # grouped weight gradients of the tail: launch them now
                ent = _WG_STREAMS.get(key)
                after = (ent[0],) if ent is not None else ()
            ddp.launch_range(ctx.off, ctx.arena.numel, after)
        return g, None, None, None


def grad_milestone(x, arena: Arena, first_tail_params):
    """Mark the point in the forward pass behind which (in module order) the arena's tail lives; see GradMilestoneFn."""
    ddp = getattr(arena, "ddp", None)
    if (not _DDP_EARLY or ddp is None or ddp.world_size == 1 or not x.requires_grad
            or not first_tail_params or not first_tail_params[0].requires_grad):   # frozen weights (the G-step's critic): nothing to reduce
        return x
    cache = arena.__dict__.setdefault("_tail_offs", {})
    key = id(first_tail_params[0])
    if key not in cache:
        cache[key] = arena_tail_offset(arena, first_tail_params)
    off = cache[key]
    return x if off is None else GradMilestoneFn.apply(x, first_tail_params[0], arena, off)


# ------------------------------------------------------------------ layout boundary
# Module boundaries are NCHW fp32 like the reference's (Generator.forward -> Discriminator.forward, train_gan.py:252-254,
# 275-276).  Where a boundary tensor is handed straight from one of our modules to another, the NHWC compute-dtype bytes
# it was converted FROM still exist: FromInternal leaves a note on its result (the internal tensor, the tensor's version),
# and ToInternal / ToInternalCat take the internal bytes instead of converting the fp32 copy back (bf16 -> fp32 -> bf16 is
# the identity, so the values are the same bits).  The same note serves an input that is converted twice in one
# iteration (the generator's two forwards on one batch).  A tensor written after the note was made has a new version and
# is converted as usual.  BGAMD_NO_HANDOVER=1 disables.
_HANDOVER = _os.environ.get("BGAMD_NO_HANDOVER") is None


def _internal_of(x: torch.Tensor, cp: int, dtype: torch.dtype):
    """The NHWC tensor `x` (NCHW fp32) was converted from or to, if it is still valid for (cp, dtype)."""
    note = getattr(x, "_bg_internal", None) if _HANDOVER else None
    if note is None:
        return None
    t, version, ptr, epoch = note
    n, c, h, w = x.shape
    # (the epoch: a note never outlives the training iteration it was made in -- a loop that feeds the same tensor object
    # again, like a benchmark cycling over a few synthetic batches, converts it again, as a run on fresh batches would)
    if (version != x._version or ptr != x.data_ptr() or epoch != StatsPool.epoch or t.dtype != dtype
            or tuple(t.shape) != (n, h, w, cp)):
        return None
    return t


class ToInternal(torch.autograd.Function):
    """NCHW fp32 (module boundary, as the reference passes tensors) -> NHWC compute dtype."""

    @staticmethod
    def forward(ctx, x, cp: int, dtype: torch.dtype):
        if x.dtype != torch.float32:
            x = x.float()
        n, c, h, w = x.shape
        ctx.c = c
        src = _internal_of(x, cp, dtype)
        if src is not None:
            return src.detach()      # the same bytes under a fresh tensor (this node's output must not carry another node's history)
        y = new_act(n, h, w, cp, dtype, x.device)
        if _HANDOVER and not x.requires_grad:
            x._bg_internal = (y, x._version, x.data_ptr(), StatsPool.epoch)    # a second conversion of this batch (G's two forwards) reuses it
        xl = x.permute(0, 2, 3, 1)
        if c == cp and xl.is_contiguous() and x.data_ptr() % 16 == 0:
            # the caller's tensor is channels-last in memory (HWC files read by the staging ring, stacked: the layout
            # of cam_numpy_singlefile_dataset.py:96-99 before its permute): no transpose, one cast pass
            L.call("bg_cast_rows", L.F32, L.dt(dtype), x.data_ptr(), c, y.data_ptr(), ld_of(y), n * h * w, c)
            return y
        x = x.contiguous()
        L.call("bg_nchw_to_nhwc", L.dt(dtype), x.data_ptr(), y.data_ptr(), n, c, h * w, cp, ld_of(y))
        return y

    @staticmethod
    def backward(ctx, g):
        g = nhwc(g)
        n, h, w, cp = g.shape
        dx = torch.empty((n, ctx.c, h, w), dtype=torch.float32, device=g.device)
        L.call("bg_nhwc_to_nchw", L.dt(g.dtype), g.data_ptr(), ld_of(g), dx.data_ptr(), n, ctx.c, h * w)
        return dx, None, None


class ToInternalCat(torch.autograd.Function):
    """torch.cat(xs, dim=0) followed by ToInternal, without materialising the concatenated NCHW tensor: every input
    is converted straight into its batch slice of ONE NHWC buffer (the D-step's D(real) + D(fake) pass,
    train_gan.py:253-254 -- the cat alone moved 1.8 GB per step at 1152 x 768 x 16, batch 8)."""

    @staticmethod
    def forward(ctx, cp: int, dtype: torch.dtype, *xs):
        xs = [x if x.dtype == torch.float32 else x.float() for x in xs]
        c, h, w = xs[0].shape[1:]
        assert all(tuple(x.shape[1:]) == (c, h, w) for x in xs), [tuple(x.shape) for x in xs]
        ctx.c, ctx.ns = c, [x.shape[0] for x in xs]
        y = new_act(sum(ctx.ns), h, w, cp, dtype, xs[0].device)
        n0 = 0
        for x in xs:
            src = _internal_of(x, cp, dtype)
            if src is not None:      # handed over by one of our modules: copy its NHWC rows instead of transposing the fp32 copy
                L.call("bg_cast_rows", L.dt(dtype), L.dt(dtype), src.data_ptr(), ld_of(src), y[n0:].data_ptr(), ld_of(y),
                       x.shape[0] * h * w, cp)
            else:
                x = x.contiguous()
                L.call("bg_nchw_to_nhwc", L.dt(dtype), x.data_ptr(), y[n0:].data_ptr(), x.shape[0], c, h * w, cp, ld_of(y))
            n0 += x.shape[0]
        return y

    @staticmethod
    def backward(ctx, g):
        g = nhwc(g)
        _, h, w, _ = g.shape
        outs, n0 = [], 0
        for i, n in enumerate(ctx.ns):
            if ctx.needs_input_grad[2 + i]:
                dx = torch.empty((n, ctx.c, h, w), dtype=torch.float32, device=g.device)
                L.call("bg_nhwc_to_nchw", L.dt(g.dtype), g[n0:].data_ptr(), ld_of(g), dx.data_ptr(), n, ctx.c, h * w)
                outs.append(dx)
            else:
                outs.append(None)
            n0 += n
        return (None, None, *outs)


class FromInternal(torch.autograd.Function):
    """NHWC (any compute dtype) -> NCHW fp32 with the first `c` channels."""

    @staticmethod
    def forward(ctx, x, c: int):
        x = nhwc(x)
        n, h, w, cp = x.shape
        ctx.cp, ctx.dtype = cp, x.dtype
        y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        L.call("bg_nhwc_to_nchw", L.dt(x.dtype), x.data_ptr(), ld_of(x), y.data_ptr(), n, c, h * w)
        if _HANDOVER:   # (channels beyond c are zero lanes in every internal activation: DESIGN.md 1, test_parity_gpu "pad lanes")
            y._bg_internal = (x.detach(), y._version, y.data_ptr(), StatsPool.epoch)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        n, c, h, w = g.shape
        dx = new_act(n, h, w, ctx.cp, ctx.dtype, g.device)
        L.call("bg_nchw_to_nhwc", L.dt(ctx.dtype), g.data_ptr(), dx.data_ptr(), n, c, h * w, ctx.cp, ld_of(dx))
        return dx, None


# ------------------------------------------------------------------------- convolution
_SPLITK = not _os.environ.get("BGAMD_NO_SPLITK")


def splitk_plan(m_out, n_out, cin, kh, kw, dtype):
    """Split count for a GEMM launch with few output tiles and a long reduction (0: run it as one launch).  Units as the
    library's: 128 x 128 tiles, K-steps of 64 bytes of channels per tap.  The count divides the K-steps into non-empty
    ranges of ceil(K-steps / splits), as the entry points require."""
    if not _SPLITK:
        return 0
    tiles = -(-m_out // 128) * -(-n_out // 128)
    ksteps = kh * kw * -(-cin // (32 if dtype == torch.bfloat16 else 16))
    if tiles > 48 or ksteps < 48:
        return 0
    splits = min(ksteps // 12, 768 // tiles)
    if splits < 4:
        return 0
    per = -(-ksteps // splits)
    return -(-ksteps // per)


def _splitk_finish(ws, y):
    """fp32 [splits, rows, C] partial tiles -> the activation tensor: slices summed in index order, rounded to its dtype."""
    splits, rows, c = ws.shape
    L.call("bg_splitk_reduce", L.dt(y.dtype), ws.data_ptr(), splits, rows, c, y.data_ptr(), ld_of(y))


# ---- fp8 operand path (BASELINE.json configs[4]; include/bgamd.h "fp8 operand path") --------------------------------
_FP8_MIN_WORK = int(_os.environ.get("BGAMD_FP8_MIN_WORK", "1500"))


def set_fp8_min_work(v: int) -> int:
    """Threshold of fp8_layer_ok (taps x channels of the other GEMM side from which a layer's operands are quantised even
    without a producer-written copy); 0 puts EVERY eligible layer on fp8 operands.  Returns the previous value."""
    global _FP8_MIN_WORK
    old, _FP8_MIN_WORK = _FP8_MIN_WORK, int(v)
    return old


def fp8_layer_ok(arena: Arena, wslot: ParamSlot, kh, kw, grad: bool = False, prequantised: bool = False) -> bool:
    """Which dense convolutions take fp8 operands.  The operand must first be quantised (3 bytes per element moved by
    bg_quant_fp8 unless the producer wrote the fp8 copy itself); the GEMM then saves ~35 % of its bf16 time.  Per element
    of the operand the saving grows with the taps x channels on the OTHER side of the GEMM (Cout for the forward pass,
    Cin for the data gradient): measured break-even at ~900 (a 728 -> 728 pointwise layer: 49 us saved, 53 us of
    quantisation, scripts/bench_fp8.py), so layers below BGAMD_FP8_MIN_WORK = 1500 stay on bf16 operands unless the fp8
    copy already exists.  Reductions below 256 or fewer than 64 channels on either side: never (HBM-bound copies)."""
    if not getattr(arena, "fp8", False) or wslot.f8 is None:
        return False
    kp, _, _, cp = wslot.phys_shape
    if not (cp * kh * kw >= 256 and kp >= 64 and cp >= 64):
        return False
    return prequantised or kh * kw * (cp if grad else kp) >= _FP8_MIN_WORK


def fp8_quant(t: torch.Tensor, fmt: int, exp_ptr: int, amax_ptr: int) -> torch.Tensor:
    """fp8 copy of an NHWC bf16 tensor: [N, H, W, Cq] bytes, Cq = C rounded up to 16 (zero lanes), pixel stride rounded
    up to 64 bytes; quantised with the site's exponent, the site's running max |value| updated."""
    n, h, w, c = t.shape
    cq = (c + 15) // 16 * 16
    ldq = (cq + 63) // 64 * 64
    q = torch.empty((n, h, w, ldq), dtype=torch.uint8, device=t.device)
    L.call("bg_quant_fp8", L.dt(t.dtype), t.data_ptr(), ld_of(t), n * h * w, c, q.data_ptr(), ldq, cq, fmt, exp_ptr, amax_ptr)
    return q[..., :cq]


def fp8_amax_only(t: torch.Tensor, fmt: int, exp_ptr: int, amax_ptr: int):
    """Calibration step: only the site's max |value| is wanted (the GEMM runs on the bf16 operand)."""
    fp8_quant(t, fmt, exp_ptr, amax_ptr)


def fp8_copy_of(t: torch.Tensor):
    """The producer-written fp8 copy attached to `t` (t._bg_fp8), or None -- also None when `t` has been written since the
    copy was made (the autograd engine accumulating a second gradient into the same tensor in place, a tensor hook): the
    attribute survives such a write, the copy would be stale, so the consumer quantises again."""
    q = getattr(t, "_bg_fp8", None)
    if q is None:
        return None
    if len(q) >= 4 and (q[2] != t._version or q[3] != t.data_ptr()):
        return None
    return q


def fp8_tag_output(y: torch.Tensor, arena: Arena, wslot: ParamSlot, kh, kw):
    """Mark a dense convolution's output with the quantisation site of its OUTPUT GRADIENT: the BatchNorm backward that
    later produces that gradient writes the e5m2 copy itself (bg_norm_act_bwd_apply_stats_q8) instead of leaving a
    quantisation pass to the convolution's backward."""
    if fp8_layer_ok(arena, wslot, kh, kw, grad=True, prequantised=True) and wslot.phys_shape[3] >= 64:
        y._bg_dy_site = (arena, wslot)


def _apply_stats_maybe_q8(q_site, dx, args, n, h, w, c):
    """bg_norm_act_bwd_apply_stats, or its _q8 form when the consumer of dx is an fp8 data-gradient GEMM whose site is
    calibrated: dx then carries its e5m2 copy (dx._bg_fp8)."""
    if q_site is not None and dx is not None and dx.dtype == torch.bfloat16:
        qa, qw = q_site
        ep, ap = qa.site_ptrs(qw, grad=True)
        if qa.site_ready(qw, grad=True):
            cq = (c + 15) // 16 * 16
            ldq = (cq + 63) // 64 * 64
            dxq = torch.empty((n, h, w, ldq), dtype=torch.uint8, device=dx.device)
            L.call("bg_norm_act_bwd_apply_stats_q8", *args, dxq.data_ptr(), ldq, ep, ap)
            qa.site_seen(qw, grad=True)
            dx._bg_fp8 = (dxq[..., :cq], ep, dx._version, dx.data_ptr())   # fp8_copy_of() checks that dx is still these bytes
            return
        # not calibrated yet: the plain kernel, then one pass that only records the site's max |dx| (the consuming
        # convolution would not take it for layers below the stand-alone threshold)
        L.call("bg_norm_act_bwd_apply_stats", *args)
        fp8_amax_only(dx, L.FP8_E5M2, ep, ap)
        qa.site_seen(qw, grad=True)
        return
    L.call("bg_norm_act_bwd_apply_stats", *args)


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d(groups=1) as implicit GEMM (bg_conv2d_*)."""

    @staticmethod
    def forward(ctx, x, weight, bias, arena: Arena, wslot: ParamSlot, bslot: Optional[ParamSlot], stride, pad, dil,
                stats=None):
        """stats: optional zeroed fp64 [2, Cout_phys]; the kernel adds sum(y), sum(y^2) of the stored
        outputs to it from its accumulators (the batch statistics of the BatchNorm that follows)."""
        x = nhwc(x)
        n, h, w, cin = x.shape
        kp, kh, kw, cp = wslot.phys_shape
        assert cin == cp, f"conv expects {cp} input channels (padded), got {cin}"
        ho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        wo = (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        y = new_act(n, ho, wo, kp, x.dtype, x.device)
        desc = L.ConvDesc(L.dt(x.dtype), n, h, w, cin, ho, wo, kp, kh, kw, stride, pad, dil, ld_of(x), ld_of(y))
        splits = splitk_plan(n * ho * wo, kp, cin, kh, kw, x.dtype) if bslot is None else 0
        xq = fp8_copy_of(x)
        use8 = x.dtype == torch.bfloat16 and not splits and fp8_layer_ok(arena, wslot, kh, kw, prequantised=xq is not None)
        if use8:
            # fp8 operands: the input's e4m3 copy (the producer's, if it made one; else one quantisation pass) against the
            # arena's e4m3 weights; bf16 output and statistics as below.  Until the sites have been calibrated (the first
            # training step) the GEMM stays on bf16 and the quantiser only records the input's max |value|.
            ep, ap = arena.site_ptrs(wslot, grad=False)
            ready = arena.site_ready(wslot, grad=False)
            if xq is None or xq[1] != ep or not ready:
                xq = (fp8_quant(x, L.FP8_E4M3, ep, ap), ep)
                arena.site_seen(wslot, grad=False)
            use8 = ready
        if use8:
            xq = xq[0]
            d8 = L.ConvDesc(L.BF16, n, h, w, xq.shape[3], ho, wo, kp, kh, kw, stride, pad, dil, ld_of(xq), ld_of(y))
            if stats is not None:
                assert bslot is None and stats.dim() == 3 and stats.shape[0] == 2 and stats.shape[2] == kp
            L.call("bg_conv2d_fwd_fp8", d8, xq.data_ptr(), arena.weight8_ptr(wslot), ep, arena.w_exp_ptr(wslot),
                   None if bslot is None else arena.master_ptr(bslot), y.data_ptr(),
                   None if stats is None else stats[0].data_ptr(), None if stats is None else stats[1].data_ptr(),
                   1 if stats is None else stats.shape[1])
        elif splits:
            ws = torch.empty((splits, n * ho * wo, kp), dtype=torch.float32, device=x.device)
            L.call("bg_conv2d_fwd_splitk", desc, x.data_ptr(), arena.weight_ptr(wslot), ws.data_ptr(), splits)
            _splitk_finish(ws, y)
            if stats is not None:   # statistics of the values as stored, like the fused epilogue
                L.call("bg_norm_stats", L.dt(y.dtype), y.data_ptr(), n * ho * wo, kp, ld_of(y), stats.shape[1],
                       stats[0].data_ptr(), stats[1].data_ptr())
        elif stats is not None:
            # stats: fp64 [2, groups, kp]; groups = sub-batches with separate BatchNorm statistics
            assert bslot is None and stats.dim() == 3 and stats.shape[0] == 2 and stats.shape[2] == kp
            assert stats.dtype == torch.float64
            L.call("bg_conv2d_fwd_stats", desc, x.data_ptr(), arena.weight_ptr(wslot), y.data_ptr(), stats[0].data_ptr(),
                   stats[1].data_ptr(), stats.shape[1])
        else:
            L.call("bg_conv2d_fwd", desc, x.data_ptr(), arena.weight_ptr(wslot),
                   None if bslot is None else arena.master_ptr(bslot), y.data_ptr())
        # the input is only needed for the weight gradient: do not keep it alive otherwise
        # (G-step: the discriminator is differentiated w.r.t. its input only)
        if weight.requires_grad:
            ctx.save_for_backward(x)
        ctx.meta = (arena, wslot, bslot, stride, pad, dil, ho, wo, tuple(x.shape), x.dtype, x.device)
        return y

    @staticmethod
    def backward(ctx, g):
        arena, wslot, bslot, stride, pad, dil, ho, wo, xshape, xdtype, xdev = ctx.meta
        g = nhwc(g)
        n, h, w, cin = xshape
        kp, kh, kw, cp = wslot.phys_shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new_act(n, h, w, cin, xdtype, xdev)
            d2 = L.ConvDesc(L.dt(xdtype), n, h, w, cin, ho, wo, kp, kh, kw, stride, pad, dil, ld_of(dx), ld_of(g))
            splits = splitk_plan(n * h * w, cin, kp, kh, kw, xdtype)
            gq = fp8_copy_of(g)
            use8 = (xdtype == torch.bfloat16 and not splits and cin >= 64
                    and fp8_layer_ok(arena, wslot, kh, kw, grad=True, prequantised=gq is not None))
            if use8:
                ep, ap = arena.site_ptrs(wslot, grad=True)
                ready = arena.site_ready(wslot, grad=True)
                if gq is None or gq[1] != ep or not ready:
                    gq = (fp8_quant(g, L.FP8_E5M2, ep, ap), ep)
                    arena.site_seen(wslot, grad=True)
                use8 = ready
            if use8:
                gq = gq[0]
                d8 = L.ConvDesc(L.BF16, n, h, w, cin, ho, wo, gq.shape[3], kh, kw, stride, pad, dil, ld_of(dx), ld_of(gq))
                L.call("bg_conv2d_bwd_data_fp8", d8, gq.data_ptr(), L.FP8_E5M2, arena.weight8_t_ptr(wslot), ep,
                       arena.w_exp_ptr(wslot), dx.data_ptr())
            elif splits:
                ws = torch.empty((splits, n * h * w, cin), dtype=torch.float32, device=xdev)
                L.call("bg_conv2d_bwd_data_splitk", d2, g.data_ptr(), arena.weight_t_ptr(wslot), ws.data_ptr(), splits)
                _splitk_finish(ws, dx)
            else:
                L.call("bg_conv2d_bwd_data", d2, g.data_ptr(), arena.weight_t_ptr(wslot), dx.data_ptr())
        if ctx.needs_input_grad[1]:
            (x,) = ctx.saved_tensors
            desc = L.ConvDesc(L.dt(xdtype), n, h, w, cin, ho, wo, kp, kh, kw, stride, pad, dil, ld_of(x), ld_of(g))
            arena.ensure_grad(wslot)
            dbias = None
            if bslot is not None and ctx.needs_input_grad[2]:
                arena.ensure_grad(bslot)
                dbias = arena.grad_ptr(bslot)
            if wgrad_group_ok(xdtype, kh, kw, stride, pad, dil, dbias is not None) and x.shape[3] == cin and g.shape[3] == kp:
                wgrad_group_add(xdev, x, g, arena.grad_ptr(wslot), n * h * w, cin, kp, desc)
            elif wgrad_taps_ok(xdtype, n, h, w, cin, kp, kh, kw, stride, pad, dil, dbias is not None, ld_of(x), ld_of(g)):
                # a k x k layer as k*k pointwise weight gradients against shifted x in the gang kernel; issued at once (a
                # single layer: nothing to group), on the side stream.  The one-row table is host memory read during the call.
                tbl = torch.tensor([[x.data_ptr(), g.data_ptr(), arena.grad_ptr(wslot), 0]], dtype=torch.int64)
                wgrad_call(xdev, (x, g), "bg_conv2d_bwd_weight_grouped_taps", desc, tbl.data_ptr(), 1)
            else:
                name, args, keep = _wgrad_layer(desc, x, g, arena.grad_ptr(wslot), dbias)
                wgrad_call(xdev, keep, name, *args)
        return dx, None, None, None, None, None, None, None, None, None


class ConvTranspose2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d(groups=1, bias=False) of the Deconv upsamplers (deeplab.py:406-431).

    A transposed convolution IS the data gradient of the convolution with the same weight tensor
    ([Cin_t, Cout_t, k, k] = that convolution's [Cout, Cin, k, k]), so the three GEMM entry points
    are used with their roles exchanged: forward = bg_conv2d_bwd_data, input gradient =
    bg_conv2d_fwd, weight gradient = bg_conv2d_bwd_weight(x := dy, dy := x)."""

    @staticmethod
    def forward(ctx, x, weight, arena: Arena, wslot: ParamSlot, stride, pad, out_pad):
        x = nhwc(x)
        n, h, w, cin = x.shape
        kp, kh, kw, cp = wslot.phys_shape            # K = Cin_t (padded), C = Cout_t (padded)
        assert cin == kp, f"transposed conv expects {kp} input channels (padded), got {cin}"
        ho = (h - 1) * stride - 2 * pad + kh + out_pad[0]
        wo = (w - 1) * stride - 2 * pad + kw + out_pad[1]
        assert (ho + 2 * pad - kh) // stride + 1 == h and (wo + 2 * pad - kw) // stride + 1 == w, "bad output_padding"
        y = new_act(n, ho, wo, cp, x.dtype, x.device)
        desc = L.ConvDesc(L.dt(x.dtype), n, ho, wo, cp, h, w, kp, kh, kw, stride, pad, 1, ld_of(y), ld_of(x))
        L.call("bg_conv2d_bwd_data", desc, x.data_ptr(), arena.weight_t_ptr(wslot), y.data_ptr())
        if weight.requires_grad:
            ctx.save_for_backward(x)
        ctx.meta = (arena, wslot, stride, pad, ho, wo, tuple(x.shape), x.dtype, x.device)
        return y

    @staticmethod
    def backward(ctx, g):
        arena, wslot, stride, pad, ho, wo, xshape, xdtype, xdev = ctx.meta
        g = nhwc(g)
        n, h, w, cin = xshape
        kp, kh, kw, cp = wslot.phys_shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new_act(n, h, w, kp, xdtype, xdev)
            desc = L.ConvDesc(L.dt(xdtype), n, ho, wo, cp, h, w, kp, kh, kw, stride, pad, 1, ld_of(g), ld_of(dx))
            L.call("bg_conv2d_fwd", desc, g.data_ptr(), arena.weight_ptr(wslot), None, dx.data_ptr())
        if ctx.needs_input_grad[1]:
            (x,) = ctx.saved_tensors
            arena.ensure_grad(wslot)
            desc = L.ConvDesc(L.dt(xdtype), n, ho, wo, cp, h, w, kp, kh, kw, stride, pad, 1, ld_of(g), ld_of(x))
            name, args, keep = _wgrad_layer(desc, x, g, arena.grad_ptr(wslot), None, swap=True)
            wgrad_call(xdev, keep, name, *args)
        return dx, None, None, None, None, None, None


class AvgPool2x2Fn(torch.autograd.Function):
    """nn.AvgPool2d(2, stride=1, padding=p), p in {0, 1} (deeplab.py:402,469,649)."""

    @staticmethod
    def forward(ctx, x, p: int):
        x = nhwc(x)
        n, h, w, c = x.shape
        ho, wo = h + 2 * p - 1, w + 2 * p - 1
        y = new_act(n, ho, wo, c, x.dtype, x.device)
        L.call("bg_avgpool2x2", L.dt(x.dtype), x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), n, h, w, ho, wo, c, -p)
        ctx.meta = (n, h, w, ho, wo, c, p)
        return y

    @staticmethod
    def backward(ctx, g):
        n, h, w, ho, wo, c, p = ctx.meta
        g = nhwc(g)
        dx = new_act(n, h, w, c, g.dtype, g.device)
        L.call("bg_avgpool2x2", L.dt(g.dtype), g.data_ptr(), ld_of(g), dx.data_ptr(), ld_of(dx), n, ho, wo, h, w, c, p - 1)
        return dx, None


def avgpool2x2(x, p: int):
    return AvgPool2x2Fn.apply(x, int(p))


class DwConv3x3Fn(torch.autograd.Function):
    """Depthwise 3x3 of SeparableConv2d_same with fixed_padding folded in (bg_dwconv3x3_*)."""

    @staticmethod
    def forward(ctx, x, weight, arena: Arena, wslot: ParamSlot, stride, dil):
        x = nhwc(x)
        n, h, w, c = x.shape
        assert wslot.phys_shape == (3, 3, c), (wslot.phys_shape, c)
        ho, wo = -(-h // stride), -(-w // stride)
        y = new_act(n, ho, wo, c, x.dtype, x.device)
        desc = L.DwDesc(L.dt(x.dtype), n, h, w, c, ho, wo, stride, dil, ld_of(x), ld_of(y))
        L.call("bg_dwconv3x3_fwd", desc, x.data_ptr(), arena.weight_ptr(wslot), y.data_ptr())
        if weight.requires_grad:
            ctx.save_for_backward(x)
        ctx.meta = (arena, wslot, stride, dil, ho, wo, tuple(x.shape), x.dtype, x.device)
        return y

    @staticmethod
    def backward(ctx, g):
        arena, wslot, stride, dil, ho, wo, xshape, xdtype, xdev = ctx.meta
        g = nhwc(g)
        n, h, w, c = xshape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new_act(n, h, w, c, xdtype, xdev)
            desc = L.DwDesc(L.dt(xdtype), n, h, w, c, ho, wo, stride, dil, ld_of(dx), ld_of(g))
            L.call("bg_dwconv3x3_bwd_data", desc, g.data_ptr(), arena.weight_ptr(wslot), dx.data_ptr())
        if ctx.needs_input_grad[1]:
            (x,) = ctx.saved_tensors
            arena.ensure_grad(wslot)
            desc = L.DwDesc(L.dt(xdtype), n, h, w, c, ho, wo, stride, dil, ld_of(x), ld_of(g))
            wgrad_call(xdev, (x, g), "bg_dwconv3x3_bwd_weight", desc, x.data_ptr(), g.data_ptr(), arena.grad_ptr(wslot))
        return dx, None, None, None, None, None


_FORK_DW = _os.environ.get("BGAMD_NO_FORK_DW") is None and _os.environ.get("BGAMD_DW_RING", "1") != "0"   # A/B switch


class ForkDwConv3x3Fn(torch.autograd.Function):
    """(dw3x3(x), x): a Block's input feeds its first depthwise convolution AND its skip path (deeplab.py:134-141).
    Forward is DwConv3x3Fn plus an alias of x; backward forms the input gradient dwT(g_main) + g_skip in the depthwise
    data-gradient kernel itself (bg_dwconv3x3_bwd_data_add) instead of a separate add pass over three tensors.
    Stride 1, dilation 1 or 2."""

    @staticmethod
    def forward(ctx, x, weight, arena: Arena, wslot: ParamSlot, dil, tail=None):
        """tail: the ops.NormTail of the node that produced x (it offers the first half of its backward), or None."""
        x = nhwc(x)
        n, h, w, c = x.shape
        assert wslot.phys_shape == (3, 3, c), (wslot.phys_shape, c)
        y = new_act(n, h, w, c, x.dtype, x.device)
        desc = L.DwDesc(L.dt(x.dtype), n, h, w, c, h, w, 1, dil, ld_of(x), ld_of(y))
        L.call("bg_dwconv3x3_fwd", desc, x.data_ptr(), arena.weight_ptr(wslot), y.data_ptr())
        ctx.tail = None
        if (tail is not None and _FORK_FUSED and dil == 1 and x.dtype == torch.bfloat16 and ctx.needs_input_grad[0]
                and tail.y_ptr == x.data_ptr() and h * w * max(ld_of(x), ld_of(y), ld_of(tail.x)) * 2 < (1 << 31)):
            ctx.tail = tail
            tail.claimed += 1
        if weight.requires_grad or ctx.tail is not None:
            ctx.save_for_backward(x)
        ctx.meta = (arena, wslot, dil, tuple(x.shape), x.dtype, x.device)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, g, gskip):
        arena, wslot, dil, xshape, xdtype, xdev = ctx.meta
        n, h, w, c = xshape
        dt = L.dt(xdtype)
        if g is not None:
            g = nhwc(g)
        if gskip is not None:
            gskip = nhwc(gskip)
        dx = None
        tail = ctx.tail
        if tail is not None and tail.claimed == 1 and ctx.needs_input_grad[0] and g is not None and gskip is not None:
            # ONE pass: depthwise data gradient + skip gradient, the producer's activation derivative, the depthwise weight
            # gradient and the producer's two BatchNorm-backward sums (bg_dwconv3x3_bwd_fork); the producer's backward finds
            # the sums in its tail and treats the gradient as already activated
            (x,) = ctx.saved_tensors
            want_dw = ctx.needs_input_grad[1]
            if want_dw:
                arena.ensure_grad(wslot)
            gout = new_act(n, h, w, c, xdtype, xdev)
            sums = _f64(2, tail.groups, c, device=xdev)
            desc = L.DwDesc(dt, n, h, w, c, h, w, 1, dil, ld_of(x), ld_of(g))
            L.call("bg_dwconv3x3_bwd_fork", desc, g.data_ptr(), arena.weight_ptr(wslot), x.data_ptr(), gskip.data_ptr(),
                   ld_of(gskip), tail.x.data_ptr(), ld_of(tail.x), tail.mean.data_ptr(), tail.rstd.data_ptr(), tail.groups,
                   tail.act, gout.data_ptr(), ld_of(gout), arena.grad_ptr(wslot) if want_dw else None, sums[0].data_ptr(),
                   sums[1].data_ptr())
            tail.sums, tail.gout = sums, gout
            return gout, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            if g is None:
                dx = gskip
            else:
                dx = new_act(n, h, w, c, xdtype, xdev)
                desc = L.DwDesc(dt, n, h, w, c, h, w, 1, dil, ld_of(dx), ld_of(g))
                if gskip is None:
                    L.call("bg_dwconv3x3_bwd_data", desc, g.data_ptr(), arena.weight_ptr(wslot), dx.data_ptr())
                else:
                    L.call("bg_dwconv3x3_bwd_data_add", desc, g.data_ptr(), arena.weight_ptr(wslot), gskip.data_ptr(),
                           ld_of(gskip), dx.data_ptr())
        if ctx.needs_input_grad[1] and g is not None:
            (x,) = ctx.saved_tensors
            arena.ensure_grad(wslot)
            desc = L.DwDesc(dt, n, h, w, c, h, w, 1, dil, ld_of(x), ld_of(g))
            wgrad_call(xdev, (x, g), "bg_dwconv3x3_bwd_weight", desc, x.data_ptr(), g.data_ptr(), arena.grad_ptr(wslot))
        return dx, None, None, None, None, None


class NormTail:
    """What a [BatchNorm (batch statistics) -> (+ residual) -> activation] node offers the SOLE consumer of its output so
    that the consumer's backward can also do the first half of this node's backward (ForkDwConv3x3Fn / bg_dwconv3x3_bwd_fork):
    the BatchNorm's input and saved statistics going out; the two sums and the already-activated gradient coming back.
    The producer only attaches it where the caller guarantees a single consumer (Block.forward(sole_consumer=True):
    Xception chains its Blocks that way) -- a second consumer's gradient would be added to an activated one."""
    __slots__ = ("x", "mean", "rstd", "groups", "act", "y_ptr", "claimed", "sums", "gout")

    def __init__(self, x, mean, rstd, groups, act, y):
        self.x, self.mean, self.rstd, self.groups, self.act, self.y_ptr = x, mean, rstd, groups, act, y.data_ptr()
        self.claimed, self.sums, self.gout = 0, None, None


_FORK_FUSED = _os.environ.get("BGAMD_FORK_FUSED", "1") != "0"   # A/B switch: 0 = data_add + weight gradient + reduce as separate launches


def fork_dw_ok(conv1) -> bool:
    """Can ForkDwConv3x3Fn run this depthwise convolution (stride 1, dilation 1 or 2, no bias)?"""
    return _FORK_DW and conv1.stride[0] == 1 and conv1.dilation[0] in (1, 2) and conv1.bias is None


# ------------------------------------------------------- norm + residual + LeakyReLU
_BN_GROUPS = [1]


class batch_groups:
    """Context: BatchNorm layers (training mode) take their statistics separately over `groups` equal
    sub-batches of the batch they see, and apply one running-statistics update per sub-batch, in order --
    exactly what the reference computes when it pushes those sub-batches through the network in separate
    calls (D(real) then D(fake), train_gan.py:253-254), but in one pass over twice the rows."""

    def __init__(self, groups: int):
        self.groups = int(groups)

    def __enter__(self):
        _BN_GROUPS.append(self.groups)
        return self

    def __exit__(self, *exc):
        _BN_GROUPS.pop()
        return False


def current_bn_groups() -> int:
    return _BN_GROUPS[-1]


_BN_REPEAT = [1]


class bn_repeat:
    """Context: a training-mode BatchNorm forward counts as `k` identical forwards for its running statistics -- k momentum
    updates with the same batch statistics, r <- (1-m)^k r + (1 - (1-m)^k) mu, and k counts of num_batches_tracked.  Used
    when a trainer evaluates the generator once where the reference's loop evaluates it twice on the same input with
    unchanged weights (train_gan.py:252,275; GANTrainer.reuse_g_forward)."""

    def __init__(self, k: int):
        self.k = int(k)

    def __enter__(self):
        _BN_REPEAT.append(self.k)
        return self

    def __exit__(self, *exc):
        _BN_REPEAT.pop()
        return False


def current_bn_repeat() -> int:
    return _BN_REPEAT[-1]



class NormActFn(torch.autograd.Function):
    """y = act( norm(x) + res ).

    kind: 'batch' (nn.BatchNorm2d: batch statistics when training, running
    statistics in eval; running stats updated in place), 'instance'
    (nn.InstanceNorm2d defaults), 'identity'.  act: LeakyReLU(0.2) or none.
    """

    @staticmethod
    def forward(ctx, x, res, gamma, beta, arena, gslot, bslot, rmean, rvar, kind, training, act, eps, momentum,
                pre_stats=None, bn_groups=1, inst_groups=None, offer_tail=False):
        """bn_groups: 'batch' statistics are taken separately over that many equal sub-batches (see
        batch_groups()).  pre_stats: fp64 [2, groups, C] sums already produced by the convolution's epilogue
        (bg_conv2d_fwd_stats); skips the separate statistics pass."""
        ctx.q_site = getattr(x, "_bg_dy_site", None)
        x = nhwc(x)
        n, h, w, c = x.shape
        dev, dt = x.device, L.dt(x.dtype)
        rows = n * h * w
        if res is not None:
            res = nhwc(res)
        use_batch_stats = (kind == "batch" and training) or kind == "instance"
        # instance statistics: one group per sample; a volume folded into the batch passes its sample count
        groups = (inst_groups or n) if kind == "instance" else (bn_groups if kind == "batch" and training else 1)
        assert n % groups == 0, f"batch of {n} does not split into {groups} statistic groups"
        mean = rstd = scale = shift = None
        y = new_act(n, h, w, c, x.dtype, dev)
        gptr = None if gslot is None else arena.master_ptr(gslot)
        bptr = None if bslot is None else arena.master_ptr(bslot)
        if kind != "identity" and use_batch_stats:
            if pre_stats is not None and pre_stats.shape[1] == groups:
                s = pre_stats
            else:
                s = _f64(2, groups, c, device=dev)
                L.call("bg_norm_stats", dt, x.data_ptr(), rows, c, ld_of(x), groups, s[0].data_ptr(), s[1].data_ptr())
            mean, rstd = _e32(2, groups, c, device=dev).unbind(0)
            upd = kind == "batch" and rmean is not None
            # finalize (mean/rstd, affine, running statistics) is folded into the apply kernel
            L.call("bg_norm_act_fwd_stats", dt, x.data_ptr(), ld_of(x), s[0].data_ptr(), s[1].data_ptr(), gptr, bptr, eps,
                   momentum, rmean.data_ptr() if upd else None, rvar.data_ptr() if upd else None, mean.data_ptr(),
                   rstd.data_ptr(), L.ptr(res), 0 if res is None else ld_of(res), y.data_ptr(), ld_of(y), rows, c, groups, int(act))
        else:
            if kind != "identity":  # BatchNorm in eval mode: affine from the running statistics
                mean, rstd, scale, shift = _e32(4, groups, c, device=dev).unbind(0)
                L.call("bg_norm_eval_affine", c, gptr, bptr, rmean.data_ptr(), rvar.data_ptr(), eps, scale.data_ptr(),
                       shift.data_ptr())
                mean.copy_(rmean.view(1, -1))
                rstd.copy_(torch.rsqrt(rvar + eps).view(1, -1))
            L.call("bg_norm_act_fwd", dt, x.data_ptr(), ld_of(x), L.ptr(scale), L.ptr(shift), L.ptr(res),
                   0 if res is None else ld_of(res), y.data_ptr(), ld_of(y), rows, c, groups, int(act))
        ctx.save_for_backward(x, y, mean, rstd)
        ctx.meta = (arena, gslot, bslot, kind, use_batch_stats, int(act), groups, res is not None)
        ctx.tail = None
        if (offer_tail and _FORK_FUSED and kind == "batch" and use_batch_stats and x.dtype == torch.bfloat16
                and ctx.needs_input_grad[0]):
            ctx.tail = y._bg_tail = NormTail(x, mean, rstd, groups, int(act), y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, mean, rstd = ctx.saved_tensors
        arena, gslot, bslot, kind, batch_stats, act, groups, has_res = ctx.meta
        g = nhwc(g)
        n, h, w, c = x.shape
        dev, dt = x.device, L.dt(x.dtype)
        rows = n * h * w
        need_dx = ctx.needs_input_grad[0]
        need_res = has_res and ctx.needs_input_grad[1]
        tail = ctx.tail
        if tail is not None and tail.sums is not None:
            # the consumer's backward (bg_dwconv3x3_bwd_fork) already multiplied the gradient by act'(y) and accumulated
            # sum g, sum g * xhat: no reduce pass, the apply pass runs without an activation, the residual gradient IS g
            if g.data_ptr() != tail.gout.data_ptr():
                raise RuntimeError("bias_gan_amd: a fused fork backward needs its producer's output to have ONE consumer "
                                   "(the incoming gradient is not the tensor the fork's backward wrote)")
            s, tail.sums, tail.gout = tail.sums, None, None
            dx = new_act(n, h, w, c, x.dtype, dev) if need_dx else None
            want_affine_grads = gslot is not None and gslot.param.requires_grad
            if need_dx or want_affine_grads:
                dg = db = None
                if want_affine_grads:
                    arena.ensure_grad(gslot)
                    arena.ensure_grad(bslot)
                    dg, db = arena.grad_ptr(gslot), arena.grad_ptr(bslot)
                args = (dt, g.data_ptr(), ld_of(g), None, 0, x.data_ptr(), ld_of(x),
                        s[0].data_ptr(), s[1].data_ptr(), arena.master_ptr(gslot), arena.master_ptr(bslot), mean.data_ptr(),
                        rstd.data_ptr(), 1, dg, db, L.ptr(dx), 0 if dx is None else ld_of(dx), None, 0, rows, c, groups, 0)
                # fp8 operand path: dx still leaves with its e5m2 copy for the data-gradient GEMM of the convolution that made x
                _apply_stats_maybe_q8(ctx.q_site, dx, args, n, h, w, c)
            return (dx, g if need_res else None) + (None,) * 16
        dx = new_act(n, h, w, c, x.dtype, dev) if need_dx else None
        dres = new_act(n, h, w, c, x.dtype, dev) if need_res else None
        if kind == "identity":
            if dx is None and dres is None:
                return (None,) * 18
            # dx and dres are the same tensor values: write once, alias
            out = dx if dx is not None else dres
            L.call("bg_norm_act_bwd_apply", dt, g.data_ptr(), ld_of(g), y.data_ptr(), ld_of(y), None, 0, None, None, None,
                   out.data_ptr(), ld_of(out), None, 0, rows, c, groups, act)
            return (out if need_dx else None, out if need_res else None) + (None,) * 16
        want_affine_grads = gslot is not None and gslot.param.requires_grad
        gptr = None if gslot is None else arena.master_ptr(gslot)
        bptr = None if bslot is None else arena.master_ptr(bslot)
        # without a residual the LeakyReLU branch follows from x and the saved statistics: the
        # backward kernels recompute it (the forward kernel's arithmetic) instead of re-reading y
        yptr = None if (batch_stats and not has_res) else y.data_ptr()
        if need_dx or want_affine_grads:
            s = _f64(2, groups, c, device=dev)
            L.call("bg_norm_act_bwd_reduce", dt, g.data_ptr(), ld_of(g), yptr, ld_of(y), x.data_ptr(), ld_of(x),
                   mean.data_ptr(), rstd.data_ptr(), gptr, bptr, rows, c, groups, act, s[0].data_ptr(), s[1].data_ptr())
            dg = db = None
            if want_affine_grads:
                arena.ensure_grad(gslot)
                arena.ensure_grad(bslot)
                dg, db = arena.grad_ptr(gslot), arena.grad_ptr(bslot)
            # finalize (coefficients, dgamma/dbeta) is folded into the apply kernel
            args = (dt, g.data_ptr(), ld_of(g), yptr, ld_of(y), x.data_ptr(), ld_of(x),
                    s[0].data_ptr(), s[1].data_ptr(), gptr, bptr, mean.data_ptr(), rstd.data_ptr(),
                    1 if batch_stats else 0, dg, db, L.ptr(dx), 0 if dx is None else ld_of(dx), L.ptr(dres),
                    0 if dres is None else ld_of(dres), rows, c, groups, act)
            _apply_stats_maybe_q8(ctx.q_site if batch_stats else None, dx, args, *x.shape)
        elif need_res:
            L.call("bg_norm_act_bwd_apply", dt, g.data_ptr(), ld_of(g), y.data_ptr(), ld_of(y), None, 0, None, None, None, None,
                   0, dres.data_ptr(), ld_of(dres), rows, c, groups, act)
        return (dx, dres) + (None,) * 16


_DW_FUSED_BWD = _os.environ.get("BGAMD_DW_FUSED_BWD", "1") != "0"   # A/B switch: 0 = the three separate backward kernels
_FOLD_FINALIZE = _os.environ.get("BGAMD_NO_FOLD_FINALIZE") is None and _os.environ.get("BGAMD_DW_RING", "1") != "0"   # A/B switch


class NormActDwConvFn(torch.autograd.Function):
    """dw3x3( act( BatchNorm2d_train(x) ) ): the [BatchNorm2d -> LeakyReLU -> SeparableConv2d_same.conv1] chain of the
    Xception units (deeplab.py:75-87, 90-143) with the activated tensor never written to memory.  The statistics are
    finalised into per-(group, channel) scale/shift (bg_norm_finalize_affine, which also applies the running-statistics
    update), and the depthwise kernel applies them to every input chunk it loads (bg_dwconv3x3_fwd_pre).  Backward:
    depthwise data gradient, depthwise weight gradient on the recomputed activation (bg_dwconv3x3_bwd_weight_pre), then
    the BatchNorm backward of NormActFn with the activation branch recomputed from x.  Values are bit-identical to
    NormActFn followed by DwConv3x3Fn."""

    @staticmethod
    def forward(ctx, x, gamma, beta, dw_weight, arena: Arena, gslot, bslot, wslot: ParamSlot, rmean, rvar, act, eps,
                momentum, pre_stats, bn_groups, dil):
        ctx.q_site = getattr(x, "_bg_dy_site", None)
        x = nhwc(x)
        n, h, w, c = x.shape
        dev, dt = x.device, L.dt(x.dtype)
        rows = n * h * w
        groups = bn_groups
        assert n % groups == 0, f"batch of {n} does not split into {groups} statistic groups"
        assert wslot.phys_shape == (3, 3, c), (wslot.phys_shape, c)
        if pre_stats is not None and pre_stats.shape[1] == groups:
            s = pre_stats
        else:
            s = _f64(2, groups, c, device=dev)
            L.call("bg_norm_stats", dt, x.data_ptr(), rows, c, ld_of(x), groups, s[0].data_ptr(), s[1].data_ptr())
        mean, rstd, scale, shift = _e32(4, groups, c, device=dev).unbind(0)
        upd = rmean is not None
        y = new_act(n, h, w, c, x.dtype, dev)
        desc = L.DwDesc(dt, n, h, w, c, h, w, 1, dil, ld_of(x), ld_of(y))
        if _FOLD_FINALIZE and c <= 4096:
            # the depthwise kernel finalises the statistics itself (no launch between the GEMM and it)
            L.call("bg_dwconv3x3_fwd_pre_stats", desc, x.data_ptr(), s[0].data_ptr(), s[1].data_ptr(), arena.master_ptr(gslot),
                   arena.master_ptr(bslot), eps, momentum, rmean.data_ptr() if upd else None, rvar.data_ptr() if upd else None,
                   mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), groups, int(act),
                   arena.weight_ptr(wslot), y.data_ptr())
        else:
            L.call("bg_norm_finalize_affine", s[0].data_ptr(), s[1].data_ptr(), rows // groups, groups, c,
                   arena.master_ptr(gslot), arena.master_ptr(bslot), eps, momentum, rmean.data_ptr() if upd else None,
                   rvar.data_ptr() if upd else None, mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
            L.call("bg_dwconv3x3_fwd_pre", desc, x.data_ptr(), scale.data_ptr(), shift.data_ptr(), groups, int(act),
                   arena.weight_ptr(wslot), y.data_ptr())
        ctx.save_for_backward(x, mean, rstd, scale, shift)
        ctx.meta = (arena, gslot, bslot, wslot, int(act), groups, dil)
        return y

    @staticmethod
    def backward(ctx, g):
        x, mean, rstd, scale, shift = ctx.saved_tensors
        arena, gslot, bslot, wslot, act, groups, dil = ctx.meta
        g = nhwc(g)
        n, h, w, c = x.shape
        dev, dt = x.device, L.dt(x.dtype)
        rows = n * h * w
        need_dx = ctx.needs_input_grad[0]
        want_affine_grads = gslot.param.requires_grad
        # the one-pass kernel (bg_dwconv3x3_bwd_fused) wherever the depthwise weight gradient is wanted: measured 1.5-1.9x
        # the three kernels at dilation 1, 1.15-1.2x at dilation 2 (scripts/bench_dwfused.py); with frozen weights (the
        # G-step's pass through D) it only beats the remaining two kernels on large tensors at dilation 1
        fused = (_DW_FUSED_BWD and x.dtype == torch.bfloat16 and (need_dx or want_affine_grads)
                 and h * w * max(ld_of(x), ld_of(g)) * 2 < (1 << 31)
                 and (ctx.needs_input_grad[3] or (dil == 1 and rows * c >= 150_000_000)))
        if ctx.needs_input_grad[3]:
            arena.ensure_grad(wslot)
            if not fused:
                desc = L.DwDesc(dt, n, h, w, c, h, w, 1, dil, ld_of(x), ld_of(g))
                wgrad_call(dev, (x, g, scale, shift), "bg_dwconv3x3_bwd_weight_pre", desc, x.data_ptr(), scale.data_ptr(),
                           shift.data_ptr(), groups, act, g.data_ptr(), arena.grad_ptr(wslot))
        if not (need_dx or want_affine_grads):
            return (None,) * 16
        da = new_act(n, h, w, c, x.dtype, dev)   # gradient of the activated tensor
        gptr, bptr = arena.master_ptr(gslot), arena.master_ptr(bslot)
        s = _f64(2, groups, c, device=dev)
        if fused:
            # ONE pass over g and x: the depthwise data gradient, the depthwise weight gradient on the recomputed activation
            # and the two statistics of the BatchNorm backward (three launches and six tensor passes in round 2)
            desc = L.DwDesc(dt, n, h, w, c, h, w, 1, dil, ld_of(x), ld_of(g))
            L.call("bg_dwconv3x3_bwd_fused", desc, g.data_ptr(), arena.weight_ptr(wslot), x.data_ptr(), scale.data_ptr(),
                   shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), groups, act, da.data_ptr(), ld_of(da),
                   arena.grad_ptr(wslot) if ctx.needs_input_grad[3] else None, s[0].data_ptr(), s[1].data_ptr())
        else:
            desc = L.DwDesc(dt, n, h, w, c, h, w, 1, dil, ld_of(da), ld_of(g))
            L.call("bg_dwconv3x3_bwd_data", desc, g.data_ptr(), arena.weight_ptr(wslot), da.data_ptr())
            L.call("bg_norm_act_bwd_reduce", dt, da.data_ptr(), ld_of(da), None, 0, x.data_ptr(), ld_of(x), mean.data_ptr(),
                   rstd.data_ptr(), gptr, bptr, rows, c, groups, act, s[0].data_ptr(), s[1].data_ptr())
        dg = db = None
        if want_affine_grads:
            arena.ensure_grad(gslot)
            arena.ensure_grad(bslot)
            dg, db = arena.grad_ptr(gslot), arena.grad_ptr(bslot)
        dx = new_act(n, h, w, c, x.dtype, dev) if need_dx else None
        args = (dt, da.data_ptr(), ld_of(da), None, 0, x.data_ptr(), ld_of(x),
                s[0].data_ptr(), s[1].data_ptr(), gptr, bptr, mean.data_ptr(), rstd.data_ptr(), 1, dg, db, L.ptr(dx),
                0 if dx is None else ld_of(dx), None, 0, rows, c, groups, act)
        _apply_stats_maybe_q8(ctx.q_site if act else None, dx, args, n, h, w, c)
        return (dx,) + (None,) * 15


def leaky_relu(x):
    """Stand-alone LeakyReLU(0.2) (a Block called on a not-yet-activated input)."""
    return NormActFn.apply(x, None, None, None, None, None, None, None, None, "identity", False, True, 0.0, 0.0)


def add(x, res, act=False):
    """x + res (+ LeakyReLU): the Block residual when its last unit is a conv (deeplab.py:141)."""
    return NormActFn.apply(x, res, None, None, None, None, None, None, None, "identity", False, act, 0.0, 0.0)


class ForkFn(torch.autograd.Function):
    """One producer, k consumers: backward sums the k gradients with our kernel
    (instead of autograd's implicit torch.add)."""

    @staticmethod
    def forward(ctx, x, k: int):
        ctx.k = k
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        gs = [nhwc(g) for g in gs if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        for g in gs[1:]:
            n, h, w, c = acc.shape
            out = new_act(n, h, w, c, acc.dtype, acc.device)
            L.call("bg_norm_act_fwd", L.dt(acc.dtype), acc.data_ptr(), ld_of(acc), None, None, g.data_ptr(), ld_of(g),
                   out.data_ptr(), ld_of(out), n * h * w, c, 1, 0)
            acc = out
        return acc, None


def fork(x, k: int):
    return ForkFn.apply(x, k)


# -------------------------------------------------------------- resampling / pooling
class ResizeBilinearFn(torch.autograd.Function):
    """F.interpolate(mode='bilinear', align_corners=True) (deeplab.py:375,379,663)."""

    @staticmethod
    def forward(ctx, x, ho: int, wo: int, out_dtype: Optional[torch.dtype]):
        x = nhwc(x)
        n, hi, wi, c = x.shape
        out_dtype = out_dtype or x.dtype
        y = new_act(n, ho, wo, c, out_dtype, x.device)
        L.call("bg_resize_bilinear_fwd", L.dt(x.dtype), L.dt(out_dtype), x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), n, hi,
               wi, ho, wo, c)
        ctx.meta = (n, hi, wi, ho, wo, c, x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        n, hi, wi, ho, wo, c, in_dtype = ctx.meta
        g = nhwc(g)
        dx = new_act(n, hi, wi, c, in_dtype, g.device)
        if hi == 1 and wi == 1:
            # broadcast forward -> plain column sum backward
            acc = _f32(n, c, device=g.device)
            L.call("bg_colsum", L.dt(g.dtype), g.data_ptr(), ld_of(g), n * ho * wo, c, n, 1.0, acc.data_ptr())
            L.call("bg_cast_rows", L.F32, L.dt(in_dtype), acc.data_ptr(), c, dx.data_ptr(), ld_of(dx), n, c)
        else:
            L.call("bg_resize_bilinear_bwd", L.dt(g.dtype), L.dt(in_dtype), g.data_ptr(), ld_of(g), dx.data_ptr(), ld_of(dx), n,
                   hi, wi, ho, wo, c)
        return dx, None, None, None


class GlobalAvgPoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d((1,1)) (deeplab.py:621)."""

    @staticmethod
    def forward(ctx, x, samples=None):
        """samples: number of pooled groups (default: the leading dimension; a volume [N*D,H,W,C] passes N)."""
        x = nhwc(x)
        nd, h, w, c = x.shape
        n = samples or nd
        rows = nd * h * w
        acc = _f32(n, c, device=x.device)
        L.call("bg_colsum", L.dt(x.dtype), x.data_ptr(), ld_of(x), rows, c, n, float(n) / rows, acc.data_ptr())
        y = new_act(n, 1, 1, c, x.dtype, x.device)
        L.call("bg_cast_rows", L.F32, L.dt(x.dtype), acc.data_ptr(), c, y.data_ptr(), ld_of(y), n, c)
        ctx.meta = (n, nd, h, w, c, x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        n, nd, h, w, c, dtype = ctx.meta
        g = g.contiguous()
        gf = _f32(n, c, device=g.device)
        L.call("bg_cast_rows", L.dt(g.dtype), L.F32, g.data_ptr(), c, gf.data_ptr(), c, n, c)
        dx = new_act(nd, h, w, c, dtype, g.device)
        rows = nd * h * w
        L.call("bg_broadcast_rows", L.dt(dtype), gf.data_ptr(), float(n) / rows, dx.data_ptr(), ld_of(dx), rows, c, n)
        return dx, None


class ConcatFn(torch.autograd.Function):
    """torch.cat(dim=1) of the reference (deeplab.py:377,664) = channel concat in NHWC."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [nhwc(x) for x in xs]
        n, h, w, _ = xs[0].shape
        cs = [x.shape[3] for x in xs]
        out = new_act(n, h, w, sum(cs), xs[0].dtype, xs[0].device)
        off = 0
        dt = L.dt(out.dtype)
        for x, c in zip(xs, cs):
            L.call("bg_cast_rows", dt, dt, x.data_ptr(), ld_of(x), out.data_ptr() + off * out.element_size(), ld_of(out),
                   n * h * w, c)
            off += c
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for c in ctx.cs:
            outs.append(g[..., off:off + c])  # zero-copy channel slices; consumers honour the pixel stride
            off += c
        return tuple(outs)


def concat(*xs):
    return ConcatFn.apply(*xs)


# ----------------------------------------------------------------------- volumes (3-D path)
class DepthUnfoldFn(torch.autograd.Function):
    """[N*D,H,W,C] -> [N*Do,H,W,KD*C]: channel block kd of slice od is input slice od*stride - pad + kd*dil (zeros
    outside).  With it nn.Conv3d runs on the 2-D GEMM kernels (bg_depth_unfold / bg_depth_fold)."""

    @staticmethod
    def forward(ctx, x, n, kd, stride, pad, dil):
        x = nhwc(x)
        nd, h, w, c = x.shape
        d = nd // n
        do = (d + 2 * pad - dil * (kd - 1) - 1) // stride + 1
        y = new_act(n * do, h, w, kd * c, x.dtype, x.device)
        L.call("bg_depth_unfold", L.dt(x.dtype), x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), n, d, do, h * w, c, kd, stride,
               pad, dil)
        ctx.meta = (n, d, do, h, w, c, kd, stride, pad, dil, x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        n, d, do, h, w, c, kd, stride, pad, dil, dtype = ctx.meta
        g = nhwc(g)
        dx = new_act(n * d, h, w, c, dtype, g.device)
        L.call("bg_depth_fold", L.dt(dtype), g.data_ptr(), ld_of(g), dx.data_ptr(), ld_of(dx), n, d, do, h * w, c, kd, stride,
               pad, dil)
        return dx, None, None, None, None, None


class DepthFoldFn(torch.autograd.Function):
    """Adjoint of DepthUnfoldFn as a forward op: [N*Dk,H,W,KD*C] -> [N*D,H,W,C], slice id of the result collects
    channel block kd of every slice od with od*stride - pad + kd*dil == id.  The depth half of nn.ConvTranspose3d
    (its in-plane half is ConvTranspose2dFn over the KD*C channels)."""

    @staticmethod
    def forward(ctx, x, n, kd, stride, pad, dil, d):
        x = nhwc(x)
        ndk, h, w, kc = x.shape
        dk, c = ndk // n, kc // kd
        assert (d + 2 * pad - dil * (kd - 1) - 1) // stride + 1 == dk, "bad output depth for this fold"
        y = new_act(n * d, h, w, c, x.dtype, x.device)
        L.call("bg_depth_fold", L.dt(x.dtype), x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), n, d, dk, h * w, c, kd, stride, pad, dil)
        ctx.meta = (n, d, dk, h, w, c, kd, stride, pad, dil, x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        n, d, dk, h, w, c, kd, stride, pad, dil, dtype = ctx.meta
        g = nhwc(g)
        dx = new_act(n * dk, h, w, kd * c, dtype, g.device)
        L.call("bg_depth_unfold", L.dt(dtype), g.data_ptr(), ld_of(g), dx.data_ptr(), ld_of(dx), n, d, dk, h * w, c, kd, stride,
               pad, dil)
        return dx, None, None, None, None, None, None


class DepthAvg2Fn(torch.autograd.Function):
    """Depth half of nn.AvgPool3d(2, stride=1, padding=p), p in {0, 1}: [N*D,H,W,C] -> [N*(D+2p-1),H,W,C]."""

    @staticmethod
    def forward(ctx, x, n, p: int):
        x = nhwc(x)
        nd, h, w, c = x.shape
        d = nd // n
        do = d + 2 * p - 1
        y = new_act(n * do, h, w, c, x.dtype, x.device)
        L.call("bg_depth_avg2", L.dt(x.dtype), x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), n, d, do, h * w, c, -p)
        ctx.meta = (n, d, do, h, w, c, p)
        return y

    @staticmethod
    def backward(ctx, g):
        n, d, do, h, w, c, p = ctx.meta
        g = nhwc(g)
        dx = new_act(n * d, h, w, c, g.dtype, g.device)
        L.call("bg_depth_avg2", L.dt(g.dtype), g.data_ptr(), ld_of(g), dx.data_ptr(), ld_of(dx), n, do, d, h * w, c, p - 1)
        return dx, None, None


def avgpool3d_2(x, n, p: int):
    """nn.AvgPool3d(2, stride=1, padding=p) on a folded volume: the mean of 8 is the depth pair mean of 2 x 2 means."""
    return avgpool2x2(DepthAvg2Fn.apply(x, n, int(p)), int(p))


class DwConv3dFn(torch.autograd.Function):
    """Depthwise 3x3x3 of SeparableConv3d_same with fixed_padding folded in (bg_dwconv3x3x3_*)."""

    @staticmethod
    def forward(ctx, x, weight, arena: Arena, wslot: ParamSlot, n, stride, dil):
        x = nhwc(x)
        nd, h, w, c = x.shape
        d = nd // n
        assert wslot.phys_shape == (3, 3, 3, c), (wslot.phys_shape, c)
        do, ho, wo = -(-d // stride), -(-h // stride), -(-w // stride)
        y = new_act(n * do, ho, wo, c, x.dtype, x.device)
        desc = L.Dw3Desc(L.dt(x.dtype), n, d, h, w, c, do, ho, wo, stride, dil, ld_of(x), ld_of(y))
        L.call("bg_dwconv3x3x3_fwd", desc, x.data_ptr(), arena.weight_ptr(wslot), y.data_ptr())
        if weight.requires_grad:
            ctx.save_for_backward(x)
        ctx.meta = (arena, wslot, n, d, h, w, c, do, ho, wo, stride, dil, x.dtype, x.device)
        return y

    @staticmethod
    def backward(ctx, g):
        arena, wslot, n, d, h, w, c, do, ho, wo, stride, dil, xdtype, xdev = ctx.meta
        g = nhwc(g)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new_act(n * d, h, w, c, xdtype, xdev)
            desc = L.Dw3Desc(L.dt(xdtype), n, d, h, w, c, do, ho, wo, stride, dil, ld_of(dx), ld_of(g))
            L.call("bg_dwconv3x3x3_bwd_data", desc, g.data_ptr(), arena.weight_ptr(wslot), dx.data_ptr())
        if ctx.needs_input_grad[1]:
            (x,) = ctx.saved_tensors
            arena.ensure_grad(wslot)
            desc = L.Dw3Desc(L.dt(xdtype), n, d, h, w, c, do, ho, wo, stride, dil, ld_of(x), ld_of(g))
            wgrad_call(xdev, (x, g), "bg_dwconv3x3x3_bwd_weight", desc, x.data_ptr(), g.data_ptr(), arena.grad_ptr(wslot))
        return dx, None, None, None, None, None, None


class DepthResizeFn(torch.autograd.Function):
    """Linear interpolation along depth, align_corners=True; trilinear = this, then ResizeBilinearFn."""

    @staticmethod
    def forward(ctx, x, n, do, out_dtype: Optional[torch.dtype]):
        x = nhwc(x)
        nd, h, w, c = x.shape
        di = nd // n
        out_dtype = out_dtype or x.dtype
        y = new_act(n * do, h, w, c, out_dtype, x.device)
        L.call("bg_depth_resize_fwd", L.dt(x.dtype), L.dt(out_dtype), x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), n, di, do,
               h * w, c)
        ctx.meta = (n, di, do, h, w, c, x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        n, di, do, h, w, c, in_dtype = ctx.meta
        g = nhwc(g)
        dx = new_act(n * di, h, w, c, in_dtype, g.device)
        L.call("bg_depth_resize_bwd", L.dt(g.dtype), L.dt(in_dtype), g.data_ptr(), ld_of(g), dx.data_ptr(), ld_of(dx), n, di, do,
               h * w, c)
        return dx, None, None, None


def resize_trilinear(x, n, do, ho, wo, out_dtype=None):
    """F.interpolate(mode='trilinear', align_corners=True) on a folded volume [N*D,H,W,C] -> [N*do,ho,wo,C]
    (separable: depth first -- on the smaller tensor when upsampling -- then the 2-D bilinear kernel)."""
    d = x.shape[0] // n
    if do != d:
        x = DepthResizeFn.apply(x, n, do, None)
    return ResizeBilinearFn.apply(x, ho, wo, out_dtype)


# ------------------------------------------------------------- partial convolution (row (f)-4)
class RowsMask:
    """A mask whose `channels` channels are identical: one fp32 value per pixel of the folded volume [N*D*H*W]
    (every update_mask of PartialConv3d is one, partialconv3d.py:62-64)."""
    __slots__ = ("rows", "channels")

    def __init__(self, rows, channels):
        self.rows, self.channels = rows, int(channels)


def mask_window(n, dims, k, stride, pad, eps, full=None, full_channels=0, rows=(), planar=False):
    """Mask half of PartialConv3d over the channel segments `full` (folded per-channel 0/1 mask [N*D,H,W,C] with
    `full_channels` real channels, or None) and up to two RowsMask: returns (update_mask, mask_ratio) as fp32
    [N*Do*Ho*Wo] vectors and the output dims.  No gradient (constants).  planar: the 2-D PartialConv2d on n*d
    independent images (k x k window, depth untouched)."""
    d, h, w = dims
    do, ho, wo = ((x + 2 * pad - k) // stride + 1 for x in (d, h, w))
    if planar:
        do = d
    assert len(rows) <= 2 and (full is not None or rows)
    dev = full.device if full is not None else rows[0].rows.device
    upd = torch.empty(n * do * ho * wo, dtype=torch.float32, device=dev)
    ratio = torch.empty_like(upd)
    r = [(m.rows.data_ptr(), m.channels) for m in rows] + [(None, 0)] * (2 - len(rows))
    for m in rows:
        assert m.rows.numel() == n * d * h * w and m.rows.dtype == torch.float32
    if full is not None:
        full = nhwc(full)
        assert tuple(full.shape[:3]) == (n * d, h, w)
    L.call("bg_mask_window", L.dt(full.dtype) if full is not None else 0, L.ptr(full), 0 if full is None else ld_of(full),
           full_channels, r[0][0], r[0][1], r[1][0], r[1][1], n, d, h, w, do, ho, wo, k, stride, pad, int(planar), float(eps),
           upd.data_ptr(), ratio.data_ptr())
    return upd, ratio, (do, ho, wo)


def nearest_rows(m: RowsMask, n, src, dst):
    """F.interpolate(mode='nearest') of a per-pixel mask (infill3d.py:218-222 for the mask)."""
    y = torch.empty(n * dst[0] * dst[1] * dst[2], dtype=torch.float32, device=m.rows.device)
    L.call("bg_resize_nearest3d_rows", m.rows.data_ptr(), y.data_ptr(), n, *src, *dst)
    return RowsMask(y, m.channels)


class MaskedConcatFn(torch.autograd.Function):
    """torch.cat([x_i], dim=1) * torch.cat([mask_i], dim=1) in one pass per segment (infill3d.py:224-225 +
    partialconv3d.py:77): segment i is written as x_i * mask_i into its channel slice.  mask_i: RowsMask (per pixel) or
    a folded per-channel tensor.  Masks are constants; the adjoint scales the slices of dy the same way."""

    @staticmethod
    def forward(ctx, masks, *xs):
        xs = [nhwc(x) for x in xs]
        n, h, w, _ = xs[0].shape
        cs = [x.shape[3] for x in xs]
        assert all(c % vec_of(xs[0].dtype) == 0 for c in cs[:-1])
        out = new_act(n, h, w, sum(cs), xs[0].dtype, xs[0].device)
        ctx.masks, ctx.cs = [m if isinstance(m, RowsMask) else nhwc(m) for m in masks], cs
        MaskedConcatFn._apply(ctx.masks, cs, xs, [0] * len(xs), out, None)
        return out

    @staticmethod
    def _apply(masks, cs, srcs, src_offs, dst, dsts):
        """forward: srcs[i] (whole) -> dst slice i; backward: slice i of srcs[0] -> dsts[i] (whole)."""
        rows = srcs[0].shape[0] * srcs[0].shape[1] * srcs[0].shape[2]
        off = 0
        for i, (m, c) in enumerate(zip(masks, cs)):
            if dsts is None:
                src, so, y, yo = srcs[i], 0, dst, off
            else:
                src, so, y, yo = srcs[0], off, dsts[i], 0
            if y is not None:
                es = src.element_size()
                sp, yp = src.data_ptr() + so * es, y.data_ptr() + yo * es
                if isinstance(m, RowsMask):
                    L.call("bg_scale_rows", L.dt(src.dtype), sp, ld_of(src), m.rows.data_ptr(), None, None, yp, ld_of(y), rows, c)
                else:
                    L.call("bg_mul_rows", L.dt(src.dtype), sp, ld_of(src), m.data_ptr(), ld_of(m), yp, ld_of(y), rows, c)
            off += c

    @staticmethod
    def backward(ctx, g):
        g = nhwc(g)
        n, h, w, _ = g.shape
        dxs = [new_act(n, h, w, c, g.dtype, g.device) if ctx.needs_input_grad[1 + i] else None for i, c in enumerate(ctx.cs)]
        MaskedConcatFn._apply(ctx.masks, ctx.cs, [g], None, None, dxs)
        return (None, *dxs)


def rows_from_scalar(s, n_slices, h, w, c, dtype):
    """[rows] fp32 -> folded tensor [n_slices,h,w,c] with every channel equal to s[row] (the Cout identical
    channels of PartialConv3d's update_mask)."""
    y = new_act(n_slices, h, w, c, dtype, s.device)
    L.call("bg_scale_rows", L.dt(dtype), None, 0, s.data_ptr(), None, None, y.data_ptr(), ld_of(y), n_slices * h * w, c)
    return y


class ScaleRowsFn(torch.autograd.Function):
    """y[r,c] = x[r,c] * s[r] (+ bias[c] * t[r]): raw_out * mask_ratio, and the bias form of partialconv3d.py:79-84."""

    @staticmethod
    def forward(ctx, x, s, bias, t, arena, bslot):
        x = nhwc(x)
        n, h, w, c = x.shape
        y = new_act(n, h, w, c, x.dtype, x.device)
        L.call("bg_scale_rows", L.dt(x.dtype), x.data_ptr(), ld_of(x), s.data_ptr(), None if bslot is None else arena.master_ptr(bslot),
               L.ptr(t), y.data_ptr(), ld_of(y), n * h * w, c)
        ctx.save_for_backward(s, t)
        ctx.meta = (arena, bslot)
        return y

    @staticmethod
    def backward(ctx, g):
        s, t = ctx.saved_tensors
        arena, bslot = ctx.meta
        g = nhwc(g)
        n, h, w, c = g.shape
        rows = n * h * w
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new_act(n, h, w, c, g.dtype, g.device)
            L.call("bg_scale_rows", L.dt(g.dtype), g.data_ptr(), ld_of(g), s.data_ptr(), None, None, dx.data_ptr(), ld_of(dx), rows, c)
        if bslot is not None and ctx.needs_input_grad[2]:
            arena.ensure_grad(bslot)
            tmp = new_act(n, h, w, c, g.dtype, g.device)
            L.call("bg_scale_rows", L.dt(g.dtype), g.data_ptr(), ld_of(g), t.data_ptr(), None, None, tmp.data_ptr(), ld_of(tmp), rows, c)
            L.call("bg_colsum", L.dt(g.dtype), tmp.data_ptr(), ld_of(tmp), rows, c, 1, 1.0, arena.grad_ptr(bslot))
        return dx, None, None, None, None, None


class NearestResize3dFn(torch.autograd.Function):
    """F.interpolate(size=(do,ho,wo), mode='nearest') on a folded volume (infill3d.py:217-222)."""

    @staticmethod
    def forward(ctx, x, n, do, ho, wo):
        x = nhwc(x)
        nd, hi, wi, c = x.shape
        di = nd // n
        y = new_act(n * do, ho, wo, c, x.dtype, x.device)
        L.call("bg_resize_nearest3d_fwd", L.dt(x.dtype), x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), n, di, hi, wi, do, ho, wo, c)
        ctx.meta = (n, di, hi, wi, do, ho, wo, c, x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        n, di, hi, wi, do, ho, wo, c, dtype = ctx.meta
        g = nhwc(g)
        dx = new_act(n * di, hi, wi, c, dtype, g.device)
        L.call("bg_resize_nearest3d_bwd", L.dt(dtype), g.data_ptr(), ld_of(g), dx.data_ptr(), ld_of(dx), n, di, hi, wi, do, ho, wo, c)
        return dx, None, None, None, None


class TrilinearResize3dFn(torch.autograd.Function):
    """F.interpolate(size=(do,ho,wo), mode='trilinear') with align_corners unset (= False) on a folded volume: what
    PConvUNet3d(upsampling_mode='trilinear') applies to the features (infill3d.py:217-220)."""

    @staticmethod
    def forward(ctx, x, n, do, ho, wo):
        x = nhwc(x)
        nd, hi, wi, c = x.shape
        di = nd // n
        y = new_act(n * do, ho, wo, c, x.dtype, x.device)
        L.call("bg_resize_trilinear3d_fwd", L.dt(x.dtype), x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), n, di, hi, wi, do, ho, wo, c)
        ctx.meta = (n, di, hi, wi, do, ho, wo, c, x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        n, di, hi, wi, do, ho, wo, c, dtype = ctx.meta
        g = nhwc(g)
        dx = new_act(n * di, hi, wi, c, dtype, g.device)
        L.call("bg_resize_trilinear3d_bwd", L.dt(dtype), g.data_ptr(), ld_of(g), dx.data_ptr(), ld_of(dx), n, di, hi, wi, do, ho, wo, c)
        return dx, None, None, None, None


class PCDropoutFn(torch.autograd.Function):
    """PCDropout3d.forward in training mode (infill3d.py:119-131) for a given draw keep[n][c] of nn.Dropout3d:
    returns (input_d, mask_d) with mask_d a per-channel folded tensor.  mask: RowsMask or a folded per-channel tensor."""

    @staticmethod
    def forward(ctx, x, mask, keep, n, scale):
        x = nhwc(x)
        nd, h, w, c = x.shape
        rows = nd * h * w
        y = new_act(nd, h, w, c, x.dtype, x.device)
        mo = new_act(nd, h, w, c, x.dtype, x.device)
        ctx.mask = mask if isinstance(mask, RowsMask) else nhwc(mask)
        ctx.keep, ctx.meta = keep, (rows, rows // n, c, float(scale))
        PCDropoutFn._run(x, ctx.mask, keep, y, mo, *ctx.meta)
        ctx.mark_non_differentiable(mo)
        return y, mo

    @staticmethod
    def _run(x, mask, keep, y, mo, rows, per_sample, c, scale):
        assert keep.dtype == torch.float32 and keep.shape[1] >= c and keep.is_contiguous()
        is_rows = isinstance(mask, RowsMask)
        L.call("bg_pc_dropout", L.dt(x.dtype), x.data_ptr(), ld_of(x), mask.rows.data_ptr() if is_rows else None,
               None if is_rows else mask.data_ptr(), 0 if is_rows else ld_of(mask), keep.data_ptr(), keep.shape[1], y.data_ptr(),
               ld_of(y), L.ptr(mo), 0 if mo is None else ld_of(mo), rows, per_sample, c, scale)

    @staticmethod
    def backward(ctx, g, _gm):
        g = nhwc(g)
        dx = new_act(*g.shape, g.dtype, g.device)
        PCDropoutFn._run(g, ctx.mask, ctx.keep, dx, None, *ctx.meta)
        return dx, None, None, None, None


def concat_full_masks(masks, reals):
    """torch.cat of per-channel folded masks along the channels (infill3d.py:225) as row-slice copies; reals[i] = real
    channels of masks[i] (every segment but the last must fill its padded width)."""
    masks = [nhwc(m) for m in masks]
    assert all(m.shape[3] == r for m, r in zip(masks[:-1], reals[:-1]))
    n, h, w, _ = masks[0].shape
    out = new_act(n, h, w, sum(m.shape[3] for m in masks), masks[0].dtype, masks[0].device)
    off, es = 0, out.element_size()
    for m in masks:
        L.call("bg_cast_rows", L.dt(m.dtype), L.dt(out.dtype), m.data_ptr(), ld_of(m), out.data_ptr() + off * es, ld_of(out),
               n * h * w, m.shape[3])
        off += m.shape[3]
    return out


class TVLossCompFn(torch.autograd.Function):
    """total_variation_loss(mask*input + (1-mask)*output) of InpaintingLoss (utils/losses.py:40-44,71,97) on
    contiguous fp32 [N,C,D,H,W] tensors (shifts along H and D, as the reference's 4-D function acts on them) or
    [N,C,H,W] ones (shifts along W and H: the same kernel on the view [N,C,H,W,1]); differentiable w.r.t. `output`."""

    @staticmethod
    def forward(ctx, output, inp, mask):
        o, i_, m = (t.contiguous().float() for t in (output, inp, mask))
        n, c, d, h, w = o.shape if o.dim() == 5 else (*o.shape, 1)
        comp = torch.empty_like(o)
        L.call("bg_blend_f32", m.data_ptr(), i_.data_ptr(), o.data_ptr(), comp.data_ptr(), o.numel())
        loss = _f32(1, device=o.device)
        L.call("bg_tv_loss_fwd", comp.data_ptr(), n * c, d, h, w, loss.data_ptr())
        ctx.save_for_backward(comp, m)
        ctx.dims = (n * c, d, h, w)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        comp, m = ctx.saved_tensors
        a, d, h, w = ctx.dims
        coef = g.reshape(1).float().contiguous()
        dcomp = torch.empty_like(comp)
        L.call("bg_tv_loss_bwd", comp.data_ptr(), a, d, h, w, coef.data_ptr(), dcomp.data_ptr())
        dout = torch.empty_like(comp)
        L.call("bg_blend_f32", m.data_ptr(), None, dcomp.data_ptr(), dout.data_ptr(), comp.numel())   # (1 - m) * dcomp
        return dout, None, None


# ---------------------------------------------------------------------------- head
class LinearHeadFn(torch.autograd.Function):
    """reshape(N,-1) + nn.Linear(F,1) on the NCHW-ordered features (deeplab_gan.py:32-35)."""

    @staticmethod
    def forward(ctx, x, weight, bias, arena: Arena, wslot: ParamSlot, bslot: ParamSlot):
        x = nhwc(x)
        n, h, w, c = x.shape
        assert wslot.numel == c * h * w, "Linear head in_features must be C*H*W of the feature map"
        logits = _f32(n, 1, device=x.device)
        L.call("bg_linear_head_fwd", L.dt(x.dtype), x.data_ptr(), ld_of(x), arena.master_ptr(wslot),
               arena.master_ptr(bslot), logits.data_ptr(), n, h * w, c)
        ctx.save_for_backward(x)
        ctx.meta = (arena, wslot, bslot)
        return logits

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        arena, wslot, bslot = ctx.meta
        n, h, w, c = x.shape
        g = g.contiguous().float()
        dx = new_act(n, h, w, c, x.dtype, x.device) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1]:
            arena.ensure_grad(wslot)
            arena.ensure_grad(bslot)
            dw, db = arena.grad_ptr(wslot), arena.grad_ptr(bslot)
        L.call("bg_linear_head_bwd", L.dt(x.dtype), x.data_ptr(), ld_of(x), arena.master_ptr(wslot), g.data_ptr(), L.ptr(dx),
               0 if dx is None else ld_of(dx), dw, db, n, h * w, c)
        return dx, None, None, None, None, None


class LinearHeadNoBiasFn(torch.autograd.Function):
    """nn.Linear(F, 1, bias=False) on pooled features (infill3d_gan.py:30,61-62); `zero_bias`: a device zero."""

    @staticmethod
    def forward(ctx, x, weight, arena: Arena, wslot: ParamSlot, zero_bias):
        x = nhwc(x)
        n, h, w, c = x.shape
        assert wslot.numel == c * h * w
        logits = _f32(n, 1, device=x.device)
        L.call("bg_linear_head_fwd", L.dt(x.dtype), x.data_ptr(), ld_of(x), arena.master_ptr(wslot), zero_bias.data_ptr(),
               logits.data_ptr(), n, h * w, c)
        ctx.save_for_backward(x)
        ctx.meta = (arena, wslot)
        return logits

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        arena, wslot = ctx.meta
        n, h, w, c = x.shape
        g = g.contiguous().float()
        dx = new_act(n, h, w, c, x.dtype, x.device) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            arena.ensure_grad(wslot)
            dw = arena.grad_ptr(wslot)
        if dx is not None or dw is not None:
            L.call("bg_linear_head_bwd", L.dt(x.dtype), x.data_ptr(), ld_of(x), arena.master_ptr(wslot), g.data_ptr(), L.ptr(dx),
                   0 if dx is None else ld_of(dx), dw, None, n, h * w, c)
        return dx, None, None, None, None


# -------------------------------------------------------------------------- losses
class BCEWithLogitsFn(torch.autograd.Function):
    """nn.BCEWithLogitsLoss() on [N,1] logits (utils/losses.py:138)."""

    @staticmethod
    def forward(ctx, logits, target):
        x = logits.contiguous().float()
        t = target.contiguous().float()
        loss = _f32(1, device=x.device)
        dx = torch.empty_like(x)
        L.call("bg_bce_logits", x.data_ptr(), t.data_ptr(), x.numel(), loss.data_ptr(), dx.data_ptr())
        ctx.save_for_backward(dx)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None  # N scalars of glue


PIXEL_LOSS_KINDS = {"l1": 0, "smooth_l1": 1, "l2": 2}


class L1LossFn(torch.autograd.Function):
    """inv_norm * sum(f(p-t) * w), f = |d| (nn.L1Loss / L1LossWeighted, losses.py:101-112), SmoothL1 (beta 1)
    or d^2 (nn.SmoothL1Loss / nn.MSELoss, train_gan.py:147-150; L2LossWeighted, losses.py:115-126)."""

    @staticmethod
    def forward(ctx, pred, target, weights, inv_norm: float, kind: int = 0):
        p = pred.contiguous().float()
        t = target.contiguous().float()
        wt = None if weights is None else weights.contiguous().float()
        loss = _f32(1, device=p.device)
        L.call("bg_pixel_loss_fwd", kind, p.data_ptr(), t.data_ptr(), L.ptr(wt), p.numel(), inv_norm, loss.data_ptr())
        ctx.save_for_backward(p, t, wt)
        ctx.inv_norm, ctx.kind = inv_norm, kind
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        p, t, wt = ctx.saved_tensors
        dp = torch.empty_like(p)
        coef = g.reshape(1).float().contiguous()
        L.call("bg_pixel_loss_bwd", ctx.kind, p.data_ptr(), t.data_ptr(), L.ptr(wt), p.numel(), ctx.inv_norm,
               coef.data_ptr(), dp.data_ptr())
        return dp, None, None, None, None


def gp_penalty_value(grad_nchw: torch.Tensor) -> torch.Tensor:
    """mean over N,H,W of (||g||_2 over C - 1)^2 (deeplab_gan.py:112); a constant (no graph)."""
    g = grad_nchw.contiguous().float()
    n, c, h, w = g.shape
    out = _f32(1, device=g.device)
    L.call("bg_gp_penalty", g.data_ptr(), n, c, h * w, 1.0 / (n * h * w), out.data_ptr())
    return out.view(())
