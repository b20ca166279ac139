"""The HIP path at BASELINE.json's full field sizes against the CPU oracle on the same seeded inputs (one whole loop
iteration: D-step + G-step, both Adam updates), and the configurations the small goldens cannot reach:
  c3  1152x768x16, ModifiedMinMax + L1           fp32 path and bf16 path vs oracle.GANStep
  c4  1152x768x16, Wasserstein + gradient penalty fp32 path vs oracle.GANStep, bf16 path vs the fp32 path
  c5  2304x1536x32, fp8 operand path (and the bf16 path beside it) vs oracle.GANStep
Batch 2 (the smallest BatchNorm accepts on the 1x1 global-pool branch): the oracle needs ~25 s and tens of GB per
iteration at this size on the box's 16 cores.

Tolerances: fp32 path d_loss 1e-3, g_loss 2e-2 (the bounds of the 64x64 reference golden: g_loss is evaluated after
D's first, sign-like Adam step).  bf16 path: 2x the measured deviation (printed), stated per assert.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

import bias_gan_amd  # noqa: E402,F401
from bias_gan_amd.architecture.gpsro import deeplab_gan as dxg  # noqa: E402
from bias_gan_amd.gpsro_train.train_gan import GANTrainer  # noqa: E402
from bias_gan_amd.utils import losses  # noqa: E402
from bias_gan_amd.utils import parsing_helpers as ph  # noqa: E402
from oracle import gan_oracle as orc  # noqa: E402  (checker only)

DEV = "cuda"
C, H, W, N = 16, 1152, 768, 2


def _hip_step(dtype, mode, labels, eta, c=C, h=H, w=W, n=N, seeds=(1, 2), field_seed=333, calibrate=False):
    gspec, dspec = orc.generator_spec(c, c, 0, "batch"), orc.discriminator_spec(c, h, w, "batch")
    G = dxg.Generator(c, c, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d, compute_dtype=dtype)
    D = dxg.Discriminator(c, normalizer=nn.BatchNorm2d, input_size=(h, w), compute_dtype=dtype)
    G.load_state_dict(orc.fill_state(gspec, seeds[0]))
    D.load_state_dict(orc.fill_state(dspec, seeds[1]))
    G.to(DEV).train(), D.to(DEV).train()
    crit = losses.GANLoss(mode, n, torch.device(DEV))
    tr = GANTrainer(G, D, ph.get_optimizer(G.parameters(), "Adam", 1e-4, 1e-8, 1e-5),
                    ph.get_optimizer(D.parameters(), "Adam", 1e-4, 1e-8, 1e-5), crit, losses.L1Loss(), loss_type_gan=mode,
                    loss_weight_gp=10.0)
    x, y = orc.synthetic_fields(n, c, h, w, field_seed)
    if calibrate:            # fp8 mode: first exponents of the quantisation sites, training state untouched (no-op otherwise)
        tr.calibrate_fp8(x.to(DEV), y.to(DEV))
    d_loss, g_loss = tr.step(x.to(DEV), y.to(DEV), labels=labels, eta=eta)
    out = (d_loss.item(), g_loss.item(),
           G.state_dict()["model.xception_features.bn1.running_mean"].cpu().numpy().copy(),
           D.state_dict()["xception_features.bn5.running_var"].cpu().numpy().copy())
    del tr, G, D
    torch.cuda.empty_cache()
    return out


def _oracle_step(mode, labels, eta):
    torch.set_num_threads(16)
    gspec, dspec = orc.generator_spec(C, C, 0, "batch"), orc.discriminator_spec(C, H, W, "batch")
    st = orc.GANStep(orc.fill_state(gspec, 1), orc.fill_state(dspec, 2), orc.trainable_keys(gspec), orc.trainable_keys(dspec),
                     "batch", mode)
    x, y = orc.synthetic_fields(N, C, H, W, 333)
    d, g = st.step(x, y, labels=labels, eta=eta)
    return d, g, st.PG["model.xception_features.bn1.running_mean"].numpy().copy(), st.PD["xception_features.bn5.running_var"].numpy().copy()


def _rel(a, b):
    return abs(a - b) / (abs(b) + 1e-30)


def test_c3_full_size_step_vs_oracle():
    torch.manual_seed(11)
    labels = orc.draw_d_labels(N)
    ref = _oracle_step("ModifiedMinMax", labels, None)
    f32 = _hip_step(torch.float32, "ModifiedMinMax", labels, None)
    b16 = _hip_step(torch.bfloat16, "ModifiedMinMax", labels, None)
    print(f"c3 1152x768x16 N=2: oracle d {ref[0]:.6f} g {ref[1]:.6f} | fp32 path d {f32[0]:.6f} g {f32[1]:.6f} "
          f"(rel {_rel(f32[0], ref[0]):.2e} {_rel(f32[1], ref[1]):.2e}) | bf16 path d {b16[0]:.6f} g {b16[1]:.6f} "
          f"(rel {_rel(b16[0], ref[0]):.2e} {_rel(b16[1], ref[1]):.2e})")
    assert _rel(f32[0], ref[0]) <= 1e-3 and _rel(f32[1], ref[1]) <= 2e-2
    for k in (2, 3):   # BatchNorm running statistics after the step (two resp. three momentum updates)
        e = np.abs(f32[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)
        assert e <= 5e-4, (k, e)      # measured 0 (G, two forward passes) and 1.0e-4 (D's last layer, after its Adam step)
        eb = np.abs(b16[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)
        print(f"   running statistic {k}: fp32 path {e:.2e}, bf16 path {eb:.2e}")
        assert eb <= 5e-2, (k, eb)
    # bf16 path: statistics over ~10^4..10^6 pixels per channel are well conditioned at this size, unlike on the 64x64
    # goldens (4x4 maps, N = 2); bounds = 2x the measured deviation
    # measured: d_loss 9.9e-2, g_loss 8.8e-2 (the randomly filled 140-layer nets amplify bf16 storage rounding, see
    # tests/test_parity_gpu.py's module docstring; per-kernel bf16 error is <= 1e-2)
    assert _rel(b16[0], ref[0]) <= 2e-1 and _rel(b16[1], ref[1]) <= 2e-1


def test_c4_wgan_gp_full_size_step_vs_oracle():
    eta = torch.tensor([0.3, 0.8]).view(N, 1, 1, 1)
    ref = _oracle_step("Wasserstein", None, eta)
    f32 = _hip_step(torch.float32, "Wasserstein", None, eta)
    b16 = _hip_step(torch.bfloat16, "Wasserstein", None, eta)
    print(f"c4 WGAN-GP 1152x768x16 N=2: oracle d {ref[0]:.6f} g {ref[1]:.6f} | fp32 path d {f32[0]:.6f} g {f32[1]:.6f} "
          f"(rel {_rel(f32[0], ref[0]):.2e} {_rel(f32[1], ref[1]):.2e}) | bf16 path d {b16[0]:.6f} g {b16[1]:.6f} "
          f"(rel {_rel(b16[0], ref[0]):.2e} {_rel(b16[1], ref[1]):.2e})")
    # d_loss = mean(fake - real) + 10 * penalty: the penalty (a mean of (|grad| - 1)^2 over pixels) dominates
    assert _rel(f32[0], ref[0]) <= 2e-3 and _rel(f32[1], ref[1]) <= 2e-2
    assert _rel(b16[0], ref[0]) <= 6e-2 and _rel(b16[1], ref[1]) <= 4e-2     # measured 2.9e-2, 1.6e-2


def test_c5_full_size_step_fp8_vs_oracle():
    """BASELINE.json configs[4]: 2304x1536x32 on the fp8 operand path (e4m3 weights / inputs, e5m2 output gradients,
    block-scaled MFMA 16x16x128, fp32 accumulation; bf16 storage) -- the generator's output and one whole loop iteration
    after calibrate_fp8 against oracle.GANStep (fp32, CPU: ~100 s and ~100 GB at this size) on the same seeded inputs and
    labels, next to the bf16 path's distance from the same oracle.  No tiling: batch 2 of these fields needs < 40 GB of
    the 288 GB.

    What the bounds mean.  e4m3 rounds every operand element to 3 mantissa bits (rms 2-3 %, unbiased); a dot product of
    random-sign terms does not average that away, so every GEMM layer adds ~4 % of noise to its output and the ~70 GEMM
    layers between the input field and D's logits accumulate sqrt(70) x 4 % ~ 30 % on the logits of these RANDOMLY
    FILLED nets (bf16: 1 %, measured below) -- d_loss, a forward-only quantity, shows exactly that.  g_loss is evaluated
    after D's first Adam step, which is sign-like (m / sqrt(v) = +-1): it measures how many of D's 38 M weight-gradient
    SIGNS agree with the oracle's, and 2-mantissa-bit gradients through 60 layers flip many of the small ones.  The
    per-kernel and per-Block properties are pinned tightly in tests/test_fp8_gpu.py (same-operand GEMM 1e-2 of max,
    Block 6.5e-2 rms teacher-forced); this test pins that the full-size path is wired correctly and stays in that regime."""
    c, h, w, n = 32, 2304, 1536, 2
    torch.manual_seed(5)
    labels = orc.draw_d_labels(n)
    torch.set_num_threads(16)
    gspec, dspec = orc.generator_spec(c, c, 0, "batch"), orc.discriminator_spec(c, h, w, "batch")
    st = orc.GANStep(orc.fill_state(gspec, 1), orc.fill_state(dspec, 2), orc.trainable_keys(gspec), orc.trainable_keys(dspec),
                     "batch", "ModifiedMinMax")
    x, y = orc.synthetic_fields(n, c, h, w, 9)
    d_ref = st.d_step(x, y, labels)
    g_ref, fake_ref = st.g_step(x, y)
    fake_ref = fake_ref.clone()
    del st
    res = {}
    from bias_gan_amd import ops as _ops
    for tag, dtype, min_work in (("bf16", torch.bfloat16, None), ("fp8", torch.float8_e4m3fn, None), ("fp8_all", torch.float8_e4m3fn, 0)):
        prev = _ops.set_fp8_min_work(min_work) if min_work is not None else None    # 0: EVERY eligible layer on fp8 operands
        G = dxg.Generator(c, c, "Interpolate", "Uniform", 0, normalizer=nn.BatchNorm2d, compute_dtype=dtype)
        D = dxg.Discriminator(c, normalizer=nn.BatchNorm2d, input_size=(h, w), compute_dtype=dtype)
        G.load_state_dict(orc.fill_state(gspec, 1)), D.load_state_dict(orc.fill_state(dspec, 2))
        G.to(DEV).train(), D.to(DEV).train()
        crit = losses.GANLoss("ModifiedMinMax", n, torch.device(DEV))
        tr = GANTrainer(G, D, ph.get_optimizer(G.parameters(), "Adam", 1e-4, 1e-8, 1e-5),
                        ph.get_optimizer(D.parameters(), "Adam", 1e-4, 1e-8, 1e-5), crit, losses.L1Loss())
        xd, yd = x.to(DEV), y.to(DEV)
        tr.calibrate_fp8(xd, yd)
        with torch.no_grad():     # (one more momentum update of the running statistics: they are not compared here)
            fake = G(xd).float().cpu()
        e_fake = float(((fake - fake_ref).double().pow(2).mean().sqrt() / fake_ref.double().pow(2).mean().sqrt()))
        d_loss, g_loss = tr.step(xd, yd, labels=labels)
        res[tag] = (d_loss.item(), g_loss.item(), e_fake)
        print(f"c5 2304x1536x32 N=2 {tag}: generator output rms-rel {e_fake:.2e} | d_loss {res[tag][0]:.5f} (oracle {d_ref:.5f}, "
              f"rel {_rel(res[tag][0], d_ref):.2e})  g_loss {res[tag][1]:.5f} (oracle {g_ref:.5f}, rel {_rel(res[tag][1], g_ref):.2e})")
        del tr, G, D, fake
        torch.cuda.empty_cache()
        if prev is not None:
            _ops.set_fp8_min_work(prev)
    b16, f8, f8a = res["bf16"], res["fp8"], res["fp8_all"]
    assert all(np.isfinite(v) for v in f8 + f8a)
    assert _rel(b16[0], d_ref) <= 2e-1 and _rel(b16[1], g_ref) <= 2e-1 and b16[2] <= 2e-1     # measured 1.0e-2, 9.1e-3
    # fp8 with the default layer rule (ops.fp8_layer_ok: operands in fp8 where the quantisation pass pays -- the exit flow,
    # the ASPP, the decoder's 3 x 3 layers), 2 x measured: generator output 1.54e-1 (bf16: 1.18e-1), d_loss 1.25e-1,
    # g_loss 1.09e-1.  With EVERY eligible layer on fp8 operands (BGAMD_FP8_MIN_WORK=0) the same run measured d_loss
    # 3.0e-1, g_loss 5.5e-1: the docstring's accumulation through ~70 layers.
    assert _rel(f8[0], d_ref) <= 2.5e-1 and _rel(f8[1], g_ref) <= 2.2e-1 and f8[2] <= 3.1e-1
    # the hard case pinned, not just described: every eligible layer on fp8 operands; bounds 2 x the
    # round-3 measurement (d_loss 3.0e-1, g_loss 5.5e-1; generator output bounded like the default rule's, x 2).  End to end
    # the fp8 path is LOOSE by construction -- parity of the path rests on the kernel- and Block-level pins of tests/test_fp8_gpu.py.
    assert _rel(f8a[0], d_ref) <= 6.0e-1 and _rel(f8a[1], g_ref) <= 1.1 and f8a[2] <= 6.2e-1
