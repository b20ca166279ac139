"""CPU: the partial-convolution U-Net oracle (oracle/infill3d_oracle.py) against vectors produced by the
reference's own modules (tests/golden/infill3d_c2_32x24x40.npz; make_golden.py infill3d)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import infill3d_oracle as oi


def _close(a, b, rtol=2e-5, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b).max()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _cs(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "infill3d_c2_32x24x40.npz"))


def test_unet3d(z):
    m = json.loads(str(z["meta"]))
    spec = oi.unet3d_spec(m["cin"], m["cout"], m["g_layers"])
    P = oi.fill_state(spec, m["g_seed"])
    keys = oi.trainable_keys(spec)
    for k in keys:
        P[k].requires_grad_(True)
    x, gt, mask = oi.synthetic_infill(m["n"], m["cin"], m["d"], m["h"], m["w"], m["field_seed"])
    out, out_mask = oi.unet3d(P, x, mask, m["g_layers"], oi.NormCtx("batch", True))
    _close(out.detach().numpy(), z["g::out"], what="out")
    assert np.array_equal(out_mask.numpy(), z["g::out_mask"])          # masks: bit-exact
    co = m["cout"]
    ld = oi.inpainting_loss(x[:, :co], out, gt[:, :co], mask[:, :co])
    for k in ("hole", "valid", "tv"):
        _close(ld[k].item(), z["g::loss_" + k], what=k)
    (6.0 * ld["hole"] + 1.0 * ld["valid"] + 0.1 * ld["tv"]).backward()
    ref = dict(zip([str(k) for k in z["g::grad_keys"]], z["g::grad_cs"]))
    assert list(ref.keys()) == keys
    for k in keys:
        got = _cs(P[k].grad)
        assert abs(got[2] - ref[k][2]) <= 1e-3 * ref[k][2] + 1e-12, (k, got, ref[k])
    for k in z.files:
        if k.startswith("g::grad::"):
            _close(P[k[9:]].grad.numpy(), z[k], rtol=5e-4, what=k)
        if k.startswith("g::buf::"):
            _close(P[k[8:]].numpy(), z[k], what=k)
    # the encoder's mask chain alone
    with torch.no_grad():
        mm = mask
        for i in range(1, m["g_layers"] + 1):
            _, mm = oi.partial_conv3d(torch.zeros_like(mm), mm, P[f"enc_{i}.conv.weight"], None, 2, 1)
            assert np.array_equal(mm[:, :1].numpy(), z[f"g::enc_mask_{i}"]), i


@pytest.mark.parametrize("tag", ["tri", "drop", "tridrop"])
def test_unet3d_options(golden_dir, tag):
    """upsampling_mode='trilinear' and PCDropout3d (infill3d.py:115-135, :217-222): the oracle against the reference's
    modules, the dropout draws being the ones recorded from the reference's nn.Dropout3d (make_golden.py infill3d_options)."""
    z = np.load(os.path.join(golden_dir, "infill3d_options_c2_18x10x14.npz"))
    m = json.loads(str(z["meta"]))
    spec = oi.unet3d_spec(m["cin"], m["cout"], m["g_layers"])
    P = oi.fill_state(spec, m["g_seed"])
    keys = oi.trainable_keys(spec)
    for k in keys:
        P[k].requires_grad_(True)
    x, gt, mask = oi.synthetic_infill(m["n"], m["cin"], m["d"], m["h"], m["w"], m["field_seed"])
    nk = int(z[tag + "::n_keeps"])
    assert nk == (0 if tag == "tri" else 2 * m["g_layers"])
    keeps = [torch.from_numpy(z[f"{tag}::keep_{i}"]) for i in range(nk)]
    if nk:
        assert any((k_ == 0).any() for k_ in keeps), "the recorded draw drops nothing: the case would not test dropout"
    out, out_mask = oi.unet3d(P, x, mask, m["g_layers"], oi.NormCtx("batch", True), upsampling_mode="nearest" if tag == "drop" else "trilinear",
                              dropout_p=m["p_drop"] if nk else 0.0, keeps=keeps)
    _close(out.detach().numpy(), z[tag + "::out"], what="out")
    assert np.array_equal(out_mask.numpy(), z[tag + "::out_mask"])
    co = m["cout"]
    ld = oi.inpainting_loss(x[:, :co], out, gt[:, :co], mask[:, :co])
    (6.0 * ld["hole"] + 1.0 * ld["valid"] + 0.1 * ld["tv"]).backward()
    ref = dict(zip([str(k) for k in z[tag + "::grad_keys"]], z[tag + "::grad_cs"]))
    assert list(ref.keys()) == keys
    for k in keys:
        got = _cs(P[k].grad)
        assert abs(got[2] - ref[k][2]) <= 1e-3 * ref[k][2] + 1e-12, (k, got, ref[k])
    pre = tag + "::grad::"
    for k in z.files:
        if k.startswith(pre):
            _close(P[k[len(pre):]].grad.numpy(), z[k], rtol=5e-4, what=k)


def test_disc3d(z):
    m = json.loads(str(z["meta"]))
    spec = oi.disc3d_spec(m["cout"], m["d_layers"])
    P = oi.fill_state(spec, m["d_seed"])
    keys = oi.trainable_keys(spec)
    for k in keys:
        P[k].requires_grad_(True)
    _, gt, mask = oi.synthetic_infill(m["n"], m["cin"], m["d"], m["h"], m["w"], m["field_seed"])
    co = m["cout"]
    x = gt[:, :co].clone().requires_grad_(True)
    logits, _ = oi.disc3d(P, x, mask[:, :co], m["d_layers"], oi.NormCtx("batch", True))
    _close(logits.detach().numpy(), z["d::logits"], rtol=1e-4, what="logits")
    tgt = torch.linspace(0.1, 0.9, m["n"]).reshape(-1, 1)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, tgt)
    _close(loss.item(), z["d::loss"], rtol=1e-4, what="loss")
    loss.backward()
    _close(x.grad.numpy(), z["d::dx"], rtol=2e-3, what="dx")
    ref = dict(zip([str(k) for k in z["d::grad_keys"]], z["d::grad_cs"]))
    # the last encoder layer runs but does not reach the logits (infill3d_gan.py:58-60): no gradient there
    used = [k for k in keys if P[k].grad is not None]
    assert list(ref.keys()) == used and not any(k.startswith(f"enc_{m['d_layers']}.") for k in used)
    for k in used:
        got = _cs(P[k].grad)
        assert abs(got[2] - ref[k][2]) <= 5e-3 * ref[k][2] + 1e-12, (k, got, ref[k])


def test_oracle_loop_matches_reference_trajectory(golden_dir):
    """Seven iterations of infill3d_gan_module.py:272-375 (warm-up, both, D only, G only -- all four flag states):
    the oracle's InfillGANStep against the losses, accuracies, flags, weights and running statistics recorded from
    the reference's modules driven by torch.optim.AdamW (make_golden.py infill3d_trajectory)."""
    import json
    import torch
    z = np.load(os.path.join(golden_dir, "trajectory_infill3d.npz"))
    m = json.loads(str(z["meta"]))
    gspec, dspec = oi.unet3d_spec(1 + m["noise_dims"], 1, m["g_layers"]), oi.disc3d_spec(1 + m["noise_dims"], m["d_layers"])
    st = oi.InfillGANStep(oi.fill_state(gspec, m["g_seed"]), oi.fill_state(dspec, m["d_seed"]), oi.trainable_keys(gspec),
                          oi.trainable_keys(dspec), m["g_layers"], m["d_layers"], m["weights"], m["loss_type"], m["warmup"],
                          m["acc_min"], m["acc_max"], lr_g=m["lr"], lr_d=m["lr"], eps=m["adam_eps"], weight_decay=m["wd"], decoupled=True)
    seen = set()
    # The first three iterations agree to fp32 rounding (1e-5 bound, measured 1e-7); from the fourth on AdamW's
    # sign-like early updates (lr 1e-3) have amplified that rounding, measured 1e-5 -> 6e-4 -> 5e-4: bound 3e-3.
    for s in range(m["steps"]):
        tol = 1e-5 if s < 3 else 3e-3
        x, gt, mask = oi.synthetic_infill(m["n"], 1, m["d"], m["h"], m["w"], m["field_seed0"] + s)
        noise = torch.randn((m["n"], m["noise_dims"], m["d"], m["h"], m["w"]), generator=torch.Generator().manual_seed(m["noise_seed0"] + s))
        labels = (torch.from_numpy(z["labels_fake"][s]), torch.from_numpy(z["labels_real"][s]), bool(z["swap"][s]))
        d_loss, g_loss = st.step(x, gt, mask, noise, labels)
        assert st.last_flags == (bool(z["train_g"][s]), bool(z["train_d"][s]))
        seen.add(st.last_flags)
        assert st.d_acc_avg == float(z["d_acc"][s])
        assert abs(d_loss - z["d_loss"][s]) <= tol * abs(z["d_loss"][s]), (s, d_loss, z["d_loss"][s])
        assert abs(g_loss - z["g_loss"][s]) <= tol * abs(z["g_loss"][s]), (s, g_loss, z["g_loss"][s])
        for k in ("hole", "valid", "tv", "adv"):
            if not np.isnan(z[k][s]):
                assert abs(st.last_terms[k] - z[k][s]) <= tol * abs(z[k][s]) + 1e-7, (s, k)
        for tag, P in (("G", st.PG), ("D", st.PD)):
            for key in [k for k in z.files if k.startswith(tag + "::") and k.endswith(("weight", "bias"))]:
                # |sum| and sum of squares; the plain sum of a zero-mean tensor cancels and is not compared
                np.testing.assert_allclose(_cs(P[key[3:]])[1:], z[key][s][1:], rtol=1e-4, err_msg=f"step {s} {key}")
        _close(st.PG["enc_2.bn.running_mean"].numpy(), z["G::enc_2.bn.running_mean"][s], 10 * tol, "G running_mean")
        _close(st.PD["enc_2.bn.running_var"].numpy(), z["D::enc_2.bn.running_var"][s], 10 * tol, "D running_var")
        assert int(st.PG["enc_2.bn.num_batches_tracked"]) == int(z["G::nbt"][s]) == 2 * (s + 1)
        assert int(st.PD["enc_2.bn.num_batches_tracked"]) == int(z["D::nbt"][s]) == 3 * (s + 1)
    assert seen == {(True, False), (True, True), (False, True)}


def _fields2d(m):
    gen = torch.Generator().manual_seed(m["field_seed"])
    gt = torch.randn((m["n"], m["c"], m["h"], m["w"]), generator=gen)
    mask = (torch.rand((m["n"], m["c"], m["h"], m["w"]), generator=gen) > 0.3).float()
    return gt * mask, gt, mask


@pytest.mark.parametrize("fixture", ["infill2d_c2_40x56.npz", "infill2d_bilinear_c2_22x26.npz"])
def test_unet2d(golden_dir, fixture):
    """The 2-D PConvUNet (infill.py) and PartialConv2d against vectors from the reference's modules
    (make_golden.py infill2d / infill2d_bilinear: upsampling_mode 'nearest' and 'bilinear')."""
    z = np.load(os.path.join(golden_dir, fixture))
    m = json.loads(str(z["meta"]))
    spec = oi.unet2d_spec(m["c"], m["c"], m["layers"])
    P = oi.fill_state(spec, m["seed"])
    keys = oi.trainable_keys(spec)
    for k in keys:
        P[k].requires_grad_(True)
    x, gt, mask = _fields2d(m)
    out, out_mask = oi.unet2d(P, x, mask, m["layers"], oi.NormCtx("batch", True), upsampling_mode=m.get("mode", "nearest"))
    _close(out.detach().numpy(), z["out"], what="out")
    assert np.array_equal(out_mask.numpy(), z["out_mask"])
    ld = oi.inpainting_loss(x, out, gt, mask, "l1")
    for k in ("hole", "valid", "tv"):
        _close(ld[k].item(), z["loss_" + k], what=k)
    (6.0 * ld["hole"] + 1.0 * ld["valid"] + 0.1 * ld["tv"]).backward()
    ref = dict(zip([str(k) for k in z["grad_keys"]], z["grad_cs"]))
    assert list(ref.keys()) == keys
    for k in keys:
        assert abs(_cs(P[k].grad)[2] - ref[k][2]) <= 1e-3 * ref[k][2] + 1e-12, k
    for k in z.files:
        if k.startswith("grad::"):
            _close(P[k[6:]].grad.numpy(), z[k], rtol=5e-4, what=k)
        if k.startswith("buf::"):
            _close(P[k[5:]].numpy(), z[k], what=k)
