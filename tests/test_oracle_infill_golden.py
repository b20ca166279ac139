"""CPU: the partial-convolution U-Net oracle (oracle/infill3d_oracle.py) against vectors produced by the
reference's own modules (tests/golden/infill3d_c2_16x16x16.npz; make_golden.py infill3d)."""
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
    return np.load(os.path.join(golden_dir, "infill3d_c2_16x16x16.npz"))


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
