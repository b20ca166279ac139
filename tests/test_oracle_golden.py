"""The oracle (oracle/gan_oracle.py) against vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only.  Tolerance: 1e-5 relative
(both sides are fp32 PyTorch CPU kernels)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import gan_oracle as orc

torch.set_num_threads(min(8, os.cpu_count() or 1))
RTOL = 1e-5


def _close(a, b, rtol=RTOL, atol_scale=1e-5, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max()
    assert err <= rtol * scale + atol_scale * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _cs(v):
    v = v.detach().double()
    return np.array([v.sum().item(), v.abs().sum().item(), (v * v).sum().item()])


def test_state_dict_keys(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    g = orc.generator_spec(16, 16, 0, "batch")
    assert [[k, list(s)] for k, s, _ in g] == ref["generator_c16"]
    assert len(g) == 513
    d = orc.discriminator_spec(16, 64, 64, "batch")
    assert [[k, list(s)] for k, s, _ in d] == ref["discriminator_c16_64x64"]
    assert len(d) == 459


@pytest.mark.parametrize("tag", ["blk_a", "blk_b", "blk_c", "blk_d"])
def test_block(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "modules.npz"))
    kw = json.loads(str(z[tag + "::cfg"]))
    cfg = dict(name="b", cin=kw["inplanes"], cout=kw["planes"], reps=kw["reps"], stride=kw["stride"],
               dil=kw["dilation"], start_relu=kw["start_with_relu"], grow_first=kw["grow_first"], is_last=kw["is_last"])
    P = {k[len(tag) + 6:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith(tag + "::sd::")}
    keys = [k for k in P if P[k].dtype.is_floating_point and "running" not in k]
    for k in keys:
        P[k].requires_grad_(True)
    x = torch.from_numpy(z[tag + "::x"]).requires_grad_(True)
    y = orc.block(P, "", cfg, x, orc.NormCtx("batch", True))
    _close(y.detach().numpy(), z[tag + "::y"], what="y")
    y.backward(torch.from_numpy(z[tag + "::go"]))
    _close(x.grad.numpy(), z[tag + "::dx"], what="dx")
    for k in keys:
        _close(P[k].grad.numpy(), z[f"{tag}::grad::{k}"], rtol=1e-4, what=k)
    # the reference activates its input in place when the block starts with the relu
    expect = orc.lrelu(x.detach()) if cfg["start_relu"] else x.detach()
    _close(expect.numpy(), z[tag + "::x_after"], what="input aliasing")


@pytest.mark.parametrize("tag", ["c4_64x64", "c8_40x56", "deconv_c4_19x37", "deconv1x_c4_19x37"])
def test_generator(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, f"generator_{tag}.npz"))
    m = json.loads(str(z["meta"]))
    spec = orc.generator_spec(m["c"], m["c"], 0, "batch", upsampler=m.get("upsampler", "Interpolate"))
    P = orc.fill_state(spec, m["seed"])
    keys = orc.trainable_keys(spec)
    for k in keys:
        P[k].requires_grad_(True)
    x, y = orc.synthetic_fields(m["n"], m["c"], m["h"], m["w"], m["field_seed"])
    out = orc.generator(P, x, orc.NormCtx("batch", True))
    _close(out.detach().numpy(), z["out"], what="out")
    loss = (out - y).abs().mean()
    _close(loss.item(), z["loss"], what="loss")
    loss.backward()
    ref = dict(zip([str(k) for k in z["grad_keys"]], z["grad_cs"]))
    assert list(ref.keys()) == keys
    for k in keys:
        got = _cs(P[k].grad)
        # abs-sum and square-sum are robust checksums; 1e-4: different reduction order in autograd
        assert abs(got[1] - ref[k][1]) <= 2e-4 * ref[k][1] + 1e-9, (k, got, ref[k])
        assert abs(got[2] - ref[k][2]) <= 4e-4 * ref[k][2] + 1e-12, (k, got, ref[k])
    for k in z.files:
        if k.startswith("grad::"):
            _close(P[k[6:]].grad.numpy(), z[k], rtol=2e-4, what=k)
        if k.startswith("buf::"):
            _close(P[k[5:]].numpy(), z[k], what=k)
    with torch.no_grad():
        out_eval = orc.generator(P, x, orc.NormCtx("batch", False))
    _close(out_eval.numpy(), z["out_eval"], rtol=1e-4, what="out_eval")


@pytest.mark.parametrize("tag", ["c4_64x64_bn", "c8_40x56_bn", "c4_64x64_in"])
def test_discriminator(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, f"discriminator_{tag}.npz"))
    m = json.loads(str(z["meta"]))
    spec = orc.discriminator_spec(m["c"], m["h"], m["w"], m["norm"])
    P = orc.fill_state(spec, m["seed"])
    keys = orc.trainable_keys(spec)
    for k in keys:
        P[k].requires_grad_(True)
    x, _ = orc.synthetic_fields(m["n"], m["c"], m["h"], m["w"], m["field_seed"])
    x.requires_grad_(True)
    logits, pred = orc.discriminator(P, x, orc.NormCtx(m["norm"], True))
    _close(logits.detach().numpy(), z["logits"], rtol=1e-4, what="logits")
    _close(pred.detach().numpy(), z["pred"], rtol=1e-4, what="pred")
    tgt = torch.linspace(0.1, 0.9, m["n"]).reshape(-1, 1)
    loss = orc.bce_logits(logits, tgt)
    _close(loss.item(), z["loss"], rtol=1e-4, what="loss")
    loss.backward()
    # InstanceNorm: torch's fp32 instance-norm backward is itself ~1e-2 away from an
    # fp64 evaluation of the same graph on this net (measured: the oracle in fp32 and
    # in fp64 agree to 7e-6, F.instance_norm in fp64 agrees with the oracle to 4e-14,
    # the fp32 reference differs from all three by 1.0e-2), so the bound for the
    # instance-norm case is the reference's own noise, not the oracle's.
    g_tol = 2e-2 if m["norm"] == "instance" else 5e-4
    _close(x.grad.numpy(), z["dx"], rtol=g_tol, what="dx")
    ref = dict(zip([str(k) for k in z["grad_keys"]], z["grad_cs"]))
    for k in keys:
        got = _cs(P[k].grad)
        assert abs(got[1] - ref[k][1]) <= g_tol * ref[k][1] + 1e-9, (k, got, ref[k])


def test_gan_losses(golden_dir):
    z = np.load(os.path.join(golden_dir, "losses.npz"))
    for seed in (0, 7, 123, 999):
        for mode in ("ModifiedMinMax", "Wasserstein"):
            p = f"{mode}_{seed}::"
            lr_, lf_ = torch.from_numpy(z[p + "logits_real"]), torch.from_numpy(z[p + "logits_fake"])
            if mode == "ModifiedMinMax":
                torch.manual_seed(seed)
                lab_f, lab_r, swap = orc.draw_d_labels(4)
                np.testing.assert_array_equal(lab_f.numpy(), z[p + "label_fake"])  # bit-exact draws
                np.testing.assert_array_equal(lab_r.numpy(), z[p + "label_real"])
                assert swap == bool(z[p + "swap_u"] < 0.05)
                d = orc.gan_d_loss(mode, lr_, lf_, lab_f, lab_r, swap)
            else:
                d = orc.gan_d_loss(mode, lr_, lf_)
            _close(d.item(), z[p + "d_loss"], what=p + "d")
            _close(orc.gan_g_loss(mode, lf_).item(), z[p + "g_loss"], what=p + "g")
    torch.manual_seed(int(z["swap_seed"]))
    lab_f, lab_r, swap = orc.draw_d_labels(4)
    assert swap
    d = orc.gan_d_loss("ModifiedMinMax", torch.from_numpy(z["swap::logits_real"]),
                       torch.from_numpy(z["swap::logits_fake"]), lab_f, lab_r, swap)
    _close(d.item(), z["swap::d_loss"], what="swapped d_loss")
    p, t, w = (torch.from_numpy(z["l1w::" + k]) for k in "ptw")
    _close(orc.l1_weighted(p, t, w).item(), z["l1w::plain"])
    _close(orc.l1_weighted(p, t, w, normalize=True).item(), z["l1w::normalized"])


def test_gradient_penalty(golden_dir):
    z = np.load(os.path.join(golden_dir, "gradient_penalty.npz"))
    m = json.loads(str(z["meta"]))
    assert not bool(z["has_graph"])  # the reference's penalty is a constant
    spec = orc.discriminator_spec(m["c"], m["h"], m["w"], "batch")
    P = orc.fill_state(spec, m["seed"])
    fake, real = orc.synthetic_fields(m["n"], m["c"], m["h"], m["w"], m["field_seed"])
    torch.manual_seed(m["seed"])
    eta = torch.rand((m["n"], 1, 1, 1))
    np.testing.assert_array_equal(eta.numpy(), z["eta"])
    gp = orc.gradient_penalty(P, fake, real, eta, orc.NormCtx("batch", True))
    assert not gp.requires_grad
    _close(gp.item(), z["gp"], rtol=1e-4, what="gp")
    _close(P["xception_features.bn1.running_mean"].numpy(), z["bn1_rm_after"], what="bn stats updated by GP forward")


@pytest.mark.parametrize("tag", ["mmm", "wgp"])
def test_one_iteration_full_nets(golden_dir, tag):
    """One D-step + G-step of the loop body (train_gan.py:244-298) on the full
    nets.  d_loss is computed before any update, g_loss after D's Adam step, the
    parameter checksums after both Adam steps."""
    z = np.load(os.path.join(golden_dir, f"trajectory_{tag}.npz"))
    m = json.loads(str(z["meta"]))
    c, h, w, n = m["c"], m["h"], m["w"], m["n"]
    gspec = orc.generator_spec(c, c, 0, "batch")
    dspec = orc.discriminator_spec(c, h, w, "batch")
    PG, PD = orc.fill_state(gspec, m["seed"]), orc.fill_state(dspec, m["seed"] + 1)
    st = orc.GANStep(PG, PD, orc.trainable_keys(gspec), orc.trainable_keys(dspec), "batch", m["mode"],
                     eps=m["adam_eps"])
    torch.manual_seed(m["torch_seed"])
    x, y = orc.synthetic_fields(n, c, h, w, m["field_seed0"])
    d, g = st.step(x, y)
    assert abs(d - z["d_loss"][0]) <= 1e-5 * abs(z["d_loss"][0]), (d, z["d_loss"][0])
    assert abs(g - z["g_loss"][0]) <= 1e-4 * abs(z["g_loss"][0]), (g, z["g_loss"][0])
    for k in z.files:
        if k.startswith("G::model.") or k.startswith("D::xception") or k.startswith("D::linear"):
            P = st.PG if k[0] == "G" else st.PD
            got = _cs(P[k[3:]])
            assert abs(got[1] - z[k][0][1]) <= 1e-5 * z[k][0][1], (k, got, z[k][0])
    _close(st.PG["model.xception_features.bn1.running_mean"].numpy(), z["G::bn1.running_mean"][0], rtol=1e-5)
    _close(st.PG["model.xception_features.bn1.running_var"].numpy(), z["G::bn1.running_var"][0], rtol=1e-5)
    _close(st.PD["xception_features.bn1.running_mean"].numpy(), z["D::bn1.running_mean"][0], rtol=1e-5)
    # two G forwards and 3 (4 with GP) D forwards per iteration each update the BN statistics
    assert int(st.PG["model.xception_features.bn1.num_batches_tracked"]) == int(z["G::bn1.nbt"][0]) == 2
    assert int(st.PD["xception_features.bn1.num_batches_tracked"]) == int(z["D::bn1.nbt"][0]) == (3 if tag == "mmm" else 4)


def test_c1_plumbing_three_iterations(golden_dir):
    """BASELINE.json configs[0]: 1-layer G + 1-layer D, 64x64x4, batch 2, three
    loop iterations with Adam(1e-4, eps 1e-8, wd 1e-5): pins update order, Adam
    arithmetic, label-draw order and BN double-updates over several steps."""
    import torch.nn.functional as F
    z = np.load(os.path.join(golden_dir, "c1_plumbing.npz"))
    m = json.loads(str(z["meta"]))
    P = {k[6:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("init::")}
    P["g.1.running_mean"], P["g.1.running_var"] = torch.zeros(m["c"]), torch.ones(m["c"])
    P["dstem.1.running_mean"], P["dstem.1.running_var"] = torch.zeros(8), torch.ones(8)
    gk = ["g.0.weight", "g.1.weight", "g.1.bias"]
    dk = ["dstem.0.weight", "dstem.1.weight", "dstem.1.bias", "dlin.weight", "dlin.bias"]
    g_opt, d_opt = orc.Adam(gk), orc.Adam(dk)

    def G(Q, x):
        return orc.lrelu(F.batch_norm(F.conv2d(x, Q["g.0.weight"], None, 1, 1), P["g.1.running_mean"],
                                      P["g.1.running_var"], Q["g.1.weight"], Q["g.1.bias"], True, 0.1, 1e-5))

    def D(Q, x):
        f = orc.lrelu(F.batch_norm(F.conv2d(x, Q["dstem.0.weight"], None, 2, 1), P["dstem.1.running_mean"],
                                   P["dstem.1.running_var"], Q["dstem.1.weight"], Q["dstem.1.bias"], True, 0.1, 1e-5))
        return F.linear(f.reshape(f.shape[0], -1), Q["dlin.weight"], Q["dlin.bias"])

    def leaves(keys):
        Q = dict(P)
        for k in keys:
            Q[k] = P[k].detach().requires_grad_(True)
        return Q

    torch.manual_seed(m["torch_seed"])
    for s in range(m["steps"]):
        x, y = orc.synthetic_fields(m["n"], m["c"], m["h"], m["w"], m["field_seed0"] + s)
        with torch.no_grad():
            fake = G(P, x)
        Q = leaves(dk)
        lf, lr_, sw = orc.draw_d_labels(m["n"])
        dl = orc.gan_d_loss("ModifiedMinMax", D(Q, y), D(Q, fake), lf, lr_, sw)
        grads = torch.autograd.grad(dl, [Q[k] for k in dk])
        d_opt.step(P, dict(zip(dk, grads)))
        Q = leaves(gk)
        fake = G(Q, x)
        gl = orc.gan_g_loss("ModifiedMinMax", D(P, fake)) + (fake - y).abs().mean()
        grads = torch.autograd.grad(gl, [Q[k] for k in gk])
        g_opt.step(P, dict(zip(gk, grads)))
        assert abs(dl.item() - z["d_loss"][s]) <= 2e-5 * abs(z["d_loss"][s]), (s, dl.item(), z["d_loss"][s])
        assert abs(gl.item() - z["g_loss"][s]) <= 2e-5 * abs(z["g_loss"][s]), (s, gl.item(), z["g_loss"][s])
    _close(P["g.0.weight"].numpy(), z["final::g.0.weight"], rtol=1e-5)
    assert abs(_cs(P["dlin.weight"])[1] - z["final::dlin.weight_cs"][1]) <= 1e-5 * z["final::dlin.weight_cs"][1]


def _lamb_fixture(seed=0):
    g = torch.Generator().manual_seed(seed)
    P = {"a": torch.randn(5, 7, generator=g), "b": torch.randn(11, generator=g), "c": torch.randn(3, 2, 3, 3, generator=g)}
    grads = [{k: 0.05 * torch.randn(v.shape, generator=g) for k, v in P.items()} for _ in range(4)]
    return P, grads


def test_lamb_is_adam_below_the_clipping_norm_without_decay():
    """The part of the LAMB restatement (oracle.Lamb; apex is not in the reference tree) that CAN be pinned: with
    weight_decay = 0 the trust ratio is off, and with ||g|| <= max_grad_norm nothing is clipped -- the step is then
    torch.optim.Adam's, moments included."""
    P, grads = _lamb_fixture()
    Q = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = torch.optim.Adam(list(Q.values()), lr=1e-3, eps=1e-6)
    lam = orc.Lamb(list(P), lr=1e-3, eps=1e-6, weight_decay=0.0)
    for g in grads:
        assert sum(float((x.double() ** 2).sum()) for x in g.values()) < 1.0
        for k in Q:
            Q[k].grad = g[k].clone()
        ref.step()
        lam.step(P, g)
        for i, k in enumerate(P):
            assert (P[k] - Q[k].detach()).abs().max().item() <= 2e-7
            st = ref.state[Q[k]]
            assert (lam.m[k] - st["exp_avg"]).abs().max().item() <= 1e-8 and (lam.v[k] - st["exp_avg_sq"]).abs().max().item() <= 1e-9


def test_lamb_clips_by_the_global_norm_and_scales_by_the_trust_ratio():
    """Gradients k times larger than the clipping norm give the step of gradients AT the norm (the global norm runs over all
    tensors); with weight decay each tensor moves by lr * ||p|| / ||u|| * u with u = adam direction + wd * p."""
    P, grads = _lamb_fixture(1)
    g = grads[0]
    gn = sum(float((x.double() ** 2).sum()) for x in g.values()) ** 0.5
    at = {k: x / gn for k, x in g.items()}                 # global norm exactly 1 = max_grad_norm
    big = {k: 7.5 * x for k, x in at.items()}
    A, B = {k: v.clone() for k, v in P.items()}, {k: v.clone() for k, v in P.items()}
    orc.Lamb(list(P), lr=1e-2, eps=1e-6, weight_decay=0.01).step(A, at)
    orc.Lamb(list(P), lr=1e-2, eps=1e-6, weight_decay=0.01).step(B, big)
    for k in P:
        assert (A[k] - B[k]).abs().max().item() <= 1e-6
        # first step: m / bc1 = s, v / bc2 = s^2 -> adam direction s / (|s| + eps)
        s_ = at[k]
        u = s_ / (s_.abs() + 1e-6) + 0.01 * P[k]
        want = P[k] - 1e-2 * (P[k].norm() / u.norm()) * u
        assert (A[k] - want).abs().max().item() <= 1e-6
