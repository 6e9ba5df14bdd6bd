"""CPU: the oracle (oracle/esrgan_oracle.py, oracle/conv_ref.c) against the committed reference-generated
fixtures (tests/golden, made by tools/make_golden.py from the imported reference)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import esrgan_oracle as O  # noqa: E402
from oracle import c_ref  # noqa: E402


def close(a, b, tol=1e-5):
    a = torch.as_tensor(a); b = torch.as_tensor(b)
    return ((a - b).abs().max() <= tol * max(1.0, b.abs().max().item())).item()


def test_G1_pixel_shuffle_bit_exact(golden_dir):
    d = np.load(os.path.join(golden_dir, "G1_pixel_shuffle.npz"))
    assert torch.equal(O.pixel_shuffle(torch.from_numpy(d["x"])), torch.from_numpy(d["y"]))
    assert np.array_equal(c_ref.pixel_shuffle2(d["x"]), d["y"])


def test_G2_conv(golden_dir):
    d = np.load(os.path.join(golden_dir, "G2_conv_lrelu.npz"))
    for (ci, co) in [(1, 64), (64, 64), (320, 64), (64, 256), (64, 1)]:
        sd = O.closed_form_fill({"weight": torch.empty(co, ci, 3, 3), "bias": torch.empty(co)})
        x = torch.from_numpy(d[f"x_{ci}_{co}"])
        y = O.lrelu(O.conv3x3(x, sd["weight"], sd["bias"]), 0.01)
        assert close(y, d[f"y_{ci}_{co}"])
        yc = c_ref.conv3x3(x.numpy(), sd["weight"].numpy(), sd["bias"].numpy(), 1, 0.01)
        assert close(yc, d[f"y_{ci}_{co}"])


def test_c_oracle_stride2_and_sumpool(golden_dir):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 5, 9, 11, generator=g); w = torch.rand(7, 5, 3, 3, generator=g) - 0.5; b = torch.rand(7, generator=g)
    assert close(c_ref.conv3x3(x.numpy(), w.numpy(), b.numpy(), 2, 0.2), O.lrelu(O.conv3x3(x, w, b, 2), 0.2))
    d = np.load(os.path.join(golden_dir, "G9_sumpool.npz"))
    assert close(c_ref.sum_pool(d["x"], 4), d["y4"]) and close(O.sum_pool(torch.from_numpy(d["x"]), 2), d["y2"])


def _gen_case(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    c, f, r, u = [int(v) for v in d["cfg"]]
    shapes = O.generator_state_shapes(c, f, r, u)
    sd = O.closed_form_fill({k: torch.ones(s) if k in ("power", "multiplier") else torch.empty(s) for k, s in shapes.items()},
                            gain=float(d["gain"]))
    assert len(sd) == int(d["n_keys"])
    sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
    x = torch.from_numpy(d["x"])
    y, srs = O.generator_forward(sdo, x, r, u, float(d["res_scale"]), training=True)
    assert close(y, d["y_train"]) and close(srs, d["srs"])
    loss = O.warmup_loss(y, torch.from_numpy(d["target"]))
    assert abs(loss.item() - float(d["loss"])) < 1e-6
    loss.backward()
    for k in d.files:
        if k.startswith("grad."):
            assert close(sdo[k[5:]].grad, d[k], 2e-5), k
    with torch.no_grad():
        ye, _ = O.generator_forward(sd, x, r, u, float(d["res_scale"]), training=False)
    assert close(ye, d["y_eval"])


def test_G4_generator(golden_dir):
    _gen_case(golden_dir, "G4_gen_f16_r1_u2")


def test_G4b_generator_3ch(golden_dir):
    _gen_case(golden_dir, "G4b_gen_c3_f16_r1_u1")


def test_G5_config0(golden_dir):
    _gen_case(golden_dir, "G5_config0")
    keys = open(os.path.join(golden_dir, "G5_state_keys.txt")).read().split()
    assert keys == list(O.generator_state_shapes(1, 32, 2, 1).keys())


def test_G6_res_scale(golden_dir):
    _gen_case(golden_dir, "G6_gen_resscale01")


def test_G3_dense_block(golden_dir):
    d = np.load(os.path.join(golden_dir, "G3_drb16.npz"))
    shapes = {}
    for k in range(1, 6):
        shapes[f"b{k}.0.weight"] = (16, 16 * k, 3, 3); shapes[f"b{k}.0.bias"] = (16,)
    sd = O.closed_form_fill({k: torch.empty(s) for k, s in shapes.items()})
    sdo = {"p." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(d["x"]).requires_grad_(True)
    y = O.dense_residual_block(sdo, "p", x)
    assert close(y, d["y"])
    y.backward(torch.from_numpy(d["gout"]))
    assert close(x.grad, d["dx"])
    for k in shapes:
        assert close(sdo["p." + k].grad, d["grad." + k], 2e-5), k


def test_G7_discriminator(golden_dir):
    d = np.load(os.path.join(golden_dir, "G7_discriminator.npz"))
    shapes = O.discriminator_state_shapes(1, (16, 32, 32, 64))
    assert list(shapes.keys()) == open(os.path.join(golden_dir, "G7_state_keys.txt")).read().split()
    sd = O.closed_form_fill({k: torch.empty(s) for k, s in shapes.items()}, gain=2.0)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    gt, gen, eps = [torch.from_numpy(d[k]) for k in ("gt", "gen", "eps")]
    assert close(O.discriminator_forward(sdo, gt), d["pred_real"])
    loss, gp = O.d_phase_loss(sdo, gt, gen, eps, 0.01)
    assert abs(gp.item() - float(d["gp"])) < 1e-7 and abs(loss.item() - float(d["loss"])) < 1e-6
    loss.backward()
    for k in shapes:
        assert close(sdo[k].grad, d["grad." + k], 2e-5), k
    for shp in [(1, 80, 80), (1, 75, 75), (1, 256, 256)]:
        o = O.discriminator_output_shape(shp)
        assert o == (1, -(-shp[1] // 16), -(-shp[2] // 16))


def test_G11_loss_heads(golden_dir):
    """The oracle's restatement of softgreater / nnz_mask / get_hitogram / DiffableHistogram / KLD_hist and of the way
    esrgan.py:522-547 composes them, against values and gradients produced by the reference's own functions
    (tools/make_golden_heads.py)."""
    d = np.load(os.path.join(golden_dir, "G11_loss_heads.npz"))
    gt = torch.from_numpy(d["gt"]); f = int(d["factor"])

    def leaf():
        return torch.from_numpy(d["gen"]).clone().requires_grad_(True)
    for name, fn in (("softgreater", lambda t: O.softgreater(t, 0.2, 30, 0.05)), ("nnz_mask", lambda t: O.nnz_mask(t, 7.0))):
        x = leaf(); y = fn(x)
        gx, = torch.autograd.grad((y * torch.from_numpy(d[name + ".w"])).sum(), x)
        assert close(y.detach(), d[name + ".y"]) and close(gx, d[name + ".dx"])
    for tag, sig in (("nnz", 50000.0), ("nnz_soft", 3.0)):
        x = leaf()
        cnt = O.softgreater(x, 0, sig).sum(1).sum(1).sum(1)
        tgt = (gt > 0).sum(1).sum(1).sum(1).float()
        loss = torch.nn.functional.mse_loss(cnt, tgt)
        gx, = torch.autograd.grad(loss, x)
        assert close(cnt.detach(), d[tag + ".count"]) and close(tgt, d[tag + ".target"]) and close(loss.detach(), d[tag + ".loss"]) and close(gx, d[tag + ".dx"])
    for tag, sig in (("mask", 5e4), ("mask_soft", 2.0)):
        x = leaf()
        loss = (O.nnz_mask(x, sig) - O.nnz_mask(gt, sig)).abs().mean()
        gx, = torch.autograd.grad(loss, x)
        assert close(loss.detach(), d[tag + ".loss"], 1e-6) and close(gx, d[tag + ".dx"], 1e-6)
    for tag, thr, sig in (("hit", 0.5, 500.0), ("hit_soft", 0.5, 2.0), ("hit_mean", 0.5, -1.0)):
        x = leaf()
        gh, th = O.get_hitogram(x, f, thr, sig), O.get_hitogram(gt, f, thr, sig)
        loss = torch.nn.functional.mse_loss(gh, th)
        gx, = torch.autograd.grad(loss, x)
        assert close(gh.detach(), d[tag + ".gen"], 1e-6) and close(th, d[tag + ".target"], 1e-6)
        assert abs(loss.item() - float(d[tag + ".loss"])) <= 1e-5 * float(d[tag + ".loss"])
        assert (gx - torch.from_numpy(d[tag + ".dx"])).abs().max().item() <= 1e-5 * np.abs(d[tag + ".dx"]).max()
    edges = O.hist_binedges(d["hist.nnz"], 6, 1.0)
    assert np.allclose(edges, d["hist.edges"], rtol=1e-6, atol=0)
    for tag, sig in (("hist", 500.0), ("hist_soft", 4.0)):
        x = leaf()
        gen_hist = O.diffable_histogram(x[x > 0], d["hist.edges"], sig)
        real_hist = O.diffable_histogram(gt[gt > 0], d["hist.edges"], sig)
        loss = O.kld_hist(gen_hist, real_hist, d["hist.edges"])
        gx, = torch.autograd.grad(loss, x)
        assert close(gen_hist.detach(), d[tag + ".gen"]) and close(real_hist, d[tag + ".real"])
        assert close(loss.detach(), d[tag + ".loss"]) and (gx - torch.from_numpy(d[tag + ".dx"])).abs().max().item() <= 1e-5 * np.abs(d[tag + ".dx"]).max()
        assert close(O.diffable_histogram(x.detach(), d["hist.edges"], sig), d[tag + ".all"])


def test_G14_sparse_jet_decode(golden_dir):
    """O.extract / cutters / sparse_jet_item against images decoded by the reference's datasets.py code path
    (tools/make_golden_jets.py): duplicates accumulate in list order, the first zero energy ends the walk."""
    d = np.load(os.path.join(golden_dir, "G14_sparse_jets.npz"))
    eta, phi, f, L = [int(v) for v in d["cfg"]]
    for tag, thr, nh, pre in (("plain", None, None, 1), ("thres", 1.5, None, 1), ("nhard", None, 4, 1), ("pre2", None, None, 2)):
        for e, row in enumerate(d["rows_" + tag]):
            lr, hr = O.sparse_jet_item(torch.from_numpy(row), eta, phi, f, pre, thr, nh)
            assert torch.equal(hr, torch.from_numpy(d["hr_" + tag][e])) and torch.equal(lr, torch.from_numpy(d["lr_" + tag][e])), (tag, e)
