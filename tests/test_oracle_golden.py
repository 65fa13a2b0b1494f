"""CPU: pins oracle/mvsgi_oracle.py against the golden vectors produced by the
reference's own modules (tools/make_goldens.py).  Tolerances: the oracle runs the
same ATen CPU kernels as the reference, so stage outputs agree to float rounding."""
import os

import numpy as np
import pytest
import torch

from golden_cases import SMALL_CASES
from mvs_gi_amd import synth
from oracle import mvsgi_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("name", list(SMALL_CASES))
def test_small_case_matches_reference(golden_dir, name):
    case = SMALL_CASES[name]
    cfg = case["cfg"]
    z = _load(golden_dir, name)
    inp = synth.make_inputs(cfg, seed=case["seed"], batch=case["batch"], grid_kind=case["grid_kind"],
                            grid_mask_dtype=case["grid_mask_dtype"])
    assert synth.digest(inp) == str(z["inputs_sha256"]), "regenerated inputs differ from the golden run's"
    t = O.to_torch(inp)
    for gain in case["gains"]:
        w = synth.make_weights(cfg, seed=case["seed"], gain=gain)
        st = O.hot_path(t["feats"], t["grids"], t["grid_masks"], t["masks"], O.to_torch(w), cfg.builder,
                        cfg.dist_cands, cfg.bf, cfg.interp_scale_factor, cfg.pre_interp, return_stages=True)
        tag = f"g{gain:g}"
        assert _rel(st["inv_dist"].numpy(), z[f"inv_dist_{tag}"]) <= 1e-5
        if "vol_raw" in z and gain == case["gains"][0]:
            assert np.array_equal(st["vol_raw"].numpy(), z["vol_raw"]), "sweep must be bit-exact"
            assert _rel(st["vol"].numpy(), z["vol"]) <= 1e-6
            assert _rel(st["costs"].numpy(), z["costs"]) <= 1e-5
            assert _rel(st["norm_costs"].numpy(), z["norm_costs"]) <= 1e-5


def test_sweep_edge_cases(golden_dir):
    z = _load(golden_dir, "sweep_edges")
    f, g, m = (torch.from_numpy(z[k]) for k in ("feats", "grids", "masks"))
    gm = torch.from_numpy(z["grid_masks_bool"])
    assert np.array_equal(O.sweep_std_masked(f, g, gm, m).numpy(), z["vol_raw_std_bool"])
    assert np.array_equal(O.sweep_std_masked(f, g, gm.float(), m).numpy(), z["vol_raw_std_bool"])
    assert np.array_equal(O.sweep_concat(f, g).numpy(), z["vol_raw_cat"])


def test_sampler_equals_aten_grid_sample():
    rng = np.random.default_rng(5)
    im = torch.from_numpy(rng.standard_normal((2, 3, 7, 9)).astype(np.float32))
    grid = torch.from_numpy(rng.uniform(-1.2, 1.2, (2, 5, 6, 2)).astype(np.float32))
    ref = torch.nn.functional.grid_sample(im, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    assert torch.allclose(O.bilinear_sample_zeros(im, grid), ref, atol=2e-6)


def test_regressor_variants(golden_dir):
    z = _load(golden_dir, "regress_variants")
    costs = torch.from_numpy(z["costs"])
    cands = list(z["dist_cands"])
    for tag, kw in dict(s2_pre=dict(interp_scale_factor=2, pre_interp=True),
                        s0_pre=dict(interp_scale_factor=0, pre_interp=True),
                        s2_post=dict(interp_scale_factor=2, pre_interp=False)).items():
        inv, pr = O.soft_argmin(costs, cands, 96.0, **kw)
        assert _rel(inv.numpy(), z[f"inv_{tag}"]) <= 1e-6
        assert _rel(pr.numpy(), z[f"pr_{tag}"]) <= 1e-6
    inv, _ = O.soft_argmin(costs, list(z["updated_cands"]), 96.0, 2, True)
    assert _rel(inv.numpy(), z["inv_updated"]) <= 1e-6


@pytest.mark.parametrize("name", ["full_G16V", "full_E8", "full_E16-48-96"])
def test_full_size_matches_reference(golden_dir, name):
    """BASELINE.json configs[1] (G16V), configs[0] (E8) and the literal reading of configs[2] ("in48ch/fint96ch" at D = 16) at
    full size: inv_dist only."""
    from golden_cases import FULL_CASES
    case = FULL_CASES[name]
    cfg = case["cfg"]
    z = _load(golden_dir, name)
    inp = synth.make_inputs(cfg, seed=case["seed"], batch=case["batch"], grid_kind=case["grid_kind"],
                            grid_mask_dtype=case["grid_mask_dtype"])
    assert synth.digest(inp) == str(z["inputs_sha256"])
    t = O.to_torch(inp)
    gain = case["gains"][-1]
    w = synth.make_weights(cfg, seed=case["seed"], gain=gain)
    inv = O.hot_path(t["feats"], t["grids"], t["grid_masks"], t["masks"], O.to_torch(w), cfg.builder,
                     cfg.dist_cands, cfg.bf, cfg.interp_scale_factor, cfg.pre_interp)
    assert _rel(inv.numpy(), z[f"inv_dist_g{gain:g}"]) <= 1e-5


def test_feature_extractor_and_end_to_end_match_reference(golden_dir):
    """SimpleFeatExtraction and the imgs -> inv_dist composition
    (torch_only.py:20-36) against the reference's outputs."""
    from mvs_gi_amd.configs import CONFIGS, DIST_8L
    z = _load(golden_dir, "extractor_small")
    cfg = CONFIGS["G16V"].scaled(feat_hw=(16, 64), mask_hw=(64, 256), cv_hw=(8, 32), dist_cands=DIST_8L)
    seed = 8
    imgs = synth.make_images(cfg, seed=seed, batch=2)
    inp = synth.make_inputs(cfg, seed=seed, batch=2)
    assert synth.digest({"imgs": imgs}) == str(z["imgs_sha256"]) and synth.digest(inp) == str(z["inputs_sha256"])
    w = synth.make_weights(cfg, seed=seed)
    w["feature_extractor"] = synth.make_extractor_weights(seed)
    wt = O.to_torch(w)
    with torch.no_grad():
        f = O.feature_extractor(torch.from_numpy(imgs).flatten(0, 1), wt["feature_extractor"])
    assert _rel(f.numpy().reshape(z["feats"].shape), z["feats"]) <= 1e-5
    t = O.to_torch(inp)
    inv = O.full_model(torch.from_numpy(imgs), t["grids"], t["grid_masks"], t["masks"], wt, cfg.builder, cfg.dist_cands)
    assert _rel(inv.numpy(), z["inv_dist"]) <= 1e-5


# ------------------------------------------------------------------ sampling-grid generator
def test_grid_oracle_matches_reference_closed_forms(golden_dir):
    """oracle/grid_oracle.py vs outputs of the reference's torch_cuda_sweep.py (tools/make_grid_goldens.py)."""
    from oracle import grid_oracle as G
    z = _load(golden_dir, "sweep_grids")
    for name in ("g16", "e8_full_sphere"):
        rays = G.rays_panorama(z[name + "_dist"], tuple(z[name + "_lon"]), tuple(z[name + "_lat"]),
                               tuple(int(v) for v in z[name + "_shape"]))
        assert np.array_equal(rays.numpy(), z[name + "_rays"])
        for i, pose in enumerate(z[name + "_poses"]):
            inv = torch.linalg.inv(torch.from_numpy(pose)).to(torch.float32)
            pts = G.transform_points(inv.unsqueeze(0), rays.unsqueeze(0))
            assert np.array_equal(pts.numpy(), z[f"{name}_pts{i}"])
            g, m = G.grid_double_sphere(pts, (-0.203, 0.589, 232.0, 232.0, 611.5, 513.5), (1028, 1224))
            assert np.array_equal(g.numpy(), z[f"{name}_ds_grid{i}"]) and np.array_equal(m.numpy(), z[f"{name}_ds_mask{i}"])
            assert np.array_equal(G.grid_equirect(pts).numpy(), z[f"{name}_eq_grid{i}"])
    g, m = G.grid_double_sphere(torch.from_numpy(z["g16_pts1"]), (0.1, 0.45, 300.0, 310.0, 320.0, 240.0), (480, 640))
    assert np.array_equal(g.numpy(), z["ds2_grid"]) and np.array_equal(m.numpy(), z["ds2_mask"])
    assert abs(G.double_sphere_w2(0.1, 0.45) - float(z["ds2_w2"])) < 1e-15


# ------------------------------------------------------------------ sphere convolution
def _sphere_args(a):
    a = [int(v) for v in a]
    return (a[0], a[1]), (a[2], a[3]), (a[4], a[5]), (a[6], a[7]), (a[8], a[9])


def test_sphere_offsets_match_reference_gen_offset(golden_dir):
    """dropin sphere_conv_offsets (vectorised restatement) == the reference's gen_offset, bit for bit."""
    from mvs_gi_amd.dropin.feature_extractor import sphere_conv_offsets
    z = _load(golden_dir, "sphere_offsets")
    names = [k for k in z.files if not k.endswith("_args")]
    assert "g16vv_final" in names
    for name in names:
        mine = sphere_conv_offsets(*_sphere_args(z[name + "_args"])).numpy()
        assert mine.shape == z[name].shape and np.array_equal(mine, z[name], equal_nan=True), name


def test_deform_conv_oracle_identities():
    """The restated torchvision operator reduces to known answers: zero offsets == F.conv2d (any stride /
    padding / dilation); an integer offset field == the same convolution of the shifted image."""
    import torch.nn.functional as F
    rng = np.random.default_rng(4)
    for (Cin, Cout, k, s, p, d) in ((4, 6, 3, 1, 1, 1), (3, 5, 3, 2, 1, 1), (4, 4, 5, 1, 4, 2), (2, 3, 2, 1, 0, 1)):
        x = torch.from_numpy(rng.standard_normal((2, Cin, 9, 13)).astype(np.float32))
        w = torch.from_numpy(rng.standard_normal((Cout, Cin, k, k)).astype(np.float32))
        b = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32))
        ref = F.conv2d(x, w, b, stride=s, padding=p, dilation=d)
        off = torch.zeros((2, 2 * k * k, *ref.shape[2:]))
        got = O.deform_conv2d(x, off, w, b, (s, s), (p, p), (d, d))
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    x = torch.from_numpy(rng.standard_normal((1, 3, 10, 12)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal((4, 3, 3, 3)).astype(np.float32))
    off = torch.zeros((1, 18, 10, 12))
    off[:, 0::2] = 2.0      # every tap two rows down
    off[:, 1::2] = -1.0     # and one column left
    shifted = torch.zeros_like(x)
    shifted[:, :, :-2, 1:] = x[:, :, 2:, :-1]
    ref = F.conv2d(shifted, w, None, padding=1)
    got = O.deform_conv2d(x, off, w, None, (1, 1), (1, 1), (1, 1))
    # identical away from the borders the shift drags zeros across
    assert float((got - ref)[:, :, 1:-3, 2:-1].abs().max()) <= 1e-5
