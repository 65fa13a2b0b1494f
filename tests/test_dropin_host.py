"""CPU: host logic of the drop-in modules -- constructor/state-dict compatibility with the
reference's names, unpickling of module objects pickled by the reference, install modes,
and the no-CPU-fallback rule."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from mvs_gi_amd import dropin, synth
from mvs_gi_amd.configs import CONFIGS, regulator_conv_specs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tag", ["G16V", "G16VV", "E8", "4cam-32"])
def test_state_dict_names_match_reference(tag):
    cfg = CONFIGS[tag]
    w = synth.make_weights(cfg, seed=0)
    Builder = dropin.SphericalSweepStdMasked if cfg.builder == "std" else dropin.SphericalSweep
    cvb = Builder(num_cams=cfg.num_cams, feat_chs=cfg.vol_chs, post_k_sz=3)
    reg = dropin.UNetCostVolumeRegulatorBase(in_chs=cfg.reg_in_chs, f_int_chs=cfg.reg_f_int_chs)
    # synth weights are keyed with the reference's names (checked against the reference
    # itself by tools/make_goldens.py: strict load_state_dict)
    cvb.load_state_dict({k: torch.from_numpy(v) for k, v in w["cv_builder"].items()}, strict=True)
    reg.load_state_dict({k: torch.from_numpy(v) for k, v in w["cv_regulator"].items()}, strict=True)
    assert len(reg.state_dict()) == 146
    assert len(regulator_conv_specs(cfg.reg_in_chs, cfg.reg_f_int_chs)) == 25


def test_old_regulator_class_is_base_in_2in():
    old = dropin.UNetCostVolumeRegulator(in_chs=16, final_chs=1, u_depth=3, blk_width=4, stage_factor=2, cost_k_sz=3,
                                         keep_last_chs=[], deconv_k_sz=3, sweep_fuse_ch_reduce=2, num_cams=3,
                                         only_one_cam=True)
    base = dropin.UNetCostVolumeRegulatorBase(16, 32)
    so, sb = old.state_dict(), base.state_dict()
    assert list(so) == list(sb)
    assert all(so[k].shape == sb[k].shape for k in so)
    cat = dropin.UNetCostVolumeRegulator(in_chs=32, final_chs=1, u_depth=3, blk_width=4, stage_factor=2, cost_k_sz=3,
                                         keep_last_chs=[], deconv_k_sz=3)   # (32*3)//2 = 48 -> Base(48, 96)
    assert [tuple(v.shape) for v in cat.state_dict().values()] == \
           [tuple(v.shape) for v in dropin.UNetCostVolumeRegulatorBase(48, 96).state_dict().values()]


def test_regressor_attributes_and_update():
    z = np.load(os.path.join(ROOT, "tests", "golden", "regress_variants.npz"))
    dr = dropin.DistanceRegressorWithFixedCandidates(bf=96, dist_cands=list(z["dist_cands"]), interp_scale_factor=2,
                                                     pre_interp=True)
    assert list(dr.state_dict()) == ["inv_dist_idx"] and tuple(dr.inv_dist_idx.shape) == (1, 5, 1, 1)
    dr.update_dist_cands(list(z["updated_cands"]))
    assert np.allclose([dr.inv_dist_idx_min, dr.inv_dist_idx_max], z["updated_minmax"])
    assert dropin.DistanceRegressorWithFixedCandidates(interp_scale_factor=-1).interp_scale_factor == 0


def test_no_cpu_fallback():
    cfg = CONFIGS["G16V"].scaled(feat_hw=(8, 16), mask_hw=(16, 32), cv_hw=(4, 8), dist_cands=(0.5, 1, 2, 4))
    inp = synth.make_inputs(cfg, seed=0)
    cvb = dropin.SphericalSweepStdMasked(3, 16, 3).eval()
    with pytest.raises(RuntimeError, match="GPU only"):
        cvb(*(torch.from_numpy(inp[k]) for k in ("feats", "grids", "grid_masks", "masks")))
    reg = dropin.UNetCostVolumeRegulatorBase(16, 32).eval()
    with pytest.raises(RuntimeError, match="GPU only"):
        reg(torch.zeros(1, 16, 4, 4, 8))
    dr = dropin.DistanceRegressorWithFixedCandidates(dist_cands=[1, 2, 3, 4], interp_scale_factor=2, pre_interp=True)
    with pytest.raises(RuntimeError, match="GPU only"):
        dr(torch.zeros(1, 1, 4, 4, 8))


def _run(code: str, extra_path=()):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, *extra_path])
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env,
                       cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_unpickle_reference_modules_into_dropin_alias_mode():
    """tests/golden/pickled_modules_tiny.pt holds module OBJECTS pickled by the reference
    (the way Lightning's save_hyperparameters stores them); with the alias install they
    unpickle into the drop-in classes without the reference on the path."""
    out = _run("""
        import torch, mvs_gi_amd
        assert mvs_gi_amd.install() == "alias"
        hp = torch.load("tests/golden/pickled_modules_tiny.pt", weights_only=False)["hyper_parameters"]
        from mvs_gi_amd import dropin
        assert type(hp["cv_builder"]) is dropin.SphericalSweepStdMasked, type(hp["cv_builder"])
        assert type(hp["cv_regulator"]) is dropin.UNetCostVolumeRegulator
        assert type(hp["dist_regressor"]) is dropin.DistanceRegressorWithFixedCandidates
        assert type(hp["cv_regulator"].down_blks[0].blks[0].blk1) is dropin.BaseConvBlk3d
        assert len(hp["cv_regulator"].state_dict()) == 146
        assert hp["cv_builder"].post_vol.conv_layer.weight.shape == (4, 4, 3, 3, 3)
        hp["dist_regressor"].update_dist_cands([1, 2, 3, 4, 5, 6, 7, 8])
        print("OK")
    """)
    assert "OK" in out


@pytest.mark.skipif(not os.path.isdir("/root/reference/dsta_mvs"), reason="reference checkout not present")
def test_patch_mode_rebinds_reference_classes():
    out = _run("""
        import sys, types
        tv, ops = types.ModuleType("torchvision"), types.ModuleType("torchvision.ops")
        ops.deform_conv2d = lambda *a, **k: None; tv.ops = ops
        sys.modules["torchvision"], sys.modules["torchvision.ops"] = tv, ops
        import torch, mvs_gi_amd
        assert mvs_gi_amd.install() == "patch"
        from dsta_mvs.model.cost_volume_regulator.unet_regulator import UNetCostVolumeRegulatorBase
        from dsta_mvs.model.cost_volume_builder import SphericalSweepStdMasked
        assert UNetCostVolumeRegulatorBase.__module__.startswith("dsta_mvs.")
        reg = UNetCostVolumeRegulatorBase(16, 32).eval()
        try:
            reg(torch.zeros(1, 16, 4, 4, 8))
        except RuntimeError as e:
            assert "GPU only" in str(e)       # the reference class now routes to the HIP path
        else:
            raise SystemExit("reference forward still runs on CPU")
        from mvs_gi_amd.dropin.install import uninstall
        uninstall()
        y = reg(torch.zeros(1, 16, 8, 8, 8))  # original forward restored
        assert tuple(y.shape) == (1, 1, 8, 8, 8)
        print("OK")
    """, extra_path=["/root/reference"])
    assert "OK" in out


def test_feature_extractor_state_dict_names():
    fe = dropin.SimpleFeatExtraction(in_size=(512, 2048), in_chs=3, chs=16, k_sz=3, layers=[5, 10])
    sd = synth.make_extractor_weights(0)    # keyed with the reference's names (strict-loaded into the reference by make_goldens)
    fe.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert len(fe.state_dict()) == 198
    assert fe.first.infer_size((512, 2048)) == (256, 1024)
    with pytest.raises(RuntimeError, match="GPU only"):
        fe.eval()(torch.zeros(1, 3, 16, 32))


def test_default_conv_mode_is_the_fp16_split():
    """a maintainer who installs the drop-in without touching any knob gets the
    split-fp16 path (the one bench.py reports); MVSGI_CONV_MODE=bf16x3 / f32 select the bf16 split / the exact-fp32 kernels."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "MVSGI_CONV_MODE"}
    code = "from mvs_gi_amd import hip_ops as H; print(H.get_conv_mode())"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "f16x3", r.stdout + r.stderr
    for mode in ("f32", "bf16x3"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, MVSGI_CONV_MODE=mode),
                           cwd=root, timeout=300)
        assert r.stdout.strip() == mode
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, MVSGI_CONV_MODE="fp16"),
                       cwd=root, timeout=300)
    assert r.returncode != 0


def test_pickled_modules_do_not_carry_derived_caches():
    """Launch records, packed weights, polyphase plans and module-owned activation buffers live under `_mvsgi_*` keys of a
    module's __dict__; a pickled module (Lightning pickles module OBJECTS: spherical_sweep_stereo.py:74) must not carry them."""
    import io
    import pickle
    from mvs_gi_amd import dropin
    reg = dropin.UNetCostVolumeRegulatorBase(in_chs=16, f_int_chs=32)
    cvb = dropin.SphericalSweepStdMasked(num_cams=3, feat_chs=16, post_k_sz=3)
    reg.__dict__["_mvsgi_poly_bufs"] = {"k": torch.zeros(4)}
    reg.down_blks[0].__dict__["_mvsgi_rs_bufs"] = {"k": torch.zeros(4)}
    reg.out_costs[1].__dict__["_mvsgi_launch"] = object()            # not even picklable on its own terms
    cvb.__dict__["_mvsgi_rs_vol"] = {"k": torch.zeros(4)}
    for m in (reg, cvb):
        m2 = pickle.loads(pickle.dumps(m))
        assert not [k for mod in m2.modules() for k in mod.__dict__ if k.startswith("_mvsgi_")]
        sd, sd2 = m.state_dict(), m2.state_dict()
        assert list(sd) == list(sd2) and all(torch.equal(sd[k], sd2[k]) for k in sd)
    buf = io.BytesIO()
    torch.save(reg, buf)
    assert buf.tell() < 20 * 2 ** 20                                 # the 4 M-parameter regulator, nothing else


def test_build_and_regulate_hand_over_decision():
    """dropin/torch_only.py:build_and_regulate: the split-padded hand-over is used only when BOTH modules offer it, the regulator
    takes the geometry and the builder produces the buffer; every other combination goes through the tensor interface."""
    import torch
    from mvs_gi_amd.dropin.torch_only import build_and_regulate
    feats = torch.zeros(2, 3, 16, 4, 8)
    grids = torch.zeros(2, 3, 5, 6, 7, 2)
    calls = []

    class Builder(torch.nn.Module):
        def __init__(self, split, gives):
            super().__init__()
            if split:
                self.forward_split = lambda *a: (calls.append("forward_split"), "XS" if gives else None)[1]

        def forward(self, f, g, gm, m):
            calls.append("builder")
            return "VOL"

    class Regulator(torch.nn.Module):
        def __init__(self, split, takes):
            super().__init__()
            if split:
                self.takes_split = lambda shape: (calls.append(("takes", tuple(shape))), takes)[1]
                self.forward_split_in = lambda xs: (calls.append(("split_in", xs)), "COSTS_S")[1]

        def forward(self, vol):
            calls.append(("regulator", vol))
            return "COSTS_T"

    def run(bs, bg, rs, rt):
        calls.clear()
        return build_and_regulate(Builder(bs, bg), Regulator(rs, rt), feats, grids, None, None), list(calls)

    out, c = run(True, True, True, True)
    assert out == "COSTS_S" and c == [("takes", (2, 5, 6, 7, 16)), "forward_split", ("split_in", "XS")]
    out, c = run(True, False, True, True)                    # the builder configuration does not produce the buffer
    assert out == "COSTS_T" and c[-2:] == ["builder", ("regulator", "VOL")]
    out, c = run(True, True, True, False)                    # the regulator does not take this geometry
    assert out == "COSTS_T" and "forward_split" not in c
    for bs, rs in ((False, True), (True, False), (False, False)):     # a foreign module on either side
        out, c = run(bs, True, rs, True)
        assert out == "COSTS_T" and c == ["builder", ("regulator", "VOL")]


def test_power_of_two_prescale_of_the_fp16_split():
    """hip_ops._pow2_unscale: the per-channel power of two that puts a channel's largest weight into (512, 1024] (its fp16 lo part is
    then a normal number), exactly invertible; a zero channel is left alone; 'f16x3' is a mode, 'bf16x3' the default."""
    import torch
    from mvs_gi_amd import hip_ops as H
    amax = torch.tensor([0.0, 1e-9, 3.7e-3, 0.11, 0.5, 1.0, 511.9, 512.0, 1023.9, 1024.0, 5e4, 3e30])
    up, un = H._pow2_unscale(amax)
    assert float(up[0]) == 1.0 and float(un[0]) == 1.0
    assert torch.equal(up * un, torch.ones_like(up))
    assert torch.equal(torch.log2(up), torch.round(torch.log2(up)))              # powers of two
    m = amax[1:-1] * up[1:-1]
    assert bool(((m > 512.0) & (m <= 1024.0)).all())
    assert float(up[-1]) == 2.0 ** -90 or float(amax[-1] * up[-1]) < 1024.0       # clamped exponent: finite, never 0 / inf
    assert "f16x3" in H.CONV_MODES and H.mode_fmt() in ("bf16", "f16")
    old = H.get_conv_mode()
    try:
        H.set_conv_mode("f16x3")
        assert H.split_mode() and H.mode_fmt() == "f16"
        H.set_conv_mode("f32")
        assert not H.split_mode() and H.mode_fmt() == "bf16"
    finally:
        H.set_conv_mode(old)


def test_winograd_dispatch_rule_follows_the_rounds_of_units():
    """dropin/cost_volume_regulator.py::_wino_pays: a Winograd unit (2 rows x 32 columns, all planes) holds a CU for the whole
    launch, so the form is dispatched when the launch's units fill at least 70 % of their rounds of one unit per CU -- on a
    256-CU part and a [8, 40, 160] level 0 (100 units per frame): not at 1 or 3 frames, from 2 frames on otherwise."""
    from mvs_gi_amd.dropin import cost_volume_regulator as cr
    old = dict(cr._CUS)
    try:
        cr._CUS["fake"] = 256
        got = {b: cr._wino_pays(b, 40, 160, "fake") for b in (1, 2, 3, 4, 5, 6, 8, 16, 64)}
        assert got == {1: False, 2: True, 3: False, 4: True, 5: True, 6: True, 8: True, 16: True, 64: True}
        cr._CUS["small"] = 64                       # a smaller part: one frame already fills it
        assert cr._wino_pays(1, 40, 160, "small")
    finally:
        cr._CUS.clear()
        cr._CUS.update(old)

