"""Seeded parity cases shared by tools/make_goldens.py (which runs the REFERENCE on
them, in the build container) and the tests (which run the oracle / the HIP path on
the regenerated, bit-identical inputs).  Inputs are never stored for these cases --
only the reference's outputs plus a sha256 of the regenerated inputs."""
from mvs_gi_amd.configs import CONFIGS, DIST_8L, DIST_10, DIST_16GI, DIST_32I, PathConfig

_small = dict(feat_hw=(16, 64), mask_hw=(64, 256), cv_hw=(8, 32))

# name -> dict(cfg, seed, batch, grid_kind, grid_mask_dtype, gains, stages)
SMALL_CASES = {
    "std_d8": dict(cfg=CONFIGS["G16V"].scaled(dist_cands=DIST_8L, **_small), seed=1, batch=1,
                   grid_kind="smooth", grid_mask_dtype="bool", gains=(1.0,), stages=True),
    "std_d16_rand": dict(cfg=CONFIGS["G16V"].scaled(dist_cands=DIST_16GI, **_small), seed=2, batch=2,
                         grid_kind="random", grid_mask_dtype="f32", gains=(1.0, 4.0, 16.0), stages=True),
    "cat_d8": dict(cfg=CONFIGS["E8-light"].scaled(dist_cands=DIST_8L, **_small), seed=3, batch=1,
                   grid_kind="smooth", grid_mask_dtype="bool", gains=(1.0, 4.0), stages=True),
    "cat4_d8": dict(cfg=CONFIGS["4cam-32"].scaled(dist_cands=DIST_32I[::4], **_small), seed=4, batch=1,
                    grid_kind="random", grid_mask_dtype="bool", gains=(1.0,), stages=False),
    # odd pyramid: D 10/5/3/2, H 12/6/3/2, W 40/20/10/5 -> second trilinear resize
    # to the skip's size (common_modules.py:343-350)
    "std_d10_odd": dict(cfg=CONFIGS["G16V"].scaled(feat_hw=(16, 64), mask_hw=(64, 256), cv_hw=(12, 40),
                                                   dist_cands=DIST_10), seed=5, batch=1,
                        grid_kind="smooth", grid_mask_dtype="bool", gains=(1.0, 4.0), stages=True),
    "std_wide_reg": dict(cfg=PathConfig("G16VV-small", 3, "std", 16, 96, DIST_8L, **_small), seed=6, batch=1,
                         grid_kind="smooth", grid_mask_dtype="bool", gains=(4.0,), stages=False),
}

# Full BASELINE.json sizes: inv_dist only (410 KB each).
FULL_CASES = {
    "full_G16V": dict(cfg=CONFIGS["G16V"], seed=0, batch=1, grid_kind="smooth", grid_mask_dtype="bool",
                      gains=(0.25, 1.0)),
    # worst-case gather locality at full size: sampling grids U[-1.1, 1.1], Bernoulli float grid masks / camera masks
    "full_G16V_rand": dict(cfg=CONFIGS["G16V"], seed=1, batch=1, grid_kind="random", grid_mask_dtype="f32",
                           gains=(1.0,)),
    "full_G16VV": dict(cfg=CONFIGS["G16VV"], seed=0, batch=1, grid_kind="smooth", grid_mask_dtype="bool",
                       gains=(1.0, 4.0)),
    "full_E8": dict(cfg=CONFIGS["E8"], seed=0, batch=1, grid_kind="smooth", grid_mask_dtype="bool",
                    gains=(4.0, 1.0)),
    "full_4cam-32": dict(cfg=CONFIGS["4cam-32"], seed=0, batch=1, grid_kind="smooth", grid_mask_dtype="bool",
                         gains=(0.05, 0.2)),
    # the literal reading of BASELINE.json's configs[2] ("in48ch/fint96ch" at D = 16: sweep_hp_config104.yaml:12-13, SURVEY 8's
    # "E16-class" row): the concat builder on 16 candidates with the (48, 96) regulator
    "full_E16-48-96": dict(cfg=CONFIGS["E16-48-96"], seed=0, batch=1, grid_kind="smooth", grid_mask_dtype="bool",
                           gains=(0.25, 1.0)),
}
