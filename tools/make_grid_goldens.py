#!/usr/bin/env python3
"""Golden vectors for the sampling-grid generator, produced by the REFERENCE's own closed forms
(dsta_mvs/support/dataset/torch_cuda_sweep.py) in the build container.

That file imports three helpers of the un-vendored `mvs_utils` submodule; they are replaced by
behaviour-free stand-ins so that the file's own arithmetic can run: the `debug` printers become
no-ops, `torch_meshgrid` is `torch.meshgrid` (it is called with indexing='ij'), and `FTensor`
(a tensor tagged with frame names) returns the plain tensor.  mvs_utils' camera models
(`CameraModelGridMaker`) are NOT covered: parity unpinned for those.

  python tools/make_grid_goldens.py      ->  tests/golden/sweep_grids.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def load_reference():
    def pkg(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m
    for n in ("dsta_mvs_ref", "dsta_mvs_ref.support", "dsta_mvs_ref.support.dataset", "dsta_mvs_ref.mvs_utils"):
        pkg(n)
    dbg = types.ModuleType("dsta_mvs_ref.mvs_utils.debug")
    dbg.show_obj = dbg.show_sum = dbg.show_elements = lambda *a, **k: None
    ft = types.ModuleType("dsta_mvs_ref.mvs_utils.ftensor")

    class FTensor:                      # FTensor(t, f0=...) -> the plain tensor; isinstance(x, FTensor) is False
        def __new__(cls, t, **k):
            return t
    ft.FTensor = FTensor
    mu = sys.modules["dsta_mvs_ref.mvs_utils"]
    mu.torch_meshgrid = lambda *a, indexing="ij": torch.meshgrid(*a, indexing=indexing)
    mu.debug, mu.ftensor = dbg, ft
    sys.modules["dsta_mvs_ref.mvs_utils.debug"], sys.modules["dsta_mvs_ref.mvs_utils.ftensor"] = dbg, ft
    spec = importlib.util.spec_from_file_location(
        "dsta_mvs_ref.support.dataset.torch_cuda_sweep",
        os.path.join(REF, "dsta_mvs", "support", "dataset", "torch_cuda_sweep.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    R = load_reference()
    from mvs_gi_amd.configs import CONFIGS
    from oracle import grid_oracle as G
    out = {}
    cases = {
        "g16": dict(dist=np.asarray(CONFIGS["G16V"].dist_cands, np.float32), shape=(8, 32), lat=(-np.pi / 2, 0.0),
                    lon=(0.0, 2 * np.pi), n_cams=3),
        "e8_full_sphere": dict(dist=np.asarray(CONFIGS["E8"].dist_cands, np.float32), shape=(10, 24), lat=(0.0, np.pi),
                               lon=(-np.pi, np.pi), n_cams=4),
    }
    for name, c in cases.items():
        rm = R.RayMaker_UEPanorama(c["dist"], c["lon"], c["lat"])
        rays = rm.make_rays_for_candidates(c["shape"])
        out[f"{name}_dist"], out[f"{name}_shape"] = c["dist"], np.asarray(c["shape"])
        out[f"{name}_lat"], out[f"{name}_lon"] = np.asarray(c["lat"]), np.asarray(c["lon"])
        out[f"{name}_rays"] = rays.numpy()
        poses = G.ring_poses(c["n_cams"])
        out[f"{name}_poses"] = np.stack([p.numpy() for p in poses])
        ds = R.DoubleSphereSampleGridMaker()
        eq = R.EquirectangularSampleGridMaker()
        for i, pose in enumerate(poses):
            inv = pose.inverse().to(torch.float32)
            pts = R.transform_3D_points_torch(inv.unsqueeze(0), rays.unsqueeze(0))
            out[f"{name}_pts{i}"] = pts.numpy()
            g, m = ds.make_grid(pts)
            out[f"{name}_ds_grid{i}"], out[f"{name}_ds_mask{i}"] = g.numpy(), m.numpy()
            out[f"{name}_eq_grid{i}"] = eq.make_grid(pts).numpy()
    # a second double-sphere parameter set (alpha <= 0.5 branch of w1)
    ds2 = R.DoubleSphereSampleGridMaker(params=[0.1, 0.45, 300.0, 310.0, 320.0, 240.0], calib_shape=[480, 640])
    pts = torch.from_numpy(out["g16_pts1"])
    g, m = ds2.make_grid(pts)
    out["ds2_grid"], out["ds2_mask"], out["ds2_w2"] = g.numpy(), m.numpy(), np.float64(ds2.w2)
    p = os.path.join(ROOT, "tests", "golden", "sweep_grids.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
