#!/usr/bin/env python3
"""Golden offset fields of the sphere convolution, from the REFERENCE's SphereConvEquirect2d.gen_offset
(dsta_mvs/model/common/common_modules.py:427-507; pure torch, imported with the torchvision stub of SURVEY
App. C -- the stubbed deform_conv2d is never called).

  python tools/make_sphere_goldens.py      ->  tests/golden/sphere_offsets.npz
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
tv, ops = types.ModuleType("torchvision"), types.ModuleType("torchvision.ops")


def _absent(*a, **k):
    raise NotImplementedError("torchvision not installed")


ops.deform_conv2d = _absent
tv.ops = ops
sys.modules["torchvision"], sys.modules["torchvision.ops"] = tv, ops
from dsta_mvs.model.common.common_modules import SphereConvEquirect2d  # noqa: E402

CASES = {
    # name: (input_size, kernel, stride, padding, dilation)
    "g16vv_final": ((128, 512), (3, 3), (1, 1), (1, 1), (1, 1)),       # the layer of sphereconv_featext.yaml
    "small": ((16, 64), (3, 3), (1, 1), (1, 1), (1, 1)),
    "k5_s2": ((20, 48), (5, 5), (2, 2), (2, 2), (1, 1)),
    "k3_dil2": ((12, 40), (3, 3), (1, 1), (2, 2), (2, 2)),
    "k2_even": ((8, 32), (2, 2), (1, 1), (0, 0), (1, 1)),
}
out = {}
for name, (size, k, s, p, d) in CASES.items():
    off = SphereConvEquirect2d.gen_offset(size, k, s, p, d)
    out[name] = off.numpy()
    out[name + "_args"] = np.asarray([*size, *k, *s, *p, *d])
    print(name, tuple(off.shape), float(off.abs().max()))
path = os.path.join(ROOT, "tests", "golden", "sphere_offsets.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
