"""Make the HIP path the implementation behind the reference's module names.

Two situations:

* the reference package is NOT importable (stand-alone use, the GPU test box): `install()`
  registers alias modules under the reference's import paths
  (dsta_mvs.model.cost_volume_builder[.spherical_sweep_avg|.spherical_sweep],
   dsta_mvs.model.cost_volume_regulator[.unet_regulator],
   dsta_mvs.model.distance_regressor[.distance_regressor],
   dsta_mvs.model.common[.common_modules], dsta_mvs.model.mvs_model.torch_only),
  so that checkpoints which pickle whole module objects
  (spherical_sweep_stereo.py:74; consumers api/inference_pytorch.py:87-101,
  dsta_mvs/test/utils.py:221-230) unpickle into the drop-in classes;

* the reference IS importable (a user's mvs_gi checkout): `install()` re-binds `forward`
  (and `sweep`) of the reference's own classes to the HIP implementations, so
  api/inference_pytorch.py:115-122 runs them without any other change.
"""
from __future__ import annotations

import importlib
import sys
import types

from . import common_modules as _cm
from . import cost_volume_builder as _cvb
from . import cost_volume_regulator as _reg
from . import distance_regressor as _dr
from . import torch_only as _to
from . import feature_extractor as _fe

_ALIAS_FLAG = "__mvsgi_alias__"
_state = {"mode": None, "saved": []}


def _pkg(name: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = []          # mark as package
    setattr(m, _ALIAS_FLAG, True)
    return m


def _export(mod: types.ModuleType, src, names=None):
    for n in (names or [k for k in vars(src) if not k.startswith("_")]):
        setattr(mod, n, getattr(src, n))


def _install_aliases():
    mods = {}

    def leaf(name, src):
        m = types.ModuleType(name)
        setattr(m, _ALIAS_FLAG, True)
        _export(m, src)
        mods[name] = m
        return m

    root = _pkg("dsta_mvs")
    model = _pkg("dsta_mvs.model")
    common = _pkg("dsta_mvs.model.common")
    _export(common, _cm, ["NoOp", "RELU_TYPE", "NORM2D_TYPE", "NORM3D_TYPE"])
    cvb = _pkg("dsta_mvs.model.cost_volume_builder")
    _export(cvb, _cvb, ["SphericalSweepStdMasked", "SphericalSweep"])
    reg = _pkg("dsta_mvs.model.cost_volume_regulator")
    _export(reg, _reg, ["UNetCostVolumeRegulatorBase", "UNetCostVolumeRegulator", "UNetDownBlk"])
    dr = _pkg("dsta_mvs.model.distance_regressor")
    _export(dr, _dr, ["DistanceRegressorWithFixedCandidates"])
    fe = _pkg("dsta_mvs.model.feature_extractor")
    _export(fe, _fe, ["SimpleFeatExtraction", "SphereEquirectFeatExtraction"])
    mm = _pkg("dsta_mvs.model.mvs_model")
    _export(mm, _to, ["SphericalSweepStereoBase"])
    mods.update({"dsta_mvs": root, "dsta_mvs.model": model, "dsta_mvs.model.common": common,
                 "dsta_mvs.model.cost_volume_builder": cvb, "dsta_mvs.model.cost_volume_regulator": reg,
                 "dsta_mvs.model.distance_regressor": dr, "dsta_mvs.model.mvs_model": mm,
                 "dsta_mvs.model.feature_extractor": fe})
    fe.simple_feature_extractor = leaf("dsta_mvs.model.feature_extractor.simple_feature_extractor", _fe)
    fe.sphere_feature_extractor = leaf("dsta_mvs.model.feature_extractor.sphere_feature_extractor", _fe)

    common.common_modules = leaf("dsta_mvs.model.common.common_modules", _cm)
    for n in ("BaseConvBlk2d", "ResConvBlk2d", "SphereConvEquirect2d", "SphereConvBlk"):   # 2-D blocks live in common_modules upstream
        setattr(common.common_modules, n, getattr(_fe, n))
    cvb.spherical_sweep_avg = leaf("dsta_mvs.model.cost_volume_builder.spherical_sweep_avg", _cvb)
    cvb.spherical_sweep = leaf("dsta_mvs.model.cost_volume_builder.spherical_sweep", _cvb)
    reg.unet_regulator = leaf("dsta_mvs.model.cost_volume_regulator.unet_regulator", _reg)
    dr.distance_regressor = leaf("dsta_mvs.model.distance_regressor.distance_regressor", _dr)
    mm.torch_only = leaf("dsta_mvs.model.mvs_model.torch_only", _to)
    root.model = model
    model.common, model.cost_volume_builder, model.cost_volume_regulator = common, cvb, reg
    model.distance_regressor, model.mvs_model, model.feature_extractor = dr, mm, fe
    for k, v in mods.items():
        sys.modules[k] = v
    _state["saved"] = list(mods)


def _patch_reference():
    saved = []

    def rebind(cls, **attrs):
        for k, fn in attrs.items():
            saved.append((cls, k, cls.__dict__.get(k)))
            setattr(cls, k, fn)

    b = importlib.import_module("dsta_mvs.model.cost_volume_builder")
    r = importlib.import_module("dsta_mvs.model.cost_volume_regulator.unet_regulator")
    d = importlib.import_module("dsta_mvs.model.distance_regressor.distance_regressor")
    c = importlib.import_module("dsta_mvs.model.common.common_modules")
    gs = _cm.module_getstate      # derived `_mvsgi_*` caches never travel with a pickled module
    rebind(b.SphericalSweepStdMasked, forward=_cvb.std_forward, forward_split=_cvb.std_forward_split,
           sweep=_cvb.SphericalSweepStdMasked.sweep, __getstate__=gs)
    rebind(b.SphericalSweep, forward=_cvb.cat_forward, sweep=_cvb.SphericalSweep.sweep, __getstate__=gs)
    rebind(r.UNetCostVolumeRegulatorBase, forward=_reg.regulator_forward, takes_split=_reg.regulator_takes_split,
           forward_split_in=_reg.regulator_forward_split_in, __getstate__=gs)
    rebind(r.UNetCostVolumeRegulator, forward=_reg.regulator_forward, takes_split=_reg.regulator_takes_split,
           forward_split_in=_reg.regulator_forward_split_in, __getstate__=gs)
    rebind(r.UNetDownBlk, forward=_reg.UNetDownBlk.forward, __getstate__=gs)
    rebind(d.DistanceRegressorWithFixedCandidates, forward=_dr.regressor_forward)
    rebind(c.BaseConvBlk3d, forward=_cm.BaseConvBlk3d.forward, __getstate__=gs)
    rebind(c.ResConvBlk3d, forward=_cm.ResConvBlk3d.forward)
    rebind(c.ResizeConv3d, forward=_cm.ResizeConv3d.forward)
    rebind(c.BaseConvBlk2d, forward=_fe.BaseConvBlk2d.forward, __getstate__=gs)
    rebind(c.ResConvBlk2d, forward=_fe.ResConvBlk2d.forward, __getstate__=gs)
    f = importlib.import_module("dsta_mvs.model.feature_extractor.simple_feature_extractor")
    rebind(f.SimpleFeatExtraction, forward=_fe.extractor_forward, __getstate__=gs)
    rebind(c.SphereConvEquirect2d, forward=_fe.SphereConvEquirect2d.forward)
    rebind(c.SphereConvBlk, forward=_fe.SphereConvBlk.forward)
    fs = importlib.import_module("dsta_mvs.model.feature_extractor.sphere_feature_extractor")
    rebind(fs.SphereEquirectFeatExtraction, forward=_fe.sphere_extractor_forward, __getstate__=gs)
    # the composition: the reference's own extract_features, then builder -> regulator with this package's hand-over
    # (the mvs_model package pulls in the training stack -- cv2, lightning: where that does not import, the modules above are
    # still patched and the volume crosses the boundary as the reference's fp32 tensor)
    try:
        mm = importlib.import_module("dsta_mvs.model.mvs_model.torch_only")
    except Exception:
        mm = None
    if mm is not None:
        rebind(mm.SphericalSweepStereoBase, forward=_to.reference_forward)
    _state["saved"] = saved


def reference_importable() -> bool:
    m = sys.modules.get("dsta_mvs")
    if m is not None:
        return not getattr(m, _ALIAS_FLAG, False)
    try:
        importlib.import_module("dsta_mvs.model.cost_volume_builder")
        return True
    except Exception:
        for k in [k for k in sys.modules if k == "dsta_mvs" or k.startswith("dsta_mvs.")]:
            del sys.modules[k]
        return False


def install(mode: str = "auto") -> str:
    """mode: 'auto' | 'alias' | 'patch'.  Returns the mode that was applied."""
    if _state["mode"] is not None:
        return _state["mode"]
    if mode == "auto":
        mode = "patch" if reference_importable() else "alias"
    if mode == "alias":
        _install_aliases()
    elif mode == "patch":
        _patch_reference()
    else:
        raise ValueError(mode)
    _state["mode"] = mode
    return mode


def uninstall() -> None:
    if _state["mode"] == "alias":
        for k in _state["saved"]:
            sys.modules.pop(k, None)
    elif _state["mode"] == "patch":
        for cls, k, old in reversed(_state["saved"]):
            if old is None:
                delattr(cls, k)
            else:
                setattr(cls, k, old)
    _state["mode"], _state["saved"] = None, []
