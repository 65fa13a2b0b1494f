"""mvs_gi_amd -- MI355X-native (gfx950) implementation of the distance-candidate plane-sweep
hot path of castacks/mvs_gi: cv_builder (spherical sweep -> cost volume), cv_regulator (3-D
UNet) and dist_regressor (soft-argmin over the candidates), as hand-written HIP kernels
behind a C ABI (include/mvsgi.h) and the reference's own nn.Module interface
(mvs_gi_amd.dropin).  See DESIGN.md and INTEGRATION.md.
"""
from . import configs, synth  # noqa: F401
from .configs import CONFIGS, PathConfig  # noqa: F401

__version__ = "0.1.0"


def install(mode: str = "auto") -> str:
    """Route the reference's module names to the HIP path (see dropin/install.py)."""
    from .dropin.install import install as _install
    return _install(mode)
