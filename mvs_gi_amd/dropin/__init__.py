from .common_modules import NoOp, BaseConvBlk3d, ResConvBlk3d, ResizeConv3d, RELU_TYPE, NORM3D_TYPE  # noqa: F401
from .cost_volume_builder import SphericalSweepStdMasked, SphericalSweep  # noqa: F401
from .cost_volume_regulator import UNetCostVolumeRegulatorBase, UNetCostVolumeRegulator, UNetDownBlk  # noqa: F401
from .distance_regressor import DistanceRegressorWithFixedCandidates  # noqa: F401
from .torch_only import SphericalSweepStereoBase  # noqa: F401
from .feature_extractor import (BaseConvBlk2d, ResConvBlk2d, SimpleFeatExtraction, SphereConvEquirect2d,  # noqa: F401
                                SphereConvBlk, SphereEquirectFeatExtraction)
from .install import install, uninstall  # noqa: F401
