"""Composition of the four stages, mirroring dsta_mvs/model/mvs_model/torch_only.py:4-36.
The feature extractor is whatever nn.Module the caller supplies: this package's HIP extractors
(dropin/feature_extractor.py) or the reference's own PyTorch module."""
from torch import nn, Tensor

from .. import hip_ops as H


class SphericalSweepStereoBase(nn.Module):
    def __init__(self, feature_extractor: nn.Module, cv_builder: nn.Module, cv_regulator: nn.Module,
                 dist_regressor: nn.Module):
        super().__init__()
        self.feature_extractor = feature_extractor
        self.cv_builder = cv_builder
        self.cv_regulator = cv_regulator
        self.dist_regressor = dist_regressor

    def extract_features(self, imgs: Tensor) -> Tensor:
        lead = imgs.shape[:2]
        feats = self.feature_extractor(imgs.reshape(lead[0] * lead[1], *imgs.shape[2:]))
        return feats.reshape(*lead, *feats.shape[1:])

    def hot_path(self, feats: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor):
        return self.dist_regressor(build_and_regulate(self.cv_builder, self.cv_regulator, feats, grids, grid_masks, masks))

    def forward(self, imgs: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor):
        return self.hot_path(self.extract_features(imgs), grids, grid_masks, masks)


def build_and_regulate(cv_builder: nn.Module, cv_regulator: nn.Module, feats: Tensor, grids: Tensor, grid_masks: Tensor,
                       masks: Tensor) -> Tensor:
    """costs = cv_regulator(cv_builder(...)) (torch_only.py:32-33).  When both modules are this package's and agree on it
    (builder.forward_split / regulator.takes_split), the cost volume crosses the module boundary as a module-owned split-padded
    buffer instead of an fp32 tensor: the regulator's stride-2 first layer then stages it by LDS-DMA (csrc/conv3d_s2rs.hip).
    Each module's own forward() keeps the reference's tensor interface.
    On entry the fp16 split's range report is applied to the frames submitted BEFORE this one (hip_ops.check_range: the path is
    asynchronous; a caller that synchronises can ask about the current frame with hip_ops.check_range(sync_device=...))."""
    H.check_range("cv_builder -> cv_regulator: an earlier frame")
    fs, takes = getattr(cv_builder, "forward_split", None), getattr(cv_regulator, "takes_split", None)
    if fs is not None and takes is not None and feats.dim() == 5 and grids.dim() == 6 and \
            takes((feats.shape[0], grids.shape[2], grids.shape[3], grids.shape[4], 16)):
        xs = fs(feats, grids, grid_masks, masks)
        if xs is not None:
            return cv_regulator.forward_split_in(xs)
    return cv_regulator(cv_builder(feats, grids, grid_masks, masks))


def reference_forward(self, imgs: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor):
    """SphericalSweepStereoBase.forward (torch_only.py:30-36) for the patched reference class: its own extract_features."""
    feats = self.extract_features(imgs)
    return self.dist_regressor(build_and_regulate(self.cv_builder, self.cv_regulator, feats, grids, grid_masks, masks))
