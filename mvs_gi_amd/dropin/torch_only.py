"""Composition of the four stages, mirroring dsta_mvs/model/mvs_model/torch_only.py:4-36.
The feature extractor is whatever nn.Module the caller supplies (PyTorch-ROCm; SURVEY.md
§8(f) rank 1 -- outside the HIP path for now)."""
from torch import nn, Tensor


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
        vol = self.cv_builder(feats, grids, grid_masks, masks)
        costs = self.cv_regulator(vol)
        return self.dist_regressor(costs)

    def forward(self, imgs: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor):
        return self.hot_path(self.extract_features(imgs), grids, grid_masks, masks)
