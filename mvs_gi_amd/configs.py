"""Named workloads of the plane-sweep hot path.

Every entry fixes what the reference fixes through its YAML recipes: number of
cameras, the distance-candidate list, which cost-volume builder is used and
the (in_chs, f_int_chs) of the 3-D UNet regulator.  The candidate lists are the
data values of the reference recipes (experiment_configs/sweep_hp_config29.yaml:3,
sweep_hp_config24.yaml:3, sweep_hp_config7.yaml:3); canonical tensor sizes follow
dsta_mvs/test/utils.py:48-57 (512x2048 images, 128x512 features, 80x320 cv camera).
"""
from dataclasses import dataclass, field
from typing import Tuple

BF = 96.0  # configs/base_model.yaml:31

DIST_16GI = (
    0.5, 0.5436655530390901, 0.5933598538089077, 0.6505899686755604,
    0.7174058117693126, 0.7966693499997578, 0.892501608318949, 1.0110614299826677,
    1.161983802830304, 1.3612365930547528, 1.637348559910646, 2.0467867239341255,
    2.7192485658566654, 4.033339718346582, 7.764432455310304, 100.0,
)
DIST_8L = (
    0.5, 0.5828476269775187, 0.6986027944111776, 0.87173100871731,
    1.1589403973509933, 1.728395061728395, 3.3980582524271843, 100.0,
)
DIST_32I = (
    0.5, 0.77694076, 1.12647449, 1.39806385, 1.66228931, 1.92209016, 2.1913548,
    2.46467309, 2.75718748, 3.05675129, 3.37572113, 3.71045734, 4.0606268,
    4.43892887, 4.85039183, 5.29536264, 5.78156954, 6.31671082, 6.90808345,
    7.57129711, 8.32825577, 9.19776793, 10.2108294, 11.40976115, 12.87013896,
    14.70264105, 17.09268621, 20.3710959, 25.1869872, 33.15291897, 49.08892996,
    100.0,
)
# distance_regressor.py:11 default list (10 candidates -> odd pyramid 10/5/3/2)
DIST_10 = (0.5, 1, 1.5, 2, 5, 10, 20, 30, 50, 100)


@dataclass(frozen=True)
class PathConfig:
    tag: str
    num_cams: int
    builder: str                   # "std" (SphericalSweepStdMasked) | "cat" (SphericalSweep)
    reg_in_chs: int
    reg_f_int_chs: int
    dist_cands: Tuple[float, ...]
    feat_chs: int = 16
    feat_hw: Tuple[int, int] = (128, 512)
    mask_hw: Tuple[int, int] = (512, 2048)
    cv_hw: Tuple[int, int] = (80, 320)
    bf: float = BF
    interp_scale_factor: int = 2
    pre_interp: bool = True

    @property
    def num_cands(self) -> int:
        return len(self.dist_cands)

    @property
    def vol_chs(self) -> int:
        return self.feat_chs if self.builder == "std" else self.feat_chs * self.num_cams

    def scaled(self, feat_hw, mask_hw, cv_hw, dist_cands=None) -> "PathConfig":
        """Same architecture on smaller tensors (parity-test sizes)."""
        d = dict(self.__dict__)
        d.update(feat_hw=tuple(feat_hw), mask_hw=tuple(mask_hw), cv_hw=tuple(cv_hw))
        if dist_cands is not None:
            d["dist_cands"] = tuple(dist_cands)
        return PathConfig(**d)


CONFIGS = {
    # configs[0] of BASELINE.json: config24-style concat recipe, as trained (48, 96)
    "E8": PathConfig("E8", 3, "cat", 48, 96, DIST_8L),
    "E8-light": PathConfig("E8-light", 3, "cat", 48, 32, DIST_8L),
    # configs[1]: G16V = sweep_hp_config29 (README.md:43,78): std builder, default (16, 32)
    "G16V": PathConfig("G16V", 3, "std", 16, 32, DIST_16GI),
    # configs[2]: G16VV = sweep_hp_config103: std builder, (16, 96)
    "G16VV": PathConfig("G16VV", 3, "std", 16, 96, DIST_16GI),
    # what BASELINE.json's "in48ch/fint96ch" wording describes (config104 family)
    "E16-48-96": PathConfig("E16-48-96", 3, "cat", 48, 96, DIST_16GI),
    # configs[4]: 4-camera rig, 32 candidates, concat builder
    "4cam-32": PathConfig("4cam-32", 4, "cat", 64, 32, DIST_32I),
}


def regulator_conv_specs(in_chs: int, f_int_chs: int, final_chs: int = 1, u_depth: int = 3,
                         blk_width: int = 4, stage_factor: int = 2, keep_last_chs=()):
    """Ordered list of the regulator's Conv3d layers as (state-dict prefix, cin, cout,
    has_norm, has_bias), following build_down / build_up / out_costs of
    cost_volume_regulator/unet_regulator.py:52-118."""
    specs = []
    int_chs = []
    cin, cout = in_chs, f_int_chs
    for i in range(u_depth):
        if i not in keep_last_chs and i != 0:
            cout *= stage_factor
        specs.append((f"down_blks.{i}.first", cin, cout, True, False))
        for j in range(blk_width - 1):
            specs.append((f"down_blks.{i}.blks.{j}.blk1", cout, cout, True, False))
            specs.append((f"down_blks.{i}.blks.{j}.blk2", cout, cout, True, False))
        int_chs.append(cout)
        cin = cout
    rev = list(reversed(int_chs))
    for i in range(u_depth - 1):
        specs.append((f"upBlks.{i}.conv", rev[i], rev[i + 1], True, False))
    specs.append(("out_costs.0.conv", int_chs[0], in_chs, True, False))
    specs.append(("out_costs.1", in_chs, final_chs, False, True))
    return specs


def path_gflop(cfg: PathConfig, batch: int = 1) -> float:
    """Algorithmic conv FLOPs (2*27*Cin*Cout*out_voxels) of post_vol + regulator,
    the quantity SURVEY.md §8(d) quotes per frame (G16V: 58.04 GFLOP)."""
    D = cfg.num_cands
    H, W = cfg.cv_hw
    C = cfg.vol_chs

    def half(n):  # stride-2, k3, pad 1
        return (n - 1) // 2 + 1

    total = 2 * 27 * C * C * D * H * W  # post_vol
    dims = [(D, H, W)]
    for _ in range(3):
        d, h, w = dims[-1]
        dims.append((half(d), half(h), half(w)))
    f = cfg.reg_f_int_chs
    chs = [f, 2 * f, 4 * f]
    cin = cfg.reg_in_chs
    for lvl in range(3):
        d, h, w = dims[lvl + 1]
        vox = d * h * w
        total += 2 * 27 * cin * chs[lvl] * vox
        total += 6 * 2 * 27 * chs[lvl] * chs[lvl] * vox
        cin = chs[lvl]
    total += 2 * 27 * chs[2] * chs[1] * dims[2][0] * dims[2][1] * dims[2][2]
    total += 2 * 27 * chs[1] * chs[0] * dims[1][0] * dims[1][1] * dims[1][2]
    total += 2 * 27 * chs[0] * cfg.reg_in_chs * D * H * W
    total += 2 * 27 * cfg.reg_in_chs * 1 * D * H * W
    return batch * total / 1e9
