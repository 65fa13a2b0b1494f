"""Build-owned counterpart of the reference's inference facade for this path
(api/inference_pytorch.py:49-122 + api/inference_class.py:111-127): constant
grids/grid_masks/masks moved to the device once, the four-stage call under no_grad, and
inv_dist / bf post-processing.  Used by bench.py, the tests and __graft_entry__.smoke().
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .configs import PathConfig
from . import dropin
from . import hip_ops as H
from .dropin.torch_only import build_and_regulate


def build_modules(cfg: PathConfig, weights: Optional[Dict[str, Dict[str, np.ndarray]]] = None, device="cuda"):
    """Instantiate the three drop-in modules for a config and (optionally) load state dicts
    keyed with the reference's names."""
    Builder = dropin.SphericalSweepStdMasked if cfg.builder == "std" else dropin.SphericalSweep
    cvb = Builder(num_cams=cfg.num_cams, feat_chs=cfg.vol_chs, post_k_sz=3)
    reg = dropin.UNetCostVolumeRegulatorBase(in_chs=cfg.reg_in_chs, f_int_chs=cfg.reg_f_int_chs)
    dr = dropin.DistanceRegressorWithFixedCandidates(bf=cfg.bf, dist_cands=list(cfg.dist_cands),
                                                     interp_scale_factor=cfg.interp_scale_factor,
                                                     pre_interp=cfg.pre_interp)
    if weights is not None:
        cvb.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights["cv_builder"].items()},
                            strict=True)
        reg.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights["cv_regulator"].items()},
                            strict=True)
    mods = [m.eval().to(device) for m in (cvb, reg, dr)]
    return tuple(mods)


class HotPath:
    """feats -> inv_dist with the rig constants resident on the device."""

    def __init__(self, cfg: PathConfig, weights, consts: Dict[str, np.ndarray], device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.cv_builder, self.cv_regulator, self.dist_regressor = build_modules(cfg, weights, device)
        self.grids = torch.from_numpy(consts["grids"]).to(self.device)
        self.grid_masks = torch.from_numpy(consts["grid_masks"]).to(self.device)
        self.masks = torch.from_numpy(consts["masks"]).to(self.device)

    def check_range(self, sync: bool = True) -> int:
        """The range report of the fp16 split for the frames submitted so far: synchronises the device (sync=True), then raises
        hip_ops.MvsgiRangeError / warns per MVSGI_RANGE_CHECK when a clamp of the default arithmetic engaged (the reference is fp32
        and has none, common_modules.py:105-115).  __call__ / replay run the same check WITHOUT synchronising on entry: they report
        the frames before the current one."""
        return H.check_range("HotPath", self.device if sync else None)

    @torch.no_grad()
    def __call__(self, feats: torch.Tensor):
        H.check_range("HotPath: an earlier frame")
        B = feats.shape[0]
        g, gm, m = self.grids, self.grid_masks, self.masks
        if g.shape[0] != B:          # rig constants are per-frame identical
            shared = self.cfg.builder == "std" and bool(getattr(self.cv_builder, "cache_rig_constants", True))
            if getattr(self, "_rig_views", None) is None or self._rig_views[0] != (B, shared):
                exp = [t[:1].expand(B, *t.shape[1:]) for t in (g, gm, m)]
                # std builder with the rig cache: ONE set presented with the batch's shape (stride-0 views; the sweep reads
                # frame 0's constants for every frame).  Otherwise replicated once.
                self._rig_views = ((B, shared), *(exp if shared else [t.contiguous() for t in exp]))
            _, g, gm, m = self._rig_views          # the same objects every call: the rig cache hits by identity
        return self.dist_regressor(build_and_regulate(self.cv_builder, self.cv_regulator, feats, g, gm, m))

    # ---- hipGraph replay: the ~35 launches of one forward captured once, replayed per batch ----
    def capture(self, feats: torch.Tensor) -> None:
        """Capture the forward for this input shape into a hipGraph (kernel launches only: the
        path does no host synchronisation or allocation outside torch's capture-aware pool).
        Afterwards `replay(feats)` copies the input into the graph's static buffer and replays."""
        self._static_in = feats.clone()
        self(self._static_in)                       # warm-up: one-time kernel attribute set-up, weight packing
        torch.cuda.synchronize(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self(self._static_in)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        # thread-local error mode: another thread's CUDA calls (a process group's watchdog polling its events) must not
        # invalidate this capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._static_out = self(self._static_in)
        self._graph = graph

    def replay(self, feats: Optional[torch.Tensor] = None):
        """Replay the captured forward; returns the graph's static output tensors
        (inv_dist, norm_costs), valid until the next replay."""
        if getattr(self, "_graph", None) is None:
            raise RuntimeError("HotPath.replay() before capture()")
        H.check_range("HotPath: an earlier frame")
        if feats is not None and feats.data_ptr() != self._static_in.data_ptr():
            if tuple(feats.shape) != tuple(self._static_in.shape) or feats.dtype != self._static_in.dtype:
                raise ValueError(f"HotPath.replay(): the graph was captured for feats {tuple(self._static_in.shape)} "
                                 f"{self._static_in.dtype}, got {tuple(feats.shape)} {feats.dtype}; capture() again")
            self._static_in.copy_(feats, non_blocking=True)
        self._graph.replay()
        return self._static_out

    def postprocess(self, inv_dist: torch.Tensor) -> np.ndarray:
        """inverse-distance index -> 1/m, on the host (api/inference_class.py:111-114)."""
        return (inv_dist / self.cfg.bf).cpu().numpy()

    @torch.no_grad()
    def precision_check(self, feats: torch.Tensor, bar: float = 5e-4) -> Dict[str, object]:
        """Which conv arithmetic does THIS checkpoint need?  Runs `feats` (one or a few real frames) through the path in the
        fp16 split (the library's default) and in the bf16 split and returns their inverse-distance discrepancy, max |d| / max |inv_dist|.
        On in-range activations the fp16 split's own error is 7-11x smaller (Precision modes), so the discrepancy IS the
        bf16 split's error to ~15 % -- small: the two agree and the default stands.  When they disagree by more than `bar` (default:
        half the north star's 1e-3) the exact-fp32 mode arbitrates: the split closer to it wins if it is inside the bar -- the fp16
        split on a sharp softmax (the error of either split grows with the sharpness of the checkpoint's softmax over the candidates,
        which only the trained weights know), the bf16 split when activations leave fp16's range (+-65504; +-16376 at the
        Winograd-form level-0 convs) or live far below 1e-3 -- else "f32".  The measurement a deployer makes once per checkpoint.
        The fp16 split's range report (hip_ops.saturation_flags) is read after its run: a clamp that engaged sends the question to
        the exact mode too and excludes the fp16 split; a non-finite output of either split does the same for that split.
        -> {"bf16x3_vs_f16x3": e, "bar": bar, "recommended": mode, "f16x3_saturated": flags, "finite": {...}
            [, "bf16x3_vs_f32": e, "f16x3_vs_f32": e, "note": ...]}"""
        old, old_policy = H.get_conv_mode(), H.get_range_check()
        outs, finite = {}, {}
        try:
            H.set_range_check("off")             # the flags are read by hand below
            torch.cuda.synchronize(self.device)
            H.saturation_flags(clear=True)
            for mode in ("bf16x3", "f16x3"):
                H.set_conv_mode(mode)
                outs[mode] = self(feats)[0].clone()
                finite[mode] = bool(torch.isfinite(outs[mode]).all())
            torch.cuda.synchronize(self.device)
            sat = H.saturation_flags(clear=True)          # raised by the f16x3 run only: the bf16 split has no range to leave
            den = float(outs["f16x3"].abs().max()) if finite["f16x3"] else float("nan")
            if not np.isfinite(den) or den == 0.0:
                den = float(outs["bf16x3"].abs().max()) if finite["bf16x3"] else 1.0
                den = den if np.isfinite(den) and den > 0.0 else 1.0
            e_b = float((outs["bf16x3"] - outs["f16x3"]).abs().max()) / den if finite["bf16x3"] and finite["f16x3"] else float("inf")
            res = {"bf16x3_vs_f16x3": e_b, "bar": bar, "recommended": "f16x3", "f16x3_saturated": sat,
                   "finite": dict(finite)}
            # the exact mode arbitrates when the splits disagree, when either is not finite (an fp32 overflow poisons the bf16
            # split too) and when the fp16 split's clamps engaged
            if not (e_b <= bar) or sat:
                H.set_conv_mode("f32")
                exact = self(feats)[0]
                ok_exact = bool(torch.isfinite(exact).all())
                den = float(exact.abs().max()) if ok_exact else den
                den = den if np.isfinite(den) and den > 0.0 else 1.0
                e = {m: (float((outs[m] - exact).abs().max()) / den if finite[m] and ok_exact else float("inf")) for m in ("bf16x3", "f16x3")}
                if sat:
                    e["f16x3"] = float("inf")          # a clamped result is excluded whatever its distance
                res["bf16x3_vs_f32"], res["f16x3_vs_f32"] = e["bf16x3"], e["f16x3"]
                best = min(e, key=e.get)
                res["recommended"] = best if e[best] <= bar else "f32"
                if sat:
                    res["note"] = ("the fp16 split saturated on these frames (flags %d): its result is clamped" % sat)
                elif e[best] > bar:
                    # two fp32-accumulating arithmetics of 22+ bits that differ by more than the bar: past this sharpness the
                    # reference's own summation order is one answer among several (~0.993 mean max-probability on G16V)
                    res["note"] = "the bar is ill-conditioned for this checkpoint: fp32 summation orders differ by more"
            return res
        finally:
            H.set_conv_mode(old)
            H.set_range_check(old_policy)


def overlapping_copy_stream(device, compute_stream: Optional[torch.cuda.Stream] = None, tries: int = 8) -> torch.cuda.Stream:
    """A HIP stream whose copies run BESIDE the kernels of `compute_stream` (default: the current stream) -- for a caller that
    feeds frames from pinned host memory while the previous batch is computed (double buffering, as the reference's robot would feed
    `api/inference_class.py:120-127` at more than one frame in flight).  HIP multiplexes its streams onto a few hardware queues (four by
    default, assigned round-robin at creation), and a stream that shares the compute stream's queue is served in submission order: its
    uploads wait for the kernels submitted before them (MI355X, 64 G16V frames from uint8 images: 40.3 ms per batch = 0.66 of the
    resident rate on such a stream, 27.4 ms = 0.975 on any other; `tools/host_feed_probe.py --queues`).  Candidates are tested with a
    spin kernel on the compute stream and a small pinned copy on the candidate; the first whose copy finishes inside the spin wins."""
    device = torch.device(device)
    comp = compute_stream if compute_stream is not None else torch.cuda.current_stream(device)
    host = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
    dst = torch.empty(1 << 20, dtype=torch.uint8, device=device)
    keep, last = [], None
    for _ in range(max(1, tries)):
        cand = torch.cuda.Stream(device=device)
        keep.append(cand)                     # (held until the choice is made: a released stream object may be handed out again)
        last = cand
        torch.cuda.synchronize(device)
        c0, c1, k1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(comp):
            c0.record(comp)
            torch.cuda._sleep(4_000_000)      # ~2 ms of spinning at ~2 GHz
            c1.record(comp)
        with torch.cuda.stream(cand):
            cand.wait_event(c0)
            dst.copy_(host, non_blocking=True)
            k1.record(cand)
        torch.cuda.synchronize(device)
        if c0.elapsed_time(k1) < 0.5 * c0.elapsed_time(c1):
            return cand
    return last


class StreamedHotPath:
    """The batch of one step cut into `n_streams` independent parts, each through its own HotPath replica (own module-owned
    activation buffers; parameters and rig constants replicated: < 150 MB) on its own HIP stream, forked from and joined to the
    caller's stream.  Frames are independent, so this is the single-GPU form of the frame sharding: the tail of one part's kernel
    -- the last, partly filled round of a persistent grid, the drain of a small layer -- is filled by the other part's launches
    instead of idling the chip (MI355X, G16V: two parts of 64 frames 5670 frames/s against 5570 for one part of 128 and 5490
    for one of 64).  capture() records the fork / join into ONE hipGraph; replay() is one submission per step."""

    def __init__(self, cfg: PathConfig, weights, consts: Dict[str, np.ndarray], device="cuda", n_streams: int = 2):
        if n_streams < 1:
            raise ValueError("n_streams must be >= 1")
        self.cfg, self.device = cfg, torch.device(device)
        self.parts = [HotPath(cfg, weights, consts, device) for _ in range(n_streams)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n_streams)]

    def _split(self, feats: torch.Tensor):
        n = len(self.parts)
        B = feats.shape[0]
        if B % n:
            raise ValueError(f"a batch of {B} frames does not split into {n} equal parts")
        return feats.split(B // n, dim=0)

    @torch.no_grad()
    def __call__(self, feats: torch.Tensor):
        """-> list of (inv_dist, norm_costs), one per part, in frame order."""
        cur = torch.cuda.current_stream(self.device)
        outs = []
        for hp, st, f in zip(self.parts, self.streams, self._split(feats)):
            st.wait_stream(cur)                    # fork: the part starts when the caller's stream has produced feats
            with torch.cuda.stream(st):
                outs.append(hp(f))
        for st in self.streams:
            cur.wait_stream(st)                    # join
        return outs

    def capture(self, feats: torch.Tensor) -> None:
        self._static_in = feats.clone()
        self(self._static_in)                       # warm-up: weight packing, rig constants, buffer allocation
        torch.cuda.synchronize(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self(self._static_in)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._static_out = self(self._static_in)
        self._graph = graph

    def replay(self, feats: Optional[torch.Tensor] = None):
        if getattr(self, "_graph", None) is None:
            raise RuntimeError("StreamedHotPath.replay() before capture()")
        H.check_range("StreamedHotPath: an earlier step")
        if feats is not None and feats.data_ptr() != self._static_in.data_ptr():
            if tuple(feats.shape) != tuple(self._static_in.shape) or feats.dtype != self._static_in.dtype:
                raise ValueError(f"StreamedHotPath.replay(): captured for feats {tuple(self._static_in.shape)}, got {tuple(feats.shape)}")
            self._static_in.copy_(feats, non_blocking=True)
        self._graph.replay()
        return self._static_out


class InferencePipeline:
    """Build-owned counterpart of InferencePytorch.__call__ (api/inference_class.py:120-127): uint8
    HWC camera images in, metric inverse distance (inv_dist / bf) as a host array out.  The uint8 ->
    float / 255 conversion is folded into the extractor's stem kernel and the division by bf into the
    soft-argmin kernel, so the only host traffic is the image upload and the result download."""

    def __init__(self, cfg: PathConfig, weights, consts, device="cuda", extractor: str = "simple"):
        """extractor: 'simple' = SimpleFeatExtraction (G16V, config29), 'sphere' = SphereEquirectFeatExtraction with the
        sphere-convolution final layer (G16VV, config103; configs/feature_extractor/sphereconv_featext.yaml)."""
        self.cfg = cfg
        self.hot = HotPath(cfg, weights, consts, device)
        Hi, Wi = cfg.feat_hw
        Extractor = {"simple": dropin.SimpleFeatExtraction, "sphere": dropin.SphereEquirectFeatExtraction}[extractor]
        fe = Extractor(in_size=(4 * Hi, 4 * Wi), in_chs=3, chs=cfg.feat_chs, k_sz=3, layers=[5, 10])
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in weights["feature_extractor"].items()}
        if extractor == "sphere":          # the offset field is a constructor-made buffer unless the checkpoint brings one
            sd.setdefault("final_layer.blk.0.offset", fe.final_layer.blk[0].offset)
        fe.load_state_dict(sd, strict=True)
        self.feature_extractor = fe.eval().to(self.hot.device)
        self.hot.dist_regressor.post_div = float(cfg.bf)          # inference_class.py:111-114
        self.hot.dist_regressor.return_norm_costs = False         # discarded by inference callers

    @torch.no_grad()
    def __call__(self, input_dict) -> np.ndarray:
        imgs = input_dict["imgs"]
        if isinstance(imgs, list):
            imgs = np.stack(imgs, axis=0)                          # [N, H, W, 3] uint8 (one frame)
        t = torch.from_numpy(np.ascontiguousarray(imgs))
        if t.dtype != torch.uint8:
            raise TypeError("InferencePipeline expects uint8 HWC camera images")
        g = getattr(self, "_graph", None)
        if g is not None and tuple(t.shape) == tuple(self._static_imgs.shape):
            self._static_imgs.copy_(t, non_blocking=True)          # the upload lands in the graph's input buffer
            g.replay()
            out = self._static_inv.squeeze(0).squeeze(0).cpu().numpy()
        else:
            out = self.forward_device(t.to(self.hot.device)).squeeze(0).squeeze(0).cpu().numpy()
        H.check_range("InferencePipeline: this frame")       # the download synchronised: the report covers the frame just computed
        return out

    @torch.no_grad()
    def forward_device(self, imgs_u8: torch.Tensor) -> torch.Tensor:
        """uint8 [N, H, W, 3] images of one frame on the device -> inv_dist / bf [1, 1, H, W] on the device."""
        f = self.feature_extractor(imgs_u8)                        # [N, C, Hi, Wi], channels-last storage
        inv, _ = self.hot(f.unsqueeze(0))
        return inv

    # ---- hipGraph replay of the whole chain for one frame (the robot's operating point: one frame at a time) ----
    @torch.no_grad()
    def capture(self, imgs_u8: torch.Tensor) -> None:
        """Capture extractor -> sweep -> regulator -> soft-argmin for images of this shape (uint8 [N, H, W, 3] on the device)
        into one hipGraph: the ~80 launches of a frame are then one submission.  __call__ replays it for inputs of that
        shape (the upload goes straight into the graph's input buffer); replay() is the device-side form."""
        if imgs_u8.dtype != torch.uint8 or not imgs_u8.is_cuda:
            raise TypeError("InferencePipeline.capture expects uint8 HWC camera images on the device")
        dev = self.hot.device
        self._graph = None
        self._static_imgs = imgs_u8.clone()
        self.forward_device(self._static_imgs)          # warm-up: weight packing, kernel attributes, module-owned buffers
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.forward_device(self._static_imgs)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._static_inv = self.forward_device(self._static_imgs)
        self._graph = graph

    @torch.no_grad()
    def replay(self, imgs_u8: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Replay the captured chain; returns the graph's static output (valid until the next replay)."""
        if getattr(self, "_graph", None) is None:
            raise RuntimeError("InferencePipeline.replay() before capture()")
        if imgs_u8 is not None and imgs_u8.data_ptr() != self._static_imgs.data_ptr():
            if tuple(imgs_u8.shape) != tuple(self._static_imgs.shape) or imgs_u8.dtype != torch.uint8:
                raise ValueError(f"InferencePipeline.replay(): captured for images {tuple(self._static_imgs.shape)} uint8, got "
                                 f"{tuple(imgs_u8.shape)} {imgs_u8.dtype}; capture() again")
            self._static_imgs.copy_(imgs_u8, non_blocking=True)
        self._graph.replay()
        return self._static_inv
