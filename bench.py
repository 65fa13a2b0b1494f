#!/usr/bin/env python3
"""bench.py -- stereo frames/s of the mvs_gi plane-sweep hot path on MI355X.

One "step" = one pass of the hot path (feats, grids, grid_masks, masks -> inv_dist:
fused sweep, post_vol, 3-D UNet regulator, soft-argmin) over one batch of synthetic
G16V frames per GPU, inputs resident in HBM.  N GPUs = N independent frame shards
(one process per GPU, no data-path collective; torch.distributed is used only for the
barrier and the max-over-ranks of the elapsed time) -> "scaling": "weak".

Prints ONE JSON line (rank 0) with the driver's contract plus
  "roofline":     the dominant kernel's achieved TFLOP/s (algorithmic conv FLOPs of its
                  launches / their HIP-event durations measured in the timed region)
                  against the gfx950 bf16-MFMA (or fp32-MFMA) peak;
  "cpu_baseline": the CPU oracle (torch-CPU restatement of the reference path) timed on this
                  host's cores on a bounded sample (rank 0, N == 1 only), whole path + per stage;
  "parity":       inverse-distance error of the benchmarked weights against that oracle (B=1,
                  outside the timed region);
and, at N == 1 (outside the headline's timed region, skipped with --no-extras):
  "extras":       the same path with the rig-constant cache off, as a hipGraph replay, at B=1
                  (latency), with the HIP feature extractor in front (images -> inverse distance)
                  and fed from pinned host memory (uint8 frames, double-buffered H2D);
  "configs":      the other BASELINE.json configurations (G16VV, E8, 4cam-32), each with
                  frames/s, ms/step, dominant kernel and roofline fraction.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

OTHER_SPLIT = {"f16x3": "bf16x3", "bf16x3": "f16x3"}
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "f16x3": 2500.0}   # MI355X_MICROARCH.md: dense fp32-MFMA / bf16-MFMA (= fp16-MFMA) peaks
DTYPE = {"f32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA operands, f32 accumulate/activations)",
         "f16x3": "f16x3 (split-fp16 MFMA operands: 11 + 11 significant bits, f32 accumulate/activations)"}
PEAK_HBM_GBS = 8000.0
EXTRA_CONFIGS = (("G16VV", 32), ("E8", 64), ("4cam-32", 16), ("E16-48-96", 32))      # (tag, frames per part (= per stream) at which the configuration runs best on MI355X: tools/config_batch_probe.py)
LIB = os.path.join(ROOT, "mvs_gi_amd", "libmvsgi_hip.so")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="G16V")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step, cut into --streams equal parts.  Default: G16V 256 "
                    "(round 6: 2 x 128; 2 x 64 until round 5: 6231 vs 6300 frames/s alternating on one box); the other configurations "
                    "--streams x their EXTRA_CONFIGS part size (4cam-32 at 256 frames does not fit the card)")
    ap.add_argument("--streams", type=int, default=2,
                    help="independent parts of a step's batch, each on its own HIP stream inside the step's one hipGraph "
                         "(StreamedHotPath: one part's kernel tails are filled by the other's launches; MI355X, G16V: 2 x 64 frames "
                         "+1.8 % over 1 x 128 and +3.3 % over 1 x 64); 1 = the whole batch on one stream")
    ap.add_argument("--mode", default="f16x3", choices=["bf16x3", "f16x3", "f32"],
                    help="conv arithmetic: split-fp16 MFMA (the library's default: 3 fp16 MFMAs per product, 11 + 11 bits per operand, "
                         "fp32 accumulate; level-0 convs in Winograd form), the same in the bf16 split (8 + 8 bits, fp32's range) or exact fp32 MFMA")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extras / configs blocks (rig cache off, hipGraph, B=1 latency, images -> inverse "
                         "distance, host feed, other BASELINE configs); they never touch the headline's timed region")
    ap.add_argument("--extra-steps", type=int, default=20, help="timed steps of each extras / configs measurement")
    ap.add_argument("--graph", action="store_true", help=argparse.SUPPRESS)         # kept: the default since round 3
    ap.add_argument("--eager", action="store_true",
                    help="headline from per-launch (Python / ctypes) submission instead of the captured hipGraph (HotPath.capture / replay)")
    ap.add_argument("--end-to-end", action="store_true", help=argparse.SUPPRESS)    # kept: now part of extras
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl=RCCL)")
    ap.add_argument("--settle-seconds", type=float, default=1.5,
                    help="after the W warm-up steps the same step keeps running untimed until this much wall time has passed since "
                         "the first step: the board reaches its power-capped clock (~1 s on MI355X) before the K timed steps, so "
                         "`value` is the rate the chip sustains (extras.sustained checks it over 3 s)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-dump-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-dump", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-dump-configs", default="", help=argparse.SUPPRESS)      # tags whose oracle frame is dumped beside --cpu-dump
    args = ap.parse_args(argv)
    if args.batch is None:
        args.batch = 256 if args.config == "G16V" else dict(EXTRA_CONFIGS).get(args.config, 32) * max(1, args.streams)
    return args


# ------------------------------------------------------------------------------------------
# distributed scaffolding (also exercised on CPU with gloo by tests/test_bench_sharding.py)
# ------------------------------------------------------------------------------------------
def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def gather_device_ordinals(local_rank: int, world: int, dev=None):
    """Every rank's device ordinal, in rank order (the line shows which GPUs the job really ran on)."""
    import torch
    import torch.distributed as dist
    on_cpu = dev is None or dist.get_backend() == "gloo"
    mine = torch.tensor([local_rank], dtype=torch.int64, device="cpu" if on_cpu else dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    return [int(t.item()) for t in allr]


def frame_shard(total_frames: int, world: int, rank: int):
    """Contiguous, balanced [lo, hi) shard of a frame sequence (strong-scaling helper for
    dataset playback; the bench itself is weak-scaled: every rank owns `batch` frames)."""
    base, rem = divmod(total_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def timed_steps(step_fn, sync_fn, steps: int, warmup: int, world: int, backend_ready: bool, device=None,
                per_rank=None):
    """W warmup steps, barrier+sync, exactly K timed steps, sync+barrier; returns the MAX
    elapsed seconds over ranks.  `per_rank` (a list) receives every rank's own elapsed seconds
    (rank order) so that a straggler is visible."""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        step_fn()
    sync_fn()
    if backend_ready:
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    own = time.perf_counter() - t0
    if backend_ready:
        dist.barrier()
    el = time.perf_counter() - t0
    if backend_ready:
        on_cpu = device is None or dist.get_backend() == "gloo"
        dv = "cpu" if on_cpu else device
        t = torch.tensor([el], dtype=torch.float64, device=dv)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        if per_rank is not None:
            mine = torch.tensor([own], dtype=torch.float64, device=dv)
            allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(allr, mine)
            per_rank[:] = [float(x.item()) for x in allr]
    elif per_rank is not None:
        per_rank[:] = [own]
    return el


# ------------------------------------------------------------------------------------------
# per-kernel attribution: HIP events around every conv launch of the timed region
# ------------------------------------------------------------------------------------------
class ConvProbe:
    """Wraps hip_ops.conv3d / conv3d_up2: records a HIP event pair (on the launch stream) around each
    call and the algorithmic FLOPs and kernel name of the launch."""

    def __init__(self, H):
        import torch
        self.H, self.torch = H, torch
        self.orig = H.conv3d
        self.orig_up2 = H.conv3d_up2
        self.orig_rs = H.conv3d_rs
        self.orig_wino = H.conv3d_wino
        self.orig_rs16 = H.conv3d_rs16
        self.orig_os = H.conv3d_out_split
        self.records = []
        self.hbm_records = []      # (kernel name, algorithmic bytes, start event, end event): the HBM-bound launches
        self.hbm_orig = {}
        self.enabled = False
        self.by_shape = False      # tools/layer_table.py: one row per (kernel, layer shape) instead of one per kernel

    def _tag(self, name, cin, cout, d, h, w, stride=1):
        return f"{name} @ {cin}->{cout} [{d},{h},{w}] s{stride}" if self.by_shape else name

    def _wrap_hbm(self, fname, describe):
        """Event pair around H.<fname>; describe(args, kwargs, result) -> (kernel name, algorithmic bytes)."""
        H, torch = self.H, self.torch
        orig = getattr(H, fname)
        self.hbm_orig[fname] = orig

        def probed(*a, **k):
            if not self.enabled:
                return orig(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(*a, **k)
            e.record()
            name, nbytes = describe(a, k, r)
            self.hbm_records.append((name, float(nbytes), s, e))
            return r
        setattr(H, fname, probed)

    def __enter__(self):
        H, torch = self.H, self.torch

        def probed(x, w_oidhw, w_packed, scale, shift, res=None, stride=1, neg_slope=0.01, impl=H.CONV_AUTO, out=None):
            if not self.enabled:
                return self.orig(x, w_oidhw, w_packed, scale, shift, res, stride, neg_slope, impl, out)
            B, D, Hh, W, Cin = x.shape
            Cout = scale.numel()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = self.orig(x, w_oidhw, w_packed, scale, shift, res, stride, neg_slope, impl, out)
            e.record()
            vox = y.numel() // Cout
            self.records.append((self._tag(H.conv3d_variant(B, Cin, D, Hh, W, Cout, stride, impl), Cin, Cout, D, Hh, W, stride), 2.0 * 27 * Cin * Cout * vox, s, e,
                                 4.0 * (x.numel() + y.numel() * (2 if res is not None else 1))))
            return y

        def probed_up2(x, w_packed_b3, scale, shift, res=None, neg_slope=0.01, out=None, w_layout=H.CONV_BF16X3):
            if not self.enabled:
                return self.orig_up2(x, w_packed_b3, scale, shift, res, neg_slope, out, w_layout)
            B, Dl, Hl, Wl, Cin = x.shape
            Cout = scale.numel()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = self.orig_up2(x, w_packed_b3, scale, shift, res, neg_slope, out, w_layout)
            e.record()
            vox = y.numel() // Cout
            self.records.append((self._tag(H.conv3d_up2_variant(B, Cin, Dl, Hl, Wl, Cout, w_layout), Cin, Cout, 2 * Dl, 2 * Hl, 2 * Wl), 2.0 * 27 * Cin * Cout * vox, s, e,
                                 4.0 * (x.numel() + y.numel() * (2 if res is not None else 1))))
            return y

        def probed_rs(x, w_packed_rs, scale, shift, res=None, neg_slope=0.01, out=None, out_f32=False):
            if not self.enabled:
                return self.orig_rs(x, w_packed_rs, scale, shift, res, neg_slope, out, out_f32)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = self.orig_rs(x, w_packed_rs, scale, shift, res, neg_slope, out, out_f32)
            e.record()
            nv = x.B * x.D * x.H * x.W
            self.records.append(("conv3d_rs32_kernel<%d%s>" % (1 if out_f32 else 0, ", true" if x.fmt == "f16" else ", false"),
                                 2.0 * 27 * x.C * scale.numel() * nv, s, e, 4.0 * nv * (x.C + scale.numel() * (2 if res is not None else 1))))
            return y

        def probed_wino(x, w_packed, scale, shift, res=None, neg_slope=0.01, out=None, out_f32=False):
            if not self.enabled:
                return self.orig_wino(x, w_packed, scale, shift, res, neg_slope, out, out_f32)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = self.orig_wino(x, w_packed, scale, shift, res, neg_slope, out, out_f32)
            e.record()
            nv = x.B * x.D * x.H * x.W
            # algorithmic FLOPs = the direct convolution's (the Winograd form issues 16 / 36 of its matrix instructions)
            self.records.append(("conv3d_wino32_kernel<0, %s, %s, %d, %s>" % ("true" if res is not None else "false", "true" if out_f32 else "false", x.D,
                                                                              "true" if x.fmt == "f32p" else "false"),
                                 2.0 * 27 * 32 * 32 * nv, s, e, 4.0 * nv * 32 * (3 if res is not None else 2)))
            return y

        def probed_os(x, w_packed_b3, scale, shift, out, res=None, stride=1, neg_slope=0.01, fmt="bf16"):
            if not self.enabled:
                return self.orig_os(x, w_packed_b3, scale, shift, out, res, stride, neg_slope, fmt)
            B, D, Hh, W, Cin = x.shape
            Cout = scale.numel()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = self.orig_os(x, w_packed_b3, scale, shift, out, res, stride, neg_slope, fmt)
            e.record()
            self.records.append((self._tag(H.conv3d_variant(B, Cin, D, Hh, W, Cout, stride, H.CONV_BF16X3 | (H.CONV_F16 if fmt == "f16" else 0)) + " [split-padded out]", Cin, Cout, D, Hh, W, stride),
                                 2.0 * 27 * Cin * Cout * out.B * out.D * out.H * out.W, s, e,
                                 4.0 * (x.numel() + out.B * out.D * out.H * out.W * Cout * (2 if res is not None else 1))))
            return y

        def probed_rs16(x, w_packed_rs, scale, shift, neg_slope=0.01, out=None, out_split=None):
            if not self.enabled:
                return self.orig_rs16(x, w_packed_rs, scale, shift, neg_slope, out, out_split)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = self.orig_rs16(x, w_packed_rs, scale, shift, neg_slope, out, out_split)
            e.record()
            self.records.append(("conv3d_rs16_kernel<%s%s>" % ("true" if out_split is not None else "false", ", true" if x.fmt == "f16" else ", false"),
                                 2.0 * 27 * 16 * 16 * x.B * x.D * x.H * x.W, s, e, 4.0 * 32 * x.B * x.D * x.H * x.W))
            return y

        self.orig_s2rs = H.conv3d_s2rs

        def probed_s2rs(x, w_packed, shift, out, neg_slope=0.01, unscale=1.0, out_f32p=False):
            if not self.enabled:
                return self.orig_s2rs(x, w_packed, shift, out, neg_slope, unscale, out_f32p)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = self.orig_s2rs(x, w_packed, shift, out, neg_slope, unscale, out_f32p)
            e.record()
            nvo = out.B * out.D * out.H * out.W
            self.records.append(("conv3d_s2rs_kernel<4, 2%s>" % ((", true, true" if out_f32p else ", true, false") if x.fmt == "f16" else ", false, false"), 2.0 * 27 * 16 * 32 * nvo, s, e, 4.0 * (16 * x.B * x.D * x.H * x.W + 32 * nvo)))
            return y

        orig_poly, orig_up2s = H.conv3d_up2_poly, H.conv3d_up2_out_split
        self.hbm_orig["conv3d_up2_poly"], self.hbm_orig["conv3d_up2_out_split"] = orig_poly, orig_up2s

        def probed_poly(x, plan, scale, shift, neg_slope=0.01, out=None):
            if not self.enabled:
                return orig_poly(x, plan, scale, shift, neg_slope, out)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = orig_poly(x, plan, scale, shift, neg_slope, out)
            e.record()
            vox = y.numel() // 16
            self.records.append(("conv3d_rs32_kernel<2%s> + up2_face_kernel + up2_edge_kernel (polyphase out_costs.0)" % (", true" if x.fmt == "f16" else ", false"),
                                 2.0 * 27 * 32 * 16 * vox, s, e, 4.0 * (x.B * x.D * x.H * x.W * 32 + y.numel())))
            return y

        def probed_up2s(x, w_packed_b3, scale, shift, out, res=None, neg_slope=0.01, w_layout=H.CONV_BF16X3):
            if not self.enabled:
                return orig_up2s(x, w_packed_b3, scale, shift, out, res, neg_slope, w_layout)
            B, Dl, Hl, Wl, Cin = x.shape
            Cout = scale.numel()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = orig_up2s(x, w_packed_b3, scale, shift, out, res, neg_slope, w_layout)
            e.record()
            vox = out.B * out.D * out.H * out.W
            self.records.append((self._tag(H.conv3d_up2_variant(B, Cin, Dl, Hl, Wl, Cout, w_layout) + " [split-padded out]", Cin, Cout, 2 * Dl, 2 * Hl, 2 * Wl),
                                 2.0 * 27 * Cin * Cout * vox, s, e, 4.0 * (x.numel() + vox * Cout * (2 if res is not None else 1))))
            return y
        orig_polys, orig_heads = H.conv3d_up2_poly_split, H.conv3d_head_split
        self.hbm_orig["conv3d_up2_poly_split"], self.hbm_orig["conv3d_head_split"] = orig_polys, orig_heads

        def probed_polys(x, plan, scale, shift, out, neg_slope=0.01, **kw):
            if not self.enabled:
                return orig_polys(x, plan, scale, shift, out, neg_slope, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = orig_polys(x, plan, scale, shift, out, neg_slope, **kw)
            e.record()
            vox = out.B * out.D * out.H * out.W
            # the main kernel the library's dispatcher launches for this geometry and batch (include/mvsgi.h, mvsgi_conv3d_up2_poly_fmt)
            wino = kw.get("wino") or (x.fmt == "f16" and not kw.get("direct") and H.conv3d_up2_poly_wino_pays(x.B, x.D, x.H, x.W))
            main = ("conv3d_wino_up2_kernel + up2_face_kernel + up2_edge_kernel (polyphase out_costs.0 in Winograd form, split-padded out)" if wino else
                    "conv3d_rs32_kernel<3%s> + up2_face_kernel + up2_edge_kernel (polyphase out_costs.0, split-padded out)" % (", true" if x.fmt == "f16" else ", false"))
            self.records.append((main, 2.0 * 27 * 32 * 16 * vox, s, e, 4.0 * (x.B * x.D * x.H * x.W * 32 + vox * 16)))
            return y

        def probed_heads(x, w_packed, scale, shift, neg_slope=1.0, out=None, f16=False):
            if not self.enabled:
                return orig_heads(x, w_packed, scale, shift, neg_slope, out, f16)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = orig_heads(x, w_packed, scale, shift, neg_slope, out, f16)
            e.record()
            vox = x.B * x.D * x.H * x.W
            self.records.append(("conv3d_head_split_kernel<%s>" % ("true" if f16 else "false"), 2.0 * 27 * x.C * vox, s, e, 4.0 * vox * (x.C + 1)))
            return y
        H.conv3d_up2_poly_split, H.conv3d_head_split = probed_polys, probed_heads
        H.conv3d_up2_poly, H.conv3d_up2_out_split = probed_poly, probed_up2s
        H.conv3d = probed
        H.conv3d_up2 = probed_up2
        H.conv3d_rs16 = probed_rs16
        H.conv3d_s2rs = probed_s2rs
        H.conv3d_rs = probed_rs
        H.conv3d_wino = probed_wino
        H.conv3d_out_split = probed_os

        # ---- the HBM-bound launches of the path (K1 sweep, K4 soft-argmin, layout transposes) ----
        def sweep_bytes(feats, grids, out_vox, C):
            # per output voxel N x 8 B of grid + 1 validity byte + C x 4 B written; the feature maps once
            N = feats.shape[1]
            return out_vox * (N * 8 + 1 + C * 4) + feats.numel() * 4

        def d_sweep_split(a, k, r):
            feats, grids = a[0], a[1]
            return f"sweep_std_nhwc_v_kernel<{feats.shape[1]}, true>", sweep_bytes(feats, grids, r.B * r.D * r.H * r.W, r.C)

        def d_sweep_valid(a, k, r):
            feats, grids = a[0], a[1]
            return f"sweep_std_nhwc_v_kernel<{feats.shape[1]}, false>", sweep_bytes(feats, grids, r.numel() // r.shape[-1], r.shape[-1])

        def d_sweep_std(a, k, r):
            feats, grids, masks = a[0], a[1], a[3]
            vox = r.numel() // r.shape[-1]
            return "sweep_std (masks re-sampled)", sweep_bytes(feats, grids, vox, r.shape[-1]) + vox * feats.shape[1] * 4 + masks.numel() * 4

        def d_sweep_cat(a, k, r):
            feats = a[0]
            vox = r.numel() // r.shape[-1]
            return "sweep_cat_nhwc_kernel", vox * (feats.shape[1] * 8 + r.shape[-1] * 4) + feats.numel() * 4

        def d_softargmin(a, k, r):
            costs, scale = a[0], a[2]
            inv, pr = r
            D = costs.shape[1]
            name = f"softargmin_rows_kernel<{16 if D <= 16 else (32 if D <= 32 else 0)}>" if scale == 2 else "softargmin_kernel"
            return name, costs.numel() * 4 + inv.numel() * 4 + (pr.numel() * 4 if pr is not None else 0)

        def d_transpose(name):
            return lambda a, k, r: (name, 8 * r.numel())

        def d_resize(a, k, r):
            return "resize_trilinear_kernel", 4 * (a[0].numel() + r.numel())

        for fname, d in (("sweep_std_valid_split", d_sweep_split), ("sweep_std_valid", d_sweep_valid), ("sweep_std", d_sweep_std),
                         ("sweep_cat", d_sweep_cat), ("softargmin", d_softargmin), ("ncdhw_to_ndhwc", d_transpose("ncv_to_nvc_kernel")),
                         ("ndhwc_to_ncdhw", d_transpose("nvc_to_ncv_kernel")), ("resize_trilinear", d_resize)):
            self._wrap_hbm(fname, d)
        orig_nhwc = H._feats_nhwc
        self.hbm_orig["_feats_nhwc"] = orig_nhwc

        def probed_nhwc(feats):
            v = feats.permute(0, 1, 3, 4, 2)
            if not self.enabled or (v.is_contiguous() and v.data_ptr() % 16 == 0):
                return orig_nhwc(feats)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_nhwc(feats)
            e.record()
            self.hbm_records.append(("ncv_to_nvc_kernel (feats)", 8.0 * r.numel(), s, e))
            return r
        H._feats_nhwc = probed_nhwc
        return self

    def __exit__(self, *a):
        self.H.conv3d = self.orig
        self.H.conv3d_up2 = self.orig_up2
        self.H.conv3d_rs = self.orig_rs
        self.H.conv3d_wino = self.orig_wino
        self.H.conv3d_rs16 = self.orig_rs16
        self.H.conv3d_s2rs = self.orig_s2rs
        self.H.conv3d_out_split = self.orig_os
        for fname, orig in self.hbm_orig.items():
            setattr(self.H, fname, orig)

    def hbm_summary(self):
        agg = {}
        for name, nbytes, s, e in self.hbm_records:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += nbytes
            a[2] += s.elapsed_time(e)
        return agg

    def summary(self):
        """kernel name -> [launches, algorithmic FLOPs, ms, layer-granular bytes (input + output (+ residual), fp32)]"""
        agg = {}
        for name, flops, s, e, nbytes in self.records:
            ms = s.elapsed_time(e)
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += ms
            a[3] += nbytes
        return agg


def kernels_block(agg, hbm_agg):
    """The line's `kernels` object: every attributed launch of the timed steps with its bound and achieved rate."""
    out = {}
    for k, v in sorted(agg.items()):
        ent = {"launches": v[0], "avg_us": round(v[2] / v[0] * 1e3, 2), "tflops": round(v[1] / (v[2] * 1e-3) / 1e12, 2),
               "GBps": round(v[3] / (v[2] * 1e-3) / 1e9, 1)}
        if "head" in k:
            ent.update(bound="hbm", hbm_frac=round(ent["GBps"] / PEAK_HBM_GBS, 4))
        else:
            ent["bound"] = "mfma"
        out[k] = ent
    for k, v in sorted(hbm_agg.items()):
        gbps = v[1] / (v[2] * 1e-3) / 1e9
        out[k] = {"launches": v[0], "avg_us": round(v[2] / v[0] * 1e3, 2), "bound": "hbm", "GBps": round(gbps, 1),
                  "hbm_frac": round(gbps / PEAK_HBM_GBS, 4), "MB_per_launch": round(v[1] / v[0] / 1e6, 2)}
    return out


def read_pmc_traffic(kernel_name: str, frames_per_launch=None, tag: str = "G16V"):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 --pmc summary
    (profiles/pmc_traffic.json, written by tools/summarize_rocprof.py from separate --pmc passes of this same
    command): (bytes | None, source string).  The counters cannot be read from inside the process, so the line
    says where the number comes from instead of presenting it as measured in this run."""
    # one summary per configuration (a kernel NAME serves different layers in different configurations): G16V's is
    # profiles/pmc_traffic.json, the others' profiles/pmc_traffic_<tag>.json; none -> null
    fn = "pmc_traffic.json" if tag == "G16V" else f"pmc_traffic_{tag}.json"
    p = os.path.join(ROOT, "profiles", fn)
    try:
        d = json.load(open(p))
        key = kernel_name.split(" [")[0]
        # (summaries written before the kernels carried their split as a template argument name them without the trailing ", false")
        def one(k):
            return (d.get(k) or d.get(k.replace(", false>", ">")) or {}).get("hbm_bytes_per_launch")
        v = one(key)
        if v is None and " + " in key:      # a timed unit of several kernels (the polyphase layer: main + face + edge kernels): their sum
            parts = [one(k.split(" (")[0].strip()) for k in key.split(" + ")]
            parts = [one(next((n for n in d if n.startswith(k.split(" (")[0].strip())), "")) if p_ is None else p_
                     for k, p_ in zip(key.split(" + "), parts)]
            v = sum(parts) if all(p_ is not None for p_ in parts) else None
        src = f"profiles/{fn} ({d.get('_source', 'rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE')})"
        # the counters were taken at the summary's own launch size; these kernels' bytes are proportional to the frames of a launch
        n0 = d.get("_frames_per_launch")
        if v is not None and n0 and frames_per_launch and int(n0) != int(frames_per_launch):
            v = v * frames_per_launch / n0
            src += f", measured at {n0} frames per launch and scaled to this run's {frames_per_launch}"
        return v, (src if v is not None else None)
    except Exception:
        return None, None


# ------------------------------------------------------------------------------------------
def effective_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box reports 256 CPUs but grants 16; oversubscribing OpenMP threads there is
    pathologically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline_subprocess(args, dump_path):
    """Run the CPU baseline in a child process, before this process touches the GPU, with a
    hard timeout so a slow host can never stall the benchmark.  The child also writes the oracle's
    inv_dist of the seed-0 frame to `dump_path` (.npy) for the parity block."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--config", args.config,
           "--cpu-seconds", str(args.cpu_seconds), "--cpu-dump", dump_path]
    if not args.no_extras:      # one oracle frame of each other configuration: the parity reference of the configs{} entries
        cmd += ["--cpu-dump-configs", ",".join(t for t, _ in EXTRA_CONFIGS if t != args.config)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_seconds * 6 + 120)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:   # report, never fail the GPU measurement
        return {"value": None, "unit": "frames/s", "cores": effective_cores(), "kind": "port",
                "sample": f"CPU baseline did not finish: {type(e).__name__}"}


def oracle_dump_subprocess(args, dump_path):
    """One oracle forward of the seed-0 frame in a child process (before this process touches the GPU): the parity reference of
    a multi-rank line, which carries no CPU baseline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-dump-only", "--config", args.config, "--cpu-dump", dump_path]
    try:
        subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    except Exception:
        pass


def config_dump_path(dump_path, tag):
    return f"{dump_path}.{tag}.npy"


def oracle_dump(cfg, dump_path):
    import numpy as np
    import torch
    from mvs_gi_amd import synth
    from oracle import mvsgi_oracle as O
    torch.set_num_threads(effective_cores())
    inp = O.to_torch(synth.make_inputs(cfg, seed=0, batch=1))
    w = O.to_torch(synth.make_weights(cfg, seed=0))
    ref = O.hot_path(inp["feats"], inp["grids"], inp["grid_masks"], inp["masks"], w, cfg.builder,
                     cfg.dist_cands, cfg.bf, cfg.interp_scale_factor, cfg.pre_interp)
    np.save(dump_path, ref.numpy())


def cpu_baseline(cfg, seconds: float, dump_path=None):
    """The oracle (CPU restatement of the reference's PyTorch path) on this host's cores:
    B=1 frames of the same workload until ~`seconds` have elapsed (at least 3 frames), then the three
    stages on their own (min / median of 3)."""
    import numpy as np
    import torch
    from mvs_gi_amd import synth
    from oracle import mvsgi_oracle as O
    cores = effective_cores()
    torch.set_num_threads(cores)
    inp = O.to_torch(synth.make_inputs(cfg, seed=0, batch=1))
    w = O.to_torch(synth.make_weights(cfg, seed=0))

    def one():
        return O.hot_path(inp["feats"], inp["grids"], inp["grid_masks"], inp["masks"], w, cfg.builder,
                          cfg.dist_cands, cfg.bf, cfg.interp_scale_factor, cfg.pre_interp)
    ref = one()
    if dump_path:
        np.save(dump_path, ref.numpy())
    n, t0, per = 0, time.perf_counter(), []
    while True:
        t1 = time.perf_counter()
        one()
        per.append(time.perf_counter() - t1)
        n += 1
        el = time.perf_counter() - t0
        if (el >= seconds and n >= 3) or n >= 200:
            break
    stages = {}
    with torch.no_grad():
        def timed(fn, reps=3):
            ts, out = [], None
            for _ in range(reps):
                t1 = time.perf_counter()
                out = fn()
                ts.append((time.perf_counter() - t1) * 1e3)
            return out, {"min_ms": round(min(ts), 2), "median_ms": round(sorted(ts)[len(ts) // 2], 2)}
        sweep = (lambda: O.sweep_std_masked(inp["feats"], inp["grids"], inp["grid_masks"], inp["masks"])) \
            if cfg.builder == "std" else (lambda: O.sweep_concat(inp["feats"], inp["grids"]))
        vol_raw, stages["cv_builder.sweep"] = timed(sweep)
        vol, stages["cv_builder.post_vol"] = timed(lambda: O.post_vol(vol_raw, w["cv_builder"]))
        costs, stages["cv_regulator"] = timed(lambda: O.regulator_forward(vol, w["cv_regulator"]))
        _, stages["dist_regressor"] = timed(lambda: O.soft_argmin(costs, cfg.dist_cands, cfg.bf, cfg.interp_scale_factor,
                                                                  cfg.pre_interp))
    per.sort()
    return {"value": round(n / el, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "frame_ms_min": round(per[0] * 1e3, 1), "frame_ms_median": round(per[len(per) // 2] * 1e3, 1),
            "stages": stages,
            "sample": f"{n} single-frame (B=1) {cfg.tag} passes of oracle/mvsgi_oracle.py (torch {torch.__version__} "
                      f"CPU, fp32) in {el:.1f} s after 1 warm-up; stages: min / median of 3"}


# ------------------------------------------------------------------------------------------
def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start the N ranks as a CHILD torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1 at a free port), relay its output and return its exit code.  Runs before this process has
    touched the GPU or imported torch; never an exec (a process must not replace itself once anything may have initialised HIP)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, effective_cores() // args.gpus)))     # N ranks share this host's cores
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, env=env)
    return r.returncode


def ensure_library(local_rank: int):
    """The library normally travels with the tree (built by __graft_entry__.build()).  When it is missing or
    stale, local rank 0 builds it (to a temporary path, renamed into place -- never a half-written file) and the
    other ranks wait until the finished file loads and passes the ABI check; a failed or overdue build exits
    non-zero on every rank."""
    from mvs_gi_amd import _lib

    def loadable():
        try:
            _lib.load()
            return True
        except Exception:
            return False
    import __graft_entry__
    stale = __graft_entry__._needs_rebuild()      # content hash of csrc/ against the one the library was built from
    if not stale and loadable():
        return
    if local_rank == 0:
        if stale and os.path.isfile(LIB):
            print("bench.py: libmvsgi_hip.so is older than its sources: rebuilding", file=sys.stderr, flush=True)
        __graft_entry__.build(force=False)
        return
    deadline = time.time() + 900
    while time.time() < deadline:
        if not __graft_entry__._needs_rebuild() and loadable():
            return
        time.sleep(1.0)
    raise SystemExit("bench.py: libmvsgi_hip.so did not appear (build by local rank 0 failed or timed out)")


def make_feats(B, shape, rng, dev, torch, np, nchw=False):
    """[B, N, C, Hi, Wi] synthetic feature maps.  Default: channels-last storage ([B, N, Hi, Wi, C] presented with the
    reference's shape), which is what the HIP feature extractor in front of this path emits; nchw=True is the reference
    extractor's contiguous layout, which the sweep transposes once per step (extras.feats_nchw)."""
    _, N, C, Hi, Wi = shape
    if nchw:
        return torch.from_numpy(rng.standard_normal((B, N, C, Hi, Wi), dtype=np.float32)).to(dev)
    return torch.from_numpy(rng.standard_normal((B, N, Hi, Wi, C), dtype=np.float32)).to(dev).permute(0, 1, 4, 2, 3)


def measure_path(cfg, B, mode, steps, warmup, dev, H, HotPath, synth, torch, np, rng, graph=False, kernels=False, streams=1, ref_dump=None):
    """frames/s, ms/step and per-conv-kernel attribution of one configuration at one batch size (single process,
    outside the headline's timed region).  frames_per_s / the kernel table: B frames as one launch chain on one stream;
    graph_replay_*: as the headline submits a step -- `streams` parts of B frames each in one hipGraph."""
    from mvs_gi_amd.configs import path_gflop
    from mvs_gi_amd.pipeline import StreamedHotPath
    H.set_conv_mode(mode)
    inp = synth.make_inputs(cfg, seed=0, batch=1)
    shp = StreamedHotPath(cfg, synth.make_weights(cfg, seed=0), inp, device=dev, n_streams=max(1, streams))
    hp = shp.parts[0]
    feats_all = make_feats(B * len(shp.parts), inp["feats"].shape, rng, dev, torch, np)
    feats = feats_all[:B]

    def step():
        hp(feats)

    def sync():
        torch.cuda.synchronize(dev)
    with ConvProbe(H) as probe:
        for _ in range(warmup):
            step()
        sync()
        probe.enabled = True
        el = timed_steps(step, sync, steps, 0, 1, False, dev)
        probe.enabled = False
        agg, hbm_agg = probe.summary(), probe.hbm_summary()
    dname, (dn, dflops, dms, _) = max(agg.items(), key=lambda kv: kv[1][2])
    ach = dflops / (dms * 1e-3) / 1e12
    res = {"frames_per_step": B, "frames_per_s": round(B * steps / el, 2), "ms_per_step": round(el / steps * 1e3, 4),
           "path_tflops": round(B * steps / el * path_gflop(cfg) / 1e3, 2),
           "dominant_kernel": dname, "dominant_avg_us": round(dms / dn * 1e3, 2),
           "dominant_tflops": round(ach, 2), "frac": round(ach / PEAK_TFLOPS[mode], 4)}
    if kernels:
        res["kernels"] = kernels_block(agg, hbm_agg)
    if graph:
        try:
            shp.capture(feats_all)
            gel = timed_steps(lambda: shp.replay(), sync, steps, warmup, 1, False, dev)
            res["graph_replay_frames_per_s"] = round(feats_all.shape[0] * steps / gel, 2)
            res["graph_replay_ms_per_step"] = round(gel / steps * 1e3, 4)
            res["graph_replay_frames_per_step"] = feats_all.shape[0]
        except Exception as e:
            res["graph_replay_error"] = f"{type(e).__name__}: {e}"
            torch.cuda.synchronize(dev)
    if ref_dump is not None and os.path.isfile(ref_dump):
        # parity AT THE OPERATING POINT: the oracle's seed-0 frame in every slot of the step's batch (other units and brick shapes
        # than at one frame), first and last frame against the oracle
        try:
            ref = np.load(ref_dump)
            f0 = torch.from_numpy(inp["feats"]).to(dev).expand(feats_all.shape[0], -1, -1, -1, -1).contiguous()
            outs = shp.replay(f0) if getattr(shp, "_graph", None) is not None else shp(f0)
            torch.cuda.synchronize(dev)
            first, last = outs[0][0][:1].cpu().numpy(), outs[-1][0][-1:].cpu().numpy()
            res["parity"] = {"max_rel": float(max(np.abs(first - ref).max(), np.abs(last - ref).max()) / np.abs(ref).max()),
                             # per-pixel: max over pixels of |d| / |ref| (inv_dist >= 0.96): what a depth consumer sees; max_rel divides by the map's maximum
                             "max_pixel_rel": float(max((np.abs(first - ref) / np.abs(ref)).max(), (np.abs(last - ref) / np.abs(ref)).max())),
                             "saturation_flags": H.saturation_flags(clear=True),      # the fp16 split's range report for this step (0 = no clamp engaged)
                             "mean_l1_rel": float(np.abs(first - ref).mean() / np.abs(ref).mean()), "bar": 1e-3, "mode": mode,
                             "frames_checked": [0, int(feats_all.shape[0]) - 1], "frames_per_step": int(feats_all.shape[0]),
                             "ref": "oracle/mvsgi_oracle.py (pinned to the reference goldens); the same step as measured"}
            del f0, outs
        except Exception as e:
            res["parity"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.synchronize(dev)
    del hp, shp, feats, feats_all
    torch.cuda.empty_cache()
    return res


def main(argv=None):
    args = parse_args(argv)
    rank, local_rank, world = dist_env()
    if args.cpu_baseline_only:
        from mvs_gi_amd.configs import CONFIGS as _C
        res_cpu = cpu_baseline(_C[args.config], args.cpu_seconds, args.cpu_dump)
        for tag in [t for t in args.cpu_dump_configs.split(",") if t]:
            try:
                oracle_dump(_C[tag], config_dump_path(args.cpu_dump, tag))
            except Exception:      # a missing dump only drops that entry's parity field
                pass
        print(json.dumps(res_cpu), flush=True)
        return
    if args.cpu_dump_only:
        from mvs_gi_amd.configs import CONFIGS as _C
        oracle_dump(_C[args.config], args.cpu_dump)
        return
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            raise SystemExit(self_launch(args, sys.argv[1:] if argv is None else argv))
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    cpu_res, ref_dump = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ref_dump = os.path.join(tempfile.mkdtemp(prefix="mvsgi_bench_"), "oracle_inv_dist.npy")
        cpu_res = cpu_baseline_subprocess(args, ref_dump)      # before any GPU initialisation
    elif rank == 0 and world > 1:
        # a multi-rank line carries no CPU baseline, but it does carry an error figure: one oracle frame, before any GPU call
        ref_dump = os.path.join(tempfile.mkdtemp(prefix="mvsgi_bench_"), "oracle_inv_dist.npy")
        oracle_dump_subprocess(args, ref_dump)
    ensure_library(local_rank)
    import numpy as np
    import torch
    import torch.distributed as dist
    from mvs_gi_amd import hip_ops as H, synth
    from mvs_gi_amd.configs import CONFIGS, path_gflop
    from mvs_gi_amd.pipeline import HotPath, StreamedHotPath

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    if os.environ.get("MVSGI_BENCH_SHARE_GPU"):      # test hook: N ranks on one device (use --backend gloo)
        local_rank = 0
    if world > 1:
        torch.set_num_threads(max(1, effective_cores() // world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend_ready = False
    if world > 1:
        backend = args.backend or "nccl"
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        backend_ready = True
    devices = gather_device_ordinals(local_rank, world, dev) if backend_ready else [local_rank]

    cfg = CONFIGS[args.config]
    B = args.batch
    H.set_conv_mode(args.mode)
    # the fp16 split's range report: a measurement run REPORTS it (parity.saturation_flags, the configs' parity blocks) instead of
    # stopping on it; the library's default for a deployment is MVSGI_RANGE_CHECK=raise
    if "MVSGI_RANGE_CHECK" not in os.environ:
        H.set_range_check("warn")
    peak = PEAK_TFLOPS[args.mode]
    inp = synth.make_inputs(cfg, seed=0, batch=1)
    weights = synth.make_weights(cfg, seed=0)
    S = max(1, min(args.streams, B))
    while B % S:                                 # a batch that does not split evenly runs in fewer parts
        S -= 1
    shp = StreamedHotPath(cfg, weights, inp, device=dev, n_streams=S)
    hp = shp.parts[0]                            # one part: what the per-kernel attribution, the parity frame and the extras run on
    rng = np.random.default_rng(1000 + rank)     # every rank owns different frames
    feats = make_feats(B, inp["feats"].shape, rng, dev, torch, np)
    out = {}

    def step():
        out["inv"], out["pr"] = shp(feats)[-1]

    def step_serial():
        # the same launches, part after part on ONE stream: per-kernel event pairs then time a kernel alone on the chip
        for part, f in zip(shp.parts, feats.split(B // S, dim=0)):
            out["inv"], out["pr"] = part(f)

    def sync():
        torch.cuda.synchronize(dev)

    per_rank = []
    t_first = time.perf_counter()
    for _ in range(args.warmup):
        step()
    sync()
    # The step as the product submits it: the ~45 launches of one forward captured once into a hipGraph (HotPath.capture,
    # during warm-up) and replayed per batch on the input buffer resident in HBM -- the same kernels on the same data as the
    # per-launch submission, without ~5 us of launch gap between them (extras.eager_launches keeps that number).
    use_graph = not args.eager
    graph_note = None
    if use_graph:
        try:
            shp.capture(feats)
        except Exception as e:      # never lose the run to the submission mode: per-launch submission measures the same kernels
            use_graph, graph_note = False, f"hipGraph capture failed ({type(e).__name__}: {e}); per-launch submission"
            torch.cuda.synchronize(dev)
    if use_graph:
        def timed_step():
            out["inv"], out["pr"] = shp.replay()[-1]
        for _ in range(max(2, args.warmup // 2)):
            timed_step()
        sync()
    else:
        timed_step = step
    # settle: the same step, untimed, until the board sits at its power-capped clock (it ramps for ~1 s from idle; a 0.25 s timed
    # region right behind a short warm-up would catch the chip still above the clock it sustains)
    settle_steps = 0
    while time.perf_counter() - t_first < args.settle_seconds:
        timed_step()
        settle_steps += 1
        if settle_steps % 8 == 0:
            sync()
    sync()
    # the timed region: K steps, nothing but the path's own launches in it
    el = timed_steps(timed_step, sync, args.steps, 0, world, backend_ready, dev, per_rank)
    # per-kernel attribution: the same K steps once more with a HIP event pair around every conv launch (on the launch
    # stream).  Kept out of the region above: ~70 event records per step cost ~3 % of it.
    with ConvProbe(H) as probe:
        probe.enabled = True
        el_ev = timed_steps(step_serial, sync, args.steps, 0, world, backend_ready, dev)
        probe.enabled = False
        sync()
        agg = probe.summary()
        hbm_agg = probe.hbm_summary()

    assert torch.isfinite(out["inv"]).all()
    frames = B * world * args.steps
    value = frames / el
    # dominant kernel = largest total time among the conv variants
    dom = max(agg.items(), key=lambda kv: kv[1][2])
    dname, (dn, dflops, dms, _) = dom
    achieved = dflops / (dms * 1e-3) / 1e12
    conv_ms = sum(v[2] for v in agg.values())
    attributed_ms = conv_ms + sum(v[2] for v in hbm_agg.values())
    traffic, traffic_src = read_pmc_traffic(dname, B // S, args.config)
    res = {
        "metric": f"stereo frames/sec/GPU ({cfg.tag}, {cfg.num_cams}-cam, D={cfg.num_cands}) + inv-dist L1 vs reference",
        "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE[args.mode], "data": "synthetic",
        "config": {"workload": f"{cfg.tag}: {cfg.num_cams} cams, D={cfg.num_cands}, builder={cfg.builder}, "
                               f"regulator=({cfg.reg_in_chs},{cfg.reg_f_int_chs}), feats {cfg.feat_hw}, cv {cfg.cv_hw}",
                   "frames_per_gpu_per_step": B, "parallelism": f"frame-sharded x{world} (no collective)",
                   "streams": (f"{S} independent parts of {B // S} frames, each on its own HIP stream inside the step's graph (fork / join): one "
                               "part's kernel tails are filled by the other's launches; extras.one_stream: the whole batch on one stream"
                               if S > 1 else "one stream"),
                   "settle": f"{settle_steps} untimed steps behind the {args.warmup} warm-up steps ({args.settle_seconds} s since the first step): "
                             "the timed region starts at the power-capped clock (extras.sustained)",
                   "submission": ("one hipGraph replay per step (captured in warm-up; extras.eager_launches: per-launch submission)" if use_graph
                                  else (graph_note or "per-launch submission from Python / ctypes")),
                   "feats_layout": "channels-last storage, as the HIP feature extractor emits (extras.feats_nchw: contiguous NCHW)",
                   "rig_constants": "grids / grid_masks / masks resident in HBM (one set shared by the batch); validity byte and packed weights lowered "
                                    "once during warm-up (DESIGN.md section 1); extras.rig_cache_off re-samples them every step",
                   "path_gflop_per_frame": round(path_gflop(cfg), 2)},
        "frames_per_sec_per_gpu": round(value / world, 2),
        "per_rank_frames_per_s": {"min": round(B * args.steps / max(per_rank), 2), "max": round(B * args.steps / min(per_rank), 2)}
        if per_rank else None,
        "path_tflops": round(value * path_gflop(cfg) / 1e3, 2),
        "roofline": {"bound": "mfma", "kernel": dname, "achieved": round(achieved, 2), "peak": peak,
                     "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "note": ("achieved = ALGORITHMIC conv FLOPs / kernel time; a split kernel issues 3 16-bit "
                              "MFMA FLOPs per algorithmic FLOP, so frac tops out at 1/3 (the Winograd-form level-0 kernel of the fp16 split: 3 x 16 / 36)"
                              if args.mode in OTHER_SPLIT else "exact fp32 MFMA"),
                     "traffic": traffic, "traffic_source": traffic_src, "launches": dn,
                     "avg_launch_us": round(dms / dn * 1e3, 2), "gflop_per_launch": round(dflops / dn / 1e9, 3),
                     "conv_time_frac_of_step": round(conv_ms / (el_ev * 1e3), 3),
                     "attributed_time_frac_of_step": round(attributed_ms / (el_ev * 1e3), 3),
                     "how": "HIP events around every conv launch in a second pass of the same K steps, the parts one after the other on "
                            f"one stream ({round(el_ev / args.steps * 1e3, 4)} ms per step with the events in): a launch is {B // S} frames"},
        "kernels": kernels_block(agg, hbm_agg),
        "n_ranks_seen": dist.get_world_size() if backend_ready else 1,
        "devices": devices,
    }
    # ---- parity of the benchmarked weights: the seed-0 frame (B=1) against the CPU oracle's inverse distance
    if ref_dump is not None and os.path.isfile(ref_dump):
        ref = np.load(ref_dump)
        got = hp(torch.from_numpy(inp["feats"]).to(dev))[0].cpu().numpy()
        res["parity"] = {"max_rel": float(np.abs(got - ref).max() / np.abs(ref).max()),
                         # per-pixel: max over pixels of |d| / |ref| (inv_dist >= 0.96 everywhere): what a depth consumer sees (distance =
                         # bf / inv_dist); max_rel (the north star's bar) divides by the MAP's maximum, 192
                         "max_pixel_rel": float((np.abs(got - ref) / np.abs(ref)).max()),
                         "mean_l1_rel": float(np.abs(got - ref).mean() / np.abs(ref).mean()),
                         "bar": 1e-3, "ref": "oracle/mvsgi_oracle.py (CPU fp32 restatement pinned to the reference goldens)",
                         "frames": 1, "mode": args.mode}
        try:      # ... and at the operating point: the oracle's frame in every slot of the timed step's batch, first and last frame
            f0 = torch.from_numpy(inp["feats"]).to(dev).expand(B, -1, -1, -1, -1).contiguous()
            outs = shp.replay(f0) if use_graph else shp(f0)
            sync()
            fl = [outs[0][0][:1].cpu().numpy(), outs[-1][0][-1:].cpu().numpy()]
            res["parity"]["at_batch"] = {"frames_per_step": B, "frames_checked": [0, B - 1],
                                         "max_rel": float(max(np.abs(x - ref).max() for x in fl) / np.abs(ref).max()),
                                         "max_pixel_rel": float(max((np.abs(x - ref) / np.abs(ref)).max() for x in fl))}
            del f0, outs
        except Exception as e:
            res["parity"]["at_batch"] = {"error": f"{type(e).__name__}: {e}"}
        # the fp16 split's range report over everything this rank has run so far -- warm-up, the timed steps, the parity frames
        # (include/mvsgi.h mvsgi_saturation_flags; 0 = no clamp of the default arithmetic engaged; MVSGI_RANGE_CHECK=raise would have
        # stopped the run otherwise)
        res["parity"]["saturation_flags"] = H.saturation_flags(clear=True) | H.range_flags_seen()
    # ---- extras: single GPU only, outside the timed region above
    if world == 1 and rank == 0 and not args.no_extras:
        args.ref_dump = ref_dump
        res["extras"] = run_extras(args, cfg, B // S, hp, feats[:B // S], weights, dev, H, synth, torch, np, rng, value, shp, feats)
        # round-to-round comparable figure beside `value` (rounds 1-3 ran one part on one stream): one part of the step, one stream
        one = res["extras"].get("one_stream", {}).get("one_part") if isinstance(res["extras"].get("one_stream"), dict) else None
        if one:
            res["one_stream_one_part"] = {"frames_per_s": one["frames_per_s"], "frames_per_step": one["frames_per_step"]}
        del hp, shp, feats
        torch.cuda.empty_cache()
        cfgs = {}
        for tag, b in EXTRA_CONFIGS:
            if tag == cfg.tag:
                continue
            try:
                m = measure_path(CONFIGS[tag], b, args.mode, args.extra_steps, 3, dev, H, HotPath, synth, torch, np, rng,
                                 graph=use_graph, kernels=True, streams=S,
                                 ref_dump=config_dump_path(ref_dump, tag) if ref_dump else None)
                if "graph_replay_frames_per_s" in m:      # as the headline: parts on their own streams, one hipGraph replay per step
                    m["one_stream_eager_frames_per_s"], m["one_stream_eager_ms_per_step"] = m["frames_per_s"], m["ms_per_step"]
                    m["frames_per_s"], m["ms_per_step"] = m.pop("graph_replay_frames_per_s"), m.pop("graph_replay_ms_per_step")
                    m["frames_per_step"] = m.pop("graph_replay_frames_per_step")
                    m["path_tflops"] = round(m["frames_per_s"] * path_gflop(CONFIGS[tag]) / 1e3, 2)
                    m["submission"] = f"{S} parts of {b} frames on their own streams, one hipGraph replay per step"
                if args.mode in OTHER_SPLIT:      # the other 16-bit split of the same step (rate and error)
                    try:
                        m16 = measure_path(CONFIGS[tag], b, OTHER_SPLIT[args.mode], max(3, args.extra_steps // 2), 2, dev, H, HotPath, synth, torch, np, rng,
                                           graph=use_graph, streams=S, ref_dump=config_dump_path(ref_dump, tag) if ref_dump else None)
                        m["mode_" + OTHER_SPLIT[args.mode]] = {"frames_per_s": m16.get("graph_replay_frames_per_s", m16["frames_per_s"]),
                                           "parity": m16.get("parity"), "dominant_kernel": m16["dominant_kernel"],
                                           "dominant_tflops": m16["dominant_tflops"]}
                    except Exception as e:
                        m["mode_" + OTHER_SPLIT[args.mode]] = {"error": f"{type(e).__name__}: {e}"}
                        torch.cuda.synchronize(dev)
                    finally:
                        H.set_conv_mode(args.mode)
                cfgs[tag] = m
            except Exception as e:       # never lose the headline to an extra
                cfgs[tag] = {"error": f"{type(e).__name__}: {e}"}
        res["configs"] = cfgs
    if cpu_res is not None:
        res["cpu_baseline"] = cpu_res
    if backend_ready:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)


def run_extras(args, cfg, B, hp, feats, weights, dev, H, synth, torch, np, rng, headline_fps, shp, feats_all):
    """Measurements beside the headline (same process, same device, after the timed region).  `hp`, `feats`, `B`: ONE part of the
    headline's step (one stream); `shp`, `feats_all`: the headline's whole step."""
    B_all = feats_all.shape[0]
    use_graph = not args.eager
    from mvs_gi_amd import dropin
    from mvs_gi_amd.pipeline import HotPath
    ex = {}
    K, W = args.extra_steps, 3

    def sync():
        torch.cuda.synchronize(dev)

    def guarded(name, fn):
        try:
            ex[name] = fn()
        except Exception as e:
            ex[name] = {"error": f"{type(e).__name__}: {e}"}

    def rig_off():
        hp.cv_builder.cache_rig_constants = False
        try:
            el = timed_steps(lambda: hp(feats), sync, K, W, 1, False, dev)
        finally:
            hp.cv_builder.cache_rig_constants = True
        return {"frames_per_s": round(B * K / el, 2), "ms_per_step": round(el / K * 1e3, 4),
                "note": "grid_masks / masks re-sampled inside every step (MVSGI_RIG_CACHE=0 behaviour)"}
    guarded("rig_cache_off", rig_off)

    def nchw():
        f2 = make_feats(B, (1, cfg.num_cams, cfg.feat_chs, *cfg.feat_hw), rng, dev, torch, np, nchw=True)
        el = timed_steps(lambda: hp(f2), sync, K, W, 1, False, dev)
        return {"frames_per_s": round(B * K / el, 2), "ms_per_step": round(el / K * 1e3, 4),
                "note": "contiguous NCHW feats (the reference extractor's layout): transposed to channels-last once per step"}
    guarded("feats_nchw", nchw)

    def inv_only():
        hp.dist_regressor.return_norm_costs = False
        try:
            el = timed_steps(lambda: hp(feats), sync, K, W, 1, False, dev)
        finally:
            hp.dist_regressor.return_norm_costs = True
        return {"frames_per_s": round(B * K / el, 2), "ms_per_step": round(el / K * 1e3, 4),
                "note": "norm_costs not stored: the deployed callers discard it (spherical_sweep_stereo.py:266, api/inference_class.py); "
                        "the headline stores it, as torch_only.py:30-36 returns it"}
    guarded("inverse_distance_only", inv_only)

    def graph():
        if use_graph:      # the headline is the graph: report the per-launch submission of the same step beside it
            el = timed_steps(lambda: shp(feats_all), sync, K, W, 1, False, dev)
        else:
            shp.capture(feats_all)
            el = timed_steps(lambda: shp.replay(), sync, K, W, 1, False, dev)
        return {"frames_per_s": round(B_all * K / el, 2), "ms_per_step": round(el / K * 1e3, 4)}
    guarded("eager_launches" if use_graph else "graph_replay", graph)

    def one_stream():
        # the headline's batch, and one part of it (round 3's headline: 64 frames), each as ONE launch chain on one stream
        r = {}
        for name, f in (("whole_batch", feats_all), ("one_part", feats)):
            n = f.shape[0]
            if use_graph:
                hp.capture(f)
                el = timed_steps(lambda: hp.replay(), sync, K, W, 1, False, dev)
            else:
                el = timed_steps(lambda: hp(f), sync, K, W, 1, False, dev)
            r[name] = {"frames_per_step": n, "frames_per_s": round(n * K / el, 2), "ms_per_step": round(el / K * 1e3, 4)}
        r["note"] = "the same kernels without the second stream: what the fork / join inside the step's graph buys"
        return r
    if len(shp.parts) > 1:
        guarded("one_stream", one_stream)

    def sustained():
        # the headline's submission (the captured graph, or per-launch with --eager) for >= 3 s and >= 250 steps; the rate over
        # the LAST second, with the board's power and shader clock (hwmon) sampled by a thread of this process during the loop
        import glob
        import threading
        if use_graph and getattr(shp, "_graph", None) is None:
            shp.capture(feats_all)
        fn = shp.replay if use_graph else (lambda: shp(feats_all))
        samples, stop = [], threading.Event()

        def read_num(path, scale):
            try:
                return int(open(path).read().split()[0]) / scale
            except Exception:
                return None

        # the hwmon directory of the DRIVEN device, found through its PCI address; a box that hides that (or a torch without the
        # properties) falls back to "the card drawing the most power" over every card of the host, and the line says which it was
        dirs, card = [], None
        try:
            pr = torch.cuda.get_device_properties(dev)
            pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            dirs = glob.glob(f"/sys/bus/pci/devices/{pci}/hwmon/hwmon*/")
            card = f"/sys/bus/pci/devices/{pci}" if dirs else None
        except Exception:
            pass
        if not dirs:
            dirs = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/")
            card = "max power over /sys/class/drm/card*/ (PCI address of the device not resolvable)"

        def sampler():
            while not stop.is_set():
                best = (None, None)
                for d in dirs:
                    w = read_num(d + "power1_average", 1e6)
                    if w is None:
                        w = read_num(d + "power1_input", 1e6)
                    if w is not None and (best[0] is None or w > best[0]):
                        best = (w, read_num(d + "freq1_input", 1e6))
                samples.append((time.perf_counter(), best[0], best[1]))
                stop.wait(0.1)
        th = threading.Thread(target=sampler, daemon=True)
        sync()
        th.start()
        marks, n, t0 = [], 0, time.perf_counter()
        while True:
            for _ in range(5):
                fn()
            sync()
            n += 5
            marks.append((time.perf_counter() - t0, n))
            if marks[-1][0] >= 3.0 and n >= 250:
                break
        stop.set()
        th.join()
        t_end, n_end = marks[-1]
        t_a, n_a = min(marks, key=lambda m: abs(m[0] - (t_end - 1.0)))
        last = B_all * (n_end - n_a) / (t_end - t_a)
        first = B_all * marks[0][1] / marks[0][0]
        pw = [(t - t0, w, f) for t, w, f in samples if w is not None or f is not None]
        tail = [x for x in pw if x[0] >= t_end - 1.0]
        avg = lambda xs: round(sum(xs) / len(xs), 1) if xs else None      # noqa: E731
        return {"frames_per_s_last_second": round(last, 2), "frames_per_s_whole_run": round(B_all * n_end / t_end, 2),
                "frames_per_s_first_5_steps": round(first, 2), "seconds": round(t_end, 2), "steps": n_end,
                "vs_headline": round(last / headline_fps, 4),
                "power_W_last_second": avg([w for _, w, _ in tail if w is not None]),
                "sclk_MHz_last_second": avg([f for _, _, f in tail if f is not None]),
                "power_W_first_300ms": avg([w for t, w, _ in pw if t <= 0.3 and w is not None]),
                "samples": len(pw), "hwmon": card, "how": "same submission as the headline, sync every 5 steps; hwmon power1_average / freq1_input "
                                           "read every 100 ms by a thread of this process (None: not exposed on this box)"}
    guarded("sustained", sustained)

    def random_grids():
        # SURVEY section 8(d)'s second input variant: sampling grids U[-1.1, 1.1] (worst-case gather locality, 9 % of the taps
        # outside the image), Bernoulli grid masks / masks, at full size
        r_inp = synth.make_inputs(cfg, seed=0, batch=1, grid_kind="random")
        hp2 = HotPath(cfg, weights, r_inp, device=dev)
        for _ in range(W):
            hp2(feats)
        sub = "per-launch submission"
        fn = lambda: hp2(feats)      # noqa: E731
        if use_graph:
            try:
                hp2.capture(feats)
                fn, sub = hp2.replay, "one hipGraph replay per step"
            except Exception:
                torch.cuda.synchronize(dev)
        el = timed_steps(fn, sync, K, W, 1, False, dev)
        with ConvProbe(H) as probe:
            probe.enabled = True
            hp2(feats)
            sync()
            probe.enabled = False
            hb = probe.hbm_summary()
        sw = {k: {"us": round(v[2] / v[0] * 1e3, 1), "GBps": round(v[1] / (v[2] * 1e-3) / 1e9, 1)} for k, v in hb.items() if "sweep" in k}
        del hp2
        return {"frames_per_s": round(B * K / el, 2), "ms_per_step": round(el / K * 1e3, 4), "submission": sub, "sweep_launches": sw,
                "note": "grids U[-1.1, 1.1] (no locality between neighbouring voxels), Bernoulli(0.9) grid masks, Bernoulli(0.95) masks"}
    guarded("random_grids", random_grids)

    def strict_interface():
        # the reference's interface taken literally, all at once: contiguous NCHW feats (the reference extractor's layout), the
        # rig's grid_masks / masks re-sampled inside every step (no constants lowered ahead of time), norm_costs stored
        f2 = make_feats(B, (1, cfg.num_cams, cfg.feat_chs, *cfg.feat_hw), rng, dev, torch, np, nchw=True)
        hp.cv_builder.cache_rig_constants = False
        hp.dist_regressor.return_norm_costs = True
        try:
            el = timed_steps(lambda: hp(f2), sync, K, W, 1, False, dev)
        finally:
            hp.cv_builder.cache_rig_constants = True
        return {"frames_per_s": round(B * K / el, 2), "ms_per_step": round(el / K * 1e3, 4), "vs_headline": round(B * K / el / headline_fps, 4),
                "submission": "per-launch submission",
                "note": "NCHW feats + rig cache off + norm_costs stored, together (each alone: extras.feats_nchw, extras.rig_cache_off)"}
    guarded("strict_interface", strict_interface)

    def b1():
        r = measure_path(cfg, 1, args.mode, 200, 20, dev, H, HotPath, synth, torch, np, rng, graph=True)
        return {"eager_ms": r["ms_per_step"], "graph_replay_ms": r["graph_replay_ms_per_step"],
                "eager_frames_per_s": r["frames_per_s"], "graph_replay_frames_per_s": r["graph_replay_frames_per_s"],
                "dominant_kernel": r["dominant_kernel"], "frac": r["frac"]}
    guarded("latency_b1", b1)

    def batch_sweep():
        # B in {1, 4, 8, 16} frames per GPU per step beside the headline's batch (B = 1 is latency_b1)
        r = {}
        # frames_per_s / ms_per_step: one hipGraph replay per step (as the headline); eager_*: per-launch submission, which
        # costs 20 % at 4 frames per step
        for b in (4, 8, 16):
            m = measure_path(cfg, b, args.mode, K, W, dev, H, HotPath, synth, torch, np, rng, graph=True)
            r[str(b)] = {"frames_per_s": m["graph_replay_frames_per_s"], "ms_per_step": m["graph_replay_ms_per_step"],
                         "eager_frames_per_s": m["frames_per_s"], "eager_ms_per_step": m["ms_per_step"],
                         "dominant_kernel": m["dominant_kernel"], "frac": m["frac"]}
        return r
    guarded("batch_sweep", batch_sweep)

    def mode_f32():
        # the exact-arithmetic companion of the split-bf16 headline: every conv on v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32
        # fmaf chain), frac against the 157.3 TFLOP/s fp32-MFMA peak
        if args.mode == "f32":
            return {"note": "the headline is the exact-fp32 mode"}
        try:
            m = measure_path(cfg, min(B, 32), "f32", max(3, K // 4), 2, dev, H, HotPath, synth, torch, np, rng)
        finally:
            H.set_conv_mode(args.mode)
        m["peak_tflops"] = PEAK_TFLOPS["f32"]
        return m
    guarded("mode_f32", mode_f32)

    def mode_other_split():
        # the other 16-bit split of the same path.  fp16 (hi = fp16(x), lo = fp16(x - hi): 11 + 11 bits per operand, saturating at
        # +-65504) is the library's default; bf16 (8 + 8 bits, fp32's range) is the deployer's answer for activations beyond fp16's
        # range (Precision modes) -- at the same matrix rate, without the Winograd form of the level-0 convs
        if args.mode not in OTHER_SPLIT:
            return {"note": "the headline is the exact-fp32 mode"}
        try:
            m = measure_path(cfg, B, OTHER_SPLIT[args.mode], max(3, K // 2), 2, dev, H, HotPath, synth, torch, np, rng, graph=use_graph,
                             streams=len(shp.parts), ref_dump=getattr(args, "ref_dump", None))
        finally:
            H.set_conv_mode(args.mode)
        if "graph_replay_frames_per_s" in m:
            m["frames_per_s_one_part_eager"] = m["frames_per_s"]
            m["frames_per_s"] = m["graph_replay_frames_per_s"]
        m["vs_headline"] = round(m["frames_per_s"] / headline_fps, 4)
        return m
    guarded("mode_" + OTHER_SPLIT.get(args.mode, "f16x3"), mode_other_split)

    def precision_check():
        # the deployer's per-checkpoint measurement (HotPath.precision_check) on the benchmark's own weights and one synthetic frame:
        # which arithmetic this checkpoint needs for inv_dist within half the north star's bar
        try:
            return hp.precision_check(feats[:1].contiguous())
        finally:
            H.set_conv_mode(args.mode)
    guarded("precision_check", precision_check)

    # ---- images -> inverse distance (HIP feature extractor in front), HBM-resident and host-fed
    Hi, Wi = cfg.feat_hw
    N = cfg.num_cams

    def make_extractor():
        """The random-init extractor of synth.make_extractor_weights, its LAST layer's BatchNorm rescaled so that the features have
        unit standard deviation on uniform random images (LeakyReLU is positively homogeneous: the features scale with it).  As
        initialised their std is ~45 and the regulator's activations behind them reach 7.5e4: beyond the fp16 split's range, which
        the range report (round 6) turns into an error -- rounds 2-5 measured this chain on silently clamped activations.  The
        hot path's own synthetic features are N(0, 1)."""
        fe = dropin.SimpleFeatExtraction(in_size=(4 * Hi, 4 * Wi), in_chs=3, chs=cfg.feat_chs, k_sz=3, layers=[5, 10])
        fe.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_extractor_weights(0).items()}, strict=True)
        fe = fe.eval().to(dev)
        with torch.no_grad():
            probe = torch.from_numpy(np.random.default_rng(7).integers(0, 256, (N, 4 * Hi, 4 * Wi, 3), dtype=np.uint8)).to(dev)
            sd = float(fe(probe).std())
            for m in fe.final_layer.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.mul_(1.0 / sd)
                    m.bias.mul_(1.0 / sd)
        return fe

    def e2e():
        fe = make_extractor()
        imgs = torch.from_numpy(rng.integers(0, 256, (B * N, 4 * Hi, 4 * Wi, 3), dtype=np.uint8)).to(dev)

        def estep():
            with torch.no_grad():
                f = fe(imgs)
            hp(f.reshape(B, N, *f.shape[1:]))

        def fstep():
            with torch.no_grad():
                fe(imgs)
        eel_eager = timed_steps(estep, sync, K, W, 1, False, dev)
        fel = timed_steps(fstep, sync, K, W, 1, False, dev)
        # as the headline: the whole chain (33 extractor launches + the path's 45) as one hipGraph replay per step
        eel, sub = eel_eager, "per-launch submission"
        if use_graph:
            try:
                ge = torch.cuda.CUDAGraph()
                side0 = torch.cuda.Stream(device=dev)
                side0.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side0):
                    estep()
                torch.cuda.current_stream(dev).wait_stream(side0)
                with torch.cuda.graph(ge, capture_error_mode="thread_local"):
                    estep()
                eel, sub = timed_steps(ge.replay, sync, K, W, 1, False, dev), "one hipGraph replay per step"
                del ge
            except Exception as e:
                sub = f"per-launch submission (graph capture failed: {type(e).__name__})"
                torch.cuda.synchronize(dev)
        r = {"frames_per_s": round(B * K / eel, 2), "ms_per_step": round(eel / K * 1e3, 4), "submission": sub,
             "eager_frames_per_s": round(B * K / eel_eager, 2),
             "feature_extractor_ms_per_step": round(fel / K * 1e3, 4),
             "feature_extractor_tflops": round(B * 58.06 / (fel / K) / 1e3, 2), "input": "uint8 HWC images resident in HBM"}
        # B = 1 latency of the whole chain
        img1 = imgs[:N].contiguous()

        def e1():
            with torch.no_grad():
                f = fe(img1)
            hp(f.reshape(1, N, *f.shape[1:]))
        e1l = timed_steps(e1, sync, 100, 10, 1, False, dev)
        r["b1_ms"] = round(e1l / 100 * 1e3, 4)
        # the same chain as ONE hipGraph replay (InferencePipeline.capture / replay do this for a deployment)
        g1 = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            e1()
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.cuda.graph(g1, capture_error_mode="thread_local"):
            e1()
        g1l = timed_steps(g1.replay, sync, 100, 10, 1, False, dev)
        r["b1_graph_replay_ms"] = round(g1l / 100 * 1e3, 4)
        del g1
        # host feed: uint8 frames in pinned memory, double-buffered H2D on a side stream overlapped with compute
        nbuf = 3
        host = [torch.from_numpy(rng.integers(0, 256, (B * N, 4 * Hi, 4 * Wi, 3), dtype=np.uint8)).pin_memory()
                for _ in range(nbuf)]
        devb = [torch.empty_like(imgs) for _ in range(2)]
        # a stream whose copies overlap the compute stream's kernels: HIP multiplexes streams onto four hardware queues, and a copy
        # stream on the compute stream's queue serialises behind it (0.66 of the resident rate instead of 0.975: what rounds 5 and 6
        # first measured as "0.68" and round 4, on a luckier stream, as "0.996")
        from mvs_gi_amd.pipeline import overlapping_copy_stream
        copy_s = overlapping_copy_stream(dev)
        ready = [torch.cuda.Event() for _ in range(2)]
        freed = [torch.cuda.Event() for _ in range(2)]
        main_s = torch.cuda.current_stream(dev)
        state = {"i": 0}

        def upload(slot, k):
            t_ = time.perf_counter()
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(freed[slot])
                devb[slot].copy_(host[k % nbuf], non_blocking=True)
                ready[slot].record(copy_s)
            state["upload_cpu_s"] = state.get("upload_cpu_s", 0.0) + time.perf_counter() - t_
            state["uploads"] = state.get("uploads", 0) + 1

        for s_ in range(2):
            freed[s_].record(main_s)
        upload(0, 0)

        def hstep():
            i = state["i"]
            slot = i & 1
            upload(slot ^ 1, i + 1)                  # next batch crosses PCIe while this one is computed
            main_s.wait_event(ready[slot])
            ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ca.record(main_s)
            with torch.no_grad():
                f = fe(devb[slot])
            hp(f.reshape(B, N, *f.shape[1:]))
            cb.record(main_s)
            state["compute_events"] = (ca, cb)       # the last step's compute span on the device (read behind the final sync)
            freed[slot].record(main_s)
            state["i"] = i + 1
        hel = timed_steps(hstep, sync, K, W, 1, False, dev)
        bytes_per_frame = N * 4 * Hi * 4 * Wi * 3
        r["host_feed"] = {"frames_per_s": round(B * K / hel, 2), "ms_per_step": round(hel / K * 1e3, 4),
                          "vs_resident": round((B * K / hel) / (B * K / eel_eager), 4),
                          "bytes_per_frame": bytes_per_frame,
                          "h2d_GBps": round(B * K * bytes_per_frame / hel / 1e9, 2),
                          "upload_call_cpu_ms": round(state["upload_cpu_s"] / state["uploads"] * 1e3, 3),      # host time inside one enqueue of the copy
                          "compute_span_ms_last_step": round(state["compute_events"][0].elapsed_time(state["compute_events"][1]), 3),
                          "how": "uint8 HWC frames in pinned host memory, two device buffers, copies on a side stream that does not share the "
                                 "compute stream's hardware queue (pipeline.overlapping_copy_stream), overlapped with the previous batch's compute"}
        return r
    guarded("images_to_inverse_distance", e2e)
    return ex


if __name__ == "__main__":
    main()
