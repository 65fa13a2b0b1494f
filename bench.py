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
                  against the gfx950 fp32-MFMA peak;
  "cpu_baseline": the CPU oracle (torch-CPU restatement of the reference path) timed on this
                  host's cores on a bounded sample (rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0}   # MI355X_MICROARCH.md: dense fp32-MFMA / bf16-MFMA peaks
DTYPE = {"f32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA operands, f32 accumulate/activations)"}
PEAK_HBM_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="G16V")
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step")
    ap.add_argument("--mode", default="bf16x3", choices=["bf16x3", "f32"],
                    help="conv arithmetic: split-bf16 MFMA (3 bf16 MFMAs per product, fp32 accumulate) or exact fp32 MFMA")
    ap.add_argument("--graph", action="store_true",
                    help="also time hipGraph replays of the same step (reported as graph_replay_*; the headline "
                         "value and the per-kernel events always come from the eager launches)")
    ap.add_argument("--end-to-end", action="store_true",
                    help="also time imgs -> inv_dist with the HIP feature extractor in front (SURVEY 8(f) rank 1); "
                         "reported as end_to_end_*; the headline metric stays the plane-sweep hot path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl=RCCL)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------
# distributed scaffolding (also exercised on CPU with gloo by tests/test_bench_sharding.py)
# ------------------------------------------------------------------------------------------
def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def frame_shard(total_frames: int, world: int, rank: int):
    """Contiguous, balanced [lo, hi) shard of a frame sequence (strong-scaling helper for
    dataset playback; the bench itself is weak-scaled: every rank owns `batch` frames)."""
    base, rem = divmod(total_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def timed_steps(step_fn, sync_fn, steps: int, warmup: int, world: int, backend_ready: bool, device=None):
    """W warmup steps, barrier+sync, exactly K timed steps, sync+barrier; returns the MAX
    elapsed seconds over ranks."""
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
    if backend_ready:
        dist.barrier()
    el = time.perf_counter() - t0
    if backend_ready:
        on_cpu = device is None or dist.get_backend() == "gloo"
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if on_cpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el


# ------------------------------------------------------------------------------------------
# per-kernel attribution: HIP events around every conv launch of the timed region
# ------------------------------------------------------------------------------------------
class ConvProbe:
    """Wraps hip_ops.conv3d: records a HIP event pair (on the launch stream) around each
    call and the algorithmic FLOPs and kernel name of the launch."""

    def __init__(self, H):
        import torch
        self.H, self.torch = H, torch
        self.orig = H.conv3d
        self.orig_up2 = H.conv3d_up2
        self.records = []
        self.enabled = False

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
            self.records.append((H.conv3d_variant(B, Cin, D, Hh, W, Cout, stride, impl), 2.0 * 27 * Cin * Cout * vox, s, e))
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
            self.records.append((H.conv3d_up2_variant(B, Cin, Dl, Hl, Wl, Cout, w_layout), 2.0 * 27 * Cin * Cout * vox, s, e))
            return y

        H.conv3d = probed
        H.conv3d_up2 = probed_up2
        return self

    def __exit__(self, *a):
        self.H.conv3d = self.orig
        self.H.conv3d_up2 = self.orig_up2

    def summary(self):
        agg = {}
        for name, flops, s, e in self.records:
            ms = s.elapsed_time(e)
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += ms
        return agg


def read_pmc_traffic(kernel_name: str):
    """HBM bytes per launch of `kernel_name` from a committed rocprofv3 --pmc summary
    (profiles/pmc_traffic.json, written by tools/summarize_rocprof.py), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
        return d.get(kernel_name, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


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


def cpu_baseline_subprocess(args):
    """Run the CPU baseline in a child process, before this process touches the GPU, with a
    hard timeout so a slow host can never stall the benchmark."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--config", args.config,
           "--cpu-seconds", str(args.cpu_seconds)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_seconds * 6 + 90)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:   # report, never fail the GPU measurement
        return {"value": None, "unit": "frames/s", "cores": effective_cores(), "kind": "port",
                "sample": f"CPU baseline did not finish: {type(e).__name__}"}


def cpu_baseline(cfg, seconds: float):
    """The oracle (CPU restatement of the reference's PyTorch path) on this host's cores:
    B=1 frames of the same workload until ~`seconds` have elapsed (at least 3 frames)."""
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
    one()
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if (el >= seconds and n >= 3) or n >= 200:
            break
    return {"value": round(n / el, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} single-frame (B=1) {cfg.tag} passes of oracle/mvsgi_oracle.py (torch {torch.__version__} "
                      f"CPU, fp32) in {el:.1f} s after 1 warm-up"}


def main(argv=None):
    args = parse_args(argv)
    rank, local_rank, world = dist_env()
    if args.cpu_baseline_only:
        from mvs_gi_amd.configs import CONFIGS as _C
        print(json.dumps(cpu_baseline(_C[args.config], args.cpu_seconds)), flush=True)
        return
    cpu_res = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_res = cpu_baseline_subprocess(args)      # before any GPU initialisation
    if not os.path.isfile(os.path.join(ROOT, "mvs_gi_amd", "libmvsgi_hip.so")):
        # the library normally travels with the tree (built by __graft_entry__.build()); build it rather than fail
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            for _ in range(600):
                if os.path.isfile(os.path.join(ROOT, "mvs_gi_amd", "libmvsgi_hip.so")):
                    break
                time.sleep(0.5)
    import numpy as np
    import torch
    import torch.distributed as dist
    from mvs_gi_amd import hip_ops as H, synth
    from mvs_gi_amd.configs import CONFIGS, path_gflop
    from mvs_gi_amd.pipeline import HotPath

    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                             f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    if os.environ.get("MVSGI_BENCH_SHARE_GPU"):      # test hook: N ranks on one device (use --backend gloo)
        local_rank = 0
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

    cfg = CONFIGS[args.config]
    B = args.batch
    H.set_conv_mode(args.mode)
    peak = PEAK_TFLOPS[args.mode]
    inp = synth.make_inputs(cfg, seed=0, batch=1)
    hp = HotPath(cfg, synth.make_weights(cfg, seed=0), inp, device=dev)
    rng = np.random.default_rng(1000 + rank)     # every rank owns different frames
    feats = torch.from_numpy(rng.standard_normal((B, *inp["feats"].shape[1:]), dtype=np.float32)).to(dev)
    out = {}

    def step():
        out["inv"], out["pr"] = hp(feats)

    def sync():
        torch.cuda.synchronize(dev)

    with ConvProbe(H) as probe:
        for _ in range(args.warmup):
            step()
        sync()
        probe.enabled = True
        el = timed_steps(step, sync, args.steps, 0, world, backend_ready, dev)
        probe.enabled = False
        sync()
        agg = probe.summary()

    assert torch.isfinite(out["inv"]).all()
    graph_res = None
    if args.graph:
        hp.capture(feats)

        def gstep():
            hp.replay()
        gel = timed_steps(gstep, sync, args.steps, args.warmup, world, backend_ready, dev)
        graph_res = {"graph_replay_frames_per_s": round(B * world * args.steps / gel, 2),
                     "graph_replay_ms_per_step": round(gel / args.steps * 1e3, 4)}
    e2e_res = None
    if args.end_to_end:
        from mvs_gi_amd import dropin
        Hi, Wi = cfg.feat_hw
        fe = dropin.SimpleFeatExtraction(in_size=(4 * Hi, 4 * Wi), in_chs=3, chs=cfg.feat_chs, k_sz=3, layers=[5, 10])
        fe.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_extractor_weights(0).items()}, strict=True)
        fe = fe.eval().to(dev)
        imgs = torch.from_numpy(rng.random((B * cfg.num_cams, 3, 4 * Hi, 4 * Wi), dtype=np.float32)).to(dev)

        def estep():
            with torch.no_grad():
                f = fe(imgs)
            out["inv"], out["pr"] = hp(f.reshape(B, cfg.num_cams, *f.shape[1:]))

        def fstep():
            with torch.no_grad():
                out["f"] = fe(imgs)
        eel = timed_steps(estep, sync, args.steps, args.warmup, world, backend_ready, dev)
        fel = timed_steps(fstep, sync, args.steps, args.warmup, world, backend_ready, dev)
        e2e_res = {"end_to_end_frames_per_s": round(B * world * args.steps / eel, 2),
                   "end_to_end_ms_per_step": round(eel / args.steps * 1e3, 4),
                   "feature_extractor_ms_per_step": round(fel / args.steps * 1e3, 4),
                   "feature_extractor_tflops": round(B * 58.06 / (fel / args.steps) / 1e3, 2)}
    frames = B * world * args.steps
    value = frames / el
    # dominant kernel = largest total time among the conv variants
    dom = max(agg.items(), key=lambda kv: kv[1][2])
    dname, (dn, dflops, dms) = dom
    achieved = dflops / (dms * 1e-3) / 1e12
    conv_ms = sum(v[2] for v in agg.values())
    res = {
        "metric": "stereo frames/sec/GPU (G16V, 3-cam, D=16) + inv-dist L1 vs reference",
        "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE[args.mode], "data": "synthetic",
        "config": {"workload": f"{cfg.tag}: {cfg.num_cams} cams, D={cfg.num_cands}, builder={cfg.builder}, "
                               f"regulator=({cfg.reg_in_chs},{cfg.reg_f_int_chs}), feats {cfg.feat_hw}, cv {cfg.cv_hw}",
                   "frames_per_gpu_per_step": B, "parallelism": f"frame-sharded x{world} (no collective)",
                   "rig_constants": "grids / grid_masks / masks resident in HBM; validity byte and packed weights lowered "
                                    "once during warm-up (DESIGN.md section 1)",
                   "path_gflop_per_frame": round(path_gflop(cfg), 2)},
        "frames_per_sec_per_gpu": round(value / world, 2),
        "path_tflops": round(value * path_gflop(cfg) / 1e3, 2),
        "roofline": {"bound": "mfma", "kernel": dname, "achieved": round(achieved, 2), "peak": peak,
                     "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "note": ("achieved = ALGORITHMIC conv FLOPs / kernel time; the split-bf16 kernel issues 3 bf16 "
                              "MFMA FLOPs per algorithmic FLOP, so frac tops out at 1/3" if args.mode == "bf16x3"
                              else "exact fp32 MFMA"),
                     "traffic": read_pmc_traffic(dname), "launches": dn,
                     "avg_launch_us": round(dms / dn * 1e3, 2), "gflop_per_launch": round(dflops / dn / 1e9, 3),
                     "conv_time_frac_of_step": round(conv_ms / (el * 1e3), 3)},
        "kernels": {k: {"launches": v[0], "avg_us": round(v[2] / v[0] * 1e3, 2),
                        "tflops": round(v[1] / (v[2] * 1e-3) / 1e12, 2)} for k, v in sorted(agg.items())},
    }
    if graph_res is not None:
        res.update(graph_res)
    if e2e_res is not None:
        res.update(e2e_res)
    if cpu_res is not None:
        res["cpu_baseline"] = cpu_res
    if backend_ready:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
