#!/usr/bin/env python3
"""images -> inverse distance fed from pinned host memory (bench.py extras ... host_feed): why does the double-buffered upload not
hide behind the compute (round 5 / 6: 0.68 of the resident rate, although the link copies 56 GB/s = 10.8 ms per 64-frame batch against
27.6 ms of compute)?  Variants of the upload: as bench.py (one copy per batch on a side stream), a high-priority copy stream, the
batch in 8 chunks, copy alone, compute alone.  `python tools/host_feed_probe.py [B]`"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvs_gi_amd import dropin, hip_ops as H, synth  # noqa: E402
from mvs_gi_amd.configs import CONFIGS  # noqa: E402
from mvs_gi_amd.pipeline import HotPath  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
cfg = CONFIGS["G16V"]
inp = synth.make_inputs(cfg, seed=0, batch=1)
if "--like-bench" in sys.argv:       # bench.py's state when its host-fed loop starts: a two-part StreamedHotPath captured and replayed
    from mvs_gi_amd.pipeline import StreamedHotPath
    shp = StreamedHotPath(cfg, synth.make_weights(cfg, seed=0), inp, device=dev, n_streams=2)
    fa = torch.randn((2 * B, cfg.num_cams, *cfg.feat_hw, cfg.feat_chs), device=dev).permute(0, 1, 4, 2, 3)
    shp.capture(fa)
    for _ in range(5):
        shp.replay()
    torch.cuda.synchronize()
    hp = shp.parts[0]
else:
    hp = HotPath(cfg, synth.make_weights(cfg, seed=0), inp, device=dev)
    hp.dist_regressor.return_norm_costs = False
N, (Hi, Wi) = cfg.num_cams, cfg.feat_hw
fe = dropin.SimpleFeatExtraction(in_size=(4 * Hi, 4 * Wi), in_chs=3, chs=cfg.feat_chs, k_sz=3, layers=[5, 10])
fe.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_extractor_weights(0).items()}, strict=True)
fe = fe.eval().to(dev)
rng = np.random.default_rng(0)
with torch.no_grad():
    probe = torch.from_numpy(rng.integers(0, 256, (N, 4 * Hi, 4 * Wi, 3), dtype=np.uint8)).to(dev)
    sd = float(fe(probe).std())
    for m in fe.final_layer.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.mul_(1.0 / sd)
            m.bias.mul_(1.0 / sd)
host = [torch.from_numpy(rng.integers(0, 256, (B * N, 4 * Hi, 4 * Wi, 3), dtype=np.uint8)).pin_memory() for _ in range(3)]
devb = [torch.empty_like(host[0], device=dev) for _ in range(2)]
main_s = torch.cuda.current_stream(dev)
K, W = 12, 3


def compute(slot):
    with torch.no_grad():
        f = fe(devb[slot])
    hp(f.reshape(B, N, *f.shape[1:]))


def run(label, copy_s, chunks, do_copy=True, do_compute=True):
    ready = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]
    for s_ in range(2):
        freed[s_].record(main_s)

    cpu_in_upload = [0.0]

    def upload(slot, k):
        t_ = time.perf_counter()
        with torch.cuda.stream(copy_s):
            copy_s.wait_event(freed[slot])
            if do_copy:
                n = host[0].shape[0]
                for c in range(chunks):
                    a, b = c * n // chunks, (c + 1) * n // chunks
                    devb[slot][a:b].copy_(host[k % 3][a:b], non_blocking=True)
            ready[slot].record(copy_s)
        cpu_in_upload[0] += time.perf_counter() - t_
    upload(0, 0)
    times = []
    for i in range(K + W):
        if i == W:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        slot = i & 1
        upload(slot ^ 1, i + 1)
        main_s.wait_event(ready[slot])
        if do_compute:
            compute(slot)
        freed[slot].record(main_s)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K
    print(f"{label:58s} {el * 1e3:7.2f} ms per {B}-frame batch = {B / el:7.1f} frames/s   (CPU inside upload(): {cpu_in_upload[0] / (K + W + 1) * 1e3:6.2f} ms per call)", flush=True)


plain = torch.cuda.Stream(device=dev)
hi = torch.cuda.Stream(device=dev, priority=-1)
if "--after-graph" in sys.argv:      # as bench.py's extras: a hipGraph of the chain was captured and replayed before the host-fed loop
    devb[0].copy_(host[0])
    compute(0)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        compute(0)
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        compute(0)
    g.replay()
    torch.cuda.synchronize()
    if "--keep-graph" not in sys.argv:
        del g
    print("(a hipGraph of the chain was captured and replayed first)")
if "--queues" in sys.argv:
    # HIP multiplexes its streams onto a few hardware queues: which of 12 consecutively created streams overlap their copies with the
    # compute stream's kernels?
    run("compute alone (resident input)", plain, 1, do_copy=False)
    for k in range(12):
        run(f"copy stream #{k} (created in this order)", torch.cuda.Stream(device=dev), 1)
    sys.exit(0)
run("compute alone (resident input)", plain, 1, do_copy=False)
run("copy alone, one copy per batch", plain, 1, do_compute=False)
run("as bench.py: one copy per batch on a side stream", plain, 1)
run("high-priority copy stream", hi, 1)
run("8 chunks per batch, side stream", plain, 8)
run("8 chunks per batch, high-priority stream", hi, 8)
run("64 chunks per batch, side stream", plain, 64)
