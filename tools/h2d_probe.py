#!/usr/bin/env python3
"""Host -> device copy rate of this box from pinned memory (what feeds the images -> inverse-distance chain from the host: bench.py
extras.images_to_inverse_distance.host_feed): one and two streams, several sizes.  `python tools/h2d_probe.py`"""
import torch

dev = "cuda:0"
for mb in (9, 64, 600):
    n = mb << 20
    for streams in (1, 2):
        host = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(streams)]
        dst = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(streams)]
        st = [torch.cuda.Stream(device=dev) for _ in range(streams)]
        reps = max(4, (2 << 30) // n // streams)
        for _ in range(2):
            for i in range(streams):
                with torch.cuda.stream(st[i]):
                    dst[i].copy_(host[i], non_blocking=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in st:
            s.wait_stream(torch.cuda.current_stream())
        for _ in range(reps):
            for i in range(streams):
                with torch.cuda.stream(st[i]):
                    dst[i].copy_(host[i], non_blocking=True)
        for s in st:
            torch.cuda.current_stream().wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"{mb:4d} MiB x {streams} stream(s): {reps * streams * n / ms / 1e6:6.2f} GB/s", flush=True)
