#!/usr/bin/env python3
"""Whole-extractor timing (SimpleFeatExtraction, G16V recipe) for rocprofv3: tools/extractor_probe.py [frames] [iters]."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd.dropin.feature_extractor import SimpleFeatExtraction

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
net = SimpleFeatExtraction(in_size=(512, 2048), in_chs=3, chs=16, layers=[5, 10]).cuda().eval()
imgs = torch.randint(0, 256, (frames * 3, 512, 2048, 3), dtype=torch.uint8, device="cuda")
chunk = int(os.environ.get("EXT_CHUNK", "0"))      # images per pass through the stack (0: all at once): does a chunk's ping-pong
#                                                   # pair of activation buffers stay in the 256 MB memory-side cache?


def run():
    if chunk <= 0:
        return net(imgs)
    return torch.cat([net(imgs[i:i + chunk]) for i in range(0, imgs.shape[0], chunk)], 0)


with torch.no_grad():
    for _ in range(2):
        y = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        y = run()
    torch.cuda.synchronize()
print(f"{frames} frames{f' in chunks of {chunk} images' if chunk else ''}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per pass, out {tuple(y.shape)}")
