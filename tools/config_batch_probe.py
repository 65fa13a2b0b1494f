#!/usr/bin/env python3
"""Frames/s of a configuration as the bench submits it (parts on their own HIP streams inside one hipGraph) over part sizes and stream
counts:  python tools/config_batch_probe.py <tag> "<frames per part> ..." "<streams> ..." [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import torch                # noqa: E402
import bench                # noqa: E402
from mvs_gi_amd import hip_ops as H, synth      # noqa: E402
from mvs_gi_amd.configs import CONFIGS          # noqa: E402
from mvs_gi_amd.pipeline import HotPath         # noqa: E402

tag = sys.argv[1]
parts = [int(v) for v in sys.argv[2].split()]
streams = [int(v) for v in sys.argv[3].split()]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for b in parts:
    for s in streams:
        m = bench.measure_path(CONFIGS[tag], b, "bf16x3", steps, 3, dev, H, HotPath, synth, torch, np, rng, graph=True, streams=s)
        print(f"{tag}: {s} x {b} frames: {m.get('graph_replay_frames_per_s')} frames/s as one graph replay ({m.get('graph_replay_ms_per_step')} ms), "
              f"one part eager {m['frames_per_s']}", flush=True)
