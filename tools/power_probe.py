#!/usr/bin/env python3
"""Average board power, clock and energy per launch of one conv kernel run back to back for a few seconds
(the chip runs these kernels at its power cap, so energy per launch -- not cycles -- is what sets the step time).
python tools/power_probe.py [streaming|rs|idle|step] [seconds]
`step` runs the whole G16V hot path (B = 64).  Besides hwmon (sampled every 50 ms) a second thread calls `rocm-smi --showpower
--showclocks` WHILE the loop runs (round-2 review: the rocm-smi reading used to be taken after the loop, on an idle chip)."""
import glob, os, subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H

which = sys.argv[1] if len(sys.argv) > 1 else "streaming"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
dev = "cuda:0"
B, d, h, w = 32, 8, 40, 160
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, 32), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((32, 32, 3, 3, 3)) / np.sqrt(27 * 32)).astype(np.float32)).to(dev)
wp, wpr = H.pack_conv_weights_bf16x3(wt), H.pack_conv_weights_rs(wt)
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
xs = H.act_to_split(x)
ys = H.SplitAct(B, d, h, w, 32, dev)
yo = torch.empty_like(x)


def hw():
    out = {}
    for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        try:
            out["W"] = max(out.get("W", 0), int(open(p).read()) / 1e6)
        except Exception:
            pass
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            cur = [l for l in open(p).read().splitlines() if l.endswith("*")]
            if cur:
                out["sclk"] = cur[0]
        except Exception:
            pass
    return out


samples, smi, stop = [], [], False


def sampler():
    while not stop:
        samples.append(hw())
        time.sleep(0.05)


def smi_sampler():
    import re
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20)
            pw = re.findall(r"Power \(W\):\s*([0-9.]+)", r.stdout)
            sc = re.findall(r"sclk clock level:\s*\d+:?\s*\(?([0-9]+)Mhz", r.stdout, flags=re.I)
            smi.append((time.perf_counter(), [float(v) for v in pw][:1], [int(v) for v in sc][:1]))
        except Exception as e:
            smi.append((time.perf_counter(), str(e), None))
        time.sleep(0.3)


if which == "step":
    from mvs_gi_amd import synth
    from mvs_gi_amd.configs import CONFIGS
    from mvs_gi_amd.pipeline import HotPath
    import bench
    cfg = CONFIGS["G16V"]
    H.set_conv_mode("bf16x3")
    inp = synth.make_inputs(cfg, seed=0, batch=1)
    hp = HotPath(cfg, synth.make_weights(cfg, seed=0), inp, device=dev)
    feats64 = bench.make_feats(64, inp["feats"].shape, rng, torch.device(dev), torch, np)
fn = {"step": (lambda: hp(feats64)) if which == "step" else None, "streaming": lambda: H.conv3d(x, wt, wp, sc, sh, res=x, impl=H.CONV_BF16X3, out=yo),
      "rs": lambda: H.conv3d_rs(xs, wpr, sc, sh, res=xs, out=ys),
      "idle": lambda: time.sleep(0.001)}[which]
for _ in range(5):
    fn()
torch.cuda.synchronize()
th = threading.Thread(target=sampler)
th.start()
th2 = threading.Thread(target=smi_sampler)
th2.start()
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < secs:
    reps = 5 if which == "step" else 50
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    n += reps
el = time.perf_counter() - t0
stop = True
th.join()
th2.join()
print("rocm-smi DURING the loop (t since start [s], package power [W], sclk [MHz]):",
      [(round(t - t0, 1), p, c) for t, p, c in smi])
ws = [s["W"] for s in samples[len(samples) // 3:] if "W" in s]
print(which, f"{el / n * 1e6:.1f} us per launch;", f"power samples {len(ws)}: mean {np.mean(ws) if ws else float('nan'):.0f} W max {max(ws) if ws else 0:.0f} W;",
      f"energy per launch {np.mean(ws) * el / n * 1e3 if ws else float('nan'):.1f} mJ;", samples[-1] if samples else None)
try:
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20)
    print(r.stdout[-600:])
except Exception as e:
    print("rocm-smi:", e)
