#!/usr/bin/env python3
"""A/B of environment toggles through bench.py on one box: python tools/ab_bench.py MVSGI_RS=0 MVSGI_RS=1 [--rounds 2]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
arms = [a for a in sys.argv[1:] if "=" in a]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
res = {a: [] for a in arms}
for r in range(rounds):
    for a in arms:
        env = dict(os.environ)
        env["MVSGI_EXPERIMENTAL"] = "1"      # the arms may name gated experiment switches (mvs_gi_amd/hip_ops.py exp_env)
        for kv in a.split(","):
            k, v = kv.split("=")
            env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline"] + extra,
                             capture_output=True, text=True, env=env).stdout
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        res[a].append(d)
names = sorted({k for a in arms for d in res[a] for k in d["kernels"]})
print(f"{'arm':28s} " + " ".join(f"{a:>14s}" for a in arms))
print(f"{'frames/s (median)':28s} " + " ".join(f"{sorted(d['value'] for d in res[a])[len(res[a]) // 2]:14.1f}" for a in arms))
print(f"{'ms/step':28s} " + " ".join(f"{sorted(d['ms_per_step'] for d in res[a])[len(res[a]) // 2]:14.3f}" for a in arms))
for k in names:
    row = []
    for a in arms:
        v = [d["kernels"][k] for d in res[a] if k in d["kernels"]]
        row.append(f"{v[0]['launches'] // 100:3d}x{sorted(x['avg_us'] for x in v)[len(v) // 2]:9.1f}" if v else f"{'-':>13s}")
    short = k.replace("conv3d_bf16x3_kernel", "b3").replace(", false", ",f").replace(", true", ",t")
    print(f"{short[:28]:28s} " + " ".join(f"{x:>14s}" for x in row))
for a in arms:
    d = res[a][-1]
    tot = sum(v["launches"] * v["avg_us"] for v in d["kernels"].values()) / d["steps"]
    print(f"{a}: conv kernels {tot:.0f} us per step of {d['ms_per_step'] * 1e3:.0f}")
