#!/usr/bin/env python3
"""Condense a rocprofv3 output directory into small text/JSON summaries for profiles/.

  python tools/summarize_rocprof.py stats <dir> <out.txt>        # --kernel-trace --stats run
  python tools/summarize_rocprof.py pmc   <dir> <out.json> ...   # --pmc runs (FETCH_SIZE / WRITE_SIZE)

PMC correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE under-reports wide
coalesced reads by exactly 2x, WRITE_SIZE is exact; both are in KiB.
  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def stats(d, out):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as fo:
        fo.write(f"# rocprofv3 --kernel-trace --stats summary of {os.path.basename(d)} (total kernel time {tot / 1e6:.2f} ms)\n")
        fo.write(f"{'kernel':86s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}\n")
        for r in rows:
            if float(r["TotalDurationNs"]) / tot < 0.0005:
                continue
            fo.write(f"{short(r['Name'])[:86]:86s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.2f} "
                     f"{float(r['MinNs']) / 1e3:10.2f} {float(r['MaxNs']) / 1e3:10.2f} {float(r['Percentage']):7.2f}\n")
    print(open(out).read())


def pmc(dirs, out):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, c in acc.items():
        e = {name: sum(v) / len(v) for name, v in c.items()}
        e["launches_sampled"] = max(len(v) for v in c.values())
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
        res[k] = e
    if os.environ.get("PMC_FRAMES_PER_LAUNCH"):       # frames per launch of the profiled command (bench.py scales to its own launch size)
        res["_frames_per_launch"] = int(os.environ["PMC_FRAMES_PER_LAUNCH"])
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, e in sorted(((k, e) for k, e in res.items() if isinstance(e, dict)), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0))[:12]:
        print(k[:70], {n: round(v, 1) for n, v in e.items()})


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[3:], sys.argv[2])
