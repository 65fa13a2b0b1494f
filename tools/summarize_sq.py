#!/usr/bin/env python3
"""SQ-counter summary of rocprofv3 --pmc passes: per kernel, the mean of every counter over its dispatches, plus the ratios
the review asks for (MFMA busy share of the busy cycles, issue / stall / parked shares of the wave cycles, LDS conflict share).

  python tools/summarize_sq.py <out.txt> <title> <kernel substring>[,...] <pmc dir> [<pmc dir> ...]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    out, title, subs, dirs = sys.argv[1], sys.argv[2], sys.argv[3].split(";"), sys.argv[4:]
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(lambda: defaultdict(float))
            names = {}
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not any(s in k for s in subs):
                    continue
                per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = k
            for did, c in per_dispatch.items():
                for n, v in c.items():
                    acc[names[did]][n].append(v)
    with open(out, "a") as fo:
        fo.write(f"## {title}\n")
        for k in sorted(acc):
            e = {n: sum(v) / len(v) for n, v in acc[k].items()}
            fo.write(f"{k}\n")
            fo.write("  " + "  ".join(f"{n}={e[n]:.4g}" for n in sorted(e)) + "\n")
            wc = e.get("SQ_WAVE_CYCLES")
            if wc:
                parts = []
                if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CYCLES" in e:
                    # MFMA_BUSY counts cycles summed over the SIMDs' matrix pipes; BUSY_CYCLES the SQ-busy cycles per SE: the ratio
                    # used in round 2 (profiles/r02_pmc_rs_vs_streaming_32to32.txt) is MFMA busy / (4 x SQ busy cycles per CU share)
                    parts.append(f"mfma_busy/wave_cycles={e['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc):.3f}")
                for n, lab in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "stalled_at_issue"), ("SQ_WAIT_ANY", "parked")):
                    if n in e:
                        parts.append(f"{lab}={e[n] / wc:.3f}")
                fo.write("  shares of wave cycles (quad-cycle units; MFMA busy in cycles / 4): " + "  ".join(parts) + "\n")
            if e.get("SQ_LDS_IDX_ACTIVE"):
                fo.write(f"  LDS bank-conflict cycles / LDS active cycles = {e.get('SQ_LDS_BANK_CONFLICT', 0) / e['SQ_LDS_IDX_ACTIVE']:.4f}\n")
        fo.write("\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
