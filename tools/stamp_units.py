#!/usr/bin/env python3
"""Whole-kernel view of the stamps of block 8 (tools/stamp_probe.py output on stdin): per-unit period of the
consumer wave, and the total span of the block (divide by the kernel's duration for the tick rate)."""
import sys
rows = {}
for line in sys.stdin:
    if line.startswith("wave "):
        w, rest = line.split(":", 1)
        rows[int(w.split()[1])] = [int(x) for x in rest.split()]
c = [x for x in rows[0] if x >= 0]
n = max(i for i, x in enumerate(c) if x > 0)
post = [c[i] for i in range(4, n + 1, 3)]          # post-barrier stamp of every unit
per = [b - a for a, b in zip(post, post[1:])]
print("units stamped", len(post), "| last stamp", c[n], "| unit periods", per)
