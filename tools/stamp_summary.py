#!/usr/bin/env python3
"""Summarise tools/stamp_probe.py output (stderr lines 'wave N: t0 t1 ...') of the bf16x3 conv
kernel: every wave = [kernel entry, ..., kernel exit]; between them consumer wave 0 = [pre-barrier, post-barrier, then per unit:
slot-loop end, pre-barrier, post-barrier]; producer wave 4 = [start, staged, post-barrier, then per unit: staged, post-barrier]."""
import sys
rows = {}
for line in sys.stdin:
    if line.startswith("wave "):
        w, rest = line.split(":", 1)
        rows[int(w.split()[1])] = [int(x) for x in rest.split()]
entry = rows[0][0]
last = max(x for r in rows.values() for x in r)
c, p = rows[0][1:], rows[4][1:]
units = [(c[2 + 3 * i] - c[1 + 3 * i], c[3 + 3 * i] - c[2 + 3 * i], c[4 + 3 * i] - c[3 + 3 * i]) for i in range(1, 10) if c[4 + 3 * i] > 0]
loop = sorted(u[0] for u in units)
print("consumer: first barrier at", c[1], "| slot loop median", loop[len(loop) // 2], "min", loop[0], "max", loop[-1],
      "| post-loop (epilogue on odd units) ", [u[1] for u in units[:6]], "| barrier wait", [u[2] for u in units[:6]])
st = [(p[3 + 2 * i] - p[2 + 2 * i], p[4 + 2 * i] - p[3 + 2 * i]) for i in range(0, 10) if p[4 + 2 * i] > 0]
print("producer: first stage", p[1] - p[0], "| stage times", [s[0] for s in st[:8]], "| barrier waits", [s[1] for s in st[:8]])
print("total ticks (consumer)", max(c), "| kernel entry -> first stamp", c[0] - entry, "| entry -> exit", last - entry)
