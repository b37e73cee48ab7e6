#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV over the STEADY-STATE steps only.

MIOpen's first-call solver search launches >1e6 throw-away kernels during warm-up, which swamps
`--stats`.  This tool keeps the dispatches of the last K training steps (delimited by the last K+1
launches of an anchor kernel, default k_smooth_bwd) and prints per-kernel totals per step.
usage: prof_summary.py <kernel_trace.csv> [--steps K] [--anchor NAME] [--top N]"""
import argparse
import csv
import collections
import sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--anchor", default="k_smooth_bwd")
ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()
rows = []
with open(a.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
anchors = [i for i, r in enumerate(rows) if a.anchor in r[2]]
if len(anchors) < a.steps + 1:
    sys.exit(f"only {len(anchors)} anchor launches")
lo, hi = anchors[-a.steps - 1] + 1, anchors[-1] + 1
# extend to the end of the last step: up to the next anchor-distance worth of kernels is unknown; use [lo, hi)
win = rows[lo:hi]
span = (win[-1][1] - win[0][0]) / a.steps
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    tot[n][0] += 1
    tot[n][1] += e - s
busy = sum(v[1] for v in tot.values()) / a.steps
print(f"steady-state window: {a.steps} steps, {len(win) / a.steps:.1f} kernels/step, wall {span / 1e3:.1f} us/step, "
      f"GPU busy {busy / 1e3:.1f} us/step ({100 * busy / span:.1f}%)")
print(f"{'us/step':>10} {'calls/step':>10} {'avg us':>9}  kernel")
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[: a.top]:
    print(f"{d / a.steps / 1e3:10.2f} {c / a.steps:10.1f} {d / c / 1e3:9.2f}  {n[:110]}")
