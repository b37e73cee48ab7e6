#!/usr/bin/env python3
"""Kernel sequence of the LAST steady-state step of a rocprofv3 --kernel-trace CSV (start offset, duration, gap before)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
anch = [i for i, r in enumerate(rows) if "k_smooth_bwd" in r[2]]
win = rows[anch[-2] + 1: anch[-1] + 1]
t0, prev_end = win[0][0], win[0][0]
for s, e, n in win:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {n.split('(')[0][:80]}")
    prev_end = max(prev_end, e)
