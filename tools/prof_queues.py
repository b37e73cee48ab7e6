#!/usr/bin/env python3
"""Per-queue (HIP stream) view of the steady-state steps of a rocprofv3 --kernel-trace CSV: busy time, launch count and
the gaps between consecutive kernels on the busiest queue.  usage: prof_queues.py <kernel_trace.csv> [--steps K]"""
import argparse, collections, csv
ap = argparse.ArgumentParser(); ap.add_argument("trace"); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--anchor", default="k_smooth_bwd"); a = ap.parse_args()
rows = []
with open(a.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
anch = [i for i, r in enumerate(rows) if a.anchor in r[2]]
win = rows[anch[-a.steps - 1] + 1: anch[-1] + 1]
byq = collections.defaultdict(list)
for r in win: byq[r[3]].append(r)
print(f"window: {a.steps} steps, wall {(win[-1][1] - win[0][0]) / a.steps / 1e3:.1f} us/step")
for q, rs in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1])):
    busy = sum(e - s for s, e, _, _ in rs) / a.steps / 1e3
    gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
    small = [g for g in gaps if 0 < g < 50000]
    print(f"queue {q}: {len(rs) / a.steps:.1f} launches/step, busy {busy:.1f} us/step, inter-kernel gaps <50us: "
          f"{sum(small) / a.steps / 1e3:.1f} us/step (mean {sum(small) / max(len(small), 1) / 1e3:.2f} us)")
    tot = collections.defaultdict(lambda: [0, 0])
    for s, e, n, _ in rs:
        tot[n.split("(")[0][:70]][0] += 1; tot[n.split("(")[0][:70]][1] += e - s
    for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:45]:
        if d / c < 30000:   # only the small ones are interesting here
            print(f"      {d / a.steps / 1e3:8.1f} us {c / a.steps:5.1f}x  {n}")
