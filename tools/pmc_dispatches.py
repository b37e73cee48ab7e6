#!/usr/bin/env python3
"""Per-dispatch counters of one kernel from a rocprofv3 --pmc CSV run, in dispatch order.
usage: pmc_dispatches.py <dir> <kernel substring> [max rows]"""
import collections, csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 12
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if pat not in r["Kernel_Name"]:
        continue
    rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in rows.values() for k in v})
print("dispatch " + " ".join(f"{n:>20s}" for n in names))
for i, (k, v) in enumerate(rows.items()):
    if i >= lim:
        break
    print(f"{k:8d} " + " ".join(f"{v.get(n, float('nan')):20.0f}" for n in names))
