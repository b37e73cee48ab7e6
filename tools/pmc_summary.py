#!/usr/bin/env python3
"""Per-kernel average of one rocprofv3 --pmc counter (CSV): prof dir -> 'kernel avg_value calls'."""
import collections, csv, glob, sys
d, name = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != name:
        continue
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][0] += 1
    acc[k][1] += float(r["Counter_Value"])
for k, (c, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if k.startswith(("void k_", "k_", "void kvae", "_ZN4kvae")):
        print(f"{name} {v / c:14.1f} avg over {c:4d} calls  {k}")
