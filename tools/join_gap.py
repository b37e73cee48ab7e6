#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of bench.py: per step, the time between the end of the decoder's last backward kernel
(k_dec_fc_bwd) and the start of the encoder's first (k_enc_head_bwd), and what ran in between.
usage: join_gap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
rows.sort()
ends = [i for i, r in enumerate(rows) if "k_dec_fc_bwd" in r[2]]
gaps = []
for i in ends[-8:]:
    nxt = next((k for k in range(i + 1, len(rows)) if "k_enc_head_bwd" in rows[k][2]), None)
    if nxt is None:
        continue
    gaps.append((rows[nxt][0] - rows[i][1]) / 1e3)
    if i == ends[-2]:
        print(f"between k_dec_fc_bwd end and k_enc_head_bwd start ({gaps[-1]:.1f} us):")
        for s, e, n in rows[i + 1:nxt]:
            print(f"   +{(s - rows[i][1]) / 1e3:7.1f} .. +{(e - rows[i][1]) / 1e3:7.1f}  {n}")
print("gap per step (us):", [round(g, 1) for g in gaps])
