#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (CSV). usage: pmc_summary.py <dir> [COUNTER ...]"""
import collections, csv, glob, sys
d, names = sys.argv[1], sys.argv[2:]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    if names and r["Counter_Name"] not in names:
        continue
    k = r["Kernel_Name"].split("(")[0][:60]
    if not any(t in k for t in ("k_smooth", "k_gains", "k_rts_bwd_items", "k_filter_bwd_items", "k_elbo", "k_alpha", "k_filter_alpha", "k_mix", "k_lstm", "k_regime", "k_vae", "k_enc_", "k_dec_", "k_colsum", "k_latent", "k_gru")):
        continue
    a = acc[k][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f"    {c:24s} {v / n:16.1f}   (avg over {n} launches)")
