#!/usr/bin/env python3
"""Print name (short), calls, average us for our kernels from a rocprofv3 kernel_stats.csv."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("k_smooth", "k_gains", "k_rts_bwd_items", "k_filter_bwd_items", "k_elbo", "k_alpha", "k_filter_alpha", "k_mix", "k_lstm", "k_regime", "k_vae", "k_enc_", "k_dec_", "k_gru", "k_colsum", "k_latent", "k_rnn", "k_linear", "k_emission", "k_clip_adam", "k_grad_sumsq", "k_loss_head")):
        print(f"{float(r['AverageNs']) / 1e3:9.2f} us  x{r['Calls']:>4}  {n.split('(')[0][:70]}")
