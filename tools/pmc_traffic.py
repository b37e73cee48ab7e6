#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-kernel PMC summaries of tools/profile_round.sh.
usage: pmc_traffic.py <tag>   (reads profiles/<tag>_lgssm_chain_{c2,c5}_pmc.txt, rewrites profiles/pmc_traffic.json)

HBM bytes per launch = corrected FETCH_SIZE + WRITE_SIZE (KB in the summaries).  FETCH correction (MI355X_MICROARCH.md, HBM
section): x2 for the n = 16 kernels, whose reads are 16-byte-per-lane row loads covering >= 128 contiguous bytes (checked: 2 x
269.4 MB = 539 MB against 5.4 KB x 102400 steps = 553 MB of known reads of k_smooth_fwd_n16; WRITE_SIZE 446 MB == the 4.36 KB x
102400 it stores); x1 for the n = 4 kernels (a quad reads 64 contiguous bytes: 2.21 MB fetched against 2.36 MB of known
operand reads of the quad-layout forward kernel, round 2; the matrix-core version, k_smooth_fwd_m4, reads the same lines) and for the LSTM kernels (4-byte-per-lane accesses, calibrated in round 1)."""
import json, re, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
# an op of the n = 4 path below 2048 sequences is several launches (lgssm_m4.h): the sweeps' kernel twice + the per-step items
GROUPS = {"smooth_fwd": ("k_smooth_fwd", "k_gains_m4"), "smooth_bwd": ("k_smooth_bwd", "k_rts_bwd_items", "k_filter_bwd_items"),
          "elbo": ("k_elbo",),
          "lstm_fwd": ("k_lstm_fwd",), "lstm_bwd": ("k_lstm_bwd",)}


def parse(path):
    out, cur = {}, None
    for line in path.read_text().splitlines():
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+\(avg over (\d+) launches\)", line)
        if m and cur:
            out.setdefault(cur, {})[m.group(1)] = float(m.group(2))
            out[cur]["launches"] = int(m.group(3))
        elif line and not line.startswith(" "):
            cur = line.strip()
    return out


def traffic(path, fetch_x2):
    per = parse(path)
    iters = min(c["launches"] for c in per.values())   # every kernel of the chain runs at least once per iteration of the tool
    res = {}
    for name, keys in GROUPS.items():
        tot = 0.0
        for kern, c in per.items():
            if any(k in kern for k in keys):
                x = 2.0 if any(t in kern for t in fetch_x2) else 1.0
                tot += (x * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * (c["launches"] / iters)   # launches per op
        res[name] = int(round(tot * 1024))
    return res


doc = {"_round": tag, "_source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) on "
       "tools/lgssm_chain.py; KB per launch from profiles/%s_lgssm_chain_{c2,c5}_pmc.txt; correction and bytes by tools/pmc_traffic.py "
       "(FETCH x2 for the n = 16 kernels, x1 for n = 4 / LSTM; see its header). Values are HBM bytes per launch = corrected fetch + "
       "write; 'elbo' sums the probe, the main launch and the (idle) jitter fallback." % tag,
       "B256_T50_n4_lstm_K3": traffic(ROOT / "profiles" / f"{tag}_lgssm_chain_c2_pmc.txt", ()),
       "B512_T200_n16_lstm_K3": traffic(ROOT / "profiles" / f"{tag}_lgssm_chain_c5_pmc.txt", ("_n16", "SDims<16"))}
for key, name, x2 in (("B32_T100_n4_switching_K7", "c4", ()), ("B512_T200_n16_switching_K3", "c5_switching", ("_n16", "SDims<16"))):
    f = ROOT / "profiles" / f"{tag}_lgssm_chain_{name}_pmc.txt"
    if f.exists():
        doc[key] = traffic(f, x2)
(ROOT / "profiles" / "pmc_traffic.json").write_text(json.dumps(doc, indent=1) + "\n")
print(json.dumps(doc, indent=1))
