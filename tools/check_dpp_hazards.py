#!/usr/bin/env python3
"""Static check of hipcc's .s output for the hand-written DPP instructions (inline asm is invisible to the compiler's hazard
recogniser): a VALU write of a VGPR followed within 2 wait states by a DPP read of that VGPR is a hazard on gfx9.
usage: check_dpp_hazards.py <file.s>   (exit 1 on a finding)"""
import re
import sys

lines = [l.strip() for l in open(sys.argv[1]) if l.strip() and not l.strip().startswith((";", ".", "/"))]
ins = [l for l in lines if re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", l)]
bad = 0
for i, l in enumerate(ins):
    m0 = re.match(r"v_(?:fmac|mul)_f32_dpp (v\d+), (v\d+),", l)
    if not m0:
        continue
    src = m0.group(2)
    waits = 0
    for k in range(i - 1, max(i - 4, -1), -1):
        p = ins[k]
        if p.startswith("s_nop"):
            waits += int(p.split()[1]) + 1
            continue
        if waits >= 2:
            break
        m = re.match(r"^v_\w+ (v\d+|v\[\d+:\d+\])", p)
        if m and p.startswith("v_") and not p.startswith("v_cmp"):
            dst = m.group(1)
            regs = {dst} if not dst.startswith("v[") else {f"v{x}" for x in range(int(dst[2:-1].split(":")[0]), int(dst[2:-1].split(":")[1]) + 1)}
            if src in regs:
                print(f"HAZARD: '{p}' then '{l}' with {waits} wait state(s)")
                bad += 1
        waits += 1
print(f"checked {sum(1 for l in ins if re.match(r'v_(fmac|mul)_f32_dpp', l))} hand-written DPP instructions, {bad} hazard(s)")
sys.exit(1 if bad else 0)
