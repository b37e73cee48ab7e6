#!/usr/bin/env python3
"""Profiling target for the MFMA encoder layers only (no MIOpen): rocprofv3 --kernel-trace --stats / --pmc ... -- python3 tools/enc_mid_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kalman-vae_amd"))
import torch
from kvae.vae.fused import EncoderMid
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda")
for side in (16, 8):
    x = torch.relu(torch.randn(N, 32, side, side, device=dev)).requires_grad_(True)
    W = (0.08 * torch.randn(32, 32, 3, 3, device=dev)).requires_grad_(True)
    b = torch.randn(32, device=dev, requires_grad=True)
    up = torch.randn(N, 32, side // 2, side // 2, device=dev)
    for _ in range(reps):
        out = EncoderMid.apply(x, W, b)
        torch.autograd.grad(out, (x, W, b), up)
    torch.cuda.synchronize()
print("done")
