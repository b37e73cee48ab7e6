#!/usr/bin/env python3
"""Decoder up-blocks alone (csrc/vae_conv_up.h direct vs csrc/vae_conv_up_wino.h Winograd): HIP-event times at the C2 frame
count and the error of each against torch's conv2d + pixel_shuffle + relu.  KVAE_WINO=0 selects the direct kernels.
usage: [KVAE_WINO=0] python tools/dec_up_probe.py [frames]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kalman-vae_amd"))
import torch
import torch.nn.functional as F
from kvae.vae.fused import DecoderUp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
dev = torch.device("cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


print(f"frames {N}, KVAE_WINO={os.environ.get('KVAE_WINO', '1')}")
for side in (8, 4):
    g = torch.Generator().manual_seed(side)
    x = torch.relu(torch.randn(N, 32, side, side, generator=g)).to(dev).requires_grad_(True)
    W = (0.08 * torch.randn(128, 32, 3, 3, generator=g)).to(dev).requires_grad_(True)
    b = (0.1 * torch.randn(128, generator=g)).to(dev).requires_grad_(True)
    up = torch.randn(N, 32, 2 * side, 2 * side, generator=g).to(dev)
    out = DecoderUp.apply(x, W, b)
    gx, gW, gb = torch.autograd.grad(out, (x, W, b), up, retain_graph=True)
    n = min(N, 512)   # reference on a slice (MIOpen in fp32), weight gradients on the same slice
    xr, Wr, br = x[:n].detach().clone().requires_grad_(True), W.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = torch.relu(F.pixel_shuffle(F.conv2d(xr, Wr, br, padding=1), 2))
    rgx, rgW, rgb = torch.autograd.grad(ref, (xr, Wr, br), up[:n])
    o2 = DecoderUp.apply(x[:n].detach().requires_grad_(True), W, b)
    gW2, gb2 = torch.autograd.grad(o2, (W, b), up[:n])
    print(f"dec_up s={side}: err out {rel(out[:n], ref):.2e}  g_x {rel(gx[:n], rgx):.2e}  g_W {rel(gW2, rgW):.2e}  g_b {rel(gb2, rgb):.2e}")
    gmac = N * side ** 2 * 128 * 288 / 1e9
    print(f"  direct-conv MFMA floor {2 * gmac / 157e3 * 1e6:6.0f} us, Winograd F(2,3) floor {2 * gmac / 2.25 / 157e3 * 1e6:6.0f} us")
    print(f"  fwd            {timed(lambda: DecoderUp.apply(x, W, b)):8.1f} us")
    print(f"  bwd (data+w)   {timed(lambda: torch.autograd.grad(out, (x, W, b), up, retain_graph=True)):8.1f} us")
    print(f"  bwd (w only)   {timed(lambda: torch.autograd.grad(out, (W, b), up, retain_graph=True)):8.1f} us")
