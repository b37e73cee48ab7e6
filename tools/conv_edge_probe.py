#!/usr/bin/env python3
"""Times the direct VAE-edge convolutions (csrc/vae_conv_edge.h) at the C2 frame count against MIOpen's kernels for the
same layers.  usage: python tools/conv_edge_probe.py [frames]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kalman-vae_amd"))
import torch
import torch.nn.functional as F
from kvae.vae.fused import DecoderHead, EncoderStem

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


h = torch.relu(torch.randn(N, 32, 16, 16, device=dev)).requires_grad_(True)
W = (0.1 * torch.randn(4, 32, 3, 3, device=dev)).requires_grad_(True)
b = torch.randn(4, device=dev, requires_grad=True)
up = torch.randn(N, 1, 32, 32, device=dev)
x = torch.rand(N, 1, 32, 32, device=dev)
We = (0.3 * torch.randn(32, 1, 3, 3, device=dev)).requires_grad_(True)
be = torch.randn(32, device=dev, requires_grad=True)
upe = torch.randn(N, 32, 16, 16, device=dev)

out = DecoderHead.apply(h, W, b)
print(f"frames {N}")
print(f"dec_head fwd           {timed(lambda: DecoderHead.apply(h, W, b)):8.1f} us   (419 MB read -> {419.4 * N / 12800 / 5.3e3 * 1e3:.0f} us at 5.3 TB/s)")
print(f"dec_head bwd (data+w)  {timed(lambda: torch.autograd.grad(out, (h, W, b), up, retain_graph=True)):8.1f} us")
print(f"dec_head bwd (w only)  {timed(lambda: torch.autograd.grad(out, (W, b), up, retain_graph=True)):8.1f} us")
ref = F.pixel_shuffle(F.conv2d(h, W, b, padding=1), 2)
print(f"miopen   fwd+shuffle   {timed(lambda: F.pixel_shuffle(F.conv2d(h, W, b, padding=1), 2)):8.1f} us")
print(f"miopen   bwd           {timed(lambda: torch.autograd.grad(ref, (h, W, b), up, retain_graph=True)):8.1f} us")
oe = EncoderStem.apply(x, We, be)
print(f"enc_stem fwd           {timed(lambda: EncoderStem.apply(x, We, be)):8.1f} us")
print(f"enc_stem bwd           {timed(lambda: torch.autograd.grad(oe, (We, be), upe, retain_graph=True)):8.1f} us")
re_ = torch.relu(F.conv2d(x, We, be, stride=2, padding=1))
print(f"miopen   fwd+relu      {timed(lambda: torch.relu(F.conv2d(x, We, be, stride=2, padding=1))):8.1f} us")
print(f"miopen   bwd           {timed(lambda: torch.autograd.grad(re_, (We, be), upe, retain_graph=True)):8.1f} us")

from kvae.vae.fused import EncoderMid
for side in (16, 8):
    xm = torch.relu(torch.randn(N, 32, side, side, device=dev)).requires_grad_(True)
    Wm = (0.08 * torch.randn(32, 32, 3, 3, device=dev)).requires_grad_(True)
    bm = torch.randn(32, device=dev, requires_grad=True)
    upm = torch.randn(N, 32, side // 2, side // 2, device=dev)
    om = EncoderMid.apply(xm, Wm, bm)
    gmac = N * (side // 2) ** 2 * 32 * 288 / 1e9
    print(f"enc_mid s={side} ({gmac:.1f} GMAC/pass, MFMA-f32 floor {2 * gmac / 157e3 * 1e6:.0f} us)")
    print(f"  mfma   fwd            {timed(lambda: EncoderMid.apply(xm, Wm, bm)):8.1f} us")
    print(f"  mfma   bwd (data+w)   {timed(lambda: torch.autograd.grad(om, (xm, Wm, bm), upm, retain_graph=True)):8.1f} us")
    print(f"  mfma   bwd (w only)   {timed(lambda: torch.autograd.grad(om, (Wm, bm), upm, retain_graph=True)):8.1f} us")
    rm = torch.relu(F.conv2d(xm, Wm, bm, stride=2, padding=1))
    print(f"  miopen fwd+relu       {timed(lambda: torch.relu(F.conv2d(xm, Wm, bm, stride=2, padding=1))):8.1f} us")
    print(f"  miopen bwd            {timed(lambda: torch.autograd.grad(rm, (xm, Wm, bm), upm, retain_graph=True)):8.1f} us")

from kvae.vae.fused import DecoderUp
for side in (8, 4):
    xm = torch.relu(torch.randn(N, 32, side, side, device=dev)).requires_grad_(True)
    Wm = (0.08 * torch.randn(128, 32, 3, 3, device=dev)).requires_grad_(True)
    bm = torch.randn(128, device=dev, requires_grad=True)
    upm = torch.randn(N, 32, 2 * side, 2 * side, device=dev)
    om = DecoderUp.apply(xm, Wm, bm)
    gmac = N * side ** 2 * 128 * 288 / 1e9
    print(f"dec_up s={side} ({gmac:.1f} GMAC/pass, MFMA-f32 floor {2 * gmac / 157e3 * 1e6:.0f} us)")
    print(f"  mfma   fwd            {timed(lambda: DecoderUp.apply(xm, Wm, bm)):8.1f} us")
    print(f"  mfma   bwd (data+w)   {timed(lambda: torch.autograd.grad(om, (xm, Wm, bm), upm, retain_graph=True)):8.1f} us")
    rm = torch.relu(F.pixel_shuffle(F.conv2d(xm, Wm, bm, padding=1), 2))
    print(f"  miopen fwd+shuffle+relu {timed(lambda: torch.relu(F.pixel_shuffle(F.conv2d(xm, Wm, bm, padding=1), 2))):8.1f} us")
    print(f"  miopen bwd            {timed(lambda: torch.autograd.grad(rm, (xm, Wm, bm), upm, retain_graph=True)):8.1f} us")
