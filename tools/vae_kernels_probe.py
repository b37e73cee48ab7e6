#!/usr/bin/env python3
"""Profiling target: every hand-written VAE kernel at the C2 frame count (12800), forward + backward, no library convs.
  rocprofv3 --kernel-trace --stats -- python3 tools/vae_kernels_probe.py
  rocprofv3 --pmc FETCH_SIZE -- python3 tools/vae_kernels_probe.py     (separate pass)
  rocprofv3 --pmc WRITE_SIZE -- python3 tools/vae_kernels_probe.py     (separate pass)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kalman-vae_amd"))
import torch
from kvae.vae.fused import (BernoulliFrameLogLik, DecoderFc, DecoderHead, DecoderUp, EncoderHead, EncoderMid, EncoderStem,
                            LatentReg)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda")
P = lambda *s: (0.05 * torch.randn(*s, device=dev)).requires_grad_(True)
x = torch.rand(N, 1, 32, 32, device=dev)
w = dict(stem=(P(32, 1, 3, 3), P(32)), m16=(P(32, 32, 3, 3), P(32)), m8=(P(32, 32, 3, 3), P(32)), mu=(P(2, 512), P(2)),
         var=(P(2, 512), P(2)), fc=(P(512, 2), P(512)), u4=(P(128, 32, 3, 3), P(128)), u8=(P(128, 32, 3, 3), P(128)),
         head=(P(4, 32, 3, 3), P(4)))
for _ in range(reps):
    h = EncoderStem.apply(x, *w["stem"])
    h = EncoderMid.apply(h, *w["m16"])
    h = EncoderMid.apply(h, *w["m8"])
    a, mu, var = EncoderHead.apply(h.flatten(1), *w["mu"], *w["var"], torch.randn(N, 2, device=dev), 0.03)
    d = DecoderFc.apply(a, *w["fc"]).unflatten(1, (32, 4, 4))
    d = DecoderUp.apply(d, *w["u4"])
    d = DecoderUp.apply(d, *w["u8"])
    logits = DecoderHead.apply(d, *w["head"])
    ll = BernoulliFrameLogLik.apply(logits.view(N // 50 if N % 50 == 0 else N, -1, 1, 32, 32), x.view(N // 50 if N % 50 == 0 else N, -1, 1, 32, 32))
    reg = LatentReg.apply(a, mu, var)
    (ll.sum() + reg.sum()).backward()
torch.cuda.synchronize()
print("done")
