#!/usr/bin/env python3
"""Micro-benchmark of the alpha-network tail kernels (rnn_wgrad, small_linear) at a given (B, T): HIP-event times per call;
run under `rocprofv3 --kernel-trace --stats` for per-kernel durations.  usage: python tools/rnn_probe.py [B] [T] [iters]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "kalman-vae_amd")]
import torch  # noqa: E402

from kvae.kalman.lgssm_ops import SmallLinear, rnn_wgrad  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = "cuda"
H, I, K = 50, 2, 3
N = B * T
d = torch.randn(N, 4 * H, device=dev)
h = torch.randn(N, H, device=dev)
x = torch.randn(N, I, device=dev)
gl = torch.randn(N, K, device=dev)
dpi, dph = torch.randn(2, N, 3 * H, device=dev), torch.randn(2, N, 3 * H, device=dev)
h2 = torch.randn(N, 2 * H, device=dev)
w = torch.randn(K, H, device=dev, requires_grad=True)
b = torch.randn(K, device=dev, requires_grad=True)
w9 = torch.randn(9, 2 * H, device=dev)
b9 = torch.randn(9, device=dev)


def timeit(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    print(f"{name:34s} {1e3 * s.elapsed_time(e) / iters:9.1f} us/call")


lstm = [dict(d=d, h=h, shift=-1, T=T, x=x)]
gru = []
for dr in (0, 1):
    gru.append(dict(d=dpi[dr], x=x))
    gru.append(dict(d=dph[dr], h=h2[:, dr * H:(dr + 1) * H], shift=-1 if dr == 0 else 1, T=T))
timeit("rnn_wgrad lstm (200 x 53)", lambda: rnn_wgrad(d, lstm))
timeit("rnn_wgrad head (3 x 51)", lambda: rnn_wgrad(d, [dict(d=gl, h=h)]))
timeit("rnn_wgrad bi-GRU (4 problems)", lambda: rnn_wgrad(d, gru))
with torch.no_grad():
    timeit("linear+softmax fwd (K=3, F=50)", lambda: SmallLinear.apply(h, w, b, True))
    timeit("linear fwd (O=9, F=100)", lambda: SmallLinear.apply(h2, w9, b9, False))
hh = h.clone().requires_grad_(True)
y = SmallLinear.apply(hh, w, b, True)
gy = torch.randn_like(y)
timeit("linear+softmax bwd (all grads)", lambda: torch.autograd.grad(y, (hh, w, b), gy, retain_graph=True))

from kvae.kalman.lgssm_ops import RegimeChain  # noqa: E402
for (Br, Tr, Kr) in ((32, 100, 7), (256, 50, 3)):
    gen = torch.Generator().manual_seed(0)
    lg = torch.randn(Br, Tr, Kr, Kr, generator=gen).to(dev).requires_grad_(True)
    il = torch.randn(Br, Kr, generator=gen).to(dev).requires_grad_(True)
    gm = (-torch.empty(Br, Tr, Kr).exponential_(generator=gen).log()).to(dev)
    Pm = torch.full((Kr, Kr), 0.1 / (Kr - 1), device=dev)
    Pm.fill_diagonal_(0.9)
    with torch.no_grad():
        timeit(f"regime fwd B={Br} T={Tr} K={Kr}", lambda: RegimeChain.apply(lg, il, gm, Pm, 0.7, False))
    yq = RegimeChain.apply(lg, il, gm, Pm, 0.7, False)
    ups = [torch.randn_like(t) for t in yq]
    timeit(f"regime bwd B={Br} T={Tr} K={Kr}", lambda: torch.autograd.grad(yq, (lg, il), ups, retain_graph=True))

from kvae.kalman.lgssm_ops import MixDynamics  # noqa: E402
for (Br, Tr, Kr, Er) in ((512, 200, 3, 768), (512, 200, 3, 544), (256, 50, 3, 40)):
    al = torch.softmax(torch.randn(Br, Tr, Kr, device=dev), -1).requires_grad_(True)
    bs = torch.randn(Kr, Er, device=dev).requires_grad_(True)
    with torch.no_grad():
        timeit(f"mix fwd rows={Br * Tr} K={Kr} E={Er}", lambda: MixDynamics.apply(al, bs))
    rec = MixDynamics.apply(al, bs)
    upm = torch.randn_like(rec)
    timeit(f"mix bwd rows={Br * Tr} K={Kr} E={Er}", lambda: torch.autograd.grad(rec, (al, bs), upm, retain_graph=True))
