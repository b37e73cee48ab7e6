#!/usr/bin/env python3
"""Timing (HIP events) and a C-oracle check of the LGSSM kernels through the op layer - filter only, RTS only, filter + RTS,
ELBO, backward - with the algorithmic GB/s of SURVEY.md section 8(d) next to each.  Defaults: the BASELINE configs[4] shard.
  python3 tools/lgssm_probe.py [--n 16] [--B 512] [--T 200] [--iters 10] [--q-per-step] [--mask] [--check]
A/B switches (read once per process): KVAE_N16=0 generic kernels for n = 16; KVAE_Q4=0 one wavefront per sequence for
n = 4 (lgssm_n4.h) instead of sixteen sequences per wavefront (lgssm_m4.h)."""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "kalman-vae_amd"), str(ROOT / "tests")]
import torch  # noqa: E402

from kvae.kalman import lgssm_ops as ops  # noqa: E402
from kvae.kalman.lgssm_ops import LgssmElbo, LgssmSmooth, Slots, mix_dynamics  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16)
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--T", type=int, default=200)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--q-per-step", action="store_true")
ap.add_argument("--check", action="store_true", help="compare with the C oracle on the first 4 sequences")
ap.add_argument("--mask", action="store_true")
a = ap.parse_args()
dev = "cuda"
B, T, n, m, p, K = a.B, a.T, a.n, a.n, 2, 3
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
A = (torch.eye(n).repeat(K, 1, 1) + 0.05 * torch.randn(K, n, n, generator=g)).to(dev).requires_grad_(True)
Bm = (0.05 * r(K, n, m)).requires_grad_(True)
Cm = (0.3 * r(K, p, n)).requires_grad_(True)
qq = 0.05 * torch.randn(K, n, n, generator=g)
Qk = (0.02 * torch.eye(n).repeat(K, 1, 1) + qq @ qq.mT).to(dev).requires_grad_(True)
alpha = torch.softmax(r(B, T, K), -1).requires_grad_(True)
Y = r(B, T, p).requires_grad_(True)
U = 0.3 * r(B, T, m)
mask = (torch.rand(B, T, generator=g) > 0.2).float().to(dev) if a.mask else None
R, Q = 0.03 * torch.eye(p, device=dev), 0.02 * torch.eye(n, device=dev)
mu0, S0 = torch.zeros(n, device=dev), 20.0 * torch.eye(n, device=dev)
eps = r(B, T, n)


def operands():
    if a.q_per_step:
        rec, offs, views = mix_dynamics(alpha, [A, Bm, Qk])
        return rec, Slots(A=offs[0], B=offs[1], Q=offs[2]), Cm[0], None, views
    rec, offs, views = mix_dynamics(alpha, [A, Bm, Cm])
    return rec, Slots(A=offs[0], B=offs[1], C=offs[2]), None, Q, views


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


rec, slots, Cop, Qop, views = operands()
recd = rec.detach()
with torch.no_grad():
    t_filter = timed(lambda: LgssmSmooth.apply(Y, U, mask, recd, None, None, Cop, Qop, R, mu0, S0, slots, False), a.iters)
    mf, Sf, mp, Sp = LgssmSmooth.apply(Y, U, mask, recd, None, None, Cop, Qop, R, mu0, S0, slots, False)
    t_rts = timed(lambda: ops.rts_only(Y, U, mask, recd, None, None, Cop, Qop, R, mu0, S0, slots, mf, Sf, mp, Sp), a.iters)
    t_smooth = timed(lambda: LgssmSmooth.apply(Y, U, mask, recd, None, None, Cop, Qop, R, mu0, S0, slots, True), a.iters)
    ms, Ss, *_ = LgssmSmooth.apply(Y, U, mask, recd, None, None, Cop, Qop, R, mu0, S0, slots, True)
    t_elbo = timed(lambda: LgssmElbo.apply(ms, Ss, eps, Y, U, mask, recd, None, None, Cop, Qop, R, mu0, S0, slots), a.iters)

ops_profile = {}
from kvae import _native  # noqa: E402

_native.profile_start()
for _ in range(a.iters):
    rec, slots, Cop, Qop, views = operands()
    ms, Ss, mf, Sf, mp, Sp = LgssmSmooth.apply(Y, U, mask, rec, None, None, Cop, Qop, R, mu0, S0, slots, True)
    total, _, levels_dev = LgssmElbo.apply(ms, Ss, eps, Y, U, mask, rec, None, None, Cop, Qop, R, mu0, S0, slots)
    (total / (B * T)).backward()
prof = _native.profile_stop()
us = {k: 1e3 * sorted(v)[len(v) // 2] for k, v in prof.items()}
qs = 1 if a.q_per_step else 0
by_fwd = 4 * (n * n * (1 + qs) + n * m + p * n + p + m + 1 + 3 * n + 3 * n * n)
by_elbo = 4 * (2 * n + n * n + p + m + n * n + n * m + p * n + qs * n * n + 1)
by_bwd = by_fwd + 4 * ((n + n * n) + n * n * (1 + qs) + n * m + p * n + p)
per = lambda t_us, by=None: f"{t_us:9.1f} us = {t_us / T * 1e3:7.0f} ns/step" + (
    f" = {by * B * T / (t_us * 1e-6) / 1e9:7.1f} GB/s algorithmic" if by and t_us > 0 else "")
print(f"n={n} B={B} T={T} q_per_step={a.q_per_step} mask={a.mask}")
print(f"  filter only   {per(t_filter)}")
print(f"  rts only      {per(t_rts)}")
print(f"  filter + rts  {per(t_smooth, by_fwd)}   (with grads/aux: {per(us.get('smooth_fwd', 0.0))})")
print(f"  elbo          {per(t_elbo, by_elbo)}   (with grads: {per(us.get('elbo', 0.0))})")
print(f"  backward      {per(us.get('smooth_bwd', 0.0), by_bwd)}")

if a.check:
    from golden_util import rel_err
    from oracle import c_oracle
    nb = min(B, 4)
    c = lambda t: t.detach().cpu()
    As, Bs, X3 = views
    if a.q_per_step:
        Cs, Qs = c(Cm[0]), c(X3)[:nb]
    else:
        Cs, Qs = c(X3)[:nb], c(Q)
    ref = c_oracle.smooth(c(Y)[:nb], c(U)[:nb], None if mask is None else c(mask)[:nb], c(As)[:nb], c(Bs)[:nb], Cs, Qs, c(R),
                          c(mu0), c(S0))
    for k, v in (("mus_smooth", ms), ("Sigmas_smooth", Ss), ("mus_filt", mf), ("Sigmas_filt", Sf), ("mus_pred", mp),
                 ("Sigmas_pred", Sp)):
        print(f"  vs C oracle {k:14s} rel err {rel_err(c(v)[:nb], ref[k]):.2e}")
