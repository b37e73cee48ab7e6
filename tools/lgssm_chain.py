#!/usr/bin/env python3
"""Run ONLY the HIP LGSSM chain (mix -> smooth fwd -> elbo -> backward) at a BASELINE config size, N times.
Profiling target for rocprofv3 (no MIOpen warm-up noise):
  rocprofv3 --kernel-trace --stats -- python3 tools/lgssm_chain.py
  rocprofv3 --pmc FETCH_SIZE -- python3 tools/lgssm_chain.py      (separate pass)
  rocprofv3 --pmc WRITE_SIZE -- python3 tools/lgssm_chain.py      (separate pass)
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "kalman-vae_amd")]
import torch  # noqa: E402

from kvae.kalman.lgssm_ops import LgssmElbo, LgssmSmooth, LstmSequence, Slots, mix_dynamics  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--T", type=int, default=50)
ap.add_argument("--n", type=int, default=4)
ap.add_argument("--K", type=int, default=3)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--q-per-step", action="store_true", help="switching-style per-step Q (A|B|Q record)")
a = ap.parse_args()
dev = "cuda"
B, T, n, m, p, K = a.B, a.T, a.n, a.n, 2, a.K
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
pert = 0.05 * (2.0 / n ** 0.5) * min(1.0, 50.0 / T)   # the same growth over the sequence at every (n, T), as bench.py's model
A = (torch.eye(n).repeat(K, 1, 1) + pert * torch.randn(K, n, n, generator=g)).to(dev).requires_grad_(True)
Bm = (0.05 * r(K, n, m)).requires_grad_(True)
Cm = (0.3 * r(K, p, n)).requires_grad_(True)
Qk = (0.02 * torch.eye(n).repeat(K, 1, 1)).to(dev).requires_grad_(True)
alpha = torch.softmax(r(B, T, K), -1).requires_grad_(True)
Y = r(B, T, p).requires_grad_(True)
U = torch.zeros(B, T, m, device=dev)
R, Q = 0.03 * torch.eye(p, device=dev), 0.02 * torch.eye(n, device=dev)
mu0, S0 = torch.zeros(n, device=dev), 20.0 * torch.eye(n, device=dev)
eps = r(B, T, n)
lstm = torch.nn.LSTM(2, 50, batch_first=True).to(dev)


def chain():
    h = LstmSequence.apply(Y, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
    if a.q_per_step:
        rec, offs, _ = mix_dynamics(alpha, [A, Bm, Qk])
        slots, Cop, Qop = Slots(A=offs[0], B=offs[1], Q=offs[2]), Cm[0], None
    else:
        rec, offs, _ = mix_dynamics(alpha, [A, Bm, Cm])
        slots, Cop, Qop = Slots(A=offs[0], B=offs[1], C=offs[2]), None, Q
    ms, Ss, *_ = LgssmSmooth.apply(Y, U, None, rec, None, None, Cop, Qop, R, mu0, S0, slots, True)
    total, _, levels_dev = LgssmElbo.apply(ms, Ss, eps, Y, U, None, rec, None, None, Cop, Qop, R, mu0, S0, slots)
    (total / (B * T) + 1e-3 * h.sum()).backward()


for _ in range(3):
    chain()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    chain()
torch.cuda.synchronize()
print(f"lgssm chain B={B} T={T} n={n}: {(time.perf_counter() - t0) / a.iters * 1e3:.3f} ms/iter (eager, incl. launch gaps)")
