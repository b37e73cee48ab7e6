"""Step-by-step GPU diagnostic: each stage prints before/after a device sync so a fault is attributable."""
import sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "kalman-vae_amd"), str(ROOT / "tests")]
import torch
from kvae.kalman.lgssm_ops import LgssmSmooth, LgssmElbo, Slots, mix_dynamics
from parity_cases import _random_problem
from oracle import c_oracle
from golden_util import rel_err
DEV = "cuda"
print(torch.cuda.get_device_name(0), flush=True)

def stage(name, fn):
    print(f"[{name}] start", flush=True)
    out = fn()
    torch.cuda.synchronize()
    print(f"[{name}] ok", flush=True)
    return out

stages = sys.argv[1:] or ["mix", "fwd_generic", "fwd_static", "elbo", "bwd"]
for (n, m, p, tag) in ((3, 2, 1, "generic"), (4, 4, 2, "static")):
    B, T, K = 2, 6, 2
    A, Bm, Cm, alpha, Y, U, mask, eps = _random_problem(B, T, n, m, p, K, 1, DEV)
    R, Q = 0.03 * torch.eye(p, device=DEV), 0.02 * torch.eye(n, device=DEV)
    mu0, S0 = torch.zeros(n, device=DEV), 20.0 * torch.eye(n, device=DEV)
    leaves = [t.clone().requires_grad_(True) for t in (A, Bm, Cm, alpha, Y, U)]
    rec, offs, (As, Bs, Cs) = stage(f"mix_{tag}", lambda: mix_dynamics(leaves[3], leaves[:3]))
    slots = Slots(A=offs[0], B=offs[1], C=offs[2])
    outs = stage(f"smooth_fwd_{tag}", lambda: LgssmSmooth.apply(leaves[4], leaves[5], mask, rec, None, None, None, Q, R, mu0, S0, slots, True))
    c = lambda t: t.detach().cpu()
    ref = c_oracle.smooth(c(Y), c(U), c(mask), c(As), c(Bs), c(Cs), c(Q), c(R), c(mu0), c(S0))
    print("   mus_smooth rel", rel_err(c(outs[0]), ref["mus_smooth"]), "Sig_smooth rel", rel_err(c(outs[1]), ref["Sigmas_smooth"]), flush=True)
    total, terms, _lv = stage(f"elbo_{tag}", lambda: LgssmElbo.apply(outs[0], outs[1], eps, leaves[4], leaves[5], mask, rec, None, None, None, Q, R, mu0, S0, slots))
    rt, lv = c_oracle.elbo_terms(c(outs[0]), c(outs[1]), c(eps), c(Y), c(U), c(mask), c(As), c(Bs), c(Cs), c(Q), c(R), c(mu0), c(S0))
    print("   terms", terms.tolist(), "oracle", rt.tolist(), flush=True)
    stage(f"backward_{tag}", lambda: total.backward())
    print("   grad alpha norm", float(leaves[3].grad.norm()), "grad Y norm", float(leaves[4].grad.norm()), flush=True)
print("ALL OK", flush=True)
