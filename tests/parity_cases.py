"""Device-agnostic parity cases: run against the host simulation (CPU tier) and the gfx950 library (GPU tier)."""
import math

import torch

from golden_util import rel_err

def _random_problem(B, T, n, m, p, K, seed, device):
    g = torch.Generator().manual_seed(seed)
    A = torch.eye(n).repeat(K, 1, 1) + 0.08 * torch.randn(K, n, n, generator=g)
    Bm = 0.1 * torch.randn(K, n, m, generator=g)
    Cm = 0.3 * torch.randn(K, p, n, generator=g)
    alpha = torch.softmax(torch.randn(B, T, K, generator=g), -1)
    Y = torch.randn(B, T, p, generator=g)
    U = 0.3 * torch.randn(B, T, m, generator=g)
    mask = (torch.rand(B, T, generator=g) > 0.2).float()
    eps = torch.randn(B, T, n, generator=g)
    return [t.to(device) for t in (A, Bm, Cm, alpha, Y, U, mask, eps)]


def vs_oracle_random(DEV, B, T, n, m, p, K, dense_q=False):
    """HIP path vs the CPU oracle (C restatement for values, torch oracle autograd for gradients) on seeded
    inputs, incl. BASELINE configs[1] (B=256,T=50,n=4) and a configs[4] shard slice (n=16,T=200); ragged
    sizes, masks, controls, T=1 and run-time dimensions.  dense_q: a full SPD process noise shared by the batch (the default
    0.02 I cannot tell a factor from its transpose - the n = 16 ELBO keeps chol(Q)^-1 AND its transpose on the lanes)."""
    from kvae.kalman.lgssm_ops import LgssmElbo, LgssmSmooth, mix_dynamics
    from oracle import c_oracle
    from oracle import torch_oracle as O
    A, Bm, Cm, alpha, Y, U, mask, eps = _random_problem(B, T, n, m, p, K, 100 + B + T, DEV)
    R = 0.03 * torch.eye(p, device=DEV)
    Q = 0.02 * torch.eye(n, device=DEV)
    if dense_q:
        Wq = torch.randn(n, n, generator=torch.Generator().manual_seed(7)).to(DEV)
        Q = Q + 0.004 * (Wq @ Wq.T)
    mu0, S0 = torch.zeros(n, device=DEV), 20.0 * torch.eye(n, device=DEV)
    leaves = [t.clone().requires_grad_(True) for t in (A, Bm, Cm, alpha, Y, U)]
    rec, offs, (As, Bs, Cs) = mix_dynamics(leaves[3], leaves[:3])
    from kvae.kalman.lgssm_ops import Slots
    slots = Slots(A=offs[0], B=offs[1], C=offs[2])
    ms, Ss, mf, Sf, mp, Sp = LgssmSmooth.apply(leaves[4], leaves[5], mask, rec, None, None, None, Q, R, mu0, S0, slots, True)
    total, terms, levels_dev = LgssmElbo.apply(ms, Ss, eps, leaves[4], leaves[5], mask, rec, None, None, None, Q, R, mu0, S0, slots)
    # values: C oracle
    c = lambda t: t.detach().cpu()
    ref = c_oracle.smooth(c(Y), c(U), c(mask), c(As), c(Bs), c(Cs), c(Q), c(R), c(mu0), c(S0))
    # n = 16: two float32 implementations of a 200-step recursion with different summation orders (matrix cores vs scalar
    # loops) can sit 1e-3 apart while both are equally close to the truth; this first bar only catches gross errors there, the
    # real one follows below - against a FLOAT64 run, within max(1e-4, 2 x the float32 oracle's own distance from it).
    tol = 2e-4 if n < 16 else 2e-3
    for k, v in (("mus_smooth", ms), ("Sigmas_smooth", Ss), ("mus_filt", mf), ("Sigmas_filt", Sf), ("mus_pred", mp),
                 ("Sigmas_pred", Sp)):
        assert rel_err(c(v), ref[k]) < tol, k
    rterms, levels = c_oracle.elbo_terms(c(ms), c(Ss), c(eps), c(Y), c(U), c(mask), c(As), c(Bs), c(Cs), c(Q), c(R),
                                         c(mu0), c(S0))
    assert list(levels) == [0, 0]
    for i in range(4):
        assert abs(float(terms[i]) - rterms[i]) <= 2e-4 * abs(rterms[i]) + 1e-3, (i, float(terms[i]), rterms[i])
    if B * T * n * n > 450000:
        return  # gradients up to configs[1] size (256,50,4) and (8,200,16); beyond, the torch-oracle tape takes minutes
    # gradients: autograd over the torch oracle
    (total / (B * T)).backward()
    v32 = {}
    want = _torch_oracle_grads((A, Bm, Cm, alpha, Y, U, mask, eps), Q, R, mu0, S0, torch.float32, values=v32)
    for name, got, ref_g in zip("A B C alpha Y U".split(), leaves, want):
        assert rel_err(got.grad.cpu(), ref_g) < 3e-3, name
    if n >= 16:   # the value bar where float32 is the limit (see above)
        v64 = {}
        _torch_oracle_grads((A, Bm, Cm, alpha, Y, U, mask, eps), Q, R, mu0, S0, torch.float64, values=v64)
        for k, got in (("mus_smooth", ms), ("Sigmas_smooth", Ss), ("elbo_sum", total)):
            tape, mine = rel_err(v32[k].double(), v64[k]), rel_err(c(got).double(), v64[k])
            assert mine < max(1e-4, 4.0 * tape), (k, mine, tape)   # (4: see values_vs_fp64_oracle)


def _torch_oracle_grads(problem, Q, R, mu0, S0, dtype, values=None):
    """d(ELBO / (B T)) / d(A, B, C, alpha, Y, U) by autograd over oracle/torch_oracle.py, computed in `dtype` on the CPU.
    values: a dict that receives the forward results (mus_smooth [B,T,n], Sigmas_smooth, elbo_sum)."""
    from oracle import torch_oracle as O
    c = lambda t: t.detach().cpu().to(dtype)
    A, Bm, Cm, alpha, Y, U, mask, eps = problem
    B, T = Y.shape[:2]
    cl = [c(t).clone().requires_grad_(True) for t in (A, Bm, Cm, alpha, Y, U)]
    Ar = torch.einsum("btk,kij->btij", cl[3], cl[0])
    Br = torch.einsum("btk,kij->btij", cl[3], cl[1])
    Cr = torch.einsum("btk,kij->btij", cl[3], cl[2])
    mu, Sig = c(mu0).expand(B, -1).unsqueeze(-1), c(S0).expand(B, -1, -1)
    mfs, Sfs, mps, Sps = [], [], [], []
    for t in range(T):
        mu, Sig, mu_p, Sig_p = O.filter_step(mu, Sig, cl[4][:, t], cl[5][:, t], Ar[:, t], Br[:, t], Cr[:, t], c(Q), c(R),
                                             c(mask)[:, t])
        mfs.append(mu), Sfs.append(Sig), mps.append(mu_p), Sps.append(Sig_p)
    mus, Sigs = [mfs[-1]], [Sfs[-1]]
    for t in range(T - 2, -1, -1):
        m_s, S_s = O.smooth_step(Sfs[t], Sps[t + 1], Sigs[0], mfs[t], mps[t + 1], mus[0], Ar[:, t + 1])
        mus.insert(0, m_s), Sigs.insert(0, S_s)
    tr, em, ini, ent = O.lgssm_elbo_terms(torch.stack(mus, 1), torch.stack(Sigs, 1), cl[4], cl[5], Ar, Br, Cr, c(Q), c(R),
                                          c(mu0), c(S0), c(mask), c(eps))
    if values is not None:
        values.update(mus_smooth=torch.stack(mus, 1).detach().squeeze(-1), Sigmas_smooth=torch.stack(Sigs, 1).detach(),
                      elbo_sum=(tr + em + ini + ent).detach())
    ((tr + em + ini + ent) / (B * T)).backward()
    return [t.grad for t in cl]


def values_vs_fp64_oracle(DEV, B, T, n, K):
    """north_star asks for ELBO and smoothed means within 1e-4 of the reference's CPU path.  At n = 16 over T = 200 float32 itself
    cannot promise that (the reference's own float32 run of stress_switch_z16_B2_T200 is 3.4e-4 from a float64 run of the same
    recursion), so here the float64 oracle is the truth and the bar is written as an assertion instead of a flat 2e-3: the HIP
    smoothed means, smoothed covariances and ELBO must lie within max(1e-4, 4 x the float32 oracle's own distance) of it.
    (Four, not two: that distance is ONE sample of amplified rounding - the same oracle lands at 0.8e-4 on one host CPU and at
    3.4e-4 on another for the same fixture - so the factor has to cover the spread between summation orders.)"""
    from kvae.kalman.lgssm_ops import LgssmElbo, LgssmSmooth, Slots, mix_dynamics
    problem = _random_problem(B, T, n, n, 2, K, 100 + B + T, DEV)
    A, Bm, Cm, alpha, Y, U, mask, eps = problem
    R, Q = 0.03 * torch.eye(2, device=DEV), 0.02 * torch.eye(n, device=DEV)
    mu0, S0 = torch.zeros(n, device=DEV), 20.0 * torch.eye(n, device=DEV)
    with torch.no_grad():
        rec, offs, _ = mix_dynamics(alpha, [A, Bm, Cm])
        slots = Slots(A=offs[0], B=offs[1], C=offs[2])
        ms, Ss, *_ = LgssmSmooth.apply(Y, U, mask, rec, None, None, None, Q, R, mu0, S0, slots, True)
        total, _, _ = LgssmElbo.apply(ms, Ss, eps, Y, U, mask, rec, None, None, None, Q, R, mu0, S0, slots)
    v64, v32 = {}, {}
    _torch_oracle_grads(problem, Q, R, mu0, S0, torch.float64, values=v64)
    _torch_oracle_grads(problem, Q, R, mu0, S0, torch.float32, values=v32)
    report = {}
    for k, got in (("mus_smooth", ms), ("Sigmas_smooth", Ss), ("elbo_sum", total)):
        tape = rel_err(v32[k].double(), v64[k])
        mine = rel_err(got.cpu().double(), v64[k])
        report[k] = (mine, tape)
        assert mine < max(1e-4, 4.0 * tape), (k, mine, tape)
    return report


_FP64_CACHE = {}


def latent_fp64_budget(g, kind, name):
    """Float64 and float32 runs of the torch oracle on a latent golden's inputs: (outputs of the float64 run, per-key distance of
    the float32 run from it).  The fixture itself is a float32 run of the REFERENCE, so it sits at the same distance."""
    if name in _FP64_CACHE:
        return _FP64_CACHE[name]
    from golden_util import sub
    from oracle import torch_oracle as O

    def run(dt):
        c = lambda t: t.to(dt) if t.is_floating_point() else t
        dyn = {k: c(v) for k, v in sub(g, "dyn.").items()}
        kw = {}
        if kind == "switching":
            kw = dict(tau=float(g["tau"]), is_training=bool(g["train"]), gumbel=c(g["gumbel"]), trans_matrix=c(g["trans_matrix"]))
        with torch.no_grad():
            return O.smooth_and_elbo(dyn, kind, c(g["a"]), c(g["u"]), c(g["mask"]), c(g["Qbuf"]), c(g["R"]), c(g["mu0"]),
                                     c(g["Sigma0"]), c(g["eps_z"]), **kw)

    o64, o32 = run(torch.float64), run(torch.float32)
    dist = {k: rel_err(o32[k].double(), o64[k]) for k in o64 if torch.is_tensor(o64[k]) and o64[k].is_floating_point()}
    # the REFERENCE's own float32 run is the fixture: its distance from float64 is data, the same on every box (the oracle's
    # float32 run depends on the host's BLAS: 0.8e-4 on one CPU, 3.4e-4 on another for the same inputs) - the budget is the
    # larger of the two
    for k in list(dist):
        if k in g and g[k].shape == o64[k].shape:
            dist[k] = max(dist[k], rel_err(g[k].double(), o64[k]))
        elif k + "_every8" in g:
            dist[k] = max(dist[k], rel_err(g[k + "_every8"].double(), o64[k][:, ::8]))
    _FP64_CACHE[name] = (o64, dist)
    return o64, dist


def grads_vs_fp64_oracle(DEV, B, T, n, K):
    """Where float32 itself is the limit.  A SINGLE dynamics mode at n = 16 (every step multiplies by the same A, prior 20 I against
    R = 0.03 I) leaves the float32 torch oracle 0.8-1.3e-2 away from its own float64 run on the gradients of B and U, so the 3e-3
    bar of vs_oracle_random against the float32 tape would measure the tape.  Here the float64 oracle is the truth and the HIP path
    must stay within the larger of that bar and four times the float32 oracle's own distance from it."""
    from kvae.kalman.lgssm_ops import LgssmElbo, LgssmSmooth, Slots, mix_dynamics
    problem = _random_problem(B, T, n, n, 2, K, 100 + B + T, DEV)
    A, Bm, Cm, alpha, Y, U, mask, eps = problem
    R, Q = 0.03 * torch.eye(2, device=DEV), 0.02 * torch.eye(n, device=DEV)
    mu0, S0 = torch.zeros(n, device=DEV), 20.0 * torch.eye(n, device=DEV)
    leaves = [t.clone().requires_grad_(True) for t in (A, Bm, Cm, alpha, Y, U)]
    rec, offs, _ = mix_dynamics(leaves[3], leaves[:3])
    slots = Slots(A=offs[0], B=offs[1], C=offs[2])
    ms, Ss, *_ = LgssmSmooth.apply(leaves[4], leaves[5], mask, rec, None, None, None, Q, R, mu0, S0, slots, True)
    total, _, levels_dev = LgssmElbo.apply(ms, Ss, eps, leaves[4], leaves[5], mask, rec, None, None, None, Q, R, mu0, S0, slots)
    (total / (B * T)).backward()
    g64 = _torch_oracle_grads(problem, Q, R, mu0, S0, torch.float64)
    g32 = _torch_oracle_grads(problem, Q, R, mu0, S0, torch.float32)
    worst = 0.0
    for name, got, w64, w32 in zip("A B C alpha Y U".split(), leaves, g64, g32):
        if float(w64.abs().max()) == 0.0:  # K = 1: softmax over one mode is constant
            assert float(got.grad.abs().max()) < 1e-6, name
            continue
        tape = rel_err(w32.double(), w64)
        mine = rel_err(got.grad.cpu().double(), w64)
        worst = max(worst, tape)
        assert mine < max(3e-3, 4.0 * tape), (name, mine, tape)
    return worst


def unaligned_fallback(DEV, n, m, p, B=3, T=9):
    """Operands that do NOT start on 16-byte boundaries take the run-time-dimension kernels instead of the specialised
    ones ((16,16,2): matrix cores; (4,4,2): sixteen sequences per wavefront - both move rows with 16-byte accesses):
    same values and same gradients either way.  B = 19 at n = 4 also leaves the last wavefront's quads ragged."""
    from kvae.kalman.lgssm_ops import LgssmSmooth, Slots
    A, Bm, Cm, alpha, Y, U, mask, _ = _random_problem(B, T, n, m, p, 1, 21, DEV)
    R, Q = 0.03 * torch.eye(p, device=DEV), 0.02 * torch.eye(n, device=DEV)
    mu0, S0 = torch.zeros(n, device=DEV), 20.0 * torch.eye(n, device=DEV)
    Astack = A[0].expand(B, T, n, n).contiguous()
    odd = torch.empty(B * T * n * n + 1, device=DEV)[1:].view(B, T, n, n)   # same values, 4 bytes off a 16-byte boundary
    odd.copy_(Astack)
    assert odd.data_ptr() % 16 != 0 and Astack.data_ptr() % 16 == 0
    res = []
    for stack in (Astack, odd):
        leaf = stack.clone().requires_grad_(True) if stack is Astack else None
        if leaf is None:                      # .clone() would re-align it: take the gradient w.r.t. a view's base instead
            base = torch.empty(B * T * n * n + 1, device=DEV)
            base[1:].copy_(stack.reshape(-1))
            base.requires_grad_(True)
            leaf_in = base[1:].view(B, T, n, n)
            assert leaf_in.data_ptr() % 16 != 0
        else:
            base, leaf_in = leaf, leaf
        outs = LgssmSmooth.apply(Y, U, mask, None, leaf_in, Bm[0], Cm[0], Q, R, mu0, S0, Slots(), True)
        gen = torch.Generator().manual_seed(5)
        loss = sum((o * torch.randn(o.shape, generator=gen).to(DEV)).sum() for o in outs)
        loss.backward()
        gA = base.grad if base is leaf else base.grad[1:].view(B, T, n, n)
        res.append(([o.detach().cpu() for o in outs], gA.cpu()))
    (fast, gfast), (slow, gslow) = res
    for a, b in zip(fast, slow):
        assert rel_err(a, b) < 2e-4
    assert rel_err(gfast, gslow) < 5e-4


def n16_generic_fallback(DEV):
    unaligned_fallback(DEV, 16, 16, 2)


def n4_generic_fallback(DEV):
    unaligned_fallback(DEV, 4, 4, 2, B=19, T=11)


def n16_indefinite_q(DEV, B=3, T=12):
    """(16,16,2) with a process noise that is NOT positive semi-definite (the reference's stability recipe does this at
    n = 4): predicted covariances go indefinite, the natural-order elimination of the smoother gain meets a non-positive
    pivot and the kernels must fall back to the partially pivoted one (getrf's pivot sequence, which is what the C
    oracle's LU and torch.linalg.solve do).  Values vs the C oracle, gradients vs the torch oracle."""
    from kvae.kalman.lgssm_ops import LgssmSmooth, Slots
    from oracle import c_oracle
    from oracle import torch_oracle as O
    n, m, p = 16, 16, 2
    g = torch.Generator().manual_seed(77)
    A = 0.6 * torch.eye(n) + 0.15 * torch.randn(n, n, generator=g)
    Bm = 0.1 * torch.randn(n, m, generator=g)
    Cm = 0.3 * torch.randn(p, n, generator=g)
    Qh = 0.3 * torch.randn(n, n, generator=g)
    Q = 0.5 * (Qh + Qh.T)                                   # symmetric, indefinite
    assert float(torch.linalg.eigvalsh(Q).min()) < -0.1
    Y, U = torch.randn(B, T, p, generator=g), 0.3 * torch.randn(B, T, m, generator=g)
    R, mu0, S0 = 0.03 * torch.eye(p), torch.zeros(n), 0.5 * torch.eye(n)
    dev = lambda t: t.to(DEV)
    leaves = [dev(t).clone().requires_grad_(True) for t in (A, Bm, Cm, Y)]
    outs = LgssmSmooth.apply(leaves[3], dev(U), None, None, leaves[0], leaves[1], leaves[2], dev(Q), dev(R), dev(mu0), dev(S0),
                             Slots(), True)
    ref = c_oracle.smooth(Y, U, None, A, Bm, Cm, Q, R, mu0, S0)
    assert float(torch.linalg.eigvalsh(0.5 * (ref["Sigmas_pred"][0, -1] + ref["Sigmas_pred"][0, -1].T)).min()) < 0   # really indefinite
    for k, v in zip(("mus_smooth", "Sigmas_smooth", "mus_filt", "Sigmas_filt", "mus_pred", "Sigmas_pred"), outs):
        assert rel_err(v.detach().cpu(), ref[k]) < 2e-3, k
    w = [torch.randn(o.shape, generator=g) for o in outs[:2]]
    (sum((o * dev(wi)).sum() for o, wi in zip(outs[:2], w))).backward()
    def oracle_grads(dt):
        c = lambda t: t.to(dt)
        cl = [c(t).clone().requires_grad_(True) for t in (A, Bm, Cm, Y)]
        mu, Sig = c(mu0).expand(B, -1).unsqueeze(-1), c(S0).expand(B, -1, -1)
        ex = lambda M: M.expand(B, -1, -1)
        mfs, Sfs, mps, Sps = [], [], [], []
        for t in range(T):
            mu, Sig, mu_p, Sig_p = O.filter_step(mu, Sig, cl[3][:, t], c(U)[:, t], ex(cl[0]), ex(cl[1]), ex(cl[2]), c(Q), c(R),
                                                 torch.ones(B, dtype=dt))
            mfs.append(mu), Sfs.append(Sig), mps.append(mu_p), Sps.append(Sig_p)
        mus, Sigs = [mfs[-1]], [Sfs[-1]]
        for t in range(T - 2, -1, -1):
            m_s, S_s = O.smooth_step(Sfs[t], Sps[t + 1], Sigs[0], mfs[t], mps[t + 1], mus[0], ex(cl[0]))
            mus.insert(0, m_s), Sigs.insert(0, S_s)
        ((torch.stack(mus, 1).squeeze(-1) * c(w[0])).sum() + (torch.stack(Sigs, 1) * c(w[1])).sum()).backward()
        return [t.grad for t in cl]
    # Indefinite predicted covariances amplify rounding inside the recursion (the INPUT condition number is modest: an fp32-ulp
    # perturbation of the inputs moves the fp64 gradients by 1e-5).  Measured against the fp64 oracle: the fp32 oracle sits 7e-4
    # (A) and < 2.5e-3 (C) away; the kernels 3e-3 .. 6e-3 (A) and 6e-3 .. 1.4e-2 (C), and a recompilation that only
    # re-associates FMAs moves them inside those ranges.  This case pins the choice of path and sane values on a matrix no
    # Kalman filter should meet; the accuracy class of the kernels is pinned by latent_fp64_budget on the reference's own cases.
    g64, g32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
    for name, got, w64, w32 in zip("A B C Y".split(), leaves, g64, g32):
        budget = max(4.0 * rel_err(w32.double(), w64), 2.5e-2)
        assert rel_err(got.grad.cpu().double(), w64) < budget, (name, budget)


def linearity(DEV, B=256, T=50):
    """Size-independent property at configs[1] size: the smoothed MEANS are linear in (y, u, mu0) for fixed
    dynamics, and the covariances do not depend on y at all."""
    from kvae.kalman.lgssm_ops import LgssmSmooth, Slots
    n, m, p = 4, 4, 2
    A, Bm, Cm, alpha, Y, U, mask, _ = _random_problem(B, T, n, m, p, 1, 7, DEV)
    R, Q = 0.03 * torch.eye(p, device=DEV), 0.02 * torch.eye(n, device=DEV)
    mu0, S0 = torch.zeros(n, device=DEV), 20.0 * torch.eye(n, device=DEV)
    run = lambda y, u: LgssmSmooth.apply(y, u, mask, None, A[0], Bm[0], Cm[0], Q, R, mu0, S0, Slots(), True)
    with torch.no_grad():
        o1, o2, o12 = run(Y, U), run(2 * Y.flip(0), -U), run(Y + 2 * Y.flip(0), U - U)
    assert rel_err(o12[0], o1[0] + o2[0]) < 2e-4
    assert rel_err(o12[1], o1[1]) < 1e-5 and rel_err(o2[1], o1[1]) < 1e-5



def safe_cholesky_levels(DEV, n=4, B=2, T=4):
    """_safe_cholesky semantics: one bad Q_t forces the WHOLE batch up the jitter ladder / to the diagonal
    fallback, exactly like the oracle (kalman_filter.py:282-302).  n = 16: the matrix-core ELBO kernels resolve the
    level in their probe and hand levels > 0 to the generic main kernel on the device; (B, T) = (40, 10): more steps than
    one round of wavefronts of the four-steps-per-wavefront launch, ragged."""
    from kvae.kalman.lgssm_ops import LgssmElbo, Slots
    from oracle import c_oracle
    m, p = n, 2
    A, Bm, Cm, alpha, Y, U, mask, eps = _random_problem(B, T, n, m, p, 1, 5, DEV)
    R = 0.03 * torch.eye(p, device=DEV)
    mu0, S0 = torch.zeros(n, device=DEV), 20.0 * torch.eye(n, device=DEV)
    mus = torch.randn(B, T, n, generator=torch.Generator().manual_seed(1)).to(DEV)
    Sig = (0.5 * torch.eye(n, device=DEV)).expand(B, T, n, n).contiguous()
    c = lambda t: t.cpu()
    for q00, want in ((0.02, 0), (-3e-6, 1), (-1.0, 5)):
        Q = (0.02 * torch.eye(n, device=DEV)).expand(B, T, n, n).contiguous()
        Q[1, 2, 0, 0] = q00
        total, terms, levels_dev = LgssmElbo.apply(mus, Sig, eps, Y, U, mask, None, A[0], Bm[0], Cm[0], Q, R, mu0, S0, Slots())
        rterms, levels = c_oracle.elbo_terms(c(mus), c(Sig), c(eps), c(Y), c(U), c(mask), c(A[0]), c(Bm[0]), c(Cm[0]), c(Q),
                                             c(R), c(mu0), c(S0))
        assert levels[1] == want, levels
        assert levels_dev.cpu().tolist()[:2] == list(levels[:2])
        if str(DEV).startswith("cuda"):   # which kernel family computed it: thread-per-step at n = 4, matrix cores at n = 16
            assert int(levels_dev[2]) in {4: (0, 1), 16: (2,)}.get(n, (0,)), (n, int(levels_dev[2]))
        for i in range(4):
            assert abs(float(terms[i]) - rterms[i]) <= 3e-4 * abs(rterms[i]) + 1e-3, (i, float(terms[i]), rterms[i])


def safe_cholesky_levels_shared_q(DEV, n=16, B=5, T=10):
    """The same ladder with ONE Q for the whole batch (lstm dynamics: a broadcast [n,n] operand) - at n = 16 the four-steps-per-
    wavefront kernels - and with the raised level coming from a smoothed covariance, so that the parked z_t have to be redone
    at the resolved level: (level of Sigma_s, level of Q) in {(0,0), (0,1), (3,0), (5,0), (2,5)}; values and every gradient
    against the C oracle / the torch oracle's autograd on the expanded stacks."""
    from kvae.kalman.lgssm_ops import LgssmElbo, Slots
    from oracle import c_oracle
    from oracle import torch_oracle as O
    m, p = n, 2
    A, Bm, Cm, alpha, Y, U, mask, eps = _random_problem(B, T, n, m, p, 1, 7, DEV)
    R = 0.03 * torch.eye(p, device=DEV)
    mu0, S0 = torch.zeros(n, device=DEV), 20.0 * torch.eye(n, device=DEV)
    gen = torch.Generator().manual_seed(2)
    mus = torch.randn(B, T, n, generator=gen).to(DEV)
    W = torch.randn(B, T, n, n, generator=gen) * 0.1
    Sig0 = (W @ W.mT + 0.3 * torch.eye(n)).to(DEV)
    c = lambda t: t.detach().cpu()
    for sig_bad, q_bad, want in ((None, None, [0, 0]), (None, -3e-6, [0, 1]), (-4e-4, None, [3, 0]), (-1.0, None, [5, 0]),
                                 (-5e-5, -1.0, [2, 5])):
        Sig = Sig0.clone()
        if sig_bad is not None:
            Sig[B - 1, T - 3] = torch.diag(torch.tensor([sig_bad] + [0.2] * (n - 1))).to(DEV)
        Q = 0.02 * torch.eye(n, device=DEV)
        if q_bad is not None:
            Q[2, 2] = q_bad
        leaves = [t.clone().requires_grad_(True) for t in (mus, Sig, Y, A[0], Bm[0], Cm[0], Q)]
        total, terms, levels_dev = LgssmElbo.apply(leaves[0], leaves[1], eps, leaves[2], U, mask, None, leaves[3], leaves[4], leaves[5],
                                                   leaves[6], R, mu0, S0, Slots())
        assert levels_dev.cpu().tolist()[:2] == want, (levels_dev.cpu().tolist(), want)
        if str(DEV).startswith("cuda") and n == 16:
            assert int(levels_dev[2]) == 2, int(levels_dev[2])   # gQ is wanted here: one step per wavefront
        ex = lambda M: c(M).expand(B, T, *M.shape).contiguous()
        rterms, levels = c_oracle.elbo_terms(c(mus), c(Sig), c(eps), c(Y), c(U), c(mask), c(A[0]), c(Bm[0]), c(Cm[0]), ex(Q),
                                             c(R), c(mu0), c(S0))
        assert list(levels[:2]) == want
        for i in range(4):
            assert abs(float(terms[i]) - rterms[i]) <= 3e-4 * abs(rterms[i]) + 1e-3, (want, i, float(terms[i]), rterms[i])
        # gradients vs the torch oracle's autograd on the same problem
        ref = [c(t).clone().requires_grad_(True) for t in (mus, Sig, Y, A[0], Bm[0], Cm[0], Q)]
        e = lambda M: M.expand(B, T, *M.shape)
        want_total = O.lgssm_elbo(ref[0].unsqueeze(-1), ref[1], ref[2], c(U), e(ref[3]), e(ref[4]), e(ref[5]), e(ref[6]), c(R), c(mu0),
                                  c(S0), c(mask), c(eps)) * c(mask).sum().clamp(min=1.0)
        assert rel_err(total.detach().cpu(), want_total.detach()) < 1e-4, want
        total.backward()
        want_total.backward()
        for k, (got, r) in enumerate(zip(leaves, ref)):
            if r.grad is None or float(r.grad.abs().max()) == 0.0:
                assert got.grad is None or float(got.grad.abs().max()) < 1e-6, (want, k)
            else:
                assert rel_err(got.grad.cpu(), r.grad) < 3e-3, (want, k)
        # without gradients of Q the shared-Q call takes the four-steps-per-wavefront kernels (family 3): same value
        with torch.no_grad():
            tot4, _, lv4 = LgssmElbo.apply(mus, Sig, eps, Y, U, mask, None, A[0], Bm[0], Cm[0], Q, R, mu0, S0, Slots())
        assert lv4.cpu().tolist()[:2] == want and rel_err(tot4.cpu(), total.detach().cpu()) < 1e-5
        if str(DEV).startswith("cuda") and n == 16:
            assert int(lv4[2]) == 3, int(lv4[2])
        # ... and with gradients w.r.t. everything but Q (the training step of the lstm model, whose Q is a buffer)
        lv2 = [t.clone().requires_grad_(True) for t in (mus, Sig, Y, A[0], Bm[0], Cm[0])]
        tot5, _, lv5 = LgssmElbo.apply(lv2[0], lv2[1], eps, lv2[2], U, mask, None, lv2[3], lv2[4], lv2[5], Q, R, mu0, S0, Slots())
        tot5.backward()
        if str(DEV).startswith("cuda") and n == 16:
            assert int(lv5[2]) == 3, int(lv5[2])
        for k, (got, r) in enumerate(zip(lv2, ref[:6])):
            if r.grad is not None and float(r.grad.abs().max()) > 0.0:
                assert rel_err(got.grad.cpu(), r.grad) < 3e-3, (want, k, "four-step")


def jitter_golden(DEV, name, levels):
    """The product's ELBO (probe + atomicMax level + terms + gradients, csrc/lgssm_elbo.h) against fixtures that drive the
    REFERENCE's own elbo past level 0 of _safe_cholesky (tests/golden/make_goldens_r2.py): resolved levels, value
    (1e-4 rel, north_star) and every gradient the reference's autograd produced (3e-3 rel)."""
    from golden_util import JITTER_GRADS, load
    from kvae.kalman.lgssm_ops import LgssmElbo, Slots
    g = load(name)
    d = {k: v.to(DEV) for k, v in g.items()}
    leaves = {k: d[k].clone().requires_grad_(True) for k in JITTER_GRADS}
    total, _, levels_dev = LgssmElbo.apply(leaves["mu_s"], leaves["Sig_s"], d["eps_z"], leaves["a"], d["u"], d["mask"], None,
                               leaves["A_list"], leaves["B_list"], leaves["C_list"], leaves["Q_list"], d["R"], d["mu0"],
                               d["Sigma0"], Slots())
    assert levels_dev.cpu().tolist()[:2] == levels
    if str(DEV).startswith("cuda") and d["Sig_s"].shape[-1] == 16:
        # the (16,16,2) matrix-core kernels computed this call at a RAISED level themselves (no generic backup launch):
        # family 2 = one step per wavefront (a per-step Q_list)
        assert int(levels_dev[2]) == 2, int(levels_dev[2])
    elbo = total / d["mask"].sum().clamp(min=1.0)
    assert rel_err(elbo.detach().cpu(), g["elbo"]) < 1e-4
    (-elbo).backward()
    for k, leaf in leaves.items():
        assert rel_err(leaf.grad.cpu(), g["grad." + k]) < 3e-3, k


def lstm_vs_torch(DEV, B, T, I, H):
    """kvae_lstm_fwd/bwd vs torch.nn.LSTM on the CPU (values and all gradients)."""
    from kvae.kalman.lgssm_ops import LstmSequence
    torch.manual_seed(B * 100 + T)
    ref = torch.nn.LSTM(I, H, batch_first=True)
    x = torch.randn(B, T, I)
    w = torch.randn(B, T, H)
    xr = x.clone().requires_grad_(True)
    (ref(xr)[0] * w).sum().backward()
    params = [p.detach().clone().to(DEV).requires_grad_(True) for p in (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0,
                                                                        ref.bias_hh_l0)]
    xd = x.clone().to(DEV).requires_grad_(True)
    h = LstmSequence.apply(xd, *params)
    (h * w.to(DEV)).sum().backward()
    assert rel_err(h.detach().cpu(), ref(x)[0].detach()) < 2e-5
    assert rel_err(xd.grad.cpu(), xr.grad) < 1e-4
    for got, want in zip(params, (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)):
        assert rel_err(got.grad.cpu(), want.grad) < 1e-4


def vae_epilogue_vs_torch(DEV, N, C, H, W, r, relu):
    """kvae_bias_shuffle_act_fwd/bwd vs conv-bias + nn.PixelShuffle + nn.ReLU of torch (values and gradients)."""
    import torch.nn.functional as F
    from kvae.vae.fused import BiasShuffleAct
    g = torch.Generator().manual_seed(N + C + H)
    x = torch.randn(N, C * r * r, H, W, generator=g)
    b = torch.randn(C * r * r, generator=g)
    w = torch.randn(N, C, H * r, W * r, generator=g)
    xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.pixel_shuffle(xr + br.view(1, -1, 1, 1), r) if r > 1 else xr + br.view(1, -1, 1, 1)
    ref = F.relu(ref) if relu else ref
    (ref * w).sum().backward()
    xd, bd = x.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
    out = BiasShuffleAct.apply(xd, bd, r, relu)
    (out * w.to(DEV)).sum().backward()
    assert torch.equal(out.detach().cpu(), ref.detach())
    assert torch.equal(xd.grad.cpu(), xr.grad)
    assert rel_err(bd.grad.cpu(), br.grad) < 1e-5


def regime_vs_torch(DEV, B, T, K, tau, hard):
    """kvae_regime_fwd/bwd vs the step-by-step torch restatement of switch_dyn_param.py:52-79 (values and grads)."""
    from kvae.kalman.lgssm_ops import RegimeChain
    from kvae.kalman.switch_dyn_param import StickyRegimePrior, SwitchingDynamicsParameter
    g = torch.Generator().manual_seed(B + T + K)
    logits = torch.randn(B, T, K, K, generator=g)
    init = torch.randn(B, K, generator=g)
    gum = -torch.empty(B, T, K).exponential_(generator=g).log()
    w_y, w_q, w_p = torch.randn(B, T, K, generator=g), torch.randn(B, T, generator=g), torch.randn(B, T, generator=g)
    dyn = SwitchingDynamicsParameter(torch.eye(4).repeat(K, 1, 1), torch.zeros(K, 4, 4), torch.zeros(K, 2, 4),
                                     prior=StickyRegimePrior(K, 0.8))
    dyn.tau = tau
    lr, ir = logits.clone().requires_grad_(True), init.clone().requires_grad_(True)
    y, lq, lp = dyn.regime_chain(lr, ir, gum, hard)
    ((y * w_y).sum() + (lq * w_q).sum() + (lp * w_p).sum()).backward()
    ld, idv = logits.clone().to(DEV).requires_grad_(True), init.clone().to(DEV).requires_grad_(True)
    P = dyn.prior.transition_matrix.to(DEV)
    y2, lq2, lp2 = RegimeChain.apply(ld, idv, gum.to(DEV), P, tau, hard)
    ((y2 * w_y.to(DEV)).sum() + (lq2 * w_q.to(DEV)).sum() + (lp2 * w_p.to(DEV)).sum()).backward()
    assert rel_err(y2.detach().cpu(), y.detach()) < 2e-5
    assert rel_err(lq2.detach().cpu(), lq.detach()) < 2e-5 and rel_err(lp2.detach().cpu(), lp.detach()) < 2e-5
    ref_lg = lr.grad if lr.grad is not None else torch.zeros_like(logits)   # T == 1: transition logits unused
    assert (ld.grad.cpu() - ref_lg).abs().max() <= 2e-4 * ref_lg.abs().max() + 1e-12
    assert rel_err(idv.grad.cpu(), ir.grad) < 2e-4


def bce_frames_vs_torch(DEV, B, T, C, H, W):
    import torch.nn.functional as F
    from kvae.vae.fused import BernoulliFrameLogLik
    g = torch.Generator().manual_seed(B + T + H)
    logits = 3 * torch.randn(B, T, C, H, W, generator=g)
    x = (torch.rand(B, T, C, H, W, generator=g) < 0.2).float()
    w = torch.randn(B, T, generator=g)
    lr = logits.clone().requires_grad_(True)
    ref = -F.binary_cross_entropy_with_logits(lr, x, reduction="none").sum(dim=(2, 3, 4))
    (ref * w).sum().backward()
    ld = logits.clone().to(DEV).requires_grad_(True)
    out = BernoulliFrameLogLik.apply(ld, x.to(DEV))
    (out * w.to(DEV)).sum().backward()
    assert rel_err(out.detach().cpu(), ref.detach()) < 1e-5
    assert rel_err(ld.grad.cpu(), lr.grad) < 1e-5


def conv_edge_vs_torch(DEV, N):
    """Direct decoder-head / encoder-stem kernels == conv2d (+ pixel_shuffle / relu) of torch, values and gradients.
    Tolerance 2e-5 relative (fp32, different summation order over 288 / 9 taps and over the frames)."""
    import torch.nn.functional as F
    from kvae.vae.fused import DecoderHead, EncoderStem
    g = torch.Generator().manual_seed(N)
    h = torch.relu(torch.randn(N, 32, 16, 16, generator=g))
    W = 0.1 * torch.randn(4, 32, 3, 3, generator=g)
    b = torch.randn(4, generator=g)
    up = torch.randn(N, 1, 32, 32, generator=g)
    hr, Wr, br = (t.clone().requires_grad_(True) for t in (h, W, b))
    ref = F.pixel_shuffle(F.conv2d(hr, Wr, br, padding=1), 2)
    (ref * up).sum().backward()
    hd, Wd, bd = (t.clone().to(DEV).requires_grad_(True) for t in (h, W, b))
    out = DecoderHead.apply(hd, Wd, bd)
    (out * up.to(DEV)).sum().backward()
    assert rel_err(out.detach().cpu(), ref.detach()) < 2e-5
    for a, r in ((hd, hr), (Wd, Wr), (bd, br)):
        assert rel_err(a.grad.cpu(), r.grad) < 2e-5

    x = torch.rand(N, 1, 32, 32, generator=g)
    W = 0.3 * torch.randn(32, 1, 3, 3, generator=g)
    b = 0.1 * torch.randn(32, generator=g)
    up = torch.randn(N, 32, 16, 16, generator=g)
    Wr, br = (t.clone().requires_grad_(True) for t in (W, b))
    ref = torch.relu(F.conv2d(x, Wr, br, stride=2, padding=1))
    (ref * up).sum().backward()
    Wd, bd = (t.clone().to(DEV).requires_grad_(True) for t in (W, b))
    out = EncoderStem.apply(x.to(DEV), Wd, bd)
    (out * up.to(DEV)).sum().backward()
    assert rel_err(out.detach().cpu(), ref.detach()) < 2e-5
    for a, r in ((Wd, Wr), (bd, br)):
        assert rel_err(a.grad.cpu(), r.grad) < 2e-5


def enc_mid_vs_torch(DEV, N, side):
    """MFMA stride-2 encoder layers == relu(conv2d(stride 2)) of torch, values and all three gradients.  Tolerance 3e-5
    relative: exact-f32 MFMA is a k-ordered fmaf chain, torch's conv sums the 288 taps in another order."""
    import torch.nn.functional as F
    from kvae.vae.fused import EncoderMid
    g = torch.Generator().manual_seed(N * side)
    x = torch.relu(torch.randn(N, 32, side, side, generator=g))
    W = 0.08 * torch.randn(32, 32, 3, 3, generator=g)
    b = 0.1 * torch.randn(32, generator=g)
    up = torch.randn(N, 32, side // 2, side // 2, generator=g)
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    ref = torch.relu(F.conv2d(xr, Wr, br, stride=2, padding=1))
    (ref * up).sum().backward()
    xd, Wd, bd = (t.clone().to(DEV).requires_grad_(True) for t in (x, W, b))
    out = EncoderMid.apply(xd, Wd, bd)
    (out * up.to(DEV)).sum().backward()
    assert rel_err(out.detach().cpu(), ref.detach()) < 3e-5
    for a, r in ((xd, xr), (Wd, Wr), (bd, br)):
        assert rel_err(a.grad.cpu(), r.grad) < 3e-5


def dec_up_vs_torch(DEV, N, side):
    """Register-stationary MFMA decoder blocks == relu(pixel_shuffle(conv2d)) of torch, values and all three gradients.
    Tolerance 3e-5 relative (fp32; different summation order over 288 / 1152 taps and over the frames)."""
    import torch.nn.functional as F
    from kvae.vae.fused import DecoderUp
    g = torch.Generator().manual_seed(N * side + 1)
    x = torch.relu(torch.randn(N, 32, side, side, generator=g))
    W = 0.08 * torch.randn(128, 32, 3, 3, generator=g)
    b = 0.1 * torch.randn(128, generator=g)
    up = torch.randn(N, 32, 2 * side, 2 * side, generator=g)
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    pre = F.pixel_shuffle(F.conv2d(xr, Wr, br, padding=1), 2)
    # a pre-activation within rounding of 0 may land on either side of the ReLU depending on the summation order (direct /
    # Winograd / torch's): such pixels carry no upstream gradient here, so the mask they pick cannot decide the comparison
    up = up * (pre.detach().abs() > 1e-5)
    ref = torch.relu(pre)
    (ref * up).sum().backward()
    xd, Wd, bd = (t.clone().to(DEV).requires_grad_(True) for t in (x, W, b))
    out = DecoderUp.apply(xd, Wd, bd)
    (out * up.to(DEV)).sum().backward()
    assert rel_err(out.detach().cpu(), ref.detach()) < 3e-5
    for a, r in ((xd, xr), (Wd, Wr), (bd, br)):
        assert rel_err(a.grad.cpu(), r.grad) < 3e-5


def vae_heads_vs_torch(DEV, N):
    """Fused encoder heads (+ reparameterisation), decoder fc and latent regulariser == the torch expressions of the
    reference, values and gradients.  Tolerance 2e-5 relative (fp32, different summation order over 512 features / N rows)."""
    import torch.nn.functional as F
    from kvae.vae.fused import DecoderFc, EncoderHead, LatentReg
    g = torch.Generator().manual_seed(N + 7)
    feat = torch.relu(torch.randn(N, 512, generator=g))
    Wm, bm = 0.05 * torch.randn(2, 512, generator=g), 0.1 * torch.randn(2, generator=g)
    Wv, bv = 0.05 * torch.randn(2, 512, generator=g), 0.1 * torch.randn(2, generator=g)
    eps = torch.randn(N, 2, generator=g)
    ups = [torch.randn(N, 2, generator=g) for _ in range(3)]
    ne = 0.03
    ref_in = [t.clone().requires_grad_(True) for t in (feat, Wm, bm, Wv, bv)]
    mu = F.linear(ref_in[0], ref_in[1], ref_in[2])
    var = ne * torch.sigmoid(F.linear(ref_in[0], ref_in[3], ref_in[4]))
    a = mu + eps * torch.sqrt(var + 1e-6)
    (a * ups[0] + mu * ups[1] + var * ups[2]).sum().backward()
    dev_in = [t.clone().to(DEV).requires_grad_(True) for t in (feat, Wm, bm, Wv, bv)]
    a_d, mu_d, var_d = EncoderHead.apply(*dev_in, eps.to(DEV), ne)
    (a_d * ups[0].to(DEV) + mu_d * ups[1].to(DEV) + var_d * ups[2].to(DEV)).sum().backward()
    for got, want in ((a_d, a), (mu_d, mu), (var_d, var)):
        assert rel_err(got.detach().cpu(), want.detach()) < 2e-5
    for got, want in zip(dev_in, ref_in):
        assert rel_err(got.grad.cpu(), want.grad) < 2e-5

    lat = torch.randn(N, 2, generator=g)
    W, b = 0.3 * torch.randn(512, 2, generator=g), 0.1 * torch.randn(512, generator=g)
    up = torch.randn(N, 512, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (lat, W, b)]
    (F.linear(*ref_in) * up).sum().backward()
    dev_in = [t.clone().to(DEV).requires_grad_(True) for t in (lat, W, b)]
    h = DecoderFc.apply(*dev_in)
    (h * up.to(DEV)).sum().backward()
    assert rel_err(h.detach().cpu(), F.linear(lat, W, b)) < 2e-5
    for got, want in zip(dev_in, ref_in):
        assert rel_err(got.grad.cpu(), want.grad) < 2e-5

    B, T = 3, max(N // 3, 1)
    a0, m0 = torch.randn(B, T, 2, generator=g), torch.randn(B, T, 2, generator=g)
    v0 = 0.01 + 0.05 * torch.rand(B, T, 2, generator=g)
    w = torch.randn(B, T, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (a0, m0, v0)]
    lg = lambda x, mean, var: -0.5 * math.log(2 * math.pi) - 0.5 * torch.log(var) - (x - mean) ** 2 / (2 * var)
    ref = (lg(ref_in[0], torch.zeros(()), torch.ones(())) - lg(*ref_in)).sum(-1)
    (ref * w).sum().backward()
    dev_in = [t.clone().to(DEV).requires_grad_(True) for t in (a0, m0, v0)]
    reg = LatentReg.apply(*dev_in)
    (reg * w.to(DEV)).sum().backward()
    assert rel_err(reg.detach().cpu(), ref.detach()) < 2e-5
    for got, want in zip(dev_in, ref_in):
        assert rel_err(got.grad.cpu(), want.grad) < 2e-5

    # scalar head of the objective
    from kvae.vae.fused import LossHead
    lpx, rg = torch.randn(B, T, generator=g) * 50, torch.randn(B, T, generator=g)
    kf = torch.randn((), generator=g)
    for mk in (None, (torch.rand(B, T, generator=g) < 0.7).float()):
        ref_in = [t.clone().requires_grad_(True) for t in (lpx, rg, kf)]
        m = torch.ones(B, T) if mk is None else mk
        denom = m.sum().clamp(min=1.0)
        recon, reg = (ref_in[0] * m).sum() / denom, (ref_in[1] * m).sum() / denom
        vae = 0.3 * recon + 0.7 * reg
        tot = 1.5 * vae + 0.8 * ref_in[2]
        (-tot * 2.0).backward()
        dev_in = [t.clone().to(DEV).requires_grad_(True) for t in (lpx, rg, kf)]
        out = LossHead.apply(dev_in[0], dev_in[1], dev_in[2], None if mk is None else mk.to(DEV), torch.tensor(0.7).to(DEV), 0.3, 1.5, 0.8)
        (out[0] * 2.0).backward()
        for got, want in zip(out, (-tot, tot, kf, vae, recon, reg)):
            assert rel_err(got.detach().cpu(), want.detach()) < 2e-5
        for got, want in zip(dev_in, ref_in):
            assert rel_err(got.grad.cpu(), want.grad) < 2e-5
        # the same through DEVICE scalars for (vae_weight, kf_weight): the by-value floats are then ignored
        dev_w = [t.clone().to(DEV).requires_grad_(True) for t in (lpx, rg, kf)]
        out_w = LossHead.apply(dev_w[0], dev_w[1], dev_w[2], None if mk is None else mk.to(DEV), torch.tensor(0.7).to(DEV), 0.3, 0.0, 0.0,
                               torch.tensor([1.5, 0.8]).to(DEV))
        (out_w[0] * 2.0).backward()
        for got, want in zip(out_w, out):
            assert torch.equal(got.detach().cpu(), want.detach().cpu())
        for got, want in zip(dev_w, dev_in):
            assert torch.equal(got.grad.cpu(), want.grad.cpu())


def colsum_pair_vs_torch(DEV):
    """_native.colsum_pair: short partials (one launch), tall partials of equal height (two launches, both tensors folded
    [64, rows/64 * cols] -> [rows/64, cols] in each) and unequal heights (two separate colsum calls) against torch's sum."""
    from kvae import _native
    g = torch.Generator().manual_seed(5)
    for ra, ca, rb, cb in [(100, 36, 100, 4), (2048, 36, 2048, 4), (1024, 288, 1024, 32), (2048, 36, 1024, 4), (1088, 7, 1088, 3),
                           (1030, 36, 1030, 4)]:
        a, b = torch.randn(ra, ca, generator=g).to(DEV), torch.randn(rb, 2, cb // 2 if cb % 2 == 0 else cb, generator=g).to(DEV)
        sa, sb = _native.colsum_pair(a, b)
        assert sa.shape == a.shape[1:] and sb.shape == b.shape[1:]
        assert rel_err(sa.cpu(), a.cpu().double().sum(0).float()) < 2e-6
        assert rel_err(sb.cpu(), b.cpu().double().sum(0).float()) < 2e-6


def dec_up_workgroup_cap(DEV):
    """kvae_dec_up_set_workgroups: fewer persistent workgroups than column sets (a stride loop with a ragged last round, fewer
    rows of weight-gradient partials) give the same block as the default; the previous value comes back and 0 restores it."""
    from kvae import _native
    lib = _native.lib_for(torch.zeros(1, device=DEV))
    default = lib.dll.kvae_dec_up_set_workgroups(7)
    try:
        assert lib.dll.kvae_dec_up_partial_rows(100, 8) == 7 and lib.dll.kvae_dec_up_partial_rows(5, 4) == 1
        dec_up_vs_torch(DEV, 100, 8)
        dec_up_vs_torch(DEV, 45, 4)
        assert lib.dll.kvae_dec_up_set_workgroups(1000) == 7   # out of range: back to the default
        assert lib.dll.kvae_dec_up_partial_rows(12800, 8) == 256
    finally:
        lib.dll.kvae_dec_up_set_workgroups(0 if default == 256 else default)


def check_phases(g, set_phase, run_step, params_now, adam_steps, value_tol=1e-4, param_tol=1e-3):
    """One implementation of the training step against the reference's three phases as recorded in phases_*.npz
    (tests/golden/make_goldens_r3.py: the reference's own set_training_phase and train_one_epoch, two steps per phase in the
    order vae -> warmup -> all, ONE Adam over all parameters).
      set_phase(phase); run_step(phase, i, kf_weight) -> dict(loss, elbo_kf, elbo_vae_total); params_now() -> {name: tensor};
      adam_steps() -> per-parameter step counts in .parameters() order."""
    buffers = ("kalman_filter.Q", "kalman_filter.R", "kalman_filter.I", "kalman_filter.mu0", "kalman_filter.Sigma0")
    names = [k[3:] for k in g if k.startswith("sd.") and k[3:] not in buffers]   # .parameters() order (the generator asserts it)
    prev = {k: v.clone() for k, v in params_now().items()}
    for phase in ("vae", "warmup", "all"):
        set_phase(phase)
        kf_w = float(g[f"{phase}.kf_weight"])
        outs = [run_step(phase, i, kf_w) for i in range(2)]
        for k in ("loss", "elbo_kf", "elbo_vae_total"):
            mean = sum(float(o[k]) for o in outs) / 2
            want = float(g[f"{phase}.mean_{k}"])
            assert abs(mean - want) <= value_tol * abs(want), (phase, k, mean, want)
        now = {k: v.clone() for k, v in params_now().items()}
        for i, k in enumerate(names):
            unchanged = bool(g[f"{phase}.unchanged"][i])
            if unchanged:   # frozen by the phase (or a zero gradient on zero moments): bit-identical, as in the reference
                assert torch.equal(now[k].cpu(), prev[k].cpu()), (phase, k, "moved but the reference left it bit-identical")
            else:
                assert not torch.equal(now[k].cpu(), prev[k].cpu()), (phase, k, "did not move")
                dn, want = float((now[k].cpu() - prev[k].cpu()).norm()), float(g[f"{phase}.delta_norm.{k}"])
                # Adam's normalised update: every entry moves by ~lr whatever its gradient, so the norm of the change is tight
                assert abs(dn - want) <= 2e-2 * want + 1e-7, (phase, k, dn, want)
            if f"{phase}.after.{k}" in g:
                # entries whose gradient is O(1e-8) are rounding-sensitive under Adam (lr*g/(|g|+eps)): 1e-3 of max|param|
                assert rel_err(now[k].cpu(), g[f"{phase}.after.{k}"]) < param_tol, (phase, k)
        prev = now
    assert [float(s) for s in adam_steps()] == [float(s) for s in g["adam_steps"]]


def rnn_wgrad_vs_torch(DEV):
    """kvae_rnn_wgrad (dW_hh | dW_ih | db of a recurrence, dW | db of a head; f32 matrix cores, split over the rows, fixed-order
    second stage) against the products torch would form: LSTM-shaped (200 x [50 | 2 | 1], h_{t-1}), both GRU directions in one
    batched call (h_{t-1} / h_{t+1} on the two halves of a [.., 2H] sequence), a head (K x [H | 1]), ragged row counts."""
    from kvae.kalman.lgssm_ops import rnn_wgrad
    g = torch.Generator().manual_seed(17)
    for Bsz, T, H, I, R in [(3, 7, 50, 2, 200), (256, 50, 50, 2, 200), (5, 1, 50, 2, 200), (2, 9, 13, 3, 37)]:
        d = torch.randn(Bsz * T, R, generator=g)
        h = torch.randn(Bsz, T, H, generator=g)
        x = torch.randn(Bsz * T, I, generator=g)
        hp = torch.cat([torch.zeros(Bsz, 1, H), h[:, :-1]], 1).reshape(Bsz * T, H)
        (gwh, gwx, gb), = rnn_wgrad(d.to(DEV), [dict(d=d.to(DEV), h=h.reshape(Bsz * T, H).to(DEV), shift=-1, T=T, x=x.to(DEV))])
        dd = d.double()
        for got, want in ((gwh, dd.t() @ hp.double()), (gwx, dd.t() @ x.double()), (gb, dd.sum(0))):
            assert rel_err(got.cpu(), want.float()) < 2e-5
    Bsz, T, H, I = 4, 11, 50, 2
    h2 = torch.randn(Bsz, T, 2 * H, generator=g)
    x = torch.randn(Bsz * T, I, generator=g)
    dpi, dph = torch.randn(2, Bsz * T, 3 * H, generator=g), torch.randn(2, Bsz * T, 3 * H, generator=g)
    zero = torch.zeros(Bsz, 1, H)
    hp = (torch.cat([zero, h2[:, :-1, :H]], 1), torch.cat([h2[:, 1:, H:], zero], 1))
    h2d = h2.reshape(Bsz * T, 2 * H).to(DEV)
    probs = []
    for dr in (0, 1):
        probs.append(dict(d=dpi[dr].to(DEV), x=x.to(DEV)))
        probs.append(dict(d=dph[dr].to(DEV), h=h2d[:, dr * H:(dr + 1) * H], shift=-1 if dr == 0 else 1, T=T))
    res = rnn_wgrad(h2d, probs)
    for dr in (0, 1):
        (_, gwx, gbi), (gwh, _, gbh) = res[2 * dr], res[2 * dr + 1]
        assert rel_err(gwx.cpu(), (dpi[dr].double().t() @ x.double()).float()) < 2e-5
        assert rel_err(gwh.cpu(), (dph[dr].double().t() @ hp[dr].reshape(Bsz * T, H).double()).float()) < 2e-5
        assert rel_err(gbi.cpu(), dpi[dr].double().sum(0).float()) < 2e-5 and rel_err(gbh.cpu(), dph[dr].double().sum(0).float()) < 2e-5
    run1 = rnn_wgrad(h2d, probs)   # fixed summation order: bit-identical from run to run
    assert all(torch.equal(a, b) for r0, r1 in zip(res, run1) for a, b in zip(r0, r1) if a is not None)


def small_linear_vs_torch(DEV):
    """SmallLinear (+ fused softmax) forward, input gradient and parameter gradients against torch.nn.functional.linear /
    softmax autograd: the alpha head (K x 50, softmax), the regime posterior's heads (K^2 x 100 on [B,T,100], K x 100 on the
    strided rows h_seq[:, 0])."""
    from kvae.kalman.lgssm_ops import SmallLinear, small_linear_supported
    g = torch.Generator().manual_seed(23)
    for lead, F, O, softmax, strided in [((6, 9), 50, 3, True, False), ((256, 50), 50, 7, True, False), ((5, 8), 100, 9, False, False),
                                         ((3, 4), 100, 49, False, False), ((7,), 100, 7, False, True), ((2, 3), 17, 5, False, False)]:
        if strided:   # rows T*F apart, as h_seq[:, 0]
            base = torch.randn(lead[0], 6, F, generator=g)
            x_ref = base[:, 0].clone().requires_grad_(True)
            base_dev = base.to(DEV).requires_grad_(True)
            x_dev = base_dev[:, 0]
        else:
            x0 = torch.randn(*lead, F, generator=g)
            x_ref, x_dev = x0.clone().requires_grad_(True), x0.to(DEV).requires_grad_(True)
        w0, b0 = torch.randn(O, F, generator=g) * 0.3, torch.randn(O, generator=g)
        up = torch.randn(*lead, O, generator=g)
        w_ref, b_ref = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        y_ref = torch.nn.functional.linear(x_ref, w_ref, b_ref)
        if softmax:
            y_ref = torch.softmax(y_ref, -1)
        (y_ref * up).sum().backward()
        w_dev, b_dev = w0.to(DEV).requires_grad_(True), b0.to(DEV).requires_grad_(True)
        assert small_linear_supported(x_dev, w_dev, softmax)
        y = SmallLinear.apply(x_dev, w_dev, b_dev, softmax)
        (y * up.to(DEV)).sum().backward()
        assert rel_err(y.detach().cpu(), y_ref.detach()) < 1e-5
        gx = base_dev.grad[:, 0] if strided else x_dev.grad
        assert rel_err(gx.cpu(), x_ref.grad) < 2e-5
        assert rel_err(w_dev.grad.cpu(), w_ref.grad) < 2e-5 and rel_err(b_dev.grad.cpu(), b_ref.grad) < 2e-5


def backward_shard_vs_slices(DEV, B, T, n, slice_b):
    """The LGSSM forward + ELBO + backward on B sequences in one launch against the same on B / slice_b launches of slice_b
    sequences each: per-sequence gradients (Y, U, alpha) bit-identical; the summed base-matrix gradients (A, B, C: reduced over
    the batch in a different grouping) to rounding.  Runs under no host-side oracle: the slices ARE the oracle-checked size."""
    from kvae.kalman.lgssm_ops import LgssmElbo, LgssmSmooth, Slots, mix_dynamics
    A, Bm, Cm, alpha, Y, U, mask, eps = _random_problem(B, T, n, n, 2, 3, 100 + B + T, DEV)
    R, Q = 0.03 * torch.eye(2, device=DEV), 0.02 * torch.eye(n, device=DEV)
    mu0, S0 = torch.zeros(n, device=DEV), 20.0 * torch.eye(n, device=DEV)

    def run(sl):
        leaves = [t.clone().requires_grad_(True) for t in (A, Bm, Cm, alpha[sl], Y[sl], U[sl])]
        rec, offs, _ = mix_dynamics(leaves[3], leaves[:3])
        slots = Slots(A=offs[0], B=offs[1], C=offs[2])
        mk = None if mask is None else mask[sl]
        ms, Ss, *_ = LgssmSmooth.apply(leaves[4], leaves[5], mk, rec, None, None, None, Q, R, mu0, S0, slots, True)
        total, _, _ = LgssmElbo.apply(ms, Ss, eps[sl], leaves[4], leaves[5], mk, rec, None, None, None, Q, R, mu0, S0, slots)
        total.backward()
        return [t.grad for t in leaves], rec.detach(), ms.detach()

    big, rec_big, ms_big = run(slice(0, B))
    assert all(torch.isfinite(g).all() for g in big)
    acc = [torch.zeros_like(g) for g in big[:3]]
    for b0 in range(0, B, slice_b):
        sl = slice(b0, b0 + slice_b)
        small, _, ms_small = run(sl)
        assert torch.equal(ms_small, ms_big[sl])
        for k, name in ((3, "alpha"), (4, "Y"), (5, "U")):
            assert torch.equal(small[k], big[k][sl]), (name, b0)
        for k in range(3):
            acc[k] += small[k]
    for k, name in enumerate("ABC"):
        assert rel_err(acc[k].cpu(), big[k].cpu()) < 1e-4, name


def mix_vs_torch(DEV):
    """kvae_mix_fwd / kvae_mix_bwd (mixture-of-K step records, reference dyn_param.py:58-60 / switch_dyn_param.py:82-84) against
    torch.einsum and its autograd: the streaming kernels' shapes (E = 40 / 48 at n = 4, 544 / 768 at n = 16, K = 3, 7, 8), ragged row
    counts, a K above their limit (element-wise fallback), and the accumulate flag of the C ABI."""
    import ctypes as C
    from kvae import _native as N
    from kvae.kalman.lgssm_ops import MixDynamics
    g = torch.Generator().manual_seed(31)
    for Bsz, T, K, E in [(256, 50, 3, 40), (3, 7, 3, 48), (32, 100, 7, 48), (5, 9, 3, 768), (64, 33, 3, 544), (2, 5, 8, 768),
                         (4, 6, 9, 40), (1, 1, 2, 12), (2, 3, 3, 42)]:
        alpha = torch.softmax(torch.randn(Bsz, T, K, generator=g), -1)
        base = torch.randn(K, E, generator=g)
        up = torch.randn(Bsz, T, E, generator=g)
        ar, br = alpha.clone().requires_grad_(True), base.clone().requires_grad_(True)
        ref = torch.einsum("btk,ke->bte", ar, br)
        (ref * up).sum().backward()
        ad, bd = alpha.to(DEV).requires_grad_(True), base.to(DEV).requires_grad_(True)
        out = MixDynamics.apply(ad, bd)
        (out * up.to(DEV)).sum().backward()
        assert rel_err(out.detach().cpu(), ref.detach()) < 1e-6, (K, E)
        assert rel_err(ad.grad.cpu(), ar.grad) < 2e-5 and rel_err(bd.grad.cpu(), br.grad) < 2e-5, (K, E)
    # accumulate_alpha = 1 through the raw C ABI: g_alpha += ...
    Bsz, T, K, E = 7, 5, 3, 768
    alpha = torch.softmax(torch.randn(Bsz * T, K, generator=g), -1).to(DEV)
    base, gout = torch.randn(K, E, generator=g).to(DEV), torch.randn(Bsz * T, E, generator=g).to(DEV)
    lib = N.lib_for(alpha)
    nblk = lib.dll.kvae_mix_bwd_partials(Bsz * T)
    partials = torch.empty(nblk, K, E, device=alpha.device)
    g_alpha0 = torch.randn(Bsz * T, K, generator=g).to(DEV)
    g_alpha, g_base = g_alpha0.clone(), torch.empty_like(base)
    assert lib.dll.kvae_mix_bwd(N.ptr(alpha), N.ptr(base), N.ptr(gout), N.ptr(g_alpha), N.ptr(g_base), N.ptr(partials), Bsz * T, K, E,
                                1, N.stream_for(alpha)) == 0
    assert rel_err((g_alpha - g_alpha0).cpu(), (gout @ base.T).cpu()) < 2e-5
    assert rel_err(g_base.cpu(), (alpha.T @ gout).cpu()) < 2e-5
