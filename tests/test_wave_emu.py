"""CPU tier: the product's WAVEFRONT-LEVEL kernels - (4,4,2) on csrc/lgssm_m4.h over the quad layout of lgssm_q4.h, (16,16,2) on
csrc/lgssm_n16.h: the code `kvae_lgssm_n16.hip` wraps in __global__ functions - run on emulated wavefronts (tests/hostsim/
wave_emu.h: 64 host threads per wavefront, every DPP move / permlane swap / ds_bpermute / ballot / f32 MFMA tile a rendezvous)
and compared with the oracle and with the generic bodies of the host simulation.  The same runs go under AddressSanitizer in
tests/test_hostsim_asan.py, which is what puts the ragged last wavefront, the unrolled loops' tails and the 16-byte vector
accesses of the default hot path under a sanitizer (GPU sanitizers are unavailable on the pool)."""
import pytest
import torch

import parity_cases
from golden_util import rel_err
from hostsim.build import build as build_hostsim

torch.set_num_threads(4)
KIND = {4: (0, 1), 16: (2, 3)}   # indices of kvae_wemu_launches: forward / backward launches at n


@pytest.fixture(scope="module", autouse=True)
def wave_emu_backend():
    from kvae import _native
    lib = _native.LgssmLib(build_hostsim())
    _native._set_test_backend(lib)
    lib.dll.kvae_hostsim_wave_emu(1)
    yield lib
    lib.dll.kvae_hostsim_wave_emu(0)
    _native._set_test_backend(None)


def launches(lib):
    return [lib.dll.kvae_wemu_launches(i) for i in range(4)]


@pytest.mark.parametrize("B,T,n,K,dense_q", [(19, 5, 4, 3, False), (3, 2, 4, 2, False), (1, 1, 4, 3, False), (17, 3, 4, 2, True),
                                             (2, 4, 16, 2, False), (1, 1, 16, 2, False), (2, 3, 16, 2, True)])
def test_wave_kernels_vs_oracle(wave_emu_backend, B, T, n, K, dense_q):
    """Values vs the C oracle, gradients vs the torch oracle's autograd: ragged last wavefront (19, 17 of 16 sequences per
    wavefront), T = 1, 2, 3 (the tails of the loops unrolled over two / three operand sets), masks, controls."""
    before = launches(wave_emu_backend)
    parity_cases.vs_oracle_random("cpu", B, T, n, n, 2, K, dense_q=dense_q)
    after = launches(wave_emu_backend)
    f, b = KIND[n]
    assert after[f] > before[f] and after[b] > before[b], (before, after)   # the emulated kernels are what ran


@pytest.mark.parametrize("n,B,T,indefinite", [(4, 18, 4, False), (16, 2, 3, False), (4, 5, 4, True)])
def test_wave_kernels_equal_generic_bodies(wave_emu_backend, n, B, T, indefinite):
    """Every variant of the two kernels (per-step Q: HAS_GQ; upstream gradients on all six stacks: HAS_FP; a mask) against the
    generic one-wavefront-per-sequence bodies of the host simulation on the same inputs: all six stacks and all gradients.
    indefinite: mode noises that are not positive semi-definite - the natural-order 4x4 solve of lgssm_m4.h meets a non-positive
    pivot and the step repeats it with the pivoted elimination of lgssm_q4.h (ballot, ds_bpermute row exchange)."""
    from kvae.kalman.lgssm_ops import LgssmSmooth, Slots, mix_dynamics
    m, p, K = n, 2, 3
    A, Bm, Cm, alpha, Y, U, mask, _ = parity_cases._random_problem(B, T, n, m, p, K, 900 + n, "cpu")
    g = torch.Generator().manual_seed(5)
    qq = 0.05 * torch.randn(K, n, n, generator=g)
    Qk = 0.02 * torch.eye(n).repeat(K, 1, 1) + qq @ qq.mT
    if indefinite:
        qh = 0.3 * torch.randn(K, n, n, generator=g)
        Qk = 0.5 * (qh + qh.mT)
        assert float(torch.linalg.eigvalsh(Qk).min()) < -0.05
    R, mu0, S0 = 0.03 * torch.eye(p), 0.1 * torch.randn(n, generator=g), 2.0 * torch.eye(n)
    w = [torch.randn(B, T, n, generator=g) if i % 2 == 0 else torch.randn(B, T, n, n, generator=g) for i in range(6)]

    def run(emu):
        wave_emu_backend.dll.kvae_hostsim_wave_emu(1 if emu else 0)
        leaves = [t.clone().requires_grad_(True) for t in (A, Bm, Qk, alpha, Y, U)]
        rec, offs, _ = mix_dynamics(leaves[3], leaves[:3])
        outs = LgssmSmooth.apply(leaves[4], leaves[5], mask, rec, None, None, Cm[0], None, R, mu0, S0,
                                 Slots(A=offs[0], B=offs[1], Q=offs[2]), True)
        sum((o * wi).sum() for o, wi in zip(outs, w)).backward()
        return [o.detach() for o in outs], [t.grad for t in leaves]

    before = launches(wave_emu_backend)
    try:
        outs_e, grads_e = run(True)
        after = launches(wave_emu_backend)
        outs_g, grads_g = run(False)
    finally:
        wave_emu_backend.dll.kvae_hostsim_wave_emu(1)
    f, b = KIND[n]
    assert after[f] > before[f] and after[b] > before[b], (before, after)
    tol = 3e-4 if (n == 16 or indefinite) else 2e-5
    for name, a, c in zip(("mus_smooth", "Sigmas_smooth", "mus_filt", "Sigmas_filt", "mus_pred", "Sigmas_pred"), outs_e, outs_g):
        assert rel_err(a, c) < tol, name
    for name, a, c in zip("A B Q alpha Y U".split(), grads_e, grads_g):
        assert rel_err(a, c) < 20 * tol, name


def test_wave_n16_pivoted_solve(wave_emu_backend):
    """The indefinite-Q case (natural-order elimination meets a non-positive pivot, the pivoted one takes over: ds_bpermute row
    exchanges, ballots) on the emulated (16,16,2) kernels."""
    before = launches(wave_emu_backend)
    parity_cases.n16_indefinite_q("cpu", B=2, T=6)
    after = launches(wave_emu_backend)
    assert after[2] > before[2] and after[3] > before[3]


def test_wave_n4_split_and_single_launch_forms_give_the_same_bits(wave_emu_backend):
    """Below 2048 sequences the (4,4,2) smoother and its adjoint run as "dependent chain, then everything that hangs off it for
    all steps at once" (3 / 4 launches); above, as one launch each.  Same operations on the same operands: the six stacks and
    every gradient must agree bit for bit (per-step Q, upstream gradients on all stacks, a mask)."""
    from kvae.kalman.lgssm_ops import LgssmSmooth, Slots, mix_dynamics
    B, T, n, m, p, K = 21, 7, 4, 4, 2, 3
    A, Bm, Cm, alpha, Y, U, mask, _ = parity_cases._random_problem(B, T, n, m, p, K, 4242, "cpu")
    g = torch.Generator().manual_seed(9)
    qq = 0.05 * torch.randn(K, n, n, generator=g)
    Qk = 0.02 * torch.eye(n).repeat(K, 1, 1) + qq @ qq.mT
    R, mu0, S0 = 0.03 * torch.eye(p), 0.1 * torch.randn(n, generator=g), 2.0 * torch.eye(n)
    w = [torch.randn(B, T, n, generator=g) if i % 2 == 0 else torch.randn(B, T, n, n, generator=g) for i in range(6)]

    def run(split_max_b):
        wave_emu_backend.dll.kvae_wemu_m4_split_max_b(split_max_b)
        leaves = [t.clone().requires_grad_(True) for t in (A, Bm, Qk, alpha, Y, U)]
        rec, offs, _ = mix_dynamics(leaves[3], leaves[:3])
        outs = LgssmSmooth.apply(leaves[4], leaves[5], mask, rec, None, None, Cm[0], None, R, mu0, S0,
                                 Slots(A=offs[0], B=offs[1], Q=offs[2]), True)
        sum((o * wi).sum() for o, wi in zip(outs, w)).backward()
        return [o.detach() for o in outs] + [t.grad for t in leaves]

    try:
        split, single = run(1 << 20), run(0)
    finally:
        wave_emu_backend.dll.kvae_wemu_m4_split_max_b(-1)
    for a, b in zip(split, single):
        assert torch.equal(a, b)
