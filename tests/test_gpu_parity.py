"""GPU tier (-m gpu): the gfx950 library against the goldens captured from the reference, against the
CPU oracle on seeded inputs, and at BASELINE config sizes.  Every call goes through the C ABI of
include/kvae_lgssm.h (ctypes -> libkvae_lgssm.so); nothing here can pass on a CPU fallback because
the product has none (test_native_loaded asserts the HIP library is the one mapped in)."""
import ctypes as C

import pytest
import torch

from golden_util import JITTER_CASES, LATENT_CASES, SMOOTH_KEYS, load, rel_err, stability_state_dict, sub
import parity_cases
from parity_cases import _random_problem
from test_hostsim_ops import check_latent, make_filter, run_latent

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_native_loaded():
    from kvae import _native
    lib = _native.hip_lib()
    assert "gfx950" in lib.build_info
    with open("/proc/self/maps") as f:
        assert "libkvae_lgssm.so" in f.read()
    assert _native._test_backend is None


@pytest.mark.parametrize("name,kind", LATENT_CASES)
def test_latent_gpu(name, kind):
    """smooth + elbo + gradients vs the reference's own outputs (ELBO / smoothed means within 1e-4 rel)."""
    g = load(name)
    kf = make_filter(g, kind, DEV)
    a, outs, elbo = run_latent(kf, g, DEV)
    check_latent(kf, g, a, outs, elbo, name, tol_scale=2.0, kind=kind)


@pytest.mark.parametrize("batch", [1, 4])
def test_rocket_gpu(batch):
    from kvae import noise
    from kvae.kalman.kalman_filter import KalmanFilter
    from kvae.kalman.switch_dyn_param import SwitchingDynamicsParameter
    g = load(f"rocket_B{batch}")
    dyn = SwitchingDynamicsParameter(g["A"], g["B"], g["C"])
    kf = KalmanFilter(2.0, 4.0, g["mu0"], g["Sigma0"], dyn).to(DEV)
    Y, U = g["Y"].to(DEV), g["U"].to(DEV)
    with torch.no_grad(), noise.inject(eps_z=g["eps_z"]):
        outs = kf.smooth(Y, U)
        elbo = kf.elbo(outs[0], outs[1], Y, U, outs[6], outs[7], outs[8])
    for k, v in zip(SMOOTH_KEYS, outs):
        assert rel_err(v.cpu(), g[k]) < 5e-5, k
    assert rel_err(elbo.cpu(), g["elbo"]) < 5e-5


@pytest.mark.parametrize("name,kind,K", [("stability_lstm", "lstm", 3), ("stability_switching", "switching", 3),
                                         ("stability_lstm_K7_T100", "lstm", 7),
                                         ("stability_switching_K7_T100", "switching", 7)])
def test_stability_recipe_gpu(name, kind, K):
    """The reference's tests/test_imputation_stability.py recipe through the drop-in KVAE.impute."""
    from kvae import noise
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    g = load(name)
    T = int(g["T"])
    model = KVAE(KVAEConfig(dynamics_model=kind, num_modes=K))
    model.load_state_dict(stability_state_dict(kind, K), strict=True)
    model.to(DEV).eval()
    torch.manual_seed(123)
    x = torch.randn(2, T, 1, 32, 32).to(DEV)
    with noise.inject(eps_a=g["eps_a"], gumbel=g.get("gumbel")):
        out = model.impute(x, g["mask"].to(DEV))
    sl = slice(None) if g["x_recon"].shape[1] == T else slice(None, None, 10)
    for k in ("x_recon", "x_imputed", "x_filtered"):
        assert (out[k].cpu()[:, sl] - g[k]).abs().max() < 2e-6, k
    for k in ("a_imputed", "a_filtered"):
        assert (out[k].cpu() - g[k]).abs().max() < 1e-6, k


@pytest.mark.parametrize("name,kind", [("trainstep_lstm_K3", "lstm"), ("trainstep_switch_K3", "switching")])
def test_train_step_gpu(name, kind):
    """One full step (forward, loss, backward, clip, Adam) vs the reference's recorded step."""
    from kvae import noise
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    g = load(name)
    model = KVAE(KVAEConfig(dynamics_model=kind, num_modes=3))
    model.load_state_dict(sub(g, "sd."), strict=True)
    model.to(DEV).train()
    model.beta = float(g["beta"])
    x = g["frames"].float().to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=float(g["lr"]))
    with noise.inject(eps_a=g["eps_a"], eps_z=g["eps_z"], gumbel=g.get("gumbel")):
        out = model(x, mask=torch.ones(x.shape[:2], device=DEV))
        losses = model.compute_loss(x, out, mask=torch.ones(x.shape[:2], device=DEV))
    losses["loss"].backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), float(g["clip"]))
    opt.step()
    assert rel_err(losses["loss"].detach().cpu(), g["loss"]) < 1e-4
    assert rel_err(losses["elbo_kf"].detach().cpu(), g["elbo_kf"]) < 1e-4
    assert rel_err(out["mus_smooth"].detach().cpu(), g["mus_smooth"]) < 1e-4
    assert rel_err(gn.cpu(), g["grad_norm"]) < 1e-3
    for k, p in model.named_parameters():
        ref = float(g["gradnorm." + k])
        assert abs(float(p.grad.norm()) - ref) <= 2e-3 * ref + 1e-6, k
        if "grad." + k in g and ref > 1e-6:
            assert rel_err(p.grad.cpu(), g["grad." + k]) < 2e-3, k


@pytest.mark.parametrize("B,T,n,m,p,K", [(256, 50, 4, 4, 2, 3), (7, 33, 4, 4, 2, 3), (5, 17, 3, 2, 1, 2),
                                         (3, 9, 8, 5, 3, 4), (64, 200, 16, 16, 2, 3), (1, 1, 4, 4, 2, 3),
                                         (2, 2, 16, 16, 2, 1), (512, 200, 16, 16, 2, 3), (8, 200, 16, 16, 2, 3),
                                         (5, 1, 16, 16, 2, 2), (8192, 6, 4, 4, 2, 3), (3, 2, 4, 4, 2, 3), (17, 3, 4, 4, 2, 2),
                                         (2, 3, 16, 16, 2, 2)])
def test_vs_oracle_random(B, T, n, m, p, K):
    """Values vs the C oracle at every size, incl. the FULL BASELINE configs[4] shard (512, 200, 16); gradients vs the
    torch oracle's autograd up to configs[1] size (256, 50, 4) and for n = 16 at (8, 200, 16).  (8192, 6, 4): many wavefronts of the sixteen-
    sequences-per-wavefront n = 4 kernels, with a ragged last one in (7, 33, 4).  T = 1, 2, 3: the tails of the sweeps' unrolled
    loops (two operand sets at n = 4, three at n = 16, rotated by name)."""
    parity_cases.vs_oracle_random(DEV, B, T, n, m, p, K)


@pytest.mark.parametrize("B,T,n", [(3, 10, 16), (2, 7, 16), (5, 9, 4)])
def test_vs_oracle_dense_shared_q(B, T, n):
    """A full SPD process noise, T not a multiple of four: the four-steps-per-wavefront ELBO of the (16,16,2) kernels (ragged last
    wavefront, chol(Q)^-1 and its transpose) and the n = 4 kernels, values and gradients."""
    parity_cases.vs_oracle_random(DEV, B, T, n, n, 2, 2, dense_q=True)


@pytest.mark.parametrize("B,T,K", [(17, 5, 1), (3, 10, 1), (4, 5, 1), (17, 12, 1), (17, 5, 3)])
def test_single_mode_n16_vs_fp64_oracle(B, T, K):
    """K = 1 at n = 16: the float32 torch tape is itself ~1e-2 from float64 there (found by a random sweep over shapes), so the
    gradients are held to the float64 oracle instead; (17, 5, 3) shows the same sweep point is well conditioned with three modes."""
    tape = parity_cases.grads_vs_fp64_oracle(DEV, B, T, 16, K)
    if K == 3:
        assert tape < 3e-3


@pytest.mark.parametrize("B,T,K", [(8, 200, 3), (3, 37, 1)])
def test_values_n16_vs_fp64_oracle(B, T, K):
    """Smoothed means / covariances / ELBO at n = 16 within max(1e-4, 4 x the float32 oracle's own distance from float64) of the
    float64 oracle: north_star's 1e-4 wherever float32 can deliver it, and an explicit, measured budget where it cannot (a
    single mode, K = 1, is the ill-conditioned corner round 2's sweep found: float32 itself is 1e-2 off there)."""
    report = parity_cases.values_vs_fp64_oracle(DEV, B, T, 16, K)
    if K > 1:
        assert all(mine < 2e-3 for mine, _ in report.values()), report


def test_backward_at_the_full_shard_equals_its_slices():
    """k_smooth_bwd_n16 at the full configs[4] shard (512, 200, 16) against the same kernel on 64 independent 8-sequence slices
    (the size gradient parity with the torch oracle is checked at): sequences do not interact, so every gradient of the big
    launch must be BIT-identical to the slice launches' - a wavefront-indexing or hand-off-record bug at 512 wavefronts shows here."""
    parity_cases.backward_shard_vs_slices(DEV, 512, 200, 16, 8)


def test_dec_up_workgroup_cap():
    parity_cases.dec_up_workgroup_cap(DEV)


def test_colsum_pair():
    parity_cases.colsum_pair_vs_torch(DEV)


def test_n16_generic_fallback():
    parity_cases.n16_generic_fallback(DEV)


def test_n4_generic_fallback():
    parity_cases.n4_generic_fallback(DEV)


def test_n16_indefinite_q_takes_the_pivoted_solve():
    parity_cases.n16_indefinite_q(DEV)


def test_linearity_full_size():
    parity_cases.linearity(DEV, 256, 50)


@pytest.mark.parametrize("n,B,T", [(4, 2, 4), (16, 2, 4), (16, 40, 10)])
def test_safe_cholesky_levels(n, B, T):
    parity_cases.safe_cholesky_levels(DEV, n, B, T)


@pytest.mark.parametrize("n,B,T", [(16, 5, 10), (16, 33, 7), (4, 3, 6), (7, 2, 5)])
def test_safe_cholesky_levels_shared_q(n, B, T):
    """Raised levels and the diagonal fallback with a batch-wide Q, for Sigma_s and Q: at n = 16 on the matrix-core kernels of
    BOTH layouts (the launch reports its family), no generic backup launch."""
    parity_cases.safe_cholesky_levels_shared_q(DEV, n, B, T)


@pytest.mark.parametrize("name,levels", JITTER_CASES)
def test_jitter_golden_gpu(name, levels):
    """_safe_cholesky past level 0, pinned to the REFERENCE (fixtures driven through its own elbo)."""
    parity_cases.jitter_golden(DEV, name, levels)


def test_c_abi_direct_and_errors():
    """Call the C ABI with raw device pointers (what a non-Python host would do) and check the error codes."""
    from kvae import _native as N
    lib = N.hip_lib()
    B, T, n, m, p = 3, 5, 4, 4, 2
    A, Bm, Cm, alpha, Y, U, mask, eps = _random_problem(B, T, n, m, p, 1, 3, DEV)
    R, Q = 0.03 * torch.eye(p, device=DEV), 0.02 * torch.eye(n, device=DEV)
    mu0, S0 = torch.zeros(n, device=DEV), 20.0 * torch.eye(n, device=DEV)
    prob = N.Problem()
    prob.B, prob.T, prob.n, prob.m, prob.p = B, T, n, m, p
    prob.A, prob.Bm = N.Stack(A.data_ptr(), 0, 0), N.Stack(Bm.data_ptr(), 0, 0)
    prob.C, prob.Q = N.Stack(Cm.data_ptr(), 0, 0), N.Stack(Q.data_ptr(), 0, 0)
    prob.R, prob.mu0, prob.Sigma0, prob.Y, prob.U = R.data_ptr(), mu0.data_ptr(), S0.data_ptr(), Y.data_ptr(), U.data_ptr()
    outs = [torch.empty(B, T, n, device=DEV) if i % 2 == 0 else torch.empty(B, T, n, n, device=DEV) for i in range(6)]
    st = N.States(*[o.data_ptr() for o in outs], None)
    assert lib.dll.kvae_lgssm_smooth_fwd(C.byref(prob), C.byref(st), None) == 0
    torch.cuda.synchronize()
    from oracle import c_oracle
    c = lambda t: t.cpu()
    ref = c_oracle.smooth(c(Y), c(U), None, c(A[0]), c(Bm[0]), c(Cm[0]), c(Q), c(R), c(mu0), c(S0))
    assert rel_err(outs[4].cpu(), ref["mus_smooth"]) < 1e-4
    prob.n = 17
    assert lib.dll.kvae_lgssm_smooth_fwd(C.byref(prob), C.byref(st), None) == 1      # KVAE_ERR_DIMS
    prob.n = n
    prob.Y = None
    assert lib.dll.kvae_lgssm_smooth_fwd(C.byref(prob), C.byref(st), None) == 2      # KVAE_ERR_NULL
    assert lib.dll.kvae_mix_fwd(None, None, None, 1, 1, 1, None) == 2
    assert lib.dll.kvae_mix_fwd(N.ptr(alpha), N.ptr(A), N.ptr(outs[1]), 1, 17, 1, None) == 4   # KVAE_ERR_ARG


@pytest.mark.parametrize("B,T,I,H", [(256, 50, 2, 50), (3, 7, 2, 50), (2, 5, 5, 13)])
def test_lstm_gpu(B, T, I, H):
    parity_cases.lstm_vs_torch(DEV, B, T, I, H)


@pytest.mark.parametrize("N,C,H,W,r,relu", [(64, 32, 8, 8, 2, True), (7, 1, 16, 16, 2, False), (33, 32, 16, 16, 1, True)])
def test_vae_epilogue_gpu(N, C, H, W, r, relu):
    parity_cases.vae_epilogue_vs_torch(DEV, N, C, H, W, r, relu)


@pytest.mark.parametrize("B,T,K,tau,hard", [(256, 50, 3, 1.0, False), (4, 100, 7, 0.5, False), (3, 10, 3, 0.7, True),
                                            (32, 100, 7, 0.7, True), (5, 9, 8, 1.3, False), (2, 1, 2, 0.5, False), (3, 4, 9, 1.0, False)])
def test_regime_gpu(B, T, K, tau, hard):
    parity_cases.regime_vs_torch(DEV, B, T, K, tau, hard)


@pytest.mark.parametrize("B,T", [(256, 50), (3, 7), (2, 1)])
def test_bigru_gpu(B, T):
    """kvae_bigru_fwd/bwd vs torch.nn.GRU(bidirectional) on the CPU: values and every gradient."""
    from kvae.kalman.lgssm_ops import BiGruSequence
    torch.manual_seed(B + T)
    ref = torch.nn.GRU(2, 50, batch_first=True, bidirectional=True)
    x = torch.randn(B, T, 2)
    wgt = torch.randn(B, T, 100)
    xr = x.clone().requires_grad_(True)
    (ref(xr)[0] * wgt).sum().backward()
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
             "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    params = [getattr(ref, n).detach().clone().to(DEV).requires_grad_(True) for n in names]
    xd = x.clone().to(DEV).requires_grad_(True)
    h = BiGruSequence.apply(xd, *params)
    (h * wgt.to(DEV)).sum().backward()
    assert rel_err(h.detach().cpu(), ref(x)[0].detach()) < 2e-5
    assert rel_err(xd.grad.cpu(), xr.grad) < 2e-4
    for n, got in zip(names, params):
        assert rel_err(got.grad.cpu(), getattr(ref, n).grad) < 2e-4, n


@pytest.mark.parametrize("N", [1, 37, 1000, 2000])
def test_conv_edge_gpu(N):
    """Direct decoder-head / encoder-stem kernels vs torch conv2d; 1000 / 2000 frames exercise the frame-strided weight-gradient loop."""
    parity_cases.conv_edge_vs_torch(DEV, N)


@pytest.mark.parametrize("N", [1, 37, 12800])
def test_vae_heads_gpu(N):
    """Fused encoder heads + reparameterisation, decoder fc, latent regulariser vs torch."""
    parity_cases.vae_heads_vs_torch(DEV, N)


@pytest.mark.parametrize("N,side", [(1, 8), (2, 8), (7, 8), (1031, 8), (1, 4), (8, 4), (13, 4), (4099, 4)])
def test_dec_up_gpu(N, side):
    """Register-stationary MFMA decoder blocks vs torch (ragged last iteration, more iterations than workgroups)."""
    parity_cases.dec_up_vs_torch(DEV, N, side)


@pytest.mark.parametrize("which,side", [("enc_mid", 8), ("dec_up", 4)])
def test_conv_chunked_launches_gpu(which, side, monkeypatch):
    """20000 frames = two launches (CHUNK = 16384).  They must equal ONE launch over all frames and five independent
    4000-frame calls (a size the kernels are checked at against the CPU above): outputs and data gradients bit-identical,
    weight gradients to fp32 summation order.  No external reference at this size: torch's CPU convolution backward
    (fp32 and fp64 disagree with each other above 16384 frames) and MIOpen's Winograd data gradient (8e-3 off) both
    proved unreliable here, while the three ways of running our kernels agree."""
    from kvae.vae.fused import DecoderUp, EncoderMid
    Fn = EncoderMid if which == "enc_mid" else DecoderUp
    N, co = 20000, (32 if which == "enc_mid" else 128)
    g = torch.Generator().manual_seed(7)
    x = torch.relu(torch.randn(N, 32, side, side, generator=g)).to(DEV)
    W = (0.08 * torch.randn(co, 32, 3, 3, generator=g)).to(DEV)
    b = (0.1 * torch.randn(co, generator=g)).to(DEV)
    o_side = side // 2 if which == "enc_mid" else 2 * side
    up = torch.linspace(-1, 1, N * 32 * o_side * o_side, device=DEV).view(N, 32, o_side, o_side)

    def run(slices):
        Ws, bs = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
        outs, gxs = [], []
        for lo, hi in slices:
            xs = x[lo:hi].clone().requires_grad_(True)
            out = Fn.apply(xs, Ws, bs)
            (out * up[lo:hi]).sum().backward()
            outs.append(out.detach())
            gxs.append(xs.grad)
        return torch.cat(outs), torch.cat(gxs), Ws.grad, bs.grad
    chunked = run([(0, N)])
    pieces = run([(i, i + 4000) for i in range(0, N, 4000)])
    monkeypatch.setattr(Fn, "CHUNK", 1 << 30)
    single = run([(0, N)])
    for other in (single, pieces):
        assert torch.equal(chunked[0], other[0]) and torch.equal(chunked[1], other[1])
        for a, r in zip(chunked[2:], other[2:]):
            assert rel_err(a.cpu(), r.cpu()) < 1e-4   # fp32 summation order over 3e5 cancelling terms


@pytest.mark.parametrize("N,side", [(1, 16), (2, 16), (7, 16), (1031, 16), (1, 8), (8, 8), (13, 8), (4099, 8)])
def test_enc_mid_gpu(N, side):
    """MFMA stride-2 encoder layers vs torch; odd frame counts exercise the ragged last iteration, the large ones the
    persistent loop (more iterations than workgroups)."""
    parity_cases.enc_mid_vs_torch(DEV, N, side)


def test_mix_gpu():
    """Mixture-of-K step records on the streaming kernels (one pass over the gradient records) vs torch.einsum autograd."""
    parity_cases.mix_vs_torch(DEV)


def test_rnn_wgrad_gpu():
    """No library GEMM on the path: LSTM / bi-GRU / head parameter gradients on the f32 matrix cores vs torch products."""
    parity_cases.rnn_wgrad_vs_torch(DEV)


def test_small_linear_gpu():
    parity_cases.small_linear_vs_torch(DEV)


@pytest.mark.parametrize("shape", [(256, 50, 1, 32, 32), (2, 3, 1, 32, 32), (1, 2, 3, 5, 7)])
def test_bce_frames_gpu(shape):
    parity_cases.bce_frames_vs_torch(DEV, *shape)


@pytest.mark.parametrize("K,T,n", [(3, 20, 4), (7, 60, 4), (2, 12, 16)])
def test_alpha_lstm_masked_matches_stepwise(K, T, n):
    """Masked frames with the lstm alpha-net: the in-kernel path (network inside the filter kernel, coupled adjoint in
    ONE backward launch) against the product's per-step differentiable path (T nn.LSTM cell steps + T single-step filter
    launches, autograd through all of it) - values, alpha, and every gradient.  The goldens masked_lstm_* pin the same
    path to the reference itself (test_latent_gpu)."""
    from kvae.kalman.lgssm_ops import LgssmSmooth, Slots
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    torch.manual_seed(K)
    model = KVAE(KVAEConfig(dynamics_model="lstm", num_modes=K, z_dim=n))
    with torch.no_grad():
        model.kalman_filter.dyn_params.A.add_(0.05 * torch.randn_like(model.kalman_filter.dyn_params.A))
        model.kalman_filter.dyn_params.head_w.bias.zero_()
        model.kalman_filter.dyn_params.head_w.weight.mul_(3.0)
    kf = model.kalman_filter.to(DEV).train()
    dyn = kf.dyn_params
    B = 5
    a = torch.randn(B, T, 2, device=DEV)
    u = 0.3 * torch.randn(B, T, n, device=DEV)
    mask = (torch.rand(B, T, device=DEV) > 0.4).float()
    mask[:, :3] = 1.0
    w_ms, w_Ss = torch.randn(B, T, n, 1, device=DEV), torch.randn(B, T, n, n, device=DEV)
    w_mf = torch.randn(B, T, n, 1, device=DEV)
    params = list(dyn.parameters())

    def loss_of(outs):
        ms, Ss, mf = outs[0], outs[1], outs[2]
        A_l, C_l = outs[6], outs[8]
        return (ms * w_ms).sum() + (Ss * w_Ss).sum() + (mf * w_mf).sum() + 0.1 * (A_l ** 2).sum() + 0.1 * (C_l ** 2).sum()

    a_f = a.clone().requires_grad_(True)
    dyn.reset_state()
    fast = kf.smooth(a_f, u, mask=mask)                       # in-kernel alpha-network
    alpha_fast = dyn.state_seq.detach().clone()
    g_fast = torch.autograd.grad(loss_of(fast), [a_f] + params, allow_unused=True)

    a_s = a.clone().requires_grad_(True)
    dyn.reset_state()
    mf, Sf, mp, Sp, A_l, B_l, C_l = kf._filter_stepwise(a_s, u, mask)     # reference-shaped per-step path
    ms, Ss, mf2, Sf2, mp2, Sp2 = LgssmSmooth.apply(a_s, u, mask, None, A_l, B_l, C_l, kf.Q, kf.R, kf.mu0, kf.Sigma0, Slots(), True)
    slow = (ms.unsqueeze(-1), Ss, mf2.unsqueeze(-1), Sf2, mp2.unsqueeze(-1), Sp2, A_l, B_l, C_l)
    alpha_slow = dyn.state_seq.detach()
    g_slow = torch.autograd.grad(loss_of(slow), [a_s] + params, allow_unused=True)

    assert rel_err(alpha_fast.cpu(), alpha_slow.cpu()) < 2e-4
    for f_, s_ in zip(fast, slow):
        assert rel_err(f_.detach().cpu(), s_.detach().cpu()) < 5e-4
    names = ["a"] + [k for k, _ in dyn.named_parameters()]
    for name, gf, gs in zip(names, g_fast, g_slow):
        assert (gf is None) == (gs is None), name
        if gf is not None:
            assert rel_err(gf.cpu(), gs.cpu()) < 3e-3, name


def test_explicit_ones_mask_equals_no_mask():
    """The reference's loop passes mask = ones (train.py:41): with lstm dynamics that takes the in-kernel alpha-network
    (no host-side inspection of the mask) and must reproduce the precomputed-alpha path of mask=None."""
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    torch.manual_seed(0)
    model = KVAE(KVAEConfig(dynamics_model="lstm", num_modes=3))
    with torch.no_grad():
        model.kalman_filter.dyn_params.A.add_(0.05 * torch.randn_like(model.kalman_filter.dyn_params.A))
        model.kalman_filter.dyn_params.head_w.bias.zero_()
    kf = model.kalman_filter.to(DEV).train()
    a = torch.randn(6, 15, 2, device=DEV)
    u = torch.zeros(6, 15, 4, device=DEV)
    outs = []
    for mk in (None, torch.ones(6, 15, device=DEV)):
        ar = a.clone().requires_grad_(True)
        kf.dyn_params.reset_state()
        o = kf.smooth(ar, u, mask=mk)
        g = torch.autograd.grad(o[0].sum() + (o[1] ** 2).sum(), [ar] + list(kf.dyn_params.parameters()), allow_unused=True)
        outs.append((o, g))
    for x, y in zip(outs[0][0], outs[1][0]):
        assert rel_err(y.detach().cpu(), x.detach().cpu()) < 2e-4
    for x, y in zip(outs[0][1], outs[1][1]):
        if x is not None:
            assert rel_err(y.cpu(), x.cpu()) < 2e-3


@pytest.mark.parametrize("env", [{"KVAE_WINO": "0"}, {"KVAE_Q4": "0"}, {"KVAE_N16": "0"}, {"KVAE_M4_SPLIT_MAX_B": "0"}])
def test_ab_switches_select_working_kernels(env):
    """The fault-isolation switches that remain (read once per process: direct instead of Winograd decoder blocks, one wavefront
    per sequence at n = 4, run-time-dimension kernels at n = 16 - each also the product's path for operands the specialised
    kernels cannot take) must select kernels that still pass parity: a fresh process per switch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r, %r]; import parity_cases as p; "
            "p.dec_up_vs_torch('cuda', 7, 8); p.dec_up_vs_torch('cuda', 13, 4); "
            "p.vs_oracle_random('cuda', 3, 10, 16, 16, 2, 2, dense_q=True); p.vs_oracle_random('cuda', 19, 11, 4, 4, 2, 3)"
            % (root, os.path.join(root, "kalman-vae_amd"), os.path.join(root, "tests")))
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


def test_m4_split_and_single_launch_forms_give_the_same_bits():
    """(4,4,2) below 2048 sequences: chain sweeps + per-step items (3 / 4 launches) against the single-launch form
    (KVAE_M4_SPLIT_MAX_B=0) on the same inputs, a fresh process each (the switch is read once): all six stacks and every gradient
    bit for bit - per-step Q, a mask, upstream gradients on all stacks."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path[:0] = [%r, %r, %r]; import torch, parity_cases as p\n"
            "from kvae.kalman.lgssm_ops import LgssmSmooth, Slots, mix_dynamics\n"
            "B, T, n, K = 37, 23, 4, 3\n"
            "A, Bm, Cm, alpha, Y, U, mask, _ = p._random_problem(B, T, n, n, 2, K, 4242, 'cuda')\n"
            "g = torch.Generator().manual_seed(9)\n"
            "qq = 0.05 * torch.randn(K, n, n, generator=g)\n"
            "Qk = (0.02 * torch.eye(n).repeat(K, 1, 1) + qq @ qq.mT).cuda()\n"
            "R, mu0, S0 = 0.03 * torch.eye(2).cuda(), (0.1 * torch.randn(n, generator=g)).cuda(), 2.0 * torch.eye(n).cuda()\n"
            "w = [(torch.randn(B, T, n, generator=g) if i %% 2 == 0 else torch.randn(B, T, n, n, generator=g)).cuda() for i in range(6)]\n"
            "leaves = [t.clone().requires_grad_(True) for t in (A, Bm, Qk, alpha, Y, U)]\n"
            "rec, offs, _ = mix_dynamics(leaves[3], leaves[:3])\n"
            "outs = LgssmSmooth.apply(leaves[4], leaves[5], mask, rec, None, None, Cm[0], None, R, mu0, S0, "
            "Slots(A=offs[0], B=offs[1], Q=offs[2]), True)\n"
            "sum((o * wi).sum() for o, wi in zip(outs, w)).backward()\n"
            "h = hashlib.sha256()\n"
            "[h.update(t.detach().cpu().numpy().tobytes()) for t in list(outs) + [l.grad for l in leaves]]\n"
            "print('DIGEST', h.hexdigest())\n" % (root, os.path.join(root, "kalman-vae_amd"), os.path.join(root, "tests")))
    digests = []
    for env in ({}, {"KVAE_M4_SPLIT_MAX_B": "0"}):
        r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert digests[0] == digests[1]
