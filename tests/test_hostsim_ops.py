"""CPU tier: run the product op layer (kvae.kalman.*, autograd Functions, ctypes structs) against the
HOST SIMULATION of the kernel bodies (tests/hostsim) and compare with the goldens captured from the
reference.  This checks the kernel arithmetic, the hand-derived backward and the host logic without
a GPU; the same comparisons run against the real gfx950 library in tests/test_gpu_parity.py."""
import pytest
import torch

from golden_util import JITTER_CASES, LATENT_CASES, SMOOTH_KEYS, load, rel_err, sub
from hostsim.build import build as build_hostsim

torch.set_num_threads(4)


@pytest.fixture(scope="module", autouse=True)
def hostsim_backend():
    from kvae import _native
    lib = _native.LgssmLib(build_hostsim())
    _native._set_test_backend(lib)
    yield lib
    _native._set_test_backend(None)


def make_filter(g, kind, device="cpu"):
    """Product KalmanFilter + dynamics module carrying the golden's parameters."""
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    dyn = sub(g, "dyn.")
    K, n = dyn["A"].shape[0], dyn["A"].shape[1]
    model = KVAE(KVAEConfig(dynamics_model=kind, num_modes=K, z_dim=n))
    kf = model.kalman_filter
    missing = kf.dyn_params.load_state_dict(dyn, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    kf.load_state_dict({"Q": g["Qbuf"], "R": g["R"], "mu0": g["mu0"], "Sigma0": g["Sigma0"]}, strict=False)
    if "tau" in g and hasattr(kf.dyn_params, "tau"):
        kf.dyn_params.tau = float(g["tau"])
    kf.train(bool(g["train"]))
    return kf.to(device)


def run_latent(kf, g, device="cpu"):
    """Fixtures without gradients run under no_grad, as the reference produced them (eval / imputation): with the
    lstm alpha-net and hidden frames that is the path with the network INSIDE the filter kernel."""
    import contextlib
    from kvae import noise
    need_grad = "grad.a" in g
    a = g["a"].to(device).clone().requires_grad_(need_grad)
    u, mask = g["u"].to(device), g["mask"].to(device)
    kf.dyn_params.reset_state()
    with noise.inject(eps_z=g["eps_z"], gumbel=g.get("gumbel")), (contextlib.nullcontext() if need_grad else torch.no_grad()):
        outs = kf.smooth(a, u, mask=None if bool((mask == 1).all()) else mask)
        elbo = kf.elbo(outs[0], outs[1], a, u, outs[6], outs[7], outs[8], mask=mask)
    return a, outs, elbo


def check_latent(kf, g, a, outs, elbo, name, tol_scale=1.0, kind=None):
    tol = (5e-5 if "z16" in name else 1e-5) * tol_scale
    if name == "stress_switch_z16_B2_T200":
        # n = 16 over T = 200 with an unstable A: float32 itself is the limit (the reference's own float32 fixture is 3.4e-4 from
        # a float64 run of the recursion).  The bar, as an assertion: within max(1e-4, 2 x the float32 distance - the larger of the
        # reference fixture's and the oracle's own) of the FLOAT64 oracle, per output - not a flat 2e-3 against the float32 fixture.
        import parity_cases
        o64, dist = parity_cases.latent_fp64_budget(g, kind or "switching", name)
        for k, v in zip(SMOOTH_KEYS, outs):
            if k in ("A_list", "B_list", "C_list"):
                continue
            want = o64[k].squeeze(-1) if k.startswith("mus") else o64[k]
            assert rel_err(v.detach().cpu().double().reshape(want.shape), want) < max(1e-4, 2.0 * dist[k]), (k, dist[k])
        assert rel_err(elbo.detach().cpu().double(), o64["elbo"]) < max(1e-4, 2.0 * dist["elbo"]), ("elbo", dist["elbo"])
        assert rel_err(kf.dyn_params.state_seq.cpu(), g["state_seq"]) < 1e-5
        return
    for k, v in zip(SMOOTH_KEYS, outs):
        if k in g:
            assert rel_err(v.cpu(), g[k]) < tol, k
        elif k + "_every8" in g:
            assert rel_err(v.cpu()[:, ::8], g[k + "_every8"]) < tol, k
    assert rel_err(kf.dyn_params.state_seq.cpu(), g["state_seq"]) < tol
    assert rel_err(elbo.cpu(), g["elbo"]) < max(tol, 2e-5), "elbo"
    if "a_imputed" in g:   # what KVAE.impute decodes (model.py:279-288): C_t mu_t|T and C_t mu_t|t, fused read-out
        a_imp, a_fil = kf.emission_means(outs[0], outs[2], outs[8])
        assert rel_err(a_imp.cpu(), g["a_imputed"]) < 1e-4 and rel_err(a_fil.cpu(), g["a_filtered"]) < 1e-4
    if "grad.a" not in g:
        return
    params = dict(kf.dyn_params.named_parameters())
    grads = torch.autograd.grad(-elbo, [a] + list(params.values()), allow_unused=True)
    gtol = 20 * tol
    assert rel_err(grads[0].cpu(), g["grad.a"]) < gtol, "grad.a"
    for (k, p), gr in zip(params.items(), grads[1:]):
        ref = g["grad.dyn." + k]
        gr = torch.zeros_like(ref) if gr is None else gr.cpu()
        if ref.abs().max() < 1e-12:
            assert gr.abs().max() < 1e-9, k
        else:
            assert rel_err(gr, ref) < gtol, k


@pytest.mark.parametrize("name,kind", LATENT_CASES)
def test_latent_hostsim(name, kind):
    g = load(name)
    kf = make_filter(g, kind)
    a, outs, elbo = run_latent(kf, g)
    check_latent(kf, g, a, outs, elbo, name, kind=kind)


@pytest.mark.parametrize("batch", [1, 4])
def test_rocket_hostsim(batch):
    """n=2, m=1, p=1: the run-time-dimension instantiation of the kernels."""
    from kvae import noise
    from kvae.kalman.kalman_filter import KalmanFilter
    from kvae.kalman.switch_dyn_param import SwitchingDynamicsParameter
    g = load(f"rocket_B{batch}")
    dyn = SwitchingDynamicsParameter(g["A"], g["B"], g["C"])
    kf = KalmanFilter(2.0, 4.0, g["mu0"], g["Sigma0"], dyn)
    with torch.no_grad(), noise.inject(eps_z=g["eps_z"]):
        outs = kf.smooth(g["Y"], g["U"])
        elbo = kf.elbo(outs[0], outs[1], g["Y"], g["U"], outs[6], outs[7], outs[8])
    for k, v in zip(SMOOTH_KEYS, outs):
        assert rel_err(v, g[k]) < 2e-5, k
    assert rel_err(elbo, g["elbo"]) < 2e-5


def test_no_cpu_fallback():
    """Without the injected simulator the product refuses host tensors."""
    from kvae import _native
    saved = _native._test_backend
    _native._set_test_backend(None)
    try:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            _native.lib_for(torch.zeros(1))
    finally:
        _native._set_test_backend(saved)


@pytest.mark.parametrize("B,T,n,m,p,K", [(7, 33, 4, 4, 2, 3), (5, 17, 3, 2, 1, 2), (3, 9, 8, 5, 3, 4), (2, 12, 16, 16, 2, 3),
                                         (1, 1, 4, 4, 2, 3), (2, 2, 16, 16, 2, 1)])
def test_vs_oracle_random_hostsim(B, T, n, m, p, K):
    import parity_cases
    parity_cases.vs_oracle_random("cpu", B, T, n, m, p, K)


@pytest.mark.parametrize("B,T", [(17, 5), (3, 10)])
def test_single_mode_n16_vs_fp64_oracle_hostsim(B, T):
    import parity_cases
    parity_cases.grads_vs_fp64_oracle("cpu", B, T, 16, 1)


def test_dec_up_workgroup_cap_hostsim():
    import parity_cases
    parity_cases.dec_up_workgroup_cap("cpu")


def test_colsum_pair_hostsim():
    import parity_cases
    parity_cases.colsum_pair_vs_torch("cpu")


def test_linearity_hostsim():
    import parity_cases
    parity_cases.linearity("cpu", 16, 20)


def test_safe_cholesky_levels_hostsim():
    import parity_cases
    parity_cases.safe_cholesky_levels("cpu")
    parity_cases.safe_cholesky_levels_shared_q("cpu", n=4, B=3, T=6)


@pytest.mark.parametrize("name,levels", JITTER_CASES)
def test_jitter_golden_hostsim(name, levels):
    import parity_cases
    parity_cases.jitter_golden("cpu", name, levels)


@pytest.mark.parametrize("B,T,I,H", [(3, 7, 2, 50), (2, 5, 5, 13), (1, 1, 2, 50)])
def test_lstm_hostsim(B, T, I, H):
    import parity_cases
    parity_cases.lstm_vs_torch("cpu", B, T, I, H)


@pytest.mark.parametrize("N,C,H,W,r,relu", [(3, 8, 4, 4, 2, True), (2, 1, 16, 16, 2, False), (5, 6, 8, 8, 1, True), (2, 3, 5, 7, 3, True)])
def test_vae_epilogue_hostsim(N, C, H, W, r, relu):
    import parity_cases
    parity_cases.vae_epilogue_vs_torch("cpu", N, C, H, W, r, relu)


def test_vae_fused_matches_unfused_hostsim():
    """Encoder/Decoder with the fused epilogues == the plain nn.Sequential path (same weights)."""
    from kvae import _native
    from kvae.utils.config import KVAEConfig
    from kvae.vae.vae import Decoder, Encoder
    torch.manual_seed(0)
    cfg = KVAEConfig()
    enc, dec = Encoder(cfg), Decoder(cfg)
    x, a = torch.rand(6, 1, 32, 32), torch.randn(6, 2)
    mu_f, var_f = enc(x)
    out_f = dec(a)
    saved = _native._test_backend
    _native._set_test_backend(None)
    try:
        mu_p, var_p = enc(x)
        out_p = dec(a)
    finally:
        _native._set_test_backend(saved)
    assert rel_err(mu_f, mu_p) < 1e-6 and rel_err(var_f, var_p) < 1e-6 and rel_err(out_f, out_p) < 1e-6


@pytest.mark.parametrize("B,T,K,tau,hard", [(3, 9, 3, 1.0, False), (2, 20, 7, 0.5, False), (2, 6, 3, 0.7, True), (1, 1, 2, 1.0, False)])
def test_regime_hostsim(B, T, K, tau, hard):
    import parity_cases
    parity_cases.regime_vs_torch("cpu", B, T, K, tau, hard)


@pytest.mark.parametrize("N", [1, 5])
def test_conv_edge_hostsim(N):
    import parity_cases
    parity_cases.conv_edge_vs_torch("cpu", N)


@pytest.mark.parametrize("N", [1, 50])
def test_vae_heads_hostsim(N):
    import parity_cases
    parity_cases.vae_heads_vs_torch("cpu", N)


@pytest.mark.parametrize("N,side", [(3, 8), (9, 4)])
def test_dec_up_hostsim(N, side):
    import parity_cases
    parity_cases.dec_up_vs_torch("cpu", N, side)


def test_conv_chunked_launches_hostsim(monkeypatch):
    """Frame counts above CHUNK are split into several launches whose weight gradients are summed."""
    import parity_cases
    from kvae.vae.fused import DecoderUp, EncoderMid
    monkeypatch.setattr(EncoderMid, "CHUNK", 2)
    monkeypatch.setattr(DecoderUp, "CHUNK", 3)
    parity_cases.enc_mid_vs_torch("cpu", 5, 8)
    parity_cases.dec_up_vs_torch("cpu", 7, 4)


@pytest.mark.parametrize("N,side", [(3, 16), (9, 8)])
def test_enc_mid_hostsim(N, side):
    import parity_cases
    parity_cases.enc_mid_vs_torch("cpu", N, side)


@pytest.mark.parametrize("shape", [(2, 3, 1, 32, 32), (1, 2, 3, 5, 7)])
def test_bce_frames_hostsim(shape):
    import parity_cases
    parity_cases.bce_frames_vs_torch("cpu", *shape)


@pytest.mark.parametrize("clip,wd,div", [(0.5, 0.0, None), (0.0, 1e-2, None), (10.0, 1e-3, 37.0)])
def test_clip_adam_c_abi_vs_torch(hostsim_backend, clip, wd, div):
    """kvae_clip_adam through the C ABI (host twin of the two GPU launches) against clip_grad_norm_ + torch.optim.Adam on the same
    flat buffers: three steps, optional weight decay, optional division by a frame count (the multi-rank path)."""
    from kvae import _native as N
    g = torch.Generator().manual_seed(3)
    n = 1000
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=3e-3, weight_decay=wd)
    p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    step, norm, ws = torch.zeros(1), torch.zeros(()), torch.empty(1024)
    cnt = torch.tensor([div]) if div else None
    for it in range(3):
        grad = torch.randn(n, generator=g) * (it + 1)
        gref = grad / div if div else grad.clone()
        ref.grad = gref.clone()
        total = torch.nn.utils.clip_grad_norm_([ref], clip) if clip > 0 else gref.norm()
        opt.step()
        rc = hostsim_backend.dll.kvae_clip_adam(N.ptr(p), N.ptr(grad), N.ptr(m), N.ptr(v), n, None, 1, None, N.ptr(step), None, 3e-3,
                                               0.9, 0.999, 1e-8, wd, clip, N.ptr(cnt) if cnt is not None else None, N.ptr(norm),
                                               N.ptr(ws), None)
        assert rc == 0
        assert abs(float(norm) - float(total)) <= 1e-5 * float(total)
    assert float(step) == 3.0
    assert float((p - ref.detach()).abs().max()) < 2e-6
    st = opt.state[ref]
    assert rel_err(m, st["exp_avg"]) < 1e-5 and rel_err(v, st["exp_avg_sq"]) < 5e-5   # (v squares the clip scale: twice its rounding)


def test_clip_adam_frozen_segments_vs_torch(hostsim_backend):
    """Three parameter tensors in one flat buffer, the middle one frozen for the first two of four steps (requires_grad False: the
    reference's training phases, train.py:142-207) - clip_grad_norm_ + torch.optim.Adam skip a parameter whose grad is None (no
    norm contribution, no moment / step / value update, bias correction from ITS OWN step count once it thaws), and so must the
    segment mask of kvae_clip_adam."""
    from kvae import _native as N
    g = torch.Generator().manual_seed(8)
    sizes = [7, 300, 41]
    refs = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in sizes]
    opt = torch.optim.Adam(refs, lr=2e-3)
    n = sum(sizes)
    p = torch.cat([r.detach().clone() for r in refs])
    m, v = torch.zeros(n), torch.zeros(n)
    seg_of = torch.repeat_interleave(torch.arange(3, dtype=torch.int32), torch.tensor(sizes))
    steps, norm, ws = torch.zeros(3), torch.zeros(()), torch.empty(1024)
    for it in range(4):
        frozen = it < 2
        active = torch.tensor([1.0, 0.0 if frozen else 1.0, 1.0])
        grads = [torch.randn(s, generator=g) * 3 for s in sizes]
        for r, gr, a in zip(refs, grads, active):
            r.grad = gr.clone() if a else None
        total = torch.nn.utils.clip_grad_norm_(refs, 1.5)
        opt.step()
        flat_g = torch.cat(grads)   # the frozen slot holds garbage on purpose: the mask, not a zero gradient, must exclude it
        rc = hostsim_backend.dll.kvae_clip_adam(N.ptr(p), N.ptr(flat_g), N.ptr(m), N.ptr(v), n, N.ptr(seg_of), 3, N.ptr(active),
                                               N.ptr(steps), None, 2e-3, 0.9, 0.999, 1e-8, 0.0, 1.5, None, N.ptr(norm), N.ptr(ws), None)
        assert rc == 0
        assert abs(float(norm) - float(total)) <= 1e-5 * float(total)
        if frozen:
            assert torch.equal(p[7:307], refs[1].detach()) and float(m[7:307].abs().max()) == 0.0   # bit-identical, no state
    assert steps.tolist() == [4.0, 2.0, 4.0]
    assert float((p - torch.cat([r.detach() for r in refs])).abs().max()) < 2e-6
    assert [float(opt.state[r]["step"]) for r in refs] == [4.0, 2.0, 4.0]
    assert rel_err(m, torch.cat([opt.state[r]["exp_avg"] for r in refs])) < 1e-5


def test_rnn_wgrad_and_small_linear_hostsim(hostsim_backend):
    import parity_cases
    parity_cases.mix_vs_torch("cpu")
    parity_cases.rnn_wgrad_vs_torch("cpu")
    parity_cases.small_linear_vs_torch("cpu")


@pytest.mark.parametrize("name,kind", [("phases_lstm_K3", "lstm"), ("phases_switch_K3", "switching")])
def test_training_phases_hostsim(name, kind, hostsim_backend):
    """The Trainer's phases (requires_grad toggles, loss weights, frozen parameters skipped by the optimizer) on the host tier:
    the kernel bodies on the CPU, torch's own Adam (which skips grad None exactly as the GPU path's slot mask does), against
    the reference's recorded vae -> warmup -> all run."""
    import parity_cases
    from kvae import noise
    from kvae.model.model import KVAE
    from kvae.train.train import Trainer
    from kvae.utils.config import KVAEConfig
    g = load(name)
    model = KVAE(KVAEConfig(dynamics_model=kind, num_modes=3, scheduled_beta=False))
    model.load_state_dict(sub(g, "sd."), strict=True)
    model.train()
    model.beta = float(g["beta"])
    tr = Trainer(model, lr=float(g["lr"]), grad_clip_norm=float(g["clip"]), use_graph=False)
    assert not tr._flat_step

    def run_step(phase, i, kf_weight):
        assert tr.kf_weight == kf_weight
        with noise.inject(eps_a=g[f"{phase}.eps_a{i}"], eps_z=g[f"{phase}.eps_z{i}"], gumbel=g.get(f"{phase}.gumbel{i}")):
            out = tr.step(g[f"frames{i}"].float())
        return {k: float(out[k]) for k in ("loss", "elbo_kf", "elbo_vae_total")}

    def steps():
        return [float(tr.opt.state[p]["step"]) if tr.opt.state.get(p) else 0.0 for p in model.parameters()]

    parity_cases.check_phases(g, tr.set_training_phase, run_step, lambda: {k: p.detach().clone() for k, p in model.named_parameters()},
                              steps, value_tol=1e-4)
