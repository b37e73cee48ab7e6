"""Pin the torch oracle (oracle/torch_oracle.py) to the golden vectors captured from the reference
(tests/golden/make_goldens.py).  CPU only."""
import pytest
import torch

from golden_util import JITTER_CASES, JITTER_GRADS, LATENT_CASES, SMOOTH_KEYS, load, rel_err, stability_state_dict, sub
from oracle import torch_oracle as O

torch.set_num_threads(4)


@pytest.mark.parametrize("batch", [1, 4])
def test_rocket(batch):
    g = load(f"rocket_B{batch}")
    dyn = dict(A=g["A"], B=g["B"], C=g["C"], Q=g["Qk"])
    out = O.smooth_and_elbo(dyn, "switching", g["Y"], g["U"], None, None, g["R"], g["mu0"], g["Sigma0"],
                            g["eps_z"])
    for k in SMOOTH_KEYS:
        assert rel_err(out[k], g[k]) < 2e-6, k
    assert rel_err(out["elbo"], g["elbo"]) < 2e-6


@pytest.mark.parametrize("name,kind", LATENT_CASES)
def test_latent(name, kind):
    g = load(name)
    dyn = {k: v.clone().requires_grad_(True) for k, v in sub(g, "dyn.").items()}
    a = g["a"].clone().requires_grad_(True)
    kw = {}
    if kind == "switching":
        kw = dict(tau=float(g["tau"]), is_training=bool(g["train"]), gumbel=g["gumbel"],
                  trans_matrix=g["trans_matrix"])
    out = O.smooth_and_elbo(dyn, kind, a, g["u"], g["mask"], g["Qbuf"], g["R"], g["mu0"], g["Sigma0"],
                            g["eps_z"], **kw)
    tol = 5e-5 if "z16" in name else 1e-5
    if name == "stress_switch_z16_B2_T200":
        # error budget: the reference's own fp32 result is 3.4e-4 (means) away from an fp64 run of the same recursion (unstable
        # A = I + 0.05*randn at n=16 over T=200 amplifies rounding), so two fp32 implementations can only agree to twice that
        # order here.  Measured, not assumed: the fixture (the REFERENCE's fp32 run) and the oracle's fp32 run must each lie
        # within max(1e-4, 2 x that distance) of the oracle's fp64 run.
        import parity_cases
        o64, dist = parity_cases.latent_fp64_budget(g, kind, name)
        for k in ("mus_smooth", "mus_filt", "mus_pred", "elbo"):
            bar = max(1e-4, 2.0 * dist[k])
            assert rel_err(g[k].double(), o64[k]) < bar and rel_err(out[k].detach().double(), o64[k]) < bar, (k, dist[k])
        for k in ("Sigmas_smooth", "Sigmas_filt", "Sigmas_pred"):
            bar = max(1e-4, 2.0 * dist[k])
            assert rel_err(g[k + "_every8"].double(), o64[k][:, ::8]) < bar and rel_err(out[k].detach().double(), o64[k]) < bar, k
        assert dist["mus_smooth"] < 1e-3   # the budget itself is part of the record
        return
    for k in SMOOTH_KEYS + ["state_seq"]:
        if k in g:
            assert rel_err(out[k], g[k]) < tol, k
    for k in ("Sigmas_smooth", "Sigmas_filt", "Sigmas_pred"):
        if k + "_every8" in g:
            assert rel_err(out[k][:, ::8], g[k + "_every8"]) < tol, k
    assert rel_err(out["elbo"], g["elbo"]) < tol
    if "grad.a" in g:
        names = list(dyn)
        grads = torch.autograd.grad(-out["elbo"], [a] + [dyn[k] for k in names], allow_unused=True)
        assert rel_err(grads[0], g["grad.a"]) < 20 * tol
        for k, gr in zip(names, grads[1:]):
            ref = g["grad.dyn." + k]
            gr = torch.zeros_like(ref) if gr is None else gr
            if ref.abs().max() < 1e-12:
                assert gr.abs().max() < 1e-9, k
            else:
                assert rel_err(gr, ref) < 20 * tol, k


@pytest.mark.parametrize("name,kind,K", [("stability_lstm", "lstm", 3), ("stability_switching", "switching", 3),
                                         ("stability_lstm_K7_T100", "lstm", 7),
                                         ("stability_switching_K7_T100", "switching", 7)])
def test_stability_recipe(name, kind, K):
    """tests/test_imputation_stability.py:16-77 of the reference, re-run through the oracle."""
    g = load(name)
    T = int(g["T"])
    sd = stability_state_dict(kind, K)
    torch.manual_seed(123)
    x = torch.randn(2, T, 1, 32, 32)
    with torch.no_grad():
        out = O.kvae_impute(sd, x, g["mask"], kind=kind, eps_a=g["eps_a"], gumbel=g.get("gumbel"),
                            tau=1.0)
    assert rel_err(out["a_vae"], g["a_vae"]) < 1e-5
    sl = slice(None) if g["x_recon"].shape[1] == T else slice(None, None, 10)
    for k in ("x_recon", "x_imputed", "x_filtered"):
        assert (out[k][:, sl] - g[k]).abs().max() < 1e-6, k
    for k in ("a_imputed", "a_filtered"):
        assert (out[k] - g[k]).abs().max() < 1e-7, k


@pytest.mark.parametrize("name,kind", [("trainstep_lstm_K3", "lstm"), ("trainstep_switch_K3", "switching")])
def test_train_step(name, kind):
    g = load(name)
    sd = sub(g, "sd.")
    tr = O.OracleTrainer(sd, kind, lr=float(g["lr"]), clip=float(g["clip"]), beta=float(g["beta"]))
    x = g["frames"].float()
    out = tr.step(x, eps_a=g["eps_a"], eps_z=g["eps_z"], gumbel=g.get("gumbel"), mask=torch.ones(x.shape[:2]))
    for k, gk in (("loss", "loss"), ("elbo_kf", "elbo_kf"), ("elbo_vae_total", "elbo_vae")):
        assert rel_err(out[k], g[gk]) < 1e-5, k
    assert rel_err(out["grad_norm"], g["grad_norm"]) < 1e-4
    for k in tr.params:
        gn = tr.sd[k].grad.norm()
        assert abs(float(gn) - float(g["gradnorm." + k])) <= 2e-4 * float(g["gradnorm." + k]) + 1e-7, k
        if "after." + k in g:
            # Adam's first update is lr*g/(|g|+1e-8): entries whose gradient is O(1e-8) are rounding-
            # sensitive, hence 1e-3 (of max|param|) here while gradients themselves are held to 2e-4
            assert rel_err(tr.sd[k].detach(), g["after." + k]) < 1e-3, k


@pytest.mark.parametrize("name,kind", [("phases_lstm_K3", "lstm"), ("phases_switch_K3", "switching")])
def test_training_phases(name, kind):
    """The oracle's phases against the reference's own set_training_phase + train_one_epoch (two steps per phase, one Adam
    over all parameters; tests/golden/make_goldens_r3.py): epoch means, who moved, the parameters after each phase and Adam's
    per-parameter step counts."""
    import parity_cases
    g = load(name)
    tr = O.OracleTrainer(sub(g, "sd."), kind, lr=float(g["lr"]), clip=float(g["clip"]), beta=float(g["beta"]))

    def run_step(phase, i, kf_weight):
        x = g[f"frames{i}"].float()
        out = tr.step(x, eps_a=g[f"{phase}.eps_a{i}"], eps_z=g[f"{phase}.eps_z{i}"], gumbel=g.get(f"{phase}.gumbel{i}"),
                      mask=torch.ones(x.shape[:2]), kf_weight=kf_weight)
        return {k: out[k].detach() for k in ("loss", "elbo_kf", "elbo_vae_total")}

    parity_cases.check_phases(g, tr.set_training_phase, run_step, lambda: {k: tr.sd[k].detach() for k in tr.params},
                              lambda: [float(tr.opt.state[tr.sd[k]]["step"]) if tr.opt.state.get(tr.sd[k]) else 0.0
                                       for k in tr.params], value_tol=1e-5)


# ---- the C restatement (oracle/lgssm_oracle.c) against the same goldens ---------------------------
@pytest.mark.parametrize("name,kind", [c for c in LATENT_CASES if "K3_B4_T50" in c[0] or "z16" in c[0]
                                       or "masked_switch" in c[0] or "T12_u" in c[0]])
def test_c_oracle_latent(name, kind):
    from oracle import c_oracle
    g = load(name)
    have_all = all(k in g for k in ("A_list", "B_list", "C_list"))
    if not have_all:  # slim fixture: rebuild the per-step stacks with the torch oracle
        dyn = sub(g, "dyn.")
        o = O.lgssm_filter(g["a"], g["u"], g["mask"], dyn, kind, g["Qbuf"], g["R"], g["mu0"], g["Sigma0"],
                           tau=float(g["tau"]), is_training=bool(g["train"]), gumbel=g.get("gumbel"),
                           trans_matrix=g.get("trans_matrix"))
        A, Bm, Cm, Q = o["A_list"], o["B_list"], o["C_list"], o.get("Q_seq", g["Qbuf"])
    else:
        A, Bm, Cm, Q = g["A_list"], g["B_list"], g["C_list"], g.get("Q_seq", g["Qbuf"])
    out = c_oracle.smooth(g["a"], g["u"], g["mask"], A, Bm, Cm, Q, g["R"], g["mu0"], g["Sigma0"])
    tol = 2e-3 if name == "stress_switch_z16_B2_T200" else (5e-5 if "z16" in name else 1e-5)
    for k in ("mus_smooth", "mus_filt", "mus_pred"):
        assert rel_err(out[k].unsqueeze(-1), g[k]) < tol, k
    for k in ("Sigmas_smooth", "Sigmas_filt", "Sigmas_pred"):
        if k in g:
            assert rel_err(out[k], g[k]) < tol, k
        else:
            assert rel_err(out[k][:, ::8], g[k + "_every8"]) < tol, k
    terms, levels = c_oracle.elbo_terms(out["mus_smooth"], out["Sigmas_smooth"], g["eps_z"], g["a"], g["u"], g["mask"],
                                        A, Bm, Cm, Q, g["R"], g["mu0"], g["Sigma0"])
    total = terms.sum()
    if "log_qseq" in g:
        total += float(g["log_pseq"].double().sum() - g["log_qseq"].double().sum())
    elbo = total / max(float(g["mask"].sum()), 1.0)
    assert abs(elbo - float(g["elbo"])) <= max(tol, 3e-5) * abs(float(g["elbo"])), (elbo, float(g["elbo"]))
    assert list(levels) == [0, 0]


def test_c_oracle_rocket():
    from oracle import c_oracle
    g = load("rocket_B4")
    out = c_oracle.smooth(g["Y"], g["U"], None, g["A"][0], g["B"][0], g["C"][0], g["Qk"][0], g["R"], g["mu0"], g["Sigma0"])
    for k in ("mus_smooth", "mus_filt", "mus_pred"):
        assert rel_err(out[k].unsqueeze(-1), g[k]) < 2e-5, k
    for k in ("Sigmas_smooth", "Sigmas_filt", "Sigmas_pred"):
        assert rel_err(out[k], g[k]) < 2e-5, k
    terms, _ = c_oracle.elbo_terms(out["mus_smooth"], out["Sigmas_smooth"], g["eps_z"], g["Y"], g["U"], None,
                                   g["A"][0], g["B"][0], g["C"][0], g["Qk"][0], g["R"], g["mu0"], g["Sigma0"])
    assert abs(terms.sum() / g["Y"].shape[0] / g["Y"].shape[1] - float(g["elbo"])) < 2e-5 * abs(float(g["elbo"]))


# ---- _safe_cholesky past level 0 (kalman_filter.py:282-302): fixtures driven through the reference's own elbo ----
@pytest.mark.parametrize("name,levels", JITTER_CASES)
def test_jitter_ladder_torch_oracle(name, levels):
    g = load(name)
    # the reference made (level+1) attempts per matrix family, 5 when it fell through to the diagonal
    assert int(g["cholesky_attempts"]) == sum(min(l + 1, 5) for l in levels)
    leaves = {k: g[k].clone().requires_grad_(True) for k in JITTER_GRADS}
    elbo = O.lgssm_elbo(leaves["mu_s"], leaves["Sig_s"], leaves["a"], g["u"], leaves["A_list"], leaves["B_list"],
                        leaves["C_list"], leaves["Q_list"], g["R"], g["mu0"], g["Sigma0"], g["mask"], g["eps_z"])
    assert rel_err(elbo.detach(), g["elbo"]) < 1e-5
    grads = torch.autograd.grad(-elbo, list(leaves.values()), allow_unused=True)
    for k, gr in zip(leaves, grads):
        ref = g["grad." + k]
        gr = torch.zeros_like(ref) if gr is None else gr
        assert rel_err(gr, ref) < 2e-4, k


@pytest.mark.parametrize("name,levels", JITTER_CASES)
def test_jitter_ladder_c_oracle(name, levels):
    from oracle import c_oracle
    g = load(name)
    terms, lv = c_oracle.elbo_terms(g["mu_s"], g["Sig_s"], g["eps_z"], g["a"], g["u"], g["mask"], g["A_list"],
                                    g["B_list"], g["C_list"], g["Q_list"], g["R"], g["mu0"], g["Sigma0"])
    assert list(lv) == levels
    elbo = terms.sum() / max(float(g["mask"].sum()), 1.0)
    assert abs(elbo - float(g["elbo"])) <= 3e-5 * abs(float(g["elbo"])), (elbo, float(g["elbo"]))
