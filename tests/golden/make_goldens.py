#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by importing the
reference implementation from /root/reference (CPU, this container only).

The reference never travels to the GPU box; only the .npz files written here do.
Each fixture holds the *inputs* (parameters, observations, masks, and every random
draw the reference consumed: reparameterisation eps, smoothed-state eps, Gumbel
noise) and the *outputs* the reference produced from them, so that the oracle
(oracle/) and the HIP path can be fed byte-identical inputs.

Run:  python tests/golden/make_goldens.py            (writes tests/golden/*.npz)

Import shim: kvae/vae/losses.py:4 of the reference imports `kvae.vae.config`, a module
that does not exist in the snapshot; aliasing it to kvae.utils.config (an ordinary
ModuleNotFoundError workaround) makes kvae.model.model importable.
"""
import contextlib
import os
import sys
from pathlib import Path

import numpy as np
import torch

REF = os.environ.get("KVAE_REFERENCE", "/root/reference")
OUT = Path(__file__).resolve().parent

os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)
import kvae.utils.config as _cfg  # noqa: E402

sys.modules["kvae.vae.config"] = _cfg
from kvae.kalman.dyn_param import DynamicsParameter  # noqa: E402
from kvae.kalman.kalman_filter import KalmanFilter  # noqa: E402
from kvae.kalman import switch_dyn_param as ref_switch  # noqa: E402
from kvae.kalman.switch_dyn_param import SwitchingDynamicsParameter  # noqa: E402
from kvae.model.model import KVAE  # noqa: E402
from kvae.utils.config import KVAEConfig  # noqa: E402
import torch.distributions.multivariate_normal as _mvn_mod  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.set_num_threads(4)


# --------------------------------------------------------------------------------------
# noise capture: record every random draw the reference makes, in call order
# --------------------------------------------------------------------------------------
class NoiseTape:
    def __init__(self):
        self.randn_like = []      # KVAE.reparameterize (model.py:81-84)
        self.std_normal = []      # MultivariateNormal.rsample (kalman_filter.py:351)
        self.gumbel = []          # F.gumbel_softmax (switch_dyn_param.py:52,69)


@contextlib.contextmanager
def record_noise():
    tape = NoiseTape()
    orig_randn_like = torch.randn_like
    orig_std_normal = _mvn_mod._standard_normal
    orig_gumbel = ref_switch.gumbel_softmax

    def randn_like(t, *a, **k):
        out = orig_randn_like(t, *a, **k)
        tape.randn_like.append(out.detach().clone())
        return out

    def std_normal(shape, dtype, device):
        out = orig_std_normal(shape, dtype, device)
        tape.std_normal.append(out.detach().clone())
        return out

    def gumbel_softmax(logits, tau=1, hard=False, eps=1e-10, dim=-1):
        # replay the RNG to recover the exact Gumbel noise F.gumbel_softmax draws
        st0 = torch.get_rng_state()
        out = orig_gumbel(logits, tau=tau, hard=hard, dim=dim)
        st1 = torch.get_rng_state()
        torch.set_rng_state(st0)
        g = -torch.empty_like(logits, memory_format=torch.legacy_contiguous_format).exponential_().log()
        torch.set_rng_state(st1)
        # self-check: same formula reproduces the soft sample
        soft = ((logits + g) / tau).softmax(dim)
        if not hard:
            assert torch.equal(soft.detach(), out.detach()), "gumbel replay mismatch"
        tape.gumbel.append(g.detach().clone())
        return out

    torch.randn_like = randn_like
    _mvn_mod._standard_normal = std_normal
    ref_switch.gumbel_softmax = gumbel_softmax
    try:
        yield tape
    finally:
        torch.randn_like = orig_randn_like
        _mvn_mod._standard_normal = orig_std_normal
        ref_switch.gumbel_softmax = orig_gumbel


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **arrays):
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"  wrote {path.name}: {path.stat().st_size / 1024:.1f} KiB, {len(arrays)} arrays")


def sd_arrays(module, prefix="sd."):
    return {prefix + k: v for k, v in module.state_dict().items()}


SMOOTH_KEYS = ["mus_smooth", "Sigmas_smooth", "mus_filt", "Sigmas_filt",
               "mus_pred", "Sigmas_pred", "A_list", "B_list", "C_list"]


# --------------------------------------------------------------------------------------
# 1. known dynamics ("rocket", recipe of kvae/kalman/test_filter.py:5-69, own seed)
# --------------------------------------------------------------------------------------
def rocket(batch, seed):
    rng = np.random.default_rng(seed)
    dt, g, N = 0.1, -9.81, 100
    t = np.arange(N) * dt
    std_obs, std_dyn = 4.0, 2.0
    Ys, Us = [], []
    for b in range(batch):
        x = np.zeros((N, 2))
        thrust = 20.0 + 2.0 * b
        for n in range(N - 1):
            a = (thrust if t[n] < 6.0 else 0.0) + g
            x[n + 1, 0] = x[n, 0] + x[n, 1] * dt + 0.5 * a * dt * dt
            x[n + 1, 1] = x[n, 1] + a * dt
        acc = (x[1:, 1] - x[:-1, 1]) / dt - g
        a_spec = np.r_[acc[0], acc]
        Us.append(a_spec + g + rng.standard_normal(N) * std_dyn ** 2)
        Ys.append(x[:, 0] + rng.standard_normal(N) * std_obs ** 2)
    A = torch.tensor([[1.0, dt], [0.0, 1.0]])
    Bm = torch.tensor([[0.5 * dt ** 2], [dt]])
    C = torch.tensor([[1.0, 0.0]])
    # StickyRegimePrior(K=1) divides by zero (switch_dyn_param.py:102), so the reference's own
    # rocket script cannot build its default prior; hand it a trivial one (unused when K == 1).
    class _Prior:
        transition_matrix = torch.ones(1, 1)
    dyn = SwitchingDynamicsParameter(A[None], Bm[None], C[None], prior=_Prior())
    kf = KalmanFilter(std_dyn, std_obs, torch.zeros(2), torch.eye(2), dyn)
    Y = torch.tensor(np.stack(Ys), dtype=torch.float32)[..., None]
    U = torch.tensor(np.stack(Us), dtype=torch.float32)[..., None]
    torch.manual_seed(seed)
    with record_noise() as tape, torch.no_grad():
        outs = kf.smooth(Y, U)
        elbo = kf.elbo(outs[0], outs[1], Y, U, outs[6], outs[7], outs[8])
    save(f"rocket_B{batch}", Y=Y, U=U, A=dyn.A, B=dyn.B, C=dyn.C, Qk=dyn.Q,
         R=kf.R, mu0=kf.mu0, Sigma0=kf.Sigma0, eps_z=tape.std_normal[0], elbo=elbo,
         **dict(zip(SMOOTH_KEYS, outs)))


# --------------------------------------------------------------------------------------
# 2/3/6. latent path of a default-initialised KVAE (A=I, B,C ~ 0.05 randn), with grads
# --------------------------------------------------------------------------------------
def build_kf(dynamics, K, z_dim=4, a_dim=2, seed=0, perturb=0.0):
    """The LGSSM part of reference KVAE.__init__ (model.py:28-78), default init."""
    torch.manual_seed(seed)
    cfg = KVAEConfig(dynamics_model=dynamics, num_modes=K, z_dim=z_dim, a_dim=a_dim)
    model = KVAE(cfg)
    kf = model.kalman_filter
    if perturb > 0:  # make the K modes distinct and the alpha-net non-trivial
        with torch.no_grad():
            kf.dyn_params.A.add_(perturb * torch.randn_like(kf.dyn_params.A))
            if hasattr(kf.dyn_params, "head_w"):
                kf.dyn_params.head_w.bias.zero_()
                kf.dyn_params.head_w.weight.mul_(3.0)
            if hasattr(kf.dyn_params, "Q"):
                q = 0.3 * perturb * torch.randn_like(kf.dyn_params.Q)
                kf.dyn_params.Q.add_(q @ q.mT)
    return cfg, kf


def latent_case(name, dynamics, K, B, T, z_dim=4, seed=0, mask_block=None, train=True,
                with_grads=True, tau=None, slim=False):
    cfg, kf = build_kf(dynamics, K, z_dim=z_dim, seed=seed, perturb=0.05)
    if tau is not None and hasattr(kf.dyn_params, "tau"):
        kf.dyn_params.tau = tau
    kf.train(train)
    g = torch.Generator().manual_seed(1000 + seed)
    # a smooth-ish 2-D trajectory plus noise, like encoder outputs
    tt = torch.linspace(0, 3.0, T)[None, :, None]
    phase = torch.rand(B, 1, 2, generator=g) * 6.28
    a = 0.8 * torch.sin(tt * (1.0 + torch.rand(B, 1, 2, generator=g)) + phase) \
        + 0.1 * torch.randn(B, T, 2, generator=g)
    a = a.clone().requires_grad_(with_grads)
    u = torch.zeros(B, T, cfg.u_dim)
    if name.endswith("_u"):
        u = 0.5 * torch.randn(B, T, cfg.u_dim, generator=g)
    mask = torch.ones(B, T)
    if mask_block is not None:
        mask[:, mask_block[0]:mask_block[1]] = 0.0
    kf.dyn_params.reset_state()
    torch.manual_seed(77 + seed)
    ctx = contextlib.nullcontext() if with_grads else torch.no_grad()
    with record_noise() as tape, ctx:
        outs = kf.smooth(a, u, mask=mask)
        elbo = kf.elbo(outs[0], outs[1], a, u, outs[6], outs[7], outs[8], mask=mask)
    arrays = dict(a=a, u=u, mask=mask, elbo=elbo, R=kf.R, Qbuf=kf.Q, mu0=kf.mu0, Sigma0=kf.Sigma0,
                  eps_z=tape.std_normal[0], state_seq=kf.dyn_params.state_seq,
                  tau=getattr(kf.dyn_params, "tau", 0.0), train=int(train))
    arrays.update(dict(zip(SMOOTH_KEYS, outs)))
    arrays.update(sd_arrays(kf.dyn_params, "dyn."))
    if tape.gumbel:
        arrays["gumbel"] = torch.stack(tape.gumbel, 1)          # [B,T,K]
        arrays["log_qseq"], arrays["log_pseq"] = kf.dyn_params.elbo_terms()
        arrays["Q_seq"] = kf.dyn_params.Q_seq
        arrays["trans_matrix"] = kf.dyn_params.prior.transition_matrix
    if slim:  # long/wide cases: keep every 8th covariance, drop recomputable stacks
        for k in ("A_list", "B_list", "C_list", "Q_seq"):
            arrays.pop(k, None)
        for k in ("Sigmas_smooth", "Sigmas_filt", "Sigmas_pred"):
            arrays[k + "_every8"] = arrays.pop(k)[:, ::8]
    if with_grads:
        params = dict(kf.dyn_params.named_parameters())
        grads = torch.autograd.grad(-elbo, [a] + list(params.values()), allow_unused=True)
        arrays["grad.a"] = grads[0]
        for (k, p), gr in zip(params.items(), grads[1:]):
            arrays["grad.dyn." + k] = gr if gr is not None else torch.zeros_like(p)
    save(name, **arrays)


# --------------------------------------------------------------------------------------
# 4. the reference's own stability-test recipe (tests/test_imputation_stability.py:16-77)
# --------------------------------------------------------------------------------------
def stability_case(name, dynamics, K=3, T=10, full_images=True):
    cfg = KVAEConfig(dynamics_model=dynamics, num_modes=K)
    model = KVAE(cfg)
    torch.manual_seed(42)
    for p in model.parameters():
        if p.requires_grad:
            p.data = torch.randn_like(p.data) * 0.01
    model.eval()
    torch.manual_seed(123)
    x = torch.randn(2, T, 1, 32, 32)
    mask = torch.ones(2, T)
    mask[:, 4:10] = 0.0
    with record_noise() as tape, torch.no_grad():
        out = model.impute(x, mask=mask)
    arrays = dict(mask=mask, a_vae=out["a_vae"], a_imputed=out["a_imputed"],
                  a_filtered=out["a_filtered"], state_probs=out["state_probs"],
                  eps_a=tape.randn_like[0], K=K, T=T)
    if tape.gumbel:
        arrays["gumbel"] = torch.stack(tape.gumbel, 1)
    if full_images:
        for k in ("x_recon", "x_imputed", "x_filtered"):
            arrays[k] = out[k]
    else:  # keep the fixture small: every 10th frame
        for k in ("x_recon", "x_imputed", "x_filtered"):
            arrays[k] = out[k][:, ::10]
    save(name, **arrays)


# --------------------------------------------------------------------------------------
# 5. one end-to-end training step (kvae/train/train.py:32-58) with default init
# --------------------------------------------------------------------------------------
def bouncing_ball(B, T, seed):
    """Binary 32x32 frames of one ball; same generator as bench.py."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(6, 25, size=(B, 2))
    ang = rng.uniform(0, 2 * np.pi, size=B)
    spd = rng.uniform(0.5, 2.0, size=B)
    vel = np.stack([np.cos(ang), np.sin(ang)], 1) * spd[:, None]
    yy, xx = np.mgrid[0:32, 0:32]
    frames = np.zeros((B, T, 1, 32, 32), np.uint8)
    for t in range(T):
        d2 = (xx[None] - pos[:, 0, None, None]) ** 2 + (yy[None] - pos[:, 1, None, None]) ** 2
        frames[:, t, 0] = d2 <= 9.0
        pos = pos + vel
        for d in range(2):
            lo, hi = pos[:, d] < 3, pos[:, d] > 28
            pos[lo, d] = 6 - pos[lo, d]
            pos[hi, d] = 56 - pos[hi, d]
            vel[lo | hi, d] *= -1
    return frames


def train_step_case(name, dynamics, K, B=4, T=20, seed=0):
    torch.manual_seed(seed)
    cfg = KVAEConfig(dynamics_model=dynamics, num_modes=K)
    model = KVAE(cfg)
    with torch.no_grad():  # make the modes distinct so alpha grads are exercised
        model.kalman_filter.dyn_params.A.add_(0.05 * torch.randn_like(model.kalman_filter.dyn_params.A))
        if hasattr(model.kalman_filter.dyn_params, "head_w"):
            model.kalman_filter.dyn_params.head_w.bias.zero_()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    model.train()
    model.beta = 0.4
    frames = bouncing_ball(B, T, 1234)
    x = torch.from_numpy(frames).float()
    mask = torch.ones(B, T)
    opt = torch.optim.Adam(model.parameters(), lr=7e-3, weight_decay=0.0)
    model.kalman_filter.dyn_params.reset_state()
    torch.manual_seed(99)
    with record_noise() as tape:
        opt.zero_grad(set_to_none=True)
        outputs = model(x, mask=mask)
        losses = model.compute_loss(x, outputs, kf_weight=1.0, vae_weight=1.0, mask=mask)
        losses["loss"].backward()
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    opt.step()
    arrays = dict(frames=frames, beta=model.beta, lr=7e-3, clip=10.0,
                  eps_a=tape.randn_like[0], eps_z=tape.std_normal[0],
                  loss=losses["loss"], elbo_kf=losses["elbo_kf"], elbo_vae=losses["elbo_vae_total"],
                  recon=losses["recon"], kl=losses["kl"], grad_norm=gnorm,
                  a_samples=outputs["a_samples"], mus_smooth=outputs["mus_smooth"],
                  Sigmas_smooth=outputs["Sigmas_smooth"], state_probs=outputs["state_probs"])
    if tape.gumbel:
        arrays["gumbel"] = torch.stack(tape.gumbel, 1)
    for k, v in sd0.items():
        arrays["sd." + k] = v
    small = [k for k, g in grads.items() if g.numel() <= 4096]
    for k in small:
        arrays["grad." + k] = grads[k]
    for k, g in grads.items():
        arrays["gradnorm." + k] = g.norm()
    sd1 = model.state_dict()
    for k in small:
        arrays["after." + k] = sd1[k]
    for k in grads:
        arrays["afternorm." + k] = (sd1[k] - sd0[k]).norm()
    save(name, **arrays)


def main():
    print("generating goldens from", REF, "torch", torch.__version__)
    rocket(1, 11)
    rocket(4, 12)
    latent_case("latent_lstm_K1_B2_T10", "lstm", 1, 2, 10, seed=1)
    latent_case("latent_lstm_K3_B2_T10", "lstm", 3, 2, 10, seed=2)
    latent_case("latent_lstm_K3_B8_T20", "lstm", 3, 8, 20, seed=3)
    latent_case("latent_lstm_K3_B4_T50", "lstm", 3, 4, 50, seed=4)
    latent_case("latent_lstm_K3_B3_T12_u", "lstm", 3, 3, 12, seed=5)
    latent_case("latent_switch_K3_B2_T10", "switching", 3, 2, 10, seed=6)
    latent_case("latent_switch_K3_B8_T20", "switching", 3, 8, 20, seed=7)
    latent_case("latent_switch_K3_B4_T50", "switching", 3, 4, 50, seed=8, tau=0.5)
    # (switching, K=1) cannot be built through KVAE: StickyRegimePrior(1) raises ZeroDivisionError
    latent_case("latent_switch_K7_B2_T100", "switching", 7, 2, 100, seed=10, tau=0.7)
    # masked, eval mode (imputation.py:4-12 block mask)
    latent_case("masked_lstm_K3_B4_T20", "lstm", 3, 4, 20, seed=11, mask_block=(4, 10),
                train=False, with_grads=False)
    latent_case("masked_switch_K3_B4_T20", "switching", 3, 4, 20, seed=12, mask_block=(4, 10),
                train=False, with_grads=False)
    latent_case("masked_lstm_K3_B2_T16_grad", "lstm", 3, 2, 16, seed=13, mask_block=(4, 10),
                train=True, with_grads=True)
    latent_case("masked_switch_K3_B2_T16_grad", "switching", 3, 2, 16, seed=14, mask_block=(4, 10),
                train=True, with_grads=True)
    # n=16 stress numerics
    latent_case("stress_switch_z16_B2_T200", "switching", 3, 2, 200, z_dim=16, seed=15,
                with_grads=False, slim=True)
    latent_case("stress_lstm_z16_B2_T40_grad", "lstm", 3, 2, 40, z_dim=16, seed=16)
    stability_case("stability_lstm", "lstm")
    stability_case("stability_switching", "switching")
    stability_case("stability_switching_K7_T100", "switching", K=7, T=100, full_images=False)
    stability_case("stability_lstm_K7_T100", "lstm", K=7, T=100, full_images=False)
    train_step_case("trainstep_lstm_K3", "lstm", 3)
    train_step_case("trainstep_switch_K3", "switching", 3)


if __name__ == "__main__":
    main()
