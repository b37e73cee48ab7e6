"""CPU oracle for the Kalman-VAE hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional (state_dict-in, tensors-out) restatement in plain PyTorch-CPU of the algorithm the
reference implements in kvae/kalman/kalman_filter.py, kvae/kalman/dyn_param.py,
kvae/kalman/switch_dyn_param.py, kvae/model/model.py, kvae/vae/{vae,losses}.py and the step body
of kvae/train/train.py:44-58.  Every function cites the reference lines it follows.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (kalman-vae_amd/kvae) never does: it fails loudly if the HIP library is missing.

Pinned against the golden vectors in tests/golden/*.npz, which were captured by importing the
reference itself (tests/golden/make_goldens.py); see tests/test_oracle_golden.py.

All random draws are *injected* (eps_a, eps_z, gumbel) so results are reproducible across
devices; gradients come from autograd over this restatement, exactly as the reference gets
them from autograd over its own ops.
"""
import math

import torch
import torch.nn.functional as F

LOG2PI = math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------------------------
# LGSSM primitives
# ----------------------------------------------------------------------------------------------
def filter_step(mu, Sig, y, u, A, Bm, C, Q, R, mask_t):
    """One predict+update (kalman_filter.py:31-104). mu [B,n,1], Sig [B,n,n], y [B,p], u [B,m]."""
    eye = torch.eye(A.shape[-1], dtype=A.dtype)
    mu_p = A @ mu + Bm @ u.unsqueeze(-1)                                  # :65
    Sig_p = A @ Sig @ A.mT + Q                                            # :67
    r = y.unsqueeze(-1) - C @ mu_p                                        # :73-75
    S = C @ Sig_p @ C.mT + R                                              # :78
    S = 0.5 * (S + S.mT)                                                  # :79
    PCT = Sig_p @ C.mT                                                    # :82
    K = torch.linalg.solve(S, PCT.mT).mT                                  # :89
    K = mask_t.view(-1, 1, 1) * K                                         # :92
    mu_f = mu_p + K @ r                                                   # :96
    IKC = eye - K @ C                                                     # :99
    Sig_f = IKC @ Sig_p @ IKC.mT + K @ R @ K.mT                           # :100
    Sig_f = 0.5 * (Sig_f + Sig_f.mT)                                      # :101
    return mu_f, Sig_f, mu_p, Sig_p


def smooth_step(Sig_f, Sig_p_next, Sig_s_next, mu_f, mu_p_next, mu_s_next, A_next):
    """One RTS step (kalman_filter.py:204-237)."""
    J = torch.linalg.solve(Sig_p_next.mT, (Sig_f @ A_next.mT).mT).mT      # :229
    mu_s = mu_f + J @ (mu_s_next - mu_p_next)                             # :232
    Sig_s = Sig_f + J @ (Sig_s_next - Sig_p_next) @ J.mT                  # :234
    Sig_s = 0.5 * (Sig_s + Sig_s.mT)                                      # :235
    return mu_s, Sig_s


def safe_cholesky(Sigma, max_tries=5, jitter_init=1e-6):
    """kalman_filter.py:282-302: jitter is added on the first try already; whole-batch retry."""
    n = Sigma.size(-1)
    Sigma = 0.5 * (Sigma + Sigma.mT)
    eye = torch.eye(n, dtype=Sigma.dtype)
    jitter = jitter_init
    for _ in range(max_tries):
        L, info = torch.linalg.cholesky_ex(Sigma + jitter * eye)
        if not bool((info != 0).any()):
            return L
        jitter *= 10.0
    diag = torch.diagonal(Sigma, dim1=-2, dim2=-1).clamp(min=1e-6)
    return torch.diag_embed(torch.sqrt(diag))


def mvn_logprob_tril(x, L):
    """log N(x; 0, L L^T) as torch.distributions.MultivariateNormal.log_prob computes it."""
    w = torch.linalg.solve_triangular(L, x.unsqueeze(-1), upper=False).squeeze(-1)
    half_log_det = torch.diagonal(L, dim1=-2, dim2=-1).log().sum(-1)
    return -0.5 * (x.shape[-1] * LOG2PI + (w * w).sum(-1)) - half_log_det


# ----------------------------------------------------------------------------------------------
# dynamics parameter networks
# ----------------------------------------------------------------------------------------------
def lstm_cell(x, state, w_ih, w_hh, b_ih, b_hh):
    """nn.LSTM single step, gate order i,f,g,o (dyn_param.py:52)."""
    H = w_hh.shape[1]
    if state is None:
        h = x.new_zeros(x.shape[0], H)
        c = x.new_zeros(x.shape[0], H)
    else:
        h, c = state
    gates = x @ w_ih.T + b_ih + h @ w_hh.T + b_hh
    i, f, g, o = gates.split(H, dim=1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, (h, c)


def lstm_dyn_step(dyn, a_prev, state):
    """DynamicsParameter.compute_step (dyn_param.py:39-63). dyn: dict with A,B,C[,lstm.*,head_w.*]."""
    K = dyn["A"].shape[0]
    Bsz = a_prev.shape[0]
    if K == 1:
        w = torch.ones(Bsz, 1, dtype=a_prev.dtype)
        return (dyn["A"][0].expand(Bsz, -1, -1), dyn["B"][0].expand(Bsz, -1, -1),
                dyn["C"][0].expand(Bsz, -1, -1), w, state)
    h, state = lstm_cell(a_prev, state, dyn["lstm.weight_ih_l0"], dyn["lstm.weight_hh_l0"],
                         dyn["lstm.bias_ih_l0"], dyn["lstm.bias_hh_l0"])
    w = torch.softmax(h @ dyn["head_w.weight"].T + dyn["head_w.bias"], dim=-1)
    A = torch.einsum("bk,kij->bij", w, dyn["A"])
    Bm = torch.einsum("bk,knm->bnm", w, dyn["B"])
    C = torch.einsum("bk,kpn->bpn", w, dyn["C"])
    return A, Bm, C, w, state


def gru_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of nn.GRU (gate order r,z,n)."""
    Bsz, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(Bsz, H)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gi = x[:, t] @ w_ih.T + b_ih
        gh = h @ w_hh.T + b_hh
        i_r, i_z, i_n = gi.split(H, 1)
        h_r, h_z, h_n = gh.split(H, 1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, 1)


def gumbel_softmax_injected(logits, g, tau, hard):
    """F.gumbel_softmax with the Gumbel noise g supplied (straight-through when hard)."""
    y_soft = ((logits + g) / tau).softmax(-1)
    if not hard:
        return y_soft
    idx = y_soft.max(-1, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(-1, idx, 1.0)
    return y_hard - y_soft.detach() + y_soft


def sticky_transition_matrix(K, p_stay):
    """StickyRegimePrior (switch_dyn_param.py:98-103)."""
    P = torch.ones(K, K) * ((1 - p_stay) / (K - 1))
    P.fill_diagonal_(p_stay)
    return P


def switching_compute_batch(dyn, a_seq, tau, is_training, gumbel, trans_matrix):
    """SwitchingDynamicsParameter.compute_batch (switch_dyn_param.py:37-92) with injected Gumbel
    noise gumbel[B,T,K]. Returns dict(A_seq,B_seq,C_seq,Q_seq,log_qseq,log_pseq,state_seq)."""
    Bsz, T, _ = a_seq.shape
    K = dyn["A"].shape[0]
    if K == 1:                                                                       # :40-49
        z = a_seq.new_zeros(Bsz, T)
        return dict(A_seq=dyn["A"][0].expand(Bsz, T, -1, -1), B_seq=dyn["B"][0].expand(Bsz, T, -1, -1),
                    C_seq=dyn["C"][0].expand(Bsz, T, -1, -1), Q_seq=dyn["Q"][0].expand(Bsz, T, -1, -1),
                    log_qseq=z, log_pseq=z, state_seq=a_seq.new_ones(Bsz, T, 1))
    pre = "markov_regime_posterior."
    hf = gru_direction(a_seq, dyn[pre + "bigru.weight_ih_l0"], dyn[pre + "bigru.weight_hh_l0"],
                       dyn[pre + "bigru.bias_ih_l0"], dyn[pre + "bigru.bias_hh_l0"], False)
    hb = gru_direction(a_seq, dyn[pre + "bigru.weight_ih_l0_reverse"], dyn[pre + "bigru.weight_hh_l0_reverse"],
                       dyn[pre + "bigru.bias_ih_l0_reverse"], dyn[pre + "bigru.bias_hh_l0_reverse"], True)
    h_seq = torch.cat([hf, hb], -1)                                                  # :123
    logits = (h_seq @ dyn[pre + "linear_head.weight"].T + dyn[pre + "linear_head.bias"]).view(Bsz, T, K, K)
    init_logits = h_seq[:, 0] @ dyn[pre + "init_head.weight"].T + dyn[pre + "init_head.bias"]   # :128
    hard = not is_training
    y0 = gumbel_softmax_injected(init_logits, gumbel[:, 0], tau, hard)               # :52
    log_q0 = torch.log_softmax(init_logits, -1)
    log_p0 = torch.full_like(log_q0, 1.0 / K).log()
    ys, lq, lp = [y0], [(y0 * log_q0).sum(-1)], [(y0 * log_p0).sum(-1)]
    y_prev = y0
    for t in range(1, T):                                                            # :67-79
        l_t = torch.matmul(y_prev.unsqueeze(1), logits[:, t]).squeeze(1)
        y_t = gumbel_softmax_injected(l_t, gumbel[:, t], tau, hard)
        lq.append((y_t * torch.log_softmax(l_t, -1)).sum(-1))
        tp = torch.matmul(y_prev.unsqueeze(1), trans_matrix).squeeze(1)
        lp.append((y_t * torch.log(tp.clamp_min(1e-8))).sum(-1))
        ys.append(y_t)
        y_prev = y_t
    y_seq = torch.stack(ys, 1)
    return dict(A_seq=torch.einsum("btk,kij->btij", y_seq, dyn["A"]),               # :82-86
                B_seq=torch.einsum("btk,knm->btnm", y_seq, dyn["B"]),
                Q_seq=torch.einsum("btk,kij->btij", y_seq, dyn["Q"]),
                C_seq=dyn["C"][0].expand(Bsz, T, -1, -1),
                log_qseq=torch.stack(lq, 1), log_pseq=torch.stack(lp, 1), state_seq=y_seq)


# ----------------------------------------------------------------------------------------------
# filter / smooth / elbo over a batch of sequences
# ----------------------------------------------------------------------------------------------
def lgssm_filter(Y, U, mask, dyn, kind, Qbuf, R, mu0, Sigma0, *, tau=1.0, is_training=True,
                 gumbel=None, trans_matrix=None):
    """KalmanFilter.filter (kalman_filter.py:107-201). kind in {"lstm","switching"}.
    Returns dict with mus_filt,Sigmas_filt,mus_pred,Sigmas_pred,A_list,B_list,C_list,state_seq
    (+ Q_seq,log_qseq,log_pseq for switching)."""
    Bsz, T, p = Y.shape
    mu = mu0.expand(Bsz, -1).unsqueeze(-1)
    Sig = Sigma0.expand(Bsz, -1, -1)
    if mask is None:
        mask = torch.ones(Bsz, T, dtype=Y.dtype)
    extra = {}
    if kind == "switching":
        sw = switching_compute_batch(dyn, Y, tau, is_training, gumbel, trans_matrix)   # :135-139
        extra = {k: sw[k] for k in ("Q_seq", "log_qseq", "log_pseq")}
    else:
        y_for_dyn = Y.new_zeros(Bsz, p)                                               # :142
        state, ws = None, []
    rec = {k: [] for k in ("mus_filt", "Sigmas_filt", "mus_pred", "Sigmas_pred", "A_list", "B_list", "C_list")}
    for t in range(T):
        if kind == "switching":
            A, Bm, C, Q_t = sw["A_seq"][:, t], sw["B_seq"][:, t], sw["C_seq"][:, t], sw["Q_seq"][:, t]
        else:
            A, Bm, C, w, state = lstm_dyn_step(dyn, y_for_dyn, state)                 # :159
            ws.append(w)
            Q_t = Qbuf
        m_t = mask[:, t]
        mu, Sig, mu_p, Sig_p = filter_step(mu, Sig, Y[:, t], U[:, t], A, Bm, C, Q_t, R, m_t)
        for k, v in zip(rec, (mu, Sig, mu_p, Sig_p, A, Bm, C)):
            rec[k].append(v)
        if kind != "switching":
            y_pred = (C @ mu_p).squeeze(-1)                                           # :183-185
            y_for_dyn = m_t.view(-1, 1) * Y[:, t] + (1.0 - m_t.view(-1, 1)) * y_pred
    out = {k: torch.stack(v, 1) for k, v in rec.items()}
    out["state_seq"] = sw["state_seq"] if kind == "switching" else torch.stack(ws, 1)
    out.update(extra)
    return out


def lgssm_smooth(Y, U, mask, dyn, kind, Qbuf, R, mu0, Sigma0, **kw):
    """KalmanFilter.smooth (kalman_filter.py:240-279)."""
    out = lgssm_filter(Y, U, mask, dyn, kind, Qbuf, R, mu0, Sigma0, **kw)
    T = Y.shape[1]
    mu_s, Sig_s = out["mus_filt"][:, -1], out["Sigmas_filt"][:, -1]
    mus, Sigs = [mu_s], [Sig_s]
    for t in range(T - 2, -1, -1):
        mu_s, Sig_s = smooth_step(out["Sigmas_filt"][:, t], out["Sigmas_pred"][:, t + 1], Sig_s,
                                  out["mus_filt"][:, t], out["mus_pred"][:, t + 1], mu_s,
                                  out["A_list"][:, t + 1])
        mus.append(mu_s)
        Sigs.append(Sig_s)
    out["mus_smooth"] = torch.stack(mus[::-1], 1)
    out["Sigmas_smooth"] = torch.stack(Sigs[::-1], 1)
    return out


def lgssm_elbo_terms(mu_s, Sig_s, y, u, A_list, B_list, C_list, Q_list, R, mu0, Sigma0, mask, eps_z):
    """The four LGSSM terms of KalmanFilter.elbo (kalman_filter.py:347-389), each summed over B,T:
    (transition, emission, init, entropy). Q_list is [B,T,n,n] or [n,n]."""
    Bsz, T, p = y.shape
    n = A_list.shape[-1]
    if mu_s.dim() == 4:
        mu_s = mu_s.squeeze(-1)
    if Q_list.dim() == 2:
        Q_list = Q_list.expand(Bsz, T, -1, -1)                                        # :345
    L = safe_cholesky(Sig_s)                                                          # :348
    z = mu_s + (L @ eps_z.unsqueeze(-1)).squeeze(-1)                                  # :351 (rsample)
    z_prev = z[:, :-1].unsqueeze(-1)
    mu_trans = (A_list[:, 1:] @ z_prev + B_list[:, 1:] @ u[:, 1:].unsqueeze(-1)).squeeze(-1)   # :353-361
    L_Q = safe_cholesky(Q_list[:, 1:])                                                # :364
    lp_trans = mvn_logprob_tril(z[:, 1:] - mu_trans, L_Q)                             # :368
    mu_emiss = (C_list @ z.unsqueeze(-1)).squeeze(-1)                                 # :372
    L_R = torch.linalg.cholesky(R)
    lp_emiss = mvn_logprob_tril(y - mu_emiss, L_R) * mask                             # :374-377
    L_0 = torch.linalg.cholesky(Sigma0)
    lp_init = mvn_logprob_tril(z[:, 0] - mu0, L_0)                                    # :380-381
    entropy = -mvn_logprob_tril(z - mu_s, L)                                          # :389
    return lp_trans.sum(), lp_emiss.sum(), lp_init.sum(), entropy.sum()


def lgssm_elbo(mu_s, Sig_s, y, u, A_list, B_list, C_list, Q_list, R, mu0, Sigma0, mask, eps_z,
               log_qseq=None, log_pseq=None):
    """KalmanFilter.elbo (kalman_filter.py:305-401)."""
    if mask is None:
        mask = torch.ones(y.shape[0], y.shape[1], dtype=y.dtype)
    tr, em, ini, ent = lgssm_elbo_terms(mu_s, Sig_s, y, u, A_list, B_list, C_list, Q_list, R, mu0,
                                        Sigma0, mask, eps_z)
    lq = log_qseq.sum() if log_qseq is not None else 0.0
    lp = log_pseq.sum() if log_pseq is not None else 0.0
    return (tr + em + ini + lp - lq + ent) / mask.sum().clamp(min=1.0)                # :392-400


def split_dyn(sd, prefix="kalman_filter.dyn_params."):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def smooth_and_elbo(dyn, kind, Y, U, mask, Qbuf, R, mu0, Sigma0, eps_z, **kw):
    """smooth() followed by elbo() the way KVAE.forward/compute_loss chain them
    (model.py:153-161, 218-222)."""
    out = lgssm_smooth(Y, U, mask, dyn, kind, Qbuf, R, mu0, Sigma0, **kw)
    Q_list = out["Q_seq"] if kind == "switching" else Qbuf                            # :342-345
    out["elbo"] = lgssm_elbo(out["mus_smooth"], out["Sigmas_smooth"], Y, U, out["A_list"], out["B_list"],
                             out["C_list"], Q_list, R, mu0, Sigma0, mask, eps_z,
                             out.get("log_qseq"), out.get("log_pseq"))
    return out


# ----------------------------------------------------------------------------------------------
# conv VAE + full model step (functional over a state_dict)
# ----------------------------------------------------------------------------------------------
def encoder(sd, x, noise_emission):
    """Encoder.forward (vae.py:53-64): 3x[conv3x3 s2 p1, ReLU] -> fc_mu, sigmoid(fc_var)*noise."""
    h = x
    for i in (0, 2, 4):
        h = F.relu(F.conv2d(h, sd[f"encoder.conv_layers.{i}.weight"], sd[f"encoder.conv_layers.{i}.bias"],
                            stride=2, padding=1))
    h = h.reshape(h.shape[0], -1)
    mu = F.linear(h, sd["encoder.fc_mu.weight"], sd["encoder.fc_mu.bias"])
    var = torch.sigmoid(F.linear(h, sd["encoder.fc_var.0.weight"], sd["encoder.fc_var.0.bias"]))
    return mu, noise_emission * var


def decoder(sd, a):
    """Decoder.forward (vae.py:106-116): fc -> [32,4,4] -> 2x[conv, PixelShuffle2, ReLU] -> conv, PS."""
    h = F.linear(a, sd["decoder.fc.weight"], sd["decoder.fc.bias"])
    c0 = sd["decoder.deconv_layers.0.weight"].shape[1]
    side = int(round((h.shape[1] // c0) ** 0.5))
    h = h.view(-1, c0, side, side)
    h = F.relu(F.pixel_shuffle(F.conv2d(h, sd["decoder.deconv_layers.0.weight"], sd["decoder.deconv_layers.0.bias"], padding=1), 2))
    h = F.relu(F.pixel_shuffle(F.conv2d(h, sd["decoder.deconv_layers.3.weight"], sd["decoder.deconv_layers.3.bias"], padding=1), 2))
    return F.pixel_shuffle(F.conv2d(h, sd["decoder.deconv_layers.6.weight"], sd["decoder.deconv_layers.6.bias"], padding=1), 2)


def log_gaussian(x, mean, var):
    """losses.py:6-19."""
    return -0.5 * LOG2PI - torch.log(var) / 2 - torch.square(x - mean) / (2 * var)


def vae_elbo_bernoulli(x, x_logits, a, a_mu, a_var, mask, scale_reconstruction, beta):
    """vae_loss, out_distr='bernoulli' (losses.py:63-112)."""
    denom = mask.sum().clamp(min=1.0)
    log_px = -(F.binary_cross_entropy_with_logits(x_logits, x, reduction="none").sum(dim=(2, 3, 4)))
    log_px = (log_px * mask).sum()
    log_q = (log_gaussian(a, a_mu, a_var).sum(-1) * mask).sum()
    log_p = (log_gaussian(a, torch.zeros_like(a), torch.ones_like(a)).sum(-1) * mask).sum()
    recon = log_px / denom
    reg = (log_p - log_q) / denom
    return scale_reconstruction * recon + beta * reg, recon, reg


def kvae_forward(sd, x, mask, *, kind, eps_a, eps_z=None, gumbel=None, u=None, training=True, tau=1.0,
                 noise_emission=0.03, sticky_p_stay=0.8, beta=1.0, scale_reconstruction=0.3,
                 kf_weight=1.0, vae_weight=1.0, with_loss=True, with_metrics=False):
    """KVAE.forward + compute_loss (model.py:134-241), bernoulli output, injected noise.
    with_metrics: also the logging-only statistics of compute_loss (count_active_units, losses.py:137-149; model.py:229)."""
    Bsz, T = x.shape[:2]
    a_mu, a_var = encoder(sd, x.reshape(-1, *x.shape[2:]), noise_emission)
    a = a_mu + eps_a * torch.sqrt(a_var + 1e-6)                                        # model.py:81-84
    a, a_mu, a_var = (t.view(Bsz, T, -1) for t in (a, a_mu, a_var))
    dyn = split_dyn(sd)
    n, m = dyn["A"].shape[1], dyn["B"].shape[2]
    if u is None:
        u = x.new_zeros(Bsz, T, m)                                                     # :149-150
    if mask is None:
        mask = x.new_ones(Bsz, T)
    kf = {k: sd["kalman_filter." + k] for k in ("Q", "R", "mu0", "Sigma0")}
    K = dyn["A"].shape[0]
    trans = sticky_transition_matrix(K, sticky_p_stay) if (kind == "switching" and K > 1) else None
    out = lgssm_smooth(a, u, mask, dyn, kind, kf["Q"], kf["R"], kf["mu0"], kf["Sigma0"], tau=tau,
                       is_training=training, gumbel=gumbel, trans_matrix=trans)
    x_logits = decoder(sd, a.reshape(-1, a.shape[-1])).view(Bsz, T, *x.shape[2:])      # :164
    out.update(a_samples=a, a_mu=a_mu, a_var=a_var, x_logits=x_logits, x_recon=torch.sigmoid(x_logits), u=u)
    if with_loss:
        vae_elbo, recon, reg = vae_elbo_bernoulli(x, x_logits, a, a_mu, a_var, mask, scale_reconstruction, beta)
        Q_list = out["Q_seq"] if kind == "switching" else kf["Q"]
        elbo_kf = lgssm_elbo(out["mus_smooth"], out["Sigmas_smooth"], a, u, out["A_list"], out["B_list"],
                             out["C_list"], Q_list, kf["R"], kf["mu0"], kf["Sigma0"], mask, eps_z,
                             out.get("log_qseq"), out.get("log_pseq"))
        elbo_total = vae_weight * vae_elbo + kf_weight * elbo_kf                       # :225
        out.update(loss=-elbo_total, elbo_kf=elbo_kf, elbo_vae_total=vae_elbo, recon=recon, kl=reg)
        if with_metrics:
            variances = a_mu.detach().reshape(-1, a_mu.shape[-1]).var(dim=0)
            out.update(active_units=int((variances > 1e-2).sum().item()), latent_variances=variances)
    return out


def kvae_impute(sd, x, mask, *, kind, eps_a, gumbel=None, **kw):
    """KVAE.impute (model.py:243-301), eval mode."""
    out = kvae_forward(sd, x, mask, kind=kind, eps_a=eps_a, gumbel=gumbel, training=False, with_loss=False, **kw)
    Bsz, T = x.shape[:2]
    a_imp = (out["C_list"] @ out["mus_smooth"]).squeeze(-1)                            # :281
    a_fil = (out["C_list"] @ out["mus_filt"]).squeeze(-1)                              # :288

    def dec(a):
        return torch.sigmoid(decoder(sd, a.reshape(-1, a.shape[-1])).view(Bsz, T, *x.shape[2:]))

    return dict(x_recon=out["x_recon"], x_imputed=dec(a_imp), x_filtered=dec(a_fil), a_vae=out["a_samples"],
                a_imputed=a_imp, a_filtered=a_fil, state_probs=out["state_seq"])


class OracleTrainer:
    """The step body of train_one_epoch (train.py:44-58) over a state_dict of leaf tensors:
    zero_grad, forward, loss, backward, clip_grad_norm_(10), Adam.step.  Used as bench.py's
    cpu_baseline ("port") and as the end-to-end parity checker."""

    BUFFERS = ("kalman_filter.Q", "kalman_filter.R", "kalman_filter.I", "kalman_filter.mu0", "kalman_filter.Sigma0")

    def __init__(self, sd, kind, lr=7e-3, weight_decay=0.0, clip=10.0, **fw):
        self.kind, self.clip, self.fw = kind, clip, fw
        self.sd = {k: v.detach().clone().float() for k, v in sd.items()}
        self.params = [k for k in self.sd if k not in self.BUFFERS]
        for k in self.params:
            self.sd[k].requires_grad_(True)
        self.opt = torch.optim.Adam([self.sd[k] for k in self.params], lr=lr, weight_decay=weight_decay)

    def set_training_phase(self, phase):
        """set_training_phase (train.py:142-207): "vae" trains encoder + decoder only; "warmup" also A, B, C (and Q of the
        switching model); the alpha-network (lstm.*, head_w.*) / regime posterior stays frozen until "all"."""
        assert phase in ("vae", "warmup", "all")
        d = "kalman_filter.dyn_params."
        for k in self.params:
            vae = k.startswith("encoder.") or k.startswith("decoder.")
            mats = k in (d + "A", d + "B", d + "C", d + "Q")
            self.sd[k].requires_grad_(phase == "all" or vae or (phase == "warmup" and mats))

    def step(self, x, *, eps_a, eps_z, gumbel=None, mask=None, **fw):
        """fw: per-step overrides of the forward's keyword arguments (kf_weight of the current phase, train.py:246-260)."""
        self.opt.zero_grad(set_to_none=True)
        out = kvae_forward(self.sd, x, mask, kind=self.kind, eps_a=eps_a, eps_z=eps_z, gumbel=gumbel, **{**self.fw, **fw})
        out["loss"].backward()
        gn = torch.nn.utils.clip_grad_norm_([self.sd[k] for k in self.params], self.clip)
        self.opt.step()
        out["grad_norm"] = gn
        return out
