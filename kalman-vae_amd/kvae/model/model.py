"""KVAE — drop-in for kvae.model.model.KVAE of the reference (model.py:11-301 there).

Same constructor (`KVAE(config)`), sub-module names (encoder, decoder, kalman_filter[.dyn_params]),
parameter registration order and state_dict keys, same `forward` / `compute_loss` / `impute`
signatures and output dictionaries.  For the reference's default shapes every layer of the frame VAE runs on the
hand-written HIP kernels of csrc/vae_*.h (kvae/vae/fused.py; other shapes: MIOpen + fused epilogues), and everything
between `a_samples` and the LGSSM ELBO runs in the HIP kernels behind `self.kalman_filter`.
"""
import os

import torch
from torch import nn

from kvae import _native, noise
from kvae.kalman import dyn_param as base_dyn_param
from kvae.kalman import switch_dyn_param
from kvae.kalman.kalman_filter import KalmanFilter
from kvae.vae.losses import LinearScheduler, count_active_units, vae_loss
from kvae.vae.vae import Decoder, Encoder


_FAST_PATH_NOTES = {
    "lgssm_specialised": "(z,u,a) is neither (4,4,2) nor (16,16,2): the LGSSM runs the run-time-dimension kernels",
    "lstm_registers": "alpha-net shape is not (hidden 50, a_dim 2): the LSTM runs the run-time-shape kernel",
    "bigru_registers": "regime posterior shape is not (hidden 50, a_dim 2): the bi-GRU runs on nn.GRU (MIOpen), "
                       "which cannot be captured into a hipGraph",
    "vae_default_shapes": "frame VAE is not the reference's default (32x32x1, channels [32,32,32], a_dim 2): "
                          "convolutions run on MIOpen with fused epilogues",
}


class _SideGradJoin(torch.autograd.Function):
    """Identity on a_samples for the frame branch; its backward adds the gradient the LGSSM branch has already produced on the
    side stream (`early_kf_backward`), after making the current stream wait for that stream."""

    @staticmethod
    def forward(ctx, a, holder):
        ctx.holder = holder
        return a.view_as(a)

    @staticmethod
    def backward(ctx, g):
        h = ctx.holder
        torch.cuda.current_stream().wait_stream(h["side"])
        ga = h["a_side"].grad
        if ga is None:   # the frame branch is being differentiated but nobody ran the LGSSM branch's backward
            raise RuntimeError("KVAE.early_kf_backward is on but the LGSSM branch has no gradient yet: call "
                               "compute_loss() (which runs that branch's backward) before loss.backward(), or clear the flag")
        return g + ga, None


# A/B switch for the scalar head of the objective, read ONCE (None: chosen per call from the step's schedule):
# 2 = frame terms through the fused head, 1 = LGSSM term too, 0 = torch's element-wise ops.
_LOSS_HEAD = os.environ.get("KVAE_LOSS_HEAD")


class KVAE(nn.Module):
    # Training-step schedule (addition over the reference; set by kvae.train.Trainer): the gradient of the LGSSM term w.r.t. the
    # encodings and the dynamics parameters does not depend on the frame terms, so compute_loss() can run that branch's backward
    # on the side stream right behind its forward; loss.backward() then only walks the frame branch and picks the result up.
    # The Trainer sets these three for the duration of its own forward+backward only (Trainer._schedule) and restores them:
    # a training-mode forward / compute_loss / backward outside a Trainer always takes the plain single-backward route.
    early_kf_backward = False
    lgssm_stream = None     # second HIP stream for the LGSSM chain: it is latency-bound (256 of ~8000 wave slots at configs[1])
    #                         and independent of the decoder, so it overlaps with the decoder convolutions; compute_loss() joins
    kf_value_only = False   # "vae" phase (reference train.py:246-250: kf_weight = 0, every LGSSM parameter frozen): the chain runs
    #                         forward only, without a tape, for the logged elbo_kf; nothing of it is differentiated

    def __init__(self, config):
        super().__init__()
        if hasattr(config, "validate"):
            config.validate()   # dimension limits of the HIP kernels (n, m, p, K <= 16) fail here, not at the first launch
            slow = [k for k, v in config.fast_path().items() if not v and k in _FAST_PATH_NOTES]
            if slow:   # shapes outside the hand-specialised kernels still run, on generic kernels / MIOpen: say so once
                import warnings
                warnings.warn("KVAE: " + "; ".join(_FAST_PATH_NOTES[k] for k in slow), stacklevel=2)
        self.config = config
        self.encoder = Encoder(config)
        self.decoder = Decoder(config)
        self.scheduler = LinearScheduler(config)
        self.beta = self.scheduler.get_beta(0) if config.scheduled_beta else 1.0
        self.K, self.z_dim, self.a_dim, self.u_dim = config.num_modes, config.z_dim, config.a_dim, config.u_dim

        # LGSSM initialisation: A_k = I, B_k, C_k ~ N(0, init_kf_matrices^2)  (reference model.py:33-45)
        A0 = torch.eye(self.z_dim).repeat(self.K, 1, 1)
        B0 = torch.randn(self.K, self.z_dim, self.u_dim) * config.init_kf_matrices
        C0 = torch.randn(self.K, self.a_dim, self.z_dim) * config.init_kf_matrices
        kind = config.dynamics_model.lower()
        if kind == "switching":
            prior = switch_dyn_param.StickyRegimePrior(self.K, p_stay=config.sticky_p_stay)
            posterior = switch_dyn_param.MarkovVariationalRegimePosterior(
                self.K, input_dim=self.a_dim, hidden_size=config.dynamics_hidden_dim)
            Q0 = torch.eye(self.z_dim).repeat(self.K, 1, 1) * config.noise_transition
            dynamics = switch_dyn_param.SwitchingDynamicsParameter(
                A0, B0, C0, Q=Q0, prior=prior, hidden_lstm=config.dynamics_hidden_dim,
                markov_regime_posterior=posterior)
            dynamics.tau = config.tau_init
        elif kind == "lstm":
            dynamics = base_dyn_param.DynamicsParameter(A0, B0, C0, hidden_lstm=config.dynamics_hidden_dim)
        else:
            raise ValueError(f"Unknown dynamics model: {config.dynamics_model}")
        # config noise values are variances
        self.kalman_filter = KalmanFilter(config.noise_transition ** 0.5, config.noise_emission ** 0.5,
                                          torch.zeros(self.z_dim), torch.eye(self.z_dim) * config.init_cov, dynamics)

    # -- VAE halves -----------------------------------------------------------------------------
    def reparameterize(self, mu, var):
        std = torch.sqrt(var + 1e-6)
        eps = noise.take("eps_a")
        eps = torch.randn_like(std) if eps is None else eps.to(device=std.device, dtype=std.dtype).reshape(std.shape)
        return mu + eps * std

    def encode_sequence(self, x):
        lead = x.shape[:2]
        feat = self.encoder.features(x.flatten(0, 1))
        eps = noise.take("eps_a")
        if eps is None:
            eps = torch.randn(feat.shape[0], self.config.a_dim, device=feat.device, dtype=feat.dtype)
        else:
            eps = eps.to(device=feat.device, dtype=feat.dtype).reshape(feat.shape[0], self.config.a_dim)
        a, mu, var = self.encoder.heads(feat, eps)   # both heads + reparameterisation: one kernel on the GPU path
        return a.unflatten(0, lead), mu.unflatten(0, lead), var.unflatten(0, lead)

    def decode_sequence(self, a):
        return self.decoder(a.flatten(0, 1)).unflatten(0, a.shape[:2])

    def _to_pixels(self, logits):
        return torch.sigmoid(logits) if self.config.out_distr.lower() == "bernoulli" else logits

    # -- full pass ------------------------------------------------------------------------------
    def forward(self, x, u=None, mask=None, with_recon=True):
        """`with_recon=False` (addition over the reference) skips sigmoid(x_logits): the training loss only
        needs the logits, so the step saves one full pass over the frame tensor."""
        a_samples, a_mu, a_var = self.encode_sequence(x)
        if u is None:
            u = torch.zeros(x.shape[0], x.shape[1], self.u_dim, device=x.device, dtype=x.dtype)
        self.kalman_filter.dyn_params.reset_state()
        side = self.lgssm_stream if (self.training and a_samples.is_cuda) else None
        a_side = None
        value_only = self.kf_value_only and self.training
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            a_lgssm = a_samples.detach() if value_only else a_samples
            if self.early_kf_backward and not value_only and torch.is_grad_enabled() and a_samples.requires_grad:
                a_side = a_lgssm = a_samples.detach().requires_grad_(True)
                a_samples = _SideGradJoin.apply(a_samples, {"a_side": a_side, "side": side})
            with torch.cuda.stream(side), torch.set_grad_enabled(torch.is_grad_enabled() and not value_only):
                smoothed = self.kalman_filter.smooth(a_lgssm, u, mask=mask)
        elif value_only:
            with torch.no_grad():
                smoothed = self.kalman_filter.smooth(a_samples.detach(), u, mask=mask)
        else:
            smoothed = self.kalman_filter.smooth(a_samples, u, mask=mask)
        (mus_smooth, Sigmas_smooth, mus_filt, Sigmas_filt, mus_pred, Sigmas_pred, A_list, B_list, C_list) = smoothed
        x_logits = self.decode_sequence(a_samples)
        return {
            "x_recon": self._to_pixels(x_logits) if with_recon else None, "x_logits": x_logits,
            "a_samples": a_samples, "a_mu": a_mu, "a_var": a_var, "a_side": a_side,
            "mus_smooth": mus_smooth, "Sigmas_smooth": Sigmas_smooth,
            "mus_filt": mus_filt, "Sigmas_filt": Sigmas_filt,
            "mus_pred": mus_pred, "Sigmas_pred": Sigmas_pred,
            "ABC": (A_list, B_list, C_list), "u": u,
            "state_probs": self.kalman_filter.dyn_params.state_seq,
        }

    def compute_loss(self, x, outputs, kf_weight=1.0, vae_weight=1.0, mask=None, with_metrics=True, weights_dev=None):
        """`with_metrics` (addition over the reference): True = the reference's behaviour (active-unit count and the
        two latent variances as Python numbers: three host syncs); "device" = the same statistics as device tensors
        (`active_units`, `latent_variances`), no sync, capturable into a hipGraph; False = skip them.
        `weights_dev` (addition): fp32 device tensor (vae_weight, kf_weight) that replaces the two floats - the kernels read it
        at run time, so a step captured into a hipGraph follows the reference's phase weights (train.py:246-260)."""
        if weights_dev is not None:
            vae_weight, kf_weight = weights_dev[0], weights_dev[1]
        B, T = x.shape[:2]
        a, a_mu, a_var = outputs["a_samples"], outputs["a_mu"], outputs["a_var"]
        A_list, B_list, C_list = outputs["ABC"]
        u = outputs.get("u")
        if u is None:
            u = torch.zeros(B, T, self.u_dim, device=x.device, dtype=x.dtype)
        x_mu = outputs["x_logits"] if outputs.get("x_logits") is not None else outputs["x_recon"]
        side = self.lgssm_stream if (self.training and a.is_cuda) else None

        a_side = outputs.get("a_side")
        stats = None
        if with_metrics == "device":   # before the LGSSM term is awaited: the main stream has this to do while the side stream finishes
            variances = a_mu.detach().reshape(-1, a_mu.shape[-1]).var(dim=0)
            stats = ((variances > 1e-2).sum(), variances)

        def kf_elbo():
            if self.kf_value_only and self.training:   # no tape, nothing to differentiate: the value for the log
                with torch.no_grad():
                    if side is None:
                        return self.kalman_filter.elbo(outputs["mus_smooth"], outputs["Sigmas_smooth"], a.detach(), u, A_list,
                                                       B_list, C_list, mask=mask)
                    with torch.cuda.stream(side):
                        v = self.kalman_filter.elbo(outputs["mus_smooth"], outputs["Sigmas_smooth"], a.detach(), u, A_list,
                                                    B_list, C_list, mask=mask)
                    torch.cuda.current_stream().wait_stream(side)
                    return v
            if a_side is not None:   # early_kf_backward: value and gradients of the LGSSM term on the side stream, now
                with torch.cuda.stream(side):
                    v = self.kalman_filter.elbo(outputs["mus_smooth"], outputs["Sigmas_smooth"], a_side, u, A_list, B_list,
                                                C_list, mask=mask)
                    done = torch.cuda.Event()
                    done.record(side)
                    # d loss / d elbo_kf = -kf_weight
                    v.backward((-kf_weight).reshape(v.shape) if torch.is_tensor(kf_weight) else torch.full_like(v, -float(kf_weight)))
                torch.cuda.current_stream().wait_event(done)   # the value only; _SideGradJoin waits for the gradients
                return v.detach()
            if side is None:
                return self.kalman_filter.elbo(outputs["mus_smooth"], outputs["Sigmas_smooth"], a, u, A_list, B_list, C_list,
                                               mask=mask)
            with torch.cuda.stream(side):
                v = self.kalman_filter.elbo(outputs["mus_smooth"], outputs["Sigmas_smooth"], a, u, A_list, B_list, C_list,
                                            mask=mask)
            torch.cuda.current_stream().wait_stream(side)   # join before the two ELBOs are combined
            return v

        head = _LOSS_HEAD
        if head is None and (a_side is not None or (self.kf_value_only and self.training and side is not None)):
            head = "2"
        if ((head in ("1", "2") or (head is None and side is None)) and self.config.out_distr.lower() == "bernoulli"
                and _native.fused_ok(x_mu) and x_mu.dtype == torch.float32
                and x.dtype == torch.float32 and a.dtype == torch.float32 and not x.requires_grad):
            # The two per-frame terms are one kernel each and the whole scalar head of the objective (masking, sums,
            # normalisation, beta / scale / weights, sign) is ONE launch each way (csrc/vae_heads.h) instead of ~40 dependent
            # element-wise launches.  With the LGSSM chain on a side stream and ONE backward, the ~40 launches were what gave
            # that chain's backward its head start over the decoder's Winograd kernels (which leave no registers for a
            # second kernel on their CUs): the fused head made the replayed graph slower.  With
            # early_kf_backward the side chain no longer needs that head start and the frame terms take the fused head ("2").
            from kvae.vae.fused import BernoulliFrameLogLik, LatentReg, LossHead
            lpx = BernoulliFrameLogLik.apply(x_mu, x)
            regf = LatentReg.apply(a, a_mu, a_var)
            mk = None if mask is None else mask.to(device=x.device, dtype=torch.float32).reshape(B, T)
            if head == "2":
                # frame terms only through the fused head (main stream, before the join); the LGSSM term joins with torch ops, so
                # its gradient reaches the side stream without waiting for the head's backward
                z = getattr(self, "_zero_kf", None)
                if z is None or z.device != x.device:
                    z = self._zero_kf = torch.zeros(1, device=x.device, dtype=torch.float32)
                wd = weights_dev   # its kf_weight multiplies the zero standing in for the LGSSM term here
                vae_loss_w, _, _, vae_elbo, recon, reg = LossHead.apply(
                    lpx, regf, z, mk, self.beta, self.config.scale_reconstruction, 0.0 if wd is not None else vae_weight, 0.0, wd)
                elbo_kf = kf_elbo()
                loss = vae_loss_w - kf_weight * elbo_kf
                out = {"loss": loss, "elbo_total": -loss.detach(), "elbo_kf": elbo_kf, "elbo_vae_total": vae_elbo,
                       "recon": recon, "kl": reg}
            else:
                elbo_kf = kf_elbo()
                loss, elbo_total, elbo_kf_v, vae_elbo, recon, reg = LossHead.apply(
                    lpx, regf, elbo_kf, mk, self.beta, self.config.scale_reconstruction,
                    0.0 if weights_dev is not None else vae_weight, 0.0 if weights_dev is not None else kf_weight, weights_dev)
                out = {"loss": loss, "elbo_total": elbo_total, "elbo_kf": elbo_kf_v, "elbo_vae_total": vae_elbo,
                       "recon": recon, "kl": reg}
        else:
            x_var = torch.tensor(self.config.noise_pixel_var, device=x.device, dtype=x_mu.dtype) \
                if self.config.out_distr.lower() != "bernoulli" else None
            vae_elbo, recon, reg = vae_loss(x, x_mu, x_var, a, a_mu, a_var,
                                            scale_reconstruction=self.config.scale_reconstruction, mask=mask,
                                            out_distr=self.config.out_distr, beta=self.beta)
            elbo_kf = kf_elbo()
            elbo_total = vae_weight * vae_elbo + kf_weight * elbo_kf
            out = {"loss": -elbo_total, "elbo_total": elbo_total, "elbo_kf": elbo_kf, "elbo_vae_total": vae_elbo,
                   "recon": recon, "kl": reg}
        if with_metrics == "device":
            out.update(active_units=stats[0], latent_variances=stats[1])
        elif with_metrics:
            active, variances = count_active_units(a_mu)
            out.update(active_units=active, latent_var_0=variances[0].item(), latent_var_1=variances[1].item())
        return out

    @torch.no_grad()
    def impute(self, x, mask, u=None):
        """Eval-mode imputation: decode C_t mu_{t|T} (smoothed) and C_t mu_{t|t} (filtered)."""
        self.eval()
        mask = mask.to(device=x.device, dtype=x.dtype)
        out = self.forward(x, u=u, mask=mask)
        _, _, C_list = out["ABC"]
        a_imputed, a_filtered = self.kalman_filter.emission_means(out["mus_smooth"], out["mus_filt"], C_list)
        # The reference decodes a_vae again (forward already did: the same pixels) and then the two read-outs one after the other
        # (model.py:275-290 there).  Frames are independent: forward's reconstruction is returned as it is, and ONE decoder pass
        # over the concatenated read-outs gives the other two - two decoder passes in all instead of four.
        x2 = self._to_pixels(self.decode_sequence(torch.cat([a_imputed, a_filtered], 0)))
        x_imputed, x_filtered = x2.chunk(2, 0)
        return {"x_recon": out["x_recon"], "x_imputed": x_imputed, "x_filtered": x_filtered,
                "a_vae": out["a_samples"], "a_imputed": a_imputed, "a_filtered": a_filtered,
                "state_probs": out["state_probs"]}
