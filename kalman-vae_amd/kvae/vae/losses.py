"""VAE part of the KVAE objective (reference kvae/vae/losses.py:6-149).  Element-wise work and
reductions on PyTorch-ROCm; same function names, argument order and return triples."""
import math

import torch
import torch.nn.functional as F

from kvae.utils.config import KVAEConfig

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def log_gaussian(x, mean, var):
    """Element-wise log N(x; mean, var)."""
    return -_HALF_LOG_2PI - 0.5 * torch.log(var) - (x - mean) ** 2 / (2 * var)


def _frame_mask(mask, x):
    B, T = x.shape[:2]
    if mask is None:
        return torch.ones(B, T, device=x.device, dtype=x.dtype)
    return mask.to(device=x.device, dtype=x.dtype).reshape(B, T)


def log_likelihood(x, x_mu, x_var, a, a_mu, a_var, mask=None):
    """(sum_bt mask*log p(x|a), sum_bt mask*log q(a|x)) with a Gaussian pixel model."""
    mk = _frame_mask(mask, x)
    lpx = log_gaussian(x, x_mu, x_var).sum(dim=(2, 3, 4))
    lqa = log_gaussian(a, a_mu, a_var).sum(dim=-1)
    return (lpx * mk).sum(), (lqa * mk).sum()


def vae_loss(x, x_mu, x_var, a, a_mu, a_var, scale_reconstruction: float = 0.3, beta: float = 1.0,
             mask=None, out_distr: str = "gaussian"):
    """Returns (vae_elbo, recon_term, regularization_term), each divided by the observed-frame count.
    vae_elbo = scale_reconstruction * E[log p(x|a)] + beta * (log p(a) - log q(a|x))."""
    mk = _frame_mask(mask, x)
    denom = mk.sum().clamp(min=1.0)
    from kvae import _native
    if out_distr.lower() == "bernoulli" and _native.fused_ok(a) and a.dtype == torch.float32 and a_var.dim() == a.dim():
        # GPU training path: Bernoulli frame term and the latent regulariser as one fused op each (csrc/vae_loss.h,
        # csrc/vae_heads.h); same quantities as the generic code below
        from kvae.vae.fused import BernoulliFrameLogLik, LatentReg
        if x_mu.dtype == torch.float32 and x.dtype == torch.float32 and not x.requires_grad:
            lpx = BernoulliFrameLogLik.apply(x_mu, x)
        else:
            lpx = -F.binary_cross_entropy_with_logits(x_mu, x, reduction="none").sum(dim=(2, 3, 4))
        recon = (lpx * mk).sum() / denom
        reg = (LatentReg.apply(a, a_mu, a_var) * mk).sum() / denom
        return scale_reconstruction * recon + beta * reg, recon, reg
    if out_distr.lower() == "bernoulli":
        if _native.fused_ok(x_mu) and x_mu.dtype == torch.float32 and x.dtype == torch.float32 and not x.requires_grad:
            from kvae.vae.fused import BernoulliFrameLogLik
            lpx = BernoulliFrameLogLik.apply(x_mu, x)    # one HIP pass per direction (csrc/vae_loss.h)
        else:
            lpx = -F.binary_cross_entropy_with_logits(x_mu, x, reduction="none").sum(dim=(2, 3, 4))
        log_px = (lpx * mk).sum()
        log_qa = (log_gaussian(a, a_mu, a_var).sum(-1) * mk).sum()
    else:
        log_px, log_qa = log_likelihood(x, x_mu, x_var, a, a_mu, a_var, mask=mask)
    log_pa = ((-_HALF_LOG_2PI - 0.5 * a ** 2).sum(-1) * mk).sum()   # log N(a; 0, 1)
    recon = log_px / denom
    reg = (log_pa - log_qa) / denom
    return scale_reconstruction * recon + beta * reg, recon, reg


class LinearScheduler:
    """beta ramps linearly from start_val (epoch <= start_epoch) to end_val (epoch >= end_epoch)."""

    def __init__(self, config: KVAEConfig):
        self.start_epoch, self.end_epoch = config.start_epoch, config.end_epoch
        self.start_val, self.end_val = config.start_val, config.end_val

    def get_beta(self, current_epoch: int) -> float:
        if current_epoch < self.start_epoch:
            return self.start_val
        if current_epoch >= self.end_epoch:
            return self.end_val
        frac = (current_epoch - self.start_epoch) / (self.end_epoch - self.start_epoch)
        return self.start_val + frac * (self.end_val - self.start_val)


def count_active_units(mu_tensor, threshold=1e-2):
    """(#latent dims whose mean varies across the batch by more than threshold, the variances)."""
    mu = mu_tensor.reshape(-1, mu_tensor.shape[-1]) if mu_tensor.dim() == 3 else mu_tensor
    variances = mu.var(dim=0)
    return int((variances > threshold).sum().item()), variances
