"""Helpers shared by the parity tests: load a golden .npz (tests/golden) as torch tensors."""
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


def load(name):
    with np.load(GOLDEN / f"{name}.npz") as z:
        return {k: torch.from_numpy(np.array(z[k])) for k in z.files}


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


LATENT_CASES = [
    ("latent_lstm_K1_B2_T10", "lstm"), ("latent_lstm_K3_B2_T10", "lstm"), ("latent_lstm_K3_B8_T20", "lstm"),
    ("latent_lstm_K3_B4_T50", "lstm"), ("latent_lstm_K3_B3_T12_u", "lstm"),
    ("latent_switch_K3_B2_T10", "switching"), ("latent_switch_K3_B8_T20", "switching"),
    ("latent_switch_K3_B4_T50", "switching"), ("latent_switch_K7_B2_T100", "switching"),
    ("masked_lstm_K3_B4_T20", "lstm"), ("masked_switch_K3_B4_T20", "switching"),
    ("masked_lstm_K3_B2_T16_grad", "lstm"), ("masked_switch_K3_B2_T16_grad", "switching"),
    ("masked_lstm_K7_B2_T100", "lstm"), ("masked_switch_K7_B2_T100", "switching"),   # configs[3], non-degenerate (round 3)
    ("stress_lstm_z16_B2_T40_grad", "lstm"), ("stress_switch_z16_B2_T200", "switching"),
]
SMOOTH_KEYS = ["mus_smooth", "Sigmas_smooth", "mus_filt", "Sigmas_filt", "mus_pred", "Sigmas_pred",
               "A_list", "B_list", "C_list"]


def reference_param_layout(kind, K, n=4, m=4, p=2, hidden=50):
    """(name, shape) of every nn.Parameter of the reference KVAE in .parameters() order
    (module construction order of kvae/model/model.py:17-78, vae.py:14-44,70-104,
    dyn_param.py:6-33, switch_dyn_param.py:8-30,114-120), KVAEConfig defaults."""
    out = []
    cin = 1
    for i in (0, 2, 4):
        out += [(f"encoder.conv_layers.{i}.weight", (32, cin, 3, 3)), (f"encoder.conv_layers.{i}.bias", (32,))]
        cin = 32
    out += [("encoder.fc_mu.weight", (p, 512)), ("encoder.fc_mu.bias", (p,)),
            ("encoder.fc_var.0.weight", (p, 512)), ("encoder.fc_var.0.bias", (p,)),
            ("decoder.fc.weight", (512, p)), ("decoder.fc.bias", (512,)),
            ("decoder.deconv_layers.0.weight", (128, 32, 3, 3)), ("decoder.deconv_layers.0.bias", (128,)),
            ("decoder.deconv_layers.3.weight", (128, 32, 3, 3)), ("decoder.deconv_layers.3.bias", (128,)),
            ("decoder.deconv_layers.6.weight", (4, 32, 3, 3)), ("decoder.deconv_layers.6.bias", (4,))]
    d = "kalman_filter.dyn_params."
    out += [(d + "A", (K, n, n)), (d + "B", (K, n, m)), (d + "C", (K, p, n))]
    if kind == "lstm":
        if K > 1:
            out += [(d + "lstm.weight_ih_l0", (4 * hidden, p)), (d + "lstm.weight_hh_l0", (4 * hidden, hidden)),
                    (d + "lstm.bias_ih_l0", (4 * hidden,)), (d + "lstm.bias_hh_l0", (4 * hidden,)),
                    (d + "head_w.weight", (K, hidden)), (d + "head_w.bias", (K,))]
    else:
        out += [(d + "Q", (K, n, n))]
        g = d + "markov_regime_posterior."
        for sfx in ("", "_reverse"):
            out += [(g + f"bigru.weight_ih_l0{sfx}", (3 * hidden, p)), (g + f"bigru.weight_hh_l0{sfx}", (3 * hidden, hidden)),
                    (g + f"bigru.bias_ih_l0{sfx}", (3 * hidden,)), (g + f"bigru.bias_hh_l0{sfx}", (3 * hidden,))]
        out += [(g + "linear_head.weight", (K * K, 2 * hidden)), (g + "linear_head.bias", (K * K,)),
                (g + "init_head.weight", (K, 2 * hidden)), (g + "init_head.bias", (K,))]
    return out


def stability_state_dict(kind, K, n=4, p=2):
    """The weights of the reference's stability recipe (tests/test_imputation_stability.py:16-22):
    torch.manual_seed(42); every parameter <- 0.01*randn_like, in .parameters() order; buffers
    keep their KVAE.__init__ values (model.py:71-76)."""
    torch.manual_seed(42)
    sd = {}
    for name, shape in reference_param_layout(kind, K):
        sd[name] = torch.randn(shape) * 0.01
    sd["kalman_filter.Q"] = 0.02 * torch.eye(n)
    sd["kalman_filter.R"] = 0.03 * torch.eye(p)
    sd["kalman_filter.I"] = torch.eye(n)
    sd["kalman_filter.mu0"] = torch.zeros(n)
    sd["kalman_filter.Sigma0"] = 20.0 * torch.eye(n)
    return sd


# round-2 fixtures (tests/golden/make_goldens_r2.py): (name, expected [level Sigma_s, level Q_t]); 5 = diagonal fallback
JITTER_CASES = [("jitter_q_level1", [0, 1]), ("jitter_sigma_level2", [2, 0]), ("jitter_diag_fallback", [0, 5]),
                # round 3, z_dim = 16 (tests/golden/make_goldens_r3.py): the shape of the matrix-core ELBO kernels
                ("jitter_n16_q_level1", [0, 1]), ("jitter_n16_sigma_level2", [2, 0]), ("jitter_n16_diag_fallback", [0, 5]),
                ("jitter_n16_sigma_diag_fallback", [5, 0])]
JITTER_GRADS = ["mu_s", "Sig_s", "a", "A_list", "B_list", "C_list", "Q_list"]
