"""Switching linear dynamics ("switching" mode): a bi-GRU Markov regime posterior, Gumbel-softmax
regime samples, a sticky Markov prior, and regime-mixed A_t, B_t, Q_t.

Same classes, constructor signatures and state_dict keys as the reference
(kvae/kalman/switch_dyn_param.py:7-129).  On a HIP device with the default shapes (hidden 50, input 2) the bi-GRU
recurrence is the hand-written `kvae_bigru_fwd/bwd` kernel (its weight gradients: one `kvae_rnn_wgrad` call),
the Gumbel-softmax regime chain is `kvae_regime_fwd/bwd`, and the mixing einsums (switch_dyn_param.py:82-84) are
`kvae_mix_fwd/bwd` producing one packed A|B|Q step record; the two linear heads are `kvae_linear_fwd/bwd_input`.  Other GRU shapes
take nn.GRU (MIOpen).  Gumbel noise can be injected (kvae.noise.inject) for parity tests.
"""
import torch
import torch.nn as nn
from torch.distributions import Multinomial

from .. import noise
from .. import _native
from .lgssm_ops import BiGruSequence, RegimeChain, Slots, mix_dynamics, small_linear


def _gumbel_softmax(logits, g, tau, hard):
    y_soft = ((logits + g) / tau).softmax(-1)
    if not hard:
        return y_soft
    idx = y_soft.argmax(-1, keepdim=True)
    return torch.zeros_like(logits).scatter_(-1, idx, 1.0) - y_soft.detach() + y_soft


class SwitchingDynamicsParameter(nn.Module):
    def __init__(self, A, B, C, Q=None, prior=None, hidden_lstm=32, markov_regime_posterior=None):
        super().__init__()
        self.is_switching_dynamics = True
        self.K = A.size(0)
        self.n, self.m, self.p = A.size(1), B.size(2), C.size(1)
        self._tau, self._tau_dev = 0.5, None
        if Q is None:
            Q = torch.eye(self.n, device=A.device, dtype=A.dtype).repeat(self.K, 1, 1)
        self.A = nn.Parameter(A.clone())
        self.B = nn.Parameter(B.clone())
        self.C = nn.Parameter(C.clone())
        self.Q = nn.Parameter(Q.clone())
        self.s_tprev = None
        self.prior = prior if prior is not None else StickyRegimePrior(self.K)
        self.markov_regime_posterior = markov_regime_posterior or MarkovVariationalRegimePosterior(
            self.K, input_dim=self.p, hidden_size=hidden_lstm)
        self.hidden_size = hidden_lstm
        self.state_seq = None
        self._record = self._slots = None

    # The Gumbel-softmax temperature.  The reference's epoch loop assigns `dyn_params.tau = max(tau_min, tau * rate)`
    # (kvae/train/train.py:270-274); here the value is mirrored into a device scalar that the regime-chain kernels read
    # at run time, so the assignment also takes effect on a step that was captured into a hipGraph.
    @property
    def tau(self):
        return self._tau

    @tau.setter
    def tau(self, value):
        self._tau = float(value)
        if self._tau_dev is not None:
            self._tau_dev.fill_(self._tau)

    def tau_scalar(self, dev):
        """0-d fp32 device tensor holding tau (created once per device; call once OUTSIDE graph capture)."""
        if self._tau_dev is None or self._tau_dev.device != dev:
            self._tau_dev = torch.full((), self._tau, device=dev, dtype=torch.float32)
        return self._tau_dev

    def reset_state(self):
        self.state_seq = None
        self._record = self._slots = None

    def _prior_matrix(self, dev, dt):
        """Device copy of the sticky prior, made once per device (a host->device copy is not allowed inside
        hipGraph capture, and the reference re-uploads it on every call, switch_dyn_param.py:65)."""
        P = self.prior.transition_matrix
        cache = getattr(self, "_P_cache", None)
        if cache is None or cache[0] is not P or cache[1].device != dev or cache[1].dtype != dt:
            cache = (P, P.to(device=dev, dtype=dt))
            self._P_cache = cache
        return cache[1]

    def regime_chain(self, logits, init_logits, gumbel, hard):
        """Sequential Gumbel-softmax Markov chain (switch_dyn_param.py:52-79).
        Returns y_seq [B,T,K], log_qseq [B,T], log_pseq [B,T]."""
        Bsz, T, K, _ = logits.shape
        P = self._prior_matrix(logits.device, logits.dtype)
        y = _gumbel_softmax(init_logits, gumbel[:, 0], self.tau, hard)
        log_q0 = torch.log_softmax(init_logits, dim=-1)
        ys = [y]
        lq = [(y * log_q0).sum(-1)]
        lp = [(y * torch.full_like(log_q0, 1.0 / K).log()).sum(-1)]
        for t in range(1, T):
            l_t = torch.bmm(y.unsqueeze(1), logits[:, t]).squeeze(1)
            y_t = _gumbel_softmax(l_t, gumbel[:, t], self.tau, hard)
            lq.append((y_t * torch.log_softmax(l_t, dim=-1)).sum(-1))
            lp.append((y_t * torch.log((y @ P).clamp_min(1e-8))).sum(-1))
            ys.append(y_t)
            y = y_t
        return torch.stack(ys, 1), torch.stack(lq, 1), torch.stack(lp, 1)

    def compute_batch(self, a_seq, is_training=True):
        Bsz, T, _ = a_seq.size()
        dev, dt = a_seq.device, a_seq.dtype
        if self.K == 1:
            ex = lambda M: M[0].expand(Bsz, T, -1, -1)
            self.log_qseq = torch.zeros(Bsz, T, device=dev, dtype=dt)
            self.log_pseq = torch.zeros(Bsz, T, device=dev, dtype=dt)
            self.Q_seq = ex(self.Q)
            self.state_seq = torch.ones(Bsz, T, 1, device=dev, dtype=dt)
            self._record = self._slots = None
            return ex(self.A), ex(self.B), ex(self.C), self.Q_seq
        logits, init_logits = self.markov_regime_posterior(a_seq)
        gumbel = noise.take("gumbel")
        if gumbel is None:
            gumbel = -torch.empty(Bsz, T, self.K, device=dev, dtype=dt).exponential_().log()
        else:
            gumbel = gumbel.to(device=dev, dtype=dt)
        if _native.fused_ok(logits) and self.K <= 16:   # one HIP launch (csrc/regime.h) instead of the T-1 step loop
            P = self._prior_matrix(dev, dt)
            tau = self.tau_scalar(dev) if logits.is_cuda else self.tau
            y_seq, self.log_qseq, self.log_pseq = RegimeChain.apply(logits, init_logits, gumbel, P, tau, not is_training)
        else:
            y_seq, self.log_qseq, self.log_pseq = self.regime_chain(logits, init_logits, gumbel, hard=not is_training)
        rec, offs, (A_seq, B_seq, Q_seq) = mix_dynamics(y_seq, [self.A, self.B, self.Q])
        self._record, self._slots = rec, Slots(A=offs[0], B=offs[1], Q=offs[2])
        C_seq = self.C[0].expand(Bsz, T, -1, -1)   # emission shared across regimes (switch_dyn_param.py:85-86)
        self.Q_seq = Q_seq
        self.state_seq = y_seq
        return A_seq, B_seq, C_seq, Q_seq

    def elbo_terms(self):
        return self.log_qseq, self.log_pseq


class StickyRegimePrior:
    """p(s_t | s_{t-1}) = p_stay on the diagonal, uniform elsewhere."""

    def __init__(self, K, p_stay=0.9):
        self.K, self.p_stay = K, p_stay
        if K > 1:
            self.transition_matrix = torch.full((K, K), (1 - p_stay) / (K - 1))
            self.transition_matrix.fill_diagonal_(p_stay)
        else:  # the reference divides by zero here; a single regime always stays
            self.transition_matrix = torch.ones(1, 1)

    def reset_state(self):
        self.state_probabilities = Multinomial(probs=torch.ones(self.K) / self.K)

    def compute_step(self, prev_probs):
        return prev_probs @ self.transition_matrix


class MarkovVariationalRegimePosterior(nn.Module):
    """bi-GRU over a_{1:T} -> transition logits [B,T,K,K] and initial logits [B,K]."""

    def __init__(self, K, input_dim, hidden_size=32):
        super().__init__()
        self.K, self.hidden_size = K, hidden_size
        self.bigru = nn.GRU(input_size=input_dim, hidden_size=hidden_size, num_layers=1, batch_first=True,
                            bidirectional=True)
        self.linear_head = nn.Linear(2 * hidden_size, K * K)
        self.init_head = nn.Linear(2 * hidden_size, K)

    def forward(self, a_seq):
        g = self.bigru
        if a_seq.is_cuda and a_seq.dtype == torch.float32 and (g.hidden_size, g.input_size) == BiGruSequence.SUPPORTED:
            # hand-written HIP recurrence (csrc/gru_fast.h): both directions in one launch, hipGraph-capturable
            h_seq = BiGruSequence.apply(a_seq, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0,
                                        g.weight_ih_l0_reverse, g.weight_hh_l0_reverse, g.bias_ih_l0_reverse,
                                        g.bias_hh_l0_reverse)
        else:   # other shapes / host tensors: PyTorch (MIOpen) GRU
            h_seq, _ = self.bigru(a_seq)
        logits = small_linear(h_seq, self.linear_head).unflatten(-1, (self.K, self.K))
        return logits, small_linear(h_seq[:, 0], self.init_head)
