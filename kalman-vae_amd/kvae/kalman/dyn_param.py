"""LSTM alpha-network dynamics of the original KVAE ("lstm" mode).

Same constructor, parameters and state_dict keys as the reference's DynamicsParameter
(kvae/kalman/dyn_param.py:5-63): A[K,n,n], B[K,n,m], C[K,p,n], lstm.*, head_w.*.
On a HIP device the recurrence is hand-written HIP: `kvae_lstm_fwd/bwd` for a fully observed sequence (the whole T loop
in one launch; the three weight gradients are one `kvae_rnn_wgrad` reduction over its d_pre output) and, when a mask is given, the
cell runs INSIDE the filter kernel (`kvae_lgssm_filter_alpha_lstm` / `kvae_lgssm_alpha_lstm_bwd`, kalman_filter.py here).
The mixing A_t = sum_k alpha_tk A_k (dyn_param.py:58-60) is `kvae_mix_fwd/bwd` and yields one packed step record.
Hidden sizes above 52 fall back to nn.LSTM (MIOpen), which cannot be captured into a hipGraph.
"""
import torch
import torch.nn as nn

from .. import _native
from .lgssm_ops import LstmSequence, Slots, mix_dynamics, small_linear


class DynamicsParameter(nn.Module):
    def __init__(self, A, B, C, hidden_lstm=50):
        super().__init__()
        self.is_switching_dynamics = False
        self.K = A.size(0)
        self.n, self.m, self.p = A.size(1), B.size(2), C.size(1)
        self.A = nn.Parameter(A.clone())
        self.B = nn.Parameter(B.clone())
        self.C = nn.Parameter(C.clone())
        self.lstm_state = None
        self.state_seq = None
        if self.K > 1:
            self.lstm = nn.LSTM(input_size=self.p, hidden_size=hidden_lstm, num_layers=1, batch_first=True)
            self.head_w = nn.Linear(hidden_lstm, self.K)
            with torch.no_grad():  # start with all weight on mode 0
                self.head_w.bias.fill_(-10.0)
                self.head_w.bias[0] = 0.0

    def reset_state(self):
        self.lstm_state = None
        self.state_seq = []

    # ---- whole-sequence path (all frames observed): alpha_t depends on a_{t-1} only -------------
    def alpha_sequence(self, a_seq):
        """alpha[B,T,K] for a fully observed sequence: the LSTM input at step t is a_{t-1} (zeros at
        t = 0), exactly what the reference's per-step loop feeds it when mask == 1
        (kalman_filter.py:142,183-185); one HIP launch for the whole recurrence instead of T cell launches."""
        Bsz, T, _ = a_seq.shape
        if self.K == 1:
            return torch.ones(Bsz, T, 1, device=a_seq.device, dtype=a_seq.dtype)
        shifted = torch.cat([a_seq.new_zeros(Bsz, 1, self.p), a_seq[:, :-1]], dim=1)
        if self.lstm.hidden_size <= _native.LSTM_MAX_H and self.p <= _native.LSTM_MAX_I:
            # hand-written HIP recurrence: one launch, weights in LDS, hipGraph-capturable
            h = LstmSequence.apply(shifted, self.lstm.weight_ih_l0, self.lstm.weight_hh_l0, self.lstm.bias_ih_l0,
                                   self.lstm.bias_hh_l0)
            self.lstm_state = None
        else:  # hidden sizes beyond the LDS-resident kernel: PyTorch-ROCm (MIOpen) recurrence
            h, self.lstm_state = self.lstm(shifted, None)
        return small_linear(h, self.head_w, softmax=True)   # head + softmax: one launch each way on a HIP device

    def step_record(self, alpha):
        """(record [B,T,E], Slots) with A|B|C mixed by alpha [B,T,K] (K > 1)."""
        rec, offs, views = mix_dynamics(alpha, [self.A, self.B, self.C])
        return rec, Slots(A=offs[0], B=offs[1], C=offs[2]), views

    # ---- per-step path (reference API; needed when frames are missing) -----------------------------
    def compute_step(self, a_tprev):
        Bsz = a_tprev.size(0)
        if self.K == 1:
            self.state_seq.append(torch.ones(Bsz, 1, device=a_tprev.device, dtype=a_tprev.dtype))
            return (self.A[0].expand(Bsz, -1, -1), self.B[0].expand(Bsz, -1, -1), self.C[0].expand(Bsz, -1, -1))
        h, self.lstm_state = self.lstm(a_tprev.unsqueeze(1), self.lstm_state)
        w = torch.softmax(self.head_w(h.squeeze(1)), dim=-1)
        _, _, (A, B, C) = self.step_record(w.unsqueeze(1))
        self.state_seq.append(w)
        return A[:, 0], B[:, 0], C[:, 0]
