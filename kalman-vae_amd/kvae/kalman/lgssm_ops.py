"""torch.autograd bridge to the HIP LGSSM kernels (C ABI: include/kvae_lgssm.h).

PyTorch is plumbing here: it owns device memory, the current HIP stream and the autograd graph.
Every numerical step of filter / RTS smoother / ELBO / mixing / recurrences / linear heads — forward and backward,
including the weight gradients of the LSTM / bi-GRU / heads (`kvae_rnn_wgrad`: no library GEMM on the path) — runs in
libkvae_lgssm.so; only trivial reductions (`terms.sum`) stay torch ops on the same stream.  Nothing computes on the host.

Per-step operands (A_t, B_t, C_t, Q_t) reach the kernels as strided "stacks": either slots of ONE
packed record tensor [B,T,E] produced by `mix_dynamics` (mixture-of-K case: a single launch writes
a whole step record, a single tensor carries its gradient) or a plain tensor that is broadcast
([r,c]) or per-step ([B,T,r,c]).
"""
import ctypes as C
from typing import NamedTuple, Optional

import torch

from .. import _native as N


class Slots(NamedTuple):
    """Float offsets of A|B|C|Q inside one packed step record (None = operand is not packed)."""
    A: Optional[int] = None
    B: Optional[int] = None
    C: Optional[int] = None
    Q: Optional[int] = None


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _stack(t, Bsz, T, r, c, packed=None, off=None):
    """(tensor_to_keep_alive, N.Stack) for one per-step operand."""
    if off is not None:
        E = packed.shape[-1]
        return packed, N.Stack(packed.data_ptr() + 4 * off, T * E, E)
    if t.dim() == 2:
        t = _f32c(t)
        return t, N.Stack(t.data_ptr(), 0, 0)
    if t.shape != (Bsz, T, r, c):
        t = t.expand(Bsz, T, r, c)
    if t.dtype != torch.float32 or t.stride(-1) != 1 or t.stride(-2) != c:
        t = t.float().contiguous()
    return t, N.Stack(t.data_ptr(), t.stride(0), t.stride(1))


class _Call:
    """Builds the kvae_lgssm_problem for one call and keeps every tensor alive until it returns."""

    def __init__(self, Y, U, mask, packed, A, Bm, Cm, Q, R, mu0, Sigma0, slots):
        self.Y, self.U = _f32c(Y), _f32c(U)
        Bsz, T, p = self.Y.shape
        m = self.U.shape[-1]
        n = Sigma0.shape[-1]
        self.dims = (Bsz, T, n, m, p)
        self.mask = _f32c(mask)
        self.packed = _f32c(packed)
        self.R, self.mu0, self.Sigma0 = _f32c(R), _f32c(mu0), _f32c(Sigma0)
        self.keep = []
        prob = N.Problem()
        prob.B, prob.T, prob.n, prob.m, prob.p = Bsz, T, n, m, p
        for name, t, r, c, off in (("A", A, n, n, slots.A), ("Bm", Bm, n, m, slots.B),
                                   ("C", Cm, p, n, slots.C), ("Q", Q, n, n, slots.Q)):
            keep, st = _stack(t, Bsz, T, r, c, self.packed, off)
            self.keep.append(keep)
            setattr(prob, name, st)
        prob.R = self.R.data_ptr()
        prob.mu0 = self.mu0.data_ptr()
        prob.mu0_sb = n if self.mu0.dim() == 2 else 0
        prob.Sigma0 = self.Sigma0.data_ptr()
        prob.Sigma0_sb = n * n if self.Sigma0.dim() == 3 else 0
        prob.Y, prob.U = self.Y.data_ptr(), self.U.data_ptr()
        prob.mask = self.mask.data_ptr() if self.mask is not None else None
        self.prob = prob
        self.lib = N.lib_for(self.Y)
        self.stream = N.stream_for(self.Y)


def _states(mf, Sf, mp, Sp, ms=None, Ss=None, aux=None):
    st = N.States()
    for k, t in (("mus_filt", mf), ("Sigmas_filt", Sf), ("mus_pred", mp), ("Sigmas_pred", Sp),
                 ("mus_smooth", ms), ("Sigmas_smooth", Ss), ("aux", aux)):
        setattr(st, k, t.data_ptr() if t is not None else None)
    return st


class _GradSink:
    """Gradient buffers for the per-step operands + the autograd return values built from them."""

    def __init__(self, call, packed, A, Bm, Cm, Q, slots, need_q):
        Bsz, T, n, m, p = call.dims
        dev = call.Y.device
        self.g = N.InputGrads()
        self.out = {}
        self.gpacked = torch.empty_like(call.packed) if packed is not None else None
        for name, t, r, c, off in (("gA", A, n, n, slots.A), ("gB", Bm, n, m, slots.B),
                                   ("gC", Cm, p, n, slots.C), ("gQ", Q, n, n, slots.Q)):
            if off is not None:
                E = self.gpacked.shape[-1]
                setattr(self.g, name, N.Stack(self.gpacked.data_ptr() + 4 * off, T * E, E))
            elif name == "gQ" and not need_q:
                setattr(self.g, name, N.Stack(None, 0, 0))
            else:
                buf = torch.empty(Bsz, T, r, c, device=dev, dtype=torch.float32)
                self.out[name] = (buf, t)
                setattr(self.g, name, N.Stack(buf.data_ptr(), T * r * c, r * c))
        self.gY = torch.empty_like(call.Y)
        self.gU = torch.empty_like(call.U)
        self.g.gY, self.g.gU = self.gY.data_ptr(), self.gU.data_ptr()

    def operand_grad(self, name, scale=None):
        """Gradient for a non-packed operand, reduced to the shape the caller passed in."""
        if name not in self.out:
            return None
        buf, t = self.out[name]
        if scale is not None:
            buf = buf * scale
        if t.dim() == 2:
            return buf.sum((0, 1))
        return buf.sum_to_size(t.shape) if tuple(t.shape) != tuple(buf.shape) else buf


# ------------------------------------------------------------------------------------------------
# filter (+ RTS smoother)
# ------------------------------------------------------------------------------------------------
class LgssmSmooth(torch.autograd.Function):
    """KalmanFilter.filter / .smooth (reference kalman_filter.py:107-201, 240-279) in one launch."""

    @staticmethod
    def forward(ctx, Y, U, mask, packed, A, Bm, Cm, Q, R, mu0, Sigma0, slots, with_rts):
        call = _Call(Y, U, mask, packed, A, Bm, Cm, Q, R, mu0, Sigma0, slots)
        Bsz, T, n, m, p = call.dims
        dev = call.Y.device
        mk = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        mf, Sf, mp, Sp = mk(Bsz, T, n), mk(Bsz, T, n, n), mk(Bsz, T, n), mk(Bsz, T, n, n)
        ms, Ss = (mk(Bsz, T, n), mk(Bsz, T, n, n)) if with_rts else (None, None)
        # gains (K | S | J) kept for the backward; consumed by the n=4,p=2 fused-phase kernels
        aux = mk(Bsz, T, n * p + p * p + n * n) if any(ctx.needs_input_grad) else None
        st = _states(mf, Sf, mp, Sp, ms, Ss, aux)
        fn = call.lib.dll.kvae_lgssm_smooth_fwd if with_rts else call.lib.dll.kvae_lgssm_filter_fwd
        call.lib.check(N.timed("smooth_fwd" if with_rts else "filter_fwd", call.Y,
                               lambda: fn(C.byref(call.prob), C.byref(st), call.stream)), "kvae_lgssm_smooth_fwd")
        ctx.slots, ctx.with_rts = slots, with_rts
        # outputs nobody differentiates arrive as None in backward (not as materialised zero stacks): the kernels take
        # NULL for "no upstream gradient" and skip the loads
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(Y, U, mask, packed, A, Bm, Cm, Q, R, mu0, Sigma0, mf, Sf, mp, Sp, ms, Ss, aux)
        if with_rts:
            return ms, Ss, mf, Sf, mp, Sp
        return mf, Sf, mp, Sp

    @staticmethod
    def backward(ctx, *gouts):
        Y, U, mask, packed, A, Bm, Cm, Q, R, mu0, Sigma0, mf, Sf, mp, Sp, ms, Ss, aux = ctx.saved_tensors
        slots, with_rts = ctx.slots, ctx.with_rts
        call = _Call(Y, U, mask, packed, A, Bm, Cm, Q, R, mu0, Sigma0, slots)
        Bsz, T, n, m, p = call.dims
        if with_rts:
            g_ms, g_Ss, g_mf, g_Sf, g_mp, g_Sp = (_f32c(g) for g in gouts)
        else:
            g_mf, g_Sf, g_mp, g_Sp = (_f32c(g) for g in gouts)
            g_ms = g_Ss = None
        need = ctx.needs_input_grad
        need_q = slots.Q is not None or (Q is not None and need[7])
        sink = _GradSink(call, packed, A, Bm, Cm, Q, slots, need_q)
        g0 = S0 = None
        if need[9]:
            g0 = torch.empty(Bsz, n, device=Y.device, dtype=torch.float32)
            sink.g.g_mu0 = g0.data_ptr()
        if need[10]:
            S0 = torch.empty(Bsz, n, n, device=Y.device, dtype=torch.float32)
            sink.g.g_Sigma0 = S0.data_ptr()
        ws = torch.empty(Bsz, T, 2 * (n + n * n), device=Y.device, dtype=torch.float32)
        saved = _states(mf, Sf, mp, Sp, ms, Ss, aux)
        up = _states(g_mf, g_Sf, g_mp, g_Sp, g_ms, g_Ss)
        call.lib.check(N.timed("smooth_bwd", call.Y, lambda: call.lib.dll.kvae_lgssm_smooth_bwd(
            C.byref(call.prob), C.byref(saved), C.byref(up), C.byref(sink.g), N.ptr(ws), int(with_rts), call.stream)),
            "kvae_lgssm_smooth_bwd")
        if g0 is not None and mu0.dim() == 1:
            g0 = N.colsum(g0)
        if S0 is not None and Sigma0.dim() == 2:
            S0 = N.colsum(S0)
        return (sink.gY if need[0] else None, sink.gU if need[1] else None, None, sink.gpacked,
                sink.operand_grad("gA"), sink.operand_grad("gB"), sink.operand_grad("gC"),
                sink.operand_grad("gQ") if need_q else None, None, g0, S0, None, None)


# ------------------------------------------------------------------------------------------------
# ELBO
# ------------------------------------------------------------------------------------------------
class LgssmElbo(torch.autograd.Function):
    """The LGSSM terms of KalmanFilter.elbo (reference kalman_filter.py:347-389), summed over B and T.
    Returns (total, per_term[4], levels int32[3]); gradients are produced in the forward launch (unit upstream) and
    scaled by the incoming scalar gradient in backward.  levels (device, no host sync): the whole-batch _safe_cholesky level
    the launch resolved for Sigma_s and for Q_t (0..4 = jitter 1e-6 * 10^level, 5 = diagonal fallback) and the kernel family
    that computed the call (include/kvae_lgssm.h)."""

    @staticmethod
    def forward(ctx, mus, Sigs, eps, Y, U, mask, packed, A, Bm, Cm, Q, R, mu0, Sigma0, slots):
        call = _Call(Y, U, mask, packed, A, Bm, Cm, Q, R, mu0, Sigma0, slots)
        Bsz, T, n, m, p = call.dims
        dev = call.Y.device
        mus_c = _f32c(mus.reshape(Bsz, T, n))
        Sigs_c, eps_c = _f32c(Sigs), _f32c(eps)
        terms = torch.empty(Bsz, T, 4, device=dev, dtype=torch.float32)
        levels = torch.empty(3, device=dev, dtype=torch.int32)   # (level of Sigma_s, level of Q_t, kernel family of the launch)
        ws_lz = torch.empty(Bsz, T, n, device=dev, dtype=torch.float32)   # z_t parked by the probe launch
        want = any(ctx.needs_input_grad)
        g_mus = g_Sigs = sink = None
        if want:
            need_q = slots.Q is not None or (Q is not None and ctx.needs_input_grad[10])
            sink = _GradSink(call, packed, A, Bm, Cm, Q, slots, need_q)
            g_mus, g_Sigs = torch.empty_like(mus_c), torch.empty_like(Sigs_c)
            ctx.need_q = need_q
        call.lib.check(N.timed("elbo", call.Y, lambda: call.lib.dll.kvae_lgssm_elbo(
            C.byref(call.prob), N.ptr(mus_c), N.ptr(Sigs_c), N.ptr(eps_c), N.ptr(terms), N.ptr(levels), N.ptr(ws_lz), N.ptr(g_mus),
            N.ptr(g_Sigs), C.byref(sink.g) if sink else None, call.stream)), "kvae_lgssm_elbo")
        per_term = terms.sum((0, 1))
        ctx.sink, ctx.g_mus, ctx.g_Sigs, ctx.mus_shape = sink, g_mus, g_Sigs, mus.shape
        ctx.mark_non_differentiable(per_term, levels)
        ctx.set_materialize_grads(False)
        return per_term.sum(), per_term, levels

    @staticmethod
    def backward(ctx, g_total, _g_terms, _g_levels):
        if g_total is None:
            return (None,) * 15
        sink, need = ctx.sink, ctx.needs_input_grad
        s = g_total
        gp = sink.gpacked * s if sink.gpacked is not None else None
        return ((ctx.g_mus * s).reshape(ctx.mus_shape) if need[0] else None,
                ctx.g_Sigs * s if need[1] else None, None,
                sink.gY * s if need[3] else None, sink.gU * s if need[4] else None, None, gp,
                sink.operand_grad("gA", s), sink.operand_grad("gB", s), sink.operand_grad("gC", s),
                sink.operand_grad("gQ", s) if ctx.need_q else None, None, None, None, None)


# ------------------------------------------------------------------------------------------------
# mixture-of-K dynamics
# ------------------------------------------------------------------------------------------------
class MixDynamics(torch.autograd.Function):
    """record[b,t,:] = sum_k alpha[b,t,k] base[k,:]  (reference dyn_param.py:58-60,
    switch_dyn_param.py:82-84), base = the K flattened matrices packed side by side."""

    @staticmethod
    def forward(ctx, alpha, base):
        a, bs = _f32c(alpha), _f32c(base)
        Bsz, T, K = a.shape
        E = bs.shape[1]
        out = torch.empty(Bsz, T, E, device=a.device, dtype=torch.float32)
        lib = N.lib_for(a)
        lib.check(lib.dll.kvae_mix_fwd(N.ptr(a), N.ptr(bs), N.ptr(out), Bsz * T, K, E, N.stream_for(a)), "kvae_mix_fwd")
        ctx.save_for_backward(a, bs)
        return out

    @staticmethod
    def backward(ctx, g_out):
        a, bs = ctx.saved_tensors
        g = _f32c(g_out)
        Bsz, T, K = a.shape
        E = bs.shape[1]
        lib = N.lib_for(a)
        nblk = lib.dll.kvae_mix_bwd_partials(Bsz * T)
        partials = torch.empty(nblk, K, E, device=a.device, dtype=torch.float32)
        g_alpha, g_base = torch.empty_like(a), torch.empty_like(bs)
        lib.check(lib.dll.kvae_mix_bwd(N.ptr(a), N.ptr(bs), N.ptr(g), N.ptr(g_alpha), N.ptr(g_base), N.ptr(partials),
                                       Bsz * T, K, E, 0, N.stream_for(a)), "kvae_mix_bwd")
        return g_alpha, g_base


def mix_dynamics(alpha, mats):
    """alpha [B,T,K]; mats: list of [K,r,c] parameters. Returns (record [B,T,E], offsets, views)."""
    base = torch.cat([mt.reshape(mt.shape[0], -1) for mt in mats], dim=1)
    rec = MixDynamics.apply(alpha, base)
    offs, views, o = [], [], 0
    for mt in mats:
        e = mt.shape[1] * mt.shape[2]
        offs.append(o)
        views.append(rec[..., o:o + e].unflatten(-1, (mt.shape[1], mt.shape[2])))
        o += e
    return rec, offs, views


# ------------------------------------------------------------------------------------------------
# parameter gradients of the recurrences and heads (csrc/rnn_wgrad.h), linear heads (csrc/small_linear.h)
# ------------------------------------------------------------------------------------------------
def rnn_wgrad(ref, problems):
    """Up to four reductions G = D^T [h_shifted | x | 1] over the (sequence, step) rows in one pair of launches.
    problems: dicts with d [N,R]; optional h (2-D view, unit column stride) with `shift` and `T`; optional x [N,I]; `bias`.
    Returns one (g_wh [R,H] | None, g_wx [R,I] | None, g_b [R] | None) per problem."""
    lib = N.lib_for(ref)
    arr = (N.WgradProblem * len(problems))()
    outs, keep = [], []
    for slot, pr in zip(arr, problems):
        d, h, x = pr["d"], pr.get("h"), pr.get("x")
        n_rows, R = d.shape
        assert d.stride(1) == 1 and (h is None or h.stride(1) == 1) and (x is None or x.stride(1) == 1)
        H, I, bias = (h.shape[1] if h is not None else 0), (x.shape[1] if x is not None else 0), int(bool(pr.get("bias", True)))
        g_wh = torch.empty(R, H, device=d.device, dtype=torch.float32) if H else None
        g_wx = torch.empty(R, I, device=d.device, dtype=torch.float32) if I else None
        g_b = torch.empty(R, device=d.device, dtype=torch.float32) if bias else None
        slot.d, slot.h, slot.x = d.data_ptr(), (h.data_ptr() if H else None), (x.data_ptr() if I else None)
        slot.g_wh, slot.g_wx, slot.g_b = (t.data_ptr() if t is not None else None for t in (g_wh, g_wx, g_b))
        slot.d_stride, slot.h_stride, slot.x_stride = d.stride(0), (h.stride(0) if H else 0), (x.stride(0) if I else 0)
        slot.N, slot.R, slot.H, slot.I, slot.bias = n_rows, R, H, I, bias
        slot.T, slot.shift = int(pr.get("T", 1)), int(pr.get("shift", 0))
        outs.append((g_wh, g_wx, g_b))
        keep += [d, h, x]
    ws = torch.empty(int(lib.dll.kvae_rnn_wgrad_ws_floats(arr, len(problems))), device=ref.device, dtype=torch.float32)
    lib.check(N.timed("rnn_wgrad", ref, lambda: lib.dll.kvae_rnn_wgrad(arr, len(problems), N.ptr(ws), N.stream_for(ref))),
              "kvae_rnn_wgrad")
    return outs


def small_linear_supported(x, weight, softmax=False):
    O, F = weight.shape
    return (N.fused_ok(x) and x.dtype == torch.float32 and weight.dtype == torch.float32 and F <= 128 and O * F <= 12288
            and (not softmax or O <= 16) and x.shape[-1] == F and x.stride(-1) == 1
            and (x.dim() == 2 or x.is_contiguous()))


class SmallLinear(torch.autograd.Function):
    """y = x W^T + b (optionally followed by a softmax over the outputs) for the heads of the alpha-networks
    (reference dyn_param.py:53-56: head_w + softmax; switch_dyn_param.py:119-129: linear_head / init_head): one launch
    forward, one for the input gradient, and the rnn_wgrad reduction for dW | db - instead of addmm / softmax / two GEMMs /
    a column sum from the BLAS library.  x: [..., F] contiguous, or a 2-D view with unit column stride (h_seq[:, 0])."""

    @staticmethod
    def forward(ctx, x, weight, bias, softmax):
        w, b = _f32c(weight), _f32c(bias)
        O, F = w.shape
        x2 = x if x.dim() == 2 else x.reshape(-1, F)
        y = torch.empty(x2.shape[0], O, device=x.device, dtype=torch.float32)
        lib = N.lib_for(x)
        lib.check(N.timed("linear_fwd", x, lambda: lib.dll.kvae_linear_fwd(
            N.ptr(x2), x2.stride(0), x2.shape[0], F, N.ptr(w), N.ptr(b), O, int(softmax), N.ptr(y), N.stream_for(x))), "kvae_linear_fwd")
        ctx.softmax, ctx.lead = bool(softmax), x.shape[:-1]
        ctx.save_for_backward(x2, w, y if softmax else None)
        return y.reshape(*x.shape[:-1], O)

    @staticmethod
    def backward(ctx, g):
        x2, w, y = ctx.saved_tensors
        O, F = w.shape
        g2 = _f32c(g).reshape(-1, O)
        lib = N.lib_for(g2)
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        dx = gl = None
        if need_x or ctx.softmax:
            dx = torch.empty(x2.shape[0], F, device=g2.device, dtype=torch.float32)
            gl = torch.empty_like(g2) if ctx.softmax else None
            lib.check(N.timed("linear_bwd", g2, lambda: lib.dll.kvae_linear_bwd_input(
                N.ptr(g2), N.ptr(y), x2.shape[0], F, N.ptr(w), O, N.ptr(gl), N.ptr(dx), F, N.stream_for(g2))), "kvae_linear_bwd_input")
        g_w = g_b = None
        if need_w or need_b:
            (g_w, _, g_b), = rnn_wgrad(g2, [dict(d=gl if ctx.softmax else g2, h=x2, bias=True)])
        return (dx.reshape(*ctx.lead, F) if need_x else None), g_w, g_b, None


def small_linear(x, linear, softmax=False):
    """nn.Linear `linear` (+ softmax) through SmallLinear when the shapes fit the kernels, else through torch."""
    if small_linear_supported(x, linear.weight, softmax):
        return SmallLinear.apply(x, linear.weight, linear.bias, softmax)
    y = linear(x)
    return torch.softmax(y, dim=-1) if softmax else y


# ------------------------------------------------------------------------------------------------
# alpha-network LSTM
# ------------------------------------------------------------------------------------------------
class LstmSequence(torch.autograd.Function):
    """h_seq = LSTM(x) from a zero state (single layer, batch_first, torch gate order): the HIP replacement
    for stepping nn.LSTM T times (reference dyn_param.py:50-52).  Backward: one BPTT launch + the rnn_wgrad reduction."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        x, w_ih, w_hh, b_ih, b_hh = (_f32c(t) for t in (x, w_ih, w_hh, b_ih, b_hh))
        Bsz, T, I = x.shape
        H = w_hh.shape[1]
        mk = lambda *s: torch.empty(*s, device=x.device, dtype=torch.float32)
        h, gates, c = mk(Bsz, T, H), mk(Bsz, T, 4 * H), mk(Bsz, T, H)
        lib = N.lib_for(x)
        lib.check(N.timed("lstm_fwd", x, lambda: lib.dll.kvae_lstm_fwd(
            N.ptr(x), N.ptr(w_ih), N.ptr(w_hh), N.ptr(b_ih), N.ptr(b_hh), N.ptr(h), N.ptr(gates), N.ptr(c),
            Bsz, T, I, H, N.stream_for(x))), "kvae_lstm_fwd")
        ctx.save_for_backward(x, w_ih, w_hh, h, gates, c)
        return h

    @staticmethod
    def backward(ctx, g_h):
        x, w_ih, w_hh, h, gates, c = ctx.saved_tensors
        g_h = _f32c(g_h)
        Bsz, T, I = x.shape
        H = w_hh.shape[1]
        d_pre = torch.empty(Bsz, T, 4 * H, device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        lib = N.lib_for(x)
        lib.check(N.timed("lstm_bwd", x, lambda: lib.dll.kvae_lstm_bwd(
            N.ptr(g_h), N.ptr(gates), N.ptr(c), N.ptr(w_ih), N.ptr(w_hh), N.ptr(d_pre), N.ptr(dx),
            Bsz, T, I, H, N.stream_for(x))), "kvae_lstm_bwd")
        if not any(ctx.needs_input_grad[1:]):   # frozen alpha-network ("vae" / "warmup" phase): the input gradient only
            return dx, None, None, None, None
        (g_whh, g_wih, g_b), = rnn_wgrad(x, [dict(d=d_pre.reshape(Bsz * T, 4 * H), h=h.reshape(Bsz * T, H), shift=-1, T=T,
                                                  x=x.reshape(Bsz * T, I))])
        return dx, g_wih, g_whh, g_b, g_b


# ------------------------------------------------------------------------------------------------
# regime chain of the switching dynamics
# ------------------------------------------------------------------------------------------------
class RegimeChain(torch.autograd.Function):
    """(y_seq, log_q, log_p) of the Gumbel-softmax Markov chain (reference switch_dyn_param.py:52-79): one
    launch forward, one BPTT launch backward instead of T-1 Python iterations of ~10 aten ops."""

    @staticmethod
    def forward(ctx, logits, init_logits, gumbel, P, tau, hard):
        """`tau`: a float (baked into the launch) or a 0-d fp32 tensor on the logits' device, which the kernel reads
        at run time - the form that follows the reference's tau schedule (train.py:270-274) under hipGraph replay."""
        logits, init_logits, gumbel, P = (_f32c(t) for t in (logits, init_logits, gumbel, P))
        Bsz, T, K, _ = logits.shape
        mk = lambda *s: torch.empty(*s, device=logits.device, dtype=torch.float32)
        y, lq, lp = mk(Bsz, T, K), mk(Bsz, T), mk(Bsz, T)
        lib = N.lib_for(logits)
        tau_t = tau if isinstance(tau, torch.Tensor) else None
        if tau_t is not None and (tau_t.device != logits.device or tau_t.dtype != torch.float32 or tau_t.numel() != 1):
            raise ValueError("RegimeChain: a tensor tau must be one fp32 element on the device of the logits")
        tau_f = 0.0 if tau_t is not None else float(tau)
        lib.check(N.timed("regime_fwd", logits, lambda: lib.dll.kvae_regime_fwd(
            N.ptr(logits), N.ptr(init_logits), N.ptr(gumbel), N.ptr(P), N.ptr(y), N.ptr(lq), N.ptr(lp), Bsz, T, K,
            tau_f, N.ptr(tau_t), int(hard), N.stream_for(logits))), "kvae_regime_fwd")
        ctx.tau, ctx.tau_t = tau_f, tau_t
        ctx.save_for_backward(logits, init_logits, gumbel, P, y)
        return y, lq, lp

    @staticmethod
    def backward(ctx, g_y, g_lq, g_lp):
        logits, init_logits, gumbel, P, y = ctx.saved_tensors
        Bsz, T, K, _ = logits.shape
        z = lambda t, *s: _f32c(t) if t is not None else torch.zeros(*s, device=logits.device, dtype=torch.float32)
        g_y, g_lq, g_lp = z(g_y, Bsz, T, K), z(g_lq, Bsz, T), z(g_lp, Bsz, T)
        g_logits, g_init = torch.empty_like(logits), torch.empty_like(init_logits)
        lib = N.lib_for(logits)
        lib.check(N.timed("regime_bwd", logits, lambda: lib.dll.kvae_regime_bwd(
            N.ptr(logits), N.ptr(init_logits), N.ptr(gumbel), N.ptr(P), N.ptr(y), N.ptr(g_y), N.ptr(g_lq), N.ptr(g_lp),
            N.ptr(g_logits), N.ptr(g_init), Bsz, T, K, ctx.tau, N.ptr(ctx.tau_t), N.stream_for(logits))), "kvae_regime_bwd")
        return g_logits, g_init, None, None, None, None


# ------------------------------------------------------------------------------------------------
# bidirectional GRU of the regime posterior
# ------------------------------------------------------------------------------------------------
def _ptr2(a, b):
    return (C.c_void_p * 2)(a.data_ptr(), b.data_ptr())


class BiGruSequence(torch.autograd.Function):
    """h_seq [B,T,2H] = nn.GRU(bidirectional, batch_first)(x) from zero states (reference switch_dyn_param.py:118,123):
    one launch for both directions forward, one BPTT launch backward, all eight parameter gradients in one rnn_wgrad call."""
    SUPPORTED = (50, 2)   # (hidden, input): the shape csrc/gru_fast.h is instantiated for

    @staticmethod
    def forward(ctx, x, wi0, wh0, bi0, bh0, wi1, wh1, bi1, bh1):
        x = _f32c(x)
        ws = [_f32c(t) for t in (wi0, wh0, bi0, bh0, wi1, wh1, bi1, bh1)]
        Bsz, T, I = x.shape
        H = ws[1].shape[1]
        h = torch.empty(Bsz, T, 2 * H, device=x.device, dtype=torch.float32)
        gates = torch.empty(2, Bsz, T, 4 * H, device=x.device, dtype=torch.float32)
        lib = N.lib_for(x)
        lib.check(N.timed("bigru_fwd", x, lambda: lib.dll.kvae_bigru_fwd(
            N.ptr(x), _ptr2(ws[0], ws[4]), _ptr2(ws[1], ws[5]), _ptr2(ws[2], ws[6]), _ptr2(ws[3], ws[7]), N.ptr(h),
            N.ptr(gates), Bsz, T, I, H, N.stream_for(x))), "kvae_bigru_fwd")
        ctx.save_for_backward(x, ws[0], ws[1], ws[4], ws[5], h, gates)
        return h

    @staticmethod
    def backward(ctx, g_h):
        x, wi0, wh0, wi1, wh1, h, gates = ctx.saved_tensors
        g_h = _f32c(g_h)
        Bsz, T, I = x.shape
        H = wh0.shape[1]
        mk = lambda *s: torch.empty(*s, device=x.device, dtype=torch.float32)
        dpi, dph, dx = mk(2, Bsz, T, 3 * H), mk(2, Bsz, T, 3 * H), mk(2, Bsz, T, I)
        lib = N.lib_for(x)
        lib.check(N.timed("bigru_bwd", x, lambda: lib.dll.kvae_bigru_bwd(
            N.ptr(g_h), N.ptr(gates), N.ptr(h), _ptr2(wi0, wi1), _ptr2(wh0, wh1), N.ptr(dpi), N.ptr(dph), N.ptr(dx),
            Bsz, T, I, H, N.stream_for(x))), "kvae_bigru_bwd")
        if not any(ctx.needs_input_grad[1:]):   # frozen regime posterior: the input gradient only
            return (dx.sum(0),) + (None,) * 8
        x2, h2 = x.reshape(Bsz * T, I), h.reshape(Bsz * T, 2 * H)
        probs = []
        for d in (0, 1):   # h_prev of the forward direction is h_{t-1}, of the reverse direction h_{t+1}
            probs.append(dict(d=dpi[d].reshape(Bsz * T, 3 * H), x=x2))
            probs.append(dict(d=dph[d].reshape(Bsz * T, 3 * H), h=h2[:, d * H:(d + 1) * H], shift=-1 if d == 0 else 1, T=T))
        res = rnn_wgrad(x, probs)
        out = [dx.sum(0)]
        for d in (0, 1):
            (_, g_wih, g_bih), (g_whh, _, g_bhh) = res[2 * d], res[2 * d + 1]
            out += [g_wih, g_whh, g_bih, g_bhh]
        return tuple(out)


# ------------------------------------------------------------------------------------------------
# masked sequences with the LSTM alpha-network: the network runs INSIDE the filter kernel, forward and backward
# ------------------------------------------------------------------------------------------------
ALPHA_LSTM_SUPPORTED = dict(hidden=50, p=2, max_K=16)


def alpha_lstm_supported(Y, lstm, K):
    """Shapes csrc/kvae_lgssm_wide.hip instantiates the in-kernel alpha-network for (KVAEConfig defaults)."""
    return (Y.is_cuda and K > 1 and K <= ALPHA_LSTM_SUPPORTED["max_K"] and lstm.hidden_size == ALPHA_LSTM_SUPPORTED["hidden"]
            and Y.shape[-1] == ALPHA_LSTM_SUPPORTED["p"] and lstm.input_size == Y.shape[-1])


class AlphaLstmSmooth(torch.autograd.Function):
    """Kalman filter (+ RTS smoother) whose per-step alpha comes from an LSTM fed with y_{t-1}, or with C mu_{t|t-1} on
    hidden frames (reference kalman_filter.py:151-185 + dyn_param.py:39-63): ONE launch forward (+ one for the smoother),
    ONE launch backward - the coupled adjoint of filter and cell (kvae_lgssm_alpha_lstm_bwd) - instead of T cell steps
    and T single-step filter launches each way.  Returns (ms, Ss,) mf, Sf, mp, Sp, record [B,T,A|B|C], alpha [B,T,K]."""

    @staticmethod
    def forward(ctx, Y, U, mask, w_ih, w_hh, b_ih, b_hh, head_w, head_b, A, Bm, Cm, Q, R, mu0, Sigma0, with_rts):
        Y, U, mask = _f32c(Y), _f32c(U), _f32c(mask)
        ws_ = [_f32c(t.detach()) for t in (w_ih, w_hh, b_ih, b_hh, head_w, head_b, A, Bm, Cm)]
        Bsz, T, p = Y.shape
        K, n, m = A.shape[0], A.shape[1], Bm.shape[2]
        H = w_hh.shape[1]
        E = n * n + n * m + p * n
        dev = Y.device
        mk = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        mf, Sf, mp, Sp = mk(Bsz, T, n), mk(Bsz, T, n, n), mk(Bsz, T, n), mk(Bsz, T, n, n)
        record, alpha = mk(Bsz, T, E), mk(Bsz, T, K)
        need = any(ctx.needs_input_grad)
        gates, c_seq, h_seq, x_seq = (mk(Bsz, T, 4 * H), mk(Bsz, T, H), mk(Bsz, T, H), mk(Bsz, T, p)) if need else (None,) * 4
        slots = Slots(A=0, B=n * n, C=n * n + n * m)
        call = _Call(Y, U, mask, record, None, None, None, Q, R, mu0, Sigma0, slots)
        ms, Ss = (mk(Bsz, T, n), mk(Bsz, T, n, n)) if with_rts else (None, None)
        st = _states(mf, Sf, mp, Sp, ms, Ss)
        call.lib.check(N.timed("alpha_lstm_fwd", Y, lambda: call.lib.dll.kvae_lgssm_filter_alpha_lstm(
            C.byref(call.prob), C.byref(st), *[N.ptr(w) for w in ws_], K, H, N.ptr(record), N.ptr(alpha), N.ptr(gates),
            N.ptr(c_seq), N.ptr(h_seq), N.ptr(x_seq), call.stream)), "kvae_lgssm_filter_alpha_lstm")
        if with_rts:
            call.lib.check(call.lib.dll.kvae_lgssm_rts_fwd(C.byref(call.prob), C.byref(st), call.stream), "kvae_lgssm_rts_fwd")
        ctx.with_rts, ctx.slots = with_rts, slots
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(Y, U, mask, *ws_, Q, R, mu0, Sigma0, mf, Sf, mp, Sp, ms, Ss, record, alpha, gates, c_seq, h_seq, x_seq)
        return ((ms, Ss) if with_rts else ()) + (mf, Sf, mp, Sp, record, alpha)

    @staticmethod
    def backward(ctx, *gouts):
        (Y, U, mask, w_ih, w_hh, b_ih, b_hh, head_w, head_b, A, Bm, Cm, Q, R, mu0, Sigma0, mf, Sf, mp, Sp, ms, Ss, record, alpha,
         gates, c_seq, h_seq, x_seq) = ctx.saved_tensors
        with_rts, slots = ctx.with_rts, ctx.slots
        if with_rts:
            g_ms, g_Ss, g_mf, g_Sf, g_mp, g_Sp, g_rec, g_alpha = (_f32c(g) for g in gouts)
        else:
            g_mf, g_Sf, g_mp, g_Sp, g_rec, g_alpha = (_f32c(g) for g in gouts)
            g_ms = g_Ss = None
        call = _Call(Y, U, mask, record, None, None, None, Q, R, mu0, Sigma0, slots)
        Bsz, T, n, m, p = call.dims
        K, H = A.shape[0], w_hh.shape[1]
        dev = Y.device
        need = ctx.needs_input_grad
        sink = _GradSink(call, record, None, None, None, Q, slots, False)   # gA|gB|gC land in one g_record buffer
        g0 = S0 = None
        if need[14]:
            g0 = torch.empty(Bsz, n, device=dev, dtype=torch.float32)
            sink.g.g_mu0 = g0.data_ptr()
        if need[15]:
            S0 = torch.empty(Bsz, n, n, device=dev, dtype=torch.float32)
            sink.g.g_Sigma0 = S0.data_ptr()
        ws = torch.empty(Bsz, T, 2 * (n + n * n), device=dev, dtype=torch.float32)
        d_pre = torch.empty(Bsz, T, 4 * H, device=dev, dtype=torch.float32)
        g_logit = torch.empty(Bsz, T, K, device=dev, dtype=torch.float32)
        saved = _states(mf, Sf, mp, Sp, ms, Ss)
        up = _states(g_mf, g_Sf, g_mp, g_Sp, g_ms, g_Ss)
        call.lib.check(N.timed("alpha_lstm_bwd", Y, lambda: call.lib.dll.kvae_lgssm_alpha_lstm_bwd(
            C.byref(call.prob), C.byref(saved), C.byref(up), C.byref(sink.g), N.ptr(ws), int(with_rts), N.ptr(w_ih), N.ptr(w_hh),
            N.ptr(head_w), N.ptr(A), N.ptr(Bm), N.ptr(Cm), K, H, N.ptr(alpha), N.ptr(gates), N.ptr(c_seq), N.ptr(g_rec),
            N.ptr(g_alpha), N.ptr(sink.gpacked), N.ptr(d_pre), N.ptr(g_logit), call.stream)), "kvae_lgssm_alpha_lstm_bwd")
        # parameter gradients: reductions over (b,t) of what the launch wrote (none for a frozen alpha-network)
        g_whh = g_wih = g_b = g_hw = g_hb = None
        if any(need[3:9]):
            h2 = h_seq.reshape(Bsz * T, H)
            (g_whh, g_wih, g_b), (g_hw, _, g_hb) = rnn_wgrad(Y, [
                dict(d=d_pre.reshape(Bsz * T, 4 * H), h=h2, shift=-1, T=T, x=x_seq.reshape(Bsz * T, p)),
                dict(d=g_logit.reshape(Bsz * T, K), h=h2)])
        base = torch.cat([t.reshape(K, -1) for t in (A, Bm, Cm)], dim=1)
        nblk = call.lib.dll.kvae_mix_bwd_partials(Bsz * T)
        E = base.shape[1]
        partials = torch.empty(nblk, K, E, device=dev, dtype=torch.float32)
        g_alpha_scratch, g_base = torch.empty_like(alpha), torch.empty_like(base)
        call.lib.check(call.lib.dll.kvae_mix_bwd(N.ptr(alpha), N.ptr(base), N.ptr(sink.gpacked), N.ptr(g_alpha_scratch), N.ptr(g_base),
                                                 N.ptr(partials), Bsz * T, K, E, 0, call.stream), "kvae_mix_bwd")
        gA, gB, gC = g_base.split([n * n, n * m, p * n], dim=1)
        gA, gB, gC = (g.reshape(t.shape) if nd else None for g, t, nd in zip((gA, gB, gC), (A, Bm, Cm), need[9:12]))
        if g0 is not None and mu0.dim() == 1:
            g0 = N.colsum(g0)
        if S0 is not None and Sigma0.dim() == 2:
            S0 = N.colsum(S0)
        return (sink.gY if need[0] else None, sink.gU if need[1] else None, None, g_wih, g_whh, g_b, g_b, g_hw, g_hb,
                gA, gB, gC, None, None, g0, S0, None)


@torch.no_grad()
def emission_means(mus_smooth, mus_filt, C_view, packed=None, c_off=None):
    """(C_t mu_t|T, C_t mu_t|t) [B,T,p] in one launch (reference model.py:279-288).  C_view: the [B,T,p,n] emission stack the
    caller holds (may be an expanded [p,n] or a slice of the packed step record `packed` at float offset `c_off`)."""
    ms, mf = _f32c(mus_smooth.squeeze(-1)), _f32c(mus_filt.squeeze(-1))
    Bsz, T, n = ms.shape
    p = C_view.shape[-2]
    prob = N.Problem()
    prob.B, prob.T, prob.n, prob.m, prob.p = Bsz, T, n, n, p
    keep, prob.C = _stack(C_view, Bsz, T, p, n, packed, c_off)
    a_s, a_f = torch.empty(Bsz, T, p, device=ms.device, dtype=torch.float32), torch.empty(Bsz, T, p, device=ms.device, dtype=torch.float32)
    lib = N.lib_for(ms)
    lib.check(lib.dll.kvae_lgssm_emission_means(C.byref(prob), N.ptr(ms), N.ptr(mf), N.ptr(a_s), N.ptr(a_f), N.stream_for(ms)),
              "kvae_lgssm_emission_means")
    return a_s, a_f


@torch.no_grad()
def rts_only(Y, U, mask, packed, A, Bm, Cm, Q, R, mu0, Sigma0, slots, mf, Sf, mp, Sp):
    """RTS smoother over already-filtered beliefs (kvae_lgssm_rts_fwd); no autograd."""
    call = _Call(Y, U, mask, packed, A, Bm, Cm, Q, R, mu0, Sigma0, slots)
    ms, Ss = torch.empty_like(mf), torch.empty_like(Sf)
    st = _states(mf, Sf, mp, Sp, ms, Ss)
    call.lib.check(call.lib.dll.kvae_lgssm_rts_fwd(C.byref(call.prob), C.byref(st), call.stream), "kvae_lgssm_rts_fwd")
    return ms, Ss
