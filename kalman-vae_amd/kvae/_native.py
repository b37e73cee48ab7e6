"""ctypes binding of the C ABI in include/kvae_lgssm.h (libkvae_lgssm.so, gfx950).

There is NO CPU implementation behind this module: if the HIP library has not been built
(`python __graft_entry__.py build`) or no HIP device is visible, the ops raise.  The only way to
run the op layer on host tensors is for a TEST to inject the host simulator of the kernel bodies
(tests/hostsim) through `_set_test_backend`; product code never does that.
"""
import ctypes as C
import os
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libkvae_lgssm.so"

KVAE_MAX_DIM = 16
KVAE_MAX_K = 16
LSTM_MAX_H, LSTM_MAX_I = 52, 16
ABI_VERSION = 10

_STATUS = {1: "KVAE_ERR_DIMS (n, m, p must be in [1,16]; B, T >= 1)", 2: "KVAE_ERR_NULL", 3: "KVAE_ERR_LAUNCH",
           4: "KVAE_ERR_ARG"}


class Stack(C.Structure):  # kvae_stack / kvae_gstack (same layout)
    _fields_ = [("ptr", C.c_void_p), ("sb", C.c_int64), ("st", C.c_int64)]


class Problem(C.Structure):  # kvae_lgssm_problem
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("p", C.c_int32),
                ("A", Stack), ("Bm", Stack), ("C", Stack), ("Q", Stack),
                ("R", C.c_void_p), ("mu0", C.c_void_p), ("mu0_sb", C.c_int64),
                ("Sigma0", C.c_void_p), ("Sigma0_sb", C.c_int64),
                ("Y", C.c_void_p), ("U", C.c_void_p), ("mask", C.c_void_p)]


class States(C.Structure):  # kvae_lgssm_states
    _fields_ = [(k, C.c_void_p) for k in ("mus_filt", "Sigmas_filt", "mus_pred", "Sigmas_pred",
                                          "mus_smooth", "Sigmas_smooth", "aux")]


class WgradProblem(C.Structure):  # kvae_wgrad_problem
    _fields_ = [("d", C.c_void_p), ("h", C.c_void_p), ("x", C.c_void_p), ("g_wh", C.c_void_p), ("g_wx", C.c_void_p),
                ("g_b", C.c_void_p), ("d_stride", C.c_int64), ("h_stride", C.c_int64), ("x_stride", C.c_int64), ("N", C.c_int64),
                ("R", C.c_int32), ("H", C.c_int32), ("I", C.c_int32), ("bias", C.c_int32), ("T", C.c_int32), ("shift", C.c_int32)]


class InputGrads(C.Structure):  # kvae_lgssm_input_grads
    _fields_ = [("gA", Stack), ("gB", Stack), ("gC", Stack), ("gQ", Stack),
                ("gY", C.c_void_p), ("gU", C.c_void_p), ("g_mu0", C.c_void_p), ("g_Sigma0", C.c_void_p)]


SYMBOLS = ("kvae_lgssm_filter_alpha_lstm", "kvae_lgssm_alpha_lstm_bwd", "kvae_lgssm_filter_fwd", "kvae_lgssm_rts_fwd", "kvae_lgssm_smooth_fwd", "kvae_lgssm_smooth_bwd",
           "kvae_lgssm_elbo", "kvae_mix_fwd", "kvae_mix_bwd", "kvae_mix_bwd_partials", "kvae_lstm_fwd",
           "kvae_lstm_bwd", "kvae_bias_shuffle_act_fwd", "kvae_bias_shuffle_act_bwd", "kvae_bias_partial_rows", "kvae_colsum", "kvae_colsum2", "kvae_clip_adam", "kvae_regime_fwd", "kvae_regime_bwd", "kvae_bigru_fwd", "kvae_bigru_bwd", "kvae_bce_frames_fwd", "kvae_bce_frames_bwd",
           "kvae_dec_head_fwd", "kvae_dec_head_bwd", "kvae_enc_stem_fwd", "kvae_enc_stem_bwd", "kvae_conv_edge_partial_rows",
           "kvae_enc_mid_fwd", "kvae_enc_mid_bwd", "kvae_enc_mid_partial_rows",
           "kvae_dec_up_fwd", "kvae_dec_up_bwd", "kvae_dec_up_partial_rows", "kvae_dec_up_set_workgroups",
           "kvae_enc_head_fwd", "kvae_enc_head_bwd", "kvae_dec_fc_fwd", "kvae_dec_fc_bwd", "kvae_head_partial_rows",
           "kvae_latent_reg_fwd", "kvae_latent_reg_bwd", "kvae_loss_head_fwd", "kvae_loss_head_bwd",
           "kvae_lgssm_emission_means", "kvae_rnn_wgrad", "kvae_rnn_wgrad_ws_floats", "kvae_linear_fwd", "kvae_linear_bwd_input",
           "kvae_abi_version",
           "kvae_last_error", "kvae_build_info")


class LgssmLib:
    """One loaded implementation of the C ABI."""

    def __init__(self, path):
        self.path = str(path)
        self.dll = C.CDLL(self.path)
        missing = [s for s in SYMBOLS if not hasattr(self.dll, s)]
        if missing:
            raise OSError(f"{self.path} does not export {missing}")
        d = self.dll
        P, S, G, vp = C.POINTER(Problem), C.POINTER(States), C.POINTER(InputGrads), C.c_void_p
        for name in ("kvae_lgssm_filter_fwd", "kvae_lgssm_rts_fwd", "kvae_lgssm_smooth_fwd"):
            getattr(d, name).argtypes = [P, S, vp]
            getattr(d, name).restype = C.c_int
        d.kvae_lgssm_filter_alpha_lstm.argtypes = [P, S] + [vp] * 9 + [C.c_int32, C.c_int32] + [vp] * 7
        d.kvae_lgssm_filter_alpha_lstm.restype = C.c_int
        d.kvae_lgssm_alpha_lstm_bwd.argtypes = [P, S, S, G, vp, C.c_int] + [vp] * 6 + [C.c_int32, C.c_int32] + [vp] * 9
        d.kvae_lgssm_alpha_lstm_bwd.restype = C.c_int
        d.kvae_lgssm_smooth_bwd.argtypes = [P, S, S, G, vp, C.c_int, vp]
        d.kvae_lgssm_smooth_bwd.restype = C.c_int
        d.kvae_lgssm_elbo.argtypes = [P, vp, vp, vp, vp, vp, vp, vp, vp, G, vp]
        d.kvae_lgssm_elbo.restype = C.c_int
        d.kvae_mix_fwd.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, vp]
        d.kvae_mix_fwd.restype = C.c_int
        d.kvae_mix_bwd.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp]
        d.kvae_mix_bwd.restype = C.c_int
        d.kvae_mix_bwd_partials.argtypes = [C.c_int64]
        d.kvae_mix_bwd_partials.restype = C.c_int64
        d.kvae_lstm_fwd.argtypes = [vp] * 8 + [C.c_int32] * 4 + [vp]
        d.kvae_lstm_fwd.restype = C.c_int
        d.kvae_lstm_bwd.argtypes = [vp] * 7 + [C.c_int32] * 4 + [vp]
        d.kvae_lstm_bwd.restype = C.c_int
        d.kvae_bias_shuffle_act_fwd.argtypes = [vp, vp, vp, C.c_int64] + [C.c_int32] * 5 + [vp]
        d.kvae_bias_shuffle_act_fwd.restype = C.c_int
        d.kvae_bias_shuffle_act_bwd.argtypes = [vp, vp, vp, vp, C.c_int64] + [C.c_int32] * 5 + [vp]
        d.kvae_bias_shuffle_act_bwd.restype = C.c_int
        d.kvae_regime_fwd.argtypes = [vp] * 7 + [C.c_int32] * 3 + [C.c_float, vp, C.c_int32, vp]
        d.kvae_regime_fwd.restype = C.c_int
        d.kvae_regime_bwd.argtypes = [vp] * 10 + [C.c_int32] * 3 + [C.c_float, vp, vp]
        d.kvae_regime_bwd.restype = C.c_int
        d.kvae_bigru_fwd.argtypes = [vp] * 7 + [C.c_int32] * 4 + [vp]
        d.kvae_bigru_fwd.restype = C.c_int
        d.kvae_bigru_bwd.argtypes = [vp] * 8 + [C.c_int32] * 4 + [vp]
        d.kvae_bigru_bwd.restype = C.c_int
        d.kvae_bce_frames_fwd.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, vp]
        d.kvae_bce_frames_fwd.restype = C.c_int
        d.kvae_bce_frames_bwd.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int32, vp]
        d.kvae_bce_frames_bwd.restype = C.c_int
        d.kvae_dec_head_fwd.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, vp]
        d.kvae_dec_head_fwd.restype = C.c_int
        d.kvae_dec_head_bwd.argtypes = [vp] * 7 + [C.c_int64, C.c_int32, C.c_int32, vp]
        d.kvae_dec_head_bwd.restype = C.c_int
        d.kvae_enc_stem_fwd.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, vp]
        d.kvae_enc_stem_fwd.restype = C.c_int
        d.kvae_enc_stem_bwd.argtypes = [vp] * 6 + [C.c_int64, C.c_int32, C.c_int32, vp]
        d.kvae_enc_stem_bwd.restype = C.c_int
        d.kvae_enc_mid_fwd.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, vp]
        d.kvae_enc_mid_fwd.restype = C.c_int
        d.kvae_enc_mid_bwd.argtypes = [vp] * 7 + [C.c_int64, C.c_int32, C.c_int32, vp]
        d.kvae_enc_mid_bwd.restype = C.c_int
        d.kvae_enc_mid_partial_rows.argtypes = [C.c_int64, C.c_int32]
        d.kvae_dec_up_fwd.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, vp]
        d.kvae_dec_up_fwd.restype = C.c_int
        d.kvae_dec_up_bwd.argtypes = [vp] * 7 + [C.c_int64, C.c_int32, C.c_int32, vp]
        d.kvae_dec_up_bwd.restype = C.c_int
        d.kvae_dec_up_partial_rows.argtypes = [C.c_int64, C.c_int32]
        d.kvae_enc_head_fwd.argtypes = [vp] * 9 + [C.c_int64, C.c_int32, C.c_int32, C.c_float, vp]
        d.kvae_enc_head_fwd.restype = C.c_int
        d.kvae_enc_head_bwd.argtypes = [vp] * 11 + [C.c_int64, C.c_int32, C.c_int32, C.c_float, vp]
        d.kvae_enc_head_bwd.restype = C.c_int
        d.kvae_dec_fc_fwd.argtypes = [vp] * 4 + [C.c_int64, C.c_int32, C.c_int32, vp]
        d.kvae_dec_fc_fwd.restype = C.c_int
        d.kvae_dec_fc_bwd.argtypes = [vp] * 6 + [C.c_int64, C.c_int32, C.c_int32, vp]
        d.kvae_dec_fc_bwd.restype = C.c_int
        d.kvae_head_partial_rows.argtypes = []
        d.kvae_head_partial_rows.restype = C.c_int64
        d.kvae_latent_reg_fwd.argtypes = [vp] * 4 + [C.c_int64, C.c_int32, vp]
        d.kvae_latent_reg_fwd.restype = C.c_int
        d.kvae_latent_reg_bwd.argtypes = [vp] * 7 + [C.c_int64, C.c_int32, vp]
        d.kvae_latent_reg_bwd.restype = C.c_int
        d.kvae_loss_head_fwd.argtypes = [vp] * 5 + [C.c_float] * 3 + [vp, vp, vp, C.c_int64, vp]
        d.kvae_loss_head_fwd.restype = C.c_int
        d.kvae_loss_head_bwd.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, vp]
        d.kvae_loss_head_bwd.restype = C.c_int
        d.kvae_dec_up_partial_rows.restype = C.c_int64
        d.kvae_dec_up_set_workgroups.argtypes = [C.c_int32]
        d.kvae_dec_up_set_workgroups.restype = C.c_int32
        d.kvae_enc_mid_partial_rows.restype = C.c_int64
        d.kvae_conv_edge_partial_rows.argtypes = [C.c_int64]
        d.kvae_conv_edge_partial_rows.restype = C.c_int64
        d.kvae_colsum.argtypes = [vp, vp, C.c_int64, C.c_int64, vp]
        d.kvae_colsum.restype = C.c_int
        d.kvae_colsum2.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, vp, C.c_int64, C.c_int64, vp]
        d.kvae_colsum2.restype = C.c_int
        d.kvae_lgssm_emission_means.argtypes = [P, vp, vp, vp, vp, vp]
        d.kvae_lgssm_emission_means.restype = C.c_int
        d.kvae_rnn_wgrad_ws_floats.argtypes = [C.POINTER(WgradProblem), C.c_int32]
        d.kvae_rnn_wgrad_ws_floats.restype = C.c_int64
        d.kvae_rnn_wgrad.argtypes = [C.POINTER(WgradProblem), C.c_int32, vp, vp]
        d.kvae_rnn_wgrad.restype = C.c_int
        d.kvae_linear_fwd.argtypes = [vp, C.c_int64, C.c_int64, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp, vp]
        d.kvae_linear_fwd.restype = C.c_int
        d.kvae_linear_bwd_input.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, C.c_int32, vp, vp, C.c_int64, vp]
        d.kvae_linear_bwd_input.restype = C.c_int
        d.kvae_clip_adam.argtypes = [vp, vp, vp, vp, C.c_int64, vp, C.c_int32, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_float, vp, vp, vp, vp]
        d.kvae_clip_adam.restype = C.c_int
        d.kvae_bias_partial_rows.argtypes = [C.c_int64]
        d.kvae_bias_partial_rows.restype = C.c_int64
        d.kvae_abi_version.restype = C.c_int
        d.kvae_last_error.restype = C.c_char_p
        d.kvae_build_info.restype = C.c_char_p
        if d.kvae_abi_version() != ABI_VERSION:
            raise OSError(f"{self.path}: ABI {d.kvae_abi_version()} != expected {ABI_VERSION}")

    def check(self, rc, what):
        if rc != 0:
            detail = self.dll.kvae_last_error().decode() if rc == 3 else ""
            raise RuntimeError(f"{what} failed: {_STATUS.get(rc, rc)} {detail}")

    @property
    def build_info(self):
        return self.dll.kvae_build_info().decode()


_hip_lib = None
_test_backend = None


def hip_lib():
    """The gfx950 library; raises (never falls back) if it is not built."""
    global _hip_lib
    if _hip_lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"HIP library {LIB_PATH} is missing. Build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the LGSSM path.")
        _hip_lib = LgssmLib(LIB_PATH)
    return _hip_lib


def _set_test_backend(lib):
    """TEST-ONLY: route ops on host tensors to the host simulator of the kernel bodies."""
    global _test_backend
    _test_backend = lib


def fused_ok(t: torch.Tensor) -> bool:
    """True when the HIP kernels can take this tensor (HIP device, or a test injected the host simulator)."""
    return t.is_cuda or _test_backend is not None


def lib_for(t: torch.Tensor):
    if t.is_cuda:
        return hip_lib()
    if _test_backend is not None:
        return _test_backend
    raise RuntimeError(
        "kvae LGSSM ops run only on a HIP device (MI355X): got a CPU tensor and there is no CPU fallback. "
        "Move the model and inputs to 'cuda'.")


def stream_for(t: torch.Tensor):
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ---- optional per-call HIP-event timing (bench.py's roofline leg) -----------------------------------
_prof = None


def profile_start():
    global _prof
    _prof = {}


def profile_stop():
    """{call name: [milliseconds per call]} measured with HIP events on the launch stream."""
    global _prof
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    return {k: [s.elapsed_time(e) for s, e in v] for k, v in (rec or {}).items()}


def timed(name, t, thunk):
    if _prof is None or not t.is_cuda:
        return thunk()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = thunk()
    e.record()
    _prof.setdefault(name, []).append((s, e))
    return rc


def _colsum_raw(p2):
    import torch
    out = torch.empty(p2.shape[1], device=p2.device, dtype=torch.float32)
    lib = lib_for(p2)
    lib.check(lib.dll.kvae_colsum(ptr(p2), ptr(out), p2.shape[0], p2.shape[1], stream_for(p2)), "kvae_colsum")
    return out


def colsum_pair(a, b):
    """(colsum(a), colsum(b)) in one launch when neither needs the two-pass folding of tall inputs (k_colsum_v4_pair): the
    weight- and bias-gradient partials of one layer."""
    import torch
    a2, b2 = a.reshape(a.shape[0], -1).contiguous(), b.reshape(b.shape[0], -1).contiguous()
    tall = [p.shape[0] >= 1024 and p.shape[1] < 4096 for p in (a2, b2)]
    if a2.device != b2.device or tall[0] != tall[1] or (tall[0] and (a2.shape[0] != b2.shape[0] or a2.shape[0] % 64)):
        return colsum(a), colsum(b)

    def pair(x, y):
        ox = torch.empty(x.shape[1], device=x.device, dtype=torch.float32)
        oy = torch.empty(y.shape[1], device=y.device, dtype=torch.float32)
        lib = lib_for(x)
        lib.check(lib.dll.kvae_colsum2(ptr(x), ptr(ox), x.shape[0], x.shape[1], ptr(y), ptr(oy), y.shape[0], y.shape[1],
                                       stream_for(x)), "kvae_colsum2")
        return ox, oy

    if tall[0]:   # the two-pass folding of colsum(), both tensors per launch
        rows = a2.shape[0]
        fa, fb = pair(a2.view(64, (rows // 64) * a2.shape[1]), b2.view(64, (rows // 64) * b2.shape[1]))
        a2, b2 = fa.view(rows // 64, a2.shape[1]), fb.view(rows // 64, b2.shape[1])
    oa, ob = pair(a2, b2)
    return oa.view(a.shape[1:]), ob.view(b.shape[1:])


def colsum(partials):
    """out[c] = sum_r partials[r, c] (k_colsum): the second stage of every partial-row reduction, and the bias
    gradients of the recurrent networks.  Tall inputs ([B*T, cols] with few columns) are folded in two passes -
    [R1, R2*cols] then [R2, cols] - so that the kernel always sees many columns to spread over the chip."""
    p2 = partials.reshape(partials.shape[0], -1).contiguous()
    rows, cols = p2.shape
    if rows >= 1024 and cols < 4096:
        r1 = 64
        while r1 > 1 and rows % r1:
            r1 //= 2
        if r1 > 1:
            p2 = _colsum_raw(p2.view(r1, (rows // r1) * cols)).view(rows // r1, cols)
    return _colsum_raw(p2).view(partials.shape[1:])
