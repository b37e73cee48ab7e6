"""ctypes wrapper of oracle/lgssm_oracle.c (CPU oracle, test infrastructure only)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
LIB = HERE / "_build" / "liblgssm_oracle.so"


class _Stack(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("sb", C.c_int64), ("st", C.c_int64)]


def _lib():
    if not LIB.exists() or LIB.stat().st_mtime < (HERE / "lgssm_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-s", "-C", str(HERE)], check=True)
    lib = C.CDLL(str(LIB))
    vp, i = C.c_void_p, C.c_int
    lib.kvae_oracle_smooth.argtypes = [i] * 5 + [vp] * 3 + [_Stack] * 4 + [vp] * 9
    lib.kvae_oracle_smooth.restype = i
    lib.kvae_oracle_elbo.argtypes = [i] * 5 + [vp] * 6 + [_Stack] * 4 + [vp] * 5
    lib.kvae_oracle_elbo.restype = i
    return lib


def _stack(t, B, T, r, c):
    t = t.detach().float()
    if t.dim() == 2:
        t = t.contiguous()
        return t, _Stack(t.data_ptr(), 0, 0)
    t = t.expand(B, T, r, c).contiguous()
    return t, _Stack(t.data_ptr(), T * r * c, r * c)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def smooth(Y, U, mask, A, Bm, Cm, Q, R, mu0, Sigma0, with_rts=True):
    """Returns dict of the six stacks ([B,T,n] means, [B,T,n,n] covariances)."""
    lib = _lib()
    Y, U = Y.detach().float().contiguous(), U.detach().float().contiguous()
    B, T, p = Y.shape
    m, n = U.shape[-1], Sigma0.shape[-1]
    keep = [_stack(A, B, T, n, n), _stack(Bm, B, T, n, m), _stack(Cm, B, T, p, n), _stack(Q, B, T, n, n)]
    mask_c = mask.detach().float().contiguous() if mask is not None else None
    R, mu0, Sigma0 = (t.detach().float().contiguous() for t in (R, mu0, Sigma0))
    z = lambda *s: torch.zeros(*s)
    out = dict(mus_filt=z(B, T, n), Sigmas_filt=z(B, T, n, n), mus_pred=z(B, T, n), Sigmas_pred=z(B, T, n, n))
    if with_rts:
        out.update(mus_smooth=z(B, T, n), Sigmas_smooth=z(B, T, n, n))
    rc = lib.kvae_oracle_smooth(B, T, n, m, p, _p(Y), _p(U), _p(mask_c), keep[0][1], keep[1][1], keep[2][1], keep[3][1],
                                _p(R), _p(mu0), _p(Sigma0), _p(out["mus_filt"]), _p(out["Sigmas_filt"]),
                                _p(out["mus_pred"]), _p(out["Sigmas_pred"]), _p(out.get("mus_smooth")),
                                _p(out.get("Sigmas_smooth")))
    if rc:
        raise RuntimeError("oracle: singular matrix in solve")
    return out


def elbo_terms(mus, Sigs, eps, Y, U, mask, A, Bm, Cm, Q, R, mu0, Sigma0):
    """(terms[4] float64 numpy: transition, emission, init, entropy; levels[2])."""
    lib = _lib()
    Y, U = Y.detach().float().contiguous(), U.detach().float().contiguous()
    B, T, p = Y.shape
    m, n = U.shape[-1], Sigma0.shape[-1]
    keep = [_stack(A, B, T, n, n), _stack(Bm, B, T, n, m), _stack(Cm, B, T, p, n), _stack(Q, B, T, n, n)]
    mus = mus.detach().float().reshape(B, T, n).contiguous()
    Sigs, eps = Sigs.detach().float().contiguous(), eps.detach().float().contiguous()
    mask_c = mask.detach().float().contiguous() if mask is not None else None
    R, mu0, Sigma0 = (t.detach().float().contiguous() for t in (R, mu0, Sigma0))
    terms = np.zeros(4, np.float64)
    levels = np.zeros(2, np.int32)
    rc = lib.kvae_oracle_elbo(B, T, n, m, p, _p(mus), _p(Sigs), _p(eps), _p(Y), _p(U), _p(mask_c), keep[0][1], keep[1][1],
                              keep[2][1], keep[3][1], _p(R), _p(mu0), _p(Sigma0),
                              terms.ctypes.data_as(C.c_void_p), levels.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("oracle: cholesky(R) or cholesky(Sigma0) failed")
    return terms, levels
