"""Noise injection for parity testing.

The reference draws its randomness from torch's global generator in three places:
  * KVAE.reparameterize          eps_a  ~ N(0,1)  [B*T, a_dim]      (model.py:81-84)
  * KalmanFilter.elbo rsample    eps_z  ~ N(0,1)  [B,T,n]           (kalman_filter.py:351)
  * gumbel_softmax               gumbel ~ Gumbel  [B,T,K]           (switch_dyn_param.py:52,69)
Results on a GPU can only be compared with the CPU oracle if both consume the same draws, so the
drop-in classes look here first: inside `with inject(eps_a=..., eps_z=..., gumbel=...)` the given
tensors are used instead of fresh device-side draws.
"""
import contextlib

_slots = {"eps_a": None, "eps_z": None, "gumbel": None}


def take(name):
    v = _slots.get(name)
    _slots[name] = None if v is None else v  # values stay for the whole context (re-usable)
    return v


@contextlib.contextmanager
def inject(eps_a=None, eps_z=None, gumbel=None):
    old = dict(_slots)
    _slots.update(eps_a=eps_a, eps_z=eps_z, gumbel=gumbel)
    try:
        yield
    finally:
        _slots.update(old)
