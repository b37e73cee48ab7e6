"""kvae — MI355X-native drop-in for the `kvae` package of rodrigo-paganini/kalman-vae.

Same module paths, class names, constructor signatures and state_dict keys as the reference
(kvae.model.model.KVAE, kvae.kalman.kalman_filter.KalmanFilter, kvae.kalman.dyn_param.
DynamicsParameter, kvae.kalman.switch_dyn_param.*, kvae.vae.{vae,losses}, kvae.utils.config),
but the LGSSM hot path (filter, RTS smoother, sampled ELBO, mixture-of-K dynamics; forward and
backward) runs in hand-written HIP kernels for gfx950 (libkvae_lgssm.so) and requires a HIP device.
Put the directory that contains this package (kalman-vae_amd/) on PYTHONPATH in place of the
reference checkout.
"""
__all__ = ["noise"]
