"""Model configuration of the Kalman-VAE.

Field names and defaults are the public contract of the reference's `KVAEConfig`
(kvae/utils/config.py:4-60 there): YAML files written for the reference (`kvae:` section of
kvae/train/config.yaml) load unchanged through `KVAEConfig(**cfg)` or `KVAEConfig.from_dict`.
"""
from dataclasses import dataclass, fields
from typing import List, Optional

_DYNAMICS = ("switching", "lstm")
_OUT = ("bernoulli", "gaussian")


@dataclass
class KVAEConfig:
    # ---- frames ----
    img_channels: int = 1
    img_size: int = 32

    # ---- latent sizes: a_t (VAE code, LGSSM observation), z_t (LGSSM state), u_t (control) ----
    a_dim: int = 2
    z_dim: int = 4
    u_dim: Optional[int] = None          # defaults to z_dim

    # ---- mixture of K linear dynamics ----
    num_modes: int = 3
    sticky_p_stay: float = 0.8           # diagonal of the sticky Markov prior ("switching")
    tau_init: float = 1.0                # Gumbel-softmax temperature schedule ("switching")
    tau_decay_rate: float = 0.995
    tau_decay_steps: int = 1
    tau_min: float = 0.2
    dynamics_model: str = "switching"    # "switching" (bi-GRU regime posterior) | "lstm" (alpha-net)
    noise_emission: float = 0.03         # R = noise_emission * I   (a variance)
    noise_transition: float = 0.02       # Q = noise_transition * I (a variance)
    init_cov: float = 20.0               # Sigma0 = init_cov * I
    init_kf_matrices: float = 0.05       # std of the B, C initialisation

    # ---- conv VAE ----
    out_distr: str = "bernoulli"
    encoder_channels: Optional[List[int]] = None   # defaults to [32, 32, 32]
    encoder_kernel_size: int = 3
    encoder_stride: int = 2
    encoder_padding: int = 1
    decoder_channels: Optional[List[int]] = None   # defaults to [32, 32, 32]
    decoder_kernel_size: int = 3
    decoder_stride: int = 2
    decoder_padding: int = 1
    noise_pixel_var: float = 0.1
    scale_reconstruction: float = 0.3

    # ---- linear beta schedule on the KL term ----
    scheduled_beta: bool = True
    start_epoch: int = 0
    end_epoch: int = 5
    start_val: float = 0.0
    end_val: float = 1.0

    # ---- dynamics parameter network / imputation masks ----
    dynamics_hidden_dim: int = 50
    t_init_mask: int = 4
    t_steps_mask: int = 12

    def __post_init__(self):
        if self.u_dim is None:
            self.u_dim = self.z_dim
        for name in ("encoder_channels", "decoder_channels"):
            if getattr(self, name) is None:
                setattr(self, name, [32, 32, 32])

    # -- additions over the reference (do not change defaults or field set) --
    @classmethod
    def from_dict(cls, cfg: dict) -> "KVAEConfig":
        known = {f.name for f in fields(cls)}
        unknown = sorted(set(cfg) - known)
        if unknown:
            raise TypeError(f"unknown KVAEConfig fields: {unknown}")
        return cls(**cfg)

    def validate(self) -> "KVAEConfig":
        if self.dynamics_model.lower() not in _DYNAMICS:
            raise ValueError(f"Unknown dynamics model: {self.dynamics_model}")
        if self.out_distr.lower() not in _OUT:
            raise ValueError(f"Unknown output distribution: {self.out_distr}")
        for name in ("a_dim", "z_dim", "u_dim"):
            v = getattr(self, name)
            if not 1 <= v <= 16:
                raise ValueError(f"{name}={v}: the HIP LGSSM kernels support dimensions 1..16")
        if not 1 <= self.num_modes <= 16:
            raise ValueError("num_modes must be in 1..16")
        return self

    # -- convenience (not in the reference) --------------------------------------------------------------------
    def to_dict(self) -> dict:
        return {f.name: getattr(self, f.name) for f in fields(self)}

    @classmethod
    def from_yaml(cls, path, section: str = "kvae") -> "KVAEConfig":
        """Read the `kvae:` section of a reference-style YAML config (kvae/train/config.yaml there)."""
        import yaml
        with open(path) as fh:
            doc = yaml.safe_load(fh) or {}
        return cls.from_dict(doc.get(section, doc) or {})

    def lgssm_dims(self):
        """(n, m, p) = (dim z, dim u, dim a) as the HIP kernels name them."""
        return self.z_dim, self.u_dim, self.a_dim

    def fast_path(self) -> dict:
        """Which hand-specialised kernels this configuration hits (everything else runs the generic instantiations)."""
        n, m, p = self.lgssm_dims()
        lstm, sw = self.dynamics_model.lower() == "lstm", self.dynamics_model.lower() == "switching"
        rnn_ok = (self.dynamics_hidden_dim, p) == (50, 2) or self.num_modes == 1
        return {   # True = on the specialised kernel (or not applicable to this configuration)
            "lgssm_specialised": (n, m, p) in ((4, 4, 2), (16, 16, 2)),
            "lstm_registers": (not lstm) or rnn_ok,
            "bigru_registers": (not sw) or rnn_ok,
            "vae_default_shapes": (self.img_size, self.img_channels, self.a_dim, list(self.encoder_channels),
                                   list(self.decoder_channels)) == (32, 1, 2, [32, 32, 32], [32, 32, 32]),
        }

    def describe(self) -> str:
        n, m, p = self.lgssm_dims()
        fp = ", ".join(k for k, v in self.fast_path().items() if v) or "generic kernels"
        return (f"KVAE[{self.dynamics_model}] K={self.num_modes} z={n} u={m} a={p} "
                f"frames {self.img_channels}x{self.img_size}x{self.img_size} out={self.out_distr} | {fp}")
