"""Conv VAE encoder / decoder of the KVAE, re-declared so that state_dict keys and shapes match the reference
(kvae/vae/vae.py:11-116).  On a HIP device with the reference's default shapes every layer runs on the hand-written
kernels behind kvae.vae.fused (csrc/vae_conv_edge.h, vae_conv_mid.h, vae_conv_up.h, vae_heads.h); any other shape takes
MIOpen for the convolution plus one fused bias/PixelShuffle/ReLU pass; host tensors take plain PyTorch:
    encoder.conv_layers.{0,2,4}, encoder.fc_mu, encoder.fc_var.0, decoder.fc, decoder.deconv_layers.{0,3,6}
"""
import os
from typing import Tuple

import torch
import torch.nn as nn

from kvae import _native
from kvae.utils.config import KVAEConfig
from kvae.vae.fused import DecoderFc, DecoderHead, DecoderUp, EncoderHead, EncoderMid, EncoderStem, conv_block


def _conv_out(size, k, s, p):
    return (size + 2 * p - k) // s + 1


class Encoder(nn.Module):
    """x [N,C,H,W] -> (mu [N,a], var [N,a]) with var = noise_emission * sigmoid(.)."""

    def __init__(self, config: KVAEConfig):
        super().__init__()
        self.config = config
        blocks, c_in, side = [], config.img_channels, config.img_size
        for c_out in config.encoder_channels:
            blocks += [nn.Conv2d(c_in, c_out, config.encoder_kernel_size, config.encoder_stride,
                                 config.encoder_padding), nn.ReLU()]
            side = _conv_out(side, config.encoder_kernel_size, config.encoder_stride, config.encoder_padding)
            c_in = c_out
        self.conv_layers = nn.Sequential(*blocks)
        self.flat_size = c_in * side * side
        self.fc_mu = nn.Linear(self.flat_size, config.a_dim)
        self.fc_var = nn.Sequential(nn.Linear(self.flat_size, config.a_dim), nn.Sigmoid())

    def features(self, x: torch.Tensor) -> torch.Tensor:
        """Flattened output of the convolution stack, [N, flat_size]."""
        if _native.fused_ok(x) and x.dtype == torch.float32:
            h = x   # hand-written convolutions for the reference's shapes; otherwise MIOpen + one fused epilogue pass
            for i, layer in enumerate(self.conv_layers):
                if isinstance(layer, nn.Conv2d):
                    if i == 0 and EncoderStem.supported(h, layer):   # 1 input channel: direct kernel, not a GEMM
                        h = EncoderStem.apply(h, layer.weight, layer.bias)
                    elif EncoderMid.supported(h, layer):             # stride 2: implicit GEMM on the f32 matrix cores
                        h = EncoderMid.apply(h, layer.weight, layer.bias)
                    else:
                        h = conv_block(h, layer, r=1, relu=True)
            return h.flatten(1)
        return self.conv_layers(x).flatten(1)

    def heads(self, feat: torch.Tensor, eps=None):
        """(a, mu, var): both heads and, if eps is given, the reparameterisation a = mu + eps sqrt(var + 1e-6) in one
        kernel (csrc/vae_heads.h); a is mu when eps is None."""
        if _native.fused_ok(feat) and feat.dtype == torch.float32 and EncoderHead.supported(feat, self.fc_mu, self.fc_var[0]):
            return EncoderHead.apply(feat, self.fc_mu.weight, self.fc_mu.bias, self.fc_var[0].weight, self.fc_var[0].bias,
                                     eps, self.config.noise_emission)
        mu, var = self.fc_mu(feat), self.config.noise_emission * self.fc_var(feat)
        return (mu if eps is None else mu + eps * torch.sqrt(var + 1e-6)), mu, var

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        _, mu, var = self.heads(self.features(x))
        return mu, var


class Decoder(nn.Module):
    """a [N,a] -> logits/means [N,C,H,W]: fc -> [c0,s,s] -> (conv3x3 -> PixelShuffle(2) -> ReLU)* -> conv -> PS."""

    def __init__(self, config: KVAEConfig):
        super().__init__()
        self.config = config
        chans = list(config.decoder_channels)
        self.init_size = config.img_size // (2 ** len(chans))
        self.init_channels = chans[0]
        self.fc = nn.Linear(config.a_dim, self.init_channels * self.init_size ** 2)
        up = []
        for c_in, c_out in zip(chans[:-1], chans[1:]):
            up += [nn.Conv2d(c_in, 4 * c_out, kernel_size=3, padding=1), nn.PixelShuffle(2), nn.ReLU()]
        up += [nn.Conv2d(chans[-1], 4 * config.img_channels, kernel_size=3, padding=1), nn.PixelShuffle(2)]
        self.deconv_layers = nn.Sequential(*up)

    def forward(self, a: torch.Tensor) -> torch.Tensor:
        fused = _native.fused_ok(a) and a.dtype == torch.float32
        h = DecoderFc.apply(a, self.fc.weight, self.fc.bias) if fused and DecoderFc.supported(a, self.fc) else self.fc(a)
        h = h.unflatten(1, (self.init_channels, self.init_size, self.init_size))
        if not fused:
            return self.deconv_layers(h)
        layers = list(self.deconv_layers)   # conv -> PixelShuffle(2) [-> ReLU] triples, fused per conv
        for i, layer in enumerate(layers):
            if isinstance(layer, nn.Conv2d):
                relu = i + 2 < len(layers) and isinstance(layers[i + 2], nn.ReLU)
                if not relu and DecoderHead.supported(h, layer):     # 4 output channels: direct kernel
                    h = DecoderHead.apply(h, layer.weight, layer.bias)
                elif relu and DecoderUp.supported(h, layer):         # 32 -> 128 + shuffle + ReLU on the f32 matrix cores
                    h = DecoderUp.apply(h, layer.weight, layer.bias)
                else:
                    h = conv_block(h, layer, r=2, relu=relu)
        return h


class VAE(nn.Module):
    """Frame-wise VAE wrapper (reference kvae/vae/vae.py:119-242): same method names and output keys."""

    def __init__(self, config: KVAEConfig):
        super().__init__()
        self.config = config
        self.encoder = Encoder(config)
        self.decoder = Decoder(config)

    def encode(self, x):
        return self.encoder(x)

    def decode(self, a):
        return self.decoder(a)

    def reparameterize(self, mu, var):
        return mu + torch.randn_like(var) * var.sqrt()

    def forward(self, x: torch.Tensor) -> dict:
        lead = x.shape[:2]
        mu, var = self.encode(x.flatten(0, 1))
        a = self.reparameterize(mu, var)
        x_mu = self.decode(a)
        x_rec = torch.sigmoid(x_mu) if self.config.out_distr.lower() == "bernoulli" else x_mu
        back = lambda t: t.unflatten(0, lead)
        return {"x_recon": back(x_rec), "x_recon_mu": back(x_mu),
                "x_recon_var": torch.tensor(self.config.noise_pixel_var, device=x.device, dtype=x_mu.dtype),
                "a_vae": back(a), "a_mu": back(mu), "a_var": back(var)}

    def sample_from_prior(self, n: int = 1, device=None) -> torch.Tensor:
        device = device or next(self.parameters()).device
        return self.decode(torch.randn(n, self.config.a_dim, device=device))

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, config: KVAEConfig = None, device: str = "cpu"):
        """Accepts a plain state_dict or a {'state_dict': ...} / {'model_state': ...} payload; only
        tensors are unpickled (weights_only=True)."""
        if not os.path.exists(checkpoint_path):
            raise FileNotFoundError(checkpoint_path)
        vae = cls(config or KVAEConfig())
        payload = torch.load(checkpoint_path, map_location=device, weights_only=True)
        for key in ("state_dict", "model_state"):
            if isinstance(payload, dict) and key in payload:
                payload = payload[key]
        for part in ("encoder", "decoder"):
            sub = {k.split(part + ".", 1)[1]: v for k, v in payload.items() if (part + ".") in k}
            if sub:
                getattr(vae, part).load_state_dict(sub)
        return vae.to(device)
