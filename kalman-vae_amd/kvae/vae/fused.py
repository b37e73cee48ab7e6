"""Fused conv epilogues of the frame VAE (HIP kernels k_vae_epilogue_fwd/bwd, csrc/vae_epilogue.h).

The convolutions themselves stay on MIOpen; what PyTorch would run after each of them as separate
full-tensor passes (bias add, nn.PixelShuffle, nn.ReLU — reference kvae/vae/vae.py:20-31, 92-101) is one
pass forward and one pass backward here."""
import torch

from .. import _native as N


class BiasShuffleAct(torch.autograd.Function):
    """out[N,C,H*r,W*r] = act(pixel_shuffle_r(x[N,C*r*r,H,W] + bias))."""

    @staticmethod
    def forward(ctx, x, bias, r, relu):
        x = x.contiguous()
        bias = bias.contiguous()
        Nb, Crr, H, W = x.shape
        C = Crr // (r * r)
        out = torch.empty(Nb, C, H * r, W * r, device=x.device, dtype=torch.float32)
        lib = N.lib_for(x)
        lib.check(lib.dll.kvae_bias_shuffle_act_fwd(N.ptr(x), N.ptr(bias), N.ptr(out), Nb, C, H, W, r, int(relu),
                                                    N.stream_for(x)), "kvae_bias_shuffle_act_fwd")
        ctx.r, ctx.relu, ctx.shape = r, relu, (Nb, C, H, W)
        ctx.save_for_backward(out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        Nb, C, H, W = ctx.shape
        g = g.contiguous()
        g_in = torch.empty(Nb, C * ctx.r * ctx.r, H, W, device=g.device, dtype=torch.float32)
        lib = N.lib_for(g)
        partials = torch.empty(lib.dll.kvae_bias_partial_rows(Nb), C * ctx.r * ctx.r, device=g.device, dtype=torch.float32)
        lib.check(lib.dll.kvae_bias_shuffle_act_bwd(N.ptr(g), N.ptr(out), N.ptr(g_in), N.ptr(partials), Nb, C, H, W, ctx.r,
                                                    int(ctx.relu), N.stream_for(g)), "kvae_bias_shuffle_act_bwd")
        return g_in, partials.sum(0), None, None


def conv_block(x, conv, r=1, relu=True):
    """conv (MIOpen, no bias) -> fused bias + PixelShuffle(r) + optional ReLU."""
    y = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    return BiasShuffleAct.apply(y, conv.bias, r, relu)


class BernoulliFrameLogLik(torch.autograd.Function):
    """[B,T] log p(x_t | a_t) = -sum_pixels BCEWithLogits(x_logits, x) (reference kvae/vae/losses.py:85-87) in one pass."""

    @staticmethod
    def forward(ctx, x_logits, x):
        lg, xx = x_logits.contiguous(), x.contiguous()
        lead = lg.shape[:2]
        frames = lead[0] * lead[1]
        pixels = lg.numel() // frames
        out = torch.empty(lead, device=lg.device, dtype=torch.float32)
        lib = N.lib_for(lg)
        lib.check(lib.dll.kvae_bce_frames_fwd(N.ptr(lg), N.ptr(xx), N.ptr(out), frames, pixels, N.stream_for(lg)),
                  "kvae_bce_frames_fwd")
        ctx.save_for_backward(lg, xx)
        return out

    @staticmethod
    def backward(ctx, g):
        lg, xx = ctx.saved_tensors
        g = g.contiguous()
        frames = g.numel()
        g_logits = torch.empty_like(lg)
        lib = N.lib_for(lg)
        lib.check(lib.dll.kvae_bce_frames_bwd(N.ptr(lg), N.ptr(xx), N.ptr(g), N.ptr(g_logits), frames, lg.numel() // frames,
                                              N.stream_for(lg)), "kvae_bce_frames_bwd")
        return g_logits, None
