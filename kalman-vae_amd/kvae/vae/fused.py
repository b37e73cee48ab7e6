"""HIP-backed pieces of the frame VAE (autograd Functions over the C ABI in include/kvae_lgssm.h).

  * EncoderStem / EncoderMid / DecoderUp / DecoderHead - the convolutions of the reference's default encoder and decoder
    (kvae/vae/vae.py:20-31, 92-104) with bias, ReLU and PixelShuffle fused (csrc/vae_conv_edge.h, vae_conv_mid.h, vae_conv_up.h);
  * EncoderHead / DecoderFc / LatentReg / BernoulliFrameLogLik - the fully-connected ends, the reparameterisation and the
    two loss terms (csrc/vae_heads.h, vae_loss.h);
  * BiasShuffleAct / conv_block - for any OTHER convolution shape: the library convolution followed by ONE fused
    bias + PixelShuffle + ReLU pass instead of three (csrc/vae_epilogue.h)."""
import torch

from .. import _native as N


colsum = N.colsum
colsum_pair = N.colsum_pair


def _chunks(n, size):
    return [(a, min(a + size, n)) for a in range(0, n, size)]


def _acc(total, part):
    return part if total is None else total + part


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
        return g_in, colsum(partials), None, None


def conv_block(x, conv, r=1, relu=True):
    """conv (MIOpen, no bias) -> fused bias + PixelShuffle(r) + optional ReLU."""
    y = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    return BiasShuffleAct.apply(y, conv.bias, r, relu)


class DecoderHead(torch.autograd.Function):
    """logits[N,1,2s,2s] = pixel_shuffle_2(conv3x3(h[N,32,s,s], W[4,32,3,3]) + b) as ONE direct-convolution kernel
    (csrc/vae_conv_edge.h; reference kvae/vae/vae.py:103-104).  A 4-output-channel implicit GEMM has nothing to
    tile over; the direct form is bound by reading h once."""

    SCRATCH = 2304   # KVAE_DEC_HEAD_SCRATCH_FLOATS

    @staticmethod
    def supported(h, conv):
        return (h.dim() == 4 and tuple(h.shape[1:]) == (32, 16, 16) and tuple(conv.weight.shape) == (4, 32, 3, 3)
                and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
                and conv.bias is not None)

    @staticmethod
    def forward(ctx, h, weight, bias):
        h, weight, bias = h.contiguous(), weight.contiguous(), bias.contiguous()
        Nb, Cin, s, _ = h.shape
        logits = torch.empty(Nb, 1, 2 * s, 2 * s, device=h.device, dtype=torch.float32)
        scratch = torch.empty(DecoderHead.SCRATCH, device=h.device, dtype=torch.float32)
        lib = N.lib_for(h)
        lib.check(lib.dll.kvae_dec_head_fwd(N.ptr(h), N.ptr(weight), N.ptr(bias), N.ptr(logits), N.ptr(scratch), Nb, Cin, s,
                                            N.stream_for(h)), "kvae_dec_head_fwd")
        ctx.save_for_backward(h, weight)
        return logits

    @staticmethod
    def backward(ctx, g):
        h, weight = ctx.saved_tensors
        g = g.contiguous()
        Nb, Cin, s, _ = h.shape
        lib = N.lib_for(h)
        rows = lib.dll.kvae_conv_edge_partial_rows(Nb)
        g_h = torch.empty_like(h) if ctx.needs_input_grad[0] else None
        wp = torch.empty(rows, weight.numel(), device=h.device, dtype=torch.float32)
        bp = torch.empty(rows, 4, device=h.device, dtype=torch.float32)
        scratch = torch.empty(DecoderHead.SCRATCH, device=h.device, dtype=torch.float32)
        lib.check(lib.dll.kvae_dec_head_bwd(N.ptr(h), N.ptr(weight), N.ptr(g), N.ptr(g_h) if g_h is not None else None,
                                            N.ptr(wp), N.ptr(bp), N.ptr(scratch), Nb, Cin, s, N.stream_for(h)),
                  "kvae_dec_head_bwd")
        gw, gb = colsum_pair(wp, bp)
        return g_h, gw.view_as(weight), gb


class EncoderStem(torch.autograd.Function):
    """out[N,32,s/2,s/2] = relu(conv3x3_stride2(x[N,1,s,s], W[32,1,3,3]) + b) as one direct kernel, and its weight /
    bias gradient with the ReLU mask fused (reference kvae/vae/vae.py:20-31).  The frames take no gradient."""

    @staticmethod
    def supported(x, conv):
        return (x.dim() == 4 and tuple(x.shape[1:]) == (1, 32, 32) and tuple(conv.weight.shape) == (32, 1, 3, 3)
                and conv.stride == (2, 2) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
                and conv.bias is not None and not x.requires_grad)

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
        Nb, _, s, _ = x.shape
        Cout = weight.shape[0]
        out = torch.empty(Nb, Cout, s // 2, s // 2, device=x.device, dtype=torch.float32)
        lib = N.lib_for(x)
        # one word of ReLU sign bits per output pixel (32 channels): the weight gradient reads these instead of `out`
        bits = torch.empty(Nb, (s // 2) ** 2, device=x.device, dtype=torch.int32) if (x.is_cuda and Cout == 32) else None
        lib.check(lib.dll.kvae_enc_stem_fwd(N.ptr(x), N.ptr(weight), N.ptr(bias), N.ptr(out), N.ptr(bits) if bits is not None else None,
                                            Nb, Cout, s, N.stream_for(x)), "kvae_enc_stem_fwd")
        ctx.save_for_backward(x, out, bits)
        ctx.wshape = weight.shape
        return out

    @staticmethod
    def backward(ctx, g):
        x, out, bits = ctx.saved_tensors
        g = g.contiguous()
        Nb, Cout = out.shape[:2]
        lib = N.lib_for(x)
        rows = lib.dll.kvae_conv_edge_partial_rows(Nb)
        wp = torch.empty(rows, Cout * 9, device=x.device, dtype=torch.float32)
        bp = torch.empty(rows, Cout, device=x.device, dtype=torch.float32)
        lib.check(lib.dll.kvae_enc_stem_bwd(N.ptr(x), N.ptr(out), N.ptr(bits) if bits is not None else None, N.ptr(g), N.ptr(wp),
                                            N.ptr(bp), Nb, Cout, x.shape[2], N.stream_for(x)), "kvae_enc_stem_bwd")
        gw, gb = colsum_pair(wp, bp)
        return None, gw.view(ctx.wshape), gb


class EncoderMid(torch.autograd.Function):
    """out[N,32,s/2,s/2] = relu(conv3x3_stride2(x[N,32,s,s], W[32,32,3,3]) + b), s in {16, 8}, as implicit GEMMs on the
    exact-f32 matrix cores (csrc/vae_conv_mid.h; reference kvae/vae/vae.py:20-31).  Backward fuses the ReLU mask."""

    @staticmethod
    def supported(x, conv):
        return (x.dim() == 4 and x.shape[1] == 32 and x.shape[2] == x.shape[3] and x.shape[2] in (16, 8)
                and tuple(conv.weight.shape) == (32, 32, 3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1)
                and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is not None)

    CHUNK = 16384   # frames per launch: a launch addresses its tensors with 32-bit byte offsets (C ABI: KVAE_ERR_ARG beyond)

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
        Nb, Cc, s, _ = x.shape
        out = torch.empty(Nb, Cc, s // 2, s // 2, device=x.device, dtype=torch.float32)
        lib = N.lib_for(x)
        for a, b in _chunks(Nb, EncoderMid.CHUNK):
            lib.check(N.timed(f"enc_mid_fwd_s{s}", x, lambda: lib.dll.kvae_enc_mid_fwd(
                N.ptr(x[a:b]), N.ptr(weight), N.ptr(bias), N.ptr(out[a:b]), b - a, Cc, s, N.stream_for(x))), "kvae_enc_mid_fwd")
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, out = ctx.saved_tensors
        g = g.contiguous()
        Nb, Cc, s, _ = x.shape
        lib = N.lib_for(x)
        g_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = gb = None
        for a, b in _chunks(Nb, EncoderMid.CHUNK):
            rows = lib.dll.kvae_enc_mid_partial_rows(b - a, s)
            wp = torch.empty(rows, weight.numel(), device=x.device, dtype=torch.float32)
            bp = torch.empty(rows, Cc, device=x.device, dtype=torch.float32)
            lib.check(lib.dll.kvae_enc_mid_bwd(N.ptr(x[a:b]), N.ptr(weight), N.ptr(out[a:b]), N.ptr(g[a:b]),
                                               N.ptr(g_x[a:b]) if g_x is not None else None, N.ptr(wp), N.ptr(bp), b - a, Cc, s,
                                               N.stream_for(x)), "kvae_enc_mid_bwd")
            cw, cb = colsum_pair(wp, bp)
            gw, gb = _acc(gw, cw), _acc(gb, cb)
        return g_x, gw.view_as(weight), gb


class DecoderUp(torch.autograd.Function):
    """out[N,32,2s,2s] = relu(pixel_shuffle_2(conv3x3(x[N,32,s,s], W[128,32,3,3]) + b)), s in {8, 4}: implicit GEMMs on the
    exact-f32 matrix cores with the weights stationary in registers (csrc/vae_conv_up.h; reference kvae/vae/vae.py:92-101).
    Bias, PixelShuffle and ReLU are part of the kernels in both directions."""

    CHUNK = 16384   # frames per launch (32-bit byte offsets inside a launch)

    @staticmethod
    def supported(x, conv):
        return (x.dim() == 4 and x.shape[1] == 32 and x.shape[2] == x.shape[3] and x.shape[2] in (8, 4)
                and tuple(conv.weight.shape) == (128, 32, 3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
                and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is not None)

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
        Nb, Cin, s, _ = x.shape
        out = torch.empty(Nb, 32, 2 * s, 2 * s, device=x.device, dtype=torch.float32)
        lib = N.lib_for(x)
        for a, b in _chunks(Nb, DecoderUp.CHUNK):
            lib.check(N.timed(f"dec_up_fwd_s{s}", x, lambda: lib.dll.kvae_dec_up_fwd(
                N.ptr(x[a:b]), N.ptr(weight), N.ptr(bias), N.ptr(out[a:b]), b - a, Cin, s, N.stream_for(x))), "kvae_dec_up_fwd")
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, out = ctx.saved_tensors
        g = g.contiguous()
        Nb, Cin, s, _ = x.shape
        lib = N.lib_for(x)
        g_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = gb = None
        for a, b in _chunks(Nb, DecoderUp.CHUNK):
            rows = lib.dll.kvae_dec_up_partial_rows(b - a, s)
            wp = torch.empty(rows, weight.numel(), device=x.device, dtype=torch.float32)
            bp = torch.empty(rows, 128, device=x.device, dtype=torch.float32)
            lib.check(N.timed(f"dec_up_bwd_s{s}", x, lambda: lib.dll.kvae_dec_up_bwd(
                N.ptr(x[a:b]), N.ptr(weight), N.ptr(out[a:b]), N.ptr(g[a:b]), N.ptr(g_x[a:b]) if g_x is not None else None, N.ptr(wp),
                N.ptr(bp), b - a, Cin, s, N.stream_for(x))), "kvae_dec_up_bwd")
            cw, cb = colsum_pair(wp, bp)
            gw, gb = _acc(gw, cw), _acc(gb, cb)
        return g_x, gw.view_as(weight), gb


def _optr(t):
    return N.ptr(t) if t is not None else None


class EncoderHead(torch.autograd.Function):
    """(a, mu, var) from the flattened encoder features in one kernel: fc_mu, fc_var + Sigmoid, noise_emission and the
    reparameterisation a = mu + eps sqrt(var + 1e-6) (reference kvae/vae/vae.py:33-41, kvae/model/model.py:81-84);
    eps=None gives a = mu.  csrc/vae_heads.h."""

    @staticmethod
    def supported(feat, fc_mu, fc_var_lin):
        return (feat.dim() == 2 and feat.shape[1] == 512 and tuple(fc_mu.weight.shape) == (2, 512)
                and tuple(fc_var_lin.weight.shape) == (2, 512) and fc_mu.bias is not None and fc_var_lin.bias is not None)

    @staticmethod
    def forward(ctx, feat, w_mu, b_mu, w_var, b_var, eps, ne):
        feat, w_mu, b_mu, w_var, b_var = (t.contiguous() for t in (feat, w_mu, b_mu, w_var, b_var))
        eps = eps.contiguous() if eps is not None else None
        Nb = feat.shape[0]
        mu, var, a = (torch.empty(Nb, 2, device=feat.device, dtype=torch.float32) for _ in range(3))
        lib = N.lib_for(feat)
        lib.check(lib.dll.kvae_enc_head_fwd(N.ptr(feat), N.ptr(w_mu), N.ptr(b_mu), N.ptr(w_var), N.ptr(b_var), _optr(eps),
                                            N.ptr(mu), N.ptr(var), N.ptr(a), Nb, 512, 2, float(ne), N.stream_for(feat)),
                  "kvae_enc_head_fwd")
        ctx.save_for_backward(feat, w_mu, w_var, var, eps)
        ctx.ne = float(ne)
        return a, mu, var

    @staticmethod
    def backward(ctx, g_a, g_mu, g_var):
        feat, w_mu, w_var, var, eps = ctx.saved_tensors
        g_a, g_mu, g_var = (t.contiguous() if t is not None else None for t in (g_a, g_mu, g_var))
        Nb = feat.shape[0]
        lib = N.lib_for(feat)
        rows = lib.dll.kvae_head_partial_rows()
        g_feat = torch.empty_like(feat)
        wp = torch.empty(rows, 4 * 512, device=feat.device, dtype=torch.float32)
        bp = torch.empty(rows, 4, device=feat.device, dtype=torch.float32)
        lib.check(lib.dll.kvae_enc_head_bwd(N.ptr(feat), N.ptr(w_mu), N.ptr(w_var), N.ptr(var), _optr(eps), _optr(g_a), _optr(g_mu),
                                            _optr(g_var), N.ptr(g_feat), N.ptr(wp), N.ptr(bp), Nb, 512, 2, ctx.ne,
                                            N.stream_for(feat)), "kvae_enc_head_bwd")
        gw, gb = colsum_pair(wp, bp)
        gw, gb = gw.view(2, 2, 512), gb.view(2, 2)
        return g_feat, gw[0], gb[0], gw[1], gb[1], None, None


class DecoderFc(torch.autograd.Function):
    """h[N,512] = a[N,2] W[512,2]^T + b (reference kvae/vae/vae.py:88-90) and its gradients, one kernel each way."""

    @staticmethod
    def supported(a, fc):
        return a.dim() == 2 and a.shape[1] == 2 and tuple(fc.weight.shape) == (512, 2) and fc.bias is not None

    @staticmethod
    def forward(ctx, a, weight, bias):
        a, weight, bias = a.contiguous(), weight.contiguous(), bias.contiguous()
        Nb = a.shape[0]
        h = torch.empty(Nb, 512, device=a.device, dtype=torch.float32)
        lib = N.lib_for(a)
        lib.check(lib.dll.kvae_dec_fc_fwd(N.ptr(a), N.ptr(weight), N.ptr(bias), N.ptr(h), Nb, 512, 2, N.stream_for(a)),
                  "kvae_dec_fc_fwd")
        ctx.save_for_backward(a, weight)
        return h

    @staticmethod
    def backward(ctx, g):
        a, weight = ctx.saved_tensors
        g = g.contiguous()
        Nb = a.shape[0]
        lib = N.lib_for(a)
        rows = lib.dll.kvae_head_partial_rows()
        g_a = torch.empty_like(a)
        wp = torch.empty(rows, 1024, device=a.device, dtype=torch.float32)
        bp = torch.empty(rows, 512, device=a.device, dtype=torch.float32)
        lib.check(lib.dll.kvae_dec_fc_bwd(N.ptr(g), N.ptr(a), N.ptr(weight), N.ptr(g_a), N.ptr(wp), N.ptr(bp), Nb, 512, 2,
                                          N.stream_for(a)), "kvae_dec_fc_bwd")
        gw, gb = colsum_pair(wp, bp)
        return g_a, gw.view(512, 2), gb


class LatentReg(torch.autograd.Function):
    """[...]-shaped sum over the latent dimension of log N(a;0,1) - log N(a;mu,var) (reference kvae/vae/losses.py:64-66)."""

    @staticmethod
    def forward(ctx, a, mu, var):
        a, mu, var = a.contiguous(), mu.contiguous(), var.contiguous()
        A = a.shape[-1]
        Nb = a.numel() // A
        reg = torch.empty(a.shape[:-1], device=a.device, dtype=torch.float32)
        lib = N.lib_for(a)
        lib.check(lib.dll.kvae_latent_reg_fwd(N.ptr(a), N.ptr(mu), N.ptr(var), N.ptr(reg), Nb, A, N.stream_for(a)),
                  "kvae_latent_reg_fwd")
        ctx.save_for_backward(a, mu, var)
        return reg

    @staticmethod
    def backward(ctx, g):
        a, mu, var = ctx.saved_tensors
        g = g.contiguous()
        A = a.shape[-1]
        g_a, g_mu, g_var = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        lib = N.lib_for(a)
        lib.check(lib.dll.kvae_latent_reg_bwd(N.ptr(a), N.ptr(mu), N.ptr(var), N.ptr(g), N.ptr(g_a), N.ptr(g_mu), N.ptr(g_var),
                                              a.numel() // A, A, N.stream_for(a)), "kvae_latent_reg_bwd")
        return g_a, g_mu, g_var


class LossHead(torch.autograd.Function):
    """(loss, elbo_total, elbo_kf, vae_elbo, recon, reg) from the per-frame terms in one launch each way (csrc/vae_heads.h;
    reference kvae/vae/losses.py:45-69, kvae/model/model.py:214-232).  Only `loss` carries gradient."""

    @staticmethod
    def forward(ctx, lpx, regf, elbo_kf, mask, beta, scale, vae_w, kf_w, weights=None):
        """weights: optional fp32 device tensor (vae_weight, kf_weight) read by the kernel at run time (overrides the floats)."""
        lpx, regf = lpx.contiguous(), regf.contiguous()
        mask = mask.contiguous() if mask is not None else None
        kf = elbo_kf.reshape(1).contiguous()
        b = beta.reshape(1).to(device=lpx.device, dtype=torch.float32).contiguous() if torch.is_tensor(beta) else \
            torch.full((1,), float(beta), device=lpx.device, dtype=torch.float32)
        out, coef = torch.empty(6, device=lpx.device, dtype=torch.float32), torch.empty(3, device=lpx.device, dtype=torch.float32)
        lib = N.lib_for(lpx)
        lib.check(lib.dll.kvae_loss_head_fwd(N.ptr(lpx), N.ptr(regf), _optr(mask), N.ptr(kf), N.ptr(b), float(scale), float(vae_w),
                                             float(kf_w), _optr(weights), N.ptr(out), N.ptr(coef), lpx.numel(), N.stream_for(lpx)),
                  "kvae_loss_head_fwd")
        ctx.save_for_backward(coef, mask)
        ctx.set_materialize_grads(False)   # five of the six outputs never carry a gradient: no zero fills for them
        ctx.shape, ctx.kf_shape = lpx.shape, elbo_kf.shape
        vals = out.unbind(0)
        ctx.mark_non_differentiable(*vals[1:])
        return vals

    @staticmethod
    def backward(ctx, g_loss, *_unused):
        if g_loss is None:
            return (None,) * 9
        coef, mask = ctx.saved_tensors
        g = g_loss.reshape(1).contiguous()
        g_lpx = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        g_reg = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        g_kf = torch.empty(1, device=g.device, dtype=torch.float32)
        lib = N.lib_for(g)
        lib.check(lib.dll.kvae_loss_head_bwd(N.ptr(g), N.ptr(coef), _optr(mask), N.ptr(g_lpx), N.ptr(g_reg), N.ptr(g_kf),
                                             g_lpx.numel(), N.stream_for(g)), "kvae_loss_head_bwd")
        return g_lpx, g_reg, g_kf.reshape(ctx.kf_shape), None, None, None, None, None, None


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
