// vae_epilogue.h — fused conv epilogues of the frame VAE (SURVEY §8f row 3 territory): the convolutions stay on
// MIOpen, but everything PyTorch appends to them as separate full-tensor passes —
//     + bias  ->  PixelShuffle(r)  ->  ReLU            (decoder, reference vae.py:92-101; r = 1 for the encoder, :20-31)
// — is ONE pass here (read the conv output once, write the activation once), and its backward
//     g_conv = pixel_unshuffle(g_out * [out > 0])
// is one pass as well.  At configs[1] the decoder's second layer alone moves 419 MB per tensor, so every avoided
// pass is ~0.1 ms.  Pure HBM streaming: coalesced along W, no LDS, grid-stride over elements.
//
//   in  : [N, C*r*r, H, W]   (conv output, no bias)      bias : [C*r*r]
//   out : [N, C, H*r, W*r]
#pragma once
#include <stdint.h>

#include "lgssm_vm.h"

namespace kvae {

struct EpiShape { int64_t N; int C, H, W, r; };

// flat index in `in` that feeds flat index `o` of `out`, and its channel
KV_DEV int64_t epi_src(const EpiShape &s, int64_t o, int *ch) {
  const int OW = s.W * s.r, OH = s.H * s.r;
  const int ow = (int)(o % OW);
  const int64_t t1 = o / OW;
  const int oh = (int)(t1 % OH);
  const int64_t t2 = t1 / OH;
  const int c = (int)(t2 % s.C);
  const int64_t n = t2 / s.C;
  const int cc = c * s.r * s.r + (oh % s.r) * s.r + (ow % s.r);
  *ch = cc;
  return ((n * (s.C * s.r * s.r) + cc) * s.H + oh / s.r) * s.W + ow / s.r;
}

KV_DEV void epi_fwd_elem(const EpiShape &s, const float *in, const float *bias, float *out, int64_t o, int relu) {
  int ch;
  const int64_t i = epi_src(s, o, &ch);
  float v = in[i] + bias[ch];
  if (relu) v = v > 0.f ? v : 0.f;
  out[o] = v;
}

KV_DEV void epi_bwd_elem(const EpiShape &s, const float *g_out, const float *out, float *g_in, int64_t o, int relu) {
  int ch;
  const int64_t i = epi_src(s, o, &ch);
  float g = g_out[o];
  if (relu && !(out[o] > 0.f)) g = 0.f;
  g_in[i] = g;
}

}  // namespace kvae
