// vae_loss.h — Bernoulli reconstruction term of the frame VAE, fused:
//   frame_ll[f] = - sum_pixels BCEWithLogits(logit, x)        (reference losses.py:85-87: F.binary_cross_entropy_with_logits
//                                                               (reduction='none').sum(dim=(2,3,4)), negated)
// One wavefront per frame: 16-byte loads, in-register accumulation, one shuffle reduction, one float out.  The reference
// materialises the per-pixel loss tensor (52 MB at configs[1]) and reduces it in a second pass; its autograd adds three
// more element-wise passes.  Backward here: g_logit = -g_frame[f] * (sigmoid(logit) - x), one pass.
#pragma once
#include "lgssm_vm.h"

namespace kvae {

// numerically stable, same formula torch uses: max(l,0) - l*x + log1p(exp(-|l|))
KV_DEV float bce_logit(float l, float x) { return fmaxf(l, 0.f) - l * x + log1pf(expf(-fabsf(l))); }
KV_DEV float sigmoid_stable(float l) {
  const float e = expf(-fabsf(l));
  return l >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
}

}  // namespace kvae
