// lstm.h — the alpha-network recurrence of the "lstm" dynamics (reference: nn.LSTM(p -> 50) stepped
// inside the filter loop, dyn_param.py:22-27,50-52) as ONE launch: one wavefront per sequence, the
// recurrent weights resident in LDS, the whole T loop in-kernel, zero initial state.
//
// Why hand-written: MIOpen's LSTM (hipBLASLt GEMMs inside) cannot be captured into a hipGraph on
// ROCm 7.2 ("operation not permitted when stream is capturing") and costs tens of launches per
// step for a 50-unit cell; with all frames observed the LSTM input at step t is simply a_{t-1}.
//
//   lstm_fwd_body : h_t, c_t and the post-activation gates (i,f,g,o order, as torch) for t = 0..T-1
//   lstm_bwd_body : BPTT t = T-1..0: pre-activation gate gradients d_pre[B,T,4H] and dx[B,T,I];
//                   the weight gradients are three plain GEMM/sum calls on d_pre (host side).
#pragma once
#include "lgssm_vm.h"

#define KVAE_LSTM_MAX_H 52
#define KVAE_LSTM_MAX_I 16

namespace kvae {

struct LstmLds {
  float W[4 * KVAE_LSTM_MAX_H * KVAE_LSTM_MAX_H];   // fwd: W_hh transposed [k][j]; bwd: W_hh row-major [j][k]
  float Wi[4 * KVAE_LSTM_MAX_H * KVAE_LSTM_MAX_I];  // W_ih [j][q]
  float bias[4 * KVAE_LSTM_MAX_H];
  float h[KVAE_LSTM_MAX_H], c[KVAE_LSTM_MAX_H], x[KVAE_LSTM_MAX_I];
  float g[4 * KVAE_LSTM_MAX_H];                      // gates of the current step (fwd: activated, bwd: d_pre)
  float dh[KVAE_LSTM_MAX_H], dc[KVAE_LSTM_MAX_H];
};

KV_DEV float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

KV_DEV void lstm_fwd_body(const float *x, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                          float *h_seq, float *gates, float *c_seq, int b, int T, int I, int H, LstmLds &L) {
  const int G = 4 * H;
  KV_PAR(e, G * H) {  // transpose so that lanes (consecutive j) read consecutive LDS words
    const int j = e / H, k = e - j * H;
    L.W[k * G + j] = w_hh[e];
  }
  KV_PAR(e, G * I) { L.Wi[e] = w_ih[e]; }
  KV_PAR(j, G) { L.bias[j] = b_ih[j] + b_hh[j]; }
  KV_PAR(u, H) { L.h[u] = 0.0f; L.c[u] = 0.0f; }
  KV_SYNC();
  for (int t = 0; t < T; ++t) {
    const int64_t q = (int64_t)b * T + t;
    KV_PAR(i, I) { L.x[i] = x[q * I + i]; }
    KV_SYNC();
    KV_PAR(j, G) {
      float acc = L.bias[j];
      for (int i = 0; i < I; ++i) acc = fmaf(L.Wi[j * I + i], L.x[i], acc);
      for (int k = 0; k < H; ++k) acc = fmaf(L.W[k * G + j], L.h[k], acc);
      const float a = (j >= 2 * H && j < 3 * H) ? tanhf(acc) : sigmoidf_(acc);
      L.g[j] = a;
      gates[q * G + j] = a;
    }
    KV_SYNC();
    KV_PAR(u, H) {
      const float cn = L.g[H + u] * L.c[u] + L.g[u] * L.g[2 * H + u];
      const float hn = L.g[3 * H + u] * tanhf(cn);
      L.c[u] = cn;
      L.h[u] = hn;
      c_seq[q * H + u] = cn;
      h_seq[q * H + u] = hn;
    }
    KV_SYNC();
  }
}

KV_DEV void lstm_bwd_body(const float *g_h, const float *gates, const float *c_seq, const float *w_ih, const float *w_hh,
                          float *d_pre, float *dx, int b, int T, int I, int H, LstmLds &L) {
  const int G = 4 * H;
  KV_PAR(e, G * H) { L.W[e] = w_hh[e]; }
  KV_PAR(e, G * I) { L.Wi[e] = w_ih[e]; }
  KV_PAR(u, H) { L.dh[u] = 0.0f; L.dc[u] = 0.0f; }
  KV_SYNC();
  for (int t = T - 1; t >= 0; --t) {
    const int64_t q = (int64_t)b * T + t;
    KV_PAR(u, H) {
      const float ig = gates[q * G + u], fg = gates[q * G + H + u], gg = gates[q * G + 2 * H + u],
                  og = gates[q * G + 3 * H + u];
      const float ct = c_seq[q * H + u];
      const float cprev = t > 0 ? c_seq[(q - 1) * H + u] : 0.0f;
      const float tc = tanhf(ct);
      const float dh = g_h[q * H + u] + L.dh[u];
      const float dct = dh * og * (1.0f - tc * tc) + L.dc[u];
      const float dai = dct * gg * ig * (1.0f - ig);
      const float daf = dct * cprev * fg * (1.0f - fg);
      const float dag = dct * ig * (1.0f - gg * gg);
      const float dao = dh * tc * og * (1.0f - og);
      L.dc[u] = dct * fg;
      L.g[u] = dai; L.g[H + u] = daf; L.g[2 * H + u] = dag; L.g[3 * H + u] = dao;
      d_pre[q * G + u] = dai; d_pre[q * G + H + u] = daf; d_pre[q * G + 2 * H + u] = dag; d_pre[q * G + 3 * H + u] = dao;
    }
    KV_SYNC();
    KV_PAR(k, H) {  // dh_{t-1} = W_hh^T d_pre
      float acc = 0.f;
      for (int j = 0; j < G; ++j) acc = fmaf(L.W[j * H + k], L.g[j], acc);
      L.dh[k] = acc;
    }
    KV_PAR(i, I) {  // dx_t = W_ih^T d_pre
      float acc = 0.f;
      for (int j = 0; j < G; ++j) acc = fmaf(L.Wi[j * I + i], L.g[j], acc);
      dx[q * I + i] = acc;
    }
    KV_SYNC();
  }
}

}  // namespace kvae
