/*
 * lgssm_oracle.c — CPU ORACLE (test infrastructure, not product code).
 *
 * A plain scalar C restatement, in fp32 and in the reference's operation order, of the forward
 * algorithm of the hot path:
 *   kvae_oracle_smooth : KalmanFilter.filter_step / filter / smooth_step / smooth
 *                        (reference kvae/kalman/kalman_filter.py:31-104, 107-201, 204-237, 240-279)
 *   kvae_oracle_elbo   : KalmanFilter._safe_cholesky / elbo (kalman_filter.py:282-302, 305-401),
 *                        torch.distributions.MultivariateNormal log_prob / rsample semantics
 *   kvae_oracle_mix    : the mixing einsums (dyn_param.py:58-60, switch_dyn_param.py:82-84)
 * Gradients are not restated here: the reference has no explicit backward (autograd), so the
 * gradient oracle is autograd over oracle/torch_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Pinned to the golden vectors captured from the reference: tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MAXD 16
#define LOG2PI 1.8378770664093453f

typedef struct { const float *ptr; int64_t sb, st; } stack_t;
static const float *at(stack_t s, int b, int t) { return s.ptr + b * s.sb + t * s.st; }

/* C[r x c] = A[r x k] * B[k x c] */
static void mm(float *C, const float *A, const float *B, int r, int k, int c) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) {
      float acc = 0.f;
      for (int q = 0; q < k; ++q) acc += A[i * k + q] * B[q * c + j];
      C[i * c + j] = acc;
    }
}
static void tr(float *T, const float *A, int r, int c) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) T[j * r + i] = A[i * c + j];
}

/* X = M^{-1} RHS, M [r x r], RHS [r x nr]; LU with partial pivoting (getrf/getrs). */
static int lu_solve(const float *M, const float *RHS, float *X, int r, int nr) {
  float a[MAXD * MAXD], b[MAXD * 2 * MAXD];
  memcpy(a, M, sizeof(float) * r * r);
  memcpy(b, RHS, sizeof(float) * r * nr);
  for (int c = 0; c < r; ++c) {
    int piv = c;
    for (int i = c + 1; i < r; ++i)
      if (fabsf(a[i * r + c]) > fabsf(a[piv * r + c])) piv = i;
    if (piv != c) {
      for (int j = 0; j < r; ++j) { float t = a[c * r + j]; a[c * r + j] = a[piv * r + j]; a[piv * r + j] = t; }
      for (int j = 0; j < nr; ++j) { float t = b[c * nr + j]; b[c * nr + j] = b[piv * nr + j]; b[piv * nr + j] = t; }
    }
    if (a[c * r + c] == 0.f) return 1;
    const float rinv = 1.0f / a[c * r + c];
    for (int i = c + 1; i < r; ++i) {
      const float l = a[i * r + c] * rinv;
      for (int j = c + 1; j < r; ++j) a[i * r + j] -= l * a[c * r + j];
      for (int j = 0; j < nr; ++j) b[i * nr + j] -= l * b[c * nr + j];
    }
  }
  for (int j = 0; j < nr; ++j)
    for (int c = r - 1; c >= 0; --c) {
      float acc = b[c * nr + j];
      for (int k = c + 1; k < r; ++k) acc -= a[c * r + k] * X[k * nr + j];
      X[c * nr + j] = acc / a[c * r + c];
    }
  return 0;
}

/* lower Cholesky (potf2 order); returns 0 on success */
static int chol(const float *A, float *L, int n) {
  memset(L, 0, sizeof(float) * n * n);
  for (int c = 0; c < n; ++c) {
    float d = A[c * n + c];
    for (int k = 0; k < c; ++k) d -= L[c * n + k] * L[c * n + k];
    if (!(d > 0.f)) return 1;
    const float sd = sqrtf(d);
    L[c * n + c] = sd;
    for (int i = c + 1; i < n; ++i) {
      float s = A[i * n + c];
      for (int k = 0; k < c; ++k) s -= L[i * n + k] * L[c * n + k];
      L[i * n + c] = s / sd;
    }
  }
  return 0;
}

int kvae_oracle_smooth(int B, int T, int n, int m, int p, const float *Y, const float *U, const float *mask,
                       stack_t A, stack_t Bm, stack_t C, stack_t Q, const float *R, const float *mu0,
                       const float *Sigma0, float *mus_f, float *Sig_f, float *mus_p, float *Sig_p, float *mus_s,
                       float *Sig_s) {
  const int nn = n * n;
  for (int b = 0; b < B; ++b) {
    float mu[MAXD], Sig[MAXD * MAXD];
    memcpy(mu, mu0, sizeof(float) * n);
    memcpy(Sig, Sigma0, sizeof(float) * nn);
    for (int t = 0; t < T; ++t) {
      const float *At = at(A, b, t), *Bt = at(Bm, b, t), *Ct = at(C, b, t), *Qt = at(Q, b, t);
      const float *y = Y + ((int64_t)b * T + t) * p, *u = U + ((int64_t)b * T + t) * m;
      const float mk = mask ? mask[(int64_t)b * T + t] : 1.0f;
      float mup[MAXD], Sp[MAXD * MAXD], t1[MAXD * MAXD], t2[MAXD * MAXD], AT[MAXD * MAXD], CT[MAXD * MAXD];
      /* :65 mu_p = A mu + B u ; :67 Sig_p = A Sig A^T + Q */
      for (int i = 0; i < n; ++i) {
        float a1 = 0.f, a2 = 0.f;
        for (int k = 0; k < n; ++k) a1 += At[i * n + k] * mu[k];
        for (int k = 0; k < m; ++k) a2 += Bt[i * m + k] * u[k];
        mup[i] = a1 + a2;
      }
      tr(AT, At, n, n);
      mm(t1, At, Sig, n, n, n);
      mm(Sp, t1, AT, n, n, n);
      for (int e = 0; e < nn; ++e) Sp[e] += Qt[e];
      /* :73-79 innovation */
      float r[MAXD], S[MAXD * MAXD], PCT[MAXD * MAXD], PCTt[MAXD * MAXD], Kt[MAXD * MAXD], K[MAXD * MAXD];
      for (int i = 0; i < p; ++i) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc += Ct[i * n + k] * mup[k];
        r[i] = y[i] - acc;
      }
      tr(CT, Ct, p, n);
      mm(t1, Ct, Sp, p, n, n);
      mm(t2, t1, CT, p, n, p);
      for (int e = 0; e < p * p; ++e) t2[e] += R[e];
      for (int i = 0; i < p; ++i)
        for (int j = 0; j < p; ++j) S[i * p + j] = 0.5f * (t2[i * p + j] + t2[j * p + i]);
      /* :82-92 gain */
      mm(PCT, Sp, CT, n, n, p);
      tr(PCTt, PCT, n, p);
      if (lu_solve(S, PCTt, Kt, p, n)) return 1;
      tr(K, Kt, p, n);
      for (int e = 0; e < n * p; ++e) K[e] *= mk;
      /* :96-101 Joseph update */
      float muf[MAXD], IKC[MAXD * MAXD], IKCt[MAXD * MAXD], KR[MAXD * MAXD], Ktm[MAXD * MAXD], F[MAXD * MAXD];
      for (int i = 0; i < n; ++i) {
        float acc = 0.f;
        for (int k = 0; k < p; ++k) acc += K[i * p + k] * r[k];
        muf[i] = mup[i] + acc;
      }
      mm(t1, K, Ct, n, p, n);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) IKC[i * n + j] = (i == j ? 1.f : 0.f) - t1[i * n + j];
      tr(IKCt, IKC, n, n);
      mm(t1, IKC, Sp, n, n, n);
      mm(F, t1, IKCt, n, n, n);
      mm(KR, K, R, n, p, p);
      tr(Ktm, K, n, p);
      mm(t2, KR, Ktm, n, p, n);
      for (int e = 0; e < nn; ++e) F[e] += t2[e];
      const int64_t q = (int64_t)b * T + t;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Sig[i * n + j] = 0.5f * (F[i * n + j] + F[j * n + i]);
      memcpy(mu, muf, sizeof(float) * n);
      memcpy(mus_f + q * n, mu, sizeof(float) * n);
      memcpy(Sig_f + q * nn, Sig, sizeof(float) * nn);
      memcpy(mus_p + q * n, mup, sizeof(float) * n);
      memcpy(Sig_p + q * nn, Sp, sizeof(float) * nn);
    }
    if (!mus_s) continue;
    /* RTS: :249-272 */
    float ms[MAXD], Ss[MAXD * MAXD];
    int64_t q = (int64_t)b * T + T - 1;
    memcpy(ms, mus_f + q * n, sizeof(float) * n);
    memcpy(Ss, Sig_f + q * nn, sizeof(float) * nn);
    memcpy(mus_s + q * n, ms, sizeof(float) * n);
    memcpy(Sig_s + q * nn, Ss, sizeof(float) * nn);
    for (int t = T - 2; t >= 0; --t) {
      q = (int64_t)b * T + t;
      const float *Sf = Sig_f + q * nn, *Spn = Sig_p + (q + 1) * nn, *An = at(A, b, t + 1);
      float AT[MAXD * MAXD], W[MAXD * MAXD], Wt[MAXD * MAXD], SpT[MAXD * MAXD], X[MAXD * MAXD], J[MAXD * MAXD], Jt[MAXD * MAXD];
      float D[MAXD * MAXD], t1[MAXD * MAXD], t2[MAXD * MAXD];
      tr(AT, An, n, n);
      mm(W, Sf, AT, n, n, n);
      tr(Wt, W, n, n);
      tr(SpT, Spn, n, n);
      if (lu_solve(SpT, Wt, X, n, n)) return 1; /* :229 */
      tr(J, X, n, n);
      tr(Jt, J, n, n);
      float nm[MAXD];
      for (int i = 0; i < n; ++i) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc += J[i * n + k] * (ms[k] - mus_p[(q + 1) * n + k]);
        nm[i] = mus_f[q * n + i] + acc; /* :232 */
      }
      for (int e = 0; e < nn; ++e) D[e] = Ss[e] - Spn[e];
      mm(t1, J, D, n, n, n);
      mm(t2, t1, Jt, n, n, n);
      for (int e = 0; e < nn; ++e) t2[e] += Sf[e]; /* :234 */
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Ss[i * n + j] = 0.5f * (t2[i * n + j] + t2[j * n + i]);
      memcpy(ms, nm, sizeof(float) * n);
      memcpy(mus_s + q * n, ms, sizeof(float) * n);
      memcpy(Sig_s + q * nn, Ss, sizeof(float) * nn);
    }
  }
  return 0;
}

/* _safe_cholesky over a batch of matrices fetched through get(i): whole-batch retry (:289-302) */
static float jitter_at(int level) { double j = 1e-6; for (int i = 0; i < level; ++i) j *= 10.0; return (float)j; }

static int safe_level(int count, int n, const float *(*get)(void *, int), void *ctx) {
  float S[MAXD * MAXD], L[MAXD * MAXD];
  for (int level = 0; level < 5; ++level) {
    int ok = 1;
    for (int i = 0; i < count && ok; ++i) {
      const float *X = get(ctx, i);
      for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) S[r * n + c] = 0.5f * (X[r * n + c] + X[c * n + r]) + (r == c ? jitter_at(level) : 0.f);
      if (chol(S, L, n)) ok = 0;
    }
    if (ok) return level;
  }
  return 5;
}
static void safe_chol(const float *X, float *L, int n, int level) {
  float S[MAXD * MAXD];
  if (level >= 5) {
    memset(L, 0, sizeof(float) * n * n);
    for (int i = 0; i < n; ++i) L[i * n + i] = sqrtf(fmaxf(X[i * n + i], 1e-6f));
    return;
  }
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) S[r * n + c] = 0.5f * (X[r * n + c] + X[c * n + r]) + (r == c ? jitter_at(level) : 0.f);
  chol(S, L, n);
}
/* log N(x; 0, L L^T) */
static float mvn_logprob(const float *x, const float *L, int n) {
  float w[MAXD], quad = 0.f, ld = 0.f;
  for (int i = 0; i < n; ++i) {
    float acc = x[i];
    for (int k = 0; k < i; ++k) acc -= L[i * n + k] * w[k];
    w[i] = acc / L[i * n + i];
    quad += w[i] * w[i];
    ld += logf(L[i * n + i]);
  }
  return -0.5f * (n * LOG2PI + quad) - ld;
}

typedef struct { const float *base; int64_t stride; } seqctx;
static const float *get_seq(void *c, int i) { seqctx *s = (seqctx *)c; return s->base + i * s->stride; }
typedef struct { stack_t Q; int T; } qctx;
static const float *get_q(void *c, int i) { qctx *s = (qctx *)c; int Tm = s->T - 1; return at(s->Q, i / Tm, 1 + i % Tm); }

/* terms[4] = sums over (b,t) of {transition, emission, init, entropy}; levels[2] = jitter levels used */
int kvae_oracle_elbo(int B, int T, int n, int m, int p, const float *mus, const float *Sigs, const float *eps,
                     const float *Y, const float *U, const float *mask, stack_t A, stack_t Bm, stack_t C, stack_t Q,
                     const float *R, const float *mu0, const float *Sigma0, double *terms, int *levels) {
  const int nn = n * n;
  seqctx sc = {Sigs, nn};
  const int lvS = safe_level(B * T, n, get_seq, &sc);
  qctx qc = {Q, T};
  const int lvQ = T > 1 ? safe_level(B * (T - 1), n, get_q, &qc) : 0;
  levels[0] = lvS; levels[1] = lvQ;
  float LR[MAXD * MAXD], L0[MAXD * MAXD];
  if (chol(R, LR, p) || chol(Sigma0, L0, n)) return 1;
  double tr_ = 0, em_ = 0, in_ = 0, en_ = 0;
  for (int b = 0; b < B; ++b) {
    float zprev[MAXD];
    for (int t = 0; t < T; ++t) {
      const int64_t q = (int64_t)b * T + t;
      float L[MAXD * MAXD], z[MAXD], d[MAXD];
      safe_chol(Sigs + q * nn, L, n, lvS);
      for (int i = 0; i < n; ++i) {
        float acc = 0.f;
        for (int k = 0; k <= i; ++k) acc += L[i * n + k] * eps[q * n + k];
        z[i] = mus[q * n + i] + acc; /* :351 */
      }
      if (t >= 1) { /* :353-369 */
        const float *At = at(A, b, t), *Bt = at(Bm, b, t), *u = U + q * m;
        float LQ[MAXD * MAXD];
        safe_chol(at(Q, b, t), LQ, n, lvQ);
        for (int i = 0; i < n; ++i) {
          float a1 = 0.f, a2 = 0.f;
          for (int k = 0; k < n; ++k) a1 += At[i * n + k] * zprev[k];
          for (int k = 0; k < m; ++k) a2 += Bt[i * m + k] * u[k];
          d[i] = z[i] - (a1 + a2);
        }
        tr_ += mvn_logprob(d, LQ, n);
      }
      { /* :372-377 */
        const float *Ct = at(C, b, t), *y = Y + q * p;
        float e[MAXD];
        for (int i = 0; i < p; ++i) {
          float acc = 0.f;
          for (int k = 0; k < n; ++k) acc += Ct[i * n + k] * z[k];
          e[i] = y[i] - acc;
        }
        em_ += (mask ? mask[q] : 1.0f) * mvn_logprob(e, LR, p);
      }
      if (t == 0) { /* :380-381 */
        for (int i = 0; i < n; ++i) d[i] = z[i] - mu0[i];
        in_ += mvn_logprob(d, L0, n);
      }
      for (int i = 0; i < n; ++i) d[i] = z[i] - mus[q * n + i];
      en_ -= mvn_logprob(d, L, n); /* :389 */
      memcpy(zprev, z, sizeof(float) * n);
    }
  }
  terms[0] = tr_; terms[1] = em_; terms[2] = in_; terms[3] = en_;
  return 0;
}

void kvae_oracle_mix(const float *alpha, const float *base, float *out, int64_t rows, int K, int E) {
  for (int64_t r = 0; r < rows; ++r)
    for (int e = 0; e < E; ++e) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc += alpha[r * K + k] * base[k * E + e];
      out[r * E + e] = acc;
    }
}
