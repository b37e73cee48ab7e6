/*
 * kvae_lgssm.h — C ABI of the MI355X-native LGSSM hot path of the Kalman-VAE
 * (libkvae_lgssm.so, built from kalman-vae_amd/csrc/kvae_lgssm.hip for gfx950).
 *
 * The reference (rodrigo-paganini/kalman-vae) has no FFI layer: this path lives behind the
 * Python class kvae.kalman.kalman_filter.KalmanFilter.  Each entry point below replaces the
 * aten-op sequence of one reference method; the file:line it replaces is cited on the function.
 * All pointers are DEVICE pointers to fp32, row-major; `stream` is a hipStream_t passed as
 * void* (NULL = default stream).  Calls are asynchronous: nothing here synchronises, allocates
 * or frees, so every call may be captured into a hipGraph.  Return value: kvae_status.
 *
 * Shapes: B sequences, T steps, n = dim z, m = dim u, p = dim a (all of n,m,p <= KVAE_MAX_DIM).
 */
#ifndef KVAE_LGSSM_H
#define KVAE_LGSSM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVAE_MAX_DIM 16
#define KVAE_ABI_VERSION 10

typedef enum {
  KVAE_OK = 0,
  KVAE_ERR_DIMS = 1,   /* n, m or p outside [1, KVAE_MAX_DIM], or B/T < 1              */
  KVAE_ERR_NULL = 2,   /* a required pointer is NULL                                    */
  KVAE_ERR_LAUNCH = 3, /* hipLaunchKernel / hipMemsetAsync failed (see kvae_last_error) */
  KVAE_ERR_ARG = 4     /* inconsistent arguments (e.g. K < 1)                           */
} kvae_status;

/* One per-step operand: element (b, t) starts at ptr + b*sb + t*st (strides in floats).
 * sb = st = 0 broadcasts a single matrix (K == 1 dynamics, constant Q). */
typedef struct {
  const float *ptr;
  int64_t sb, st;
} kvae_stack;

/* The LGSSM problem: what KalmanFilter.filter/smooth/elbo read (kalman_filter.py:8-28,107-150). */
typedef struct {
  int32_t B, T, n, m, p;
  kvae_stack A;            /* [n,n]  transition        A_t  (kalman_filter.py:153-160)      */
  kvae_stack Bm;           /* [n,m]  control           B_t                                  */
  kvae_stack C;            /* [p,n]  emission          C_t                                  */
  kvae_stack Q;            /* [n,n]  process noise     Q_t  (self.Q buffer or dyn.Q_seq)    */
  const float *R;          /* [p,p]  observation noise (kalman_filter.py:23)                */
  const float *mu0;        /* [n]    belief on z_{-1}; per sequence when mu0_sb != 0        */
  int64_t mu0_sb;
  const float *Sigma0;     /* [n,n]                                                         */
  int64_t Sigma0_sb;
  const float *Y;          /* [B,T,p] contiguous observations a_t                           */
  const float *U;          /* [B,T,m] contiguous controls                                   */
  const float *mask;       /* [B,T] 1 = observed, 0 = missing; NULL = all observed          */
} kvae_lgssm_problem;

/* Filtered / predicted / smoothed beliefs, each [B,T,...] contiguous (kalman_filter.py:193-201,274-279). */
typedef struct {
  float *mus_filt;      /* [B,T,n]   */
  float *Sigmas_filt;   /* [B,T,n,n] */
  float *mus_pred;      /* [B,T,n]   */
  float *Sigmas_pred;   /* [B,T,n,n] */
  float *mus_smooth;    /* [B,T,n]   (NULL for filter-only calls) */
  float *Sigmas_smooth; /* [B,T,n,n] */
  float *aux;           /* optional [B,T,KVAE_AUX(n,p)]: gains saved by the forward for the backward
                           (K unmasked [n,p] | S [p,p] | J [n,n]); NULL = recompute them. Only the
                           n=4,p=2 fast path reads/writes it. */
} kvae_lgssm_states;
#define KVAE_AUX(n, p) ((n) * (p) + (p) * (p) + (n) * (n))

/* Writable counterpart of kvae_stack: element (b,t) of a gradient stack starts at
 * ptr + b*sb + t*st.  Lets gA/gB/gC/gQ land directly in the slots of one packed [B,T,E] record. */
typedef struct {
  float *ptr;
  int64_t sb, st;
} kvae_gstack;

/* Gradients w.r.t. the problem's per-step inputs.  gA, gB, gC, gY are required by the backward
 * entry points; gQ.ptr, gU, g_mu0, g_Sigma0 may be NULL (not wanted).
 * gY [B,T,p], gU [B,T,m] contiguous; g_mu0 [B,n], g_Sigma0 [B,n,n] per sequence. */
typedef struct {
  kvae_gstack gA, gB, gC, gQ; /* per step [n,n] [n,m] [p,n] [n,n] */
  float *gY, *gU;
  float *g_mu0, *g_Sigma0;
} kvae_lgssm_input_grads;

/* ---- forward ------------------------------------------------------------------------------ */

/* Kalman filter over T steps from (mu0, Sigma0): replaces KalmanFilter.filter's time loop and
 * filter_step (kalman_filter.py:31-104, 151-201). Writes the four filt/pred stacks. */
int kvae_lgssm_filter_fwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *out, void *stream);

/* RTS smoother over already-filtered beliefs: replaces smooth_step and the reverse loop of
 * KalmanFilter.smooth (kalman_filter.py:204-237, 249-272). Reads filt/pred, writes smooth. */
int kvae_lgssm_rts_fwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *io, void *stream);

/* Filter + RTS in ONE call, the whole T loop in-kernel: sixteen sequences per wavefront at (n,m,p) = (4,4,2), one sequence per
 * wavefront on the f32 matrix cores at (16,16,2) and in the run-time-dimension kernels otherwise.  One launch, except at (4,4,2)
 * below 2048 sequences: filter sweep | all smoother gains at once | smoother sweep, three launches on `stream` (same results).
 * Replaces KalmanFilter.smooth (kalman_filter.py:240-279). Writes all six stacks (and, if out->aux is NULL at (4,4,2), uses
 * Sigmas_smooth as scratch for the gains before it holds the result). */
int kvae_lgssm_smooth_fwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *out, void *stream);

/* Kalman filter with the LSTM alpha-network stepped INSIDE the kernel (masked sequences: the network input of a
 * hidden step is C_t mu_{t|t-1}, kalman_filter.py:183-185; dyn_param.py:39-63).  prob->A/Bm/C are ignored: the step
 * matrices are mixed from the K mode matrices A[K,n,n], Bm[K,n,m], C[K,p,n] and ALSO written to record [B,T,n*n+n*m+p*n]
 * (A|B|C per step) and alpha [B,T,K].  The last four pointers are optional (NULL): the cell's internals kept for
 * kvae_lgssm_alpha_lstm_bwd - gates [B,T,4H] (post-activation, torch order i,f,g,o), c_seq and h_seq [B,T,H], x_seq [B,T,p]
 * (the cell inputs y_for_dyn).  Limits: H == 50, p == 2, K <= 16 (else KVAE_ERR_DIMS). */
int kvae_lgssm_filter_alpha_lstm(const kvae_lgssm_problem *prob, const kvae_lgssm_states *out, const float *w_ih,
                                 const float *w_hh, const float *b_ih, const float *b_hh, const float *head_w,
                                 const float *head_b, const float *A, const float *Bm, const float *C, int32_t K, int32_t H,
                                 float *record, float *alpha, float *gates, float *c_seq, float *h_seq, float *x_seq,
                                 void *stream);

/* ---- backward ----------------------------------------------------------------------------- */

/* Reverse-mode of kvae_lgssm_smooth_fwd (with_rts = 1) or kvae_lgssm_filter_fwd (with_rts = 0):
 * replaces what autograd records for kalman_filter.py:151-185, 257-272.  `saved` holds the forward
 * results; `up` holds upstream gradients of the six stacks (any pointer may be NULL = zero).
 * Scratch: ws [B,T,2*(n+n*n)] floats (adjoints handed from the smoother sweep to the filter sweep).  At (4,4,2) below 2048
 * sequences the call is four launches (each adjoint's dependent chain, then what hangs off it for all steps at once); the output
 * buffers gA, gB, gU hold intermediate values between them.  Outputs are complete when the call's last launch has run. */
int kvae_lgssm_smooth_bwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *saved,
                          const kvae_lgssm_states *up, const kvae_lgssm_input_grads *out,
                          float *ws, int with_rts, void *stream);

/* Reverse-mode of kvae_lgssm_filter_alpha_lstm (with_rts = 0) or of it followed by kvae_lgssm_rts_fwd (with_rts = 1) in
 * ONE launch: the adjoints of the filter and of the LSTM cell are interleaved step by step, because on hidden frames the
 * cell input is C_t mu_{t|t-1} (kalman_filter.py:183-185; dyn_param.py:39-63) - what autograd unrolls into T cell steps and
 * T filter steps.  prob->A/Bm/C must be the stacks of the forward's `record`; out->gA/gB/gC must point into g_record
 * [B,T,n*n+n*m+p*n] with the same slot offsets.  Upstream: `up` (six stacks, any NULL), g_record_up / g_alpha_up (NULL = none).
 * Results: out->gY, gU (+ g_mu0, g_Sigma0), g_record (total gradient of the step records: reduce with kvae_mix_bwd for the
 * mode matrices), d_pre [B,T,4H] (gate pre-activation gradients: the LSTM weight gradients are GEMMs on it, as for
 * kvae_lstm_bwd) and g_logit [B,T,K] (head gradients likewise).  ws as for kvae_lgssm_smooth_bwd. */
int kvae_lgssm_alpha_lstm_bwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                              const kvae_lgssm_input_grads *out, float *ws, int with_rts, const float *w_ih,
                              const float *w_hh, const float *head_w, const float *A, const float *Bm, const float *C,
                              int32_t K, int32_t H, const float *alpha, const float *gates, const float *c_seq,
                              const float *g_record_up, const float *g_alpha_up, float *g_record, float *d_pre,
                              float *g_logit, void *stream);

/* ---- ELBO --------------------------------------------------------------------------------- */

/* The LGSSM terms of KalmanFilter.elbo (kalman_filter.py:347-389) for z = mu_s + chol(Sigma_s) eps:
 *   terms[(b*T + t)*4 + {0,1,2,3}] = {transition, emission, init, entropy} of step (b,t)
 * (the caller adds log p(s) - log q(s) and divides by mask.sum().clamp(1), :392-400).
 * _safe_cholesky (kalman_filter.py:282-302) is reproduced as a whole-batch jitter level resolved
 * on the device: chol_levels[0] (Sigma_s) and chol_levels[1] (Q_t) receive the first level
 * 0..4 (jitter 1e-6 * 10^level) at which every factorisation succeeds, or 5 = diagonal fallback.
 * If `g` / g_mus / g_Sigmas are non-NULL the gradients of SUM(terms) w.r.t. mus_smooth,
 * Sigmas_smooth and the problem inputs are written as well (unit upstream; the caller scales).
 * eps: [B,T,n] standard normal draws. chol_levels: 3 ints of device memory (zeroed by the call); [2] receives the kernel family
 * the main launch ran (0 wave-per-step generic, 1 thread-per-step (4,4,2), 2 / 3 the (16,16,2) matrix-core kernels with one /
 * four steps per wavefront) - all families take every level.
 * ws_lz: optional scratch [B,T,n]: the probe launch parks its level-0 sample z_t there so that the main launch does not
 * re-factorise the neighbouring steps (NULL = always recompute). */
int kvae_lgssm_elbo(const kvae_lgssm_problem *prob, const float *mus_smooth, const float *Sigmas_smooth,
                    const float *eps, float *terms, int32_t *chol_levels, float *ws_lz, float *g_mus,
                    float *g_Sigmas, const kvae_lgssm_input_grads *g, void *stream);

/* ---- mixture-of-K dynamics ---------------------------------------------------------------- */

/* out[r, :] = sum_k alpha[r, k] * base[k, :] for r < rows (= B*T), E = row length, K <= 16.
 * Replaces the einsums of dyn_param.py:58-60 and switch_dyn_param.py:82-84; the host side packs
 * A|B|C (lstm) or A|B|Q (switching) into ONE [K,E] base so that one launch mixes a whole step record. */
int kvae_mix_fwd(const float *alpha, const float *base, float *out, int64_t rows, int32_t K, int32_t E,
                 void *stream);

/* g_alpha[r,k] (+)= <g_out[r,:], base[k,:]>;  g_base[k,:] = sum_r alpha[r,k] g_out[r,:].
 * accumulate_alpha != 0 adds into g_alpha (several mixed operands share one alpha).
 * partials: scratch of kvae_mix_bwd_partials(rows) * K * E floats (deterministic two-stage sum). */
int kvae_mix_bwd(const float *alpha, const float *base, const float *g_out, float *g_alpha, float *g_base,
                 float *partials, int64_t rows, int32_t K, int32_t E, int32_t accumulate_alpha, void *stream);
int64_t kvae_mix_bwd_partials(int64_t rows);

/* ---- alpha-network recurrence ("lstm" dynamics) -------------------------------------------- */

/* Single-layer LSTM over a whole batch of sequences from a zero state, torch gate order (i,f,g,o):
 * replaces the T single-step nn.LSTM calls of DynamicsParameter.compute_step (dyn_param.py:50-52)
 * when every frame is observed (its input at step t is then a_{t-1}, kalman_filter.py:142,183-185).
 * x [B,T,I]; w_ih [4H,I]; w_hh [4H,H]; b_ih, b_hh [4H]; outputs h_seq [B,T,H], gates [B,T,4H]
 * (post-activation, kept for the backward), c_seq [B,T,H].  Limits: H <= 52, I <= 16. */
int kvae_lstm_fwd(const float *x, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                  float *h_seq, float *gates, float *c_seq, int32_t B, int32_t T, int32_t I, int32_t H, void *stream);

/* BPTT of kvae_lstm_fwd: from g_h [B,T,H] (gradient w.r.t. h_seq) produce d_pre [B,T,4H] (gradient
 * w.r.t. the pre-activation gates) and dx [B,T,I].  The parameter gradients follow as plain GEMMs:
 * dW_hh = d_pre^T h_{t-1}, dW_ih = d_pre^T x, db_ih = db_hh = sum d_pre. */
int kvae_lstm_bwd(const float *g_h, const float *gates, const float *c_seq, const float *w_ih, const float *w_hh,
                  float *d_pre, float *dx, int32_t B, int32_t T, int32_t I, int32_t H, void *stream);

/* ---- regime chain of the switching dynamics -------------------------------------------------- */

/* Sequential Gumbel-softmax Markov chain over T steps (switch_dyn_param.py:52-79): logits [B,T,K,K] (slice t=0
 * unused), init_logits [B,K], gumbel noise [B,T,K], prior transition P [K,K], temperature tau, hard != 0 for the
 * straight-through one-hot of eval mode.  Outputs y_seq [B,T,K], log_q [B,T], log_p [B,T].  K <= 16.
 * tau_dev (may be NULL): DEVICE scalar holding the temperature; when given it is read by the kernel at run time and
 * `tau` is ignored, so that a launch captured into a hipGraph follows the tau schedule of the reference's epoch loop
 * (kvae/train/train.py:270-274) without re-capture. */
int kvae_regime_fwd(const float *logits, const float *init_logits, const float *gumbel, const float *P, float *y_seq,
                    float *log_q, float *log_p, int32_t B, int32_t T, int32_t K, float tau, const float *tau_dev,
                    int32_t hard, void *stream);
/* BPTT of kvae_regime_fwd: upstream g_y [B,T,K], g_log_q [B,T], g_log_p [B,T] -> g_logits [B,T,K,K], g_init [B,K]. */
int kvae_regime_bwd(const float *logits, const float *init_logits, const float *gumbel, const float *P,
                    const float *y_seq, const float *g_y, const float *g_log_q, const float *g_log_p, float *g_logits,
                    float *g_init, int32_t B, int32_t T, int32_t K, float tau, const float *tau_dev, void *stream);

/* ---- bidirectional GRU of the regime posterior ("switching" dynamics) ------------------------- */

/* nn.GRU(I -> H, bidirectional, batch_first) from zero states (switch_dyn_param.py:118,123): x [B,T,I]; per direction
 * d in {0 forward, 1 reverse} w_ih[d] [3H,I], w_hh[d] [3H,H], b_ih[d], b_hh[d] [3H] (torch gate order r,z,n).
 * Outputs h_seq [B,T,2H] and gates [2,B,T,4H] = (r,z,n,W_hn h + b_hn) kept for the backward.
 * Only (H, I) = (50, 2) is built (KVAEConfig defaults); other shapes return KVAE_ERR_DIMS. */
int kvae_bigru_fwd(const float *x, const float *const w_ih[2], const float *const w_hh[2], const float *const b_ih[2],
                   const float *const b_hh[2], float *h_seq, float *gates, int32_t B, int32_t T, int32_t I, int32_t H,
                   void *stream);
/* BPTT: g_h [B,T,2H] -> d_pre_i, d_pre_h [2,B,T,3H] and dx [2,B,T,I] (sum the two directions for dL/dx).
 * Parameter gradients are GEMMs: dW_ih[d] = d_pre_i[d]^T x, dW_hh[d] = d_pre_h[d]^T h_prev[d], biases = column sums. */
int kvae_bigru_bwd(const float *g_h, const float *gates, const float *h_seq, const float *const w_ih[2],
                   const float *const w_hh[2], float *d_pre_i, float *d_pre_h, float *dx, int32_t B, int32_t T,
                   int32_t I, int32_t H, void *stream);

/* ---- fused conv epilogues of the frame VAE ------------------------------------------------- */

/* out[N,C,H*r,W*r] = act(pixel_shuffle_r(in[N,C*r*r,H,W] + bias[C*r*r])), act = ReLU if relu != 0, r in {1,2,..}:
 * replaces the separate bias-add / nn.PixelShuffle / nn.ReLU passes after each conv of the reference's
 * Encoder / Decoder (kvae/vae/vae.py:20-31, 92-101). */
int kvae_bias_shuffle_act_fwd(const float *in, const float *bias, float *out, int64_t N, int32_t C, int32_t H,
                              int32_t W, int32_t r, int32_t relu, void *stream);
/* g_in = pixel_unshuffle_r(g_out * [out > 0]).  If bias_partials != NULL ([kvae_bias_partial_rows(N), C*r*r] floats,
 * zeroed by the call) it receives per-sample-chunk partial sums of g_in over (n,h,w): the bias gradient is their
 * column sum. */
int kvae_bias_shuffle_act_bwd(const float *g_out, const float *out, float *g_in, float *bias_partials, int64_t N,
                              int32_t C, int32_t H, int32_t W, int32_t r, int32_t relu, void *stream);
int64_t kvae_bias_partial_rows(int64_t N);
/* out[c] = sum_r partials[r, c]: the second stage of the partial-row reductions above and below. */
int kvae_colsum(const float *partials, float *out, int64_t rows, int64_t cols, void *stream);
/* Two independent column sums in ONE launch (a layer's weight- and bias-gradient partials: the training step has eleven such
 * pairs, and a launch of a few microseconds of work costs as much again in the stream). */
int kvae_colsum2(const float *partials_a, float *out_a, int64_t rows_a, int64_t cols_a, const float *partials_b, float *out_b,
                 int64_t rows_b, int64_t cols_b, void *stream);

/* ---- optimizer step on flat buffers (reference train.py:52-56: clip_grad_norm_ + Adam.step) ------------------- */

/* One training step's tail on four flat fp32 buffers of n elements (parameters, gradients, Adam's exp_avg and exp_avg_sq):
 *   g <- g / max(*div_dev, 1)  (div_dev may be NULL: the multi-rank frame count);  total = ||g||_2  -> *norm_out (may be NULL);
 *   g <- g * min(1, clip / (total + 1e-6))  if clip > 0   (torch.nn.utils.clip_grad_norm_);
 *   step += 1;  Adam exactly as torch's fused kernel computes it (L2 weight decay added to g, exp_avg lerp,
 *   step_size = lr / (1 - beta1^step), denom = sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps).
 * Segments = parameter tensors: seg_of[i] in [0, n_seg) names the tensor of element i (NULL: one segment, n_seg == 1);
 * seg_steps[n_seg] are the per-parameter step counts torch's Adam keeps; seg_active[n_seg] (NULL: all active) is 0 for a FROZEN
 * parameter - requires_grad False in the reference's training phases (train.py:142-207), i.e. .grad is None: it does not enter
 * the norm, and neither its moments, its step count nor its values change, exactly as clip_grad_norm_ and Adam skip it.
 * lr is read from *lr_dev when given (a device scalar follows the LR schedule under hipGraph replay), else from `lr`.
 * The gradient buffer is left unscaled (it is overwritten by the next step).  ws: >= 1024 floats; n_seg <= 1024.  Two launches,
 * fixed summation order. */
int kvae_clip_adam(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, const int32_t *seg_of,
                   int32_t n_seg, const float *seg_active, float *seg_steps, const float *lr_dev, float lr, float beta1, float beta2,
                   float eps, float weight_decay, float clip, const float *div_dev, float *norm_out, float *ws, void *stream);

/* ---- imputation read-out (kvae/model/model.py:279-288) --------------------------------------------------- */

/* a_imputed[b,t,:] = C_t mu_{t|T},  a_filtered[b,t,:] = C_t mu_{t|t}  (both [B,T,p]; either output may be NULL together with its
 * input) with C_t the emission stack of the problem `prob` the means came from: what KVAE.impute decodes.  Only prob->B, T, n,
 * p and prob->C are read. */
int kvae_lgssm_emission_means(const kvae_lgssm_problem *prob, const float *mus_smooth, const float *mus_filt, float *a_imputed,
                              float *a_filtered, void *stream);

/* ---- parameter gradients of the alpha-network recurrences and the linear heads (no library GEMM on the path) ---- */

/* G[r][c] = sum_q d[q][r] * X[q][c] over the N = B*T rows, X[q] = [ h[q + shift] (H columns; zero when step t + shift leaves
 * the sequence) | x[q] (I columns) | 1 (when bias) ]: dW_hh | dW_ih | db of nn.LSTM / one nn.GRU direction from the d_pre rows
 * its BPTT kernel wrote (the autograd of kvae/kalman/dyn_param.py:50-56 and switch_dyn_param.py:113-129 in the reference), or
 * dW | db of a linear head (shift 0).  Outputs (each may be NULL): g_wh [R,H], g_wx [R,I], g_b [R].  R <= 256,
 * H + I + bias <= 256, N a multiple of T. */
typedef struct kvae_wgrad_problem {
  const float *d;            /* [N,R], row stride d_stride (floats) */
  const float *h;            /* hidden sequence, row stride h_stride; NULL when H == 0 */
  const float *x;            /* inputs, row stride x_stride; NULL when I == 0 */
  float *g_wh, *g_wx, *g_b;
  int64_t d_stride, h_stride, x_stride, N;
  int32_t R, H, I, bias, T, shift;   /* shift: -1 = h_{t-1}, +1 = h_{t+1} (reverse direction), 0 = h_t */
} kvae_wgrad_problem;
/* Up to four problems in ONE pair of launches (f32 matrix cores, split over the rows; a second launch sums the partials in a
 * fixed order).  ws: kvae_rnn_wgrad_ws_floats(probs, n) floats. */
int64_t kvae_rnn_wgrad_ws_floats(const kvae_wgrad_problem *probs, int32_t n);
int kvae_rnn_wgrad(const kvae_wgrad_problem *probs, int32_t n, float *ws, void *stream);

/* y[N,O] = x[N,F] W[O,F]^T + b (b may be NULL); softmax != 0: softmax over the O <= 16 outputs fused (head_w + softmax,
 * dyn_param.py:53-56).  x rows are x_stride floats apart.  F <= 128, O*F <= 12288. */
int kvae_linear_fwd(const float *x, int64_t x_stride, int64_t N, int32_t F, const float *W, const float *b, int32_t O,
                    int32_t softmax, float *y, void *stream);
/* dx[N,F] = gl W with gl = g, or - when y (the softmax output of the forward) is given - gl = y * (g - <g, y>), which is then
 * also written to g_logit [N,O] (the rows kvae_rnn_wgrad reduces to dW, db).  dx rows are dx_stride floats apart. */
int kvae_linear_bwd_input(const float *g, const float *y, int64_t N, int32_t F, const float *W, int32_t O, float *g_logit,
                          float *dx, int64_t dx_stride, void *stream);

/* ---- fused Bernoulli reconstruction term of the frame VAE --------------------------------- */

/* frame_ll[f] = -sum_{pixels} BCEWithLogits(logits[f,:], x[f,:]) for f < frames (kvae/vae/losses.py:85-87). */
int kvae_bce_frames_fwd(const float *logits, const float *x, float *frame_ll, int64_t frames, int32_t pixels, void *stream);
/* g_logits[f,:] = -g_frame[f] * (sigmoid(logits[f,:]) - x[f,:]). */
int kvae_bce_frames_bwd(const float *logits, const float *x, const float *g_frame, float *g_logits, int64_t frames,
                        int32_t pixels, void *stream);

/* ---- direct convolutions for the two thin layers of the frame VAE --------------------------- */

/* Decoder head (kvae/vae/vae.py:103-104): logits[N,1,2s,2s] = pixel_shuffle_2(conv3x3_pad1(in[N,Cin,s,s], W[4,Cin,3,3]) + b[4]).
 * Built for Cin = 32, s = 16 (the reference's default decoder); other shapes return KVAE_ERR_DIMS and the caller
 * keeps the library convolution. */
#define KVAE_DEC_HEAD_SCRATCH_FLOATS 2304 /* caller-owned scratch: the weights re-laid for scalar loads */
int kvae_dec_head_fwd(const float *in, const float *W, const float *bias, float *logits, float *w_scratch, int64_t N,
                      int32_t Cin, int32_t side, void *stream);
/* g_in[N,Cin,s,s] (data gradient; may be NULL to skip), w_partials [kvae_conv_edge_partial_rows(N), 4*Cin*9] and
 * b_partials [rows, 4]: the weight / bias gradients are their column sums. */
int kvae_dec_head_bwd(const float *in, const float *W, const float *g_logits, float *g_in, float *w_partials,
                      float *b_partials, float *w_scratch, int64_t N, int32_t Cin, int32_t side, void *stream);
/* Encoder stem (kvae/vae/vae.py:20-31): out[N,Cout,s/2,s/2] = relu(conv3x3_stride2_pad1(x[N,1,s,s], W[Cout,1,3,3]) + b).
 * Built for Cout = 32, s = 32.  relu_bits (may be NULL): [N, (s/2)^2] words, bit co of word (n, pixel) = out[n,co,pixel] > 0 -
 * the ReLU mask in 1/32 of the bytes of out, for kvae_enc_stem_bwd. */
int kvae_enc_stem_fwd(const float *x, const float *W, const float *bias, float *out, uint32_t *relu_bits, int64_t N, int32_t Cout,
                      int32_t side, void *stream);
/* Weight / bias gradient partials ([rows, Cout*9], [rows, Cout]) of the stem with the ReLU mask fused; the input frames need
 * no gradient.  The mask comes from relu_bits when given (then `out` is not read and may be NULL), else from out > 0. */
int kvae_enc_stem_bwd(const float *x, const float *out, const uint32_t *relu_bits, const float *g_out, float *w_partials,
                      float *b_partials, int64_t N, int32_t Cout, int32_t side, void *stream);
int64_t kvae_conv_edge_partial_rows(int64_t N);

/* Encoder middle layers (kvae/vae/vae.py:20-31): out[N,32,s/2,s/2] = relu(conv3x3_stride2_pad1(in[N,32,s,s], W[32,32,3,3]) + b)
 * on the f32 matrix cores.  Built for C = 32 and s in {16, 8}; other shapes return KVAE_ERR_DIMS. */
int kvae_enc_mid_fwd(const float *in, const float *W, const float *bias, float *out, int64_t N, int32_t C, int32_t side,
                     void *stream);
/* g_in[N,32,s,s] (may be NULL) = data gradient of g_out * (out > 0); w_partials [rows, 32*32*9] and
 * b_partials [rows, 32] with rows = kvae_enc_mid_partial_rows(N, side): gradients are the column sums. */
int kvae_enc_mid_bwd(const float *in, const float *W, const float *out, const float *g_out, float *g_in,
                     float *w_partials, float *b_partials, int64_t N, int32_t C, int32_t side, void *stream);
int64_t kvae_enc_mid_partial_rows(int64_t N, int32_t side);

/* Decoder up-sampling blocks (kvae/vae/vae.py:92-101):
 * out[N,32,2s,2s] = relu(pixel_shuffle_2(conv3x3_pad1(x[N,32,s,s], W[128,32,3,3]) + b[128])) on the f32 matrix cores,
 * weights stationary in registers.  Built for Cin = 32 and s in {8, 4}; other shapes return KVAE_ERR_DIMS. */
int kvae_dec_up_fwd(const float *x, const float *W, const float *bias, float *out, int64_t N, int32_t Cin, int32_t side,
                    void *stream);
/* g_x[N,32,s,s] (may be NULL) = data gradient of g_out * (out > 0) (g_out, out in the shuffled [N,32,2s,2s] layout);
 * w_partials [rows, 128*32*9], b_partials [rows, 128], rows = kvae_dec_up_partial_rows(N, side): column sums. */
int kvae_dec_up_bwd(const float *x, const float *W, const float *out, const float *g_out, float *g_x, float *w_partials,
                    float *b_partials, int64_t N, int32_t Cin, int32_t side, void *stream);
int64_t kvae_dec_up_partial_rows(int64_t N, int32_t side);
/* The decoder blocks run as persistent workgroups, one per CU (256).  A caller that overlaps them with a second stream whose
 * kernels cannot share a CU with them (kvae/train/train.py: the LGSSM chain of the switching model) may ask for fewer, which
 * leaves CUs free while they run; partial_rows follows.  n outside [1, 256] restores the default; returns the previous value.
 * Process-wide: set it around the launches (or the graph capture) it is meant for.  No reference counterpart. */
int32_t kvae_dec_up_set_workgroups(int32_t n);

/* ---- skinny fully-connected ends of the frame VAE and the latent regulariser --------------- */

/* Encoder heads (kvae/vae/vae.py:33-41) + reparameterisation (kvae/model/model.py:81-84), F = 512, A = 2:
 * mu = feat Wmu^T + bmu; var = noise_emission * sigmoid(feat Wvar^T + bvar); a = mu + eps * sqrt(var + 1e-6)
 * (eps may be NULL: a = mu).  Other shapes return KVAE_ERR_DIMS. */
int kvae_enc_head_fwd(const float *feat, const float *Wmu, const float *bmu, const float *Wvar, const float *bvar,
                      const float *eps, float *mu, float *var, float *a, int64_t N, int32_t F, int32_t A,
                      float noise_emission, void *stream);
/* Upstream g_a, g_mu, g_var ([N,A]; each may be NULL = 0) -> g_feat [N,F]; w_partials [rows, 2*A*F] laid out
 * (Wmu | Wvar) and b_partials [rows, 2*A] (bmu | bvar), rows = kvae_head_partial_rows(): column sums. */
int kvae_enc_head_bwd(const float *feat, const float *Wmu, const float *Wvar, const float *var, const float *eps,
                      const float *g_a, const float *g_mu, const float *g_var, float *g_feat, float *w_partials,
                      float *b_partials, int64_t N, int32_t F, int32_t A, float noise_emission, void *stream);
/* Decoder fc (kvae/vae/vae.py:88-90): h[N,F] = a[N,A] W[F,A]^T + b[F], and its gradients (partial rows as above:
 * w_partials [rows, F*A], b_partials [rows, F]). */
int kvae_dec_fc_fwd(const float *a, const float *W, const float *b, float *h, int64_t N, int32_t F, int32_t A, void *stream);
int kvae_dec_fc_bwd(const float *g_h, const float *a, const float *W, float *g_a, float *w_partials, float *b_partials,
                    int64_t N, int32_t F, int32_t A, void *stream);
int64_t kvae_head_partial_rows(void);
/* reg[n] = sum_j log N(a_nj; 0, 1) - log N(a_nj; mu_nj, var_nj) (kvae/vae/losses.py:64-66) and its three gradients. */
int kvae_latent_reg_fwd(const float *a, const float *mu, const float *var, float *reg, int64_t N, int32_t A, void *stream);
int kvae_latent_reg_bwd(const float *a, const float *mu, const float *var, const float *g, float *g_a, float *g_mu,
                        float *g_var, int64_t N, int32_t A, void *stream);

/* The scalar head of the objective (kvae/vae/losses.py:45-69, kvae/model/model.py:214-232) over n = B*T frames:
 * recon = sum(lpx*mk)/denom, reg = sum(regf*mk)/denom, denom = max(sum mk, 1) (mask NULL = all ones);
 * out6 = (loss, elbo_total, elbo_kf, vae_elbo, recon, reg), loss = -(vae_weight*(scale*recon + beta*reg) + kf_weight*elbo_kf);
 * coef3 = per-frame d loss/d lpx, d loss/d regf, and kf_weight (for the backward).  elbo_kf, beta: device scalars.
 * weights_dev (may be NULL): device scalars (vae_weight, kf_weight) that replace the by-value weights - the reference's training
 * phases change kf_weight between epochs (train.py:246-260) and a step captured into a hipGraph follows the device values. */
int kvae_loss_head_fwd(const float *lpx, const float *regf, const float *mask, const float *elbo_kf, const float *beta,
                       float scale_reconstruction, float vae_weight, float kf_weight, const float *weights_dev, float *out6,
                       float *coef3, int64_t n, void *stream);
int kvae_loss_head_bwd(const float *g_loss, const float *coef3, const float *mask, float *g_lpx, float *g_regf,
                       float *g_elbo_kf, int64_t n, void *stream);

/* ---- misc --------------------------------------------------------------------------------- */
int kvae_abi_version(void);
const char *kvae_last_error(void); /* text of the last KVAE_ERR_LAUNCH on this thread */
const char *kvae_build_info(void); /* "gfx950 ..." */

#ifdef __cplusplus
}
#endif
#endif /* KVAE_LGSSM_H */
