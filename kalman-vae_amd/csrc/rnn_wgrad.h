// rnn_wgrad.h — parameter gradients of the recurrent alpha-networks and their linear heads as ONE reduction kernel family:
//
//     G[r][c] = sum_q D[q][r] * X[q][c],      q over the N = B*T (sequence, step) rows,
//     X[q]    = [ h[q + shift] (H columns, zero outside the sequence) | x[q] (I columns) | 1 (bias column) ]
//
// i.e. dW_hh | dW_ih | db of an LSTM / GRU direction from the d_pre rows its BPTT kernel wrote (reference: the autograd of
// nn.LSTM in kvae/kalman/dyn_param.py:50-56 and of nn.GRU + the two heads in kvae/kalman/switch_dyn_param.py:113-129), and
// dW | db of a linear head.  Round 2 left these to rocBLAS: 2 x 150 us per step for a 200 x 50 output with a 12800-deep
// reduction (one workgroup walks the whole K dimension) - the largest GPU-time row of the configs[1] step.
//
// Mapping (gfx950, v_mfma_f32_16x16x4_f32 - exact fp32, a k-ordered fmaf chain): the reduction index q is the MFMA K dimension,
// 16 gradient rows x 16 gradient columns per accumulator tile.  A workgroup (4 wavefronts) owns one chunk of q (split-K over
// the grid) and up to four 16-column tiles, one per wavefront; a wavefront keeps ALL row tiles of its column tile in accumulators
// (<= 16 tiles = 64 VGPRs); the operands reach the matrix cores through LDS, staged QB rows at a time by the whole workgroup.
// Stage two sums the chunk partials in a FIXED order (run-to-run identical) and scatters the columns to dW_h | dW_x | db.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kvae {

constexpr int WG_MAX_ROW_TILES = 16;    // R <= 256 gradient rows
constexpr int WG_MAX_CHUNKS = 256;      // split-K partials (one workgroup per CU)

struct WgradProblem {
  const float *d;      // [N, R] rows of d_pre / g_logit, row stride d_stride
  const float *h;      // hidden sequence (may be NULL: H = 0), row stride h_stride; X[q][c<H] = h[q + shift][c]
  const float *x;      // inputs (may be NULL: I = 0), row stride x_stride
  float *g_wh, *g_wx, *g_b;   // outputs [R,H], [R,I], [R] (each may be NULL)
  int64_t d_stride, h_stride, x_stride;
  int64_t N;           // rows
  int32_t R, H, I, bias;      // bias: 1 = append the column of ones
  int32_t T, shift;    // sequence length and time shift of the hidden operand (-1: h_{t-1}, +1: h_{t+1}, 0: h_t)
};
struct WgradBatch {
  WgradProblem p[4];
  int32_t n;
};

__device__ __forceinline__ float wg_x_operand(const WgradProblem &P, int64_t q, int c) {
  if (q >= P.N) return 0.f;
  if (c < P.H) {
    const int t = (int)(q % P.T) + P.shift;
    return (t >= 0 && t < P.T) ? P.h[(q + P.shift) * P.h_stride + c] : 0.f;
  }
  if (c < P.H + P.I) return P.x[q * P.x_stride + (c - P.H)];
  return (P.bias && c == P.H + P.I) ? 1.0f : 0.f;
}

// Padded LDS row lengths: a wavefront's operand read is 16 consecutive floats of 4 consecutive rows, conflict-free when the
// row stride is 16 (mod 64 banks).
template <int RT> struct WgLds {
  static constexpr int RP = RT <= 1 ? 16 : (RT <= 4 ? 80 : (RT <= 13 ? 208 : 272));
  static constexpr int XP = 80, QB = 32;   // QB rows of D / X per stage
};

// grid = (chunks, column groups of 4 tiles, problems); block = 256.  partials[problem][chunk][R][Cpad] with Cpad = 16 * col tiles.
// Stage loop: the workgroup fetches the next QB rows of D and X into registers while the four wavefronts multiply the current
// QB rows out of LDS; a barrier pair swaps.  What the profile taught (profiles/r03_rnn_probe_*.txt, 200 x 53 at 12800 rows):
//   one wavefront per column tile with one k-step of register prefetch, no LDS                48 us (load latency per k-step)
//   LDS staging, element (row, col) = f(i * 256 + tid), `cond ? load : 0`                     41 us (700 basic blocks: a branch
//                                                                                                   and a wait per load; 64-bit
//                                                                                                   offsets and q % T per element)
//   unconditional loads + a guard per MFMA                                                     35 us (branch + LDS wait per MFMA)
//   all RT tiles unconditionally, operands of k-step ks+1 read under the MFMAs of ks           26 us (~800 VALU per stage)
//   this version: thread = one D column (rows walk by a stride add, LDS offsets are immediates), no masks outside the last
//   stage (padding columns only feed padding rows; rows past the chunk are killed by a zero X operand), step index by add
template <int RT>
__global__ __launch_bounds__(256) void k_rnn_wgrad_partial(const WgradBatch batch, float *__restrict__ partials,
                                                           int64_t partial_stride, int chunks) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  constexpr int RP = WgLds<RT>::RP, XP = WgLds<RT>::XP, QB = WgLds<RT>::QB;
  constexpr int XN = QB / 4;
  __shared__ float sD[QB * RP];
  __shared__ float sX[QB * XP];
  const WgradProblem &P = batch.p[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = P.R, T = P.T;
  const int C = P.H + P.I + (P.bias ? 1 : 0), col_tiles = (C + 15) / 16, row_tiles = (R + 15) / 16;
  const int ct = blockIdx.y * 4 + wave;
  const bool live = ct < col_tiles;                          // (dead wavefronts still help to load and keep the barriers)
  const int64_t per = ((P.N + chunks - 1) / chunks + 3) / 4 * 4;
  const int64_t q_lo = (int64_t)blockIdx.x * per, q_hi = q_lo + per < P.N ? q_lo + per : P.N;
  const int kk = lane >> 4, nn = lane & 15;
  f4 acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = (f4){0.f, 0.f, 0.f, 0.f};
  float dreg[QB], xreg[XN];
  // D: thread `tid` owns column min(tid, R - 1) of the stage tile (threads past the padded width idle); its QB rows are
  // d_stride apart.  Columns R .. 16 RT - 1 hold copies of column R - 1: they only reach accumulator rows that are never stored.
  const bool dthread = tid < RT * 16;
  const unsigned dcol = tid < R ? tid : R - 1, dstride = (unsigned)P.d_stride;
  // X: element i of this thread is (row 4 i + wave, column xc); the column - and with it the source - never changes
  const int xc = blockIdx.y * 64 + lane;
  const int xkind = xc < P.H ? 0 : (xc < P.H + P.I ? 1 : ((P.bias && xc == P.H + P.I) ? 2 : 3));
  const float *xsrc = xkind == 0 ? P.h + xc : (xkind == 1 ? P.x + (xc - P.H) : P.d);
  const unsigned xstride = xkind == 0 ? (unsigned)P.h_stride : (xkind == 1 ? (unsigned)P.x_stride : 0u);
  const int xshift = xkind == 0 ? P.shift : 0;
  const int bad_t = xshift < 0 ? 0 : (xshift > 0 ? T - 1 : -1);   // the step whose shifted neighbour leaves the sequence
  int tx[XN];                                                // step index of this thread's X rows, advanced by QB % T per stage
#pragma unroll
  for (int i = 0; i < XN; ++i) tx[i] = (int)((q_lo + i * 4 + wave) % T);
  const int qb_mod_t = QB % T;
  auto fetch = [&](int64_t q0) {
    const int rows_left = (int)(q_hi - q0);
    const float *dbase = P.d + q0 * P.d_stride;
    // (q0 + xshift) * stride, never before the tensor: the first row of a sequence is masked by bad_t and reads row q0 itself
    const float *xbase = xsrc + q0 * (int64_t)xstride;
    if (rows_left >= QB) {                                   // uniform: every stage but the last
      if (dthread) {
        unsigned off = dcol;
#pragma unroll
        for (int i = 0; i < QB; ++i, off += dstride) dreg[i] = dbase[off];
      }
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        const int row = i * 4 + wave;
        const bool ok = xkind <= 1 && tx[i] != bad_t;
        const float v = xbase[ok ? (row + xshift) * (int)xstride : 0]   /* signed: row + xshift may be -1 */;
        xreg[i] = ok ? v : (xkind == 2 ? 1.0f : 0.f);
      }
    } else {                                                 // last stage of the chunk: rows past q_hi read row 0, X = 0 kills them
      if (dthread) {
#pragma unroll
        for (int i = 0; i < QB; ++i) dreg[i] = dbase[(i < rows_left ? (unsigned)i * dstride : 0u) + dcol];
      }
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        const int row = i * 4 + wave;
        const bool in = row < rows_left, ok = in && xkind <= 1 && tx[i] != bad_t;
        const float v = xbase[ok ? (row + xshift) * (int)xstride : 0]   /* signed: row + xshift may be -1 */;
        xreg[i] = ok ? v : ((xkind == 2 && in) ? 1.0f : 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < XN; ++i) {
      tx[i] += qb_mod_t;
      tx[i] -= tx[i] >= T ? T : 0;
    }
  };
  auto stash = [&]() {
    if (dthread) {
#pragma unroll
      for (int i = 0; i < QB; ++i) sD[i * RP + tid] = dreg[i];
    }
#pragma unroll
    for (int i = 0; i < XN; ++i) sX[(i * 4 + wave) * XP + lane] = xreg[i];
  };
  (void)row_tiles;
  if (q_lo < q_hi) fetch(q_lo);
  for (int64_t q0 = q_lo; q0 < q_hi; q0 += QB) {
    __syncthreads();                                         // the previous stage's reads are done
    stash();
    __syncthreads();
    if (q0 + QB < q_hi) fetch(q0 + QB);                      // in flight under the products below
    if (live) {
      // All RT tiles unconditionally (a guard per MFMA became a branch per MFMA with its LDS read waited for right in front of
      // it).  Operands of k-step ks + 1 are read while k-step ks is on the matrix core; sched_barrier keeps hipcc from sinking
      // the reads back next to their use.
      float av[2][RT], bv[2];
      auto rd = [&](int ks, int slot) {
        bv[slot] = sX[(ks * 4 + kk) * XP + wave * 16 + nn];
#pragma unroll
        for (int r = 0; r < RT; ++r) av[slot][r] = sD[(ks * 4 + kk) * RP + r * 16 + nn];
      };
      rd(0, 0);
#pragma unroll
      for (int ks = 0; ks < QB / 4; ++ks) {
        if (ks + 1 < QB / 4) rd(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks & 1][r], bv[ks & 1], acc[r], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (!live) return;
  // accumulator layout: lane (n = lane & 15, g = lane >> 4), register i holds tile[4 g + i][n]
  const int c = ct * 16 + nn;
  float *out = partials + (int64_t)blockIdx.z * partial_stride + (int64_t)blockIdx.x * R * (col_tiles * 16);
#pragma unroll
  for (int r = 0; r < RT; ++r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r * 16 + 4 * kk + i;
      if (row < R) out[(int64_t)row * (col_tiles * 16) + c] = acc[r][i];
    }
  }
}

// 32 gradient elements x 8 chunk lanes per block: lane j sums chunks j, j+8, ... (loads eight deep), the eight partial sums are
// folded in lane order through LDS - a fixed order, whatever the launch
__global__ __launch_bounds__(256) void k_rnn_wgrad_final(const WgradBatch batch, const float *__restrict__ partials,
                                                         int64_t partial_stride, int chunks) {
  __shared__ float red[8][33];
  const WgradProblem &P = batch.p[blockIdx.y];
  const int C = P.H + P.I + (P.bias ? 1 : 0), Cpad = (C + 15) / 16 * 16;
  const int el = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  const bool ok = e < P.R * C;
  const int row = ok ? e / C : 0, c = ok ? e % C : 0;
  const float *src = partials + (int64_t)blockIdx.y * partial_stride + (int64_t)row * Cpad + c;
  const int64_t cs = (int64_t)P.R * Cpad;
  float s = 0.f;
  for (int k0 = cl; k0 < chunks; k0 += 64) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (ok && k0 + 8 * u < chunks) ? src[(int64_t)(k0 + 8 * u) * cs] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  red[cl][el] = s;
  __syncthreads();
  if (cl != 0 || !ok) return;
  float t = red[0][el];
#pragma unroll
  for (int j = 1; j < 8; ++j) t += red[j][el];
  if (c < P.H) {
    if (P.g_wh) P.g_wh[(int64_t)row * P.H + c] = t;
  } else if (c < P.H + P.I) {
    if (P.g_wx) P.g_wx[(int64_t)row * P.I + (c - P.H)] = t;
  } else if (P.g_b) {
    P.g_b[row] = t;
  }
}

}  // namespace kvae
