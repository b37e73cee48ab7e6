// vae_conv_up_wino.h — the decoder's up-sampling blocks (Conv2d(32 -> 128, 3x3, pad 1) + PixelShuffle(2) + ReLU on 4x4 and
// 8x8 frames, reference kvae/vae/vae.py:92-101) as Winograd F(2x2, 3x3) on the exact-f32 matrix cores: 16 multiplies per
// 2x2 output tile and channel pair instead of 36, still fp32 throughout (the transforms only add, subtract and halve).
// The direct kernels of vae_conv_up.h sit at ~85 % of the f32 matrix rate, so fewer multiplies is the only way down.
//
//   Y = A^T [ (G g G^T) o (B^T d B) ] A      d: 4x4 input patch of the tile, g: 3x3 filter, o: element-wise over the 16 points,
//   summed over the input channels BEFORE the output transform: per point p one GEMM  M_p[co][tile] = U_p[co][ci] V_p[ci][tile].
//
// Mapping (v_mfma_f32_16x16x4_f32, D[16 co][16 tiles] += A[16 co][4 ci] B[4 ci][16 tiles]):
//   * a wave owns 16 output channels; its transformed filters U_p[co][ci] - 16 points x 8 k-steps = 128 registers per lane -
//     stay in registers for the whole (persistent) kernel, transformed once from the raw weights;
//   * the 16 columns are the 16 tiles of one 8x8 frame (or of four 4x4 frames); lane (j = lane & 15, g = lane >> 4) reads the
//     4x4 patch of (tile j, channel 4s + g) from a zero-bordered LDS image (eight ds_read_b64, compile-time offsets, no bank
//     conflicts), transforms it in registers (32 add/sub) and thereby HOLDS the B operands of all 16 points for k-step s;
//   * 16 accumulators of four registers; the C/D layout gives lane (j, g) the four channels 4g..4g+3 of tile j
//     = exactly one PixelShuffle output channel and its (dy, dx) sub-pixels: the output transform (24 add/sub per channel),
//     bias, shuffle and ReLU happen in registers and the 4x4 block of the final image leaves as four 16-byte stores.
#pragma once
#include <type_traits>
#include "vae_conv_up.h"

namespace kvae {

typedef float wn_f4 __attribute__((ext_vector_type(4)));
#ifdef KVAE_EM_STAMPS   // tools/wino_stamp*.hip only: s_memtime stamps of workgroup 0, every wave (waves w and w + 4 share a SIMD)
#define WN_STAMP(slot, k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (slot) < 40) \
    em_stamps[((threadIdx.x >> 6) * 40 + (slot)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WN_STAMP(slot, k) do {} while (0)
#endif
#define WN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// (Raising the issue priority of one wave per SIMD with s_setprio, to break a suspected lockstep of the two waves of a SIMD, was
//  measured: 2 % on the kernels alone, -1.5 % on the training step, where the side-stream kernels then wait longer.)

template <int S>
struct WinoDims {
  static constexpr int TS = S / 2;                       // tiles per side
  static constexpr int TPF = TS * TS;                    // 2x2-output tiles per frame (16 or 4)
  static constexpr int FPC = 16 / TPF;                   // frames per column set (1 or 4)
  static constexpr int RS = S == 8 ? 12 : 8;             // padded row length: 2 RS ty + 2 tx (+ 4 f) hit 16 distinct bank pairs
  static constexpr int PLANE = (S + 2) * RS;             // zero-bordered channel plane (120 or 48 floats)
  static constexpr int FS = 32 * PLANE + (S == 8 ? 0 : 4);   // frame stride (4x4: frames of a column set 4 banks apart)
  static constexpr int FPI = UpDims<S>::FPI;             // frames per iteration = 2 column sets
  static constexpr int LDS_IN = FPI * FS;
};

// rows of B^T d (and, applied again along the other axis, of (B^T d) B): (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
#define WN_BT(o0, o1, o2, o3, d0, d1, d2, d3) do { o0 = (d0) - (d2); o1 = (d1) + (d2); o2 = (d2) - (d1); o3 = (d1) - (d3); } while (0)

// U = G g G^T of one 3x3 filter (row-major g[9]) -> u[16], point p = 4 u + v
__device__ __forceinline__ void wino_filter(const float *g, float *u) {
  float r[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    r[0][c] = g[c];
    r[1][c] = 0.5f * (g[c] + g[3 + c] + g[6 + c]);
    r[2][c] = 0.5f * (g[c] - g[3 + c] + g[6 + c]);
    r[3][c] = g[6 + c];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    u[4 * a + 0] = r[a][0];
    u[4 * a + 1] = 0.5f * (r[a][0] + r[a][1] + r[a][2]);
    u[4 * a + 2] = 0.5f * (r[a][0] - r[a][1] + r[a][2]);
    u[4 * a + 3] = r[a][2];
  }
}

// A^T M A of one channel: m[p] (16 points) -> y[2a + b]
__device__ __forceinline__ void wino_out(const float (&m)[16], float (&y)[4]) {
  float t[2][4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    t[0][v] = m[v] + m[4 + v] + m[8 + v];
    t[1][v] = m[4 + v] - m[8 + v] - m[12 + v];
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    y[2 * a + 0] = t[a][0] + t[a][1] + t[a][2];
    y[2 * a + 1] = t[a][1] - t[a][2] - t[a][3];
  }
}

// the 4x4 patch of one (tile, channel) from the zero-bordered LDS image -> B^T d B, v[4 u + w]
__device__ __forceinline__ void wino_patch_load(const float *p, int RS, float2 (&d)[4][2]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    d[r][0] = *reinterpret_cast<const float2 *>(p + r * RS);
    d[r][1] = *reinterpret_cast<const float2 *>(p + r * RS + 2);
  }
}
__device__ __forceinline__ void wino_patch_xform(const float2 (&d)[4][2], float (&v)[16]) {
  float t[4][4];
  WN_BT(t[0][0], t[1][0], t[2][0], t[3][0], d[0][0].x, d[1][0].x, d[2][0].x, d[3][0].x);
  WN_BT(t[0][1], t[1][1], t[2][1], t[3][1], d[0][0].y, d[1][0].y, d[2][0].y, d[3][0].y);
  WN_BT(t[0][2], t[1][2], t[2][2], t[3][2], d[0][1].x, d[1][1].x, d[2][1].x, d[3][1].x);
  WN_BT(t[0][3], t[1][3], t[2][3], t[3][3], d[0][1].y, d[1][1].y, d[2][1].y, d[3][1].y);
#pragma unroll
  for (int u = 0; u < 4; ++u) WN_BT(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3], t[u][0], t[u][1], t[u][2], t[u][3]);
}

// ---------------------------------------------------------------------------------------------------------------
// forward.  A workgroup covers 64 of the 128 output channels (blockIdx & 1), a wave 16 of them: 128 filter registers + 64
// accumulator registers (32 channels per wave = 256 + 128 spilt into the loop); the two workgroups of a pair walk the same
// column sets (the second one reads the frames from L2).  The patch transform is SHARED by the four waves through LDS: each
// wave transforms two of the eight k-steps of the NEXT column set (vt[], rows of 16 points padded to 20 floats: conflict-free
// ds_read_b128) while all four run the 128 MFMAs of the current one, whose B operands are then plain 16-byte LDS reads.
// One barrier per column set; input frames and transformed operands are double-buffered.
// ---------------------------------------------------------------------------------------------------------------
template <int S>
struct WinoFwd {
  using D = UpDims<S>;
  using Wd = WinoDims<S>;
  static constexpr int CSX = Wd::FPC * Wd::FS;       // zero-bordered input of one column set (floats)
  static constexpr int CSG = Wd::FPC * D::XFRAME;    // its 2048 floats in global memory
  static constexpr int VROW = 20, VSET = 8 * 64 * VROW;
};

template <int S>
__global__ __launch_bounds__(512) void k_dec_up_fwd_wino(const float *__restrict__ x, const float *__restrict__ W,
                                                         const float *__restrict__ bias, float *__restrict__ out, int64_t N) {
  using D = UpDims<S>;
  using Wd = WinoDims<S>;
  using K = WinoFwd<S>;
  __shared__ float xin[2 * K::CSX];
  __shared__ __attribute__((aligned(16))) float vt[2 * K::VSET];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, g = lane >> 4;
  const int co0 = 16 * wv;                                        // this wave's 16 output channels (8 waves)
  const int64_t nsets = (N + Wd::FPC - 1) / Wd::FPC, stride = gridDim.x;
  const __amdgpu_buffer_rsrc_t rx = em_rsrc(x, N * D::XFRAME * 4), ry = em_rsrc(out, N * D::YFRAME * 4);
  for (int i = tid; i < 2 * K::CSX; i += 512) xin[i] = 0.f;      // the borders stay zero for the whole kernel

  float U[16][8];                                                 // A operands: U_p[co = co0 + j][ci = 4 s + g]
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float *w = W + ((co0 + j) * UP_CI + 4 * s + g) * 9;
    float gk[9], u[16];
#pragma unroll
    for (int t = 0; t < 9; ++t) gk[t] = w[t];
    wino_filter(gk, u);
#pragma unroll
    for (int p = 0; p < 16; ++p) U[p][s] = u[p];
  }
  float bv[4];                                                    // D rows of this lane: co = co0 + 4 g + r
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = bias[co0 + 4 * g + r];

  // this lane's tile inside a column set: frame fl (4x4: four frames per set), tile (ty, tx)
  const int fl = S == 8 ? 0 : (j >> 2), ty = S == 8 ? (j >> 2) : ((j >> 1) & 1), tx = S == 8 ? (j & 3) : (j & 1);
  const int pbase = fl * Wd::FS + g * Wd::PLANE + 2 * ty * Wd::RS + 2 * tx + wv * 4 * Wd::PLANE;   // k-step wv
  const int vbase_w = (wv * 64 + lane) * K::VROW, vbase_r = lane * K::VROW;
  // ... and where its 4x4 block of the final [32, 2S, 2S] image goes: channel co0 / 4 + g, rows 4 ty.., columns 4 tx..
  const int obase = fl * D::YFRAME + (co0 / 4 + g) * 4 * D::PF + 4 * ty * 2 * S + 4 * tx;
  // staging: two 16-byte pieces of the column set per thread
  // The two waves of a SIMD (wv, wv + 4) run the interval in OPPOSITE order - waves 0-3: staging + patch transform, then MFMAs;
  // waves 4-7: MFMAs first, the shared work last - so that one wave's VALU / LDS phase lies beside the other's MFMA phase
  // (both in the same order: 11.3k cycles per column set; shared work on waves 0-3 only: 10.8k; stamps, tools/wino_stamp.hip).
  const bool early = wv < 4;
  int sdst[1];
#pragma unroll
  for (int q = 0; q < 1; ++q) {
    const int e = (tid + 512 * q) * 4, f = e / D::XFRAME, ci = (e % D::XFRAME) / D::PF, pix = e % D::PF;
    sdst[q] = f * Wd::FS + ci * Wd::PLANE + (pix / S + 1) * Wd::RS + pix % S + 1;
  }

  float4 pre[1];
  auto fetch = [&](int64_t k) {
#pragma unroll
    for (int q = 0; q < 1; ++q) pre[q] = em_ld4(rx, (uint32_t)(k * K::CSG + (tid + 512 * q) * 4) * 4u);
  };
  auto stage = [&](float *xw) {
#pragma unroll
    for (int q = 0; q < 1; ++q) {
      float *d = xw + sdst[q];
      d[0] = pre[q].x; d[1] = pre[q].y; d[2] = pre[q].z; d[3] = pre[q].w;
    }
  };
  auto transform = [&](const float *xr, float *vw) {   // this wave's k-step of a column set
#pragma unroll
    for (int h = 0; h < 1; ++h) {
      float2 d[4][2];
      float v[16];
      wino_patch_load(xr + pbase + h * 4 * Wd::PLANE, Wd::RS, d);
      wino_patch_xform(d, v);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4 *>(vw + vbase_w + h * 64 * K::VROW + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
  };
  auto compute_p = [&](const float *vr, int64_t k) {
    float y[4][4];                                   // [r][2a + b]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) y[r][q] = 0.f;
    float2 bq[8], bn[8];                             // B operands of the pair (4u + 2h, 4u + 2h + 1), all eight k-steps
#pragma unroll
    for (int s = 0; s < 8; ++s) bq[s] = *reinterpret_cast<const float2 *>(vr + vbase_r + s * 64 * K::VROW);
    float sb[4][2];                                  // per output sub-pixel r: sum_v A^T[b][v] M[u][v] of the current row u
    wn_f4 p0 = wn_f4{0.f, 0.f, 0.f, 0.f}, p1 = p0;   // the finished pair that is being folded
    // fold of pair pq = (u, h), sub-pixel r.  A^T = (1 1 1 0 / 0 1 -1 -1): columns v = 0, 1 (h = 0) or 2, 3 (h = 1) of row u;
    // with the row complete (h = 1): y[a][b] += A^T[a][u] s_b
    auto fold = [&](int pq, int r) {
      const int u = pq >> 1, h = pq & 1;
      if (h == 0) sb[r][0] = p0[r] + p1[r], sb[r][1] = p1[r];
      else {
        sb[r][0] += p0[r], sb[r][1] -= p0[r] + p1[r];
        if (u == 0) y[r][0] += sb[r][0], y[r][1] += sb[r][1];
        if (u == 1) y[r][0] += sb[r][0], y[r][1] += sb[r][1], y[r][2] += sb[r][0], y[r][3] += sb[r][1];
        if (u == 2) y[r][0] += sb[r][0], y[r][1] += sb[r][1], y[r][2] -= sb[r][0], y[r][3] -= sb[r][1];
        if (u == 3) y[r][2] -= sb[r][0], y[r][3] -= sb[r][1];
      }
      // computed HERE: LLVM otherwise sinks these adds (their results are only read after the last MFMA) below every barrier
      asm volatile("" : "+v"(sb[r][0]), "+v"(sb[r][1]), "+v"(y[r][0]), "+v"(y[r][1]), "+v"(y[r][2]), "+v"(y[r][3]));
    };
    // Hand-placed order, pinned group by group (left alone hipcc issues all 128 MFMAs first and the folds in one block after
    // them): two MFMAs (one per chain), one operand read of the next pair, the fold of one sub-pixel of the previous pair every
    // other group - ~4 vector instructions in a 64-cycle MFMA gap.
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      wn_f4 a0 = wn_f4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        a0 = WN_MFMA(U[2 * pp][s], bq[s].x, a0);
        a1 = WN_MFMA(U[2 * pp + 1][s], bq[s].y, a1);
        if (pp + 1 < 8) bn[s] = *reinterpret_cast<const float2 *>(vr + vbase_r + s * 64 * K::VROW + 2 * (pp + 1));
        if (pp > 0 && (s & 1)) fold(pp - 1, s >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      p0 = a0, p1 = a1;
#pragma unroll
      for (int s = 0; s < 8; ++s) bq[s] = bn[s];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) fold(7, r);
    // register r = sub-pixel (dy, dx) = (r >> 1, r & 1) of shuffle channel co0 / 4 + g
    const uint32_t o0 = (uint32_t)(k * Wd::FPC * D::YFRAME + obase);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) y[r][q] = fmaxf(y[r][q] + bv[r], 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {               // image row 4 ty + 2a + dy: columns (b, dx) = (0,0) (0,1) (1,0) (1,1)
        const float4 row = make_float4(y[2 * dy][2 * a], y[2 * dy + 1][2 * a], y[2 * dy][2 * a + 1], y[2 * dy + 1][2 * a + 1]);
        em_st4(ry, (o0 + (uint32_t)((2 * a + dy) * 2 * S)) * 4u, row);
      }
  };

#define WN_COMPUTE(vr, k) compute_p((vr), (k))

  // pipeline: at step n, xin[(n+1)&1] holds set n+1, vt[n&1] the transformed set n, pre the frames of set n+2
  int64_t k = blockIdx.x;
  fetch(k);
  __syncthreads();                                   // zero fill done
  stage(xin);
  fetch(k + stride);
  __syncthreads();
  transform(xin, vt);
  stage(xin + K::CSX);
  fetch(k + 2 * stride);
  __syncthreads();
  for (int slot = 0; k < nsets; k += 2 * stride, ++slot) {
    WN_STAMP(slot, 0);
    if (early) {
      stage(xin);                                    // set n+2 (its buffer held set n, transformed one step ago)
      fetch(k + 3 * stride);
      transform(xin + K::CSX, vt + K::VSET);         // set n+1
    }
    WN_STAMP(slot, 1);
    WN_COMPUTE(vt, k);                               // set n
    WN_STAMP(slot, 2);
    if (!early) {
      stage(xin);
      fetch(k + 3 * stride);
      transform(xin + K::CSX, vt + K::VSET);
    }
    __syncthreads();
    WN_STAMP(slot, 3);
    if (early) {
      stage(xin + K::CSX);
      fetch(k + 4 * stride);
      transform(xin, vt);
    }
    WN_STAMP(slot, 4);
    WN_COMPUTE(vt + K::VSET, k + stride);
    WN_STAMP(slot, 5);
    if (!early) {
      stage(xin + K::CSX);
      fetch(k + 4 * stride);
      transform(xin, vt);
    }
    __syncthreads();
    WN_STAMP(slot, 6);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// data gradient: g_x[ci] = sum_co flip(W[co][ci]) (*) gy[co], gy = unshuffle(g_out * (out > 0)) - the same
// Winograd convolution with the roles of the channels swapped and K = 128.  Wave (hh, kq) = (wv & 1, wv >> 1) produces input
// channels 16 hh .. 16 hh + 15 from output channels 32 kq .. 32 kq + 31 (128 filter + 64 accumulator registers again); each wave
// transforms its own patches (the operand set of all 128 channels would not fit in LDS; only the two hh-waves of a kq repeat
// each other).  g_out and out reach LDS by DMA (buffer_load ... lds, no registers: there are none to spare) one column set
// ahead, in their own [32][16][16] layout; a short phase between two barriers turns them into masked, un-shuffled,
// zero-bordered planes, and folds the four K-quarters of the previous set (red[]) into one coalesced 16-byte store per thread.
// ---------------------------------------------------------------------------------------------------------------
// 4x4 layer: a column set is four frames; their planes keep the left / right border columns but NOT the top / bottom border
// rows (4 x 6 floats instead of 6 x 6: 128 channels x 4 frames must fit beside the DMA buffers) - those two patch rows are read
// from beyond the LDS allocation, which returns zeros (EM_OOB); frames sit 24 banks apart (16 distinct bank pairs per 16 lanes).
template <int S>
struct WinoBwd {
  static constexpr int RAW = 8192;                   // floats of g_out (and of out) per column set
  static constexpr int RS = S == 8 ? 12 : 6;
  static constexpr int PLANE = S == 8 ? 120 : 24;
  static constexpr int FSB = 128 * PLANE + (S == 8 ? 0 : 24);
  static constexpr int GY = WinoDims<S>::FPC * FSB;
  static constexpr int RED = 2048;                   // one K-quarter's partial g_x of a column set
};

template <int S>
__global__ __launch_bounds__(512) void k_dec_up_bwd_data_wino(const float *__restrict__ W, const float *__restrict__ out,
                                                              const float *__restrict__ g_out, float *__restrict__ g_x, int64_t N) {
  using D = UpDims<S>;
  using Wd = WinoDims<S>;
  using K = WinoBwd<S>;
  // two objects on purpose: the DMA target must be provably distinct from the arrays the MFMA phase reads, or hipcc drains
  // the DMA (s_waitcnt vmcnt(0)) in front of the first ds_read after it
  __shared__ __attribute__((aligned(16))) float raw[2 * K::RAW];
  __shared__ __attribute__((aligned(16))) float work[K::GY + 4 * K::RED];
  float *rawg = raw, *rawo = raw + K::RAW, *gy = work, *red = work + K::GY;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, g = lane >> 4, hh = wv & 1, kq = wv >> 1;
  const int64_t stride = gridDim.x, nsets = (N + Wd::FPC - 1) / Wd::FPC;
  const __amdgpu_buffer_rsrc_t rg = em_rsrc(g_out, N * D::YFRAME * 4), ro = em_rsrc(out, N * D::YFRAME * 4),
                               rgx = em_rsrc(g_x, N * D::XFRAME * 4);
  for (int i = tid; i < K::GY; i += 512) gy[i] = 0.f;             // the borders stay zero for the whole kernel

  float U[16][8];                                                 // A operands: flip(W)[co = 32 kq + 4 s + g][ci = 16 hh + j]
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float *w = W + ((32 * kq + 4 * s + g) * UP_CI + 16 * hh + j) * 9;
    float gk[9], u[16];
#pragma unroll
    for (int t = 0; t < 9; ++t) gk[t] = w[8 - t];
    wino_filter(gk, u);
#pragma unroll
    for (int p = 0; p < 16; ++p) U[p][s] = u[p];
  }
  // this lane's tile: frame fl of the set, tile (ty, tx); patch row r of k-step s starts at gy[prow[r] + s * 4 * PLANE]
  const int fl = S == 8 ? 0 : (j >> 2), ty = S == 8 ? (j >> 2) : ((j >> 1) & 1), tx = S == 8 ? (j & 3) : (j & 1);
  int prow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = S == 8 ? 2 * ty + r : 2 * ty - 1 + r;         // 8x8: planes carry their border rows; 4x4: rows -1 and 4 read zeros
    prow[r] = (S == 4 && (row < 0 || row > 3)) ? EM_OOB : fl * K::FSB + (32 * kq + g) * K::PLANE + row * K::RS + 2 * tx;
  }
  const int rbase = kq * K::RED + fl * D::XFRAME + (16 * hh + 4 * g) * D::PF + 2 * ty * S + 2 * tx;

  // (the staging phases recompute their addresses from an opaque copy of tid: hoisted out of the loop they would sit in
  //  registers through the MFMA phase, which has none to spare)
  // eight 8 KB pieces per column set (g_out and out, four each).  Issued back to back they hold the later waves of the workgroup
  // for ~3000 cycles (the memory queue is full: stamps, tools/wino_stamp_bwd.hip) before their first MFMA; dealt out one pair
  // per k-step over the first four k-steps they queue behind the partner wave's MFMAs instead.
  auto dma_piece = [&](int64_t k, int q) {
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    const uint32_t off = (uint32_t)(k * K::RAW + (t2 + 512 * q) * 4) * 4u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void *)(rawg + (wv * 64 + 512 * q) * 4), 16, off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ro, (__attribute__((address_space(3))) void *)(rawo + (wv * 64 + 512 * q) * 4), 16, off, 0, 0, 0);
  };
  auto dma = [&](int64_t k) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dma_piece(k, q);
  };
  auto convert = [&]() {                              // raw -> masked, un-shuffled planes with zero borders (co = 4 c + 2 dy + dx)
    int t2 = tid;
    asm volatile("" : "+v"(t2));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = (t2 + 512 * q) * 4;
      const float4 gv = *reinterpret_cast<const float4 *>(rawg + e), ov = *reinterpret_cast<const float4 *>(rawo + e);
      float *d;
      if constexpr (S == 8) {
        const int c = e >> 8, y = (e >> 4) & 15, xx = e & 15;
        d = gy + (4 * c + 2 * (y & 1)) * K::PLANE + ((y >> 1) + 1) * K::RS + (xx >> 1) + 1;
      } else {
        const int f = e >> 11, c = (e >> 6) & 31, y = (e >> 3) & 7, xx = e & 7;
        d = gy + f * K::FSB + (4 * c + 2 * (y & 1)) * K::PLANE + (y >> 1) * K::RS + (xx >> 1) + 1;
      }
      d[0] = ov.x > 0.f ? gv.x : 0.f;
      d[K::PLANE] = ov.y > 0.f ? gv.y : 0.f;
      d[1] = ov.z > 0.f ? gv.z : 0.f;
      d[K::PLANE + 1] = ov.w > 0.f ? gv.w : 0.f;
    }
  };
  uint32_t done = 0x80000000u;                        // byte offset of the set whose partials sit in red[] (none yet)
  auto fold = [&]() {
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    const float4 a = *reinterpret_cast<const float4 *>(red + 4 * t2), b = *reinterpret_cast<const float4 *>(red + K::RED + 4 * t2),
                 c = *reinterpret_cast<const float4 *>(red + 2 * K::RED + 4 * t2), d = *reinterpret_cast<const float4 *>(red + 3 * K::RED + 4 * t2);
    em_st4(rgx, done + (uint32_t)t2 * 16u, make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z),
                                                        (a.w + b.w) + (c.w + d.w)));
  };
  auto patch = [&](int s, float2 (&d)[4][2]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float *p = gy + prow[r] + s * 4 * K::PLANE;
      d[r][0] = *reinterpret_cast<const float2 *>(p);
      d[r][1] = *reinterpret_cast<const float2 *>(p + 2);
    }
  };
  auto compute = [&](int64_t knext) {
    wn_f4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = wn_f4{0.f, 0.f, 0.f, 0.f};
    float2 d[4][2], dn[4][2];
    patch(0, d);
#pragma unroll
    for (int s = 0; s < 8; ++s) {                     // the next patch is requested before this k-step's MFMAs (pinned, as above)
      if (s + 1 < 8) patch(s + 1, dn);
      float v[16];
      wino_patch_xform(d, v);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < 16; ++p) acc[p] = WN_MFMA(U[p][s], v[p], acc[p]);
      __builtin_amdgcn_sched_barrier(0);
      if (s < 4) dma_piece(knext, s);                 // the raw buffers are free since the convert phase
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) d[r][0] = dn[r][0], d[r][1] = dn[r][1];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                     // D row r: input channel 16 hh + 4 g + r; pixels (2 ty + a, 2 tx + b)
      float m[16], y[4];
#pragma unroll
      for (int p = 0; p < 16; ++p) m[p] = acc[p][r];
      wino_out(m, y);
      *reinterpret_cast<float2 *>(red + rbase + r * D::PF) = make_float2(y[0], y[1]);
      *reinterpret_cast<float2 *>(red + rbase + r * D::PF + S) = make_float2(y[2], y[3]);
    }
  };

  int64_t k = blockIdx.x;
  dma(k);
  __syncthreads();                                    // zero fill done, set k landed (the barrier's fence drains the DMA)
  for (int slot = 0; k < nsets; k += stride, ++slot) {
    WN_STAMP(slot, 0);
    convert();
    fold();                                           // previous set's four partial sums -> g_x
    WN_STAMP(slot, 1);
    __syncthreads();
    WN_STAMP(slot, 2);
    WN_STAMP(slot, 3);
    compute(k + stride);                              // ... and the next set's DMA, dealt out between its k-steps
    done = (uint32_t)(k * 2048) * 4u;
    WN_STAMP(slot, 4);
    __syncthreads();
    WN_STAMP(slot, 5);
  }
  fold();
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dW[co][ci] = G^T Z G,  Z_p[co][ci] = sum over frames and tiles of (A gy A^T)_p[co][tile] (B^T x B)_p[ci][tile]
// - the adjoint of the forward's element-wise product, so again 16 multiplies per (tile, channel pair) instead of 36.  The tiles
// are the K dimension now (four per MFMA); wave wv keeps Z for output channels 16 wv .. 16 wv + 15, all 32 input channels and all
// 16 points in 128 accumulator registers for the whole kernel and applies G^T . G once at the end.  Per frame: the x patches are
// transformed once for the whole workgroup (one patch per thread, through LDS, as in the forward); every wave expands its own
// 2x2 tiles of gy = unshuffle(g_out * (out > 0)), which a short phase between two barriers lays out so that a lane's tile is
// one ds_read_b128 (g_out / out arrive by DMA, x through 4 registers, one frame ahead).
// ---------------------------------------------------------------------------------------------------------------
template <int S>
struct WinoWrw {
  static constexpr int RAW = 8192;                   // floats of g_out (and of out) per column set (one 8x8 or four 4x4 frames)
  static constexpr int GY4 = 8192;                   // [ks 4][co 128][tile in group 4][2a + b]
  static constexpr int RS = S == 8 ? 12 : 6;         // row length of the zero-bordered x planes
  static constexpr int XP = S == 8 ? 130 : 38;       // plane stride: 16 channels -> 16 distinct bank pairs (130 = 2, 38 = 6 mod 32)
  static constexpr int XPL = WinoDims<S>::FPC * 32 * XP;
  static constexpr int VROW = 20, VSET = 8 * 64 * VROW;
};

template <int S>
__global__ __launch_bounds__(512) void k_dec_up_wrw_wino(const float *__restrict__ x, const float *__restrict__ out,
                                                         const float *__restrict__ g_out, float *__restrict__ w_partials,
                                                         float *__restrict__ b_partials, int64_t N) {
  using D = UpDims<S>;
  using Wd = WinoDims<S>;
  using K = WinoWrw<S>;
  __shared__ __attribute__((aligned(16))) float raw[2 * K::RAW];                        // DMA targets (own object: see bwd_data)
  __shared__ __attribute__((aligned(16))) float work[K::GY4 + K::XPL + K::VSET];
  float *rawg = raw, *rawo = raw + K::RAW, *gy4 = work, *xp = work + K::GY4, *vt = xp + K::XPL;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, g = lane >> 4;
  const int64_t stride = gridDim.x, nsets = (N + Wd::FPC - 1) / Wd::FPC;
  const __amdgpu_buffer_rsrc_t rx = em_rsrc(x, N * D::XFRAME * 4), rg = em_rsrc(g_out, N * D::YFRAME * 4),
                               ro = em_rsrc(out, N * D::YFRAME * 4);
  for (int i = tid; i < K::XPL; i += 512) xp[i] = 0.f;            // the borders stay zero for the whole kernel

  wn_f4 acc[16][2];                                               // Z_p[co = 16 wv + 4 g + r][ci = 16 nh + j]
#pragma unroll
  for (int p = 0; p < 16; ++p) acc[p][0] = acc[p][1] = wn_f4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  auto dma_piece = [&](int64_t k, int q) {            // dealt out one pair per k-step (see the data gradient)
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    const uint32_t off = (uint32_t)(k * K::RAW + (t2 + 512 * q) * 4) * 4u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void *)(rawg + (wv * 64 + 512 * q) * 4), 16, off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ro, (__attribute__((address_space(3))) void *)(rawo + (wv * 64 + 512 * q) * 4), 16, off, 0, 0, 0);
  };
  auto dma = [&](int64_t k) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dma_piece(k, q);
  };
  float4 pre;
  auto fetch = [&](int64_t k) {
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    pre = em_ld4(rx, (uint32_t)(k * 2048 + t2 * 4) * 4u);
  };
  auto stage = [&]() {                                // x[f][ci][h][w..w+3] -> plane (f, ci), row h + 1, column w + 1
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    const int e = t2 * 4, pl = e / D::PF, pix = e % D::PF;         // plane index f * 32 + ci
    float *d = xp + pl * K::XP + (pix / S + 1) * K::RS + pix % S + 1;
    d[0] = pre.x; d[1] = pre.y; d[2] = pre.z; d[3] = pre.w;
  };
  // raw -> gy4[ks][co][tile in group][2a + b], masked (co = 4 c + 2 dy + dx; row y = 4 ty + 2 a + dy, column xx = 4 tx + 2 b + dx).
  // 8x8: k-step = tile row ty, tile in group = tx; 4x4: k-step = frame of the set, tile in group = 2 ty + tx.
  auto convert = [&]() {
    int t2 = tid;
    asm volatile("" : "+v"(t2));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = (t2 + 512 * q) * 4;
      const float4 gv = *reinterpret_cast<const float4 *>(rawg + e), ov = *reinterpret_cast<const float4 *>(rawo + e);
      int ks, co, tg, a;
      if constexpr (S == 8) {
        const int c = e >> 8, y = (e >> 4) & 15, xx = e & 15;
        ks = y >> 2, co = 4 * c + 2 * (y & 1), tg = xx >> 2, a = (y >> 1) & 1;
      } else {
        const int f = e >> 11, c = (e >> 6) & 31, y = (e >> 3) & 7, xx = e & 7;
        ks = f, co = 4 * c + 2 * (y & 1), tg = 2 * (y >> 2) + (xx >> 2), a = (y >> 1) & 1;
      }
      float *d = gy4 + ((ks * 128 + co) * 4 + tg) * 4 + 2 * a;
      *reinterpret_cast<float2 *>(d) = make_float2(ov.x > 0.f ? gv.x : 0.f, ov.z > 0.f ? gv.z : 0.f);          // dx = 0: b = 0, 1
      *reinterpret_cast<float2 *>(d + 16) = make_float2(ov.y > 0.f ? gv.y : 0.f, ov.w > 0.f ? gv.w : 0.f);     // dx = 1 (next co)
    }
  };
  // operand row of this thread in vt: (ks = wv >> 1, nh = wv & 1, lane): channel 16 nh + j, tile g of k-step ks
  auto transform = [&]() {
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    const int l2 = t2 & 63, w2 = t2 >> 6, ks = w2 >> 1, ci = 16 * (w2 & 1) + (l2 & 15), tg = l2 >> 4;
    const int pb = S == 8 ? ci * K::XP + 2 * ks * K::RS + 2 * tg : (ks * 32 + ci) * K::XP + 2 * (tg >> 1) * K::RS + 2 * (tg & 1);
    float2 d[4][2];
    float v[16];
    wino_patch_load(xp + pb, K::RS, d);
    wino_patch_xform(d, v);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4 *>(vt + t2 * K::VROW + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  };
  auto compute = [&](int64_t knext) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // A operands: (A gy A^T)_p of (co = 16 wv + j, tile g of the k-step); gy tile = (y00, y01, y10, y11)
      const float4 t = *reinterpret_cast<const float4 *>(gy4 + ((ks * 128 + 16 * wv + j) * 4 + g) * 4);
      bsum += (t.x + t.y) + (t.z + t.w);
      float yh[16];
      {
        const float r0[2] = {t.x, t.y}, r1[2] = {t.x + t.z, t.y + t.w}, r2[2] = {t.x - t.z, t.y - t.w}, r3[2] = {-t.z, -t.w};
        const float *rr[4] = {r0, r1, r2, r3};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          yh[4 * u + 0] = rr[u][0];
          yh[4 * u + 1] = rr[u][0] + rr[u][1];
          yh[4 * u + 2] = rr[u][0] - rr[u][1];
          yh[4 * u + 3] = -rr[u][1];
        }
      }
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        float4 b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<const float4 *>(vt + ((2 * ks + nh) * 64 + lane) * K::VROW + 4 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[4 * q + 0][nh] = WN_MFMA(yh[4 * q + 0], b[q].x, acc[4 * q + 0][nh]);
          acc[4 * q + 1][nh] = WN_MFMA(yh[4 * q + 1], b[q].y, acc[4 * q + 1][nh]);
          acc[4 * q + 2][nh] = WN_MFMA(yh[4 * q + 2], b[q].z, acc[4 * q + 2][nh]);
          acc[4 * q + 3][nh] = WN_MFMA(yh[4 * q + 3], b[q].w, acc[4 * q + 3][nh]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the raw buffers are free since the convert phase; all pieces go out in the FIRST half of the set: the barrier that ends it
      // drains the DMA, and a piece issued behind the last k-step showed up as ~1400 cycles of every wave waiting there (stamps)
      if (ks < 2) dma_piece(knext, 2 * ks), dma_piece(knext, 2 * ks + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  int64_t k = blockIdx.x;
  dma(k);
  fetch(k);
  __syncthreads();                                    // zero fill done, raw(k) landed
  stage();
  fetch(k + stride);
  __syncthreads();
  for (int slot = 0; k < nsets; k += stride, ++slot) {
    WN_STAMP(slot, 0);
    convert();                                        // raw(k) -> gy4
    WN_STAMP(slot, 1);
    transform();                                      // planes(k) -> vt
    WN_STAMP(slot, 2);
    __syncthreads();
    WN_STAMP(slot, 3);
    stage();                                          // planes <- x(k + stride)
    fetch(k + 2 * stride);
    compute(k + stride);                              // ... and the DMA of raw(k + stride), dealt out between the k-steps
    WN_STAMP(slot, 4);
    __syncthreads();
    WN_STAMP(slot, 5);
  }
  // G^T Z G per (co, ci), partial sums of this workgroup
  float *wp = w_partials + (int64_t)blockIdx.x * UP_W;
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t[3][4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float z0 = acc[v][nh][r], z1 = acc[4 + v][nh][r], z2 = acc[8 + v][nh][r], z3 = acc[12 + v][nh][r];
        t[0][v] = z0 + 0.5f * (z1 + z2);
        t[1][v] = 0.5f * (z1 - z2);
        t[2][v] = 0.5f * (z1 + z2) + z3;
      }
      float *o = wp + ((16 * wv + 4 * g + r) * UP_CI + 16 * nh + j) * 9;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        o[3 * a + 0] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
        o[3 * a + 1] = 0.5f * (t[a][1] - t[a][2]);
        o[3 * a + 2] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
      }
    }
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  if (g == 0) b_partials[(int64_t)blockIdx.x * UP_CO + 16 * wv + j] = bsum;
}

}  // namespace kvae
