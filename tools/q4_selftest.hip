// Self-test of the quad-layout primitives of csrc/lgssm_q4.h against plain loops (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. tools/q4_selftest.hip -o tools/_bin/q4_selftest && tools/_bin/q4_selftest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../kalman-vae_amd/csrc/lgssm_q4.h"
using namespace kvae::q4;

__global__ void k_test(const float *A, const float *B, const float *v, float *out) {
  const int lane = threadIdx.x & 63, i = lane & 3, qd = lane >> 2;
  Mat a = load_rows(A + qd * 16, i), b = load_rows(B + qd * 16, i);
  float vv = v[qd * 4 + i];
  guard(a); guard(b); guard(vv);
  const Mat nn = mul_nn(a, b);
  const Mat nt = mul_nt(a, b, zero());
  const float mv = matvec(a, vv, 0.0f);
  const Mat tr = transpose(a, i);
  Mat o = eye(i);
  outer_acc(o, vv, vv);
  // solve a X = b  (a made diagonally dominant by the host)
  const Mat x = solve(a, b, i, lane);
  float *o0 = out + qd * 16 * 6;
  store_rows(o0, nn, i); store_rows(o0 + 16, nt, i); store_rows(o0 + 32, tr, i); store_rows(o0 + 48, o, i); store_rows(o0 + 64, x, i);
  o0[80 + i] = mv; o0[84 + i] = qsum(vv);
}

int main() {
  const int Q = 16;
  std::vector<float> A(Q * 16), B(Q * 16), v(Q * 4), out(Q * 96, 0.f);
  srand(1);
  for (auto &x : A) x = (rand() % 2001 - 1000) / 1000.f;
  for (auto &x : B) x = (rand() % 2001 - 1000) / 1000.f;
  for (auto &x : v) x = (rand() % 2001 - 1000) / 1000.f;
  for (int q = 0; q < Q; ++q) if (q % 2 == 0) for (int i = 0; i < 4; ++i) A[q * 16 + i * 4 + i] += 4.f;   // half the quads need pivoting
  float *dA, *dB, *dv, *dout;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dv, v.size() * 4); hipMalloc(&dout, out.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice);
  k_test<<<1, 64>>>(dA, dB, dv, dout);
  hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  double e_nn = 0, e_nt = 0, e_tr = 0, e_o = 0, e_x = 0, e_mv = 0, e_qs = 0;
  for (int q = 0; q < Q; ++q) {
    const float *a = &A[q * 16], *b = &B[q * 16], *vv = &v[q * 4], *o = &out[q * 96];
    for (int i = 0; i < 4; ++i) {
      double mv = 0, qs = 0;
      for (int k = 0; k < 4; ++k) mv += a[i * 4 + k] * vv[k], qs += vv[k];
      e_mv = fmax(e_mv, fabs(mv - o[80 + i])); e_qs = fmax(e_qs, fabs(qs - o[84 + i]));
      for (int c = 0; c < 4; ++c) {
        double nn = 0, nt = 0;
        for (int k = 0; k < 4; ++k) nn += a[i * 4 + k] * b[k * 4 + c], nt += a[i * 4 + k] * b[c * 4 + k];
        e_nn = fmax(e_nn, fabs(nn - o[i * 4 + c])); e_nt = fmax(e_nt, fabs(nt - o[16 + i * 4 + c]));
        e_tr = fmax(e_tr, fabs(a[c * 4 + i] - o[32 + i * 4 + c]));
        e_o = fmax(e_o, fabs((i == c ? 1.0 : 0.0) + vv[i] * vv[c] - o[48 + i * 4 + c]));
        double r = 0;   // residual of a x = b
        for (int k = 0; k < 4; ++k) r += a[i * 4 + k] * o[64 + k * 4 + c];
        e_x = fmax(e_x, fabs(r - b[i * 4 + c]));
      }
    }
  }
  printf("max abs err: mul_nn %.2e  mul_nt %.2e  transpose %.2e  outer %.2e  solve residual %.2e  matvec %.2e  qsum %.2e\n", e_nn, e_nt,
         e_tr, e_o, e_x, e_mv, e_qs);
  const bool ok = e_nn < 1e-5 && e_nt < 1e-5 && e_tr == 0 && e_o < 1e-6 && e_x < 1e-3 && e_mv < 1e-5 && e_qs < 1e-5;
  printf(ok ? "q4 primitives OK\n" : "q4 primitives FAILED\n");
  return ok ? 0 : 1;
}
