// Self-test of the matrix-core quad primitives of csrc/lgssm_m4.h against plain loops (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Iinclude tools/m4_selftest.hip -o tools/_bin/m4_selftest
//   tools/_bin/m4_selftest        -> one line per primitive, "m4_selftest OK" / exit code 1
// What it pins: the operand convention of v_mfma_f32_4x4x1_16B_f32 the kernels rely on (P(X, Y, C) = R(Y X^T + C)), that a
// product with the operands exchanged is the transpose BIT FOR BIT, the rank-one form, the column loads, the natural-order
// solve (residual, and bit equality with the pivoted solve of lgssm_q4.h when that one exchanges nothing).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../kalman-vae_amd/csrc/lgssm_m4.h"
using namespace kvae;
using q4::Mat;

constexpr int NOUT = 9;   // matrices written per quad
__global__ void k_test(const float *X, const float *Y, const float *C, const float *S, const float *v, float *out, int *flags) {
  const int lane = threadIdx.x & 63, i = lane & 3, qd = lane >> 2;
  const Mat x = q4::load_rows(X + qd * 16, i), y = q4::load_rows(Y + qd * 16, i), c = q4::load_rows(C + qd * 16, i);
  const Mat ct = m4::load_cols(C + qd * 16, i), spd = q4::load_rows(S + qd * 16, i);
  const float a = v[qd * 8 + i], b = v[qd * 8 + 4 + i];
  const Mat I4 = q4::eye(i);
  float *o = out + qd * 16 * NOUT;
  q4::store_rows(o, m4::P(x, y, c), i);                        // 0: Y X^T + C
  q4::store_rows(o + 16, m4::P(y, x, ct), i);                  // 1: X Y^T + C^T   (bitwise transpose of 0)
  q4::store_rows(o + 32, m4::P(x, I4), i);                     // 2: X^T
  q4::store_rows(o + 48, m4::outer2(a, b, b, a, c), i);        // 3: C + b a^T + a b^T
  q4::store_rows(o + 64, ct, i);                               // 4: C^T by column loads
  bool bad;
  const Mat sol = m4::solve_natural(spd, y, i, bad);           // 5: S^{-1} Y
  q4::store_rows(o + 80, sol, i);
  q4::store_rows(o + 96, m4::solve_pivoted(spd, y, i, lane), i);   // 6: the same through q4::solve
  bool bad2;
  const Mat sol2 = m4::solve_natural(x, y, i, bad2);           // X is not positive definite for most quads: must say so
  q4::store_rows(o + 112, sol2, i);
  const m4::Vec4 av = m4::spread(a);
  Mat d;
  d.c[0] = m4::dot(x, av, b), d.c[1] = av.c[0], d.c[2] = av.c[3], d.c[3] = q4::qsum(a);
  q4::store_rows(o + 128, d, i);                               // 8: [ b + X a | a_0 | a_3 | sum a ]
  flags[lane] = (bad ? 1 : 0) | (bad2 ? 2 : 0);
}

int main() {
  const int Q = 16;
  std::vector<float> X(Q * 16), Y(Q * 16), C(Q * 16), S(Q * 16), v(Q * 8), out(Q * 16 * NOUT, 0.f);
  std::vector<int> flags(64, 0);
  srand(3);
  auto rnd = [] { return (rand() % 2001 - 1000) / 1000.f; };
  for (auto &t : X) t = rnd();
  for (auto &t : Y) t = rnd();
  for (auto &t : C) t = rnd();
  for (auto &t : v) t = rnd();
  for (int q = 0; q < Q; ++q) {   // S = G G^T + 2 I, diagonally heavy enough that partial pivoting exchanges nothing
    float G[16];
    for (auto &t : G) t = 0.3f * rnd();
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        float s = i == j ? 2.0f : 0.0f;
        for (int k = 0; k < 4; ++k) s += G[i * 4 + k] * G[j * 4 + k];
        S[q * 16 + i * 4 + j] = s;
      }
    X[q * 16] = -fabsf(X[q * 16]) - 0.1f;   // a negative leading pivot: never positive definite
  }
  float *dX, *dY, *dC, *dS, *dv, *dout;
  int *dflags;
  const size_t mb = Q * 16 * 4;
  hipMalloc(&dX, mb), hipMalloc(&dY, mb), hipMalloc(&dC, mb), hipMalloc(&dS, mb), hipMalloc(&dv, Q * 8 * 4);
  hipMalloc(&dout, out.size() * 4), hipMalloc(&dflags, 64 * 4);
  hipMemcpy(dX, X.data(), mb, hipMemcpyHostToDevice), hipMemcpy(dY, Y.data(), mb, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), mb, hipMemcpyHostToDevice), hipMemcpy(dS, S.data(), mb, hipMemcpyHostToDevice);
  hipMemcpy(dv, v.data(), Q * 8 * 4, hipMemcpyHostToDevice);
  k_test<<<1, 64>>>(dX, dY, dC, dS, dv, dout, dflags);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("m4_selftest: kernel failed\n");
    return 1;
  }
  hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(flags.data(), dflags, 64 * 4, hipMemcpyDeviceToHost);
  double e_p = 0, e_tr = 0, e_o = 0, e_x = 0, e_d = 0;
  int bit_tr = 0, bit_cols = 0, bit_solve = 0, flag_err = 0;
  for (int q = 0; q < Q; ++q) {
    const float *x = &X[q * 16], *y = &Y[q * 16], *c = &C[q * 16], *s = &S[q * 16], *a = &v[q * 8], *b = a + 4;
    const float *o = &out[q * 16 * NOUT];
    double sa = 0;
    for (int k = 0; k < 4; ++k) sa += a[k];
    for (int i = 0; i < 4; ++i) {
      double xa = b[i];
      for (int k = 0; k < 4; ++k) xa += (double)x[i * 4 + k] * a[k];
      e_d = fmax(e_d, fmax(fabs(xa - o[128 + i * 4]), fmax(fabs(a[0] - o[128 + i * 4 + 1]), fmax(fabs(a[3] - o[128 + i * 4 + 2]),
                                                                                                 fabs(sa - o[128 + i * 4 + 3])))));
      for (int j = 0; j < 4; ++j) {
        double p = c[i * 4 + j];
        for (int k = 0; k < 4; ++k) p += (double)y[i * 4 + k] * x[j * 4 + k];
        e_p = fmax(e_p, fabs(p - o[i * 4 + j]));
        bit_tr += memcmp(&o[i * 4 + j], &o[16 + j * 4 + i], 4) != 0;
        e_tr = fmax(e_tr, fabs(x[j * 4 + i] - o[32 + i * 4 + j]));
        e_o = fmax(e_o, fabs((double)c[i * 4 + j] + (double)b[i] * a[j] + (double)a[i] * b[j] - o[48 + i * 4 + j]));
        bit_cols += memcmp(&c[j * 4 + i], &o[64 + i * 4 + j], 4) != 0;
        double r = 0;   // residual of S sol = Y
        for (int k = 0; k < 4; ++k) r += (double)s[i * 4 + k] * o[80 + k * 4 + j];
        e_x = fmax(e_x, fabs(r - y[i * 4 + j]));
        bit_solve += memcmp(&o[80 + i * 4 + j], &o[96 + i * 4 + j], 4) != 0;
      }
    }
    for (int i = 0; i < 4; ++i) flag_err += flags[q * 4 + i] != 2;   // S accepted, X refused
  }
  printf("P(X,Y,C) = Y X^T + C        max err %.2e\n", e_p);
  printf("P(Y,X,C^T) bitwise transpose  mismatches %d\n", bit_tr);
  printf("P(X,I) = X^T                max err %.2e\n", e_tr);
  printf("outer2                      max err %.2e\n", e_o);
  printf("load_cols                   mismatches %d\n", bit_cols);
  printf("solve_natural residual      max err %.2e ; vs pivoted solve mismatches %d ; pivot flags wrong %d\n", e_x, bit_solve, flag_err);
  printf("spread / dot / qsum         max err %.2e\n", e_d);
  const bool ok = e_p < 2e-6 && bit_tr == 0 && e_tr == 0 && e_o < 2e-6 && bit_cols == 0 && e_x < 2e-5 && bit_solve == 0 &&
                  flag_err == 0 && e_d < 2e-6;
  printf(ok ? "m4_selftest OK\n" : "m4_selftest FAILED\n");
  return ok ? 0 : 1;
}
