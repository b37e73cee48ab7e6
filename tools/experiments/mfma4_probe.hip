// mfma4_probe.hip - v_mfma_f32_4x4x1_16B_f32 on gfx950: operand / result layout and dependent-chain latency, next to a chain of
// DPP-folded FMAs (the quad layout's product).  build: hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma4_probe.hip -o tools/_bin/mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float *a, const float *b, float *d) {
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[threadIdx.x * 4 + r] = c[r];
}
// dependent accumulate chain: c <- c + a b, N times
__global__ void k_chain_mfma(float *out, int n, long long *cyc) {
  f4 c = {0, 0, 0, 0};
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 1e-3f;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = c[0] + c[1] + c[2] + c[3];
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
// product chain: the RESULT of a 4-MFMA product is the A operand of the next (what A Sig A^T looks like)
__global__ void k_chain_product(float *out, int n, long long *cyc) {
  f4 x = {1.0f, 0.5f, 0.25f, 0.125f};
  const float b = 1e-1f + threadIdx.x * 1e-4f;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(x[0], b, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(x[1], b, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(x[2], b, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(x[3], b, c, 0, 0, 0);
    x = c;
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x[0] + x[1] + x[2] + x[3];
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
// the quad layout's product: 16 v_fmac_f32_dpp (4 chains of 4), result feeds the next product
__global__ void k_chain_dpp(float *out, int n, long long *cyc) {
  float x0 = 1.0f, x1 = 0.5f, x2 = 0.25f, x3 = 0.125f;
  const float b = 1e-1f + threadIdx.x * 1e-4f;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    float c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#define F(K, acc, src) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #K "," #K "," #K "," #K "] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(b))
    asm volatile("s_nop 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    F(0, c0, x0); F(0, c1, x1); F(0, c2, x2); F(0, c3, x3);
    F(1, c0, x0); F(1, c1, x1); F(1, c2, x2); F(1, c3, x3);
    F(2, c0, x0); F(2, c1, x1); F(2, c2, x2); F(2, c3, x3);
    F(3, c0, x0); F(3, c1, x1); F(3, c2, x2); F(3, c3, x3);
    x0 = c0, x1 = c1, x2 = c2, x3 = c3;
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float *a, *b, *d;
  long long *cyc;
  hipMalloc(&a, 256), hipMalloc(&b, 256), hipMalloc(&d, 1024), hipMalloc(&cyc, 8);
  std::vector<float> ha(64), hb(64), hd(256);
  for (int l = 0; l < 64; ++l) ha[l] = 1 + l, hb[l] = 101 + 2 * l;
  hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice), hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(a, b, d);
  hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
  // D[lane][r] = a[x] * b[y]: find x, y
  for (int l : {0, 1, 2, 3, 4, 5, 17, 63}) {
    printf("lane %2d:", l);
    for (int r = 0; r < 4; ++r) {
      int fx = -1, fy = -1;
      for (int x = 0; x < 64 && fx < 0; ++x)
        for (int y = 0; y < 64; ++y)
          if (ha[x] * hb[y] == hd[l * 4 + r]) { fx = x, fy = y; break; }
      printf("  reg%d = a[lane %2d] * b[lane %2d]", r, fx, fy);
    }
    printf("\n");
  }
  const int n = 4096;
  long long h;
  for (int rep = 0; rep < 2; ++rep) {
    k_chain_mfma<<<1, 64>>>(d, n, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("dependent accumulate chain : %.1f clock ticks per MFMA\n", (double)h / (4.0 * n));
    k_chain_product<<<1, 64>>>(d, n, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("product -> operand chain    : %.1f clock ticks per 4x4x4 product (4 MFMAs)\n", (double)h / n);
    k_chain_dpp<<<1, 64>>>(d, n, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("quad-layout DPP product     : %.1f clock ticks per 4x4x4 product (16 v_fmac_f32_dpp)\n", (double)h / n);
  }
  return 0;
}
