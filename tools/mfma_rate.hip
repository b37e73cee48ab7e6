// Micro-test: cycles per v_mfma_f32_16x16x4_f32 on one SIMD with 1 or 2 waves issuing (4 or 8 waves per workgroup, one
// workgroup per CU), 16 independent accumulators each.   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/_bin/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int FILL>
__global__ void k(float *out, int iters) {
  f4 acc[16];
  for (int p = 0; p < 16; ++p) acc[p] = f4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f, c = 1.f, d = 2.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[p], 0, 0, 0);
      if (FILL >= 1) c = c * 1.0001f + d;
      if (FILL >= 2) d = d * 0.9999f + c;
      if (FILL >= 3) c = c * 1.0002f + d;
      if (FILL >= 4) d = d * 0.9998f + c;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = c + d;
  for (int p = 0; p < 16; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}
template <int FILL>
void run(int waves, float *out) {
  const int iters = 2000;
  k<FILL><<<256, waves * 64>>>(out, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<FILL><<<256, waves * 64>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms, ticks; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&ticks, out + (1 << 20), 4, hipMemcpyDeviceToHost);
  const double mf = (double)iters * 16 * (waves / 4);   // MFMAs per SIMD
  printf("waves/WG %d fillers/MFMA %d: %.1f us, %.1f memtime ticks per MFMA (per SIMD), %.1f TF/s\n", waves, FILL, ms * 1e3, ticks / mf,
         256.0 * waves * iters * 16 * 2048 / (ms * 1e-3) / 1e12);
}
int main() {
  float *out; hipMalloc(&out, ((1 << 20) + 16) * 4);
  run<0>(4, out); run<0>(8, out); run<2>(4, out); run<2>(8, out); run<4>(4, out); run<4>(8, out);
  return 0;
}
