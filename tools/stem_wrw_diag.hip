// k_enc_stem_wrw_mfma<true> alone at 12800 frames (HIP events).  The per-part costs quoted in DESIGN.md section 4b came from this
// harness with parts of the kernel compiled out one at a time (mask arithmetic, MFMAs, frame tile load, LDS staging writes:
// temporary #if blocks, not kept in the kernel).  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ikalman-vae_amd/csrc -Iinclude tools/stem_wrw_diag.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "vae_conv_edge.h"
int main() {
  const int64_t N = 12800;
  float *x, *g, *W, *b, *wp, *bp;
  hipMalloc(&x, N * 1024 * 4); hipMalloc(&g, N * 8192 * 4); hipMalloc(&W, N * 256 * 4); hipMalloc(&b, 128); hipMalloc(&wp, 1024 * 288 * 4); hipMalloc(&bp, 1024 * 32 * 4);
  hipMemset(x, 0, N * 1024 * 4); hipMemset(g, 0, N * 8192 * 4); hipMemset(W, 0, N * 256 * 4); hipMemset(b, 0, 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) kvae::k_enc_stem_wrw_mfma<true><<<1024, 256>>>(x, nullptr, (const uint32_t *)W, g, wp, bp, N);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) kvae::k_enc_stem_wrw_mfma<true><<<1024, 256>>>(x, nullptr, (const uint32_t *)W, g, wp, bp, N);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%.1f us per launch (%.2f TB/s of 0.47 GB)\n", ms * 200, 0.471e9 / (ms * 200e-6) / 1e12);
  return 0;
}
