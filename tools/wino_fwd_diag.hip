// k_dec_up_fwd_wino<8, true> alone at 12800 frames on RANDOM data (HIP events; zero-filled operands run ~10 % faster: the clock).
// With parts of the kernel compiled out (temporary #if blocks, not kept) this harness gave, of 289 us: patch-transform arithmetic
// 12, B-operand LDS reads 15, output stores 23 - the rest is the 128 MFMAs per wave and set and the folds that consume them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ikalman-vae_amd/csrc -Iinclude tools/wino_fwd_diag.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "vae_conv_up_wino.h"
int main() {
  const int64_t N = 12800;
  float *in, *W, *b, *out;
  hipMalloc(&in, N * 2048 * 4); hipMalloc(&W, 36864 * 4); hipMalloc(&b, 512); hipMalloc(&out, N * 8192 * 4);
  std::vector<float> h(N * 2048);
  for (auto &v : h) v = rand() / (float)RAND_MAX;
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int i = 0; i < 36864; ++i) h[i] = 0.1f * (rand() / (float)RAND_MAX - 0.5f);
  hipMemcpy(W, h.data(), 36864 * 4, hipMemcpyHostToDevice);
  hipMemset(b, 0, 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) kvae::k_dec_up_fwd_wino<8, true><<<256, 512>>>(in, W, b, out, N);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) kvae::k_dec_up_fwd_wino<8, true><<<256, 512>>>(in, W, b, out, N);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%.1f us per launch\n", ms * 200);
  return 0;
}
