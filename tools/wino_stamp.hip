// Debug harness: per-phase s_memtime stamps (100 MHz) of k_dec_up_fwd_wino<8>, workgroup 0, waves 0 and 4 (one SIMD).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DKVAE_EM_STAMPS -Ikalman-vae_amd/csrc tools/wino_stamp.hip -o tools/_bin/wino_stamp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "vae_conv_up_wino.h"
int main() {
  const int64_t N = 12800;
  float *in, *W, *b, *out;
  hipMalloc(&in, N * 2048 * 4); hipMalloc(&W, 36864 * 4); hipMalloc(&b, 512); hipMalloc(&out, N * 8192 * 4);
  hipMemset(in, 0, N * 2048 * 4); hipMemset(W, 0, 36864 * 4); hipMemset(b, 0, 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) kvae::k_dec_up_fwd_wino<8, true><<<256, 512>>>(in, W, b, out, N);
  hipEventRecord(e0);
  kvae::k_dec_up_fwd_wino<8, true><<<256, 512>>>(in, W, b, out, N);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("kernel %.1f us (zero operands)\n", ms * 1e3);
  std::vector<unsigned long long> h(4096);
  hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(kvae::em_stamps), 4096 * 8);
  for (int w = 0; w < 8; ++w) {
    printf("wave %d  slot: first part | second part | barrier || first | second | barrier || two sets (shader cycles; waves 0-3: shared "
           "vector work then MFMAs + folds, waves 4-7: the reverse)\n", w);
    for (int s = 6; s < 8; ++s) {
      unsigned long long *t = &h[(w * 40 + s) * 8];
      printf("%2d: %5llu | %5llu | %5llu || %5llu | %5llu | %5llu || %6llu\n", s, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4],
             t[6] - t[5], t[6] - t[0]);
    }
  }
  return 0;
}
