// Debug harness: per-phase s_memtime stamps of k_dec_up_wrw_wino<8>, workgroup 0, waves 0 and 4 (one SIMD).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DKVAE_EM_STAMPS -Ikalman-vae_amd/csrc tools/wino_stamp_bwd.hip -o tools/_bin/wino_stamp_bwd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "vae_conv_up_wino.h"
int main() {
  const int64_t N = 12800;
  float *x, *o, *g, *wp, *bp;
  hipMalloc(&x, N * 2048 * 4); hipMalloc(&wp, 256 * 36864 * 4); hipMalloc(&bp, 256 * 128 * 4); hipMalloc(&o, N * 8192 * 4); hipMalloc(&g, N * 8192 * 4); 
  hipMemset(x, 0, N * 2048 * 4); hipMemset(o, 0, N * 8192 * 4); hipMemset(g, 0, N * 8192 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) kvae::k_dec_up_wrw_wino<8><<<256, 512>>>(x, o, g, wp, bp, N);
  hipEventRecord(e0);
  kvae::k_dec_up_wrw_wino<8><<<256, 512>>>(x, o, g, wp, bp, N);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("kernel %.1f us (zero operands)\n", ms * 1e3);
  std::vector<unsigned long long> h(4096);
  hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(kvae::em_stamps), 4096 * 8);
  for (int w = 0; w < 8; ++w) {
    printf("wave %d  slot: convert | transform | barrier | stage+compute | barrier || set total (shader cycles)\n", w);
    for (int s = 6; s < 8; ++s) {
      unsigned long long *t = &h[(w * 40 + s) * 8];
      printf("%2d: %5llu | %5llu | %5llu | %5llu | %5llu || %6llu\n", s, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]);
    }
  }
  return 0;
}
