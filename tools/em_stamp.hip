// Debug harness: per-phase s_memtime stamps of k_enc_mid_fwd<16> (workgroup 0, thread 0).
//   hipcc --offload-arch=gfx950 -O3 -DKVAE_EM_STAMPS -Ikalman-vae_amd/csrc tools/em_stamp.hip -o gpurun_out/em_stamp && gpurun_out/em_stamp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "vae_conv_mid.h"
int main() {
  const int64_t N = 12800;
  float *in, *W, *b, *out;
  hipMalloc(&in, N * 32 * 256 * 4); hipMalloc(&W, 9216 * 4); hipMalloc(&b, 128); hipMalloc(&out, N * 32 * 64 * 4);
  hipMemset(in, 0, N * 32 * 256 * 4); hipMemset(W, 0, 9216 * 4); hipMemset(b, 0, 128);
  for (int rep = 0; rep < 3; ++rep) kvae::k_enc_mid_fwd<16><<<256, 256>>>(in, W, b, out, N);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(4096);
  hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(kvae::em_stamps), 4096 * 8);
  printf("slot: top->bar1  stage  bar2  flush+fetch  mfma  bar3  otwrite | iter total (s_memtime ticks, 100 MHz)\n");
  for (int s = 0; s < 25; ++s) {
    unsigned long long *t = &h[s * 8];
    printf("%2d: %6llu %6llu %6llu %6llu %6llu %6llu %6llu | %6llu\n", s, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3],
           t[5] - t[4], t[6] - t[5], t[7] - t[6], s ? t[0] - h[(s - 1) * 8] : 0ull);
  }
  return 0;
}
