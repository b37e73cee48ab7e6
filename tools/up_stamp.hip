// Debug harness: per-phase s_memtime stamps of k_dec_up_fwd<8> (workgroup 0, thread 0).
//   hipcc --offload-arch=gfx950 -O3 -DKVAE_EM_STAMPS -Ikalman-vae_amd/csrc tools/up_stamp.hip -o tools/_bin/up_stamp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "vae_conv_up.h"
int main() {
  const int64_t N = 12800;
  float *x, *W, *b, *out;
  (void)hipMalloc(&x, N * 32 * 64 * 4); (void)hipMalloc(&W, 36864 * 4); (void)hipMalloc(&b, 512); (void)hipMalloc(&out, N * 128 * 64 * 4);
  (void)hipMemset(x, 0, N * 32 * 64 * 4); (void)hipMemset(W, 0, 36864 * 4); (void)hipMemset(b, 0, 512);
  for (int rep = 0; rep < 3; ++rep) kvae::k_dec_up_fwd<8><<<256, 256>>>(x, W, b, out, N);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(4096);
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(kvae::em_stamps), 4096 * 8);
  printf("slot: top->staged  t0 mfma  bar  t1..t3  last epilogue | iter total (cycles)\n");
  for (int s = 0; s < 25; ++s) {
    unsigned long long *t = &h[s * 8];
    printf("%2d: %6llu %6llu %6llu %6llu %6llu | %6llu\n", s, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4],
           s ? t[0] - h[(s - 1) * 8] : 0ull);
  }
  return 0;
}
