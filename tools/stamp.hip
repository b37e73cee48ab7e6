// One-wave kernel that writes the GPU's constant 100 MHz clock into a slot: a time stamp that can be captured into a hipGraph on
// either stream (tools/join_wait_probe.py; external event-record nodes are not available on ROCm).
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/stamp.hip -o tools/_bin/libstamp.so
#include <hip/hip_runtime.h>
__global__ void k_stamp(unsigned long long *p) {
  if (threadIdx.x == 0) *p = wall_clock64();
}
extern "C" int kvae_tool_stamp(unsigned long long *buf, int slot, void *stream) {
  k_stamp<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(buf + slot);
  return (int)hipGetLastError();
}
