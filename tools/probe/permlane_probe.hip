// what v_permlane32_swap_b32 does on gfx950, as __builtin_amdgcn_permlane32_swap(a, b) returns it
// build + run: hipcc --offload-arch=gfx950 -O2 tools/probe/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
  unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
  asm volatile("" : "+v"(a), "+v"(b));
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned *d, h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("a = 100 + lane, b = 200 + lane\nr[0]: lane 0 -> %u, lane 31 -> %u, lane 32 -> %u, lane 63 -> %u\n", h[0], h[31], h[32], h[63]);
  printf("r[1]: lane 0 -> %u, lane 31 -> %u, lane 32 -> %u, lane 63 -> %u\n", h[64], h[95], h[96], h[127]);
  return 0;
}
