// Probe: operand and result layout of v_mfma_f64_16x16x4_f64 (D[16x16] += A[16x4] B[4x16]) as hipcc's builtin exposes it --
// lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; which D elements does it get back?  (design aid for the Float64 dW
// GEMM of nf_generic64.hip)
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_f64_probe.hip -o tools/probe/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double *out) {
  const int l = threadIdx.x;
  // A[i][k] = 1 + i + 100 k, B[k][j] = (k == 0 ? 1 : 0) * (1 + j): D[i][j] = (1 + i)(1 + j)
  const int i = l & 15, kk = l >> 4;
  const double a = 1.0 + i + 100.0 * kk, b = kk == 0 ? 1.0 + (l & 15) : 0.0;
  f64x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
int main() {
  double *o, h[256];
  hipMalloc(&o, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  // decode: D[i][j] = (1+i)(1+j) -> find (i, j) for lane 0..63, r 0..3 under hypotheses
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const double v = h[l * 4 + r];
      const int j = l & 15, i1 = 4 * (l >> 4) + r, i2 = (l >> 4) + 4 * r;
      if (v != (1.0 + i1) * (1.0 + j)) ok1 = 0;
      if (v != (1.0 + i2) * (1.0 + j)) ok2 = 0;
    }
  printf("D layout: col = lane & 15; row = 4 (lane >> 4) + r: %s; row = (lane >> 4) + 4 r: %s\n", ok1 ? "YES" : "no", ok2 ? "YES" : "no");
  printf("lane 0: %g %g %g %g | lane 16: %g %g %g %g | lane 1: %g %g %g %g\n", h[0], h[1], h[2], h[3], h[64], h[65], h[66], h[67], h[4], h[5], h[6], h[7]);
  return 0;
}
