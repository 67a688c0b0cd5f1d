// accuracy of a hardware-transcendental Box-Muller against the libm one (kernel-tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(float *err, double *errd, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned a = i * 2654435761u, b = (i ^ 0x9e3779b9u) * 2246822519u;
  const float u0 = ((float)(a >> 9) + 0.5f) * 1.1920928955078125e-07f;
  const float u1 = ((float)(b >> 9) + 0.5f) * 1.1920928955078125e-07f;
  const float rad = sqrtf(-2.0f * logf(u0));
  float s, c;
  sincospif(2.0f * u1, &s, &c);
  const float z0 = rad * c, z1 = rad * s;
  const float radf = __builtin_amdgcn_sqrtf(-2.0f * __logf(u0));
  const float z0f = radf * __builtin_amdgcn_cosf(u1), z1f = radf * __builtin_amdgcn_sinf(u1);
  const double rd = sqrt(-2.0 * log((double)u0));
  const double z0d = rd * cospi(2.0 * (double)u1), z1d = rd * sinpi(2.0 * (double)u1);
  err[i] = fmaxf(fabsf(z0 - z0f), fabsf(z1 - z1f));
  errd[2 * i] = fmax(fabs(z0d - z0f), fabs(z1d - z1f));
  errd[2 * i + 1] = fmax(fabs(z0d - z0), fabs(z1d - z1));
}
int main() {
  const int n = 1 << 22;
  float *e; double *ed;
  hipMalloc(&e, n * 4); hipMalloc(&ed, n * 16);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, e, ed, n);
  float *h = new float[n]; double *hd = new double[2 * n];
  hipMemcpy(h, e, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hd, ed, n * 16, hipMemcpyDeviceToHost);
  float m = 0; double mf = 0, ml = 0;
  for (int i = 0; i < n; ++i) { m = fmaxf(m, h[i]); mf = fmax(mf, hd[2 * i]); ml = fmax(ml, hd[2 * i + 1]); }
  printf("max |fast - libm| = %.3e   max |fast - f64| = %.3e   max |libm - f64| = %.3e\n", m, mf, ml);
  return 0;
}
