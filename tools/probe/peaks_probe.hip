// Probe ("re-measure the peaks with a copy / MFMA microbenchmark on the box and report both"): what THIS MI355X
// delivers for the three peaks every roofline fraction in DESIGN.md is quoted against --
//   HBM: read-only sum, write-only fill and copy of a 2 GiB buffer (16 bytes per lane per access, grid-stride),
//   fp32 matrix pipe: v_mfma_f32_32x32x2_f32 on four independent accumulator chains per wave, 1 / 2 / 4 waves per SIMD,
//   bf16 matrix pipe: v_mfma_f32_32x32x16_bf16 the same way,
// together with the shader clock each loop ran at (s_memtime against s_memrealtime, 100 MHz).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/peaks_probe.hip -o tools/probe/peaks_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

__global__ void k_read(const float4 *__restrict__ a, size_t n, float *out) {
  float4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = a[i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;  // never true: keeps the loads
}
__global__ void k_fill(float4 *__restrict__ a, size_t n) {
  const float4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = v;
}
__global__ void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_mfma(int iters, float *out, long long *clk) {
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  const u32x4 ab = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (BF16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ab), __builtin_bit_cast(bf16x8, ab), acc[c], 0, 0, 0);
      else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0];
  if (s == 12345.678f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

static double time_ms(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
  const size_t bytes = (size_t)2 << 30, n4 = bytes / 16;
  float4 *a, *b;
  float *out;
  long long *clk;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&clk, 64));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 16;
  hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, a, n4);
  hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, b, n4);
  CHECK(hipDeviceSynchronize());
  const int reps = 10;
  for (int which = 0; which < 3; ++which) {
    for (int w = 0; w < 3; ++w) {  // warm-up (clock ramp)
      if (which == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n4, out);
      if (which == 1) hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, a, n4);
      if (which == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n4);
    }
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) {
      if (which == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n4, out);
      if (which == 1) hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, a, n4);
      if (which == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n4);
    }
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    const double ms = time_ms(e0, e1) / reps, moved = (which == 2 ? 2.0 : 1.0) * bytes;
    printf("HBM %-5s 2 GiB: %.3f ms per pass = %.2f TB/s (%s)\n", which == 0 ? "read" : which == 1 ? "write" : "copy", ms, moved / ms / 1e9,
           which == 2 ? "read + written bytes" : "bytes moved once");
  }
  for (int bf = 0; bf < 2; ++bf)
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD: blocks of 256 threads = one wave per SIMD each
      const int iters = 20000, blocks = 256 * wps;
      for (int w = 0; w < 2; ++w) {
        if (bf) hipLaunchKernelGGL((k_mfma<true>), dim3(blocks), dim3(256), 0, 0, iters, out, clk);
        else hipLaunchKernelGGL((k_mfma<false>), dim3(blocks), dim3(256), 0, 0, iters, out, clk);
      }
      hipEventRecord(e0);
      if (bf) hipLaunchKernelGGL((k_mfma<true>), dim3(blocks), dim3(256), 0, 0, iters, out, clk);
      else hipLaunchKernelGGL((k_mfma<false>), dim3(blocks), dim3(256), 0, 0, iters, out, clk);
      hipEventRecord(e1);
      CHECK(hipDeviceSynchronize());
      long long h[2];
      hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double ms = time_ms(e0, e1);
      const double flop = (double)blocks * 4 /*waves*/ * iters * 4 /*chains*/ * (bf ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2);
      const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
      printf("%s MFMA, %d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s, shader clock %.2f GHz\n",
             bf ? "bf16 32x32x16" : "fp32 32x32x2 ", wps, ms, flop / ms / 1e9, ghz);
    }
  return 0;
}
