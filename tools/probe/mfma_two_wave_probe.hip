// Micro-benchmark (kernel-tuning aid): does a SECOND wave per SIMD hide the VALU / LDS work that, with one wave,
// adds to the fp32 MFMA time?  Each wave runs [1 MFMA, K VALU fma] x 16 per iteration; WAVES = 4 or 8 per workgroup
// (1 or 2 per SIMD).  Reported: SIMD clocks per MFMA *executed on that SIMD* (64 = the matrix pipe is never idle for
// 32x32x2, 32 for 16x16x4).  build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_two_wave_probe.hip -o ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int WAVES, int SMALL>
__global__ __launch_bounds__(64 * WAVES, 1) void probe(float *out, long long *cyc, int iters) {
  f32x16 acc = {0};
  f32x4 acc4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float a0 = threadIdx.x * 1e-3f, b0 = 1.0f;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (SMALL == 2) {  // bf16 32x32x16: 8x the MACs of the fp32 32x32x2, on the matrix cores
        bf16x8 ab, bb;
        for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)a0; bb[e] = (__bf16)b0; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
      } else if (SMALL) {  // two 16x16x4 = the MACs of one 32x32x2
        acc4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc4[0], 0, 0, 0);
        acc4[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc4[1], 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int s = (m * K + k) & 15;
        v[s] = v[s] * 1.0001f + 0.5f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float sum = 0.f;
  for (int r = 0; r < 16; ++r) sum += acc[r] + v[r];
  for (int r = 0; r < 4; ++r) sum += acc4[0][r] + acc4[1][r];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int K, int WAVES, int SMALL>
void run(float *out, long long *cyc) {
  const int iters = 200;
  hipLaunchKernelGGL((probe<K, WAVES, SMALL>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep)
    hipLaunchKernelGGL((probe<K, WAVES, SMALL>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  // wall time per 32x32x2-equivalent MFMA executed by one SIMD, in ns (64 clk at 2.4 GHz = 26.7 ns)
  const double ns_per_mfma = ms * 1e6 / 5.0 / (iters * 16.0 * (WAVES / 4));
  long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per_wave = (double)c / (iters * 16.0);           // clocks per (MFMA + K VALU) step of ONE wave
  const double per_simd = per_wave / (WAVES / 4);               // SIMD clocks per 32x32x2-equivalent MFMA
  printf("%s  waves/SIMD=%d  K=%2d VALU per MFMA:  %6.1f clk per step per wave, %6.1f counter clk per MFMA-eq, %6.2f ns wall per MFMA-eq (%.1f %% of the 2.4 GHz peak)\n",
         SMALL == 2 ? "bf16 32x32x16" : SMALL ? "2x 16x16x4" : "1x 32x32x2", WAVES / 4, K, per_wave, per_simd, ns_per_mfma, 100.0 * 26.667 / ns_per_mfma);
}

int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
#define ROWS(SMALL) \
  run<0, 4, SMALL>(out, cyc); run<4, 4, SMALL>(out, cyc); run<8, 4, SMALL>(out, cyc); run<12, 4, SMALL>(out, cyc); \
  run<0, 8, SMALL>(out, cyc); run<4, 8, SMALL>(out, cyc); run<8, 8, SMALL>(out, cyc); run<12, 8, SMALL>(out, cyc);
  ROWS(0)
  ROWS(1)
  ROWS(2)
  return 0;
}
