// Micro-benchmark (kernel-tuning aid, not part of the library): v_mfma_f32_32x32x16_bf16 issued as a chain on ONE accumulator, or
// rotating over 2 / 4 accumulators, with one or two waves per SIMD, bare or with K vector instructions between two MFMAs -- does a
// dependent accumulation chain (the dX GEMMs of k_affine_bwd_pair since round 6: 24 MFMAs on one 32 x 32 block) run at the issue rate,
// and what does a second wave's chain on the same matrix pipe do to it?
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_bf16_chain_probe.hip -o tools/probe/mfma_bf16_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int K, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void probe(float *out, long long *cyc, int iters) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f + 0.01f * i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      const int a = m % NACC;
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) v[(m * K + k) & 7] = v[(m * K + k) & 7] * 1.0001f + 0.5f;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float sum = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) sum += acc[a][r];
  for (int i = 0; i < 8; ++i) sum += v[i];
  out[blockIdx.x * THREADS + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int K, int THREADS>
void run(float *out, long long *cyc) {
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<NACC, K, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
  }
  long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%d wave(s) per SIMD, %d accumulator(s), %d vector instructions between MFMAs: %6.1f clk per MFMA of one wave (pipe: %5.1f)\n",
         THREADS / 256, NACC, K, (double)c / (iters * 24.0), (double)c / (iters * 24.0) / (THREADS / 256));
}

int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
#define ROWS(T) run<1, 0, T>(out, cyc); run<2, 0, T>(out, cyc); run<4, 0, T>(out, cyc); run<1, 2, T>(out, cyc); run<2, 2, T>(out, cyc); \
  run<4, 2, T>(out, cyc); run<1, 5, T>(out, cyc); run<2, 5, T>(out, cyc); run<1, 8, T>(out, cyc); run<2, 8, T>(out, cyc);
  ROWS(256)
  ROWS(512)
  return 0;
}
