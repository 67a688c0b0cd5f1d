// Micro-benchmark (kernel-tuning aid, not part of the library): how many VALU / LDS instructions
// can be issued between two fp32 MFMAs (v_mfma_f32_32x32x2_f32, 64 cycles each) for free, with ONE
// wave per SIMD -- for independent accumulators (rotating over 4) and for a dependent chain.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_issue_probe.hip -o tools/probe/mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int MODE, bool DEP>  // MODE 0: VALU fma, 1: ds_read_b32, 2: transcendental (v_exp_f32)
__global__ __launch_bounds__(256, 1) void probe(float *out, long long *cyc, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 0.001f;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float a0 = threadIdx.x * 1e-3f, b0 = 1.0f;
  const float *lp = lds + (threadIdx.x & 63);
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int a = DEP ? 0 : (m & 3);
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[a], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int s = (m * K + k) & 15;
        if (MODE == 0) v[s] = v[s] * 1.0001f + 0.5f;
        else if (MODE == 1) v[s] += lp[((m * K + k) & 31) * 64];
        else v[s] = __expf(v[s]) * 0.001f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float sum = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) sum += acc[a][r];
  for (int i = 0; i < 16; ++i) sum += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int K, int MODE, bool DEP>
void run(const char *name, float *out, long long *cyc) {
  const int iters = 200;
  hipLaunchKernelGGL((probe<K, MODE, DEP>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<K, MODE, DEP>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (double)c / (iters * 16.0);
  printf("%-34s K=%2d  %7.1f clk/MFMA   (%.1f us, %.2f GHz-equivalent)\n", name, K, per, ms * 1e3, c / (ms * 1e6));
}

int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
#define ROW(MODE, DEP, NAME) \
  run<0, MODE, DEP>(NAME, out, cyc); run<1, MODE, DEP>(NAME, out, cyc); run<2, MODE, DEP>(NAME, out, cyc); \
  run<4, MODE, DEP>(NAME, out, cyc); run<8, MODE, DEP>(NAME, out, cyc); run<12, MODE, DEP>(NAME, out, cyc); \
  run<16, MODE, DEP>(NAME, out, cyc);
  ROW(0, false, "VALU fma, independent accumulators")
  ROW(0, true, "VALU fma, dependent MFMA chain")
  ROW(1, false, "ds_read_b32, independent")
  ROW(1, true, "ds_read_b32, dependent chain")
  ROW(2, false, "v_exp_f32+mul, independent")
  return 0;
}
