// Micro-benchmark (kernel-tuning aid, not part of the library): what ONE GEMM stage of k_rqs_bwd_coop costs per MFMA in
// isolation -- the library's own register-chained GEMM helpers (nf_mfma.h) at the cfg-3 chunk shape (one 32-sample tile,
// 32 x 96 weights in an LDS image of row stride 388), run back to back by 1 or 4 waves of a workgroup, one workgroup per CU.
// The in-kernel timeline of the real kernel (tools/trace_rqs.py) shows 76-93 clocks per MFMA in these stages against 64 for
// the instruction alone; this separates the LDS operand traffic from the issue rate.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I normalizingflows.jl_amd/csrc tools/probe/gemm_stage_probe.hip -o tools/probe/gemm_stage_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "nf_mfma.h"

constexpr int S3 = 388;

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float *out, long long *cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 33 * S3; i += blockDim.x) lds[i] = (i % 97) * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const float *w = lds + wave * 96;  // this wave's chunk of columns
  f32x16 delta[3], a2[1], d[1], o3[3];
  for (int b = 0; b < 3; ++b)
    for (int r = 0; r < 16; ++r) delta[b][r] = (lane + r + b) * 1e-3f;
  for (int r = 0; r < 16; ++r) a2[0][r] = (lane * 3 + r) * 1e-3f;
  float keep = 0.f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // dX3 as the kernel runs it: two interleaved accumulator chains, A operands by ds_read_b128
      dense_bwd_x_split<1, 3, S3, 2>(w, delta, d, l31, hi);
      delta[0][it & 15] += d[0][0] * 1e-30f;
    } else if (MODE == 1) {  // dX3, single chain
      dense_bwd_x<1, 3, S3>(w, delta, d, l31, hi);
      delta[0][it & 15] += d[0][0] * 1e-30f;
    } else if (MODE == 2) {  // the output-layer chunk GEMM (three accumulators, A operands by ds_read_b32)
      dense_fwd<1, 3, S3>(w, lds + 32 * S3, a2, o3, l31, hi);
      a2[0][it & 15] += o3[0][0] * 1e-30f + o3[1][1] * 1e-30f + o3[2][2] * 1e-30f;
    } else if (MODE == 3) {  // 48 MFMAs on two chains with every operand in registers (no LDS traffic at all)
      f32x16 p0 = {0}, p1 = {0};
#pragma unroll
      for (int t = 0; t < 48; t += 2) {
        p0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0][t & 15], delta[t / 16][t % 16], p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0][(t + 1) & 15], delta[(t + 1) / 16][(t + 1) % 16], p1, 0, 0, 0);
      }
      delta[0][it & 15] += (p0[0] + p1[0]) * 1e-30f;
    } else if (MODE == 4) {  // as 3, ONE chain
      f32x16 p0 = {0};
#pragma unroll
      for (int t = 0; t < 48; ++t) p0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0][t & 15], delta[t / 16][t % 16], p0, 0, 0, 0);
      delta[0][it & 15] += p0[0] * 1e-30f;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = clock64();
  for (int b = 0; b < 3; ++b)
    for (int r = 0; r < 16; ++r) keep += delta[b][r];
  for (int r = 0; r < 16; ++r) keep += a2[0][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves, float *out, long long *cyc) {
  const int iters = 400;
  const size_t lds = 34 * S3 * 4;
  hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  // 100 KB of dynamic LDS: one workgroup per CU, as in the real kernel
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(64 * waves), 100 * 1024, 0, out, cyc, iters);
    hipDeviceSynchronize();
  }
  long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-64s %d wave(s)/CU  %6.1f clk per MFMA\n", name, waves, (double)c / (iters * 48.0));
  (void)lds;
}

int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  for (int waves : {1, 4}) {
    run<3>("48 MFMAs, two chains, operands in registers", waves, out, cyc);
    run<4>("48 MFMAs, one chain, operands in registers", waves, out, cyc);
    run<0>("dX3 chunk (dense_bwd_x_split<1,3,388,2>: ds_read_b128 operands)", waves, out, cyc);
    run<1>("dX3 chunk (dense_bwd_x<1,3,388>: one chain)", waves, out, cyc);
    run<2>("output-layer chunk (dense_fwd<1,3,388>: ds_read_b32 operands)", waves, out, cyc);
  }
  return 0;
}
