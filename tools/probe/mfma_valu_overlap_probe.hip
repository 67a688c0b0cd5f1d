// Micro-benchmark (kernel-tuning aid, not part of the library): do one wave's vector instructions issue while ANOTHER wave of the same
// SIMD has a matrix instruction in the pipe?  A 512-thread workgroup: waves 0-3 (one per SIMD) issue bare v_mfma_f32_32x32x16_bf16,
// waves 4-7 (their SIMD partners) issue independent v_fma_f32 (inline asm: the compiler neither packs nor moves them).  Each role is
// also run alone.  Same question within ONE wave: K fillers behind every MFMA.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_overlap_probe.hip -o tools/probe/mfma_valu_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define FMA8(v)                                                                                                              \
  asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"   \
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])             \
               : "v"(c1), "v"(c2))

// MODE 0: both roles; 1: only the MFMA waves work; 2: only the VALU waves work.  NV: fma instructions per MFMA of the partner (x8)
template <int MODE, int NV8>
__global__ __launch_bounds__(512, 1) void two_roles(float *out, long long *cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[2];
  for (int a = 0; a < 2; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f + 0.01f * i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float c1 = 1.0001f, c2 = 0.5f;
  __syncthreads();
  const long long t0 = clock64();
  if (wave < 4) {
    if (MODE != 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m & 1], 0, 0, 0);
      }
  } else {
    if (MODE != 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16 * NV8; ++m) FMA8(v);
      }
  }
  const long long t1 = clock64();
  float sum = 0.f;
  for (int a = 0; a < 2; ++a)
    for (int r = 0; r < 16; ++r) sum += acc[a][r];
  for (int i = 0; i < 8; ++i) sum += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

// one wave per SIMD: K8 x 8 fma behind every MFMA
template <int K8>
__global__ __launch_bounds__(256, 1) void one_wave(float *out, long long *cyc, int iters) {
  f32x16 acc[2];
  for (int a = 0; a < 2; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f + 0.01f * i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float c1 = 1.0001f, c2 = 0.5f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m & 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K8; ++k) FMA8(v);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float sum = 0.f;
  for (int a = 0; a < 2; ++a)
    for (int r = 0; r < 16; ++r) sum += acc[a][r];
  for (int i = 0; i < 8; ++i) sum += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// two MFMA streams on one SIMD: the older wave (0-3) issues MFMA + KA8 x 8 fma per MFMA on NA accumulators, the younger (4-7) bare
// MFMAs on NB accumulators -- does a chain on ONE accumulator lose time when the partner's MFMAs slip in between its links?
template <int NA, int NB, int KA8, int KB8 = 0>
__global__ __launch_bounds__(512, 1) void two_chains(float *out, long long *cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f + 0.01f * i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float c1 = 1.0001f, c2 = 0.5f;
  __syncthreads();
  const long long t0 = clock64();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc[m % NA] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m % NA], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KA8; ++k) FMA8(v);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc[m % NB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m % NB], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KB8; ++k) FMA8(v);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = clock64();
  float sum = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) sum += acc[a][r];
  for (int i = 0; i < 8; ++i) sum += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

// the price of a filler by KIND (one wave per SIMD, 8 of them behind every MFMA; and without MFMAs): the kinds the six-term kernels use
#define F2(OP, A, B) OP " %" #A ", %" #A ", %" #B "\n"
template <int KIND, bool WITH_MFMA>
__global__ __launch_bounds__(256, 1) void filler_kind(float *out, long long *cyc, int iters) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x16 acc[2];
  for (int a = 0; a < 2; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f + 0.01f * i); }
  float v[8];
  f32x2 w[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.01f + i; w[i] = f32x2{v[i], v[i] + 1.f}; u[i] = threadIdx.x * 77u + i; }
  const f32x2 pc = {1.0001f, 0.9999f};
  const unsigned uc = 0x3F803F80u;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (WITH_MFMA) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m & 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[i]) : "v"(pc));
        else if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[i]) : "v"(pc));
        else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[(i + 1) & 7]));
        else if (KIND == 3) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[i]) : "v"(u[i]), "v"(uc));
        else if (KIND == 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(uc));
        else if (KIND == 5) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
        else if (KIND == 6) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(pc.x));
        else if (KIND == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(uc));
        else if (KIND == 8) asm volatile("v_bfe_i32 %0, %1, 3, 1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
        else if (KIND == 9) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(u[i]) : "v"(uc), "v"(u[(i + 1) & 7]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float sum = 0.f;
  for (int a = 0; a < 2; ++a)
    for (int r = 0; r < 16; ++r) sum += acc[a][r];
  for (int i = 0; i < 8; ++i) sum += v[i] + w[i].x + w[i].y + (float)u[i];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
  const int iters = 300;
  long long c[2];
#define TWO(MODE, NV8)                                                                                                     \
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((two_roles<MODE, NV8>), dim3(256), dim3(512), 0, 0, out, cyc, iters); hipDeviceSynchronize(); } \
  hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);                                                                            \
  printf("two roles, mode %d (0 both, 1 MFMA waves only, 2 vector waves only), %2d fma per partner MFMA: MFMA wave %6.1f clk per MFMA, vector wave %6.2f clk per fma\n", \
         MODE, 8 * NV8, (double)c[0] / (iters * 16.0), (double)c[1] / (iters * 16.0 * 8 * NV8));
  TWO(1, 1) TWO(2, 1) TWO(0, 1) TWO(2, 2) TWO(0, 2) TWO(2, 4) TWO(0, 4)
#define ONE(K8)                                                                                                            \
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((one_wave<K8>), dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); } \
  hipMemcpy(c, cyc, 8, hipMemcpyDeviceToHost);                                                                             \
  printf("one wave per SIMD, %2d fma behind every MFMA: %6.1f clk per MFMA + fillers\n", 8 * K8, (double)c[0] / (iters * 16.0));
  ONE(0) ONE(1) ONE(2) ONE(3) ONE(4)
#define CH(NA, NB, KA8)                                                                                                    \
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((two_chains<NA, NB, KA8>), dim3(256), dim3(512), 0, 0, out, cyc, iters); hipDeviceSynchronize(); } \
  hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);                                                                            \
  printf("two chains: older wave %d accumulator(s) + %2d fma per MFMA, younger wave %d accumulator(s) bare: older %6.1f, younger %6.1f clk per own MFMA\n", \
         NA, 8 * KA8, NB, (double)c[0] / (iters * 16.0), (double)c[1] / (iters * 16.0));
#define CH2(KA8, KB8)                                                                                                      \
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((two_chains<2, 2, KA8, KB8>), dim3(256), dim3(512), 0, 0, out, cyc, iters); hipDeviceSynchronize(); } \
  hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);                                                                            \
  printf("two mixed streams: older wave %2d fma per MFMA, younger wave %2d fma per MFMA: older %6.1f, younger %6.1f clk per own MFMA\n", \
         8 * KA8, 8 * KB8, (double)c[0] / (iters * 16.0), (double)c[1] / (iters * 16.0));
  CH2(1, 1) CH2(2, 2) CH2(1, 2) CH2(2, 1) CH2(3, 3)
  CH(1, 1, 0) CH(2, 2, 0) CH(1, 1, 1) CH(2, 2, 1) CH(4, 4, 1) CH(1, 2, 1) CH(2, 1, 1) CH(1, 1, 2) CH(2, 2, 2) CH(4, 4, 2)
#define KIND(K, NAME)                                                                                                      \
  { long long cw, co;                                                                                                      \
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((filler_kind<K, true>), dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); } \
    hipMemcpy(&cw, cyc, 8, hipMemcpyDeviceToHost);                                                                         \
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((filler_kind<K, false>), dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); } \
    hipMemcpy(&co, cyc, 8, hipMemcpyDeviceToHost);                                                                         \
    printf("8 x %-18s behind every MFMA: %6.1f clk per MFMA + fillers;  alone: %5.2f clk each\n", NAME, (double)cw / (iters * 16.0), (double)co / (iters * 128.0)); }
  KIND(6, "v_sub_f32") KIND(0, "v_pk_add_f32") KIND(1, "v_pk_mul_f32") KIND(2, "v_cvt_pk_bf16_f32") KIND(3, "v_dot2c_f32_bf16") KIND(4, "v_and_b32")
  KIND(5, "v_lshlrev_b32") KIND(7, "v_perm_b32") KIND(8, "v_bfe_i32") KIND(9, "v_bitop3_b32")
  return 0;
}
