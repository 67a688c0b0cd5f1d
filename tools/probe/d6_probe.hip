// Probe (design aid, not part of the library): the transposed hand-over of bf16 triples between the two waves of the pair
// kernel (split_C -> split_to_lds -> ds_read_b128, nf_mfma.h).  One wave writes a known tile as the producer does, reads it
// back as the consumer does, rebuilds h + m + l per (feature, sample) and compares with the input; also checks
// v_dot2c_f32_bf16 against (1, 1) as a sum of two bf16 values.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I normalizingflows.jl_amd/csrc -I include tools/probe/d6_probe.hip -o tools/probe/d6_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "nf_common.h"
#include "nf_mfma.h"

__device__ __host__ inline float val(int f, int s) { return __uint_as_float(0x3F800000u + 7919u * (unsigned)(f * 32 + s) + ((unsigned)(f & 3) << 23)); }  // bit patterns: no arithmetic to contract

__global__ void k(float *out, int *bad) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  constexpr int NB = 2;
  f32x16 d[NB];
  for (int b = 0; b < NB; ++b)
    for (int r = 0; r < 16; ++r) d[b][r] = val(32 * b + nf_row(r, hi), l31);
  SplitC<NB> s;
  split_C<NB>(d, s);
  split_to_lds<NB>(lds, s, l31, hi);
  __syncthreads();
  int nbad = 0;
  const nf_u32x4 *pd = reinterpret_cast<const nf_u32x4 *>(lds + l31 * D6_ROW + hi * 16);
  for (int ob = 0; ob < NB; ++ob) {
    float bs = 0.f, ref = 0.f;
    for (int g = 0; g < 2; ++g) {
      nf_u32x4 dc[3];
      for (int c = 0; c < 3; ++c) dc[c] = pd[ob * (32 * D6_ROW / 16) + c * 4 + g * 2];
      for (int j = 0; j < 8; ++j) {
        float v = 0.f;
        for (int c = 2; c >= 0; --c) {
          const unsigned w = dc[c][j >> 1];
          v += __uint_as_float((j & 1) ? (w & 0xFFFF0000u) : (w << 16));
        }
        const float want = val(32 * ob + l31, 2 * (8 * g + j) + hi);
        if (v != want) { ++nbad; if (nbad <= 2) printf("lane %d (f %d) ob %d g %d j %d (sample %d): got %g want %g\n", lane, 32 * ob + l31, ob, g, j, 2 * (8 * g + j) + hi, v, want); }
        ref += want;
      }
      for (int i = 0; i < 12; ++i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(bs) : "v"(dc[2 - i / 4][i % 4]), "v"(0x3F803F80u));
    }
    out[(ob * 64 + lane) * 2] = bs;
    out[(ob * 64 + lane) * 2 + 1] = ref;
  }
  bad[lane] = nbad;
}

__device__ __host__ inline float vala(int f, int s) { return 0.25f + 0.001f * (float)((f * 37 + s * 11) % 101) - 0.0003f * (float)(f % 7); }
__device__ __host__ inline float vald(int f, int s) { return -0.5f + 0.002f * (float)((f * 13 + s * 29) % 97) + 0.0001f * (float)(s % 5); }

// the consumer's GEMM on such a tile: dW^T[in][out] = sum_s a[in][s] delta[out][s], bias[out] = sum_s delta[out][s]
template <int IB, int OB>
__global__ void k2(float *dw, float *bias) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  f32x16 d[OB];
  for (int b = 0; b < OB; ++b)
    for (int r = 0; r < 16; ++r) d[b][r] = vald(32 * b + nf_row(r, hi), l31);
  SplitC<OB> s;
  split_C<OB>(d, s);
  split_to_lds<OB>(lds, s, l31, hi);
  float at[IB][16];
  for (int ib = 0; ib < IB; ++ib)
    for (int t = 0; t < 16; ++t) at[ib][t] = vala(32 * ib + l31, 2 * t + hi);
  SplitT<IB> as;
  split_T<IB>(at, as);
  __syncthreads();
  f32x16 acc[IB][OB];
  float bsum[OB];
  for (int ob = 0; ob < OB; ++ob) {
    bsum[ob] = 0.f;
    for (int ib = 0; ib < IB; ++ib)
      for (int r = 0; r < 16; ++r) acc[ib][ob][r] = 0.f;
  }
  dw_accumulate_t6<IB, OB>(as, lds, acc, bsum, l31, hi);
  for (int ib = 0; ib < IB; ++ib)
    for (int ob = 0; ob < OB; ++ob)
      for (int r = 0; r < 16; ++r) dw[(32 * ib + nf_row(r, hi)) * (32 * OB) + 32 * ob + l31] = acc[ib][ob][r];
  for (int ob = 0; ob < OB; ++ob) bias[(32 * ob + l31) * 2 + hi] = bsum[ob];
}

template <int IB, int OB>
int check_gemm() {
  float *dw, *bias;
  hipMalloc(&dw, 32 * IB * 32 * OB * 4);
  hipMalloc(&bias, 64 * OB * 4);
  hipLaunchKernelGGL((k2<IB, OB>), dim3(1), dim3(64), D6_BUF, 0, dw, bias);
  std::vector<float> w(32 * IB * 32 * OB), b(64 * OB);
  hipMemcpy(w.data(), dw, w.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), bias, b.size() * 4, hipMemcpyDeviceToHost);
  double e = 0, eb = 0;
  for (int i = 0; i < 32 * IB; ++i)
    for (int o = 0; o < 32 * OB; ++o) {
      double r = 0, sa = 0;
      for (int sidx = 0; sidx < 32; ++sidx) { r += (double)vala(i, sidx) * vald(o, sidx); sa += fabs((double)vala(i, sidx) * vald(o, sidx)); }
      e = fmax(e, fabs(w[i * 32 * OB + o] - r) / sa);
    }
  for (int o = 0; o < 32 * OB; ++o) {
    double r = 0, sa = 0;
    for (int sidx = 0; sidx < 32; ++sidx) { r += vald(o, sidx); sa += fabs(vald(o, sidx)); }
    eb = fmax(eb, fabs((double)b[o * 2] + b[o * 2 + 1] - r) / sa);
  }
  printf("dW GEMM %d x %d blocks: max error %.3e of sum |terms|, bias sums %.3e\n", IB, OB, e, eb);
  return (e > 1e-6 || eb > 1e-6);
}

int main() {
  float *out;
  int *bad;
  hipMalloc(&out, 256 * 4 * 2);
  hipMalloc(&bad, 64 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 2 * D6_BUF, 0, out, bad);
  for (int bytes : {155648, 163840}) {  // the pair kernel's LDS request with the triple buffers, and the CU's whole LDS
    hipError_t e1 = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), bytes, 0, out, bad);
    hipError_t e2 = hipGetLastError(), e3 = hipDeviceSynchronize();
    printf("dynamic LDS %d: set-attribute %d launch %d sync %d\n", bytes, (int)e1, (int)e2, (int)e3);
  }
  std::vector<float> o(512);
  std::vector<int> b(64);
  hipMemcpy(o.data(), out, 512 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), bad, 64 * 4, hipMemcpyDeviceToHost);
  int nb = 0;
  double e = 0;
  for (int i = 0; i < 64; ++i) nb += b[i];
  for (int i = 0; i < 256; ++i) e = fmax(e, fabs(o[2 * i] - o[2 * i + 1]) / fmax(1.0, fabs(o[2 * i + 1])));
  printf("mismatches %d of %d; dot2 sum max rel err %.3e (first: %g vs %g)\n", nb, 64 * 32, e, o[0], o[1]);
  return (nb != 0) | check_gemm<2, 1>() | check_gemm<2, 2>() | check_gemm<1, 2>();
}
