// Probe (design aid, not part of the library): the three-way bf16 split with its two exact subtractions as v_dot2c_f32_bf16 --
//     r0 = x0 + dot((h.lo, h.hi), (-1, 0)),  r1 = x1 + dot((h.lo, h.hi), (0, -1))      (in place; no widening of h)
// seven instructions per pair of values instead of nine (cvt, dot, dot, cvt, dot, dot, perm).  Is it BIT-IDENTICAL to nf_split2 on
// normal, tie, subnormal-residual, zero and huge inputs (does the dot unit flush or round anything)?  And what does a burst of it cost?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I normalizingflows.jl_amd/csrc -I include tools/probe/split_dot_probe.hip -o tools/probe/split_dot_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "nf_common.h"
#include "nf_mfma.h"

__device__ __forceinline__ void split2_dot(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  const unsigned m10 = 0x0000BF80u, m01 = 0xBF800000u;  // bf16 (-1, 0) and (0, -1): low half first
  h = nf_cvt_pk_bf16(nf_f32x2{x0, x1});
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x0) : "v"(h), "v"(m10));
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x1) : "v"(h), "v"(m01));
  m = nf_cvt_pk_bf16(nf_f32x2{x0, x1});
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x0) : "v"(m), "v"(m10));
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x1) : "v"(m), "v"(m01));
  l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}

__global__ void k_cmp(const float *in, long n2, unsigned long long *bad, unsigned *first) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
    const float x0 = in[2 * i], x1 = in[2 * i + 1];
    unsigned h, m, l, h2, m2, l2;
    nf_split2(x0, x1, h, m, l);
    split2_dot(x0, x1, h2, m2, l2);
    if (h != h2 || m != m2 || l != l2) {
      if (atomicAdd(bad, 1ull) == 0) {
        first[0] = __float_as_uint(x0); first[1] = __float_as_uint(x1); first[2] = h; first[3] = h2; first[4] = m; first[5] = m2; first[6] = l; first[7] = l2;
      }
    }
  }
}

template <int KIND>
__global__ __launch_bounds__(256, 1) void k_time(float *out, long long *cyc, int iters) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.37f + i * 1.01f;
  unsigned acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      unsigned h, m, l;
      if (KIND == 0) nf_split2(v[2 * p], v[2 * p + 1], h, m, l);
      else split2_dot(v[2 * p], v[2 * p + 1], h, m, l);
      acc ^= h + m + l;
      v[2 * p] += 1.0f; v[2 * p + 1] += 0.5f;
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  const long n = 1 << 24;
  std::vector<float> x(n);
  std::mt19937_64 g(4321);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_int_distribution<int> ex(-100, 100), kind(0, 15);
  std::uniform_int_distribution<unsigned> bits(0, 0xFFFFFFFFu);
  for (long i = 0; i < n; ++i) {
    const int kd = kind(g);
    float v;
    if (kd < 8) v = std::ldexp(nd(g), ex(g) / 3);
    else if (kd < 11) {  // on / next to bf16 ties at either level
      unsigned b = bits(g) & 0x7FFFFFFFu;
      if ((b >> 23) >= 0xFD) b &= 0x7E7FFFFFu;
      b = (b & 0xFFFF0000u) | (0x8000u + (unsigned)(int)(bits(g) % 3) - 1u);
      if (bits(g) & 1) b = (b & 0xFFFFFF00u) | (0x80u + (unsigned)(int)(bits(g) % 3) - 1u);
      if (bits(g) & 1) b |= 0x80000000u;
      memcpy(&v, &b, 4);
    } else if (kd == 11) v = std::ldexp(nd(g), -120 - (int)(bits(g) % 8));  // residuals fall into the subnormal range
    else if (kd == 12) { unsigned b = bits(g) % 0x00800000u; if (bits(g) & 1) b |= 0x80000000u; memcpy(&v, &b, 4); }  // subnormal inputs
    else if (kd == 13) v = (bits(g) & 1) ? 0.f : -0.f;
    else if (kd == 14) v = std::ldexp(nd(g), 120);  // large, finite after rounding
    else { unsigned b = bits(g); if (((b >> 23) & 0xFF) == 0xFF) b &= 0xFF7FFFFFu; if (((b >> 16) & 0x7FFF) >= 0x7F7F) b &= 0xFF7EFFFFu; memcpy(&v, &b, 4); }
    x[i] = v;
  }
  float *din; unsigned long long *dbad; unsigned *dfirst;
  hipMalloc(&din, n * 4); hipMalloc(&dbad, 8); hipMalloc(&dfirst, 32);
  hipMemcpy(din, x.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 8);
  hipLaunchKernelGGL(k_cmp, dim3(1024), dim3(256), 0, 0, din, n / 2, dbad, dfirst);
  hipDeviceSynchronize();
  unsigned long long bad = 0; unsigned first[8];
  hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost); hipMemcpy(first, dfirst, 32, hipMemcpyDeviceToHost);
  printf("%ld pairs, %llu with a differing word\n", n / 2, bad);
  if (bad) printf("first: x0 %08x x1 %08x  h %08x / %08x  m %08x / %08x  l %08x / %08x\n", first[0], first[1], first[2], first[3], first[4], first[5], first[6], first[7]);
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  for (int kd = 0; kd < 2; ++kd) {
    long long c = 0;
    for (int rep = 0; rep < 2; ++rep) {
      if (kd == 0) hipLaunchKernelGGL(k_time<0>, dim3(256), dim3(256), 0, 0, out, cyc, 500);
      else hipLaunchKernelGGL(k_time<1>, dim3(256), dim3(256), 0, 0, out, cyc, 500);
      hipDeviceSynchronize();
    }
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s: %.1f clocks per pair of values (one wave per SIMD, vector burst)\n", kd == 0 ? "nf_split2 (nine instructions)" : "dot form (seven instructions)", (double)c / (500.0 * 8));
  }
  return 0;
}
