// nf_philox.h -- counter-based base-distribution draws shared by the element-wise sampler kernels
// and the fused ELBO-forward kernel (reference seam: _device_specific_rand,
// src/NormalizingFlows.jl:94-127; specification mirrored in oracle/nf_oracle.py).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// ---------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011); specification mirrored in oracle/nf_oracle.py
// ---------------------------------------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
    U4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
template <class T>
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, T &z0, T &z1);
template <>
__device__ __forceinline__ void box_muller<float>(uint32_t a, uint32_t b, float &z0, float &z1) {
  const float u0 = ((float)(a >> 9) + 0.5f) * 1.1920928955078125e-07f;  // 2^-23, exact in fp32
  const float u1 = ((float)(b >> 9) + 0.5f) * 1.1920928955078125e-07f;
  // hardware log / sqrt / sin / cos (v_sin_f32 and v_cos_f32 take their argument in revolutions, which is
  // what Box-Muller has).  Measured over 4M uniforms against a float64 evaluation: max abs error 9.1e-7,
  // libm's logf / sincospif give 7.3e-7 (tools/probe/boxmuller_probe.hip) -- at a quarter of the instructions.
  const float rad = __builtin_amdgcn_sqrtf(-2.0f * __logf(u0));
  z0 = rad * __builtin_amdgcn_cosf(u1);
  z1 = rad * __builtin_amdgcn_sinf(u1);
}
template <>
__device__ __forceinline__ void box_muller<double>(uint32_t a, uint32_t b, double &z0, double &z1) {
  const double u0 = ((double)(a >> 9) + 0.5) * 1.1920928955078125e-07;
  const double u1 = ((double)(b >> 9) + 0.5) * 1.1920928955078125e-07;
  const double rad = sqrt(-2.0 * log(u0));
  double s, c;
  sincospi(2.0 * u1, &s, &c);
  z0 = rad * c;
  z1 = rad * s;
}
