// Probe (design aid for DESIGN section 7, not part of the library): an fp32-grade GEMM on the bf16 matrix cores.
// x = xh + xm + xl with three bf16 (8 + 8 + 8 mantissa bits: an EXACT split of an fp32 value), same for w; the six largest
// of the nine partial products (hh, hm, mh, hl, lh, mm) on v_mfma_f32_32x32x16_bf16 (16 384 MACs in 8 passes against 2 048 in
// 16 for v_mfma_f32_32x32x2_f32), accumulated in fp32 by the instruction.  Checks, for D[32 x 32] = W[32 x K] X[K x 32]:
//   (1) the operand layout the library would use (A: lane <-> row, k = 8 (lane >> 5) + j; B: lane <-> column, same k),
//   (2) the error of the six-term product against a float64 reference, next to the fp32 MFMA's own,
//   (3) clocks per instruction of both, and of the split (VALU) per activation.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/bf16x6_probe.hip -o tools/probe/bf16x6_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 64;

// exact three-way split of eight fp32 values into packed bf16 (truncation: the remainders carry the rest)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 &h, u32x4 &m, u32x4 &l) {
  unsigned hb[8], mb[8], lb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned xb = __float_as_uint(v[j]);
    const float hi = __uint_as_float(xb & 0xFFFF0000u);
    const float r1 = v[j] - hi;
    const unsigned rb = __float_as_uint(r1);
    const float mid = __uint_as_float(rb & 0xFFFF0000u);
    const float lo = r1 - mid;
    hb[j] = xb; mb[j] = rb; lb[j] = __float_as_uint(lo);
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {  // (hi16 of value 2p+1) : (hi16 of value 2p)
    h[p] = __builtin_amdgcn_perm(hb[2 * p + 1], hb[2 * p], 0x07060302u);
    m[p] = __builtin_amdgcn_perm(mb[2 * p + 1], mb[2 * p], 0x07060302u);
    l[p] = __builtin_amdgcn_perm(lb[2 * p + 1], lb[2 * p], 0x07060302u);
  }
}
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// W: [32][K] row-major, X: [K][32]; out6: six-term bf16 product, out32: fp32 MFMA, both [32][32] row-major
__global__ __launch_bounds__(64) void k_check(const float *W, const float *X, float *out6, float *out32, long long *cyc) {
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  f32x16 acc6 = {0}, acc32 = {0};
  long long t0 = clock64();
  for (int g = 0; g < K / 16; ++g) {
    float wv[8], xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      wv[j] = W[l31 * K + g * 16 + 8 * hi + j];    // A: row m = l31, k = 16 g + 8 hi + j
      xv[j] = X[(g * 16 + 8 * hi + j) * 32 + l31]; // B: column n = l31, same k
    }
    u32x4 wh, wm, wl, xh, xm, xl;
    split8(wv, wh, wm, wl);
    split8(xv, xh, xm, xl);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wl), as_bf(xh), acc6, 0, 0, 0);  // smallest terms first
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wh), as_bf(xl), acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wm), as_bf(xm), acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wm), as_bf(xh), acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wh), as_bf(xm), acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wh), as_bf(xh), acc6, 0, 0, 0);
  }
  long long t1 = clock64();
  for (int t = 0; t < K / 2; ++t)  // fp32 MFMA: k-step t contracts k = 2 t + hi
    acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(W[l31 * K + 2 * t + hi], X[(2 * t + hi) * 32 + l31], acc32, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
    out6[m * 32 + l31] = acc6[r];
    out32[m * 32 + l31] = acc32[r];
  }
  if (lane == 0) cyc[0] = t1 - t0;
}

// throughput: MODE 0 fp32 32x32x2 chain, 1 bf16 32x32x16 chain, 2 split8 only (VALU cost per 8 values)
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float *out, long long *cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc = {0};
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = lane * 0.37f + j * 1.01f;
  u32x4 h = {1, 2, 3, 4}, m = {5, 6, 7, 8}, l = {9, 10, 11, 12};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 7], v[(u + 1) & 7], acc, 0, 0, 0);
      if (MODE == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(h), as_bf(m), acc, 0, 0, 0);
      if (MODE == 2) {
        split8(v, h, m, l);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * 1.0001f + __uint_as_float((h[j & 3] ^ m[j & 3] ^ l[j & 3]) & 0x007FFFFFu);
      }
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s + h[0] + m[1] + l[2];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  std::vector<float> W(32 * K), X(K * 32);
  srand(1);
  auto rnd = []() { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0); };
  for (auto &w : W) w = rnd() * 0.3f;
  for (auto &x : X) x = rnd() * 2.0f;
  float *dW, *dX, *d6, *d32, *dout; long long *dc;
  hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&d6, 4096); hipMalloc(&d32, 4096); hipMalloc(&dc, 64);
  hipMalloc(&dout, 256 * 256 * 4);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dW, dX, d6, d32, dc);
  std::vector<float> o6(1024), o32(1024);
  hipMemcpy(o6.data(), d6, 4096, hipMemcpyDeviceToHost);
  hipMemcpy(o32.data(), d32, 4096, hipMemcpyDeviceToHost);
  double e6 = 0, e32 = 0, scale = 0, ecpu = 0;
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      double ref = 0, absum = 0;
      float f = 0.f;
      for (int k = 0; k < K; ++k) {
        ref += (double)W[m * K + k] * (double)X[k * 32 + n];
        absum += std::fabs((double)W[m * K + k] * (double)X[k * 32 + n]);
        f = fmaf(W[m * K + k], X[k * 32 + n], f);
      }
      e6 = std::fmax(e6, std::fabs(o6[m * 32 + n] - ref) / absum);
      e32 = std::fmax(e32, std::fabs(o32[m * 32 + n] - ref) / absum);
      ecpu = std::fmax(ecpu, std::fabs((double)f - ref) / absum);
      scale = std::fmax(scale, absum);
    }
  printf("K = %d: max |error| / sum|terms|:  six-term bf16 %.3e   fp32 MFMA %.3e   sequential fp32 fma (CPU) %.3e   (2^-24 = %.3e)\n", K, e6, e32, ecpu, std::ldexp(1.0, -24));
  long long c;
  for (int mode = 0; mode < 3; ++mode) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(256), 0, 0, dout, dc, iters);
      if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(256), 0, 0, dout, dc, iters);
      if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(256), 0, 0, dout, dc, iters);
      hipDeviceSynchronize();
    }
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const char *names[] = {"v_mfma_f32_32x32x2_f32 (2 048 MACs)", "v_mfma_f32_32x32x16_bf16 (16 384 MACs)", "split of 8 fp32 into 3 x 8 bf16 (+ 8 dependent fma)"};
    printf("%-56s %7.1f clk each, one wave per SIMD\n", names[mode], (double)c / (iters * 16.0));
  }
  return 0;
}
