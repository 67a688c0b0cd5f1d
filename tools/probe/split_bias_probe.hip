// Probe (design aid, not part of the library): is the six-term bf16 product BIASED?  VERDICT r4 weak 1 measured the device
// 2 x worse than plain fp32 after round 4 moved the GEMMs to the bf16 cores with a TRUNCATING three-way split.  This probe
// runs NT independent 32 x K x 32 GEMMs (positive operands, so that a one-sided error shows as a mean) through
//   (a) truncating split, six products       (round 4)
//   (b) round-to-nearest split, six products (round 5, nf_split2 in nf_mfma.h)
//   (c) round-to-nearest split, eight products (+ m l', l m')
//   (d) the fp32 MFMA chain
// and prints mean signed error and rms error in units of 2^-24 |result| against float64, plus -- for the instruction itself --
// the error of ONE v_mfma_f32_32x32x16_bf16 on exact bf16 inputs against the correctly rounded sum of its 16 products + C
// (does the instruction's internal adder round or truncate?).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/split_bias_probe.hip -o tools/probe/split_bias_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef PK
#define PK 64
#endif
constexpr int K = PK, NT = 512;

template <bool RN>
__device__ __forceinline__ void split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
#pragma clang fp contract(off)
  const f32x2 x = {x0, x1};
  f32x2 lo;
  if (RN) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    const u32x2 hw = {h << 16, h & 0xFFFF0000u};
    const f32x2 r = x - __builtin_bit_cast(f32x2, hw);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const u32x2 mw = {m << 16, m & 0xFFFF0000u};
    lo = r - __builtin_bit_cast(f32x2, mw);
  } else {
    const u32x2 xb = __builtin_bit_cast(u32x2, x);
    const f32x2 r = x - __builtin_bit_cast(f32x2, xb & 0xFFFF0000u);
    const u32x2 rb = __builtin_bit_cast(u32x2, r);
    lo = r - __builtin_bit_cast(f32x2, rb & 0xFFFF0000u);
    h = __builtin_amdgcn_perm(xb.y, xb.x, 0x07060302u);
    m = __builtin_amdgcn_perm(rb.y, rb.x, 0x07060302u);
  }
  const u32x2 lb = __builtin_bit_cast(u32x2, lo);
  l = __builtin_amdgcn_perm(lb.y, lb.x, 0x07060302u);
}
template <bool RN>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 &h, u32x4 &m, u32x4 &l) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned a, b, c;
    split2<RN>(v[2 * p], v[2 * p + 1], a, b, c);
    h[p] = a; m[p] = b; l[p] = c;
  }
}
__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// W: [NT][32][K], X: [NT][K][32]; out: [NF][NT][32][32]
constexpr int NF = 7;
__global__ __launch_bounds__(64) void k_forms(const float *W, const float *X, float *out) {
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5, t = blockIdx.x;
  W += (size_t)t * 32 * K; X += (size_t)t * K * 32;
  f32x16 a6t = {0}, a6r = {0}, a8r = {0}, a32 = {0}, asm_ = {0}, abg = {0}, apos = {0}, aneg = {0}, a6l = {0};
  for (int g = 0; g < K / 16; ++g) {
    float wv[8], xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      wv[j] = W[l31 * K + g * 16 + 8 * hi + j];
      xv[j] = X[(g * 16 + 8 * hi + j) * 32 + l31];
    }
    u32x4 wh, wm, wl, xh, xm, xl;
    split8<false>(wv, wh, wm, wl); split8<false>(xv, xh, xm, xl);
    a6t = mm(wl, xh, a6t); a6t = mm(wh, xl, a6t); a6t = mm(wm, xm, a6t); a6t = mm(wm, xh, a6t); a6t = mm(wh, xm, a6t); a6t = mm(wh, xh, a6t);
    split8<true>(wv, wh, wm, wl); split8<true>(xv, xh, xm, xl);
    a6r = mm(wl, xh, a6r); a6r = mm(wh, xl, a6r); a6r = mm(wm, xm, a6r); a6r = mm(wm, xh, a6r); a6r = mm(wh, xm, a6r); a6r = mm(wh, xh, a6r);
    // (e) the three small products in an accumulator of their own, added once at the end
    asm_ = mm(wl, xh, asm_); asm_ = mm(wh, xl, asm_); asm_ = mm(wm, xm, asm_);
    abg = mm(wm, xh, abg); abg = mm(wh, xm, abg); abg = mm(wh, xh, abg);
    // (f) even k-groups with +w into one accumulator, odd ones with -w into another; result = difference
    {
      u32x4 nh = wh ^ 0x80008000u, nm = wm ^ 0x80008000u, nl = wl ^ 0x80008000u;
      if (g & 1) { aneg = mm(nl, xh, aneg); aneg = mm(nh, xl, aneg); aneg = mm(nm, xm, aneg); aneg = mm(nm, xh, aneg); aneg = mm(nh, xm, aneg); aneg = mm(nh, xh, aneg); }
      else { apos = mm(wl, xh, apos); apos = mm(wh, xl, apos); apos = mm(wm, xm, apos); apos = mm(wm, xh, apos); apos = mm(wh, xm, apos); apos = mm(wh, xh, apos); }
    }
    // (g) largest products first
    a6l = mm(wh, xh, a6l); a6l = mm(wh, xm, a6l); a6l = mm(wm, xh, a6l); a6l = mm(wm, xm, a6l); a6l = mm(wh, xl, a6l); a6l = mm(wl, xh, a6l);
    a8r = mm(wm, xl, a8r); a8r = mm(wl, xm, a8r);
    a8r = mm(wl, xh, a8r); a8r = mm(wh, xl, a8r); a8r = mm(wm, xm, a8r); a8r = mm(wm, xh, a8r); a8r = mm(wh, xm, a8r); a8r = mm(wh, xh, a8r);
  }
  for (int s = 0; s < K / 2; ++s) a32 = __builtin_amdgcn_mfma_f32_32x32x2f32(W[l31 * K + 2 * s + hi], X[(2 * s + hi) * 32 + l31], a32, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
    out[((size_t)(0 * NT + t) * 32 + m) * 32 + l31] = a6t[r];
    out[((size_t)(1 * NT + t) * 32 + m) * 32 + l31] = a6r[r];
    out[((size_t)(2 * NT + t) * 32 + m) * 32 + l31] = a8r[r];
    out[((size_t)(3 * NT + t) * 32 + m) * 32 + l31] = a32[r];
    out[((size_t)(4 * NT + t) * 32 + m) * 32 + l31] = abg[r] + asm_[r];
    out[((size_t)(5 * NT + t) * 32 + m) * 32 + l31] = apos[r] - aneg[r];
    out[((size_t)(6 * NT + t) * 32 + m) * 32 + l31] = a6l[r];
  }
}
// one instruction on bf16-exact inputs: A[32][16], B[16][32] given as fp32 values that ARE bf16; C[32][32]
__global__ __launch_bounds__(64) void k_one(const float *A, const float *B, const float *C, float *out) {
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5, t = blockIdx.x;
  A += (size_t)t * 512; B += (size_t)t * 512; C += (size_t)t * 1024; out += (size_t)t * 1024;
  u32x4 a, b;
  for (int p = 0; p < 4; ++p) {
    a[p] = (__float_as_uint(A[l31 * 16 + 8 * hi + 2 * p]) >> 16) | (__float_as_uint(A[l31 * 16 + 8 * hi + 2 * p + 1]) & 0xFFFF0000u);
    b[p] = (__float_as_uint(B[(8 * hi + 2 * p) * 32 + l31]) >> 16) | (__float_as_uint(B[(8 * hi + 2 * p + 1) * 32 + l31]) & 0xFFFF0000u);
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31];
  c = mm(a, b, c);
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = c[r];
}
// the library's tanh (nf_mfma.h nf_tanh: hardware exp2 + rcp) and an odd polynomial for small arguments
__device__ __forceinline__ float tanh_lib(float x) {
  const float xc = fminf(fmaxf(x, -10.f), 10.f);
  const float e2 = __expf(2.f * xc);
  return (e2 - 1.f) * __builtin_amdgcn_rcpf(e2 + 1.f);
}
__device__ __forceinline__ float tanh_em1(float x) {  // -expm1(-2|x|) / (2 + expm1(-2|x|)) with expm1 by exp2 and a small-argument polynomial
  const float ax = fminf(fabsf(x), 10.f);
  float r;
  if (ax < 0.35f) {
    const float z = ax * ax;  // odd Taylor series to x^11: truncation < 1e-9 relative at 0.35
    r = ax * (1.f + z * (-0.33333334f + z * (0.13333334f + z * (-0.053968254f + z * (0.021869488f + z * -0.0088632355f)))));
  } else {
    const float e2 = __expf(2.f * ax);
    r = (e2 - 1.f) * __builtin_amdgcn_rcpf(e2 + 1.f);
  }
  return copysignf(r, x);
}
__global__ void k_tanh(const float *x, float *o1, float *o2, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { o1[i] = tanh_lib(x[i]); o2[i] = tanh_em1(x[i]); }
}
static float bf16_rn(float x) { unsigned b; memcpy(&b, &x, 4); b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000u; float y; memcpy(&y, &b, 4); return y; }
int main() {
  std::vector<float> W((size_t)NT * 32 * K), X((size_t)NT * K * 32);
  srand(7);
  auto rnd = []() { return (float)(rand() / (double)RAND_MAX); };
  for (int mode = 0; mode < 2; ++mode) {  // 0: positive operands, 1: both signs
    for (auto &w : W) w = mode ? (rnd() * 2.f - 1.f) * 0.3f : (0.05f + rnd()) * 0.3f;
    for (auto &x : X) x = mode ? (rnd() * 2.f - 1.f) * 2.f : (0.05f + rnd()) * 2.f;
    float *dW, *dX, *dO;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dO, (size_t)NF * NT * 1024 * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_forms, dim3(NT), dim3(64), 0, 0, dW, dX, dO);
    std::vector<float> O((size_t)NF * NT * 1024);
    hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
    double mean[NF] = {0}, rms[NF] = {0}, mx[NF] = {0};
    const double u = std::ldexp(1.0, -24);
    for (int t = 0; t < NT; ++t)
      for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
          double ref = 0, ab = 0;
          for (int k = 0; k < K; ++k) { const double p = (double)W[((size_t)t * 32 + m) * K + k] * X[((size_t)t * K + k) * 32 + n]; ref += p; ab += std::fabs(p); }
          for (int f = 0; f < NF; ++f) {
            const double e = (O[((size_t)(f * NT + t) * 32 + m) * 32 + n] - ref) / (u * ab);
            mean[f] += e; rms[f] += e * e; mx[f] = std::fmax(mx[f], std::fabs(e));
          }
        }
    const double cnt = (double)NT * 1024;
    const char *nm[NF] = {"truncating split, 6 products", "round-to-nearest split, 6 products", "round-to-nearest split, 8 products", "fp32 MFMA chain", "RN, small products in own accumulator", "RN, +w even / -w odd k-groups, difference", "RN, 6 products, LARGEST first"};
    printf("%s operands, K = %d, %d GEMMs of 32 x 32; error in units of 2^-24 sum|terms|\n", mode ? "mixed-sign" : "positive", K, NT);
    for (int f = 0; f < NF; ++f) printf("  %-44s mean %+8.4f   rms %7.4f   max %7.3f\n", nm[f], mean[f] / cnt, std::sqrt(rms[f] / cnt), mx[f]);
    hipFree(dW); hipFree(dX); hipFree(dO);
  }
  {
    const int n = 1 << 20;
    std::vector<float> x(n), a(n), b(n);
    for (double sc : {0.05, 0.3, 1.0, 3.0}) {
      for (auto &v : x) v = (float)((rnd() * 2.0 - 1.0) * sc);
      float *dx, *d1, *d2;
      hipMalloc(&dx, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
      hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_tanh, dim3(n / 256), dim3(256), 0, 0, dx, d1, d2, n);
      hipMemcpy(a.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, n * 4, hipMemcpyDeviceToHost);
      double m1 = 0, r1 = 0, m2 = 0, r2 = 0, x1 = 0, x2 = 0;
      for (int i = 0; i < n; ++i) {
        const double t = std::tanh((double)x[i]);
        const double e1 = a[i] - t, e2 = b[i] - t;
        m1 += e1; r1 += e1 * e1; m2 += e2; r2 += e2 * e2; x1 = std::fmax(x1, std::fabs(e1)); x2 = std::fmax(x2, std::fabs(e2));
      }
      printf("tanh on U(-%.2f, %.2f): library (exp2 + rcp) mean %+.3e rms %.3e max %.3e | small-|x| polynomial form mean %+.3e rms %.3e max %.3e  (absolute)\n",
             sc, sc, m1 / n, std::sqrt(r1 / n), x1, m2 / n, std::sqrt(r2 / n), x2);
      hipFree(dx); hipFree(d1); hipFree(d2);
    }
  }
  // one instruction, exact inputs
  {
    const int T = 256;
    std::vector<float> A((size_t)T * 512), B((size_t)T * 512), C((size_t)T * 1024), O((size_t)T * 1024);
    for (auto &a : A) a = bf16_rn(0.05f + rnd());
    for (auto &b : B) b = bf16_rn(0.05f + rnd());
    for (auto &c : C) c = rnd() * 8.f;
    float *dA, *dB, *dC, *dO;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dO, O.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_one, dim3(T), dim3(64), 0, 0, dA, dB, dC, dO);
    hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
    double mean = 0, rms = 0, mx = 0; long exact = 0;
    for (int t = 0; t < T; ++t)
      for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
          double ref = C[(size_t)t * 1024 + m * 32 + n];
          for (int k = 0; k < 16; ++k) ref += (double)A[(size_t)t * 512 + m * 16 + k] * B[(size_t)t * 512 + k * 32 + n];
          const float got = O[(size_t)t * 1024 + m * 32 + n];
          const double ulp = std::ldexp(1.0, std::ilogb(ref) - 23);
          const double e = (got - ref) / ulp;
          mean += e; rms += e * e; mx = std::fmax(mx, std::fabs(e));
          exact += ((float)ref == got);
        }
    const double cnt = (double)T * 1024;
    printf("ONE v_mfma_f32_32x32x16_bf16 (positive bf16 inputs, C in [0, 8)): error in ulps of the result: mean %+.4f rms %.4f max %.3f; correctly rounded in %.1f %%\n",
           mean / cnt, std::sqrt(rms / cnt), mx, 100.0 * exact / cnt);
  }
  return 0;
}
