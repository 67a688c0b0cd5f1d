// nf_targets.h -- the built-in target log-densities, one feature at a time (shared by the stand-alone
// target kernels of nf_elementwise.hip and the fused ELBO forward of nf_simple.hip).
#pragma once
#include "nf_common.h"

// Built-in targets.  target_term returns feature i's additive share of log p(y) (their sum over
// i = 0..d-1 is log p) and g = d log p / d y_i.  y0, y1 are the sample's first two coordinates and
// s2 = sum_{i>=1} y_i^2 (Funnel only).  s0, s1 are the two scalar parameters of the target:
//   DIAGGAUSS  MvNormal(mu, Diagonal(var))                   test/flow.jl:43-46
//   BANANA     (b, var)       example/targets/banana.jl:58-63,77-83
//   FUNNEL     (mu, sigma)    example/targets/neal_funnel.jl:53-72 (`score` is the gradient)
//   WARPED     (sigma1, sigma2), d = 2   example/targets/warped_gaussian.jl:51-87 (with its + log r term)
//   CROSS      (mu, sigma), d = 2        example/targets/cross.jl:30-37 (components as the code builds them)
template <int kind, class T>
__device__ __forceinline__ T target_term(int d, int i, T v, T y0, T y1, T s2, const T *__restrict__ mu,
                                         const T *__restrict__ var, T s0, T s1, T &g) {
  const T L2PI = (T)1.8378770664093453;
  if (kind == NF_TARGET_DIAGGAUSS) {
    const T vv = var[i];
    const T r = v - mu[i];
    g = -r / vv;
    return (T)-0.5 * (L2PI + log(vv) + r * r / vv);
  }
  if (kind == NF_TARGET_BANANA) {
    const T y2 = y1 + s0 * y0 * y0 - s1 * s0;
    if (i == 0) {
      g = -v / s1 - (T)2 * s0 * v * y2;
      return (T)-0.5 * v * v / s1 - (log(s1) / (T)d + L2PI) * (T)d / (T)2;
    }
    if (i == 1) {
      g = -y2;
      return (T)-0.5 * y2 * y2;
    }
    g = -v;
    return (T)-0.5 * v * v;
  }
  if (kind == NF_TARGET_FUNNEL) {
    const T a = exp(-y0);
    if (i == 0) {
      const T z = (y0 - s0) / s1;
      g = -z / s1 - (T)(d - 1) / (T)2 + a * s2 / (T)2;
      return (T)-0.5 * L2PI - log(s1) - (T)0.5 * z * z - (T)0.5 * (T)(d - 1) * (L2PI + y0);
    }
    g = -a * v;
    return (T)-0.5 * a * v * v;
  }
  if (kind == NF_TARGET_WARPED) {
    const T r = sqrt(y0 * y0 + y1 * y1);
    const T th = atan2(y1, y0) + r / (T)2;
    const T c = cos(th), sn = sin(th);
    const T zx = r * c, zy = r * sn;
    const T dr = (i == 0 ? y0 : y1) / r;
    const T dth = i == 0 ? -y1 / (r * r) + y0 / ((T)2 * r) : y0 / (r * r) + y1 / ((T)2 * r);
    g = -zx / (s0 * s0) * (dr * c - r * sn * dth) - zy / (s1 * s1) * (dr * sn + r * c * dth) + dr / r;
    if (i != 0) return (T)0;
    return (T)-0.5 * (zx * zx / (s0 * s0) + zy * zy / (s1 * s1)) - L2PI - log(s0) - log(s1) + log(r);
  }
  // CROSS: equal-weight mixture of 4 diagonal Gaussians
  const T mx[4] = {(T)0, -s0, s0, (T)0}, my[4] = {s0, (T)1, (T)1, -s0};
  const T sx[4] = {s1, (T)1, (T)1, s1}, sy[4] = {(T)1, s1, s1, (T)1};
  T lg[4], m = (T)-1e300;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const T a = (y0 - mx[k]) / sx[k], b = (y1 - my[k]) / sy[k];
    lg[k] = -L2PI - log(sx[k]) - log(sy[k]) - (T)0.5 * (a * a + b * b);
    m = lg[k] > m ? lg[k] : m;
  }
  T sw = 0, gw = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const T w = exp(lg[k] - m);
    sw += w;
    gw += w * (i == 0 ? -(y0 - mx[k]) / (sx[k] * sx[k]) : -(y1 - my[k]) / (sy[k] * sy[k]));
  }
  g = gw / sw;
  if (i != 0) return (T)0;
  return log((T)0.25) + m + log(sw);
}
__host__ __device__ inline bool target_needs_d2(int kind) { return kind == NF_TARGET_WARPED || kind == NF_TARGET_CROSS; }
