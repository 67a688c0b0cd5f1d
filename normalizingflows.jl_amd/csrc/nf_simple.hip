// nf_simple.hip -- PlanarLayer, RadialLayer and the mean-field Shift o Scale flow (gfx950).
//
// Reference: the layer constructors src/flows/planar_radial.jl:21-29,52-60 stack
// Bijectors.PlanarLayer / RadialLayer; their arithmetic lives in Bijectors.jl
// (planar_layer.jl / radial_layer.jl) and is corroborated in-tree by
// test/ext/CUDA/cuda.jl:12-30.  Formulas: SURVEY.md App. A.1/A.2, oracle/nf_oracle.py.
//
// These layers are HBM-bound (AI < 1 flop/B).  LPS = 16 lanes per sample: lane q owns
// features q, q+16, ...; the d-length dot products / norms are 4-step shuffle reductions
// inside the 16-lane group.  The forward/inverse kernel is chain-fused: the state stays
// in registers across all layers, the batch is read once and written once.
#include <cstdint>
#include <initializer_list>
#include <vector>

#include <type_traits>

#include "nf_common.h"
#include "nf_mfma.h"
#include "nf_philox.h"
#include "nf_targets.h"

#define LPS 16
#define SB 256
#define SPB (SB / LPS)

enum { LK_PLANAR = 0, LK_RADIAL = 1, LK_SHIFT = 2, LK_SCALE = 3 };

// Sum over the 16 lanes of a sample (= one DPP row), result in every lane: four DPP-modified moves / adds
// (lane ^ 1, lane ^ 2 by quad_perm, then row_half_mirror and row_mirror) -- no LDS crossbar (ds_bpermute) and no
// waitcnt in the per-layer dependency chain.
__device__ __forceinline__ int dpp_quad_x1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false); }  // [1,0,3,2]
__device__ __forceinline__ int dpp_quad_x2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false); }  // [2,3,0,1]
__device__ __forceinline__ int dpp_half_mirror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false); }
__device__ __forceinline__ int dpp_row_mirror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false); }
template <int STEP>
__device__ __forceinline__ int dpp_step(int v) {
  return STEP == 0 ? dpp_quad_x1(v) : STEP == 1 ? dpp_quad_x2(v) : STEP == 2 ? dpp_half_mirror(v) : dpp_row_mirror(v);
}
template <int STEP>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, dpp_step<STEP>(__builtin_bit_cast(int, v)));
}
template <int STEP>
__device__ __forceinline__ double dpp_add(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)dpp_step<STEP>((int)(unsigned)b), hi = (unsigned)dpp_step<STEP>((int)(unsigned)(b >> 32));
  return v + __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
template <class T>
__device__ __forceinline__ T g16sum(T v) {
  v = dpp_add<0>(v);
  v = dpp_add<1>(v);
  v = dpp_add<2>(v);
  v = dpp_add<3>(v);
  return v;
}
template <class T>
__device__ __forceinline__ T softplus_(T x) {
  return log1p(exp(-fabs(x))) + (x > (T)0 ? x : (T)0);
}
template <class T>
__device__ __forceinline__ T sigmoid_(T x) {
  const T e = exp(-fabs(x));
  return x >= (T)0 ? (T)1 / ((T)1 + e) : e / ((T)1 + e);
}

// Per-sample scalar math of the layers.  Float64 keeps libm; Float32 uses the hardware transcendentals
// (v_exp / v_log / v_rcp / v_sqrt: ~1 ulp, absolute error of tanh and log1p <= 2e-7) -- these kernels are
// HBM-bound only as long as the 16 lanes of a sample do not spend their time in libm's fp32 expansions.
template <class T>
struct Fm {
  static __device__ __forceinline__ T tanh_(T x) { return tanh(x); }
  static __device__ __forceinline__ T log_(T x) { return log(x); }
  static __device__ __forceinline__ T log1p_(T x) { return log1p(x); }
  static __device__ __forceinline__ T sqrt_(T x) { return sqrt(x); }
  static __device__ __forceinline__ T div_(T a, T b) { return a / b; }
};
template <>
struct Fm<float> {
  static __device__ __forceinline__ float tanh_(float x) {
    const float e = __expf(2.f * x);  // inf -> 1, 0 -> -1
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
  }
  static __device__ __forceinline__ float log_(float x) { return nf_log(x); }
  static __device__ __forceinline__ float log1p_(float x) { return nf_log(1.f + x); }
  static __device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
};

struct SimpleArgs {
  int kind;     // NF_KIND_*
  int d;
  int nl;       // number of flat layers of the flow
  int lo, hi;   // flat layer range to apply
  int inverse;
  int vec;      // rows may be accessed with 16-byte (8-byte for two-element chunks) vector loads / stores
  long N;
};

__host__ __device__ inline int layer_kind(int flow_kind, int l) {
  if (flow_kind == NF_KIND_PLANAR) return LK_PLANAR;
  if (flow_kind == NF_KIND_RADIAL) return LK_RADIAL;
  return l == 0 ? LK_SHIFT : LK_SCALE;  // Shift is the outer function (test/interface.jl:23-25)
}
__host__ __device__ inline long layer_off(int flow_kind, int d, int l) {
  if (flow_kind == NF_KIND_PLANAR) return (long)l * (2 * d + 1);
  if (flow_kind == NF_KIND_RADIAL) return (long)l * (d + 2);
  return (long)l * d;
}
// stride of one layer's cache / gradient slab: 2d + 2 entries, rounded to 16-byte rows
__host__ __device__ inline int lp_of(int d) { return (2 * d + 2 + 3) & ~3; }

// ---- row access ---------------------------------------------------------------------------------
// Lane q of a sample's 16-lane group owns the CONTIGUOUS features [q DPL, (q+1) DPL): one dwordx4 (dwordx2)
// per lane and row instead of DPL strided dwords -- a wave instruction then moves 1 KB of four consecutive
// rows.  Vector access needs d % VW == 0 (every vector wholly inside or outside the row, 16-byte alignment
// of every row); otherwise the element-wise path is taken.  Works on global and LDS pointers alike.
template <class T, int DPL>
struct RowVec {
  static constexpr int VW = (DPL * (int)sizeof(T) >= 16) ? 16 / (int)sizeof(T) : DPL;
};
template <class T, int DPL>
__device__ __forceinline__ void row_load(const T *row, int i0, int d, bool vec, T (&z)[DPL]) {
  constexpr int VW = RowVec<T, DPL>::VW;
  if constexpr (VW > 1) {
    if (vec) {
      typedef T V __attribute__((ext_vector_type(VW)));
#pragma unroll
      for (int k = 0; k < DPL; k += VW) {
        if (i0 + k < d) {
          const V v = *reinterpret_cast<const V *>(row + i0 + k);
#pragma unroll
          for (int u = 0; u < VW; ++u) z[k + u] = v[u];
        } else {
#pragma unroll
          for (int u = 0; u < VW; ++u) z[k + u] = (T)0;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < DPL; ++k) z[k] = (i0 + k < d) ? row[i0 + k] : (T)0;
}
template <class T, int DPL>
__device__ __forceinline__ void row_store(T *row, int i0, int d, bool vec, const T (&z)[DPL]) {
  constexpr int VW = RowVec<T, DPL>::VW;
  if constexpr (VW > 1) {
    if (vec) {
      typedef T V __attribute__((ext_vector_type(VW)));
#pragma unroll
      for (int k = 0; k < DPL; k += VW)
        if (i0 + k < d) {
          V v;
#pragma unroll
          for (int u = 0; u < VW; ++u) v[u] = z[k + u];
          *reinterpret_cast<V *>(row + i0 + k) = v;
        }
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < DPL; ++k)
    if (i0 + k < d) row[i0 + k] = z[k];
}

// per-layer cache in LDS, stride LP = lp_of(d):
//   planar: w[d] | uhat[d] | b | 1 + c      (get_u_hat: test/ext/CUDA/cuda.jl:12-18; c = w'uhat = softplus(w'u) - 1.
//           1 + c = softplus(w'u) is kept instead of c: the Jacobian factor 1 + c sech^2 = (1+c) sech^2 + tanh^2 is
//           then a sum of non-negative terms -- no cancellation as c -> -1, where 1 + c (1 - t^2) loses every digit)
//   radial: z0[d] | alpha | beta_hat
//   shift : a[d]
//   scale : a[d] | sum(log|a|)
template <class T>
__device__ void build_layer_cache(T *cache, const SimpleArgs &a, const T *__restrict__ theta) {
  const int d = a.d, LP = lp_of(d);
  for (int l = a.lo + (int)threadIdx.x; l < a.hi; l += blockDim.x) {
    T *c = cache + (long)(l - a.lo) * LP;
    const T *p = theta + layer_off(a.kind, d, l);
    const int lk = layer_kind(a.kind, l);
    if (lk == LK_PLANAR) {
      T wu = 0, ww = 0;
      for (int i = 0; i < d; ++i) {
        wu += p[i] * p[d + i];
        ww += p[i] * p[i];
      }
      const T scale = (softplus_(-wu) - (T)1) / ww;
      for (int i = 0; i < d; ++i) {
        c[i] = p[i];
        c[d + i] = p[d + i] + scale * p[i];
      }
      c[2 * d] = p[2 * d];
      c[2 * d + 1] = softplus_(wu);
    } else if (lk == LK_RADIAL) {
      const T alpha = softplus_(p[0]);
      for (int i = 0; i < d; ++i) c[i] = p[2 + i];
      c[d] = alpha;
      c[d + 1] = -alpha + softplus_(p[1]);
    } else {
      T sl = 0;
      for (int i = 0; i < d; ++i) {
        c[i] = p[i];
        sl += log(fabs(p[i]));
      }
      c[d] = sl;
    }
  }
}

template <class T, int DPL>
__device__ __forceinline__ T layer_forward(int lk, const T *c, int d, int i0, bool vec, T (&z)[DPL]) {
  T p0[DPL];
  row_load<T, DPL>(c, i0, d, vec, p0);  // planar: w, radial: z0, shift / scale: a   (0 beyond d)
  if (lk == LK_PLANAR) {
    T dot = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) dot += p0[k] * z[k];
    const T t = Fm<T>::tanh_(g16sum(dot) + c[2 * d]);
    T uh[DPL];
    row_load<T, DPL>(c + d, i0, d, vec, uh);
#pragma unroll
    for (int k = 0; k < DPL; ++k) z[k] += uh[k] * t;
    return Fm<T>::log_(c[2 * d + 1] * ((T)1 - t * t) + t * t);
  }
  if (lk == LK_RADIAL) {
    const T alpha = c[d], bh = c[d + 1];
    T ss = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const T dl = (i0 + k < d) ? z[k] - p0[k] : (T)0;
      ss += dl * dl;
    }
    const T r = Fm<T>::sqrt_(g16sum(ss));
    const T h = Fm<T>::div_((T)1, alpha + r);
#pragma unroll
    for (int k = 0; k < DPL; ++k)
      if (i0 + k < d) z[k] += bh * h * (z[k] - p0[k]);
    return (T)(d - 1) * Fm<T>::log1p_(bh * h) + Fm<T>::log1p_(bh * h - bh * h * h * r);
  }
  if (lk == LK_SHIFT) {
#pragma unroll
    for (int k = 0; k < DPL; ++k) z[k] += p0[k];
    return (T)0;
  }
#pragma unroll
  for (int k = 0; k < DPL; ++k) z[k] *= p0[k];
  return c[d];
}

template <class T, int DPL>
__device__ __forceinline__ T layer_inverse(int lk, const T *c, int d, int i0, bool vec, T (&z)[DPL]) {
  T p0[DPL];
  row_load<T, DPL>(c, i0, d, vec, p0);
  if (lk == LK_PLANAR) {
    // solve alpha + c tanh(alpha + b) = w'y for alpha = w'z  (monotone: c > -1)
    T dot = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) dot += p0[k] * z[k];
    const T wy = g16sum(dot), b = c[2 * d], sp = c[2 * d + 1], cc = sp - (T)1;
    // f(al) = al + c tanh(al + b) - w'y is increasing (f' = (1+c) sech^2 + tanh^2 > 0), root in [w'y - |c|, w'y + |c|]:
    // a fixed bisection schedule, then two Newton steps.  (A bracketed Newton with a wave-uniform early exit was
    // tried: Newton 2-cycles that stay inside the bracket for large c > 0 need the full rtsafe step test, and with it
    // the average iteration count in Float64 is no better than this schedule.)
    T lo = wy - fabs(cc), hi = wy + fabs(cc);
    const int iters = sizeof(T) == 8 ? 64 : 40;
    for (int it = 0; it < iters; ++it) {
      const T mid = (T)0.5 * (lo + hi);
      const T f = mid + cc * Fm<T>::tanh_(mid + b) - wy;
      if (f > (T)0) hi = mid; else lo = mid;
    }
    T al = (T)0.5 * (lo + hi);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {  // Newton polish
      const T t = Fm<T>::tanh_(al + b);
      al -= Fm<T>::div_(al + cc * t - wy, sp * ((T)1 - t * t) + t * t);
    }
    const T t = Fm<T>::tanh_(al + b);
    T uh[DPL];
    row_load<T, DPL>(c + d, i0, d, vec, uh);
#pragma unroll
    for (int k = 0; k < DPL; ++k) z[k] -= uh[k] * t;
    return -Fm<T>::log_(sp * ((T)1 - t * t) + t * t);
  }
  if (lk == LK_RADIAL) {
    const T alpha = c[d], bh = c[d + 1];
    T ss = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const T dl = (i0 + k < d) ? z[k] - p0[k] : (T)0;
      ss += dl * dl;
    }
    const T rho = Fm<T>::sqrt_(g16sum(ss));
    const T aa = (alpha + bh) - rho;
    const T r = (T)0.5 * (Fm<T>::sqrt_(aa * aa + (T)4 * alpha * rho) - aa);
    const T f = Fm<T>::div_(alpha + r, alpha + bh + r);
#pragma unroll
    for (int k = 0; k < DPL; ++k)
      if (i0 + k < d) z[k] = p0[k] + f * (z[k] - p0[k]);
    const T h = Fm<T>::div_((T)1, alpha + r);
    return -((T)(d - 1) * Fm<T>::log1p_(bh * h) + Fm<T>::log1p_(bh * h - bh * h * h * r));
  }
  if (lk == LK_SHIFT) {
#pragma unroll
    for (int k = 0; k < DPL; ++k) z[k] -= p0[k];
    return (T)0;
  }
#pragma unroll
  for (int k = 0; k < DPL; ++k)
    if (i0 + k < d) z[k] /= p0[k];
  return -c[d];
}

// The ELBO forward of a training step, fused into the chain kernel (fu.on): the base draws are generated in
// registers (Philox4x32-10 + Box-Muller, same counters as k_base_sample: (sample, feature group of 4, stream)) or
// read from xs, log q0 and the log-det never touch memory, and the target log-density, ybar = gscale grad log p(y)
// and the block's partial sum of pscale * elbo_j come out of the same launch.  The flow output is not written:
// the reverse pass needs only ybar and the per-layer inputs (stash).
struct SimpleFused {
  int on, draw, tkind;
  uint32_t k0, k1, stream;
  uint64_t off;
  const void *mu, *var;
  double s0, s1, gscale, pscale;
  void *gbar;       // [N][d]
  double *partial;  // [gridDim.x]
};

template <class T, int DPL>
__device__ __forceinline__ void draw_row(const SimpleFused &fu, long j, int i0, int d, T (&z)[DPL]) {
  const uint64_t gj = fu.off + (uint64_t)j;
  constexpr int NG = DPL >= 4 ? DPL / 4 : 1;
#pragma unroll
  for (int m = 0; m < NG; ++m) {
    const int g = i0 / 4 + m;
    T n4[4] = {(T)0, (T)0, (T)0, (T)0};
    if (4 * g < d) philox_normals4<T>(gj, (uint32_t)g, fu.stream, fu.k0, fu.k1, n4);
    if constexpr (DPL >= 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) z[4 * m + e] = (i0 + 4 * m + e < d) ? n4[e] : (T)0;
    } else {
      const int e0 = i0 & 3;  // DPL = 1, 2: this lane's share of the group of four
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        T v = n4[0];
#pragma unroll
        for (int e = 1; e < 4; ++e) v = (e0 + k == e) ? n4[e] : v;
        z[k] = (i0 + k < d) ? v : (T)0;
      }
    }
  }
}

// chain-fused forward / inverse over flat layers [lo, hi).  If `stash` != nullptr the reverse pass's
// evaluation points are left behind: forward -- the INPUT of every layer at stash[e][N][d] (execution
// index e); inverse -- the OUTPUT of every inverse layer at stash[l][N][d] (flat index l), i.e. the
// forward-sense input of layer l, which is where the inverse chain's reverse pass evaluates it.
template <class T, int DPL>
__global__ __launch_bounds__(SB) void k_simple_apply(SimpleArgs a, const T *__restrict__ theta, const T *x,
                                                     T *y, T *__restrict__ ladj, T *__restrict__ stash, SimpleFused fu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *cache = (T *)smem;
  const int d = a.d, LP = lp_of(d), nlr = a.hi - a.lo;
  const bool vec = a.vec != 0;
  build_layer_cache<T>(cache, a, theta);
  __syncthreads();
  const int q = threadIdx.x & (LPS - 1), i0 = q * DPL;
  double contrib = 0.0;
  // a 16-lane group shares j, so it is converged for the shuffles
  for (long j = (long)blockIdx.x * SPB + threadIdx.x / LPS; j < a.N; j += (long)gridDim.x * SPB) {
    T z[DPL];
    if (fu.draw) draw_row<T, DPL>(fu, j, i0, d, z);
    else row_load<T, DPL>(x + j * d, i0, d, vec, z);
    T logq = 0;
    if (fu.on) {
      T ss = 0;
#pragma unroll
      for (int k = 0; k < DPL; ++k) ss += z[k] * z[k];
      logq = (T)(-0.5 * 1.8378770664093453 * d) - (T)0.5 * g16sum(ss);
    }
    T lsum = 0;
    for (int e = 0; e < nlr; ++e) {
      const int l = a.inverse ? a.lo + e : a.hi - 1 - e;
      if (stash && !a.inverse) row_store<T, DPL>(stash + ((long)e * a.N + j) * d, i0, d, vec, z);
      const T *c = cache + (long)(l - a.lo) * LP;
      const int lk = layer_kind(a.kind, l);
      lsum += a.inverse ? layer_inverse<T, DPL>(lk, c, d, i0, vec, z) : layer_forward<T, DPL>(lk, c, d, i0, vec, z);
      if (stash && a.inverse) row_store<T, DPL>(stash + ((long)l * a.N + j) * d, i0, d, vec, z);
    }
    if (y) row_store<T, DPL>(y + j * d, i0, d, vec, z);
    if (q == 0 && ladj) ladj[j] = lsum;
    if (fu.on) {
      const T y0 = __shfl(z[0], 0, LPS);
      const T y1 = DPL >= 2 ? __shfl(z[DPL >= 2 ? 1 : 0], 0, LPS) : __shfl(z[0], 1, LPS);
      T s2 = 0;
      if (fu.tkind == NF_TARGET_FUNNEL) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) s2 += (i0 + k >= 1) ? z[k] * z[k] : (T)0;
        s2 = g16sum(s2);
      }
      T acc = 0, gr[DPL];
      auto run = [&](auto kc) {  // the target kind is resolved once per sample, outside the feature loop
        constexpr int KD = decltype(kc)::value;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
          T gk = 0;
          if (i0 + k < d)
            acc += target_term<KD, T>(d, i0 + k, z[k], y0, y1, s2, (const T *)fu.mu, (const T *)fu.var, (T)fu.s0, (T)fu.s1, gk);
          gr[k] = (T)fu.gscale * gk;
        }
      };
      switch (fu.tkind) {
        case NF_TARGET_DIAGGAUSS: run(std::integral_constant<int, NF_TARGET_DIAGGAUSS>{}); break;
        case NF_TARGET_BANANA: run(std::integral_constant<int, NF_TARGET_BANANA>{}); break;
        case NF_TARGET_FUNNEL: run(std::integral_constant<int, NF_TARGET_FUNNEL>{}); break;
        case NF_TARGET_WARPED: run(std::integral_constant<int, NF_TARGET_WARPED>{}); break;
        default: run(std::integral_constant<int, NF_TARGET_CROSS>{}); break;
      }
      acc = g16sum(acc);
      if (fu.gbar) row_store<T, DPL>((T *)fu.gbar + j * d, i0, d, vec, gr);
      if (q == 0) contrib += fu.pscale * (double)(acc - logq + lsum);
    }
  }
  if (fu.on) {  // deterministic block sum of the ELBO terms
    __shared__ double sm[SB / 64];
    double c = contrib;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < SB / 64; ++w) t += sm[w];
      fu.partial[blockIdx.x] = t;
    }
  }
}

// Reverse pass of ONE layer for one sample (a 16-lane group): z = layer input, g: in = dL/d(output), out =
// dL/d(input); raw parameter sums are accumulated per lane:
//   planar: acc0 = wbar_raw[d], acc1 = uhat_bar[d], s0 = bbar, s1 = cbar
//   radial: acc0 = z0bar[d], s0 = alpha_bar, s1 = betahat_bar
//   shift : acc0 = abar[d]            scale: acc0 = sum(ybar .* x)[d], s0 = sum(lbar)
//
// INV: reverse pass of the INVERSE layer (forward-KL training).  z = the inverse layer's OUTPUT w (stash of
// the inverse pass), g in = cotangent of w, out = cotangent vbar of the inverse layer's input, lb = cotangent
// of ladj_inv.  Implicit-function form:
//   vbar = J^-T (wbar - lbar grad_w ladj_fwd),  parameter sums = the forward formulas with (-vbar, -lbar);
// J^-T is closed-form for every layer here (Sherman-Morrison for planar / radial).
template <class T, int DPL, bool INV>
__device__ __forceinline__ void layer_bwd(int lk, const T *c, int d, int i0, int q, bool vec, T (&z)[DPL], T (&g)[DPL], T lb,
                                          T (&acc0)[DPL], T (&acc1)[DPL], T &s0, T &s1) {
  T p0[DPL], v[DPL];
  row_load<T, DPL>(c, i0, d, vec, p0);
#pragma unroll
  for (int k = 0; k < DPL; ++k) v[k] = (T)0;
  if (lk == LK_PLANAR) {
    T uh[DPL];
    row_load<T, DPL>(c + d, i0, d, vec, uh);
    T dot = 0, ug = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      dot += p0[k] * z[k];
      ug += uh[k] * g[k];
    }
    const T sp = c[2 * d + 1], cc = sp - (T)1;
    const T t = Fm<T>::tanh_(g16sum(dot) + c[2 * d]);
    ug = g16sum(ug);
    const T gg = (T)1 - t * t, D = sp * gg + t * t, iD = Fm<T>::div_((T)1, D);
    if (INV) {
      // J^T = I + gg w uhat^T, grad_z ladj = kap w, uhat^T w = cc:  vbar = g - w beta
      const T kap = -(T)2 * cc * t * gg * iD;
      const T uap = ug - lb * kap * cc;
      const T beta = lb * kap + gg * uap * iD;
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        v[k] = g[k] - p0[k] * beta;
        g[k] = -v[k];
      }
      ug = -(ug - cc * beta);
      lb = -lb;
    }
    const T ab = ug * gg - (T)2 * lb * cc * t * gg * iD;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      acc0[k] += ab * z[k];  // wbar_raw
      acc1[k] += t * g[k];   // uhat_bar
      g[k] += p0[k] * ab;    // zbar
    }
    if (q == 0) {
      s0 += ab;            // bbar
      s1 += lb * gg * iD;  // cbar
    }
  } else if (lk == LK_RADIAL) {
    const T alpha = c[d], bh = c[d + 1];
    T ss = 0, yd = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      z[k] = (i0 + k < d) ? z[k] - p0[k] : (T)0;  // delta
      ss += z[k] * z[k];
      yd += g[k] * z[k];
    }
    const T r = Fm<T>::sqrt_(g16sum(ss));
    yd = g16sum(yd);
    const T h = Fm<T>::div_((T)1, alpha + r);
    const T qq = bh * h, bah2 = bh * alpha * h * h;
    const T iq = Fm<T>::div_((T)1, (T)1 + qq), ib = Fm<T>::div_((T)1, (T)1 + bah2), ir = r > (T)0 ? Fm<T>::div_((T)1, r) : (T)0;
    const T dL_dh = (T)(d - 1) * bh * iq + (T)2 * bh * alpha * h * ib;
    const T dL_db = (T)(d - 1) * h * iq + alpha * h * h * ib;
    const T dL_da = bh * h * h * ib;
    if (INV) {
      // J = A I + Bc delta delta^T (symmetric), A + Bc r^2 = 1 + bah2, grad_z ladj = -dL_dh h^2 delta / r
      const T e = lb * dL_dh * h * h * ir;
      const T dap = yd + lb * dL_dh * h * h * r;
      const T f2 = (-bh * h * h * ir) * dap * ib;
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        v[k] = (g[k] + (e - f2) * z[k]) * iq;  // 1 / A, A = 1 + qq
        g[k] = -v[k];
      }
      yd = -dap * ib;
      lb = -lb;
    }
    const T hbar = bh * yd + lb * dL_dh;
    const T rbar_over_r = -h * h * hbar * ir;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const T db = qq * g[k] + rbar_over_r * z[k];
      acc0[k] -= db;  // z0bar
      g[k] += db;     // zbar
    }
    if (q == 0) {
      s0 += -h * h * hbar + lb * dL_da;  // alpha_bar
      s1 += h * yd + lb * dL_db;         // betahat_bar
    }
  } else if (lk == LK_SHIFT) {
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      acc0[k] += INV ? -g[k] : g[k];
      v[k] = g[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      if (INV) {
        v[k] = (i0 + k < d) ? g[k] / p0[k] : (T)0;
        acc0[k] -= v[k] * z[k];
      } else {
        acc0[k] += g[k] * z[k];
        g[k] *= p0[k];
      }
    }
    if (q == 0) s0 += INV ? -lb : lb;
  }
  if (INV) {
#pragma unroll
    for (int k = 0; k < DPL; ++k) g[k] = v[k];
  }
}

// ---- k_simple_step's form of the two layers: ONE stashed scalar per (sample, layer) ----------------------------------
// A planar layer's input is recoverable from its output and t = tanh(w'z + b):  z = y - uhat t;  a radial layer's from
// r = ||z - z0||:  z = z0 + (y - z0) / (1 + beta_hat h), h = 1 / (alpha + r).  So the training step keeps t (resp. r)
// of every layer -- one register per layer instead of DPL -- walks the state back alongside the cotangent, and the reverse
// pass starts from the stashed scalar instead of recomputing the dot product / norm, its lane reduction and the tanh /
// sqrt.  Both maps are smooth, so the reconstruction (one rounding per layer) moves nothing discretely.
template <class T, int DPL>
__device__ __forceinline__ T layer_forward_s(int lk, const T *c, int d, int i0, bool vec, T (&z)[DPL], T &keep) {
  T p0[DPL];
  row_load<T, DPL>(c, i0, d, vec, p0);
  if (lk == LK_PLANAR) {
    T dot = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) dot += p0[k] * z[k];
    const T t = Fm<T>::tanh_(g16sum(dot) + c[2 * d]);
    keep = t;
    T uh[DPL];
    row_load<T, DPL>(c + d, i0, d, vec, uh);
#pragma unroll
    for (int k = 0; k < DPL; ++k) z[k] += uh[k] * t;
    return Fm<T>::log_(c[2 * d + 1] * ((T)1 - t * t) + t * t);
  }
  const T alpha = c[d], bh = c[d + 1];
  T ss = 0;
#pragma unroll
  for (int k = 0; k < DPL; ++k) {
    const T dl = (i0 + k < d) ? z[k] - p0[k] : (T)0;
    ss += dl * dl;
  }
  const T r = Fm<T>::sqrt_(g16sum(ss));
  keep = r;
  const T h = Fm<T>::div_((T)1, alpha + r);
#pragma unroll
  for (int k = 0; k < DPL; ++k)
    if (i0 + k < d) z[k] += bh * h * (z[k] - p0[k]);
  return (T)(d - 1) * Fm<T>::log1p_(bh * h) + Fm<T>::log1p_(bh * h - bh * h * h * r);
}

// z: the layer's OUTPUT on entry, its INPUT on exit;  g: dL/d(output) -> dL/d(input);  parameter sums as layer_bwd
template <class T, int DPL>
__device__ __forceinline__ void layer_bwd_s(int lk, const T *c, int d, int i0, int q, bool vec, T (&z)[DPL], T (&g)[DPL], T lb,
                                            T keep, T (&acc0)[DPL], T (&acc1)[DPL], T &s0, T &s1) {
  T p0[DPL];
  row_load<T, DPL>(c, i0, d, vec, p0);
  if (lk == LK_PLANAR) {
    T uh[DPL];
    row_load<T, DPL>(c + d, i0, d, vec, uh);
    const T t = keep;
    T ug = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      z[k] -= uh[k] * t;  // the layer input
      ug += uh[k] * g[k];
    }
    const T sp = c[2 * d + 1], cc = sp - (T)1;
    ug = g16sum(ug);
    const T gg = (T)1 - t * t, D = sp * gg + t * t, iD = Fm<T>::div_((T)1, D);
    const T ab = ug * gg - (T)2 * lb * cc * t * gg * iD;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      acc0[k] += ab * z[k];  // wbar_raw
      acc1[k] += t * g[k];   // uhat_bar
      g[k] += p0[k] * ab;    // zbar
    }
    if (q == 0) {
      s0 += ab;            // bbar
      s1 += lb * gg * iD;  // cbar
    }
    return;
  }
  const T alpha = c[d], bh = c[d + 1];
  const T r = keep;
  const T h = Fm<T>::div_((T)1, alpha + r);
  const T qq = bh * h, bah2 = bh * alpha * h * h;
  const T iq = Fm<T>::div_((T)1, (T)1 + qq), ib = Fm<T>::div_((T)1, (T)1 + bah2), ir = r > (T)0 ? Fm<T>::div_((T)1, r) : (T)0;
  T dl[DPL], yd = 0;
#pragma unroll
  for (int k = 0; k < DPL; ++k) {
    dl[k] = (i0 + k < d) ? (z[k] - p0[k]) * iq : (T)0;  // y - z0 = delta (1 + beta_hat h)
    if (i0 + k < d) z[k] = p0[k] + dl[k];               // the layer input
    yd += g[k] * dl[k];
  }
  yd = g16sum(yd);
  const T dL_dh = (T)(d - 1) * bh * iq + (T)2 * bh * alpha * h * ib;
  const T dL_db = (T)(d - 1) * h * iq + alpha * h * h * ib;
  const T dL_da = bh * h * h * ib;
  const T hbar = bh * yd + lb * dL_dh;
  const T rbar_over_r = -h * h * hbar * ir;
#pragma unroll
  for (int k = 0; k < DPL; ++k) {
    const T db = qq * g[k] + rbar_over_r * dl[k];
    acc0[k] -= db;  // z0bar
    g[k] += db;     // zbar
  }
  if (q == 0) {
    s0 += -h * h * hbar + lb * dL_da;  // alpha_bar
    s1 += h * yd + lb * dL_db;         // betahat_bar
  }
}

// layers per pass of the reverse kernel: as many as ~64 accumulator registers per thread allow (at most 4)
template <class T, int DPL>
struct BwdCfg {
  static constexpr int RAW = 128 / (DPL * (int)sizeof(T));
  static constexpr int LPP = RAW >= 4 ? 4 : (RAW >= 2 ? 2 : 1);
};

// Reverse pass over the whole batch, every layer in one launch.  The layers are taken LPP at a time: a
// thread carries a sample's cotangent through the LPP layers of a pass in registers (gbar is read and written
// once per PASS, each layer's input once), with one set of parameter accumulators per layer of the pass.
// A thread keeps its samples from pass to pass (it re-reads the gbar it wrote), so only the block-wide
// parameter reduction needs barriers: 4 groups of a wave by shuffles, the 4 waves through LDS, in a fixed
// order (deterministic).  Raw parameter sums go to slabs[layer][block][LP] (layout: layer_bwd).
// INV: the inverse chain's reverse pass -- layers in forward execution order, see layer_bwd.
template <class T, int DPL, bool INV>
__global__ __launch_bounds__(SB, (DPL * (int)sizeof(T) <= 16 ? 4 : 1)) void k_simple_bwd_layers(SimpleArgs a, int nl, const T *__restrict__ theta,
                                                          const T *__restrict__ stash, long stash_stride,
                                                          T *__restrict__ gbar, const T *__restrict__ lbar, T lbar_const,
                                                          T *__restrict__ slabs, long slab_stride) {
  constexpr int LPP = BwdCfg<T, DPL>::LPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int d = a.d, LP = lp_of(d);
  const bool vec = a.vec != 0;
  T *cache = (T *)smem;            // [LPP][LP]
  T *red = cache + LPP * LP;       // [SB / 64][LPP * LP]
  const int q = threadIdx.x & (LPS - 1), i0 = q * DPL, grp = threadIdx.x / LPS, wave = threadIdx.x >> 6;
#pragma unroll 1
  for (int pass = 0; pass < nl; pass += LPP) {
    const int nlp = nl - pass < LPP ? nl - pass : LPP;
    // step s = pass + u  <->  flat layer l = INV ? nl-1-s : s   (forward chain: flat order = reverse of execution order)
    const int lo = INV ? nl - pass - nlp : pass;
    SimpleArgs span = a;
    span.lo = lo;
    span.hi = lo + nlp;
    build_layer_cache<T>(cache, span, theta);
    __syncthreads();
    T acc0[LPP][DPL], acc1[LPP][DPL], s0[LPP], s1[LPP];
#pragma unroll
    for (int u = 0; u < LPP; ++u) {
      s0[u] = s1[u] = (T)0;
#pragma unroll
      for (int k = 0; k < DPL; ++k) acc0[u][k] = acc1[u][k] = (T)0;
    }
    for (long j = (long)blockIdx.x * SPB + grp; j < a.N; j += (long)gridDim.x * SPB) {
      T g[DPL];
      row_load<T, DPL>(gbar + j * d, i0, d, vec, g);
      const T lb = lbar ? lbar[j] : lbar_const;
#pragma unroll
      for (int u = 0; u < LPP; ++u) {
        if (u < nlp) {
          const int l = INV ? nl - 1 - (pass + u) : pass + u;
          T z[DPL];
          row_load<T, DPL>(stash + (long)(INV ? l : nl - 1 - l) * stash_stride + j * d, i0, d, vec, z);
          layer_bwd<T, DPL, INV>(layer_kind(a.kind, l), cache + (long)(l - lo) * LP, d, i0, q, vec, z, g, lb, acc0[u],
                                 acc1[u], s0[u], s1[u]);
        }
      }
      row_store<T, DPL>(gbar + j * d, i0, d, vec, g);
    }
    // deterministic block reduction: the wave's 4 sample groups by shuffles, then the 4 waves through LDS
#pragma unroll
    for (int u = 0; u < LPP; ++u) {
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        acc0[u][k] += __shfl_xor(acc0[u][k], 16, 64);
        acc0[u][k] += __shfl_xor(acc0[u][k], 32, 64);
        acc1[u][k] += __shfl_xor(acc1[u][k], 16, 64);
        acc1[u][k] += __shfl_xor(acc1[u][k], 32, 64);
      }
      s0[u] += __shfl_xor(s0[u], 16, 64);
      s0[u] += __shfl_xor(s0[u], 32, 64);
      s1[u] += __shfl_xor(s1[u], 16, 64);
      s1[u] += __shfl_xor(s1[u], 32, 64);
    }
    if ((threadIdx.x & 63) < LPS) {
      T *mine = red + (long)wave * LPP * LP;
#pragma unroll
      for (int u = 0; u < LPP; ++u) {
#pragma unroll
        for (int k = 0; k < DPL; ++k)
          if (i0 + k < d) {
            mine[u * LP + i0 + k] = acc0[u][k];
            mine[u * LP + d + i0 + k] = acc1[u][k];
          }
        if (q == 0) {
          mine[u * LP + 2 * d] = s0[u];
          mine[u * LP + 2 * d + 1] = s1[u];
        }
      }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nlp * LP; s += SB) {
      const int u = s / LP, e = s - u * LP;
      if (e < 2 * d + 2) {
        const int l = INV ? nl - 1 - (pass + u) : pass + u;
        constexpr int NW = SB / 64;
        T vsum = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) vsum += red[(long)w * LPP * LP + s];
        slabs[(long)l * slab_stride + (long)blockIdx.x * LP + e] = vsum;
      }
    }
    __syncthreads();  // cache and reduction buffer are rebuilt for the next pass
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The whole reverse-KL training step of a planar / radial / mean-field flow in ONE launch, with no activation ever
// written to memory (src/objectives/elbo.jl:65-97 under _value_and_gradient, src/optimize.jl:86).
//
// k_simple_apply + k_simple_bwd_layers move 11 + 16 rows of d elements per sample through HBM (the per-layer input
// stash and ybar): 7.1 KB per sample at d = 64 against 516 B of algorithmic traffic.  But the draws are counter-based
// and a sample's state is DPL registers per lane, so a 16-lane group can keep the input of EVERY layer in registers
// (NL * DPL of them), evaluate the target, and walk straight back: the only global traffic left is the parameter
// slabs.  The kernel is then bound by VALU / transcendental issue, not by HBM.
//
// Register budget: NL * DPL stash + NL * (2 DPL + 2) parameter accumulators per thread.  NLMAX bounds the unrolled
// layer loops (l is a compile-time constant in every access; layers l >= nl are skipped by a wave-uniform guard);
// flows with more layers, or wider than the budget, take the stash path above.
// NLMAX: unroll bound of the layer loops, 4 or 12 (the smaller that holds the flow's layers; 2 for mean-field: unused
// slots still cost their accumulator registers); NLMAX * DPL stash elements must fit the budget of 64 (Float32) /
// 32 (Float64) per thread.
template <class T, int DPL, int KIND, int NLMAX>
__global__ __launch_bounds__(SB, 2) void k_simple_step(SimpleArgs a, const T *__restrict__ theta, const T *__restrict__ xs,
                                                    SimpleFused fu, T lbar_const, T *__restrict__ slabs, long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int d = a.d, LP = lp_of(d), nl = a.nl;
  const bool vec = a.vec != 0;
  T *cache = (T *)smem;            // [nl][LP]
  T *red = cache + (long)nl * LP;  // [SB / 64][nl * LP]
  build_layer_cache<T>(cache, a, theta);
  __syncthreads();
  const int q = threadIdx.x & (LPS - 1), i0 = q * DPL, wave = threadIdx.x >> 6;
  T acc0[NLMAX][DPL], acc1[NLMAX][DPL], s0[NLMAX], s1[NLMAX];
#pragma unroll
  for (int l = 0; l < NLMAX; ++l) {
    s0[l] = s1[l] = (T)0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) acc0[l][k] = acc1[l][k] = (T)0;
  }
  double contrib = 0.0;
  for (long j = (long)blockIdx.x * SPB + threadIdx.x / LPS; j < a.N; j += (long)gridDim.x * SPB) {
    // The layer caches are loop-invariant LDS data; with the layer loops unrolled hipcc would hoist every row of every
    // layer out of the sample loop (2 * DPL registers per layer) and spill.  The clobber keeps the reads inside.
    asm volatile("" ::: "memory");
    T z[DPL];
    if (fu.draw) draw_row<T, DPL>(fu, j, i0, d, z);
    else row_load<T, DPL>(xs + j * d, i0, d, vec, z);
    T ss = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) ss += z[k] * z[k];
    const T logq = (T)(-0.5 * 1.8378770664093453 * d) - (T)0.5 * g16sum(ss);
    // forward: layers execute last-listed first (src/flows/utils.jl:23-26).  Planar / radial: one scalar per layer is
    // kept (layer_forward_s); mean-field: zs[l] = input of flat layer l (two trivial layers)
    constexpr bool SCALAR = KIND != NF_KIND_MEANFIELD;
    T zs[SCALAR ? 1 : NLMAX][DPL];
    T ks[SCALAR ? NLMAX : 1];
    T lsum = 0;
#pragma unroll
    for (int l = NLMAX - 1; l >= 0; --l) {
      if (l < nl) {
        if constexpr (SCALAR) {
          lsum += layer_forward_s<T, DPL>(layer_kind(KIND, l), cache + (long)l * LP, d, i0, vec, z, ks[l]);
        } else {
#pragma unroll
          for (int k = 0; k < DPL; ++k) zs[l][k] = z[k];
          lsum += layer_forward<T, DPL>(layer_kind(KIND, l), cache + (long)l * LP, d, i0, vec, z);
        }
      }
    }
    // target log-density, ybar = gscale * grad log p(y), ELBO term
    T g[DPL];
    {
      const T y0 = __shfl(z[0], 0, LPS);
      const T y1 = DPL >= 2 ? __shfl(z[DPL >= 2 ? 1 : 0], 0, LPS) : __shfl(z[0], 1, LPS);
      T s2 = 0;
      if (fu.tkind == NF_TARGET_FUNNEL) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) s2 += (i0 + k >= 1) ? z[k] * z[k] : (T)0;
        s2 = g16sum(s2);
      }
      T acc = 0;
      auto run = [&](auto kc) {
        constexpr int KD = decltype(kc)::value;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
          T gk = 0;
          if (i0 + k < d)
            acc += target_term<KD, T>(d, i0 + k, z[k], y0, y1, s2, (const T *)fu.mu, (const T *)fu.var, (T)fu.s0, (T)fu.s1, gk);
          g[k] = (T)fu.gscale * gk;
        }
      };
      switch (fu.tkind) {
        case NF_TARGET_DIAGGAUSS: run(std::integral_constant<int, NF_TARGET_DIAGGAUSS>{}); break;
        case NF_TARGET_BANANA: run(std::integral_constant<int, NF_TARGET_BANANA>{}); break;
        case NF_TARGET_FUNNEL: run(std::integral_constant<int, NF_TARGET_FUNNEL>{}); break;
        case NF_TARGET_WARPED: run(std::integral_constant<int, NF_TARGET_WARPED>{}); break;
        default: run(std::integral_constant<int, NF_TARGET_CROSS>{}); break;
      }
      acc = g16sum(acc);
      if (q == 0) contrib += fu.pscale * (double)(acc - logq + lsum);
    }
    // reverse: flat order = reverse of execution order; the cotangent never leaves the registers, and (planar / radial)
    // the state z walks back from the flow output to the base draw alongside it
#pragma unroll
    for (int l = 0; l < NLMAX; ++l) {
      if (l < nl) {
        if constexpr (SCALAR)
          layer_bwd_s<T, DPL>(layer_kind(KIND, l), cache + (long)l * LP, d, i0, q, vec, z, g, lbar_const, ks[l], acc0[l],
                              acc1[l], s0[l], s1[l]);
        else
          layer_bwd<T, DPL, false>(layer_kind(KIND, l), cache + (long)l * LP, d, i0, q, vec, zs[l], g, lbar_const, acc0[l],
                                   acc1[l], s0[l], s1[l]);
      }
    }
  }
  // deterministic block reduction of the parameter sums: the wave's 4 sample groups by shuffles, then the 4 waves
  // through LDS in a fixed order; raw sums go to slabs[layer][block][LP] (layout: layer_bwd), as k_simple_bwd_layers
#pragma unroll
  for (int l = 0; l < NLMAX; ++l) {
    if (l < nl) {
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        acc0[l][k] += __shfl_xor(acc0[l][k], 16, 64);
        acc0[l][k] += __shfl_xor(acc0[l][k], 32, 64);
        acc1[l][k] += __shfl_xor(acc1[l][k], 16, 64);
        acc1[l][k] += __shfl_xor(acc1[l][k], 32, 64);
      }
      s0[l] += __shfl_xor(s0[l], 16, 64);
      s0[l] += __shfl_xor(s0[l], 32, 64);
      s1[l] += __shfl_xor(s1[l], 16, 64);
      s1[l] += __shfl_xor(s1[l], 32, 64);
      if ((threadIdx.x & 63) < LPS) {
        T *mine = red + ((long)wave * nl + l) * LP;
#pragma unroll
        for (int k = 0; k < DPL; ++k)
          if (i0 + k < d) {
            mine[i0 + k] = acc0[l][k];
            mine[d + i0 + k] = acc1[l][k];
          }
        if (q == 0) {
          mine[2 * d] = s0[l];
          mine[2 * d + 1] = s1[l];
        }
      }
    }
  }
  __syncthreads();
  for (int s = threadIdx.x; s < nl * LP; s += SB) {
    const int l = s / LP, e = s - l * LP;
    if (e < 2 * d + 2) {
      constexpr int NW = SB / 64;
      T vsum = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) vsum += red[(long)w * nl * LP + s];
      slabs[(long)l * slab_stride + (long)blockIdx.x * LP + e] = vsum;
    }
  }
  {  // deterministic block sum of the ELBO terms
    __shared__ double sm[SB / 64];
    double c = contrib;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < SB / 64; ++w) t += sm[w];
      fu.partial[blockIdx.x] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Planar training step on the matrix pipe (k_planar_step): d <= 64, up to 16 layers, Float32.
// ------------------------------------------------------------------------------------------------------------------
// A planar layer adds a multiple of ONE fixed vector: z_{e+1} = z_e + uhat_e tanh(w_e'z_e + b_e)
// (src/flows/planar_radial.jl:21-29, Bijectors.PlanarLayer).  With the layers in flat order (flat layer nl-1 executes
// first, src/flows/utils.jl:23-26) the state in front of layer l is z0 + sum_{m>l} uhat_m t_m, so
//     a_l  = w_l'z0 + b_l + sum_{m>l} C[l][m] t_m,     C[l][m] = w_l'uhat_m,        t_l = tanh(a_l),
//     y    = z0 + Uhat t,
// and in the reverse sweep (cotangent in front of layer l's output: ybar + sum_{m<l} w_m abar_m)
//     ug_l   = uhat_l'ybar + sum_{m<l} C[m][l] abar_m,      abar_l = ug_l (1 - t_l^2) + lbar dladj_l/da_l,
//     wbar_l = sum_j abar_l^j z_l^j   = [Z0 Abar']_l + sum_{m>l} uhat_m G[l][m],    G[l][m] = sum_j abar_l^j t_m^j,
//     ubar_l = sum_j t_l^j gbar_l^j   = [Ybar T']_l  + sum_{m<l} w_m G[m][l].
// Every d-length dot product, the state update and every sum over samples is therefore a GEMM with a 32-sample tile on
// one side -- W Z0, Uhat T, Uhat'Ybar, Z0 Abar', Ybar T', Abar T' -- and what is left per sample is a triangular recurrence
// over nl scalars.  The two per-layer scalar sums ride along: with q_l = lbar dladj_l/d(1+c_l) stacked under abar and a row
// of ones under t, G' = [Abar; Q] [T; 1]' holds G, bbar_l = sum_j abar_l^j (last column) and the sum of q_l.  k_simple_step (16 lanes per sample, DPP reductions, 16-fold redundant scalar math) needs ~1 900
// VALU instructions per four samples at d = 64 x 10 layers; here a wave issues ~160 MFMAs and ~1 800 VALU instructions
// per THIRTY-TWO samples (over half of those are the Philox / Box-Muller draws).  Same arithmetic as the reference up to
// the order of the additions (z_l is never formed; its inner product with w_l is summed term by term).
//
// Layout.  A lane holds sample l31 of the tile and the features f = 32 blk + nf_row(r, hi) (the 32 x 32 C layout): whole Philox
// groups of four, and z0 is the accumulator that Uhat T is added to (v_mfma_f32_32x32x2_f32: all 32 rows are features there).
// The five GEMMs with LAYERS on one side use v_mfma_f32_16x16x4_f32 (round 5; 32 clocks for a 16 x 16 block instead of 64 for a
// 32 x 32 one of which 10 rows carried layers): W Z0 and Uhat'Ybar take their sample-side operand from the wave's
// [sample][feature] LDS tile (z0 and ybar pass through it anyway: they are the lane <-> feature operands of the K = samples
// GEMMs) and hand their 16 rows to the samples' lanes through a [sample][layer] tile; Z0 Abar', Ybar T' and Abar T' accumulate
// in 16 x 16 blocks (lane = layer column, 4 registers per 16 features).  The recurrences run in both half-waves (each lane all
// layers of its sample).  Parameter sums stay in MFMA accumulators over the wave's tiles; the block epilogue adds the waves in
// a fixed order, applies the two triangular corrections and writes k_simple_step's slab layout (k_simple_finalize).
template <int DB_, int NLR_>
struct PlanarGeo {
  static constexpr int DB = DB_, NLR = NLR_;
  static constexpr int FD = 32 * DB;                                   // padded feature count
  static constexpr int NL = NLR <= 4 ? 2 * NLR : NLR <= 6 ? 10 : NLR <= 8 ? 16 : 32;  // layers carried (rows 0 .. NL-1)
  static constexpr int ECOLS = 8 * ((NLR + 3) / 4);                    // layer columns of the Uhat[f][e] image
  static constexpr int SW = FD + 4;                                    // row stride of W[e][f], Uhat'[e][f]
  static constexpr int SU = ECOLS + 4;                                 // row stride of Uhat[f][e]
  static constexpr int OFF_W = 0;
  static constexpr int OFF_UT = OFF_W + NL * SW;
  static constexpr int OFF_U = OFF_UT + NL * SW;
  static constexpr int OFF_C = OFF_U + FD * SU;                        // C[l][m], NL x NL
  static constexpr int OFF_B = OFF_C + NL * NL;                        // b[l]
  static constexpr int OFF_SP = OFF_B + NL;                            // 1 + c_l = softplus(w'u)
  static constexpr int OFF_TG = OFF_SP + NL;                           // target: mu[f] | 1/var[f] | log 2pi + log var[f]
  static constexpr int SHARED = ((OFF_TG + 3 * FD + 3) / 4) * 4;
  // per wave (round 5, the 16x16x4 form): three [sample][row] tiles.  Row strides = 4 (mod 16) floats: rows stay 16-byte aligned
  // (a lane's four consecutive features / layers move as one b128), and with the contraction order s(t, g) = (t & 3) +
  // 16 (t >> 2) + 4 g of the K = samples GEMMs the four lane groups of a k-step sit 16 banks apart (4 g * stride = 16 g mod 64)
  static constexpr int SX = FD + 4;                                    // [sample][feature]: z0, later ybar
  static constexpr int AR = 2 * NL, TR = NL + 1;                       // entries of the two [sample][layer] tiles (below)
  static constexpr int SLA = AR <= 20 ? 20 : 36, SLT = 20;
  static constexpr int OFF_X = 0;
  static constexpr int OFF_A = OFF_X + 32 * SX;                        //           abar[s][l] | q[s][NL + l]   (first: the gathered GEMM rows)
  static constexpr int OFF_T = OFF_A + 32 * SLA;                       //           t[s][l] | 1
  static constexpr int WAVE = ((OFF_T + 32 * SLT + 16 + 3) / 4) * 4;        // + 16: the second column block of G' reads past the last row
  static constexpr int RB = FD / 16;                                   // 16-feature row blocks of M1 / M2
  static constexpr int GRB = (AR + 15) / 16, GCB = (TR + 15) / 16;     // 16 x 16 blocks of G'
  static_assert(TR <= SLT && AR <= SLA && NL <= 16, "tile strides");
  // block epilogue (aliases the waves' tiles): per wave M1[l][f] | M2[l][f] | G'[2 NL][NL + 1]
  static constexpr int R_M1 = 0, R_M2 = NL * FD, R_G = 2 * NL * FD, REGION = R_G + AR * TR;
  static_assert(AR <= 32, "abar and q rows share one operand tile");
  static_assert(REGION <= WAVE, "epilogue region must fit a wave's tiles");
  static constexpr size_t lds_floats(int nl, int lp) { return (size_t)SHARED + 4 * (size_t)WAVE + (size_t)nl * lp; }
};

__device__ __forceinline__ void planar_gather(float v, float &lo, float &hi) {
  // lo = the value lane (l31, 0) holds, hi = the value lane (l31, 1) holds, in both half-waves: v_permlane32_swap_b32 of two
  // copies ([a.lo | b.lo], [a.hi | b.hi]; tools/probe/permlane_probe.hip).  Two things hipcc 7.2 does not do by itself:
  // the copy must be opaque (identical arguments are folded into one register), and a VALU read of the results needs wait
  // states after the swap -- without the s_nop the recurrences read stale rows on MI355X (loss 30.6 instead of 42.3 at
  // d = 64 x 10 layers).
  unsigned va = __builtin_bit_cast(unsigned, v), vb = va;
  asm volatile("" : "+v"(vb));
  const auto r = __builtin_amdgcn_permlane32_swap(va, vb, false, false);
  unsigned r0 = r[0], r1 = r[1];
  asm volatile("s_nop 1" : "+v"(r0), "+v"(r1));
  lo = __builtin_bit_cast(float, r0);
  hi = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float planar_xhalf_sum(float v) {
  float lo, hi;
  planar_gather(v, lo, hi);
  return lo + hi;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 planar_mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// the contraction order of the K = samples GEMMs: k-step t, lane group g <-> sample planar_s(t) + 4 g
__device__ __forceinline__ constexpr int planar_s(int t) { return (t & 3) + 16 * (t >> 2); }

// out[T][r] = sum_f rows[4 g + r][f] X[f][s16 + 16 T] for the tile's two sample halves T, as v_mfma_f32_16x16x4_f32 leaves it
// (column = lane & 15 = s16, rows 4 g .. 4 g + 3; g = lane >> 4).  `rowp` = row (lane & 15) of a [layer][feature] image + (FD / 4) g,
// `xp` = row s16 of the wave's [sample][feature] tile + (FD / 4) g: lane group g contracts features (FD / 4) g .. (FD / 4)(g + 1) - 1,
// four k-steps per ds_read_b128 on either side.  Round 5: with 10 (16) useful rows the 32 x 32 x 2 instruction spent 64 clocks on
// a product of which 10 / 32 was used; this one spends 32 on 10 / 16.
template <int FD, int S, int SX>
__device__ __forceinline__ void planar_rows_gemm16(const float *__restrict__ rowp, const float *__restrict__ xp, f32x4 (&out)[2]) {
  constexpr int NG = FD / 16;
  float wn[4], an[4], bn[4], wc[4], ac[4], bc[4];
  nf_ld4<S>(rowp, wn[0], wn[1], wn[2], wn[3]);
  nf_ld4<SX>(xp, an[0], an[1], an[2], an[3]);
  nf_ld4<SX>(xp + 16 * SX, bn[0], bn[1], bn[2], bn[3]);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wc[e] = wn[e];
      ac[e] = an[e];
      bc[e] = bn[e];
    }
    if (g + 1 < NG) {
      nf_ld4<S>(rowp + 4 * (g + 1), wn[0], wn[1], wn[2], wn[3]);
      nf_ld4<SX>(xp + 4 * (g + 1), an[0], an[1], an[2], an[3]);
      nf_ld4<SX>(xp + 16 * SX + 4 * (g + 1), bn[0], bn[1], bn[2], bn[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      out[0] = planar_mfma16(wc[e], ac[e], out[0]);
      out[1] = planar_mfma16(wc[e], bc[e], out[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// The rows of such a result to the two lanes of every sample (all NL of them: the recurrences run per sample), through the
// [sample][layer] tile `gt` (the abar tile, idle at both call sites): one b128 write per half, NL / 4 b128 reads.
template <class PG>
__device__ __forceinline__ void planar_rows_to_lanes(float *__restrict__ gt, const f32x4 (&c)[2], int s16, int lg, int l31, float (&v)[PG::NL]) {
  *reinterpret_cast<f32x4 *>(gt + s16 * PG::SLA + 4 * lg) = c[0];
  *reinterpret_cast<f32x4 *>(gt + (s16 + 16) * PG::SLA + 4 * lg) = c[1];
  wave_lds_fence();
#pragma unroll
  for (int q = 0; q < (PG::NL + 3) / 4; ++q) {
    const f32x4 r = *reinterpret_cast<const f32x4 *>(gt + l31 * PG::SLA + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * q + e < PG::NL) v[4 * q + e] = r[e];
  }
  wave_lds_fence();  // the tile is rewritten (abar | q) further down
}

// DIAG: the diagonal-Gaussian target has its own instantiation -- with the five targets behind one switch the register
// allocator sizes the kernel for the 2-d targets' atan2 / exp chains and spills the GEMM accumulators around the draws.
template <class PG, bool DIAG>
__global__ __launch_bounds__(SB, (PG::NLR <= 6 ? 2 : 1)) void k_planar_step(SimpleArgs a, const float *__restrict__ theta,
                                                                           const float *__restrict__ xs, SimpleFused fu,
                                                                           float lbar_const, float *__restrict__ slabs,
                                                                           long slab_stride) {
  constexpr int DB = PG::DB, NLR = PG::NLR, NL = PG::NL, FD = PG::FD, SW = PG::SW, SU = PG::SU;
  extern __shared__ __attribute__((aligned(16))) float psm[];
  const int d = a.d, nl = a.nl, LP = lp_of(d);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  float *sh = psm;
  float *wv = psm + PG::SHARED + wave * PG::WAVE;
  float *cache = psm + PG::SHARED + 4 * PG::WAVE;  // [nl][LP]: w | uhat | b | 1 + c  (build_layer_cache)
  build_layer_cache<float>(cache, a, theta);
  __syncthreads();
  // operand images (zero outside the flow's layers / features: they sit on contraction axes)
  for (int i = tid; i < NL * SW; i += SB) {
    const int e = i / SW, f = i - e * SW;
    const bool in = e < nl && f < d;
    sh[PG::OFF_W + i] = in ? cache[e * LP + f] : 0.f;
    sh[PG::OFF_UT + i] = in ? cache[e * LP + d + f] : 0.f;
  }
  for (int i = tid; i < FD * SU; i += SB) {
    const int f = i / SU, e = i - f * SU;
    sh[PG::OFF_U + i] = (e < nl && f < d) ? cache[e * LP + d + f] : 0.f;
  }
  for (int i = tid; i < NL * NL; i += SB) {
    const int l = i / NL, m = i - l * NL;
    float c = 0.f;
    if (l < nl && m < nl)
      for (int f = 0; f < d; ++f) c += cache[l * LP + f] * cache[m * LP + d + f];
    sh[PG::OFF_C + i] = c;
  }
  for (int i = tid; i < NL; i += SB) {
    sh[PG::OFF_B + i] = i < nl ? cache[i * LP + 2 * d] : 0.f;
    sh[PG::OFF_SP + i] = i < nl ? cache[i * LP + 2 * d + 1] : 1.f;
  }
  if (DIAG)
    for (int i = tid; i < FD; i += SB) {
      const float vv = i < d ? ((const float *)fu.var)[i] : 1.f;
      sh[PG::OFF_TG + i] = i < d ? ((const float *)fu.mu)[i] : 0.f;
      sh[PG::OFF_TG + FD + i] = i < d ? 1.f / vv : 0.f;
      sh[PG::OFF_TG + 2 * FD + i] = i < d ? 1.8378770664093453f + logf(vv) : 0.f;
    }
  __syncthreads();
  // wave-uniform scalars of the recurrences, read once: SGPRs (readfirstlane marks them uniform; both sweeps use the
  // strict upper triangle of C only).  (Round 5: with the 32 hoisted "feature < d" lane masks they overflow the scalar file and
  // hipcc parks the excess in VGPR lanes -- 217 v_readlane_b32 per tile.  Reading C from LDS in the sweeps and keeping the masks
  // from being hoisted removed 190 of them and changed nothing: 219.9 against 218.5 us.)
  auto sc = [&](int off) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sh[off]))); };
  float Cu[NL][NL], bs[NL], sps[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    bs[l] = sc(PG::OFF_B + l);
    sps[l] = sc(PG::OFF_SP + l);
#pragma unroll
    for (int m = l + 1; m < NL; ++m) Cu[l][m] = sc(PG::OFF_C + l * NL + m);
  }

  f32x4 M1[PG::RB], M2[PG::RB], G[PG::GRB][PG::GCB];  // lane (l = lane & 15, g): features 16 R + 4 g + r of layer l; G' rows 16 rb + 4 g + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int b = 0; b < PG::RB; ++b) M1[b][r] = M2[b][r] = 0.f;
#pragma unroll
    for (int rb = 0; rb < PG::GRB; ++rb)
#pragma unroll
      for (int cb = 0; cb < PG::GCB; ++cb) G[rb][cb][r] = 0.f;
  }
  double contrib = 0.0;
  // layer rows this lane fetches as A / B operands (clamped rows feed output columns nobody reads)
  const int s16 = lane & 15, lg = lane >> 4;
  const int erow = s16 < NL ? s16 : NL - 1;
  if (hi == 0) wv[PG::OFF_T + l31 * PG::SLT + NL] = 1.f;  // the ones column (the tiles are not touched until the epilogue)
  const long ntiles = (a.N + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    asm volatile("" ::: "memory");  // keep the operand fetches inside the loop (hoisting them costs the registers)
    const long j = tile * 32 + l31;
    const bool valid = j < a.N;
    // ---- base draws (or the caller's): features f = 32 blk + 8 q + 4 hi + e <-> C register 4 q + e
    f32x16 z[DB];
    float ss = 0.f;
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * b + 8 * q + 4 * hi;
        float n4[4] = {0.f, 0.f, 0.f, 0.f};
        if (f0 < d && valid) {
          if (fu.draw) {
            philox_normals4<float>(fu.off + (uint64_t)j, (uint32_t)(f0 >> 2), fu.stream, fu.k0, fu.k1, n4);
          } else {
            const float *row = xs + j * d + f0;
            if (a.vec) {
              const float4 v = *reinterpret_cast<const float4 *>(row);
              n4[0] = v.x; n4[1] = v.y; n4[2] = v.z; n4[3] = v.w;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) n4[e] = f0 + e < d ? row[e] : 0.f;
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = (f0 + e < d) ? n4[e] : 0.f;
          z[b][4 * q + e] = v;
          ss += v * v;
        }
      }
    // ---- z0 into the [sample][feature] tile (a lane's Philox groups are four consecutive features: b128 writes).  It is the
    //      B operand of W Z0 and, read back here before ybar replaces it, the A operand of Z0 Abar':
    //      zt[R][t] = z0[feature 16 R + (lane & 15)][sample planar_s(t) + 4 g]
    float zt[PG::RB][8];
    float *const xw = wv + PG::OFF_X + l31 * PG::SX + 4 * hi;
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4 *>(xw + 32 * b + 8 * q) = f32x4{z[b][4 * q], z[b][4 * q + 1], z[b][4 * q + 2], z[b][4 * q + 3]};
    wave_lds_fence();
#pragma unroll
    for (int R = 0; R < PG::RB; ++R)
#pragma unroll
      for (int t = 0; t < 8; ++t) zt[R][t] = wv[PG::OFF_X + (planar_s(t) + 4 * lg) * PG::SX + 16 * R + s16];
    // ---- A0[e][s] = w_e'z0_s
    f32x4 c0[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) c0[0][r] = c0[1][r] = 0.f;
    planar_rows_gemm16<FD, SW, PG::SX>(sh + PG::OFF_W + erow * SW + (FD / 4) * lg, wv + PG::OFF_X + s16 * PG::SX + (FD / 4) * lg, c0);
    // ---- forward recurrence: every lane all layers of its sample
    float A[NL], tl[NL];
    planar_rows_to_lanes<PG>(wv + PG::OFF_A, c0, s16, lg, l31, A);
    float lsum = 0.f;
#pragma unroll
    for (int l = NL - 1; l >= 0; --l) {
      tl[l] = 0.f;
      if (l < nl) {
        float av = A[l] + bs[l];
#pragma unroll
        for (int m = l + 1; m < NL; ++m) av += Cu[l][m] * tl[m];
        const float t = Fm<float>::tanh_(av);
        tl[l] = t;
        lsum += Fm<float>::log_(sps[l] * (1.f - t * t) + t * t);
      }
    }
    // ---- y = z0 + Uhat t   (B operand: this lane's C rows of t)
    {
      float tb[NLR];
#pragma unroll
      for (int r = 0; r < NLR; ++r) {
        const float t0 = nf_row(r, 0) < NL ? tl[nf_row(r, 0)] : 0.f, t1 = nf_row(r, 1) < NL ? tl[nf_row(r, 1)] : 0.f;
        tb[r] = hi ? t1 : t0;
      }
#pragma unroll
      for (int b = 0; b < DB; ++b) {
        const float *ul = sh + PG::OFF_U + (32 * b + l31) * SU + 4 * hi;
#pragma unroll
        for (int g = 0; g < (NLR + 3) / 4; ++g) {
          float u4[4];
          nf_ld4<SU>(ul + 8 * g, u4[0], u4[1], u4[2], u4[3]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (4 * g + e < NLR) z[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(u4[e], tb[4 * g + e], z[b], 0, 0, 0);
        }
      }
    }
    // ---- target: log p(y), ybar = gscale grad log p(y)   (z <- ybar)
    float acc = 0.f;
    {
      float y0, y1, dummy;
      planar_gather(z[0][0], y0, dummy);
      planar_gather(z[0][1], y1, dummy);
      float s2 = 0.f;
      if (!DIAG && fu.tkind == NF_TARGET_FUNNEL) {
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) s2 += (32 * b + nf_row(r, hi) >= 1) ? z[b][r] * z[b][r] : 0.f;
        s2 = planar_xhalf_sum(s2);
      }
      auto run = [&](auto kc) {
        constexpr int KD = decltype(kc)::value;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f = 32 * b + nf_row(r, hi);
            float gk = 0.f;
            if constexpr (KD == NF_TARGET_DIAGGAUSS) {
              const float rr = z[b][r] - sh[PG::OFF_TG + f], iv = sh[PG::OFF_TG + FD + f];
              acc -= 0.5f * (sh[PG::OFF_TG + 2 * FD + f] + rr * rr * iv);
              gk = -rr * iv;
            } else {
              if (f < d)
                acc += target_term<KD, float>(d, f, z[b][r], y0, y1, s2, (const float *)fu.mu, (const float *)fu.var, (float)fu.s0,
                                              (float)fu.s1, gk);
            }
            z[b][r] = valid ? (float)fu.gscale * gk : 0.f;
          }
      };
      if constexpr (DIAG) {
        run(std::integral_constant<int, NF_TARGET_DIAGGAUSS>{});
      } else {
        switch (fu.tkind) {
          case NF_TARGET_BANANA: run(std::integral_constant<int, NF_TARGET_BANANA>{}); break;
          case NF_TARGET_FUNNEL: run(std::integral_constant<int, NF_TARGET_FUNNEL>{}); break;
          case NF_TARGET_WARPED: run(std::integral_constant<int, NF_TARGET_WARPED>{}); break;
          default: run(std::integral_constant<int, NF_TARGET_CROSS>{}); break;
        }
      }
      acc = planar_xhalf_sum(acc);
      ss = planar_xhalf_sum(ss);
      const float logq = (float)(-0.5 * 1.8378770664093453 * d) - 0.5f * ss;
      if (valid && hi == 0) contrib += fu.pscale * (double)(acc - logq + lsum);
    }
    // ---- ybar into the tile (the A operand of Ybar T'), UG0[e][s] = uhat_e'ybar_s
    wave_lds_fence();  // every z0 read of the tile is complete (same wave: program order through the LDS queue)
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4 *>(xw + 32 * b + 8 * q) = f32x4{z[b][4 * q], z[b][4 * q + 1], z[b][4 * q + 2], z[b][4 * q + 3]};
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 4; ++r) c0[0][r] = c0[1][r] = 0.f;
    planar_rows_gemm16<FD, SW, PG::SX>(sh + PG::OFF_UT + erow * SW + (FD / 4) * lg, wv + PG::OFF_X + s16 * PG::SX + (FD / 4) * lg, c0);
    // ---- reverse recurrence
    float ab[NL], qv[NL];
    {
      float UG[NL];
      planar_rows_to_lanes<PG>(wv + PG::OFF_A, c0, s16, lg, l31, UG);
      const float lb = valid ? lbar_const : 0.f;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        ab[l] = qv[l] = 0.f;
        if (l < nl) {
          float ug = UG[l];
#pragma unroll
          for (int m = 0; m < l; ++m) ug += Cu[m][l] * ab[m];
          const float t = tl[l], sp = sps[l], cc = sp - 1.f;
          const float gg = 1.f - t * t, D = sp * gg + t * t, iD = Fm<float>::div_(1.f, D);
          const float av = ug * gg - 2.f * lb * cc * t * gg * iD;
          ab[l] = av;
          qv[l] = lb * gg * iD;
        }
      }
    }
    // ---- abar | q and t as [sample][layer] tiles; the three K = samples GEMMs: G' = [Abar; Q] [T; 1]', M1 = Z0 Abar', M2 = Ybar T'
    {
      float *ar = wv + PG::OFF_A + l31 * PG::SLA + (hi ? NL : 0), *tr = wv + PG::OFF_T + l31 * PG::SLT;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        ar[l] = hi ? qv[l] : ab[l];
        if (hi == 0) tr[l] = tl[l];
      }
    }
    wave_lds_fence();
    {
      // lane (i = lane & 15, g): A operands = row i of a block, B operands = column i; k-step t <-> sample planar_s(t) + 4 g.
      // Entries past a tile's row (rows >= 2 NL of G', columns >= NL of M1 / >= NL + 1 of G') read the neighbouring sample's:
      // finite values that end up in result rows / columns nobody reads.
      const float *pa = wv + PG::OFF_A + 4 * lg * PG::SLA + s16, *pt = wv + PG::OFF_T + 4 * lg * PG::SLT + s16;
      const float *px = wv + PG::OFF_X + 4 * lg * PG::SX + s16;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float av[PG::GRB], tv[PG::GCB], xv[PG::RB];
#pragma unroll
        for (int rb = 0; rb < PG::GRB; ++rb) av[rb] = pa[planar_s(t) * PG::SLA + 16 * rb];
#pragma unroll
        for (int cb = 0; cb < PG::GCB; ++cb) tv[cb] = pt[planar_s(t) * PG::SLT + 16 * cb];
#pragma unroll
        for (int R = 0; R < PG::RB; ++R) xv[R] = px[planar_s(t) * PG::SX + 16 * R];
#pragma unroll
        for (int rb = 0; rb < PG::GRB; ++rb)
#pragma unroll
          for (int cb = 0; cb < PG::GCB; ++cb) G[rb][cb] = planar_mfma16(av[rb], tv[cb], G[rb][cb]);
#pragma unroll
        for (int R = 0; R < PG::RB; ++R) {
          M1[R] = planar_mfma16(zt[R][t], av[0], M1[R]);
          M2[R] = planar_mfma16(xv[R], tv[0], M2[R]);
        }
      }
    }
    wave_lds_fence();  // the next tile overwrites X / A / T
  }
  // ---- block epilogue: waves in a fixed order, triangular corrections, slabs in k_simple_step's layout
  __syncthreads();
  {
    float *rg = psm + PG::SHARED + wave * PG::WAVE;
    if (s16 < NL) {
#pragma unroll
      for (int R = 0; R < PG::RB; ++R)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          rg[PG::R_M1 + s16 * FD + 16 * R + 4 * lg + r] = M1[R][r];
          rg[PG::R_M2 + s16 * FD + 16 * R + 4 * lg + r] = M2[R][r];
        }
    }
#pragma unroll
    for (int rb = 0; rb < PG::GRB; ++rb)
#pragma unroll
      for (int cb = 0; cb < PG::GCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * rb + 4 * lg + r, col = 16 * cb + s16;
          if (row < PG::AR && col < PG::TR) rg[PG::R_G + row * PG::TR + col] = G[rb][cb][r];
        }
  }
  __syncthreads();
  {
    float *r0 = psm + PG::SHARED;
    for (int i = tid; i < PG::REGION; i += SB) {
      float v = r0[i];
#pragma unroll
      for (int w = 1; w < 4; ++w) v += r0[w * PG::WAVE + i];
      r0[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < nl * d; i += SB) {
      const int l = i / d, f = i - l * d;
      float wb = r0[PG::R_M1 + l * FD + f], ub = r0[PG::R_M2 + l * FD + f];
      for (int m = l + 1; m < nl; ++m) wb += cache[m * LP + d + f] * r0[PG::R_G + l * PG::TR + m];
      for (int m = 0; m < l; ++m) ub += cache[m * LP + f] * r0[PG::R_G + m * PG::TR + l];
      float *out = slabs + (long)l * slab_stride + (long)blockIdx.x * LP;
      out[f] = wb;
      out[d + f] = ub;
    }
    for (int l = tid; l < nl; l += SB) {
      float *out = slabs + (long)l * slab_stride + (long)blockIdx.x * LP;
      out[2 * d] = r0[PG::R_G + l * PG::TR + NL];             // bbar_l = sum_j abar_l^j
      out[2 * d + 1] = r0[PG::R_G + (NL + l) * PG::TR + NL];  // sum_j q_l^j
    }
  }
  {  // deterministic block sum of the ELBO terms
    __shared__ double sm[SB / 64];
    double c = contrib;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) sm[wave] = c;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < SB / 64; ++w) t += sm[w];
      fu.partial[blockIdx.x] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Radial training step, one lane per (sample, feature half) (k_radial_step): d <= 64, up to 16 layers, Float32.
// ------------------------------------------------------------------------------------------------------------------
// Same layout as k_planar_step -- a lane holds sample l31 of a 32-sample tile and the C-layout features of its half-wave
// (whole Philox groups) -- but no GEMMs: a radial layer's update direction z - z0 changes with the sample, so the planar
// reformulation does not carry over.  What the layout buys here is the scalar math once per sample
// instead of sixteen times (sqrt, the reciprocals, the log terms, the reverse pass's chain of scalars), norms and dot
// products as 32 in-lane FMAs + one half-wave exchange instead of four DPP steps per four features, and ONE stashed scalar
// per layer.  The only cross-sample sums, z0bar_l[f] = -sum_j db_j[f], are 32 values per lane to be summed over the 32
// lanes of a half-wave: wave_transpose_reduce32 leaves the sum of value i in lane i (v_permlane16_swap across the two
// rows, then DPP row_ror:8, row_half_mirror, quad_perm xor 2, xor 1 with the lane's own bits selecting what it keeps:
// 77 instructions; tools/probe/transpose_reduce_probe.hip), so the accumulators cost one register per layer.
__device__ __forceinline__ float radial_dpp(float v, int stage) {
  const int b = __builtin_bit_cast(int, v);
  switch (stage) {
    case 0: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, b, 0x128, 0xF, 0xF, false));  // row_ror:8
    case 1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, b, 0x141, 0xF, 0xF, false));  // row_half_mirror
    case 2: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
  }
}
// v[16 b + r]: 32 values per lane; returns, in lane l31 of either half-wave, the sum over that half's 32 lanes of v[l31]
__device__ __forceinline__ float wave_transpose_reduce32(const float (&v)[32], int lane) {
  float w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned a = __builtin_bit_cast(unsigned, v[i]), b = __builtin_bit_cast(unsigned, v[i + 16]);
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);  // [a.row0 b.row0 a.row2 b.row2], [a.row1 b.row1 ...]
    unsigned r0 = r[0], r1 = r[1];
    asm volatile("s_nop 1" : "+v"(r0), "+v"(r1));  // as v_permlane32_swap: a VALU read of the results needs wait states
    w[i] = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
  }
  float u[8], t[4], q[2];
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = (b3 ? w[i + 8] : w[i]) + radial_dpp(b3 ? w[i] : w[i + 8], 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = (b2 ? u[i + 4] : u[i]) + radial_dpp(b2 ? u[i] : u[i + 4], 1);
#pragma unroll
  for (int i = 0; i < 2; ++i) q[i] = (b1 ? t[i + 2] : t[i]) + radial_dpp(b1 ? t[i] : t[i + 2], 2);
  return (b0 ? q[1] : q[0]) + radial_dpp(b0 ? q[0] : q[1], 3);
}

template <int DB_, int NL_>
struct RadialGeo {
  static constexpr int DB = DB_, NL = NL_, FD = 32 * DB;
  static constexpr int OFF_C = 0;                   // centres z0_l[f], [NL][FD], zero beyond d / nl
  static constexpr int OFF_A = OFF_C + NL * FD;     // alpha_l
  static constexpr int OFF_BH = OFF_A + NL;         // beta_hat_l
  static constexpr int OFF_TG = ((OFF_BH + NL + 3) / 4) * 4;  // target: mu[f] | 1/var[f] | log 2pi + log var[f]
  static constexpr int SHARED = OFF_TG + 3 * FD;
  static constexpr int ROW = NL * FD + 2 * NL;      // per wave, block epilogue: z0bar[l][f] | alpha_bar[l] | betahat_bar[l]
  static constexpr int RS = NL * 64;                // per wave: the stashed r_l of every layer, [NL][lane] (round 6: 20 registers
                                                    // of the d = 64 instantiations went to scratch for them)
  static constexpr size_t lds_floats(int nl, int lp) { return (size_t)SHARED + 4 * (size_t)ROW + (size_t)nl * lp + 4 * (size_t)RS; }
};

template <class RG, bool DIAG>
__global__ __launch_bounds__(SB, 2) void k_radial_step(SimpleArgs a, const float *__restrict__ theta, const float *__restrict__ xs,
                                                       SimpleFused fu, float lbar_const, float *__restrict__ slabs,
                                                       long slab_stride) {
  constexpr int DB = RG::DB, NL = RG::NL, FD = RG::FD;
  extern __shared__ __attribute__((aligned(16))) float rsm[];
  const int d = a.d, nl = a.nl, LP = lp_of(d);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  float *sh = rsm;
  float *cache = rsm + RG::SHARED + 4 * RG::ROW;  // [nl][LP]: z0 | alpha | beta_hat  (build_layer_cache)
  build_layer_cache<float>(cache, a, theta);
  __syncthreads();
  for (int i = tid; i < NL * FD; i += SB) {
    const int l = i / FD, f = i - l * FD;
    sh[RG::OFF_C + i] = (l < nl && f < d) ? cache[l * LP + f] : 0.f;
  }
  for (int i = tid; i < NL; i += SB) {
    sh[RG::OFF_A + i] = i < nl ? cache[i * LP + d] : 1.f;
    sh[RG::OFF_BH + i] = i < nl ? cache[i * LP + d + 1] : 0.f;
  }
  if (DIAG)
    for (int i = tid; i < FD; i += SB) {
      const float vv = i < d ? ((const float *)fu.var)[i] : 1.f;
      sh[RG::OFF_TG + i] = i < d ? ((const float *)fu.mu)[i] : 0.f;
      sh[RG::OFF_TG + FD + i] = i < d ? 1.f / vv : 0.f;
      sh[RG::OFF_TG + 2 * FD + i] = i < d ? 1.8378770664093453f + logf(vv) : 0.f;
    }
  __syncthreads();
  auto sc = [&](int off) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sh[off]))); };
  // (alpha_l, beta_hat_l are fetched where they are used, inside the tile loop: held in 2 NL registers across it they were what
  // the d = 64 instantiations spilled -- 84 bytes of scratch at ten layers, round 5)
  // acc: z0bar of the feature wave_transpose_reduce32 leaves in this lane; sab: both half-waves carry every sample, so the
  // lower half sums the samples' alpha_bar terms and the upper half their betahat_bar terms (one register per layer)
  float acc[NL], sab[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) acc[l] = sab[l] = 0.f;
  double contrib = 0.0;
  const float dm1 = (float)(d - 1);
  const long ntiles = (a.N + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    asm volatile("" ::: "memory");  // keep the centre fetches inside the loop
    const long j = tile * 32 + l31;
    const bool valid = j < a.N;
    f32x16 z[DB];
    float ss = 0.f;
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * b + 8 * q + 4 * hi;
        float n4[4] = {0.f, 0.f, 0.f, 0.f};
        if (f0 < d && valid) {
          if (fu.draw) {
            philox_normals4<float>(fu.off + (uint64_t)j, (uint32_t)(f0 >> 2), fu.stream, fu.k0, fu.k1, n4);
          } else {
            const float *row = xs + j * d + f0;
            if (a.vec) {
              const float4 v = *reinterpret_cast<const float4 *>(row);
              n4[0] = v.x; n4[1] = v.y; n4[2] = v.z; n4[3] = v.w;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) n4[e] = f0 + e < d ? row[e] : 0.f;
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = (f0 + e < d) ? n4[e] : 0.f;
          z[b][4 * q + e] = v;
          ss += v * v;
        }
      }
    // ---- forward: r_l is all that is kept of a layer
    float *rs = cache + nl * LP + wave * RG::RS + lane;  // rs[l * 64]: this lane's r_l, written and read by the same lane
    float lsum = 0.f;
#pragma unroll
    for (int l = NL - 1; l >= 0; --l) {
      if (l < nl) {
        const float *cl = sh + RG::OFF_C + l * FD + 4 * hi;
        f32x16 dl[DB];
        float s2 = 0.f;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float c4[4];
            nf_ld4<FD>(cl + 32 * b + 8 * q, c4[0], c4[1], c4[2], c4[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float dv = z[b][4 * q + e] - c4[e];  // features beyond d: 0 - 0
              dl[b][4 * q + e] = dv;
              s2 += dv * dv;
            }
          }
        const float r = Fm<float>::sqrt_(planar_xhalf_sum(s2));
        const float h = Fm<float>::div_(1.f, sc(RG::OFF_A + l) + r), qv = sc(RG::OFF_BH + l) * h;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) z[b][e] += qv * dl[b][e];
        lsum += dm1 * Fm<float>::log1p_(qv) + Fm<float>::log1p_(qv - qv * h * r);
        rs[l * 64] = r;
      }
    }
    // ---- target: log p(y), g = gscale grad log p(y)
    f32x16 g[DB];
    float tacc = 0.f;
    {
      float y0, y1, dummy;
      planar_gather(z[0][0], y0, dummy);
      planar_gather(z[0][1], y1, dummy);
      float s2 = 0.f;
      if (!DIAG && fu.tkind == NF_TARGET_FUNNEL) {
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) s2 += (32 * b + nf_row(r, hi) >= 1) ? z[b][r] * z[b][r] : 0.f;
        s2 = planar_xhalf_sum(s2);
      }
      auto run = [&](auto kc) {
        constexpr int KD = decltype(kc)::value;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f = 32 * b + nf_row(r, hi);
            float gk = 0.f;
            if constexpr (KD == NF_TARGET_DIAGGAUSS) {
              const float rr = z[b][r] - sh[RG::OFF_TG + f], iv = sh[RG::OFF_TG + FD + f];
              tacc -= 0.5f * (sh[RG::OFF_TG + 2 * FD + f] + rr * rr * iv);
              gk = -rr * iv;
            } else {
              if (f < d)
                tacc += target_term<KD, float>(d, f, z[b][r], y0, y1, s2, (const float *)fu.mu, (const float *)fu.var, (float)fu.s0,
                                               (float)fu.s1, gk);
            }
            g[b][r] = valid ? (float)fu.gscale * gk : 0.f;
          }
      };
      if constexpr (DIAG) {
        run(std::integral_constant<int, NF_TARGET_DIAGGAUSS>{});
      } else {
        switch (fu.tkind) {
          case NF_TARGET_BANANA: run(std::integral_constant<int, NF_TARGET_BANANA>{}); break;
          case NF_TARGET_FUNNEL: run(std::integral_constant<int, NF_TARGET_FUNNEL>{}); break;
          case NF_TARGET_WARPED: run(std::integral_constant<int, NF_TARGET_WARPED>{}); break;
          default: run(std::integral_constant<int, NF_TARGET_CROSS>{}); break;
        }
      }
      tacc = planar_xhalf_sum(tacc);
      ss = planar_xhalf_sum(ss);
      const float logq = (float)(-0.5 * 1.8378770664093453 * d) - 0.5f * ss;
      if (valid && hi == 0) contrib += fu.pscale * (double)(tacc - logq + lsum);
    }
    // ---- reverse: the state walks back alongside the cotangent (layer_bwd_s's algebra, once per sample)
    const float lb = valid ? lbar_const : 0.f;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (l < nl) {
        const float alpha = sc(RG::OFF_A + l), bh = sc(RG::OFF_BH + l), r = rs[l * 64];
        const float h = Fm<float>::div_(1.f, alpha + r), qq = bh * h, bah2 = bh * alpha * h * h;
        const float iq = Fm<float>::div_(1.f, 1.f + qq), ib = Fm<float>::div_(1.f, 1.f + bah2), ir = r > 0.f ? Fm<float>::div_(1.f, r) : 0.f;
        const float *cl = sh + RG::OFF_C + l * FD + 4 * hi;
        float dl[32];
        float yd = 0.f;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float c4[4];
            nf_ld4<FD>(cl + 32 * b + 8 * q, c4[0], c4[1], c4[2], c4[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float dv = (z[b][4 * q + e] - c4[e]) * iq;  // y - z0 = delta (1 + beta_hat h)
              z[b][4 * q + e] = c4[e] + dv;                    // the layer input
              dl[16 * b + 4 * q + e] = dv;
              yd += g[b][4 * q + e] * dv;
            }
          }
#pragma unroll
        for (int i = 16 * DB; i < 32; ++i) dl[i] = 0.f;
        yd = planar_xhalf_sum(yd);
        const float dL_dh = dm1 * bh * iq + 2.f * bh * alpha * h * ib;
        const float dL_db = dm1 * h * iq + alpha * h * h * ib;
        const float dL_da = bh * h * h * ib;
        const float hbar = bh * yd + lb * dL_dh;
        const float rho = -h * h * hbar * ir;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float dbv = qq * g[b][e] + rho * dl[16 * b + e];
            dl[16 * b + e] = dbv;
            g[b][e] += dbv;
          }
        acc[l] -= wave_transpose_reduce32(dl, lane);  // z0bar
        sab[l] += hi ? (h * yd + lb * dL_db) : (-h * h * hbar + lb * dL_da);
      }
    }
  }
  // ---- block epilogue: waves in a fixed order, slabs in k_simple_step's layout (z0bar | - | alpha_bar | betahat_bar)
  {
    float *rg = rsm + RG::SHARED + wave * RG::ROW;
    const int facc = 32 * (l31 >> 4) + nf_row(l31 & 15, hi);  // the feature whose sum wave_transpose_reduce32 left in this lane
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (facc < FD) rg[l * FD + facc] = acc[l];
      float v0 = sab[l];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v0 += __shfl_xor(v0, o, 64);  // within the half-wave
      if (l31 == 0) rg[NL * FD + hi * NL + l] = v0;
    }
  }
  __syncthreads();
  {
    float *r0 = rsm + RG::SHARED;
    for (int i = tid; i < RG::ROW; i += SB) {
      float v = r0[i];
#pragma unroll
      for (int w = 1; w < 4; ++w) v += r0[w * RG::ROW + i];
      r0[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < nl * d; i += SB) {
      const int l = i / d, f = i - l * d;
      float *out = slabs + (long)l * slab_stride + (long)blockIdx.x * LP;
      out[f] = r0[l * FD + f];
      out[d + f] = 0.f;
    }
    for (int l = tid; l < nl; l += SB) {
      float *out = slabs + (long)l * slab_stride + (long)blockIdx.x * LP;
      out[2 * d] = r0[NL * FD + l];
      out[2 * d + 1] = r0[NL * FD + NL + l];
    }
  }
  {  // deterministic block sum of the ELBO terms
    __shared__ double sm[SB / 64];
    double c = contrib;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) sm[wave] = c;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < SB / 64; ++w) t += sm[w];
      fu.partial[blockIdx.x] = t;
    }
  }
}

// sums the per-block slabs of every layer and applies the parameter-space chain rule
// (get_u_hat for planar, softplus re-parameterisation for radial).  One block of FB threads per layer: FB / 64
// row groups each sum every (FB/64)-th slab with 64 consecutive columns per wave (coalesced, independent loads),
// then the row groups are added in a fixed order (deterministic).
#define FB 512
template <class T>
__global__ __launch_bounds__(FB) void k_simple_finalize(SimpleArgs a, const T *__restrict__ theta,
                                                        const T *__restrict__ slabs, int nblk_bwd,
                                                        T *__restrict__ gtheta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int d = a.d, LP = lp_of(d);
  T *sum = (T *)smem;  // LP
  T *red = sum + LP;   // [FB / 64][LP]
  constexpr int RG = FB / 64;
  const int l = blockIdx.x;
  const T *sl = slabs + (long)l * nblk_bwd * LP;
  const int rg = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int s = lane; s < 2 * d + 2; s += 64) {
    T v = 0;
#pragma unroll 4
    for (int b = rg; b < nblk_bwd; b += RG) v += sl[(long)b * LP + s];
    red[rg * LP + s] = v;
  }
  __syncthreads();
  for (int s = threadIdx.x; s < 2 * d + 2; s += FB) {
    T v = 0;
#pragma unroll
    for (int r = 0; r < RG; ++r) v += red[r * LP + s];
    sum[s] = v;
  }
  __syncthreads();
  const T *p = theta + layer_off(a.kind, d, l);
  T *g = gtheta + layer_off(a.kind, d, l);
  const int lk = layer_kind(a.kind, l);
  if (lk == LK_PLANAR) {
    __shared__ double sc[3];
    if (threadIdx.x == 0) {
      double m = 0, ww = 0, uw = 0;
      for (int i = 0; i < d; ++i) {
        m += (double)p[i] * (double)p[d + i];
        ww += (double)p[i] * (double)p[i];
        uw += (double)sum[d + i] * (double)p[i];
      }
      sc[0] = m; sc[1] = ww; sc[2] = uw;
    }
    __syncthreads();
    const T m = (T)sc[0], ww = (T)sc[1], uw = (T)sc[2];
    const T sg = sigmoid_(m), spn = softplus_(-m) - (T)1;
    const T mbar = sum[2 * d + 1] * sg + uw * (sg - (T)1) / ww;
    for (int i = threadIdx.x; i < d; i += FB) {
      const T ub = sum[d + i];
      g[i] = sum[i] + mbar * p[d + i] + spn * (ub / ww - (T)2 * uw * p[i] / (ww * ww));
      g[d + i] = ub + mbar * p[i];
    }
    if (threadIdx.x == 0) g[2 * d] = sum[2 * d];
  } else if (lk == LK_RADIAL) {
    for (int i = threadIdx.x; i < d; i += FB) g[2 + i] = sum[i];
    if (threadIdx.x == 0) {
      g[0] = (sum[2 * d] - sum[2 * d + 1]) * sigmoid_(p[0]);
      g[1] = sum[2 * d + 1] * sigmoid_(p[1]);
    }
  } else if (lk == LK_SHIFT) {
    for (int i = threadIdx.x; i < d; i += FB) g[i] = sum[i];
  } else {
    for (int i = threadIdx.x; i < d; i += FB) g[i] = sum[i] + sum[2 * d] / p[i];
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static inline int dpl_for(int d) {
  const int need = (d + LPS - 1) / LPS;
  int dpl = 1;
  while (dpl < need) dpl *= 2;
  return dpl;
}

// vector row access: the row length must be a whole number of vectors and every base pointer aligned to one
template <class T>
static int vec_ok(int d, int dpl, std::initializer_list<const void *> ptrs) {
  const int vw = (dpl * (int)sizeof(T) >= 16) ? 16 / (int)sizeof(T) : dpl;
  if (vw <= 1 || d % vw != 0) return 0;
  for (const void *p : ptrs)
    if (p && ((uintptr_t)p % (vw * sizeof(T))) != 0) return 0;
  return 1;
}

bool nf_simple_supported(const nf_flow_desc *desc) {
  if (desc->d > 256) return false;
  const long LP = lp_of(desc->d);
  const long es = desc->dtype == NF_DTYPE_F64 ? 8 : 4;
  const int nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  return (long)nl * LP * es <= 64 * 1024;
}

static SimpleArgs make_sargs(const nf_flow_desc *desc, int lo, int hi, bool inverse, long N) {
  SimpleArgs a;
  a.kind = desc->kind;
  a.d = desc->d;
  a.nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  a.lo = lo;
  a.hi = hi;
  a.inverse = inverse ? 1 : 0;
  a.vec = 0;
  a.N = N;
  return a;
}

// grid of a grid-stride kernel: what the device holds at once (no second, partly filled round of workgroups)
template <class K>
static inline long resident_blocks(nf_ctx *ctx, K kernel, size_t lds) {
  // the occupancy query is a driver call: remember the answer per (kernel, dynamic LDS) -- the small flows
  // of the demos are launch-bound
  struct Entry { const void *k; size_t lds; int device; int per_cu; };
  static std::vector<Entry> seen;
  static std::mutex seen_mu;  // contexts on different devices may be driven from different threads
  std::lock_guard<std::mutex> guard(seen_mu);
  for (const Entry &e : seen)
    if (e.k == (const void *)kernel && e.lds == lds && e.device == ctx->device) return (long)e.per_cu * ctx->num_cu;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, SB, lds) != hipSuccess || per_cu < 1) per_cu = 2;
  seen.push_back({(const void *)kernel, lds, ctx->device, per_cu});
  return (long)per_cu * ctx->num_cu;
}
static inline unsigned grid_for(long N, long cap) {
  long g = (N + SPB - 1) / SPB;
  if (g > cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}

template <class T>
static int apply_t(nf_ctx *ctx, const SimpleArgs &a0, const void *theta, const void *x, void *y, void *ladj, void *stash,
                   const SimpleFused *fused = nullptr, long *nblocks_out = nullptr) {
  SimpleArgs a = a0;
  SimpleFused fu{};
  if (fused) fu = *fused;
  a.vec = vec_ok<T>(a.d, dpl_for(a.d), {x, y, stash, fu.gbar});
  const size_t lds = (size_t)(a.hi - a.lo) * lp_of(a.d) * sizeof(T);
  ProfScope ps(ctx, "simple_apply");
#define LAUNCH_APPLY(DPLv)                                                                                      \
  do {                                                                                                          \
    const unsigned gridv = grid_for(a.N, resident_blocks(ctx, k_simple_apply<T, DPLv>, lds));                   \
    if (nblocks_out) *nblocks_out = gridv;                                                                      \
    hipLaunchKernelGGL((k_simple_apply<T, DPLv>), dim3(gridv), dim3(SB), lds, ctx->stream, a, (const T *)theta, \
                       (const T *)x, (T *)y, (T *)ladj, (T *)stash, fu);                                        \
  } while (0)
  switch (dpl_for(a.d)) {
    case 1: LAUNCH_APPLY(1); break;
    case 2: LAUNCH_APPLY(2); break;
    case 4: LAUNCH_APPLY(4); break;
    case 8: LAUNCH_APPLY(8); break;
    case 16: LAUNCH_APPLY(16); break;
    default: return NF_ERR_UNSUPPORTED;
  }
#undef LAUNCH_APPLY
  return (int)hipGetLastError();
}

int nf_simple_apply(nf_ctx *ctx, const nf_flow_desc *desc, int lo, int hi, bool inverse, const void *theta,
                    const void *x, long N, void *y, void *ladj) {
  if (N <= 0) return NF_OK;
  SimpleArgs a = make_sargs(desc, lo, hi, inverse, N);
  if (desc->dtype == NF_DTYPE_F32) return apply_t<float>(ctx, a, theta, x, y, ladj, nullptr);
  return apply_t<double>(ctx, a, theta, x, y, ladj, nullptr);
}

static inline int bwd_blocks(nf_ctx *ctx, long N) {
  long g = (N + SPB - 1) / SPB;
  const long cap = 4L * ctx->num_cu;  // 16 waves per CU in flight: the kernel is HBM-bound
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}

size_t nf_simple_bwd_ws_bytes(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  const size_t es = desc->dtype == NF_DTYPE_F64 ? 8 : 4;
  const int nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  const size_t LP = lp_of(desc->d);
  return carve_bytes((size_t)nl * N * desc->d * es) + carve_bytes((size_t)N * desc->d * es) +
         carve_bytes((size_t)nl * bwd_blocks(ctx, N) * LP * es);
}

// forward pass that leaves the input of every layer in the reverse pass's workspace (same carving as
// bwd_t), so that nf_simple_bwd(..., have_stash = true) need not recompute it
int nf_simple_apply_stash(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x, long N, void *y,
                          void *ladj, void *ws, bool inverse) {
  if (N <= 0) return NF_OK;
  const int nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  SimpleArgs a = make_sargs(desc, 0, nl, inverse, N);
  Carver cv(ws);
  if (desc->dtype == NF_DTYPE_F32) return apply_t<float>(ctx, a, theta, x, y, ladj, cv.take<float>((size_t)nl * N * desc->d));
  return apply_t<double>(ctx, a, theta, x, y, ladj, cv.take<double>((size_t)nl * N * desc->d));
}

// The training step's forward pass in ONE launch (see SimpleFused): xs == nullptr draws in-library.  Leaves the
// per-layer inputs in `ws` (carving of bwd_t), ybar in gbar, block partials of pscale * elbo_j in partial
// (*npartial of them; room for nf_simple_elbo_max_partials()).
long nf_simple_elbo_max_partials(nf_ctx *ctx) { return 16L * ctx->num_cu; }
int nf_simple_elbo_forward(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta, const void *xs,
                           long N, uint64_t seed, uint64_t off, uint32_t stream_id, void *gbar, double gscale, double *partial,
                           double pscale, void *ws, long *npartial) {
  if (N <= 0) return NF_OK;
  const int nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  SimpleArgs a = make_sargs(desc, 0, nl, false, N);
  SimpleFused fu{};
  fu.on = 1;
  fu.draw = xs ? 0 : 1;
  fu.tkind = target->kind;
  fu.k0 = (uint32_t)seed; fu.k1 = (uint32_t)(seed >> 32); fu.stream = stream_id; fu.off = off;
  fu.mu = target->p0; fu.var = target->p1; fu.s0 = target->s0; fu.s1 = target->s1;
  fu.gscale = gscale; fu.pscale = pscale; fu.gbar = gbar; fu.partial = partial;
  // ws == nullptr (ELBO value only): no per-layer inputs are kept
  Carver cv(ws);
  if (desc->dtype == NF_DTYPE_F32)
    return apply_t<float>(ctx, a, theta, xs, nullptr, nullptr, ws ? cv.take<float>((size_t)nl * N * desc->d) : nullptr, &fu,
                          npartial);
  return apply_t<double>(ctx, a, theta, xs, nullptr, nullptr, ws ? cv.take<double>((size_t)nl * N * desc->d) : nullptr, &fu,
                         npartial);
}

// rand(rng, flow, n): the base draws are generated in the chain kernel's registers and pushed through the
// transform -- one launch, only the samples are written
int nf_simple_rand(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, long N, uint64_t seed, uint64_t off,
                   uint32_t stream_id, void *y) {
  if (N <= 0) return NF_OK;
  const int nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  SimpleArgs a = make_sargs(desc, 0, nl, false, N);
  SimpleFused fu{};
  fu.draw = 1;
  fu.k0 = (uint32_t)seed; fu.k1 = (uint32_t)(seed >> 32); fu.stream = stream_id; fu.off = off;
  if (desc->dtype == NF_DTYPE_F32) return apply_t<float>(ctx, a, theta, nullptr, y, nullptr, nullptr, &fu);
  return apply_t<double>(ctx, a, theta, nullptr, y, nullptr, nullptr, &fu);
}

template <class T>
static int bwd_t(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x, const void *ybar,
                 const void *lbar, double lbar_const, long N, void *xbar_out, void *gtheta_out, void *ws,
                 bool have_stash, bool inv) {
  if (inv && !have_stash) return NF_ERR_ARG;  // the inverse chain's points come from nf_simple_apply_stash(inverse)
  const int nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  const int d = desc->d;
  const size_t LP = lp_of(d);
  const int nb = bwd_blocks(ctx, N);
  Carver cv(ws);
  T *stash = cv.take<T>((size_t)nl * N * d);
  T *ytmp = cv.take<T>((size_t)N * d);
  T *slabs = cv.take<T>((size_t)nl * nb * LP);
  SimpleArgs a = make_sargs(desc, 0, nl, false, N);
  // forward recompute, stashing the input of every layer (execution index e <-> flat layer nl-1-e)
  if (!have_stash) NF_TRY(apply_t<T>(ctx, a, theta, x, ytmp, nullptr, stash));
  if (xbar_out != ybar)
    NF_HIP(hipMemcpyAsync(xbar_out, ybar, (size_t)N * d * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
  a.vec = vec_ok<T>(d, dpl_for(d), {stash, xbar_out});
  int nb_used = nb;
  {
    ProfScope ps(ctx, "simple_bwd");
    // LDS: the pass's layer caches + one reduction row per wave; grid: what is resident at once, at most `nb`
#define LAUNCH_BWD_V(DPLv, INVv)                                                                                       \
  do {                                                                                                                 \
    const size_t ldsv = (size_t)(1 + SB / 64) * (BwdCfg<T, DPLv>::LPP) * LP * sizeof(T);                                 \
    const long res = resident_blocks(ctx, k_simple_bwd_layers<T, DPLv, INVv>, ldsv);                                    \
    nb_used = (int)(res < nb ? res : nb);                                                                              \
    hipLaunchKernelGGL((k_simple_bwd_layers<T, DPLv, INVv>), dim3(nb_used), dim3(SB), ldsv, ctx->stream, a, nl,          \
                       (const T *)theta, (const T *)stash, (long)N * d, (T *)xbar_out, (const T *)lbar, (T)lbar_const, \
                       slabs, (long)nb_used * LP);                                                                     \
  } while (0)
#define LAUNCH_BWD(DPLv)           \
  do {                             \
    if (inv) LAUNCH_BWD_V(DPLv, true); \
    else LAUNCH_BWD_V(DPLv, false);    \
  } while (0)
    switch (dpl_for(d)) {
      case 1: LAUNCH_BWD(1); break;
      case 2: LAUNCH_BWD(2); break;
      case 4: LAUNCH_BWD(4); break;
      case 8: LAUNCH_BWD(8); break;
      case 16: LAUNCH_BWD(16); break;
      default: return NF_ERR_UNSUPPORTED;
    }
#undef LAUNCH_BWD
#undef LAUNCH_BWD_V
    NF_HIP(hipGetLastError());
  }
  ProfScope pf(ctx, "simple_finalize");
  hipLaunchKernelGGL(k_simple_finalize<T>, dim3(nl), dim3(FB), (size_t)(1 + FB / 64) * LP * sizeof(T), ctx->stream, a, (const T *)theta,
                     (const T *)slabs, nb_used, (T *)gtheta_out);
  return (int)hipGetLastError();
}

int nf_simple_bwd(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x, const void *ybar,
                  const void *lbar, double lbar_const, long N, void *xbar_out, void *gtheta_out, void *ws,
                  bool have_stash, bool inv) {
  if (desc->dtype == NF_DTYPE_F32)
    return bwd_t<float>(ctx, desc, theta, x, ybar, lbar, lbar_const, N, xbar_out, gtheta_out, ws, have_stash, inv);
  return bwd_t<double>(ctx, desc, theta, x, ybar, lbar, lbar_const, N, xbar_out, gtheta_out, ws, have_stash, inv);
}

// ---- the stash-free training step (k_simple_step) ----------------------------------------------------------------
template <class T, int DPL, int KIND, int NLMAX>
static int step_launch(nf_ctx *ctx, SimpleArgs a, const void *theta, const void *xs, const SimpleFused &fu, double lbar_const,
                       T *slabs, int *nb_out) {
  const size_t LP = lp_of(a.d);
  const size_t lds = (size_t)(1 + SB / 64) * a.nl * LP * sizeof(T);
  static AttrOnce attr_once;  // once per device
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_simple_step<T, DPL, KIND, NLMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    return NF_OK;
  }));
  long nb = (a.N + SPB - 1) / SPB;
  const long res = resident_blocks(ctx, k_simple_step<T, DPL, KIND, NLMAX>, lds);
  if (nb > res) nb = res;
  if (nb < 1) nb = 1;
  *nb_out = (int)nb;
  ProfScope ps(ctx, "simple_step");
  hipLaunchKernelGGL((k_simple_step<T, DPL, KIND, NLMAX>), dim3((unsigned)nb), dim3(SB), lds, ctx->stream, a, (const T *)theta,
                     (const T *)xs, fu, (T)lbar_const, slabs, (long)nb * (long)LP);
  return (int)hipGetLastError();
}

// ---- the planar step on the matrix pipe (k_planar_step) ------------------------------------------------------------
#ifndef NF_PLANAR_MFMA_MIN_D
#define NF_PLANAR_MFMA_MIN_D 2
#endif
static bool planar_mfma_ok(const nf_flow_desc *desc) {
  static const bool off = std::getenv("NF_PLANAR_NO_MFMA") != nullptr;  // A/B switch: k_simple_step
  return !off && desc->kind == NF_KIND_PLANAR && desc->dtype == NF_DTYPE_F32 && desc->d >= NF_PLANAR_MFMA_MIN_D && desc->d <= 64 &&
         desc->nlayers >= 1 && desc->nlayers <= 16;
}
template <class PG, bool DIAG>
static int planar_launch_t(nf_ctx *ctx, SimpleArgs a, const void *theta, const void *xs, const SimpleFused &fu, double lbar_const,
                         float *slabs, int *nb_out) {
  const size_t LP = lp_of(a.d);
  const size_t lds = PG::lds_floats(a.nl, (int)LP) * sizeof(float);
  static AttrOnce attr_once;  // once per device
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_planar_step<PG, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    return NF_OK;
  }));
  a.vec = (xs && a.d % 4 == 0 && (uintptr_t)xs % 16 == 0) ? 1 : 0;
  long nb = ((a.N + 31) / 32 + 3) / 4;
  const long res = resident_blocks(ctx, k_planar_step<PG, DIAG>, lds);
  if (nb > res) nb = res;
  if (nb < 1) nb = 1;
  *nb_out = (int)nb;
  ProfScope ps(ctx, "planar_step");
  hipLaunchKernelGGL((k_planar_step<PG, DIAG>), dim3((unsigned)nb), dim3(SB), lds, ctx->stream, a, (const float *)theta, (const float *)xs, fu,
                     (float)lbar_const, slabs, (long)nb * (long)LP);
  return (int)hipGetLastError();
}
template <class PG>
static int planar_launch(nf_ctx *ctx, const SimpleArgs &a, const void *theta, const void *xs, const SimpleFused &fu, double lbar_const,
                         float *slabs, int *nb_out) {
  return fu.tkind == NF_TARGET_DIAGGAUSS ? planar_launch_t<PG, true>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out)
                                         : planar_launch_t<PG, false>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
}
static int planar_step(nf_ctx *ctx, const SimpleArgs &a, const void *theta, const void *xs, const SimpleFused &fu, double lbar_const,
                       float *slabs, int *nb_out) {
  const bool wide = a.d > 32;
  if (a.nl <= 10)
    return wide ? planar_launch<PlanarGeo<2, 6>>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out)
                : planar_launch<PlanarGeo<1, 6>>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
  return wide ? planar_launch<PlanarGeo<2, 8>>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out)
              : planar_launch<PlanarGeo<1, 8>>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
}

// ---- the radial step, one lane per (sample, feature half) (k_radial_step) -------------------------------------------
static bool radial_lane_ok(const nf_flow_desc *desc) {
  static const bool off = std::getenv("NF_RADIAL_NO_LANE") != nullptr;  // A/B switch: k_simple_step
  return !off && desc->kind == NF_KIND_RADIAL && desc->dtype == NF_DTYPE_F32 && desc->d >= 8 && desc->d <= 64 &&
         desc->nlayers >= 1 && desc->nlayers <= 16;
}
template <class RG, bool DIAG>
static int radial_launch_t(nf_ctx *ctx, SimpleArgs a, const void *theta, const void *xs, const SimpleFused &fu, double lbar_const,
                           float *slabs, int *nb_out) {
  const size_t LP = lp_of(a.d);
  const size_t lds = RG::lds_floats(a.nl, (int)LP) * sizeof(float);
  static AttrOnce attr_once;  // once per device
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_radial_step<RG, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    return NF_OK;
  }));
  a.vec = (xs && a.d % 4 == 0 && (uintptr_t)xs % 16 == 0) ? 1 : 0;
  long nb = ((a.N + 31) / 32 + 3) / 4;
  const long res = resident_blocks(ctx, k_radial_step<RG, DIAG>, lds);
  if (nb > res) nb = res;
  if (nb < 1) nb = 1;
  *nb_out = (int)nb;
  ProfScope ps(ctx, "radial_step");
  hipLaunchKernelGGL((k_radial_step<RG, DIAG>), dim3((unsigned)nb), dim3(SB), lds, ctx->stream, a, (const float *)theta, (const float *)xs,
                     fu, (float)lbar_const, slabs, (long)nb * (long)LP);
  return (int)hipGetLastError();
}
template <class RG>
static int radial_launch(nf_ctx *ctx, const SimpleArgs &a, const void *theta, const void *xs, const SimpleFused &fu, double lbar_const,
                         float *slabs, int *nb_out) {
  return fu.tkind == NF_TARGET_DIAGGAUSS ? radial_launch_t<RG, true>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out)
                                         : radial_launch_t<RG, false>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
}
static int radial_step(nf_ctx *ctx, const SimpleArgs &a, const void *theta, const void *xs, const SimpleFused &fu, double lbar_const,
                       float *slabs, int *nb_out) {
  const bool wide = a.d > 32;
  if (a.nl <= 10)
    return wide ? radial_launch<RadialGeo<2, 10>>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out)
                : radial_launch<RadialGeo<1, 10>>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
  return wide ? radial_launch<RadialGeo<2, 16>>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out)
              : radial_launch<RadialGeo<1, 16>>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
}

// smallest unroll bound that holds nl layers
// three unroll bounds (4, 10, 12): unused slots still cost their accumulator registers, and ten layers is the shape of
// BASELINE cfg 1 and of the reference's planar / radial tests (test/flow.jl:137,203)
static inline int step_bound(int nl) { return nl <= 4 ? 4 : nl <= 10 ? 10 : 12; }

template <class T, int DPL, int KIND>
static int step_dpl(nf_ctx *ctx, const SimpleArgs &a, const void *theta, const void *xs, const SimpleFused &fu,
                    double lbar_const, T *slabs, int *nb_out) {
  constexpr int BUDGET = (sizeof(T) == 4 ? 64 : 32) / DPL;  // layers whose inputs fit the stash registers
  const int nb = step_bound(a.nl);
  if constexpr (KIND == NF_KIND_MEANFIELD) return step_launch<T, DPL, KIND, 2>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
  if constexpr (BUDGET >= 4)
    if (nb == 4) return step_launch<T, DPL, KIND, 4>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
  if constexpr (BUDGET >= 10)
    if (nb == 10) return step_launch<T, DPL, KIND, 10>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
  if constexpr (BUDGET >= 12)
    if (nb == 12) return step_launch<T, DPL, KIND, 12>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
  return NF_ERR_UNSUPPORTED;
}

template <class T, int KIND>
static int step_kind(nf_ctx *ctx, const SimpleArgs &a, const void *theta, const void *xs, const SimpleFused &fu,
                     double lbar_const, T *slabs, int *nb_out) {
  switch (dpl_for(a.d)) {
    case 1: return step_dpl<T, 1, KIND>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
    case 2: return step_dpl<T, 2, KIND>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
    case 4: return step_dpl<T, 4, KIND>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
    case 8: return step_dpl<T, 8, KIND>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
    case 16: return step_dpl<T, 16, KIND>(ctx, a, theta, xs, fu, lbar_const, slabs, nb_out);
    default: return NF_ERR_UNSUPPORTED;
  }
}

static int step_nlmax(const nf_flow_desc *desc) {
  if (desc->kind == NF_KIND_MEANFIELD) return 2;
  const int budget = (desc->dtype == NF_DTYPE_F64 ? 32 : 64) / dpl_for(desc->d);
  return budget >= 12 ? 12 : budget >= 10 ? 10 : budget >= 4 ? 4 : 0;
}

// flows whose every layer input fits the register budget of k_simple_step and whose caches + reduction rows fit LDS
bool nf_simple_step_supported(const nf_flow_desc *desc) {
  if (planar_mfma_ok(desc) || radial_lane_ok(desc)) return true;
  if (!nf_simple_supported(desc) || dpl_for(desc->d) > 16) return false;
  const int nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  const size_t es = desc->dtype == NF_DTYPE_F64 ? 8 : 4;
  if (nl > step_nlmax(desc)) return false;
  return (size_t)(1 + SB / 64) * nl * lp_of(desc->d) * es <= 144 * 1024;
}

size_t nf_simple_step_ws_bytes(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  const size_t es = desc->dtype == NF_DTYPE_F64 ? 8 : 4;
  const int nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  long nb = (N + SPB - 1) / SPB;
  const long cap = 16L * ctx->num_cu;  // upper bound of the resident-block count
  if (nb > cap) nb = cap;
  return carve_bytes((size_t)nl * nb * lp_of(desc->d) * es);
}

// loss partials (pscale * elbo_j, *npartial block sums in `partial`) and the gradient of the step in two launches:
// k_simple_step (draws or xs, chain, target, reverse pass, slabs) and k_simple_finalize.  gscale / lbar_const are the
// cotangents: -1 / N_global for loss = -elbo_batch.
template <class T>
static int step_t(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta, const void *xs, long N,
                  uint64_t seed, uint64_t off, uint32_t stream_id, double gscale, double lbar_const, double *partial,
                  double pscale, void *ws, void *gtheta_out, long *npartial) {
  const int nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  SimpleArgs a = make_sargs(desc, 0, nl, false, N);
  a.vec = vec_ok<T>(a.d, dpl_for(a.d), {xs});
  SimpleFused fu{};
  fu.on = 1;
  fu.draw = xs ? 0 : 1;
  fu.tkind = target->kind;
  fu.k0 = (uint32_t)seed; fu.k1 = (uint32_t)(seed >> 32); fu.stream = stream_id; fu.off = off;
  fu.mu = target->p0; fu.var = target->p1; fu.s0 = target->s0; fu.s1 = target->s1;
  fu.gscale = gscale; fu.pscale = pscale; fu.gbar = nullptr; fu.partial = partial;
  T *slabs = (T *)ws;
  int nb = 0;
  int st;
  if constexpr (std::is_same<T, float>::value) {
    if (planar_mfma_ok(desc) || radial_lane_ok(desc)) {
      NF_TRY(planar_mfma_ok(desc) ? planar_step(ctx, a, theta, xs, fu, lbar_const, slabs, &nb)
                                  : radial_step(ctx, a, theta, xs, fu, lbar_const, slabs, &nb));
      *npartial = nb;
      ProfScope pf(ctx, "simple_finalize");
      hipLaunchKernelGGL(k_simple_finalize<T>, dim3(nl), dim3(FB), (size_t)(1 + FB / 64) * lp_of(a.d) * sizeof(T), ctx->stream, a,
                         (const T *)theta, (const T *)slabs, nb, (T *)gtheta_out);
      return (int)hipGetLastError();
    }
  }
  if (desc->kind == NF_KIND_PLANAR) st = step_kind<T, NF_KIND_PLANAR>(ctx, a, theta, xs, fu, lbar_const, slabs, &nb);
  else if (desc->kind == NF_KIND_RADIAL) st = step_kind<T, NF_KIND_RADIAL>(ctx, a, theta, xs, fu, lbar_const, slabs, &nb);
  else st = step_kind<T, NF_KIND_MEANFIELD>(ctx, a, theta, xs, fu, lbar_const, slabs, &nb);
  NF_TRY(st);
  *npartial = nb;
  const size_t LP = lp_of(a.d);
  ProfScope pf(ctx, "simple_finalize");
  hipLaunchKernelGGL(k_simple_finalize<T>, dim3(nl), dim3(FB), (size_t)(1 + FB / 64) * LP * sizeof(T), ctx->stream, a,
                     (const T *)theta, (const T *)slabs, nb, (T *)gtheta_out);
  return (int)hipGetLastError();
}

int nf_simple_elbo_step(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta, const void *xs,
                        long N, uint64_t seed, uint64_t off, uint32_t stream_id, double gscale, double lbar_const,
                        double *partial, double pscale, void *ws, void *gtheta_out, long *npartial) {
  if (N <= 0) return NF_OK;
  if (desc->dtype == NF_DTYPE_F32)
    return step_t<float>(ctx, desc, target, theta, xs, N, seed, off, stream_id, gscale, lbar_const, partial, pscale, ws,
                         gtheta_out, npartial);
  return step_t<double>(ctx, desc, target, theta, xs, N, seed, off, stream_id, gscale, lbar_const, partial, pscale, ws,
                        gtheta_out, npartial);
}
