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
#include "nf_common.h"

#define LPS 16
#define SB 256
#define SPB (SB / LPS)

enum { LK_PLANAR = 0, LK_RADIAL = 1, LK_SHIFT = 2, LK_SCALE = 3 };

template <class T>
__device__ __forceinline__ T g16sum(T v) {
  v += __shfl_xor(v, 8, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 1, 16);
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

struct SimpleArgs {
  int kind;     // NF_KIND_*
  int d;
  int nl;       // number of flat layers of the flow
  int lo, hi;   // flat layer range to apply
  int inverse;
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

// per-layer cache in LDS, stride LP = 2d + 2:
//   planar: w[d] | uhat[d] | b | 1 + c      (get_u_hat: test/ext/CUDA/cuda.jl:12-18; c = w'uhat = softplus(w'u) - 1.
//           1 + c = softplus(w'u) is kept instead of c: the Jacobian factor 1 + c sech^2 = (1+c) sech^2 + tanh^2 is
//           then a sum of non-negative terms -- no cancellation as c -> -1, where 1 + c (1 - t^2) loses every digit)
//   radial: z0[d] | alpha | beta_hat
//   shift : a[d]
//   scale : a[d] | sum(log|a|)
template <class T>
__device__ void build_layer_cache(T *cache, const SimpleArgs &a, const T *__restrict__ theta) {
  const int d = a.d, LP = 2 * d + 2;
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
__device__ __forceinline__ T layer_forward(int lk, const T *c, int d, int q, T (&z)[DPL]) {
  if (lk == LK_PLANAR) {
    T dot = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      if (i < d) dot += c[i] * z[k];
    }
    const T t = tanh(g16sum(dot) + c[2 * d]);
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      if (i < d) z[k] += c[d + i] * t;
    }
    return log(c[2 * d + 1] * ((T)1 - t * t) + t * t);
  }
  if (lk == LK_RADIAL) {
    const T alpha = c[d], bh = c[d + 1];
    T ss = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      if (i < d) {
        const T dl = z[k] - c[i];
        ss += dl * dl;
      }
    }
    const T r = sqrt(g16sum(ss));
    const T h = (T)1 / (alpha + r);
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      if (i < d) z[k] += bh * h * (z[k] - c[i]);
    }
    return (T)(d - 1) * log1p(bh * h) + log1p(bh * h - bh * h * h * r);
  }
  if (lk == LK_SHIFT) {
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      if (i < d) z[k] += c[i];
    }
    return (T)0;
  }
#pragma unroll
  for (int k = 0; k < DPL; ++k) {
    const int i = q + LPS * k;
    if (i < d) z[k] *= c[i];
  }
  return c[d];
}

template <class T, int DPL>
__device__ __forceinline__ T layer_inverse(int lk, const T *c, int d, int q, T (&z)[DPL]) {
  if (lk == LK_PLANAR) {
    // solve alpha + c tanh(alpha + b) = w'y for alpha = w'z  (monotone: c > -1)
    T dot = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      if (i < d) dot += c[i] * z[k];
    }
    const T wy = g16sum(dot), b = c[2 * d], sp = c[2 * d + 1], cc = sp - (T)1;
    T lo = wy - fabs(cc), hi = wy + fabs(cc);
    const int iters = sizeof(T) == 8 ? 64 : 40;
    for (int it = 0; it < iters; ++it) {
      const T mid = (T)0.5 * (lo + hi);
      const T f = mid + cc * tanh(mid + b) - wy;
      if (f > (T)0) hi = mid; else lo = mid;
    }
    T al = (T)0.5 * (lo + hi);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {  // Newton polish
      const T t = tanh(al + b);
      al -= (al + cc * t - wy) / (sp * ((T)1 - t * t) + t * t);
    }
    const T t = tanh(al + b);
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      if (i < d) z[k] -= c[d + i] * t;
    }
    return -log(sp * ((T)1 - t * t) + t * t);
  }
  if (lk == LK_RADIAL) {
    const T alpha = c[d], bh = c[d + 1];
    T ss = 0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      if (i < d) {
        const T dl = z[k] - c[i];
        ss += dl * dl;
      }
    }
    const T rho = sqrt(g16sum(ss));
    const T aa = (alpha + bh) - rho;
    const T r = (T)0.5 * (sqrt(aa * aa + (T)4 * alpha * rho) - aa);
    const T f = (alpha + r) / (alpha + bh + r);
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      if (i < d) z[k] = c[i] + f * (z[k] - c[i]);
    }
    const T h = (T)1 / (alpha + r);
    return -((T)(d - 1) * log1p(bh * h) + log1p(bh * h - bh * h * h * r));
  }
  if (lk == LK_SHIFT) {
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      if (i < d) z[k] -= c[i];
    }
    return (T)0;
  }
#pragma unroll
  for (int k = 0; k < DPL; ++k) {
    const int i = q + LPS * k;
    if (i < d) z[k] /= c[i];
  }
  return -c[d];
}

// chain-fused forward / inverse over flat layers [lo, hi).  If `stash` != nullptr the reverse pass's
// evaluation points are left behind: forward -- the INPUT of every layer at stash[e][N][d] (execution
// index e); inverse -- the OUTPUT of every inverse layer at stash[l][N][d] (flat index l), i.e. the
// forward-sense input of layer l, which is where the inverse chain's reverse pass evaluates it.
template <class T, int DPL>
__global__ __launch_bounds__(SB) void k_simple_apply(SimpleArgs a, const T *__restrict__ theta, const T *x,
                                                     T *y, T *__restrict__ ladj, T *__restrict__ stash) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *cache = (T *)smem;
  const int d = a.d, LP = 2 * d + 2, nlr = a.hi - a.lo;
  build_layer_cache<T>(cache, a, theta);
  __syncthreads();
  const int q = threadIdx.x & (LPS - 1);
  for (long j = (long)blockIdx.x * SPB + threadIdx.x / LPS; j < a.N; j += (long)gridDim.x * SPB) {
    const bool valid = true;  // a 16-lane group shares j, so it is converged for the shuffles
    const long jj = valid ? j : 0;
    T z[DPL];
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      z[k] = (valid && i < d) ? x[jj * d + i] : (T)0;
    }
    T lsum = 0;
    for (int e = 0; e < nlr; ++e) {
      const int l = a.inverse ? a.lo + e : a.hi - 1 - e;
      if (stash && valid && !a.inverse) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
          const int i = q + LPS * k;
          if (i < d) stash[((long)e * a.N + j) * d + i] = z[k];
        }
      }
      const T *c = cache + (long)(l - a.lo) * LP;
      const int lk = layer_kind(a.kind, l);
      lsum += a.inverse ? layer_inverse<T, DPL>(lk, c, d, q, z) : layer_forward<T, DPL>(lk, c, d, q, z);
      if (stash && valid && a.inverse) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
          const int i = q + LPS * k;
          if (i < d) stash[((long)l * a.N + j) * d + i] = z[k];
        }
      }
    }
    if (valid) {
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        const int i = q + LPS * k;
        if (i < d) y[j * d + i] = z[k];
      }
      if (q == 0 && ladj) ladj[j] = lsum;
    }
  }
}

// Reverse pass of ONE layer over the whole batch: zin = layer input (from the stash),
// gbar: in = dL/d(output), out = dL/d(input).  Raw parameter sums go to slab[block][2d+2]:
//   planar: wbar_raw[d] | uhat_bar[d] | bbar | cbar
//   radial: z0bar[d] | alpha_bar | betahat_bar
//   shift : abar[d]            scale: sum(ybar .* x)[d] | sum(lbar)
//
// INV: reverse pass of the INVERSE chain (forward-KL training).  Layers are walked in forward execution
// order; zin = the inverse layer's OUTPUT w (stash of the inverse pass), gbar in = cotangent of w, out =
// cotangent vbar of the inverse layer's input, lbar = cotangent of ladj_inv.  Implicit-function form:
//   vbar = J^-T (wbar - lbar grad_w ladj_fwd),  parameter sums = the forward formulas with (-vbar, -lbar);
// J^-T is closed-form for every layer here (Sherman-Morrison for planar / radial).
template <class T, int DPL, bool INV>
__global__ __launch_bounds__(SB) void k_simple_bwd_layers(SimpleArgs a, int nl, const T *__restrict__ theta,
                                                          const T *__restrict__ stash, long stash_stride,
                                                          T *__restrict__ gbar, const T *__restrict__ lbar, T lbar_const,
                                                          T *__restrict__ slabs, long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *cache = (T *)smem;                 // one layer: 2d+2
  const int d = a.d, LP = 2 * d + 2;
  T *red = cache + LP;                  // [SPB][LP] block reduction buffer
  // every layer in one launch: a thread keeps its samples from layer to layer (it re-reads the gbar it
  // wrote), so only the block-wide parameter reduction needs the barriers
#pragma unroll 1
  for (int step = 0; step < nl; ++step) {
    const int l = INV ? nl - 1 - step : step;  // forward chain: flat order = reverse of execution order
    const T *zin = stash + (long)(INV ? l : nl - 1 - l) * stash_stride;
    T *slab = slabs + (long)l * slab_stride;
    SimpleArgs one = a;
    one.lo = l;
    one.hi = l + 1;
    build_layer_cache<T>(cache, one, theta);
    __syncthreads();
    const int q = threadIdx.x & (LPS - 1), grp = threadIdx.x / LPS;
    const int lk = layer_kind(a.kind, l);
    const T *c = cache;
    T acc0[DPL], acc1[DPL];
    T s0 = 0, s1 = 0;
  #pragma unroll
    for (int k = 0; k < DPL; ++k) acc0[k] = acc1[k] = 0;

    for (long j = (long)blockIdx.x * SPB + grp; j < a.N; j += (long)gridDim.x * SPB) {
      const bool valid = true;
      const long jj = valid ? j : 0;
      T z[DPL], g[DPL], v[DPL];
  #pragma unroll
      for (int k = 0; k < DPL; ++k) {
        const int i = q + LPS * k;
        const bool ok = valid && i < d;
        z[k] = ok ? zin[jj * d + i] : (T)0;
        g[k] = ok ? gbar[jj * d + i] : (T)0;
        v[k] = (T)0;
      }
      T lb = valid ? (lbar ? lbar[jj] : lbar_const) : (T)0;
      if (lk == LK_PLANAR) {
        T dot = 0, ug = 0;
  #pragma unroll
        for (int k = 0; k < DPL; ++k) {
          const int i = q + LPS * k;
          if (i < d) {
            dot += c[i] * z[k];
            ug += c[d + i] * g[k];
          }
        }
        const T sp = c[2 * d + 1], cc = sp - (T)1;
        const T t = tanh(g16sum(dot) + c[2 * d]);
        ug = g16sum(ug);
        const T gg = (T)1 - t * t, D = sp * gg + t * t;
        if (INV) {
          // J^T = I + gg w uhat^T, grad_z ladj = kap w, uhat^T w = cc:  vbar = g - w beta
          const T kap = -(T)2 * cc * t * gg / D;
          const T uap = ug - lb * kap * cc;
          const T beta = lb * kap + gg * uap / D;
  #pragma unroll
          for (int k = 0; k < DPL; ++k) {
            const int i = q + LPS * k;
            if (i < d) {
              v[k] = g[k] - c[i] * beta;
              g[k] = -v[k];
            }
          }
          ug = -(ug - cc * beta);
          lb = -lb;
        }
        const T ab = ug * gg - (T)2 * lb * cc * t * gg / D;
  #pragma unroll
        for (int k = 0; k < DPL; ++k) {
          const int i = q + LPS * k;
          if (i < d) {
            acc0[k] += ab * z[k];  // wbar_raw
            acc1[k] += t * g[k];   // uhat_bar
            g[k] += c[i] * ab;     // zbar
          }
        }
        if (q == 0) {
          s0 += ab;            // bbar
          s1 += lb * gg / D;   // cbar
        }
      } else if (lk == LK_RADIAL) {
        const T alpha = c[d], bh = c[d + 1];
        T ss = 0, yd = 0;
  #pragma unroll
        for (int k = 0; k < DPL; ++k) {
          const int i = q + LPS * k;
          if (i < d) {
            z[k] -= c[i];  // delta
            ss += z[k] * z[k];
            yd += g[k] * z[k];
          }
        }
        const T r = sqrt(g16sum(ss));
        yd = g16sum(yd);
        const T h = (T)1 / (alpha + r);
        const T qq = bh * h, bah2 = bh * alpha * h * h;
        const T dL_dh = (T)(d - 1) * bh / ((T)1 + qq) + (T)2 * bh * alpha * h / ((T)1 + bah2);
        const T dL_db = (T)(d - 1) * h / ((T)1 + qq) + alpha * h * h / ((T)1 + bah2);
        const T dL_da = bh * h * h / ((T)1 + bah2);
        if (INV) {
          // J = A I + Bc delta delta^T (symmetric), A + Bc r^2 = 1 + bah2, grad_z ladj = -dL_dh h^2 delta / r
          const T A = (T)1 + qq;
          const T e = r > (T)0 ? lb * dL_dh * h * h / r : (T)0;
          const T dap = yd + lb * dL_dh * h * h * r;
          const T f2 = r > (T)0 ? (-bh * h * h / r) * dap / ((T)1 + bah2) : (T)0;
  #pragma unroll
          for (int k = 0; k < DPL; ++k) {
            const int i = q + LPS * k;
            if (i < d) {
              v[k] = (g[k] + (e - f2) * z[k]) / A;
              g[k] = -v[k];
            }
          }
          yd = -dap / ((T)1 + bah2);
          lb = -lb;
        }
        const T hbar = bh * yd + lb * dL_dh;
        const T rbar_over_r = r > (T)0 ? -h * h * hbar / r : (T)0;
  #pragma unroll
        for (int k = 0; k < DPL; ++k) {
          const int i = q + LPS * k;
          if (i < d) {
            const T db = qq * g[k] + rbar_over_r * z[k];
            acc0[k] -= db;  // z0bar
            g[k] += db;     // zbar
          }
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
          const int i = q + LPS * k;
          if (i < d) {
            if (INV) {
              v[k] = g[k] / c[i];
              acc0[k] -= v[k] * z[k];
            } else {
              acc0[k] += g[k] * z[k];
              g[k] *= c[i];
            }
          }
        }
        if (q == 0) s0 += INV ? -lb : lb;
      }
      if (valid) {
  #pragma unroll
        for (int k = 0; k < DPL; ++k) {
          const int i = q + LPS * k;
          if (i < d) gbar[j * d + i] = INV ? v[k] : g[k];
        }
      }
    }
    // deterministic block reduction over the SPB sample groups
    T *mine = red + (long)grp * LP;
  #pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int i = q + LPS * k;
      if (i < d) {
        mine[i] = acc0[k];
        mine[d + i] = acc1[k];
      }
    }
    if (q == 0) {
      mine[2 * d] = s0;
      mine[2 * d + 1] = s1;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < LP; s += SB) {
      T v = 0;
      for (int gI = 0; gI < SPB; ++gI) v += red[(long)gI * LP + s];
      slab[(long)blockIdx.x * LP + s] = v;
    }
    __syncthreads();  // cache and reduction buffer are rebuilt for the next layer
  }
}

// sums the per-block slabs of every layer and applies the parameter-space chain rule
// (get_u_hat for planar, softplus re-parameterisation for radial).  One block per layer.
template <class T>
__global__ __launch_bounds__(SB) void k_simple_finalize(SimpleArgs a, const T *__restrict__ theta,
                                                        const T *__restrict__ slabs, int nblk_bwd,
                                                        T *__restrict__ gtheta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int d = a.d, LP = 2 * d + 2;
  T *sum = (T *)smem;  // LP
  const int l = blockIdx.x;
  const T *sl = slabs + (long)l * nblk_bwd * LP;
  for (int s = threadIdx.x; s < LP; s += SB) {
    T v = 0;
    for (int b = 0; b < nblk_bwd; ++b) v += sl[(long)b * LP + s];
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
    for (int i = threadIdx.x; i < d; i += SB) {
      const T ub = sum[d + i];
      g[i] = sum[i] + mbar * p[d + i] + spn * (ub / ww - (T)2 * uw * p[i] / (ww * ww));
      g[d + i] = ub + mbar * p[i];
    }
    if (threadIdx.x == 0) g[2 * d] = sum[2 * d];
  } else if (lk == LK_RADIAL) {
    for (int i = threadIdx.x; i < d; i += SB) g[2 + i] = sum[i];
    if (threadIdx.x == 0) {
      g[0] = (sum[2 * d] - sum[2 * d + 1]) * sigmoid_(p[0]);
      g[1] = sum[2 * d + 1] * sigmoid_(p[1]);
    }
  } else if (lk == LK_SHIFT) {
    for (int i = threadIdx.x; i < d; i += SB) g[i] = sum[i];
  } else {
    for (int i = threadIdx.x; i < d; i += SB) g[i] = sum[i] + sum[2 * d] / p[i];
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

bool nf_simple_supported(const nf_flow_desc *desc) {
  if (desc->d > 256) return false;
  const long LP = 2L * desc->d + 2;
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
  a.N = N;
  return a;
}

static inline unsigned grid_for(nf_ctx *ctx, long N) {
  long g = (N + SPB - 1) / SPB;
  const long cap = 8L * ctx->num_cu;
  if (g > cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}

template <class T>
static int apply_t(nf_ctx *ctx, const SimpleArgs &a, const void *theta, const void *x, void *y, void *ladj, void *stash) {
  const size_t lds = (size_t)(a.hi - a.lo) * (2 * a.d + 2) * sizeof(T);
  const unsigned grid = grid_for(ctx, a.N);
  ProfScope ps(ctx, "simple_apply");
#define LAUNCH_APPLY(DPLv)                                                                                   \
  hipLaunchKernelGGL((k_simple_apply<T, DPLv>), dim3(grid), dim3(SB), lds, ctx->stream, a, (const T *)theta, \
                     (const T *)x, (T *)y, (T *)ladj, (T *)stash)
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
  const long cap = 2L * ctx->num_cu;
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}

size_t nf_simple_bwd_ws_bytes(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  const size_t es = desc->dtype == NF_DTYPE_F64 ? 8 : 4;
  const int nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  const size_t LP = 2 * (size_t)desc->d + 2;
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

template <class T>
static int bwd_t(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x, const void *ybar,
                 const void *lbar, double lbar_const, long N, void *xbar_out, void *gtheta_out, void *ws,
                 bool have_stash, bool inv) {
  if (inv && !have_stash) return NF_ERR_ARG;  // the inverse chain's points come from nf_simple_apply_stash(inverse)
  const int nl = desc->kind == NF_KIND_MEANFIELD ? 2 : desc->nlayers;
  const int d = desc->d;
  const size_t LP = 2 * (size_t)d + 2;
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
  const size_t lds = (LP + (size_t)SPB * LP) * sizeof(T);
  {
    ProfScope ps(ctx, "simple_bwd");
#define LAUNCH_BWD_V(DPLv, INVv)                                                                                     \
  hipLaunchKernelGGL((k_simple_bwd_layers<T, DPLv, INVv>), dim3(nb), dim3(SB), lds, ctx->stream, a, nl, (const T *)theta, \
                     (const T *)stash, (long)N * d, (T *)xbar_out, (const T *)lbar, (T)lbar_const, slabs, (long)nb * LP)
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
  hipLaunchKernelGGL(k_simple_finalize<T>, dim3(nl), dim3(SB), LP * sizeof(T), ctx->stream, a, (const T *)theta,
                     (const T *)slabs, nb, (T *)gtheta_out);
  return (int)hipGetLastError();
}

int nf_simple_bwd(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x, const void *ybar,
                  const void *lbar, double lbar_const, long N, void *xbar_out, void *gtheta_out, void *ws,
                  bool have_stash, bool inv) {
  if (desc->dtype == NF_DTYPE_F32)
    return bwd_t<float>(ctx, desc, theta, x, ybar, lbar, lbar_const, N, xbar_out, gtheta_out, ws, have_stash, inv);
  return bwd_t<double>(ctx, desc, theta, x, ybar, lbar, lbar_const, N, xbar_out, gtheta_out, ws, have_stash, inv);
}
