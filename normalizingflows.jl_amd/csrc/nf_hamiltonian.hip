// nf_hamiltonian.hip -- the Hamiltonian flow of example/demo_hamiltonian_flow.jl:27-146 (SURVEY.md 8f,
// rank 4): a mean-field Gaussian reference on the joint z = [x; rho] followed by n blocks
//     (momentum Shift o Scale)  o  LeapFrog(L steps, per-dimension step sizes exp(log_eps), score of the target)
// LeapFrog is symplectic (log|det J| = 0, demo :86-93); its inverse is the same integrator with -eps
// (:74-84).  The reverse pass is hand-derived and needs Hessian-vector products of the target's
// log-density, which exist in closed form for the diagonal Gaussian, Banana and Funnel targets.
//
// theta = [shift0(2D), scale0(2D), then per block: shift_rho(D), scale_rho(D), log_eps(D)]
// (transformed(q0, T) destructures its dist -- q0's own Shift o Scale, demo :135-137 -- before its
// transform; a block is ComposedFunction(outer = momentum layer, inner = LeapFrog), :144).
// Flat layer index (nf_layer_apply): 0 .. n-1 = blocks, n = the reference's affine map.
//
// One thread per sample, standard layout, Float32 and Float64 (the demo recommends Float64: the
// dynamics is chaotic).  Small problems by nature (the demo: D = 2, 15 blocks, 16 samples per step).
#include "nf_common.h"

#define HF_MAXD 32  // position dimensions (joint dimension 2D <= 64)
#define HF_MAXL 16  // leapfrog steps per block
#define HF_BLOCK 64

struct HfArgs {
  int D, n, L, tkind;
  const void *p0, *p1;  // diagonal Gaussian: mu[D], var[D]
  double s0, s1;
  long N;
};

// score (gradient of log p) and Hessian-vector product of the built-in targets
// (example/targets/banana.jl:58-83, neal_funnel.jl:53-72; MvNormal(mu, Diagonal(var)))
template <class T>
__device__ void hf_score(const HfArgs &a, const T *x, T *g) {
  const int D = a.D;
  if (a.tkind == NF_TARGET_DIAGGAUSS) {
    const T *mu = (const T *)a.p0, *var = (const T *)a.p1;
    for (int i = 0; i < D; ++i) g[i] = -(x[i] - mu[i]) / var[i];
  } else if (a.tkind == NF_TARGET_BANANA) {
    const T b = (T)a.s0, s = (T)a.s1;
    const T y2 = x[1] + b * x[0] * x[0] - s * b;
    g[0] = -x[0] / s - (T)2 * b * x[0] * y2;
    g[1] = -y2;
    for (int i = 2; i < D; ++i) g[i] = -x[i];
  } else {  // Funnel
    const T mu = (T)a.s0, sg = (T)a.s1, e = exp(-x[0]);
    T s2 = 0;
    for (int i = 1; i < D; ++i) s2 += x[i] * x[i];
    g[0] = (mu - x[0]) / (sg * sg) - (T)(D - 1) / (T)2 + e * s2 / (T)2;
    for (int i = 1; i < D; ++i) g[i] = -e * x[i];
  }
}
template <class T>
__device__ void hf_hvp(const HfArgs &a, const T *x, const T *v, T *out) {
  const int D = a.D;
  if (a.tkind == NF_TARGET_DIAGGAUSS) {
    const T *var = (const T *)a.p1;
    for (int i = 0; i < D; ++i) out[i] = -v[i] / var[i];
  } else if (a.tkind == NF_TARGET_BANANA) {
    const T b = (T)a.s0, s = (T)a.s1;
    const T y2 = x[1] + b * x[0] * x[0] - s * b;
    const T h11 = -(T)1 / s - (T)2 * b * y2 - (T)4 * b * b * x[0] * x[0], h12 = -(T)2 * b * x[0];
    out[0] = h11 * v[0] + h12 * v[1];
    out[1] = h12 * v[0] - v[1];
    for (int i = 2; i < D; ++i) out[i] = -v[i];
  } else {
    const T sg = (T)a.s1, e = exp(-x[0]);
    T s2 = 0, xv = 0;
    for (int i = 1; i < D; ++i) { s2 += x[i] * x[i]; xv += x[i] * v[i]; }
    out[0] = (-(T)1 / (sg * sg) - e * s2 / (T)2) * v[0] + e * xv;
    for (int i = 1; i < D; ++i) out[i] = e * x[i] * v[0] - e * v[i];
  }
}

// _leapfrog of the demo (:49-60); sign = -1 runs it backwards in time.  If tr != nullptr the states
// (x_s, v_s, g(x_s)), s = 0..L, are recorded: tr[(s*3 + {0,1,2})*HF_MAXD + i].
template <class T>
__device__ void hf_leapfrog(const HfArgs &a, const T *leps, T sign, T *x, T *v, T *tr) {
  const int D = a.D, L = a.L;
  T eps[HF_MAXD], g[HF_MAXD];
  for (int i = 0; i < D; ++i) eps[i] = sign * exp(leps[i]);
  hf_score<T>(a, x, g);
  for (int i = 0; i < D; ++i) v[i] += eps[i] / (T)2 * g[i];
  auto rec = [&](int s) {
    if (tr)
      for (int i = 0; i < D; ++i) {
        tr[(s * 3 + 0) * HF_MAXD + i] = x[i];
        tr[(s * 3 + 1) * HF_MAXD + i] = v[i];
        tr[(s * 3 + 2) * HF_MAXD + i] = g[i];
      }
  };
  rec(0);
  for (int s = 1; s < L; ++s) {
    for (int i = 0; i < D; ++i) x[i] += eps[i] * v[i];
    hf_score<T>(a, x, g);
    for (int i = 0; i < D; ++i) v[i] += eps[i] * g[i];
    rec(s);
  }
  for (int i = 0; i < D; ++i) x[i] += eps[i] * v[i];
  hf_score<T>(a, x, g);
  for (int i = 0; i < D; ++i) v[i] += eps[i] / (T)2 * g[i];
  rec(L);
}

// flat layers [lo, hi) of the chain, forward (executed hi-1 .. lo) or inverse (lo .. hi-1)
template <class T>
__global__ __launch_bounds__(HF_BLOCK) void k_hf_apply(HfArgs a, int lo, int hi, int inverse, const T *__restrict__ theta,
                                                       const T *x_in, T *y_out, T *__restrict__ ladj) {
  const long j = (long)blockIdx.x * HF_BLOCK + threadIdx.x;
  if (j >= a.N) return;
  const int D = a.D, d2 = 2 * D;
  T z[2 * HF_MAXD];
  for (int i = 0; i < d2; ++i) z[i] = x_in[j * d2 + i];
  T ls = 0;
  for (int s = 0; s < hi - lo; ++s) {
    const int l = inverse ? lo + s : hi - 1 - s;
    if (l == a.n) {  // reference map z = shift0 + scale0 .* x0
      const T *sh = theta, *sc = theta + d2;
      for (int i = 0; i < d2; ++i) {
        z[i] = inverse ? (z[i] - sh[i]) / sc[i] : sh[i] + sc[i] * z[i];
        ls += (inverse ? -(T)1 : (T)1) * log(fabs(sc[i]));
      }
    } else {
      const T *shr = theta + 4 * D + 3 * D * l, *scr = shr + D, *leps = scr + D;
      if (!inverse) {
        hf_leapfrog<T>(a, leps, (T)1, z, z + D, nullptr);
        for (int i = 0; i < D; ++i) {
          z[D + i] = shr[i] + scr[i] * z[D + i];
          ls += log(fabs(scr[i]));
        }
      } else {
        for (int i = 0; i < D; ++i) {
          z[D + i] = (z[D + i] - shr[i]) / scr[i];
          ls -= log(fabs(scr[i]));
        }
        hf_leapfrog<T>(a, leps, -(T)1, z, z + D, nullptr);
      }
    }
  }
  for (int i = 0; i < d2; ++i) y_out[j * d2 + i] = z[i];
  ladj[j] = ls;
}

// Reverse pass of hf_leapfrog (either time direction: eps carries the sign) from its recorded states tr:
// (xb, vb) = cotangents of (x_L, v_L) on entry, of (x_0, rho) on exit; ebar = d/d eps (per dimension).
template <class T>
__device__ void hf_leapfrog_bwd(const HfArgs &a, const T *eps, T *tr, T *xb, T *vb, T *ebar) {
  const int D = a.D, L = a.L;
  T tmp[HF_MAXD], hv[HF_MAXD];
  auto V = [&](int s, int i) { return tr[(s * 3 + 1) * HF_MAXD + i]; };
  auto G = [&](int s, int i) { return tr[(s * 3 + 2) * HF_MAXD + i]; };
  // v_L = v_{L-1} + eps/2 g(x_L);  x_L = x_{L-1} + eps v_{L-1}
  for (int i = 0; i < D; ++i) tmp[i] = eps[i] / (T)2 * vb[i];
  hf_hvp<T>(a, &tr[(L * 3 + 0) * HF_MAXD], tmp, hv);
  for (int i = 0; i < D; ++i) {
    xb[i] += hv[i];
    ebar[i] += vb[i] * G(L, i) / (T)2 + xb[i] * V(L - 1, i);
    vb[i] += eps[i] * xb[i];
  }
  for (int s = L - 1; s >= 1; --s) {  // v_s = v_{s-1} + eps g(x_s);  x_s = x_{s-1} + eps v_{s-1}
    for (int i = 0; i < D; ++i) tmp[i] = eps[i] * vb[i];
    hf_hvp<T>(a, &tr[(s * 3 + 0) * HF_MAXD], tmp, hv);
    for (int i = 0; i < D; ++i) {
      xb[i] += hv[i];
      ebar[i] += vb[i] * G(s, i) + xb[i] * V(s - 1, i);
      vb[i] += eps[i] * xb[i];
    }
  }
  for (int i = 0; i < D; ++i) tmp[i] = eps[i] / (T)2 * vb[i];  // v_0 = rho + eps/2 g(x_0)
  hf_hvp<T>(a, &tr[0], tmp, hv);
  for (int i = 0; i < D; ++i) {
    xb[i] += hv[i];
    ebar[i] += vb[i] * G(0, i) / (T)2;
  }
}

// Parameter gradients without atomics (round 3): a workgroup is one wavefront; every contribution is summed over the wave's
// 64 samples (fixed butterfly order) and added by lane 0 to the workgroup's accumulator row in LDS; after the wave's last
// tile the row goes to slab[blockIdx.x][P], and nf_launch_reduce_slabs adds the slabs in block order: deterministic.
template <class T>
__device__ __forceinline__ void hf_acc(T *__restrict__ row, long idx, T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (threadIdx.x == 0) row[idx] += v;
}
#define HF_MAX_BLOCKS 2048

// reverse pass of the whole chain at the flow INPUT x0: ybar -> xbar (in gbar), dL/dtheta to slab[blockIdx.x][P].
// ws: [n][N][2D] inputs of every block (written here).
template <class T>
__global__ __launch_bounds__(HF_BLOCK) void k_hf_bwd(HfArgs a, const T *__restrict__ theta, const T *__restrict__ x0,
                                                     T *gbar, const T *__restrict__ lbar, T lbar_const,
                                                     T *__restrict__ ws, T *__restrict__ slab, long P) {
  extern __shared__ __attribute__((aligned(16))) char hf_smem[];
  T *g = (T *)hf_smem;  // [P] this workgroup's sums
  for (long i = threadIdx.x; i < P; i += HF_BLOCK) g[i] = 0;
  __syncthreads();
  const int D = a.D, d2 = 2 * D, n = a.n;
  for (long j0 = (long)blockIdx.x * HF_BLOCK; j0 < a.N; j0 += (long)gridDim.x * HF_BLOCK) {
  const bool valid = j0 + threadIdx.x < a.N;
  const long j = valid ? j0 + threadIdx.x : a.N - 1;  // idle lanes shadow the last sample with zero cotangents
  const T lb = valid ? (lbar ? lbar[j] : lbar_const) : (T)0;
  T z[2 * HF_MAXD];
  const T *sh0 = theta, *sc0 = theta + d2;
  for (int i = 0; i < d2; ++i) z[i] = sh0[i] + sc0[i] * x0[j * d2 + i];
  for (int bi = n - 1; bi >= 0; --bi) {  // forward, remembering what enters each block
    T *slot = ws + ((long)bi * a.N + j) * d2;
    if (valid)
      for (int i = 0; i < d2; ++i) slot[i] = z[i];
    const T *shr = theta + 4 * D + 3 * D * bi, *scr = shr + D, *leps = scr + D;
    hf_leapfrog<T>(a, leps, (T)1, z, z + D, nullptr);
    for (int i = 0; i < D; ++i) z[D + i] = shr[i] + scr[i] * z[D + i];
  }
  T zb[2 * HF_MAXD], tr[(HF_MAXL + 1) * 3 * HF_MAXD], eps[HF_MAXD], ebar[HF_MAXD];
  for (int i = 0; i < d2; ++i) zb[i] = valid ? gbar[j * d2 + i] : (T)0;
  for (int bi = 0; bi < n; ++bi) {  // reverse of execution order
    const long o0 = 4 * D + 3 * D * bi;
    const T *shr = theta + o0, *scr = shr + D, *leps = scr + D;
    const T *slot = ws + ((long)bi * a.N + j) * d2;
    for (int i = 0; i < d2; ++i) z[i] = slot[i];
    hf_leapfrog<T>(a, leps, (T)1, z, z + D, tr);  // z = (x_L, v_L): the momentum layer's input is v_L
    T *xb = zb, *vb = zb + D;
    for (int i = 0; i < D; ++i) {
      eps[i] = exp(leps[i]);
      ebar[i] = 0;
      hf_acc<T>(g, o0 + i, vb[i]);                              // shift_rho
      hf_acc<T>(g, o0 + D + i, vb[i] * z[D + i] + lb / scr[i]);  // scale_rho (+ d ladj / d scale)
      vb[i] *= scr[i];
    }
    hf_leapfrog_bwd<T>(a, eps, tr, xb, vb, ebar);
    for (int i = 0; i < D; ++i) hf_acc<T>(g, o0 + 2 * D + i, ebar[i] * eps[i]);  // d/d log_eps
  }
  for (int i = 0; i < d2; ++i) {  // reference map
    const T xi = x0[j * d2 + i];
    hf_acc<T>(g, i, zb[i]);
    hf_acc<T>(g, d2 + i, zb[i] * xi + lb / sc0[i]);
    if (valid) gbar[j * d2 + i] = zb[i] * sc0[i];
  }
  }
  __syncthreads();
  for (long i = threadIdx.x; i < P; i += HF_BLOCK) slab[(long)blockIdx.x * P + i] = g[i];
}

// Reverse pass of the INVERSE chain (forward-KL training, `train_flow(loglikelihood, flow, xs)`).  The inverse of
// every layer here is explicit -- the momentum layer's affine inverse, LeapFrog with -eps (demo :74-84), the
// reference map's affine inverse -- so it is differentiated directly: u = data, gbar in = cotangent of the base
// point x0 = T^-1(u), lb = cotangent of ladj_inv = -sum log|scale|.  ws: [n][N][2D] inputs of every inverse block.
template <class T>
__global__ __launch_bounds__(HF_BLOCK) void k_hf_bwd_inv(HfArgs a, const T *__restrict__ theta, const T *__restrict__ u,
                                                         const T *__restrict__ gbar, T lb_all, T *__restrict__ ws,
                                                         T *__restrict__ slab, long P) {
  extern __shared__ __attribute__((aligned(16))) char hf_smem[];
  T *g = (T *)hf_smem;  // [P] this workgroup's sums
  for (long i = threadIdx.x; i < P; i += HF_BLOCK) g[i] = 0;
  __syncthreads();
  const int D = a.D, d2 = 2 * D, n = a.n;
  for (long j0 = (long)blockIdx.x * HF_BLOCK; j0 < a.N; j0 += (long)gridDim.x * HF_BLOCK) {
  const bool valid = j0 + threadIdx.x < a.N;
  const long j = valid ? j0 + threadIdx.x : a.N - 1;
  const T lb = valid ? lb_all : (T)0;
  T z[2 * HF_MAXD];
  for (int i = 0; i < d2; ++i) z[i] = u[j * d2 + i];
  for (int bi = 0; bi < n; ++bi) {  // inverse chain: flat blocks 0 .. n-1, then the reference map
    T *slot = ws + ((long)bi * a.N + j) * d2;
    if (valid)
      for (int i = 0; i < d2; ++i) slot[i] = z[i];
    const T *shr = theta + 4 * D + 3 * D * bi, *scr = shr + D, *leps = scr + D;
    for (int i = 0; i < D; ++i) z[D + i] = (z[D + i] - shr[i]) / scr[i];
    hf_leapfrog<T>(a, leps, -(T)1, z, z + D, nullptr);
  }
  const T *sh0 = theta, *sc0 = theta + d2;
  T zb[2 * HF_MAXD], tr[(HF_MAXL + 1) * 3 * HF_MAXD], eps[HF_MAXD], ebar[HF_MAXD];
  for (int i = 0; i < d2; ++i) {  // x0 = (z - sh0) / sc0
    const T x0 = (z[i] - sh0[i]) / sc0[i];
    const T b = valid ? gbar[j * d2 + i] / sc0[i] : (T)0;
    hf_acc<T>(g, i, -b);
    hf_acc<T>(g, d2 + i, -b * x0 - lb / sc0[i]);
    zb[i] = b;
  }
  for (int bi = n - 1; bi >= 0; --bi) {
    const long o0 = 4 * D + 3 * D * bi;
    const T *shr = theta + o0, *scr = shr + D, *leps = scr + D;
    const T *slot = ws + ((long)bi * a.N + j) * d2;
    T rp[HF_MAXD];
    for (int i = 0; i < D; ++i) {
      z[i] = slot[i];
      rp[i] = (slot[D + i] - shr[i]) / scr[i];  // rho' = the integrator's initial momentum
      z[D + i] = rp[i];
      eps[i] = -exp(leps[i]);
      ebar[i] = 0;
    }
    hf_leapfrog<T>(a, leps, -(T)1, z, z + D, tr);
    T *xb = zb, *vb = zb + D;
    hf_leapfrog_bwd<T>(a, eps, tr, xb, vb, ebar);
    for (int i = 0; i < D; ++i) {
      hf_acc<T>(g, o0 + 2 * D + i, ebar[i] * eps[i]);  // d(-exp(log_eps)) / d log_eps = eps (signed)
      const T b = vb[i] / scr[i];
      hf_acc<T>(g, o0 + i, -b);
      hf_acc<T>(g, o0 + D + i, -b * rp[i] - lb / scr[i]);
      vb[i] = b;
    }
  }
  }
  __syncthreads();
  for (long i = threadIdx.x; i < P; i += HF_BLOCK) slab[(long)blockIdx.x * P + i] = g[i];
}

// ---- host side --------------------------------------------------------------------------------
bool nf_hf_supported(const nf_flow_desc *desc) {
  if (desc->kind != NF_KIND_HAMILTONIAN) return false;
  if (desc->d < 2 || (desc->d & 1) || desc->d / 2 > HF_MAXD || desc->nlayers < 1) return false;
  if (desc->K < 1 || desc->K > HF_MAXL || !desc->score) return false;
  const int tk = desc->score->kind;
  if (tk == NF_TARGET_DIAGGAUSS) return desc->score->p0 && desc->score->p1;
  if (tk == NF_TARGET_BANANA) return desc->d / 2 >= 2 && desc->score->s1 > 0;
  if (tk == NF_TARGET_FUNNEL) return desc->d / 2 >= 2 && desc->score->s1 > 0;
  return false;  // WarpedGauss / Cross: no closed-form Hessian-vector product here
}

static HfArgs make_hf_args(const nf_flow_desc *desc, long N) {
  HfArgs a;
  a.D = desc->d / 2; a.n = desc->nlayers; a.L = desc->K; a.tkind = desc->score->kind;
  a.p0 = desc->score->p0; a.p1 = desc->score->p1; a.s0 = desc->score->s0; a.s1 = desc->score->s1; a.N = N;
  return a;
}

int nf_hf_apply(nf_ctx *ctx, const nf_flow_desc *desc, int lo, int hi, bool inverse, const void *theta, const void *x,
                long N, void *y, void *ladj) {
  if (N <= 0) return NF_OK;
  const HfArgs a = make_hf_args(desc, N);
  const unsigned grid = (unsigned)((N + HF_BLOCK - 1) / HF_BLOCK);
  ProfScope ps(ctx, "hf_apply");
  if (desc->dtype == NF_DTYPE_F64)
    hipLaunchKernelGGL(k_hf_apply<double>, dim3(grid), dim3(HF_BLOCK), 0, ctx->stream, a, lo, hi, inverse ? 1 : 0,
                       (const double *)theta, (const double *)x, (double *)y, (double *)ladj);
  else
    hipLaunchKernelGGL(k_hf_apply<float>, dim3(grid), dim3(HF_BLOCK), 0, ctx->stream, a, lo, hi, inverse ? 1 : 0,
                       (const float *)theta, (const float *)x, (float *)y, (float *)ladj);
  return (int)hipGetLastError();
}

static inline long hf_param_count(const nf_flow_desc *desc) { return 2L * desc->d + 3L * (desc->d / 2) * desc->nlayers; }
static inline unsigned hf_bwd_grid(long N) {
  const long nb = (N + HF_BLOCK - 1) / HF_BLOCK;
  return (unsigned)(nb < HF_MAX_BLOCKS ? (nb < 1 ? 1 : nb) : HF_MAX_BLOCKS);
}
// the blocks' remembered inputs [n][N][2D], then the gradient slabs [blocks][P] (element size of the widest type)
static inline size_t hf_stash_bytes(const nf_flow_desc *desc, long N) { return (((size_t)desc->nlayers * (size_t)N * desc->d * sizeof(double)) + 255) / 256 * 256; }
size_t nf_hf_bwd_ws_bytes(const nf_flow_desc *desc, long N) {
  return hf_stash_bytes(desc, N) + (size_t)hf_bwd_grid(N) * (size_t)hf_param_count(desc) * sizeof(double);
}
int nf_launch_reduce_slabs(nf_ctx *ctx, int dtype, const void *slab, int nslab, long P, void *g);

int nf_hf_bwd(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x, const void *ybar, const void *lbar,
              double lbar_const, long N, void *xbar_out, void *gtheta_out, void *ws) {
  const size_t es = desc->dtype == NF_DTYPE_F64 ? 8 : 4;
  const long P = hf_param_count(desc);
  if (N <= 0) {
    NF_HIP(hipMemsetAsync(gtheta_out, 0, (size_t)P * es, ctx->stream));
    return NF_OK;
  }
  if (xbar_out != ybar) NF_HIP(hipMemcpyAsync(xbar_out, ybar, (size_t)N * desc->d * es, hipMemcpyDeviceToDevice, ctx->stream));
  const HfArgs a = make_hf_args(desc, N);
  const unsigned grid = hf_bwd_grid(N);
  char *slab = (char *)ws + hf_stash_bytes(desc, N);
  {
    ProfScope ps(ctx, "hf_bwd");
    if (desc->dtype == NF_DTYPE_F64)
      hipLaunchKernelGGL(k_hf_bwd<double>, dim3(grid), dim3(HF_BLOCK), (size_t)P * 8, ctx->stream, a, (const double *)theta,
                         (const double *)x, (double *)xbar_out, (const double *)lbar, lbar_const, (double *)ws, (double *)slab, P);
    else
      hipLaunchKernelGGL(k_hf_bwd<float>, dim3(grid), dim3(HF_BLOCK), (size_t)P * 4, ctx->stream, a, (const float *)theta,
                         (const float *)x, (float *)xbar_out, (const float *)lbar, (float)lbar_const, (float *)ws, (float *)slab, P);
    NF_HIP(hipGetLastError());
  }
  return nf_launch_reduce_slabs(ctx, desc->dtype, slab, (int)grid, P, gtheta_out);
}

// forward-KL reverse pass: u = data (d x N), gbar = cotangent of T^-1(u), lbar_const = cotangent of ladj_inv
int nf_hf_bwd_inv(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *u, const void *gbar, double lbar_const,
                  long N, void *gtheta_out, void *ws) {
  const size_t es = desc->dtype == NF_DTYPE_F64 ? 8 : 4;
  const long P = hf_param_count(desc);
  if (N <= 0) {
    NF_HIP(hipMemsetAsync(gtheta_out, 0, (size_t)P * es, ctx->stream));
    return NF_OK;
  }
  const HfArgs a = make_hf_args(desc, N);
  const unsigned grid = hf_bwd_grid(N);
  char *slab = (char *)ws + hf_stash_bytes(desc, N);
  {
    ProfScope ps(ctx, "hf_bwd");
    if (desc->dtype == NF_DTYPE_F64)
      hipLaunchKernelGGL(k_hf_bwd_inv<double>, dim3(grid), dim3(HF_BLOCK), (size_t)P * 8, ctx->stream, a, (const double *)theta,
                         (const double *)u, (const double *)gbar, lbar_const, (double *)ws, (double *)slab, P);
    else
      hipLaunchKernelGGL(k_hf_bwd_inv<float>, dim3(grid), dim3(HF_BLOCK), (size_t)P * 4, ctx->stream, a, (const float *)theta,
                         (const float *)u, (const float *)gbar, (float)lbar_const, (float *)ws, (float *)slab, P);
    NF_HIP(hipGetLastError());
  }
  return nf_launch_reduce_slabs(ctx, desc->dtype, slab, (int)grid, P, gtheta_out);
}
