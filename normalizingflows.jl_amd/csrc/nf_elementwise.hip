// nf_elementwise.hip -- HBM-bound kernels around the coupling chain (gfx950):
//   base sampler + log q0 (a4, a5), built-in targets and ELBO assembly (a2, a14),
//   partial-sum reductions, gradient-slab reduction, Adam + gradient norm (a15).
//
// Layout reminder: batch is d x N, sample j contiguous (x[j*d + i]).  These kernels use
// LPS = 16 lanes per sample: lane q of a sample group owns features q, q+16, q+32, ... so a
// 16-lane group reads 64 contiguous bytes per step, and per-sample sums (||x||^2, log p)
// are 4-step DPP/shuffle reductions inside the group.
#include <type_traits>

#include "nf_common.h"
#include "nf_philox.h"
#include "nf_targets.h"

#define LPS 16
#define EW_BLOCK 256
#define SPB (EW_BLOCK / LPS)  // samples per block

template <class T>
__device__ __forceinline__ T group16_sum(T v) {
  v += __shfl_xor(v, 8, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 1, 16);
  return v;
}

// block-wide sum of one double per thread -> thread 0 (EW_BLOCK threads)
__device__ __forceinline__ double block_sum(double v, double *sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sm[w];
  __syncthreads();
  return r;
}


// x ~ N(0, I), logq = logpdf(MvNormal(0, I), x).  Reference seam: _device_specific_rand
// (src/NormalizingFlows.jl:109-115; device version ext/NormalizingFlowsCUDAExt.jl:43-48).
template <class T>
__global__ __launch_bounds__(EW_BLOCK) void k_base_sample(int d, long N, uint32_t k0, uint32_t k1, uint64_t off,
                                                          uint32_t stream, T *__restrict__ x, T *__restrict__ logq) {
  const int q = threadIdx.x & (LPS - 1);
  const long j = (long)blockIdx.x * SPB + (threadIdx.x / LPS);
  const bool valid = j < N;
  const int ng = (d + 3) / 4;
  T ss = 0;
  if (valid) {
    const uint64_t gj = off + (uint64_t)j;
    for (int g = q; g < ng; g += LPS) {
      T z[4];
      philox_normals4<T>(gj, (uint32_t)g, stream, k0, k1, z);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * g + e;
        if (i < d) {
          x[j * d + i] = z[e];
          ss += z[e] * z[e];
        }
      }
    }
  }
  ss = group16_sum(ss);
  if (valid && q == 0 && logq) logq[j] = (T)(-0.5 * 1.8378770664093453 * d) - (T)0.5 * ss;
}

template <class T>
__global__ __launch_bounds__(EW_BLOCK) void k_base_logpdf(int d, long N, const T *__restrict__ x, T *__restrict__ logq) {
  const int q = threadIdx.x & (LPS - 1);
  const long j = (long)blockIdx.x * SPB + (threadIdx.x / LPS);
  const bool valid = j < N;
  T ss = 0;
  if (valid)
    for (int i = q; i < d; i += LPS) {
      const T v = x[j * d + i];
      ss += v * v;
    }
  ss = group16_sum(ss);
  if (valid && q == 0) logq[j] = (T)(-0.5 * 1.8378770664093453 * d) - (T)0.5 * ss;
}

// ---------------------------------------------------------------------------------------
// general MvNormal(mu, Sigma) base distributions
// ---------------------------------------------------------------------------------------
// _device_specific_rand(rng, ::MvNormal, n) draws x = mu + L eps with Sigma = L L' ("unwhiten", Distributions'
// _rand!; the device version is ext/NormalizingFlowsCUDAExt.jl:43-48; a dense Sigma is what test/ext/CUDA/cuda.jl:33-45
// exercises), and logpdf(flow.dist, xs) (src/objectives/elbo.jl:6,68) is
//   -d/2 log 2pi - log|det L| - 1/2 ||L^-1 (x - mu)||^2.
// kind 1: Sigma = Diagonal(sigma^2), scale = sigma[d];  kind 2: dense, scale = L, d x d lower triangular, column-major.
// One thread per sample for the dense case (a d^2 / 2 triangular product / substitution per sample; d <= 256 here and
// the base is a cold path next to the flow), in place: eps -> x.
template <class T>
__global__ __launch_bounds__(EW_BLOCK) void k_base_unwhiten(int kind, int d, long N, const T *__restrict__ mu,
                                                            const T *__restrict__ scale, T *__restrict__ x) {
  const long j = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  if (j >= N) return;
  T *r = x + j * d;
  if (kind == NF_BASE_DIAG) {
    for (int i = 0; i < d; ++i) r[i] = mu[i] + scale[i] * r[i];
  } else {
    // x_i = mu_i + sum_{k <= i} L[i][k] eps_k: rows from the bottom up, so eps is still intact where it is read
    for (int i = d - 1; i >= 0; --i) {
      T acc = mu[i];
      for (int k = 0; k <= i; ++k) acc += scale[(long)k * d + i] * r[k];
      r[i] = acc;
    }
  }
}

// logq_out[j] (optional) = logpdf(base, x_j);  corr_out[j] (optional) = logpdf(MvNormal(0, I), x_j) - logpdf(base, x_j):
// what must be ADDED to an ELBO term that was assembled with the standard-normal log q0.
template <class T>
__global__ __launch_bounds__(EW_BLOCK) void k_base_general_logpdf(int kind, int d, long N, const T *__restrict__ mu,
                                                                  const T *__restrict__ scale, T logdet,
                                                                  const T *__restrict__ x, T *__restrict__ logq_out,
                                                                  T *__restrict__ corr_out, T *__restrict__ zbuf) {
  const long j = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  if (j >= N) return;
  const T *r = x + j * d;
  T ss = 0, s0 = 0;
  if (kind == NF_BASE_DIAG) {
    for (int i = 0; i < d; ++i) {
      const T z = (r[i] - mu[i]) / scale[i];
      ss += z * z;
      s0 += r[i] * r[i];
    }
  } else {
    T *z = zbuf + j * d;  // forward substitution L z = x - mu
    for (int i = 0; i < d; ++i) {
      T acc = r[i] - mu[i];
      for (int k = 0; k < i; ++k) acc -= scale[(long)k * d + i] * z[k];
      const T zi = acc / scale[(long)i * d + i];
      z[i] = zi;
      ss += zi * zi;
      s0 += r[i] * r[i];
    }
  }
  const T c0 = (T)(-0.5 * 1.8378770664093453 * d);
  const T lq = c0 - logdet - (T)0.5 * ss;
  if (logq_out) logq_out[j] = lq;
  if (corr_out) corr_out[j] = (c0 - (T)0.5 * s0) - lq;
}

// logq_out[j] = logpdf(base, x_j) and score_out[j] = gscale * d logpdf(base, x_j) / dx = -gscale * Sigma^-1 (x_j - mu):
// the seed of the forward-KL reverse pass for a general base (src/objectives/loglikelihood.jl:27-36 with flow.dist a
// general MvNormal).  Dense: L z = x - mu forwards, then L' w = z backwards, both in the sample's row of zbuf.
template <class T>
__global__ __launch_bounds__(EW_BLOCK) void k_base_general_score(int kind, int d, long N, const T *__restrict__ mu,
                                                                 const T *__restrict__ scale, T logdet,
                                                                 const T *__restrict__ x, T *__restrict__ logq_out,
                                                                 T *__restrict__ score_out, T gscale, T *__restrict__ zbuf) {
  const long j = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  if (j >= N) return;
  const T *r = x + j * d;
  T *g = score_out + j * d;
  T ss = 0;
  if (kind == NF_BASE_DIAG) {
    for (int i = 0; i < d; ++i) {
      const T z = (r[i] - mu[i]) / scale[i];
      ss += z * z;
      g[i] = -gscale * z / scale[i];
    }
  } else {
    T *z = zbuf + j * d;
    for (int i = 0; i < d; ++i) {
      T acc = r[i] - mu[i];
      for (int k = 0; k < i; ++k) acc -= scale[(long)k * d + i] * z[k];
      const T zi = acc / scale[(long)i * d + i];
      z[i] = zi;
      ss += zi * zi;
    }
    for (int i = d - 1; i >= 0; --i) {  // (L')[i][k] = L[k][i], k > i: column i of L below the diagonal
      T acc = z[i];
      for (int k = i + 1; k < d; ++k) acc -= scale[(long)i * d + k] * z[k];
      const T wi = acc / scale[(long)i * d + i];
      z[i] = wi;
      g[i] = -gscale * wi;
    }
  }
  if (logq_out) logq_out[j] = (T)(-0.5 * 1.8378770664093453 * d) - logdet - (T)0.5 * ss;
}

int nf_launch_base_unwhiten(nf_ctx *ctx, int dtype, int kind, int d, long N, const void *mu, const void *scale, void *x) {
  if (N <= 0) return NF_OK;
  const unsigned grid = (unsigned)((N + EW_BLOCK - 1) / EW_BLOCK);
  ProfScope ps(ctx, "base_unwhiten");
  if (dtype == NF_DTYPE_F32)
    hipLaunchKernelGGL(k_base_unwhiten<float>, dim3(grid), dim3(EW_BLOCK), 0, ctx->stream, kind, d, N, (const float *)mu,
                       (const float *)scale, (float *)x);
  else
    hipLaunchKernelGGL(k_base_unwhiten<double>, dim3(grid), dim3(EW_BLOCK), 0, ctx->stream, kind, d, N, (const double *)mu,
                       (const double *)scale, (double *)x);
  return (int)hipGetLastError();
}

int nf_launch_base_general_logpdf(nf_ctx *ctx, int dtype, int kind, int d, long N, const void *mu, const void *scale,
                                  double logdet, const void *x, void *logq_out, void *corr_out, void *zbuf) {
  if (N <= 0) return NF_OK;
  const unsigned grid = (unsigned)((N + EW_BLOCK - 1) / EW_BLOCK);
  ProfScope ps(ctx, "base_logpdf");
  if (dtype == NF_DTYPE_F32)
    hipLaunchKernelGGL(k_base_general_logpdf<float>, dim3(grid), dim3(EW_BLOCK), 0, ctx->stream, kind, d, N, (const float *)mu,
                       (const float *)scale, (float)logdet, (const float *)x, (float *)logq_out, (float *)corr_out, (float *)zbuf);
  else
    hipLaunchKernelGGL(k_base_general_logpdf<double>, dim3(grid), dim3(EW_BLOCK), 0, ctx->stream, kind, d, N, (const double *)mu,
                       (const double *)scale, logdet, (const double *)x, (double *)logq_out, (double *)corr_out, (double *)zbuf);
  return (int)hipGetLastError();
}

int nf_launch_base_general_score(nf_ctx *ctx, int dtype, int kind, int d, long N, const void *mu, const void *scale,
                                 double logdet, const void *x, void *logq_out, void *score_out, double gscale, void *zbuf) {
  if (N <= 0) return NF_OK;
  const unsigned grid = (unsigned)((N + EW_BLOCK - 1) / EW_BLOCK);
  ProfScope ps(ctx, "base_score");
  if (dtype == NF_DTYPE_F32)
    hipLaunchKernelGGL(k_base_general_score<float>, dim3(grid), dim3(EW_BLOCK), 0, ctx->stream, kind, d, N, (const float *)mu,
                       (const float *)scale, (float)logdet, (const float *)x, (float *)logq_out, (float *)score_out,
                       (float)gscale, (float *)zbuf);
  else
    hipLaunchKernelGGL(k_base_general_score<double>, dim3(grid), dim3(EW_BLOCK), 0, ctx->stream, kind, d, N, (const double *)mu,
                       (const double *)scale, logdet, (const double *)x, (double *)logq_out, (double *)score_out, gscale,
                       (double *)zbuf);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// targets + ELBO assembly
// ---------------------------------------------------------------------------------------
// target_term<kind, T> and target_needs_d2: nf_targets.h (shared with the fused forward of nf_simple.hip)

// logp per sample, optional outputs:
//   logp_out[j]                                 (nf_target_logp)
//   grad_out[j*d+i] = gscale * dlogp/dy_i        (gscale = 1 for the API, -1/N for the loss)
//   elbos_out[j]   = logp - logq + ladj          (src/objectives/elbo.jl:68)
//   partial[blockIdx.x] = sum over the block's samples of pscale * (logp - logq + ladj)
template <class T>
__global__ __launch_bounds__(EW_BLOCK) void k_target(int kind, int d, long N, const T *__restrict__ y,
                                                     const T *__restrict__ mu, const T *__restrict__ var, T b_ban,
                                                     T var_ban, const T *__restrict__ logq, const T *__restrict__ ladj,
                                                     T *__restrict__ logp_out, T *__restrict__ grad_out, T gscale,
                                                     T *__restrict__ elbos_out, double *__restrict__ partial,
                                                     double pscale, int dj) {
  // dj > 0: joint density of the Hamiltonian flows -- the target describes the first dj coordinates,
  // the remaining d - dj (momenta) are standard normal (logp_joint, example/demo_hamiltonian_flow.jl:121-128)
  const int dt_ = dj > 0 ? dj : d;
  __shared__ double sm[EW_BLOCK / 64];
  const int q = threadIdx.x & (LPS - 1);
  const long j = (long)blockIdx.x * SPB + (threadIdx.x / LPS);
  const bool valid = j < N;
  T acc = 0;
  T s2 = 0;
  if (kind == NF_TARGET_FUNNEL) {  // sum_{i>=1} y_i^2 of the sample, needed by feature 0's gradient
    if (valid)
      for (int i = q; i < dt_; i += LPS)
        if (i >= 1) s2 += y[j * d + i] * y[j * d + i];
    s2 = group16_sum(s2);
  }
  if (valid) {
    const T *yr = y + j * d;
    const T y0 = yr[0], y1 = dt_ > 1 ? yr[1] : (T)0;
    auto run = [&](auto kc) {  // the target kind is resolved once, outside the feature loop
      constexpr int KD = decltype(kc)::value;
      for (int i = q; i < d; i += LPS) {
        T g;
        if (i < dt_) {
          acc += target_term<KD, T>(dt_, i, yr[i], y0, y1, s2, mu, var, b_ban, var_ban, g);
        } else {
          g = -yr[i];
          acc += (T)-0.5 * ((T)1.8378770664093453 + yr[i] * yr[i]);
        }
        if (grad_out) grad_out[j * d + i] = gscale * g;
      }
    };
    switch (kind) {
      case NF_TARGET_DIAGGAUSS: run(std::integral_constant<int, NF_TARGET_DIAGGAUSS>{}); break;
      case NF_TARGET_BANANA: run(std::integral_constant<int, NF_TARGET_BANANA>{}); break;
      case NF_TARGET_FUNNEL: run(std::integral_constant<int, NF_TARGET_FUNNEL>{}); break;
      case NF_TARGET_WARPED: run(std::integral_constant<int, NF_TARGET_WARPED>{}); break;
      default: run(std::integral_constant<int, NF_TARGET_CROSS>{}); break;
    }
  }
  acc = group16_sum(acc);
  double contrib = 0.0;
  if (valid && q == 0) {
    if (logp_out) logp_out[j] = acc;
    T e = acc;
    if (logq) e -= logq[j];
    if (ladj) e += ladj[j];
    if (elbos_out) elbos_out[j] = e;
    contrib = pscale * (double)e;
  }
  if (partial) {
    const double s = block_sum(contrib, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
  }
}

// partial sums of (a[j] + b[j]) * scale over blocks (loglikelihood: log q0(x) + ladj)
template <class T>
__global__ __launch_bounds__(EW_BLOCK) void k_sum2(long N, const T *__restrict__ a, const T *__restrict__ b,
                                                   T *__restrict__ out, double *__restrict__ partial, double scale) {
  __shared__ double sm[EW_BLOCK / 64];
  const long j = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  double c = 0.0;
  if (j < N) {
    const T v = a[j] + (b ? b[j] : (T)0);
    if (out) out[j] = v;
    c = scale * (double)v;
  }
  const double s = block_sum(c, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// deterministic final reduction of block partials (single block); optional sqrt.
// dst_d (double*) and/or dst_f (float*, element `idx`) receive the result.
__global__ __launch_bounds__(EW_BLOCK) void k_finish_sum(const double *__restrict__ partial, long n, int take_sqrt,
                                                         double *__restrict__ dst_d, float *__restrict__ dst_f,
                                                         double *__restrict__ dst_d2, unsigned *__restrict__ bump) {
  __shared__ double sm[EW_BLOCK / 64];
  double c = 0.0;
  for (long i = threadIdx.x; i < n; i += EW_BLOCK) c += partial[i];
  double s = block_sum(c, sm);
  if (threadIdx.x == 0) {
    if (take_sqrt) s = sqrt(s);
    if (dst_d) *dst_d = s;
    if (dst_f) *dst_f = (float)s;
    if (dst_d2) *dst_d2 = s;
    if (bump) *bump = *bump + 1u;  // the device-resident step counter of nf_elbo_step_enqueue
  }
}

// ---------------------------------------------------------------------------------------
// gradient slabs -> gradient ; Adam
// ---------------------------------------------------------------------------------------
// g[p] = sum_s slab[s*P + p]   (deterministic: fixed summation order)
template <class T>
__global__ __launch_bounds__(EW_BLOCK) void k_reduce_slabs(const T *__restrict__ slab, int nslab, long P,
                                                           T *__restrict__ g) {
  const long p = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  if (p >= P) return;
  T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int s = 0;
  for (; s + 3 < nslab; s += 4) {
    a0 += slab[(long)s * P + p];
    a1 += slab[(long)(s + 1) * P + p];
    a2 += slab[(long)(s + 2) * P + p];
    a3 += slab[(long)(s + 3) * P + p];
  }
  for (; s < nslab; ++s) a0 += slab[(long)s * P + p];
  g[p] = (a0 + a1) + (a2 + a3);
}

// Many slabs, few parameters (the general couplings' per-coupling sums: 1 024 slabs x 29 k floats): one thread per
// parameter is a chain of nslab dependent-latency loads on a third of the chip.  Here a block takes RS_PW parameters and
// splits the slab range over RS_PARTS thread rows -- each row the same four-accumulator walk over its contiguous share --
// and the rows are added in a fixed tree: the order depends on (nslab, P) only, so the sum stays reproducible.
#define RS_PW 32
#define RS_PARTS 8
template <class T>
__global__ __launch_bounds__(RS_PW * RS_PARTS) void k_reduce_slabs_split(const T *__restrict__ slab, int nslab, long P,
                                                                         T *__restrict__ g) {
  __shared__ T part[RS_PARTS][RS_PW];
  const int col = threadIdx.x % RS_PW, row = threadIdx.x / RS_PW;
  const long p = (long)blockIdx.x * RS_PW + col;
  const int per = (nslab + RS_PARTS - 1) / RS_PARTS;
  const int s0 = row * per, s1 = s0 + per < nslab ? s0 + per : nslab;
  T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  if (p < P) {
    int s = s0;
    for (; s + 3 < s1; s += 4) {
      a0 += slab[(long)s * P + p];
      a1 += slab[(long)(s + 1) * P + p];
      a2 += slab[(long)(s + 2) * P + p];
      a3 += slab[(long)(s + 3) * P + p];
    }
    for (; s < s1; ++s) a0 += slab[(long)s * P + p];
  }
  part[row][col] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (row == 0 && p < P) {
    static_assert(RS_PARTS == 8, "fixed tree");
    g[p] = ((part[0][col] + part[1][col]) + (part[2][col] + part[3][col])) +
           ((part[4][col] + part[5][col]) + (part[6][col] + part[7][col]));
  }
}

// Optimisers.Adam (Optimisers.jl 0.4 `apply!`): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// theta -= lr * (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps).  Also block partials of g^2
// for gradient_norm (src/optimize.jl:89).
template <class T>
__global__ __launch_bounds__(EW_BLOCK) void k_adam(T *__restrict__ theta, const T *__restrict__ g, T *__restrict__ m,
                                                   T *__restrict__ v, long P, T lr, T b1, T b2, T eps, T c1, T c2,
                                                   double *__restrict__ partial) {
  __shared__ double sm[EW_BLOCK / 64];
  const long p = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  double gg = 0.0;
  if (p < P) {
    const T gi = g[p];
    T th = theta[p], mi = m[p], vi = v[p];
    nf_adam_elem<T>(th, mi, vi, gi, lr, b1, b2, eps, c1, c2);
    m[p] = mi;
    v[p] = vi;
    theta[p] = th;
    gg = (double)gi * (double)gi;
  }
  if (partial) {
    const double s = block_sum(gg, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
  }
}

// Optimisers.Descent (vel == nullptr): theta -= lr g.  Optimisers.Momentum: vel = rho vel - lr g,
// theta += vel.  Same gradient-norm partials as k_adam.
template <class T>
__global__ __launch_bounds__(EW_BLOCK) void k_sgd(T *__restrict__ theta, const T *__restrict__ g, T *__restrict__ vel,
                                                  long P, T lr, T rho, double *__restrict__ partial) {
  __shared__ double sm[EW_BLOCK / 64];
  const long p = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  double gg = 0.0;
  if (p < P) {
    const T gi = g[p];
    if (vel) {
      const T vi = rho * vel[p] - lr * gi;
      vel[p] = vi;
      theta[p] += vi;
    } else {
      theta[p] -= lr * gi;
    }
    gg = (double)gi * (double)gi;
  }
  if (partial) {
    const double s = block_sum(gg, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------
// TILED batch layout used between the coupling kernels (internal to the library):
// samples are grouped in tiles of 32; element (feature f, sample s of tile t) lives at
// [(t * d + f) * 32 + s].  A half-wave then reads/writes one full 128-byte line per feature,
// which is exactly the register layout of the MFMA operands (nf_mfma.h).  Buffers are padded
// to whole tiles.  Here: one thread per sample (lane <-> sample), loops over features.
// ---------------------------------------------------------------------------------------
#define TL 32

// Thread mapping of the tiled element-wise kernels: a block of 256 threads = 32 samples (one tile,
// lane & 31) x 8 feature slices (threadIdx.x >> 5); slice q owns features q, q+8, ... (groups of 4
// for the sampler).  Every access is one 128-byte line per 32 lanes; per-sample sums are combined
// across the 8 slices through LDS.
#define FS 8  // feature slices per tile

__global__ __launch_bounds__(EW_BLOCK) void k_base_sample_tiled(int d, long N, uint32_t k0, uint32_t k1, uint64_t off,
                                                                uint32_t stream, float *__restrict__ xt,
                                                                float *__restrict__ logq) {
  __shared__ float red[FS][TL];
  const long tile = blockIdx.x;
  const int s = threadIdx.x & (TL - 1), q = threadIdx.x >> 5;
  const long j = tile * TL + s;
  const bool valid = j < N;
  float *base = xt + tile * d * TL + s;
  const int ng = (d + 3) / 4;
  const uint64_t gj = off + (uint64_t)j;
  float ss = 0.f;
  for (int g = q; g < ng; g += FS) {
    U4 c = {(uint32_t)gj, (uint32_t)(gj >> 32), (uint32_t)g, stream};
    const U4 r = philox4x32_10(c, k0, k1);
    float z[4];
    box_muller<float>(r.x, r.y, z[0], z[1]);
    box_muller<float>(r.z, r.w, z[2], z[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * g + e;
      if (i < d) {
        base[(long)i * TL] = valid ? z[e] : 0.f;  // padding samples are kept finite
        ss += z[e] * z[e];
      }
    }
  }
  red[q][s] = ss;
  __syncthreads();
  if (q == 0 && valid && logq) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < FS; ++k) t += red[k][s];
    logq[j] = (float)(-0.5 * 1.8378770664093453 * d) - 0.5f * t;
  }
}

__global__ __launch_bounds__(EW_BLOCK) void k_base_logpdf_tiled(int d, long N, const float *__restrict__ xt,
                                                                float *__restrict__ logq) {
  __shared__ float red[FS][TL];
  const long tile = blockIdx.x;
  const int s = threadIdx.x & (TL - 1), q = threadIdx.x >> 5;
  const long j = tile * TL + s;
  const float *base = xt + tile * d * TL + s;
  float ss = 0.f;
  for (int i = q; i < d; i += FS) {
    const float v = base[(long)i * TL];
    ss += v * v;
  }
  red[q][s] = ss;
  __syncthreads();
  if (q == 0 && j < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < FS; ++k) t += red[k][s];
    logq[j] = (float)(-0.5 * 1.8378770664093453 * d) - 0.5f * t;
  }
}

// tiled version of k_target (same outputs); one block per tile
__global__ __launch_bounds__(EW_BLOCK) void k_target_tiled(int kind, int d, long N, const float *__restrict__ yt,
                                                           const float *__restrict__ mu, const float *__restrict__ var,
                                                           float b_ban, float var_ban, const float *__restrict__ logq,
                                                           const float *__restrict__ ladj, float *__restrict__ gt,
                                                           float gscale, float *__restrict__ elbos_out,
                                                           double *__restrict__ partial, double pscale) {
  __shared__ float red[FS][TL];
  __shared__ double sm[EW_BLOCK / 64];
  const long tile = blockIdx.x;
  const int s = threadIdx.x & (TL - 1), q = threadIdx.x >> 5;
  const long j = tile * TL + s;
  const bool valid = j < N;
  const float *yb = yt + tile * d * TL + s;
  float *gb = gt ? gt + tile * d * TL + s : nullptr;
  float acc = 0.f;
  float s2 = 0.f;
  if (kind == NF_TARGET_FUNNEL) {
    for (int i = q; i < d; i += FS)
      if (i >= 1) s2 += yb[(long)i * TL] * yb[(long)i * TL];
    red[q][s] = s2;
    __syncthreads();
    s2 = 0.f;
#pragma unroll
    for (int k = 0; k < FS; ++k) s2 += red[k][s];
    __syncthreads();
  }
  {
    const float y0 = yb[0], y1 = d > 1 ? yb[TL] : 0.f;
    auto run = [&](auto kc) {
      constexpr int KD = decltype(kc)::value;
      for (int i = q; i < d; i += FS) {
        float g;
        acc += target_term<KD, float>(d, i, yb[(long)i * TL], y0, y1, s2, mu, var, b_ban, var_ban, g);
        if (gb) gb[(long)i * TL] = valid ? gscale * g : 0.f;
      }
    };
    switch (kind) {
      case NF_TARGET_DIAGGAUSS: run(std::integral_constant<int, NF_TARGET_DIAGGAUSS>{}); break;
      case NF_TARGET_BANANA: run(std::integral_constant<int, NF_TARGET_BANANA>{}); break;
      case NF_TARGET_FUNNEL: run(std::integral_constant<int, NF_TARGET_FUNNEL>{}); break;
      case NF_TARGET_WARPED: run(std::integral_constant<int, NF_TARGET_WARPED>{}); break;
      default: run(std::integral_constant<int, NF_TARGET_CROSS>{}); break;
    }
  }
  red[q][s] = acc;
  __syncthreads();
  double contrib = 0.0;
  if (q == 0 && valid) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < FS; ++k) t += red[k][s];
    float e = t;
    if (logq) e -= logq[j];
    if (ladj) e += ladj[j];
    if (elbos_out) elbos_out[j] = e;
    contrib = pscale * (double)e;
  }
  if (partial) {
    const double sum = block_sum(contrib, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = sum;
  }
}

// standard (sample-major, x[j*d + i]) <-> tiled.  One block per tile: the tile is a contiguous
// run of 32*d floats in the standard layout; it goes through LDS as a [32][d+1] transpose.
__global__ __launch_bounds__(EW_BLOCK) void k_layout_convert(int d, long N, const float *__restrict__ src,
                                                             float *__restrict__ dst, int to_tiled) {
  extern __shared__ float tsm[];  // 32 * (d + 1)
  const long tile = blockIdx.x;
  const long j0 = tile * TL;
  const int nvalid = (int)((N - j0) < TL ? (N - j0) : TL);
  const int total = TL * d;
  if (to_tiled) {
    for (int idx = threadIdx.x; idx < total; idx += EW_BLOCK) {
      const int sI = idx / d, f = idx - sI * d;
      tsm[sI * (d + 1) + f] = sI < nvalid ? src[j0 * d + idx] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < total; idx += EW_BLOCK) {
      const int f = idx / TL, sI = idx - f * TL;
      dst[tile * total + idx] = tsm[sI * (d + 1) + f];
    }
  } else {
    for (int idx = threadIdx.x; idx < total; idx += EW_BLOCK) {
      const int f = idx / TL, sI = idx - f * TL;
      tsm[sI * (d + 1) + f] = src[tile * total + idx];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < total; idx += EW_BLOCK) {
      const int sI = idx / d;
      if (sI < nvalid) dst[j0 * d + idx] = tsm[sI * (d + 1) + (idx - sI * d)];
    }
  }
}

template <class T>
__global__ void k_fill(T *p, long n, T v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------
// host launchers (typed on NF_DTYPE_*)
// ---------------------------------------------------------------------------------------
static inline unsigned nblk(long n, int per) { return (unsigned)((n + per - 1) / per > 0 ? (n + per - 1) / per : 1); }

int nf_launch_base_sample(nf_ctx *ctx, int dtype, int d, long N, uint64_t seed, uint64_t off, uint32_t stream, void *x,
                          void *logq) {
  if (N <= 0) return NF_OK;
  ProfScope ps(ctx, "base_sample");
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  if (dtype == NF_DTYPE_F32)
    hipLaunchKernelGGL(k_base_sample<float>, dim3(nblk(N, SPB)), dim3(EW_BLOCK), 0, ctx->stream, d, N, k0, k1, off,
                       stream, (float *)x, (float *)logq);
  else
    hipLaunchKernelGGL(k_base_sample<double>, dim3(nblk(N, SPB)), dim3(EW_BLOCK), 0, ctx->stream, d, N, k0, k1, off,
                       stream, (double *)x, (double *)logq);
  return (int)hipGetLastError();
}

int nf_launch_base_logpdf(nf_ctx *ctx, int dtype, int d, long N, const void *x, void *logq) {
  if (N <= 0) return NF_OK;
  if (dtype == NF_DTYPE_F32)
    hipLaunchKernelGGL(k_base_logpdf<float>, dim3(nblk(N, SPB)), dim3(EW_BLOCK), 0, ctx->stream, d, N, (const float *)x,
                       (float *)logq);
  else
    hipLaunchKernelGGL(k_base_logpdf<double>, dim3(nblk(N, SPB)), dim3(EW_BLOCK), 0, ctx->stream, d, N,
                       (const double *)x, (double *)logq);
  return (int)hipGetLastError();
}

// number of block partials k_target produces for a batch of N
// argument conventions of the built-in targets
int nf_target_check(const nf_target *t, int d) {
  switch (t->kind) {
    case NF_TARGET_DIAGGAUSS: return (t->p0 && t->p1) ? NF_OK : NF_ERR_ARG;
    case NF_TARGET_BANANA: return (d >= 2 && t->s1 > 0) ? NF_OK : NF_ERR_ARG;       // banana.jl:40-44
    case NF_TARGET_FUNNEL: return (d >= 2 && t->s1 > 0) ? NF_OK : NF_ERR_ARG;       // neal_funnel.jl:31-35
    case NF_TARGET_WARPED: return (d == 2 && t->s0 > 0 && t->s1 > 0) ? NF_OK : NF_ERR_ARG;  // warped_gaussian.jl:29-33,79
    case NF_TARGET_CROSS: return (d == 2 && t->s1 > 0) ? NF_OK : NF_ERR_ARG;
    default: return NF_ERR_ARG;
  }
}

long nf_target_nblocks(long N) { return nblk(N, SPB); }

int nf_launch_target(nf_ctx *ctx, int dtype, const nf_target *t, int d, long N, const void *y, const void *logq,
                     const void *ladj, void *logp_out, void *grad_out, double gscale, void *elbos_out, double *partial,
                     double pscale, int joint_d) {
  if (N <= 0) return NF_OK;
  NF_TRY(nf_target_check(t, joint_d > 0 ? joint_d : d));
  ProfScope ps(ctx, "target");
  if (dtype == NF_DTYPE_F32)
    hipLaunchKernelGGL(k_target<float>, dim3(nblk(N, SPB)), dim3(EW_BLOCK), 0, ctx->stream, t->kind, d, N,
                       (const float *)y, (const float *)t->p0, (const float *)t->p1, (float)t->s0, (float)t->s1,
                       (const float *)logq, (const float *)ladj, (float *)logp_out, (float *)grad_out, (float)gscale,
                       (float *)elbos_out, partial, pscale, joint_d);
  else
    hipLaunchKernelGGL(k_target<double>, dim3(nblk(N, SPB)), dim3(EW_BLOCK), 0, ctx->stream, t->kind, d, N,
                       (const double *)y, (const double *)t->p0, (const double *)t->p1, (double)t->s0, (double)t->s1,
                       (const double *)logq, (const double *)ladj, (double *)logp_out, (double *)grad_out, gscale,
                       (double *)elbos_out, partial, pscale, joint_d);
  return (int)hipGetLastError();
}

long nf_sum2_nblocks(long N) { return nblk(N, EW_BLOCK); }

int nf_launch_sum2(nf_ctx *ctx, int dtype, long N, const void *a, const void *b, void *out, double *partial,
                   double scale) {
  if (dtype == NF_DTYPE_F32)
    hipLaunchKernelGGL(k_sum2<float>, dim3(nblk(N, EW_BLOCK)), dim3(EW_BLOCK), 0, ctx->stream, N, (const float *)a,
                       (const float *)b, (float *)out, partial, scale);
  else
    hipLaunchKernelGGL(k_sum2<double>, dim3(nblk(N, EW_BLOCK)), dim3(EW_BLOCK), 0, ctx->stream, N, (const double *)a,
                       (const double *)b, (double *)out, partial, scale);
  return (int)hipGetLastError();
}

int nf_launch_finish_sum(nf_ctx *ctx, const double *partial, long n, int take_sqrt, double *dst_d, float *dst_f,
                         double *dst_d2, unsigned *bump) {
  hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(EW_BLOCK), 0, ctx->stream, partial, n, take_sqrt, dst_d, dst_f, dst_d2, bump);
  return (int)hipGetLastError();
}

int nf_launch_reduce_slabs(nf_ctx *ctx, int dtype, const void *slab, int nslab, long P, void *g) {
  ProfScope ps(ctx, "reduce_slabs");
  if (nslab >= 64 && P < (long)ctx->num_cu * 1024) {  // one thread per parameter would leave most of the chip idle
    const unsigned grid = (unsigned)nblk(P, RS_PW);
    if (dtype == NF_DTYPE_F32)
      hipLaunchKernelGGL(k_reduce_slabs_split<float>, dim3(grid), dim3(RS_PW * RS_PARTS), 0, ctx->stream, (const float *)slab, nslab, P,
                         (float *)g);
    else
      hipLaunchKernelGGL(k_reduce_slabs_split<double>, dim3(grid), dim3(RS_PW * RS_PARTS), 0, ctx->stream, (const double *)slab, nslab,
                         P, (double *)g);
    return (int)hipGetLastError();
  }
  if (dtype == NF_DTYPE_F32)
    hipLaunchKernelGGL(k_reduce_slabs<float>, dim3(nblk(P, EW_BLOCK)), dim3(EW_BLOCK), 0, ctx->stream,
                       (const float *)slab, nslab, P, (float *)g);
  else
    hipLaunchKernelGGL(k_reduce_slabs<double>, dim3(nblk(P, EW_BLOCK)), dim3(EW_BLOCK), 0, ctx->stream,
                       (const double *)slab, nslab, P, (double *)g);
  return (int)hipGetLastError();
}

long nf_adam_nblocks(long P) { return nblk(P, EW_BLOCK); }

int nf_launch_adam(nf_ctx *ctx, int dtype, void *theta, const void *g, void *m, void *v, long P, double lr, double b1,
                   double b2, double eps, long t, double *partial) {
  ProfScope ps(ctx, "adam");
  const double c1 = 1.0 - pow(b1, (double)t), c2 = 1.0 - pow(b2, (double)t);
  if (dtype == NF_DTYPE_F32)
    hipLaunchKernelGGL(k_adam<float>, dim3(nblk(P, EW_BLOCK)), dim3(EW_BLOCK), 0, ctx->stream, (float *)theta,
                       (const float *)g, (float *)m, (float *)v, P, (float)lr, (float)b1, (float)b2, (float)eps,
                       (float)c1, (float)c2, partial);
  else
    hipLaunchKernelGGL(k_adam<double>, dim3(nblk(P, EW_BLOCK)), dim3(EW_BLOCK), 0, ctx->stream, (double *)theta,
                       (const double *)g, (double *)m, (double *)v, P, lr, b1, b2, eps, c1, c2, partial);
  return (int)hipGetLastError();
}

int nf_launch_sgd(nf_ctx *ctx, int dtype, void *theta, const void *g, void *vel, long P, double lr, double rho,
                  double *partial) {
  ProfScope ps(ctx, "adam");
  if (dtype == NF_DTYPE_F32)
    hipLaunchKernelGGL(k_sgd<float>, dim3(nblk(P, EW_BLOCK)), dim3(EW_BLOCK), 0, ctx->stream, (float *)theta,
                       (const float *)g, (float *)vel, P, (float)lr, (float)rho, partial);
  else
    hipLaunchKernelGGL(k_sgd<double>, dim3(nblk(P, EW_BLOCK)), dim3(EW_BLOCK), 0, ctx->stream, (double *)theta,
                       (const double *)g, (double *)vel, P, lr, rho, partial);
  return (int)hipGetLastError();
}

int nf_launch_fill(nf_ctx *ctx, int dtype, void *p, long n, double v) {
  if (n <= 0) return NF_OK;
  if (dtype == NF_DTYPE_F32)
    hipLaunchKernelGGL(k_fill<float>, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, (float *)p, n, (float)v);
  else
    hipLaunchKernelGGL(k_fill<double>, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, (double *)p, n, v);
  return (int)hipGetLastError();
}

// ---- tiled-layout launchers (RealNVP / NSF internal path, fp32) ---------------------------
static inline long ntiles32(long N) { return (N + TL - 1) / TL; }

int nf_launch_base_sample_tiled(nf_ctx *ctx, int d, long N, uint64_t seed, uint64_t off, uint32_t stream, float *xt,
                                float *logq) {
  if (N <= 0) return NF_OK;
  ProfScope ps(ctx, "base_sample");
  hipLaunchKernelGGL(k_base_sample_tiled, dim3((unsigned)ntiles32(N)), dim3(EW_BLOCK), 0, ctx->stream, d, N,
                     (uint32_t)seed, (uint32_t)(seed >> 32), off, stream, xt, logq);
  return (int)hipGetLastError();
}

int nf_launch_base_logpdf_tiled(nf_ctx *ctx, int d, long N, const float *xt, float *logq) {
  if (N <= 0) return NF_OK;
  hipLaunchKernelGGL(k_base_logpdf_tiled, dim3((unsigned)ntiles32(N)), dim3(EW_BLOCK), 0, ctx->stream, d, N, xt, logq);
  return (int)hipGetLastError();
}

long nf_target_tiled_nblocks(long N) { return ntiles32(N); }

int nf_launch_target_tiled(nf_ctx *ctx, const nf_target *t, int d, long N, const float *yt, const float *logq,
                           const float *ladj, float *gt, double gscale, float *elbos_out, double *partial,
                           double pscale) {
  if (N <= 0) return NF_OK;
  NF_TRY(nf_target_check(t, d));
  ProfScope ps(ctx, "target");
  hipLaunchKernelGGL(k_target_tiled, dim3((unsigned)nf_target_tiled_nblocks(N)), dim3(EW_BLOCK), 0, ctx->stream,
                     t->kind, d, N, yt, (const float *)t->p0, (const float *)t->p1, (float)t->s0, (float)t->s1, logq,
                     ladj, gt, (float)gscale, elbos_out, partial, pscale);
  return (int)hipGetLastError();
}

int nf_launch_layout_convert(nf_ctx *ctx, int d, long N, const float *src, float *dst, int to_tiled) {
  if (N <= 0) return NF_OK;
  ProfScope ps(ctx, "layout_convert");
  hipLaunchKernelGGL(k_layout_convert, dim3((unsigned)ntiles32(N)), dim3(EW_BLOCK), (size_t)TL * (d + 1) * sizeof(float),
                     ctx->stream, d, N, src, dst, to_tiled);
  return (int)hipGetLastError();
}
