// nf_generic64.hip -- general AffineCoupling (RealNVP) and NeuralSplineCoupling kernels, Float64
// and Float32.
//
// The reference runs its coupling-flow tests in Float32 AND Float64 (test/flow.jl:7,72), keeps
// eltype in = eltype out (test/flow.jl:20-21) and accepts any `hdims` vector.  The MFMA kernels of
// this library are fp32 with two hidden layers and bounded widths; this translation unit covers the
// rest -- `paramtype = Float64`, 1/3/4 hidden layers, other K -- with a plain kernel family: one
// thread per sample, the conditioner MLP as scalar FMA loops straight over theta
// (Optimisers.destructure order: Dense weight out x in column-major, i.e. W[i * nout + j]), the
// spline in the flow's element type.  It is a correctness path for the sizes such flows are used at
// (tests, small problems) -- not tuned, and never on the benchmark path.  Parameter gradients are
// DETERMINISTIC since round 3: a workgroup is one wavefront, every dW element is summed over the wave's 64 samples by a
// fixed shuffle tree and written (lane 0) to the workgroup's own slab; k_reduce_slabs adds the slabs in order.  (Rounds
// 1-2 issued one atomicAdd per parameter per sample: 60 ms per coupling at NSF d=32 / hidden 64 / 131 072 samples, and a
// summation order that changed from run to run.)
//
// Reference arithmetic: src/flows/realnvp.jl:57-110, src/flows/neuralspline.jl:65-140,
// src/flows/utils.jl:71-100; MonotonicSplines 0.3.3 as restated in oracle/nf_oracle.py.
#include <cstdlib>

#include "nf_common.h"

// Two size classes (per-thread scratch arrays are sized at compile time): SMALL keeps the shapes such flows are
// normally used at cheap, LARGE takes the general kernels to the d <= 256 / hidden <= 256 envelope of the MFMA paths
// (Float64 RealNVP at the cfg 4 geometry, test/flow.jl's Float64 runs at any width the fp32 kernels take).
struct G64Small { static constexpr int MAXH = 128, MAXO = 512, MAXC = 64; };    // widest hidden / output layer, dims per coupling
struct G64Large { static constexpr int MAXH = 256, MAXO = 1024, MAXC = 128; };
#define G64_MAXK 16
#define G64_BLOCK 64

struct G64Net {
  long w[NF_MAX_HIDDEN + 1], b[NF_MAX_HIDDEN + 1];  // theta offsets per Dense layer
  int dims[NF_MAX_HIDDEN + 2];                      // nin, hidden..., nout
  int nl;                                           // number of Dense layers
};

struct G64Args {
  G64Net net[2];  // RealNVP: s, t; NSF: net[0] only
  int kind, d, c, m, par_t, K;
  double B;
  long N;
};

template <class T>
__device__ __forceinline__ T g64_lrelu(T z) { return z > (T)0.0 ? z : (T)0.01 * z; }

// forward through one MLP; hidden post-activations are kept in acts[layer][.]
template <class T, class SZ>
__device__ void g64_net_fwd(const T *__restrict__ th, const G64Net &n, const T *in,
                            T (*acts)[SZ::MAXH], T *out) {
  const T *cur = in;
  for (int l = 0; l < n.nl; ++l) {
    const int nin = n.dims[l], nout = n.dims[l + 1];
    const T *W = th + n.w[l], *b = th + n.b[l];
    T *dst = (l < n.nl - 1) ? acts[l] : out;
    // sixteen outputs at a time: an input (a per-thread array: scratch memory) is fetched once per sixteen FMAs instead of once
    // per FMA; every output is still the sum over i in ascending order, bit for bit what the one-at-a-time loop gave
    constexpr int JB = 16;
    for (int j0 = 0; j0 < nout; j0 += JB) {
      T acc[JB];
#pragma unroll
      for (int r = 0; r < JB; ++r) acc[r] = b[j0 + r < nout ? j0 + r : nout - 1];
      for (int i = 0; i < nin; ++i) {
        const T ci = cur[i];
        const T *wr = W + (long)i * nout;
#pragma unroll
        for (int r = 0; r < JB; ++r) acc[r] += wr[j0 + r < nout ? j0 + r : nout - 1] * ci;
      }
#pragma unroll
      for (int r = 0; r < JB; ++r)
        if (j0 + r < nout) dst[j0 + r] = (l < n.nl - 1) ? g64_lrelu(acc[r]) : acc[r];
    }
    cur = dst;
  }
}

// sum over the 64 lanes (= samples) of the wavefront, the same tree on every run
template <class T>
__device__ __forceinline__ T g64_wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// reverse pass of g64_net_fwd: delta (cotangent of the output, overwritten) -> din; parameter gradients: the wave's
// sum over its samples goes to slab[theta index - slab_off] (this workgroup's slab; `first`: overwrite, else add --
// a workgroup that walks several sample tiles accumulates in its slab).  Must be called by all 64 lanes (lanes
// without a sample carry delta = 0).
template <class T, class SZ>
__device__ void g64_net_bwd(const T *__restrict__ th, const G64Net &n, const T *in,
                            T (*acts)[SZ::MAXH], T *delta, T *din, T *__restrict__ slab, long slab_off, bool first) {
  T tmp[SZ::MAXH];
  const bool writer = (threadIdx.x & 63) == 0;
  for (int l = n.nl - 1; l >= 0; --l) {
    const int nin = n.dims[l], nout = n.dims[l + 1];
    const T *W = th + n.w[l];
    const T *prev = (l == 0) ? in : acts[l - 1];
    for (int j = 0; j < nout; ++j) {
      const T v = g64_wave_sum(delta[j]);
      if (writer) {
        T *p = slab + (n.b[l] + j - slab_off);
        *p = first ? v : *p + v;
      }
    }
    constexpr bool F64_MFMA = sizeof(T) == 8 && G64_BLOCK == 64;
    if constexpr (F64_MFMA) {
      // Float64 (round 4): the weight gradient of the layer, dW[i][j] = sum over the wave's 64 samples of prev_s[i] delta_s[j], is
      // a GEMM with K = samples.  One wave sum per element (12 cross-lane steps each) was nine tenths of the Float64 step;
      // here the samples' rows go through LDS -- prev in chunks of 32 inputs, delta in chunks of 16 outputs (25 KB: six waves
      // per CU; with 64-input chunks it was 42 KB and three, and the scratch-bound GEMVs around this ran with nothing to hide
      // their latency behind) -- and a 16 x 16
      // block of dW is 16 v_mfma_f64_16x16x4_f64 (lane l supplies A[l & 15][l >> 4], B[l >> 4][l & 15] and receives
      // D[(l >> 4) + 4 r][l & 15], r = 0 .. 3: tools/probe/mfma_f64_probe.hip).  The sum over samples is the instruction's
      // (a fixed order); every lane writes its four elements of the block.
      typedef double f64x4 __attribute__((ext_vector_type(4)));
      constexpr int PC = 16;  // inputs per staged chunk
      __shared__ double pbuf[64 * (PC + 1)], dbuf[64 * 17];
      const int lane = threadIdx.x & 63, c16 = lane & 15, kq = lane >> 4;
      for (int i0 = 0; i0 < nin; i0 += PC) {
        __syncthreads();
        for (int ii = 0; ii < PC; ++ii) pbuf[lane * (PC + 1) + ii] = i0 + ii < nin ? (double)prev[i0 + ii] : 0.0;
        for (int j0 = 0; j0 < nout; j0 += 16) {
          __syncthreads();
          for (int jj = 0; jj < 16; ++jj) dbuf[lane * 17 + jj] = j0 + jj < nout ? (double)delta[j0 + jj] : 0.0;
          __syncthreads();
          for (int ib = 0; ib < PC / 16 && i0 + 16 * ib < nin; ++ib) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pbuf[(4 * ks + kq) * (PC + 1) + 16 * ib + c16], dbuf[(4 * ks + kq) * 17 + c16], acc, 0, 0, 0);
            const int j = j0 + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = i0 + 16 * ib + kq + 4 * r;
              if (i < nin && j < nout) {
                T *p = slab + (n.w[l] + (long)i * nout + j - slab_off);
                *p = first ? (T)acc[r] : *p + (T)acc[r];
              }
            }
          }
        }
      }
      __syncthreads();
    }
    if constexpr (F64_MFMA) {  // the dX part alone, eight inputs at a time (one scratch fetch of delta[j] per eight FMAs)
      constexpr int IB8 = 8;
      for (int i0 = 0; i0 < nin; i0 += IB8) {
        T acc[IB8];
#pragma unroll
        for (int r = 0; r < IB8; ++r) acc[r] = (T)0.0;
        for (int j0 = 0; j0 < nout; j0 += IB8) {  // 8 x 8 blocks: eight contiguous weights per row and fetch, j ascending per sum
          T dj[IB8];
#pragma unroll
          for (int q = 0; q < IB8; ++q) dj[q] = j0 + q < nout ? delta[j0 + q] : (T)0.0;
#pragma unroll
          for (int r = 0; r < IB8; ++r) {
            const T *wr = W + (long)(i0 + r < nin ? i0 + r : nin - 1) * nout;
#pragma unroll
            for (int q = 0; q < IB8; ++q) acc[r] += wr[j0 + q < nout ? j0 + q : nout - 1] * dj[q];
          }
        }
#pragma unroll
        for (int r = 0; r < IB8; ++r) {
          const int i = i0 + r;
          if (i < nin) {
            if (l == 0) din[i] = acc[r];
            else tmp[i] = acc[r] * (acts[l - 1][i] > (T)0.0 ? (T)1.0 : (T)0.01);  // leaky-ReLU' from the post-activation sign
          }
        }
      }
    } else {
    for (int i = 0; i < nin; ++i) {
      T s = (T)0.0;
      for (int j = 0; j < nout; ++j) {
        s += W[(long)i * nout + j] * delta[j];
        const T v = g64_wave_sum(prev[i] * delta[j]);
        if (writer) {
          T *p = slab + (n.w[l] + (long)i * nout + j - slab_off);
          *p = first ? v : *p + v;
        }
      }
      if (l == 0) din[i] = s;
      else tmp[i] = s * (acts[l - 1][i] > (T)0.0 ? (T)1.0 : (T)0.01);  // leaky-ReLU' from the post-activation sign
    }
    }
    if (l > 0)
      for (int i = 0; i < nin; ++i) delta[i] = tmp[i];
  }
}

// ---- rational-quadratic spline, one (dim, sample) at a time, T ----------------------------
template <class T>
struct G64Spline {
  T pX[G64_MAXK + 1], pY[G64_MAXK + 1], dd[G64_MAXK + 1], smw[G64_MAXK], smh[G64_MAXK];
};
template <class T>
__device__ __forceinline__ T g64_softplus(T x) { return log1p(exp(-fabs(x))) + fmax(x, (T)0.0); }
template <class T>
__device__ __forceinline__ T g64_sigmoid(T x) {
  const T e = exp(-fabs(x));
  return x >= (T)0.0 ? (T)1.0 / ((T)1.0 + e) : e / ((T)1.0 + e);
}
template <class T>
__device__ void g64_knots(const T *v, int K, T B, T *sm, T *p) {
  T mx = v[0];
  for (int k = 1; k < K; ++k) mx = fmax(mx, v[k]);
  T sum = (T)0.0;
  for (int k = 0; k < K; ++k) { sm[k] = exp(v[k] - mx); sum += sm[k]; }
  T cs = (T)0.0;
  p[0] = -B;
  for (int k = 0; k < K; ++k) { sm[k] /= sum; cs += sm[k]; p[k + 1] = -B + (T)2.0 * B * cs; }
}
template <class T>
__device__ void g64_build(const T *raw, int K, T B, G64Spline<T> &sp) {
  g64_knots(raw, K, B, sp.smw, sp.pX);
  g64_knots(raw + K, K, B, sp.smh, sp.pY);
  sp.dd[0] = (T)1.0;
  sp.dd[K] = (T)1.0;
  for (int k = 1; k < K; ++k) sp.dd[k] = g64_softplus(raw[2 * K + k - 1]);
}
template <class T>
__device__ __forceinline__ int g64_bin(const T *p, int K, T v, bool &inside) {
  inside = (v >= p[0]) && (v < p[K]);
  int k = 0;
  for (int j = 1; j < K; ++j) k += (v >= p[j]) ? 1 : 0;
  return k;
}
template <class T>
__device__ __forceinline__ T g64_logderiv(T s, T d0, T d1, T xi) {
  const T om = (T)1.0 - xi, den = s + (d1 + d0 - (T)2.0 * s) * xi * om;
  return (T)2.0 * log(s) + log(d1 * xi * xi + (T)2.0 * s * xi * om + d0 * om * om) - (T)2.0 * log(den);
}
template <class T>
__device__ T g64_spline_fwd(const G64Spline<T> &sp, int K, T x, T &logd) {
  bool inside;
  const int k = g64_bin(sp.pX, K, x, inside);
  if (!inside) return x;
  const T dx = sp.pX[k + 1] - sp.pX[k], dy = sp.pY[k + 1] - sp.pY[k], d0 = sp.dd[k], d1 = sp.dd[k + 1];
  const T s = dy / dx, xi = (x - sp.pX[k]) / dx, om = (T)1.0 - xi;
  const T den = s + (d1 + d0 - (T)2.0 * s) * xi * om;
  logd += g64_logderiv(s, d0, d1, xi);
  return sp.pY[k] + dy * (s * xi * xi + d0 * xi * om) / den;
}
template <class T>
__device__ T g64_spline_inv(const G64Spline<T> &sp, int K, T y, T &logd) {
  bool inside;
  const int k = g64_bin(sp.pY, K, y, inside);
  if (!inside) return y;
  const T dx = sp.pX[k + 1] - sp.pX[k], dy = sp.pY[k + 1] - sp.pY[k], d0 = sp.dd[k], d1 = sp.dd[k + 1];
  const T s = dy / dx, yy = y - sp.pY[k], q = d1 + d0 - (T)2.0 * s;
  const T a = dy * (s - d0) + yy * q, bb = dy * d0 - yy * q, c = -s * yy;
  const T disc = fmax(bb * bb - (T)4.0 * a * c, (T)0.0);
  const T xi = (T)2.0 * c / (-bb - sqrt(disc));
  logd -= g64_logderiv(s, d0, d1, xi);
  return xi * dx + sp.pX[k];
}
// reverse pass of g64_spline_fwd at x: (ybar, lbar) -> xbar, thbar[3K-1].
// inv: reverse pass of the INVERSE spline v -> x = S^-1(v), ladj_inv = -log S'(x), at its output x, with
// (ybar, lbar) the cotangents of (x, ladj_inv); implicit-function form: vbar = (ybar - lbar dlogS'/dx) / S',
// thbar = the forward reverse pass with cotangents (-vbar, -lbar).  Returns vbar.
template <class T>
__device__ T g64_spline_bwd(const G64Spline<T> &sp, const T *raw, int K, T B, T x, T ybar,
                                 T lbar, T *thbar, bool inv = false) {
  const int P = 3 * K - 1;
  for (int i = 0; i < P; ++i) thbar[i] = (T)0.0;
  bool inside;
  const int k = g64_bin(sp.pX, K, x, inside);
  if (!inside) return ybar;
  const T dx = sp.pX[k + 1] - sp.pX[k], dy = sp.pY[k + 1] - sp.pY[k], d0 = sp.dd[k], d1 = sp.dd[k + 1];
  const T s = dy / dx, xi = (x - sp.pX[k]) / dx, om = (T)1.0 - xi, q = d1 + d0 - (T)2.0 * s;
  const T den = s + q * xi * om, num = s * xi * xi + d0 * xi * om;
  const T nd = d1 * xi * xi + (T)2.0 * s * xi * om + d0 * om * om;
  const T dnum_dxi = (T)2.0 * s * xi + d0 * ((T)1.0 - (T)2.0 * xi), dden_dxi = q * ((T)1.0 - (T)2.0 * xi);
  const T dnd_dxi = (T)2.0 * d1 * xi + (T)2.0 * s * ((T)1.0 - (T)2.0 * xi) - (T)2.0 * d0 * om;
  const T dy_dxi = dy * (dnum_dxi * den - num * dden_dxi) / (den * den);
  const T dL_dxi = dnd_dxi / nd - (T)2.0 * dden_dxi / den;
  T vbar = (T)0.0;
  if (inv) {
    vbar = (ybar - lbar * dL_dxi / dx) / (dy_dxi / dx);
    ybar = -vbar;
    lbar = -lbar;
  }
  const T dden_ds = (T)1.0 - (T)2.0 * xi * om;
  const T dy_ds = dy * (xi * xi * den - num * dden_ds) / (den * den);
  const T dL_ds = (T)2.0 / s + (T)2.0 * xi * om / nd - (T)2.0 * dden_ds / den;
  const T dy_dd0 = dy * (xi * om * den - num * xi * om) / (den * den), dL_dd0 = om * om / nd - (T)2.0 * xi * om / den;
  const T dy_dd1 = dy * (-num * xi * om) / (den * den), dL_dd1 = xi * xi / nd - (T)2.0 * xi * om / den;
  const T xibar = ybar * dy_dxi + lbar * dL_dxi, sbar = ybar * dy_ds + lbar * dL_ds;
  const T d0bar = ybar * dy_dd0 + lbar * dL_dd0, d1bar = ybar * dy_dd1 + lbar * dL_dd1;
  const T dybar = ybar * num / den + sbar / dx;
  const T dxbar = -sbar * s / dx - xibar * xi / dx;
  const T xkbar = -xibar / dx - dxbar, xk1bar = dxbar, ykbar = ybar - dybar, yk1bar = dybar;
  // p[j] = -B + 2B sum_{i<j} sm_i: dL/dsm_i = 2B sum_{j>i} pbar[j], only pbar[k], pbar[k+1] nonzero
  T dotw = (T)0.0, doth = (T)0.0;
  for (int i = 0; i < K; ++i) {
    const T sbw = (T)2.0 * B * ((i < k) ? (xkbar + xk1bar) : ((i == k) ? xk1bar : (T)0.0));
    const T sbh = (T)2.0 * B * ((i < k) ? (ykbar + yk1bar) : ((i == k) ? yk1bar : (T)0.0));
    thbar[i] = sbw;
    thbar[K + i] = sbh;
    dotw += sbw * sp.smw[i];
    doth += sbh * sp.smh[i];
  }
  for (int i = 0; i < K; ++i) {
    thbar[i] = sp.smw[i] * (thbar[i] - dotw);
    thbar[K + i] = sp.smh[i] * (thbar[K + i] - doth);
  }
  if (k >= 1) thbar[2 * K + k - 1] = d0bar * g64_sigmoid(raw[2 * K + k - 1]);
  if (k + 1 <= K - 1) thbar[2 * K + k] = d1bar * g64_sigmoid(raw[2 * K + k]);
  return inv ? vbar : xibar / dx;
}

// ---- one coupling, forward or inverse, standard layout (x[j*d + i]) ---------------------------
template <class T, class SZ>
__global__ __launch_bounds__(G64_BLOCK) void k_g64_apply(G64Args a, int inverse, const T *__restrict__ theta,
                                                        const T *x, T *y, T *__restrict__ ladj) {
  const long j = (long)blockIdx.x * G64_BLOCK + threadIdx.x;
  if (j >= a.N) return;
  const T *xr = x + j * a.d;
  T *yr = y + j * a.d;
  T x2[SZ::MAXC], acts[NF_MAX_HIDDEN][SZ::MAXH], out[SZ::MAXO];
  const int par_c = 1 - a.par_t;
  for (int q = 0; q < a.m; ++q) x2[q] = xr[2 * q + par_c];
  T lsum = (T)0.0;
  if (a.kind == NF_KIND_REALNVP) {
    T s[SZ::MAXC];
    g64_net_fwd<T, SZ>(theta, a.net[0], x2, acts, out);
    for (int p = 0; p < a.c; ++p) s[p] = tanh(out[p]);
    g64_net_fwd<T, SZ>(theta, a.net[1], x2, acts, out);  // out = t
    for (int p = 0; p < a.c; ++p) {
      const T v = xr[2 * p + a.par_t];
      yr[2 * p + a.par_t] = inverse ? (v - out[p]) * exp(-s[p]) : v * exp(s[p]) + out[p];
      lsum += inverse ? -s[p] : s[p];
    }
  } else {
    g64_net_fwd<T, SZ>(theta, a.net[0], x2, acts, out);
    const int P = 3 * a.K - 1;
    G64Spline<T> sp;
    for (int p = 0; p < a.c; ++p) {
      g64_build<T>(out + p * P, a.K, (T)a.B, sp);
      const T v = xr[2 * p + a.par_t];
      yr[2 * p + a.par_t] = inverse ? g64_spline_inv(sp, a.K, v, lsum) : g64_spline_fwd(sp, a.K, v, lsum);
    }
  }
  if (y != x)
    for (int q = 0; q < a.m; ++q) yr[2 * q + par_c] = x2[q];
  ladj[j] += lsum;
}

// reverse pass of one coupling at its INPUT x: gbar holds ybar on entry, xbar on exit.
// inv != 0: reverse pass of the INVERSE coupling at its OUTPUT x (same point): gbar holds the cotangent of
// x on entry and of the inverse's input on exit, lbar is the cotangent of ladj_inv.
// A workgroup (one wavefront) walks sample tiles blockIdx.x, blockIdx.x + gridDim.x, ...; slab: [gridDim.x][Pc] partial
// parameter gradients of THIS coupling (Pc parameters starting at theta index slab_off).
template <class T, class SZ>
__global__ __launch_bounds__(G64_BLOCK) void k_g64_bwd(G64Args a, int inv, const T *__restrict__ theta,
                                                      const T *__restrict__ x, T *gbar,
                                                      const T *__restrict__ lbar, T lbar_const,
                                                      T *__restrict__ slabs, long Pc, long slab_off) {
  static_assert(G64_BLOCK == 64, "one wavefront per workgroup: g64_net_bwd sums over the wave");
  T *slab = slabs + (long)blockIdx.x * Pc;
  const long ntiles = (a.N + G64_BLOCK - 1) / G64_BLOCK;
  bool first = true;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x, first = false) {
  const long j = tile * G64_BLOCK + threadIdx.x;
  const bool valid = j < a.N;  // lanes without a sample run along with zero cotangents (the wave sums need all lanes)
  const long jr = valid ? j : a.N - 1;
  const T *xr = x + jr * a.d;
  T *gr = gbar + jr * a.d;
  const T vm = valid ? (T)1.0 : (T)0.0;
  const T lb = (lbar ? lbar[jr] : lbar_const) * vm;
  auto gld = [&](int idx) { return gr[idx] * vm; };
  auto gst = [&](int idx, T v) { if (valid) gr[idx] = v; };
  T x2[SZ::MAXC], acts[NF_MAX_HIDDEN][SZ::MAXH], out[SZ::MAXO], din[SZ::MAXC];
  const int par_c = 1 - a.par_t;
  for (int q = 0; q < a.m; ++q) x2[q] = xr[2 * q + par_c];
  if (a.kind == NF_KIND_REALNVP && inv) {
    // x1 = (v1 - t) exp(-s), ladj_inv = -sum s:  v1bar = x1bar exp(-s), sbar = -x1bar x1 - lbar, tbar = -v1bar
    T v1b[SZ::MAXC];
    g64_net_fwd<T, SZ>(theta, a.net[0], x2, acts, out);
    for (int p = 0; p < a.c; ++p) {
      const T s = tanh(out[p]), x1 = xr[2 * p + a.par_t], xb = gld(2 * p + a.par_t);
      v1b[p] = xb * exp(-s);
      gst(2 * p + a.par_t, v1b[p]);
      out[p] = (-xb * x1 - lb) * ((T)1.0 - s * s);
    }
    g64_net_bwd<T, SZ>(theta, a.net[0], x2, acts, out, din, slab, slab_off, first);
    T acc2[SZ::MAXC];
    for (int q = 0; q < a.m; ++q) acc2[q] = gld(2 * q + par_c) + din[q];
    g64_net_fwd<T, SZ>(theta, a.net[1], x2, acts, out);
    for (int p = 0; p < a.c; ++p) out[p] = -v1b[p];
    g64_net_bwd<T, SZ>(theta, a.net[1], x2, acts, out, din, slab, slab_off, first);
    for (int q = 0; q < a.m; ++q) gst(2 * q + par_c, acc2[q] + din[q]);
  } else if (a.kind == NF_KIND_REALNVP) {
    // t net: y1 = x1 exp(s) + t  =>  tbar = ybar1
    g64_net_fwd<T, SZ>(theta, a.net[1], x2, acts, out);
    for (int p = 0; p < a.c; ++p) out[p] = gld(2 * p + a.par_t);
    g64_net_bwd<T, SZ>(theta, a.net[1], x2, acts, out, din, slab, slab_off, first);
    T acc2[SZ::MAXC];
    for (int q = 0; q < a.m; ++q) acc2[q] = gld(2 * q + par_c) + din[q];
    // s net: sbar = ybar1 x1 exp(s) + lbar, through tanh
    g64_net_fwd<T, SZ>(theta, a.net[0], x2, acts, out);
    for (int p = 0; p < a.c; ++p) {
      const T s = tanh(out[p]), es = exp(s), x1 = xr[2 * p + a.par_t], yb = gld(2 * p + a.par_t);
      gst(2 * p + a.par_t, yb * es);
      out[p] = (yb * x1 * es + lb) * ((T)1.0 - s * s);
    }
    g64_net_bwd<T, SZ>(theta, a.net[0], x2, acts, out, din, slab, slab_off, first);
    for (int q = 0; q < a.m; ++q) gst(2 * q + par_c, acc2[q] + din[q]);
  } else {
    g64_net_fwd<T, SZ>(theta, a.net[0], x2, acts, out);
    const int P = 3 * a.K - 1;
    G64Spline<T> sp;
    T thb[3 * G64_MAXK];
    for (int p = 0; p < a.c; ++p) {
      g64_build<T>(out + p * P, a.K, (T)a.B, sp);
      const T xb = g64_spline_bwd<T>(sp, out + p * P, a.K, (T)a.B, xr[2 * p + a.par_t], gld(2 * p + a.par_t), lb, thb, inv != 0);
      gst(2 * p + a.par_t, xb);
      for (int i = 0; i < P; ++i) out[p * P + i] = thb[i] * vm;
    }
    g64_net_bwd<T, SZ>(theta, a.net[0], x2, acts, out, din, slab, slab_off, first);
    for (int q = 0; q < a.m; ++q) gst(2 * q + par_c, gld(2 * q + par_c) + din[q]);
  }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// "l64": the conditioner MLP of the general coupling kernels on the matrix pipe (Float32; round 3)
// ------------------------------------------------------------------------------------------------------------------
// The thread-per-sample kernels above spend their time in the MLP (one scalar FMA per weight load: 22.8 ms per coupling
// for the reverse pass at NSF d = 32, hidden [64, 64], K = 8, N = 131 072).  For every Float32 flow of this family's
// envelope (d <= 256, hidden <= 256, depth 1-4, K <= 16) the MLP runs layer by layer on fp32 MFMAs instead, with the
// primitives of the resident kernels (dense_fwd / dense_bwd_x / dw_accumulate, nf_mfma.h): a layer's weight block is staged
// from theta (Dense weight out x in column-major = W[i][o] row-major, no packing) into LDS, a wavefront owns 32-sample
// tiles, activations travel between launches in the tiled layout [tile][row][32 samples] of a scratch buffer (the
// context's image buffer, nf_wimg_reserve).  The coupling arithmetic itself (tanh / exp, the spline in the element type
// with run-time K) stays what it was -- the same device functions, one thread per sample -- reading the net outputs from
// the tiles.  So every entry point that reaches nf_g64_apply / g64_launch_bwd (forward, inverse, rand, ELBO, training
// step, forward-KL, pullbacks, compositions) takes this path for those shapes; Float64 and wider nets keep the scalar MLP.
#include "nf_mfma.h"
#include "nf_rqs_elem.h"  // the fused spline kernels' element functions (K a compile-time 8): k_l64_nsf_top_fwd / _bwd

#define L64_TILE 32
struct L64Layer {
  long w_off, b_off;  // theta offsets of W (nin x nout, row-major) and b
  int nin, nout;      // real sizes
  int o0;             // first output column of this launch's block group
};
// a [rows][samples] operand: tiled scratch (F rows per tile) or, for a first layer, the conditioner half of the state in the
// standard layout (x[j * d + 2 q + par]; d > 0 selects it)
struct L64Src {
  const float *p;
  int F, row0;   // tiled: rows per tile, first row
  int d, par;    // standard layout: feature count and parity of the rows
};
// Tile I/O through buffer descriptors (as the fused kernels' tile_load / tile_store, nf_mfma.h): one descriptor per (operand,
// tile) in scalar registers, one lane offset, and the row of a register as a compile-time scalar offset -- no 64-bit address
// per element (the first form computed one for each of up to 128 loads in flight: 478 registers for a [64 -> 128] block,
// one wavefront per SIMD).  Rows and samples outside the descriptor's extent read as 0 and are dropped on a store, which is
// what the padding needs.
struct L64Io {
  __amdgpu_buffer_rsrc_t rs;
  int voff;
};
// blockIdx.y = net (RealNVP's s and t nets have the same shapes: one launch serves both): theta offsets and the three
// operand pointers of a kernel move by these strides per net (0 for an operand the nets share)
struct L64Y {
  long dtheta, da, db, dc;
};
// the constant part of a register's row: row(b, r, hi) = l64_rc(b, r) + 4 hi
__device__ __forceinline__ constexpr int l64_rc(int b, int r) { return 32 * b + (r & 3) + 8 * (r >> 2); }
// tiled operand [tile][F rows][32 samples], rows from row0 on
__device__ __forceinline__ L64Io l64_io_tiled(const float *p, int F, int row0, long tile, int l31, int hi) {
  L64Io io;
  const int rows = F - row0 > 0 ? F - row0 : 0;
  io.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p) + (tile * F + row0) * L64_TILE, 0, rows * L64_TILE * 4, 0x00020000);
  io.voff = (4 * hi * L64_TILE + l31) * 4;
  return io;
}
// the conditioner half of the standard-layout state, x[j * d + 2 row + par], samples of one tile (those < N)
__device__ __forceinline__ L64Io l64_io_std(const float *p, int d, int par, long tile, long N, int l31, int hi) {
  L64Io io;
  long rem = N - tile * L64_TILE;
  rem = rem < 0 ? 0 : rem > L64_TILE ? L64_TILE : rem;
  io.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p) + tile * L64_TILE * d, 0, (int)rem * d * 4, 0x00020000);
  io.voff = (l31 * d + 8 * hi + par) * 4;
  return io;
}
__device__ __forceinline__ float l64_ld(const L64Io &io, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(io.rs, io.voff, soff, 0));
}
__device__ __forceinline__ void l64_st(const L64Io &io, int soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), io.rs, io.voff, soff, 0);
}
template <int NB>
__device__ __forceinline__ void l64_load(const L64Src &s, long tile, int l31, int hi, long N, int nrows, f32x16 (&v)[NB]) {
  if (s.d > 0) {
    const L64Io io = l64_io_std(s.p, s.d, s.par, tile, N, l31, hi);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = l64_ld(io, l64_rc(b, r) * 8);
        v[b][r] = l64_rc(b, r) + 4 * hi < nrows ? x : 0.f;  // past the half's rows sits the next sample
      }
  } else {
    const L64Io io = l64_io_tiled(s.p, s.F, s.row0, tile, l31, hi);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[b][r] = l64_ld(io, l64_rc(b, r) * (L64_TILE * 4));
  }
}
// stage W[i][o0 + o] (i < 32 IB, o < 32 OB) and b[o0 + o] from theta, zero outside the layer
template <int IB, int OB>
__device__ __forceinline__ void l64_stage(float *__restrict__ w, float *__restrict__ b, const float *__restrict__ theta,
                                          const L64Layer &L, int tid, int nthreads) {
  constexpr int S = 32 * OB + NF_IMG_PAD, NE = 32 * IB * 32 * OB, U = 16;
  // sixteen requests per thread in flight at a time (round 5): one element per iteration exposed a round trip to L2 for each of
  // a block's 16-64 elements per thread -- the same pattern cost k_g64m_apply 39 % of its launch (tools/trace_g64m.py)
#pragma unroll 1
  for (int e0 = tid; e0 < NE; e0 += nthreads * U) {
    float v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int e = e0 + nthreads * k, i = e / (32 * OB), o = e - i * (32 * OB);
      v[k] = (e < NE && i < L.nin && L.o0 + o < L.nout) ? theta[L.w_off + (long)i * L.nout + L.o0 + o] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int e = e0 + nthreads * k, i = e / (32 * OB), o = e - i * (32 * OB);
      if (e < NE) w[i * S + o] = v[k];
    }
  }
  for (int o = tid; o < 32 * OB; o += nthreads) b[o] = (L.o0 + o < L.nout) ? theta[L.b_off + L.o0 + o] : 0.f;
}

// out rows [o0, o0 + 32 OB) of dst <- W' in + b (leaky-ReLU if act)
template <int IB, int OB>
__global__ __launch_bounds__(256) void k_l64_fwd(const float *__restrict__ theta, L64Layer L, L64Src src, float *__restrict__ dst, int Fd,
                                                 long N, int act, L64Y yy) {
  constexpr int S = 32 * OB + NF_IMG_PAD;
  L.w_off += blockIdx.y * yy.dtheta; L.b_off += blockIdx.y * yy.dtheta; src.p += blockIdx.y * yy.da; dst += blockIdx.y * yy.db;
  __shared__ __attribute__((aligned(16))) float w[32 * IB * S + 32 * OB];
  float *b = w + 32 * IB * S;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  l64_stage<IB, OB>(w, b, theta, L, tid, 256);
  __syncthreads();
  const long ntiles = (N + L64_TILE - 1) / L64_TILE;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    f32x16 in[IB], out[OB];
    l64_load<IB>(src, tile, l31, hi, N, L.nin, in);
    dense_fwd<IB, OB, S>(w, b, in, out, l31, hi);
    const L64Io od = l64_io_tiled(dst, Fd, L.o0, tile, l31, hi);
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = out[ob][r];
        l64_st(od, l64_rc(ob, r) * (L64_TILE * 4), act ? nf_lrelu(v) : v);
      }
  }
}

// ALL output rows of a wide layer in one launch (as k_l64_bwdx_all): the NG groups' weight blocks side by side in LDS
// (dynamic), the input tile read once, eight waves per workgroup.
template <int IB, int OB>
__global__ __launch_bounds__(512) void k_l64_fwd_all(const float *__restrict__ theta, L64Layer L, int NG, L64Src src, float *__restrict__ dst,
                                                     int Fd, long N, int act, L64Y yy) {
  constexpr int S = 32 * OB + NF_IMG_PAD, WG = 32 * IB * S + 32 * OB;
  L.w_off += blockIdx.y * yy.dtheta; L.b_off += blockIdx.y * yy.dtheta; src.p += blockIdx.y * yy.da; dst += blockIdx.y * yy.db;
  extern __shared__ __attribute__((aligned(16))) float wdyn[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  for (int q = 0; q < NG; ++q) {
    L64Layer Lq = L;
    Lq.o0 = 32 * OB * q;
    l64_stage<IB, OB>(wdyn + q * WG, wdyn + q * WG + 32 * IB * S, theta, Lq, tid, 512);
  }
  __syncthreads();
  const long ntiles = (N + L64_TILE - 1) / L64_TILE;
  for (long tile = (long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long)gridDim.x * 8) {
    f32x16 in[IB];
    l64_load<IB>(src, tile, l31, hi, N, L.nin, in);
#pragma unroll 1
    for (int q = 0; q < NG; ++q) {
      f32x16 out[OB];
      dense_fwd<IB, OB, S>(wdyn + q * WG, wdyn + q * WG + 32 * IB * S, in, out, l31, hi);
      const L64Io od = l64_io_tiled(dst, Fd, 32 * OB * q, tile, l31, hi);
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = out[ob][r];
          l64_st(od, l64_rc(ob, r) * (L64_TILE * 4), act ? nf_lrelu(v) : v);
        }
    }
  }
}

// Consecutive NARROW layers (every width <= 64) of a net in one launch: all their weight blocks in LDS, a tile's activations
// chained through registers from layer to layer (each layer's output still goes to its tile buffer -- the reverse pass reads
// it), eight waves per workgroup.  (Tried in the first form of these kernels, one wavefront per SIMD and 64-bit addressing:
// no gain; with the buffer-descriptor I/O the launches it saves count.)
#define L64_CHAIN_MAX (NF_MAX_HIDDEN + 1)
struct L64Chain {
  int nl;
  long w_off[L64_CHAIN_MAX], b_off[L64_CHAIN_MAX];
  int nin[L64_CHAIN_MAX], nout[L64_CHAIN_MAX], act[L64_CHAIN_MAX], F[L64_CHAIN_MAX];
  float *dst[L64_CHAIN_MAX];
};
__global__ __launch_bounds__(512) void k_l64_fwd_chain(const float *__restrict__ theta, L64Chain ch, L64Src src, long N, L64Y yy) {
  constexpr int S = 64 + NF_IMG_PAD, WG = 64 * S + 64;
  extern __shared__ __attribute__((aligned(16))) float wdyn[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  src.p += blockIdx.y * yy.da;
  for (int l = 0; l < ch.nl; ++l) {
    const L64Layer L{ch.w_off[l] + (long)blockIdx.y * yy.dtheta, ch.b_off[l] + (long)blockIdx.y * yy.dtheta, ch.nin[l], ch.nout[l], 0};
    l64_stage<2, 2>(wdyn + l * WG, wdyn + l * WG + 64 * S, theta, L, tid, 512);
  }
  __syncthreads();
  const long ntiles = (N + L64_TILE - 1) / L64_TILE;
  for (long tile = (long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long)gridDim.x * 8) {
    f32x16 cur[2];
    l64_load<2>(src, tile, l31, hi, N, ch.nin[0], cur);
#pragma unroll 1
    for (int l = 0; l < ch.nl; ++l) {
      f32x16 out[2];
      dense_fwd<2, 2, S>(wdyn + l * WG, wdyn + l * WG + 64 * S, cur, out, l31, hi);
      const L64Io od = l64_io_tiled(ch.dst[l] + blockIdx.y * yy.db, ch.F[l], 0, tile, l31, hi);
      const int act = ch.act[l];
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = act ? nf_lrelu(out[ob][r]) : out[ob][r];
          cur[ob][r] = v;  // rows past the layer's outputs are 0 (zero weights, zero bias): the next layer's padding
          l64_st(od, l64_rc(ob, r) * (L64_TILE * 4), v);
        }
    }
  }
}

// delta of a layer's output rows: the stored cotangent, times leaky-ReLU' from the sign of the stashed post-activation
template <int OB>
__device__ __forceinline__ void l64_delta(const L64Src &g, const float *__restrict__ act, int Fa, int o0, long tile, int l31, int hi,
                                          long N, f32x16 (&dl)[OB]) {
  const L64Io gi = l64_io_tiled(g.p, g.F, g.row0, tile, l31, hi);
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) dl[ob][r] = l64_ld(gi, l64_rc(ob, r) * (L64_TILE * 4));
  if (act) {
    const L64Io ai = l64_io_tiled(act, Fa, o0, tile, l31, hi);
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) dl[ob][r] *= l64_ld(ai, l64_rc(ob, r) * (L64_TILE * 4)) > 0.f ? 1.f : 0.01f;
  }
}

// the input cotangent of a layer: into the tiled buffer (Fd rows; accumulate: added to what is there) or, with xd > 0, added
// to the conditioner half of the standard-layout cotangent gbar[j * xd + 2 row + par]
template <int IB>
__device__ __forceinline__ void l64_store_din(const f32x16 (&din)[IB], float *__restrict__ dst, int Fd, int accumulate, int xd, int xpar,
                                              int nin, long tile, long N, int l31, int hi) {
  if (xd > 0) {
    const L64Io io = l64_io_std(dst, xd, xpar, tile, N, l31, hi);
    float old[IB][16];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) old[ib][r] = l64_ld(io, l64_rc(ib, r) * 8);
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (l64_rc(ib, r) + 4 * hi < nin) l64_st(io, l64_rc(ib, r) * 8, old[ib][r] + din[ib][r]);
  } else {
    const L64Io io = l64_io_tiled(dst, Fd, 0, tile, l31, hi);
    if (accumulate) {
      float old[IB][16];
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) old[ib][r] = l64_ld(io, l64_rc(ib, r) * (L64_TILE * 4));
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) l64_st(io, l64_rc(ib, r) * (L64_TILE * 4), old[ib][r] + din[ib][r]);
    } else {
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) l64_st(io, l64_rc(ib, r) * (L64_TILE * 4), din[ib][r]);
    }
  }
}

// din (+)= W[:, o0 : o0 + 32 OB] delta.  dst tiled (Fd rows; accumulate: add to what is there) or, with xd > 0, the
// conditioner half of the standard-layout cotangent gbar[j * xd + 2 q + par] (always accumulated: x2bar += din)
template <int IB, int OB>
__global__ __launch_bounds__(256) void k_l64_bwdx(const float *__restrict__ theta, L64Layer L, L64Src g, const float *__restrict__ act,
                                                  int Fa, float *__restrict__ dst, int Fd, int accumulate, int xd, int xpar, long N, L64Y yy) {
  constexpr int S = 32 * OB + NF_IMG_PAD;
  L.w_off += blockIdx.y * yy.dtheta; L.b_off += blockIdx.y * yy.dtheta; g.p += blockIdx.y * yy.da; dst += blockIdx.y * yy.dc;
  if (act) act += blockIdx.y * yy.db;
  __shared__ __attribute__((aligned(16))) float w[32 * IB * S + 32 * OB];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  l64_stage<IB, OB>(w, w + 32 * IB * S, theta, L, tid, 256);
  __syncthreads();
  const long ntiles = (N + L64_TILE - 1) / L64_TILE;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    f32x16 dl[OB], din[IB];
    l64_delta<OB>(g, act, Fa, L.o0, tile, l31, hi, N, dl);
    dense_bwd_x<IB, OB, S>(w, dl, din, l31, hi);
    l64_store_din<IB>(din, dst, Fd, accumulate, xd, xpar, L.nin, tile, N, l31, hi);
  }
}

// The same product over ALL output rows of the layer in one launch: the NG groups' weight blocks sit side by side in LDS
// (dynamic; the caller checks the size), a tile's input cotangent is accumulated over the groups in registers and stored
// once -- a wide layer (the spline's 3K - 1 parameters per dimension: 12 blocks at the docstring shape) otherwise re-reads
// and re-writes the 64-row destination once per group.  Eight waves per workgroup share the staged weights.
template <int IB, int OB>
__global__ __launch_bounds__(512) void k_l64_bwdx_all(const float *__restrict__ theta, L64Layer L, int NG, L64Src g, const float *__restrict__ act,
                                                      int Fa, float *__restrict__ dst, int Fd, int xd, int xpar, long N, L64Y yy) {
  constexpr int S = 32 * OB + NF_IMG_PAD, WG = 32 * IB * S + 32 * OB;
  L.w_off += blockIdx.y * yy.dtheta; L.b_off += blockIdx.y * yy.dtheta; g.p += blockIdx.y * yy.da; dst += blockIdx.y * yy.dc;
  if (act) act += blockIdx.y * yy.db;
  extern __shared__ __attribute__((aligned(16))) float wdyn[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  for (int q = 0; q < NG; ++q) {
    L64Layer Lq = L;
    Lq.o0 = 32 * OB * q;
    l64_stage<IB, OB>(wdyn + q * WG, wdyn + q * WG + 32 * IB * S, theta, Lq, tid, 512);
  }
  __syncthreads();
  const long ntiles = (N + L64_TILE - 1) / L64_TILE;
  for (long tile = (long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long)gridDim.x * 8) {
    f32x16 dl[OB], din[IB];
    {
      L64Src gq = g;
      l64_delta<OB>(gq, act, Fa, 0, tile, l31, hi, N, dl);
      dense_bwd_x<IB, OB, S, false>(wdyn, dl, din, l31, hi);
    }
#pragma unroll 1
    for (int q = 1; q < NG; ++q) {
      L64Src gq = g;
      gq.row0 = g.row0 + 32 * OB * q;
      l64_delta<OB>(gq, act, Fa, 32 * OB * q, tile, l31, hi, N, dl);
      dense_bwd_x<IB, OB, S, true>(wdyn + q * WG, dl, din, l31, hi);
    }
    l64_store_din<IB>(din, dst, Fd, 0, xd, xpar, L.nin, tile, N, l31, hi);
  }
}

// The input cotangents of consecutive NARROW layers (widths <= 64), top down, in one launch: layer q's delta from the cotangent
// in registers and the sign of its stashed outputs, din = W' delta, stored (the dW kernel of the layer below reads it) and
// carried on as the next layer's cotangent.  Entry 0 of the chain is the highest layer.
struct L64BChain {
  int nl;
  long w_off[L64_CHAIN_MAX];
  int nin[L64_CHAIN_MAX], nout[L64_CHAIN_MAX], Fa[L64_CHAIN_MAX];
  const float *act[L64_CHAIN_MAX];  // the layer's stashed outputs (null: the net's output layer, no activation)
  float *dst[L64_CHAIN_MAX];        // cotangent of the layer's inputs (Fd rows)
};
__global__ __launch_bounds__(512) void k_l64_bwdx_chain(const float *__restrict__ theta, L64BChain ch, L64Src gtop, int Fd, long N, L64Y yy) {
  constexpr int S = 64 + NF_IMG_PAD, WG = 64 * S + 64;
  extern __shared__ __attribute__((aligned(16))) float wdyn[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  gtop.p += blockIdx.y * yy.da;
  for (int q = 0; q < ch.nl; ++q) {
    const L64Layer L{ch.w_off[q] + (long)blockIdx.y * yy.dtheta, 0, ch.nin[q], ch.nout[q], 0};
    l64_stage<2, 2>(wdyn + q * WG, wdyn + q * WG + 64 * S, theta, L, tid, 512);  // (the bias row is staged and not used)
  }
  __syncthreads();
  const long ntiles = (N + L64_TILE - 1) / L64_TILE;
  for (long tile = (long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long)gridDim.x * 8) {
    f32x16 g[2];
    l64_load<2>(gtop, tile, l31, hi, N, 64, g);
#pragma unroll 1
    for (int q = 0; q < ch.nl; ++q) {
      if (ch.act[q]) {
        const L64Io ai = l64_io_tiled(ch.act[q] + blockIdx.y * yy.db, ch.Fa[q], 0, tile, l31, hi);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
          for (int r = 0; r < 16; ++r) g[ob][r] *= l64_ld(ai, l64_rc(ob, r) * (L64_TILE * 4)) > 0.f ? 1.f : 0.01f;
      }
      f32x16 din[2];
      dense_bwd_x<2, 2, S, false>(wdyn + q * WG, g, din, l31, hi);
      const L64Io od = l64_io_tiled(ch.dst[q] + blockIdx.y * yy.dc, Fd, 0, tile, l31, hi);
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          l64_st(od, l64_rc(ib, r) * (L64_TILE * 4), din[ib][r]);
          g[ib][r] = din[ib][r];
        }
    }
  }
}

// this workgroup's partial of dW[:, o0 : o0 + 32 OB] = sum_j a_j delta_j' and db, written in theta order into its slab
template <int IB, int OB>
__global__ __launch_bounds__(256) void k_l64_dw(L64Layer L, L64Src a, L64Src g, const float *__restrict__ act, int Fa, long N,
                                                float *__restrict__ slabs, long Pc, long slab_off, L64Y yy) {
  constexpr int SA = IB * 32 * NF_TS, SD = OB * 32 * NF_TS;
  L.w_off += blockIdx.y * yy.dtheta; L.b_off += blockIdx.y * yy.dtheta; a.p += blockIdx.y * yy.da; g.p += blockIdx.y * yy.db;
  if (act) act += blockIdx.y * yy.dc;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // 4 (SA + SD) floats (beyond the 64 KB static limit at IB = OB = 2)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  float *sa = sm + wave * (SA + SD), *sd = sa + SA;
  f32x16 acc[IB][OB];
  float bsum[OB];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) {
    bsum[ob] = 0.f;
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ib][ob][r] = 0.f;
  }
  const long ntiles = (N + L64_TILE - 1) / L64_TILE;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    f32x16 av[IB], dl[OB];
    l64_load<IB>(a, tile, l31, hi, N, L.nin, av);
    l64_delta<OB>(g, act, Fa, L.o0, tile, l31, hi, N, dl);
    tile_to_scratch<IB>(sa, av, l31, hi);
    tile_to_scratch<OB>(sd, dl, l31, hi);
    wave_lds_fence();
    dw_accumulate<IB, OB>(sa, sd, acc, bsum, l31, hi);
    wave_lds_fence();
  }
  // waves in a fixed order through one [32 IB][32 OB] image (+ bias row) in LDS, then the slab in theta order
  __syncthreads();
  float *img = sm;  // stride 32 OB; the per-wave tiles are dead
  constexpr int SI = 32 * OB;
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float *p = img + (32 * ib + nf_row(r, hi)) * SI + 32 * ob + l31;
            *p = wv == 0 ? acc[ib][ob][r] : *p + acc[ib][ob][r];
          }
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) {
        const float v = bsum[ob] + __shfl_xor(bsum[ob], 32);
        if (hi == 0) {
          float *p = img + 32 * IB * SI + 32 * ob + l31;
          *p = wv == 0 ? v : *p + v;
        }
      }
    }
    __syncthreads();
  }
  float *slab = slabs + (long)blockIdx.x * Pc - slab_off;
  for (int e = tid; e < 32 * IB * SI; e += 256) {
    const int i = e / SI, o = e - i * SI;
    if (i < L.nin && L.o0 + o < L.nout) slab[L.w_off + (long)i * L.nout + L.o0 + o] = img[e];
  }
  for (int o = tid; o < SI; o += 256)
    if (L.o0 + o < L.nout) slab[L.b_off + L.o0 + o] = img[32 * IB * SI + o];
}

// dW of a WIDE layer in one launch: wave w of every workgroup owns the output columns [32 OB w, 32 OB (w + 1)) and all four
// waves walk the workgroup's tiles together, so the activation operand leaves HBM once per tile and every delta once in
// total (k_l64_dw over column groups re-reads the activations per group: six launches for the spline layer's 12 blocks).
// Each wave accumulates its own [32 IB][32 OB] block and writes it from its registers -- columns are disjoint, there is no
// cross-wave fold.  Needs 4 OB >= the layer's blocks.
template <int IB, int OB>
__global__ __launch_bounds__(256) void k_l64_dw_cols(L64Layer L, L64Src a, L64Src g, const float *__restrict__ act, int Fa, long N,
                                                     float *__restrict__ slabs, long Pc, long slab_off, L64Y yy) {
  constexpr int SA = IB * 32 * NF_TS, SD = OB * 32 * NF_TS;
  L.w_off += blockIdx.y * yy.dtheta; L.b_off += blockIdx.y * yy.dtheta; a.p += blockIdx.y * yy.da; g.p += blockIdx.y * yy.db;
  if (act) act += blockIdx.y * yy.dc;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // SA + 4 SD floats: ONE activation tile, a delta tile per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  float *sa = sm, *sd = sm + SA + wave * SD;
  const int o0w = 32 * OB * wave;
  f32x16 acc[IB][OB];
  float bsum[OB];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) {
    bsum[ob] = 0.f;
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ib][ob][r] = 0.f;
  }
  L64Src gw = g;
  gw.row0 = g.row0 + o0w;
  const long ntiles = (N + L64_TILE - 1) / L64_TILE;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // the activation tile is shared: wave w fetches and places registers 4w .. 4w+3 of every block (rows 8w + {0..3} + 4 hi)
    float aq[IB][4];
    if (a.d > 0) {
      const L64Io io = l64_io_std(a.p, a.d, a.par, tile, N, l31, hi);
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int rc = 32 * ib + 8 * wave + e;
          const float x = l64_ld(io, rc * 8);
          aq[ib][e] = rc + 4 * hi < L.nin ? x : 0.f;
        }
    } else {
      const L64Io io = l64_io_tiled(a.p, a.F, a.row0, tile, l31, hi);
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int e = 0; e < 4; ++e) aq[ib][e] = l64_ld(io, (32 * ib + 8 * wave + e) * (L64_TILE * 4));
    }
    f32x16 dl[OB];
    l64_delta<OB>(gw, act, Fa, L.o0 + o0w, tile, l31, hi, N, dl);
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int e = 0; e < 4; ++e) sa[(32 * ib + 8 * wave + e + 4 * hi) * NF_TS + l31] = aq[ib][e];
    tile_to_scratch<OB>(sd, dl, l31, hi);
    __syncthreads();
    dw_accumulate<IB, OB>(sa, sd, acc, bsum, l31, hi);
    __syncthreads();  // every wave is done with the shared tile before the next one lands
  }
  float *slab = slabs + (long)blockIdx.x * Pc - slab_off;
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) {
    const int o = L.o0 + o0w + 32 * ob + l31;
    if (o < L.nout) {
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = 32 * ib + nf_row(r, hi);
          if (i < L.nin) slab[L.w_off + (long)i * L.nout + o] = acc[ib][ob][r];
        }
    }
    const float v = bsum[ob] + __shfl_xor(bsum[ob], 32);
    if (hi == 0 && o < L.nout) slab[L.b_off + o] = v;
  }
}

// ---- the coupling arithmetic on net outputs held in tiles (one thread per sample; as k_g64_apply / k_g64_bwd) -------
__device__ __forceinline__ float l64_out(const float *__restrict__ buf, int F, long j, int row) {
  return buf[(((j >> 5) * F + row) << 5) + (j & 31)];
}
// ---- the spline of one (dimension, sample) in registers ---------------------------------------------------------------
// g64_build / g64_spline_* keep knots, softmax terms and cotangents in arrays indexed by the (run-time) bin: with K a
// run-time value those live in scratch memory (544 / 736 bytes per thread, every access a memory operation).  Here the
// 3K - 1 raw parameters of a dimension (K rows apart in the tile) are loaded ONCE into arrays of a compile-time size
// KM >= K and every loop over bins is unrolled to KM with wave-uniform `i < K` guards: all indexing is static, so the
// arrays are registers, the loads of a pass are issued together, and nothing is read twice.  (The first form of round 3
// walked the rows in run-time loops, a load -> use chain per bin and pass: 127 / 262 us per coupling at the docstring
// shape.)  The bin itself stays a run-time value, compared against the unrolled index.
// Same formulas as above (MonotonicSplines 0.3.3 as restated in oracle/nf_oracle.py); sm_i = exp(v_i - max) / sum.
struct L64Bin {
  int k;            // bin, -1: outside [-B, B)
  float x0, x1;     // knots p[k], p[k + 1]
  float sm;         // softmax term of the bin
};
// the 3K - 1 rows of one (tile, dimension) through a buffer descriptor of the tile (wave-uniform) and a per-lane offset to
// this thread's dimension and sample: row i at scalar offset i * 128 bytes; the same addressing for the cotangent rows
struct L64Raw {
  __amdgpu_buffer_rsrc_t rs;
  int voff;
  __device__ __forceinline__ float operator()(int i) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, i * (L64_TILE * 4), 0));
  }
  __device__ __forceinline__ void put(int i, float v) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, i * (L64_TILE * 4), 0);
  }
};
// rows [row0, ...) of tile `tile` of a tiled buffer with F rows, for the thread of sample `smp`
__device__ __forceinline__ L64Raw l64_raw(const float *buf, int F, long tile, int row0, int smp) {
  L64Raw r;
  r.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(buf) + tile * F * L64_TILE, 0, F * L64_TILE * 4, 0x00020000);
  r.voff = (row0 * L64_TILE + smp) * 4;
  return r;
}
template <int KM>
struct L64Par {
  float w[KM], h[KM], dv[KM];  // widths, heights, interior derivatives (K - 1 of them) as the net wrote them
};
template <int KM, class Raw>
__device__ __forceinline__ void l64_load_par(const Raw &raw, int K, L64Par<KM> &q) {
#pragma unroll
  for (int i = 0; i < KM; ++i) {
    q.w[i] = i < K ? raw(i) : 0.f;
    q.h[i] = i < K ? raw(K + i) : 0.f;
    q.dv[i] = i < K - 1 ? raw(2 * K + i) : 0.f;
  }
}
// v <- exp(v - max) in place, inv <- 1 / their sum: every later pass (bin search, knots, cotangents) multiplies these by inv
// instead of evaluating the exponentials again (round 5: 48 -> 16 expf per dimension in the reverse pass; the same values)
template <int KM>
__device__ __forceinline__ void l64_softmax_exp(float (&v)[KM], int K, float &inv) {
  float mx = v[0];
#pragma unroll
  for (int i = 1; i < KM; ++i)
    if (i < K) mx = fmaxf(mx, v[i]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < KM; ++i)
    if (i < K) { v[i] = expf(v[i] - mx); sum += v[i]; }
  inv = 1.f / sum;
}
// the bin of v among the knots p[j] = -B + 2B cumsum(sm)[j] (g64_bin: the count of interior knots <= v)
template <int KM>
__device__ __forceinline__ L64Bin l64_find(const float (&r)[KM], int K, float B, float inv, float v) {
  L64Bin b{-1, 0.f, 0.f, 0.f};
  float cs = 0.f, left = -B;
  int k = 0;
  float x0 = -B, x1 = -B, smk = 0.f, pK = -B;
#pragma unroll
  for (int i = 0; i < KM; ++i)
    if (i < K) {
      const float sm = r[i] * inv;
      cs += sm;
      const float right = -B + 2.f * B * cs;
      // bin i holds v when i is the number of interior knots (p[1] .. p[K-1]) that are <= v
      const bool take = (i == 0 || v >= left) && (i == K - 1 || !(v >= right));
      if (take) { k = i; x0 = left; x1 = right; smk = sm; }
      left = right;
      pK = right;
    }
  if (v >= -B && v < pK) { b.k = k; b.x0 = x0; b.x1 = x1; b.sm = smk; }
  return b;
}
// knots k, k + 1 of the OTHER axis for a known bin
template <int KM>
__device__ __forceinline__ void l64_knots_at(const float (&r)[KM], int K, float B, float inv, int k, float &y0, float &y1, float &smk) {
  float cs = 0.f;
  y0 = -B; y1 = -B; smk = 0.f;
#pragma unroll
  for (int i = 0; i < KM; ++i)
    if (i < K) {
      const float sm = r[i] * inv;
      if (i == k) { y0 = -B + 2.f * B * cs; smk = sm; }
      if (i <= k) cs += sm;
    }
  y1 = -B + 2.f * B * cs;
}
__device__ __forceinline__ float l64_softplus(float x) { return log1pf(expf(-fabsf(x))) + fmaxf(x, 0.f); }
__device__ __forceinline__ float l64_sigmoid(float x) {
  const float e = expf(-fabsf(x));
  return x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}
// raw interior derivatives number k - 1 and k (the bin's two ends; whichever exists)
template <int KM>
__device__ __forceinline__ void l64_dv_at(const float (&dv)[KM], int k, float &r0, float &r1) {
  r0 = 0.f; r1 = 0.f;
#pragma unroll
  for (int i = 0; i < KM; ++i) {
    if (i == k - 1) r0 = dv[i];
    if (i == k) r1 = dv[i];
  }
}
// forward / inverse of one dimension; logd accumulates log S'(x) (forward) or -log S'(x) (inverse)
template <int KM, class Raw = L64Raw>
__device__ __forceinline__ float l64_spline_apply(const Raw &raw, int K, float B, float v, bool inverse, float &logd) {
  L64Par<KM> q;
  l64_load_par<KM>(raw, K, q);
  float invw, invh;
  l64_softmax_exp<KM>(q.w, K, invw);
  l64_softmax_exp<KM>(q.h, K, invh);
  float x0, x1, y0, y1, dummy;
  int k;
  if (!inverse) {
    const L64Bin b = l64_find<KM>(q.w, K, B, invw, v);
    if (b.k < 0) return v;
    k = b.k; x0 = b.x0; x1 = b.x1;
    l64_knots_at<KM>(q.h, K, B, invh, k, y0, y1, dummy);
  } else {
    const L64Bin b = l64_find<KM>(q.h, K, B, invh, v);
    if (b.k < 0) return v;
    k = b.k; y0 = b.x0; y1 = b.x1;
    l64_knots_at<KM>(q.w, K, B, invw, k, x0, x1, dummy);
  }
  float r0, r1;
  l64_dv_at<KM>(q.dv, k, r0, r1);
  const float d0 = k >= 1 ? l64_softplus(r0) : 1.f, d1 = k + 1 <= K - 1 ? l64_softplus(r1) : 1.f;
  const float dx = x1 - x0, dy = y1 - y0, sl = dy / dx;
  if (!inverse) {
    const float xi = (v - x0) / dx, om = 1.f - xi, den = sl + (d1 + d0 - 2.f * sl) * xi * om;
    logd += g64_logderiv(sl, d0, d1, xi);
    return y0 + dy * (sl * xi * xi + d0 * xi * om) / den;
  }
  const float yy = v - y0, qq = d1 + d0 - 2.f * sl;
  const float aa = dy * (sl - d0) + yy * qq, bb = dy * d0 - yy * qq, cc = -sl * yy;
  const float disc = fmaxf(bb * bb - 4.f * aa * cc, 0.f);
  const float xi = 2.f * cc / (-bb - sqrtf(disc));
  logd -= g64_logderiv(sl, d0, d1, xi);
  return xi * dx + x0;
}
// reverse pass at x (g64_spline_bwd's algebra); writes the 3K - 1 parameter cotangents to out (tile rows, stride 32) and
// returns xbar (inv: the cotangent of the inverse's input, see g64_spline_bwd)
template <int KM, class Out = L64Raw>
__device__ __forceinline__ float l64_spline_bwd(const L64Raw &raw, const Out &out, int K, float B, float x, float ybar, float lbar,
                                                bool inv) {
  L64Par<KM> q;
  l64_load_par<KM>(raw, K, q);
  float invw, invh;
  l64_softmax_exp<KM>(q.w, K, invw);
  l64_softmax_exp<KM>(q.h, K, invh);
  const L64Bin b = l64_find<KM>(q.w, K, B, invw, x);
  if (b.k < 0) {
#pragma unroll
    for (int i = 0; i < KM; ++i) {
      if (i < K) { out.put(i, 0.f); out.put(K + i, 0.f); }
      if (i < K - 1) out.put(2 * K + i, 0.f);
    }
    return ybar;
  }
  const int k = b.k;
  float y0, y1, smh_k;
  l64_knots_at<KM>(q.h, K, B, invh, k, y0, y1, smh_k);
  float r0, r1;
  l64_dv_at<KM>(q.dv, k, r0, r1);
  const float d0 = k >= 1 ? l64_softplus(r0) : 1.f, d1 = k + 1 <= K - 1 ? l64_softplus(r1) : 1.f;
  const float dx = b.x1 - b.x0, dy = y1 - y0;
  const float s = dy / dx, xi = (x - b.x0) / dx, om = 1.f - xi, qd = d1 + d0 - 2.f * s;
  const float den = s + qd * xi * om, num = s * xi * xi + d0 * xi * om;
  const float nd = d1 * xi * xi + 2.f * s * xi * om + d0 * om * om;
  const float dnum_dxi = 2.f * s * xi + d0 * (1.f - 2.f * xi), dden_dxi = qd * (1.f - 2.f * xi);
  const float dnd_dxi = 2.f * d1 * xi + 2.f * s * (1.f - 2.f * xi) - 2.f * d0 * om;
  const float dy_dxi = dy * (dnum_dxi * den - num * dden_dxi) / (den * den);
  const float dL_dxi = dnd_dxi / nd - 2.f * dden_dxi / den;
  float vbar = 0.f;
  if (inv) {
    vbar = (ybar - lbar * dL_dxi / dx) / (dy_dxi / dx);
    ybar = -vbar;
    lbar = -lbar;
  }
  const float dden_ds = 1.f - 2.f * xi * om;
  const float dy_ds = dy * (xi * xi * den - num * dden_ds) / (den * den);
  const float dL_ds = 2.f / s + 2.f * xi * om / nd - 2.f * dden_ds / den;
  const float dy_dd0 = dy * (xi * om * den - num * xi * om) / (den * den), dL_dd0 = om * om / nd - 2.f * xi * om / den;
  const float dy_dd1 = dy * (-num * xi * om) / (den * den), dL_dd1 = xi * xi / nd - 2.f * xi * om / den;
  const float xibar = ybar * dy_dxi + lbar * dL_dxi, sbar = ybar * dy_ds + lbar * dL_ds;
  const float d0bar = ybar * dy_dd0 + lbar * dL_dd0, d1bar = ybar * dy_dd1 + lbar * dL_dd1;
  const float dybar = ybar * num / den + sbar / dx;
  const float dxbar = -sbar * s / dx - xibar * xi / dx;
  const float xkbar = -xibar / dx - dxbar, xk1bar = dxbar, ykbar = ybar - dybar, yk1bar = dybar;
  // sbw_i = 2B (i < k ? xkbar + xk1bar : i == k ? xk1bar : 0);  dotw = sum_i sbw_i sm_i with sum_{i<k} sm_i = (p[k] + B) / 2B
  const float aw = 2.f * B * (xkbar + xk1bar), bw = 2.f * B * xk1bar, ah = 2.f * B * (ykbar + yk1bar), bh = 2.f * B * yk1bar;
  const float dotw = aw * (b.x0 + B) / (2.f * B) + bw * b.sm, doth = ah * (y0 + B) / (2.f * B) + bh * smh_k;
#pragma unroll
  for (int i = 0; i < KM; ++i)
    if (i < K) {
      const float smw = q.w[i] * invw, smh = q.h[i] * invh;
      out.put(i, smw * ((i < k ? aw : i == k ? bw : 0.f) - dotw));
      out.put(K + i, smh * ((i < k ? ah : i == k ? bh : 0.f) - doth));
    }
  // (r0, r1 are q.dv[k - 1], q.dv[k]: two sigmoids, not one per unrolled comparison)
  const float t0 = d0bar * l64_sigmoid(r0), t1 = d1bar * l64_sigmoid(r1);
#pragma unroll
  for (int i = 0; i < KM; ++i)
    if (i < K - 1) out.put(2 * K + i, i == k ? t1 : i == k - 1 ? t0 : 0.f);
  return inv ? vbar : xibar / dx;
}

// A workgroup is one 32-sample tile x 8 dimension lanes: thread (sample = tid & 31, lane = tid >> 5) walks the transformed
// dimensions lane, lane + 8, ... (a spline is 3K - 1 parameters, two softmaxes and a handful of logs per dimension: one thread
// per SAMPLE, as in k_g64_apply, leaves 16 of them in a row on a thread and the chip a quarter full at N = 131 072 --
// 516 + 788 us per coupling against 120 us of MFMA layers).  The log-det terms are added over the 8 lanes in a fixed order.
#define L64_DL 8
template <int KM>
__global__ __launch_bounds__(256) void k_l64_couple_fwd(G64Args a, int inverse, const float *__restrict__ os, int Fs,
                                                        const float *__restrict__ ot, int Ft, float *xy, float *__restrict__ ladj) {
  __shared__ float part[L64_DL][32];
  const int smp = threadIdx.x & 31, dl = threadIdx.x >> 5;
  const long j = (long)blockIdx.x * 32 + smp;
  const bool valid = j < a.N;
  float lsum = 0.f;
  if (valid) {
    float *r = xy + j * a.d;
    if (a.kind == NF_KIND_REALNVP) {
      for (int p = dl; p < a.c; p += L64_DL) {
        const float s = tanh(l64_out(os, Fs, j, p)), t = l64_out(ot, Ft, j, p), v = r[2 * p + a.par_t];
        r[2 * p + a.par_t] = inverse ? (v - t) * exp(-s) : v * exp(s) + t;
        lsum += inverse ? -s : s;
      }
    } else {
      const int P = 3 * a.K - 1;
      for (int p = dl; p < a.c; p += L64_DL) {
        const L64Raw raw = l64_raw(os, Fs, blockIdx.x, p * P, smp);
        const float v = r[2 * p + a.par_t];
        r[2 * p + a.par_t] = l64_spline_apply<KM>(raw, a.K, (float)a.B, v, inverse != 0, lsum);
      }
    }
  }
  part[dl][smp] = lsum;
  __syncthreads();
  if (dl == 0 && valid) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < L64_DL; ++q) t += part[q][smp];
    ladj[j] += t;
  }
}
// x: the coupling's input (inv: the point the inverse is differentiated at); gbar: ybar -> cotangent of the transformed half;
// ds / dt <- cotangents of the nets' outputs (tiles)
template <int KM>
__global__ __launch_bounds__(256) void k_l64_couple_bwd(G64Args a, int inv, const float *__restrict__ os, int Fs, const float *__restrict__ x,
                                                        float *gbar, const float *__restrict__ lbar, float lbar_const,
                                                        float *__restrict__ ds, float *__restrict__ dt, int Ft) {
  const int smp = threadIdx.x & 31, dl = threadIdx.x >> 5;
  const long j = (long)blockIdx.x * 32 + smp;
  const bool valid = j < a.N;
  auto put = [&](float *buf, int F, int row, float v) { buf[(((j >> 5) * F + row) << 5) + (j & 31)] = v; };
  const int P = 3 * a.K - 1;
  const int rows = a.kind == NF_KIND_REALNVP ? a.c : P * a.c;
  // rows beyond the net's outputs sit on the next GEMM's contraction axis, and so do the padding samples of the last tile:
  // zero, not whatever the buffer held
  for (int p = (valid ? rows : 0) + dl; p < Fs; p += L64_DL) put(ds, Fs, p, 0.f);
  if (a.kind == NF_KIND_REALNVP)
    for (int p = (valid ? rows : 0) + dl; p < Ft; p += L64_DL) put(dt, Ft, p, 0.f);
  if (!valid) return;
  const float *xr = x + j * a.d;
  float *gr = gbar + j * a.d;
  const float lb = lbar ? lbar[j] : lbar_const;
  if (a.kind == NF_KIND_REALNVP && inv) {
    for (int p = dl; p < a.c; p += L64_DL) {
      const float s = tanh(l64_out(os, Fs, j, p)), x1 = xr[2 * p + a.par_t], xb = gr[2 * p + a.par_t];
      const float v1b = xb * exp(-s);
      gr[2 * p + a.par_t] = v1b;
      put(ds, Fs, p, (-xb * x1 - lb) * (1.f - s * s));
      put(dt, Ft, p, -v1b);
    }
  } else if (a.kind == NF_KIND_REALNVP) {
    for (int p = dl; p < a.c; p += L64_DL) {
      const float s = tanh(l64_out(os, Fs, j, p)), es = exp(s), x1 = xr[2 * p + a.par_t], yb = gr[2 * p + a.par_t];
      put(dt, Ft, p, yb);
      gr[2 * p + a.par_t] = yb * es;
      put(ds, Fs, p, (yb * x1 * es + lb) * (1.f - s * s));
    }
  } else {
    for (int p = dl; p < a.c; p += L64_DL) {
      const L64Raw raw = l64_raw(os, Fs, blockIdx.x, p * P, smp), out = l64_raw(ds, Fs, blockIdx.x, p * P, smp);
      gr[2 * p + a.par_t] = l64_spline_bwd<KM>(raw, out, a.K, (float)a.B, xr[2 * p + a.par_t], gr[2 * p + a.par_t], lb, inv != 0);
    }
  }
}

// The reverse pass of a spline coupling's OUTPUT layer in one kernel (round 5).  As separate launches the (3K - 1) c parameter
// cotangents of every sample -- 368 floats at the docstring shape nsf(q0, [64, 64], 8, 3.0, 6), 193 MB per coupling at
// N = 131 072 -- were written by k_l64_couple_bwd and read back twice (k_l64_dw_cols, k_l64_bwdx_all): three of the coupling's
// six passes over a buffer of that size.  Here a workgroup of eight waves takes one 32-sample tile at a time:
//   spline stage: every thread is one (sample, dimension): the spline's reverse (l64_spline_bwd, unchanged) with its 3K - 1
//      cotangents written to an LDS tile [parameter row][sample] -- the layout dw_accumulate reads;
//   matrix stage: waves 0-3: dW += a' delta for their 96 of the 384 columns (as k_l64_dw_cols: disjoint columns, no fold);
//      waves 4-7: the same 96 columns' part of the input cotangent, W[:, cols] delta, with the weight slice held in
//      REGISTERS (96 per lane: the 98 KB layer does not fit LDS beside the tiles) -- one dW and one dX wave per SIMD, 96
//      fp32 matrix instructions each per tile;
//   the four partial input cotangents are summed through LDS in a fixed order and stored as the tile the next layer's
//      kernels read.
// The stages of consecutive tiles overlap: between two barriers a dW wave runs [matrix stage of tile t, spline stage of
// tile t + 1] and the dX wave on the same SIMD [spline stage of t + 1, matrix stage of t] -- one wave's matrix instructions
// run under the other's spline arithmetic (tools/trace_l64_top.py: back to back the two stages took 16-19 k + 13 k clocks
// per tile, the matrix pipe's own 12.3 k being the floor).  Hence two delta / activation tiles in LDS.
// The two roles share one register array (accumulators / weight slice): as two arrays both would be live in every wave.
// Needs nin <= 64, (3K - 1) c <= 384, c <= 16 (one dimension per thread); other shapes keep the three launches.
// after each matrix instruction of the matrix stages: the wave idles for most of the instruction's 64 clocks instead of
// presenting the next one at once -- a matrix instruction that waits for the pipe waits in the SIMD's vector issue stage, and
// the other wave's spline arithmetic waits behind it (measured: its loads, requested at the start of the interval, were
// consumed only when the matrix stage of the wave beside it had ended)
#ifndef NF_L64_PACE_NOP
#define NF_L64_PACE_NOP 11  // 48 clocks of the instruction's 64 (9 / 13 / 15 measured beside it: see profiles/r5g)
#endif
struct L64Pace {
  __device__ __forceinline__ void operator()(int) const {
#ifndef NF_L64_NO_PACE
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop %0" ::"n"(NF_L64_PACE_NOP) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
};
struct L64LdsOut {
  float *p;  // row 0 of this (dimension, sample) in the LDS tile
  __device__ __forceinline__ void put(int i, float v) const { p[i * NF_TS] = v; }
};
#define L64_TOP_ROWS 384
#define L64_TOP_LDS ((2 * L64_TOP_ROWS + 2 * 64 + 4 * 64) * NF_TS * 4)
// KC = 8: K is 8 and the element arithmetic is nf_rqs_elem.h's (compile-time K, unnormalised prefix sums, packed width /
// height pairs, hardware exp / log / rcp -- what the fused spline kernels of the hidden <= 32 shapes ship with): 16 k -> 9 k
// clocks of spline stage per tile.  KC = 0: run-time K <= KM, l64_spline_bwd.
template <int KM, int KC>
__global__ __launch_bounds__(512) void k_l64_nsf_top_bwd(G64Args a, int inv, const float *__restrict__ theta, L64Layer L,
                                                         const float *__restrict__ os, int Fs, const float *__restrict__ hact, int Fh,
                                                         const float *__restrict__ x, float *gbar, const float *__restrict__ lbar,
                                                         float lbar_const, float *__restrict__ gdst, int Fd, float *__restrict__ slabs,
                                                         long Pc, long slab_off, long long *trace) {
#ifdef NF_KERNEL_TRACE  // tools/trace_l64_top.py: wave 0 (a dW wave) at [0 ...], wave 4 (a dX wave) at [64 ...] of workgroup 0
#ifdef NF_TRACE_HIDDEN  // (tools/trace_l64_top.py HIDDEN=1: the buffer is k_l64_hidden_bwd's)
  long long *tr = nullptr;
#else
  long long *tr = trace && blockIdx.x == 0 && (threadIdx.x & 255) == 0 ? trace + (threadIdx.x >> 8) * 64 : nullptr;
#endif
#define L64T_STAMP(slot) do { if (tr) { __builtin_amdgcn_sched_barrier(0); tr[slot] = clock64(); } } while (0)
#else
#define L64T_STAMP(slot) do { } while (0)
#endif
  constexpr int SD = L64_TOP_ROWS * NF_TS, SA = 64 * NF_TS, WS = L64_TOP_ROWS + 1;
  static_assert(64 * WS <= 2 * SD, "the prologue stages the whole layer in the two delta tiles");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *sdt = sm, *sat = sm + 2 * SD, *red = sat + 2 * SA;  // 2 delta tiles, 2 activation tiles, 4 partial cotangents
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int smp = tid & 31, dl = tid >> 5;
  const bool dxw = wave >= 4;
  const int o0w = 96 * (wave & 3);
  L64T_STAMP(0);
  // the layer through LDS (coalesced rows, sixteen requests per thread in flight), then each dX wave's slice into registers
  {
    constexpr int NE = 64 * L64_TOP_ROWS, U = 16;
#pragma unroll 1
    for (int e0 = tid; e0 < NE; e0 += 512 * U) {
      float v[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int e = e0 + 512 * k, i = e / L64_TOP_ROWS, o = e - i * L64_TOP_ROWS;
        v[k] = (i < L.nin && o < L.nout) ? theta[L.w_off + (long)i * L.nout + o] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int e = e0 + 512 * k, i = e / L64_TOP_ROWS, o = e - i * L64_TOP_ROWS;
        sm[i * WS + o] = v[k];
      }
    }
  }
  __syncthreads();
  f32x16 R[2][3];  // dW waves: acc[ib][ob]; dX waves: W[32 ib + l31][o0w + 8 g + 4 hi + e] at flat index (ib * 12 + g) * 4 + e
  float bsum[3] = {0.f, 0.f, 0.f};
  if (dxw) {
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int g = 0; g < 12; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = (ib * 12 + g) * 4 + e;
          R[f / 48][(f / 16) % 3][f % 16] = sm[(32 * ib + l31) * WS + o0w + 8 * g + 4 * hi + e];
        }
  } else {
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int ob = 0; ob < 3; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) R[ib][ob][r] = 0.f;
  }
  __syncthreads();
  const int P = 3 * a.K - 1, rows = P * a.c;
  const long ntiles = (a.N + L64_TILE - 1) / L64_TILE;
  L64T_STAMP(1);
  int tslot = 2;
  long tile = -1, nxt = blockIdx.x;
  int cur = 0;  // the buffers of `tile`; the spline stage of `nxt` fills the other pair
  for (;;) {
    const bool have = tile >= 0, more = nxt < ntiles;  // workgroup-uniform
    if (!have && !more) break;
    float *sd = sdt + cur * SD, *sa = sat + cur * SA;
    L64T_STAMP(tslot + 0);
    if (have && !dxw) dw_accumulate<2, 3>(sa, sd + o0w * NF_TS, R, bsum, l31, hi, L64Pace());
    L64T_STAMP(tslot + 1);
    if (more) {
      float *sdn = sdt + (cur ^ 1) * SD, *san = sat + (cur ^ 1) * SA;
      const long j = nxt * L64_TILE + smp;
      const bool valid = j < a.N;
      // the activation tile: requested first, placed after the spline arithmetic
      float av[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = tid + 512 * q, row = e >> 5;
        av[q] = row < Fh ? hact[(nxt * Fh + row) * L64_TILE + (e & 31)] : 0.f;
      }
      // rows past the parameters (and every row of a padding sample) sit on the contraction axes: zero
      for (int p = (valid ? rows : 0) + dl; p < L64_TOP_ROWS; p += 16) sdn[p * NF_TS + smp] = 0.f;
      if (valid) {
        const float *xr = x + j * a.d;
        float *gr = gbar + j * a.d;
        const float lb = lbar ? lbar[j] : lbar_const;
        for (int p = dl; p < a.c; p += 16) {
          const L64Raw raw = l64_raw(os, Fs, nxt, p * P, smp);
#ifdef NF_KERNEL_TRACE  // (interval 1 only) [56]: the tile's loads have arrived -- a probe load and a full wait, trace builds only
          if (tslot == 8) {
            float probe = raw(0) + xr[2 * p + a.par_t] + gr[2 * p + a.par_t];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(probe)::"memory");
            L64T_STAMP(56);
            if (probe == 1.2345e-30f) sdn[0] = probe;
          }
#endif
          const L64LdsOut out{sdn + p * P * NF_TS + smp};
          if constexpr (KC == 8) {
            float rw[23], thb[23];
#pragma unroll
            for (int i = 0; i < 23; ++i) rw[i] = raw(i);
            Knots<8> kn;
            build_knots<8>(rw, (float)a.B, kn);
            Bin<8> bn;
            const float xv = xr[2 * p + a.par_t], yb = gr[2 * p + a.par_t];
            find_bin<8>(kn, xv, bn);
            const float xi = nf_fdiv(xv - bn.xk, bn.dx);
            const float xb = inv ? rqs_bwd_elem<8, true>(kn, bn, xi, (float)a.B, yb, lb, thb)
                                 : rqs_bwd_elem<8, false>(kn, bn, xi, (float)a.B, yb, lb, thb);
#pragma unroll
            for (int i = 0; i < 23; ++i) out.put(i, thb[i]);
            gr[2 * p + a.par_t] = xb;
          } else
          gr[2 * p + a.par_t] = l64_spline_bwd<KM>(raw, out, a.K, (float)a.B, xr[2 * p + a.par_t], gr[2 * p + a.par_t], lb, inv != 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = tid + 512 * q;
        san[(e >> 5) * NF_TS + (e & 31)] = av[q];
      }
    }
    L64T_STAMP(tslot + 2);
    if (have && dxw) {
      f32x16 dlr[3], din[2];
#pragma unroll
      for (int ob = 0; ob < 3; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) dlr[ob][r] = sd[(o0w + 32 * ob + nf_row(r, hi)) * NF_TS + l31];
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) din[ib][r] = 0.f;
#pragma unroll
      for (int g = 0; g < 12; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ib = 0; ib < 2; ++ib) {
            const int f = (ib * 12 + g) * 4 + e;
            din[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(R[f / 48][(f / 16) % 3][f % 16], dlr[g / 4][(g % 4) * 4 + e], din[ib], 0, 0, 0);
            L64Pace()(0);
          }
      float *mine = red + (wave & 3) * SA;
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(32 * ib + nf_row(r, hi)) * NF_TS + l31] = din[ib][r];
    }
    L64T_STAMP(tslot + 3);
    __syncthreads();
    L64T_STAMP(tslot + 4);
    if (have) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = tid + 512 * q, row = e >> 5, at = row * NF_TS + (e & 31);
        const float v = ((red[at] + red[SA + at]) + red[2 * SA + at]) + red[3 * SA + at];
        if (row < Fd) gdst[(tile * Fd + row) * L64_TILE + (e & 31)] = v;
      }
    }
    __syncthreads();  // `red` is read before the next interval's dX waves write it
    L64T_STAMP(tslot + 5);
    if (tslot < 50) tslot += 6;
    if (!more) break;
    tile = nxt;
    nxt += gridDim.x;
    cur ^= 1;
  }
  L64T_STAMP(62);
  if (!dxw) {
    float *slab = slabs + (long)blockIdx.x * Pc - slab_off;
#pragma unroll
    for (int ob = 0; ob < 3; ++ob) {
      const int o = o0w + 32 * ob + l31;
      if (o < L.nout) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = 32 * ib + nf_row(r, hi);
            if (i < L.nin) slab[L.w_off + (long)i * L.nout + o] = R[ib][ob][r];
          }
      }
      const float v = bsum[ob] + __shfl_xor(bsum[ob], 32);
      if (hi == 0 && o < L.nout) slab[L.b_off + o] = v;
    }
  }
  L64T_STAMP(63);
}

// The FORWARD of a spline coupling's output layer with the spline itself (round 5): k_l64_fwd_all wrote the 3K - 1 raw
// parameters per dimension and k_l64_couple_fwd read them back -- here they go to HBM once (the reverse pass reads them) and
// to the spline through LDS.  A workgroup takes two 32-sample tiles at a time, four waves each: wave w of a tile owns output
// columns [96 w, 96 w + 96) with its weight slice in registers (as the dX waves of k_l64_nsf_top_bwd), 96 fp32 matrix
// instructions per tile; then the tile's 256 threads are (sample, dimension lane) exactly as k_l64_couple_fwd's -- the same
// walk over the dimensions and the same fixed-order sum of the log-determinant terms, so the results are the same bits.
struct L64LdsIn {
  const float *p;  // row 0 of this (dimension, sample) in the LDS tile
  __device__ __forceinline__ float operator()(int i) const { return p[i * NF_TS]; }
};
#define L64_TOPF_LDS ((2 * L64_TOP_ROWS + 2 * 64 + 2 * L64_DL) * NF_TS * 4 + L64_TOP_ROWS * 4)
template <int KM, int KC>  // KC: as k_l64_nsf_top_bwd
__global__ __launch_bounds__(512) void k_l64_nsf_top_fwd(G64Args a, int inverse, const float *__restrict__ theta, L64Layer L,
                                                         const float *__restrict__ hact, int Fh, float *__restrict__ os, int Fs, float *xy,
                                                         float *__restrict__ ladj, long long *trace) {
#ifdef NF_KERNEL_TRACE  // tools/trace_l64_top.py: workgroup 0 / wave 0 at [0 ...]
  long long *trf = trace && blockIdx.x == 0 && threadIdx.x == 0 ? trace : nullptr;
#define L64F_STAMP(slot) do { if (trf) { __builtin_amdgcn_sched_barrier(0); trf[slot] = clock64(); } } while (0)
#else
#define L64F_STAMP(slot) do { } while (0)
#endif
  L64F_STAMP(0);
  int fslot = 2;
  constexpr int SD = L64_TOP_ROWS * NF_TS, SA = 64 * NF_TS, WS = L64_TOP_ROWS + 1;
  static_assert(64 * WS <= 2 * SD, "the prologue stages the whole layer in the two parameter tiles");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *sdt = sm, *sat = sm + 2 * SD, *part = sat + 2 * SA, *bias = part + 2 * L64_DL * NF_TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int half = tid >> 8, t256 = tid & 255, smp = tid & 31, dl = (tid >> 5) & 7;
  const int o0w = 96 * (wave & 3);
  {
    constexpr int NE = 64 * L64_TOP_ROWS, U = 16;
#pragma unroll 1
    for (int e0 = tid; e0 < NE; e0 += 512 * U) {
      float v[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int e = e0 + 512 * k, i = e / L64_TOP_ROWS, o = e - i * L64_TOP_ROWS;
        v[k] = (i < L.nin && o < L.nout) ? theta[L.w_off + (long)i * L.nout + o] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int e = e0 + 512 * k, i = e / L64_TOP_ROWS, o = e - i * L64_TOP_ROWS;
        sm[i * WS + o] = v[k];
      }
    }
    if (tid < L64_TOP_ROWS) bias[tid] = tid < L.nout ? theta[L.b_off + tid] : 0.f;
  }
  __syncthreads();
  float W[3][32];  // W[2 kk + hi][o0w + 32 ob + l31]: the A operand of k-step kk for output block ob
#pragma unroll
  for (int ob = 0; ob < 3; ++ob)
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) W[ob][kk] = sm[(2 * kk + hi) * WS + o0w + 32 * ob + l31];
  __syncthreads();
  float *sd = sdt + half * SD, *sa = sat + half * SA, *pt = part + half * L64_DL * NF_TS;
  const int P = 3 * a.K - 1;
  const long ntiles = (a.N + L64_TILE - 1) / L64_TILE;
  // a tile's activations [64][32] go through registers: requested an interval ahead (tools/trace_l64_top.py: 6 k of a pair's
  // 40 k clocks were this load's round trip)
  float hn[8];
  auto request = [&](long t) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = t256 + 256 * q, row = e >> 5;
      hn[q] = (t < ntiles && row < Fh) ? hact[(t * Fh + row) * L64_TILE + (e & 31)] : 0.f;
    }
  };
  request(2 * (long)blockIdx.x + half);
  L64F_STAMP(1);
  for (long pair = blockIdx.x; 2 * pair < ntiles; pair += gridDim.x) {
    const long tile = 2 * pair + half;
    const bool live = tile < ntiles;  // (wave-uniform: the second tile of the last pair may not exist)
    L64F_STAMP(fslot + 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = t256 + 256 * q;
      sa[(e >> 5) * NF_TS + (e & 31)] = hn[q];
    }
    request(2 * (pair + gridDim.x) + half);
    // this tile's state and log-determinant values: requested now, used after the matrix stage
    const long j = tile * L64_TILE + smp;
    const bool valid = live && j < a.N;
    float *xr = xy + j * a.d;
    float xv[2] = {0.f, 0.f}, lold = 0.f;
    if (valid) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (dl + L64_DL * u < a.c) xv[u] = xr[2 * (dl + L64_DL * u) + a.par_t];
      if (dl == 0) lold = ladj[j];
    }
    __syncthreads();
    L64F_STAMP(fslot + 1);
    if (live) {
      f32x16 out[3];
#pragma unroll
      for (int ob = 0; ob < 3; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ob][r] = bias[o0w + 32 * ob + nf_row(r, hi)];
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) {
        const float hb = sa[(2 * kk + hi) * NF_TS + l31];
#pragma unroll
        for (int ob = 0; ob < 3; ++ob) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[ob][kk], hb, out[ob], 0, 0, 0);
      }
      L64F_STAMP(fslot + 2);
#pragma unroll
      for (int ob = 0; ob < 3; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) sd[(o0w + 32 * ob + nf_row(r, hi)) * NF_TS + l31] = out[ob][r];
    }
    L64F_STAMP(fslot + 3);
    __syncthreads();
    L64F_STAMP(fslot + 4);
    if (live) {
      // the raw parameters to HBM for the reverse pass: 16 bytes per lane from the LDS tile (as 4-byte stores straight from the
      // accumulators this was 6 k clocks of the matrix stage: 384 store instructions per pair and CU, all workgroups at once)
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int idx = t256 + 256 * q, row = idx >> 3, g4 = idx & 7;
        const float *src = sd + row * NF_TS + 4 * g4;
        const float4 v = make_float4(src[0], src[1], src[2], src[3]);
        if (row < Fs) *reinterpret_cast<float4 *>(os + (tile * Fs + row) * L64_TILE + 4 * g4) = v;
      }
    }
    float lsum = 0.f;
    if (valid) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int p = dl + L64_DL * u;
        if (p < a.c) {
          const L64LdsIn raw{sd + p * P * NF_TS + smp};
          if constexpr (KC == 8) {
            float rw[23];
#pragma unroll
            for (int i = 0; i < 23; ++i) rw[i] = raw(i);
            Knots<8> kn;
            build_knots<8>(rw, (float)a.B, kn);
            float xi;
            if (!inverse) {
              unsigned code;
              xr[2 * p + a.par_t] = rqs_fwd_elem<8>(kn, xv[u], lsum, code, xi);
            } else {
              Bin<8> bn;
              xr[2 * p + a.par_t] = rqs_inv_elem<8>(kn, xv[u], lsum, bn, xi);
            }
          } else
          xr[2 * p + a.par_t] = l64_spline_apply<KM>(raw, a.K, (float)a.B, xv[u], inverse != 0, lsum);
        }
      }
    }
    pt[dl * NF_TS + smp] = lsum;
    L64F_STAMP(fslot + 5);
    __syncthreads();
    if (dl == 0 && valid) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < L64_DL; ++q) t += pt[q * NF_TS + smp];
      ladj[j] = lold + t;
    }
    L64F_STAMP(fslot + 6);
    if (fslot < 50) fslot += 7;
    // (the next pair's stores into sa / sd / part come after its first barrier or after this one: no wave still reads them)
  }
}

// The reverse pass of a net's layers BELOW the output layer in one kernel (round 5; spline couplings whose output layer ran in
// k_l64_nsf_top_bwd): per hidden layer the general path launched k_l64_dw and k_l64_bwdx -- four launches and eight passes
// over 64-row tiles for two hidden layers.  Here a wave takes a 32-sample tile through all of them: delta = cotangent x
// leaky-ReLU'(stashed output), dW += a' delta (operands through the wave's LDS scratch, as k_l64_dw), cotangent <- W delta
// (weights of every layer in LDS), the first layer's input cotangent added to the conditioner half of gbar.  Each stashed
// activation tile is read once (it is the mask of its own layer and the `a` operand of the one above).
// NH hidden layers, every width <= 64; IB0 = 32-row blocks of the first layer's inputs.
struct L64Hid {
  long w_off[2], b_off[2];
  int nin[2], nout[2], F[2];
  const float *act[2];  // the layers' stashed outputs (tiles, F rows)
};
template <int IB, int OB>
__device__ __forceinline__ void l64_fold_to_slab(float *img, const f32x16 (&acc)[IB][OB], const float (&bsum)[OB], long w_off, long b_off,
                                                 int nin, int nout, float *slab, int tid, int wave, int l31, int hi) {
  // waves in a fixed order through one [32 IB][32 OB] image (+ bias row) in LDS, then the slab in theta order (as k_l64_dw)
  constexpr int SI = 32 * OB;
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float *p = img + (32 * ib + nf_row(r, hi)) * SI + 32 * ob + l31;
            *p = wv == 0 ? acc[ib][ob][r] : *p + acc[ib][ob][r];
          }
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) {
        const float v = bsum[ob] + __shfl_xor(bsum[ob], 32);
        if (hi == 0) {
          float *p = img + 32 * IB * SI + 32 * ob + l31;
          *p = wv == 0 ? v : *p + v;
        }
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < 32 * IB * SI; e += 256) {
    const int i = e / SI, o = e - i * SI;
    if (i < nin && o < nout) slab[w_off + (long)i * nout + o] = img[e];
  }
  for (int o = tid; o < SI; o += 256)
    if (o < nout) slab[b_off + o] = img[32 * IB * SI + o];
  __syncthreads();
}
template <int NH, int IB0>
__global__ __launch_bounds__(256) void k_l64_hidden_bwd(const float *__restrict__ theta, L64Hid hd, const float *__restrict__ gtop, int Fg,
                                                        L64Src xin, float *__restrict__ gbar, long N, float *__restrict__ slabs, long Pc,
                                                        long slab_off, long long *trace) {
#if defined(NF_KERNEL_TRACE) && defined(NF_TRACE_HIDDEN)  // tools/trace_l64_top.py HIDDEN=1: workgroup 0 / wave 0
  long long *trh = trace && blockIdx.x == 0 && threadIdx.x == 0 ? trace : nullptr;
#define L64H_STAMP(slot) do { if (trh) { __builtin_amdgcn_sched_barrier(0); trh[slot] = clock64(); } } while (0)
#else
#define L64H_STAMP(slot) do { } while (0)
#endif
  L64H_STAMP(0);
  int hslot = 2;
  constexpr int S = 64 + NF_IMG_PAD, WG = 64 * S + 64, SC = 4 * 32 * NF_TS;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // NH weight images, then a scratch pair per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  for (int l = 0; l < NH; ++l) {
    const L64Layer L{hd.w_off[l], hd.b_off[l], hd.nin[l], hd.nout[l], 0};
    l64_stage<2, 2>(sm + l * WG, sm + l * WG + 64 * S, theta, L, tid, 256);  // (the bias row is staged and not used)
  }
  __syncthreads();
  float *scr = sm + NH * WG, *sa = scr + wave * SC, *sd = sa + 2 * 32 * NF_TS;
  f32x16 acc1[2][2], acc0[IB0][2];
  float bs1[2] = {0.f, 0.f}, bs0[2] = {0.f, 0.f};
#pragma unroll
  for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[ib][ob][r] = 0.f;
#pragma unroll
    for (int ib = 0; ib < IB0; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[ib][ob][r] = 0.f;
  }
  const long ntiles = (N + L64_TILE - 1) / L64_TILE;
  L64H_STAMP(1);
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    L64H_STAMP(hslot + 0);
    f32x16 g[2], hl[2];
    l64_load<2>(L64Src{gtop, Fg, 0, 0, 0}, tile, l31, hi, N, 64, g);
    l64_load<2>(L64Src{hd.act[NH - 1], hd.F[NH - 1], 0, 0, 0}, tile, l31, hi, N, 64, hl);
    // the first layer's inputs and the cotangent its result is added to: requested with the rest (the matrix stages between
    // here and their use are fenced, so the compiler leaves a load where it is written: tools/trace_l64_top.py HIDDEN=1 had
    // 2.2-3.3 k + 3-4 k clocks of a tile's 24 k in these two round trips)
    f32x16 x0[IB0];
    l64_load<IB0>(xin, tile, l31, hi, N, hd.nin[0], x0);
    const L64Io gio = l64_io_std(gbar, xin.d, xin.par, tile, N, l31, hi);
    float gold[IB0][16];
#pragma unroll
    for (int ib = 0; ib < IB0; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) gold[ib][r] = l64_ld(gio, l64_rc(ib, r) * 8);
    if (NH == 2) {
      f32x16 av[2], din[2];
      l64_load<2>(L64Src{hd.act[0], hd.F[0], 0, 0, 0}, tile, l31, hi, N, 64, av);
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[ob][r] *= hl[ob][r] > 0.f ? 1.f : 0.01f;
      L64H_STAMP(hslot + 1);
      tile_to_scratch<2>(sa, av, l31, hi);
      tile_to_scratch<2>(sd, g, l31, hi);
      wave_lds_fence();
      L64H_STAMP(hslot + 2);
      dw_accumulate<2, 2>(sa, sd, acc1, bs1, l31, hi);
      wave_lds_fence();
      L64H_STAMP(hslot + 3);
      dense_bwd_x<2, 2, S, false>(sm + WG, g, din, l31, hi);
      L64H_STAMP(hslot + 4);
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) { g[ib][r] = din[ib][r]; hl[ib][r] = av[ib][r]; }
    }
    {
      f32x16 din[IB0];
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[ob][r] *= hl[ob][r] > 0.f ? 1.f : 0.01f;
      L64H_STAMP(hslot + 5);
      tile_to_scratch<IB0>(sa, x0, l31, hi);
      tile_to_scratch<2>(sd, g, l31, hi);
      wave_lds_fence();
      dw_accumulate<IB0, 2>(sa, sd, acc0, bs0, l31, hi);
      wave_lds_fence();
      L64H_STAMP(hslot + 6);
      dense_bwd_x<IB0, 2, S, false>(sm, g, din, l31, hi);
      L64H_STAMP(hslot + 7);
#pragma unroll
      for (int ib = 0; ib < IB0; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (l64_rc(ib, r) + 4 * hi < hd.nin[0]) l64_st(gio, l64_rc(ib, r) * 8, gold[ib][r] + din[ib][r]);  // (as l64_store_din)
    }
    L64H_STAMP(hslot + 8);
    if (hslot < 40) hslot += 9;
  }
  L64H_STAMP(60);
  __syncthreads();
  L64H_STAMP(61);
  float *slab = slabs + (long)blockIdx.x * Pc - slab_off;
  if (NH == 2) l64_fold_to_slab<2, 2>(scr, acc1, bs1, hd.w_off[1], hd.b_off[1], hd.nin[1], hd.nout[1], slab, tid, wave, l31, hi);
  l64_fold_to_slab<IB0, 2>(scr, acc0, bs0, hd.w_off[0], hd.b_off[0], hd.nin[0], hd.nout[0], slab, tid, wave, l31, hi);
  L64H_STAMP(62);
}
#define L64_HID_LDS(NH) (((NH) * (64 * (64 + NF_IMG_PAD) + 64) + 4 * 4 * 32 * NF_TS) * 4)

// ---- host side --------------------------------------------------------------------------------
template <class SZ>
static bool g64_fits(const nf_flow_desc *desc) {
  for (int i = 0; i < desc->n_hidden; ++i)
    if (desc->hdims[i] < 1 || desc->hdims[i] > SZ::MAXH) return false;
  const int c = (desc->d + 1) / 2;
  if (c > SZ::MAXC) return false;
  if (desc->kind == NF_KIND_NSF) {
    if (desc->K < 2 || desc->K > G64_MAXK || !(desc->B > 0.f)) return false;
    if ((3 * desc->K - 1) * c > SZ::MAXO) return false;
  }
  return true;
}

bool nf_g64_supported(const nf_flow_desc *desc) {
  if (desc->dtype != NF_DTYPE_F64 && desc->dtype != NF_DTYPE_F32) return false;
  if (desc->kind != NF_KIND_REALNVP && desc->kind != NF_KIND_NSF) return false;
  if (desc->n_hidden < 1 || desc->n_hidden > NF_MAX_HIDDEN || desc->d < 2) return false;
  return g64_fits<G64Large>(desc);
}

static long fill_net(G64Net *n, long off, int nin, const nf_flow_desc *desc, int nout) {
  n->nl = desc->n_hidden + 1;
  n->dims[0] = nin;
  for (int i = 0; i < desc->n_hidden; ++i) n->dims[i + 1] = desc->hdims[i];
  n->dims[n->nl] = nout;
  for (int l = 0; l < n->nl; ++l) {
    n->w[l] = off;
    off += (long)n->dims[l] * n->dims[l + 1];
    n->b[l] = off;
    off += n->dims[l + 1];
  }
  return off;
}

static G64Args make_g64_args(const nf_flow_desc *desc, int k, long N) {
  const CouplingInfo ci = nf_coupling_info(desc, k);
  G64Args a;
  a.kind = desc->kind; a.d = desc->d; a.c = ci.c; a.m = ci.m; a.par_t = ci.par_t; a.K = desc->K; a.B = desc->B; a.N = N;
  if (desc->kind == NF_KIND_REALNVP) {
    const long off = fill_net(&a.net[0], ci.theta_off, ci.m, desc, ci.c);
    fill_net(&a.net[1], off, ci.m, desc, ci.c);
  } else {
    fill_net(&a.net[0], ci.theta_off, ci.m, desc, (3 * desc->K - 1) * ci.c);
    a.net[1] = a.net[0];
  }
  return a;
}

// launches of the two kernels in the size class the flow fits (Float32 flows: the MLP on the matrix pipe, "l64" above)
static bool l64_ok(const nf_flow_desc *desc);
static int l64_apply(nf_ctx *ctx, const nf_flow_desc *desc, const G64Args &a, int inverse, const float *theta, float *xy, float *ladj,
                     int slot);
static int l64_bwd(nf_ctx *ctx, const nf_flow_desc *desc, int k, const G64Args &a, int inv, const float *theta, const float *x, float *gbar,
                   const float *lbar, float lbar_const, float *g, float *slabs, bool kept);
#include "nf_g64m.h"  // Float64 RealNVP couplings on v_mfma_f64_16x16x4_f64 (round 5)

template <class T>
static int g64_launch_apply(nf_ctx *ctx, const nf_flow_desc *desc, unsigned grid, const G64Args &a, int inverse, const T *theta,
                            const T *x, T *y, T *ladj, int slot = 0) {
  if constexpr (sizeof(T) == 8) {
    if (const int gm = g64m_geo(desc)) {
      ProfScope ps(ctx, "g64m_apply");
#define G64M_CALL(G) g64m_launch_apply<G>(ctx, a, inverse, theta, x, y, ladj)
      return G64M_DISPATCH(gm, G64M_CALL);
#undef G64M_CALL
    }
    if (g64m_nsf_ok(desc)) {
      ProfScope ps(ctx, "g64m_apply");
      return g64m_nsf_launch_apply(ctx, a, inverse, theta, x, y, ladj);
    }
  }
  if constexpr (sizeof(T) == 4) {
    if (l64_ok(desc)) {
      if (y != x) NF_HIP(hipMemcpyAsync(y, x, (size_t)a.N * a.d * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
      return l64_apply(ctx, desc, a, inverse, theta, y, ladj, slot);  // (its scratch reservation can fail in a caller's arena)
    }
  }
  if (g64_fits<G64Small>(desc))
    hipLaunchKernelGGL((k_g64_apply<T, G64Small>), dim3(grid), dim3(G64_BLOCK), 0, ctx->stream, a, inverse, theta, x, y, ladj);
  else
    hipLaunchKernelGGL((k_g64_apply<T, G64Large>), dim3(grid), dim3(G64_BLOCK), 0, ctx->stream, a, inverse, theta, x, y, ladj);
  NF_HIP(hipGetLastError());
  return NF_OK;
}
// workgroups (= gradient slabs) of the reverse kernel: one per 64-sample tile up to a cap that keeps the slab area of
// the largest coupling under 256 MB (a workgroup then walks several tiles, accumulating in its slab)
static long g64_coupling_params(const nf_flow_desc *desc) {
  long m = 0;
  for (int k = 0; k < 2 && k < 2 * desc->nlayers; ++k) {
    const long p = nf_coupling_info(desc, k).nparams;
    if (p > m) m = p;
  }
  return m;
}
static unsigned g64_bwd_blocks(const nf_flow_desc *desc, long N) {
  const size_t es = desc->dtype == NF_DTYPE_F64 ? 8 : 4;
  long cap = (long)((size_t)256 << 20) / (long)(g64_coupling_params(desc) * es + 1);
  cap = cap < 64 ? 64 : cap > 2048 ? 2048 : cap;
  long nb = (N + G64_BLOCK - 1) / G64_BLOCK;
  nb = nb > cap ? cap : nb;
  return (unsigned)(nb < 1 ? 1 : nb);
}
static size_t g64_slab_bytes(const nf_flow_desc *desc, long N) {
  return (((size_t)g64_bwd_blocks(desc, N) * g64_coupling_params(desc) * sizeof(double)) + 255) / 256 * 256;  // sized for f64
}
int nf_launch_reduce_slabs(nf_ctx *, int, const void *, int, long, void *);

// reverse kernel of coupling k (partial gradients into `slabs`) + the ordered sum of the slabs into g[theta_off ...]
template <class T>
static int g64_launch_bwd(nf_ctx *ctx, const nf_flow_desc *desc, int k, const G64Args &a, int inv, const T *theta,
                          const T *x, T *gbar, const T *lbar, T lbar_const, T *g, T *slabs, bool kept = false) {
  if constexpr (sizeof(T) == 4) {
    if (l64_ok(desc)) return l64_bwd(ctx, desc, k, a, inv, theta, x, gbar, lbar, lbar_const, g, slabs, kept);
  }
  const CouplingInfo ci = nf_coupling_info(desc, k);
  if constexpr (sizeof(T) == 8) {
    if (const int gm = g64m_geo(desc)) {
      unsigned gridm = g64_bwd_blocks(desc, a.N);  // (the slab area is sized for that many workgroups)
      if (gridm > (unsigned)ctx->num_cu) gridm = (unsigned)ctx->num_cu;
      {
        ProfScope ps(ctx, "g64m_bwd");
#define G64M_CALL(G) g64m_launch_bwd<G>(ctx, a, inv, theta, x, gbar, lbar, lbar_const, slabs, ci.nparams, ci.theta_off, gridm)
        NF_TRY(G64M_DISPATCH(gm, G64M_CALL));
#undef G64M_CALL
      }
      return nf_launch_reduce_slabs(ctx, NF_DTYPE_F64, slabs, (int)gridm, ci.nparams, g + ci.theta_off);
    }
    if (g64m_nsf_ok(desc)) {
      unsigned gridm = g64_bwd_blocks(desc, a.N);
      if (gridm > (unsigned)ctx->num_cu) gridm = (unsigned)ctx->num_cu;
      {
        ProfScope ps(ctx, "g64m_bwd");
        NF_TRY(g64m_nsf_launch_bwd(ctx, a, inv, theta, x, gbar, lbar, lbar_const, slabs, ci.nparams, ci.theta_off, gridm));
      }
      return nf_launch_reduce_slabs(ctx, NF_DTYPE_F64, slabs, (int)gridm, ci.nparams, g + ci.theta_off);
    }
  }
  const unsigned grid = g64_bwd_blocks(desc, a.N);
  if (g64_fits<G64Small>(desc))
    hipLaunchKernelGGL((k_g64_bwd<T, G64Small>), dim3(grid), dim3(G64_BLOCK), 0, ctx->stream, a, inv, theta, x, gbar, lbar, lbar_const,
                       slabs, ci.nparams, ci.theta_off);
  else
    hipLaunchKernelGGL((k_g64_bwd<T, G64Large>), dim3(grid), dim3(G64_BLOCK), 0, ctx->stream, a, inv, theta, x, gbar, lbar, lbar_const,
                       slabs, ci.nparams, ci.theta_off);
  NF_HIP(hipGetLastError());
  return nf_launch_reduce_slabs(ctx, sizeof(T) == 8 ? NF_DTYPE_F64 : NF_DTYPE_F32, slabs, (int)grid, ci.nparams, g + ci.theta_off);
}

// ---- l64 host side ---------------------------------------------------------------------------------------------
int nf_wimg_reserve(nf_ctx *ctx, size_t bytes);
static inline int l64_pad32(int n) { return (n + 31) / 32 * 32; }
static bool l64_ok(const nf_flow_desc *desc) {
  constexpr bool off = false;  // (round 6: the NF_G64_NO_MFMA A/B switch is retired -- its question is answered, README "switches")
  if (off || desc->dtype != NF_DTYPE_F32 || (desc->kind != NF_KIND_REALNVP && desc->kind != NF_KIND_NSF)) return false;
  if (desc->n_hidden < 1 || desc->n_hidden > NF_MAX_HIDDEN || desc->d < 2 || desc->d > 256) return false;
  for (int i = 0; i < desc->n_hidden; ++i)
    if (desc->hdims[i] > 256 || desc->hdims[i] < 1) return false;
  if (desc->kind == NF_KIND_NSF && (desc->K < 1 || desc->K > G64_MAXK)) return false;
  const int nout = desc->kind == NF_KIND_REALNVP ? (desc->d + 1) / 2 : (3 * desc->K - 1) * ((desc->d + 1) / 2);
  return nout <= G64Large::MAXO;
}
static int l64_gh_rows(const nf_flow_desc *desc) {  // rows of the two hidden-cotangent buffers: the widest hidden layer
  int m = 32;
  for (int i = 0; i < desc->n_hidden; ++i) m = desc->hdims[i] > m ? desc->hdims[i] : m;
  return l64_pad32(m);
}
// scratch floats per sample: per COUPLING the nets' stashed layer outputs (a reverse pass that follows a kept forward --
// g64_forward_keep -- reads them back instead of evaluating the nets again, so every coupling has its own slot); shared by
// all couplings: per net the output cotangent and one cotangent buffer per hidden layer (the nets' reverse passes run side by side,
// and the input-cotangent chain of the narrow layers leaves all of them behind at once)
static size_t l64_act_floats_per_sample(const nf_flow_desc *desc) {
  const int c = (desc->d + 1) / 2;
  const int nout = desc->kind == NF_KIND_REALNVP ? c : (3 * desc->K - 1) * c;
  size_t per_net = (size_t)l64_pad32(nout);
  for (int i = 0; i < desc->n_hidden; ++i) per_net += l64_pad32(desc->hdims[i]);
  return (desc->kind == NF_KIND_REALNVP ? 2 : 1) * per_net;
}
static size_t l64_shared_floats_per_sample(const nf_flow_desc *desc) {
  const int c = (desc->d + 1) / 2;
  const int nout = desc->kind == NF_KIND_REALNVP ? c : (3 * desc->K - 1) * c;
  const int nets = desc->kind == NF_KIND_REALNVP ? 2 : 1;
  return nets * ((size_t)l64_pad32(nout) + (size_t)desc->n_hidden * l64_gh_rows(desc));
}
// layout: [shared | slot 0 | slot 1 | ...]; `slots` of them (a stand-alone forward needs one, a kept forward all 2 nlayers)
static size_t l64_scratch_bytes_for(const nf_flow_desc *desc, long N, int slots) {
  const size_t per = (size_t)slots * l64_act_floats_per_sample(desc) + l64_shared_floats_per_sample(desc);
  return ((size_t)((N + 31) / 32) * 32 * per * sizeof(float) + 255) / 256 * 256;
}
size_t nf_l64_scratch_bytes(const nf_flow_desc *desc, long N) {  // what nf_workspace_bytes plans for: the training step's need
  if (!l64_ok(desc)) return 0;
  return l64_scratch_bytes_for(desc, N, 2 * desc->nlayers);
}
struct L64Bufs {
  float *act[2][NF_MAX_HIDDEN + 1];  // [net][layer]: tiled outputs (the last one: the net's output)
  int F[NF_MAX_HIDDEN + 1];          // rows of those tiles
  float *dout[2], *gh[2][NF_MAX_HIDDEN];  // per net: output cotangent; gh[n][l] = cotangent of hidden layer l's outputs (GH rows)
  int nets, nl, GH;
};
// slot: the coupling whose activation area the call uses (flat index; stand-alone calls use any, they do not look back)
static int l64_bufs(nf_ctx *ctx, const nf_flow_desc *desc, const G64Args &a, L64Bufs *b, int slot) {
  const int nc = 2 * desc->nlayers;
  if (slot < 0 || slot >= nc) slot = 0;
  // own (grow-only) buffer: as many slots as this call reaches -- a kept forward starts at the LAST slot, so the buffer has
  // its full size before anything is left in it; a caller's arena is carved once, at the size nf_workspace_bytes planned
  NF_TRY(nf_wimg_reserve(ctx, l64_scratch_bytes_for(desc, a.N, ctx->arena ? nc : slot + 1)));
  ctx->wimg_owner = nullptr;  // the buffer doubles as the fused kernels' packed-image store: whatever it cached is gone
  const size_t Np = (size_t)((a.N + 31) / 32) * 32;
  b->nets = desc->kind == NF_KIND_REALNVP ? 2 : 1;
  b->nl = a.net[0].nl;
  for (int l = 0; l < b->nl; ++l) b->F[l] = l64_pad32(a.net[0].dims[l + 1]);
  // region sizes come from the DESCRIPTOR (the wider of the two masks' couplings: c = ceil(d / 2)), not from this coupling's
  // own widths -- every coupling must find the shared buffers and its slot at the same place
  const int cmax = (desc->d + 1) / 2;
  const size_t out_rows = (size_t)l64_pad32(desc->kind == NF_KIND_REALNVP ? cmax : (3 * desc->K - 1) * cmax);
  float *p = (float *)ctx->wimg;
  for (int n = 0; n < b->nets; ++n) { b->dout[n] = p; p += Np * out_rows; }
  b->GH = l64_gh_rows(desc);
  for (int n = 0; n < b->nets; ++n)
    for (int i = 0; i < desc->n_hidden; ++i) { b->gh[n][i] = p; p += Np * b->GH; }
  p += (size_t)slot * Np * l64_act_floats_per_sample(desc);
  for (int n = 0; n < b->nets; ++n)
    for (int l = 0; l < b->nl; ++l) { b->act[n][l] = p; p += Np * b->F[l]; }
  return NF_OK;
}
static inline unsigned l64_grid(nf_ctx *ctx, long N, long cap) {
  long nb = ((N + 31) / 32 + 3) / 4;
  if (nb > cap) nb = cap;
  return (unsigned)(nb < 1 ? 1 : nb);
}
// block sizes: a layer's inputs are padded to 1, 2, 4 or 8 blocks of 32; its outputs go in groups of 1, 2 or 4 blocks,
// smaller the wider the input (the weight block [32 IB][32 OB] and, for dW, the operand tiles must fit LDS)
static inline int l64_ibp(int nin) { const int b = (nin + 31) / 32; return b <= 1 ? 1 : b <= 2 ? 2 : b <= 4 ? 4 : 8; }
static inline int l64_maxg(int ibp, bool dw) { return ibp <= 2 ? (dw ? 2 : 4) : ibp == 4 ? 2 : 1; }
#define L64_DISPATCH(IBv, OBv, CALL)                                                                  \
  do {                                                                                               \
    switch ((IBv) * 8 + (OBv)) {                                                                     \
      case 8 + 1: CALL(1, 1); break;  case 8 + 2: CALL(1, 2); break;  case 8 + 4: CALL(1, 4); break;  \
      case 16 + 1: CALL(2, 1); break; case 16 + 2: CALL(2, 2); break; case 16 + 4: CALL(2, 4); break; \
      case 32 + 1: CALL(4, 1); break; case 32 + 2: CALL(4, 2); break;                                 \
      default: CALL(8, 1); break;                                                                    \
    }                                                                                                \
  } while (0)
static inline int l64_group(int blocks_left, int maxg) { return blocks_left >= 4 && maxg >= 4 ? 4 : blocks_left >= 2 && maxg >= 2 ? 2 : 1; }

// 2 when the coupling's two nets can share launches: same layer sizes, one constant theta stride, one constant buffer stride
static int l64_nets_merge(const G64Args &a, const L64Bufs &b) {
  constexpr bool off = false;  // (round 6: the NF_L64_NO_NET_MERGE A/B switch is retired -- its question is answered, README "switches")
  if (b.nets != 2 || off || a.net[0].nl != a.net[1].nl) return 1;
  const long dth = a.net[1].w[0] - a.net[0].w[0];
  for (int l = 0; l < a.net[0].nl; ++l) {
    if (a.net[0].dims[l] != a.net[1].dims[l] || a.net[0].dims[l + 1] != a.net[1].dims[l + 1]) return 1;
    if (a.net[1].w[l] - a.net[0].w[l] != dth || a.net[1].b[l] - a.net[0].b[l] != dth) return 1;
  }
  return 2;
}
template <int IB, int OB>
static int l64_fwd_all_launch(nf_ctx *ctx, unsigned grid, size_t lds, const float *theta, const L64Layer &L, int NG, const L64Src &src,
                              float *dst, int Fd, long N, int act, int ny, const L64Y &yy) {
  static AttrOnce attr_once;
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_l64_fwd_all<IB, OB>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    return NF_OK;
  }));
  hipLaunchKernelGGL((k_l64_fwd_all<IB, OB>), dim3(grid, ny), dim3(512), lds, ctx->stream, theta, L, NG, src, dst, Fd, N, act, yy);
  return (int)hipGetLastError();
}
// the fused output-layer kernels of a spline coupling (k_l64_nsf_top_fwd / k_l64_nsf_top_bwd) take: one hidden layer at least,
// <= 64 inputs and <= 384 outputs in the output layer, K <= 8, <= 16 transformed dimensions
static bool l64_top_fusable(const G64Args &a, int last) {
  constexpr bool no_top = false;  // (round 6: the NF_L64_NO_TOP_FUSE A/B switch is retired -- its question is answered, README "switches")
  return !no_top && a.kind == NF_KIND_NSF && last >= 1 && a.K <= 8 && a.c <= 16 && a.net[0].dims[last] <= 64 &&
         a.net[0].dims[last + 1] <= L64_TOP_ROWS;
}
// K = 8: the fused kernels' element arithmetic is nf_rqs_elem.h's (compile-time K)
static bool l64_top_k8(const G64Args &a) {
  constexpr bool off = false;  // (round 6: the NF_L64_TOP_GENERIC_SPLINE A/B switch is retired -- its question is answered, README "switches")
  return a.K == 8 && !off;
}
// the nets of one coupling, layer by layer, on the conditioner half of `x` (standard layout); outputs stay in b->act
static int l64_nets_fwd(nf_ctx *ctx, const nf_flow_desc *desc, const G64Args &a, const float *theta, const float *x, const L64Bufs &b,
                        bool skip_top = false) {  // skip_top: the output layer is the caller's (k_l64_nsf_top_fwd)
  const unsigned grid = l64_grid(ctx, a.N, 4L * ctx->num_cu);
  const int ny = l64_nets_merge(a, b);  // RealNVP: both nets in one launch (blockIdx.y), per-net strides in yy
  for (int n = 0; n < b.nets; n += ny) {
    const G64Net &net = a.net[n];
    // the leading run of narrow layers (widths <= 64) in one launch
    int lc = 0;
    constexpr bool no_chain = false;  // (round 6: the NF_L64_NO_FWD_CHAIN A/B switch is retired -- its question is answered, README "switches")
    const int nlim = net.nl - (skip_top ? 1 : 0);
    while (lc < nlim && net.dims[lc] <= 64 && net.dims[lc + 1] <= 64) ++lc;
    if (lc < 2 || no_chain) lc = 0;
    if (lc) {
      L64Chain ch;
      ch.nl = lc;
      for (int l = 0; l < lc; ++l) {
        ch.w_off[l] = net.w[l]; ch.b_off[l] = net.b[l]; ch.nin[l] = net.dims[l]; ch.nout[l] = net.dims[l + 1];
        ch.act[l] = l < net.nl - 1 ? 1 : 0; ch.F[l] = b.F[l]; ch.dst[l] = b.act[n][l];
      }
      const L64Src src{x, 0, 0, a.d, 1 - a.par_t};
      const L64Y yy{ny == 2 ? a.net[1].w[0] - a.net[0].w[0] : 0, 0, ny == 2 ? b.act[1][0] - b.act[0][0] : 0, 0};
      const size_t lds = (size_t)lc * (64 * (64 + NF_IMG_PAD) + 64) * sizeof(float);
      static AttrOnce attr_once;
      NF_TRY(attr_once.run(ctx->device, [&]() -> int {
        NF_HIP(hipFuncSetAttribute((const void *)k_l64_fwd_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        return NF_OK;
      }));
      const unsigned grid8 = (unsigned)std::min<long>(((a.N + 31) / 32 + 7) / 8, 2L * ctx->num_cu);
      ProfScope ps(ctx, "l64_fwd");
      hipLaunchKernelGGL(k_l64_fwd_chain, dim3(grid8, ny), dim3(512), lds, ctx->stream, theta, ch, src, a.N, yy);
      NF_HIP(hipGetLastError());
    }
    for (int l = lc; l < nlim; ++l) {
      const int nin = net.dims[l], nout = net.dims[l + 1];
      const int IB = l64_ibp(nin), blocks = (nout + 31) / 32;
      L64Src src;
      if (l == 0) src = L64Src{x, 0, 0, a.d, 1 - a.par_t};
      else src = L64Src{b.act[n][l - 1], b.F[l - 1], 0, 0, 0};
      L64Y yy{0, 0, 0, 0};
      if (ny == 2) yy = L64Y{a.net[1].w[l] - a.net[0].w[l], l == 0 ? 0 : b.act[1][l - 1] - b.act[0][l - 1], b.act[1][l] - b.act[0][l], 0};
      const int OBa = l64_maxg(IB, false) >= 4 ? 4 : l64_maxg(IB, false);
      const int NGa = (blocks + OBa - 1) / OBa;
      const size_t lds_all = (size_t)NGa * (32 * IB * (32 * OBa + NF_IMG_PAD) + 32 * OBa) * sizeof(float);
      constexpr bool no_all = false;  // (round 6: the NF_L64_NO_FWD_ALL A/B switch is retired -- its question is answered, README "switches")
      if (NGa >= 2 && lds_all <= 144 * 1024 && !no_all) {  // every output block of a wide layer in one launch
        const L64Layer L{net.w[l], net.b[l], nin, nout, 0};
        const unsigned grid8 = (unsigned)std::min<long>(((a.N + 31) / 32 + 7) / 8, (long)ctx->num_cu);
        ProfScope ps(ctx, "l64_fwd");
#define CALL(I, O) NF_TRY((l64_fwd_all_launch<I, O>(ctx, grid8, lds_all, theta, L, NGa, src, b.act[n][l], b.F[l], a.N, l < net.nl - 1 ? 1 : 0, ny, yy)))
        L64_DISPATCH(IB, OBa, CALL);
#undef CALL
      } else
      for (int ob0 = 0; ob0 < blocks;) {
        const int OB = l64_group(blocks - ob0, l64_maxg(IB, false));
        const L64Layer L{net.w[l], net.b[l], nin, nout, 32 * ob0};
        ProfScope ps(ctx, "l64_fwd");
#define CALL(I, O) hipLaunchKernelGGL((k_l64_fwd<I, O>), dim3(grid, ny), dim3(256), 0, ctx->stream, theta, L, src, b.act[n][l], b.F[l], a.N, l < net.nl - 1 ? 1 : 0, yy)
        L64_DISPATCH(IB, OB, CALL);
#undef CALL
        NF_HIP(hipGetLastError());
        ob0 += OB;
      }
    }
  }
  return NF_OK;
}
static int l64_apply(nf_ctx *ctx, const nf_flow_desc *desc, const G64Args &a, int inverse, const float *theta, float *xy, float *ladj,
                     int slot) {
  L64Bufs b;
  NF_TRY(l64_bufs(ctx, desc, a, &b, slot));
  const int last = b.nl - 1;
  if (l64_top_fusable(a, last)) {  // a spline coupling's output layer and the spline in one kernel
    NF_TRY(l64_nets_fwd(ctx, desc, a, theta, xy, b, true));
    const G64Net &net = a.net[0];
    const L64Layer L{net.w[last], net.b[last], net.dims[last], net.dims[last + 1], 0};
    static AttrOnce attr_once;
    NF_TRY(attr_once.run(ctx->device, [&]() -> int {
      NF_HIP(hipFuncSetAttribute((const void *)k_l64_nsf_top_fwd<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, L64_TOPF_LDS));
      NF_HIP(hipFuncSetAttribute((const void *)k_l64_nsf_top_fwd<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, L64_TOPF_LDS));
      return NF_OK;
    }));
    const long pairs = ((a.N + 31) / 32 + 1) / 2;
    const unsigned gridp = (unsigned)std::min<long>(pairs, (long)ctx->num_cu);
    ProfScope ps(ctx, "l64_top_fwd");
    if (l64_top_k8(a))
      hipLaunchKernelGGL((k_l64_nsf_top_fwd<8, 8>), dim3(gridp), dim3(512), L64_TOPF_LDS, ctx->stream, a, inverse, theta, L,
                         (const float *)b.act[0][last - 1], b.F[last - 1], b.act[0][last], b.F[last], xy, ladj, (long long *)ctx->trace);
    else
      hipLaunchKernelGGL((k_l64_nsf_top_fwd<8, 0>), dim3(gridp), dim3(512), L64_TOPF_LDS, ctx->stream, a, inverse, theta, L,
                         (const float *)b.act[0][last - 1], b.F[last - 1], b.act[0][last], b.F[last], xy, ladj, (long long *)ctx->trace);
    return (int)hipGetLastError();
  }
  NF_TRY(l64_nets_fwd(ctx, desc, a, theta, xy, b));
  ProfScope ps(ctx, "l64_couple");
  if (a.K <= 8)
    hipLaunchKernelGGL(k_l64_couple_fwd<8>, dim3((unsigned)((a.N + 31) / 32)), dim3(256), 0, ctx->stream, a, inverse,
                       (const float *)b.act[0][last], b.F[last], (const float *)b.act[b.nets - 1][last], b.F[last], xy, ladj);
  else
    hipLaunchKernelGGL(k_l64_couple_fwd<G64_MAXK>, dim3((unsigned)((a.N + 31) / 32)), dim3(256), 0, ctx->stream, a, inverse,
                       (const float *)b.act[0][last], b.F[last], (const float *)b.act[b.nets - 1][last], b.F[last], xy, ladj);
  return (int)hipGetLastError();
}
template <int IB, int OB>
static int l64_dw_launch(nf_ctx *ctx, unsigned grid, const L64Layer &L, const L64Src &av, const L64Src &g, const float *act, int Fa, long N,
                         float *slabs, long Pc, long slab_off, int ny, const L64Y &yy) {
  const size_t lds = (size_t)4 * (IB + OB) * 32 * NF_TS * sizeof(float);
  static AttrOnce attr_once;
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_l64_dw<IB, OB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  hipLaunchKernelGGL((k_l64_dw<IB, OB>), dim3(grid, ny), dim3(256), lds, ctx->stream, L, av, g, act, Fa, N, slabs, Pc, slab_off, yy);
  return (int)hipGetLastError();
}
template <int IB, int OB>
static int l64_dw_cols_launch(nf_ctx *ctx, unsigned grid, const L64Layer &L, const L64Src &av, const L64Src &g, const float *act, int Fa, long N,
                              float *slabs, long Pc, long slab_off, int ny, const L64Y &yy) {
  const size_t lds = (size_t)(IB + 4 * OB) * 32 * NF_TS * sizeof(float);
  static AttrOnce attr_once;
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_l64_dw_cols<IB, OB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  hipLaunchKernelGGL((k_l64_dw_cols<IB, OB>), dim3(grid, ny), dim3(256), lds, ctx->stream, L, av, g, act, Fa, N, slabs, Pc, slab_off, yy);
  return (int)hipGetLastError();
}
template <int IB, int OB>
static int l64_bwdx_all_launch(nf_ctx *ctx, unsigned grid, size_t lds, const float *theta, const L64Layer &L, int NG, const L64Src &g,
                               const float *act, int Fa, float *dst, int Fd, int xd, int xpar, long N, int ny, const L64Y &yy) {
  static AttrOnce attr_once;
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_l64_bwdx_all<IB, OB>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    return NF_OK;
  }));
  hipLaunchKernelGGL((k_l64_bwdx_all<IB, OB>), dim3(grid, ny), dim3(512), lds, ctx->stream, theta, L, NG, g, act, Fa, dst, Fd, xd, xpar, N, yy);
  return (int)hipGetLastError();
}
// reverse pass of coupling k at x (forward: its input; inv: the inverse's output): gbar updated in place, this coupling's
// parameter gradient in g[theta_off ...] through per-workgroup slabs
static int l64_bwd(nf_ctx *ctx, const nf_flow_desc *desc, int k, const G64Args &a, int inv, const float *theta, const float *x, float *gbar,
                   const float *lbar, float lbar_const, float *g, float *slabs, bool kept) {
  const CouplingInfo ci = nf_coupling_info(desc, k);
  L64Bufs b;
  NF_TRY(l64_bufs(ctx, desc, a, &b, k));
  // kept: the forward of this very pass (g64_forward_keep) left this coupling's layer outputs in its slot
  if (!kept) NF_TRY(l64_nets_fwd(ctx, desc, a, theta, x, b));
  const int last = b.nl - 1;
  const unsigned grid = l64_grid(ctx, a.N, 4L * ctx->num_cu);
  unsigned gridw = l64_grid(ctx, a.N, (long)g64_bwd_blocks(desc, a.N));
  // a spline coupling's output layer: the spline's reverse, dW and the input cotangent in one kernel (k_l64_nsf_top_bwd)
  const bool top_fused = l64_top_fusable(a, last);
  if (top_fused) {
    // one workgroup (= one slab) per CU: the fused kernel stages the whole layer per workgroup and writes a 96 KB slab; the
    // coupling's other dW kernels share the count (measured against two per CU: step 3.23 -> 3.17 ms at the docstring shape)
    if (gridw > (unsigned)ctx->num_cu) gridw = (unsigned)ctx->num_cu;
    const G64Net &net = a.net[0];
    const L64Layer L{net.w[last], net.b[last], net.dims[last], net.dims[last + 1], 0};
    static AttrOnce attr_once;
    NF_TRY(attr_once.run(ctx->device, [&]() -> int {
      NF_HIP(hipFuncSetAttribute((const void *)k_l64_nsf_top_bwd<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, L64_TOP_LDS));
      NF_HIP(hipFuncSetAttribute((const void *)k_l64_nsf_top_bwd<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, L64_TOP_LDS));
      return NF_OK;
    }));
    {
      ProfScope ps(ctx, "l64_top_bwd");
      if (l64_top_k8(a))
        hipLaunchKernelGGL((k_l64_nsf_top_bwd<8, 8>), dim3(gridw), dim3(512), L64_TOP_LDS, ctx->stream, a, inv, theta, L,
                           (const float *)b.act[0][last], b.F[last], (const float *)b.act[0][last - 1], b.F[last - 1], x, gbar, lbar,
                           lbar_const, b.gh[0][last - 1], b.GH, slabs, (long)ci.nparams, (long)ci.theta_off, (long long *)ctx->trace);
      else
        hipLaunchKernelGGL((k_l64_nsf_top_bwd<8, 0>), dim3(gridw), dim3(512), L64_TOP_LDS, ctx->stream, a, inv, theta, L,
                           (const float *)b.act[0][last], b.F[last], (const float *)b.act[0][last - 1], b.F[last - 1], x, gbar, lbar,
                           lbar_const, b.gh[0][last - 1], b.GH, slabs, (long)ci.nparams, (long)ci.theta_off, (long long *)ctx->trace);
      NF_HIP(hipGetLastError());
    }
    // the layers below it in one more launch when there are one or two, none wider than 64 (k_l64_hidden_bwd)
    constexpr bool no_hid = false;  // (round 6: the NF_L64_NO_HIDDEN_FUSE A/B switch is retired -- its question is answered, README "switches")
    bool narrow = !no_hid && last <= 2 && net.dims[0] <= 64;
    for (int l = 0; l < last; ++l) narrow = narrow && net.dims[l + 1] <= 64;
    if (narrow) {
      L64Hid hd;
      for (int l = 0; l < 2; ++l) {
        const int q = l < last ? l : 0;
        hd.w_off[l] = net.w[q]; hd.b_off[l] = net.b[q]; hd.nin[l] = net.dims[q]; hd.nout[l] = net.dims[q + 1]; hd.F[l] = b.F[q];
        hd.act[l] = b.act[0][q];
      }
      const L64Src xin{x, 0, 0, a.d, 1 - a.par_t};
      const int ib0 = net.dims[0] <= 32 ? 1 : 2;
      ProfScope ps2(ctx, "l64_hidden_bwd");
#define CALLH(NHv, IBv)                                                                                                                  \
  do {                                                                                                                                   \
    static AttrOnce once;                                                                                                                \
    NF_TRY(once.run(ctx->device, [&]() -> int {                                                                                          \
      NF_HIP(hipFuncSetAttribute((const void *)k_l64_hidden_bwd<NHv, IBv>, hipFuncAttributeMaxDynamicSharedMemorySize, L64_HID_LDS(NHv))); \
      return NF_OK;                                                                                                                      \
    }));                                                                                                                                 \
    hipLaunchKernelGGL((k_l64_hidden_bwd<NHv, IBv>), dim3(gridw), dim3(256), L64_HID_LDS(NHv), ctx->stream, theta, hd,                   \
                       (const float *)b.gh[0][last - 1], b.GH, xin, gbar, a.N, slabs, (long)ci.nparams, (long)ci.theta_off,              \
                       (long long *)ctx->trace);                                                                                         \
  } while (0)
      if (last == 2 && ib0 == 2) CALLH(2, 2);
      else if (last == 2) CALLH(2, 1);
      else if (ib0 == 2) CALLH(1, 2);
      else CALLH(1, 1);
#undef CALLH
      NF_HIP(hipGetLastError());
      return nf_launch_reduce_slabs(ctx, NF_DTYPE_F32, slabs, (int)gridw, ci.nparams, g + ci.theta_off);
    }
  } else {
    ProfScope ps(ctx, "l64_couple");
    if (a.K <= 8)
      hipLaunchKernelGGL(k_l64_couple_bwd<8>, dim3((unsigned)((a.N + 31) / 32)), dim3(256), 0, ctx->stream, a, inv,
                         (const float *)b.act[0][last], b.F[last], x, gbar, lbar, lbar_const, b.dout[0], b.dout[b.nets - 1], b.F[last]);
    else
      hipLaunchKernelGGL(k_l64_couple_bwd<G64_MAXK>, dim3((unsigned)((a.N + 31) / 32)), dim3(256), 0, ctx->stream, a, inv,
                         (const float *)b.act[0][last], b.F[last], x, gbar, lbar, lbar_const, b.dout[0], b.dout[b.nets - 1], b.F[last]);
    NF_HIP(hipGetLastError());
  }
  const int nym = l64_nets_merge(a, b);
  for (int n = 0; n < b.nets; n += nym) {
    const G64Net &net = a.net[n];
    // layers 1 .. ct are narrow (widths <= 64): their input cotangents in one launch (k_l64_bwdx_chain)
    int ct = 0;
    constexpr bool no_bchain = false;  // (round 6: the NF_L64_NO_BWDX_CHAIN A/B switch is retired -- its question is answered, README "switches")
    while (ct + 1 < net.nl && net.dims[ct + 1] <= 64 && net.dims[ct + 2] <= 64) ++ct;
    if (top_fused && ct == net.nl - 1) --ct;  // the output layer's input cotangent is already there
    if (ct < 2 || no_bchain) ct = 0;
    for (int l = net.nl - 1 - (top_fused ? 1 : 0); l >= 0; --l) {
      const int nin = net.dims[l], nout = net.dims[l + 1];
      const int IB = l64_ibp(nin), blocks = (nout + 31) / 32;
      // delta of this layer's outputs: the net output's cotangent as is; a hidden layer's through leaky-ReLU'
      const bool top = l == net.nl - 1;
      const float *gsrc = top ? b.dout[n] : b.gh[n][l];
      const int Fg = top ? b.F[last] : b.GH;
      const float *act = top ? nullptr : b.act[n][l];
      float *gdst = l > 0 ? b.gh[n][l - 1] : nullptr;  // cotangent of this layer's inputs = of layer l - 1's outputs
      L64Src av;
      if (l == 0) av = L64Src{x, 0, 0, a.d, 1 - a.par_t};
      else av = L64Src{b.act[n][l - 1], b.F[l - 1], 0, 0, 0};
      // per-net strides (blockIdx.y): dW kernels (a, g, act), input-cotangent kernels (g, act, dst)
      const long dth = nym == 2 ? a.net[1].w[l] - a.net[0].w[l] : 0;
      const long dgs = nym == 2 ? (top ? b.dout[1] - b.dout[0] : b.gh[1][0] - b.gh[0][0]) : 0;
      const long dac = nym == 2 && !top ? b.act[1][l] - b.act[0][l] : 0;
      const L64Y yw{dth, nym == 2 && l > 0 ? b.act[1][l - 1] - b.act[0][l - 1] : 0, dgs, dac};
      const L64Y yx{dth, dgs, dac, nym == 2 ? b.gh[1][0] - b.gh[0][0] : 0};
      const L64Y y0{0, 0, 0, 0};
      // the first layer's input cotangent is ADDED to the one conditioner half both nets share: one net after the other
      const int nyx = l == 0 ? 1 : nym;
      // a layer wider than one launch of k_l64_dw takes: all its columns at once, a column range per wave
      const int OBc = (blocks + 3) / 4;
      constexpr bool no_cols = false;  // (round 6: the NF_L64_NO_DW_COLS A/B switch is retired -- its question is answered, README "switches")
      if (blocks > l64_maxg(IB, true) && IB * OBc <= 8 && OBc <= 4 && !no_cols) {
        const L64Layer L{net.w[l], net.b[l], nin, nout, 0};
        const L64Src gs{gsrc, Fg, 0, 0, 0};
        ProfScope ps(ctx, "l64_dw");
#define CALLC(I, O) NF_TRY((l64_dw_cols_launch<I, O>(ctx, gridw, L, av, gs, act, b.F[l], a.N, slabs, ci.nparams, ci.theta_off, nym, yw)))
        switch (IB * 8 + OBc) {
          case 8 + 1: CALLC(1, 1); break;  case 8 + 2: CALLC(1, 2); break;  case 8 + 3: CALLC(1, 3); break;  case 8 + 4: CALLC(1, 4); break;
          case 16 + 1: CALLC(2, 1); break; case 16 + 2: CALLC(2, 2); break; case 16 + 3: CALLC(2, 3); break; case 16 + 4: CALLC(2, 4); break;
          case 32 + 1: CALLC(4, 1); break; case 32 + 2: CALLC(4, 2); break; case 64 + 1: CALLC(8, 1); break;
          default: return NF_ERR_UNSUPPORTED;  // (unreachable: IB in {1, 2, 4, 8}, IB * OBc <= 8)
        }
#undef CALLC
      } else
      for (int ob0 = 0; ob0 < blocks;) {
        const int OBw = l64_group(blocks - ob0, l64_maxg(IB, true));
        const L64Layer L{net.w[l], net.b[l], nin, nout, 32 * ob0};
        const L64Src gs{gsrc, Fg, 32 * ob0, 0, 0};
        ProfScope ps(ctx, "l64_dw");
#define CALL(I, O) NF_TRY((l64_dw_launch<I, O>(ctx, gridw, L, av, gs, act, b.F[l], a.N, slabs, ci.nparams, ci.theta_off, nym, yw)))
        L64_DISPATCH(IB, OBw, CALL);
#undef CALL
        ob0 += OBw;
      }
      if (ct && l >= 1 && l <= ct) {
        if (l == ct) {  // the chain, once, from its highest layer down to layer 1
          L64BChain ch;
          ch.nl = ct;
          for (int q = 0; q < ct; ++q) {
            const int lq = ct - q;
            ch.w_off[q] = net.w[lq]; ch.nin[q] = net.dims[lq]; ch.nout[q] = net.dims[lq + 1];
            ch.act[q] = lq == net.nl - 1 ? nullptr : b.act[n][lq]; ch.Fa[q] = b.F[lq]; ch.dst[q] = b.gh[n][lq - 1];
          }
          const L64Src gtop{gsrc, Fg, 0, 0, 0};
          const L64Y yc{dth, dgs, nym == 2 ? b.act[1][0] - b.act[0][0] : 0, nym == 2 ? b.gh[1][0] - b.gh[0][0] : 0};
          const size_t lds = (size_t)ct * (64 * (64 + NF_IMG_PAD) + 64) * sizeof(float);
          static AttrOnce attr_once;
          NF_TRY(attr_once.run(ctx->device, [&]() -> int {
            NF_HIP(hipFuncSetAttribute((const void *)k_l64_bwdx_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            return NF_OK;
          }));
          const unsigned grid8 = (unsigned)std::min<long>(((a.N + 31) / 32 + 7) / 8, 2L * ctx->num_cu);
          ProfScope ps(ctx, "l64_bwdx");
          hipLaunchKernelGGL(k_l64_bwdx_chain, dim3(grid8, nym), dim3(512), lds, ctx->stream, theta, ch, gtop, b.GH, a.N, yc);
          NF_HIP(hipGetLastError());
        }
        continue;
      }
      // every output block in one launch when the layer's weight blocks fit LDS side by side
      const int OBa = l64_maxg(IB, false) >= 4 ? 4 : l64_maxg(IB, false);
      const int NGa = (blocks + OBa - 1) / OBa;
      const size_t lds_all = (size_t)NGa * (32 * IB * (32 * OBa + NF_IMG_PAD) + 32 * OBa) * sizeof(float);
      constexpr bool no_all = false;  // (round 6: the NF_L64_NO_BWDX_ALL A/B switch is retired -- its question is answered, README "switches")
      for (int m = 0; m < (l == 0 ? nym : 1); ++m) {  // l == 0: net by net (nyx = 1); else one launch for both (m = 0 only)
        const G64Net &nm = a.net[n + m];
        const float *gsrc_m = top ? b.dout[n + m] : b.gh[n + m][l];
        const float *act_m = top ? nullptr : b.act[n + m][l];
        float *gdst_m = l > 0 ? b.gh[n + m][l - 1] : nullptr;
        const L64Y &yq = l == 0 ? y0 : yx;
        if (NGa >= 2 && lds_all <= 144 * 1024 && !no_all) {
          const L64Layer L{nm.w[l], nm.b[l], nin, nout, 0};
          const L64Src gs{gsrc_m, Fg, 0, 0, 0};
          const unsigned grid8 = (unsigned)std::min<long>(((a.N + 31) / 32 + 7) / 8, (long)ctx->num_cu);
          ProfScope ps(ctx, "l64_bwdx");
#define CALL(I, O) NF_TRY((l64_bwdx_all_launch<I, O>(ctx, grid8, lds_all, theta, L, NGa, gs, act_m, b.F[l], l == 0 ? gbar : gdst_m, \
                                                       l == 0 ? 0 : b.GH, l == 0 ? a.d : 0, l == 0 ? 1 - a.par_t : 0, a.N, nyx, yq)))
          L64_DISPATCH(IB, OBa, CALL);
#undef CALL
        } else
        for (int ob0 = 0, first = 1; ob0 < blocks; first = 0) {
          const int OB = l64_group(blocks - ob0, l64_maxg(IB, false));
          const L64Layer L{nm.w[l], nm.b[l], nin, nout, 32 * ob0};
          const L64Src gs{gsrc_m, Fg, 32 * ob0, 0, 0};
          ProfScope ps(ctx, "l64_bwdx");
          if (l == 0) {
#define CALL(I, O) hipLaunchKernelGGL((k_l64_bwdx<I, O>), dim3(grid, nyx), dim3(256), 0, ctx->stream, theta, L, gs, act_m, b.F[l], gbar, 0, 1, a.d, 1 - a.par_t, a.N, yq)
            L64_DISPATCH(IB, OB, CALL);
#undef CALL
          } else {
#define CALL(I, O) hipLaunchKernelGGL((k_l64_bwdx<I, O>), dim3(grid, nyx), dim3(256), 0, ctx->stream, theta, L, gs, act_m, b.F[l], gdst_m, b.GH, first ? 0 : 1, 0, 0, a.N, yq)
            L64_DISPATCH(IB, OB, CALL);
#undef CALL
          }
          NF_HIP(hipGetLastError());
          ob0 += OB;
        }
      }
    }
  }
  return nf_launch_reduce_slabs(ctx, NF_DTYPE_F32, slabs, (int)gridw, ci.nparams, g + ci.theta_off);
}

// couplings [layer_lo, layer_hi) in flat order (forward: applied last-listed first); y may alias x
int nf_g64_apply(nf_ctx *ctx, const nf_flow_desc *desc, int layer_lo, int layer_hi, bool inverse, const void *theta,
                 const void *x, long N, void *y, void *ladj) {
  if (N <= 0) return NF_OK;
  const bool f64 = desc->dtype == NF_DTYPE_F64;
  const size_t es = f64 ? 8 : 4;
  NF_HIP(hipMemsetAsync(ladj, 0, (size_t)N * es, ctx->stream));
  if (y != x) NF_HIP(hipMemcpyAsync(y, x, (size_t)N * desc->d * es, hipMemcpyDeviceToDevice, ctx->stream));
  const unsigned grid = (unsigned)((N + G64_BLOCK - 1) / G64_BLOCK);
  for (int s = 0; s < layer_hi - layer_lo; ++s) {
    const int k = inverse ? layer_lo + s : layer_hi - 1 - s;
    const G64Args a = make_g64_args(desc, k, N);
    ProfScope ps(ctx, "g64_apply");
    if (f64)
      NF_TRY(g64_launch_apply<double>(ctx, desc, grid, a, inverse ? 1 : 0, (const double *)theta, (const double *)y, (double *)y, (double *)ladj));
    else
      NF_TRY(g64_launch_apply<float>(ctx, desc, grid, a, inverse ? 1 : 0, (const float *)theta, (const float *)y, (float *)y, (float *)ladj));
  }
  return NF_OK;
}

// workspace: the input of every coupling (nc * N * d doubles)
size_t nf_g64_bwd_ws_bytes(const nf_flow_desc *desc, long N) {
  const size_t a = (size_t)2 * desc->nlayers * (size_t)N * desc->d * sizeof(double) + (size_t)N * sizeof(double);  // sized for f64
  return (a + 255) / 256 * 256 + g64_slab_bytes(desc, N);
}

// x = flow input; ybar -> xbar_out (may alias), gtheta_out <- dL/dtheta
// the chain from x with the input of coupling k kept in inputs[k] (execution order k = nc-1 ... 0); the last coupling's
// output goes to y_last, or is not computed when that is null; ladj accumulates (zeroed by the caller)
template <class T>
static int g64_forward_keep(nf_ctx *ctx, const nf_flow_desc *desc, const T *theta, const T *x, long N, T *inputs, T *y_last, T *ladj) {
  const int nc = 2 * desc->nlayers;
  const size_t nd = (size_t)N * desc->d;
  const unsigned grid = (unsigned)((N + G64_BLOCK - 1) / G64_BLOCK);
  const T *cur = x;
  for (int k = nc - 1; k >= 0; --k) {
    T *slot = inputs + (size_t)k * nd;
    if (cur != slot) NF_HIP(hipMemcpyAsync(slot, cur, nd * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    T *next = k > 0 ? inputs + (size_t)(k - 1) * nd : y_last;
    if (next) {
      const G64Args a = make_g64_args(desc, k, N);
      ProfScope ps(ctx, "g64_apply");
      NF_TRY(g64_launch_apply<T>(ctx, desc, grid, a, 0, theta, (const T *)slot, next, ladj, k));
      cur = next;
    }
  }
  return NF_OK;
}

// the whole flow forward, keeping every coupling's input in `ws` in nf_g64_bwd's layout: nf_g64_bwd(x = nullptr) on the
// same ws is then the reverse pass of THIS forward (no second forward from x).  y must not alias ws.
int nf_g64_apply_keep(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x, long N, void *y, void *ladj, void *ws) {
  if (N <= 0) return NF_OK;
  const size_t es = desc->dtype == NF_DTYPE_F64 ? 8 : 4;
  NF_HIP(hipMemsetAsync(ladj, 0, (size_t)N * es, ctx->stream));
  if (desc->dtype == NF_DTYPE_F64)
    return g64_forward_keep<double>(ctx, desc, (const double *)theta, (const double *)x, N, (double *)ws, (double *)y, (double *)ladj);
  return g64_forward_keep<float>(ctx, desc, (const float *)theta, (const float *)x, N, (float *)ws, (float *)y, (float *)ladj);
}

template <class T>
static int g64_bwd_t(nf_ctx *ctx, const nf_flow_desc *desc, const T *theta, const T *x, const T *ybar, const T *lbar,
                     double lbar_const, long N, T *xbar_out, T *gtheta_out, void *ws) {
  const int nc = 2 * desc->nlayers;
  const size_t nd = (size_t)N * desc->d;
  T *inputs = (T *)ws;
  T *scr_ladj = inputs + (size_t)nc * nd;
  T *slabs = (T *)((char *)ws + (((size_t)nc * nd * sizeof(double) + (size_t)N * sizeof(double)) + 255) / 256 * 256);
  const CouplingInfo last = nf_coupling_info(desc, nc - 1);
  const long P = last.theta_off + last.nparams;
  NF_HIP(hipMemsetAsync(gtheta_out, 0, (size_t)P * sizeof(T), ctx->stream));
  if (N <= 0) return NF_OK;
  // forward, keeping the input of each coupling: execution order k = nc-1 ... 0
  // (x == nullptr: nf_g64_apply_keep left them there -- the training step's own forward is the tape)
  if (x) {
    NF_HIP(hipMemsetAsync(scr_ladj, 0, (size_t)N * sizeof(T), ctx->stream));
    NF_TRY(g64_forward_keep<T>(ctx, desc, theta, x, N, inputs, nullptr, scr_ladj));
  }
  if (xbar_out != ybar) NF_HIP(hipMemcpyAsync(xbar_out, ybar, nd * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
  for (int k = 0; k < nc; ++k) {  // reverse of execution order
    const G64Args a = make_g64_args(desc, k, N);
    ProfScope ps(ctx, "g64_bwd");
    // the forward above (or nf_g64_apply_keep) evaluated coupling k's nets unless it stopped short of the last coupling
    const bool kept = k > 0 || x == nullptr;
    NF_TRY(g64_launch_bwd<T>(ctx, desc, k, a, 0, theta, (const T *)(inputs + (size_t)k * nd), xbar_out, lbar, (T)lbar_const, gtheta_out, slabs, kept));
  }
  return NF_OK;
}

int nf_g64_bwd(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x, const void *ybar,
               const void *lbar, double lbar_const, long N, void *xbar_out, void *gtheta_out, void *ws) {
  if (desc->dtype == NF_DTYPE_F64)
    return g64_bwd_t<double>(ctx, desc, (const double *)theta, (const double *)x, (const double *)ybar, (const double *)lbar,
                             lbar_const, N, (double *)xbar_out, (double *)gtheta_out, ws);
  return g64_bwd_t<float>(ctx, desc, (const float *)theta, (const float *)x, (const float *)ybar, (const float *)lbar,
                          lbar_const, N, (float *)xbar_out, (float *)gtheta_out, ws);
}

// Reverse pass of the INVERSE chain (forward-KL training).  `z` holds T^-1(data) on entry and the data
// again on exit; `gbar` the cotangent of z on entry; lbar_const the cotangent of every sample's
// ladj_inv.  Couplings are walked in forward execution order: reverse pass of inverse coupling k at z,
// then z <- coupling_k(z).  ws: N elements (scratch log-det).
template <class T>
static int g64_bwd_inv_t(nf_ctx *ctx, const nf_flow_desc *desc, const T *theta, T *z, T *gbar, double lbar_const, long N,
                         T *gtheta_out, void *ws) {
  const int nc = 2 * desc->nlayers;
  const CouplingInfo last = nf_coupling_info(desc, nc - 1);
  const long P = last.theta_off + last.nparams;
  NF_HIP(hipMemsetAsync(gtheta_out, 0, (size_t)P * sizeof(T), ctx->stream));
  if (N <= 0) return NF_OK;
  T *scr_ladj = (T *)ws;
  T *slabs = (T *)((char *)ws + ((size_t)N * sizeof(double) + 255) / 256 * 256);
  NF_HIP(hipMemsetAsync(scr_ladj, 0, (size_t)N * sizeof(T), ctx->stream));
  const unsigned grid = (unsigned)((N + G64_BLOCK - 1) / G64_BLOCK);
  for (int k = nc - 1; k >= 0; --k) {
    const G64Args a = make_g64_args(desc, k, N);
    {
      ProfScope ps(ctx, "g64_bwd");
      NF_TRY(g64_launch_bwd<T>(ctx, desc, k, a, 1, theta, (const T *)z, gbar, (const T *)nullptr, (T)lbar_const, gtheta_out, slabs));
    }
    NF_TRY(g64_launch_apply<T>(ctx, desc, grid, a, 0, theta, (const T *)z, z, scr_ladj));
  }
  return NF_OK;
}

size_t nf_g64_bwd_inv_ws_bytes(const nf_flow_desc *desc, long N) {
  return ((size_t)N * sizeof(double) + 255) / 256 * 256 + g64_slab_bytes(desc, N);
}

int nf_g64_bwd_inv(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, void *z, void *gbar, double lbar_const, long N,
                   void *gtheta_out, void *ws) {
  if (desc->dtype == NF_DTYPE_F64)
    return g64_bwd_inv_t<double>(ctx, desc, (const double *)theta, (double *)z, (double *)gbar, lbar_const, N,
                                 (double *)gtheta_out, ws);
  return g64_bwd_inv_t<float>(ctx, desc, (const float *)theta, (float *)z, (float *)gbar, lbar_const, N, (float *)gtheta_out, ws);
}
