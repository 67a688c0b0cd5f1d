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
    for (int j = 0; j < nout; ++j) {
      T s = b[j];
      for (int i = 0; i < nin; ++i) s += W[(long)i * nout + j] * cur[i];
      dst[j] = (l < n.nl - 1) ? g64_lrelu(s) : s;
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

// launches of the two kernels in the size class the flow fits
template <class T>
static void g64_launch_apply(nf_ctx *ctx, const nf_flow_desc *desc, unsigned grid, const G64Args &a, int inverse, const T *theta,
                             const T *x, T *y, T *ladj) {
  if (g64_fits<G64Small>(desc))
    hipLaunchKernelGGL((k_g64_apply<T, G64Small>), dim3(grid), dim3(G64_BLOCK), 0, ctx->stream, a, inverse, theta, x, y, ladj);
  else
    hipLaunchKernelGGL((k_g64_apply<T, G64Large>), dim3(grid), dim3(G64_BLOCK), 0, ctx->stream, a, inverse, theta, x, y, ladj);
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
                          const T *x, T *gbar, const T *lbar, T lbar_const, T *g, T *slabs) {
  const CouplingInfo ci = nf_coupling_info(desc, k);
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
      g64_launch_apply<double>(ctx, desc, grid, a, inverse ? 1 : 0, (const double *)theta, (const double *)y, (double *)y, (double *)ladj);
    else
      g64_launch_apply<float>(ctx, desc, grid, a, inverse ? 1 : 0, (const float *)theta, (const float *)y, (float *)y, (float *)ladj);
    NF_HIP(hipGetLastError());
  }
  return NF_OK;
}

// workspace: the input of every coupling (nc * N * d doubles)
size_t nf_g64_bwd_ws_bytes(const nf_flow_desc *desc, long N) {
  const size_t a = (size_t)2 * desc->nlayers * (size_t)N * desc->d * sizeof(double) + (size_t)N * sizeof(double);  // sized for f64
  return (a + 255) / 256 * 256 + g64_slab_bytes(desc, N);
}

// x = flow input; ybar -> xbar_out (may alias), gtheta_out <- dL/dtheta
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
  const unsigned grid = (unsigned)((N + G64_BLOCK - 1) / G64_BLOCK);
  // forward, keeping the input of each coupling: execution order k = nc-1 ... 0
  NF_HIP(hipMemsetAsync(scr_ladj, 0, (size_t)N * sizeof(T), ctx->stream));
  const T *cur = x;
  for (int k = nc - 1; k >= 0; --k) {
    T *slot = inputs + (size_t)k * nd;
    if (cur != slot) NF_HIP(hipMemcpyAsync(slot, cur, nd * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    if (k > 0) {
      T *next = inputs + (size_t)(k - 1) * nd;
      const G64Args a = make_g64_args(desc, k, N);
      g64_launch_apply<T>(ctx, desc, grid, a, 0, theta, (const T *)slot, next, scr_ladj);
      NF_HIP(hipGetLastError());
      cur = next;
    }
  }
  if (xbar_out != ybar) NF_HIP(hipMemcpyAsync(xbar_out, ybar, nd * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
  for (int k = 0; k < nc; ++k) {  // reverse of execution order
    const G64Args a = make_g64_args(desc, k, N);
    ProfScope ps(ctx, "g64_bwd");
    NF_TRY(g64_launch_bwd<T>(ctx, desc, k, a, 0, theta, (const T *)(inputs + (size_t)k * nd), xbar_out, lbar, (T)lbar_const, gtheta_out, slabs));
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
    g64_launch_apply<T>(ctx, desc, grid, a, 0, theta, (const T *)z, z, scr_ladj);
    NF_HIP(hipGetLastError());
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
