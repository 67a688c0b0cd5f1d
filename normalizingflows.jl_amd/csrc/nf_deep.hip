// nf_deep.hip -- LDS-resident RealNVP couplings whose conditioner nets have 1, 3 or 4 hidden layers (round 5).
//
// Reference: fnn(input_dim, hidden_dims, output_dim; ...) accepts any number of hidden layers (src/flows/utils.jl:71-100) and
// realnvp(q0, hdims, nlayers) passes `hdims` straight through (src/flows/realnvp.jl:166-180); the coupling arithmetic is
// AffineCoupling's (src/flows/realnvp.jl:57-110).  The fused kernels of nf_coupling.hip are written for exactly two hidden
// layers (three GEMMs, two activations in named registers, a stash layout for them); every other depth used to run the
// general path of nf_generic64.hip -- one launch per layer, activations through HBM: 3.09 ms per step for d = 64,
// hidden [64, 64, 64], 65 536 samples against 0.58 ms for the two-hidden-layer shape ("missing").
//
// Here: the same design as nf_coupling.hip's chain / reverse kernels -- tiled batch layout, one wavefront per 32-sample tile,
// activations chained through the fp32 matrix pipe in registers (nf_mfma.h), weights as padded fp32 images in LDS, masks as
// index arithmetic through buffer descriptors -- with the depth a template parameter:
//   DeepGeo<NH, HB>   NH hidden layers, every hidden width padded to 32 HB (HB = 1, 2); d <= 64, so the conditioner and the
//                     transformed half are one 32-row block each.
//   k_deep_chain      all couplings (or one) forward / inverse in one launch; a workgroup = 8 tiles; both nets of the current
//                     coupling in LDS, restaged per coupling.
//   k_deep_bwd        reverse pass of all couplings (or one) in one launch with invertible recompute (x1 = (y1 - T) exp(-S),
//                     src/flows/realnvp.jl:107): per coupling two phases (t-net, then s-net), one net in LDS, the dW^T
//                     accumulators of all its layers in registers, a layer's operands transposed through a per-wave LDS tile
//                     just before its dW GEMM; gradient slabs in image layout, reduced in a fixed order (deterministic).
// The host functions are reached through nf_coupling.hip's nf_affine_* entry points (a deep flow is "a resident RealNVP
// without a stash" to nf_api.hip), so every API path -- forward, inverse, rand, ELBO, training step, forward-KL, pullbacks,
// compositions -- takes these kernels.
#include <cstdlib>

#include "nf_common.h"
#include "nf_mfma.h"

template <int NH_, int HB_>
struct DeepGeo {
  static constexpr int NH = NH_, HB = HB_, NL = NH_ + 1;   // hidden layers, blocks per hidden layer, Dense layers
  static constexpr int ib(int l) { return l == 0 ? 1 : HB; }
  static constexpr int ob(int l) { return l == NH ? 1 : HB; }
  static constexpr int S(int l) { return 32 * ob(l) + NF_IMG_PAD; }
  static constexpr int W(int l) {  // image offset of layer l's weights [in][S(l)]; W(NL) = end
    int o = 0;
    for (int i = 0; i < l; ++i) o += 32 * ib(i) * S(i) + 32 * ob(i);
    return o;
  }
  static constexpr int B(int l) { return W(l) + 32 * ib(l) * S(l); }
  static constexpr int SIZE = ((W(NL) + 3) / 4) * 4;       // floats, 16-byte multiple
  static constexpr int SH = 32 * HB + NF_IMG_PAD;          // row stride of every layer but the last
  static constexpr int SO = 32 + NF_IMG_PAD;
};

struct DeepDims {  // one net
  int nl;                      // Dense layers = hidden + 1
  int n[NF_MAX_HIDDEN + 2];    // widths: n[0] = fan-in, n[1..nl-1] hidden, n[nl] = fan-out
  long off;                    // theta offset (Optimisers.destructure order: W1, b1, W2, b2, ...)
};
struct DeepPack {
  int d, ncoup, nh;
  int h[NF_MAX_HIDDEN];
  long pair_params, odd_params;
};
__host__ __device__ inline long deep_net_params(const int *h, int nh, int m, int c) {
  long p = 0;
  int in = m;
  for (int i = 0; i < nh; ++i) { p += (long)in * h[i] + h[i]; in = h[i]; }
  return p + (long)in * c + c;
}
__host__ __device__ inline DeepDims deep_dims_of(const DeepPack &p, int k, int net) {
  DeepDims dd;
  const int c = (k & 1) ? p.d / 2 : (p.d + 1) / 2, m = p.d - c;
  dd.nl = p.nh + 1;
  dd.n[0] = m;
  for (int i = 0; i < p.nh; ++i) dd.n[i + 1] = p.h[i];
  dd.n[p.nh + 1] = c;
  dd.off = (long)(k >> 1) * p.pair_params + ((k & 1) ? p.odd_params : 0) + (net ? deep_net_params(p.h, p.nh, m, c) : 0);
  return dd;
}
// theta index of element e of a net's padded image, or -1 for padding
template <class G>
__host__ __device__ inline long deep_theta_index(const DeepDims &dd, int e) {
  long woff = dd.off;
  for (int l = 0; l < G::NL; ++l) {
    const int nin = dd.n[l], nout = dd.n[l + 1];
    if (e < G::W(l + 1)) {
      const int r = e - G::W(l), rows = 32 * G::ib(l), S = G::S(l);
      if (r < rows * S) {
        const int i = r / S, o = r - i * S;
        return (i < nin && o < nout) ? woff + (long)i * nout + o : -1;
      }
      const int o = r - rows * S;
      return o < nout ? woff + (long)nin * nout + o : -1;
    }
    woff += (long)nin * nout + nout;
  }
  return -1;
}

template <class G>
__global__ __launch_bounds__(256) void k_deep_pack(DeepPack p, const float *__restrict__ theta, float *__restrict__ out) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)p.ncoup * 2 * G::SIZE) return;
  const int k = (int)(gid / (2 * G::SIZE)), net = (int)((gid / G::SIZE) & 1), e = (int)(gid % G::SIZE);
  const long ti = deep_theta_index<G>(deep_dims_of(p, k, net), e);
  out[gid] = ti >= 0 ? theta[ti] : 0.f;
}

// g[theta index] = sum over workgroup slabs (fixed order: deterministic); 64 image elements per block, the slabs split over
// the block's four waves (as k_rqs_reduce_slabs)
template <class G>
__global__ __launch_bounds__(256) void k_deep_reduce_slabs(DeepPack p, const float *__restrict__ slab, int nslab, long slab_stride,
                                                           float *__restrict__ g, const double *__restrict__ lpart, int nlpart,
                                                           float *__restrict__ lout) {
  if (lout && blockIdx.x == 0) {  // block 0 also finishes the step's loss from the forward launch's partials (fixed order)
    __shared__ double sm[4];
    double c = 0.0;
    for (int i = threadIdx.x; i < nlpart; i += 256) c += lpart[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) *lout = (float)((sm[0] + sm[1]) + (sm[2] + sm[3]));
  }
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long gid = (long)blockIdx.x * 64 + lane;
  long ti = -1;
  if (gid < (long)p.ncoup * 2 * G::SIZE) {
    const int k = (int)(gid / (2 * G::SIZE)), net = (int)((gid / G::SIZE) & 1), e = (int)(gid % G::SIZE);
    ti = deep_theta_index<G>(deep_dims_of(p, k, net), e);
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (ti >= 0) {
    int s = q;
    for (; s + 12 < nslab; s += 16) {
      a0 += slab[(long)s * slab_stride + gid];
      a1 += slab[(long)(s + 4) * slab_stride + gid];
      a2 += slab[(long)(s + 8) * slab_stride + gid];
      a3 += slab[(long)(s + 12) * slab_stride + gid];
    }
    for (; s < nslab; s += 4) a0 += slab[(long)s * slab_stride + gid];
  }
  part[q][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (q == 0 && ti >= 0) g[ti] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// out = net(x): the NH hidden layers with leaky-ReLU, the output layer without activation (src/flows/utils.jl:71-100)
template <class G>
__device__ __forceinline__ void deep_forward(const float *__restrict__ img, const f32x16 (&x)[1], f32x16 (&out)[1], int l31, int hi) {
  f32x16 cur[G::HB];
  dense_fwd<1, G::HB, G::SH>(img + G::W(0), img + G::B(0), x, cur, l31, hi);
#pragma unroll
  for (int b = 0; b < G::HB; ++b) nf_lrelu16(cur[b]);
#pragma unroll
  for (int l = 1; l < G::NH; ++l) {
    f32x16 nxt[G::HB];
    dense_fwd<G::HB, G::HB, G::SH>(img + G::W(l), img + G::B(l), cur, nxt, l31, hi);
#pragma unroll
    for (int b = 0; b < G::HB; ++b) {
      nf_lrelu16(nxt[b]);
      cur[b] = nxt[b];
    }
  }
  dense_fwd<G::HB, 1, G::SO>(img + G::W(G::NH), img + G::B(G::NH), cur, out, l31, hi);
}

struct DeepChainArgs {
  const float *wimg;  // [coupling][s | t][G::SIZE]
  int d, ncoup;
  int k_only;         // -1: every coupling; otherwise only the coupling with this flat index
  int accumulate;     // ladj += instead of =
  long N;
};

// whole-flow forward / inverse in one launch (with_logabsdet_jacobian of the ComposedFunction, src/objectives/elbo.jl:67;
// the inverse chain from loglikelihood.jl:31).  State of a tile in registers, split by feature parity (E = features 0, 2, ..:
// transformed by the 1:2:d couplings; O = 1, 3, ..: by the 2:2:d ones), as k_affine_chain.
template <class G, bool INVERSE>
__global__ __launch_bounds__(512) void k_deep_chain(DeepChainArgs a, float *xt, float *__restrict__ ladj) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const long ngroups = (ntiles + 7) / 8;
  const int c_odd = (a.d + 1) / 2, c_even = a.d / 2;  // mask 1:2:d / 2:2:d
  auto coupling_at = [&](int s) { return INVERSE ? s : a.ncoup - 1 - s; };
  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long tile = grp * 8 + wave;
    const bool live = tile < ntiles;
    const long tl = live ? tile : 0;
    const long j = tl * NF_TILE + l31;
    const bool valid = live && j < a.N;
    const TileIO io = make_tile_io(xt, tl, a.d, l31, hi);
    f32x16 E[1], O[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = tile_load(io, tile_soff(0, r, 0));
      const float o = tile_load(io, tile_soff(0, r, 1));
      E[0][r] = valid ? e : 0.f;
      O[0][r] = valid ? o : 0.f;
    }
    float lsum = 0.f;
#pragma unroll 1
    for (int s = 0; s < a.ncoup; ++s) {
      const int k = coupling_at(s);
      if (a.k_only >= 0 && a.k_only != k) continue;  // (uniform)
      __syncthreads();  // every wave is done with the previous coupling's images
      stage_packed<2 * G::SIZE, 512>(lds, a.wimg + (size_t)k * 2 * G::SIZE, tid);
      __syncthreads();
      const bool odd_mask = (k & 1) == 0;  // flat coupling k even: mask 1:2:d, transformed half = E
      const int c = odd_mask ? c_odd : c_even;
      f32x16 S[1], T[1];
      if (odd_mask) {
        deep_forward<G>(lds, O, S, l31, hi);
        deep_forward<G>(lds + G::SIZE, O, T, l31, hi);
      } else {
        deep_forward<G>(lds, E, S, l31, hi);
        deep_forward<G>(lds + G::SIZE, E, T, l31, hi);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = nf_row(r, hi);
        const float sv = nf_tanh(S[0][r]);  // padded rows: zero weights and bias => s = 0, t = 0
        const float v = odd_mask ? E[0][r] : O[0][r];
        const float o = INVERSE ? (v - T[0][r]) * nf_exp(-sv) : v * nf_exp(sv) + T[0][r];
        const bool ok = p < c;
        if (odd_mask) E[0][r] = ok ? o : v; else O[0][r] = ok ? o : v;
        lsum += ok ? sv : 0.f;
      }
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        tile_store(io, tile_soff(0, r, 0), E[0][r]);
        tile_store(io, tile_soff(0, r, 1), O[0][r]);
      }
      lsum += __shfl_xor(lsum, 32);
      if (hi == 0 && valid) {
        const float base = a.accumulate ? ladj[j] : 0.f;
        ladj[j] = INVERSE ? base - lsum : base + lsum;
      }
    }
  }
}

// ---- reverse pass ---------------------------------------------------------------------------------------------------
template <class G>
struct DeepAcc {
  f32x16 w0[1][G::HB];                                   // layer 0: [conditioner block][hidden blocks]
  f32x16 wh[G::NH > 1 ? G::NH - 1 : 1][G::HB][G::HB];    // hidden -> hidden layers 1 .. NH-1
  f32x16 wo[G::HB][1];                                   // output layer
  float b0[G::HB], bh[G::NH > 1 ? G::NH - 1 : 1][G::HB], bo[1];
};
template <class G>
struct DeepLds {  // floats
  static constexpr int OFF_X = 0;                          // x2^T            [32 rows][NF_TS]
  static constexpr int OFF_A = OFF_X + 32 * NF_TS;         // a_{l-1}^T of the layer whose dW is next
  static constexpr int OFF_D = OFF_A + 32 * G::HB * NF_TS; // delta_l^T
  static constexpr int SCRATCH = OFF_D + 32 * G::HB * NF_TS;
  static constexpr size_t BYTES = (size_t)(G::SIZE + 4 * SCRATCH) * sizeof(float);
};
template <int IB, int OB>
__device__ __forceinline__ void deep_zero(f32x16 (&a)[IB][OB], float (&b)[OB]) {
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[i][o][r] = 0.f;
#pragma unroll
  for (int o = 0; o < OB; ++o) b[o] = 0.f;
}
// The blocks of a layer whose running number (B0 + the block's index in the layer) is `sel` modulo 4: see the fold in k_deep_bwd.
template <int IB, int OB, int S, int B0>
__device__ __forceinline__ void deep_fold(float *__restrict__ w, float *__restrict__ b, const f32x16 (&a)[IB][OB], const float (&bs)[OB],
                                          bool first, int sel, int l31, int hi) {
  // A block's sixteen partial sums are read in one go, then written back (round 5).  As one read-modify-write after the other
  // (`*p = *p + a`) every LDS round trip was exposed: tools/trace_deep_bwd.py showed the fold at 15.0 k of a phase's 56.6 k
  // clocks for one hidden layer of 64 and 52.8 k of 183 k for three -- more than a quarter of the reverse kernel.
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int o = 0; o < OB; ++o) {
      if (((B0 + i * OB + o) & 3) != sel) continue;  // (wave-uniform)
      float *p = w + (i * 32 + 4 * hi) * S + o * 32 + l31;  // row nf_row(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi
      float old[16];
      if (!first) {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = p[((r & 3) + 8 * (r >> 2)) * S];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2)) * S] = first ? a[i][o][r] : old[r] + a[i][o][r];
    }
#pragma unroll
  for (int o = 0; o < OB; ++o) {  // the bias sums travel with the layer's block (0, o)
    if (((B0 + o) & 3) != sel) continue;
    const float v = bs[o] + __shfl_xor(bs[o], 32);
    if (hi == 0) {
      float *p = b + o * 32 + l31;
      *p = first ? v : *p + v;
    }
  }
}

// every block of a layer (the waves-in-turn fold)
template <int IB, int OB, int S>
__device__ __forceinline__ void deep_fold_all(float *__restrict__ w, float *__restrict__ b, const f32x16 (&a)[IB][OB], const float (&bs)[OB],
                                              bool first, int l31, int hi) {
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int o = 0; o < OB; ++o) {
      float *p = w + (i * 32 + 4 * hi) * S + o * 32 + l31;  // row nf_row(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi
      float old[16];
      if (!first) {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = p[((r & 3) + 8 * (r >> 2)) * S];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2)) * S] = first ? a[i][o][r] : old[r] + a[i][o][r];
    }
#pragma unroll
  for (int o = 0; o < OB; ++o) {
    const float v = bs[o] + __shfl_xor(bs[o], 32);
    if (hi == 0) {
      float *p = b + o * 32 + l31;
      *p = first ? v : *p + v;
    }
  }
}

struct DeepBwdArgs {
  const float *wimg;
  int d, ncoup, k_lo, k_hi;  // flat couplings [k_lo, k_hi)
  long N;
  long long *trace;  // NF_KERNEL_TRACE builds: clock stamps of workgroup 0 / wave 0 (tools/trace_deep_bwd.py), else unused
};
#ifdef NF_KERNEL_TRACE
#define DEEP_STAMP(slot) do { if (tr) { __builtin_amdgcn_sched_barrier(0); tr[slot] = clock64(); } } while (0)
#else
#define DEEP_STAMP(slot) do { (void)tr; } while (0)
#endif

// reverse pass of one (tile, coupling, net).  PHASE_S / INVD as bwd_tile of nf_coupling.hip:
//   forward chain:  phase T first (y1 <- u = y1 - T, T-bar = ybar1), then phase S (x1 = u exp(-s), x1bar = ybar1 exp(s),
//                   S-bar = (ybar1 u + lbar) (1 - s^2));
//   INVD (reverse pass of the INVERSE coupling at its output w, forward-KL training): phase S first (w1 exp(s), v1bar =
//                   w1bar exp(-s), S-bar = -(w1bar w1 + lbar)(1 - s^2)), then phase T (v1 = . + t, T-bar = -v1bar).
template <class G, bool PHASE_S, bool INVD>
__device__ __forceinline__ void deep_bwd_tile(const float *__restrict__ img, float *__restrict__ sc, DeepAcc<G> &acc,
                                              float *__restrict__ y, float *__restrict__ ybar, const float *__restrict__ lbar,
                                              float lbar_const, int d, int c, int par_t, long N, long tile, int l31, int hi,
                                              long long *tr = nullptr) {
  using L = DeepLds<G>;
  DEEP_STAMP(0);
  const long j = tile * NF_TILE + l31;
  const bool valid = j < N;
  const int par_c = 1 - par_t;
  const TileIO yio = make_tile_io(y, tile, d, l31, hi);
  const TileIO gio = make_tile_io(ybar, tile, d, l31, hi);
  float *sx = sc + L::OFF_X, *sa = sc + L::OFF_A, *sd = sc + L::OFF_D;

  f32x16 act[G::NH][G::HB];  // post-activation hidden layers
  unsigned msk[G::NH][G::HB];
  f32x16 d3[1], y1[1], g1[1];
  {
    f32x16 xb[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = tile_load(yio, tile_soff(0, r, par_c));  // features >= d read as 0
      xb[0][r] = valid ? v : 0.f;
    }
    DEEP_STAMP(1);  // (the first use of xb forces the wait: x2 has arrived)
    tile_to_scratch<1>(sx, xb, l31, hi);
    dense_fwd<1, G::HB, G::SH>(img + G::W(0), img + G::B(0), xb, act[0], l31, hi);
  }
#pragma unroll
  for (int b = 0; b < G::HB; ++b) {
    nf_lrelu16(act[0][b]);
    msk[0][b] = nf_sign_mask16(act[0][b]);
  }
#pragma unroll
  for (int l = 1; l < G::NH; ++l) {
    dense_fwd<G::HB, G::HB, G::SH>(img + G::W(l), img + G::B(l), act[l - 1], act[l], l31, hi);
#pragma unroll
    for (int b = 0; b < G::HB; ++b) {
      nf_lrelu16(act[l][b]);
      msk[l][b] = nf_sign_mask16(act[l][b]);
    }
  }
  // operands of the element-wise stage: requested here, consumed after the last forward layer
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    y1[0][r] = tile_load(yio, tile_soff(0, r, par_t));
    g1[0][r] = tile_load(gio, tile_soff(0, r, par_t));
  }
  DEEP_STAMP(2);
  dense_fwd<G::HB, 1, G::SO>(img + G::W(G::NH), img + G::B(G::NH), act[G::NH - 1], d3, l31, hi);
  DEEP_STAMP(3);

  const float lb = valid ? (lbar ? lbar[j < N ? j : 0] : lbar_const) : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int p = nf_row(r, hi);
    const bool ok = (p < c) && valid;  // rows >= c: the loads returned 0, the stores are dropped
    const float yv = y1[0][r], gv = g1[0][r];
    if (INVD && !PHASE_S) {
      tile_store(yio, tile_soff(0, r, par_t), yv + d3[0][r]);  // v1 = w1 exp(s) + t
      d3[0][r] = ok ? -gv : 0.f;                                // T-bar = -v1bar
    } else if (INVD) {
      const float s = nf_tanh(d3[0][r]);
      const float es = nf_exp(s);
      tile_store(yio, tile_soff(0, r, par_t), yv * es);           // w1 exp(s)
      tile_store(gio, tile_soff(0, r, par_t), nf_fdiv(gv, es));   // v1bar
      d3[0][r] = ok ? -(gv * yv + lb) * (1.f - s * s) : 0.f;      // S-bar through tanh
    } else if (!PHASE_S) {
      tile_store(yio, tile_soff(0, r, par_t), yv - d3[0][r]);  // u = x1 exp(S)
      d3[0][r] = ok ? gv : 0.f;                                 // T-bar = ybar1
    } else {
      const float s = nf_tanh(d3[0][r]);
      const float es = nf_exp(s);
      tile_store(yio, tile_soff(0, r, par_t), nf_fdiv(yv, es));  // x1 = u exp(-s)
      tile_store(gio, tile_soff(0, r, par_t), gv * es);           // x1bar
      d3[0][r] = ok ? (gv * yv + lb) * (1.f - s * s) : 0.f;       // S-bar through tanh
    }
  }

  DEEP_STAMP(4);
  // ---- output layer: dX, then dW^T from the transposed operands
  f32x16 dh[G::HB];
  dense_bwd_x<G::HB, 1, G::SO>(img + G::W(G::NH), d3, dh, l31, hi);
  tile_to_scratch<G::HB>(sa, act[G::NH - 1], l31, hi);
  tile_to_scratch<1>(sd, d3, l31, hi);
  wave_lds_fence();
  dw_accumulate<G::HB, 1>(sa, sd, acc.wo, acc.bo, l31, hi);
#pragma unroll
  for (int b = 0; b < G::HB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[b][r] *= nf_mask_slope(msk[G::NH - 1][b], r);
  wave_lds_fence();
  DEEP_STAMP(5);
  // ---- hidden -> hidden layers, last to first
#pragma unroll
  for (int l = G::NH - 1; l >= 1; --l) {
    f32x16 dp[G::HB];
    dense_bwd_x<G::HB, G::HB, G::SH>(img + G::W(l), dh, dp, l31, hi);
    tile_to_scratch<G::HB>(sa, act[l - 1], l31, hi);
    tile_to_scratch<G::HB>(sd, dh, l31, hi);
    wave_lds_fence();
    dw_accumulate<G::HB, G::HB>(sa, sd, acc.wh[l - 1], acc.bh[l - 1], l31, hi);
#pragma unroll
    for (int b = 0; b < G::HB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[b][r] = dp[b][r] * nf_mask_slope(msk[l - 1][b], r);
    wave_lds_fence();
  }
  DEEP_STAMP(6);
  // ---- layer 0: x2bar accumulates ybar2 + W0t^T d (phase T) + W0s^T d (phase S)
  f32x16 g2[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) g2[0][r] = tile_load(gio, tile_soff(0, r, par_c));
  dense_bwd_x<1, G::HB, G::SH, true>(img + G::W(0), dh, g2, l31, hi);
  tile_to_scratch<G::HB>(sd, dh, l31, hi);
  wave_lds_fence();
  dw_accumulate<1, G::HB>(sx, sd, acc.w0, acc.b0, l31, hi);
#pragma unroll
  for (int r = 0; r < 16; ++r) tile_store(gio, tile_soff(0, r, par_c), g2[0][r]);
  wave_lds_fence();
  DEEP_STAMP(7);
}

// Reverse pass of the couplings [k_lo, k_hi) in one launch: a wave's tiles never change hands and coupling k + 1 only reads what
// the same wave wrote for coupling k (y <- x, ybar <- xbar), so each workgroup walks the couplings on its own (as
// k_affine_bwd_all).  INVD: the inverse chain's reverse pass, couplings in execution order.
template <class G, bool INVD>
__global__ __launch_bounds__(256, 1) void k_deep_bwd(DeepBwdArgs a, float *__restrict__ y, float *__restrict__ ybar,
                                                     const float *__restrict__ lbar, float lbar_const, float *__restrict__ slab,
                                                     long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *img = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  float *sc = lds + G::SIZE + wave * DeepLds<G>::SCRATCH;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
#pragma unroll 1
  for (int step = 0; step < a.k_hi - a.k_lo; ++step) {
    const int k = INVD ? a.k_hi - 1 - step : a.k_lo + step;
    const int par_t = k & 1;
    const int c = (k & 1) ? a.d / 2 : (a.d + 1) / 2;
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
      const bool is_s = INVD ? phase == 0 : phase == 1;  // forward chain: T then S; inverse chain: S then T
      // stamps: [0..4] of the first phase: start, image staged, tiles done, folded, slab written; [8 + 8 i ..] tile i of that phase
      long long *tr = (a.trace && blockIdx.x == 0 && tid == 0 && step == 0 && phase == 0) ? a.trace : nullptr;
      DEEP_STAMP(0);
      stage_packed<G::SIZE, 256>(img, a.wimg + ((size_t)k * 2 + (is_s ? 0 : 1)) * G::SIZE, tid);
      __syncthreads();
      DEEP_STAMP(1);
      int tcount = 0;
      DeepAcc<G> acc;
      deep_zero(acc.w0, acc.b0);
#pragma unroll
      for (int l = 0; l < (G::NH > 1 ? G::NH - 1 : 1); ++l) deep_zero(acc.wh[l], acc.bh[l]);
      deep_zero(acc.wo, acc.bo);
#pragma unroll 1
      for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        long long *trt = (tr && tcount < 8) ? tr + 8 + 8 * tcount : nullptr;
        ++tcount;
        if (is_s)
          deep_bwd_tile<G, true, INVD>(img, sc, acc, y, ybar, lbar, lbar_const, a.d, c, par_t, a.N, tile, l31, hi, trt);
        else
          deep_bwd_tile<G, false, INVD>(img, sc, acc, y, ybar, lbar, lbar_const, a.d, c, par_t, a.N, tile, l31, hi, trt);
      }
      __syncthreads();  // the weight image is dead: it becomes the fold target
      DEEP_STAMP(2);
      // Four steps, every wave busy in each: in step s wave w adds its sums to the blocks numbered (w + s) mod 4 -- block b is
      // written by wave b mod 4 first, then added to by waves b - 1, b - 2, b - 3 (mod 4): a fixed order per element, and the
      // four waves one after the other over the WHOLE image took four times as long (hidden [64]: 364 -> 332 us per launch).
      // Three hidden layers of 64 keep the waves-in-turn form (ROT = false): the kernel sits at 512 registers with scratch
      // spills, and with the rotating fold hipcc's allocation of its TILE loop came out 11 % slower at 1 M samples
      // (15.4 against 13.9 ms; tools/bench_deep_n.py, one box).
      constexpr bool ROT = !(G::NH >= 3 && G::HB == 2);
      if constexpr (ROT) {
#pragma unroll 1
        for (int st = 0; st < 4; ++st) {
          const int sel = (wave + st) & 3;
          constexpr int NB0 = G::HB, NBH = G::HB * G::HB;  // blocks of layer 0 / of a hidden -> hidden layer
          deep_fold<1, G::HB, G::SH, 0>(img + G::W(0), img + G::B(0), acc.w0, acc.b0, st == 0, sel, l31, hi);
#pragma unroll
          for (int l = 1; l < G::NH; ++l) {
            // (B0 must be a constant expression: the layers are unrolled by hand for up to four hidden layers)
            if (l == 1) deep_fold<G::HB, G::HB, G::SH, NB0>(img + G::W(1), img + G::B(1), acc.wh[0], acc.bh[0], st == 0, sel, l31, hi);
            if (l == 2) deep_fold<G::HB, G::HB, G::SH, NB0 + NBH>(img + G::W(2), img + G::B(2), acc.wh[G::NH > 2 ? 1 : 0], acc.bh[G::NH > 2 ? 1 : 0], st == 0, sel, l31, hi);
            if (l == 3) deep_fold<G::HB, G::HB, G::SH, NB0 + 2 * NBH>(img + G::W(3), img + G::B(3), acc.wh[G::NH > 3 ? 2 : 0], acc.bh[G::NH > 3 ? 2 : 0], st == 0, sel, l31, hi);
          }
          deep_fold<G::HB, 1, G::SO, NB0 + (G::NH - 1) * NBH>(img + G::W(G::NH), img + G::B(G::NH), acc.wo, acc.bo, st == 0, sel, l31, hi);
          __syncthreads();
        }
      } else {
#pragma unroll 1
        for (int w = 0; w < 4; ++w) {
          if (wave == w) {
            deep_fold_all<1, G::HB, G::SH>(img + G::W(0), img + G::B(0), acc.w0, acc.b0, w == 0, l31, hi);
#pragma unroll
            for (int l = 1; l < G::NH; ++l)
              deep_fold_all<G::HB, G::HB, G::SH>(img + G::W(l), img + G::B(l), acc.wh[l - 1], acc.bh[l - 1], w == 0, l31, hi);
            deep_fold_all<G::HB, 1, G::SO>(img + G::W(G::NH), img + G::B(G::NH), acc.wo, acc.bo, w == 0, l31, hi);
          }
          __syncthreads();
        }
      }
      DEEP_STAMP(3);
      {
        const float4 *c0 = reinterpret_cast<const float4 *>(img);
        float4 *dst = reinterpret_cast<float4 *>(slab + (long)blockIdx.x * slab_stride + ((long)k * 2 + (is_s ? 0 : 1)) * G::SIZE);
        for (int i = tid; i < G::SIZE / 4; i += 256) dst[i] = c0[i];
      }
      __syncthreads();  // the image region is restaged by the next phase (its loads of y / ybar see this phase's stores:
                        // same wave, vmcnt(0) at the barrier, write-through L1 -- as k_affine_bwd_all)
      DEEP_STAMP(4);
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------------------------
using DeepG11 = DeepGeo<1, 1>;
using DeepG12 = DeepGeo<1, 2>;
using DeepG31 = DeepGeo<3, 1>;
using DeepG32 = DeepGeo<3, 2>;
using DeepG41 = DeepGeo<4, 1>;

// 0: not a deep-resident shape; otherwise 10 * hidden layers + blocks per hidden layer
int nf_deep_geo_id(const nf_flow_desc *desc) {
  static const bool off = std::getenv("NF_DEEP_OFF") != nullptr;  // A/B: these shapes on the general layer-by-layer path
  if (off || desc->kind != NF_KIND_REALNVP || desc->dtype != NF_DTYPE_F32) return 0;
  if (desc->d < 2 || desc->d > 64) return 0;
  const int nh = desc->n_hidden;
  if (nh != 1 && nh != 3 && nh != 4) return 0;
  int hmax = 0;
  for (int i = 0; i < nh; ++i) {
    if (desc->hdims[i] < 1) return 0;
    hmax = desc->hdims[i] > hmax ? desc->hdims[i] : hmax;
  }
  const int hb = (hmax + 31) / 32;
  if (hb > 2 || (nh == 4 && hb > 1)) return 0;  // four hidden layers of 64: 256 accumulator registers, left to the general path
  return 10 * nh + hb;
}
#define DEEP_DISPATCH(ID, CALL)                                                                          \
  ((ID) == 11 ? CALL(DeepG11) : (ID) == 12 ? CALL(DeepG12) : (ID) == 31 ? CALL(DeepG31) : (ID) == 32 ? CALL(DeepG32) : CALL(DeepG41))
static int deep_size(int id) {
  return id == 11 ? DeepG11::SIZE : id == 12 ? DeepG12::SIZE : id == 31 ? DeepG31::SIZE : id == 32 ? DeepG32::SIZE : id == 41 ? DeepG41::SIZE : 0;
}
int nf_deep_image_floats(const nf_flow_desc *desc) { return deep_size(nf_deep_geo_id(desc)); }
size_t nf_deep_wimg_bytes(const nf_flow_desc *desc) { return (size_t)2 * desc->nlayers * 2 * deep_size(nf_deep_geo_id(desc)) * sizeof(float); }
long nf_deep_slab_floats(const nf_flow_desc *desc) { return (long)2 * desc->nlayers * 2 * deep_size(nf_deep_geo_id(desc)); }

static DeepPack deep_pack_args(const nf_flow_desc *desc) {
  DeepPack p;
  p.d = desc->d; p.ncoup = 2 * desc->nlayers; p.nh = desc->n_hidden;
  for (int i = 0; i < NF_MAX_HIDDEN; ++i) p.h[i] = i < desc->n_hidden ? desc->hdims[i] : 0;
  const CouplingInfo c0 = nf_coupling_info(desc, 0), c1 = nf_coupling_info(desc, 1);
  p.odd_params = c0.nparams;
  p.pair_params = c0.nparams + c1.nparams;
  return p;
}

int nf_wimg_reserve(nf_ctx *ctx, size_t bytes);

template <class G>
static int deep_launch_pack(nf_ctx *ctx, unsigned grid, const DeepPack &p, const float *theta) {
  hipLaunchKernelGGL((k_deep_pack<G>), dim3(grid), dim3(256), 0, ctx->stream, p, theta, (float *)ctx->wimg);
  return (int)hipGetLastError();
}
template <class G>
static int deep_launch_reduce(nf_ctx *ctx, unsigned grid, const DeepPack &p, const float *slab, int nslab, long total, float *g,
                              const double *lpart, int nlpart, float *lout) {
  hipLaunchKernelGGL((k_deep_reduce_slabs<G>), dim3(grid), dim3(256), 0, ctx->stream, p, slab, nslab, total, g, lpart, nlpart, lout);
  return (int)hipGetLastError();
}

int nf_deep_pack(nf_ctx *ctx, const nf_flow_desc *desc, const float *theta) {
  const int id = nf_deep_geo_id(desc);
  if (!id) return NF_ERR_UNSUPPORTED;
  NF_TRY(nf_wimg_reserve(ctx, nf_deep_wimg_bytes(desc)));
  const DeepPack p = deep_pack_args(desc);
  const long total = (long)p.ncoup * 2 * deep_size(id);
  const unsigned grid = (unsigned)((total + 255) / 256);
  ProfScope ps(ctx, "pack_weights");
#define DEEP_CALL(G) deep_launch_pack<G>(ctx, grid, p, theta)
  return DEEP_DISPATCH(id, DEEP_CALL);
#undef DEEP_CALL
}

int nf_deep_reduce_slabs(nf_ctx *ctx, const nf_flow_desc *desc, const float *slab, int nslab, float *g, const double *lpart, int nlpart,
                         float *lout) {
  const int id = nf_deep_geo_id(desc);
  if (!id) return NF_ERR_UNSUPPORTED;
  const DeepPack p = deep_pack_args(desc);
  const long total = (long)p.ncoup * 2 * deep_size(id);
  const unsigned grid = (unsigned)((total + 63) / 64);
  ProfScope ps(ctx, "reduce_slabs");
#define DEEP_CALL(G) deep_launch_reduce<G>(ctx, grid, p, slab, nslab, total, g, lpart, nlpart, lout)
  return DEEP_DISPATCH(id, DEEP_CALL);
#undef DEEP_CALL
}

template <class G, bool INV>
static int deep_launch_chain(nf_ctx *ctx, const DeepChainArgs &a, float *xt, float *ladj) {
  const size_t lds = 2 * (size_t)G::SIZE * sizeof(float);
  static AttrOnce attr_once;  // once per device
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_deep_chain<G, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  const long ngroups = ((a.N + NF_TILE - 1) / NF_TILE + 7) / 8;
  long grid = ngroups < ctx->num_cu ? ngroups : ctx->num_cu;
  if (grid < 1) grid = 1;
  ProfScope ps(ctx, "deep_chain");
  hipLaunchKernelGGL((k_deep_chain<G, INV>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a, xt, ladj);
  return (int)hipGetLastError();
}

// all couplings (k_only < 0) or one, in place on the tiled buffer; packed images must be current
int nf_deep_chain(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, float *xt, long N, float *ladj, int k_only, int accumulate) {
  const int id = nf_deep_geo_id(desc);
  if (!id || !ctx->wimg) return NF_ERR_UNSUPPORTED;
  DeepChainArgs a;
  a.wimg = (const float *)ctx->wimg;
  a.d = desc->d; a.ncoup = 2 * desc->nlayers; a.k_only = k_only; a.accumulate = accumulate; a.N = N;
#define DEEP_CALL(G) (inverse ? deep_launch_chain<G, true>(ctx, a, xt, ladj) : deep_launch_chain<G, false>(ctx, a, xt, ladj))
  return DEEP_DISPATCH(id, DEEP_CALL);
#undef DEEP_CALL
}

template <class G, bool INVD>
static int deep_launch_bwd(nf_ctx *ctx, const DeepBwdArgs &a, float *y, float *ybar, const float *lbar, float lbar_const, float *slab,
                           long slab_stride, int grid) {
  const size_t lds = DeepLds<G>::BYTES;
  static AttrOnce attr_once;  // once per device
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_deep_bwd<G, INVD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  ProfScope ps(ctx, INVD ? "deep_bwd_inv" : "deep_bwd");
  hipLaunchKernelGGL((k_deep_bwd<G, INVD>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, y, ybar, lbar, lbar_const, slab, slab_stride);
  return (int)hipGetLastError();
}

// reverse pass of the flat couplings [k_lo, k_hi) (every coupling: 0, 2 nlayers) with invertible recompute; slab layout
// [workgroup][coupling][s | t][image].  inv_dir: the inverse chain's reverse pass (forward-KL training).
int nf_deep_bwd(nf_ctx *ctx, const nf_flow_desc *desc, int k_lo, int k_hi, float *y, float *ybar, const float *lbar, float lbar_const,
                long N, float *slab, long slab_stride, int grid, bool inv_dir) {
  const int id = nf_deep_geo_id(desc);
  if (!id || !ctx->wimg) return NF_ERR_UNSUPPORTED;
  DeepBwdArgs a;
  a.wimg = (const float *)ctx->wimg;
  a.d = desc->d; a.ncoup = 2 * desc->nlayers; a.k_lo = k_lo; a.k_hi = k_hi; a.N = N;
  a.trace = (long long *)ctx->trace;
#define DEEP_CALL(G) (inv_dir ? deep_launch_bwd<G, true>(ctx, a, y, ybar, lbar, lbar_const, slab, slab_stride, grid) \
                              : deep_launch_bwd<G, false>(ctx, a, y, ybar, lbar, lbar_const, slab, slab_stride, grid))
  return DEEP_DISPATCH(id, DEEP_CALL);
#undef DEEP_CALL
}
