// nf_coupling.hip -- AffineCoupling (RealNVP) kernels for gfx950.
//
// Reference arithmetic: src/flows/realnvp.jl:57-110
//   forward  y1 = x1 .* exp.(s(x2)) .+ t(x2),  logjac =  sum(s(x2); dims=1)
//   inverse  x1 = (y1 .- t(y2)) .* exp.(-s(y2)), logjac = -sum(s(y2); dims=1)
// with s, t = fnn(...) (src/flows/utils.jl:71-100; s ends in tanh, realnvp.jl:50).
// PartitionMask partition/combine (realnvp.jl:59,62) is index arithmetic in the loads
// and stores here: transformed feature p <-> 2p + par_t, conditioner q <-> 2q + 1 - par_t.
//
// Work decomposition: one wavefront = one tile of 32 samples, activations chained through
// the fp32 matrix pipe in registers (nf_mfma.h); weights of the coupling live in LDS.
#include "nf_common.h"
#include "nf_mfma.h"

struct CouplingArgs {
  const float *theta;
  NetDims s, t;
  int d, c, m, par_t;
  long N;
};

// ------------------------------------------------------------------------------------
// forward / inverse of one coupling (no gradients): both nets resident in LDS
// ------------------------------------------------------------------------------------
template <class G, bool INVERSE>
__global__ __launch_bounds__(512) void k_affine_apply(CouplingArgs a, const float *x_in,
                                                      float *y_out, float *__restrict__ ladj,
                                                      int ladj_accumulate) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *img_s = lds;
  float *img_t = lds + G::SIZE;
  const int tid = threadIdx.x;
  stage_net<G>(img_s, a.theta, a.s, tid, 512);
  stage_net<G>(img_t, a.theta, a.t, tid, 512);
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int par_c = 1 - a.par_t;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const bool copy_cond = (y_out != x_in);

  for (long tile = (long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long)gridDim.x * 8) {
    const long j = tile * NF_TILE + l31;
    const bool valid = j < a.N;
    const float *xr = x_in + j * a.d;
    float *yr = y_out + j * a.d;

    f32x16 xb[G::MB];
#pragma unroll
    for (int b = 0; b < G::MB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = b * 32 + nf_row(r, hi);
        const bool ok = valid && q < a.m;
        const float v = ok ? xr[2 * q + par_c] : 0.f;
        xb[b][r] = v;
        if (copy_cond && ok) yr[2 * q + par_c] = v;
      }

    f32x16 S[G::CB], T[G::CB];
    {
      f32x16 a1[G::H1B], a2[G::H2B];
      net_forward<G>(img_s, xb, a1, a2, S, l31, hi);
    }
    {
      f32x16 a1[G::H1B], a2[G::H2B];
      net_forward<G>(img_t, xb, a1, a2, T, l31, hi);
    }
    float lsum = 0.f;
#pragma unroll
    for (int b = 0; b < G::CB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = b * 32 + nf_row(r, hi);
        const bool ok = valid && p < a.c;
        const float s = tanhf(S[b][r]);
        const float v = ok ? xr[2 * p + a.par_t] : 0.f;
        float o;
        if (INVERSE)
          o = (v - T[b][r]) * expf(-s);
        else
          o = v * expf(s) + T[b][r];
        if (ok) {
          yr[2 * p + a.par_t] = o;
          lsum += s;
        }
      }
    lsum += __shfl_xor(lsum, 32);
    if (hi == 0 && valid) {
      const float base = ladj_accumulate ? ladj[j] : 0.f;
      ladj[j] = INVERSE ? base - lsum : base + lsum;
    }
  }
}

// ------------------------------------------------------------------------------------
// reverse pass of one coupling with invertible recompute
// ------------------------------------------------------------------------------------
// In:  y    = coupling OUTPUT (d x N),  ybar = dL/dy,  lbar = dL/d(ladj) per sample
// Out: y    <- coupling INPUT x (reconstructed: x1 = (y1 - T) exp(-S), realnvp.jl:107),
//      ybar <- dL/dx,
//      slab[blockIdx.x] <- this workgroup's partial sum of dL/dtheta for this coupling.
//
// Two phases inside one launch, each with ONE net in LDS and its dW^T accumulators in
// registers:
//   phase T: T = t(x2);  u = y1 - T -> y1 slots;  delta3 = ybar1;            x2bar += W1t^T d1
//   phase S: S = s(x2);  x1 = u exp(-S) -> y1 slots; delta3 = (ybar1*u + lbar)(1 - S^2);
//            x1bar = ybar1 exp(S) -> ybar1 slots;                            x2bar += W1s^T d1
template <class G>
struct BwdAcc {
  f32x16 w1[G::MB][G::H1B];
  f32x16 w2[G::H1B][G::H2B];
  f32x16 w3[G::H2B][G::CB];
  float b1[G::H1B], b2[G::H2B], b3[G::CB];
};

template <int IB, int OB>
__device__ __forceinline__ void zero_acc(f32x16 (&a)[IB][OB], float (&b)[OB]) {
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[i][o][r] = 0.f;
#pragma unroll
  for (int o = 0; o < OB; ++o) b[o] = 0.f;
}

// fold one wave's accumulators into the LDS image (layout of the weight image)
template <int IB, int OB>
__device__ __forceinline__ void fold_acc(float *__restrict__ w, float *__restrict__ b, const f32x16 (&a)[IB][OB],
                                         const float (&bs)[OB], bool first, int l31, int hi) {
  constexpr int S = 32 * OB + 1;
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float *p = w + (i * 32 + nf_row(r, hi)) * S + o * 32 + l31;
        *p = first ? a[i][o][r] : *p + a[i][o][r];
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

template <int S>
__device__ __forceinline__ void unstage_dense(const float *__restrict__ img, int rows_pad, float *__restrict__ out,
                                              int nin, int nout, int tid, int nthreads) {
  for (int idx = tid; idx < rows_pad * S; idx += nthreads) {
    const int i = idx / S, o = idx - i * S;
    if (i < nin && o < nout) out[(long)i * nout + o] = img[idx];
  }
}

// sign masks of post-leakyrelu activations (bit r set <=> a[r] > 0): all the reverse pass
// needs of a1/a2 besides their LDS copies, so the activations themselves can die early.
template <int NB>
__device__ __forceinline__ void sign_masks(const f32x16 (&v)[NB], unsigned (&m)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    unsigned bits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) bits |= (v[b][r] > 0.f ? 1u : 0u) << r;
    m[b] = bits;
  }
}
template <int NB>
__device__ __forceinline__ void apply_lrelu_grad(f32x16 (&d)[NB], const unsigned (&m)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) d[b][r] *= ((m[b] >> r) & 1u) ? 1.f : 0.01f;
}

// per-wave LDS scratch of the reverse pass, [feature][sample] tiles with row stride NF_TS:
//   x2 | a1 | a2 | delta(current layer)
template <class G>
struct BwdLds {
  static constexpr int DROWS = (G::H1B > G::H2B ? (G::H1B > G::CB ? G::H1B : G::CB) : (G::H2B > G::CB ? G::H2B : G::CB));
  static constexpr int OFF_X = 0;
  static constexpr int OFF_A1 = OFF_X + G::MB * 32 * NF_TS;
  static constexpr int OFF_A2 = OFF_A1 + G::H1B * 32 * NF_TS;
  static constexpr int OFF_D = OFF_A2 + G::H2B * 32 * NF_TS;
  static constexpr int SCRATCH = OFF_D + DROWS * 32 * NF_TS;  // floats per wave
  static constexpr int WAVES = 4;
  static constexpr size_t BYTES = (size_t)(G::SIZE + WAVES * SCRATCH) * sizeof(float);
};

template <class G, bool PHASE_S>
__device__ __forceinline__ void bwd_tile(const CouplingArgs &a, const float *__restrict__ img, float *__restrict__ sc,
                                         BwdAcc<G> &acc, float *__restrict__ y, float *__restrict__ ybar,
                                         const float *__restrict__ lbar, float lbar_const, long tile, int l31, int hi) {
  using L = BwdLds<G>;
  const long j = tile * NF_TILE + l31;
  const bool valid = j < a.N;
  const int par_c = 1 - a.par_t;
  float *yr = y + j * a.d;
  float *gr = ybar + j * a.d;

  f32x16 d3[G::CB];
  unsigned m1[G::H1B], m2[G::H2B];
  {
    f32x16 xb[G::MB];
#pragma unroll
    for (int b = 0; b < G::MB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = b * 32 + nf_row(r, hi);
        xb[b][r] = (valid && q < a.m) ? yr[2 * q + par_c] : 0.f;
      }
    tile_to_scratch<G::MB>(sc + L::OFF_X, xb, l31, hi);
    f32x16 a1[G::H1B];
    dense_fwd<G::MB, G::H1B>(img + G::W1, img + G::B1, xb, a1, l31, hi);
#pragma unroll
    for (int b = 0; b < G::H1B; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[b][r] = nf_lrelu(a1[b][r]);
    sign_masks<G::H1B>(a1, m1);
    tile_to_scratch<G::H1B>(sc + L::OFF_A1, a1, l31, hi);
    f32x16 a2[G::H2B];
    dense_fwd<G::H1B, G::H2B>(img + G::W2, img + G::B2, a1, a2, l31, hi);
#pragma unroll
    for (int b = 0; b < G::H2B; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[b][r] = nf_lrelu(a2[b][r]);
    sign_masks<G::H2B>(a2, m2);
    tile_to_scratch<G::H2B>(sc + L::OFF_A2, a2, l31, hi);
    dense_fwd<G::H2B, G::CB>(img + G::W3, img + G::B3, a2, d3, l31, hi);  // T (phase T) or pre-tanh S
  }

  const float lb = valid ? (lbar ? lbar[j] : lbar_const) : 0.f;
#pragma unroll
  for (int b = 0; b < G::CB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = b * 32 + nf_row(r, hi);
      const bool ok = valid && p < a.c;
      const float y1 = ok ? yr[2 * p + a.par_t] : 0.f;
      const float g1 = ok ? gr[2 * p + a.par_t] : 0.f;
      if (!PHASE_S) {
        if (ok) yr[2 * p + a.par_t] = y1 - d3[b][r];  // u = x1 * exp(S)
        d3[b][r] = g1;                                 // T-bar = ybar1
      } else {
        const float s = tanhf(d3[b][r]);
        if (ok) {
          yr[2 * p + a.par_t] = y1 * expf(-s);  // x1
          gr[2 * p + a.par_t] = g1 * expf(s);   // x1bar
        }
        d3[b][r] = ok ? (g1 * y1 + lb) * (1.f - s * s) : 0.f;  // S-bar through tanh
      }
    }

  float *sd = sc + L::OFF_D;
  // ---- layer 3: dW3^T += a2 * d3^T ; d2 = (W3^T d3) .* lrelu'(a2)
  tile_to_scratch<G::CB>(sd, d3, l31, hi);
  wave_lds_fence();
  dw_accumulate<G::H2B, G::CB>(sc + L::OFF_A2, sd, acc.w3, acc.b3, l31, hi);
  f32x16 d2[G::H2B];
  dense_bwd_x<G::H2B, G::CB>(img + G::W3, d3, d2, l31, hi);
  apply_lrelu_grad<G::H2B>(d2, m2);
  wave_lds_fence();
  // ---- layer 2
  tile_to_scratch<G::H2B>(sd, d2, l31, hi);
  wave_lds_fence();
  dw_accumulate<G::H1B, G::H2B>(sc + L::OFF_A1, sd, acc.w2, acc.b2, l31, hi);
  f32x16 d1[G::H1B];
  dense_bwd_x<G::H1B, G::H2B>(img + G::W2, d2, d1, l31, hi);
  apply_lrelu_grad<G::H1B>(d1, m1);
  wave_lds_fence();
  // ---- layer 1
  tile_to_scratch<G::H1B>(sd, d1, l31, hi);
  wave_lds_fence();
  dw_accumulate<G::MB, G::H1B>(sc + L::OFF_X, sd, acc.w1, acc.b1, l31, hi);
  f32x16 g2[G::MB];
  dense_bwd_x<G::MB, G::H1B>(img + G::W1, d1, g2, l31, hi);
  wave_lds_fence();
#pragma unroll
  for (int b = 0; b < G::MB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = b * 32 + nf_row(r, hi);
      if (valid && q < a.m) gr[2 * q + par_c] += g2[b][r];
    }
}

template <class G>
__global__ __launch_bounds__(256, 1) void k_affine_bwd(CouplingArgs a, float *__restrict__ y, float *__restrict__ ybar,
                                                       const float *__restrict__ lbar, float lbar_const,
                                                       float *__restrict__ slab, long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *img = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  float *sc = lds + G::SIZE + wave * BwdLds<G>::SCRATCH;
  float *my_slab = slab + (long)blockIdx.x * slab_stride;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;

#pragma unroll 1
  for (int phase = 0; phase < 2; ++phase) {
    const NetDims &nd = phase == 0 ? a.t : a.s;
    stage_net<G>(img, a.theta, nd, tid, 256);
    __syncthreads();
    BwdAcc<G> acc;
    zero_acc(acc.w1, acc.b1);
    zero_acc(acc.w2, acc.b2);
    zero_acc(acc.w3, acc.b3);
#pragma unroll 1
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
      if (phase == 0)
        bwd_tile<G, false>(a, img, sc, acc, y, ybar, lbar, lbar_const, tile, l31, hi);
      else
        bwd_tile<G, true>(a, img, sc, acc, y, ybar, lbar, lbar_const, tile, l31, hi);
    }
    __syncthreads();  // every wave is done reading the weight image
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
        fold_acc(img + G::W1, img + G::B1, acc.w1, acc.b1, w == 0, l31, hi);
        fold_acc(img + G::W2, img + G::B2, acc.w2, acc.b2, w == 0, l31, hi);
        fold_acc(img + G::W3, img + G::B3, acc.w3, acc.b3, w == 0, l31, hi);
      }
      __syncthreads();
    }
    unstage_dense<G::S1>(img + G::W1, 32 * G::MB, my_slab + nd.w1, nd.m, nd.h1, tid, 256);
    unstage_dense<G::S2>(img + G::W2, 32 * G::H1B, my_slab + nd.w2, nd.h1, nd.h2, tid, 256);
    unstage_dense<G::S3>(img + G::W3, 32 * G::H2B, my_slab + nd.w3, nd.h2, nd.c, tid, 256);
    for (int i = tid; i < nd.h1; i += 256) my_slab[nd.b1 + i] = img[G::B1 + i];
    for (int i = tid; i < nd.h2; i += 256) my_slab[nd.b2 + i] = img[G::B2 + i];
    for (int i = tid; i < nd.c; i += 256) my_slab[nd.b3 + i] = img[G::B3 + i];
    __syncthreads();  // image is restaged next phase
  }
}

// ------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------
static inline int blocks32(int n) { return (n + 31) / 32; }

static int make_args(const nf_flow_desc *desc, int k, const float *theta, long N, CouplingArgs *out) {
  if (desc->n_hidden != 2) return NF_ERR_UNSUPPORTED;
  CouplingInfo ci = nf_coupling_info(desc, k);
  CouplingArgs a;
  a.theta = theta;
  a.d = desc->d; a.c = ci.c; a.m = ci.m; a.par_t = ci.par_t; a.N = N;
  const int h1 = desc->hdims[0], h2 = desc->hdims[1];
  a.s = make_net_dims(ci.theta_off, ci.m, h1, h2, ci.c);
  a.t = make_net_dims(ci.theta_off + net_param_count(ci.m, h1, h2, ci.c), ci.m, h1, h2, ci.c);
  *out = a;
  return NF_OK;
}

#define NF_GEO_DISPATCH(MBv, H1v, H2v, CBv, BODY)                          \
  if (mb == MBv && h1b == H1v && h2b == H2v && cb == CBv) {                 \
    using G = NetGeo<MBv, H1v, H2v, CBv>;                                   \
    BODY                                                                    \
  }

template <class G, bool INV>
static int launch_apply(nf_ctx *ctx, const CouplingArgs &a, const float *x, float *y, float *ladj, int accumulate) {
  const size_t lds = 2 * (size_t)G::SIZE * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_apply<G, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  long grid = (ntiles + 7) / 8;
  const long cap = 2L * ctx->num_cu;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  ProfScope ps(ctx, "affine_apply");
  hipLaunchKernelGGL((k_affine_apply<G, INV>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a, x, y, ladj, accumulate);
  return (int)hipGetLastError();
}

int nf_affine_apply(nf_ctx *ctx, const nf_flow_desc *desc, int k, bool inverse, const float *theta, const float *x,
                    long N, float *y, float *ladj, int accumulate) {
  CouplingArgs a;
  NF_TRY(make_args(desc, k, theta, N, &a));
  const int mb = blocks32(a.m > a.c ? a.m : a.c), h1b = blocks32(desc->hdims[0]), h2b = blocks32(desc->hdims[1]), cb = mb;
#define BODY_APPLY                                                              \
  return inverse ? launch_apply<G, true>(ctx, a, x, y, ladj, accumulate)        \
                 : launch_apply<G, false>(ctx, a, x, y, ladj, accumulate);
  NF_GEO_DISPATCH(1, 1, 1, 1, BODY_APPLY)
  NF_GEO_DISPATCH(1, 2, 2, 1, BODY_APPLY)
#undef BODY_APPLY
  return NF_ERR_UNSUPPORTED;
}

template <class G>
static int launch_bwd(nf_ctx *ctx, const CouplingArgs &a, float *y, float *ybar, const float *lbar, float lbar_const,
                      float *slab, long slab_stride, int grid) {
  const size_t lds = BwdLds<G>::BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_bwd<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  ProfScope ps(ctx, "affine_bwd");
  hipLaunchKernelGGL((k_affine_bwd<G>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, y, ybar, lbar, lbar_const,
                     slab, slab_stride);
  return (int)hipGetLastError();
}

// number of workgroups (= partial-gradient slabs) the reverse pass uses for a batch of N
int nf_affine_bwd_grid(nf_ctx *ctx, long N) {
  const long ntiles = (N + NF_TILE - 1) / NF_TILE;
  long grid = (ntiles + 3) / 4;
  if (grid > ctx->num_cu) grid = ctx->num_cu;
  if (grid < 1) grid = 1;
  return (int)grid;
}

int nf_affine_bwd(nf_ctx *ctx, const nf_flow_desc *desc, int k, const float *theta, float *y, float *ybar,
                  const float *lbar, float lbar_const, long N, float *slab, long slab_stride, int grid) {
  CouplingArgs a;
  NF_TRY(make_args(desc, k, theta, N, &a));
  const int mb = blocks32(a.m > a.c ? a.m : a.c), h1b = blocks32(desc->hdims[0]), h2b = blocks32(desc->hdims[1]), cb = mb;
#define BODY_BWD return launch_bwd<G>(ctx, a, y, ybar, lbar, lbar_const, slab, slab_stride, grid);
  NF_GEO_DISPATCH(1, 1, 1, 1, BODY_BWD)
  NF_GEO_DISPATCH(1, 2, 2, 1, BODY_BWD)
#undef BODY_BWD
  return NF_ERR_UNSUPPORTED;
}

bool nf_affine_supported(const nf_flow_desc *desc) {
  if (desc->n_hidden != 2) return false;
  const int c = (desc->d + 1) / 2;  // larger of the two partitions
  const int mb = blocks32(c), h1b = blocks32(desc->hdims[0]), h2b = blocks32(desc->hdims[1]), cb = mb;
  (void)cb;
  return (mb == 1 && h1b == 1 && h2b == 1) || (mb == 1 && h1b == 2 && h2b == 2);
}
