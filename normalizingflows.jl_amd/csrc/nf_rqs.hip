// nf_rqs.hip -- NeuralSplineCoupling (rational-quadratic spline coupling) kernels for gfx950.
//
// Reference: src/flows/neuralspline.jl:65-140 (get_nsc_params + rqs_forward / rqs_inverse call
// sites); spline arithmetic from MonotonicSplines.jl 0.3.3 (rqs_params_from_nn, rqs_forward,
// rqs_inverse) restated in oracle/nf_oracle.py and SURVEY.md App. A.4:
//   raw = nn(x2), (3K-1)*c x N; per transformed dim: K widths, K heights, K-1 derivatives;
//   knots = -B + 2B * cumsum(softmax(.)), boundary derivatives 1, interior softplus;
//   identity (zero log-derivative) outside [-B, B].
//
// Design.  The conditioner MLP uses the register-chained fp32 MFMA primitives of nf_mfma.h.
// The raw parameter tensor ((3K-1)*c x N, 193 MB at the cfg-3 size in the reference) never
// exists: the output layer is evaluated in CHUNKS of 2*QCH transformed dims, and the rows of W3
// are PERMUTED when the weights are packed so that, in the MFMA accumulator layout, the
// 3K-1 parameters of a dim land in consecutive registers of the one lane that also holds
// that dim's x1 value:
//   transformed dim p  <->  lane half hi = (p >> 2) & 1,  local index q = (p & 3) + 4 * (p >> 3)
//   (exactly the register index of x1 in the C layout), chunk = q / QCH,
//   parameter prm of local dim ql = q % QCH  ->  slot = ql * P + prm  ->  accumulator block
//   slot / 16, register slot % 16.
// The spline (softmax / cumsum / bin search / rational quadratic, and its hand-derived
// reverse pass) then runs entirely in registers, one (dim, sample) per lane at a time.
#include <cstdlib>
#include <type_traits>

#include "nf_common.h"
#include "nf_mfma.h"
// k_rqs_bwd_coop6's matrix instructions on the wave's weight triples.  Round 6 tried them as inline asm with the A operand in
// the accumulation half of the register file (what VERDICT r5 item 2 asked for: hipcc keeps the 144 weight registers there
// and copies each operand into architectural registers next to the MFMA that reads it -- 289 v_accvgpr_read in the kernel):
// the copies went away (289 -> 135, none in the chunk phase; scratch 28 -> 0 bytes), the kernel gained 0.5 % (107.8 -> 107.3 us)
// -- and the gradient of golden nsf_d32_k8 came out 4 % wrong in the NEXT build, whose only difference was the address the
// kernel was linked at (an unrelated kernel removed from the file; identical ISA, tools/isa_stats.py).  hipcc's hazard
// recognizer does not look inside inline asm, so every wait state between a vector instruction and a matrix instruction that
// shares a register with it is the author's to supply, and the set nf_mfma_settle() supplied (12 states behind an accumulation
// chain) was evidently not the whole set.  Not shipped: the builtin, whose hazards the compiler handles.
#define RQS6_MFMA_W(a, b, c) nf_mfma_bf16(a, b, c)
#define RQS6_SETTLE(c) ((void)0)
#include "nf_philox.h"

template <int MB_, int H1B_, int H2B_, int K_, int NCH_, int QCH_ = 2>
struct RqsGeo {
  static constexpr int MB = MB_, H1B = H1B_, H2B = H2B_, CB = MB_, K = K_, NCH = NCH_;
  static constexpr int P = 3 * K - 1;            // raw parameters per transformed dim
  static constexpr int QCH = QCH_;               // local dims per lane half per chunk
  static constexpr int OBC = (QCH * P + 15) / 16;  // accumulator blocks per chunk
  static constexpr int OB3 = OBC * NCH;          // output blocks of the last layer
  static constexpr int NCOLS = OB3 * 32;
  static constexpr int CMAX = 2 * QCH * NCH;     // transformed dims covered
  static constexpr int S1 = 32 * H1B + NF_IMG_PAD, S2 = 32 * H2B + NF_IMG_PAD, S3 = NCOLS + NF_IMG_PAD;
  static constexpr int W1 = 0;
  static constexpr int B1 = W1 + 32 * MB * S1;
  static constexpr int W2 = B1 + 32 * H1B;
  static constexpr int B2 = W2 + 32 * H1B * S2;
  static constexpr int W3 = B2 + 32 * H2B;
  static constexpr int B3 = W3 + 32 * H2B * S3;
  static constexpr int END = B3 + NCOLS;
  static constexpr int SIZE = ((END + 3) / 4) * 4;
};

struct RqsDims {
  int m, h1, h2, c;              // conditioner fan-in, hidden sizes, transformed dims
  long w1, b1, w2, b2, w3, b3;   // theta offsets (Optimisers.destructure order)
};
__host__ __device__ inline long rqs_param_count(int m, int h1, int h2, int c, int P) {
  return (long)m * h1 + h1 + (long)h1 * h2 + h2 + (long)h2 * c * P + (long)c * P;
}
__host__ __device__ inline RqsDims make_rqs_dims(long off, int m, int h1, int h2, int c, int P) {
  RqsDims n;
  n.m = m; n.h1 = h1; n.h2 = h2; n.c = c;
  n.w1 = off; n.b1 = n.w1 + (long)m * h1;
  n.w2 = n.b1 + h1; n.b2 = n.w2 + (long)h1 * h2;
  n.w3 = n.b2 + h2; n.b3 = n.w3 + (long)h2 * c * P;
  return n;
}

// theta index of element e of the padded / permuted image, or -1 for padding
template <class G>
__device__ __forceinline__ long rqs_image_theta_index(const RqsDims &nd, int e) {
  if (e < G::W2) {
    const int r = e - G::W1;
    if (r < 32 * G::MB * G::S1) {
      const int i = r / G::S1, o = r - i * G::S1;
      return (i < nd.m && o < nd.h1) ? nd.w1 + (long)i * nd.h1 + o : -1;
    }
    const int o = r - 32 * G::MB * G::S1;
    return o < nd.h1 ? nd.b1 + o : -1;
  }
  if (e < G::W3) {
    const int r = e - G::W2;
    if (r < 32 * G::H1B * G::S2) {
      const int i = r / G::S2, o = r - i * G::S2;
      return (i < nd.h1 && o < nd.h2) ? nd.w2 + (long)i * nd.h2 + o : -1;
    }
    const int o = r - 32 * G::H1B * G::S2;
    return o < nd.h2 ? nd.b2 + o : -1;
  }
  if (e >= G::END) return -1;
  int i = -1, col;
  if (e < G::B3) {
    const int r = e - G::W3;
    i = r / G::S3;
    col = r - i * G::S3;
    if (col >= G::NCOLS || i >= nd.h2) return -1;
  } else {
    col = e - G::B3;
  }
  // column -> (chunk, block, row-in-block) -> (lane half, register) -> slot -> (local dim, param)
  const int ch = col / (G::OBC * 32), cc = col - ch * (G::OBC * 32);
  const int b = cc >> 5, rr = cc & 31;
  const int hi = (rr >> 2) & 1, reg = (rr & 3) + 4 * (rr >> 3);
  const int slot = b * 16 + reg;
  if (slot >= G::QCH * G::P) return -1;
  const int ql = slot / G::P, prm = slot - ql * G::P;
  const int q = ch * G::QCH + ql;
  const int p = (q & 3) + 8 * (q >> 2) + 4 * hi;
  if (p >= nd.c) return -1;
  const int orig = p * G::P + prm;  // row of the reference's output layer
  const int nout = nd.c * G::P;
  return i >= 0 ? nd.w3 + (long)i * nout + orig : nd.b3 + orig;
}

struct RqsPackArgs {
  int d, h1, h2, ncoup;
  long pair_params, odd_params;
};

template <class G>
__device__ __forceinline__ RqsDims rqs_dims_of(const RqsPackArgs &p, int k) {
  const int c = (k & 1) ? p.d / 2 : (p.d + 1) / 2, m = p.d - c;
  const long off = (long)(k >> 1) * p.pair_params + ((k & 1) ? p.odd_params : 0);
  return make_rqs_dims(off, m, p.h1, p.h2, c, G::P);
}

template <class G>
__global__ __launch_bounds__(256) void k_rqs_pack(RqsPackArgs p, const float *__restrict__ theta,
                                                  float *__restrict__ out) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)p.ncoup * G::SIZE) return;
  const int k = (int)(gid / G::SIZE), e = (int)(gid - (long)k * G::SIZE);
  const long ti = rqs_image_theta_index<G>(rqs_dims_of<G>(p, k), e);
  out[gid] = ti >= 0 ? theta[ti] : 0.f;
}

// 64 image elements per block, the slabs split over the block's four waves (an HBM stream of ~120 MB at cfg 3: with
// one element per thread over all slabs too few waves were in flight -- see k_reduce_image_slabs, nf_pack.h)
template <class G>
__global__ __launch_bounds__(256) void k_rqs_reduce_slabs(RqsPackArgs p, const float *__restrict__ slab, int nslab,
                                                          long slab_stride, float *__restrict__ g) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long gid = (long)blockIdx.x * 64 + lane;
  long ti = -1;
  if (gid < (long)p.ncoup * G::SIZE) {
    const int k = (int)(gid / G::SIZE), e = (int)(gid - (long)k * G::SIZE);
    ti = rqs_image_theta_index<G>(rqs_dims_of<G>(p, k), e);
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (ti >= 0) {  // wave q sums slabs q, q + 4, q + 8, ...
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

// The fused epilogue of nf_elbo_step for spline couplings (round 6; the structure of k_affine_epilogue, nf_pack.h): every thread of
// wave 0 owns one element of the padded fp32 images -- slab sum (k_rqs_reduce_slabs's order, bit for bit), gradient, Adam
// (Optimisers.update!, src/optimize.jl:99), the packed image of the UPDATED theta and the block's partial of sum g^2; block 0 also
// finishes the forward's loss partials.  Replaces four launches of the step (loss sum, slab reduction, Adam, the next step's pack).
struct RqsEpiArgs {
  const float *slab;
  int nslab;
  long slab_stride;
  float *g;             // [P + 2]: gradient, loss, gradient norm
  long P;
  const double *lpart;  // loss partials of the forward launch, finished into g[P]
  int nlpart;
  float *theta, *m, *v, *wimg;
  float lr, b1, b2, eps, c1, c2;  // c1 = 1 - b1^t, c2 = 1 - b2^t, computed on the host as nf_launch_adam does
  double *gpart;        // [gridDim.x] partial sums of g^2
};
template <class G>
__global__ __launch_bounds__(256) void k_rqs_epilogue(RqsPackArgs p, RqsEpiArgs a) {
  if (a.lpart && blockIdx.x == 0) {
    __shared__ double sm[4];
    double c = 0.0;
    for (int i = threadIdx.x; i < a.nlpart; i += 256) c += a.lpart[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) a.g[a.P] = (float)((sm[0] + sm[1]) + (sm[2] + sm[3]));
  }
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long gid = (long)blockIdx.x * 64 + lane;
  long ti = -1;
  if (gid < (long)p.ncoup * G::SIZE) {
    const int k = (int)(gid / G::SIZE), e = (int)(gid - (long)k * G::SIZE);
    ti = rqs_image_theta_index<G>(rqs_dims_of<G>(p, k), e);
  }
  float th = 0.f, mi = 0.f, vi = 0.f;  // Adam's operands are requested before the slab stream
  if (q == 0 && ti >= 0) {
    th = a.theta[ti];
    mi = a.m[ti];
    vi = a.v[ti];
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (ti >= 0) {  // wave q sums slabs q, q + 4, q + 8, ... (k_rqs_reduce_slabs)
    int s = q;
    for (; s + 12 < a.nslab; s += 16) {
      a0 += a.slab[(long)s * a.slab_stride + gid];
      a1 += a.slab[(long)(s + 4) * a.slab_stride + gid];
      a2 += a.slab[(long)(s + 8) * a.slab_stride + gid];
      a3 += a.slab[(long)(s + 12) * a.slab_stride + gid];
    }
    for (; s < a.nslab; s += 4) a0 += a.slab[(long)s * a.slab_stride + gid];
  }
  part[q][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (q != 0) return;
  double gg = 0.0;
  if (ti >= 0) {
    const float gsum = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    a.g[ti] = gsum;
    nf_adam_elem<float>(th, mi, vi, gsum, a.lr, a.b1, a.b2, a.eps, a.c1, a.c2);
    a.m[ti] = mi;
    a.v[ti] = vi;
    a.theta[ti] = th;
    a.wimg[gid] = th;  // padding elements of the image stay zero from the first pack
    gg = (double)gsum * (double)gsum;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gg += __shfl_xor(gg, o, 64);
  if (lane == 0) a.gpart[blockIdx.x] = gg;
}

#include "nf_rqs_elem.h"  // the spline of one (dimension, sample): build_knots, find_bin, rqs_fwd_elem / rqs_inv_elem / rqs_bwd_elem

// pull the P raw parameters of local dim `ql` of a chunk out of its accumulator blocks
template <class G>
__device__ __forceinline__ void chunk_get(const f32x16 (&out)[G::OBC], int ql, float *raw) {
#pragma unroll
  for (int prm = 0; prm < G::P; ++prm) {
    const int slot = ql * G::P + prm;
    raw[prm] = out[slot / 16][slot % 16];
  }
}
template <class G>
__device__ __forceinline__ void chunk_put(f32x16 (&out)[G::OBC], int ql, const float *v) {
#pragma unroll
  for (int prm = 0; prm < G::P; ++prm) {
    const int slot = ql * G::P + prm;
    out[slot / 16][slot % 16] = v[prm];
  }
}

// the output layer of a coupling as bf16 triples (six-term products, nf_mfma.h "B6"): image layout and its pack kernel; used by
// the chain kernel's B6 variant below and by the cooperative reverse kernel k_rqs_bwd_coop6 (where the design is described)
template <class G>
struct RqsB6Geo {  // 16-byte units; per coupling [chunk][k-group][component][half][row]
  static constexpr int FROWS = G::OBC * 32, FKG = 2 * G::H2B;  // recompute: rows = the chunk's columns, k over a2's features
  static constexpr int TROWS = 32 * G::H2B, TKG = 2 * G::OBC;  // dX3: rows = a2's features, k over the chunk's columns
  static constexpr int F_CH = FKG * 3 * 2 * FROWS, T_CH = TKG * 3 * 2 * TROWS;
  static constexpr int OFF_T = G::NCH * F_CH;
  static constexpr int U4 = G::NCH * (F_CH + T_CH);
  static constexpr size_t BYTES = (size_t)U4 * 16;
};

template <class G>
__global__ __launch_bounds__(256) void k_rqs_b6_from_images(int nimg, const float *__restrict__ wimg, nf_u32x4 *__restrict__ out) {
  using B = RqsB6Geo<G>;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)nimg * B::U4) return;
  const int img = (int)(gid / B::U4);
  int e = (int)(gid - (long)img * B::U4);
  const float *src = wimg + (size_t)img * G::SIZE + G::W3;
  const bool tr = e >= B::OFF_T;
  if (tr) e -= B::OFF_T;
  const int rows = tr ? B::TROWS : B::FROWS, nkg = tr ? B::TKG : B::FKG;
  const int row = e % rows, hi = (e / rows) & 1, comp = (e / (2 * rows)) % 3, kg = (e / (6 * rows)) % nkg, ch = e / (6 * rows * nkg);
  unsigned short part[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kf = 16 * kg + (j & 3) + 8 * (j >> 2) + 4 * hi;  // k-slot (kg, hi, j) in the C-layout register order
    const float w = tr ? src[row * G::S3 + ch * G::OBC * 32 + kf]    // [in = row][column kf of the chunk]
                       : src[kf * G::S3 + ch * G::OBC * 32 + row];   // [in = kf][column row of the chunk]
    unsigned short h, m, l;
    nf_split1(w, h, m, l);
    part[j] = comp == 0 ? h : comp == 1 ? m : l;
  }
  nf_u32x4 q;
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) q[pp] = (unsigned)part[2 * pp] | ((unsigned)part[2 * pp + 1] << 16);
  out[gid] = q;
}

// ---------------------------------------------------------------------------------------
// whole-flow forward / inverse in one launch (same structure as k_affine_chain)
// ---------------------------------------------------------------------------------------
struct RqsChainArgs {
  RqsTape tape;       // spline tape to leave behind (training), or null pointers
  const float *wimg;  // [coupling][G::SIZE]
  const nf_u32x4 *wimg6;  // [coupling][RqsB6Geo<G>::U4]: the output layers as bf16 triples (B6 variant), or null
  int d, ncoup;
  int k_only;  // -1: every coupling; otherwise only the coupling with this flat index
  float B;
  long N;
  long long *trace;  // NF_KERNEL_TRACE builds: clock stamps for tools/trace_rqs_chain.py ([64 + 28 g ..], g = the workgroup's first two tile groups)
};

// tape: buffer descriptor of this (tile, coupling)'s slot of the spline tape; extent 0 (no tape wanted, or an idle wave)
// drops the stores in hardware, so the step has no branch on it and needs no per-lane 64-bit address
// B6: the output layer as six-term bf16 products.  `img` then holds only layers 1-2 at their image offsets and the output
// layer's bias at G::W3 (the "small image"); `w6` = the coupling's forward triples [chunk][k-group][h | m | l][half][row] in
// LDS, which the workgroup's DMA may still be bringing in while layers 1-2 run: the barrier in the middle is that hand-over.
template <class G, bool INVERSE, bool B6 = false>
__device__ __forceinline__ float rqs_coupling_step(const float *__restrict__ img, f32x16 (&x1)[G::CB],
                                                   const f32x16 (&xb)[G::MB], int c, int m, float B, int l31, int hi,
                                                   __amdgpu_buffer_rsrc_t tape, const nf_u32x4 *__restrict__ w6 = nullptr) {
  const int tvoff = (hi * 32 + l31) * 4;
  f32x16 a2[G::H2B];
  {
    f32x16 a1[G::H1B];
#ifdef RQS_COOP_R3
    dense_fwd<G::MB, G::H1B>(img + G::W1, img + G::B1, xb, a1, l31, hi);
#else
    dense_fwd_dyn<G::MB, G::H1B>(img + G::W1, img + G::B1, xb, a1, l31, hi, (m + 7) >> 3);
#endif
#pragma unroll
    for (int b = 0; b < G::H1B; ++b)
      nf_lrelu16(a1[b]);
    dense_fwd<G::H1B, G::H2B>(img + G::W2, img + G::B2, a1, a2, l31, hi);
#pragma unroll
    for (int b = 0; b < G::H2B; ++b)
      nf_lrelu16(a2[b]);
  }
  float lsum = 0.f;
  SplitC<G::H2B> a2s;  // (B6) a2 as bf16 triples, split once for the four chunks
  if constexpr (B6) {
    split_C<G::H2B>(a2, a2s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the triples' DMA have landed ...
    __syncthreads();                                  // ... and every other wave's
  }
#pragma unroll
  for (int ch = 0; ch < G::NCH; ++ch) {
    f32x16 out[G::OBC];
    if constexpr (B6) {
      using B6G = RqsB6Geo<G>;
      const float *b3c = img + G::W3 + ch * G::OBC * 32;
#pragma unroll
      for (int ob = 0; ob < G::OBC; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ob][r] = b3c[ob * 32 + nf_row(r, hi)];
      const nf_u32x4 *wl = w6 + ch * B6G::F_CH + hi * B6G::FROWS + l31;
#pragma unroll
      for (int kg = 0; kg < B6G::FKG; ++kg)
#pragma unroll
        for (int ob = 0; ob < G::OBC; ++ob) {  // smallest terms first, as dense_fwd_b6
          const nf_u32x4 wh = wl[((kg * 3 + 0) * 2) * B6G::FROWS + ob * 32], wm = wl[((kg * 3 + 1) * 2) * B6G::FROWS + ob * 32],
                         wlo = wl[((kg * 3 + 2) * 2) * B6G::FROWS + ob * 32];
          out[ob] = nf_mfma_bf16(wlo, a2s.h[kg], out[ob]);
          out[ob] = nf_mfma_bf16(wh, a2s.l[kg], out[ob]);
          out[ob] = nf_mfma_bf16(wm, a2s.m[kg], out[ob]);
          out[ob] = nf_mfma_bf16(wm, a2s.h[kg], out[ob]);
          out[ob] = nf_mfma_bf16(wh, a2s.m[kg], out[ob]);
          out[ob] = nf_mfma_bf16(wh, a2s.h[kg], out[ob]);
        }
    } else {
      dense_fwd<G::H2B, G::OBC, G::S3>(img + G::W3 + ch * G::OBC * 32, img + G::B3 + ch * G::OBC * 32, a2, out, l31, hi);
    }
#pragma unroll
    for (int ql = 0; ql < G::QCH; ++ql) {
      const int q = ch * G::QCH + ql;            // register index of x1 (block q / 16)
      const int p = (q & 3) + 8 * (q >> 2) + 4 * hi;
      float raw[G::P];
      chunk_get<G>(out, ql, raw);
      Knots<G::K> kn;
      build_knots<G::K>(raw, B, kn);
      const float v = x1[q / 16][q % 16];
      float logd = 0.f, res, xi;
      unsigned code;
      if (INVERSE) {
        Bin<G::K> bn;
        res = rqs_inv_elem<G::K>(kn, v, logd, bn, xi);
        code = bn.code;
      } else {
        res = rqs_fwd_elem<G::K>(kn, v, logd, code, xi);
      }
      const bool ok = p < c;  // padded dims: keep the zero, contribute nothing
      x1[q / 16][q % 16] = ok ? res : v;
      lsum += ok ? logd : 0.f;
      const float tv = rqs_tape_encode(code, xi);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tv), tape, tvoff, q * 256, 0);
    }
  }
  return lsum;
}

// FUSED (forward only): the ELBO forward of the training step in the same launch, as k_affine_chain<.., FUSED> does
// (nf_coupling.hip): the tile's base draws are generated in registers (Philox4x32-10 + Box-Muller, the counters of
// k_base_sample_tiled), log q0 and the accumulated log|det J| never touch memory, and after the last coupling the
// diagonal-Gaussian target, ybar = gscale * grad log p(y) and the workgroup's partial sum of pscale * elbo_j come out
// of the registers (src/objectives/elbo.jl:65-70,93-97).
struct RqsFusedArgs {
  uint32_t k0, k1, stream;
  uint64_t off;           // global index of this shard's first sample
  const float *mu, *var;  // diagonal-Gaussian target (test/flow.jl:43-46)
  float *gt;              // ybar out (tiled), or nullptr
  float gscale;
  double *partial;        // [gridDim.x] out
  double pscale;
};

// LDS map of the chain kernel, in floats: the weight area, then (FUSED) the target parameters and the per-wave sums.
//   fp32 form: two whole images [2][G::SIZE], the next coupling's brought in by DMA while the current one computes.
//   B6 form (round 5): two SMALL images [2][SMALL] (layers 1-2 at their image offsets, the output layer's bias at G::W3) and
//   ONE copy of the output layer as bf16 triples (74 KB at K = 8: two would not fit).  The next coupling's triples can only
//   be requested once every wave is done with the current ones (the barrier that ends a coupling), and are needed after the
//   next coupling's layers 1-2 -- whose 24 fp32 MFMAs + activations cover most of the transfer; the barrier in the middle of
//   rqs_coupling_step completes the hand-over.  Two workgroup barriers per coupling instead of one, 144 bf16 MFMAs of 32
//   clocks instead of 192 fp32 ones of 64 per tile and coupling.
template <class G, bool B6>
struct RqsChainLds {
  static constexpr int SMALL = G::W3 + G::NCOLS;
  static constexpr int F_U4 = G::NCH * RqsB6Geo<G>::F_CH;  // the forward triples of one coupling, 16-byte units
  static constexpr int WAREA = B6 ? 2 * SMALL + 4 * F_U4 : 2 * G::SIZE;
  static constexpr size_t BYTES = (size_t)(WAREA + 2 * 64 * G::CB + 2) * sizeof(float) + 8 * sizeof(double);
};
template <class G, bool INVERSE, bool FUSED = false, bool B6 = false>
__global__ __launch_bounds__(512) void k_rqs_chain(RqsChainArgs a, float *xt, float *__restrict__ ladj, RqsFusedArgs fa) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using CL = RqsChainLds<G, B6>;
  constexpr int NV4 = G::SIZE / 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const long ngroups = (ntiles + 7) / 8;
  auto coupling_at = [&](int s) { return INVERSE ? s : a.ncoup - 1 - s; };
  // global -> LDS by DMA (buffer_load ... lds), 1 KB pieces dealt round-robin to the eight waves; complete at the issuing wave's
  // next vmcnt(0) + a workgroup barrier
  typedef __attribute__((address_space(3))) void lds_void_t;
  auto dma = [&](float *dst, const void *src, int nbytes) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(src), 0, nbytes, 0x00020000);
    const int np = (nbytes + 1023) / 1024;
    for (int p = wave; p < np; p += 8)
      if (p * 1024 + lane * 16 < nbytes)  // the last piece may be partial: its idle lanes must not write
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(dst + p * 256), 16, lane * 16, p * 1024, 0, 0);
  };
  float *const t6 = lds + 2 * CL::SMALL;  // (B6) the triples
  auto dma_small = [&](float *dst, int k) {  // layers 1-2 | output bias
    dma(dst, a.wimg + (long)k * G::SIZE, G::W3 * 4);
    dma(dst + G::W3, a.wimg + (long)k * G::SIZE + G::B3, G::NCOLS * 4);
  };
  if constexpr (B6) {
    dma_small(lds, coupling_at(0));
    dma(t6, a.wimg6 + (long)coupling_at(0) * RqsB6Geo<G>::U4, CL::F_U4 * 16);
  } else {
    const float4 *src = reinterpret_cast<const float4 *>(a.wimg + (long)coupling_at(0) * G::SIZE);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    for (int i = tid; i < NV4; i += 512) dst[i] = src[i];
  }
  // FUSED: target parameters by feature, zero padded: tmu[f], tiv[f] = 1/var[f]; tc0 = d log 2pi + sum log var
  constexpr int TP = 64 * G::CB;
  float *tmu = lds + CL::WAREA, *tiv = tmu + TP, *tc0 = tiv + TP;
  double *wsum = reinterpret_cast<double *>(tc0 + 2);  // [8] per-wave partial sums (G::SIZE and TP are even)
  if (FUSED) {
    for (int i = tid; i < TP; i += 512) {
      tmu[i] = i < a.d ? fa.mu[i] : 0.f;
      tiv[i] = i < a.d ? 1.f / fa.var[i] : 0.f;
    }
    if (tid == 0) {
      float c = 1.8378770664093453f * (float)a.d;
      for (int i = 0; i < a.d; ++i) c += logf(fa.var[i]);
      tc0[0] = c;
    }
  }
  double wg_total = 0.0;
  __syncthreads();
  const int c_odd = (a.d + 1) / 2, c_even = a.d / 2;  // mask 1:2:d / 2:2:d
  int buf = 0;
  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
#ifdef NF_KERNEL_TRACE
    if (a.trace && blockIdx.x == 0 && tid == 0 && (grp - blockIdx.x) / gridDim.x < 2) {
      __builtin_amdgcn_sched_barrier(0);
      a.trace[64 + 28 * (int)((grp - blockIdx.x) / gridDim.x)] = clock64();
    }
#endif
    const long tile = grp * 8 + wave;
    const bool live = tile < ntiles;
    const long tl = live ? tile : 0;
    const long j = tl * NF_TILE + l31;
    const bool valid = live && j < a.N;
    const TileIO io = make_tile_io(xt, tl, a.d, l31, hi);
    f32x16 E[G::CB], O[G::MB];
    float zz = 0.f;  // FUSED: this lane's share of ||x||^2
    if (!FUSED) {
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = tile_load(io, tile_soff(b, r, 0));
          const float o = tile_load(io, tile_soff(b, r, 1));
          E[b][r] = valid ? e : 0.f;
          O[b][r] = valid ? o : 0.f;
        }
    } else {
      // registers (b, 4q..4q+3) of E and O are features base..base+7, base = 64b + 16q + 8hi:
      // Philox groups base/4 (-> E0 O0 E1 O1) and base/4 + 1 (-> E2 O2 E3 O3)
      const uint64_t gj = fa.off + (uint64_t)j;
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int base = 64 * b + 16 * q + 8 * hi;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int g = base / 4 + h;
            U4 c = {(uint32_t)gj, (uint32_t)(gj >> 32), (uint32_t)g, fa.stream};
            const U4 rr = philox4x32_10(c, fa.k0, fa.k1);
            float z[4];
            box_muller<float>(rr.x, rr.y, z[0], z[1]);
            box_muller<float>(rr.z, rr.w, z[2], z[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool in = (4 * g + e < a.d) && valid;  // features >= d and padding samples stay 0
              const float v = in ? z[e] : 0.f;
              zz += v * v;
              if (e & 1) O[b][4 * q + 2 * h + (e >> 1)] = v;
              else E[b][4 * q + 2 * h + (e >> 1)] = v;
            }
          }
        }
    }
    float lsum = 0.f;
    const bool more_groups = grp + gridDim.x < ngroups;
#ifdef NF_KERNEL_TRACE
    const int gidx = (int)((grp - blockIdx.x) / gridDim.x);
    long long *trc = (a.trace && blockIdx.x == 0 && tid == 0 && gidx < 2) ? a.trace + 64 + 28 * gidx : nullptr;
#define RC_STAMP(slot) do { if (trc) { __builtin_amdgcn_sched_barrier(0); trc[slot] = clock64(); } } while (0)
#else
#define RC_STAMP(slot) do { } while (0)
#endif
    RC_STAMP(1);  // (the draws are done; [0] is stamped at the group's top)
#pragma unroll 1
    for (int s = 0; s < a.ncoup; s += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int pos = s + half;
        const bool have_next = pos + 1 < a.ncoup || more_groups;
        const int knext = coupling_at(pos + 1 < a.ncoup ? pos + 1 : 0);
        if (have_next) {  // LDS-DMA straight into the other image buffer (as in k_affine_chain); complete at the barrier below
          if constexpr (B6) dma_small(lds + (buf ^ 1) * CL::SMALL, knext);
          else dma(lds + (buf ^ 1) * G::SIZE, a.wimg + (long)knext * G::SIZE, G::SIZE * 4);
        }
        const float *img = lds + buf * (B6 ? CL::SMALL : G::SIZE);
        // forward: position 0 is the last flat coupling (odd index, mask 2:2:d): x1 = O
        if (a.k_only < 0 || a.k_only == coupling_at(pos)) {
          constexpr int ROWS = G::NCH * G::QCH;
          const long slot = tl * a.ncoup + coupling_at(pos);
          const __amdgpu_buffer_rsrc_t tp = __builtin_amdgcn_make_buffer_rsrc(
              a.tape.base + (a.tape.base ? slot * (ROWS * 64) : 0), 0, (a.tape.base && live) ? ROWS * 256 : 0, 0x00020000);
          if (INVERSE ? (half == 1) : (half == 0))
            lsum += rqs_coupling_step<G, INVERSE, B6>(img, O, E, c_even, a.d - c_even, a.B, l31, hi, tp, reinterpret_cast<const nf_u32x4 *>(t6));
          else
            lsum += rqs_coupling_step<G, INVERSE, B6>(img, E, O, c_odd, a.d - c_odd, a.B, l31, hi, tp, reinterpret_cast<const nf_u32x4 *>(t6));
        }
        if (pos < 8) RC_STAMP(2 + 3 * pos);
        __syncthreads();
        if (pos < 8) RC_STAMP(3 + 3 * pos);
        if constexpr (B6) {  // every wave is done with this coupling's triples: the next coupling's may come
          if (have_next) dma(t6, a.wimg6 + (long)knext * RqsB6Geo<G>::U4, CL::F_U4 * 16);
        }
        if (pos < 8) RC_STAMP(4 + 3 * pos);
        buf ^= 1;
      }
    }
    if (live) {
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          tile_store(io, tile_soff(b, r, 0), E[b][r]);
          tile_store(io, tile_soff(b, r, 1), O[b][r]);
        }
      lsum += __shfl_xor(lsum, 32);
      if (!FUSED) {
        if (hi == 0 && valid) ladj[j] = lsum;  // inverse: rqs_inv_elem already accumulates -log dy/dx
      }
    }
    if (FUSED) {
      // elbo_j = log p(y_j) - log q0(x_j) + ladj_j ;  ybar = gscale * grad log p(y)
      const TileIO gio = make_tile_io(fa.gt ? fa.gt : xt, tl, a.d, l31, hi);
      float t = 0.f;
      // ONE lane-dependent base per table, made opaque here: the tables sit beyond the 64 KB a ds_read's immediate offset
      // reaches, and hipcc otherwise materialises an address register per element in the kernel's prologue and keeps the
      // thirty-two of them across the tile loop (fifteen in scratch: 64 bytes, round 5)
      const float *tm = tmu + 8 * hi, *tv = tiv + 8 * hi;
      asm volatile("" : "+v"(tm), "+v"(tv));
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int fe = 2 * (b * 32 + (r & 3) + 8 * (r >> 2));  // + 8 hi: in the bases
          const float re = E[b][r] - tm[fe], ro = O[b][r] - tm[fe + 1];
          const float ge = re * tv[fe], go = ro * tv[fe + 1];
          t += re * ge + ro * go;
          if (fa.gt && live) {
            tile_store(gio, tile_soff(b, r, 0), valid ? -fa.gscale * ge : 0.f);
            tile_store(gio, tile_soff(b, r, 1), valid ? -fa.gscale * go : 0.f);
          }
        }
      t += __shfl_xor(t, 32);
      zz += __shfl_xor(zz, 32);
      double contrib = 0.0;
      if (hi == 0 && valid) {
        const float logq = (float)(-0.5 * 1.8378770664093453 * a.d) - 0.5f * zz;
        const float e = -0.5f * (tc0[0] + t) - logq + lsum;
        contrib = fa.pscale * (double)e;
      }
#pragma unroll
      for (int sft = 16; sft >= 1; sft >>= 1) contrib += __shfl_xor(contrib, sft);  // lanes 0..31 carry the terms
      if (lane == 0) wsum[wave] = contrib;
      __syncthreads();
      if (tid == 0) {
        double sgrp = 0.0;
        for (int w = 0; w < 8; ++w) sgrp += wsum[w];
        wg_total += sgrp;
      }
      __syncthreads();
    }
    RC_STAMP(27);
  }
  if (FUSED && tid == 0) fa.partial[blockIdx.x] = wg_total;
}

// ---------------------------------------------------------------------------------------
// reverse pass of one coupling (invertible recompute), one net
// ---------------------------------------------------------------------------------------
struct RqsBwdArgs {
  RqsTape tape;      // the forward's spline tape (required)
  int k, ncoup;      // this coupling's flat index / couplings of the flow (tape addressing)
  const float *img;  // packed image of this coupling
  const nf_u32x4 *img6;  // its output layer as bf16 triples (RqsB6Geo), or null
  int d, c, m, par_t;
  float B;
  long N;
  long long *trace;
};

template <class G>
struct RqsAcc {
  f32x16 w1[G::MB][G::H1B];
  f32x16 w2[G::H1B][G::H2B];
  f32x16 w3[G::H2B][G::OB3];
  float b1[G::H1B], b2[G::H2B], b3[G::OB3];
};

template <class G>
struct RqsLds {
  static constexpr int DROWS = (G::H1B > G::H2B ? G::H1B : G::H2B) > 2 ? (G::H1B > G::H2B ? G::H1B : G::H2B) : 2;
  static constexpr int OFF_X = 0;
  static constexpr int OFF_A1 = OFF_X + G::MB * 32 * NF_TS;
  static constexpr int OFF_A2 = OFF_A1 + G::H1B * 32 * NF_TS;
  static constexpr int OFF_D = OFF_A2 + G::H2B * 32 * NF_TS;
  static constexpr int SCRATCH = OFF_D + DROWS * 32 * NF_TS;
  static constexpr int WAVES = 4;
  static constexpr size_t BYTES = (size_t)(G::SIZE + WAVES * SCRATCH) * sizeof(float);
};

template <int IB, int OB>
__device__ __forceinline__ void rqs_zero(f32x16 (&a)[IB][OB], float (&b)[OB]) {
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[i][o][r] = 0.f;
#pragma unroll
  for (int o = 0; o < OB; ++o) b[o] = 0.f;
}

template <int IB, int OB, int S>
__device__ __forceinline__ void rqs_fold(float *__restrict__ w, float *__restrict__ b, const f32x16 (&a)[IB][OB],
                                         const float (&bs)[OB], bool first, int l31, int hi) {
  // a block's sixteen partial sums are read in one go, then written back: as one read-modify-write after the other every LDS
  // round trip is exposed (round 5, tools/trace_deep_bwd.py on the same fold in nf_deep.hip)
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

#ifndef RQS_BWD_LAZY
#define RQS_BWD_LAZY false
#endif
#ifndef RQS_DX3_SPLIT
#define RQS_DX3_SPLIT 2  // independent accumulators of the cooperative kernel's dX3 GEMM (1 = the single chain)
#endif
#ifndef RQS_COOP_LAZY
#define RQS_COOP_LAZY true  // the cooperative kernel has the registers for the lazy knot derivatives
#endif
// INVD: reverse pass of the INVERSE coupling at its output (forward-KL training): `y` holds w and is advanced
// to coupling(w); `ybar` the cotangent of w -> that of the inverse's input; lbar the cotangent of ladj_inv.
template <class G, bool INVD = false>
__device__ __forceinline__ void rqs_bwd_tile(const RqsBwdArgs &a, const float *__restrict__ img, float *__restrict__ sc,
                                             RqsAcc<G> &acc, float *__restrict__ y, float *__restrict__ ybar,
                                             const float *__restrict__ lbar, float lbar_const, long tile, int l31,
                                             int hi, long long *tr) {
  using L = RqsLds<G>;
#ifdef NF_KERNEL_TRACE
#define RQS_STAMP(slot)                                                   \
  do {                                                                    \
    if (tr) { __builtin_amdgcn_sched_barrier(0); tr[slot] = clock64(); }  \
  } while (0)
#else
#define RQS_STAMP(slot) do { (void)tr; } while (0)
#endif
  RQS_STAMP(0);
  const long j = tile * NF_TILE + l31;
  const bool valid = j < a.N;
  const int par_c = 1 - a.par_t;
  const TileIO yio = make_tile_io(y, tile, a.d, l31, hi);
  const TileIO gio = make_tile_io(ybar, tile, a.d, l31, hi);

  unsigned m1[G::H1B], m2[G::H2B];
  f32x16 a2[G::H2B];
  constexpr int ROWS = G::NCH * G::QCH;
  const float *trow = a.tape.base + (tile * a.ncoup + a.k) * (ROWS * 64) + (hi * 32 + l31);
  {
    f32x16 xb[G::MB];
#pragma unroll
    for (int b = 0; b < G::MB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = tile_load(yio, tile_soff(b, r, par_c));
        xb[b][r] = valid ? v : 0.f;
      }
    tile_to_scratch<G::MB>(sc + L::OFF_X, xb, l31, hi);
    f32x16 a1[G::H1B];
    dense_fwd<G::MB, G::H1B>(img + G::W1, img + G::B1, xb, a1, l31, hi);
#pragma unroll
    for (int b = 0; b < G::H1B; ++b) {
      nf_lrelu16(a1[b]);
      m1[b] = nf_sign_mask16(a1[b]);  // bit set <=> slope 0.01
    }
    tile_to_scratch<G::H1B>(sc + L::OFF_A1, a1, l31, hi);
    dense_fwd<G::H1B, G::H2B>(img + G::W2, img + G::B2, a1, a2, l31, hi);
#pragma unroll
    for (int b = 0; b < G::H2B; ++b) {
      nf_lrelu16(a2[b]);
      m2[b] = nf_sign_mask16(a2[b]);
    }
    tile_to_scratch<G::H2B>(sc + L::OFF_A2, a2, l31, hi);
  }
  const float lb = valid ? (lbar ? lbar[j] : lbar_const) : 0.f;
  RQS_STAMP(1);

  f32x16 d2[G::H2B];
#pragma unroll
  for (int b = 0; b < G::H2B; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) d2[b][r] = 0.f;
  float *sd = sc + L::OFF_D;

#pragma unroll
  for (int ch = 0; ch < G::NCH; ++ch) {
    // this chunk's transformed values and cotangents: fetched here (two registers per local dim
    // instead of the whole tile held for the duration), consumed after the chunk's output GEMM
    float yq[G::QCH], gq[G::QCH];
#pragma unroll
    for (int ql = 0; ql < G::QCH; ++ql) {
      const int q = ch * G::QCH + ql;
      yq[ql] = tile_load(yio, tile_soff(q / 16, q % 16, a.par_t));
      gq[ql] = tile_load(gio, tile_soff(q / 16, q % 16, a.par_t));
    }
    f32x16 out[G::OBC];
    dense_fwd<G::H2B, G::OBC, G::S3>(img + G::W3 + ch * G::OBC * 32, img + G::B3 + ch * G::OBC * 32, a2, out, l31, hi);
    RQS_STAMP(2 + 4 * ch);
#pragma unroll
    for (int ql = 0; ql < G::QCH; ++ql) {
      const int q = ch * G::QCH + ql;
      const int p = (q & 3) + 8 * (q >> 2) + 4 * hi;
      const bool ok = valid && p < a.c;
      float raw[G::P], thb[G::P];
      chunk_get<G>(out, ql, raw);
      Knots<G::K> kn;
      build_knots<G::K, RQS_BWD_LAZY>(raw, a.B, kn);
      const float yv = yq[ql];
      const float gv = ok ? gq[ql] : 0.f;
      // the element's bin and in-bin coordinate as the FORWARD found them (spline tape), the bin's knots from the
      // recomputed parameters; then the coupling input (forward chain: x = xi dx + xk; inverse chain: the state holds
      // the spline's input and is advanced through the spline) and the reverse pass of the forward map at that point
      Bin<G::K> bn;
      unsigned code;
      float xi;
      rqs_tape_decode(trow[q * 64], code, xi);
      find_bin<G::K, RQS_BWD_LAZY, true>(kn, 0.f, bn, code);
      float xv;
      {
        const float dx = bn.dx, dy = bn.dy;
        if (INVD) {
          const float sl = nf_fdiv(dy, dx), om = 1.f - xi;
          const float den = sl + (bn.d1 + bn.d0 - 2.f * sl) * xi * om;
          xv = bn.inside ? bn.yk + nf_fdiv(dy * (sl * xi * xi + bn.d0 * xi * om), den) : yv;
        } else {
          xv = bn.inside ? fmaf(xi, dx, bn.xk) : yv;
        }
      }
      const float xbar = rqs_bwd_elem<G::K, INVD>(kn, bn, xi, a.B, gv, ok ? lb : 0.f, thb);  // gv, lb are 0 when !ok
      chunk_put<G>(out, ql, thb);
      tile_store(yio, tile_soff(q / 16, q % 16, a.par_t), xv);    // coupling input x1
      tile_store(gio, tile_soff(q / 16, q % 16, a.par_t), xbar);  // its cotangent
#ifndef NF_RQS_NO_ELEM_FENCE
      // one element at a time: interleaving the two elements of a chunk doubles the live temporaries, and this kernel
      // sits at the register wall (224 dW accumulators in AGPRs + the chunk's 48 raw outputs + knots)
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    // unused slots of the chunk (beyond QCH * P) carry raw outputs of zero-weight rows: clear them
#pragma unroll
    for (int slot = G::QCH * G::P; slot < G::OBC * 16; ++slot) out[slot / 16][slot % 16] = 0.f;
    RQS_STAMP(3 + 4 * ch);
    // dX through the last layer, accumulated over chunks
    dense_bwd_x<G::H2B, G::OBC, G::S3, true>(img + G::W3 + ch * G::OBC * 32, out, d2, l31, hi);
    RQS_STAMP(4 + 4 * ch);
    // dW3^T: two accumulator blocks of delta at a time through the scratch transpose
#pragma unroll
    for (int pc = 0; pc < G::OBC; pc += 2) {
      constexpr int dummy_unused = 0;
      (void)dummy_unused;
      if (pc + 1 < G::OBC) {
        f32x16 two[2] = {out[pc], out[pc + 1]};
        tile_to_scratch<2>(sd, two, l31, hi);
        wave_lds_fence();
        dw_accumulate_at<G::H2B, 2, G::OB3>(sc + L::OFF_A2, sd, acc.w3, acc.b3, ch * G::OBC + pc, l31, hi);
      } else {
        f32x16 one[1] = {out[pc]};
        tile_to_scratch<1>(sd, one, l31, hi);
        wave_lds_fence();
        dw_accumulate_at<G::H2B, 1, G::OB3>(sc + L::OFF_A2, sd, acc.w3, acc.b3, ch * G::OBC + pc, l31, hi);
      }
      wave_lds_fence();
    }
    RQS_STAMP(5 + 4 * ch);
  }
  // ---- layer 2
#pragma unroll
  for (int b = 0; b < G::H2B; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) d2[b][r] *= nf_mask_slope(m2[b], r);
  tile_to_scratch<G::H2B>(sd, d2, l31, hi);
  wave_lds_fence();
  dw_accumulate<G::H1B, G::H2B>(sc + L::OFF_A1, sd, acc.w2, acc.b2, l31, hi);
  f32x16 d1[G::H1B];
  dense_bwd_x<G::H1B, G::H2B>(img + G::W2, d2, d1, l31, hi);
#pragma unroll
  for (int b = 0; b < G::H1B; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) d1[b][r] *= nf_mask_slope(m1[b], r);
  wave_lds_fence();
  // ---- layer 1
  tile_to_scratch<G::H1B>(sd, d1, l31, hi);
  wave_lds_fence();
  dw_accumulate<G::MB, G::H1B>(sc + L::OFF_X, sd, acc.w1, acc.b1, l31, hi);
  f32x16 g2[G::MB], gold[G::MB];
#pragma unroll
  for (int b = 0; b < G::MB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) gold[b][r] = tile_load(gio, tile_soff(b, r, par_c));
  dense_bwd_x<G::MB, G::H1B>(img + G::W1, d1, g2, l31, hi);
  wave_lds_fence();
#pragma unroll
  for (int b = 0; b < G::MB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) tile_store(gio, tile_soff(b, r, par_c), gold[b][r] + g2[b][r]);
  RQS_STAMP(2 + 4 * G::NCH);
#undef RQS_STAMP
}

// reverse pass of ONE coupling by this workgroup: stage the image, walk this workgroup's tiles, fold the four waves'
// accumulators (deterministic, wave-ordered) and write the workgroup's slab
template <class G, bool INVD>
__device__ __forceinline__ void rqs_bwd_coupling(const RqsBwdArgs &a, float *__restrict__ y, float *__restrict__ ybar,
                                                 const float *__restrict__ lbar, float lbar_const,
                                                 float *__restrict__ slab, long slab_stride, float *lds) {
  float *img = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  float *sc = lds + G::SIZE + wave * RqsLds<G>::SCRATCH;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  stage_packed<G::SIZE, 256>(img, a.img, tid);
  __syncthreads();
  RqsAcc<G> acc;
  rqs_zero(acc.w1, acc.b1);
  rqs_zero(acc.w2, acc.b2);
  rqs_zero(acc.w3, acc.b3);
#pragma unroll 1
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4)
    rqs_bwd_tile<G, INVD>(a, img, sc, acc, y, ybar, lbar, lbar_const, tile, l31, hi,
                    (a.trace && blockIdx.x == 0 && tid == 0 && tile == (long)wave) ? a.trace : nullptr);
  __syncthreads();  // the weight image is dead: it becomes the (deterministic, wave-ordered) fold target
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      rqs_fold<G::MB, G::H1B, G::S1>(img + G::W1, img + G::B1, acc.w1, acc.b1, w == 0, l31, hi);
      rqs_fold<G::H1B, G::H2B, G::S2>(img + G::W2, img + G::B2, acc.w2, acc.b2, w == 0, l31, hi);
      rqs_fold<G::H2B, G::OB3, G::S3>(img + G::W3, img + G::B3, acc.w3, acc.b3, w == 0, l31, hi);
    }
    __syncthreads();
  }
  {
    const float4 *c0 = reinterpret_cast<const float4 *>(img);
    float4 *dst = reinterpret_cast<float4 *>(slab + (long)blockIdx.x * slab_stride);
    for (int i = tid; i < G::SIZE / 4; i += 256) dst[i] = c0[i];
  }
  __syncthreads();  // the image region is restaged by the next coupling
}

template <class G, bool INVD>
__global__ __launch_bounds__(256, 1) void k_rqs_bwd(RqsBwdArgs a, float *__restrict__ y, float *__restrict__ ybar,
                                                    const float *__restrict__ lbar, float lbar_const,
                                                    float *__restrict__ slab, long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  rqs_bwd_coupling<G, INVD>(a, y, ybar, lbar, lbar_const, slab, slab_stride, lds);
}

// The whole chain's reverse pass in ONE launch (the structure of k_affine_bwd_all, nf_coupling.hip): a wave's tiles
// never change hands and coupling k + 1 only reads what the same wave wrote for coupling k (y <- x, ybar <- xbar),
// so each workgroup walks the couplings on its own -- seven launch gaps and ramp-up / tail phases less per step.
// ---------------------------------------------------------------------------------------
// reverse pass of one coupling, COOPERATIVE form (geometries with NCH == 4 chunks)
// ---------------------------------------------------------------------------------------
// k_rqs_bwd keeps the dW^T accumulators of the whole output layer in every wave: H2B x OB3 blocks = 192 (K = 8) or
// 256 (K = 10, d = 32) registers, on top of the chunk's raw outputs and the spline's working set -- the kernel sits at
// the register wall (256 VGPR + 256 AGPR + scratch spills; 1 400 of its 9 800 instructions per tile are AGPR <-> VGPR
// moves), and the K = 10 / d = 32 shape (the reference's default nsf(q0) on d > 16, neuralspline.jl:232-234) does not
// fit at all.  Here the four waves of a workgroup split the OUTPUT COLUMNS instead of the tiles:
//   home phase   wave w recomputes layers 1-2 of its own tile (a1, a2 and their sign masks) and leaves x2, a1, a2 as
//                [feature][sample] tiles in its LDS scratch;
//   chunk phase  wave w owns chunk w of the output layer for ALL FOUR tiles of the group: reads tile t's a2 from wave
//                t's scratch, evaluates its chunk of the output layer, runs the spline and its reverse pass for the
//                chunk's dims, its part of dX3 (a partial d2, handed to the home wave through an LDS slot) and its
//                columns of dW3^T -- OBC accumulator blocks instead of OB3;
//   home phase   wave w sums the four partial d2 of its tile (fixed order: deterministic) and finishes layers 2 and 1.
// Same MFMA count as k_rqs_bwd, 80-96 accumulator registers per wave instead of 224-288, 9 workgroup barriers per 4 tiles.
template <class G>
struct RqsCoopLds {
  static_assert(G::NCH == 4, "one chunk per wave");
  static constexpr int OFF_X = 0;
  static constexpr int OFF_A1 = OFF_X + G::MB * 32 * NF_TS;
  static constexpr int OFF_A2 = OFF_A1 + G::H1B * 32 * NF_TS;
  static constexpr int OFF_D = OFF_A2 + G::H2B * 32 * NF_TS;
  static constexpr int DBLK = (G::H1B > G::H2B ? G::H1B : G::H2B);  // delta tile: one block for dW3, a whole hidden layer for dW2 / dW1
  static constexpr int SCRATCH = OFF_D + DBLK * 32 * NF_TS;
  static constexpr int SLOT = G::H2B * 16 * 64;  // one partial d2 in register-dump order
  // two sets of slots (tile t uses set t & 1) let the chunk phase run with ONE workgroup barrier per tile: the home wave
  // sums tile t's partials while the others already work on tile t + 1.  160 KB of LDS hold them for K = 8 only.
  static constexpr bool TWO_SETS = (size_t)(G::SIZE + 4 * SCRATCH + 8 * SLOT) * sizeof(float) <= 160 * 1024;
  static constexpr int NSETS = TWO_SETS ? 2 : 1;
  static constexpr size_t BYTES = (size_t)(G::SIZE + 4 * SCRATCH + 4 * NSETS * SLOT) * sizeof(float);
};


// Orders a wave's own LDS accesses for the COMPILER only.  The hardware needs nothing: a wave's LDS instructions execute in
// issue order, so a ds_read issued after a ds_write of the same wave sees it (and a ds_write after a ds_read cannot
// overtake it).  wave_lds_fence() (nf_mfma.h) additionally makes the wave WAIT for its outstanding LDS traffic
// (s_waitcnt lgkmcnt(0)), which the transpose round trips below do not need: the consumer of the read has its own wait.
__device__ __forceinline__ void wave_lds_order() {
#ifdef RQS_COOP_R3
  wave_lds_fence();
#else
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
#endif
}

// dW3^T of one (tile, chunk): acc[ib][pc] += a2^T x delta_pc^T for the chunk's OBC blocks of delta (`out` after the spline's
// reverse pass), bsum[pc] += row sums of delta.  Round 4 form: the a2^T operand (16 k-steps = the tile's 32 samples) is
// fetched ONCE and serves all OBC blocks (round 3 fetched it per block), a block's whole delta^T operand is fetched in one
// go, so that the NEXT block's transpose write is already under way while this block's MFMAs run (the wave's transpose
// tile holds one block), and the bias sums are pairwise trees of two-wide adds instead of a 16-long chain.
template <class G>
__device__ __forceinline__ void rqs_dw3_chunk(const float *__restrict__ sa2, float *__restrict__ sd, const f32x16 (&out)[G::OBC],
                                              f32x16 (&acc)[G::H2B][G::OBC], float (&bsum)[G::OBC], int l31, int hi) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const float *pa = sa2 + l31 * NF_TS + hi;
  const float *pd = sd + l31 * NF_TS + hi;
  float aT[G::H2B][16];
#pragma unroll
  for (int ib = 0; ib < G::H2B; ++ib)
#pragma unroll
    for (int t = 0; t < 16; ++t) aT[ib][t] = pa[ib * 32 * NF_TS + 2 * t];
  {
    f32x16 one[1] = {out[0]};
    tile_to_scratch<1>(sd, one, l31, hi);
  }
#pragma unroll
  for (int pc = 0; pc < G::OBC; ++pc) {
    wave_lds_order();
    float dT[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) dT[t] = pd[2 * t];
    wave_lds_order();
    if (pc + 1 < G::OBC) {
      f32x16 one[1] = {out[pc + 1]};
      tile_to_scratch<1>(sd, one, l31, hi);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x2 s2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) s2[t] = f32x2{dT[2 * t], dT[2 * t + 1]};
#pragma unroll
    for (int w = 4; w >= 1; w >>= 1)
#pragma unroll
      for (int t = 0; t < w; ++t) s2[t] += s2[t + w];
    bsum[pc] += s2[0][0] + s2[0][1];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int ib = 0; ib < G::H2B; ++ib)
        acc[ib][pc] = __builtin_amdgcn_mfma_f32_32x32x2f32(aT[ib][t], dT[t], acc[ib][pc], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_lds_order();
}

template <class G>
struct RqsCoopAcc {
  f32x16 w1[G::MB][G::H1B];
  f32x16 w2[G::H1B][G::H2B];
  f32x16 w3[G::H2B][G::OBC];  // this wave's chunk of the output layer
  float b1[G::H1B], b2[G::H2B], b3[G::OBC];
};

template <class G, bool INVD>
__device__ __forceinline__ void rqs_bwd_coop_coupling(const RqsBwdArgs &a, float *__restrict__ y, float *__restrict__ ybar,
                                                      const float *__restrict__ lbar, float lbar_const,
                                                      float *__restrict__ slab, long slab_stride, float *lds) {
  using L = RqsCoopLds<G>;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  // LDS map: the waves' transpose tiles FIRST, then the partial-d2 slots, the weight image last.  A ds instruction carries a
  // 16-bit byte offset; with the 60 KB image in front (round 3) the rows of a transpose tile straddled the 64 KB mark and
  // hipcc materialised a separate address register for most of them (131 v_add_u32 in the kernel, 59 per tile of the chunk
  // phase).  The image is addressed through one per-lane pointer plus row offsets below 50 KB wherever it sits.
#ifdef RQS_COOP_R3
  float *img = lds;
  float *sc_all = lds + G::SIZE;
#else
  float *sc_all = lds;
  float *img = lds + 4 * L::SCRATCH + 4 * L::NSETS * L::SLOT;
#endif
  float *sc = sc_all + wave * L::SCRATCH;          // this wave's tiles
  float *slots = sc_all + 4 * L::SCRATCH;          // [NSETS][4][SLOT]
  float *sd = sc + L::OFF_D;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const long ngroups = (ntiles + 3) / 4;
  const int par_c = 1 - a.par_t;
  const int ch = wave;                              // the chunk this wave owns
  stage_packed<G::SIZE, 256>(img, a.img, tid);
  __syncthreads();
  RqsCoopAcc<G> acc;
  rqs_zero(acc.w1, acc.b1);
  rqs_zero(acc.w2, acc.b2);
  rqs_zero(acc.w3, acc.b3);
  // The conditioner half of a group's home tile is requested ONE GROUP AHEAD (right after barrier B0, when the registers
  // are free again): with one wave per SIMD nothing else hides the HBM latency of these loads -- the home phase took
  // 9.1 k cycles for 2 k cycles of MFMA work before (tools/trace_rqs.py).
  // (round 4: the loads stay RAW here and padding samples are zeroed where the tile is consumed, one group later.  With the
  // select next to the loads hipcc waited for all sixteen of them -- a full HBM round trip -- at the head of the closing
  // home phase, the very latency the prefetch is there to hide: ISA of round 3, `s_waitcnt vmcnt(15..0)` + v_cndmask pairs.)
  f32x16 xb[G::MB];
  auto load_home = [&](long g) {
    const long tile_ = g * 4 + wave;
    const bool live_ = tile_ < ntiles;
    const long tl_ = live_ ? tile_ : 0;
    const TileIO yio_ = make_tile_io(y, tl_, a.d, l31, hi);
#pragma unroll
    for (int b = 0; b < G::MB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = tile_load(yio_, tile_soff(b, r, par_c));
#ifdef RQS_COOP_R3
        xb[b][r] = (live_ && tl_ * NF_TILE + l31 < a.N) ? v : 0.f;
#else
        xb[b][r] = v;
#endif
      }
  };
#ifndef RQS_NO_PREFETCH
  if ((long)blockIdx.x < ngroups) load_home(blockIdx.x);
#endif
#ifdef NF_KERNEL_TRACE
  long long *tr = (a.trace && blockIdx.x == 0 && tid == 0) ? a.trace : nullptr;
  if (tr) { tr[100] = clock64(); tr[101] = wall_clock64(); }  // shader clock against the constant 100 MHz counter: the clock the kernel ran at
#define COOP_STAMP(slot) do { if (tr && grp == (long)blockIdx.x) { __builtin_amdgcn_sched_barrier(0); tr[slot] = clock64(); } } while (0)
#else
#define COOP_STAMP(slot) do { } while (0)
#endif
#pragma unroll 1
  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    COOP_STAMP(0);
    // ---------------- home phase: layers 1-2 of this wave's own tile ----------------
    const long tile = grp * 4 + wave;
    const bool live = tile < ntiles;               // wave-uniform; every barrier below is reached by all waves
    const long tl = live ? tile : 0;
    const long j = tl * NF_TILE + l31;
    const bool valid = live && j < a.N;
    const TileIO yio = make_tile_io(y, tl, a.d, l31, hi);
    const TileIO gio = make_tile_io(ybar, tl, a.d, l31, hi);
    unsigned m1[G::H1B], m2[G::H2B];
    (void)yio;
#ifdef RQS_NO_PREFETCH
    load_home(grp);
#endif
    if (live) {
#ifndef RQS_COOP_R3
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) xb[b][r] = valid ? xb[b][r] : 0.f;
#endif
      tile_to_scratch<G::MB>(sc + L::OFF_X, xb, l31, hi);
      f32x16 a1[G::H1B];
#ifdef RQS_COOP_R3
      dense_fwd<G::MB, G::H1B>(img + G::W1, img + G::B1, xb, a1, l31, hi);
#else
      dense_fwd_dyn<G::MB, G::H1B>(img + G::W1, img + G::B1, xb, a1, l31, hi, (a.m + 7) >> 3);
#endif
#pragma unroll
      for (int b = 0; b < G::H1B; ++b) {
        nf_lrelu16(a1[b]);
        m1[b] = nf_sign_mask16(a1[b]);
      }
      tile_to_scratch<G::H1B>(sc + L::OFF_A1, a1, l31, hi);
      f32x16 a2[G::H2B];
      dense_fwd<G::H1B, G::H2B>(img + G::W2, img + G::B2, a1, a2, l31, hi);
#pragma unroll
      for (int b = 0; b < G::H2B; ++b) {
        nf_lrelu16(a2[b]);
        m2[b] = nf_sign_mask16(a2[b]);
      }
      tile_to_scratch<G::H2B>(sc + L::OFF_A2, a2, l31, hi);
    }
    COOP_STAMP(1);
    __syncthreads();  // B0: every home tile's a2 is in LDS
    COOP_STAMP(2);
    // (the next group's home tile is requested at the start of the closing home phase: vector loads return in order, so
    // requesting it here put 16 loads in front of the ones tile 0 of the chunk phase waits for -- 4.4 k cycles, measured)

    // ---------------- chunk phase: this wave's chunk of the output layer, for each tile of the group ----------------
    f32x16 d2[G::H2B];  // the home tile's summed cotangent of a2 (filled when t == wave)
#pragma unroll
    for (int b = 0; b < G::H2B; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) d2[b][r] = 0.f;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      const long tt = grp * 4 + t;
      const bool tlive = tt < ntiles;  // uniform over the workgroup
      if (tlive) {
        const float *sct = sc_all + t * L::SCRATCH;  // wave t's tiles
        const long jt = tt * NF_TILE + l31;
        const bool tvalid = jt < a.N;
        const TileIO yt = make_tile_io(y, tt, a.d, l31, hi);
        const TileIO gt = make_tile_io(ybar, tt, a.d, l31, hi);
        const float lb = tvalid ? (lbar ? lbar[jt] : lbar_const) : 0.f;
        float yq[G::QCH], gq[G::QCH], xiq[G::QCH];
        constexpr int ROWS = G::NCH * G::QCH;
        const float *trow = a.tape.base + (tt * a.ncoup + a.k) * (ROWS * 64) + lane;
#pragma unroll
        for (int ql = 0; ql < G::QCH; ++ql) {
          const int q = ch * G::QCH + ql;
          yq[ql] = tile_load(yt, tile_soff(q / 16, q % 16, a.par_t));
          gq[ql] = tile_load(gt, tile_soff(q / 16, q % 16, a.par_t));
          xiq[ql] = trow[q * 64];
        }

        f32x16 out[G::OBC];
        {
          f32x16 a2c[G::H2B];  // tile t's a2 back in the accumulator layout (B operand of the output layer)
#pragma unroll
          for (int b = 0; b < G::H2B; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) a2c[b][r] = sct[L::OFF_A2 + (b * 32 + nf_row(r, hi)) * NF_TS + l31];
          dense_fwd<G::H2B, G::OBC, G::S3>(img + G::W3 + ch * G::OBC * 32, img + G::B3 + ch * G::OBC * 32, a2c, out, l31, hi);
        }
        COOP_STAMP(3 + 6 * t);
#pragma unroll
        for (int ql = 0; ql < G::QCH; ++ql) {
          const int q = ch * G::QCH + ql;
          const int p = (q & 3) + 8 * (q >> 2) + 4 * hi;
          const bool ok = tvalid && p < a.c;
          float raw[G::P], thb[G::P];
          chunk_get<G>(out, ql, raw);
          Knots<G::K> kn;
          build_knots<G::K, RQS_COOP_LAZY>(raw, a.B, kn);
          const float yv = yq[ql];
          const float gv = ok ? gq[ql] : 0.f;
          // bin and in-bin coordinate from the forward's spline tape (see RqsTape), knots from the recomputed parameters
          Bin<G::K> bn;
          unsigned code;
          float xi;
          rqs_tape_decode(xiq[ql], code, xi);
          find_bin<G::K, RQS_COOP_LAZY, true>(kn, 0.f, bn, code);
          float xv;
          {
            const float dx = bn.dx, dy = bn.dy;
            if (INVD) {
              const float sl = nf_fdiv(dy, dx), om = 1.f - xi;
              const float den = sl + (bn.d1 + bn.d0 - 2.f * sl) * xi * om;
              xv = bn.inside ? bn.yk + nf_fdiv(dy * (sl * xi * xi + bn.d0 * xi * om), den) : yv;
            } else {
              xv = bn.inside ? fmaf(xi, dx, bn.xk) : yv;
            }
          }
          const float xbar = rqs_bwd_elem<G::K, INVD>(kn, bn, xi, a.B, gv, ok ? lb : 0.f, thb);
          chunk_put<G>(out, ql, thb);
          tile_store(yt, tile_soff(q / 16, q % 16, a.par_t), xv);
          tile_store(gt, tile_soff(q / 16, q % 16, a.par_t), xbar);
        }
#pragma unroll
        for (int slot = G::QCH * G::P; slot < G::OBC * 16; ++slot) out[slot / 16][slot % 16] = 0.f;
        COOP_STAMP(4 + 6 * t);
        // this chunk's share of dX3 -> the home wave's slot
        {
          f32x16 d2p[G::H2B];
#ifdef RQS_OLD_DX3
          dense_bwd_x<G::H2B, G::OBC, G::S3>(img + G::W3 + ch * G::OBC * 32, out, d2p, l31, hi);
#else
          dense_bwd_x_split<G::H2B, G::OBC, G::S3, RQS_DX3_SPLIT>(img + G::W3 + ch * G::OBC * 32, out, d2p, l31, hi);
#endif
          float *mys = slots + ((L::NSETS == 2 ? (t & 1) : 0) * 4 + wave) * L::SLOT;
#pragma unroll
          for (int b = 0; b < G::H2B; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mys[(b * 16 + r) * 64 + lane] = d2p[b][r];
        }
        COOP_STAMP(5 + 6 * t);
        // this chunk's columns of dW3^T: one block of delta at a time through the wave's own transpose tile
#if defined(RQS_COOP_R3) || defined(RQS_COOP_OLD_DW3)
#pragma unroll
        for (int pc = 0; pc < G::OBC; ++pc) {
          f32x16 one[1] = {out[pc]};
          tile_to_scratch<1>(sd, one, l31, hi);
          wave_lds_fence();
          dw_accumulate_at<G::H2B, 1, G::OBC>(sct + L::OFF_A2, sd, acc.w3, acc.b3, pc, l31, hi);
          wave_lds_fence();
        }
#else
        rqs_dw3_chunk<G>(sct + L::OFF_A2, sd, out, acc.w3, acc.b3, l31, hi);
#endif
        COOP_STAMP(6 + 6 * t);
      }
      __syncthreads();  // B1: the four partial d2 of tile t are in the slots
      if (tlive && t == wave) {
        const float *set = slots + (L::NSETS == 2 ? (t & 1) : 0) * 4 * L::SLOT;
#pragma unroll
        for (int w = 0; w < 4; ++w)  // fixed order: deterministic
#pragma unroll
          for (int b = 0; b < G::H2B; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) d2[b][r] += set[w * L::SLOT + (b * 16 + r) * 64 + lane];
      }
      COOP_STAMP(7 + 6 * t);
      // B2 (slots free for tile t + 1) only with ONE set of slots: with two, tile t + 1 writes the other set, and
      // tile t + 2 writes this one after B1 of tile t + 1, which the home wave reaches only after the sum above
      if (L::NSETS == 1) __syncthreads();
      COOP_STAMP(8 + 6 * t);
    }

    // ---------------- home phase: layers 2 and 1 of this wave's own tile ----------------
#ifndef RQS_NO_PREFETCH
    if (grp + gridDim.x < ngroups) load_home(grp + gridDim.x);  // in flight behind the 64 MFMAs of this phase
#endif
    if (live) {
#pragma unroll
      for (int b = 0; b < G::H2B; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) d2[b][r] *= nf_mask_slope(m2[b], r);
      tile_to_scratch<G::H2B>(sd, d2, l31, hi);
      wave_lds_order();
      dw_accumulate<G::H1B, G::H2B>(sc + L::OFF_A1, sd, acc.w2, acc.b2, l31, hi);
      f32x16 d1[G::H1B];
      dense_bwd_x<G::H1B, G::H2B>(img + G::W2, d2, d1, l31, hi);
#if !defined(RQS_COOP_R3) && !defined(RQS_COOP_GOLD_LATE)
      // the conditioner half of the cotangent tile is the accumulator the dX1 GEMM starts from.  Requested here, two GEMMs
      // ahead of its use (d2 is dead now; held from the start of the phase it cost 250-320 bytes of scratch spills)
      f32x16 g2[G::MB];
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) g2[b][r] = tile_load(gio, tile_soff(b, r, par_c));
#endif
#pragma unroll
      for (int b = 0; b < G::H1B; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) d1[b][r] *= nf_mask_slope(m1[b], r);
      wave_lds_order();
      tile_to_scratch<G::H1B>(sd, d1, l31, hi);
      wave_lds_order();
      dw_accumulate<G::MB, G::H1B>(sc + L::OFF_X, sd, acc.w1, acc.b1, l31, hi);
#if defined(RQS_COOP_R3) || defined(RQS_COOP_GOLD_LATE)
      f32x16 g2[G::MB], gold[G::MB];
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) gold[b][r] = tile_load(gio, tile_soff(b, r, par_c));
      dense_bwd_x<G::MB, G::H1B>(img + G::W1, d1, g2, l31, hi);
      wave_lds_order();
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile_store(gio, tile_soff(b, r, par_c), gold[b][r] + g2[b][r]);
#else
      dense_bwd_x<G::MB, G::H1B, G::S1, true>(img + G::W1, d1, g2, l31, hi);
      wave_lds_order();
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile_store(gio, tile_soff(b, r, par_c), g2[b][r]);
#endif
    }
    COOP_STAMP(27);
    // no barrier here: the next group's home phase writes only this wave's own tiles, which the other waves stopped
    // reading at the last B2, and B0 orders those writes before anybody reads them
  }
#ifdef NF_KERNEL_TRACE
  if (tr) { tr[102] = clock64(); tr[103] = wall_clock64(); }
#endif
  __syncthreads();  // the weight image is dead: it becomes the fold target
  // layers 1 and 2: four partial sums (one per wave), added in wave order; output layer: each wave owns its columns
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      rqs_fold<G::MB, G::H1B, G::S1>(img + G::W1, img + G::B1, acc.w1, acc.b1, w == 0, l31, hi);
      rqs_fold<G::H1B, G::H2B, G::S2>(img + G::W2, img + G::B2, acc.w2, acc.b2, w == 0, l31, hi);
    }
    __syncthreads();
  }
  rqs_fold<G::H2B, G::OBC, G::S3>(img + G::W3 + ch * G::OBC * 32, img + G::B3 + ch * G::OBC * 32, acc.w3, acc.b3, true, l31, hi);
  __syncthreads();
  {
    const float4 *c0 = reinterpret_cast<const float4 *>(img);
    float4 *dst = reinterpret_cast<float4 *>(slab + (long)blockIdx.x * slab_stride);
    for (int i = tid; i < G::SIZE / 4; i += 256) dst[i] = c0[i];
  }
  __syncthreads();
}

// Register budget.  With one wave per SIMD (launch bound 1) a wave may use 512 registers and hipcc splits them into 256
// VGPRs + 256 AGPRs, selecting the AGPR form for EVERY MFMA of the kernel: each forward / dX result then has to be read
// back with v_accvgpr_read before the VALU can touch it (447 of this kernel's 3 400 instructions were AGPR <-> VGPR
// moves).  The cooperative kernel needs only 80 accumulator registers, so for K = 8 the whole working set fits 256
// VGPRs (234 used, no scratch): declaring a budget of two waves per SIMD makes hipcc select the VGPR form of the MFMAs and
// drop the AGPR file altogether -- 3 062 instructions instead of 3 399.  (LDS still admits one workgroup per CU; the
// second argument is a register budget here, not an occupancy.)  K = 10 would spill 88 bytes per lane and keeps the split.
#ifndef RQS_COOP_WAVES_PER_SIMD
#define RQS_COOP_WAVES_PER_SIMD(G) (G::K <= 8 ? 2 : 1)
#endif
template <class G, bool INVD>
__global__ __launch_bounds__(256, RQS_COOP_WAVES_PER_SIMD(G)) void k_rqs_bwd_coop(RqsBwdArgs a, float *__restrict__ y, float *__restrict__ ybar,
                                                         const float *__restrict__ lbar, float lbar_const,
                                                         float *__restrict__ slab, long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  rqs_bwd_coop_coupling<G, INVD>(a, y, ybar, lbar, lbar_const, slab, slab_stride, lds);
}

// ---------------------------------------------------------------------------------------
// cooperative reverse pass, output layer on the bf16 matrix cores (round 5)
// ---------------------------------------------------------------------------------------
// The three GEMMs of the chunk phase -- raw-parameter recompute W3 a2 (32 -> OBC x 32 columns), dX3 = W3^T delta and
// dW3^T = a2 delta^T -- are 576 of the group's 664 fp32 MFMAs (43 k of its 68 k clocks, DESIGN section 4 "cfg 3").  Here they
// run as six-term bf16 products (nf_mfma.h "B6": 108 instructions of 32 clocks instead of 144 of 64 per tile and chunk).
// What made that not fit in round 4 was the weight operand: W3 as bf16 triples in two orientations is 2 x 74 KB and the LDS
// was full.  But a wave of this kernel owns ONE chunk of the output layer for the whole launch -- its weights never change:
// they are loaded ONCE per launch from the triple images in global memory (RqsB6Geo, written by k_rqs_b6_from_images
// behind the fp32 images) into REGISTERS, 18 + 18 sixteen-byte vectors = 144 registers per lane (one wave per SIMD owns
// 512), and the fp32 W3 image (50 KB) leaves the LDS altogether.  The operands that change per tile:
//   a2 (B operand of the recompute, lane <-> sample)    split once by the tile's HOME wave, six 16-byte stores, read by
//                                                       the four chunk waves with six ds_read_b128;
//   a2^T (A operand of dW3, lane <-> feature)           the same triples, stored transposed by the home wave (D6 layout of
//                                                       nf_mfma.h: one ds_write_b16 per value and component);
//   delta (B operand of dX3, lane <-> sample)           the chunk's spline cotangents, split k-group by k-group;
//   delta^T (B operand of dW3, lane <-> column)         those very triples through a two-block D6 ring in LDS -- the
//                                                       dX3 MFMAs of block pc + 1 run beside the dW3 MFMAs of block pc.
// x2 and a1 of the home tile wait in registers for the closing home phase (their transpose tiles alias the delta ring),
// which together with the missing W3 image leaves room for both sets of partial-d2 slots at every geometry.
template <class G>
struct RqsCoop6Lds {
  static_assert(G::NCH == 4, "one chunk per wave");
  static constexpr int DBLK = (G::H1B > G::H2B ? G::H1B : G::H2B);
  // per wave, bytes
  static constexpr int A2C = 0;                                   // a2 triples, B-operand order: [k-group][component][lane] x 16 B
  static constexpr int A2C_BYTES = 2 * G::H2B * 3 * 64 * 16;
  // (round 6: both through gfx950's transposing LDS read -- the writers keep their natural [sample][16 features] rows, two 8-byte
  // stores per (k-group, component) instead of eight 2-byte ones; nf_mfma.h "TR layout")
  static constexpr int A2T = A2C + A2C_BYTES;                     // a2 triples for the A operand of dW3: [component][2 H2B tiles] x TR_TILE
  static constexpr int A2T_BYTES = 3 * 2 * G::H2B * TR_TILE;
  static constexpr int DT = A2T + A2T_BYTES;                      // ring of two delta blocks ([component][2 tiles] x TR_TILE each); closing home phase: fp32 tiles
  static constexpr int RING_BLK = 3 * 2 * TR_TILE;
  static constexpr int RING_BYTES = 2 * RING_BLK;
  static constexpr int TILE_X = 0, TILE_A1 = G::MB * 32 * NF_TS, TILE_D = TILE_A1 + G::H1B * 32 * NF_TS;  // floats from DT
  static constexpr int TILES_BYTES = (TILE_D + DBLK * 32 * NF_TS) * 4;
  static constexpr int WAVE_BYTES = ((DT + (RING_BYTES > TILES_BYTES ? RING_BYTES : TILES_BYTES)) + 15) / 16 * 16;
  static constexpr int SLOT = G::H2B * 16 * 64;                   // floats: one partial d2 in register-dump order
  static constexpr int IMG = G::W3 + G::NCOLS;                    // floats: W1, b1, W2, b2 of the fp32 image, then b3
  static constexpr size_t WORK_BYTES = (size_t)4 * WAVE_BYTES + (size_t)8 * SLOT * 4 + (size_t)IMG * 4;
  static constexpr size_t FOLD_BYTES = (size_t)G::SIZE * 4;       // the slab image the accumulators are folded into at the end
  static constexpr size_t BYTES = WORK_BYTES > FOLD_BYTES ? WORK_BYTES : FOLD_BYTES;
  // (K = 10 at d = 32 -- 4 output blocks per chunk, 192 weight registers -- spills 390 bytes per lane: it keeps the fp32 kernel)
  static constexpr bool FITS = BYTES <= 160 * 1024 && G::OBC <= 3;
};

// this wave's weights of the output layer, in registers for the whole launch
template <class G>
struct RqsW6 {
  nf_u32x4 f[2 * G::H2B][G::OBC][3];  // recompute: [k-group over a2's features][output block of the chunk][h, m, l]
  nf_u32x4 t[2 * G::OBC][G::H2B][3];  // dX3: [k-group over the chunk's columns][a2 block][h, m, l]
};

template <class G, bool INVD>
__device__ __forceinline__ void rqs_bwd_coop6_coupling(const RqsBwdArgs &a, float *__restrict__ y, float *__restrict__ ybar,
                                                       const float *__restrict__ lbar, float lbar_const,
                                                       float *__restrict__ slab, long slab_stride, char *lds) {
  using L = RqsCoop6Lds<G>;
  using B = RqsB6Geo<G>;
  const int tid = threadIdx.x;
#ifdef NF_KERNEL_TRACE  // tools/trace_rqs6.py: workgroup 0 / wave 0: [0] start, [1] prologue done, [2 + 12 g ..] group g: home forward done (B0), then per
                        // tile t the chunk phase's end (B1) and the d2 sum's end, closing home phase done; [60] groups done, [61] folded, [62] slab written
  long long *tr6 = (a.trace && blockIdx.x == 0 && tid == 0) ? a.trace : nullptr;
#define C6_STAMP(slot) do { if (tr6) { __builtin_amdgcn_sched_barrier(0); tr6[slot] = clock64(); } } while (0)
#else
#define C6_STAMP(slot) do { } while (0)
#endif
  C6_STAMP(0);
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  char *wv = lds + wave * L::WAVE_BYTES;                                   // this wave's region
  float *slots = reinterpret_cast<float *>(lds + 4 * L::WAVE_BYTES);       // [2][4][SLOT]
  float *img = slots + 8 * L::SLOT;                                        // W1, b1, W2, b2 at their image offsets; b3 at G::W3
  const float *b3 = img + G::W3;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const long ngroups = (ntiles + 3) / 4;
  const int par_c = 1 - a.par_t;
  const int ch = wave;
  {  // stage layers 1-2 and the output layer's bias (16-byte vectors; G::W3 and G::B3 are multiples of 4)
    const float4 *s0 = reinterpret_cast<const float4 *>(a.img);
    float4 *d0 = reinterpret_cast<float4 *>(img);
    for (int i = tid; i < G::W3 / 4; i += 256) d0[i] = s0[i];
    const float4 *s1 = reinterpret_cast<const float4 *>(a.img + G::B3);
    float4 *d1 = reinterpret_cast<float4 *>(img + G::W3);
    for (int i = tid; i < G::NCOLS / 4; i += 256) d1[i] = s1[i];
  }
  RqsW6<G> W;
  {
    const nf_u32x4 *wf = a.img6 + (size_t)ch * B::F_CH + hi * B::FROWS + l31;
#pragma unroll
    for (int kg = 0; kg < B::FKG; ++kg)
#pragma unroll
      for (int ob = 0; ob < G::OBC; ++ob)
#pragma unroll
        for (int c = 0; c < 3; ++c) W.f[kg][ob][c] = wf[((kg * 3 + c) * 2) * B::FROWS + ob * 32];
    const nf_u32x4 *wt = a.img6 + B::OFF_T + (size_t)ch * B::T_CH + hi * B::TROWS + l31;
#pragma unroll
    for (int kg = 0; kg < B::TKG; ++kg)
#pragma unroll
      for (int ib = 0; ib < G::H2B; ++ib)
#pragma unroll
        for (int c = 0; c < 3; ++c) W.t[kg][ib][c] = wt[((kg * 3 + c) * 2) * B::TROWS + ib * 32];
  }
  __syncthreads();
  C6_STAMP(1);
  int gcount6 = 0;
  RqsCoopAcc<G> acc;
  rqs_zero(acc.w1, acc.b1);
  rqs_zero(acc.w2, acc.b2);
  rqs_zero(acc.w3, acc.b3);
  f32x16 xb[G::MB];
  auto load_home = [&](long g) {
    const long tile_ = g * 4 + wave;
    const long tl_ = tile_ < ntiles ? tile_ : 0;
    const TileIO yio_ = make_tile_io(y, tl_, a.d, l31, hi);
#pragma unroll
    for (int b = 0; b < G::MB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) xb[b][r] = tile_load(yio_, tile_soff(b, r, par_c));
  };
  if ((long)blockIdx.x < ngroups) load_home(blockIdx.x);
  float *tiles = reinterpret_cast<float *>(wv + L::DT);  // closing home phase: x2^T, a1^T, delta^T (fp32, row stride 33)
#pragma unroll 1
  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    // ---------------- home phase: layers 1-2 of this wave's own tile ----------------
    const long tile = grp * 4 + wave;
    const bool live = tile < ntiles;
    const long tl = live ? tile : 0;
    const long j = tl * NF_TILE + l31;
    const bool valid = live && j < a.N;
    const TileIO gio = make_tile_io(ybar, tl, a.d, l31, hi);
    unsigned m1[G::H1B], m2[G::H2B];
    f32x16 a1[G::H1B];
    if (live) {
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) xb[b][r] = valid ? xb[b][r] : 0.f;
      dense_fwd_dyn<G::MB, G::H1B>(img + G::W1, img + G::B1, xb, a1, l31, hi, (a.m + 7) >> 3);
#pragma unroll
      for (int b = 0; b < G::H1B; ++b) {
        nf_lrelu16(a1[b]);
        m1[b] = nf_sign_mask16(a1[b]);
      }
      f32x16 a2[G::H2B];
      dense_fwd<G::H1B, G::H2B>(img + G::W2, img + G::B2, a1, a2, l31, hi);
#pragma unroll
      for (int b = 0; b < G::H2B; ++b) {
        nf_lrelu16(a2[b]);
        m2[b] = nf_sign_mask16(a2[b]);
      }
      // a2 as bf16 triples, once for the four chunk waves: in B-operand order and transposed
      SplitC<G::H2B> s2;
      split_C<G::H2B>(a2, s2);
      nf_u32x4 *pc_ = reinterpret_cast<nf_u32x4 *>(wv + L::A2C) + lane;
#pragma unroll
      for (int kg = 0; kg < 2 * G::H2B; ++kg) {
        pc_[(kg * 3 + 0) * 64] = s2.h[kg];
        pc_[(kg * 3 + 1) * 64] = s2.m[kg];
        pc_[(kg * 3 + 2) * 64] = s2.l[kg];
      }
      split_to_lds_tr<G::H2B, 2 * G::H2B>(wv + L::A2T, s2, l31, hi);
    }
    __syncthreads();  // B0: every home tile's a2 triples are in LDS
    const int gb6 = 2 + 12 * (gcount6 < 4 ? gcount6 : 3);
    ++gcount6;
    (void)gb6;
    C6_STAMP(gb6 + 0);

    // ---------------- chunk phase ----------------
    f32x16 d2[G::H2B];
#pragma unroll
    for (int b = 0; b < G::H2B; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) d2[b][r] = 0.f;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      const long tt = grp * 4 + t;
      const bool tlive = tt < ntiles;
      if (tlive) {
        const char *wt_ = lds + t * L::WAVE_BYTES;  // wave t's region
        const long jt = tt * NF_TILE + l31;
        const bool tvalid = jt < a.N;
        const TileIO yt = make_tile_io(y, tt, a.d, l31, hi);
        const TileIO gt = make_tile_io(ybar, tt, a.d, l31, hi);
        const float lb = tvalid ? (lbar ? lbar[jt] : lbar_const) : 0.f;
        float yq[G::QCH], gq[G::QCH], xiq[G::QCH];
        constexpr int ROWS = G::NCH * G::QCH;
        const float *trow = a.tape.base + (tt * a.ncoup + a.k) * (ROWS * 64) + lane;
#pragma unroll
        for (int ql = 0; ql < G::QCH; ++ql) {
          const int q = ch * G::QCH + ql;
          yq[ql] = tile_load(yt, tile_soff(q / 16, q % 16, a.par_t));
          gq[ql] = tile_load(gt, tile_soff(q / 16, q % 16, a.par_t));
          xiq[ql] = trow[q * 64];
        }
        // ---- raw spline parameters of this chunk: out = W3[chunk] a2 + b3[chunk], six-term bf16 products
        f32x16 out[G::OBC];
#pragma unroll
        for (int ob = 0; ob < G::OBC; ++ob)
#pragma unroll
          for (int r = 0; r < 16; ++r) out[ob][r] = b3[(ch * G::OBC + ob) * 32 + nf_row(r, hi)];
        {
          const nf_u32x4 *pa = reinterpret_cast<const nf_u32x4 *>(wt_ + L::A2C) + lane;
#pragma unroll
          for (int kg = 0; kg < 2 * G::H2B; ++kg) {
            const nf_u32x4 xh = pa[(kg * 3 + 0) * 64], xm = pa[(kg * 3 + 1) * 64], xl = pa[(kg * 3 + 2) * 64];
#pragma unroll
            for (int ob = 0; ob < G::OBC; ++ob) {  // smallest terms first, as dense_fwd_b6
              out[ob] = RQS6_MFMA_W(W.f[kg][ob][2], xh, out[ob]);
              out[ob] = RQS6_MFMA_W(W.f[kg][ob][0], xl, out[ob]);
              out[ob] = RQS6_MFMA_W(W.f[kg][ob][1], xm, out[ob]);
              out[ob] = RQS6_MFMA_W(W.f[kg][ob][1], xh, out[ob]);
              out[ob] = RQS6_MFMA_W(W.f[kg][ob][0], xm, out[ob]);
              out[ob] = RQS6_MFMA_W(W.f[kg][ob][0], xh, out[ob]);
            }
          }
          RQS6_SETTLE(out);
        }
        // ---- the spline and its reverse pass, in place (as k_rqs_bwd_coop)
#pragma unroll
        for (int ql = 0; ql < G::QCH; ++ql) {
          const int q = ch * G::QCH + ql;
          const int p = (q & 3) + 8 * (q >> 2) + 4 * hi;
          const bool ok = tvalid && p < a.c;
          float raw[G::P], thb[G::P];
          chunk_get<G>(out, ql, raw);
          Knots<G::K> kn;
          build_knots<G::K, RQS_COOP_LAZY>(raw, a.B, kn);
          const float yv = yq[ql];
          const float gv = ok ? gq[ql] : 0.f;
          Bin<G::K> bn;
          unsigned code;
          float xi;
          rqs_tape_decode(xiq[ql], code, xi);
          find_bin<G::K, RQS_COOP_LAZY, true>(kn, 0.f, bn, code);
          float xv;
          {
            const float dx = bn.dx, dy = bn.dy;
            if (INVD) {
              const float sl = nf_fdiv(dy, dx), om = 1.f - xi;
              const float den = sl + (bn.d1 + bn.d0 - 2.f * sl) * xi * om;
              xv = bn.inside ? bn.yk + nf_fdiv(dy * (sl * xi * xi + bn.d0 * xi * om), den) : yv;
            } else {
              xv = bn.inside ? fmaf(xi, dx, bn.xk) : yv;
            }
          }
          const float xbar = rqs_bwd_elem<G::K, INVD>(kn, bn, xi, a.B, gv, ok ? lb : 0.f, thb);
          chunk_put<G>(out, ql, thb);
          tile_store(yt, tile_soff(q / 16, q % 16, a.par_t), xv);
          tile_store(gt, tile_soff(q / 16, q % 16, a.par_t), xbar);
        }
#pragma unroll
        for (int slot = G::QCH * G::P; slot < G::OBC * 16; ++slot) out[slot / 16][slot % 16] = 0.f;
        // ---- dX3 (this chunk's share, to the home wave's slot) and dW3^T (this chunk's columns), block by block: the
        // cotangent block is split once per k-group; the triples feed the dX3 MFMAs from registers and go to the D6 ring
        // transposed, from where the dW3 MFMAs of the same block read them back
        {
          SplitT<G::H2B> a2t;  // tile t's a2^T triples (A operand of dW3)
          {
            const TrLane ta = nf_tr_lane(wt_ + L::A2T, l31, hi);
            constexpr int NT = 2 * G::H2B;
#pragma unroll
            for (int ib = 0; ib < G::H2B; ++ib)
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                a2t.h[ib][g] = nf_tr_operand(ta.p0, ta.p1, (0 * NT + 2 * ib) * TR_TILE + 512 * g);
                a2t.m[ib][g] = nf_tr_operand(ta.p0, ta.p1, (1 * NT + 2 * ib) * TR_TILE + 512 * g);
                a2t.l[ib][g] = nf_tr_operand(ta.p0, ta.p1, (2 * NT + 2 * ib) * TR_TILE + 512 * g);
              }
          }
          f32x16 d2p[G::H2B];  // (one chain: the dW3 MFMAs of the previous block run beside it)
#pragma unroll
          for (int b = 0; b < G::H2B; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) d2p[b][r] = 0.f;
          const unsigned ones = 0x3F803F80u;  // bf16 (1, 1)
#pragma unroll
          for (int pc = 0; pc < G::OBC; ++pc) {
            char *ring = wv + L::DT + (pc & 1) * L::RING_BLK;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int kg = 2 * pc + g;
              float v[8];
#pragma unroll
              for (int jj = 0; jj < 8; ++jj) v[jj] = out[pc][8 * g + jj];
              nf_u32x4 dh, dm, dl;
              nf_split8(v, dh, dm, dl);
              kg_to_lds_tr<2>(ring, g, dh, dm, dl, l31, hi);
#pragma unroll
              for (int ib = 0; ib < G::H2B; ++ib) {
                f32x16 &dd = d2p[ib];
                dd = RQS6_MFMA_W(W.t[kg][ib][2], dh, dd);
                dd = RQS6_MFMA_W(W.t[kg][ib][0], dl, dd);
                dd = RQS6_MFMA_W(W.t[kg][ib][1], dm, dd);
                dd = RQS6_MFMA_W(W.t[kg][ib][1], dh, dd);
                dd = RQS6_MFMA_W(W.t[kg][ib][0], dm, dd);
                dd = RQS6_MFMA_W(W.t[kg][ib][0], dh, dd);
              }
            }
            wave_lds_order();
            // dW3^T[:, block pc] += a2^T delta_pc^T, bias gradient = row sums of delta_pc
            const TrLane tr = nf_tr_lane(ring, l31, hi);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const nf_u32x4 dc0 = nf_tr_operand(tr.p0, tr.p1, 0 * 2 * TR_TILE + 512 * g), dc1 = nf_tr_operand(tr.p0, tr.p1, 1 * 2 * TR_TILE + 512 * g),
                             dc2 = nf_tr_operand(tr.p0, tr.p1, 2 * 2 * TR_TILE + 512 * g);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                acc.b3[pc] = nf_dot2_bf16(dc0[i], ones, acc.b3[pc]);
                acc.b3[pc] = nf_dot2_bf16(dc1[i], ones, acc.b3[pc]);
                acc.b3[pc] = nf_dot2_bf16(dc2[i], ones, acc.b3[pc]);
              }
#pragma unroll
              for (int ib = 0; ib < G::H2B; ++ib) {
                f32x16 &ww = acc.w3[ib][pc];
                ww = nf_mfma_bf16(a2t.l[ib][g], dc0, ww);
                ww = nf_mfma_bf16(a2t.h[ib][g], dc2, ww);
                ww = nf_mfma_bf16(a2t.m[ib][g], dc1, ww);
                ww = nf_mfma_bf16(a2t.m[ib][g], dc0, ww);
                ww = nf_mfma_bf16(a2t.h[ib][g], dc1, ww);
                ww = nf_mfma_bf16(a2t.h[ib][g], dc0, ww);
              }
            }
            wave_lds_order();
          }
          RQS6_SETTLE(d2p);
          float *mys = slots + ((t & 1) * 4 + wave) * L::SLOT;
#pragma unroll
          for (int b = 0; b < G::H2B; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mys[(b * 16 + r) * 64 + lane] = d2p[b][r];
        }
      }
      __syncthreads();  // B1: the four partial d2 of tile t are in the slots (set t & 1)
      C6_STAMP(gb6 + 1 + 2 * t);
      if (tlive && t == wave) {
        const float *set = slots + (t & 1) * 4 * L::SLOT;
#pragma unroll
        for (int w = 0; w < 4; ++w)  // fixed order: deterministic
#pragma unroll
          for (int b = 0; b < G::H2B; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) d2[b][r] += set[w * L::SLOT + (b * 16 + r) * 64 + lane];
      }
      C6_STAMP(gb6 + 2 + 2 * t);
    }

    // ---------------- home phase: layers 2 and 1 of this wave's own tile ----------------
    // (the wave's delta ring is dead until the next group's chunk phase: it holds the fp32 transpose tiles now)
    if (live) {
      tile_to_scratch<G::MB>(tiles + L::TILE_X, xb, l31, hi);
      tile_to_scratch<G::H1B>(tiles + L::TILE_A1, a1, l31, hi);
    }
    if (grp + gridDim.x < ngroups) load_home(grp + gridDim.x);  // in flight behind the 64 MFMAs of this phase
    if (live) {
      float *sd = tiles + L::TILE_D;
#pragma unroll
      for (int b = 0; b < G::H2B; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) d2[b][r] *= nf_mask_slope(m2[b], r);
      tile_to_scratch<G::H2B>(sd, d2, l31, hi);
      wave_lds_order();
      dw_accumulate<G::H1B, G::H2B>(tiles + L::TILE_A1, sd, acc.w2, acc.b2, l31, hi);
      f32x16 d1[G::H1B];
      dense_bwd_x<G::H1B, G::H2B>(img + G::W2, d2, d1, l31, hi);
      f32x16 g2[G::MB];
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) g2[b][r] = tile_load(gio, tile_soff(b, r, par_c));
#pragma unroll
      for (int b = 0; b < G::H1B; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) d1[b][r] *= nf_mask_slope(m1[b], r);
      wave_lds_order();
      tile_to_scratch<G::H1B>(sd, d1, l31, hi);
      wave_lds_order();
      dw_accumulate<G::MB, G::H1B>(tiles + L::TILE_X, sd, acc.w1, acc.b1, l31, hi);
      dense_bwd_x<G::MB, G::H1B, G::S1, true>(img + G::W1, d1, g2, l31, hi);
      wave_lds_order();
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile_store(gio, tile_soff(b, r, par_c), g2[b][r]);
    }
    C6_STAMP(gb6 + 9);
  }
  __syncthreads();  // every working area is dead: the LDS becomes the fold target (the slab image, G's fp32 layout)
  C6_STAMP(60);
  // Layers 1-2: every wave leaves its own sums in its own copy of the image's first G::W3 floats (behind the slab image), and
  // the slab write adds the four copies in wave order -- ((w0 + w1) + w2) + w3, the association of the waves-in-turn fold this
  // replaces (round 5, tools/trace_rqs6.py: that fold and its four barriers were 6.5 k of a launch's 247 k clocks).  The
  // output layer's chunk belongs to this wave alone.
  float *fold = reinterpret_cast<float *>(lds);
  static_assert(((size_t)G::SIZE + 4 * (size_t)G::W3) * 4 <= L::BYTES && G::W3 % 4 == 0, "the four partial images fit behind the slab image");
  {
    float *mine = fold + G::SIZE + wave * G::W3;
    rqs_fold<G::MB, G::H1B, G::S1>(mine + G::W1, mine + G::B1, acc.w1, acc.b1, true, l31, hi);
    rqs_fold<G::H1B, G::H2B, G::S2>(mine + G::W2, mine + G::B2, acc.w2, acc.b2, true, l31, hi);
  }
  rqs_fold<G::H2B, G::OBC, G::S3>(fold + G::W3 + ch * G::OBC * 32, fold + G::B3 + ch * G::OBC * 32, acc.w3, acc.b3, true, l31, hi);
  __syncthreads();
  C6_STAMP(61);
  {
    const float4 *c0 = reinterpret_cast<const float4 *>(fold), *p0 = reinterpret_cast<const float4 *>(fold + G::SIZE);
    float4 *dst = reinterpret_cast<float4 *>(slab + (long)blockIdx.x * slab_stride);
    constexpr int Q = G::W3 / 4;
    for (int i = tid; i < G::SIZE / 4; i += 256) {
      float4 v;
      if (i < Q) {
        const float4 u0 = p0[i], u1 = p0[Q + i], u2 = p0[2 * Q + i], u3 = p0[3 * Q + i];
        v = float4{((u0.x + u1.x) + u2.x) + u3.x, ((u0.y + u1.y) + u2.y) + u3.y, ((u0.z + u1.z) + u2.z) + u3.z, ((u0.w + u1.w) + u2.w) + u3.w};
      } else {
        v = c0[i];
      }
      dst[i] = v;
    }
  }
  __syncthreads();
  C6_STAMP(62);
}

template <class G, bool INVD>
__global__ __launch_bounds__(256, 1) void k_rqs_bwd_coop6(RqsBwdArgs a, float *__restrict__ y, float *__restrict__ ybar,
                                                          const float *__restrict__ lbar, float lbar_const,
                                                          float *__restrict__ slab, long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) char lds6[];
  rqs_bwd_coop6_coupling<G, INVD>(a, y, ybar, lbar, lbar_const, slab, slab_stride, lds6);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
using GeoK8 = RqsGeo<1, 1, 1, 8, 4>;    // d <= 32, hidden <= 32, K = 8   (cfg 3)
using GeoK10 = RqsGeo<1, 1, 1, 10, 2>;  // d <= 16, hidden <= 32, K = 10  (test/flow.jl:65-78)
using GeoK10L = RqsGeo<1, 1, 1, 10, 4>; // d <= 32, hidden <= 32, K = 10: the reference's default nsf(q0) widths and K
                                        // (neuralspline.jl:232-234) up to d = 32; reverse pass only in the cooperative form

static inline int b32(int n) { return (n + 31) / 32; }

// 1 -> GeoK8, 2 -> GeoK10, 3 -> GeoK10L, 0 -> unsupported
#define RQS_DISPATCH(ID, CALL) ((ID) == 1 ? CALL(GeoK8) : (ID) == 2 ? CALL(GeoK10) : CALL(GeoK10L))
#define RQS_DISPATCH_STMT(ID, CALL) \
  do {                              \
    if ((ID) == 1) { CALL(GeoK8); } else if ((ID) == 2) { CALL(GeoK10); } else { CALL(GeoK10L); } \
  } while (0)
static int rqs_geo_id(const nf_flow_desc *desc) {
  if (desc->n_hidden != 2) return 0;
  const int cmax = (desc->d + 1) / 2;
  if (b32(desc->hdims[0]) != 1 || b32(desc->hdims[1]) != 1 || b32(cmax) != 1) return 0;
  if (desc->K == 8 && cmax <= GeoK8::CMAX) return 1;
  if (desc->K == 10 && cmax <= GeoK10::CMAX) return 2;
  if (desc->K == 10 && cmax <= GeoK10L::CMAX) return 3;
  return 0;
}

bool nf_rqs_supported(const nf_flow_desc *desc) { return desc->d >= 2 && desc->B > 0.f && rqs_geo_id(desc) != 0; }

static int rqs_geo_size(const nf_flow_desc *desc) {
  const int id = rqs_geo_id(desc);
  return id == 1 ? GeoK8::SIZE : id == 2 ? GeoK10::SIZE : id == 3 ? GeoK10L::SIZE : 0;
}

static RqsPackArgs rqs_pack_args(const nf_flow_desc *desc) {
  RqsPackArgs p;
  p.d = desc->d; p.h1 = desc->hdims[0]; p.h2 = desc->hdims[1]; p.ncoup = 2 * desc->nlayers;
  const CouplingInfo c0 = nf_coupling_info(desc, 0), c1 = nf_coupling_info(desc, 1);
  p.odd_params = c0.nparams;
  p.pair_params = c0.nparams + c1.nparams;
  return p;
}

// the spline tape of a batch of N: [tile][coupling][NE rows][64 lanes] floats (RqsTape)
static int rqs_ne(const nf_flow_desc *desc) {
  const int id = rqs_geo_id(desc);
  return id == 1 ? GeoK8::NCH * GeoK8::QCH : id == 2 ? GeoK10::NCH * GeoK10::QCH : GeoK10L::NCH * GeoK10L::QCH;
}
size_t nf_rqs_tape_bytes(const nf_flow_desc *desc, long N) {
  if (!rqs_geo_id(desc)) return 0;
  const size_t slots = (size_t)((N + NF_TILE - 1) / NF_TILE) * (size_t)(2 * desc->nlayers);
  return slots * 64 * 4 * (size_t)rqs_ne(desc);
}
static RqsTape rqs_tape_at(const nf_flow_desc *desc, long N, void *tape) {
  (void)desc;
  (void)N;
  RqsTape t;
  t.base = (float *)tape;
  return t;
}

long nf_rqs_slab_floats(const nf_flow_desc *desc) { return (long)2 * desc->nlayers * rqs_geo_size(desc); }

// ctx->wimg of an NSF flow: [fp32 images][256-byte aligned: the output layers as bf16 triples (RqsB6Geo; the geometries whose
// reverse pass runs the cooperative bf16 form)]
static size_t rqs_fp32_bytes(const nf_flow_desc *desc) { return ((size_t)2 * desc->nlayers * rqs_geo_size(desc) * sizeof(float) + 255) / 256 * 256; }
static size_t rqs_b6_bytes_per_coupling(const nf_flow_desc *desc) {
  const int id = rqs_geo_id(desc);
  return id == 1 ? RqsB6Geo<GeoK8>::BYTES : 0;
}
size_t nf_rqs_wimg_bytes(const nf_flow_desc *desc) { return rqs_fp32_bytes(desc) + (size_t)2 * desc->nlayers * rqs_b6_bytes_per_coupling(desc); }
static bool rqs_bwd_b6() {
  static const bool off = std::getenv("NF_RQS_BWD_FP32") != nullptr;  // A/B switch: the fp32-MFMA cooperative kernel of rounds 2-4
  return !off;
}
// the triple images, rebuilt from the fp32 images when those have been rewritten since (ctx->wimg_gen)
template <class G>
static int rqs_b6_refresh(nf_ctx *ctx, const nf_flow_desc *desc) {
  if (ctx->b6_gen == ctx->wimg_gen) return NF_OK;
  using B = RqsB6Geo<G>;
  const int nimg = 2 * desc->nlayers;
  const long total = (long)nimg * B::U4;
  ProfScope ps(ctx, "pack_weights");
  hipLaunchKernelGGL((k_rqs_b6_from_images<G>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, nimg, (const float *)ctx->wimg,
                     (nf_u32x4 *)((char *)ctx->wimg + rqs_fp32_bytes(desc)));
  NF_HIP(hipGetLastError());
  ctx->b6_gen = ctx->wimg_gen;
  return NF_OK;
}

int nf_rqs_pack(nf_ctx *ctx, const nf_flow_desc *desc, const float *theta) {
  const int id = rqs_geo_id(desc);
  if (!id) return NF_ERR_UNSUPPORTED;
  const int size = rqs_geo_size(desc);
  const int nc = 2 * desc->nlayers;
  NF_TRY(nf_wimg_reserve(ctx, nf_rqs_wimg_bytes(desc)));
  const RqsPackArgs p = rqs_pack_args(desc);
  const long total = (long)nc * size;
  const unsigned grid = (unsigned)((total + 255) / 256);
  ProfScope ps(ctx, "pack_weights");
#define RQS_CALL(G) hipLaunchKernelGGL((k_rqs_pack<G>), dim3(grid), dim3(256), 0, ctx->stream, p, theta, (float *)ctx->wimg)
  RQS_DISPATCH_STMT(id, RQS_CALL);
#undef RQS_CALL
  return (int)hipGetLastError();
}

int nf_rqs_reduce_slabs(nf_ctx *ctx, const nf_flow_desc *desc, const float *slab, int nslab, float *g) {
  const int id = rqs_geo_id(desc);
  if (!id) return NF_ERR_UNSUPPORTED;
  const RqsPackArgs p = rqs_pack_args(desc);
  const long total = (long)p.ncoup * rqs_geo_size(desc);
  const unsigned grid = (unsigned)((total + 63) / 64);  // 64 elements per block
  ProfScope ps(ctx, "reduce_slabs");
#define RQS_CALL(G) hipLaunchKernelGGL((k_rqs_reduce_slabs<G>), dim3(grid), dim3(256), 0, ctx->stream, p, slab, nslab, total, g)
  RQS_DISPATCH_STMT(id, RQS_CALL);
#undef RQS_CALL
  return (int)hipGetLastError();
}

long nf_rqs_epilogue_blocks(const nf_flow_desc *desc) { return ((long)2 * desc->nlayers * rqs_geo_size(desc) + 63) / 64; }
// reduce + Adam + the images of the updated theta in one launch (single-rank nf_elbo_step); gpart[nf_rqs_epilogue_blocks] receives the
// blocks' partials of sum g^2 (finished by nf_launch_finish_sum), g[P] the loss
int nf_rqs_epilogue(nf_ctx *ctx, const nf_flow_desc *desc, const float *slab, int nslab, float *g, const double *lpart, int nlpart,
                    float *theta, float *m, float *v, double lr, double b1, double b2, double eps, unsigned t_val, double *gpart) {
  const int id = rqs_geo_id(desc);
  if (!id || !ctx->wimg) return NF_ERR_UNSUPPORTED;
  const RqsPackArgs p = rqs_pack_args(desc);
  RqsEpiArgs a;
  a.slab = slab; a.nslab = nslab; a.slab_stride = (long)p.ncoup * rqs_geo_size(desc);
  a.g = g; a.P = nf_param_count(desc);
  a.lpart = lpart; a.nlpart = nlpart;
  a.theta = theta; a.m = m; a.v = v; a.wimg = (float *)ctx->wimg;
  a.lr = (float)lr; a.b1 = (float)b1; a.b2 = (float)b2; a.eps = (float)eps;
  a.c1 = (float)(1.0 - pow(b1, (double)t_val + 1.0));
  a.c2 = (float)(1.0 - pow(b2, (double)t_val + 1.0));
  a.gpart = gpart;
  const unsigned grid = (unsigned)nf_rqs_epilogue_blocks(desc);
  ctx->wimg_gen++;  // the fp32 images are rewritten (Adam's theta): the triple images are stale
  ProfScope ps(ctx, "reduce_slabs");
#define RQS_CALL(G) hipLaunchKernelGGL((k_rqs_epilogue<G>), dim3(grid), dim3(256), 0, ctx->stream, p, a)
  RQS_DISPATCH_STMT(id, RQS_CALL);
#undef RQS_CALL
  return (int)hipGetLastError();
}

static bool rqs_fwd_b6() {
  static const bool off = std::getenv("NF_RQS_FWD_FP32") != nullptr;  // A/B switch: the chain kernel's output layer on fp32 MFMAs
  return !off;
}
template <class G>
static int launch_rqs_chain(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, float *xt, long N, float *ladj,
                            int k_only, const RqsFusedArgs *fused = nullptr, void *tape = nullptr) {
  RqsChainArgs a;
  a.tape = rqs_tape_at(desc, N, tape);
  a.wimg = (const float *)ctx->wimg;
  a.wimg6 = nullptr;
  a.d = desc->d; a.ncoup = 2 * desc->nlayers; a.B = desc->B; a.N = N; a.k_only = k_only;
  a.trace = (long long *)ctx->trace;
  const long ngroups = ((N + NF_TILE - 1) / NF_TILE + 7) / 8;
  long grid = ngroups < ctx->num_cu ? ngroups : ctx->num_cu;
  if (grid < 1) grid = 1;
  RqsFusedArgs none{};
  // whole-chain launches of the K = 8 geometry (cfg 3): the output layer as six-term bf16 products from the triple images
  if constexpr (std::is_same<G, GeoK8>::value) {
    if (k_only < 0 && rqs_fwd_b6()) {
      const size_t lds6 = RqsChainLds<G, true>::BYTES;
      static AttrOnce attr_once6;  // once per device
      NF_TRY(attr_once6.run(ctx->device, [&]() -> int {
        NF_HIP(hipFuncSetAttribute((const void *)k_rqs_chain<G, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6));
        NF_HIP(hipFuncSetAttribute((const void *)k_rqs_chain<G, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6));
        NF_HIP(hipFuncSetAttribute((const void *)k_rqs_chain<G, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6));
        return NF_OK;
      }));
      NF_TRY(rqs_b6_refresh<G>(ctx, desc));
      a.wimg6 = (const nf_u32x4 *)((const char *)ctx->wimg + rqs_fp32_bytes(desc));
      ProfScope ps(ctx, "rqs_chain");
      if (fused)
        hipLaunchKernelGGL((k_rqs_chain<G, false, true, true>), dim3((unsigned)grid), dim3(512), lds6, ctx->stream, a, xt, ladj, *fused);
      else if (inverse)
        hipLaunchKernelGGL((k_rqs_chain<G, true, false, true>), dim3((unsigned)grid), dim3(512), lds6, ctx->stream, a, xt, ladj, none);
      else
        hipLaunchKernelGGL((k_rqs_chain<G, false, false, true>), dim3((unsigned)grid), dim3(512), lds6, ctx->stream, a, xt, ladj, none);
      return (int)hipGetLastError();
    }
  }
  // two double-buffered images + target parameters and per-wave sums of the fused variant
  const size_t lds = RqsChainLds<G, false>::BYTES;
  static AttrOnce attr_once;  // once per device: a context on another GPU needs its own
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_rqs_chain<G, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NF_HIP(hipFuncSetAttribute((const void *)k_rqs_chain<G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NF_HIP(hipFuncSetAttribute((const void *)k_rqs_chain<G, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  ProfScope ps(ctx, "rqs_chain");
  if (fused)
    hipLaunchKernelGGL((k_rqs_chain<G, false, true>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a, xt, ladj, *fused);
  else if (inverse)
    hipLaunchKernelGGL((k_rqs_chain<G, true>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a, xt, ladj, none);
  else
    hipLaunchKernelGGL((k_rqs_chain<G, false>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a, xt, ladj, none);
  return (int)hipGetLastError();
}

long nf_rqs_chain_grid(nf_ctx *ctx, long N) {
  const long ngroups = ((N + NF_TILE - 1) / NF_TILE + 7) / 8;
  const long grid = ngroups < ctx->num_cu ? ngroups : ctx->num_cu;
  return grid < 1 ? 1 : grid;
}

// base draws + whole chain forward + diagonal-Gaussian target + ELBO partial sums in one launch (packed images must
// be current).  yt <- flow output (tiled), gt <- gscale * grad log p(y) (or null), partial[nf_rqs_chain_grid] <- sums
// of pscale * elbo_j.
int nf_rqs_chain_elbo(nf_ctx *ctx, const nf_flow_desc *desc, long N, uint64_t seed, uint64_t off, uint32_t stream,
                      const float *mu, const float *var, float *yt, float *gt, double gscale, double *partial,
                      double pscale, void *tape) {
  const int id = rqs_geo_id(desc);
  if (!id || !ctx->wimg) return NF_ERR_UNSUPPORTED;
  RqsFusedArgs fa;
  fa.k0 = (uint32_t)seed; fa.k1 = (uint32_t)(seed >> 32); fa.stream = stream; fa.off = off;
  fa.mu = mu; fa.var = var; fa.gt = gt; fa.gscale = (float)gscale; fa.partial = partial; fa.pscale = pscale;
#define RQS_CALL(G) launch_rqs_chain<G>(ctx, desc, false, yt, N, nullptr, -1, &fa, tape)
  return RQS_DISPATCH(id, RQS_CALL);
#undef RQS_CALL
}

// whole chain (k_only < 0) or a single coupling (flat index k_only), in place on the tiled buffer
// tape (optional): the spline tape the reverse kernels need (nf_rqs_tape_bytes)
int nf_rqs_chain(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, float *xt, long N, float *ladj, int k_only, void *tape) {
  const int id = rqs_geo_id(desc);
  if (!id || !ctx->wimg) return NF_ERR_UNSUPPORTED;
#define RQS_CALL(G) launch_rqs_chain<G>(ctx, desc, inverse, xt, N, ladj, k_only, nullptr, tape)
  return RQS_DISPATCH(id, RQS_CALL);
#undef RQS_CALL
}

int nf_rqs_bwd_grid(nf_ctx *ctx, long N) {
  const long ntiles = (N + NF_TILE - 1) / NF_TILE;
  long grid = (ntiles + 3) / 4;
  if (grid > ctx->num_cu) grid = ctx->num_cu;
  return (int)(grid < 1 ? 1 : grid);
}

template <class G, bool INVD>
static int launch_rqs_bwd_coop(nf_ctx *ctx, const nf_flow_desc *desc, int k, float *y, float *ybar, const float *lbar,
                               float lbar_const, long N, float *slab, long slab_stride, int grid, void *tape) {
  const size_t lds = RqsCoopLds<G>::BYTES;
  static AttrOnce attr_once;  // once per device
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_rqs_bwd_coop<G, INVD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  const CouplingInfo ci = nf_coupling_info(desc, k);
  RqsBwdArgs a;
  a.tape = rqs_tape_at(desc, N, tape);
  a.k = k;
  a.ncoup = 2 * desc->nlayers;
  a.img = (const float *)ctx->wimg + (size_t)k * G::SIZE;
  a.img6 = nullptr;
  a.d = desc->d; a.c = ci.c; a.m = ci.m; a.par_t = ci.par_t; a.B = desc->B; a.N = N;
  a.trace = (long long *)ctx->trace;
  ProfScope ps(ctx, INVD ? "rqs_bwd_inv" : "rqs_bwd");
  hipLaunchKernelGGL((k_rqs_bwd_coop<G, INVD>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, y, ybar, lbar, lbar_const,
                     slab + (long)k * G::SIZE, slab_stride);
  return (int)hipGetLastError();
}

template <class G, bool INVD>
static int launch_rqs_bwd_coop6(nf_ctx *ctx, const nf_flow_desc *desc, int k, float *y, float *ybar, const float *lbar,
                                float lbar_const, long N, float *slab, long slab_stride, int grid, void *tape) {
  const size_t lds = RqsCoop6Lds<G>::BYTES;
  static AttrOnce attr_once;  // once per device
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_rqs_bwd_coop6<G, INVD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  NF_TRY(rqs_b6_refresh<G>(ctx, desc));
  const CouplingInfo ci = nf_coupling_info(desc, k);
  RqsBwdArgs a;
  a.tape = rqs_tape_at(desc, N, tape);
  a.k = k;
  a.ncoup = 2 * desc->nlayers;
  a.img = (const float *)ctx->wimg + (size_t)k * G::SIZE;
  a.img6 = (const nf_u32x4 *)((const char *)ctx->wimg + rqs_fp32_bytes(desc)) + (size_t)k * RqsB6Geo<G>::U4;
  a.d = desc->d; a.c = ci.c; a.m = ci.m; a.par_t = ci.par_t; a.B = desc->B; a.N = N;
  a.trace = (long long *)ctx->trace;
  ProfScope ps(ctx, INVD ? "rqs_bwd_inv" : "rqs_bwd");
  hipLaunchKernelGGL((k_rqs_bwd_coop6<G, INVD>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, y, ybar, lbar, lbar_const,
                     slab + (long)k * G::SIZE, slab_stride);
  return (int)hipGetLastError();
}

template <class G, bool INVD>
static int launch_rqs_bwd(nf_ctx *ctx, const nf_flow_desc *desc, int k, float *y, float *ybar, const float *lbar,
                          float lbar_const, long N, float *slab, long slab_stride, int grid, void *tape) {
  if (!tape) return NF_ERR_ARG;  // the reverse kernels differentiate the forward's own bins (RqsTape)
  if constexpr (G::NCH == 4) {
    if constexpr (RqsCoop6Lds<G>::FITS) {
      static const bool perwave = std::getenv("NF_RQS_BWD_PERWAVE") != nullptr;
      if (rqs_bwd_b6() && !perwave) return launch_rqs_bwd_coop6<G, INVD>(ctx, desc, k, y, ybar, lbar, lbar_const, N, slab, slab_stride, grid, tape);
    }
    static const bool old_form = std::getenv("NF_RQS_BWD_PERWAVE") != nullptr;  // A/B switch: the per-wave-tile kernel
    if (!old_form || G::OB3 > 12) return launch_rqs_bwd_coop<G, INVD>(ctx, desc, k, y, ybar, lbar, lbar_const, N, slab, slab_stride, grid, tape);
  }
  if constexpr (G::OB3 > 12) {  // the per-wave-tile kernel would need more accumulators than there are registers
    return NF_ERR_UNSUPPORTED;
  } else {
  const size_t lds = RqsLds<G>::BYTES;
  static AttrOnce attr_once;  // once per device: a context on another GPU needs its own
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_rqs_bwd<G, INVD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  const CouplingInfo ci = nf_coupling_info(desc, k);
  RqsBwdArgs a;
  a.tape = rqs_tape_at(desc, N, tape);
  a.k = k;
  a.ncoup = 2 * desc->nlayers;
  a.img = (const float *)ctx->wimg + (size_t)k * G::SIZE;
  a.img6 = nullptr;
  a.d = desc->d; a.c = ci.c; a.m = ci.m; a.par_t = ci.par_t; a.B = desc->B; a.N = N;
  a.trace = (long long *)ctx->trace;
  ProfScope ps(ctx, INVD ? "rqs_bwd_inv" : "rqs_bwd");
  hipLaunchKernelGGL((k_rqs_bwd<G, INVD>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, y, ybar, lbar, lbar_const,
                     slab + (long)k * G::SIZE, slab_stride);
  return (int)hipGetLastError();
  }
}

// inv_dir: reverse pass of the INVERSE coupling k at its output (see rqs_bwd_tile)
int nf_rqs_bwd(nf_ctx *ctx, const nf_flow_desc *desc, int k, float *y, float *ybar, const float *lbar, float lbar_const,
               long N, float *slab, long slab_stride, int grid, bool inv_dir, void *tape) {
  const int id = rqs_geo_id(desc);
  if (!id || !ctx->wimg) return NF_ERR_UNSUPPORTED;
#define RQS_CALL(G) launch_rqs_bwd<G, true>(ctx, desc, k, y, ybar, lbar, lbar_const, N, slab, slab_stride, grid, tape)
  if (inv_dir) return RQS_DISPATCH(id, RQS_CALL);
#undef RQS_CALL
#define RQS_CALL(G) launch_rqs_bwd<G, false>(ctx, desc, k, y, ybar, lbar, lbar_const, N, slab, slab_stride, grid, tape)
  return RQS_DISPATCH(id, RQS_CALL);
#undef RQS_CALL
}

