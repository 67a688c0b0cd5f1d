// nf_wide.hip -- AffineCoupling (RealNVP) kernels for conditioner nets that do NOT fit in LDS
// (BASELINE cfg 4: d = 256, 16 couplings, hidden [256,256]: one net is 512 KB of weights).
//
// Reference arithmetic: src/flows/realnvp.jl:57-110 (same as nf_coupling.hip); conditioner
// src/flows/utils.jl:71-100.
//
// Design (MI355X): a workgroup is 4 wavefronts, each owning one 32-sample tile whose activations
// stay in registers in the MFMA C layout (nf_mfma.h).  The weights are STREAMED: the packed image
// of a net (nf_pack.h, rows padded to an odd stride) is cut into chunks of 32 input rows; chunk
// q+1 is copied global -> LDS by the DMA path (global_load_lds, no staging registers) into the
// other half of a double buffer while the matrix pipe works on chunk q.  One chunk feeds
// 16 x OB MFMAs per wave, so one barrier per chunk costs ~1 % at OB = 8.
//   forward/inverse : k_wide_apply   (S net, T net, affine update, ladj)
//   reverse pass    : k_wide_bwd<PHASE>  recompute + dX chain for ONE net; activations and
//                     deltas go to an HBM stash in [tile][feature][32 samples] layout, and
//                     k_wide_dw turns the stash into weight gradients: a split-K (over sample
//                     tiles) MFMA GEMM whose operands are transposed through LDS (row stride 33).
// The dW accumulators of a 256x256 layer (64 MFMA blocks = 1024 registers) cannot stay resident
// per wave the way nf_coupling.hip keeps them, hence the stash.  Per sample and net it is
// (2*h1 + 2*h2 + c) floats, written once and read once.
#include "nf_common.h"
// The split's two exact subtractions as scalar v_sub_f32 pairs in this file (nf_mfma.h: NF_SPLIT_SCALAR).  A packed f32 VALU
// instruction next to bf16 MFMAs costs more than the issue slot it saves with one wave per SIMD (these kernels); measured on
// one box, round 5: k_wide_dw_b6 56.2 -> 54.4 us, k_wide_bwd_stashed_b6 75.7 -> 74.4 us per launch.  The two-waves-per-SIMD
// kernels keep v_pk_add_f32 (the cfg-2 pair kernel: 336-338 against 331-334 us with scalar subtractions).
#define NF_SPLIT_SCALAR
#include "nf_mfma.h"
#include "nf_pack.h"

typedef __attribute__((address_space(3))) void lds_void_t;

// optional clock stamps of block 0 / wave 0 (nf_debug_trace, tools/trace_wide.py)
#ifdef NF_KERNEL_TRACE
#define WIDE_STAMP(slot)                                                  \
  do {                                                                    \
    if (tr) { __builtin_amdgcn_sched_barrier(0); tr[slot] = clock64(); }  \
  } while (0)
#else
#define WIDE_STAMP(slot) do { (void)tr; } while (0)
#endif

using GW = NetGeo<4, 8, 8, 4>;  // every wide flow is zero-padded into this geometry

template <class G>
struct Wide {
  static constexpr int CF1 = 32 * G::S1, CF2 = 32 * G::S2, CF3 = 32 * G::S3;  // chunk sizes (floats)
  static constexpr int CFMAX = CF1 > CF2 ? (CF1 > CF3 ? CF1 : CF3) : (CF2 > CF3 ? CF2 : CF3);
  static constexpr int CHBUF = ((CFMAX * 4 + 1023) / 1024) * 256;  // floats per buffer, whole 1-KiB DMA pieces
  static constexpr int NBIAS = 32 * (G::H1B + G::H2B + G::CB);
  static constexpr int WAVES = 4;
  static constexpr size_t LDS_APPLY = (size_t)(2 * CHBUF + 2 * NBIAS) * sizeof(float);
  static constexpr size_t LDS_BWD = (size_t)(2 * CHBUF + NBIAS) * sizeof(float);
  // B6 (round 4): the forward's GEMMs as six bf16 products of exactly split operands (nf_mfma.h); the streamed image is
  // the net's B6 image (bf16 triples, [k-group][component][half][row][8]), a chunk = the two k-groups of one input block:
  // 2 x 6 x rows x 16 bytes, contiguous.
  using B = B6Geo<G>;
  static constexpr int BF1 = 48 * B::R1, BF2 = 48 * B::R2, BF3 = 48 * B::R3;  // chunk sizes in floats (whole 1-KiB pieces)
  static constexpr int BFMAX = BF1 > BF2 ? (BF1 > BF3 ? BF1 : BF3) : (BF2 > BF3 ? BF2 : BF3);
  static constexpr size_t LDS_APPLY_B6 = (size_t)(2 * BFMAX + 2 * NBIAS) * sizeof(float);
};
// offsets (floats into the streamed image) and sizes of the chunks of the three layers, fp32 image or B6 image
template <class G, bool B6>
struct WideChunks {
  using W = Wide<G>;
  static constexpr int CHBUF = B6 ? W::BFMAX : W::CHBUF;
  static constexpr int N1 = B6 ? W::BF1 : W::CF1, N2 = B6 ? W::BF2 : W::CF2, N3 = B6 ? W::BF3 : W::CF3;
  static constexpr int O1 = B6 ? B6Geo<G>::L1 * 4 : G::W1, O2 = B6 ? B6Geo<G>::L2 * 4 : G::W2, O3 = B6 ? B6Geo<G>::L3 * 4 : G::W3;
  static constexpr int IMG_FLOATS = B6 ? B6Geo<G>::BYTES / 4 : G::SIZE;
};

// DMA one chunk global -> LDS (buffer_load_dwordx4 ... lds): 1 KiB (64 lanes x 16 B) per
// instruction, pieces dealt round-robin to the 4 waves.  `img` is a buffer descriptor over one
// net's packed image, `off` the chunk's offset in floats: all address arithmetic is scalar, the
// only vector register is lane * 16.  The last piece may run past the chunk into the following
// image bytes (descriptor bound: reads as 0; the LDS buffer is a whole number of pieces).
typedef __amdgpu_buffer_rsrc_t wide_img_t;
__device__ __forceinline__ wide_img_t make_img(const float *img, int nfloats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, nfloats * 4, 0x00020000);
}
__device__ __forceinline__ void issue_chunk(wide_img_t img, int off, int nfloats, float *ldst, int wave, int lane) {
  for (int p = wave; p * 256 < nfloats; p += 4)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(img, (lds_void_t *)(ldst + p * 256), 16, lane * 16, (off + p * 256) * 4, 0, 0);
}

// The DMA of the NEXT chunk, handed to the chunk routines, which issue it in one block in front of
// their MFMA loop.  (Measured: spreading the pieces between the MFMAs of the loop is slower --
// 157 vs 146 us forward, 179 vs 169 us reverse per launch -- anything placed between two fp32 MFMAs
// costs more issue time than it hides.)
struct DmaJob {
  wide_img_t img;
  int off, nfloats;
  float *ldst;
  int wave, lane;
  __device__ __forceinline__ void issue() const {
    if (nfloats) issue_chunk(img, off, nfloats, ldst, wave, lane);
  }
};

// out[ob] += W[rows of this chunk][:] * in_blk   (chunk = 32 input rows, all output columns)
template <int OB, int S>
__device__ __forceinline__ void wide_fwd_chunk(const float *__restrict__ ch, const f32x16 &in, f32x16 (&out)[OB], int l31,
                                               int hi, const DmaJob &dma) {
  dma.issue();
  const float *wl = ch + (4 * hi) * S + l31;
  float an[OB], ac[OB];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) an[ob] = wl[ob * 32];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) ac[ob] = an[ob];
    if (t + 1 < 16) {
      const int row = ((t + 1) & 3) + 8 * ((t + 1) >> 2);
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) an[ob] = wl[row * S + ob * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[ob], in[t], out[ob], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same chunk GEMM on the bf16 matrix cores: `ch` holds the two k-groups of this input block as bf16 triples, the input
// block is split here (16 values -> 2 x 3 x 4 registers); pipeline unit = (k-group, output block), its three A operands
// requested one unit ahead (dense_fwd_b6 of nf_mfma.h without the bias start).
// (Measured and removed: the next chunk's DMA spread over the units instead of issued in one block in front of the loop.  The
// block costs ~900 clocks per 48-KB chunk -- 12 LDS-DMA instructions per wave through the CU's one texture-address path -- but
// inside the loop every piece delays the dependent MFMA chain by ~40 clocks: 4.35 k against 4.68 k clocks per chunk in the
// trace, 115 / 80 us per launch against 114 / 77 for forward / reverse in the untraced kernels.)
// `neg`: contract with -in instead (a constant after unrolling; see FLIP3 of wide_net_fwd).
template <int OB>
__device__ __forceinline__ void wide_fwd_chunk_b6(const float *__restrict__ ch, const f32x16 &in, f32x16 (&out)[OB], int l31,
                                                  int hi, const DmaJob &dma, bool neg = false) {
  dma.issue();
  constexpr int ROWS = 32 * OB, NU = 2 * OB;
  const nf_u32x4 *wl = reinterpret_cast<const nf_u32x4 *>(ch) + hi * ROWS + l31;
  nf_u32x4 an[3], ac[3], xh, xm, xl;
#pragma unroll
  for (int c = 0; c < 3; ++c) an[c] = wl[c * 2 * ROWS];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int kg = u / OB, ob = u % OB;
    if (ob == 0) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = neg ? -in[8 * kg + j] : in[8 * kg + j];
      nf_split8(v, xh, xm, xl);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) ac[c] = an[c];
    if (u + 1 < NU) {
      const int kg1 = (u + 1) / OB, ob1 = (u + 1) % OB;
#pragma unroll
      for (int c = 0; c < 3; ++c) an[c] = wl[(kg1 * 3 + c) * 2 * ROWS + ob1 * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
    out[ob] = nf_mfma_bf16(ac[2], xh, out[ob]);  // smallest terms first: wl xh, wh xl, wm xm, wm xh, wh xm, wh xh
    out[ob] = nf_mfma_bf16(ac[0], xl, out[ob]);
    out[ob] = nf_mfma_bf16(ac[1], xm, out[ob]);
    out[ob] = nf_mfma_bf16(ac[1], xh, out[ob]);
    out[ob] = nf_mfma_bf16(ac[0], xm, out[ob]);
    out[ob] = nf_mfma_bf16(ac[0], xh, out[ob]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// din = W[rows of this chunk][:]^T-contracted with delta: din[i] = sum_o W[i][o] delta[o]
template <int OB, int S>
__device__ __forceinline__ void wide_bwdx_chunk(const float *__restrict__ ch, const f32x16 (&delta)[OB], f32x16 &din,
                                                int l31, int hi, const DmaJob &dma) {
  dma.issue();
  const float *wl = ch + l31 * S + 4 * hi;
  constexpr int NG = OB * 4;
  float an[4], ac[4];
#pragma unroll
  for (int r = 0; r < 16; ++r) din[r] = 0.f;
  nf_ld4<S>(wl, an[0], an[1], an[2], an[3]);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) ac[e] = an[e];
    if (g + 1 < NG) {
      nf_ld4<S>(wl + (g + 1) * 8, an[0], an[1], an[2], an[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      din = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[e], delta[g / 4][(g % 4) * 4 + e], din, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Hooks of wide_net_fwd.  after_l1/after_l2 see the post-activation hidden layers; l2_step(ib) /
// l3_step(ib) run right before the MFMAs of chunk `ib` of layer 2 / 3 -- the reverse pass uses them
// to trickle its stash stores and tile loads out a few per chunk, so that they drain behind the
// matrix pipe instead of in one HBM-bound burst.
struct NoHooks {
  static constexpr int YOUNG = 0;
  template <class T>
  __device__ __forceinline__ void after_l1(T &) const {}
  template <class T>
  __device__ __forceinline__ void after_l2(T &) const {}
  template <class T>
  __device__ __forceinline__ void l2_step(int, T &) const {}
  template <class T>
  __device__ __forceinline__ void l3_step(int, T &) const {}
};

// The barrier behind a chunk: every wave is done with the chunk, and the NEXT chunk's DMA (issued in front of this chunk's
// hook and GEMM) has landed.  YOUNG = vector-memory operations a hook issues per chunk BEHIND that DMA (the stash stores
// trickled out 16 per chunk): the counter retires in order, so "at most YOUNG outstanding" already implies the DMA is done,
// while __syncthreads() (vmcnt(0)) would also wait for those stores' round trip to HBM -- affordable behind 128 fp32 MFMAs
// of 64 clocks, not behind 96 bf16 ones of 32.
template <int YOUNG>
__device__ __forceinline__ void wide_chunk_barrier() {
  if constexpr (YOUNG == 0) {
    __syncthreads();
  } else {
    static_assert(YOUNG < 64, "vmcnt has six bits");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt((YOUNG & 0xF) | 0x70 | ((YOUNG >> 4) << 14));  // vmcnt(YOUNG) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <int NB>
__device__ __forceinline__ void init_bias(f32x16 (&v)[NB], const float *__restrict__ b, int hi) {
#pragma unroll
  for (int ob = 0; ob < NB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[ob][r] = b[ob * 32 + nf_row(r, hi)];
}

// Streams one net forward.  On entry the chunk (W1, rows 0..31) of `img` is resident in
// cb[buf]; on exit the chunk (next_src, next_floats) is resident in cb[buf] (or nothing if
// next_floats == 0).
// FLIP3 (bf16 form, the scale net): the output layer accumulates its first H2B / 2 input blocks as they are and the rest in
// the NEGATED frame (accumulator and inputs negated at the half-way point, the result negated back at the end).  The bf16
// MFMA's internal adder drops low bits toward minus infinity whatever the signs (tools/probe/split_bias_probe.hip: mean
// error -0.017 of 2^-24 sum|terms| per output against +0.0003 for the fp32 chain; a twentieth of the rms error, invisible in
// any single output) -- but log|det J| of a coupling is the SUM of c = 128 outputs' tanh, 2 048 per sample over cfg 4's 16
// couplings, and a one-sided error grows with n where rounding grows with sqrt n: tools/parity_ab.py measured a mean of
// -0.24 of the tolerance on cfg 4's ladj (rms 0.43) against +0.002 (0.25) for fp32 MFMAs.  Half the accumulation steps in
// the negated frame make the drift cancel on average; cost: 2 x 16 CB sign flips per net and tile.
template <class G, class HK, bool B6 = false, bool FLIP3 = false>
__device__ __forceinline__ void wide_net_fwd(wide_img_t img, const float *__restrict__ bias, float *cb, int &buf,
                                             wide_img_t next_img, int next_off, int next_floats,
                                             const f32x16 (&xb)[G::MB], f32x16 (&out)[G::CB], int wave, int lane, HK &hk,
                                             long long *tr = nullptr) {
  using C = WideChunks<G, B6>;
  const int l31 = lane & 31, hi = lane >> 5;
  f32x16 a1[G::H1B];
  init_bias<G::H1B>(a1, bias, hi);
#pragma unroll
  for (int ib = 0; ib < G::MB; ++ib) {
    DmaJob dj{img, 0, 0, cb, wave, lane};
    if (ib + 1 < G::MB)
      dj = DmaJob{img, C::O1 + (ib + 1) * C::N1, C::N1, cb + (buf ^ 1) * C::CHBUF, wave, lane};
    else
      dj = DmaJob{img, C::O2, C::N2, cb + (buf ^ 1) * C::CHBUF, wave, lane};
    if constexpr (B6) wide_fwd_chunk_b6<G::H1B>(cb + buf * C::CHBUF, xb[ib], a1, l31, hi, dj);
    else wide_fwd_chunk<G::H1B, G::S1>(cb + buf * C::CHBUF, xb[ib], a1, l31, hi, dj);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int b = 0; b < G::H1B; ++b)
    nf_lrelu16(a1[b]);
  WIDE_STAMP(2);
  hk.after_l1(a1);
  WIDE_STAMP(3);
  f32x16 a2[G::H2B];
  init_bias<G::H2B>(a2, bias + 32 * G::H1B, hi);
#pragma unroll
  for (int ib = 0; ib < G::H1B; ++ib) {
    DmaJob dj{img, 0, 0, cb, wave, lane};
    if (ib + 1 < G::H1B)
      dj = DmaJob{img, C::O2 + (ib + 1) * C::N2, C::N2, cb + (buf ^ 1) * C::CHBUF, wave, lane};
    else
      dj = DmaJob{img, C::O3, C::N3, cb + (buf ^ 1) * C::CHBUF, wave, lane};
    const DmaJob none{img, 0, 0, cb, wave, lane};
    if constexpr (B6) {  // the DMA in front of the hook's stores (wide_chunk_barrier)
      if (ib < 4) WIDE_STAMP(32 + 4 * ib);  // (tools/trace_wide_apply.py: inside the first chunks of layer 2)
      dj.issue();
      hk.l2_step(ib, a1);
      if (ib < 4) WIDE_STAMP(33 + 4 * ib);
      wide_fwd_chunk_b6<G::H2B>(cb + buf * C::CHBUF, a1[ib], a2, l31, hi, none);
      if (ib < 4) WIDE_STAMP(34 + 4 * ib);
      wide_chunk_barrier<HK::YOUNG>();
      if (ib < 4) WIDE_STAMP(35 + 4 * ib);
    } else {
      hk.l2_step(ib, a1);
      wide_fwd_chunk<G::H2B, G::S2>(cb + buf * C::CHBUF, a1[ib], a2, l31, hi, dj);
      __syncthreads();
    }
    buf ^= 1;
  }
#pragma unroll
  for (int b = 0; b < G::H2B; ++b)
    nf_lrelu16(a2[b]);
  WIDE_STAMP(4);
  hk.after_l2(a2);
  WIDE_STAMP(5);
  init_bias<G::CB>(out, bias + 32 * (G::H1B + G::H2B), hi);
#pragma unroll
  for (int ib = 0; ib < G::H2B; ++ib) {
    DmaJob dj{img, 0, 0, cb, wave, lane};
    if (ib + 1 < G::H2B)
      dj = DmaJob{img, C::O3 + (ib + 1) * C::N3, C::N3, cb + (buf ^ 1) * C::CHBUF, wave, lane};
    else if (next_floats)
      dj = DmaJob{next_img, next_off, next_floats, cb + (buf ^ 1) * C::CHBUF, wave, lane};
    const DmaJob none{img, 0, 0, cb, wave, lane};
    if constexpr (B6) {
      dj.issue();
      hk.l3_step(ib, a2);
      constexpr bool FL = FLIP3 && G::H2B >= 2;
      if (FL && ib == G::H2B / 2) {
#pragma unroll
        for (int ob = 0; ob < G::CB; ++ob) out[ob] = -out[ob];
      }
      wide_fwd_chunk_b6<G::CB>(cb + buf * C::CHBUF, a2[ib], out, l31, hi, none, FL && ib >= G::H2B / 2);
      if (FL && ib == G::H2B - 1) {
#pragma unroll
        for (int ob = 0; ob < G::CB; ++ob) out[ob] = -out[ob];
      }
      wide_chunk_barrier<HK::YOUNG>();
    } else {
      hk.l3_step(ib, a2);
      wide_fwd_chunk<G::CB, G::S3>(cb + buf * C::CHBUF, a2[ib], out, l31, hi, dj);
      __syncthreads();
    }
    buf ^= 1;
  }
}

template <class G>
__device__ __forceinline__ void stage_biases(float *__restrict__ dst, const float *__restrict__ img, int tid) {
  for (int i = tid; i < Wide<G>::NBIAS; i += 256) {
    int src;
    if (i < 32 * G::H1B) src = G::B1 + i;
    else if (i < 32 * (G::H1B + G::H2B)) src = G::B2 + i - 32 * G::H1B;
    else src = G::B3 + i - 32 * (G::H1B + G::H2B);
    dst[i] = img[src];
  }
}

struct WideArgs {
  const float *img_s, *img_t;
  const unsigned char *b6_s, *b6_t;  // the nets' B6 images (k_wide_apply<..., B6 = true>), or nullptr
  const unsigned char *b6t_s, *b6t_t;  // their transposed B6T images (k_wide_bwd_stashed_b6)
  long long *trace;  // optional clock stamps of block 0 / wave 0 (nf_debug_trace)
  int d, c, m, par_t;
  long N;
};

// ------------------------------------------------------------------------------------
// forward / inverse of one coupling, in place on the tiled batch
// ------------------------------------------------------------------------------------
struct WideStash {
  float *a1, *a2, *d1, *d2, *d3;
};

struct StashIO {
  __amdgpu_buffer_rsrc_t rs;
  int voff;
};
__device__ __forceinline__ StashIO make_stash_io(float *t, long tile, int F, int l31, int hi) {
  StashIO s;
  s.rs = __builtin_amdgcn_make_buffer_rsrc(t + tile * F * NF_TILE, 0, F * NF_TILE * 4, 0x00020000);
  s.voff = l31 * 4 + hi * (4 * NF_TILE * 4);
  return s;
}
template <int NB>
__device__ __forceinline__ void stash_store(const StashIO &s, const f32x16 (&v)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float val = v[b][r];  // (bit_cast straight from the vector-element expression reads element 0)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), s.rs, s.voff,
                                            (b * 32 + (r & 3) + 8 * (r >> 2)) * (NF_TILE * 4), 0);
    }
}

// elements [e0, e0 + n) of the flattened (block, reg) index space
template <int NB>
__device__ __forceinline__ void stash_store_range(const StashIO &s, const f32x16 (&v)[NB], int e0, int n) {
#pragma unroll
  for (int q = 0; q < NB * 16; ++q)
    if (q >= e0 && q < e0 + n) {
      const float val = v[q >> 4][q & 15];
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), s.rs, s.voff,
                                            ((q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2)) * (NF_TILE * 4), 0);
    }
}

template <int NB>
__device__ __forceinline__ void wide_sign_masks(const f32x16 (&v)[NB], unsigned (&m)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) m[b] = nf_sign_mask16(v[b]);
}

// hooks of the forward kernel: optional stash (STASH) and, for the t net, the prefetch of the
// transformed half x1 behind the last layer
template <class G, bool STASH>
struct ApplyHooks {
  static constexpr int YOUNG = STASH ? 16 : 0;  // stash stores per l2_step / l3_step (wide_chunk_barrier)
  StashIO sa1, sa2;
  unsigned *mask;  // this lane's slot in the tile's mask block, or nullptr
  const TileIO *io;
  f32x16 (*x1)[G::CB];
  int par_t;
  __device__ __forceinline__ void after_l1(f32x16 (&a1)[G::H1B]) const {
    if (STASH) {
      unsigned m[G::H1B];
      wide_sign_masks<G::H1B>(a1, m);
      if (mask) {
#pragma unroll
        for (int b = 0; b < G::H1B; ++b) mask[b * 64] = m[b];
      }
    }
  }
  __device__ __forceinline__ void after_l2(f32x16 (&a2)[G::H2B]) const {
    if (STASH) {
      unsigned m[G::H2B];
      wide_sign_masks<G::H2B>(a2, m);
      if (mask) {
#pragma unroll
        for (int b = 0; b < G::H2B; ++b) mask[(8 + b) * 64] = m[b];
      }
    }
  }
  __device__ __forceinline__ void l2_step(int ib, f32x16 (&a1)[G::H1B]) const {
    if (STASH) stash_store_range<G::H1B>(sa1, a1, ib * 16, 16);
  }
  __device__ __forceinline__ void l3_step(int ib, f32x16 (&a2)[G::H2B]) const {
    if (STASH) stash_store_range<G::H2B>(sa2, a2, ib * 16, 16);
    if (io && ib == 0) {
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) (*x1)[b][r] = tile_load(*io, tile_soff(b, r, par_t));
    }
  }
};

// Forward stash of the TRAINING step (memory is cheap on this part: 86 KB per sample for the whole
// cfg-4 flow): per coupling and net the hidden activations a1, a2 (the dW GEMM's A operands), the
// output layer's result (S before tanh, or T) and the leaky-ReLU sign masks, so that the reverse pass
// needs no recompute at all.  Index 0 = s net, 1 = t net.  Masks: [tile][16 words][64 lanes].
struct FwdStash {
  float *a1[2], *a2[2], *out[2];
  unsigned *mask[2];
};

template <class G, bool INVERSE, bool STASH = false, bool B6 = false>
__global__ __launch_bounds__(256, 1) void k_wide_apply(WideArgs a, float *xt, float *__restrict__ ladj, int accumulate,
                                                       FwdStash fs) {
  using W = Wide<G>;
  using C = WideChunks<G, B6>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *cb = lds;
  float *bias = lds + 2 * C::CHBUF;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int par_c = 1 - a.par_t;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const long ngroups = (ntiles + 3) / 4;

  const wide_img_t img_s = make_img(B6 ? reinterpret_cast<const float *>(a.b6_s) : a.img_s, C::IMG_FLOATS),
                   img_t = make_img(B6 ? reinterpret_cast<const float *>(a.b6_t) : a.img_t, C::IMG_FLOATS);
  issue_chunk(img_s, C::O1, C::N1, cb, wave, lane);
  stage_biases<G>(bias, a.img_s, tid);
  stage_biases<G>(bias + W::NBIAS, a.img_t, tid);
  __syncthreads();
  int buf = 0;

  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long tile = grp * 4 + wave;
    const bool live = tile < ntiles;
    const long tl = live ? tile : 0;
    const long j = tl * NF_TILE + l31;
    const bool valid = live && j < a.N;
    const bool more = grp + gridDim.x < ngroups;
    const TileIO io = make_tile_io(xt, tl, a.d, l31, hi);
    f32x16 S[G::CB], T[G::CB], x1[G::CB];
#ifdef NF_KERNEL_TRACE  // tools/trace_wide_apply.py: block 0, wave 0, first tile group; s net at [0..6], t net at [16..22], end at [23]
    long long *tr = (a.trace && blockIdx.x == 0 && tid == 0 && grp == blockIdx.x) ? a.trace : nullptr;
#else
    long long *tr = nullptr;
#endif
    WIDE_STAMP(0);
    {
      f32x16 xb[G::MB];
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = tile_load(io, tile_soff(b, r, par_c));
          xb[b][r] = valid ? v : 0.f;
        }
      ApplyHooks<G, STASH> hk{make_stash_io(fs.a1[0], tl, (STASH && live) ? 32 * G::H1B : 0, l31, hi),
                              make_stash_io(fs.a2[0], tl, (STASH && live) ? 32 * G::H2B : 0, l31, hi),
                              (STASH && live) ? fs.mask[0] + tl * (16 * 64) + lane : nullptr, nullptr, nullptr, 0};
      WIDE_STAMP(1);
      wide_net_fwd<G, ApplyHooks<G, STASH>, B6, B6>(img_s, bias, cb, buf, img_t, C::O1, C::N1, xb, S, wave, lane, hk, tr);
      WIDE_STAMP(6);
      if (STASH) stash_store<G::CB>(make_stash_io(fs.out[0], tl, live ? 32 * G::CB : 0, l31, hi), S);
    }
    {
      f32x16 xb[G::MB];
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = tile_load(io, tile_soff(b, r, par_c));
          xb[b][r] = valid ? v : 0.f;
        }
      // the transformed half is fetched while the last layer of the t net runs (one batch of loads,
      // not interleaved with the stores below: the compiler must keep load/store order on one buffer)
      ApplyHooks<G, STASH> hk{make_stash_io(fs.a1[1], tl, (STASH && live) ? 32 * G::H1B : 0, l31, hi),
                              make_stash_io(fs.a2[1], tl, (STASH && live) ? 32 * G::H2B : 0, l31, hi),
                              (STASH && live) ? fs.mask[1] + tl * (16 * 64) + lane : nullptr, &io, &x1, a.par_t};
      WIDE_STAMP(17);
      wide_net_fwd<G, ApplyHooks<G, STASH>, B6>(img_t, bias + W::NBIAS, cb, buf, img_s, C::O1, more ? C::N1 : 0, xb, T, wave, lane, hk, tr ? tr + 16 : nullptr);
      WIDE_STAMP(22);
      if (STASH) stash_store<G::CB>(make_stash_io(fs.out[1], tl, live ? 32 * G::CB : 0, l31, hi), T);
    }
    float lpart[G::CB];
#pragma unroll
    for (int b = 0; b < G::CB; ++b) {
      float sv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s = nf_tanh(S[b][r]);  // padded rows: zero weights and bias => s = 0
        const float v = x1[b][r];
        const float o = INVERSE ? nf_fdiv(v - T[b][r], nf_exp(s)) : v * nf_exp(s) + T[b][r];
        if (live) tile_store(io, tile_soff(b, r, a.par_t), o);  // rows >= c fall outside the descriptor
        sv[r] = s;
      }
      lpart[b] = nf_tree_sum16(sv);
    }
    float lsum = lpart[0];
    if (G::CB == 2) lsum = lpart[0] + lpart[1];
    if (G::CB == 4) lsum = (lpart[0] + lpart[1]) + (lpart[2] + lpart[3]);
    static_assert(G::CB == 1 || G::CB == 2 || G::CB == 4, "log-det tree");
    lsum += __shfl_xor(lsum, 32);
    if (hi == 0 && valid) {
      const float base = accumulate ? ladj[j] : 0.f;
      ladj[j] = INVERSE ? base - lsum : base + lsum;
    }
    WIDE_STAMP(23);
  }
}

// ------------------------------------------------------------------------------------
// reverse pass, one net per launch (PHASE_S = false: t net, true: s net); see nf_coupling.hip
// for the two-phase algebra.  Stash tensors: [tile][feature][32 samples].
// ------------------------------------------------------------------------------------
// dX chain of the reverse pass: W3^T, W2^T, W1^T, one output block per chunk; every chunk step also
// sends a slice of the pending stash stores (d3, d2, d1) on its way.  On entry chunk (W3, rows 0..31)
// is resident in cb[buf]; on exit chunk (next_off, next_floats) is (or nothing if next_floats == 0).
// x2bar: ybar's conditioner half += W1^T d1.
template <class G>
__device__ __forceinline__ void wide_dx_chain(wide_img_t img, float *cb, int &buf, f32x16 (&d3)[G::CB],
                                              const unsigned (&m1)[G::H1B], const unsigned (&m2)[G::H2B],
                                              const StashIO &sd1, const StashIO &sd2, const StashIO &sd3,
                                              const TileIO &gio, int par_c, bool live, int next_off, int next_floats,
                                              int wave, int lane, long long *tr) {
  using W = Wide<G>;
  const int l31 = lane & 31, hi = lane >> 5;
  // ---- dX chain: W3^T, W2^T, W1^T, one output block per chunk; every chunk step also sends a
  // slice of the pending stash stores on its way
  f32x16 d2[G::H2B];
#pragma unroll
  for (int ib = 0; ib < G::H2B; ++ib) {
    DmaJob dj{img, 0, 0, cb, wave, lane};
    if (ib + 1 < G::H2B)
      dj = DmaJob{img, G::W3 + (ib + 1) * W::CF3, W::CF3, cb + (buf ^ 1) * W::CHBUF, wave, lane};
    else
      dj = DmaJob{img, G::W2, W::CF2, cb + (buf ^ 1) * W::CHBUF, wave, lane};
    {
      constexpr int PER = (G::CB * 16 + G::H2B - 1) / G::H2B;
      stash_store_range<G::CB>(sd3, d3, ib * PER, PER);
    }
    if (ib > 0) stash_store_range<G::H2B>(sd2, d2, (ib - 1) * 16, 16);
    wide_bwdx_chunk<G::CB, G::S3>(cb + buf * W::CHBUF, d3, d2[ib], l31, hi, dj);
#pragma unroll
    for (int r = 0; r < 16; ++r) d2[ib][r] *= nf_mask_slope(m2[ib], r);
    __syncthreads();
    buf ^= 1;
  }
  WIDE_STAMP(8);
  WIDE_STAMP(9);
  WIDE_STAMP(10);
  f32x16 d1[G::H1B];
#pragma unroll
  for (int ib = 0; ib < G::H1B; ++ib) {
    DmaJob dj{img, 0, 0, cb, wave, lane};
    if (ib + 1 < G::H1B)
      dj = DmaJob{img, G::W2 + (ib + 1) * W::CF2, W::CF2, cb + (buf ^ 1) * W::CHBUF, wave, lane};
    else
      dj = DmaJob{img, G::W1, W::CF1, cb + (buf ^ 1) * W::CHBUF, wave, lane};
    if (ib == 0) stash_store_range<G::H2B>(sd2, d2, (G::H2B - 1) * 16, 16);
    if (ib > 0) stash_store_range<G::H1B>(sd1, d1, (ib - 1) * 16, 16);
    wide_bwdx_chunk<G::H2B, G::S2>(cb + buf * W::CHBUF, d2, d1[ib], l31, hi, dj);
#pragma unroll
    for (int r = 0; r < 16; ++r) d1[ib][r] *= nf_mask_slope(m1[ib], r);
    __syncthreads();
    buf ^= 1;
  }
  WIDE_STAMP(11);
  WIDE_STAMP(12);
  f32x16 g2p;  // x2bar block of the previous chunk, stored behind the next chunk's MFMAs
#pragma unroll
  for (int ib = 0; ib < G::MB; ++ib) {
    DmaJob dj{img, 0, 0, cb, wave, lane};
    if (ib + 1 < G::MB)
      dj = DmaJob{img, G::W1 + (ib + 1) * W::CF1, W::CF1, cb + (buf ^ 1) * W::CHBUF, wave, lane};
    else if (next_floats)
      dj = DmaJob{img, next_off, next_floats, cb + (buf ^ 1) * W::CHBUF, wave, lane};
    if (ib == 0) stash_store_range<G::H1B>(sd1, d1, (G::H1B - 1) * 16, 16);
    if (ib > 0 && live) {
#pragma unroll
      for (int r = 0; r < 16; ++r) tile_store(gio, tile_soff(ib - 1, r, par_c), g2p[r]);
    }
    f32x16 gold;
#pragma unroll
    for (int r = 0; r < 16; ++r) gold[r] = tile_load(gio, tile_soff(ib, r, par_c));
    f32x16 g2;
    wide_bwdx_chunk<G::H1B, G::S1>(cb + buf * W::CHBUF, d1, g2, l31, hi, dj);
#pragma unroll
    for (int r = 0; r < 16; ++r) g2p[r] = gold[r] + g2[r];
    __syncthreads();
    buf ^= 1;
  }
  if (live) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tile_store(gio, tile_soff(G::MB - 1, r, par_c), g2p[r]);
  }
}

// INVD: reverse pass of the INVERSE coupling at its output (forward-KL training), as in nf_coupling.hip:
// the s net runs first (v1bar = w1bar exp(-s), sbar = -(w1bar w1 + lbar)), then the t net (tbar = -v1bar).
template <class G, bool PHASE_S, bool INVD = false>
__global__ __launch_bounds__(256, 1) void k_wide_bwd(WideArgs a, float *__restrict__ y, float *__restrict__ ybar,
                                                     const float *__restrict__ lbar, float lbar_const, WideStash st) {
  using W = Wide<G>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *cb = lds;
  float *bias = lds + 2 * W::CHBUF;
  const float *imgp = PHASE_S ? a.img_s : a.img_t;
  const wide_img_t img = make_img(imgp, G::SIZE);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int par_c = 1 - a.par_t;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const long ngroups = (ntiles + 3) / 4;

  issue_chunk(img, G::W1, W::CF1, cb, wave, lane);
  stage_biases<G>(bias, imgp, tid);
  __syncthreads();
  int buf = 0;

  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long tile = grp * 4 + wave;
    const bool live = tile < ntiles;
    const long tl = live ? tile : 0;
    const long j = tl * NF_TILE + l31;
    const bool valid = live && j < a.N;
    const bool more = grp + gridDim.x < ngroups;
    const TileIO yio = make_tile_io(y, tl, a.d, l31, hi);
    const TileIO gio = make_tile_io(ybar, tl, a.d, l31, hi);
    // a dead wave (tile >= ntiles) keeps the barriers company; its stash descriptors are empty
    const StashIO sa1 = make_stash_io(st.a1, tl, live ? 32 * G::H1B : 0, l31, hi);
    const StashIO sa2 = make_stash_io(st.a2, tl, live ? 32 * G::H2B : 0, l31, hi);
    const StashIO sd1 = make_stash_io(st.d1, tl, live ? 32 * G::H1B : 0, l31, hi);
    const StashIO sd2 = make_stash_io(st.d2, tl, live ? 32 * G::H2B : 0, l31, hi);
    const StashIO sd3 = make_stash_io(st.d3, tl, live ? 32 * G::CB : 0, l31, hi);

    long long *tr = (a.trace && blockIdx.x == 0 && tid == 0 && grp == blockIdx.x) ? a.trace + (PHASE_S ? 32 : 0) : nullptr;
    WIDE_STAMP(0);
    unsigned m1[G::H1B], m2[G::H2B];
    f32x16 d3[G::CB], y1[G::CB], g1[G::CB];
    {
      f32x16 xb[G::MB];
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = tile_load(yio, tile_soff(b, r, par_c));
          xb[b][r] = valid ? v : 0.f;
        }
      WIDE_STAMP(1);
      // masks right after each hidden layer; the stash stores of a1 / a2 and the loads of the
      // element-wise stage's operands go out a block per chunk, behind the MFMAs of layers 2 / 3
      struct BwdHooks {
        unsigned (&m1)[G::H1B];
        unsigned (&m2)[G::H2B];
        const StashIO &sa1, &sa2;
        const TileIO &yio, &gio;
        f32x16 (&y1)[G::CB];
        f32x16 (&g1)[G::CB];
        int par_t;
        __device__ __forceinline__ void after_l1(f32x16 (&a1)[G::H1B]) const { wide_sign_masks<G::H1B>(a1, m1); }
        __device__ __forceinline__ void after_l2(f32x16 (&a2)[G::H2B]) const { wide_sign_masks<G::H2B>(a2, m2); }
        __device__ __forceinline__ void l2_step(int ib, f32x16 (&a1)[G::H1B]) const {
          stash_store_range<G::H1B>(sa1, a1, ib * 16, 16);
        }
        __device__ __forceinline__ void l3_step(int ib, f32x16 (&a2)[G::H2B]) const {
          stash_store_range<G::H2B>(sa2, a2, ib * 16, 16);
          constexpr int PER = (G::CB * 16 + G::H2B - 1) / G::H2B;
#pragma unroll
          for (int q = 0; q < G::CB * 16; ++q)
            if (q >= ib * PER && q < (ib + 1) * PER) {
              y1[q >> 4][q & 15] = tile_load(yio, tile_soff(q >> 4, q & 15, par_t));
              g1[q >> 4][q & 15] = tile_load(gio, tile_soff(q >> 4, q & 15, par_t));
            }
        }
      } hk{m1, m2, sa1, sa2, yio, gio, y1, g1, a.par_t};
      wide_net_fwd<G>(img, bias, cb, buf, img, G::W3, W::CF3, xb, d3, wave, lane, hk, tr);
    }
    WIDE_STAMP(6);
    // element-wise stage (same algebra as bwd_tile in nf_coupling.hip)
    const float lb = valid ? (lbar ? lbar[j] : lbar_const) : 0.f;
#pragma unroll
    for (int b = 0; b < G::CB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = b * 32 + nf_row(r, hi);
        const bool ok = (p < a.c) && valid;
        const float yv = y1[b][r], gv = g1[b][r];
        if (INVD && !PHASE_S) {
          if (live) tile_store(yio, tile_soff(b, r, a.par_t), yv + d3[b][r]);  // v1 = w1 exp(s) + t
          d3[b][r] = ok ? -gv : 0.f;
        } else if (INVD) {
          const float s = nf_tanh(d3[b][r]);
          const float es = nf_exp(s);
          if (live) {
            tile_store(yio, tile_soff(b, r, a.par_t), yv * es);           // w1 exp(s)
            tile_store(gio, tile_soff(b, r, a.par_t), nf_fdiv(gv, es));   // v1bar
          }
          d3[b][r] = ok ? -(gv * yv + lb) * (1.f - s * s) : 0.f;
        } else if (!PHASE_S) {
          if (live) tile_store(yio, tile_soff(b, r, a.par_t), yv - d3[b][r]);  // u = x1 * exp(S)
          d3[b][r] = ok ? gv : 0.f;
        } else {
          const float s = nf_tanh(d3[b][r]);
          const float es = nf_exp(s);
          if (live) {
            tile_store(yio, tile_soff(b, r, a.par_t), nf_fdiv(yv, es));  // x1 = u * exp(-s)
            tile_store(gio, tile_soff(b, r, a.par_t), gv * es);             // x1bar
          }
          d3[b][r] = ok ? (gv * yv + lb) * (1.f - s * s) : 0.f;
        }
      }
    WIDE_STAMP(7);

    wide_dx_chain<G>(img, cb, buf, d3, m1, m2, sd1, sd2, sd3, gio, par_c, live, G::W1, more ? W::CF1 : 0, wave, lane, tr);
    WIDE_STAMP(13);
  }
}

// ------------------------------------------------------------------------------------
// reverse pass of the TRAINING step: no recompute.  The forward stash supplies the output layer's
// result and the sign masks; what is left per net is the element-wise stage and the dX chain.
// ------------------------------------------------------------------------------------
template <class G, bool PHASE_S>
__global__ __launch_bounds__(256, 1) void k_wide_bwd_stashed(WideArgs a, float *__restrict__ y, float *__restrict__ ybar,
                                                             const float *__restrict__ lbar, float lbar_const,
                                                             WideStash st, const float *__restrict__ fout,
                                                             const unsigned *__restrict__ fmask) {
  using W = Wide<G>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *cb = lds;
  const float *imgp = PHASE_S ? a.img_s : a.img_t;
  const wide_img_t img = make_img(imgp, G::SIZE);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int par_c = 1 - a.par_t;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const long ngroups = (ntiles + 3) / 4;

  issue_chunk(img, G::W3, W::CF3, cb, wave, lane);
  __syncthreads();
  int buf = 0;

  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long tile = grp * 4 + wave;
    const bool live = tile < ntiles;
    const long tl = live ? tile : 0;
    const long j = tl * NF_TILE + l31;
    const bool valid = live && j < a.N;
    const bool more = grp + gridDim.x < ngroups;
    const TileIO yio = make_tile_io(y, tl, a.d, l31, hi);
    const TileIO gio = make_tile_io(ybar, tl, a.d, l31, hi);
    const StashIO sd1 = make_stash_io(st.d1, tl, live ? 32 * G::H1B : 0, l31, hi);
    const StashIO sd2 = make_stash_io(st.d2, tl, live ? 32 * G::H2B : 0, l31, hi);
    const StashIO sd3 = make_stash_io(st.d3, tl, live ? 32 * G::CB : 0, l31, hi);
    const StashIO so = make_stash_io(const_cast<float *>(fout), tl, 32 * G::CB, l31, hi);
    long long *tr = nullptr;

    unsigned m1[G::H1B], m2[G::H2B];
    {
      const unsigned *mp = fmask + tl * (16 * 64) + lane;
#pragma unroll
      for (int b = 0; b < G::H1B; ++b) m1[b] = mp[b * 64];
#pragma unroll
      for (int b = 0; b < G::H2B; ++b) m2[b] = mp[(8 + b) * 64];
    }
    f32x16 d3[G::CB], y1[G::CB], g1[G::CB];
#pragma unroll
    for (int b = 0; b < G::CB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        d3[b][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(so.rs, so.voff, (b * 32 + (r & 3) + 8 * (r >> 2)) * (NF_TILE * 4), 0));
        y1[b][r] = tile_load(yio, tile_soff(b, r, a.par_t));
        g1[b][r] = tile_load(gio, tile_soff(b, r, a.par_t));
      }
    const float lb = valid ? (lbar ? lbar[j] : lbar_const) : 0.f;
#pragma unroll
    for (int b = 0; b < G::CB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = b * 32 + nf_row(r, hi);
        const bool ok = (p < a.c) && valid;
        const float yv = y1[b][r], gv = g1[b][r];
        if (!PHASE_S) {
          if (live) tile_store(yio, tile_soff(b, r, a.par_t), yv - d3[b][r]);  // u = x1 * exp(S)
          d3[b][r] = ok ? gv : 0.f;
        } else {
          const float s = nf_tanh(d3[b][r]);
          const float es = nf_exp(s);
          if (live) {
            tile_store(yio, tile_soff(b, r, a.par_t), nf_fdiv(yv, es));  // x1 = u * exp(-s)
            tile_store(gio, tile_soff(b, r, a.par_t), gv * es);             // x1bar
          }
          d3[b][r] = ok ? (gv * yv + lb) * (1.f - s * s) : 0.f;
        }
      }
    wide_dx_chain<G>(img, cb, buf, d3, m1, m2, sd1, sd2, sd3, gio, par_c, live, G::W3, more ? W::CF3 : 0, wave, lane, tr);
  }
}

// The same reverse pass with its dX GEMMs on the bf16 matrix cores (round 4).  The streamed image is the net's B6T image
// (rows = a layer's INPUT features, k-groups over its OUTPUT features), so a dX GEMM is the forward chunk routine run
// through the transposed net: a chunk = the two k-groups of ONE cotangent block (split once, when its chunk arrives) and
// all rows, every output block of the layer accumulates at once (128 accumulator registers; one wave per SIMD has 512).
// The fp32 form chunks over output rows instead -- one accumulator block, every cotangent block live -- and would have
// to split the whole cotangent again for every chunk.  A finished tensor's stash stores trickle out 16 per chunk of the
// NEXT GEMM, behind that chunk's DMA (wide_chunk_barrier).
template <class G>
struct WideT {
  using T = B6TGeo<G>;
  static constexpr int N3 = 48 * T::R3, N2 = 48 * T::R2, N1 = 48 * T::R1;  // chunk sizes in floats
  static constexpr int O3 = T::T3 * 4, O2 = T::T2 * 4, O1 = T::T1 * 4;     // layer offsets in floats
  static constexpr int CH = N3 > N2 ? (N3 > N1 ? N3 : N1) : (N2 > N1 ? N2 : N1);
  static constexpr size_t LDS = (size_t)2 * CH * sizeof(float);
};

template <class G>
__device__ __forceinline__ void wide_dx_chain_b6(wide_img_t img, float *cb, int &buf, f32x16 (&d3)[G::CB],
                                                 const unsigned (&m1)[G::H1B], const unsigned (&m2)[G::H2B],
                                                 const StashIO &sd1, const StashIO &sd2, const StashIO &sd3,
                                                 const TileIO &gio, int par_c, bool live, int next_floats, int wave, int lane) {
  using WT = WideT<G>;
  const int l31 = lane & 31, hi = lane >> 5;
  const DmaJob none{img, 0, 0, cb, wave, lane};
  static_assert(G::CB * 16 == 16 * G::CB, "");
  f32x16 d2[G::H2B];
#pragma unroll
  for (int b = 0; b < G::H2B; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) d2[b][r] = 0.f;
#pragma unroll
  for (int ob = 0; ob < G::CB; ++ob) {  // dX3: d2 += W3^T[:, block ob] d3[ob]
    if (ob + 1 < G::CB) issue_chunk(img, WT::O3 + (ob + 1) * WT::N3, WT::N3, cb + (buf ^ 1) * WT::CH, wave, lane);
    else issue_chunk(img, WT::O2, WT::N2, cb + (buf ^ 1) * WT::CH, wave, lane);
    stash_store_range<G::CB>(sd3, d3, ob * 16, 16);
    wide_fwd_chunk_b6<G::H2B>(cb + buf * WT::CH, d3[ob], d2, l31, hi, none);
    wide_chunk_barrier<16>();
    buf ^= 1;
  }
#pragma unroll
  for (int b = 0; b < G::H2B; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) d2[b][r] *= nf_mask_slope(m2[b], r);
  f32x16 d1[G::H1B];
#pragma unroll
  for (int b = 0; b < G::H1B; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) d1[b][r] = 0.f;
#pragma unroll
  for (int ob = 0; ob < G::H2B; ++ob) {  // dX2
    if (ob + 1 < G::H2B) issue_chunk(img, WT::O2 + (ob + 1) * WT::N2, WT::N2, cb + (buf ^ 1) * WT::CH, wave, lane);
    else issue_chunk(img, WT::O1, WT::N1, cb + (buf ^ 1) * WT::CH, wave, lane);
    stash_store_range<G::H2B>(sd2, d2, ob * 16, 16);
    wide_fwd_chunk_b6<G::H1B>(cb + buf * WT::CH, d2[ob], d1, l31, hi, none);
    wide_chunk_barrier<16>();
    buf ^= 1;
  }
#pragma unroll
  for (int b = 0; b < G::H1B; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) d1[b][r] *= nf_mask_slope(m1[b], r);
  f32x16 g2[G::MB];  // starts from the cotangent's conditioner half: x2bar = ybar2 + W1^T d1
#pragma unroll
  for (int ob = 0; ob < G::H1B; ++ob) {  // dX1
    if (ob + 1 < G::H1B) issue_chunk(img, WT::O1 + (ob + 1) * WT::N1, WT::N1, cb + (buf ^ 1) * WT::CH, wave, lane);
    else if (next_floats) issue_chunk(img, WT::O3, next_floats, cb + (buf ^ 1) * WT::CH, wave, lane);
    stash_store_range<G::H1B>(sd1, d1, ob * 16, 16);
    if (ob == 0) {
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) g2[b][r] = tile_load(gio, tile_soff(b, r, par_c));
    }
    wide_fwd_chunk_b6<G::MB>(cb + buf * WT::CH, d1[ob], g2, l31, hi, none);
    wide_chunk_barrier<16>();
    buf ^= 1;
  }
  if (live) {
#pragma unroll
    for (int b = 0; b < G::MB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) tile_store(gio, tile_soff(b, r, par_c), g2[b][r]);
  }
}

template <class G, bool PHASE_S>
__global__ __launch_bounds__(256, 1) void k_wide_bwd_stashed_b6(WideArgs a, float *__restrict__ y, float *__restrict__ ybar,
                                                                const float *__restrict__ lbar, float lbar_const,
                                                                WideStash st, const float *__restrict__ fout,
                                                                const unsigned *__restrict__ fmask) {
  using WT = WideT<G>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *cb = lds;
  const wide_img_t img = make_img(reinterpret_cast<const float *>(PHASE_S ? a.b6t_s : a.b6t_t), B6TGeo<G>::BYTES / 4);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int par_c = 1 - a.par_t;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const long ngroups = (ntiles + 3) / 4;

  issue_chunk(img, WT::O3, WT::N3, cb, wave, lane);
  __syncthreads();
  int buf = 0;

  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long tile = grp * 4 + wave;
    const bool live = tile < ntiles;
    const long tl = live ? tile : 0;
    const long j = tl * NF_TILE + l31;
    const bool valid = live && j < a.N;
    const bool more = grp + gridDim.x < ngroups;
    const TileIO yio = make_tile_io(y, tl, a.d, l31, hi);
    const TileIO gio = make_tile_io(ybar, tl, a.d, l31, hi);
    const StashIO sd1 = make_stash_io(st.d1, tl, live ? 32 * G::H1B : 0, l31, hi);
    const StashIO sd2 = make_stash_io(st.d2, tl, live ? 32 * G::H2B : 0, l31, hi);
    const StashIO sd3 = make_stash_io(st.d3, tl, live ? 32 * G::CB : 0, l31, hi);
    const StashIO so = make_stash_io(const_cast<float *>(fout), tl, 32 * G::CB, l31, hi);

    unsigned m1[G::H1B], m2[G::H2B];
    {
      const unsigned *mp = fmask + tl * (16 * 64) + lane;
#pragma unroll
      for (int b = 0; b < G::H1B; ++b) m1[b] = mp[b * 64];
#pragma unroll
      for (int b = 0; b < G::H2B; ++b) m2[b] = mp[(8 + b) * 64];
    }
    f32x16 d3[G::CB], y1[G::CB], g1[G::CB];
#pragma unroll
    for (int b = 0; b < G::CB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        d3[b][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(so.rs, so.voff, (b * 32 + (r & 3) + 8 * (r >> 2)) * (NF_TILE * 4), 0));
        y1[b][r] = tile_load(yio, tile_soff(b, r, a.par_t));
        g1[b][r] = tile_load(gio, tile_soff(b, r, a.par_t));
      }
    const float lb = valid ? (lbar ? lbar[j] : lbar_const) : 0.f;
#pragma unroll
    for (int b = 0; b < G::CB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = b * 32 + nf_row(r, hi);
        const bool ok = (p < a.c) && valid;
        const float yv = y1[b][r], gv = g1[b][r];
        if (!PHASE_S) {
          if (live) tile_store(yio, tile_soff(b, r, a.par_t), yv - d3[b][r]);  // u = x1 * exp(S)
          d3[b][r] = ok ? gv : 0.f;
        } else {
          const float s = nf_tanh(d3[b][r]);
          const float es = nf_exp(s);
          if (live) {
            tile_store(yio, tile_soff(b, r, a.par_t), nf_fdiv(yv, es));  // x1 = u * exp(-s)
            tile_store(gio, tile_soff(b, r, a.par_t), gv * es);             // x1bar
          }
          d3[b][r] = ok ? (gv * yv + lb) * (1.f - s * s) : 0.f;
        }
      }
    wide_dx_chain_b6<G>(img, cb, buf, d3, m1, m2, sd1, sd2, sd3, gio, par_c, live, more ? WT::N3 : 0, wave, lane);
  }
}

// ------------------------------------------------------------------------------------
// weight gradients from the stash: split-K GEMM  dW^T[i][o] = sum_samples A[i][j] D[o][j]
// ------------------------------------------------------------------------------------
// A workgroup (4 waves) owns one JOB = a block of the weight image and one slice of the sample
// tiles (ks of KS).  Per sample tile it brings 384 stash rows ([feature][32 samples]) into LDS with
// row stride 33 -- the transposition the MFMA operands need (lane <-> feature, k <-> sample) -- and
// every wave accumulates a 4 x 2 arrangement of 32x32 blocks.  Wave grid WI x WO (WI*WO = 4): the
// job covers 128*WI rows of A and 64*WO rows of D.
struct DwJob {
  const float *A;     // A-operand tensor
  const float *D;     // delta tensor
  long a_tile_stride; // floats between tiles of A
  int a_extent;       // bytes of one A tile (buffer bound: rows past it read 0)
  int a_rstride;      // row r lives at ((a_row0 + r) * a_rstride + a_roff) * 128 bytes
  int a_roff;
  int a_row0;
  long d_tile_stride;
  int d_extent;
  int d_row0;
  int cfg;            // 0: WI=1,WO=4   1: WI=2,WO=2
  int w_off, w_stride;// image offset / row stride of this layer's weight block
  int b_off;          // image offset of the bias block, or -1 if another job owns it
  int ib_tot, ob_tot; // layer size in blocks (writes beyond are dropped)
};
#define NF_WIDE_MAXJOBS 12
struct DwArgs {
  DwJob job[NF_WIDE_MAXJOBS];
  int njobs, ksplit;
  long ntiles;
  long slab_stride;  // floats between ksplit partials
  long long *trace;  // optional clock stamps of block 0 / wave 0 (nf_debug_trace, tools/trace_wide_dw.py)
};

#define DW_TS 33
#define DW_ROWS 384

// A tile's base address is wave-uniform (tile and job are), but hipcc does the 64-bit multiply on the vector ALU and then
// wraps every buffer load in a waterfall loop over "the lanes' descriptors" (12 loops of 18 instructions per tile): hand it
// the address back through SGPRs.
__device__ __forceinline__ float *wave_uniform_ptr(const float *p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<float *>(((unsigned long long)hi << 32) | lo);
}

template <int WI, int WO>
__device__ __forceinline__ void dw_job(const DwJob &jb, const DwArgs &a, int ks, float *__restrict__ out, float *lds) {
  constexpr int AROWS = 128 * WI;
  constexpr int NLD = DW_ROWS * 8 / 256;  // float4 loads per thread per tile = 12
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wi = wave / WO, wo = wave % WO;
  const int lrow = tid >> 3, part = tid & 7;

  f32x16 acc[4][2];
  float bsum[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][o][r] = 0.f;

  float4 stg[NLD];
  auto load_tile = [&](long tile) {
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(wave_uniform_ptr(jb.A + tile * jb.a_tile_stride), 0, jb.a_extent, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(wave_uniform_ptr(jb.D + tile * jb.d_tile_stride), 0, jb.d_extent, 0x00020000);
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int row = lrow + 32 * k;
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      u32x4 v;
      if (32 * k < AROWS)
        v = __builtin_amdgcn_raw_buffer_load_b128(ra, ((jb.a_row0 + row) * jb.a_rstride + jb.a_roff) * 128 + part * 16, 0, 0);
      else
        v = __builtin_amdgcn_raw_buffer_load_b128(rd, (jb.d_row0 + row - AROWS) * 128 + part * 16, 0, 0);
      const unsigned v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;  // (bit_cast of a vector-element expression reads element 0)
      stg[k] = make_float4(__builtin_bit_cast(float, v0), __builtin_bit_cast(float, v1), __builtin_bit_cast(float, v2),
                           __builtin_bit_cast(float, v3));
    }
  };
  auto put_tile = [&](float *dst) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      float *p = dst + (lrow + 32 * k) * DW_TS + part * 4;
      p[0] = stg[k].x; p[1] = stg[k].y; p[2] = stg[k].z; p[3] = stg[k].w;
    }
  };

  long tile = ks;
  if (tile < a.ntiles) {
    load_tile(tile);
    put_tile(lds);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < a.ntiles; tile += a.ksplit) {
    const bool has_next = tile + a.ksplit < a.ntiles;
    if (has_next) load_tile(tile + a.ksplit);
    const float *cur = lds + buf * (DW_ROWS * DW_TS);
    const float *pa = cur + (wi * 128 + l31) * DW_TS + hi;
    const float *pd = cur + (AROWS + wo * 64 + l31) * DW_TS + hi;
    float an[2][4], dn[2][2], ac[2][4], dc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) an[u][ib] = pa[ib * 32 * DW_TS + 2 * u];
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) dn[u][ob] = pd[ob * 32 * DW_TS + 2 * u];
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) ac[u][ib] = an[u][ib];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) dc[u][ob] = dn[u][ob];
      }
      if (g + 1 < 8) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int ib = 0; ib < 4; ++ib) an[u][ib] = pa[ib * 32 * DW_TS + 2 * ((g + 1) * 2 + u)];
#pragma unroll
          for (int ob = 0; ob < 2; ++ob) dn[u][ob] = pd[ob * 32 * DW_TS + 2 * ((g + 1) * 2 + u)];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) bsum[ob] += dc[u][ob];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
          for (int ob = 0; ob < 2; ++ob)
            acc[ib][ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u][ib], dc[u][ob], acc[ib][ob], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (has_next) put_tile(lds + (buf ^ 1) * (DW_ROWS * DW_TS));
    __syncthreads();
    buf ^= 1;
  }

  // partial result in image layout
#pragma unroll
  for (int ib = 0; ib < 4; ++ib) {
    const int iblk = jb.a_row0 / 32 + wi * 4 + ib;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      const int oblk = jb.d_row0 / 32 + wo * 2 + ob;
      if (iblk < jb.ib_tot && oblk < jb.ob_tot) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          out[jb.w_off + (iblk * 32 + nf_row(r, hi)) * jb.w_stride + oblk * 32 + l31] = acc[ib][ob][r];
      }
    }
  }
  if (jb.b_off >= 0 && wi == 0) {
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      const int oblk = jb.d_row0 / 32 + wo * 2 + ob;
      const float v = bsum[ob] + __shfl_xor(bsum[ob], 32);
      if (hi == 0 && oblk < jb.ob_tot) out[jb.b_off + oblk * 32 + l31] = v;
    }
  }
}

// The same job on the bf16 matrix cores (round 4).  Both operands come from the stash in fp32; they are split ONCE per tile,
// by the thread that brought them in, on their way into LDS (a row's 32 samples as bf16 triples in the D6 layout of
// nf_mfma.h: [component][sample group][parity][8] x 2 bytes, sample = 2 (8 g + j) + parity -- a thread's four consecutive
// samples are two (even, odd) pairs of neighbouring k-slots, i.e. two packed dwords per component), so the GEMM loop is
// 16-byte LDS reads and MFMAs only: 96 bf16 MFMAs of 32 clocks per wave and tile instead of 128 fp32 ones of 64.  Two LDS
// buffers (384 rows x 208 bytes = 78 KB each, 156 of the CU's 160 KB): the next tile is fetched into registers behind the
// MFMAs, split and stored into the other buffer, one barrier per tile.
// (the split itself: nf_split2, nf_mfma.h)
template <int WI, int WO>
__device__ __forceinline__ void dw_job_b6(const DwJob &jb, const DwArgs &a, int ks, float *__restrict__ out, char *lds) {
  constexpr int AROWS = 128 * WI;
  constexpr int NLD = DW_ROWS * 8 / 256;  // 16-byte loads per thread per tile = 12
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wi = wave / WO, wo = wave % WO;
  const int lrow = tid >> 2, part = tid & 3;  // a thread brings in EIGHT consecutive samples of a row (two 16-byte loads)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

  f32x16 acc[4][2];
  float bsum[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][o][r] = 0.f;

  // Software pipeline (round 5).  The stamps of the one-stage form (tools/trace_wide_dw.py: per tile 0.9-3.0 k clocks to ISSUE the
  // twelve 16-byte requests, 3.9 k for the 96 MFMAs, 0.3 k waiting, 1.9 k to split and store the next tile, 0.1 k at the barrier --
  // 7.2-9.6 k, all of it one after the other in each wave's instruction stream) say where the time is: not in HBM latency but in
  // three streams that never run at the same time.  Here they are woven together by hand, one small piece of each of the other
  // two behind every four MFMAs (one term of the six for the four A blocks):
  //   * the split + LDS stores of tile i + 1, which arrived during the previous iteration: one nf_split2 per quad, the six
  //     ds_write_b64 of a row pair behind its fourth;
  //   * the requests for tile i + 2, one per two quads from the fifth quad on (the address unit takes them at its own pace
  //     under the MFMAs) -- each into the staging register whose tile i + 1 content the split has just consumed, so ONE set
  //     of twelve staging registers carries both tiles (a second set pushed the kernel to 481 registers and the compiler to
  //     ~370 AGPR <-> VGPR moves per tile);
  //   * the MFMAs of tile i.
  // A tile past the end is requested through descriptors of extent 0 (no memory access, zeros back) and its "split" is stored
  // into the idle buffer: the request count of an iteration must not depend on a branch, or hipcc's wait-count pass falls
  // back to s_waitcnt vmcnt(0) and waits for the stream.
  const int poff = lrow * D6_ROW + (part >> 1) * 32 + (part & 1) * 8;
  constexpr int RB = 32 * D6_ROW / 16;  // 16-byte units per block of 32 rows
  const unsigned ones = 0x3F803F80u;
  const bool need_bias = jb.b_off >= 0 && wi == 0;
  auto descriptors = [&](long tile, __amdgpu_buffer_rsrc_t &ra, __amdgpu_buffer_rsrc_t &rd) {
    const bool in = tile < a.ntiles;
    const long tl = in ? tile : 0;
    ra = __builtin_amdgcn_make_buffer_rsrc(wave_uniform_ptr(jb.A + tl * jb.a_tile_stride), 0, in ? jb.a_extent : 0, 0x00020000);
    rd = __builtin_amdgcn_make_buffer_rsrc(wave_uniform_ptr(jb.D + tl * jb.d_tile_stride), 0, in ? jb.d_extent : 0, 0x00020000);
  };
  // byte offset of load k within its tile: computed once (in the loop hipcc re-read the job's fields with s_load and waited
  // on lgkmcnt(0) -- the LDS counter -- for each of them)
  int voff[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int row = lrow + 64 * (k >> 1), seg = part * 32 + (k & 1) * 16;
    voff[k] = 64 * (k >> 1) < AROWS ? ((jb.a_row0 + row) * jb.a_rstride + jb.a_roff) * 128 + seg : (jb.d_row0 + row - AROWS) * 128 + seg;
  }
  auto request = [&](const __amdgpu_buffer_rsrc_t &ra, const __amdgpu_buffer_rsrc_t &rd, int k) -> u32x4 {  // load k of a tile (k compile-time)
    return __builtin_amdgcn_raw_buffer_load_b128(64 * (k >> 1) < AROWS ? ra : rd, voff[k], 0, 0);
  };
  // samples 8 part .. 8 part + 7 of a row: the four even ones are slots j .. j + 3 of (group part >> 1, parity 0) -- two packed
  // dwords side by side, one ds_write_b64 per component --, the four odd ones the same slots of parity 1.
  // piece j of row pair kp (loads 2 kp, 2 kp + 1): j = 0 .. 3 one split each, the stores behind the last
  unsigned sp_[12];  // eh0 em0 el0 | eh1 em1 el1 | oh0 om0 ol0 | oh1 om1 ol1 of the row pair in progress
  u32x4 stg[NLD];
  auto put_piece = [&](char *dst, int kp, int j) {
    const u32x4 &va = stg[2 * kp], &vb = stg[2 * kp + 1];
    const unsigned a0 = va.x, a1 = va.y, a2 = va.z, a3 = va.w, b0 = vb.x, b1 = vb.y, b2 = vb.z, b3 = vb.w;
    if (j == 0) nf_split2(__uint_as_float(a0), __uint_as_float(a2), sp_[0], sp_[1], sp_[2]);
    if (j == 1) nf_split2(__uint_as_float(b0), __uint_as_float(b2), sp_[3], sp_[4], sp_[5]);
    if (j == 2) nf_split2(__uint_as_float(a1), __uint_as_float(a3), sp_[6], sp_[7], sp_[8]);
    if (j == 3) {
      nf_split2(__uint_as_float(b1), __uint_as_float(b3), sp_[9], sp_[10], sp_[11]);
      u32x2 *p = reinterpret_cast<u32x2 *>(dst + poff + 64 * kp * D6_ROW);
      p[0] = u32x2{sp_[0], sp_[3]}; p[2] = u32x2{sp_[6], sp_[9]};      // component h: parity 0 at +0, parity 1 at +16 bytes
      p[8] = u32x2{sp_[1], sp_[4]}; p[10] = u32x2{sp_[7], sp_[10]};    // component m at +64 bytes
      p[16] = u32x2{sp_[2], sp_[5]}; p[18] = u32x2{sp_[8], sp_[11]};   // component l at +128 bytes
    }
  };
  long long *tr = (a.trace && blockIdx.x == 0 && tid == 0) ? a.trace : nullptr;
  int tslot = 0;  // seven stamps per tile for the first eight tiles (slots 1, 4, 5 unused in this form)
  long tile = ks;
  {
    __amdgpu_buffer_rsrc_t ra, rd;
    descriptors(tile, ra, rd);
#pragma unroll
    for (int k = 0; k < NLD; ++k) stg[k] = request(ra, rd, k);
#pragma unroll
    for (int kp = 0; kp < NLD / 2; ++kp)
#pragma unroll
      for (int j = 0; j < 4; ++j) put_piece(lds, kp, j);
    descriptors(tile + a.ksplit, ra, rd);
#pragma unroll
    for (int k = 0; k < NLD; ++k) stg[k] = request(ra, rd, k);
  }
  __syncthreads();
  int buf = 0;
  // one tile per iteration: MFMAs from buffer `buf`, tile + ksplit (in stg) split into the other buffer, tile + 2 ksplit requested
  for (; tile < a.ntiles; tile += a.ksplit) {
    const char *cur = lds + buf * (DW_ROWS * D6_ROW);
    char *nxt = lds + (buf ^ 1) * (DW_ROWS * D6_ROW);
    if (tslot >= 56) tr = nullptr;
    WIDE_STAMP(tslot + 0);
    __amdgpu_buffer_rsrc_t ra, rd;
    descriptors(tile + 2L * a.ksplit, ra, rd);
    const nf_u32x4 *pa = reinterpret_cast<const nf_u32x4 *>(cur + (wi * 128 + l31) * D6_ROW + hi * 16);
    const nf_u32x4 *pd = reinterpret_cast<const nf_u32x4 *>(cur + (AROWS + wo * 64 + l31) * D6_ROW + hi * 16);
    nf_u32x4 An[4][3], Dn[2][3];  // the operands of the sample group after this one: requested behind its 48 MFMAs
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int c = 0; c < 3; ++c) An[ib][c] = pa[ib * RB + c * 4];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int c = 0; c < 3; ++c) Dn[ob][c] = pd[ob * RB + c * 4];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      nf_u32x4 A[4][3], D[2][3];
#pragma unroll
      for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int c = 0; c < 3; ++c) A[ib][c] = An[ib][c];
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int c = 0; c < 3; ++c) D[ob][c] = Dn[ob][c];
      if (g == 0) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
          for (int c = 0; c < 3; ++c) An[ib][c] = pa[ib * RB + c * 4 + 2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
          for (int c = 0; c < 3; ++c) Dn[ob][c] = pd[ob * RB + c * 4 + 2];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
        if (need_bias)  // (wave-uniform: the job owns the bias block and this wave's rows are its first)
#pragma unroll
          for (int i = 0; i < 12; ++i) bsum[ob] = nf_dot2_bf16(D[ob][2 - i / 4][i % 4], ones, bsum[ob]);
#pragma unroll
        for (int term = 0; term < 6; ++term) {  // smallest first: al dh, ah dl, am dm, am dh, ah dm, ah dh
#pragma unroll
          for (int ib = 0; ib < 4; ++ib) {
            const nf_u32x4 &av = term == 0 ? A[ib][2] : (term == 2 || term == 3) ? A[ib][1] : A[ib][0];
            const nf_u32x4 &dv = term == 1 ? D[ob][2] : (term == 2 || term == 4) ? D[ob][1] : D[ob][0];
            acc[ib][ob] = nf_mfma_bf16(av, dv, acc[ib][ob]);
          }
          const int q = (g * 2 + ob) * 6 + term;  // 0 .. 23
          put_piece(nxt, q / 4, q % 4);
          // request k goes out behind quad 2 k + 4 (row pair k / 2 was consumed by quad 2 k + 3 at the latest); the last two
          // behind the last quad
          if (q >= 4 && q % 2 == 0 && q < 23) stg[(q - 4) / 2] = request(ra, rd, (q - 4) / 2);
          if (q == 23) {
            stg[10] = request(ra, rd, 10);
            stg[11] = request(ra, rd, 11);
          }
          // the piece's vector instructions in the shadows of the quad's MFMAs (an MFMA holds the matrix pipe for 32 clocks,
          // the next one cannot issue before: room for seven VALU instructions), not behind the quad
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);  // up to seven VALU
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      WIDE_STAMP(tslot + 2 + g);
    }
    __syncthreads();
    WIDE_STAMP(tslot + 6);
    tslot += 7;
    buf ^= 1;
  }

  // partial result in image layout
#pragma unroll
  for (int ib = 0; ib < 4; ++ib) {
    const int iblk = jb.a_row0 / 32 + wi * 4 + ib;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      const int oblk = jb.d_row0 / 32 + wo * 2 + ob;
      if (iblk < jb.ib_tot && oblk < jb.ob_tot) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          out[jb.w_off + (iblk * 32 + nf_row(r, hi)) * jb.w_stride + oblk * 32 + l31] = acc[ib][ob][r];
      }
    }
  }
  if (jb.b_off >= 0 && wi == 0) {
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      const int oblk = jb.d_row0 / 32 + wo * 2 + ob;
      const float v = bsum[ob] + __shfl_xor(bsum[ob], 32);
      if (hi == 0 && oblk < jb.ob_tot) out[jb.b_off + oblk * 32 + l31] = v;
    }
  }
}

__global__ __launch_bounds__(256, 1) void k_wide_dw_b6(DwArgs a, float *__restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char lds_b[];
  const int j = blockIdx.x % a.njobs, ks = blockIdx.x / a.njobs;
  const DwJob &jb = a.job[j];
  float *out = slab + (long)ks * a.slab_stride;
  if (jb.cfg == 0)
    dw_job_b6<1, 4>(jb, a, ks, out, lds_b);
  else
    dw_job_b6<2, 2>(jb, a, ks, out, lds_b);
}

__global__ __launch_bounds__(256, 1) void k_wide_dw(DwArgs a, float *__restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int j = blockIdx.x % a.njobs, ks = blockIdx.x / a.njobs;
  const DwJob &jb = a.job[j];
  float *out = slab + (long)ks * a.slab_stride;
  if (jb.cfg == 0)
    dw_job<1, 4>(jb, a, ks, out, lds);
  else
    dw_job<2, 2>(jb, a, ks, out, lds);
}

// g[theta index] = sum over split-K partials of ONE net's image
template <class G>
__global__ __launch_bounds__(256) void k_wide_reduce(NetDims nd, const float *__restrict__ slab, int nslab, long slab_stride,
                                                     float *__restrict__ g) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= G::B3 + 32 * G::CB) return;
  const long ti = image_theta_index<G>(nd, e);
  if (ti < 0) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 3 < nslab; s += 4) {
    a0 += slab[(long)s * slab_stride + e];
    a1 += slab[(long)(s + 1) * slab_stride + e];
    a2 += slab[(long)(s + 2) * slab_stride + e];
    a3 += slab[(long)(s + 3) * slab_stride + e];
  }
  for (; s < nslab; ++s) a0 += slab[(long)s * slab_stride + e];
  g[ti] = (a0 + a1) + (a2 + a3);
}

// The training step's form: every (coupling, net) of the reverse pass leaves its split-K partials in its own slab
// region, and ONE launch at the end sums them all (blockIdx.y = 2 * coupling + phase; phase 0 = t net, 1 = s net).
template <class G>
__global__ __launch_bounds__(256) void k_wide_reduce_all(PackArgs p, const float *__restrict__ slab, int nslab, long slab_stride,
                                                         float *__restrict__ g, int job0) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= G::B3 + 32 * G::CB) return;
  const int job = blockIdx.y + job0, k = job >> 1, phase = job & 1;  // job0: first job of a gradient bucket (0: all at once)
  const int c = (k & 1) ? p.d / 2 : (p.d + 1) / 2, m = p.d - c;
  long off = (long)(k >> 1) * p.pair_params + ((k & 1) ? p.odd_params : 0);
  if (phase == 0) off += net_param_count(m, p.h1, p.h2, c);
  const NetDims nd = make_net_dims(off, m, p.h1, p.h2, c);
  const long ti = image_theta_index<G>(nd, e);
  if (ti < 0) return;
  slab += (size_t)job * nslab * slab_stride;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 3 < nslab; s += 4) {  // the order of k_wide_reduce
    a0 += slab[(long)s * slab_stride + e];
    a1 += slab[(long)(s + 1) * slab_stride + e];
    a2 += slab[(long)(s + 2) * slab_stride + e];
    a3 += slab[(long)(s + 3) * slab_stride + e];
  }
  for (; s < nslab; ++s) a0 += slab[(long)s * slab_stride + e];
  g[ti] = (a0 + a1) + (a2 + a3);
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
static inline int wblocks32(int n) { return (n + 31) / 32; }

bool nf_wide_supported(const nf_flow_desc *desc) {
  if (desc->kind != NF_KIND_REALNVP || desc->n_hidden != 2 || desc->dtype != NF_DTYPE_F32) return false;
  const int c = (desc->d + 1) / 2;
  return wblocks32(c) <= GW::MB && wblocks32(desc->hdims[0]) <= GW::H1B && wblocks32(desc->hdims[1]) <= GW::H2B;
}

// Host code per padded geometry G; the entry points below pick the geometry from the flow shape.
template <class G>
struct WideHost {
// packed images of every net: [coupling][s|t][G::SIZE] in ctx->wimg, plus DMA slack at the end
// and behind them, 256-byte aligned, their B6 copies (bf16 triples; rebuilt from the fp32 images when those changed)
static size_t wide_b6_offset(const nf_flow_desc *desc) {
  return (((size_t)2 * desc->nlayers * 2 * G::SIZE * sizeof(float) + 4096) + 255) / 256 * 256;
}
static size_t wide_b6t_offset(const nf_flow_desc *desc) {
  return ((wide_b6_offset(desc) + (size_t)2 * desc->nlayers * 2 * B6Geo<G>::BYTES + 4096) + 255) / 256 * 256;
}
static size_t wide_wimg_bytes(nf_ctx *, const nf_flow_desc *desc) {
  return wide_b6t_offset(desc) + (size_t)2 * desc->nlayers * 2 * B6TGeo<G>::BYTES + 4096;
}
static int wide_b6t_refresh(nf_ctx *ctx, const nf_flow_desc *desc) {
  if (ctx->b6t_gen == ctx->wimg_gen) return NF_OK;
  using T = B6TGeo<G>;
  const int nimg = 2 * desc->nlayers * 2;
  constexpr long PER = 2 * G::CB * 2 * T::R3 + 2 * G::H2B * 2 * T::R2 + 2 * G::H1B * 2 * T::R1;
  const long total = (long)nimg * PER;
  ProfScope ps(ctx, "pack_weights");
  hipLaunchKernelGGL((k_b6t_from_images<G>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, nimg, (const float *)ctx->wimg,
                     (unsigned char *)ctx->wimg + wide_b6t_offset(desc));
  NF_HIP(hipGetLastError());
  ctx->b6t_gen = ctx->wimg_gen;
  return NF_OK;
}
// NF_WIDE_FP32=1: the forward kernels on fp32 MFMAs (A/B switch)
static bool wide_b6() {
  static const bool fp32 = std::getenv("NF_WIDE_FP32") != nullptr;
  return !fp32;
}
static int wide_b6_refresh(nf_ctx *ctx, const nf_flow_desc *desc) {
  if (ctx->b6_gen == ctx->wimg_gen) return NF_OK;
  using B = B6Geo<G>;
  const int nimg = 2 * desc->nlayers * 2;
  constexpr long PER = 2 * G::MB * 2 * B::R1 + 2 * G::H1B * 2 * B::R2 + 2 * G::H2B * 2 * B::R3 + B::R1 + B::R2 + B::R3;
  const long total = (long)nimg * PER;
  ProfScope ps(ctx, "pack_weights");
  hipLaunchKernelGGL((k_b6_from_images<G>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, nimg, (const float *)ctx->wimg,
                     (unsigned char *)ctx->wimg + wide_b6_offset(desc));
  NF_HIP(hipGetLastError());
  ctx->b6_gen = ctx->wimg_gen;
  return NF_OK;
}

static int wide_pack(nf_ctx *ctx, const nf_flow_desc *desc, const float *theta) {
  if (!nf_wide_supported(desc)) return NF_ERR_UNSUPPORTED;
  const int nc = 2 * desc->nlayers;
  NF_TRY(nf_wimg_reserve(ctx, wide_wimg_bytes(ctx, desc)));
  const PackArgs p = make_pack_args(desc);
  const long total = (long)nc * 2 * G::SIZE;
  ProfScope ps(ctx, "pack_weights");
  hipLaunchKernelGGL((k_pack_net_images<G>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, p, theta,
                     (float *)ctx->wimg);
  return (int)hipGetLastError();
}

static WideArgs make_wide_args(nf_ctx *ctx, const nf_flow_desc *desc, int k, long N) {
  const CouplingInfo ci = nf_coupling_info(desc, k);
  WideArgs a;
  a.img_s = (const float *)ctx->wimg + (size_t)(2 * k) * G::SIZE;
  a.img_t = a.img_s + G::SIZE;
  a.b6_s = (const unsigned char *)ctx->wimg + wide_b6_offset(desc) + (size_t)(2 * k) * B6Geo<G>::BYTES;
  a.b6_t = a.b6_s + B6Geo<G>::BYTES;
  a.b6t_s = (const unsigned char *)ctx->wimg + wide_b6t_offset(desc) + (size_t)(2 * k) * B6TGeo<G>::BYTES;
  a.b6t_t = a.b6t_s + B6TGeo<G>::BYTES;
  a.trace = (long long *)ctx->trace;
  a.d = desc->d; a.c = ci.c; a.m = ci.m; a.par_t = ci.par_t; a.N = N;
  return a;
}

static long wide_groups(long N) { return ((N + NF_TILE - 1) / NF_TILE + 3) / 4; }

static int wide_apply(nf_ctx *ctx, const nf_flow_desc *desc, int k, bool inverse, float *xt, long N, float *ladj,
                  int accumulate) {
  if (!ctx->wimg) return NF_ERR_UNSUPPORTED;
  const WideArgs a = make_wide_args(ctx, desc, k, N);
  const bool b6 = wide_b6();
  if (b6) NF_TRY(wide_b6_refresh(ctx, desc));
  const size_t lds = b6 ? Wide<G>::LDS_APPLY_B6 : Wide<G>::LDS_APPLY;
  static AttrOnce attr_once;  // once per device: a context on another GPU needs its own
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_apply<G, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wide<G>::LDS_APPLY));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_apply<G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wide<G>::LDS_APPLY));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_apply<G, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wide<G>::LDS_APPLY_B6));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_apply<G, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wide<G>::LDS_APPLY_B6));
    return NF_OK;
  }));
  long grid = wide_groups(N);
  if (grid > ctx->num_cu) grid = ctx->num_cu;
  if (grid < 1) grid = 1;
  ProfScope ps(ctx, "wide_apply");
  const FwdStash none{};
  if (b6 && inverse)
    hipLaunchKernelGGL((k_wide_apply<G, true, false, true>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, xt, ladj, accumulate, none);
  else if (b6)
    hipLaunchKernelGGL((k_wide_apply<G, false, false, true>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, xt, ladj, accumulate, none);
  else if (inverse)
    hipLaunchKernelGGL((k_wide_apply<G, true>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, xt, ladj, accumulate, none);
  else
    hipLaunchKernelGGL((k_wide_apply<G, false>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, xt, ladj, accumulate, none);
  return (int)hipGetLastError();
}

// split-K factor of the weight-gradient GEMM and its job list (one net)
static int wide_ksplit(nf_ctx *ctx, long ntiles, int njobs) {
  long ks = (ctx->num_cu + njobs - 1) / njobs;  // one workgroup per CU
  if (ks > ntiles) ks = ntiles;
  if (ks < 1) ks = 1;
  return (int)ks;
}

static int build_jobs(DwArgs *args, const float *x2src, long x2_tile_stride, int x2_extent, int x2_roff,
                      const WideStash &st) {
  int n = 0;
  auto add_layer = [&](const float *A, long ats, int aext, int arstride, int aroff, int IB, const float *D, int OB,
                       int w_off, int w_stride, int b_off) {
    int cfg, imac, omac;  // macro tile in blocks
    if (OB >= 8) { cfg = 0; imac = 4; omac = 8; }
    else { cfg = 1; imac = 8; omac = 4; }
    for (int i0 = 0; i0 < IB; i0 += imac)
      for (int o0 = 0; o0 < OB; o0 += omac) {
        DwJob &j = args->job[n++];
        j.A = A; j.D = D;
        j.a_tile_stride = ats; j.a_extent = aext; j.a_rstride = arstride; j.a_roff = aroff; j.a_row0 = 32 * i0;
        j.d_tile_stride = (long)OB * 32 * NF_TILE; j.d_extent = OB * 32 * NF_TILE * 4; j.d_row0 = 32 * o0;
        j.cfg = cfg; j.w_off = w_off; j.w_stride = w_stride; j.b_off = (i0 == 0) ? b_off : -1;
        j.ib_tot = IB; j.ob_tot = OB;
      }
  };
  add_layer(x2src, x2_tile_stride, x2_extent, 2, x2_roff, G::MB, st.d1, G::H1B, G::W1, G::S1, G::B1);
  add_layer(st.a1, (long)G::H1B * 32 * NF_TILE, G::H1B * 32 * NF_TILE * 4, 1, 0, G::H1B, st.d2, G::H2B, G::W2, G::S2, G::B2);
  add_layer(st.a2, (long)G::H2B * 32 * NF_TILE, G::H2B * 32 * NF_TILE * 4, 1, 0, G::H2B, st.d3, G::CB, G::W3, G::S3, G::B3);
  args->njobs = n;
  return n;
}

static int wide_njobs() {
  DwArgs tmp;
  WideStash st = {nullptr, nullptr, nullptr, nullptr, nullptr};
  return build_jobs(&tmp, nullptr, 0, 0, 0, st);
}

// floats of device workspace the reverse pass needs for a batch of N
static size_t wide_bwd_ws_floats(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  (void)desc;
  const long ntiles = (N + NF_TILE - 1) / NF_TILE;
  const size_t stash = (size_t)ntiles * NF_TILE * 32 * (2 * G::H1B + 2 * G::H2B + G::CB);
  const int ks = wide_ksplit(ctx, ntiles, wide_njobs());
  return stash + (size_t)ks * G::SIZE + 1024;
}

// reverse pass over the whole chain; state/gbar as in realnvp_bwd (nf_api.hip)
// inv_dir: reverse pass of the INVERSE chain (state: T^-1(data) -> data; couplings in execution order, s net first)
static int wide_bwd(nf_ctx *ctx, const nf_flow_desc *desc, float *state, float *gbar, const float *lbar, float lbar_const,
                long N, float *ws, float *g_out, bool inv_dir) {
  if (!ctx->wimg) return NF_ERR_UNSUPPORTED;
  const long ntiles = (N + NF_TILE - 1) / NF_TILE;
  const size_t per = (size_t)ntiles * NF_TILE * 32;
  WideStash st;
  float *p = ws;
  st.a1 = p; p += per * G::H1B;
  st.a2 = p; p += per * G::H2B;
  st.d1 = p; p += per * G::H1B;
  st.d2 = p; p += per * G::H2B;
  st.d3 = p; p += per * G::CB;
  float *slab = p;
  const int njobs = wide_njobs();
  const int ks = wide_ksplit(ctx, ntiles, njobs);

  static AttrOnce attr_once;  // once per device
  const size_t lds_bwd = Wide<G>::LDS_BWD;
  const size_t lds_dw = (size_t)2 * DW_ROWS * DW_TS * sizeof(float);
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_bwd<G, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bwd));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_bwd<G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bwd));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_bwd<G, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bwd));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_bwd<G, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bwd));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_dw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dw));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_dw_b6, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DW_ROWS * D6_ROW));
    return NF_OK;
  }));
  long grid = wide_groups(N);
  if (grid > ctx->num_cu) grid = ctx->num_cu;
  if (grid < 1) grid = 1;
  const int nc = 2 * desc->nlayers;
  const int h1 = desc->hdims[0], h2 = desc->hdims[1];
  for (int step = 0; step < nc; ++step) {
    const int k = inv_dir ? nc - 1 - step : step;  // forward chain: flat order = reverse of execution order
    const WideArgs a = make_wide_args(ctx, desc, k, N);
    const CouplingInfo ci = nf_coupling_info(desc, k);
    for (int phase = 0; phase < 2; ++phase) {  // forward chain: t net, then s net; inverse chain: s, then t
      const bool is_s = inv_dir ? phase == 0 : phase == 1;
      {
        ProfScope ps(ctx, "wide_bwd");
        if (inv_dir && is_s)
          hipLaunchKernelGGL((k_wide_bwd<G, true, true>), dim3((unsigned)grid), dim3(256), lds_bwd, ctx->stream, a, state, gbar, lbar, lbar_const, st);
        else if (inv_dir)
          hipLaunchKernelGGL((k_wide_bwd<G, false, true>), dim3((unsigned)grid), dim3(256), lds_bwd, ctx->stream, a, state, gbar, lbar, lbar_const, st);
        else if (!is_s)
          hipLaunchKernelGGL((k_wide_bwd<G, false>), dim3((unsigned)grid), dim3(256), lds_bwd, ctx->stream, a, state, gbar, lbar, lbar_const, st);
        else
          hipLaunchKernelGGL((k_wide_bwd<G, true>), dim3((unsigned)grid), dim3(256), lds_bwd, ctx->stream, a, state, gbar, lbar, lbar_const, st);
        NF_HIP(hipGetLastError());
      }
      DwArgs da;
      build_jobs(&da, state, (long)desc->d * NF_TILE, desc->d * NF_TILE * 4, 1 - ci.par_t, st);
      da.ksplit = ks;
      da.ntiles = ntiles;
      da.slab_stride = G::SIZE;
      da.trace = (long long *)ctx->trace;
      {
        ProfScope ps(ctx, "wide_dw");
        if (wide_b6())
          hipLaunchKernelGGL(k_wide_dw_b6, dim3((unsigned)(da.njobs * ks)), dim3(256), 2 * DW_ROWS * D6_ROW, ctx->stream, da, slab);
        else
          hipLaunchKernelGGL(k_wide_dw, dim3((unsigned)(da.njobs * ks)), dim3(256), lds_dw, ctx->stream, da, slab);
        NF_HIP(hipGetLastError());
      }
      long off = ci.theta_off;
      if (!is_s) off += net_param_count(ci.m, h1, h2, ci.c);  // t net follows the s net in theta
      const NetDims nd = make_net_dims(off, ci.m, h1, h2, ci.c);
      {
        ProfScope ps(ctx, "reduce_slabs");
        constexpr int NE = G::B3 + 32 * G::CB;
        hipLaunchKernelGGL((k_wide_reduce<G>), dim3((NE + 255) / 256), dim3(256), 0, ctx->stream, nd, slab, ks, (long)G::SIZE, g_out);
        NF_HIP(hipGetLastError());
      }
    }
  }
  return NF_OK;
}

// ------------------------------------------------------------------------------------
// training step (nf_elbo_value_and_grad): forward with stash, reverse pass without recompute
// ------------------------------------------------------------------------------------
// workspace layout (floats): [forward stash: per (coupling, net) a1 | a2 | out | masks] [d1 d2 d3] [split-K slab]
static size_t fwd_stash_floats_per_net(long ntiles) {
  return (size_t)ntiles * ((size_t)NF_TILE * 32 * (G::H1B + G::H2B + G::CB) + 16 * 64);
}
static FwdStash fwd_stash_at(float *base, long ntiles, int k) {
  FwdStash fs;
  const size_t per = (size_t)ntiles * NF_TILE * 32;
  for (int net = 0; net < 2; ++net) {
    float *p = base + (size_t)(2 * k + net) * fwd_stash_floats_per_net(ntiles);
    fs.a1[net] = p; p += per * G::H1B;
    fs.a2[net] = p; p += per * G::H2B;
    fs.out[net] = p; p += per * G::CB;
    fs.mask[net] = (unsigned *)p;
  }
  return fs;
}

static size_t wide_train_ws_floats(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  const long ntiles = (N + NF_TILE - 1) / NF_TILE;
  const size_t fwd = (size_t)2 * desc->nlayers * 2 * fwd_stash_floats_per_net(ntiles);
  const size_t dstash = (size_t)ntiles * NF_TILE * 32 * (G::H1B + G::H2B + G::CB);
  const int ks = wide_ksplit(ctx, ntiles, wide_njobs());
  return fwd + dstash + (size_t)4 * desc->nlayers * ks * G::SIZE + 1024;  // one slab region per (coupling, net)
}

// whole chain forward on a base draw (in place on xt), stashing for the reverse pass
static int wide_train_forward(nf_ctx *ctx, const nf_flow_desc *desc, float *xt, long N, float *ladj, float *ws) {
  if (!ctx->wimg) return NF_ERR_UNSUPPORTED;
  const long ntiles = (N + NF_TILE - 1) / NF_TILE;
  const bool b6 = wide_b6();
  if (b6) NF_TRY(wide_b6_refresh(ctx, desc));
  const size_t lds = b6 ? Wide<G>::LDS_APPLY_B6 : Wide<G>::LDS_APPLY;
  static AttrOnce attr_once;  // once per device: a context on another GPU needs its own
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_apply<G, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wide<G>::LDS_APPLY));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_apply<G, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wide<G>::LDS_APPLY_B6));
    return NF_OK;
  }));
  long grid = wide_groups(N);
  if (grid > ctx->num_cu) grid = ctx->num_cu;
  if (grid < 1) grid = 1;
  const int nc = 2 * desc->nlayers;
  for (int s = 0; s < nc; ++s) {  // forward applies the LAST flat coupling first
    const int k = nc - 1 - s;
    const WideArgs a = make_wide_args(ctx, desc, k, N);
    const FwdStash fs = fwd_stash_at(ws, ntiles, k);
    ProfScope ps(ctx, "wide_apply");
    if (b6)
      hipLaunchKernelGGL((k_wide_apply<G, false, true, true>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, xt, ladj, s > 0 ? 1 : 0, fs);
    else
      hipLaunchKernelGGL((k_wide_apply<G, false, true>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, xt, ladj, s > 0 ? 1 : 0, fs);
    NF_HIP(hipGetLastError());
  }
  return NF_OK;
}

// floats of the forward stash alone / of the reverse pass's own scratch (delta stash + split-K slabs): the two parts of
// wide_train_ws_floats, for callers that keep the stash elsewhere (the tape of nf_flow_fwd_keep)
static size_t wide_fwd_stash_floats(nf_ctx *, const nf_flow_desc *desc, long N) {
  const long ntiles = (N + NF_TILE - 1) / NF_TILE;
  return (size_t)2 * desc->nlayers * 2 * fwd_stash_floats_per_net(ntiles);
}
static size_t wide_train_scratch_floats(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  return wide_train_ws_floats(ctx, desc, N) - wide_fwd_stash_floats(ctx, desc, N);
}

// scratch == nullptr: the reverse pass's scratch follows the forward stash inside ws (the training step's layout)
static int wide_train_backward(nf_ctx *ctx, const nf_flow_desc *desc, float *state, float *gbar, const float *lbar,
                           float lbar_const, long N, float *ws, float *g_out, float *scratch) {
  if (!ctx->wimg) return NF_ERR_UNSUPPORTED;
  const long ntiles = (N + NF_TILE - 1) / NF_TILE;
  const size_t per = (size_t)ntiles * NF_TILE * 32;
  const int nc = 2 * desc->nlayers;
  float *p = scratch ? scratch : ws + (size_t)nc * 2 * fwd_stash_floats_per_net(ntiles);
  WideStash st;
  st.a1 = nullptr; st.a2 = nullptr;
  st.d1 = p; p += per * G::H1B;
  st.d2 = p; p += per * G::H2B;
  st.d3 = p; p += per * G::CB;
  float *slab = p;
  const int njobs = wide_njobs();
  const int ks = wide_ksplit(ctx, ntiles, njobs);
  static AttrOnce attr_once;  // once per device
  const size_t lds_bwd = (size_t)2 * Wide<G>::CHBUF * sizeof(float);
  const size_t lds_dw = (size_t)2 * DW_ROWS * DW_TS * sizeof(float);
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_bwd_stashed<G, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bwd));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_bwd_stashed<G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bwd));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_bwd_stashed_b6<G, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WideT<G>::LDS));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_bwd_stashed_b6<G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WideT<G>::LDS));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_dw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dw));
    NF_HIP(hipFuncSetAttribute((const void *)k_wide_dw_b6, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DW_ROWS * D6_ROW));
    return NF_OK;
  }));
  long grid = wide_groups(N);
  if (grid > ctx->num_cu) grid = ctx->num_cu;
  if (grid < 1) grid = 1;
  const bool b6 = wide_b6();
  if (b6) NF_TRY(wide_b6t_refresh(ctx, desc));
  for (int k = 0; k < nc; ++k) {  // flat order = reverse of execution order
    const WideArgs a = make_wide_args(ctx, desc, k, N);
    const CouplingInfo ci = nf_coupling_info(desc, k);
    const FwdStash fs = fwd_stash_at(ws, ntiles, k);
    for (int phase = 0; phase < 2; ++phase) {  // 0: t net, 1: s net
      const int net = phase == 0 ? 1 : 0;
      {
        ProfScope ps(ctx, "wide_bwd");
        if (b6 && phase == 0)
          hipLaunchKernelGGL((k_wide_bwd_stashed_b6<G, false>), dim3((unsigned)grid), dim3(256), WideT<G>::LDS, ctx->stream, a, state, gbar, lbar, lbar_const, st, (const float *)fs.out[net], (const unsigned *)fs.mask[net]);
        else if (b6)
          hipLaunchKernelGGL((k_wide_bwd_stashed_b6<G, true>), dim3((unsigned)grid), dim3(256), WideT<G>::LDS, ctx->stream, a, state, gbar, lbar, lbar_const, st, (const float *)fs.out[net], (const unsigned *)fs.mask[net]);
        else if (phase == 0)
          hipLaunchKernelGGL((k_wide_bwd_stashed<G, false>), dim3((unsigned)grid), dim3(256), lds_bwd, ctx->stream, a, state, gbar, lbar, lbar_const, st, (const float *)fs.out[net], (const unsigned *)fs.mask[net]);
        else
          hipLaunchKernelGGL((k_wide_bwd_stashed<G, true>), dim3((unsigned)grid), dim3(256), lds_bwd, ctx->stream, a, state, gbar, lbar, lbar_const, st, (const float *)fs.out[net], (const unsigned *)fs.mask[net]);
        NF_HIP(hipGetLastError());
      }
      WideStash sj = st;
      sj.a1 = fs.a1[net];
      sj.a2 = fs.a2[net];
      DwArgs da;
      build_jobs(&da, state, (long)desc->d * NF_TILE, desc->d * NF_TILE * 4, 1 - ci.par_t, sj);
      da.ksplit = ks;
      da.ntiles = ntiles;
      da.slab_stride = G::SIZE;
      da.trace = (long long *)ctx->trace;
      {
        ProfScope ps(ctx, "wide_dw");
        if (wide_b6())
          hipLaunchKernelGGL(k_wide_dw_b6, dim3((unsigned)(da.njobs * ks)), dim3(256), 2 * DW_ROWS * D6_ROW, ctx->stream, da,
                             slab + (size_t)(2 * k + phase) * ks * G::SIZE);
        else
          hipLaunchKernelGGL(k_wide_dw, dim3((unsigned)(da.njobs * ks)), dim3(256), lds_dw, ctx->stream, da,
                             slab + (size_t)(2 * k + phase) * ks * G::SIZE);
        NF_HIP(hipGetLastError());
      }
    }
    // Data-parallel step with a bucketed all-reduce (nf_elbo_step under a communicator): the couplings [k0, k] are final --
    // sum their split-K partials now and hand their theta range (contiguous: flat coupling order IS destructure order) to
    // the second stream; the last bucket carries the loss at g_out[P] along.
    if (ctx->bucket.on && ((k + 1) % ctx->bucket.couplings == 0 || k == nc - 1)) {
      const int k0 = (k / ctx->bucket.couplings) * ctx->bucket.couplings;
      {
        ProfScope ps(ctx, "reduce_slabs");
        constexpr int NE = G::B3 + 32 * G::CB;
        hipLaunchKernelGGL((k_wide_reduce_all<G>), dim3((NE + 255) / 256, 2 * (k + 1 - k0)), dim3(256), 0, ctx->stream, make_pack_args(desc),
                           slab, ks, (long)G::SIZE, g_out, 2 * k0);
        NF_HIP(hipGetLastError());
      }
      const long lo = nf_coupling_info(desc, k0).theta_off;
      const long hi = k == nc - 1 ? nf_param_count(desc) + 1 : nf_coupling_info(desc, k + 1).theta_off;
      NF_TRY(nf_comm_bucket_issue(ctx, NF_DTYPE_F32, g_out + lo, hi - lo));
    }
  }
  if (!ctx->bucket.on) {
    ProfScope ps(ctx, "reduce_slabs");
    constexpr int NE = G::B3 + 32 * G::CB;
    hipLaunchKernelGGL((k_wide_reduce_all<G>), dim3((NE + 255) / 256, 2 * nc), dim3(256), 0, ctx->stream, make_pack_args(desc), slab, ks,
                       (long)G::SIZE, g_out, 0);
    NF_HIP(hipGetLastError());
  }
  return NF_OK;
}

};

using GM = NetGeo<2, 4, 4, 2>;  // d <= 128, hidden <= 128: a quarter of GW's padded flops

static bool wide_fits_mid(const nf_flow_desc *desc) {
  const int c = (desc->d + 1) / 2;
  return wblocks32(c) <= GM::MB && wblocks32(desc->hdims[0]) <= GM::H1B && wblocks32(desc->hdims[1]) <= GM::H2B;
}
#define WIDE_DISPATCH(CALL) (wide_fits_mid(desc) ? WideHost<GM>::CALL : WideHost<GW>::CALL)

int nf_wide_pack(nf_ctx *ctx, const nf_flow_desc *desc, const float *theta) { return WIDE_DISPATCH(wide_pack(ctx, desc, theta)); }
size_t nf_wide_wimg_bytes(nf_ctx *ctx, const nf_flow_desc *desc) { return WIDE_DISPATCH(wide_wimg_bytes(ctx, desc)); }
int nf_wide_apply(nf_ctx *ctx, const nf_flow_desc *desc, int k, bool inverse, float *xt, long N, float *ladj, int accumulate) { return WIDE_DISPATCH(wide_apply(ctx, desc, k, inverse, xt, N, ladj, accumulate)); }
size_t nf_wide_bwd_ws_floats(nf_ctx *ctx, const nf_flow_desc *desc, long N) { return WIDE_DISPATCH(wide_bwd_ws_floats(ctx, desc, N)); }
int nf_wide_bwd(nf_ctx *ctx, const nf_flow_desc *desc, float *state, float *gbar, const float *lbar, float lbar_const, long N, float *ws, float *g_out, bool inv_dir) { return WIDE_DISPATCH(wide_bwd(ctx, desc, state, gbar, lbar, lbar_const, N, ws, g_out, inv_dir)); }
size_t nf_wide_train_ws_floats(nf_ctx *ctx, const nf_flow_desc *desc, long N) { return WIDE_DISPATCH(wide_train_ws_floats(ctx, desc, N)); }
int nf_wide_train_forward(nf_ctx *ctx, const nf_flow_desc *desc, float *xt, long N, float *ladj, float *ws) { return WIDE_DISPATCH(wide_train_forward(ctx, desc, xt, N, ladj, ws)); }
int nf_wide_train_backward(nf_ctx *ctx, const nf_flow_desc *desc, float *state, float *gbar, const float *lbar, float lbar_const, long N, float *ws, float *g_out, float *scratch) { return WIDE_DISPATCH(wide_train_backward(ctx, desc, state, gbar, lbar, lbar_const, N, ws, g_out, scratch)); }
size_t nf_wide_fwd_stash_floats(nf_ctx *ctx, const nf_flow_desc *desc, long N) { return WIDE_DISPATCH(wide_fwd_stash_floats(ctx, desc, N)); }
size_t nf_wide_train_scratch_floats(nf_ctx *ctx, const nf_flow_desc *desc, long N) { return WIDE_DISPATCH(wide_train_scratch_floats(ctx, desc, N)); }
