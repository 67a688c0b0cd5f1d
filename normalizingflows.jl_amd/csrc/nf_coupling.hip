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
#include <cmath>
#include <cstdlib>

#include "nf_common.h"
// A/B seams of this translation unit only (tools/ab_build.py): the split's subtractions / the leaky-ReLU slopes as scalar f32 instructions.
// Packed f32 instructions do not overlap a matrix instruction in flight (tools/probe/mfma_valu_overlap_probe.hip: eight v_pk_add_f32
// behind an MFMA cost 79.6 clocks against 42.4 for eight v_sub_f32), but here two scalar instructions per packed one buy nothing:
// step 0.6042 / 0.6036 against 0.6040 ms over six alternating runs (profiles/r6l_pair_scalar_forms_ab2.txt)
#ifdef NF_COUPLING_SPLIT_SCALAR
#define NF_SPLIT_SCALAR
#endif
#ifdef NF_COUPLING_SLOPE_SCALAR
#define NF_SLOPE_SCALAR
#endif
#include "nf_mfma.h"
#include "nf_pack.h"
#include "nf_philox.h"

struct CouplingArgs {
  const float *theta;
  const float *img_s, *img_t;  // pre-packed LDS images of the s / t nets (k_pack_net_images)
  long long *trace;            // optional s_memtime stamps of block 0 / wave 0 (nf_debug_trace)
  NetDims s, t;
  int d, c, m, par_t;
  long N;
};

// ------------------------------------------------------------------------------------
// forward / inverse of one coupling (no gradients): both nets resident in LDS
// ------------------------------------------------------------------------------------
template <class G, bool INVERSE, bool FULL>
__global__ __launch_bounds__(512) void k_affine_apply(CouplingArgs a, const float *x_in,
                                                      float *y_out, float *__restrict__ ladj,
                                                      int ladj_accumulate) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *img_s = lds;
  float *img_t = lds + G::SIZE;
  const int tid = threadIdx.x;
  stage_packed<G::SIZE, 512>(img_s, a.img_s, tid);
  stage_packed<G::SIZE, 512>(img_t, a.img_t, tid);
  __syncthreads();

  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int l31 = lane & 31, hi = lane >> 5;
  const int par_c = 1 - a.par_t;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const bool copy_cond = (y_out != x_in);

  for (long tile = (long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long)gridDim.x * 8) {
    const long j = tile * NF_TILE + l31;
    const bool valid = FULL ? true : j < a.N;
    const TileIO xin = make_tile_io(const_cast<float *>(x_in), tile, a.d, l31, hi);
    const TileIO yout = make_tile_io(y_out, tile, a.d, l31, hi);

    f32x16 xb[G::MB], x1[G::CB];
#pragma unroll
    for (int b = 0; b < G::MB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = tile_load(xin, tile_soff(b, r, par_c));
        xb[b][r] = valid ? v : 0.f;
        if (copy_cond) tile_store(yout, tile_soff(b, r, par_c), v);
      }
#pragma unroll
    for (int b = 0; b < G::CB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) x1[b][r] = tile_load(xin, tile_soff(b, r, a.par_t));

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
        const float s = nf_tanh(S[b][r]);  // padded rows: zero weights and bias => s = 0
        const float v = x1[b][r];
        float o;
        if (INVERSE)
          o = (v - T[b][r]) * nf_exp(-s);
        else
          o = v * nf_exp(s) + T[b][r];
        tile_store(yout, tile_soff(b, r, a.par_t), o);  // rows >= c are outside the descriptor
        lsum += (FULL || p < a.c) ? s : 0.f;
      }
    lsum += __shfl_xor(lsum, 32);
    if (hi == 0 && valid) {
      const float base = ladj_accumulate ? ladj[j] : 0.f;
      ladj[j] = INVERSE ? base - lsum : base + lsum;
    }
  }
}

// ------------------------------------------------------------------------------------
// whole-flow forward / inverse in ONE launch (with_logabsdet_jacobian of the ComposedFunction,
// reached from src/objectives/elbo.jl:67; inverse chain from loglikelihood.jl:31)
// ------------------------------------------------------------------------------------
// A workgroup (8 waves) takes a group of 8 tiles through every coupling.  The state of a tile
// lives in registers for the whole chain, split by feature parity: E = features 0,2,4,..
// (transformed by the 1:2:d couplings), O = features 1,3,5,.. (transformed by the 2:2:d ones),
// so "partition/combine" is just which of the two register blocks plays x1 and which x2.
// The two nets of the NEXT coupling are fetched from their packed images (L2) while the current
// coupling computes, into the other half of a double-buffered LDS image.
struct ChainArgs {
  const float *wimg;  // [coupling][s|t][G::SIZE] packed images
  const unsigned char *wimg_b6;  // [coupling][s|t][B6Geo<G>::BYTES]: the same weights as bf16 triples (B6 kernels)
  int d, ncoup;
  long N;
};

// the conditioner net on the bf16 matrix cores (nf_mfma.h "B6"): same layers, the image is a B6Geo<G> image
// PIPE: the layers as software pipelines (dense_fwd_b6p); false: the plain form, 44 registers less (three waves per SIMD)
#ifndef NF_CHAIN_LRIN
#define NF_CHAIN_LRIN 1
#endif
#ifndef NF_CHAIN_LEAN
#define NF_CHAIN_LEAN false  // the plain form without its operand double buffer (12 registers less)
#endif
template <class G, bool PIPE = true>
__device__ __forceinline__ void net_forward_b6(const float *__restrict__ img, const f32x16 (&x)[G::MB], f32x16 (&out)[G::CB],
                                               int l31, int hi) {
  using B = B6Geo<G>;
  const nf_u32x4 *w = reinterpret_cast<const nf_u32x4 *>(img);
  const float *bias = reinterpret_cast<const float *>(w + B::BIAS);
  f32x16 a1[G::H1B], a2[G::H2B];
  if constexpr (PIPE && NF_CHAIN_LRIN) {  // the leaky ReLUs inside the next layer's splits (dense_fwd_b6p<..., LRIN>): same values, same bits
    dense_fwd_b6p<G::MB, G::H1B>(w + B::L1, bias + B::B1, x, a1, l31, hi);
    dense_fwd_b6p<G::H1B, G::H2B, true>(w + B::L2, bias + B::B2, a1, a2, l31, hi);
    dense_fwd_b6p<G::H2B, G::CB, true>(w + B::L3, bias + B::B3, a2, out, l31, hi);
    return;
  }
  if constexpr (PIPE) dense_fwd_b6p<G::MB, G::H1B>(w + B::L1, bias + B::B1, x, a1, l31, hi);
  else dense_fwd_b6<G::MB, G::H1B, NoSideJob, NF_CHAIN_LEAN>(w + B::L1, bias + B::B1, x, a1, l31, hi);
#pragma unroll
  for (int b = 0; b < G::H1B; ++b) nf_lrelu16(a1[b]);
  if constexpr (PIPE) dense_fwd_b6p<G::H1B, G::H2B>(w + B::L2, bias + B::B2, a1, a2, l31, hi);
  else dense_fwd_b6<G::H1B, G::H2B, NoSideJob, NF_CHAIN_LEAN>(w + B::L2, bias + B::B2, a1, a2, l31, hi);
#pragma unroll
  for (int b = 0; b < G::H2B; ++b) nf_lrelu16(a2[b]);
  if constexpr (PIPE) dense_fwd_b6p<G::H2B, G::CB>(w + B::L3, bias + B::B3, a2, out, l31, hi);
  else dense_fwd_b6<G::H2B, G::CB, NoSideJob, NF_CHAIN_LEAN>(w + B::L3, bias + B::B3, a2, out, l31, hi);
}

struct NoBetween {
  __device__ __forceinline__ void operator()() const {}
};
// B6: img_s / img_t are B6 images; `between` runs between the two nets (the B6 chain kernel's image rotation: a workgroup
// barrier and the request for the image after next)
#ifdef NF_KERNEL_TRACE
#define NF_CS_STAMP(tr, slot) do { if (tr) { __builtin_amdgcn_sched_barrier(0); (tr)[slot] = clock64(); } } while (0)
#else
#define NF_CS_STAMP(tr, slot) do { } while (0)
#endif
template <class G, bool INVERSE, bool B6 = false, class BT = NoBetween, bool PIPE = true>
__device__ __forceinline__ float coupling_step(const float *__restrict__ img_s, const float *__restrict__ img_t,
                                               f32x16 (&x1)[G::CB], const f32x16 (&xb)[G::MB], int l31, int hi, BT between = BT(),
                                               long long *tr = nullptr) {
  f32x16 S[G::CB], T[G::CB];
  if constexpr (B6) {
    NF_CS_STAMP(tr, 0);  // tools/trace_chain.py (NOSTASH=1): after the phase barrier + the request for the image after next
    net_forward_b6<G, PIPE>(img_s, xb, S, l31, hi);
    NF_CS_STAMP(tr, 1);
    between();
    NF_CS_STAMP(tr, 2);
    net_forward_b6<G, PIPE>(img_t, xb, T, l31, hi);
    NF_CS_STAMP(tr, 3);
  } else {
  {
    f32x16 a1[G::H1B], a2[G::H2B];
    net_forward<G>(img_s, xb, a1, a2, S, l31, hi);
  }
  between();
  {
    f32x16 a1[G::H1B], a2[G::H2B];
    net_forward<G>(img_t, xb, a1, a2, T, l31, hi);
  }
  }
  float lsum = 0.f;
#pragma unroll
  for (int b = 0; b < G::CB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // rows >= c have zero weights and biases in the packed image: s = 0, T = 0, x1 stays 0
      const float s = nf_tanh(S[b][r]);
      if (INVERSE)
        x1[b][r] = nf_fdiv(x1[b][r] - T[b][r], nf_exp(s));
      else
        x1[b][r] = x1[b][r] * nf_exp(s) + T[b][r];
      lsum += s;
    }
  return lsum;
}

// ---- activation stash of the training step (forward writes, reverse pass reads) ---------------------------
// The reverse pass needs, per (tile, coupling): the conditioner input x2, both nets' hidden activations a1 / a2 and
// their leaky-ReLU sign masks, s = tanh(.) and u = x1 exp(s).  Recomputing them (k_affine_bwd_all) costs a third of
// that kernel's MFMAs plus the activations' LDS transposes; with STASH the fused forward writes them once, in the
// layouts the reverse pass consumes directly:
//   "T layout" (x2, a1, a2 -- the dW operands): [feature 32][sample parity 2][16] per 32-feature block, so the lane
//     that owns feature f in the dW GEMM reads its 16 k-steps (samples 2t + parity) as four 16-byte loads;
//   "lane layout" (s, u, masks -- element-wise operands in the MFMA C layout): [lane 64][16] per block.
// 46 KiB per (tile, coupling) for d = 64 / hidden 64: 772 MB per step at cfg 2, written once and read once.
// SLIM (round 4; optional, NF_STASH_SLIM=1 -- see stash_slim() for the measurement that keeps it off): WITHOUT a1.  The first hidden layer's activations are 35 %
// of the full stash (128 of 368 floats per sample and coupling) and the cheapest thing in it to rebuild: the consumer wave
// recomputes a1^T = leakyrelu(W1 x2 + b1) from x2 -- kept a second time in the MFMA C layout (XL), which is the A operand
// of that product -- with 32 MFMAs per net and tile while the producer wave is in its MFMA-free prologue.  34 KiB instead
// of 46 KiB per (tile, coupling): 570 instead of 772 MB per cfg-2 step, written once and read once.
template <class G, bool SLIM = false>
struct StashGeo {
  static constexpr int XT = 0;
  static constexpr int XL = XT + G::MB * 1024;  // SLIM only: x2 in the C ("lane") layout
  static constexpr int SV = XL + (SLIM ? G::MB * 1024 : 0);
  static constexpr int UV = SV + G::CB * 1024;
  static constexpr int NET0 = UV + G::CB * 1024;  // net 0 = s, net 1 = t
  static constexpr int A1 = 0, A2 = SLIM ? 0 : G::H1B * 1024, MSK = A2 + G::H2B * 1024, NETSZ = MSK + 256;
  static constexpr int SIZE = NET0 + 2 * NETSZ;  // floats per (tile, coupling), a multiple of 4
};
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// A buffer store of more than 64 bits must not be followed directly by a VALU write of its data registers.  hipcc's hazard
// recognizer inserts the wait state only when the instruction's soffset is NOT an SGPR (the ISA manual exempts the SGPR
// form).  Measured on MI355X, the SGPR form races as well: in k_affine_chain<hidden 32, FUSED, STASH> the mask store was
// followed by `v_add_u32` into its first data register, and lanes 12-15 of every 16-lane row of that word arrived in
// memory as the sum -- an LDS address -- in most runs with 8 live waves per workgroup (tools/kernel_resources.py,
// DESIGN.md section 5).  So wide stores pass their whole offset through the VGPR / immediate fields and soffset = 0, which
// puts them under the compiler's own guard.
__device__ __forceinline__ void nf_buffer_store_b128(u32x4 v, __amdgpu_buffer_rsrc_t rs, int voff, int coff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + coff, 0, 0);
}
struct StashIO {
  __amdgpu_buffer_rsrc_t rs;
  int vT, vL;  // per-lane byte offsets: T-layout element stores (C-layout lane), lane-layout rows
};
// live = false: a descriptor of extent 0, every store is dropped (idle waves of the last tile group)
__device__ __forceinline__ StashIO make_stash_io(float *stash, long slot, int size, bool live, int l31, int hi) {
  StashIO st;
  st.rs = __builtin_amdgcn_make_buffer_rsrc(stash + slot * size, 0, live ? size * 4 : 0, 0x00020000);
  st.vT = ((4 * hi) * 32 + (l31 & 1) * 16 + (l31 >> 1)) * 4;
  st.vL = (hi * 32 + l31) * 64;
  return st;
}
// element e (= block * 16 + reg) of a C-layout array into its T-layout slot
template <int NB>
__device__ __forceinline__ void stash_put_T(const StashIO &st, int base, const f32x16 (&v)[NB], int e) {
  if (e < NB * 16) {
    const int r = e & 15;
    const float val = v[e >> 4][r];  // (bit_cast straight on a vector element reads element 0 under hipcc 7.2)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), st.rs, st.vT,
                                          (base + ((e >> 4) * 32 + (r & 3) + 8 * (r >> 2)) * 32) * 4, 0);
  }
}
template <int NB>
__device__ __forceinline__ void stash_put_lane(const StashIO &st, int base, const f32x16 (&v)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float val = v[b][4 * q + e];
        w[e] = __builtin_bit_cast(unsigned, val);
      }
      nf_buffer_store_b128(w, st.rs, st.vL, (base + b * 1024) * 4 + q * 16);
    }
}

template <class G, bool STORE_X, bool SLIM = false, bool B6 = false>
__device__ __forceinline__ void net_forward_stash(const float *__restrict__ img, const f32x16 (&x)[G::MB], f32x16 (&out)[G::CB],
                                                  int l31, int hi, const StashIO &st, int nbase) {
  using SG = StashGeo<G, SLIM>;
  f32x16 a1[G::H1B], a2[G::H2B];
  unsigned m1[2] = {0u, 0u}, m2[2] = {0u, 0u};
  if constexpr (B6) {  // the same stores ride on the bf16 product's (fewer) MFMAs: two side-job slots per instruction
    using B = B6Geo<G>;
    const nf_u32x4 *w = reinterpret_cast<const nf_u32x4 *>(img);
    const float *bias = reinterpret_cast<const float *>(w + B::BIAS);
    auto sjx = [&](int e) { stash_put_T<G::MB>(st, SG::XT, x, e); };
    auto sj1 = [&](int e) { stash_put_T<G::H1B>(st, nbase + SG::A1, a1, e); };
    auto sj2 = [&](int e) { stash_put_T<G::H2B>(st, nbase + SG::A2, a2, e); };
    if (STORE_X) {
      if (SLIM) stash_put_lane<G::MB>(st, SG::XL, x);
      dense_fwd_b6<G::MB, G::H1B, decltype(sjx), true>(w + B::L1, bias + B::B1, x, a1, l31, hi, sjx);
    } else
      dense_fwd_b6<G::MB, G::H1B, NoSideJob, true>(w + B::L1, bias + B::B1, x, a1, l31, hi);
#pragma unroll
    for (int b = 0; b < G::H1B; ++b) {
      nf_lrelu16(a1[b]);
      m1[b] = nf_sign_mask16(a1[b]);
    }
    if (SLIM)
      dense_fwd_b6<G::H1B, G::H2B, NoSideJob, true>(w + B::L2, bias + B::B2, a1, a2, l31, hi);
    else
      dense_fwd_b6<G::H1B, G::H2B, decltype(sj1), true>(w + B::L2, bias + B::B2, a1, a2, l31, hi, sj1);
#pragma unroll
    for (int b = 0; b < G::H2B; ++b) {
      nf_lrelu16(a2[b]);
      m2[b] = nf_sign_mask16(a2[b]);
    }
    dense_fwd_b6<G::H2B, G::CB, decltype(sj2), true>(w + B::L3, bias + B::B3, a2, out, l31, hi, sj2);
    const u32x4 mkb = {m1[0], m1[1], m2[0], m2[1]};
    nf_buffer_store_b128(mkb, st.rs, (hi * 32 + l31) * 16, (nbase + SG::MSK) * 4);
    return;
  }
  if (STORE_X) {
    if (SLIM) stash_put_lane<G::MB>(st, SG::XL, x);
    dense_fwd<G::MB, G::H1B>(img + G::W1, img + G::B1, x, a1, l31, hi, [&](int e) { stash_put_T<G::MB>(st, SG::XT, x, e); });
  } else
    dense_fwd<G::MB, G::H1B>(img + G::W1, img + G::B1, x, a1, l31, hi);
#pragma unroll
  for (int b = 0; b < G::H1B; ++b) {
    nf_lrelu16(a1[b]);
    m1[b] = nf_sign_mask16(a1[b]);
  }
  if (SLIM)
    dense_fwd<G::H1B, G::H2B>(img + G::W2, img + G::B2, a1, a2, l31, hi);
  else
    dense_fwd<G::H1B, G::H2B>(img + G::W2, img + G::B2, a1, a2, l31, hi,
                              [&](int e) { stash_put_T<G::H1B>(st, nbase + SG::A1, a1, e); });
#pragma unroll
  for (int b = 0; b < G::H2B; ++b) {
    nf_lrelu16(a2[b]);
    m2[b] = nf_sign_mask16(a2[b]);
  }
  dense_fwd<G::H2B, G::CB>(img + G::W3, img + G::B3, a2, out, l31, hi,
                           [&](int e) { stash_put_T<G::H2B>(st, nbase + SG::A2, a2, e); });
  const u32x4 mk = {m1[0], m1[1], m2[0], m2[1]};
  nf_buffer_store_b128(mk, st.rs, (hi * 32 + l31) * 16, (nbase + SG::MSK) * 4);
}

#ifdef NF_KERNEL_TRACE
#define NF_CH_STAMP(tr, slot) do { if (tr) { __builtin_amdgcn_sched_barrier(0); (tr)[slot] = clock64(); } } while (0)
#else
#define NF_CH_STAMP(tr, slot) do { } while (0)
#endif
// forward coupling of the training step: as coupling_step<G, false>, leaving the reverse pass's operands behind
// INVERSE (forward-KL training: the chain runs data -> base): w1 = (v1 - t) exp(-s); the UV slot then holds w1, which is
// what the reverse pass of the inverse coupling needs next to s (bwd_tile's INVD algebra)
template <class G, bool INVERSE = false, bool SLIM = false, bool B6 = false, class BT = NoBetween>
__device__ __forceinline__ float coupling_step_stash(const float *__restrict__ img_s, const float *__restrict__ img_t,
                                                     f32x16 (&x1)[G::CB], const f32x16 (&xb)[G::MB], int l31, int hi,
                                                     const StashIO &st, long long *tr = nullptr, BT between = BT()) {
  using SG = StashGeo<G, SLIM>;
  static_assert(G::H1B <= 2 && G::H2B <= 2, "mask words");
  f32x16 S[G::CB], T[G::CB];
  net_forward_stash<G, true, SLIM, B6>(img_s, xb, S, l31, hi, st, SG::NET0);
  NF_CH_STAMP(tr, 1);
  between();
  net_forward_stash<G, false, SLIM, B6>(img_t, xb, T, l31, hi, st, SG::NET0 + SG::NETSZ);
  NF_CH_STAMP(tr, 2);
  float lsum = 0.f;
#pragma unroll
  for (int b = 0; b < G::CB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float s = nf_tanh(S[b][r]);
      if (INVERSE) {
        const float w = nf_fdiv(x1[b][r] - T[b][r], nf_exp(s));
        x1[b][r] = w;
        T[b][r] = w;
      } else {
        const float u = x1[b][r] * nf_exp(s);
        x1[b][r] = u + T[b][r];
        T[b][r] = u;
      }
      S[b][r] = s;
      lsum += s;
    }
  stash_put_lane<G::CB>(st, SG::SV, S);
  stash_put_lane<G::CB>(st, SG::UV, T);
  return lsum;
}

// FUSED (forward only): the ELBO forward of the training step in the same launch -- the tile's base
// draws are generated in registers (Philox4x32-10 + Box-Muller, same counters as k_base_sample_tiled:
// (sample lo, sample hi, feature group, stream)), log q0 and log|det J| never touch memory, and after
// the last coupling the diagonal-Gaussian target, ybar = gscale * grad log p(y) and the workgroup's
// partial sum of pscale * elbo_j are computed from the registers (src/objectives/elbo.jl:65-70,93-97).
struct FusedArgs {
  uint32_t k0, k1, stream;
  const uint32_t *stream_ptr;  // non-null: the Philox stream id is read from device memory (hipGraph replay of the step)
  uint64_t off;           // global index of this shard's first sample
  const float *mu, *var;  // diagonal-Gaussian target (test/flow.jl:43-46)
  float *gt;              // ybar out (tiled), or nullptr
  float gscale;
  double *partial;        // [gridDim.x] out
  double pscale;
  float *stash;           // STASH: [tile][coupling][StashGeo<G>::SIZE] out
  long long *trace;       // NF_KERNEL_TRACE builds: clock stamps for tools/trace_chain.py, else unused
};

// B6 (round 4, the default): the conditioner GEMMs on the bf16 matrix cores with six-term products (nf_mfma.h), 2.67 x
// the fp32 MFMA's rate at fp32 accuracy.  The B6 images are 1.5 x the fp32 ones (49 KB per net at hidden 64), so the LDS
// holds THREE of them and rotates per NET instead of two (s, t) pairs per coupling: image i of the workgroup's sequence
// s(c0), t(c0), s(c1), ... lives in slot i mod 3; the phase of image i starts with a workgroup barrier (image i complete,
// everybody done with image i - 1) and then requests image i + 2 into the slot image i - 1 just left.
// NW: wavefronts (= tiles of a group) per workgroup; 12 (three per SIMD) where the instantiation fits 168 registers (round 6)
template <class G, bool INVERSE, bool FUSED = false, bool STASH = false, bool SLIM = false, bool B6 = false, int NW = 8>
__global__ __launch_bounds__(64 * NW) void k_affine_chain(ChainArgs a, float *xt, float *__restrict__ ladj, FusedArgs fa) {
  static_assert(STASH || !SLIM, "SLIM is a stash layout");
  static_assert(!STASH || !FUSED || !INVERSE, "the fused ELBO forward runs base -> data");
  static_assert(G::MB == G::CB, "parity blocks must have equal padded size");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int IMG2 = 2 * G::SIZE;  // s and t images of one coupling are adjacent in wimg
  constexpr int NV4 = IMG2 / 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const long ngroups = (ntiles + NW - 1) / NW;
  // coupling executed at position s of the chain: forward applies the LAST flat coupling first
  auto coupling_at = [&](int s) { return INVERSE ? s : a.ncoup - 1 - s; };

  using BG = B6Geo<G>;
  constexpr int B6F = BG::BYTES / 4;  // floats per B6 image
  // B6: image i of this workgroup's sequence (position (i mod 2 ncoup) / 2 of the chain, net i & 1) -> LDS slot i mod 3, by
  // LDS-DMA; complete at the next workgroup barrier (s_waitcnt vmcnt(0) of every wave for its own pieces)
  const int my_groups = (long)blockIdx.x < ngroups ? (int)((ngroups - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
  const int my_images = my_groups * 2 * a.ncoup;
  int req = 0, req_idx = 0, req_slot = 0;  // images requested so far, the next one's index in the chain and its slot
  auto b6_request_next = [&]() {  // requests are issued strictly in sequence order: no division, three small counters
    if (req >= my_images) return;
    typedef __attribute__((address_space(3))) void lds_void_t;
    const int kk = coupling_at(req_idx >> 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(a.wimg_b6) + (size_t)(2 * kk + (req_idx & 1)) * BG::BYTES, 0, BG::BYTES, 0x00020000);
    float *dstb = lds + req_slot * B6F;
    constexpr int NP = (BG::BYTES + 1023) / 1024;
    for (int p = wave; p < NP; p += NW)
      if (p * 1024 + lane * 16 < BG::BYTES)  // the last piece is partial: its idle lanes must not write
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(dstb + p * 256), 16, lane * 16, p * 1024, 0, 0);
    ++req;
    req_idx = req_idx + 1 == 2 * a.ncoup ? 0 : req_idx + 1;
    req_slot = req_slot == 2 ? 0 : req_slot + 1;
  };
  int cur_slot = 0;  // B6: slot of the next image to be used
  // prologue: image of the first coupling -> buffer 0
  if constexpr (B6) {
    b6_request_next();
    b6_request_next();
  } else {
    const float4 *src = reinterpret_cast<const float4 *>(a.wimg + (long)coupling_at(0) * IMG2);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    for (int i = tid; i < NV4; i += 64 * NW) dst[i] = src[i];
  }
  // FUSED: target parameters by feature, zero padded: tmu[f], tiv[f] = 1/var[f]; tc0 = d log 2pi + sum log var
  constexpr int TP = 64 * G::CB;  // padded feature count (E and O halves)
  float *tmu = lds + (B6 ? 3 * B6F : 2 * IMG2), *tiv = tmu + TP, *tc0 = tiv + TP;
  double *wsum = reinterpret_cast<double *>(tc0 + 2);  // [NW] per-wave partial sums (8-byte aligned: TP even)
  if (FUSED) {
    for (int i = tid; i < TP; i += 64 * NW) {
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

  int buf = 0;
  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long tile = grp * NW + wave;
    const bool live = tile < ntiles;          // wave-uniform
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
          const float e = tile_load(io, tile_soff(b, r, 0));  // features >= d read as 0
          const float o = tile_load(io, tile_soff(b, r, 1));
          E[b][r] = valid ? e : 0.f;
          O[b][r] = valid ? o : 0.f;
        }
    } else {
      // registers (b, 4q..4q+3) of E and O are features base..base+7, base = 64b + 16q + 8hi:
      // Philox groups base/4 (-> E0 O0 E1 O1) and base/4 + 1 (-> E2 O2 E3 O3)
      const uint64_t gj = fa.off + (uint64_t)j;
      const uint32_t pstream = fa.stream_ptr ? __builtin_amdgcn_readfirstlane(*fa.stream_ptr) : fa.stream;
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int base = 64 * b + 16 * q + 8 * hi;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int g = base / 4 + h;
            U4 c = {(uint32_t)gj, (uint32_t)(gj >> 32), (uint32_t)g, pstream};
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
#pragma unroll 1
    for (int s = 0; s < a.ncoup; s += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int pos = s + half;
        if constexpr (B6) {
          // The phase barrier must see this wave's pieces of the image DMA complete -- NOT its stash stores: the vector-memory
          // counter is in order, and at least 64 stores follow every DMA request of a stashing kernel (81 per s net, 73 per
          // t net and tile), so "at most 60 operations outstanding" already implies the DMA is done, while __syncthreads()
          // (vmcnt(0)) would drain the stores to HBM twice per coupling.  Kernels without a stash wait for everything.
          auto phase_barrier = [&](bool first_of_group = false) {
#ifndef NF_B6_FULL_DRAIN
            if (STASH && !first_of_group) {  // (a tile group's first image may have nothing but the DMA behind it)
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_s_waitcnt(0xC07C);  // vmcnt(60) lgkmcnt(0)
              __builtin_amdgcn_s_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else
#endif
              __syncthreads();
          };
          phase_barrier(s == 0 && half == 0);  // this coupling's s image is complete, every wave is done with the image before it
          b6_request_next();  // the image after next, into the slot just vacated
          const int slot_t = cur_slot == 2 ? 0 : cur_slot + 1;
          const float *img_s = lds + cur_slot * B6F, *img_t = lds + slot_t * B6F;
          auto between = [&]() {
            phase_barrier();  // the t image is complete, every wave is done with the s image
            b6_request_next();
          };
          float ls;
          if (STASH) {
            const StashIO st = make_stash_io(fa.stash, tl * a.ncoup + coupling_at(pos), StashGeo<G, SLIM>::SIZE, live, l31, hi);
#ifdef NF_KERNEL_TRACE  // tools/trace_chain.py (as the fp32 form below; stamp 3 = end of the combine, 4 = the same: the barrier opens the NEXT position)
            long long *tr = (fa.trace && blockIdx.x == 0 && (tid & 255) == 0 && pos < 8) ? fa.trace + 32 + (tid >> 8) * 64 + pos * 4 - 1 : nullptr;
#else
            long long *tr = nullptr;
#endif
            if (INVERSE ? (half == 1) : (half == 0)) ls = coupling_step_stash<G, INVERSE, SLIM, true>(img_s, img_t, O, E, l31, hi, st, tr, between);
            else ls = coupling_step_stash<G, INVERSE, SLIM, true>(img_s, img_t, E, O, l31, hi, st, tr, between);
            NF_CH_STAMP(tr, 3);
            NF_CH_STAMP(tr, 4);
          } else {
#ifdef NF_KERNEL_TRACE  // stamps [32 + wave/4 * 64 + position * 5 + 0..4] of the second tile group (the first one's images arrive cold)
            long long *tr = (fa.trace && blockIdx.x == 0 && (tid & 255) == 0 && pos < 8 && grp != (long)blockIdx.x && grp == (long)blockIdx.x + gridDim.x)
                                ? fa.trace + 32 + (tid >> 8) * 48 + pos * 5 : nullptr;
#else
            long long *tr = nullptr;
#endif
            if (INVERSE ? (half == 1) : (half == 0)) ls = coupling_step<G, INVERSE, true, decltype(between), NW == 8>(img_s, img_t, O, E, l31, hi, between, tr);
            else ls = coupling_step<G, INVERSE, true, decltype(between), NW == 8>(img_s, img_t, E, O, l31, hi, between, tr);
            NF_CS_STAMP(tr, 4);
          }
          lsum += ls;
          cur_slot = slot_t == 2 ? 0 : slot_t + 1;
          continue;
        }
        // prefetch the next coupling's images (wraps to the first coupling for the next tile group)
        const bool have_next = pos + 1 < a.ncoup || more_groups;
        const int knext = coupling_at(pos + 1 < a.ncoup ? pos + 1 : 0);
        if (have_next) {
          // LDS-DMA (buffer_load_dwordx4 ... lds) straight into the other image buffer: no staging
          // registers, no ds_write pass; complete at the barrier below (s_waitcnt vmcnt(0))
          typedef __attribute__((address_space(3))) void lds_void_t;
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wimg) + (long)knext * IMG2, 0, IMG2 * 4, 0x00020000);
          float *dstb = lds + (buf ^ 1) * IMG2;
          constexpr int NP = (IMG2 * 4 + 1023) / 1024;
          for (int p = wave; p < NP; p += NW)
            if (p * 1024 + lane * 16 < IMG2 * 4)  // the last piece is partial: its idle lanes must not write
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(dstb + p * 256), 16, lane * 16, p * 1024, 0, 0);
        }
        const float *img_s = lds + buf * IMG2;
        const float *img_t = img_s + G::SIZE;
        // forward order: position 0 is the last flat coupling (odd index => mask 2:2:d => x1 = O)
        const bool x1_is_O = (coupling_at(pos) & 1) != 0;  // compile-time per (INVERSE, half) when ncoup is even
        float ls;
        if (STASH) {
          const StashIO st = make_stash_io(fa.stash, tl * a.ncoup + coupling_at(pos), StashGeo<G, SLIM>::SIZE, live, l31, hi);
#ifdef NF_KERNEL_TRACE  // tools/trace_chain.py: block 0, waves 0 and 4 (one SIMD), stamps [32 + wave/4 * 64 + position * 4 + 0..3]
          // (the slots the reverse kernels of the same step leave alone)
          long long *tr = (fa.trace && blockIdx.x == 0 && (tid & 255) == 0 && pos < 8) ? fa.trace + 32 + (tid >> 8) * 64 + pos * 4 - 1 : nullptr;
#else
          long long *tr = nullptr;
#endif
          if (INVERSE ? (half == 1) : (half == 0)) ls = coupling_step_stash<G, INVERSE, SLIM>(img_s, img_t, O, E, l31, hi, st, tr);
          else ls = coupling_step_stash<G, INVERSE, SLIM>(img_s, img_t, E, O, l31, hi, st, tr);
          NF_CH_STAMP(tr, 3);
          lsum += ls;
          __syncthreads();
          NF_CH_STAMP(tr, 4);
          buf ^= 1;
          continue;
        } else if (INVERSE ? (half == 1) : (half == 0)) {
          (void)x1_is_O;
          ls = coupling_step<G, INVERSE>(img_s, img_t, O, E, l31, hi);
        } else {
          ls = coupling_step<G, INVERSE>(img_s, img_t, E, O, l31, hi);
        }
        lsum += ls;
        __syncthreads();
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
        if (hi == 0 && valid) ladj[j] = INVERSE ? -lsum : lsum;
      }
    }
    if (FUSED) {
      // elbo_j = log p(y_j) - log q0(x_j) + ladj_j ;  ybar = gscale * grad log p(y)
      const TileIO gio = make_tile_io(fa.gt ? fa.gt : xt, tl, a.d, l31, hi);
      float t = 0.f;
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int fe = 2 * (b * 32 + nf_row(r, hi));
          const float re = E[b][r] - tmu[fe], ro = O[b][r] - tmu[fe + 1];
          const float ge = re * tiv[fe], go = ro * tiv[fe + 1];
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
        for (int w = 0; w < NW; ++w) sgrp += wsum[w];
        wg_total += sgrp;
      }
      __syncthreads();
    }
  }
  if (FUSED && tid == 0) fa.partial[blockIdx.x] = wg_total;
}

// ------------------------------------------------------------------------------------
// the six-term chain WITHOUT a stash, two tiles per wavefront (round 6)
// ------------------------------------------------------------------------------------
// tools/trace_chain_b6.py on k_affine_chain<..., B6> at cfg 5: the two waves of a SIMD do not overlap -- the older one runs its net
// in 5.4 k clocks (96 MFMAs = 3.1 k) and then waits 5 k at the phase barrier for the younger one, which only gets the issue slots the
// older leaves (9-10 k per net): a phase is the SUM of the two.  Here ONE wave per SIMD (256 threads, up to 512 registers) carries two
// tiles through every net with their instruction streams interleaved by construction (dense_fwd_b6p2): one tile's splits and layer
// boundaries ride in the other's matrix instructions, the weight operands are read from LDS once for both.  Same images, same rotation
// (three LDS slots, one per net), same per-accumulator term order as k_affine_chain (parity suite green with NF_CHAIN_DUAL=1).
template <class G>
__device__ __forceinline__ void net_forward_b6_dual(const float *__restrict__ img, const f32x16 (&x0)[G::MB], const f32x16 (&x1)[G::MB],
                                                    f32x16 (&out0)[G::CB], f32x16 (&out1)[G::CB], int l31, int hi) {
  using B = B6Geo<G>;
  const nf_u32x4 *w = reinterpret_cast<const nf_u32x4 *>(img);
  const float *bias = reinterpret_cast<const float *>(w + B::BIAS);
  f32x16 a10[G::H1B], a11[G::H1B], a20[G::H2B], a21[G::H2B];
  dense_fwd_b6p2<G::MB, G::H1B>(w + B::L1, bias + B::B1, x0, x1, a10, a11, l31, hi);
#pragma unroll
  for (int b = 0; b < G::H1B; ++b) { nf_lrelu16(a10[b]); nf_lrelu16(a11[b]); }
  dense_fwd_b6p2<G::H1B, G::H2B>(w + B::L2, bias + B::B2, a10, a11, a20, a21, l31, hi);
#pragma unroll
  for (int b = 0; b < G::H2B; ++b) { nf_lrelu16(a20[b]); nf_lrelu16(a21[b]); }
  dense_fwd_b6p2<G::H2B, G::CB>(w + B::L3, bias + B::B3, a20, a21, out0, out1, l31, hi);
}

template <class G, bool INVERSE>
__global__ __launch_bounds__(256, 1) void k_affine_chain_dual(ChainArgs a, float *xt, float *__restrict__ ladj) {
  static_assert(G::MB == G::CB, "parity blocks must have equal padded size");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  const long ngroups = (ntiles + 7) / 8;
  auto coupling_at = [&](int s) { return INVERSE ? s : a.ncoup - 1 - s; };
  using BG = B6Geo<G>;
  constexpr int B6F = BG::BYTES / 4;
  const int my_groups = (long)blockIdx.x < ngroups ? (int)((ngroups - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
  const int my_images = my_groups * 2 * a.ncoup;
  int req = 0, req_idx = 0, req_slot = 0;
  auto b6_request_next = [&]() {  // as k_affine_chain's: image i of the sequence s(c0), t(c0), s(c1), ... -> slot i mod 3 by LDS-DMA
    if (req >= my_images) return;
    typedef __attribute__((address_space(3))) void lds_void_t;
    const int kk = coupling_at(req_idx >> 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(a.wimg_b6) + (size_t)(2 * kk + (req_idx & 1)) * BG::BYTES, 0, BG::BYTES, 0x00020000);
    float *dstb = lds + req_slot * B6F;
    constexpr int NP = (BG::BYTES + 1023) / 1024;
    for (int p = wave; p < NP; p += 4)
      if (p * 1024 + lane * 16 < BG::BYTES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(dstb + p * 256), 16, lane * 16, p * 1024, 0, 0);
    ++req;
    req_idx = req_idx + 1 == 2 * a.ncoup ? 0 : req_idx + 1;
    req_slot = req_slot == 2 ? 0 : req_slot + 1;
  };
  int cur_slot = 0;
  b6_request_next();
  b6_request_next();
  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    f32x16 E[2][G::CB], O[2][G::MB];
    TileIO io[2];
    bool live[2], valid[2];
    long jj[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long tile = grp * 8 + 2 * wave + t;
      live[t] = tile < ntiles;  // wave-uniform
      const long tl = live[t] ? tile : 0;
      jj[t] = tl * NF_TILE + l31;
      valid[t] = live[t] && jj[t] < a.N;
      io[t] = make_tile_io(xt, tl, a.d, l31, hi);
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = tile_load(io[t], tile_soff(b, r, 0));  // features >= d read as 0
          const float o = tile_load(io[t], tile_soff(b, r, 1));
          E[t][b][r] = valid[t] ? e : 0.f;
          O[t][b][r] = valid[t] ? o : 0.f;
        }
    }
    float lsum[2] = {0.f, 0.f};
#pragma unroll 1
    for (int s = 0; s < a.ncoup; s += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        __syncthreads();    // this coupling's s image is complete, every wave is done with the image before it
        b6_request_next();  // the image after next, into the slot just vacated
        const int slot_t = cur_slot == 2 ? 0 : cur_slot + 1;
        const float *img_s = lds + cur_slot * B6F, *img_t = lds + slot_t * B6F;
        const bool x1_is_O = INVERSE ? (half == 1) : (half == 0);
        f32x16 (&x10)[G::CB] = x1_is_O ? O[0] : E[0];
        f32x16 (&x11)[G::CB] = x1_is_O ? O[1] : E[1];
        const f32x16 (&xb0)[G::MB] = x1_is_O ? E[0] : O[0];
        const f32x16 (&xb1)[G::MB] = x1_is_O ? E[1] : O[1];
        f32x16 S0[G::CB], S1[G::CB], T0[G::CB], T1[G::CB];
        float ls0 = 0.f, ls1 = 0.f;  // per coupling first, as coupling_step returns it: the same bits in ladj
        net_forward_b6_dual<G>(img_s, xb0, xb1, S0, S1, l31, hi);
        __syncthreads();  // the t image is complete, every wave is done with the s image
        b6_request_next();
        net_forward_b6_dual<G>(img_t, xb0, xb1, T0, T1, l31, hi);
#pragma unroll
        for (int b = 0; b < G::CB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            // rows >= c have zero weights and biases in the packed image: s = 0, T = 0, x1 stays 0
            const float s0 = nf_tanh(S0[b][r]), s1 = nf_tanh(S1[b][r]);
            if (INVERSE) {
              x10[b][r] = nf_fdiv(x10[b][r] - T0[b][r], nf_exp(s0));
              x11[b][r] = nf_fdiv(x11[b][r] - T1[b][r], nf_exp(s1));
            } else {
              x10[b][r] = x10[b][r] * nf_exp(s0) + T0[b][r];
              x11[b][r] = x11[b][r] * nf_exp(s1) + T1[b][r];
            }
            ls0 += s0;
            ls1 += s1;
          }
        lsum[0] += ls0;
        lsum[1] += ls1;
        cur_slot = slot_t == 2 ? 0 : slot_t + 1;
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!live[t]) continue;
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          tile_store(io[t], tile_soff(b, r, 0), E[t][b][r]);
          tile_store(io[t], tile_soff(b, r, 1), O[t][b][r]);
        }
      float ls = lsum[t];
      ls += __shfl_xor(ls, 32);
      if (hi == 0 && valid[t]) ladj[jj[t]] = INVERSE ? -ls : ls;
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
  constexpr int S = 32 * OB + NF_IMG_PAD;
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

#ifdef NF_KERNEL_TRACE
#define NF_TS_STAMP(slot)                                                  \
  do {                                                                     \
    if (tr) { __builtin_amdgcn_sched_barrier(0); tr[slot] = clock64(); }   \
  } while (0)
#define NF_TSB(slot)                                                        \
  do {                                                                      \
    if (trb) { __builtin_amdgcn_sched_barrier(0); trb[slot] = clock64(); }  \
  } while (0)
#else
#define NF_TS_STAMP(slot) do { (void)tr; } while (0)
#define NF_TSB(slot) do { } while (0)
#endif

// Sign masks of the hidden pre-activations (bit r set <=> z[r] has its sign bit set, i.e. the
// leaky-ReLU slope is 0.01): all the reverse pass needs of a1/a2 besides their LDS copies, so the
// activations themselves can die early.  leakyrelu keeps the sign, so the mask is taken from the
// post-activation value.  z = +0 counts as slope 1 where the oracle uses 0.01; the zero-padded rows
// (z = 0 exactly) carry a zero cotangent, so the two agree.  Built by nf_sign_mask16 (nf_mfma.h), one
// v_alignbit per element.  (Note for maintainers: __builtin_bit_cast applied directly to a
// vector-element expression such as v[b][r] reads element 0 under hipcc 7.2 -- copy to a scalar first.)
template <int NB>
__device__ __forceinline__ void sign_masks(const f32x16 (&v)[NB], unsigned (&m)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) m[b] = nf_sign_mask16(v[b]);
}
__device__ __forceinline__ float lrelu_slope(unsigned mask, int r) { return nf_mask_slope(mask, r); }
template <int NB>
__device__ __forceinline__ void apply_lrelu_grad(f32x16 (&d)[NB], const unsigned (&m)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) nf_lrelu_grad16(d[b], m[b]);
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

// INVD: reverse pass of the INVERSE coupling (forward-KL training) at its output w: `y` holds w and is
// advanced to coupling(w), `ybar` holds the cotangent of w and leaves that of the inverse's input.
//   w1 = (v1 - t) exp(-s), ladj_inv = -sum s:   v1bar = w1bar exp(-s),  sbar = -(w1bar w1 + lbar),  tbar = -v1bar
// Phase S runs first there (it needs w1bar and w1), then phase T.
template <class G, bool PHASE_S, bool FULL, bool INVD = false>
__device__ __forceinline__ void bwd_tile(const CouplingArgs &a, const float *__restrict__ img, float *__restrict__ sc,
                                         BwdAcc<G> &acc, float *__restrict__ y, float *__restrict__ ybar,
                                         const float *__restrict__ lbar, float lbar_const, long tile, int l31, int hi,
                                         long long *tr) {
  using L = BwdLds<G>;
  NF_TS_STAMP(0);
  const long j = tile * NF_TILE + l31;
  // FULL: N is a multiple of the tile, so no sample mask is needed (feature bounds are handled by
  // the buffer descriptors either way).
  const bool valid = FULL ? true : j < a.N;
  const int par_c = 1 - a.par_t;
  const TileIO yio = make_tile_io(y, tile, a.d, l31, hi);
  const TileIO gio = make_tile_io(ybar, tile, a.d, l31, hi);
  float *sd = sc + L::OFF_D;

  // Every LDS stash write / activation-derivative scaling / tile store below is a SIDE JOB of the
  // neighbouring GEMM (nf_mfma.h): program order is
  //   L1 [stash x2] -> L2 [stash a1] -> L3 [stash a2] -> element-wise ->
  //   dX3 [stash d3] -> dW3 [d2 *= lrelu'] -> dX2 [stash d2] -> dW2 [d1 *= lrelu'] ->
  //   dX1 [stash d1] -> dW1 [x2bar stores]
  f32x16 d3[G::CB], y1[G::CB], g1[G::CB];
  unsigned m1[G::H1B], m2[G::H2B];
  {
    f32x16 xb[G::MB];
#pragma unroll
    for (int b = 0; b < G::MB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = tile_load(yio, tile_soff(b, r, par_c));  // features >= d read as 0
        xb[b][r] = valid ? v : 0.f;
      }
    NF_TS_STAMP(1);
    f32x16 a1[G::H1B];
    dense_fwd<G::MB, G::H1B>(img + G::W1, img + G::B1, xb, a1, l31, hi,
                             [&](int e) { scratch_put<G::MB>(sc + L::OFF_X, xb, e, l31, hi); });
#pragma unroll
    for (int b = 0; b < G::H1B; ++b)
      nf_lrelu16(a1[b]);
    sign_masks<G::H1B>(a1, m1);
    f32x16 a2[G::H2B];
    dense_fwd<G::H1B, G::H2B>(img + G::W2, img + G::B2, a1, a2, l31, hi,
                              [&](int e) { scratch_put<G::H1B>(sc + L::OFF_A1, a1, e, l31, hi); });
#pragma unroll
    for (int b = 0; b < G::H2B; ++b)
      nf_lrelu16(a2[b]);
    sign_masks<G::H2B>(a2, m2);
    // operands of the element-wise stage: issued here, consumed after the last forward layer
#pragma unroll
    for (int b = 0; b < G::CB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        y1[b][r] = tile_load(yio, tile_soff(b, r, a.par_t));
        g1[b][r] = tile_load(gio, tile_soff(b, r, a.par_t));
      }
    dense_fwd<G::H2B, G::CB>(img + G::W3, img + G::B3, a2, d3, l31, hi,
                             [&](int e) { scratch_put<G::H2B>(sc + L::OFF_A2, a2, e, l31, hi); });
  }
  NF_TS_STAMP(2);

  const float lb = valid ? (lbar ? lbar[FULL ? j : (j < a.N ? j : 0)] : lbar_const) : 0.f;
#pragma unroll
  for (int b = 0; b < G::CB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = b * 32 + nf_row(r, hi);
      const bool ok = (p < a.c) && valid;  // rows >= c: the loads returned 0, the stores are dropped
      const float yv = y1[b][r], gv = g1[b][r];
      if (INVD && !PHASE_S) {
        tile_store(yio, tile_soff(b, r, a.par_t), yv + d3[b][r]);  // v1 = w1 exp(s) + t
        d3[b][r] = ok ? -gv : 0.f;                                  // T-bar = -v1bar
      } else if (INVD) {
        const float s = nf_tanh(d3[b][r]);
        const float es = nf_exp(s);
        tile_store(yio, tile_soff(b, r, a.par_t), yv * es);            // w1 exp(s)
        tile_store(gio, tile_soff(b, r, a.par_t), nf_fdiv(gv, es));    // v1bar
        d3[b][r] = ok ? -(gv * yv + lb) * (1.f - s * s) : 0.f;         // S-bar through tanh
      } else if (!PHASE_S) {
        tile_store(yio, tile_soff(b, r, a.par_t), yv - d3[b][r]);  // u = x1 * exp(S)
        d3[b][r] = ok ? gv : 0.f;                                   // T-bar = ybar1
      } else {
        const float s = nf_tanh(d3[b][r]);
        const float es = nf_exp(s);
        tile_store(yio, tile_soff(b, r, a.par_t), nf_fdiv(yv, es));  // x1 = u * exp(-s)
        tile_store(gio, tile_soff(b, r, a.par_t), gv * es);             // x1bar
        d3[b][r] = ok ? (gv * yv + lb) * (1.f - s * s) : 0.f;           // S-bar through tanh
      }
    }
  NF_TS_STAMP(3);

  // ---- layer 3
  f32x16 d2[G::H2B];
  dense_bwd_x<G::H2B, G::CB>(img + G::W3, d3, d2, l31, hi, [&](int e) { scratch_put<G::CB>(sd, d3, e, l31, hi); });
  NF_TS_STAMP(4);
  wave_lds_fence();
  dw_accumulate<G::H2B, G::CB>(sc + L::OFF_A2, sd, acc.w3, acc.b3, l31, hi, [&](int e) {
    if (e < G::H2B * 16) d2[e >> 4][e & 15] *= lrelu_slope(m2[e >> 4], e & 15);
  });
  NF_TS_STAMP(5);
  wave_lds_fence();
  // ---- layer 2
  f32x16 d1[G::H1B];
  dense_bwd_x<G::H1B, G::H2B>(img + G::W2, d2, d1, l31, hi, [&](int e) { scratch_put<G::H2B>(sd, d2, e, l31, hi); });
  NF_TS_STAMP(6);
  wave_lds_fence();
  dw_accumulate<G::H1B, G::H2B>(sc + L::OFF_A1, sd, acc.w2, acc.b2, l31, hi, [&](int e) {
    if (e < G::H1B * 16) d1[e >> 4][e & 15] *= lrelu_slope(m1[e >> 4], e & 15);
  });
  NF_TS_STAMP(7);
  wave_lds_fence();
  // ---- layer 1: x2bar accumulates ybar2 + W1t^T d1t (phase T) + W1s^T d1s (phase S)
  f32x16 g2[G::MB], gold[G::MB];
#pragma unroll
  for (int b = 0; b < G::MB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) gold[b][r] = tile_load(gio, tile_soff(b, r, par_c));
  dense_bwd_x<G::MB, G::H1B>(img + G::W1, d1, g2, l31, hi, [&](int e) { scratch_put<G::H1B>(sd, d1, e, l31, hi); });
  NF_TS_STAMP(8);
  wave_lds_fence();
  dw_accumulate<G::MB, G::H1B>(sc + L::OFF_X, sd, acc.w1, acc.b1, l31, hi, [&](int e) {
    if (e < G::MB * 16) tile_store(gio, tile_soff(e >> 4, e & 15, par_c), gold[e >> 4][e & 15] + g2[e >> 4][e & 15]);
  });
  NF_TS_STAMP(9);
  wave_lds_fence();
  NF_TS_STAMP(10);
}

// reverse pass of ONE coupling by this workgroup (both phases, fold, slab write)
template <class G, bool FULL, bool INVD = false>
__device__ __forceinline__ void bwd_coupling(const CouplingArgs &a, float *__restrict__ y, float *__restrict__ ybar,
                                             const float *__restrict__ lbar, float lbar_const,
                                             float *__restrict__ slab, long slab_stride, float *lds) {
  float *img = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int l31 = lane & 31, hi = lane >> 5;
  float *sc = lds + G::SIZE + wave * BwdLds<G>::SCRATCH;
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  long long *tr0 = (a.trace && blockIdx.x == 0 && tid == 0) ? a.trace : nullptr;
  if (tr0) tr0[0] = clock64();

#pragma unroll 1
  for (int phase = 0; phase < 2; ++phase) {
    long long *tr = tr0 ? tr0 + 8 + phase * 40 : nullptr;  // [8 + phase*40 + tileidx*12 + slot]
    const bool is_s = INVD ? phase == 0 : phase == 1;      // forward chain: T then S; inverse chain: S then T
    stage_packed<G::SIZE, 256>(img, is_s ? a.img_s : a.img_t, tid);
    __syncthreads();
    if (tr0) tr0[1 + phase * 3] = clock64();
    BwdAcc<G> acc;
    zero_acc(acc.w1, acc.b1);
    zero_acc(acc.w2, acc.b2);
    zero_acc(acc.w3, acc.b3);
#pragma unroll 1
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
      if (!is_s)
        bwd_tile<G, false, FULL, INVD>(a, img, sc, acc, y, ybar, lbar, lbar_const, tile, l31, hi, tr);
      else
        bwd_tile<G, true, FULL, INVD>(a, img, sc, acc, y, ybar, lbar, lbar_const, tile, l31, hi, tr);
      if (tr) tr += 12;
    }
    if (tr0) tr0[2 + phase * 3] = clock64();
    __syncthreads();  // every wave is done with the weight image and its scratch
    // Each wave drops its accumulators, in image layout, into its own G::SIZE-float region of the
    // (now free) LDS; then all threads add the four copies and write the workgroup's slab with
    // 16-byte stores.  Slabs keep the padded image layout; k_reduce_image_slabs maps them to theta.
    {
      float *mine = lds + wave * G::SIZE;
      fold_acc(mine + G::W1, mine + G::B1, acc.w1, acc.b1, true, l31, hi);
      fold_acc(mine + G::W2, mine + G::B2, acc.w2, acc.b2, true, l31, hi);
      fold_acc(mine + G::W3, mine + G::B3, acc.w3, acc.b3, true, l31, hi);
    }
    __syncthreads();
    {
      const float4 *c0 = reinterpret_cast<const float4 *>(lds);
      float4 *dst = reinterpret_cast<float4 *>(slab + ((long)blockIdx.x * slab_stride + (is_s ? 0 : 1) * (long)G::SIZE));
      constexpr int NV4 = G::SIZE / 4;
      for (int i = tid; i < NV4; i += 256) {
        const float4 p0 = c0[i], p1 = c0[i + NV4], p2 = c0[i + 2 * NV4], p3 = c0[i + 3 * NV4];
        float4 r;
        r.x = (p0.x + p1.x) + (p2.x + p3.x);
        r.y = (p0.y + p1.y) + (p2.y + p3.y);
        r.z = (p0.z + p1.z) + (p2.z + p3.z);
        r.w = (p0.w + p1.w) + (p2.w + p3.w);
        dst[i] = r;
      }
    }
    __syncthreads();  // image is restaged next phase
    if (tr0) tr0[3 + phase * 3] = clock64();
  }
}

template <class G, bool FULL>
__global__ __launch_bounds__(256, 1) void k_affine_bwd(CouplingArgs a, float *__restrict__ y, float *__restrict__ ybar,
                                                       const float *__restrict__ lbar, float lbar_const,
                                                       float *__restrict__ slab, long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bwd_coupling<G, FULL>(a, y, ybar, lbar, lbar_const, slab, slab_stride, lds);
}

// The whole chain's reverse pass in ONE launch.  A wave's tiles never change hands, and coupling k+1
// only reads what the same wave wrote for coupling k (y <- x, ybar <- xbar), so there is no
// cross-workgroup dependency: each workgroup simply walks the couplings in flat order.  Saves the launch
// gap and the ramp-up / tail of seven launches per step.
struct BwdAllArgs {
  const float *wimg;  // [coupling][s|t][G::SIZE]
  const unsigned char *wimg_b6t;  // [coupling][s|t][B6TGeo<G>::BYTES]: the bf16-triple images of the dX GEMMs (pair kernel, PB6)
  long long *trace;
  int d, ncoup;
  long N;
};
template <class G, bool FULL, bool INVD>
__global__ __launch_bounds__(256, 1) void k_affine_bwd_all(BwdAllArgs aa, float *__restrict__ y, float *__restrict__ ybar,
                                                           const float *__restrict__ lbar, float lbar_const,
                                                           float *__restrict__ slab, long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#pragma unroll 1
  for (int step = 0; step < aa.ncoup; ++step) {
    const int k = INVD ? aa.ncoup - 1 - step : step;  // the inverse chain's reverse pass runs in execution order
    CouplingArgs a;
    a.theta = nullptr;
    a.img_s = aa.wimg + (size_t)(2 * k) * G::SIZE;
    a.img_t = a.img_s + G::SIZE;
    a.trace = k == 0 ? aa.trace : nullptr;
    a.d = aa.d;
    a.par_t = k & 1;
    a.c = (k & 1) ? aa.d / 2 : (aa.d + 1) / 2;
    a.m = aa.d - a.c;
    a.N = aa.N;
    bwd_coupling<G, FULL, INVD>(a, y, ybar, lbar, lbar_const, slab + (long)k * 2 * G::SIZE, slab_stride, lds);
    // What coupling k + 1 loads are this same wave's stores of coupling k.  They are ordered by the
    // barrier at the end of bwd_coupling (s_waitcnt vmcnt(0): the stores have reached L2) and the vector
    // L1 is write-through, exactly as between the two phases inside one coupling -- no agent-scope fence:
    // on this part that would write back / invalidate the XCD's L2 (measured: 90 instead of 74 us per coupling).
  }
}

// ------------------------------------------------------------------------------------
// reverse pass of the training step from the forward's stash: no recompute
// ------------------------------------------------------------------------------------
// Same mathematics and phase structure as bwd_tile / k_affine_bwd_all (phase T then phase S per coupling, one net in
// LDS, its dW accumulators in registers), but x2 / a1 / a2 arrive from the stash already in the dW operand layout
// (registers, no LDS transpose), s / u / masks arrive in the C layout, and nothing of the forward is recomputed:
// per tile-phase 256 MFMAs instead of 384, the flow state `y` is neither read nor written (it stays the flow output).
template <int NB>
__device__ __forceinline__ void stash_get_T(const StashIO &st, int base, int voff, float (&at)[NB][16]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(st.rs, voff, (base + b * 1024) * 4 + q * 16, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned bits = w[e];
        at[b][4 * q + e] = __builtin_bit_cast(float, bits);
      }
    }
}
template <int NB>
__device__ __forceinline__ void stash_get_lane(const StashIO &st, int base, f32x16 (&v)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(st.rs, st.vL, (base + b * 1024) * 4 + q * 16, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned bits = w[e];
        v[b][4 * q + e] = __builtin_bit_cast(float, bits);
      }
    }
}

// dW^T accumulation with the activation operand in registers: at[ib][t] = a[feature ib*32 + l31][sample 2t + hi]
// CORDER: k-step t contracts the sample (t & 3) + 8 (t >> 2) + 4 hi instead -- the order in which the MFMA C layout holds
// the rows of a recomputed operand (recompute_a1t); the delta tile is read in the same order, the sum is the same sum.
template <bool CORDER>
__device__ __forceinline__ constexpr int dw_sample(int t) { return CORDER ? (t & 3) + 8 * (t >> 2) : 2 * t; }
template <int IB, int OB, class SJ = NoSideJob, bool CORDER = false>
__device__ __forceinline__ void dw_accumulate_reg(const float (&at)[IB][16], const float *__restrict__ sd,
                                                  f32x16 (&acc)[IB][OB], float (&bsum)[OB], int l31, int hi, SJ sj = SJ()) {
  constexpr int TG = 2, NG = 16 / TG;
  const float *pd = sd + l31 * NF_TS + (CORDER ? 4 * hi : hi);
  float dn[TG][OB], dc[TG][OB];
#pragma unroll
  for (int u = 0; u < TG; ++u)
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) dn[u][ob] = pd[ob * 32 * NF_TS + dw_sample<CORDER>(u)];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int u = 0; u < TG; ++u)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) dc[u][ob] = dn[u][ob];
    if (g + 1 < NG) {
#pragma unroll
      for (int u = 0; u < TG; ++u)
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) dn[u][ob] = pd[ob * 32 * NF_TS + dw_sample<CORDER>((g + 1) * TG + u)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < TG; ++u) {
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) bsum[ob] += dc[u][ob];
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
          acc[ib][ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[ib][g * TG + u], dc[u][ob], acc[ib][ob], 0, 0, 0);
          sj(((g * TG + u) * IB + ib) * OB + ob);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dW^T accumulation on the bf16 matrix cores (six-term products, nf_mfma.h "B6"): same operands as dw_accumulate_reg -- the
// activation operand in registers, at[ib][t] = a[feature][sample 2t + hi], the delta operand from the LDS tile at the same
// samples -- with the sample <-> k-slot assignment  slot (g, hi, j) <-> sample 2 (8 g + j) + hi : registers 8g .. 8g+7 of `at`
// ARE the lane's eight k-values of group g, and the delta reads are the ones the fp32 form issues.  Both operands are split
// here (8 values -> 3 x 4 registers each); 12 bf16 MFMAs of 32 clocks replace 16 fp32 ones of 64 per (ib, ob) block pair.
// The bias gradient stays the fp32 sum of the deltas.
// The activation operand arrives already split (split_T, done by the consumer while it waits for the producer's delta tile:
// 48 registers per 32-feature block pair instead of 32 fp32 + the splits of the group in flight); one delta block's eight
// values and their split are live at a time, the next unit's delta values are requested behind the current unit's MFMAs.
// Pipeline unit = (sample group g, delta block ob).
template <int IB, int OB, class SJ = NoSideJob>
__device__ __forceinline__ void dw_accumulate_reg_b6(const SplitT<IB> &as, const float *__restrict__ sd, f32x16 (&acc)[IB][OB],
                                                     float (&bsum)[OB], int l31, int hi, SJ sj = SJ()) {
  const float *pd = sd + l31 * NF_TS + hi;
  constexpr int NU = 2 * OB;
  float dn[8], dc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dn[j] = pd[2 * j];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int g = u / OB, ob = u % OB;
#pragma unroll
    for (int j = 0; j < 8; ++j) dc[j] = dn[j];
    if (u + 1 < NU) {
      const int g1 = (u + 1) / OB, ob1 = (u + 1) % OB;
#pragma unroll
      for (int j = 0; j < 8; ++j) dn[j] = pd[ob1 * 32 * NF_TS + 2 * (8 * g1 + j)];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[ob] += dc[j];
    nf_u32x4 dh, dm, dl;
    nf_split8(dc, dh, dm, dl);
    __builtin_amdgcn_sched_barrier(0);
    // smallest terms first (as dense_fwd_b6): al dh, ah dl, am dm, am dh, ah dm, ah dh -- the IB accumulators interleaved
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) {
        const nf_u32x4 &a = term == 0 ? as.l[ib][g] : (term == 2 || term == 3) ? as.m[ib][g] : as.h[ib][g];
        const nf_u32x4 &d = term == 1 ? dl : (term == 2 || term == 4) ? dm : dh;
        acc[ib][ob] = nf_mfma_bf16(a, d, acc[ib][ob]);
        const int c = (u * 6 + term) * IB + ib;  // running MFMA index: two side-job slots each
        sj(2 * c);
        sj(2 * c + 1);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// a1^T = leakyrelu(W1 x2 + b1) as the dW2 GEMM wants it -- lane <-> hidden unit, register r <-> sample
// (r & 3) + 8 (r >> 2) + 4 hi (dw_accumulate_reg<..., CORDER>) -- from the conditioner input in the MFMA C layout:
// D' = x2^T W1^T, i.e. dense_fwd with the operand roles swapped (A: the input's registers, B: the weight image's rows, lanes
// along a row as in the forward fetch).  Same k order and operand values as the forward's own first layer.
template <class G>
__device__ __forceinline__ void recompute_a1t(const float *__restrict__ img, const f32x16 (&x2c)[G::MB], float (&a1t)[G::H1B][16],
                                              int l31, int hi) {
  constexpr int NG = G::MB * 4, S = G::S1, OB = G::H1B;
  f32x16 z[OB];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) {
    const float bias = img[G::B1 + ob * 32 + l31];
#pragma unroll
    for (int r = 0; r < 16; ++r) z[ob][r] = bias;
  }
  const float *wl = img + G::W1 + (4 * hi) * S + l31;
  float wn[OB][4], wc[OB][4];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int e = 0; e < 4; ++e) wn[ob][e] = wl[e * S + ob * 32];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
      for (int e = 0; e < 4; ++e) wc[ob][e] = wn[ob][e];
    if (g + 1 < NG) {
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int e = 0; e < 4; ++e) wn[ob][e] = wl[((g + 1) * 8 + e) * S + ob * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
        z[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(x2c[g / 4][(g % 4) * 4 + e], wc[ob][e], z[ob], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) {
    nf_lrelu16(z[ob]);
#pragma unroll
    for (int r = 0; r < 16; ++r) a1t[ob][r] = z[ob][r];
  }
}

template <class G>
struct BwdStashLds {
  static constexpr int DROWS = BwdLds<G>::DROWS;
  static constexpr int SCRATCH = DROWS * 32 * NF_TS;  // one delta tile per wave
  static constexpr int WAVES = 4;
  static constexpr int IMGMAX = B6TGeo<G>::BYTES / 4 > G::SIZE ? B6TGeo<G>::BYTES / 4 : G::SIZE;  // fp32 or B6T image
  static constexpr int FLOATS = (IMGMAX + WAVES * SCRATCH) > WAVES * G::SIZE ? (IMGMAX + WAVES * SCRATCH) : WAVES * G::SIZE;
  static constexpr size_t BYTES = (size_t)FLOATS * sizeof(float);  // the fold at the end of a phase needs 4 images
};

// s and u of a phase-S tile are loaded one tile ahead, into the same registers, while the previous tile runs its last
// two GEMMs: every wave of the chip asks for its tile's operands at the same moment (25 MB at once), and without the
// head start the element-wise stage waits 3.5-6.7 k cycles for them (tools/trace_bwd_stashed.py).
// (Round 4, measured and removed: the tile's two cotangent halves requested the same way in the pair kernel's phase S -- 32
// more registers in the producer, no spills, 339-341 us per launch against 338-342 without, and the inverse-direction kernel
// 1 944 us per chunk against 1 680.)
template <class G>
struct StashFirst {
  f32x16 sv[G::CB], uv[G::CB];
};
template <class G, bool SLIM = false>
__device__ __forceinline__ void stash_issue_first(StashFirst<G> &f, float *stash, int k, int ncoup, long tile, int l31, int hi) {
  using SG = StashGeo<G, SLIM>;
  const StashIO st = make_stash_io(stash, tile * ncoup + k, SG::SIZE, true, l31, hi);
  stash_get_lane<G::CB>(st, SG::SV, f.sv);
  stash_get_lane<G::CB>(st, SG::UV, f.uv);
}

// INVD: reverse pass of the INVERSE coupling (forward-KL training; algebra as in bwd_tile): the UV slot holds w1, phase S
// runs first (it needs w1bar and w1 and leaves v1bar = w1bar exp(-s) behind), phase T seeds with -v1bar.
// SB6 (round 4): all six GEMMs of the tile on the bf16 matrix cores (dense_bwd_x_b6 / dw_accumulate_reg_b6); `img` is then
// the net's B6T image.  One wave per SIMD: 512 registers, the split operands fit next to the 128 accumulators here.
template <class G, bool PHASE_S, bool FULL, bool INVD = false, bool SB6 = false>
__device__ __forceinline__ void bwd_tile_stashed(const CouplingArgs &a, const float *__restrict__ img, float *__restrict__ sd,
                                                 BwdAcc<G> &acc, StashFirst<G> &f, float *stash, int k, int ncoup,
                                                 float *__restrict__ ybar, const float *__restrict__ lbar, float lbar_const,
                                                 long tile, long next_tile, int l31, int hi, long long *tr = nullptr) {
  NF_TS_STAMP(0);
  using SG = StashGeo<G>;
  const long j = tile * NF_TILE + l31;
  const bool valid = FULL ? true : j < a.N;
  const int par_c = 1 - a.par_t;
  const TileIO gio = make_tile_io(ybar, tile, a.d, l31, hi);
  const StashIO st = make_stash_io(stash, tile * ncoup + k, SG::SIZE, true, l31, hi);
  constexpr int nbase = SG::NET0 + (PHASE_S ? 0 : SG::NETSZ);
  const int vT = (l31 * 32 + hi * 16) * 4;  // this lane's feature row and sample parity in a T-layout block

  // loads in the order of first use
  f32x16 g1[G::CB];
#pragma unroll
  for (int b = 0; b < G::CB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) g1[b][r] = tile_load(gio, tile_soff(b, r, a.par_t));
  const u32x4 mk = __builtin_amdgcn_raw_buffer_load_b128(st.rs, (hi * 32 + l31) * 16, (nbase + SG::MSK) * 4, 0);
  float a2t[G::H2B][16];
  stash_get_T<G::H2B>(st, nbase + SG::A2, vT, a2t);

  NF_TS_STAMP(1);
  const float lb = valid ? (lbar ? lbar[FULL ? j : (j < a.N ? j : 0)] : lbar_const) : 0.f;
  f32x16 d3[G::CB];
#pragma unroll
  for (int b = 0; b < G::CB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = b * 32 + nf_row(r, hi);
      const bool ok = (p < a.c) && valid;
      const float gv = g1[b][r];
      if (!PHASE_S) {
        d3[b][r] = ok ? (INVD ? -gv : gv) : 0.f;  // T-bar = ybar1 (inverse: -v1bar)
      } else if (INVD) {
        const float s = f.sv[b][r];
        tile_store(gio, tile_soff(b, r, a.par_t), nf_fdiv(gv, nf_exp(s)));  // v1bar
        d3[b][r] = ok ? -(gv * f.uv[b][r] + lb) * (1.f - s * s) : 0.f;     // S-bar through tanh (uv = w1)
      } else {
        const float s = f.sv[b][r];
        tile_store(gio, tile_soff(b, r, a.par_t), gv * nf_exp(s));  // x1bar
        d3[b][r] = ok ? (gv * f.uv[b][r] + lb) * (1.f - s * s) : 0.f;  // S-bar through tanh
      }
    }
  const unsigned mk0 = mk[0], mk1 = mk[1], mk2 = mk[2], mk3 = mk[3];
  const unsigned m1[2] = {mk0, mk1}, m2[2] = {mk2, mk3};

  NF_TS_STAMP(2);
  // ---- layer 3
  f32x16 d2[G::H2B];
  const nf_u32x4 *wt = reinterpret_cast<const nf_u32x4 *>(img);  // SB6: the staged image is the net's B6T image
  if constexpr (SB6) dense_bwd_x_b6<G::H2B, G::CB>(wt + B6TGeo<G>::T3, d3, d2, l31, hi, [&](int e) { scratch_put<G::CB>(sd, d3, e, l31, hi); });
  else dense_bwd_x<G::H2B, G::CB>(img + G::W3, d3, d2, l31, hi, [&](int e) { scratch_put<G::CB>(sd, d3, e, l31, hi); });
  float a1t[G::H1B][16];
  stash_get_T<G::H1B>(st, nbase + SG::A1, vT, a1t);
  NF_TS_STAMP(3);
  wave_lds_fence();
  auto sj3 = [&](int e) {
    if (e < G::H2B * 16) d2[e >> 4][e & 15] *= lrelu_slope(m2[e >> 4], e & 15);
  };
  if constexpr (SB6) {
    SplitT<G::H2B> a2s;
    split_T<G::H2B>(a2t, a2s);
    dw_accumulate_reg_b6<G::H2B, G::CB>(a2s, sd, acc.w3, acc.b3, l31, hi, sj3);
  } else
    dw_accumulate_reg<G::H2B, G::CB>(a2t, sd, acc.w3, acc.b3, l31, hi, sj3);
  NF_TS_STAMP(4);
  wave_lds_fence();
  // ---- layer 2
  f32x16 d1[G::H1B];
  if constexpr (SB6) dense_bwd_x_b6<G::H1B, G::H2B>(wt + B6TGeo<G>::T2, d2, d1, l31, hi, [&](int e) { scratch_put<G::H2B>(sd, d2, e, l31, hi); });
  else dense_bwd_x<G::H1B, G::H2B>(img + G::W2, d2, d1, l31, hi, [&](int e) { scratch_put<G::H2B>(sd, d2, e, l31, hi); });
  float x2t[G::MB][16];
  stash_get_T<G::MB>(st, SG::XT, vT, x2t);
  NF_TS_STAMP(5);
  wave_lds_fence();
  auto sj2 = [&](int e) {
    if (e < G::H1B * 16) d1[e >> 4][e & 15] *= lrelu_slope(m1[e >> 4], e & 15);
  };
  if constexpr (SB6) {
    SplitT<G::H1B> a1s;
    split_T<G::H1B>(a1t, a1s);
    dw_accumulate_reg_b6<G::H1B, G::H2B>(a1s, sd, acc.w2, acc.b2, l31, hi, sj2);
  } else
    dw_accumulate_reg<G::H1B, G::H2B>(a1t, sd, acc.w2, acc.b2, l31, hi, sj2);
  NF_TS_STAMP(6);
  wave_lds_fence();
  // ---- layer 1: x2bar accumulates ybar2 + W1t^T d1t (phase T) + W1s^T d1s (phase S)
  f32x16 g2[G::MB], gold[G::MB];
#pragma unroll
  for (int b = 0; b < G::MB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) gold[b][r] = tile_load(gio, tile_soff(b, r, par_c));
  if (PHASE_S && next_tile >= 0) stash_issue_first<G>(f, stash, k, ncoup, next_tile, l31, hi);
  if constexpr (SB6) dense_bwd_x_b6<G::MB, G::H1B>(wt + B6TGeo<G>::T1, d1, g2, l31, hi, [&](int e) { scratch_put<G::H1B>(sd, d1, e, l31, hi); });
  else dense_bwd_x<G::MB, G::H1B>(img + G::W1, d1, g2, l31, hi, [&](int e) { scratch_put<G::H1B>(sd, d1, e, l31, hi); });
  NF_TS_STAMP(7);
  wave_lds_fence();
  auto sj1 = [&](int e) {
    if (e < G::MB * 16) tile_store(gio, tile_soff(e >> 4, e & 15, par_c), gold[e >> 4][e & 15] + g2[e >> 4][e & 15]);
  };
  if constexpr (SB6) {
    SplitT<G::MB> x2s;
    split_T<G::MB>(x2t, x2s);
    dw_accumulate_reg_b6<G::MB, G::H1B>(x2s, sd, acc.w1, acc.b1, l31, hi, sj1);
  } else
    dw_accumulate_reg<G::MB, G::H1B>(x2t, sd, acc.w1, acc.b1, l31, hi, sj1);
  wave_lds_fence();
  NF_TS_STAMP(8);
}

template <class G, bool FULL, bool INVD = false, bool SB6 = false>
__global__ __launch_bounds__(256, 1) void k_affine_bwd_stashed(BwdAllArgs aa, float *stash, float *__restrict__ ybar,
                                                               const float *__restrict__ lbar, float lbar_const,
                                                               float *__restrict__ slab, long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *img = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  constexpr int IMGF = SB6 ? B6TGeo<G>::BYTES / 4 : G::SIZE;  // floats of the staged weight image (fp32, or B6T)
  float *sd = lds + IMGF + wave * BwdStashLds<G>::SCRATCH;
  const long ntiles = (aa.N + NF_TILE - 1) / NF_TILE;
  const long tile0 = (long)blockIdx.x * 4 + wave, tstride = (long)gridDim.x * 4;
  StashFirst<G> f;
  // clock stamps of block 0 / thread 0 for tools/trace_bwd_stashed.py (first coupling processed): [0] start;
  // per phase p: [1+4p] image staged, [2+4p] tiles done, [3+4p] every wave done, [4+4p] folded + slab written;
  // [32 + 24p + 12i + 0..8]: stages of tile i (< 2) of phase p
#ifdef NF_KERNEL_TRACE
  long long *tr0 = (aa.trace && blockIdx.x == 0 && tid == 0) ? aa.trace : nullptr;
  if (tr0) tr0[0] = clock64();
#define NF_ST_STAMP(slot) do { if (tr0 && step == 0) { __builtin_amdgcn_sched_barrier(0); tr0[slot] = clock64(); } } while (0)
#else
#define NF_ST_STAMP(slot) do { } while (0)
#endif
  if (INVD && tile0 < ntiles) stash_issue_first<G>(f, stash, aa.ncoup - 1, aa.ncoup, tile0, l31, hi);  // S runs first
#pragma unroll 1
  for (int step = 0; step < aa.ncoup; ++step) {
    const int k = INVD ? aa.ncoup - 1 - step : step;  // the inverse chain's reverse pass runs in execution order
    CouplingArgs a;
    a.theta = nullptr;
    a.img_s = aa.wimg + (size_t)(2 * k) * G::SIZE;
    a.img_t = a.img_s + G::SIZE;
    a.trace = nullptr;
    a.d = aa.d;
    a.par_t = k & 1;
    a.c = (k & 1) ? aa.d / 2 : (aa.d + 1) / 2;
    a.m = aa.d - a.c;
    a.N = aa.N;
    float *kslab = slab + (long)k * 2 * G::SIZE;
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
      const bool is_s = INVD ? phase == 0 : phase == 1;
      if constexpr (SB6)
        stage_packed<IMGF, 256>(img, reinterpret_cast<const float *>(aa.wimg_b6t + (size_t)(2 * k + (is_s ? 0 : 1)) * B6TGeo<G>::BYTES), tid);
      else
        stage_packed<G::SIZE, 256>(img, is_s ? a.img_s : a.img_t, tid);
      __syncthreads();
      BwdAcc<G> acc;
      zero_acc(acc.w1, acc.b1);
      zero_acc(acc.w2, acc.b2);
      zero_acc(acc.w3, acc.b3);
      NF_ST_STAMP(1 + phase * 4);
#pragma unroll 1
      for (long tile = tile0; tile < ntiles; tile += tstride) {
        const long nt = tile + tstride < ntiles ? tile + tstride : -1;
#ifdef NF_KERNEL_TRACE
        const long ti = (tile - tile0) / tstride;
        long long *tr = (tr0 && step == 0 && ti < 2) ? tr0 + 32 + phase * 24 + ti * 12 : nullptr;
#else
        long long *tr = nullptr;
#endif
        if (!is_s) bwd_tile_stashed<G, false, FULL, INVD, SB6>(a, img, sd, acc, f, stash, k, aa.ncoup, ybar, lbar, lbar_const, tile, nt, l31, hi, tr);
        else bwd_tile_stashed<G, true, FULL, INVD, SB6>(a, img, sd, acc, f, stash, k, aa.ncoup, ybar, lbar, lbar_const, tile, nt, l31, hi, tr);
      }
      NF_ST_STAMP(2 + phase * 4);
      __syncthreads();  // every wave is done with the weight image and its scratch
      NF_ST_STAMP(3 + phase * 4);
      // s and u of the next phase-S's first tile fly behind the fold, the slab write and the staging of the next image
      if (tile0 < ntiles) {
        if (!INVD && !is_s) stash_issue_first<G>(f, stash, k, aa.ncoup, tile0, l31, hi);
        if (INVD && !is_s && step + 1 < aa.ncoup) stash_issue_first<G>(f, stash, k - 1, aa.ncoup, tile0, l31, hi);
      }
      {
        float *mine = lds + wave * G::SIZE;
        fold_acc(mine + G::W1, mine + G::B1, acc.w1, acc.b1, true, l31, hi);
        fold_acc(mine + G::W2, mine + G::B2, acc.w2, acc.b2, true, l31, hi);
        fold_acc(mine + G::W3, mine + G::B3, acc.w3, acc.b3, true, l31, hi);
      }
      __syncthreads();
      {
        const float4 *c0 = reinterpret_cast<const float4 *>(lds);
        float4 *dst = reinterpret_cast<float4 *>(kslab + ((long)blockIdx.x * slab_stride + (is_s ? 0 : 1) * (long)G::SIZE));
        constexpr int NV4 = G::SIZE / 4;
        for (int i = tid; i < NV4; i += 256) {
          const float4 p0 = c0[i], p1 = c0[i + NV4], p2 = c0[i + 2 * NV4], p3 = c0[i + 3 * NV4];
          float4 r;
          r.x = (p0.x + p1.x) + (p2.x + p3.x);
          r.y = (p0.y + p1.y) + (p2.y + p3.y);
          r.z = (p0.z + p1.z) + (p2.z + p3.z);
          r.w = (p0.w + p1.w) + (p2.w + p3.w);
          dst[i] = r;
        }
      }
      __syncthreads();  // image is restaged next phase; ybar stores of this phase are read by the next (same wave)
      NF_ST_STAMP(4 + phase * 4);
    }
  }
}

// ------------------------------------------------------------------------------------
// reverse pass from the stash, TWO wavefronts per tile (k_affine_bwd_pair)
// ------------------------------------------------------------------------------------
// k_affine_bwd_stashed runs one wavefront per SIMD (its 128 dW accumulator registers plus the delta / operand tiles
// need the whole 512-register file), so nothing fills the matrix pipe while that wave fetches operands, scales deltas
// or waits for its stash loads: 0.55 MFMA-busy (profiles/r3a_pmc_summary.json).  Here a tile is shared by a PRODUCER
// wave, which walks the delta chain (element-wise stage, dX3, dX2, dX1: 128 MFMAs, no accumulators) and leaves every
// delta tile in LDS, and a CONSUMER wave on the same SIMD, which owns the net's dW accumulators and contracts the
// stashed activations with those tiles (dW3, dW2, dW1: 128 MFMAs).  Both fit 256 registers, so the SIMD holds both, and
// the three workgroup barriers per tile line the two up GEMM by GEMM with equal MFMA counts on either side
// (32 | 64 | 32): whatever one wave does besides MFMAs is covered by the other's.
//   producer: d3 -> LDS | B1 | dX3, d2 -> LDS | B2 | dX2, d1 -> LDS | B3 | dX1, x2bar
//   consumer: a2 loads  | B1 | dW3 (d3)       | B2 | dW2 (d2)       | B3 | dW1 (d1)
// One buffer per delta suffices: d3 of the next tile is written after B3, when dW3 has long read it (before B2), etc.
// Whole tiles only (d = 64, N a multiple of 32) and every pair the same number of tiles (barriers inside the tile loop).
#ifndef NF_PAIR_DW_B6
#define NF_PAIR_DW_B6 0  // bit mask: which of the consumer's dW GEMMs (1: dW3, 2: dW2, 4: dW1) run on the bf16 matrix cores as well.
// Measured (profiles/r4m_pair_b6_ab.txt): 7 -> 613 us per launch (392 bytes of scratch: the consumer holds 128 accumulators and the
// split operands do not fit the remaining registers), 6 -> 591, 4 -> 338, 0 -> 334 (the producer's dX GEMMs only; fp32: 365).
#endif
// PB6 (round 4): the producer's three dX GEMMs on the bf16 matrix cores (dense_bwd_x_b6); the LDS then holds the net's
// B6T image instead of the fp32 one -- nobody else reads weights in this kernel (the consumer contracts activations with
// deltas), except the SLIM stash's a1 recompute, which therefore keeps the fp32 image.
// DW6 (with PB6): the consumer's dW GEMMs on the bf16 cores as well.  The producer splits every cotangent once (it needs the
// triples for its own dX GEMM), leaves them in LDS transposed (split_to_lds) instead of the fp32 tile, and the consumer splits
// only the stashed activations.  Two triple buffers per pair, used alternately (d3 | d2 | d1 | next d3 | ...): a buffer is
// rewritten two barriers after the GEMM that read it.
#ifndef NF_PAIR_CONS_PRIO
#define NF_PAIR_CONS_PRIO 0  // s_setprio of the consumer waves (A/B builds).  Measured (profiles/r6h_pair_consumer_prio_ab.txt): 1 and 3 alike
// 331-338 us against 316-322 -- the consumer's stages shorten (dW3 3.7 -> 2.6 k clocks, dW2 5.3 -> 4.55 k) and the producer's grow by
// more (prologue 3.6 -> 7.4 k, dX2 3.2 -> 5.4 k): the arbitration is all or nothing, and the two waves' issue work in a stage adds up
#endif
#ifndef NF_TRACE_PAIR
#define NF_TRACE_PAIR 0  // trace builds: which of the workgroup's four pairs tools/trace_bwd_pair.py sees
#endif
#ifndef NF_PAIR_SPREAD
#define NF_PAIR_SPREAD 0
#endif
#ifndef NF_PAIR_MSPLIT
#define NF_PAIR_MSPLIT 0  // bit 1: the anti-phase consumer's splits through nf_split16_mfma
#endif
#ifndef NF_PAIR_ANTI
#define NF_PAIR_ANTI 3  // bit 0: the producer's stages as [matrix burst | vector burst]; bit 1: the consumer's as [vector burst | matrix burst]
// -- ANTI-PHASE instead of woven streams (the probe: two mixed streams on one SIMD add up, a matrix burst and a vector burst overlap).
// Measured (profiles/r6w_pair_anti_phase_ab*.txt, alternating on one box): kernel 314.3 (0) / 320.5 (1) / 310.7 (2) / 307.7 us (3), step
// 0.5775 -> 0.5737 ms over eight pairs of runs (3 against 0).  Far less than the probe's ideal: the consumer's stage is its own serial
// [split 1.9 k | TR reads, 48 MFMAs, dots 2.6 k clocks] whatever the producer does beside it.  Same operations in the same order per
// accumulator: bit-identical results.
#endif
#ifndef NF_PAIR_WEAVE
#define NF_PAIR_WEAVE 1  // the consumer's splits of a1 / x2 in the issue shadows of dW3's / dW2's last matrix instructions (SplitTJob,
// nf_mfma.h) instead of behind them: 301.7-305.3 against 305.7-309.8 us alternating on one box (profiles/r6j_pair_weave_ab.txt); the
// same instructions on the same values, results bit-identical.  The traced stages barely move (dW3 3.7 k, dW2 5.1-5.3 k clocks): a
// stage's length is the two waves' matrix instructions taking turns on the pipe, not either wave's vector work
#endif
#ifndef NF_PAIR_TR
#define NF_PAIR_TR 1  // the producer -> consumer hand-over of the cotangent triples through ds_read_b64_tr_b16 (nf_mfma.h, round 6); 0: split_to_lds
#endif
#if NF_PAIR_TR
#define NF_PAIR_BUF TR_BUF
#define NF_PAIR_PUT(NB, buf, s) split_to_lds_tr<NB>(buf, s, l31, hi)
#define NF_PAIR_DW(IB, OB, as, buf, w, b) dw_accumulate_tr6<IB, OB>(as, buf, w, b, l31, hi)
#else
#define NF_PAIR_BUF D6_BUF
#define NF_PAIR_PUT(NB, buf, s) split_to_lds<NB>(buf, s, l31, hi)
#define NF_PAIR_DW(IB, OB, as, buf, w, b) dw_accumulate_t6<IB, OB>(as, buf, w, b, l31, hi)
#endif
template <class G, bool PB6 = false, bool DW6 = false>
struct BwdPairLds {
  static_assert(!DW6 || (PB6 && G::CB <= 2 && G::H1B <= 2 && G::H2B <= 2), "triple buffers hold two blocks");
  static constexpr int D3 = 0, D2 = D3 + G::CB * 32 * NF_TS, D1 = D2 + G::H2B * 32 * NF_TS;
  static constexpr int PAIR = DW6 ? 2 * NF_PAIR_BUF / 4 : D1 + G::H1B * 32 * NF_TS;
  static constexpr int PAIRS = 4;
  static constexpr int IMG = PB6 ? B6TGeo<G>::BYTES / 4 : G::SIZE;  // floats of the staged weight image
  static constexpr int FLOATS = (IMG + PAIRS * PAIR) > PAIRS * G::SIZE ? (IMG + PAIRS * PAIR) : PAIRS * G::SIZE;
  static constexpr size_t BYTES = (size_t)FLOATS * sizeof(float);  // the fold at the end of a phase needs 4 images
};

// FULL: d = 64 and N a multiple of the tile (no sample / feature masks).  INVD: reverse pass of the INVERSE coupling
// (forward-KL training; algebra of bwd_tile_stashed: phase S first, the UV slot holds w1).  live: this pair has a tile in
// this round of the workgroup's tile loop -- a pair without one only keeps the barrier count.
template <class G, bool PHASE_S, bool FULL, bool INVD, bool SLIM, bool PB6 = false, bool DW6 = false>
__device__ __forceinline__ void pair_produce(const CouplingArgs &a, const float *__restrict__ img, float *__restrict__ sp,
                                             StashFirst<G> &f, float *stash, int k, int ncoup, float *__restrict__ ybar,
                                             const float *__restrict__ lbar, float lbar_const, long tile, long next_tile,
                                             bool live, int l31, int hi, int par, long long *tr = nullptr) {
  using SG = StashGeo<G, SLIM>;
  using L = BwdPairLds<G, PB6, DW6>;
  char *bufa = reinterpret_cast<char *>(sp) + par * NF_PAIR_BUF, *bufb = reinterpret_cast<char *>(sp) + (par ^ 1) * NF_PAIR_BUF;  // DW6
  if (!live) {
    __syncthreads();
    __syncthreads();
    __syncthreads();
    return;
  }
  NF_TS_STAMP(0);
  const long j = tile * NF_TILE + l31;
  const bool valid = FULL ? true : j < a.N;
  const int par_c = 1 - a.par_t;
  const TileIO gio = make_tile_io(ybar, tile, a.d, l31, hi);
  const StashIO st = make_stash_io(stash, tile * ncoup + k, SG::SIZE, true, l31, hi);
  constexpr int nbase = SG::NET0 + (PHASE_S ? 0 : SG::NETSZ);
  f32x16 g1[G::CB];
#pragma unroll
  for (int b = 0; b < G::CB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) g1[b][r] = tile_load(gio, tile_soff(b, r, a.par_t));
  const u32x4 mk = __builtin_amdgcn_raw_buffer_load_b128(st.rs, (hi * 32 + l31) * 16, (nbase + SG::MSK) * 4, 0);
  f32x16 gold[G::MB];
#pragma unroll
  for (int b = 0; b < G::MB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) gold[b][r] = tile_load(gio, tile_soff(b, r, par_c));
  const float lb = valid ? (lbar ? lbar[FULL ? j : (j < a.N ? j : 0)] : lbar_const) : 0.f;
  f32x16 d3[G::CB];
#pragma unroll
  for (int b = 0; b < G::CB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = (b * 32 + nf_row(r, hi) < a.c) && valid;
      const float gv = g1[b][r];
      if (!PHASE_S) {
        d3[b][r] = ok ? (INVD ? -gv : gv) : 0.f;  // T-bar = ybar1 (inverse: -v1bar)
      } else if (INVD) {
        const float sv = f.sv[b][r];
        tile_store(gio, tile_soff(b, r, a.par_t), nf_fdiv(gv, nf_exp(sv)));  // v1bar
        d3[b][r] = ok ? -(gv * f.uv[b][r] + lb) * (1.f - sv * sv) : 0.f;    // S-bar through tanh (uv = w1)
      } else {
        const float sv = f.sv[b][r];
        tile_store(gio, tile_soff(b, r, a.par_t), gv * nf_exp(sv));        // x1bar
        d3[b][r] = ok ? (gv * f.uv[b][r] + lb) * (1.f - sv * sv) : 0.f;  // S-bar through tanh
      }
    }
  SplitC<DW6 ? G::CB : 1> s3;
  if constexpr (DW6) {
    split_C<G::CB>(d3, s3);
    NF_PAIR_PUT(G::CB, bufa, s3);
  } else {
    tile_to_scratch<G::CB>(sp + L::D3, d3, l31, hi);
  }
  unsigned m1[G::H1B], m2[G::H2B];
#pragma unroll
  for (int b = 0; b < G::H1B; ++b) m1[b] = mk[b];
#pragma unroll
  for (int b = 0; b < G::H2B; ++b) m2[b] = mk[2 + b];
  NF_TS_STAMP(1);
  __syncthreads();  // B1: d3 is in LDS (and the consumer is done with the previous tile's d1)
  NF_TS_STAMP(2);
  f32x16 d2[G::H2B];
  const nf_u32x4 *wt = reinterpret_cast<const nf_u32x4 *>(img);  // PB6: the staged image is the net's B6T image
  SplitC<DW6 ? G::H2B : 1> s2;
#if NF_PAIR_TR
  // (round 6) the GEMM block by block: what follows a finished block of d2 -- slopes, split, hand-over stores -- rides in the
  // issue shadows of the next block's matrix instructions; only the last block's share is left behind the GEMM
  const int xw = (l31 >> 2) & 3;
  char *const pw0 = reinterpret_cast<char *>(sp) + l31 * 32 + 8 * (hi ^ xw), *const pw1 = reinterpret_cast<char *>(sp) + l31 * 32 + 8 * ((2 + hi) ^ xw);
  if constexpr (DW6) {
    const PairPost<G::H2B> post{d2, m2, s2, pw0 + (bufb - reinterpret_cast<char *>(sp)), pw1 + (bufb - reinterpret_cast<char *>(sp))};
#if NF_PAIR_ANTI & 1  // the GEMM as ONE burst of matrix instructions, then ONE burst of vector work (see NF_PAIR_ANTI)
    dense_bwd_x_b6s<G::H2B, G::CB>(wt + B6TGeo<G>::T3, s3, d2, l31, hi);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < G::H2B; ++b) post.all(b);
#else
    dense_bwd_x_b6s_blocks<G::H2B, G::CB>(wt + B6TGeo<G>::T3, s3, d2, l31, hi, [&](int ib, int i) { if (ib > 0) post.template at_hook<24 * G::CB>(ib - 1, i); });
    post.all(G::H2B - 1);
#endif
  } else
#endif
  {
  if constexpr (DW6) dense_bwd_x_b6s<G::H2B, G::CB>(wt + B6TGeo<G>::T3, s3, d2, l31, hi);
  else if constexpr (PB6) dense_bwd_x_b6<G::H2B, G::CB>(wt + B6TGeo<G>::T3, d3, d2, l31, hi);
  else dense_bwd_x<G::H2B, G::CB>(img + G::W3, d3, d2, l31, hi);
  apply_lrelu_grad<G::H2B>(d2, m2);
  if constexpr (DW6) {
    split_C<G::H2B>(d2, s2);
    NF_PAIR_PUT(G::H2B, bufb, s2);
  } else {
    tile_to_scratch<G::H2B>(sp + L::D2, d2, l31, hi);
  }
  }
  NF_TS_STAMP(3);
  __syncthreads();  // B2
  NF_TS_STAMP(4);
  f32x16 d1[G::H1B];
  SplitC<DW6 ? G::H1B : 1> s1;
#if NF_PAIR_TR
  if constexpr (DW6) {
    const PairPost<G::H1B> post{d1, m1, s1, pw0 + (bufa - reinterpret_cast<char *>(sp)), pw1 + (bufa - reinterpret_cast<char *>(sp))};
#if NF_PAIR_ANTI & 1
    dense_bwd_x_b6s<G::H1B, G::H2B>(wt + B6TGeo<G>::T2, s2, d1, l31, hi);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < G::H1B; ++b) post.all(b);
#else
    dense_bwd_x_b6s_blocks<G::H1B, G::H2B>(wt + B6TGeo<G>::T2, s2, d1, l31, hi, [&](int ib, int i) { if (ib > 0) post.template at_hook<24 * G::H2B>(ib - 1, i); });
    post.all(G::H1B - 1);
#endif
  } else
#endif
  {
  if constexpr (DW6) dense_bwd_x_b6s<G::H1B, G::H2B>(wt + B6TGeo<G>::T2, s2, d1, l31, hi);
  else if constexpr (PB6) dense_bwd_x_b6<G::H1B, G::H2B>(wt + B6TGeo<G>::T2, d2, d1, l31, hi);
  else dense_bwd_x<G::H1B, G::H2B>(img + G::W2, d2, d1, l31, hi);
  apply_lrelu_grad<G::H1B>(d1, m1);
  if constexpr (DW6) {
    split_C<G::H1B>(d1, s1);
    NF_PAIR_PUT(G::H1B, bufa, s1);
  } else {
    tile_to_scratch<G::H1B>(sp + L::D1, d1, l31, hi);
  }
  }
  NF_TS_STAMP(5);
  __syncthreads();  // B3
  NF_TS_STAMP(6);
  if (PHASE_S && next_tile >= 0) stash_issue_first<G, SLIM>(f, stash, k, ncoup, next_tile, l31, hi);
  f32x16 g2[G::MB];
  if constexpr (DW6) dense_bwd_x_b6s<G::MB, G::H1B>(wt + B6TGeo<G>::T1, s1, g2, l31, hi);
  else if constexpr (PB6) dense_bwd_x_b6<G::MB, G::H1B>(wt + B6TGeo<G>::T1, d1, g2, l31, hi);
  else dense_bwd_x<G::MB, G::H1B>(img + G::W1, d1, g2, l31, hi);
#pragma unroll
  for (int b = 0; b < G::MB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) tile_store(gio, tile_soff(b, r, par_c), gold[b][r] + g2[b][r]);
  NF_TS_STAMP(7);
}

// split_T of one operand with the 4 NB requests of the next operand spread between its 2 NA k-group splits (NF_PAIR_SPREAD)
template <int NA, int NB>
__device__ __forceinline__ void consumer_split_and_request(const StashIO &st, int base, int voff, const float (&at)[NA][16], SplitT<NA> &s,
                                                           float (&bt)[NB][16]) {
  constexpr int NS = 2 * NA, NL = 4 * NB;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int ib = i >> 1, g = i & 1;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = at[ib][8 * g + j];
#pragma unroll
    for (int l = i * NL / NS; l < (i + 1) * NL / NS; ++l) {  // this k-group's share of the requests, in front of its split
      const int b = l >> 2, q = l & 3;
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(st.rs, voff, (base + b * 1024) * 4 + q * 16, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned bits = w[e];
        bt[b][4 * q + e] = __builtin_bit_cast(float, bits);
      }
    }
    nf_split8(v, s.h[ib][g], s.m[ib][g], s.l[ib][g]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// (Round 6, measured and removed: the consumer held back by s_sleep behind each barrier -- 256 / 512 / 768 clocks in stage 1,
// twice that in stage 2 -- so that its matrix instructions would run beside the producer's vector tail instead of beside its
// GEMM: 323.2-325.2 against 325.5 us on one box, nothing.  The older wave already wins the arbitration.)
// (Measured and removed: dW1 of a tile moved in front of the NEXT tile's first barrier, where the producer issues no
// MFMAs -- 374-379 us against 366 in one process; and the producer's next-unit operands requested behind B3 -- 370.)
template <class G, bool SLIM, bool PB6 = false, bool DW6 = false>
// (`sp` is NOT __restrict__: the DW6 triple buffers are rewritten by the producer between this wave's GEMMs, and through a
// const restrict pointer hipcc reuses the registers of an earlier read of the same address across the barriers.)
__device__ __forceinline__ void pair_consume(const float *__restrict__ img, const float *sp, BwdAcc<G> &acc, float *stash,
                                             int k, int ncoup, long tile, bool is_s, bool live, int l31, int hi, int par,
                                             long long *tr = nullptr) {
  using SG = StashGeo<G, SLIM>;
  using L = BwdPairLds<G, PB6, DW6>;
  if (!live) {
    __syncthreads();
    __syncthreads();
    __syncthreads();
    return;
  }
  NF_TS_STAMP(0);
  const int nbase = SG::NET0 + (is_s ? 0 : SG::NETSZ);
  const StashIO st = make_stash_io(stash, tile * ncoup + k, SG::SIZE, true, l31, hi);
  const int vT = (l31 * 32 + hi * 16) * 4;
  if constexpr (DW6) {
    const char *bufa = reinterpret_cast<const char *>(sp) + par * NF_PAIR_BUF, *bufb = reinterpret_cast<const char *>(sp) + (par ^ 1) * NF_PAIR_BUF;
    float a1t[G::H1B][16];
#if (NF_PAIR_ANTI & 2) && NF_PAIR_TR
    // ANTI-PHASE (round 6): two MIXED matrix + vector streams on one SIMD add up, a matrix burst and a vector burst overlap
    // (tools/probe/mfma_valu_overlap_probe.hip).  So every stage of this wave is [split the stage's own activation operand | dW GEMM],
    // the producer's [dX GEMM | slopes, split, hand-over stores]: while one wave is in its matrix burst the other is in its vector
    // burst.  The operand of stage k + 1 is requested at the head of stage k; only one operand's triples are live at a time.
    {
#if NF_PAIR_MSPLIT & 2  // the consumer's splits with their subtractions on the matrix pipe (nf_split16_mfma): a third of the vector work
      const SplitSel sel = nf_split_sel(l31, hi);
#define NF_CONS_SPLIT(NB, at, xs)                                              \
  _Pragma("unroll") for (int ib_ = 0; ib_ < NB; ++ib_) {                       \
    f32x16 v_;                                                               \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) v_[r_] = at[ib_][r_];    \
    nf_split16_mfma(sel, v_, xs.h[ib_], xs.m[ib_], xs.l[ib_]);               \
  }
#else
#define NF_CONS_SPLIT(NB, at, xs) split_T<NB>(at, xs)
#endif
      // the NEXT stage's operand is requested while this stage's is split.  NF_PAIR_SPREAD: its 16-byte requests go out one by one
      // between the split's k-groups instead of in one batch behind the barrier -- there all eight waves of the CU issue theirs within
      // the same few hundred clocks and a wave is held 590-780 clocks at ISSUE (64 bytes per clock and CU; tools/trace_bwd_pair.py)
#if NF_PAIR_SPREAD
#define NF_CONS_REQ_SPLIT(NA, at, xs, NBQ, baseq, bt) consumer_split_and_request<NA, NBQ>(st, baseq, vT, at, xs, bt)
#else
#define NF_CONS_REQ_SPLIT(NA, at, xs, NBQ, baseq, bt) \
  do { stash_get_T<NBQ>(st, baseq, vT, bt); NF_CONS_SPLIT(NA, at, xs); } while (0)
#endif
      float x2t[G::MB][16];
      {
        float a2t[G::H2B][16];
        stash_get_T<G::H2B>(st, nbase + SG::A2, vT, a2t);
        NF_TS_STAMP(1);
        __syncthreads();  // B1
        NF_TS_STAMP(2);
        SplitT<G::H2B> a2s;
        NF_CONS_REQ_SPLIT(G::H2B, a2t, a2s, G::H1B, nbase + SG::A1, a1t);
        __builtin_amdgcn_sched_barrier(0);
        NF_PAIR_DW(G::H2B, G::CB, a2s, bufa, acc.w3, acc.b3);
      }
      NF_TS_STAMP(3);
      __syncthreads();  // B2
      NF_TS_STAMP(4);
      {
        SplitT<G::H1B> a1s;
        NF_CONS_REQ_SPLIT(G::H1B, a1t, a1s, G::MB, SG::XT, x2t);
        __builtin_amdgcn_sched_barrier(0);
        NF_PAIR_DW(G::H1B, G::H2B, a1s, bufb, acc.w2, acc.b2);
      }
      NF_TS_STAMP(5);
      __syncthreads();  // B3
      NF_TS_STAMP(6);
      {
        SplitT<G::MB> x2s;
        NF_CONS_SPLIT(G::MB, x2t, x2s);
        __builtin_amdgcn_sched_barrier(0);
        NF_PAIR_DW(G::MB, G::H1B, x2s, bufa, acc.w1, acc.b1);
      }
      NF_TS_STAMP(7);
      return;
    }
#endif
    {
      // (Measured and removed: the NEXT tile's a2 requested behind B3 -- 32 more registers across the loop's back edge, which
      // hipcc spills to scratch right behind the loads: 428 us per launch against 345.)
      SplitT<G::H2B> a2s;
      {
        float a2t[G::H2B][16];
        stash_get_T<G::H2B>(st, nbase + SG::A2, vT, a2t);
        split_T<G::H2B>(a2t, a2s);
      }
      stash_get_T<G::H1B>(st, nbase + SG::A1, vT, a1t);  // in flight behind dW3
      NF_TS_STAMP(1);
      __syncthreads();  // B1
      NF_TS_STAMP(2);
#if NF_PAIR_WEAVE && NF_PAIR_TR
      // a1's split in the issue shadows of dW3's second half (24 matrix instructions, a1 arrives during the first)
      SplitT<G::H1B> a1s;
      {
        constexpr int NH = 12 * G::H2B * G::CB, F1 = NH > 8 * G::H1B ? NH - 8 * G::H1B : 0;
        const SplitTJob<G::H1B, F1, 1> job{a1t, a1s};
        dw_accumulate_tr6<G::H2B, G::CB>(a2s, bufa, acc.w3, acc.b3, l31, hi, job);
        job.template finish<NH>();
      }
      float x2t[G::MB][16];
      stash_get_T<G::MB>(st, SG::XT, vT, x2t);
      NF_TS_STAMP(3);
      __syncthreads();  // B2
      NF_TS_STAMP(4);
      SplitT<G::MB> x2s;
      {
        constexpr int NH = 12 * G::H1B * G::H2B, F2 = NH > 16 * G::MB ? NH - 16 * G::MB : 0;
        const SplitTJob<G::MB, F2, 2> job{x2t, x2s};
        dw_accumulate_tr6<G::H1B, G::H2B>(a1s, bufb, acc.w2, acc.b2, l31, hi, job);
        job.template finish<NH>();
      }
      NF_TS_STAMP(5);
      __syncthreads();  // B3
      NF_TS_STAMP(6);
      NF_PAIR_DW(G::MB, G::H1B, x2s, bufa, acc.w1, acc.b1);
      NF_TS_STAMP(7);
      return;
#endif
      NF_PAIR_DW(G::H2B, G::CB, a2s, bufa, acc.w3, acc.b3);
    }
    float x2t[G::MB][16];
    {
      SplitT<G::H1B> a1s;
      split_T<G::H1B>(a1t, a1s);
      stash_get_T<G::MB>(st, SG::XT, vT, x2t);
      NF_TS_STAMP(3);
      __syncthreads();  // B2
      NF_TS_STAMP(4);
      NF_PAIR_DW(G::H1B, G::H2B, a1s, bufb, acc.w2, acc.b2);
    }
    {
      SplitT<G::MB> x2s;
      split_T<G::MB>(x2t, x2s);
      NF_TS_STAMP(5);
      __syncthreads();  // B3
      NF_TS_STAMP(6);
      NF_PAIR_DW(G::MB, G::H1B, x2s, bufa, acc.w1, acc.b1);
      NF_TS_STAMP(7);
    }
    return;  // (the dW1 block below belongs to the two fp32-tile forms)
  } else if constexpr (SLIM) {
    // a1 is not in the stash: rebuilt here, in this wave's MFMA-free prologue (the producer is in its own: operand loads,
    // element-wise stage, d3 -> LDS), from x2 in the C layout and the W1 rows of the staged image
    float a1t[G::H1B][16];
    {
      f32x16 x2c[G::MB];
      stash_get_lane<G::MB>(st, SG::XL, x2c);
      float a2t[G::H2B][16];
      stash_get_T<G::H2B>(st, nbase + SG::A2, vT, a2t);
      recompute_a1t<G>(img, x2c, a1t, l31, hi);
      NF_TS_STAMP(1);
      __syncthreads();  // B1
      NF_TS_STAMP(2);
      dw_accumulate_reg<G::H2B, G::CB>(a2t, sp + L::D3, acc.w3, acc.b3, l31, hi);
    }
    NF_TS_STAMP(3);
    __syncthreads();  // B2
    NF_TS_STAMP(4);
    dw_accumulate_reg<G::H1B, G::H2B, NoSideJob, true>(a1t, sp + L::D2, acc.w2, acc.b2, l31, hi);
  } else {
  {
    float a2t[G::H2B][16];
    stash_get_T<G::H2B>(st, nbase + SG::A2, vT, a2t);
    if constexpr (PB6 && (NF_PAIR_DW_B6 & 1)) {
      SplitT<G::H2B> a2s;
      split_T<G::H2B>(a2t, a2s);
      NF_TS_STAMP(1);
      __syncthreads();  // B1
      NF_TS_STAMP(2);
      dw_accumulate_reg_b6<G::H2B, G::CB>(a2s, sp + L::D3, acc.w3, acc.b3, l31, hi);
    } else {
    NF_TS_STAMP(1);
    __syncthreads();  // B1
    NF_TS_STAMP(2);
    dw_accumulate_reg<G::H2B, G::CB>(a2t, sp + L::D3, acc.w3, acc.b3, l31, hi);
    }
  }
  if constexpr (PB6 && NF_PAIR_DW_B6 != 0) __builtin_amdgcn_sched_barrier(0);  // the next operand block is requested AFTER this GEMM: registers
  {
    float a1t[G::H1B][16];
    stash_get_T<G::H1B>(st, nbase + SG::A1, vT, a1t);
    if constexpr (PB6 && (NF_PAIR_DW_B6 & 2)) {
      SplitT<G::H1B> a1s;
      split_T<G::H1B>(a1t, a1s);
      NF_TS_STAMP(3);
      __syncthreads();  // B2
      NF_TS_STAMP(4);
      dw_accumulate_reg_b6<G::H1B, G::H2B>(a1s, sp + L::D2, acc.w2, acc.b2, l31, hi);
    } else {
    NF_TS_STAMP(3);
    __syncthreads();  // B2
    NF_TS_STAMP(4);
    dw_accumulate_reg<G::H1B, G::H2B>(a1t, sp + L::D2, acc.w2, acc.b2, l31, hi);
    }
  }
  }
  if constexpr (PB6 && NF_PAIR_DW_B6 != 0) __builtin_amdgcn_sched_barrier(0);
  {
    float x2t[G::MB][16];
    stash_get_T<G::MB>(st, SG::XT, vT, x2t);
    if constexpr (PB6 && (NF_PAIR_DW_B6 & 4)) {
      SplitT<G::MB> x2s;
      split_T<G::MB>(x2t, x2s);
      NF_TS_STAMP(5);
      __syncthreads();  // B3
      NF_TS_STAMP(6);
      dw_accumulate_reg_b6<G::MB, G::H1B>(x2s, sp + L::D1, acc.w1, acc.b1, l31, hi);
    } else {
    NF_TS_STAMP(5);
    __syncthreads();  // B3
    NF_TS_STAMP(6);
    dw_accumulate_reg<G::MB, G::H1B>(x2t, sp + L::D1, acc.w1, acc.b1, l31, hi);
    }
    NF_TS_STAMP(7);
  }
}

// the two roles' common step of a phase: the slab write after the consumers' fold
template <class G>
__device__ __forceinline__ void pair_slab_write(const float *__restrict__ lds, float *__restrict__ dstf, int tid) {
  const float4 *c0 = reinterpret_cast<const float4 *>(lds);
  constexpr int NV4 = G::SIZE / 4;
  // through a buffer descriptor: the slab's base is wave-uniform and the element offset 32 bits wide -- as a flat pointer the
  // 64-bit per-lane address of this loop cost the consumer two registers it does not have (12 bytes of scratch, round 5)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dstf, 0, G::SIZE * 4, 0x00020000);
  for (int i = tid; i < NV4; i += 512) {
    const float4 p0 = c0[i], p1 = c0[i + NV4], p2 = c0[i + 2 * NV4], p3 = c0[i + 3 * NV4];
    u32x4 r;
    r[0] = __builtin_bit_cast(unsigned, (p0.x + p1.x) + (p2.x + p3.x));
    r[1] = __builtin_bit_cast(unsigned, (p0.y + p1.y) + (p2.y + p3.y));
    r[2] = __builtin_bit_cast(unsigned, (p0.z + p1.z) + (p2.z + p3.z));
    r[3] = __builtin_bit_cast(unsigned, (p0.w + p1.w) + (p2.w + p3.w));
    nf_buffer_store_b128(r, rs, i * 16, 0);
  }
}

// The role branch is the OUTERMOST statement: the producer's prefetch registers and the consumer's accumulators then
// never count against the other role's 256 registers (inside the coupling loop both are live across either branch).
// Every wave executes the same barriers: per phase 1 (image staged) + 3 per round of the tile loop + 3 (tiles done,
// folded, slab written); the tile loop runs as many rounds as the workgroup's first pair needs.
template <class G, bool FULL, bool INVD, bool SLIM, bool PB6 = false, bool DW6 = false>
__global__ __launch_bounds__(512) void k_affine_bwd_pair(BwdAllArgs aa, float *stash, float *__restrict__ ybar,
                                                         const float *__restrict__ lbar, float lbar_const,
                                                         float *__restrict__ slab, long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *img = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave & 3, role = wave >> 2;  // waves p and p + 4 sit on the same SIMD; role 0 produces, 1 consumes
  const int l31 = lane & 31, hi = lane >> 5;
  static_assert(!(SLIM && PB6), "the SLIM stash's a1 recompute reads the fp32 image");
  using PL = BwdPairLds<G, PB6, DW6>;
  using BT = B6TGeo<G>;
  float *sp = lds + PL::IMG + pair * PL::PAIR;
  auto stage_image = [&](int k, bool is_s) {  // the net's weight image of this phase -> LDS (fp32, or B6T for the producer's dX GEMMs)
    if constexpr (PB6)
      stage_packed<BT::BYTES / 4, 512>(img, reinterpret_cast<const float *>(aa.wimg_b6t + (size_t)(2 * k + (is_s ? 0 : 1)) * BT::BYTES), tid);
    else
      stage_packed<G::SIZE, 512>(img, aa.wimg + (size_t)(2 * k + (is_s ? 0 : 1)) * G::SIZE, tid);
  };
  const long ntiles = (aa.N + NF_TILE - 1) / NF_TILE;
  const long tile0 = (long)blockIdx.x * 4 + pair, tstride = (long)gridDim.x * 4;
  const long wg0 = (long)blockIdx.x * 4;
  const int rounds = wg0 < ntiles ? (int)((ntiles - wg0 + tstride - 1) / tstride) : 0;
  if (role == 0) {
    StashFirst<G> f;
    if (INVD && tile0 < ntiles) stash_issue_first<G, SLIM>(f, stash, aa.ncoup - 1, aa.ncoup, tile0, l31, hi);  // S runs first
#pragma unroll 1
    for (int step = 0; step < aa.ncoup; ++step) {
      const int k = INVD ? aa.ncoup - 1 - step : step;  // the inverse chain's reverse pass runs in execution order
      CouplingArgs a;
      a.theta = nullptr;
      a.img_s = aa.wimg + (size_t)(2 * k) * G::SIZE;
      a.img_t = a.img_s + G::SIZE;
      a.trace = nullptr;
      a.d = aa.d;
      a.par_t = k & 1;
      a.c = (k & 1) ? aa.d / 2 : (aa.d + 1) / 2;
      a.m = aa.d - a.c;
      a.N = aa.N;
#pragma unroll 1
      for (int phase = 0; phase < 2; ++phase) {
        const bool is_s = INVD ? phase == 0 : phase == 1;
#ifdef NF_KERNEL_TRACE  // the phase boundary: producer stamps [32 + phase 8 + 0..7], consumer [96 + ...]
        long long *trb = (aa.trace && blockIdx.x == 0 && tid == 64 * NF_TRACE_PAIR && step == 0) ? aa.trace + 32 + phase * 8 : nullptr;
#endif
        NF_TSB(0);
        stage_image(k, is_s);
        NF_TSB(1);
        __syncthreads();
        NF_TSB(2);
#pragma unroll 1
        for (int it = 0; it < rounds; ++it) {
          const long tile = tile0 + (long)it * tstride;
          const long nt = tile + tstride < ntiles ? tile + tstride : -1;
#ifdef NF_KERNEL_TRACE  // tools/trace_bwd_pair.py: block 0, pair 0, first coupling; producer stamps [phase 16 + tile 8 + 0..7]
          long long *tr = (aa.trace && blockIdx.x == 0 && tid == 64 * NF_TRACE_PAIR && step == 0 && it < 2) ? aa.trace + phase * 16 + it * 8 : nullptr;
#else
          long long *tr = nullptr;
#endif
          if (!is_s) pair_produce<G, false, FULL, INVD, SLIM, PB6, DW6>(a, img, sp, f, stash, k, aa.ncoup, ybar, lbar, lbar_const, tile, nt, tile < ntiles, l31, hi, it & 1, tr);
          else pair_produce<G, true, FULL, INVD, SLIM, PB6, DW6>(a, img, sp, f, stash, k, aa.ncoup, ybar, lbar, lbar_const, tile, nt, tile < ntiles, l31, hi, it & 1, tr);
        }
        NF_TSB(3);
        __syncthreads();  // every wave is done with the weight image and the delta tiles
        NF_TSB(4);
        // s and u of the next phase-S's first tile fly behind the fold, the slab write and the staging of the next image
        if (tile0 < ntiles) {
          if (!INVD && !is_s) stash_issue_first<G, SLIM>(f, stash, k, aa.ncoup, tile0, l31, hi);
          if (INVD && !is_s && step + 1 < aa.ncoup) stash_issue_first<G, SLIM>(f, stash, k - 1, aa.ncoup, tile0, l31, hi);
        }
        __syncthreads();  // the consumers have folded
        NF_TSB(5);
        pair_slab_write<G>(lds, slab + (long)k * 2 * G::SIZE + ((long)blockIdx.x * slab_stride + (is_s ? 0 : 1) * (long)G::SIZE), tid);
        NF_TSB(6);
        __syncthreads();
        NF_TSB(7);
      }
    }
  } else {
#if NF_PAIR_CONS_PRIO
    __builtin_amdgcn_s_setprio(NF_PAIR_CONS_PRIO);
#endif
#pragma unroll 1
    for (int step = 0; step < aa.ncoup; ++step) {
      const int k = INVD ? aa.ncoup - 1 - step : step;
      const float *img_s = aa.wimg + (size_t)(2 * k) * G::SIZE;
#pragma unroll 1
      for (int phase = 0; phase < 2; ++phase) {
        const bool is_s = INVD ? phase == 0 : phase == 1;
#ifdef NF_KERNEL_TRACE
        long long *trb = (aa.trace && blockIdx.x == 0 && tid == 256 + 64 * NF_TRACE_PAIR && step == 0) ? aa.trace + 96 + phase * 8 : nullptr;
#endif
        NF_TSB(0);
        stage_image(k, is_s);
        NF_TSB(1);
        __syncthreads();
        NF_TSB(2);
        BwdAcc<G> acc;
        zero_acc(acc.w1, acc.b1);
        zero_acc(acc.w2, acc.b2);
        zero_acc(acc.w3, acc.b3);
#pragma unroll 1
        for (int it = 0; it < rounds; ++it) {
          const long tile = tile0 + (long)it * tstride;
#ifdef NF_KERNEL_TRACE  // consumer stamps at [64 + ...]
          long long *tr = (aa.trace && blockIdx.x == 0 && tid == 256 + 64 * NF_TRACE_PAIR && step == 0 && it < 2) ? aa.trace + 64 + phase * 16 + it * 8 : nullptr;
#else
          long long *tr = nullptr;
#endif
          pair_consume<G, SLIM, PB6, DW6>(img, sp, acc, stash, k, aa.ncoup, tile, is_s, tile < ntiles, l31, hi, it & 1, tr);
        }
        NF_TSB(3);
        __syncthreads();  // every wave is done with the weight image and the delta tiles
        NF_TSB(4);
        {
          float *mine = lds + pair * G::SIZE;
          fold_acc(mine + G::W1, mine + G::B1, acc.w1, acc.b1, true, l31, hi);
          fold_acc(mine + G::W2, mine + G::B2, acc.w2, acc.b2, true, l31, hi);
          fold_acc(mine + G::W3, mine + G::B3, acc.w3, acc.b3, true, l31, hi);
        }
        __syncthreads();
        NF_TSB(5);
        pair_slab_write<G>(lds, slab + (long)k * 2 * G::SIZE + ((long)blockIdx.x * slab_stride + (is_s ? 0 : 1) * (long)G::SIZE), tid);
        NF_TSB(6);
        __syncthreads();
        NF_TSB(7);
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------
static inline int blocks32(int n) { return (n + 31) / 32; }
// nets with 1, 3 or 4 hidden layers (nf_deep.hip): to nf_api.hip they are "resident RealNVP flows without a stash", so every
// nf_affine_* entry point below hands them over first
int nf_deep_geo_id(const nf_flow_desc *desc);
int nf_deep_image_floats(const nf_flow_desc *desc);
size_t nf_deep_wimg_bytes(const nf_flow_desc *desc);
long nf_deep_slab_floats(const nf_flow_desc *desc);
int nf_deep_pack(nf_ctx *ctx, const nf_flow_desc *desc, const float *theta);
int nf_deep_reduce_slabs(nf_ctx *ctx, const nf_flow_desc *desc, const float *slab, int nslab, float *g, const double *lpart, int nlpart,
                         float *lout);
int nf_deep_chain(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, float *xt, long N, float *ladj, int k_only, int accumulate);
int nf_deep_bwd(nf_ctx *ctx, const nf_flow_desc *desc, int k_lo, int k_hi, float *y, float *ybar, const float *lbar, float lbar_const,
                long N, float *slab, long slab_stride, int grid, bool inv_dir);
#define NF_GEO_H32 NetGeo<1, 1, 1, 1>
#define NF_GEO_H64 NetGeo<1, 2, 2, 1>

static int geo_size(const nf_flow_desc *desc) {
  if (desc->n_hidden != 2) return 0;
  const int c = (desc->d + 1) / 2;
  const int mb = blocks32(c), h1b = blocks32(desc->hdims[0]), h2b = blocks32(desc->hdims[1]);
  if (mb == 1 && h1b == 1 && h2b == 1) return NetGeo<1, 1, 1, 1>::SIZE;
  if (mb == 1 && h1b == 2 && h2b == 2) return NetGeo<1, 2, 2, 1>::SIZE;
  return 0;
}

// floats of one workgroup's gradient slab (image layout): [coupling][net s|t][G::SIZE]
long nf_affine_slab_floats(const nf_flow_desc *desc) {
  if (nf_deep_geo_id(desc)) return nf_deep_slab_floats(desc);
  return (long)2 * desc->nlayers * 2 * geo_size(desc);
}

int nf_affine_reduce_slabs(nf_ctx *ctx, const nf_flow_desc *desc, const float *slab, int nslab, float *g,
                           const double *lpart, int nlpart, float *lout) {
  if (nf_deep_geo_id(desc)) return nf_deep_reduce_slabs(ctx, desc, slab, nslab, g, lpart, nlpart, lout);
  const int size = geo_size(desc);
  if (!size) return NF_ERR_UNSUPPORTED;
  const PackArgs p = make_pack_args(desc);
  const long total = (long)p.ncoup * 2 * size;
  const unsigned grid = (unsigned)((total + 63) / 64);  // 64 elements per block (k_reduce_image_slabs)
  ProfScope ps(ctx, "reduce_slabs");
  if (size == NetGeo<1, 1, 1, 1>::SIZE)
    hipLaunchKernelGGL((k_reduce_image_slabs<NetGeo<1, 1, 1, 1>>), dim3(grid), dim3(64 * NF_REDUCE_WAVES), 0, ctx->stream, p, slab, nslab, total, g, lpart, nlpart, lout);
  else
    hipLaunchKernelGGL((k_reduce_image_slabs<NetGeo<1, 2, 2, 1>>), dim3(grid), dim3(64 * NF_REDUCE_WAVES), 0, ctx->stream, p, slab, nslab, total, g, lpart, nlpart, lout);
  return (int)hipGetLastError();
}

// the fused epilogue of nf_elbo_step (nf_pack.h: k_affine_epilogue).  mode: 1 = slabs -> gradient + loss only (multi-GPU,
// before the all-reduce), 2 = Adam + ||g|| + packed images from a finished gradient (after it), 3 = both (single GPU).
long nf_affine_epilogue_blocks(const nf_flow_desc *desc) { return ((long)2 * desc->nlayers * 2 * geo_size(desc) + 63) / 64; }
int nf_affine_epilogue(nf_ctx *ctx, const nf_flow_desc *desc, int mode, const float *slab, int nslab, float *g,
                       const double *lpart, int nlpart, float *theta, float *m, float *v, double lr, double b1, double b2,
                       double eps, unsigned t_val, unsigned *t_ptr, double *gpart) {
  const int size = geo_size(desc);
  if (!size || !ctx->wimg) return NF_ERR_UNSUPPORTED;
  const PackArgs p = make_pack_args(desc);
  EpiArgs a;
  a.slab = slab; a.nslab = nslab; a.slab_stride = (long)p.ncoup * 2 * size;
  a.g = g; a.P = nf_param_count(desc);
  a.lpart = lpart; a.nlpart = nlpart;
  a.theta = theta; a.m = m; a.v = v; a.wimg = (float *)ctx->wimg;
  a.lr = (float)lr; a.b1 = (float)b1; a.b2 = (float)b2; a.eps = (float)eps; a.b1d = b1; a.b2d = b2;
  a.t_val = t_val; a.t_ptr = t_ptr; a.gpart = gpart;
  a.c1 = (float)(1.0 - pow(b1, (double)t_val + 1.0));
  a.c2 = (float)(1.0 - pow(b2, (double)t_val + 1.0));
  const unsigned grid = (unsigned)nf_affine_epilogue_blocks(desc);
  if (mode != 1) ctx->wimg_gen++;  // the fp32 images are rewritten (Adam's theta): B6 copies are stale
  // (Measured and removed: the epilogue refreshing the B6T images element by element as well, which saves the next step its 7 us
  // conversion launch -- the reverse kernel then stages images that were written a whole forward kernel earlier instead of a
  // moment ago and runs 4-5 us slower: 0.594 against 0.594 ms per step.)
  ProfScope ps(ctx, mode == 2 ? "adam" : "reduce_slabs");
  const bool h64 = size != NetGeo<1, 1, 1, 1>::SIZE;
#define NF_EPI(GEO)                                                                                                          \
  do {                                                                                                                       \
    if (mode == 3) hipLaunchKernelGGL((k_affine_epilogue<GEO, true, true, NF_REDUCE_WAVES>), dim3(grid), dim3(64 * NF_REDUCE_WAVES), 0, ctx->stream, p, a); \
    else if (mode == 1) hipLaunchKernelGGL((k_affine_epilogue<GEO, true, false, NF_REDUCE_WAVES>), dim3(grid), dim3(64 * NF_REDUCE_WAVES), 0, ctx->stream, p, a); \
    else hipLaunchKernelGGL((k_affine_epilogue<GEO, false, true, 1>), dim3(grid), dim3(64), 0, ctx->stream, p, a);             \
  } while (0)
  if (h64) NF_EPI(NF_GEO_H64); else NF_EPI(NF_GEO_H32);
#undef NF_EPI
  return (int)hipGetLastError();
}

// packs every net of the flow into ctx->wimg (grow-only) -- call once per API entry
// Where the B6 images sit: behind the fp32 images of the same flow in ctx->wimg (nf_affine_wimg_bytes covers both).
template <class G>
static size_t b6_offset_bytes(const nf_flow_desc *desc) { return (size_t)2 * desc->nlayers * 2 * G::SIZE * sizeof(float); }
// Which chain launches take the bf16 six-term products (B6).  Measured A/B on one box (profiles/r4i_b6_ab.txt):
//   * chains WITHOUT a stash (nf_flow_fwd / nf_flow_inv, nf_elbo_batch, nf_loglikelihood: BASELINE cfg 5, 1 M samples):
//     1.69 against 2.46 ms = 1.45 x, 620 M samples/s, 1.03 of the fp32-MFMA roofline -- ON by default, NF_FWD_FP32=1 is the
//     A/B switch back;
//   * the training step's stashing forward (cfg 2): 207-210 against 195 us, AND the unchanged reverse kernel behind it runs
//     400 instead of 362 us (the chip's power management couples consecutive kernels: tools/bench_ramp.py shows the clock is
//     set by the load of the last milliseconds) -- 0.659 against 0.612 ms per step: OFF by default, NF_FWD_B6_STASH=1 turns
//     it on (120 bytes of scratch spills in that variant are the first thing to remove).
static bool fwd_b6(bool stashing = false) {
  static const bool off = std::getenv("NF_FWD_FP32") != nullptr, stash_on = std::getenv("NF_FWD_B6_STASH") != nullptr;
  return !off && (!stashing || stash_on);
}

static size_t b6_image_bytes(int size) { return size == NetGeo<1, 1, 1, 1>::SIZE ? B6Geo<NetGeo<1, 1, 1, 1>>::BYTES : B6Geo<NetGeo<1, 2, 2, 1>>::BYTES; }
static size_t b6t_image_bytes(int size) { return size == NetGeo<1, 1, 1, 1>::SIZE ? B6TGeo<NetGeo<1, 1, 1, 1>>::BYTES : B6TGeo<NetGeo<1, 2, 2, 1>>::BYTES; }
// ctx->wimg of a resident RealNVP flow: [fp32 images][B6 images (forward chain)][B6T images (pair kernel's dX GEMMs)]
size_t nf_affine_wimg_bytes(const nf_flow_desc *desc) {
  if (nf_deep_geo_id(desc)) return nf_deep_wimg_bytes(desc);
  const int size = geo_size(desc);
  return (size_t)2 * desc->nlayers * 2 * ((size_t)size * sizeof(float) + (size ? b6_image_bytes(size) + b6t_image_bytes(size) : 0));
}
template <class G>
static size_t b6t_offset_bytes(const nf_flow_desc *desc) { return b6_offset_bytes<G>(desc) + (size_t)2 * desc->nlayers * 2 * B6Geo<G>::BYTES; }
template <class G>
static int b6t_refresh(nf_ctx *ctx, const nf_flow_desc *desc) {
  if (ctx->b6t_gen == ctx->wimg_gen) return NF_OK;
  using B = B6TGeo<G>;
  const int nimg = 2 * desc->nlayers * 2;
  constexpr long PER = 2 * G::CB * 2 * B::R3 + 2 * G::H2B * 2 * B::R2 + 2 * G::H1B * 2 * B::R1;
  const long total = (long)nimg * PER;
  ProfScope ps(ctx, "pack_weights");
  hipLaunchKernelGGL((k_b6t_from_images<G>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, nimg, (const float *)ctx->wimg,
                     (unsigned char *)ctx->wimg + b6t_offset_bytes<G>(desc));
  NF_HIP(hipGetLastError());
  ctx->b6t_gen = ctx->wimg_gen;
  return NF_OK;
}

// the bf16-triple images of the chain kernels that use the six-term products (B6), rebuilt from the fp32 images when those
// have been rewritten since (ctx->wimg_gen): behind the fp32 images in ctx->wimg
template <class G>
static int b6_refresh(nf_ctx *ctx, const nf_flow_desc *desc) {
  if (ctx->b6_gen == ctx->wimg_gen) return NF_OK;
  using B = B6Geo<G>;
  const int nimg = 2 * desc->nlayers * 2;
  constexpr long PER = 2 * G::MB * 2 * B::R1 + 2 * G::H1B * 2 * B::R2 + 2 * G::H2B * 2 * B::R3 + B::R1 + B::R2 + B::R3;
  const long total = (long)nimg * PER;
  ProfScope ps(ctx, "pack_weights");
  hipLaunchKernelGGL((k_b6_from_images<G>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, nimg, (const float *)ctx->wimg,
                     (unsigned char *)ctx->wimg + b6_offset_bytes<G>(desc));
  NF_HIP(hipGetLastError());
  ctx->b6_gen = ctx->wimg_gen;
  return NF_OK;
}

int nf_affine_pack(nf_ctx *ctx, const nf_flow_desc *desc, const float *theta) {
  if (nf_deep_geo_id(desc)) return nf_deep_pack(ctx, desc, theta);
  if (desc->n_hidden != 2) return NF_ERR_UNSUPPORTED;
  const int size = geo_size(desc);
  if (!size) return NF_ERR_UNSUPPORTED;
  const int nc = 2 * desc->nlayers;
  NF_TRY(nf_wimg_reserve(ctx, nf_affine_wimg_bytes(desc)));
  const PackArgs p = make_pack_args(desc);
  const long total = (long)nc * 2 * size;
  const unsigned grid = (unsigned)((total + 255) / 256);
  {
    ProfScope ps(ctx, "pack_weights");
    if (size == NetGeo<1, 1, 1, 1>::SIZE)
      hipLaunchKernelGGL((k_pack_net_images<NetGeo<1, 1, 1, 1>>), dim3(grid), dim3(256), 0, ctx->stream, p, theta, (float *)ctx->wimg);
    else
      hipLaunchKernelGGL((k_pack_net_images<NetGeo<1, 2, 2, 1>>), dim3(grid), dim3(256), 0, ctx->stream, p, theta, (float *)ctx->wimg);
    NF_HIP(hipGetLastError());
  }
  return NF_OK;
}

static int make_args(nf_ctx *ctx, const nf_flow_desc *desc, int k, const float *theta, long N, CouplingArgs *out) {
  if (desc->n_hidden != 2) return NF_ERR_UNSUPPORTED;
  CouplingInfo ci = nf_coupling_info(desc, k);
  CouplingArgs a;
  a.theta = theta;
  const int size = geo_size(desc);
  if (!size || !ctx->wimg) return NF_ERR_UNSUPPORTED;
  a.img_s = (const float *)ctx->wimg + (size_t)(2 * k) * size;
  a.img_t = a.img_s + size;
  a.trace = (long long *)ctx->trace;
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

template <class G, bool INV, bool FULL>
static int launch_apply_v(nf_ctx *ctx, const CouplingArgs &a, const float *x, float *y, float *ladj, int accumulate) {
  const size_t lds = 2 * (size_t)G::SIZE * sizeof(float);
  static AttrOnce attr_once;  // once per device: a context on another GPU needs its own
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_apply<G, INV, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  const long ntiles = (a.N + NF_TILE - 1) / NF_TILE;
  long grid = (ntiles + 7) / 8;
  const long cap = 2L * ctx->num_cu;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  ProfScope ps(ctx, "affine_apply");
  hipLaunchKernelGGL((k_affine_apply<G, INV, FULL>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a, x, y, ladj, accumulate);
  return (int)hipGetLastError();
}
template <class G, bool INV>
static int launch_apply(nf_ctx *ctx, const CouplingArgs &a, const float *x, float *y, float *ladj, int accumulate) {
  const bool full = a.m == 32 * G::MB && a.c == 32 * G::CB && a.N % NF_TILE == 0;
  return full ? launch_apply_v<G, INV, true>(ctx, a, x, y, ladj, accumulate)
              : launch_apply_v<G, INV, false>(ctx, a, x, y, ladj, accumulate);
}

int nf_affine_apply(nf_ctx *ctx, const nf_flow_desc *desc, int k, bool inverse, const float *theta, const float *x,
                    long N, float *y, float *ladj, int accumulate) {
  if (nf_deep_geo_id(desc)) {
    if (x != y) return NF_ERR_UNSUPPORTED;  // (every caller applies a coupling in place on the tiled buffer)
    return nf_deep_chain(ctx, desc, inverse, y, N, ladj, k, accumulate);
  }
  CouplingArgs a;
  NF_TRY(make_args(ctx, desc, k, theta, N, &a));
  const int mb = blocks32(a.m > a.c ? a.m : a.c), h1b = blocks32(desc->hdims[0]), h2b = blocks32(desc->hdims[1]), cb = mb;
#define BODY_APPLY                                                              \
  return inverse ? launch_apply<G, true>(ctx, a, x, y, ladj, accumulate)        \
                 : launch_apply<G, false>(ctx, a, x, y, ladj, accumulate);
  NF_GEO_DISPATCH(1, 1, 1, 1, BODY_APPLY)
  NF_GEO_DISPATCH(1, 2, 2, 1, BODY_APPLY)
#undef BODY_APPLY
  return NF_ERR_UNSUPPORTED;
}

template <class G, bool FULL>
static int launch_bwd_v(nf_ctx *ctx, const CouplingArgs &a, float *y, float *ybar, const float *lbar, float lbar_const,
                        float *slab, long slab_stride, int grid) {
  const size_t lds = BwdLds<G>::BYTES;
  static AttrOnce attr_once;  // once per device: a context on another GPU needs its own
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_bwd<G, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  ProfScope ps(ctx, "affine_bwd");
  hipLaunchKernelGGL((k_affine_bwd<G, FULL>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, y, ybar, lbar,
                     lbar_const, slab, slab_stride);
  return (int)hipGetLastError();
}
// Which stash layout the training step of this flow uses (forward writer and reverse reader must agree).  SLIM is OFF by
// default: measured A/B in one process on one box (cfg 2, driver's command, profiles/r4c_stash_slim_ab.txt) the forward gains
// 4 us of 199 (it is issue-bound, not write-bound: 124 fewer store instructions per coupling buy 2 %), the reverse kernel
// LOSES 52 us of 369 -- the 32 extra MFMAs per net and tile do not hide in the consumer's prologue, they wait there for x2 --
// 0.730 against 0.691 ms per step.  NF_STASH_SLIM=1 selects it where the two-waves-per-tile kernel reads the stash (hidden
// 33-64): 26 % less stash memory and HBM traffic for 5 % more time.
static bool stash_slim(int size) {
  static const bool no_pair = std::getenv("NF_BWD_NO_PAIR") != nullptr, slim = std::getenv("NF_STASH_SLIM") != nullptr;
  return size == NetGeo<1, 2, 2, 1>::SIZE && !no_pair && slim;
}

template <class G, bool SLIM = false, bool B6 = false>
static int launch_chain(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, float *xt, long N, float *ladj,
                        const FusedArgs *fused = nullptr, float *stash_plain = nullptr) {
  // two double-buffered (s,t) image pairs (B6: three single images) + target parameters and per-wave sums of the fused variant
  const size_t lds = (B6 ? (size_t)3 * B6Geo<G>::BYTES : 4 * (size_t)G::SIZE * sizeof(float)) + (2 * 64 * G::CB + 2) * sizeof(float) +
                     12 * sizeof(double);
  // (round 6, measured: NOT the default) the six-term chains WITHOUT a stash (nf_flow_fwd / nf_flow_inv, nf_loglikelihood: BASELINE cfg 5)
  // with THREE wavefronts per SIMD -- twelve tiles per workgroup, NF_CHAIN_NW=12 at build time: the kernel is a serial chain per wave
  // (GEMM -> split -> GEMM ...) at 53 % of the matrix pipe with two.  At hidden 64 the instantiation needs 212 registers; held to the 168
  // of three waves it spills 180 bytes and cfg 5 runs 1.893 against 1.624 ms; with the plain GEMM form (PIPE = false: 68 bytes) 1.747
  // against 1.675 (profiles/r6r_chain_nw.txt).  Hidden 32 fits (156).
#ifndef NF_CHAIN_NW
#define NF_CHAIN_NW 8
#endif
#ifndef NF_CHAIN_DUAL
#define NF_CHAIN_DUAL 0  // 1: the plain six-term chains on k_affine_chain_dual (one wave per SIMD, two tiles per wave) -- measured slower:
// cfg 5 1.72 against 1.63 ms (profiles/r6u_chain_dual.txt): with one wave per SIMD nothing covers the layer boundaries
#endif
  constexpr int NWP = B6 ? NF_CHAIN_NW : 8;  // waves per workgroup of the plain chains below
  static AttrOnce attr_once;  // once per device: a context on another GPU needs its own
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_chain<G, false, false, false, false, B6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_chain<G, true, false, false, false, B6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_chain<G, false, true, false, false, B6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_chain<G, false, true, true, SLIM, B6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_chain<G, false, false, true, SLIM, B6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_chain<G, true, false, true, SLIM, B6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (B6 && NF_CHAIN_DUAL) {
      NF_HIP(hipFuncSetAttribute((const void *)k_affine_chain_dual<G, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      NF_HIP(hipFuncSetAttribute((const void *)k_affine_chain_dual<G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (NWP != 8) {
      NF_HIP(hipFuncSetAttribute((const void *)k_affine_chain<G, false, false, false, false, B6, NWP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      NF_HIP(hipFuncSetAttribute((const void *)k_affine_chain<G, true, false, false, false, B6, NWP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    return NF_OK;
  }));
  if (B6) NF_TRY(b6_refresh<G>(ctx, desc));
  ChainArgs a;
  a.wimg = (const float *)ctx->wimg;
  a.wimg_b6 = (const unsigned char *)ctx->wimg + b6_offset_bytes<G>(desc);
  a.d = desc->d;
  a.ncoup = 2 * desc->nlayers;
  a.N = N;
  const long ngroups = ((N + NF_TILE - 1) / NF_TILE + 7) / 8;
  long grid = ngroups < ctx->num_cu ? ngroups : ctx->num_cu;
  if (grid < 1) grid = 1;
  ProfScope ps(ctx, "affine_chain");
  FusedArgs none{};
  none.trace = (long long *)ctx->trace;
  if (fused && fused->stash)
    hipLaunchKernelGGL((k_affine_chain<G, false, true, true, SLIM, B6>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a, xt, ladj, *fused);
  else if (fused)
    hipLaunchKernelGGL((k_affine_chain<G, false, true, false, false, B6>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a, xt, ladj, *fused);
  else if (inverse && stash_plain) {  // forward-KL training: the inverse chain leaves the stash of ITS reverse pass
    none.stash = stash_plain;
    hipLaunchKernelGGL((k_affine_chain<G, true, false, true, SLIM, B6>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a, xt, ladj, none);
  } else if (inverse && B6 && NF_CHAIN_DUAL) {
    hipLaunchKernelGGL((k_affine_chain_dual<G, true>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, xt, ladj);
  } else if (!inverse && !stash_plain && B6 && NF_CHAIN_DUAL) {
    hipLaunchKernelGGL((k_affine_chain_dual<G, false>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, xt, ladj);
  } else if (inverse) {
    const long g12 = (((N + NF_TILE - 1) / NF_TILE + NWP - 1) / NWP);
    const long gridp = g12 < 1 ? 1 : g12 < ctx->num_cu ? g12 : ctx->num_cu;
    hipLaunchKernelGGL((k_affine_chain<G, true, false, false, false, B6, NWP>), dim3((unsigned)gridp), dim3(64 * NWP), lds, ctx->stream, a, xt, ladj, none);
  }
  else if (stash_plain) {  // caller-supplied draws: the plain forward chain, leaving the stash behind
    none.stash = stash_plain;
    hipLaunchKernelGGL((k_affine_chain<G, false, false, true, SLIM, B6>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a, xt, ladj, none);
  }
  else {
    const long g12 = (((N + NF_TILE - 1) / NF_TILE + NWP - 1) / NWP);
    const long gridp = g12 < 1 ? 1 : g12 < ctx->num_cu ? g12 : ctx->num_cu;
    hipLaunchKernelGGL((k_affine_chain<G, false, false, false, false, B6, NWP>), dim3((unsigned)gridp), dim3(64 * NWP), lds, ctx->stream, a, xt, ladj, none);
  }
  return (int)hipGetLastError();
}

// number of workgroups (= entries of `partial`) the fused ELBO-forward launch uses
long nf_affine_chain_grid(nf_ctx *ctx, long N) {
  const long ngroups = ((N + NF_TILE - 1) / NF_TILE + 7) / 8;
  long grid = ngroups < ctx->num_cu ? ngroups : ctx->num_cu;
  return grid < 1 ? 1 : grid;
}

// base draws + whole chain forward + diagonal-Gaussian target + ELBO partial sums in one launch
// (packed images must be current).  yt <- flow output (tiled), gt <- gscale * grad log p(y) (or null),
// partial[nf_affine_chain_grid] <- sums of pscale * elbo_j.
int nf_affine_chain_elbo(nf_ctx *ctx, const nf_flow_desc *desc, long N, uint64_t seed, uint64_t off, uint32_t stream,
                         const float *mu, const float *var, float *yt, float *gt, double gscale, double *partial,
                         double pscale, float *stash, const uint32_t *stream_ptr) {
  const int size = geo_size(desc);
  if (!size || !ctx->wimg) return NF_ERR_UNSUPPORTED;
  FusedArgs fa;
  fa.stash = stash;
  fa.trace = (long long *)ctx->trace;
  fa.stream_ptr = stream_ptr;
  fa.k0 = (uint32_t)seed; fa.k1 = (uint32_t)(seed >> 32); fa.stream = stream; fa.off = off;
  fa.mu = mu; fa.var = var; fa.gt = gt; fa.gscale = (float)gscale; fa.partial = partial; fa.pscale = pscale;
  const bool b6 = fwd_b6(stash != nullptr);
  if (size == NetGeo<1, 1, 1, 1>::SIZE)
    return b6 ? launch_chain<NetGeo<1, 1, 1, 1>, false, true>(ctx, desc, false, yt, N, nullptr, &fa)
              : launch_chain<NetGeo<1, 1, 1, 1>>(ctx, desc, false, yt, N, nullptr, &fa);
  if (stash && stash_slim(size))
    return b6 ? launch_chain<NetGeo<1, 2, 2, 1>, true, true>(ctx, desc, false, yt, N, nullptr, &fa)
              : launch_chain<NetGeo<1, 2, 2, 1>, true>(ctx, desc, false, yt, N, nullptr, &fa);
  return b6 ? launch_chain<NetGeo<1, 2, 2, 1>, false, true>(ctx, desc, false, yt, N, nullptr, &fa)
            : launch_chain<NetGeo<1, 2, 2, 1>>(ctx, desc, false, yt, N, nullptr, &fa);
}

// whole chain in one launch, in place on the tiled buffer (packed images must be current)
int nf_affine_chain(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, float *xt, long N, float *ladj, float *stash) {
  if (nf_deep_geo_id(desc)) return stash ? NF_ERR_UNSUPPORTED : nf_deep_chain(ctx, desc, inverse, xt, N, ladj, -1, 0);
  const int size = geo_size(desc);
  if (!size || !ctx->wimg) return NF_ERR_UNSUPPORTED;
  const bool b6 = fwd_b6(stash != nullptr);
  if (size == NetGeo<1, 1, 1, 1>::SIZE)
    return b6 ? launch_chain<NetGeo<1, 1, 1, 1>, false, true>(ctx, desc, inverse, xt, N, ladj, nullptr, stash)
              : launch_chain<NetGeo<1, 1, 1, 1>>(ctx, desc, inverse, xt, N, ladj, nullptr, stash);
  if (stash && stash_slim(size))
    return b6 ? launch_chain<NetGeo<1, 2, 2, 1>, true, true>(ctx, desc, inverse, xt, N, ladj, nullptr, stash)
              : launch_chain<NetGeo<1, 2, 2, 1>, true>(ctx, desc, inverse, xt, N, ladj, nullptr, stash);
  return b6 ? launch_chain<NetGeo<1, 2, 2, 1>, false, true>(ctx, desc, inverse, xt, N, ladj, nullptr, stash)
            : launch_chain<NetGeo<1, 2, 2, 1>>(ctx, desc, inverse, xt, N, ladj, nullptr, stash);
}

template <class G>
static int launch_bwd(nf_ctx *ctx, const CouplingArgs &a, float *y, float *ybar, const float *lbar, float lbar_const,
                      float *slab, long slab_stride, int grid) {
  const bool full = a.m == 32 * G::MB && a.c == 32 * G::CB && a.N % NF_TILE == 0;
  return full ? launch_bwd_v<G, true>(ctx, a, y, ybar, lbar, lbar_const, slab, slab_stride, grid)
              : launch_bwd_v<G, false>(ctx, a, y, ybar, lbar, lbar_const, slab, slab_stride, grid);
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
  if (nf_deep_geo_id(desc)) return nf_deep_bwd(ctx, desc, k, k + 1, y, ybar, lbar, lbar_const, N, slab, slab_stride, grid, false);
  CouplingArgs a;
  NF_TRY(make_args(ctx, desc, k, theta, N, &a));
  const int mb = blocks32(a.m > a.c ? a.m : a.c), h1b = blocks32(desc->hdims[0]), h2b = blocks32(desc->hdims[1]), cb = mb;
#define BODY_BWD return launch_bwd<G>(ctx, a, y, ybar, lbar, lbar_const, slab + (long)k * 2 * G::SIZE, slab_stride, grid);
  NF_GEO_DISPATCH(1, 1, 1, 1, BODY_BWD)
  NF_GEO_DISPATCH(1, 2, 2, 1, BODY_BWD)
#undef BODY_BWD
  return NF_ERR_UNSUPPORTED;
}

template <class G, bool FULL, bool INVD = false>
static int launch_bwd_all_v(nf_ctx *ctx, const BwdAllArgs &aa, float *y, float *ybar, const float *lbar, float lbar_const,
                            float *slab, long slab_stride, int grid) {
  const size_t lds = BwdLds<G>::BYTES;
  static AttrOnce attr_once;  // once per device: a context on another GPU needs its own
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_bwd_all<G, FULL, INVD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  ProfScope ps(ctx, INVD ? "affine_bwd_inv" : "affine_bwd");
  hipLaunchKernelGGL((k_affine_bwd_all<G, FULL, INVD>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, aa, y, ybar, lbar,
                     lbar_const, slab, slab_stride);
  return (int)hipGetLastError();
}

// reverse pass of ALL couplings in one launch (flat order = reverse of execution order); packed images
// must be current.  Slab layout as with nf_affine_bwd called for k = 0 .. ncoup-1.
// inv_dir: reverse pass of the INVERSE chain instead (y: T^-1(data) -> data, couplings in execution order).
int nf_affine_bwd_all(nf_ctx *ctx, const nf_flow_desc *desc, float *y, float *ybar, const float *lbar, float lbar_const,
                      long N, float *slab, long slab_stride, int grid, bool inv_dir) {
  if (nf_deep_geo_id(desc)) return nf_deep_bwd(ctx, desc, 0, 2 * desc->nlayers, y, ybar, lbar, lbar_const, N, slab, slab_stride, grid, inv_dir);
  const int size = geo_size(desc);
  if (!size || !ctx->wimg || desc->n_hidden != 2) return NF_ERR_UNSUPPORTED;
  BwdAllArgs aa;
  aa.wimg = (const float *)ctx->wimg;
  aa.wimg_b6t = nullptr;
  aa.trace = (long long *)ctx->trace;
  aa.d = desc->d;
  aa.ncoup = 2 * desc->nlayers;
  aa.N = N;
  const bool h64 = size != NetGeo<1, 1, 1, 1>::SIZE;
  const bool full = desc->d == 64 && N % NF_TILE == 0;  // both partitions fill their 32-row block
  if (inv_dir) {
    if (h64)
      return full ? launch_bwd_all_v<NetGeo<1, 2, 2, 1>, true, true>(ctx, aa, y, ybar, lbar, lbar_const, slab, slab_stride, grid)
                  : launch_bwd_all_v<NetGeo<1, 2, 2, 1>, false, true>(ctx, aa, y, ybar, lbar, lbar_const, slab, slab_stride, grid);
    return launch_bwd_all_v<NetGeo<1, 1, 1, 1>, false, true>(ctx, aa, y, ybar, lbar, lbar_const, slab, slab_stride, grid);
  }
  if (h64)
    return full ? launch_bwd_all_v<NetGeo<1, 2, 2, 1>, true>(ctx, aa, y, ybar, lbar, lbar_const, slab, slab_stride, grid)
                : launch_bwd_all_v<NetGeo<1, 2, 2, 1>, false>(ctx, aa, y, ybar, lbar, lbar_const, slab, slab_stride, grid);
  return full ? launch_bwd_all_v<NetGeo<1, 1, 1, 1>, true>(ctx, aa, y, ybar, lbar, lbar_const, slab, slab_stride, grid)
              : launch_bwd_all_v<NetGeo<1, 1, 1, 1>, false>(ctx, aa, y, ybar, lbar, lbar_const, slab, slab_stride, grid);
}

// floats of the training step's activation stash (0: shape without a stash kernel)
size_t nf_affine_stash_floats(const nf_flow_desc *desc, long N) {
  const int size = geo_size(desc);
  if (!size || desc->n_hidden != 2) return 0;
  const size_t ntiles = (size_t)((N + NF_TILE - 1) / NF_TILE);
  const size_t per = size == NetGeo<1, 1, 1, 1>::SIZE ? StashGeo<NetGeo<1, 1, 1, 1>>::SIZE
                     : stash_slim(size) ? StashGeo<NetGeo<1, 2, 2, 1>, true>::SIZE : StashGeo<NetGeo<1, 2, 2, 1>>::SIZE;
  return ntiles * (size_t)(2 * desc->nlayers) * per;
}

// Is the stash the faster reverse pass for this shape?  Measured A/B on one box at batch 65 536, d = 64: hidden 64
// (NetGeo<1,2,2,1>) 0.667 vs 0.734 ms per step; hidden 32 (NetGeo<1,1,1,1>) 0.381 vs 0.370 ms -- the narrow nets' recompute
// is cheaper than their stash's HBM round trip, so there the stash is used only on request (nf_ctx_set_stash_budget > 0).
bool nf_affine_stash_pays(const nf_flow_desc *desc) { return geo_size(desc) == NetGeo<1, 2, 2, 1>::SIZE; }

template <class G, bool FULL, bool INVD = false>
static int launch_bwd_stashed_v(nf_ctx *ctx, const BwdAllArgs &aa0, float *stash, float *ybar, const float *lbar, float lbar_const,
                                float *slab, long slab_stride, int grid, const nf_flow_desc *desc = nullptr) {
  const size_t lds = BwdStashLds<G>::BYTES;
  static AttrOnce attr_once;  // once per device
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_bwd_stashed<G, FULL, INVD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_bwd_stashed<G, FULL, INVD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  // NF_BWD_ONE_WAVE_B6=1 (with NF_BWD_NO_PAIR=1 for hidden 33-64): every GEMM of the one-wave kernel on the bf16 matrix cores
  static const bool sb6 = std::getenv("NF_BWD_ONE_WAVE_B6") != nullptr;
  BwdAllArgs aa = aa0;
  ProfScope ps(ctx, INVD ? "affine_bwd_inv" : "affine_bwd");
  if (sb6 && desc) {
    NF_TRY(b6t_refresh<G>(ctx, desc));
    aa.wimg_b6t = (const unsigned char *)ctx->wimg + b6t_offset_bytes<G>(desc);
    hipLaunchKernelGGL((k_affine_bwd_stashed<G, FULL, INVD, true>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, aa, stash, ybar, lbar,
                       lbar_const, slab, slab_stride);
  } else
    hipLaunchKernelGGL((k_affine_bwd_stashed<G, FULL, INVD>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, aa, stash, ybar, lbar,
                       lbar_const, slab, slab_stride);
  return (int)hipGetLastError();
}

template <class G, bool FULL, bool INVD, bool SLIM, bool PB6, bool DW6 = false>
static int launch_bwd_pair_v(nf_ctx *ctx, const nf_flow_desc *desc, BwdAllArgs aa, float *stash, float *ybar, const float *lbar,
                             float lbar_const, float *slab, long slab_stride, int grid) {
  const size_t lds = BwdPairLds<G, PB6, DW6>::BYTES;
  static AttrOnce attr_once;  // once per device
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_bwd_pair<G, FULL, INVD, SLIM, PB6, DW6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  if (PB6) {
    NF_TRY(b6t_refresh<G>(ctx, desc));
    aa.wimg_b6t = (const unsigned char *)ctx->wimg + b6t_offset_bytes<G>(desc);
  }
  ProfScope ps(ctx, INVD ? "affine_bwd_inv" : "affine_bwd");
  hipLaunchKernelGGL((k_affine_bwd_pair<G, FULL, INVD, SLIM, PB6, DW6>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, aa, stash, ybar, lbar,
                     lbar_const, slab, slab_stride);
  return (int)hipGetLastError();
}
// NF_BWD_FP32=1: the A/B switch back to fp32-MFMA GEMMs in the pair kernel; NF_BWD_DW_FP32=1: only the consumer's dW GEMMs
template <class G, bool FULL, bool INVD>
static int launch_bwd_pair(nf_ctx *ctx, const nf_flow_desc *desc, const BwdAllArgs &aa, float *stash, float *ybar, const float *lbar,
                           float lbar_const, float *slab, long slab_stride, int grid) {
  static const bool fp32 = std::getenv("NF_BWD_FP32") != nullptr;
  if (stash_slim(G::SIZE)) return launch_bwd_pair_v<G, FULL, INVD, true, false>(ctx, desc, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid);
  static const bool dw_fp32 = std::getenv("NF_BWD_DW_FP32") != nullptr;
  if (fp32) return launch_bwd_pair_v<G, FULL, INVD, false, false>(ctx, desc, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid);
  return dw_fp32 ? launch_bwd_pair_v<G, FULL, INVD, false, true>(ctx, desc, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid)
                 : launch_bwd_pair_v<G, FULL, INVD, false, true, true>(ctx, desc, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid);
}

// reverse pass of all couplings from the stash nf_affine_chain_elbo(..., stash) left (same slab layout as
// nf_affine_bwd_all; ybar: cotangent of the flow output on entry, of the flow input on exit)
int nf_affine_bwd_stashed(nf_ctx *ctx, const nf_flow_desc *desc, float *stash, float *ybar, const float *lbar, float lbar_const,
                          long N, float *slab, long slab_stride, int grid, bool inv_dir) {
  const int size = geo_size(desc);
  if (!size || !ctx->wimg || desc->n_hidden != 2 || !stash) return NF_ERR_UNSUPPORTED;
  BwdAllArgs aa;
  aa.wimg = (const float *)ctx->wimg;
  aa.wimg_b6t = nullptr;
  aa.trace = (long long *)ctx->trace;
  aa.d = desc->d;
  aa.ncoup = 2 * desc->nlayers;
  aa.N = N;
  const bool h64 = size != NetGeo<1, 1, 1, 1>::SIZE;
  const bool full = desc->d == 64 && N % NF_TILE == 0;
  // hidden 33-64: the two-waves-per-tile kernel (both directions, ragged batches and d < 64 included)
  static const bool no_pair = std::getenv("NF_BWD_NO_PAIR") != nullptr;  // A/B switch: k_affine_bwd_stashed
  // (hidden <= 32 was measured with this kernel too: 226.9 against 227.8 us at d = 64, batch 65 536 -- no gain, not instantiated)
  if (h64 && !no_pair) {
    using GP = NetGeo<1, 2, 2, 1>;
    if (inv_dir)
      return full ? launch_bwd_pair<GP, true, true>(ctx, desc, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid)
                  : launch_bwd_pair<GP, false, true>(ctx, desc, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid);
    return full ? launch_bwd_pair<GP, true, false>(ctx, desc, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid)
                : launch_bwd_pair<GP, false, false>(ctx, desc, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid);
  }
  if (inv_dir) {  // forward-KL training: the stash of the inverse chain (nf_affine_chain(inverse, stash))
    if (h64)
      return full ? launch_bwd_stashed_v<NetGeo<1, 2, 2, 1>, true, true>(ctx, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid, desc)
                  : launch_bwd_stashed_v<NetGeo<1, 2, 2, 1>, false, true>(ctx, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid, desc);
    return launch_bwd_stashed_v<NetGeo<1, 1, 1, 1>, false, true>(ctx, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid, desc);
  }
  if (h64)
    return full ? launch_bwd_stashed_v<NetGeo<1, 2, 2, 1>, true>(ctx, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid, desc)
                : launch_bwd_stashed_v<NetGeo<1, 2, 2, 1>, false>(ctx, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid, desc);
  return full ? launch_bwd_stashed_v<NetGeo<1, 1, 1, 1>, true>(ctx, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid, desc)
              : launch_bwd_stashed_v<NetGeo<1, 1, 1, 1>, false>(ctx, aa, stash, ybar, lbar, lbar_const, slab, slab_stride, grid, desc);
}

// the two-hidden-layer kernels with their fused forward (draws + chain + target in one launch), stash and epilogue
bool nf_affine_fused_ok(const nf_flow_desc *desc) { return geo_size(desc) != 0; }
bool nf_affine_supported(const nf_flow_desc *desc) {
  if (nf_deep_geo_id(desc)) return true;
  if (desc->n_hidden != 2) return false;
  const int c = (desc->d + 1) / 2;  // larger of the two partitions
  const int mb = blocks32(c), h1b = blocks32(desc->hdims[0]), h2b = blocks32(desc->hdims[1]), cb = mb;
  (void)cb;
  return (mb == 1 && h1b == 1 && h2b == 1) || (mb == 1 && h1b == 2 && h2b == 2);
}
