// nf_coupling16.hip -- reverse pass of one AffineCoupling on 16-sample half-tiles with
// v_mfma_f32_16x16x4_f32, TWO wavefronts per SIMD.
//
// Why a second reverse-pass kernel: in k_affine_bwd (32-sample tiles, v_mfma_f32_32x32x2_f32) a
// wave needs 29.6 KB of LDS for the [feature][sample] stash that feeds the weight-gradient GEMM,
// so only four waves (one per SIMD) fit next to the 34 KB weight image, and with one wave per
// SIMD nothing overlaps the VALU/LDS epilogues between GEMMs: the matrix pipe idles ~40 % of the
// time (profiles/r1b_pmc_summary.json).  Halving the tile halves the stash (15.2 KB) and the
// activation registers, so eight waves fit in LDS and in 256 registers each, and the second
// wave on every SIMD fills the first one's bubbles.  Same arithmetic, same reference semantics
// (src/flows/realnvp.jl:57-110) as nf_coupling.hip.
//
// Register layout ("C16"): the C/D layout of v_mfma_f32_16x16x4_f32 --
//   lane l, register r (0..3)  <->  sample s = l & 15, feature f = 16*block + 4*(l >> 4) + r.
// As in nf_mfma.h the contraction order is chosen so that accumulator register t of one layer is
// the B operand of k-step t of the next: k_t(lane) = 4*(l >> 4) + t.
#include "nf_common.h"
#include "nf_mfma.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define HT 16      // samples per wave (half of a 32-sample memory tile)
#define TS16 17    // stash row stride: conflict-free operand reads

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// out[ob] = W * in + b.  IB / OB count 16-feature blocks.  Image layout [in][out], row stride S
// with S = 4 (mod 8): the two rows a 32-lane group touches sit 16 banks apart.
template <int IB, int OB, int S>
__device__ __forceinline__ void dense16_fwd(const float *__restrict__ w, const float *__restrict__ b,
                                            const f32x4 (&in)[IB], f32x4 (&out)[OB], int s, int qd) {
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) out[ob] = *reinterpret_cast<const f32x4 *>(b + 16 * ob + 4 * qd);
  const float *wl = w + (4 * qd) * S + s;
  constexpr int NG = IB * 2;  // groups of two k-steps
  float an[2][OB], ac[2][OB];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) an[u][ob] = wl[u * S + 16 * ob];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) ac[u][ob] = an[u][ob];
    if (g + 1 < NG) {
      const int kb = (g + 1) >> 1, t0 = ((g + 1) & 1) * 2;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) an[u][ob] = wl[(16 * kb + t0 + u) * S + 16 * ob];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) out[ob] = mfma16(ac[u][ob], in[g >> 1][(g & 1) * 2 + u], out[ob]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// din[ib] = W^T * delta.  The four k-steps of an out-block are contiguous along `out` in the image,
// so each (ib, ob) pair is ONE 16-byte LDS read.
template <int IB, int OB, int S>
__device__ __forceinline__ void dense16_bwd_x(const float *__restrict__ w, const f32x4 (&delta)[OB], f32x4 (&din)[IB],
                                              int s, int qd) {
#pragma unroll
  for (int ib = 0; ib < IB; ++ib) din[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float *wl = w + s * S + 4 * qd;
  // A operands are fetched two in-blocks at a time (8 registers in flight); with two waves per SIMD
  // the partner wave covers the LDS latency, so no deeper software pipeline is needed here
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
    for (int ib0 = 0; ib0 < IB; ib0 += 2) {
      f32x4 ac[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (ib0 + u < IB) ac[u] = *reinterpret_cast<const f32x4 *>(wl + 16 * (ib0 + u) * S + 16 * ob);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (ib0 + u < IB) din[ib0 + u] = mfma16(ac[u][t], delta[ob][t], din[ib0 + u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int NB>
__device__ __forceinline__ void stash16(float *__restrict__ sc, const f32x4 (&v)[NB], int s, int qd) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[(16 * b + 4 * qd + r) * TS16 + s] = v[b][r];
}

// acc[ib][ob] (col = out feature, rows = in features) += sum over the 16 samples of a x delta^T
template <int IB, int OB>
__device__ __forceinline__ void dw16(const float *__restrict__ sa, const float *__restrict__ sd, f32x4 (&acc)[IB][OB],
                                     float (&bsum)[OB], int s, int qd) {
  const float *pa = sa + s * TS16 + qd;  // lane <-> feature s of a block, sample 4*ks + qd
  const float *pd = sd + s * TS16 + qd;
  float an[IB], dn[OB], ac[IB], dc[OB];
#pragma unroll
  for (int ib = 0; ib < IB; ++ib) an[ib] = pa[16 * ib * TS16];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) dn[ob] = pd[16 * ob * TS16];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) ac[ib] = an[ib];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) dc[ob] = dn[ob];
    if (ks + 1 < 4) {
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) an[ib] = pa[16 * ib * TS16 + 4 * (ks + 1)];
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) dn[ob] = pd[16 * ob * TS16 + 4 * (ks + 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) bsum[ob] += dc[ob];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) acc[ib][ob] = mfma16(ac[ib], dc[ob], acc[ib][ob]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// half-tile I/O: same buffer-descriptor scheme as TileIO (nf_mfma.h) with the C16 feature map
//   f = 2 * (16 b + 4 qd + r) + parity  ->  byte offset b*4096 + qd*1024 + r*256 + parity*128
struct TileIO16 {
  __amdgpu_buffer_rsrc_t rs;
  int voff;
};
__device__ __forceinline__ TileIO16 make_tile_io16(float *array, long tile, int half, int d, int s, int qd) {
  TileIO16 t;
  t.rs = __builtin_amdgcn_make_buffer_rsrc(array + tile * d * NF_TILE, 0, d * NF_TILE * 4, 0x00020000);
  t.voff = (half * HT + s) * 4 + qd * 1024;
  return t;
}
__device__ __forceinline__ int soff16(int b, int r, int parity) { return b * 4096 + r * 256 + parity * 128; }
__device__ __forceinline__ float load16(const TileIO16 &t, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.rs, t.voff, soff, 0));
}
__device__ __forceinline__ void store16(const TileIO16 &t, int soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), t.rs, t.voff, soff, 0);
}

#define TS16_STAMP(slot)                                                \
  do {                                                                   \
    if (tr) { __builtin_amdgcn_sched_barrier(0); tr[slot] = clock64(); } \
  } while (0)

struct Bwd16Args {
  long long *trace;
  const float *img_s, *img_t;  // packed images, PAD = 4 layout
  int d, c, m, par_t;
  long N;
};

template <class G>
struct Geo16 {  // 16-feature block counts and LDS plan
  static constexpr int M = 2 * G::MB, H1 = 2 * G::H1B, H2 = 2 * G::H2B, C = 2 * G::CB;
  static constexpr int DB = (H1 > H2 ? (H1 > C ? H1 : C) : (H2 > C ? H2 : C));
  static constexpr int OFF_X = 0;
  static constexpr int OFF_A1 = OFF_X + 16 * M * TS16;
  static constexpr int OFF_A2 = OFF_A1 + 16 * H1 * TS16;
  static constexpr int OFF_D = OFF_A2 + 16 * H2 * TS16;
  static constexpr int SCRATCH = ((OFF_D + 16 * DB * TS16 + 3) / 4) * 4;  // floats per wave
  static constexpr int WAVES = 8;
  static constexpr int BODY = G::SIZE + WAVES * SCRATCH;
  static constexpr int FOLD = 4 * G::SIZE;
  static constexpr size_t BYTES = (size_t)(BODY > FOLD ? BODY : FOLD) * sizeof(float);
};

template <class G>
struct Acc16 {
  f32x4 w1[2 * G::MB][2 * G::H1B];
  f32x4 w2[2 * G::H1B][2 * G::H2B];
  f32x4 w3[2 * G::H2B][2 * G::CB];
  float b1[2 * G::H1B], b2[2 * G::H2B], b3[2 * G::CB];
};

template <int IB, int OB>
__device__ __forceinline__ void zero16(f32x4 (&a)[IB][OB], float (&b)[OB]) {
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int o = 0; o < OB; ++o) a[i][o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < OB; ++o) b[o] = 0.f;
}

// write (first) or add one wave's accumulators into an image-layout region
template <int IB, int OB, int S>
__device__ __forceinline__ void fold16(float *__restrict__ w, float *__restrict__ b, const f32x4 (&a)[IB][OB],
                                       const float (&bs)[OB], bool first, int s, int qd) {
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float *p = w + (16 * i + 4 * qd + r) * S + 16 * o + s;
        *p = first ? a[i][o][r] : *p + a[i][o][r];
      }
#pragma unroll
  for (int o = 0; o < OB; ++o) {
    float v = bs[o];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (qd == 0) {
      float *p = b + 16 * o + s;
      *p = first ? v : *p + v;
    }
  }
}

template <int NB>
__device__ __forceinline__ unsigned signs16(const f32x4 (&v)[NB]) {
  unsigned bits = 0;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) bits |= (__float_as_int(v[b][r]) < 0 ? 1u : 0u) << (4 * b + r);
  return bits;
}

template <class G, bool PHASE_S, bool FULL>
__device__ __forceinline__ void bwd16_tile(const Bwd16Args &a, const float *__restrict__ img, float *__restrict__ sc,
                                           Acc16<G> &acc, float *__restrict__ y, float *__restrict__ ybar,
                                           const float *__restrict__ lbar, float lbar_const, long htile, int s, int qd,
                                           long long *tr) {
  using L = Geo16<G>;
  TS16_STAMP(0);
  const long tile = htile >> 1;
  const int half = (int)(htile & 1);
  const long j = tile * NF_TILE + half * HT + s;
  const bool valid = FULL ? true : j < a.N;
  const int par_c = 1 - a.par_t;
  const TileIO16 yio = make_tile_io16(y, tile, half, a.d, s, qd);
  const TileIO16 gio = make_tile_io16(ybar, tile, half, a.d, s, qd);
  float *sd = sc + L::OFF_D;

  f32x4 d3[L::C], y1[L::C], g1[L::C];
  unsigned m1, m2;
  {
    f32x4 xb[L::M];
#pragma unroll
    for (int b = 0; b < L::M; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = load16(yio, soff16(b, r, par_c));  // features >= d read as 0
        xb[b][r] = valid ? v : 0.f;
      }
    stash16<L::M>(sc + L::OFF_X, xb, s, qd);
    TS16_STAMP(1);
    f32x4 a1[L::H1];
    dense16_fwd<L::M, L::H1, G::S1>(img + G::W1, img + G::B1, xb, a1, s, qd);
#pragma unroll
    for (int b = 0; b < L::H1; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) a1[b][r] = nf_lrelu(a1[b][r]);
    m1 = signs16<L::H1>(a1);
    stash16<L::H1>(sc + L::OFF_A1, a1, s, qd);
    f32x4 a2[L::H2];
    dense16_fwd<L::H1, L::H2, G::S2>(img + G::W2, img + G::B2, a1, a2, s, qd);
#pragma unroll
    for (int b = 0; b < L::H2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) a2[b][r] = nf_lrelu(a2[b][r]);
    m2 = signs16<L::H2>(a2);
    stash16<L::H2>(sc + L::OFF_A2, a2, s, qd);
    dense16_fwd<L::H2, L::C, G::S3>(img + G::W3, img + G::B3, a2, d3, s, qd);  // T, or pre-tanh S
  }
  TS16_STAMP(2);
  // operands of the element-wise stage (the second wave on this SIMD covers their latency)
#pragma unroll
  for (int b = 0; b < L::C; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      y1[b][r] = load16(yio, soff16(b, r, a.par_t));
      g1[b][r] = load16(gio, soff16(b, r, a.par_t));
    }

  const float lb = valid ? (lbar ? lbar[FULL ? j : (j < a.N ? j : 0)] : lbar_const) : 0.f;
#pragma unroll
  for (int b = 0; b < L::C; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = 16 * b + 4 * qd + r;
      const bool ok = (p < a.c) && valid;
      const float yv = y1[b][r], gv = g1[b][r];
      if (!PHASE_S) {
        store16(yio, soff16(b, r, a.par_t), yv - d3[b][r]);  // u = x1 * exp(S)
        d3[b][r] = ok ? gv : 0.f;                             // T-bar = ybar1
      } else {
        const float sv = nf_tanh(d3[b][r]);
        const float es = nf_exp(sv);
        store16(yio, soff16(b, r, a.par_t), __fdividef(yv, es));  // x1 = u * exp(-s)
        store16(gio, soff16(b, r, a.par_t), gv * es);             // x1bar
        d3[b][r] = ok ? (gv * yv + lb) * (1.f - sv * sv) : 0.f;   // S-bar through tanh
      }
    }

  TS16_STAMP(3);
  // ---- layer 3
  stash16<L::C>(sd, d3, s, qd);
  wave_lds_fence();
  TS16_STAMP(4);
  dw16<L::H2, L::C>(sc + L::OFF_A2, sd, acc.w3, acc.b3, s, qd);
  TS16_STAMP(5);
  f32x4 d2[L::H2];
  dense16_bwd_x<L::H2, L::C, G::S3>(img + G::W3, d3, d2, s, qd);
#pragma unroll
  for (int b = 0; b < L::H2; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) d2[b][r] *= ((m2 >> (4 * b + r)) & 1u) ? 0.01f : 1.f;
  wave_lds_fence();
  TS16_STAMP(6);
  // ---- layer 2
  stash16<L::H2>(sd, d2, s, qd);
  wave_lds_fence();
  dw16<L::H1, L::H2>(sc + L::OFF_A1, sd, acc.w2, acc.b2, s, qd);
  TS16_STAMP(7);
  f32x4 d1[L::H1];
  dense16_bwd_x<L::H1, L::H2, G::S2>(img + G::W2, d2, d1, s, qd);
#pragma unroll
  for (int b = 0; b < L::H1; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) d1[b][r] *= ((m1 >> (4 * b + r)) & 1u) ? 0.01f : 1.f;
  wave_lds_fence();
  TS16_STAMP(8);
  // ---- layer 1
  stash16<L::H1>(sd, d1, s, qd);
  wave_lds_fence();
  dw16<L::M, L::H1>(sc + L::OFF_X, sd, acc.w1, acc.b1, s, qd);
  TS16_STAMP(9);
  f32x4 g2[L::M], gold[L::M];
#pragma unroll
  for (int b = 0; b < L::M; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) gold[b][r] = load16(gio, soff16(b, r, par_c));
  dense16_bwd_x<L::M, L::H1, G::S1>(img + G::W1, d1, g2, s, qd);
  wave_lds_fence();
#pragma unroll
  for (int b = 0; b < L::M; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) store16(gio, soff16(b, r, par_c), gold[b][r] + g2[b][r]);
  TS16_STAMP(10);
}

template <class G, bool FULL>
__global__ __launch_bounds__(512, 2) void k_affine_bwd16(Bwd16Args a, float *__restrict__ y, float *__restrict__ ybar,
                                                         const float *__restrict__ lbar, float lbar_const,
                                                         float *__restrict__ slab, long slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using L = Geo16<G>;
  float *img = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = lane & 15, qd = lane >> 4;
  float *sc = lds + G::SIZE + wave * L::SCRATCH;
  const long nhalf = 2 * ((a.N + NF_TILE - 1) / NF_TILE);
  long long *tr0 = (a.trace && blockIdx.x == 0 && tid == 0) ? a.trace : nullptr;
  if (tr0) tr0[0] = clock64();

#pragma unroll 1
  for (int phase = 0; phase < 2; ++phase) {
    long long *tr = tr0 ? tr0 + 8 + phase * 40 : nullptr;
    stage_packed<G::SIZE, 512>(img, phase == 0 ? a.img_t : a.img_s, tid);
    __syncthreads();
    if (tr0) tr0[1 + phase * 3] = clock64();
    Acc16<G> acc;
    zero16(acc.w1, acc.b1);
    zero16(acc.w2, acc.b2);
    zero16(acc.w3, acc.b3);
#pragma unroll 1
    for (long ht = (long)blockIdx.x * 8 + wave; ht < nhalf; ht += (long)gridDim.x * 8) {
      if (phase == 0)
        bwd16_tile<G, false, FULL>(a, img, sc, acc, y, ybar, lbar, lbar_const, ht, s, qd, tr);
      else
        bwd16_tile<G, true, FULL>(a, img, sc, acc, y, ybar, lbar, lbar_const, ht, s, qd, tr);
      if (tr) tr += 12;
    }
    if (tr0) tr0[2 + phase * 3] = clock64();
    __syncthreads();  // weights and stashes are dead: LDS becomes four image-sized fold regions
    // waves 0-3 write their accumulators into regions 0-3, then waves 4-7 add theirs (fixed order:
    // deterministic), then all threads sum the four regions into the workgroup's slab
#pragma unroll 1
    for (int round = 0; round < 2; ++round) {
      if ((wave >> 2) == round) {
        float *mine = lds + (wave & 3) * G::SIZE;
        fold16<2 * G::MB, 2 * G::H1B, G::S1>(mine + G::W1, mine + G::B1, acc.w1, acc.b1, round == 0, s, qd);
        fold16<2 * G::H1B, 2 * G::H2B, G::S2>(mine + G::W2, mine + G::B2, acc.w2, acc.b2, round == 0, s, qd);
        fold16<2 * G::H2B, 2 * G::CB, G::S3>(mine + G::W3, mine + G::B3, acc.w3, acc.b3, round == 0, s, qd);
      }
      __syncthreads();
    }
    {
      const float4 *c0 = reinterpret_cast<const float4 *>(lds);
      float4 *dst = reinterpret_cast<float4 *>(slab + ((long)blockIdx.x * slab_stride + (phase == 0 ? 1 : 0) * (long)G::SIZE));
      constexpr int NV4 = G::SIZE / 4;
      for (int i = tid; i < NV4; i += 512) {
        const float4 p0 = c0[i], p1 = c0[i + NV4], p2 = c0[i + 2 * NV4], p3 = c0[i + 3 * NV4];
        float4 r;
        r.x = (p0.x + p1.x) + (p2.x + p3.x);
        r.y = (p0.y + p1.y) + (p2.y + p3.y);
        r.z = (p0.z + p1.z) + (p2.z + p3.z);
        r.w = (p0.w + p1.w) + (p2.w + p3.w);
        dst[i] = r;
      }
    }
    __syncthreads();
    if (tr0) tr0[3 + phase * 3] = clock64();
  }
}

// ------------------------------------------------------------------------------------
// packing (PAD = 4 images) and slab reduction for this kernel's image layout
// ------------------------------------------------------------------------------------
struct Pack16Args {
  int d, h1, h2, ncoup;
  long pair_params, odd_params;
};
template <class G>
__device__ __forceinline__ NetDims dims16(const Pack16Args &p, int img) {
  const int k = img >> 1, net = img & 1;
  const int c = (k & 1) ? p.d / 2 : (p.d + 1) / 2, m = p.d - c;
  long off = (long)(k >> 1) * p.pair_params + ((k & 1) ? p.odd_params : 0);
  if (net) off += net_param_count(m, p.h1, p.h2, c);
  return make_net_dims(off, m, p.h1, p.h2, c);
}
template <class G>
__global__ __launch_bounds__(256) void k_pack16(Pack16Args p, const float *__restrict__ theta, float *__restrict__ out) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)p.ncoup * 2 * G::SIZE) return;
  const int img = (int)(gid / G::SIZE), e = (int)(gid - (long)img * G::SIZE);
  const long ti = e < G::B3 + 32 * G::CB ? image_theta_index<G>(dims16<G>(p, img), e) : -1;
  out[gid] = ti >= 0 ? theta[ti] : 0.f;
}
template <class G>
__global__ __launch_bounds__(256) void k_reduce16(Pack16Args p, const float *__restrict__ slab, int nslab, long stride,
                                                  float *__restrict__ g) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)p.ncoup * 2 * G::SIZE) return;
  const int img = (int)(gid / G::SIZE), e = (int)(gid - (long)img * G::SIZE);
  const long ti = e < G::B3 + 32 * G::CB ? image_theta_index<G>(dims16<G>(p, img), e) : -1;
  if (ti < 0) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int sI = 0;
  for (; sI + 3 < nslab; sI += 4) {
    a0 += slab[(long)sI * stride + gid];
    a1 += slab[(long)(sI + 1) * stride + gid];
    a2 += slab[(long)(sI + 2) * stride + gid];
    a3 += slab[(long)(sI + 3) * stride + gid];
  }
  for (; sI < nslab; ++sI) a0 += slab[(long)sI * stride + gid];
  g[ti] = (a0 + a1) + (a2 + a3);
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
using G16a = NetGeo<1, 1, 1, 1, 4>;
using G16b = NetGeo<1, 2, 2, 1, 4>;
static inline int bl32(int n) { return (n + 31) / 32; }
static int geo16_id(const nf_flow_desc *desc) {
  if (desc->kind != NF_KIND_REALNVP || desc->n_hidden != 2) return 0;
  const int mb = bl32((desc->d + 1) / 2), h1 = bl32(desc->hdims[0]), h2 = bl32(desc->hdims[1]);
  if (mb == 1 && h1 == 1 && h2 == 1) return 1;
  if (mb == 1 && h1 == 2 && h2 == 2) return 2;
  return 0;
}
static int geo16_size(const nf_flow_desc *desc) {
  const int id = geo16_id(desc);
  return id == 1 ? G16a::SIZE : (id == 2 ? G16b::SIZE : 0);
}
bool nf_bwd16_supported(const nf_flow_desc *desc) { return geo16_id(desc) != 0; }
long nf_bwd16_slab_floats(const nf_flow_desc *desc) { return (long)2 * desc->nlayers * 2 * geo16_size(desc); }

static Pack16Args pack16_args(const nf_flow_desc *desc) {
  Pack16Args p;
  p.d = desc->d; p.h1 = desc->hdims[0]; p.h2 = desc->hdims[1]; p.ncoup = 2 * desc->nlayers;
  const CouplingInfo c0 = nf_coupling_info(desc, 0), c1 = nf_coupling_info(desc, 1);
  p.odd_params = c0.nparams;
  p.pair_params = c0.nparams + c1.nparams;
  return p;
}

int nf_bwd16_pack(nf_ctx *ctx, const nf_flow_desc *desc, const float *theta) {
  const int id = geo16_id(desc);
  if (!id) return NF_ERR_UNSUPPORTED;
  const int size = geo16_size(desc);
  const size_t bytes = (size_t)2 * desc->nlayers * 2 * size * sizeof(float);
  if (bytes > ctx->wimg16_bytes) {
    NF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->wimg16) NF_HIP(hipFree(ctx->wimg16));
    ctx->wimg16 = nullptr;
    ctx->wimg16_bytes = 0;
    NF_HIP(hipMalloc(&ctx->wimg16, bytes));
    ctx->wimg16_bytes = bytes;
  }
  const Pack16Args p = pack16_args(desc);
  const long total = (long)p.ncoup * 2 * size;
  const unsigned grid = (unsigned)((total + 255) / 256);
  ProfScope ps(ctx, "pack_weights");
  if (id == 1)
    hipLaunchKernelGGL((k_pack16<G16a>), dim3(grid), dim3(256), 0, ctx->stream, p, theta, (float *)ctx->wimg16);
  else
    hipLaunchKernelGGL((k_pack16<G16b>), dim3(grid), dim3(256), 0, ctx->stream, p, theta, (float *)ctx->wimg16);
  return (int)hipGetLastError();
}

int nf_bwd16_reduce_slabs(nf_ctx *ctx, const nf_flow_desc *desc, const float *slab, int nslab, float *g) {
  const int id = geo16_id(desc);
  if (!id) return NF_ERR_UNSUPPORTED;
  const Pack16Args p = pack16_args(desc);
  const long total = (long)p.ncoup * 2 * geo16_size(desc);
  const unsigned grid = (unsigned)((total + 255) / 256);
  ProfScope ps(ctx, "reduce_slabs");
  if (id == 1)
    hipLaunchKernelGGL((k_reduce16<G16a>), dim3(grid), dim3(256), 0, ctx->stream, p, slab, nslab, total, g);
  else
    hipLaunchKernelGGL((k_reduce16<G16b>), dim3(grid), dim3(256), 0, ctx->stream, p, slab, nslab, total, g);
  return (int)hipGetLastError();
}

int nf_bwd16_grid(nf_ctx *ctx, long N) {
  const long nhalf = 2 * ((N + NF_TILE - 1) / NF_TILE);
  long grid = (nhalf + 7) / 8;
  if (grid > ctx->num_cu) grid = ctx->num_cu;
  return (int)(grid < 1 ? 1 : grid);
}

template <class G, bool FULL>
static int launch16(nf_ctx *ctx, const Bwd16Args &a, float *y, float *ybar, const float *lbar, float lbar_const,
                    float *slab, long stride, int grid) {
  const size_t lds = Geo16<G>::BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    NF_HIP(hipFuncSetAttribute((const void *)k_affine_bwd16<G, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  ProfScope ps(ctx, "affine_bwd");
  hipLaunchKernelGGL((k_affine_bwd16<G, FULL>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a, y, ybar, lbar,
                     lbar_const, slab, stride);
  return (int)hipGetLastError();
}

int nf_bwd16(nf_ctx *ctx, const nf_flow_desc *desc, int k, float *y, float *ybar, const float *lbar, float lbar_const,
             long N, float *slab, long stride, int grid) {
  const int id = geo16_id(desc);
  if (!id || !ctx->wimg16) return NF_ERR_UNSUPPORTED;
  const int size = geo16_size(desc);
  const CouplingInfo ci = nf_coupling_info(desc, k);
  Bwd16Args a;
  a.img_s = (const float *)ctx->wimg16 + (size_t)(2 * k) * size;
  a.img_t = a.img_s + size;
  a.d = desc->d; a.c = ci.c; a.m = ci.m; a.par_t = ci.par_t; a.N = N;
  a.trace = (long long *)ctx->trace;
  const bool full = N % NF_TILE == 0;
  float *sl = slab + (long)k * 2 * size;
  if (id == 1)
    return full ? launch16<G16a, true>(ctx, a, y, ybar, lbar, lbar_const, sl, stride, grid)
                : launch16<G16a, false>(ctx, a, y, ybar, lbar, lbar_const, sl, stride, grid);
  return full ? launch16<G16b, true>(ctx, a, y, ybar, lbar, lbar_const, sl, stride, grid)
              : launch16<G16b, false>(ctx, a, y, ybar, lbar, lbar_const, sl, stride, grid);
}
