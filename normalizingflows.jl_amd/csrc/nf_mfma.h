// nf_mfma.h -- register-chained fp32 MFMA primitives for the conditioner MLPs
// of the coupling layers (reference: Flux Dense chain built by fnn,
// src/flows/utils.jl:71-100; used at src/flows/realnvp.jl:50-52,79-80).
//
// gfx950 only.  One wavefront (64 lanes) owns a TILE of 32 samples.
//
// Register layout ("C layout") of a [features x 32 samples] activation block:
// it is exactly the C/D layout of v_mfma_f32_32x32x2_f32 --
//     lane l, register r  <->  sample  j = l & 31,
//                              feature f = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
// (one f32x16 per block of 32 features).
//
// The trick that keeps activations out of LDS: in D = A * B the B operand of
// k-step t is, per lane, "feature k_t of sample l&31" with k_t split over the
// two half-waves.  The contraction order over k is free, so we choose
//     k_t(l) = (t & 3) + 8 * (t >> 2) + 4 * (l >> 5),   t = 0..15
// which makes B-operand t of the NEXT layer identical to accumulator register
// t of THIS layer: bias/activation are applied in place and the registers are
// fed straight back into the matrix pipe.  Only the weight (A) operands are
// fetched, from an LDS image in the reference's own memory order
// (Dense weight is out x in column-major => [in][out], `out` contiguous).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NF_TILE 32  // samples per wave tile

__device__ __forceinline__ int nf_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Flux.leakyrelu, slope 0.01: max(z, 0.01 z) (identical to z > 0 ? z : 0.01 z).  fmaxf() costs a third instruction under
// hipcc: IEEE mode makes it quiet a possible signalling NaN first (`v_max_f32 z, z, z`), 128 of them per coupling in the
// cfg-2 forward; the instruction itself is what is wanted.
__device__ __forceinline__ float nf_vmax(float a, float b) {
#ifdef NF_LRELU_FMAX  // A/B switch: the compiler's own three-instruction form
  return fmaxf(a, b);
#endif
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float nf_lrelu(float z) { return nf_vmax(z, 0.01f * z); }
// a whole accumulator block: the products two per instruction (v_pk_mul_f32), 1.5 VALU ops per element
__device__ __forceinline__ void nf_lrelu16(f32x16 &a) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const f32x2 p = {a[r], a[r + 1]};
    const f32x2 m = p * 0.01f;
    a[r] = nf_vmax(p[0], m[0]);
    a[r + 1] = nf_vmax(p[1], m[1]);
  }
}

// sum of 16 values as a balanced tree: the same 15 additions as a running sum, with error growing with log n instead
// of n (the log-det of a wide coupling is a sum of 128 values of both signs)
__device__ __forceinline__ float nf_tree_sum16(const float (&v)[16]) {
  float a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = v[2 * i] + v[2 * i + 1];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = a[2 * i] + a[2 * i + 1];
  return (b[0] + b[1]) + (b[2] + b[3]);
}

// tanh / exp for the coupling's scale branch (s = tanh(.), exp(+-s); src/flows/realnvp.jl:50,79).
// x / y through the hardware reciprocal (v_rcp_f32, 1 ulp): two instructions.  hipcc lowers
// __fdividef to the full IEEE sequence (div_scale x2, rcp, 4 fma, div_fmas, div_fixup).
__device__ __forceinline__ float nf_fdiv(float x, float y) { return x * __builtin_amdgcn_rcpf(y); }

// tanh.  Round 5: the rational approximation the REFERENCE itself evaluates -- Flux's Dense applies NNlib.fast_act(tanh),
// which for Float32 arrays is NNlib.tanh_fast: x n(x^2) / d(x^2) with two quartics, +-1 beyond x^2 = 66 (NNlib.jl
// src/activations.jl, third-party, restated from the published source; the coefficients below reproduce tanh to 3.2e-7
// absolute over the whole line and to 2e-9 rms below |x| = 0.05, checked against float64 tanh).  Eight FMAs, one v_rcp_f32.
// Rounds 1-4 evaluated (e^2x - 1) / (e^2x + 1) with the hardware exp2: as accurate at its worst point (1.3e-7), but the
// hardware exponential errs to ONE side -- tools/probe/split_bias_probe.hip: mean error -4e-10 ... -3.6e-9 against +-1e-11
// for the rational -- and the log-determinant of a coupling is a SUM of c such values: at d = 256 (2 048 terms per sample)
// that bias was the larger part of cfg 4's ladj error (mean -0.21 of the tolerance for fp32 MFMAs against +0.002 for
// numpy's float32; tools/parity_ab.py).
__device__ __forceinline__ float nf_tanh(float x) {
#ifdef NF_TANH_ACCURATE
  return tanhf(x);
#endif
#ifdef NF_TANH_EXP2  // the rounds-1-4 form (A/B)
  const float xe = fminf(fmaxf(x, -10.f), 10.f);
  const float e2 = __expf(2.f * xe);
  return nf_fdiv(e2 - 1.f, e2 + 1.f);
#endif
  const float xc = __builtin_amdgcn_fmed3f(x, -8.124f, 8.124f);  // beyond x^2 = 66 the reference returns sign(x); the rational is 1 - 2e-7 there
  const float z = xc * xc;
  const float n = fmaf(fmaf(fmaf(fmaf(1.587199e-8f, z, 2.2332108e-5f), z, 0.0035974074f), z, 0.1346604f), z, 1.f);
  const float d = fmaf(fmaf(fmaf(fmaf(8.7767893e-7f, z, 0.0003453992f), z, 0.026262015f), z, 0.4679937f), z, 1.f);
  return xc * nf_fdiv(n, d);
}

// Leaky-ReLU sign masks: one v_alignbit per element shifts the sign bit of v[r] into the mask, so
// element r ends up at bit 15 - r.  leakyrelu keeps the sign, so the mask may be taken from the
// post-activation value.  The empty asm pins the computation where it is written (hipcc otherwise
// sinks it to the mask's first use and keeps the 16 activations alive until then).
__device__ __forceinline__ unsigned nf_sign_mask16(const f32x16 &v) {
  unsigned bits = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) bits = __builtin_amdgcn_alignbit(bits, (unsigned)__float_as_int(v[r]), 31);
  asm volatile("" : "+v"(bits));
  return bits;
}
// slope of element r from the packed mask: sign-extend its bit (v_bfe_i32 -> 0 / -1) and blend the two constants'
// bit patterns with it (v_bfi_b32): two instructions, no VCC round trip (the select form costs v_and + v_cmp + s_nop +
// v_cndmask per element)
__device__ __forceinline__ float nf_mask_slope(unsigned mask, int r) {
#ifdef NF_SLOPE_SELECT  // the round-1 form, kept for A/B measurements
  return ((mask >> (15 - r)) & 1u) ? 0.01f : 1.f;
#else
  const unsigned t = (unsigned)__builtin_amdgcn_sbfe((int)mask, 15 - r, 1);
  return __builtin_bit_cast(float, (t & 0x3C23D70Au) | (~t & 0x3F800000u));
#endif
}
__device__ __forceinline__ float nf_exp(float x) { return __expf(x); }
// a whole cotangent block times its slopes, 2.5 instructions per element and no more: v_bfe_i32 (0 / -1 from the element's mask
// bit), v_bitop3_b32 ((t & (1.0 ^ 0.01)) ^ 1.0 -> the slope's bit pattern), one v_pk_mul_f32 per pair.  In inline asm because hipcc
// 7.2, left to itself, evaluates the block TWICE when the product feeds both a conversion and an MFMA C operand (once with a
// five-instruction slope), seen in the pair kernel's ISA: 9 instructions per element (round 6).
__device__ __forceinline__ void nf_lrelu_grad16(f32x16 &d, unsigned mask) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const unsigned one = 0x3F800000u, flip = 0x3F800000u ^ 0x3C23D70Au;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    int t0, t1;
    unsigned s0, s1;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t0) : "v"(mask), "n"(15 - 2 * p));
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t1) : "v"(mask), "n"(14 - 2 * p));
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c" : "=v"(s0) : "v"(t0), "v"(one), "v"(flip));
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c" : "=v"(s1) : "v"(t1), "v"(one), "v"(flip));
    f32x2_t v = {d[2 * p], d[2 * p + 1]};
    const f32x2_t sl = {__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1)};
#ifdef NF_SLOPE_SCALAR  // two v_mul_f32: a packed f32 instruction does not overlap a matrix instruction in flight (tools/probe/mfma_valu_overlap_probe.hip)
    asm("v_mul_f32 %0, %1, %2" : "=v"(v.x) : "v"(v.x), "v"(sl.x));
    asm("v_mul_f32 %0, %1, %2" : "=v"(v.y) : "v"(v.y), "v"(sl.y));
#else
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(sl));
#endif
    d[2 * p] = v.x;
    d[2 * p + 1] = v.y;
  }
}

// LDS image of one Dense layer: W[i][o] at w[i * S + o], S = 32*OB + 1 (odd stride so
// that both the forward (lanes along o) and the transposed (lanes along i) operand
// reads are bank-conflict free with ds_read_b32), followed by the bias.
struct DenseLds {
  const float *w;
  const float *b;
};

// All MFMA loops below are software-pipelined by hand in groups of four k-steps: the A
// operands of group g+1 are fetched from LDS while the matrix pipe works on group g, and
// __builtin_amdgcn_sched_barrier(0) pins that order.  Without the pins hipcc hoists every
// ds_read of a layer (hundreds) above the first MFMA and spills.

// out[ob] = W * in + b   (no activation).  IB input blocks, OB output blocks.
// Side jobs.  The MFMA loops take an optional functor `sj(slot)` invoked once after every MFMA
// (slot = running MFMA index) with small independent pieces of work (LDS stash of the previous
// activations, leaky-ReLU' scaling of the previous dX result, tile stores).  hipcc is left free to
// place them: measured on MI355X, PINNING a VALU/LDS instruction between two fp32 MFMAs costs more
// than it hides (+6 % on the reverse pass), because back-to-back MFMAs on one accumulator lose their
// fast issue path -- so this only removes separate loops, it does not buy overlap.
// Row padding of every LDS weight image, in floats.  4 keeps rows 16-byte aligned, so the transposed (dX) operand fetch
// is ONE ds_read_b128 per four k-steps (lane <-> row: 8 consecutive rows cover the 32 banks), while the forward fetch
// (lanes along a row, rows 4 apart for the two half-waves: 4 * S = 16 banks apart) stays conflict-free with ds_read_b32.
// 1 is the rounds-1/2 odd stride (four ds_read_b32 per four k-steps in the dX GEMMs), kept for A/B measurements.
#ifndef NF_IMG_PAD
#define NF_IMG_PAD 4
#endif
template <int S>
__device__ __forceinline__ void nf_ld4(const float *__restrict__ p, float &a, float &b, float &c, float &d) {
#ifdef NF_LD4_SCALAR
  if constexpr (false) {
#else
  if constexpr (S % 4 == 0) {
#endif
    const float4 v = *reinterpret_cast<const float4 *>(p);
    a = v.x; b = v.y; c = v.z; d = v.w;
  } else {
    a = p[0]; b = p[1]; c = p[2]; d = p[3];
  }
}

struct NoSideJob {
  __device__ __forceinline__ void operator()(int) const {}
};
template <int IB, int OB, int S = 32 * OB + NF_IMG_PAD, class SJ = NoSideJob>
__device__ __forceinline__ void dense_fwd(const float *__restrict__ w, const float *__restrict__ b,
                                          const f32x16 (&in)[IB], f32x16 (&out)[OB], int l31, int hi,
                                          SJ sj = SJ()) {
  constexpr int NG = IB * 4;  // groups of 4 k-steps
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[ob][r] = b[ob * 32 + nf_row(r, hi)];
  const float *wl = w + (4 * hi) * S + l31;  // lane-dependent part of the address
  float an[OB][4], ac[OB][4];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int e = 0; e < 4; ++e) an[ob][e] = wl[e * S + ob * 32];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
      for (int e = 0; e < 4; ++e) ac[ob][e] = an[ob][e];
    if (g + 1 < NG) {
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int e = 0; e < 4; ++e) an[ob][e] = wl[((g + 1) * 8 + e) * S + ob * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) {
        out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[ob][e], in[g / 4][(g % 4) * 4 + e], out[ob], 0, 0, 0);
        sj((g * 4 + e) * OB + ob);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dense_fwd with a run-time number of ACTIVE k-groups (a group = four k-steps = input features 8g .. 8g+7): inputs beyond
// the layer's real fan-in are zero padding, so their k-steps contribute nothing and are skipped (a wave-uniform branch per
// group).  The spline couplings' first layer has fan-in d - c = 16 of a 32-wide block at cfg 3: 8 MFMAs instead of 16.
template <int IB, int OB, int S = 32 * OB + NF_IMG_PAD>
__device__ __forceinline__ void dense_fwd_dyn(const float *__restrict__ w, const float *__restrict__ b,
                                              const f32x16 (&in)[IB], f32x16 (&out)[OB], int l31, int hi, int nga) {
  constexpr int NG = IB * 4;
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[ob][r] = b[ob * 32 + nf_row(r, hi)];
  const float *wl = w + (4 * hi) * S + l31;
  float an[OB][4], ac[OB][4];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int e = 0; e < 4; ++e) an[ob][e] = wl[e * S + ob * 32];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g < nga) {
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int e = 0; e < 4; ++e) ac[ob][e] = an[ob][e];
      if (g + 1 < NG) {
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
          for (int e = 0; e < 4; ++e) an[ob][e] = wl[((g + 1) * 8 + e) * S + ob * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
          out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[ob][e], in[g / 4][(g % 4) * 4 + e], out[ob], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// din[ib] = W^T * delta : the dX GEMM of the reverse pass, same register chaining.
template <int IB, int OB, int S = 32 * OB + NF_IMG_PAD, bool ACCUM = false, class SJ = NoSideJob>
__device__ __forceinline__ void dense_bwd_x(const float *__restrict__ w, const f32x16 (&delta)[OB],
                                            f32x16 (&din)[IB], int l31, int hi, SJ sj = SJ()) {
  constexpr int NG = OB * 4;
  if (!ACCUM) {
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) din[ib][r] = 0.f;
  }
  const float *wl = w + l31 * S + 4 * hi;
  float an[IB][4], ac[IB][4];
#pragma unroll
  for (int ib = 0; ib < IB; ++ib) nf_ld4<S>(wl + ib * 32 * S, an[ib][0], an[ib][1], an[ib][2], an[ib][3]);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int e = 0; e < 4; ++e) ac[ib][e] = an[ib][e];
    if (g + 1 < NG) {
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) nf_ld4<S>(wl + ib * 32 * S + (g + 1) * 8, an[ib][0], an[ib][1], an[ib][2], an[ib][3]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) {
        din[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[ib][e], delta[g / 4][(g % 4) * 4 + e], din[ib], 0, 0, 0);
        sj((g * 4 + e) * IB + ib);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// din = W^T * delta with the k-steps dealt round-robin to NS independent accumulators (summed at the end): a single
// input block (IB = 1) otherwise makes every MFMA of the GEMM depend on the one before it -- measured on the NSF reverse
// kernel (tools/trace_rqs.py): 103 cycles per MFMA in the one-accumulator chain against 81-85 where chains interleave.
template <int IB, int OB, int S, int NS, int GK = 8, class SJ = NoSideJob>
__device__ __forceinline__ void dense_bwd_x_split(const float *__restrict__ w, const f32x16 (&delta)[OB], f32x16 (&din)[IB],
                                                  int l31, int hi, SJ sj = SJ()) {
  // GK k-steps per pipeline group (a multiple of 4): the A operands of group g + 1 are requested GK MFMAs ahead
  constexpr int NK = OB * 16, NG = NK / GK;
  static_assert(NK % GK == 0 && GK % 4 == 0, "group size");
  f32x16 part[NS][IB];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[s][ib][r] = 0.f;
  const float *wl = w + l31 * S + 4 * hi;
  // k-step t (0 .. NK-1) contracts output o = 8 (t / 4) + (t % 4) + 4 hi: row offset 8 (t / 4) + (t % 4) from wl
  float an[IB][GK], ac[IB][GK];
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int e = 0; e < GK; e += 4) nf_ld4<S>(wl + ib * 32 * S + 8 * (e / 4), an[ib][e], an[ib][e + 1], an[ib][e + 2], an[ib][e + 3]);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int e = 0; e < GK; ++e) ac[ib][e] = an[ib][e];
    if (g + 1 < NG) {
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int e = 0; e < GK; e += 4) {
          const int t = (g + 1) * GK + e;
          nf_ld4<S>(wl + ib * 32 * S + 8 * (t / 4), an[ib][e], an[ib][e + 1], an[ib][e + 2], an[ib][e + 3]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < GK; ++e)
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) {
        const int t = g * GK + e;
        part[e % NS][ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[ib][e], delta[t / 16][t % 16], part[e % NS][ib], 0, 0, 0);
        sj(t * IB + ib);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float t = part[0][ib][r];
#pragma unroll
      for (int s = 1; s < NS; ++s) t += part[s][ib][r];
      din[ib][r] = t;
    }
}

// ---- weight-gradient GEMM: contraction over SAMPLES -------------------------------
// dW^T[i][o] += sum_j a[i][j] * delta[o][j].  Both operands must have lane <-> feature,
// i.e. the transpose of the C layout, so they take one round trip through a per-wave
// LDS scratch tile [feature][sample] with row stride 33 (conflict-free both ways).
#define NF_TS 33

template <int NB>
__device__ __forceinline__ void tile_to_scratch(float *__restrict__ sc, const f32x16 (&v)[NB], int l31, int hi) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[(b * 32 + nf_row(r, hi)) * NF_TS + l31] = v[b][r];
}

// one element (flat index e = block * 16 + reg) of tile_to_scratch, for use in side jobs
template <int NB>
__device__ __forceinline__ void scratch_put(float *__restrict__ sc, const f32x16 (&v)[NB], int e, int l31, int hi) {
  if (e < NB * 16) sc[((e >> 4) * 32 + nf_row(e & 15, hi)) * NF_TS + l31] = v[e >> 4][e & 15];
}

// acc[ib][ob] (C layout: col = o, rows = i) += A(a) * B(delta); bsum[ob] += sum over this
// half-wave's samples of delta (lane <-> feature o).  sa/sd: scratch tiles of a and delta.
template <int IB, int OB, class SJ = NoSideJob>
__device__ __forceinline__ void dw_accumulate(const float *__restrict__ sa, const float *__restrict__ sd,
                                              f32x16 (&acc)[IB][OB], float (&bsum)[OB], int l31, int hi,
                                              SJ sj = SJ()) {
  constexpr int TG = 2;        // k-steps (sample pairs) per pipeline group
  constexpr int NG = 16 / TG;
  const float *pa = sa + l31 * NF_TS + hi;
  const float *pd = sd + l31 * NF_TS + hi;
  float an[TG][IB], dn[TG][OB], ac[TG][IB], dc[TG][OB];
#pragma unroll
  for (int u = 0; u < TG; ++u) {
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) an[u][ib] = pa[ib * 32 * NF_TS + 2 * u];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) dn[u][ob] = pd[ob * 32 * NF_TS + 2 * u];
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int u = 0; u < TG; ++u) {
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) ac[u][ib] = an[u][ib];
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) dc[u][ob] = dn[u][ob];
    }
    if (g + 1 < NG) {
#pragma unroll
      for (int u = 0; u < TG; ++u) {
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) an[u][ib] = pa[ib * 32 * NF_TS + 2 * ((g + 1) * TG + u)];
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) dn[u][ob] = pd[ob * 32 * NF_TS + 2 * ((g + 1) * TG + u)];
      }
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
          acc[ib][ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u][ib], dc[u][ob], acc[ib][ob], 0, 0, 0);
          sj(((g * TG + u) * IB + ib) * OB + ob);
          }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// same, accumulating into columns [o0, o0 + OB) of a wider accumulator array (o0 is a constant
// after unrolling at every call site)
template <int IB, int OB, int OBTOT>
__device__ __forceinline__ void dw_accumulate_at(const float *__restrict__ sa, const float *__restrict__ sd,
                                                 f32x16 (&acc)[IB][OBTOT], float (&bsum)[OBTOT], int o0, int l31,
                                                 int hi) {
  const float *pa = sa + l31 * NF_TS + hi;
  const float *pd = sd + l31 * NF_TS + hi;
  float an[2][IB], dn[2][OB], ac[2][IB], dc[2][OB];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) an[u][ib] = pa[ib * 32 * NF_TS + 2 * u];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) dn[u][ob] = pd[ob * 32 * NF_TS + 2 * u];
  }
#pragma unroll
  for (int g = 0; g < 8; ++g) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) ac[u][ib] = an[u][ib];
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) dc[u][ob] = dn[u][ob];
    }
    if (g + 1 < 8) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) an[u][ib] = pa[ib * 32 * NF_TS + 2 * ((g + 1) * 2 + u)];
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) dn[u][ob] = pd[ob * 32 * NF_TS + 2 * ((g + 1) * 2 + u)];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) bsum[o0 + ob] += dc[u][ob];
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
          acc[ib][o0 + ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u][ib], dc[u][ob], acc[ib][o0 + ob], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Orders a wave's own LDS writes before its own LDS reads (cross-lane exchange inside
// one wavefront: no s_barrier needed, the LDS queue is in order per wave).
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- LDS image of a 2-hidden-layer conditioner net ----------------------------------
// Geometry in blocks of 32: MB (conditioner inputs), H1B, H2B (hidden), CB (outputs).
// PAD = row padding in floats (odd for the 32-sample kernels; 4 for the 16-sample kernel, which
// reads four consecutive rows' worth of one column group with ds_read_b128).
template <int MB_, int H1B_, int H2B_, int CB_, int PAD_ = NF_IMG_PAD>
struct NetGeo {
  static constexpr int MB = MB_, H1B = H1B_, H2B = H2B_, CB = CB_, PAD = PAD_;
  static constexpr int S1 = 32 * H1B + PAD, S2 = 32 * H2B + PAD, S3 = 32 * CB + PAD;
  static constexpr int W1 = 0;
  static constexpr int B1 = W1 + 32 * MB * S1;
  static constexpr int W2 = B1 + 32 * H1B;
  static constexpr int B2 = W2 + 32 * H1B * S2;
  static constexpr int W3 = B2 + 32 * H2B;
  static constexpr int B3 = W3 + 32 * H2B * S3;
  static constexpr int SIZE = ((B3 + 32 * CB + 3) / 4) * 4;  // floats, 16-byte multiple
  static constexpr int NACC = MB * H1B + H1B * H2B + H2B * CB;  // f32x16 dW accumulators
};

// Actual (unpadded) sizes and theta offsets of one net.
struct NetDims {
  int m, h1, h2, c;  // fan-in, hidden, hidden, fan-out
  long w1, b1, w2, b2, w3, b3;  // offsets into theta (Optimisers.destructure order)
};

__host__ __device__ __forceinline__ NetDims make_net_dims(long off, int m, int h1, int h2, int c) {
  NetDims n;
  n.m = m; n.h1 = h1; n.h2 = h2; n.c = c;
  n.w1 = off; n.b1 = n.w1 + (long)m * h1;
  n.w2 = n.b1 + h1; n.b2 = n.w2 + (long)h1 * h2;
  n.w3 = n.b2 + h2; n.b3 = n.w3 + (long)h2 * c;
  return n;
}
__host__ __device__ __forceinline__ long net_param_count(int m, int h1, int h2, int c) {
  return (long)m * h1 + h1 + (long)h1 * h2 + h2 + (long)h2 * c + c;
}

// Copy one Dense layer theta[in][out] -> padded LDS image, zero fill.
template <int S>
__device__ __forceinline__ void stage_dense(float *__restrict__ img, int rows_pad, const float *__restrict__ theta,
                                            int nin, int nout, int tid, int nthreads) {
  for (int idx = tid; idx < rows_pad * S; idx += nthreads) {
    const int i = idx / S, o = idx - i * S;
    img[idx] = (i < nin && o < nout) ? theta[(long)i * nout + o] : 0.f;
  }
}
__device__ __forceinline__ void stage_bias(float *__restrict__ img, int n_pad, const float *__restrict__ theta,
                                           int n, int tid, int nthreads) {
  for (int idx = tid; idx < n_pad; idx += nthreads) img[idx] = idx < n ? theta[idx] : 0.f;
}

template <class G>
__device__ __forceinline__ void stage_net(float *__restrict__ img, const float *__restrict__ theta, const NetDims &nd,
                                          int tid, int nthreads) {
  stage_dense<G::S1>(img + G::W1, 32 * G::MB, theta + nd.w1, nd.m, nd.h1, tid, nthreads);
  stage_bias(img + G::B1, 32 * G::H1B, theta + nd.b1, nd.h1, tid, nthreads);
  stage_dense<G::S2>(img + G::W2, 32 * G::H1B, theta + nd.w2, nd.h1, nd.h2, tid, nthreads);
  stage_bias(img + G::B2, 32 * G::H2B, theta + nd.b2, nd.h2, tid, nthreads);
  stage_dense<G::S3>(img + G::W3, 32 * G::H2B, theta + nd.w3, nd.h2, nd.c, tid, nthreads);
  stage_bias(img + G::B3, 32 * G::CB, theta + nd.b3, nd.c, tid, nthreads);
}

// Stage a PRE-PACKED image (k_pack_net_images: same layout, already padded and zero filled)
// from global memory: a straight 16-byte-vector copy with every load issued before the first
// LDS store, so the HBM/L2 latency is paid once per staging instead of once per element.
template <int SIZE, int NTHREADS>
__device__ __forceinline__ void stage_packed(float *__restrict__ img, const float *__restrict__ packed, int tid) {
  constexpr int NV4 = SIZE / 4;
  constexpr int PER = (NV4 + NTHREADS - 1) / NTHREADS;
  const float4 *src = reinterpret_cast<const float4 *>(packed);
  float4 *dst = reinterpret_cast<float4 *>(img);
  float4 tmp[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int idx = tid + k * NTHREADS;
    tmp[k] = idx < NV4 ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int idx = tid + k * NTHREADS;
    if (idx < NV4) dst[idx] = tmp[k];
  }
}

// theta index of element `e` of a net's padded image, or -1 for padding
template <class G>
__device__ __forceinline__ long image_theta_index(const NetDims &nd, int e) {
  // selects on VALUES: written as assignments inside an if-chain hipcc keeps the six offsets in a 48-byte scratch array and
  // indexes it by layer (seen as private_segment_fixed_size 48 in every kernel that calls this; tests/test_kernel_resources_cpu.py)
  const bool l1 = e < G::W2, l2 = e < G::W3;
  const int base = l1 ? G::W1 : l2 ? G::W2 : G::W3;
  const int rows = l1 ? 32 * G::MB : l2 ? 32 * G::H1B : 32 * G::H2B;
  const int S = l1 ? G::S1 : l2 ? G::S2 : G::S3;
  const int dm = nd.m, dh1 = nd.h1, dh2 = nd.h2, dc = nd.c;
  const int nin = l1 ? dm : l2 ? dh1 : dh2;
  const int nout = l1 ? dh1 : l2 ? dh2 : dc;
  const long w1 = nd.w1, w2 = nd.w2, w3 = nd.w3, b1 = nd.b1, b2 = nd.b2, b3 = nd.b3;
  const long w = l1 ? w1 : l2 ? w2 : w3;
  const long b = l1 ? b1 : l2 ? b2 : b3;
  const int r = e - base;
  if (r < rows * S) {
    const int i = r / S, o = r - i * S;
    return (i < nin && o < nout) ? w + (long)i * nout + o : -1;
  }
  const int o = r - rows * S;
  return o < nout ? b + o : -1;
}

// Forward through the 2-hidden-layer net.  a1/a2 are the post-leakyrelu activations
// (kept for the reverse pass), out the pre-output-activation result.
template <class G>
__device__ __forceinline__ void net_forward(const float *__restrict__ img, const f32x16 (&x)[G::MB],
                                            f32x16 (&a1)[G::H1B], f32x16 (&a2)[G::H2B], f32x16 (&out)[G::CB],
                                            int l31, int hi) {
  dense_fwd<G::MB, G::H1B>(img + G::W1, img + G::B1, x, a1, l31, hi);
#pragma unroll
  for (int b = 0; b < G::H1B; ++b)
    nf_lrelu16(a1[b]);
  dense_fwd<G::H1B, G::H2B>(img + G::W2, img + G::B2, a1, a2, l31, hi);
#pragma unroll
  for (int b = 0; b < G::H2B; ++b)
    nf_lrelu16(a2[b]);
  dense_fwd<G::H2B, G::CB>(img + G::W3, img + G::B3, a2, out, l31, hi);
}

// ---- fp32-grade GEMMs on the bf16 matrix cores ("B6": six bf16 products per fp32 product) -------------------------
// v_mfma_f32_32x32x16_bf16 does 16 384 MACs in 32 clocks, v_mfma_f32_32x32x2_f32 2 048 in 64 (tools/probe/bf16x6_probe.hip,
// measured): sixteen times the rate.  An fp32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significant bits:
// x = xh + xm + xl by two roundings and two exact subtractions, nf_split2 below), so x w = sum of nine bf16 x bf16 products,
// each exact in the instruction's fp32 accumulator; the three smallest (xm wl, xl wm, xl wl: <= 2^-25 of the product
// together, either sign) are dropped, the other six are issued smallest first.  Six instructions of 32 clocks replace eight of 64 per 16 k-steps: 2.67 x the matrix-pipe
// rate of the fp32 path, paid for with ~3.5 VALU instructions per ACTIVATION for the split (the weights are split once,
// when theta is packed).  Round 4: the chain kernels without a stash, all six GEMMs of the pair kernel (PB6 + DW6) and the
// weight-streaming kernels; round 5: the NSF kernels' output layer.
//
// Operand layout of the instruction (A: 32 x 16, B: 16 x 32): lane <-> row / column l & 31, the lane's eight bf16 are
// k = 8 (lane >> 5) + j.  The register chaining of nf_mfma.h carries over: a lane of half `hi` holds, of a 32-feature
// block in the C layout, the features (r & 3) + 8 (r >> 2) + 4 hi, r = 0 .. 15; registers 8 g .. 8 g + 7 are therefore the
// eight k-values of that lane for k-group g (16 features) of the block, with the contraction order
//     feature(g, hi, j) = 16 g + (j & 3) + 8 (j >> 2) + 4 hi,
// and the weight image stores each row's eight weights in exactly that order, so an A operand is one ds_read_b128.
typedef unsigned nf_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 nf_bf16x8 __attribute__((ext_vector_type(8)));

// The split (round 5: ROUND-TO-NEAREST parts).  h = bf16(x), m = bf16(x - h), l = (x - h) - m, every subtraction exact and
// l itself a bf16 value (x - h is a multiple of ulp_f32(x) below 2^-8 |x|: 16 significant bits, m takes eight, the rest
// fits eight) -- so x = h + m + l exactly, as with the truncating split of round 4, but the parts are SIGNED residuals:
// |m| <= 2^-9 |x|, |l| <= 2^-17 |x|, and the three dropped products (m l', l m', l l') are <= 2^-25 of a product with
// either sign.  With truncation (x & 0xFFFF0000) m and l always carried the sign of x, every dropped term the sign of the
// product, and the deficit (mean 4e-8 of a product, measured) added coherently over k -- VERDICT r4 weak 1: golden
// realnvp_d64_h64 ys 5.3 -> 8.9 x the tolerance, cfg 5 ladj_inv 2.9 -> 9.6 x.  gfx950 rounds two values per instruction
// (v_cvt_pk_bf16_f32, what hipcc emits for the conversion below) and hands back the PACKED dword the MFMA operand wants, so
// h and m need no v_perm any more: cvt, shl, and, pk_add, cvt, shl, and, pk_add, perm = nine instructions per pair of
// values, what the truncating form cost (and, and, pk_add, and, and, pk_add, 3 perm).
typedef float nf_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned nf_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 nf_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned nf_cvt_pk_bf16(nf_f32x2 x) {  // (bf16(x.y) << 16) | bf16(x.x), round to nearest even
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, nf_bf16x2));
}
__device__ __forceinline__ nf_f32x2 nf_widen_pk_bf16(unsigned p) {
  const nf_u32x2 w = {p << 16, p & 0xFFFF0000u};
  return __builtin_bit_cast(nf_f32x2, w);
}
// two values -> one dword per component: (part of x1) : (part of x0)
__device__ __forceinline__ void nf_split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
// No contraction in here: with -ffp-contract=fast hipcc folds a multiply that PRODUCED x into the first subtraction (an fma on
// the unrounded product), so h + m + l is then not the x every other user of the register sees -- measured as 4.5 x the
// round-trip error of the 1 M-sample inverse / forward pair of cfg 5.
#pragma clang fp contract(off)
  const nf_f32x2 x = {x0, x1};
#ifdef NF_SPLIT_TRUNC  // the round-4 split (A/B of the arithmetic; tools/split_ab.sh)
  const nf_u32x2 xb = __builtin_bit_cast(nf_u32x2, x);
  const nf_f32x2 r = x - __builtin_bit_cast(nf_f32x2, xb & 0xFFFF0000u);
  const nf_u32x2 rb = __builtin_bit_cast(nf_u32x2, r);
  const nf_f32x2 lo = r - __builtin_bit_cast(nf_f32x2, rb & 0xFFFF0000u);
  h = __builtin_amdgcn_perm(xb.y, xb.x, 0x07060302u);
  m = __builtin_amdgcn_perm(rb.y, rb.x, 0x07060302u);
#elif defined(NF_SPLIT_SCALAR)  // A/B: the two subtractions as scalar v_sub_f32 pairs instead of v_pk_add_f32
  h = nf_cvt_pk_bf16(x);
  const nf_f32x2 hw = nf_widen_pk_bf16(h);
  float r0 = x0 - hw.x, r1 = x1 - hw.y;
  asm("" : "+v"(r0));
  asm("" : "+v"(r1));
  const nf_f32x2 r = {r0, r1};
  m = nf_cvt_pk_bf16(r);
  const nf_f32x2 mw = nf_widen_pk_bf16(m);
  float l0 = r0 - mw.x, l1 = r1 - mw.y;
  asm("" : "+v"(l0));
  asm("" : "+v"(l1));
  const nf_f32x2 lo = {l0, l1};
#else
  h = nf_cvt_pk_bf16(x);
  const nf_f32x2 r = x - nf_widen_pk_bf16(h);   // exact
  m = nf_cvt_pk_bf16(r);
  const nf_f32x2 lo = r - nf_widen_pk_bf16(m);  // exact, at most 8 significant bits: its upper half IS the value
#endif
  const nf_u32x2 lb = __builtin_bit_cast(nf_u32x2, lo);
  l = __builtin_amdgcn_perm(lb.y, lb.x, 0x07060302u);
}
// one value (the weight images are split element by element when theta is packed: nf_pack.h)
__device__ __forceinline__ void nf_split1(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
  unsigned ph, pm, pl;
  nf_split2(x, 0.f, ph, pm, pl);
  h = (unsigned short)ph; m = (unsigned short)pm; l = (unsigned short)pl;
}
__device__ __forceinline__ void nf_split8(const float (&v)[8], nf_u32x4 &h, nf_u32x4 &m, nf_u32x4 &l) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned ph, pm, pl;
    nf_split2(v[2 * p], v[2 * p + 1], ph, pm, pl);
    h[p] = ph; m[p] = pm; l[p] = pl;
#ifdef NF_SPLIT_PINNED
    __builtin_amdgcn_sched_barrier(0);  // (a pair at a time: 6 temporaries instead of 24)
#endif
  }
}
__device__ __forceinline__ f32x16 nf_mfma_bf16(nf_u32x4 a, nf_u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nf_bf16x8, a), __builtin_bit_cast(nf_bf16x8, b), c, 0, 0, 0);
}

// ---- the split with its two exact subtractions ON THE MATRIX PIPE (round 6) -----------------------------------------------
// nf_split2 spends 4.5 vector instructions per value, two thirds of them on r = x - h and l = r - m (widen h: shl + and per
// pair, then v_pk_add_f32; the same for m).  Both differences are exact, and an MFMA computes exact differences for free:
//     D = C + A B   with C = the fp32 block x (16 registers per lane, C layout), B = its packed bf16 part h (the very
//                   registers the split produces), A = MINUS the selection matrix that maps B's k-order onto C's rows
// gives D = x - h for the whole 32 x 32 block: every output is ONE product (-1 x h, exact) plus C, and x - h is a multiple
// of ulp(x) below 2^-8 |x|, so the instruction's adder has nothing to drop (tools/probe/split_mfma_probe.hip compares all
// three parts bit for bit with nf_split2 on random, tie, subnormal-adjacent and huge inputs).  A block of 16 values then
// costs 8 + 8 + 8 conversions (1.5 instructions per value) and four v_mfma_f32_32x32x16_bf16 (two k-groups x two levels).
// MEASURED (round 6): bit-identical, halves the vector instructions of the cfg-2 reverse kernel -- and buys
// nothing there, in the cfg-5 chain or in k_rqs_bwd_coop6 (the split's MFMAs join the dependent chain of the wave that was
// already the slower of its SIMD's two; these kernels are not bound by vector issue).  The primitive and its probe stay for
// the next kernel that IS; the call sites that were tried are tools/experiments/pair_chain_split_on_matrix_pipe.patch.
// The selection operand: hardware k = 8 hi + j of k-group g is C row 16 g + (j & 3) + 8 (j >> 2) + 4 hi (nf_row(8 g + j, hi)),
// so lane (row i = l31, half hi) holds -1 at element j iff i - 16 g - 4 hi = (j & 3) + 8 (j >> 2), zeros elsewhere.
struct SplitSel {
  nf_u32x4 a[2];  // A operands of the two k-groups
};
__device__ __forceinline__ SplitSel nf_split_sel(int l31, int hi) {
  SplitSel s;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int q = l31 - 16 * g - 4 * hi;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int j0 = 2 * p, j1 = 2 * p + 1;
      const unsigned lo = q == (j0 & 3) + 8 * (j0 >> 2) ? 0xBF80u : 0u, hv = q == (j1 & 3) + 8 * (j1 >> 2) ? 0xBF800000u : 0u;
      s.a[g][p] = lo | hv;
    }
  }
  // pinned: hipcc otherwise rematerialises the eight compares + selects in front of every use
  asm volatile("" : "+v"(s.a[0]), "+v"(s.a[1]));
  return s;
}
// one C-layout block (16 values per lane) -> its triples per k-group g = register >> 3: h[g], m[g], l[g]
__device__ __forceinline__ void nf_split16_mfma(const SplitSel &sel, const f32x16 &x, nf_u32x4 (&h)[2], nf_u32x4 (&m)[2],
                                                nf_u32x4 (&l)[2]) {
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int p = 0; p < 4; ++p) h[g][p] = nf_cvt_pk_bf16(nf_f32x2{x[8 * g + 2 * p], x[8 * g + 2 * p + 1]});
  f32x16 r = nf_mfma_bf16(sel.a[0], h[0], x);
  r = nf_mfma_bf16(sel.a[1], h[1], r);  // r = x - h
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int p = 0; p < 4; ++p) m[g][p] = nf_cvt_pk_bf16(nf_f32x2{r[8 * g + 2 * p], r[8 * g + 2 * p + 1]});
  f32x16 lo = nf_mfma_bf16(sel.a[0], m[0], r);
  lo = nf_mfma_bf16(sel.a[1], m[1], lo);  // lo = r - m: at most 8 significant bits, its upper half IS the value
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float e0 = lo[8 * g + 2 * p], e1 = lo[8 * g + 2 * p + 1];  // (scalars first: bit_cast of a vector element, see nf_coupling.hip)
      l[g][p] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, e1), __builtin_bit_cast(unsigned, e0), 0x07060302u);
    }
}

// B6 image of one net (geometry G = NetGeo<..>), in 16-byte units: per layer [k-group][component h|m|l][half][row][8 bf16],
// rows = padded fan-out, k-groups = 2 x input blocks; then the three bias vectors in fp32.
template <class G>
struct B6Geo {
  static constexpr int R1 = 32 * G::H1B, R2 = 32 * G::H2B, R3 = 32 * G::CB;   // rows (outputs) per layer
  static constexpr int L1 = 0;
  static constexpr int L2 = L1 + 2 * G::MB * 6 * R1;
  static constexpr int L3 = L2 + 2 * G::H1B * 6 * R2;
  static constexpr int BIAS = L3 + 2 * G::H2B * 6 * R3;                       // 16-byte units up to here
  static constexpr int B1 = 0, B2 = R1, B3 = R1 + R2;                         // float offsets inside the bias block
  static constexpr int U4 = BIAS + (R1 + R2 + R3 + 3) / 4;                    // size in 16-byte units
  static constexpr int BYTES = U4 * 16;
};

// out[ob] = W in + b through the six-term bf16 product.  `w`: the layer's part of a B6 image (LDS), ROWS = 32 * OB.
// Pipeline unit = one (k-group, output block): its three A operands (12 registers) are requested one unit ahead, behind
// the six MFMAs (192 clocks) of the unit before.
// LEAN: no operand double buffer (the unit's three A operands are requested when the unit starts): 12 registers less, the
// LDS latency is left to the other wave of the SIMD -- for kernels at their register wall (the stashing forward).
template <int IB, int OB, class SJ = NoSideJob, bool LEAN = false>
__device__ __forceinline__ void dense_fwd_b6(const nf_u32x4 *__restrict__ w, const float *__restrict__ b, const f32x16 (&in)[IB],
                                             f32x16 (&out)[OB], int l31, int hi, SJ sj = SJ()) {
  constexpr int ROWS = 32 * OB, NKG = 2 * IB, NU = NKG * OB;
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[ob][r] = b[ob * 32 + nf_row(r, hi)];
  const nf_u32x4 *wl = w + hi * ROWS + l31;  // lane part of the address; (k-group, component, block) are immediates
  nf_u32x4 an[3], ac[3];
  if (!LEAN) {
#pragma unroll
    for (int c = 0; c < 3; ++c) an[c] = wl[c * 2 * ROWS];
  }
  nf_u32x4 xh, xm, xl;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int kg = u / OB, ob = u % OB;
    if (LEAN) {
#pragma unroll
      for (int c = 0; c < 3; ++c) ac[c] = wl[(kg * 3 + c) * 2 * ROWS + ob * 32];
    }
    if (ob == 0) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = in[kg >> 1][8 * (kg & 1) + j];
      nf_split8(v, xh, xm, xl);
    }
    if (!LEAN) {
#pragma unroll
      for (int c = 0; c < 3; ++c) ac[c] = an[c];
      if (u + 1 < NU) {
        const int kg1 = (u + 1) / OB, ob1 = (u + 1) % OB;
#pragma unroll
        for (int c = 0; c < 3; ++c) an[c] = wl[(kg1 * 3 + c) * 2 * ROWS + ob1 * 32];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // smallest terms first: wl xh, wh xl, wm xm, wm xh, wh xm, wh xh
    out[ob] = nf_mfma_bf16(ac[2], xh, out[ob]); sj(12 * u + 0); sj(12 * u + 1);
    out[ob] = nf_mfma_bf16(ac[0], xl, out[ob]); sj(12 * u + 2); sj(12 * u + 3);
    out[ob] = nf_mfma_bf16(ac[1], xm, out[ob]); sj(12 * u + 4); sj(12 * u + 5);
    out[ob] = nf_mfma_bf16(ac[1], xh, out[ob]); sj(12 * u + 6); sj(12 * u + 7);
    out[ob] = nf_mfma_bf16(ac[0], xm, out[ob]); sj(12 * u + 8); sj(12 * u + 9);
    out[ob] = nf_mfma_bf16(ac[0], xh, out[ob]); sj(12 * u + 10); sj(12 * u + 11);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same product as a software pipeline (round 5; tools/trace_chain_b6.py).  In dense_fwd_b6 a k-group's split (45 vector
// instructions) runs, then the six MFMAs of a unit wait for one another on one accumulator: a wave alone needs 6.6 k clocks for a
// 32-64-64-32 net whose 96 MFMAs occupy the matrix pipe for 3.1 k, and the second wave of the SIMD only fills part of the gaps
// (BASELINE cfg 5: 11.5 k per net and SIMD, 53 % of the pipe).  Here the k-group kg + 1 is split -- and its weights are
// requested from LDS -- in the issue shadows of k-group kg's MFMAs (sched_group_barrier: one MFMA, a few VALU), and the OB
// accumulators of a k-group alternate term by term, so that no MFMA waits for the one before it (OB = 2).  The order of the
// terms of every accumulator is the one of dense_fwd_b6: bit-identical results.
// LRIN: `in` holds PRE-activation values and the leaky ReLU is applied where a k-group's eight values are read for their split -- in the
// issue shadows of the previous k-group's matrix instructions instead of as a pass of its own between two layers (chains without a stash)
template <int IB, int OB, bool LRIN = false>
__device__ __forceinline__ void dense_fwd_b6p(const nf_u32x4 *__restrict__ w, const float *__restrict__ b, const f32x16 (&in)[IB],
                                              f32x16 (&out)[OB], int l31, int hi) {
  constexpr int ROWS = 32 * OB, NKG = 2 * IB;
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[ob][r] = b[ob * 32 + nf_row(r, hi)];
  const nf_u32x4 *wl = w + hi * ROWS + l31;  // lane part of the address; (k-group, component, block) are immediates
  nf_u32x4 an[OB][3], xn[3];
  {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = LRIN ? nf_vmax(in[0][j], 0.01f * in[0][j]) : in[0][j];
    nf_split8(v, xn[0], xn[1], xn[2]);
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
      for (int c = 0; c < 3; ++c) an[ob][c] = wl[c * 2 * ROWS + ob * 32];
  }
#pragma unroll
  for (int kg = 0; kg < NKG; ++kg) {
    nf_u32x4 ac[OB][3], xc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      xc[c] = xn[c];
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) ac[ob][c] = an[ob][c];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kg + 1 < NKG) {
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int c = 0; c < 3; ++c) an[ob][c] = wl[((kg + 1) * 3 + c) * 2 * ROWS + ob * 32];
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = in[(kg + 1) >> 1][8 * ((kg + 1) & 1) + j];
        v[j] = LRIN ? nf_vmax(t, 0.01f * t) : t;
      }
      nf_split8(v, xn[0], xn[1], xn[2]);
    }
    // smallest terms first: wl xh, wh xl, wm xm, wm xh, wh xm, wh xh (components: 0 = h, 1 = m, 2 = l)
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) {
        const nf_u32x4 &av = term == 0 ? ac[ob][2] : (term == 2 || term == 3) ? ac[ob][1] : ac[ob][0];
        const nf_u32x4 &xv = term == 1 ? xc[2] : (term == 2 || term == 4) ? xc[1] : xc[0];
        out[ob] = nf_mfma_bf16(av, xv, out[ob]);
      }
    if (kg + 1 < NKG) {
      __builtin_amdgcn_sched_group_barrier(0x100, 3 * OB, 0);  // the next k-group's weights: requested first
#pragma unroll
      for (int i = 0; i < 6 * OB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, OB == 1 ? 8 : 4, 0);  // its shadow: a slice of the next split
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dense_fwd_b6p for TWO tiles at once (round 6; k_affine_chain_dual): the unit's weight operands are read from LDS once and serve both
// tiles, 2 OB accumulators alternate term by term, and the two next-k-group splits (72 vector instructions) ride behind 12 OB matrix
// instructions instead of 6 OB.  Per accumulator the order of the terms is dense_fwd_b6's: bit-identical results.
template <int IB, int OB>
__device__ __forceinline__ void dense_fwd_b6p2(const nf_u32x4 *__restrict__ w, const float *__restrict__ b, const f32x16 (&in0)[IB],
                                               const f32x16 (&in1)[IB], f32x16 (&out0)[OB], f32x16 (&out1)[OB], int l31, int hi) {
  constexpr int ROWS = 32 * OB, NKG = 2 * IB;
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) out0[ob][r] = out1[ob][r] = b[ob * 32 + nf_row(r, hi)];
  const nf_u32x4 *wl = w + hi * ROWS + l31;
  nf_u32x4 an[OB][3], xn0[3], xn1[3];
  {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = in0[0][j];
    nf_split8(v, xn0[0], xn0[1], xn0[2]);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = in1[0][j];
    nf_split8(v, xn1[0], xn1[1], xn1[2]);
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
      for (int c = 0; c < 3; ++c) an[ob][c] = wl[c * 2 * ROWS + ob * 32];
  }
#pragma unroll
  for (int kg = 0; kg < NKG; ++kg) {
    nf_u32x4 ac[OB][3], xc0[3], xc1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      xc0[c] = xn0[c];
      xc1[c] = xn1[c];
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) ac[ob][c] = an[ob][c];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kg + 1 < NKG) {
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int c = 0; c < 3; ++c) an[ob][c] = wl[((kg + 1) * 3 + c) * 2 * ROWS + ob * 32];
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = in0[(kg + 1) >> 1][8 * ((kg + 1) & 1) + j];
      nf_split8(v, xn0[0], xn0[1], xn0[2]);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = in1[(kg + 1) >> 1][8 * ((kg + 1) & 1) + j];
      nf_split8(v, xn1[0], xn1[1], xn1[2]);
    }
    // smallest terms first: wl xh, wh xl, wm xm, wm xh, wh xm, wh xh (components: 0 = h, 1 = m, 2 = l)
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) {
        const nf_u32x4 &av = term == 0 ? ac[ob][2] : (term == 2 || term == 3) ? ac[ob][1] : ac[ob][0];
        const int xi = term == 1 ? 2 : (term == 2 || term == 4) ? 1 : 0;
        out0[ob] = nf_mfma_bf16(av, xc0[xi], out0[ob]);
        out1[ob] = nf_mfma_bf16(av, xc1[xi], out1[ob]);
      }
    if (kg + 1 < NKG) {
      __builtin_amdgcn_sched_group_barrier(0x100, 3 * OB, 0);  // the next k-group's weights: requested first
#pragma unroll
      for (int i = 0; i < 12 * OB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, OB == 1 ? 8 : 4, 0);  // its shadow: a slice of the two next splits
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The transposed counterpart for the dX GEMMs of the reverse pass (din = W^T delta): rows = the layer's INPUT features, the
// k-groups run over its OUTPUT features, in the order the cotangent's C-layout registers hold them (same mapping as above).
// Per layer [k-group][component][half][row][8 bf16]; no biases.  T3 serves dX3 (rows: a2 features), T2 dX2, T1 dX1.
template <class G>
struct B6TGeo {
  static constexpr int R3 = 32 * G::H2B, R2 = 32 * G::H1B, R1 = 32 * G::MB;  // rows per layer
  static constexpr int T3 = 0;
  static constexpr int T2 = T3 + 2 * G::CB * 6 * R3;
  static constexpr int T1 = T2 + 2 * G::H2B * 6 * R2;
  static constexpr int U4 = T1 + 2 * G::H1B * 6 * R1;  // 16-byte units
  static constexpr int BYTES = U4 * 16;
};

// din[ib] = W^T delta through the six-term bf16 product; `w`: the layer's part of a B6T image (LDS), ROWS = 32 * IB.
template <int IB, int OB, class SJ = NoSideJob>
__device__ __forceinline__ void dense_bwd_x_b6(const nf_u32x4 *__restrict__ w, const f32x16 (&delta)[OB], f32x16 (&din)[IB], int l31,
                                               int hi, SJ sj = SJ()) {
  constexpr int ROWS = 32 * IB, NKG = 2 * OB, NU = NKG * IB;
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int r = 0; r < 16; ++r) din[ib][r] = 0.f;
  const nf_u32x4 *wl = w + hi * ROWS + l31;
  nf_u32x4 an[3], ac[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) an[c] = wl[c * 2 * ROWS];
  nf_u32x4 xh, xm, xl;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int kg = u / IB, ib = u % IB;
    if (ib == 0) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = delta[kg >> 1][8 * (kg & 1) + j];
      nf_split8(v, xh, xm, xl);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) ac[c] = an[c];
    if (u + 1 < NU) {
      const int kg1 = (u + 1) / IB, ib1 = (u + 1) % IB;
#pragma unroll
      for (int c = 0; c < 3; ++c) an[c] = wl[(kg1 * 3 + c) * 2 * ROWS + ib1 * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
    din[ib] = nf_mfma_bf16(ac[2], xh, din[ib]); sj(12 * u + 0); sj(12 * u + 1);
    din[ib] = nf_mfma_bf16(ac[0], xl, din[ib]); sj(12 * u + 2); sj(12 * u + 3);
    din[ib] = nf_mfma_bf16(ac[1], xm, din[ib]); sj(12 * u + 4); sj(12 * u + 5);
    din[ib] = nf_mfma_bf16(ac[1], xh, din[ib]); sj(12 * u + 6); sj(12 * u + 7);
    din[ib] = nf_mfma_bf16(ac[0], xm, din[ib]); sj(12 * u + 8); sj(12 * u + 9);
    din[ib] = nf_mfma_bf16(ac[0], xh, din[ib]); sj(12 * u + 10); sj(12 * u + 11);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// A C-layout tensor split ONCE into its bf16 triples, per k-group kg = 2 * block + (register >> 3): what dense_bwd_x_b6 does
// inside its loop, kept -- the producer of the pair kernel hands the same triples to the consumer's dW GEMM through LDS.
template <int NB>
struct SplitC {
  nf_u32x4 h[2 * NB], m[2 * NB], l[2 * NB];
};
template <int NB>
__device__ __forceinline__ void split_C(const f32x16 (&d)[NB], SplitC<NB> &s) {
#pragma unroll
  for (int kg = 0; kg < 2 * NB; ++kg) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = d[kg >> 1][8 * (kg & 1) + j];
    nf_split8(v, s.h[kg], s.m[kg], s.l[kg]);
  }
}
// dense_bwd_x_b6 on a cotangent that arrives split
template <int IB, int OB, class SJ = NoSideJob>
__device__ __forceinline__ void dense_bwd_x_b6s(const nf_u32x4 *__restrict__ w, const SplitC<OB> &ds, f32x16 (&din)[IB], int l31,
                                                int hi, SJ sj = SJ()) {
  constexpr int ROWS = 32 * IB, NKG = 2 * OB, NU = NKG * IB;
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int r = 0; r < 16; ++r) din[ib][r] = 0.f;
  const nf_u32x4 *wl = w + hi * ROWS + l31;
  nf_u32x4 an[3], ac[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) an[c] = wl[c * 2 * ROWS];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int kg = u / IB, ib = u % IB;
#pragma unroll
    for (int c = 0; c < 3; ++c) ac[c] = an[c];
    if (u + 1 < NU) {
      const int kg1 = (u + 1) / IB, ib1 = (u + 1) % IB;
#pragma unroll
      for (int c = 0; c < 3; ++c) an[c] = wl[(kg1 * 3 + c) * 2 * ROWS + ib1 * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
    din[ib] = nf_mfma_bf16(ac[2], ds.h[kg], din[ib]); sj(12 * u + 0); sj(12 * u + 1);
    din[ib] = nf_mfma_bf16(ac[0], ds.l[kg], din[ib]); sj(12 * u + 2); sj(12 * u + 3);
    din[ib] = nf_mfma_bf16(ac[1], ds.m[kg], din[ib]); sj(12 * u + 4); sj(12 * u + 5);
    din[ib] = nf_mfma_bf16(ac[1], ds.h[kg], din[ib]); sj(12 * u + 6); sj(12 * u + 7);
    din[ib] = nf_mfma_bf16(ac[0], ds.m[kg], din[ib]); sj(12 * u + 8); sj(12 * u + 9);
    din[ib] = nf_mfma_bf16(ac[0], ds.h[kg], din[ib]); sj(12 * u + 10); sj(12 * u + 11);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dense_bwd_x_b6s with the OUTPUT BLOCKS one after the other (round 6): block ib's matrix instructions carry a side job --
// sj(ib, i), i = 0 .. 12 NKG - 1, two calls behind every MFMA -- so that what follows a finished block (leaky-ReLU slopes,
// split, hand-over stores: PairPost below) runs in the issue shadows of the NEXT block's instructions instead of behind the
// whole GEMM.  The same terms in the same order per accumulator as dense_bwd_x_b6s: identical bits.
template <int IB, int OB, class SJ>
__device__ __forceinline__ void dense_bwd_x_b6s_blocks(const nf_u32x4 *__restrict__ w, const SplitC<OB> &ds, f32x16 (&din)[IB], int l31,
                                                       int hi, SJ sj) {
  constexpr int ROWS = 32 * IB, NKG = 2 * OB, NU = NKG * IB;
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int r = 0; r < 16; ++r) din[ib][r] = 0.f;
  const nf_u32x4 *wl = w + hi * ROWS + l31;
  nf_u32x4 an[3], ac[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) an[c] = wl[c * 2 * ROWS];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int ib = u / NKG, kg = u % NKG;
#pragma unroll
    for (int c = 0; c < 3; ++c) ac[c] = an[c];
    if (u + 1 < NU) {
      const int ib1 = (u + 1) / NKG, kg1 = (u + 1) % NKG;
#pragma unroll
      for (int c = 0; c < 3; ++c) an[c] = wl[(kg1 * 3 + c) * 2 * ROWS + ib1 * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
    din[ib] = nf_mfma_bf16(ac[2], ds.h[kg], din[ib]); sj(ib, 12 * kg + 0); sj(ib, 12 * kg + 1);
    din[ib] = nf_mfma_bf16(ac[0], ds.l[kg], din[ib]); sj(ib, 12 * kg + 2); sj(ib, 12 * kg + 3);
    din[ib] = nf_mfma_bf16(ac[1], ds.m[kg], din[ib]); sj(ib, 12 * kg + 4); sj(ib, 12 * kg + 5);
    din[ib] = nf_mfma_bf16(ac[1], ds.h[kg], din[ib]); sj(ib, 12 * kg + 6); sj(ib, 12 * kg + 7);
    din[ib] = nf_mfma_bf16(ac[0], ds.m[kg], din[ib]); sj(ib, 12 * kg + 8); sj(ib, 12 * kg + 9);
    din[ib] = nf_mfma_bf16(ac[0], ds.h[kg], din[ib]); sj(ib, 12 * kg + 10); sj(ib, 12 * kg + 11);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The transposed hand-over of such triples: the dW GEMM contracts over SAMPLES, so its delta operand wants lane <-> feature
// and a lane's eight k-values = eight samples.  LDS tile of one cotangent tensor: row = feature (D6_ROW bytes: three
// components x 64 bytes + 16 of padding -- 52 dwords, so eight consecutive rows' 16-byte reads cover the 32 banks once),
// inside a component [sample group g][sample parity][j] x 2 bytes with sample = 2 (8 g + j) + parity: the order the T layout
// of the stash gives the activation operand (dw_accumulate_reg_b6).  The writer holds a sample per lane and two features
// per packed register: one ds_write_b16 for the low half, one ds_write_b16_d16_hi for the high half, no unpacking.
constexpr int D6_ROW = 208, D6_BUF = 64 * D6_ROW;
template <int NB>
__device__ __forceinline__ void split_to_lds(char *__restrict__ buf, const SplitC<NB> &s, int l31, int hi) {
  const int t = l31 >> 1;
  char *p = buf + (4 * hi) * D6_ROW + (t >> 3) * 32 + (l31 & 1) * 16 + (t & 7) * 2;
#pragma unroll
  for (int kg = 0; kg < 2 * NB; ++kg)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const nf_u32x4 &v = c == 0 ? s.h[kg] : c == 1 ? s.m[kg] : s.l[kg];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j0 = 2 * q, j1 = 2 * q + 1;
        const int f0 = 16 * kg + (j0 & 3) + 8 * (j0 >> 2), f1 = 16 * kg + (j1 & 3) + 8 * (j1 >> 2);
        *reinterpret_cast<unsigned short *>(p + f0 * D6_ROW + c * 64) = (unsigned short)v[q];
        *reinterpret_cast<unsigned short *>(p + f1 * D6_ROW + c * 64) = (unsigned short)(v[q] >> 16);
      }
    }
}

// The activation operand of a dW GEMM (T layout of the stash: at[ib][t] = a[feature ib * 32 + l31][sample 2 t + hi]) as triples
template <int IB>
struct SplitT {  // the activation operand of a dW GEMM as bf16 triples: [block][sample group] x (h, m, l)
  nf_u32x4 h[IB][2], m[IB][2], l[IB][2];
};
template <int IB>
__device__ __forceinline__ void split_T(const float (&at)[IB][16], SplitT<IB> &s) {
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = at[ib][8 * g + j];
      nf_split8(v, s.h[ib][g], s.m[ib][g], s.l[ib][g]);
      __builtin_amdgcn_sched_barrier(0);  // one split at a time: interleaved, their temporaries (3 x 8 each) spill the accumulators
    }
}

// Both operands arrive split: the activation from split_T, the cotangent as the bf16 triples the PRODUCER wave made for its own
// dX GEMM and left in LDS transposed (split_to_lds, nf_mfma.h): three ds_read_b128 per (sample group, delta block) and no
// VALU work on the cotangent at all.  The bias gradient is the sum of the triples' components: v_dot2c_f32_bf16 against
// (1, 1) adds two bf16 values into an fp32 register per instruction.  Inline asm: with the literal as an operand hipcc 7.0's
// __builtin_amdgcn_fdot2_f32_bf16 emits the FIRST register of a vector for all four of its elements (seen in the ISA, and as
// NaN gradients on the device).
__device__ __forceinline__ float nf_dot2_bf16(unsigned x, unsigned y, float acc) {
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(x), "v"(y));
  return acc;
}
template <int IB, int OB, bool LEAN = false>
__device__ __forceinline__ void dw_accumulate_t6(const SplitT<IB> &as, const char *buf, f32x16 (&acc)[IB][OB],
                                                 float (&bsum)[OB], int l31, int hi) {
  const nf_u32x4 *pd = reinterpret_cast<const nf_u32x4 *>(buf + l31 * D6_ROW + hi * 16);
  constexpr int NU = 2 * OB, RB = 32 * D6_ROW / 16;  // 16-byte units per block of 32 rows
  nf_u32x4 dn[3], dc[3];
  const unsigned ones = 0x3F803F80u;  // bf16 (1, 1)
  if (!LEAN) {
#pragma unroll
    for (int c = 0; c < 3; ++c) dn[c] = pd[c * 4];
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int g = u / OB, ob = u % OB;
    if (LEAN) {  // no operand double buffer: 12 registers less, the LDS latency is left to the other wave of the SIMD
#pragma unroll
      for (int c = 0; c < 3; ++c) dc[c] = pd[ob * RB + c * 4 + g * 2];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) dc[c] = dn[c];
      if (u + 1 < NU) {
        const int g1 = (u + 1) / OB, ob1 = (u + 1) % OB;
#pragma unroll
        for (int c = 0; c < 3; ++c) dn[c] = pd[ob1 * RB + c * 4 + g1 * 2];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) {
        const nf_u32x4 &a = term == 0 ? as.l[ib][g] : (term == 2 || term == 3) ? as.m[ib][g] : as.h[ib][g];
        const nf_u32x4 &d = term == 1 ? dc[2] : (term == 2 || term == 4) ? dc[1] : dc[0];
        acc[ib][ob] = nf_mfma_bf16(a, d, acc[ib][ob]);
        if (term * IB + ib < 6) {  // the twelve bias-sum instructions ride between the unit's first MFMAs
          const int i0 = 2 * (term * IB + ib);
#pragma unroll
          for (int i = i0; i < i0 + 2; ++i) bsum[ob] = nf_dot2_bf16(dc[2 - i / 4][i % 4], ones, bsum[ob]);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- the same hand-over through gfx950's transposing LDS read (round 6) ------------------------------------------------
// split_to_lds transposes on the WRITE side: 8 two-byte stores per (k-group, component) and lane, 96 store instructions for a
// 64-feature cotangent -- at the end of the producer's stage, on the pair kernel's critical path, through a store path four
// producers share.  ds_read_b64_tr_b16 transposes on the READ side for free: within a 16-lane group, lane c receives element
// (c & 3) of what lane 4 j + (c >> 2) read, j = 0..3 (tools/probe/ds_read_tr_probe.hip) -- a 4 x 16 tile handed over
// column-wise.  So the writer keeps its natural layout, [sample][16 features] rows of 32 bytes per (16-feature tile =
// k-group, component): a lane holds, for its sample, features 4 hi .. 4 hi + 3 and 8 + 4 hi .. 8 + 4 hi + 3 of a k-group as
// two register pairs = TWO 8-byte stores per (k-group, component), 24 instead of 96 for 64 features; and the reader's
// lane (feature column c of its tile, jj = c >> 2, q = c & 3) reads the 8-byte chunk q of row hi + 16 g + 2 (jj + 4 r),
// r = 0, 1 -- the samples the T layout of the stashed activations pairs with element j = jj + 4 r of sample group g -- and
// receives its own column of those four rows: two reads per operand.  Chunk q of row s is stored at position q ^ ((s >> 2) & 3)
// (each lane supplies its own address, so rows and chunks may sit anywhere): a 16-lane write group then covers 16 different
// even banks, a read group 4 x 8 banks 16 apart, and with tiles 1 056 bytes apart the two groups of a half-wave interleave
// -- both directions conflict-free.  Buffer of one tensor: [component 3][tile 4] x 1 056 bytes.
constexpr int TR_TILE = 1056, TR_BUF = 12 * TR_TILE;
typedef short nf_s16x4 __attribute__((ext_vector_type(4)));
// NT: tiles per component of the buffer (the pair kernel's buffers hold up to two blocks: 4; k_rqs_bwd_coop6's hold one: 2)
template <int NB, int NT = 4>
__device__ __forceinline__ void split_to_lds_tr(char *__restrict__ buf, const SplitC<NB> &s, int l31, int hi) {
  static_assert(2 * NB <= NT, "tiles per component");
  const int x = (l31 >> 2) & 3;
  char *p0 = buf + l31 * 32 + 8 * (hi ^ x), *p1 = buf + l31 * 32 + 8 * ((2 + hi) ^ x);
#pragma unroll
  for (int kg = 0; kg < 2 * NB; ++kg)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const nf_u32x4 &v = c == 0 ? s.h[kg] : c == 1 ? s.m[kg] : s.l[kg];
      *reinterpret_cast<nf_u32x2 *>(p0 + (c * NT + kg) * TR_TILE) = nf_u32x2{v[0], v[1]};
      *reinterpret_cast<nf_u32x2 *>(p1 + (c * NT + kg) * TR_TILE) = nf_u32x2{v[2], v[3]};
    }
}
// one k-group's triples into tile `t` of a buffer with NT tiles per component (the per-k-group form)
template <int NT>
__device__ __forceinline__ void kg_to_lds_tr(char *__restrict__ buf, int t, nf_u32x4 h, nf_u32x4 m, nf_u32x4 l, int l31, int hi) {
  const int x = (l31 >> 2) & 3;
  char *p0 = buf + l31 * 32 + 8 * (hi ^ x) + t * TR_TILE, *p1 = buf + l31 * 32 + 8 * ((2 + hi) ^ x) + t * TR_TILE;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const nf_u32x4 &v = c == 0 ? h : c == 1 ? m : l;
    *reinterpret_cast<nf_u32x2 *>(p0 + c * NT * TR_TILE) = nf_u32x2{v[0], v[1]};
    *reinterpret_cast<nf_u32x2 *>(p1 + c * NT * TR_TILE) = nf_u32x2{v[2], v[3]};
  }
}
// the reader's two lane addresses in such a buffer (rows hi + 2 jj and + 8, chunk q at its swizzled position, the lane's tile of a block)
struct TrLane {
  const char *p0, *p1;
};
__device__ __forceinline__ TrLane nf_tr_lane(const char *buf, int l31, int hi) {
  const int c16 = l31 & 15, jj = c16 >> 2, q = c16 & 3, r0 = hi + 2 * jj, r1 = r0 + 8;
  return TrLane{buf + (l31 >> 4) * TR_TILE + r0 * 32 + 8 * (q ^ ((r0 >> 2) & 3)), buf + (l31 >> 4) * TR_TILE + r1 * 32 + 8 * (q ^ ((r1 >> 2) & 3))};
}
// What the pair kernel's producer does with a finished 32-feature block of a cotangent, in twenty steps of a few instructions
// each (the side job of dense_bwd_x_b6s_blocks): steps 0..7 pair p of the block -- its two leaky-ReLU slopes from the mask
// (nf_lrelu_grad16's form), the product, the three-way split of the pair; steps 8..19 the twelve 8-byte hand-over stores of
// the block's two k-groups (split_to_lds_tr's).  Steps beyond 19 do nothing.  `i` is a constant after unrolling.
template <int NB>
struct PairPost {
  f32x16 (&d)[NB];
  const unsigned (&msk)[NB];
  SplitC<NB> &s;
  char *p0, *p1;  // the lane's two chunk addresses in the hand-over buffer (split_to_lds_tr)
  __device__ __forceinline__ void step(int b, int i) const {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    if (i < 8) {
      const int p = i, kg = 2 * b + (p >> 2), e = p & 3;
      const unsigned one = 0x3F800000u, flip = 0x3F800000u ^ 0x3C23D70Au;
      int t0, t1;
      unsigned s0, s1;
      asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t0) : "v"(msk[b]), "n"(i < 8 ? 15 - 2 * i : 0));
      asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t1) : "v"(msk[b]), "n"(i < 8 ? 14 - 2 * i : 0));
      asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c" : "=v"(s0) : "v"(t0), "v"(one), "v"(flip));
      asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c" : "=v"(s1) : "v"(t1), "v"(one), "v"(flip));
      f32x2_t v = {d[b][2 * p], d[b][2 * p + 1]};
      const f32x2_t sl = {__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1)};
#ifdef NF_SLOPE_SCALAR
      asm("v_mul_f32 %0, %1, %2" : "=v"(v.x) : "v"(v.x), "v"(sl.x));
      asm("v_mul_f32 %0, %1, %2" : "=v"(v.y) : "v"(v.y), "v"(sl.y));
#else
      asm("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(sl));
#endif
      unsigned h, m, l;
      nf_split2(v.x, v.y, h, m, l);
      s.h[kg][e] = h; s.m[kg][e] = m; s.l[kg][e] = l;
    } else if (i < 20) {
      const int w = i - 8, kg = 2 * b + w / 6, c = (w % 6) / 2, half = w & 1;
      const nf_u32x4 &v = c == 0 ? s.h[kg] : c == 1 ? s.m[kg] : s.l[kg];
      char *q = (half ? p1 : p0) + (c * 4 + kg) * TR_TILE;
      *reinterpret_cast<nf_u32x2 *>(q) = half ? nf_u32x2{v[2], v[3]} : nf_u32x2{v[0], v[1]};
    }
  }
  __device__ __forceinline__ void all(int b) const {
#pragma unroll
    for (int i = 0; i < 20; ++i) step(b, i);
  }
  // hook h of NH (two behind every matrix instruction of the next block): the eight pair steps (14 instructions each) spread over
  // the whole block, a k-group's six stores behind its fourth pair
  template <int NH>
  static constexpr int hook_of(int i) {
    return i < 8 ? i * NH / 9 : i < 14 ? NH / 3 + 1 + (i - 8) * (NH / 3 - 1) / 6 : 7 * NH / 9 + 1 + (i - 14) * (NH - 7 * NH / 9 - 2) / 6;
  }
  template <int NH>
  __device__ __forceinline__ void at_hook(int b, int h) const {
#pragma unroll
    for (int i = 0; i < 20; ++i)
      if (hook_of<NH>(i) == h) step(b, i);
  }
};
// one MFMA operand: the lane's feature column over the eight samples of sample group g (two transposing reads)
__device__ __forceinline__ nf_u32x4 nf_tr_operand(const char *p0, const char *p1, int off) {
  typedef __attribute__((address_space(3))) nf_s16x4 lds_s16x4;
  const nf_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0 + off));
  const nf_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p1 + off));
  const nf_u32x2 lo = __builtin_bit_cast(nf_u32x2, a), hi2 = __builtin_bit_cast(nf_u32x2, b);
  return nf_u32x4{lo.x, lo.y, hi2.x, hi2.y};
}
// The split of the consumer's NEXT activation operand, a pair of values (nine vector instructions) per call, for the issue shadows of
// the GEMM that runs while it arrives: pair p of `at` at hook FIRST + EVERY p.  finish() splits what the hooks did not reach.
template <int IB, int FIRST, int EVERY>
struct SplitTJob {
  const float (&at)[IB][16];
  SplitT<IB> &s;
  __device__ __forceinline__ void pair(int p) const {
    const int ib = p >> 3, g = (p >> 2) & 1, e = p & 3;
    unsigned h, m, l;
    nf_split2(at[ib][8 * g + 2 * e], at[ib][8 * g + 2 * e + 1], h, m, l);
    s.h[ib][g][e] = h; s.m[ib][g][e] = m; s.l[ib][g][e] = l;
  }
  __device__ __forceinline__ void operator()(int i) const {
    if (i >= FIRST && (i - FIRST) % EVERY == 0 && (i - FIRST) / EVERY < 8 * IB) {
      pair((i - FIRST) / EVERY);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  template <int NH>  // NH: number of hooks the GEMM offered
  __device__ __forceinline__ void finish() const {
    constexpr int done = NH <= FIRST ? 0 : (NH - 1 - FIRST) / EVERY + 1;
#pragma unroll
    for (int p = done < 8 * IB ? done : 8 * IB; p < 8 * IB; ++p) pair(p);
  }
};
// dw_accumulate_t6 on a cotangent left by split_to_lds_tr
#ifndef NF_DW_DBUF
#define NF_DW_DBUF 0
#endif
// (`sj(i)` is called behind the i-th of the 12 IB OB matrix instructions: the consumer's side job is the split of its NEXT
// activation operand, SplitTJob below)
template <int IB, int OB, class SJ = NoSideJob>
__device__ __forceinline__ void dw_accumulate_tr6(const SplitT<IB> &as, const char *buf, f32x16 (&acc)[IB][OB],
                                                  float (&bsum)[OB], int l31, int hi, SJ sj = SJ()) {
  // lane offsets of the two reads: row hi + 2 jj (+ 8 for the second), chunk q at its swizzled position, the lane's tile of the pair
  const int c16 = l31 & 15, jj = c16 >> 2, q = c16 & 3, r0 = hi + 2 * jj, r1 = r0 + 8;
  const char *p0 = buf + (l31 >> 4) * TR_TILE + r0 * 32 + 8 * (q ^ ((r0 >> 2) & 3));
  const char *p1 = buf + (l31 >> 4) * TR_TILE + r1 * 32 + 8 * (q ^ ((r1 >> 2) & 3));
  constexpr int NU = 2 * OB;
  nf_u32x4 dc[3];
  const unsigned ones = 0x3F803F80u;  // bf16 (1, 1)
#if NF_DW_DBUF  // the next unit's operands requested behind the current unit's matrix instructions (a second operand set: 12 registers)
  nf_u32x4 dn[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) dn[c] = nf_tr_operand(p0, p1, (c * 4) * TR_TILE);
#endif
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int g = u / OB, ob = u % OB;
#if NF_DW_DBUF
#pragma unroll
    for (int c = 0; c < 3; ++c) dc[c] = dn[c];
    if (u + 1 < NU) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dn[c] = nf_tr_operand(p0, p1, (c * 4 + 2 * ((u + 1) % OB)) * TR_TILE + 512 * ((u + 1) / OB));
    }
#else
    // (no operand double buffer: with the woven consumer of round 6's first half the twelve registers of a second operand set cost
    // 28 bytes of scratch around the tile loop, the exposed LDS round trip per unit nothing measurable)
#pragma unroll
    for (int c = 0; c < 3; ++c) dc[c] = nf_tr_operand(p0, p1, (c * 4 + 2 * ob) * TR_TILE + 512 * g);
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) {
        const nf_u32x4 &a = term == 0 ? as.l[ib][g] : (term == 2 || term == 3) ? as.m[ib][g] : as.h[ib][g];
        const nf_u32x4 &d = term == 1 ? dc[2] : (term == 2 || term == 4) ? dc[1] : dc[0];
        acc[ib][ob] = nf_mfma_bf16(a, d, acc[ib][ob]);
        if (term * IB + ib < 6) {  // the twelve bias-sum instructions ride between the unit's first MFMAs
          const int i0 = 2 * (term * IB + ib);
#pragma unroll
          // (v_dot2c_f32_bf16 does not overlap a matrix instruction in flight -- tools/probe/mfma_valu_overlap_probe.hip --, but the
          // four-instruction widen-and-add form that does costs the consumer 92-200 bytes of scratch: 370 against 330 us)
          for (int i = i0; i < i0 + 2; ++i) bsum[ob] = nf_dot2_bf16(dc[2 - i / 4][i % 4], ones, bsum[ob]);
        }
        sj(u * 6 * IB + term * IB + ib);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- tile I/O through buffer descriptors ---------------------------------------------------
// One descriptor per (array, tile): base = array + tile * d * 32 floats (wave-uniform), extent =
// d * 32 floats.  Element (feature f, lane's sample) is at byte f * 128 + (lane & 31) * 4, and in
// the MFMA register layout f = const(block, reg) + 8 * (lane >> 5) + parity, so ONE per-lane
// voffset serves every access and the rest is a scalar offset: no 64-bit per-lane addresses.
// Features >= d fall outside the descriptor: the hardware returns 0 for such loads and drops such
// stores, which is exactly the PartitionMask edge handling (odd d, c != m) -- no branches.
struct TileIO {
  __amdgpu_buffer_rsrc_t rs;
  int voff;
};
__device__ __forceinline__ TileIO make_tile_io(float *array, long tile, int d, int l31, int hi) {
  TileIO t;
  t.rs = __builtin_amdgcn_make_buffer_rsrc(array + tile * d * NF_TILE, 0, d * NF_TILE * 4, 0x00020000);
  t.voff = l31 * 4 + hi * (8 * NF_TILE * 4);
  return t;
}
// feature index without the lane-dependent 8*hi term: f0 = 2 * (b*32 + (r&3) + 8*(r>>2)) + parity
__device__ __forceinline__ int tile_soff(int b, int r, int parity) {
  return (2 * (b * 32 + (r & 3) + 8 * (r >> 2)) + parity) * (NF_TILE * 4);
}
__device__ __forceinline__ float tile_load(const TileIO &t, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.rs, t.voff, soff, 0));
}
__device__ __forceinline__ void tile_store(const TileIO &t, int soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), t.rs, t.voff, soff, 0);
}

