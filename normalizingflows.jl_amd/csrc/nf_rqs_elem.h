// The rational-quadratic spline of one (dimension, sample) in registers: knots from the raw parameters, bin search, forward,
// inverse and reverse pass (MonotonicSplines 0.3.3's algebra as restated in oracle/nf_oracle.py; reference call sites
// src/flows/neuralspline.jl:102-108).  Shared by the fused spline kernels (nf_rqs.hip) and, from round 5, by the general
// path's fused output-layer kernels (nf_generic64.hip: k_l64_nsf_top_fwd / k_l64_nsf_top_bwd) -- moved here unchanged.
// Needs nf_common.h (nf_log) and nf_mfma.h (nf_fdiv) before it.
#pragma once

// ---------------------------------------------------------------------------------------
// the spline, one (dim, sample) per lane, everything in registers
// ---------------------------------------------------------------------------------------
// softplus with hardware exp/log: abs error ~1e-7 (the derivatives it produces are O(1))
__device__ __forceinline__ float softplus_f(float x) { return nf_log(1.f + __expf(-fabsf(x))) + fmaxf(x, 0.f); }
__device__ __forceinline__ float sigmoid_f(float x) {
  const float e = __expf(-fabsf(x));
  return x >= 0.f ? nf_fdiv(1.f, 1.f + e) : nf_fdiv(e, 1.f + e);
}

// The knots of one (dim, sample), UNNORMALISED and in (width, height) pairs (round 4).  With e_k = exp(raw_k - max) and
// the inclusive prefix sums cs_k = e_0 + .. + e_k (cs_{K-1} = the softmax denominator), knot j is
//     p_j = -B + 2B cs_{j-1} / cs_{K-1} = fma(cs_{j-1}, sc, -B),   sc = 2B / cs_{K-1},   p_0 = -B, p_K = B,
// and bin k spans [p_k, p_k + e_k sc).  Nothing else is ever formed: no normalised softmax vector (the reverse pass folds
// 1 / cs_{K-1} into three coefficients instead of into K weights), no knot vector (the bin search compares the element,
// mapped into prefix-sum space once, with the prefix sums; only the bin's own two knots are evaluated).  Widths and heights
// go through identical arithmetic, so they are carried as two-wide values and hipcc issues v_pk_fma / v_pk_add / v_pk_mul
// for them: 24 packed + 18 scalar VALU instructions and 18 transcendentals per element, against 106 + 18 before
// (softmax x2: max, exp, sum, normalise, cumsum, knots).  The kernels that evaluate splines are VALU-issue-bound next to
// their MFMAs (fp32 MFMA and VALU do not overlap), so instructions are what counts.
// The knot DERIVATIVES are never built as a vector either: only the two at the ends of the bin an element falls into are
// used, so find_bin selects their raw parameters and evaluates those two softplus' (LAZY; EAGER: see build_knots).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int K>
struct Knots {
  f32x2 e[K], cs[K];  // (width, height) pairs
  f32x2 sc;           // 2B / cs[K-1]
  f32x2 usc;          // cs[K-1] / 2B: element -> prefix-sum space, u = (v + B) * usc
  float B;
  const float *rawd;  // raw[2K .. 3K-2]: interior derivative parameters (registers, compile-time indexed)
  float dd[K + 1];    // EAGER mode only (the per-wave reverse kernel): all knot derivatives
};

constexpr float NF_LOG2E = 1.4426950408889634f;

// raw[0:K] widths, raw[K:2K] heights, raw[2K:3K-1] interior derivatives.
// LAZY (forward / inverse chain, cooperative reverse kernel): the two derivatives an element needs are evaluated by
// find_bin.  EAGER (per-wave reverse kernel): all of them here -- measured on that kernel, the lazy form keeps the raw
// derivative parameters live through the bin selection and tips hipcc's allocation over the register wall (224 dW
// accumulators): 496 B of scratch spills, 204 instead of 166 us per launch, although it executes 10 % fewer instructions.
template <int K, bool LAZY = true>
__device__ __forceinline__ void build_knots(const float *raw, float B, Knots<K> &kn) {
  float mw = raw[0], mh = raw[K];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    mw = fmaxf(mw, raw[k]);
    mh = fmaxf(mh, raw[K + k]);
  }
  const f32x2 nm = f32x2{-mw, -mh} * NF_LOG2E;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const f32x2 arg = f32x2{raw[k], raw[K + k]} * NF_LOG2E + nm;  // (v - max) log2 e: one packed fma
    kn.e[k] = f32x2{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
  }
  kn.cs[0] = kn.e[0];
#pragma unroll
  for (int k = 1; k < K; ++k) kn.cs[k] = kn.cs[k - 1] + kn.e[k];
  const f32x2 inv = f32x2{__builtin_amdgcn_rcpf(kn.cs[K - 1][0]), __builtin_amdgcn_rcpf(kn.cs[K - 1][1])};
  kn.sc = inv * (2.f * B);
  kn.usc = kn.cs[K - 1] * nf_fdiv(0.5f, B);
  kn.B = B;
  kn.rawd = raw + 2 * K;
  if (!LAZY) {
    kn.dd[0] = 1.f;
    kn.dd[K] = 1.f;
#pragma unroll
    for (int k = 1; k < K; ++k) kn.dd[k] = softplus_f(raw[2 * K + k - 1]);
  }
}

// The bin an element falls into: its left knots (xk, yk), its extent (dx, dy), the two knot derivatives, and the
// CONDITIONS ge[j] <=> "the element is at or beyond knot j" (ge[0] = true, ge[K] = false).  The knot vector is increasing,
// so the bin index is k = #{j in 1..K-1 : ge[j]} and
//     i <  k  <=>  ge[i + 1],        i == k  <=>  ge[i] && !ge[i + 1]
// -- every "is this the bin / is this left of the bin" test of the reverse pass is one of these lane masks,
// already sitting in scalar registers; no integer index is ever formed or compared.
template <int K>
struct Bin {
  float xk, dx, yk, dy, d0, d1;
  bool ge[K + 1];
  bool inside;
  unsigned code;  // bin index k (0 .. K-1), or NF_RQS_OUTSIDE: what the forward leaves behind for the reverse pass
};
#define NF_RQS_OUTSIDE 15u

// The spline tape.  Which bin an element falls into is a DISCRETE decision, and the gradient of log S'(x) with respect
// to the knot parameters jumps across a knot (S is only C^1).  A reverse pass that re-decides the bin on a float32
// reconstruction of the element (inverting the coupling from its output) can land on the other side of a knot for an
// element within ~1e-6 of it -- measured: one such element in a 77-sample batch moved the gradient of a coupling by
// 2e-2 |g|inf.  The reference differentiates the forward's own tape (MonotonicSplines' rrule pullbacks receive the
// forward's x), so the chain kernels leave, per slot = (tile, coupling), NE rows of 64 floats (NE = NCH * QCH elements per
// lane, element e = chunk * QCH + local dim): base[(slot * NE + e) * 64 + lane] = k + xi, the bin index plus the in-bin
// coordinate (clamped to [0, 1 - 2^-19] so that the sum keeps its integer part; xi then carries 19-20 bits, ~1e-6 of the
// bin), or -1 outside the box.  One float per element: the chain kernels sit at their register budget, and a separate
// word of packed bin codes cost them 240 bytes of scratch spills per lane.  The reverse kernels read it back instead of
// searching and solving again (which also saves them the inverse's quadratic).
struct RqsTape {
  float *base;
};
__device__ __forceinline__ float rqs_tape_encode(unsigned code, float xi) {
  const float t = (float)code + fminf(fmaxf(xi, 0.f), 0.99999809265f);
  return code == NF_RQS_OUTSIDE ? -1.f : t;
}
__device__ __forceinline__ void rqs_tape_decode(float t, unsigned &code, float &xi) {
  const float f = floorf(fmaxf(t, 0.f));
  code = t < 0.f ? NF_RQS_OUTSIDE : (unsigned)f;
  xi = t - f;
}

// bin with p[k] <= v < p[k+1] along AXIS (0: the x knots, forward; 1: the y knots, inverse): one ascending select chain
// over the prefix sums.  FROM_CODE: the conditions come from a recorded bin code instead of a search (v is not read).
template <int K, bool LAZY = true, bool FROM_CODE = false, int AXIS = 0>
__device__ __forceinline__ void find_bin(const Knots<K> &kn, float v, Bin<K> &b, unsigned code_in = 0u) {
  static_assert(K <= 15, "bin codes are four bits");
  const float B = kn.B;
  b.inside = FROM_CODE ? (code_in != NF_RQS_OUTSIDE) : ((v >= -B) && (v < B));
  b.code = 0u;
  b.ge[0] = true;
  b.ge[K] = false;
  const float u = FROM_CODE ? 0.f : (v + B) * kn.usc[AXIS];  // v >= knot j  <=>  u >= cs[j-1]
  f32x2 csp = {0.f, 0.f}, ek = kn.e[0];
  float r0 = 0.f, r1 = LAZY ? kn.rawd[0] : 0.f;  // raw derivative parameters of knots k and k+1 (knot j <-> rawd[j-1])
  if (!LAZY) {
    b.d0 = kn.dd[0];
    b.d1 = kn.dd[1];
  }
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const bool c = FROM_CODE ? (code_in >= (unsigned)j) : (u >= kn.cs[j - 1][AXIS]);
    b.ge[j] = c;
    if (!FROM_CODE) b.code = c ? (unsigned)j : b.code;
    csp[0] = c ? kn.cs[j - 1][0] : csp[0];
    csp[1] = c ? kn.cs[j - 1][1] : csp[1];
    ek[0] = c ? kn.e[j][0] : ek[0];
    ek[1] = c ? kn.e[j][1] : ek[1];
    if (LAZY) {
      r0 = c ? kn.rawd[j - 1] : r0;
      if (j < K - 1) r1 = c ? kn.rawd[j] : r1;
    } else {
      b.d0 = c ? kn.dd[j] : b.d0;
      b.d1 = c ? kn.dd[j + 1] : b.d1;
    }
  }
  const f32x2 pk = csp * kn.sc - B, dk = ek * kn.sc;
  b.xk = pk[0]; b.yk = pk[1];
  b.dx = dk[0]; b.dy = dk[1];
  if (LAZY) {
    b.d0 = b.ge[1] ? softplus_f(r0) : 1.f;      // k == 0: boundary derivative 1
    b.d1 = b.ge[K - 1] ? 1.f : softplus_f(r1);  // k == K-1: boundary derivative 1
  }
  b.code = FROM_CODE ? code_in : (b.inside ? b.code : NF_RQS_OUTSIDE);
}

__device__ __forceinline__ float rq_logderiv(float s, float d0, float d1, float xi) {
  const float om = 1.f - xi;
  const float den = s + (d1 + d0 - 2.f * s) * xi * om;
  // 2 log s + log(nd) - 2 log den as ONE logarithm: log(s^2 nd / den^2) (one v_rcp + one v_log instead of three logs)
  const float nd = d1 * xi * xi + 2.f * s * xi * om + d0 * om * om;
#ifdef RQS_THREE_LOGS
  return 2.f * nf_log(s) + nf_log(nd) - 2.f * nf_log(den);
#else
  return nf_log(nf_fdiv(s * s * nd, den * den));
#endif
}

// rqs_forward for one element: returns y, adds log dy/dx to logd
template <int K>
__device__ __forceinline__ float rqs_fwd_elem(const Knots<K> &kn, float x, float &logd, unsigned &code_out, float &xi_out) {
  Bin<K> b;
  find_bin<K>(kn, x, b);
  code_out = b.code;
  const float dx = b.dx, dy = b.dy;
  const float s = nf_fdiv(dy, dx);
  const float xi = nf_fdiv(x - b.xk, dx), om = 1.f - xi;
  const float den = s + (b.d1 + b.d0 - 2.f * s) * xi * om;
  const float y = b.yk + nf_fdiv(dy * (s * xi * xi + b.d0 * xi * om), den);
  xi_out = xi;
  logd += b.inside ? rq_logderiv(s, b.d0, b.d1, xi) : 0.f;
  return b.inside ? y : x;
}

// rqs_inverse for one element: returns x and the bin / xi it lies in; adds -log dy/dx to logd when WANT_LOGD
template <int K, bool WANT_LOGD = true, bool LAZY = true>
__device__ __forceinline__ float rqs_inv_elem(const Knots<K> &kn, float y, float &logd, Bin<K> &b, float &xi_out) {
  find_bin<K, LAZY, false, 1>(kn, y, b);
  const float dx = b.dx, dy = b.dy;
  const float s = nf_fdiv(dy, dx);
  const float yy = y - b.yk;
  const float q = b.d1 + b.d0 - 2.f * s;
  const float a = dy * (s - b.d0) + yy * q;
  const float bb = dy * b.d0 - yy * q;
  const float c = -s * yy;
  const float disc = fmaxf(bb * bb - 4.f * a * c, 0.f);
  const float xi = nf_fdiv(2.f * c, -bb - __builtin_amdgcn_sqrtf(disc));
  xi_out = xi;
  if (WANT_LOGD) logd -= b.inside ? rq_logderiv(s, b.d0, b.d1, xi) : 0.f;
  return b.inside ? xi * dx + b.xk : y;
}

// reverse pass of rqs_forward at (x in bin b, xi): ybar, lbar -> xbar and raw-parameter gradients.
// INVD: reverse pass of the INVERSE spline at its output x (same point), (ybar, lbar) = cotangents of
// (x, ladj_inv = -log S'(x)); implicit-function form vbar = (ybar - lbar dlogS'/dx) / S', parameters =
// the forward formulas with (-vbar, -lbar).  Returns vbar.
template <int K, bool INVD = false>
__device__ __forceinline__ float rqs_bwd_elem(const Knots<K> &kn, const Bin<K> &b, float xi, float B, float ybar,
                                              float lbar, float *thbar) {
  const float dx = b.dx, dy = b.dy;
  const float s = nf_fdiv(dy, dx), om = 1.f - xi;
  const float d0 = b.d0, d1 = b.d1;
  const float q = d1 + d0 - 2.f * s;
  const float xo = xi * om;
  const float den = s + q * xo;
  const float num = s * xi * xi + d0 * xo;
  const float nd = d1 * xi * xi + 2.f * s * xo + d0 * om * om;
  const float iden = nf_fdiv(1.f, den), ind = nf_fdiv(1.f, nd), idx = nf_fdiv(1.f, dx);
  const float iden2 = iden * iden;
  const float t12 = 1.f - 2.f * xi;
  const float dnum_dxi = 2.f * s * xi + d0 * t12;
  const float dden_dxi = q * t12;
  const float dnd_dxi = 2.f * d1 * xi + 2.f * s * t12 - 2.f * d0 * om;
  const float dyi = dy * iden2;
  const float dy_dxi = dyi * (dnum_dxi * den - num * dden_dxi);
  const float dL_dxi = dnd_dxi * ind - 2.f * dden_dxi * iden;
  const float dden_ds = 1.f - 2.f * xo;
  const float dy_ds = dyi * (xi * xi * den - num * dden_ds);
  const float dL_ds = nf_fdiv(2.f, s) + 2.f * xo * ind - 2.f * dden_ds * iden;
  const float nxo = num * xo;
  const float dy_dd0 = dyi * (xo * den - nxo);
  const float dL_dd0 = om * om * ind - 2.f * xo * iden;
  const float dy_dd1 = -dyi * nxo;
  const float dL_dd1 = xi * xi * ind - 2.f * xo * iden;
  // every parameter cotangent is linear in (ybar, lbar): zeroing them outside the box (identity
  // branch) makes all of them vanish without per-parameter selects
  float yb = b.inside ? ybar : 0.f, lbr = b.inside ? lbar : 0.f;
  float vbar = 0.f;
  if (INVD) {
    vbar = b.inside ? nf_fdiv(yb - lbr * dL_dxi * idx, dy_dxi * idx) : 0.f;
    yb = -vbar;
    lbr = -lbr;
  }
  const float xibar = yb * dy_dxi + lbr * dL_dxi;
  const float sbar = yb * dy_ds + lbr * dL_ds;
  const float d0bar = yb * dy_dd0 + lbr * dL_dd0;
  const float d1bar = yb * dy_dd1 + lbr * dL_dd1;
  const float dybar = yb * num * iden + sbar * idx;
  const float dxbar = -sbar * s * idx - xibar * xi * idx;
  const float xkbar = -xibar * idx - dxbar, xk1bar = dxbar;
  const float ykbar = yb - dybar, yk1bar = dybar;
  // knots: p[j] = -B + 2B sum_{i<j} sm_i  =>  dL/dsm_i = 2B * (pbar[k] + pbar[k+1]) for i < k, 2B * pbar[k+1] for
  // i == k, 0 beyond.  The softmax pullback needs dot = sum_i dL/dsm_i * sm_i, which has a closed form:
  // sum_{i<k} sm_i = (p[k] + B) / 2B and sm_k = (p[k+1] - p[k]) / 2B, both already in hand.  The raw-parameter cotangent
  // is sm_i (dL/dsm_i - dot) with sm_i = e_i / cs_{K-1}: the normalisation goes into the three values the bracket can take
  // (i < k, i == k, i > k), and with L_i = [i < k] as a float the bracket is  c_gt + L_i (c_lt - c_eq) + L_{i-1} (c_eq - c_gt)
  // -- two packed fma and one packed multiply per i for widths and heights together, K - 1 selects for the L's.
  const float twoB = 2.f * B;
  const float aw = xkbar + xk1bar, ah = ykbar + yk1bar;
  const float dotw = fmaf(aw, b.xk + B, xk1bar * dx);
  const float doth = fmaf(ah, b.yk + B, yk1bar * dy);
  const f32x2 inv = kn.sc * nf_fdiv(1.f, twoB);  // 1 / cs[K-1]
  const f32x2 c_lt = f32x2{twoB * aw - dotw, twoB * ah - doth} * inv;
  const f32x2 c_eq = f32x2{twoB * xk1bar - dotw, twoB * yk1bar - doth} * inv;
  const f32x2 c_gt = f32x2{-dotw, -doth} * inv;
  const f32x2 d_le = c_lt - c_eq, d_eg = c_eq - c_gt;
  float Lprev = 1.f;  // L_{-1}
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const float Li = (i + 1 < K) ? (b.ge[i + 1] ? 1.f : 0.f) : 0.f;
    const f32x2 t = kn.e[i] * ((c_gt + d_le * Li) + d_eg * Lprev);
    thbar[i] = t[0];
    thbar[K + i] = t[1];
    Lprev = Li;
  }
  // d/draw softplus = sigmoid(raw) = 1 - exp(-softplus(raw)); only knots k and k+1 carry a cotangent:
  // knot j is the bin's left end iff ge[j] && !ge[j+1], its right end iff ge[j-1] && !ge[j]
  const float g0 = d0bar * (1.f - __expf(-d0)), g1 = d1bar * (1.f - __expf(-d1));
#pragma unroll
  for (int j = 1; j < K; ++j) thbar[2 * K + j - 1] = b.ge[j + 1] ? 0.f : (b.ge[j] ? g0 : (b.ge[j - 1] ? g1 : 0.f));
  return b.inside ? (INVD ? vbar : xibar * idx) : ybar;
}
