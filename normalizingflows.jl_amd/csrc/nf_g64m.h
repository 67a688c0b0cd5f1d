// nf_g64m.h -- Float64 RealNVP couplings on the f64 matrix instruction (round 5).  Included by nf_generic64.hip (G64Args).
//
// The reference's DEFAULT constructors are Float64 (realnvp(q0; paramtype = Float64): src/flows/realnvp.jl:190-192;
// test/flow.jl:7 runs both element types).  Rounds 1-4 ran Float64 couplings one thread per sample with the MLP as scalar FMA
// loops over per-thread arrays in scratch memory (k_g64_apply / k_g64_bwd; round 4 moved the weight gradients to the matrix
// pipe): 23.2 ms per step at d = 64, hidden [64, 64], 65 536 samples -- 40 x the Float32 step, where the two matrix pipes
// differ by 2 x (VERDICT r4 missing 4).  Here the whole MLP runs on v_mfma_f64_16x16x4_f64 with the register chaining of
// nf_mfma.h carried over to the f64 fragment layout (tools/probe/mfma_f64_probe.hip):
//   lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15], receives D[(l >> 4) + 4 r][l & 15], r = 0..3.
//   A wavefront owns a tile of 16 samples (j = l & 15); a 16-feature block of an activation is one f64x4 per lane, register r
//   <-> feature q + 4 r (q = l >> 4): the D layout.  Contracting the next layer's k-step t over the features q + 4 t makes
//   B-operand t of a block identical to its register t -- activations never leave the registers; only the weight (A) operands
//   are fetched, from a padded f64 image of the net in LDS (row = input feature, row stride = 16 OB + 2 doubles: the
//   transposed fetch of the dX GEMM is bank-conflict free, the forward fetch two-way -- 4 LDS cycles per 64-clock MFMA).
//   The weight-gradient GEMM contracts over SAMPLES (lane <-> feature for both operands): the four waves of a workgroup
//   leave a layer's input and cotangent tiles in LDS as [sample][feature], then SPLIT THE 16 x 16 BLOCKS of dW among them
//   and each contracts its blocks over all 64 samples of the group (16 MFMAs per block) -- so a wave holds a quarter of a
//   net's dW accumulators (64 registers at 32-64-64-32) instead of all of them (256), two workgroup barriers per layer.
// Shapes: one or two hidden layers up to 64 wide, d <= 64 (conditioner and transformed half up to 32).  Standard batch layout
// x[j d + i] (the general path's), the coupling's INPUT kept by the caller as the tape (as k_g64_bwd).  Gradient slabs in theta
// order, reduced by nf_launch_reduce_slabs in a fixed order: deterministic.  Everything else Float64 (NSF, deeper / wider nets)
// keeps the scalar kernels.
#pragma once

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int MB_, int HB_, int CB_ = MB_>
struct G64M {
  static constexpr int MB = MB_, HB = HB_, CB = CB_;  // blocks of 16: conditioner / hidden / net outputs (RealNVP: the transformed
                                                      // half, = MB; neural spline: (3K - 1) parameters per transformed dim)
  static constexpr int SH = 16 * HB + 2, SC = 16 * CB + 2;  // row strides (doubles) of layers ending in a hidden / the output layer
  // image of a net with NHID hidden layers: W0 [16 MB][SH] b0 [16 HB] | (W1 [16 HB][SH] b1 [16 HB]) | Wo [16 HB][SC] bo [16 CB]
  static constexpr int W0 = 0, B0 = W0 + 16 * MB * SH;
  static constexpr int W1 = B0 + 16 * HB, B1 = W1 + 16 * HB * SH;
  static constexpr int img_wo(int nhid) { return nhid == 2 ? B1 + 16 * HB : W1; }
  static constexpr int img_bo(int nhid) { return img_wo(nhid) + 16 * HB * SC; }
  static constexpr int img_size(int nhid) { return (img_bo(nhid) + 16 * CB + 1) / 2 * 2; }
  static constexpr int ST = 16 * ((HB > MB ? HB : MB) > CB ? (HB > MB ? HB : MB) : CB) + 2;  // row stride of the [sample][feature] tiles of the dW stage
  static constexpr int TILE = 16 * ST;                       // doubles per tile
  static constexpr size_t apply_lds(int nhid) { return (size_t)img_size(nhid) * 8; }
  static constexpr size_t bwd_lds(int nhid) { return (size_t)(img_size(nhid) + 2 * 4 * TILE) * 8; }
};

// Stage the Dense layers theta[in][out] (Flux: out x in column-major) of a net into their padded image rows, zero fill.
// Sizes are template arguments, every thread's elements of ALL layers are requested before the first is stored (round 5): with
// run-time sizes and one element per iteration every element cost an integer division and its own exposed round trip to L2 --
// tools/trace_g64m.py showed the staging at 30 k of a pass's 70-84 k clocks in k_g64m_apply (39 % of the launch; all
// workgroups read the same 64 KB at the same moment, so a round trip is ~2.5 k clocks).
template <int ROWS, int S>
struct G64MStage {
  static constexpr int NE = ROWS * S, U = (NE + 255) / 256;
  double v[U];
  __device__ __forceinline__ void load(const double *__restrict__ theta, long w_off, int nin, int nout, int tid) {
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int e = tid + 256 * k, i = e / S, o = e - i * S;  // (S is a constant: multiply-shift)
      v[k] = (e < NE && i < nin && o < nout) ? theta[w_off + (long)i * nout + o] : 0.0;
    }
  }
  // a window of the layer's columns: [c0, c0 + ncols) of `nout`, into image columns 0 .. ncols - 1 (zero beyond)
  __device__ __forceinline__ void load_cols(const double *__restrict__ theta, long w_off, int nin, int nout, int c0, int ncols, int tid) {
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int e = tid + 256 * k, i = e / S, o = e - i * S;
      v[k] = (e < NE && i < nin && o < ncols) ? theta[w_off + (long)i * nout + c0 + o] : 0.0;
    }
  }
  __device__ __forceinline__ void store(double *__restrict__ w, int tid) const {
#pragma unroll
    for (int k = 0; k < U; ++k)
      if (tid + 256 * k < NE) w[tid + 256 * k] = v[k];
  }
};
template <int COLS>
__device__ __forceinline__ void g64m_stage_bias(double *__restrict__ b, const double *__restrict__ theta, long b_off, int nout, int tid) {
  for (int o = tid; o < COLS; o += 256) b[o] = o < nout ? theta[b_off + o] : 0.0;
}
template <class G>
__device__ __forceinline__ void g64m_stage_net(double *__restrict__ img, const double *__restrict__ theta, const G64Net &n, int tid,
                                               bool with_out = true) {
  const int nhid = n.nl - 1;
  G64MStage<16 * G::MB, G::SH> l0;
  G64MStage<16 * G::HB, G::SH> l1;
  G64MStage<16 * G::HB, G::SC> lo;
  l0.load(theta, n.w[0], n.dims[0], n.dims[1], tid);
  if (nhid == 2) l1.load(theta, n.w[1], n.dims[1], n.dims[2], tid);
  if (with_out) lo.load(theta, n.w[nhid], n.dims[nhid], n.dims[nhid + 1], tid);
  g64m_stage_bias<16 * G::HB>(img + G::B0, theta, n.b[0], n.dims[1], tid);
  if (nhid == 2) g64m_stage_bias<16 * G::HB>(img + G::B1, theta, n.b[1], n.dims[2], tid);
  if (with_out) g64m_stage_bias<16 * G::CB>(img + G::img_bo(nhid), theta, n.b[nhid], n.dims[nhid + 1], tid);
  l0.store(img + G::W0, tid);
  if (nhid == 2) l1.store(img + G::W1, tid);
  if (with_out) lo.store(img + G::img_wo(nhid), tid);
}
// Columns [c0, c0 + ncols) of the OUTPUT layer into the image's output-layer block (round 6: spline couplings whose
// (3K - 1) c raw parameters exceed the block's 16 CB columns take the layer in passes of whole dimensions -- the
// reference's default nsf(q0) = [32, 32], K = 10 has 29 per dimension, 464 at d = 32; src/flows/neuralspline.jl:232-234).
// The caller brackets this with workgroup barriers.
template <class G>
__device__ __forceinline__ void g64m_stage_out_cols(double *__restrict__ img, const double *__restrict__ theta, const G64Net &n, int c0,
                                                    int ncols, int tid) {
  const int nhid = n.nl - 1, nout = n.dims[nhid + 1];
  G64MStage<16 * G::HB, G::SC> lo;
  lo.load_cols(theta, n.w[nhid], n.dims[nhid], nout, c0, ncols, tid);
  double *b = img + G::img_bo(nhid);
  for (int o = tid; o < 16 * G::CB; o += 256) b[o] = o < ncols ? theta[n.b[nhid] + c0 + o] : 0.0;
  lo.store(img + G::img_wo(nhid), tid);
}
// dimensions per pass / passes of a spline coupling with P = 3K - 1 parameters per dimension and c transformed dimensions
template <class G>
__device__ __host__ inline int g64m_nsf_dp(int P) { return (16 * G::CB) / P; }

__device__ __forceinline__ f64x4 g64m_mfma(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// out[ob] = W' in + b.  w: [16 IB rows][S], b: [16 OB]
template <int IB, int OB, int S>
__device__ __forceinline__ void g64m_fwd(const double *__restrict__ w, const double *__restrict__ b, const f64x4 (&in)[IB], f64x4 (&out)[OB],
                                         int c16, int q) {
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[ob][r] = b[16 * ob + q + 4 * r];
  const double *wl = w + q * S + c16;
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) out[ob] = g64m_mfma(wl[(16 * ib + 4 * t) * S + 16 * ob], in[ib][t], out[ob]);
}
// din[ib] (+)= W delta (the dX GEMM): rows of w are the layer's inputs
template <int IB, int OB, int S, bool ACCUM>
__device__ __forceinline__ void g64m_bwdx(const double *__restrict__ w, const f64x4 (&delta)[OB], f64x4 (&din)[IB], int c16, int q) {
  if (!ACCUM) {
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) din[ib] = f64x4{0.0, 0.0, 0.0, 0.0};
  }
  const double *wl = w + c16 * S + q;
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) din[ib] = g64m_mfma(wl[16 * ib * S + 16 * ob + 4 * t], delta[ob][t], din[ib]);
}
template <int NB>
__device__ __forceinline__ void g64m_lrelu(f64x4 (&v)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[b][r] = v[b][r] > 0.0 ? v[b][r] : 0.01 * v[b][r];
}
// delta *= leakyrelu'(z), from the sign of the post-activation value (leakyrelu keeps the sign)
template <int NB>
__device__ __forceinline__ void g64m_lrelu_grad(f64x4 (&d)[NB], const f64x4 (&act)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) d[b][r] *= act[b][r] > 0.0 ? 1.0 : 0.01;
}
// a wave's C-layout blocks -> its [sample][feature] tile (row stride ST): the operand layout of the dW GEMM
template <int NB, int ST>
__device__ __forceinline__ void g64m_to_tile(double *__restrict__ tile, const f64x4 (&v)[NB], int c16, int q) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[c16 * ST + 16 * b + q + 4 * r] = v[b][r];
}

// One layer's weight gradient, cooperatively: the layer has IB x OB blocks of 16 x 16; wave `wave` owns the blocks e = wave,
// wave + 4, ... (e = ib * OB + ob) and contracts each over the 64 samples of the group's four tiles.  atiles / dtiles: the four
// waves' input / cotangent tiles.  bsum: the bias gradient's partial sums ride on the blocks with ib == 0.
template <int IB, int OB, int ST>
struct G64MAcc {
  static constexpr int NS = (IB * OB + 3) / 4;
  f64x4 w[NS];
  double b[NS];
};
template <int IB, int OB, int ST>
__device__ __forceinline__ void g64m_zero(G64MAcc<IB, OB, ST> &a) {
#pragma unroll
  for (int s = 0; s < G64MAcc<IB, OB, ST>::NS; ++s) {
    a.w[s] = f64x4{0.0, 0.0, 0.0, 0.0};
    a.b[s] = 0.0;
  }
}
template <int IB, int OB, int ST>
__device__ __forceinline__ void g64m_dw(const double *__restrict__ atiles, const double *__restrict__ dtiles, G64MAcc<IB, OB, ST> &acc,
                                        int wave, int c16, int q) {
  constexpr int TILE = 16 * ST;
#pragma unroll
  for (int s = 0; s < G64MAcc<IB, OB, ST>::NS; ++s) {
    const int e = wave + 4 * s;
    if (e < IB * OB) {  // (wave-uniform)
      const int ib = e / OB, ob = e - ib * OB;
      const double *pa = atiles + q * ST + 16 * ib + c16;
      const double *pd = dtiles + q * ST + 16 * ob + c16;
      f64x4 w = acc.w[s];
      double bs = 0.0;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double av = pa[tt * TILE + 4 * t * ST], dv = pd[tt * TILE + 4 * t * ST];
          w = g64m_mfma(av, dv, w);
          bs += dv;
        }
      acc.w[s] = w;
      acc.b[s] += ib == 0 ? bs : 0.0;
    }
  }
}
// a wave's share of a layer's gradient -> the workgroup's slab (theta order, `slab` already offset by -slab_off)
template <int IB, int OB, int ST>
__device__ __forceinline__ void g64m_store(double *__restrict__ slab, const G64MAcc<IB, OB, ST> &acc, long w_off, long b_off, int nin,
                                           int nout, int wave, int c16, int q) {
#pragma unroll
  for (int s = 0; s < G64MAcc<IB, OB, ST>::NS; ++s) {
    const int e = wave + 4 * s;
    if (e < IB * OB) {
      const int ib = e / OB, ob = e - ib * OB;
      const int o = 16 * ob + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ib + q + 4 * r;
        if (i < nin && o < nout) slab[w_off + (long)i * nout + o] = acc.w[s][r];
      }
      if (ib == 0) {
        double v = acc.b[s];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (q == 0 && o < nout) slab[b_off + o] = v;
      }
    }
  }
}

// ... ADDED into the slab at a column window of the layer (the spline couplings' output layer taken in passes: a pass's block
// of accumulators is summed over the group's 64 samples, added to the workgroup's slab -- which the kernel zeroed -- and reused)
template <int IB, int OB, int ST>
__device__ __forceinline__ void g64m_store_add(double *__restrict__ slab, const G64MAcc<IB, OB, ST> &acc, long w_off, long b_off, int nin,
                                               int nout, int c0, int ncols, int wave, int c16, int q) {
#pragma unroll
  for (int s = 0; s < G64MAcc<IB, OB, ST>::NS; ++s) {
    const int e = wave + 4 * s;
    if (e < IB * OB) {
      const int ib = e / OB, ob = e - ib * OB;
      const int o = 16 * ob + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ib + q + 4 * r;
        if (i < nin && o < ncols) slab[w_off + (long)i * nout + c0 + o] += acc.w[s][r];
      }
      if (ib == 0) {
        double v = acc.b[s];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (q == 0 && o < ncols) slab[b_off + c0 + o] += v;
      }
    }
  }
}

// the conditioner half of the standard-layout state into C-layout blocks (zero beyond m / N)
template <int MB>
__device__ __forceinline__ void g64m_load_cond(const double *__restrict__ row, int m, int par_c, bool valid, f64x4 (&xb)[MB], int q) {
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * b + q + 4 * r;
      xb[b][r] = (valid && f < m) ? row[2 * f + par_c] : 0.0;
    }
}

// net(x2) of one tile: hidden layers with leaky-ReLU; a1 / a2 kept for the reverse pass (a2 unused with one hidden layer)
template <class G>
__device__ __forceinline__ void g64m_net_fwd(const double *__restrict__ img, int nhid, const f64x4 (&xb)[G::MB], f64x4 (&a1)[G::HB],
                                             f64x4 (&a2)[G::HB], f64x4 (&out)[G::CB], int c16, int q) {
  g64m_fwd<G::MB, G::HB, G::SH>(img + G::W0, img + G::B0, xb, a1, c16, q);
  g64m_lrelu<G::HB>(a1);
  if (nhid == 2) {
    g64m_fwd<G::HB, G::HB, G::SH>(img + G::W1, img + G::B1, a1, a2, c16, q);
    g64m_lrelu<G::HB>(a2);
    g64m_fwd<G::HB, G::CB, G::SC>(img + G::img_wo(2), img + G::img_bo(2), a2, out, c16, q);
  } else {
    g64m_fwd<G::HB, G::CB, G::SC>(img + G::img_wo(1), img + G::img_bo(1), a1, out, c16, q);
  }
}

// the hidden layers only (the output layer is the caller's: spline couplings in passes)
template <class G>
__device__ __forceinline__ void g64m_hidden_fwd(const double *__restrict__ img, int nhid, const f64x4 (&xb)[G::MB], f64x4 (&a1)[G::HB],
                                                f64x4 (&a2)[G::HB], int c16, int q) {
  g64m_fwd<G::MB, G::HB, G::SH>(img + G::W0, img + G::B0, xb, a1, c16, q);
  g64m_lrelu<G::HB>(a1);
  if (nhid == 2) {
    g64m_fwd<G::HB, G::HB, G::SH>(img + G::W1, img + G::B1, a1, a2, c16, q);
    g64m_lrelu<G::HB>(a2);
  }
}

// ---- one coupling forward / inverse: y1 = x1 exp(s) + t (src/flows/realnvp.jl:57-63), x1 = (y1 - t) exp(-s) (:86-110) -------
// Two passes over the workgroup's tiles, one net in LDS each: forward  s-net: y1 <- x1 exp(s), ladj += sum s;  t-net: y1 += t
//                                                              inverse  t-net: y1 <- x1 - t;  s-net: y1 <- y1 exp(-s), ladj -= sum s
// (both nets read only the conditioner half, which a coupling does not change).
template <class G>
__global__ __launch_bounds__(256) void k_g64m_apply(G64Args a, int inverse, const double *__restrict__ theta, const double *x, double *y,
                                                    double *__restrict__ ladj, long long *trace) {
#ifdef NF_KERNEL_TRACE  // tools/trace_g64m.py: workgroup 0 / wave 0; [16 pass + 0] start, [+1] net staged, [+2 + 4 i ..] tile i: x loaded, net done, element-wise done
#define G64M_STAMP(slot) do { if (tr) { __builtin_amdgcn_sched_barrier(0); tr[slot] = clock64(); } } while (0)
  long long *tr = (trace && blockIdx.x == 0 && threadIdx.x == 0) ? trace : nullptr;
#else
#define G64M_STAMP(slot) do { } while (0)
#endif
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, q = lane >> 4;
  const int nhid = a.net[0].nl - 1, par_c = 1 - a.par_t;
  const long ntiles = (a.N + 15) / 16;
  for (int pass = 0; pass < 2; ++pass) {
    const bool is_s = inverse ? pass == 1 : pass == 0;
    __syncthreads();
    G64M_STAMP(16 * pass + 0);
    g64m_stage_net<G>(lds64, theta, a.net[is_s ? 0 : 1], tid);
    __syncthreads();
    G64M_STAMP(16 * pass + 1);
    int tcount = 0;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
      const int tb = 16 * pass + 2 + 4 * (tcount < 3 ? tcount : 2);
      ++tcount;
      (void)tb;
      const long j = tile * 16 + c16;
      const bool valid = j < a.N;
      const long jr = valid ? j : a.N - 1;
      const double *xr = x + jr * a.d;
      double *yr = y + jr * a.d;
      f64x4 xb[G::MB], a1[G::HB], a2[G::HB], out[G::CB];
      G64M_STAMP(tb + 0);
      g64m_load_cond<G::MB>(xr, a.m, par_c, valid, xb, q);
      g64m_net_fwd<G>(lds64, nhid, xb, a1, a2, out, c16, q);
      G64M_STAMP(tb + 1);
      double lsum = 0.0;
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = 16 * b + q + 4 * r;
          if (valid && p < a.c) {
            const int idx = 2 * p + a.par_t;
            if (is_s) {
              const double s = tanh(out[b][r]);
              const double v = pass == 0 ? xr[idx] : yr[idx];
              yr[idx] = inverse ? v * exp(-s) : v * exp(s);
              lsum += inverse ? -s : s;
            } else {
              const double v = pass == 0 ? xr[idx] : yr[idx];
              yr[idx] = inverse ? v - out[b][r] : v + out[b][r];
            }
          }
        }
      if (is_s) {
        lsum += __shfl_xor(lsum, 16);
        lsum += __shfl_xor(lsum, 32);
        if (q == 0 && valid) ladj[j] += lsum;
      }
      if (pass == 0 && y != x) {
#pragma unroll
        for (int b = 0; b < G::MB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = 16 * b + q + 4 * r;
            if (valid && f < a.m) yr[2 * f + par_c] = xb[b][r];
          }
      }
      G64M_STAMP(tb + 2);
    }
    G64M_STAMP(16 * pass + 14);
  }
}

// ---- reverse pass of one coupling at its INPUT x (gbar: ybar -> xbar), or of the INVERSE coupling at its output (inv != 0):
// the semantics of k_g64_bwd, nf_generic64.hip.  slabs: [gridDim.x][Pc], THIS coupling's parameters from theta index slab_off.
template <class G, bool NSF = false>
__global__ __launch_bounds__(256) void k_g64m_bwd(G64Args a, int inv, const double *__restrict__ theta, const double *__restrict__ x,
                                                  double *gbar, const double *__restrict__ lbar, double lbar_const,
                                                  double *__restrict__ slabs, long Pc, long slab_off, long long *trace) {
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c16 = lane & 15, q = lane >> 4;
#ifdef NF_KERNEL_TRACE  // tools/trace_g64m.py: [32 + 24 phase + ..]: 0 start, 1 staged, 2 + 5 g .. group g (forward + element-wise | output layer |
                        // hidden layer | first layer + stores), 22 groups done, 23 stored
  long long *tr = (trace && blockIdx.x == 0 && threadIdx.x == 0) ? trace + 32 : nullptr;
#endif
  const int nhid = a.net[0].nl - 1, par_c = 1 - a.par_t;
  double *img = lds64;
  double *atiles = lds64 + G::img_size(nhid), *dtiles = atiles + 4 * G::TILE;  // [4 waves][16 samples][ST]
  double *mya = atiles + wave * G::TILE, *myd = dtiles + wave * G::TILE;
  double *slab = slabs + (long)blockIdx.x * Pc - slab_off;
  const long ntiles = (a.N + 15) / 16, ngroups = (ntiles + 3) / 4;
  for (int phase = 0; phase < (NSF ? 1 : 2); ++phase) {  // a spline coupling has ONE net
    const bool is_s = inv ? phase == 0 : phase == 1;  // forward coupling: t-net first (it reads ybar1 before the s phase rescales it)
    const G64Net &net = a.net[NSF ? 0 : (is_s ? 0 : 1)];
    __syncthreads();
    G64M_STAMP(24 * phase + 0);
    // spline couplings: P raw parameters per transformed dimension, DP whole dimensions per pass of the output layer
    const int P = NSF ? 3 * a.K - 1 : 1, DP = NSF ? g64m_nsf_dp<G>(P) : 1, NP = NSF ? (a.c + DP - 1) / DP : 1;
    g64m_stage_net<G>(img, theta, net, tid, !(NSF && NP > 1));
    if (NSF && NP > 1) {  // the passes ADD their weight-gradient blocks into the slab
      const long nwo = (long)net.dims[nhid] * net.dims[nhid + 1];
      for (long i = tid; i < nwo; i += 256) slab[net.w[nhid] + i] = 0.0;
      for (int i = tid; i < net.dims[nhid + 1]; i += 256) slab[net.b[nhid] + i] = 0.0;
    }
    __syncthreads();
    G64M_STAMP(24 * phase + 1);
    int gcount = 0;
    G64MAcc<G::MB, G::HB, G::ST> acc0;
    G64MAcc<G::HB, G::HB, G::ST> acc1;
    G64MAcc<G::HB, G::CB, G::ST> acco;
    g64m_zero(acc0);
    g64m_zero(acc1);
    g64m_zero(acco);
    for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
      const int gb = 24 * phase + 2 + 5 * (gcount < 4 ? gcount : 3);
      ++gcount;
      (void)gb;
      G64M_STAMP(gb + 0);
      const long tile = grp * 4 + wave;
      const long j = tile * 16 + c16;
      const bool valid = j < a.N;
      const long jr = valid ? j : a.N - 1;
      const double *xr = x + jr * a.d;
      double *gr = gbar + jr * a.d;
      const double lb = valid ? (lbar ? lbar[jr] : lbar_const) : 0.0;
      f64x4 xb[G::MB], a1[G::HB], a2[G::HB], dout[G::CB];
      f64x4 dh[G::HB];
      g64m_load_cond<G::MB>(xr, a.m, par_c, valid, xb, q);
      if constexpr (NSF) {
        // The net's outputs are the splines' raw parameters, (3K - 1) per transformed dim, consecutive per dim: through the
        // wave's [sample][feature] tile (the dW stage's operand layout) a lane gets the ones of its (sample, dim) as one
        // contiguous row -- lane group q takes dims q, q + 4, ... -- and the scalar spline code of the general kernels
        // (g64_build / g64_spline_bwd, nf_generic64.hip) runs on them unchanged; its parameter cotangents go back the same way.
        // Round 6: the output layer in NP passes of DP whole dimensions (<= 16 CB columns each): its slice of the weights is
        // staged per (group, pass) into the image's output-layer block, the input cotangent accumulates over the passes, the
        // pass's weight-gradient blocks are added to the slab.  One pass (NP = 1): round 5's kernel, resident accumulators.
        g64m_hidden_fwd<G>(img, nhid, xb, a1, a2, c16, q);
        const f64x4 (&hid)[G::HB] = nhid == 2 ? a2 : a1;
        g64m_to_tile<G::HB, G::ST>(mya, hid, c16, q);  // the A operand of every pass's weight-gradient GEMM
#pragma unroll
        for (int b = 0; b < G::HB; ++b) dh[b] = f64x4{0.0, 0.0, 0.0, 0.0};
        const double *wo = img + G::img_wo(nhid == 2 ? 2 : 1), *bo = img + G::img_bo(nhid == 2 ? 2 : 1);
#pragma unroll 1
        for (int ps = 0; ps < NP; ++ps) {
          const int dim0 = ps * DP, nd = a.c - dim0 < DP ? a.c - dim0 : DP, c0 = dim0 * P, ncols = nd * P;
          if (NP > 1) {
            __syncthreads();  // every wave is done with the previous slice (and the previous pass's cotangent tiles)
            g64m_stage_out_cols<G>(img, theta, net, c0, ncols, tid);
            __syncthreads();
          }
          g64m_fwd<G::HB, G::CB, G::SC>(wo, bo, hid, dout, c16, q);
          g64m_to_tile<G::CB, G::ST>(myd, dout, c16, q);
          wave_lds_fence();
          for (int pl = q; pl < nd; pl += 4) {
            double *raw = myd + c16 * G::ST + pl * P;
            double thb[3 * G64_MAXK];
            for (int i = 0; i < P; ++i) thb[i] = 0.0;
            if (valid) {
              G64Spline<double> sp;
              g64_build<double>(raw, a.K, a.B, sp);
              const int idx = 2 * (dim0 + pl) + a.par_t;
              gr[idx] = g64_spline_bwd<double>(sp, raw, a.K, a.B, xr[idx], gr[idx], lb, thb, inv != 0);
            }
            for (int i = 0; i < P; ++i) raw[i] = thb[i];
          }
          wave_lds_fence();
#pragma unroll
          for (int b = 0; b < G::CB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) dout[b][r] = myd[c16 * G::ST + 16 * b + q + 4 * r];  // (columns past the slice: zero weights, zero bias: 0)
          g64m_bwdx<G::HB, G::CB, G::SC, true>(wo, dout, dh, c16, q);
          __syncthreads();
          if (NP == 1) {
            g64m_dw<G::HB, G::CB, G::ST>(atiles, dtiles, acco, wave, c16, q);
          } else {
            G64MAcc<G::HB, G::CB, G::ST> accp;
            g64m_zero(accp);
            g64m_dw<G::HB, G::CB, G::ST>(atiles, dtiles, accp, wave, c16, q);
            g64m_store_add<G::HB, G::CB, G::ST>(slab, accp, net.w[nhid], net.b[nhid], net.dims[nhid], net.dims[nhid + 1], c0, ncols, wave, c16, q);
          }
          __syncthreads();
        }
      } else {
      g64m_net_fwd<G>(img, nhid, xb, a1, a2, dout, c16, q);
      // element-wise stage -> cotangent of the net's output (rows beyond c and samples beyond N: 0)
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = 16 * b + q + 4 * r;
          const bool ok = valid && p < a.c;
          const int idx = 2 * p + a.par_t;
          double dv = 0.0;
          if (ok) {
            if (is_s) {
              const double s = tanh(dout[b][r]), x1 = xr[idx], gb = gr[idx];
              if (inv) {  // x1 = (v1 - t) exp(-s), ladj_inv = -sum s:  v1bar = x1bar exp(-s), sbar = -x1bar x1 - lbar
                gr[idx] = gb * exp(-s);
                dv = (-gb * x1 - lb) * (1.0 - s * s);
              } else {    // y1 = x1 exp(s) + t:  x1bar = ybar1 exp(s), sbar = ybar1 x1 exp(s) + lbar
                const double es = exp(s);
                gr[idx] = gb * es;
                dv = (gb * x1 * es + lb) * (1.0 - s * s);
              }
            } else {
              dv = inv ? -gr[idx] : gr[idx];  // tbar = ybar1 (forward); -v1bar (inverse: after the s phase rescaled it)
            }
          }
          dout[b][r] = dv;
        }
      }
      G64M_STAMP(gb + 1);
      // ---- output layer (spline couplings: done above, pass by pass)
      if constexpr (!NSF) {
        if (nhid == 2) {
          g64m_to_tile<G::HB, G::ST>(mya, a2, c16, q);
          g64m_bwdx<G::HB, G::CB, G::SC, false>(img + G::img_wo(2), dout, dh, c16, q);
        } else {
          g64m_to_tile<G::HB, G::ST>(mya, a1, c16, q);
          g64m_bwdx<G::HB, G::CB, G::SC, false>(img + G::img_wo(1), dout, dh, c16, q);
        }
        g64m_to_tile<G::CB, G::ST>(myd, dout, c16, q);
        __syncthreads();
        g64m_dw<G::HB, G::CB, G::ST>(atiles, dtiles, acco, wave, c16, q);
        __syncthreads();
      }
      G64M_STAMP(gb + 2);
      if (nhid == 2) {
        g64m_lrelu_grad<G::HB>(dh, a2);
        f64x4 d1[G::HB];
        g64m_to_tile<G::HB, G::ST>(mya, a1, c16, q);
        g64m_to_tile<G::HB, G::ST>(myd, dh, c16, q);
        g64m_bwdx<G::HB, G::HB, G::SH, false>(img + G::W1, dh, d1, c16, q);
        __syncthreads();
        g64m_dw<G::HB, G::HB, G::ST>(atiles, dtiles, acc1, wave, c16, q);
        __syncthreads();
#pragma unroll
        for (int b = 0; b < G::HB; ++b) dh[b] = d1[b];
      }
      G64M_STAMP(gb + 3);
      g64m_lrelu_grad<G::HB>(dh, a1);
      // ---- first layer: x2bar += W0 delta
      f64x4 g2[G::MB];
      g64m_to_tile<G::MB, G::ST>(mya, xb, c16, q);
      g64m_to_tile<G::HB, G::ST>(myd, dh, c16, q);
      g64m_bwdx<G::MB, G::HB, G::SH, false>(img + G::W0, dh, g2, c16, q);
      __syncthreads();
      g64m_dw<G::MB, G::HB, G::ST>(atiles, dtiles, acc0, wave, c16, q);
      __syncthreads();
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * b + q + 4 * r;
          if (valid && f < a.m) gr[2 * f + par_c] += g2[b][r];
        }
      G64M_STAMP(gb + 4);
    }
    G64M_STAMP(24 * phase + 22);
    g64m_store<G::MB, G::HB, G::ST>(slab, acc0, net.w[0], net.b[0], net.dims[0], net.dims[1], wave, c16, q);
    if (nhid == 2) g64m_store<G::HB, G::HB, G::ST>(slab, acc1, net.w[1], net.b[1], net.dims[1], net.dims[2], wave, c16, q);
    if (!(NSF && NP > 1))  // (in passes: already in the slab)
      g64m_store<G::HB, G::CB, G::ST>(slab, acco, net.w[nhid], net.b[nhid], net.dims[nhid], net.dims[nhid + 1], wave, c16, q);
    G64M_STAMP(24 * phase + 23);
  }
}

// ---- neural spline coupling, forward or inverse: ONE net; its outputs reach the scalar spline code through the wave's tile ----
template <class G>
__global__ __launch_bounds__(256) void k_g64m_nsf_apply(G64Args a, int inverse, const double *__restrict__ theta, const double *x, double *y,
                                                        double *__restrict__ ladj) {
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, q = lane >> 4;
  const int nhid = a.net[0].nl - 1, par_c = 1 - a.par_t, P = 3 * a.K - 1;
  // the output layer in NP passes of DP whole dimensions (round 6; one pass: the whole layer stays staged, as in round 5)
  const int DP = g64m_nsf_dp<G>(P), NP = (a.c + DP - 1) / DP;
  const long ntiles = (a.N + 15) / 16, ngroups = (ntiles + 3) / 4;
  double *mytile = lds64 + G::img_size(nhid) + wave * G::TILE;
  const double *wo = lds64 + G::img_wo(nhid == 2 ? 2 : 1), *bo = lds64 + G::img_bo(nhid == 2 ? 2 : 1);
  g64m_stage_net<G>(lds64, theta, a.net[0], tid, NP == 1);
  __syncthreads();
  // (groups of four tiles, the four waves in step: the passes' staging needs workgroup barriers inside the loop)
  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long tile = grp * 4 + wave;
    const long j = tile * 16 + c16;
    const bool valid = j < a.N;
    const long jr = valid ? j : a.N - 1;
    const double *xr = x + jr * a.d;
    double *yr = y + jr * a.d;
    f64x4 xb[G::MB], a1[G::HB], a2[G::HB], out[G::CB];
    g64m_load_cond<G::MB>(xr, a.m, par_c, valid, xb, q);
    g64m_hidden_fwd<G>(lds64, nhid, xb, a1, a2, c16, q);
    const f64x4 (&hid)[G::HB] = nhid == 2 ? a2 : a1;
    double lsum = 0.0;
#pragma unroll 1
    for (int ps = 0; ps < NP; ++ps) {
      const int dim0 = ps * DP, nd = a.c - dim0 < DP ? a.c - dim0 : DP;
      if (NP > 1) {
        __syncthreads();  // every wave is done with the previous slice
        g64m_stage_out_cols<G>(lds64, theta, a.net[0], dim0 * P, nd * P, tid);
        __syncthreads();
      }
      g64m_fwd<G::HB, G::CB, G::SC>(wo, bo, hid, out, c16, q);
      g64m_to_tile<G::CB, G::ST>(mytile, out, c16, q);
      wave_lds_fence();
      if (valid)
        for (int pl = q; pl < nd; pl += 4) {
          G64Spline<double> sp;
          g64_build<double>(mytile + c16 * G::ST + pl * P, a.K, a.B, sp);
          const int idx = 2 * (dim0 + pl) + a.par_t;
          const double v = xr[idx];
          yr[idx] = inverse ? g64_spline_inv(sp, a.K, v, lsum) : g64_spline_fwd(sp, a.K, v, lsum);
        }
      wave_lds_fence();  // the next pass's (or tile's) outputs overwrite the tile
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    if (q == 0 && valid) ladj[j] += lsum;
    if (y != x) {
#pragma unroll
      for (int b = 0; b < G::MB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * b + q + 4 * r;
          if (valid && f < a.m) yr[2 * f + par_c] = xb[b][r];
        }
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
// 0: not a shape of these kernels; otherwise 10 * MB + HB (blocks of 16)
static int g64m_geo(const nf_flow_desc *desc) {
  static const bool off = std::getenv("NF_G64_NO_F64_MFMA") != nullptr;  // A/B: the scalar Float64 kernels
  if (off || desc->kind != NF_KIND_REALNVP || desc->dtype != NF_DTYPE_F64) return 0;
  if (desc->n_hidden < 1 || desc->n_hidden > 2 || desc->d < 2 || desc->d > 64) return 0;
  int hmax = 0;
  for (int i = 0; i < desc->n_hidden; ++i) hmax = desc->hdims[i] > hmax ? desc->hdims[i] : hmax;
  if (hmax < 1 || hmax > 64) return 0;
  const int cmax = (desc->d + 1) / 2;
  return 10 * (cmax <= 16 ? 1 : 2) + (hmax <= 32 ? 2 : 4);
}
using G64M12 = G64M<1, 2>;
using G64M14 = G64M<1, 4>;
using G64M22 = G64M<2, 2>;
using G64M24 = G64M<2, 4>;
#define G64M_DISPATCH(ID, CALL) ((ID) == 12 ? CALL(G64M12) : (ID) == 14 ? CALL(G64M14) : (ID) == 22 ? CALL(G64M22) : CALL(G64M24))
// Float64 neural spline couplings: conditioner <= 16 inputs (d <= 32), hidden <= 32 -- the reference's test shape
// nsf(q0; paramtype = Float64) at d = 5, K = 10, hidden [32, 32] (test/flow.jl:65-78) in ONE pass of the output layer
// ((3K - 1) ceil(d / 2) <= 96 net outputs: round 5), and since round 6 every d <= 32 in passes of 96 / (3K - 1) whole
// dimensions (the reference's DEFAULT constructor nsf(q0) = [32, 32], K = 10, B = 30, Float64, src/flows/neuralspline.jl:232-234:
// 464 outputs at d = 32, six passes of three dimensions).
using G64MN = G64M<1, 2, 6>;
static bool g64m_nsf_ok(const nf_flow_desc *desc) {
  static const bool off = std::getenv("NF_G64_NO_F64_MFMA") != nullptr;
  if (off || desc->kind != NF_KIND_NSF || desc->dtype != NF_DTYPE_F64) return false;
  if (desc->n_hidden < 1 || desc->n_hidden > 2 || desc->d < 2 || desc->K < 1 || desc->K > G64_MAXK) return false;
  for (int i = 0; i < desc->n_hidden; ++i)
    if (desc->hdims[i] < 1 || desc->hdims[i] > 32) return false;
  const int cmax = (desc->d + 1) / 2;
  // (round 6: any number of transformed dimensions up to 16 -- the output layer is taken in passes of whole dimensions when its
  // (3K - 1) c columns exceed the image's 16 CB: d <= 32 at every K <= 16, the reference's default nsf(q0) included)
  return cmax <= 16 && (3 * desc->K - 1) <= 16 * G64MN::CB;
}

template <class G>
static int g64m_launch_apply(nf_ctx *ctx, const G64Args &a, int inverse, const double *theta, const double *x, double *y, double *ladj) {
  const size_t lds = G::apply_lds(2);
  static AttrOnce attr_once;
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_g64m_apply<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  // as many workgroups as the device holds at once (a workgroup stages both nets' images: a second round of workgroups pays
  // that again -- at d = 64 one workgroup fits a CU, and 2 num_cu of them ran as two rounds)
  const long ngroups = ((a.N + 15) / 16 + 3) / 4;
  const long per_cu = lds > 80 * 1024 ? 1 : 2;  // 160 KB of LDS per CU
  long grid = ngroups < per_cu * ctx->num_cu ? ngroups : per_cu * ctx->num_cu;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_g64m_apply<G>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, inverse, theta, x, y, ladj, (long long *)ctx->trace);
  return (int)hipGetLastError();
}
static int g64m_nsf_launch_apply(nf_ctx *ctx, const G64Args &a, int inverse, const double *theta, const double *x, double *y, double *ladj) {
  using G = G64MN;
  const size_t lds = (size_t)(G::img_size(2) + 4 * G::TILE) * 8;
  static AttrOnce attr_once;
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_g64m_nsf_apply<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  const long ngroups = ((a.N + 15) / 16 + 3) / 4;
  long grid = ngroups < (long)ctx->num_cu ? ngroups : (long)ctx->num_cu;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_g64m_nsf_apply<G>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, a, inverse, theta, x, y, ladj);
  return (int)hipGetLastError();
}
static int g64m_nsf_launch_bwd(nf_ctx *ctx, const G64Args &a, int inv, const double *theta, const double *x, double *gbar, const double *lbar,
                               double lbar_const, double *slabs, long Pc, long slab_off, unsigned grid) {
  using G = G64MN;
  const size_t lds = G::bwd_lds(2);
  static AttrOnce attr_once;
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_g64m_bwd<G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  hipLaunchKernelGGL((k_g64m_bwd<G, true>), dim3(grid), dim3(256), lds, ctx->stream, a, inv, theta, x, gbar, lbar, lbar_const, slabs, Pc, slab_off,
                     (long long *)ctx->trace);
  return (int)hipGetLastError();
}
template <class G>
static int g64m_launch_bwd(nf_ctx *ctx, const G64Args &a, int inv, const double *theta, const double *x, double *gbar, const double *lbar,
                           double lbar_const, double *slabs, long Pc, long slab_off, unsigned grid) {
  const size_t lds = G::bwd_lds(2);
  static AttrOnce attr_once;
  NF_TRY(attr_once.run(ctx->device, [&]() -> int {
    NF_HIP(hipFuncSetAttribute((const void *)k_g64m_bwd<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return NF_OK;
  }));
  hipLaunchKernelGGL((k_g64m_bwd<G>), dim3(grid), dim3(256), lds, ctx->stream, a, inv, theta, x, gbar, lbar, lbar_const, slabs, Pc, slab_off,
                     (long long *)ctx->trace);
  return (int)hipGetLastError();
}
