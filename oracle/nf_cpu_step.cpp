// nf_cpu_step.cpp -- C++ / OpenMP restatement of the reference's RealNVP reverse-KL training step, fp32.
//
// TEST / MEASUREMENT INFRASTRUCTURE ONLY (same rule as nf_oracle.py): built and called by tests/ (pinned against the
// numpy oracle) and by bench.py's `cpu_baseline` leg (form 2: "the C++/OpenMP host
// implementation").  Nothing in the product path links or loads it; the product has no CPU path.
//
// What it computes is the reference's step, with the same parameter vector (Optimisers.destructure order):
//   xs ~ N(0, I)                                              src/objectives/elbo.jl:94
//   per coupling (last-listed first):                          src/flows/realnvp.jl:77-83, src/flows/utils.jl:71-100
//     x1, x2 = partition(mask, x); s = tanh(fnn_s(x2)); t = fnn_t(x2); y1 = x1 .* exp.(s) .+ t; ladj += sum(s)
//   elbo = mean(logp(ys) - logpdf(q0, xs) + ladj)              src/objectives/elbo.jl:65-70,96
//   g = d(-elbo)/dtheta (hand-derived reverse pass); Adam; norm(g)      src/optimize.jl:86,89,99
// How: one OpenMP thread per tile of TS samples; a tile runs through all couplings with its activations kept in a
// per-thread scratch (cache-resident), every Dense layer as loops the compiler vectorises over the `out` index (the
// contiguous one in Flux's out x in column-major weights); per-thread gradient buffers, reduced in a fixed order.
//
// Build: g++ -O3 -march=native -fno-math-errno -fopenmp -shared -fPIC (bench.py / tests do it at run time, on the box
// whose cores are being timed).  No -ffast-math: a library built with it switches the whole process to flush-to-zero.
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int TS = 64;  // samples per tile

struct Net {
  long w1, b1, w2, b2, w3, b3;  // offsets into theta
  int m, h1, h2, c;
};
struct Coupling {
  Net s, t;
  int par_t, c, m;  // transformed features are 2p + par_t
};

inline float lrelu(float z) { return z > 0.f ? z : 0.01f * z; }

// Register-blocked small GEMMs.  Outputs are produced 64 at a time into a local accumulator array of constant size,
// which the compiler keeps in vector registers (4 zmm / 8 ymm) across the contraction loop.
constexpr int KB = 64;

// out[j][o] = act(b[o] + sum_i in[j][i] W[i][o]);  W is [nin][nout] (= Flux weight out x in, column-major)
inline void dense_fwd(const float *__restrict__ W, const float *__restrict__ b, const float *__restrict__ in, int nin, int nout,
                      float *__restrict__ out, int ld_in, int ld_out, bool relu) {
  for (int k0 = 0; k0 < nout; k0 += KB) {
    const int kn = std::min(KB, nout - k0);
    for (int j = 0; j < TS; ++j) {
      float acc[KB];
      for (int k = 0; k < KB; ++k) acc[k] = k < kn ? b[k0 + k] : 0.f;
      const float *x = in + (long)j * ld_in;
      if (kn == KB) {
        for (int i = 0; i < nin; ++i) {
          const float a = x[i];
          const float *w = W + (long)i * nout + k0;
#pragma omp simd
          for (int k = 0; k < KB; ++k) acc[k] += a * w[k];
        }
      } else {
        for (int i = 0; i < nin; ++i) {
          const float a = x[i];
          const float *w = W + (long)i * nout + k0;
          for (int k = 0; k < kn; ++k) acc[k] += a * w[k];
        }
      }
      float *o = out + (long)j * ld_out + k0;
      for (int k = 0; k < kn; ++k) o[k] = relu ? lrelu(acc[k]) : acc[k];
    }
  }
}

// reverse pass of one Dense layer: delta [TS][nout] (cotangent of the PRE-activation), in [TS][nin];
//   gW[i][o] += sum_j in[j][i] delta[j][o];  gb[o] += sum_j delta[j][o];  din[j][i] = sum_o WT[o][i] delta[j][o]
// WT is the transposed copy [nout][nin] (made once per step), so that din is accumulated along a contiguous index too.
inline void dense_bwd(const float *__restrict__ WT, const float *__restrict__ in, const float *__restrict__ delta, int nin,
                      int nout, int ld_in, int ld_d, float *__restrict__ gW, float *__restrict__ gb, float *__restrict__ din,
                      int ld_din) {
  // weight gradient: for each input i, 64 outputs at a time, the sum over the tile's samples stays in registers
  for (int k0 = 0; k0 < nout; k0 += KB) {
    const int kn = std::min(KB, nout - k0);
    {
      float acc[KB];
      for (int k = 0; k < KB; ++k) acc[k] = 0.f;
      for (int j = 0; j < TS; ++j) {
        const float *d = delta + (long)j * ld_d + k0;
        for (int k = 0; k < kn; ++k) acc[k] += d[k];
      }
      for (int k = 0; k < kn; ++k) gb[k0 + k] += acc[k];
    }
    for (int i = 0; i < nin; ++i) {
      float acc[KB];
      for (int k = 0; k < KB; ++k) acc[k] = 0.f;
      if (kn == KB) {
        for (int j = 0; j < TS; ++j) {
          const float a = in[(long)j * ld_in + i];
          const float *d = delta + (long)j * ld_d + k0;
#pragma omp simd
          for (int k = 0; k < KB; ++k) acc[k] += a * d[k];
        }
      } else {
        for (int j = 0; j < TS; ++j) {
          const float a = in[(long)j * ld_in + i];
          const float *d = delta + (long)j * ld_d + k0;
          for (int k = 0; k < kn; ++k) acc[k] += a * d[k];
        }
      }
      float *g = gW + (long)i * nout + k0;
      for (int k = 0; k < kn; ++k) g[k] += acc[k];
    }
  }
  if (!din) return;
  for (int i0 = 0; i0 < nin; i0 += KB) {
    const int in_n = std::min(KB, nin - i0);
    for (int j = 0; j < TS; ++j) {
      float acc[KB];
      for (int k = 0; k < KB; ++k) acc[k] = 0.f;
      const float *d = delta + (long)j * ld_d;
      if (in_n == KB) {
        for (int o = 0; o < nout; ++o) {
          const float a = d[o];
          const float *w = WT + (long)o * nin + i0;
#pragma omp simd
          for (int k = 0; k < KB; ++k) acc[k] += a * w[k];
        }
      } else {
        for (int o = 0; o < nout; ++o) {
          const float a = d[o];
          const float *w = WT + (long)o * nin + i0;
          for (int k = 0; k < in_n; ++k) acc[k] += a * w[k];
        }
      }
      float *x = din + (long)j * ld_din + i0;
      for (int k = 0; k < in_n; ++k) x[k] = acc[k];
    }
  }
}

struct Rng {  // xoshiro256++ + Box-Muller: the role of Julia's randn in the reference's step
  uint64_t s[4];
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  explicit Rng(uint64_t seed) {
    uint64_t z = seed;
    for (auto &v : s) {  // splitmix64
      z += 0x9E3779B97F4A7C15ull;
      uint64_t r = z;
      r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull;
      r = (r ^ (r >> 27)) * 0x94D049BB133111EBull;
      v = r ^ (r >> 31);
    }
  }
  uint64_t next() {
    const uint64_t r = rotl(s[0] + s[3], 23) + s[0], t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  void normals2(float &a, float &b) {
    const uint64_t r = next();
    const float u0 = ((float)(uint32_t)(r >> 40) + 0.5f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(uint32_t)((r >> 8) & 0xFFFFFF) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = std::sqrt(-2.0f * std::log(u0)), ang = 6.283185307179586f * u1;
    a = rad * std::cos(ang);
    b = rad * std::sin(ang);
  }
};

Net make_net(long &off, int m, int h1, int h2, int c) {
  Net n;
  n.m = m; n.h1 = h1; n.h2 = h2; n.c = c;
  n.w1 = off; off += (long)m * h1;
  n.b1 = off; off += h1;
  n.w2 = off; off += (long)h1 * h2;
  n.b2 = off; off += h2;
  n.w3 = off; off += (long)h2 * c;
  n.b3 = off; off += c;
  return n;
}

}  // namespace

extern "C" {

long nfcpu_param_count(int d, int h1, int h2, int nlayers) {
  long off = 0;
  for (int k = 0; k < 2 * nlayers; ++k) {
    const int c = (k & 1) ? d / 2 : (d + 1) / 2, m = d - c;
    make_net(off, m, h1, h2, c);
    make_net(off, m, h1, h2, c);
  }
  return off;
}

// One step (or, with lr = 0 and grad_out != NULL, just loss and gradient).  xs: NULL (draw in place with `seed`) or
// N x d row-major (x[j*d + i], the memory image of Julia's d x N matrix).  N must be a multiple of 64.
// Returns 0, or -1 for bad arguments.
int nfcpu_realnvp_step(int d, int h1, int h2, int nlayers, float *theta, float *mom, float *vel, const float *mu,
                       const float *var, const float *xs, long N, uint64_t seed, int t_step, float lr, float *loss_out,
                       float *gnorm_out, float *grad_out, int nthreads) {
  if (d < 2 || h1 < 1 || h2 < 1 || nlayers < 1 || N < TS || N % TS != 0 || !theta || !mu || !var) return -1;
  const int nc = 2 * nlayers;
  std::vector<Coupling> cps(nc);
  long off = 0;
  for (int k = 0; k < nc; ++k) {
    Coupling &cp = cps[k];
    cp.par_t = k & 1;
    cp.c = (k & 1) ? d / 2 : (d + 1) / 2;
    cp.m = d - cp.c;
    cp.s = make_net(off, cp.m, h1, h2, cp.c);
    cp.t = make_net(off, cp.m, h1, h2, cp.c);
  }
  const long P = off;
  if (nthreads < 1) nthreads = omp_get_max_threads();
  // transposed copy of every weight matrix at the same offsets ([nout][nin]), for the dX products
  std::vector<float> thetaT((size_t)P, 0.f);
  auto transpose = [&](long woff, int nin, int nout) {
    for (int i = 0; i < nin; ++i)
      for (int k = 0; k < nout; ++k) thetaT[(size_t)woff + (size_t)k * nin + i] = theta[woff + (long)i * nout + k];
  };
  for (const Coupling &cp : cps)
    for (const Net *n : {&cp.s, &cp.t}) {
      transpose(n->w1, n->m, n->h1);
      transpose(n->w2, n->h1, n->h2);
      transpose(n->w3, n->h2, n->c);
    }
  const float *tT = thetaT.data();
  const long ntiles = N / TS;
  const int cmax = (d + 1) / 2;
  // per thread: gradient buffer, tile state, per-coupling activations of both nets
  const long per_net = (long)TS * (cmax + h1 + h2 + cmax);  // x2 | a1 | a2 | out
  const long scratch = (long)TS * d * 2 + (long)nc * 2 * per_net + (long)TS * (cmax * 3 + h1 + h2) + TS;
  std::vector<float> gbuf((size_t)nthreads * P, 0.f), sbuf((size_t)nthreads * scratch);
  std::vector<double> lossbuf(nthreads, 0.0);
  double c0 = 1.8378770664093453 * d;
  for (int i = 0; i < d; ++i) c0 += std::log((double)var[i]);
  const float invN = 1.0f / (float)N;

#pragma omp parallel num_threads(nthreads)
  {
    const int tid = omp_get_thread_num();
    float *g = gbuf.data() + (size_t)tid * P;
    float *sc = sbuf.data() + (size_t)tid * scratch;
    float *x = sc;                       // [TS][d] current state
    float *gb = x + (long)TS * d;        // [TS][d] cotangent
    float *acts = gb + (long)TS * d;     // [nc][2][per_net]
    float *tmp = acts + (long)nc * 2 * per_net;
    float *svals = tmp;                  // [TS][cmax]  tanh(s) of the current coupling (backward)
    float *d3 = svals + (long)TS * cmax; // [TS][cmax]
    float *d2 = d3 + (long)TS * cmax;    // [TS][h2]
    float *d1 = d2 + (long)TS * h2;      // [TS][h1]
    float *dx2 = d1 + (long)TS * h1;     // [TS][cmax]
    float *ladj = dx2 + (long)TS * cmax; // [TS]
    double lacc = 0.0;
    Rng rng(seed * 0x9E3779B97F4A7C15ull + (uint64_t)tid + 1);
#pragma omp for schedule(static)
    for (long tile = 0; tile < ntiles; ++tile) {
      // ---- draws + log q0
      float logq[TS];
      if (xs) {
        std::memcpy(x, xs + tile * TS * d, sizeof(float) * TS * d);
      } else {
        for (long e = 0; e + 1 < (long)TS * d; e += 2) rng.normals2(x[e], x[e + 1]);
        if ((TS * d) & 1) { float a, b; rng.normals2(a, b); x[(long)TS * d - 1] = a; }
      }
      for (int j = 0; j < TS; ++j) {
        float ss = 0.f;
        for (int i = 0; i < d; ++i) ss += x[j * d + i] * x[j * d + i];
        logq[j] = -0.5f * 1.8378770664093453f * d - 0.5f * ss;
        ladj[j] = 0.f;
      }
      // ---- forward: the LAST flat coupling first
      for (int k = nc - 1; k >= 0; --k) {
        const Coupling &cp = cps[k];
        float *as = acts + ((long)k * 2 + 0) * per_net, *at = acts + ((long)k * 2 + 1) * per_net;
        float *x2 = as, *a1s = x2 + (long)TS * cmax, *a2s = a1s + (long)TS * h1, *os = a2s + (long)TS * h2;
        float *a1t = at + (long)TS * cmax, *a2t = a1t + (long)TS * h1, *ot = a2t + (long)TS * h2;
        const int pc = 1 - cp.par_t;
        for (int j = 0; j < TS; ++j)
          for (int q = 0; q < cp.m; ++q) x2[j * cmax + q] = x[j * d + 2 * q + pc];
        dense_fwd(theta + cp.s.w1, theta + cp.s.b1, x2, cp.m, h1, a1s, cmax, h1, true);
        dense_fwd(theta + cp.s.w2, theta + cp.s.b2, a1s, h1, h2, a2s, h1, h2, true);
        dense_fwd(theta + cp.s.w3, theta + cp.s.b3, a2s, h2, cp.c, os, h2, cmax, false);
        dense_fwd(theta + cp.t.w1, theta + cp.t.b1, x2, cp.m, h1, a1t, cmax, h1, true);
        dense_fwd(theta + cp.t.w2, theta + cp.t.b2, a1t, h1, h2, a2t, h1, h2, true);
        dense_fwd(theta + cp.t.w3, theta + cp.t.b3, a2t, h2, cp.c, ot, h2, cmax, false);
        for (int j = 0; j < TS; ++j)
          for (int p = 0; p < cp.c; ++p) {
            const float s = std::tanh(os[j * cmax + p]);
            os[j * cmax + p] = s;  // keep tanh(s); x1 is recovered from y1 in the backward sweep
            float &v = x[j * d + 2 * p + cp.par_t];
            v = v * std::exp(s) + ot[j * cmax + p];
            ladj[j] += s;
          }
      }
      // ---- target, ELBO terms, cotangent of y
      for (int j = 0; j < TS; ++j) {
        float tq = 0.f;
        for (int i = 0; i < d; ++i) {
          const float r = x[j * d + i] - mu[i], gi = r / var[i];
          tq += r * gi;
          gb[j * d + i] = gi * invN;  // d(-elbo/N)/dy = (y - mu)/var / N
        }
        const float elbo = -0.5f * ((float)c0 + tq) - logq[j] + ladj[j];
        lacc += (double)elbo;
      }
      // ---- backward: flat order = reverse of execution order; lbar = -1/N for every coupling
      for (int k = 0; k < nc; ++k) {
        const Coupling &cp = cps[k];
        float *as = acts + ((long)k * 2 + 0) * per_net, *at = acts + ((long)k * 2 + 1) * per_net;
        float *x2 = as, *a1s = x2 + (long)TS * cmax, *a2s = a1s + (long)TS * h1, *os = a2s + (long)TS * h2;
        float *a1t = at + (long)TS * cmax, *a2t = a1t + (long)TS * h1, *ot = a2t + (long)TS * h2;
        const int pc = 1 - cp.par_t;
        // t net: Tbar = ybar1
        for (int j = 0; j < TS; ++j)
          for (int p = 0; p < cp.c; ++p) d3[j * cmax + p] = gb[j * d + 2 * p + cp.par_t];
        auto net_bwd = [&](const Net &n, const float *a1, const float *a2) {
          dense_bwd(tT + n.w3, a2, d3, h2, cp.c, h2, cmax, g + n.w3, g + n.b3, d2, h2);
          for (long e = 0; e < (long)TS * h2; ++e) d2[e] *= a2[e] > 0.f ? 1.f : 0.01f;
          dense_bwd(tT + n.w2, a1, d2, h1, h2, h1, h2, g + n.w2, g + n.b2, d1, h1);
          for (long e = 0; e < (long)TS * h1; ++e) d1[e] *= a1[e] > 0.f ? 1.f : 0.01f;
          dense_bwd(tT + n.w1, x2, d1, cp.m, h1, cmax, h1, g + n.w1, g + n.b1, dx2, cmax);
          for (int j = 0; j < TS; ++j)
            for (int q = 0; q < cp.m; ++q) gb[j * d + 2 * q + pc] += dx2[j * cmax + q];
        };
        net_bwd(cp.t, a1t, a2t);
        // s net: x1 = (y1 - T) exp(-S);  Sbar = (ybar1 x1 exp(S) + lbar)(1 - S^2);  x1bar = ybar1 exp(S)
        for (int j = 0; j < TS; ++j)
          for (int p = 0; p < cp.c; ++p) {
            const float s = os[j * cmax + p], es = std::exp(s);
            float &yv = x[j * d + 2 * p + cp.par_t];
            float &gv = gb[j * d + 2 * p + cp.par_t];
            const float u = yv - ot[j * cmax + p];
            d3[j * cmax + p] = (gv * u - invN) * (1.f - s * s);
            yv = u / es;
            gv = gv * es;
          }
        net_bwd(cp.s, a1s, a2s);
      }
    }
    lossbuf[tid] = lacc;
  }
  // ---- fixed-order reduction, Adam, norm
  double lsum = 0.0;
  for (int t = 0; t < nthreads; ++t) lsum += lossbuf[t];
  double gn2 = 0.0;
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  const float c1 = 1.f - std::pow(b1, (float)t_step), c2 = 1.f - std::pow(b2, (float)t_step);
#pragma omp parallel for num_threads(nthreads) reduction(+ : gn2) schedule(static)
  for (long p = 0; p < P; ++p) {
    float gsum = 0.f;
    for (int t = 0; t < nthreads; ++t) gsum += gbuf[(size_t)t * P + p];
    gn2 += (double)gsum * gsum;
    if (grad_out) grad_out[p] = gsum;
    if (lr > 0.f && mom && vel) {
      mom[p] = b1 * mom[p] + (1.f - b1) * gsum;
      vel[p] = b2 * vel[p] + (1.f - b2) * gsum * gsum;
      theta[p] -= lr * (mom[p] / c1) / (std::sqrt(vel[p] / c2) + eps);
    }
  }
  if (loss_out) *loss_out = (float)(-lsum / (double)N);
  if (gnorm_out) *gnorm_out = (float)std::sqrt(gn2);
  return 0;
}

}  // extern "C"
