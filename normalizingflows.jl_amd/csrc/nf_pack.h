// nf_pack.h -- theta <-> padded-image packing shared by the coupling translation units.
// Reference parameter order: Optimisers.destructure(flow) (src/NormalizingFlows.jl:67); couplings
// alternate odd mask / even mask (src/flows/realnvp.jl:138-144).
#pragma once
#include "nf_common.h"
#include "nf_mfma.h"

// ------------------------------------------------------------------------------------
// weight packing: theta -> padded LDS images, one per (coupling, net), once per call
// ------------------------------------------------------------------------------------
struct PackArgs {
  int d, h1, h2, ncoup;
  long pair_params;   // parameters of one RealNVP_layer (two couplings)
  long odd_params;    // parameters of the odd-mask coupling
};

template <class G>
__global__ __launch_bounds__(256) void k_pack_net_images(PackArgs p, const float *__restrict__ theta,
                                                         float *__restrict__ out) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)p.ncoup * 2 * G::SIZE;
  if (gid >= total) return;
  const int img = (int)(gid / G::SIZE), e = (int)(gid - (long)img * G::SIZE);
  const int k = img >> 1, net = img & 1;
  const int c = (k & 1) ? p.d / 2 : (p.d + 1) / 2, m = p.d - c;
  long off = (long)(k >> 1) * p.pair_params + ((k & 1) ? p.odd_params : 0);
  if (net) off += net_param_count(m, p.h1, p.h2, c);
  const NetDims nd = make_net_dims(off, m, p.h1, p.h2, c);
  const long ti = e < G::B3 + 32 * G::CB ? image_theta_index<G>(nd, e) : -1;
  out[gid] = ti >= 0 ? theta[ti] : 0.f;
}

// B6 images (nf_mfma.h, B6Geo) FROM the fp32 images: every weight split into three bf16, rows of eight in the MFMA's k
// order.  The fp32 images are what every writer of weights maintains (k_pack_net_images, the fused epilogue); the chain
// launches that use the bf16 products rebuild their B6 copy from them when the images have changed since (ctx->wimg_gen).
// One thread per (image, layer, k-group, half, row): eight weights in, 3 x 16 bytes out; one more per bias element.
template <class G>
__global__ __launch_bounds__(256) void k_b6_from_images(int nimg, const float *__restrict__ wimg, unsigned char *__restrict__ out) {
  using B = B6Geo<G>;
  constexpr int N1 = 2 * G::MB * 2 * B::R1, N2 = 2 * G::H1B * 2 * B::R2, N3 = 2 * G::H2B * 2 * B::R3;  // (kg, hi, row) triples
  constexpr int NB = B::R1 + B::R2 + B::R3;
  constexpr int PER = N1 + N2 + N3 + NB;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)nimg * PER) return;
  const int img = (int)(gid / PER);
  int e = (int)(gid - (long)img * PER);
  const float *src = wimg + (size_t)img * G::SIZE;
  unsigned char *base = out + (size_t)img * B::BYTES;
  int lay, rows, wsrc, S;
  if (e < N1) { lay = B::L1; rows = B::R1; wsrc = G::W1; S = G::S1; }
  else if (e < N1 + N2) { e -= N1; lay = B::L2; rows = B::R2; wsrc = G::W2; S = G::S2; }
  else if (e < N1 + N2 + N3) { e -= N1 + N2; lay = B::L3; rows = B::R3; wsrc = G::W3; S = G::S3; }
  else {  // bias element (fp32): the three bias vectors back to back
    e -= N1 + N2 + N3;
    const float v = e < B::R1 ? src[G::B1 + e] : e < B::R1 + B::R2 ? src[G::B2 + e - B::R1] : src[G::B3 + e - B::R1 - B::R2];
    reinterpret_cast<float *>(base + (size_t)B::BIAS * 16)[e] = v;
    return;
  }
  const int row = e % rows, hi = (e / rows) & 1, kg = e / (2 * rows);
  unsigned short h[8], mi[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int f = 16 * kg + (j & 3) + 8 * (j >> 2) + 4 * hi;  // input feature of k-slot (kg, hi, j)
    const float w = src[wsrc + f * S + row];                   // image element [in f][out row], zero padded
    nf_split1(w, h[j], mi[j], l[j]);  // round-to-nearest parts, as the activations' (nf_mfma.h)
  }
  auto put = [&](int comp, const unsigned short(&v)[8]) {
    nf_u32x4 q;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) q[pp] = (unsigned)v[2 * pp] | ((unsigned)v[2 * pp + 1] << 16);
    reinterpret_cast<nf_u32x4 *>(base)[lay + ((kg * 3 + comp) * 2 + hi) * rows + row] = q;
  };
  put(0, h);
  put(1, mi);
  put(2, l);
}

// B6T images (nf_mfma.h, B6TGeo) from the fp32 images: the same weights, rows = a layer's inputs, k over its outputs.
template <class G>
__global__ __launch_bounds__(256) void k_b6t_from_images(int nimg, const float *__restrict__ wimg, unsigned char *__restrict__ out) {
  using B = B6TGeo<G>;
  constexpr int N3 = 2 * G::CB * 2 * B::R3, N2 = 2 * G::H2B * 2 * B::R2, N1 = 2 * G::H1B * 2 * B::R1;  // (kg, hi, row) triples
  constexpr int PER = N3 + N2 + N1;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)nimg * PER) return;
  const int img = (int)(gid / PER);
  int e = (int)(gid - (long)img * PER);
  const float *src = wimg + (size_t)img * G::SIZE;
  unsigned char *base = out + (size_t)img * B::BYTES;
  int lay, rows, wsrc, S;
  if (e < N3) { lay = B::T3; rows = B::R3; wsrc = G::W3; S = G::S3; }
  else if (e < N3 + N2) { e -= N3; lay = B::T2; rows = B::R2; wsrc = G::W2; S = G::S2; }
  else { e -= N3 + N2; lay = B::T1; rows = B::R1; wsrc = G::W1; S = G::S1; }
  const int row = e % rows, hi = (e / rows) & 1, kg = e / (2 * rows);
  unsigned short h[8], mi[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int o = 16 * kg + (j & 3) + 8 * (j >> 2) + 4 * hi;  // output feature of k-slot (kg, hi, j)
    const float w = src[wsrc + row * S + o];                   // image element [in row][out o]
    nf_split1(w, h[j], mi[j], l[j]);  // round-to-nearest parts, as the activations' (nf_mfma.h)
  }
  auto put = [&](int comp, const unsigned short(&v)[8]) {
    nf_u32x4 q;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) q[pp] = (unsigned)v[2 * pp] | ((unsigned)v[2 * pp + 1] << 16);
    reinterpret_cast<nf_u32x4 *>(base)[lay + ((kg * 3 + comp) * 2 + hi) * rows + row] = q;
  };
  put(0, h);
  put(1, mi);
  put(2, l);
}

// g[theta index] = sum over workgroup slabs of the image-layout partial gradients
// Block 0 can also finish a deterministic sum of `nlpart` double partials into *lout (the step's loss:
// saves a separate one-block launch in the training step).
#ifndef NF_REDUCE_WAVES
#define NF_REDUCE_WAVES 4
#endif
template <class G>
__global__ __launch_bounds__(64 * NF_REDUCE_WAVES) void k_reduce_image_slabs(PackArgs p, const float *__restrict__ slab, int nslab,
                                                            long slab_stride, float *__restrict__ g,
                                                            const double *__restrict__ lpart, int nlpart,
                                                            float *__restrict__ lout) {
  if (lout && blockIdx.x == 0) {
    __shared__ double sm[NF_REDUCE_WAVES];
    double c = 0.0;
    for (int i = threadIdx.x; i < nlpart; i += 64 * NF_REDUCE_WAVES) c += lpart[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < NF_REDUCE_WAVES; ++w) t += sm[w];
      *lout = (float)t;
    }
  }
  // 64 image elements per block, the slabs split over the block's waves (a streaming kernel of 139 MB at cfg 2:
  // with one element per thread over all slabs only ~8 waves per CU were in flight and it ran at 3.4 TB/s)
  __shared__ float part[NF_REDUCE_WAVES][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long gid = (long)blockIdx.x * 64 + lane;
  const long total = (long)p.ncoup * 2 * G::SIZE;
  long ti = -1;
  if (gid < total) {
    const int img = (int)(gid / G::SIZE), e = (int)(gid - (long)img * G::SIZE);
    const int k = img >> 1, net = img & 1;
    const int c = (k & 1) ? p.d / 2 : (p.d + 1) / 2, m = p.d - c;
    long off = (long)(k >> 1) * p.pair_params + ((k & 1) ? p.odd_params : 0);
    if (net) off += net_param_count(m, p.h1, p.h2, c);
    const NetDims nd = make_net_dims(off, m, p.h1, p.h2, c);
    ti = e < G::B3 + 32 * G::CB ? image_theta_index<G>(nd, e) : -1;
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (ti >= 0) {  // wave q sums slabs q, q + W, q + 2 W, ...
    constexpr int W = NF_REDUCE_WAVES;
    int s = q;
    for (; s + 3 * W < nslab; s += 4 * W) {
      a0 += slab[(long)s * slab_stride + gid];
      a1 += slab[(long)(s + W) * slab_stride + gid];
      a2 += slab[(long)(s + 2 * W) * slab_stride + gid];
      a3 += slab[(long)(s + 3 * W) * slab_stride + gid];
    }
    for (; s < nslab; s += W) a0 += slab[(long)s * slab_stride + gid];
  }
  part[q][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (q == 0 && ti >= 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NF_REDUCE_WAVES; ++w) t += part[w][lane];
    g[ti] = t;
  }
}

// ------------------------------------------------------------------------------------
// fused epilogue of the LDS-resident RealNVP training step (nf_elbo_step)
// ------------------------------------------------------------------------------------
// One launch does what k_reduce_image_slabs + k_adam + k_finish_sum + the NEXT step's k_pack_net_images did in four:
// every thread owns one element of the padded weight images, and the slab sum, the gradient, Adam
// (Optimisers.update!, src/optimize.jl:99), the gradient-norm partial (src/optimize.jl:89) and the packed image of the
// UPDATED theta are all element-wise in that index; the blocks' partial sums of g^2 are finished by k_finish_sum.
//   REDUCE: g <- sum of slabs (else g is read: the multi-GPU form, after the all-reduce of [grad ; loss])
//   ADAM:   theta / m / v / wimg updated, g[P + 1] <- ||g||
// Adam's step count t = step + 1 comes from a.t_val or, for hipGraph replay, from the device counter a.t_ptr (incremented
// by the finishing launch, after every block of this one has read it).
struct EpiArgs {
  const float *slab;
  int nslab;
  long slab_stride;
  float *g;             // [P + 2]: gradient, loss, gradient norm
  long P;
  const double *lpart;  // loss partials of the forward launch (REDUCE), finished into g[P]
  int nlpart;
  float *theta, *m, *v, *wimg;
  float lr, b1, b2, eps;
  double b1d, b2d;
  float c1, c2;  // 1 - b1^t, 1 - b2^t for t = t_val + 1, computed on the host as nf_launch_adam does (t_ptr == nullptr)
  unsigned t_val;
  unsigned *t_ptr;
  double *gpart;        // [gridDim.x] partial sums of g^2 (ADAM)
};

template <class G, bool REDUCE, bool ADAM, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_affine_epilogue(PackArgs p, EpiArgs a) {
  if (REDUCE && a.lpart && blockIdx.x == 0) {
    __shared__ double sm[WAVES];
    double c = 0.0;
    for (int i = threadIdx.x; i < a.nlpart; i += 64 * WAVES) c += a.lpart[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < WAVES; ++w) t += sm[w];
      a.g[a.P] = (float)t;
    }
  }
  __shared__ float part[WAVES][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long gid = (long)blockIdx.x * 64 + lane;
  const long total = (long)p.ncoup * 2 * G::SIZE;
  long ti = -1;
  if (gid < total) {
    const int img = (int)(gid / G::SIZE), e = (int)(gid - (long)img * G::SIZE);
    const int k = img >> 1, net = img & 1;
    const int c = (k & 1) ? p.d / 2 : (p.d + 1) / 2, m = p.d - c;
    long off = (long)(k >> 1) * p.pair_params + ((k & 1) ? p.odd_params : 0);
    if (net) off += net_param_count(m, p.h1, p.h2, c);
    const NetDims nd = make_net_dims(off, m, p.h1, p.h2, c);
    ti = e < G::B3 + 32 * G::CB ? image_theta_index<G>(nd, e) : -1;
  }
  // Adam's operands are requested before the slab stream so that their latency hides behind it
  float th = 0.f, mi = 0.f, vi = 0.f;
  if (ADAM && q == 0 && ti >= 0) {
    th = a.theta[ti];
    mi = a.m[ti];
    vi = a.v[ti];
  }
  float gsum = 0.f;
  if (REDUCE) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (ti >= 0) {  // wave q sums slabs q, q + W, q + 2 W, ... (the order of k_reduce_image_slabs)
      constexpr int W = WAVES;
      int s = q;
      // eight loads in flight per thread (same summation order as the four-way loop below: 39-41 -> 37 us at cfg 2)
      // (round 4, measured and removed: four elements per thread with 16-byte loads -- 35.6-36.2 us, no change; 16 waves per
      // block instead of 4 -- 45 us)
      for (; s + 7 * W < a.nslab; s += 8 * W) {
        const float v0 = a.slab[(long)s * a.slab_stride + gid], v1 = a.slab[(long)(s + W) * a.slab_stride + gid];
        const float v2 = a.slab[(long)(s + 2 * W) * a.slab_stride + gid], v3 = a.slab[(long)(s + 3 * W) * a.slab_stride + gid];
        const float v4 = a.slab[(long)(s + 4 * W) * a.slab_stride + gid], v5 = a.slab[(long)(s + 5 * W) * a.slab_stride + gid];
        const float v6 = a.slab[(long)(s + 6 * W) * a.slab_stride + gid], v7 = a.slab[(long)(s + 7 * W) * a.slab_stride + gid];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        a0 += v4; a1 += v5; a2 += v6; a3 += v7;
      }
      for (; s + 3 * W < a.nslab; s += 4 * W) {
        a0 += a.slab[(long)s * a.slab_stride + gid];
        a1 += a.slab[(long)(s + W) * a.slab_stride + gid];
        a2 += a.slab[(long)(s + 2 * W) * a.slab_stride + gid];
        a3 += a.slab[(long)(s + 3 * W) * a.slab_stride + gid];
      }
      for (; s < a.nslab; s += W) a0 += a.slab[(long)s * a.slab_stride + gid];
    }
    part[q][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q == 0) {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) gsum += part[w][lane];
    }
  } else if (q == 0 && ti >= 0) {
    gsum = a.g[ti];
  }
  if (q != 0) return;
  double gg = 0.0;
  if (ti >= 0) {
    if (REDUCE) a.g[ti] = gsum;
    if (ADAM) {
      float c1 = a.c1, c2 = a.c2;
      if (a.t_ptr) {  // graph replay: the step count lives on the device
        const double t = (double)(*a.t_ptr + 1u);
        c1 = (float)(1.0 - pow(a.b1d, t));
        c2 = (float)(1.0 - pow(a.b2d, t));
      }
      nf_adam_elem<float>(th, mi, vi, gsum, a.lr, a.b1, a.b2, a.eps, c1, c2);
      a.m[ti] = mi;
      a.v[ti] = vi;
      a.theta[ti] = th;
      a.wimg[gid] = th;  // padding elements of the image stay zero from the first pack
      gg = (double)gsum * (double)gsum;
    }
  }
  if (!ADAM) return;
  // norm(g): this block's partial of sum g^2; k_finish_sum (the step's last, one-block launch) adds the partials in
  // block order and takes the root.  (A completion counter inside this kernel -- last block finishes -- was measured:
  // 2 128 agent-scope atomics cost 12.5 us, acquire / release fences 65 us; the separate launch costs 4.)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gg += __shfl_xor(gg, o, 64);
  if (lane == 0) a.gpart[blockIdx.x] = gg;
}

static inline PackArgs make_pack_args(const nf_flow_desc *desc) {
  PackArgs p;
  p.d = desc->d; p.h1 = desc->hdims[0]; p.h2 = desc->hdims[1]; p.ncoup = 2 * desc->nlayers;
  const CouplingInfo c0 = nf_coupling_info(desc, 0), c1 = nf_coupling_info(desc, 1);
  p.odd_params = c0.nparams;
  p.pair_params = c0.nparams + c1.nparams;
  return p;
}

