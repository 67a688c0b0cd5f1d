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

// g[theta index] = sum over workgroup slabs of the image-layout partial gradients
// Block 0 can also finish a deterministic sum of `nlpart` double partials into *lout (the step's loss:
// saves a separate one-block launch in the training step).
template <class G>
__global__ __launch_bounds__(256) void k_reduce_image_slabs(PackArgs p, const float *__restrict__ slab, int nslab,
                                                            long slab_stride, float *__restrict__ g,
                                                            const double *__restrict__ lpart, int nlpart,
                                                            float *__restrict__ lout) {
  if (lout && blockIdx.x == 0) {
    __shared__ double sm[4];
    double c = 0.0;
    for (int i = threadIdx.x; i < nlpart; i += 256) c += lpart[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) *lout = (float)((sm[0] + sm[1]) + (sm[2] + sm[3]));
  }
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
  if (ti < 0) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 3 < nslab; s += 4) {
    a0 += slab[(long)s * slab_stride + gid];
    a1 += slab[(long)(s + 1) * slab_stride + gid];
    a2 += slab[(long)(s + 2) * slab_stride + gid];
    a3 += slab[(long)(s + 3) * slab_stride + gid];
  }
  for (; s < nslab; ++s) a0 += slab[(long)s * slab_stride + gid];
  g[ti] = (a0 + a1) + (a2 + a3);
}

static inline PackArgs make_pack_args(const nf_flow_desc *desc) {
  PackArgs p;
  p.d = desc->d; p.h1 = desc->hdims[0]; p.h2 = desc->hdims[1]; p.ncoup = 2 * desc->nlayers;
  const CouplingInfo c0 = nf_coupling_info(desc, 0), c1 = nf_coupling_info(desc, 1);
  p.odd_params = c0.nparams;
  p.pair_params = c0.nparams + c1.nparams;
  return p;
}

