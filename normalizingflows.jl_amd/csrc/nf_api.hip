// nf_api.hip -- extern "C" entry points of libnfhip.so (see include/nfhip.h for the
// reference interface each one stands behind).  Host-side orchestration only: every
// arithmetic operation happens in a gfx950 kernel; there is no CPU fallback.
#include <cmath>
#include <cstdlib>

#include "nf_common.h"

// ---- kernels' host launchers (other translation units) ---------------------------------
int nf_launch_base_sample(nf_ctx *, int, int, long, uint64_t, uint64_t, uint32_t, void *, void *);
int nf_launch_base_logpdf(nf_ctx *, int, int, long, const void *, void *);
int nf_launch_base_unwhiten(nf_ctx *, int dtype, int kind, int d, long N, const void *mu, const void *scale, void *x);
int nf_launch_base_general_logpdf(nf_ctx *, int dtype, int kind, int d, long N, const void *mu, const void *scale, double logdet,
                                  const void *x, void *logq_out, void *corr_out, void *zbuf);
int nf_launch_base_general_score(nf_ctx *, int dtype, int kind, int d, long N, const void *mu, const void *scale, double logdet,
                                 const void *x, void *logq_out, void *score_out, double gscale, void *zbuf);
long nf_target_nblocks(long N);
int nf_launch_target(nf_ctx *, int, const nf_target *, int, long, const void *, const void *, const void *, void *,
                     void *, double, void *, double *, double, int joint_d);
long nf_sum2_nblocks(long N);
int nf_launch_sum2(nf_ctx *, int, long, const void *, const void *, void *, double *, double);
int nf_launch_finish_sum(nf_ctx *, const double *, long, int, double *, float *, double *, unsigned *bump = nullptr);
int nf_launch_reduce_slabs(nf_ctx *, int, const void *, int, long, void *);
long nf_adam_nblocks(long P);
int nf_launch_adam(nf_ctx *, int, void *, const void *, void *, void *, long, double, double, double, double, long,
                   double *);
int nf_launch_sgd(nf_ctx *, int, void *, const void *, void *, long, double, double, double *);
int nf_launch_fill(nf_ctx *, int, void *, long, double);
int nf_launch_base_sample_tiled(nf_ctx *, int d, long N, uint64_t seed, uint64_t off, uint32_t stream, float *xt, float *logq);
int nf_launch_base_logpdf_tiled(nf_ctx *, int d, long N, const float *xt, float *logq);
long nf_target_tiled_nblocks(long N);
int nf_launch_target_tiled(nf_ctx *, const nf_target *, int d, long N, const float *yt, const float *logq,
                           const float *ladj, float *gt, double gscale, float *elbos_out, double *partial, double pscale);
int nf_launch_layout_convert(nf_ctx *, int d, long N, const float *src, float *dst, int to_tiled);

bool nf_affine_supported(const nf_flow_desc *desc);
int nf_affine_apply(nf_ctx *, const nf_flow_desc *, int k, bool inverse, const float *theta, const float *x, long N,
                    float *y, float *ladj, int accumulate);
int nf_affine_bwd_grid(nf_ctx *, long N);
int nf_affine_pack(nf_ctx *, const nf_flow_desc *, const float *theta);
int nf_affine_chain(nf_ctx *, const nf_flow_desc *, bool inverse, float *xt, long N, float *ladj, float *stash = nullptr);
long nf_affine_chain_grid(nf_ctx *, long N);
int nf_affine_chain_elbo(nf_ctx *, const nf_flow_desc *, long N, uint64_t seed, uint64_t off, uint32_t stream,
                         const float *mu, const float *var, float *yt, float *gt, double gscale, double *partial,
                         double pscale, float *stash = nullptr, const uint32_t *stream_ptr = nullptr);
long nf_affine_epilogue_blocks(const nf_flow_desc *desc);
int nf_affine_epilogue(nf_ctx *, const nf_flow_desc *, int mode, const float *slab, int nslab, float *g, const double *lpart,
                       int nlpart, float *theta, float *m, float *v, double lr, double b1, double b2, double eps, unsigned t_val,
                       unsigned *t_ptr, double *gpart);
size_t nf_affine_stash_floats(const nf_flow_desc *desc, long N);
bool nf_affine_stash_pays(const nf_flow_desc *desc);
bool nf_affine_fused_ok(const nf_flow_desc *desc);
int nf_affine_bwd_stashed(nf_ctx *, const nf_flow_desc *, float *stash, float *ybar, const float *lbar, float lbar_const, long N,
                          float *slab, long slab_stride, int grid, bool inv_dir = false);
long nf_affine_slab_floats(const nf_flow_desc *desc);
int nf_affine_reduce_slabs(nf_ctx *, const nf_flow_desc *, const float *slab, int nslab, float *g,
                           const double *lpart = nullptr, int nlpart = 0, float *lout = nullptr);
int nf_affine_bwd(nf_ctx *, const nf_flow_desc *, int k, const float *theta, float *y, float *ybar, const float *lbar,
                  float lbar_const, long N, float *slab, long slab_stride, int grid);
int nf_affine_bwd_all(nf_ctx *, const nf_flow_desc *, float *y, float *ybar, const float *lbar, float lbar_const, long N,
                      float *slab, long slab_stride, int grid, bool inv_dir = false);

// RealNVP with conditioner nets streamed from L2 (nf_wide.hip): d <= 256, hidden <= 256
bool nf_wide_supported(const nf_flow_desc *desc);
int nf_wide_pack(nf_ctx *, const nf_flow_desc *, const float *theta);
int nf_wide_apply(nf_ctx *, const nf_flow_desc *, int k, bool inverse, float *xt, long N, float *ladj, int accumulate);
size_t nf_wide_bwd_ws_floats(nf_ctx *, const nf_flow_desc *, long N);
int nf_wide_bwd(nf_ctx *, const nf_flow_desc *, float *state, float *gbar, const float *lbar, float lbar_const, long N,
                float *ws, float *g_out, bool inv_dir = false);
size_t nf_wide_train_ws_floats(nf_ctx *, const nf_flow_desc *, long N);
int nf_wide_train_forward(nf_ctx *, const nf_flow_desc *, float *xt, long N, float *ladj, float *ws);
int nf_wide_train_backward(nf_ctx *, const nf_flow_desc *, float *state, float *gbar, const float *lbar,
                           float lbar_const, long N, float *ws, float *g_out, float *scratch = nullptr);
size_t nf_wide_fwd_stash_floats(nf_ctx *, const nf_flow_desc *, long N);
size_t nf_wide_train_scratch_floats(nf_ctx *, const nf_flow_desc *, long N);

// neural spline couplings (nf_rqs.hip)
bool nf_rqs_supported(const nf_flow_desc *desc);
int nf_rqs_pack(nf_ctx *, const nf_flow_desc *, const float *theta);
int nf_rqs_chain(nf_ctx *, const nf_flow_desc *, bool inverse, float *xt, long N, float *ladj, int k_only, void *tape = nullptr);
size_t nf_rqs_tape_bytes(const nf_flow_desc *desc, long N);
int nf_rqs_bwd_grid(nf_ctx *, long N);
int nf_rqs_bwd(nf_ctx *, const nf_flow_desc *, int k, float *y, float *ybar, const float *lbar, float lbar_const, long N,
               float *slab, long slab_stride, int grid, bool inv_dir = false, void *tape = nullptr);
long nf_rqs_slab_floats(const nf_flow_desc *desc);
long nf_rqs_chain_grid(nf_ctx *, long N);
int nf_rqs_chain_elbo(nf_ctx *, const nf_flow_desc *, long N, uint64_t seed, uint64_t off, uint32_t stream,
                      const float *mu, const float *var, float *yt, float *gt, double gscale, double *partial,
                      double pscale, void *tape = nullptr);
int nf_rqs_reduce_slabs(nf_ctx *, const nf_flow_desc *, const float *slab, int nslab, float *g);
long nf_rqs_epilogue_blocks(const nf_flow_desc *desc);
int nf_rqs_epilogue(nf_ctx *, const nf_flow_desc *, const float *slab, int nslab, float *g, const double *lpart, int nlpart, float *theta,
                    float *m, float *v, double lr, double b1, double b2, double eps, unsigned t_val, double *gpart);

// planar / radial / mean-field flows (nf_simple.hip)
bool nf_simple_supported(const nf_flow_desc *desc);
int nf_simple_apply(nf_ctx *, const nf_flow_desc *, int layer_lo, int layer_hi, bool inverse, const void *theta,
                    const void *x, long N, void *y, void *ladj);
size_t nf_simple_bwd_ws_bytes(nf_ctx *, const nf_flow_desc *, long N);
int nf_simple_bwd(nf_ctx *, const nf_flow_desc *, const void *theta, const void *x, const void *ybar, const void *lbar,
                  double lbar_const, long N, void *xbar_out, void *gtheta_out, void *ws, bool have_stash, bool inv = false);
int nf_simple_apply_stash(nf_ctx *, const nf_flow_desc *, const void *theta, const void *x, long N, void *y, void *ladj,
                          void *ws, bool inverse = false);

long nf_simple_elbo_max_partials(nf_ctx *);
int nf_simple_elbo_forward(nf_ctx *, const nf_flow_desc *, const nf_target *, const void *theta, const void *xs, long N,
                           uint64_t seed, uint64_t off, uint32_t stream_id, void *gbar, double gscale, double *partial,
                           double pscale, void *ws, long *npartial);
int nf_target_check(const nf_target *t, int d);
bool nf_simple_step_supported(const nf_flow_desc *desc);
size_t nf_simple_step_ws_bytes(nf_ctx *, const nf_flow_desc *, long N);
int nf_simple_elbo_step(nf_ctx *, const nf_flow_desc *, const nf_target *, const void *theta, const void *xs, long N,
                        uint64_t seed, uint64_t off, uint32_t stream_id, double gscale, double lbar_const, double *partial,
                        double pscale, void *ws, void *gtheta_out, long *npartial);
int nf_simple_rand(nf_ctx *, const nf_flow_desc *, const void *theta, long N, uint64_t seed, uint64_t off, uint32_t stream_id,
                   void *y);

// general coupling kernels (nf_generic64.hip): Float64, and the Float32 shapes the MFMA paths do not
// build; one thread per sample, standard layout
bool nf_g64_supported(const nf_flow_desc *desc);
int nf_g64_apply(nf_ctx *, const nf_flow_desc *, int layer_lo, int layer_hi, bool inverse, const void *theta,
                 const void *x, long N, void *y, void *ladj);
size_t nf_g64_bwd_ws_bytes(const nf_flow_desc *desc, long N);
int nf_g64_apply_keep(nf_ctx *, const nf_flow_desc *, const void *theta, const void *x, long N, void *y, void *ladj, void *ws);
int nf_g64_bwd(nf_ctx *, const nf_flow_desc *, const void *theta, const void *x, const void *ybar, const void *lbar,
               double lbar_const, long N, void *xbar_out, void *gtheta_out, void *ws);

size_t nf_g64_bwd_inv_ws_bytes(const nf_flow_desc *desc, long N);
int nf_g64_bwd_inv(nf_ctx *, const nf_flow_desc *, const void *theta, void *z, void *gbar, double lbar_const, long N,
                   void *gtheta_out, void *ws);

// Hamiltonian flow of the demos (nf_hamiltonian.hip)
bool nf_hf_supported(const nf_flow_desc *desc);
int nf_hf_apply(nf_ctx *, const nf_flow_desc *, int lo, int hi, bool inverse, const void *theta, const void *x, long N,
                void *y, void *ladj);
size_t nf_hf_bwd_ws_bytes(const nf_flow_desc *desc, long N);
int nf_hf_bwd(nf_ctx *, const nf_flow_desc *, const void *theta, const void *x, const void *ybar, const void *lbar,
              double lbar_const, long N, void *xbar_out, void *gtheta_out, void *ws);

int nf_hf_bwd_inv(nf_ctx *, const nf_flow_desc *, const void *theta, const void *u, const void *gbar, double lbar_const,
                  long N, void *gtheta_out, void *ws);

// ---- helpers -----------------------------------------------------------------------------
static size_t ws_need_bound(nf_ctx *ctx, const nf_flow_desc *desc, long N);
static size_t flow_bwd_need(nf_ctx *ctx, const nf_flow_desc *desc, long N);
static size_t step_fused_need(nf_ctx *ctx, const nf_flow_desc *desc, long N);
static size_t vg_composite_need(nf_ctx *ctx, const nf_flow_desc *desc, long N);
static size_t base_extra_bytes(const nf_flow_desc *desc, long N);
static inline size_t esize(int dtype) { return dtype == NF_DTYPE_F64 ? 8 : 4; }

// general MvNormal(mu, Sigma) base of a flow, or nullptr for the standard normal
static inline const nf_base *flow_base(const nf_flow_desc *d) {
  return (d->base && d->base->kind != NF_BASE_STANDARD) ? d->base : nullptr;
}
static int check_base(const nf_base *b) {
  if (!b || b->kind == NF_BASE_STANDARD) return NF_OK;
  if (b->kind != NF_BASE_DIAG && b->kind != NF_BASE_DENSE) return NF_ERR_ARG;
  return (b->mu && b->scale) ? NF_OK : NF_ERR_ARG;
}

static inline bool is_composite(const nf_flow_desc *d) { return d->kind == NF_KIND_COMPOSITE; }
static int check_desc(const nf_flow_desc *d);
static int check_composite(const nf_flow_desc *d) {
  if (d->nsegments < 1 || d->nsegments > 64 || !d->segments) return NF_ERR_ARG;
  for (int s = 0; s < d->nsegments; ++s) {
    const nf_flow_desc *g = &d->segments[s];
    if (g->kind == NF_KIND_COMPOSITE || g->kind == NF_KIND_HAMILTONIAN) return NF_ERR_ARG;
    if (g->d != d->d || g->dtype != d->dtype || g->base) return NF_ERR_ARG;  // q0 belongs to the composition
    NF_TRY(check_desc(g));
  }
  return NF_OK;
}

static int check_desc(const nf_flow_desc *d) {
  if (!d) return NF_ERR_ARG;
  if (d->d < 1) return NF_ERR_ARG;
  if (d->dtype != NF_DTYPE_F32 && d->dtype != NF_DTYPE_F64) return NF_ERR_ARG;
  NF_TRY(check_base(d->base));
  if (d->kind == NF_KIND_COMPOSITE) return check_composite(d);
  if (d->nlayers < 1) return NF_ERR_ARG;
  if (d->dtype != NF_DTYPE_F32 && d->dtype != NF_DTYPE_F64) return NF_ERR_ARG;
  switch (d->kind) {
    case NF_KIND_PLANAR:
    case NF_KIND_RADIAL:
    case NF_KIND_MEANFIELD:
      return nf_simple_supported(d) ? NF_OK : NF_ERR_UNSUPPORTED;
    case NF_KIND_REALNVP:
      if (d->n_hidden < 1 || d->n_hidden > NF_MAX_HIDDEN) return NF_ERR_ARG;
      if (d->d < 2) return NF_ERR_ARG;
      if (d->dtype != NF_DTYPE_F32) return nf_g64_supported(d) ? NF_OK : NF_ERR_UNSUPPORTED;
      return (nf_affine_supported(d) || nf_wide_supported(d) || nf_g64_supported(d)) ? NF_OK : NF_ERR_UNSUPPORTED;
    case NF_KIND_HAMILTONIAN:
      if (d->d < 2 || (d->d & 1) || d->K < 1 || !d->score) return NF_ERR_ARG;
      return nf_hf_supported(d) ? NF_OK : NF_ERR_UNSUPPORTED;
    case NF_KIND_NSF:
      if (d->n_hidden < 1 || d->n_hidden > NF_MAX_HIDDEN || d->d < 2 || d->K < 2) return NF_ERR_ARG;
      if (d->dtype != NF_DTYPE_F32) return nf_g64_supported(d) ? NF_OK : NF_ERR_UNSUPPORTED;
      return (nf_rqs_supported(d) || nf_g64_supported(d)) ? NF_OK : NF_ERR_UNSUPPORTED;
    default:
      return NF_ERR_ARG;
  }
}

static long mlp_params(int nin, const nf_flow_desc *d, int nout) {
  long n = 0;
  int prev = nin;
  for (int i = 0; i < d->n_hidden; ++i) {
    n += (long)prev * d->hdims[i] + d->hdims[i];
    prev = d->hdims[i];
  }
  return n + (long)prev * nout + nout;
}

CouplingInfo nf_coupling_info(const nf_flow_desc *desc, int k) {
  const int d = desc->d;
  const int c_odd = (d + 1) / 2, c_even = d / 2;  // mask 1:2:d / 2:2:d
  auto npar = [&](int c) {
    const int m = d - c;
    if (desc->kind == NF_KIND_REALNVP) return 2 * mlp_params(m, desc, c);
    return mlp_params(m, desc, (3 * desc->K - 1) * c);
  };
  CouplingInfo ci;
  const long pair = npar(c_odd) + npar(c_even);
  ci.theta_off = (long)(k / 2) * pair + ((k & 1) ? npar(c_odd) : 0);
  ci.par_t = k & 1;
  ci.c = (k & 1) ? c_even : c_odd;
  ci.m = d - ci.c;
  ci.nparams = npar(ci.c);
  return ci;
}

extern "C" int64_t nf_param_count(const nf_flow_desc *d) {
  if (!d) return NF_ERR_ARG;
  if (d->kind == NF_KIND_COMPOSITE) {
    if (d->nsegments < 1 || !d->segments) return NF_ERR_ARG;
    int64_t tot = 0;
    for (int s = 0; s < d->nsegments; ++s) {
      if (d->segments[s].kind == NF_KIND_COMPOSITE) return NF_ERR_ARG;
      const int64_t p = nf_param_count(&d->segments[s]);
      if (p < 0) return p;
      tot += p;
    }
    return tot;
  }
  switch (d->kind) {
    case NF_KIND_PLANAR: return (int64_t)d->nlayers * (2 * d->d + 1);
    case NF_KIND_RADIAL: return (int64_t)d->nlayers * (d->d + 2);
    case NF_KIND_MEANFIELD: return 2 * (int64_t)d->d;
    case NF_KIND_HAMILTONIAN: return 2 * (int64_t)d->d + 3 * (int64_t)(d->d / 2) * d->nlayers;
    case NF_KIND_REALNVP:
    case NF_KIND_NSF: {
      CouplingInfo last = nf_coupling_info(d, 2 * d->nlayers - 1);
      return last.theta_off + last.nparams;
    }
    default: return NF_ERR_ARG;
  }
}

extern "C" int32_t nf_layer_count(const nf_flow_desc *d) {
  if (!d) return NF_ERR_ARG;
  if (d->kind == NF_KIND_COMPOSITE) {
    if (d->nsegments < 1 || !d->segments) return NF_ERR_ARG;
    int32_t tot = 0;
    for (int s = 0; s < d->nsegments; ++s) {
      if (d->segments[s].kind == NF_KIND_COMPOSITE) return NF_ERR_ARG;
      const int32_t c = nf_layer_count(&d->segments[s]);
      if (c < 0) return c;
      tot += c;
    }
    return tot;
  }
  switch (d->kind) {
    case NF_KIND_PLANAR:
    case NF_KIND_RADIAL: return d->nlayers;
    case NF_KIND_MEANFIELD: return 2;
    case NF_KIND_HAMILTONIAN: return d->nlayers + 1;  // blocks, then the reference's affine map
    case NF_KIND_REALNVP:
    case NF_KIND_NSF: return 2 * d->nlayers;
    default: return NF_ERR_ARG;
  }
}

// ---- library / context -------------------------------------------------------------------
extern "C" int nf_abi_version(void) { return NF_ABI_VERSION; }

extern "C" const char *nf_strerror(int code) {
  switch (code) {
    case NF_OK: return "ok";
    case NF_ERR_ARG: return "nfhip: invalid argument";
    case NF_ERR_UNSUPPORTED: return "nfhip: flow shape/dtype not built into this library";
    case NF_ERR_NO_DEVICE: return "nfhip: no usable HIP device";
    case NF_ERR_NONFINITE: return "nfhip: non-finite loss or gradient norm";
    case NF_ERR_WORKSPACE: return "nfhip: the caller-provided arena is too small (size it with nf_workspace_bytes)";
    case NF_ERR_NO_RCCL: return "nfhip: librccl.so.1 could not be loaded";
    case NF_ERR_RCCL: return nf_comm_last_error();
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "nfhip: unknown error";
  }
}

extern "C" int nf_ctx_create(int device, void *hip_stream, nf_ctx **out) {
  if (!out) return NF_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NF_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return NF_ERR_ARG;
  NF_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  NF_HIP(hipGetDeviceProperties(&prop, device));
  nf_ctx *c = new nf_ctx();
  c->device = device;
  c->stream = (hipStream_t)hip_stream;
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  hipError_t e = hipHostMalloc((void **)&c->host_scratch, 8 * sizeof(double), hipHostMallocDefault);
  if (e != hipSuccess) {
    delete c;
    return (int)e;
  }
  *out = c;
  return NF_OK;
}

extern "C" int nf_ctx_destroy(nf_ctx *ctx) {
  if (!ctx) return NF_ERR_ARG;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  nf_comm_destroy(ctx);
  for (auto &e : ctx->prof_pool) hipEventDestroy(e);
  if (!ctx->arena) {  // arena mode: ws / wimg / gbuf are the caller's memory
    if (ctx->ws) hipFree(ctx->ws);
    if (ctx->gbuf) hipFree(ctx->gbuf);
    if (ctx->wimg) hipFree(ctx->wimg);
  }
  if (ctx->trace) hipFree(ctx->trace);
  if (ctx->host_scratch) hipHostFree(ctx->host_scratch);
  delete ctx;
  return NF_OK;
}

extern "C" int nf_ctx_set_stream(nf_ctx *ctx, void *hip_stream) {
  if (!ctx) return NF_ERR_ARG;
  ctx->stream = (hipStream_t)hip_stream;
  ctx->wimg_owner = nullptr;  // cached weight images are only ordered against the stream they were written on
  return NF_OK;
}

extern "C" int nf_ctx_synchronize(nf_ctx *ctx) {
  if (!ctx) return NF_ERR_ARG;
  NF_HIP(hipStreamSynchronize(ctx->stream));
  return NF_OK;
}

// ---- workspace ------------------------------------------------------------------------------------
// Default: grow-only device arena (the first call of a larger shape allocates and synchronises once, steady-state
// steps never do).  With a caller-provided arena (nf_ctx_set_arena, sized by nf_workspace_bytes) no compute entry
// point allocates, frees or synchronises for memory at all: requests beyond the arena fail with NF_ERR_WORKSPACE.
int nf_ws_reserve(nf_ctx *ctx, size_t bytes) {
  // an outer entry point (general-base wrappers) keeps its own buffers behind the first ws_guard bytes
  if (ctx->ws_guard && bytes > ctx->ws_guard) return NF_ERR_WORKSPACE;
  if (ctx->arena) {
    const size_t avail = ctx->arena_bytes - ctx->arena_tail;
    if (bytes > avail) return NF_ERR_WORKSPACE;
    // front high-water mark of the running entry point: a later tail carve (weight images, nf_elbo_step's buffer)
    // must not reach into it.  Inside a wrapper (guard set) the outer request stays the mark.
    ctx->arena_front = (ctx->ws_guard && ctx->arena_front > bytes) ? ctx->arena_front : bytes;
    ctx->ws = ctx->arena;
    ctx->ws_bytes = avail;
    return NF_OK;
  }
  if (bytes <= ctx->ws_bytes) return NF_OK;
  NF_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->ws) NF_HIP(hipFree(ctx->ws));
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  const size_t want = carve_bytes(bytes + bytes / 8);  // multiple of 256: the optimiser's scratch is carved off the tail
  NF_HIP(hipMalloc(&ctx->ws, want));
  ctx->ws_bytes = want;
  return NF_OK;
}

// carve `bytes` off the tail of the caller's arena (arena mode only)
static int arena_tail_take(nf_ctx *ctx, size_t bytes, void **out) {
  const size_t b = carve_bytes(bytes);
  if (ctx->arena_tail + b > ctx->arena_bytes) return NF_ERR_WORKSPACE;
  // the running entry point has carved [0, arena_front) off the front already (nf_ws_reserve came first): a tail that
  // reaches into it would silently overlap live intermediates
  if (ctx->arena_bytes - ctx->arena_tail - b < ctx->arena_front) return NF_ERR_WORKSPACE;
  ctx->arena_tail += b;
  *out = (char *)ctx->arena + (ctx->arena_bytes - ctx->arena_tail);
  ctx->ws = ctx->arena;
  ctx->ws_bytes = ctx->arena_bytes - ctx->arena_tail;
  return NF_OK;
}

int nf_wimg_reserve(nf_ctx *ctx, size_t bytes) {
  ctx->wimg_gen++;  // every writer of packed weights comes through here first: derived copies (B6 images) are stale
  if (bytes <= ctx->wimg_bytes) return NF_OK;
  if (ctx->arena) {
    NF_TRY(arena_tail_take(ctx, bytes, &ctx->wimg));
    ctx->wimg_bytes = carve_bytes(bytes);
    return NF_OK;
  }
  NF_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->wimg) NF_HIP(hipFree(ctx->wimg));
  ctx->wimg = nullptr;
  ctx->wimg_bytes = 0;
  ctx->wimg_owner = nullptr;
  NF_HIP(hipMalloc(&ctx->wimg, bytes));
  ctx->wimg_bytes = bytes;
  return NF_OK;
}

// nf_elbo_step's device buffer: [grad (P) ; loss ; gradient norm] in the flow's element type, then (256-byte aligned) the
// fused epilogue's completion counter and the device-resident step counter of the graph-replay form.  Zeroed when
// (re)established: the completion counter must start at 0.
static inline size_t gbuf_state_off(long P, size_t es) { return carve_bytes((size_t)(P + 2) * es); }
static inline size_t gbuf_need(long P, size_t es) { return gbuf_state_off(P, es) + 256; }
static int gbuf_reserve(nf_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->gbuf_bytes) return NF_OK;
  if (ctx->arena) {
    NF_TRY(arena_tail_take(ctx, bytes, &ctx->gbuf));
    ctx->gbuf_bytes = carve_bytes(bytes);
  } else {
    NF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->gbuf) NF_HIP(hipFree(ctx->gbuf));
    ctx->gbuf = nullptr;
    ctx->gbuf_bytes = 0;
    NF_HIP(hipMalloc(&ctx->gbuf, bytes));
    ctx->gbuf_bytes = bytes;
  }
  NF_HIP(hipMemsetAsync(ctx->gbuf, 0, ctx->gbuf_bytes, ctx->stream));
  return NF_OK;
}

extern "C" int nf_ctx_set_arena(nf_ctx *ctx, void *arena, size_t bytes) {
  if (!ctx || (arena && bytes < 4096) || ((uintptr_t)arena & 255)) return NF_ERR_ARG;
  NF_HIP(hipSetDevice(ctx->device));
  NF_HIP(hipStreamSynchronize(ctx->stream));
  // drop whatever the context owns or had carved: the next calls re-establish their buffers in the new mode
  if (!ctx->arena) {
    if (ctx->ws) NF_HIP(hipFree(ctx->ws));
    if (ctx->wimg) NF_HIP(hipFree(ctx->wimg));
    if (ctx->gbuf) NF_HIP(hipFree(ctx->gbuf));
  }
  ctx->ws = ctx->wimg = ctx->gbuf = nullptr;
  ctx->ws_bytes = ctx->wimg_bytes = ctx->gbuf_bytes = 0;
  ctx->arena = arena;
  ctx->arena_bytes = arena ? (bytes / 256) * 256 : 0;
  ctx->arena_tail = 0;
  ctx->arena_front = 0;
  ctx->wimg_owner = nullptr;
  return NF_OK;
}

static int read_scalar(nf_ctx *ctx, const double *dev, double *host) {
  NF_HIP(hipMemcpyAsync(ctx->host_scratch, dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NF_HIP(hipStreamSynchronize(ctx->stream));
  *host = ctx->host_scratch[0];
  return NF_OK;
}

// buffers and helpers of the general-base wrappers (defined with the objectives below)
struct BaseBufs {
  char *x, *corr, *tmp, *z;
  double *partial, *result;
  long nb;
};
static int base_bufs(nf_ctx *ctx, const nf_flow_desc *desc, const nf_flow_desc *inner, long N, BaseBufs *bb) {
  const nf_base *b = flow_base(desc);
  const size_t es = esize(desc->dtype);
  const size_t in_need = ws_need_bound(ctx, inner, N);
  NF_TRY(nf_ws_reserve(ctx, in_need + base_extra_bytes(desc, N)));
  Carver cv((char *)ctx->ws + in_need);
  bb->x = cv.take<char>((size_t)N * desc->d * es);
  bb->z = b->kind == NF_BASE_DENSE ? cv.take<char>((size_t)N * desc->d * es) : nullptr;
  bb->corr = cv.take<char>((size_t)N * es);
  bb->tmp = cv.take<char>((size_t)N * es);
  bb->nb = nf_sum2_nblocks(N);
  bb->partial = cv.take<double>(bb->nb);
  bb->result = cv.take<double>(8);
  ctx->ws_guard = in_need;  // the inner entry point must stay in front of these buffers
  return NF_OK;
}
struct GuardReset {  // puts the enclosing wrapper's guard (or none) back on every exit path
  nf_ctx *ctx;
  size_t prev;
  ~GuardReset() { ctx->ws_guard = prev; }
};

static int base_draw(nf_ctx *ctx, const nf_flow_desc *desc, long N, uint64_t seed, uint64_t off, uint32_t stream_id, void *x,
                     void *logq_eps) {
  const nf_base *b = flow_base(desc);
  NF_TRY(nf_launch_base_sample(ctx, desc->dtype, desc->d, N, seed, off, stream_id, x, logq_eps));
  return nf_launch_base_unwhiten(ctx, desc->dtype, b->kind, desc->d, N, b->mu, b->scale, x);
}


struct CompBufs {
  char *y, *gbar, *logq, *ladj, *tmp, *lbar, *spare;
  double *partial_t, *partial_s, *result;
};
static int composite_bufs(nf_ctx *ctx, const nf_flow_desc *desc, long N, CompBufs *cb);
static int composite_chain(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, const void *theta, const void *x_in, long N,
                           void *y_out, void *ladj, CompBufs &cb);
static int elbo_forward_general_base(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                                     const void *xs, long N, uint64_t seed, uint64_t off, uint32_t stream_id,
                                     void *elbos_out, double *elbo_host);
static int loglikelihood_general_base(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *ys, long N,
                                      void *logliks_out, double *ll_host);
static int loglikelihood_composite(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *ys, long N,
                                   void *logliks_out, double *ll_host);
static int value_and_grad_composite(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                                    const void *xs, int64_t N_local, int64_t N_global, uint64_t seed, uint64_t sample_offset,
                                    uint32_t stream_id, void *out);
static int value_and_grad_general_base(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                                       const void *xs, int64_t N_local, int64_t N_global, uint64_t seed, uint64_t sample_offset,
                                       uint32_t stream_id, void *out);
static int fkl_general(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *ys, int64_t N_local,
                       int64_t N_global, void *out);
static size_t fkl_general_need(nf_ctx *ctx, const nf_flow_desc *desc, long N);

// ---- base distribution ---------------------------------------------------------------------
extern "C" int nf_base_sample_logpdf(nf_ctx *ctx, int32_t dtype, int32_t d, int64_t N, uint64_t seed,
                                     uint64_t sample_offset, uint32_t stream_id, void *x_out, void *logq_out) {
  if (!ctx || !x_out || d < 1 || N < 0) return NF_ERR_ARG;
  if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_F64) return NF_ERR_ARG;
  NF_HIP(hipSetDevice(ctx->device));
  return nf_launch_base_sample(ctx, dtype, d, N, seed, sample_offset, stream_id, x_out, logq_out);
}

extern "C" int nf_base_logpdf(nf_ctx *ctx, int32_t dtype, int32_t d, int64_t N, const void *x, void *logq_out) {
  if (!ctx || !x || !logq_out || d < 1 || N < 0) return NF_ERR_ARG;
  if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_F64) return NF_ERR_ARG;
  NF_HIP(hipSetDevice(ctx->device));
  return nf_launch_base_logpdf(ctx, dtype, d, N, x, logq_out);
}

// ---- transforms ----------------------------------------------------------------------------
// Coupling flows work on the TILED batch layout internally (nf_elementwise.hip); user-facing
// batches are converted at the API edge.  Element count of a tiled buffer:
static inline size_t tiled_elems(const nf_flow_desc *desc, long N) { return (size_t)((N + 31) / 32) * 32 * desc->d; }
// coupling flows on the tiled fp32 MFMA path; Float64 couplings and the Float32 shapes those kernels do
// not build (other than two hidden layers, wider NSF nets, other K) take the standard-layout route below
static inline bool is_coupling(const nf_flow_desc *desc) {
  if (desc->dtype != NF_DTYPE_F32) return false;
  if (desc->kind == NF_KIND_REALNVP) return nf_affine_supported(desc) || nf_wide_supported(desc);
  if (desc->kind == NF_KIND_NSF) return nf_rqs_supported(desc);
  return false;
}
static inline bool is_g64(const nf_flow_desc *desc) {
  return (desc->kind == NF_KIND_REALNVP || desc->kind == NF_KIND_NSF) && !is_coupling(desc);
}
// standard-layout flows: planar / radial / mean-field (nf_simple.hip) and Float64 couplings
static int flat_apply(nf_ctx *ctx, const nf_flow_desc *desc, int lo, int hi, bool inverse, const void *theta,
                      const void *x, long N, void *y, void *ladj) {
  if (desc->kind == NF_KIND_HAMILTONIAN) return nf_hf_apply(ctx, desc, lo, hi, inverse, theta, x, N, y, ladj);
  if (is_g64(desc))
    return nf_g64_apply(ctx, desc, lo, hi, inverse, theta, x, N, y, ladj);
  return nf_simple_apply(ctx, desc, lo, hi, inverse, theta, x, N, y, ladj);
}
static size_t flat_bwd_ws_bytes(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  if (desc->kind == NF_KIND_HAMILTONIAN) return nf_hf_bwd_ws_bytes(desc, N);
  return is_g64(desc) ? nf_g64_bwd_ws_bytes(desc, N) : nf_simple_bwd_ws_bytes(ctx, desc, N);
}
static int flat_bwd(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x, const void *ybar,
                    const void *lbar, double lbar_const, long N, void *xbar_out, void *gtheta_out, void *ws) {
  if (desc->kind == NF_KIND_HAMILTONIAN)
    return nf_hf_bwd(ctx, desc, theta, x, ybar, lbar, lbar_const, N, xbar_out, gtheta_out, ws);
  if (is_g64(desc))
    return nf_g64_bwd(ctx, desc, theta, x, ybar, lbar, lbar_const, N, xbar_out, gtheta_out, ws);
  return nf_simple_bwd(ctx, desc, theta, x, ybar, lbar, lbar_const, N, xbar_out, gtheta_out, ws, false);
}
static inline bool is_nsf(const nf_flow_desc *desc) { return desc->kind == NF_KIND_NSF; }
// Hamiltonian flows: ELBO targets describe x (the first d/2 coordinates); the momenta are standard normal
static inline int joint_dims(const nf_flow_desc *desc) { return desc->kind == NF_KIND_HAMILTONIAN ? desc->d / 2 : 0; }
// RealNVP shapes whose nets do not fit in LDS take the weight-streaming kernels
static inline bool is_wide(const nf_flow_desc *desc) {
  return desc->kind == NF_KIND_REALNVP && !nf_affine_supported(desc) && nf_wide_supported(desc);
}
static int coupling_pack(nf_ctx *ctx, const nf_flow_desc *desc, const float *theta) {
  ctx->wimg_owner = nullptr;  // whatever nf_elbo_step had cached is overwritten
  if (is_wide(desc)) return nf_wide_pack(ctx, desc, theta);
  return is_nsf(desc) ? nf_rqs_pack(ctx, desc, theta) : nf_affine_pack(ctx, desc, theta);
}
static int coupling_bwd_grid(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  if (is_wide(desc)) return 1;
  return is_nsf(desc) ? nf_rqs_bwd_grid(ctx, N) : nf_affine_bwd_grid(ctx, N);
}
// floats of reverse-pass workspace per workgroup slab (wide path: the whole stash + split-K area)
static long coupling_slab_floats(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  if (is_wide(desc)) return (long)nf_wide_bwd_ws_floats(ctx, desc, N);
  return is_nsf(desc) ? nf_rqs_slab_floats(desc) : nf_affine_slab_floats(desc);
}

// The ELBO forward of a training step fuses into ONE launch (draws + chain + target + partial sums)
// when the draws are in-library, the nets are LDS-resident and the target is the diagonal Gaussian.
static inline bool elbo_fusable(const nf_flow_desc *desc, const nf_target *target, const void *xs) {
  const bool resident = (desc->kind == NF_KIND_REALNVP && nf_affine_supported(desc) && nf_affine_fused_ok(desc)) ||
                        (desc->kind == NF_KIND_NSF && nf_rqs_supported(desc));
  return !xs && desc->dtype == NF_DTYPE_F32 && resident && target->kind == NF_TARGET_DIAGGAUSS && target->p0 && target->p1;
}
// the fused forward launch (draws + chain + target + ELBO partial sums) of the two LDS-resident coupling families
static int fused_chain_elbo(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, long N, uint64_t seed,
                            uint64_t off, uint32_t stream_id, float *yt, float *gt, double gscale, double *partial,
                            double pscale, float *stash = nullptr) {
  if (desc->kind == NF_KIND_NSF)  // `stash`: the spline tape (nf_rqs_tape_bytes)
    return nf_rqs_chain_elbo(ctx, desc, N, seed, off, stream_id, (const float *)target->p0, (const float *)target->p1, yt, gt,
                             gscale, partial, pscale, stash);
  return nf_affine_chain_elbo(ctx, desc, N, seed, off, stream_id, (const float *)target->p0, (const float *)target->p1, yt, gt,
                              gscale, partial, pscale, stash);
}
// The LDS-resident RealNVP training step keeps the forward's activations for the reverse pass (nf_coupling.hip,
// "activation stash") while they fit the budget: 46 KiB per 32-sample tile and coupling at d = 64 / hidden 64.  NF_AFFINE_STASH_MAX_MB (default 4096) bounds it; beyond, or with
// NF_AFFINE_NO_STASH set (A/B measurements), the reverse pass recomputes them (k_affine_bwd_all).
#define NF_STASH_TILE 32
static inline bool affine_stash_off(nf_ctx *ctx);
static size_t affine_stash_bytes(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  if (desc->kind != NF_KIND_REALNVP || desc->dtype != NF_DTYPE_F32 || !nf_affine_supported(desc)) return 0;
  static const size_t env_cap = [] {
    const char *e = std::getenv("NF_AFFINE_STASH_MAX_MB");
    return (size_t)(e ? std::atol(e) : 4096) << 20;
  }();
  // default policy (round 3): every LDS-resident shape WITH TWO HIDDEN LAYERS keeps the forward's activations -- the reverse
  // pass then differentiates the forward's own tape, as the reference's AD does.  The invertible-recompute kernel is the explicit
  // nf_ctx_set_stash_budget(0) mode only (for hidden width 32 it is 3 % faster, and it re-decides leaky-ReLU slopes on a
  // float32 reconstruction: DESIGN.md section 5).
  // EXCEPTION (round 5, stated here since round 6 -- ADVICE r5): nets with 1, 3 or 4 hidden layers (nf_deep.hip; also
  // "supported" by nf_affine_supported) have NO stash mode: nf_affine_stash_floats() is 0 for them, k_deep_bwd always
  // recomputes, and the stash budget has no effect.  Their gradient therefore carries the recompute's kink term
  // (tests/test_gpu_parity.py::test_deep_realnvp_step_at_eight_couplings_and_a_large_batch_against_oracle measures and bounds it
  // at 8 couplings x 8 197 samples); NF_DEEP_OFF=1 (read once per process) puts those shapes back on the layer-by-layer path,
  // whose reverse pass reads the forward's kept activations.
  if (affine_stash_off(ctx)) return 0;
  const size_t cap = ctx->stash_budget > 0 ? (size_t)ctx->stash_budget : env_cap;
  const size_t b = nf_affine_stash_floats(desc, N) * sizeof(float);
  return b <= cap ? b : 0;
}
// Batches whose stash exceeds the budget run the step in CHUNKS of samples (forward + reverse pass per chunk through one
// stash-sized buffer, every chunk's gradient slabs reduced together at the end): chunk size in samples, a multiple of
// 1024 tiles so that every chunk fills the chip evenly; N itself if one chunk does; 0 if the stash is off.
static long affine_stash_chunk(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  if (affine_stash_bytes(ctx, desc, N)) return N;
  // largest multiple of `unit` tiles whose stash passes the budget test (unit: 1024 tiles = 4 per wave of the chip, or
  // one workgroup's 4 tiles under a very small budget)
  const long unit = affine_stash_bytes(ctx, desc, 1024L * NF_STASH_TILE) ? 1024 : 4;
  long lo = 0, hi = (N / NF_STASH_TILE) / unit + 1;
  while (lo + 1 < hi) {
    const long mid = (lo + hi) / 2;
    if (affine_stash_bytes(ctx, desc, mid * unit * NF_STASH_TILE)) lo = mid; else hi = mid;
  }
  return lo * unit * NF_STASH_TILE;
}
// floats of gradient slabs a chunked step writes: every chunk's workgroups leave one slab each
static int coupling_bwd_grid(nf_ctx *ctx, const nf_flow_desc *desc, long N);
static size_t chunked_slab_floats(nf_ctx *ctx, const nf_flow_desc *desc, long N, long chunk, long stride) {
  if (!chunk || chunk >= N) return (size_t)coupling_bwd_grid(ctx, desc, N) * stride;
  const long full = N / chunk, rem = N - full * chunk;
  return ((size_t)full * coupling_bwd_grid(ctx, desc, chunk) + (rem ? coupling_bwd_grid(ctx, desc, rem) : 0)) * stride;
}
extern "C" int nf_ctx_set_stash_budget(nf_ctx *ctx, int64_t max_bytes) {
  if (!ctx) return NF_ERR_ARG;
  ctx->stash_budget = max_bytes < 0 ? -1 : max_bytes;
  return NF_OK;
}
// bytes of the spline tape of an MFMA-path NSF flow (0 for everything else)
static inline size_t rqs_tape_b(const nf_flow_desc *desc, long N) {
  return (desc->kind == NF_KIND_NSF && desc->dtype == NF_DTYPE_F32 && nf_rqs_supported(desc)) ? carve_bytes(nf_rqs_tape_bytes(desc, N)) : 0;
}
static long fused_chain_grid(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  return desc->kind == NF_KIND_NSF ? nf_rqs_chain_grid(ctx, N) : nf_affine_chain_grid(ctx, N);
}

// all couplings (or one, if k_only >= 0) in execution order / inverse order, in place on `xt`
// rqs_tape (NSF only, optional): the spline tape a later reverse pass of this chain needs (rqs_tape_b bytes)
static int coupling_chain_tiled(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, const float *theta, float *xt,
                                long N, float *ladj, int k_only, void *rqs_tape = nullptr) {
  NF_TRY(coupling_pack(ctx, desc, theta));
  if (is_nsf(desc)) return nf_rqs_chain(ctx, desc, inverse, xt, N, ladj, k_only, rqs_tape);
  if (is_wide(desc)) {
    if (k_only >= 0) return nf_wide_apply(ctx, desc, k_only, inverse, xt, N, ladj, 0);
    const int nc = 2 * desc->nlayers;
    for (int s = 0; s < nc; ++s)  // forward applies the LAST flat coupling first
      NF_TRY(nf_wide_apply(ctx, desc, inverse ? s : nc - 1 - s, inverse, xt, N, ladj, s > 0));
    return NF_OK;
  }
  if (k_only >= 0) return nf_affine_apply(ctx, desc, k_only, inverse, theta, xt, N, xt, ladj, 0);
  return nf_affine_chain(ctx, desc, inverse, xt, N, ladj);
}

static int composite_apply(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, int layer, const void *theta,
                           const void *x_in, long N, void *y_out, void *ladj);

// standard-layout in/out wrapper used by nf_flow_fwd / nf_flow_inv / nf_layer_apply
static int apply_std(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, int layer, const void *theta,
                     const void *x_in, long N, void *y_out, void *ladj) {
  if (N == 0) return NF_OK;
  if (is_composite(desc)) return composite_apply(ctx, desc, inverse, layer, theta, x_in, N, y_out, ladj);
  if (is_coupling(desc)) {
    NF_TRY(nf_ws_reserve(ctx, carve_bytes(tiled_elems(desc, N) * 4)));
    float *xt = (float *)ctx->ws;
    NF_TRY(nf_launch_layout_convert(ctx, desc->d, N, (const float *)x_in, xt, 1));
    NF_TRY(coupling_chain_tiled(ctx, desc, inverse, (const float *)theta, xt, N, (float *)ladj, layer));
    return nf_launch_layout_convert(ctx, desc->d, N, xt, (float *)y_out, 0);
  }
  const int nl = nf_layer_count(desc);
  if (layer >= 0) return flat_apply(ctx, desc, layer, layer + 1, inverse, theta, x_in, N, y_out, ladj);
  return flat_apply(ctx, desc, 0, nl, inverse, theta, x_in, N, y_out, ladj);
}

extern "C" int nf_flow_fwd(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x_in, int64_t N,
                           void *y_out, void *ladj_out) {
  if (!ctx || !theta || !x_in || !y_out || !ladj_out || N < 0) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  return apply_std(ctx, desc, false, -1, theta, x_in, N, y_out, ladj_out);
}

extern "C" int nf_flow_inv(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *y_in, int64_t N,
                           void *x_out, void *ladj_out) {
  if (!ctx || !theta || !y_in || !x_out || !ladj_out || N < 0) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  return apply_std(ctx, desc, true, -1, theta, y_in, N, x_out, ladj_out);
}

// rand(rng, flow, n) / _device_specific_rand(rng, flow, n): N base draws (Philox, as nf_base_sample_logpdf with the
// same seed / offset / stream) pushed through the transform
extern "C" int nf_flow_rand(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, int64_t N, uint64_t seed,
                            uint64_t sample_offset, uint32_t stream_id, void *y_out) {
  if (!ctx || !theta || !y_out || N < 0) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  if (N == 0) return NF_OK;
  if (flow_base(desc)) {  // x = mu + L eps into y_out, then the transform in place (log-det into scratch)
    nf_flow_desc inner = *desc;
    inner.base = nullptr;
    BaseBufs bb;
    const size_t prev_guard = ctx->ws_guard;
    NF_TRY(base_bufs(ctx, desc, &inner, N, &bb));
    GuardReset gr{ctx, prev_guard};
    NF_TRY(base_draw(ctx, desc, N, seed, sample_offset, stream_id, y_out, nullptr));
    return apply_std(ctx, &inner, false, -1, theta, y_out, N, y_out, bb.tmp);
  }
  if (is_composite(desc)) {  // draws into y_out, then the chain in place
    CompBufs cb;
    const size_t prev_guard = ctx->ws_guard;
    NF_TRY(composite_bufs(ctx, desc, N, &cb));
    GuardReset gr{ctx, prev_guard};
    NF_TRY(nf_launch_base_sample(ctx, desc->dtype, desc->d, N, seed, sample_offset, stream_id, y_out, nullptr));
    return composite_chain(ctx, desc, false, theta, y_out, N, y_out, cb.ladj, cb);
  }
  if (is_coupling(desc)) {
    NF_TRY(nf_ws_reserve(ctx, carve_bytes(tiled_elems(desc, N) * 4) + carve_bytes((size_t)N * 4)));
    Carver cv(ctx->ws);
    float *xt = cv.take<float>(tiled_elems(desc, N));
    float *ladj = cv.take<float>((size_t)N);
    NF_TRY(nf_launch_base_sample_tiled(ctx, desc->d, N, seed, sample_offset, stream_id, xt, nullptr));
    NF_TRY(coupling_chain_tiled(ctx, desc, false, (const float *)theta, xt, N, ladj, -1));
    return nf_launch_layout_convert(ctx, desc->d, N, xt, (float *)y_out, 0);
  }
  if (!is_g64(desc) && desc->kind != NF_KIND_HAMILTONIAN)
    return nf_simple_rand(ctx, desc, theta, N, seed, sample_offset, stream_id, y_out);
  NF_TRY(nf_ws_reserve(ctx, carve_bytes((size_t)N * esize(desc->dtype))));
  NF_TRY(nf_launch_base_sample(ctx, desc->dtype, desc->d, N, seed, sample_offset, stream_id, y_out, nullptr));
  return flat_apply(ctx, desc, 0, nf_layer_count(desc), false, theta, y_out, N, y_out, ctx->ws);
}

extern "C" int nf_layer_apply(nf_ctx *ctx, const nf_flow_desc *desc, int32_t layer, int32_t inverse, const void *theta,
                              const void *x_in, int64_t N, void *y_out, void *ladj_out) {
  if (!ctx || !theta || !x_in || !y_out || !ladj_out || N < 0) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  if (layer < 0 || layer >= nf_layer_count(desc)) return NF_ERR_ARG;
  NF_HIP(hipSetDevice(ctx->device));
  return apply_std(ctx, desc, inverse != 0, layer, theta, x_in, N, y_out, ladj_out);
}

// reverse pass over the coupling chain (tiled buffers).  `state` holds the flow OUTPUT on entry and
// the flow INPUT on exit (invertible recompute); `gbar` holds ybar on entry and xbar on exit.
// lpart/nlpart/lout: optional loss partials for the slab-reduction kernel to finish (resident RealNVP only)
static int realnvp_bwd(nf_ctx *ctx, const nf_flow_desc *desc, const float *theta, float *state, float *gbar,
                       const float *lbar, float lbar_const, long N, float *slab, int grid, float *g_out,
                       const double *lpart = nullptr, int nlpart = 0, float *lout = nullptr, void *rqs_tape = nullptr) {
  if (is_wide(desc)) return nf_wide_bwd(ctx, desc, state, gbar, lbar, lbar_const, N, slab, g_out);
  const long stride = coupling_slab_floats(ctx, desc, N);
  const int nc = 2 * desc->nlayers;
  if (!is_nsf(desc)) {  // LDS-resident RealNVP: every coupling in one launch
    NF_TRY(nf_affine_bwd_all(ctx, desc, state, gbar, lbar, lbar_const, N, slab, stride, grid));
    return nf_affine_reduce_slabs(ctx, desc, slab, grid, g_out, lpart, nlpart, lout);
  }
  // Neural spline couplings: one launch per coupling.  The single-launch form EXISTED through round 5 (k_rqs_bwd_all, the structure of
  // k_affine_bwd_all; NF_RQS_BWD_FUSED=1 selects it) but MEASURED SLOWER on this kernel, A/B on one box: 1329 us for
  // the 8 couplings against 8 x 159.5 us = 1276 us, step 1.729 vs 1.657 ms -- the kernel sits at the register wall
  // (256 VGPR + 256 AGPR), and the outer coupling loop costs it 40 more bytes of scratch spills per lane than the
  // launch gaps it saves (profiles/r2_cfg3_fused_vs_split.txt).
  // (round 6: k_rqs_bwd_all and its NF_RQS_BWD_FUSED switch are gone with the measurement above)
  for (int k = 0; k < nc; ++k) NF_TRY(nf_rqs_bwd(ctx, desc, k, state, gbar, lbar, lbar_const, N, slab, stride, grid, false, rqs_tape));
  return nf_rqs_reduce_slabs(ctx, desc, slab, grid, g_out);
}

// reverse pass of the INVERSE coupling chain on tiled buffers (forward-KL training): `state` holds
// T^-1(data) on entry and the data on exit, `gbar` the cotangent of z; couplings in forward execution order
static inline bool coupling_inv_bwd_tiled(const nf_flow_desc *desc) { return is_coupling(desc); }
static int coupling_inv_bwd(nf_ctx *ctx, const nf_flow_desc *desc, const float *theta, float *state, float *gbar,
                            float lbar_const, long N, float *slab, int grid, float *g_out, void *rqs_tape = nullptr) {
  (void)theta;
  if (is_wide(desc)) return nf_wide_bwd(ctx, desc, state, gbar, nullptr, lbar_const, N, slab, g_out, true);
  if (desc->kind == NF_KIND_REALNVP && nf_affine_supported(desc)) {
    const long stride = coupling_slab_floats(ctx, desc, N);
    NF_TRY(nf_affine_bwd_all(ctx, desc, state, gbar, nullptr, lbar_const, N, slab, stride, grid, true));
    return nf_affine_reduce_slabs(ctx, desc, slab, grid, g_out);
  }
  if (desc->kind == NF_KIND_NSF && nf_rqs_supported(desc)) {
    const long stride = coupling_slab_floats(ctx, desc, N);
    for (int k = 2 * desc->nlayers - 1; k >= 0; --k)  // forward execution order (one launch per coupling: see realnvp_bwd)
      NF_TRY(nf_rqs_bwd(ctx, desc, k, state, gbar, nullptr, lbar_const, N, slab, stride, grid, true, rqs_tape));
    return nf_rqs_reduce_slabs(ctx, desc, slab, grid, g_out);
  }
  return NF_ERR_UNSUPPORTED;
}

static long seg_theta_off(const nf_flow_desc *desc, int s);
// ---- the tape: what a forward pass keeps for its pullback -------------------------------------------------------
// The reference differentiates the forward's own tape (Zygote on src/objectives/elbo.jl:65-70 under
// src/optimize.jl:12-14; MonotonicSplines' rrules, test/ad.jl:126-127).  nf_flow_fwd_keep writes that tape into
// caller-owned memory and nf_flow_bwd_kept consumes it, so the pullback uses the forward's own activations and
// leaky-ReLU slopes -- nothing is re-derived by inverting the flow in float32.  Per family:
//   LDS-resident RealNVP        the activation stash of the training step (k_affine_chain<STASH> -> k_affine_bwd_stashed);
//                               with nf_ctx_set_stash_budget(0): the tiled flow output (invertible recompute, explicit opt-in)
//   weight-streaming RealNVP    k_wide_apply's forward stash + the tiled flow output (k_wide_bwd_stashed, k_wide_dw)
//   neural spline couplings     the tiled flow output (the reverse kernel recomputes from it)
//   planar / radial / mean-field / Float64 couplings / Hamiltonian: the flow input (their reverse pass re-runs the
//                               forward from x in the element type -- the same arithmetic, hence the same activations)
//   compositions                the segments' tapes, concatenated in flat order
// The tape's size and content are a function of (context settings, desc, N) only: both calls must see the same.
enum TapeKind { TAPE_X, TAPE_AFFINE_STASH, TAPE_TILED_Y, TAPE_WIDE };
static inline bool affine_stash_off(nf_ctx *ctx) {
  static const bool env_off = std::getenv("NF_AFFINE_NO_STASH") != nullptr;
  return ctx->stash_budget == 0 || (ctx->stash_budget < 0 && env_off);
}
// The activation stash is kept only while it passes the SAME budget test as the training step's (nf_ctx_set_stash_budget, or
// NF_AFFINE_STASH_MAX_MB: default 4 GiB): a batch whose stash would exceed it keeps the tiled flow output instead and its
// pullback recomputes (these entry points used to size the stash for the whole batch whatever the budget said --
// 12 GiB at N = 1 M).  Forward and pullback of one tape must see the same budget.
static TapeKind tape_kind(nf_ctx *ctx, const nf_flow_desc *g, long N) {
  if (!is_coupling(g)) return TAPE_X;
  if (is_wide(g)) return TAPE_WIDE;
  if (is_nsf(g)) return TAPE_TILED_Y;
  return affine_stash_bytes(ctx, g, N) ? TAPE_AFFINE_STASH : TAPE_TILED_Y;
}
static size_t tape_seg_bytes(nf_ctx *ctx, const nf_flow_desc *g, long N) {
  switch (tape_kind(ctx, g, N)) {
    case TAPE_X: return carve_bytes((size_t)N * g->d * esize(g->dtype));
    case TAPE_AFFINE_STASH: return carve_bytes(nf_affine_stash_floats(g, N) * 4);
    case TAPE_TILED_Y: return carve_bytes(tiled_elems(g, N) * 4) + rqs_tape_b(g, N);  // spline couplings: + bins and xi
    case TAPE_WIDE: return carve_bytes(nf_wide_fwd_stash_floats(ctx, g, N) * 4) + carve_bytes(tiled_elems(g, N) * 4);
  }
  return 0;
}
static size_t tape_bytes_of(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  if (!is_composite(desc)) return tape_seg_bytes(ctx, desc, N);
  size_t tot = 0;
  for (int s = 0; s < desc->nsegments; ++s) tot += tape_seg_bytes(ctx, &desc->segments[s], N);
  return tot;
}
// intermediates (front of the context workspace) of the two passes
static size_t tape_fwd_need_seg(nf_ctx *, const nf_flow_desc *g, long N) {
  return is_coupling(g) ? carve_bytes(tiled_elems(g, N) * 4) : 0;
}
static size_t tape_bwd_need_seg(nf_ctx *ctx, const nf_flow_desc *g, long N) {
  const size_t tb = is_coupling(g) ? carve_bytes(tiled_elems(g, N) * 4) : 0;
  switch (tape_kind(ctx, g, N)) {
    case TAPE_X: return flat_bwd_ws_bytes(ctx, g, N);
    case TAPE_AFFINE_STASH: return tb + carve_bytes((size_t)coupling_bwd_grid(ctx, g, N) * coupling_slab_floats(ctx, g, N) * 4);
    case TAPE_TILED_Y: return 2 * tb + carve_bytes((size_t)coupling_bwd_grid(ctx, g, N) * coupling_slab_floats(ctx, g, N) * 4);
    case TAPE_WIDE: return 2 * tb + carve_bytes(nf_wide_train_scratch_floats(ctx, g, N) * 4);
  }
  return 0;
}
// a composition keeps a log-det scratch vector and the block partials nf_launch_sum2 insists on behind its segments' needs
static size_t tape_comp_extra(const nf_flow_desc *desc, long N) {
  return carve_bytes((size_t)N * esize(desc->dtype)) + carve_bytes((size_t)nf_sum2_nblocks(N) * 8);
}
static size_t tape_need(nf_ctx *ctx, const nf_flow_desc *desc, long N, bool bwd) {
  if (!is_composite(desc)) return bwd ? tape_bwd_need_seg(ctx, desc, N) : tape_fwd_need_seg(ctx, desc, N);
  size_t m = 0;
  for (int s = 0; s < desc->nsegments; ++s) {
    const nf_flow_desc *g = &desc->segments[s];
    const size_t v = bwd ? tape_bwd_need_seg(ctx, g, N) : tape_fwd_need_seg(ctx, g, N);
    if (v > m) m = v;
  }
  return m + (bwd ? 0 : tape_comp_extra(desc, N));
}

// forward of ONE homogeneous segment, leaving its tape; y_out may alias x_in
static int tape_fwd_seg(nf_ctx *ctx, const nf_flow_desc *g, const void *theta, const void *x_in, long N, void *y_out, void *ladj,
                        void *tape) {
  const TapeKind tk = tape_kind(ctx, g, N);
  if (tk == TAPE_X) {
    NF_HIP(hipMemcpyAsync(tape, x_in, (size_t)N * g->d * esize(g->dtype), hipMemcpyDeviceToDevice, ctx->stream));
    return flat_apply(ctx, g, 0, nf_layer_count(g), false, theta, x_in, N, y_out, ladj);
  }
  const size_t te = tiled_elems(g, N);
  NF_TRY(nf_ws_reserve(ctx, carve_bytes(te * 4)));
  float *xt = (float *)ctx->ws;
  NF_TRY(nf_launch_layout_convert(ctx, g->d, N, (const float *)x_in, xt, 1));
  NF_TRY(coupling_pack(ctx, g, (const float *)theta));
  if (tk == TAPE_AFFINE_STASH) {
    NF_TRY(nf_affine_chain(ctx, g, false, xt, N, (float *)ladj, (float *)tape));
  } else if (tk == TAPE_WIDE) {
    NF_TRY(nf_wide_train_forward(ctx, g, xt, N, (float *)ladj, (float *)tape));
    char *ty = (char *)tape + carve_bytes(nf_wide_fwd_stash_floats(ctx, g, N) * 4);
    NF_HIP(hipMemcpyAsync(ty, xt, te * 4, hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    NF_TRY(coupling_chain_tiled(ctx, g, false, (const float *)theta, xt, N, (float *)ladj, -1,
                                rqs_tape_b(g, N) ? (char *)tape + carve_bytes(te * 4) : nullptr));
    NF_HIP(hipMemcpyAsync(tape, xt, te * 4, hipMemcpyDeviceToDevice, ctx->stream));
  }
  return nf_launch_layout_convert(ctx, g->d, N, xt, (float *)y_out, 0);
}

// pullback of ONE homogeneous segment from its tape; xbar_out may alias ybar.  lbar == nullptr: every sample's
// log-det cotangent is lbar_const.  The tape is left intact (a pullback may be called more than once).
static int tape_bwd_seg(nf_ctx *ctx, const nf_flow_desc *g, const void *theta, const void *tape, const void *ybar,
                        const void *lbar, double lbar_const, long N, void *xbar_out, void *gtheta_out) {
  const TapeKind tk = tape_kind(ctx, g, N);
  if (tk == TAPE_X) {
    NF_TRY(nf_ws_reserve(ctx, flat_bwd_ws_bytes(ctx, g, N)));
    return flat_bwd(ctx, g, theta, tape, ybar, lbar, lbar_const, N, xbar_out, gtheta_out, ctx->ws);
  }
  const size_t te = tiled_elems(g, N);
  NF_TRY(nf_ws_reserve(ctx, tape_bwd_need_seg(ctx, g, N)));
  Carver cv(ctx->ws);
  float *gt = cv.take<float>(te);
  NF_TRY(coupling_pack(ctx, g, (const float *)theta));
  NF_TRY(nf_launch_layout_convert(ctx, g->d, N, (const float *)ybar, gt, 1));
  if (tk == TAPE_AFFINE_STASH) {
    const int grid = coupling_bwd_grid(ctx, g, N);
    const long stride = coupling_slab_floats(ctx, g, N);
    float *slab = cv.take<float>((size_t)grid * stride);
    NF_TRY(nf_affine_bwd_stashed(ctx, g, (float *)const_cast<void *>(tape), gt, (const float *)lbar, (float)lbar_const, N, slab,
                                 stride, grid));
    NF_TRY(nf_affine_reduce_slabs(ctx, g, slab, grid, (float *)gtheta_out));
  } else if (tk == TAPE_WIDE) {
    float *state = cv.take<float>(te);
    float *scratch = cv.take<float>(nf_wide_train_scratch_floats(ctx, g, N));
    const char *ty = (const char *)tape + carve_bytes(nf_wide_fwd_stash_floats(ctx, g, N) * 4);
    NF_HIP(hipMemcpyAsync(state, ty, te * 4, hipMemcpyDeviceToDevice, ctx->stream));
    NF_TRY(nf_wide_train_backward(ctx, g, state, gt, (const float *)lbar, (float)lbar_const, N,
                                  (float *)const_cast<void *>(tape), (float *)gtheta_out, scratch));
  } else {
    float *state = cv.take<float>(te);
    const int grid = coupling_bwd_grid(ctx, g, N);
    float *slab = cv.take<float>((size_t)grid * coupling_slab_floats(ctx, g, N));
    NF_HIP(hipMemcpyAsync(state, tape, te * 4, hipMemcpyDeviceToDevice, ctx->stream));
    NF_TRY(realnvp_bwd(ctx, g, (const float *)theta, state, gt, (const float *)lbar, (float)lbar_const, N, slab, grid,
                       (float *)gtheta_out, nullptr, 0, nullptr,
                       rqs_tape_b(g, N) ? (char *)const_cast<void *>(tape) + carve_bytes(te * 4) : nullptr));
  }
  if (!xbar_out) return NF_OK;  // the caller wants the parameter gradient only (the base draws are constants, elbo.jl:94)
  return nf_launch_layout_convert(ctx, g->d, N, gt, (float *)xbar_out, 0);
}

// whole flow (a composition chains its segments through y_out in place; the LAST segment is applied first)
static int tape_fwd(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x_in, long N, void *y_out,
                    void *ladj, void *tape) {
  if (!is_composite(desc)) return tape_fwd_seg(ctx, desc, theta, x_in, N, y_out, ladj, tape);
  const size_t es = esize(desc->dtype);
  const int ns = desc->nsegments;
  size_t in_need = 0, toff[65];
  toff[0] = 0;
  for (int s = 0; s < ns; ++s) {
    const size_t v = tape_fwd_need_seg(ctx, &desc->segments[s], N);
    if (v > in_need) in_need = v;
    toff[s + 1] = toff[s] + tape_seg_bytes(ctx, &desc->segments[s], N);
  }
  const size_t prev_guard = ctx->ws_guard;
  NF_TRY(nf_ws_reserve(ctx, in_need + tape_comp_extra(desc, N)));
  GuardReset gr{ctx, prev_guard};
  Carver cv((char *)ctx->ws + in_need);
  char *tmp = cv.take<char>((size_t)N * es);
  double *partial = cv.take<double>(nf_sum2_nblocks(N));
  if (in_need) ctx->ws_guard = in_need;
  const void *cur = x_in;
  for (int i = 0; i < ns; ++i) {
    const int s = ns - 1 - i;
    const char *th = (const char *)theta + (size_t)seg_theta_off(desc, s) * es;
    NF_TRY(tape_fwd_seg(ctx, &desc->segments[s], th, cur, N, y_out, i == 0 ? ladj : (void *)tmp, (char *)tape + toff[s]));
    if (i > 0) NF_TRY(nf_launch_sum2(ctx, desc->dtype, N, ladj, tmp, ladj, partial, 0.0));
    cur = y_out;
  }
  return NF_OK;
}

static int tape_bwd(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *tape, const void *ybar,
                    const void *lbar, double lbar_const, long N, void *xbar_out, void *gtheta_out) {
  if (!is_composite(desc)) return tape_bwd_seg(ctx, desc, theta, tape, ybar, lbar, lbar_const, N, xbar_out, gtheta_out);
  const size_t es = esize(desc->dtype);
  size_t toff = 0;
  const void *cur = ybar;
  for (int s = 0; s < desc->nsegments; ++s) {  // flat order = last applied first
    const nf_flow_desc *g = &desc->segments[s];
    const size_t off = (size_t)seg_theta_off(desc, s) * es;
    NF_TRY(tape_bwd_seg(ctx, g, (const char *)theta + off, (const char *)tape + toff, cur, lbar, lbar_const, N, xbar_out,
                        (char *)gtheta_out + off));
    toff += tape_seg_bytes(ctx, g, N);
    cur = xbar_out;
  }
  return NF_OK;
}

extern "C" int64_t nf_tape_bytes(nf_ctx *ctx, const nf_flow_desc *desc, int64_t N) {
  if (!ctx || N < 0) return NF_ERR_ARG;
  const int st = check_desc(desc);
  if (st != NF_OK) return st;
  return (int64_t)tape_bytes_of(ctx, desc, N > 0 ? N : 1);
}

extern "C" int nf_flow_fwd_keep(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x_in, int64_t N,
                                void *y_out, void *ladj_out, void *tape, size_t tape_bytes) {
  if (!ctx || !theta || !x_in || !y_out || !ladj_out || !tape || N < 0 || ((uintptr_t)tape & 255)) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  if (N == 0) return NF_OK;
  if (tape_bytes < tape_bytes_of(ctx, desc, N)) return NF_ERR_WORKSPACE;
  return tape_fwd(ctx, desc, theta, x_in, N, y_out, ladj_out, tape);
}

extern "C" int nf_flow_bwd_kept(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *tape, size_t tape_bytes,
                                const void *ybar, const void *lbar, int64_t N, void *xbar_out, void *gtheta_out) {
  if (!ctx || !theta || !tape || !ybar || !lbar || !gtheta_out || N < 0 || ((uintptr_t)tape & 255)) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  // xbar_out == NULL (parameter gradient only) is accepted where the cotangent lives in the tiled layout anyway: single-
  // family coupling flows on the MFMA kernels; it saves the layout conversion of a d x N matrix nobody reads
  if (!xbar_out && (is_composite(desc) || !is_coupling(desc))) return NF_ERR_ARG;
  NF_HIP(hipSetDevice(ctx->device));
  if (N == 0) return nf_launch_fill(ctx, desc->dtype, gtheta_out, nf_param_count(desc), 0.0);
  if (tape_bytes < tape_bytes_of(ctx, desc, N)) return NF_ERR_WORKSPACE;
  return tape_bwd(ctx, desc, theta, tape, ybar, lbar, 0.0, N, xbar_out, gtheta_out);
}

// The pullback for callers that kept only x: the forward is run again FROM x with its tape in the context workspace
// (not reconstructed from y -- y is accepted for the signature of the rrule and not read), then the tape is pulled back.
static size_t flow_bwd_need(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  const size_t a = tape_need(ctx, desc, N, false), b = tape_need(ctx, desc, N, true);
  return (a > b ? a : b) + tape_bytes_of(ctx, desc, N) + carve_bytes((size_t)N * desc->d * esize(desc->dtype)) +
         carve_bytes((size_t)N * esize(desc->dtype));
}
extern "C" int nf_flow_bwd(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x, const void *y,
                           const void *ybar, const void *lbar, int64_t N, void *xbar_out, void *gtheta_out) {
  if (!ctx || !theta || !x || !y || !ybar || !lbar || !xbar_out || !gtheta_out || N < 0) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  const long P = nf_param_count(desc);
  if (N == 0) return nf_launch_fill(ctx, desc->dtype, gtheta_out, P, 0.0);
  const size_t es = esize(desc->dtype);
  const size_t a = tape_need(ctx, desc, N, false), b = tape_need(ctx, desc, N, true);
  const size_t in_need = a > b ? a : b;
  const size_t prev_guard = ctx->ws_guard;
  NF_TRY(nf_ws_reserve(ctx, flow_bwd_need(ctx, desc, N)));
  GuardReset gr{ctx, prev_guard};
  Carver cv((char *)ctx->ws + in_need);
  char *tape = cv.take<char>(tape_bytes_of(ctx, desc, N));
  char *yscr = cv.take<char>((size_t)N * desc->d * es);
  char *lscr = cv.take<char>((size_t)N * es);
  if (in_need) ctx->ws_guard = in_need;
  NF_TRY(tape_fwd(ctx, desc, theta, x, N, yscr, lscr, tape));
  return tape_bwd(ctx, desc, theta, tape, ybar, lbar, 0.0, N, xbar_out, gtheta_out);
}

// ---- targets -------------------------------------------------------------------------------
extern "C" int nf_target_logp(nf_ctx *ctx, int32_t dtype, const nf_target *target, int32_t d, int64_t N, const void *y,
                              void *logp_out, void *grad_out) {
  if (!ctx || !target || !y || d < 1 || N < 0) return NF_ERR_ARG;
  if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_F64) return NF_ERR_ARG;
  NF_HIP(hipSetDevice(ctx->device));
  return nf_launch_target(ctx, dtype, target, d, N, y, nullptr, nullptr, logp_out, grad_out, 1.0, nullptr, nullptr, 0.0, 0);
}

// ---- objectives ----------------------------------------------------------------------------
static int elbo_forward_composite(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                                  const void *xs, long N, uint64_t seed, uint64_t off, uint32_t stream_id, void *elbos_out,
                                  double *elbo_host);
static int elbo_forward(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                        const void *xs, long N, uint64_t seed, uint64_t off, uint32_t stream_id, void *elbos_out,
                        double *elbo_host) {
  if (is_composite(desc)) return elbo_forward_composite(ctx, desc, target, theta, xs, N, seed, off, stream_id, elbos_out, elbo_host);
  const size_t es = esize(desc->dtype);
  const bool cp = is_coupling(desc);
  const long nb = cp ? nf_target_tiled_nblocks(N) : nf_target_nblocks(N);
  const long nb_alloc = nb < ctx->num_cu ? ctx->num_cu : nb;  // the fused forward leaves one partial per workgroup
  const size_t xe = cp ? tiled_elems(desc, N) : (size_t)N * desc->d;
  const size_t need = carve_bytes(xe * es) + 2 * carve_bytes((size_t)N * es) + carve_bytes((size_t)nb_alloc * 8) + carve_bytes(64);
  NF_TRY(nf_ws_reserve(ctx, need));
  Carver cv(ctx->ws);
  char *x = cv.take<char>(xe * es);
  char *logq = cv.take<char>((size_t)N * es);
  char *ladj = cv.take<char>((size_t)N * es);
  double *partial = cv.take<double>(nb_alloc);
  double *result = cv.take<double>(8);
  if (cp && !elbos_out && elbo_fusable(desc, target, xs)) {
    NF_TRY(coupling_pack(ctx, desc, (const float *)theta));
    NF_TRY(fused_chain_elbo(ctx, desc, target, N, seed, off, stream_id, (float *)x, nullptr, 0.0, partial, 1.0 / (double)N));
    NF_TRY(nf_launch_finish_sum(ctx, partial, fused_chain_grid(ctx, desc, N), 0, result, nullptr, nullptr));
    return read_scalar(ctx, result, elbo_host);
  }
  if (cp) {
    float *xt = (float *)x;
    if (xs) {
      NF_TRY(nf_launch_layout_convert(ctx, desc->d, N, (const float *)xs, xt, 1));
      NF_TRY(nf_launch_base_logpdf_tiled(ctx, desc->d, N, xt, (float *)logq));
    } else {
      NF_TRY(nf_launch_base_sample_tiled(ctx, desc->d, N, seed, off, stream_id, xt, (float *)logq));
    }
    NF_TRY(coupling_chain_tiled(ctx, desc, false, (const float *)theta, xt, N, (float *)ladj, -1));
    NF_TRY(nf_launch_target_tiled(ctx, target, desc->d, N, xt, (const float *)logq, (const float *)ladj, nullptr, 0.0,
                                  (float *)elbos_out, partial, 1.0 / (double)N));
  } else if (!elbos_out && !is_g64(desc) && desc->kind != NF_KIND_HAMILTONIAN) {
    // planar / radial / mean-field, value only: draws (or xs), chain, target and ELBO sums in one launch
    NF_TRY(nf_target_check(target, desc->d));
    long np = 0;
    NF_TRY(nf_simple_elbo_forward(ctx, desc, target, theta, xs, N, seed, off, stream_id, nullptr, 0.0, partial, 1.0 / (double)N,
                                  nullptr, &np));
    NF_TRY(nf_launch_finish_sum(ctx, partial, np, 0, result, nullptr, nullptr));
    return read_scalar(ctx, result, elbo_host);
  } else {
    if (xs) {
      NF_HIP(hipMemcpyAsync(x, xs, (size_t)N * desc->d * es, hipMemcpyDeviceToDevice, ctx->stream));
      NF_TRY(nf_launch_base_logpdf(ctx, desc->dtype, desc->d, N, x, logq));
    } else {
      NF_TRY(nf_launch_base_sample(ctx, desc->dtype, desc->d, N, seed, off, stream_id, x, logq));
    }
    NF_TRY(flat_apply(ctx, desc, 0, nf_layer_count(desc), false, theta, x, N, x, ladj));
    NF_TRY(nf_launch_target(ctx, desc->dtype, target, desc->d, N, x, logq, ladj, nullptr, nullptr, 0.0, elbos_out,
                            partial, 1.0 / (double)N, joint_dims(desc)));
  }
  NF_TRY(nf_launch_finish_sum(ctx, partial, nb, 0, result, nullptr, nullptr));
  return read_scalar(ctx, result, elbo_host);
}

extern "C" int nf_elbo_batch(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                             const void *xs, int64_t N, void *elbos_out, double *elbo_host) {
  if (!ctx || !target || !theta || !xs || !elbo_host || N < 1) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  if (flow_base(desc)) return elbo_forward_general_base(ctx, desc, target, theta, xs, N, 0, 0, 0, elbos_out, elbo_host);
  return elbo_forward(ctx, desc, target, theta, xs, N, 0, 0, 0, elbos_out, elbo_host);
}

extern "C" int nf_elbo_batch_rng(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                                 int64_t N, uint64_t seed, uint64_t sample_offset, uint32_t stream_id,
                                 double *elbo_host) {
  if (!ctx || !target || !theta || !elbo_host || N < 1) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  if (flow_base(desc))
    return elbo_forward_general_base(ctx, desc, target, theta, nullptr, N, seed, sample_offset, stream_id, nullptr, elbo_host);
  return elbo_forward(ctx, desc, target, theta, nullptr, N, seed, sample_offset, stream_id, nullptr, elbo_host);
}

extern "C" int nf_loglikelihood(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *ys, int64_t N,
                                void *logliks_out, double *ll_host) {
  if (!ctx || !theta || !ys || !ll_host || N < 1) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  if (flow_base(desc)) return loglikelihood_general_base(ctx, desc, theta, ys, N, logliks_out, ll_host);
  if (is_composite(desc)) return loglikelihood_composite(ctx, desc, theta, ys, N, logliks_out, ll_host);
  const size_t es = esize(desc->dtype);
  const bool cp = is_coupling(desc);
  const long nb = nf_sum2_nblocks(N);
  const size_t xe = cp ? tiled_elems(desc, N) : (size_t)N * desc->d;
  const size_t need = carve_bytes(xe * es) + 2 * carve_bytes((size_t)N * es) + carve_bytes((size_t)nb * 8) + carve_bytes(64);
  NF_TRY(nf_ws_reserve(ctx, need));
  Carver cv(ctx->ws);
  char *x = cv.take<char>(xe * es);
  char *logq = cv.take<char>((size_t)N * es);
  char *ladj = cv.take<char>((size_t)N * es);
  double *partial = cv.take<double>(nb);
  double *result = cv.take<double>(8);
  if (cp) {
    float *xt = (float *)x;
    NF_TRY(nf_launch_layout_convert(ctx, desc->d, N, (const float *)ys, xt, 1));
    NF_TRY(coupling_chain_tiled(ctx, desc, true, (const float *)theta, xt, N, (float *)ladj, -1));
    NF_TRY(nf_launch_base_logpdf_tiled(ctx, desc->d, N, xt, (float *)logq));
  } else {
    NF_TRY(flat_apply(ctx, desc, 0, nf_layer_count(desc), true, theta, ys, N, x, ladj));
    NF_TRY(nf_launch_base_logpdf(ctx, desc->dtype, desc->d, N, x, logq));
  }
  NF_TRY(nf_launch_sum2(ctx, desc->dtype, N, logq, ladj, logliks_out, partial, 1.0 / (double)N));
  NF_TRY(nf_launch_finish_sum(ctx, partial, nb, 0, result, nullptr, nullptr));
  return read_scalar(ctx, result, ll_host);
}

// ---- forward-KL training step -----------------------------------------------------------------
// loss = -(1/Ng) sum_j [log q0(z_j) + ladj_inv_j],  z = T^-1(ys).  One inverse pass, then the reverse pass of
// the INVERSE chain: layers in forward execution order, each with the implicit-function form of its inverse
// (cotangent of z: z / Ng; cotangent of every ladj_inv: -1 / Ng).  The standard-normal base density and its
// gradient come from the diagonal-Gaussian target kernel with mu = 0, var = 1.
extern "C" int nf_loglikelihood_value_and_grad(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *ys,
                                               int64_t N_local, int64_t N_global, void *out) {
  if (!ctx || !theta || !out || N_local < 0 || N_global < 1 || (N_local > 0 && !ys)) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  // general bases and heterogeneous compositions: segment by segment in the standard layout (fkl_general below)
  if (flow_base(desc) || is_composite(desc)) return fkl_general(ctx, desc, theta, ys, N_local, N_global, out);
  const long N = N_local;
  const long P = nf_param_count(desc);
  const int dt = desc->dtype;
  const size_t es = esize(dt);
  if (N == 0) return nf_launch_fill(ctx, dt, out, P + 1, 0.0);
  const double inv = 1.0 / (double)N_global;
  const bool coupling_kind = desc->kind == NF_KIND_REALNVP || desc->kind == NF_KIND_NSF;
  const bool hf = desc->kind == NF_KIND_HAMILTONIAN;
  const bool tiled = is_coupling(desc) && coupling_inv_bwd_tiled(desc);
  if (coupling_kind && !tiled && !nf_g64_supported(desc)) return NF_ERR_UNSUPPORTED;
  const long nb = tiled ? nf_target_tiled_nblocks(N) : nf_target_nblocks(N);
  const int grid = tiled ? coupling_bwd_grid(ctx, desc, N) : 0;
  const size_t slabf = tiled ? (size_t)grid * coupling_slab_floats(ctx, desc, N) : 0;
  const size_t flat_ws = tiled ? 0 : hf ? nf_hf_bwd_ws_bytes(desc, N) : coupling_kind ? nf_g64_bwd_inv_ws_bytes(desc, N)
                                                                                       : nf_simple_bwd_ws_bytes(ctx, desc, N);
  const size_t xe = tiled ? tiled_elems(desc, N) : (size_t)N * desc->d;
  // LDS-resident RealNVP: activation stash of the inverse chain, the batch in chunks beyond the budget
  const long stash_nc = tiled ? affine_stash_chunk(ctx, desc, N) : 0;
  const int stash_nch = stash_nc ? (int)((N + stash_nc - 1) / stash_nc) : 0;
  const size_t stash_b = stash_nc ? affine_stash_bytes(ctx, desc, stash_nc) : 0;
  const size_t slabf_all = tiled ? chunked_slab_floats(ctx, desc, N, stash_nc, coupling_slab_floats(ctx, desc, N)) : 0;
  (void)slabf; (void)stash_nch;
  const size_t rqs_b = tiled ? rqs_tape_b(desc, N) : 0;
  const size_t need = 2 * carve_bytes(xe * es) + carve_bytes((size_t)N * es) + carve_bytes((size_t)nb * 8) +
                      carve_bytes(2 * (size_t)desc->d * es) + carve_bytes(slabf_all * es) + carve_bytes(flat_ws) + carve_bytes(stash_b) + rqs_b;
  NF_TRY(nf_ws_reserve(ctx, need));
  Carver cv(ctx->ws);
  char *z = cv.take<char>(xe * es);
  char *gbar = cv.take<char>(xe * es);
  char *ladj = cv.take<char>((size_t)N * es);
  double *partial = cv.take<double>(nb);
  char *q0par = cv.take<char>(2 * (size_t)desc->d * es);
  char *slab = cv.take<char>(slabf_all * es);
  char *fws = cv.take<char>(flat_ws);
  float *stash = stash_b ? cv.take<float>(stash_b / 4) : nullptr;
  void *rqs_tape = rqs_b ? (void *)cv.take<char>(rqs_b) : nullptr;
  nf_target q0;
  q0.kind = NF_TARGET_DIAGGAUSS;
  q0.p0 = q0par;
  q0.p1 = q0par + (size_t)desc->d * es;
  q0.s0 = q0.s1 = 0.0;
  NF_TRY(nf_launch_fill(ctx, dt, q0par, desc->d, 0.0));
  NF_TRY(nf_launch_fill(ctx, dt, q0par + (size_t)desc->d * es, desc->d, 1.0));

  if (tiled) {
    float *zt = (float *)z, *gt = (float *)gbar;
    NF_TRY(nf_launch_layout_convert(ctx, desc->d, N, (const float *)ys, zt, 1));
    if (stash) {  // the inverse chain leaves the operands of its own reverse pass (nf_coupling.hip, "activation stash")
      NF_TRY(coupling_pack(ctx, desc, (const float *)theta));
      const long stride = coupling_slab_floats(ctx, desc, N);
      long nslab = 0, npart = 0;
      for (long o = 0; o < N; o += stash_nc) {
        const long nc = N - o < stash_nc ? N - o : stash_nc;
        const int gc = coupling_bwd_grid(ctx, desc, nc);
        NF_TRY(nf_affine_chain(ctx, desc, true, zt + o * desc->d, nc, (float *)ladj + o, stash));
        NF_TRY(nf_launch_target_tiled(ctx, &q0, desc->d, nc, zt + o * desc->d, nullptr, (const float *)ladj + o, gt + o * desc->d, -inv,
                                      nullptr, partial + npart, -inv));
        NF_TRY(nf_affine_bwd_stashed(ctx, desc, stash, gt + o * desc->d, nullptr, (float)(-inv), nc, (float *)slab + nslab * stride,
                                     stride, gc, true));
        nslab += gc;
        npart += nf_target_tiled_nblocks(nc);
      }
      NF_TRY(nf_launch_finish_sum(ctx, partial, npart, 0, nullptr, (float *)out + P, nullptr));
      return nf_affine_reduce_slabs(ctx, desc, (const float *)slab, (int)nslab, (float *)out);
    }
    NF_TRY(coupling_chain_tiled(ctx, desc, true, (const float *)theta, zt, N, (float *)ladj, -1, rqs_tape));
    NF_TRY(nf_launch_target_tiled(ctx, &q0, desc->d, N, zt, nullptr, (const float *)ladj, gt, -inv, nullptr, partial, -inv));
    NF_TRY(nf_launch_finish_sum(ctx, partial, nb, 0, nullptr, (float *)out + P, nullptr));
    return coupling_inv_bwd(ctx, desc, (const float *)theta, zt, gt, (float)(-inv), N, (float *)slab, grid, (float *)out, rqs_tape);
  }
  if (hf) {  // every inverse layer of the Hamiltonian flow is explicit: differentiated directly (nf_hamiltonian.hip)
    NF_TRY(nf_hf_apply(ctx, desc, 0, nf_layer_count(desc), true, theta, ys, N, z, ladj));
    NF_TRY(nf_launch_target(ctx, dt, &q0, desc->d, N, z, nullptr, ladj, nullptr, gbar, -inv, nullptr, partial, -inv, 0));
    NF_TRY(nf_hf_bwd_inv(ctx, desc, theta, ys, gbar, -inv, N, out, fws));
  } else if (coupling_kind) {
    NF_TRY(nf_g64_apply(ctx, desc, 0, nf_layer_count(desc), true, theta, ys, N, z, ladj));
    NF_TRY(nf_launch_target(ctx, dt, &q0, desc->d, N, z, nullptr, ladj, nullptr, gbar, -inv, nullptr, partial, -inv, 0));
    NF_TRY(nf_g64_bwd_inv(ctx, desc, theta, z, gbar, -inv, N, out, fws));
  } else {
    NF_TRY(nf_simple_apply_stash(ctx, desc, theta, ys, N, z, ladj, fws, true));
    NF_TRY(nf_launch_target(ctx, dt, &q0, desc->d, N, z, nullptr, ladj, nullptr, gbar, -inv, nullptr, partial, -inv, 0));
    NF_TRY(nf_simple_bwd(ctx, desc, theta, z, gbar, nullptr, -inv, N, gbar, out, fws, true, true));
  }
  if (dt == NF_DTYPE_F32) return nf_launch_finish_sum(ctx, partial, nb, 0, nullptr, (float *)out + P, nullptr);
  return nf_launch_finish_sum(ctx, partial, nb, 0, (double *)out + P, nullptr, nullptr);
}


// ---- forward-KL for general bases and heterogeneous compositions --------------------------------------------
// The same computation as above, assembled from per-segment pieces in the standard layout: z_0 = ys,
// z_{s+1} = T_s^-1(z_s) with every z kept; the seed is the base's own score, -Sigma^-1 (z - mu) (k_base_general_score;
// the standard normal goes through the diagonal-Gaussian target kernel as above); then the reverse pass of each
// segment's inverse, last segment first.  A single-family flow over a general base is the one-segment case.

static size_t composite_inner_need(nf_ctx *ctx, const nf_flow_desc *desc, long N);
static long seg_theta_off(const nf_flow_desc *desc, int s);

// reverse pass of ONE homogeneous segment's inverse: yin = the segment's inverse input, zout = its inverse output
// (clobbered), gbar = cotangent of zout on entry, of yin on exit; gtheta_out = the segment's parameter gradient.
// Intermediates come from the front of the context workspace.
static int inv_bwd_std(nf_ctx *ctx, const nf_flow_desc *g, const void *theta, const void *yin, void *zout, void *gbar,
                       double lbar_const, long N, void *gtheta_out) {
  const size_t es = esize(g->dtype);
  if (is_coupling(g)) {
    const int grid = coupling_bwd_grid(ctx, g, N);
    const size_t te = tiled_elems(g, N);
    const size_t slabf = (size_t)grid * coupling_slab_floats(ctx, g, N);
    const size_t rqs_b = rqs_tape_b(g, N);
    NF_TRY(nf_ws_reserve(ctx, 2 * carve_bytes(te * 4) + carve_bytes(slabf * 4) + rqs_b + (rqs_b ? carve_bytes((size_t)N * 4) : 0)));
    Carver cv(ctx->ws);
    float *state = cv.take<float>(te);
    float *gt = cv.take<float>(te);
    float *slab = cv.take<float>(slabf);
    NF_TRY(coupling_pack(ctx, g, (const float *)theta));
    void *rqs_tape = nullptr;
    if (rqs_b) {
      // spline couplings: the reverse pass differentiates the inverse chain's own bins (RqsTape, nf_rqs.hip), so the
      // segment's inverse runs again from its input, leaving its output in the tiled state AND the tape
      rqs_tape = cv.take<char>(rqs_b);
      float *scr_ladj = cv.take<float>((size_t)N);
      NF_TRY(nf_launch_layout_convert(ctx, g->d, N, (const float *)yin, state, 1));
      NF_TRY(nf_rqs_chain(ctx, g, true, state, N, scr_ladj, -1, rqs_tape));
    } else {
      NF_TRY(nf_launch_layout_convert(ctx, g->d, N, (const float *)zout, state, 1));
    }
    NF_TRY(nf_launch_layout_convert(ctx, g->d, N, (const float *)gbar, gt, 1));
    NF_TRY(coupling_inv_bwd(ctx, g, (const float *)theta, state, gt, (float)lbar_const, N, slab, grid, (float *)gtheta_out, rqs_tape));
    return nf_launch_layout_convert(ctx, g->d, N, gt, (float *)gbar, 0);
  }
  if (g->kind == NF_KIND_HAMILTONIAN) {  // single-segment only (check_composite): the data cotangent is not needed
    NF_TRY(nf_ws_reserve(ctx, nf_hf_bwd_ws_bytes(g, N)));
    return nf_hf_bwd_inv(ctx, g, theta, yin, gbar, lbar_const, N, gtheta_out, ctx->ws);
  }
  if (is_g64(g)) {
    NF_TRY(nf_ws_reserve(ctx, nf_g64_bwd_inv_ws_bytes(g, N)));
    return nf_g64_bwd_inv(ctx, g, theta, zout, gbar, lbar_const, N, gtheta_out, ctx->ws);
  }
  // planar / radial / mean-field: the inverse again with every layer's point stashed (the same z), then the reverse pass
  const size_t sw = nf_simple_bwd_ws_bytes(ctx, g, N);
  NF_TRY(nf_ws_reserve(ctx, carve_bytes(sw) + carve_bytes((size_t)N * es)));
  char *scr_ladj = (char *)ctx->ws + carve_bytes(sw);
  NF_TRY(nf_simple_apply_stash(ctx, g, theta, yin, N, zout, scr_ladj, ctx->ws, true));
  return nf_simple_bwd(ctx, g, theta, zout, gbar, nullptr, lbar_const, N, gbar, gtheta_out, ctx->ws, true, true);
}

static inline long fkl_general_nb(long N) {
  const long a = nf_target_nblocks(N), b = nf_sum2_nblocks(N);
  return a > b ? a : b;
}
static size_t fkl_general_extra(const nf_flow_desc *desc, long N) {
  const size_t es = esize(desc->dtype);
  const nf_base *b = flow_base(desc);
  const int ns = is_composite(desc) ? desc->nsegments : 1;
  const size_t xb = carve_bytes((size_t)N * desc->d * es), cn = carve_bytes((size_t)N * es);
  return (size_t)(ns + 1 + (b && b->kind == NF_BASE_DENSE ? 1 : 0)) * xb + 3 * cn + carve_bytes((size_t)fkl_general_nb(N) * 8) +
         carve_bytes(2 * (size_t)desc->d * es);
}
static size_t fkl_general_inner(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  if (is_composite(desc)) return composite_inner_need(ctx, desc, N);
  nf_flow_desc inner = *desc;
  inner.base = nullptr;
  return ws_need_bound(ctx, &inner, N);
}
static size_t fkl_general_need(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  return fkl_general_inner(ctx, desc, N) + fkl_general_extra(desc, N);
}

static int fkl_general(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *ys, int64_t N_local,
                       int64_t N_global, void *out) {
  const long N = N_local;
  const long P = nf_param_count(desc);
  const int dt = desc->dtype;
  const size_t es = esize(dt);
  if (N == 0) return nf_launch_fill(ctx, dt, out, P + 1, 0.0);
  const double inv = 1.0 / (double)N_global;
  const nf_base *b = flow_base(desc);
  nf_flow_desc single = *desc;
  single.base = nullptr;
  const bool comp = is_composite(desc);
  const nf_flow_desc *segs = comp ? desc->segments : &single;
  const int ns = comp ? desc->nsegments : 1;
  if (!comp) {  // the shapes the single-family entry point refuses
    const bool coupling_kind = single.kind == NF_KIND_REALNVP || single.kind == NF_KIND_NSF;
    if (coupling_kind && !is_coupling(&single) && !nf_g64_supported(&single)) return NF_ERR_UNSUPPORTED;
  }
  const size_t in_need = fkl_general_inner(ctx, desc, N);
  const size_t prev_guard = ctx->ws_guard;
  NF_TRY(nf_ws_reserve(ctx, in_need + fkl_general_extra(desc, N)));
  GuardReset gr{ctx, prev_guard};
  Carver cv((char *)ctx->ws + in_need);
  const size_t xe = (size_t)N * desc->d;
  char *z[65];
  for (int s = 1; s <= ns; ++s) z[s] = cv.take<char>(xe * es);
  char *gbar = cv.take<char>(xe * es);
  char *zbuf = (b && b->kind == NF_BASE_DENSE) ? cv.take<char>(xe * es) : nullptr;
  char *ladj = cv.take<char>((size_t)N * es);
  char *tmp = cv.take<char>((size_t)N * es);
  char *logq = cv.take<char>((size_t)N * es);
  double *partial = cv.take<double>(fkl_general_nb(N));
  char *q0par = cv.take<char>(2 * (size_t)desc->d * es);
  ctx->ws_guard = in_need;  // the segments' own intermediates stay in front of these buffers

  auto seg_theta = [&](int s) -> const char * {
    return comp ? (const char *)theta + (size_t)seg_theta_off(desc, s) * es : (const char *)theta;
  };
  // the inverse chain, every segment output kept
  for (int s = 0; s < ns; ++s) {
    const void *in = s == 0 ? ys : (const void *)z[s];
    NF_TRY(apply_std(ctx, &segs[s], true, -1, seg_theta(s), in, N, z[s + 1], s == 0 ? ladj : tmp));
    if (s > 0) NF_TRY(nf_launch_sum2(ctx, dt, N, ladj, tmp, ladj, partial, 0.0));
  }
  // loss partials and the seed  gbar = -(1/Ng) dlog q0/dz
  long nbl;
  if (b) {
    NF_TRY(nf_launch_base_general_score(ctx, dt, b->kind, desc->d, N, b->mu, b->scale, b->logdet, z[ns], logq, gbar, -inv, zbuf));
    NF_TRY(nf_launch_sum2(ctx, dt, N, logq, ladj, nullptr, partial, -inv));
    nbl = nf_sum2_nblocks(N);
  } else {
    nf_target q0;
    q0.kind = NF_TARGET_DIAGGAUSS;
    q0.p0 = q0par;
    q0.p1 = q0par + (size_t)desc->d * es;
    q0.s0 = q0.s1 = 0.0;
    NF_TRY(nf_launch_fill(ctx, dt, q0par, desc->d, 0.0));
    NF_TRY(nf_launch_fill(ctx, dt, q0par + (size_t)desc->d * es, desc->d, 1.0));
    NF_TRY(nf_launch_target(ctx, dt, &q0, desc->d, N, z[ns], nullptr, ladj, nullptr, gbar, -inv, nullptr, partial, -inv, 0));
    nbl = nf_target_nblocks(N);
  }
  if (dt == NF_DTYPE_F32) NF_TRY(nf_launch_finish_sum(ctx, partial, nbl, 0, nullptr, (float *)out + P, nullptr));
  else NF_TRY(nf_launch_finish_sum(ctx, partial, nbl, 0, (double *)out + P, nullptr, nullptr));
  // the reverse pass of the inverse chain, last segment first
  for (int s = ns - 1; s >= 0; --s) {
    const void *in = s == 0 ? ys : (const void *)z[s];
    const long off = comp ? seg_theta_off(desc, s) : 0;
    NF_TRY(inv_bwd_std(ctx, &segs[s], seg_theta(s), in, z[s + 1], gbar, -inv, N, (char *)out + (size_t)off * es));
  }
  return NF_OK;
}

// ---- heterogeneous compositions: create_flow((L1, ..., Ln), q0) with mixed families ---------------------
// (src/flows/utils.jl:23-26.)  A composite is a list of homogeneous segments in flat order; every operation chains
// the segments' own kernels through standard-layout buffers that live behind the segments' intermediates (ws_guard).
static size_t composite_inner_need(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  size_t m = 0;
  for (int s = 0; s < desc->nsegments; ++s) {
    const size_t v = ws_need_bound(ctx, &desc->segments[s], N);
    if (v > m) m = v;
  }
  return m;
}
static size_t composite_extra_bytes(const nf_flow_desc *desc, long N) {
  const size_t es = esize(desc->dtype);
  return 2 * carve_bytes((size_t)N * desc->d * es) + 5 * carve_bytes((size_t)N * es) +
         carve_bytes((size_t)nf_target_nblocks(N) * 8) + carve_bytes((size_t)nf_sum2_nblocks(N) * 8) + carve_bytes(64);
}
static long seg_theta_off(const nf_flow_desc *desc, int s) {
  long off = 0;
  for (int t = 0; t < s; ++t) off += nf_param_count(&desc->segments[t]);
  return off;
}
static int composite_bufs(nf_ctx *ctx, const nf_flow_desc *desc, long N, CompBufs *cb) {
  const size_t es = esize(desc->dtype);
  const size_t in_need = composite_inner_need(ctx, desc, N);
  NF_TRY(nf_ws_reserve(ctx, in_need + composite_extra_bytes(desc, N)));
  Carver cv((char *)ctx->ws + in_need);
  cb->y = cv.take<char>((size_t)N * desc->d * es);
  cb->gbar = cv.take<char>((size_t)N * desc->d * es);
  cb->logq = cv.take<char>((size_t)N * es);
  cb->ladj = cv.take<char>((size_t)N * es);
  cb->tmp = cv.take<char>((size_t)N * es);
  cb->lbar = cv.take<char>((size_t)N * es);
  cb->spare = cv.take<char>((size_t)N * es);
  cb->partial_t = cv.take<double>(nf_target_nblocks(N));
  cb->partial_s = cv.take<double>(nf_sum2_nblocks(N));
  cb->result = cv.take<double>(8);
  ctx->ws_guard = in_need;
  return NF_OK;
}

// forward: the LAST segment is applied first; inverse: the first.  layer >= 0: one bijector, flat index over the segments.
// ladj is overwritten.  x_in may alias y_out.
static int composite_chain(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, const void *theta, const void *x_in, long N,
                           void *y_out, void *ladj, CompBufs &cb) {
  const size_t es = esize(desc->dtype);
  const int ns = desc->nsegments;
  const void *cur = x_in;
  for (int i = 0; i < ns; ++i) {
    const int sidx = inverse ? i : ns - 1 - i;
    const nf_flow_desc *g = &desc->segments[sidx];
    const char *th = (const char *)theta + (size_t)seg_theta_off(desc, sidx) * es;
    void *dst = y_out;
    NF_TRY(apply_std(ctx, g, inverse, -1, th, cur, N, dst, i == 0 ? ladj : (void *)cb.tmp));
    if (i > 0) NF_TRY(nf_launch_sum2(ctx, desc->dtype, N, ladj, cb.tmp, ladj, cb.partial_s, 0.0));
    cur = dst;
  }
  return NF_OK;
}

static int composite_apply(nf_ctx *ctx, const nf_flow_desc *desc, bool inverse, int layer, const void *theta,
                           const void *x_in, long N, void *y_out, void *ladj) {
  const size_t es = esize(desc->dtype);
  if (layer >= 0) {  // a single bijector: find its segment
    int base_l = 0;
    for (int s = 0; s < desc->nsegments; ++s) {
      const int c = nf_layer_count(&desc->segments[s]);
      if (layer < base_l + c)
        return apply_std(ctx, &desc->segments[s], inverse, layer - base_l, (const char *)theta + (size_t)seg_theta_off(desc, s) * es,
                         x_in, N, y_out, ladj);
      base_l += c;
    }
    return NF_ERR_ARG;
  }
  CompBufs cb;
  const size_t prev_guard = ctx->ws_guard;
  NF_TRY(composite_bufs(ctx, desc, N, &cb));
  GuardReset gr{ctx, prev_guard};
  return composite_chain(ctx, desc, inverse, theta, x_in, N, y_out, ladj, cb);
}

static int elbo_forward_composite(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                                  const void *xs, long N, uint64_t seed, uint64_t off, uint32_t stream_id, void *elbos_out,
                                  double *elbo_host) {
  const size_t es = esize(desc->dtype);
  CompBufs cb;
  const size_t prev_guard = ctx->ws_guard;
  NF_TRY(composite_bufs(ctx, desc, N, &cb));
  GuardReset gr{ctx, prev_guard};
  if (xs) {
    NF_HIP(hipMemcpyAsync(cb.y, xs, (size_t)N * desc->d * es, hipMemcpyDeviceToDevice, ctx->stream));
    NF_TRY(nf_launch_base_logpdf(ctx, desc->dtype, desc->d, N, cb.y, cb.logq));
  } else {
    NF_TRY(nf_launch_base_sample(ctx, desc->dtype, desc->d, N, seed, off, stream_id, cb.y, cb.logq));
  }
  NF_TRY(composite_chain(ctx, desc, false, theta, cb.y, N, cb.y, cb.ladj, cb));
  NF_TRY(nf_launch_target(ctx, desc->dtype, target, desc->d, N, cb.y, cb.logq, cb.ladj, nullptr, nullptr, 0.0, elbos_out,
                          cb.partial_t, 1.0 / (double)N, 0));
  NF_TRY(nf_launch_finish_sum(ctx, cb.partial_t, nf_target_nblocks(N), 0, cb.result, nullptr, nullptr));
  return read_scalar(ctx, cb.result, elbo_host);
}

// training step of a composition: forward with every segment's tape kept (behind the segments' intermediates), target
// and seed, then the segments' pullbacks from their tapes, last applied first
static size_t vg_composite_extra(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  const size_t es = esize(desc->dtype);
  return tape_bytes_of(ctx, desc, N) + 2 * carve_bytes((size_t)N * desc->d * es) + 2 * carve_bytes((size_t)N * es) +
         carve_bytes((size_t)nf_target_nblocks(N) * 8);
}
static size_t vg_composite_need(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  const size_t a = tape_need(ctx, desc, N, false), b = tape_need(ctx, desc, N, true);
  return (a > b ? a : b) + vg_composite_extra(ctx, desc, N);
}
static int value_and_grad_composite(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                                    const void *xs, int64_t N_local, int64_t N_global, uint64_t seed, uint64_t sample_offset,
                                    uint32_t stream_id, void *out) {
  const long N = N_local;
  const long P = nf_param_count(desc);
  const int dt = desc->dtype;
  const size_t es = esize(dt);
  if (N == 0) return nf_launch_fill(ctx, dt, out, P + 1, 0.0);
  const double inv = 1.0 / (double)N_global;
  const size_t a = tape_need(ctx, desc, N, false), b = tape_need(ctx, desc, N, true);
  const size_t in_need = a > b ? a : b;
  const size_t prev_guard = ctx->ws_guard;
  NF_TRY(nf_ws_reserve(ctx, in_need + vg_composite_extra(ctx, desc, N)));
  GuardReset gr{ctx, prev_guard};
  Carver cv((char *)ctx->ws + in_need);
  char *y = cv.take<char>((size_t)N * desc->d * es);
  char *gbar = cv.take<char>((size_t)N * desc->d * es);
  char *logq = cv.take<char>((size_t)N * es);
  char *ladj = cv.take<char>((size_t)N * es);
  double *partial = cv.take<double>(nf_target_nblocks(N));
  char *tape = cv.take<char>(tape_bytes_of(ctx, desc, N));
  if (in_need) ctx->ws_guard = in_need;
  if (xs) {
    NF_HIP(hipMemcpyAsync(y, xs, (size_t)N * desc->d * es, hipMemcpyDeviceToDevice, ctx->stream));
    NF_TRY(nf_launch_base_logpdf(ctx, dt, desc->d, N, y, logq));
  } else {
    NF_TRY(nf_launch_base_sample(ctx, dt, desc->d, N, seed, sample_offset, stream_id, y, logq));
  }
  NF_TRY(tape_fwd(ctx, desc, theta, y, N, y, ladj, tape));
  // gbar = d(-elbo/Ng)/dy, loss partials
  NF_TRY(nf_launch_target(ctx, dt, target, desc->d, N, y, logq, ladj, nullptr, gbar, -inv, nullptr, partial, -inv, 0));
  if (dt == NF_DTYPE_F32) NF_TRY(nf_launch_finish_sum(ctx, partial, nf_target_nblocks(N), 0, nullptr, (float *)out + P, nullptr));
  else NF_TRY(nf_launch_finish_sum(ctx, partial, nf_target_nblocks(N), 0, (double *)out + P, nullptr, nullptr));
  return tape_bwd(ctx, desc, theta, tape, gbar, nullptr, -inv, N, gbar, out);
}

static int loglikelihood_composite(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *ys, long N,
                                   void *logliks_out, double *ll_host) {
  CompBufs cb;
  const size_t prev_guard = ctx->ws_guard;
  NF_TRY(composite_bufs(ctx, desc, N, &cb));
  GuardReset gr{ctx, prev_guard};
  NF_TRY(composite_chain(ctx, desc, true, theta, ys, N, cb.y, cb.ladj, cb));
  NF_TRY(nf_launch_base_logpdf(ctx, desc->dtype, desc->d, N, cb.y, cb.logq));
  NF_TRY(nf_launch_sum2(ctx, desc->dtype, N, cb.logq, cb.ladj, logliks_out, cb.partial_s, 1.0 / (double)N));
  NF_TRY(nf_launch_finish_sum(ctx, cb.partial_s, nf_sum2_nblocks(N), 0, cb.result, nullptr, nullptr));
  return read_scalar(ctx, cb.result, ll_host);
}

// ---- general MvNormal(mu, Sigma) bases -----------------------------------------------------------
// q0 carries no trainable parameter (@leaf MvNormal), so a general base only changes (i) where the draws come from,
// x = mu + L eps, and (ii) log q0(x).  The wrappers below produce x in a buffer of their own, run the standard-normal
// entry point on it as caller-supplied xs (every fused / MFMA kernel untouched), and add the exact per-sample
// correction  log N(x; 0, I) - log q0(x)  to the ELBO terms; gradients with respect to theta are unaffected.
static int elbo_forward(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                        const void *xs, long N, uint64_t seed, uint64_t off, uint32_t stream_id, void *elbos_out,
                        double *elbo_host);
extern "C" int nf_base_logpdf_general(nf_ctx *ctx, int32_t dtype, const nf_base *base, int32_t d, int64_t N, const void *x,
                                      void *logq_out);

static int elbo_forward_general_base(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                                     const void *xs, long N, uint64_t seed, uint64_t off, uint32_t stream_id,
                                     void *elbos_out, double *elbo_host) {
  const nf_base *b = flow_base(desc);
  nf_flow_desc inner = *desc;
  inner.base = nullptr;
  BaseBufs bb;
  const size_t prev_guard = ctx->ws_guard;
  NF_TRY(base_bufs(ctx, desc, &inner, N, &bb));
  GuardReset gr{ctx, prev_guard};
  const void *x = xs;
  if (!xs) {
    NF_TRY(base_draw(ctx, desc, N, seed, off, stream_id, bb.x, nullptr));
    x = bb.x;
  }
  NF_TRY(nf_launch_base_general_logpdf(ctx, desc->dtype, b->kind, desc->d, N, b->mu, b->scale, b->logdet, x, nullptr, bb.corr, bb.z));
  double v = 0.0;
  NF_TRY(elbo_forward(ctx, &inner, target, theta, x, N, 0, 0, 0, elbos_out, &v));
  // per-sample terms and the mean: add the correction
  NF_TRY(nf_launch_sum2(ctx, desc->dtype, N, bb.corr, elbos_out, elbos_out, bb.partial, 1.0 / (double)N));
  if (elbos_out) {  // sum2 accumulated corr + elbos into the partials: the mean of the corrected terms
    NF_TRY(nf_launch_finish_sum(ctx, bb.partial, bb.nb, 0, bb.result, nullptr, nullptr));
    return read_scalar(ctx, bb.result, elbo_host);
  }
  NF_TRY(nf_launch_finish_sum(ctx, bb.partial, bb.nb, 0, bb.result, nullptr, nullptr));
  double c = 0.0;
  NF_TRY(read_scalar(ctx, bb.result, &c));
  *elbo_host = v + c;
  return NF_OK;
}

static int value_and_grad_general_base(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, const void *theta,
                                       const void *xs, int64_t N_local, int64_t N_global, uint64_t seed, uint64_t sample_offset,
                                       uint32_t stream_id, void *out) {
  const nf_base *b = flow_base(desc);
  const long N = N_local;
  const long P = nf_param_count(desc);
  if (N == 0) return nf_launch_fill(ctx, desc->dtype, out, P + 1, 0.0);
  nf_flow_desc inner = *desc;
  inner.base = nullptr;
  BaseBufs bb;
  const size_t prev_guard = ctx->ws_guard;
  NF_TRY(base_bufs(ctx, desc, &inner, N, &bb));
  GuardReset gr{ctx, prev_guard};
  const void *x = xs;
  if (!xs) {
    NF_TRY(base_draw(ctx, desc, N, seed, sample_offset, stream_id, bb.x, nullptr));
    x = bb.x;
  }
  NF_TRY(nf_launch_base_general_logpdf(ctx, desc->dtype, b->kind, desc->d, N, b->mu, b->scale, b->logdet, x, nullptr, bb.corr, bb.z));
  NF_TRY(nf_elbo_value_and_grad(ctx, &inner, target, theta, x, N_local, N_global, 0, 0, 0, out));
  // loss += sum_j -(corr_j) / N_global
  const size_t es = esize(desc->dtype);
  NF_TRY(nf_launch_sum2(ctx, desc->dtype, N, bb.corr, nullptr, nullptr, bb.partial, -1.0 / (double)N_global));
  if (desc->dtype == NF_DTYPE_F32) NF_TRY(nf_launch_finish_sum(ctx, bb.partial, bb.nb, 0, nullptr, (float *)bb.tmp, nullptr));
  else NF_TRY(nf_launch_finish_sum(ctx, bb.partial, bb.nb, 0, (double *)bb.tmp, nullptr, nullptr));
  char *lossp = (char *)out + (size_t)P * es;
  return nf_launch_sum2(ctx, desc->dtype, 1, lossp, bb.tmp, lossp, bb.partial, 0.0);
}

static int loglikelihood_general_base(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *ys, long N,
                                      void *logliks_out, double *ll_host) {
  const nf_base *b = flow_base(desc);
  nf_flow_desc inner = *desc;
  inner.base = nullptr;
  BaseBufs bb;
  const size_t prev_guard = ctx->ws_guard;
  NF_TRY(base_bufs(ctx, desc, &inner, N, &bb));
  GuardReset gr{ctx, prev_guard};
  // z = T^-1 y and ladj_inv with the public inverse, then log q0(z) with the general density
  NF_TRY(nf_flow_inv(ctx, &inner, theta, ys, N, bb.x, bb.tmp));
  NF_TRY(nf_launch_base_general_logpdf(ctx, desc->dtype, b->kind, desc->d, N, b->mu, b->scale, b->logdet, bb.x, bb.corr, nullptr, bb.z));
  NF_TRY(nf_launch_sum2(ctx, desc->dtype, N, bb.corr, bb.tmp, logliks_out, bb.partial, 1.0 / (double)N));
  NF_TRY(nf_launch_finish_sum(ctx, bb.partial, bb.nb, 0, bb.result, nullptr, nullptr));
  return read_scalar(ctx, bb.result, ll_host);
}

extern "C" int nf_base_rand(nf_ctx *ctx, int32_t dtype, const nf_base *base, int32_t d, int64_t N, uint64_t seed,
                            uint64_t sample_offset, uint32_t stream_id, void *x_out, void *logq_out) {
  if (!ctx || !x_out || d < 1 || N < 0) return NF_ERR_ARG;
  if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_F64) return NF_ERR_ARG;
  NF_TRY(check_base(base));
  NF_HIP(hipSetDevice(ctx->device));
  if (!base || base->kind == NF_BASE_STANDARD)
    return nf_launch_base_sample(ctx, dtype, d, N, seed, sample_offset, stream_id, x_out, logq_out);
  NF_TRY(nf_launch_base_sample(ctx, dtype, d, N, seed, sample_offset, stream_id, x_out, nullptr));
  NF_TRY(nf_launch_base_unwhiten(ctx, dtype, base->kind, d, N, base->mu, base->scale, x_out));
  if (!logq_out) return NF_OK;
  return nf_base_logpdf_general(ctx, dtype, base, d, N, x_out, logq_out);
}

extern "C" int nf_base_logpdf_general(nf_ctx *ctx, int32_t dtype, const nf_base *base, int32_t d, int64_t N, const void *x,
                                      void *logq_out) {
  if (!ctx || !x || !logq_out || d < 1 || N < 0) return NF_ERR_ARG;
  if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_F64) return NF_ERR_ARG;
  NF_TRY(check_base(base));
  NF_HIP(hipSetDevice(ctx->device));
  if (!base || base->kind == NF_BASE_STANDARD) return nf_launch_base_logpdf(ctx, dtype, d, N, x, logq_out);
  void *zbuf = nullptr;
  if (base->kind == NF_BASE_DENSE) {
    NF_TRY(nf_ws_reserve(ctx, carve_bytes((size_t)N * d * esize(dtype))));
    zbuf = ctx->ws;
  }
  return nf_launch_base_general_logpdf(ctx, dtype, base->kind, d, N, base->mu, base->scale, base->logdet, x, logq_out, nullptr, zbuf);
}

// ---- training step -------------------------------------------------------------------------
extern "C" int nf_elbo_value_and_grad(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target,
                                      const void *theta, const void *xs, int64_t N_local, int64_t N_global,
                                      uint64_t seed, uint64_t sample_offset, uint32_t stream_id, void *out) {
  if (!ctx || !target || !theta || !out || N_local < 0 || N_global < 1) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  if (flow_base(desc))
    return value_and_grad_general_base(ctx, desc, target, theta, xs, N_local, N_global, seed, sample_offset, stream_id, out);
  if (is_composite(desc))
    return value_and_grad_composite(ctx, desc, target, theta, xs, N_local, N_global, seed, sample_offset, stream_id, out);
  const long N = N_local;
  const long P = nf_param_count(desc);
  const int dt = desc->dtype;
  const size_t es = esize(dt);
  if (N == 0) return nf_launch_fill(ctx, dt, out, P + 1, 0.0);
  const double inv = 1.0 / (double)N_global;
  const bool cp = is_coupling(desc);
  const long nb = cp ? nf_target_tiled_nblocks(N) : nf_target_nblocks(N);
  const long nb_alloc = nb < ctx->num_cu ? ctx->num_cu : nb;  // the fused forward leaves one partial per workgroup
  const int grid = cp ? coupling_bwd_grid(ctx, desc, N) : 0;
  const bool simple_kind = !cp && !is_g64(desc) && desc->kind != NF_KIND_HAMILTONIAN;
  // planar / radial / mean-field within the register budget: the whole step in one launch, nothing stashed
  constexpr bool no_step = false;  // (round 6: the NF_SIMPLE_STASH A/B switch is retired -- its question is answered, README "switches")
  const bool simple_step = simple_kind && !no_step && nf_simple_step_supported(desc);
  const size_t simple_ws = cp ? 0 : simple_step ? nf_simple_step_ws_bytes(ctx, desc, N) : flat_bwd_ws_bytes(ctx, desc, N);
  // LDS-resident RealNVP: activation stash for the reverse pass, the batch in chunks if it exceeds the budget (in-library
  // draws; caller-supplied draws take one chunk or the recompute kernel)
  const bool fusable = cp && elbo_fusable(desc, target, xs);
  const long stash_nc = cp ? affine_stash_chunk(ctx, desc, N) : 0;
  const int stash_nch = stash_nc ? (int)((N + stash_nc - 1) / stash_nc) : 0;
  const size_t stash_b = stash_nc ? affine_stash_bytes(ctx, desc, stash_nc) : 0;
  const bool wide = cp && is_wide(desc);
  (void)stash_nch;
  const size_t slabf = wide ? nf_wide_train_ws_floats(ctx, desc, N)
                            : cp ? chunked_slab_floats(ctx, desc, N, stash_nc, coupling_slab_floats(ctx, desc, N)) : 0;
  const size_t xe = cp ? tiled_elems(desc, N) : simple_step ? 0 : (size_t)N * desc->d;
  const size_t rqs_b = cp ? rqs_tape_b(desc, N) : 0;
  const size_t need = 3 * carve_bytes(xe * es) + 2 * carve_bytes((size_t)N * es) + carve_bytes((size_t)nb_alloc * 8) +
                      carve_bytes(64) + carve_bytes(slabf * es) + carve_bytes(simple_ws) + carve_bytes(stash_b) + rqs_b;
  NF_TRY(nf_ws_reserve(ctx, need));
  Carver cv(ctx->ws);
  char *x = cv.take<char>(xe * es);
  char *gbar = cv.take<char>(xe * es);
  char *x0 = cv.take<char>(xe * es);  // flow input kept for the non-invertible (planar/radial) recompute
  char *logq = cv.take<char>((size_t)N * es);
  char *ladj = cv.take<char>((size_t)N * es);
  double *partial = cv.take<double>(nb_alloc);
  double *result = cv.take<double>(8);
  (void)result;
  char *slab = cv.take<char>(slabf * es);
  char *sws = cv.take<char>(simple_ws);
  float *stash = stash_b ? cv.take<float>(stash_b / 4) : nullptr;
  void *rqs_tape = rqs_b ? (void *)cv.take<char>(rqs_b) : nullptr;  // spline couplings: the forward's bins and xi

  if (fusable) {
    float *xt = (float *)x, *gt = (float *)gbar;
    NF_TRY(coupling_pack(ctx, desc, (const float *)theta));
    if (stash) {  // forward with stash + reverse pass from it, chunk by chunk: no recompute, the state is not touched
      const long stride = coupling_slab_floats(ctx, desc, N);
      long nslab = 0, npart = 0;
      for (long o = 0; o < N; o += stash_nc) {
        const long nc = N - o < stash_nc ? N - o : stash_nc;
        const int gc = coupling_bwd_grid(ctx, desc, nc);
        NF_TRY(fused_chain_elbo(ctx, desc, target, nc, seed, sample_offset + (uint64_t)o, stream_id, xt + o * desc->d,
                                gt + o * desc->d, -inv, partial + npart, -inv, stash));
        NF_TRY(nf_affine_bwd_stashed(ctx, desc, stash, gt + o * desc->d, nullptr, (float)(-inv), nc,
                                     (float *)slab + nslab * stride, stride, gc));
        nslab += gc;
        npart += nf_affine_chain_grid(ctx, nc);
      }
      return nf_affine_reduce_slabs(ctx, desc, (const float *)slab, (int)nslab, (float *)out, partial, (int)npart, (float *)out + P);
    }
    NF_TRY(fused_chain_elbo(ctx, desc, target, N, seed, sample_offset, stream_id, xt, gt, -inv, partial, -inv,
                            is_nsf(desc) ? (float *)rqs_tape : nullptr));
    if (is_nsf(desc)) {
      NF_TRY(nf_launch_finish_sum(ctx, partial, fused_chain_grid(ctx, desc, N), 0, nullptr, (float *)out + P, nullptr));
      return realnvp_bwd(ctx, desc, (const float *)theta, xt, gt, nullptr, (float)(-inv), N, (float *)slab, grid, (float *)out,
                         nullptr, 0, nullptr, rqs_tape);
    }
    // the loss partials of the fused forward are finished by the slab-reduction launch
    return realnvp_bwd(ctx, desc, (const float *)theta, xt, gt, nullptr, (float)(-inv), N, (float *)slab, grid, (float *)out,
                       partial, (int)nf_affine_chain_grid(ctx, N), (float *)out + P);
  }
  if (wide) {
    // wide RealNVP: the forward pass keeps its activations (HBM stash), the reverse pass recomputes nothing
    float *xt = (float *)x, *gt = (float *)gbar;
    if (xs) {
      NF_TRY(nf_launch_layout_convert(ctx, desc->d, N, (const float *)xs, xt, 1));
      NF_TRY(nf_launch_base_logpdf_tiled(ctx, desc->d, N, xt, (float *)logq));
    } else {
      NF_TRY(nf_launch_base_sample_tiled(ctx, desc->d, N, seed, sample_offset, stream_id, xt, (float *)logq));
    }
    NF_TRY(coupling_pack(ctx, desc, (const float *)theta));
    NF_TRY(nf_wide_train_forward(ctx, desc, xt, N, (float *)ladj, (float *)slab));
    NF_TRY(nf_launch_target_tiled(ctx, target, desc->d, N, xt, (const float *)logq, (const float *)ladj, gt, -inv,
                                  nullptr, partial, -inv));
    NF_TRY(nf_launch_finish_sum(ctx, partial, nb, 0, nullptr, (float *)out + P, nullptr));
    return nf_wide_train_backward(ctx, desc, xt, gt, nullptr, (float)(-inv), N, (float *)slab, (float *)out);
  }
  if (cp) {
    float *xt = (float *)x, *gt = (float *)gbar;
    if (xs) {
      NF_TRY(nf_launch_layout_convert(ctx, desc->d, N, (const float *)xs, xt, 1));
      NF_TRY(nf_launch_base_logpdf_tiled(ctx, desc->d, N, xt, (float *)logq));
    } else {
      NF_TRY(nf_launch_base_sample_tiled(ctx, desc->d, N, seed, sample_offset, stream_id, xt, (float *)logq));
    }
    if (stash) {
      // LDS-resident RealNVP with caller-supplied draws or a target other than the diagonal Gaussian: the plain forward
      // chain leaves the stash, too -- chunk by chunk through the stash buffer like the fused form above
      NF_TRY(coupling_pack(ctx, desc, (const float *)theta));
      const long stride = coupling_slab_floats(ctx, desc, N);
      long nslab = 0, npart = 0;
      for (long o = 0; o < N; o += stash_nc) {
        const long nc = N - o < stash_nc ? N - o : stash_nc;
        const int gc = coupling_bwd_grid(ctx, desc, nc);
        NF_TRY(nf_affine_chain(ctx, desc, false, xt + o * desc->d, nc, (float *)ladj + o, stash));
        // gt = d(-elbo/Ng)/dy = -(1/Ng) grad logp(y);  partial sums of -elbo_j/Ng
        NF_TRY(nf_launch_target_tiled(ctx, target, desc->d, nc, xt + o * desc->d, (const float *)logq + o, (const float *)ladj + o,
                                      gt + o * desc->d, -inv, nullptr, partial + npart, -inv));
        NF_TRY(nf_affine_bwd_stashed(ctx, desc, stash, gt + o * desc->d, nullptr, (float)(-inv), nc,
                                     (float *)slab + nslab * stride, stride, gc));
        nslab += gc;
        npart += nf_target_tiled_nblocks(nc);
      }
      NF_TRY(nf_launch_finish_sum(ctx, partial, npart, 0, nullptr, (float *)out + P, nullptr));
      return nf_affine_reduce_slabs(ctx, desc, (const float *)slab, (int)nslab, (float *)out);
    }
    NF_TRY(coupling_chain_tiled(ctx, desc, false, (const float *)theta, xt, N, (float *)ladj, -1, rqs_tape));
    // gt = d(-elbo/Ng)/dy = -(1/Ng) grad logp(y);  partial sums of -elbo_j/Ng
    NF_TRY(nf_launch_target_tiled(ctx, target, desc->d, N, xt, (const float *)logq, (const float *)ladj, gt, -inv,
                                  nullptr, partial, -inv));
    NF_TRY(nf_launch_finish_sum(ctx, partial, nb, 0, nullptr, (float *)out + P, nullptr));
    NF_TRY(realnvp_bwd(ctx, desc, (const float *)theta, xt, gt, nullptr, (float)(-inv), N, (float *)slab, grid,
                       (float *)out, nullptr, 0, nullptr, rqs_tape));
  } else {
    if (simple_step) {
      // planar / radial / mean-field: draws (or xs), chain, target, ELBO sums AND the reverse pass in one launch;
      // every layer input stays in registers, only the parameter slabs are written (k_simple_step)
      NF_TRY(nf_target_check(target, desc->d));
      long np = 0;
      NF_TRY(nf_simple_elbo_step(ctx, desc, target, theta, xs, N, seed, sample_offset, stream_id, -inv, -inv, partial, -inv, sws,
                                 out, &np));
      if (dt == NF_DTYPE_F32) return nf_launch_finish_sum(ctx, partial, np, 0, nullptr, (float *)out + P, nullptr);
      return nf_launch_finish_sum(ctx, partial, np, 0, (double *)out + P, nullptr, nullptr);
    }
    if (simple_kind) {
      // flows beyond the register budget of k_simple_step: draws + chain + target + ELBO partial sums in one launch
      // (the per-layer inputs stay behind for the reverse pass), then the reverse pass over all layers
      NF_TRY(nf_target_check(target, desc->d));
      long np = 0;
      NF_TRY(nf_simple_elbo_forward(ctx, desc, target, theta, xs, N, seed, sample_offset, stream_id, gbar, -inv, partial, -inv,
                                    sws, &np));
      NF_TRY(nf_simple_bwd(ctx, desc, theta, nullptr, gbar, nullptr, -inv, N, gbar, out, sws, true));
      if (dt == NF_DTYPE_F32) return nf_launch_finish_sum(ctx, partial, np, 0, nullptr, (float *)out + P, nullptr);
      return nf_launch_finish_sum(ctx, partial, np, 0, (double *)out + P, nullptr, nullptr);
    }
    if (xs) {
      NF_HIP(hipMemcpyAsync(x0, xs, (size_t)N * desc->d * es, hipMemcpyDeviceToDevice, ctx->stream));
      NF_TRY(nf_launch_base_logpdf(ctx, dt, desc->d, N, x0, logq));
    } else {
      NF_TRY(nf_launch_base_sample(ctx, dt, desc->d, N, seed, sample_offset, stream_id, x0, logq));
    }
    {
      // general couplings: the forward leaves every coupling's input in the reverse pass's workspace (no second forward)
      const bool keep = is_g64(desc);
      if (keep) NF_TRY(nf_g64_apply_keep(ctx, desc, theta, x0, N, x, ladj, sws));
      else NF_TRY(flat_apply(ctx, desc, 0, nf_layer_count(desc), false, theta, x0, N, x, ladj));
      NF_TRY(nf_launch_target(ctx, dt, target, desc->d, N, x, logq, ladj, nullptr, gbar, -inv, nullptr, partial, -inv, joint_dims(desc)));
      NF_TRY(flat_bwd(ctx, desc, theta, keep ? nullptr : x0, gbar, nullptr, -inv, N, gbar, out, sws));
    }
  }
  if (cp) return NF_OK;
  if (dt == NF_DTYPE_F32)
    return nf_launch_finish_sum(ctx, partial, nb, 0, nullptr, (float *)out + P, nullptr);
  return nf_launch_finish_sum(ctx, partial, nb, 0, (double *)out + P, nullptr, nullptr);
}

extern "C" int nf_adam_update(nf_ctx *ctx, int32_t dtype, void *theta, const void *g, void *m, void *v, int64_t P,
                              double lr, double beta1, double beta2, double eps, int64_t t, void *gnorm_out) {
  if (!ctx || !theta || !g || !m || !v || P < 0 || t < 1) return NF_ERR_ARG;
  if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_F64) return NF_ERR_ARG;
  NF_HIP(hipSetDevice(ctx->device));
  if (P == 0) return NF_OK;
  double *partial = nullptr;
  const long nb = nf_adam_nblocks(P);
  if (gnorm_out) {
    // the arena may be in use by a preceding call on the same stream: stream order makes reuse safe
    NF_TRY(nf_ws_reserve(ctx, carve_bytes((size_t)nb * 8)));
    partial = (double *)((char *)ctx->ws + ctx->ws_bytes - carve_bytes((size_t)nb * 8));
  }
  NF_TRY(nf_launch_adam(ctx, dtype, theta, g, m, v, P, lr, beta1, beta2, eps, t, partial));
  if (gnorm_out) {
    if (dtype == NF_DTYPE_F32) return nf_launch_finish_sum(ctx, partial, nb, 1, nullptr, (float *)gnorm_out, nullptr);
    return nf_launch_finish_sum(ctx, partial, nb, 1, (double *)gnorm_out, nullptr, nullptr);
  }
  return NF_OK;
}

extern "C" int nf_sgd_update(nf_ctx *ctx, int32_t dtype, void *theta, const void *g, void *vel, int64_t P, double lr,
                             double rho, void *gnorm_out) {
  if (!ctx || !theta || !g || P < 0) return NF_ERR_ARG;
  if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_F64) return NF_ERR_ARG;
  NF_HIP(hipSetDevice(ctx->device));
  if (P == 0) return NF_OK;
  double *partial = nullptr;
  const long nb = nf_adam_nblocks(P);
  if (gnorm_out) {
    NF_TRY(nf_ws_reserve(ctx, carve_bytes((size_t)nb * 8)));
    partial = (double *)((char *)ctx->ws + ctx->ws_bytes - carve_bytes((size_t)nb * 8));
  }
  NF_TRY(nf_launch_sgd(ctx, dtype, theta, g, vel, P, lr, rho, partial));
  if (gnorm_out) {
    if (dtype == NF_DTYPE_F32) return nf_launch_finish_sum(ctx, partial, nb, 1, nullptr, (float *)gnorm_out, nullptr);
    return nf_launch_finish_sum(ctx, partial, nb, 1, (double *)gnorm_out, nullptr, nullptr);
  }
  return NF_OK;
}

// ---- the fused training step of the LDS-resident RealNVP path -----------------------------------------------------
// nf_elbo_step for cfg-2-like flows is three launches plus a one-block finish: k_affine_chain<FUSED, STASH> (draws + chain +
// target + ELBO sums, leaving the activation stash), k_affine_bwd_stashed (reverse pass of every coupling), k_affine_epilogue
// (slab sum -> gradient, loss, Adam, partials of ||g||^2, packed weight images of the UPDATED theta) and k_finish_sum (||g||).
// The packed images survive from one step to the next: no k_pack_net_images, no separate slab-reduction / Adam launches.  With a communicator on the
// context (nf_comm_init_*) the step is the data-parallel one: this rank draws samples [rank N, (rank + 1) N) of a global
// batch of N * nranks, the epilogue splits around ONE all-reduce of [grad ; loss] (P + 1 floats).
// step_ptr != nullptr: the Philox stream id and Adam's step count come from that device counter, which the epilogue
// increments -- every launch argument is then the same from step to step, so one call can be captured into a hipGraph
// and replayed (nf_elbo_step_enqueue).
static inline unsigned long long flow_sig(const nf_flow_desc *d) {
  // FNV-1a over every descriptor field the packed images depend on (kind, element type, d, depth, conditioner shape)
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](unsigned long long v) { h = (h ^ v) * 1099511628211ull; };
  mix((unsigned long long)d->kind);
  mix((unsigned long long)d->dtype);
  mix((unsigned long long)d->d);
  mix((unsigned long long)d->nlayers);
  mix((unsigned long long)d->n_hidden);
  for (int i = 0; i < d->n_hidden && i < 4; ++i) mix((unsigned long long)d->hdims[i]);
  return h | (1ull << 63);
}
static bool step_fusable(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, long N) {
  constexpr bool off = false;  // (round 6: the NF_STEP_UNFUSED A/B switch is retired -- its question is answered, README "switches")
  if (off || flow_base(desc) || is_composite(desc) || desc->dtype != NF_DTYPE_F32) return false;
  if (!(desc->kind == NF_KIND_REALNVP && nf_affine_supported(desc))) return false;
  if (!elbo_fusable(desc, target, nullptr)) return false;
  return affine_stash_chunk(ctx, desc, N) > 0;
}
static size_t step_fused_need(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  const size_t te = tiled_elems(desc, N);
  const long nb = nf_target_tiled_nblocks(N);
  const long nb_alloc = nb < ctx->num_cu ? ctx->num_cu : nb;
  const long snc = affine_stash_chunk(ctx, desc, N);
  return 2 * carve_bytes(te * 4) + carve_bytes((size_t)nb_alloc * 8) +
         carve_bytes(chunked_slab_floats(ctx, desc, N, snc, coupling_slab_floats(ctx, desc, N)) * 4) +
         carve_bytes(snc ? affine_stash_bytes(ctx, desc, snc) : 0) + carve_bytes((size_t)nf_affine_epilogue_blocks(desc) * 8);
}
static int elbo_step_fused(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, float *theta, float *m, float *v,
                           long N, uint64_t seed, uint32_t step_val, uint32_t *step_ptr, double lr, double beta1, double beta2,
                           double eps) {
  const long P = nf_param_count(desc);
  const int world = ctx->comm ? ctx->comm_size : 1;
  const long Ng = N * world;
  const uint64_t off0 = (uint64_t)(ctx->comm ? ctx->comm_rank : 0) * (uint64_t)N;
  const double inv = 1.0 / (double)Ng;
  const size_t te = tiled_elems(desc, N);
  const long nb = nf_target_tiled_nblocks(N);
  const long nb_alloc = nb < ctx->num_cu ? ctx->num_cu : nb;
  const long stash_nc = affine_stash_chunk(ctx, desc, N);
  const long stride = coupling_slab_floats(ctx, desc, N);
  const size_t slabf = chunked_slab_floats(ctx, desc, N, stash_nc, stride);
  const size_t stash_b = affine_stash_bytes(ctx, desc, stash_nc);
  const long eblocks = nf_affine_epilogue_blocks(desc);
  NF_TRY(nf_ws_reserve(ctx, step_fused_need(ctx, desc, N)));
  Carver cv(ctx->ws);
  float *xt = cv.take<float>(te);
  float *gt = cv.take<float>(te);
  double *partial = cv.take<double>(nb_alloc);
  float *slab = cv.take<float>(slabf);
  float *stash = cv.take<float>(stash_b / 4);
  double *gpart = cv.take<double>(eblocks);
  float *gbuf = (float *)ctx->gbuf;
  // packed images: those the previous step's epilogue left, or a fresh pack
  // (ONLY under nf_ctx_set_weight_cache(ctx, 1): by default every step packs from theta, one 8 us launch, so a theta
  // that was edited in place, or freed and re-allocated at the same address, can never meet stale images)
  constexpr bool repack = false;  // (round 6: the NF_STEP_REPACK A/B switch is retired -- its question is answered, README "switches")
  if (repack || !ctx->wimg_cache ||
      !(ctx->wimg && ctx->wimg_owner == (const void *)theta && ctx->wimg_sig == flow_sig(desc))) {
    NF_TRY(coupling_pack(ctx, desc, theta));
  }
  long nslab = 0, npart = 0;
  for (long o = 0; o < N; o += stash_nc) {
    const long nc = N - o < stash_nc ? N - o : stash_nc;
    const int gc = coupling_bwd_grid(ctx, desc, nc);
    NF_TRY(nf_affine_chain_elbo(ctx, desc, nc, seed, off0 + (uint64_t)o, step_val, (const float *)target->p0, (const float *)target->p1,
                                xt + o * desc->d, gt + o * desc->d, -inv, partial + npart, -inv, stash, step_ptr));
    NF_TRY(nf_affine_bwd_stashed(ctx, desc, stash, gt + o * desc->d, nullptr, (float)(-inv), nc, slab + nslab * stride, stride, gc));
    nslab += gc;
    npart += nf_affine_chain_grid(ctx, nc);
  }
  if (world == 1) {
    NF_TRY(nf_affine_epilogue(ctx, desc, 3, slab, (int)nslab, gbuf, partial, (int)npart, theta, m, v, lr, beta1, beta2, eps, step_val,
                              step_ptr, gpart));
  } else {
    NF_TRY(nf_affine_epilogue(ctx, desc, 1, slab, (int)nslab, gbuf, partial, (int)npart, nullptr, nullptr, nullptr, lr, beta1, beta2,
                              eps, step_val, nullptr, gpart));
    NF_TRY(nf_allreduce_grad_loss(ctx, NF_DTYPE_F32, gbuf, P + 1));
    NF_TRY(nf_affine_epilogue(ctx, desc, 2, nullptr, 0, gbuf, nullptr, 0, theta, m, v, lr, beta1, beta2, eps, step_val, step_ptr,
                              gpart));
  }
  // norm(g) = sqrt of the epilogue's block partials (and, in the graph-replay form, the step counter's increment)
  NF_TRY(nf_launch_finish_sum(ctx, gpart, eblocks, 1, nullptr, gbuf + P + 1, nullptr, step_ptr));
  ctx->wimg_owner = theta;
  ctx->wimg_sig = flow_sig(desc);
  return NF_OK;
}

// The same for LDS-resident spline couplings on ONE rank (round 6): fused forward (draws + chain + target + loss partials, leaving the
// spline tape), one reverse launch per coupling, and nf_rqs_epilogue (loss, slab reduction, Adam, the updated images) -- four launches
// fewer than nf_elbo_value_and_grad + nf_adam_update + the next step's pack.  Multi-rank contexts keep the generic sequence (the
// all-reduce sits between the reduction and Adam).
static bool step_fusable_rqs(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target) {
#ifdef NF_STEP_RQS_UNFUSED  // A/B builds (tools/ab_build.py): the generic sequence
  return false;
#endif
  if (ctx->comm || flow_base(desc) || is_composite(desc) || desc->dtype != NF_DTYPE_F32) return false;
  return desc->kind == NF_KIND_NSF && nf_rqs_supported(desc) && elbo_fusable(desc, target, nullptr);
}
static size_t step_fused_need_rqs(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  const size_t te = tiled_elems(desc, N);
  const long nb = nf_target_tiled_nblocks(N);
  const long nb_alloc = nb < ctx->num_cu ? ctx->num_cu : nb;
  return 2 * carve_bytes(te * 4) + carve_bytes((size_t)nb_alloc * 8) +
         carve_bytes((size_t)coupling_bwd_grid(ctx, desc, N) * coupling_slab_floats(ctx, desc, N) * 4) + rqs_tape_b(desc, N) +
         carve_bytes((size_t)nf_rqs_epilogue_blocks(desc) * 8);
}
static int elbo_step_fused_rqs(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, float *theta, float *m, float *v, long N,
                               uint64_t seed, uint32_t step_val, double lr, double beta1, double beta2, double eps) {
  const long P = nf_param_count(desc);
  const double inv = 1.0 / (double)N;
  const size_t te = tiled_elems(desc, N);
  const long nb = nf_target_tiled_nblocks(N);
  const long nb_alloc = nb < ctx->num_cu ? ctx->num_cu : nb;
  const int grid = coupling_bwd_grid(ctx, desc, N);
  const long stride = coupling_slab_floats(ctx, desc, N);
  const long eblocks = nf_rqs_epilogue_blocks(desc);
  NF_TRY(nf_ws_reserve(ctx, step_fused_need_rqs(ctx, desc, N)));
  Carver cv(ctx->ws);
  float *xt = cv.take<float>(te);
  float *gt = cv.take<float>(te);
  double *partial = cv.take<double>(nb_alloc);
  float *slab = cv.take<float>((size_t)grid * stride);
  void *tape = (void *)cv.take<char>(rqs_tape_b(desc, N));
  double *gpart = cv.take<double>(eblocks);
  float *gbuf = (float *)ctx->gbuf;
  // packed images: those the previous step's epilogue left (only under nf_ctx_set_weight_cache(ctx, 1)), or a fresh pack
  if (!ctx->wimg_cache || !(ctx->wimg && ctx->wimg_owner == (const void *)theta && ctx->wimg_sig == flow_sig(desc)))
    NF_TRY(coupling_pack(ctx, desc, theta));
  NF_TRY(fused_chain_elbo(ctx, desc, target, N, seed, 0, step_val, xt, gt, -inv, partial, -inv, (float *)tape));
  const int nc = 2 * desc->nlayers;
  for (int k = 0; k < nc; ++k) NF_TRY(nf_rqs_bwd(ctx, desc, k, xt, gt, nullptr, (float)(-inv), N, slab, stride, grid, false, tape));
  NF_TRY(nf_rqs_epilogue(ctx, desc, slab, grid, gbuf, partial, (int)fused_chain_grid(ctx, desc, N), theta, m, v, lr, beta1, beta2, eps,
                         step_val, gpart));
  NF_TRY(nf_launch_finish_sum(ctx, gpart, eblocks, 1, nullptr, gbuf + P + 1, nullptr, nullptr));
  ctx->wimg_owner = theta;
  ctx->wimg_sig = flow_sig(desc);
  return NF_OK;
}

extern "C" int nf_ctx_weights_changed(nf_ctx *ctx) {
  if (!ctx) return NF_ERR_ARG;
  ctx->wimg_owner = nullptr;
  return NF_OK;
}

extern "C" int nf_ctx_set_weight_cache(nf_ctx *ctx, int32_t enable) {
  if (!ctx) return NF_ERR_ARG;
  ctx->wimg_cache = enable != 0;
  ctx->wimg_owner = nullptr;  // either way the next step packs from theta
  return NF_OK;
}

static int step_readback(nf_ctx *ctx, const nf_flow_desc *desc, long P, double *loss_host, double *gnorm_host) {
  const size_t es = esize(desc->dtype);
  NF_HIP(hipMemcpyAsync(ctx->host_scratch, (char *)ctx->gbuf + (size_t)P * es, 2 * es, hipMemcpyDeviceToHost, ctx->stream));
  NF_HIP(hipStreamSynchronize(ctx->stream));
  double l, gnv;
  if (desc->dtype == NF_DTYPE_F32) {
    l = ((float *)ctx->host_scratch)[0];
    gnv = ((float *)ctx->host_scratch)[1];
  } else {
    l = ctx->host_scratch[0];
    gnv = ctx->host_scratch[1];
  }
  if (loss_host) *loss_host = l;
  if (gnorm_host) *gnorm_host = gnv;
  // the reference's tests require finite ELBOs (test/flow.jl:58-60); theta has already been updated with the
  // non-finite gradient, as Optimisers.update! would have done -- the caller decides whether to stop
  if (!std::isfinite(l) || !std::isfinite(gnv)) return NF_ERR_NONFINITE;
  return NF_OK;
}

// couplings per all-reduce bucket for this flow under the context's communicator; 0: one message (no communicator, a flow
// whose reverse pass does not finish coupling by coupling on the host side, or a gradient smaller than two buckets)
static int comm_bucket_couplings(nf_ctx *ctx, const nf_flow_desc *desc) {
  if (!ctx->comm || ctx->comm_bucket_bytes == 0) return 0;
  if (flow_base(desc) || is_composite(desc) || desc->dtype != NF_DTYPE_F32 || !is_coupling(desc) || !is_wide(desc)) return 0;
  const long long bytes = ctx->comm_bucket_bytes < 0 ? (4ll << 20) : ctx->comm_bucket_bytes;
  const int nc = 2 * desc->nlayers;
  const long long per = (long long)nf_param_count(desc) * 4 / nc;  // bytes of one coupling's parameters
  if (per * nc < 2 * bytes) return 0;
  long long c = (bytes + per / 2) / per;
  if (c < 1) c = 1;
  return c >= nc ? 0 : (int)c;
}

extern "C" int nf_comm_bucket_count(nf_ctx *ctx, const nf_flow_desc *desc) {
  if (!ctx) return NF_ERR_ARG;
  const int st = check_desc(desc);
  if (st != NF_OK) return st;
  const int cpb = comm_bucket_couplings(ctx, desc);
  if (cpb <= 0) return ctx->comm ? 1 : 0;
  return (2 * desc->nlayers + cpb - 1) / cpb;
}

extern "C" int nf_elbo_step(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, void *theta, void *m,
                            void *v, int64_t N, uint64_t seed, uint32_t step, double lr, double beta1, double beta2,
                            double eps, double *loss_host, double *gnorm_host) {
  if (!ctx || !target || !theta || !m || !v || N < 1) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  const long P = nf_param_count(desc);
  const size_t es = esize(desc->dtype);
  NF_TRY(gbuf_reserve(ctx, gbuf_need(P, es)));
  void *gbuf = ctx->gbuf;
  if (step_fusable(ctx, desc, target, N)) {
    NF_TRY(elbo_step_fused(ctx, desc, target, (float *)theta, (float *)m, (float *)v, N, seed, step, nullptr, lr, beta1, beta2, eps));
  } else if (step_fusable_rqs(ctx, desc, target)) {
    NF_TRY(elbo_step_fused_rqs(ctx, desc, target, (float *)theta, (float *)m, (float *)v, N, seed, step, lr, beta1, beta2, eps));
  } else {
    const int world = ctx->comm ? ctx->comm_size : 1;
    const uint64_t off = (uint64_t)(ctx->comm ? ctx->comm_rank : 0) * (uint64_t)N;
    // weight-streaming RealNVP with many parameters (cfg 4: 16.9 MB): the all-reduce goes out in buckets of whole couplings
    // on the second stream as the reverse pass produces them (nf_comm.hip); everything else: one message after the gradient
    const int cpb = comm_bucket_couplings(ctx, desc);
    ctx->bucket.on = cpb > 0;
    ctx->bucket.couplings = cpb;
    ctx->bucket.issued = 0;
    const int st_vg = nf_elbo_value_and_grad(ctx, desc, target, theta, nullptr, N, N * world, seed, off, step, gbuf);
    ctx->bucket.on = false;
    if (cpb > 0) {
      // Join whatever was issued even when the gradient failed on this rank: the compute stream must not run ahead
      // of collectives still in flight on the second stream, and the caller must learn that the step's buckets are incomplete
      // -- the peers have enqueued ALL of theirs and will wait in RCCL for the ones this rank never issued.  That state is not
      // recoverable inside the library: the error is returned, the context's communicator is unusable from here on
      // (nf_comm_destroy + a new nf_comm_init_rank on every rank), which the header says of any failed collective call.
      const int st_join = nf_comm_bucket_join(ctx);
      const int want = (2 * desc->nlayers + cpb - 1) / cpb;
      // whatever else went wrong, a rank that issued fewer messages than nf_comm_bucket_count promises leaves its
      // peers waiting in RCCL -- that is a property of the COMMUNICATOR, reported as NF_ERR_RCCL and remembered (every later
      // collective call on this context fails until nf_comm_destroy); a local failure with a complete sequence keeps its code
      if (ctx->bucket.issued != want) {
        ctx->comm_poisoned = true;
        return NF_ERR_RCCL;
      }
      if (st_vg != NF_OK) return st_vg;
      NF_TRY(st_join);
    } else {
      NF_TRY(st_vg);
      if (world > 1) NF_TRY(nf_allreduce_grad_loss(ctx, desc->dtype, gbuf, P + 1));
    }
    char *gnorm_dev = (char *)gbuf + (size_t)(P + 1) * es;
    NF_TRY(nf_adam_update(ctx, desc->dtype, theta, gbuf, m, v, P, lr, beta1, beta2, eps, (int64_t)step + 1, gnorm_dev));
  }
  if (loss_host || gnorm_host) return step_readback(ctx, desc, P, loss_host, gnorm_host);
  return NF_OK;
}

// The graph-capturable form: no host-side value changes from step to step.  *step_device (a uint32 in device memory,
// owned by the caller, initialised to the first step index) is read as the Philox stream id and as Adam's t - 1 and
// is incremented by the step itself; loss and ||g|| stay on the device (out_loss_gnorm_device[2], optional copy of the
// step's [loss ; norm]).  Capture ONE call between hipStreamBeginCapture / EndCapture on the context's stream after a
// warm-up call (the warm-up sizes the workspace and sets kernel attributes -- neither is capturable), then replay.
extern "C" int nf_elbo_step_enqueue(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, void *theta, void *m, void *v,
                                    int64_t N, uint64_t seed, uint32_t *step_device, double lr, double beta1, double beta2,
                                    double eps, void *out_loss_gnorm_device) {
  if (!ctx || !target || !theta || !m || !v || !step_device || N < 1) return NF_ERR_ARG;
  NF_TRY(check_desc(desc));
  NF_HIP(hipSetDevice(ctx->device));
  const long P = nf_param_count(desc);
  if (!step_fusable(ctx, desc, target, N)) return NF_ERR_UNSUPPORTED;
  NF_TRY(gbuf_reserve(ctx, gbuf_need(P, 4)));
  NF_TRY(elbo_step_fused(ctx, desc, target, (float *)theta, (float *)m, (float *)v, N, seed, 0, step_device, lr, beta1, beta2, eps));
  if (out_loss_gnorm_device)
    NF_HIP(hipMemcpyAsync(out_loss_gnorm_device, (char *)ctx->gbuf + (size_t)P * 4, 8, hipMemcpyDeviceToDevice, ctx->stream));
  return NF_OK;
}

// ---- arena sizing ------------------------------------------------------------------------------
// Upper bound, over EVERY compute entry point called with this flow and up to N samples, of the device memory the
// context needs: the intermediates arena (the per-entry-point `need` formulas above, restated here -- the arena-mode
// GPU test runs every entry point against exactly this bound, so a formula that drifts fails loudly with
// NF_ERR_WORKSPACE), the optimiser's norm partials, the packed weight images and the nf_elbo_step buffer.
size_t nf_affine_wimg_bytes(const nf_flow_desc *desc);
size_t nf_l64_scratch_bytes(const nf_flow_desc *desc, long N);
size_t nf_rqs_wimg_bytes(const nf_flow_desc *desc);
size_t nf_wide_wimg_bytes(nf_ctx *, const nf_flow_desc *desc);

// intermediates only (what nf_ws_reserve is asked for), standard-normal base
static size_t composite_inner_need(nf_ctx *ctx, const nf_flow_desc *desc, long N);
static size_t composite_extra_bytes(const nf_flow_desc *desc, long N);
static size_t ws_need_bound(nf_ctx *ctx, const nf_flow_desc *desc, long N) {
  if (is_composite(desc)) {
    size_t need = composite_inner_need(ctx, desc, N) + composite_extra_bytes(desc, N);
    const size_t v = vg_composite_need(ctx, desc, N), f = flow_bwd_need(ctx, desc, N);
    if (v > need) need = v;
    if (f > need) need = f;
    return need;
  }
  const size_t es = esize(desc->dtype);
  const bool cp = is_coupling(desc);
  const size_t te = cp ? tiled_elems(desc, N) : 0;
  const size_t xe = cp ? te : (size_t)N * desc->d;
  const size_t cn = carve_bytes((size_t)N * es);
  size_t need = 0;
  auto upd = [&](size_t v) { if (v > need) need = v; };
  // nf_flow_fwd / inv / layer_apply, nf_flow_rand
  upd(cp ? carve_bytes(te * 4) + carve_bytes((size_t)N * 4) : cn);
  // nf_flow_bwd (its own forward + tape), nf_flow_fwd_keep / nf_flow_bwd_kept (tape in the caller's memory)
  const int grid = cp ? coupling_bwd_grid(ctx, desc, N) : 0;
  const size_t slab_pull = cp ? (size_t)grid * coupling_slab_floats(ctx, desc, N) : 0;
  upd(flow_bwd_need(ctx, desc, N));
  // nf_elbo_batch(_rng), nf_loglikelihood
  const long nbt = cp ? nf_target_tiled_nblocks(N) : nf_target_nblocks(N);
  const long nb_alloc = nbt < ctx->num_cu ? ctx->num_cu : nbt;
  upd(carve_bytes(xe * es) + 2 * cn + carve_bytes((size_t)nb_alloc * 8) + carve_bytes(64));
  upd(carve_bytes(xe * es) + 2 * cn + carve_bytes((size_t)nf_sum2_nblocks(N) * 8) + carve_bytes(64));
  // nf_loglikelihood_value_and_grad
  {
    const bool coupling_kind = desc->kind == NF_KIND_REALNVP || desc->kind == NF_KIND_NSF;
    const bool hf = desc->kind == NF_KIND_HAMILTONIAN;
    const bool tiled = cp && coupling_inv_bwd_tiled(desc);
    if (!(coupling_kind && !tiled && !nf_g64_supported(desc))) {
      const long snc = tiled ? affine_stash_chunk(ctx, desc, N) : 0;
      const size_t slabf = tiled ? chunked_slab_floats(ctx, desc, N, snc, coupling_slab_floats(ctx, desc, N)) : 0;
      const size_t flat_ws = tiled ? 0 : hf ? nf_hf_bwd_ws_bytes(desc, N) : coupling_kind ? nf_g64_bwd_inv_ws_bytes(desc, N)
                                                                                       : nf_simple_bwd_ws_bytes(ctx, desc, N);
      upd(2 * carve_bytes(xe * es) + cn + carve_bytes((size_t)nbt * 8) + carve_bytes(2 * (size_t)desc->d * es) +
          carve_bytes(slabf * es) + carve_bytes(flat_ws) + carve_bytes(snc ? affine_stash_bytes(ctx, desc, snc) : 0) +
          (tiled ? rqs_tape_b(desc, N) : 0));
    }
  }
  // nf_elbo_value_and_grad / nf_elbo_step (both the stash-free and the stash form of the simple flows)
  {
    const bool wide = cp && is_wide(desc);
    const long snc = cp ? affine_stash_chunk(ctx, desc, N) : 0;
    const size_t slabf = wide ? nf_wide_train_ws_floats(ctx, desc, N)
                              : cp ? chunked_slab_floats(ctx, desc, N, snc, coupling_slab_floats(ctx, desc, N)) : 0;
    size_t simple_ws = cp ? 0 : flat_bwd_ws_bytes(ctx, desc, N);
    if (!cp && !is_g64(desc) && desc->kind != NF_KIND_HAMILTONIAN && nf_simple_step_supported(desc)) {
      const size_t sw = nf_simple_step_ws_bytes(ctx, desc, N);
      if (sw > simple_ws) simple_ws = sw;
    }
    upd(3 * carve_bytes(xe * es) + 2 * cn + carve_bytes((size_t)nb_alloc * 8) + carve_bytes(64) +
        carve_bytes(slabf * es) + carve_bytes(simple_ws) + carve_bytes(snc ? affine_stash_bytes(ctx, desc, snc) : 0) +
        (cp ? rqs_tape_b(desc, N) : 0));
    // forward-KL over a general base / inside a composition: inv_bwd_std's own intermediates (spline couplings re-run
    // their inverse chain with the tape)
    if (cp) upd(2 * carve_bytes(te * 4) + carve_bytes(slab_pull * 4) + rqs_tape_b(desc, N) + cn);
  }
  return need;
}

// what the general-base wrappers keep behind the inner entry point's intermediates: x (N x d), the per-sample
// correction, the substitution scratch of a dense base, block partials and a result slot
static size_t base_extra_bytes(const nf_flow_desc *desc, long N) {
  const nf_base *b = flow_base(desc);
  if (!b) return 0;
  const size_t es = esize(desc->dtype);
  const size_t xb = carve_bytes((size_t)N * desc->d * es);
  return xb + (b->kind == NF_BASE_DENSE ? xb : 0) + 2 * carve_bytes((size_t)N * es) +
         carve_bytes((size_t)nf_sum2_nblocks(N) * 8) + carve_bytes(64);
}

extern "C" int64_t nf_workspace_bytes(nf_ctx *ctx, const nf_flow_desc *desc, int64_t N) {
  if (!ctx || N < 0) return NF_ERR_ARG;
  const int st = check_desc(desc);
  if (st != NF_OK) return st;
  if (N == 0) N = 1;
  const size_t es = esize(desc->dtype);
  const bool cp = is_coupling(desc);
  const long P = nf_param_count(desc);
  size_t need = ws_need_bound(ctx, desc, N) + base_extra_bytes(desc, N);
  if (flow_base(desc) || is_composite(desc)) {  // nf_loglikelihood_value_and_grad's segment-wise form
    const size_t f = fkl_general_need(ctx, desc, N);
    if (f > need) need = f;
  }
  // nf_elbo_step's three-launch form
  if (desc->kind == NF_KIND_REALNVP && desc->dtype == NF_DTYPE_F32 && !is_composite(desc) && nf_affine_supported(desc) &&
      affine_stash_chunk(ctx, desc, N) > 0) {
    const size_t f = step_fused_need(ctx, desc, N);
    if (f > need) need = f;
  }
  if (desc->kind == NF_KIND_NSF && desc->dtype == NF_DTYPE_F32 && !is_composite(desc) && !flow_base(desc) && nf_rqs_supported(desc)) {
    const size_t f = step_fused_need_rqs(ctx, desc, N);
    if (f > need) need = f;
  }
  // nf_adam_update / nf_sgd_update: gradient-norm partials at the tail of the intermediates arena
  need += carve_bytes((size_t)nf_adam_nblocks(P) * 8);
  // packed weight images, nf_elbo_step's [grad ; loss ; norm] buffer
  size_t wimg = 0;
  auto wimg_of = [&](const nf_flow_desc *g) -> size_t {
    if (!is_coupling(g)) return nf_l64_scratch_bytes(g, N);  // general Float32 couplings: the MFMA MLP's activation tiles
    return is_wide(g) ? nf_wide_wimg_bytes(ctx, g) : is_nsf(g) ? nf_rqs_wimg_bytes(g) : nf_affine_wimg_bytes(g);
  };
  if (is_composite(desc)) {
    for (int sgi = 0; sgi < desc->nsegments; ++sgi) {
      wimg += carve_bytes(wimg_of(&desc->segments[sgi]));  // grow-only tail carves: a later, larger segment carves again
    }
  } else {
    wimg = wimg_of(desc);
  }
  need += carve_bytes(wimg) + carve_bytes(gbuf_need(P, es)) + 4096;
  return (int64_t)need;
}

// ---- measurement support -------------------------------------------------------------------
extern "C" int nf_prof_enable(nf_ctx *ctx, int32_t mode) {
  if (!ctx || mode < 0 || mode > 3) return NF_ERR_ARG;
  NF_HIP(hipStreamSynchronize(ctx->stream));
  ctx->prof_events.clear();
  ctx->prof_pool_next = 0;
  if (mode && ctx->prof_pool.empty()) {
    ctx->prof_pool.resize(16384);
    for (auto &e : ctx->prof_pool) NF_HIP(hipEventCreate(&e));
  }
  ctx->prof_mode = mode;
  ctx->prof_tick = 0;
  return NF_OK;
}

extern "C" int nf_prof_read(nf_ctx *ctx, const char *kernel_name, double *avg_ms_host, int64_t *count_host) {
  if (!ctx || !kernel_name || !avg_ms_host) return NF_ERR_ARG;
  NF_HIP(hipStreamSynchronize(ctx->stream));
  auto it = ctx->prof_events.find(kernel_name);
  if (it == ctx->prof_events.end() || it->second.empty()) {
    *avg_ms_host = 0.0;
    if (count_host) *count_host = 0;
    return NF_OK;
  }
  double tot = 0.0;
  for (auto &p : it->second) {
    float ms = 0.f;
    NF_HIP(hipEventElapsedTime(&ms, p.first, p.second));
    tot += ms;
  }
  *avg_ms_host = tot / (double)it->second.size();
  if (count_host) *count_host = (int64_t)it->second.size();
  return NF_OK;
}

// in-kernel timeline of the coupling reverse pass (block 0 / wave 0), for kernel tuning only
extern "C" int nf_debug_trace(nf_ctx *ctx, int32_t on, int64_t *stamps_host, int32_t n) {
  if (!ctx) return NF_ERR_ARG;
  NF_HIP(hipStreamSynchronize(ctx->stream));
  if (stamps_host && ctx->trace && n > 0)
    NF_HIP(hipMemcpy(stamps_host, ctx->trace, sizeof(int64_t) * (size_t)(n < 128 ? n : 128), hipMemcpyDeviceToHost));
  if (on && !ctx->trace) {
    NF_HIP(hipMalloc(&ctx->trace, 128 * sizeof(int64_t)));
    NF_HIP(hipMemset(ctx->trace, 0, 128 * sizeof(int64_t)));
  } else if (!on && ctx->trace) {
    NF_HIP(hipFree(ctx->trace));
    ctx->trace = nullptr;
  }
  return NF_OK;
}
