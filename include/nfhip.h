/* nfhip.h -- C ABI of libnfhip.so: the MI355X (gfx950) implementation of the
 * ELBO / reverse-KL hot path of TuringLang/NormalizingFlows.jl.
 *
 * This is the drop-in boundary.  The reference is pure
 * Julia with no FFI of its own; the seams it offers for a device back end are
 * generic functions selected by dispatch.  Every entry point below names the
 * reference interface it stands behind (paths relative to the reference
 * checkout).  The Julia-side binding (`ccall`) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - Every function returns an int status: 0 = ok, > 0 = hipError_t of a failed
 *     HIP call, < 0 = NF_ERR_* argument / capability error.  Nothing throws or
 *     aborts; the host wrapper turns non-zero into error(nf_strerror(code)).
 *   - All array pointers are DEVICE pointers owned by the caller (AMDGPU.jl
 *     ROCArray -> pointer(A); torch tensor -> data_ptr()), unless the parameter
 *     name ends in _host.
 *   - A batch is d x N column-major with one sample per column, i.e. x[j*d + i]
 *     (reference: src/objectives/elbo.jl:52,60; src/flows/realnvp.jl:29).
 *   - theta is the flat parameter vector produced by Optimisers.destructure(flow)
 *     (src/NormalizingFlows.jl:67); its layout is documented in DESIGN.md and
 *     matches oracle/nf_oracle.py:layers_flat_order.
 *   - Work is enqueued on the context's HIP stream and is asynchronous unless
 *     the function returns a host scalar.
 *   - A context is not thread-safe (the reference's caller is one Julia task).
 */
#ifndef NFHIP_H
#define NFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NF_ABI_VERSION 4

/* status codes (< 0: library errors; > 0: hipError_t) */
#define NF_OK 0
#define NF_ERR_ARG -1          /* null pointer, negative size, bad enum            */
#define NF_ERR_UNSUPPORTED -2  /* flow shape / dtype not built into this library   */
#define NF_ERR_NO_DEVICE -3    /* no gfx950 device visible                         */
#define NF_ERR_NONFINITE -4    /* loss or gradient norm became non-finite (nf_elbo_step; the reference's
                                  tests require finite ELBOs, test/flow.jl:58-60)  */
#define NF_ERR_WORKSPACE -7    /* the caller-provided arena (nf_ctx_set_arena) is too small         */
#define NF_ERR_NO_RCCL -5      /* librccl.so.1 could not be loaded (multi-GPU entry points only) */
#define NF_ERR_RCCL -6         /* an RCCL call failed (nf_strerror gives RCCL's message), or a step left this rank's collective
                                  sequence incomplete (nf_elbo_step with bucketed all-reduce: fewer messages issued than
                                  nf_comm_bucket_count promises -- the peers wait in RCCL).  Either way the context's
                                  communicator is unusable: every later collective call returns this code until
                                  nf_comm_destroy + a new nf_comm_init_rank on EVERY rank.  Any other code from nf_elbo_step
                                  under a communicator is a local failure with a complete collective sequence.          */

/* flow kinds: the constructors of src/flows/*.jl */
#define NF_KIND_PLANAR 0    /* planarflow  src/flows/planar_radial.jl:21-29        */
#define NF_KIND_RADIAL 1    /* radialflow  src/flows/planar_radial.jl:52-60        */
#define NF_KIND_REALNVP 2   /* realnvp     src/flows/realnvp.jl:170-180            */
#define NF_KIND_NSF 3       /* nsf         src/flows/neuralspline.jl:218-234       */
#define NF_KIND_MEANFIELD 4 /* Shift o Scale, test/interface.jl:22-25              */
#define NF_KIND_HAMILTONIAN 5 /* mean-field reference + (momentum Shift o Scale) o LeapFrog blocks on the
                                 joint [x; rho], example/demo_hamiltonian_flow.jl:27-146: d = 2 * dims,
                                 nlayers = blocks, K = leapfrog steps per block, score = target whose
                                 score drives the integrator; targets passed to the ELBO entry points
                                 describe x, the library adds log N(rho; 0, I) (logp_joint, demo :121-128).
                                 theta = [shift0(d); scale0(d); per block: shift_rho(d/2), scale_rho(d/2),
                                 log_eps(d/2)] -- the reference distribution's affine map FIRST (destructure of
                                 a TransformedDistribution whose `dist` is q0 = transformed(MvNormal, Shift o
                                 Scale), demo :135-137).  nf_param_count cannot detect a permuted theta: bind
                                 by name, see INTEGRATION.md "theta order of the Hamiltonian demo flow" */

#define NF_KIND_COMPOSITE 6 /* create_flow((L1, ..., Ln), q0) with MIXED bijector families (src/flows/utils.jl:23-26:
                               any list of bijectors composes).  `segments` lists the maximal runs of one family in FLAT
                               order (segment 0 = outermost = applied last), each a descriptor of one of the kinds above
                               with the same d / dtype; theta = the segments' thetas concatenated in that order, which is
                               the order Optimisers.destructure walks the composition.  Forward, inverse, per-layer
                               application, rand, elbo / elbo_batch, the training step, loglikelihood, forward-KL
                               training and nf_flow_bwd are all built (reverse passes chain the segments' own). */

#define NF_DTYPE_F32 0
#define NF_DTYPE_F64 1

#define NF_TARGET_DIAGGAUSS 0 /* MvNormal(mu, Diagonal(var)), test/flow.jl:43-46        */
#define NF_TARGET_BANANA 1    /* Banana(d, b, var), example/targets/banana.jl:58-83     */
#define NF_TARGET_FUNNEL 2    /* Funnel(d, mu, sigma), example/targets/neal_funnel.jl:53-72 (s0 = mu, s1 = sigma) */
#define NF_TARGET_WARPED 3    /* WarpedGauss(s1, s2), d = 2, example/targets/warped_gaussian.jl:51-87 (s0, s1) */
#define NF_TARGET_CROSS 4     /* Cross(mu, sigma), d = 2, example/targets/cross.jl:30-37 (s0 = mu, s1 = sigma) */

#define NF_MAX_HIDDEN 4

/* Base distribution q0 of a flow (the `dist` of the TransformedDistribution).  Every reference configuration uses
 * MvNormal(zeros(d), I) (test/flow.jl:9, demo_planar_flow.jl:24) -- that is base == NULL, and the only form the fused
 * kernels draw in registers.  General MvNormal(mu, Sigma) bases (what _device_specific_rand(rng, ::MvNormal, n) and
 * logpdf(flow.dist, xs) accept: src/NormalizingFlows.jl:109-115, ext/NormalizingFlowsCUDAExt.jl:43-48,
 * test/ext/CUDA/cuda.jl:33-45) are drawn as x = mu + L eps and enter the objectives through an exact per-sample
 * correction of log q0; q0 is a leaf of destructure (@leaf MvNormal), so it carries no trainable parameter. */
#define NF_BASE_STANDARD 0 /* MvNormal(zeros(d), I)                                                    */
#define NF_BASE_DIAG 1     /* MvNormal(mu, Diagonal(sigma.^2)): scale = sigma[d]  (device)              */
#define NF_BASE_DENSE 2    /* MvNormal(mu, Sigma), Sigma = L L': scale = L, d x d lower triangular,
                              column-major (device)                                                     */
typedef struct nf_base {
  int32_t kind;      /* NF_BASE_*                                     */
  const void *mu;    /* [d], device, the flow's element type          */
  const void *scale; /* see NF_BASE_*                                 */
  double logdet;     /* log|det L| = sum(log(diag(L))) / sum(log(sigma)) */
} nf_base;

/* Static (non-trainable) description of a flow: the fields of the reference's
 * layer structs that Optimisers.destructure leaves out (dim, mask, K, B, hidden
 * sizes; src/flows/realnvp.jl:33-38, src/flows/neuralspline.jl:35-42). */
struct nf_target;
struct nf_base;
typedef struct nf_flow_desc {
  int32_t kind;                 /* NF_KIND_*                                         */
  int32_t dtype;                /* NF_DTYPE_*                                        */
  int32_t d;                    /* length(q0)                                        */
  int32_t nlayers;              /* planar/radial: layers; realnvp/nsf: RealNVP_layer /
                                   NSF_layer blocks (two couplings each)             */
  int32_t n_hidden;             /* length(hdims)                                     */
  int32_t hdims[NF_MAX_HIDDEN]; /* conditioner hidden widths, src/flows/utils.jl:71  */
  int32_t K;                    /* spline bins (nsf)                                 */
  float B;                      /* spline box bound (nsf)                            */
  const struct nf_target *score; /* NF_KIND_HAMILTONIAN: the target behind LeapFrog's
                                    score function (host pointer); NULL otherwise     */
  const struct nf_base *base;    /* q0 (host pointer); NULL = MvNormal(zeros(d), I)  */
  int32_t nsegments;             /* NF_KIND_COMPOSITE: number of segments, else 0    */
  const struct nf_flow_desc *segments; /* NF_KIND_COMPOSITE: host array [nsegments]  */
} nf_flow_desc;

/* Built-in target log-densities (the `logp` closure of src/objectives/elbo.jl:68
 * for the benchmark/test targets).  p0/p1 are device pointers for DIAGGAUSS
 * (mu[d], var[d]); for BANANA they are ignored and (b, var) are the scalars. */
typedef struct nf_target {
  int32_t kind;
  const void *p0;
  const void *p1;
  double s0;
  double s1;
} nf_target;

typedef struct nf_ctx nf_ctx;

/* ---- library / context --------------------------------------------------- */
int nf_abi_version(void);
const char *nf_strerror(int code);
/* hip_stream: a hipStream_t to enqueue on (NULL = the device's default stream). */
int nf_ctx_create(int device, void *hip_stream, nf_ctx **out);
int nf_ctx_destroy(nf_ctx *ctx);
int nf_ctx_set_stream(nf_ctx *ctx, void *hip_stream);
int nf_ctx_synchronize(nf_ctx *ctx);
/* Memory.  By default a context owns a grow-only arena: the first call of a larger shape allocates (and synchronises
 * the stream once), steady-state steps never do.  For callers that want NO allocation or synchronisation inside the
 * compute entry points (hipGraph capture, a Julia GC-managed ROCArray as backing store):
 *   bytes = nf_workspace_bytes(ctx, desc, N_max);  nf_ctx_set_arena(ctx, device_ptr, bytes);   (device_ptr 256-byte aligned)
 * after which every entry point with this flow and N <= N_max works inside the arena and a larger request returns
 * NF_ERR_WORKSPACE.  nf_ctx_set_arena(ctx, NULL, 0) returns to the owned mode.  ("the library allocates
 * nothing persistent except the ctx; arena sized by nf_workspace_bytes".) */
int64_t nf_workspace_bytes(nf_ctx *ctx, const nf_flow_desc *desc, int64_t N);
int nf_ctx_set_arena(nf_ctx *ctx, void *arena_device, size_t bytes);
/* The training step of the LDS-resident RealNVP path keeps the forward's activations for the reverse pass (an
 * "activation stash": 46 KiB per 32-sample tile and coupling at d = 64 / hidden 64, i.e. 772 MB at BASELINE cfg 2) --
 * the Zygote tape of src/optimize.jl:12-14 in kernel form.  max_bytes bounds the buffer: a batch whose stash is larger
 * runs chunk by chunk through it (forward + reverse pass per chunk, all chunks' gradient slabs reduced together; same
 * result up to float32 summation order).  nf_elbo_value_and_grad / nf_elbo_step and nf_loglikelihood_value_and_grad
 * (the inverse chain stashes for ITS reverse pass) use it.  0 disables the stash: the reverse pass then recomputes the
 * activations from the flow output (invertible recompute: no extra memory, but every leaky-ReLU slope is decided
 * again on a float32 reconstruction of the layer input -- DESIGN.md section 5; an explicit opt-in, never the default).
 * A negative value restores the default: 4 GiB (or the environment's NF_AFFINE_STASH_MAX_MB / NF_AFFINE_NO_STASH) for
 * every LDS-resident shape.  nf_workspace_bytes and nf_tape_bytes reflect the setting. */
int nf_ctx_set_stash_budget(nf_ctx *ctx, int64_t max_bytes);

/* ---- layout -------------------------------------------------------------- */
/* length(first(Optimisers.destructure(flow)))  (src/NormalizingFlows.jl:67) */
int64_t nf_param_count(const nf_flow_desc *desc);
/* number of bijector layers in execution terms (couplings count individually) */
int32_t nf_layer_count(const nf_flow_desc *desc);

/* ---- a4 + a5: base distribution ------------------------------------------ */
/* _device_specific_rand(rng, MvNormal(zeros(d), I), N) fused with
 * logpdf(flow.dist, xs)  (src/NormalizingFlows.jl:94-115, ext/NormalizingFlowsCUDAExt.jl:43-48,
 * src/objectives/elbo.jl:68,94).  Philox4x32-10 keyed by `seed`, counter =
 * (sample_offset + j, feature_group, stream_id): the draw for global sample j
 * does not depend on how the batch is sharded.  logq_out may be NULL. */
int nf_base_sample_logpdf(nf_ctx *ctx, int32_t dtype, int32_t d, int64_t N, uint64_t seed,
                          uint64_t sample_offset, uint32_t stream_id, void *x_out, void *logq_out);
/* logpdf(MvNormal(zeros(d), I), xs) for caller-supplied xs */
int nf_base_logpdf(nf_ctx *ctx, int32_t dtype, int32_t d, int64_t N, const void *x, void *logq_out);
/* The same two for a general MvNormal(mu, Sigma) base (base == NULL or NF_BASE_STANDARD: identical to the above):
 * x = mu + L eps with the eps of nf_base_sample_logpdf(seed, sample_offset, stream_id), and its log-density. */
int nf_base_rand(nf_ctx *ctx, int32_t dtype, const nf_base *base, int32_t d, int64_t N, uint64_t seed,
                 uint64_t sample_offset, uint32_t stream_id, void *x_out, void *logq_out);
int nf_base_logpdf_general(nf_ctx *ctx, int32_t dtype, const nf_base *base, int32_t d, int64_t N, const void *x,
                           void *logq_out);

/* ---- a6, a7, a10, a12, a13: transforms ----------------------------------- */
/* Bijectors.with_logabsdet_jacobian(flow.transform, xs)  (src/objectives/elbo.jl:67):
 * applies the layers last-listed first and sums per-sample logdets.
 * y_out may alias x_in.  ladj_out[N] is overwritten. */
int nf_flow_fwd(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x_in,
                int64_t N, void *y_out, void *ladj_out);
/* with_logabsdet_jacobian(inverse(flow.transform), ys)  (src/flows/realnvp.jl:99-110,
 * src/flows/neuralspline.jl:134-140; reached from loglikelihood.jl:31 via logpdf). */
int nf_flow_inv(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *y_in,
                int64_t N, void *x_out, void *ladj_out);
/* rand(rng, flow, n) / _device_specific_rand(rng, flow, n)  (src/NormalizingFlows.jl:117-127; the CUDA extension's
 * per-column version: ext/NormalizingFlowsCUDAExt.jl:61-74): N base draws -- the same Philox stream
 * nf_base_sample_logpdf(seed, sample_offset, stream_id) produces -- pushed through the transform, batched.
 * Planar / radial / mean-field flows do it in one launch (draws in registers); y_out is d x N. */
int nf_flow_rand(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, int64_t N, uint64_t seed,
                 uint64_t sample_offset, uint32_t stream_id, void *y_out);
/* One bijector layer, `layer` in FLAT order (0 = L1, the outermost / last applied):
 * Bijectors.with_logabsdet_jacobian(layer, x)  (src/flows/realnvp.jl:77-83,
 * src/flows/neuralspline.jl:102-108).  inverse != 0 selects Inverse{layer}.
 * ladj_out[N] is overwritten. */
int nf_layer_apply(nf_ctx *ctx, const nf_flow_desc *desc, int32_t layer, int32_t inverse,
                   const void *theta, const void *x_in, int64_t N, void *y_out, void *ladj_out);

/* Forward pass that keeps its tape, and the pullback from that tape: the two halves of a ChainRules
 * rrule for with_logabsdet_jacobian(flow.transform, xs) (the mechanism MonotonicSplines uses, test/ad.jl:126-127;
 * the reference's default path is Zygote differentiating the forward's own tape: src/optimize.jl:12-14 on
 * src/objectives/elbo.jl:65-70).  An arbitrary `logp` closure trains through this pair:
 *   tape = device buffer of nf_tape_bytes(ctx, desc, N) bytes, 256-byte aligned, owned by the caller (the rrule's closure)
 *   nf_flow_fwd_keep(..., x, N, y, ladj, tape, bytes)       y, ladj as nf_flow_fwd; the tape holds the forward's
 *                                                            activations in the layouts the reverse kernels consume
 *   ybar = d loss / d y from the caller's AD of logp, lbar = d loss / d ladj
 *   nf_flow_bwd_kept(..., tape, bytes, ybar, lbar, N, xbar, gtheta)
 * The pullback differentiates the forward's OWN activations and leaky-ReLU slopes (nothing is re-derived by inverting
 * the flow in float32), leaves the tape intact (it may be called again) and costs what the built-in training step's
 * reverse pass costs.  nf_tape_bytes depends on the context's nf_ctx_set_stash_budget setting (0 = keep only the flow
 * output and recompute by inversion; a positive budget -- or the default 4 GiB -- smaller than the activations of THIS
 * batch does the same for it: the tape never exceeds the budget, callers who want the forward's own activations for a
 * larger batch split it): both calls must run under the setting the size was queried with.
 * y_out may alias x_in; xbar_out may alias ybar, or be NULL for a single-family float32 coupling flow on the MFMA
 * kernels when only the parameter gradient is wanted (NF_ERR_ARG otherwise); gtheta_out[P] is overwritten. */
int64_t nf_tape_bytes(nf_ctx *ctx, const nf_flow_desc *desc, int64_t N);
int nf_flow_fwd_keep(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x_in, int64_t N,
                     void *y_out, void *ladj_out, void *tape, size_t tape_bytes);
int nf_flow_bwd_kept(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *tape, size_t tape_bytes,
                     const void *ybar, const void *lbar, int64_t N, void *xbar_out, void *gtheta_out);

/* Pullback of nf_flow_fwd for callers that kept only x: runs the forward again FROM x with the tape in the context
 * workspace, then pulls it back (y is part of the rrule's signature and is not read; no float32 inversion).
 * Inputs: x (flow input), y (flow output), ybar[d*N], lbar[N] (cotangent of ladj).
 * Outputs: xbar_out[d*N] (may alias ybar), gtheta_out[P] (overwritten). */
int nf_flow_bwd(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *x,
                const void *y, const void *ybar, const void *lbar, int64_t N, void *xbar_out,
                void *gtheta_out);

/* ---- a14: built-in targets ------------------------------------------------ */
/* logp(ys) per column and (optionally) its gradient w.r.t. ys. */
int nf_target_logp(nf_ctx *ctx, int32_t dtype, const nf_target *target, int32_t d, int64_t N,
                   const void *y, void *logp_out, void *grad_out);

/* ---- a1, a2, a3: objectives ------------------------------------------------ */
/* elbo_batch(flow, logp, xs) / elbo(flow, logp, xs)  (src/objectives/elbo.jl:31-34,89-92):
 * mean_j[logp(y_j) - log q0(x_j) + ladj_j] for caller-supplied xs.
 * elbos_out[N] (optional) receives the per-sample terms (_batched_elbos, :65-70). */
int nf_elbo_batch(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target,
                  const void *theta, const void *xs, int64_t N, void *elbos_out,
                  double *elbo_host);
/* elbo_batch(rng, flow, logp, n)  (src/objectives/elbo.jl:93-97): draws xs in-library. */
int nf_elbo_batch_rng(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target,
                      const void *theta, int64_t N, uint64_t seed, uint64_t sample_offset,
                      uint32_t stream_id, double *elbo_host);
/* a16: loglikelihood(rng, flow, ys)  (src/objectives/loglikelihood.jl:26-33):
 * mean_j[log q0(T^-1 y_j) + ladj_inv_j]; logliks_out[N] optional. */
int nf_loglikelihood(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta, const void *ys,
                     int64_t N, void *logliks_out, double *ll_host);

/* ---- a15: the training step ------------------------------------------------ */
/* loss(theta) = -elbo_batch(rng, re(theta), logp, n) and its gradient
 * (_value_and_gradient, src/optimize.jl:12-14,86; loss closure src/NormalizingFlows.jl:69)
 * for THIS rank's shard of a global batch:
 *   samples [sample_offset, sample_offset + N_local) of N_global,
 *   out[0..P)  = sum_{j in shard} d(-elbo_j / N_global)/dtheta,
 *   out[P]     = sum_{j in shard} (-elbo_j / N_global).
 * Summing `out` over ranks (one all-reduce of P+1 elements) gives (grad, loss).
 * xs may be NULL (draw in-library from Philox as nf_base_sample_logpdf does) or a
 * caller-supplied d x N_local batch (the elbo_batch(flow, logp, xs) form). */
int nf_elbo_value_and_grad(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target,
                           const void *theta, const void *xs, int64_t N_local, int64_t N_global,
                           uint64_t seed, uint64_t sample_offset, uint32_t stream_id,
                           void *out_grad_loss);
/* Forward-KL training, `train_flow(loglikelihood, flow, ys)`: loss(theta) =
 * -loglikelihood(rng, re(theta), ys) (src/objectives/loglikelihood.jl:26-33 inside the loss closure
 * src/NormalizingFlows.jl:69) and its gradient (_value_and_gradient, src/optimize.jl:12-14,86) for
 * THIS rank's N_local columns of an N_global-column data set:
 *   out[0..P) = sum_{j in shard} d(-loglik_j / N_global)/dtheta,   out[P] = sum (-loglik_j / N_global).
 * The chain is inverted once; its reverse pass walks the layers in forward order with the
 * implicit-function form of each inverse (no root-find or spline inversion is differentiated); the
 * Hamiltonian flow's inverse layers are explicit (LeapFrog with -eps) and are differentiated directly.
 * A general desc->base seeds the reverse pass with that base's score -Sigma^-1 (z - mu); a composite is
 * inverted and differentiated segment by segment. */
int nf_loglikelihood_value_and_grad(nf_ctx *ctx, const nf_flow_desc *desc, const void *theta,
                                    const void *ys, int64_t N_local, int64_t N_global,
                                    void *out_grad_loss);
/* Optimisers.update!(st, theta, g) for Optimisers.Adam (src/optimize.jl:99), in place on
 * theta/m/v; t = 1-based step count.  gnorm_out (device scalar, optional) receives
 * norm(g) (src/optimize.jl:89). */
int nf_adam_update(nf_ctx *ctx, int32_t dtype, void *theta, const void *g, void *m, void *v,
                   int64_t P, double lr, double beta1, double beta2, double eps, int64_t t,
                   void *gnorm_out);
/* Optimisers.Descent(lr) (vel = NULL: theta -= lr g) and Optimisers.Momentum(lr, rho)
 * (vel = rho vel - lr g; theta += vel) -- the other rules the reference's `optimiser` keyword
 * is commonly given (src/optimize.jl:67); gnorm_out as in nf_adam_update. */
int nf_sgd_update(nf_ctx *ctx, int32_t dtype, void *theta, const void *g, void *vel, int64_t P,
                  double lr, double rho, void *gnorm_out);
/* One whole iteration of the reference's training loop (src/optimize.jl:85-99: value and gradient of
 * -elbo_batch(rng, re(theta), logp, n), Optimisers.update! with Adam, norm(g)) in one call; loss_host / gnorm_host
 * (optional) receive the stat tuple of src/optimize.jl:89 -- asking for them synchronises the stream, passing NULL for
 * both keeps the call asynchronous ([loss ; norm] of the last step stay readable through a later call that asks).
 * Philox stream id = `step`, Adam's t = step + 1.  With a communicator on the context (nf_comm_init_rank /
 * nf_comm_init_all) this is the data-parallel step: N is THIS rank's batch, the rank draws samples
 * [rank * N, (rank + 1) * N) of a global batch of N * nranks, and the one all-reduce of [grad ; loss] happens inside.
 * LDS-resident RealNVP flows with a diagonal-Gaussian target (BASELINE cfg 2) run as three launches -- fused forward,
 * reverse pass, fused epilogue (slab sum, loss, Adam, norm, and the packed weight images of the UPDATED theta for the
 * next step).  BY DEFAULT every call packs its weight images from theta (one small launch): whatever happened to theta
 * between two calls -- an in-place edit, a free and a re-allocation at the same address -- the step runs on the weights
 * theta holds now.  A training loop that OWNS theta (src/optimize.jl:85-99 does: nobody else touches theta between
 * Optimisers.update! and the next gradient) may opt in to reusing the images the previous step's epilogue wrote:
 *   nf_ctx_set_weight_cache(ctx, 1)  -- the caller promises that between consecutive nf_elbo_step /
 *       nf_elbo_step_enqueue calls on this context with the same theta pointer and flow, theta is modified by nobody
 *       else; the cache is keyed on (theta pointer, hash of the descriptor's shape fields) and dropped by
 *       nf_ctx_weights_changed, nf_ctx_set_stream, nf_ctx_set_arena, nf_ctx_set_weight_cache itself and by every other
 *       library call that packs weights;
 *   nf_ctx_weights_changed(ctx)      -- declares an edit of theta made under that promise (clipping, a callback).
 * `train_flow` of the Python mirror (objectives._optimize_fused) and `train_flow_fused` of ext/NormalizingFlowsNFHipExt.jl
 * opt in for the duration of their loop and opt out on return. */
int nf_elbo_step(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, void *theta,
                 void *m, void *v, int64_t N, uint64_t seed, uint32_t step, double lr,
                 double beta1, double beta2, double eps, double *loss_host, double *gnorm_host);
int nf_ctx_weights_changed(nf_ctx *ctx);
int nf_ctx_set_weight_cache(nf_ctx *ctx, int32_t enable);
/* The same step with NO per-step host values, for hipGraph capture and replay: *step_device (uint32, device memory,
 * caller-owned, initialised to the first step index) supplies the Philox stream id and Adam's t - 1 and is incremented
 * by the step; out_loss_gnorm_device (optional, 2 elements of the flow's type) receives [loss ; norm(g)].  After one
 * warm-up call (workspace sizing and kernel attributes are not capturable; use nf_ctx_set_arena or the warm-up's
 * grow-only allocation), capture a call between hipStreamBeginCapture / hipStreamEndCapture on the context's stream
 * and replay the graph.  NF_ERR_UNSUPPORTED for flows without the three-launch form. */
int nf_elbo_step_enqueue(nf_ctx *ctx, const nf_flow_desc *desc, const nf_target *target, void *theta, void *m, void *v,
                         int64_t N, uint64_t seed, uint32_t *step_device, double lr, double beta1, double beta2,
                         double eps, void *out_loss_gnorm_device);

/* ---- (e) multi-GPU: the path's one collective ----------------------------------- */
/* The ELBO is a mean over independent draws (src/objectives/elbo.jl:68,91,96), so ranks take sample shards
 * (nf_elbo_value_and_grad's sample_offset / N_local / N_global) and the ONLY exchange per step is an in-place SUM
 * all-reduce of the packed [grad ; loss] buffer (P + 1 elements) -- RCCL over xGMI, enqueued on the context's
 * stream, so it is ordered after the reverse pass and before nf_adam_update without host synchronisation.  Every
 * rank then applies the identical Adam update: replicas stay bit-identical, no broadcast.  The reference has no
 * multi-device code to cite; this is north_star's "single RCCL all-reduce over xGMI of the scalar ELBO and
 * parameter gradients per step" (SURVEY.md 8e).  librccl is bound at run time (dlopen), see NF_ERR_NO_RCCL.
 *
 * One process (or thread) per GPU:  rank 0 calls nf_comm_get_unique_id, ships the NF_COMM_ID_BYTES bytes to the
 * other ranks by any host channel, every rank calls nf_comm_init_rank.
 * One process driving G contexts (the Julia form: one task, G devices): nf_comm_init_all + nf_allreduce_grad_loss_all. */
#define NF_COMM_ID_BYTES 128
int nf_comm_get_unique_id(void *id_out_host);
int nf_comm_init_rank(nf_ctx *ctx, const void *id_host, int32_t nranks, int32_t rank);
int nf_comm_init_all(nf_ctx **ctxs, int32_t ngpus);
int nf_comm_size(nf_ctx *ctx); /* ranks of the context's communicator (1 if none) */
/* The step's one logical all-reduce may travel as several messages: under a communicator nf_elbo_step sends the gradient
 * of a weight-streaming RealNVP flow (BASELINE cfg 4: P + 1 = 4 214 785 floats, 16.9 MB) in buckets of whole couplings
 * (contiguous theta ranges, Optimisers.destructure order) on a second stream, each as soon as its couplings' reverse pass
 * and slab sum are done, and joins before the optimiser update; every rank issues the same buckets in the same order and
 * receives the same reduced bits.  bucket_bytes < 0: automatic (4 MiB, the default), 0: always one message, > 0: target
 * bucket size.  Gradients smaller than two buckets (cfg 2: 0.5 MB) always travel as one message.
 * nf_comm_bucket_count: all-reduce calls nf_elbo_step issues per step for this flow (0 without a communicator). */
int nf_ctx_set_comm_bucket_bytes(nf_ctx *ctx, int64_t bucket_bytes);
int nf_comm_bucket_count(nf_ctx *ctx, const nf_flow_desc *desc);
/* in place: buf[0..count) <- sum over ranks; count = P + 1 for the training step */
int nf_allreduce_grad_loss(nf_ctx *ctx, int32_t dtype, void *buf, int64_t count);
int nf_allreduce_grad_loss_all(nf_ctx **ctxs, int32_t ngpus, int32_t dtype, void **bufs, int64_t count);
int nf_comm_destroy(nf_ctx *ctx);

/* ---- measurement support ---------------------------------------------------- */
/* Kernel durations from HIP events recorded on the context stream around launches since the
 * last nf_prof_enable (used by bench.py's roofline object).  mode 0 = off, 1 = bracket only the
 * dominant kernel (the coupling reverse pass), 2 = bracket every kernel, 3 = bracket every 4th launch
 * of the dominant kernel (what bench.py uses inside its timed region). */
int nf_prof_enable(nf_ctx *ctx, int32_t mode);
int nf_prof_read(nf_ctx *ctx, const char *kernel_name, double *avg_ms_host, int64_t *count_host);
/* Kernel-tuning aid: when on, block 0 / wave 0 of the coupling reverse pass writes s_memtime
 * stamps at its phase boundaries; a later call copies up to 128 of them to stamps_host. */
int nf_debug_trace(nf_ctx *ctx, int32_t on, int64_t *stamps_host, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* NFHIP_H */
