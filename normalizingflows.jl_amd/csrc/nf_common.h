// nf_common.h -- context, status handling and launch helpers shared by the
// translation units of libnfhip.so (gfx950 only).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/nfhip.h"

#define NF_HIP(call)                         \
  do {                                       \
    hipError_t e__ = (call);                 \
    if (e__ != hipSuccess) return (int)e__;  \
  } while (0)

#define NF_TRY(call)            \
  do {                          \
    int s__ = (call);           \
    if (s__ != NF_OK) return s__; \
  } while (0)

struct nf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cu = 256;
  // grow-only device arena for intermediates (never freed until destroy, so steady-state
  // steps allocate nothing)
  void *ws = nullptr;
  size_t ws_bytes = 0;
  // pinned host scalars for results that are returned by value
  double *host_scratch = nullptr;
  // packed (padded) LDS images of the conditioner nets, rebuilt from theta once per API call
  void *wimg = nullptr;
  size_t wimg_bytes = 0;
  // optional in-kernel s_memtime trace (nf_debug_trace): 128 slots, device memory
  void *trace = nullptr;
  // device gradient buffer of nf_elbo_step (P + 2 elements), grow-only
  void *gbuf = nullptr;
  size_t gbuf_bytes = 0;
  // per-kernel HIP-event timing (bench.py roofline): name -> list of (start, stop)
  int prof_mode = 0;
  unsigned prof_tick = 0;
  std::vector<hipEvent_t> prof_pool;  // pre-created events, reused
  size_t prof_pool_next = 0;
  std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> prof_events;
  // caller-provided arena (nf_ctx_set_arena): when set, NOTHING is allocated or freed by compute entry points; the
  // intermediates arena `ws` is its front, packed weight images and the nf_elbo_step buffer are carved off its tail
  void *arena = nullptr;
  size_t arena_bytes = 0, arena_tail = 0;
  size_t arena_front = 0;  // bytes the running entry point asked for at the front (nf_ws_reserve): tail carves stay behind it
  // > 0 while a wrapper entry point has buffers of its own behind the first ws_guard bytes of `ws`: inner requests
  // beyond the guard fail instead of overlapping them
  size_t ws_guard = 0;
  // activation-stash budget of the LDS-resident RealNVP training step (nf_ctx_set_stash_budget); -1: the default
  // (NF_AFFINE_STASH_MAX_MB or 4 GiB; 0 with NF_AFFINE_NO_STASH)
  long long stash_budget = -1;
  // nf_elbo_step keeps the packed images of the theta ITS epilogue wrote: valid while wimg_owner == that theta pointer
  // and wimg_sig == the flow's signature; every other pack, a reallocation of wimg and nf_ctx_weights_changed reset it
  const void *wimg_owner = nullptr;
  unsigned long long wimg_sig = 0;
  bool wimg_cache = false;  // nf_ctx_set_weight_cache: off by default (every nf_elbo_step packs from theta)
  // generation of the fp32 weight images (bumped by every writer) and the one the bf16-triple copies were built from
  unsigned long long wimg_gen = 1, b6_gen = 0, b6t_gen = 0;
  // RCCL communicator of this context (nf_comm.hip); null for single-GPU use
  void *comm = nullptr;
  int comm_size = 1, comm_rank = 0;
  // bucketed all-reduce (nf_comm.hip): the gradient of a flow with many parameters travels in buckets of whole couplings
  // on a second stream while the reverse pass of the next couplings runs (cfg 4, 16.9 MB per step)
  hipStream_t comm_stream = nullptr;
  std::vector<hipEvent_t> comm_events;  // pooled; [0] is the join event
  size_t comm_event_next = 1;
  long long comm_bucket_bytes = -1;     // nf_ctx_set_comm_bucket_bytes: < 0 automatic (4 MiB), 0 never bucket
  // set when a step left the communicator with a partly issued collective sequence (this rank issued fewer buckets than its
  // peers wait for): every later collective call returns NF_ERR_RCCL until nf_comm_destroy + a new nf_comm_init_rank on
  // every rank (a local gradient failure must be distinguishable from a communicator that has to be torn down)
  bool comm_poisoned = false;
  struct {
    bool on = false;      // set by nf_elbo_step around its gradient call
    int couplings = 0;    // couplings per bucket
    int issued = 0;       // all-reduces issued for the current step
  } bucket;
};

const char *nf_comm_last_error();
// bucketed form of the step's one logical all-reduce (nf_comm.hip): `count` elements at `buf` are final on ctx->stream now;
// the all-reduce runs on the context's second stream.  nf_comm_bucket_join makes ctx->stream wait for every issued bucket.
int nf_comm_bucket_issue(nf_ctx *ctx, int32_t dtype, void *buf, int64_t count);
int nf_comm_bucket_join(nf_ctx *ctx);

int nf_ws_reserve(nf_ctx *ctx, size_t bytes);
// packed weight images (ctx->wimg) of at least `bytes`: grow-only allocation, or a tail carve of the caller's arena
int nf_wimg_reserve(nf_ctx *ctx, size_t bytes);

// carve helper over the arena: returns 256-byte aligned sub-buffers
struct Carver {
  char *base;
  size_t off = 0;
  explicit Carver(void *b) : base((char *)b) {}
  template <class T>
  T *take(size_t n) {
    T *p = (T *)(base + off);
    off += ((n * sizeof(T) + 255) / 256) * 256;
    return p;
  }
};
inline size_t carve_bytes(size_t nbytes) { return ((nbytes + 255) / 256) * 256; }

// profiling bracket: records a pair of pooled HIP events on ctx->stream around a launch.
// prof_mode 1 brackets only the dominant kernel ("affine_bwd" / "rqs_bwd" / "wide_bwd"), 2 brackets all,
// 3 brackets every 4th launch of the dominant kernel (lowest perturbation of a timed region).
// roctx ranges around the library's launch sites (tracing): with NF_ROCTX=1 in the environment every
// ProfScope pushes / pops a named range, which `rocprofv3 --marker-trace --kernel-trace` shows next to the kernels.  The
// marker library is bound at run time (as librccl is in nf_comm.hip): no link-time dependency, nothing happens without it.
struct NfRoctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
};
inline const NfRoctx &nf_roctx() {
  static const NfRoctx r = [] {
    NfRoctx x;
    if (!std::getenv("NF_ROCTX")) return x;
    for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      x.push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
      x.pop = (int (*)())dlsym(h, "roctxRangePop");
      if (x.push && x.pop) break;
      x = NfRoctx();
    }
    return x;
  }();
  return r;
}

struct ProfScope {
  nf_ctx *ctx;
  hipEvent_t b = nullptr;
  bool ranged = false;
  ProfScope(nf_ctx *c, const char *name) : ctx(c) {
    if (nf_roctx().push) {
      nf_roctx().push(name);
      ranged = true;
    }
    if (!ctx->prof_mode) return;
    if (ctx->prof_mode != 2) {
      if (std::strcmp(name, "affine_bwd") != 0 && std::strcmp(name, "rqs_bwd") != 0 && std::strcmp(name, "wide_bwd") != 0)
        return;
      if (ctx->prof_mode == 3 && (ctx->prof_tick++ & 3) != 0) return;
    }
    if (ctx->prof_pool_next + 2 > ctx->prof_pool.size()) return;  // pool exhausted: stop sampling
    hipEvent_t a = ctx->prof_pool[ctx->prof_pool_next++];
    b = ctx->prof_pool[ctx->prof_pool_next++];
    hipEventRecord(a, ctx->stream);
    ctx->prof_events[name].push_back({a, b});
  }
  ~ProfScope() {
    if (b) hipEventRecord(b, ctx->stream);
    if (ranged) nf_roctx().pop();
  }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize, ...) once per (launcher, device).  A process may hold contexts
// on several devices (nf_ctx_create(device, ...)), possibly driven from several threads.
struct AttrOnce {
  std::mutex mu;
  unsigned long long done = 0;
  template <class F>
  int run(int device, F f) {
    std::lock_guard<std::mutex> g(mu);
    const unsigned long long bit = 1ull << (device & 63);
    if (done & bit) return NF_OK;
    const int st = f();
    if (st == NF_OK) done |= bit;
    return st;
  }
};

// natural log through v_log_f32 (log2, 1 ulp) and one multiply.  hipcc expands __logf into the denormal-safe,
// extended-precision sequence of logf (14 instructions); every argument on this path is a normal positive number.
#ifdef NF_LOG_PRECISE
__device__ __forceinline__ float nf_log(float x) { return __logf(x); }
#else
__device__ __forceinline__ float nf_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
#endif

// Optimisers.Adam (Optimisers.jl 0.4 `apply!`, reached from src/optimize.jl:99) for ONE parameter: shared by k_adam and the
// fused step epilogue so that both produce the same bits (contraction off: the compiler may otherwise fuse b1 m + (1 - b1) g
// differently at the two call sites).  c1 = 1 - b1^t, c2 = 1 - b2^t.
template <class T>
__device__ __forceinline__ void nf_adam_elem(T &theta, T &m, T &v, T g, T lr, T b1, T b2, T eps, T c1, T c2) {
#pragma clang fp contract(off)
  const T mi = b1 * m + ((T)1 - b1) * g;
  const T vi = b2 * v + ((T)1 - b2) * g * g;
  m = mi;
  v = vi;
  theta = theta - lr * (mi / c1) / (sqrt(vi / c2) + eps);
}

// layer bookkeeping shared by host code --------------------------------------------
struct CouplingInfo {
  long theta_off;   // offset of this coupling's parameters in theta
  long nparams;
  int par_t;        // transformed features are 2p + par_t  (mask 1:2:d -> 0, 2:2:d -> 1)
  int c, m;         // transformed / conditioner sizes
};

inline int nf_desc_h(const nf_flow_desc *d, int i) { return d->hdims[i]; }

// coupling `k` in FLAT order (k even: odd mask 1:2:d, k odd: even mask 2:2:d);
// reference: src/flows/realnvp.jl:138-144, src/flows/neuralspline.jl:176-183.
CouplingInfo nf_coupling_info(const nf_flow_desc *desc, int k);
