// nf_comm.hip -- the one collective of the path: an in-place sum all-reduce of the packed [grad ; loss]
// buffer (P + 1 elements) over RCCL / xGMI, enqueued on the context's stream.
//
// librccl is bound at run time (dlopen by soname), not at link time: the single-GPU product has no
// dependency on it, and inside a process that already loaded RCCL (PyTorch-ROCm bundles librccl.so.1)
// the same instance is shared.  Host code only; no kernels here.
#include <dlfcn.h>

#include <mutex>

#include "nf_common.h"

namespace {

struct rccl_unique_id {
  char internal[NF_COMM_ID_BYTES];
};
typedef void *rccl_comm_t;
enum { RCCL_SUM = 0, RCCL_FLOAT32 = 7, RCCL_FLOAT64 = 8 };  // ncclSum, ncclFloat32, ncclFloat64 (rccl.h)

struct RcclApi {
  int (*GetUniqueId)(rccl_unique_id *) = nullptr;
  int (*CommInitRank)(rccl_comm_t *, int, rccl_unique_id, int) = nullptr;
  int (*CommInitAll)(rccl_comm_t *, int, const int *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*CommCount)(rccl_comm_t, int *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};

RcclApi g_api;
std::once_flag g_once;
// per thread: the header allows one host thread per GPU, and concurrent RCCL failures must not mix their messages
thread_local char g_last_error[256] = "nfhip: librccl.so.1 not found";

void load_rccl() {
  void *h = nullptr;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return;
  auto sym = [&](const char *n) { return dlsym(h, n); };
  g_api.GetUniqueId = (decltype(g_api.GetUniqueId))sym("ncclGetUniqueId");
  g_api.CommInitRank = (decltype(g_api.CommInitRank))sym("ncclCommInitRank");
  g_api.CommInitAll = (decltype(g_api.CommInitAll))sym("ncclCommInitAll");
  g_api.AllReduce = (decltype(g_api.AllReduce))sym("ncclAllReduce");
  g_api.GroupStart = (decltype(g_api.GroupStart))sym("ncclGroupStart");
  g_api.GroupEnd = (decltype(g_api.GroupEnd))sym("ncclGroupEnd");
  g_api.CommDestroy = (decltype(g_api.CommDestroy))sym("ncclCommDestroy");
  g_api.CommCount = (decltype(g_api.CommCount))sym("ncclCommCount");
  g_api.GetErrorString = (decltype(g_api.GetErrorString))sym("ncclGetErrorString");
  g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommInitAll && g_api.AllReduce && g_api.GroupStart &&
             g_api.GroupEnd && g_api.CommDestroy;
}

int api() {
  std::call_once(g_once, load_rccl);
  return g_api.ok ? NF_OK : NF_ERR_NO_RCCL;
}

int rccl_status(int r) {
  if (r == 0) return NF_OK;
  if (g_api.GetErrorString) snprintf(g_last_error, sizeof(g_last_error), "nfhip: RCCL: %s", g_api.GetErrorString(r));
  return NF_ERR_RCCL;
}

int rccl_dtype(int dtype) { return dtype == NF_DTYPE_F64 ? RCCL_FLOAT64 : RCCL_FLOAT32; }

}  // namespace

const char *nf_comm_last_error() { return g_last_error; }

extern "C" int nf_comm_get_unique_id(void *id_out_host) {
  if (!id_out_host) return NF_ERR_ARG;
  NF_TRY(api());
  return rccl_status(g_api.GetUniqueId((rccl_unique_id *)id_out_host));
}

extern "C" int nf_comm_init_rank(nf_ctx *ctx, const void *id_host, int32_t nranks, int32_t rank) {
  if (!ctx || !id_host || nranks < 1 || rank < 0 || rank >= nranks || ctx->comm) return NF_ERR_ARG;
  NF_TRY(api());
  NF_HIP(hipSetDevice(ctx->device));
  rccl_unique_id id;
  memcpy(&id, id_host, sizeof(id));
  rccl_comm_t c = nullptr;
  NF_TRY(rccl_status(g_api.CommInitRank(&c, nranks, id, rank)));
  ctx->comm = c;
  ctx->comm_size = nranks;
  ctx->comm_rank = rank;
  return NF_OK;
}

extern "C" int nf_comm_init_all(nf_ctx **ctxs, int32_t ngpus) {
  if (!ctxs || ngpus < 1 || ngpus > 64) return NF_ERR_ARG;
  for (int i = 0; i < ngpus; ++i)
    if (!ctxs[i] || ctxs[i]->comm) return NF_ERR_ARG;
  NF_TRY(api());
  rccl_comm_t comms[64];
  int devs[64];
  for (int i = 0; i < ngpus; ++i) devs[i] = ctxs[i]->device;
  NF_TRY(rccl_status(g_api.CommInitAll(comms, ngpus, devs)));
  for (int i = 0; i < ngpus; ++i) {
    ctxs[i]->comm = comms[i];
    ctxs[i]->comm_size = ngpus;
    ctxs[i]->comm_rank = i;
  }
  return NF_OK;
}

extern "C" int nf_comm_size(nf_ctx *ctx) { return !ctx ? NF_ERR_ARG : ctx->comm ? ctx->comm_size : 1; }

extern "C" int nf_allreduce_grad_loss(nf_ctx *ctx, int32_t dtype, void *buf, int64_t count) {
  if (!ctx || !buf || count < 0) return NF_ERR_ARG;
  if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_F64) return NF_ERR_ARG;
  if (!ctx->comm) return NF_ERR_ARG;  // no communicator: a single-GPU caller simply does not call this
  if (ctx->comm_poisoned) return NF_ERR_RCCL;  // an earlier step left a partly issued bucket sequence behind
  NF_TRY(api());
  NF_HIP(hipSetDevice(ctx->device));
  return rccl_status(g_api.AllReduce(buf, buf, (size_t)count, rccl_dtype(dtype), RCCL_SUM, ctx->comm, ctx->stream));
}

extern "C" int nf_allreduce_grad_loss_all(nf_ctx **ctxs, int32_t ngpus, int32_t dtype, void **bufs, int64_t count) {
  if (!ctxs || !bufs || ngpus < 1 || count < 0) return NF_ERR_ARG;
  if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_F64) return NF_ERR_ARG;
  for (int i = 0; i < ngpus; ++i)
    if (!ctxs[i] || !ctxs[i]->comm || !bufs[i]) return NF_ERR_ARG;
  NF_TRY(api());
  NF_TRY(rccl_status(g_api.GroupStart()));
  int st = NF_OK;
  for (int i = 0; i < ngpus && st == NF_OK; ++i) {
    if (hipSetDevice(ctxs[i]->device) != hipSuccess) st = NF_ERR_ARG;
    if (st == NF_OK)
      st = rccl_status(g_api.AllReduce(bufs[i], bufs[i], (size_t)count, rccl_dtype(dtype), RCCL_SUM, ctxs[i]->comm, ctxs[i]->stream));
  }
  const int ge = rccl_status(g_api.GroupEnd());
  return st != NF_OK ? st : ge;
}

// ---- bucketed form ---------------------------------------------------------------------------------------
// The step's all-reduce of [grad ; loss] is ONE logical collective (every rank adds the same P + 1 numbers), but nothing
// forces it to be one message: the gradient of coupling k is final as soon as that coupling's reverse pass and slab sum are
// done, 15 couplings before the last one at cfg 4.  A bucket = the theta range of a few whole couplings (contiguous in
// Optimisers.destructure order); its all-reduce is issued on the context's SECOND stream behind an event, so xGMI moves it
// while the matrix pipe runs the next couplings, and the optimiser update waits for the join.  Every rank issues the same
// buckets in the same order (the schedule depends on the flow shape only), every rank receives the same reduced bits, so
// replicas stay bit-identical with each other.
int nf_comm_bucket_issue(nf_ctx *ctx, int32_t dtype, void *buf, int64_t count) {
  if (!ctx || !ctx->comm || !buf || count < 0) return NF_ERR_ARG;
  if (ctx->comm_poisoned) return NF_ERR_RCCL;
  NF_TRY(api());
  if (!ctx->comm_stream) NF_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
  if (ctx->comm_events.empty()) {
    ctx->comm_events.resize(65);
    for (auto &e : ctx->comm_events) NF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  if (ctx->comm_event_next >= ctx->comm_events.size()) ctx->comm_event_next = 1;  // 64 buckets in flight at most; a step has <= 32
  hipEvent_t ready = ctx->comm_events[ctx->comm_event_next++];
  NF_HIP(hipEventRecord(ready, ctx->stream));
  NF_HIP(hipStreamWaitEvent(ctx->comm_stream, ready, 0));
  ctx->bucket.issued++;
  return rccl_status(g_api.AllReduce(buf, buf, (size_t)count, rccl_dtype(dtype), RCCL_SUM, ctx->comm, ctx->comm_stream));
}

int nf_comm_bucket_join(nf_ctx *ctx) {
  if (!ctx) return NF_ERR_ARG;
  if (!ctx->comm_stream || ctx->comm_events.empty()) return NF_OK;
  NF_HIP(hipEventRecord(ctx->comm_events[0], ctx->comm_stream));
  NF_HIP(hipStreamWaitEvent(ctx->stream, ctx->comm_events[0], 0));
  return NF_OK;
}

extern "C" int nf_ctx_set_comm_bucket_bytes(nf_ctx *ctx, int64_t bytes) {
  if (!ctx) return NF_ERR_ARG;
  ctx->comm_bucket_bytes = bytes;
  return NF_OK;
}

extern "C" int nf_comm_destroy(nf_ctx *ctx) {
  if (!ctx) return NF_ERR_ARG;
  if (ctx->comm_stream) {
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->comm_stream);
    for (auto &e : ctx->comm_events) hipEventDestroy(e);
    ctx->comm_events.clear();
    hipStreamDestroy(ctx->comm_stream);
    ctx->comm_stream = nullptr;
  }
  if (!ctx->comm) return NF_OK;
  NF_TRY(api());
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  const int r = g_api.CommDestroy(ctx->comm);
  ctx->comm = nullptr;
  ctx->comm_size = 1;
  ctx->comm_rank = 0;
  ctx->comm_poisoned = false;
  return rccl_status(r);
}
