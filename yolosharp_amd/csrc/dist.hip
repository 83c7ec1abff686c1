// dist.hip -- data-parallel exchange step behind the C ABI: a SUM all-reduce of the flat fp32 gradient
// buffer over RCCL (xGMI), issued per backward segment on a communication stream so that it overlaps the backward of the
// next segment.  A C# host cannot use torch.distributed; this gives it the same path bench.py drives through
// yolosharp_amd/dist.py.  RCCL is opened lazily with dlopen (librccl.so) -- the library has no link-time dependency on it and a
// single-GPU process never loads it.
#include "ys_internal.h"
#include <cstring>
#ifndef YS_EMU_BUILD
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.h) return YS_OK;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
  if (!h) { ys_set_error("ys_dist: cannot open librccl.so (%s)", dlerror()); return YS_ERR_UNSUPPORTED; }
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce) {
    ys_set_error("ys_dist: librccl.so lacks the expected symbols");
    dlclose(h);
    return YS_ERR_UNSUPPORTED;
  }
  g_rccl.h = h;
  return YS_OK;
}
#define YS_CHECK_NCCL(call)                                                                                      \
  do {                                                                                                           \
    ncclResult_t r_ = (call);                                                                                    \
    if (r_ != ncclSuccess) {                                                                                     \
      ys_set_error("%s -> %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error");        \
      return YS_ERR_HIP;                                                                                         \
    }                                                                                                            \
  } while (0)
}  // namespace
#endif

// implemented in model.hip
int ys_model_segment_grad_range(ys_model* m, int seg, int64_t* offset, int64_t* count);
int ys_model_grad_buffer(ys_model* m, float** dptr, int64_t* count);
ys_ctx* ys_model_ctx(ys_model* m);

extern "C" {

int ys_dist_unique_id(void* id128) {
#ifdef YS_EMU_BUILD
  (void)id128; ys_set_error("ys_dist: not available in the interpreter build"); return YS_ERR_UNSUPPORTED;
#else
  YS_REQUIRE(id128, "ys_dist_unique_id: null argument");
  YS_TRY(rccl_load());
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  YS_CHECK_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return YS_OK;
#endif
}

int ys_dist_init(ys_ctx* ctx, int rank, int world, const void* id128) {
#ifdef YS_EMU_BUILD
  (void)ctx; (void)rank; (void)world; (void)id128; ys_set_error("ys_dist: not available in the interpreter build"); return YS_ERR_UNSUPPORTED;
#else
  YS_REQUIRE(ctx && id128 && world >= 1 && rank >= 0 && rank < world, "ys_dist_init: bad argument (rank %d of %d)", rank, world);
  YS_REQUIRE(!ctx->dist_comm, "ys_dist_init: this context already has a communicator");
  YS_TRY(rccl_load());
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  // the context is touched only when every resource exists: a failure half way destroys what was created and leaves it unchanged
  ncclComm_t comm = nullptr;
  YS_CHECK_NCCL(g_rccl.CommInitRank(&comm, world, id, rank));
  hipStream_t cs = nullptr; hipEvent_t e_ready = nullptr, e_done = nullptr;
  hipError_t he = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipEventCreateWithFlags(&e_ready, hipEventDisableTiming);
  if (he == hipSuccess) he = hipEventCreateWithFlags(&e_done, hipEventDisableTiming);
  if (he != hipSuccess) {
    if (e_done) hipEventDestroy(e_done);
    if (e_ready) hipEventDestroy(e_ready);
    if (cs) hipStreamDestroy(cs);
    g_rccl.CommDestroy(comm);
    ys_set_error("ys_dist_init: %s", hipGetErrorString(he));
    return YS_ERR_HIP;
  }
  // One collective right away (the C-path equivalent of bench.py's dist.barrier() after init_process_group): RCCL creates its internal
  // streams / proxy resources lazily at the first collective, and hardware-queue assignment follows stream creation order.  With the
  // communicator fully built BEFORE the model creates its weight-gradient stream the engine's two streams get queues of their own
  // (measured at one rank: 11.6 ms/step without it, 10.3 with).  It also proves the ranks can talk before any training state exists.
  {
    float* probe = nullptr;
    he = hipMalloc((void**)&probe, sizeof(float));
    if (he == hipSuccess) he = hipMemsetAsync(probe, 0, sizeof(float), cs);
    ncclResult_t nr = ncclSuccess;
    if (he == hipSuccess) nr = g_rccl.AllReduce(probe, probe, 1, ncclFloat32, ncclSum, comm, cs);
    if (he == hipSuccess && nr == ncclSuccess) he = hipStreamSynchronize(cs);
    if (probe) hipFree(probe);
    if (he != hipSuccess || nr != ncclSuccess) {
      hipEventDestroy(e_done); hipEventDestroy(e_ready); hipStreamDestroy(cs);
      g_rccl.CommDestroy(comm);
      ys_set_error("ys_dist_init: first all-reduce failed (%s)", he != hipSuccess ? hipGetErrorString(he) : (g_rccl.GetErrorString ? g_rccl.GetErrorString(nr) : "rccl error"));
      return YS_ERR_HIP;
    }
  }
  ctx->dist_comm = comm; ctx->dist_rank = rank; ctx->dist_world = world;
  ctx->dist_stream = cs; ctx->dist_ready = e_ready; ctx->dist_done = e_done;
  return YS_OK;
#endif
}

int ys_dist_destroy(ys_ctx* ctx) {
#ifdef YS_EMU_BUILD
  (void)ctx; return YS_OK;
#else
  if (!ctx || !ctx->dist_comm) return YS_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->dist_stream);
  g_rccl.CommDestroy((ncclComm_t)ctx->dist_comm);
  hipEventDestroy(ctx->dist_ready); hipEventDestroy(ctx->dist_done);
  hipStreamDestroy(ctx->dist_stream);
  ctx->dist_comm = nullptr; ctx->dist_stream = nullptr; ctx->dist_ready = ctx->dist_done = nullptr; ctx->dist_pending = false;
  return YS_OK;
#endif
}

// SUM all-reduce of one backward segment's gradient range (seg < 0: the whole buffer), asynchronous: the communication
// stream waits for everything enqueued on the engine stream so far, the engine stream is NOT blocked (ys_dist_wait does that).
int ys_dist_allreduce_grads(ys_model* m, int seg) {
#ifdef YS_EMU_BUILD
  (void)m; (void)seg; ys_set_error("ys_dist: not available in the interpreter build"); return YS_ERR_UNSUPPORTED;
#else
  YS_REQUIRE(m, "ys_dist_allreduce_grads: null model");
  ys_ctx* ctx = ys_model_ctx(m);
  YS_REQUIRE(ctx->dist_comm, "ys_dist_allreduce_grads: call ys_dist_init first");
  float* g = nullptr; int64_t n = 0, off = 0, cnt = 0;
  YS_TRY(ys_model_grad_buffer(m, &g, &n));
  if (seg < 0) { off = 0; cnt = n; } else YS_TRY(ys_model_segment_grad_range(m, seg, &off, &cnt));
  if (cnt == 0) return YS_OK;
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  YS_CHECK_HIP(hipEventRecord(ctx->dist_ready, ctx->stream));
  YS_CHECK_HIP(hipStreamWaitEvent(ctx->dist_stream, ctx->dist_ready, 0));
  if (seg >= 0) YS_TRY(ys_model_segment_fence(m, seg, (void*)ctx->dist_stream));   // a segment ended asynchronously: its weight-gradient stream as well
  YS_CHECK_NCCL(g_rccl.AllReduce(g + off, g + off, (size_t)cnt, ncclFloat32, ncclSum, (ncclComm_t)ctx->dist_comm, ctx->dist_stream));
  ctx->dist_pending = true;
  return YS_OK;
#endif
}

// the engine stream waits for every all-reduce issued so far (call before ys_optim_adamw_step)
int ys_dist_wait(ys_model* m) {
#ifdef YS_EMU_BUILD
  (void)m; return YS_OK;
#else
  YS_REQUIRE(m, "ys_dist_wait: null model");
  ys_ctx* ctx = ys_model_ctx(m);
  if (!ctx->dist_comm || !ctx->dist_pending) return YS_OK;
  YS_CHECK_HIP(hipEventRecord(ctx->dist_done, ctx->dist_stream));
  YS_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->dist_done, 0));
  ctx->dist_pending = false;
  return YS_OK;
#endif
}

// backward (head -> neck -> late backbone -> stem) with the all-reduce of each finished segment overlapped with the next one
int ys_model_backward_allreduce(ys_model* m) {
  YS_REQUIRE(m, "ys_model_backward_allreduce: null model");
  const int nseg = ys_model_backward_segments(m);
  for (int s = 0; s < nseg; s++) {
    YS_TRY(ys_model_backward_segment_async(m, s));   // the engine stream does not wait for the weight-gradient stream at the boundary
    YS_TRY(ys_dist_allreduce_grads(m, s));
  }
  return ys_dist_wait(m);
}

}  // extern "C"
