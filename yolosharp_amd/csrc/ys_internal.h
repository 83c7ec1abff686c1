// ys_internal.h -- host-side internals shared by the translation units of libyolosharp_hip.
#pragma once
#include "ys_hip.h"
#include "../../include/yolosharp_hip.h"
#include <string>
#include <vector>
#include <map>
#include <cstdio>
#include <cstdarg>

void ys_set_error(const char* fmt, ...);

// ---- tuning / routing options (round 5).  ONE process-wide table replaces the ~45 getenv("YS_*") sites of rounds 1-4: ys_set_option (C ABI) writes it, the
// kernel plans read it through YS_OPT_INT / YS_OPT_F (a cached value per call site, refreshed when the table's version changes -- so a test or a host can flip a
// gate between two launches of one process, which a `static const ... = getenv(...)` could not).  The environment is read in exactly one place: when the library
// is loaded, every YS_<KEY>=<number> variable seeds the table (the A/B scripts under tools/ keep working); later changes of the environment are not seen.
double ys_opt_get(const char* key, double def);
unsigned ys_opt_version();
int ys_cu_count();       // compute units of the current device (core.hip)
#include <atomic>
#define YS_OPT_F(key, def) ([]() -> double { static std::atomic<unsigned> ver_{0xffffffffu}; static std::atomic<double> val_{0.0}; \
    const unsigned g_ = ys_opt_version(); if (ver_.load(std::memory_order_acquire) != g_) { val_.store(ys_opt_get(key, (double)(def)), std::memory_order_relaxed); ver_.store(g_, std::memory_order_release); } \
    return val_.load(std::memory_order_relaxed); }())
#define YS_OPT_INT(key, def) ((long)YS_OPT_F(key, def))

#define YS_CHECK_HIP(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      ys_set_error("HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
      return YS_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

#define YS_REQUIRE(cond, ...)                \
  do {                                       \
    if (!(cond)) {                           \
      ys_set_error(__VA_ARGS__);             \
      return YS_ERR_INVALID_ARG;             \
    }                                        \
  } while (0)

#define YS_TRY(expr)                 \
  do {                               \
    int _s = (expr);                 \
    if (_s != YS_OK) return _s;      \
  } while (0)

struct ys_ctx {
  int device = 0;
  hipStream_t stream = 0;
  bool own_stream = true;
  bool profile = false;
  // NMS workspace (grown on demand)
  void* nms_ws = nullptr;
  size_t nms_ws_bytes = 0;
  bool nms_count_clean = false;   // the per-image candidate counters are zero (nms_sort_kernel clears what it consumed)
  int nms_count_clean_B = 0;      // ... for this many images (a later call with more images clears again)
  // scratch for per-operator entry points
  std::map<std::string, float> last_ms;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // data-parallel state (dist.hip): RCCL communicator, its stream, hand-over events
  void* dist_comm = nullptr; int dist_rank = 0, dist_world = 1;
  hipStream_t dist_stream = nullptr; hipEvent_t dist_ready = nullptr, dist_done = nullptr; bool dist_pending = false;
};

// simple RAII-free timing helper: records events on the ctx stream when profiling is on
struct YsTimer {
  ys_ctx* c; const char* name; bool on;
  YsTimer(ys_ctx* ctx, const char* n) : c(ctx), name(n), on(ctx->profile) {
    if (on) hipEventRecord(c->ev0, c->stream);
  }
  ~YsTimer() {
    if (on) {
      hipEventRecord(c->ev1, c->stream);
      hipEventSynchronize(c->ev1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, c->ev0, c->ev1);
      c->last_ms[name] = ms;
    }
  }
};

// ---- per-kernel-class timing with HIP events on the launch stream (off by default; bench.py enables it
//      for a few untimed steps to measure the dominant kernel's average launch duration)
bool ys_kprof_enabled();
void ys_kprof_begin(hipStream_t st, const char* name, const char* label = nullptr);
void ys_kprof_end(hipStream_t st, const char* name);
struct YsKprofScope {
  hipStream_t st; const char* name;
  YsKprofScope(hipStream_t s, const char* n, const char* label = nullptr) : st(s), name(n) { ys_kprof_begin(st, name, label); }
  ~YsKprofScope() { ys_kprof_end(st, name); }
};

// ---- kernels' host launchers (defined in the .hip files) ----
int ys_nms_launch(ys_ctx* ctx, float* pred_dev, int B, int C, int A, float conf, float iou, int max_det,
                  int nc, int max_nms, int max_wh, float* out_rows, int64_t* out_keep, int32_t* out_count, int rotated = 0);
int ys_probiou_launch(hipStream_t st, const float* o1, const float* o2, long n, long m, int pairwise, int ciou, float eps, float* out);
