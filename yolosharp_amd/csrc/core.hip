// core.hip -- context, error reporting, memory helpers and the NMS entry point of the C ABI.
#include "ys_internal.h"
#include "ys_kernels.h"
#include <cstring>

static thread_local char g_err[1024] = "";

// ---- option table (ys_internal.h)
#include <mutex>
#include <cstdlib>
extern char** environ;
namespace {
// THE registered keys (round 6: an unknown key is an error in ys_set_option and ignored -- with the rest of the unrelated YS_* environment -- when the table is seeded).
// Read at MODEL CREATION (a later change does not re-plan an existing model): BN_ATOMIC, BNRED, HEAD_FUSE, GROUP (head grouping), OVERLAP, STEM_DIRECT.
// Read at every convolution PLAN (i.e. per launch, through the cached YS_OPT_INT sites): the routing gates GEMM_MIN_M, GEMM_MIN_CIN, GEMM_HALO, HALO_MIN_FILL, HALO_MR4,
// HALO_MAX_GRID, WGEMM_MIN_M, WGEMM_KT, F8_MIN_CIN, F8_MIN_TAPS, GROUP (grouped launches), ATTN_MFMA.  DBG / GEMM_DBG exist in the ablation builds only.
// The table is process-wide: a host that flips a gate from one thread while another thread's model is between its plan-only dry run and the launch re-routes that launch;
// set options before creating models, or from the thread that drives them.
const char* const kOptKeys[] = {"ATTN_MFMA", "BNRED", "BNRED_LOG", "BN_ATOMIC", "F8_MIN_CIN", "F8_MIN_TAPS", "GEMM_HALO", "GEMM_MIN_CIN", "GEMM_MIN_M", "GROUP", "HALO_MAX_GRID",
                                "HALO_MIN_FILL", "HALO_MR4", "HEAD_FUSE", "OVERLAP", "STEM_DIRECT", "WGEMM_KT", "WGEMM_MIN_M", "DBG", "GEMM_DBG"};
bool opt_known(const std::string& k) {
  for (const char* n : kOptKeys) if (k == n) return true;
  return false;
}
struct OptTable {
  std::mutex mu;
  std::map<std::string, double> v;
  std::atomic<unsigned> version{1};
  OptTable() {
    // the ONE place the environment is read: YS_<KEY>=<number> of a REGISTERED key seeds the table, at the first option access of the process (= the first plan or
    // model creation).  Values are numbers: YS_GROUP=0 turns grouping OFF (rounds 1-4 treated some switches as presence flags); a non-numeric value is ignored.
    for (char** e = environ; e && *e; e++) {
      if (strncmp(*e, "YS_", 3) != 0) continue;
      const char* eq = strchr(*e, '=');
      if (!eq || eq == *e + 3) continue;
      const std::string key(*e + 3, (size_t)(eq - (*e + 3)));
      if (!opt_known(key)) continue;
      char* end = nullptr;
      const double d = strtod(eq + 1, &end);
      if (end && end != eq + 1 && *end == 0) v[key] = d;
    }
  }
};
OptTable& opt_table() { static OptTable t; return t; }
}  // namespace
double ys_opt_get(const char* key, double def) {
  OptTable& t = opt_table();
  std::lock_guard<std::mutex> g(t.mu);
  auto it = t.v.find(key);
  return it == t.v.end() ? def : it->second;
}
unsigned ys_opt_version() { return opt_table().version.load(std::memory_order_acquire); }
extern "C" __attribute__((visibility("default"))) int ys_set_option(const char* key, double value) {
  if (!key || !*key) { ys_set_error("ys_set_option: empty key"); return YS_ERR_INVALID_ARG; }
  const char* k = strncmp(key, "YS_", 3) == 0 ? key + 3 : key;
  if (!opt_known(k)) { ys_set_error("ys_set_option: unknown key '%s'", k); return YS_ERR_INVALID_ARG; }
  OptTable& t = opt_table();
  { std::lock_guard<std::mutex> g(t.mu); t.v[k] = value; }
  t.version.fetch_add(1, std::memory_order_acq_rel);
  return YS_OK;
}
extern "C" __attribute__((visibility("default"))) int ys_get_option(const char* key, double* value, int* is_set) {
  if (!key || !*key) { ys_set_error("ys_get_option: empty key"); return YS_ERR_INVALID_ARG; }
  OptTable& t = opt_table();
  std::lock_guard<std::mutex> g(t.mu);
  auto it = t.v.find(strncmp(key, "YS_", 3) == 0 ? key + 3 : key);
  if (is_set) *is_set = it != t.v.end() ? 1 : 0;
  if (value) *value = it != t.v.end() ? it->second : 0.0;
  return YS_OK;
}
extern "C" __attribute__((visibility("default"))) int ys_unset_option(const char* key) {
  if (!key || !*key) { ys_set_error("ys_unset_option: empty key"); return YS_ERR_INVALID_ARG; }
  OptTable& t = opt_table();
  { std::lock_guard<std::mutex> g(t.mu); t.v.erase(strncmp(key, "YS_", 3) == 0 ? key + 3 : key); }
  t.version.fetch_add(1, std::memory_order_acq_rel);
  return YS_OK;
}

// compute units of the current device (MI355X: 256), cached per device; the persistent-grid plans size themselves with it
int ys_cu_count() {
  static std::atomic<int> cache[32];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int v = cache[dev & 31].load(std::memory_order_relaxed);
  if (v > 0) return v;
#ifdef YS_EMU_BUILD
  int n = 256;             // the test interpreter plans like the MI355X it stands in for
#else
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
#endif
  cache[dev & 31].store(n, std::memory_order_relaxed);
  return n;
}

void ys_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
struct KprofEntry { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; std::vector<std::string> labels; hipEvent_t open = nullptr; std::string open_label; };
bool g_kprof_on = false;
std::map<std::string, KprofEntry> g_kprof;
}  // namespace

bool ys_kprof_enabled() { return g_kprof_on; }
void ys_kprof_begin(hipStream_t st, const char* name, const char* label) {
  if (!g_kprof_on) return;
  KprofEntry& e = g_kprof[name];
  e.open_label = label ? label : "";
  hipEvent_t a = nullptr;
  if (hipEventCreate(&a) != hipSuccess) return;
  hipEventRecord(a, st);
  e.open = a;
}
void ys_kprof_end(hipStream_t st, const char* name) {
  if (!g_kprof_on) return;
  KprofEntry& e = g_kprof[name];
  if (!e.open) return;
  hipEvent_t b = nullptr;
  if (hipEventCreate(&b) != hipSuccess) return;
  hipEventRecord(b, st);
  e.ev.push_back({e.open, b});
  e.labels.push_back(e.open_label);
  e.open = nullptr;
}

// device scratch of the host-pointer convenience paths: freed on every exit path (an early return on a later allocation or copy
// used to leak the earlier ones)
namespace {
struct ScratchBuf {
  void* p = nullptr;
  ~ScratchBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
  template <class U> U* as() const { return (U*)p; }
};
}  // namespace

extern "C" {

int ys_ctx_kernel_profile(ys_ctx* ctx, int enable) {
  YS_REQUIRE(ctx, "null ctx");
  YS_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (enable) {
    for (auto& kv : g_kprof) for (auto& p : kv.second.ev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    g_kprof.clear();
  }
  g_kprof_on = enable != 0;
  return YS_OK;
}

int ys_ctx_kernel_profile_read(ys_ctx* ctx, const char* name, int32_t* launches, float* total_ms) {
  YS_REQUIRE(ctx && name && launches && total_ms, "ys_ctx_kernel_profile_read: null argument");
  YS_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  *launches = 0; *total_ms = 0.f;
  auto it = g_kprof.find(name);
  if (it == g_kprof.end()) return YS_OK;
  for (auto& p : it->second.ev) {
    float ms = 0.f;
    if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { *total_ms += ms; *launches += 1; }
  }
  return YS_OK;
}

// per-launch dump (class, label, microseconds) of everything recorded since profiling was enabled; triage tool
int ys_ctx_kernel_profile_dump(ys_ctx* ctx, const char* path) {
  YS_REQUIRE(ctx && path, "ys_ctx_kernel_profile_dump: null argument");
  YS_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  FILE* f = fopen(path, "w");
  YS_REQUIRE(f, "ys_ctx_kernel_profile_dump: cannot open %s", path);
  fprintf(f, "class,label,us\n");
  for (auto& kv : g_kprof)
    for (size_t i = 0; i < kv.second.ev.size(); i++) {
      float ms = 0.f;
      if (hipEventSynchronize(kv.second.ev[i].second) == hipSuccess && hipEventElapsedTime(&ms, kv.second.ev[i].first, kv.second.ev[i].second) == hipSuccess)
        fprintf(f, "%s,%s,%.2f\n", kv.first.c_str(), kv.second.labels[i].c_str(), ms * 1000.f);
    }
  fclose(f);
  return YS_OK;
}

const char* ys_last_error(void) { return g_err; }
int ys_version(void) { return 100; }
int ys_is_device_build(void) {
#ifdef YS_EMU_BUILD
  return 0;
#else
  return 1;
#endif
}

static int ctx_create_common(int device, void* stream, bool own, ys_ctx** out) {
  YS_REQUIRE(out != nullptr, "ys_ctx_create: out is null");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    ys_set_error("ys_ctx_create: no HIP device available (hipGetDeviceCount -> %d, n=%d): the engine has no CPU path", (int)e, ndev);
    return YS_ERR_HIP;
  }
  YS_REQUIRE(device >= 0 && device < ndev, "ys_ctx_create: device %d out of range [0,%d)", device, ndev);
  YS_CHECK_HIP(hipSetDevice(device));
  ys_ctx* c = new ys_ctx();
  c->device = device;
  if (own) {
    YS_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  } else {
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
  }
  YS_CHECK_HIP(hipEventCreate(&c->ev0));
  YS_CHECK_HIP(hipEventCreate(&c->ev1));
  *out = c;
  return YS_OK;
}

int ys_ctx_create(int device, ys_ctx** out) { return ctx_create_common(device, nullptr, true, out); }
int ys_ctx_create_on_stream(int device, void* hip_stream, ys_ctx** out) {
  return ctx_create_common(device, hip_stream, false, out);
}

int ys_ctx_destroy(ys_ctx* ctx) {
  if (!ctx) return YS_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  ys_dist_destroy(ctx);
  if (ctx->nms_ws) hipFree(ctx->nms_ws);
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return YS_OK;
}

int ys_ctx_synchronize(ys_ctx* ctx) {
  YS_REQUIRE(ctx, "ys_ctx_synchronize: null ctx");
  YS_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return YS_OK;
}

void* ys_ctx_stream(ys_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int ys_ctx_profile_enable(ys_ctx* ctx, int enable) {
  YS_REQUIRE(ctx, "null ctx");
  ctx->profile = enable != 0;
  return YS_OK;
}

int ys_ctx_last_ms(ys_ctx* ctx, const char* name, float* ms) {
  YS_REQUIRE(ctx && name && ms, "ys_ctx_last_ms: null argument");
  auto it = ctx->last_ms.find(name);
  YS_REQUIRE(it != ctx->last_ms.end(), "ys_ctx_last_ms: no timing recorded for '%s'", name);
  *ms = it->second;
  return YS_OK;
}

int ys_device_malloc(ys_ctx* ctx, size_t bytes, void** dptr) {
  YS_REQUIRE(ctx && dptr, "ys_device_malloc: null argument");
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  YS_CHECK_HIP(hipMalloc(dptr, bytes ? bytes : 1));
  return YS_OK;
}
int ys_device_free(ys_ctx* ctx, void* dptr) {
  YS_REQUIRE(ctx, "null ctx");
  YS_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (dptr) YS_CHECK_HIP(hipFree(dptr));
  return YS_OK;
}
int ys_memcpy_h2d(ys_ctx* ctx, void* dst, const void* src, size_t bytes) {
  YS_REQUIRE(ctx && dst && src, "ys_memcpy_h2d: null argument");
  YS_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  YS_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return YS_OK;
}
int ys_memcpy_d2h(ys_ctx* ctx, void* dst, const void* src, size_t bytes) {
  YS_REQUIRE(ctx && dst && src, "ys_memcpy_d2h: null argument");
  YS_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  YS_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return YS_OK;
}

static int nms_batched_impl(ys_ctx* ctx, float* pred, int on_device, int B, int C, int A, float conf_thres,
                            float iou_thres, int max_det, int nc, int max_nms, int max_wh, float* out_rows,
                            int64_t* out_keep, int32_t* out_count, int rotated) {
  YS_REQUIRE(ctx && pred && out_rows && out_keep && out_count, "ys_nms_batched: null argument");
  // Ops.cs:248-255: ArgumentException for thresholds outside [0,1]
  YS_REQUIRE(conf_thres >= 0.f && conf_thres <= 1.f, "Invalid Confidence threshold %g, valid values are between 0.0 and 1.0", conf_thres);
  YS_REQUIRE(iou_thres >= 0.f && iou_thres <= 1.f, "Invalid IoU %g, valid values are between 0.0 and 1.0", iou_thres);
  YS_REQUIRE(B > 0 && A > 0 && C > 4 && max_det > 0 && max_nms > 0, "ys_nms_batched: bad shape B=%d C=%d A=%d", B, C, A);
  if (nc == 0) nc = C - 4;  // Ops.cs:269
  YS_REQUIRE(nc > 0 && nc <= C - 4, "ys_nms_batched: nc=%d incompatible with C=%d", nc, C);
  YS_REQUIRE(!rotated || C - 4 - nc >= 1, "ys_nms_rotated_batched: oriented boxes carry their angle as the last channel (C=%d, nc=%d)", C, nc);
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  const int extra = C - 4 - nc;
  const size_t n_pred = (size_t)B * C * A, n_rows = (size_t)B * max_det * (6 + extra), n_keep = (size_t)B * max_det;
  YsTimer timer(ctx, "nms");
  if (on_device) return ys_nms_launch(ctx, pred, B, C, A, conf_thres, iou_thres, max_det, nc, max_nms, max_wh, out_rows, out_keep, out_count, rotated);
  ScratchBuf b_pred, b_rows, b_keep, b_cnt;
  YS_CHECK_HIP(b_pred.alloc(n_pred * 4));
  YS_CHECK_HIP(b_rows.alloc(n_rows * 4));
  YS_CHECK_HIP(b_keep.alloc(n_keep * 8));
  YS_CHECK_HIP(b_cnt.alloc((size_t)B * 4));
  float* d_pred = b_pred.as<float>(); float* d_rows = b_rows.as<float>(); int64_t* d_keep = b_keep.as<int64_t>(); int32_t* d_cnt = b_cnt.as<int32_t>();
  YS_CHECK_HIP(hipMemcpyAsync(d_pred, pred, n_pred * 4, hipMemcpyHostToDevice, ctx->stream));
  int st = ys_nms_launch(ctx, d_pred, B, C, A, conf_thres, iou_thres, max_det, nc, max_nms, max_wh, d_rows, d_keep, d_cnt, rotated);
  if (st == YS_OK) {
    hipMemcpyAsync(pred, d_pred, n_pred * 4, hipMemcpyDeviceToHost, ctx->stream);  // in-place xyxy visible to the caller
    hipMemcpyAsync(out_rows, d_rows, n_rows * 4, hipMemcpyDeviceToHost, ctx->stream);
    hipMemcpyAsync(out_keep, d_keep, n_keep * 8, hipMemcpyDeviceToHost, ctx->stream);
    hipMemcpyAsync(out_count, d_cnt, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream);
  }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (st != YS_OK) return st;
  if (e != hipSuccess) { ys_set_error("ys_nms_batched: %s", hipGetErrorString(e)); return YS_ERR_HIP; }
  return YS_OK;
}

int ys_nms_batched(ys_ctx* ctx, float* pred, int on_device, int B, int C, int A, float conf_thres,
                   float iou_thres, int max_det, int nc, int max_nms, int max_wh, float* out_rows,
                   int64_t* out_keep, int32_t* out_count) {
  return nms_batched_impl(ctx, pred, on_device, B, C, A, conf_thres, iou_thres, max_det, nc, max_nms, max_wh, out_rows, out_keep, out_count, 0);
}

// Ops.non_max_suppression(rotated: true) (Ops.cs:286,349-353) = Ops.nms_rotated (:373-401) on Metrics.batch_probiou
int ys_nms_rotated_batched(ys_ctx* ctx, float* pred, int on_device, int B, int C, int A, float conf_thres,
                           float iou_thres, int max_det, int nc, int max_nms, int max_wh, float* out_rows,
                           int64_t* out_keep, int32_t* out_count) {
  return nms_batched_impl(ctx, pred, on_device, B, C, A, conf_thres, iou_thres, max_det, nc, max_nms, max_wh, out_rows, out_keep, out_count, 1);
}

// Metrics.probiou (pairwise [n] with optional CIoU term, Metrics.cs:137-177) / Metrics.batch_probiou ([n, m], :223-258) on xywhr boxes
static int probiou_impl(ys_ctx* ctx, const float* obb1, const float* obb2, int on_device, long n, long m, int pairwise, int ciou, float eps, float* out) {
  YS_REQUIRE(ctx && obb1 && obb2 && out, "ys_probiou: null argument");
  YS_REQUIRE(n >= 0 && m >= 0, "ys_probiou: bad shape");
  const long total = pairwise ? n : n * m;
  if (total == 0) return YS_OK;
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  if (on_device) return ys_probiou_launch(ctx->stream, obb1, obb2, n, m, pairwise, ciou, eps, out);
  ScratchBuf b1, b2, bo;
  YS_CHECK_HIP(b1.alloc((size_t)n * 20)); YS_CHECK_HIP(b2.alloc((size_t)m * 20)); YS_CHECK_HIP(bo.alloc((size_t)total * 4));
  YS_CHECK_HIP(hipMemcpyAsync(b1.p, obb1, (size_t)n * 20, hipMemcpyHostToDevice, ctx->stream));
  YS_CHECK_HIP(hipMemcpyAsync(b2.p, obb2, (size_t)m * 20, hipMemcpyHostToDevice, ctx->stream));
  YS_TRY(ys_probiou_launch(ctx->stream, b1.as<float>(), b2.as<float>(), n, m, pairwise, ciou, eps, bo.as<float>()));
  YS_CHECK_HIP(hipMemcpyAsync(out, bo.p, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
  YS_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return YS_OK;
}
int ys_probiou(ys_ctx* ctx, const float* obb1, const float* obb2, int on_device, int n, int ciou, float eps, float* out) {
  return probiou_impl(ctx, obb1, obb2, on_device, n, n, 1, ciou, eps, out);
}
int ys_batch_probiou(ys_ctx* ctx, const float* obb1, int n, const float* obb2, int m, int on_device, float eps, float* out) {
  return probiou_impl(ctx, obb1, obb2, on_device, n, m, 0, 0, eps, out);
}

// Augment.LetterBox.LetterboxImage (Data/Augment.cs:757-778) and Augment.Rectangle.RectangleImage (:836-857) on the device:
// ratio = min(fit_w / w, fit_h / h) in fp32, new = (int)(size * ratio), centred on an out_w x out_h canvas filled with `color`
// (LetterBox: fit = out; Rectangle: fit = the label's resized shape, out = its rectangle shape).  src / dst: planes [C][h][w] ->
// [C][out_h][out_w], uint8 (images) or fp32 (masks, is_float = 1).  pad_l / pad_u are returned for the box / keypoint offsets.
int ys_letterbox(ys_ctx* ctx, const void* src, int is_float, int on_device, int C, int h, int w, int fit_w, int fit_h, int out_w, int out_h,
                 int color, void* dst, int32_t* pad_l, int32_t* pad_u) {
  YS_REQUIRE(ctx && src && dst, "ys_letterbox: null argument");
  YS_REQUIRE(C > 0 && h > 0 && w > 0 && fit_w > 0 && fit_h > 0 && out_w > 0 && out_h > 0, "ys_letterbox: bad geometry");
  const float ratio_w = (float)fit_w / (float)w, ratio_h = (float)fit_h / (float)h;
  const float ratio = ratio_w < ratio_h ? ratio_w : ratio_h;
  const int new_w = (int)((float)w * ratio), new_h = (int)((float)h * ratio);
  YS_REQUIRE(new_w > 0 && new_h > 0 && new_w <= out_w && new_h <= out_h, "ys_letterbox: resized image %dx%d does not fit the %dx%d canvas", new_w, new_h, out_w, out_h);
  const int pl = (out_w - new_w) / 2, pu = (out_h - new_h) / 2;
  if (pad_l) *pad_l = pl;
  if (pad_u) *pad_u = pu;
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t es = is_float ? 4 : 1;
  const size_t nin = (size_t)C * h * w * es, nout = (size_t)C * out_h * out_w * es;
  if (on_device) return ys_letterbox_launch(st, is_float, src, C, h, w, dst, out_h, out_w, new_h, new_w, pu, pl, (float)color);
  void *d_in = nullptr, *d_out = nullptr;
  int rc = YS_OK;
  if (hipMalloc(&d_in, nin) != hipSuccess || hipMalloc(&d_out, nout) != hipSuccess) { ys_set_error("ys_letterbox: out of device memory"); rc = YS_ERR_OOM; }
  if (rc == YS_OK && hipMemcpyAsync(d_in, src, nin, hipMemcpyHostToDevice, st) != hipSuccess) rc = YS_ERR_HIP;
  if (rc == YS_OK) rc = ys_letterbox_launch(st, is_float, d_in, C, h, w, d_out, out_h, out_w, new_h, new_w, pu, pl, (float)color);
  if (rc == YS_OK && hipMemcpyAsync(dst, d_out, nout, hipMemcpyDeviceToHost, st) != hipSuccess) rc = YS_ERR_HIP;
  if (hipStreamSynchronize(st) != hipSuccess && rc == YS_OK) rc = YS_ERR_HIP;
  if (d_in) hipFree(d_in);
  if (d_out) hipFree(d_out);
  if (rc == YS_ERR_HIP) ys_set_error("ys_letterbox: HIP error");
  return rc;
}

// Ops.process_mask (Utils/Ops.cs:462-489): masks_in[n][nm] @ protos[nm][mh][mw], crop to the boxes, optional bilinear upsample
// to (ih, iw), threshold > 0.  out: uint8 [n][oh][ow] with (oh, ow) = upsample ? (ih, iw) : (mh, mw).
int ys_process_mask(ys_ctx* ctx, const float* protos, const float* masks_in, const float* boxes, int on_device, int n, int nm,
                    int mh, int mw, int ih, int iw, int upsample, int crop_mode, uint8_t* out) {
  YS_REQUIRE(ctx && protos && out, "ys_process_mask: null argument");
  YS_REQUIRE(n >= 0 && nm > 0 && mh > 0 && mw > 0 && ih > 0 && iw > 0, "ys_process_mask: bad shape n=%d nm=%d mask %dx%d image %dx%d", n, nm, mh, mw, ih, iw);
  if (n == 0) return YS_OK;
  YS_REQUIRE(masks_in && boxes, "ys_process_mask: null argument");
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  YsTimer timer(ctx, "process_mask");
  if (on_device) return ys_process_mask_launch(ctx->stream, protos, masks_in, boxes, n, nm, mh, mw, ih, iw, upsample, crop_mode, out);
  const size_t np = (size_t)nm * mh * mw, no = (size_t)n * (upsample ? (size_t)ih * iw : (size_t)mh * mw);
  ScratchBuf b_p, b_m, b_b, b_o;
  YS_CHECK_HIP(b_p.alloc(np * 4));
  YS_CHECK_HIP(b_m.alloc((size_t)n * nm * 4));
  YS_CHECK_HIP(b_b.alloc((size_t)n * 16));
  YS_CHECK_HIP(b_o.alloc(no));
  float *d_p = b_p.as<float>(), *d_m = b_m.as<float>(), *d_b = b_b.as<float>(); unsigned char* d_o = b_o.as<unsigned char>();
  hipMemcpyAsync(d_p, protos, np * 4, hipMemcpyHostToDevice, ctx->stream);
  hipMemcpyAsync(d_m, masks_in, (size_t)n * nm * 4, hipMemcpyHostToDevice, ctx->stream);
  hipMemcpyAsync(d_b, boxes, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream);
  int st = ys_process_mask_launch(ctx->stream, d_p, d_m, d_b, n, nm, mh, mw, ih, iw, upsample, crop_mode, d_o);
  if (st == YS_OK) hipMemcpyAsync(out, d_o, no, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (st != YS_OK) return st;
  if (e != hipSuccess) { ys_set_error("ys_process_mask: %s", hipGetErrorString(e)); return YS_ERR_HIP; }
  return YS_OK;
}

}  // extern "C"
