// valmetrics.hip -- the per-image part of Detector.Val (Models/Detector.cs:103-120) on the device:
//   GT boxes = bboxes[batch_idx == b] * (W, H, W, H) -> xyxy          (Ops.xywh2xyxy, Utils/Ops.cs:68-81)
//   iou      = Metrics.box_iou(gt, pred[:, 0:4])                       (Utils/Metrics.cs:16-34, eps 1e-7)
//   correct  = match_predictions(pred[:, 5], cls, iou)                 (Models/YoloBaseTaskModel.cs:377-446)
// The reference walks images on the host and de-duplicates matches with per-element .item() loops
// (GetUniqueByColumn, :422-444); here one workgroup owns an image and nothing leaves the device.
//
// match_predictions, restated.  For a threshold t the reference takes all (label, detection) pairs with
// iou * (class match) >= t, orders them by IoU descending, keeps the FIRST pair of every detection (result ordered by
// detection index, since torch.unique sorts), then the first pair of every label in THAT order.  Hence:
//   best(d)  = the class-matching label with the largest IoU for detection d (independent of t),
//   label l is credited to the smallest d with best(d) == l and iou(best(d), d) >= t,   correct[d][t] = 1 for those d.
// Equal IoUs are ordered by the sort implementation upstream (unpinned); here the lower label index wins.
// Compiled with -ffp-contract=off: the IoU arithmetic is the reference's operation order in plain fp32.
#include "ys_internal.h"
#include "ys_kernels.h"

#define VM_THREADS 256
#define VM_NT 10          // IoU thresholds linspace(0.5, 0.95, 10)

struct VmThr { float t[VM_NT]; };

__device__ inline float vm_iou(float a1x, float a1y, float a2x, float a2y, float b1x, float b1y, float b2x, float b2y, float eps) {
  // inter = (min(a2, b2) - max(a1, b1)).clamp(0).prod ; iou = inter / (area1 + area2 - inter + eps)
  float iw = fminf(a2x, b2x) - fmaxf(a1x, b1x);
  float ih = fminf(a2y, b2y) - fmaxf(a1y, b1y);
  iw = iw < 0.f ? 0.f : iw;
  ih = ih < 0.f ? 0.f : ih;
  const float inter = iw * ih;
  const float area1 = (a2x - a1x) * (a2y - a1y), area2 = (b2x - b1x) * (b2y - b1y);
  return inter / (area1 + area2 - inter + eps);
}

__global__ void __launch_bounds__(VM_THREADS)
box_iou_kernel(const float* __restrict__ b1, int n, const float* __restrict__ b2, int m, float eps, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * m) return;
  const int r = (int)(i / m), c = (int)(i - (long)r * m);
  out[i] = vm_iou(b1[4 * r], b1[4 * r + 1], b1[4 * r + 2], b1[4 * r + 3], b2[4 * c], b2[4 * c + 1], b2[4 * c + 2], b2[4 * c + 3], eps);
}

// Metrics.kpt_iou (Metrics.cs:186-212): OKS of ground-truth keypoints kpt1 [n][K][3] (x, y, visibility) against predictions
// kpt2 [m][K][D]; area [n]; sigma = OKS sigmas when K == 17, else 1/K.  One thread per (gt, prediction) pair.
struct KptSigma { float s[64]; };
__global__ void __launch_bounds__(VM_THREADS)
kpt_iou_kernel(const float* __restrict__ k1, int n, const float* __restrict__ k2, int m, const float* __restrict__ area, int K, int D,
               KptSigma sg, float eps, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * m) return;
  const int r = (int)(i / m), c = (int)(i - (long)r * m);
  const float* g = k1 + (long)r * K * 3;
  const float* p = k2 + (long)c * K * D;
  const float ar = area[r] + eps;
  float acc = 0.f, cnt = 0.f;
  for (int k = 0; k < K; k++) {
    const float dx = g[3 * k] - p[D * k], dy = g[3 * k + 1] - p[D * k + 1];
    const float s2 = 2.0f * sg.s[k];
    const float e = (dx * dx + dy * dy) / (s2 * s2 * ar * 2.0f);
    const float mk = g[3 * k + 2] != 0.f ? 1.f : 0.f;
    acc += expf(-e) * mk;
    cnt += mk;
  }
  out[i] = acc / (cnt + eps);
}

// one workgroup per image
__global__ void __launch_bounds__(VM_THREADS)
val_match_kernel(const float* __restrict__ rows, const int* __restrict__ count, int max_det, int row_stride,
                 const float* __restrict__ batch_idx, const float* __restrict__ cls, const float* __restrict__ bboxes, int n_labels,
                 float img_w, float img_h, VmThr thr, int lcap, int* __restrict__ ws_lab /*[B][lcap]*/,
                 float* __restrict__ ws_best /*[B][max_det][2]*/, unsigned char* __restrict__ correct /*[B][max_det][10]*/, int* __restrict__ overflow) {
  __shared__ int s_nl;
  const int b = blockIdx.x, tid = threadIdx.x;
  int* lab = ws_lab + (long)b * lcap;
  float* best = ws_best + (long)b * max_det * 2;
  unsigned char* cor = correct + (long)b * max_det * VM_NT;
  const int D = count[b] < max_det ? count[b] : max_det;
  for (int i = tid; i < max_det * VM_NT; i += VM_THREADS) cor[i] = 0;
  // labels of this image, in collate order (boolean-mask indexing keeps the order, Detector.cs:110-112)
  if (tid == 0) {
    int k = 0;
    for (int j = 0; j < n_labels; j++)
      if ((int)batch_idx[j] == b) { if (k < lcap) lab[k] = j; k++; }
    if (k > lcap) { atomicMax(overflow, k); k = lcap; }
    s_nl = k;
  }
  __syncthreads();
  const int L = s_nl;
  // best class-matching label per detection
  for (int d = tid; d < D; d += VM_THREADS) {
    const float* pr = rows + ((long)b * max_det + d) * row_stride;
    const float px1 = pr[0], py1 = pr[1], px2 = pr[2], py2 = pr[3], pc = pr[5];
    float bi = -1.f; int bl = -1;
    for (int k = 0; k < L; k++) {
      const int j = lab[k];
      if (cls[j] != pc) continue;
      const float cx = bboxes[4 * j] * img_w, cy = bboxes[4 * j + 1] * img_h, w = bboxes[4 * j + 2] * img_w, h = bboxes[4 * j + 3] * img_h;
      const float iou = vm_iou(cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, px1, py1, px2, py2, 1e-7f);
      if (iou > bi) { bi = iou; bl = k; }
    }
    best[2 * d] = bi;
    best[2 * d + 1] = (float)bl;
  }
  __syncthreads();
  // every (label, threshold): the first detection that chose this label and clears the threshold
  for (int e = tid; e < L * VM_NT; e += VM_THREADS) {
    const int k = e / VM_NT, ti = e - k * VM_NT;
    const float t = thr.t[ti];
    for (int d = 0; d < D; d++)
      if ((int)best[2 * d + 1] == k && best[2 * d] >= t) { cor[d * VM_NT + ti] = 1; break; }
  }
}

// Metrics.mask_iou (Utils/Metrics.cs:120-125) as Segmenter.Val calls it (Models/Segmenter.cs:131-143): mask1[k] = (gt_ids == k + 1)
// from the overlap-encoded id map, mask2[j] = the 0/1 masks of Ops.process_mask.  The reference's float matmul of 0/1 values is
// an exact integer count (npix < 2^24), so  iou = inter / ((area1 + area2 - inter) + eps)  is reproduced bit for bit.
// One workgroup per predicted mask; LDS histograms over the label ids (dynamic LDS: 2 * nl ints).
__global__ void __launch_bounds__(VM_THREADS)
mask_iou_kernel(const float* __restrict__ gt_ids, int nl, const unsigned char* __restrict__ pm, int n, int npix, float eps,
                float* __restrict__ iou /*[nl][n]*/) {
  YS_DYN_LDS(lds);
  int* s_hist = (int*)lds;                 // [0, nl): |mask1[k]|, [nl, 2 nl): |mask1[k] & mask2[j]|
  __shared__ int s_area2;
  const int j = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < 2 * nl; i += VM_THREADS) s_hist[i] = 0;
  if (tid == 0) s_area2 = 0;
  __syncthreads();
  const unsigned char* m2 = pm + (long)j * npix;
  int a2 = 0;
  for (int p = tid; p < npix; p += VM_THREADS) {
    const float idf = gt_ids[p];
    const int id = (int)idf;
    const bool lab = id >= 1 && id <= nl && (float)id == idf;       // (batch_mask == index) is an exact float comparison
    const bool on = m2[p] != 0;
    if (on) a2++;
    if (lab) {
      atomicAdd(&s_hist[id - 1], 1);
      if (on) atomicAdd(&s_hist[nl + id - 1], 1);
    }
  }
  atomicAdd(&s_area2, a2);
  __syncthreads();
  for (int k = tid; k < nl; k += VM_THREADS) {
    const float inter = (float)s_hist[nl + k];
    const float uni = ((float)s_hist[k] + (float)s_area2) - inter;
    iou[(long)k * n + j] = inter / (uni + eps);
  }
}

// match_predictions (Models/YoloBaseTaskModel.cs:377-446) on a caller-supplied IoU matrix [L][D]; same restatement as
// val_match_kernel above (one workgroup).
__global__ void __launch_bounds__(VM_THREADS)
match_iou_kernel(const float* __restrict__ pred_cls, int D, const float* __restrict__ true_cls, int L, const float* __restrict__ iou,
                 VmThr thr, float* __restrict__ best /*[D][2]*/, unsigned char* __restrict__ cor /*[D][10]*/) {
  const int tid = threadIdx.x;
  for (int i = tid; i < D * VM_NT; i += VM_THREADS) cor[i] = 0;
  for (int d = tid; d < D; d += VM_THREADS) {
    const float pc = pred_cls[d];
    float bi = -1.f; int bl = -1;
    for (int k = 0; k < L; k++) {
      if (true_cls[k] != pc) continue;
      const float v = iou[(long)k * D + d];
      if (v > bi) { bi = v; bl = k; }
    }
    best[2 * d] = bi;
    best[2 * d + 1] = (float)bl;
  }
  __syncthreads();
  for (int e = tid; e < L * VM_NT; e += VM_THREADS) {
    const int k = e / VM_NT, ti = e - k * VM_NT;
    const float t = thr.t[ti];
    for (int d = 0; d < D; d++)
      if ((int)best[2 * d + 1] == k && best[2 * d] >= t) { cor[d * VM_NT + ti] = 1; break; }
  }
}

// torch.linspace(0.5, 0.95, 10) in fp32 (ATen: step = (end-start)/(steps-1); first half start + step*i, second half
// end - step*(steps-1-i))
static VmThr vm_thresholds() {
  VmThr t;
  const float start = 0.5f, end = 0.95f;
  const float step = (end - start) / (float)(VM_NT - 1);
  for (int i = 0; i < VM_NT; i++) t.t[i] = i < VM_NT / 2 ? start + step * (float)i : end - step * (float)(VM_NT - 1 - i);
  return t;
}

extern "C" {

int ys_box_iou(ys_ctx* ctx, const float* box1, int n, const float* box2, int m, float eps, int on_device, float* iou) {
  YS_REQUIRE(ctx && iou && n >= 0 && m >= 0, "ys_box_iou: bad argument");
  if ((long)n * m == 0) return YS_OK;
  YS_REQUIRE(box1 && box2, "ys_box_iou: null boxes");
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (on_device) {
    YS_LAUNCH(box_iou_kernel, ys_cdiv((long)n * m, VM_THREADS), VM_THREADS, st, box1, n, box2, m, eps, iou);
    return YS_OK;
  }
  float *d1 = nullptr, *d2 = nullptr, *d3 = nullptr;
  YS_CHECK_HIP(hipMalloc(&d1, (size_t)n * 16));
  YS_CHECK_HIP(hipMalloc(&d2, (size_t)m * 16));
  YS_CHECK_HIP(hipMalloc(&d3, (size_t)n * m * 4));
  hipMemcpyAsync(d1, box1, (size_t)n * 16, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d2, box2, (size_t)m * 16, hipMemcpyHostToDevice, st);
  YS_LAUNCH(box_iou_kernel, ys_cdiv((long)n * m, VM_THREADS), VM_THREADS, st, (const float*)d1, n, (const float*)d2, m, eps, d3);
  hipError_t e = hipMemcpyAsync(iou, d3, (size_t)n * m * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(d1); hipFree(d2); hipFree(d3);
  if (e != hipSuccess) { ys_set_error("ys_box_iou: %s", hipGetErrorString(e)); return YS_ERR_HIP; }
  return YS_OK;
}

// scratch + staging helper for the small validation calls: device pointers are used as they are, host arrays are staged
namespace {
struct VmStage {
  hipStream_t st; bool on_device; std::vector<void*> tmp; bool ok = true;
  VmStage(hipStream_t s, int od) : st(s), on_device(od != 0) {}
  ~VmStage() { for (void* p : tmp) hipFree(p); }
  void* alloc(size_t bytes) { void* p = nullptr; if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) { ok = false; return nullptr; } tmp.push_back(p); return p; }
  const void* in(const void* host, size_t bytes) {
    if (on_device || !host) return host;
    void* d = alloc(bytes);
    if (d && bytes && hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, st) != hipSuccess) ok = false;
    return d;
  }
  void* out(void* host, size_t bytes) { return on_device ? host : alloc(bytes); }
};
}  // namespace

int ys_mask_iou(ys_ctx* ctx, const float* gt_ids, int nl, const uint8_t* pred_masks, int n, int npix, float eps, int on_device, float* iou) {
  YS_REQUIRE(ctx && iou && nl >= 0 && n >= 0 && npix > 0, "ys_mask_iou: bad argument");
  if ((long)nl * n == 0) return YS_OK;
  YS_REQUIRE(gt_ids && pred_masks, "ys_mask_iou: null masks");
  YS_REQUIRE(npix < (1 << 24), "ys_mask_iou: %d pixels exceed the exact-integer range of the fp32 reference", npix);
  YS_REQUIRE(nl <= 8192, "ys_mask_iou: %d labels exceed the LDS histogram capacity (8192)", nl);
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  VmStage sg(st, on_device);
  const float* d_ids = (const float*)sg.in(gt_ids, (size_t)npix * 4);
  const unsigned char* d_pm = (const unsigned char*)sg.in(pred_masks, (size_t)n * npix);
  float* d_iou = (float*)sg.out(iou, (size_t)nl * n * 4);
  if (!sg.ok) { ys_set_error("ys_mask_iou: out of device memory"); return YS_ERR_OOM; }
  YS_LAUNCH_LDS(mask_iou_kernel, n, VM_THREADS, (size_t)2 * nl * sizeof(int) + 16, st, d_ids, nl, d_pm, n, npix, eps, d_iou);
  YS_CHECK_HIP(hipGetLastError());
  if (!on_device) {
    YS_CHECK_HIP(hipMemcpyAsync(iou, d_iou, (size_t)nl * n * 4, hipMemcpyDeviceToHost, st));
    YS_CHECK_HIP(hipStreamSynchronize(st));
  }
  return YS_OK;
}

int ys_kpt_iou(ys_ctx* ctx, const float* kpt1, int n, const float* kpt2, int m, const float* area, int kpt_num, int kpt_dim, float eps,
               int on_device, float* iou) {
  YS_REQUIRE(ctx && iou && n >= 0 && m >= 0, "ys_kpt_iou: bad argument");
  YS_REQUIRE(kpt_num > 0 && kpt_num <= 64 && (kpt_dim == 2 || kpt_dim == 3), "ys_kpt_iou: %d keypoints of dim %d", kpt_num, kpt_dim);
  if ((long)n * m == 0) return YS_OK;
  YS_REQUIRE(kpt1 && kpt2 && area, "ys_kpt_iou: null argument");
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  VmStage sg(st, on_device);
  const float* d1 = (const float*)sg.in(kpt1, (size_t)n * kpt_num * 3 * 4);
  const float* d2 = (const float*)sg.in(kpt2, (size_t)m * kpt_num * kpt_dim * 4);
  const float* da = (const float*)sg.in(area, (size_t)n * 4);
  float* d_iou = (float*)sg.out(iou, (size_t)n * m * 4);
  if (!sg.ok) { ys_set_error("ys_kpt_iou: out of device memory"); return YS_ERR_OOM; }
  static const float oks[17] = {0.026f, 0.025f, 0.025f, 0.035f, 0.035f, 0.079f, 0.079f, 0.072f, 0.072f, 0.062f, 0.062f, 0.107f, 0.107f,
                                0.087f, 0.087f, 0.089f, 0.089f};                       // PoseDetector.cs:12-19
  KptSigma ks{};
  for (int k = 0; k < kpt_num; k++) ks.s[k] = kpt_num == 17 ? oks[k] : 1.0f / (float)kpt_num;   // Metrics.cs:205
  YS_LAUNCH(kpt_iou_kernel, ys_cdiv((long)n * m, VM_THREADS), VM_THREADS, st, d1, n, d2, m, da, kpt_num, kpt_dim, ks, eps, d_iou);
  YS_CHECK_HIP(hipGetLastError());
  if (!on_device) {
    YS_CHECK_HIP(hipMemcpyAsync(iou, d_iou, (size_t)n * m * 4, hipMemcpyDeviceToHost, st));
    YS_CHECK_HIP(hipStreamSynchronize(st));
  }
  return YS_OK;
}

int ys_match_predictions(ys_ctx* ctx, const float* pred_cls, int n, const float* true_cls, int nl, const float* iou, int on_device,
                         uint8_t* correct) {
  YS_REQUIRE(ctx && n >= 0 && nl >= 0, "ys_match_predictions: bad argument");
  if (n == 0) return YS_OK;
  YS_REQUIRE(pred_cls && correct && (nl == 0 || (true_cls && iou)), "ys_match_predictions: null argument");
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  VmStage sg(st, on_device);
  const float* d_pc = (const float*)sg.in(pred_cls, (size_t)n * 4);
  const float* d_tc = (const float*)sg.in(true_cls, (size_t)nl * 4);
  const float* d_iou = (const float*)sg.in(iou, (size_t)nl * n * 4);
  unsigned char* d_cor = (unsigned char*)sg.out(correct, (size_t)n * VM_NT);
  float* d_best = (float*)sg.alloc((size_t)n * 8);
  if (!sg.ok) { ys_set_error("ys_match_predictions: out of device memory"); return YS_ERR_OOM; }
  YS_LAUNCH(match_iou_kernel, 1, VM_THREADS, st, d_pc, n, d_tc, nl, d_iou, vm_thresholds(), d_best, d_cor);
  if (!on_device) YS_CHECK_HIP(hipMemcpyAsync(correct, d_cor, (size_t)n * VM_NT, hipMemcpyDeviceToHost, st));
  YS_CHECK_HIP(hipStreamSynchronize(st));      // the scratch buffers are released on return
  return YS_OK;
}

int ys_val_match_batched(ys_ctx* ctx, const float* rows, const int32_t* count, int on_device, int batch, int max_det, int row_stride,
                         const float* batch_idx, const float* cls, const float* bboxes, int n_labels, float img_w, float img_h,
                         uint8_t* correct) {
  YS_REQUIRE(ctx && rows && count && correct, "ys_val_match_batched: null argument");
  YS_REQUIRE(batch > 0 && max_det > 0 && row_stride >= 6 && n_labels >= 0, "ys_val_match_batched: bad shape");
  YS_REQUIRE(n_labels == 0 || (batch_idx && cls && bboxes), "ys_val_match_batched: null label arrays");
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int lcap = n_labels > 0 ? n_labels : 1;          // an image can hold all labels of the batch
  const size_t nrow = (size_t)batch * max_det * row_stride, ncor = (size_t)batch * max_det * VM_NT;
  const float *d_rows = rows, *d_bi = batch_idx, *d_cl = cls, *d_bb = bboxes;
  const int* d_cnt = count;
  unsigned char* d_cor = correct;
  std::vector<void*> tmp;
  auto dalloc = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return nullptr; tmp.push_back(p); return p; };
  int* d_lab = (int*)dalloc((size_t)batch * lcap * 4);
  float* d_best = (float*)dalloc((size_t)batch * max_det * 8);
  int* d_ovf = (int*)dalloc(4);
  bool ok = d_lab && d_best && d_ovf;
  if (ok && !on_device) {
    float* a = (float*)dalloc(nrow * 4); int* c = (int*)dalloc((size_t)batch * 4);
    float* b1 = (float*)dalloc((size_t)lcap * 4); float* b2 = (float*)dalloc((size_t)lcap * 4); float* b3 = (float*)dalloc((size_t)lcap * 16);
    unsigned char* co = (unsigned char*)dalloc(ncor);
    ok = a && c && b1 && b2 && b3 && co;
    if (ok) {
      hipMemcpyAsync(a, rows, nrow * 4, hipMemcpyHostToDevice, st);
      hipMemcpyAsync(c, count, (size_t)batch * 4, hipMemcpyHostToDevice, st);
      if (n_labels > 0) {
        hipMemcpyAsync(b1, batch_idx, (size_t)n_labels * 4, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(b2, cls, (size_t)n_labels * 4, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(b3, bboxes, (size_t)n_labels * 16, hipMemcpyHostToDevice, st);
      }
      d_rows = a; d_cnt = c; d_bi = b1; d_cl = b2; d_bb = b3; d_cor = co;
    }
  }
  int rc = YS_OK;
  if (!ok) { ys_set_error("ys_val_match_batched: out of device memory"); rc = YS_ERR_OOM; }
  if (rc == YS_OK) {
    hipMemsetAsync(d_ovf, 0, 4, st);
    YS_LAUNCH(val_match_kernel, batch, VM_THREADS, st, d_rows, d_cnt, max_det, row_stride, d_bi, d_cl, d_bb, n_labels, img_w, img_h,
              vm_thresholds(), lcap, d_lab, d_best, d_cor, d_ovf);
    hipError_t e = on_device ? hipSuccess : hipMemcpyAsync(correct, d_cor, ncor, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);      // the scratch buffers are released below
    if (e != hipSuccess) { ys_set_error("ys_val_match_batched: %s", hipGetErrorString(e)); rc = YS_ERR_HIP; }
  }
  for (void* p : tmp) hipFree(p);
  return rc;
}

}  // extern "C"
