// nms.hip -- batched per-class non-maximum suppression for gfx950.
// Replaces Utils/Ops.cs:239-371 (non_max_suppression) including the third-party
// torchvision.ops.nms call at Ops.cs:357.  Contract restated in SURVEY.md B.5:
//   * candidates: max_c pred[b,4+c,a] > conf (strict), class = first argmax
//   * boxes xywh -> xyxy IN PLACE for every anchor (Ops.cs:288-291, formula :76-79)
//   * if n > max_nms keep the max_nms best scores
//   * boxes offset by cls*max_wh in fp32 BEFORE the IoU arithmetic (Ops.cs:345,356)
//   * greedy: descending score (ties: lower anchor index first), suppress iff
//     inter/(area_i+area_j-inter) > iou  (fp32, IEEE division), keep <= max_det
// HBM/latency-bound integer+fp32 work: no MFMA.  Three kernels:
//   nms_filter : one thread per (image, anchor), coalesced along the anchor axis
//   nms_select : one 1024-thread workgroup per image: LDS bitonic sort of 64-bit
//                (score,index) keys, then the greedy pass as a workgroup-parallel
//                IoU sweep per kept box (O(kept*n), no n^2 mask in HBM)
//   (outputs are written by nms_select)
// Compiled with -ffp-contract=off: every fp32 op below rounds exactly like the CPU reference.
#include "ys_internal.h"

#define NMS_THREADS 1024
#define NMS_LDS_KEYS 16384

__global__ void __launch_bounds__(256)
nms_filter_kernel(float* __restrict__ pred, int C, int A, int nc, float conf_thres,
                  int* __restrict__ count, unsigned long long* __restrict__ keys, int keys_stride,
                  float* __restrict__ confs, int* __restrict__ clss) {
  const int b = blockIdx.y;
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A) return;
  float* p = pred + (size_t)b * C * A;
  // xywh -> xyxy in place (Ops.cs:76-79): x - w/2, y - h/2, x + w/2, y + h/2
  const float cx = p[a], cy = p[(size_t)A + a], w = p[2 * (size_t)A + a], h = p[3 * (size_t)A + a];
  const float hw = w / 2.0f, hh = h / 2.0f;
  p[a] = cx - hw;
  p[(size_t)A + a] = cy - hh;
  p[2 * (size_t)A + a] = cx + hw;
  p[3 * (size_t)A + a] = cy + hh;
  float best = p[4 * (size_t)A + a];
  int bi = 0;
  for (int c = 1; c < nc; c++) {
    const float v = p[(size_t)(4 + c) * A + a];
    if (v > best) { best = v; bi = c; }
  }
  if (best > conf_thres) {
    const int slot = atomicAdd(&count[b], 1);
    const unsigned bits = ys_f2u(best);  // best > conf >= 0 -> positive float, bit pattern monotone
    keys[(size_t)b * keys_stride + slot] = ((unsigned long long)(~bits) << 32) | (unsigned)a;
    confs[(size_t)b * A + a] = best;
    clss[(size_t)b * A + a] = bi;
  }
}

__device__ inline void nms_bitonic_sort(unsigned long long* k, int np2) {
  for (int size = 2; size <= np2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (np2 >> 1); t += NMS_THREADS) {
        const int pos = 2 * t - (t & (stride - 1));
        const int par = pos + stride;
        const bool asc = (pos & size) == 0;
        const unsigned long long x = k[pos], y = k[par];
        if ((x > y) == asc) { k[pos] = y; k[par] = x; }
      }
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(NMS_THREADS)
nms_select_kernel(const float* __restrict__ pred, int C, int A, int nc, float iou_thres, int max_det,
                  int max_nms, float max_wh, const int* __restrict__ count,
                  unsigned long long* __restrict__ keys_g, int keys_stride,
                  const float* __restrict__ confs, const int* __restrict__ clss,
                  float4* __restrict__ sbox, float* __restrict__ sarea, int* __restrict__ sidx,
                  unsigned char* __restrict__ supp_g, int ncap,
                  float* __restrict__ out_rows, long long* __restrict__ out_keep, int* __restrict__ out_count) {
  __shared__ unsigned long long skeys[NMS_LDS_KEYS];
  __shared__ int s_red[NMS_THREADS / 64];
  __shared__ int s_cur;
  __shared__ int s_kept[1];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const float* p = pred + (size_t)b * C * A;
  const int extra = C - 4 - nc;
  const int row_w = 6 + extra;
  int n = count[b];
  if (n > A) n = A;
  float* orow = out_rows + (size_t)b * max_det * row_w;
  long long* okeep = out_keep + (size_t)b * max_det;
  // outputs of images with nothing kept stay zero / count 0 (Ops.cs:298-299,315-318)
  for (int i = tid; i < max_det * row_w; i += NMS_THREADS) orow[i] = 0.0f;
  for (int i = tid; i < max_det; i += NMS_THREADS) okeep[i] = 0;
  if (n == 0) {
    if (tid == 0) out_count[b] = 0;
    return;
  }
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  unsigned long long* kg = keys_g + (size_t)b * keys_stride;
  unsigned long long* k = (np2 <= NMS_LDS_KEYS) ? skeys : kg;
  if (np2 <= NMS_LDS_KEYS) {
    for (int i = tid; i < np2; i += NMS_THREADS) skeys[i] = (i < n) ? kg[i] : ~0ull;
  } else {
    for (int i = n + tid; i < np2; i += NMS_THREADS) kg[i] = ~0ull;
  }
  __syncthreads();
  nms_bitonic_sort(k, np2);
  if (n > max_nms) n = max_nms;  // Ops.cs:338-342
  float4* bx = sbox + (size_t)b * ncap;
  float* ar = sarea + (size_t)b * ncap;
  int* si = sidx + (size_t)b * ncap;
  unsigned char* supp = supp_g + (size_t)b * ncap;
  for (int i = tid; i < n; i += NMS_THREADS) {
    const int a = (int)(k[i] & 0xffffffffull);
    const float off = (float)clss[(size_t)b * A + a] * max_wh;  // Ops.cs:345
    float4 q;
    q.x = p[a] + off;                       // Ops.cs:356 boxes = x[:, :4] + c
    q.y = p[(size_t)A + a] + off;
    q.z = p[2 * (size_t)A + a] + off;
    q.w = p[3 * (size_t)A + a] + off;
    bx[i] = q;
    ar[i] = (q.z - q.x) * (q.w - q.y);      // torchvision nms: areas = (x2-x1)*(y2-y1)
    si[i] = a;
    supp[i] = 0;
  }
  if (tid == 0) { s_cur = 0; s_kept[0] = 0; }
  __syncthreads();
  // greedy pass: workgroup-parallel IoU sweep per kept box
  int myp = tid;  // private monotone cursor over j == tid (mod NMS_THREADS)
  int kept = 0;
  for (;;) {
    const int i = s_cur;
    if (i >= n) break;
    if (tid == 0) {
      const int a = si[i];
      float* r = orow + (size_t)kept * row_w;
      r[0] = p[a]; r[1] = p[(size_t)A + a]; r[2] = p[2 * (size_t)A + a]; r[3] = p[3 * (size_t)A + a];
      r[4] = confs[(size_t)b * A + a];
      r[5] = (float)clss[(size_t)b * A + a];
      for (int e = 0; e < extra; e++) r[6 + e] = p[(size_t)(4 + nc + e) * A + a];
      okeep[kept] = a;
    }
    kept++;
    if (kept >= max_det) break;  // i = i[:max_det] (Ops.cs:360): later boxes can never be emitted
    const float4 bi = bx[i];
    const float ai = ar[i];
    // first j > i with j == tid (mod NMS_THREADS)
    int j0 = tid;
    if (j0 <= i) j0 += ((i - j0) / NMS_THREADS + 1) * NMS_THREADS;
    for (int j = j0; j < n; j += NMS_THREADS) {
      if (supp[j]) continue;
      const float4 bj = bx[j];
      const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
      const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
      const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
      const float inter = w * h;
      const float ovr = inter / (ai + ar[j] - inter);
      if (ovr > iou_thres) supp[j] = 1;
    }
    __syncthreads();
    // next unsuppressed index > i
    while (myp < n && (myp <= i || supp[myp])) myp += NMS_THREADS;
    int cand = (myp < n) ? myp : 0x7fffffff;
    for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(cand, m); cand = o < cand ? o : cand; }
    if ((tid & 63) == 0) s_red[tid >> 6] = cand;
    __syncthreads();
    if (tid == 0) {
      int mn = s_red[0];
      for (int w = 1; w < NMS_THREADS / 64; w++) mn = s_red[w] < mn ? s_red[w] : mn;
      s_cur = mn;
    }
    __syncthreads();
  }
  if (tid == 0) out_count[b] = kept;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int ys_nms_launch(ys_ctx* ctx, float* pred, int B, int C, int A, float conf, float iou, int max_det,
                  int nc, int max_nms, int max_wh, float* out_rows, int64_t* out_keep, int32_t* out_count) {
  int np2 = 1;
  while (np2 < A) np2 <<= 1;
  const int ncap = A < max_nms ? A : max_nms;
  // workspace carve-up
  size_t off = 0;
  const size_t o_count = off; off = align_up(off + sizeof(int) * B, 256);
  const size_t o_keys = off;  off = align_up(off + sizeof(unsigned long long) * (size_t)B * np2, 256);
  const size_t o_conf = off;  off = align_up(off + sizeof(float) * (size_t)B * A, 256);
  const size_t o_cls = off;   off = align_up(off + sizeof(int) * (size_t)B * A, 256);
  const size_t o_box = off;   off = align_up(off + sizeof(float4) * (size_t)B * ncap, 256);
  const size_t o_area = off;  off = align_up(off + sizeof(float) * (size_t)B * ncap, 256);
  const size_t o_idx = off;   off = align_up(off + sizeof(int) * (size_t)B * ncap, 256);
  const size_t o_supp = off;  off = align_up(off + (size_t)B * ncap, 256);
  if (off > ctx->nms_ws_bytes) {
    if (ctx->nms_ws) { YS_CHECK_HIP(hipStreamSynchronize(ctx->stream)); YS_CHECK_HIP(hipFree(ctx->nms_ws)); ctx->nms_ws = nullptr; ctx->nms_ws_bytes = 0; }
    YS_CHECK_HIP(hipMalloc(&ctx->nms_ws, off));
    ctx->nms_ws_bytes = off;
  }
  char* ws = (char*)ctx->nms_ws;
  int* count = (int*)(ws + o_count);
  YS_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int) * B, ctx->stream));
  YsKprofScope prof(ctx->stream, "nms");
  dim3 g1(ys_cdiv(A, 256), B);
  YS_LAUNCH(nms_filter_kernel, g1, 256, ctx->stream, pred, C, A, nc, conf, count,
            (unsigned long long*)(ws + o_keys), np2, (float*)(ws + o_conf), (int*)(ws + o_cls));
  YS_LAUNCH(nms_select_kernel, B, NMS_THREADS, ctx->stream, (const float*)pred, C, A, nc, iou, max_det, max_nms,
            (float)max_wh, (const int*)count, (unsigned long long*)(ws + o_keys), np2,
            (const float*)(ws + o_conf), (const int*)(ws + o_cls), (float4*)(ws + o_box), (float*)(ws + o_area),
            (int*)(ws + o_idx), (unsigned char*)(ws + o_supp), ncap, out_rows, (long long*)out_keep, (int*)out_count);
  YS_CHECK_HIP(hipGetLastError());
  return YS_OK;
}
