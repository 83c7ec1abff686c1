/* oracle/nms_ref.c -- TEST INFRASTRUCTURE ONLY (the checker, never the product path).
 *
 * Plain-C, single-thread restatement of the reference's NMS path:
 *   YoloSharp/Utils/Ops.cs:239-371  non_max_suppression (pre-filter, class offset, max_det)
 *   YoloSharp/Utils/Ops.cs:68-81    xywh2xyxy (x - w/2, y - h/2, x + w/2, y + h/2)
 *   torchvision.ops.nms (NuGet TorchVision 0.105.2, single call site Ops.cs:357) -- third-party,
 *     not vendored in /root/reference; restated from torchvision's published CPU algorithm
 *     (torchvision/csrc/ops/cpu/nms_kernel.cpp): stable descending sort of the scores, then
 *       for i in order: if suppressed[i] continue; keep i;
 *         for j after i: inter = max(0,min(x2)-max(x1)) * max(0,min(y2)-max(y1));
 *                        ovr = inter / (area_i + area_j - inter); if (ovr > thr) suppressed[j] = 1;
 *     with areas = (x2-x1)*(y2-y1), all in fp32.
 * PARITY UNPINNED: the reference holds no golden vectors or tests for this path
 * and cannot be executed here (C#, no dotnet).  Tie order of equal scores is unspecified upstream;
 * the rule fixed here and in the HIP kernel is "stable: lower anchor index first".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -shared -fPIC)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float score; int32_t idx; } cand_t;

/* descending score, ties -> lower index first (stable w.r.t. anchor order) */
static int cand_cmp(const void* pa, const void* pb) {
  const cand_t* a = (const cand_t*)pa; const cand_t* b = (const cand_t*)pb;
  if (a->score > b->score) return -1;
  if (a->score < b->score) return 1;
  return (a->idx > b->idx) - (a->idx < b->idx);
}

static float fmaxf_(float a, float b) { return a > b ? a : b; }
static float fminf_(float a, float b) { return a < b ? a : b; }

/* pred [B,C,A] is modified in place exactly like the reference (boxes -> xyxy).
 * out_rows [B,max_det,6+extra], out_keep [B,max_det], out_count [B].  Returns 0, or 1 on invalid thresholds. */
int ys_oracle_nms(float* pred, int B, int C, int A, float conf_thres, float iou_thres, int max_det,
                  int nc, int max_nms, int max_wh, float* out_rows, int64_t* out_keep, int32_t* out_count) {
  if (conf_thres < 0.f || conf_thres > 1.f) return 1;   /* Ops.cs:248-251 */
  if (iou_thres < 0.f || iou_thres > 1.f) return 1;     /* Ops.cs:252-255 */
  if (nc == 0) nc = C - 4;                               /* Ops.cs:269 */
  const int extra = C - nc - 4;                          /* Ops.cs:270 */
  const int row_w = 6 + extra;
  cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * (size_t)A);
  int32_t* cls = (int32_t*)malloc(sizeof(int32_t) * (size_t)A);
  float* bx = (float*)malloc(sizeof(float) * 4 * (size_t)A);
  float* area = (float*)malloc(sizeof(float) * (size_t)A);
  unsigned char* supp = (unsigned char*)malloc((size_t)A);
  memset(out_rows, 0, sizeof(float) * (size_t)B * max_det * row_w);
  memset(out_keep, 0, sizeof(int64_t) * (size_t)B * max_det);
  for (int b = 0; b < B; b++) {
    float* p = pred + (size_t)b * C * A;
    /* Ops.cs:290: prediction[..., 0:4] = xywh2xyxy(...) for every anchor, in place */
    for (int a = 0; a < A; a++) {
      const float cx = p[a], cy = p[(size_t)A + a], w = p[2 * (size_t)A + a], h = p[3 * (size_t)A + a];
      p[a] = cx - w / 2; p[(size_t)A + a] = cy - h / 2;
      p[2 * (size_t)A + a] = cx + w / 2; p[3 * (size_t)A + a] = cy + h / 2;
    }
    /* Ops.cs:272,310,325-329: candidates amax(cls) > conf; (conf, j) = cls.max(1) (first max) */
    int n = 0;
    for (int a = 0; a < A; a++) {
      float best = p[(size_t)4 * A + a]; int bi = 0;
      for (int c = 1; c < nc; c++) { const float v = p[(size_t)(4 + c) * A + a]; if (v > best) { best = v; bi = c; } }
      if (best > conf_thres) { cand[n].score = best; cand[n].idx = a; n++; cls[a] = bi; }
    }
    out_count[b] = 0;
    if (n == 0) continue;                                 /* Ops.cs:315-318,333-336 */
    /* Ops.cs:338-342 (top max_nms by score) and torchvision's stable descending sort are one sort here */
    qsort(cand, (size_t)n, sizeof(cand_t), cand_cmp);
    if (n > max_nms) n = max_nms;
    for (int i = 0; i < n; i++) {
      const int a = cand[i].idx;
      const float off = (float)cls[a] * (float)max_wh;   /* Ops.cs:345 c = cls * max_wh */
      bx[4 * i + 0] = p[a] + off;                        /* Ops.cs:356 boxes = box + c */
      bx[4 * i + 1] = p[(size_t)A + a] + off;
      bx[4 * i + 2] = p[2 * (size_t)A + a] + off;
      bx[4 * i + 3] = p[3 * (size_t)A + a] + off;
      area[i] = (bx[4 * i + 2] - bx[4 * i + 0]) * (bx[4 * i + 3] - bx[4 * i + 1]);
      supp[i] = 0;
    }
    int kept = 0;
    for (int i = 0; i < n && kept < max_det; i++) {      /* Ops.cs:360 i = i[:max_det] */
      if (supp[i]) continue;
      const int a = cand[i].idx;
      float* r = out_rows + ((size_t)b * max_det + kept) * row_w;   /* Ops.cs:328,361 */
      r[0] = p[a]; r[1] = p[(size_t)A + a]; r[2] = p[2 * (size_t)A + a]; r[3] = p[3 * (size_t)A + a];
      r[4] = cand[i].score; r[5] = (float)cls[a];
      for (int e = 0; e < extra; e++) r[6 + e] = p[(size_t)(4 + nc + e) * A + a];
      out_keep[(size_t)b * max_det + kept] = a;
      kept++;
      const float ix1 = bx[4 * i], iy1 = bx[4 * i + 1], ix2 = bx[4 * i + 2], iy2 = bx[4 * i + 3], ia = area[i];
      for (int j = i + 1; j < n; j++) {
        if (supp[j]) continue;
        const float xx1 = fmaxf_(ix1, bx[4 * j]), yy1 = fmaxf_(iy1, bx[4 * j + 1]);
        const float xx2 = fminf_(ix2, bx[4 * j + 2]), yy2 = fminf_(iy2, bx[4 * j + 3]);
        const float w = fmaxf_(0.f, xx2 - xx1), h = fmaxf_(0.f, yy2 - yy1);
        const float inter = w * h;
        const float ovr = inter / (ia + area[j] - inter);
        if (ovr > iou_thres) supp[j] = 1;
      }
    }
    out_count[b] = kept;
  }
  free(cand); free(cls); free(bx); free(area); free(supp);
  return 0;
}
