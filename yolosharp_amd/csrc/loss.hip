// loss.hip -- v8DetectionLoss forward + analytic backward, fully on device (no host syncs).
//
// Restates (reference file:line under YoloSharp/):
//   Utils/Loss.cs:411-477   get_assigned_targets_and_loss, loss (x batch_size)
//   Utils/Loss.cs:363-390   preprocess (pad GT per image, cxcywh*imgsz -> xyxy pixels)
//   Utils/Loss.cs:398-409   bbox_decode (softmax . arange(reg_max), dist2bbox xyxy)
//   Utils/Tal.cs:50-255     TaskAlignedAssigner (topk 10, alpha 0.5, beta 6, eps 1e-9)
//   Utils/Loss.cs:94-167    DFLoss / BboxLoss
//   Utils/Metrics.cs:36-111 bbox_iou(CIoU), eps 1e-7, alpha NOT detached
//   Utils/Tal.cs:313-379    make_anchors / dist2bbox / bbox2dist
// The reference reaches the host four times per step (Loss.cs:380,444,450; Tal.cs:231); here
// n_max_boxes / target_scores_sum / fg counts stay in HBM and only ys_loss_read() synchronises.
// Gradients of sum(loss*B) w.r.t. the head outputs are written directly (dpd, dps): the loss has
// closed-form derivatives, CIoU's through forward-mode dual numbers (4 partials).
// All arithmetic is fp32; pd/ps/dpd/dps are stored in the model dtype T.
#include "ys_internal.h"
#include "ys_kernels.h"

#define LS_THREADS 256
#define TAL_LIST_CAP 8192   // in-box anchors of one (image, box) pair walked through an LDS list (16 KB of 16-bit indices; A <= 65535 on that path)
#ifndef TAL_T
#define TAL_T 256           // threads of a tal_metrics_kernel workgroup.  Measured (round 4, ~550 pairs, flat grid): 128 threads 74 us, 256 64 us, 512 105 us, 1024 119 us
#endif
#define TAL_REGS ((8448 + TAL_T - 1) / TAL_T)   // assigner top-k: metrics per thread kept in registers (covers the 8400 anchors of 640x640)
#define CIOU_EPS 1e-7f

// ------------------------------------------------------------------ dual numbers (value + N partials)
template <int N>
struct DualN {
  float v;
  float g[N];
};
typedef DualN<4> Dual4;   // CIoU: d/d(x1, y1, x2, y2)
typedef DualN<5> Dual5;   // probiou: d/d(x, y, w, h, angle)
template <int N> __device__ inline DualN<N> dconstN(float c) { DualN<N> r; r.v = c; for (int i = 0; i < N; i++) r.g[i] = 0.f; return r; }
template <int N> __device__ inline DualN<N> dvarN(float c, int i) { DualN<N> r = dconstN<N>(c); r.g[i] = 1.f; return r; }
__device__ inline Dual4 dconst(float c) { return dconstN<4>(c); }
__device__ inline Dual4 dvar(float c, int i) { return dvarN<4>(c, i); }
template <int N> __device__ inline DualN<N> operator+(const DualN<N>& a, const DualN<N>& b) { DualN<N> r; r.v = a.v + b.v; for (int i = 0; i < N; i++) r.g[i] = a.g[i] + b.g[i]; return r; }
template <int N> __device__ inline DualN<N> operator-(const DualN<N>& a, const DualN<N>& b) { DualN<N> r; r.v = a.v - b.v; for (int i = 0; i < N; i++) r.g[i] = a.g[i] - b.g[i]; return r; }
template <int N> __device__ inline DualN<N> operator*(const DualN<N>& a, const DualN<N>& b) { DualN<N> r; r.v = a.v * b.v; for (int i = 0; i < N; i++) r.g[i] = a.g[i] * b.v + a.v * b.g[i]; return r; }
template <int N> __device__ inline DualN<N> operator/(const DualN<N>& a, const DualN<N>& b) {
  DualN<N> r; r.v = a.v / b.v;
  const float inv = 1.0f / b.v;
  for (int i = 0; i < N; i++) r.g[i] = (a.g[i] - r.v * b.g[i]) * inv;
  return r;
}
template <int N> __device__ inline DualN<N> dscale(const DualN<N>& a, float d) { DualN<N> r; r.v = a.v; for (int i = 0; i < N; i++) r.g[i] = a.g[i] * d; return r; }
template <int N> __device__ inline DualN<N> dmax(const DualN<N>& a, const DualN<N>& b) { return a.v >= b.v ? a : b; }
template <int N> __device__ inline DualN<N> dmin(const DualN<N>& a, const DualN<N>& b) { return a.v <= b.v ? a : b; }
template <int N> __device__ inline DualN<N> dclamp_min(const DualN<N>& a, float lo) { return a.v >= lo ? a : dconstN<N>(lo); }
template <int N> __device__ inline DualN<N> dclamp(const DualN<N>& a, float lo, float hi) { return a.v < lo ? dconstN<N>(lo) : (a.v > hi ? dconstN<N>(hi) : a); }
template <int N> __device__ inline DualN<N> datan(const DualN<N>& a) { DualN<N> r = dscale(a, 1.0f / (1.0f + a.v * a.v)); r.v = atanf(a.v); return r; }
template <int N> __device__ inline DualN<N> dsqrt(const DualN<N>& a) { const float q = sqrtf(a.v); DualN<N> r = dscale(a, 0.5f / q); r.v = q; return r; }
template <int N> __device__ inline DualN<N> dlog(const DualN<N>& a) { DualN<N> r = dscale(a, 1.0f / a.v); r.v = logf(a.v); return r; }
template <int N> __device__ inline DualN<N> dexp(const DualN<N>& a) { const float e = expf(a.v); DualN<N> r = dscale(a, e); r.v = e; return r; }
template <int N> __device__ inline DualN<N> dcos(const DualN<N>& a) { DualN<N> r = dscale(a, -sinf(a.v)); r.v = cosf(a.v); return r; }
template <int N> __device__ inline DualN<N> dsin(const DualN<N>& a) { DualN<N> r = dscale(a, cosf(a.v)); r.v = sinf(a.v); return r; }
// plain-float overloads so ciou<R>() / probiou<R>() are written once
__device__ inline float dmax(float a, float b) { return a >= b ? a : b; }
__device__ inline float dmin(float a, float b) { return a <= b ? a : b; }
__device__ inline float dclamp_min(float a, float lo) { return a >= lo ? a : lo; }
__device__ inline float dclamp(float a, float lo, float hi) { return a < lo ? lo : (a > hi ? hi : a); }
__device__ inline float datan(float a) { return atanf(a); }
__device__ inline float dsqrt(float a) { return sqrtf(a); }
__device__ inline float dlog(float a) { return logf(a); }
__device__ inline float dexp(float a) { return expf(a); }
__device__ inline float dcos(float a) { return cosf(a); }
__device__ inline float dsin(float a) { return sinf(a); }
template <class R> __device__ inline R rconst(float c);
template <> __device__ inline float rconst<float>(float c) { return c; }
template <> __device__ inline Dual4 rconst<Dual4>(float c) { return dconstN<4>(c); }
template <> __device__ inline Dual5 rconst<Dual5>(float c) { return dconstN<5>(c); }

// Metrics.probiou (Metrics.cs:137-160, _get_covariance_matrix :264-283): boxes (x, y, w, h, angle), eps 1e-7
template <class R>
__device__ inline void obb_cov(const R o[5], R& A, R& B, R& C) {
  const R a = o[2] * o[2] / rconst<R>(12.0f), b = o[3] * o[3] / rconst<R>(12.0f);
  const R cs = dcos(o[4]), sn = dsin(o[4]);
  const R c2 = cs * cs, s2 = sn * sn;
  A = a * c2 + b * s2;
  B = a * s2 + b * c2;
  C = (a - b) * cs * sn;
}
template <class R>
__device__ inline R probiou_t(const R o1[5], const R o2[5]) {
  const float eps = 1e-7f;
  R a1, b1, c1, a2, b2, c2;
  obb_cov<R>(o1, a1, b1, c1);
  obb_cov<R>(o2, a2, b2, c2);
  const R sa = a1 + a2, sb = b1 + b2, sc = c1 + c2;
  const R den = sa * sb - sc * sc;
  const R dy = o1[1] - o2[1], dxa = o1[0] - o2[0], dxb = o2[0] - o1[0];
  const R t1 = (sa * (dy * dy) + sb * (dxa * dxa)) / (den + rconst<R>(eps)) * rconst<R>(0.25f);
  const R t2 = (sc * dxb * dy) / (den + rconst<R>(eps)) * rconst<R>(0.5f);
  const R d1 = dclamp_min(a1 * b1 - c1 * c1, 0.0f), d2 = dclamp_min(a2 * b2 - c2 * c2, 0.0f);
  const R t3 = dlog(den / (rconst<R>(4.0f) * dsqrt(d1 * d2) + rconst<R>(eps)) + rconst<R>(eps)) * rconst<R>(0.5f);
  const R bd = dclamp(t1 + t2 + t3, eps, 100.0f);
  const R hd = dsqrt(rconst<R>(1.0f) - dexp(rconst<R>(0.0f) - bd) + rconst<R>(eps));
  return rconst<R>(1.0f) - hd;
}

// Metrics.cs:36-111 with xywh=false, CIoU=true
template <class R>
__device__ inline R ciou_xyxy(const R b1[4], const R b2[4]) {
  const R w1 = b1[2] - b1[0];
  const R h1 = dclamp_min(b1[3] - b1[1], CIOU_EPS);
  const R w2 = b2[2] - b2[0];
  const R h2 = dclamp_min(b2[3] - b2[1], CIOU_EPS);
  const R inter = dclamp_min(dmin(b1[2], b2[2]) - dmax(b1[0], b2[0]), 0.0f) *
                  dclamp_min(dmin(b1[3], b2[3]) - dmax(b1[1], b2[1]), 0.0f);
  const R uni = w1 * h1 + w2 * h2 - inter + rconst<R>(CIOU_EPS);
  const R iou = inter / uni;
  const R cw = dmax(b1[2], b2[2]) - dmin(b1[0], b2[0]);
  const R ch = dmax(b1[3], b2[3]) - dmin(b1[1], b2[1]);
  const R c2 = cw * cw + ch * ch + rconst<R>(CIOU_EPS);
  const R dx = b2[0] + b2[2] - b1[0] - b1[2];
  const R dy = b2[1] + b2[3] - b1[1] - b1[3];
  const R rho2 = (dx * dx + dy * dy) / rconst<R>(4.0f);
  const R da = datan(w2 / h2) - datan(w1 / h1);
  const R v = rconst<R>(0.40528473456935108577f) * (da * da);  // 4/pi^2
  const R alpha = v / (v - iou + rconst<R>(1.0f + CIOU_EPS));
  return iou - (rho2 / c2 + v * alpha);
}

struct AnchorInfo { float ax, ay, stride; };
// Constant indices into the level tables (kernel arguments): scalar loads the compiler hoists out of the anchor loops.  The round-1 form -- a search loop
// over a.lvl_off[i], then a.lvl_w[l] with a per-lane l -- compiled to one kernel-argument load + wait per search step and a per-lane global load
// from the argument segment for every anchor: ~1 us per loop iteration of tal_metrics_kernel (three phases of ~33 iterations at 35 us each).
__device__ inline AnchorInfo anchor_of(const LossArgs& a, int idx) {
  int off = a.lvl_off[0], w = a.lvl_w[0], st = a.lvl_stride[0];
#pragma unroll
  for (int i = 1; i < 4; i++) {
    const bool ge = (bool)((int)(i < a.nl) & (int)(idx >= a.lvl_off[i]));   // offsets increase with the level: the last hit wins
    off = ge ? a.lvl_off[i] : off; w = ge ? a.lvl_w[i] : w; st = ge ? a.lvl_stride[i] : st;
  }
  const int cell = idx - off;                 // < 2^24: float reciprocal + fix-up is exact
  int y = (int)((float)cell * (1.0f / (float)w));
  int x = cell - y * w;
  if (x >= w) { y++; x -= w; }
  if (x < 0) { y--; x += w; }
  AnchorInfo r;
  r.ax = (float)x + 0.5f;   // Tal.cs:325-326 grid_cell_offset 0.5
  r.ay = (float)y + 0.5f;
  r.stride = (float)st;
  return r;
}

// ---- round 6: the criterion's four global sums as fixed-point integer accumulators (the scheme of the BatchNorm statistics, ys_kernels.h ys_stat_acc_add).
// Rounds 1-5: every producer wrote one partial row per workgroup and a one-workgroup loss_sum_kernel launch added them -- three 7 us launches (+ boundaries) on the
// step's critical path between forward and backward.  Now a workgroup adds its partial -- rounded to 2^-30 (target-score sum: |partial| <= 256) or 2^-20 (loss sums:
// |partial| < 2^42) -- with ONE 64-bit integer atomic into shard (workgroup index mod LOSS_SHARDS): integer addition commutes, so the totals are bit-reproducible
// whatever the arrival order.  (Unsharded -- 21 thousand workgroups of loss_cls_kernel adding to ONE address -- the atomics serialised in the L2: 51 -> 261 us.)
// Consumers: loss_items sums the shards of every word; the target-score sum is folded into scalars[0] by tal_targets_kernel's last workgroup (below).  The words live behind the 64 float scalars:
// (u64*)(scalars + 64) [shard][8]: [0] tss, [1] cls, [2] iou, [3] dfl, [4] angle, [5] poison (a non-finite or out-of-range partial: the items become NaN, as the
// reference's do on divergence).
#define LOSS_SHARDS 64
#define LOSS_ACC(a) ((unsigned long long*)((a).scalars + 64))
#define LOSS_FIX_T 1073741824.0f      // 2^30
#define LOSS_FIX_L 1048576.0f         // 2^20
__device__ inline void loss_acc_add(unsigned long long* acc, int word, float t, float fix, int shards = LOSS_SHARDS) {
  unsigned long long* a = acc + (size_t)(blockIdx.x & (shards - 1)) * 8;
  if (!(fabsf(t) * fix < 9.0e18f)) { atomicOr(a + 5, 1ull); return; }       // NaN fails the comparison too
#ifdef YS_EMU_BUILD
  const long long q = (long long)llrintf(t * fix);
#else
  const long long q = __float2ll_rn(t * fix);
#endif
  if (q != 0) atomicAdd(a + word, (unsigned long long)q);
}
// one thread: every shard of a word (loss_items_kernel)
__device__ inline float loss_acc_get(const unsigned long long* acc, int word, float fix) {
  long long t = 0; unsigned long long bad = 0;
  for (int s = 0; s < LOSS_SHARDS; s++) { t += (long long)acc[s * 8 + word]; bad |= acc[s * 8 + 5]; }
  if (bad) return __builtin_nanf("");
  return (float)((double)t * (1.0 / (double)fix));
}
__device__ inline float loss_tss_of(float t) { return t > 1.0f ? t : (t != t ? t : 1.0f); }        // target_scores_sum = max(sum, 1) (Loss.cs:444); NaN stays NaN
// The target-score sum has readers in every workgroup of loss_cls / loss_box, so it must be ONE float again by the time they start (reading and adding its
// shards per wave -- 64-bit shuffles, an i64 -> f64 -> f32 conversion per lane -- cost loss_cls_kernel 51 -> 74 us; a workgroup-wide read behind a barrier 92 us).
// tal_targets_kernel's LAST workgroup to arrive does it: a workgroup adds its partial with a RETURNING atomic, then takes a ticket with a second atomic that is only
// issued once the first has returned (the empty asm makes the ticket's operand depend on the returned value); whoever draws the last ticket reads the shards back
// with atomics (coherent at the point where device-scope atomics execute: no fence, no L2 write-back -- every datum involved is an atomic's) and stores
// scalars[0] = max(sum, 1), which the following launches read as they did in rounds 1-5.  Ticket words: word 6 of every shard (first level), shard 0 word 7 (second).
#define LOSS_TSS_SHARDS 16
// workgroup sum of up to four values -> lane 0 of wave 0 (fixed order: wave butterflies, then the waves in index order)
__device__ inline void block_sum4(float& v0, float& v1, float& v2, float& v3) {
  __shared__ float s[LS_THREADS / 64][4];
  v0 = ys_wave_sum(v0); v1 = ys_wave_sum(v1); v2 = ys_wave_sum(v2); v3 = ys_wave_sum(v3);
  const int tid = threadIdx.x;
  if ((tid & 63) == 0) { s[tid >> 6][0] = v0; s[tid >> 6][1] = v1; s[tid >> 6][2] = v2; s[tid >> 6][3] = v3; }
  __syncthreads();
  if (tid == 0) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < LS_THREADS / 64; w++) for (int k = 0; k < 4; k++) t[k] += s[w][k];
    v0 = t[0]; v1 = t[1]; v2 = t[2]; v3 = t[3];
  }
}

// ------------------------------------------------------------------ K0: GT padding (Loss.cs:363-390,431)
// (round 6: runs as the FIRST workgroup of the bbox_decode launch -- the two are independent -- instead of a one-workgroup launch of its own)
__device__ inline void loss_prep_body(const LossArgs& a, int* gt_valid) {
  const int tid = threadIdx.x;
  for (int i = tid; i < LOSS_SHARDS * 8; i += LS_THREADS) LOSS_ACC(a)[i] = 0ull;
  for (int i = tid; i < a.B; i += LS_THREADS) a.gt_count[i] = 0;
  for (int i = tid; i < a.B * a.gcap; i += LS_THREADS) { a.pos_align[i] = 0u; a.pos_ov[i] = 0u; }
  if (tid < 8) a.scalars[tid] = 0.f;
  if (tid == 12 || tid == 13) a.scalars[tid] = 0.f;
  const int bs_ = a.rot ? 5 : 4;                       // floats per label box / per padded GT row
  // v8OBBLoss drops oriented boxes thinner than 2 pixels before padding (Loss.cs:561-563); a dropped label takes no slot
  auto kept = [&](int i) { return !a.rot || (a.bboxes[5 * i + 2] * (float)a.W >= 2.0f && a.bboxes[5 * i + 3] * (float)a.H >= 2.0f); };
  // image index of every label staged in LDS: the rank loop below is O(n^2) and was a chain of global loads (57 us for 544 labels)
  constexpr int SB = 4096;
  __shared__ short s_b[SB];
  const bool staged = a.n_labels <= SB;
  if (staged) for (int i = tid; i < a.n_labels; i += LS_THREADS) { const int bb = (int)a.batch_idx[i]; s_b[i] = (short)(bb < -1 || bb > 32766 || !kept(i) ? -1 : bb); }
  __syncthreads();
  for (int i = tid; i < a.n_labels; i += LS_THREADS) {
    const int b = (int)a.batch_idx[i];
    if (b < 0 || b >= a.B || !kept(i)) continue;
    int slot = 0;  // rank among labels of the same image, in order of appearance
    if (staged) { for (int j = 0; j < i; j++) slot += ((int)s_b[j] == b) ? 1 : 0; }
    else { for (int j = 0; j < i; j++) slot += ((int)a.batch_idx[j] == b && kept(j)) ? 1 : 0; }
    atomicAdd(&a.gt_count[b], 1);
    if (slot >= a.gcap) continue;  // capacity exceeded (count is clamped below)
    const float sw = (float)a.W, shh = (float)a.H;
    const float cx = a.bboxes[bs_ * i + 0] * sw, cy = a.bboxes[bs_ * i + 1] * shh;
    float w = a.bboxes[bs_ * i + 2] * sw, h = a.bboxes[bs_ * i + 3] * shh;
    float* gb = a.gt_box + ((long)b * a.gcap + slot) * bs_;
    if (a.rot) {
      // xywh * imgsz, angle as given (Loss.cs:524-527); mask_gt = sum(xywhr) > 0 (Loss.cs:571).  RotatedTaskAlignedAssigner then
      // overwrites, IN the padded tensor, widths / heights below stride[0] with stride_val for valid rows (Tal.cs:283-287): every
      // later use (overlaps, targets, losses) sees the widened box, so it is applied here once
      const float r = a.bboxes[5 * i + 4];
      const int vld = (cx + cy + w + h + r) > 0.0f ? 1 : 0;
      const float s0 = (float)a.lvl_stride[0], sv = (float)a.lvl_stride[a.nl > 1 ? 1 : 0];
      if (vld && w < s0) w = sv;
      if (vld && h < s0) h = sv;
      gb[0] = cx; gb[1] = cy; gb[2] = w; gb[3] = h; gb[4] = r;
      a.gt_cls[(long)b * a.gcap + slot] = (int)a.cls[i];
      gt_valid[(long)b * a.gcap + slot] = vld;
      gt_valid[(long)a.B * a.gcap + (long)b * a.gcap + slot] = i;
      continue;
    }
    gb[0] = cx - w / 2; gb[1] = cy - h / 2; gb[2] = cx + w / 2; gb[3] = cy + h / 2;  // Ops.cs:76-79
    a.gt_cls[(long)b * a.gcap + slot] = (int)a.cls[i];
    gt_valid[(long)b * a.gcap + slot] = (gb[0] + gb[1] + gb[2] + gb[3]) > 0.0f ? 1 : 0;  // Loss.cs:431
    gt_valid[(long)a.B * a.gcap + (long)b * a.gcap + slot] = i;   // gt_src: the label row of this slot (keypoints of v8PoseLoss, Loss.cs:1001-1005)
  }
  __syncthreads();
  // The reference pads to the batch's true per-image maximum (Loss.cs:363-390); a fixed-capacity workspace must never drop
  // labels silently: the largest per-image count is recorded (scalars[15]) and ys_loss_read* / ys_model_backward refuse the
  // result when it exceeds gcap (host-label calls grow the workspace up front, so only device-label callers can see this).
  if (tid == 0) {
    int mx = 0;
    for (int i = 0; i < a.B; i++) mx = a.gt_count[i] > mx ? a.gt_count[i] : mx;
    a.scalars[15] = (float)mx;
  }
  __syncthreads();
  for (int i = tid; i < a.B; i += LS_THREADS)
    if (a.gt_count[i] > a.gcap) a.gt_count[i] = a.gcap;
}

// ------------------------------------------------------------------ K1: bbox_decode (Loss.cs:398-409)
// four consecutive lanes own one anchor (one side each): 32-byte contiguous logit reads instead of a 128-byte stride per lane
template <class T, int RR>     // RR = reg_max when it is 16 (row of 16 bins as 16-byte vector loads, unrolled), 0 = run-time
__global__ void __launch_bounds__(LS_THREADS)
loss_decode_kernel(LossArgs a, int* gt_valid) {
  if (blockIdx.x == 0) { loss_prep_body(a, gt_valid); return; }     // workgroup-uniform; dispatched first: its O(labels^2) rank loop is the longest workgroup of the launch
  constexpr int EPL = Elem<T>::EPL;
  const long i = (long)(blockIdx.x - 1) * blockDim.x + threadIdx.x;
  const long total = (long)a.B * a.A * 4;
  const bool inb = i < total;
  const long row = inb ? i >> 2 : 0;
  const int s = (int)(i & 3);
  const int lane = threadIdx.x & 63;
  float d = 0.f;
  if (inb) {
    const int R = RR ? RR : a.reg_max;
    const T* lr = (const T*)a.pd + row * a.ld_pd + s * R;
    float mx = -INFINITY, se = 0.f, sw = 0.f;
    if (RR) {
      float x[RR ? RR : 1];
#pragma unroll
      for (int v = 0; v < RR / EPL; v++) ys_unpack<T>(ys_ld16(lr + v * EPL), x + v * EPL);
#pragma unroll
      for (int j = 0; j < RR; j++) mx = fmaxf(mx, x[j]);
#pragma unroll
      for (int j = 0; j < RR; j++) { const float e = __expf(x[j] - mx); se += e; sw += e * (float)j; }
    } else {
      for (int j = 0; j < R; j++) mx = fmaxf(mx, Elem<T>::to_f(lr[j]));
      for (int j = 0; j < R; j++) { const float e = __expf(Elem<T>::to_f(lr[j]) - mx); se += e; sw += e * (float)j; }
    }
    d = sw / se;
  }
  const int base = lane & ~3;
  const float d0 = __shfl(d, base + 0), d1 = __shfl(d, base + 1), d2 = __shfl(d, base + 2), d3 = __shfl(d, base + 3);
  if (inb && s == 0 && a.rot) {   // bbox_decode of v8OBBLoss (Loss.cs:634-645): dist2rbox (Tal.cs:389-408) + the angle, grid units
    const AnchorInfo an = anchor_of(a, (int)(row % a.A));
    const float ang = (ys_sigmoid(Elem<T>::to_f(((const T*)a.pa)[row * a.ld_pa])) - 0.25f) * 3.14159265358979323846f;
    const float cs = cosf(ang), sn = sinf(ang);
    const float xf = (d2 - d0) / 2.0f, yf = (d3 - d1) / 2.0f;
    float* o = a.pbox + row * 5;
    o[0] = xf * cs - yf * sn + an.ax; o[1] = xf * sn + yf * cs + an.ay; o[2] = d0 + d2; o[3] = d1 + d3; o[4] = ang;
  } else if (inb && s == 0) {
    const AnchorInfo an = anchor_of(a, (int)(row % a.A));
    float4 o;
    o.x = an.ax - d0; o.y = an.ay - d1; o.z = an.ax + d2; o.w = an.ay + d3;   // Tal.cs:345-346
    *(float4*)(a.pbox + row * 4) = o;
  }
}

// ------------------------------------------------------------------ K2: metrics + top-k per (image, gt)
template <class T, bool ROT>     // ROT: RotatedTaskAlignedAssigner (Tal.cs:260-310) -- compile-time so the plain path keeps its registers
__device__ inline void tal_metrics_pair(const LossArgs& a, const int* __restrict__ gt_valid, const int g, const int b) {
  __shared__ unsigned s_ingt[1056];   // A <= 33792 anchors
  __shared__ unsigned s_taken[1056];
  __shared__ float s_v[TAL_T / 64];
  __shared__ int s_i[TAL_T / 64];
  __shared__ int s_sel;
  const int tid = threadIdx.x;
  const long gi = (long)b * a.gcap + g;
  const bool valid = gt_valid[gi] != 0;
  const float* gb = a.gt_box + gi * (ROT ? 5 : 4);
  const float g4[4] = {gb[0], gb[1], gb[2], gb[3]};
  const float g5[5] = {gb[0], gb[1], gb[2], gb[3], ROT ? gb[4] : 0.f};
  const int cls = a.gt_cls[gi];
  float* ovr = a.ov + gi * a.A;
  float* alr = a.align + gi * a.A;
  unsigned char* mp = a.mpos + gi * a.A;
  const int nw = (a.A + 31) / 32;
  for (int i = tid; i < nw; i += TAL_T) { s_ingt[i] = 0u; s_taken[i] = 0u; }
  __syncthreads();
  // select_candidates_in_gts (Tal.cs:202-223): boxes smaller than stride[0]=8 are inflated to stride[1]=16
  float cx = (g4[0] + g4[2]) / 2, cy = (g4[1] + g4[3]) / 2, w = g4[2] - g4[0], h = g4[3] - g4[1];  // Ops.cs:98-101
  if (valid && w < (float)a.lvl_stride[0]) w = (float)a.lvl_stride[a.nl > 1 ? 1 : 0];
  if (valid && h < (float)a.lvl_stride[0]) h = (float)a.lvl_stride[a.nl > 1 ? 1 : 0];
  const float ix1 = cx - w / 2, iy1 = cy - h / 2, ix2 = cx + w / 2, iy2 = cy + h / 2;
  // rotated in-box test (Tal.cs:289-306): corners a = ctr + v1 + v2, b = ctr + v1 - v2, d = ctr - v1 + v2 (Ops.cs:24-33)
  float cax = 0.f, cay = 0.f, abx = 0.f, aby = 0.f, adx = 0.f, ady = 0.f, nab = 0.f, nad = 0.f;
  if (ROT) {
    const float cs = cosf(g5[4]), sn = sinf(g5[4]);
    const float v1x = g5[2] / 2 * cs, v1y = g5[2] / 2 * sn, v2x = -g5[3] / 2 * sn, v2y = g5[3] / 2 * cs;
    cax = g5[0] + v1x + v2x; cay = g5[1] + v1y + v2y;
    const float cbx = g5[0] + v1x - v2x, cby = g5[1] + v1y - v2y, cdx = g5[0] - v1x + v2x, cdy = g5[1] - v1y + v2y;
    abx = cbx - cax; aby = cby - cay; adx = cdx - cax; ady = cdy - cay;
    nab = abx * abx + aby * aby; nad = adx * adx + ady * ady;
  }
  // Two passes (round 4).  The metric of an anchor outside the box is zero, and a box holds ~10 % of the anchors: evaluated inside the anchor loop, every
  // iteration in which ANY lane of a wave was inside ran the whole CIoU / score path -- two dependent global loads behind an exec-masked branch, i.e. one L2
  // round trip per iteration, 33 iterations per thread (120 us for ~550 boxes).  Pass 1 is the in-box test alone (bit set, zeros for the anchors outside);
  // the set is compacted into a list in LDS and pass 2 walks it densely: every lane evaluates an anchor, ~3 iterations per thread.
  auto in_box = [&](const AnchorInfo& an) -> bool {
    const float px = an.ax * an.stride, py = an.ay * an.stride;   // anchor_points * stride_tensor (Loss.cs:439)
    const float dmin_ = fminf(fminf(px - ix1, py - iy1), fminf(ix2 - px, iy2 - py));
    bool ingt = dmin_ > 1e-9f;                                     // Tal.cs:221
    if (ROT) {
      const float apx = px - cax, apy = py - cay;
      const float dab = apx * abx + apy * aby, dad = apx * adx + apy * ady;
      ingt = dab >= 0.f && dab <= nab && dad >= 0.f && dad <= nad;
    }
    return ingt;
  };
  // an anchor inside a valid box: the operands (predicted box, class logit) are requested first, for several anchors at a time, so that a trip
  // of the dense loop pays one L2 round trip instead of one per anchor (the largest boxes hold ~3 thousand anchors = 12 trips, and they set the kernel's duration)
  struct MetIn { float pb[ROT ? 5 : 4]; T ps; };
  const int cc = cls < 0 ? 0 : (cls >= a.nc ? a.nc - 1 : cls);
  auto met_load = [&](int ai, MetIn& in) {
    const float* pb = a.pbox + ((long)b * a.A + ai) * (ROT ? 5 : 4);
    if (ROT) { for (int e = 0; e < 5; e++) in.pb[e] = pb[e]; }
    else { const float4 v = *(const float4*)pb; in.pb[0] = v.x; in.pb[1] = v.y; in.pb[2] = v.z; in.pb[3] = v.w; }
    in.ps = ((const T*)a.ps)[((long)b * a.A + ai) * a.ld_ps + cc];
  };
  auto met_eval = [&](const MetIn& in, const AnchorInfo& an, float& o, float& al) {
    if (ROT) {
      const float p5[5] = {in.pb[0] * an.stride, in.pb[1] * an.stride, in.pb[2] * an.stride, in.pb[3] * an.stride, in.pb[ROT ? 4 : 0]};   // Loss.cs:580-583
      o = probiou_t<float>(g5, p5);                              // Tal.cs:267-270 (obb1 = gt, obb2 = pred)
    } else {
      const float p4[4] = {in.pb[0] * an.stride, in.pb[1] * an.stride, in.pb[2] * an.stride, in.pb[3] * an.stride};  // Loss.cs:438
      o = ciou_xyxy<float>(g4, p4);                              // Tal.cs:141 (box1 = gt, box2 = pred)
    }
    o = o > 0.f ? o : 0.f;                                        // .clamp(0)
    const float sc = ys_sigmoid(Elem<T>::to_f(in.ps));
    al = sqrtf(sc) * powf(o, 6.0f);                               // Tal.cs:134 (alpha 0.5, beta 6)
  };
  auto metrics = [&](int ai, const AnchorInfo& an, float& o, float& al) { MetIn in; met_load(ai, in); met_eval(in, an, o, al); };
  __shared__ unsigned short s_list[TAL_LIST_CAP];
  __shared__ int s_wsum[TAL_T / 64];
  __shared__ int s_nin;
  // pass 1: in-box bits; the anchors outside (and every anchor of an invalid box) get their zeros here
  for (int ai = tid; ai < a.A; ai += TAL_T) {
    const bool ingt = in_box(anchor_of(a, ai));
    if (!(ingt && valid)) { ovr[ai] = 0.f; alr[ai] = 0.f; }
    mp[ai] = 0;
    if (ingt) atomicOr(&s_ingt[ai >> 5], 1u << (ai & 31));
  }
  __syncthreads();
#if defined(TAL_ABLATE) && TAL_ABLATE == 1
  return;
#endif
  if (valid) {
    // compaction: thread t owns the words [t * wpt, (t + 1) * wpt) -- ascending anchor order in the list, so the result does not depend on timing
    const int wpt = (nw + TAL_T - 1) / TAL_T;
    int cnt = 0;
    for (int k = 0; k < wpt; k++) { const int wi = tid * wpt + k; if (wi < nw) cnt += __popc(s_ingt[wi]); }
    int incl = cnt;
    for (int m = 1; m < 64; m <<= 1) { const int v = __shfl_up(incl, m); if ((tid & 63) >= m) incl += v; }
    if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    int base = incl - cnt;
    for (int wv = 0; wv < (tid >> 6); wv++) base += s_wsum[wv];
    if (tid == TAL_T - 1) s_nin = base + cnt;
    __syncthreads();
    const int nin = s_nin;
    if (nin <= TAL_LIST_CAP) {
      for (int k = 0; k < wpt; k++) {
        const int wi = tid * wpt + k;
        unsigned bits = wi < nw ? s_ingt[wi] : 0u;
        while (bits) { const int bt = __ffsll((unsigned long long)bits) - 1; bits &= bits - 1u; s_list[base++] = (unsigned short)(wi * 32 + bt); }
      }
      __syncthreads();
      // pass 2: dense over the list
      constexpr int DU = 4;
      for (int k0 = tid; k0 < nin; k0 += TAL_T * DU) {
        int ai[DU];
        MetIn in[DU];
#pragma unroll
        for (int u = 0; u < DU; u++) { const int k = k0 + u * TAL_T; ai[u] = k < nin ? (int)s_list[k] : -1; }
#pragma unroll
        for (int u = 0; u < DU; u++) met_load(ai[u] >= 0 ? ai[u] : 0, in[u]);      // unconditional: all requests of the trip in flight together
#pragma unroll
        for (int u = 0; u < DU; u++) {
          if (ai[u] >= 0) {
            float o, al;
            met_eval(in[u], anchor_of(a, ai[u]), o, al);
            ovr[ai[u]] = o;
            alr[ai[u]] = al;
          }
        }
      }
    } else {                                                       // a box that holds more anchors than the list: the one-pass form
      for (int ai = tid; ai < a.A; ai += TAL_T) {
        if (s_ingt[ai >> 5] & (1u << (ai & 31))) {
          float o, al;
          metrics(ai, anchor_of(a, ai), o, al);
          ovr[ai] = o;
          alr[ai] = al;
        }
      }
    }
  }
  __syncthreads();                                                  // the top-k rounds read other threads' alr entries (global, same workgroup)
#if defined(TAL_ABLATE) && TAL_ABLATE == 2
  return;
#endif
  // select_topk_candidates (Tal.cs:144-168): 10 largest align values; ties -> lowest anchor index
  if (a.A <= TAL_REGS * TAL_T) {
    // every BASELINE shape (A = 8400 <= 33 * 256): a thread's metrics (anchors tid + j * 256) stay in registers for the ten rounds --
    // one batch of independent loads instead of a reload per round; a taken entry becomes -2
    float val[TAL_REGS];
#pragma unroll
    for (int j = 0; j < TAL_REGS; j++) { const int ai = tid + j * TAL_T; val[j] = ai < a.A ? alr[ai] : -3.f; }
    // the selected anchors are marked AFTER the rounds (thread k keeps round k's pick): a global store inside a round sits in front of the
    // round's barrier, whose s_waitcnt vmcnt(0) then waits out the store's round trip -- ten times (3.7 us per round, round 4 ablation)
    int mine = -1;
    for (int k = 0; k < a.topk; k++) {
      float bv = -1.f;
      int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < TAL_REGS; j++) if (val[j] > bv) { bv = val[j]; bi = tid + j * TAL_T; }   // increasing index: strict '>' keeps the lowest among equals
      for (int m = 32; m >= 1; m >>= 1) {
        const float ov_ = __shfl_xor(bv, m);
        const int oi = __shfl_xor(bi, m);
        if (ov_ > bv || (ov_ == bv && oi < bi)) { bv = ov_; bi = oi; }
      }
      if ((tid & 63) == 0) { s_v[tid >> 6] = bv; s_i[tid >> 6] = bi; }
      __syncthreads();
      float fv = s_v[0]; int fi = s_i[0];
      for (int wv = 1; wv < TAL_T / 64; wv++)
        if (s_v[wv] > fv || (s_v[wv] == fv && s_i[wv] < fi)) { fv = s_v[wv]; fi = s_i[wv]; }
      if (fi < a.A) {
        if (tid == k && valid && (s_ingt[fi >> 5] & (1u << (fi & 31)))) mine = fi;     // mask_topk * mask_in_gts * mask_gt (Tal.cs:99)
        const int js = fi / TAL_T;
        if (fi - js * TAL_T == tid) {
#pragma unroll
          for (int j = 0; j < TAL_REGS; j++) val[j] = (j == js) ? -2.f : val[j];
        }
      }
      __syncthreads();                                            // s_v / s_i are rewritten by the next round
    }
    if (mine >= 0) mp[mine] = 1;
    return;
  }
  for (int k = 0; k < a.topk; k++) {
    float bv = -1.f;
    int bi = 0x7fffffff;
    // unconditional loads (a `continue` in front of the load made every iteration wait for its own L2 round trip: ~20 us per
    // round); a thread visits its anchors in increasing order, so a strict '>' keeps the lowest index among equals
#pragma unroll 8
    for (int ai = tid; ai < a.A; ai += TAL_T) {
      float v = alr[ai];
      v = (s_taken[ai >> 5] & (1u << (ai & 31))) ? -2.f : v;
      if (v > bv) { bv = v; bi = ai; }
    }
    for (int m = 32; m >= 1; m >>= 1) {
      const float ov_ = __shfl_xor(bv, m);
      const int oi = __shfl_xor(bi, m);
      if (ov_ > bv || (ov_ == bv && oi < bi)) { bv = ov_; bi = oi; }
    }
    if ((tid & 63) == 0) { s_v[tid >> 6] = bv; s_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      float fv = s_v[0]; int fi = s_i[0];
      for (int wv = 1; wv < TAL_T / 64; wv++)
        if (s_v[wv] > fv || (s_v[wv] == fv && s_i[wv] < fi)) { fv = s_v[wv]; fi = s_i[wv]; }
      s_sel = fi;
      if (fi < a.A) {
        s_taken[fi >> 5] |= 1u << (fi & 31);
        // mask_pos = mask_topk * mask_in_gts * mask_gt (Tal.cs:99); rows with mask_gt == 0 have their
        // top-k indices forced to 0 and then dropped by the count>1 rule (Tal.cs:155,165)
        if (valid && (s_ingt[fi >> 5] & (1u << (fi & 31)))) mp[fi] = 1;
      }
    }
    __syncthreads();
  }
}
// The (image, box) pairs of a batch are a small, data-dependent subset of the [B][gcap] workspace (labels resident on the device: the host does
// not know the counts).  A [gcap][B] grid launched 4096 workgroups for ~550 pairs at the headline batch, and the kernel's 96 us were the DISPATCH
// of the 3.5 thousand that exit at once (SQ_WAVE_CYCLES / SQ_WAVES: the waves that work live ~9 us).  Flat form: every workgroup scans the B
// counts (one wave, LDS), then walks the pairs p = blockIdx.x, + gridDim.x, ...; (image, slot) of pair p by binary search.
#define TAL_FLAT_MAXB 1024
template <class T, bool ROT>
__global__ void __launch_bounds__(TAL_T)
tal_metrics_kernel(LossArgs a, const int* __restrict__ gt_valid) {
  if (gridDim.y > 1) {                                            // [gcap][B] form (B > TAL_FLAT_MAXB)
    if ((int)blockIdx.x >= a.gt_count[blockIdx.y]) return;
    tal_metrics_pair<T, ROT>(a, gt_valid, (int)blockIdx.x, (int)blockIdx.y);
    return;
  }
  __shared__ int s_pref[TAL_FLAT_MAXB + 1];
  const int tid = threadIdx.x;
  for (int i = tid; i < a.B; i += TAL_T) s_pref[i + 1] = a.gt_count[i];
  __syncthreads();
  if (tid < 64) {                                                 // inclusive scan: lane l owns entries [16 l, 16 l + 16)
    int loc[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const int idx = tid * 16 + k; sum += idx < a.B ? s_pref[idx + 1] : 0; loc[k] = sum; }
    int incl = sum;
    for (int m = 1; m < 64; m <<= 1) { const int v = __shfl_up(incl, m); if (tid >= m) incl += v; }
    const int base = incl - sum;
#pragma unroll
    for (int k = 0; k < 16; k++) { const int idx = tid * 16 + k; if (idx < a.B) s_pref[idx + 1] = base + loc[k]; }
    if (tid == 0) s_pref[0] = 0;
  }
  __syncthreads();
  const int total = s_pref[a.B];
  for (int p = blockIdx.x; p < total; p += gridDim.x) {
    int lo = 0, hi = a.B;                                          // largest b with s_pref[b] <= p
    while (lo + 1 < hi) { const int mid = (lo + hi) >> 1; if (s_pref[mid] <= p) lo = mid; else hi = mid; }
    tal_metrics_pair<T, ROT>(a, gt_valid, p - s_pref[lo], lo);
    __syncthreads();                                              // the next pair re-initialises the bit sets the last top-k round still reads
  }
}

// ------------------------------------------------------------------ K3: select_highest_overlaps (Tal.cs:225-255)
__global__ void __launch_bounds__(LS_THREADS)
tal_resolve_kernel(LossArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)a.B * a.A) return;
  const int b = (int)(i / a.A), ai = (int)(i % a.A);
  const int n = a.gt_count[b];
  int cnt = 0, first = -1;
  for (int g = 0; g < n; g++) {
    if (a.mpos[((long)b * a.gcap + g) * a.A + ai]) { if (first < 0) first = g; cnt++; }
  }
  int sel = first;
  if (cnt > 1) {
    // anchors claimed by several GTs keep the GT with the largest overlap (argmax over ALL rows, first max)
    float bo = a.ov[((long)b * a.gcap) * a.A + ai];
    sel = 0;
    for (int g = 1; g < n; g++) {
      const float o = a.ov[((long)b * a.gcap + g) * a.A + ai];
      if (o > bo) { bo = o; sel = g; }
    }
    for (int g = 0; g < n; g++) a.mpos[((long)b * a.gcap + g) * a.A + ai] = (g == sel) ? 1 : 0;
  }
  a.fg_gt[i] = sel;  // -1 = background
  if (sel >= 0) {
    const long gi = (long)b * a.gcap + sel;
    // pos_align_metrics / pos_overlaps: amax over anchors of (metric * mask_pos) (Tal.cs:83-85);
    // non-negative floats order like their bit patterns
    atomicMax(&a.pos_align[gi], ys_f2u(a.align[gi * a.A + ai]));
    atomicMax(&a.pos_ov[gi], ys_f2u(a.ov[gi * a.A + ai]));
  }
}

// ------------------------------------------------------------------ K4: normalised target scores (Tal.cs:86-87)
__global__ void __launch_bounds__(LS_THREADS)
tal_targets_kernel(LossArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float nrm = 0.f;
  if (i < (long)a.B * a.A) {
    const int b = (int)(i / a.A), ai = (int)(i % a.A);
    const int g = a.fg_gt[i];
    if (g >= 0) {
      const long gi = (long)b * a.gcap + g;
      const float pa = ys_u2f(a.pos_align[gi]), po = ys_u2f(a.pos_ov[gi]);
      nrm = a.align[gi * a.A + ai] * po / (pa + 1e-9f);
    }
    a.tnorm[i] = nrm;
  }
  float z1 = 0.f, z2 = 0.f, z3 = 0.f;
  block_sum4(nrm, z1, z2, z3);
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    unsigned long long* acc = LOSS_ACC(a);
    unsigned long long* mine = acc + (size_t)(blockIdx.x & (LOSS_TSS_SHARDS - 1)) * 8;
    unsigned long long got = 0ull;
    if (!(fabsf(nrm) * LOSS_FIX_T < 9.0e18f)) got = atomicOr(mine + 5, 1ull);
    else {
#ifdef YS_EMU_BUILD
      const long long q = (long long)llrintf(nrm * LOSS_FIX_T);
#else
      const long long q = __float2ll_rn(nrm * LOSS_FIX_T);
#endif
      got = atomicAdd(mine, (unsigned long long)q);           // returning form, also for q = 0: the ticket below must come after it
    }
    unsigned one = 1u;
#ifndef YS_EMU_BUILD
    asm volatile("" : "+v"(one) : "v"(got));                   // the ticket is not issued before the add has returned
#else
    (void)got;
#endif
    // two-level ticket: 2100 workgroups drawing from ONE counter serialise in the L2 like the unsharded sums did (tal_targets_kernel 12.6 -> 36 us); the last
    // workgroup of a shard (its shard's total is complete: every add to it returned before its owner's ticket) draws from the second counter
    const int sh = (int)(blockIdx.x & (LOSS_TSS_SHARDS - 1));
    const unsigned n_sh = ((unsigned)gridDim.x - (unsigned)sh + (LOSS_TSS_SHARDS - 1)) / LOSS_TSS_SHARDS;
    const unsigned n_act = gridDim.x < LOSS_TSS_SHARDS ? gridDim.x : LOSS_TSS_SHARDS;
    const unsigned t1 = atomicAdd((unsigned*)(mine + 6), one);
    int last = 0;
    if (t1 == n_sh - 1) last = atomicAdd((unsigned*)(acc + 7), 1u) == n_act - 1 ? 1 : 0;
    s_last = last;
  }
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    unsigned long long* acc = LOSS_ACC(a);
    long long t = 0; unsigned long long bad = 0ull;
    for (int sh = 0; sh < LOSS_TSS_SHARDS; sh++) { t += (long long)atomicAdd(acc + sh * 8, 0ull); bad |= atomicOr(acc + sh * 8 + 5, 0ull); }
    a.scalars[0] = bad ? __builtin_nanf("") : loss_tss_of((float)((double)t * (1.0 / (double)LOSS_FIX_T)));
  }
}

// ------------------------------------------------------------------ K5: BCE cls loss + gradient (Loss.cs:447)
template <class T>
__global__ void __launch_bounds__(LS_THREADS)
loss_cls_kernel(LossArgs a) {
  constexpr int EPL = Elem<T>::EPL;
  const int vpr = a.ld_ps / EPL;  // 16-byte vectors per anchor row (row padded to EPL)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float lsum = 0.f;
  const long total = (long)a.B * a.A * vpr;
  const bool inb = i < total;
  long row; int c0, b;
  {
    const long ii = inb ? i : 0;
    if (total < (1L << 31)) {    // 32-bit index arithmetic (a 64-bit division is ~80 VALU instructions)
      const unsigned iu = (unsigned)ii, r = iu / (unsigned)vpr;
      row = r; c0 = (int)(iu - r * (unsigned)vpr) * EPL; b = (int)(r / (unsigned)a.A);
    } else {
      row = ii / vpr; c0 = (int)(ii - row * vpr) * EPL; b = (int)(row / a.A);
    }
  }
  const uint4 xv16 = ys_ld16((const T*)a.ps + row * a.ld_ps + c0);
  const int g = a.fg_gt[row];
  const float tss_w = a.scalars[0];            // written by tal_targets_kernel's last workgroup
  if (inb) {
    int tc = -1;
    float tv = 0.f;
    if (g >= 0) {
      tc = a.gt_cls[(long)b * a.gcap + g];
      tc = tc < 0 ? 0 : tc;  // target_labels.clamp_(0) (Tal.cs:183)
      tv = a.tnorm[row];
    }
    float x[EPL], gr[EPL];
    const float gs = a.hyp_cls * (float)a.B / tss_w;
    ys_unpack<T>(xv16, x);
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      const int c = c0 + e;
      if (c < a.nc) {
        const float t = (c == tc) ? tv : 0.f;
        const float xv = x[e];
        // BCEWithLogits (reduction none) and its derivative from ONE exponential: e = exp(-|x|), sigmoid = 1/(1+e) or e/(1+e).
        // log(1 + e) through the hardware log: absolute error <= 1 ulp of (1 + e) ~ 6e-8 per element, far inside the 1e-3 budget
        // of the summed loss (log1pf's software path was ~40 instructions of a VALU-bound kernel)
        const float ex = __expf(-fabsf(xv));
        const float r = ys_rcp(1.0f + ex);
        lsum += fmaxf(xv, 0.f) - xv * t + __logf(1.0f + ex);
        gr[e] = ((xv >= 0.f ? r : ex * r) - t) * gs;
      } else {
        gr[e] = 0.f;
      }
    }
    ys_st16((T*)a.dps + row * a.ld_ps + c0, ys_pack<T>(gr));
  }
  float z0 = 0.f, z2 = 0.f, z3 = 0.f;
  block_sum4(z0, lsum, z2, z3);
  if (threadIdx.x == 0) loss_acc_add(LOSS_ACC(a), 1, lsum, LOSS_FIX_L);
}

// ------------------------------------------------------------------ K6: CIoU + DFL loss + gradient (Loss.cs:134-166)
// four consecutive lanes own one anchor (one side l,t,r,b each)
// RR = reg_max when it is the usual 16 (compile-time: the 16 bins stay in registers, rows move as 16-byte vectors), 0 = run-time.
// Only foreground anchors (a few per cent) need the softmax at all; every other row just receives a zero gradient.
template <class T, int RR, bool ROT>
__global__ void __launch_bounds__(LS_THREADS)
loss_box_kernel(LossArgs a) {
  constexpr int EPL = Elem<T>::EPL;
  constexpr int RM = RR ? RR : 32;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)a.B * a.A * 4;
  const bool inb = i < total;
  const long row = inb ? i >> 2 : 0;
  const int s = (int)(i & 3);
  const int lane = threadIdx.x & 63;
  const int R = RR ? RR : a.reg_max;
  const int g = inb ? a.fg_gt[row] : -1;
  float p[RM], xl[RM], gl[RM];
  float dist = 0.f, lse = 0.f;
  const T* lrow = (const T*)a.pd + row * a.ld_pd + s * R;
#pragma unroll
  for (int j = 0; j < RM; j++) gl[j] = 0.f;
  if (g >= 0) {
    if (RR) {
#pragma unroll
      for (int v = 0; v < RM / EPL; v++) ys_unpack<T>(ys_ld16(lrow + v * EPL), xl + v * EPL);
    } else {
      for (int j = 0; j < R; j++) xl[j] = Elem<T>::to_f(lrow[j]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < RM; j++) if (j < R) mx = fmaxf(mx, xl[j]);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < RM; j++) if (j < R) { p[j] = __expf(xl[j] - mx); se += p[j]; }
    const float inv = 1.0f / se;
#pragma unroll
    for (int j = 0; j < RM; j++) if (j < R) { p[j] *= inv; dist += p[j] * (float)j; }
    lse = mx + __logf(se);
  }
  // the four side distances of this anchor (all four lanes of an anchor share g, so they take the branch above together)
  const int base = lane & ~3;
  const float d0 = __shfl(dist, base + 0), d1 = __shfl(dist, base + 1), d2 = __shfl(dist, base + 2), d3 = __shfl(dist, base + 3);
  float l_iou = 0.f, l_dfl = 0.f, l_ang = 0.f;
  const float tss_blk = a.scalars[0];          // written by tal_targets_kernel's last workgroup
  if (g >= 0) {
    const int b = (int)(row / a.A), ai = (int)(row - (long)b * a.A);
    const AnchorInfo an = anchor_of(a, ai);
    const float w = a.tnorm[row];                      // weight = target_scores.sum(-1) (Loss.cs:138)
    const float tss = tss_blk;
    const float gbox = a.hyp_box * (float)a.B / tss * w;
    float gd, t;                                       // d(total)/d(dist_s) through the box term; DFL target of this side
    if (ROT) {
      // RotatedBboxLoss (Loss.cs:197-227) + calculate_angle_loss (:657-676) on pred = (dist2rbox(dist, angle), angle), grid units
      const float* gb = a.gt_box + ((long)b * a.gcap + g) * 5;
      const float tb[5] = {gb[0] / an.stride, gb[1] / an.stride, gb[2] / an.stride, gb[3] / an.stride, gb[4]};   // Loss.cs:596
      const float PI_F = 3.14159265358979323846f;
      const float sg = ys_sigmoid(Elem<T>::to_f(((const T*)a.pa)[row * a.ld_pa]));
      const float ang = (sg - 0.25f) * PI_F;
      const float cs = cosf(ang), sn = sinf(ang);
      const float xf = (d2 - d0) / 2.0f, yf = (d3 - d1) / 2.0f;
      const Dual5 p5[5] = {dvarN<5>(xf * cs - yf * sn + an.ax, 0), dvarN<5>(xf * sn + yf * cs + an.ay, 1), dvarN<5>(d0 + d2, 2),
                           dvarN<5>(d1 + d3, 3), dvarN<5>(ang, 4)};
      const Dual5 t5[5] = {dconstN<5>(tb[0]), dconstN<5>(tb[1]), dconstN<5>(tb[2]), dconstN<5>(tb[3]), dconstN<5>(tb[4])};
      const Dual5 pi = probiou_t<Dual5>(p5, t5);       // Loss.cs:203 (obb1 = pred, obb2 = target)
      if (s == 0) l_iou = (1.0f - pi.v) * w;
      const float Gx = -gbox * pi.g[0], Gy = -gbox * pi.g[1], Gw = -gbox * pi.g[2], Gh = -gbox * pi.g[3], Gt = -gbox * pi.g[4];
      // x = xf cos - yf sin + ax, y = xf sin + yf cos + ay, w = l + r, h = t + b with xf = (r - l)/2, yf = (b - t)/2
      gd = (s == 0) ? (Gx * (-cs / 2) + Gy * (-sn / 2) + Gw) : (s == 1) ? (Gx * (sn / 2) + Gy * (-cs / 2) + Gh)
         : (s == 2) ? (Gx * (cs / 2) + Gy * (sn / 2) + Gw) : (Gx * (-sn / 2) + Gy * (cs / 2) + Gh);
      const float lar = logf((tb[2] + 1e-9f) / (tb[3] + 1e-9f));
      const float swt = expf(-(lar * lar) / 9.0f);     // lambda_val = 3
      const float dlt = ang - tb[4];
      const float wrp = dlt - rintf(dlt / PI_F) * PI_F;
      const float s2 = sinf(2.0f * wrp);
      if (s == 0) {
        l_ang = swt * s2 * s2 * w;
        const float gang = a.hyp_angle * (float)a.B / tss * w * swt * 2.0f * sinf(4.0f * wrp);
        const float dth = Gx * (-xf * sn - yf * cs) + Gy * (xf * cs - yf * sn) + Gt + gang;
        ((T*)a.dpa)[row * a.ld_pa] = Elem<T>::from_f(dth * PI_F * sg * (1.0f - sg));   // through (sigmoid - 0.25) * pi (Head.cs:429)
      }
      // rbox2dist (Tal.cs:418-453) target distances
      const float ox = tb[0] - an.ax, oy = tb[1] - an.ay, ct = cosf(tb[4]), stn = sinf(tb[4]);
      const float xft = ox * ct + oy * stn, yft = -ox * stn + oy * ct;
      t = (s == 0) ? (tb[2] / 2 - xft) : (s == 1) ? (tb[3] / 2 - yft) : (s == 2) ? (tb[2] / 2 + xft) : (tb[3] / 2 + yft);
    } else {
      const float* gb = a.gt_box + ((long)b * a.gcap + g) * 4;
      const float tb[4] = {gb[0] / an.stride, gb[1] / an.stride, gb[2] / an.stride, gb[3] / an.stride};  // Loss.cs:456
      // CIoU(pred, target) with d/d(pred x1,y1,x2,y2)
      Dual4 b1[4] = {dvar(an.ax - d0, 0), dvar(an.ay - d1, 1), dvar(an.ax + d2, 2), dvar(an.ay + d3, 3)};
      Dual4 b2[4] = {dconst(tb[0]), dconst(tb[1]), dconst(tb[2]), dconst(tb[3])};
      const Dual4 ci = ciou_xyxy<Dual4>(b1, b2);
      if (s == 0) l_iou = (1.0f - ci.v) * w;
      // d(total)/d(dist_s): loss_box*B = hyp_box*B/tss * sum((1-ciou)*w); x1 = ax - l, y1 = ay - t, x2 = ax + r, y2 = ay + b
      const float cg = (s == 0) ? ci.g[0] : (s == 1) ? ci.g[1] : (s == 2) ? ci.g[2] : ci.g[3];
      gd = (s < 2) ? (gbox * cg) : (-gbox * cg);   // -(dciou/dx1)*(-1) = +g ; -(dciou/dx2)*(+1) = -g
      // DFL (Loss.cs:104-118, Tal.cs:365-379): target ltrb clamped to [0, reg_max-1-0.01]
      t = (s == 0) ? (an.ax - tb[0]) : (s == 1) ? (an.ay - tb[1]) : (s == 2) ? (tb[2] - an.ax) : (tb[3] - an.ay);
    }
    const float tmax = (float)(R - 1) - 0.01f;
    t = fminf(fmaxf(t, 0.f), tmax);
    const int tl = (int)t;
    const int tr = tl + 1;
    const float wl = (float)tr - t, wr = 1.0f - wl;
    float x_l = 0.f, x_r = 0.f;                         // xl[tl], xl[tr] without dynamic register indexing
#pragma unroll
    for (int j = 0; j < RM; j++) { x_l = (j == tl) ? xl[j] : x_l; x_r = (j == tr) ? xl[j] : x_r; }
    const float ce_l = lse - x_l, ce_r = lse - x_r;
    l_dfl = (ce_l * wl + ce_r * wr) * 0.25f * w;        // mean over the 4 sides, x weight
    const float gdfl = a.hyp_dfl * (float)a.B / tss * w * 0.25f;
#pragma unroll
    for (int j = 0; j < RM; j++) {
      if (j < R) {
        float gj = gd * p[j] * ((float)j - dist);          // through softmax expectation
        gj += gdfl * (p[j] - (j == tl ? wl : 0.f) - (j == tr ? wr : 0.f));
        gl[j] = gj;
      }
    }
  }
  if (inb) {
    T* drow = (T*)a.dpd + row * a.ld_pd + s * R;
    if (RR) {
#pragma unroll
      for (int v = 0; v < RM / EPL; v++) ys_st16(drow + v * EPL, ys_pack<T>(gl + v * EPL));
    } else {
      for (int j = 0; j < R; j++) drow[j] = Elem<T>::from_f(gl[j]);
    }
  }
  float z1 = 0.f;
  block_sum4(l_ang, z1, l_iou, l_dfl);
  if (threadIdx.x == 0) {
    unsigned long long* acc = LOSS_ACC(a);
    loss_acc_add(acc, 2, l_iou, LOSS_FIX_L); loss_acc_add(acc, 3, l_dfl, LOSS_FIX_L);
    if (ROT) loss_acc_add(acc, 4, l_ang, LOSS_FIX_L);
  }
}

// ------------------------------------------------------------------ K7: items (Loss.cs:463-476)
__global__ void loss_items_kernel(LossArgs a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float* sc = a.scalars;
    const unsigned long long* acc = LOSS_ACC(a);
    const float tss = sc[0];                   // folded by tal_targets_kernel's last workgroup: the value loss_cls / loss_box scaled their gradients with
    sc[5] = loss_acc_get(acc, 1, LOSS_FIX_L); sc[6] = loss_acc_get(acc, 2, LOSS_FIX_L); sc[7] = loss_acc_get(acc, 3, LOSS_FIX_L);
    if (a.rot) sc[12] = loss_acc_get(acc, 4, LOSS_FIX_L);
    const float l_cls = sc[5] / tss, l_iou = sc[6] / tss, l_dfl = sc[7] / tss;
    sc[1] = l_iou * a.hyp_box;
    sc[2] = l_cls * a.hyp_cls;
    sc[3] = l_dfl * a.hyp_dfl;
    sc[4] = (sc[1] + sc[2] + sc[3]) * (float)a.B;
    if (a.rot) { sc[13] = sc[12] / tss * a.hyp_angle; sc[4] += sc[13] * (float)a.B; }   // Loss.cs:604,617
  }
}

static int cls_blocks(const LossArgs& a, int epl) { return ys_cdiv((long)a.B * a.A * (a.ld_ps / epl), LS_THREADS); }
static int box_blocks(const LossArgs& a) { return ys_cdiv((long)a.B * a.A * 4, LS_THREADS); }
static int anc_blocks(const LossArgs& a) { return ys_cdiv((long)a.B * a.A, LS_THREADS); }

size_t ys_loss_partial_floats(int B, int A) {
  // three partial regions: targets (B*A/256 blocks), cls (<= B*A*ld/4/256), box (B*A*4/256); sized generously
  const size_t anc = ((size_t)B * A + LS_THREADS - 1) / LS_THREADS;
  return 4 * (anc + anc * 64 + anc * 4) + 64;
}

template <class T>
static int loss_launch_t(hipStream_t st, const LossArgs& a) {
  constexpr int EPL = Elem<T>::EPL;
  if (a.reg_max > 32) { ys_set_error("loss: reg_max %d > 32 unsupported", a.reg_max); return YS_ERR_UNSUPPORTED; }
  if (a.A > 1056 * 32) { ys_set_error("loss: %d anchors exceed the assigner capacity", a.A); return YS_ERR_UNSUPPORTED; }
  if (a.ld_ps % EPL) { ys_set_error("loss: ld_ps %d must be a multiple of %d", a.ld_ps, EPL); return YS_ERR_INVALID_ARG; }
  // gt_valid lives behind gt_cls ([B][gcap] ints each)
  int* gt_valid = a.gt_cls + (long)a.B * a.gcap;
  const int nb_a = anc_blocks(a), nb_c = cls_blocks(a, EPL), nb_b = box_blocks(a);
  if (a.rot) {
    if (!a.pa || !a.dpa) { ys_set_error("loss: the OBB criterion needs the angle logits"); return YS_ERR_INVALID_ARG; }
    YS_CHECK_HIP(hipMemsetAsync(a.dpa, 0, (size_t)a.B * a.A * a.ld_pa * sizeof(T), st));   // background anchors: no angle gradient
  }
  // seven launches (rounds 1-5: eleven): [bbox_decode + GT padding] -> metrics / top-k -> resolve -> targets -> cls -> box -> items
  if (a.reg_max == 16) YS_LAUNCH((loss_decode_kernel<T, 16>), nb_b + 1, LS_THREADS, st, a, gt_valid);
  else YS_LAUNCH((loss_decode_kernel<T, 0>), nb_b + 1, LS_THREADS, st, a, gt_valid);
  // flat grid: enough workgroups for ~12 boxes per image in one trip, never more than the host-known pair bound
  const long pair_cap = (long)(a.gmax > 0 ? a.gmax : a.gcap) * a.B;
  long flat = (long)a.B * 12 > 256 ? (long)a.B * 12 : 256;
  if (flat > pair_cap) flat = pair_cap;
  const dim3 tgrid = a.B <= TAL_FLAT_MAXB ? dim3((unsigned)flat) : dim3(a.gmax > 0 ? a.gmax : a.gcap, a.B);
  if (a.rot) YS_LAUNCH((tal_metrics_kernel<T, true>), tgrid, TAL_T, st, a, (const int*)gt_valid);
  else YS_LAUNCH((tal_metrics_kernel<T, false>), tgrid, TAL_T, st, a, (const int*)gt_valid);
  YS_LAUNCH(tal_resolve_kernel, nb_a, LS_THREADS, st, a);
  YS_LAUNCH(tal_targets_kernel, nb_a, LS_THREADS, st, a);
  YS_LAUNCH((loss_cls_kernel<T>), nb_c, LS_THREADS, st, a);
  if (a.rot) {
    if (a.reg_max == 16) YS_LAUNCH((loss_box_kernel<T, 16, true>), nb_b, LS_THREADS, st, a);
    else YS_LAUNCH((loss_box_kernel<T, 0, true>), nb_b, LS_THREADS, st, a);
  } else {
    if (a.reg_max == 16) YS_LAUNCH((loss_box_kernel<T, 16, false>), nb_b, LS_THREADS, st, a);
    else YS_LAUNCH((loss_box_kernel<T, 0, false>), nb_b, LS_THREADS, st, a);
  }
  YS_LAUNCH(loss_items_kernel, 1, 64, st, a);
  return YS_OK;
}

int ys_loss_detect_launch(hipStream_t st, int dtype, const LossArgs& a) {
  if (dtype == YS_BF16) return loss_launch_t<bf16_t>(st, a);
  return loss_launch_t<float>(st, a);
}
