// segloss.hip -- v8SegmentationLoss mask term (Utils/Loss.cs:711-863) and Ops.process_mask (Utils/Ops.cs:409-489).
//
//   for every foreground anchor a of image i (assigned GT g):           Loss.cs:838-852
//     gt_mask   = (masks[i] == g + 1)                                    overlap-encoded instance ids (YoloDataset.cs:265-267)
//     pred_mask = coeff[a, 0:nm] . proto[i, 0:nm, :, :]                  einsum "in,nhw->ihw" (:826)
//     l_a       = mean_hw( crop(BCEWithLogits(pred_mask, gt_mask), box_a) ) / area_a    (:827-828)
//   seg = sum_a l_a / sum(fg_mask) * hyp_box                             (:861, :777)
// crop_mask is the float form of Ops.cs:437-447 (x1 <= col < x2, y1 <= row < y2); the reference's CPU-only "< 50 rows"
// integer-truncation branch (Ops.cs:421-435) is selectable with `trunc_crop` for CPU-parity runs.
// Gradients w.r.t. the mask coefficients and the prototypes are produced analytically (two deterministic passes,
// fixed-order reductions, no atomics).  The detection part (box/cls/dfl + assignment) is loss.hip and runs first.
#include "ys_internal.h"
#include "ys_kernels.h"

#define SG_THREADS 256
#define SG_NM_MAX 64

// ---------------------------------------------------------------- foreground list (ordered by image, then anchor)
__global__ void __launch_bounds__(SG_THREADS)
seg_count_kernel(const int* __restrict__ fg_gt, int A, int* __restrict__ cnt) {
  __shared__ int s[SG_THREADS];
  const int b = blockIdx.x, tid = threadIdx.x;
  int c = 0;
  for (int a = tid; a < A; a += SG_THREADS) c += fg_gt[(long)b * A + a] >= 0 ? 1 : 0;
  s[tid] = c;
  __syncthreads();
  for (int st = SG_THREADS / 2; st > 0; st >>= 1) { if (tid < st) s[tid] += s[tid + st]; __syncthreads(); }
  if (tid == 0) cnt[b] = s[0];
}

__global__ void seg_offsets_kernel(const int* __restrict__ cnt, int B, int* __restrict__ off /*[B+1]*/) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int o = 0;
    for (int b = 0; b < B; b++) { off[b] = o; o += cnt[b]; }
    off[B] = o;
  }
}

__global__ void __launch_bounds__(SG_THREADS)
seg_compact_kernel(const int* __restrict__ fg_gt, int A, const int* __restrict__ off, int* __restrict__ list) {
  __shared__ int s[SG_THREADS + 1];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int chunk = (A + SG_THREADS - 1) / SG_THREADS;
  const int a0 = tid * chunk, a1 = (a0 + chunk) < A ? (a0 + chunk) : A;
  int c = 0;
  for (int a = a0; a < a1; a++) c += fg_gt[(long)b * A + a] >= 0 ? 1 : 0;
  s[tid + 1] = c;
  if (tid == 0) s[0] = 0;
  __syncthreads();
  if (tid == 0) for (int i = 1; i <= SG_THREADS; i++) s[i] += s[i - 1];
  __syncthreads();
  int o = off[b] + s[tid];
  for (int a = a0; a < a1; a++) if (fg_gt[(long)b * A + a] >= 0) list[o++] = a;
}

int ys_fg_list_launch(hipStream_t st, const int* fg_gt, int B, int A, int* cnt, int* off, int* list) {
  YS_LAUNCH(seg_count_kernel, B, SG_THREADS, st, fg_gt, A, cnt);
  YS_LAUNCH(seg_offsets_kernel, 1, 64, st, (const int*)cnt, B, off);
  YS_LAUNCH(seg_compact_kernel, B, SG_THREADS, st, fg_gt, A, (const int*)off, list);
  return YS_OK;
}

struct SegArgs {
  const void* mc;     // mask coefficients [B][A][ld_mc]
  const void* proto;  // prototypes [B][mh*mw][ld_pr]
  void* dmc; void* dproto;
  const float* masks; // [B][mh][mw] overlap-encoded instance ids
  const int* fg_gt; const float* gt_box; const int* cnt; const int* off; const int* list;
  float* ent;         // per list entry: x1,y1,x2,y2 (mask units), scale (1/(mh*mw*area)), id, 0, 0
  float* part;        // per entry loss partial
  float* scalars;
  int B, A, nm, ld_mc, ld_pr, mh, mw, gcap, H, W, trunc_crop;
  float hyp_box;
};

__device__ inline void seg_crop_bounds(const float* e, int mw, int mh, int trunc, int n_rows, int& c0, int& c1, int& r0, int& r1) {
  if (trunc && n_rows < 50) {          // Ops.cs:421-435 (C# ToInt32 truncation toward zero)
    c0 = (int)e[0]; r0 = (int)e[1]; c1 = (int)e[2]; r1 = (int)e[3];
  } else {                             // Ops.cs:437-447: x1 <= col < x2
    c0 = (int)ceilf(e[0]); c1 = (int)ceilf(e[2]); r0 = (int)ceilf(e[1]); r1 = (int)ceilf(e[3]);
  }
  c0 = c0 < 0 ? 0 : c0; r0 = r0 < 0 ? 0 : r0;
  c1 = c1 > mw ? mw : c1; r1 = r1 > mh ? mh : r1;
}

// NM prototypes of one pixel -> floats.  NMC = the compile-time count (32: Proto's nm, Head.cs:238; vector loads, registers) or 0 =
// run-time a.nm (then the per-thread arrays are indexed by a run-time loop and live in scratch: the round-1 form, kept for other nm)
template <class T, int NMC>
__device__ inline void seg_load_row(const T* pr, int nm, float* pv) {
  if (NMC > 0) {
    constexpr int EPL = Elem<T>::EPL;
#pragma unroll
    for (int k = 0; k < (NMC > 0 ? NMC : 1); k += EPL) ys_unpack<T>(ys_ld16(pr + k), pv + k);
  } else {
    for (int k = 0; k < nm; k++) pv[k] = Elem<T>::to_f(pr[k]);
  }
}

// pass 1: one workgroup per foreground anchor: loss term + d(coeff)
template <class T, int NMC>
__global__ void __launch_bounds__(SG_THREADS)
seg_anchor_kernel(SegArgs a) {
  const int nm = NMC > 0 ? NMC : a.nm;
  __shared__ float sco[SG_NM_MAX];
  __shared__ float sred[SG_THREADS / 64][SG_NM_MAX + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntot = a.off[a.B];
  for (int e = blockIdx.x; e < ntot; e += gridDim.x) {
    // image of this entry
    int b = 0;
    while (b + 1 < a.B && a.off[b + 1] <= e) b++;
    const int an = a.list[e];
    const long row = (long)b * a.A + an;
    const int g = a.fg_gt[row];
    const float* gb = a.gt_box + ((long)b * a.gcap + g) * 4;
    // Loss.cs:830-836: normalise by imgsz (W,H,W,H), area of the normalised box, scale to mask size
    const float nx1 = gb[0] / (float)a.W, ny1 = gb[1] / (float)a.H, nx2 = gb[2] / (float)a.W, ny2 = gb[3] / (float)a.H;
    const float area = (nx2 - nx1) * (ny2 - ny1);
    float* en = a.ent + (long)e * 8;
    const float scale = 1.0f / ((float)(a.mh * a.mw) * area);
    if (tid == 0) {
      en[0] = nx1 * (float)a.mw; en[1] = ny1 * (float)a.mh; en[2] = nx2 * (float)a.mw; en[3] = ny2 * (float)a.mh;
      en[4] = scale; en[5] = (float)(g + 1);
    }
    if (tid < nm) sco[tid] = Elem<T>::to_f(((const T*)a.mc)[row * a.ld_mc + tid]);
    __syncthreads();
    int c0, c1, r0, r1;
    seg_crop_bounds(en, a.mw, a.mh, a.trunc_crop, a.cnt[b], c0, c1, r0, r1);
    const int bw = c1 - c0, bh = r1 - r0;
    const int npx = bw > 0 && bh > 0 ? bw * bh : 0;
    float lsum = 0.f;
    float dco[SG_NM_MAX];
#pragma unroll
    for (int k = 0; k < (NMC > 0 ? NMC : SG_NM_MAX); k++) if (k < nm) dco[k] = 0.f;
    const float gid = (float)(g + 1);
    for (int i = tid; i < npx; i += SG_THREADS) {
      const int rr = r0 + i / bw, cc = c0 + i % bw;
      const long p = (long)b * a.mh * a.mw + (long)rr * a.mw + cc;
      const T* pr = (const T*)a.proto + p * a.ld_pr;
      float pv[SG_NM_MAX];
      seg_load_row<T, NMC>(pr, nm, pv);
      float x = 0.f;
#pragma unroll
      for (int k = 0; k < (NMC > 0 ? NMC : SG_NM_MAX); k++) if (k < nm) x += sco[k] * pv[k];
      const float t = a.masks[p] == gid ? 1.0f : 0.0f;
      lsum += fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
      const float d = ys_sigmoid(x) - t;
#pragma unroll
      for (int k = 0; k < (NMC > 0 ? NMC : SG_NM_MAX); k++) if (k < nm) dco[k] += d * pv[k];
    }
    // workgroup reduction of (loss, dcoeff[nm]) in a fixed order
    lsum = ys_wave_sum(lsum);
#pragma unroll
    for (int k = 0; k < (NMC > 0 ? NMC : SG_NM_MAX); k++) if (k < nm) dco[k] = ys_wave_sum(dco[k]);
    if (lane == 0) {
      sred[wave][nm] = lsum;
#pragma unroll
      for (int k = 0; k < (NMC > 0 ? NMC : SG_NM_MAX); k++) if (k < nm) sred[wave][k] = dco[k];
    }
    __syncthreads();
    const float gsc = a.hyp_box * (float)a.B / (float)ntot * scale;
    if (tid <= nm) {
      float s = 0.f;
      for (int w = 0; w < SG_THREADS / 64; w++) s += sred[w][tid];
      if (tid == nm) a.part[e] = s * scale;
      else ((T*)a.dmc)[row * a.ld_mc + tid] = Elem<T>::from_f(s * gsc);
    }
    __syncthreads();
  }
}

// pass 2: one thread per prototype pixel: d(proto)[p][k] = sum over the image's foreground anchors whose crop holds p, in list
// order (fixed summation order).  The image's entries (crop bounds, gradient scale, instance id, coefficient vector) are staged
// through LDS in chunks by the whole workgroup and read back as broadcasts; an entry whose crop rows miss the workgroup's pixel
// rows is skipped by all threads at once.  (Round-1 form: every thread re-read every entry from global memory and kept its
// nm-vectors in scratch: 1.72 ms on BASELINE config 4.)
#define SG_ECH 32                       // entries per LDS chunk
template <class T, int NMC>
__global__ void __launch_bounds__(SG_THREADS)
seg_proto_grad_kernel(SegArgs a) {
  const int nm = NMC > 0 ? NMC : a.nm;
  __shared__ int sbnd[SG_ECH][4];       // c0, c1, r0, r1
  __shared__ float sval[SG_ECH][2];     // gradient scale, instance id
  __shared__ float scf[SG_ECH][SG_NM_MAX];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int npix = a.mh * a.mw;
  const int p0 = blockIdx.x * SG_THREADS;
  const int p = p0 + tid;
  const bool live = p < npix;
  const int pc = live ? p : npix - 1;
  const int rr = pc / a.mw, cc = pc - rr * a.mw;
  const int wr0 = p0 / a.mw, wr1 = ((p0 + SG_THREADS - 1 < npix ? p0 + SG_THREADS - 1 : npix - 1)) / a.mw;   // pixel rows of this workgroup
  const long gp = (long)b * npix + pc;
  float pv[SG_NM_MAX], acc[SG_NM_MAX];
  seg_load_row<T, NMC>((const T*)a.proto + gp * a.ld_pr, nm, pv);
#pragma unroll
  for (int k = 0; k < (NMC > 0 ? NMC : SG_NM_MAX); k++) if (k < nm) acc[k] = 0.f;
  const int ntot = a.off[a.B];
  const int e0 = a.off[b], e1 = a.off[b + 1];
  const float mval = a.masks[gp];
  for (int eb = e0; eb < e1; eb += SG_ECH) {
    const int ne = (e1 - eb) < SG_ECH ? (e1 - eb) : SG_ECH;
    __syncthreads();
    if (tid < ne) {
      const float* en = a.ent + (long)(eb + tid) * 8;
      int c0, c1, r0, r1;
      seg_crop_bounds(en, a.mw, a.mh, a.trunc_crop, e1 - e0, c0, c1, r0, r1);
      sbnd[tid][0] = c0; sbnd[tid][1] = c1; sbnd[tid][2] = r0; sbnd[tid][3] = r1;
      sval[tid][0] = en[4]; sval[tid][1] = en[5];
    }
    for (int i = tid; i < ne * nm; i += SG_THREADS) {
      const int el = i / nm, k = i - el * nm;
      scf[el][k] = Elem<T>::to_f(((const T*)a.mc)[((long)b * a.A + a.list[eb + el]) * a.ld_mc + k]);
    }
    __syncthreads();
    for (int el = 0; el < ne; el++) {
      const int r0 = sbnd[el][2], r1 = sbnd[el][3];
      if (r1 <= wr0 || r0 > wr1) continue;                               // uniform: the crop misses this workgroup's rows
      if (cc < sbnd[el][0] || cc >= sbnd[el][1] || rr < r0 || rr >= r1) continue;
      float x = 0.f;
#pragma unroll
      for (int k = 0; k < (NMC > 0 ? NMC : SG_NM_MAX); k++) if (k < nm) x += scf[el][k] * pv[k];
      const float t = mval == sval[el][1] ? 1.0f : 0.0f;
      const float d = (ys_sigmoid(x) - t) * sval[el][0] * a.hyp_box * (float)a.B / (float)ntot;   // same association as the round-1 kernel
#pragma unroll
      for (int k = 0; k < (NMC > 0 ? NMC : SG_NM_MAX); k++) if (k < nm) acc[k] += d * scf[el][k];
    }
  }
  if (live) {
    T* dp = (T*)a.dproto + gp * a.ld_pr;
    if (NMC > 0) {
      constexpr int EPL = Elem<T>::EPL;
#pragma unroll
      for (int k = 0; k < (NMC > 0 ? NMC : 1); k += EPL) ys_st16(dp + k, ys_pack<T>(acc + k));
    } else {
      for (int k = 0; k < nm; k++) dp[k] = Elem<T>::from_f(acc[k]);
    }
  }
}

__global__ void __launch_bounds__(SG_THREADS)
seg_finalize_kernel(SegArgs a) {
  __shared__ double sbuf[SG_THREADS];
  const int tid = threadIdx.x;
  const int ntot = a.off[a.B];
  double s = 0.0;
  for (int e = tid; e < ntot; e += SG_THREADS) s += (double)a.part[e];
  sbuf[tid] = s;
  __syncthreads();
  for (int st = SG_THREADS / 2; st > 0; st >>= 1) { if (tid < st) sbuf[tid] += sbuf[tid + st]; __syncthreads(); }
  if (tid == 0) {
    const float seg = ntot > 0 ? (float)(sbuf[0] / (double)ntot) * a.hyp_box : 0.f;   // Loss.cs:861,777
    a.scalars[8] = seg;
    a.scalars[9] = (float)ntot;
    a.scalars[4] += seg * (float)a.B;     // total = sum(items) * B (Loss.cs:778)
  }
}

int ys_loss_segment_launch(hipStream_t st, int dtype, const void* mc, void* dmc, int ld_mc, const void* proto, void* dproto, int ld_pr,
                           const float* masks, const int* fg_gt, const float* gt_box, int* cnt, int* off, int* list, float* ent,
                           float* part, float* scalars, int B, int A, int nm, int mh, int mw, int gcap, int H, int W, int trunc_crop) {
  if (nm > SG_NM_MAX) { ys_set_error("segment loss: nm=%d > %d", nm, SG_NM_MAX); return YS_ERR_UNSUPPORTED; }
  SegArgs a{};
  a.mc = mc; a.dmc = dmc; a.proto = proto; a.dproto = dproto; a.masks = masks; a.fg_gt = fg_gt; a.gt_box = gt_box;
  a.cnt = cnt; a.off = off; a.list = list; a.ent = ent; a.part = part; a.scalars = scalars;
  a.B = B; a.A = A; a.nm = nm; a.ld_mc = ld_mc; a.ld_pr = ld_pr; a.mh = mh; a.mw = mw; a.gcap = gcap; a.H = H; a.W = W;
  a.trunc_crop = trunc_crop; a.hyp_box = 7.5f;
  const size_t es = dtype == YS_BF16 ? 2 : 4;
  YS_CHECK_HIP(hipMemsetAsync(dmc, 0, (size_t)B * A * ld_mc * es, st));   // background anchors get no mask gradient
  YS_TRY(ys_fg_list_launch(st, fg_gt, B, A, cnt, off, list));
  const int g1 = 2048;
  dim3 g2(ys_cdiv(mh * mw, SG_THREADS), B);
  const bool fast = nm == 32 && ld_mc % 8 == 0 && ld_pr % 8 == 0;      // Proto / Segment default (Head.cs:238): compile-time nm, 16-byte rows
  if (dtype == YS_BF16) {
    if (fast) { YS_LAUNCH((seg_anchor_kernel<bf16_t, 32>), g1, SG_THREADS, st, a); YS_LAUNCH((seg_proto_grad_kernel<bf16_t, 32>), g2, SG_THREADS, st, a); }
    else { YS_LAUNCH((seg_anchor_kernel<bf16_t, 0>), g1, SG_THREADS, st, a); YS_LAUNCH((seg_proto_grad_kernel<bf16_t, 0>), g2, SG_THREADS, st, a); }
  } else {
    if (fast) { YS_LAUNCH((seg_anchor_kernel<float, 32>), g1, SG_THREADS, st, a); YS_LAUNCH((seg_proto_grad_kernel<float, 32>), g2, SG_THREADS, st, a); }
    else { YS_LAUNCH((seg_anchor_kernel<float, 0>), g1, SG_THREADS, st, a); YS_LAUNCH((seg_proto_grad_kernel<float, 0>), g2, SG_THREADS, st, a); }
  }
  YS_LAUNCH(seg_finalize_kernel, 1, SG_THREADS, st, a);
  return YS_OK;
}

// ---------------------------------------------------------------- Ops.process_mask (Ops.cs:462-489)
// masks = masks_in[n,nm] @ protos[nm, mh*mw]; crop to the box scaled by (mw/iw, mh/ih); optional bilinear upsample to
// (ih, iw) with align_corners=false; > 0.  One thread per output pixel.
__global__ void __launch_bounds__(SG_THREADS)
process_mask_kernel(const float* __restrict__ protos /*[nm][mh][mw]*/, const float* __restrict__ masks_in /*[n][nm]*/,
                    const float* __restrict__ boxes /*[n][4] xyxy in image pixels*/, int n, int nm, int mh, int mw, int ih, int iw,
                    int upsample, int trunc_crop, unsigned char* __restrict__ out) {
  const int oh = upsample ? ih : mh, ow = upsample ? iw : mw;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * oh * ow) return;
  const int x = (int)(i % ow), y = (int)((i / ow) % oh), j = (int)(i / ((long)ow * oh));
  const float wr = (float)mw / (float)iw, hr = (float)mh / (float)ih;
  float e[4] = {boxes[4 * j] * wr, boxes[4 * j + 1] * hr, boxes[4 * j + 2] * wr, boxes[4 * j + 3] * hr};
  int c0, c1, r0, r1;
  seg_crop_bounds(e, mw, mh, trunc_crop, n, c0, c1, r0, r1);
  auto val = [&](int rr, int cc) -> float {
    if (cc < c0 || cc >= c1 || rr < r0 || rr >= r1) return 0.f;
    float s = 0.f;
    for (int k = 0; k < nm; k++) s += masks_in[j * nm + k] * protos[((long)k * mh + rr) * mw + cc];
    return s;
  };
  float v;
  if (!upsample) {
    v = val(y, x);
  } else {
    // F.interpolate(mode=bilinear, align_corners=false): src = (dst + 0.5) * scale - 0.5, clamped at 0
    float sy = ((float)y + 0.5f) * ((float)mh / (float)ih) - 0.5f;
    float sx = ((float)x + 0.5f) * ((float)mw / (float)iw) - 0.5f;
    sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < mh - 1 ? 1 : 0), x1 = x0 + (x0 < mw - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    v = (1.f - ly) * ((1.f - lx) * val(y0, x0) + lx * val(y0, x1)) + ly * ((1.f - lx) * val(y1, x0) + lx * val(y1, x1));
  }
  out[i] = v > 0.f ? 1 : 0;
}

int ys_process_mask_launch(hipStream_t st, const float* protos, const float* masks_in, const float* boxes, int n, int nm, int mh,
                           int mw, int ih, int iw, int upsample, int trunc_crop, unsigned char* out) {
  const long total = (long)n * (upsample ? (long)ih * iw : (long)mh * mw);
  if (total <= 0) return YS_OK;
  YS_LAUNCH(process_mask_kernel, ys_cdiv(total, SG_THREADS), SG_THREADS, st, protos, masks_in, boxes, n, nm, mh, mw, ih, iw, upsample, trunc_crop, out);
  return YS_OK;
}
