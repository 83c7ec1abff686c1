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
//   nms_filter : one thread per (image, 4 anchors), 16-byte loads along the anchor axis
//   nms_sort   : one 1024-thread workgroup per image: LDS bitonic sort of 64-bit (score,index) keys, gather of the
//                class-offset boxes in score order
//   nms_mask   : n <= 4096 candidates: the upper-triangular suppression bit matrix, all (image, 64x64 block) pairs in parallel
//   nms_scan   : one wave per image walks the score order with the "removed" set in registers (torchvision's own scheme)
//   nms_greedy_big : n > 4096: serial workgroup-parallel sweep per kept box (O(kept*n), no n^2 matrix)
// Compiled with -ffp-contract=off: every fp32 op below rounds exactly like the CPU reference.
#include "ys_internal.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>

#define NMS_THREADS 1024
#define NMS_LDS_KEYS 16384
#define NMS_MASK_MAX 4096          // images with <= this many candidates take the bitmask path (mask rows of 64 x 64-bit words)
#define NMS_MASK_WORDS (NMS_MASK_MAX / 64)
#define NMS_MASK_WGS 16            // workgroups (256 threads) per image in the mask kernel
#define NMS_LDS_N_FWD 1024         // = NMS_LDS_N (defined with nms_scan_kernel)

// Candidate filter: one thread per (image, 4 consecutive anchors) when A % 4 == 0 (16-byte loads along the anchor axis, every
// wave instruction one contiguous 1 KB read, eight class channels in flight per trip), else one thread per anchor.  A pure
// stream of B*C*A*4 bytes.  (Splitting the classes over adjacent lanes: 3x slower -- scattered 16-byte accesses; over the four
// waves of a workgroup with an LDS merge: 83 vs 70 us on [64, 84, 8400] -- dropped.)
// ROT: oriented boxes (Ops.cs:286: `if (!rotated)` around the xywh -> xyxy conversion) -- the boxes stay xywh, the angle is the last channel
template <int V, bool ROT>
__global__ void __launch_bounds__(256)
nms_filter_kernel(float* __restrict__ pred, int C, int A, int nc, float conf_thres,
                  int* __restrict__ count, unsigned long long* __restrict__ keys, int keys_stride,
                  float* __restrict__ confs, int* __restrict__ clss) {
  const int b = blockIdx.y;
  const int a0r = (blockIdx.x * blockDim.x + threadIdx.x) * V;
  const bool live = a0r < A;                  // lanes past the last anchor run on anchor 0's data and are masked out of every store
  const int a0 = live ? a0r : 0;              // (no early return: the slot allocation below is a wave-wide operation)
  float* p = pred + (size_t)b * C * A;
  auto ld = [&](int ch, float (&o)[V]) {
    if (V == 4) { const float4 q = *(const float4*)(p + (size_t)ch * A + a0); o[0] = q.x; o[V > 1 ? 1 : 0] = q.y; o[V > 2 ? 2 : 0] = q.z; o[V > 3 ? 3 : 0] = q.w; }
    else o[0] = p[(size_t)ch * A + a0];
  };
  auto st = [&](int ch, const float (&o)[V]) {
    if (V == 4) { float4 q; q.x = o[0]; q.y = o[V > 1 ? 1 : 0]; q.z = o[V > 2 ? 2 : 0]; q.w = o[V > 3 ? 3 : 0]; *(float4*)(p + (size_t)ch * A + a0) = q; }
    else p[(size_t)ch * A + a0] = o[0];
  };
  if (!ROT) {
    // xywh -> xyxy in place (Ops.cs:76-79): x - w/2, y - h/2, x + w/2, y + h/2
    float cx[V], cy[V], w[V], h[V], x1[V], y1[V], x2[V], y2[V];
    ld(0, cx); ld(1, cy); ld(2, w); ld(3, h);
#pragma unroll
    for (int v = 0; v < V; v++) {
      const float hw = w[v] / 2.0f, hh = h[v] / 2.0f;
      x1[v] = cx[v] - hw; y1[v] = cy[v] - hh; x2[v] = cx[v] + hw; y2[v] = cy[v] + hh;
    }
    if (live) { st(0, x1); st(1, y1); st(2, x2); st(3, y2); }
  }
  float best[V];
  int bi[V];
  ld(4, best);                                // class maximum, first arg-max (strict >)
#pragma unroll
  for (int v = 0; v < V; v++) bi[v] = 0;
  int c = 1;
  for (; c + 15 < nc; c += 16) {              // 16 class channels (16 KB per wave) in flight per trip
    float q[16][V];
#pragma unroll
    for (int k = 0; k < 16; k++) ld(4 + c + k, q[k]);
#pragma unroll
    for (int k = 0; k < 16; k++)
#pragma unroll
      for (int v = 0; v < V; v++) if (q[k][v] > best[v]) { best[v] = q[k][v]; bi[v] = c + k; }
  }
  for (; c + 7 < nc; c += 8) {
    float q[8][V];
#pragma unroll
    for (int k = 0; k < 8; k++) ld(4 + c + k, q[k]);
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
      for (int v = 0; v < V; v++) if (q[k][v] > best[v]) { best[v] = q[k][v]; bi[v] = c + k; }
  }
  for (; c < nc; c++) {
    float v0[V];
    ld(4 + c, v0);
#pragma unroll
    for (int v = 0; v < V; v++) if (v0[v] > best[v]) { best[v] = v0[v]; bi[v] = c; }
  }
  // Candidate slots: ONE returning atomic per wave (the per-candidate form was up to V dependent device-scope round trips per
  // lane at the tail of a streaming kernel).  Slot order inside the image is irrelevant: the sort that follows is a total order
  // (score, then anchor index).
  const int lane = threadIdx.x & 63;
  unsigned long long bal[V];
  int total = 0;
#pragma unroll
  for (int v = 0; v < V; v++) { bal[v] = __ballot(live && best[v] > conf_thres); total += __popcll(bal[v]); }
  if (total == 0) return;                     // wave-uniform
  int base = 0;
  if (lane == 0) base = atomicAdd(&count[b], total);
  base = __shfl(base, 0);
#pragma unroll
  for (int v = 0; v < V; v++) {
    if (live && best[v] > conf_thres) {
      const int a = a0 + v;
      const int slot = base + __popcll(bal[v] & ((1ull << lane) - 1ull));
      const unsigned bits = ys_f2u(best[v]);  // best > conf >= 0 -> positive float, bit pattern monotone
      keys[(size_t)b * keys_stride + slot] = ((unsigned long long)(~bits) << 32) | (unsigned)a;
      confs[(size_t)b * A + a] = best[v];
      clss[(size_t)b * A + a] = bi[v];
    }
    base += __popcll(bal[v]);
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

// Per image: sort the (score, anchor) keys (descending score, ties -> lower anchor), cut to max_nms, gather the class-offset
// boxes / areas / anchor indices in that order, zero the outputs.  n_sorted[b] = candidates entering the greedy pass.
__global__ void __launch_bounds__(NMS_THREADS)
nms_sort_kernel(const float* __restrict__ pred, int C, int A, int nc, int max_det, int max_nms, float max_wh,
                int* __restrict__ count, unsigned long long* __restrict__ keys_g, int keys_stride,
                const int* __restrict__ clss, float4* __restrict__ sbox, float* __restrict__ sarea, int* __restrict__ sidx,
                unsigned char* __restrict__ supp_g, int ncap, int* __restrict__ n_sorted,
                float* __restrict__ out_rows, long long* __restrict__ out_keep, int* __restrict__ out_count,
                float4* __restrict__ scov) {
  __shared__ unsigned long long skeys[NMS_LDS_KEYS];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const float* p = pred + (size_t)b * C * A;
  const int row_w = 6 + (C - 4 - nc);
  int n = count[b];
  if (n > A) n = A;
  __syncthreads();
  if (tid == 0) count[b] = 0;                 // consumed: the next call's filter starts from zero without a memset launch
  float* orow = out_rows + (size_t)b * max_det * row_w;
  long long* okeep = out_keep + (size_t)b * max_det;
  // outputs of images with nothing kept stay zero / count 0 (Ops.cs:298-299,315-318)
  for (int i = tid; i < max_det * row_w; i += NMS_THREADS) orow[i] = 0.0f;
  for (int i = tid; i < max_det; i += NMS_THREADS) okeep[i] = 0;
  if (n == 0) {
    if (tid == 0) { out_count[b] = 0; n_sorted[b] = 0; }
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
    if (scov) {
      // rotated (Ops.cs:351): boxes = (xy + c, wh, angle = last column); covariance terms of Metrics._get_covariance_matrix
      // (Metrics.cs:264-283) once per box, as the reference computes them once per tensor
      q.x = p[a] + off; q.y = p[(size_t)A + a] + off; q.z = p[2 * (size_t)A + a]; q.w = p[3 * (size_t)A + a];
      const float r = p[(size_t)(C - 1) * A + a];
      const float ga = q.z * q.z / 12.0f, gb = q.w * q.w / 12.0f;
      const float cs = cosf(r), sn = sinf(r);
      const float cos2 = cs * cs, sin2 = sn * sn;
      float4 cv;
      cv.x = ga * cos2 + gb * sin2; cv.y = ga * sin2 + gb * cos2; cv.z = (ga - gb) * cs * sn; cv.w = 0.f;
      (scov + (size_t)b * ncap)[i] = cv;
      bx[i] = q; ar[i] = 0.f; si[i] = a; supp[i] = 0;
      continue;
    }
    q.x = p[a] + off;                       // Ops.cs:356 boxes = x[:, :4] + c
    q.y = p[(size_t)A + a] + off;
    q.z = p[2 * (size_t)A + a] + off;
    q.w = p[3 * (size_t)A + a] + off;
    bx[i] = q;
    ar[i] = (q.z - q.x) * (q.w - q.y);      // torchvision nms: areas = (x2-x1)*(y2-y1)
    si[i] = a;
    supp[i] = 0;
  }
  if (tid == 0) n_sorted[b] = n;
}

// torchvision's IoU test, operation for operation (fp32, IEEE division; compiled with -ffp-contract=off)
__device__ inline bool nms_suppresses(const float4& bi, float ai, const float4& bj, float aj, float iou_thres) {
  const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
  const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
  const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
  const float inter = w * h;
  const float ovr = inter / (ai + aj - inter);
  return ovr > iou_thres;
}

// Bitmask path, part 1 (n <= NMS_MASK_MAX): mask[i][w] = bits j - 64w of every later candidate j that box i would suppress.
// A wave owns a (64-row block, word) pair of the upper triangle: lane l is row 64 rb + l, the word's 64 column boxes are staged
// in LDS once and read as broadcasts.  NMS_MASK_WGS x 4 waves per image walk the pairs -- all images and all pairs in parallel
// instead of one serial O(kept x n) sweep with three workgroup barriers per kept box.
__global__ void __launch_bounds__(256)
nms_mask_kernel(const int* __restrict__ n_sorted, const float4* __restrict__ sbox, const float* __restrict__ sarea, int ncap,
                float iou_thres, unsigned long long* __restrict__ mask) {
  __shared__ float4 sB[4][64];
  __shared__ float sA[4][64];
  const int b = blockIdx.x;
  const int n = n_sorted[b];
  if (n <= 0 || n > NMS_MASK_MAX) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nb = (n + 63) >> 6;
  const int npairs = nb * (nb + 1) / 2;
  const float4* bx = sbox + (size_t)b * ncap;
  const float* ar = sarea + (size_t)b * ncap;
  unsigned long long* mk = mask + (size_t)b * NMS_MASK_MAX * NMS_MASK_WORDS;
  for (int pidx = blockIdx.y * 4 + wave; pidx < npairs; pidx += NMS_MASK_WGS * 4) {
    int rb = 0, rem = pidx;                 // rows rb own the words rb .. nb-1
    while (rem >= nb - rb) { rem -= nb - rb; rb++; }
    const int w = rb + rem;
    const int i = rb * 64 + lane, j0 = w * 64;
    const int jl = j0 + lane;
    ys_wave_sync();                         // the previous pair's LDS reads are done
    sB[wave][lane] = jl < n ? bx[jl] : make_float4(0.f, 0.f, 0.f, 0.f);
    sA[wave][lane] = jl < n ? ar[jl] : 0.f;
    ys_wave_sync();
    if (i < n) {
      const float4 bi = bx[i];
      const float ai = ar[i];
      unsigned long long bits = 0ull;
      for (int t = 0; t < 64; t++) {
        const int j = j0 + t;
        if (j > i && j < n && nms_suppresses(bi, ai, sB[wave][t], sA[wave][t], iou_thres)) bits |= 1ull << t;
      }
      mk[(size_t)i * NMS_MASK_WORDS + w] = bits;
    }
  }
}

// Bitmask path, part 2: the greedy order, one 1024-thread workgroup per image.
//   n <= NMS_LDS_N (the usual case after the confidence filter): the image's bit matrix (nms_mask_kernel) is staged in LDS by
//   all 1024 threads (building it here, 16 waves per image on 64 CUs, measured 2.5x slower than the chip-wide mask kernel), then
//   wave 0 walks the candidates in score order, word by word (64 candidates): lane l keeps word l of the "removed" set in
//   registers, the word being walked lives in a scalar that is tested / updated with scalar instructions (the diagonal mask word
//   of a kept row comes from a v_readlane) -- no cross-lane or memory round trip per candidate;
//   NMS_LDS_N < n <= NMS_MASK_MAX: the same walk over the matrix nms_mask_kernel left in global memory, 32 rows requested at a
//   time, the next chunk in flight while the current one is walked.
// The kept positions go to LDS and all waves gather the output rows in parallel (Ops.cs:357-366: rows, anchor indices, count).
#define NMS_LDS_N 1024
#define NMS_LDS_WORDS (NMS_LDS_N / 64)
#define NMS_SCAN_LDS ((size_t)NMS_LDS_N * NMS_LDS_WORDS * 8 + 16 * 64 * 20)
__device__ inline void
nms_greedy_big_body(const float* __restrict__ pred, int C, int A, int nc, float iou_thres, int max_det,
                    const int* __restrict__ n_sorted, const float* __restrict__ confs, const int* __restrict__ clss,
                    const float4* __restrict__ sbox, const float* __restrict__ sarea, const int* __restrict__ sidx,
                    unsigned char* __restrict__ supp_g, int ncap,
                    float* __restrict__ out_rows, long long* __restrict__ out_keep, int* __restrict__ out_count);
__global__ void __launch_bounds__(NMS_THREADS)
nms_scan_kernel(const float* __restrict__ pred, int C, int A, int nc, float iou_thres, int max_det, const int* __restrict__ n_sorted,
                const float4* __restrict__ sbox, const float* __restrict__ sarea, const int* __restrict__ sidx, int ncap,
                const float* __restrict__ confs, const int* __restrict__ clss, const unsigned long long* __restrict__ mask,
                unsigned char* __restrict__ supp_g,
                float* __restrict__ out_rows, long long* __restrict__ out_keep, int* __restrict__ out_count) {
  YS_DYN_LDS(lds);
  unsigned long long* lmask = (unsigned long long*)lds;                       // [n][nw] (n <= NMS_LDS_N)
  char* aux = (char*)lds + (size_t)NMS_LDS_N * NMS_LDS_WORDS * 8;
  float4* sB = (float4*)aux;                                                   // [16 waves][64] column boxes (matrix phase)
  float* sA = (float*)(aux + 16 * 64 * 16);                                    // [16][64] column areas
  int* s_kept = (int*)aux;                                                     // walk / gather phases (<= NMS_MASK_MAX entries fit: 20 KB)
  __shared__ int s_nkept;
  const int b = blockIdx.x;
  const int n = n_sorted[b];
  if (n <= 0) return;
  if (n > NMS_MASK_MAX) {                     // large image: the serial sweep (was a launch of its own, made for every call whatever n)
    nms_greedy_big_body(pred, C, A, nc, iou_thres, max_det, n_sorted, confs, clss, sbox, sarea, sidx, supp_g, ncap, out_rows, out_keep, out_count);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* p = pred + (size_t)b * C * A;
  const int extra = C - 4 - nc, row_w = 6 + extra;
  float* orow = out_rows + (size_t)b * max_det * row_w;
  long long* okeep = out_keep + (size_t)b * max_det;
  const int* si = sidx + (size_t)b * ncap;
  const int nw = (n + 63) >> 6;
  const bool in_lds = n <= NMS_LDS_N;
  if (in_lds) {
    // stage the image's bit matrix (built by nms_mask_kernel on the whole chip) in LDS, compact row pitch nw: 1024 threads, coalesced
    const unsigned long long* gm = mask + (size_t)b * NMS_MASK_MAX * NMS_MASK_WORDS;
    for (int idx = tid; idx < n * nw; idx += NMS_THREADS) {
      const int i = idx / nw, w = idx - i * nw;
      lmask[idx] = w >= (i >> 6) ? gm[(size_t)i * NMS_MASK_WORDS + w] : 0ull;
    }
    __syncthreads();
  }
  if (tid < 64) {
    const unsigned long long* mk = in_lds ? lmask : mask + (size_t)b * NMS_MASK_MAX * NMS_MASK_WORDS;
    const int pitch = in_lds ? nw : NMS_MASK_WORDS;
    unsigned long long removed = 0ull;
    int kept = 0;
    constexpr int CH = 16;                                            // 1024-thread workgroup: 128 registers per lane
    const int nch = (n + CH - 1) / CH;                                // chunks of 16 candidates; chunk c lies in word c >> 2
    unsigned long long rA[CH], rB[CH];
    auto fetch = [&](int c, unsigned long long (&rows)[CH]) {
      const int w = c >> 2, i0 = c * CH;
#pragma unroll
      for (int r = 0; r < CH; r++) {
        const int i = i0 + r < n ? i0 + r : n - 1;
        rows[r] = (lane < nw && lane >= w) ? mk[(size_t)i * pitch + lane] : 0ull;   // words below the diagonal were never written
      }
    };
    unsigned long long cur = 0ull;
    auto walk = [&](int c, const unsigned long long (&rows)[CH]) {
      const int w = c >> 2, i0 = c * CH, sh = (c & 3) * CH;
      if ((c & 3) == 0) cur = ys_readlane64(removed, w);              // wave-uniform: "removed" bits of candidates 64w .. 64w+63
#pragma unroll
      for (int r = 0; r < CH; r++) {
        const int i = i0 + r;
        if (i < n && kept < max_det && !((cur >> (sh + r)) & 1ull)) {
          if (lane == 0) s_kept[kept] = i;
          kept++;
          removed |= rows[r];
          cur |= ys_readlane64(rows[r], w);
        }
      }
    };
    if (in_lds) {                                                      // LDS rows: no prefetch needed
      for (int c = 0; c < nch && kept < max_det; c++) { fetch(c, rA); walk(c, rA); }
    } else {
      fetch(0, rA);
      for (int c = 0; c < nch && kept < max_det; c += 2) {             // the next chunk's rows are in flight while this one is walked
        if (c + 1 < nch) fetch(c + 1, rB);
        walk(c, rA);
        if (c + 1 >= nch || kept >= max_det) break;
        if (c + 2 < nch) fetch(c + 2, rA);
        walk(c + 1, rB);
      }
    }
    if (lane == 0) s_nkept = kept;
  }
  __syncthreads();
  const int kept = s_nkept;
  for (int idx = tid; idx < kept * row_w; idx += NMS_THREADS) {
    const int k = idx / row_w, e = idx - k * row_w;
    const int a = si[s_kept[k]];
    float v;
    if (e < 4) v = p[(size_t)e * A + a];
    else if (e == 4) v = confs[(size_t)b * A + a];
    else if (e == 5) v = (float)clss[(size_t)b * A + a];
    else v = p[(size_t)(4 + nc + e - 6) * A + a];
    orow[idx] = v;
  }
  for (int k = tid; k < kept; k += NMS_THREADS) okeep[k] = si[s_kept[k]];
  if (tid == 0) out_count[b] = kept;
}

// Large images (n > NMS_MASK_MAX, up to max_nms = 30000 candidates: an n x n bit matrix would not pay): the serial sweep, one
// 1024-thread workgroup per image, workgroup-parallel IoU pass per kept box (O(kept x n)).
__device__ inline void
nms_greedy_big_body(const float* __restrict__ pred, int C, int A, int nc, float iou_thres, int max_det,
                    const int* __restrict__ n_sorted, const float* __restrict__ confs, const int* __restrict__ clss,
                    const float4* __restrict__ sbox, const float* __restrict__ sarea, const int* __restrict__ sidx,
                    unsigned char* __restrict__ supp_g, int ncap,
                    float* __restrict__ out_rows, long long* __restrict__ out_keep, int* __restrict__ out_count) {
  __shared__ int s_red[NMS_THREADS / 64];
  __shared__ int s_cur;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int n = n_sorted[b];
  if (n <= NMS_MASK_MAX) return;
  const float* p = pred + (size_t)b * C * A;
  const int extra = C - 4 - nc;
  const int row_w = 6 + extra;
  float* orow = out_rows + (size_t)b * max_det * row_w;
  long long* okeep = out_keep + (size_t)b * max_det;
  const float4* bx = sbox + (size_t)b * ncap;
  const float* ar = sarea + (size_t)b * ncap;
  const int* si = sidx + (size_t)b * ncap;
  unsigned char* supp = supp_g + (size_t)b * ncap;
  if (tid == 0) s_cur = 0;
  __syncthreads();
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
    int j0 = tid;
    if (j0 <= i) j0 += ((i - j0) / NMS_THREADS + 1) * NMS_THREADS;
    for (int j = j0; j < n; j += NMS_THREADS) {
      if (supp[j]) continue;
      if (nms_suppresses(bi, ai, bx[j], ar[j], iou_thres)) supp[j] = 1;
    }
    __syncthreads();
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

// ------------------------------------------------------------------ oriented boxes: probiou + nms_rotated
// Metrics.batch_probiou (Metrics.cs:223-258), operation for operation in fp32: box 1 = (x1, y1, a1, b1, c1), box 2 likewise
__device__ inline float nms_probiou(float x1, float y1, float a1, float b1, float c1, float x2, float y2, float a2, float b2, float c2, float eps) {
  const float sa = a1 + a2, sb = b1 + b2, sc = c1 + c2;
  const float dy = y1 - y2, dx = x1 - x2;
  const float det = sa * sb - sc * sc;
  const float t1 = ((sa * (dy * dy) + sb * (dx * dx)) / (det + eps)) * 0.25f;
  const float t2 = ((sc * (x2 - x1) * dy) / (det + eps)) * 0.5f;
  const float d1 = fmaxf(a1 * b1 - c1 * c1, 0.f), d2 = fmaxf(a2 * b2 - c2 * c2, 0.f);
  const float t3 = logf(det / (4.0f * sqrtf(d1 * d2) + eps) + eps) * 0.5f;
  float bd = t1 + t2 + t3;
  bd = fminf(fmaxf(bd, eps), 100.0f);
  const float hd = sqrtf(1.0f - expf(-bd) + eps);
  return 1.0f - hd;
}

// Ops.nms_rotated (Ops.cs:373-401, use_triu): in score order, candidate j is dropped iff ANY earlier candidate i < j has
// probiou(i, j) >= threshold -- not the greedy rule of torchvision.nms: a dropped box still drops others.  One thread per j,
// the earlier boxes staged through LDS in tiles of 256; O(n^2 / 2) evaluations per image, any n up to max_nms.
__global__ void __launch_bounds__(256)
nms_rot_removed_kernel(const int* __restrict__ n_sorted, const float4* __restrict__ sbox, const float4* __restrict__ scov, int ncap,
                       float iou_thres, unsigned char* __restrict__ supp_g) {
  __shared__ float4 sB[256], sC[256];
  const int b = blockIdx.y;
  const int n = n_sorted[b];
  if ((int)blockIdx.x * 256 >= n) return;
  const float4* bx = sbox + (size_t)b * ncap;
  const float4* cv = scov + (size_t)b * ncap;
  const int j = blockIdx.x * 256 + threadIdx.x;
  float4 bj = make_float4(0.f, 0.f, 0.f, 0.f), cj = bj;
  if (j < n) { bj = bx[j]; cj = cv[j]; }
  bool removed = false;
  for (int t0 = 0; t0 <= (int)blockIdx.x * 256; t0 += 256) {
    __syncthreads();
    const int i0 = t0 + threadIdx.x;
    sB[threadIdx.x] = i0 < n ? bx[i0] : make_float4(0.f, 0.f, 0.f, 0.f);
    sC[threadIdx.x] = i0 < n ? cv[i0] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (j < n && !removed) {
      const int lim = (j - t0) < 256 ? (j - t0) : 256;                 // rows i = t0 .. t0 + lim - 1 < j
      for (int t = 0; t < lim; t++) {
        const float4 bi = sB[t], ci = sC[t];
        if (nms_probiou(bi.x, bi.y, ci.x, ci.y, ci.z, bj.x, bj.y, cj.x, cj.y, cj.z, 1e-7f) >= iou_thres) { removed = true; break; }
      }
    }
  }
  if (j < n) supp_g[(size_t)b * ncap + j] = removed ? 1 : 0;
}

// kept candidates in score order, first max_det (Ops.cs:360), output rows / anchor indices / count: one workgroup per image
__global__ void __launch_bounds__(NMS_THREADS)
nms_rot_pick_kernel(const float* __restrict__ pred, int C, int A, int nc, int max_det, const int* __restrict__ n_sorted,
                    const int* __restrict__ sidx, const unsigned char* __restrict__ supp_g, int ncap, const float* __restrict__ confs,
                    const int* __restrict__ clss, float* __restrict__ out_rows, long long* __restrict__ out_keep, int* __restrict__ out_count) {
  __shared__ int s_wcnt[NMS_THREADS / 64];
  __shared__ int s_base;
  YS_DYN_LDS(nms_pick_dyn);
  int* s_kept = (int*)nms_pick_dyn;               // [max_det]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = n_sorted[b];
  const int row_w = 6 + (C - 4 - nc);
  const float* p = pred + (size_t)b * C * A;
  const int* si = sidx + (size_t)b * ncap;
  const unsigned char* supp = supp_g + (size_t)b * ncap;
  float* orow = out_rows + (size_t)b * max_det * row_w;
  long long* okeep = out_keep + (size_t)b * max_det;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += NMS_THREADS) {
    const int i = c0 + tid;
    const bool keep = i < n && supp[i] == 0;
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wave; w++) before += s_wcnt[w];
    const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
    if (keep && pos < max_det) s_kept[pos] = i;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < NMS_THREADS / 64; w++) t += s_wcnt[w]; s_base += t; }
    __syncthreads();
    if (s_base >= max_det) break;
  }
  const int kept = s_base < max_det ? s_base : max_det;
  for (int idx = tid; idx < kept * row_w; idx += NMS_THREADS) {
    const int k = idx / row_w, e = idx - k * row_w;
    const int a = si[s_kept[k]];
    float v;
    if (e < 4) v = p[(size_t)e * A + a];
    else if (e == 4) v = confs[(size_t)b * A + a];
    else if (e == 5) v = (float)clss[(size_t)b * A + a];
    else v = p[(size_t)(4 + nc + e - 6) * A + a];
    orow[idx] = v;
  }
  for (int k = tid; k < kept; k += NMS_THREADS) okeep[k] = si[s_kept[k]];
  if (tid == 0) out_count[b] = kept;
}

// Metrics.probiou (pairwise, Metrics.cs:137-177; CIoU adds only the aspect-ratio term) and Metrics.batch_probiou (N x M)
__global__ void __launch_bounds__(256)
probiou_kernel(const float* __restrict__ o1, const float* __restrict__ o2, long n, long m, int pairwise, int ciou, float eps, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = pairwise ? n : n * m;
  if (idx >= total) return;
  const long i = pairwise ? idx : idx / m, j = pairwise ? idx : idx - i * m;
  const float* p1 = o1 + i * 5; const float* p2 = o2 + j * 5;
  auto cov = [](const float* q, float& a, float& b, float& c) {
    const float ga = q[2] * q[2] / 12.0f, gb = q[3] * q[3] / 12.0f;
    const float cs = cosf(q[4]), sn = sinf(q[4]);
    const float cos2 = cs * cs, sin2 = sn * sn;
    a = ga * cos2 + gb * sin2; b = ga * sin2 + gb * cos2; c = (ga - gb) * cs * sn;
  };
  float a1, b1, c1, a2, b2, c2;
  cov(p1, a1, b1, c1); cov(p2, a2, b2, c2);
  float iou = nms_probiou(p1[0], p1[1], a1, b1, c1, p2[0], p2[1], a2, b2, c2, eps);
  if (pairwise && ciou) {
    const float d = atanf(p2[2] / p2[3]) - atanf(p1[2] / p1[3]);
    const float v = 0.40528473456935109f * (d * d);                    // 4 / pi^2
    const float alpha = v / (v - iou + (1.0f + eps));
    iou = iou - v * alpha;
  }
  out[idx] = iou;
}

int ys_probiou_launch(hipStream_t st, const float* o1, const float* o2, long n, long m, int pairwise, int ciou, float eps, float* out) {
  const long total = pairwise ? n : n * m;
  if (total <= 0) return YS_OK;
  YS_LAUNCH(probiou_kernel, ys_cdiv(total, 256), 256, st, o1, o2, n, m, pairwise, ciou, eps, out);
  return YS_OK;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int ys_nms_launch(ys_ctx* ctx, float* pred, int B, int C, int A, float conf, float iou, int max_det,
                  int nc, int max_nms, int max_wh, float* out_rows, int64_t* out_keep, int32_t* out_count, int rotated) {
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
  const size_t o_nsort = off; off = align_up(off + sizeof(int) * B, 256);
  const bool big_possible = ncap > NMS_MASK_MAX;
  const size_t o_mask = off;  off = align_up(off + sizeof(unsigned long long) * (size_t)B * NMS_MASK_MAX * NMS_MASK_WORDS, 256);
  const size_t o_cov = off;   if (rotated) off = align_up(off + sizeof(float4) * (size_t)B * ncap, 256);
  if (off > ctx->nms_ws_bytes) {
    if (ctx->nms_ws) { YS_CHECK_HIP(hipStreamSynchronize(ctx->stream)); YS_CHECK_HIP(hipFree(ctx->nms_ws)); ctx->nms_ws = nullptr; ctx->nms_ws_bytes = 0; }
    YS_CHECK_HIP(hipMalloc(&ctx->nms_ws, off));
    ctx->nms_ws_bytes = off;
    ctx->nms_count_clean = false;
    ctx->nms_count_clean_B = 0;
  }
  char* ws = (char*)ctx->nms_ws;
  int* count = (int*)(ws + o_count);
  // new workspace / a call that did not reach its sort kernel / MORE images than the counters known to be zero (a smaller-B call only
  // cleared its own B counters; with a layout that still fits the old workspace the rest is padding or old key bytes)
  if (!ctx->nms_count_clean || B > ctx->nms_count_clean_B) YS_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int) * B, ctx->stream));
  ctx->nms_count_clean = false;
  YsKprofScope prof(ctx->stream, "nms");
  unsigned long long* keys = (unsigned long long*)(ws + o_keys);
  float* confs = (float*)(ws + o_conf); int* clss = (int*)(ws + o_cls);
  float4* sbox = (float4*)(ws + o_box); float* sarea = (float*)(ws + o_area); int* sidx = (int*)(ws + o_idx);
  unsigned char* supp = (unsigned char*)(ws + o_supp); int* nsort = (int*)(ws + o_nsort);
  unsigned long long* mask = (unsigned long long*)(ws + o_mask);
  float4* scov = rotated ? (float4*)(ws + o_cov) : nullptr;
  if (A % 4 == 0) {
    dim3 g1(ys_cdiv(A / 4, 64), B);
    if (rotated) YS_LAUNCH((nms_filter_kernel<4, true>), g1, 64, ctx->stream, pred, C, A, nc, conf, count, keys, np2, confs, clss);
    else YS_LAUNCH((nms_filter_kernel<4, false>), g1, 64, ctx->stream, pred, C, A, nc, conf, count, keys, np2, confs, clss);
  } else {
    dim3 g1(ys_cdiv(A, 256), B);
    if (rotated) YS_LAUNCH((nms_filter_kernel<1, true>), g1, 256, ctx->stream, pred, C, A, nc, conf, count, keys, np2, confs, clss);
    else YS_LAUNCH((nms_filter_kernel<1, false>), g1, 256, ctx->stream, pred, C, A, nc, conf, count, keys, np2, confs, clss);
  }
  YS_LAUNCH(nms_sort_kernel, B, NMS_THREADS, ctx->stream, (const float*)pred, C, A, nc, max_det, max_nms, (float)max_wh, count,
            keys, np2, (const int*)clss, sbox, sarea, sidx, supp, ncap, nsort, out_rows, (long long*)out_keep, (int*)out_count, scov);
  ctx->nms_count_clean = true;                // nms_sort_kernel zeroes the B counters it read
  ctx->nms_count_clean_B = B;
  if (rotated) {
    YS_LAUNCH(nms_rot_removed_kernel, dim3(ys_cdiv(ncap, 256), B), 256, ctx->stream, (const int*)nsort, (const float4*)sbox, (const float4*)scov, ncap, iou, supp);
    YS_LAUNCH_LDS(nms_rot_pick_kernel, B, NMS_THREADS, (size_t)max_det * sizeof(int), ctx->stream, (const float*)pred, C, A, nc, max_det, (const int*)nsort,
                  (const int*)sidx, (const unsigned char*)supp, ncap, (const float*)confs, (const int*)clss, out_rows, (long long*)out_keep, (int*)out_count);
    YS_CHECK_HIP(hipGetLastError());
    return YS_OK;
  }
  YS_LAUNCH(nms_mask_kernel, dim3(B, NMS_MASK_WGS), 256, ctx->stream, (const int*)nsort, (const float4*)sbox, (const float*)sarea, ncap, iou, mask);
  static_assert(NMS_LDS_N_FWD == NMS_LDS_N, "mask / scan kernels disagree on the LDS-resident size");
  static std::atomic<unsigned> attr_done{0};
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (!(attr_done.load(std::memory_order_relaxed) & (1u << (dev_id & 31)))) {
    hipFuncSetAttribute((const void*)nms_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NMS_SCAN_LDS);   // + 4 B static: 160 KB flat is refused
    attr_done.fetch_or(1u << (dev_id & 31), std::memory_order_relaxed);
  }
  YS_LAUNCH_LDS(nms_scan_kernel, B, NMS_THREADS, NMS_SCAN_LDS, ctx->stream, (const float*)pred, C, A, nc, iou, max_det, (const int*)nsort,
                (const float4*)sbox, (const float*)sarea, (const int*)sidx, ncap, (const float*)confs, (const int*)clss,
                (const unsigned long long*)mask, supp, out_rows, (long long*)out_keep, (int*)out_count);
  (void)big_possible;
  YS_CHECK_HIP(hipGetLastError());
  return YS_OK;
}
