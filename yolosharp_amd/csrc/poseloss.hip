// poseloss.hip -- v8PoseLoss keypoint terms (Utils/Loss.cs:870-1071, KeypointLoss :169-188).
//
//   for every foreground anchor a of image i (assigned GT g, stride s, anchor point (ax, ay)):        Loss.cs:1043-1066
//     gt_k   = keypoints[label of (i, g)][k] * (W, H) / s                                               (:948-949, :1040)
//     pred_k = (raw_x * 2 + ax - 0.5, raw_y * 2 + ay - 0.5, raw_v)                                      kpts_decode (:967-974)
//     area   = w * h of the assigned box / s^2                                                          (:1047-1049)
//     mask_k = gt visibility != 0 (kpt_dim 3) or 1                                                      (:1051)
//     e_k    = |pred_k - gt_k|^2 / ((2 sigma_k)^2 (area + 1e-9) 2)                                      (:183)
//     pose  += K / (sum_k mask_k + 1e-6) * (1 - exp(-e_k)) * mask_k ;  kobj += BCEWithLogits(raw_v, mask_k)
//   pose = mean over (fg anchors x K) * hyp_pose, kobj likewise * hyp_kobj                               (:185, :962-963)
// The detection part (assignment, box / cls / dfl) is loss.hip and runs first; the keypoint row of a padded GT slot is the label
// row loss_prep_kernel put there (gt_src), i.e. the reference's within-image rank for collate-ordered labels (:1001-1005).
// Gradients w.r.t. the raw keypoint outputs are written analytically: one thread per (foreground entry, keypoint), fixed-order
// reductions, no atomics.
#include "ys_internal.h"
#include "ys_kernels.h"

#define PL_THREADS 256

template <class T>
__global__ void __launch_bounds__(PL_THREADS)
pose_entry_kernel(PoseArgs a) {
  __shared__ float s_k[PL_THREADS], s_o[PL_THREADS];
  const int tid = threadIdx.x;
  const int ntot = a.off[a.B];
  const long total = (long)ntot * a.K;
  const T* kp = (const T*)a.kp;
  T* dkp = (T*)a.dkp;
  float lk = 0.f, lo = 0.f;
  for (long i = (long)blockIdx.x * PL_THREADS + tid; i < total; i += (long)gridDim.x * PL_THREADS) {
    const int e = (int)(i / a.K), k = (int)(i - (long)e * a.K);
    int lo_b = 0, hi_b = a.B - 1;                       // image of the entry: off[b] <= e < off[b+1]
    while (lo_b < hi_b) { const int mid = (lo_b + hi_b + 1) >> 1; if (a.off[mid] <= e) lo_b = mid; else hi_b = mid - 1; }
    const int b = lo_b, an = a.list[e];
    const long row = (long)b * a.A + an;
    const int g = a.fg_gt[row];
    const int src = a.gt_src[(long)b * a.gcap + g];
    int l_off = a.lvl_off[0], l_w = a.lvl_w[0], l_s = a.lvl_stride[0];
    for (int l = 1; l < a.nl; l++) if (an >= a.lvl_off[l]) { l_off = a.lvl_off[l]; l_w = a.lvl_w[l]; l_s = a.lvl_stride[l]; }
    const int cell = an - l_off;
    const float st = (float)l_s, gxo = (float)(cell % l_w), gyo = (float)(cell / l_w);   // anchor point - 0.5
    const float* lab = a.keypoints + ((long)src * a.K) * a.D;
    int nvis = a.K;
    if (a.D == 3) { nvis = 0; for (int j = 0; j < a.K; j++) nvis += lab[j * 3 + 2] != 0.f ? 1 : 0; }
    const float fac = (float)a.K / ((float)nvis + 1e-6f);
    const float gx = lab[k * a.D] * (float)a.W / st, gy = lab[k * a.D + 1] * (float)a.H / st;
    const float mk = a.D == 3 ? (lab[k * 3 + 2] != 0.f ? 1.f : 0.f) : 1.f;
    const float* gb = a.gt_box + ((long)b * a.gcap + g) * 4;
    const float area = (gb[2] / st - gb[0] / st) * (gb[3] / st - gb[1] / st);
    const T* r = kp + row * a.ld + (long)k * a.D;
    const float rx = Elem<T>::to_f(r[0]), ry = Elem<T>::to_f(r[1]);
    const float px = rx * 2.0f + gxo, py = ry * 2.0f + gyo;
    const float dx = px - gx, dy = py - gy;
    const float sg = 2.0f * a.sigma[k];
    const float den = sg * sg * (area + 1e-9f) * 2.0f;
    const float ev = (dx * dx + dy * dy) / den;
    const float ex = expf(-ev);
    lk += fac * (1.0f - ex) * mk;
    const float sc = a.B / ((float)ntot * (float)a.K);                 // d(loss.sum()) / d(mean term): items * batch_size (:965)
    const float gp = sc * a.hyp_pose * fac * mk * ex / den * 2.0f;     // d/dpx = gp * dx ; d/d raw_x = 2 * that
    T* d = dkp + row * a.ld + (long)k * a.D;
    d[0] = Elem<T>::from_f(gp * dx * 2.0f);
    d[1] = Elem<T>::from_f(gp * dy * 2.0f);
    if (a.D == 3) {
      const float rv = Elem<T>::to_f(r[2]);
      lo += fmaxf(rv, 0.f) - rv * mk + log1pf(expf(-fabsf(rv)));      // BCEWithLogits (:1064)
      d[2] = Elem<T>::from_f(sc * a.hyp_kobj * (ys_sigmoid(rv) - mk));
    }
  }
  s_k[tid] = lk; s_o[tid] = lo;
  __syncthreads();
  for (int st = PL_THREADS / 2; st > 0; st >>= 1) {
    if (tid < st) { s_k[tid] += s_k[tid + st]; s_o[tid] += s_o[tid + st]; }
    __syncthreads();
  }
  if (tid == 0) { a.part[2 * blockIdx.x] = s_k[0]; a.part[2 * blockIdx.x + 1] = s_o[0]; }
}

__global__ void __launch_bounds__(PL_THREADS)
pose_finalize_kernel(PoseArgs a, int nblk) {
  __shared__ double s_k[PL_THREADS], s_o[PL_THREADS];
  const int tid = threadIdx.x;
  double k = 0.0, o = 0.0;
  for (int i = tid; i < nblk; i += PL_THREADS) { k += (double)a.part[2 * i]; o += (double)a.part[2 * i + 1]; }
  s_k[tid] = k; s_o[tid] = o;
  __syncthreads();
  for (int st = PL_THREADS / 2; st > 0; st >>= 1) {
    if (tid < st) { s_k[tid] += s_k[tid + st]; s_o[tid] += s_o[tid + st]; }
    __syncthreads();
  }
  if (tid == 0) {
    const int ntot = a.off[a.B];
    const double n = (double)ntot * (double)a.K;
    const float pose = ntot > 0 ? (float)(s_k[0] / n) * a.hyp_pose : 0.f;                   // Loss.cs:945 guard, :962
    const float kobj = ntot > 0 && a.D == 3 ? (float)(s_o[0] / n) * a.hyp_kobj : 0.f;       // :963
    a.scalars[10] = pose;
    a.scalars[11] = kobj;
    a.scalars[4] += (pose + kobj) * (float)a.B;                                             // total = sum(items) * batch_size (:965)
  }
}

int ys_loss_pose_grid(int B, int A) { const long n = ((long)B * A + PL_THREADS - 1) / PL_THREADS; return (int)(n < 1024 ? n : 1024); }

int ys_loss_pose_launch(hipStream_t st, int dtype, const PoseArgs& a0, int* cnt, int* off, int* list) {
  PoseArgs a = a0;
  if (a.K < 1 || a.K > YS_POSE_KMAX || (a.D != 2 && a.D != 3)) { ys_set_error("pose loss: %d keypoints of dim %d unsupported", a.K, a.D); return YS_ERR_UNSUPPORTED; }
  a.off = off; a.list = list;
  const size_t es = dtype == YS_BF16 ? 2 : 4;
  YS_CHECK_HIP(hipMemsetAsync(a.dkp, 0, (size_t)a.B * a.A * a.ld * es, st));   // background anchors get no keypoint gradient
  YS_TRY(ys_fg_list_launch(st, a.fg_gt, a.B, a.A, cnt, off, list));
  const int nblk = ys_loss_pose_grid(a.B, a.A);
  if (dtype == YS_BF16) YS_LAUNCH((pose_entry_kernel<bf16_t>), nblk, PL_THREADS, st, a);
  else YS_LAUNCH((pose_entry_kernel<float>), nblk, PL_THREADS, st, a);
  YS_LAUNCH(pose_finalize_kernel, 1, PL_THREADS, st, a, nblk);
  return YS_OK;
}
