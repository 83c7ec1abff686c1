// elementwise.hip -- HBM-bound NHWC kernels around the convolutions (gfx950).
//   BatchNorm2d(eps 1e-3, momentum 0.03) + SiLU forward/backward  (Modules/Convs.cs:36-62)
//   MaxPool2d(5,1,2) chain of SPPF                                  (Modules/Block.cs:236-285)
//   Upsample(x2, nearest) + Concat as channel-slice writes           (Models/Yolo.cs:70-75, Convs.cs:435-448)
//   Detect._inference decode                                         (Modules/Head.cs:204-223, Block.cs:40-45)
//   AdamW step                                                       (YoloBaseTaskModel.cs:144-153, Amp.cs:355-372)
// All tensors are [rows][ldc] views with a channel offset; one thread moves 16 bytes (8 bf16 / 4 f32).
#include "ys_internal.h"
#include "ys_kernels.h"
#include <cstdlib>

#define EW_THREADS 256

// ------------------------------------------------------------------ input pack / unpack
// one thread per pixel: C planar reads (coalesced across threads), one packed row of cpad channels written with 16-byte stores
template <class T, class SRC>
__global__ void __launch_bounds__(EW_THREADS)
pack_input_kernel(const SRC* __restrict__ x, int B, int C, int h, int w, int H, int W, int cpad, float div, float padv,
                  T* __restrict__ y) {
  constexpr int EPL = Elem<T>::EPL;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long HW = (long)H * W;
  if (i >= (long)B * HW) return;
  long b; int py, px;
  if ((long)B * HW < (1L << 31)) {                        // 32-bit index arithmetic (two 64-bit divisions are ~160 VALU instructions)
    const unsigned iu = (unsigned)i, hw = (unsigned)HW, bu = iu / hw, pu = iu - bu * hw;
    b = bu; py = (int)(pu / (unsigned)W); px = (int)(pu - (unsigned)py * (unsigned)W);
  } else {
    b = i / HW; const long p = i - b * HW;
    py = (int)(p / W); px = (int)(p - (long)py * W);
  }
  const bool in = py < h && px < w;                       // bottom / right padding (Detector.cs:33-41)
  const bool unit = div == 1.0f;                          // fp32 images in [0,1]: no division at all
  for (int c0 = 0; c0 < cpad; c0 += EPL) {
    float f[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      const int c = c0 + e;
      float v = 0.f;
      if (c < C) {
        v = padv;
        if (in) { const float r = (float)x[((b * C + c) * h + py) * (long)w + px]; v = unit ? r : r / div; }
      }
      f[e] = v;
    }
    ys_st16(y + i * cpad + c0, ys_pack<T>(f));
  }
}
// fp32 image planes, C <= EPL (one 16-byte vector per pixel), W % 4 == 0: four consecutive pixels per thread -- 16-byte plane loads,
// 64 contiguous bytes written
template <class T>
__global__ void __launch_bounds__(EW_THREADS)
pack_input4_kernel(const float* __restrict__ x, long npix4, int C, long HW, T* __restrict__ y) {
  constexpr int EPL = Elem<T>::EPL;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // group of 4 pixels
  if (i >= npix4) return;
  const long pix = i * 4;
  const long b = pix / HW, p = pix - b * HW;
  float4 pl[EPL];
#pragma unroll
  for (int c = 0; c < EPL; c++) pl[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < EPL; c++)
    if (c < C) pl[c] = *(const float4*)(x + (b * C + c) * HW + p);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    float f[EPL];
#pragma unroll
    for (int c = 0; c < EPL; c++) f[c] = k == 0 ? pl[c].x : k == 1 ? pl[c].y : k == 2 ? pl[c].z : pl[c].w;
    ys_st16(y + (pix + k) * EPL, ys_pack<T>(f));
  }
}

int ys_pack_input_launch(hipStream_t st, int dtype, const float* x, int B, int C, int H, int W, int cpad, void* y) {
  const long n = (long)B * H * W;
  const int epl = dtype == YS_BF16 ? 8 : 4;
  if (cpad == epl && C <= epl && C <= 4 && W % 4 == 0 && ((size_t)x & 15) == 0) {   // the model's image input
    const long n4 = n / 4;
    if (dtype == YS_BF16) YS_LAUNCH((pack_input4_kernel<bf16_t>), ys_cdiv(n4, EW_THREADS), EW_THREADS, st, x, n4, C, (long)H * W, (bf16_t*)y);
    else YS_LAUNCH((pack_input4_kernel<float>), ys_cdiv(n4, EW_THREADS), EW_THREADS, st, x, n4, C, (long)H * W, (float*)y);
    return YS_OK;
  }
  if (dtype == YS_BF16) YS_LAUNCH((pack_input_kernel<bf16_t, float>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, x, B, C, H, W, H, W, cpad, 1.0f, 0.0f, (bf16_t*)y);
  else YS_LAUNCH((pack_input_kernel<float, float>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, x, B, C, H, W, H, W, cpad, 1.0f, 0.0f, (float*)y);
  return YS_OK;
}
// uint8 [B,C,h,w] (0..255) -> NHWC T [B,H,W,cpad]: value / 255, bottom / right padding with 114 / 255 (Detector.cs:33-41)
int ys_pack_input_u8_launch(hipStream_t st, int dtype, const unsigned char* x, int B, int C, int h, int w, int H, int W, int cpad, void* y) {
  const long n = (long)B * H * W;
  const float s = 255.0f, pv = 114.0f / 255.0f;        // true division like the reference's `/ 255.0f`
  if (dtype == YS_BF16) YS_LAUNCH((pack_input_kernel<bf16_t, unsigned char>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, x, B, C, h, w, H, W, cpad, s, pv, (bf16_t*)y);
  else YS_LAUNCH((pack_input_kernel<float, unsigned char>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, x, B, C, h, w, H, W, cpad, s, pv, (float*)y);
  return YS_OK;
}

// LetterBox / Rectangle (Data/Augment.cs:698-857): out[c, y, x] = in[c, src_y(y - pad_u), src_x(x - pad_l)] inside the resized
// window [pad_u, pad_u + new_h) x [pad_l, pad_l + new_w), `color` elsewhere.  The resize is
// torchvision.transforms.functional.resize(img, new_h, new_w) of TorchVision(.NET), whose default interpolation is NEAREST
// (third-party, not vendored): ATen's legacy nearest rule  src = min(floor(dst * (float)in / out), in - 1),
// evaluated here in the same fp32 arithmetic.  One thread per output pixel and plane; ET = unsigned char (images) or float (masks).
template <class ET>
__global__ void __launch_bounds__(EW_THREADS)
letterbox_kernel(const ET* __restrict__ x, int C, int h, int w, ET* __restrict__ y, int H, int W, int new_h, int new_w,
                 int pad_u, int pad_l, float color) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)C * H * W;
  if (i >= n) return;
  const int ox = (int)(i % W);
  const int oy = (int)((i / W) % H);
  const int c = (int)(i / ((long)W * H));
  const int ry = oy - pad_u, rx = ox - pad_l;
  ET v = (ET)color;
  if (ry >= 0 && ry < new_h && rx >= 0 && rx < new_w) {
    const float sy = (float)h / (float)new_h, sx = (float)w / (float)new_w;
    int iy = (int)floorf((float)ry * sy), ix = (int)floorf((float)rx * sx);
    iy = iy < h - 1 ? iy : h - 1; ix = ix < w - 1 ? ix : w - 1;
    v = x[((long)c * h + iy) * w + ix];
  }
  y[i] = v;
}
int ys_letterbox_launch(hipStream_t st, int is_float, const void* x, int C, int h, int w, void* y, int H, int W, int new_h, int new_w,
                        int pad_u, int pad_l, float color) {
  const long n = (long)C * H * W;
  if (n <= 0) return YS_OK;
  if (is_float) YS_LAUNCH((letterbox_kernel<float>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const float*)x, C, h, w, (float*)y, H, W, new_h, new_w, pad_u, pad_l, color);
  else YS_LAUNCH((letterbox_kernel<unsigned char>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const unsigned char*)x, C, h, w, (unsigned char*)y, H, W, new_h, new_w, pad_u, pad_l, color);
  return YS_OK;
}

template <class T>
__global__ void __launch_bounds__(EW_THREADS)
unpack_nchw_kernel(const T* __restrict__ x, int ldc, int coff, int B, int C, long rpb, float* __restrict__ y) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * C * rpb;
  if (i >= n) return;
  const long p = i % rpb;
  const long bc = i / rpb;
  const int c = (int)(bc % C);
  const long b = bc / C;
  y[i] = Elem<T>::to_f(x[(b * rpb + p) * ldc + coff + c]);
}
int ys_unpack_nchw_launch(hipStream_t st, int dtype, const void* x, int ldc, int coff, int B, int C, long rpb, float* y) {
  const long n = (long)B * C * rpb;
  if (dtype == YS_BF16) YS_LAUNCH((unpack_nchw_kernel<bf16_t>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const bf16_t*)x, ldc, coff, B, C, rpb, y);
  else YS_LAUNCH((unpack_nchw_kernel<float>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const float*)x, ldc, coff, B, C, rpb, y);
  return YS_OK;
}

// [B][rpb][ldc] view -> channel block of a wider NCHW tensor: y[b*y_bstride + y_off + c*rpb + p]
template <class T>
__global__ void __launch_bounds__(EW_THREADS)
unpack_nchw_strided_kernel(const T* __restrict__ x, int ldc, int coff, int B, int C, long rpb, float* __restrict__ y, long y_bstride, long y_off) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C * rpb) return;
  const long p = i % rpb;
  const long bc = i / rpb;
  const int c = (int)(bc % C);
  const long b = bc / C;
  y[b * y_bstride + y_off + (long)c * rpb + p] = Elem<T>::to_f(x[(b * rpb + p) * ldc + coff + c]);
}
int ys_unpack_nchw_strided_launch(hipStream_t st, int dtype, const void* x, int ldc, int coff, int B, int C, long rpb, float* y,
                                  long y_bstride, long y_off) {
  const long n = (long)B * C * rpb;
  if (dtype == YS_BF16) YS_LAUNCH((unpack_nchw_strided_kernel<bf16_t>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const bf16_t*)x, ldc, coff, B, C, rpb, y, y_bstride, y_off);
  else YS_LAUNCH((unpack_nchw_strided_kernel<float>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const float*)x, ldc, coff, B, C, rpb, y, y_bstride, y_off);
  return YS_OK;
}

// ------------------------------------------------------------------ block reduction helper (double)
// two sums at once: wave shuffles, then one LDS exchange of the EW_THREADS/64 wave totals combined in a fixed order (the LDS
// tree above costs 9 barriers per value -- most of a ~5 us finalize kernel)
__device__ inline void block_sum2_d(double& a, double& b, double* sbuf) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
  if (lane == 0) { sbuf[2 * wave] = a; sbuf[2 * wave + 1] = b; }
  __syncthreads();
  double ta = 0.0, tb = 0.0;
#pragma unroll
  for (int w = 0; w < EW_THREADS / 64; w++) { ta += sbuf[2 * w]; tb += sbuf[2 * w + 1]; }
  a = ta; b = tb;
}

// ------------------------------------------------------------------ BN forward finalize
// one workgroup per channel: deterministic sum of the conv epilogue's partials in double
__global__ void __launch_bounds__(EW_THREADS)
bn_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count, const float* __restrict__ gamma,
                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ run_mean,
                   float* __restrict__ run_var, float* __restrict__ nbt, float* __restrict__ scale,
                   float* __restrict__ shift, float* __restrict__ mean_o, float* __restrict__ rstd_o) {
  __shared__ double sbuf[EW_THREADS];
  const int c = blockIdx.x;
  // per-channel operands of the tail are fetched first: their latency overlaps the partial loads instead of forming a second
  // dependent round trip after the reduction
  float g = 0.f, bt = 0.f, rm0 = 0.f, rv0 = 0.f, nbt0 = 0.f;
  if (threadIdx.x == 0) { g = gamma[c]; bt = beta[c]; rm0 = run_mean[c]; rv0 = run_var[c]; if (c == 0 && nbt) nbt0 = nbt[0]; }
  double s1 = 0.0, s2 = 0.0;
  for (int k = threadIdx.x; k < nblk; k += EW_THREADS) {
    s1 += (double)partial[((long)k * 2 + 0) * C + c];
    s2 += (double)partial[((long)k * 2 + 1) * C + c];
  }
  block_sum2_d(s1, s2, sbuf);
  if (threadIdx.x == 0) {
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;  // biased (torch BatchNorm2d training normalisation)
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    scale[c] = g * rstd;
    shift[c] = bt - (float)mean * g * rstd;
    mean_o[c] = (float)mean;
    rstd_o[c] = rstd;
    // running stats: momentum 0.03, unbiased variance (Convs.cs:41-42,48)
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    run_mean[c] = (1.0f - momentum) * rm0 + momentum * (float)mean;
    run_var[c] = (1.0f - momentum) * rv0 + momentum * (float)unb;
    if (c == 0 && nbt) nbt[0] = nbt0 + 1.0f;
  }
}
int ys_bn_finalize_launch(hipStream_t st, const float* partial, int nblk, int C, long count, const float* gamma,
                          const float* beta, float eps, float momentum, float* run_mean, float* run_var, float* nbt,
                          float* scale, float* shift, float* mean, float* rstd) {
  YS_LAUNCH(bn_finalize_kernel, C, EW_THREADS, st, partial, nblk, C, (double)count, gamma, beta, eps, momentum,
            run_mean, run_var, nbt, scale, shift, mean, rstd);
  return YS_OK;
}

__global__ void __launch_bounds__(EW_THREADS)
bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.0f / sqrtf(rv[c] + eps);
  scale[c] = gamma[c] * rstd;
  shift[c] = beta[c] - rm[c] * gamma[c] * rstd;
}
int ys_bn_eval_coeffs_launch(hipStream_t st, int C, const float* gamma, const float* beta, const float* rm,
                             const float* rv, float eps, float* scale, float* shift) {
  YS_LAUNCH(bn_eval_coeffs_kernel, ys_cdiv(C, EW_THREADS), EW_THREADS, st, C, gamma, beta, rm, rv, eps, scale, shift);
  return YS_OK;
}

// ------------------------------------------------------------------ BN + act apply (training forward, pass 2)
// ACT is a template parameter: a run-time flag inside the unrolled element loop splits it into basic blocks (see chan_reduce);
// rows * CG < 2^31 (every graph here) takes 32-bit index arithmetic instead of a ~80-instruction 64-bit division.
template <class T, bool ACT>
__global__ void __launch_bounds__(EW_THREADS)
bn_act_apply_kernel(const T* __restrict__ y, long rows, int C, const float* __restrict__ scale,
                    const float* __restrict__ shift, const T* __restrict__ res, int res_ldc, int res_coff,
                    T* __restrict__ z, int z_ldc, int z_coff, int small, unsigned* __restrict__ amax) {
  constexpr int EPL = Elem<T>::EPL;
  const int CG = C / EPL;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float mx = 0.f;
  if (i < rows * CG) {
    long row; int c;
    if (small) { const unsigned iu = (unsigned)i, r = iu / (unsigned)CG; row = r; c = (int)(iu - r * (unsigned)CG) * EPL; }
    else { row = i / CG; c = (int)(i - row * CG) * EPL; }
    float f[EPL], r[EPL], sc[EPL], sh[EPL];
    ys_unpack<T>(ys_ld16(y + row * C + c), f);
    if (res) ys_unpack<T>(ys_ld16(res + row * res_ldc + res_coff + c), r);
    ys_ldcoef<EPL>(scale + c, sc);
    ys_ldcoef<EPL>(shift + c, sh);
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      float u = f[e] * sc[e] + sh[e];
      if (ACT) u = ys_silu(u);
      if (res) u += r[e];
      f[e] = u;
      mx = fmaxf(mx, fabsf(u));
    }
    ys_st16(z + row * z_ldc + z_coff + c, ys_pack<T>(f));
  }
  // fp8 mode: amax(|z|) of the tensor this pass writes feeds the NEXT step's activation scale (delayed scaling, f8.hip); the
  // maximum of non-negative floats is order-independent, so the atomic keeps the step deterministic
  if (amax) ys_amax_update(amax, mx);
}
// amax bookkeeping of a grid-stride pass: workgroup maximum through LDS, then ONE slot update per workgroup
__device__ inline void ys_amax_update_wg(unsigned* slots, float mx) {
  __shared__ float s_wmax[EW_THREADS / 64];
  mx = ys_wave_max(mx);
  if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = s_wmax[0];
    for (int w = 1; w < EW_THREADS / 64; w++) v = fmaxf(v, s_wmax[w]);
    if (v > 0.f) {
      unsigned* slot = slots + (blockIdx.x & (YS_AMAX_WAYS - 1));
      const unsigned cur = *(volatile unsigned*)slot;
      if (ys_f2u(v) > cur) atomicMax(slot, ys_f2u(v));
    }
  }
}

// fp8 mode, bf16 storage: the same pass also writes the e4m3 image (dense [rows][C], the consumer's delayed activation scale) of the
// tensor it produces and records amax(|z|) -- used when the next convolution of the schedule reads exactly this
// view and will run the fp8 blocked-GEMM kernel (Bottleneck cv1 -> cv2), instead of a quantisation pass over z.  Grid-stride with a
// bounded grid: one amax atomic per workgroup.
template <bool ACT>
__global__ void __launch_bounds__(EW_THREADS)
bn_act_apply_q8_kernel(const bf16_t* __restrict__ y, long rows, int C, const float* __restrict__ scale, const float* __restrict__ shift,
                       const bf16_t* __restrict__ res, int res_ldc, int res_coff, bf16_t* __restrict__ z, int z_ldc, int z_coff,
                       unsigned char* __restrict__ q8, const float* __restrict__ qscale, unsigned* __restrict__ amax) {
  const int CG = C / 8;
  const long n = rows * CG;
  const float qs = qscale[0];
  float mx = 0.f;
  for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * EW_THREADS) {
    const long row = i / CG; const int c = (int)(i - row * CG) * 8;
    float f[8], r[8], sc[8], sh[8];
    ys_unpack<bf16_t>(ys_ld16(y + row * C + c), f);
    if (res) ys_unpack<bf16_t>(ys_ld16(res + row * res_ldc + res_coff + c), r);
    ys_ldcoef<8>(scale + c, sc);
    ys_ldcoef<8>(shift + c, sh);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float u = f[e] * sc[e] + sh[e];
      if (ACT) u = ys_silu(u);
      if (res) u += r[e];
      f[e] = u;
    }
    const uint4 pk = ys_pack<bf16_t>(f);
    ys_st16(z + row * z_ldc + z_coff + c, pk);
    ys_unpack<bf16_t>(pk, f);                   // quantise the stored (bf16-rounded) value: what every other reader of z sees
#pragma unroll
    for (int e = 0; e < 8; e++) { mx = fmaxf(mx, fabsf(f[e])); f[e] *= qs; }
    *(uint2*)(q8 + row * C + c) = ys_pack_f8x8<0>(f);
  }
  if (amax) ys_amax_update_wg(amax, mx);
}
int ys_bn_act_apply_q8_launch(hipStream_t st, const void* y, long rows, int C, const float* scale, const float* shift, int act,
                              const void* res, int res_ldc, int res_coff, void* z, int z_ldc, int z_coff, void* q8,
                              const float* qscale, unsigned* amax) {
  const long n = rows * (C / 8);
  const long gcap = 2048;
  long g = ys_cdiv(n, EW_THREADS * 4L); if (g > gcap) g = gcap; if (g < 1) g = 1;
#define BAQ_LAUNCH(AF) YS_LAUNCH((bn_act_apply_q8_kernel<AF>), (int)g, EW_THREADS, st, (const bf16_t*)y, rows, C, scale, shift, (const bf16_t*)res, res_ldc, res_coff, (bf16_t*)z, z_ldc, z_coff, (unsigned char*)q8, qscale, amax)
  if (act) BAQ_LAUNCH(true); else BAQ_LAUNCH(false);
#undef BAQ_LAUNCH
  return YS_OK;
}

int ys_bn_act_apply_launch(hipStream_t st, int dtype, const void* y, long rows, int C, const float* scale,
                           const float* shift, int act, const void* res, int res_ldc, int res_coff, void* z,
                           int z_ldc, int z_coff, unsigned* amax) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const long n = rows * (C / epl);
  const int small = n < (1L << 31) ? 1 : 0;
#define BA_LAUNCH(TT, AF) YS_LAUNCH((bn_act_apply_kernel<TT, AF>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const TT*)y, rows, C, scale, shift, (const TT*)res, res_ldc, res_coff, (TT*)z, z_ldc, z_coff, small, amax)
  if (dtype == YS_BF16) { if (act) BA_LAUNCH(bf16_t, true); else BA_LAUNCH(bf16_t, false); }
  else { if (act) BA_LAUNCH(float, true); else BA_LAUNCH(float, false); }
#undef BA_LAUNCH
  return YS_OK;
}

// ------------------------------------------------------------------ BN finalize + apply in one pass (round 5)
// The convolution left the unit's batch statistics as fixed-point integer sums (ys_stat_acc_add: 64-bit atomics, YS_STAT_SHARDS copies).  Every workgroup adds the
// copies and does bn_finalize_kernel's arithmetic for all C channels itself (same integers in, same double arithmetic: the same coefficients in every workgroup, no
// hand-off between workgroups) into LDS; workgroup 0 also stores scale / shift / mean / rstd for the backward pass and updates the running statistics.  Grid-stride
// with a bounded grid so that the C-channel prologue is paid ~2 thousand times per launch, not once per 256 vectors.  The separate bn_finalize launch (4.9 us + a
// kernel boundary, 45 per YOLOv8n step) goes.
template <class T, bool ACT>
__global__ void __launch_bounds__(EW_THREADS)
bn_fin_apply_kernel(const T* __restrict__ y, long rows, int C, BnAccFin f, const T* __restrict__ res, int res_ldc, int res_coff,
                    T* __restrict__ z, int z_ldc, int z_coff) {
  constexpr int EPL = Elem<T>::EPL;
  YS_DYN_LDS(lds);
  float* s_sc = (float*)lds;                   // [C] scale, [C] shift
  float* s_sh = s_sc + C;
  for (int c = threadIdx.x; c < C; c += EW_THREADS) {
    long long a1 = 0, a2 = 0;
    bool poisoned = false;
#pragma unroll
    for (int s = 0; s < YS_STAT_SHARDS; s++) {
      const unsigned long long w2 = f.acc[((long)s * C + c) * 2 + 1];
      poisoned |= (w2 >> 62) != 0;               // a producer workgroup saw a non-finite / out-of-range partial (ys_stat_acc_add)
      a1 += (long long)f.acc[((long)s * C + c) * 2 + 0];
      a2 += (long long)w2;
    }
    const double s1 = (double)a1 * (1.0 / (double)YS_STAT_FIX), s2 = (double)a2 * (1.0 / (double)YS_STAT_FIX);
    double mean = s1 / f.count;
    double var = s2 / f.count - mean * mean;   // biased (torch BatchNorm2d training normalisation)
    if (var < 0.0) var = 0.0;
    if (poisoned) { mean = __builtin_nan(""); var = mean; }
    const float g = f.gamma[c], bt = f.beta[c];
    const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float sc = g * rstd, sh = bt - (float)mean * g * rstd;
    s_sc[c] = sc; s_sh[c] = sh;
    if (blockIdx.x == 0) {
      f.scale[c] = sc; f.shift[c] = sh; f.mean[c] = (float)mean; f.rstd[c] = rstd;
      // running stats: momentum 0.03, unbiased variance (Convs.cs:41-42,48)
      const double unb = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
      f.run_mean[c] = (1.0f - f.momentum) * f.run_mean[c] + f.momentum * (float)mean;
      f.run_var[c] = (1.0f - f.momentum) * f.run_var[c] + f.momentum * (float)unb;
      if (c == 0 && f.nbt) f.nbt[0] = f.nbt[0] + 1.0f;
    }
  }
  __syncthreads();
  const int CG = C / EPL;
  const long n = rows * CG;
  for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * EW_THREADS) {
    const long row = i / CG; const int c = (int)(i - row * CG) * EPL;
    float v[EPL], r[EPL];
    ys_unpack<T>(ys_ld16(y + row * C + c), v);
    if (res) ys_unpack<T>(ys_ld16(res + row * res_ldc + res_coff + c), r);
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      float u = v[e] * s_sc[c + e] + s_sh[c + e];
      if (ACT) u = ys_silu(u);
      if (res) u += r[e];
      v[e] = u;
    }
    ys_st16(z + row * z_ldc + z_coff + c, ys_pack<T>(v));
  }
}
int ys_bn_fin_apply_launch(hipStream_t st, int dtype, const void* y, long rows, int C, const BnAccFin& f, int act, const void* res, int res_ldc, int res_coff,
                           void* z, int z_ldc, int z_coff) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const long n = rows * (C / epl);
  const long gcap = 2048;
  long g = ys_cdiv(n, EW_THREADS * 2L); if (g > gcap) g = gcap; if (g < 1) g = 1;
  const size_t lds = (size_t)2 * C * sizeof(float);
#define BFA_LAUNCH(TT, AF) YS_LAUNCH_LDS((bn_fin_apply_kernel<TT, AF>), (int)g, EW_THREADS, lds, st, (const TT*)y, rows, C, f, (const TT*)res, res_ldc, res_coff, (TT*)z, z_ldc, z_coff)
  if (dtype == YS_BF16) { if (act) BFA_LAUNCH(bf16_t, true); else BFA_LAUNCH(bf16_t, false); }
  else { if (act) BFA_LAUNCH(float, true); else BFA_LAUNCH(float, false); }
#undef BFA_LAUNCH
  return YS_OK;
}

// ------------------------------------------------------------------ per-channel reductions over rows
// thread t owns channel vector cv = t % CG and row lane t / CG; workgroup blk owns a contiguous row range.
// MODE 0: BN backward (sum du, sum du*xhat), optional res_grad += dz.   MODE 1: plain column sum.
#ifndef CR_U
#define CR_U 2
#endif
template <class T, int MODE, bool RG = false, bool ACT = false>
__global__ void __launch_bounds__(EW_THREADS)
chan_reduce_kernel(const T* __restrict__ dz, int dz_ldc, int dz_coff, const T* __restrict__ y, long rows, int C,
                   const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                   const float* __restrict__ rstd, int act, T* __restrict__ rg, int rg_ldc, int rg_coff,
                   float* __restrict__ partial, long rpb, long bstride) {
  constexpr int EPL = Elem<T>::EPL;
  constexpr int NV = MODE == 1 ? EPL : 2 * EPL;          // running sums per thread
  __shared__ float sAcc[NV][EW_THREADS];                 // [value][thread]: conflict-free for consecutive threads
  const int CG = C / EPL;
  const int RP = EW_THREADS / CG;  // rows per pass
  const int tid = threadIdx.x;
  const int cv = tid % CG, rl = tid / CG;
  const int c = cv * EPL;
  const long rows_per_blk = (rows + gridDim.x - 1) / gridDim.x;
  const long r0 = (long)blockIdx.x * rows_per_blk;
  long r1 = r0 + rows_per_blk;
  if (r1 > rows) r1 = rows;
  float a1[EPL], a2[EPL];
#pragma unroll
  for (int e = 0; e < EPL; e++) { a1[e] = 0.f; a2[e] = 0.f; }
  if (rl < RP) {
    // the second BN-backward sum is accumulated against the raw conv output (sum du*y); chan_finalize turns it into
    // sum du*xhat = rstd * (sum du*y - mean * sum du) in double.  Keeping mean / rstd out of the loop saves 16 VGPRs per
    // thread (occupancy 3 -> 5 waves per SIMD), which is what this latency-bound pass needs.
    float sc[EPL], sh[EPL];
    if (MODE == 0) { ys_ldcoef<EPL>(scale + c, sc); ys_ldcoef<EPL>(shift + c, sh); }
    // Software-pipelined over trips of U rows: the loads of trip t+1 are issued before the arithmetic of trip t.  The SiLU
    // derivative is ~18 VALU issue slots per element, i.e. about as long as the HBM latency of a trip; without the
    // prefetch every wave of a SIMD alternates in step between "all waiting" and "all computing" and the two never overlap.
    constexpr int U = CR_U;
    const bool dense = MODE != 1 || rpb == rows;         // no per-image row stride (always for the BN passes): skip the 64-bit division
    const long step = (long)RP * U;
    auto issue = [&](long rowb, uint4 (&gv)[U], uint4 (&fv)[U], uint4 (&ov)[U]) {
#pragma unroll
      for (int k = 0; k < U; k++) {
        // rows past the end are clamped to the last row and ignored by consume(): loads inside exec-masked branches make the
        // compiler fall back to s_waitcnt vmcnt(0) at the join, which would wait for the prefetch as well
        long row = rowb + (long)k * RP;
        row = row < r1 ? row : r1 - 1;
        long zrow = row;
        if (!dense) { const long zb = row / rpb; zrow = zb * bstride + (row - zb * rpb); }   // dz rows strided per image (head outputs)
        gv[k] = ys_ld16(dz + zrow * dz_ldc + dz_coff + c);
        if (MODE == 0) {
          fv[k] = ys_ld16(y + row * C + c);
          if (RG) ov[k] = ys_ld16(rg + row * rg_ldc + rg_coff + c);
        } else { fv[k] = ys_zero16(); }
        if (!RG) ov[k] = ys_zero16();
      }
    };
    auto consume = [&](long rowb, const uint4 (&gv)[U], const uint4 (&fv)[U], const uint4 (&ov)[U]) {
#pragma unroll
      for (int k = 0; k < U; k++) {
        const long row = rowb + (long)k * RP;
        if (row >= r1) continue;
        float g[EPL];
        ys_unpack<T>(gv[k], g);
        if (MODE == 0) {
          float f[EPL];
          ys_unpack<T>(fv[k], f);
          if (RG) {
            float o[EPL];
            ys_unpack<T>(ov[k], o);
#pragma unroll
            for (int e = 0; e < EPL; e++) o[e] += g[e];
            ys_st16(rg + row * rg_ldc + rg_coff + c, ys_pack<T>(o));
          }
#pragma unroll
          for (int e = 0; e < EPL; e++) {
            const float u = f[e] * sc[e] + sh[e];
            const float du = ACT ? g[e] * ys_silu_grad(u) : g[e];   // compile-time: a run-time flag here splits the unrolled
            a1[e] += du;                                            // elements into basic blocks and serialises the exp/rcp chains
            a2[e] += du * f[e];
          }
        } else if (MODE == 2) {
#pragma unroll
          for (int e = 0; e < EPL; e++) { a1[e] += g[e]; a2[e] += g[e] * g[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < EPL; e++) a1[e] += g[e];
        }
      }
    };
    uint4 gA[U], fA[U], oA[U], gB[U], fB[U], oB[U];
    long rowb = r0 + rl;
    if (rowb < r1) {
      issue(rowb, gA, fA, oA);
      while (true) {                                     // two trips per iteration: the buffers alternate without dynamic indexing
        issue(rowb + step, gB, fB, oB);                  // (unconditional: past the end it re-reads the last row)
        consume(rowb, gA, fA, oA);
        rowb += step;
        if (rowb >= r1) break;
        issue(rowb + step, gA, fA, oA);
        consume(rowb, gB, fB, oB);
        rowb += step;
        if (rowb >= r1) break;
      }
    }
  }
  // workgroup reduction over the RP row lanes: pairwise halving with every thread taking part (the previous form -- CG threads
  // each walking RP rows -- was a serial chain of up to 2048 LDS reads per workgroup for the 16-channel layers)
#pragma unroll
  for (int e = 0; e < EPL; e++) { sAcc[e][tid] = a1[e]; if (MODE != 1) sAcc[EPL + e][tid] = a2[e]; }
  __syncthreads();
  for (int n = RP; n > 1;) {
    const int half = (n + 1) >> 1;
    if (rl + half < n) {
#pragma unroll
      for (int v = 0; v < NV; v++) sAcc[v][tid] += sAcc[v][tid + half * CG];
    }
    __syncthreads();
    n = half;
  }
  if (tid < CG) {
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      partial[((long)blockIdx.x * 2 + 0) * C + c + e] = sAcc[e][tid];
      partial[((long)blockIdx.x * 2 + 1) * C + c + e] = MODE != 1 ? sAcc[(MODE != 1 ? EPL : 0) + e][tid] : 0.f;
    }
  }
}

static int reduce_blocks(long rows, int C, int epl) {
  const int cg = C / epl;
  const int rp = EW_THREADS / cg;
  // prefer 16 row passes per workgroup, but keep >= 512 workgroups (2 per CU) while a workgroup still has a few trips of
  // work; at most 768 (3 per CU, one resident round at 4-5 waves per SIMD): more partial rows only lengthen the finalize
  // kernel and add a ragged second round (measured: 2048 -> 12.86, 1024 -> 12.77, 768/512 -> 12.75 ms/step)
  const long maxnb = 768, minnb = 512;
  long passes = 16;
  while (passes > 4 && (rows + (long)rp * passes - 1) / ((long)rp * passes) < minnb) passes >>= 1;
  long nb = (rows + (long)rp * passes - 1) / ((long)rp * passes);
  if (nb > maxnb) nb = maxnb;
  if (nb < 1) nb = 1;
  return (int)nb;
}
int ys_colsum_blocks(long rows, int C, int dtype) { return reduce_blocks(rows, C, dtype == YS_BF16 ? 8 : 4); }

int ys_bn_bwd_reduce_launch(hipStream_t st, int dtype, const void* dz, int dz_ldc, int dz_coff, const void* y, long rows,
                            int C, const float* scale, const float* shift, const float* mean, const float* rstd,
                            int act, void* res_grad, int rg_ldc, int rg_coff, float* partial, int* nblk_out) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  if (C % epl || C / epl > EW_THREADS) { ys_set_error("bn_bwd: unsupported channel count %d", C); return YS_ERR_UNSUPPORTED; }
  const int nb = reduce_blocks(rows, C, epl);
  *nblk_out = nb;
#define CR_LAUNCH(TT, RGF, ACTF) \
  YS_LAUNCH((chan_reduce_kernel<TT, 0, RGF, ACTF>), nb, EW_THREADS, st, (const TT*)dz, dz_ldc, dz_coff, (const TT*)y, rows, C, scale, shift, mean, rstd, act, (TT*)res_grad, rg_ldc, rg_coff, partial, rows, rows)
  const int variant = (dtype == YS_BF16 ? 4 : 0) | (res_grad ? 2 : 0) | (act ? 1 : 0);
  switch (variant) {
    case 0: CR_LAUNCH(float, false, false); break;
    case 1: CR_LAUNCH(float, false, true); break;
    case 2: CR_LAUNCH(float, true, false); break;
    case 3: CR_LAUNCH(float, true, true); break;
    case 4: CR_LAUNCH(bf16_t, false, false); break;
    case 5: CR_LAUNCH(bf16_t, false, true); break;
    case 6: CR_LAUNCH(bf16_t, true, false); break;
    default: CR_LAUNCH(bf16_t, true, true); break;
  }
#undef CR_LAUNCH
  return YS_OK;
}

// per-channel (sum, sum of squares) partials of a dense [rows][C] tensor (BN statistics of depthwise conv outputs)
int ys_chan_stats_launch(hipStream_t st, int dtype, const void* y, long rows, int C, float* partial, int* nblk_out) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  if (C % epl || C / epl > EW_THREADS) { ys_set_error("chan_stats: unsupported channel count %d", C); return YS_ERR_UNSUPPORTED; }
  const int nb = reduce_blocks(rows, C, epl);
  *nblk_out = nb;
  if (dtype == YS_BF16)
    YS_LAUNCH((chan_reduce_kernel<bf16_t, 2>), nb, EW_THREADS, st, (const bf16_t*)y, C, 0, (const bf16_t*)nullptr, rows, C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, (bf16_t*)nullptr, 0, 0, partial, rows, rows);
  else
    YS_LAUNCH((chan_reduce_kernel<float, 2>), nb, EW_THREADS, st, (const float*)y, C, 0, (const float*)nullptr, rows, C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, (float*)nullptr, 0, 0, partial, rows, rows);
  return YS_OK;
}

// one workgroup per channel: sums partials; MODE 0 -> BN grads + coefficients, MODE 1 -> bias grad.
// The partial rows of a channel come from the source that covers it (FinSrc): one source = the channel-reduction pass; several =
// the dgrad launches whose epilogues produced the sums (fused BN-backward reduction, BnRedSeg), each with its own row count.
template <int MODE>
__global__ void __launch_bounds__(EW_THREADS)
chan_finalize_kernel(FinSrc src, int C, double count, float* __restrict__ g0,
                     float* __restrict__ g1, float* __restrict__ c1, float* __restrict__ c2,
                     const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ rstd) {
  __shared__ double sbuf[EW_THREADS];
  const int c = blockIdx.x;
  const float* __restrict__ partial = src.p[0]; int nblk = src.nblk[0];
#pragma unroll
  for (int k = 1; k < YS_BNRED_MAXSEG; k++) if (k < src.n && c >= src.c1[k - 1]) { partial = src.p[k]; nblk = src.nblk[k]; }
  float sc = 0.f, mu = 0.f, rs = 0.f, g0p = 0.f, g1p = 0.f;   // tail operands first (see bn_finalize_kernel)
  if (threadIdx.x == 0) {
    g0p = g0[c];
    if (MODE == 0) { sc = scale[c]; mu = mean[c]; rs = rstd[c]; g1p = g1[c]; }
  }
  double s1 = 0.0, s2 = 0.0;
  for (int k = threadIdx.x; k < nblk; k += EW_THREADS) {
    s1 += (double)partial[((long)k * 2 + 0) * C + c];
    if (MODE == 0) s2 += (double)partial[((long)k * 2 + 1) * C + c];
  }
  block_sum2_d(s1, s2, sbuf);
  if (threadIdx.x == 0) {
    if (MODE == 0) {
      s2 = (double)rs * (s2 - (double)mu * s1);   // sum(du * xhat) from sum(du * y) (chan_reduce_kernel)
      g0[c] = g0p + (float)s2;  // dgamma += sum(du * xhat)
      g1[c] = g1p + (float)s1;  // dbeta  += sum(du)
      // dy = gamma*rstd*(du - m1 - xhat*m2) = scale*du - k2 - y*k3  (m1 = mean(du), m2 = mean(du*xhat))
      const float m1 = (float)(s1 / count), m2 = (float)(s2 / count);
      c1[c] = sc * (m1 - mu * rs * m2);   // k2
      c2[c] = sc * rs * m2;               // k3
    } else {
      g0[c] = g0p + (float)s1;
    }
  }
}
int ys_bn_bwd_finalize_src_launch(hipStream_t st, const FinSrc& src, int C, long count, float* dgamma, float* dbeta, float* c1,
                                  float* c2, const float* scale, const float* mean, const float* rstd) {
  YS_LAUNCH((chan_finalize_kernel<0>), C, EW_THREADS, st, src, C, (double)count, dgamma, dbeta, c1, c2, scale, mean, rstd);
  return YS_OK;
}
int ys_bn_bwd_finalize_launch(hipStream_t st, const float* partial, int nblk, int C, long count, float* dgamma,
                              float* dbeta, float* c1, float* c2, const float* scale, const float* mean, const float* rstd) {
  FinSrc src{}; src.n = 1; src.p[0] = partial; src.nblk[0] = nblk; src.c1[0] = C;
  return ys_bn_bwd_finalize_src_launch(st, src, C, count, dgamma, dbeta, c1, c2, scale, mean, rstd);
}

int ys_colsum_launch(hipStream_t st, int dtype, const void* x, int ldc, int coff, long rows, long rows_per_b,
                     long bstride, int C, float* partial, float* grad) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const int Cp = (C + epl - 1) / epl * epl;  // padded channels of the view are zero
  if (Cp / epl > EW_THREADS) { ys_set_error("colsum: unsupported channel count %d", C); return YS_ERR_UNSUPPORTED; }
  const int nb = reduce_blocks(rows, Cp, epl);
  if (dtype == YS_BF16)
    YS_LAUNCH((chan_reduce_kernel<bf16_t, 1>), nb, EW_THREADS, st, (const bf16_t*)x, ldc, coff, (const bf16_t*)nullptr, rows, Cp, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, (bf16_t*)nullptr, 0, 0, partial, rows_per_b, bstride);
  else
    YS_LAUNCH((chan_reduce_kernel<float, 1>), nb, EW_THREADS, st, (const float*)x, ldc, coff, (const float*)nullptr, rows, Cp, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, (float*)nullptr, 0, 0, partial, rows_per_b, bstride);
  // partial rows are Cp wide; finalize only the C real channels
  FinSrc src{}; src.n = 1; src.p[0] = partial; src.nblk[0] = nb; src.c1[0] = Cp;
  YS_LAUNCH((chan_finalize_kernel<1>), C, EW_THREADS, st, src, Cp, 1.0, grad, (float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
  return YS_OK;
}

template <class T, bool ACT>
__global__ void __launch_bounds__(EW_THREADS)
bn_bwd_apply_kernel(const T* __restrict__ dz, int dz_ldc, int dz_coff, const T* __restrict__ y, long rows, int C,
                    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ k2,
                    const float* __restrict__ k3, T* __restrict__ dy, int small, unsigned* __restrict__ amax,
                    T* __restrict__ rg, int rg_ldc, int rg_coff) {
  constexpr int EPL = Elem<T>::EPL;
  const int CG = C / EPL;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float mx = 0.f;
  if (i < rows * CG) {
    long row; int c;
    if (small) { const unsigned iu = (unsigned)i, r = iu / (unsigned)CG; row = r; c = (int)(iu - r * (unsigned)CG) * EPL; }
    else { row = i / CG; c = (int)(i - row * CG) * EPL; }
    float g[EPL], f[EPL], sc[EPL], sh[EPL], a2[EPL], a3[EPL];
    ys_unpack<T>(ys_ld16(dz + row * dz_ldc + dz_coff + c), g);
    ys_unpack<T>(ys_ld16(y + row * C + c), f);
    if (rg) {     // d(residual input) += dz: the Bottleneck shortcut's share, done here when the reduction pass that used to carry it is fused away
      float o[EPL];
      T* rp = rg + row * rg_ldc + rg_coff + c;
      ys_unpack<T>(ys_ld16(rp), o);
#pragma unroll
      for (int e = 0; e < EPL; e++) o[e] += g[e];
      ys_st16(rp, ys_pack<T>(o));
    }
    ys_ldcoef<EPL>(scale + c, sc); ys_ldcoef<EPL>(shift + c, sh); ys_ldcoef<EPL>(k2 + c, a2); ys_ldcoef<EPL>(k3 + c, a3);
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      const float u = f[e] * sc[e] + sh[e];
      const float du = ACT ? g[e] * ys_silu_grad(u) : g[e];
      f[e] = sc[e] * du - a2[e] - f[e] * a3[e];
      mx = fmaxf(mx, fabsf(f[e]));
    }
    ys_st16(dy + row * C + c, ys_pack<T>(f));
  }
  if (amax) ys_amax_update(amax, mx);          // fp8 mode: amax(|dy|) for the next step's gradient scale (f8.hip)
}
// fp8 mode, bf16 storage: the same pass also writes the e5m2 image of dy (dense [rows][C], scaled by the layer's delayed gradient
// scale) that the fp8 dgrad kernel consumes, and records amax(|dy|) for the next step's scale -- instead of a separate quantisation
// pass over dy (f8_quant_view_kernel: 2 B read + 1 B written per element and layer).  Grid-stride with a bounded grid so that the
// amax bookkeeping is one atomic per workgroup (one-shot waves each paying a dependent global round trip were measured 5x slower).
template <bool ACT>
__global__ void __launch_bounds__(EW_THREADS)
bn_bwd_apply_q8_kernel(const bf16_t* __restrict__ dz, int dz_ldc, int dz_coff, const bf16_t* __restrict__ y, long rows, int C,
                       const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ k2,
                       const float* __restrict__ k3, bf16_t* __restrict__ dy, unsigned char* __restrict__ q8,
                       const float* __restrict__ qscale, unsigned* __restrict__ amax, bf16_t* __restrict__ rg, int rg_ldc, int rg_coff) {
  const int CG = C / 8;
  const long n = rows * CG;
  const float qs = qscale[0];
  float mx = 0.f;
  for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * EW_THREADS) {
    const long row = i / CG; const int c = (int)(i - row * CG) * 8;
    float g[8], f[8], sc[8], sh[8], a2[8], a3[8];
    ys_unpack<bf16_t>(ys_ld16(dz + row * dz_ldc + dz_coff + c), g);
    ys_unpack<bf16_t>(ys_ld16(y + row * C + c), f);
    if (rg) {
      float o[8];
      bf16_t* rp = rg + row * rg_ldc + rg_coff + c;
      ys_unpack<bf16_t>(ys_ld16(rp), o);
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] += g[e];
      ys_st16(rp, ys_pack<bf16_t>(o));
    }
    ys_ldcoef<8>(scale + c, sc); ys_ldcoef<8>(shift + c, sh); ys_ldcoef<8>(k2 + c, a2); ys_ldcoef<8>(k3 + c, a3);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float u = f[e] * sc[e] + sh[e];
      const float du = ACT ? g[e] * ys_silu_grad(u) : g[e];
      f[e] = sc[e] * du - a2[e] - f[e] * a3[e];
    }
    const uint4 pk = ys_pack<bf16_t>(f);
    ys_st16(dy + row * C + c, pk);
    ys_unpack<bf16_t>(pk, f);                   // quantise what the bf16 consumers (wgrad) see: the rounded value
#pragma unroll
    for (int e = 0; e < 8; e++) { mx = fmaxf(mx, fabsf(f[e])); f[e] *= qs; }
    *(uint2*)(q8 + row * C + c) = ys_pack_f8x8<1>(f);
  }
  if (amax) ys_amax_update_wg(amax, mx);
}
int ys_bn_bwd_apply_q8_launch(hipStream_t st, const void* dz, int dz_ldc, int dz_coff, const void* y, long rows, int C,
                              const float* scale, const float* shift, const float* k2, const float* k3, int act, void* dy,
                              void* q8, const float* qscale, unsigned* amax, void* rg, int rg_ldc, int rg_coff) {
  const long n = rows * (C / 8);
  const long gcap = 2048;
  long g = ys_cdiv(n, EW_THREADS * 4L); if (g > gcap) g = gcap; if (g < 1) g = 1;
#define BQ_LAUNCH(AF) YS_LAUNCH((bn_bwd_apply_q8_kernel<AF>), (int)g, EW_THREADS, st, (const bf16_t*)dz, dz_ldc, dz_coff, (const bf16_t*)y, rows, C, scale, shift, k2, k3, (bf16_t*)dy, (unsigned char*)q8, qscale, amax, (bf16_t*)rg, rg_ldc, rg_coff)
  if (act) BQ_LAUNCH(true); else BQ_LAUNCH(false);
#undef BQ_LAUNCH
  return YS_OK;
}

int ys_bn_bwd_apply_launch(hipStream_t st, int dtype, const void* dz, int dz_ldc, int dz_coff, const void* y, long rows,
                           int C, const float* scale, const float* shift, const float* k2, const float* k3, int act, void* dy,
                           unsigned* amax, void* rg, int rg_ldc, int rg_coff) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const long n = rows * (C / epl);
  const int small = n < (1L << 31) ? 1 : 0;
#define BB_LAUNCH(TT, AF) YS_LAUNCH((bn_bwd_apply_kernel<TT, AF>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const TT*)dz, dz_ldc, dz_coff, (const TT*)y, rows, C, scale, shift, k2, k3, (TT*)dy, small, amax, (TT*)rg, rg_ldc, rg_coff)
  if (dtype == YS_BF16) { if (act) BB_LAUNCH(bf16_t, true); else BB_LAUNCH(bf16_t, false); }
  else { if (act) BB_LAUNCH(float, true); else BB_LAUNCH(float, false); }
#undef BB_LAUNCH
  return YS_OK;
}

// ------------------------------------------------------------------ max-pool 5x5 s1 p2
// argmax = first maximum in (kh,kw) scan order with strict '>' (ATen max_pool2d CPU semantics)
template <class T>
__global__ void __launch_bounds__(EW_THREADS)
maxpool5_fwd_kernel(const T* __restrict__ x, int x_ldc, int x_coff, int B, int H, int W, int C, T* __restrict__ y,
                    int y_ldc, int y_coff, unsigned char* __restrict__ amax) {
  constexpr int EPL = Elem<T>::EPL;
  const int CG = C / EPL;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * H * W * CG;
  if (i >= n) return;
  const int c = (int)(i % CG) * EPL;
  const long pix = i / CG;
  const int w = (int)(pix % W);
  const int h = (int)((pix / W) % H);
  const long b = pix / ((long)W * H);
  float best[EPL];
  int bi[EPL];
#pragma unroll
  for (int e = 0; e < EPL; e++) { best[e] = -INFINITY; bi[e] = 0; }
  bool first = true;
  for (int kh = 0; kh < 5; kh++) {
    const int ih = h + kh - 2;
    if (ih < 0 || ih >= H) continue;
    for (int kw = 0; kw < 5; kw++) {
      const int iw = w + kw - 2;
      if (iw < 0 || iw >= W) continue;
      float f[EPL];
      ys_unpack<T>(ys_ld16(x + ((b * H + ih) * W + iw) * x_ldc + x_coff + c), f);
#pragma unroll
      for (int e = 0; e < EPL; e++)
        if (first || f[e] > best[e]) { best[e] = f[e]; bi[e] = kh * 5 + kw; }
      first = false;
    }
  }
  ys_st16(y + pix * y_ldc + y_coff + c, ys_pack<T>(best));
  if (amax) {
#pragma unroll
    for (int e = 0; e < EPL; e++) amax[pix * C + c + e] = (unsigned char)bi[e];
  }
}
int ys_maxpool5_fwd_launch(hipStream_t st, int dtype, const void* x, int x_ldc, int x_coff, int B, int H, int W, int C,
                           void* y, int y_ldc, int y_coff, unsigned char* argmax) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const long n = (long)B * H * W * (C / epl);
  if (dtype == YS_BF16)
    YS_LAUNCH((maxpool5_fwd_kernel<bf16_t>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const bf16_t*)x, x_ldc, x_coff, B, H, W, C, (bf16_t*)y, y_ldc, y_coff, argmax);
  else
    YS_LAUNCH((maxpool5_fwd_kernel<float>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const float*)x, x_ldc, x_coff, B, H, W, C, (float*)y, y_ldc, y_coff, argmax);
  return YS_OK;
}

template <class T>
__global__ void __launch_bounds__(EW_THREADS)
maxpool5_bwd_kernel(const T* __restrict__ dy, int dy_ldc, int dy_coff, int B, int H, int W, int C,
                    const unsigned char* __restrict__ amax, T* __restrict__ dx, int dx_ldc, int dx_coff, int accumulate) {
  constexpr int EPL = Elem<T>::EPL;
  const int CG = C / EPL;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * H * W * CG;
  if (i >= n) return;
  const int c = (int)(i % CG) * EPL;
  const long pix = i / CG;
  const int w = (int)(pix % W);
  const int h = (int)((pix / W) % H);
  const long b = pix / ((long)W * H);
  float acc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; e++) acc[e] = 0.f;
  for (int kh = 0; kh < 5; kh++) {
    const int oh = h - kh + 2;  // output whose window holds (h,w) at offset (kh,kw)
    if (oh < 0 || oh >= H) continue;
    for (int kw = 0; kw < 5; kw++) {
      const int ow = w - kw + 2;
      if (ow < 0 || ow >= W) continue;
      const long op = (b * H + oh) * W + ow;
      float g[EPL];
      ys_unpack<T>(ys_ld16(dy + op * dy_ldc + dy_coff + c), g);
      const int code = kh * 5 + kw;
#pragma unroll
      for (int e = 0; e < EPL; e++)
        if (amax[op * C + c + e] == code) acc[e] += g[e];
    }
  }
  T* dp = dx + pix * dx_ldc + dx_coff + c;
  if (accumulate) {
    float o[EPL];
    ys_unpack<T>(ys_ld16(dp), o);
#pragma unroll
    for (int e = 0; e < EPL; e++) acc[e] += o[e];
  }
  ys_st16(dp, ys_pack<T>(acc));
}
int ys_maxpool5_bwd_launch(hipStream_t st, int dtype, const void* dy, int dy_ldc, int dy_coff, int B, int H, int W,
                           int C, const unsigned char* argmax, void* dx, int dx_ldc, int dx_coff, int accumulate) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const long n = (long)B * H * W * (C / epl);
  if (dtype == YS_BF16)
    YS_LAUNCH((maxpool5_bwd_kernel<bf16_t>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const bf16_t*)dy, dy_ldc, dy_coff, B, H, W, C, argmax, (bf16_t*)dx, dx_ldc, dx_coff, accumulate);
  else
    YS_LAUNCH((maxpool5_bwd_kernel<float>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const float*)dy, dy_ldc, dy_coff, B, H, W, C, argmax, (float*)dx, dx_ldc, dx_coff, accumulate);
  return YS_OK;
}

// ------------------------------------------------------------------ nearest 2x upsample
template <class T>
__global__ void __launch_bounds__(EW_THREADS)
upsample2x_fwd_kernel(const T* __restrict__ x, int x_ldc, int x_coff, int B, int H, int W, int C, T* __restrict__ y,
                      int y_ldc, int y_coff) {
  constexpr int EPL = Elem<T>::EPL;
  const int CG = C / EPL;
  const int OH = 2 * H, OW = 2 * W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * OH * OW * CG;
  if (i >= n) return;
  const int c = (int)(i % CG) * EPL;
  const long pix = i / CG;
  const int ow = (int)(pix % OW);
  const int oh = (int)((pix / OW) % OH);
  const long b = pix / ((long)OW * OH);
  const uint4 v = ys_ld16(x + ((b * H + (oh >> 1)) * W + (ow >> 1)) * x_ldc + x_coff + c);
  ys_st16(y + pix * y_ldc + y_coff + c, v);
}
int ys_upsample2x_fwd_launch(hipStream_t st, int dtype, const void* x, int x_ldc, int x_coff, int B, int H, int W, int C,
                             void* y, int y_ldc, int y_coff) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const long n = (long)B * 4 * H * W * (C / epl);
  if (dtype == YS_BF16)
    YS_LAUNCH((upsample2x_fwd_kernel<bf16_t>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const bf16_t*)x, x_ldc, x_coff, B, H, W, C, (bf16_t*)y, y_ldc, y_coff);
  else
    YS_LAUNCH((upsample2x_fwd_kernel<float>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const float*)x, x_ldc, x_coff, B, H, W, C, (float*)y, y_ldc, y_coff);
  return YS_OK;
}

template <class T>
__global__ void __launch_bounds__(EW_THREADS)
upsample2x_bwd_kernel(const T* __restrict__ dy, int dy_ldc, int dy_coff, int B, int H, int W, int C, T* __restrict__ dx,
                      int dx_ldc, int dx_coff, int accumulate) {
  constexpr int EPL = Elem<T>::EPL;
  const int CG = C / EPL;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)B * H * W * CG;
  if (i >= n) return;
  const int c = (int)(i % CG) * EPL;
  const long pix = i / CG;
  const int w = (int)(pix % W);
  const int h = (int)((pix / W) % H);
  const long b = pix / ((long)W * H);
  float acc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; e++) acc[e] = 0.f;
  for (int dh = 0; dh < 2; dh++)
    for (int dw = 0; dw < 2; dw++) {
      float g[EPL];
      ys_unpack<T>(ys_ld16(dy + ((b * 2 * H + 2 * h + dh) * 2 * W + 2 * w + dw) * dy_ldc + dy_coff + c), g);
#pragma unroll
      for (int e = 0; e < EPL; e++) acc[e] += g[e];
    }
  T* dp = dx + pix * dx_ldc + dx_coff + c;
  if (accumulate) {
    float o[EPL];
    ys_unpack<T>(ys_ld16(dp), o);
#pragma unroll
    for (int e = 0; e < EPL; e++) acc[e] += o[e];
  }
  ys_st16(dp, ys_pack<T>(acc));
}
int ys_upsample2x_bwd_launch(hipStream_t st, int dtype, const void* dy, int dy_ldc, int dy_coff, int B, int H, int W,
                             int C, void* dx, int dx_ldc, int dx_coff, int accumulate) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const long n = (long)B * H * W * (C / epl);
  if (dtype == YS_BF16)
    YS_LAUNCH((upsample2x_bwd_kernel<bf16_t>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const bf16_t*)dy, dy_ldc, dy_coff, B, H, W, C, (bf16_t*)dx, dx_ldc, dx_coff, accumulate);
  else
    YS_LAUNCH((upsample2x_bwd_kernel<float>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const float*)dy, dy_ldc, dy_coff, B, H, W, C, (float*)dx, dx_ldc, dx_coff, accumulate);
  return YS_OK;
}

// ------------------------------------------------------------------ view copy / accumulate
template <class T>
__global__ void __launch_bounds__(EW_THREADS)
copy_view_kernel(const T* __restrict__ src, int s_ldc, int s_coff, long rows, int C, T* __restrict__ dst, int d_ldc,
                 int d_coff, int accumulate) {
  constexpr int EPL = Elem<T>::EPL;
  const int CG = C / EPL;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * CG) return;
  const long row = i / CG;
  const int c = (int)(i - row * CG) * EPL;
  uint4 v = ys_ld16(src + row * s_ldc + s_coff + c);
  T* dp = dst + row * d_ldc + d_coff + c;
  if (accumulate) {
    float a[EPL], o[EPL];
    ys_unpack<T>(v, a);
    ys_unpack<T>(ys_ld16(dp), o);
#pragma unroll
    for (int e = 0; e < EPL; e++) a[e] += o[e];
    v = ys_pack<T>(a);
  }
  ys_st16(dp, v);
}
int ys_copy_view_launch(hipStream_t st, int dtype, const void* src, int s_ldc, int s_coff, long rows, int C, void* dst,
                        int d_ldc, int d_coff, int accumulate) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const long n = rows * (C / epl);
  if (dtype == YS_BF16)
    YS_LAUNCH((copy_view_kernel<bf16_t>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const bf16_t*)src, s_ldc, s_coff, rows, C, (bf16_t*)dst, d_ldc, d_coff, accumulate);
  else
    YS_LAUNCH((copy_view_kernel<float>), ys_cdiv(n, EW_THREADS), EW_THREADS, st, (const float*)src, s_ldc, s_coff, rows, C, (float*)dst, d_ldc, d_coff, accumulate);
  return YS_OK;
}

// ------------------------------------------------------------------ AdamW (torch.optim.AdamW single-tensor maths)
__global__ void __launch_bounds__(EW_THREADS)
adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
             float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  float pi = p[i] * (1.0f - lr * wd);           // param.mul_(1 - lr*weight_decay)
  const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);  // exp_avg.lerp_(grad, 1-beta1)
  const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi; m[i] = mi; v[i] = vi;
}
int ys_adamw_launch(hipStream_t st, float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                    float beta2, float eps, float wd, float bc1, float bc2) {
  if (n <= 0) return YS_OK;
  YS_LAUNCH(adamw_kernel, ys_cdiv(n, EW_THREADS), EW_THREADS, st, p, g, m, v, n, lr, beta1, beta2, eps, wd, bc1, sqrtf(bc2));
  return YS_OK;
}

// All parameter groups in one launch: the flat parameter vector is a few contiguous [offset, count) ranges (segment x group,
// model.hip layout_params), each with its own learning rate.
__global__ void __launch_bounds__(EW_THREADS)
adamw_ranges_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                    AdamwRanges rg, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt, AdamwDup dup) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float lr = 0.f; bool hit = false;
#pragma unroll
  for (int k = 0; k < YS_ADAMW_MAX_RANGES; k++)
    if (k < rg.n && i >= rg.off[k] && i < rg.off[k] + rg.count[k]) { lr = rg.lr[k]; hit = true; }
  if (!hit) return;
  const float gi = g[i];
  // "reference" parameter groups (YoloBaseTaskModel.cs:144-151 as written): a BatchNorm parameter sits in two groups that share
  // one optimizer state, so one optimizer.step() applies two consecutive AdamW updates with the same gradient -- the second with
  // the bn group's learning rate -- and its step counter advances by two (bias corrections of steps 2t-1 and 2t)
  const bool twice = dup.mask != nullptr && dup.mask[i] != 0;
  const float b1 = twice ? dup.bc1_first : bc1, b2s = twice ? dup.bc2s_first : bc2_sqrt;
  float pi = p[i] * (1.0f - lr * wd);
  float mi = m[i] + (gi - m[i]) * (1.0f - beta1);
  float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
  float denom = sqrtf(vi) / b2s + eps;
  pi -= (lr / b1) * (mi / denom);
  if (twice) {
    pi *= (1.0f - dup.lr_second * wd);
    mi = mi + (gi - mi) * (1.0f - beta1);
    vi = beta2 * vi + (1.0f - beta2) * gi * gi;
    denom = sqrtf(vi) / dup.bc2s_second + eps;
    pi -= (dup.lr_second / dup.bc1_second) * (mi / denom);
  }
  p[i] = pi; m[i] = mi; v[i] = vi;
}
int ys_adamw_ranges_launch(hipStream_t st, float* p, const float* g, float* m, float* v, long n, const AdamwRanges& rg,
                           float beta1, float beta2, float eps, float wd, float bc1, float bc2, const AdamwDup* dup) {
  if (n <= 0 || rg.n <= 0) return YS_OK;
  AdamwDup d{};
  if (dup) d = *dup;
  YS_LAUNCH(adamw_ranges_kernel, ys_cdiv(n, EW_THREADS), EW_THREADS, st, p, g, m, v, n, rg, beta1, beta2, eps, wd, bc1, sqrtf(bc2), d);
  return YS_OK;
}

__global__ void __launch_bounds__(EW_THREADS) fill_kernel(float* p, long n, float v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
int ys_fill_launch(hipStream_t st, float* p, long n, float v) {
  if (n <= 0) return YS_OK;
  YS_LAUNCH(fill_kernel, ys_cdiv(n, EW_THREADS), EW_THREADS, st, p, n, v);
  return YS_OK;
}

// ------------------------------------------------------------------ Detect._inference decode
// pred[b, 0:4, a] = dist2bbox(DFL(boxes), anchors, xywh) * stride ; pred[b, 4:4+nc, a] = sigmoid(scores)
// (Head.cs:204-223; DFL = softmax over reg_max bins, expectation with weights 0..reg_max-1, Block.cs:40-45)
#define DEC_ANCH 64     // anchors per workgroup
template <class T>
__global__ void __launch_bounds__(EW_THREADS)
detect_decode_kernel(const T* __restrict__ pd, int ld_pd, const T* __restrict__ ps, int ld_ps, int B, int A, int nc,
                     int reg_max, int nl, int o0, int o1, int o2, int w0, int w1, int w2, int s0, int s1, int s2,
                     float* __restrict__ pred, int pred_C, const T* __restrict__ px, int ld_px, int xkind, int nx, int kdim) {
  // One workgroup decodes DEC_ANCH consecutive rows of the [B*A][ld] head outputs: the rows are staged in LDS with
  // coalesced 16-byte loads, (anchor, side) threads take the DFL expectation, and the class probabilities are written
  // channel-major so that consecutive lanes store consecutive anchors of pred[b][c][:].
  constexpr int EPL = Elem<T>::EPL;
  YS_DYN_LDS(lds);
  const int nb = 4 * reg_max;                       // box logits per anchor
  const int pb = nb + 1, pc = nc | 1;               // odd LDS pitches (floats): conflict-free column reads
  float* sB = (float*)lds;                          // [DEC_ANCH][pb]
  float* sC = sB + DEC_ANCH * pb;                   // [DEC_ANCH][pc]
  float* sD = sC + DEC_ANCH * pc;                   // [DEC_ANCH][4]
  const int px_p = nx | 1;
  float* sX = sD + DEC_ANCH * 4;                    // [DEC_ANCH][px_p]: the Obb angle logit / the Pose keypoint outputs
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * DEC_ANCH;
  const long rows = (long)B * A;
  const int ub = (nb + EPL - 1) / EPL, uc = (nc + EPL - 1) / EPL;
  for (int idx = tid; idx < DEC_ANCH * ub; idx += EW_THREADS) {
    const int r = idx / ub, u = idx - r * ub;
    if (r0 + r < rows) {
      float f[EPL];
      ys_unpack<T>(ys_ld16(pd + (r0 + r) * ld_pd + u * EPL), f);
#pragma unroll
      for (int e = 0; e < EPL; e++) if (u * EPL + e < nb) sB[r * pb + u * EPL + e] = f[e];
    }
  }
  for (int idx = tid; idx < DEC_ANCH * uc; idx += EW_THREADS) {
    const int r = idx / uc, u = idx - r * uc;
    if (r0 + r < rows) {
      float f[EPL];
      ys_unpack<T>(ys_ld16(ps + (r0 + r) * ld_ps + u * EPL), f);
#pragma unroll
      for (int e = 0; e < EPL; e++) if (u * EPL + e < nc) sC[r * pc + u * EPL + e] = f[e];
    }
  }
  if (xkind) {
    const int ux = (nx + EPL - 1) / EPL;
    for (int idx = tid; idx < DEC_ANCH * ux; idx += EW_THREADS) {
      const int r = idx / ux, u = idx - r * ux;
      if (r0 + r < rows) {
        float f[EPL];
        ys_unpack<T>(ys_ld16(px + (r0 + r) * ld_px + u * EPL), f);
#pragma unroll
        for (int e = 0; e < EPL; e++) if (u * EPL + e < nx) sX[r * px_p + u * EPL + e] = f[e];
      }
    }
  }
  __syncthreads();
  {   // DFL expectation: thread = (anchor, side)
    const int r = tid >> 2, sd = tid & 3;
    if (r < DEC_ANCH && r0 + r < rows) {
      const float* row = sB + r * pb + sd * reg_max;
      float mx = -INFINITY;
      for (int j = 0; j < reg_max; j++) mx = fmaxf(mx, row[j]);
      float se = 0.f, sw = 0.f;
      for (int j = 0; j < reg_max; j++) {
        const float e = __expf(row[j] - mx);
        se += e;
        sw += e * (float)j;
      }
      sD[r * 4 + sd] = sw / se;
    }
  }
  __syncthreads();
  if (tid < DEC_ANCH && r0 + tid < rows) {
    const long i = r0 + tid;
    const int a = (int)(i % A);
    const long b = i / A;
    int lo = o0, lw = w0, ls = s0;
    if (nl > 1 && a >= o1) { lo = o1; lw = w1; ls = s1; }
    if (nl > 2 && a >= o2) { lo = o2; lw = w2; ls = s2; }
    const int cell = a - lo;
    const float ax = (float)(cell % lw) + 0.5f, ay = (float)(cell / lw) + 0.5f;
    const float* d = sD + tid * 4;
    const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
    float* o = pred + b * (long)pred_C * A + a;
    const float st = (float)ls;
    if (xkind == 2) {
      // Obb (Head.cs:429,435-438): angle = (sigmoid(raw) - 0.25) * pi; boxes = dist2rbox (Tal.cs:389-408) * stride;
      // the angle is appended after the class probabilities (Head.cs:411-418)
      const float ang = (ys_sigmoid(sX[tid * px_p]) - 0.25f) * 3.14159265358979323846f;
      const float cs = cosf(ang), sn = sinf(ang);
      const float xf = (d[2] - d[0]) / 2.0f, yf = (d[3] - d[1]) / 2.0f;
      o[0] = (xf * cs - yf * sn + ax) * st;
      o[(long)A] = (xf * sn + yf * cs + ay) * st;
      o[2 * (long)A] = (d[0] + d[2]) * st;
      o[3 * (long)A] = (d[1] + d[3]) * st;
      o[(long)(4 + nc) * A] = ang;
    } else {
      o[0] = (x1 + x2) / 2.0f * st;
      o[(long)A] = (y1 + y2) / 2.0f * st;
      o[2 * (long)A] = (x2 - x1) * st;
      o[3 * (long)A] = (y2 - y1) * st;
    }
  }
  if (xkind == 3) {   // Pose.kpts_decode (Head.cs:590-605): lane = anchor, the 4 waves stride over the keypoint channels
    const int r = tid & (DEC_ANCH - 1);
    if (r0 + r < rows) {
      const long i = r0 + r;
      const int a = (int)(i % A);
      const long b = i / A;
      int lo = o0, lw = w0, ls = s0;
      if (nl > 1 && a >= o1) { lo = o1; lw = w1; ls = s1; }
      if (nl > 2 && a >= o2) { lo = o2; lw = w2; ls = s2; }
      const int cell = a - lo;
      const float gx = (float)(cell % lw), gy = (float)(cell / lw), st = (float)ls;   // anchors - 0.5
      float* o = pred + b * (long)pred_C * A + (long)(4 + nc) * A + a;
      for (int c = tid / DEC_ANCH; c < nx; c += EW_THREADS / DEC_ANCH) {
        const float v = sX[r * px_p + c];
        const int k = c % kdim;
        o[(long)c * A] = k == 0 ? (v * 2.0f + gx) * st : k == 1 ? (v * 2.0f + gy) * st : k == 2 && kdim == 3 ? ys_sigmoid(v) : v;
      }
    }
  }
  {   // class probabilities: lane = anchor, the 4 waves stride over the classes
    const int r = tid & (DEC_ANCH - 1);
    if (r0 + r < rows) {
      const long i = r0 + r;
      const int a = (int)(i % A);
      const long b = i / A;
      float* o = pred + b * (long)pred_C * A + a;
      for (int c = tid / DEC_ANCH; c < nc; c += EW_THREADS / DEC_ANCH) o[(long)(4 + c) * A] = ys_sigmoid(sC[r * pc + c]);
    }
  }
}
__global__ void __launch_bounds__(EW_THREADS) obb_angle_kernel(float* __restrict__ p, long n) {
  const long i = (long)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i < n) p[i] = (ys_sigmoid(p[i]) - 0.25f) * 3.14159265358979323846f;
}
int ys_obb_angle_launch(hipStream_t st, float* p, long n) {
  if (n > 0) YS_LAUNCH(obb_angle_kernel, ys_cdiv(n, EW_THREADS), EW_THREADS, st, p, n);
  return YS_OK;
}
int ys_detect_decode_launch(hipStream_t st, int dtype, const void* pd, int ld_pd, const void* ps, int ld_ps, int B, int A,
                            int nc, int reg_max, int nl, const int* lo, const int* lw, const int* ls, float* pred, int pred_C,
                            const void* px, int ld_px, int xkind, int nx, int kdim) {
  const long n = (long)B * A;
  const int o1 = nl > 1 ? lo[1] : 0, o2 = nl > 2 ? lo[2] : 0, w1 = nl > 1 ? lw[1] : 1, w2 = nl > 2 ? lw[2] : 1;
  const int s1 = nl > 1 ? ls[1] : 1, s2 = nl > 2 ? ls[2] : 1;
  if (!px) xkind = 0;
  if (!xkind) nx = 0;
  const size_t lds_bytes = (size_t)DEC_ANCH * ((4 * reg_max + 1) + (nc | 1) + 4 + (nx | 1)) * 4;
  if (lds_bytes > 60 * 1024 || DEC_ANCH * 4 > EW_THREADS) { ys_set_error("detect decode: nc=%d reg_max=%d too large", nc, reg_max); return YS_ERR_UNSUPPORTED; }
  if (dtype == YS_BF16)
    YS_LAUNCH_LDS((detect_decode_kernel<bf16_t>), ys_cdiv(n, DEC_ANCH), EW_THREADS, lds_bytes, st, (const bf16_t*)pd, ld_pd, (const bf16_t*)ps, ld_ps, B, A, nc, reg_max, nl, lo[0], o1, o2, lw[0], w1, w2, ls[0], s1, s2, pred, pred_C, (const bf16_t*)px, ld_px, xkind, nx, kdim);
  else
    YS_LAUNCH_LDS((detect_decode_kernel<float>), ys_cdiv(n, DEC_ANCH), EW_THREADS, lds_bytes, st, (const float*)pd, ld_pd, (const float*)ps, ld_ps, B, A, nc, reg_max, nl, lo[0], o1, o2, lw[0], w1, w2, ls[0], s1, s2, pred, pred_C, (const float*)px, ld_px, xkind, nx, kdim);
  return YS_OK;
}

// ------------------------------------------------------------------ grouped (multi-problem) BatchNorm passes
// The BN passes of INDEPENDENT Conv units that the planner runs side by side (the tower layers of the three pyramid levels of a
// head, model.hip run_conv_fwd_group / run_conv_bwd_group) as one launch each: workgroups [end[i-1], end[i]) serve problem i.  The
// P4 / P5 instances are 5-14 us launch-latency-bound kernels on their own; same arithmetic, same order of operations per element
// and per channel as the single-problem kernels above (bit-identical results).
template <class G> __device__ inline int ys_ew_group_pick(const G& g, int bx, int& start) {
  int pi = 0;
#pragma unroll
  for (int k = 0; k + 1 < YS_EW_GROUP_MAX; k++) pi += (int)(k + 1 < g.n && bx >= g.end[k]);
  start = pi ? g.end[pi - 1] : 0;
  return pi;
}

__global__ void __launch_bounds__(EW_THREADS)
bn_finalize_group_kernel(BnFinGroup grp, float eps, float momentum) {
  __shared__ double sbuf[EW_THREADS];
  int start;
  const BnFinProb& p = grp.p[ys_ew_group_pick(grp, (int)blockIdx.x, start)];
  const int c = (int)blockIdx.x - start, C = p.C;
  const float* __restrict__ partial = p.partial;
  float g = 0.f, bt = 0.f, rm0 = 0.f, rv0 = 0.f, nbt0 = 0.f;
  if (threadIdx.x == 0) { g = p.gamma[c]; bt = p.beta[c]; rm0 = p.run_mean[c]; rv0 = p.run_var[c]; if (c == 0 && p.nbt) nbt0 = p.nbt[0]; }
  double s1 = 0.0, s2 = 0.0;
  for (int k = threadIdx.x; k < p.nblk; k += EW_THREADS) {
    s1 += (double)partial[((long)k * 2 + 0) * C + c];
    s2 += (double)partial[((long)k * 2 + 1) * C + c];
  }
  block_sum2_d(s1, s2, sbuf);
  if (threadIdx.x == 0) {
    const double count = p.count;
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    p.scale[c] = g * rstd;
    p.shift[c] = bt - (float)mean * g * rstd;
    p.mean[c] = (float)mean;
    p.rstd[c] = rstd;
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    p.run_mean[c] = (1.0f - momentum) * rm0 + momentum * (float)mean;
    p.run_var[c] = (1.0f - momentum) * rv0 + momentum * (float)unb;
    if (c == 0 && p.nbt) p.nbt[0] = nbt0 + 1.0f;
  }
}
int ys_bn_finalize_group_launch(hipStream_t st, const BnFinProb* probs, int n, float eps, float momentum) {
  if (n < 1 || n > YS_EW_GROUP_MAX) { ys_set_error("bn finalize group: %d problems", n); return YS_ERR_INVALID_ARG; }
  BnFinGroup grp{};
  grp.n = n;
  int total = 0;
  for (int i = 0; i < n; i++) { grp.p[i] = probs[i]; total += probs[i].C; grp.end[i] = total; }
  for (int i = n; i < YS_EW_GROUP_MAX; i++) grp.end[i] = total;
  YS_LAUNCH(bn_finalize_group_kernel, total, EW_THREADS, st, grp, eps, momentum);
  return YS_OK;
}

template <class T, bool ACT>
__global__ void __launch_bounds__(EW_THREADS)
bn_act_apply_group_kernel(BnApplyGroup grp) {
  constexpr int EPL = Elem<T>::EPL;
  int start;
  const BnApplyProb& p = grp.p[ys_ew_group_pick(grp, (int)blockIdx.x, start)];
  const int C = p.C, CG = C / EPL;
  const unsigned iu = (unsigned)((int)blockIdx.x - start) * (unsigned)EW_THREADS + threadIdx.x;     // rows * CG < 2^31 (checked by the launcher)
  if ((long)iu < p.rows * CG) {
    const unsigned r = iu / (unsigned)CG;
    const long row = r; const int c = (int)(iu - r * (unsigned)CG) * EPL;
    const T* __restrict__ y = (const T*)p.y; const T* __restrict__ res = (const T*)p.res; T* __restrict__ z = (T*)p.z;
    float f[EPL], rr[EPL], sc[EPL], sh[EPL];
    ys_unpack<T>(ys_ld16(y + row * C + c), f);
    if (res) ys_unpack<T>(ys_ld16(res + row * p.res_ldc + p.res_coff + c), rr);
    ys_ldcoef<EPL>(p.scale + c, sc);
    ys_ldcoef<EPL>(p.shift + c, sh);
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      float u = f[e] * sc[e] + sh[e];
      if (ACT) u = ys_silu(u);
      if (res) u += rr[e];
      f[e] = u;
    }
    ys_st16(z + row * p.z_ldc + p.z_coff + c, ys_pack<T>(f));
  }
}
int ys_bn_act_apply_group_launch(hipStream_t st, int dtype, const BnApplyProb* probs, int n, int act) {
  if (n < 1 || n > YS_EW_GROUP_MAX) { ys_set_error("bn apply group: %d problems", n); return YS_ERR_INVALID_ARG; }
  const int epl = dtype == YS_BF16 ? 8 : 4;
  BnApplyGroup grp{};
  grp.n = n;
  long total = 0;
  for (int i = 0; i < n; i++) {
    const long nn = probs[i].rows * (probs[i].C / epl);
    if (nn >= (1L << 31)) { ys_set_error("bn apply group: problem too large"); return YS_ERR_UNSUPPORTED; }
    grp.p[i] = probs[i]; total += ys_cdiv(nn, EW_THREADS); grp.end[i] = (int)total;
  }
  for (int i = n; i < YS_EW_GROUP_MAX; i++) grp.end[i] = (int)total;
#define BAG_LAUNCH(TT, AF) YS_LAUNCH((bn_act_apply_group_kernel<TT, AF>), (int)total, EW_THREADS, st, grp)
  if (dtype == YS_BF16) { if (act) BAG_LAUNCH(bf16_t, true); else BAG_LAUNCH(bf16_t, false); }
  else { if (act) BAG_LAUNCH(float, true); else BAG_LAUNCH(float, false); }
#undef BAG_LAUNCH
  return YS_OK;
}

// MODE 0 (BN backward) of chan_finalize_kernel for several units at once
__global__ void __launch_bounds__(EW_THREADS)
chan_finalize_group_kernel(ChanFinGroup grp) {
  __shared__ double sbuf[EW_THREADS];
  int start;
  const ChanFinProb& p = grp.p[ys_ew_group_pick(grp, (int)blockIdx.x, start)];
  const int c = (int)blockIdx.x - start, C = p.C;
  const float* __restrict__ partial = p.src.p[0]; int nblk = p.src.nblk[0];
#pragma unroll
  for (int k = 1; k < YS_BNRED_MAXSEG; k++) if (k < p.src.n && c >= p.src.c1[k - 1]) { partial = p.src.p[k]; nblk = p.src.nblk[k]; }
  float sc = 0.f, mu = 0.f, rs = 0.f, g0p = 0.f, g1p = 0.f;
  if (threadIdx.x == 0) { g0p = p.g0[c]; sc = p.scale[c]; mu = p.mean[c]; rs = p.rstd[c]; g1p = p.g1[c]; }
  double s1 = 0.0, s2 = 0.0;
  for (int k = threadIdx.x; k < nblk; k += EW_THREADS) {
    s1 += (double)partial[((long)k * 2 + 0) * C + c];
    s2 += (double)partial[((long)k * 2 + 1) * C + c];
  }
  block_sum2_d(s1, s2, sbuf);
  if (threadIdx.x == 0) {
    const double count = p.count;
    s2 = (double)rs * (s2 - (double)mu * s1);
    p.g0[c] = g0p + (float)s2;
    p.g1[c] = g1p + (float)s1;
    const float m1 = (float)(s1 / count), m2 = (float)(s2 / count);
    p.c1[c] = sc * (m1 - mu * rs * m2);
    p.c2[c] = sc * rs * m2;
  }
}
int ys_bn_bwd_finalize_group_launch(hipStream_t st, const ChanFinProb* probs, int n) {
  if (n < 1 || n > YS_EW_GROUP_MAX) { ys_set_error("bn backward finalize group: %d problems", n); return YS_ERR_INVALID_ARG; }
  ChanFinGroup grp{};
  grp.n = n;
  int total = 0;
  for (int i = 0; i < n; i++) { grp.p[i] = probs[i]; total += probs[i].C; grp.end[i] = total; }
  for (int i = n; i < YS_EW_GROUP_MAX; i++) grp.end[i] = total;
  YS_LAUNCH(chan_finalize_group_kernel, total, EW_THREADS, st, grp);
  return YS_OK;
}

template <class T, bool ACT>
__global__ void __launch_bounds__(EW_THREADS)
bn_bwd_apply_group_kernel(BnBwdGroup grp) {
  constexpr int EPL = Elem<T>::EPL;
  int start;
  const BnBwdProb& p = grp.p[ys_ew_group_pick(grp, (int)blockIdx.x, start)];
  const int C = p.C, CG = C / EPL;
  const unsigned iu = (unsigned)((int)blockIdx.x - start) * (unsigned)EW_THREADS + threadIdx.x;
  if ((long)iu < p.rows * CG) {
    const unsigned r = iu / (unsigned)CG;
    const long row = r; const int c = (int)(iu - r * (unsigned)CG) * EPL;
    const T* __restrict__ dz = (const T*)p.dz; const T* __restrict__ y = (const T*)p.y; T* __restrict__ dy = (T*)p.dy; T* __restrict__ rg = (T*)p.rg;
    float g[EPL], f[EPL], sc[EPL], sh[EPL], a2[EPL], a3[EPL];
    ys_unpack<T>(ys_ld16(dz + row * p.dz_ldc + p.dz_coff + c), g);
    ys_unpack<T>(ys_ld16(y + row * C + c), f);
    if (rg) {
      float o[EPL];
      T* rp = rg + row * p.rg_ldc + p.rg_coff + c;
      ys_unpack<T>(ys_ld16(rp), o);
#pragma unroll
      for (int e = 0; e < EPL; e++) o[e] += g[e];
      ys_st16(rp, ys_pack<T>(o));
    }
    ys_ldcoef<EPL>(p.scale + c, sc); ys_ldcoef<EPL>(p.shift + c, sh); ys_ldcoef<EPL>(p.k2 + c, a2); ys_ldcoef<EPL>(p.k3 + c, a3);
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      const float u = f[e] * sc[e] + sh[e];
      const float du = ACT ? g[e] * ys_silu_grad(u) : g[e];
      f[e] = sc[e] * du - a2[e] - f[e] * a3[e];
    }
    ys_st16(dy + row * C + c, ys_pack<T>(f));
  }
}
int ys_bn_bwd_apply_group_launch(hipStream_t st, int dtype, const BnBwdProb* probs, int n, int act) {
  if (n < 1 || n > YS_EW_GROUP_MAX) { ys_set_error("bn backward apply group: %d problems", n); return YS_ERR_INVALID_ARG; }
  const int epl = dtype == YS_BF16 ? 8 : 4;
  BnBwdGroup grp{};
  grp.n = n;
  long total = 0;
  for (int i = 0; i < n; i++) {
    const long nn = probs[i].rows * (probs[i].C / epl);
    if (nn >= (1L << 31)) { ys_set_error("bn backward apply group: problem too large"); return YS_ERR_UNSUPPORTED; }
    grp.p[i] = probs[i]; total += ys_cdiv(nn, EW_THREADS); grp.end[i] = (int)total;
  }
  for (int i = n; i < YS_EW_GROUP_MAX; i++) grp.end[i] = (int)total;
#define BBG_LAUNCH(TT, AF) YS_LAUNCH((bn_bwd_apply_group_kernel<TT, AF>), (int)total, EW_THREADS, st, grp)
  if (dtype == YS_BF16) { if (act) BBG_LAUNCH(bf16_t, true); else BBG_LAUNCH(bf16_t, false); }
  else { if (act) BBG_LAUNCH(float, true); else BBG_LAUNCH(float, false); }
#undef BBG_LAUNCH
  return YS_OK;
}
