// f8.hip -- scaling machinery of the fp8 convolution path (BASELINE config 5: "fp8 MFMA implicit-GEMM conv path").
//
// Recipe (per-tensor scaling, OCP formats; the reference has no fp8 mode -- Amp.cs knows fp16 / bf16 only -- so this is a
// performance mode validated against fp32 on quantised operands, like bf16):
//   weights      e4m3, CURRENT scaling: s_w = 448 / amax(|W|) of the fp32 master weights, recomputed with every weight refresh
//   activations  e4m3, DELAYED scaling: s_x = 0.5 * 448 / amax(|x|) of the same convolution input in the previous step (recorded by
//                the fp8 kernel itself while it stages its input patch; first step: a bootstrap pass); factor 2 of headroom, saturating
//   gradients    e5m2, DELAYED scaling: s_g = 8192 / amax(|dy|) of the previous step (57344 is the largest finite e5m2)
// Accumulation is fp32 (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales); y = acc / (s_x * s_w).  BatchNorm statistics,
// the stored activations (bf16), the weight gradients (bf16 operands) and the optimizer are those of the bf16 mode.
#include "ys_internal.h"
#include "ys_kernels.h"

// amax of each layer's fp32 master weights: grid (layer, chunk); four independent 16-byte loads per thread and trip, one atomic
// per workgroup into the layer's slot (cleared by the launcher; max of non-negative floats = max of their bit patterns)
#define F8_WCHUNKS 16
__global__ void __launch_bounds__(256)
f8_weight_amax_kernel(const float* __restrict__ params, const F8Layer* __restrict__ layers, int n, float* __restrict__ amax_w) {
  __shared__ float s[4];
  const int l = blockIdx.x;
  if (l >= n) return;
  const F8Layer d = layers[l];
  const float* w = params + d.w_off;
  float m = 0.f;
  const long per = (d.count + F8_WCHUNKS - 1) / F8_WCHUNKS;
  const long lo = (long)blockIdx.y * per, hi = lo + per < d.count ? lo + per : d.count;
  long i = lo + threadIdx.x;
  for (; i + 768 < hi; i += 1024) {
    const float a = w[i], b = w[i + 256], c = w[i + 512], e = w[i + 768];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(a), fabsf(b))), fmaxf(fabsf(c), fabsf(e)));
  }
  for (; i < hi; i += 256) m = fmaxf(m, fabsf(w[i]));
  m = ys_wave_max(m);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    if (t > 0.f) atomicMax((unsigned*)amax_w + l, ys_f2u(t));
  }
}

// Per convolution c: out[4c..] = {s_x, 1/(s_x s_w), s_g, 1/(s_g s_w)}.  amax_x / amax_dy hold YS_AMAX_WAYS slots per convolution
// (float bits of non-negative values), written by the fp8 convolution kernels themselves while they stage their input (or by
// the bootstrap pass of the first step) and consumed here: a convolution with a recorded maximum gets a new scale and its slots
// are cleared; one without (nothing ran since the last refresh) keeps its scale.  The weight scale is always current.
__global__ void __launch_bounds__(256)
f8_scales_kernel(const F8Conv* __restrict__ convs, int n, const float* __restrict__ amax_w, unsigned* __restrict__ amax_x,
                 unsigned* __restrict__ amax_dy, float* __restrict__ out) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
    const float aw = amax_w[convs[c].layer];
    const float sw = aw > 0.f ? YS_E4M3_MAX / aw : 1.0f;
    float ax = 0.f, ag = 0.f;
    for (int w = 0; w < YS_AMAX_WAYS; w++) {
      ax = fmaxf(ax, ys_u2f(amax_x[c * YS_AMAX_WAYS + w])); amax_x[c * YS_AMAX_WAYS + w] = 0u;
      ag = fmaxf(ag, ys_u2f(amax_dy[c * YS_AMAX_WAYS + w])); amax_dy[c * YS_AMAX_WAYS + w] = 0u;
    }
    float sx = out[4 * c + 0], sg = out[4 * c + 2];
    if (ax > 0.f) sx = 0.5f * YS_E4M3_MAX / ax;
    if (ag > 0.f) sg = 8192.0f / ag;
    if (!(sx > 0.f)) sx = 1.0f;
    if (!(sg > 0.f)) sg = 1.0f;
    out[4 * c + 0] = sx; out[4 * c + 1] = 1.0f / (sx * sw);
    out[4 * c + 2] = sg; out[4 * c + 3] = 1.0f / (sg * sw);
  }
}

int ys_f8_weight_amax_launch(hipStream_t st, const float* params, const F8Layer* layers, int n, float* amax_w) {
  if (n <= 0) return YS_OK;
  YS_CHECK_HIP(hipMemsetAsync(amax_w, 0, (size_t)n * 4, st));
  YS_LAUNCH(f8_weight_amax_kernel, dim3(n, F8_WCHUNKS), 256, st, params, layers, n, amax_w);
  return YS_OK;
}
int ys_f8_scales_launch(hipStream_t st, const F8Conv* convs, int n, const float* amax_w, unsigned* amax_x, unsigned* amax_dy, float* out) {
  if (n <= 0) return YS_OK;
  YS_LAUNCH(f8_scales_kernel, ys_cdiv(n, 256), 256, st, convs, n, amax_w, amax_x, amax_dy, out);
  return YS_OK;
}

// amax(|x|) of a [rows][C] bf16 view (bootstrap of the delayed scales in the first step; the stateless per-operator API)
__global__ void __launch_bounds__(256)
f8_view_amax_kernel(const bf16_t* __restrict__ x, long rows, int C, int ldc, int coff, unsigned* __restrict__ slots) {
  const int CG = C / 8;
  const long n = rows * CG;
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / CG; const int c = (int)(i - r * CG) * 8;
    float f[8];
    ys_unpack<bf16_t>(ys_ld16(x + r * ldc + coff + c), f);
#pragma unroll
    for (int e = 0; e < 8; e++) m = fmaxf(m, fabsf(f[e]));
  }
  ys_amax_update(slots, m);
}
int ys_f8_view_amax_launch(hipStream_t st, const void* x, long rows, int C, int ldc, int coff, unsigned* slots) {
  if (rows <= 0 || C <= 0) return YS_OK;
  long g = ys_cdiv(rows * (C / 8), 256 * 4L); if (g > 2048) g = 2048; if (g < 1) g = 1;
  YS_LAUNCH(f8_view_amax_kernel, (int)g, 256, st, (const bf16_t*)x, rows, C, ldc, coff, slots);
  return YS_OK;
}

// [rows][C] bf16 view -> dense [rows][C] fp8 image (FMT 0 = e4m3 activations, 1 = e5m2 gradients) with the tensor's delayed scale,
// recording amax(|x|) for the next step's scale on the way: the operand of conv_gemm_kernel<F8 = 1>, which stages its tiles by LDS
// DMA and therefore cannot quantise on the fly the way conv_p2_kernel does.  16 elements per thread: two 16-byte loads, one store.
template <int FMT>
__global__ void __launch_bounds__(256)
f8_quant_view_kernel(const bf16_t* __restrict__ x, long rows, int C, int ldc, int coff, const float* __restrict__ qscale,
                     unsigned char* __restrict__ out, unsigned* __restrict__ slots) {
  const int CG = C / 16;
  const long n = rows * CG;
  const float qs = qscale[0];
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / CG; const int c = (int)(i - r * CG) * 16;
    const bf16_t* src = x + r * ldc + coff + c;
    float f[16];
    ys_unpack<bf16_t>(ys_ld16(src), f);
    ys_unpack<bf16_t>(ys_ld16(src + 8), f + 8);
#pragma unroll
    for (int e = 0; e < 16; e++) { m = fmaxf(m, fabsf(f[e])); f[e] *= qs; }
    const uint2 lo = ys_pack_f8x8<FMT>(f), hi = ys_pack_f8x8<FMT>(f + 8);
    ys_st16(out + r * C + c, make_uint4(lo.x, lo.y, hi.x, hi.y));
  }
  if (slots) ys_amax_update(slots, m);
}
int ys_f8_quant_view_launch(hipStream_t st, int fmt, const void* x, long rows, int C, int ldc, int coff, const float* qscale,
                            void* out, unsigned* slots) {
  if (rows <= 0 || C <= 0) return YS_OK;
  long g = ys_cdiv(rows * (C / 16), 256 * 4L); if (g > 2048) g = 2048; if (g < 1) g = 1;
  if (fmt) YS_LAUNCH(f8_quant_view_kernel<1>, (int)g, 256, st, (const bf16_t*)x, rows, C, ldc, coff, qscale, (unsigned char*)out, slots);
  else YS_LAUNCH(f8_quant_view_kernel<0>, (int)g, 256, st, (const bf16_t*)x, rows, C, ldc, coff, qscale, (unsigned char*)out, slots);
  return YS_OK;
}

// element-wise e4m3 copy of a bf16 weight shadow with the layer's scale (same element order -> same offsets)
__global__ void __launch_bounds__(256)
f8_quant_weights_kernel(const bf16_t* __restrict__ w, long n, const float* __restrict__ amax_w, unsigned char* __restrict__ w8) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n) return;
  const float aw = amax_w[0];
  const float sw = aw > 0.f ? YS_E4M3_MAX / aw : 1.0f;
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; e++) f[e] = (i + e < n) ? Elem<bf16_t>::to_f(w[i + e]) * sw : 0.f;
  const uint2 v = ys_pack_f8x8<0>(f);
  if (i + 8 <= n) *(uint2*)(w8 + i) = v;
  else for (int e = 0; e < 8 && i + e < n; e++) w8[i + e] = (unsigned char)((e < 4 ? v.x >> (8 * e) : v.y >> (8 * (e - 4))) & 255u);
}
int ys_f8_quant_weights_launch(hipStream_t st, const void* w_bf16, long n, const float* amax_w, void* w8) {
  if (n <= 0) return YS_OK;
  YS_LAUNCH(f8_quant_weights_kernel, (int)ys_cdiv(n, 256 * 8L), 256, st, (const bf16_t*)w_bf16, n, amax_w, (unsigned char*)w8);
  return YS_OK;
}
