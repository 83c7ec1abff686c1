// ys_hip.h -- common device/host definitions for the yolosharp_hip kernels (gfx950 / CDNA4).
// Built by hipcc for the product library.  When YS_EMU_BUILD is defined the same sources are
// compiled by g++ against tools/hipemu (a TEST-ONLY SIMT interpreter; never shipped, never a
// fallback -- see tools/hipemu/hip_emu.h).
#pragma once
#ifdef YS_EMU_BUILD
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stddef.h>

#define YS_WAVE 64

// ---------------------------------------------------------------- launch helper
#ifdef YS_EMU_BUILD
#define YS_LAUNCH(KERNEL, GRID, BLOCK, STREAM, ...) \
  emu::launch(dim3(GRID), dim3(BLOCK), [=]() { KERNEL(__VA_ARGS__); })
#else
#define YS_LAUNCH(KERNEL, GRID, BLOCK, STREAM, ...) \
  hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, STREAM, __VA_ARGS__)
#endif

// s_waitcnt vmcnt(0) as a real instruction the compiler's wait-count bookkeeping sees (gfx9 encoding: vmcnt = 0, expcnt = 7,
// lgkmcnt = 15 -> 0x0F70); nothing to do in the interpreter
#ifdef YS_EMU_BUILD
#define YS_WAIT_VM0() ((void)0)
#else
#define YS_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)
#endif
// s_waitcnt vmcnt(N), N <= 63 (gfx9 encoding: vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14; expcnt / lgkmcnt left at "no wait"):
// at most N of this wave's vector-memory instructions still in flight -- loads and stores retire in issue order on gfx9-family parts (one counter,
// hipcc's own s_waitcnt insertion relies on it).  A wait whose N counts only LOADS issued after the one waited for is sound even without
// that assumption (conv_gemm_kernel's K-loop waits); conv_p2_body's tile-top wait counts later STORES and does rely on it.
template <int N> __device__ inline void ys_wait_vm() {
#ifndef YS_EMU_BUILD
  static_assert(N >= 0 && N <= 63, "vmcnt");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
#endif
}

// the same with a wave-uniform run-time count (0..15; larger counts wait as for 15, i.e. longer than asked)
__device__ inline void ys_wait_vm_dyn(int n) {
#ifndef YS_EMU_BUILD
  switch (__builtin_amdgcn_readfirstlane(n)) {
    case 0: ys_wait_vm<0>(); break; case 1: ys_wait_vm<1>(); break; case 2: ys_wait_vm<2>(); break; case 3: ys_wait_vm<3>(); break;
    case 4: ys_wait_vm<4>(); break; case 5: ys_wait_vm<5>(); break; case 6: ys_wait_vm<6>(); break; case 7: ys_wait_vm<7>(); break;
    case 8: ys_wait_vm<8>(); break; case 9: ys_wait_vm<9>(); break; case 10: ys_wait_vm<10>(); break; case 11: ys_wait_vm<11>(); break;
    case 12: ys_wait_vm<12>(); break; case 13: ys_wait_vm<13>(); break; case 14: ys_wait_vm<14>(); break; default: ys_wait_vm<15>(); break;
  }
#else
  (void)n;
#endif
}

// Scheduling fence: hipcc's machine scheduler otherwise sinks LDS reads next to their first use (fewer live registers), which
// serialises every read's latency with the MFMAs; a fence keeps "issue all reads, then all MFMAs" as written.
#ifdef YS_EMU_BUILD
#define YS_SCHED_FENCE() ((void)0)
#else
#define YS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// launch with dynamic LDS; YS_DYN_LDS(name) declares the dynamic region inside a kernel as `uint4* name`
#ifdef YS_EMU_BUILD
#define YS_LAUNCH_LDS(KERNEL, GRID, BLOCK, LDS_BYTES, STREAM, ...) \
  emu::launch(dim3(GRID), dim3(BLOCK), [=]() { KERNEL(__VA_ARGS__); }, (size_t)(LDS_BYTES))
#define YS_DYN_LDS(name) uint4* name = (uint4*)emu::dyn_lds()
#else
#define YS_LAUNCH_LDS(KERNEL, GRID, BLOCK, LDS_BYTES, STREAM, ...) \
  hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), (LDS_BYTES), STREAM, __VA_ARGS__)
#define YS_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) uint4 name[]
#endif

// ---------------------------------------------------------------- bf16 storage type
struct bf16_t { unsigned short v; };

__host__ __device__ inline float ys_u2f(unsigned u) { return __builtin_bit_cast(float, u); }
__host__ __device__ inline unsigned ys_f2u(float f) { return __builtin_bit_cast(unsigned, f); }

__host__ __device__ inline float bf16_bits_to_f32(unsigned short h) { return ys_u2f(((unsigned)h) << 16); }
// round-to-nearest-even, NaN preserved (quiet)
__host__ __device__ inline unsigned short f32_to_bf16_bits(float f) {
  unsigned u = ys_f2u(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

template <class T> struct Elem;
template <> struct Elem<float> {
  static constexpr int EPL = 4;  // elements per 16-byte lane load
  __host__ __device__ static inline float to_f(float x) { return x; }
  __host__ __device__ static inline float from_f(float x) { return x; }
};
template <> struct Elem<bf16_t> {
  static constexpr int EPL = 8;
  __host__ __device__ static inline float to_f(bf16_t x) { return bf16_bits_to_f32(x.v); }
  __host__ __device__ static inline bf16_t from_f(float x) {
    bf16_t r;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(YS_EMU_BUILD)
    r.v = __builtin_bit_cast(unsigned short, (__bf16)x);          // v_cvt_pk_bf16_f32
#else
    r.v = f32_to_bf16_bits(x);
#endif
    return r;
  }
};

// unpack / pack one 16-byte vector of T to floats
template <class T> __device__ inline void ys_unpack(const uint4& v, float* f);
template <> __device__ inline void ys_unpack<float>(const uint4& v, float* f) {
  f[0] = ys_u2f(v.x); f[1] = ys_u2f(v.y); f[2] = ys_u2f(v.z); f[3] = ys_u2f(v.w);
}
template <> __device__ inline void ys_unpack<bf16_t>(const uint4& v, float* f) {
  f[0] = ys_u2f(v.x << 16); f[1] = ys_u2f(v.x & 0xffff0000u);
  f[2] = ys_u2f(v.y << 16); f[3] = ys_u2f(v.y & 0xffff0000u);
  f[4] = ys_u2f(v.z << 16); f[5] = ys_u2f(v.z & 0xffff0000u);
  f[6] = ys_u2f(v.w << 16); f[7] = ys_u2f(v.w & 0xffff0000u);
}
template <class T> __device__ inline uint4 ys_pack(const float* f);
template <> __device__ inline uint4 ys_pack<float>(const float* f) {
  return make_uint4(ys_f2u(f[0]), ys_f2u(f[1]), ys_f2u(f[2]), ys_f2u(f[3]));
}
// two floats -> packed bf16 pair, round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950 (the bit-twiddled form costs ~8
// VALU instructions per element, i.e. ~64 per stored 16-byte vector in every bf16-writing kernel)
__device__ inline unsigned ys_pack_bf16x2(float lo, float hi) {
#ifdef YS_EMU_BUILD
  return (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
#else
  typedef __bf16 ys_bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float ys_f32x2_t __attribute__((ext_vector_type(2)));
  ys_f32x2_t v; v[0] = lo; v[1] = hi;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ys_bf16x2_t));
#endif
}
template <> __device__ inline uint4 ys_pack<bf16_t>(const float* f) {
  return make_uint4(ys_pack_bf16x2(f[0], f[1]), ys_pack_bf16x2(f[2], f[3]),
                    ys_pack_bf16x2(f[4], f[5]), ys_pack_bf16x2(f[6], f[7]));
}

// gather EPL storage elements into one 16-byte fragment (no type punning through memory)
__device__ inline uint4 ys_pack_elems(const float* e) {
  return make_uint4(ys_f2u(e[0]), ys_f2u(e[1]), ys_f2u(e[2]), ys_f2u(e[3]));
}
__device__ inline uint4 ys_pack_elems(const bf16_t* e) {
  return make_uint4((unsigned)e[0].v | ((unsigned)e[1].v << 16), (unsigned)e[2].v | ((unsigned)e[3].v << 16),
                    (unsigned)e[4].v | ((unsigned)e[5].v << 16), (unsigned)e[6].v | ((unsigned)e[7].v << 16));
}

// ---------------------------------------------------------------- accumulator vector
#ifdef YS_EMU_BUILD
struct f32x4 {
  float d[4];
  float& operator[](int i) { return d[i]; }
  const float& operator[](int i) const { return d[i]; }
};
#else
typedef float f32x4 __attribute__((ext_vector_type(4)));
#endif
__device__ inline f32x4 f32x4_zero() { f32x4 z; z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f; return z; }

// ---------------------------------------------------------------- MFMA wrappers
// v_mfma_f32_16x16x32_bf16: D[16x16] = A[16x32] * B[32x16] + C.
//   lane l supplies A[i = l&15][k = 8*(l>>4) .. +8) and B[k = 8*(l>>4) .. +8)[j = l&15];
//   it receives D[i = 4*(l>>4) + r][j = l&15], r = 0..3.
__device__ inline f32x4 mfma_16x16x32_bf16(const uint4& a, const uint4& b, const f32x4& c) {
#ifdef YS_EMU_BUILD
  struct Dep { uint4 a, b; } dep{a, b};
  f32x4 d = c;
  const int l = emu::lane();
  emu::wave_collective(&dep, sizeof(dep), [&](unsigned char (*slot)[128]) {
    const int j = l & 15;
    for (int r = 0; r < 4; r++) {
      const int i = 4 * (l >> 4) + r;
      float acc = d[r];
      for (int k = 0; k < 32; k++) {
        Dep da, db;
        memcpy(&da, slot[i + 16 * (k >> 3)], sizeof(Dep));
        memcpy(&db, slot[j + 16 * (k >> 3)], sizeof(Dep));
        const unsigned short* pa = (const unsigned short*)&da.a;
        const unsigned short* pb = (const unsigned short*)&db.b;
        acc += bf16_bits_to_f32(pa[k & 7]) * bf16_bits_to_f32(pb[k & 7]);
      }
      d[r] = acc;
    }
  });
  return d;
#else
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
}

// v_mfma_f32_16x16x4_f32 (exact f32, k-ordered fma chain): lane l supplies A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15]; receives D[4*(l>>4)+r][l&15].
__device__ inline f32x4 mfma_16x16x4_f32(float a, float b, const f32x4& c) {
#ifdef YS_EMU_BUILD
  struct Dep { float a, b; } dep{a, b};
  f32x4 d = c;
  const int l = emu::lane();
  emu::wave_collective(&dep, sizeof(dep), [&](unsigned char (*slot)[128]) {
    const int j = l & 15;
    for (int r = 0; r < 4; r++) {
      const int i = 4 * (l >> 4) + r;
      float acc = d[r];
      for (int k = 0; k < 4; k++) {
        Dep da, db;
        memcpy(&da, slot[i + 16 * k], sizeof(Dep));
        memcpy(&db, slot[j + 16 * k], sizeof(Dep));
        acc = fmaf(da.a, db.b, acc);
      }
      d[r] = acc;
    }
  });
  return d;
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

// ---------------------------------------------------------------- OCP fp8 (e4m3fn) / bf8 (e5m2) for the fp8 convolution path
// Hardware facts measured with tools/probe/probe_f8.hip on MI355X: v_cvt_pk_{fp8,bf8}_f32 round to nearest even, produce
// subnormals, and do NOT saturate (e4m3: |x| >= 480 -> NaN 0x7F; e5m2: |x| >= 61440 -> inf) -- so values are clamped to the
// largest finite magnitude first.  The software forms below are bit-identical for finite inputs (interpreter, tests).
#define YS_E4M3_MAX 448.0f
#define YS_E5M2_MAX 57344.0f
__host__ __device__ inline float ys_e4m3_to_f32(unsigned char b) {
  const int e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf((float)m, -9) : ((e == 15 && m == 7) ? __builtin_nanf("") : ldexpf(1.0f + (float)m * 0.125f, e - 7));
  return (b & 0x80) ? -v : v;
}
__host__ __device__ inline float ys_e5m2_to_f32(unsigned char b) {
  const int e = (b >> 2) & 31, m = b & 3;
  float v = e == 0 ? ldexpf((float)m, -16) : (e == 31 ? (m ? __builtin_nanf("") : __builtin_inff()) : ldexpf(1.0f + (float)m * 0.25f, e - 15));
  return (b & 0x80) ? -v : v;
}
// round-to-nearest-even to a format with MB mantissa bits, exponent bias BIAS, after clamping |x| to MAXV
template <int MB, int BIAS>
__host__ __device__ inline unsigned char ys_f32_to_f8_sw(float x, float maxv) {
  const unsigned char sign = (ys_f2u(x) >> 31) ? 0x80 : 0;
  if (x != x) return (unsigned char)(sign | 0x7F);
  float a = x < 0.f ? -x : x;
  if (a > maxv) a = maxv;
  const float min_normal = ldexpf(1.0f, 1 - BIAS);
  if (a < min_normal) {                                   // subnormal grid: multiples of 2^(1 - BIAS - MB)
    const int qv = (int)nearbyintf(ldexpf(a, BIAS - 1 + MB));
    return (unsigned char)(sign | qv);                    // qv == 2^MB is exactly the smallest normal's encoding
  }
  int e;
  const float fr = frexpf(a, &e);                         // a = fr * 2^e, fr in [0.5, 1)
  e -= 1;                                                 // a = (2 fr) * 2^e, 2 fr in [1, 2)
  int qv = (int)nearbyintf((2.0f * fr - 1.0f) * (float)(1 << MB));
  if (qv == (1 << MB)) { qv = 0; e += 1; }
  return (unsigned char)(sign | ((e + BIAS) << MB) | qv);
}
__host__ __device__ inline unsigned char ys_f32_to_e4m3(float x) { return ys_f32_to_f8_sw<3, 7>(x, YS_E4M3_MAX); }
__host__ __device__ inline unsigned char ys_f32_to_e5m2(float x) { return ys_f32_to_f8_sw<2, 15>(x, YS_E5M2_MAX); }

// one float -> e4m3 byte, saturating (device: v_cvt_pk_fp8_f32 after the clamp; interpreter: the software form)
__device__ inline unsigned char ys_f32_to_e4m3_dev(float x) {
#ifdef YS_EMU_BUILD
  return ys_f32_to_e4m3(x);
#else
  return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(x, -YS_E4M3_MAX, YS_E4M3_MAX), 0.f, 0u, false) & 255u);
#endif
}

// 8 floats -> 8 fp8 bytes (FMT 0 = e4m3, 1 = e5m2), saturating
template <int FMT> __device__ inline uint2 ys_pack_f8x8(const float* f) {
#ifdef YS_EMU_BUILD
  unsigned char b[8];
  for (int i = 0; i < 8; i++) b[i] = FMT == 0 ? ys_f32_to_e4m3(f[i]) : ys_f32_to_e5m2(f[i]);
  uint2 r;
  r.x = (unsigned)b[0] | ((unsigned)b[1] << 8) | ((unsigned)b[2] << 16) | ((unsigned)b[3] << 24);
  r.y = (unsigned)b[4] | ((unsigned)b[5] << 8) | ((unsigned)b[6] << 16) | ((unsigned)b[7] << 24);
  return r;
#else
  const float mx = FMT == 0 ? YS_E4M3_MAX : YS_E5M2_MAX;
  float c[8];
#pragma unroll
  for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_fmed3f(f[i], -mx, mx);
  unsigned lo = 0u, hi = 0u;
  if (FMT == 0) {
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false); hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  } else {
    lo = __builtin_amdgcn_cvt_pk_bf8_f32(c[0], c[1], lo, false); lo = __builtin_amdgcn_cvt_pk_bf8_f32(c[2], c[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_bf8_f32(c[4], c[5], hi, false); hi = __builtin_amdgcn_cvt_pk_bf8_f32(c[6], c[7], hi, true);
  }
  uint2 r; r.x = lo; r.y = hi;
  return r;
#endif
}

// v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (E8M0 = 127): the only K = 128 fp8 MFMA (2x the bf16 MFMA rate).
// Each lane supplies 32 K-contiguous bytes of the row operand a (e4m3: weights) and of the column operand b (e4m3 activations,
// or e5m2 gradients when B_BF8); lane (li, q) owns row / column li and the q-th quarter of the 128 K values -- both operands
// use the same (lane, byte) -> k map, so any consistent K order gives the same dot products (probe_f8: exact).  Result
// layout as every 16x16 MFMA: D[4q + r][li].
template <int B_BF8>
__device__ inline f32x4 mfma_scale_16x16x128_f8(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1, const f32x4& c) {
#ifdef YS_EMU_BUILD
  struct Dep { uint4 a0, a1, b0, b1; } dep{a0, a1, b0, b1};
  f32x4 d = c;
  const int l = emu::lane();
  emu::wave_collective(&dep, sizeof(dep), [&](unsigned char (*slot)[128]) {
    const int j = l & 15;
    for (int r = 0; r < 4; r++) {
      const int i = 4 * (l >> 4) + r;
      float acc = d[r];
      for (int k = 0; k < 128; k++) {
        Dep da, db;
        memcpy(&da, slot[i + 16 * (k >> 5)], sizeof(Dep));
        memcpy(&db, slot[j + 16 * (k >> 5)], sizeof(Dep));
        const unsigned char* pa = (const unsigned char*)&da.a0;      // a0, a1 contiguous: 32 bytes
        const unsigned char* pb = (const unsigned char*)&db.b0;
        const float fb = B_BF8 ? ys_e5m2_to_f32(pb[k & 31]) : ys_e4m3_to_f32(pb[k & 31]);
        acc += ys_e4m3_to_f32(pa[k & 31]) * fb;
      }
      d[r] = acc;
    }
  });
  return d;
#else
  typedef int ys_v8i __attribute__((ext_vector_type(8)));
  ys_v8i a, b;
  a[0] = (int)a0.x; a[1] = (int)a0.y; a[2] = (int)a0.z; a[3] = (int)a0.w; a[4] = (int)a1.x; a[5] = (int)a1.y; a[6] = (int)a1.z; a[7] = (int)a1.w;
  b[0] = (int)b0.x; b[1] = (int)b0.y; b[2] = (int)b0.z; b[3] = (int)b0.w; b[4] = (int)b1.x; b[5] = (int)b1.y; b[6] = (int)b1.z; b[7] = (int)b1.w;
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, B_BF8 ? 1 : 0, 0, 127, 0, 127);
#endif
}

// One "k-step" of the tile product for storage type T: each lane holds 16 bytes of the
// row operand (a) and 16 bytes of the column operand (b), both K-contiguous.
//   bf16: one 16x16x32 MFMA (lane's 8 elements = k 8q..8q+7).
//   f32 : four 16x16x4 MFMAs (element t of every lane forms k-slot q of MFMA t) -- any
//         consistent permutation of k between a and b is a valid dot product.
template <class T> __device__ inline f32x4 ys_mma(const uint4& a, const uint4& b, f32x4 c);
template <> __device__ inline f32x4 ys_mma<bf16_t>(const uint4& a, const uint4& b, f32x4 c) {
  return mfma_16x16x32_bf16(a, b, c);
}
template <> __device__ inline f32x4 ys_mma<float>(const uint4& a, const uint4& b, f32x4 c) {
#ifdef YS_EMU_BUILD
  // interpreter only: the four chained 16x16x4 MFMAs below as ONE wave rendezvous (same fma order, bit-identical) --
  // the rendezvous, not the arithmetic, dominates the CPU test suite's run time
  struct Dep { uint4 a, b; } dep{a, b};
  f32x4 d = c;
  const int l = emu::lane();
  emu::wave_collective(&dep, sizeof(dep), [&](unsigned char (*slot)[128]) {
    const int j = l & 15;
    for (int r = 0; r < 4; r++) {
      const int i = 4 * (l >> 4) + r;
      float acc = d[r];
      for (int t = 0; t < 4; t++)
        for (int k = 0; k < 4; k++) {
          Dep da, db;
          memcpy(&da, slot[i + 16 * k], sizeof(Dep));
          memcpy(&db, slot[j + 16 * k], sizeof(Dep));
          const unsigned* pa = (const unsigned*)&da.a;
          const unsigned* pb = (const unsigned*)&db.b;
          acc = fmaf(ys_u2f(pa[t]), ys_u2f(pb[t]), acc);
        }
      d[r] = acc;
    }
  });
  return d;
#endif
  c = mfma_16x16x4_f32(ys_u2f(a.x), ys_u2f(b.x), c);
  c = mfma_16x16x4_f32(ys_u2f(a.y), ys_u2f(b.y), c);
  c = mfma_16x16x4_f32(ys_u2f(a.z), ys_u2f(b.z), c);
  c = mfma_16x16x4_f32(ys_u2f(a.w), ys_u2f(b.w), c);
  return c;
}

// 1 / d with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the IEEE division sequence (~10 VALU instructions): the BN /
// SiLU passes are VALU-bound on exactly this; the error is far inside the 1e-3 parity budget
__device__ inline float ys_rcp(float d) {
#ifdef YS_EMU_BUILD
  return 1.0f / d;
#else
  return __builtin_amdgcn_rcpf(d);
#endif
}
__device__ inline float ys_sigmoid(float x) { return ys_rcp(1.0f + __expf(-x)); }
__device__ inline float ys_silu(float x) { return x * ys_rcp(1.0f + __expf(-x)); }
// d/dx [x*sigmoid(x)] = s*(1 + x*(1-s))
__device__ inline float ys_silu_grad(float x) { float s = ys_sigmoid(x); return s * (1.0f + x * (1.0f - s)); }

__device__ inline uint4 ys_ld16(const void* p) { return *(const uint4*)p; }
__device__ inline void ys_st16(void* p, const uint4& v) { *(uint4*)p = v; }
__device__ inline uint4 ys_zero16() { return make_uint4(0u, 0u, 0u, 0u); }

// EPL consecutive per-channel fp32 coefficients (16-byte aligned) as float4 loads
template <int EPL> __device__ inline void ys_ldcoef(const float* p, float* o) {
#pragma unroll
  for (int v = 0; v < EPL / 4; v++) {
    const float4 t = *(const float4*)(p + 4 * v);
    o[4 * v + 0] = t.x; o[4 * v + 1] = t.y; o[4 * v + 2] = t.z; o[4 * v + 3] = t.w;
  }
}

// wave reductions (64 lanes)
__device__ inline float ys_wave_sum(float v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ inline float ys_wave_max(float v) {
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// amax slot update (fp8 delayed scaling): wave maximum, then ONE atomic per wave -- and only when the wave's value would raise
// the slot.  The plain read may be stale (lower), which costs an unnecessary atomic, never a wrong result; after the first few
// workgroups nearly every wave skips.  (One unconditional atomic per wave serialised ~10^6 updates of a single address per
// large tensor: +260 ms on the YOLOv8x 1280 step.)  Non-negative floats order like their bit patterns.
// Each tensor owns YS_AMAX_WAYS slots (workgroup index modulo): the first wavefront of a grid -- thousands of waves that all
// still read 0 -- would otherwise queue its atomics on one address (~160 us per pass); the consumer takes the max of the ways.
#define YS_AMAX_WAYS 64
__device__ inline void ys_amax_update(unsigned* slots, float mx) {
  mx = ys_wave_max(mx);
  if ((threadIdx.x & 63) == 0 && mx > 0.f) {
    unsigned* slot = slots + (blockIdx.x & (YS_AMAX_WAYS - 1));
    const unsigned cur = *(volatile unsigned*)slot;
    if (ys_f2u(mx) > cur) atomicMax(slot, ys_f2u(mx));
  }
}

// 64-bit value of lane `src` (wave-uniform index) as a scalar: v_readlane_b32 x2 instead of a ds_bpermute round trip
__device__ inline unsigned long long ys_readlane64(unsigned long long v, int src) {
#ifdef YS_EMU_BUILD
  return __shfl(v, src);
#else
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
#endif
}

// wave-level ordering point for LDS traffic that is private to one wave (rows written by some lanes, read by others): the wave's
// outstanding LDS operations complete (s_waitcnt lgkmcnt(0)), and the compiler must not reorder across it (the interpreter needs a
// real rendezvous).  Rounds 1-2 had no wait here, on the assumption that a wave's LDS operations execute in order; round 3 measured
// a case where that is not enough (ys_wave_sync_lds below: reads issued right behind bank-conflicted writes of other lanes' rows
// returned stale data about once per 10^6 vectors), so the wait is part of the primitive now.
__device__ inline void ys_wave_sync() {
#ifdef YS_EMU_BUILD
  int z = 0;
  emu::wave_collective(&z, sizeof(z), [](unsigned char (*)[128]) {});
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// The same ordering point WITH a wait for the wave's outstanding LDS operations: a row staged by ds_write and read back by OTHER
// lanes of the wave.  Round 3: without the wait (reads issued right behind the writes) the blocked-GEMM kernel's BatchNorm sums
// differed from run to run on wide layers (cin800 -> cout320 1x1: every rerun; one stale 16-byte vector in ~10^6) -- a bank-
// conflicted ds_write_b64 had not finished all of its passes when the ds_read_b128 of another lane's row went through.  The
// one-iteration-at-a-time epilogue had an s_waitcnt lgkmcnt(0) there by accident (the row-table value was consumed first).
__device__ inline void ys_wave_sync_lds() { ys_wave_sync(); }   // (the wait moved into ys_wave_sync itself; the name marks the site that showed the defect)

// ---------------------------------------------------------------- LDS transpose read (gfx950 ds_read_b64_tr_b16)
// Every lane passes the LDS address of 4 contiguous 16-bit elements (8-byte aligned).  Within each 16-lane group the
// lanes' 16 addresses describe a [4 rows][16 cols] block (lane 4*row + col/4 holds cols 4*(col/4)..+3 of its row; the row
// stride is free) and lane `li` of the group receives column li of the 4 rows:
//   result(l)[j] = elem[ addr(lane (l & ~15) + 4*j + ((l & 15) >> 2)) ][ (l & 15) & 3 ]      j = 0..3
// (lane mapping measured on MI355X with tools/probe/probe_tr.hip).  Two reads give the 8 consecutive-K values an MFMA
// operand lane needs when the LDS image has K as the row index (wgrad: K = pixels, NHWC rows).
// The read is asynchronous: call ys_lds_tr_wait() on every value before it is used.
__device__ inline uint2 ys_lds_tr_b64(const void* p) {
#ifdef YS_EMU_BUILD
  const void* mine = p;
  uint2 out = make_uint2(0u, 0u);
  const int l = emu::lane();
  emu::wave_collective(&mine, sizeof(mine), [&](unsigned char (*slot)[128]) {
    unsigned e[4];
    for (int j = 0; j < 4; j++) {
      const int src = (l & ~15) + 4 * j + ((l & 15) >> 2);
      const unsigned short* sp;
      memcpy(&sp, slot[src], sizeof(sp));
      e[j] = sp[l & 3];
    }
    out.x = e[0] | (e[1] << 16);
    out.y = e[2] | (e[3] << 16);
  });
  return out;
#else
  uint2 v;
  const unsigned addr = (unsigned)(uintptr_t)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
#endif
}
__device__ inline void ys_lds_tr_wait(uint2& a, uint2& b) {
#ifndef YS_EMU_BUILD
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
#else
  (void)a; (void)b;
#endif
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (loads and stores share the counter
// on gfx9-family parts), which would stall on the next tile's global prefetch at every barrier; this one waits for the
// wave's own LDS operations and leaves global loads in flight.  Use only where no global data is exchanged inside the
// workgroup across the barrier.
// s_waitcnt lgkmcnt(0): every LDS (and scalar-memory) operation of this wave has completed.  the anchors tie the wait to the values the
// reads produce, so that the compiler can move neither the reads below it nor their uses above it.
__device__ inline void ys_wait_lds_all(uint4& anchor, unsigned& anchor2) {
#ifndef YS_EMU_BUILD
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" : "+v"(anchor.x), "+v"(anchor.y), "+v"(anchor.z), "+v"(anchor.w), "+v"(anchor2) : : "memory");
#else
  (void)anchor; (void)anchor2;
#endif
}
__device__ inline void ys_barrier_lds() {
#ifdef YS_EMU_BUILD
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// LDS DMA (gfx950 global_load_lds_dwordx4): every lane names its own 16-byte global source; the destination is wave-uniform --
// lane l lands at lds_wave_base + 16 * l, no VGPR in between.  Completion is counted on vmcnt like a load: wait (YS_WAIT_VM0)
// and cross a workgroup barrier before another wave reads the bytes.  A swizzled LDS image is obtained by permuting the SOURCE
// addresses across lanes (the destination order is fixed).
__device__ inline void ys_glds16(const void* gsrc, void* lds_wave_base) {
#ifdef YS_EMU_BUILD
  memcpy((char*)lds_wave_base + emu::lane() * 16, gsrc, 16);
#else
  // Inline asm, not __builtin_amdgcn_global_load_lds: hipcc treats the builtin as a store to LDS that any later ds_read may alias
  // and puts s_waitcnt vmcnt(0) in front of the next fragment read -- the DMA of tile k+1 was drained before tile k was
  // multiplied (ISA of the first conv_gemm_kernel build), i.e. no overlap.  The asm form is invisible to that bookkeeping; the
  // kernels wait for it explicitly (YS_WAIT_VM0 + barrier).  M0 = LDS byte address of the destination, written in the same
  // statement that uses it and restored (the compiler owns M0).
  unsigned keep;
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
#endif
}

// LDS DMA through a buffer descriptor (buffer_load_dwordx4 ... offen lds): source = descriptor base + voffset (per lane, 32 bit) +
// soffset (scalar), destination as ys_glds16.  Measured on MI355X (tools/probe/probe_buflds.hip): a lane whose voffset + soffset
// is >= the descriptor's byte count gets ZEROS in its LDS slot -- so padding is an out-of-range offset (YS_BUF_OOB), not a
// 64-bit select against a zero line, and whatever is uniform across the wave moves from per-lane VALU into the scalar offset.
#define YS_BUF_OOB 0x80000000u               // descriptors are built with < 2^31 bytes
#ifdef YS_EMU_BUILD
struct ys_rsrc_t { const char* base; unsigned bytes; };
__device__ inline ys_rsrc_t ys_make_rsrc(const void* base, unsigned bytes) { ys_rsrc_t r; r.base = (const char*)base; r.bytes = bytes; return r; }
__device__ inline void ys_bufld_lds16(const ys_rsrc_t& r, unsigned voff, unsigned soff, void* lds_wave_base) {
  char* d = (char*)lds_wave_base + emu::lane() * 16;
  if ((unsigned long long)voff + soff + 16ull > (unsigned long long)r.bytes) memset(d, 0, 16);
  else memcpy(d, r.base + voff + soff, 16);
}
#else
typedef int ys_rsrc_t __attribute__((ext_vector_type(4)));
// base / bytes must be wave-uniform (kernel arguments and blockIdx-derived scalars): the descriptor lives in SGPRs
__device__ inline ys_rsrc_t ys_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  ys_rsrc_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((p >> 32) & 0xffffull));   // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ inline void ys_bufld_lds16(const ys_rsrc_t& r, unsigned voff, unsigned soff, void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds_wave_base);
  const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(r), "s"(so), "s"(dst) : "memory");
}
#endif

// The same without saving / restoring M0 (declared clobbered): two scalar instructions fewer per request where the request sits in the issue slots between MFMAs
// (conv_halo_kernel).  gfx9-family LDS instructions do not read M0, so hipcc has no standing value in it; the clobber covers the uses it does have (s_movrel, v_readlane).
__device__ inline void ys_bufld_lds16_nom0(const ys_rsrc_t& r, unsigned voff, unsigned soff, void* lds_wave_base) {
#ifdef YS_EMU_BUILD
  ys_bufld_lds16(r, voff, soff, lds_wave_base);
#else
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds_wave_base);
  const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               : : "v"(voff), "s"(r), "s"(so), "s"(dst) : "memory", "m0");
#endif
}

// 16-byte load through a buffer descriptor into registers: zeros for an out-of-range offset.  A compiler-visible load (builtin):
// hipcc counts it in its vmcnt bookkeeping like a global load.  The descriptor type is the compiler's own.
#ifdef YS_EMU_BUILD
typedef ys_rsrc_t ys_rsrcv_t;
__device__ inline ys_rsrcv_t ys_make_rsrcv(const void* base, unsigned bytes) { return ys_make_rsrc(base, bytes); }
__device__ inline uint4 ys_bufld16(const ys_rsrcv_t& r, unsigned voff) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if ((unsigned long long)voff + 16ull <= (unsigned long long)r.bytes) memcpy(&v, r.base + voff, 16);
  return v;
}
#else
typedef __amdgpu_buffer_rsrc_t ys_rsrcv_t;
__device__ inline ys_rsrcv_t ys_make_rsrcv(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)bytes, 0x00020000);
}
__device__ inline uint4 ys_bufld16(const ys_rsrcv_t& r, unsigned voff) {
  typedef unsigned ys_u32x4 __attribute__((ext_vector_type(4)));
  const ys_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  return make_uint4(q[0], q[1], q[2], q[3]);
}
#endif

// 16-byte store through a buffer descriptor: a lane with an out-of-range offset stores nothing (hardware drops it), so masked
// lanes need no branch and the number of store instructions per wave is static -- the compiler can count them in vmcnt.
__device__ inline void ys_bufst16(const ys_rsrcv_t& r, unsigned voff, const uint4& v) {
#ifdef YS_EMU_BUILD
  if ((unsigned long long)voff + 16ull <= (unsigned long long)r.bytes) memcpy((char*)r.base + voff, &v, 16);
#else
  typedef unsigned ys_u32x4 __attribute__((ext_vector_type(4)));
  ys_u32x4 q; q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
  __builtin_amdgcn_raw_buffer_store_b128(q, r, (int)voff, 0, 0);
#endif
}

// 8-byte forms (the direct epilogue of conv_epi.h: a lane owns 4 consecutive bf16 channels of a pixel after the MFMA)
__device__ inline uint2 ys_bufld8(const ys_rsrcv_t& r, unsigned voff) {
#ifdef YS_EMU_BUILD
  uint2 v; v.x = 0u; v.y = 0u;
  if ((unsigned long long)voff + 8ull <= (unsigned long long)r.bytes) memcpy(&v, r.base + voff, 8);
  return v;
#else
  typedef unsigned ys_u32x2 __attribute__((ext_vector_type(2)));
  const ys_u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, 0, 0);
  uint2 v; v.x = q[0]; v.y = q[1];
  return v;
#endif
}
__device__ inline void ys_bufst8(const ys_rsrcv_t& r, unsigned voff, const uint2& v) {
#ifdef YS_EMU_BUILD
  if ((unsigned long long)voff + 8ull <= (unsigned long long)r.bytes) memcpy((char*)r.base + voff, &v, 8);
#else
  typedef unsigned ys_u32x2 __attribute__((ext_vector_type(2)));
  ys_u32x2 q; q[0] = v.x; q[1] = v.y;
  __builtin_amdgcn_raw_buffer_store_b64(q, r, (int)voff, 0, 0);
#endif
}
// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane of the row: xor-1, xor-2 butterflies by quad_perm,
// then row_half_mirror and row_mirror -- four VALU adds, no LDS crossbar, a fixed tree (deterministic).  The interpreter's xor
// butterfly (1, 2, 4, 8) adds the same pairs, so both builds produce identical bits.
__device__ inline float ys_row16_sum(float v) {
#ifdef YS_EMU_BUILD
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
#else
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
#endif
}

static inline int ys_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Division of 0 <= n < 2^31 by a launch-constant divisor 1 <= d < 2^31 as one multiply-high and one shift (the tile loops of the blocked
// GEMM kernels turned every output row index into (image, row, column) with two integer divisions -- ~45 VALU instructions each,
// 16-24 of them per tile and thread: ~3 thousand cycles in front of a tile's first request).  For d >= 2: l = ceil(log2 d),
// mul = ceil(2^(31 + l) / d) < 2^32, n / d = (n * mul) >> (31 + l): the error term n * e / 2^(31 + l) stays below 1 / d for n < 2^31.
struct YsFastDiv { unsigned mul, sh, d; };
static inline YsFastDiv ys_fastdiv_make(unsigned d) {
  YsFastDiv f; f.d = d; f.mul = 0; f.sh = 0;
  if (d >= 2) {
    unsigned l = 0; while ((1ull << l) < d) l++;
    f.mul = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
    f.sh = l - 1;
  }
  return f;
}
__host__ __device__ inline unsigned ys_fastdiv(unsigned n, const YsFastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(YS_EMU_BUILD)
  const unsigned q = __umulhi(n, f.mul) >> f.sh;
#else
  const unsigned q = (unsigned)(((unsigned long long)n * f.mul) >> 32) >> f.sh;
#endif
  return f.d == 1 ? n : q;
}
