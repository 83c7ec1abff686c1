// probe_pkmov.hip -- is `v_pk_mov_b32 ... op_sel` -> `s_nop 0` -> `v_pk_add_f32 / v_pk_fma_f32` safe on gfx950?
//
// Round 4, determinism bisection of the staged convolution epilogue (csrc/conv_epi.h): with the forward store loop fully unrolled
// hipcc (ROCm 7.2) accumulates the BatchNorm statistics of SOME iterations as
//     v_pk_mov_b32 v[38:39], v[30:31], v[24:25] op_sel:[1,0]      ; (elem1, elem0) of one bf16 dword, gathered from two pairs
//     s_nop 0
//     v_pk_add_f32 v[130:131], v[130:131], v[38:39]
//     v_pk_fma_f32 v[132:133], v[38:39], v[38:39], v[132:133]
//     v_pk_mov_b32 v[38:39], v[32:33], v[26:27] op_sel:[1,0]      ; next dword, same temporary pair
//     ...
// and the sums of exactly the low halves of those temporaries (elements 1 / 5 of each 8-channel vector) differ from run to run
// (3-6 of 6 reruns of a 160-wide blocked-GEMM layer), while the same build with the sums forced through single v_add_f32 /
// v_fmac_f32 (inline asm) is bit-stable, as is the default build whose iterations unpack straight into the operand pairs (no v_pk_mov).
// This probe runs that instruction pattern in isolation -- optionally next to waves that issue MFMAs on the same SIMDs, which is what
// the second resident workgroup does in the real kernel -- and checks every accumulator against a scalar replay.
//
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 tools/probe/probe_pkmov.hip -o /tmp/probe_pkmov && /tmp/probe_pkmov
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode bit 0: waves with an odd index spin on MFMAs instead (co-execution on the SIMD); bit 1: two s_nop instead of one
template <int NOPS>
__global__ void __launch_bounds__(256) k_pattern(float* out, int iters, int mfma_neighbours) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int wave = threadIdx.x >> 6;
  if (mfma_neighbours && (wave & 1)) {
    v8s a, b; v4f c = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 8; k++) { a[k] = (short)(0x3f80 + tid + k); b[k] = (short)(0x3f80 + k); }
    for (int i = 0; i < iters * 2; i++) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    if (c[0] == 12345.f) out[0] = c[1];
    return;
  }
  v2f s1a = {0.f, 0.f}, s2a = {0.f, 0.f}, s1b = {0.f, 0.f}, s2b = {0.f, 0.f};   // packed accumulators (pattern under test)
  float r1a0 = 0.f, r1a1 = 0.f, r2a0 = 0.f, r2a1 = 0.f, r1b0 = 0.f, r1b1 = 0.f, r2b0 = 0.f, r2b1 = 0.f;   // scalar replay
  for (int i = 0; i < iters; i++) {
    // two bf16 dwords per iteration, values in [-2, 2)
    const unsigned w0 = hash((unsigned)tid * 2654435761u + (unsigned)i * 2u), w1 = hash((unsigned)tid * 2654435761u + (unsigned)i * 2u + 1u);
    const unsigned d0 = ((w0 & 0x807f807fu) | 0x3f803f80u), d1 = ((w1 & 0x807f807fu) | 0x3f803f80u);
    // the compiler's unpack: lo -> pair A.lo, hi -> pair B.hi (pair B.lo holds junk, pair A.hi is not written yet)
    v2f pa0, pb0, pa1, pb1, t;
    pa0[0] = __builtin_bit_cast(float, d0 << 16); pa0[1] = 0.f;
    pb0[1] = __builtin_bit_cast(float, d0 & 0xffff0000u); pb0[0] = __builtin_bit_cast(float, d0 & 16u);
    pa1[0] = __builtin_bit_cast(float, d1 << 16); pa1[1] = 0.f;
    pb1[1] = __builtin_bit_cast(float, d1 & 0xffff0000u); pb1[0] = __builtin_bit_cast(float, d1 & 16u);
    if (NOPS == 1)
      asm volatile("v_pk_mov_b32 %4, %5, %6 op_sel:[1,0]\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %4\n\tv_pk_fma_f32 %1, %4, %4, %1\n\t"
                   "v_pk_mov_b32 %4, %7, %8 op_sel:[1,0]\n\ts_nop 0\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_fma_f32 %3, %4, %4, %3"
                   : "+v"(s1a), "+v"(s2a), "+v"(s1b), "+v"(s2b), "=&v"(t) : "v"(pb0), "v"(pa0), "v"(pb1), "v"(pa1));
    else
      asm volatile("v_pk_mov_b32 %4, %5, %6 op_sel:[1,0]\n\ts_nop 1\n\tv_pk_add_f32 %0, %0, %4\n\tv_pk_fma_f32 %1, %4, %4, %1\n\ts_nop 1\n\t"
                   "v_pk_mov_b32 %4, %7, %8 op_sel:[1,0]\n\ts_nop 1\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_fma_f32 %3, %4, %4, %3"
                   : "+v"(s1a), "+v"(s2a), "+v"(s1b), "+v"(s2b), "=&v"(t) : "v"(pb0), "v"(pa0), "v"(pb1), "v"(pa1));
    // replay: D.lo = src0.hi (element 1), D.hi = src1.lo (element 0)
    const float e1 = pb0[1], e0 = pa0[0], f1 = pb1[1], f0 = pa1[0];
    r1a0 += e1; r1a1 += e0; r2a0 = __builtin_fmaf(e1, e1, r2a0); r2a1 = __builtin_fmaf(e0, e0, r2a1);
    r1b0 += f1; r1b1 += f0; r2b0 = __builtin_fmaf(f1, f1, r2b0); r2b1 = __builtin_fmaf(f0, f0, r2b1);
  }
  float* o = out + (size_t)tid * 16;
  o[0] = s1a[0]; o[1] = s1a[1]; o[2] = s2a[0]; o[3] = s2a[1]; o[4] = s1b[0]; o[5] = s1b[1]; o[6] = s2b[0]; o[7] = s2b[1];
  o[8] = r1a0; o[9] = r1a1; o[10] = r2a0; o[11] = r2a1; o[12] = r1b0; o[13] = r1b1; o[14] = r2b0; o[15] = r2b1;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int blocks = 2048, threads = 256;
  const size_t n = (size_t)blocks * threads * 16;
  float* d; hipMalloc(&d, n * sizeof(float));
  std::vector<float> h(n);
  for (int nops = 1; nops <= 2; nops++)
    for (int nb = 0; nb <= 1; nb++) {
      hipMemset(d, 0, n * sizeof(float));
      if (nops == 1) hipLaunchKernelGGL(k_pattern<1>, dim3(blocks), dim3(threads), 0, 0, d, iters, nb);
      else hipLaunchKernelGGL(k_pattern<2>, dim3(blocks), dim3(threads), 0, 0, d, iters, nb);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
      long bad[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lanes = 0;
      for (size_t t = 0; t < (size_t)blocks * threads; t++) {
        if (nb && (((t % threads) >> 6) & 1)) continue;
        lanes++;
        for (int k = 0; k < 8; k++) if (memcmp(&h[t * 16 + k], &h[t * 16 + 8 + k], 4) != 0) bad[k]++;
      }
      printf("s_nop x%d, MFMA neighbours %d: lanes %ld x %d iterations; mismatching accumulators [s1.lo s1.hi s2.lo s2.hi | second pair]: %ld %ld %ld %ld | %ld %ld %ld %ld\n",
             nops, nb, lanes, iters, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6], bad[7]);
    }
  hipFree(d);
  return 0;
}
