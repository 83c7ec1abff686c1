// probe_dma_rate.hip -- what one `buffer_load_dwordx4 ... offen lds` (1 KB LDS-DMA piece) costs on gfx950 (round 5):
//   * issue: cycles the ISSUING wave spends per piece when it issues P pieces back to back (s_memtime around the issue burst, no wait inside)
//   * sustained: pieces per microsecond and CU with W waves per CU each keeping P pieces in flight (issue P, wait vmcnt(0), repeat)
// for three source patterns: 0 = contiguous 1 KB per piece; 1 = 16 rows x 64 B (row pitch 5760 B: the weight rows of a 3x3 320-channel layer);
// 2 = every lane out of range (zeros, no memory access); and for the register path (global_load_dwordx4 -> VGPR -> ds_write_b128) as pattern 3.
// hipcc --offload-arch=gfx950 -O2 tools/probe/probe_dma_rate.hip -o build/probe_dma_rate && build/probe_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int P, int PAT>
__global__ void __launch_bounds__(256) k(const char* __restrict__ src, unsigned bytes, int rounds, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  i32x4 r;
  const unsigned long long p = (unsigned long long)src;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p); r[1] = __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xffff));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes); r[3] = 0x00020000;
  char* dst = lds + wave * (P * 1024);
  unsigned voff0;
  if (PAT == 0 || PAT == 3) voff0 = lane * 16;
  else if (PAT == 1) voff0 = (lane >> 2) * 5760 + (lane & 3) * 16;
  else if (PAT == 4) voff0 = (lane >> 3) * 5760 + (lane & 7) * 16;     // 8 rows x 128 B, pitch 5760 B (weight rows, 64-channel chunk)
  else if (PAT == 5) voff0 = (lane >> 3) * 640 + (lane & 7) * 16;      // 8 rows x 128 B, pitch 640 B (pixels of a 320-channel NHWC tensor)
  else voff0 = 0x80000000u;
  unsigned long long t_issue = 0, t0 = __builtin_readcyclecounter();
  unsigned base = (blockIdx.x * 4 + wave) * 65536u % (bytes / 2);
  uint4 regs[PAT == 3 ? P : 1];
  for (int it = 0; it < rounds; it++) {
    const unsigned long long a = __builtin_readcyclecounter();
    if (PAT == 3) {
#pragma unroll
      for (int j = 0; j < P; j++) regs[j] = *(const uint4*)(src + base + voff0 + j * 1024);
    } else {
#pragma unroll
      for (int j = 0; j < P; j++) {
        unsigned keep;
        const unsigned d = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(dst + j * 1024));
        const unsigned vo = PAT == 2 ? voff0 : base + voff0 + (PAT == 1 ? j * 16 * 5760u : PAT == 4 ? j * 8 * 5760u : PAT == 5 ? j * 8 * 640u : j * 1024u);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(r), "s"(0), "s"(d) : "memory");
      }
    }
    const unsigned long long b = __builtin_readcyclecounter();
    t_issue += b - a;
    if (PAT == 3) {
#pragma unroll
      for (int j = 0; j < P; j++) *(uint4*)(dst + j * 1024 + lane * 16) = regs[j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    base = (base + 4096u) % (bytes / 2);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t_issue; out[(blockIdx.x * 4 + wave) * 2 + 1] = t1 - t0; }
  if (threadIdx.x == 9999) out[0] = lds[threadIdx.x];
}
template <int P, int PAT> void run(const char* src, unsigned bytes, unsigned long long* out, int grid, int block, const char* name) {
  const int rounds = 200;
  hipFuncSetAttribute((const void*)k<P, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<P, PAT>), grid, block, (block / 64) * P * 1024, 0, src, bytes, 10, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<P, PAT>), grid, block, (block / 64) * P * 1024, 0, src, bytes, rounds, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * 4 * 2);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  const int waves = grid * (block / 64);
  double ti = 0, tt = 0; for (int w = 0; w < waves; w++) { ti += h[(w / (block / 64)) * 8 + (w % (block / 64)) * 2]; tt += h[(w / (block / 64)) * 8 + (w % (block / 64)) * 2 + 1]; }
  ti /= waves; tt /= waves;
  const double pieces = (double)waves * rounds * P;
  printf("%-34s grid %4d x %3d  P %2d: issue %6.1f cyc/piece/wave, round trip %7.0f cyc/round, %7.1f pieces/us/CU = %5.1f B/clk/CU (2.4 GHz), kernel %.1f us\n", name, grid, block, P,
         ti / rounds / P, tt / rounds, pieces / (ms * 1e3) / 256.0 * (grid < 256 ? 256.0 / grid : 1.0), pieces * 1024 / (ms * 1e-3) / std::min(grid, 256) / 2.4e9, ms * 1e3);
}
int main() {
  const unsigned bytes = 64u << 20;
  char* src; hipMalloc(&src, bytes + (1 << 20)); hipMemset(src, 1, bytes + (1 << 20));
  unsigned long long* out; hipMalloc(&out, 2048 * 4 * 2 * 8);
#define ALL(P_) \
  run<P_, 0>(src, bytes, out, 256, 64, "contiguous, 1 wave/CU"); run<P_, 0>(src, bytes, out, 256, 256, "contiguous, 4 waves/CU"); run<P_, 0>(src, bytes, out, 512, 256, "contiguous, 8 waves/CU"); \
  run<P_, 1>(src, bytes, out, 256, 64, "16 x 64 B rows, 1 wave/CU"); run<P_, 1>(src, bytes, out, 256, 256, "16 x 64 B rows, 4 waves/CU"); run<P_, 1>(src, bytes, out, 512, 256, "16 x 64 B rows, 8 waves/CU"); \
  run<P_, 4>(src, bytes, out, 256, 256, "8 x 128 B rows pitch 5760, 4 waves/CU"); run<P_, 4>(src, bytes, out, 512, 256, "8 x 128 B rows pitch 5760, 8 waves/CU"); \
  run<P_, 5>(src, bytes, out, 256, 256, "8 x 128 B rows pitch 640, 4 waves/CU"); run<P_, 5>(src, bytes, out, 512, 256, "8 x 128 B rows pitch 640, 8 waves/CU"); \
  run<P_, 2>(src, bytes, out, 256, 64, "out of range, 1 wave/CU"); run<P_, 2>(src, bytes, out, 512, 256, "out of range, 8 waves/CU"); \
  run<P_, 3>(src, bytes, out, 256, 64, "registers + ds_write, 1 wave/CU"); run<P_, 3>(src, bytes, out, 256, 256, "registers + ds_write, 4 waves/CU"); run<P_, 3>(src, bytes, out, 512, 256, "registers + ds_write, 8 waves/CU");
  ALL(2) ALL(4) ALL(8)
  return 0;
}
