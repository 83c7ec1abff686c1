// Probe (round 3): is a wave-private LDS row exchange -- rows written with ds_write_b64 by the lanes that own them in the MFMA
// accumulator layout, read back with ds_read_b128 by OTHER lanes of the same wave -- safe without an s_waitcnt between the writes and
// the reads, and with several reads in flight?  Mirrors the staged epilogue of conv_epi.h for a 64-pixel x 80-channel wave tile
// (row pitch 176 B).  Every value encodes (iteration, pixel, channel pair); a read that returns anything else is counted.
//   mode bit 0: s_waitcnt lgkmcnt(0) between the write phase and the read phase
//   mode bit 1: reads in batches of 4 (all issued before the first is checked) instead of one at a time
//   mode bit 2: s_waitcnt lgkmcnt(0) after each batch's reads are issued, before the first check
// usage: lds_row_exchange <iterations> ; prints mismatching 16-byte vectors per mode out of the total read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int MR = 4, NR = 5, BN = NR * 16, PITCH = (BN + 8) * 2, NPX = 16 * MR, VPP = BN / 8, PPI = 64 / VPP, NITER = (NPX + PPI - 1) / PPI;
__device__ inline unsigned enc(unsigned it, unsigned px, unsigned chpair) { return (it * 2654435761u) ^ (px << 20) ^ (chpair << 8) ^ 0x5a5a0000u; }
template <int MODE>
__global__ void __launch_bounds__(256) probe(unsigned long long* bad, int iters) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, q = lane >> 4;
  char* stg = lds + wave * (NPX * PITCH + 1024);
  const int cv = lane % VPP, pl = lane / VPP;
  const bool active = lane < PPI * VPP;
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; it++) {
    const unsigned tag = (unsigned)it * 977u + blockIdx.x;
#pragma unroll
    for (int mf = 0; mf < MR; mf++)
#pragma unroll
      for (int nf = 0; nf < NR; nf++) {
        const int px = mf * 16 + li, ch = nf * 16 + 4 * q;
        uint2 pk; pk.x = enc(tag, px, ch / 2); pk.y = enc(tag, px, ch / 2 + 1);
        *(uint2*)(stg + px * PITCH + ch * 2) = pk;
      }
    if (MODE & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int B = (MODE & 2) ? 4 : 1;
#pragma unroll
    for (int i0 = 0; i0 < NITER; i0 += B) {
      uint4 v[B];
#pragma unroll
      for (int j = 0; j < B; j++) {
        const int px = (i0 + j) * PPI + pl;
        const int pxs = px < NPX ? px : 0;
        v[j] = *(const uint4*)(stg + pxs * PITCH + cv * 16);
      }
      if (MODE & 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0].x) :: "memory");
#pragma unroll
      for (int j = 0; j < B; j++) {
        const int px = (i0 + j) * PPI + pl;
        if (active && px < NPX && i0 + j < NITER) {
          const unsigned c2 = cv * 4;
          const bool ok = v[j].x == enc(tag, px, c2) && v[j].y == enc(tag, px, c2 + 1) && v[j].z == enc(tag, px, c2 + 2) && v[j].w == enc(tag, px, c2 + 3);
          nbad += ok ? 0 : 1;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
  }
  if (nbad) atomicAdd(bad, nbad);
}
template <int MODE> static void run(int iters) {
  unsigned long long* d; hipMalloc(&d, 8); hipMemset(d, 0, 8);
  const size_t lds = 4 * (NPX * PITCH + 1024);
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  probe<MODE><<<768, 256, lds>>>(d, iters);
  unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const double total = 768.0 * 4 * iters * NPX * VPP;
  printf("mode %d (wait after writes %d, batch of 4 %d, full wait after reads %d): %llu bad of %.3g vectors\n", MODE, MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, h, total);
  hipFree(d);
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  run<0>(iters); run<1>(iters); run<2>(iters); run<3>(iters); run<6>(iters); run<7>(iters);
  return 0;
}
