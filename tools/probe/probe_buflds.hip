// probe_buflds.hip -- semantics of `buffer_load_dwordx4 ... offen lds` (LDS DMA through a buffer descriptor) on gfx950:
//   * address = descriptor base + voffset (per lane) + soffset (scalar); LDS destination = M0 + 16 * lane
//   * a lane whose voffset is >= num_records: does its LDS slot receive zeros, or keep its old bytes?
// hipcc --offload-arch=gfx950 -O2 tools/probe/probe_buflds.hip -o build/probe_buflds && build/probe_buflds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* __restrict__ src, unsigned* dst, unsigned bytes, int soff) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xAAAAAAAAu;
  __syncthreads();
  i32x4 r;
  const unsigned long long p = (unsigned long long)src;
  r[0] = (int)(unsigned)p; r[1] = (int)((p >> 32) & 0xffff); r[2] = (int)bytes; r[3] = 0x00020000;
  unsigned keep;
  const unsigned dstl = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds);
  unsigned voff = (63 - threadIdx.x) * 16;              // reversed: lane l fetches unit 63 - l
  if (threadIdx.x % 5 == 4) voff = 0x80000000u;         // out of range
  if (threadIdx.x == 7) voff = bytes - 16;              // last valid unit
  if (threadIdx.x == 9) voff = bytes - 16 - soff + 16;  // voffset in range, voffset + soffset past the end
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(r), "s"(soff), "s"(dstl) : "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) dst[i] = lds[i];
}
int main() {
  const unsigned n = 1024;                               // dwords in the buffer = 4096 bytes
  std::vector<unsigned> h(n + 64);
  for (unsigned i = 0; i < n + 64; i++) h[i] = i;        // dword value = its index
  unsigned *d_src, *d_dst;
  hipMalloc(&d_src, (n + 64) * 4); hipMalloc(&d_dst, 256 * 4);
  hipMemcpy(d_src, h.data(), (n + 64) * 4, hipMemcpyHostToDevice);
  const int soff = 1024;
  hipLaunchKernelGGL(k, 1, 64, 0, 0, d_src, d_dst, n * 4, soff);
  std::vector<unsigned> o(256);
  hipMemcpy(o.data(), d_dst, 256 * 4, hipMemcpyDeviceToHost);
  for (int l = 0; l < 12; l++) printf("lane %2d: %08x %08x %08x %08x\n", l, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
  printf("expect lane 0: dword index (63*16 + 1024)/4 = %u; lane 4 (OOB): zeros or aaaaaaaa?; lane 7: %u (voff + soff past end?); lane 9\n", (63 * 16 + 1024) / 4, (n * 4 - 16 + 1024) / 4);
  return 0;
}
