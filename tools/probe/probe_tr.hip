// probe_tr.hip -- empirical lane/address semantics of ds_read_b64_tr_b16 on gfx950 (development probe, not product code).
// Every lane supplies the byte address of 4 contiguous 16-bit elements; we fill LDS with element indices and print,
// for each lane, which elements come back.  Build: hipcc --offload-arch=gfx950 -O2 probe_tr.hip -o probe_tr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(const int* lane_elem_off, uint16_t* out) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const unsigned addr = (unsigned)(uintptr_t)(lds + lane_elem_off[threadIdx.x]);
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = v.x & 0xffff;
  out[threadIdx.x * 4 + 1] = v.x >> 16;
  out[threadIdx.x * 4 + 2] = v.y & 0xffff;
  out[threadIdx.x * 4 + 3] = v.y >> 16;
}

int main() {
  int h_off[64];
  uint16_t h_out[256];
  int* d_off; uint16_t* d_out;
  hipMalloc(&d_off, sizeof(h_off)); hipMalloc(&d_out, sizeof(h_out));
  for (int exp = 0; exp < 2; exp++) {
    // exp 0: lane l -> 1000*(l>>4) + 100*((l&15)>>2) + 4*((l&15)&3): "row" = (l&15)>>2 (pitch 100), col4 = l&3
    // exp 1: lane l -> 64*l (every lane far apart) to see which lane's address each returned element comes from
    for (int l = 0; l < 64; l++) h_off[l] = exp == 0 ? 1000 * (l >> 4) + 100 * ((l & 15) >> 2) + 4 * (l & 3) : 64 * l;
    hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_off, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("experiment %d\n", exp);
    for (int l = 0; l < 64; l++) printf("lane %2d (addr elem %4d): %4d %4d %4d %4d\n", l, h_off[l], h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
  }
  return 0;
}
