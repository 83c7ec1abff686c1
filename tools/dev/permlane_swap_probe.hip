// probe: lane mapping of v_permlane16_swap_b32 on gfx950 (which 16-lane rows of the two operands trade places)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
  unsigned a = 0x100u + threadIdx.x, b = 0x200u + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  p[threadIdx.x] = r[0]; p[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int w = 0; w < 2; w++) { printf("out%d:", w); for (int l = 0; l < 64; l += 8) printf(" [%d]=%x", l, h[w * 64 + l]); printf("\n"); }
  return 0;
}
