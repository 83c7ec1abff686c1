// probe_barrier.hip -- cost of a workgroup barrier interval on gfx950 (development probe, not product code).
// 512 workgroups x 256 threads, LDS footprint chosen so that 2 workgroups share a CU; each iteration does one small LDS
// write + (a) __syncthreads, (b) s_waitcnt lgkmcnt(0) + s_barrier, (c) no barrier.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(int iters, float* out) {
  extern __shared__ float lds[];
  float acc = 0.f;
  for (int i = 0; i < iters; i++) {
    lds[threadIdx.x] = acc + (float)i;
    if (MODE == 0) __syncthreads();
    if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    acc += lds[(threadIdx.x + 64) & 255];
    if (MODE == 0) __syncthreads();
    if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (acc == 12345.f) out[0] = acc;
}

template <int MODE>
static void run(const char* name, int grid, size_t ldsb) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k<MODE><<<grid, 256, ldsb>>>(10, d);
  hipEventRecord(a);
  k<MODE><<<grid, 256, ldsb>>>(iters, d);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s grid %4d lds %6zu: %.1f ns per iteration (2 barrier intervals)\n", name, grid, ldsb, ms * 1e6 / iters);
  hipFree(d);
}

int main() {
  for (size_t l : {(size_t)8 * 1024, (size_t)64 * 1024}) {
    for (int g : {256, 512, 1536}) {
      run<0>("__syncthreads", g, l);
      run<1>("lgkmcnt(0)+s_barrier", g, l);
      run<2>("no barrier", g, l);
    }
  }
  return 0;
}
