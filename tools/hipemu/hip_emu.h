// hipemu -- a TEST-ONLY lockstep SIMT interpreter for the kernel sources in
// yolosharp_amd/csrc.  It exists because the development container has no GPU:
// compiling the *same* kernel source with g++ against this header lets the
// not-gpu test-suite execute every kernel (blocks, 64-lane waves, LDS, barriers,
// cross-lane shuffles and the two MFMA shapes we use) against the oracle.
//
// It is NOT a product path: the shipped library (libyolosharp_hip.so) is built by
// hipcc from the same sources with no CPU fallback of any kind, and the package
// never loads the emu build.  Only tests/ load libyolosharp_emu.so, by explicit path.
//
// Model: each GPU thread is a ucontext coroutine; all threads of a block run on one
// OS thread; blocks are distributed over OS threads with OpenMP.  Barriers and
// wave collectives are real rendezvous points (arrival counters), so divergent
// waves behave like hardware.
#pragma once
#include <ucontext.h>

// Coroutine switch.  glibc's swapcontext saves / restores the signal mask with a system call on every switch, which
// dominated the interpreter's run time; on x86-64 a 14-instruction switch of the callee-saved registers replaces it.
#if defined(__x86_64__)
#define EMU_FAST_SWITCH 1
struct emu_ctx { void* sp; };
extern "C" void emu_switch(emu_ctx* from, emu_ctx* to);
#else
#define EMU_FAST_SWITCH 0
#endif
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <vector>
#include <functional>
#include <algorithm>

#define YS_EMU 1

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static thread_local

namespace emu {

struct Thread {
#if EMU_FAST_SWITCH
  emu_ctx ctx;
#else
  ucontext_t ctx;
#endif
  dim3 tid;
  int lin;      // linear thread id in block
  bool done;
  int par;      // wave collective parity
};

struct Wave {
  alignas(16) unsigned char slot[2][64][128];  // per-lane deposit area (two generations)
  int arrived[2];
  int gen[2];
  int alive;
};

struct Block {
  dim3 bid, bdim, gdim;
  int nthreads;
  std::vector<Thread> th;
  std::vector<Wave> waves;
  Thread* cur;
#if EMU_FAST_SWITCH
  emu_ctx sched;
#else
  ucontext_t sched;
#endif
  int alive;
  int bar_arrived;
  int bar_gen;
  std::function<void()>* fn;
  std::vector<char> stacks;
  size_t stack_size;
};

Block& blk();
void yield();
void launch(dim3 grid, dim3 block, std::function<void()> fn, size_t dyn_lds_bytes = 0);
unsigned char* dyn_lds();   // dynamic LDS of the running block (16-byte aligned)

inline Thread& cur() { return *blk().cur; }
inline Wave& wave() { return blk().waves[cur().lin >> 6]; }
inline int lane() { return cur().lin & 63; }

// Generic wave rendezvous: every alive lane deposits `nbytes` (<=128) and, after all
// have arrived, `reader` may read any lane's deposit.
template <class F>
inline void wave_collective(const void* dep, int nbytes, F reader) {
  Thread& t = cur();
  Wave& w = wave();
  int p = t.par;
  t.par ^= 1;
  memcpy(w.slot[p][t.lin & 63], dep, nbytes);
  const int my_gen = w.gen[p];
  w.arrived[p]++;
  for (;;) {
    if (w.gen[p] != my_gen) break;                       // completed by another lane
    if (w.arrived[p] >= w.alive) { w.arrived[p] = 0; w.gen[p]++; break; }  // I complete it
    yield();
  }
  // slots of generation p stay valid until parity p is reused, which needs every alive lane to
  // have completed the next (p^1) collective, i.e. to have finished this read.
  reader(w.slot[p]);
}

void barrier();
}  // namespace emu

#define threadIdx (emu::cur().tid)
#define blockIdx (emu::blk().bid)
#define blockDim (emu::blk().bdim)
#define gridDim (emu::blk().gdim)
static const int warpSize = 64;

inline void __syncthreads() { emu::barrier(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

template <class T> inline T __shfl(T v, int src, int width = 64) {
  T r;
  int l = emu::lane();
  int s = (l & ~(width - 1)) | (src & (width - 1));
  emu::wave_collective(&v, sizeof(T), [&](unsigned char (*slot)[128]) { memcpy(&r, slot[s], sizeof(T)); });
  return r;
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
  T r;
  int s = emu::lane() ^ mask;
  emu::wave_collective(&v, sizeof(T), [&](unsigned char (*slot)[128]) { memcpy(&r, slot[s & 63], sizeof(T)); });
  return r;
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  T r;
  int l = emu::lane();
  int s = l + (int)d;
  if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l;
  emu::wave_collective(&v, sizeof(T), [&](unsigned char (*slot)[128]) { memcpy(&r, slot[s & 63], sizeof(T)); });
  return r;
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
  T r;
  int l = emu::lane();
  int s = l - (int)d;
  if (s < 0 || (s & ~(width - 1)) != (l & ~(width - 1))) s = l;
  emu::wave_collective(&v, sizeof(T), [&](unsigned char (*slot)[128]) { memcpy(&r, slot[s & 63], sizeof(T)); });
  return r;
}
inline unsigned long long __ballot(int pred) {
  unsigned long long r = 0;
  int v = pred ? 1 : 0;
  // lanes that already exited contribute 0 (their slot may be stale -> track by alive mask)
  struct D { int v; int tag; } d{v, 0x5a5a5a5a};
  emu::Wave& w = emu::wave();
  (void)w;
  emu::Block& b = emu::blk();
  int wbase = (emu::cur().lin >> 6) << 6;
  emu::wave_collective(&d, sizeof(d), [&](unsigned char (*slot)[128]) {
    for (int i = 0; i < 64; i++) {
      int lin = wbase + i;
      if (lin >= b.nthreads || b.th[lin].done) continue;
      D x; memcpy(&x, slot[i], sizeof(D));
      if (x.v) r |= (1ull << i);
    }
  });
  return r;
}
inline int __any(int p) { return __ballot(p) != 0; }
inline int __all(int p) {
  // all alive lanes
  unsigned long long m = __ballot(!p);
  return m == 0;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }

// ---- atomics on global memory (blocks run on several OS threads) ----
inline float atomicAdd(float* p, float v) {
  unsigned* up = (unsigned*)p;
  unsigned old = __atomic_load_n(up, __ATOMIC_RELAXED);
  for (;;) {
    float f; memcpy(&f, &old, 4);
    float nf = f + v; unsigned nu; memcpy(&nu, &nf, 4);
    if (__atomic_compare_exchange_n(up, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
  }
}
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicMax(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline int atomicCAS(int* p, int cmp, int v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return cmp;
}

// ---- math aliases ----
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float a) { return 1.0f / a; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }

// ---- tiny HIP runtime surface ----
typedef int hipError_t;
typedef struct emuStream_* hipStream_t;
typedef struct emuEvent_ { double t; }* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };

hipError_t hipMalloc(void** p, size_t n);
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned flags = 0) { return hipHostMalloc((void**)p, n, flags); }
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = 0);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = 0);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned);
#define hipDeviceMallocUncached 0x3
hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipDeviceSynchronize();
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = 0);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int dev);
hipError_t hipMemGetInfo(size_t* free_, size_t* total);
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
