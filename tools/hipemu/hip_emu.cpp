// hipemu runtime: block scheduler + the tiny HIP runtime surface (test-only, see hip_emu.h)
#include "hip_emu.h"
#include <omp.h>
#include <chrono>
#include <mutex>

#if EMU_FAST_SWITCH
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch,.-emu_switch
)");
#define EMU_SWAP(from, to) emu_switch(&(from), &(to))
#else
#define EMU_SWAP(from, to) swapcontext(&(from), &(to))
#endif

namespace emu {

static thread_local Block* g_blk = nullptr;
Block& blk() { return *g_blk; }

void yield() {
  Block& b = *g_blk;
  EMU_SWAP(b.cur->ctx, b.sched);
}

static void trampoline() {
  Block& b = *g_blk;
  (*b.fn)();
  Block& b2 = *g_blk;
  Thread* t = b2.cur;
  t->done = true;
  b2.alive--;
  b2.waves[t->lin >> 6].alive--;
  EMU_SWAP(t->ctx, b2.sched);  // never resumed
}

void barrier() {
  Block& b = blk();
  int gen = b.bar_gen;
  b.bar_arrived++;
  for (;;) {
    if (b.bar_gen != gen) return;
    if (b.bar_arrived >= b.alive) { b.bar_arrived = 0; b.bar_gen++; return; }
    yield();
  }
}

static void run_block(Block& b) {
  const int n = b.nthreads;
  const int nw = (n + 63) / 64;
  b.th.resize(n);
  b.waves.resize(nw);
  for (int w = 0; w < nw; w++) {
    Wave& wv = b.waves[w];
    wv.arrived[0] = wv.arrived[1] = 0; wv.gen[0] = wv.gen[1] = 0;
    wv.alive = std::min(64, n - w * 64);
  }
  b.alive = n;
  b.bar_arrived = 0;
  b.bar_gen = 0;
  if (b.stacks.size() < (size_t)n * b.stack_size) b.stacks.resize((size_t)n * b.stack_size);
  for (int i = 0; i < n; i++) {
    Thread& t = b.th[i];
    t.lin = i;
    t.tid.x = i % b.bdim.x;
    t.tid.y = (i / b.bdim.x) % b.bdim.y;
    t.tid.z = i / (b.bdim.x * b.bdim.y);
    t.done = false;
    t.par = 0;
#if EMU_FAST_SWITCH
    {   // initial frame: six callee-saved register slots, then the entry point as the return address of emu_switch
      uintptr_t top = ((uintptr_t)(b.stacks.data() + (size_t)(i + 1) * b.stack_size)) & ~(uintptr_t)15;
      void** sp = (void**)top;
      *--sp = nullptr;                      // fake return address of trampoline (it never returns)
      *--sp = (void*)trampoline;
      for (int r = 0; r < 6; r++) *--sp = nullptr;
      t.ctx.sp = sp;
    }
#else
    getcontext(&t.ctx);
    t.ctx.uc_stack.ss_sp = b.stacks.data() + (size_t)i * b.stack_size;
    t.ctx.uc_stack.ss_size = b.stack_size;
    t.ctx.uc_link = nullptr;
    makecontext(&t.ctx, (void (*)())trampoline, 0);
#endif
  }
  long sweeps = 0;
  while (b.alive > 0) {
    for (int i = 0; i < n; i++) {
      Thread& t = b.th[i];
      if (t.done) continue;
      b.cur = &t;
      EMU_SWAP(b.sched, t.ctx);
    }
    if (++sweeps > 200000000L) { fprintf(stderr, "hipemu: deadlock suspected\n"); abort(); }
  }
}

static thread_local std::vector<uint4>* g_dyn = nullptr;
unsigned char* dyn_lds() { return (unsigned char*)g_dyn->data(); }

void launch(dim3 grid, dim3 block, std::function<void()> fn, size_t dyn_lds_bytes) {
  const long total = (long)grid.x * grid.y * grid.z;
  const int nthreads = block.x * block.y * block.z;
  if (total <= 0 || nthreads <= 0) return;
#pragma omp parallel
  {
    static thread_local Block* tl = nullptr;
    if (!tl) { tl = new Block(); tl->stack_size = 96 * 1024; }
    Block& b = *tl;
    g_blk = &b;
    static thread_local std::vector<uint4> dynbuf;
    if (dynbuf.size() * 16 < dyn_lds_bytes + 16) dynbuf.resize(dyn_lds_bytes / 16 + 1);
    g_dyn = &dynbuf;
    b.fn = &fn;
    b.bdim = block;
    b.gdim = grid;
    b.nthreads = nthreads;
#pragma omp for schedule(dynamic, 1)
    for (long i = 0; i < total; i++) {
      b.bid.x = (unsigned)(i % grid.x);
      b.bid.y = (unsigned)((i / grid.x) % grid.y);
      b.bid.z = (unsigned)(i / ((long)grid.x * grid.y));
      run_block(b);
    }
  }
}

}  // namespace emu

// ---------------- HIP runtime surface ----------------
hipError_t hipMalloc(void** p, size_t n) {
  size_t sz = (n + 255) & ~size_t(255);
  if (sz == 0) sz = 256;
  void* q = aligned_alloc(256, sz);
  if (!q) return hipErrorOutOfMemory;
  memset(q, 0xFF, sz);  // poison: NaN for f32/bf16, catches uninitialised reads
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // launches are synchronous
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent_{0.0}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "hipemu (CPU SIMT interpreter, test-only)");
  strcpy(p->gcnArchName, "emu");
  p->multiProcessorCount = 8;
  p->totalGlobalMem = (size_t)32 << 30;
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)32 << 30; *t = (size_t)32 << 30; return hipSuccess; }
