#include <stdio.h>
#include <stdlib.h>
#include <string.h>
// TEST INFRASTRUCTURE ONLY - a host stand-in for <hip/hip_runtime.h>.
//
// The product sources under dynaboa_amd/csrc/*.hip are plain HIP for gfx950 and contain no
// conditional compilation.  To check their indexing / synchronisation logic in the build
// container (which has no GPU) the `-m "not gpu"` tests compile those same files with the host
// clang++ and `-I tests/emu/include`, so that this header is found instead of the real one.
// Every workgroup is executed by cooperative fibers (one per work-item, see emu_runtime.cpp), workgroups of a
// grid on a few OS threads (hence __shared__ = static thread_local):
// __syncthreads(), wave-64 shuffles and the 32x32x2 fp32 MFMA are emulated with the hardware's
// lane -> element maps.  The resulting library (tests/emu/_build/libdynaboa_emu.so) is loaded
// only by tests; dynaboa_amd/_lib.py never looks for it.
#pragma once
#include <functional>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __shared__ static thread_local
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 {
  float x, y, z, w;
};
struct float2 {
  float x, y;
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
static inline hipError_t hipGetLastError() { return hipSuccess; }
// Streams.  By default every operation runs at once, in issue order (one implicit queue).  In "lazy" mode (emu_lazy(1), tests only)
// operations are queued per stream and run when emu_flush(order) drains the queues - one stream to completion first, the others only
// as far as its event waits demand - so that a missing cross-stream wait shows up as a wrong result: order 0 runs the first-used
// stream as early and the others as late as the recorded waits allow, order 1 the other way round (see emu_runtime.cpp).
namespace emu {
struct Event;
void submit(void* stream, std::function<void()> op, const char* what = "copy");
Event* event_new();
void event_delete(Event* e);
void event_record(Event* e, void* stream);
void stream_wait(void* stream, Event* e);
}  // namespace emu
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
  emu::submit(st, [=]() { memcpy(d, s, n); });
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
  emu::submit(st, [=]() { memset(d, v, n); });
  return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind,
                                          hipStream_t st) {
  emu::submit(st, [=]() {
    for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
  });
  return hipSuccess;
}

typedef emu::Event* hipEvent_t;
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = emu::event_new(); return hipSuccess; }
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, sync) ((void)0)
static inline long long clock64() { return 0; }
// raw buffer loads (stride 0): an offset whose per-lane part is >= num_records reads as 0.  Whether the hardware also adds the
// scalar offset into that range check is not relied upon by the kernels, so the emulator aborts if an in-range per-lane offset
// plus the scalar offset leaves the buffer.
struct __amdgpu_buffer_rsrc_t { const char* p; unsigned n; };
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int n, int) {
  return __amdgpu_buffer_rsrc_t{reinterpret_cast<const char*>(p), (unsigned)n};
}
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));
static inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  emu_u32x4 v = {0u, 0u, 0u, 0u};
  if ((unsigned)voff >= r.n) return v;
  const unsigned long long off = (unsigned long long)(unsigned)voff + (unsigned)soff;
  if (off + 16 > r.n) { fprintf(stderr, "emu: buffer load leaves the buffer through the scalar offset (%llu + 16 > %u)\n", off, r.n); abort(); }
  memcpy(&v, r.p + off, 16);
  return v;
}
// raw buffer store: a per-lane offset >= num_records is dropped (cache-policy bits ignored)
static inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  if ((unsigned)voff >= r.n) return;
  const unsigned long long off = (unsigned long long)(unsigned)voff + (unsigned)soff;
  if (off + 16 > r.n) { fprintf(stderr, "emu: buffer store leaves the buffer (%llu + 16 > %u)\n", off, r.n); abort(); }
  memcpy(const_cast<char*>(r.p) + off, &v, 16);
}
static inline void __threadfence_system() {}
// Inter-workgroup hand-offs (the one-pass GroupNorm backward: a few workgroups of consecutive ids meet on a counter).  The
// emulator claims workgroups in id order on a pool of OS threads (emu_runtime.cpp: at least 8), so a workgroup that polls for
// peers with nearby ids makes progress as on the device; s_sleep yields the OS thread.
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
// (__hip_atomic_load is a clang builtin in host mode as well)
void emu_os_yield();
static inline void __builtin_amdgcn_s_sleep(int) { emu_os_yield(); }
static inline long long wall_clock64() { return 0; }
#define __builtin_amdgcn_s_waitcnt(imm) ((void)0)
static inline long long __double_as_longlong(double d) {
  long long v;
  std::memcpy(&v, &d, sizeof v);
  return v;
}
static inline double __longlong_as_double(long long v) {
  double d;
  std::memcpy(&d, &v, sizeof d);
  return d;
}
static inline float __uint_as_float(unsigned u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
// device globals are plain globals here; the copy runs in stream order like every other operation
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipMemcpyFromSymbolAsync(void* d, const void* sym, size_t n, size_t off, hipMemcpyKind, hipStream_t st) {
  emu::submit(st, [=]() { memcpy(d, (const char*)sym + off, n); });
  return hipSuccess;
}
void emu_stream_synchronize(void* st);
static inline hipError_t hipStreamSynchronize(hipStream_t st) { emu_stream_synchronize(st); return hipSuccess; }
// streams of the library's own (the stepper's parallel passes): a stream is just a queue key here
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = malloc(8); return *s ? hipSuccess : 1; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = emu::event_new(); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { emu::event_delete(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) { emu::event_record(e, st); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned) { emu::stream_wait(st, e); return hipSuccess; }

// graphs are not emulated: capture reports failure and the engine stays on its eager path
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return 1; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return 1; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return 1; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

namespace emu {
extern thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
void syncthreads();
void wave_exchange(const void* mine, void* theirs, int src_lane, int bytes);
void mfma_32x32x2(float a, float b, float* c16);
void mfma_32x32x16_bf16(const unsigned short* a8, const unsigned short* b8, float* c16);
int lane_id();
void run_grid(dim3 grid, dim3 block, void (*call)(void*), void* ctx);
template <class F>
void launch(dim3 grid, dim3 block, F body) {
  run_grid(grid, block, [](void* p) { (*static_cast<F*>(p))(); }, &body);
}
// kernel arguments are taken by value at the launch call (as a real launch copies them into the kernel-argument segment)
template <class K, class... A>
void launch_on(const char* what, void* stream, dim3 grid, dim3 block, K kernel, A... args) {
  submit(stream, [=]() { launch(grid, block, [=]() { kernel(args...); }); }, what);
}
}  // namespace emu

#define threadIdx (emu::g_threadIdx)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)
#define __syncthreads() emu::syncthreads()

template <class T>
static inline T __shfl_xor(T v, int mask) {
  T r;
  emu::wave_exchange(&v, &r, emu::lane_id() ^ mask, (int)sizeof(T));
  return r;
}
template <class T>
static inline T __shfl(T v, int lane) {
  T r;
  emu::wave_exchange(&v, &r, lane, (int)sizeof(T));
  return r;
}
template <class T>
static inline T __shfl_down(T v, int d) {
  T r;
  int src = emu::lane_id() + d;
  emu::wave_exchange(&v, &r, src > 63 ? emu::lane_id() : src, (int)sizeof(T));
  return r;
}
// v_mov_b32_dpp with a quad_perm control (ctrl < 0x100): lane l reads lane (l & ~3) | ((ctrl >> 2 (l & 3)) & 3)
static inline int __builtin_amdgcn_update_dpp(int, int src, int ctrl, int, int, bool) {
  if (ctrl >= 0x100) { fprintf(stderr, "emu: only quad_perm DPP controls are emulated\n"); abort(); }
  const int l = emu::lane_id();
  int r;
  emu::wave_exchange(&src, &r, (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3), 4);
  return r;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z)                          \
  ({                                                                                    \
    float emu_c_[16];                                                                   \
    for (int emu_i_ = 0; emu_i_ < 16; ++emu_i_) emu_c_[emu_i_] = (c)[emu_i_];           \
    emu::mfma_32x32x2((a), (b), emu_c_);                                                \
    __typeof__(c) emu_r_;                                                               \
    for (int emu_i_ = 0; emu_i_ < 16; ++emu_i_) emu_r_[emu_i_] = emu_c_[emu_i_];        \
    emu_r_;                                                                             \
  })

// v_mfma_f32_32x32x16_bf16: A fragment = row lane&31, eight consecutive k at 8*(lane>>5); B likewise by column
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)                       \
  ({                                                                                    \
    float emu_c_[16];                                                                   \
    unsigned short emu_a_[8], emu_b_[8];                                                \
    __typeof__(a) emu_av_ = (a);                                                        \
    __typeof__(b) emu_bv_ = (b);                                                        \
    memcpy(emu_a_, &emu_av_, 16);                                                       \
    memcpy(emu_b_, &emu_bv_, 16);                                                       \
    for (int emu_i_ = 0; emu_i_ < 16; ++emu_i_) emu_c_[emu_i_] = (c)[emu_i_];           \
    emu::mfma_32x32x16_bf16(emu_a_, emu_b_, emu_c_);                                    \
    __typeof__(c) emu_r_;                                                               \
    for (int emu_i_ = 0; emu_i_ < 16; ++emu_i_) emu_r_[emu_i_] = emu_c_[emu_i_];        \
    emu_r_;                                                                             \
  })
struct uint2 {
  unsigned x, y;
};
static inline unsigned __float_as_uint(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return u;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch_on(#kernel, (void*)(stream), (grid), (block), kernel, __VA_ARGS__)
