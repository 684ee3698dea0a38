// TEST INFRASTRUCTURE ONLY - cooperative-fiber executor behind tests/emu/include/hip/hip_runtime.h.
// One fiber per work-item; a workgroup's fibers run round-robin on the calling OS thread and
// yield at __syncthreads() / cross-lane operations, which gives exactly the barrier semantics the
// kernels rely on.  The workgroups of a grid are spread over a few OS threads (DYB_EMU_THREADS, default = the CPU
// count, max 16): every piece of executor state, the built-in index variables and the kernels' __shared__ arrays are
// thread_local.  Workgroups are claimed in id order, so the few kernels whose workgroups meet on a counter (groups of <= 64
// consecutive ids; the pool has 72 threads) make progress as they do on the device.  x86-64 SysV only.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <unistd.h>

#include <atomic>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

// Run permits: the pool has many more threads (72) than cores because the workgroups of a meeting (kernels whose workgroups wait for each
// other on a counter) must all be claimed at once; only `permits` of them run workgroups at any time, and a workgroup that sleeps in a
// wait loop (s_sleep -> emu_os_yield) lends its permit to another thread meanwhile.
namespace emu_permits {
static std::mutex& mu = *new std::mutex();
static std::condition_variable& cv = *new std::condition_variable();
static int avail = -1;
static thread_local bool held = false;
static void init_locked() {
  if (avail >= 0) return;
  long v = 2 * sysconf(_SC_NPROCESSORS_ONLN);         // (fibers of a workgroup switch often: a little oversubscription pays)
  if (const char* e = getenv("DYB_EMU_PERMITS")) v = atol(e);
  avail = (int)(v < 4 ? 4 : v > 32 ? 32 : v);
}
static void acquire() {
  std::unique_lock<std::mutex> lk(mu);
  init_locked();
  cv.wait(lk, [] { return avail > 0; });
  --avail;
  held = true;
}
static void release() {
  {
    std::lock_guard<std::mutex> lk(mu);
    ++avail;
    held = false;
  }
  cv.notify_one();
}
}  // namespace emu_permits
void emu_os_yield() {
  if (!emu_permits::held) { std::this_thread::yield(); return; }
  emu_permits::release();
  std::this_thread::yield();
  emu_permits::acquire();
}

namespace emu {
thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_ctx_switch, .-emu_ctx_switch
)");

enum { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
static const size_t kStack = 128 * 1024;
static const int kMaxThreads = 1024;

struct Fiber {
  void* sp;
  int state;
  dim3 tid;
  int lane, wave;
  unsigned long long wait_gen;
};
struct Wave {
  int alive, arrived;
  unsigned long long gen;
  unsigned char xbuf[2][64][16];
  unsigned long long dep_gen[2][64];   // generation in which each lane last deposited (+1; 0 = never)
  float a[2][64], b[2][64];
  float a8[2][64][8], b8[2][64][8];
};

static thread_local char* g_stacks = nullptr;
static thread_local Fiber g_f[kMaxThreads];
static thread_local Wave g_w[kMaxThreads / 64];
static thread_local void* g_sched_sp;
static thread_local int g_cur, g_n, g_alive, g_arrived;
static thread_local unsigned long long g_gen;
static thread_local void (*g_call)(void*);
static thread_local void* g_ctx;

static void yield_to_sched() { emu_ctx_switch(&g_f[g_cur].sp, g_sched_sp); }

static void release_block_if_complete() {
  if (g_alive > 0 && g_arrived == g_alive) {
    g_arrived = 0;
    ++g_gen;
  }
}
static void release_wave_if_complete(Wave& w) {
  if (w.alive > 0 && w.arrived == w.alive) {
    w.arrived = 0;
    ++w.gen;
  }
}

static void fiber_main() {
  g_call(g_ctx);
  Fiber& f = g_f[g_cur];
  f.state = DONE;
  --g_alive;
  Wave& w = g_w[f.wave];
  --w.alive;
  release_block_if_complete();   // exited work-items do not take part in later barriers
  release_wave_if_complete(w);
  yield_to_sched();
  fprintf(stderr, "emu: resumed a finished fiber\n");
  abort();
}

void syncthreads() {
  Fiber& f = g_f[g_cur];
  ++g_arrived;
  if (g_arrived == g_alive) {
    g_arrived = 0;
    ++g_gen;
    return;
  }
  f.wait_gen = g_gen;
  f.state = WAIT_BLOCK;
  yield_to_sched();
}

static void wave_sync(Wave& w) {
  Fiber& f = g_f[g_cur];
  ++w.arrived;
  if (w.arrived == w.alive) {
    w.arrived = 0;
    ++w.gen;
    return;
  }
  f.wait_gen = w.gen;
  f.state = WAIT_WAVE;
  yield_to_sched();
}

int lane_id() { return g_f[g_cur].lane; }

void wave_exchange(const void* mine, void* theirs, int src_lane, int bytes) {
  Fiber& f = g_f[g_cur];
  Wave& w = g_w[f.wave];
  const unsigned long long op = w.gen;
  int par = (int)(op & 1);
  memcpy(w.xbuf[par][f.lane], mine, bytes);
  w.dep_gen[par][f.lane] = op + 1;
  wave_sync(w);
  // a partner is valid iff it took part in THIS exchange (it may have finished the kernel since);
  // lanes that never reached it (exited earlier / beyond the block) read as the caller's own value
  int src = (src_lane >= 0 && src_lane < 64 && w.dep_gen[par][src_lane] == op + 1) ? src_lane : f.lane;
  memcpy(theirs, w.xbuf[par][src], bytes);
}

void mfma_32x32x2(float a, float b, float* c) {
  Fiber& f = g_f[g_cur];
  Wave& w = g_w[f.wave];
  int par = (int)(w.gen & 1);
  w.a[par][f.lane] = a;
  w.b[par][f.lane] = b;
  wave_sync(w);
  // A[i][k] is held by lane i + 32k, B[k][j] by lane j + 32k; D[i][j] in lane j + 32*((i>>2)&1),
  // register (i&3) + 4*(i>>3)  <=>  row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
  int j = f.lane & 31, hi = f.lane >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    acc = fmaf(w.a[par][i], w.b[par][j], acc);
    acc = fmaf(w.a[par][i + 32], w.b[par][j + 32], acc);
    c[r] = acc;
  }
}

static inline float bf16_to_f32(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
void mfma_32x32x16_bf16(const unsigned short* a8, const unsigned short* b8, float* c) {
  Fiber& f = g_f[g_cur];
  Wave& w = g_w[f.wave];
  int par = (int)(w.gen & 1);
  for (int i = 0; i < 8; ++i) { w.a8[par][f.lane][i] = bf16_to_f32(a8[i]); w.b8[par][f.lane][i] = bf16_to_f32(b8[i]); }
  wave_sync(w);
  // A[i][k = 8*kb + e] is element e of lane i + 32*kb, B[k][j] element e of lane j + 32*kb; D as for every 32x32 shape
  int j = f.lane & 31, hi = f.lane >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int kb = 0; kb < 2; ++kb)
      for (int e = 0; e < 8; ++e) acc = fmaf(w.a8[par][i + 32 * kb][e], w.b8[par][j + 32 * kb][e], acc);
    c[r] = acc;
  }
}

static void init_fiber(int idx) {
  char* top = g_stacks + (size_t)(idx + 1) * kStack;
  void** sp = reinterpret_cast<void**>(top);
  *--sp = nullptr;                                   // fake return address of fiber_main
  *--sp = reinterpret_cast<void*>(&fiber_main);      // `ret` target of the first switch
  for (int i = 0; i < 6; ++i) *--sp = nullptr;       // rbp rbx r12 r13 r14 r15
  g_f[idx].sp = sp;
}

static void run_block(dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz, void (*call)(void*), void* ctx) {
  const int n = (int)(block.x * block.y * block.z);
  if (!g_stacks) {
    g_stacks = static_cast<char*>(mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE,
                                       MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (g_stacks == MAP_FAILED) abort();
  }
  g_call = call;
  g_ctx = ctx;
  g_blockDim = block;
  g_gridDim = grid;
  g_blockIdx = dim3(bx, by, bz);
  g_n = g_alive = n;
  g_arrived = 0;
  g_gen = 0;
  int nw = (n + 63) / 64;
  for (int w = 0; w < nw; ++w) {
    int cnt = n - w * 64;
    g_w[w].alive = cnt > 64 ? 64 : cnt;
    g_w[w].arrived = 0;
    g_w[w].gen = 0;
    memset(g_w[w].dep_gen, 0, sizeof(g_w[w].dep_gen));
  }
  for (int i = 0; i < n; ++i) {
    init_fiber(i);
    g_f[i].state = READY;
    g_f[i].lane = i & 63;
    g_f[i].wave = i >> 6;
    g_f[i].tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
  }
  int done = 0;
  while (done < n) {
    int progressed = 0;
    for (int i = 0; i < n; ++i) {
      Fiber& f = g_f[i];
      if (f.state == DONE) continue;
      if (f.state == WAIT_BLOCK) {
        if (g_gen == f.wait_gen) continue;
        f.state = READY;
      } else if (f.state == WAIT_WAVE) {
        if (g_w[f.wave].gen == f.wait_gen) continue;
        f.state = READY;
      }
      g_cur = i;
      g_threadIdx = f.tid;
      emu_ctx_switch(&g_sched_sp, f.sp);
      ++progressed;
      if (f.state == DONE) ++done;
    }
    if (!progressed) {
      fprintf(stderr, "emu: deadlock in block (%u,%u,%u): %d of %d work-items finished\n", bx, by, bz, done, n);
      abort();
    }
  }
}

// ---- a small persistent pool: the launching thread publishes one grid at a time and takes part itself ----
struct Job {
  dim3 grid, block;
  void (*call)(void*);
  void* ctx;
  unsigned long long total;
};
// heap objects that are never destroyed: the pool threads are detached and still wait on them when the process exits
// (destroying a condition variable with waiters blocks forever)
static std::mutex& g_mu = *new std::mutex();                    // serialises launches (several host threads may drive the library)
static std::mutex& g_pool_mu = *new std::mutex();
static std::condition_variable &g_cv_work = *new std::condition_variable(), &g_cv_done = *new std::condition_variable();
static std::vector<std::thread>& g_pool = *new std::vector<std::thread>();
static Job g_job;
static unsigned long long g_job_id = 0;
static std::atomic<unsigned long long> g_next{0};
static int g_active = 0;

static void work_on(const Job& j) {
  for (;;) {
    emu_permits::acquire();                          // before the claim: workgroups start in id order
    struct Rel { ~Rel() { emu_permits::release(); } } rel;
    unsigned long long b = g_next.fetch_add(1, std::memory_order_relaxed);
    if (b >= j.total) return;
    unsigned bx = (unsigned)(b % j.grid.x), by = (unsigned)((b / j.grid.x) % j.grid.y), bz = (unsigned)(b / ((unsigned long long)j.grid.x * j.grid.y));
    run_block(j.grid, j.block, bx, by, bz, j.call, j.ctx);
  }
}
static void pool_main() {
  unsigned long long seen = 0;
  for (;;) {
    Job j;
    {
      std::unique_lock<std::mutex> lk(g_pool_mu);
      g_cv_work.wait(lk, [&] { return g_job_id != seen; });
      seen = g_job_id;
      j = g_job;
    }
    work_on(j);
    {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      if (--g_active == 0) g_cv_done.notify_all();
    }
  }
}
static int pool_size() {
  static int n = [] {
    const char* e = getenv("DYB_EMU_THREADS");
    long v = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    if (v > 16) v = 16;
    if (v < 72) v = 72;                // kernels that meet on a counter need every workgroup of a meeting claimed (<= 64 of consecutive ids + the
                                       // 4 per-group workgroups of the GroupNorm tangent's backward that wait for every image)
    return (int)v;
  }();
  return n;
}

void run_grid(dim3 grid, dim3 block, void (*call)(void*), void* ctx) {
  int n = (int)(block.x * block.y * block.z);
  if (n <= 0 || n > kMaxThreads) {
    fprintf(stderr, "emu: bad block size %d\n", n);
    abort();
  }
  std::lock_guard<std::mutex> launch_lock(g_mu);
  Job j{grid, block, call, ctx, (unsigned long long)grid.x * grid.y * grid.z};
  const int helpers = pool_size() - 1;
  if (helpers <= 0 || j.total < 2) {
    g_next.store(0);
    work_on(j);
    return;
  }
  if (g_pool.empty())
    for (int i = 0; i < helpers; ++i) { g_pool.emplace_back(pool_main); g_pool.back().detach(); }
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_next.store(0);
    g_job = j;
    g_active = helpers;
    ++g_job_id;
  }
  g_cv_work.notify_all();
  work_on(j);
  std::unique_lock<std::mutex> lk(g_pool_mu);
  g_cv_done.wait(lk, [&] { return g_active == 0; });
}

// ---- streams: immediate execution, or per-stream queues drained by emu_flush (see hip_runtime.h) ----------------------------------
struct Event {
  unsigned long long submitted = 0, completed = 0;
};
namespace {
struct Op {
  int kind;                            // 0 work, 1 record, 2 wait
  std::function<void()> fn;
  Event* ev;
  unsigned long long gen;
  const char* what = "";
};
bool trace_on() {
  static const bool v = getenv("EMU_STREAM_TRACE") != nullptr;
  return v;
}
struct Queue {
  void* stream;
  std::deque<Op> ops;
  bool draining = false;
};
std::mutex& q_mu = *new std::mutex();
bool g_lazy = false;
std::vector<Queue*>& g_queues = *new std::vector<Queue*>();      // in order of first use

Queue& queue_of(void* stream) {
  for (Queue* q : g_queues)
    if (q->stream == stream) return *q;
  g_queues.push_back(new Queue{stream});
  return *g_queues.back();
}
// run q until its record of (until, gen) has executed (until == nullptr: to the end)
void drain(Queue& q, Event* until, unsigned long long gen) {
  if (q.draining) { fprintf(stderr, "emu: circular wait between streams\n"); abort(); }
  q.draining = true;
  while (!q.ops.empty()) {
    Op op = std::move(q.ops.front());
    q.ops.pop_front();
    if (trace_on())
      fprintf(stderr, "emu stream %p: %s %s ev %p gen %llu\n", q.stream, op.kind == 0 ? "run" : op.kind == 1 ? "record" : "wait", op.what,
              (void*)op.ev, op.gen);
    if (op.kind == 0) {
      op.fn();
    } else if (op.kind == 1) {
      op.ev->completed = op.gen;
      if (op.ev == until && op.gen >= gen) break;
    } else if (op.ev->completed < op.gen) {
      Queue* src = nullptr;
      for (Queue* o : g_queues)
        for (const Op& c : o->ops)
          if (c.kind == 1 && c.ev == op.ev && c.gen == op.gen) src = o;
      if (!src) { fprintf(stderr, "emu: wait on an event record that no stream holds\n"); abort(); }
      drain(*src, op.ev, op.gen);
    }
  }
  q.draining = false;
}
}  // namespace
void submit(void* stream, std::function<void()> op, const char* what) {
  {
    std::lock_guard<std::mutex> lk(q_mu);
    if (g_lazy) {
      queue_of(stream).ops.push_back(Op{0, std::move(op), nullptr, 0, what});
      return;
    }
  }
  op();
}
Event* event_new() { return new Event(); }
void event_delete(Event* e) { delete e; }
void event_record(Event* e, void* stream) {
  std::lock_guard<std::mutex> lk(q_mu);
  if (!g_lazy || !e) return;
  queue_of(stream).ops.push_back(Op{1, {}, e, ++e->submitted});
}
void stream_wait(void* stream, Event* e) {
  std::lock_guard<std::mutex> lk(q_mu);
  if (!g_lazy || !e || e->submitted == 0) return;                // never recorded: no dependency (as on the device)
  queue_of(stream).ops.push_back(Op{2, {}, e, e->submitted});
}
}  // namespace emu

// hipStreamSynchronize: in immediate mode everything has run; in lazy mode the stream's queue is drained (waits pull in what they need)
void emu_stream_synchronize(void* st) {
  emu::Queue* q = nullptr;
  {
    std::lock_guard<std::mutex> lk(emu::q_mu);
    if (!emu::g_lazy) return;
    q = &emu::queue_of(st);
  }
  const bool was = emu::g_lazy;
  emu::g_lazy = false;
  emu::drain(*q, nullptr, 0);
  emu::g_lazy = was;
}

// test hooks (exported next to the product's C ABI in libdynaboa_emu.so)
extern "C" void emu_lazy(int on) {
  std::lock_guard<std::mutex> lk(emu::q_mu);
  emu::g_lazy = on != 0;
}
// order 0: streams in order of first use, each to completion (later ones only run ahead where a wait demands it); 1: the reverse.
// Returns the number of streams that held work.
extern "C" int emu_flush(int order) {
  std::vector<emu::Queue*> qs;
  {
    std::lock_guard<std::mutex> lk(emu::q_mu);
    qs = emu::g_queues;
  }
  int n = 0;
  for (emu::Queue* q : qs) n += q->ops.empty() ? 0 : 1;
  if (order) std::reverse(qs.begin(), qs.end());
  const bool was = emu::g_lazy;
  emu::g_lazy = false;                 // operations issued by running work (none today) would run at once
  for (emu::Queue* q : qs) emu::drain(*q, nullptr, 0);
  emu::g_lazy = was;
  return n;
}
