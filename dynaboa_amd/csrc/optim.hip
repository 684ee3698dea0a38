// Flat-arena parameter updates: the MAML fast-weight step, Adam, the mean-teacher EMA, and the
// feature cosine that gates the dynamic-BOA loop.  All 169 HMR tensors live in ONE contiguous
// fp32 arena (27.0 M floats, 108 MB), so each of these is a single streaming launch instead of
// 169 small ones.  Pure HBM-bound: 16 B per lane, grid-stride, 2048 workgroups.
//
//   fast weights : p' = p - lr * g                 learn2learn MAML.adapt (call
//                                                  sites reference dynaboa_benchmark.py:136,140)
//   Adam         : torch.optim.Adam single-tensor formula, no amsgrad / weight decay
//                                                  (reference base_adaptor.py:126, dynaboa_benchmark.py:149-151)
//   EMA          : t = alpha*t + (1-alpha)*p       reference base_adaptor.py:193-201
//   cosine       : F.cosine_similarity(a.flatten(), b.flatten(), dim=0, eps)   base_adaptor.py:211-219
#include "dyb_common.h"

// workgroups of a streaming launch: 2048, or the calling thread's cap (DybStreamCapScope: a pass that runs BESIDE convolutions on
// another queue takes fewer - it only has to finish before its consumer, and 2048 light workgroups would fill every workgroup
// slot of the chip in front of the convolutions' own)
static thread_local int t_stream_cap = 2048;
DybStreamCapScope::DybStreamCapScope(int cap) : saved(t_stream_cap) { t_stream_cap = cap > 0 ? cap : 2048; }
DybStreamCapScope::~DybStreamCapScope() { t_stream_cap = saved; }
static int stream_blocks(size_t n4) {
  size_t b = (n4 + 255) / 256;
  if (b > (size_t)t_stream_cap) b = (size_t)t_stream_cap;
  if (b < 1) b = 1;
  return (int)b;
}

// g2 / g3 (may be NULL): further gradient arenas of the same level - the history-frame pass and the exemplar pass of the full term set
// (base_adaptor.py:380-398) - summed here, (g + g2) + g3, instead of in accumulation passes of their own (12 B per parameter each)
__global__ __launch_bounds__(256) void fastweight_kernel(const float4* __restrict__ p, const float4* __restrict__ g,
                                                         const float4* __restrict__ g2, const float4* __restrict__ g3,
                                                         float4* __restrict__ out, float lr, size_t n4, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, p); DYB_RB(Rp, g); DYB_RB(Rp, g2); DYB_RB(Rp, g3); DYB_RB(Rp, out);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 a = p[i], b = g[i];
    if (g2) { const float4 c = g2[i]; b.x = c.x + b.x; b.y = c.y + b.y; b.z = c.z + b.z; b.w = c.w + b.w; }
    if (g3) { const float4 c = g3[i]; b.x = c.x + b.x; b.y = c.y + b.y; b.z = c.z + b.z; b.w = c.w + b.w; }
    a.x = dyb_fast_one(a.x, b.x, lr); a.y = dyb_fast_one(a.y, b.y, lr); a.z = dyb_fast_one(a.z, b.z, lr); a.w = dyb_fast_one(a.w, b.w, lr);
    out[i] = a;
  }
}
int dyb_fastweight_update3(const float* p, const float* g, const float* g2, const float* g3, float* out, float lr, size_t n, hipStream_t st) {
  DYB_REQUIRE(p && g && out && n % 4 == 0, DYB_ERR_ARG);
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(fastweight_kernel, dim3(stream_blocks(n / 4), 1, Rp.n), dim3(256), 0, st, (const float4*)p, (const float4*)g,
                     (const float4*)g2, (const float4*)g3, (float4*)out, lr, n / 4, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// The fast-weight step over a LIST of arena segments in one launch (round 6, "fuse_fast"): the spans whose weight gradient wrote
// p - lr * g itself (igemm_tp.inc epilogue, DybWgradUpdateScope) are left out; what remains - GroupNorm affines, the regressor, the
// convolutions that ran split - is up to 64 segments between them.  Workgroup b belongs to segment s with blk[s] <= b < blk[s + 1].
__global__ __launch_bounds__(256) void fastweight_segs_kernel(const float4* __restrict__ p, const float4* __restrict__ g,
                                                              float4* __restrict__ out, float lr, DybFwSegs t, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, p); DYB_RB(Rp, g); DYB_RB(Rp, out);
  const unsigned b = blockIdx.x;
  unsigned s = 0;
  while (s + 1 < t.n && b >= t.blk[s + 1]) ++s;              // (workgroup-uniform)
  const unsigned nb = t.blk[s + 1] - t.blk[s], lb = b - t.blk[s];
  const size_t base = t.start4[s];
  for (size_t i = (size_t)lb * 256 + threadIdx.x; i < t.count4[s]; i += (size_t)nb * 256) {
    float4 a = p[base + i];
    const float4 c = g[base + i];
    a.x = dyb_fast_one(a.x, c.x, lr); a.y = dyb_fast_one(a.y, c.y, lr); a.z = dyb_fast_one(a.z, c.z, lr); a.w = dyb_fast_one(a.w, c.w, lr);
    out[base + i] = a;
  }
}
static void fw_segs_blocks(DybFwSegs& t) {
  size_t total4 = 0;
  for (unsigned i = 0; i < t.n; ++i) total4 += t.count4[i];
  const size_t want = (total4 + 1023) / 1024;
  const double scale = want > (size_t)t_stream_cap ? (double)t_stream_cap / (double)want : 1.0;
  unsigned acc = 0;
  for (unsigned i = 0; i < t.n; ++i) {
    t.blk[i] = acc;
    unsigned nb = (unsigned)(((size_t)t.count4[i] + 1023) / 1024 * scale);
    acc += nb < 1 ? 1 : nb;
  }
  t.blk[t.n] = acc;
}
int dyb_fastweight_update_segs(const float* p, const float* g, float* out, float lr, const DybFwSegs& segs, hipStream_t st) {
  DYB_REQUIRE(p && g && out && segs.n >= 1 && segs.n <= DYB_FW_MAX_SEGS, DYB_ERR_ARG);
  DybFwSegs t = segs;
  // workgroups per segment: one per 1024 float4 (16 KB), at least one, the launch as a whole within the streaming cap
  fw_segs_blocks(t);
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(fastweight_segs_kernel, dim3(t.blk[t.n], 1, Rp.n), dim3(256), 0, st, (const float4*)p, (const float4*)g, (float4*)out, lr, t, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
extern "C" int dyb_fastweight_update(const float* p, const float* g, float* out, float lr, size_t n, hipStream_t st) {
  return dyb_fastweight_update3(p, g, nullptr, nullptr, out, lr, n, st);
}

#define adam_one dyb_adam_one      // (dyb_common.h: shared with the weight-gradient epilogue)
// bias corrections per PHYSICAL replica: sequence replicas stepped in lockstep may have taken different numbers of Adam steps
// (the dynamic-BOA loop repeats the outer step for some of them only; sequences of different lengths)
struct AdamRepScal {
  float step_size[DYB_MAX_REPLICAS], bc2_sqrt[DYB_MAX_REPLICAS];
};
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, const float4* __restrict__ g, const float4* __restrict__ g2,
                                                   const float4* __restrict__ g3, float4* __restrict__ m,
                                                   float4* __restrict__ v, float b1, float b2, AdamRepScal sc, float eps, size_t n4,
                                                   float4* __restrict__ teacher, float alpha, DybRep Rp) {
  // teacher != NULL (round 6): the mean teacher's EMA of the element just updated rides in the same pass (update_teacher follows
  // optimizer.step() in the reference, dynaboa_benchmark.py:149-153): theta is not streamed a second time
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, p); DYB_RB(Rp, g); DYB_RB(Rp, g2); DYB_RB(Rp, g3); DYB_RB(Rp, m); DYB_RB(Rp, v); DYB_RB(Rp, teacher);
  const float om = 1.f - alpha;
  const float step_size = sc.step_size[dyb_rep], bc2_sqrt = sc.bc2_sqrt[dyb_rep];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
    if (g2) { const float4 c = g2[i]; gg.x = c.x + gg.x; gg.y = c.y + gg.y; gg.z = c.z + gg.z; gg.w = c.w + gg.w; }   // (see fastweight_kernel)
    if (g3) { const float4 c = g3[i]; gg.x = c.x + gg.x; gg.y = c.y + gg.y; gg.z = c.z + gg.z; gg.w = c.w + gg.w; }
    adam_one(pp.x, gg.x, mm.x, vv.x, b1, b2, step_size, bc2_sqrt, eps);
    adam_one(pp.y, gg.y, mm.y, vv.y, b1, b2, step_size, bc2_sqrt, eps);
    adam_one(pp.z, gg.z, mm.z, vv.z, b1, b2, step_size, bc2_sqrt, eps);
    adam_one(pp.w, gg.w, mm.w, vv.w, b1, b2, step_size, bc2_sqrt, eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (teacher) {
      float4 t = teacher[i];
      t.x = dyb_ema_one(t.x, pp.x, alpha, om); t.y = dyb_ema_one(t.y, pp.y, alpha, om);
      t.z = dyb_ema_one(t.z, pp.z, alpha, om); t.w = dyb_ema_one(t.w, pp.w, alpha, om);
      teacher[i] = t;
    }
  }
}
// Adam over a LIST of arena segments (see fastweight_segs_kernel): the spans whose weight gradient applied Adam itself ("fuse_adam") left out
__global__ __launch_bounds__(256) void adam_segs_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                        float4* __restrict__ v, float b1, float b2, AdamRepScal sc, float eps, DybFwSegs t,
                                                        DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, p); DYB_RB(Rp, g); DYB_RB(Rp, m); DYB_RB(Rp, v);
  const float step_size = sc.step_size[dyb_rep], bc2_sqrt = sc.bc2_sqrt[dyb_rep];
  const unsigned b = blockIdx.x;
  unsigned s = 0;
  while (s + 1 < t.n && b >= t.blk[s + 1]) ++s;
  const unsigned nb = t.blk[s + 1] - t.blk[s], lb = b - t.blk[s];
  const size_t base = t.start4[s];
  for (size_t i = (size_t)lb * 256 + threadIdx.x; i < t.count4[s]; i += (size_t)nb * 256) {
    float4 pp = p[base + i], gg = g[base + i], mm = m[base + i], vv = v[base + i];
    adam_one(pp.x, gg.x, mm.x, vv.x, b1, b2, step_size, bc2_sqrt, eps);
    adam_one(pp.y, gg.y, mm.y, vv.y, b1, b2, step_size, bc2_sqrt, eps);
    adam_one(pp.z, gg.z, mm.z, vv.z, b1, b2, step_size, bc2_sqrt, eps);
    adam_one(pp.w, gg.w, mm.w, vv.w, b1, b2, step_size, bc2_sqrt, eps);
    p[base + i] = pp; m[base + i] = mm; v[base + i] = vv;
  }
}
int dyb_adam_step_segs(float* p, const float* g, float* m, float* v, float beta1, float beta2, const float* step_size, const float* bc2_sqrt,
                       float eps, const DybFwSegs& segs, hipStream_t st) {
  DYB_REQUIRE(p && g && m && v && step_size && bc2_sqrt && segs.n >= 1 && segs.n <= DYB_FW_MAX_SEGS, DYB_ERR_ARG);
  DybFwSegs t = segs;
  fw_segs_blocks(t);
  const DybRep& Rp = dyb_rep_current();
  AdamRepScal sc;
  for (int r = 0; r < DYB_MAX_REPLICAS; ++r) { sc.step_size[r] = step_size[r]; sc.bc2_sqrt[r] = bc2_sqrt[r]; }
  hipLaunchKernelGGL(adam_segs_kernel, dim3(t.blk[t.n], 1, Rp.n), dim3(256), 0, st, (float4*)p, (const float4*)g, (float4*)m, (float4*)v, beta1,
                     beta2, sc, eps, t, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// each replica's (step_size, bc2_sqrt) into dst[0..1] of ITS copy of a per-replica arena: what a weight-gradient epilogue that applies Adam reads
__global__ void adam_scalars_kernel(float* __restrict__ dst, AdamRepScal sc, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, dst);
  if (threadIdx.x == 0) { dst[0] = sc.step_size[dyb_rep]; dst[1] = sc.bc2_sqrt[dyb_rep]; }
}
int dyb_adam_write_scalars(float* dst, const float* step_size, const float* bc2_sqrt, hipStream_t st) {
  DYB_REQUIRE(dst && step_size && bc2_sqrt, DYB_ERR_ARG);
  const DybRep& Rp = dyb_rep_current();
  AdamRepScal sc;
  for (int r = 0; r < DYB_MAX_REPLICAS; ++r) { sc.step_size[r] = step_size[r]; sc.bc2_sqrt[r] = bc2_sqrt[r]; }
  hipLaunchKernelGGL(adam_scalars_kernel, dim3(1, 1, Rp.n), dim3(64), 0, st, dst, sc, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// Adam on a gradient that is still in two pieces, g - alpha * h: the last accumulation of the second-order outer gradient
// (v_0 = v_1 - lr * H v_1, dynaboa_amd/maml.py) is formed here instead of in a pass of its own, so the outer step of the
// second-order path is ONE streaming launch (reads p, g, h, m, v / writes p, m, v).
__global__ __launch_bounds__(256) void adam_accum_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                         const float4* __restrict__ h, float alpha, float4* __restrict__ m,
                                                         float4* __restrict__ v, float b1, float b2, float step_size, float bc2_sqrt,
                                                         float eps, size_t n4, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, p); DYB_RB(Rp, g); DYB_RB(Rp, h); DYB_RB(Rp, m); DYB_RB(Rp, v);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 pp = p[i], gg = g[i], hh = h[i], mm = m[i], vv = v[i];
    gg.x = gg.x - alpha * hh.x; gg.y = gg.y - alpha * hh.y; gg.z = gg.z - alpha * hh.z; gg.w = gg.w - alpha * hh.w;
    adam_one(pp.x, gg.x, mm.x, vv.x, b1, b2, step_size, bc2_sqrt, eps);
    adam_one(pp.y, gg.y, mm.y, vv.y, b1, b2, step_size, bc2_sqrt, eps);
    adam_one(pp.z, gg.z, mm.z, vv.z, b1, b2, step_size, bc2_sqrt, eps);
    adam_one(pp.w, gg.w, mm.w, vv.w, b1, b2, step_size, bc2_sqrt, eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}
extern "C" int dyb_adam_step_accum(float* p, const float* g, const float* h, float alpha, float* m, float* v, float beta1,
                                   float beta2, float step_size, float bc2_sqrt, float eps, size_t n, hipStream_t st) {
  DYB_REQUIRE(p && g && h && m && v && n % 4 == 0, DYB_ERR_ARG);
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(adam_accum_kernel, dim3(stream_blocks(n / 4), 1, Rp.n), dim3(256), 0, st, (float4*)p, (const float4*)g,
                     (const float4*)h, alpha, (float4*)m, (float4*)v, beta1, beta2, step_size, bc2_sqrt, eps, n / 4, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// step_size = lr / (1 - b1^t), bc2_sqrt = sqrt(1 - b2^t): computed by the caller in double.
extern "C" int dyb_adam_step(float* p, const float* g, float* m, float* v, float beta1, float beta2, float step_size,
                             float bc2_sqrt, float eps, size_t n, hipStream_t st) {
  DYB_REQUIRE(p && g && m && v && n % 4 == 0, DYB_ERR_ARG);
  const DybRep& Rp = dyb_rep_current();
  AdamRepScal sc;
  for (int r = 0; r < DYB_MAX_REPLICAS; ++r) { sc.step_size[r] = step_size; sc.bc2_sqrt[r] = bc2_sqrt; }
  hipLaunchKernelGGL(adam_kernel, dim3(stream_blocks(n / 4), 1, Rp.n), dim3(256), 0, st, (float4*)p, (const float4*)g, (const float4*)nullptr,
                     (const float4*)nullptr, (float4*)m, (float4*)v, beta1, beta2, sc, eps, n / 4, (float4*)nullptr, 0.f, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// the same with the two bias-correction scalars given per physical replica (host arrays of DYB_MAX_REPLICAS = 64 floats; entries of
// replicas outside the current launch scope are ignored)
int dyb_adam_step_rep3_ema(float* p, const float* g, const float* g2, const float* g3, float* m, float* v, float beta1, float beta2,
                           const float* step_size, const float* bc2_sqrt, float eps, size_t n, float* teacher, float alpha, hipStream_t st) {
  DYB_REQUIRE(p && g && m && v && step_size && bc2_sqrt && n % 4 == 0, DYB_ERR_ARG);
  const DybRep& Rp = dyb_rep_current();
  AdamRepScal sc;
  for (int r = 0; r < DYB_MAX_REPLICAS; ++r) { sc.step_size[r] = step_size[r]; sc.bc2_sqrt[r] = bc2_sqrt[r]; }
  hipLaunchKernelGGL(adam_kernel, dim3(stream_blocks(n / 4), 1, Rp.n), dim3(256), 0, st, (float4*)p, (const float4*)g, (const float4*)g2,
                     (const float4*)g3, (float4*)m, (float4*)v, beta1, beta2, sc, eps, n / 4, (float4*)teacher, alpha, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
int dyb_adam_step_rep3(float* p, const float* g, const float* g2, const float* g3, float* m, float* v, float beta1, float beta2,
                       const float* step_size, const float* bc2_sqrt, float eps, size_t n, hipStream_t st) {
  return dyb_adam_step_rep3_ema(p, g, g2, g3, m, v, beta1, beta2, step_size, bc2_sqrt, eps, n, nullptr, 0.f, st);
}
int dyb_adam_step_rep(float* p, const float* g, float* m, float* v, float beta1, float beta2, const float* step_size,
                      const float* bc2_sqrt, float eps, size_t n, hipStream_t st) {
  return dyb_adam_step_rep3(p, g, nullptr, nullptr, m, v, beta1, beta2, step_size, bc2_sqrt, eps, n, st);
}

__global__ __launch_bounds__(256) void ema_kernel(float4* __restrict__ t, const float4* __restrict__ p, float alpha, size_t n4,
                                                  DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, t); DYB_RB(Rp, p);
  const float om = 1.f - alpha;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 a = t[i], b = p[i];
    a.x = dyb_ema_one(a.x, b.x, alpha, om); a.y = dyb_ema_one(a.y, b.y, alpha, om);
    a.z = dyb_ema_one(a.z, b.z, alpha, om); a.w = dyb_ema_one(a.w, b.w, alpha, om);
    t[i] = a;
  }
}
extern "C" int dyb_ema_update(float* teacher, const float* p, float alpha, size_t n, hipStream_t st) {
  DYB_REQUIRE(teacher && p && n % 4 == 0, DYB_ERR_ARG);
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(ema_kernel, dim3(stream_blocks(n / 4), 1, Rp.n), dim3(256), 0, st, (float4*)teacher, (const float4*)p, alpha,
                     n / 4, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// y = a*x + b*y  (gradient accumulation across several forward passes that share theta)
__global__ __launch_bounds__(256) void axpby_kernel(const float4* __restrict__ x, float4* __restrict__ y, float a, float b,
                                                    size_t n4, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, x); DYB_RB(Rp, y);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 u = x[i], w = y[i];
    w.x = a * u.x + b * w.x; w.y = a * u.y + b * w.y; w.z = a * u.z + b * w.z; w.w = a * u.w + b * w.w;
    y[i] = w;
  }
}
extern "C" int dyb_axpby(const float* x, float* y, float a, float b, size_t n, hipStream_t st) {
  DYB_REQUIRE(x && y && n % 4 == 0, DYB_ERR_ARG);
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(axpby_kernel, dim3(stream_blocks(n / 4), 1, Rp.n), dim3(256), 0, st, (const float4*)x, (float4*)y, a, b, n / 4, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// cosine over n floats with arbitrary row strides: element e lives at base + (e / cols) * ld + e % cols
// (covers both contiguous features and the NHWC activations, whose flattening order does not
// matter for a cosine).  One workgroup; n <= ~1 M in this model.  out[0] = cos.
__global__ __launch_bounds__(1024) void cosine_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                      float eps, float* __restrict__ out) {
  __shared__ float sm[16][3];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float ab = 0.f, aa = 0.f, bb = 0.f;
  for (size_t i = t; i < n; i += 1024) {
    float x = a[i], y = b[i];
    ab += x * y; aa += x * x; bb += y * y;
  }
  ab = dyb_wave_sum(ab); aa = dyb_wave_sum(aa); bb = dyb_wave_sum(bb);
  if (lane == 0) { sm[wave][0] = ab; sm[wave][1] = aa; sm[wave][2] = bb; }
  __syncthreads();
  if (t == 0) {
    float x = 0.f, y = 0.f, z = 0.f;
    for (int w = 0; w < 16; ++w) { x += sm[w][0]; y += sm[w][1]; z += sm[w][2]; }
    // torch: x.y / sqrt(clamp(|x|^2 * |y|^2, eps^2))
    float den = sqrtf(fmaxf(y * z, eps * eps));
    out[0] = x / den;
  }
}
extern "C" int dyb_cosine_sim(const float* a, const float* b, size_t n, float eps, float* out, hipStream_t st) {
  DYB_REQUIRE(a && b && out && n > 0, DYB_ERR_ARG);
  hipLaunchKernelGGL(cosine_kernel, dim3(1), dim3(1024), 0, st, a, b, n, eps, out);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// Diagnostic: n dependent launches of a one-workgroup no-op kernel issued from C++ on `stream`
// (measures the per-launch floor of a dependent kernel chain on the system; not used by the path).
__global__ void chain_probe_kernel(float* p) {
  if (threadIdx.x == 0) p[0] += 1.0f;
}
extern "C" int dyb_debug_launch_chain(float* scratch, int n, int blocks, hipStream_t st) {
  DYB_REQUIRE(scratch && n > 0 && blocks > 0, DYB_ERR_ARG);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(chain_probe_kernel, dim3(blocks), dim3(256), 0, st, scratch);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
