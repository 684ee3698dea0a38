// Shared internals of libdynaboa_hip.so (gfx950 / CDNA4 only).
// Public C ABI lives in include/dynaboa_hip.h; this header is private to csrc/.
#pragma once
// gfx950 only.  The in-kernel hand-offs (one-pass GroupNorm backward, GroupNorm tangents, the convolutions' split-K fold) order their
// device-scope write-through stores with a gfx9-encoded s_waitcnt vmcnt(0) and rely on sc1 cache policies: on a target where stores are
// counted separately (vscnt) or the encodings differ they would silently hand over incomplete data - refuse to build for one.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libdynaboa_hip is written for gfx950 (MI355X) only"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#define DYB_OK 0
#define DYB_ERR_ARG (-1)      // bad pointer / dimension
#define DYB_ERR_LAUNCH (-2)   // hipGetLastError() after a launch
#define DYB_ERR_UNSUPPORTED (-3)
#define DYB_ERR_WORKSPACE (-4)

#define DYB_GN_GROUPS 4       // reference model/hmr.py:18  GroupNorm(32 // 8, planes)
#define DYB_GN_EPS 1e-5f

#define DYB_CHECK_LAUNCH()                                  \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return DYB_ERR_LAUNCH; \
  } while (0)

#define DYB_REQUIRE(cond, code) \
  do {                          \
    if (!(cond)) return (code); \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));          // A / B fragment of v_mfma_f32_32x32x16_bf16
// fp32 -> bf16 bits, round to nearest even (NaN payloads are not preserved; the path never produces them)
__device__ __forceinline__ unsigned dyb_f2bf(float f) {
  const unsigned u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int dyb_ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
__device__ __forceinline__ int dyb_ilog2_dev(int v) { return 31 - __clz(v); }
static inline bool dyb_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int dyb_cdiv(int a, int b) { return (a + b - 1) / b; }

// 64-lane butterfly sum (wave = 64 on gfx950).
__device__ __forceinline__ float dyb_wave_sum(float v) {
  v += __shfl_xor(v, 32);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

// ---- sequence replicas in the grid ------------------------------------------------------------------------------
// The per-frame chain of ONE sequence is a few thousand small dependent launches that leave most of the 256 CUs
// idle, and independent sequences (own weights, own Adam state: SURVEY 8e) are the only parallel axis the algorithm
// has.  So a launch may cover n such replicas at once: the grid's z extent is multiplied by n, workgroup z / own_z
// is the replica index, and every pointer argument that lies inside replica 0's copy of a registered arena is moved
// by replica * stride (pointers into shared tables - SMPL, GMM prior, regressors - match no arena and stay).
// The host side carries the current replica set in a thread-local (DybRepScope), so the launch wrappers need no
// extra parameters; without a scope n = 1 and every kernel behaves exactly as before.
// A launch may also cover a SUBSET of the replicas (the "active set": sequences still inside the dynamic-BOA loop, sequences
// that have frames left): `map[i]` is the physical replica - the index its arenas are addressed with - of the launch's i-th
// replica.  Identity outside such a scope.
#define DYB_MAX_ARENAS 8       // (every arena is three 64-bit kernel-argument words each kernel compares its pointers against: 12 of them pushed
                               // five register-heavy kernels past the scalar-register file and their DybRep argument into scratch)
#define DYB_MAX_REPLICAS 64
struct DybRep {
  int n;                                        // replicas covered by the launch
  int narenas;
  int ident;                                    // the map is the identity: kernels skip the decode
  int pad_;
  const char* lo[DYB_MAX_ARENAS];               // replica 0's range of each arena
  unsigned long long span[DYB_MAX_ARENAS];      // bytes
  unsigned long long stride[DYB_MAX_ARENAS];    // bytes between consecutive replicas
  // launch replica -> physical replica, ten 6-bit entries per word.  (Not a byte array: a dynamically indexed member of a by-value
  // kernel argument made the compiler copy the whole struct to scratch in three register-heavy kernels - GroupNorm apply ran 2x
  // slower; the decode below only ever indexes with constants.)
  unsigned long long mapw[7];
};
static inline int dyb_rep_phys(const DybRep& R, int i) { return (int)((R.mapw[i / 10] >> (6 * (i % 10))) & 63ull); }
static inline void dyb_rep_identity(DybRep& R) {
  for (int w = 0; w < 7; ++w) R.mapw[w] = 0;
  for (int i = 0; i < DYB_MAX_REPLICAS; ++i) R.mapw[i / 10] |= (unsigned long long)i << (6 * (i % 10));
  R.ident = 1;
}
static inline void dyb_rep_set_map(DybRep& R, const int* idx, int n) {
  R.n = n;
  R.ident = 1;
  for (int i = 0; i < n; ++i) {
    R.mapw[i / 10] = (R.mapw[i / 10] & ~(63ull << (6 * (i % 10)))) | ((unsigned long long)(idx[i] & 63) << (6 * (i % 10)));
    if (idx[i] != i) R.ident = 0;
  }
}
__device__ __forceinline__ int dyb_rep_phys_dev(const DybRep& R, int l) {
  const int q = (l * 205) >> 11, r = l - 10 * q;                  // l / 10 for l < 64
  unsigned long long w = R.mapw[0];
  w = q == 1 ? R.mapw[1] : w; w = q == 2 ? R.mapw[2] : w; w = q == 3 ? R.mapw[3] : w;
  w = q == 4 ? R.mapw[4] : w; w = q == 5 ? R.mapw[5] : w; w = q == 6 ? R.mapw[6] : w;
  return (int)((w >> (6 * r)) & 63ull);
}
template <class T>
__device__ __forceinline__ T* dyb_rb(T* p, const DybRep& R, int rep) {
  const char* c = reinterpret_cast<const char*>(p);
#pragma unroll
  for (int a = 0; a < DYB_MAX_ARENAS; ++a)
    if (a < R.narenas && (unsigned long long)(c - R.lo[a]) < R.span[a])
      return reinterpret_cast<T*>(const_cast<char*>(c) + (unsigned long long)rep * R.stride[a]);
  return p;
}
// buffer resource over [p, p + bytes) (raw, stride 0; gfx9 data-format word) and a 16-byte load through it
// component-wise (a select between two float4 objects makes hipcc spill both to scratch and pick by address)
__device__ __forceinline__ float4 tp_mask4(float4 v, bool ok) {
  v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
  return v;
}
#define TP_OOB 0x80000000u
typedef unsigned tp_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tp_rsrc(const float* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), (short)0, (int)(bytes < 0x7fffffffu ? bytes : 0x7fffffffu), 0x00020000);
}
__device__ __forceinline__ float4 tp_buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const tp_u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  const unsigned a = x.x, b = x.y, c = x.z, d = x.w;      // (bit_cast straight from a vector element reads element 0 on host clang)
  return make_float4(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b), __builtin_bit_cast(float, c), __builtin_bit_cast(float, d));
}
// 16-byte store through a buffer resource with cache policy AUX (0 plain, 16 = sc1: write-through at device scope, 2 = nt, 17 = sc0 sc1);
// a per-lane offset >= num_records (TP_OOB) is dropped by the hardware
template <int AUX>
__device__ __forceinline__ void tp_buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, float4 v) {
  tp_u4 x;
  x.x = __builtin_bit_cast(unsigned, v.x); x.y = __builtin_bit_cast(unsigned, v.y);
  x.z = __builtin_bit_cast(unsigned, v.z); x.w = __builtin_bit_cast(unsigned, v.w);
  __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)voff, 0, AUX);
}
// the same load past the vector L1 (sc1): data another workgroup of this launch stored write-through
__device__ __forceinline__ float4 tp_buf_load4_dev(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  const tp_u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 16);
  const unsigned a = x.x, b = x.y, c = x.z, d = x.w;
  return make_float4(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b), __builtin_bit_cast(float, c), __builtin_bit_cast(float, d));
}
// replica index (dyb_lrep: within the launch; dyb_rep: physical, what arenas and per-replica argument tables are indexed with)
// and the kernel's own z coordinates; then DYB_RB(ptr)... for every pointer the kernel dereferences
#define DYB_REP_PROLOGUE(R)                                   \
  const unsigned dyb_gz = gridDim.z / (unsigned)(R).n;        \
  const int dyb_lrep = (int)(blockIdx.z / dyb_gz);            \
  const int dyb_rep = (R).ident ? dyb_lrep : dyb_rep_phys_dev((R), dyb_lrep); \
  const unsigned dyb_bz = blockIdx.z - (unsigned)dyb_lrep * dyb_gz; \
  (void)dyb_bz
#define DYB_RB(R, p) \
  do {               \
    if (dyb_rep) (p) = dyb_rb((p), (R), dyb_rep); \
  } while (0)
const DybRep& dyb_rep_current();                 // thread-local; {n = 1} outside a scope (igemm_conv.hip)
struct DybRepScope {
  DybRep saved;
  explicit DybRepScope(const DybRep& r);
  ~DybRepScope();
};

// counter region of the in-kernel split-K fold for the calling host thread's conv launches (igemm_conv.hip): `nwords` words, zero
// before the first launch that uses them (every launch leaves them zero); one word per (replica slot of the launch, tile), shared by
// the replicas of a launch (never rebased).  One region per stream that convolutions are issued on concurrently.
struct DybConvSync {
  unsigned* ctr;
  int nwords;
};
// torch.optim.Adam, single tensor, no amsgrad / weight decay (reference base_adaptor.py:126, dynaboa_benchmark.py:149-151):
// exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2); denom = sqrt(v)/bc2_sqrt + eps; p.addcdiv_(m, denom, -step_size).
// ONE definition for the streaming kernels (optim.hip) and the weight-gradient epilogue (igemm_tp.inc "fuse_adam"): the two must round alike.
// Every rounding is spelled out (contraction off, the three multiply-adds as explicit fmaf): left to the compiler's contraction pass the
// same source rounded differently in the two call sites (1 ulp on a few elements per million: 4e-8 of |theta| after three frames, s7).
__device__ __forceinline__ void dyb_adam_one(float& p, float g, float& m, float& v, float b1, float b2, float step_size, float bc2_sqrt,
                                             float eps) {
#pragma clang fp contract(off)
  m = fmaf(1.f - b1, g - m, m);
  const float g1 = (1.f - b2) * g;
  v = fmaf(g1, g, v * b2);
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = fmaf(-step_size, m / denom, p);
}
// the MAML fast-weight step p' = p - lr * g as ONE fused multiply-add everywhere (streaming kernels, conv and linear weight-gradient epilogues)
__device__ __forceinline__ float dyb_fast_one(float p, float g, float lr) { return fmaf(-lr, g, p); }
// update_teacher (reference base_adaptor.py:193-201): t = alpha * t + (1 - alpha) * p, one rounding of alpha * t and one fused multiply-add,
// the same in the EMA launch and in the Adam kernel that carries the EMA of the element it has just updated
__device__ __forceinline__ float dyb_ema_one(float t, float p, float alpha, float om) { return fmaf(om, p, t * alpha); }
// weight-update scope (igemm_conv.hip "fuse_fast" / "fuse_adam"): see DybWgradUpdateScope's definition there
struct DybSpan {
  size_t off, n;         // floats, relative to the gradient arena
};
struct DybWgradUpdate {
  const float* grads;    // gradient arena the weight gradients would be written into (replica 0's copy)
  size_t bytes;
  const float* p_cur;    // current weights (same layout)
  float* p_next;         // where p_cur - lr * g goes
  float lr;
  std::vector<DybSpan>* spans;   // fused spans are appended here (may be NULL)
  // Adam instead of the fast-weight step ("fuse_adam"): p_next == p_cur = theta, updated in place from the accumulators together with
  // the moments; adam_sc = two floats (step_size, bc2_sqrt) inside a PER-REPLICA arena (each replica's own bias corrections)
  float* adam_m = nullptr;
  float* adam_v = nullptr;
  const float* adam_sc = nullptr;
  float b1 = 0.f, b2 = 0.f, eps = 0.f;
};
// segment list of a streaming fast-weight launch (optim.hip dyb_fastweight_update_segs): float4 units relative to the arena base
#define DYB_FW_MAX_SEGS 64
struct DybFwSegs {
  unsigned n;
  unsigned blk[DYB_FW_MAX_SEGS + 1];       // filled by the launcher
  unsigned start4[DYB_FW_MAX_SEGS], count4[DYB_FW_MAX_SEGS];
};
const DybWgradUpdate& dyb_wgrad_update_current();      // the calling thread's scope (grads == NULL: none)
struct DybWgradUpdateScope {
  DybWgradUpdate saved;
  explicit DybWgradUpdateScope(const DybWgradUpdate& u);
  ~DybWgradUpdateScope();
};
struct DybConvSyncScope {
  DybConvSync saved;
  DybConvSyncScope(unsigned* ctr, int nwords);
  ~DybConvSyncScope();
};
#define DYB_CONV_SYNC_WORDS 16384

// bf16 matrix-core mode of the calling host thread (igemm_conv.hip)
bool dyb_bf16_current();
struct DybBf16Scope {
  bool saved;
  explicit DybBf16Scope(bool on);
  ~DybBf16Scope();
};

// workgroup cap of the flat-arena streaming launches of the calling host thread (optim.hip)
struct DybStreamCapScope {
  int saved;
  explicit DybStreamCapScope(int cap);
  ~DybStreamCapScope();
};

// ---- cross-file internals (not part of the C ABI) ----------------------------------------
struct ConvDesc {
  int N, H, W, C, K, R, S, stride, pad;
};
// forward conv that may leave `*nslabs` (>1) un-reduced split-K slabs in `ws` for the GroupNorm
// statistics kernel to fold (igemm_conv.hip)
// operand pairs of the tangent passes (igemm_conv.hip): out = op(a1, b1) + op(a2, b2) (+ addend, data gradient only) as one launch;
// mode 0 / 1 / 2 = forward / data gradient / weight gradient.  Latency form only: ask dyb_conv_pair_supported first.
bool dyb_conv_pair_supported(int mode, const ConvDesc& d);
int dyb_conv_pair(int mode, const ConvDesc& d, const float* a1, const float* b1, const float* a2, const float* b2, float* out,
                  const float* addend, void* ws, size_t ws_bytes, hipStream_t st);
int dyb_conv_fwd_raw(const ConvDesc& d, const float* x, const float* w, float* y, void* ws, size_t ws_bytes,
                     int* nslabs, hipStream_t st);
// chunking of the GroupNorm-backward partial sums (norm_pool.hip): nch row chunks x ncolb column blocks
void dyb_gn_bwd_layout(int N, int HW, int C, int* nch, int* ncolb);
int dyb_gn_bwd_reduce_slabs(const float* dout, int nslabs, size_t slab_stride, const float* addend, const float* out,
                            const float* y, const float* stats, const float* gamma, const float* beta, float* dm,
                            float* part, int N, int HW, int C, int relu, hipStream_t st, hipEvent_t done);
int dyb_gn_fwd_chunks(int N, int HW);
// data gradient with the GroupNorm backward in its loader that may leave `*nslabs` (>1) un-reduced
// split-K slabs in `ws` (addend NOT applied then) for the next GroupNorm-backward reduce to fold
struct GnBwdSrc {
  const float *dm, *y, *stats, *part, *gamma;
  int nch, ncolb;          // partial-block layout; 0 = the one dyb_groupnorm_bwd_reduce uses
};
bool dyb_conv_dgrad_k4_ok(const ConvDesc& d);
int dyb_conv_dgrad_k4(const ConvDesc& d, const GnBwdSrc& src, const float* w, const float* addend, const float* y_p,
                      const float* out_p, const float* stats_p, const float* gamma_p, const float* beta_p, float* dm_p,
                      float* part_p, int* nch, int* ncolb, hipStream_t st, hipEvent_t done);
int dyb_conv_dgrad_gn_raw(const ConvDesc& d, const GnBwdSrc& src, const float* w, float* dx, const float* addend, void* ws,
                          size_t ws_bytes, int* nslabs, hipStream_t st);
int dyb_splitk_fold(const float* slabs, int nslabs, size_t n, const float* addend, float* out, hipStream_t st);
// throughput schedule (several sequence replicas per launch): dy materialised once per layer, plain gradient convolutions
// throughput schedule: "rep_split" switch on and >= "tp_min" (8) replicas in the current launch scope, or - switch
// "tp_batch_min" > 0 (off by default: unmeasured) - a batch of at least that many images
bool dyb_throughput_mode(int batch = 1);
int dyb_gn_replica_share(int N);                // > 0: workgroups per image a GroupNorm launch may use (replica-aware chunking)
// one-pass GroupNorm backward of the throughput schedule (norm_pool.hip): chunks per slab (0: shape does not qualify), the floats
// of partial sums the k-chunk form needs, the launch, a replica-aware zero fill (arrival counters), the policy switches
int dyb_gn_onepass_chunks(int N, int HW, int C, int cap);
size_t dyb_gn_onepass_part_floats(int k, int C);
int dyb_gn_bwd_onepass(const float* din, int nslabs, size_t slab_stride, const float* addend, const float* out, const float* y,
                       const float* stats, const float* gamma, const float* beta, float* dm, float* dy, float* dgamma, float* dbeta,
                       int HW, int C, int relu, int k, float* part, unsigned* ctr, hipStream_t st);
int dyb_zero_words(unsigned* p, int n, hipStream_t st);
int dyb_tp_gn_onepass();
int dyb_tp_gn_cap();
int dyb_tp_gn_threads();
int dyb_tp_gn_poll();
int dyb_tp_gn_wt();
int dyb_gn_bwd_apply_dy(const float* dm, const float* y, const float* stats, const float* part, int nch, int ncolb,
                        const float* gamma, float* dy, float* dgamma, float* dbeta, int N, int HW, int C, hipStream_t st);
int dyb_conv_dgrad_plain_raw(const ConvDesc& d, const float* dy, const float* w, float* dx, const float* addend, void* ws,
                             size_t ws_bytes, int* nslabs, hipStream_t st);
// dw = x^T dy with x plain (x != NULL) or relu(gn(y_prev)) formed in the loader from the saved statistics
int dyb_conv_wgrad_plain(const ConvDesc& d, const float* x, const float* y_prev, const float* stats_prev, const float* gamma_prev,
                         const float* beta_prev, const float* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st);
int dyb_avgpool_fwd_tail(const float* x, float* const* dsts, int ndst, int ld, int N, int HW, int C, const float* tail,
                         int tail_ld, int tail_cols, int tail_dst_col, hipStream_t st);

// ---- engine internals used by the native frame stepper (adapt_step.hip) ---------------------------------------------
// cross-stream events of ONE backward chain (per conv layer + the final join)
struct DybEvents {
  std::vector<hipEvent_t> dy;
  hipEvent_t join = nullptr;
};
DybEvents* dyb_hmr_events_create(const void* plan);
void dyb_hmr_events_destroy(DybEvents* e);
// Weight-ready gates of a forward: the parameter arena is laid out in forward order, so a caller that updates the weights by arena
// ranges on another stream (the frame stepper: fast-weight steps / Adam beside the next forward's first layers) hands over one event
// per range - ev[0]: layer3's weights onward, ev[1]: layer4 + regressor - and the forward waits for each right before its first
// reader.  dyb_hmr_param_groups: the float offsets where those two ranges begin.
struct DybFwdGates {
  hipEvent_t ev[2];
  // optional: work the caller wants issued when the forward reaches layer3 (the last range's update: it then runs beside layer3's
  // compute-bound convolutions instead of beside the stem / layer1 / layer2 GroupNorm applies) - called once, after `mid` has been
  // recorded on the forward's stream; it must make ev[1] happen
  hipEvent_t mid = nullptr;
  int (*late)(void* user) = nullptr;
  void* user = nullptr;
};
void dyb_hmr_param_groups(const void* plan, size_t bounds[2]);
int dyb_hmr_forward_plain(void* plan, const float* params, const float* image, const float* init_state, int n_iter, float* acts,
                          void* ws, size_t ws_bytes, hipStream_t st, const DybFwdGates* gates = nullptr);
int dyb_hmr_backward_ev(void* plan, const float* params, const float* acts, const float* d_rotmat, const float* d_state,
                        int n_iter, float* grads, void* ws, size_t ws_bytes, hipStream_t st, hipStream_t aux, const DybEvents* ev);
