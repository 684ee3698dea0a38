// Tangent (forward-mode) kernels for the exact Hessian-vector product of the second-order path.
//
// Second-order MAML (reference base_adaptor.py:119 with first_order=False) needs H v = d/de grad L(theta + e v)
// per inner step.  Convolutions, linears, average pooling and the fixed-index max-pool are (bi)linear, so their tangents reuse
// the first-order kernels on tangent operands; what is left is GroupNorm(+ReLU, +residual) - forward tangent and the tangent
// of its backward - and the max-pool gather.  These are those kernels; hvp_engine.inc walks the network with them.
//
// A (image, group) slab of HW x C/4 values is cut into `chunks` row ranges, one workgroup each: a sums launch leaves per-chunk
// partial sums (double) in scratch, the apply launch adds them up in chunk order (the same order in every workgroup: results do not
// depend on the chunk count's scheduling) and writes its rows.  At one image per call a layer has 4 slabs - 4 workgroups on a
// 256-CU part without the cut (profiles/r03_so_kernel_stats_S1.csv: 50 / 70 us per launch, 34 % of the second-order frame).
//
// With a counter array (`sync`: one word per (image, group) slab, zeroed by the caller once per pass) sums and apply are ONE launch:
// a workgroup publishes its chunk's partial sums with device-scope stores, arrives on its slab's counter and waits there for the
// slab's other chunks before it applies (the hand-off of the one-pass GroupNorm backward, norm_pool.hip: no cache-wide fence; the
// grid is at most 128 x 4 workgroups, all resident, so the wait is finite; a poll that lasts 0.2 s raises the error word sync[-1]
// and goes on).  At one image per call the frame is bound by the host's launch rate: two launches less per layer and pass.
#include "dyb_common.h"

#define G DYB_GN_GROUPS

__device__ __forceinline__ void jvp_store_dev(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double jvp_load_dev(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void jvp_store_dev(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float jvp_load_dev(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// thread 0 of a workgroup: arrive on `ctr` (after the workgroup's device-scope stores have drained: callers put s_waitcnt vmcnt(0) and a
// barrier in front) and wait until `need` workgroups have
__device__ unsigned g_jvp_sync_errors = 0u;              // timed-out waits since load (norm_pool.hip dyb_sync_error_count)
int dyb_hvp_sync_errors(unsigned* host_out, hipStream_t st) {
  return hipMemcpyFromSymbolAsync(host_out, HIP_SYMBOL(g_jvp_sync_errors), sizeof(unsigned), 0, hipMemcpyDeviceToHost, st) == hipSuccess
             ? DYB_OK : DYB_ERR_LAUNCH;
}
__device__ __forceinline__ void jvp_arrive(unsigned* ctr) { atomicAdd(ctr, 1u); }
__device__ __forceinline__ void jvp_wait(unsigned* ctr, unsigned need, unsigned* err) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > 20000000LL) {            // 100 MHz: 0.2 s
      atomicAdd(err, 1u);
      atomicAdd(&g_jvp_sync_errors, 1u);
      break;
    }
  }
}

// block-wide sums of K doubles (256 threads); result in every thread
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* s_red) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double x = v[k];
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    if (lane == 0) s_red[wave * K + k] = x;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = (s_red[k] + s_red[K + k]) + (s_red[2 * K + k] + s_red[3 * K + k]);
  __syncthreads();
}

// chunk geometry of a slab: rows per chunk (>= 1) and the chunk count, from a target of ~1024 float4 per workgroup; images x chunks
// stays <= 128 so that the per-channel column sums stay a short loop
struct GnJvpGeom { int chunks, rows; };
static GnJvpGeom gn_jvp_geom(int N, int HW, int C) {
  const int cq = C / G / 4;
  int want = dyb_cdiv(HW * cq, 1024);
  int cap = 128 / N;
  if (cap > 64) cap = 64;
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  GnJvpGeom g;
  g.rows = dyb_cdiv(HW, want);
  g.chunks = dyb_cdiv(HW, g.rows);
  return g;
}
// floats of scratch the two entry points need: partial sums [N][G][chunks][4] (double) + per-chunk channel sums [N][chunks][2][C]
extern "C" size_t dyb_gn_jvp_scratch_floats(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0 || C % 16) return 0;
  const GnJvpGeom g = gn_jvp_geom(N, HW, C);
  return (size_t)N * G * g.chunks * 4 * 2 + (size_t)N * g.chunks * 2 * C;
}

// sum of K doubles over the chunks of slab (n, g), in chunk order
template <int K>
__device__ __forceinline__ void chunk_total(const double* __restrict__ part, int slab, int chunks, double (&v)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = 0.0;
  for (int ch = 0; ch < chunks; ++ch)
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += part[((size_t)slab * chunks + ch) * K + k];
}

template <int K>
__device__ __forceinline__ void chunk_total_dev(const double* part, int slab, int chunks, double (&v)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = 0.0;
  for (int ch = 0; ch < chunks; ++ch)
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += jvp_load_dev(part + ((size_t)slab * chunks + ch) * K + k);
}

// out = relu?(gamma * xhat + beta + res), tangent tout = mask * (tgamma * xhat + gamma * txhat + tbeta + tres) with
// xhat = (y - mu) r, txhat = r (ty - tmu - xhat a), tmu = mean(ty), a = mean(xhat ty) over the group; (tmu, a) saved for the
// backward tangent.  grid (chunks, G, N), block 256.  Sums: part[slab][chunk][2] = (sum ty, sum xhat ty) of the chunk's rows;
// with ty2 the tangent is ty + ty2 (the two halves of a convolution's tangent), summed into ty on the way.
struct GnJvpFwd {
  const float* y;
  float* ty;
  const float *ty2, *stats;
  double* part;
  const float *gamma, *beta, *tgamma, *tbeta, *res, *tres;
  float *out, *tout, *tstats;
  unsigned* sync;            // one-pass form: [N * G] arrival counters, sync[-1] the error word
  int HW, C, rows, relu;
};
// DEV: the partial sums leave with device-scope stores (one-pass form: read by other workgroups of the same launch)
template <bool DEV>
__device__ __forceinline__ void gn_jvp_fwd_sums(const GnJvpFwd& a, double* s_red) {
  const int ch = blockIdx.x, g = blockIdx.y, n = blockIdx.z, Cg = a.C / G, cq = Cg >> 2;
  const int slab = n * G + g;
  const float mu = a.stats[(size_t)slab * 2], r = a.stats[(size_t)slab * 2 + 1];
  const size_t base = (size_t)n * a.HW * a.C + (size_t)g * Cg;
  const int r0 = ch * a.rows, r1 = (r0 + a.rows < a.HW) ? r0 + a.rows : a.HW;
  double acc[2] = {0.0, 0.0};
  for (int i = r0 * cq + threadIdx.x; i < r1 * cq; i += 256) {
    const size_t off = base + (size_t)(i / cq) * a.C + (size_t)(i % cq) * 4;
    const float4 v = *reinterpret_cast<const float4*>(a.y + off);
    float4 t = *reinterpret_cast<const float4*>(a.ty + off);
    if (a.ty2) {
      const float4 u = *reinterpret_cast<const float4*>(a.ty2 + off);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      *reinterpret_cast<float4*>(a.ty + off) = t;
    }
    acc[0] += (double)t.x + (double)t.y + (double)t.z + (double)t.w;
    acc[1] += (double)((v.x - mu) * r) * t.x + (double)((v.y - mu) * r) * t.y + (double)((v.z - mu) * r) * t.z +
              (double)((v.w - mu) * r) * t.w;
  }
  block_sum<2>(acc, s_red);
  if (threadIdx.x == 0) {
    double* o = a.part + ((size_t)slab * gridDim.x + ch) * 2;
    if constexpr (DEV) { jvp_store_dev(o, acc[0]); jvp_store_dev(o + 1, acc[1]); }
    else { o[0] = acc[0]; o[1] = acc[1]; }
  }
}
// acc: the slab's totals (sum ty, sum xhat ty)
__device__ __forceinline__ void gn_jvp_fwd_apply(const GnJvpFwd& a, const double (&acc)[2]) {
  const int ch = blockIdx.x, g = blockIdx.y, n = blockIdx.z, Cg = a.C / G, cq = Cg >> 2;
  const int slab = n * G + g;
  const float mu = a.stats[(size_t)slab * 2], r = a.stats[(size_t)slab * 2 + 1];
  const size_t base = (size_t)n * a.HW * a.C + (size_t)g * Cg;
  const double cnt = (double)a.HW * (double)Cg;
  const float tmu = (float)(acc[0] / cnt), aa = (float)(acc[1] / cnt);
  if (threadIdx.x == 0 && ch == 0 && a.tstats) {
    a.tstats[(size_t)slab * 2] = tmu;
    a.tstats[(size_t)slab * 2 + 1] = aa;
  }
  const int r0 = ch * a.rows, r1 = (r0 + a.rows < a.HW) ? r0 + a.rows : a.HW;
  for (int i = r0 * cq + threadIdx.x; i < r1 * cq; i += 256) {
    const int c = g * Cg + (i % cq) * 4;
    const size_t off = base + (size_t)(i / cq) * a.C + (size_t)(i % cq) * 4;
    const float4 v = *reinterpret_cast<const float4*>(a.y + off), t = *reinterpret_cast<const float4*>(a.ty + off);
    const float4 ga = *reinterpret_cast<const float4*>(a.gamma + c), be = *reinterpret_cast<const float4*>(a.beta + c);
    const float4 tg = *reinterpret_cast<const float4*>(a.tgamma + c), tb = *reinterpret_cast<const float4*>(a.tbeta + c);
    float vv[4] = {v.x, v.y, v.z, v.w}, tt[4] = {t.x, t.y, t.z, t.w};
    const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
    const float tgg[4] = {tg.x, tg.y, tg.z, tg.w}, tbb[4] = {tb.x, tb.y, tb.z, tb.w};
    float rr[4] = {0.f, 0.f, 0.f, 0.f}, trr[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.res) {
      const float4 q = *reinterpret_cast<const float4*>(a.res + off);
      rr[0] = q.x; rr[1] = q.y; rr[2] = q.z; rr[3] = q.w;
    }
    if (a.tres) {
      const float4 q = *reinterpret_cast<const float4*>(a.tres + off);
      trr[0] = q.x; trr[1] = q.y; trr[2] = q.z; trr[3] = q.w;
    }
    float o[4], to[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (vv[k] - mu) * r;
      const float txh = r * (tt[k] - tmu - xh * aa);
      o[k] = fmaf(xh, gg[k], bb[k]) + rr[k];
      to[k] = tgg[k] * xh + gg[k] * txh + tbb[k] + trr[k];
      if (a.relu) {
        const bool on = o[k] > 0.f;
        o[k] = on ? o[k] : 0.f;
        to[k] = on ? to[k] : 0.f;
      }
    }
    if (a.out) *reinterpret_cast<float4*>(a.out + off) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(a.tout + off) = make_float4(to[0], to[1], to[2], to[3]);
  }
}
__global__ __launch_bounds__(256) void gn_jvp_fwd_sums_kernel(GnJvpFwd a) {
  __shared__ double s_red[4 * 2];
  gn_jvp_fwd_sums<false>(a, s_red);
}
__global__ __launch_bounds__(256) void gn_jvp_fwd_apply_kernel(GnJvpFwd a) {
  double acc[2];
  chunk_total<2>(a.part, blockIdx.z * G + blockIdx.y, gridDim.x, acc);
  gn_jvp_fwd_apply(a, acc);
}
// the workgroup has published (device-scope stores by thread 0): drain them, arrive on the slab's counter, wait for the slab's other
// chunks; thread 0 then adds the K partial sums up in chunk order and hands them to the workgroup through LDS
template <int K>
__device__ __forceinline__ void gn_jvp_meet(const double* part, unsigned* sync, int slab, double (&acc)[K], double* s_tot) {
  __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this thread's stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    jvp_arrive(sync + slab);
    jvp_wait(sync + slab, gridDim.x, sync - 1);
    chunk_total_dev<K>(part, slab, gridDim.x, acc);
#pragma unroll
    for (int k = 0; k < K; ++k) s_tot[k] = acc[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = s_tot[k];
}
__global__ __launch_bounds__(256) void gn_jvp_fwd_onepass_kernel(GnJvpFwd a) {
  __shared__ double s_red[4 * 2];
  __shared__ double s_tot[2];
  gn_jvp_fwd_sums<true>(a, s_red);
  double acc[2];
  gn_jvp_meet<2>(a.part, a.sync, blockIdx.z * G + blockIdx.y, acc, s_tot);
  gn_jvp_fwd_apply(a, acc);
}

// Backward of the same layer and its tangent.  dm = mask * dout, tdm = mask * tdout (mask = out_mask > 0 when relu);
// dxh = gamma dm, c1 = mean(dxh), c2 = mean(dxh xhat), dy = r (dxh - c1 - xhat c2);
// tdxh = tgamma dm + gamma tdm, tc1 = mean(tdxh), tc2 = mean(tdxh xhat + dxh txhat), tr = -r^2 a,
// tdy = tr (dxh - c1 - xhat c2) + r (tdxh - tc1 - txhat c2 - xhat tc2);
// per chunk and channel: tdgb[n][chunk][0][c] = sum_p tdm (tangent of dbeta), tdgb[n][chunk][1][c] = sum_p (tdm xhat + dm txhat)
// (of dgamma).  grid (chunks, G, N), block 256: sums (part[slab][chunk][4] + the channel sums), then apply (whose first workgroup
// per group also adds the channel sums up over images and chunks).
struct GnJvpBwdIn {
  const float *dout, *tdout, *out_mask, *y, *ty;
  float mu, r, tmu, a;
  int relu;
};
struct GnJvpBwd {
  const float *dout, *tdout, *out_mask, *y, *ty, *stats, *tstats, *gamma, *tgamma;
  double* part;
  float *dm, *tdm, *dy, *tdy, *tdgb, *tdbeta, *tdgamma;
  unsigned* sync;            // one-pass form: [N * G] arrival counters, sync[-1] the error word
  int HW, C, rows, relu;
};
__device__ __forceinline__ void gn_jvp_bwd_fetch(const GnJvpBwdIn& in, size_t off, float (&d)[4], float (&td)[4], float (&xh)[4],
                                                 float (&txh)[4]) {
  const float4 v = *reinterpret_cast<const float4*>(in.y + off), t = *reinterpret_cast<const float4*>(in.ty + off);
  const float4 d4 = *reinterpret_cast<const float4*>(in.dout + off), t4 = *reinterpret_cast<const float4*>(in.tdout + off);
  const float vv[4] = {v.x, v.y, v.z, v.w}, tt[4] = {t.x, t.y, t.z, t.w};
  d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w;
  td[0] = t4.x; td[1] = t4.y; td[2] = t4.z; td[3] = t4.w;
  if (in.relu) {
    const float4 m = *reinterpret_cast<const float4*>(in.out_mask + off);
    const float mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (!(mm[k] > 0.f)) { d[k] = 0.f; td[k] = 0.f; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    xh[k] = (vv[k] - in.mu) * in.r;
    txh[k] = in.r * (tt[k] - in.tmu - xh[k] * in.a);
  }
}
__device__ __forceinline__ GnJvpBwdIn gn_jvp_bwd_in(const GnJvpBwd& a, int slab) {
  return GnJvpBwdIn{a.dout, a.tdout, a.out_mask, a.y, a.ty, a.stats[(size_t)slab * 2], a.stats[(size_t)slab * 2 + 1],
                    a.tstats[(size_t)slab * 2], a.tstats[(size_t)slab * 2 + 1], a.relu};
}
template <bool DEV>
__device__ __forceinline__ void gn_jvp_bwd_sums(const GnJvpBwd& a, double* s_red, float (*s_ch)[8]) {
  const int ch = blockIdx.x, g = blockIdx.y, n = blockIdx.z, Cg = a.C / G, cq = Cg >> 2;
  const int slab = n * G + g;
  const GnJvpBwdIn in = gn_jvp_bwd_in(a, slab);
  const size_t base = (size_t)n * a.HW * a.C + (size_t)g * Cg;
  const int q = threadIdx.x % cq;                     // this thread's channel quad (cq divides 256, chunks start on a row)
  const int c = g * Cg + q * 4;
  const float4 ga4 = *reinterpret_cast<const float4*>(a.gamma + c), tg4 = *reinterpret_cast<const float4*>(a.tgamma + c);
  const float gg[4] = {ga4.x, ga4.y, ga4.z, ga4.w}, tgg[4] = {tg4.x, tg4.y, tg4.z, tg4.w};
  const int r0 = ch * a.rows, r1 = (r0 + a.rows < a.HW) ? r0 + a.rows : a.HW;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  float chb[4] = {0.f, 0.f, 0.f, 0.f}, chg[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = r0 * cq + threadIdx.x; i < r1 * cq; i += 256) {
    const size_t off = base + (size_t)(i / cq) * a.C + (size_t)q * 4;
    float d[4], td[4], xh[4], txh[4];
    gn_jvp_bwd_fetch(in, off, d, td, xh, txh);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dxh = gg[k] * d[k], tdxh = tgg[k] * d[k] + gg[k] * td[k];
      acc[0] += (double)dxh;
      acc[1] += (double)dxh * xh[k];
      acc[2] += (double)tdxh;
      acc[3] += (double)tdxh * xh[k] + (double)dxh * txh[k];
      chb[k] += td[k];
      chg[k] += td[k] * xh[k] + d[k] * txh[k];
    }
  }
  block_sum<4>(acc, s_red);
  if (threadIdx.x == 0) {
    double* o = a.part + ((size_t)slab * gridDim.x + ch) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (DEV) jvp_store_dev(o + k, acc[k]);
      else o[k] = acc[k];
    }
  }
  // per-channel sums: threads with the same channel quad sit 'cq' apart
#pragma unroll
  for (int k = 0; k < 4; ++k) { s_ch[threadIdx.x][k] = chb[k]; s_ch[threadIdx.x][4 + k] = chg[k]; }
  __syncthreads();
  if ((int)threadIdx.x < cq) {
    float sb[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = threadIdx.x; j < 256; j += cq)
#pragma unroll
      for (int k = 0; k < 4; ++k) { sb[k] += s_ch[j][k]; sg[k] += s_ch[j][4 + k]; }
    float* ob = a.tdgb + (((size_t)n * gridDim.x + ch) * 2 + 0) * a.C + c;
    float* og = a.tdgb + (((size_t)n * gridDim.x + ch) * 2 + 1) * a.C + c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (DEV) { jvp_store_dev(ob + k, sb[k]); jvp_store_dev(og + k, sg[k]); }
      else { ob[k] = sb[k]; og[k] = sg[k]; }
    }
  }
}
// the group's channel sums over images and chunks (rows of tdgb in (image, chunk) order): tangents of dbeta / dgamma
template <bool DEV>
__device__ __forceinline__ void gn_jvp_bwd_channels(const GnJvpBwd& a) {
  const int g = blockIdx.y, Cg = a.C / G;
  const int nrows = gridDim.x * gridDim.z;
  for (int cc = threadIdx.x; cc < Cg; cc += 256) {
    const int c = g * Cg + cc;
    float sb = 0.f, sg = 0.f;
    for (int rw = 0; rw < nrows; ++rw) {
      if constexpr (DEV) {
        sb += jvp_load_dev(a.tdgb + ((size_t)rw * 2 + 0) * a.C + c);
        sg += jvp_load_dev(a.tdgb + ((size_t)rw * 2 + 1) * a.C + c);
      } else {
        sb += a.tdgb[((size_t)rw * 2 + 0) * a.C + c];
        sg += a.tdgb[((size_t)rw * 2 + 1) * a.C + c];
      }
    }
    a.tdbeta[c] = sb;
    a.tdgamma[c] = sg;
  }
}
// acc: the slab's totals (sum dxh, sum dxh xhat, sum tdxh, sum (tdxh xhat + dxh txhat))
__device__ __forceinline__ void gn_jvp_bwd_apply(const GnJvpBwd& a, const double (&acc)[4]) {
  const int ch = blockIdx.x, g = blockIdx.y, n = blockIdx.z, Cg = a.C / G, cq = Cg >> 2;
  const int slab = n * G + g;
  const GnJvpBwdIn in = gn_jvp_bwd_in(a, slab);
  const float r = in.r, tr = -r * r * in.a;
  const size_t base = (size_t)n * a.HW * a.C + (size_t)g * Cg;
  const int q = threadIdx.x % cq;
  const int c = g * Cg + q * 4;
  const float4 ga4 = *reinterpret_cast<const float4*>(a.gamma + c), tg4 = *reinterpret_cast<const float4*>(a.tgamma + c);
  const float gg[4] = {ga4.x, ga4.y, ga4.z, ga4.w}, tgg[4] = {tg4.x, tg4.y, tg4.z, tg4.w};
  const double cnt = (double)a.HW * (double)Cg;
  const float c1 = (float)(acc[0] / cnt), c2 = (float)(acc[1] / cnt), tc1 = (float)(acc[2] / cnt), tc2 = (float)(acc[3] / cnt);
  const int r0 = ch * a.rows, r1 = (r0 + a.rows < a.HW) ? r0 + a.rows : a.HW;
  for (int i = r0 * cq + threadIdx.x; i < r1 * cq; i += 256) {
    const size_t off = base + (size_t)(i / cq) * a.C + (size_t)q * 4;
    float d[4], td[4], xh[4], txh[4];
    gn_jvp_bwd_fetch(in, off, d, td, xh, txh);
    float o[4], to[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dxh = gg[k] * d[k], tdxh = tgg[k] * d[k] + gg[k] * td[k];
      const float core = dxh - c1 - xh[k] * c2;
      o[k] = r * core;
      to[k] = tr * core + r * (tdxh - tc1 - txh[k] * c2 - xh[k] * tc2);
    }
    if (a.dm) *reinterpret_cast<float4*>(a.dm + off) = make_float4(d[0], d[1], d[2], d[3]);
    if (a.tdm) *reinterpret_cast<float4*>(a.tdm + off) = make_float4(td[0], td[1], td[2], td[3]);
    *reinterpret_cast<float4*>(a.dy + off) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(a.tdy + off) = make_float4(to[0], to[1], to[2], to[3]);
  }
}
__global__ __launch_bounds__(256) void gn_jvp_bwd_sums_kernel(GnJvpBwd a) {
  __shared__ double s_red[4 * 4];
  __shared__ float s_ch[256][8];
  gn_jvp_bwd_sums<false>(a, s_red, s_ch);
}
__global__ __launch_bounds__(256) void gn_jvp_bwd_apply_kernel(GnJvpBwd a) {
  if (blockIdx.x == 0 && blockIdx.z == 0) gn_jvp_bwd_channels<false>(a);
  double acc[4];
  chunk_total<4>(a.part, blockIdx.z * G + blockIdx.y, gridDim.x, acc);
  gn_jvp_bwd_apply(a, acc);
}
// one launch: the gradient tensors may alias nothing the sums phase of ANOTHER workgroup still reads - dm / tdm / dy / tdy are written
// only after this workgroup's slab has met, and a workgroup reads only its own rows of dout / tdout (hvp_engine.inc hands in distinct
// buffers anyway)
__global__ __launch_bounds__(256) void gn_jvp_bwd_onepass_kernel(GnJvpBwd a) {
  __shared__ double s_red[4 * 4];
  __shared__ float s_ch[256][8];
  __shared__ double s_tot[4];
  gn_jvp_bwd_sums<true>(a, s_red, s_ch);
  double acc[4];
  gn_jvp_meet<4>(a.part, a.sync, blockIdx.z * G + blockIdx.y, acc, s_tot);
  gn_jvp_bwd_apply(a, acc);
  if (blockIdx.x == 0 && blockIdx.z == 0) {
    // this group's channel sums need every image's chunks: wait for the other slabs of the group as well (last: the rows above are out)
    if (threadIdx.x == 0)
      for (int n = 1; n < (int)gridDim.z; ++n) jvp_wait(a.sync + n * G + blockIdx.y, gridDim.x, a.sync - 1);
    __syncthreads();
    gn_jvp_bwd_channels<true>(a);
  }
}

// tangent of MaxPool2d(3, 2, 1): ty[j] = tx[winning tap of j] (the tap index the forward stored, one byte per channel)
__global__ __launch_bounds__(256) void maxpool_jvp_fwd_kernel(const float* __restrict__ tx, const uint32_t* __restrict__ idx,
                                                              float* __restrict__ ty, int N, int H, int W, int C, int Ho, int Wo) {
  const int CQ = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * CQ;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cq = (int)(i % CQ);
    size_t t = i / CQ;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho), n = (int)(t / Ho);
    const uint32_t id = idx[i];
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int tap = (int)((id >> (8 * k)) & 255u);
      const int hi = ho * 2 - 1 + tap / 3, wi = wo * 2 - 1 + tap % 3;
      o[k] = tx[(((size_t)n * H + hi) * W + wi) * C + (size_t)cq * 4 + k];
    }
    *reinterpret_cast<float4*>(ty + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ty2 (may be NULL): second half of the tangent, added into ty by the sums launch; scratch: dyb_gn_jvp_scratch_floats floats.
// sync (may be NULL: two launches): dyb_gn_jvp_sync_words(N) zeroed words nobody else uses until the launch has finished -> one launch
static int gn_jvp_fwd_launch(const float* y, float* ty, const float* ty2, const float* stats, const float* gamma, const float* beta,
                             const float* tgamma, const float* tbeta, const float* res, const float* tres, float* out, float* tout,
                             float* tstats, float* scratch, unsigned* sync, int N, int HW, int C, int relu, hipStream_t st) {
  DYB_REQUIRE(y && ty && stats && gamma && beta && tgamma && tbeta && tout && scratch, DYB_ERR_ARG);
  DYB_REQUIRE(C % 16 == 0 && 256 % (C / G / 4) == 0 && N > 0 && HW > 0, DYB_ERR_UNSUPPORTED);
  DYB_REQUIRE(dyb_rep_current().n == 1, DYB_ERR_UNSUPPORTED);
  const GnJvpGeom ge = gn_jvp_geom(N, HW, C);
  GnJvpFwd a{y, ty, ty2, stats, reinterpret_cast<double*>(scratch), gamma, beta, tgamma, tbeta, res, tres, out, tout, tstats,
             sync ? sync + 1 : nullptr, HW, C, ge.rows, relu};
  const dim3 grid(ge.chunks, G, N);
  if (sync) {
    hipLaunchKernelGGL(gn_jvp_fwd_onepass_kernel, grid, dim3(256), 0, st, a);
    DYB_CHECK_LAUNCH();
    return DYB_OK;
  }
  hipLaunchKernelGGL(gn_jvp_fwd_sums_kernel, grid, dim3(256), 0, st, a);
  DYB_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_jvp_fwd_apply_kernel, grid, dim3(256), 0, st, a);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// scratch: dyb_gn_jvp_scratch_floats floats; tdbeta / tdgamma [C] receive the channel sums over images and chunks
static int gn_jvp_bwd_launch(const float* dout, const float* tdout, const float* out_mask, const float* y, const float* ty,
                             const float* stats, const float* tstats, const float* gamma, const float* tgamma, float* dm, float* tdm,
                             float* dy, float* tdy, float* scratch, unsigned* sync, float* tdgamma, float* tdbeta, int N, int HW, int C,
                             int relu, hipStream_t st) {
  DYB_REQUIRE(dout && tdout && y && ty && stats && tstats && gamma && tgamma && dy && tdy && scratch && tdgamma && tdbeta, DYB_ERR_ARG);
  DYB_REQUIRE(!relu || out_mask, DYB_ERR_ARG);
  DYB_REQUIRE(C % 16 == 0 && 256 % (C / G / 4) == 0 && N > 0 && HW > 0, DYB_ERR_UNSUPPORTED);
  DYB_REQUIRE(dyb_rep_current().n == 1, DYB_ERR_UNSUPPORTED);
  const GnJvpGeom ge = gn_jvp_geom(N, HW, C);
  GnJvpBwd a{dout, tdout, out_mask, y, ty, stats, tstats, gamma, tgamma, reinterpret_cast<double*>(scratch), dm, tdm, dy, tdy,
             scratch + (size_t)N * G * ge.chunks * 4 * 2, tdbeta, tdgamma, sync ? sync + 1 : nullptr, HW, C, ge.rows, relu};
  const dim3 grid(ge.chunks, G, N);
  if (sync) {
    hipLaunchKernelGGL(gn_jvp_bwd_onepass_kernel, grid, dim3(256), 0, st, a);
    DYB_CHECK_LAUNCH();
    return DYB_OK;
  }
  hipLaunchKernelGGL(gn_jvp_bwd_sums_kernel, grid, dim3(256), 0, st, a);
  DYB_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_jvp_bwd_apply_kernel, grid, dim3(256), 0, st, a);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
extern "C" int dyb_gn_jvp_fwd(const float* y, float* ty, const float* ty2, const float* stats, const float* gamma, const float* beta,
                              const float* tgamma, const float* tbeta, const float* res, const float* tres, float* out, float* tout,
                              float* tstats, float* scratch, int N, int HW, int C, int relu, hipStream_t st) {
  return gn_jvp_fwd_launch(y, ty, ty2, stats, gamma, beta, tgamma, tbeta, res, tres, out, tout, tstats, scratch, nullptr, N, HW, C, relu, st);
}
extern "C" int dyb_gn_jvp_bwd(const float* dout, const float* tdout, const float* out_mask, const float* y, const float* ty,
                              const float* stats, const float* tstats, const float* gamma, const float* tgamma, float* dm, float* tdm,
                              float* dy, float* tdy, float* scratch, float* tdgamma, float* tdbeta, int N, int HW, int C, int relu,
                              hipStream_t st) {
  return gn_jvp_bwd_launch(dout, tdout, out_mask, y, ty, stats, tstats, gamma, tgamma, dm, tdm, dy, tdy, scratch, nullptr, tdgamma, tdbeta,
                           N, HW, C, relu, st);
}
// the same as ONE launch each: sync = dyb_gn_jvp_sync_words(N) 32-bit words (the error word, then one arrival counter per (image,
// group) slab), zero on entry, private to the call until it has finished on the stream
extern "C" size_t dyb_gn_jvp_sync_words(int N) { return N > 0 ? (size_t)N * G + 1 : 0; }
extern "C" int dyb_gn_jvp_fwd_onepass(const float* y, float* ty, const float* ty2, const float* stats, const float* gamma,
                                      const float* beta, const float* tgamma, const float* tbeta, const float* res, const float* tres,
                                      float* out, float* tout, float* tstats, float* scratch, unsigned* sync, int N, int HW, int C,
                                      int relu, hipStream_t st) {
  DYB_REQUIRE(sync, DYB_ERR_ARG);
  return gn_jvp_fwd_launch(y, ty, ty2, stats, gamma, beta, tgamma, tbeta, res, tres, out, tout, tstats, scratch, sync, N, HW, C, relu, st);
}
extern "C" int dyb_gn_jvp_bwd_onepass(const float* dout, const float* tdout, const float* out_mask, const float* y, const float* ty,
                                      const float* stats, const float* tstats, const float* gamma, const float* tgamma, float* dm,
                                      float* tdm, float* dy, float* tdy, float* scratch, unsigned* sync, float* tdgamma, float* tdbeta,
                                      int N, int HW, int C, int relu, hipStream_t st) {
  DYB_REQUIRE(sync, DYB_ERR_ARG);
  return gn_jvp_bwd_launch(dout, tdout, out_mask, y, ty, stats, tstats, gamma, tgamma, dm, tdm, dy, tdy, scratch, sync, tdgamma, tdbeta, N,
                           HW, C, relu, st);
}
extern "C" int dyb_maxpool3x3s2_jvp(const float* tx, const uint32_t* idx, float* ty, int N, int H, int W, int C, hipStream_t st) {
  DYB_REQUIRE(tx && idx && ty && C % 4 == 0, DYB_ERR_ARG);
  DYB_REQUIRE(dyb_rep_current().n == 1, DYB_ERR_UNSUPPORTED);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(maxpool_jvp_fwd_kernel, dim3(blocks), dim3(256), 0, st, tx, idx, ty, N, H, W, C, Ho, Wo);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
