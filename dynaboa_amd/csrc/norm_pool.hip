// GroupNorm(4 groups)+ReLU(+residual) forward/backward, 3x3/s2 max-pool, 7x7 average pool and the
// NCHW->NHWC4 image repack, all NHWC fp32, 16 B per lane.
//
// Replaces nn.GroupNorm / nn.ReLU / nn.MaxPool2d / nn.AvgPool2d and the `out += residual` of
// reference model/hmr.py:14-18,40-60,138-156.  Elementwise / reduction work, every tensor streamed
// once, coalesced along channels; at batch 1 each kernel is a few dependent memory round trips.
//
// GroupNorm is split so that no kernel needs a grid-wide barrier:
//   gn_stats      : per (image, chunk-of-rows, group) partial sum / sum-of-squares (one wave per group);
//                   folds the split-K slabs left by the convolution and writes y.
//   gn_apply      : every workgroup re-reduces the (<= 256) partials in double, then
//                   out = relu?((y-mean)*rstd*gamma + beta (+ residual)); saves mean/rstd.  In the engine it runs
//                   only for block outputs; inside a bottleneck the consumer conv normalises on load (igemm_conv.hip).
//   gn_bwd_reduce : folds the incoming gradient (split-K slabs + residual edge), ReLU mask, dm, per-channel and
//                   per-group partial sums.  In the engine the other half of the backward - dy from (dm, y) and the
//                   coefficients - happens in the data- / weight-gradient convs' loaders; gn_bwd_apply is the
//                   stand-alone form behind dyb_groupnorm_bwd.
#include <hip/hip_ext.h>

#include "dyb_common.h"

#define G DYB_GN_GROUPS

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// grid (nchunks, N), block 256 = 4 waves, wave w <-> group w
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ src, float* __restrict__ y, int nslabs,
                                                       size_t slab_stride, float* __restrict__ partials, int HW,
                                                       int C, int rows_per_chunk, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, src); DYB_RB(R, y); DYB_RB(R, partials);
  const int n = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cqg = C >> 4;                        // float4 columns per group
  const int row0 = chunk * rows_per_chunk;
  int row1 = row0 + rows_per_chunk;
  if (row1 > HW) row1 = HW;
  const int items = (row1 - row0) * cqg;
  float s1 = 0.f, s2 = 0.f;
  for (int idx = lane; idx < items; idx += 64) {
    int row = row0 + idx / cqg;
    int cq = wave * cqg + idx % cqg;
    size_t off = ((size_t)n * HW + row) * C + (size_t)cq * 4;
    float4 v = *reinterpret_cast<const float4*>(src + off);
    if (nslabs > 1) {
      // eight slab loads in flight per round trip (the fold is the latency chain of this kernel)
      for (int z0 = 1; z0 < nslabs; z0 += 8) {
        float4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          t[j] = (z0 + j < nslabs) ? *reinterpret_cast<const float4*>(src + (size_t)(z0 + j) * slab_stride + off)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        v.x += ((t[0].x + t[1].x) + (t[2].x + t[3].x)) + ((t[4].x + t[5].x) + (t[6].x + t[7].x));
        v.y += ((t[0].y + t[1].y) + (t[2].y + t[3].y)) + ((t[4].y + t[5].y) + (t[6].y + t[7].y));
        v.z += ((t[0].z + t[1].z) + (t[2].z + t[3].z)) + ((t[4].z + t[5].z) + (t[6].z + t[7].z));
        v.w += ((t[0].w + t[1].w) + (t[2].w + t[3].w)) + ((t[4].w + t[5].w) + (t[6].w + t[7].w));
      }
      *reinterpret_cast<float4*>(y + off) = v;
    }
    s1 += (v.x + v.y) + (v.z + v.w);
    s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  s1 = dyb_wave_sum(s1);
  s2 = dyb_wave_sum(s2);
  if (lane == 0) {
    float* p = partials + (((size_t)n * nchunks + chunk) * G + wave) * 2;
    p[0] = s1;
    p[1] = s2;
  }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// The residual operand may itself be a GroupNorm (no ReLU) of a raw conv output - the shortcut
// branch's downsample.1 - normalised here on the fly from its own partials instead of in a launch of
// its own.
struct GnResidual {
  const float* y;          // plain residual, or the raw conv output to normalise when partials != NULL
  const float* partials;   // [N][nchunks][G][2] or NULL
  const float* gamma;
  const float* beta;
  float* stats;            // [N][G][2] saved (mean, rstd) of the residual's GroupNorm
  int nchunks;
};

// grid (blocks_per_image, N), block 256
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ y, const float* __restrict__ partials,
                                                       int nchunks, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, GnResidual rs,
                                                       float* __restrict__ out, float* __restrict__ stats, int HW, int C,
                                                       int relu, float eps, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, y); DYB_RB(R, partials); DYB_RB(R, gamma); DYB_RB(R, beta); DYB_RB(R, out); DYB_RB(R, stats);
  DYB_RB(R, rs.y); DYB_RB(R, rs.partials); DYB_RB(R, rs.gamma); DYB_RB(R, rs.beta); DYB_RB(R, rs.stats);
  __shared__ float s_mean[2][G], s_rstd[2][G];
  const int n = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int CQ = C >> 2, cqg = C >> 4;
  const size_t total = (size_t)HW * CQ;
  const size_t base = (size_t)n * HW * C;
  const float* res = rs.y;
  const bool res_gn = rs.partials != nullptr;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // issue this thread's first element loads BEFORE the statistics prologue: the two memory round
  // trips (partials, data) then overlap instead of adding up (these kernels are latency-bound)
  const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  float4 v0 = zero4, ga0 = zero4, be0 = zero4, r0 = zero4, rg0 = zero4, rb0 = zero4;
  if (i0 < total) {
    int cq = (int)(i0 % CQ);
    v0 = *reinterpret_cast<const float4*>(y + base + i0 * 4);
    ga0 = *reinterpret_cast<const float4*>(gamma + cq * 4);
    be0 = *reinterpret_cast<const float4*>(beta + cq * 4);
    if (res) r0 = *reinterpret_cast<const float4*>(res + base + i0 * 4);
    if (res_gn) {
      rg0 = *reinterpret_cast<const float4*>(rs.gamma + cq * 4);
      rb0 = *reinterpret_cast<const float4*>(rs.beta + cq * 4);
    }
  }
  for (int which = 0; which < (res_gn ? 2 : 1); ++which) {
    const float* pp = which ? rs.partials : partials;
    const int nch = which ? rs.nchunks : nchunks;
    double a = 0.0, b = 0.0;
    for (int c0 = lane; c0 < nch; c0 += 256) {
      float2 t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {          // clamped unconditional loads, masked below: a per-lane conditional load is waited for on the spot
        const int ch = c0 + 64 * j;
        t[j] = *reinterpret_cast<const float2*>(pp + (((size_t)n * nch + (ch < nch ? ch : nch - 1)) * G + wave) * 2);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + 64 * j >= nch) t[j] = make_float2(0.f, 0.f);
      a += ((double)t[0].x + (double)t[1].x) + ((double)t[2].x + (double)t[3].x);
      b += ((double)t[0].y + (double)t[1].y) + ((double)t[2].y + (double)t[3].y);
    }
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    if (lane == 0) {
      double cnt = (double)HW * (double)(C / G);
      double mean = a / cnt;
      double var = b / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)eps));
      s_mean[which][wave] = m;
      s_rstd[which][wave] = r;
      if (blockIdx.x == 0) {
        float* so = which ? rs.stats : stats;
        so[((size_t)n * G + wave) * 2 + 0] = m;
        so[((size_t)n * G + wave) * 2 + 1] = r;
      }
    }
  }
  __syncthreads();
  for (size_t i = i0; i < total; i += (size_t)gridDim.x * 256) {
    int cq = (int)(i % CQ);
    int g = cq / cqg;
    float mean = s_mean[0][g], rstd = s_rstd[0][g];
    float4 v, ga, be, r, rg, rb;
    if (i == i0) {
      v = v0; ga = ga0; be = be0; r = r0; rg = rg0; rb = rb0;
    } else {
      v = *reinterpret_cast<const float4*>(y + base + i * 4);
      ga = *reinterpret_cast<const float4*>(gamma + cq * 4);
      be = *reinterpret_cast<const float4*>(beta + cq * 4);
      r = res ? *reinterpret_cast<const float4*>(res + base + i * 4) : zero4;
      rg = res_gn ? *reinterpret_cast<const float4*>(rs.gamma + cq * 4) : zero4;
      rb = res_gn ? *reinterpret_cast<const float4*>(rs.beta + cq * 4) : zero4;
    }
    float4 o;
    o.x = fmaf((v.x - mean) * rstd, ga.x, be.x);
    o.y = fmaf((v.y - mean) * rstd, ga.y, be.y);
    o.z = fmaf((v.z - mean) * rstd, ga.z, be.z);
    o.w = fmaf((v.w - mean) * rstd, ga.w, be.w);
    if (res_gn) {
      float rm = s_mean[1][g], rr = s_rstd[1][g];
      r.x = fmaf((r.x - rm) * rr, rg.x, rb.x);
      r.y = fmaf((r.y - rm) * rr, rg.y, rb.y);
      r.z = fmaf((r.z - rm) * rr, rg.z, rb.z);
      r.w = fmaf((r.w - rm) * rr, rg.w, rb.w);
    }
    if (res) { o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
    if (relu) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    *reinterpret_cast<float4*>(out + base + i * 4) = o;
  }
}

// Under the throughput policy a launch covers many sequence replicas: the chunk counts then come from a workgroup budget over
// the whole launch (dyb_gn_replica_share) - sized for one sequence they make every GroupNorm launch 3-8 k workgroups at 32
// sequences, and workgroup dispatch (~90 / us) rather than HBM bounds it.  Producer and consumers of a partial block
// call these functions inside the same replica scope, so they agree.
static int gn_chunks(int HW, int N) {
  const int share = dyb_gn_replica_share(N);
  const int plan = 256 / (N > 0 ? N : 1);            // what the partial slots are sized for (plans are made outside any replica scope)
  int want = share > 0 && share < plan ? share : plan;   // the share only ever lowers the count: never past the slot (tp_min / tp_gn_wgs are public switches)
  if (want < 1) want = 1;
  int nch = HW < want ? HW : want;
  int rows = dyb_cdiv(HW, nch);
  return dyb_cdiv(HW, rows);
}

// backward: ~256 workgroups in the reduce kernel (its per-thread row loop is the latency chain),
// at most 128 chunks (the apply kernel folds them 4 lanes wide per channel)
static int gn_chunks_bwd(int HW, int N, int C) {
  int CQ = C / 4;
  int colblocks = CQ > 256 ? CQ / 256 : 1;
  const int share = dyb_gn_replica_share(N);
  const int plan = 256 / (N * colblocks);            // plan-time sizing; the replica share may only lower it (slot capacity)
  int want = share > 0 && share / colblocks < plan ? share / colblocks : plan;
  if (want < 1) want = 1;
  if (want > 128) want = 128;
  int nch = HW < want ? HW : want;
  int rows = dyb_cdiv(HW, nch);
  return dyb_cdiv(HW, rows);
}

extern "C" size_t dyb_groupnorm_workspace_bytes(int N, int HW, int C) {
  // forward partials [N][nchunks][G][2]; backward partials [N][nchunks][2][C] + coef [N][G][2]
  size_t nch = (size_t)gn_chunks(HW, N), nchb = (size_t)gn_chunks_bwd(HW, N, C);
  size_t fwd = (size_t)N * nch * G * 2;
  size_t bwd = (size_t)N * nchb * 2 * C + (size_t)N * nchb * 2 * G * 2;      // + gpart (<= 2 column blocks)
  return (fwd > bwd ? fwd : bwd) * sizeof(float);
}

int dyb_gn_fwd_chunks(int N, int HW) { return gn_chunks(HW, N); }

// statistics half: folds the split-K slabs (nslabs > 1) into y and leaves per-chunk (sum, sum of
// squares) partials [N][dyb_gn_fwd_chunks][G][2] in `partials`
extern "C" int dyb_groupnorm_stats(const float* slabs, int nslabs, float* y, float* partials, int N, int HW, int C,
                                   hipStream_t st) {
  DYB_REQUIRE(y && partials, DYB_ERR_ARG);
  DYB_REQUIRE(C % 16 == 0 && N > 0 && HW > 0, DYB_ERR_UNSUPPORTED);
  DYB_REQUIRE(nslabs >= 1 && (nslabs == 1 || slabs), DYB_ERR_ARG);
  int nch = gn_chunks(HW, N);
  int rows = dyb_cdiv(HW, nch);
  const float* src = nslabs > 1 ? slabs : y;
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nch, N, R.n), dim3(256), 0, st, src, y, nslabs, (size_t)N * HW * C, partials, HW,
                     C, rows, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// apply half: out = relu?(gn(y) + residual), the residual being absent (NULL), plain, or - when
// res_partials != NULL - the GroupNorm (no ReLU) of the raw conv output `residual` with its own
// partials / gamma / beta (its (mean, rstd) are saved to res_stats).
// explicit partial counts: nch (res_nch) = number of [G][2] partial records per image behind `partials`
// (`res_partials`) - dyb_gn_fwd_chunks(N, HW) when they come from dyb_groupnorm_stats, the tile count when a conv
// wrote them in its epilogue (igemm_conv.hip, K4)
extern "C" int dyb_groupnorm_apply_n(const float* y, const float* partials, int nch, const float* gamma, const float* beta,
                                     const float* residual, const float* res_partials, int res_nch, const float* res_gamma,
                                     const float* res_beta, float* res_stats, float* out, float* stats, int N, int HW, int C,
                                     int relu, hipStream_t st) {
  DYB_REQUIRE(y && partials && gamma && beta && out && stats && nch > 0, DYB_ERR_ARG);
  DYB_REQUIRE(!res_partials || (residual && res_gamma && res_beta && res_stats && res_nch > 0), DYB_ERR_ARG);
  DYB_REQUIRE(C % 16 == 0 && N > 0 && HW > 0, DYB_ERR_UNSUPPORTED);
  size_t total4 = (size_t)HW * (C / 4);
  int bpi = (int)((total4 + 1023) / 1024);
  const int share = dyb_gn_replica_share(N);
  int cap = share > 0 ? share : (1024 / N > 1 ? 1024 / N : 1);
  if (bpi > cap) bpi = cap;
  if (bpi < 1) bpi = 1;
  GnResidual rs{residual, res_partials, res_gamma, res_beta, res_stats, res_nch};
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(gn_apply_kernel, dim3(bpi, N, R.n), dim3(256), 0, st, y, partials, nch, gamma, beta, rs, out, stats, HW, C,
                     relu, DYB_GN_EPS, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
extern "C" int dyb_groupnorm_apply(const float* y, const float* partials, const float* gamma, const float* beta,
                                   const float* residual, const float* res_partials, const float* res_gamma,
                                   const float* res_beta, float* res_stats, float* out, float* stats, int N, int HW, int C,
                                   int relu, hipStream_t st) {
  const int nch = gn_chunks(HW, N);
  return dyb_groupnorm_apply_n(y, partials, nch, gamma, beta, residual, res_partials, nch, res_gamma, res_beta, res_stats, out,
                               stats, N, HW, C, relu, st);
}

// y: conv output [N][HW][C] (written here when nslabs > 1 from `slabs`), out: normalised result,
// stats: [N][G][2] (mean, rstd) saved for backward.
extern "C" int dyb_groupnorm_fwd(const float* slabs, int nslabs, float* y, const float* gamma, const float* beta,
                                 const float* residual, float* out, float* stats, int N, int HW, int C, int relu,
                                 void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(y && gamma && beta && out && stats && ws, DYB_ERR_ARG);
  DYB_REQUIRE(C % 16 == 0 && N > 0 && HW > 0, DYB_ERR_UNSUPPORTED);
  int nch = gn_chunks(HW, N);
  DYB_REQUIRE(ws_bytes >= (size_t)N * nch * G * 2 * sizeof(float), DYB_ERR_WORKSPACE);
  float* partials = reinterpret_cast<float*>(ws);
  int rc = dyb_groupnorm_stats(slabs, nslabs, y, partials, N, HW, C, st);
  if (rc != DYB_OK) return rc;
  return dyb_groupnorm_apply(y, partials, gamma, beta, residual, nullptr, nullptr, nullptr, nullptr, out, stats, N, HW, C,
                             relu, st);
}

// ------------------------------------------------------------------------------------------
// backward  (two launches)
//   gn_bwd_reduce : per (image, chunk of rows) per-channel sums A_c = sum dy, B_c = sum dy*xhat
//                   -> partials [n][chunk][2][C], and per-group sums of gamma*A, gamma*B over the
//                   workgroup's channel span -> gpart [n][chunk][colblock][G][2]
//   gn_bwd_apply  : every workgroup folds gpart (<= 32 chunks x <= 2 column blocks) into the two
//                   per-group coefficients, then dx = rstd*(gamma*dy - c1 - xhat*c2); the first
//                   ceil(C/256) workgroups also fold the per-channel partials into dgamma / dbeta.
// ------------------------------------------------------------------------------------------
// grid (CQ/TX, nchunks, N), block 256 = TX x TY
// If nslabs > 1 or addend != NULL the incoming gradient is  sum_z dout[z*slab_stride + .] + addend[.]
// (the un-folded split-K slabs of the data-gradient convolution that produced it, plus the
// residual-edge gradient); it is folded here, once, and written to `folded` for the apply kernel.
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ dout, int nslabs, size_t slab_stride,
                                                            const float* __restrict__ addend, float* __restrict__ folded,
                                                            float* __restrict__ dm, const float* __restrict__ out,
                                                            const float* __restrict__ y, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ partials,
                                                            float* __restrict__ gpart, int HW, int C, int rows_per_chunk,
                                                            int relu, int TX, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, dout); DYB_RB(R, addend); DYB_RB(R, folded); DYB_RB(R, dm); DYB_RB(R, out); DYB_RB(R, y); DYB_RB(R, stats);
  DYB_RB(R, gamma); DYB_RB(R, beta); DYB_RB(R, partials); DYB_RB(R, gpart);
  __shared__ float sm[256 * 8];
  __shared__ float sg[256][2];
  const int n = (int)dyb_bz, chunk = blockIdx.y, nchunks = gridDim.y;
  const int TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cq = blockIdx.x * TX + tx;
  const int cqg = C >> 4;
  const int g = cq / cqg;
  const float mean = stats[((size_t)n * G + g) * 2], rstd = stats[((size_t)n * G + g) * 2 + 1];
  const float4 ga = *reinterpret_cast<const float4*>(gamma + (size_t)cq * 4);   // used after the loop: load it early
  const int row0 = chunk * rows_per_chunk;
  int row1 = row0 + rows_per_chunk;
  if (row1 > HW) row1 = HW;
  float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
  // out == NULL with relu: the activation was never materialised (it was normalised on the fly by its
  // one consumer); its sign is recomputed from y with exactly the consumer's expression
  const bool mask_y = relu && out == nullptr;
  const float4 be = mask_y ? *reinterpret_cast<const float4*>(beta + (size_t)cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  auto accumulate = [&](float4 d, float4 v, float4 o, size_t off) {
    if (mask_y) {
      o.x = fmaf((v.x - mean) * rstd, ga.x, be.x); o.y = fmaf((v.y - mean) * rstd, ga.y, be.y);
      o.z = fmaf((v.z - mean) * rstd, ga.z, be.z); o.w = fmaf((v.w - mean) * rstd, ga.w, be.w);
    }
    if (relu) {
      d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f;
      d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
    }
    if (dm) *reinterpret_cast<float4*>(dm + off) = d;     // masked gradient: residual-edge gradient / fused-dy operand
    a[0] += d.x; a[1] += d.y; a[2] += d.z; a[3] += d.w;
    b[0] += d.x * ((v.x - mean) * rstd); b[1] += d.y * ((v.y - mean) * rstd);
    b[2] += d.z * ((v.z - mean) * rstd); b[3] += d.w * ((v.w - mean) * rstd);
  };
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool fold = nslabs > 1 || addend != nullptr;
  int row = row0 + ty;
  if (!fold) {
    // plain gradient: two rows (six 16-byte loads) in flight per iteration - the loop is latency-bound
    for (; row + TY < row1; row += 2 * TY) {
      size_t o0 = ((size_t)n * HW + row) * C + (size_t)cq * 4, o1 = o0 + (size_t)TY * C;
      float4 d0 = *reinterpret_cast<const float4*>(dout + o0), d1 = *reinterpret_cast<const float4*>(dout + o1);
      float4 v0 = *reinterpret_cast<const float4*>(y + o0), v1 = *reinterpret_cast<const float4*>(y + o1);
      float4 q0 = (relu && out) ? *reinterpret_cast<const float4*>(out + o0) : zero4;
      float4 q1 = (relu && out) ? *reinterpret_cast<const float4*>(out + o1) : zero4;
      accumulate(d0, v0, q0, o0);
      accumulate(d1, v1, q1, o1);
    }
  }
  // folded gradient: two rows per trip - every load of both rows (base slab, y, mask source, addend, up to 8 more slabs
  // each) is issued before the first use; a missing second row re-reads the first and is dropped
  auto fold_row = [&](float4& d, const float4 (&t)[8]) {
    d.x += ((t[0].x + t[1].x) + (t[2].x + t[3].x)) + ((t[4].x + t[5].x) + (t[6].x + t[7].x));
    d.y += ((t[0].y + t[1].y) + (t[2].y + t[3].y)) + ((t[4].y + t[5].y) + (t[6].y + t[7].y));
    d.z += ((t[0].z + t[1].z) + (t[2].z + t[3].z)) + ((t[4].z + t[5].z) + (t[6].z + t[7].z));
    d.w += ((t[0].w + t[1].w) + (t[2].w + t[3].w)) + ((t[4].w + t[5].w) + (t[6].w + t[7].w));
  };
  for (; row < row1; row += 2 * TY) {
    const bool two = row + TY < row1;
    const size_t off0 = ((size_t)n * HW + row) * C + (size_t)cq * 4;
    const size_t off1 = two ? off0 + (size_t)TY * C : off0;
    float4 d0 = *reinterpret_cast<const float4*>(dout + off0), d1 = *reinterpret_cast<const float4*>(dout + off1);
    float4 v0 = *reinterpret_cast<const float4*>(y + off0), v1 = *reinterpret_cast<const float4*>(y + off1);
    float4 q0 = (relu && out) ? *reinterpret_cast<const float4*>(out + off0) : zero4;
    float4 q1 = (relu && out) ? *reinterpret_cast<const float4*>(out + off1) : zero4;
    if (fold) {
      float4 p0 = addend ? *reinterpret_cast<const float4*>(addend + off0) : zero4;
      float4 p1 = addend ? *reinterpret_cast<const float4*>(addend + off1) : zero4;
      d0.x += p0.x; d0.y += p0.y; d0.z += p0.z; d0.w += p0.w;
      d1.x += p1.x; d1.y += p1.y; d1.z += p1.z; d1.w += p1.w;
      for (int z0 = 1; z0 < nslabs; z0 += 8) {
        float4 t0[8], t1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const size_t zo = (size_t)(z0 + j < nslabs ? z0 + j : 0) * slab_stride;      // uniform clamp, zeroed below
          t0[j] = *reinterpret_cast<const float4*>(dout + zo + off0);
          t1[j] = *reinterpret_cast<const float4*>(dout + zo + off1);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (z0 + j >= nslabs) { t0[j] = zero4; t1[j] = zero4; }
        fold_row(d0, t0);
        fold_row(d1, t1);
      }
      if (folded) {
        *reinterpret_cast<float4*>(folded + off0) = d0;
        if (two) *reinterpret_cast<float4*>(folded + off1) = d1;
      }
    }
    accumulate(d0, v0, q0, off0);
    if (two) accumulate(d1, v1, q1, off1);
  }
  float* mine = sm + threadIdx.x * 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) { mine[j] = a[j]; mine[4 + j] = b[j]; }
  __syncthreads();
  if (ty == 0) {
    for (int t = 1; t < TY; ++t) {
      const float* o = sm + (t * TX + tx) * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] += o[j]; b[j] += o[4 + j]; }
    }
    float* p = partials + ((size_t)n * nchunks + chunk) * 2 * C + (size_t)cq * 4;
    *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<float4*>(p + C) = make_float4(b[0], b[1], b[2], b[3]);
    sg[tx][0] = (ga.x * a[0] + ga.y * a[1]) + (ga.z * a[2] + ga.w * a[3]);
    sg[tx][1] = (ga.x * b[0] + ga.y * b[1]) + (ga.z * b[2] + ga.w * b[3]);
  }
  __syncthreads();
  if (threadIdx.x < 2 * G) {
    // group q's channel-quads inside this workgroup's span [blockIdx.x*TX, +TX)
    const int q = threadIdx.x >> 1, which = threadIdx.x & 1;
    int lo = q * cqg - blockIdx.x * TX, hi = lo + cqg;
    if (lo < 0) lo = 0;
    if (hi > TX) hi = TX;
    float s = 0.f;
    for (int t = lo; t < hi; ++t) s += sg[t][which];
    gpart[((((size_t)n * nchunks + chunk) * gridDim.x + blockIdx.x) * G + q) * 2 + which] = s;
  }
}

// grid-stride elementwise over [N][HW][C]; workgroups [0, ceil(C/64)) also write dgamma/dbeta
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                           const float* __restrict__ y, const float* __restrict__ stats,
                                                           const float* __restrict__ partials, const float* __restrict__ gpart,
                                                           int nchunks, int ncolb, const float* __restrict__ gamma,
                                                           float* __restrict__ dy, float* __restrict__ dres,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int HW,
                                                           int C, int relu, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, dout); DYB_RB(R, out); DYB_RB(R, y); DYB_RB(R, stats); DYB_RB(R, partials); DYB_RB(R, gpart); DYB_RB(R, gamma);
  DYB_RB(R, dy); DYB_RB(R, dres); DYB_RB(R, dgamma); DYB_RB(R, dbeta);
  __shared__ float s_coef[16 * G * 2];          // N <= 16 per call path; larger N handled in slices below
  const int CQ = C >> 2, cqg = C >> 4;
  const float inv_m = 1.0f / ((float)(C / G) * (float)HW);
  // dgamma / dbeta: workgroup b < ceil(C/64) owns channels [64b, 64b+64); 4 lanes per channel split the
  // (n, chunk) range and meet in LDS
  __shared__ float s_gb[4][64][2];
  if (blockIdx.x * 64 < C) {
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float A = 0.f, B = 0.f;
    if (c < C) {
      const int total = N * nchunks;
      float A1 = 0.f, B1 = 0.f;
      int k = part;
      for (; k + 4 < total; k += 8) {
        const float* p0 = partials + (size_t)k * 2 * C + c;
        const float* p1 = partials + (size_t)(k + 4) * 2 * C + c;
        float a0 = p0[0], b0 = p0[C], a1 = p1[0], b1 = p1[C];
        A += a0; B += b0; A1 += a1; B1 += b1;
      }
      if (k < total) {
        const float* p0 = partials + (size_t)k * 2 * C + c;
        A += p0[0]; B += p0[C];
      }
      A += A1; B += B1;
    }
    s_gb[part][cl][0] = A;
    s_gb[part][cl][1] = B;
    __syncthreads();
    if (part == 0 && c < C) {
      dbeta[c] = (s_gb[0][cl][0] + s_gb[1][cl][0]) + (s_gb[2][cl][0] + s_gb[3][cl][0]);
      dgamma[c] = (s_gb[0][cl][1] + s_gb[1][cl][1]) + (s_gb[2][cl][1] + s_gb[3][cl][1]);
    }
  }
  __shared__ float s_red[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t per = (size_t)HW * CQ, total = (size_t)N * per;
  for (int n0 = 0; n0 < N; n0 += 16) {
    const int nn = (N - n0) < 16 ? (N - n0) : 16;
    const size_t lo = (size_t)n0 * per, hi = (size_t)(n0 + nn) * per;
    // first element of this slice: loads go out before the coefficient prologue (latency overlap)
    const size_t i0 = lo + (size_t)blockIdx.x * 256 + threadIdx.x;
    float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f), v0 = d0, ga0 = d0, o0 = d0;
    if (i0 < hi) {
      d0 = *reinterpret_cast<const float4*>(dout + i0 * 4);
      v0 = *reinterpret_cast<const float4*>(y + i0 * 4);
      ga0 = *reinterpret_cast<const float4*>(gamma + (int)(i0 % CQ) * 4);
      if (relu) o0 = *reinterpret_cast<const float4*>(out + i0 * 4);
    }
    // coefficients: all 256 threads fold the (chunk, column-block) partials, 8 values per sample
    const int K = nchunks * ncolb;
    for (int n = n0; n < n0 + nn; ++n) {
      const float* gp = gpart + (size_t)n * K * G * 2 + (threadIdx.x & 7);
      float sacc = 0.f;
      for (int k = threadIdx.x >> 3; k < K; k += 32) sacc += gp[(size_t)k * G * 2];
      sacc += __shfl_xor(sacc, 8);
      sacc += __shfl_xor(sacc, 16);
      sacc += __shfl_xor(sacc, 32);
      __syncthreads();
      if (lane < 8) s_red[wave][lane] = sacc;
      __syncthreads();
      if (threadIdx.x < 8)
        s_coef[(n - n0) * G * 2 + threadIdx.x] =
            ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x])) * inv_m;
    }
    __syncthreads();
    for (size_t i = i0; i < hi; i += (size_t)gridDim.x * 256) {
      int n = (int)(i / per);
      int cq = (int)(i % CQ);
      int g = cq / cqg;
      float mean = stats[((size_t)n * G + g) * 2], rstd = stats[((size_t)n * G + g) * 2 + 1];
      float c1 = s_coef[((n - n0) * G + g) * 2], c2 = s_coef[((n - n0) * G + g) * 2 + 1];
      float4 d, v, ga, o;
      if (i == i0) {
        d = d0; v = v0; ga = ga0; o = o0;
      } else {
        d = *reinterpret_cast<const float4*>(dout + i * 4);
        v = *reinterpret_cast<const float4*>(y + i * 4);
        ga = *reinterpret_cast<const float4*>(gamma + cq * 4);
        o = relu ? *reinterpret_cast<const float4*>(out + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (relu) {
        d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f;
        d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
      }
      if (dres) *reinterpret_cast<float4*>(dres + i * 4) = d;
      float4 r;
      r.x = rstd * (ga.x * d.x - c1 - ((v.x - mean) * rstd) * c2);
      r.y = rstd * (ga.y * d.y - c1 - ((v.y - mean) * rstd) * c2);
      r.z = rstd * (ga.z * d.z - c1 - ((v.z - mean) * rstd) * c2);
      r.w = rstd * (ga.w * d.w - c1 - ((v.w - mean) * rstd) * c2);
      *reinterpret_cast<float4*>(dy + i * 4) = r;
    }
  }
}

// dout: gradient w.r.t. the kernel's forward output (after residual add / ReLU); out: that forward
// output (ReLU mask); y/stats: saved conv output and (mean,rstd).  Writes dy (grad w.r.t. y),
// dgamma, dbeta and, if dres != NULL, the gradient flowing into the residual operand.
// Fold variant: the incoming gradient is sum_z dout_slabs[z] (+ addend); `folded` (required when
// nslabs > 1 or addend) receives the sum.  `folded` may alias neither input.
extern "C" int dyb_groupnorm_bwd_fold(const float* dout_slabs, int nslabs, size_t slab_stride, const float* addend,
                                      float* folded, const float* out, const float* y, const float* stats,
                                      const float* gamma, float* dy, float* dres, float* dgamma, float* dbeta, int N, int HW,
                                      int C, int relu, void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(dout_slabs && y && stats && gamma && dy && dgamma && dbeta && ws, DYB_ERR_ARG);
  DYB_REQUIRE(!relu || out, DYB_ERR_ARG);
  DYB_REQUIRE(nslabs >= 1, DYB_ERR_ARG);
  const bool fold = nslabs > 1 || addend != nullptr;
  DYB_REQUIRE(!fold || folded, DYB_ERR_ARG);
  DYB_REQUIRE(C % 16 == 0 && C <= 2048 && N > 0 && HW > 0, DYB_ERR_UNSUPPORTED);
  int nch = gn_chunks_bwd(HW, N, C);
  int CQ = C / 4;
  int TX = CQ < 256 ? CQ : 256;
  int ncolb = CQ / TX;
  size_t need = ((size_t)N * nch * 2 * C + (size_t)N * nch * ncolb * G * 2) * sizeof(float);
  DYB_REQUIRE(ws_bytes >= need, DYB_ERR_WORKSPACE);
  int rows = dyb_cdiv(HW, nch);
  float* partials = reinterpret_cast<float*>(ws);
  float* gpart = partials + (size_t)N * nch * 2 * C;
  const DybRep R1 = {1, 0, {}, {}, {}};       // the stand-alone backward is not replica-aware (its apply half is not)
  DYB_REQUIRE(dyb_rep_current().n == 1, DYB_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(ncolb, nch, N), dim3(256), 0, st, dout_slabs, nslabs, slab_stride, addend,
                     fold ? folded : (float*)nullptr, (float*)nullptr, out, y, stats, gamma, (const float*)nullptr, partials, gpart,
                     HW, C, rows, relu, TX, R1);
  DYB_CHECK_LAUNCH();
  const float* dsrc = fold ? folded : dout_slabs;
  size_t total4 = (size_t)N * HW * CQ;
  int blocks = (int)((total4 + 1023) / 1024);
  if (blocks > 2048) blocks = 2048;
  int minb = dyb_cdiv(C, 64);
  if (blocks < minb) blocks = minb;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, dsrc, out, y, stats, (const float*)partials,
                     (const float*)gpart, nch, ncolb, gamma, dy, dres, dgamma, dbeta, N, HW, C, relu, R1);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// The apply half on its own, for the engine's throughput schedule (several sequence replicas per launch): dy =
// rstd*(gamma*dm - c1 - xhat*c2) is MATERIALISED once per layer from the masked gradient dm and the partial block `part`
// (layout nch x ncolb, 0 = the reduce kernel's) instead of being re-formed in every tile of the data- and weight-gradient
// convolutions' loaders - with the chip full, that recomputation (x9 taps, x Cout/64 column tiles) is what those kernels
// spend their time on, while the extra launch it saves at one sequence per launch no longer matters.  Also folds dgamma / dbeta.
int dyb_gn_bwd_apply_dy(const float* dm, const float* y, const float* stats, const float* part, int nch, int ncolb,
                        const float* gamma, float* dy, float* dgamma, float* dbeta, int N, int HW, int C, hipStream_t st) {
  DYB_REQUIRE(dm && y && stats && part && gamma && dy && dgamma && dbeta, DYB_ERR_ARG);
  DYB_REQUIRE(C % 16 == 0 && C <= 2048 && N > 0 && HW > 0, DYB_ERR_UNSUPPORTED);
  if (nch <= 0 || ncolb <= 0) dyb_gn_bwd_layout(N, HW, C, &nch, &ncolb);
  const float* gpart = part + (size_t)N * nch * 2 * C;
  const int CQ = C / 4;
  size_t total4 = (size_t)N * HW * CQ;
  int blocks = (int)((total4 + 1023) / 1024);
  const int share = dyb_gn_replica_share(1);           // the grid-stride apply: a workgroup budget per replica (all images)
  if (blocks > (share > 0 ? 2 * share : 2048)) blocks = share > 0 ? 2 * share : 2048;
  int minb = dyb_cdiv(C, 64);
  if (blocks < minb) blocks = minb;
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(blocks, 1, R.n), dim3(256), 0, st, dm, (const float*)nullptr, y, stats, part, gpart, nch,
                     ncolb, gamma, dy, (float*)nullptr, dgamma, dbeta, N, HW, C, 0, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// ---- one-pass backward (throughput schedule, one image per sequence replica) ------------------------------------------------
// The two-launch backward streams every tensor of a layer twice: the reduce reads the incoming gradient, y and the ReLU mask
// source and writes the masked gradient dm; the apply reads dm and y again and writes dy - 6 to 7 tensor passes per layer, the
// second-largest kernel family of a frame step at 32 sequences (profiles/r03_final_kernel_stats_S32.csv).  Here a workgroup (256
// work-items by default) keeps its share of an (image, group) slab IN REGISTERS between the two phases (8 x 16 bytes of masked gradient
// and of xhat per work-item), so the layer costs one read of each input and one write of each output: 3 passes for the layers
// inside a bottleneck (gradient, y -> dy), 5 for a block output (gradient, y, activation -> dm, dy).
//   grid (k, G, replicas): workgroup (c, g, r) owns rows [c * rows, (c + 1) * rows) of replica r's slab of group g;
//   k = 1  (slab <= 2048 float4: the 7x7 x 512 layers): sums, coefficients, dy,
//          dgamma / dbeta - everything in the one workgroup;
//   k > 1  (2 .. 25: everything else; the stem and the 56x56 block outputs are the 25): the k workgroups of a slab - CONSECUTIVE linear ids, so
//          they are dispatched together - leave their per-channel and per-group sums in `part`, arrive on the slab's counter and
//          poll it (one lane, s_sleep between polls) until all k have; every workgroup then adds the k group sums in chunk
//          order (identical coefficients everywhere), chunk 0 also folds dgamma / dbeta.  The exchanged words travel as
//          device-scope write-through stores / cache-bypassing loads: no cache-wide fence.  A workgroup only ever waits for
//          workgroups with nearby ids of its own launch: in-order dispatch makes that wait finite whatever else shares the
//          chip.  A poll that lasts longer than ~0.2 s raises the error word and goes on (results of that launch are then
//          wrong, the queue is not blocked).
// Deterministic (fixed summation orders); agrees with the two-launch form to fp32 rounding (different orders).
#define OP_IT 8
#define OP_KMAX 32
// Process-wide count of hand-offs that TIMED OUT (a poll that lasted longer than ~0.2 s goes on with incomplete sums: the results of
// that launch are wrong).  The launch's own error word is lost with the next zero fill of the counters; this one stays, and the host
// reads it wherever it synchronises anyway (dyb_sync_error_count: the drivers' metric flush raises on a non-zero count).
__device__ unsigned g_gn_sync_errors = 0u;
// device-scope (write-through, cache-bypassing) accesses for the few words the workgroups of a slab exchange: global_store /
// global_load ... sc1.  With them the hand-off needs NO cache-wide fence - a release fence at device scope writes back, an acquire
// fence invalidates, the WHOLE L2 of the XCD, and ~900 workgroups doing both per launch made the first version of this kernel run
// at a third of its traffic's rate and slowed the convolutions on the other queue (profiles/r04_s1_*).
__device__ __forceinline__ void op_store_dev(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float op_load_dev(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
struct GnOnepass {
  const float* din;        // incoming gradient (first split-K slab)
  const float* addend;     // + residual-edge gradient, or NULL
  const float* out;        // saved activation (ReLU mask) or NULL: the mask is recomputed from y
  const float* y;
  const float* stats;      // [G][2] mean, rstd
  const float* gamma;
  const float* beta;
  float* dm;               // masked gradient out, or NULL
  float* dy;
  float* dgamma;
  float* dbeta;
  float* part;             // k > 1: [k][2][C] per-channel sums, then [k][G][2] group sums
  unsigned* ctr;           // k > 1: [G] arrival counters (zero before the launch) + [1] error word
  size_t slab_stride;
  int nslabs, HW, C, rows, relu;
  int poll_sleep;          // s_sleep(8) repetitions between two polls of the slab's counter
  int wt;                  // dy / dm leave write-through ("tp_gn_wt": as the convolutions' result tiles, igemm_tp.inc)
};
template <int OP_T>
__global__ __launch_bounds__(OP_T) void gn_bwd_onepass_kernel(GnOnepass a, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, a.din); DYB_RB(R, a.addend); DYB_RB(R, a.out); DYB_RB(R, a.y); DYB_RB(R, a.stats); DYB_RB(R, a.gamma); DYB_RB(R, a.beta);
  DYB_RB(R, a.dm); DYB_RB(R, a.dy); DYB_RB(R, a.dgamma); DYB_RB(R, a.dbeta); DYB_RB(R, a.part); DYB_RB(R, a.ctr);
  constexpr int NW = OP_T / 64;
  __shared__ float sm[NW][64][8];
  __shared__ float s_grp[2][2];
  __shared__ float s_c[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.x, k = gridDim.x, g = blockIdx.y;
  const int C = a.C, cqg = C >> 4;                        // float4 columns of a group: 4 .. 128, a power of two (host-checked)
  const int row0 = chunk * a.rows;
  const int row1 = row0 + a.rows < a.HW ? row0 + a.rows : a.HW;
  const int q = tid & (cqg - 1);                           // this work-item's column inside the group (OP_T % cqg == 0)
  const int c0 = (g * cqg + q) * 4;                        // its first channel
  const int rstep = OP_T / cqg;                            // rows between its consecutive items
  const int rfirst = row0 + tid / cqg;
  const float mean = a.stats[g * 2], rstd = a.stats[g * 2 + 1];
  const float4 ga = *reinterpret_cast<const float4*>(a.gamma + c0);
  const bool mask_y = a.relu && a.out == nullptr;
  const float4 be = mask_y ? *reinterpret_cast<const float4*>(a.beta + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  float4 d[OP_IT], xh[OP_IT];
  float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
  // phase 1, four items at a time: every load of the four (gradient, y, mask source, addend) is in flight before the first use
#pragma unroll
  for (int h = 0; h < OP_IT; h += 4) {
    size_t off[4];
    bool ok[4];
    float4 dv[4], yv[4], ov[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = rfirst + (h + j) * rstep;
      ok[j] = row < row1;
      off[j] = (size_t)(ok[j] ? row : row0) * C + c0;      // (a missing item re-reads the chunk's first row and is dropped)
      dv[j] = *reinterpret_cast<const float4*>(a.din + off[j]);
      yv[j] = *reinterpret_cast<const float4*>(a.y + off[j]);
      ov[j] = (a.relu && a.out) ? *reinterpret_cast<const float4*>(a.out + off[j]) : zero4;
    }
    if (a.addend) {
      float4 p[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] = *reinterpret_cast<const float4*>(a.addend + off[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) { dv[j].x += p[j].x; dv[j].y += p[j].y; dv[j].z += p[j].z; dv[j].w += p[j].w; }
    }
    for (int z = 1; z < a.nslabs; ++z) {                   // un-folded split-K slabs of the data gradient that produced din
      float4 p[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] = *reinterpret_cast<const float4*>(a.din + (size_t)z * a.slab_stride + off[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) { dv[j].x += p[j].x; dv[j].y += p[j].y; dv[j].z += p[j].z; dv[j].w += p[j].w; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 x;
      x.x = (yv[j].x - mean) * rstd; x.y = (yv[j].y - mean) * rstd; x.z = (yv[j].z - mean) * rstd; x.w = (yv[j].w - mean) * rstd;
      float4 o = ov[j];
      if (mask_y) {                                          // exactly the consumer's expression (igemm_conv.hip gnf_apply)
        o.x = fmaf(x.x, ga.x, be.x); o.y = fmaf(x.y, ga.y, be.y); o.z = fmaf(x.z, ga.z, be.z); o.w = fmaf(x.w, ga.w, be.w);
      }
      float4 v = dv[j];
      if (a.relu) {
        v.x = o.x > 0.f ? v.x : 0.f; v.y = o.y > 0.f ? v.y : 0.f; v.z = o.z > 0.f ? v.z : 0.f; v.w = o.w > 0.f ? v.w : 0.f;
      }
      if (!ok[j]) { v = zero4; x = zero4; }
      d[h + j] = v;
      xh[h + j] = x;
      sa[0] += v.x; sa[1] += v.y; sa[2] += v.z; sa[3] += v.w;
      sb[0] += v.x * x.x; sb[1] += v.y * x.y; sb[2] += v.z * x.z; sb[3] += v.w * x.w;
    }
  }
  // per-channel sums of the chunk: lanes of a wave that share a column first (cqg < 64), then the 16 waves through LDS
  for (int m = 32; m >= cqg; m >>= 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { sa[i] += __shfl_xor(sa[i], m); sb[i] += __shfl_xor(sb[i], m); }
  }
  const int W = cqg < 64 ? cqg : 64;                       // distinct columns per wave
  if (lane < W) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { sm[wave][lane][i] = sa[i]; sm[wave][lane][4 + i] = sb[i]; }
  }
  __syncthreads();
  float A[4] = {0.f, 0.f, 0.f, 0.f}, Bv[4] = {0.f, 0.f, 0.f, 0.f};
  float s1 = 0.f, s2 = 0.f;
  if (tid < cqg) {
    // column tid lives in lane (tid % 64) of the waves w with (w * 64) % cqg == tid - tid % 64
    const int l = tid & 63, wbase = tid >> 6, wstep = cqg > 64 ? cqg >> 6 : 1;
    for (int w = wbase; w < NW; w += wstep) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { A[i] += sm[w][l][i]; Bv[i] += sm[w][l][4 + i]; }
    }
    s1 = (ga.x * A[0] + ga.y * A[1]) + (ga.z * A[2] + ga.w * A[3]);      // (tid < cqg: this work-item's own column is column tid)
    s2 = (ga.x * Bv[0] + ga.y * Bv[1]) + (ga.z * Bv[2] + ga.w * Bv[3]);
    const int cc = (g * cqg + tid) * 4;
    if (k == 1) {
      *reinterpret_cast<float4*>(a.dbeta + cc) = make_float4(A[0], A[1], A[2], A[3]);
      *reinterpret_cast<float4*>(a.dgamma + cc) = make_float4(Bv[0], Bv[1], Bv[2], Bv[3]);
    } else {
      float* p = a.part + (size_t)chunk * 2 * C + cc;
#pragma unroll
      for (int i = 0; i < 4; ++i) { op_store_dev(p + i, A[i]); op_store_dev(p + C + i, Bv[i]); }
    }
  }
  // group sums of gamma * A, gamma * B: whole-wave butterflies (work-items without a column hold zeros), waves 0 and 1 carry the columns
  for (int m = 32; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
  if (lane == 0 && wave < 2) { s_grp[wave][0] = s1; s_grp[wave][1] = s2; }
  __syncthreads();
  const float inv_m = 1.0f / ((float)(C / G) * (float)a.HW);
  if (k == 1) {
    if (tid == 0) {
      s_c[0] = (s_grp[0][0] + s_grp[1][0]) * inv_m;          // (wave 1 holds zeros unless cqg = 128)
      s_c[1] = (s_grp[0][1] + s_grp[1][1]) * inv_m;
    }
  } else {
    float* gpart = a.part + (size_t)k * 2 * C;             // [k][G][2]
    if (tid == 0) {
      op_store_dev(gpart + (chunk * G + g) * 2 + 0, s_grp[0][0] + s_grp[1][0]);      // (wave 1 holds zeros unless cqg = 128)
      op_store_dev(gpart + (chunk * G + g) * 2 + 1, s_grp[0][1] + s_grp[1][1]);
    }
    // publish = the write-through stores above, drained (s_waitcnt vmcnt(0)) before this workgroup arrives on the slab's counter
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (tid == 0) {
      atomicAdd(a.ctr + g, 1u);
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(a.ctr + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)k) {
        for (int i = 0; i < a.poll_sleep; ++i) __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 20000000LL) {            // 100 MHz: 0.2 s
          atomicAdd(a.ctr + G, 1u);
          atomicAdd(&g_gn_sync_errors, 1u);
          break;
        }
      }
      float t1 = 0.f, t2 = 0.f;
      for (int c = 0; c < k; ++c) { t1 += op_load_dev(gpart + (c * G + g) * 2); t2 += op_load_dev(gpart + (c * G + g) * 2 + 1); }
      s_c[0] = t1 * inv_m;
      s_c[1] = t2 * inv_m;
    }
  }
  __syncthreads();
  const float c1 = s_c[0], c2 = s_c[1];
  // phase 2: dy (and dm) from the registers
#pragma unroll
  for (int j = 0; j < OP_IT; ++j) {
    const int row = rfirst + j * rstep;
    if (row < row1) {
      const size_t off = (size_t)row * C + c0;
      float4 r;
      r.x = rstd * (ga.x * d[j].x - c1 - xh[j].x * c2);
      r.y = rstd * (ga.y * d[j].y - c1 - xh[j].y * c2);
      r.z = rstd * (ga.z * d[j].z - c1 - xh[j].z * c2);
      r.w = rstd * (ga.w * d[j].w - c1 - xh[j].w * c2);
      if (a.wt) {
        const size_t tb = (size_t)a.HW * C * sizeof(float);
        tp_buf_store4<16>(tp_rsrc(a.dy, tb), (unsigned)(off * 4), r);
        if (a.dm) tp_buf_store4<16>(tp_rsrc(a.dm, tb), (unsigned)(off * 4), d[j]);
      } else {
        *reinterpret_cast<float4*>(a.dy + off) = r;
        if (a.dm) *reinterpret_cast<float4*>(a.dm + off) = d[j];
      }
    }
  }
  if (k > 1 && chunk == 0 && tid < cqg) {
    // dgamma / dbeta of this group's channels: the k chunks' per-channel sums in chunk order (every chunk has arrived)
    const int cc = (g * cqg + tid) * 4;
    float uA[4] = {0.f, 0.f, 0.f, 0.f}, uB[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < k; ++c) {
      const float* p = a.part + (size_t)c * 2 * C + cc;
#pragma unroll
      for (int i = 0; i < 4; ++i) { uA[i] += op_load_dev(p + i); uB[i] += op_load_dev(p + C + i); }
    }
    *reinterpret_cast<float4*>(a.dbeta + cc) = make_float4(uA[0], uA[1], uA[2], uA[3]);
    *reinterpret_cast<float4*>(a.dgamma + cc) = make_float4(uB[0], uB[1], uB[2], uB[3]);
  }
}
int dyb_hvp_sync_errors(unsigned* host_out, hipStream_t st);      // hvp_kernels.hip: the tangent kernels' own count
// Timed-out in-kernel hand-offs since the library was loaded (one-pass GroupNorm backward, one-launch GroupNorm tangents): copies the
// count to the host and WAITS for `stream` - call it where the host synchronises anyway.  Non-zero: some launch went on with
// incomplete sums (co-residency of a slab's workgroups not met within 0.2 s) and results since the last check are not to be trusted.
extern "C" int dyb_sync_error_count(unsigned* count_host, hipStream_t st) {
  DYB_REQUIRE(count_host, DYB_ERR_ARG);
  unsigned a = 0u, b = 0u;
  if (hipMemcpyFromSymbolAsync(&a, HIP_SYMBOL(g_gn_sync_errors), sizeof(unsigned), 0, hipMemcpyDeviceToHost, st) != hipSuccess) return DYB_ERR_LAUNCH;
  int rc = dyb_hvp_sync_errors(&b, st);
  if (rc != DYB_OK) return rc;
  if (hipStreamSynchronize(st) != hipSuccess) return DYB_ERR_LAUNCH;
  *count_host = a + b;
  return DYB_OK;
}
// zero fill of a small per-replica region (the arrival counters): replica-aware, unlike a memset node
__global__ void zero_words_kernel(unsigned* p, int n, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, p);
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}
int dyb_zero_words(unsigned* p, int n, hipStream_t st) {
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(zero_words_kernel, dim3(1, 1, R.n), dim3(256), 0, st, p, n, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// chunks per (image, group) slab of the one-pass backward, 0 = the shape does not qualify.  `cap` = float4 per workgroup: a
// workgroup of T work-items holds T * OP_IT; 0 = the policy's workgroup size ("tp_gn_threads", 1024: alone on the chip it is 1.3 -
// 1.5x faster than 256 on every layer shape - 3.0 - 4.9 TB/s, the wait costing 3 - 5 % - and needs no poll spacing; beside the
// weight-gradient queue the two measure the same, profiles/r04_gn_lab.txt) - the tests force several chunks on small shapes with
// small caps
static int onepass_threads(int cap) {
  if (cap <= 0) {
    const int t = dyb_tp_gn_threads();
    return t >= 1024 ? 1024 : t >= 512 ? 512 : 256;
  }
  return cap > 512 * OP_IT ? 1024 : cap > 256 * OP_IT ? 512 : 256;
}
int dyb_gn_onepass_chunks(int N, int HW, int C, int cap) {
  if (N != 1 || C % 16 != 0 || C < 64 || C > 2048 || !dyb_is_pow2(C)) return 0;
  const int T = onepass_threads(cap);
  if (cap <= 0 || cap > T * OP_IT) cap = T * OP_IT;
  const int cqg = C / 16;
  int k = 1;
  while (k <= OP_KMAX && dyb_cdiv(HW, k) * cqg > cap) ++k;
  if (k > OP_KMAX) return 0;
  return dyb_cdiv(HW, dyb_cdiv(HW, k));
}
// floats of `part` the k-chunk form needs (k > 1)
size_t dyb_gn_onepass_part_floats(int k, int C) { return k > 1 ? (size_t)k * 2 * C + (size_t)k * G * 2 : 0; }
int dyb_gn_bwd_onepass(const float* din, int nslabs, size_t slab_stride, const float* addend, const float* out, const float* y,
                       const float* stats, const float* gamma, const float* beta, float* dm, float* dy, float* dgamma, float* dbeta,
                       int HW, int C, int relu, int k, float* part, unsigned* ctr, hipStream_t st) {
  DYB_REQUIRE(din && y && stats && gamma && dy && dgamma && dbeta && nslabs >= 1 && k >= 1 && k <= OP_KMAX, DYB_ERR_ARG);
  DYB_REQUIRE(!relu || out || beta, DYB_ERR_ARG);
  DYB_REQUIRE(k == 1 || (part && ctr), DYB_ERR_ARG);
  const int rows = dyb_cdiv(HW, k);
  DYB_REQUIRE(dyb_is_pow2(C) && C >= 64 && C <= 2048 && rows * (C / 16) <= 1024 * OP_IT && dyb_cdiv(HW, rows) == k, DYB_ERR_UNSUPPORTED);
  GnOnepass a{din, addend, out, y, stats, gamma, beta, dm == din ? nullptr : dm, dy, dgamma, dbeta, part, ctr, slab_stride, nslabs, HW, C,
              rows, relu, dyb_tp_gn_poll() > 0 ? dyb_tp_gn_poll() : 1, dyb_tp_gn_wt()};
  const DybRep& R = dyb_rep_current();
  const int items = rows * (C / 16);
  if (items <= 256 * OP_IT) hipLaunchKernelGGL(gn_bwd_onepass_kernel<256>, dim3(k, G, R.n), dim3(256), 0, st, a, R);
  else if (items <= 512 * OP_IT) hipLaunchKernelGGL(gn_bwd_onepass_kernel<512>, dim3(k, G, R.n), dim3(512), 0, st, a, R);
  else hipLaunchKernelGGL(gn_bwd_onepass_kernel<1024>, dim3(k, G, R.n), dim3(1024), 0, st, a, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// Stand-alone form (tests, direct callers): one image, ws >= dyb_groupnorm_bwd_onepass_workspace_bytes; cap = float4 per
// workgroup (0: the kernel's 8192).  The arrival counters inside ws are zeroed here, on `st`.
extern "C" size_t dyb_groupnorm_bwd_onepass_workspace_bytes(int HW, int C) {
  return ((size_t)OP_KMAX * 2 * C + OP_KMAX * G * 2 + 64) * sizeof(float);
}
extern "C" int dyb_groupnorm_bwd_onepass(const float* dout_slabs, int nslabs, size_t slab_stride, const float* addend, const float* out,
                                         const float* y, const float* stats, const float* gamma, const float* beta, float* dm, float* dy,
                                         float* dgamma, float* dbeta, int HW, int C, int relu, int cap, void* ws, size_t ws_bytes,
                                         hipStream_t st) {
  const int k = dyb_gn_onepass_chunks(1, HW, C, cap);
  DYB_REQUIRE(k > 0, DYB_ERR_UNSUPPORTED);
  DYB_REQUIRE(ws && ws_bytes >= dyb_groupnorm_bwd_onepass_workspace_bytes(HW, C), DYB_ERR_WORKSPACE);
  float* part = reinterpret_cast<float*>(ws);
  unsigned* ctr = reinterpret_cast<unsigned*>(part + (size_t)OP_KMAX * 2 * C + OP_KMAX * G * 2);
  if (k > 1) {
    int rc = dyb_zero_words(ctr, G + 1, st);
    if (rc != DYB_OK) return rc;
  }
  return dyb_gn_bwd_onepass(dout_slabs, nslabs, slab_stride, addend, out, y, stats, gamma, beta, dm, dy, dgamma, dbeta, HW, C, relu, k, part,
                            ctr, st);
}

// Diagnostic (tools/gn_lab.py): the one-pass backward for `nrep` sequence replicas in ONE launch, the way the stepper issues it.
// blob: [nrep] x { din | y | out | dm | dy (HW*C floats each) | stats (8) | dgamma (C) | dbeta (C) | ws } with `blob_floats` floats per
// replica (>= 5*HW*C + 8 + 2*C + workspace floats); gamma / beta shared.  mode bits: 1 = mask from the saved activation, 2 = write dm.
extern "C" int dyb_debug_gn_onepass_replicas(float* blob, size_t blob_floats, int nrep, const float* gamma, const float* beta, int HW, int C,
                                             int relu, int mode, hipStream_t st) {
  DYB_REQUIRE(blob && gamma && beta && nrep >= 1 && nrep <= DYB_MAX_REPLICAS, DYB_ERR_ARG);
  const size_t n = (size_t)HW * C;
  const size_t need = 5 * n + 8 + 2 * (size_t)C + dyb_groupnorm_bwd_onepass_workspace_bytes(HW, C) / sizeof(float);
  DYB_REQUIRE(blob_floats >= need, DYB_ERR_WORKSPACE);
  DybRep Rp{};
  Rp.n = nrep;
  dyb_rep_identity(Rp);
  Rp.lo[0] = reinterpret_cast<const char*>(blob); Rp.span[0] = blob_floats * sizeof(float); Rp.stride[0] = blob_floats * sizeof(float);
  Rp.narenas = 1;
  DybRepScope scope(Rp);
  float *din = blob, *y = blob + n, *out = blob + 2 * n, *dm = blob + 3 * n, *dy = blob + 4 * n, *stats = blob + 5 * n;
  float *dgamma = stats + 8, *dbeta = dgamma + C, *ws = dbeta + C;
  const int k = dyb_gn_onepass_chunks(1, HW, C, 0);
  DYB_REQUIRE(k > 0, DYB_ERR_UNSUPPORTED);
  unsigned* ctr = reinterpret_cast<unsigned*>(ws + (size_t)OP_KMAX * 2 * C + OP_KMAX * G * 2);
  if (k > 1) {
    int rc = dyb_zero_words(ctr, G + 1, st);
    if (rc != DYB_OK) return rc;
  }
  return dyb_gn_bwd_onepass(din, 1, 0, nullptr, (relu && (mode & 1)) ? out : nullptr, y, stats, gamma, beta, (mode & 2) ? dm : nullptr, dy, dgamma,
                            dbeta, HW, C, relu, k, ws, ctr, st);
}

// ---- reduce-only form for consumers that form dy in their operand loaders (igemm_conv.hip) ----------
void dyb_gn_bwd_layout(int N, int HW, int C, int* nch, int* ncolb) {
  *nch = gn_chunks_bwd(HW, N, C);
  int CQ = C / 4;
  int TX = CQ < 256 ? CQ : 256;
  *ncolb = CQ / TX;
}
// Large enough for BOTH layouts a layer's partial block can have: the reduce kernel's (nch row chunks x ncolb column blocks
// per image) and the one a single-launch 1x1 data gradient leaves when it performs the producer's reduce in its epilogue
// (32-row tiles x 32-channel blocks per image, igemm_conv.hip) - at batch 16 the latter is the bigger one for 28x28 maps
// (the reduce layout shrinks its chunk count with the batch, the tile count does not).
extern "C" size_t dyb_groupnorm_bwd_partial_floats(int N, int HW, int C) {
  int nch, ncolb;
  dyb_gn_bwd_layout(N, HW, C, &nch, &ncolb);
  const size_t a = (size_t)N * nch * 2 * C + (size_t)N * nch * ncolb * G * 2;
  const int tch = dyb_cdiv(HW, 32), tcolb = dyb_cdiv(C, 32);
  const size_t b = (size_t)N * tch * 2 * C + (size_t)N * tch * tcolb * G * 2;
  return a > b ? a : b;
}
// dm = dout masked by the ReLU (relu == 0: dm may equal dout, nothing is written then); part receives
// the per-channel and per-group partial sums ([N*nch][2][C] | [N][nch*ncolb][G][2]).  out == NULL with
// relu: the mask is recomputed from y as fma(xhat, gamma, beta) > 0 (beta required).
extern "C" int dyb_groupnorm_bwd_reduce(const float* dout, const float* out, const float* y, const float* stats,
                                        const float* gamma, const float* beta, float* dm, float* part, int N, int HW, int C,
                                        int relu, hipStream_t st) {
  return dyb_gn_bwd_reduce_slabs(dout, 1, 0, nullptr, out, y, stats, gamma, beta, dm, part, N, HW, C, relu, st, nullptr);
}
// same, the incoming gradient being sum_z dout[z*slab_stride + .] (+ addend): the un-folded split-K slabs
// of the data-gradient convolution that produced it plus the residual-edge gradient (dm != dout required
// then); saves the stand-alone fold launch on the critical chain.
int dyb_gn_bwd_reduce_slabs(const float* dout, int nslabs, size_t slab_stride, const float* addend, const float* out,
                            const float* y, const float* stats, const float* gamma, const float* beta, float* dm,
                            float* part, int N, int HW, int C, int relu, hipStream_t st, hipEvent_t done) {
  DYB_REQUIRE(dout && y && stats && gamma && dm && part && nslabs >= 1, DYB_ERR_ARG);
  DYB_REQUIRE(!relu || out || beta, DYB_ERR_ARG);      // ReLU mask: from the saved activation, or recomputed from y
  DYB_REQUIRE(!(nslabs > 1 || addend) || dm != dout, DYB_ERR_ARG);
  DYB_REQUIRE(C % 16 == 0 && C <= 2048 && N > 0 && HW > 0, DYB_ERR_UNSUPPORTED);
  int nch, ncolb;
  dyb_gn_bwd_layout(N, HW, C, &nch, &ncolb);
  int CQ = C / 4;
  int TX = CQ < 256 ? CQ : 256;
  int rows = dyb_cdiv(HW, nch);
  float* gpart = part + (size_t)N * nch * 2 * C;
  // `done`: the event rides on the kernel's own completion signal - a separate hipEventRecord would put a
  // marker packet into the stream and ~6 us of bubble in front of the next kernel (measured)
  float* folded_none = nullptr;
  float* dm_arg = dm == dout ? (float*)nullptr : dm;
  const DybRep& R = dyb_rep_current();
  if (done)
    hipExtLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(ncolb, nch, N * R.n), dim3(256), 0, st, nullptr, done, 0, dout, nslabs,
                          slab_stride, addend, folded_none, dm_arg, out, y, stats, gamma, beta, part, gpart, HW, C, rows, relu,
                          TX, R);
  else
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(ncolb, nch, N * R.n), dim3(256), 0, st, dout, nslabs, slab_stride, addend,
                       folded_none, dm_arg, out, y, stats, gamma, beta, part, gpart, HW, C, rows, relu, TX, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
extern "C" int dyb_groupnorm_bwd(const float* dout, const float* out, const float* y, const float* stats,
                                 const float* gamma, float* dy, float* dres, float* dgamma, float* dbeta, int N, int HW,
                                 int C, int relu, void* ws, size_t ws_bytes, hipStream_t st) {
  return dyb_groupnorm_bwd_fold(dout, 1, 0, nullptr, nullptr, out, y, stats, gamma, dy, dres, dgamma, dbeta, N, HW, C, relu,
                                ws, ws_bytes, st);
}

// ------------------------------------------------------------------------------------------
// pooling and layout
// ------------------------------------------------------------------------------------------
// image [N][3][H][W] -> [N][H][W][4] (4th channel 0)
__global__ __launch_bounds__(256) void nchw3_to_nhwc4_kernel(DybRep R, const float* __restrict__ x, float* __restrict__ y, int N,
                                                             int HW) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, x); DYB_RB(R, y);
  size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    size_t n = i / HW, p = i % HW;
    const float* b = x + n * 3 * HW + p;
    *reinterpret_cast<float4*>(y + i * 4) = make_float4(b[0], b[HW], b[2 * (size_t)HW], 0.f);
  }
}
extern "C" int dyb_nchw3_to_nhwc4(const float* x, float* y, int N, int H, int W, hipStream_t st) {
  DYB_REQUIRE(x && y && N > 0, DYB_ERR_ARG);
  size_t total = (size_t)N * H * W;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(blocks, 1, R.n), dim3(256), 0, st, R, x, y, N, H * W);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// 3x3 stride 2 pad 1.  idx holds, per channel, the winning tap (0..8) in one byte.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(DybRep R, const float* __restrict__ x, float* __restrict__ y,
                                                          uint32_t* __restrict__ idx, int N, int H, int W, int C, int Ho,
                                                          int Wo) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, x); DYB_RB(R, y); DYB_RB(R, idx);
  const int CQ = C >> 2;
  size_t total = (size_t)N * Ho * Wo * CQ;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int cq = (int)(i % CQ);
    size_t t = i / CQ;
    int wo = (int)(t % Wo);
    t /= Wo;
    int ho = (int)(t % Ho);
    int n = (int)(t / Ho);
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uint32_t bi[4] = {255u, 255u, 255u, 255u};
    // the nine taps are fetched together (out-of-range taps re-read the clamped pixel and are skipped below), then
    // compared in the reference's scan order: a load inside the `continue` structure is one dependent round trip per tap
    float4 tapv[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        int hi = ho * 2 - 1 + r, wi = wo * 2 - 1 + s;
        hi = hi < 0 ? 0 : (hi >= H ? H - 1 : hi);
        wi = wi < 0 ? 0 : (wi >= W ? W - 1 : wi);
        tapv[r * 3 + s] = *reinterpret_cast<const float4*>(x + (((size_t)n * H + hi) * W + wi) * C + (size_t)cq * 4);
      }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * 2 - 1 + r;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wi = wo * 2 - 1 + s;
        if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
        const float4 v = tapv[r * 3 + s];
        const uint32_t tap = r * 3 + s;
        if (v.x > best.x || bi[0] == 255u) { best.x = v.x; bi[0] = tap; }
        if (v.y > best.y || bi[1] == 255u) { best.y = v.y; bi[1] = tap; }
        if (v.z > best.z || bi[2] == 255u) { best.z = v.z; bi[2] = tap; }
        if (v.w > best.w || bi[3] == 255u) { best.w = v.w; bi[3] = tap; }
      }
    }
    *reinterpret_cast<float4*>(y + i * 4) = best;
    idx[i] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
  }
}
// gather form: every input pixel looks at the (<= 4) windows that contain it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(DybRep R, const float* __restrict__ dy, const uint32_t* __restrict__ idx,
                                                          float* __restrict__ dx, int N, int H, int W, int C, int Ho,
                                                          int Wo) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, dy); DYB_RB(R, idx); DYB_RB(R, dx);
  const int CQ = C >> 2;
  size_t total = (size_t)N * H * W * CQ;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int cq = (int)(i % CQ);
    size_t t = i / CQ;
    int w = (int)(t % W);
    t /= W;
    int h = (int)(t % H);
    int n = (int)(t / H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ho = h >> 1; ho <= (h + 1) >> 1; ++ho) {
      if (ho >= Ho) continue;
      int r = h - (2 * ho - 1);
      for (int wo = w >> 1; wo <= (w + 1) >> 1; ++wo) {
        if (wo >= Wo) continue;
        int s = w - (2 * wo - 1);
        uint32_t tap = (uint32_t)(r * 3 + s);
        size_t o = (((size_t)n * Ho + ho) * Wo + wo) * CQ + cq;
        uint32_t k = idx[o];
        float4 d = *reinterpret_cast<const float4*>(dy + o * 4);
        if ((k & 255u) == tap) acc.x += d.x;
        if (((k >> 8) & 255u) == tap) acc.y += d.y;
        if (((k >> 16) & 255u) == tap) acc.z += d.z;
        if ((k >> 24) == tap) acc.w += d.w;
      }
    }
    *reinterpret_cast<float4*>(dx + i * 4) = acc;
  }
}
extern "C" int dyb_maxpool3x3s2_fwd(const float* x, float* y, uint32_t* idx, int N, int H, int W, int C,
                                    hipStream_t st) {
  DYB_REQUIRE(x && y && idx && C % 4 == 0, DYB_ERR_ARG);
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  size_t total = (size_t)N * Ho * Wo * (C / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks, 1, R.n), dim3(256), 0, st, R, x, y, idx, N, H, W, C, Ho, Wo);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
extern "C" int dyb_maxpool3x3s2_bwd(const float* dy, const uint32_t* idx, float* dx, int N, int H, int W, int C,
                                    hipStream_t st) {
  DYB_REQUIRE(dy && dx && idx && C % 4 == 0, DYB_ERR_ARG);
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  size_t total = (size_t)N * H * W * (C / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks, 1, R.n), dim3(256), 0, st, R, dy, idx, dx, N, H, W, C, Ho, Wo);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// global average over HW rows; result broadcast to `ndst` destinations (row stride ld floats):
// the pooled 2048-vector is the head of each of the three regressor inputs xc (model/hmr.py:162).
struct AvgDst {
  float* p[4];
};
// 4 lanes per (image, channel quad) split the pixels, each with up to 7 loads in flight per round trip
// (a 7x7 map: 2 trips); workgroups past the pooling ones copy `tail` (the regressor's initial state,
// reference model/hmr.py:157-159 torch.cat([xf, pred_pose, pred_shape, pred_cam], 1)) behind the
// pooled features of the first destination, which saves a separate copy launch.
struct AvgTail {
  const float* src;   // [N][src_ld] or NULL
  int src_ld, cols, dst_col;
};
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, AvgDst dst, int ndst, int ld, int N,
                                                          int HW, int C, AvgTail tail, int pool_blocks, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, x); DYB_RB(R, tail.src);
  if (dyb_rep)
    for (int d = 0; d < 4; ++d) dst.p[d] = dyb_rb(dst.p[d], R, dyb_rep);
  if ((int)blockIdx.x >= pool_blocks) {
    int i = (blockIdx.x - pool_blocks) * 256 + threadIdx.x;
    if (i < N * tail.cols) {
      int n = i / tail.cols, c = i % tail.cols;
      dst.p[0][(size_t)n * ld + tail.dst_col + c] = tail.src[(size_t)n * tail.src_ld + c];
    }
    return;
  }
  const int CQ = C >> 2;
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 2, part = threadIdx.x & 3;
  const bool live = i < N * CQ;
  const int n = live ? i / CQ : 0, cq = live ? i % CQ : 0;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    for (int p0 = part; p0 < HW; p0 += 28) {
      float4 t[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int p = p0 + 4 * j;
        t[j] = p < HW ? *reinterpret_cast<const float4*>(x + ((size_t)n * HW + p) * C + (size_t)cq * 4)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) { s.x += t[j].x; s.y += t[j].y; s.z += t[j].z; s.w += t[j].w; }
    }
  }
  s.x += __shfl_xor(s.x, 1); s.y += __shfl_xor(s.y, 1); s.z += __shfl_xor(s.z, 1); s.w += __shfl_xor(s.w, 1);
  s.x += __shfl_xor(s.x, 2); s.y += __shfl_xor(s.y, 2); s.z += __shfl_xor(s.z, 2); s.w += __shfl_xor(s.w, 2);
  if (live && part == 0) {
    float inv = 1.0f / (float)HW;
    s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
    for (int d = 0; d < ndst; ++d) *reinterpret_cast<float4*>(dst.p[d] + (size_t)n * ld + (size_t)cq * 4) = s;
  }
}
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dxf, int ld, float* __restrict__ dx,
                                                          int N, int HW, int C, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, dxf); DYB_RB(R, dx);
  const int CQ = C >> 2;
  size_t total = (size_t)N * HW * CQ;
  float inv = 1.0f / (float)HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int cq = (int)(i % CQ);
    int n = (int)(i / ((size_t)HW * CQ));
    float4 g = *reinterpret_cast<const float4*>(dxf + (size_t)n * ld + (size_t)cq * 4);
    g.x *= inv; g.y *= inv; g.z *= inv; g.w *= inv;
    *reinterpret_cast<float4*>(dx + i * 4) = g;
  }
}
int dyb_avgpool_fwd_tail(const float* x, float* const* dsts, int ndst, int ld, int N, int HW, int C, const float* tail,
                         int tail_ld, int tail_cols, int tail_dst_col, hipStream_t st) {
  DYB_REQUIRE(x && dsts && ndst >= 1 && ndst <= 4 && C % 4 == 0 && ld % 4 == 0, DYB_ERR_ARG);
  AvgDst d{};
  for (int i = 0; i < ndst; ++i) d.p[i] = dsts[i];
  int pool_blocks = dyb_cdiv(N * (C / 4) * 4, 256);
  AvgTail t{tail, tail_ld, tail ? tail_cols : 0, tail_dst_col};
  int tail_blocks = tail ? dyb_cdiv(N * tail_cols, 256) : 0;
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(pool_blocks + tail_blocks, 1, R.n), dim3(256), 0, st, x, d, ndst, ld, N, HW, C, t,
                     pool_blocks, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
extern "C" int dyb_avgpool_fwd(const float* x, float* const* dsts, int ndst, int ld, int N, int HW, int C,
                               hipStream_t st) {
  return dyb_avgpool_fwd_tail(x, dsts, ndst, ld, N, HW, C, nullptr, 0, 0, 0, st);
}
extern "C" int dyb_avgpool_bwd(const float* dxf, int ld, float* dx, int N, int HW, int C, hipStream_t st) {
  DYB_REQUIRE(dxf && dx && C % 4 == 0 && ld % 4 == 0, DYB_ERR_ARG);
  size_t total = (size_t)N * HW * (C / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(blocks, 1, R.n), dim3(256), 0, st, dxf, ld, dx, N, HW, C, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
