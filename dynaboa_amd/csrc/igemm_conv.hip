// NHWC implicit-GEMM convolution on the fp32 matrix cores of gfx950:
//   forward, data-gradient and weight-gradient as three gather flavours of one tiled GEMM
//   built on v_mfma_f32_32x32x2_f32 (exact fp32, same numerics as an fmaf chain).
//
// Replaces, for the ResNet-50(GN) backbone of reference model/hmr.py:40-60,138-153, what the
// reference gets from cuDNN through nn.Conv2d (all 53 convs are bias-free) and its autograd.
//
// Layouts (all fp32):
//   activations  x[N][H][W][C]          (C = Cin, multiple of 4; the 3-channel image is padded to 4)
//   weights      w[R][S][C][K]          (K = Cout)  == GEMM "B" matrix [R*S*C][K] row-major
//   outputs      y[N][Ho][Wo][K]
// GEMM views:
//   fwd   : Y[m=(n,ho,wo)][k]        = sum_{(r,s,c)}  X(m;(r,s,c)) * W[(r,s,c)][k]
//   dgrad : dX[m=(n,h,w)][c]         = sum_{(r,s,k)}  dY(m;(r,s,k)) * W[(r,s,c)][k]
//   wgrad : dW[i=(r,s,c)][k]         = sum_{p=(n,ho,wo)} X(p;i) * dY[p][k]
//
// Tiling: 256 threads = 4 waves per workgroup, block tile 64x64, K-step 32, each wave owns one
// 32x32 accumulator (16 VGPRs).  Operands are staged through LDS k-major ([32][64+4]) so the
// MFMA fragment read (lane -> row lane&31, k = lane>>5) is a conflict-free ds_read_b32 and the
// "contiguous along the tile row" sources are written with one ds_write_b128.  Global loads are
// 16 B per lane, four per thread per K-step (two per operand), register-prefetched one K-step
// ahead of the 16 MFMAs of the current step (two LDS buffers, one barrier per step; a second
// register set - two steps in flight - measured no gain and cost half the occupancy).  At batch 1
// most layers have only 1..50 output tiles, so the K loop is split over blockIdx.z into fp32 slabs
// that a second kernel (or the GroupNorm statistics kernel) folds - deterministic, no atomics.
#include <hip/hip_ext.h>
#include <stdlib.h>

#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>
#include <map>
#include <string>
#include <stdio.h>

#include "dyb_common.h"

#define BM 64
#define BN 64
#ifndef BK
#define BK 32
#endif
#define NH (BK / 16)      // 16-byte pieces per operand per thread per K-step
#define LDS_LD (64 + 4)

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };

struct IgemmArgs {
  const float* A;        // fwd: x      dgrad: dy     wgrad: x
  const float* B;        // fwd: w      dgrad: w      wgrad: dy
  float* out;            // result, or slab base when nsplit > 1
  const float* addend;   // optional, only honoured when nsplit == 1 (out = acc + addend)
  float out_scale;       // throughput form, nsplit == 1 with an addend: out = addend + out_scale * acc (1: the plain sum).  A weight gradient
                         // with addend = the layer's current weights and out_scale = -fastlr writes the MAML fast weights p - fastlr * g straight
                         // from the accumulators ("fuse_fast", DybWgradUpdateScope): the gradient never travels to HBM and back
  int N, H, W, C;        // input activation geometry
  int K;                 // Cout
  int R, S, stride, pad;
  int Ho, Wo;
  int logC, logK;
  int M, Ncols, Kdim;    // GEMM sizes for this mode
  int ktiles, tiles_per_split, nsplit;
  int xcd;               // throughput form: XCD-contiguous workgroup order
  int wt;                // throughput form: cache policy of the result stores ("tp_wt": 0 plain, 1 sc1 write-through, 2 nt, 3 sc0 sc1)
  int cls_tile0[4];      // throughput data gradient: first tile (grid x) of each phase class (stride 2)
  int compact;           // throughput data gradient of a 1x1 stride-2 conv: only the one class that has a tap is computed, rows written
                         // COMPACT (class-local index) into slabs of slab_rows rows; a scatter fold places them (run_igemm_tp)
  int slab_rows;         // rows of one split-K slab (M unless compact)
  unsigned long long* probe;   // throughput form: per-wave phase clocks (dyb_conv_probe_set), normally NULL
  const float* A2;       // operand PAIR (latency form, no fused loaders): out = A (x) B + A2 (x) B2 as ONE K loop - K-tiles [0, ktiles1)
  const float* B2;       // read (A, B), K-tiles [ktiles1, ktiles) read (A2, B2) at tile index - ktiles1.  The tangent passes of the exact
  int ktiles1;           // Hessian-vector product are made of such pairs (hvp_engine.inc); ktiles1 == ktiles: a plain conv
  float* adam_m;         // throughput weight gradient, nsplit == 1, addend = out = the layer's weights ("fuse_adam"): Adam applied from the
  float* adam_v;         // accumulators - the moments (same offsets as out) and the weights updated in place, adam_sc = (step_size, bc2_sqrt)
  const float* adam_sc;  // of THIS replica (a per-replica arena); NULL: no Adam
  float adam_b1, adam_b2, adam_eps;
  float* fold_out;       // in-kernel split-K fold (nsplit > 1, a counter region in scope): the workgroup that arrives LAST on a tile's counter adds
  const float* fold_addend;  // the tile's nsplit slabs in split order (+ fold_addend) into fold_out and, forward, leaves the tile's GroupNorm
  unsigned* fold_ctr;    // statistics in gn_part; NULL: slabs are left for a fold launch / the consumer.  fold_ctr: one word per tile, zero
  float* gn_part;        // throughput forward, one image, nsplit == 1 (or folded in-kernel): the GroupNorm statistics of the OUTPUT leave with the tile - one
                         // [G][2] (sum, sum of squares) record per wave tile, [(lx * WM + wm) * ntiles_n * WN + ly * WN + wn] - instead of
                         // a statistics launch re-reading y (igemm_tp.inc epilogue)
};

// ---- tile loaders: each returns the 16 bytes this thread contributes to the K-step tile ----

// rows = output pixels, k = (r,s,c) with c fastest                       (fwd A)
struct FwdARow {
  int base_n, hi0, wi0, n;
  bool valid;
};
__device__ __forceinline__ FwdARow fwd_a_row(const IgemmArgs& g, int m) {
  FwdARow r;
  r.valid = m < g.M;
  int mm = r.valid ? m : 0;
  int wo = mm % g.Wo;
  int t = mm / g.Wo;
  int ho = t % g.Ho;
  int n = t / g.Ho;
  r.base_n = n * g.H * g.W;
  r.n = n;
  r.hi0 = ho * g.stride - g.pad;
  r.wi0 = wo * g.stride - g.pad;
  return r;
}
__device__ __forceinline__ float4 fwd_a_load(const IgemmArgs& g, const float* __restrict__ A, const FwdARow& row, int k) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row.valid && k < g.Kdim) {
    int rs = k >> g.logC;
    int c = k & (g.C - 1);
    int r = rs / g.S;
    int s = rs - r * g.S;
    int hi = row.hi0 + r, wi = row.wi0 + s;
    if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W)
      v = *reinterpret_cast<const float4*>(A + (((size_t)(row.base_n + hi * g.W + wi)) << g.logC) + c);
  }
  return v;
}

// float offset of that piece inside x[N][H][W][C], or -1 for a padding tap / out-of-range element
__device__ __forceinline__ long fwd_a_off(const IgemmArgs& g, const FwdARow& row, int k) {
  if (row.valid && k < g.Kdim) {
    int rs = k >> g.logC;
    int c = k & (g.C - 1);
    int r = rs / g.S;
    int s = rs - r * g.S;
    int hi = row.hi0 + r, wi = row.wi0 + s;
    if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) return (long)((((size_t)(row.base_n + hi * g.W + wi)) << g.logC) + c);
  }
  return -1;
}

// rows = input pixels, k = (r,s,ko) with ko fastest                      (dgrad A)
struct DgARow {
  int n, h, w;
  bool valid;
};
__device__ __forceinline__ DgARow dg_a_row(const IgemmArgs& g, int m) {
  DgARow r;
  r.valid = m < g.M;
  int mm = r.valid ? m : 0;
  r.w = mm % g.W;
  int t = mm / g.W;
  r.h = t % g.H;
  r.n = t / g.H;
  return r;
}
__device__ __forceinline__ float4 dg_a_load(const IgemmArgs& g, const float* __restrict__ A, const DgARow& row, int kk) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row.valid && kk < g.Kdim) {
    int rs = kk >> g.logK;
    int ko = kk & (g.K - 1);
    int r = rs / g.S;
    int s = rs - r * g.S;
    int th = row.h + g.pad - r, tw = row.w + g.pad - s;
    int sm = g.stride - 1;                       // stride is 1 or 2
    if (th >= 0 && tw >= 0 && (th & sm) == 0 && (tw & sm) == 0) {
      int ho = th >> (g.stride >> 1), wo = tw >> (g.stride >> 1);
      if (ho < g.Ho && wo < g.Wo)
        v = *reinterpret_cast<const float4*>(A + (((size_t)((row.n * g.Ho + ho) * g.Wo + wo)) << g.logK) + ko);
    }
  }
  return v;
}
// float offset of that 16-byte piece inside a [N][Ho][Wo][K] tensor, or -1 for a structural zero
__device__ __forceinline__ long dg_a_off(const IgemmArgs& g, const DgARow& row, int kk) {
  if (row.valid && kk < g.Kdim) {
    int rs = kk >> g.logK;
    int ko = kk & (g.K - 1);
    int r = rs / g.S;
    int s = rs - r * g.S;
    int th = row.h + g.pad - r, tw = row.w + g.pad - s;
    int sm = g.stride - 1;
    if (th >= 0 && tw >= 0 && (th & sm) == 0 && (tw & sm) == 0) {
      int ho = th >> (g.stride >> 1), wo = tw >> (g.stride >> 1);
      if (ho < g.Ho && wo < g.Wo) return (long)((((size_t)((row.n * g.Ho + ho) * g.Wo + wo)) << g.logK) + ko);
    }
  }
  return -1;
}
// rows = cin, k = (r,s,ko) with ko fastest: W[(rs*C + c)*K + ko]         (dgrad B, transposing)
__device__ __forceinline__ float4 dg_b_load(const IgemmArgs& g, const float* __restrict__ Bw, int c, int kk) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < g.Ncols && kk < g.Kdim) {
    int rs = kk >> g.logK;
    int ko = kk & (g.K - 1);
    v = *reinterpret_cast<const float4*>(Bw + ((((size_t)rs << g.logC) + c) << g.logK) + ko);
  }
  return v;
}
// direct [k][n] row-major matrix with leading dimension ld               (fwd B, wgrad B)
__device__ __forceinline__ float4 direct_load(const float* base, int ld, int rows, int cols, int k, int n) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < rows && n < cols) v = *reinterpret_cast<const float4*>(base + (size_t)k * ld + n);
  return v;
}
// rows i = (r,s,c) (4 consecutive c per thread), column p = pixel        (wgrad A, direct)
struct WgARow {
  int r, s, c;
  bool valid;
};
__device__ __forceinline__ WgARow wg_a_row(const IgemmArgs& g, int i) {
  WgARow w;
  w.valid = i < g.M;
  int ii = w.valid ? i : 0;
  int rs = ii >> g.logC;
  w.c = ii & (g.C - 1);
  w.r = rs / g.S;
  w.s = rs - w.r * g.S;
  return w;
}
__device__ __forceinline__ float4 wg_a_load(const IgemmArgs& g, const float* __restrict__ A, const WgARow& row, int p) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row.valid && p < g.Kdim) {
    int wo = p % g.Wo;
    int t = p / g.Wo;
    int ho = t % g.Ho;
    int n = t / g.Ho;
    int hi = ho * g.stride - g.pad + row.r, wi = wo * g.stride - g.pad + row.s;
    if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W)
      v = *reinterpret_cast<const float4*>(A + (((size_t)((n * g.H + hi) * g.W + wi)) << g.logC) + row.c);
  }
  return v;
}

// ---- GroupNorm backward applied in the operand loader --------------------------------------------
// The gradient w.r.t. a conv output y is  dy = rstd*(gamma*dm - c1 - xhat*c2)  with dm the (ReLU-
// masked) gradient of the GroupNorm output, xhat = (y-mean)*rstd and c1, c2 two per-(image, group)
// means.  dy is only ever consumed by the data-gradient conv (as its A operand) and the weight-
// gradient conv (as its B operand), so instead of materialising it in a kernel of its own (one more
// dependent launch on the critical chain per layer) both form it on the fly from (dm, y): the
// coefficients come from the per-chunk partial sums that gn_bwd_reduce left behind, folded in each
// workgroup's prologue while its first operand loads are in flight.  The weight-gradient launch also
// folds the per-channel partials into dgamma / dbeta (it runs off the critical path).
struct GnBwdFuse {
  const float* y;          // [N][HW][K]  conv output that was normalised
  const float* stats;      // [N][G][2]   mean, rstd
  const float* gpart;      // [N][kparts][G][2]  per-chunk sums of gamma*dm, gamma*dm*xhat
  const float* gamma;      // [K]
  const float* partials;   // [N*nchunks][2][K]  per-channel sums (wgrad launch: dgamma / dbeta)
  float* dgamma;
  float* dbeta;
  int kparts, nchunks, HW;
  float inv_m;
};
struct Frag {
  float4 d, v, ga;
  bool ok;
};
__device__ __forceinline__ float4 gnb_apply(const Frag& f, const float* cf) {
  const float mean = cf[0], rstd = cf[1], c1 = cf[2], c2 = cf[3];
  float4 r;
  r.x = rstd * (f.ga.x * f.d.x - c1 - ((f.v.x - mean) * rstd) * c2);
  r.y = rstd * (f.ga.y * f.d.y - c1 - ((f.v.y - mean) * rstd) * c2);
  r.z = rstd * (f.ga.z * f.d.z - c1 - ((f.v.z - mean) * rstd) * c2);
  r.w = rstd * (f.ga.w * f.d.w - c1 - ((f.v.w - mean) * rstd) * c2);
  return r;
}

__device__ __forceinline__ long wg_a_off(const IgemmArgs& g, const WgARow& row, int p) {
  if (row.valid && p < g.Kdim) {
    int wo = p % g.Wo;
    int t = p / g.Wo;
    int ho = t % g.Ho;
    int n = t / g.Ho;
    int hi = ho * g.stride - g.pad + row.r, wi = wo * g.stride - g.pad + row.s;
    if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) return (long)((((size_t)((n * g.H + hi) * g.W + wi)) << g.logC) + row.c);
  }
  return -1;
}

// ---- GroupNorm(+ReLU) of the PRODUCER applied in the operand loader ------------------------------
// Inside a bottleneck the normalised activation relu(gn(y_prev)) has exactly one consumer, the next
// conv (forward: its A operand; backward: the A operand of that conv's weight gradient), so it is
// never written: the conv reads the producer's raw output y_prev and normalises on the fly.  Forward
// launches fold the producer's per-chunk (sum, sum of squares) partials into (mean, rstd) in their
// prologue (in double, as gn_apply does) and workgroup (0,0,0) saves them for backward; the
// weight-gradient launch reads the saved ones.
struct GnFwdFuse {
  const float* partials;   // [N][nchunks][G][2]; NULL: use stats_in
  const float* stats_in;   // [N][G][2]
  const float* gamma;
  const float* beta;
  float* stats_out;        // [N][G][2] or NULL
  int nchunks, HW;         // HW: pixels per image of the producer's output
  float eps;
  int relu;
};
// replica rebasing of the argument blocks (dyb_common.h: sequence replicas in the grid)
__device__ __forceinline__ void rebase(IgemmArgs& g, const DybRep& R, int rep) {
  g.A = dyb_rb(g.A, R, rep); g.B = dyb_rb(g.B, R, rep); g.out = dyb_rb(g.out, R, rep); g.addend = dyb_rb(g.addend, R, rep);
  g.gn_part = dyb_rb(g.gn_part, R, rep); g.A2 = dyb_rb(g.A2, R, rep); g.B2 = dyb_rb(g.B2, R, rep);
  g.fold_out = dyb_rb(g.fold_out, R, rep); g.fold_addend = dyb_rb(g.fold_addend, R, rep);     // (fold_ctr is shared: indexed by the launch's replica slot)
  g.adam_m = dyb_rb(g.adam_m, R, rep); g.adam_v = dyb_rb(g.adam_v, R, rep); g.adam_sc = dyb_rb(g.adam_sc, R, rep);
}
__device__ __forceinline__ void rebase(GnBwdFuse& f, const DybRep& R, int rep) {
  f.y = dyb_rb(f.y, R, rep); f.stats = dyb_rb(f.stats, R, rep); f.gpart = dyb_rb(f.gpart, R, rep); f.gamma = dyb_rb(f.gamma, R, rep);
  f.partials = dyb_rb(f.partials, R, rep); f.dgamma = dyb_rb(f.dgamma, R, rep); f.dbeta = dyb_rb(f.dbeta, R, rep);
}
__device__ __forceinline__ void rebase(GnFwdFuse& nf, const DybRep& R, int rep) {
  nf.partials = dyb_rb(nf.partials, R, rep); nf.stats_in = dyb_rb(nf.stats_in, R, rep); nf.gamma = dyb_rb(nf.gamma, R, rep);
  nf.beta = dyb_rb(nf.beta, R, rep); nf.stats_out = dyb_rb(nf.stats_out, R, rep);
}
__device__ __forceinline__ float4 gnf_apply(float4 y, float4 ga, float4 be, float mean, float rstd, int relu) {
  float4 o;
  o.x = fmaf((y.x - mean) * rstd, ga.x, be.x);
  o.y = fmaf((y.y - mean) * rstd, ga.y, be.y);
  o.z = fmaf((y.z - mean) * rstd, ga.z, be.z);
  o.w = fmaf((y.w - mean) * rstd, ga.w, be.w);
  if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
  return o;
}
__device__ __forceinline__ double igemm_shfl_xor_f64(double v, int m) { return __shfl_xor(v, m); }

// (mean, rstd, c1, c2) per (image, group) into s_coef for the GroupNorm-backward loaders: the per-chunk group sums the
// reduce left behind are folded here (eight loads in flight per lane, clamped + masked).  Ends with a barrier.
__device__ __forceinline__ void gnb_prologue(const GnBwdFuse& f, int N, int tid, float* s_raw, float* s_coef) {
  const int nvals = N * DYB_GN_GROUPS * 2;
  int L = 32;
  while (L > 1 && L * nvals > 256) L >>= 1;
  const int per_pass = 256 / L;
  for (int base = 0; base < nvals; base += per_pass) {
    const int v = base + tid / L, sub = tid % L;
    float s = 0.f;
    if (v < nvals) {
      const float* gp = f.gpart + (size_t)(v >> 3) * f.kparts * 8 + (v & 7);
      for (int k0 = sub; k0 < f.kparts; k0 += 8 * L) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = k0 + j * L;
          t[j] = gp[(size_t)(k < f.kparts ? k : f.kparts - 1) * 8];        // clamped, masked below (see gnf_prologue)
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (k0 + j * L < f.kparts) ? t[j] : 0.f;
        s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
      }
    }
    for (int m = L >> 1; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (v < nvals && sub == 0) s_raw[v] = s * f.inv_m;
  }
  __syncthreads();
  for (int i = tid; i < N * DYB_GN_GROUPS; i += 256) {
    s_coef[i * 4 + 0] = f.stats[i * 2];
    s_coef[i * 4 + 1] = f.stats[i * 2 + 1];
    s_coef[i * 4 + 2] = s_raw[i * 2];
    s_coef[i * 4 + 3] = s_raw[i * 2 + 1];
  }
  __syncthreads();
}

// (mean, rstd) of the producer per (image, group) into s_nrm, from its per-chunk partials (folded in double, eight
// loads in flight per lane: this sits on the consumer's critical path) or from the saved statistics; `first`
// workgroup also saves them for backward.  Ends with a barrier.
__device__ __forceinline__ void gnf_prologue(const GnFwdFuse& nf, int N, int C, int tid, bool first, double* s_rawd,
                                             float* s_nrm) {
  const int nvals = N * DYB_GN_GROUPS * 2;
  if (nf.partials) {
    int L = 32;
    while (L > 1 && L * nvals > 256) L >>= 1;
    const int per_pass = 256 / L;
    for (int base = 0; base < nvals; base += per_pass) {
      const int v = base + tid / L, sub = tid % L;
      double s = 0.0;
      if (v < nvals) {
        const float* pp = nf.partials + (size_t)(v >> 3) * nf.nchunks * 8 + (v & 7);
        for (int k0 = sub; k0 < nf.nchunks; k0 += 8 * L) {
          // unconditional loads from clamped indices first, masks afterwards: a per-lane `cond ? load : 0` is waited
          // for load by load (K4 kernel's note) and this fold is on the consumer's critical path
          float t[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = k0 + j * L;
            t[j] = pp[(size_t)(k < nf.nchunks ? k : nf.nchunks - 1) * 8];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = (k0 + j * L < nf.nchunks) ? t[j] : 0.f;
          s += (((double)t[0] + (double)t[1]) + ((double)t[2] + (double)t[3])) +
               (((double)t[4] + (double)t[5]) + ((double)t[6] + (double)t[7]));
        }
      }
      for (int m = L >> 1; m >= 1; m >>= 1) s += igemm_shfl_xor_f64(s, m);
      if (v < nvals && sub == 0) s_rawd[v] = s;
    }
    __syncthreads();
    for (int i = tid; i < N * DYB_GN_GROUPS; i += 256) {
      const double cnt = (double)nf.HW * (double)(C / DYB_GN_GROUPS);
      const double mean = s_rawd[i * 2] / cnt;
      double var = s_rawd[i * 2 + 1] / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      const float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)nf.eps));
      s_nrm[i * 2] = m;
      s_nrm[i * 2 + 1] = r;
      if (nf.stats_out && first) {
        nf.stats_out[i * 2] = m;
        nf.stats_out[i * 2 + 1] = r;
      }
    }
  } else {
    for (int i = tid; i < nvals; i += 256) s_nrm[i] = nf.stats_in[i];
  }
  __syncthreads();
}

// BF: the bf16 matrix-core variant (BASELINE configs[4]).  Everything up to the staging store is the same fp32 code - operands
// come from fp32 HBM tensors (master weights, fp32 activations), the GroupNorm arithmetic of the fused loaders and the
// accumulators stay fp32 - but the operand tiles are rounded to bf16 (nearest-even) when they are staged and the product runs
// on v_mfma_f32_32x32x16_bf16 (16x the fp32 rate).  LDS tiles are then row-major [row][k] (k contiguous, +8 pad): a fragment
// (lane -> row lane&31, eight consecutive k at 8*(lane>>5)) is one 16-byte read, a "4 consecutive k of one row" source one
// 8-byte store.  Never the parity default: selected per plan (dyb_hmr_set_bf16).
#define BFK (BK + 8)
template <int MODE, bool GB, bool FA, bool BF = false>
__global__ __launch_bounds__(256) void igemm_mfma_kernel(IgemmArgs g, GnBwdFuse f, GnFwdFuse nf, DybRep R) {
  DYB_REP_PROLOGUE(R);
  if (dyb_rep) {
    rebase(g, R, dyb_rep);
    if constexpr (GB) rebase(f, R, dyb_rep);
    if constexpr (FA) rebase(nf, R, dyb_rep);
  }
  __shared__ __attribute__((aligned(16))) float As[BF ? 1 : 2][BF ? 1 : BK][BF ? 4 : LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[BF ? 1 : 2][BF ? 1 : BK][BF ? 4 : LDS_LD];
  __shared__ __attribute__((aligned(16))) unsigned short Ah[BF ? 2 : 1][BF ? BM : 1][BF ? BFK : 8];
  __shared__ __attribute__((aligned(16))) unsigned short Bh[BF ? 2 : 1][BF ? BN : 1][BF ? BFK : 8];
  __shared__ float s_gbh[BF ? 4 * 64 * 2 : 4];
  // staging stores: `k4` = a thread's 4 consecutive k of tile row `row`; `col4` = one k, 4 consecutive tile rows / columns
  auto put_k4 = [&](bool isA, int buf, int k, int row, float4 v) {
    if constexpr (BF) {
      uint2 pk;
      pk.x = dyb_f2bf(v.x) | (dyb_f2bf(v.y) << 16);
      pk.y = dyb_f2bf(v.z) | (dyb_f2bf(v.w) << 16);
      *reinterpret_cast<uint2*>(isA ? &Ah[buf][row][k] : &Bh[buf][row][k]) = pk;
    } else {
      float(*T)[LDS_LD] = isA ? As[buf] : Bs[buf];
      T[k + 0][row] = v.x; T[k + 1][row] = v.y; T[k + 2][row] = v.z; T[k + 3][row] = v.w;
    }
  };
  auto put_col4 = [&](bool isA, int buf, int k, int col0, float4 v) {
    if constexpr (BF) {
      unsigned short(*T)[BFK] = isA ? Ah[buf] : Bh[buf];
      T[col0 + 0][k] = (unsigned short)dyb_f2bf(v.x); T[col0 + 1][k] = (unsigned short)dyb_f2bf(v.y);
      T[col0 + 2][k] = (unsigned short)dyb_f2bf(v.z); T[col0 + 3][k] = (unsigned short)dyb_f2bf(v.w);
    } else {
      *reinterpret_cast<float4*>(isA ? &As[buf][k][col0] : &Bs[buf][k][col0]) = v;
    }
  };
  __shared__ float s_coef[GB ? 64 * DYB_GN_GROUPS * 4 : 4];   // [n][g] -> mean, rstd, c1, c2
  __shared__ float s_raw[GB ? 64 * DYB_GN_GROUPS * 2 : 4];
  __shared__ float s_nrm[FA ? 64 * DYB_GN_GROUPS * 2 : 4];     // [n][g] -> mean, rstd of the producer
  __shared__ double s_rawd[FA ? 64 * DYB_GN_GROUPS * 2 : 1];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int kt_begin = (int)dyb_bz * g.tiles_per_split;
  int kt_end = kt_begin + g.tiles_per_split;
  if (kt_end > g.ktiles) kt_end = g.ktiles;

  // Each thread moves TWO 16-byte pieces per operand per K-step (h = 0, 1):
  // "transposing" mapping: one tile row, 4 consecutive k at k-offset t_kq + 16h      (t_row, t_kq)
  // "direct" mapping     : k-offset d_k + 16h, 4 consecutive tile columns            (d_k, d_q)
  const int t_row = tid >> 2, t_kq = (tid & 3) * 4;
  const int d_k = tid >> 4, d_q = (tid & 15) * 4;

  FwdARow fa;
  DgARow da;
  WgARow wa;
  if constexpr (MODE == MODE_FWD) fa = fwd_a_row(g, m0 + t_row);
  if constexpr (MODE == MODE_DGRAD) da = dg_a_row(g, m0 + t_row);
  if constexpr (MODE == MODE_WGRAD) wa = wg_a_row(g, m0 + d_q);
  const int logKg = g.logK - 2;                                 // log2(channels per group)
  const int logCg = g.logC - 2;
  float4 wa_gamma = make_float4(0.f, 0.f, 0.f, 0.f), wa_beta = wa_gamma;
  if constexpr (FA && MODE == MODE_WGRAD) {
    if (wa.valid) {
      wa_gamma = *reinterpret_cast<const float4*>(nf.gamma + wa.c);
      wa_beta = *reinterpret_cast<const float4*>(nf.beta + wa.c);
    }
  }
  float4 wg_gamma = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (GB && MODE == MODE_WGRAD) {
    if (n0 + d_q < g.Ncols) wg_gamma = *reinterpret_cast<const float4*>(f.gamma + n0 + d_q);
  }

  // the second operand pair of a paired launch (uniform per K-step; the fused loaders never see one: run_igemm)
  auto load_a = [&](int kt, int h, Frag& o) {
    const bool second = kt >= g.ktiles1;
    const float* __restrict__ Ap = second ? g.A2 : g.A;
    if constexpr (!(GB || FA)) kt = second ? kt - g.ktiles1 : kt;
    if constexpr (MODE == MODE_FWD) {
      if constexpr (FA) {
        const int k = kt * BK + 16 * h + t_kq;
        // branch-free (a padding tap reads element 0 and is zeroed when staged): a load under `if (ok)` is waited for
        // on the spot, which serialises the cold-L2 round trips of a K-step (see the K4 kernel's note)
        long off = fwd_a_off(g, fa, k);
        o.ok = off >= 0;
        o.d = *reinterpret_cast<const float4*>(g.A + (o.ok ? off : 0));
        o.v = *reinterpret_cast<const float4*>(nf.gamma + (k & (g.C - 1)));
        o.ga = *reinterpret_cast<const float4*>(nf.beta + (k & (g.C - 1)));
      } else {
        o.d = fwd_a_load(g, Ap, fa, kt * BK + 16 * h + t_kq);
      }
    } else if constexpr (MODE == MODE_DGRAD) {
      if constexpr (GB) {
        long off = dg_a_off(g, da, kt * BK + 16 * h + t_kq);
        o.ok = off >= 0;
        off = o.ok ? off : 0;
        o.d = *reinterpret_cast<const float4*>(g.A + off);
        o.v = *reinterpret_cast<const float4*>(f.y + off);
        o.ga = *reinterpret_cast<const float4*>(f.gamma + ((kt * BK + 16 * h + t_kq) & (g.K - 1)));
      } else {
        o.d = dg_a_load(g, Ap, da, kt * BK + 16 * h + t_kq);
      }
    } else {
      if constexpr (FA) {
        long off = wg_a_off(g, wa, kt * BK + 16 * h + d_k);
        o.ok = off >= 0;
        o.d = *reinterpret_cast<const float4*>(g.A + (o.ok ? off : 0));
      } else {
        o.d = wg_a_load(g, Ap, wa, kt * BK + 16 * h + d_k);
      }
    }
  };
  auto load_b = [&](int kt, int h, Frag& o) {
    const bool second = kt >= g.ktiles1;
    const float* __restrict__ Bp = second ? g.B2 : g.B;
    if constexpr (!(GB || FA)) kt = second ? kt - g.ktiles1 : kt;
    if constexpr (MODE == MODE_FWD) o.d = direct_load(Bp, g.K, g.Kdim, g.Ncols, kt * BK + 16 * h + d_k, n0 + d_q);
    else if constexpr (MODE == MODE_DGRAD) o.d = dg_b_load(g, Bp, n0 + t_row, kt * BK + 16 * h + t_kq);
    else {
      if constexpr (GB) {
        const int p = kt * BK + 16 * h + d_k, col = n0 + d_q;
        o.ok = p < g.Kdim && col < g.Ncols;
        const size_t off = o.ok ? (size_t)p * g.K + col : 0;
        o.d = *reinterpret_cast<const float4*>(g.B + off);
        o.v = *reinterpret_cast<const float4*>(f.y + off);
      } else {
        o.d = direct_load(Bp, g.K, g.Kdim, g.Ncols, kt * BK + 16 * h + d_k, n0 + d_q);
      }
    }
  };
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto store_a = [&](int buf, int kt, int h, const Frag& o) {
    if constexpr (MODE == MODE_WGRAD) {
      float4 v = o.d;
      if constexpr (FA) {
        const int p = kt * BK + 16 * h + d_k;
        const int n = g.N > 1 ? p / (g.Ho * g.Wo) : 0;
        const float* st = &s_nrm[(n * DYB_GN_GROUPS + (wa.c >> logCg)) * 2];
        v = o.ok ? gnf_apply(o.d, wa_gamma, wa_beta, st[0], st[1], nf.relu) : zero4;
      }
      put_col4(true, buf, 16 * h + d_k, d_q, v);
    } else {
      float4 v = o.d;
      if constexpr (FA && MODE == MODE_FWD) {
        const int c = (kt * BK + 16 * h + t_kq) & (g.C - 1);
        const float* st = &s_nrm[(fa.n * DYB_GN_GROUPS + (c >> logCg)) * 2];
        v = o.ok ? gnf_apply(o.d, o.v, o.ga, st[0], st[1], nf.relu) : zero4;
      }
      if constexpr (GB && MODE == MODE_DGRAD) {
        const int ko = (kt * BK + 16 * h + t_kq) & (g.K - 1);
        v = o.ok ? gnb_apply(o, &s_coef[(da.n * DYB_GN_GROUPS + (ko >> logKg)) * 4]) : zero4;
      }
      put_k4(true, buf, 16 * h + t_kq, t_row, v);
    }
  };
  auto store_b = [&](int buf, int kt, int h, const Frag& o) {
    if constexpr (MODE == MODE_DGRAD) {
      put_k4(false, buf, 16 * h + t_kq, t_row, o.d);
    } else {
      float4 v = o.d;
      if constexpr (GB && MODE == MODE_WGRAD) {
        const int p = kt * BK + 16 * h + d_k;
        const int n = g.N > 1 ? p / f.HW : 0;
        Frag t = o;
        t.ga = wg_gamma;
        v = o.ok ? gnb_apply(t, &s_coef[(n * DYB_GN_GROUPS + ((n0 + d_q) >> logKg)) * 4]) : zero4;
      }
      put_col4(false, buf, 16 * h + d_k, d_q, v);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  Frag ra[NH], rb[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) { ra[h].d = ra[h].v = ra[h].ga = zero4; ra[h].ok = false; rb[h] = ra[h]; }
  if (kt_begin < kt_end) {
#pragma unroll
    for (int h = 0; h < NH; ++h) load_a(kt_begin, h, ra[h]);
#pragma unroll
    for (int h = 0; h < NH; ++h) load_b(kt_begin, h, rb[h]);
  }
  if constexpr (GB) gnb_prologue(f, g.N, tid, s_raw, s_coef);

  if constexpr (FA) gnf_prologue(nf, g.N, g.C, tid, blockIdx.x == 0 && blockIdx.y == 0 && dyb_bz == 0, s_rawd, s_nrm);

  if (kt_begin < kt_end) {
    // register prefetch one K-step ahead, two LDS buffers, one barrier per step
#pragma unroll
    for (int h = 0; h < NH; ++h) { store_a(0, kt_begin, h, ra[h]); store_b(0, kt_begin, h, rb[h]); }
    __syncthreads();
    const int arow = wm * 32 + (lane & 31), bcol = wn * 32 + (lane & 31), khalf = lane >> 5;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int buf = (kt - kt_begin) & 1;
      const bool more = kt + 1 < kt_end;
      if (more) {
#pragma unroll
        for (int h = 0; h < NH; ++h) load_a(kt + 1, h, ra[h]);
#pragma unroll
        for (int h = 0; h < NH; ++h) load_b(kt + 1, h, rb[h]);
      }
      // keep the three phases apart: nothing that consumes the loads (staging stores, the zero-select of padding taps,
      // the GroupNorm arithmetic of the fused loaders) may be scheduled above the MFMAs - its wait would expose the
      // whole memory latency instead of overlapping it with the matrix work
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BF) {
#pragma unroll
        for (int kb = 0; kb < BK; kb += 16) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Ah[buf][arow][kb + 8 * khalf]);
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bh[buf][bcol][kb + 8 * khalf]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int k2 = 0; k2 < BK; k2 += 2) {
          float a = As[buf][k2 + khalf][arow];
          float b = Bs[buf][k2 + khalf][bcol];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
#pragma unroll
        for (int h = 0; h < NH; ++h) { store_a(buf ^ 1, kt + 1, h, ra[h]); store_b(buf ^ 1, kt + 1, h, rb[h]); }
      }
      __syncthreads();
    }
  }

  // C/D fragment of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* outp = g.out + (g.nsplit > 1 ? (size_t)dyb_bz * g.M * g.Ncols : 0);
  if constexpr (BF) {
  const int col = n0 + wn * 32 + (lane & 31);
  if (g.nsplit == 1 && g.addend) {
    // the 16 addend values are fetched together (clamped addresses), not one conditional load + wait per row
    float add[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const bool ok = row < g.M && col < g.Ncols;
      add[r] = g.addend[ok ? (size_t)row * g.Ncols + col : 0];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += add[r];
  }
  if (col < g.Ncols) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < g.M) outp[(size_t)row * g.Ncols + col] = acc[r];
    }
  }
  } else {
    // The 64 x 64 tile goes through LDS (the operand stages are free: the K loop ended on a barrier) so that a work-item holds four
    // 16-byte row pieces instead of sixteen scalars of one column: 16-byte stores, and a layout in which
    //   * a split launch with a counter region in scope folds in-kernel ("lat_fold"): every workgroup stores its partial tile
    //     write-through into its split's slab and arrives on the tile's counter; the LAST to arrive adds the nsplit slabs in split
    //     order (the result does not depend on who was last), adds the addend and writes the result - no fold launch;
    //   * a forward launch of ONE image leaves the tile's GroupNorm statistics (one [G][2] record per workgroup tile) - from the
    //     accumulators when K is unsplit, from the folded tile otherwise - instead of a statistics launch re-reading y.
    float(*T)[LDS_LD] = reinterpret_cast<float(*)[LDS_LD]>(&As[0][0][0]);
    __shared__ float s_st[4][4][2];
    __shared__ int s_last;
#pragma unroll
    for (int r = 0; r < 16; ++r) T[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][wn * 32 + (lane & 31)] = acc[r];
    __syncthreads();
    const int er = tid >> 4, ec = (tid & 15) * 4;
    const int col4 = n0 + ec;
    float4 v[4];
    unsigned vo[4];                                      // byte offset of the piece in the result matrix, TP_OOB: past the end
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = m0 + er + 16 * j;
      v[j] = *reinterpret_cast<const float4*>(&T[er + 16 * j][ec]);
      vo[j] = (row < g.M && col4 < g.Ncols) ? (unsigned)(((size_t)row * g.Ncols + col4) * 4) : TP_OOB;
    }
    const size_t mat_bytes = (size_t)g.M * g.Ncols * sizeof(float);
    const bool fold = g.fold_out != nullptr;
    bool finish = true;                                  // this workgroup holds the finished tile in v[]
    const float* addp = g.addend;
    float* dst = outp;
    if (fold) {
      const __amdgpu_buffer_rsrc_t rsP = tp_rsrc(outp, mat_bytes);
#pragma unroll
      for (int j = 0; j < 4; ++j) tp_buf_store4<16>(rsP, vo[j], v[j]);
      __builtin_amdgcn_s_waitcnt(0x0F70);                // the write-through stores have left before this workgroup arrives
      __syncthreads();
      if (tid == 0) {
        unsigned* c = g.fold_ctr + (((size_t)dyb_lrep * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y);
        const unsigned old = atomicAdd(c, 1u);
        const bool last = old + 1u == (unsigned)g.nsplit;
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last ? 1 : 0;
      }
      __syncthreads();
      finish = s_last != 0;
      if (finish) {
        float4 sum[4] = {zero4, zero4, zero4, zero4};
        for (int z0 = 0; z0 < g.nsplit; z0 += 4) {       // four slabs' pieces in flight per trip; added in split order
          float4 tz[4][4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int z = z0 + q;
            const __amdgpu_buffer_rsrc_t rsZ = tp_rsrc(g.out + (size_t)(z < g.nsplit ? z : 0) * g.M * g.Ncols, z < g.nsplit ? mat_bytes : 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) tz[q][j] = tp_buf_load4_dev(rsZ, vo[j]);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) { sum[j].x += tz[q][j].x; sum[j].y += tz[q][j].y; sum[j].z += tz[q][j].z; sum[j].w += tz[q][j].w; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sum[j];
        addp = g.fold_addend;
        dst = g.fold_out;
      }
    } else if (g.nsplit > 1) {
      addp = nullptr;
    }
    if (finish) {
      if (addp) {
        float4 a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = tp_mask4(*reinterpret_cast<const float4*>(addp + (vo[j] != TP_OOB ? vo[j] / 4 : 0)), vo[j] != TP_OOB);
        // (pieces past the end stay exact zeros: the GroupNorm statistics below sum all four pieces - ADVICE r5)
#pragma unroll
        for (int j = 0; j < 4; ++j) {                    // addend + out_scale * tile (out_scale 1: the plain sum, bit for bit)
          v[j].x = fmaf(g.out_scale, v[j].x, a[j].x); v[j].y = fmaf(g.out_scale, v[j].y, a[j].y);
          v[j].z = fmaf(g.out_scale, v[j].z, a[j].z); v[j].w = fmaf(g.out_scale, v[j].w, a[j].w);
        }
      }
      if constexpr (MODE == MODE_FWD) {
        if (g.gn_part != nullptr && (g.nsplit == 1 || fold)) {
          // rows / columns past the end hold exact zeros.  A work-item's pieces lie in one group (4 adjacent columns, groups are >= 16
          // wide); lanes l, l^16, l^32 hold other rows of the same columns; qpg = 16-byte column pieces of the tile per group
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s1 += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            s2 += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
          }
          s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
          const int cpg = g.K >> 2;                        // columns per group (>= 16, a power of two: host-checked)
          const int qpg = (cpg < 64 ? cpg : 64) >> 2;      // 4, 8 or 16
          for (int m = 1; m < qpg; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
          if (lane < 16 && (lane & (qpg - 1)) == 0) { s_st[wave][lane / qpg][0] = s1; s_st[wave][lane / qpg][1] = s2; }
          __syncthreads();
          if (tid < DYB_GN_GROUPS) {
            const int gi = tid, c0 = gi * cpg;             // group gi covers columns [c0, c0 + cpg)
            const int lo = c0 > n0 ? c0 : n0, hi = (c0 + cpg) < (n0 + 64) ? (c0 + cpg) : (n0 + 64);
            float t1 = 0.f, t2 = 0.f;
            if (lo < hi) {
              const int seg = (lo - n0) / (4 * qpg);
              t1 = (s_st[0][seg][0] + s_st[1][seg][0]) + (s_st[2][seg][0] + s_st[3][seg][0]);
              t2 = (s_st[0][seg][1] + s_st[1][seg][1]) + (s_st[2][seg][1] + s_st[3][seg][1]);
            }
            float* rec = g.gn_part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (DYB_GN_GROUPS * 2);
            rec[gi * 2] = t1;
            rec[gi * 2 + 1] = t2;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (vo[j] != TP_OOB) *reinterpret_cast<float4*>(dst + vo[j] / 4) = v[j];
    }
  }

  if constexpr (GB && MODE == MODE_WGRAD) {
    // dgamma / dbeta of the 64 channels of this column block: 4 lanes per channel split the
    // (image, chunk) range of the per-channel partials and meet in LDS (reusing the A stage)
    if (blockIdx.x == 0 && dyb_bz == 0) {
      __syncthreads();
      float(*s_gb)[64][2] = reinterpret_cast<float(*)[64][2]>(BF ? &s_gbh[0] : &As[0][0][0]);
      const int cl = tid & 63, part = tid >> 6;
      const int c = n0 + cl, C = g.K;
      float A = 0.f, B = 0.f;
      if (c < C) {
        // eight (A, B) pairs in flight per lane and trip: with ~100 chunks the fold was 14 dependent trips, the
        // longest thing in the launch
        const int total = g.N * f.nchunks;
        for (int k0 = part; k0 < total; k0 += 32) {
          float a8[8], b8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = k0 + 4 * j;
            const float* p0 = f.partials + (size_t)(k < total ? k : total - 1) * 2 * C + c;
            a8[j] = p0[0];
            b8[j] = p0[C];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (k0 + 4 * j >= total) { a8[j] = 0.f; b8[j] = 0.f; }
          A += ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
          B += ((b8[0] + b8[1]) + (b8[2] + b8[3])) + ((b8[4] + b8[5]) + (b8[6] + b8[7]));
        }
      }
      s_gb[part][cl][0] = A;
      s_gb[part][cl][1] = B;
      __syncthreads();
      if (part == 0 && c < C) {
        f.dbeta[c] = (s_gb[0][cl][0] + s_gb[1][cl][0]) + (s_gb[2][cl][0] + s_gb[3][cl][0]);
        f.dgamma[c] = (s_gb[0][cl][1] + s_gb[1][cl][1]) + (s_gb[2][cl][1] + s_gb[3][cl][1]);
      }
    }
  }
}

// ---- 1x1 forward conv with the GroupNorm statistics in its epilogue ("K4") -------------------------
// The small 1x1 layers (layers 2-4: M = Ho*Wo <= 784 pixels at batch 1, Cin <= 512) are where a forward
// conv spends its time in launch + the statistics kernel that follows it, not in arithmetic.  Here the
// four waves of a workgroup split the K range of ONE 32x32 output tile (K-step 128 = 32 per wave), sum
// their accumulators through LDS, and - because no split-K slabs are left to fold - finish the layer in
// the same launch: y tile out, plus this tile's (sum, sum of squares) as one GroupNorm partial (a tile's
// 32 columns lie inside one of the 4 channel groups when Cout/4 >= 32).  The separate gn_stats launch
// disappears; consumers fold gridDim.x*gridDim.y partials exactly as they fold gn_stats'.
#define K4_BK 128
#define K4_LD 36
struct K4Args {
  const float* x;        // [H][W][C] input (batch 1), or the producer's raw output when FA
  const float* w;        // [C][K]
  float* y;              // [Ho*Wo][K]
  float* partials;       // [mtiles*ntiles][G][2]
  int H, W, C, K, stride, Ho, Wo, M;      // M = Ho*Wo: pixels of ONE image
  int N, tpi;                             // batch, 32-row tiles per image (MULTI: tiles never straddle images)
};
// MULTI (batch > 1, experimental behind DYB_K4_BATCH=1): blockIdx.x = image * tpi + tile; partials are per image.
template <bool FA, bool MULTI>
__global__ __launch_bounds__(256) void igemm_k4_fwd_kernel(K4Args g, GnFwdFuse nf, DybRep R) {
  DYB_REP_PROLOGUE(R);
  if (dyb_rep) {
    g.x = dyb_rb(g.x, R, dyb_rep); g.w = dyb_rb(g.w, R, dyb_rep); g.y = dyb_rb(g.y, R, dyb_rep); g.partials = dyb_rb(g.partials, R, dyb_rep);
    if constexpr (FA) rebase(nf, R, dyb_rep);
  }
  __shared__ __attribute__((aligned(16))) float As[2][K4_BK][K4_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][K4_BK][K4_LD];
  __shared__ float s_nrm[FA ? (MULTI ? 64 : 1) * DYB_GN_GROUPS * 2 : 4];
  __shared__ double s_rawd[FA ? (MULTI ? 64 : 1) * DYB_GN_GROUPS * 2 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = MULTI ? blockIdx.x / g.tpi : 0, mtile = MULTI ? blockIdx.x - img * g.tpi : blockIdx.x;
  if constexpr (MULTI) {
    g.x += (size_t)img * g.H * g.W * g.C;
    g.y += (size_t)img * g.M * g.K;
  }
  const int m0 = mtile * 32, n0 = blockIdx.y * 32;
  const int ktiles = (g.C + K4_BK - 1) / K4_BK;
  // A pieces: tile row a_row, 4 consecutive k at 32h + a_kq (h = 0..3 -> consumed by wave h); B pieces: k row 32h + b_k, 4 columns
  const int a_row = tid >> 3, a_kq = (tid & 7) * 4;
  const int b_k = tid >> 3, b_q = (tid & 7) * 4;
  // No bounds logic in the loaders: Cin % 128 == 0 and Cout % 32 == 0 keep every piece in range, and rows past the end
  // of a ragged last tile re-read row M-1 (their results are never stored or counted).  Any per-piece condition - a
  // branch, or even a select on the loaded value - makes the compiler wait for that load on the spot, which serialises
  // the eight cold-L2 round trips of a K-step (seen in the ISA as one s_waitcnt vmcnt per load).
  const int m = (m0 + a_row < g.M) ? m0 + a_row : g.M - 1;
  const int ho = m / g.Wo, wo = m - (m / g.Wo) * g.Wo;
  const float* arow = g.x + ((size_t)(ho * g.stride) * g.W + (size_t)wo * g.stride) * g.C;
  const int logCg = dyb_ilog2_dev(g.C) - 2;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  (void)zero4;

  auto load_a = [&](int kt, int h, Frag& o) {
    const int k = kt * K4_BK + 32 * h + a_kq;
    o.d = *reinterpret_cast<const float4*>(arow + k);
    if constexpr (FA) {
      o.v = *reinterpret_cast<const float4*>(nf.gamma + k);
      o.ga = *reinterpret_cast<const float4*>(nf.beta + k);
    }
  };
  auto load_b = [&](int kt, int h, Frag& o) {
    o.d = *reinterpret_cast<const float4*>(g.w + (size_t)(kt * K4_BK + 32 * h + b_k) * g.K + n0 + b_q);
  };
  auto store = [&](int buf, int kt, int h, const Frag& a, const Frag& b) {
    float4 v = a.d;
    if constexpr (FA) {
      const float* st = &s_nrm[(img * DYB_GN_GROUPS + ((kt * K4_BK + 32 * h + a_kq) >> logCg)) * 2];
      v = gnf_apply(a.d, a.v, a.ga, st[0], st[1], nf.relu);
    }
    As[buf][32 * h + a_kq + 0][a_row] = v.x;
    As[buf][32 * h + a_kq + 1][a_row] = v.y;
    As[buf][32 * h + a_kq + 2][a_row] = v.z;
    As[buf][32 * h + a_kq + 3][a_row] = v.w;
    *reinterpret_cast<float4*>(&Bs[buf][32 * h + b_k][b_q]) = b.d;
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  Frag ra[4], rb[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) { ra[h].d = ra[h].v = ra[h].ga = zero4; ra[h].ok = false; rb[h] = ra[h]; }
#pragma unroll
  for (int h = 0; h < 4; ++h) { load_a(0, h, ra[h]); load_b(0, h, rb[h]); }
  if constexpr (FA) gnf_prologue(nf, MULTI ? g.N : 1, g.C, tid, blockIdx.x == 0 && blockIdx.y == 0, s_rawd, s_nrm);
#pragma unroll
  for (int h = 0; h < 4; ++h) store(0, 0, h, ra[h], rb[h]);
  __syncthreads();
  const int khalf = lane >> 5, l31 = lane & 31;
  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < ktiles;
    if (more) {
#pragma unroll
      for (int h = 0; h < 4; ++h) { load_a(kt + 1, h, ra[h]); load_b(kt + 1, h, rb[h]); }
    }
    // scheduling barriers: the staging stores below target the OTHER LDS buffer, so nothing but these keeps the compiler
    // from hoisting them (and the wait for the loads they consume) above the MFMAs - which would expose the full
    // memory latency every step instead of overlapping it with the matrix work
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k2 = 0; k2 < 32; k2 += 2) {
      float a = As[buf][32 * wave + k2 + khalf][l31];
      float b = Bs[buf][32 * wave + k2 + khalf][l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
#pragma unroll
      for (int h = 0; h < 4; ++h) store(buf ^ 1, kt + 1, h, ra[h], rb[h]);
    }
    __syncthreads();
  }
  // ---- sum the four waves' accumulators: red[wave][r][lane] (16 KB, in the A stage) -> fin[32][33] (in the B stage)
  float* red = &As[0][0][0];
  float* fin = &Bs[0][0][0];
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int s = tid + 256 * j, r = s >> 6, ln = s & 63;
    const float v = (red[(0 * 16 + r) * 64 + ln] + red[(1 * 16 + r) * 64 + ln]) + (red[(2 * 16 + r) * 64 + ln] + red[(3 * 16 + r) * 64 + ln]);
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), col = ln & 31;        // C/D fragment of the 32x32 MFMA
    fin[row * 33 + col] = v;
  }
  __syncthreads();
  {
    const int row = tid >> 3, cq = (tid & 7) * 4;
    if (m0 + row < g.M && n0 + cq < g.K) {
      float4 v = make_float4(fin[row * 33 + cq], fin[row * 33 + cq + 1], fin[row * 33 + cq + 2], fin[row * 33 + cq + 3]);
      *reinterpret_cast<float4*>(g.y + (size_t)(m0 + row) * g.K + n0 + cq) = v;
    }
  }
  // ---- GroupNorm partial of this tile: 8 row-parts x 32 columns, then the first wave folds
  {
    const int col = tid & 31, part = tid >> 5;
    float s1 = 0.f, s2 = 0.f;
    if (n0 + col < g.K) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = part * 4 + i;
        if (m0 + row < g.M) { const float v = fin[row * 33 + col]; s1 += v; s2 += v * v; }
      }
    }
    float* st = red;                         // [8][32][2], the accumulator stage is free again
    __syncthreads();
    st[(part * 32 + col) * 2] = s1;
    st[(part * 32 + col) * 2 + 1] = s2;
    __syncthreads();
    if (tid < 64) {
      const int c = tid & 31, which = tid >> 5;
      float t = 0.f;
#pragma unroll
      for (int p8 = 0; p8 < 8; ++p8) t += st[(p8 * 32 + c) * 2 + which];
      t += __shfl_xor(t, 16); t += __shfl_xor(t, 8); t += __shfl_xor(t, 4); t += __shfl_xor(t, 2); t += __shfl_xor(t, 1);
      if (c < DYB_GN_GROUPS) {
        const int grp = n0 / (g.K / DYB_GN_GROUPS);
        // record index: [image][tile * ntiles + ntile]; blockIdx.x = image * tpi + tile, so the flat form covers both cases
        g.partials[(((size_t)blockIdx.x * gridDim.y + blockIdx.y) * DYB_GN_GROUPS + c) * 2 + which] = (c == grp) ? t : 0.f;
      }
    }
  }
}

// ---- 1x1 data gradient with the NEXT GroupNorm-backward reduce in its epilogue ("K4 dgrad") --------------
// Mirror image of the K4 forward kernel on the backward chain (reduce -> dgrad per layer): for a small 1x1 conv L the
// four waves split the K (= Cout) range of one 32-pixel x 32-channel dx tile, dy being formed on the fly from (dm, y)
// of L as in the tiled kernel.  The finished tile IS the gradient of the producer P's GroupNorm output (P = the layer
// whose normalised output feeds L), so the epilogue does P's reduce right there: add the residual-edge gradient, apply
// P's ReLU mask (from the stored activation, or recomputed from y_P where it was never stored), write dm_P, and leave
// this tile's per-channel (sum dm, sum dm*xhat) and per-group gamma-weighted sums as one row chunk / column block of
// P's partial block.  P's own gn_bwd_reduce launch disappears from the chain.
struct K4DgradArgs {
  // conv L
  const float* dm;        // [M][K]   masked GroupNorm-output gradient of L
  const float* w;         // [C][K]
  const float* addend;    // [M][C] or NULL: residual-edge gradient added to dx
  // producer P (GroupNorm over [M][C])
  const float* y_p;       // [M][C] raw conv output of P
  const float* out_p;     // [M][C] stored activation (mask source) or NULL: recompute from y_p
  const float* stats_p;   // [G][2]
  const float* gamma_p;
  const float* beta_p;
  float* dm_p;            // [M][C]
  float* partials_p;      // [mtiles][2][C]
  float* gpart_p;         // [mtiles*ntiles][G][2]
  int M, C, K;            // M = pixels of ONE image
  int N, tpi;             // batch, 32-row tiles per image
};
// MULTI (batch > 1, experimental behind DYB_K4_BATCH=1): blockIdx.x = image * tpi + tile, per-image coefficients / statistics
template <bool MULTI>
__global__ __launch_bounds__(256) void igemm_k4_dgrad_kernel(K4DgradArgs g, GnBwdFuse f, DybRep R) {
  DYB_REP_PROLOGUE(R);
  if (dyb_rep) {
    g.dm = dyb_rb(g.dm, R, dyb_rep); g.w = dyb_rb(g.w, R, dyb_rep); g.addend = dyb_rb(g.addend, R, dyb_rep);
    g.y_p = dyb_rb(g.y_p, R, dyb_rep); g.out_p = dyb_rb(g.out_p, R, dyb_rep); g.stats_p = dyb_rb(g.stats_p, R, dyb_rep);
    g.gamma_p = dyb_rb(g.gamma_p, R, dyb_rep); g.beta_p = dyb_rb(g.beta_p, R, dyb_rep); g.dm_p = dyb_rb(g.dm_p, R, dyb_rep);
    g.partials_p = dyb_rb(g.partials_p, R, dyb_rep); g.gpart_p = dyb_rb(g.gpart_p, R, dyb_rep);
    rebase(f, R, dyb_rep);
  }
  __shared__ __attribute__((aligned(16))) float As[2][K4_BK][K4_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][K4_BK][K4_LD];
  __shared__ float s_coef[(MULTI ? 64 : 1) * DYB_GN_GROUPS * 4], s_raw[(MULTI ? 64 : 1) * DYB_GN_GROUPS * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = MULTI ? blockIdx.x / g.tpi : 0, mtile = MULTI ? blockIdx.x - img * g.tpi : blockIdx.x;
  if constexpr (MULTI) {
    const size_t ok = (size_t)img * g.M * g.K, oc = (size_t)img * g.M * g.C;
    g.dm += ok; f.y += ok;
    g.y_p += oc; g.dm_p += oc;
    if (g.addend) g.addend += oc;
    if (g.out_p) g.out_p += oc;
    g.stats_p += img * DYB_GN_GROUPS * 2;
  }
  const int m0 = mtile * 32, n0 = blockIdx.y * 32;
  const int ktiles = g.K / K4_BK;
  const int a_row = tid >> 3, a_kq = (tid & 7) * 4;
  const int m = (m0 + a_row < g.M) ? m0 + a_row : g.M - 1;              // ragged last tile re-reads the last pixel
  const float* dmrow = g.dm + (size_t)m * g.K;
  const float* yrow = f.y + (size_t)m * g.K;
  const float* wrow = g.w + (size_t)(n0 + a_row) * g.K;                  // B piece: channel n0 + a_row, 4 consecutive k
  const int logKg = dyb_ilog2_dev(g.K) - 2;

  auto load_a = [&](int kt, int h, Frag& o) {
    const int k = kt * K4_BK + 32 * h + a_kq;
    o.d = *reinterpret_cast<const float4*>(dmrow + k);
    o.v = *reinterpret_cast<const float4*>(yrow + k);
    o.ga = *reinterpret_cast<const float4*>(f.gamma + k);
  };
  auto load_b = [&](int kt, int h, Frag& o) { o.d = *reinterpret_cast<const float4*>(wrow + kt * K4_BK + 32 * h + a_kq); };
  auto store = [&](int buf, int kt, int h, const Frag& a, const Frag& b) {
    const int k = kt * K4_BK + 32 * h + a_kq;
    Frag t = a;
    t.ok = true;
    const float4 v = gnb_apply(t, &s_coef[(img * DYB_GN_GROUPS + (k >> logKg)) * 4]);
    As[buf][32 * h + a_kq + 0][a_row] = v.x;
    As[buf][32 * h + a_kq + 1][a_row] = v.y;
    As[buf][32 * h + a_kq + 2][a_row] = v.z;
    As[buf][32 * h + a_kq + 3][a_row] = v.w;
    Bs[buf][32 * h + a_kq + 0][a_row] = b.d.x;
    Bs[buf][32 * h + a_kq + 1][a_row] = b.d.y;
    Bs[buf][32 * h + a_kq + 2][a_row] = b.d.z;
    Bs[buf][32 * h + a_kq + 3][a_row] = b.d.w;
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  Frag ra[4], rb[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) { load_a(0, h, ra[h]); load_b(0, h, rb[h]); }
  // epilogue operands that do not depend on the matrix product go out now as well
  const int e_row = tid >> 3, e_cq = (tid & 7) * 4;
  const bool e_ok = m0 + e_row < g.M;
  const size_t e_off = (size_t)(e_ok ? m0 + e_row : g.M - 1) * g.C + n0 + e_cq;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 e_y = *reinterpret_cast<const float4*>(g.y_p + e_off);
  const float4 e_add = g.addend ? *reinterpret_cast<const float4*>(g.addend + e_off) : zero4;
  const float4 e_out = g.out_p ? *reinterpret_cast<const float4*>(g.out_p + e_off) : zero4;
  const float4 e_ga = *reinterpret_cast<const float4*>(g.gamma_p + n0 + e_cq);
  const float4 e_be = *reinterpret_cast<const float4*>(g.beta_p + n0 + e_cq);
  const int grp_p = n0 / (g.C / DYB_GN_GROUPS);
  const float mean_p = g.stats_p[grp_p * 2], rstd_p = g.stats_p[grp_p * 2 + 1];
  gnb_prologue(f, MULTI ? g.N : 1, tid, s_raw, s_coef);
#pragma unroll
  for (int h = 0; h < 4; ++h) store(0, 0, h, ra[h], rb[h]);
  __syncthreads();
  const int khalf = lane >> 5, l31 = lane & 31;
  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < ktiles;
    if (more) {
#pragma unroll
      for (int h = 0; h < 4; ++h) { load_a(kt + 1, h, ra[h]); load_b(kt + 1, h, rb[h]); }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k2 = 0; k2 < 32; k2 += 2) {
      float a = As[buf][32 * wave + k2 + khalf][l31];
      float b = Bs[buf][32 * wave + k2 + khalf][l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
#pragma unroll
      for (int h = 0; h < 4; ++h) store(buf ^ 1, kt + 1, h, ra[h], rb[h]);
    }
    __syncthreads();
  }
  // ---- four partial accumulators -> dx tile fin[32][33]
  float* red = &As[0][0][0];
  float* fin = &Bs[0][0][0];
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int s = tid + 256 * j, r = s >> 6, ln = s & 63;
    const float v = (red[(0 * 16 + r) * 64 + ln] + red[(1 * 16 + r) * 64 + ln]) + (red[(2 * 16 + r) * 64 + ln] + red[(3 * 16 + r) * 64 + ln]);
    fin[((r & 3) + 8 * (r >> 2) + 4 * (ln >> 5)) * 33 + (ln & 31)] = v;
  }
  __syncthreads();
  // ---- GroupNorm-backward reduce of the producer on this tile
  float dmv[4], xh[4];
  {
    const float yv[4] = {e_y.x, e_y.y, e_y.z, e_y.w}, av[4] = {e_add.x, e_add.y, e_add.z, e_add.w};
    const float ov[4] = {e_out.x, e_out.y, e_out.z, e_out.w}, gv[4] = {e_ga.x, e_ga.y, e_ga.z, e_ga.w};
    const float bv[4] = {e_be.x, e_be.y, e_be.z, e_be.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[j] = (yv[j] - mean_p) * rstd_p;
      const float act = g.out_p ? ov[j] : fmaf(xh[j], gv[j], bv[j]);          // the consumer's expression when not stored
      const float d = fin[e_row * 33 + e_cq + j] + av[j];
      dmv[j] = (e_ok && act > 0.f) ? d : 0.f;
    }
    if (e_ok) *reinterpret_cast<float4*>(g.dm_p + e_off) = make_float4(dmv[0], dmv[1], dmv[2], dmv[3]);
  }
  __syncthreads();                                   // fin consumed: reuse the B stage for the column sums
  float* sa = fin;                                   // [32 rows][33] A contributions, then [32][33] B contributions
  float* sb = fin + 32 * 33;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sa[e_row * 33 + e_cq + j] = dmv[j];
    sb[e_row * 33 + e_cq + j] = dmv[j] * xh[j];
  }
  __syncthreads();
  if (tid < 64) {
    const int c = tid & 31, which = tid >> 5;
    const float* src = which ? sb : sa;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += src[r * 33 + c];
    g.partials_p[((size_t)blockIdx.x * 2 + which) * g.C + n0 + c] = t;          // [mtile][2][C]
    // per-group gamma-weighted sums of this column block (its 32 channels lie in one group)
    float wsum = t * g.gamma_p[n0 + c];
    wsum += __shfl_xor(wsum, 16); wsum += __shfl_xor(wsum, 8); wsum += __shfl_xor(wsum, 4);
    wsum += __shfl_xor(wsum, 2); wsum += __shfl_xor(wsum, 1);
    if (c < DYB_GN_GROUPS)
      g.gpart_p[(((size_t)blockIdx.x * gridDim.y + blockIdx.y) * DYB_GN_GROUPS + c) * 2 + which] = (c == grp_p) ? wsum : 0.f;
  }
}

// out[i] = sum_z slab[z][i] (+ addend[i]);  n4 = element count / 4
// out = addend + scale * sum_z slabs[z] (scale 1: the plain fold; -fastlr with addend = the current weights: the fast-weight step of a
// SPLIT weight gradient, "fuse_fast")
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float4* __restrict__ slabs, const float4* __restrict__ addend,
                                                             float4* __restrict__ out, int nsplit, size_t n4, DybRep R, float scale) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, slabs); DYB_RB(R, addend); DYB_RB(R, out);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += step) {
    // eight slab loads in flight per trip (uniform bounds: no per-lane condition around a load)
    float4 s = slabs[i];
    for (int z0 = 1; z0 < nsplit; z0 += 8) {
      float4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = slabs[(size_t)(z0 + j < nsplit ? z0 + j : 0) * n4 + i];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (z0 + j >= nsplit) t[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      s.x += ((t[0].x + t[1].x) + (t[2].x + t[3].x)) + ((t[4].x + t[5].x) + (t[6].x + t[7].x));
      s.y += ((t[0].y + t[1].y) + (t[2].y + t[3].y)) + ((t[4].y + t[5].y) + (t[6].y + t[7].y));
      s.z += ((t[0].z + t[1].z) + (t[2].z + t[3].z)) + ((t[4].z + t[5].z) + (t[6].z + t[7].z));
      s.w += ((t[0].w + t[1].w) + (t[2].w + t[3].w)) + ((t[4].w + t[5].w) + (t[6].w + t[7].w));
    }
    if (addend) {
      float4 t = addend[i];
      s.x = fmaf(scale, s.x, t.x); s.y = fmaf(scale, s.y, t.y); s.z = fmaf(scale, s.z, t.z); s.w = fmaf(scale, s.w, t.w);
    }
    out[i] = s;
  }
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
static int conv_out_dim(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

static int fill_args(IgemmArgs& g, const ConvDesc& d, int mode) {
  DYB_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0, DYB_ERR_ARG);
  DYB_REQUIRE(dyb_is_pow2(d.C) && d.C >= 4 && dyb_is_pow2(d.K) && d.K >= 4, DYB_ERR_UNSUPPORTED);
  DYB_REQUIRE(d.stride == 1 || d.stride == 2, DYB_ERR_UNSUPPORTED);
  g.N = d.N; g.H = d.H; g.W = d.W; g.C = d.C; g.K = d.K; g.R = d.R; g.S = d.S;
  g.stride = d.stride; g.pad = d.pad;
  g.out_scale = 1.f;
  g.Ho = conv_out_dim(d.H, d.R, d.stride, d.pad);
  g.Wo = conv_out_dim(d.W, d.S, d.stride, d.pad);
  g.logC = dyb_ilog2(d.C); g.logK = dyb_ilog2(d.K);
  if (mode == MODE_FWD) { g.M = d.N * g.Ho * g.Wo; g.Ncols = d.K; g.Kdim = d.R * d.S * d.C; }
  else if (mode == MODE_DGRAD) { g.M = d.N * d.H * d.W; g.Ncols = d.C; g.Kdim = d.R * d.S * d.K; }
  else { g.M = d.R * d.S * d.C; g.Ncols = d.K; g.Kdim = d.N * g.Ho * g.Wo; }
  g.ktiles = dyb_cdiv(g.Kdim, BK);
  g.ktiles1 = g.ktiles;
  return DYB_OK;
}

// Split-K policy, a small cost model fitted to measurements on MI355X at these sizes (the sweep is
// tools/ab_split.sh; the landscape is flat within ~4 % around these values):
//   a K-step costs ~0.5 us of exposed latency;
//   forward       : the GroupNorm statistics kernel folds the slabs for ~0.15 us per slab;
//   backward, raw : the next GroupNorm-backward reduce folds them for ~0.2 + 0.05 us per slab;
//   backward, stand-alone fold launch (public entry points, shortcut branch): ~5.8 us + 0.1 per slab.
// Bounded so that the grid stays <= ~1024 workgroups and every split keeps >= 2 K-steps.  The DYB_*
// environment variables exist for that sweep only.
static float env_float(const char* name, float dflt) {
  const char* v = getenv(name);
  return v ? (float)atof(v) : dflt;
}

// ---- bf16 matrix-core mode of the calling host thread: set by the engine around a forward / backward of a plan that asked
// for it (dyb_hmr_set_bf16); the single-launch 1x1 kernels have no bf16 form, the tiled kernel + statistics launch serve
static thread_local bool t_bf16 = false;
DybBf16Scope::DybBf16Scope(bool on) : saved(t_bf16) { t_bf16 = on; }
DybBf16Scope::~DybBf16Scope() { t_bf16 = saved; }

// ---- current replica set of the calling host thread (dyb_common.h) ------------------------------------------------
static DybRep rep_single() {
  DybRep r{};
  r.n = 1;
  dyb_rep_identity(r);
  return r;
}
static thread_local DybRep t_rep = rep_single();
const DybRep& dyb_rep_current() { return t_rep; }
DybRepScope::DybRepScope(const DybRep& r) : saved(t_rep) { t_rep = r; }
DybRepScope::~DybRepScope() { t_rep = saved; }

// Counter region for the in-kernel split-K fold of the calling host thread's conv launches (the engine sets one per stream it issues
// convolutions on: two launches that may run concurrently must not share counters).  Outside a scope: no region, split launches
// leave slabs as before.  The words must be zero when the first launch uses them; every launch leaves them zero.  One word per
// (replica slot of the launch, tile): the region is NOT replicated - a pointer into replica 0's workspace serves every replica.
static thread_local DybConvSync t_conv_sync = {nullptr, 0};
// Weight-update scope of the calling thread (round 6, "fuse_fast"): while one is open, a throughput-form WEIGHT GRADIENT that (a) would
// write its result into [grads, grads + bytes), (b) runs unsplit (nsplit == 1: the accumulators hold the finished gradient tile) leaves
// p_next[off] = p_cur[off] - lr * g instead of g - the fast-weight step of learn2learn's MAML.adapt (p' = p - lr * dL/dp; call sites
// reference dynaboa_benchmark.py:136,140) fused into the epilogue: one read of the current weights and one write of the new ones instead
// of writing g and a streaming pass reading p and g and writing p'.  The span is appended to `spans` so that the caller's streaming
// pass can leave it out.  First order only (the inner gradients have no other reader).
static thread_local DybWgradUpdate t_wupd = {};
DybWgradUpdateScope::DybWgradUpdateScope(const DybWgradUpdate& u) : saved(t_wupd) { t_wupd = u; }
DybWgradUpdateScope::~DybWgradUpdateScope() { t_wupd = saved; }
const DybWgradUpdate& dyb_wgrad_update_current() { return t_wupd; }
static std::vector<DybSpan> t_debug_spans;
// tests / lab: a scope for the calling thread's plain weight-gradient calls until reset with grads = NULL; dyb_debug_wgrad_update_spans
// reports how many launches took the fused form since the scope was set
extern "C" int dyb_debug_set_wgrad_update(const float* grads, size_t bytes, const float* p_cur, float* p_next, float lr) {
  t_debug_spans.clear();
  if (!grads) { t_wupd = DybWgradUpdate{}; return DYB_OK; }
  DYB_REQUIRE(bytes && p_cur && p_next, DYB_ERR_ARG);
  t_wupd = DybWgradUpdate{grads, bytes, p_cur, p_next, lr, &t_debug_spans};
  return DYB_OK;
}
// the same for Adam ("fuse_adam"): theta / m / v updated in place from the accumulators; sc = device pointer to (step_size, bc2_sqrt)
extern "C" int dyb_debug_set_wgrad_adam(const float* grads, size_t bytes, float* theta, float* m, float* v, const float* sc, float b1, float b2,
                                        float eps) {
  t_debug_spans.clear();
  DYB_REQUIRE(grads && bytes && theta && m && v && sc, DYB_ERR_ARG);
  t_wupd = DybWgradUpdate{grads, bytes, theta, theta, 0.f, &t_debug_spans};
  t_wupd.adam_m = m; t_wupd.adam_v = v; t_wupd.adam_sc = sc; t_wupd.b1 = b1; t_wupd.b2 = b2; t_wupd.eps = eps;
  return DYB_OK;
}
extern "C" int dyb_debug_wgrad_update_spans() { return (int)t_debug_spans.size(); }
DybConvSyncScope::DybConvSyncScope(unsigned* ctr, int nwords) : saved(t_conv_sync) { t_conv_sync = DybConvSync{ctr, nwords}; }
DybConvSyncScope::~DybConvSyncScope() { t_conv_sync = saved; }
// tests / lab: a region for the calling thread's plain conv calls until reset with (NULL, 0)
extern "C" int dyb_debug_set_conv_sync(unsigned* ctr, int nwords) {
  DYB_REQUIRE((ctr == nullptr) == (nwords == 0) && nwords >= 0, DYB_ERR_ARG);
  t_conv_sync = DybConvSync{ctr, nwords};
  return DYB_OK;
}

// ---- run-time switches ------------------------------------------------------------------------------------
// Read from the environment ONCE (first use), never on the dispatch path; dyb_set_option changes one afterwards
// (tests / A-B runs).  Names: "k4" (single-launch 1x1 forward + statistics), "k4_bwd" (1x1 data gradient carries the
// producer's GroupNorm-backward reduce), "k4_batch" (both at batch > 1), "k4_maxc" (their channel limit), "rep_split"
// (replica-aware policy: split-K depth chosen for the replica-multiplied grid and, from "tp_min" replicas per launch on,
// the throughput schedule - dy materialised once per layer, plain gradient convolutions, no single-launch 1x1 kernels),
// "tp_kernel" (throughput schedule runs igemm_tp_kernel: 128x128-class tiles, 2 = its software-pipelined loop (default), 1 = round 2's
// phase-separated loop; 0 = the 64x64 kernel), "tp_grid" (workgroups
// its split-K aims for), "tp_batch_min" (8; > 0: the throughput schedule also for single-sequence launches of at least that batch:
// +13.5 % at batch 16 in BENCH_r02, and the schedule the bf16 form of igemm_tp_kernel needs), "tp_gn_wgs" (workgroups a GroupNorm launch aims for over all replicas
// under the throughput policy: their chunk counts are otherwise sized for one sequence and the launches dispatch-bound), "bf16" (bf16 matrix cores for direct calls of the conv entry points).
struct DybSwitches {
  std::atomic<int> k4, k4_bwd, k4_batch, k4_maxc, rep_split, bf16, tp_min, tp_kernel, tp_grid, tp_xcd, tp_batch_min, tp_gn_wgs, tp_occ, tp_gn_onepass,
      tp_gn_cap, tp_gn_threads, tp_gn_fuse_stats, tp_gn_poll, tp_fwd_nosplit2, pair, tp_wt, tp_fold, lat_fold, stat_folds, tp_gn_wt, tp_stem;
  DybSwitches() {
    auto env = [](const char* n, int d) { const char* v = getenv(n); return v ? atoi(v) : d; };
    k4 = env("DYB_K4", 1);
    k4_bwd = env("DYB_K4_BWD", 1);
    k4_batch = env("DYB_K4_BATCH", 1);
    k4_maxc = env("DYB_K4_MAXC", 1024);
    rep_split = env("DYB_REP_SPLIT", 0);
    bf16 = 0;
    tp_min = env("DYB_TP_MIN", 8);
    tp_kernel = env("DYB_TP_KERNEL", 2);
    tp_grid = env("DYB_TP_GRID", 512);
    tp_xcd = env("DYB_TP_XCD", 1);
    tp_batch_min = env("DYB_TP_BATCH_MIN", 8);       // measured crossover (r05 s17): batch 6: 257.7 | 251.0, 8: 280.8 | 294.9, 12: 304.0 | 366.6, 16: 320.9 | 418.6 frames/s latency | throughput
    tp_gn_wgs = env("DYB_TP_GN_WGS", 1024);
    tp_occ = env("DYB_TP_OCC", 0);
    tp_gn_onepass = env("DYB_TP_GN_ONEPASS", 2);
    tp_gn_cap = env("DYB_TP_GN_CAP", 0);
    tp_gn_threads = env("DYB_TP_GN_THREADS", 1024);
    tp_gn_fuse_stats = env("DYB_TP_GN_FUSE_STATS", 1);
    tp_gn_poll = env("DYB_TP_GN_POLL", 8);
    tp_fwd_nosplit2 = env("DYB_TP_FWD_NOSPLIT2", 1);
    pair = env("DYB_CONV_PAIR", 1);
    tp_wt = env("DYB_TP_WT", 1);
    tp_stem = env("DYB_TP_STEM", 1);
    tp_fold = env("DYB_TP_FOLD", 0);       // measured (r05 s3): 32 sequences 461 vs 463 frames/s off, 16: 362 vs 376 - the folding workgroups are a tail
    lat_fold = env("DYB_LAT_FOLD", 1);
    stat_folds = 0;
    tp_gn_wt = env("DYB_TP_GN_WT", 0);
  }
};
static DybSwitches& switches() {
  static DybSwitches* s = new DybSwitches();
  return *s;
}
// bf16 mode: the engine's scope for this thread, or the process-wide "bf16" switch (direct calls of the conv entry points)
bool dyb_bf16_current() { return t_bf16 || switches().bf16.load(std::memory_order_relaxed) != 0; }
static std::atomic<int>* find_switch(const char* name) {
  if (!name) return nullptr;
  DybSwitches& s = switches();
  if (!strcmp(name, "k4")) return &s.k4;
  if (!strcmp(name, "k4_bwd")) return &s.k4_bwd;
  if (!strcmp(name, "k4_batch")) return &s.k4_batch;
  if (!strcmp(name, "k4_maxc")) return &s.k4_maxc;
  if (!strcmp(name, "rep_split")) return &s.rep_split;
  if (!strcmp(name, "bf16")) return &s.bf16;
  if (!strcmp(name, "tp_min")) return &s.tp_min;
  if (!strcmp(name, "tp_kernel")) return &s.tp_kernel;
  if (!strcmp(name, "tp_grid")) return &s.tp_grid;
  if (!strcmp(name, "tp_xcd")) return &s.tp_xcd;
  if (!strcmp(name, "tp_batch_min")) return &s.tp_batch_min;
  if (!strcmp(name, "tp_gn_wgs")) return &s.tp_gn_wgs;
  if (!strcmp(name, "tp_occ")) return &s.tp_occ;
  if (!strcmp(name, "tp_gn_onepass")) return &s.tp_gn_onepass;
  if (!strcmp(name, "tp_gn_cap")) return &s.tp_gn_cap;
  if (!strcmp(name, "tp_gn_threads")) return &s.tp_gn_threads;
  if (!strcmp(name, "tp_gn_fuse_stats")) return &s.tp_gn_fuse_stats;
  if (!strcmp(name, "tp_gn_poll")) return &s.tp_gn_poll;
  if (!strcmp(name, "tp_fwd_nosplit2")) return &s.tp_fwd_nosplit2;
  if (!strcmp(name, "conv_pair")) return &s.pair;
  if (!strcmp(name, "tp_wt")) return &s.tp_wt;
  if (!strcmp(name, "tp_stem")) return &s.tp_stem;
  if (!strcmp(name, "tp_fold")) return &s.tp_fold;
  if (!strcmp(name, "lat_fold")) return &s.lat_fold;
  if (!strcmp(name, "tp_gn_wt")) return &s.tp_gn_wt;
  if (!strcmp(name, "stat_folds")) return &s.stat_folds;      // (a counter, not a switch: conv launches that folded their split in-kernel)
  return nullptr;
}
extern "C" int dyb_set_option(const char* name, int value) {
  std::atomic<int>* p = find_switch(name);
  DYB_REQUIRE(p, DYB_ERR_ARG);
  p->store(value);
  return DYB_OK;
}
extern "C" int dyb_get_option(const char* name, int* value) {
  std::atomic<int>* p = find_switch(name);
  DYB_REQUIRE(p && value, DYB_ERR_ARG);
  *value = p->load();
  return DYB_OK;
}
// `raw`: the consumer folds the slabs itself (forward: GroupNorm statistics; backward: the next
// GroupNorm-backward reduce), so a split costs per-slab read time there instead of a launch.
static int choose_split(const IgemmArgs& g, size_t ws_floats, int mode, bool raw = false) {
  static const float fold_us = env_float("DYB_BWD_FOLD_US", 5.8f), raw_fold_us = env_float("DYB_RAW_FOLD_US", 0.2f),
                     raw_slab_us = env_float("DYB_RAW_SLAB_US", 0.05f), kstep_us = env_float("DYB_KSTEP_US", 0.5f),
                     fwd_slab_us = env_float("DYB_FWD_SLAB_US", 0.15f);
  static const int grid_cap = (int)env_float("DYB_GRID_CAP", 1024.f), min_steps = (int)env_float("DYB_MIN_STEPS", 2.f);
  int tiles = dyb_cdiv(g.M, BM) * dyb_cdiv(g.Ncols, BN);
  // rep_split = 1: a launch covering n sequence replicas already has n x the workgroups, so the grid cap counts them and
  // the K loop is split less (fewer slabs to write and fold).  Off by default: the split depth fixes the summation order,
  // and with it off a replica's results are bit-identical to the same sequence running alone.
  if (switches().rep_split.load(std::memory_order_relaxed)) tiles *= dyb_rep_current().n;
  int maxs = g.ktiles / (min_steps > 0 ? min_steps : 1);
  if (maxs < 1) maxs = 1;
  int gridcap = grid_cap / tiles;
  if (gridcap < 1) gridcap = 1;
  if (maxs > gridcap) maxs = gridcap;
  int best = 1;
  float bt = kstep_us * g.ktiles;
  for (int s = 2; s <= maxs; ++s) {
    float steps = (float)dyb_cdiv(g.ktiles, s);
    float t = kstep_us * steps + (mode == MODE_FWD ? fwd_slab_us * s : raw ? raw_fold_us + raw_slab_us * s : fold_us + 0.1f * s);
    if (t < bt - 0.25f) { bt = t; best = s; }
  }
  size_t per = (size_t)g.M * g.Ncols;
  while (best > 1 && (size_t)best * per > ws_floats) --best;
  return best;
}

extern "C" size_t dyb_conv2d_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
  // worst case over the three modes: nsplit * M * Ncols floats, nsplit bounded by the policy above
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  size_t best = 0;
  for (int mode = 0; mode < 3; ++mode) {
    IgemmArgs g{};
    if (fill_args(g, d, mode) != DYB_OK) return 0;
    int s = choose_split(g, (size_t)-1 / 8, mode);
    int sr = choose_split(g, (size_t)-1 / 8, mode, true);
    if (sr > s) s = sr;
    size_t need = (s > 1) ? (size_t)s * g.M * g.Ncols * sizeof(float) : 0;
    if (need > best) best = need;
  }
  return best;
}

// ---- optional per-launch timing of the conv kernels (bench.py's roofline leg) ----------------------
// While a timing scope is open (process-wide: torch's autograd issues the backward from its own thread,
// the metric forwards come from a third) every igemm launch carries a (start, stop) event
// pair on its own dispatch (hipExtLaunchKernelGGL: no extra packets in the stream), so the durations
// are those of the kernels as they run inside the real path, on whatever stream they were issued to.
struct IgemmTimingRec { char tag; int nsplit, nrep; ConvDesc d; double flop; };
struct IgemmTiming {
  std::vector<hipEvent_t> ev;
  std::vector<IgemmTimingRec> rec;             // one per timed launch (event pair i <-> rec[i])
  size_t used = 0;
  double flop = 0.0, bytes = 0.0;
};
static std::string g_timing_table;             // per-shape table of the last closed scope (dyb_conv_timing_table)
static double g_timing_union_ms = 0.0;         // of the last closed scope: time during which at least one timed launch was running
static IgemmTiming* g_timing = nullptr;
static std::mutex g_timing_mu;

extern "C" int dyb_conv_timing_begin(int max_launches) {
  std::lock_guard<std::mutex> lock(g_timing_mu);
  DYB_REQUIRE(max_launches > 0 && !g_timing, DYB_ERR_ARG);
  IgemmTiming* t = new IgemmTiming();
  t->ev.resize((size_t)max_launches * 2);
  for (auto& e : t->ev)
    if (hipEventCreate(&e) != hipSuccess) return DYB_ERR_LAUNCH;
  g_timing = t;
  return DYB_OK;
}
// Closes the scope (the caller has synchronised the device): total kernel milliseconds, number of timed
// launches, algorithmic flop and algorithmic bytes (operands read once + result written once) they stand for.
extern "C" int dyb_conv_timing_end(double* ms_total, long long* launches, double* flop, double* bytes) {
  std::lock_guard<std::mutex> lock(g_timing_mu);
  DYB_REQUIRE(g_timing && ms_total && launches && flop && bytes, DYB_ERR_ARG);
  IgemmTiming* t = g_timing;
  g_timing = nullptr;
  double ms = 0.0;
  struct Row { double ms = 0.0, flop = 0.0; long long n = 0; };
  std::map<std::string, Row> rows;
  for (size_t i = 0; i + 1 < t->used; i += 2) {
    float one = 0.f;
    if (hipEventElapsedTime(&one, t->ev[i], t->ev[i + 1]) != hipSuccess) continue;
    ms += one;
    const IgemmTimingRec& r = t->rec[i / 2];
    char key[160];
    snprintf(key, sizeof key, "%c,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d", r.tag, r.d.N, r.d.H, r.d.W, r.d.C, r.d.K, r.d.R, r.d.stride,
             r.d.pad, r.nsplit, r.nrep);
    Row& row = rows[key];
    row.ms += one; row.flop += r.flop; row.n += 1;
  }
  // union of the launches' [start, stop] intervals (launches on the two queues overlap: their durations add up to more than the time
  // the chip spent on convolutions)
  {
    std::vector<std::pair<float, float>> iv;
    for (size_t i = 0; i + 1 < t->used; i += 2) {
      float a = 0.f, b = 0.f;
      if (hipEventElapsedTime(&a, t->ev[0], t->ev[i]) != hipSuccess || hipEventElapsedTime(&b, t->ev[0], t->ev[i + 1]) != hipSuccess) continue;
      iv.emplace_back(a, b);
    }
    std::sort(iv.begin(), iv.end());
    double u = 0.0;
    float lo = 0.f, hi = -1.f;
    for (const auto& p : iv) {
      if (hi < 0.f || p.first > hi) {
        if (hi >= 0.f) u += hi - lo;
        lo = p.first; hi = p.second;
      } else if (p.second > hi) {
        hi = p.second;
      }
    }
    if (hi >= 0.f) u += hi - lo;
    g_timing_union_ms = u;
  }
  g_timing_table = "kind,N,H,W,C,K,R,stride,pad,nsplit,nrep,launches,ms_total,gflop_total\n";
  for (const auto& kv : rows) {
    char tail[96];
    snprintf(tail, sizeof tail, ",%lld,%.6f,%.6f\n", kv.second.n, kv.second.ms, kv.second.flop * 1e-9);
    g_timing_table += kv.first + tail;
  }
  *ms_total = ms; *launches = (long long)(t->used / 2); *flop = t->flop; *bytes = t->bytes;
  for (auto& e : t->ev) (void)hipEventDestroy(e);
  delete t;
  return DYB_OK;
}

// of the scope closed last: milliseconds during which at least one timed conv launch was executing (union of the launches' intervals)
extern "C" double dyb_conv_timing_union_ms() {
  std::lock_guard<std::mutex> lock(g_timing_mu);
  return g_timing_union_ms;
}
// Per-shape table of the scope closed last (CSV, header line first): kind f/d/w (t/u/v: throughput form) = tiled forward / data gradient / weight
// gradient, F/D = the single-launch 1x1 kernels; returns the bytes needed including the terminator.
extern "C" size_t dyb_conv_timing_table(char* buf, size_t cap) {
  std::lock_guard<std::mutex> lock(g_timing_mu);
  if (buf && cap) {
    size_t n = g_timing_table.size() < cap - 1 ? g_timing_table.size() : cap - 1;
    memcpy(buf, g_timing_table.data(), n);
    buf[n] = 0;
  }
  return g_timing_table.size() + 1;
}

// a (start, stop) event pair + the launch's algorithmic flop / bytes booked, when a timing scope is open
static void timing_acquire(const ConvDesc& d, hipEvent_t* ev0, hipEvent_t* ev1, char tag = '?', int nsplit = 1, int fused_update = 0) {
  if (!g_timing) return;                      // unlocked fast path when no scope is open
  std::lock_guard<std::mutex> lock(g_timing_mu);
  if (!g_timing || g_timing->used + 2 > g_timing->ev.size()) return;
  *ev0 = g_timing->ev[g_timing->used];
  *ev1 = g_timing->ev[g_timing->used + 1];
  g_timing->used += 2;
  const double creal = d.C == 4 ? 3.0 : (double)d.C;        // the stem's 4th input channel is padding
  const double px = (double)d.N * conv_out_dim(d.H, d.R, d.stride, d.pad) * conv_out_dim(d.W, d.S, d.stride, d.pad);
  const double nrep = (double)dyb_rep_current().n;          // a launch covering n sequence replicas does n convolutions
  const double fl = nrep * 2.0 * px * d.K * d.R * d.S * creal;
  g_timing->rec.push_back(IgemmTimingRec{tag, nsplit, (int)nrep, d, fl});
  g_timing->flop += fl;
  // operands once + result once; a weight gradient that writes the fast weights itself ("fuse_fast", 1) also reads the current weights, one
  // that applies Adam ("fuse_adam", 2) reads and writes the weights and both moments instead of writing the gradient
  const double wpass = fused_update == 2 ? 6.0 : fused_update == 1 ? 2.0 : 1.0;
  g_timing->bytes += nrep * 4.0 * ((double)d.N * d.H * d.W * creal + (double)d.R * d.S * creal * d.K * wpass + px * d.K);
}

// ---- phase probe of the throughput kernel (tools/tp_probe.py) ------------------------------------------------------------
// While set, launches of igemm_tp_kernel whose (mode, H, C, K, R) match write, per wave, eight 64-bit words into `buf`
// ([workgroup][wave][8]: start clock, end clock, cycles issuing loads, cycles in the MFMA block, cycles waiting for the loads +
// staging them, cycles at the barrier, K-steps, replica) - s_memtime stamps taken by lane 0.  Later matching launches overwrite.
struct TpProbe { unsigned long long* buf; long cap_wgs; int mode, H, C, K, R; };
static TpProbe g_probe = {nullptr, 0, 0, 0, 0, 0, 0};
static std::atomic<bool> g_probe_on{false};           // unlocked fast path of the launch wrapper
extern "C" int dyb_conv_probe_set(void* buf, long cap_wgs, int mode, int H, int C, int K, int R) {
  std::lock_guard<std::mutex> lock(g_timing_mu);
  g_probe_on.store(false);
  g_probe = TpProbe{reinterpret_cast<unsigned long long*>(buf), cap_wgs, mode, H, C, K, R};
  g_probe_on.store(buf != nullptr);
  return DYB_OK;
}
// the probe buffer for a launch of `grid` workgroups of this layer, or NULL
static unsigned long long* probe_for(int mode, const ConvDesc& d, long wgs) {
  if (!g_probe_on.load(std::memory_order_acquire)) return nullptr;
  std::lock_guard<std::mutex> lock(g_timing_mu);
  const TpProbe& p = g_probe;
  return (p.buf && p.mode == mode && p.H == d.H && p.C == d.C && p.K == d.K && p.R == d.R && wgs <= p.cap_wgs) ? p.buf : nullptr;
}

// Fold of a COMPACT stride-2 1x1 data gradient: out[n][h][w][:] = (h, w both even ? sum_z slab[z][(n, h/2, w/2)][:] : 0) (+ addend).
// The other three phase classes of such a conv receive no tap at all: they used to be tiles of their own that wrote zeros into
// every split's slab (75 % of the slab traffic, and workgroups that only stored) - now the zeros are written once, here.
__global__ __launch_bounds__(256) void fold_scatter_s2_kernel(const float4* __restrict__ slabs, int nsplit, size_t slab4,
                                                               const float4* __restrict__ addend, float4* __restrict__ out, int N, int H,
                                                               int W, int C4, int Hc, int Wc, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, slabs); DYB_RB(R, addend); DYB_RB(R, out);
  const size_t total = (size_t)N * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const size_t pix = i / C4;
    const int w = (int)(pix % W);
    const size_t t = pix / W;
    const int h = (int)(t % H), n = (int)(t / H);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (((h | w) & 1) == 0) {
      const size_t src = ((size_t)(n * Hc + (h >> 1)) * Wc + (w >> 1)) * C4 + c4;
      s = slabs[src];
      for (int z = 1; z < nsplit; ++z) {
        const float4 v = slabs[(size_t)z * slab4 + src];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    if (addend) {
      const float4 a = addend[i];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    out[i] = s;
  }
}

#include "igemm_tp.inc"

// The throughput form (igemm_tp.inc) of one mode; same contract as run_igemm below.
static bool tp_eligible(int mode, const ConvDesc& d, const GnBwdFuse* fuse) {
  if (fuse || !dyb_throughput_mode(d.N)) return false;
  const int tpk_ = switches().tp_kernel.load(std::memory_order_relaxed);
  if (!tpk_) return false;
  if (dyb_bf16_current() && (tpk_ < 2 || (mode == MODE_WGRAD && TPK / conv_out_dim(d.W, d.S, d.stride, d.pad) >= conv_out_dim(d.H, d.R, d.stride, d.pad))))
    return false;                                  // the bf16 form exists for the pipelined loop only
  // buffer addressing: 32-bit byte offsets, one mask bit per filter tap
  const size_t lim = 0x7fffffffu / sizeof(float);
  const int Ho = conv_out_dim(d.H, d.R, d.stride, d.pad), Wo = conv_out_dim(d.W, d.S, d.stride, d.pad);
  if ((size_t)d.N * d.H * d.W * d.C + (size_t)d.W * d.C * 8 >= lim || (size_t)d.N * Ho * Wo * d.K + (size_t)Wo * d.K * 8 >= lim ||
      (size_t)d.R * d.S * d.C * d.K >= lim)
    return false;
  // the stem (Cin = 4): its own loader form of the pipelined forward kernel ("C4", igemm_tp.inc; switch tp_stem)
  if (mode == MODE_FWD && d.C == 4)
    return tpk_ >= 2 && d.K <= 64 && d.S >= TPK / 4 && !dyb_bf16_current() && switches().tp_stem.load(std::memory_order_relaxed) != 0;
  if (mode == MODE_FWD) return d.C % TPK == 0 && d.R * d.S <= 32;
  if (mode == MODE_DGRAD) return d.K % TPK == 0 && d.R * d.S <= 32;
  return true;
}
// stats_part / stats_nrec (forward only): where the GroupNorm statistics of the output may leave with the tiles ("tp_gn_fuse_stats":
// one image, nsplit == 1, at most min(Ho*Wo, 256) records - what a layer's partial slot holds); *stats_nrec = records written, 0 = none
static int run_igemm_tp(int mode, const ConvDesc& d, IgemmArgs g, float* out, const float* addend, void* ws, size_t ws_bytes,
                        int* raw_slabs_out, hipStream_t st, const GnFwdFuse* nfuse, float* stats_part = nullptr, int* stats_nrec = nullptr) {
  const DybRep& R = dyb_rep_current();
  // tile form: 128x128, or one 64-slab along the short side.  A stride-2 data gradient enumerates its rows by phase class
  // ((h+pad)&1, (w+pad)&1) - tiles never mix classes, each class loops over its own taps - so the row count that matters is
  // a class's
  const bool classes = mode == MODE_DGRAD && d.stride == 2;
  const int rows_form = classes ? d.N * ((d.H + 1) >> 1) * ((d.W + 1) >> 1) : g.M;
  const int form = g.Ncols <= 64 ? 2 : (rows_form <= 64 ? 1 : 0);
  const int TM = form == 0 ? 128 : form == 1 ? 64 : 256, TN = form == 0 ? 128 : form == 1 ? 256 : 64;
  g.ktiles = dyb_cdiv(g.Kdim, TPK);
  g.xcd = switches().tp_xcd.load(std::memory_order_relaxed);
  g.wt = switches().tp_wt.load(std::memory_order_relaxed);
  int mtiles = dyb_cdiv(g.M, TM), work_mtiles = mtiles;
  g.cls_tile0[0] = g.cls_tile0[1] = g.cls_tile0[2] = g.cls_tile0[3] = 0;
  if (classes) {
    mtiles = work_mtiles = 0;
    for (int c = 0; c < 4; ++c) {
      const int h0 = ((c >> 1) + d.pad) & 1, w0 = ((c & 1) + d.pad) & 1;
      const int rows = d.N * ((d.H - h0 + 1) >> 1) * ((d.W - w0 + 1) >> 1);
      DYB_REQUIRE(rows < (1 << 20) && d.H <= 1024 && d.W <= 1024, DYB_ERR_UNSUPPORTED);
      g.cls_tile0[c] = mtiles;
      mtiles += dyb_cdiv(rows, TM);
      const bool has_taps = ((d.R - (c >> 1) + 1) >> 1) > 0 && ((d.S - (c & 1) + 1) >> 1) > 0;
      if (has_taps) work_mtiles += dyb_cdiv(rows, TM);                 // a class without taps only writes zeros (+ addend)
    }
    g.ktiles = ((d.R + 1) >> 1) * ((d.S + 1) >> 1) * (d.K / TPK);      // the deepest class (split policy)
  }
  // 1x1 stride-2 (the downsample convs): one class has the tap, three have none - compute that class alone into compact slabs
  g.compact = 0;
  g.slab_rows = g.M;
  const int Hc = (d.H + 1) >> 1, Wc = (d.W + 1) >> 1;
  if (classes && d.R == 1 && d.S == 1 && d.pad == 0 && ws && (size_t)d.N * Hc * Wc * g.Ncols * sizeof(float) <= ws_bytes) {
    g.compact = 1;
    g.slab_rows = d.N * Hc * Wc;
    mtiles = work_mtiles = dyb_cdiv(g.slab_rows, TM);
    g.cls_tile0[0] = 0;
    g.cls_tile0[1] = g.cls_tile0[2] = g.cls_tile0[3] = 0x7fffffff;
  }
  const long tiles = (long)work_mtiles * dyb_cdiv(g.Ncols, TN) * R.n;
  int s = (int)(switches().tp_grid.load(std::memory_order_relaxed) / tiles);
  const int maxs = g.ktiles / 4 > 0 ? g.ktiles / 4 : 1;                 // every split keeps >= 4 K-steps
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  // forward, one image: an unsplit launch leaves the output's GroupNorm statistics with its tiles (no statistics launch, no slabs to
  // write and fold) - worth more than the second workgroup per CU a split of two would buy ("tp_fwd_nosplit2")
  bool stats_fusable = false;
  int stats_records = 0;
  if (mode == MODE_FWD && stats_part && stats_nrec && d.N == 1 && d.K >= 64 && d.K <= 2048 &&
      switches().tp_gn_fuse_stats.load(std::memory_order_relaxed)) {
    stats_records = mtiles * dyb_cdiv(g.Ncols, TN) * (TM / 64) * (TN / 64);
    const int HWo = g.Ho * g.Wo;
    stats_fusable = stats_records <= (HWo < 256 ? HWo : 256);
  }
  if (stats_fusable && s == 2 && switches().tp_fwd_nosplit2.load(std::memory_order_relaxed)) s = 1;
  const size_t per = (size_t)g.slab_rows * g.Ncols, ws_floats = ws ? ws_bytes / sizeof(float) : 0;
  // the result tiles leave through 32-bit byte offsets into a buffer resource clamped to 2^31 - 1 bytes: a larger slab would alias TP_OOB
  // and its stores would be dropped silently
  DYB_REQUIRE(per * sizeof(float) < 0x7fffffffull, DYB_ERR_UNSUPPORTED);
  while (s > 1 && (size_t)s * per > ws_floats) --s;
  g.nsplit = s;
  g.tiles_per_split = dyb_cdiv(g.ktiles, g.nsplit);
  g.nsplit = dyb_cdiv(g.ktiles, g.tiles_per_split);
  if (raw_slabs_out) *raw_slabs_out = 1;
  const bool split = g.nsplit > 1;
  g.out = (split || g.compact) ? reinterpret_cast<float*>(ws) : out;          // (compact: always through the scatter fold)
  g.addend = (split || g.compact) ? nullptr : addend;
  g.out_scale = 1.f;
  // (a SPLIT weight gradient whose slabs the fold launch below adds: the fast-weight step rides in that launch - not Adam, not the in-kernel fold)
  const float* red_addend = addend;
  float* red_out = out;
  float red_scale = 1.f;
  if (mode == MODE_WGRAD && t_wupd.grads && !t_wupd.adam_m && split && !g.compact && !addend && !raw_slabs_out &&
      !((switches().tp_fold.load(std::memory_order_relaxed) >> mode) & 1)) {
    const char *lo = reinterpret_cast<const char*>(t_wupd.grads), *o = reinterpret_cast<const char*>(out);
    if (o >= lo && o + per * sizeof(float) <= lo + t_wupd.bytes) {
      const size_t off = (size_t)(o - lo) / sizeof(float);
      red_out = t_wupd.p_next + off; red_addend = t_wupd.p_cur + off; red_scale = -t_wupd.lr;
      if (t_wupd.spans) t_wupd.spans->push_back(DybSpan{off, per});
    }
  }
  if (mode == MODE_WGRAD && t_wupd.grads && !split && !g.compact && !addend) {
    const char *lo = reinterpret_cast<const char*>(t_wupd.grads), *o = reinterpret_cast<const char*>(out);
    if (o >= lo && o + per * sizeof(float) <= lo + t_wupd.bytes) {
      const size_t off = (size_t)(o - lo) / sizeof(float);
      g.out = t_wupd.p_next + off;
      g.addend = t_wupd.p_cur + off;
      g.out_scale = -t_wupd.lr;
      if (t_wupd.adam_m) {
        g.adam_m = t_wupd.adam_m + off; g.adam_v = t_wupd.adam_v + off; g.adam_sc = t_wupd.adam_sc;
        g.adam_b1 = t_wupd.b1; g.adam_b2 = t_wupd.b2; g.adam_eps = t_wupd.eps;
        g.out_scale = 0.f;                  // (only marks the launch as a fused update for the byte accounting)
      }
      if (t_wupd.spans) t_wupd.spans->push_back(DybSpan{off, per});
    }
  }
  dim3 grid(mtiles, dyb_cdiv(g.Ncols, TN), g.nsplit * R.n);
  g.probe = probe_for(mode, d, (long)grid.x * grid.y * grid.z);
  // "tp_kernel" 2 (default): the software-pipelined loop (PIPE 1), 3: the same with two K-steps of loads in flight (PIPE 2);
  // 1: round 2's phase-separated loop (also what the phase probe and
  // weight gradients over maps too small for the branch-free pixel walk use)
  const int tpk = switches().tp_kernel.load(std::memory_order_relaxed);
  const int pipe = (tpk >= 2 && !g.probe && !(mode == MODE_WGRAD && TPK / g.Wo >= g.Ho)) ? (tpk >= 3 ? 2 : 1) : 0;
  // in-kernel fold ("tp_fold"): a counter region in scope, the pipelined kernel, plain slabs (not the compact stride-2 form)
  const bool fold = split && !g.compact && pipe != 0 && t_conv_sync.ctr && (long)grid.x * grid.y * R.n <= (long)t_conv_sync.nwords &&
                    !(raw_slabs_out && mode != MODE_FWD) && ((switches().tp_fold.load(std::memory_order_relaxed) >> mode) & 1);   // bit per mode
  g.fold_out = nullptr; g.fold_addend = nullptr; g.fold_ctr = nullptr;
  if (fold) {
    g.fold_out = out; g.fold_addend = addend; g.fold_ctr = t_conv_sync.ctr;
    switches().stat_folds.fetch_add(1, std::memory_order_relaxed);
  }
  g.gn_part = nullptr;
  if (stats_nrec) *stats_nrec = 0;
  if (stats_fusable && (!split || fold)) {
    g.gn_part = stats_part;
    *stats_nrec = stats_records;
  }
  const bool bf = dyb_bf16_current();
  DYB_REQUIRE(!bf || pipe != 0, DYB_ERR_UNSUPPORTED);
  DYB_REQUIRE(!g.adam_m || pipe != 0, DYB_ERR_UNSUPPORTED);        // (the Adam epilogue lives in the pipelined forms)
  GnFwdFuse nf{};
  if (nfuse) nf = *nfuse;
  DYB_REQUIRE(!nfuse || d.N <= 64, DYB_ERR_UNSUPPORTED);
  // "tp_occ" = k > 0: at most k workgroups of this launch per CU - unused dynamic LDS makes a (k+1)-th not fit (160 KiB per CU)
  unsigned lds_pad = 0;
  if (const int occ = switches().tp_occ.load(std::memory_order_relaxed); occ > 0) {
    const unsigned lds_static = 2u * TPK * (unsigned)(TM + 4 + TN + 4) * 4u + (nfuse ? 64u * DYB_GN_GROUPS * 2u * 12u : 24u);
    const unsigned want = 163840u / (unsigned)occ - 1024u;
    if (want > lds_static) lds_pad = want - lds_static;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  timing_acquire(d, &ev0, &ev1, mode == MODE_FWD ? 't' : mode == MODE_DGRAD ? 'u' : 'v', g.nsplit, g.adam_m ? 2 : (g.out_scale != 1.f ? 1 : 0));
#define DYB_TP_LAUNCH3(M_, FA_, WM_, WN_, P_)                                                                                      \
  do {                                                                                                                             \
    if (ev0) hipExtLaunchKernelGGL((igemm_tp_kernel<M_, FA_, WM_, WN_, P_>), grid, dim3(256), lds_pad, st, ev0, ev1, 0, g, nf, R); \
    else hipLaunchKernelGGL((igemm_tp_kernel<M_, FA_, WM_, WN_, P_>), grid, dim3(256), lds_pad, st, g, nf, R);                     \
  } while (0)
#define DYB_TP_LAUNCH2(M_, FA_, WM_, WN_)                 \
  do {                                                    \
    if (bf) DYB_TP_LAUNCH_BF(M_, FA_, WM_, WN_);          \
    else if (pipe == 2) DYB_TP_LAUNCH3(M_, FA_, WM_, WN_, 2);  \
    else if (pipe) DYB_TP_LAUNCH3(M_, FA_, WM_, WN_, 1);  \
    else DYB_TP_LAUNCH3(M_, FA_, WM_, WN_, 0);            \
  } while (0)
#define DYB_TP_LAUNCH_BF(M_, FA_, WM_, WN_)                                                                                          \
  do {                                                                                                                             \
    if (ev0) hipExtLaunchKernelGGL((igemm_tp_kernel<M_, FA_, WM_, WN_, 1, true>), grid, dim3(256), lds_pad, st, ev0, ev1, 0, g, nf, R); \
    else hipLaunchKernelGGL((igemm_tp_kernel<M_, FA_, WM_, WN_, 1, true>), grid, dim3(256), lds_pad, st, g, nf, R);                     \
  } while (0)
#define DYB_TP_LAUNCH(M_, FA_)                        \
  do {                                                \
    if (form == 0) DYB_TP_LAUNCH2(M_, FA_, 2, 2);     \
    else if (form == 1) DYB_TP_LAUNCH2(M_, FA_, 1, 4); \
    else DYB_TP_LAUNCH2(M_, FA_, 4, 1);               \
  } while (0)
  if (mode == MODE_FWD && d.C == 4) {
    DYB_REQUIRE(!nfuse && form == 2 && !bf, DYB_ERR_UNSUPPORTED);
    if (ev0) hipExtLaunchKernelGGL((igemm_tp_kernel<MODE_FWD, false, 4, 1, 1, false, true>), grid, dim3(256), lds_pad, st, ev0, ev1, 0, g, nf, R);
    else hipLaunchKernelGGL((igemm_tp_kernel<MODE_FWD, false, 4, 1, 1, false, true>), grid, dim3(256), lds_pad, st, g, nf, R);
  } else if (mode == MODE_FWD) {
    if (nfuse) DYB_TP_LAUNCH(MODE_FWD, true);
    else DYB_TP_LAUNCH(MODE_FWD, false);
  } else if (mode == MODE_DGRAD) {
    DYB_REQUIRE(!nfuse, DYB_ERR_UNSUPPORTED);
    DYB_TP_LAUNCH(MODE_DGRAD, false);
  } else {
    if (nfuse) DYB_TP_LAUNCH(MODE_WGRAD, true);
    else DYB_TP_LAUNCH(MODE_WGRAD, false);
  }
#undef DYB_TP_LAUNCH
#undef DYB_TP_LAUNCH2
#undef DYB_TP_LAUNCH3
#undef DYB_TP_LAUNCH_BF
  DYB_CHECK_LAUNCH();
  if (g.compact) {
    const size_t tot4 = (size_t)g.M * g.Ncols / 4;
    int blocks = (int)((tot4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fold_scatter_s2_kernel, dim3(blocks, 1, R.n), dim3(256), 0, st, reinterpret_cast<const float4*>(ws), g.nsplit, per / 4,
                       reinterpret_cast<const float4*>(addend), reinterpret_cast<float4*>(out), d.N, d.H, d.W, g.Ncols / 4, Hc, Wc, R);
    DYB_CHECK_LAUNCH();
    return DYB_OK;
  }
  if (fold) return DYB_OK;                       // the result (and, forward, its statistics) left with the last workgroup of every tile
  if (split) {
    if (raw_slabs_out) { *raw_slabs_out = g.nsplit; return DYB_OK; }
    size_t n4 = per / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, 1, R.n), dim3(256), 0, st, reinterpret_cast<const float4*>(ws),
                       reinterpret_cast<const float4*>(red_addend), reinterpret_cast<float4*>(red_out), g.nsplit, n4, R, red_scale);
    DYB_CHECK_LAUNCH();
  }
  return DYB_OK;
}

// Runs one mode.  If `raw_slabs_out` is non-null and the policy picks nsplit>1 the slabs are left
// in the workspace un-reduced and *raw_slabs_out = nsplit (caller folds them, e.g. inside the
// GroupNorm statistics kernel); otherwise the result lands in `out`.
static int run_igemm(int mode, const ConvDesc& d, const float* A, const float* B, float* out, const float* addend,
                     void* ws, size_t ws_bytes, int* raw_slabs_out, hipStream_t st, const GnBwdFuse* fuse = nullptr,
                     const GnFwdFuse* nfuse = nullptr, float* stats_part = nullptr, int* stats_nrec = nullptr, const float* A2 = nullptr,
                     const float* B2 = nullptr) {
  DYB_REQUIRE(A && B && out, DYB_ERR_ARG);
  IgemmArgs g{};
  int rc = fill_args(g, d, mode);
  if (rc != DYB_OK) return rc;
  g.A = A; g.B = B;
  if (stats_nrec) *stats_nrec = 0;
  if (A2 || B2) {            // operand pair: one K loop over both (latency form without fused loaders only - dyb_conv_pair_supported)
    DYB_REQUIRE(A2 && B2 && !fuse && !nfuse && !tp_eligible(mode, d, nullptr) && !dyb_bf16_current(), DYB_ERR_UNSUPPORTED);
    g.A2 = A2; g.B2 = B2;
    g.ktiles = 2 * g.ktiles1;
  } else if (tp_eligible(mode, d, fuse)) return run_igemm_tp(mode, d, g, out, addend, ws, ws_bytes, raw_slabs_out, st, nfuse, stats_part, stats_nrec);
  g.nsplit = choose_split(g, ws ? ws_bytes / sizeof(float) : 0, mode, raw_slabs_out != nullptr && mode != MODE_FWD);
  g.tiles_per_split = dyb_cdiv(g.ktiles, g.nsplit);
  g.nsplit = dyb_cdiv(g.ktiles, g.tiles_per_split);       // drop empty tail splits
  if (raw_slabs_out) *raw_slabs_out = 1;
  const bool split = g.nsplit > 1;
  g.out = split ? reinterpret_cast<float*>(ws) : out;
  g.addend = split ? nullptr : addend;
  const DybRep& R = dyb_rep_current();
  dim3 grid(dyb_cdiv(g.M, BM), dyb_cdiv(g.Ncols, BN), g.nsplit * R.n);
  // in-kernel split-K fold ("lat_fold": a counter region in scope; fp32 form - its epilogue goes through LDS) and, forward of one
  // image, the output's GroupNorm statistics with the tiles (one record per workgroup tile; what a layer's partial slot holds)
  // the fp32 epilogue addresses a result matrix / slab through 32-bit byte offsets (TP_OOB = 2^31 marks "past the end"): larger is refused
  // rather than silently dropped
  DYB_REQUIRE(dyb_bf16_current() || (size_t)g.M * g.Ncols * sizeof(float) < 0x7fffffffu, DYB_ERR_UNSUPPORTED);
  const bool lat = !dyb_bf16_current() && switches().lat_fold.load(std::memory_order_relaxed);
  // (not where the caller's next kernel folds the slabs while it reads them anyway: a gradient handed on as raw slabs)
  const bool kfold = split && lat && t_conv_sync.ctr && (long)grid.x * grid.y * R.n <= (long)t_conv_sync.nwords &&
                     !(raw_slabs_out && mode != MODE_FWD);
  g.fold_out = nullptr; g.fold_addend = nullptr; g.fold_ctr = nullptr;
  if (kfold) {
    g.fold_out = out; g.fold_addend = addend; g.fold_ctr = t_conv_sync.ctr;
    switches().stat_folds.fetch_add(1, std::memory_order_relaxed);
  }
  // "fuse_fast" in the latency form (one sequence): the finished weight-gradient tile - unsplit, or folded in-kernel by the last workgroup
  // to arrive, or folded by the fold launch below - leaves p_next = p_cur - lr * g instead of g (see DybWgradUpdateScope)
  const float* red_addend = addend;
  float* red_out = out;
  float red_scale = 1.f;
  if (mode == MODE_WGRAD && t_wupd.grads && !t_wupd.adam_m && !dyb_bf16_current() && !A2 && !addend && !raw_slabs_out) {
    const char *lo = reinterpret_cast<const char*>(t_wupd.grads), *o = reinterpret_cast<const char*>(out);
    const size_t cnt = (size_t)g.M * g.Ncols;
    if (o >= lo && o + cnt * sizeof(float) <= lo + t_wupd.bytes) {
      const size_t off = (size_t)(o - lo) / sizeof(float);
      if (!split) { g.out = t_wupd.p_next + off; g.addend = t_wupd.p_cur + off; g.out_scale = -t_wupd.lr; }
      else if (kfold) { g.fold_out = t_wupd.p_next + off; g.fold_addend = t_wupd.p_cur + off; g.out_scale = -t_wupd.lr; }
      else { red_out = t_wupd.p_next + off; red_addend = t_wupd.p_cur + off; red_scale = -t_wupd.lr; }
      if (t_wupd.spans) t_wupd.spans->push_back(DybSpan{off, cnt});
    }
  }
  g.gn_part = nullptr;
  if (mode == MODE_FWD && stats_part && stats_nrec && lat && d.N == 1 && (!split || kfold) && d.K >= 64 && d.K % 64 == 0 && dyb_is_pow2(d.K) &&
      !A2 && (int)(grid.x * grid.y) <= (g.Ho * g.Wo < 256 ? g.Ho * g.Wo : 256)) {
    g.gn_part = stats_part;
    *stats_nrec = (int)(grid.x * grid.y);
  }
  GnBwdFuse f{};
  GnFwdFuse nf{};
  if (fuse) f = *fuse;
  if (nfuse) nf = *nfuse;
  DYB_REQUIRE(!(fuse || nfuse) || d.N <= 64, DYB_ERR_UNSUPPORTED);
  const dim3 blk(256);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  timing_acquire(d, &ev0, &ev1, mode == MODE_FWD ? 'f' : mode == MODE_DGRAD ? 'd' : 'w', g.nsplit);
  const bool bf = dyb_bf16_current();
#define DYB_IGEMM_LAUNCH1(M_, GB_, FA_, BF_)                                                                           \
  do {                                                                                                                 \
    if (ev0) hipExtLaunchKernelGGL((igemm_mfma_kernel<M_, GB_, FA_, BF_>), grid, blk, 0, st, ev0, ev1, 0, g, f, nf, R); \
    else hipLaunchKernelGGL((igemm_mfma_kernel<M_, GB_, FA_, BF_>), grid, blk, 0, st, g, f, nf, R);                    \
  } while (0)
#define DYB_IGEMM_LAUNCH(M_, GB_, FA_)              \
  do {                                              \
    if (bf) DYB_IGEMM_LAUNCH1(M_, GB_, FA_, true);  \
    else DYB_IGEMM_LAUNCH1(M_, GB_, FA_, false);    \
  } while (0)
  if (mode == MODE_FWD) {
    DYB_REQUIRE(!fuse, DYB_ERR_UNSUPPORTED);
    if (nfuse) DYB_IGEMM_LAUNCH(MODE_FWD, false, true);
    else DYB_IGEMM_LAUNCH(MODE_FWD, false, false);
  } else if (mode == MODE_DGRAD) {
    DYB_REQUIRE(!nfuse, DYB_ERR_UNSUPPORTED);
    if (fuse) DYB_IGEMM_LAUNCH(MODE_DGRAD, true, false);
    else DYB_IGEMM_LAUNCH(MODE_DGRAD, false, false);
  } else {
    if (fuse && nfuse) DYB_IGEMM_LAUNCH(MODE_WGRAD, true, true);
    else if (fuse) DYB_IGEMM_LAUNCH(MODE_WGRAD, true, false);
    else if (nfuse) DYB_IGEMM_LAUNCH(MODE_WGRAD, false, true);
    else DYB_IGEMM_LAUNCH(MODE_WGRAD, false, false);
  }
#undef DYB_IGEMM_LAUNCH
#undef DYB_IGEMM_LAUNCH1
  DYB_CHECK_LAUNCH();
  if (kfold) return DYB_OK;                      // folded by the last workgroup of every tile
  if (split) {
    if (raw_slabs_out) { *raw_slabs_out = g.nsplit; return DYB_OK; }
    size_t n4 = (size_t)g.M * g.Ncols / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, 1, R.n), dim3(256), 0, st, reinterpret_cast<const float4*>(ws),
                       reinterpret_cast<const float4*>(red_addend), reinterpret_cast<float4*>(red_out), g.nsplit, n4, R, red_scale);
    DYB_CHECK_LAUNCH();
  }
  return DYB_OK;
}

// Operand pairs (internal: the tangent passes of hvp_engine.inc).  mode 0 / 1 / 2 = forward / data gradient / weight gradient:
//   forward          out = conv(a1, b1) + conv(a2, b2)                 a = activations, b = weights
//   data gradient    out = dgrad(a1, b1) + dgrad(a2, b2) (+ addend)    a = output gradients, b = weights
//   weight gradient  out = wgrad(a1, b1) + wgrad(a2, b2)               a = activations, b = output gradients
// as one launch with a K loop over both pairs (and one split-K fold instead of two folds and an add).
bool dyb_conv_pair_supported(int mode, const ConvDesc& d) {
  return switches().pair.load(std::memory_order_relaxed) && !tp_eligible(mode, d, nullptr) && !dyb_bf16_current();
}
int dyb_conv_pair(int mode, const ConvDesc& d, const float* a1, const float* b1, const float* a2, const float* b2, float* out,
                  const float* addend, void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(mode >= 0 && mode <= 2 && a2 && b2, DYB_ERR_ARG);
  DYB_REQUIRE(mode == MODE_DGRAD || !addend, DYB_ERR_UNSUPPORTED);
  return run_igemm(mode, d, a1, b1, out, addend, ws, ws_bytes, nullptr, st, nullptr, nullptr, nullptr, nullptr, a2, b2);
}

extern "C" int dyb_debug_conv_pair(int mode, const float* a1, const float* b1, const float* a2, const float* b2, float* out,
                                   const float* addend, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, void* ws,
                                   size_t ws_bytes, hipStream_t st) {
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  DYB_REQUIRE(mode >= 0 && mode <= 2, DYB_ERR_ARG);
  DYB_REQUIRE(dyb_conv_pair_supported(mode, d), DYB_ERR_UNSUPPORTED);
  return dyb_conv_pair(mode, d, a1, b1, a2, b2, out, addend, ws, ws_bytes, st);
}

int dyb_conv_fwd_raw(const ConvDesc& d, const float* x, const float* w, float* y, void* ws, size_t ws_bytes,
                     int* nslabs, hipStream_t st) {
  return run_igemm(MODE_FWD, d, x, w, y, nullptr, ws, ws_bytes, nslabs, st);
}

extern "C" int dyb_conv2d_nhwc_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int K, int R,
                                   int S, int stride, int pad, void* ws, size_t ws_bytes, hipStream_t st) {
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  return run_igemm(MODE_FWD, d, x, w, y, nullptr, ws, ws_bytes, nullptr, st);
}
// dx = conv_transpose(dy, w) (+ addend, e.g. the gradient arriving over the residual edge)
extern "C" int dyb_conv2d_nhwc_dgrad(const float* dy, const float* w, float* dx, const float* addend, int N, int H,
                                     int W, int C, int K, int R, int S, int stride, int pad, void* ws,
                                     size_t ws_bytes, hipStream_t st) {
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  return run_igemm(MODE_DGRAD, d, dy, w, dx, addend, ws, ws_bytes, nullptr, st);
}
extern "C" int dyb_conv2d_nhwc_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K,
                                     int R, int S, int stride, int pad, void* ws, size_t ws_bytes, hipStream_t st) {
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  return run_igemm(MODE_WGRAD, d, x, dy, dw, nullptr, ws, ws_bytes, nullptr, st);
}

// Diagnostic (tools/tp_lab.py): one convolution mode for `nrep` sequence replicas in ONE launch, the way the native stepper issues
// them - x / w / dy / out are [nrep][...] stacks (replica stride = one tensor), each replica with its own weights.  mode 0 forward
// (out = y), 1 data gradient (x unused, out = dx), 2 weight gradient (w unused, out = dw).  ws: nrep equal slices.
extern "C" int dyb_debug_conv_replicas(int mode, const float* x, const float* w, const float* dy, float* out, int nrep, int N, int H,
                                       int W, int C, int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(mode >= 0 && mode <= 2 && nrep >= 1 && nrep <= DYB_MAX_REPLICAS && out, DYB_ERR_ARG);
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  const int Ho = conv_out_dim(H, R, stride, pad), Wo = conv_out_dim(W, S, stride, pad);
  const size_t nx = (size_t)N * H * W * C, nw = (size_t)R * S * C * K, ny = (size_t)N * Ho * Wo * K;
  DybRep Rp{};
  Rp.n = nrep;
  dyb_rep_identity(Rp);
  auto arena = [&](const void* lo, size_t bytes) {
    if (!lo) return;
    Rp.lo[Rp.narenas] = reinterpret_cast<const char*>(lo); Rp.span[Rp.narenas] = bytes; Rp.stride[Rp.narenas] = bytes; ++Rp.narenas;
  };
  arena(x, nx * 4); arena(w, nw * 4); arena(dy, ny * 4);
  arena(out, (mode == 0 ? ny : mode == 1 ? nx : nw) * 4);
  const size_t wsl = (ws_bytes / (size_t)nrep) & ~(size_t)255;
  arena(ws, wsl);
  DybRepScope scope(Rp);
  if (mode == 0) return run_igemm(MODE_FWD, d, x, w, out, nullptr, ws, wsl, nullptr, st);
  if (mode == 1) return run_igemm(MODE_DGRAD, d, dy, w, out, nullptr, ws, wsl, nullptr, st);
  return run_igemm(MODE_WGRAD, d, x, dy, out, nullptr, ws, wsl, nullptr, st);
}

// ---- data / weight gradient with the GroupNorm backward of the conv's output formed in the loader ----
// dm: masked gradient of the GroupNorm output, y_gn/stats: saved conv output and (mean, rstd), part: the
// partial-sum block dyb_groupnorm_bwd_reduce wrote for this layer (N images, Ho*Wo pixels, K channels).
// nch / ncolb: row-chunk and column-block counts of the partial block (0: the layout dyb_groupnorm_bwd_reduce uses;
// a producer that wrote the partials from a conv epilogue passes its tile counts)
static int make_fuse(GnBwdFuse& f, const ConvDesc& d, const float* y_gn, const float* stats, const float* part,
                     const float* gamma, float* dgamma, float* dbeta, int nch = 0, int ncolb = 0) {
  DYB_REQUIRE(y_gn && stats && part && gamma, DYB_ERR_ARG);
  DYB_REQUIRE(d.K % 16 == 0, DYB_ERR_UNSUPPORTED);
  int Ho = conv_out_dim(d.H, d.R, d.stride, d.pad), Wo = conv_out_dim(d.W, d.S, d.stride, d.pad);
  if (nch <= 0 || ncolb <= 0) dyb_gn_bwd_layout(d.N, Ho * Wo, d.K, &nch, &ncolb);
  f.y = y_gn; f.stats = stats; f.gamma = gamma;
  f.partials = part;
  f.gpart = part + (size_t)d.N * nch * 2 * d.K;
  f.dgamma = dgamma; f.dbeta = dbeta;
  f.kparts = nch * ncolb; f.nchunks = nch; f.HW = Ho * Wo;
  f.inv_m = 1.0f / ((float)(d.K / DYB_GN_GROUPS) * (float)(Ho * Wo));
  return DYB_OK;
}
extern "C" int dyb_conv2d_nhwc_dgrad_gn(const float* dm, const float* y_gn, const float* stats, const float* part,
                                        const float* gamma, const float* w, float* dx, const float* addend, int N, int H,
                                        int W, int C, int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes,
                                        hipStream_t st) {
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  GnBwdFuse f{};
  int rc = make_fuse(f, d, y_gn, stats, part, gamma, nullptr, nullptr);
  if (rc != DYB_OK) return rc;
  return run_igemm(MODE_DGRAD, d, dm, w, dx, addend, ws, ws_bytes, nullptr, st, &f);
}
extern "C" int dyb_conv2d_nhwc_wgrad_gn(const float* x, const float* dm, const float* y_gn, const float* stats,
                                        const float* part, const float* gamma, float* dw, float* dgamma, float* dbeta,
                                        int N, int H, int W, int C, int K, int R, int S, int stride, int pad, void* ws,
                                        size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(dgamma && dbeta, DYB_ERR_ARG);
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  GnBwdFuse f{};
  int rc = make_fuse(f, d, y_gn, stats, part, gamma, dgamma, dbeta);
  if (rc != DYB_OK) return rc;
  return run_igemm(MODE_WGRAD, d, x, dm, dw, nullptr, ws, ws_bytes, nullptr, st, &f);
}
bool dyb_conv_dgrad_k4_ok(const ConvDesc& d) {
  const DybSwitches& sw = switches();                   // k4_bwd on: 1.40 -> 1.31 ms per backward
  const int enabled = sw.k4_bwd.load(std::memory_order_relaxed), max_k = sw.k4_maxc.load(std::memory_order_relaxed);
  const bool batch_ok = d.N == 1 || (sw.k4_batch.load(std::memory_order_relaxed) && d.N <= 64);
  return enabled && !dyb_bf16_current() && !dyb_throughput_mode(d.N) && batch_ok && d.R == 1 && d.S == 1 && d.pad == 0 && d.stride == 1 && dyb_is_pow2(d.K) && d.K >= 128 &&
         d.K <= max_k && d.C % 128 == 0 && d.H * d.W <= 784;
}
// dx of the 1x1 conv `d` (never materialised as such) -> dm / partials of the producer's GroupNorm; *nch, *ncolb = the
// partial block's row-chunk / column-block counts (tile counts) for the producer's own fused gradients.
int dyb_conv_dgrad_k4(const ConvDesc& d, const GnBwdSrc& src, const float* w, const float* addend, const float* y_p,
                      const float* out_p, const float* stats_p, const float* gamma_p, const float* beta_p, float* dm_p,
                      float* part_p, int* nch, int* ncolb, hipStream_t st, hipEvent_t done) {
  DYB_REQUIRE(dyb_conv_dgrad_k4_ok(d) && w && y_p && stats_p && gamma_p && beta_p && dm_p && part_p && nch && ncolb,
              DYB_ERR_UNSUPPORTED);
  GnBwdFuse f{};
  int rc = make_fuse(f, d, src.y, src.stats, src.part, src.gamma, nullptr, nullptr, src.nch, src.ncolb);
  if (rc != DYB_OK) return rc;
  const int M = d.H * d.W, tpi = dyb_cdiv(M, 32);
  const DybRep& R = dyb_rep_current();
  dim3 grid(d.N * tpi, d.C / 32, R.n);
  K4DgradArgs g{src.dm, w, addend, y_p, out_p, stats_p, gamma_p, beta_p, dm_p, part_p, part_p + (size_t)grid.x * 2 * d.C, M, d.C, d.K,
                d.N, tpi};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  timing_acquire(d, &ev0, &ev1, 'D');                 // bench.py's conv timing scope
#define DYB_K4D_LAUNCH(MU_)                                                                                          \
  do {                                                                                                               \
    if (ev0) hipExtLaunchKernelGGL((igemm_k4_dgrad_kernel<MU_>), grid, dim3(256), 0, st, ev0, ev1, 0, g, f, R);      \
    else if (done) hipExtLaunchKernelGGL((igemm_k4_dgrad_kernel<MU_>), grid, dim3(256), 0, st, nullptr, done, 0, g, f, R); \
    else hipLaunchKernelGGL((igemm_k4_dgrad_kernel<MU_>), grid, dim3(256), 0, st, g, f, R);                          \
  } while (0)
  if (d.N == 1) DYB_K4D_LAUNCH(false);
  else DYB_K4D_LAUNCH(true);
#undef DYB_K4D_LAUNCH
  DYB_CHECK_LAUNCH();
  // inside a timing scope the launch carries the timing pair, so the caller's completion event is a record of its own
  if (ev0 && done && hipEventRecord(done, st) != hipSuccess) return DYB_ERR_LAUNCH;
  *nch = tpi;                                  // row chunks per image
  *ncolb = (int)grid.y;
  return DYB_OK;
}
int dyb_conv_dgrad_gn_raw(const ConvDesc& d, const GnBwdSrc& src, const float* w, float* dx, const float* addend, void* ws,
                          size_t ws_bytes, int* nslabs, hipStream_t st) {
  GnBwdFuse f{};
  int rc = make_fuse(f, d, src.y, src.stats, src.part, src.gamma, nullptr, nullptr, src.nch, src.ncolb);
  if (rc != DYB_OK) return rc;
  return run_igemm(MODE_DGRAD, d, src.dm, w, dx, addend, ws, ws_bytes, nslabs, st, &f);
}
static void make_nfuse(GnFwdFuse& nf, const ConvDesc& d, const float* partials, const float* stats_in, const float* gamma,
                       const float* beta, float* stats_out, int relu);
// ---- throughput schedule: plain gradient convolutions over a materialised dy (dyb_common.h) -----------------------------
// workgroups per image the GroupNorm kernels' chunking may use when a launch covers several sequence replicas under the
// throughput policy (0: the single-sequence sizing applies): "tp_gn_wgs" (1024) over all replicas of the launch
int dyb_gn_replica_share(int N) {
  if (!dyb_throughput_mode()) return 0;
  const int reps = dyb_rep_current().n;
  if (reps <= 1) return 0;
  int w = switches().tp_gn_wgs.load(std::memory_order_relaxed) / (reps * (N > 0 ? N : 1));
  return w < 1 ? 1 : w;
}
// throughput schedule, GroupNorm backward: "tp_gn_onepass" 2 (default) = the one-pass kernel for every layer that qualifies (slabs
// of several row chunks meet on a counter), 1 = only layers whose (image, group) slab is one workgroup, 0 = the two-launch
// reduce + apply; "tp_gn_cap" = float4 per workgroup (0: 8 x "tp_gn_threads", the workgroup size 256 / 512 / 1024; tests force
// several chunks on small shapes)
int dyb_tp_gn_onepass() { return switches().tp_gn_onepass.load(std::memory_order_relaxed); }
int dyb_tp_gn_cap() { return switches().tp_gn_cap.load(std::memory_order_relaxed); }
int dyb_tp_gn_threads() { return switches().tp_gn_threads.load(std::memory_order_relaxed); }
int dyb_tp_gn_poll() { return switches().tp_gn_poll.load(std::memory_order_relaxed); }
int dyb_tp_gn_wt() { return switches().tp_gn_wt.load(std::memory_order_relaxed); }
bool dyb_throughput_mode(int batch) {
  const DybSwitches& sw = switches();
  if (sw.rep_split.load(std::memory_order_relaxed) != 0 && dyb_rep_current().n >= sw.tp_min.load(std::memory_order_relaxed)) return true;
  const int bmin = sw.tp_batch_min.load(std::memory_order_relaxed);
  return bmin > 0 && batch >= bmin;
}
int dyb_conv_dgrad_plain_raw(const ConvDesc& d, const float* dy, const float* w, float* dx, const float* addend, void* ws,
                             size_t ws_bytes, int* nslabs, hipStream_t st) {
  return run_igemm(MODE_DGRAD, d, dy, w, dx, addend, ws, ws_bytes, nslabs, st);
}
int dyb_conv_wgrad_plain(const ConvDesc& d, const float* x, const float* y_prev, const float* stats_prev, const float* gamma_prev,
                         const float* beta_prev, const float* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st) {
  if (x) return run_igemm(MODE_WGRAD, d, x, dy, dw, nullptr, ws, ws_bytes, nullptr, st);
  DYB_REQUIRE(y_prev && stats_prev && gamma_prev && beta_prev && d.C % 16 == 0, DYB_ERR_ARG);
  GnFwdFuse nf{};
  make_nfuse(nf, d, nullptr, stats_prev, gamma_prev, beta_prev, nullptr, 1);
  return run_igemm(MODE_WGRAD, d, y_prev, dy, dw, nullptr, ws, ws_bytes, nullptr, st, nullptr, &nf);
}
// out = sum_z slabs[z] (+ addend) over n floats (n % 4 == 0)
int dyb_splitk_fold(const float* slabs, int nslabs, size_t n, const float* addend, float* out, hipStream_t st) {
  size_t n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, 1, R.n), dim3(256), 0, st, reinterpret_cast<const float4*>(slabs),
                     reinterpret_cast<const float4*>(addend), reinterpret_cast<float4*>(out), nslabs, n4, R, 1.f);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}


// ---- K4 host side ------------------------------------------------------------------------------------
bool dyb_conv_k4_ok(const ConvDesc& d) {
  const int Ho = conv_out_dim(d.H, d.R, d.stride, d.pad), Wo = conv_out_dim(d.W, d.S, d.stride, d.pad);
  const DybSwitches& sw = switches();
  const int enabled = sw.k4.load(std::memory_order_relaxed), max_c = sw.k4_maxc.load(std::memory_order_relaxed);
  // Cin <= 512: a K-step of this kernel costs ~1.9 us (measured: 8.6 / 11.8 / 20 us at 2 / 4 / 8 steps - every step is a
  // cold-L2 round trip), so beyond 4 steps the tiled kernel's split-K over more workgroups + the statistics launch is faster
  const bool batch_ok = d.N == 1 || (sw.k4_batch.load(std::memory_order_relaxed) && d.N <= 64);
  return enabled && !dyb_bf16_current() && !dyb_throughput_mode(d.N) && batch_ok && d.R == 1 && d.S == 1 && d.pad == 0 && dyb_is_pow2(d.C) && d.C >= 128 && d.C <= max_c &&
         d.K % 128 == 0 && Ho * Wo <= 784;
}
// conv (+ producer GroupNorm in the loader when nf) -> y and its GroupNorm partials in one launch; *nchunks = partial count
int dyb_conv_fwd_k4(const ConvDesc& d, const float* x, const float* w, float* y, float* partials, const GnFwdFuse* nf,
                    int* nchunks, hipStream_t st) {
  DYB_REQUIRE(x && w && y && partials && nchunks && dyb_conv_k4_ok(d), DYB_ERR_UNSUPPORTED);
  K4Args g{x, w, y, partials, d.H, d.W, d.C, d.K, d.stride, conv_out_dim(d.H, 1, d.stride, 0), conv_out_dim(d.W, 1, d.stride, 0), 0,
           d.N, 0};
  g.M = g.Ho * g.Wo;
  g.tpi = dyb_cdiv(g.M, 32);
  const DybRep& R = dyb_rep_current();
  dim3 grid(d.N * g.tpi, d.K / 32, R.n);
  GnFwdFuse f{};
  if (nf) f = *nf;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  timing_acquire(d, &ev0, &ev1, 'F');
#define DYB_K4_LAUNCH(FA_, MU_)                                                                                       \
  do {                                                                                                                \
    if (ev0) hipExtLaunchKernelGGL((igemm_k4_fwd_kernel<FA_, MU_>), grid, dim3(256), 0, st, ev0, ev1, 0, g, f, R);    \
    else hipLaunchKernelGGL((igemm_k4_fwd_kernel<FA_, MU_>), grid, dim3(256), 0, st, g, f, R);                        \
  } while (0)
  if (d.N == 1) {
    if (nf) DYB_K4_LAUNCH(true, false);
    else DYB_K4_LAUNCH(false, false);
  } else {
    if (nf) DYB_K4_LAUNCH(true, true);
    else DYB_K4_LAUNCH(false, true);
  }
#undef DYB_K4_LAUNCH
  DYB_CHECK_LAUNCH();
  *nchunks = (int)(g.tpi * grid.y);             // partial records per image
  return DYB_OK;
}

// ---- producer GroupNorm(+ReLU) applied in the loader (GnFwdFuse) ---------------------------------
static void make_nfuse(GnFwdFuse& nf, const ConvDesc& d, const float* partials, const float* stats_in, const float* gamma,
                       const float* beta, float* stats_out, int relu) {
  nf.partials = partials; nf.stats_in = stats_in; nf.gamma = gamma; nf.beta = beta; nf.stats_out = stats_out;
  nf.HW = d.H * d.W;
  nf.nchunks = dyb_gn_fwd_chunks(d.N, nf.HW);
  nf.eps = DYB_GN_EPS;
  nf.relu = relu;
}
int dyb_conv_fwd_gnin_raw(const ConvDesc& d, const float* y_prev, const float* part_prev, int nch_prev,
                          const float* gamma_prev, const float* beta_prev, int relu_prev, float* stats_prev_out, const float* w,
                          float* y, void* ws, size_t ws_bytes, int* nslabs, hipStream_t st) {
  DYB_REQUIRE(y_prev && part_prev && gamma_prev && beta_prev && d.C % 16 == 0, DYB_ERR_ARG);
  GnFwdFuse nf{};
  make_nfuse(nf, d, part_prev, nullptr, gamma_prev, beta_prev, stats_prev_out, relu_prev);
  if (nch_prev > 0) nf.nchunks = nch_prev;
  return run_igemm(MODE_FWD, d, y_prev, w, y, nullptr, ws, ws_bytes, nslabs, st, nullptr, &nf);
}
// Forward conv of one layer INCLUDING the GroupNorm statistics of its output: y and per-chunk (sum, sum of squares)
// partials (*nchunks records of [G][2]).  x is a plain activation, or - part_prev != NULL - the producer's raw conv
// output whose GroupNorm(+ReLU) is applied in the loader from its nch_prev partials.  Small 1x1 layers at batch 1 take
// the single-launch K4 kernel; everything else is the tiled conv + dyb_groupnorm_stats (which folds the split-K slabs).
extern "C" int dyb_groupnorm_stats(const float*, int, float*, float*, int, int, int, hipStream_t);
extern "C" int dyb_conv2d_nhwc_fwd_gnstats(const float* x, const float* part_prev, int nch_prev, const float* gamma_prev,
                                           const float* beta_prev, int relu_prev, float* stats_prev_out, const float* w,
                                           float* y, float* partials, int* nchunks, int N, int H, int W, int C, int K, int R,
                                           int S, int stride, int pad, void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(x && w && y && partials && nchunks, DYB_ERR_ARG);
  DYB_REQUIRE(!part_prev || (gamma_prev && beta_prev && nch_prev > 0 && C % 16 == 0), DYB_ERR_ARG);
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  const int Ho = conv_out_dim(H, R, stride, pad), Wo = conv_out_dim(W, S, stride, pad);
  GnFwdFuse nf{};
  if (part_prev) {
    make_nfuse(nf, d, part_prev, nullptr, gamma_prev, beta_prev, stats_prev_out, relu_prev);
    nf.nchunks = nch_prev;
  }
  if (dyb_conv_k4_ok(d)) return dyb_conv_fwd_k4(d, x, w, y, partials, part_prev ? &nf : nullptr, nchunks, st);
  int nslabs = 1, nrec = 0;
  int rc = run_igemm(MODE_FWD, d, x, w, y, nullptr, ws, ws_bytes, &nslabs, st, nullptr, part_prev ? &nf : nullptr, partials, &nrec);
  if (rc != DYB_OK) return rc;
  if (nrec > 0) {                        // throughput kernel, no split: the statistics left with the tiles (igemm_tp.inc epilogue)
    *nchunks = nrec;
    return DYB_OK;
  }
  *nchunks = dyb_gn_fwd_chunks(N, Ho * Wo);
  return dyb_groupnorm_stats(reinterpret_cast<const float*>(ws), nslabs, y, partials, N, Ho * Wo, K, st);
}
// y = conv(relu?(gn(y_prev)), w): y_prev is the producer's raw conv output [N][H][W][C], part_prev its
// dyb_groupnorm_stats partials; the producer's (mean, rstd) are saved to stats_prev_out.
extern "C" int dyb_conv2d_nhwc_fwd_gnin(const float* y_prev, const float* part_prev, const float* gamma_prev,
                                        const float* beta_prev, int relu_prev, float* stats_prev_out, const float* w, float* y,
                                        int N, int H, int W, int C, int K, int R, int S, int stride, int pad, void* ws,
                                        size_t ws_bytes, hipStream_t st) {
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  return dyb_conv_fwd_gnin_raw(d, y_prev, part_prev, -1, gamma_prev, beta_prev, relu_prev, stats_prev_out, w, y, ws, ws_bytes,
                               nullptr, st);
}
// weight gradient of such a conv: A operand relu?(gn(y_prev)) formed on the fly from the saved
// (mean, rstd) of the producer, B operand dy formed on the fly as in dyb_conv2d_nhwc_wgrad_gn.
extern "C" int dyb_conv2d_nhwc_wgrad_gn_gnin(const float* y_prev, const float* stats_prev, const float* gamma_prev,
                                             const float* beta_prev, int relu_prev, const float* dm, const float* y_gn,
                                             const float* stats, const float* part, const float* gamma, float* dw,
                                             float* dgamma, float* dbeta, int N, int H, int W, int C, int K, int R, int S,
                                             int stride, int pad, void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(y_prev && stats_prev && gamma_prev && beta_prev && dgamma && dbeta && C % 16 == 0, DYB_ERR_ARG);
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  GnBwdFuse f{};
  int rc = make_fuse(f, d, y_gn, stats, part, gamma, dgamma, dbeta);
  if (rc != DYB_OK) return rc;
  GnFwdFuse nf{};
  make_nfuse(nf, d, nullptr, stats_prev, gamma_prev, beta_prev, nullptr, relu_prev);
  return run_igemm(MODE_WGRAD, d, y_prev, dm, dw, nullptr, ws, ws_bytes, nullptr, st, &f, &nf);
}

// Data gradient of conv (N,H,W,C,K,...) fused with the GroupNorm-backward reduce of the PRODUCER of its input (the
// layer whose normalised output feeds this conv; its GroupNorm is over [N][H*W][C]): dm_p / part_p come out, dx itself
// is scratch.  Small 1x1 layers at batch 1 with DYB_K4_BWD=1 run as one launch (K4 dgrad); otherwise this is
// dyb_conv2d_nhwc_dgrad_gn into dx_scratch followed by dyb_groupnorm_bwd_reduce.  *nch_p / *ncolb_p: layout of part_p
// for the producer's own fused gradients (pass them on as nch / ncolb there).
extern "C" int dyb_groupnorm_bwd_reduce(const float*, const float*, const float*, const float*, const float*, const float*, float*,
                                        float*, int, int, int, int, hipStream_t);
extern "C" int dyb_conv2d_nhwc_dgrad_gn_reduce(const float* dm, const float* y_gn, const float* stats, const float* part, int nch,
                                               int ncolb, const float* gamma, const float* w, const float* addend,
                                               const float* y_p, const float* out_p, const float* stats_p,
                                               const float* gamma_p, const float* beta_p, float* dm_p, float* part_p,
                                               int* nch_p, int* ncolb_p, float* dx_scratch, int N, int H, int W, int C, int K,
                                               int R, int S, int stride, int pad, void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(nch_p && ncolb_p && dm_p && part_p && y_p && stats_p && gamma_p && beta_p, DYB_ERR_ARG);
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  GnBwdSrc src{dm, y_gn, stats, part, gamma, nch, ncolb};
  if (dyb_conv_dgrad_k4_ok(d))
    return dyb_conv_dgrad_k4(d, src, w, addend, y_p, out_p, stats_p, gamma_p, beta_p, dm_p, part_p, nch_p, ncolb_p, st, nullptr);
  DYB_REQUIRE(dx_scratch, DYB_ERR_ARG);
  int rc = dyb_conv_dgrad_gn_raw(d, src, w, dx_scratch, addend, ws, ws_bytes, nullptr, st);
  if (rc != DYB_OK) return rc;
  dyb_gn_bwd_layout(N, H * W, C, nch_p, ncolb_p);
  return dyb_groupnorm_bwd_reduce(dx_scratch, out_p, y_p, stats_p, gamma_p, beta_p, dm_p, part_p, N, H * W, C, 1, st);
}
// the fused gradients with an explicit partial-block layout (what a K4 dgrad producer hands on)
extern "C" int dyb_conv2d_nhwc_dgrad_gn_n(const float* dm, const float* y_gn, const float* stats, const float* part, int nch,
                                          int ncolb, const float* gamma, const float* w, float* dx, const float* addend, int N,
                                          int H, int W, int C, int K, int R, int S, int stride, int pad, void* ws,
                                          size_t ws_bytes, hipStream_t st) {
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  GnBwdSrc src{dm, y_gn, stats, part, gamma, nch, ncolb};
  return dyb_conv_dgrad_gn_raw(d, src, w, dx, addend, ws, ws_bytes, nullptr, st);
}
extern "C" int dyb_conv2d_nhwc_wgrad_gn_n(const float* x, const float* y_prev, const float* stats_prev, const float* gamma_prev,
                                          const float* beta_prev, const float* dm, const float* y_gn, const float* stats,
                                          const float* part, int nch, int ncolb, const float* gamma, float* dw, float* dgamma,
                                          float* dbeta, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                          void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(dgamma && dbeta && (x || (y_prev && stats_prev && gamma_prev && beta_prev)), DYB_ERR_ARG);
  ConvDesc d{N, H, W, C, K, R, S, stride, pad};
  GnBwdFuse f{};
  int rc = make_fuse(f, d, y_gn, stats, part, gamma, dgamma, dbeta, nch, ncolb);
  if (rc != DYB_OK) return rc;
  if (!y_prev) return run_igemm(MODE_WGRAD, d, x, dm, dw, nullptr, ws, ws_bytes, nullptr, st, &f);
  GnFwdFuse nf{};
  make_nfuse(nf, d, nullptr, stats_prev, gamma_prev, beta_prev, nullptr, 1);
  return run_igemm(MODE_WGRAD, d, y_prev, dm, dw, nullptr, ws, ws_bytes, nullptr, st, &f, &nf);
}
