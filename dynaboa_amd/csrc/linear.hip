// Fully-connected layers of the iterative SMPL-parameter regressor (reference model/hmr.py:82-90,
// 158-172): fc1 2205->1024, fc2 1024->1024 and the three decoder heads (144+10+3 rows, stored as
// one [160][1024] matrix).  At batch 1..16 these are GEMV-class and bound by streaming the weight
// rows (fc1 = 9 MB), so: one wave per output row, 16 B per lane, butterfly reduction; the
// transposed product for backward reads the same rows coalesced and splits the output-row range
// over blockIdx.y into partial slabs (deterministic fold, no atomics); weight gradients of all
// three regressor iterations are produced by ONE rank-(3B) outer-product pass per matrix so each
// dW is written exactly once.
//
// Weight layout is the reference's (out_features, in_features) row-major with the row stride
// `ldw` padded to a multiple of 4 floats (fc1: 2205 -> 2208); pad columns are kept at zero.
#include <stdlib.h>

#include "dyb_common.h"

#define LIN_BT 4   // batch tile held in registers

// y[b][o] = bias[o] + W[o][:] . x[b][:] (+ res[b][o]);  grid = ceil(O/4), block = 256 (4 waves)
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                         int ldw, const float* __restrict__ bias,
                                                         const float* __restrict__ res, int ldres, float* __restrict__ y,
                                                         int ldy, int B, int I, int O, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, x); DYB_RB(R, w); DYB_RB(R, bias); DYB_RB(R, res); DYB_RB(R, y);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + wave;
  if (o >= O) return;
  const float* wr = w + (size_t)o * ldw;
  const int I4 = I >> 2;
  for (int b0 = 0; b0 < B; b0 += LIN_BT) {
    float acc[LIN_BT], addv[LIN_BT];
#pragma unroll
    for (int t = 0; t < LIN_BT; ++t) acc[t] = 0.f;
    // bias (+ residual) fetched up front by every lane (same address: one broadcast line), not by lane 0 after the
    // reduction - there it would be two more dependent round trips at the end of a 5 us kernel
    {
      const float bo = bias ? bias[o] : 0.f;
#pragma unroll
      for (int t = 0; t < LIN_BT; ++t) {
        const int bb = b0 + t < B ? b0 + t : B - 1;
        addv[t] = bo + (res ? res[(size_t)bb * ldres + o] : 0.f);
      }
    }
    // four weight pieces (+ the matching x pieces) in flight per round trip: the row stream is the
    // latency chain of this kernel at batch 1 (fc1: 9 pieces per lane)
    for (int i0 = lane; i0 < I4; i0 += 256) {
      // clamped unconditional loads (a per-lane `cond ? load : 0` is waited for load by load); the clamped repeats
      // are zeroed through the weight piece only
      float4 wv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i4 = i0 + 64 * j;
        wv[j] = *reinterpret_cast<const float4*>(wr + (size_t)(i4 < I4 ? i4 : I4 - 1) * 4);
      }
#pragma unroll
      for (int t = 0; t < LIN_BT; ++t) {
        if (b0 + t < B) {
          float4 xv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i4 = i0 + 64 * j;
            xv[j] = *reinterpret_cast<const float4*>(x + (size_t)(b0 + t) * ldx + (size_t)(i4 < I4 ? i4 : I4 - 1) * 4);
          }
          if (t == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (i0 + 64 * j >= I4) wv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[t] += (wv[j].x * xv[j].x + wv[j].y * xv[j].y) + (wv[j].z * xv[j].z + wv[j].w * xv[j].w);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < LIN_BT; ++t) {
      float s = dyb_wave_sum(acc[t]);
      if (lane == 0 && b0 + t < B) y[(size_t)(b0 + t) * ldy + o] = s + addv[t];
    }
  }
}

extern "C" int dyb_linear_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* res,
                              int ldres, float* y, int ldy, int B, int I, int O, hipStream_t st) {
  DYB_REQUIRE(x && w && y && B > 0 && O > 0, DYB_ERR_ARG);          // (bias may be NULL: a partial product added to `res`)
  DYB_REQUIRE(I % 4 == 0 && ldw % 4 == 0 && ldx % 4 == 0, DYB_ERR_UNSUPPORTED);
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(dyb_cdiv(O, 4), 1, R.n), dim3(256), 0, st, x, ldx, w, ldw, bias, res, ldres, y, ldy,
                     B, I, O, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// partial[s][b][i] = sum_{o in split s} dy[b][o] * W[o][i]
// grid (ceil(I/256), nsplit), block 256 = 64 float4-columns x 4 row lanes
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                            int ldw, float* __restrict__ partial, int B, int I, int O,
                                                            int rows_per_split, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, dy); DYB_RB(R, w); DYB_RB(R, partial);
  __shared__ float sm[4][64][4];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i4 = blockIdx.x * 64 + tx;
  const bool live = i4 * 4 < I;
  const int o0 = blockIdx.y * rows_per_split;
  int o1 = o0 + rows_per_split;
  if (o1 > O) o1 = O;
  for (int b0 = 0; b0 < B; b0 += LIN_BT) {
    float4 acc[LIN_BT];
#pragma unroll
    for (int t = 0; t < LIN_BT; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
      // four rows per trip, every load issued before the first use (rows past the end re-read the last row with a zero
      // gradient): the weight rows are cold L2 misses, one round trip per trip instead of one per row
      for (int ob = o0 + ty; ob < o1; ob += 16) {
        float4 wv[4];
        float d[4][LIN_BT];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int o = ob + 4 * j, oc = o < o1 ? o : o1 - 1;
          wv[j] = *reinterpret_cast<const float4*>(w + (size_t)oc * ldw + (size_t)i4 * 4);
#pragma unroll
          for (int t = 0; t < LIN_BT; ++t) d[j][t] = dy[(size_t)(b0 + t < B ? b0 + t : B - 1) * lddy + oc];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool rok = ob + 4 * j < o1;
#pragma unroll
          for (int t = 0; t < LIN_BT; ++t) {
            const float dd = (rok && b0 + t < B) ? d[j][t] : 0.f;
            acc[t].x += dd * wv[j].x; acc[t].y += dd * wv[j].y; acc[t].z += dd * wv[j].z; acc[t].w += dd * wv[j].w;
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < LIN_BT; ++t) {
      __syncthreads();
      sm[ty][tx][0] = acc[t].x; sm[ty][tx][1] = acc[t].y; sm[ty][tx][2] = acc[t].z; sm[ty][tx][3] = acc[t].w;
      __syncthreads();
      if (ty == 0 && live && b0 + t < B) {
        float4 r;
        r.x = (sm[0][tx][0] + sm[1][tx][0]) + (sm[2][tx][0] + sm[3][tx][0]);
        r.y = (sm[0][tx][1] + sm[1][tx][1]) + (sm[2][tx][1] + sm[3][tx][1]);
        r.z = (sm[0][tx][2] + sm[1][tx][2]) + (sm[2][tx][2] + sm[3][tx][2]);
        r.w = (sm[0][tx][3] + sm[1][tx][3]) + (sm[2][tx][3] + sm[3][tx][3]);
        *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.y * B + (b0 + t)) * I + (size_t)i4 * 4) = r;
      }
    }
  }
}

// dst[b][i] = sum_s partial[s][b][i] (+ add[b][i]); optional second destination for columns >= split_col:
//   i <  split_col -> dstA[b*ldA + i]           (accumulate into dstA when accA != 0)
//   i >= split_col -> dstB[b*ldB + i-split_col] = value + addB[b*ldaddB + i-split_col]
__global__ __launch_bounds__(256) void linear_dx_fold_kernel(const float* __restrict__ partial, int nsplit, int B, int I,
                                                             float* __restrict__ dstA, int ldA, int accA, int split_col,
                                                             float* __restrict__ dstB, int ldB, const float* __restrict__ addB,
                                                             int ldaddB, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, partial); DYB_RB(R, dstA); DYB_RB(R, dstB); DYB_RB(R, addB);
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * I) return;
  int b = idx / I, i = idx % I;
  float s = 0.f;
  for (int z0 = 0; z0 < nsplit; z0 += 8) {         // eight loads in flight, not a dependent chain of nsplit
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (z0 + j < nsplit) ? partial[((size_t)(z0 + j) * B + b) * I + i] : 0.f;
    s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
  if (i < split_col) {
    float* d = dstA + (size_t)b * ldA + i;
    *d = accA ? (*d + s) : s;
  } else {
    int j = i - split_col;
    dstB[(size_t)b * ldB + j] = s + (addB ? addB[(size_t)b * ldaddB + j] : 0.f);
  }
}

static int lin_split(int I, int O) {
  int cols = dyb_cdiv(I, 256);
  int s = dyb_cdiv(256, cols);           // ~256 workgroups
  int cap = O / 16 > 1 ? O / 16 : 1;     // >= 16 rows per split
  return s < cap ? s : cap;
}
extern "C" size_t dyb_linear_bwd_workspace_bytes(int B, int I, int O) {
  return (size_t)lin_split(I, O) * B * I * sizeof(float);
}
// dx = dy @ W, routed: columns [0,split_col) -> dstA (+= when accA), columns [split_col,I) -> dstB (+ addB).
// Pass split_col = I and dstB = NULL for a plain dx.
extern "C" int dyb_linear_bwd_dx(const float* dy, int lddy, const float* w, int ldw, int B, int I, int O, float* dstA,
                                 int ldA, int accA, int split_col, float* dstB, int ldB, const float* addB, int ldaddB,
                                 void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(dy && w && dstA && ws && B > 0, DYB_ERR_ARG);
  DYB_REQUIRE(I % 4 == 0 && ldw % 4 == 0, DYB_ERR_UNSUPPORTED);
  DYB_REQUIRE(split_col == I || dstB, DYB_ERR_ARG);
  int ns = lin_split(I, O);
  DYB_REQUIRE(ws_bytes >= (size_t)ns * B * I * sizeof(float), DYB_ERR_WORKSPACE);
  int rps = dyb_cdiv(O, ns);
  ns = dyb_cdiv(O, rps);
  float* partial = reinterpret_cast<float*>(ws);
  const DybRep& R = dyb_rep_current();
  hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(dyb_cdiv(I, 256), ns, R.n), dim3(256), 0, st, dy, lddy, w, ldw, partial, B, I,
                     O, rps, R);
  DYB_CHECK_LAUNCH();
  hipLaunchKernelGGL(linear_dx_fold_kernel, dim3(dyb_cdiv(B * I, 256), 1, R.n), dim3(256), 0, st, (const float*)partial, ns, B,
                     I, dstA, ldA, accA, split_col, dstB, ldB, addB, ldaddB, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// dW[o][i] = sum_{t<T} sum_b dy_t[b][o] * x_t[b][i];  db[o] = sum_t sum_b dy_t[b][o]
struct OuterArgs {
  const float* dy[4];
  const float* x[4];
  int lddy[4], ldx[4];
};
// grid (ceil(I4/64), ceil(O/32)), block 256: 64 float4-columns x 4 row lanes, each thread OUTER_ROWS = 8 rows (o = 32 by + ty + 4 j).
// One output float4 per thread (4 rows per workgroup) made this kernel workgroup-dispatch-bound: 2304 workgroups for fc1 took
// 21.7 us for one sequence and 73.7 k workgroups 815 us for 32 (~90 workgroups/us, 0.36 TB/s of a pure-write kernel;
// profiles/r02_s5_kernel_stats_S32.csv).  Eight rows per thread = 8x fewer workgroups, x loaded once per (t, b) instead of
// once per row; every output element still sums its (t, b) terms in the same order: results unchanged bit for bit.
#define OUTER_ROWS 8
// The weight update of the calling thread's scope (DybWgradUpdateScope, round 6) applied to the finished gradient element instead of storing
// it: kind 1 = MAML fast weights p_next = p_cur - lr * g, kind 2 = Adam on (p, m, v) in place (dyb_adam_one) - the regressor's fc1 / fc2 /
// decoder matrices are 13 % of the parameters, each of their gradient elements is complete in ONE work-item here.  Biases keep the streaming pass.
struct OuterUpd {
  int kind;
  const float* p_cur;
  float *p_next, *m, *v;
  const float* sc;
  float lr, b1, b2, eps;
};
__global__ __launch_bounds__(256) void linear_outer_kernel(OuterArgs a, int T, int B, int I, int O, float* __restrict__ dw,
                                                           int ldw, float* __restrict__ db, OuterUpd u, DybRep R) {
  DYB_REP_PROLOGUE(R);
  DYB_RB(R, dw); DYB_RB(R, db);
  if (u.kind) { u.p_cur = dyb_rb(u.p_cur, R, dyb_rep); u.p_next = dyb_rb(u.p_next, R, dyb_rep); u.m = dyb_rb(u.m, R, dyb_rep);
                u.v = dyb_rb(u.v, R, dyb_rep); u.sc = dyb_rb(u.sc, R, dyb_rep); }
  if (dyb_rep) {
#pragma unroll
    for (int t = 0; t < 4; ++t) { a.dy[t] = dyb_rb(a.dy[t], R, dyb_rep); a.x[t] = dyb_rb(a.x[t], R, dyb_rep); }
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i4 = blockIdx.x * 64 + tx;
  const int o0 = blockIdx.y * (4 * OUTER_ROWS) + ty;
  const bool live = i4 * 4 < I;
  float4 acc[OUTER_ROWS];
  float bs[OUTER_ROWS];
#pragma unroll
  for (int j = 0; j < OUTER_ROWS; ++j) { acc[j] = make_float4(0.f, 0.f, 0.f, 0.f); bs[j] = 0.f; }
#pragma unroll
  for (int t = 0; t < 4; ++t) {                        // constant indices into the argument block: it stays in registers
    if (t >= T) break;
    for (int b = 0; b < B; ++b) {
      const float4 xv = *reinterpret_cast<const float4*>(a.x[t] + (size_t)b * a.ldx[t] + (size_t)(live ? i4 : 0) * 4);
      const float* dyr = a.dy[t] + (size_t)b * a.lddy[t];
#pragma unroll
      for (int j = 0; j < OUTER_ROWS; ++j) {
        const int o = o0 + 4 * j;
        const float d = dyr[o < O ? o : O - 1];               // clamped: rows past the end are never stored
        bs[j] += d;
        acc[j].x += d * xv.x; acc[j].y += d * xv.y; acc[j].z += d * xv.z; acc[j].w += d * xv.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < OUTER_ROWS; ++j) {
    const int o = o0 + 4 * j;
    if (o >= O) continue;
    const size_t e = (size_t)o * ldw + (size_t)i4 * 4;
    if (live) {
      if (u.kind == 1) {
        float4 p = *reinterpret_cast<const float4*>(u.p_cur + e);
        p.x = dyb_fast_one(p.x, acc[j].x, u.lr); p.y = dyb_fast_one(p.y, acc[j].y, u.lr);
        p.z = dyb_fast_one(p.z, acc[j].z, u.lr); p.w = dyb_fast_one(p.w, acc[j].w, u.lr);
        *reinterpret_cast<float4*>(u.p_next + e) = p;
      } else if (u.kind == 2) {
        float4 p = *reinterpret_cast<const float4*>(u.p_cur + e), m = *reinterpret_cast<const float4*>(u.m + e),
               v = *reinterpret_cast<const float4*>(u.v + e);
        const float ss = u.sc[0], bc = u.sc[1];
        dyb_adam_one(p.x, acc[j].x, m.x, v.x, u.b1, u.b2, ss, bc, u.eps);
        dyb_adam_one(p.y, acc[j].y, m.y, v.y, u.b1, u.b2, ss, bc, u.eps);
        dyb_adam_one(p.z, acc[j].z, m.z, v.z, u.b1, u.b2, ss, bc, u.eps);
        dyb_adam_one(p.w, acc[j].w, m.w, v.w, u.b1, u.b2, ss, bc, u.eps);
        *reinterpret_cast<float4*>(u.p_next + e) = p;
        *reinterpret_cast<float4*>(u.m + e) = m;
        *reinterpret_cast<float4*>(u.v + e) = v;
      } else {
        *reinterpret_cast<float4*>(dw + e) = acc[j];
      }
    }
    if (blockIdx.x == 0 && tx == 0) db[o] = bs[j];
  }
}
static bool switches_linear_fuse() {
  static const int on = [] { const char* e = getenv("DYB_FUSE_LINEAR"); return e ? atoi(e) : 1; }();
  return on != 0;
}
extern "C" int dyb_linear_bwd_dw(const float* const* dys, const int* lddys, const float* const* xs, const int* ldxs,
                                 int T, int B, int I, int O, float* dw, int ldw, float* db, hipStream_t st) {
  DYB_REQUIRE(dys && xs && dw && db && T >= 1 && T <= 4, DYB_ERR_ARG);
  DYB_REQUIRE(I % 4 == 0 && ldw % 4 == 0, DYB_ERR_UNSUPPORTED);
  OuterArgs a{};
  for (int t = 0; t < T; ++t) {
    a.dy[t] = dys[t]; a.x[t] = xs[t]; a.lddy[t] = lddys[t]; a.ldx[t] = ldxs[t];
    DYB_REQUIRE(a.ldx[t] % 4 == 0, DYB_ERR_UNSUPPORTED);
  }
  const DybRep& R = dyb_rep_current();
  // a weight-update scope of the calling thread (stepper: a lower level's fast-weight step / the outer level's Adam): the matrix [O][ldw]
  // is one contiguous tensor of the arena (I == ldw for fc1 / fc2 / the decoder) - updated here, its span reported like a convolution's
  OuterUpd u{};
  const DybWgradUpdate& W = dyb_wgrad_update_current();
  if (W.grads && I == ldw && switches_linear_fuse()) {
    const char *lo = reinterpret_cast<const char*>(W.grads), *o = reinterpret_cast<const char*>(dw);
    const size_t cnt = (size_t)O * ldw;
    if (o >= lo && o + cnt * sizeof(float) <= lo + W.bytes) {
      const size_t off = (size_t)(o - lo) / sizeof(float);
      u.kind = W.adam_m ? 2 : 1;
      u.p_cur = W.p_cur + off; u.p_next = W.p_next + off; u.lr = W.lr;
      if (W.adam_m) { u.m = W.adam_m + off; u.v = W.adam_v + off; u.sc = W.adam_sc; u.b1 = W.b1; u.b2 = W.b2; u.eps = W.eps; }
      if (W.spans) W.spans->push_back(DybSpan{off, cnt});
    }
  }
  hipLaunchKernelGGL(linear_outer_kernel, dim3(dyb_cdiv(I / 4, 64), dyb_cdiv(O, 4 * OUTER_ROWS), R.n), dim3(256), 0, st, a, T, B, I, O, dw,
                     ldw, db, u, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
