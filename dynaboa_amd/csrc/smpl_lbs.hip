// SMPL linear-blend-skinning forward and backward (6890 vertices x 24 joints), and the 6-D ->
// rotation-matrix map that feeds it.
//
// Replaces smplx.lbs.lbs + smplx VertexJointSelector + the reference wrapper's extra-joint
// regression and 49-joint gather (reference model/smpl.py:25-37; algorithm restated in SURVEY
// Appendix B - smplx itself is not vendored by the reference), and utils/geometry.py:47-61.
//
// Work split (HBM/L2-bound: posedirs 17.1 MB dominates, everything else < 1 MB):
//   lbs_pose   (1 workgroup / sample): rest joints J = J_template + J_shapedirs*beta (the
//              24x6890 joint regressor is folded into two small tables on the host, exact up to
//              fp32 re-association), pose feature R[1:]-I, the 24-step kinematic chain run
//              12 lanes wide, skinning transforms A = [G_R | G_t - G_R J].
//   lbs_skin   (64 vertices / workgroup, 4 waves): A, pose feature and betas staged in LDS; the 207
//              pose-corrective rows of posedirs (coalesced segments) are split over the 4 waves
//              and folded through LDS; wave 0 adds blend shapes (10), blends the transform from
//              the transposed skin-weight table, writes the vertex and wave-butterfly partial
//              sums of the 9 regressed extra joints.
//   lbs_joints (1 workgroup / sample): fold partials, gather the 49 output joints.
// Backward mirrors it: scatter d_joints49 -> 54 sources; per-vertex sweep producing partial
// reductions of dA (288), d pose-feature (207), d beta (10) through a 63-shuffle transpose-reduce
// per 64 quantities; chain backward 12 lanes wide.
#include "dyb_common.h"

#define NV 6890
#define NJ 24
#define NPF 207
#define NPF_PAD 208
#define NB 10
#define NEXTRA 9
#define NVJ 21
#define NJ54 54
#define NJ49 49
#define LBS_VB 64                               // vertices per workgroup
#define LBS_NBLK ((NV + LBS_VB - 1) / LBS_VB)   // 108
#define LBS_PQ 52                               // pose-corrective rows per wave (4 waves cover 207)
#define NRED (NJ * 12 + NPF + NB)               // 505 partial reductions per workgroup

struct SmplTables {
  const float* v_template;     // [NV][3]
  const float* shapedirs;      // [NV*3][10]
  const float* posedirs;       // [207][NV*3]
  const float* weights_t;      // [24][NV]   (lbs_weights transposed)
  const float* j_template;     // [24][3]    J_regressor @ v_template
  const float* j_shapedirs;    // [72][10]   J_regressor @ shapedirs
  const float* j_extra;        // [9][NV]    J_regressor_extra
  const int* parents;          // [24]
  const int* vertex_joint_ids; // [21]
  const int* joint_map;        // [49] into [24 | 21 | 9]
};

// ------------------------------------------------------------------------------------------
// rot6d <-> rotmat   (reference utils/geometry.py:47-61; x.view(-1,3,2): a1 = x[0,2,4], a2 = x[1,3,5])
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void rot6d_one(const float* x, float* R) {
  float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
  float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
  float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
  float s = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  float u[3] = {a2[0] - s * b1[0], a2[1] - s * b1[1], a2[2] - s * b1[2]};
  float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
  float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
  float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
  for (int r = 0; r < 3; ++r) {
    R[r * 3 + 0] = b1[r];
    R[r * 3 + 1] = b2[r];
    R[r * 3 + 2] = b3[r];
  }
}
__global__ __launch_bounds__(64) void rot6d_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ R, int n, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, x); DYB_RB(Rp, R);
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  int b = i / NJ, j = i % NJ;
  rot6d_one(x + (size_t)b * ldx + j * 6, R + (size_t)i * 9);
}
__global__ __launch_bounds__(64) void rot6d_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dR,
                                                       float* __restrict__ dx, int lddx, int n, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, x); DYB_RB(Rp, dR); DYB_RB(Rp, dx);
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  int b = i / NJ, j = i % NJ;
  const float* xi = x + (size_t)b * ldx + j * 6;
  const float* g = dR + (size_t)i * 9;
  float a1[3] = {xi[0], xi[2], xi[4]}, a2[3] = {xi[1], xi[3], xi[5]};
  float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
  float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
  float s = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  float u[3] = {a2[0] - s * b1[0], a2[1] - s * b1[1], a2[2] - s * b1[2]};
  float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
  float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
  float G1[3] = {g[0], g[3], g[6]}, G2[3] = {g[1], g[4], g[7]}, G3[3] = {g[2], g[5], g[8]};
  // b3 = b1 x b2
  float gb1[3] = {G1[0] + (b2[1] * G3[2] - b2[2] * G3[1]), G1[1] + (b2[2] * G3[0] - b2[0] * G3[2]),
                  G1[2] + (b2[0] * G3[1] - b2[1] * G3[0])};
  float gb2[3] = {G2[0] + (G3[1] * b1[2] - G3[2] * b1[1]), G2[1] + (G3[2] * b1[0] - G3[0] * b1[2]),
                  G2[2] + (G3[0] * b1[1] - G3[1] * b1[0])};
  // b2 = u/|u|
  float d2 = b2[0] * gb2[0] + b2[1] * gb2[1] + b2[2] * gb2[2];
  float gu[3] = {(gb2[0] - b2[0] * d2) / n2, (gb2[1] - b2[1] * d2) / n2, (gb2[2] - b2[2] * d2) / n2};
  // u = a2 - (b1.a2) b1
  float bu = b1[0] * gu[0] + b1[1] * gu[1] + b1[2] * gu[2];
  float ga2[3] = {gu[0] - bu * b1[0], gu[1] - bu * b1[1], gu[2] - bu * b1[2]};
  for (int k = 0; k < 3; ++k) gb1[k] += -bu * a2[k] - s * gu[k];
  // b1 = a1/|a1|
  float d1 = b1[0] * gb1[0] + b1[1] * gb1[1] + b1[2] * gb1[2];
  float ga1[3] = {(gb1[0] - b1[0] * d1) / n1, (gb1[1] - b1[1] * d1) / n1, (gb1[2] - b1[2] * d1) / n1};
  float* o = dx + (size_t)b * lddx + j * 6;
  o[0] = ga1[0]; o[2] = ga1[1]; o[4] = ga1[2];
  o[1] = ga2[0]; o[3] = ga2[1]; o[5] = ga2[2];
}
extern "C" int dyb_rot6d_fwd(const float* x6, int ldx, float* rotmat, int B, hipStream_t st) {
  DYB_REQUIRE(x6 && rotmat && B > 0, DYB_ERR_ARG);
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(rot6d_fwd_kernel, dim3(dyb_cdiv(B * NJ, 64), 1, Rp.n), dim3(64), 0, st, x6, ldx, rotmat, B * NJ, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
extern "C" int dyb_rot6d_bwd(const float* x6, int ldx, const float* drotmat, float* dx6, int lddx, int B,
                             hipStream_t st) {
  DYB_REQUIRE(x6 && drotmat && dx6 && B > 0, DYB_ERR_ARG);
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(rot6d_bwd_kernel, dim3(dyb_cdiv(B * NJ, 64), 1, Rp.n), dim3(64), 0, st, x6, ldx, drotmat, dx6, lddx, B * NJ, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// axis-angle -> rotation matrix as smplx.lbs.batch_rodrigues does it (angle = ||r + 1e-8||,
// R = I + sin K + (1-cos) K^2): the pose2rot=True path of SMPL.forward, used for the ground-truth
// meshes of the metric path (reference dynaboa_benchmark.py:221-227,242).  No gradient needed.
__global__ __launch_bounds__(64) void rodrigues_kernel(const float* __restrict__ aa, float* __restrict__ R, int n, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, aa); DYB_RB(Rp, R);
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float x = aa[i * 3], y = aa[i * 3 + 1], z = aa[i * 3 + 2];
  float ex = x + 1e-8f, ey = y + 1e-8f, ez = z + 1e-8f;
  float ang = sqrtf(ex * ex + ey * ey + ez * ez);
  float kx = x / ang, ky = y / ang, kz = z / ang;
  float s = sinf(ang), c1 = 1.f - cosf(ang);
  float* o = R + (size_t)i * 9;
  // K = [[0,-kz,ky],[kz,0,-kx],[-ky,kx,0]];  K^2 = k k^T - |k|^2 I
  float k2 = kx * kx + ky * ky + kz * kz;
  o[0] = 1.f + c1 * (kx * kx - k2); o[1] = -s * kz + c1 * kx * ky;   o[2] = s * ky + c1 * kx * kz;
  o[3] = s * kz + c1 * kx * ky;     o[4] = 1.f + c1 * (ky * ky - k2); o[5] = -s * kx + c1 * ky * kz;
  o[6] = -s * ky + c1 * kx * kz;    o[7] = s * kx + c1 * ky * kz;     o[8] = 1.f + c1 * (kz * kz - k2);
}
extern "C" int dyb_rodrigues_fwd(const float* aa, float* rotmat, int n, hipStream_t st) {
  DYB_REQUIRE(aa && rotmat && n > 0, DYB_ERR_ARG);
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(rodrigues_kernel, dim3(dyb_cdiv(n, 64), 1, Rp.n), dim3(64), 0, st, aa, rotmat, n, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// ------------------------------------------------------------------------------------------
// LBS forward
// ------------------------------------------------------------------------------------------
// saved per sample: A[24][12] (rows [G_R | A_t]), Jp[24][3] (posed joints = G_t), J[24][3], pf[208]
__global__ __launch_bounds__(128) void lbs_pose_kernel(SmplTables T, const float* __restrict__ betas, int ldb,
                                                       const float* __restrict__ rot, float* __restrict__ A,
                                                       float* __restrict__ Jp, float* __restrict__ Jrest,
                                                       float* __restrict__ pf, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, betas); DYB_RB(Rp, rot); DYB_RB(Rp, A); DYB_RB(Rp, Jp); DYB_RB(Rp, Jrest); DYB_RB(Rp, pf);
  __shared__ float sJ[NJ * 3], sGR[NJ * 9], sGt[NJ * 3], sR[NJ * 9];
  __shared__ int sPar[NJ];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* be = betas + (size_t)b * ldb;
  const float* R = rot + (size_t)b * NJ * 9;
  if (t < NJ * 3) {
    float s = T.j_template[t];
    for (int l = 0; l < NB; ++l) s += T.j_shapedirs[t * NB + l] * be[l];
    sJ[t] = s;
    Jrest[(size_t)b * NJ * 3 + t] = s;
  }
  if (t < NJ) sPar[t] = T.parents[t];
  for (int i = t; i < NJ * 9; i += 128) sR[i] = R[i];
  for (int i = t; i < NPF_PAD; i += 128) {
    float v = 0.f;
    if (i < NPF) {
      int e = i % 9;
      v = R[9 + i] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
    pf[(size_t)b * NPF_PAD + i] = v;
  }
  __syncthreads();
  if (t < 9) sGR[t] = sR[t];
  if (t >= 9 && t < 12) sGt[t - 9] = sJ[t - 9];
  __syncthreads();
  for (int i = 1; i < NJ; ++i) {
    const int p = sPar[i];
    if (t < 9) {
      int r = t / 3, c = t % 3;
      sGR[i * 9 + t] = sGR[p * 9 + r * 3 + 0] * sR[i * 9 + 0 + c] + sGR[p * 9 + r * 3 + 1] * sR[i * 9 + 3 + c] +
                       sGR[p * 9 + r * 3 + 2] * sR[i * 9 + 6 + c];
    } else if (t < 12) {
      int r = t - 9;
      float rel0 = sJ[i * 3 + 0] - sJ[p * 3 + 0], rel1 = sJ[i * 3 + 1] - sJ[p * 3 + 1], rel2 = sJ[i * 3 + 2] - sJ[p * 3 + 2];
      sGt[i * 3 + r] = sGR[p * 9 + r * 3 + 0] * rel0 + sGR[p * 9 + r * 3 + 1] * rel1 + sGR[p * 9 + r * 3 + 2] * rel2 +
                       sGt[p * 3 + r];
    }
    __syncthreads();
  }
  if (t < NJ * 3) {
    int i = t / 3, r = t % 3;
    float at = sGt[t] - (sGR[i * 9 + r * 3 + 0] * sJ[i * 3 + 0] + sGR[i * 9 + r * 3 + 1] * sJ[i * 3 + 1] +
                         sGR[i * 9 + r * 3 + 2] * sJ[i * 3 + 2]);
    float* a = A + ((size_t)b * NJ + i) * 12 + r * 4;
    a[0] = sGR[i * 9 + r * 3 + 0];
    a[1] = sGR[i * 9 + r * 3 + 1];
    a[2] = sGR[i * 9 + r * 3 + 2];
    a[3] = at;
    Jp[(size_t)b * NJ * 3 + t] = sGt[t];
  }
}

// grid (108, B), block 256 = 4 waves x 64 vertices.  The 207-row pose-corrective sum (the only
// real traffic: posedirs, 17.1 MB) is split over the 4 waves and folded through LDS; wave 0 then
// finishes the vertex (blend shapes, blended transform, output, extra-joint partial sums).
__global__ __launch_bounds__(256) void lbs_skin_kernel(SmplTables T, const float* __restrict__ betas, int ldb,
                                                       const float* __restrict__ A, const float* __restrict__ pf,
                                                       float* __restrict__ verts, float* __restrict__ vposed,
                                                       float* __restrict__ extra_part, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, betas); DYB_RB(Rp, A); DYB_RB(Rp, pf); DYB_RB(Rp, verts); DYB_RB(Rp, vposed); DYB_RB(Rp, extra_part);
  __shared__ float sA[NJ * 12], sPf[NPF_PAD], sBe[NB], sPart[4][LBS_VB][3];
  const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  {
    // the three small tables are fetched in ONE round trip (clamped addresses, all loads before the first LDS store);
    // three staging loops in a row are three dependent trips in front of the sweep
    const float a0 = A[(size_t)b * NJ * 12 + t];
    const float a1 = A[(size_t)b * NJ * 12 + (t + 256 < NJ * 12 ? t + 256 : 0)];
    const float p0 = pf[(size_t)b * NPF_PAD + (t < NPF_PAD ? t : 0)];
    const float be = betas[(size_t)b * ldb + (t < NB ? t : 0)];
    sA[t] = a0;
    if (t + 256 < NJ * 12) sA[t + 256] = a1;
    if (t < NPF_PAD) sPf[t] = p0;
    if (t < NB) sBe[t] = be;
  }
  __syncthreads();
  const int v = blockIdx.x * LBS_VB + lane;
  const bool live = v < NV;
  const int vv = live ? v : 0;
  {
    const float* pd = T.posedirs + (size_t)vv * 3;
    const int p0i = wave * LBS_PQ;
    int p1i = p0i + LBS_PQ;
    if (p1i > NPF) p1i = NPF;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, r0 = 0.f, r1 = 0.f, r2 = 0.f;
    // 13 rows = 39 independent loads issued before the first use: the sweep is latency-bound
    for (int p = p0i; p < p1i; p += 13) {
      float x[13][3];
#pragma unroll
      for (int u = 0; u < 13; ++u) {
        const int pp = (p + u < p1i) ? (p + u) : p;       // clamp; weight below is 0 for the clamped rows
        const float* ra = pd + (size_t)pp * (NV * 3);
        x[u][0] = ra[0]; x[u][1] = ra[1]; x[u][2] = ra[2];
      }
#pragma unroll
      for (int u = 0; u < 13; ++u) {
        const float f = (p + u < p1i) ? sPf[p + u] : 0.f;
        if (u & 1) { r0 += f * x[u][0]; r1 += f * x[u][1]; r2 += f * x[u][2]; }
        else { q0 += f * x[u][0]; q1 += f * x[u][1]; q2 += f * x[u][2]; }
      }
    }
    sPart[wave][lane][0] = q0 + r0;
    sPart[wave][lane][1] = q1 + r1;
    sPart[wave][lane][2] = q2 + r2;
  }
  __syncthreads();
  if (wave != 0) return;
  float vp[3], out[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float s = T.v_template[vv * 3 + c];
    const float* sd = T.shapedirs + (size_t)(vv * 3 + c) * NB;
#pragma unroll
    for (int l = 0; l < NB; ++l) s += sd[l] * sBe[l];
    vp[c] = s + ((sPart[0][lane][c] + sPart[1][lane][c]) + (sPart[2][lane][c] + sPart[3][lane][c]));
  }
  float Tm[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) Tm[e] = 0.f;
  for (int j = 0; j < NJ; ++j) {
    float w = T.weights_t[(size_t)j * NV + vv];
#pragma unroll
    for (int e = 0; e < 12; ++e) Tm[e] += w * sA[j * 12 + e];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) out[c] = Tm[c * 4 + 0] * vp[0] + Tm[c * 4 + 1] * vp[1] + Tm[c * 4 + 2] * vp[2] + Tm[c * 4 + 3];
  if (live) {
    size_t o = ((size_t)b * NV + v) * 3;
    verts[o] = out[0]; verts[o + 1] = out[1]; verts[o + 2] = out[2];
    vposed[o] = vp[0]; vposed[o + 1] = vp[1]; vposed[o + 2] = vp[2];
  }
  float* ep = extra_part + ((size_t)b * LBS_NBLK + blockIdx.x) * (NEXTRA * 3);
  for (int e = 0; e < NEXTRA; ++e) {
    float x = live ? T.j_extra[(size_t)e * NV + v] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float s = dyb_wave_sum(x * out[c]);
      if (lane == 0) ep[e * 3 + c] = s;
    }
  }
}

__global__ __launch_bounds__(64) void lbs_joints_kernel(SmplTables T, const float* __restrict__ extra_part,
                                                        const float* __restrict__ Jp, const float* __restrict__ verts,
                                                        float* __restrict__ joints49, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, extra_part); DYB_RB(Rp, Jp); DYB_RB(Rp, verts); DYB_RB(Rp, joints49);
  __shared__ float sE[NEXTRA * 3];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < NEXTRA * 3) {
    float s = 0.f;
    for (int k = 0; k < LBS_NBLK; ++k) s += extra_part[((size_t)b * LBS_NBLK + k) * (NEXTRA * 3) + t];
    sE[t] = s;
  }
  __syncthreads();
  if (t < NJ49) {
    int jm = T.joint_map[t];
    for (int c = 0; c < 3; ++c) {
      float v;
      if (jm < NJ) v = Jp[((size_t)b * NJ + jm) * 3 + c];
      else if (jm < NJ + NVJ) v = verts[((size_t)b * NV + T.vertex_joint_ids[jm - NJ]) * 3 + c];
      else v = sE[(jm - NJ - NVJ) * 3 + c];
      joints49[((size_t)b * NJ49 + t) * 3 + c] = v;
    }
  }
}

static SmplTables make_tables(const float* const* f, const int* const* i) {
  SmplTables T;
  T.v_template = f[0]; T.shapedirs = f[1]; T.posedirs = f[2]; T.weights_t = f[3];
  T.j_template = f[4]; T.j_shapedirs = f[5]; T.j_extra = f[6];
  T.parents = i[0]; T.vertex_joint_ids = i[1]; T.joint_map = i[2];
  return T;
}

extern "C" size_t dyb_lbs_saved_floats(int B) {
  // A[24*12] + Jp[72] + J[72] + pf[208] + vposed[NV*3] + extra_part[27*27]
  return (size_t)B * (NJ * 12 + NJ * 3 + NJ * 3 + NPF_PAD + (size_t)NV * 3 + LBS_NBLK * NEXTRA * 3);
}
extern "C" size_t dyb_lbs_bwd_workspace_bytes(int B) {
  return (size_t)B * (NJ54 * 3 + (size_t)LBS_NBLK * NRED) * sizeof(float);
}
struct LbsSaved {
  float *A, *Jp, *J, *pf, *vposed, *extra_part;
};
static LbsSaved carve_saved(float* base, int B) {
  LbsSaved s;
  s.A = base; base += (size_t)B * NJ * 12;
  s.Jp = base; base += (size_t)B * NJ * 3;
  s.J = base; base += (size_t)B * NJ * 3;
  s.pf = base; base += (size_t)B * NPF_PAD;
  s.vposed = base; base += (size_t)B * NV * 3;
  s.extra_part = base;
  return s;
}

// tables_f: {v_template, shapedirs, posedirs, weights_t, j_template, j_shapedirs, j_extra}
// tables_i: {parents, vertex_joint_ids, joint_map}
extern "C" int dyb_lbs_fwd(const float* const* tables_f, const int* const* tables_i, const float* betas, int ldb,
                           const float* rotmat, float* verts, float* joints49, float* saved, int B, hipStream_t st) {
  DYB_REQUIRE(tables_f && tables_i && betas && rotmat && verts && joints49 && saved && B > 0, DYB_ERR_ARG);
  SmplTables T = make_tables(tables_f, tables_i);
  LbsSaved s = carve_saved(saved, B);
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(lbs_pose_kernel, dim3(B, 1, Rp.n), dim3(128), 0, st, T, betas, ldb, rotmat, s.A, s.Jp, s.J, s.pf, Rp);
  DYB_CHECK_LAUNCH();
  hipLaunchKernelGGL(lbs_skin_kernel, dim3(LBS_NBLK, B, Rp.n), dim3(256), 0, st, T, betas, ldb, (const float*)s.A,
                     (const float*)s.pf, verts, s.vposed, s.extra_part, Rp);
  DYB_CHECK_LAUNCH();
  hipLaunchKernelGGL(lbs_joints_kernel, dim3(B, 1, Rp.n), dim3(64), 0, st, T, (const float*)s.extra_part, (const float*)s.Jp,
                     (const float*)verts, joints49, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// ------------------------------------------------------------------------------------------
// LBS backward
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lbs_bwd_scatter_kernel(SmplTables T, const float* __restrict__ dj49,
                                                             float* __restrict__ dj54, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, dj49); DYB_RB(Rp, dj54);
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= NJ54) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int q = 0; q < NJ49; ++q) {
    if (T.joint_map[q] == t) {
      const float* g = dj49 + ((size_t)b * NJ49 + q) * 3;
      s0 += g[0]; s1 += g[1]; s2 += g[2];
    }
  }
  float* o = dj54 + ((size_t)b * NJ54 + t) * 3;
  o[0] = s0; o[1] = s1; o[2] = s2;
}

// Sum 64 per-lane quantities across the 64 lanes of a wave with 63 shuffles (a butterfly that
// halves the live quantities at every stage) instead of 64 x 6: on return lane l holds the
// wave-wide sum of quantity l.
__device__ __forceinline__ float transpose_reduce64(float (&v)[64], int lane) {
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    bool up = (lane & 32) != 0;
    float send = up ? v[i] : v[i + 32], keep = up ? v[i + 32] : v[i];
    v[i] = keep + __shfl_xor(send, 32);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    bool up = (lane & 16) != 0;
    float send = up ? v[i] : v[i + 16], keep = up ? v[i + 16] : v[i];
    v[i] = keep + __shfl_xor(send, 16);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    bool up = (lane & 8) != 0;
    float send = up ? v[i] : v[i + 8], keep = up ? v[i + 8] : v[i];
    v[i] = keep + __shfl_xor(send, 8);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bool up = (lane & 4) != 0;
    float send = up ? v[i] : v[i + 4], keep = up ? v[i + 4] : v[i];
    v[i] = keep + __shfl_xor(send, 4);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    bool up = (lane & 2) != 0;
    float send = up ? v[i] : v[i + 2], keep = up ? v[i + 2] : v[i];
    v[i] = keep + __shfl_xor(send, 2);
  }
  {
    bool up = (lane & 1) != 0;
    float send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
    v[0] = keep + __shfl_xor(send, 1);
  }
  return v[0];
}

// quantities [BASE, BASE+64) of the dA block: q = j*12 + r*4 + c  ->  w_j * dv[r] * (c < 3 ? vp[c] : 1)
template <int BASE>
__device__ __forceinline__ void fill_dA(float (&vals)[64], const float (&w)[NJ], const float (&dv)[3], const float (&vp)[3]) {
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int q = BASE + i;
    if (q < NJ * 12) {
      const int j = q / 12, r = (q % 12) / 4, c = q % 4;
      vals[i] = w[j] * dv[r] * (c < 3 ? vp[c] : 1.f);
    } else {
      vals[i] = 0.f;
    }
  }
}
// quantities [BASE, BASE+64) of the [207 pose-feature | 10 beta] block
template <int BASE>
__device__ __forceinline__ void fill_dpf(float (&vals)[64], const float* __restrict__ pd, const float* __restrict__ sd,
                                         const float (&dvp)[3], bool live) {
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int q = BASE + i;
    // unconditional loads (dead lanes read vertex 0 and are masked through `lv`): a load under `if (live)` is waited
    // for on the spot, which made this sweep ~200 dependent round trips (tools/isa_scan.py)
    const float lv = live ? 1.f : 0.f;
    float x = 0.f;
    if (q < NPF) {
      const float* row = pd + (size_t)q * (NV * 3);
      x = (row[0] * dvp[0] + row[1] * dvp[1] + row[2] * dvp[2]) * lv;
    } else if (q < NPF + NB) {
      const int l = q - NPF;
      x = (sd[l] * dvp[0] + sd[NB + l] * dvp[1] + sd[2 * NB + l] * dvp[2]) * lv;
    }
    vals[i] = x;
  }
}

// grid (108, B), block 64 (one wave, one vertex per lane).  Per-workgroup partial sums, layout
// [0,288) dA, [288,495) d pose-feature, [495,505) d beta.
__global__ __launch_bounds__(64) void lbs_bwd_skin_kernel(SmplTables T, const float* __restrict__ A,
                                                          const float* __restrict__ vposed,
                                                          const float* __restrict__ dverts,
                                                          const float* __restrict__ dj54, float* __restrict__ part, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, A); DYB_RB(Rp, vposed); DYB_RB(Rp, dverts); DYB_RB(Rp, dj54); DYB_RB(Rp, part);
  __shared__ float sA[NJ * 12], sDj[NJ54 * 3];
  __shared__ int sVj[NVJ];
  const int b = blockIdx.y, lane = threadIdx.x;
  {
    // 288 + 162 + 21 staged values: all loads first (clamped), then the LDS stores - one round trip, not eight
    float a5[5], d3[3];
#pragma unroll
    for (int u = 0; u < 5; ++u) a5[u] = A[(size_t)b * NJ * 12 + (lane + 64 * u < NJ * 12 ? lane + 64 * u : 0)];
#pragma unroll
    for (int u = 0; u < 3; ++u) d3[u] = dj54[(size_t)b * NJ54 * 3 + (lane + 64 * u < NJ54 * 3 ? lane + 64 * u : 0)];
    const int vj = T.vertex_joint_ids[lane < NVJ ? lane : 0];
#pragma unroll
    for (int u = 0; u < 5; ++u)
      if (lane + 64 * u < NJ * 12) sA[lane + 64 * u] = a5[u];
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (lane + 64 * u < NJ54 * 3) sDj[lane + 64 * u] = d3[u];
    if (lane < NVJ) sVj[lane] = vj;
  }
  __syncthreads();
  const int v = blockIdx.x * LBS_VB + lane;
  const bool live = v < NV;
  const int vv = live ? v : 0;
  // every global load of this prologue is unconditional and issued before the first use (dead lanes read vertex 0
  // and are masked): skin weights, the extra-joint regressor column, vposed / dverts
  const float lv = live ? 1.f : 0.f;
  float w[NJ], xe[NEXTRA];
#pragma unroll
  for (int j = 0; j < NJ; ++j) w[j] = T.weights_t[(size_t)j * NV + vv];
#pragma unroll
  for (int e = 0; e < NEXTRA; ++e) xe[e] = T.j_extra[(size_t)e * NV + vv];
  const size_t ov = ((size_t)b * NV + vv) * 3;
  float dv[3] = {0.f, 0.f, 0.f}, vp[3];
  if (dverts) { dv[0] = dverts[ov]; dv[1] = dverts[ov + 1]; dv[2] = dverts[ov + 2]; }     // uniform condition
  vp[0] = vposed[ov]; vp[1] = vposed[ov + 1]; vp[2] = vposed[ov + 2];
#pragma unroll
  for (int j = 0; j < NJ; ++j) w[j] *= lv;
#pragma unroll
  for (int c = 0; c < 3; ++c) { dv[c] *= lv; vp[c] *= lv; }
#pragma unroll
  for (int e = 0; e < NEXTRA; ++e) {
    const float x = xe[e] * lv;
    dv[0] += x * sDj[(NJ + NVJ + e) * 3 + 0];
    dv[1] += x * sDj[(NJ + NVJ + e) * 3 + 1];
    dv[2] += x * sDj[(NJ + NVJ + e) * 3 + 2];
  }
  if (live) {
    for (int k = 0; k < NVJ; ++k) {
      if (sVj[k] == v) {
        dv[0] += sDj[(NJ + k) * 3 + 0];
        dv[1] += sDj[(NJ + k) * 3 + 1];
        dv[2] += sDj[(NJ + k) * 3 + 2];
      }
    }
  }
  float TR[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) TR[e] = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) TR[r * 3 + c] += w[j] * sA[j * 12 + r * 4 + c];
  float dvp[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) dvp[c] = TR[0 * 3 + c] * dv[0] + TR[1 * 3 + c] * dv[1] + TR[2 * 3 + c] * dv[2];

  float* o = part + ((size_t)b * LBS_NBLK + blockIdx.x) * NRED;
  float vals[64];
  float r;
  fill_dA<0>(vals, w, dv, vp);   r = transpose_reduce64(vals, lane); o[lane] = r;
  fill_dA<64>(vals, w, dv, vp);  r = transpose_reduce64(vals, lane); o[64 + lane] = r;
  fill_dA<128>(vals, w, dv, vp); r = transpose_reduce64(vals, lane); o[128 + lane] = r;
  fill_dA<192>(vals, w, dv, vp); r = transpose_reduce64(vals, lane); o[192 + lane] = r;
  fill_dA<256>(vals, w, dv, vp); r = transpose_reduce64(vals, lane); if (lane < NJ * 12 - 256) o[256 + lane] = r;
  const float* pd = T.posedirs + (size_t)vv * 3;
  const float* sd = T.shapedirs + (size_t)vv * 3 * NB;
  float* o2 = o + NJ * 12;
  fill_dpf<0>(vals, pd, sd, dvp, live);   r = transpose_reduce64(vals, lane); o2[lane] = r;
  fill_dpf<64>(vals, pd, sd, dvp, live);  r = transpose_reduce64(vals, lane); o2[64 + lane] = r;
  fill_dpf<128>(vals, pd, sd, dvp, live); r = transpose_reduce64(vals, lane); o2[128 + lane] = r;
  fill_dpf<192>(vals, pd, sd, dvp, live); r = transpose_reduce64(vals, lane); if (lane < NPF + NB - 192) o2[192 + lane] = r;
}

__global__ __launch_bounds__(64) void lbs_bwd_chain_kernel(SmplTables T, const float* __restrict__ part,
                                                           const float* __restrict__ dj54, const float* __restrict__ rot,
                                                           const float* __restrict__ A, const float* __restrict__ Jrest,
                                                           float* __restrict__ drot, float* __restrict__ dbetas, int lddb, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, part); DYB_RB(Rp, dj54); DYB_RB(Rp, rot); DYB_RB(Rp, A); DYB_RB(Rp, Jrest); DYB_RB(Rp, drot); DYB_RB(Rp, dbetas);
  __shared__ float sTot[NRED], sR[NJ * 9], sGR[NJ * 9], sJ[NJ * 3];
  __shared__ float dGR[NJ * 9], dGt[NJ * 3], dJ[NJ * 3], dR[NJ * 9];
  __shared__ int sPar[NJ];
  const int b = blockIdx.x, t = threadIdx.x;
  for (int i = t; i < NRED; i += 64) {
    const float* pp = part + (size_t)b * LBS_NBLK * NRED + i;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int k = 0; k < LBS_NBLK; k += 4) {            // LBS_NBLK = 108 is a multiple of 4
      float a0 = pp[(size_t)k * NRED], a1 = pp[(size_t)(k + 1) * NRED], a2 = pp[(size_t)(k + 2) * NRED], a3 = pp[(size_t)(k + 3) * NRED];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    sTot[i] = (s0 + s1) + (s2 + s3);
  }
  for (int i = t; i < NJ * 9; i += 64) {
    sR[i] = rot[(size_t)b * NJ * 9 + i];
    int j = i / 9, e = i % 9;
    sGR[i] = A[((size_t)b * NJ + j) * 12 + (e / 3) * 4 + (e % 3)];
  }
  for (int i = t; i < NJ * 3; i += 64) sJ[i] = Jrest[(size_t)b * NJ * 3 + i];
  if (t < NJ) sPar[t] = T.parents[t];
  __syncthreads();
  // seeds:  A_R = G_R, A_t = G_t - G_R J, posed joint = G_t
  for (int i = t; i < NJ * 3; i += 64) {
    int j = i / 3, r = i % 3;
    float dAt = sTot[j * 12 + r * 4 + 3];
    dGt[i] = dAt + dj54[((size_t)b * NJ54 + j) * 3 + r];
  }
  for (int i = t; i < NJ * 9; i += 64) {
    int j = i / 9, r = (i % 9) / 3, c = i % 3;
    dGR[i] = sTot[j * 12 + r * 4 + c] - sTot[j * 12 + r * 4 + 3] * sJ[j * 3 + c];
  }
  for (int i = t; i < NJ * 3; i += 64) {
    int j = i / 3, c = i % 3;
    float s = 0.f;
    for (int r = 0; r < 3; ++r) s += sGR[j * 9 + r * 3 + c] * sTot[j * 12 + r * 4 + 3];
    dJ[i] = -s;
  }
  __syncthreads();
  for (int i = NJ - 1; i >= 1; --i) {
    const int p = sPar[i];
    float add = 0.f, drel = 0.f;
    if (t < 9) {
      int r = t / 3, c = t % 3;
      // dR_i = G_p^T dG_i
      dR[i * 9 + t] = sGR[p * 9 + 0 + r] * dGR[i * 9 + 0 + c] + sGR[p * 9 + 3 + r] * dGR[i * 9 + 3 + c] +
                      sGR[p * 9 + 6 + r] * dGR[i * 9 + 6 + c];
      // dG_p += dG_i R_i^T + dGt_i (x) rel_i
      float rel_c = sJ[i * 3 + c] - sJ[p * 3 + c];
      add = dGR[i * 9 + r * 3 + 0] * sR[i * 9 + c * 3 + 0] + dGR[i * 9 + r * 3 + 1] * sR[i * 9 + c * 3 + 1] +
            dGR[i * 9 + r * 3 + 2] * sR[i * 9 + c * 3 + 2] + dGt[i * 3 + r] * rel_c;
    } else if (t < 12) {
      int r = t - 9;
      drel = sGR[p * 9 + 0 + r] * dGt[i * 3 + 0] + sGR[p * 9 + 3 + r] * dGt[i * 3 + 1] + sGR[p * 9 + 6 + r] * dGt[i * 3 + 2];
    }
    __syncthreads();
    if (t < 9) dGR[p * 9 + t] += add;
    else if (t < 12) {
      int r = t - 9;
      dGt[p * 3 + r] += dGt[i * 3 + r];
      dJ[i * 3 + r] += drel;
      dJ[p * 3 + r] -= drel;
    }
    __syncthreads();
  }
  if (t < 9) dR[t] = dGR[t];
  if (t >= 9 && t < 12) dJ[t - 9] += dGt[t - 9];
  __syncthreads();
  for (int i = t; i < NJ * 9; i += 64) {
    float g = dR[i];
    if (i >= 9) g += sTot[NJ * 12 + (i - 9)];
    drot[(size_t)b * NJ * 9 + i] = g;
  }
  if (t < NB) {
    float s = sTot[NJ * 12 + NPF + t];
    for (int k = 0; k < NJ * 3; ++k) s += T.j_shapedirs[k * NB + t] * dJ[k];
    dbetas[(size_t)b * lddb + t] = s;
  }
}

// dverts may be NULL (no loss term touches vertices directly).  Outputs: drot [B][24][9], dbetas [B][lddb].
extern "C" int dyb_lbs_bwd(const float* const* tables_f, const int* const* tables_i, const float* rotmat,
                           const float* saved, const float* djoints49, const float* dverts, float* drot, float* dbetas,
                           int lddb, int B, void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(tables_f && tables_i && rotmat && saved && djoints49 && drot && dbetas && ws && B > 0, DYB_ERR_ARG);
  DYB_REQUIRE(ws_bytes >= dyb_lbs_bwd_workspace_bytes(B), DYB_ERR_WORKSPACE);
  SmplTables T = make_tables(tables_f, tables_i);
  LbsSaved s = carve_saved(const_cast<float*>(saved), B);
  float* dj54 = reinterpret_cast<float*>(ws);
  float* part = dj54 + (size_t)B * NJ54 * 3;
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(lbs_bwd_scatter_kernel, dim3(B, 1, Rp.n), dim3(64), 0, st, T, djoints49, dj54, Rp);
  DYB_CHECK_LAUNCH();
  hipLaunchKernelGGL(lbs_bwd_skin_kernel, dim3(LBS_NBLK, B, Rp.n), dim3(64), 0, st, T, (const float*)s.A,
                     (const float*)s.vposed, dverts, (const float*)dj54, part, Rp);
  DYB_CHECK_LAUNCH();
  hipLaunchKernelGGL(lbs_bwd_chain_kernel, dim3(B, 1, Rp.n), dim3(64), 0, st, T, (const float*)part, (const float*)dj54, rotmat,
                     (const float*)s.A, (const float*)s.J, drot, dbetas, lddb, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// joints[b][j][:] = sum_v reg[j][v] * verts[b][v][:]   (H36M 17-joint regressor of the metric path,
// reference dynaboa_benchmark.py:220-233).  grid (nj, B), block 256.
__global__ __launch_bounds__(256) void regress_joints_kernel(const float* __restrict__ reg, const float* __restrict__ verts,
                                                             float* __restrict__ out, int nj, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, verts); DYB_RB(Rp, out);
  __shared__ float sm[4][3];
  const int j = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int v = t; v < NV; v += 256) {
    float w = reg[(size_t)j * NV + v];
    const float* p = verts + ((size_t)b * NV + v) * 3;
    s0 += w * p[0]; s1 += w * p[1]; s2 += w * p[2];
  }
  s0 = dyb_wave_sum(s0); s1 = dyb_wave_sum(s1); s2 = dyb_wave_sum(s2);
  if (lane == 0) { sm[wave][0] = s0; sm[wave][1] = s1; sm[wave][2] = s2; }
  __syncthreads();
  if (t < 3) out[((size_t)b * nj + j) * 3 + t] = (sm[0][t] + sm[1][t]) + (sm[2][t] + sm[3][t]);
}
extern "C" int dyb_regress_joints(const float* reg, const float* verts, float* out, int nj, int B, hipStream_t st) {
  DYB_REQUIRE(reg && verts && out && nj > 0 && B > 0, DYB_ERR_ARG);
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(regress_joints_kernel, dim3(nj, B, Rp.n), dim3(256), 0, st, reg, verts, out, nj, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
