// Frame-loss head: value AND gradient of
//     w2d * kp2d_loss + wshape * shape_prior + wpose * pose_prior
// in one launch (one workgroup per sample), plus stand-alone projection fwd/bwd used by the
// temporal / teacher terms.
//
// Follows reference base_adaptor.py:160-170 (projection), :229,234 (confidence-masked 2-D MSE over
// the 24 GT-style joints), :401-409 (shape / pose priors), utils/geometry.py:184-306
// (rotation_matrix_to_angle_axis, kornia-derived 4-branch quaternion formula, NaN -> 0) and
// utils/smplify/prior.py:181-196 (merged max-mixture GMM).  The analytic backward is the chain
// rule of exactly that formula (same branch selection), so it matches torch autograd of the
// reference away from the theta -> 0 singularity (where autograd itself produces inf*0).
#include "dyb_common.h"

#define NJ 24
#define NJ49 49
#define NG 8
#define ND 69
#define FOCAL 5000.0f
#define IMG_RES 224.0f

struct QuatFwd {
  int branch;
  float t;        // selected trace term
  float u[4];     // un-normalised quaternion (one component equals t)
  float q[4];     // 0.5*u/sqrt(t)
};
__device__ __forceinline__ QuatFwd rot_to_quat(const float* R) {
  const float eps = 1e-6f;
  float r00 = R[0], r01 = R[1], r02 = R[2], r10 = R[3], r11 = R[4], r12 = R[5], r20 = R[6], r21 = R[7], r22 = R[8];
  bool d2 = r22 < eps, d01 = r00 > r11, d0n1 = r00 < -r11;
  QuatFwd f;
  if (d2 && d01) {
    f.branch = 0; f.t = 1.f + r00 - r11 - r22;
    f.u[0] = r21 - r12; f.u[1] = f.t; f.u[2] = r10 + r01; f.u[3] = r02 + r20;
  } else if (d2) {
    f.branch = 1; f.t = 1.f - r00 + r11 - r22;
    f.u[0] = r02 - r20; f.u[1] = r10 + r01; f.u[2] = f.t; f.u[3] = r21 + r12;
  } else if (d0n1) {
    f.branch = 2; f.t = 1.f - r00 - r11 + r22;
    f.u[0] = r10 - r01; f.u[1] = r02 + r20; f.u[2] = r21 + r12; f.u[3] = f.t;
  } else {
    f.branch = 3; f.t = 1.f + r00 + r11 + r22;
    f.u[0] = f.t; f.u[1] = r21 - r12; f.u[2] = r02 - r20; f.u[3] = r10 - r01;
  }
  float k = 0.5f / sqrtf(f.t);
  for (int i = 0; i < 4; ++i) f.q[i] = f.u[i] * k;
  return f;
}
__device__ __forceinline__ void quat_to_aa(const float* q, float* aa) {
  float w = q[0];
  float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  float s = sqrtf(s2);
  float tt = 2.f * (w < 0.f ? atan2f(-s, -w) : atan2f(s, w));
  float k = s2 > 0.f ? tt / s : 2.f;
  for (int i = 0; i < 3; ++i) {
    float v = q[1 + i] * k;
    aa[i] = (v != v) ? 0.f : v;
  }
}
// dR (9) from g = dL/d(aa)
__device__ __forceinline__ void aa_bwd(const float* R, const float* g, float* dR) {
  QuatFwd f = rot_to_quat(R);
  const float* q = f.q;
  float w = q[0];
  float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  float s = sqrtf(s2);
  float gq[4] = {0.f, 0.f, 0.f, 0.f};
  if (s2 > 0.f) {
    float tt = 2.f * (w < 0.f ? atan2f(-s, -w) : atan2f(s, w));
    float k = tt / s;
    float gk = g[0] * q[1] + g[1] * q[2] + g[2] * q[3];
    float den = s2 + w * w;
    float g_tt = gk / s;
    float g_s = -gk * tt / s2 + g_tt * (2.f * w / den);
    gq[0] = g_tt * (-2.f * s / den);
    float g_s2 = g_s / (2.f * s);
    for (int i = 0; i < 3; ++i) gq[1 + i] = g[i] * k + 2.f * q[1 + i] * g_s2;
  } else {
    for (int i = 0; i < 3; ++i) gq[1 + i] = g[i] * 2.f;
  }
  // q = 0.5 * u / sqrt(t)
  float rt = sqrtf(f.t);
  float gu[4];
  float dot = 0.f;
  for (int i = 0; i < 4; ++i) {
    gu[i] = 0.5f * gq[i] / rt;
    dot += gq[i] * f.u[i];
  }
  float gt = -0.25f * dot / (f.t * rt);
  for (int i = 0; i < 9; ++i) dR[i] = 0.f;
  // index order: r00 0, r01 1, r02 2, r10 3, r11 4, r12 5, r20 6, r21 7, r22 8
  switch (f.branch) {
    case 0:
      gt += gu[1];
      dR[7] += gu[0]; dR[5] -= gu[0];
      dR[3] += gu[2]; dR[1] += gu[2];
      dR[2] += gu[3]; dR[6] += gu[3];
      dR[0] += gt; dR[4] -= gt; dR[8] -= gt;
      break;
    case 1:
      gt += gu[2];
      dR[2] += gu[0]; dR[6] -= gu[0];
      dR[3] += gu[1]; dR[1] += gu[1];
      dR[7] += gu[3]; dR[5] += gu[3];
      dR[0] -= gt; dR[4] += gt; dR[8] -= gt;
      break;
    case 2:
      gt += gu[3];
      dR[3] += gu[0]; dR[1] -= gu[0];
      dR[2] += gu[1]; dR[6] += gu[1];
      dR[7] += gu[2]; dR[5] += gu[2];
      dR[0] -= gt; dR[4] -= gt; dR[8] += gt;
      break;
    default:
      gt += gu[0];
      dR[7] += gu[1]; dR[5] -= gu[1];
      dR[2] += gu[2]; dR[6] -= gu[2];
      dR[3] += gu[3]; dR[1] -= gu[3];
      dR[0] += gt; dR[4] += gt; dR[8] += gt;
      break;
  }
}

__global__ __launch_bounds__(64) void rotmat_to_aa_kernel(const float* __restrict__ R, float* __restrict__ aa, int n) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  QuatFwd f = rot_to_quat(R + (size_t)i * 9);
  quat_to_aa(f.q, aa + (size_t)i * 3);
}
__global__ __launch_bounds__(64) void rotmat_to_aa_bwd_kernel(const float* __restrict__ R, const float* __restrict__ g,
                                                              float* __restrict__ dR, int n) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float d[9];
  aa_bwd(R + (size_t)i * 9, g + (size_t)i * 3, d);
  for (int k = 0; k < 9; ++k) dR[(size_t)i * 9 + k] = d[k];
}
extern "C" int dyb_rotmat_to_aa_fwd(const float* R, float* aa, int n, hipStream_t st) {
  DYB_REQUIRE(R && aa && n > 0, DYB_ERR_ARG);
  hipLaunchKernelGGL(rotmat_to_aa_kernel, dim3(dyb_cdiv(n, 64)), dim3(64), 0, st, R, aa, n);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
extern "C" int dyb_rotmat_to_aa_bwd(const float* R, const float* daa, float* dR, int n, hipStream_t st) {
  DYB_REQUIRE(R && daa && dR && n > 0, DYB_ERR_ARG);
  hipLaunchKernelGGL(rotmat_to_aa_bwd_kernel, dim3(dyb_cdiv(n, 64)), dim3(64), 0, st, R, daa, dR, n);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// ------------------------------------------------------------------------------------------
// projection (fwd / bwd), n points per sample
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void projection_fwd_kernel(const float* __restrict__ cam, int ldc,
                                                            const float* __restrict__ p3, float* __restrict__ p2, int np,
                                                            int total) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= total) return;
  int b = i / np;
  const float* c = cam + (size_t)b * ldc;
  float tz = 2.f * FOCAL / (IMG_RES * c[0] + 1e-9f);
  float x = p3[(size_t)i * 3] + c[1], y = p3[(size_t)i * 3 + 1] + c[2], z = p3[(size_t)i * 3 + 2] + tz;
  p2[(size_t)i * 2] = FOCAL * (x / z) / (IMG_RES * 0.5f);
  p2[(size_t)i * 2 + 1] = FOCAL * (y / z) / (IMG_RES * 0.5f);
}
// dp3 per point; dcam reduced per sample by one workgroup (grid = B)
__global__ __launch_bounds__(64) void projection_bwd_kernel(const float* __restrict__ cam, int ldc,
                                                            const float* __restrict__ p3, const float* __restrict__ g2,
                                                            float* __restrict__ dp3, float* __restrict__ dcam, int lddc,
                                                            int np) {
  const int b = blockIdx.x, t = threadIdx.x;
  const float* c = cam + (size_t)b * ldc;
  float den = IMG_RES * c[0] + 1e-9f;
  float tz = 2.f * FOCAL / den;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int j = t; j < np; j += 64) {
    size_t i = (size_t)b * np + j;
    float x = p3[i * 3] + c[1], y = p3[i * 3 + 1] + c[2], z = p3[i * 3 + 2] + tz;
    float k = FOCAL / (IMG_RES * 0.5f);
    float gx = g2[i * 2] * k / z, gy = g2[i * 2 + 1] * k / z;
    float gz = -(gx * x + gy * y) / z;
    dp3[i * 3] = gx; dp3[i * 3 + 1] = gy; dp3[i * 3 + 2] = gz;
    sx += gx; sy += gy; sz += gz;
  }
  sx = dyb_wave_sum(sx); sy = dyb_wave_sum(sy); sz = dyb_wave_sum(sz);
  if (t == 0) {
    float* d = dcam + (size_t)b * lddc;
    d[0] = sz * (-2.f * FOCAL * IMG_RES / (den * den));
    d[1] = sx;
    d[2] = sy;
  }
}
extern "C" int dyb_projection_fwd(const float* cam, int ldc, const float* p3, float* p2, int B, int np, hipStream_t st) {
  DYB_REQUIRE(cam && p3 && p2 && B > 0 && np > 0, DYB_ERR_ARG);
  hipLaunchKernelGGL(projection_fwd_kernel, dim3(dyb_cdiv(B * np, 64)), dim3(64), 0, st, cam, ldc, p3, p2, np, B * np);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
extern "C" int dyb_projection_bwd(const float* cam, int ldc, const float* p3, const float* g2, float* dp3, float* dcam,
                                  int lddc, int B, int np, hipStream_t st) {
  DYB_REQUIRE(cam && p3 && g2 && dp3 && dcam && B > 0 && np > 0, DYB_ERR_ARG);
  hipLaunchKernelGGL(projection_bwd_kernel, dim3(B), dim3(64), 0, st, cam, ldc, p3, g2, dp3, dcam, lddc, np);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// ------------------------------------------------------------------------------------------
// utils/geometry.py:63-91 perspective_projection: out = f * (R p + t).xy / (R p + t).z + c  - the general form behind
// BaseAdaptor.projection (whose identity-rotation / weak-perspective case above is what the adaptation path itself uses).
// Differentiable w.r.t. the points and the translation (what the reference differentiates: the predicted joints and camera).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void persp_fwd_kernel(const float* __restrict__ pts, const float* __restrict__ rot,
                                                       const float* __restrict__ tr, const float* __restrict__ focal, int ldf,
                                                       const float* __restrict__ cen, float* __restrict__ out, int B, int np) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= B * np) return;
  const int b = i / np;
  const float* R = rot + (size_t)b * 9;
  const float* p = pts + (size_t)i * 3;
  const float x = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + tr[b * 3], y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + tr[b * 3 + 1],
              z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + tr[b * 3 + 2];
  const float f = focal[(size_t)b * ldf];
  out[(size_t)i * 2] = f * (x / z) + cen[b * 2];
  out[(size_t)i * 2 + 1] = f * (y / z) + cen[b * 2 + 1];
}
// one workgroup per sample: d(points) per point, d(translation) summed over the sample's points (np <= 4096)
__global__ __launch_bounds__(64) void persp_bwd_kernel(const float* __restrict__ pts, const float* __restrict__ rot,
                                                       const float* __restrict__ tr, const float* __restrict__ focal, int ldf,
                                                       const float* __restrict__ g2, float* __restrict__ dpts,
                                                       float* __restrict__ dtr, int np) {
  const int b = blockIdx.x, t = threadIdx.x;
  const float* R = rot + (size_t)b * 9;
  const float f = focal[(size_t)b * ldf];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int j = t; j < np; j += 64) {
    const size_t i = (size_t)b * np + j;
    const float* p = pts + i * 3;
    const float x = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + tr[b * 3], y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + tr[b * 3 + 1],
                z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + tr[b * 3 + 2];
    const float gx = f * g2[i * 2] / z, gy = f * g2[i * 2 + 1] / z, gz = -(gx * x + gy * y) / z;
    dpts[i * 3] = R[0] * gx + R[3] * gy + R[6] * gz;
    dpts[i * 3 + 1] = R[1] * gx + R[4] * gy + R[7] * gz;
    dpts[i * 3 + 2] = R[2] * gx + R[5] * gy + R[8] * gz;
    sx += gx; sy += gy; sz += gz;
  }
  sx = dyb_wave_sum(sx); sy = dyb_wave_sum(sy); sz = dyb_wave_sum(sz);
  if (t == 0) { dtr[b * 3] = sx; dtr[b * 3 + 1] = sy; dtr[b * 3 + 2] = sz; }
}
// focal: per sample with stride ldf floats (ldf = 0: one value for all)
extern "C" int dyb_perspective_projection_fwd(const float* points, const float* rotation, const float* translation, const float* focal,
                                              int ldf, const float* center, float* out, int B, int np, hipStream_t st) {
  DYB_REQUIRE(points && rotation && translation && focal && center && out && B > 0 && np > 0 && ldf >= 0, DYB_ERR_ARG);
  hipLaunchKernelGGL(persp_fwd_kernel, dim3(dyb_cdiv(B * np, 64)), dim3(64), 0, st, points, rotation, translation, focal, ldf, center, out,
                     B, np);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
extern "C" int dyb_perspective_projection_bwd(const float* points, const float* rotation, const float* translation, const float* focal,
                                              int ldf, const float* g2, float* dpoints, float* dtranslation, int B, int np,
                                              hipStream_t st) {
  DYB_REQUIRE(points && rotation && translation && focal && g2 && dpoints && dtranslation && B > 0 && np > 0 && ldf >= 0, DYB_ERR_ARG);
  hipLaunchKernelGGL(persp_bwd_kernel, dim3(B), dim3(64), 0, st, points, rotation, translation, focal, ldf, g2, dpoints, dtranslation, np);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// ------------------------------------------------------------------------------------------
// MaxMixturePrior.merged_log_likelihood (utils/smplify/prior.py:181-196) on an axis-angle body pose [B][69]: per-sample
// min_m (0.5 d_m^T P_m d_m - log w_m) and its gradient w.r.t. the pose - the module-level form of the prior (the adaptation path
// uses the copy inside frame_losses_kernel, which starts from rotation matrices).  One workgroup per sample.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gmm_prior_kernel(const float* __restrict__ pose, const float* __restrict__ means,
                                                        const float* __restrict__ prec, const float* __restrict__ logw,
                                                        float* __restrict__ out, float* __restrict__ dpose) {
  __shared__ float sD[NG][ND], sRow[NG][ND], sCol[NG][ND], sQ[NG];
  __shared__ int sBest;
  const int b = blockIdx.x, t = threadIdx.x;
  for (int i = t; i < NG * ND; i += 256) sD[i / ND][i % ND] = pose[(size_t)b * ND + i % ND] - means[i];
  __syncthreads();
  for (int i = t; i < NG * ND; i += 256) {
    const int m = i / ND, k = i % ND;
    const float* P = prec + (size_t)m * ND * ND;
    float r = 0.f, c = 0.f;
    for (int j = 0; j < ND; ++j) {
      const float d = sD[m][j];
      r += P[k * ND + j] * d;
      c += P[j * ND + k] * d;
    }
    sRow[m][k] = r;
    sCol[m][k] = c;
  }
  __syncthreads();
  if (t < NG) {
    float q = 0.f;
    for (int k = 0; k < ND; ++k) q += sRow[t][k] * sD[t][k];
    sQ[t] = 0.5f * q - logw[t];
  }
  __syncthreads();
  if (t == 0) {
    int best = 0;
    for (int m = 1; m < NG; ++m)
      if (sQ[m] < sQ[best]) best = m;
    sBest = best;
    out[b] = sQ[best];
  }
  __syncthreads();
  if (dpose && t < ND) dpose[(size_t)b * ND + t] = 0.5f * (sRow[sBest][t] + sCol[sBest][t]);
}
extern "C" int dyb_gmm_prior(const float* pose69, const float* gmm_means, const float* gmm_prec, const float* gmm_logw, float* out,
                             float* dpose69, int B, hipStream_t st) {
  DYB_REQUIRE(pose69 && gmm_means && gmm_prec && gmm_logw && out && B > 0, DYB_ERR_ARG);
  hipLaunchKernelGGL(gmm_prior_kernel, dim3(B), dim3(256), 0, st, pose69, gmm_means, gmm_prec, gmm_logw, out, dpose69);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// ------------------------------------------------------------------------------------------
// fused frame losses, value + gradient
// ------------------------------------------------------------------------------------------
struct FrameLossArgs {
  const float* rot;      // [B][24][9]
  const float* shape;    // [B][lds]  (10 used)
  const float* cam;      // [B][ldc]  (3 used)
  const float* joints;   // [B][49][3]
  const float* kp;       // [B][49][3] (x, y, conf)
  const float* means;    // [8][69]
  const float* prec;     // [8][69][69]
  const float* logw;     // [8]   log(nll_weights)
  float* parts;          // [B][4]  per-sample (s2d, shape, pose, weighted total) already / B etc.
  float* drot;           // [B][24][9]
  float* dshape;         // [B][ldds]
  float* dcam;           // [B][lddc]
  float* djoints;        // [B][49][3]
  int lds, ldc, ldds, lddc, B;
  float w2d, wshape, wpose;
};

__global__ __launch_bounds__(256) void frame_losses_kernel(FrameLossArgs a, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  if (dyb_rep) {
    a.rot = dyb_rb(a.rot, Rp, dyb_rep); a.shape = dyb_rb(a.shape, Rp, dyb_rep); a.cam = dyb_rb(a.cam, Rp, dyb_rep);
    a.joints = dyb_rb(a.joints, Rp, dyb_rep); a.kp = dyb_rb(a.kp, Rp, dyb_rep); a.parts = dyb_rb(a.parts, Rp, dyb_rep);
    a.drot = dyb_rb(a.drot, Rp, dyb_rep); a.dshape = dyb_rb(a.dshape, Rp, dyb_rep); a.dcam = dyb_rb(a.dcam, Rp, dyb_rep);
    a.djoints = dyb_rb(a.djoints, Rp, dyb_rep);
  }
  __shared__ float sAA[ND], sD[NG][ND], sRow[NG][ND], sCol[NG][ND], sQ[NG], sG[ND];
  __shared__ float sCamG[NJ][3], sL2d[NJ];
  __shared__ int sBest;
  const int b = blockIdx.x, t = threadIdx.x;
  const float invB = 1.0f / (float)a.B;
  const float* R = a.rot + (size_t)b * NJ * 9;

  // --- pose prior: axis-angle of the 23 body joints
  if (t < NJ - 1) {
    QuatFwd f = rot_to_quat(R + (t + 1) * 9);
    quat_to_aa(f.q, &sAA[t * 3]);
  }
  __syncthreads();
  for (int i = t; i < NG * ND; i += 256) {
    int m = i / ND, k = i % ND;
    sD[m][k] = sAA[k] - a.means[i];
  }
  __syncthreads();
  for (int i = t; i < NG * ND; i += 256) {
    int m = i / ND, k = i % ND;
    const float* P = a.prec + (size_t)m * ND * ND;
    float r = 0.f, c = 0.f;
    for (int j = 0; j < ND; ++j) {
      float d = sD[m][j];
      r += P[k * ND + j] * d;
      c += P[j * ND + k] * d;
    }
    sRow[m][k] = r;
    sCol[m][k] = c;
  }
  __syncthreads();
  if (t < NG) {
    float q = 0.f;
    for (int k = 0; k < ND; ++k) q += sRow[t][k] * sD[t][k];
    sQ[t] = 0.5f * q - a.logw[t];
  }
  __syncthreads();
  if (t == 0) {
    int best = 0;
    for (int m = 1; m < NG; ++m)
      if (sQ[m] < sQ[best]) best = m;
    sBest = best;
  }
  __syncthreads();
  const int mb = sBest;
  if (t < ND) sG[t] = a.wpose * invB * 0.5f * (sRow[mb][t] + sCol[mb][t]);
  __syncthreads();
  if (t < NJ) {
    float d[9];
    if (t == 0) {
      for (int k = 0; k < 9; ++k) d[k] = 0.f;
    } else {
      aa_bwd(R + t * 9, &sG[(t - 1) * 3], d);
    }
    for (int k = 0; k < 9; ++k) a.drot[((size_t)b * NJ + t) * 9 + k] = d[k];
  }

  // --- 2-D keypoint loss on joints 25..48
  const float* c = a.cam + (size_t)b * a.ldc;
  const float den = IMG_RES * c[0] + 1e-9f;
  const float tz = 2.f * FOCAL / den;
  for (int i = t; i < NJ49 * 3; i += 256) a.djoints[(size_t)b * NJ49 * 3 + i] = 0.f;
  __syncthreads();
  if (t < NJ) {
    int j = 25 + t;
    const float* p = a.joints + ((size_t)b * NJ49 + j) * 3;
    const float* k = a.kp + ((size_t)b * NJ49 + j) * 3;
    float x = p[0] + c[1], y = p[1] + c[2], z = p[2] + tz;
    const float sc = FOCAL / (IMG_RES * 0.5f);
    float ex = sc * (x / z) - k[0], ey = sc * (y / z) - k[1];
    float conf = k[2];
    float norm = invB / (float)(NJ * 2);
    sL2d[t] = conf * (ex * ex + ey * ey) * norm;
    float gx2 = a.w2d * 2.f * conf * ex * norm, gy2 = a.w2d * 2.f * conf * ey * norm;
    float gx = gx2 * sc / z, gy = gy2 * sc / z;
    float gz = -(gx * x + gy * y) / z;
    float* dj = a.djoints + ((size_t)b * NJ49 + j) * 3;
    dj[0] = gx; dj[1] = gy; dj[2] = gz;
    sCamG[t][0] = gx; sCamG[t][1] = gy; sCamG[t][2] = gz;
  }
  __syncthreads();
  if (t == 0) {
    float sx = 0.f, sy = 0.f, sz = 0.f, l2d = 0.f;
    for (int j = 0; j < NJ; ++j) {
      sx += sCamG[j][0]; sy += sCamG[j][1]; sz += sCamG[j][2];
      l2d += sL2d[j];
    }
    float* dc = a.dcam + (size_t)b * a.lddc;
    dc[0] = sz * (-2.f * FOCAL * IMG_RES / (den * den));
    dc[1] = sx;
    dc[2] = sy;
    const float* be = a.shape + (size_t)b * a.lds;
    float lsh = 0.f;
    for (int l = 0; l < 10; ++l) {
      lsh += be[l] * be[l];
      a.dshape[(size_t)b * a.ldds + l] = a.wshape * 2.f * be[l] * invB;
    }
    lsh *= invB;
    float lpo = sQ[mb] * invB;
    float* o = a.parts + (size_t)b * 4;
    o[0] = l2d; o[1] = lsh; o[2] = lpo;
    o[3] = a.w2d * l2d + a.wshape * lsh + a.wpose * lpo;
  }
}

// out[0..3] = sum_b parts[b][0..3]
__global__ void loss_fold_kernel(const float* __restrict__ parts, float* __restrict__ out, int B, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, parts); DYB_RB(Rp, out);
  int t = threadIdx.x;
  if (t < 4) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += parts[b * 4 + t];
    out[t] = s;
  }
}

// losses_out[4] = (s2dloss, shape_prior, pose_prior, weighted total) as the reference defines them
// (means over the batch); gradients are those of the weighted total.
extern "C" int dyb_frame_losses(const float* rotmat, const float* shape, int lds, const float* cam, int ldc,
                                const float* joints49, const float* kp2d, const float* gmm_means,
                                const float* gmm_prec, const float* gmm_logw, float w2d, float wshape, float wpose,
                                float* losses_out, float* drot, float* dshape, int ldds, float* dcam, int lddc,
                                float* djoints49, int B, void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(rotmat && shape && cam && joints49 && kp2d && gmm_means && gmm_prec && gmm_logw, DYB_ERR_ARG);
  DYB_REQUIRE(losses_out && drot && dshape && dcam && djoints49 && ws && B > 0, DYB_ERR_ARG);
  DYB_REQUIRE(ws_bytes >= (size_t)B * 4 * sizeof(float), DYB_ERR_WORKSPACE);
  FrameLossArgs a;
  a.rot = rotmat; a.shape = shape; a.cam = cam; a.joints = joints49; a.kp = kp2d;
  a.means = gmm_means; a.prec = gmm_prec; a.logw = gmm_logw;
  a.parts = reinterpret_cast<float*>(ws);
  a.drot = drot; a.dshape = dshape; a.dcam = dcam; a.djoints = djoints49;
  a.lds = lds; a.ldc = ldc; a.ldds = ldds; a.lddc = lddc; a.B = B;
  a.w2d = w2d; a.wshape = wshape; a.wpose = wpose;
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(frame_losses_kernel, dim3(B, 1, Rp.n), dim3(256), 0, st, a, Rp);
  DYB_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_fold_kernel, dim3(1, 1, Rp.n), dim3(64), 0, st, (const float*)a.parts, losses_out, B, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// ---- gradient assembly of the fused HMR + SMPL + frame-loss autograd node (dynaboa_amd/fused_level.py) ----
// out[i] = g * a[i] (+ ext[i]);  g: device scalar (the incoming gradient of the loss total), NULL = 1
__global__ __launch_bounds__(256) void scale_add_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                        const float* __restrict__ ext, float* __restrict__ out, size_t n,
                                                        DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, g); DYB_RB(Rp, a); DYB_RB(Rp, ext); DYB_RB(Rp, out);
  const float s = g ? g[0] : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = s * a[i] + (ext ? ext[i] : 0.f);
}
extern "C" int dyb_scale_add(const float* g, const float* a, const float* ext, float* out, size_t n, hipStream_t st) {
  DYB_REQUIRE(a && out && n > 0, DYB_ERR_ARG);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(scale_add_kernel, dim3(blocks, 1, Rp.n), dim3(256), 0, st, g, a, ext, out, n, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// d_rot[B][216]   = g*drot_loss + drot_smpl (+ drot_ext)
// d_state[B][160] : cols 144..153 = g*dshape_loss + dbetas_smpl (+ dshape_ext), cols 154..156 = g*dcam_loss (+ dcam_ext)
// (exactly what dyb_hmr_backward consumes; the loss-side pieces come from dyb_frame_losses, the SMPL-side ones from dyb_lbs_bwd)
struct HeadGradArgs {
  const float *g, *drot_l, *drot_s, *drot_e, *dshape_l, *dbetas_s, *dshape_e, *dcam_l, *dcam_e;
  float *d_rot, *d_state;
  int B;
};
__global__ __launch_bounds__(256) void head_grad_kernel(HeadGradArgs a, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  if (dyb_rep) {
    a.g = dyb_rb(a.g, Rp, dyb_rep); a.drot_l = dyb_rb(a.drot_l, Rp, dyb_rep); a.drot_s = dyb_rb(a.drot_s, Rp, dyb_rep);
    a.drot_e = dyb_rb(a.drot_e, Rp, dyb_rep); a.dshape_l = dyb_rb(a.dshape_l, Rp, dyb_rep); a.dbetas_s = dyb_rb(a.dbetas_s, Rp, dyb_rep);
    a.dshape_e = dyb_rb(a.dshape_e, Rp, dyb_rep); a.dcam_l = dyb_rb(a.dcam_l, Rp, dyb_rep); a.dcam_e = dyb_rb(a.dcam_e, Rp, dyb_rep);
    a.d_rot = dyb_rb(a.d_rot, Rp, dyb_rep); a.d_state = dyb_rb(a.d_state, Rp, dyb_rep);
  }
  const float s = a.g ? a.g[0] : 1.f;
  const int per = 216 + 13;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.B * per; i += gridDim.x * 256) {
    const int b = i / per, j = i % per;
    if (j < 216) {
      const int o = b * 216 + j;
      a.d_rot[o] = s * a.drot_l[o] + a.drot_s[o] + (a.drot_e ? a.drot_e[o] : 0.f);
    } else if (j < 226) {
      const int c = j - 216;
      a.d_state[b * 160 + 144 + c] = s * a.dshape_l[b * 10 + c] + a.dbetas_s[b * 10 + c] + (a.dshape_e ? a.dshape_e[b * 10 + c] : 0.f);
    } else {
      const int c = j - 226;
      a.d_state[b * 160 + 154 + c] = s * a.dcam_l[b * 3 + c] + (a.dcam_e ? a.dcam_e[b * 3 + c] : 0.f);
    }
  }
}
extern "C" int dyb_head_grad_combine(const float* g, const float* drot_loss, const float* drot_smpl, const float* drot_ext,
                                     const float* dshape_loss, const float* dbetas_smpl, const float* dshape_ext,
                                     const float* dcam_loss, const float* dcam_ext, float* d_rot, float* d_state, int B,
                                     hipStream_t st) {
  DYB_REQUIRE(drot_loss && drot_smpl && dshape_loss && dbetas_smpl && dcam_loss && d_rot && d_state && B > 0, DYB_ERR_ARG);
  HeadGradArgs a{g, drot_loss, drot_smpl, drot_ext, dshape_loss, dbetas_smpl, dshape_ext, dcam_loss, dcam_ext, d_rot, d_state, B};
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(head_grad_kernel, dim3(dyb_cdiv(B * 229, 256), 1, Rp.n), dim3(256), 0, st, a, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// ---- PA-MPJPE on the device: reference utils/pose_utils.py:9-57 (compute_similarity_transform) + the error of
// dynaboa_benchmark.py:236-240, one thread per sample.  Similarity-align pred (J x 3) onto gt: centre both,
// K = X1^T X2, SVD K = U S V^T, R = V diag(1,1,sign det(U V^T)) U^T, scale = tr(R K)/|X1|^2, t = mu2 - scale R mu1;
// out = mean_j |scale R p_j + t - g_j|.  The 3x3 SVD is a one-sided Jacobi in double (converges in <= 6 sweeps;
// tr(R K) = s1 + s2 + z s3 needs no explicit product); the rotation is unique whenever rank(K) >= 2.
__device__ void pa_svd3(const double K[3][3], double U[3][3], double S[3], double V[3][3]) {
  double A[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { A[i][j] = K[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double a = 0.0, b = 0.0, c = 0.0;          // |col p|^2, |col q|^2, col p . col q
        for (int i = 0; i < 3; ++i) { a += A[i][p] * A[i][p]; b += A[i][q] * A[i][q]; c += A[i][p] * A[i][q]; }
        off += c * c;
        if (fabs(c) <= 1e-300 || c * c <= 1e-32 * a * b) continue;
        double zeta = (b - a) / (2.0 * c);
        double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < 3; ++i) {
          double x = A[i][p], y = A[i][q];
          A[i][p] = cs * x - sn * y; A[i][q] = sn * x + cs * y;
          x = V[i][p]; y = V[i][q];
          V[i][p] = cs * x - sn * y; V[i][q] = sn * x + cs * y;
        }
      }
    if (off <= 1e-60) break;
  }
  for (int j = 0; j < 3; ++j) S[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
  // order descending (columns of A and V together)
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (S[b] > S[a]) {
        double t = S[a]; S[a] = S[b]; S[b] = t;
        for (int i = 0; i < 3; ++i) {
          t = A[i][a]; A[i][a] = A[i][b]; A[i][b] = t;
          t = V[i][a]; V[i][a] = V[i][b]; V[i][b] = t;
        }
      }
  for (int j = 0; j < 2; ++j) {
    double inv = S[j] > 0.0 ? 1.0 / S[j] : 0.0;
    for (int i = 0; i < 3; ++i) U[i][j] = A[i][j] * inv;
  }
  if (S[2] > 1e-12 * S[0]) {
    for (int i = 0; i < 3; ++i) U[i][2] = A[i][2] / S[2];
  } else {             // rank 2: complete U with the cross product (any sign: Z fixes the handedness of R)
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  }
}
__device__ double pa_det3(const double M[3][3]) {
  return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
         M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}
__global__ __launch_bounds__(64) void pa_mpjpe_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                      float* __restrict__ out, float* __restrict__ aligned, int n, int J) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= n) return;
  const float* P = pred + (size_t)s * J * 3;
  const float* Gt = gt + (size_t)s * J * 3;
  double m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0};
  for (int j = 0; j < J; ++j)
    for (int c = 0; c < 3; ++c) { m1[c] += P[j * 3 + c]; m2[c] += Gt[j * 3 + c]; }
  for (int c = 0; c < 3; ++c) { m1[c] /= J; m2[c] /= J; }
  double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0.0;
  for (int j = 0; j < J; ++j) {
    double x[3], y[3];
    for (int c = 0; c < 3; ++c) { x[c] = P[j * 3 + c] - m1[c]; y[c] = Gt[j * 3 + c] - m2[c]; var1 += x[c] * x[c]; }
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) K[a][b] += x[a] * y[b];
  }
  double U[3][3], S[3], V[3][3], UVt[3][3];
  pa_svd3(K, U, S, V);
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) UVt[a][b] = U[a][0] * V[b][0] + U[a][1] * V[b][1] + U[a][2] * V[b][2];
  const double z = pa_det3(UVt) < 0.0 ? -1.0 : 1.0;
  double R[3][3];                                    // R = V diag(1,1,z) U^T
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) R[a][b] = V[a][0] * U[b][0] + V[a][1] * U[b][1] + z * V[a][2] * U[b][2];
  const double scale = (S[0] + S[1] + z * S[2]) / var1;
  double t[3];
  for (int a = 0; a < 3; ++a) t[a] = m2[a] - scale * (R[a][0] * m1[0] + R[a][1] * m1[1] + R[a][2] * m1[2]);
  double err = 0.0;
  for (int j = 0; j < J; ++j) {
    double d2 = 0.0;
    for (int a = 0; a < 3; ++a) {
      double h = scale * (R[a][0] * P[j * 3] + R[a][1] * P[j * 3 + 1] + R[a][2] * P[j * 3 + 2]) + t[a];
      if (aligned) aligned[((size_t)s * J + j) * 3 + a] = (float)h;
      double d = h - Gt[j * 3 + a];
      d2 += d * d;
    }
    err += sqrt(d2);
  }
  out[s] = (float)(err / J);
}
// pred, gt: [n][J][3]; out[n] = Procrustes-aligned mean per-joint error (same unit as the inputs);
// aligned (optional, [n][J][3]) receives the aligned prediction (compute_similarity_transform_batch's return value)
extern "C" int dyb_pa_mpjpe(const float* pred, const float* gt, float* out, float* aligned, int n, int J, hipStream_t st) {
  DYB_REQUIRE(pred && gt && out && n > 0 && J >= 3, DYB_ERR_ARG);
  hipLaunchKernelGGL(pa_mpjpe_kernel, dim3(dyb_cdiv(n, 64)), dim3(64), 0, st, pred, gt, out, aligned, n, J);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// ------------------------------------------------------------------------------------------
// The other terms of the level losses, value + gradient, one launch each (one workgroup; <= 16 samples):
//   mode 0  mean-teacher consistency   reference base_adaptor.py:320-343 (cal_teacher_loss)
//           5*mse(s2d, t_s2d) + 5*mse(t_s3d, s3d) + 0.001*mse(shape, t_shape) + mse(rotmat, t_rotmat)
//   mode 1  motion                     :379-398 (cal_motion_loss): masked mse of (s2d - hist_s2d) against the keypoint motion,
//           over the 24 GT-style joints, confidence = both frames' confidences are 1
//   mode 2  labelled exemplar          :346-376 (adapt_on_labeled_data) + :412-422 (cal_s3d_loss, hip-centred):
//           5*l2d + 5*l3d + 0.001*mse(shape, betas) + mse(rotmat, rodrigues(pose))
// "s2d" is the [-1,1]-normalised projection of the 49 joints with the predicted camera (:160-170), formed here.  Gradients are
// those of weight * term w.r.t. the student pass's rotmat / shape / cam / joints49 (dense [B][216], [B][10], [B][3], [B][147];
// accumulate != 0: added to what is there - several terms attach to one pass) and, for the motion term, w.r.t. the history
// pass's cam / joints49.  vals: mode 0 -> {s2d, s3d, shape, pose, loss}; mode 1 -> {motion}; mode 2 -> {s2d, s3d, shape, pose,
// loss} (un-weighted, as the reference logs them).
// ------------------------------------------------------------------------------------------
#define AUX_MAXB 16
struct AuxArgs {
  int mode, B, accumulate;
  float weight;
  const float *rot, *shape, *cam, *joints;        // student pass (shape / cam rows have stride lds / ldc)
  int lds, ldc;
  const float *rot2, *shape2, *cam2, *joints2;    // teacher outputs (mode 0) / history-pass outputs (mode 1: cam2, joints2)
  int lds2, ldc2;
  const float *kp, *kp2;                          // [B][49][3]: frame (mode 1) or exemplar (mode 2) keypoints; history keypoints
  const float *gt_rot, *gt_betas, *gt_s3d;        // mode 2: [B][216], [B][10], [B][24][4]
  float *vals;                                    // [5]
  float *d_rot, *d_shape, *d_cam, *d_joints;
  float *d_cam2, *d_joints2;                      // mode 1
};
__device__ __forceinline__ void aux_project(const float* c, const float* p, float& u, float& v, float& x, float& y, float& z) {
  const float tz = 2.f * FOCAL / (IMG_RES * c[0] + 1e-9f);
  x = p[0] + c[1]; y = p[1] + c[2]; z = p[2] + tz;
  u = FOCAL * (x / z) / (IMG_RES * 0.5f);
  v = FOCAL * (y / z) / (IMG_RES * 0.5f);
}
// block reduction of one value per thread (256 threads) -> every thread gets the total
__device__ __forceinline__ float aux_block_sum(float v, float* sred) {
  v = dyb_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sred[0] + sred[1]) + (sred[2] + sred[3]);
}
__global__ __launch_bounds__(256) void aux_terms_kernel(AuxArgs a, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  if (dyb_rep) {
    a.rot = dyb_rb(a.rot, Rp, dyb_rep); a.shape = dyb_rb(a.shape, Rp, dyb_rep); a.cam = dyb_rb(a.cam, Rp, dyb_rep);
    a.joints = dyb_rb(a.joints, Rp, dyb_rep); a.rot2 = dyb_rb(a.rot2, Rp, dyb_rep); a.shape2 = dyb_rb(a.shape2, Rp, dyb_rep);
    a.cam2 = dyb_rb(a.cam2, Rp, dyb_rep); a.joints2 = dyb_rb(a.joints2, Rp, dyb_rep); a.kp = dyb_rb(a.kp, Rp, dyb_rep);
    a.kp2 = dyb_rb(a.kp2, Rp, dyb_rep); a.gt_rot = dyb_rb(a.gt_rot, Rp, dyb_rep); a.gt_betas = dyb_rb(a.gt_betas, Rp, dyb_rep);
    a.gt_s3d = dyb_rb(a.gt_s3d, Rp, dyb_rep); a.vals = dyb_rb(a.vals, Rp, dyb_rep); a.d_rot = dyb_rb(a.d_rot, Rp, dyb_rep);
    a.d_shape = dyb_rb(a.d_shape, Rp, dyb_rep); a.d_cam = dyb_rb(a.d_cam, Rp, dyb_rep); a.d_joints = dyb_rb(a.d_joints, Rp, dyb_rep);
    a.d_cam2 = dyb_rb(a.d_cam2, Rp, dyb_rep); a.d_joints2 = dyb_rb(a.d_joints2, Rp, dyb_rep);
  }
  __shared__ float sG[AUX_MAXB * NJ49][3];          // projection-side gradient of every student joint (for the per-sample camera sum)
  __shared__ float sG2[AUX_MAXB * NJ49][3];         // same for the history pass (mode 1)
  __shared__ float sHip[AUX_MAXB][3];               // mode 2: sum over joints of the hip-centred 3-D gradient
  __shared__ float sred[4];
  const int t = threadIdx.x, B = a.B, NP = B * NJ49;
  const float w = a.weight;
  const float k2 = FOCAL / (IMG_RES * 0.5f);
  float l2d = 0.f, l3d = 0.f, lsh = 0.f, lpo = 0.f;
  const float n2d = (a.mode == 0) ? (float)(B * NJ49 * 2) : (float)(B * 24 * 2);
  const float n3d = (a.mode == 0) ? (float)(B * NJ49 * 3) : (float)(B * 24 * 3);
  if (a.mode == 2) {
    // hip centres first: pred - (pred[2] + pred[3]) / 2 over the 24 GT-style joints (49-joint indices 25 + j)
    for (int i = t; i < B * 3; i += 256) sHip[i / 3][i % 3] = 0.f;
    __syncthreads();
  }
  for (int i = t; i < NP; i += 256) {
    const int b = i / NJ49, j = i - b * NJ49;
    const float* c = a.cam + (size_t)b * a.ldc;
    const float* p = a.joints + (size_t)i * 3;
    float u, v, x, y, z;
    aux_project(c, p, u, v, x, y, z);
    float gu = 0.f, gv = 0.f;                       // d(weighted term)/d(u, v) of the student
    float g3[3] = {0.f, 0.f, 0.f};                  // direct 3-D part
    float hu = 0.f, hv = 0.f, hx = 0.f, hy = 0.f, hz = 1.f;
    if (a.mode == 0) {
      float tu, tv, tx, ty, tzz;
      aux_project(a.cam2 + (size_t)b * a.ldc2, a.joints2 + (size_t)i * 3, tu, tv, tx, ty, tzz);
      const float du = u - tu, dv = v - tv;
      l2d += du * du + dv * dv;
      gu = w * 5.f * 2.f * du / n2d; gv = w * 5.f * 2.f * dv / n2d;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float d = p[k] - a.joints2[(size_t)i * 3 + k];
        l3d += d * d;
        g3[k] = w * 5.f * 2.f * d / n3d;
      }
    } else if (a.mode == 1) {
      if (j >= 25) {
        aux_project(a.cam2 + (size_t)b * a.ldc2, a.joints2 + (size_t)i * 3, hu, hv, hx, hy, hz);
        const float* kc = a.kp + (size_t)i * 3;
        const float* kh = a.kp2 + (size_t)i * 3;
        const float conf = (kh[2] + kc[2] == 2.f) ? 1.f : 0.f;
        const float eu = (u - hu) - (kc[0] - kh[0]), ev = (v - hv) - (kc[1] - kh[1]);
        l2d += conf * (eu * eu + ev * ev);
        gu = w * 2.f * conf * eu / n2d; gv = w * 2.f * conf * ev / n2d;
      }
    } else {
      if (j >= 25) {
        const float* kc = a.kp + (size_t)i * 3;
        const float conf = kc[2];
        const float du = u - kc[0], dv = v - kc[1];
        l2d += conf * (du * du + dv * dv);
        gu = w * 5.f * 2.f * conf * du / n2d; gv = w * 5.f * 2.f * conf * dv / n2d;
        // hip-centred 3-D term: centres from joints 2, 3 of the 24 (49-joint indices 27, 28) and of the ground truth
        const float* pj = a.joints + (size_t)(b * NJ49 + 27) * 3;
        const float* gt = a.gt_s3d + (size_t)(b * 24 + (j - 25)) * 4;
        const float* g2 = a.gt_s3d + (size_t)(b * 24 + 2) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float pc = 0.5f * (pj[k] + pj[3 + k]), gc = 0.5f * (g2[k] + g2[4 + k]);
          const float d = (p[k] - pc) - (gt[k] - gc);
          l3d += conf * d * d;
          g3[k] = w * 5.f * 2.f * conf * d / n3d;
        }
      }
    }
    // through the projection: (gu, gv) -> (gx, gy, gz) of the point, summed per sample for the camera
    const float gx = gu * k2 / z, gy = gv * k2 / z, gz = -(gx * x + gy * y) / z;
    sG[i][0] = gx; sG[i][1] = gy; sG[i][2] = gz;
    float* dj = a.d_joints + (size_t)i * 3;
    const float o0 = gx + g3[0], o1 = gy + g3[1], o2 = gz + g3[2];
    if (a.accumulate) { dj[0] += o0; dj[1] += o1; dj[2] += o2; }
    else { dj[0] = o0; dj[1] = o1; dj[2] = o2; }
    if (a.mode == 1) {
      const float qx = -gu * k2 / hz, qy = -gv * k2 / hz, qz = -(qx * hx + qy * hy) / hz;
      sG2[i][0] = qx; sG2[i][1] = qy; sG2[i][2] = qz;
      float* d2 = a.d_joints2 + (size_t)i * 3;
      d2[0] = qx; d2[1] = qy; d2[2] = qz;
    }
    if (a.mode == 2 && j >= 25) {                   // the two hip joints also receive -1/2 of every joint's centred gradient
      // (serialised per sample through LDS below)
      sG2[i][0] = g3[0]; sG2[i][1] = g3[1]; sG2[i][2] = g3[2];
    }
  }
  __syncthreads();
  // per-sample sums: camera gradients (and the hip correction of mode 2)
  for (int i = t; i < B * 3; i += 256) {
    const int b = i / 3, k = i % 3;
    float s = 0.f, s2 = 0.f, sh = 0.f;
    for (int j = 0; j < NJ49; ++j) {
      s += sG[b * NJ49 + j][k];
      if (a.mode == 1) s2 += sG2[b * NJ49 + j][k];
      if (a.mode == 2 && j >= 25) sh += sG2[b * NJ49 + j][k];
    }
    sG[b * NJ49][k] = s;                            // reuse row 0 of the sample for its sums
    if (a.mode == 1) sG2[b * NJ49][k] = s2;
    if (a.mode == 2) sHip[b][k] = sh;
  }
  __syncthreads();
  for (int b = t; b < B; b += 256) {
    const float* c = a.cam + (size_t)b * a.ldc;
    const float den = IMG_RES * c[0] + 1e-9f;
    const float dc0 = sG[b * NJ49][2] * (-2.f * FOCAL * IMG_RES / (den * den)), dc1 = sG[b * NJ49][0], dc2 = sG[b * NJ49][1];
    float* d = a.d_cam + (size_t)b * 3;
    if (a.accumulate) { d[0] += dc0; d[1] += dc1; d[2] += dc2; }
    else { d[0] = dc0; d[1] = dc1; d[2] = dc2; }
    if (a.mode == 1) {
      const float* c2 = a.cam2 + (size_t)b * a.ldc2;
      const float den2 = IMG_RES * c2[0] + 1e-9f;
      float* e = a.d_cam2 + (size_t)b * 3;
      e[0] = sG2[b * NJ49][2] * (-2.f * FOCAL * IMG_RES / (den2 * den2));
      e[1] = sG2[b * NJ49][0];
      e[2] = sG2[b * NJ49][1];
    }
    if (a.mode == 2) {
      for (int h = 0; h < 2; ++h)
        for (int k = 0; k < 3; ++k) a.d_joints[(size_t)(b * NJ49 + 27 + h) * 3 + k] -= 0.5f * sHip[b][k];
    }
  }
  // rotmat / shape terms (teacher, labelled exemplar)
  if (a.mode != 1) {
    const float* rt = a.mode == 0 ? a.rot2 : a.gt_rot;
    for (int i = t; i < B * 216; i += 256) {
      const float d = a.rot[i] - rt[i];
      lpo += d * d;
      const float gq = w * 2.f * d / (float)(B * 216);
      if (a.accumulate) a.d_rot[i] += gq; else a.d_rot[i] = gq;
    }
    for (int i = t; i < B * 10; i += 256) {
      const int b = i / 10, k = i % 10;
      const float tv = a.mode == 0 ? a.shape2[(size_t)b * a.lds2 + k] : a.gt_betas[i];
      const float d = a.shape[(size_t)b * a.lds + k] - tv;
      lsh += d * d;
      const float gq = w * 0.001f * 2.f * d / (float)(B * 10);
      if (a.accumulate) a.d_shape[i] += gq; else a.d_shape[i] = gq;
    }
  } else if (!a.accumulate) {
    for (int i = t; i < B * 216; i += 256) a.d_rot[i] = 0.f;
    for (int i = t; i < B * 10; i += 256) a.d_shape[i] = 0.f;
  }
  l2d = aux_block_sum(l2d, sred) / n2d;
  l3d = aux_block_sum(l3d, sred) / n3d;
  lsh = aux_block_sum(lsh, sred) / (float)(B * 10);
  lpo = aux_block_sum(lpo, sred) / (float)(B * 216);
  if (t == 0) {
    if (a.mode == 1) {
      a.vals[0] = l2d; a.vals[1] = a.vals[2] = a.vals[3] = 0.f; a.vals[4] = l2d;
    } else {
      a.vals[0] = l2d; a.vals[1] = l3d; a.vals[2] = lsh; a.vals[3] = lpo;
      a.vals[4] = 5.f * l2d + 5.f * l3d + 0.001f * lsh + lpo;
    }
  }
}
// mode 0 teacher / 1 motion / 2 labelled exemplar (see above).  Unused pointers may be NULL.  B <= 16.
extern "C" int dyb_aux_loss_terms(int mode, int B, int accumulate, float weight, const float* rot, const float* shape, int lds,
                                  const float* cam, int ldc, const float* joints49, const float* rot2, const float* shape2,
                                  int lds2, const float* cam2, int ldc2, const float* joints2, const float* kp, const float* kp2,
                                  const float* gt_rot, const float* gt_betas, const float* gt_s3d, float* vals5, float* d_rot,
                                  float* d_shape, float* d_cam, float* d_joints49, float* d_cam2, float* d_joints2, hipStream_t st) {
  DYB_REQUIRE(mode >= 0 && mode <= 2 && B > 0 && B <= AUX_MAXB, DYB_ERR_UNSUPPORTED);
  DYB_REQUIRE(rot && shape && cam && joints49 && vals5 && d_rot && d_shape && d_cam && d_joints49, DYB_ERR_ARG);
  if (mode == 0) DYB_REQUIRE(rot2 && shape2 && cam2 && joints2, DYB_ERR_ARG);
  if (mode == 1) DYB_REQUIRE(cam2 && joints2 && kp && kp2 && d_cam2 && d_joints2, DYB_ERR_ARG);
  if (mode == 2) DYB_REQUIRE(kp && gt_rot && gt_betas && gt_s3d, DYB_ERR_ARG);
  AuxArgs a{mode, B, accumulate, weight, rot, shape, cam, joints49, lds, ldc, rot2, shape2, cam2, joints2, lds2, ldc2, kp, kp2,
            gt_rot, gt_betas, gt_s3d, vals5, d_rot, d_shape, d_cam, d_joints49, d_cam2, d_joints2};
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(aux_terms_kernel, dim3(1, 1, Rp.n), dim3(256), 0, st, a, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
