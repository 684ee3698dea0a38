// Frame preprocessing on the device: bounding-box crop -> anti-aliased bilinear resize to 224x224 -> /255 -> per-channel
// normalisation -> CHW, from the decoded RGB frame (uint8, HWC) straight to the tensor HMR.forward consumes.
//
// Replaces, for the test-time path (no augmentation: rot = 0, no flip), reference utils/dataprocess.py:48-96 `crop()` as
// called from boa_dataset/pw3d.py:131-136 (`rgb_processing`) and base_adaptor.py:529-533 (SourceDataset), followed by
// `np.transpose(.., (2,0,1)) / 255.0` and torchvision `Normalize(IMG_NORM_MEAN, IMG_NORM_STD)` (pw3d.py:121-123).
// `crop()` pastes the box (zero outside the frame) into a float image and calls skimage.transform.resize (0.17.2 per the
// reference's requirements.txt) whose defaults are: order 1, mode 'reflect', anti_aliasing on = a Gaussian filter with
// sigma = max(0, (in/out - 1) / 2) per axis (scipy.ndimage.gaussian_filter, truncate 4.0, boundary 'mirror'), then a
// bilinear warp sampling input coordinate (j + 0.5) * in/out - 0.5 with the same mirror boundary.  The integer box
// corners (the reference's `transform(..., invert=1)` arithmetic) are host logic and arrive as arguments.
// Parity status: box / paste / normalise are pinned to the reference's crop() (golden g7); the resize is written from the
// documented skimage defaults above and checked against the oracle's restatement of them + known answers - skimage itself is
// absent from the build image, so resize parity against scikit-image 0.17.2 is unverified (tests/test_preprocess.py header).
//
// Three small kernels (the Gaussian is separable; axis 0 first, as scipy does): blur along rows reading the frame with
// the zero fill of the paste, blur along columns, bilinear + normalise.  fp32 throughout (the reference computes the
// filter in float64 and casts to float32 afterwards; agreement 1e-5 on the normalised values, tests).
#include "dyb_common.h"

#define CROP_MAX_TAPS 129
struct CropTaps {
  float w[CROP_MAX_TAPS];
  int radius;              // taps = 2 * radius + 1; radius 0 = no filtering along this axis
};
// scipy 'mirror' / skimage 'reflect' ("d c b | a b c d | c b a"): whole-sample symmetric, period 2 (n - 1)
__device__ __forceinline__ int mirror_idx(int i, int n) {
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i = i % p;
  if (i < 0) i += p;
  return i < n ? i : p - i;
}
// pasted crop value at (y, x) of the box: the frame pixel, or 0 outside the frame
__device__ __forceinline__ float box_px(const uint8_t* img, int H, int W, int ul_x, int ul_y, int y, int x, int c) {
  const int yy = y + ul_y, xx = x + ul_x;
  return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)img[((size_t)yy * W + xx) * 3 + c] : 0.f;
}
__global__ __launch_bounds__(256) void crop_blur_rows_kernel(const uint8_t* __restrict__ img, int H, int W, int ul_x, int ul_y,
                                                             int ch, int cw, CropTaps t, float* __restrict__ out) {
  const size_t total = (size_t)ch * cw * 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % 3);
    const int x = (int)((i / 3) % cw), y = (int)(i / ((size_t)3 * cw));
    float s = 0.f;
    for (int k = -t.radius; k <= t.radius; ++k) s += t.w[k + t.radius] * box_px(img, H, W, ul_x, ul_y, mirror_idx(y + k, ch), x, c);
    out[i] = s;
  }
}
__global__ __launch_bounds__(256) void crop_blur_cols_kernel(const float* __restrict__ in, int ch, int cw, CropTaps t,
                                                             float* __restrict__ out) {
  const size_t total = (size_t)ch * cw * 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % 3);
    const int x = (int)((i / 3) % cw), y = (int)(i / ((size_t)3 * cw));
    float s = 0.f;
    for (int k = -t.radius; k <= t.radius; ++k) s += t.w[k + t.radius] * in[((size_t)y * cw + mirror_idx(x + k, cw)) * 3 + c];
    out[i] = s;
  }
}
// out[c][j][i] = (bilinear(blurred box, r(j), q(i)) / 255 - mean[c]) / std[c]
__global__ __launch_bounds__(256) void crop_resize_norm_kernel(const float* __restrict__ in, int ch, int cw, int res, float fr,
                                                               float fc, float m0, float m1, float m2, float s0, float s1,
                                                               float s2, float* __restrict__ out) {
  const int total = res * res * 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int c = i / (res * res), j = (i / res) % res, q = i % res;
    const float r = fr * ((float)j + 0.5f) - 0.5f, col = fc * ((float)q + 0.5f) - 0.5f;
    const float r0f = floorf(r), c0f = floorf(col);
    const int r0 = mirror_idx((int)r0f, ch), r1 = mirror_idx((int)ceilf(r), ch);
    const int c0 = mirror_idx((int)c0f, cw), c1 = mirror_idx((int)ceilf(col), cw);
    const float dr = r - r0f, dc = col - c0f;
    const float top = (1.f - dc) * in[((size_t)r0 * cw + c0) * 3 + c] + dc * in[((size_t)r0 * cw + c1) * 3 + c];
    const float bot = (1.f - dc) * in[((size_t)r1 * cw + c0) * 3 + c] + dc * in[((size_t)r1 * cw + c1) * 3 + c];
    const float v = ((1.f - dr) * top + dr * bot) / 255.0f;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[i] = (v - mean) / sd;
  }
}

// scipy.ndimage.gaussian_filter1d weights for `sigma` (truncate 4.0): radius int(4 sigma + 0.5), exp(-x^2 / 2 sigma^2), sum 1
static int make_taps(double sigma, CropTaps& t) {
  t.radius = 0;
  t.w[0] = 1.f;
  if (sigma <= 1e-15) return DYB_OK;
  const int r = (int)(4.0 * sigma + 0.5);
  DYB_REQUIRE(2 * r + 1 <= CROP_MAX_TAPS, DYB_ERR_UNSUPPORTED);
  double w[CROP_MAX_TAPS], s = 0.0;
  for (int k = -r; k <= r; ++k) { w[k + r] = exp(-0.5 / (sigma * sigma) * (double)k * (double)k); s += w[k + r]; }
  for (int k = 0; k <= 2 * r; ++k) t.w[k] = (float)(w[k] / s);
  t.radius = r;
  return DYB_OK;
}

// scratch: two float images of the box size
extern "C" size_t dyb_crop_workspace_bytes(int box_h, int box_w) {
  if (box_h <= 0 || box_w <= 0) return 0;
  return 2 * (((size_t)box_h * box_w * 3 + 63) & ~(size_t)63) * sizeof(float);
}
// img: decoded frame [H][W][3] uint8 RGB on the device; (ul_x, ul_y) / (br_x, br_y): upper-left / bottom-right corner of the
// box in frame pixels as the reference computes them (dataprocess.py:51-54; may lie outside the frame - zero fill);
// out: [3][res][res] fp32, normalised with mean / std per channel.
extern "C" int dyb_crop_resize_normalize(const uint8_t* img, int H, int W, int ul_x, int ul_y, int br_x, int br_y, float* out,
                                         int res, float mean0, float mean1, float mean2, float std0, float std1, float std2,
                                         void* ws, size_t ws_bytes, hipStream_t st) {
  DYB_REQUIRE(img && out && ws && H > 0 && W > 0 && res > 0, DYB_ERR_ARG);
  const int ch = br_y - ul_y, cw = br_x - ul_x;
  DYB_REQUIRE(ch > 0 && cw > 0, DYB_ERR_ARG);
  DYB_REQUIRE(ws_bytes >= dyb_crop_workspace_bytes(ch, cw), DYB_ERR_WORKSPACE);
  const double fr = (double)ch / res, fc = (double)cw / res;
  CropTaps t0, t1;
  int rc = make_taps(fr > 1.0 ? (fr - 1.0) / 2.0 : 0.0, t0);
  if (rc != DYB_OK) return rc;
  rc = make_taps(fc > 1.0 ? (fc - 1.0) / 2.0 : 0.0, t1);
  if (rc != DYB_OK) return rc;
  float* a = reinterpret_cast<float*>(ws);
  float* b = a + (((size_t)ch * cw * 3 + 63) & ~(size_t)63);
  const size_t n = (size_t)ch * cw * 3;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(crop_blur_rows_kernel, dim3(blocks), dim3(256), 0, st, img, H, W, ul_x, ul_y, ch, cw, t0, a);
  DYB_CHECK_LAUNCH();
  hipLaunchKernelGGL(crop_blur_cols_kernel, dim3(blocks), dim3(256), 0, st, (const float*)a, ch, cw, t1, b);
  DYB_CHECK_LAUNCH();
  hipLaunchKernelGGL(crop_resize_norm_kernel, dim3(dyb_cdiv(res * res * 3, 256)), dim3(256), 0, st, (const float*)b, ch, cw, res,
                     (float)fr, (float)fc, mean0, mean1, mean2, std0, std1, std2, out);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
