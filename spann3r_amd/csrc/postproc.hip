// Post-forward geometry on the device (reference: demo.py:147-215, dust3r/post_process.py:12-60):
//   * focal length of the first camera from its pointmap, the reference's 'weiszfeld' estimator (closed-form L2 start + 10
//     re-weighted least-squares rounds) and its 'median' estimator's inputs are not needed by demo.py;
//   * confidence filtering of the reconstructed cloud: conf_sig = (conf - 1) / conf > thresh, points and colours compacted
//     in pixel order (what the boolean indexing of demo.py:209-211 produces) without a host round trip per frame.
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ double block_sum_d(double v, double* sh, int nw) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

// one workgroup per image: focal = argmin sum |pixel - focal * (x, y) / z|  (post_process.py:38-53)
__global__ __launch_bounds__(1024) void focal_weiszfeld_kernel(const float* __restrict__ pts, int H, int W, float ppx, float ppy, int iters,
                                                               float fmin, float fmax, float* __restrict__ focal) {
  __shared__ double sh[16];
  const int HW = H * W;
  const float* p = pts + (int64_t)blockIdx.x * HW * 3;
  // xy_over_z with nan_to_num(posinf = 0, neginf = 0) -- NaN (0/0) also becomes 0
  auto xyz = [&](int i, float& ax, float& ay, float& u, float& v) {
    const float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
    ax = x / z; ay = y / z;
    if (!isfinite(ax)) ax = 0.f;
    if (!isfinite(ay)) ay = 0.f;
    u = (float)(i % W) - ppx; v = (float)(i / W) - ppy;
  };
  double s_px = 0.0, s_xx = 0.0;
  for (int i = threadIdx.x; i < HW; i += 1024) {
    float ax, ay, u, v; xyz(i, ax, ay, u, v);
    s_px += (double)(ax * u + ay * v);
    s_xx += (double)(ax * ax + ay * ay);
  }
  s_px = block_sum_d(s_px, sh, 16); s_xx = block_sum_d(s_xx, sh, 16);
  float f = (float)((s_px / HW) / (s_xx / HW));
  for (int it = 0; it < iters; ++it) {
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < HW; i += 1024) {
      float ax, ay, u, v; xyz(i, ax, ay, u, v);
      const float dx = u - f * ax, dy = v - f * ay;
      const float w = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy), 1e-8f);
      a += (double)(w * (ax * u + ay * v));
      b += (double)(w * (ax * ax + ay * ay));
    }
    a = block_sum_d(a, sh, 16); b = block_sum_d(b, sh, 16);
    f = (float)((a / HW) / (b / HW));
  }
  if (threadIdx.x == 0) focal[blockIdx.x] = fminf(fmaxf(f, fmin), fmax);
}

// ---- stream compaction of the confident points, pixel order preserved: counts per 1024-element chunk, exclusive scan of the
// chunk counts (one workgroup), scatter.
__device__ __forceinline__ bool keep_conf(float c, float thr) { return (c - 1.0f) / c > thr; }

__global__ __launch_bounds__(256) void conf_count_kernel(const float* __restrict__ conf, int64_t n, float thr, int* __restrict__ counts) {
  __shared__ int sh[4];
  const int64_t base = (int64_t)blockIdx.x * 1024;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + threadIdx.x * 4 + k;
    c += (i < n && keep_conf(conf[i], thr)) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(1024) void scan_counts_kernel(int* __restrict__ counts, int nblk, int64_t* __restrict__ total) {
  __shared__ int sh[1024];
  int64_t carry = 0;
  for (int b0 = 0; b0 < nblk; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int v = i < nblk ? counts[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                 // Hillis-Steele inclusive scan
      const int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblk) counts[i] = (int)(carry + sh[threadIdx.x] - v);      // exclusive offset (clouds stay below 2^31 points)
    carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void conf_scatter_kernel(const float* __restrict__ conf, const float* __restrict__ pts, const float* __restrict__ rgb,
                                                           int64_t n, float thr, const int* __restrict__ offs, float* __restrict__ out_pts,
                                                           float* __restrict__ out_rgb) {
  __shared__ int wsum[4];
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  bool k[4];
  int c = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) { k[e] = (base + e < n) && keep_conf(conf[base + e], thr); c += k[e]; }
  // exclusive prefix of c over the workgroup's 256 threads (thread order = pixel order)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int pos = offs[blockIdx.x] + inc - c;
  for (int q = 0; q < w; ++q) pos += wsum[q];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (!k[e]) continue;
    const int64_t i = base + e;
    out_pts[3 * (int64_t)pos] = pts[3 * i]; out_pts[3 * (int64_t)pos + 1] = pts[3 * i + 1]; out_pts[3 * (int64_t)pos + 2] = pts[3 * i + 2];
    if (rgb) { out_rgb[3 * (int64_t)pos] = rgb[3 * i]; out_rgb[3 * (int64_t)pos + 1] = rgb[3 * i + 1]; out_rgb[3 * (int64_t)pos + 2] = rgb[3 * i + 2]; }
    ++pos;
  }
}

}  // namespace

extern "C" int sp3_focal_weiszfeld(const float* pts3d, int B, int H, int W, float ppx, float ppy, int iters, float focal_min, float focal_max,
                                   float* focal, void* stream) {
  SP3_CHECK(pts3d && focal && B > 0 && H > 0 && W > 0 && iters >= 0, "sp3_focal_weiszfeld: bad arguments");
  hipLaunchKernelGGL(focal_weiszfeld_kernel, dim3(B), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), pts3d, H, W, ppx, ppy, iters,
                     focal_min, focal_max, focal);
  SP3_LAUNCH_CHECK("sp3_focal_weiszfeld");
  return 0;
}

// scratch: ceil(n / 1024) ints; total (device int64) receives the number of kept points; out_pts / out_rgb hold up to n rows
extern "C" int sp3_conf_filter(const float* conf, const float* pts, const float* rgb, int64_t n, float thresh, int* scratch, int64_t* total,
                               float* out_pts, float* out_rgb, void* stream) {
  SP3_CHECK(conf && pts && scratch && total && out_pts && n > 0 && (rgb == nullptr || out_rgb != nullptr), "sp3_conf_filter: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nblk = (int)((n + 1023) / 1024);
  hipLaunchKernelGGL(conf_count_kernel, dim3(nblk), dim3(256), 0, st, conf, n, thresh, scratch);
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, st, scratch, nblk, total);
  hipLaunchKernelGGL(conf_scatter_kernel, dim3(nblk), dim3(256), 0, st, conf, pts, rgb, n, thresh, scratch, out_pts, out_rgb);
  SP3_LAUNCH_CHECK("sp3_conf_filter");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Camera pose of a frame from its pointmap (demo.py:170-186 calls cv2.solvePnPRansac with all H*W pixel <-> point pairs).
// The device does the O(H*W) part of a calibrated PnP as fixed-order double-precision reductions, one workgroup per frame;
// the host solves the 12x12 / 6x6 systems (spann3r_amd/postprocess.py):
//   pnp_dlt_accum: the normal matrix of the calibrated DLT  [X~ 0 -x X~; 0 X~ -y X~] p = 0  over the (inlier) points, as the
//     four symmetric 4x4 blocks S = sum w X~X~^T, Sx = sum w x X~X~^T, Sy, Sr = sum w (x^2 + y^2) X~X~^T (X~ = Hartley-
//     normalised homogeneous point, (x, y) = normalised pixel) -> 40 doubles + the inlier count;
//   pnp_gn_accum: Gauss-Newton normal equations of the reprojection error in pixels over the inliers (error < thr):
//     H (21 unique), g (6), cost, count -> 29 doubles.
namespace {

__device__ __forceinline__ void block_sum_n(double* v, int n, double* sh, double* out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = 0; k < n; ++k) {
    double x = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) sh[w * 48 + k] = x;
  }
  __syncthreads();
  if ((int)threadIdx.x < n) {
    double r = 0.0;
    for (int i = 0; i < 16; ++i) r += sh[i * 48 + threadIdx.x];
    out[threadIdx.x] = r;
  }
}

// reprojection error (pixels) of point X under pose Rt (row-major R | t), or a large value for points behind the camera
__device__ __forceinline__ float reproj_err(const float* Rt, float X, float Y, float Z, float f, float u, float v, float cx, float cy,
                                            float& xc, float& yc, float& zc) {
  xc = Rt[0] * X + Rt[1] * Y + Rt[2] * Z + Rt[9];
  yc = Rt[3] * X + Rt[4] * Y + Rt[5] * Z + Rt[10];
  zc = Rt[6] * X + Rt[7] * Y + Rt[8] * Z + Rt[11];
  if (!(zc > 1e-9f)) return 1e30f;
  const float du = f * xc / zc + cx - u, dv = f * yc / zc + cy - v;
  return sqrtf(du * du + dv * dv);
}

// OpenCV's consensus test (calib3d/src/solvepnp.cpp PnPRansacCallback::computeError + ptsetreg.cpp findInliers): the point is projected in
// double precision, NO test of the sign of the depth (cvProjectPoints2: z = z ? 1 / z : 1), the squared pixel error is rounded to
// float and compared with (float)(thresh^2) by <=.
__device__ __forceinline__ bool cv_inlier(const float* Rt, float X, float Y, float Z, float f, float u, float v, float cx, float cy, float thr) {
  const double xc = (double)Rt[0] * X + (double)Rt[1] * Y + (double)Rt[2] * Z + Rt[9];
  const double yc = (double)Rt[3] * X + (double)Rt[4] * Y + (double)Rt[5] * Z + Rt[10];
  double zc = (double)Rt[6] * X + (double)Rt[7] * Y + (double)Rt[8] * Z + Rt[11];
  zc = zc != 0.0 ? 1.0 / zc : 1.0;
  const double du = (double)f * xc * zc + cx - u, dv = (double)f * yc * zc + cy - v;
  return (float)(du * du + dv * dv) <= thr * thr;
}

__global__ __launch_bounds__(1024) void pnp_dlt_accum_kernel(const float* __restrict__ pts, int HW, int W, float f, float cx, float cy,
                                                             const float* __restrict__ norm4, const float* __restrict__ Rt_all, float thr,
                                                             int cv_mode, double* __restrict__ out) {
  __shared__ double sh[16 * 48];
  const float* p = pts + (int64_t)blockIdx.x * HW * 3;
  const float* nm = norm4 + blockIdx.x * 4;                 // centroid (3), scale
  const float* Rt = Rt_all ? Rt_all + blockIdx.x * 12 : nullptr;
  double a[41];
  for (int k = 0; k < 41; ++k) a[k] = 0.0;
  for (int i = threadIdx.x; i < HW; i += 1024) {
    const float X = p[3 * i], Y = p[3 * i + 1], Z = p[3 * i + 2];
    if (!(isfinite(X) && isfinite(Y) && isfinite(Z))) continue;
    const float u = (float)(i % W), v = (float)(i / W);
    if (Rt) {
      float xc, yc, zc;
      if (cv_mode ? !cv_inlier(Rt, X, Y, Z, f, u, v, cx, cy, thr) : reproj_err(Rt, X, Y, Z, f, u, v, cx, cy, xc, yc, zc) >= thr) continue;
    }
    const double x = (double)(u - cx) / f, y = (double)(v - cy) / f;
    const double h[4] = {(double)(X - nm[0]) * nm[3], (double)(Y - nm[1]) * nm[3], (double)(Z - nm[2]) * nm[3], 1.0};
    const double rr = x * x + y * y;
    int k = 0;
    for (int r = 0; r < 4; ++r)
      for (int c = r; c < 4; ++c, ++k) {
        const double hh = h[r] * h[c];
        a[k] += hh; a[10 + k] += x * hh; a[20 + k] += y * hh; a[30 + k] += rr * hh;
      }
    a[40] += 1.0;
  }
  block_sum_n(a, 41, sh, out + (int64_t)blockIdx.x * 41);
}

__global__ __launch_bounds__(1024) void pnp_gn_accum_kernel(const float* __restrict__ pts, int HW, int W, float f, float cx, float cy,
                                                            const double* __restrict__ Rt_all, const float* __restrict__ Rt_mask_all, float thr,
                                                            double* __restrict__ out) {
  __shared__ double sh[16 * 48];
  const float* p = pts + (int64_t)blockIdx.x * HW * 3;
  const double* Rt = Rt_all + blockIdx.x * 12;             // the pose under refinement, in double: its float rounding would be the fixed point's error
  float Rf[12];
  for (int k = 0; k < 12; ++k) Rf[k] = (float)Rt[k];
  const float* Rm = Rt_mask_all ? Rt_mask_all + blockIdx.x * 12 : nullptr;   // fixed consensus set (OpenCV): the inliers of THIS pose
  double a[29];
  for (int k = 0; k < 29; ++k) a[k] = 0.0;
  for (int i = threadIdx.x; i < HW; i += 1024) {
    const float X = p[3 * i], Y = p[3 * i + 1], Z = p[3 * i + 2];
    if (!(isfinite(X) && isfinite(Y) && isfinite(Z))) continue;
    const float u = (float)(i % W), v = (float)(i / W);
    double xc, yc, zc;
    if (Rm) {
      if (!cv_inlier(Rm, X, Y, Z, f, u, v, cx, cy, thr)) continue;
      xc = Rt[0] * X + Rt[1] * Y + Rt[2] * Z + Rt[9];
      yc = Rt[3] * X + Rt[4] * Y + Rt[5] * Z + Rt[10];
      zc = Rt[6] * X + Rt[7] * Y + Rt[8] * Z + Rt[11];
      if (zc == 0.0) zc = 1.0;
    } else {
      float xf, yf, zf;
      if (reproj_err(Rf, X, Y, Z, f, u, v, cx, cy, xf, yf, zf) >= thr) continue;
      xc = xf; yc = yf; zc = zf;
    }
    const double iz = 1.0 / zc, xn = xc * iz, yn = yc * iz;
    const double ru = f * xn + cx - u, rv = f * yn + cy - v;
    // d(residual) / d(omega, delta) for Xc' = Xc + omega x Xc + delta
    const double fz = f * iz;
    const double Ju[6] = {fz * (-xn * yc), fz * (zc + xn * xc), fz * (-yc), fz, 0.0, -fz * xn};
    const double Jv[6] = {fz * (-zc - yn * yc), fz * (yn * xc), fz * xc, 0.0, fz, -fz * yn};
    int k = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c, ++k) a[k] += Ju[r] * Ju[c] + Jv[r] * Jv[c];
    for (int r = 0; r < 6; ++r) a[21 + r] += Ju[r] * ru + Jv[r] * rv;
    a[27] += ru * ru + rv * rv;
    a[28] += 1.0;
  }
  block_sum_n(a, 29, sh, out + (int64_t)blockIdx.x * 29);
}

}  // namespace

extern "C" int sp3_pnp_dlt_accum(const float* pts, int F, int H, int W, float focal, float cx, float cy, const float* norm4,
                                 const float* Rt, float thresh, int cv_mode, double* out41, void* stream) {
  SP3_CHECK(pts && norm4 && out41 && F > 0 && H > 0 && W > 0 && focal > 0.f, "sp3_pnp_dlt_accum: bad arguments");
  hipLaunchKernelGGL(pnp_dlt_accum_kernel, dim3(F), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), pts, H * W, W, focal, cx, cy, norm4,
                     Rt, thresh, cv_mode, out41);
  SP3_LAUNCH_CHECK("sp3_pnp_dlt_accum");
  return 0;
}

extern "C" int sp3_pnp_gn_accum(const float* pts, int F, int H, int W, float focal, float cx, float cy, const double* Rt, const float* Rt_mask,
                                float thresh, double* out29, void* stream) {
  SP3_CHECK(pts && Rt && out29 && F > 0 && H > 0 && W > 0 && focal > 0.f, "sp3_pnp_gn_accum: bad arguments");
  hipLaunchKernelGGL(pnp_gn_accum_kernel, dim3(F), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), pts, H * W, W, focal, cx, cy, Rt,
                     Rt_mask, thresh, out29);
  SP3_LAUNCH_CHECK("sp3_pnp_gn_accum");
  return 0;
}

// number of points whose reprojection error under hypothesis h of frame f is below thr: counts[f][h]; grid (hyps, frames)
namespace {
__global__ __launch_bounds__(256) void pnp_score_kernel(const float* __restrict__ pts, int HW, int W, float f, float cx, float cy,
                                                        const float* __restrict__ Rt_all, int nh, float thr, int cv_mode, int* __restrict__ counts) {
  __shared__ int sh[4];
  const float* p = pts + (int64_t)blockIdx.y * HW * 3;
  const float* Rt = Rt_all + ((int64_t)blockIdx.y * nh + blockIdx.x) * 12;
  float r[12];
  for (int k = 0; k < 12; ++k) r[k] = Rt[k];
  int c = 0;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float X = p[3 * i], Y = p[3 * i + 1], Z = p[3 * i + 2];
    if (!(isfinite(X) && isfinite(Y) && isfinite(Z))) continue;
    float xc, yc, zc;
    if (cv_mode) c += cv_inlier(r, X, Y, Z, f, (float)(i % W), (float)(i / W), cx, cy, thr);
    else c += reproj_err(r, X, Y, Z, f, (float)(i % W), (float)(i / W), cx, cy, xc, yc, zc) < thr;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.y * nh + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
}  // namespace

extern "C" int sp3_pnp_score(const float* pts, int F, int H, int W, float focal, float cx, float cy, const float* Rt, int n_hyp, float thresh,
                             int cv_mode, int* counts, void* stream) {
  SP3_CHECK(pts && Rt && counts && F > 0 && H > 0 && W > 0 && n_hyp > 0 && focal > 0.f, "sp3_pnp_score: bad arguments");
  hipLaunchKernelGGL(pnp_score_kernel, dim3(n_hyp, F), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pts, H * W, W, focal, cx, cy, Rt,
                     n_hyp, thresh, cv_mode, counts);
  SP3_LAUNCH_CHECK("sp3_pnp_score");
  return 0;
}
