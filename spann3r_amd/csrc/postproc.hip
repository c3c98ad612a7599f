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
