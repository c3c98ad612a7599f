// Input pipeline kernels: decoded RGB frames (uint8, HWC) -> the normalised, landscape-rectified
// float image Spann3R.forward consumes.  Replaces, on the device, the per-image CPU work of the reference's dataset path:
// PIL Image.crop / Image.resize(LANCZOS) (dust3r/datasets/utils/cropping.py:54-111), ImgNorm = ToTensor + Normalize(0.5, 0.5)
// (dust3r/utils/image.py:23) and transpose_to_landscape (dust3r/datasets/base/base_stereo_view_dataset.py:215-220).
// The resize reproduces Pillow's 8-bit resampler bit for bit (src/libImaging/Resample.c): two separable passes with an 8-bit
// intermediate image, coefficients normalised in double and rounded to 22-bit fixed point ON THE HOST
// (spann3r_amd/preprocess.py), int32 accumulation from 2^21, arithmetic shift, clamp.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// pass 1 (horizontal) on the crop window [t, t+H1) x [l, l+W1) of the source: tmp[y][xo][c], one thread per (y, xo)
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, int64_t row_stride, int l, int t, int H1,
                                                         int W2, const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk,
                                                         int ksize, uint8_t* __restrict__ tmp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)H1 * W2) return;
  const int y = (int)(i / W2), xo = (int)(i - (int64_t)y * W2);
  const int x0 = bounds[2 * xo], n = bounds[2 * xo + 1];
  const int32_t* k = kk + (int64_t)xo * ksize;
  const uint8_t* p = src + (int64_t)(y + t) * row_stride + (int64_t)(l + x0) * 3;
  int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
  for (int x = 0; x < n; ++x) {
    const int w = k[x];
    a0 += p[3 * x] * w; a1 += p[3 * x + 1] * w; a2 += p[3 * x + 2] * w;
  }
  uint8_t* o = tmp + i * 3;
  o[0] = (uint8_t)clip8(a0); o[1] = (uint8_t)clip8(a1); o[2] = (uint8_t)clip8(a2);
}

// pass 2 (vertical) + final crop + ToTensor / Normalize + optional transpose: out fp32 [3][outH][outW] (or [3][outW][outH])
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ tmp, int W2, const int32_t* __restrict__ bounds,
                                                              const int32_t* __restrict__ kk, int ksize, int l2, int t2, int outW,
                                                              int outH, int transpose, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)outH * outW) return;
  const int yo = (int)(i / outW), xo = (int)(i - (int64_t)yo * outW);
  const int yr = t2 + yo;                                   // row of the resized image
  const int y0 = bounds[2 * yr], n = bounds[2 * yr + 1];
  const int32_t* k = kk + (int64_t)yr * ksize;
  const uint8_t* p = tmp + ((int64_t)y0 * W2 + (l2 + xo)) * 3;
  int a[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
  for (int y = 0; y < n; ++y) {
    const int w = k[y];
    const uint8_t* q = p + (int64_t)y * W2 * 3;
    a[0] += q[0] * w; a[1] += q[1] * w; a[2] += q[2] * w;
  }
  const int64_t plane = (int64_t)outH * outW;
  const int64_t o = transpose ? ((int64_t)xo * outH + yo) : ((int64_t)yo * outW + xo);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float u = __fdiv_rn((float)clip8(a[c]), 255.0f);  // ToTensor: uint8 / 255 (correctly rounded, as torch.div)
    out[c * plane + o] = (u - 0.5f) / 0.5f;                 // Normalize(0.5, 0.5)
  }
}

}  // namespace

extern "C" int sp3_preprocess_image(const uint8_t* src, int64_t src_row_stride, int crop_l, int crop_t, int H1, int W1,
                                    const int32_t* hbounds, const int32_t* hcoef, int hksize, int W2,
                                    const int32_t* vbounds, const int32_t* vcoef, int vksize, int H2,
                                    int crop2_l, int crop2_t, int outW, int outH, int transpose, uint8_t* tmp, float* out,
                                    void* stream) {
  SP3_CHECK(src && hbounds && hcoef && vbounds && vcoef && tmp && out, "sp3_preprocess_image: null pointer");
  SP3_CHECK(H1 > 0 && W1 > 0 && W2 > 0 && H2 > 0 && outW > 0 && outH > 0 && hksize > 0 && vksize > 0,
            "sp3_preprocess_image: bad geometry");
  SP3_CHECK(crop2_l >= 0 && crop2_t >= 0 && crop2_l + outW <= W2 && crop2_t + outH <= H2,
            "sp3_preprocess_image: final crop (%d,%d)+(%dx%d) outside the resized image %dx%d", crop2_l, crop2_t, outW, outH, W2, H2);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n1 = (int64_t)H1 * W2, n2 = (int64_t)outH * outW;
  hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, src, src_row_stride, crop_l, crop_t, H1,
                     W2, hbounds, hcoef, hksize, tmp);
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, tmp, W2, vbounds, vcoef, vksize,
                     crop2_l, crop2_t, outW, outH, transpose, out);
  SP3_LAUNCH_CHECK("sp3_preprocess_image");
  return 0;
}
