// DPT-head helpers: patch im2col, bilinear x2 (align_corners=True), final 1x1 conv + postprocess.
// All feature maps are NHWC fp32 so that 1x1 convs are plain sp3_gemm calls on the token matrix and
// 3x3 convs are sp3_gemm's implicit-GEMM loader (reference: croco/models/dpt_block.py, dust3r/heads).
#include "common.h"
#include <math.h>

namespace {

// out[(b*nh*nw + py_*nw + px_), c*p*p + iy*p + ix] = img[b, c, py_*p+iy, px_*p+ix]  (generic strides)
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ img, int64_t sb, int64_t sc, int64_t sy,
                                                           int64_t sx, int C, int H, int W, int p, void* __restrict__ out,
                                                           int out_bf16, int out_packed, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int K = C * p * p;
  const int64_t row = idx / K;
  const int k = (int)(idx - row * K);
  const int nw = W / p, nh = H / p;
  const int b = (int)(row / (nh * nw));
  const int rem = (int)(row - (int64_t)b * nh * nw);
  const int ty = rem / nw, tx = rem - ty * nw;
  const int c = k / (p * p), r2 = k - c * p * p;
  const int iy = r2 / p, ix = r2 - iy * p;
  const float val = img[b * sb + c * sc + (int64_t)(ty * p + iy) * sy + (int64_t)(tx * p + ix) * sx];
  const int64_t o = out_packed ? packed_off((int)row, k, K, out_bf16 != 0) : idx;
  if (out_bf16) reinterpret_cast<__bf16*>(out)[o] = (__bf16)val;
  else reinterpret_cast<float*>(out)[o] = val;
}

// F.interpolate(scale_factor=2, bilinear, align_corners=True) on NHWC; optional crop to [outH,outW].
// Source index as ATen computes it: scale = (in-1)/(out-1) in float, src = scale * dst.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __bf16* p) {
  const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
}
__device__ __forceinline__ void st4(float* p, const float4 v) { st_out(reinterpret_cast<float4*>(p), v); }
__device__ __forceinline__ void st4(__bf16* p, const float4 v) {
  bf16x4 t;
  t[0] = (__bf16)v.x; t[1] = (__bf16)v.y; t[2] = (__bf16)v.z; t[3] = (__bf16)v.w;
  st_out(reinterpret_cast<bf16x4*>(p), t);
}

// T = float, or __bf16 (bf16 mode of the DPT heads: maps stored as bf16, the interpolation itself in fp32)
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W,
                                                         int C, int outH, int outW, int64_t total4) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int c4 = C >> 2;
  const int cc = (int)(idx % c4);
  int64_t pix = idx / c4;
  const int ox = (int)(pix % outW);
  pix /= outW;
  const int oy = (int)(pix % outH);
  const int b = (int)(pix / outH);
  const int OH = 2 * H, OW = 2 * W;
  const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const float fy = sh * (float)oy, fx = sw * (float)ox;
  int y0 = (int)fy, x0 = (int)fx;
  y0 = y0 < H - 1 ? y0 : H - 1;
  x0 = x0 < W - 1 ? x0 : W - 1;
  const int y1 = y0 < H - 1 ? y0 + 1 : y0, x1 = x0 < W - 1 ? x0 + 1 : x0;
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const T* base = in + (int64_t)b * H * W * C;
  const float4 v00 = ld4(base + (((int64_t)y0 * W + x0) * c4 + cc) * 4);
  const float4 v01 = ld4(base + (((int64_t)y0 * W + x1) * c4 + cc) * 4);
  const float4 v10 = ld4(base + (((int64_t)y1 * W + x0) * c4 + cc) * 4);
  const float4 v11 = ld4(base + (((int64_t)y1 * W + x1) * c4 + cc) * 4);
  float4 o;
  // same association as ATen's upsample_bilinear2d: hy*(hx*v00 + lx*v01) + ly*(hx*v10 + lx*v11)
  o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
  o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
  o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
  o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
  st4(out + idx * 4, o);
}

// bf16 maps with C % 8 == 0 (every map of the DPT heads): 8 channels = 16 bytes per thread, one output row per blockIdx.y (no
// per-thread divisions by the image size), the same fp32 arithmetic per element as above (bit-identical results).  A pure
// bandwidth kernel in front of a convolution: 224 x 224 x 128 went from 11.2 us (8-byte accesses) to the time below.
__global__ __launch_bounds__(256) void upsample2x_bf16x8_kernel(const __bf16* __restrict__ in, __bf16* __restrict__ out, int H, int W,
                                                                int C, int outH, int outW) {
  const int c8 = C >> 3;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= outW * c8) return;
  const int ox = t / c8, cg = t - ox * c8;
  const int oy = blockIdx.y, b = blockIdx.z;
  const int OH = 2 * H, OW = 2 * W;
  const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const float fy = sh * (float)oy, fx = sw * (float)ox;
  int y0 = (int)fy, x0 = (int)fx;
  y0 = y0 < H - 1 ? y0 : H - 1;
  x0 = x0 < W - 1 ? x0 : W - 1;
  const int y1 = y0 < H - 1 ? y0 + 1 : y0, x1 = x0 < W - 1 ? x0 + 1 : x0;
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const __bf16* base = in + (int64_t)b * H * W * C + cg * 8;
  const bf16x8 v00 = *reinterpret_cast<const bf16x8*>(base + ((int64_t)y0 * W + x0) * C);
  const bf16x8 v01 = *reinterpret_cast<const bf16x8*>(base + ((int64_t)y0 * W + x1) * C);
  const bf16x8 v10 = *reinterpret_cast<const bf16x8*>(base + ((int64_t)y1 * W + x0) * C);
  const bf16x8 v11 = *reinterpret_cast<const bf16x8*>(base + ((int64_t)y1 * W + x1) * C);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e)
    o[e] = (__bf16)(hy * (hx * (float)v00[e] + lx * (float)v01[e]) + ly * (hx * (float)v10[e] + lx * (float)v11[e]));
  st_out(reinterpret_cast<bf16x8*>(out + (((int64_t)b * outH + oy) * outW + ox) * C + cg * 8), o);
}

// 8 lanes per pixel: each lane dots C/8 channels against the 4 output filters, xor-shuffle reduce.
template <typename T>
__global__ __launch_bounds__(256) void head_final_kernel(const T* __restrict__ feat, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int64_t pixels, int C,
                                                         float* __restrict__ pts, float* __restrict__ conf,
                                                         float* __restrict__ raw) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t pix = gid >> 3;
  const int sub = (int)(gid & 7);
  const bool valid = pix < pixels;
  const int64_t pp = valid ? pix : pixels - 1;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const T* f = feat + pp * C;
  for (int c = sub * 4; c < C; c += 32) {
    const float4 x = ld4(f + c);
    const float4 w0 = *reinterpret_cast<const float4*>(w + c);
    const float4 w1 = *reinterpret_cast<const float4*>(w + C + c);
    const float4 w2 = *reinterpret_cast<const float4*>(w + 2 * C + c);
    const float4 w3 = *reinterpret_cast<const float4*>(w + 3 * C + c);
    a0 += (x.x * w0.x + x.y * w0.y) + (x.z * w0.z + x.w * w0.w);
    a1 += (x.x * w1.x + x.y * w1.y) + (x.z * w1.z + x.w * w1.w);
    a2 += (x.x * w2.x + x.y * w2.y) + (x.z * w2.z + x.w * w2.w);
    a3 += (x.x * w3.x + x.y * w3.y) + (x.z * w3.z + x.w * w3.w);
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); a3 += __shfl_xor(a3, o);
  }
  if (valid && sub == 0) {
    const float x = a0 + bias[0], y = a1 + bias[1], z = a2 + bias[2], c = a3 + bias[3];
    if (raw) { raw[pix * 4 + 0] = x; raw[pix * 4 + 1] = y; raw[pix * 4 + 2] = z; raw[pix * 4 + 3] = c; }
    // dust3r/heads/postprocess.py:36-46: d = |xyz|; xyz / clip(d, 1e-8) * expm1(d)
    const float d = sqrtf(x * x + y * y + z * z);
    const float sc = expm1f(d) / fmaxf(d, 1e-8f);
    pts[pix * 3 + 0] = x * sc; pts[pix * 3 + 1] = y * sc; pts[pix * 3 + 2] = z * sc;
    conf[pix] = 1.0f + expf(c);                                   // postprocess.py:54
  }
}

// bf16 maps, C = 8 LP with LP a power of two <= 64 (the heads' 128-channel map: 16 lanes per pixel): a lane keeps its 8 channels of the
// four filters in registers and walks PPT pixels with 16-byte loads (the kernel above re-loads 64 bytes of weights next to every 8
// bytes of features); the channel sums meet by xor-shuffles inside the LP-lane group.  224 x 224 x 128: 11.7 us -> see profiles/.
template <int LP, int PPT>
__global__ __launch_bounds__(256) void head_final_bf16x8_kernel(const __bf16* __restrict__ feat, const float* __restrict__ w,
                                                                const float* __restrict__ bias, int64_t pixels, float* __restrict__ pts,
                                                                float* __restrict__ conf, float* __restrict__ raw) {
  constexpr int C = 8 * LP, PPB = 256 / LP;                       // pixels per workgroup per step
  const int li = threadIdx.x % LP, pl = threadIdx.x / LP;
  float wr[4][8];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const float4 a = *reinterpret_cast<const float4*>(w + f * C + li * 8), b = *reinterpret_cast<const float4*>(w + f * C + li * 8 + 4);
    wr[f][0] = a.x; wr[f][1] = a.y; wr[f][2] = a.z; wr[f][3] = a.w; wr[f][4] = b.x; wr[f][5] = b.y; wr[f][6] = b.z; wr[f][7] = b.w;
  }
  const float b0 = bias[0], b1 = bias[1], b2 = bias[2], b3 = bias[3];
  const int64_t p0 = (int64_t)blockIdx.x * PPB * PPT + pl;
  bf16x8 x[PPT];
#pragma unroll
  for (int q = 0; q < PPT; ++q) {
    int64_t pix = p0 + (int64_t)q * PPB;
    pix = pix < pixels ? pix : pixels - 1;
    x[q] = *reinterpret_cast<const bf16x8*>(feat + pix * C + li * 8);
  }
#pragma unroll
  for (int q = 0; q < PPT; ++q) {
    const int64_t pix = p0 + (int64_t)q * PPB;
    float a[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { t0 = fmaf((float)x[q][e], wr[f][e], t0); t1 = fmaf((float)x[q][4 + e], wr[f][4 + e], t1); }
      a[f] = t0 + t1;
    }
#pragma unroll
    for (int o = 1; o < LP; o <<= 1) {
      a[0] += __shfl_xor(a[0], o); a[1] += __shfl_xor(a[1], o); a[2] += __shfl_xor(a[2], o); a[3] += __shfl_xor(a[3], o);
    }
    if (li == 0 && pix < pixels) {
      const float X = a[0] + b0, Y = a[1] + b1, Z = a[2] + b2, Cc = a[3] + b3;
      if (raw) *reinterpret_cast<float4*>(raw + pix * 4) = make_float4(X, Y, Z, Cc);
      const float d = sqrtf(X * X + Y * Y + Z * Z);
      const float sc = expm1f(d) / fmaxf(d, 1e-8f);
      pts[pix * 3 + 0] = X * sc; pts[pix * 3 + 1] = Y * sc; pts[pix * 3 + 2] = Z * sc;
      conf[pix] = 1.0f + expf(Cc);
    }
  }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int sp3_im2col_patch(const float* img, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int B, int C, int H, int W,
                                int p, void* out, int out_bf16, int out_packed, void* stream) {
  SP3_CHECK(img && out && B > 0 && C > 0 && p > 0, "sp3_im2col_patch: bad arguments");
  // same assertion as dust3r/patch_embed.py:22-23
  SP3_CHECK(H % p == 0 && W % p == 0, "Input image size (%d,%d) is not a multiple of patch size (%d)", H, W, p);
  const int64_t total = (int64_t)B * (H / p) * (W / p) * C * p * p;
  hipLaunchKernelGGL(im2col_patch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST(stream), img, sb, sc, sy, sx,
                     C, H, W, p, out, out_bf16, out_packed, total);
  SP3_LAUNCH_CHECK("sp3_im2col_patch");
  return 0;
}

extern "C" int sp3_upsample2x(const float* in, float* out, int B, int H, int W, int C, int outH, int outW, void* stream) {
  SP3_CHECK(in && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "sp3_upsample2x: bad arguments");
  SP3_CHECK(outH > 0 && outH <= 2 * H && outW > 0 && outW <= 2 * W, "sp3_upsample2x: bad crop");
  const int64_t total4 = (int64_t)B * outH * outW * (C / 4);
  hipLaunchKernelGGL(upsample2x_kernel<float>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, ST(stream), in, out, H, W, C,
                     outH, outW, total4);
  SP3_LAUNCH_CHECK("sp3_upsample2x");
  return 0;
}

extern "C" int sp3_upsample2x_bf16(const void* in, void* out, int B, int H, int W, int C, int outH, int outW, void* stream) {
  SP3_CHECK(in && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "sp3_upsample2x_bf16: bad arguments");
  SP3_CHECK(outH > 0 && outH <= 2 * H && outW > 0 && outW <= 2 * W, "sp3_upsample2x_bf16: bad crop");
  if (C % 8 == 0 && outH <= 65535 && B <= 65535) {
    hipLaunchKernelGGL(upsample2x_bf16x8_kernel, dim3((unsigned)((outW * (C / 8) + 255) / 256), outH, B), dim3(256), 0, ST(stream),
                       reinterpret_cast<const __bf16*>(in), reinterpret_cast<__bf16*>(out), H, W, C, outH, outW);
    SP3_LAUNCH_CHECK("sp3_upsample2x_bf16");
    return 0;
  }
  const int64_t total4 = (int64_t)B * outH * outW * (C / 4);
  hipLaunchKernelGGL(upsample2x_kernel<__bf16>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, ST(stream),
                     reinterpret_cast<const __bf16*>(in), reinterpret_cast<__bf16*>(out), H, W, C, outH, outW, total4);
  SP3_LAUNCH_CHECK("sp3_upsample2x_bf16");
  return 0;
}

extern "C" int sp3_head_final(const float* feat, const float* w, const float* b, int64_t pixels, int C, float* pts, float* conf,
                              float* raw, void* stream) {
  SP3_CHECK(feat && w && b && pts && conf && pixels > 0 && C > 0 && C % 32 == 0, "sp3_head_final: bad arguments");
  const int64_t threads = pixels * 8;
  hipLaunchKernelGGL(head_final_kernel<float>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ST(stream), feat, w, b, pixels, C,
                     pts, conf, raw);
  SP3_LAUNCH_CHECK("sp3_head_final");
  return 0;
}

extern "C" int sp3_head_final_bf16(const void* feat, const float* w, const float* b, int64_t pixels, int C, float* pts, float* conf,
                                   float* raw, void* stream) {
  SP3_CHECK(feat && w && b && pts && conf && pixels > 0 && C > 0 && C % 32 == 0, "sp3_head_final_bf16: bad arguments");
  if (C == 128 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 && (!raw || (reinterpret_cast<uintptr_t>(raw) & 15) == 0)) {
    constexpr int LP = 16, PPT = 4, PPB = 256 / LP;
    hipLaunchKernelGGL((head_final_bf16x8_kernel<LP, PPT>), dim3((unsigned)((pixels + PPB * PPT - 1) / (PPB * PPT))), dim3(256), 0, ST(stream),
                       reinterpret_cast<const __bf16*>(feat), w, b, pixels, pts, conf, raw);
    SP3_LAUNCH_CHECK("sp3_head_final_bf16");
    return 0;
  }
  const int64_t threads = pixels * 8;
  hipLaunchKernelGGL(head_final_kernel<__bf16>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ST(stream),
                     reinterpret_cast<const __bf16*>(feat), w, b, pixels, C, pts, conf, raw);
  SP3_LAUNCH_CHECK("sp3_head_final_bf16");
  return 0;
}
