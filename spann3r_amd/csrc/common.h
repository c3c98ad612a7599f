// Shared device helpers for the gfx950 kernels of libspann3r_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/spann3r_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define SP3_WAVE 64

void sp3_set_error(const char* fmt, ...);

#define SP3_CHECK(cond, ...)                     \
  do {                                           \
    if (!(cond)) {                               \
      sp3_set_error(__VA_ARGS__);                \
      return 1;                                  \
    }                                            \
  } while (0)

#define SP3_LAUNCH_CHECK(what)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      sp3_set_error("%s: launch failed: %s", what, hipGetErrorString(e_));  \
      return 2;                                                             \
    }                                                                       \
  } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ bf16x8 cvt8(const float4 a, const float4 b) {
  bf16x8 r;
  r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
  r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
  return r;
}

__device__ __forceinline__ float4 relu4(float4 v) {
  v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  return v;
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. at fp32 rounding level for the GELU below): one
// v_rcp_f32, one v_exp_f32 and six FMAs instead of the ~150-instruction branchy libm erff, which made the GELU epilogue of
// an fc1 GEMM cost more shader time than its whole K loop.
__device__ __forceinline__ float erf_fast(float z) {
  const float a = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float y = fmaf(-p * t, __expf(-a * a), 1.0f);
  return copysignf(y, z);
}

// nn.GELU() (approximate='none'): 0.5 x (1 + erf(x / sqrt 2))
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f));
}

// Fragment-order ("packed") operand layout shared by sp3_gemm's a_packed / w_packed and every producer's
// out_packed option: [ceil(rows/16)][ceil(K/KB)][2 halves h][4 lane groups g][16 rows r][CH/2], KB/CH = 64/16 (bf16),
// 32/8 (fp32): lane 16g + r of a wave owns k = kb*KB + g*CH + [0, CH) of row r, and its first and second 16 bytes live in
// two separate 1 KB pieces, so that EVERY operand access of a wave -- a global_load_dwordx4, a global_load_lds piece, a
// ds_read_b128 -- covers one contiguous kilobyte (whole 128-byte lines for the texture addresser, conflict-free LDS banks).
// Element (row, k) lives at the offset below; 4 consecutive k (k % 4 == 0) stay contiguous.
__device__ __forceinline__ int64_t packed_off(int row, int k, int K, bool bf16) {
  const int lkb = bf16 ? 6 : 5, lch = bf16 ? 4 : 3;
  const int nkb = (K + (1 << lkb) - 1) >> lkb;
  const int kb = k >> lkb, kk = k & ((1 << lkb) - 1);
  const int g = kk >> lch, e = kk & ((1 << lch) - 1);
  const int h = e >> (lch - 1), eh = e & ((1 << (lch - 1)) - 1);
  return (((((int64_t)(row >> 4) * nkb + kb) * 2 + h) * 4 + g) * 16 + (row & 15)) * (1 << (lch - 1)) + eh;
}
