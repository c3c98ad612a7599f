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

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// Fragment-order ("packed") operand layout shared by sp3_gemm's a_packed / w_packed and every producer's
// out_packed option: [ceil(rows/16)][ceil(K/KB)][4 lane groups g][16 rows r][CH], KB/CH = 64/16 (bf16), 32/8 (fp32).
// Element (row, k) lives at the offset below; 4 consecutive k (k % 4 == 0) stay contiguous.
__device__ __forceinline__ int64_t packed_off(int row, int k, int K, bool bf16) {
  const int lkb = bf16 ? 6 : 5, lch = bf16 ? 4 : 3;
  const int nkb = (K + (1 << lkb) - 1) >> lkb;
  const int kb = k >> lkb, kk = k & ((1 << lkb) - 1);
  const int g = kk >> lch, e = kk & ((1 << lch) - 1);
  return ((((int64_t)(row >> 4) * nkb + kb) * 4 + g) * 16 + (row & 15)) * (1 << lch) + e;
}
