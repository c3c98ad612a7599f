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

// nn.GELU() (approximate='none') = x Phi(x), by the normal tail of Abramowitz & Stegun 26.2.17:
//   x Phi(x) = max(x, 0) - |x| Q(|x|),   Q(a) = phi(a) (b1 t + .. + b5 t^5),   t = 1 / (1 + 0.2316419 a),   |error of Q| < 7.5e-8
// one v_rcp_f32, one v_exp_f32 and ten FMA-class ops: measured against float64 over [-12, 12] the absolute error stays below
// 3.4e-7 (the 7.1.26 erf form this replaces: 4.7e-7 with four more VALU ops; libm erff is a ~150-instruction branchy routine that
// made the GELU epilogue of an fc1 GEMM cost more shader time than its K loop).  NaN and +-inf inputs give NaN.
__device__ __forceinline__ float gelu_erf(float x) {
  const float a = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.2316419f, a, 1.0f));
  float p = fmaf(1.330274429f, t, -1.821255978f);
  p = fmaf(p, t, 1.781477937f);
  p = fmaf(p, t, -0.356563782f);
  p = fmaf(p, t, 0.319381530f);
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * a * a);      // exp(-a^2 / 2)
  const float q = (p * t) * (e * 0.3989422804014327f);                           // / sqrt(2 pi)
  return fmaf(-a, q, fmaxf(x, 0.f));
}

// Write-through stores for GEMM / convolution epilogues (global_store ... sc1).  Every workgroup of a one-round launch reaches
// its epilogue at the same time, so with plain stores the whole output (16 MB for an encoder fc1) sits dirty in the XCD L2s
// when the kernel ends and the release at the launch boundary writes it back while nothing runs: measured 1.9 us of a 25 us
// launch (tools/ubench/gemm_bm.hip, profiles/r05_gemm_manyrow_structure_probe_*.txt).  sc1 stores go through to the memory side as they are issued --
// under the epilogue's own VALU work -- and leave nothing to flush; the consumer is another launch (its L2s start clean anyway).
// -DSP3_NO_WT_STORES builds the plain form (A/B).
#ifndef SP3_NO_WT_STORES
__device__ __forceinline__ void st_out(bf16x4* p, bf16x4 v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_out(float2* p, float2 v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_out(float4* p, float4 v) {
  // (no builtin for a 16-byte sc1 store without a buffer descriptor; s_nop 1: the data registers may be reused right behind it)
  const f32x4 t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_out(bf16x8* p, bf16x8 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
#else
template <typename T> __device__ __forceinline__ void st_out(T* p, T v) { *p = v; }
#endif

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
