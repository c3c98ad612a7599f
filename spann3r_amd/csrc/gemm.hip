// sp3_gemm: MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   C[M,N] = epilogue(alpha * A[M,K] . W[N,K]^T)
//
// Design (DESIGN.md §"GEMM"):
//  * The hot path is small-M (M = 196 tokens) weight streaming: 1.3 GB of bf16 weights per frame,
//    AI ~= 265 FLOP/B.  Every wave streams its operands STRAIGHT from global memory into MFMA
//    operand registers (no LDS staging, no barrier in the K loop) through a STAGES-deep register
//    ring, so a workgroup keeps (STAGES-1) k-blocks of loads in flight per wave.
//  * Both operands are K-contiguous (nn.Linear layout), so the contraction index may be permuted
//    freely as long as A and W use the same permutation: lane group g = lane>>4 owns CH contiguous
//    elements [kb*KB + g*CH, +CH) of every row -> each row is read as ONE contiguous 128-byte line
//    per k-block (bf16) instead of the 64-byte fragment-shaped pieces of the textbook mapping.
//  * Small-M tiles split K over the 4 waves of the workgroup (WK=4: no operand is loaded twice inside
//    a workgroup) and, when a GEMM has too few tiles to fill 256 CUs, additionally over `splitk`
//    workgroups (grid.z) whose fp32 partials are finished by sp3_reduce_ln (bias + residual + the
//    LayerNorm that follows on the residual stream).  Large-M tiles give each wave its own sub-tile.
//    Either way the accumulators go through LDS once, which decouples the MFMA C layout from the
//    store layout: the epilogue (bias, exact-erf GELU / ReLU, residuals, 2-D RoPE, per-head V^T
//    store, ConvTranspose pixel-shuffle) runs on coalesced 16-byte row segments.
//  * blockIdx -> tile mapping is XCD-aware (block b runs on XCD b%8): all tiles that share a weight
//    panel (small M) or an activation panel (large M) are placed on the same XCD, adjacent in
//    dispatch order, so the shared panel is fetched from HBM/MALL into ONE L2.
//  * fp32 mode uses v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain), bf16 mode v_mfma_f32_16x16x32_bf16
//    with fp32 accumulation; bf16-mode activations are either already bf16 (a_bf16) or fp32 converted
//    on load (v_cvt_pk_bf16_f32).
#include "common.h"
#include "gemm_sm.h"
#include <cstdlib>
#include <type_traits>

namespace {

// d2 / nb1: sp3_gemm2 -- a second, differently shaped group of problems in the same launch: blockIdx.y >= nb1 works on d2
struct GemmArgs {
  sp3_gemm_desc d;
  sp3_gemm_desc d2;
  int nb1;
  int xcd_slices;        // split-K slices mapped to XCDs (see the tile map)
};

// ------------------------------------------------------------------ per-dtype operand handling
// MM<TA, TW>: TA = dtype of A in memory, TW = dtype of W = MFMA dtype.
template <typename TA, typename TW> struct MM;

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.f;
  return z;
}

__device__ __forceinline__ bf16x8 relu8(bf16x8 v) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)v[i] > 0.f ? v[i] : (__bf16)0.f;
  return v;
}

// Loads are UNCONDITIONAL (addresses are clamped by the caller) and invalid halves are zeroed with a bitwise AND:
// a `cond ? load : 0` select makes hipcc branch around each load and wait vmcnt(0) per element, which serialises
// the whole operand stream (cdna_hip_programming.md §5 "three .s-level traps" (c)).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ bf16x8 mask8(bf16x8 v, unsigned m) {
  u32x4 b = __builtin_bit_cast(u32x4, v);
  b &= m;
  return __builtin_bit_cast(bf16x8, b);
}
__device__ __forceinline__ float4 mask4(float4 v, unsigned m) {
  v.x = __uint_as_float(__float_as_uint(v.x) & m);
  v.y = __uint_as_float(__float_as_uint(v.y) & m);
  v.z = __uint_as_float(__float_as_uint(v.z) & m);
  v.w = __uint_as_float(__float_as_uint(v.w) & m);
  return v;
}

struct WRegB { bf16x8 v[2]; };
// off1: element offset of the second half (CH/2 when it is inside K, 0 otherwise).  Masks (0 or ~0) and the
// optional ReLU are applied by fix*() right before the MFMAs consume the registers, so the only vmcnt wait of a
// stage sits at its first use.
__device__ __forceinline__ void loadW_b(WRegB& r, const __bf16* p, int off1) {
  r.v[0] = *reinterpret_cast<const bf16x8*>(p);
  r.v[1] = *reinterpret_cast<const bf16x8*>(p + off1);
}
__device__ __forceinline__ void fixW_b(WRegB& r, unsigned m0, unsigned m1) {
  r.v[0] = mask8(r.v[0], m0);
  r.v[1] = mask8(r.v[1], m1);
}

template <> struct MM<float, __bf16> {
  static constexpr int KB = 64;   // k elements per block
  static constexpr int CH = 16;   // k elements per lane per block
  struct AReg { float4 v[4]; };
  using WReg = WRegB;
  static __device__ __forceinline__ void loadA(AReg& r, const float* p, int off1) {
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4* q1 = reinterpret_cast<const float4*>(p + off1);
    r.v[0] = q[0]; r.v[1] = q[1]; r.v[2] = q1[0]; r.v[3] = q1[1];
  }
  static __device__ __forceinline__ void fixA(AReg& r, unsigned m0, unsigned m1, bool relu) {
    r.v[0] = mask4(r.v[0], m0); r.v[1] = mask4(r.v[1], m0); r.v[2] = mask4(r.v[2], m1); r.v[3] = mask4(r.v[3], m1);
    if (relu) { r.v[0] = relu4(r.v[0]); r.v[1] = relu4(r.v[1]); r.v[2] = relu4(r.v[2]); r.v[3] = relu4(r.v[3]); }
  }
  static __device__ __forceinline__ void loadW(WReg& r, const __bf16* p, int off1) { loadW_b(r, p, off1); }
  static __device__ __forceinline__ void fixW(WReg& r, unsigned m0, unsigned m1) { fixW_b(r, m0, m1); }
  template <int MF, int NF>
  static __device__ __forceinline__ void mma(f32x4 (&acc)[MF][NF], const AReg (&a)[MF], const WReg (&w)[NF]) {
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const bf16x8 a0 = cvt8(a[m].v[0], a[m].v[1]);
      const bf16x8 a1 = cvt8(a[m].v[2], a[m].v[3]);
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w[n].v[0], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w[n].v[1], acc[m][n], 0, 0, 0);
      }
    }
  }
  template <int MF, int NF>
  static __device__ __forceinline__ void mma_half(f32x4 (&)[MF][NF], const AReg (&)[MF], const WReg (&)[NF]) {}
};

template <> struct MM<__bf16, __bf16> {
  static constexpr int KB = 64;
  static constexpr int CH = 16;
  struct AReg { bf16x8 v[2]; };
  using WReg = WRegB;
  static __device__ __forceinline__ void loadA(AReg& r, const __bf16* p, int off1) {
    r.v[0] = *reinterpret_cast<const bf16x8*>(p);
    r.v[1] = *reinterpret_cast<const bf16x8*>(p + off1);
  }
  static __device__ __forceinline__ void fixA(AReg& r, unsigned m0, unsigned m1, bool relu) {
    r.v[0] = mask8(r.v[0], m0); r.v[1] = mask8(r.v[1], m1);
    if (relu) { r.v[0] = relu8(r.v[0]); r.v[1] = relu8(r.v[1]); }
  }
  static __device__ __forceinline__ void loadW(WReg& r, const __bf16* p, int off1) { loadW_b(r, p, off1); }
  static __device__ __forceinline__ void fixW(WReg& r, unsigned m0, unsigned m1) { fixW_b(r, m0, m1); }
  template <int MF, int NF>
  static __device__ __forceinline__ void mma(f32x4 (&acc)[MF][NF], const AReg (&a)[MF], const WReg (&w)[NF]) {
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m].v[0], w[n].v[0], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m].v[1], w[n].v[1], acc[m][n], 0, 0, 0);
      }
  }
  // one 16-byte half of the k-block (v[0] only): the LDS-staged loop with the contraction split over a wave pair
  template <int MF, int NF>
  static __device__ __forceinline__ void mma_half(f32x4 (&acc)[MF][NF], const AReg (&a)[MF], const WReg (&w)[NF]) {
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m].v[0], w[n].v[0], acc[m][n], 0, 0, 0);
  }
  // one 16-byte fragment half against one 16-byte fragment half (the pipelined LDS loop keeps bare halves in registers)
  static __device__ __forceinline__ void mma16(f32x4& acc, const bf16x8& a, const bf16x8& w) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w, acc, 0, 0, 0);
  }
};

template <> struct MM<float, float> {
  static constexpr int KB = 32;
  static constexpr int CH = 8;
  struct AReg { float4 v[2]; };
  struct WReg { float4 v[2]; };
  static __device__ __forceinline__ void loadA(AReg& r, const float* p, int off1) {
    r.v[0] = *reinterpret_cast<const float4*>(p);
    r.v[1] = *reinterpret_cast<const float4*>(p + off1);
  }
  static __device__ __forceinline__ void fixA(AReg& r, unsigned m0, unsigned m1, bool relu) {
    r.v[0] = mask4(r.v[0], m0); r.v[1] = mask4(r.v[1], m1);
    if (relu) { r.v[0] = relu4(r.v[0]); r.v[1] = relu4(r.v[1]); }
  }
  static __device__ __forceinline__ void loadW(WReg& r, const float* p, int off1) {
    r.v[0] = *reinterpret_cast<const float4*>(p);
    r.v[1] = *reinterpret_cast<const float4*>(p + off1);
  }
  static __device__ __forceinline__ void fixW(WReg& r, unsigned m0, unsigned m1) {
    r.v[0] = mask4(r.v[0], m0); r.v[1] = mask4(r.v[1], m1);
  }
  template <int MF, int NF>
  static __device__ __forceinline__ void mma(f32x4 (&acc)[MF][NF], const AReg (&a)[MF], const WReg (&w)[NF]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < MF; ++m) {
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].v[h].x, w[n].v[h].x, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].v[h].y, w[n].v[h].y, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].v[h].z, w[n].v[h].z, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].v[h].w, w[n].v[h].w, acc[m][n], 0, 0, 0);
        }
      }
    }
  }
  // f32x3: the 8 fp32 values a lane holds of a k-block are split into bf16 (hi, lo) pairs and the block's product is three
  // v_mfma_f32_16x16x32_bf16 (hi.hi + hi.lo + lo.hi; the dropped lo.lo term is 2^-16 relative) instead of eight
  // v_mfma_f32_16x16x4_f32: products carry 16 mantissa bits (TF32, which the reference enables on its own GPUs, carries 10),
  // accumulation stays fp32.  The fp32 MFMA rate, not the bytes, bounds the fp32 GEMMs of the per-frame step.
  static __device__ __forceinline__ void split8(const float4& x0, const float4& x1, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 h = (__bf16)x[e];
      hi[e] = h;
      lo[e] = (__bf16)(x[e] - (float)h);
    }
  }
  template <int MF, int NF>
  static __device__ __forceinline__ void mma_x3(f32x4 (&acc)[MF][NF], const AReg (&a)[MF], const WReg (&w)[NF]) {
    bf16x8 wh[NF], wl[NF];
#pragma unroll
    for (int n = 0; n < NF; ++n) split8(w[n].v[0], w[n].v[1], wh[n], wl[n]);
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      bf16x8 ah, al;
      split8(a[m].v[0], a[m].v[1], ah, al);
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh[n], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl[n], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[n], acc[m][n], 0, 0, 0);
      }
    }
  }
  // f32x3 = 3 ("f32x6"): a THREE-way bf16 split x = h + m + l (24 mantissa bits: the whole fp32 significand) and the six products
  // down to 2^-16 of the leading one (hh, hm, mh, mm, hl, lh; the dropped ml, lm, ll are <= 2^-24): fp32-grade products at 6 bf16 MFMAs
  // per k-block instead of 8 fp32 ones at 1/16 of the rate each.  The two-way split of mode 1 keeps 16 operand bits, which is not
  // enough once the residual stream carries massive-activation channels: the LayerNorm fold subtracts rstd * mean * s from a GEMM
  // whose terms are 40x larger than the result (tests/test_model_gpu.py stress fixture: 1.4e-3 with x3, fp32-level with x6).
  static __device__ __forceinline__ void split8x3(const float4& x0, const float4& x1, bf16x8& h, bf16x8& m, bf16x8& l) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 hh = (__bf16)x[e];
      const float r1 = x[e] - (float)hh;
      const __bf16 mm = (__bf16)r1;
      h[e] = hh; m[e] = mm; l[e] = (__bf16)(r1 - (float)mm);
    }
  }
  template <int MF, int NF>
  static __device__ __forceinline__ void mma_x6(f32x4 (&acc)[MF][NF], const AReg (&a)[MF], const WReg (&w)[NF]) {
    bf16x8 wh[NF], wm[NF], wl[NF];
#pragma unroll
    for (int n = 0; n < NF; ++n) split8x3(w[n].v[0], w[n].v[1], wh[n], wm[n], wl[n]);
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      bf16x8 ah, am, al;
      split8x3(a[m].v[0], a[m].v[1], ah, am, al);
#pragma unroll
      for (int n = 0; n < NF; ++n) {                    // smallest terms first
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh[n], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl[n], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wm[n], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wh[n], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wm[n], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[n], acc[m][n], 0, 0, 0);
      }
    }
  }
  // f32x3 = 4 ("f16x3"): a TWO-way split into fp16 halves, x = h + l * 2^-11 with h = fp16(x), l = fp16((x - h) * 2^11): 22 operand
  // bits (bf16 halves: 16) for the same three MFMAs -- hh into the main accumulator, hl + lh into a second one that is scaled by
  // 2^-11 when the accumulators leave the K loop.  Every product is exact in fp32 (11 x 11 bits), accumulation is fp32.  Operands
  // must stay below fp16's 65504 (activations and weights of this model do; the fp32 / f32x6 modes have no such limit).
  typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
  static __device__ __forceinline__ void split8h(const float4& x0, const float4& x1, f16x8& h, f16x8& l) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 hh = (_Float16)x[e];
      h[e] = hh;
      l[e] = (_Float16)((x[e] - (float)hh) * 2048.0f);
    }
  }
  // WPRE (sp3_gemm_desc.w_packed == 2): the weight already IS its two fp16 planes -- the first 16-byte half of a lane's fragment holds
  // h of its 8 k, the second half l (split once at pack time, ops.PackedWeight(halves=True): the same two conversions, so the same
  // bits) -- and the split of W leaves the K loop (a third of the loop's VALU work on the 32x32 wave tile)
  template <int MF, int NF, bool WPRE = false>
  static __device__ __forceinline__ void mma_h3(f32x4 (&acc)[MF][NF], f32x4 (&accx)[MF][NF], const AReg (&a)[MF], const WReg (&w)[NF]) {
    f16x8 wh[NF], wl[NF];
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      if constexpr (WPRE) {
        wh[n] = __builtin_bit_cast(f16x8, w[n].v[0]);
        wl[n] = __builtin_bit_cast(f16x8, w[n].v[1]);
      } else {
        split8h(w[n].v[0], w[n].v[1], wh[n], wl[n]);
      }
    }
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      f16x8 ah, al;
      split8h(a[m].v[0], a[m].v[1], ah, al);
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        accx[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[n], accx[m][n], 0, 0, 0);
        accx[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[n], accx[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[n], acc[m][n], 0, 0, 0);
      }
    }
  }
  // f32x3 = 2 ("bf16 products"): fp32 operands in memory, rounded to bf16 on their way into ONE v_mfma_f32_16x16x32_bf16 per
  // k-block, fp32 accumulate -- the arithmetic of a bf16-autocast matmul on fp32 master tensors (training step, bf16 mode)
  template <int MF, int NF>
  static __device__ __forceinline__ void mma_x1(f32x4 (&acc)[MF][NF], const AReg (&a)[MF], const WReg (&w)[NF]) {
    bf16x8 wh[NF];
#pragma unroll
    for (int n = 0; n < NF; ++n) wh[n] = cvt8(w[n].v[0], w[n].v[1]);
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const bf16x8 ah = cvt8(a[m].v[0], a[m].v[1]);
#pragma unroll
      for (int n = 0; n < NF; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[n], acc[m][n], 0, 0, 0);
    }
  }
  template <int MF, int NF>
  static __device__ __forceinline__ void mma_half(f32x4 (&acc)[MF][NF], const AReg (&a)[MF], const WReg (&w)[NF]) {
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].v[0].x, w[n].v[0].x, acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].v[0].y, w[n].v[0].y, acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].v[0].z, w[n].v[0].z, acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].v[0].w, w[n].v[0].w, acc[m][n], 0, 0, 0);
      }
  }
  static __device__ __forceinline__ void mma16(f32x4& acc, const float4& a, const float4& w) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
  }
};

// ------------------------------------------------------------------ A-row addressing
// PLAIN: row m at A + m*lda (optionally a second source A2 for k >= K1).
// CONV3X3: row m is output pixel (b, oy, ox) of an NHWC map.
template <typename TA, int LOADER, int MF> struct ARows;

template <typename TA, int MF> struct ARows<TA, SP3_LOAD_PLAIN, MF> {
  const TA* base[MF];
  const TA* base2[MF];
  // a_packed: A is in MFMA-fragment order [ceil(M/16)][ceil(K/KB)][64 lanes][CH] (written that way by the producing
  // kernel), so a wave's operand load is one contiguous run; otherwise row-major [M, lda].
  __device__ __forceinline__ void init(const sp3_gemm_desc& d, const TA* A, int row0, int lane, int KB, int CH) {
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      if (d.a_packed) {
        int mb = (row0 + m * 16) >> 4;
        const int mb_max = (d.M + 15) / 16 - 1;
        mb = mb < mb_max ? mb : mb_max;
        const int nkb = (d.K + KB - 1) / KB;
        // ptr() adds k = kb*KB + g*CH; fold "- g*CH" here and scale kb*KB -> kb*64*CH through pk_mul
        base[m] = A + ((int64_t)mb * nkb * 128 + lane) * (CH / 2);      // first 1 KB half of the block; + 64*CH/2 = second
        base2[m] = base[m];
      } else {
        int r = row0 + m * 16 + (lane & 15);
        r = r < d.M ? r : d.M - 1;
        base[m] = A + (int64_t)r * d.lda;
        base2[m] = d.A2 ? reinterpret_cast<const TA*>(d.A2) + (int64_t)r * d.lda2 - d.K1 : base[m];
      }
    }
  }
  __device__ __forceinline__ const TA* ptr(const sp3_gemm_desc& d, int m, int k, int kb, int CH, bool& inb) const {
    inb = true;
    if (d.a_packed) return base[m] + (int64_t)kb * 64 * CH;
    return (k < d.K1 ? base[m] : base2[m]) + k;
  }
};

// SOFTMAX: row-major fp32 scores [M, lda]; the values become probabilities at consume time (smx4 below).
template <typename TA, int MF> struct ARows<TA, SP3_LOAD_SOFTMAX, MF> {
  const TA* base[MF];
  __device__ __forceinline__ void init(const sp3_gemm_desc& d, const TA* A, int row0, int lane, int, int) {
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      int r = row0 + m * 16 + (lane & 15);
      r = r < d.M ? r : d.M - 1;
      base[m] = A + (int64_t)r * d.lda;
    }
  }
  __device__ __forceinline__ const TA* ptr(const sp3_gemm_desc&, int m, int k, int, int, bool& inb) const {
    inb = true;
    return base[m] + k;
  }
};

// scores -> thresholded probabilities of 4 consecutive keys: p = exp(s - m) / Z, dropped below thr and past the bank's
// end (`left` keys remain from this group on); z collects the kept mass of the row (the renormalisation of model.py:170-172)
// (c = -m log2(e) - log2(Z): one fma and one v_exp_f32 per score)
__device__ __forceinline__ float4 smx4(float4 s, float c, float thr, int left, float& z) {
  constexpr float L2E = 1.44269504088896341f;
  float p[4] = {__builtin_amdgcn_exp2f(fmaf(s.x, L2E, c)), __builtin_amdgcn_exp2f(fmaf(s.y, L2E, c)),
                __builtin_amdgcn_exp2f(fmaf(s.z, L2E, c)), __builtin_amdgcn_exp2f(fmaf(s.w, L2E, c))};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    p[e] = (p[e] < thr || e >= left) ? 0.f : p[e];
    z += p[e];
  }
  return make_float4(p[0], p[1], p[2], p[3]);
}

template <typename TA, int MF> struct ARows<TA, SP3_LOAD_CONV3X3, MF> {
  const TA* img[MF];
  int iy0[MF], ix0[MF];
  const TA* safe;
  __device__ __forceinline__ void init(const sp3_gemm_desc& d, const TA* A, int row0, int lane, int, int) {
    safe = A;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      int r = row0 + m * 16 + (lane & 15);
      r = r < d.M ? r : d.M - 1;
      const int per = d.conv_OH * d.conv_OW;
      const int b = r / per;
      const int rem = r - b * per;
      const int oy = rem / d.conv_OW;
      const int ox = rem - oy * d.conv_OW;
      img[m] = A + (int64_t)b * d.conv_H * d.conv_W * d.conv_C;
      iy0[m] = oy * d.conv_stride - 1;
      ix0[m] = ox * d.conv_stride - 1;
    }
  }
  __device__ __forceinline__ const TA* ptr(const sp3_gemm_desc& d, int m, int k, int, int, bool& inb) const {
    const int tap = k / d.conv_C;
    const int ci = k - tap * d.conv_C;
    const int dy = tap / 3;
    const int iy = iy0[m] + dy;
    const int ix = ix0[m] + (tap - dy * 3);
    inb = (iy >= 0) && (iy < d.conv_H) && (ix >= 0) && (ix < d.conv_W);
    return inb ? img[m] + ((int64_t)iy * d.conv_W + ix) * d.conv_C + ci : safe;
  }
};

// ------------------------------------------------------------------ the kernel
// 16 bytes per lane, global -> LDS without passing through registers (global_load_lds_dwordx4): lane l's bytes land at
// the wave-uniform `lds` + 16 l.
__device__ __forceinline__ void glds16(const char* gsrc, char* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
#else
  (void)gsrc; (void)lds;
#endif
}

// LDSK = true selects the LDS-staged K loop (fragment-order A and W in the MFMA dtype, STAGES = 2..4 ring slots, one
// barrier per k-block): both operand tiles arrive once per workgroup by global_load_lds and every wave reads its fragments
// from LDS.  Two uses: (a) tiles too wide for the per-wave register ring (128 x 128: the many-row GEMMs of the
// whole-sequence encoder; 2 slots measured best there, three workgroups per CU beat a deeper ring); (b) the 196-row
// weight-streaming GEMMs of the per-frame step: 112- or 208-row tiles read each weight panel once or twice instead of
// seven times (32-row tiles), which is what bounds those launches (L2 -> CU bytes), with a 3-4 slot ring because only
// one workgroup fits a CU.  The epilogues are shared.
// The hot 32x32 bf16 tile sits at the edge of 3 waves per SIMD (512 / 3 = 170 registers); small edits used to tip it
// over to 2, which costs 10-25 % on the wide-N launches (tools/bench_block.py), so its register budget is pinned.
template <typename TA, int LOADER, int MF, int NF, int WK>
constexpr int gemm_min_waves() { return (sizeof(TA) == 2 && LOADER == SP3_LOAD_PLAIN && MF == 2 && NF == 2 && WK == 4) ? 3 : 1; }

// LOOP: 0 = register ring, 1 = both operands through LDS (above), >= 2 = weight streaming with roles (below) and
// LOOP loader waves.
// LOOP >= 2 (the per-frame step's 196-row GEMMs with HBM-cold weights).  Measured on MI355X (tools/ubench/stream.hip, tools/
// bench_gemm2.py): a CU pulls 40-50 B/clk of L2-hot data but every cold weight byte is ~3000 clk away, so a loop that
// keeps its weights and activations in ONE in-order vmcnt queue runs at latency / ring depth per k-block.  Here the
// queues are split by wave role: WM*WN*WK consumer waves keep their own weight fragments in a 12-k-block REGISTER ring
// (plain loads, refilled right after use: 12 k-blocks of prefetch distance, compiler-counted waits), and NLOAD loader
// waves do nothing but DMA the shared activation tile (BM x 64 per k-block) into a STAGES-slot LDS ring.
// compile-time loop with early exit: f(integral_constant<int, I>) -> keep going?
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    if (f(std::integral_constant<int, I>{})) static_for<I + 1, N>(f);
  }
}

constexpr int kWRing = 12;
constexpr int kMaxKb = 64;                 // k-blocks per K slice the role loop is unrolled for (K / splitk <= 4096 in bf16)

// XM (fp32 operands, register-ring tiles): how a k-block's products are formed -- 0: fp32 MFMAs; 1 / 3: three / six bf16 MFMAs of a
// two- / three-way split (f32x3 / f32x6); 2: one bf16 MFMA of the rounded operands.  A compile-time choice: as a run-time branch the
// six-product code cost every mode its second wave per SIMD (187 -> 209 registers).  4: three fp16 MFMAs of the two-way split x = h + l 2^-11
// ("f16x3"); 5: the same with W already split into its (h, l) planes at pack time (w_packed == 2).
template <typename TA, typename TW, int LOADER, int MF, int NF, int WM, int WN, int WK, int STAGES, int LOOP = 0, int XM = 0>
__global__ __launch_bounds__(64 * (WM * WN * WK + (LOOP >= 2 ? LOOP : 0)), (gemm_min_waves<TA, LOADER, MF, NF, WK>()))
void gemm_kernel(const GemmArgs args) {
  // local copy: grouped launches shift the per-problem pointers below (kernarg segment, wave-uniform select)
  const bool second = args.nb1 > 0 && (int)blockIdx.y >= args.nb1;
  sp3_gemm_desc d = second ? args.d2 : args.d;
  using M_ = MM<TA, TW>;
  constexpr bool LDSK = LOOP != 0;
  constexpr int NCW = WM * WN * WK;          // consumer (MFMA) waves
  constexpr int BM = MF * 16 * WM, BN = NF * 16 * WN, NT = 64 * (NCW + (LOOP >= 2 ? LOOP : 0));
  constexpr int KB = M_::KB, CH = M_::CH;
  constexpr int LDS_LD = BN + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [WK][BM][LDS_LD]

  const int mt = (d.M + BM - 1) / BM, nt = (d.N + BN - 1) / BN;
  // XCD-aware tile mapping (block b -> XCD b & 7); grid is padded, out-of-range tiles exit.
  // Split-K over a multiple of 8 slices (the long-bank P.V GEMM of the memory read: 16 slices x 32 tiles): XCD x takes the slices
  // {x, x + 8, ..} with ALL their tiles, so the two operand slabs of a slice are fetched by ONE L2 instead of by all eight (PMC,
  // profiles/r05_memread_long_bank_launch_breakdown.txt: 904 MB fetched for 206 MB of operands with the slices spread over the XCDs).
  // Dispatch order L = x + gridDim.x * z (gridDim.x % 8 == 0, gridDim.y == 1) -> XCD L % 8.
  int tile_m, tile_n;
  int bx = blockIdx.x, kz = blockIdx.z;
  if (args.xcd_slices) {
    const unsigned L = blockIdx.x + gridDim.x * blockIdx.z, xc = L & 7, jj = L >> 3, q = jj / gridDim.x;
    kz = (int)(xc + 8 * q);
    bx = (int)(jj - q * gridDim.x);
  }
  {
    const int b = bx, xcd = b & 7, j = b >> 3;
    if (mt >= nt) {          // large M: the tiles that share an A panel sit on one XCD
      tile_n = j % nt;
      tile_m = (j / nt) * 8 + xcd;
    } else {                 // small M: the tiles that share a W panel sit on one XCD
      tile_m = j % mt;
      tile_n = (j / mt) * 8 + xcd;
    }
  }
  if (tile_m >= mt || tile_n >= nt) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
  const int g = lane >> 4;
  const int bz = second ? (int)blockIdx.y - args.nb1 : (int)blockIdx.y;
  if (d.batch > 1) {                         // grouped launch: problem bz (wave-uniform pointer arithmetic)
    auto shift = [&](auto*& p, int64_t bytes) {
      using P = std::remove_reference_t<decltype(p)>;
      if (p) p = reinterpret_cast<P>(reinterpret_cast<uintptr_t>(p) + (uintptr_t)(bz * bytes));
    };
    shift(d.A2, d.sb_A2); shift(d.bias, d.sb_bias); shift(d.ln_stats, d.sb_ln_stats); shift(d.ln_s, d.sb_ln_s);
    shift(d.stats_out, d.sb_stats_out); shift(d.c2, d.sb_c2); shift(d.vt, d.sb_vt);
    const int64_t csz = d.epi == SP3_EPI_ROPE_VT ? (int64_t)sizeof(TW) : (d.out_bf16 ? 2 : 4);
    shift(d.C, d.strideC * csz);
    d.strideC = 0;
  }
  const TA* A = reinterpret_cast<const TA*>(d.A) + (int64_t)bz * d.strideA;
  const TW* W = reinterpret_cast<const TW*>(d.W) + (int64_t)bz * d.strideW;

  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int64_t* tr0 = ((LOOP == 0 || LOOP == -1) && args.d.trace && blockIdx.x < 8 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) ? args.d.trace + blockIdx.x * 64 : nullptr;
  if (tr0) { tr0[0] = clock64(); tr0[3] = wall_clock64(); }
  ARows<TA, LOADER, MF> arows;
  arows.init(d, A, m0 + wm * MF * 16, lane, M_::KB, M_::CH);
  // W operand addressing.  Row-major [N, ldw]: lane (g, r) reads row c at k = kb*KB + g*CH (16 rows per load
  // instruction -> 4 different cache lines per lane quad).  Packed (w_packed, weights re-ordered once at load time into
  // MFMA-fragment order [N/16][K/KB][lane][CH]): a wave reads 64*CH contiguous elements per fragment -> fully coalesced.
  const TW* wbase[NF];
  int64_t wstep;
  if (d.w_packed) {
    const int nkb_pad = (int)((d.ldw + KB - 1) / KB);      // k-blocks ALLOCATED per row panel (ldw >= K: a bank that grows along K)
    wstep = 64 * CH;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      int nb = (n0 + wn * NF * 16 + n * 16) >> 4;
      const int nb_max = (d.N + 15) / 16 - 1;
      nb = nb < nb_max ? nb : nb_max;
      wbase[n] = W + ((int64_t)nb * nkb_pad * 128 + lane) * (CH / 2);
    }
  } else {
    wstep = KB;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      int c = n0 + wn * NF * 16 + n * 16 + (lane & 15);
      c = c < d.N ? c : d.N - 1;
      wbase[n] = W + (int64_t)c * d.ldw + g * CH;
    }
  }

  f32x4 acc[MF][NF];
  constexpr bool XACC = sizeof(TA) == 4 && sizeof(TW) == 4 && (XM == 4 || XM == 5);      // f16x3: the cross terms' accumulator
  f32x4 accx[XACC ? MF : 1][XACC ? NF : 1];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (XACC) {
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) accx[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // Epilogue operands that depend on the column only (bias, LayerNorm column sums): NT is a multiple of BN/4, so a thread
  // keeps one 4-column group through the whole epilogue -> request them NOW.  Every launch has its own vectors, cold in
  // HBM: loaded in the epilogue they cost each workgroup a ~3000-clk round trip after its last MFMA.
  static_assert(NT % (BN / 4) == 0, "epilogue column group must be fixed per thread");
  const int ec4 = (tid % (BN / 4)) * 4;
  const bool epre = (n0 + ec4 + 4) <= d.N && (d.N & 3) == 0 && d.epi != SP3_EPI_PIXSHUF;
  float4 pre_b4 = make_float4(0.f, 0.f, 0.f, 0.f), pre_s4 = pre_b4;
  if (epre) {
    if (d.bias) pre_b4 = *reinterpret_cast<const float4*>(d.bias + n0 + ec4);
    if (d.ln_stats) pre_s4 = *reinterpret_cast<const float4*>(d.ln_s + n0 + ec4);
  }

  // Folded LayerNorm: the producer's per-32-column (sum, sum of squares) partials of this tile's rows are requested
  // BEFORE the K loop (TPR threads per row, 4 partials each in flight) and reduced after it: their latency hides
  // behind the GEMM instead of sitting between two barriers.
  // threads per tile row: the largest power of two (lane-aligned groups) with TPR * BM <= NT; spare threads idle here
  constexpr int TPR = NT >= 8 * BM ? 8 : NT >= 4 * BM ? 4 : NT >= 2 * BM ? 2 : 1;
  // partials requested up front per thread: 4 where registers are short (the register-ring tiles, the 128-row LDS tiles
  // that live on three workgroups per CU); the weight-streaming LDS tiles take all of a 1024-wide row (32 groups)
  constexpr int NPRE = (LDSK && LOOP != -1 && STAGES >= 3) ? 32 / TPR : 4;
  const int srow = tid / TPR, sj = tid % TPR;
  float2 sp[NPRE];
#pragma unroll
  for (int q = 0; q < NPRE; ++q) sp[q] = make_float2(0.f, 0.f);
  const float2* sps = nullptr;
  // the conv loader never folds a LayerNorm (keeps its registers); in the role loop the loader waves own the statistics
  constexpr bool LNF = LOADER == SP3_LOAD_PLAIN && LOOP < 2;
  if (LNF && d.ln_stats && srow < BM) {
    const int sgm = (m0 + srow) < d.M ? (m0 + srow) : d.M - 1;
    sps = reinterpret_cast<const float2*>(d.ln_stats) + (int64_t)sgm * d.ln_nt;
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
      const int t = sj + q * TPR;
      sp[q] = sps[t < d.ln_nt ? t : d.ln_nt - 1];     // unconditional load, clamped; masked when summed
    }
  }

  // k-blocks of this workgroup's K slice (split-K over grid.z), interleaved over the WK waves
  const int nkb_all = (d.K + KB - 1) / KB;
  const int per = (nkb_all + d.splitk - 1) / d.splitk;
  const int kb_lo = kz * per;
  const int kb_hi = (kb_lo + per) < nkb_all ? (kb_lo + per) : nkb_all;
  const bool relu = d.relu_in != 0;

  typename M_::AReg a[STAGES][MF];
  typename M_::WReg w[STAGES][NF];
  // masks of each stage (0 / ~0): K tail per half, and per A row the conv zero-padding; applied at consume time
  unsigned am0[STAGES][MF], am1[STAGES][MF], wm0[STAGES], wm1[STAGES];
  constexpr bool CONV = LOADER == SP3_LOAD_CONV3X3;
  // softmax loader: per lane the (max, 1/Z) of its MF rows, the kept probability mass, and per stage the keys left from
  // the lane's first element to the end of the bank
  constexpr bool SMX = LOADER == SP3_LOAD_SOFTMAX;
  float sm_c[MF], sm_z[MF];
  int sm_left[STAGES];
#pragma unroll
  for (int m = 0; m < MF; ++m) { sm_c[m] = 0.f; sm_z[m] = 0.f; }

  // FULL = the k-block lies entirely inside K.  Loads are never predicated; addresses are clamped instead.
  auto load = [&](auto full_tag, int st, int kb) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int k = kb * KB + g * CH;
    const bool v0 = FULL || k < d.K, v1 = FULL || (k + CH / 2) < d.K;
    const int kc = v0 ? k : 0;
    const int off1 = v1 ? CH / 2 : 0;
    wm0[st] = v0 ? 0xffffffffu : 0u;
    wm1[st] = v1 ? 0xffffffffu : 0u;
    if constexpr (SMX) sm_left[st] = d.K - k;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      bool inb;
      const TA* p = arows.ptr(d, m, kc, kb, CH, inb);
      const unsigned im = inb ? 0xffffffffu : 0u;
      am0[st][m] = wm0[st] & im;
      am1[st][m] = wm1[st] & im;
      M_::loadA(a[st][m], p, d.a_packed ? 64 * CH / 2 : off1);
    }
    // row-major W: same clamp as A (kc - g*CH is the k-block base); packed W is zero-padded to whole k-blocks
    const int64_t woff = d.w_packed ? (int64_t)kb * wstep : (int64_t)(kc - g * CH);
    const int woff1 = d.w_packed ? 64 * CH / 2 : off1;
#pragma unroll
    for (int n = 0; n < NF; ++n) M_::loadW(w[st][n], wbase[n] + woff, woff1);
  };
  auto consume = [&](auto full_tag, int st) {
    constexpr bool FULL = decltype(full_tag)::value;
    if constexpr (SMX) {
      if constexpr (sizeof(TA) == 4) {
        constexpr int NV = sizeof(typename M_::AReg) / 16;       // float4s per fragment: 4 (bf16 MFMA, 16 keys) or 2 (fp32, 8 keys)
        const int left = FULL ? 64 : sm_left[st];
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int q = 0; q < NV; ++q) a[st][m].v[q] = smx4(a[st][m].v[q], sm_c[m], d.sm_thresh, left - 4 * q, sm_z[m]);
      }
    } else if (!FULL || CONV) {
#pragma unroll
      for (int m = 0; m < MF; ++m) M_::fixA(a[st][m], am0[st][m], am1[st][m], relu);
    }
    if (!FULL) {
#pragma unroll
      for (int n = 0; n < NF; ++n) M_::fixW(w[st][n], wm0[st], wm1[st]);
    }
    if constexpr (sizeof(TA) == 4 && sizeof(TW) == 4 && XM != 0) {
      if constexpr (XM == 1) M_::template mma_x3<MF, NF>(acc, a[st], w[st]);
      else if constexpr (XM == 2) M_::template mma_x1<MF, NF>(acc, a[st], w[st]);
      else if constexpr (XM == 4) M_::template mma_h3<MF, NF>(acc, accx, a[st], w[st]);
      else if constexpr (XM == 5) M_::template mma_h3<MF, NF, true>(acc, accx, a[st], w[st]);
      else M_::template mma_x6<MF, NF>(acc, a[st], w[st]);
    } else {
      M_::template mma<MF, NF>(acc, a[st], w[st]);
    }
  };
  using FullT = std::integral_constant<bool, true>;
  using TailT = std::integral_constant<bool, false>;

  if constexpr (LOOP >= 2) {
    constexpr int kLoaderWaves = LOOP;
    static_assert(sizeof(TA) == sizeof(TW) && WK <= 2 && NF == 1 && WM == 1, "role loop: fragment-order operands, one column block per wave");
    constexpr int MBLK = BM / 16, STAGE_BYTES = MBLK * 2048, NINSTR = 2 * MBLK;
    constexpr int PER = (NINSTR + kLoaderWaves - 1) / kLoaderWaves, NST = STAGES;
    static_assert(NST >= 2 && NST <= 4 && PER * (NST - 1) < 64, "activation ring: 2..4 slots");
    char* lds_b = reinterpret_cast<char*>(smem);
    const int nkb_a = (d.K + KB - 1) / KB;
    const int nk = kb_hi - kb_lo;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int64_t* tr = (d.trace && blockIdx.x < 8 && blockIdx.y == 0 && blockIdx.z == 0) ? d.trace + blockIdx.x * 64 : nullptr;
    if (tr && tid == 0) { tr[0] = clock64(); tr[3] = wall_clock64(); }
    if (wave_u >= NCW) {
      // ---------------- loader waves: the activation tile, k-block by k-block, into the LDS ring
      const int lw = wave_u - NCW;
      const int rb_max = (d.M + 15) / 16 - 1;
      const char* src[PER];
      int dst[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        int j = lw + i * kLoaderWaves;
        j = j < NINSTR ? j : NINSTR - 1;                  // surplus slot: repeats the last piece
        int rb = (m0 >> 4) + (j >> 1);
        rb = rb < rb_max ? rb : rb_max;                   // rows past M: re-read the last block (masked at the store)
        src[i] = reinterpret_cast<const char*>(A + (int64_t)rb * nkb_a * (2048 / sizeof(TA))) + (j & 1) * 1024 + lane * 16;
        dst[i] = j * 1024;
      }
      auto issue = [&](int slot, int kb) {
#pragma unroll
        for (int i = 0; i < PER; ++i) glds16(src[i] + (int64_t)kb * 2048, lds_b + slot * STAGE_BYTES + dst[i]);
      };
      if (tr && wave_u == NCW && lane == 0) tr[5] = clock64();
#pragma unroll
      for (int s_ = 0; s_ < NST - 1; ++s_)
        if (s_ < nk) issue(s_, kb_lo + s_);
      asm volatile("s_barrier" ::: "memory");             // the first activation stages are queued ahead of the weight flood
      // folded LayerNorm: the loader waves own the row statistics (their registers are free; the consumers' hold the
      // weight ring): one thread per row requests all per-32-column partials of the producer now, sums them after the loop
      constexpr int RPT = (BM + 64 * kLoaderWaves - 1) / (64 * kLoaderWaves);
      float2 lnp[RPT][32];
      if (d.ln_stats) {
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
          const int row = tid - 64 * NCW + rr * 64 * kLoaderWaves;
          const int gm = (m0 + row) < d.M ? (m0 + row) : d.M - 1;          // clamped, unconditional loads
          const float2* ps = reinterpret_cast<const float2*>(d.ln_stats) + (int64_t)gm * d.ln_nt;
#pragma unroll
          for (int q = 0; q < 32; ++q) lnp[rr][q] = ps[q < d.ln_nt ? q : d.ln_nt - 1];
        }
      }
      int fill = NST - 1;
      for (int i = 0; i < nk; ++i) {
        const int ahead = nk - 1 - i;
        const int pending = ahead < NST - 2 ? ahead : NST - 2;
        if (pending >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
        else if (pending == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tr && wave_u == NCW && lane == 0 && i == 0) tr[6] = clock64();
        asm volatile("s_barrier" ::: "memory");           // stage i visible to the consumers; slot of stage i-1 is free
        if (tr && wave_u == NCW && lane == 0 && (i & 3) == 0 && i < 96) tr[8 + (i >> 2)] = clock64();
        if (i + NST - 1 < nk) issue(fill, kb_lo + i + NST - 1);
        fill = fill + 1 == NST ? 0 : fill + 1;
      }
      // folded LayerNorm: finish mean / rstd of the tile's rows (partials requested before the DMA loop).  rowstat sits
      // behind the epilogue slabs, beyond the ring, so writing it while the consumers still multiply is safe.
      if (d.ln_stats) {
        float* rs = smem + (size_t)WK * BM * LDS_LD;
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
          const int row = tid - 64 * NCW + rr * 64 * kLoaderWaves;
          if (row < BM) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int q = 0; q < 32; ++q)
              if (q < d.ln_nt) { s1 += lnp[rr][q].x; s2 += lnp[rr][q].y; }
            const float mean = s1 / (float)d.ln_C;
            const float var = fmaxf(s2 / (float)d.ln_C - mean * mean, 0.f);
            rs[2 * row] = mean;
            rs[2 * row + 1] = 1.0f / sqrtf(var + d.ln_eps);
          }
        }
      }
    } else {
      // ---------------- consumer waves: own weight fragments in registers, activations from the LDS ring
      using W16 = std::remove_reference_t<decltype(((typename M_::WReg*)nullptr)->v[0])>;
      using V16 = std::remove_reference_t<decltype(((typename M_::AReg*)nullptr)->v[0])>;
      static_assert(sizeof(W16) == 16 && sizeof(V16) == 16, "fragment halves are 16 bytes");
      const int nkb_w = (int)((d.ldw + KB - 1) / KB);
      int nb = (n0 >> 4) + wn;
      const int nb_max = (d.N + 15) / 16 - 1;
      nb = nb < nb_max ? nb : nb_max;
      // WK = 2: the pair splits every k-block by its two 1 KB halves
      const int hoff = WK == 2 ? wk * 1024 : 0;
      const char* wsrc = reinterpret_cast<const char*>(W + (int64_t)nb * nkb_w * (2048 / sizeof(TW))) + lane * 16;
      W16 wr[kWRing][WK == 2 ? 1 : 2];
      auto loadw = [&](int slot, int kb) {                // kb clamped by the caller
        const char* q = wsrc + (int64_t)kb * 2048;
        wr[slot][0] = *reinterpret_cast<const W16*>(q + hoff);
        if constexpr (WK == 1) wr[slot][1] = *reinterpret_cast<const W16*>(q + 1024);
      };
      const int kb_last = kb_hi - 1;
      // the first kWHead k-blocks of weights go out at once; the rest of the ring waits until the loaders have queued their
      // first activation stages (the memory pipe of a CU is first come, first served: behind 128 KB of cold weight
      // requests the first activation stage used to land ~6000 clk late)
      constexpr int kWHead = 4;
#pragma unroll
      for (int s_ = 0; s_ < kWHead; ++s_) { const int kb = kb_lo + s_; loadw(s_, kb < kb_last ? kb : kb_last); }
      asm volatile("s_barrier" ::: "memory");
#pragma unroll
      for (int s_ = kWHead; s_ < kWRing; ++s_) { const int kb = kb_lo + s_; loadw(s_, kb < kb_last ? kb : kb_last); }
      if (tr && tid == 0) tr[7] = clock64();
      // Fully unrolled over the k-blocks of the slice (<= kMaxKb; uniform early exit): straight-line code lets hipcc count
      // the register loads exactly -- inside a real loop its waitcnt pass drains the whole ring at every back edge.
      // Software-pipelined over the LDS ring: step i first passes the barrier of stage i+1 and requests ITS activation
      // fragments, then multiplies stage i from the registers filled one step earlier (ds_read latency hides under MFMAs).
      // The lgkmcnt(0) in front of each barrier retires the reads of the stage whose slot the loaders refill next.
      typename M_::AReg af[2][MF];
      auto read_a = [&](int buf, int stage) {
        const char* st = lds_b + (stage % NST) * STAGE_BYTES + lane * 16;
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          af[buf][m].v[0] = *reinterpret_cast<const V16*>(st + m * 2048 + hoff);
          if constexpr (WK == 1) af[buf][m].v[1] = *reinterpret_cast<const V16*>(st + m * 2048 + 1024);
        }
      };
      asm volatile("s_barrier" ::: "memory");             // stage 0 landed
      read_a(0, 0);
      static_for<0, kMaxKb>([&](auto ic) -> bool {
        constexpr int i = decltype(ic)::value;
        if (i >= nk) return false;                        // wave-uniform
        if (i + 1 < nk) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          read_a((i + 1) & 1, i + 1);
        }
        typename M_::WReg wf[1];
        wf[0].v[0] = wr[i % kWRing][0];
        if constexpr (WK == 1) { wf[0].v[1] = wr[i % kWRing][1]; M_::template mma<MF, 1>(acc, af[i & 1], wf); }
        else M_::template mma_half<MF, 1>(acc, af[i & 1], wf);
        if (i + kWRing < kMaxKb) {                        // refill this register slot kWRing k-blocks ahead: unconditional
          const int kn = kb_lo + i + kWRing;              // (clamped) -- a predicated load would drain the queue
          loadw(i % kWRing, kn < kb_last ? kn : kb_last);
        }
        if constexpr ((i & 3) == 0) {
          if (tr && tid == 0) { asm volatile("" ::"v"(acc[0][0][0])); tr[32 + (i >> 2)] = clock64(); }
        }
        return true;
      });
    }
    if (tr && tid == 0) tr[1] = clock64();
    __syncthreads();                                     // the ring is dead: the epilogue slab re-uses its bytes
  } else
  if constexpr (LOOP == -1) {
    // ---- pipelined LDS-staged K loop (the many-row GEMMs: whole-sequence encoder, 512x512 steps, long-bank memory reads,
    // training).  Same ring of fragment-order stages as LOOP 1 (one stage = BM/16 A blocks + BN/16 W blocks of 2 KB, moved
    // by global_load_lds pieces dealt round-robin over the waves), but
    //  * a slot is refilled right behind the barrier that publishes the NEXT stage (by then every wave holds the old
    //    stage in registers), so NST-1 stages are in flight while one is multiplied (HBM-cold weight panels);
    //  * a wave keeps TWO fragment sets in registers: while the MFMAs of one 16-byte half (WK = 1) / of one stage's own
    //    half (WK = 2: the wave pair splits every k-block) run, the ds_reads of the next one are already in flight --
    //    LOOP 1 reads a whole stage and then waits for it before its first MFMA;
    //  * 8-wave workgroups (two waves per SIMD): 256x128 (4x2 waves of 64x64) and 128x128 (2x2 waves x 2 K halves).
    static_assert(sizeof(TA) == sizeof(TW), "LDS-staged loop: A and W in the MFMA dtype (fragment order)");
    static_assert(WK <= 2, "pipelined LDS loop: K over at most 2 waves");
    constexpr int NW = NT / 64;
    constexpr int NBLK = BM / 16 + BN / 16, STAGE_BYTES = NBLK * 2048, NINSTR = 2 * NBLK, PER = (NINSTR + NW - 1) / NW;
    constexpr int NST = STAGES;
    static_assert(NST >= 2 && NST <= 4 && PER * (NST - 1) < 64, "stage ring: 2..4 stages, vmcnt is a 6-bit count");
    char* lds_b = reinterpret_cast<char*>(smem);
    const int nkb_pad = (d.K + KB - 1) / KB;
    const int rb_max = (d.M + 15) / 16 - 1, nb_max = (d.N + 15) / 16 - 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* src[PER];
    int dst[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int j = wave_u + i * NW;                            // piece 0 .. NINSTR-1: A row blocks first, then W column blocks
      j = j < NINSTR ? j : NINSTR - 1;                    // surplus slot: repeats the last piece (constant vmcnt per stage)
      const int blk = j >> 1;
      const char* base;
      if (blk < BM / 16) {
        int rb = (m0 >> 4) + blk;
        rb = rb < rb_max ? rb : rb_max;                   // rows past M: re-read the last block (masked at the store)
        base = reinterpret_cast<const char*>(A + (int64_t)rb * nkb_pad * (2048 / sizeof(TA)));
      } else {
        int nb = (n0 >> 4) + blk - BM / 16;
        nb = nb < nb_max ? nb : nb_max;
        base = reinterpret_cast<const char*>(W + (int64_t)nb * (int)((d.ldw + KB - 1) / KB) * (2048 / sizeof(TW)));
      }
      src[i] = base + (j & 1) * 1024 + lane * 16;
      dst[i] = j * 1024;
    }
    auto issue = [&](int slot, int kb) {
#pragma unroll
      for (int i = 0; i < PER; ++i) glds16(src[i] + (int64_t)kb * 2048, lds_b + slot * STAGE_BYTES + dst[i]);
    };
    // this wave's DMA pieces retire in order: "at most pend stages' worth outstanding" == every older stage has landed
    auto wait_pending = [&](int pend) {
      if (pend >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
      else if (pend == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
      else if (pend == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int nk = kb_hi - kb_lo;
#pragma unroll
    for (int s_ = 0; s_ < NST; ++s_)
      if (s_ < nk) issue(s_, kb_lo + s_);
    using V16 = std::remove_reference_t<decltype(((typename M_::AReg*)nullptr)->v[0])>;
    using W16 = std::remove_reference_t<decltype(((typename M_::WReg*)nullptr)->v[0])>;
    static_assert(sizeof(V16) == 16 && sizeof(W16) == 16, "fragment halves are 16 bytes");
    V16 fa[2][MF];
    W16 fw[2][NF];
    auto read_half = [&](auto buf_tag, int slot, int half) {
      constexpr int BUF = decltype(buf_tag)::value;
      const char* st = lds_b + slot * STAGE_BYTES + half * 1024 + lane * 16;
#pragma unroll
      for (int m = 0; m < MF; ++m) fa[BUF][m] = *reinterpret_cast<const V16*>(st + (wm * MF + m) * 2048);
#pragma unroll
      for (int n = 0; n < NF; ++n) fw[BUF][n] = *reinterpret_cast<const W16*>(st + (BM / 16 + wn * NF + n) * 2048);
    };
    // A unit = the MF x NF MFMAs of one register set.  Its FIRST MFMA is where hipcc puts the lgkmcnt wait for the set
    // (a full drain: across the loop's back edge its waitcnt pass does not count), so the ds_reads of the NEXT set are
    // issued right behind that first MFMA -- nothing else is outstanding at the wait -- and fly under the other MF*NF-1.
    auto mma_first = [&](auto buf_tag) {
      constexpr int BUF = decltype(buf_tag)::value;
      M_::mma16(acc[0][0], fa[BUF][0], fw[BUF][0]);
    };
    auto mma_rest = [&](auto buf_tag) {
      constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n)
          if (m + n > 0) M_::mma16(acc[m][n], fa[BUF][m], fw[BUF][n]);
    };
    // publish stage s (>= 1).  A wave gets here right behind the first MFMA of its last unit of stage s-1, i.e. with every
    // read of stage s-1 returned: own pieces of stage s landed -> barrier (everybody's pieces landed, nobody reads slot
    // s-1 any more) -> that slot is refilled with stage s+NST-1.  NST-1 stages are in flight while one is multiplied.
    auto sync_stage = [&](int s) {
      const int newer = nk - 1 - s;                                    // stages after s that exist
      wait_pending(newer < NST - 2 ? newer : NST - 2);
      // (the reads of stage s-1 were issued a unit ago and have returned; the explicit drain makes the refill below safe
      // whatever partial lgkmcnt hipcc chooses for the MFMA in front of this call)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (s + NST - 1 < nk) issue((s + NST - 1) % NST, kb_lo + s + NST - 1);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    if (tr0) tr0[5] = clock64();
    {
      const int newer = nk - 1;
      wait_pending(newer < NST - 1 ? newer : NST - 1);
      asm volatile("s_barrier" ::: "memory");
    }
    if (tr0) tr0[6] = clock64();
    int slot = 0;
    if constexpr (WK == 1) {
      read_half(B0{}, 0, 0);
      for (int i = 0; i < nk; ++i) {
        mma_first(B0{});                                   // (stage i, half 0)
        __builtin_amdgcn_sched_barrier(0);
        read_half(B1{}, slot, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_rest(B0{});
        mma_first(B1{});                                   // (stage i, half 1)
        __builtin_amdgcn_sched_barrier(0);
        slot = slot + 1 == NST ? 0 : slot + 1;
        if (i + 1 < nk) {
          sync_stage(i + 1);
          read_half(B0{}, slot, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_rest(B1{});
      }
    } else {
      read_half(B0{}, 0, wk);
      for (int i = 0; i < nk; i += 2) {
        mma_first(B0{});                                   // stage i (this wave's half)
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < nk) {
          slot = slot + 1 == NST ? 0 : slot + 1;
          sync_stage(i + 1);
          read_half(B1{}, slot, wk);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_rest(B0{});
        if (i + 1 < nk) {
          mma_first(B1{});                                 // stage i+1
          __builtin_amdgcn_sched_barrier(0);
          if (i + 2 < nk) {
            slot = slot + 1 == NST ? 0 : slot + 1;
            sync_stage(i + 2);
            read_half(B0{}, slot, wk);
          }
          __builtin_amdgcn_sched_barrier(0);
          mma_rest(B1{});
        }
      }
    }
    __syncthreads();                                     // the stages are dead: the epilogue slab re-uses their bytes
  } else
  if constexpr (LDSK) {
    // ---- LDS-staged K loop.  Stage = BM/16 A blocks + BN/16 W blocks of 2 KB (one fragment block = 64 lanes x 32 B, the
    // same bytes in HBM and in LDS, so the DMA's "wave-uniform base + lane * 16" destination needs no address math).
    // A stage is moved by 2*NBLK DMA instructions of 1 KB, dealt round-robin to the waves; when they do not divide evenly
    // the surplus slots repeat the last piece (same bytes to the same place), so every wave issues exactly PER
    // instructions per stage and the hand-counted vmcnt below is the same constant for all of them.
    static_assert(sizeof(TA) == sizeof(TW), "LDS-staged loop: A and W in the MFMA dtype (fragment order)");
    constexpr int NW = NT / 64;
    constexpr int NBLK = BM / 16 + BN / 16, STAGE_BYTES = NBLK * 2048, NINSTR = 2 * NBLK, PER = (NINSTR + NW - 1) / NW;
    constexpr int NST = STAGES;
    static_assert(NST >= 2 && NST <= 4 && PER * (NST - 1) < 64, "stage ring: 2..4 stages, vmcnt is a 6-bit count");
    char* lds_b = reinterpret_cast<char*>(smem);
    const int nkb_pad = (d.K + KB - 1) / KB;
    const int rb_max = (d.M + 15) / 16 - 1, nb_max = (d.N + 15) / 16 - 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* src[PER];
    int dst[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int j = wave_u + i * NW;                            // piece 0 .. NINSTR-1: A row blocks first, then W column blocks
      j = j < NINSTR ? j : NINSTR - 1;
      const int blk = j >> 1;
      const char* base;
      if (blk < BM / 16) {
        int rb = (m0 >> 4) + blk;
        rb = rb < rb_max ? rb : rb_max;                   // rows past M: re-read the last block (masked at the store)
        base = reinterpret_cast<const char*>(A + (int64_t)rb * nkb_pad * (2048 / sizeof(TA)));
      } else {
        int nb = (n0 >> 4) + blk - BM / 16;
        nb = nb < nb_max ? nb : nb_max;
        base = reinterpret_cast<const char*>(W + (int64_t)nb * (int)((d.ldw + KB - 1) / KB) * (2048 / sizeof(TW)));
      }
      src[i] = base + (j & 1) * 1024 + lane * 16;
      dst[i] = j * 1024;
    }
    auto issue = [&](int slot, int kb) {
#pragma unroll
      for (int i = 0; i < PER; ++i) glds16(src[i] + (int64_t)kb * 2048, lds_b + slot * STAGE_BYTES + dst[i]);
    };
    // NST stages: while k-block i is multiplied, i+1 .. i+NST-2 are in flight and i+NST-1 is issued.  Waits are counted by
    // hand (hipcc cannot see what a DMA wrote): this wave's loads retire in order, so "at most `pending` stages' worth
    // outstanding" means stage i has landed; the barrier then makes every wave's share visible and proves nobody still
    // reads the slot that is refilled next (it was multiplied in iteration i-1).
    const int nk = kb_hi - kb_lo;
#pragma unroll
    for (int s_ = 0; s_ < NST - 1; ++s_)
      if (s_ < nk) issue(s_, kb_lo + s_);
    int slot = 0, fill = NST - 1;
    for (int i = 0; i < nk; ++i) {
      const int ahead = nk - 1 - i;
      const int pending = ahead < NST - 2 ? ahead : NST - 2;
      if (pending >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
      else if (pending == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (i + NST - 1 < nk) issue(fill, kb_lo + i + NST - 1);
      const char* st = lds_b + slot * STAGE_BYTES + lane * 16;
      typename M_::AReg af[MF];
      typename M_::WReg wf[NF];
      using V16 = std::remove_reference_t<decltype(af[0].v[0])>;
      using W16 = std::remove_reference_t<decltype(wf[0].v[0])>;
      static_assert(sizeof(V16) == 16 && sizeof(W16) == 16, "fragment halves are 16 bytes");
      if constexpr (WK == 1) {
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          const char* q = st + (wm * MF + m) * 2048;
          af[m].v[0] = *reinterpret_cast<const V16*>(q);
          af[m].v[1] = *reinterpret_cast<const V16*>(q + 1024);
        }
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          const char* q = st + (BM / 16 + wn * NF + n) * 2048;
          wf[n].v[0] = *reinterpret_cast<const W16*>(q);
          wf[n].v[1] = *reinterpret_cast<const W16*>(q + 1024);
        }
        M_::template mma<MF, NF>(acc, af, wf);
      } else {
        // WK = 2: the two waves of a pair split the contraction inside the k-block (one 1 KB half each)
        static_assert(WK <= 2, "LDS-staged loop: K over at most 2 waves");
        const int hoff = wk * 1024;
#pragma unroll
        for (int m = 0; m < MF; ++m) af[m].v[0] = *reinterpret_cast<const V16*>(st + (wm * MF + m) * 2048 + hoff);
#pragma unroll
        for (int n = 0; n < NF; ++n) wf[n].v[0] = *reinterpret_cast<const W16*>(st + (BM / 16 + wn * NF + n) * 2048 + hoff);
        M_::template mma_half<MF, NF>(acc, af, wf);
      }
      slot = slot + 1 == NST ? 0 : slot + 1;
      fill = fill + 1 == NST ? 0 : fill + 1;
    }
    __syncthreads();                                     // the stages are dead: the epilogue slab re-uses their bytes
  } else
  // (trace: tr0[5] = set-up done)
  // STAGES-deep register ring: (STAGES-1) k-blocks of loads in flight per wave.
  // prologue (uniform branches), branch-free steady state over full k-blocks, masked drain.
  {
    if (tr0) tr0[5] = clock64();
    const int kb_full = d.K / KB;                       // k-blocks [0, kb_full) are complete
    const int kb_hi_full = kb_hi < kb_full ? kb_hi : kb_full;
    int kb = kb_lo + wk;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (kb + s * WK < kb_hi) load(TailT{}, s, kb + s * WK);
    if constexpr (SMX) {
      // (max, sum exp) partials of the score GEMM -> (m, 1/Z) of this lane's rows: the 4 lanes that share a row split
      // the 32-key groups, then merge by shuffle.  Issued behind the first operand loads, 4 groups per row in flight
      // per round trip (a one-at-a-time loop is nt/4 dependent L2 latencies: 8 us at 1764 keys).
      constexpr int CHK = 4;
      const float2* ps[MF];
      float mx[MF], z[MF];
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        int r = m0 + wm * MF * 16 + m * 16 + (lane & 15);
        r = r < d.M ? r : d.M - 1;
        ps[m] = reinterpret_cast<const float2*>(d.sm_stats) + (int64_t)r * d.sm_nt;
        mx[m] = -INFINITY; z[m] = 0.f;
      }
      for (int t0 = g; t0 < d.sm_nt; t0 += 4 * CHK) {
        float2 v[MF][CHK];
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int c = 0; c < CHK; ++c) {
            const int t = t0 + 4 * c;
            v[m][c] = ps[m][t < d.sm_nt ? t : d.sm_nt - 1];          // clamped, unconditional
          }
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          float cm = mx[m];
#pragma unroll
          for (int c = 0; c < CHK; ++c) cm = (t0 + 4 * c < d.sm_nt) ? fmaxf(cm, v[m][c].x) : cm;
          float zc = z[m] * __expf(mx[m] - cm);                       // (first round: 0 * exp(-inf) = 0)
#pragma unroll
          for (int c = 0; c < CHK; ++c) zc += (t0 + 4 * c < d.sm_nt) ? v[m][c].y * __expf(v[m][c].x - cm) : 0.f;
          mx[m] = cm; z[m] = zc;
        }
      }
#pragma unroll
      for (int m = 0; m < MF; ++m) {
#pragma unroll
        for (int o_ = 16; o_ < 64; o_ <<= 1) {
          const float mo = __shfl_xor(mx[m], o_), zo = __shfl_xor(z[m], o_);
          const float mn = fmaxf(mx[m], mo);
          z[m] = (mx[m] == -INFINITY ? 0.f : z[m] * __expf(mx[m] - mn)) + (mo == -INFINITY ? 0.f : zo * __expf(mo - mn));
          mx[m] = mn;
        }
        sm_c[m] = -mx[m] * 1.44269504088896341f - __log2f(z[m]);
        if (d.sm_zout && tile_n == 0 && wk == 0 && wn == 0 && g == 0) {      // (m, 1/Z) of the row for sp3_colsum_softmax
          const int r = m0 + wm * MF * 16 + m * 16 + (lane & 15);
          if (r < d.M) { d.sm_zout[4 * (int64_t)r + 1] = mx[m]; d.sm_zout[4 * (int64_t)r + 2] = 1.0f / z[m]; }
        }
      }
    }
    if (kb + (2 * STAGES - 2) * WK < kb_hi_full) {
      // the prologue stages are full blocks here, but were loaded through the masked path: consume them masked once
      do {
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
          load(FullT{}, (s + STAGES - 1) % STAGES, kb + (s + STAGES - 1) * WK);
          consume(TailT{}, s);
        }
        kb += STAGES * WK;
      } while (false);
      while (kb + (2 * STAGES - 2) * WK < kb_hi_full) {
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
          load(FullT{}, (s + STAGES - 1) % STAGES, kb + (s + STAGES - 1) * WK);
          consume(FullT{}, s);
        }
        kb += STAGES * WK;
      }
    }
    for (; kb < kb_hi; kb += STAGES * WK) {
#pragma unroll
      for (int s = 0; s < STAGES; ++s) {
        const int kcur = kb + s * WK;
        if (kcur < kb_hi) {
          const int knext = kcur + (STAGES - 1) * WK;
          if (knext < kb_hi) load(TailT{}, (s + STAGES - 1) % STAGES, knext);
          consume(TailT{}, s);
        }
      }
    }
  }

  if (tr0) { asm volatile("" ::"v"(acc[0][0][0])); tr0[1] = clock64(); }
  // Residual rows of the generic epilogue's FIRST iteration (the only one of the 32x32 tile: BM * BN / 4 == NT), requested
  // before the accumulators meet in LDS: the stream x was written launches ago by other XCDs, so loaded inside the
  // epilogue loop it is an exposed miss (~1400 clk) after the last MFMA.
  float4 pre_r1 = make_float4(0.f, 0.f, 0.f, 0.f), pre_r2 = pre_r1;
  bool pre_res = false;
  if constexpr (!LDSK) {
    const bool al = ((reinterpret_cast<uintptr_t>(d.res1) | reinterpret_cast<uintptr_t>(d.res2)) & 15) == 0 && ((d.ldr1 | d.ldr2) & 3) == 0;
    if ((d.res1 || d.res2) && al && d.epi == SP3_EPI_PLAIN && (d.N & 3) == 0 && tid < BM * (BN / 4)) {
      const int prow = tid / (BN / 4);
      const int pgm = m0 + prow, pgn = n0 + ec4;
      if (pgm < d.M && pgn < d.N) {
        pre_res = true;
        if (d.res1) pre_r1 = *reinterpret_cast<const float4*>(d.res1 + ((int64_t)bz * d.M + pgm) * d.ldr1 + pgn);
        if (d.res2) pre_r2 = *reinterpret_cast<const float4*>(d.res2 + ((int64_t)bz * d.M + pgm) * d.ldr2 + pgn);
      }
    }
  }
  if constexpr (XACC) {           // f16x3: fold the cross terms in (x = h + l 2^-11)
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][r] = fmaf(accx[m][n][r], 1.0f / 2048.0f, acc[m][n][r]);
  }
  // ---- accumulators -> LDS (C layout: col = lane&15, row = 4*(lane>>4) + reg)
  if (LOOP < 2 || wave < NCW) {
    float* slab = smem + (size_t)wk * BM * LDS_LD;
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * MF * 16 + m * 16 + 4 * g + r;
          const int col = wn * NF * 16 + n * 16 + (lane & 15);
          slab[row * LDS_LD + col] = acc[m][n][r];
        }
    if constexpr (SMX) {        // spare column BN of the slab row: this wave's share of the row's kept probability mass
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        float z = sm_z[m];
        z += __shfl_xor(z, 16);
        z += __shfl_xor(z, 32);
        if (g == 0 && wn == 0) slab[(wm * MF * 16 + m * 16 + (lane & 15)) * LDS_LD + BN] = z;
      }
    }
  }
  // folded LayerNorm: finish mean / rstd of this tile's rows (rowstat does not alias the slabs: one barrier covers both)
  float* rowstat = smem + (size_t)WK * BM * LDS_LD;     // [BM][2]
  if (LNF && d.ln_stats && srow < BM) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < NPRE; ++q)
      if (sj + q * TPR < d.ln_nt) { s1 += sp[q].x; s2 += sp[q].y; }
    for (int t = sj + NPRE * TPR; t < d.ln_nt; t += TPR) { const float2 v = sps[t]; s1 += v.x; s2 += v.y; }
#pragma unroll
    for (int o_ = 1; o_ < TPR; o_ <<= 1) { s1 += __shfl_xor(s1, o_); s2 += __shfl_xor(s2, o_); }
    if (sj == 0) {
      const float mean = s1 / (float)d.ln_C;
      const float var = fmaxf(s2 / (float)d.ln_C - mean * mean, 0.f);
      rowstat[2 * srow] = mean;
      rowstat[2 * srow + 1] = 1.0f / sqrtf(var + d.ln_eps);
    }
  }
  __syncthreads();
  if constexpr (LOOP >= 2) {
    if (d.trace && blockIdx.x < 8 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) d.trace[blockIdx.x * 64 + 48] = clock64();
  }
  if (tr0) tr0[48] = clock64();

  // y = rstd * acc - rstd * mean * s[n]   (bias, already folded with beta . W^T, is added by the epilogues below)
  auto ln_fold = [&](float acc, int row, int gn) -> float {
    const float mean = rowstat[2 * row], rstd = rowstat[2 * row + 1];
    return rstd * acc - rstd * mean * d.ln_s[gn];
  };
  // the same for 4 consecutive columns gn..gn+3 (gn % 4 == 0, all inside N)
  auto ln_fold4 = [&](float (&x)[4], int row, int gn) {
    const float mean = rowstat[2 * row], rstd = rowstat[2 * row + 1];
    const float4 s4 = *reinterpret_cast<const float4*>(d.ln_s + gn);
    const float rm = rstd * mean;
    x[0] = rstd * x[0] - rm * s4.x; x[1] = rstd * x[1] - rm * s4.y;
    x[2] = rstd * x[2] - rm * s4.z; x[3] = rstd * x[3] - rm * s4.w;
  };
  auto add4 = [&](float (&x)[4], const float* p) {
    const float4 b4 = *reinterpret_cast<const float4*>(p);
    x[0] += b4.x; x[1] += b4.y; x[2] += b4.z; x[3] += b4.w;
  };

  auto lds_sum4 = [&](int row, int c4) -> float4 {
    float4 v = *reinterpret_cast<const float4*>(smem + row * LDS_LD + c4);
#pragma unroll
    for (int s = 1; s < WK; ++s) {
      const float4 t = *reinterpret_cast<const float4*>(smem + (size_t)s * BM * LDS_LD + row * LDS_LD + c4);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    return v;
  };

  const float alpha = d.alpha;

  if (d.epi == SP3_EPI_ROPE_VT && n0 >= d.rope_cols) {
    // ---------------- V part: store transposed per head, vt[((b*heads+h)*64+dd)*vt_ld + n]
    TW* vt = reinterpret_cast<TW*>(d.vt);
    if ((d.tokens & 3) == 0) {
      // 4 consecutive tokens of one column per thread: they are contiguous in both V layouts (8- or 16-byte store)
      for (int idx = tid; idx < (BM / 4) * BN; idx += NT) {
        const int rq = idx % (BM / 4), col = idx / (BM / 4);
        const int gm = m0 + 4 * rq, gn = n0 + col;
        if (gm >= d.M || gn >= d.N) continue;                       // M % 4 == 0 here (M = B * tokens)
        const float bias = d.bias ? d.bias[gn] : 0.f;
        const float sn_ = d.ln_stats ? d.ln_s[gn] : 0.f;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 4 * rq + i;
          float x = smem[row * LDS_LD + col];
#pragma unroll
          for (int s_ = 1; s_ < WK; ++s_) x += smem[(size_t)s_ * BM * LDS_LD + row * LDS_LD + col];
          x *= alpha;
          if (d.ln_stats) { const float mean = rowstat[2 * row], rstd = rowstat[2 * row + 1]; x = rstd * x - rstd * mean * sn_; }
          v[i] = x + bias;
        }
        const int vc = gn - d.rope_cols;
        const int h = vc >> 6, dd = vc & 63;
        const int b = gm / d.tokens, n = gm - b * d.tokens;
        int64_t off;
        if (d.qkv_packed) {
          // PV-operand order [(b,h)][key/32][d/16][lane = 16*g + d%16][8]: the 8 keys a lane feeds to one
          // v_mfma_f32_16x16x32 (keys 32u + 16*(e>>2) + 4g + (e&3)) are contiguous -> 1 KB contiguous per wave load
          const int u = n >> 5, kk = n & 31, w16 = kk & 15;
          const int e = 4 * (kk >> 4), lane_ = (w16 >> 2) * 16 + (dd & 15);
          const int64_t nU = d.vt_ld >> 5;
          off = ((((int64_t)(b * d.heads + h) * nU + u) * 4 + (dd >> 4)) * 64 + lane_) * 8 + e;
        } else {
          off = ((int64_t)(b * d.heads + h) * 64 + dd) * d.vt_ld + n;
        }
        if constexpr (sizeof(TW) == 2) {
          bf16x4 ob;
          ob[0] = (__bf16)v[0]; ob[1] = (__bf16)v[1]; ob[2] = (__bf16)v[2]; ob[3] = (__bf16)v[3];
          *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(vt) + off) = ob;
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(vt) + off) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      return;
    }
    for (int idx = tid; idx < BM * BN; idx += NT) {
      const int row = idx % BM, col = idx / BM;
      const int gm = m0 + row, gn = n0 + col;
      if (gm >= d.M || gn >= d.N) continue;
      float v = smem[row * LDS_LD + col];
#pragma unroll
      for (int s = 1; s < WK; ++s) v += smem[(size_t)s * BM * LDS_LD + row * LDS_LD + col];
      v = v * alpha;
      if (d.ln_stats) v = ln_fold(v, row, gn);
      v += (d.bias ? d.bias[gn] : 0.f);
      const int vc = gn - d.rope_cols;
      const int h = vc >> 6, dd = vc & 63;
      const int b = gm / d.tokens, n = gm - b * d.tokens;
      if (d.qkv_packed) {
        const int u = n >> 5, kk = n & 31, w16 = kk & 15;
        const int e = (w16 & 3) + 4 * (kk >> 4), lane_ = (w16 >> 2) * 16 + (dd & 15);
        const int64_t nU = d.vt_ld >> 5;
        vt[((((int64_t)(b * d.heads + h) * nU + u) * 4 + (dd >> 4)) * 64 + lane_) * 8 + e] = (TW)v;
      } else {
        vt[((int64_t)(b * d.heads + h) * 64 + dd) * d.vt_ld + n] = (TW)v;
      }
    }
    return;
  }

  // ---------------- plain epilogue of the LDS-staged / role tiles: a thread keeps ONE column group (NT is a multiple of
  // BN/4) and walks the rows, so bias and the LayerNorm column sums are loaded once, and the residual rows of all its
  // iterations are requested up front (in the generic loop below every iteration is a dependent L2 round trip: loads of
  // iteration i+1 cannot be hoisted above the stores of iteration i)
  if constexpr (LDSK && (NT % (BN / 4)) == 0) {
    constexpr int CG = BN / 4, RSTEP = NT / CG, ITER = (BM + RSTEP - 1) / RSTEP;
    // (the pipelined tiles walk up to 16 row steps per thread: residual rows are requested 4 steps at a time)
    if (d.epi == SP3_EPI_PLAIN && (d.N & 3) == 0 && (ITER <= 4 || LOOP == -1) && !d.sm_stats_out) {
      const int c4 = ec4, gn = n0 + c4, row0 = tid / CG;
      if (gn < d.N) {
        const float4 b4 = pre_b4, s4 = pre_s4;            // (N % 4 == 0: the group is whole, epre held)
        constexpr int CHK = ITER < 4 ? ITER : 4;
#pragma unroll
        for (int it0 = 0; it0 < ITER; it0 += CHK) {
        float4 r1[CHK], r2[CHK];
        int gmc[CHK];
#pragma unroll
        for (int it = 0; it < CHK; ++it) {
          const int gm = m0 + row0 + (it0 + it) * RSTEP;
          gmc[it] = gm < d.M ? gm : d.M - 1;              // clamped: unconditional loads
          r1[it] = r2[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (d.res1) {
#pragma unroll
          for (int it = 0; it < CHK; ++it) r1[it] = *reinterpret_cast<const float4*>(d.res1 + ((int64_t)bz * d.M + gmc[it]) * d.ldr1 + gn);
        }
        if (d.res2) {
#pragma unroll
          for (int it = 0; it < CHK; ++it) r2[it] = *reinterpret_cast<const float4*>(d.res2 + ((int64_t)bz * d.M + gmc[it]) * d.ldr2 + gn);
        }
#pragma unroll
        for (int it = 0; it < CHK; ++it) {
          const int row = row0 + (it0 + it) * RSTEP, gm = m0 + row;
          if (row < BM && gm < d.M) {
            const float4 a4 = lds_sum4(row, c4);
            float v[4] = {a4.x * alpha, a4.y * alpha, a4.z * alpha, a4.w * alpha};
            if (d.ln_stats) {
              const float mean = rowstat[2 * row], rstd = rowstat[2 * row + 1], rm = rstd * mean;
              v[0] = rstd * v[0] - rm * s4.x; v[1] = rstd * v[1] - rm * s4.y;
              v[2] = rstd * v[2] - rm * s4.z; v[3] = rstd * v[3] - rm * s4.w;
            }
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
            if (d.act == SP3_ACT_GELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            } else if (d.act == SP3_ACT_RELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            v[0] += r1[it].x + r2[it].x; v[1] += r1[it].y + r2[it].y; v[2] += r1[it].z + r2[it].z; v[3] += r1[it].w + r2[it].w;
            if (d.stats_out) {
              float s1 = (v[0] + v[1]) + (v[2] + v[3]);
              float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
              for (int o_ = 1; o_ < 8; o_ <<= 1) { s1 += __shfl_xor(s1, o_); s2 += __shfl_xor(s2, o_); }
              if (((gn >> 2) & 7) == 0)
                reinterpret_cast<float2*>(d.stats_out)[(int64_t)gm * (d.N >> 5) + (gn >> 5)] = make_float2(s1, s2);
            }
            if (d.c2) {
              const bool cb = d.wdtype == SP3_BF16;
              const int64_t o2 = packed_off(gm, gn, d.N, cb);
              if (cb) {
                bf16x4 ob;
                ob[0] = (__bf16)v[0]; ob[1] = (__bf16)v[1]; ob[2] = (__bf16)v[2]; ob[3] = (__bf16)v[3];
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(d.c2) + o2) = ob;
              } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.c2) + o2) = make_float4(v[0], v[1], v[2], v[3]);
              }
            }
            const int64_t off = d.out_packed ? packed_off(gm, gn, d.N, d.out_bf16 != 0) : (int64_t)bz * d.strideC + (int64_t)gm * d.ldc + gn;
            if (d.out_bf16) {
              __bf16* o = reinterpret_cast<__bf16*>(d.C) + off;
              bf16x4 ob;
              ob[0] = (__bf16)v[0]; ob[1] = (__bf16)v[1]; ob[2] = (__bf16)v[2]; ob[3] = (__bf16)v[3];
              if ((off & 3) == 0) *reinterpret_cast<bf16x4*>(o) = ob;
              else { o[0] = ob[0]; o[1] = ob[1]; o[2] = ob[2]; o[3] = ob[3]; }
            } else {
              float* o = reinterpret_cast<float*>(d.C) + off;
              if ((off & 3) == 0) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
              else { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
            }
          }
        }
        }
      }
      if constexpr (LOOP >= 2) {
        if (d.trace && blockIdx.x < 8 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
          d.trace[blockIdx.x * 64 + 2] = clock64();
          d.trace[blockIdx.x * 64 + 4] = wall_clock64();
        }
      }
      if (tr0) { tr0[2] = clock64(); tr0[4] = wall_clock64(); }
      return;
    }
  }

  const bool rope_epi = d.epi == SP3_EPI_ROPE_VT;
  for (int idx = tid; idx < BM * (BN / 4); idx += NT) {
    // RoPE pairs column c with c ^ 16: the partner group's bias / column sums sit 4 lanes away (whole waves enter an
    // iteration together: BM * BN / 4 is a multiple of 64), so they are fetched by shuffle instead of a second cold load
    float4 pb4 = pre_b4, ps4 = pre_s4;
    if (rope_epi) {
      pb4 = make_float4(__shfl_xor(pre_b4.x, 4), __shfl_xor(pre_b4.y, 4), __shfl_xor(pre_b4.z, 4), __shfl_xor(pre_b4.w, 4));
      ps4 = make_float4(__shfl_xor(pre_s4.x, 4), __shfl_xor(pre_s4.y, 4), __shfl_xor(pre_s4.z, 4), __shfl_xor(pre_s4.w, 4));
    }
    const int row = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
    const int gm = m0 + row, gn = n0 + c4;
    // (softmax statistics: the 8 lanes of a 32-column group stay together through the shuffles, also past N)
    if (gm >= d.M || (gn >= d.N && !d.sm_stats_out)) continue;
    float4 acc4 = lds_sum4(row, c4);
    float v[4] = {acc4.x * alpha, acc4.y * alpha, acc4.z * alpha, acc4.w * alpha};
    const int nvalid = (d.N - gn) < 4 ? (d.N - gn) : 4;
    if constexpr (SMX) {
      // renormalise by the kept probability mass of the row (summed over the K-split waves), hand it to the column sums
      float zs = smem[row * LDS_LD + BN];
#pragma unroll
      for (int s_ = 1; s_ < WK; ++s_) zs += smem[(size_t)s_ * BM * LDS_LD + row * LDS_LD + BN];
      const float izs = 1.0f / zs;
      v[0] *= izs; v[1] *= izs; v[2] *= izs; v[3] *= izs;
      if (gn == 0 && d.sm_zout) d.sm_zout[4 * (int64_t)gm] = zs;
    }

    if (d.epi == SP3_EPI_PARTIAL) {
      float* o = reinterpret_cast<float*>(d.C) + ((int64_t)kz * d.M + gm) * d.ldc + gn;
      if (nvalid == 4) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      else
        for (int e = 0; e < nvalid; ++e) o[e] = v[e];
      continue;
    }

    if (d.epi == SP3_EPI_ROPE_VT) {
      // bias, then RoPE with the partner column (col ^ 16 inside the 64-wide head); everything in groups of 4 columns
      // (rope_cols % 64 == 0, so the group and its partner group lie inside N and are 16-byte aligned)
      float4 part4 = lds_sum4(row, c4 ^ 16);
      float pv[4] = {part4.x * alpha, part4.y * alpha, part4.z * alpha, part4.w * alpha};
      const int hc = gn & 63;
      const int axis = hc >> 5, is_v = (hc >> 4) & 1, i0 = hc & 15;
      const int pos = d.pos[(int64_t)gm * 2 + axis];
      if (epre) {
        if (d.ln_stats) {
          const float mean = rowstat[2 * row], rstd = rowstat[2 * row + 1], rm = rstd * mean;
          v[0] = rstd * v[0] - rm * pre_s4.x; v[1] = rstd * v[1] - rm * pre_s4.y;
          v[2] = rstd * v[2] - rm * pre_s4.z; v[3] = rstd * v[3] - rm * pre_s4.w;
          pv[0] = rstd * pv[0] - rm * ps4.x; pv[1] = rstd * pv[1] - rm * ps4.y;
          pv[2] = rstd * pv[2] - rm * ps4.z; pv[3] = rstd * pv[3] - rm * ps4.w;
        }
        v[0] += pre_b4.x; v[1] += pre_b4.y; v[2] += pre_b4.z; v[3] += pre_b4.w;
        pv[0] += pb4.x; pv[1] += pb4.y; pv[2] += pb4.z; pv[3] += pb4.w;
      } else {
        if (d.ln_stats) { ln_fold4(v, row, gn); ln_fold4(pv, row, gn ^ 16); }
        if (d.bias) { add4(v, d.bias + gn); add4(pv, d.bias + (gn ^ 16)); }
      }
      const float4 cs4 = *reinterpret_cast<const float4*>(d.rope_cos + pos * 16 + i0);
      const float4 sn4 = *reinterpret_cast<const float4*>(d.rope_sin + pos * 16 + i0);
      const float cs[4] = {cs4.x, cs4.y, cs4.z, cs4.w}, sn[4] = {sn4.x, sn4.y, sn4.z, sn4.w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = is_v ? (v[e] * cs[e] + pv[e] * sn[e]) : (v[e] * cs[e] - pv[e] * sn[e]);
      TW* out;
      if (d.qkv_packed) {      // fragment order, rows padded per image to vt_ld (a multiple of 64) tokens
        const int b = gm / d.tokens, n = gm - b * d.tokens;
        out = reinterpret_cast<TW*>(d.C) + packed_off(b * (int)d.vt_ld + n, gn, d.rope_cols, true);
      } else {
        out = reinterpret_cast<TW*>(d.C) + (int64_t)bz * d.strideC + (int64_t)gm * d.ldc + gn;
      }
      if constexpr (sizeof(TW) == 2) {
        bf16x4 ob;
        ob[0] = (__bf16)o[0]; ob[1] = (__bf16)o[1]; ob[2] = (__bf16)o[2]; ob[3] = (__bf16)o[3];
        if ((reinterpret_cast<uintptr_t>(out) & 7) == 0) *reinterpret_cast<bf16x4*>(out) = ob;
        else { out[0] = ob[0]; out[1] = ob[1]; out[2] = ob[2]; out[3] = ob[3]; }
      } else {
        if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) *reinterpret_cast<float4*>(out) = make_float4(o[0], o[1], o[2], o[3]);
        else { out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = o[3]; }
      }
      continue;
    }

    // bias + activation
    if (epre) {                                           // c4 == ec4 for every iteration of this thread
      if (d.ln_stats) {
        const float mean = rowstat[2 * row], rstd = rowstat[2 * row + 1], rm = rstd * mean;
        v[0] = rstd * v[0] - rm * pre_s4.x; v[1] = rstd * v[1] - rm * pre_s4.y;
        v[2] = rstd * v[2] - rm * pre_s4.z; v[3] = rstd * v[3] - rm * pre_s4.w;
      }
      v[0] += pre_b4.x; v[1] += pre_b4.y; v[2] += pre_b4.z; v[3] += pre_b4.w;
    } else if (nvalid == 4 && d.epi != SP3_EPI_PIXSHUF) {
      if (d.ln_stats) ln_fold4(v, row, gn);
      if (d.bias) add4(v, d.bias + gn);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (e < nvalid) {
          float x = v[e];
          if (d.ln_stats) x = ln_fold(x, row, gn + e);
          if (d.epi == SP3_EPI_PIXSHUF) x += d.bias ? d.bias[(gn + e) % d.ps_C] : 0.f;
          else x += d.bias ? d.bias[gn + e] : 0.f;
          v[e] = x;
        }
      }
    }
    if (d.act == SP3_ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    } else if (d.act == SP3_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }

    int64_t off;
    if (d.epi == SP3_EPI_PIXSHUF) {
      const int kk = gn / d.ps_C, co = gn - kk * d.ps_C;
      const int ky = kk / d.ps_k, kx = kk - ky * d.ps_k;
      const int pp = d.ps_H * d.ps_W;
      const int b = gm / pp, rem = gm - b * pp;
      const int y = rem / d.ps_W, x = rem - y * d.ps_W;
      off = (((int64_t)b * d.ps_H * d.ps_k + y * d.ps_k + ky) * ((int64_t)d.ps_W * d.ps_k) + x * d.ps_k + kx) * d.ps_C + co;
    } else if (d.out_packed) {
      off = packed_off(gm, gn, d.N, d.out_bf16 != 0);      // C is the next GEMM's A operand, fragment order
    } else {
      off = (int64_t)bz * d.strideC + (int64_t)gm * d.ldc + gn;
    }
    if (pre_res && idx == tid) {                         // requested before the reduction (same row / column group)
      v[0] += pre_r1.x + pre_r2.x; v[1] += pre_r1.y + pre_r2.y; v[2] += pre_r1.z + pre_r2.z; v[3] += pre_r1.w + pre_r2.w;
    } else {
      if (d.res1) {
        const float* r = d.res1 + ((int64_t)bz * d.M + gm) * d.ldr1 + gn;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nvalid) v[e] += r[e];
      }
      if (d.res2) {
        const float* r = d.res2 + ((int64_t)bz * d.M + gm) * d.ldr2 + gn;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nvalid) v[e] += r[e];
      }
    }
    if (d.sm_stats_out) {
      // per-32-column (max, sum exp(x - max)) of the finished scores: 8 consecutive lanes share a row group
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nvalid) mx = fmaxf(mx, v[e]);
#pragma unroll
      for (int o_ = 1; o_ < 8; o_ <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o_));
      float se = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nvalid) se += __expf(v[e] - mx);
#pragma unroll
      for (int o_ = 1; o_ < 8; o_ <<= 1) se += __shfl_xor(se, o_);
      if (((gn >> 2) & 7) == 0)
        reinterpret_cast<float2*>(d.sm_stats_out)[(int64_t)gm * ((d.N + 31) >> 5) + (gn >> 5)] = make_float2(mx, se);
      if (nvalid <= 0) continue;
    }
    if (d.stats_out) {
      // per-32-column partial (sum, sum of squares) of the finished rows: 8 consecutive lanes share a row group
      float s1 = (v[0] + v[1]) + (v[2] + v[3]);
      float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
      for (int o_ = 1; o_ < 8; o_ <<= 1) { s1 += __shfl_xor(s1, o_); s2 += __shfl_xor(s2, o_); }
      if (((gn >> 2) & 7) == 0)
        reinterpret_cast<float2*>(d.stats_out)[(int64_t)gm * (d.N >> 5) + (gn >> 5)] = make_float2(s1, s2);
    }
    if (d.c2) {
      const bool cb = d.wdtype == SP3_BF16;
      const int64_t o2 = packed_off(gm, gn, d.N, cb);
      if (cb) {
        bf16x4 ob;
        ob[0] = (__bf16)v[0]; ob[1] = (__bf16)v[1]; ob[2] = (__bf16)v[2]; ob[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(d.c2) + o2) = ob;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.c2) + o2) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    if (d.out_bf16) {
      __bf16* o = reinterpret_cast<__bf16*>(d.C) + off;
      if (nvalid == 4 && ((off & 3) == 0)) {
        bf16x4 ob;
        ob[0] = (__bf16)v[0]; ob[1] = (__bf16)v[1]; ob[2] = (__bf16)v[2]; ob[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(o) = ob;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nvalid) o[e] = (__bf16)v[e];
      }
    } else {
      float* o = reinterpret_cast<float*>(d.C) + off;
      if (nvalid == 4 && ((off & 3) == 0)) {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nvalid) o[e] = v[e];
      }
    }
  }
  if constexpr (LOOP >= 2) {
    if (d.trace && blockIdx.x < 8 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
      d.trace[blockIdx.x * 64 + 2] = clock64();
      d.trace[blockIdx.x * 64 + 4] = wall_clock64();
    }
  }
  if (tr0) { tr0[2] = clock64(); tr0[4] = wall_clock64(); }
}

// sp3_gemm2: the second group of problems of the launch being dispatched (same kernel instance), or null
thread_local const sp3_gemm_desc* g_pair = nullptr;

template <typename TA, typename TW, int LOADER, int MF, int NF, int WM, int WN, int WK, int STAGES, int LOOP = 0, int XM = 0>
int launch(const sp3_gemm_desc& d, hipStream_t stream) {
  constexpr bool LDSK = LOOP != 0;
  constexpr int BM = MF * 16 * WM, BN = NF * 16 * WN, NT = 64 * (WM * WN * WK + (LOOP >= 2 ? LOOP : 0));
  auto grid_x = [](const sp3_gemm_desc& q) {
    const int mt = (q.M + BM - 1) / BM, nt = (q.N + BN - 1) / BN;
    return mt >= nt ? ((mt + 7) / 8) * 8 * nt : ((nt + 7) / 8) * 8 * mt;
  };
  int blocks = grid_x(d);
  size_t lds = ((size_t)WK * BM * (BN + 4) + 2 * BM) * sizeof(float);
  if (LDSK) {
    const size_t stages = (size_t)STAGES * (BM / 16 + (LOOP >= 2 ? 0 : BN / 16)) * 2048;      // the epilogue slab aliases the ring
    lds = lds > stages ? lds : stages;
  }
  auto kern = gemm_kernel<TA, TW, LOADER, MF, NF, WM, WN, WK, STAGES, LOOP, XM>;
  if (lds > 64 * 1024) {
    static bool raised = false;     // one-time opt-in to > 64 KiB of dynamic LDS for this instantiation
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { sp3_set_error("sp3_gemm: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return 2; }
      raised = true;
    }
  }
  GemmArgs a;
  a.d = d;
  a.nb1 = 0;
  int gy = d.batch;
  if (g_pair) {                      // workgroups past a group's own tile count exit at once
    a.d2 = *g_pair;
    a.nb1 = d.batch;
    gy += g_pair->batch;
    const int b2 = grid_x(*g_pair);
    blocks = blocks > b2 ? blocks : b2;
  } else {
    a.d2 = d;
  }
  a.xcd_slices = (gy == 1 && d.splitk >= 8 && d.splitk % 8 == 0 && blocks % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(blocks, gy, d.splitk), dim3(NT), lds, stream, a);
  SP3_LAUNCH_CHECK("sp3_gemm");
  return 0;
}

template <typename TA, typename TW, int LOADER>
int dispatch_tile(const sp3_gemm_desc& d, int tile, hipStream_t stream) {
  if constexpr (sizeof(TA) == 4 && sizeof(TW) == 4) {
    // fp32 operands on the register-ring tiles 0-2: the product mode (sp3_gemm_desc.f32x3) selects the kernel instance
    const int xm = d.f32x3;
    if (xm >= 1 && xm <= 4 && tile >= 0 && tile <= 2) {
#define SP3_XM_TILES(XM_)                                                                         \
      switch (tile) {                                                                             \
        case 0: return launch<TA, TW, LOADER, 2, 2, 1, 1, 4, 3, 0, XM_>(d, stream);               \
        case 1: return launch<TA, TW, LOADER, 2, 2, 2, 2, 1, 2, 0, XM_>(d, stream);               \
        default: return launch<TA, TW, LOADER, 2, 4, 2, 2, 1, 2, 0, XM_>(d, stream);              \
      }
      if (xm == 1) { SP3_XM_TILES(1) }
      if (xm == 2) { SP3_XM_TILES(2) }
      if (xm == 4 && d.w_packed == 2) { SP3_XM_TILES(5) }
      if (xm == 4) { SP3_XM_TILES(4) }
      SP3_XM_TILES(3)
#undef SP3_XM_TILES
    }
  }
  switch (tile) {
    case 0: return launch<TA, TW, LOADER, 2, 2, 1, 1, 4, 3>(d, stream);   // 32x32, K over 4 waves
    case 1: return launch<TA, TW, LOADER, 2, 2, 2, 2, 1, 2>(d, stream);   // 64x64, wave tile 32x32
    case 2: return launch<TA, TW, LOADER, 2, 4, 2, 2, 1, 2>(d, stream);   // 64x128, wave tile 32x64
    case 3: return launch<TA, TW, LOADER, 4, 4, 1, 1, 4, 3>(d, stream);   // 64x64, K over 4 waves
    case 4: case 18:                                                       // 32x64 / 16x64, K over 4 waves (bf16 operands)
      if constexpr (sizeof(TA) == 2 && LOADER == SP3_LOAD_PLAIN) {
        if (tile == 4) return launch<TA, TW, LOADER, 2, 4, 1, 1, 4, 3>(d, stream);
        return launch<TA, TW, LOADER, 1, 4, 1, 1, 4, 3>(d, stream);
      }
      sp3_set_error("sp3_gemm: tile %d needs bf16 A and the plain loader", tile);
      return 1;
    case 20: case 21: case 22: case 23: case 24: case 25:   // pipelined LDS-staged operands (many-row GEMMs; 24 / 25: the memory read)
      if constexpr (sizeof(TA) == sizeof(TW) && LOADER == SP3_LOAD_PLAIN) {
        if (d.a_packed && d.w_packed && !d.A2 && d.K % MM<TA, TW>::KB == 0) {
          switch (tile) {
            case 20: return launch<TA, TW, LOADER, 4, 4, 4, 2, 1, 3, -1>(d, stream);   // 256x128, 4x2 waves of 64x64, 3 slots
            case 21: return launch<TA, TW, LOADER, 4, 4, 2, 2, 2, 4, -1>(d, stream);   // 128x128, 2x2 waves of 64x64 x 2 K halves, 4 slots
            case 22: return launch<TA, TW, LOADER, 4, 4, 2, 1, 2, 3, -1>(d, stream);   // 128x64,  2x1 waves of 64x64 x 2 K halves
            case 24: return launch<TA, TW, LOADER, 4, 2, 1, 1, 2, 4, -1>(d, stream);   // 64x32,   1 wave pair, 4 slots (score GEMM of the read)
            case 25: return launch<TA, TW, LOADER, 2, 2, 1, 1, 2, 4, -1>(d, stream);   // 32x32,   1 wave pair, 4 slots
            default: return launch<TA, TW, LOADER, 4, 4, 1, 1, 2, 4, -1>(d, stream);   // 64x64,   1 wave pair (2 K halves), 4 slots
          }
        }
      }
      sp3_set_error("sp3_gemm: tile %d (pipelined LDS) needs fragment-order A and W in the MFMA dtype, whole k-blocks, no split A", tile);
      return 1;
    case 5: case 6: case 7: case 8: case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: case 17:     // LDS-staged operands
      if constexpr (sizeof(TA) == sizeof(TW) && LOADER == SP3_LOAD_PLAIN) {
        if (d.a_packed && d.w_packed && !d.A2 && d.K % MM<TA, TW>::KB == 0) {
          switch (tile) {
            case 5: return launch<TA, TW, LOADER, 4, 4, 2, 2, 1, 2, 1>(d, stream);    // 128x128, 2x2 waves of 64x64
            case 6: return launch<TA, TW, LOADER, 4, 2, 2, 2, 1, 2, 1>(d, stream);    // 128x64,  2x2 waves of 64x32
            case 7: return launch<TA, TW, LOADER, 7, 1, 1, 4, 1, 4, 1>(d, stream);    // 112x64,  4 waves of 112x16, 4 slots
            case 8: return launch<TA, TW, LOADER, 13, 1, 1, 4, 1, 3, 1>(d, stream);   // 208x64,  4 waves of 208x16, 3 slots
            case 9: return launch<TA, TW, LOADER, 7, 1, 1, 2, 1, 4, 1>(d, stream);    // 112x32,  2 waves of 112x16, 4 slots
            case 10: return launch<TA, TW, LOADER, 7, 2, 1, 2, 1, 4, 1>(d, stream);   // 112x64,  2 waves of 112x32, 4 slots
            case 11: return launch<TA, TW, LOADER, 7, 1, 1, 4, 2, 4, 1>(d, stream);   // 112x64,  4x2 waves (K halves), 4 slots
            case 12: return launch<TA, TW, LOADER, 13, 1, 1, 4, 2, 3, 1>(d, stream);  // 208x64,  4x2 waves (K halves), 3 slots
            // weight streaming with roles: consumers keep their weights in a register ring, 2 loader waves DMA the activations
            default: break;
          }
          const int nkb_slice = ((d.K / MM<TA, TW>::KB) + d.splitk - 1) / d.splitk;
          if (d.ln_stats && d.ln_nt > 32) {
            sp3_set_error("sp3_gemm: tile %d folds a LayerNorm over at most 1024 columns (ln_C=%d)", tile, d.ln_C);
            return 1;
          }
          if (nkb_slice > kMaxKb) {
            sp3_set_error("sp3_gemm: tile %d streams at most %d k-blocks per K slice (K=%d, splitk=%d)", tile, kMaxKb, d.K, d.splitk);
            return 1;
          }
          if constexpr (sizeof(TA) == 2) switch (tile) {      // bf16 only (the fp32 mode is MFMA-bound, not weight-bound)
            case 13: return launch<TA, TW, LOADER, 7, 1, 1, 4, 2, 4, 2>(d, stream);   // 112x64,  4x2 consumer waves + 2 loaders
            case 14: return launch<TA, TW, LOADER, 7, 1, 1, 4, 1, 4, 2>(d, stream);   // 112x64,  4 consumer waves + 2 loaders
            case 15: return launch<TA, TW, LOADER, 13, 1, 1, 4, 2, 3, 2>(d, stream);  // 208x64,  4x2 consumer waves + 2 loaders
            case 16: return launch<TA, TW, LOADER, 7, 1, 1, 4, 1, 4, 6>(d, stream);   // 112x64,  4 consumer waves + 6 loaders
            case 17: return launch<TA, TW, LOADER, 7, 1, 1, 4, 2, 4, 4>(d, stream);   // 112x64,  4x2 consumer waves + 4 loaders
          }
        }
      }
      sp3_set_error("sp3_gemm: tile %d (LDS-staged) needs fragment-order A and W in the MFMA dtype, whole k-blocks, no split A", tile);
      return 1;
    default: sp3_set_error("sp3_gemm: bad tile %d", tile); return 1;
  }
}

}  // namespace

// validation, defaults and tile choice of one descriptor
static int gemm_prepare(sp3_gemm_desc& d, int& tile_out) {
  SP3_CHECK(d.A && d.W && d.C, "sp3_gemm: null A/W/C");
  SP3_CHECK(d.M > 0 && d.N > 0 && d.K > 0, "sp3_gemm: bad shape M=%d N=%d K=%d", d.M, d.N, d.K);
  SP3_CHECK(d.K % 8 == 0 || (d.loader == SP3_LOAD_SOFTMAX && d.K % 4 == 0), "sp3_gemm: K=%d must be a multiple of 8", d.K);
  // (sm_stats_out next to a fragment-order bf16 output = the score stage of the long-bank read, probabilities instead of scores: lean tile 45)
  SP3_CHECK(!d.sm_stats_out || (d.epi == SP3_EPI_PLAIN && d.N % 4 == 0 && (!d.out_packed == !d.out_bf16) && d.batch == 1),
            "sp3_gemm: sm_stats_out needs the plain epilogue (fp32 rows, or fragment-order bf16), N %% 4 == 0, one problem");
  SP3_CHECK(d.wdtype == SP3_F32 || d.wdtype == SP3_BF16, "sp3_gemm: bad wdtype %d", d.wdtype);
  SP3_CHECK(!d.a_bf16 || d.wdtype == SP3_BF16, "sp3_gemm: a_bf16 needs wdtype bf16");
  SP3_CHECK((reinterpret_cast<uintptr_t>(d.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(d.W) & 15) == 0,
            "sp3_gemm: A and W must be 16-byte aligned");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  SP3_CHECK(al16(d.bias) && al16(d.ln_s) && al16(d.ln_stats) && al16(d.rope_cos) && al16(d.rope_sin),
            "sp3_gemm: bias / ln_s / ln_stats / rope tables must be 16-byte aligned (vector loads in the epilogue)");
  if (d.batch <= 0) d.batch = 1;
  if (d.splitk <= 0) d.splitk = 1;
  if (d.ldw <= 0) d.ldw = d.K;
  SP3_CHECK(!d.ln_stats || (d.ln_s && d.ln_nt > 0 && d.ln_C == 32 * d.ln_nt && d.epi != SP3_EPI_PARTIAL &&
                            d.loader == SP3_LOAD_PLAIN),
            "sp3_gemm: folded LayerNorm needs ln_s, ln_nt, ln_C = 32*ln_nt, the plain loader");
  SP3_CHECK((!d.stats_out && !d.c2) || (d.epi == SP3_EPI_PLAIN && d.N % 32 == 0 && !d.out_packed),
            "sp3_gemm: stats_out / c2 need the plain epilogue, N %% 32 == 0");
  SP3_CHECK(!d.out_packed || (d.epi == SP3_EPI_PLAIN && d.N % 4 == 0), "sp3_gemm: out_packed needs the plain epilogue, N %% 4 == 0");
  // (a fragment-order A split along K -- A2 a second packed matrix with K - K1 columns -- is served by the lean instances only)
  SP3_CHECK(!d.a_packed || d.loader == SP3_LOAD_PLAIN || (d.loader == SP3_LOAD_SOFTMAX && d.a_bf16), "sp3_gemm: packed A needs the plain loader (or the softmax loader's probability form)");
  SP3_CHECK(!(d.a_packed && d.A2) || (d.K1 % 64 == 0 && d.K1 > 0 && d.K1 < d.K && d.K % 64 == 0),
            "sp3_gemm: packed split A needs K1 and K - K1 in whole 64-column blocks");
  SP3_CHECK(d.batch == 1 || ((d.sb_bias | d.sb_ln_stats | d.sb_ln_s | d.sb_stats_out | d.sb_c2 | d.sb_vt | d.sb_A2) & 15) == 0,
            "sp3_gemm: per-batch byte offsets must keep 16-byte alignment");
  SP3_CHECK(!d.a_packed || ((d.a_bf16 != 0) == (d.wdtype == SP3_BF16)), "sp3_gemm: packed A must have the MFMA dtype (its fragment geometry)");
  if (!d.A2) d.K1 = d.K;
  const int aalign = d.a_bf16 ? 8 : 4;    // elements per 16 bytes
  SP3_CHECK(d.ldw >= d.K && d.ldw % 8 == 0, "sp3_gemm: ldw=%lld must be >= K and a multiple of 8", (long long)d.ldw);
  SP3_CHECK(!d.A2 || (d.loader == SP3_LOAD_PLAIN && d.K1 % 64 == 0 && d.K1 > 0 && d.K1 < d.K && (d.a_packed || d.lda2 % aalign == 0)),
            "sp3_gemm: bad split-A configuration (K1=%d)", d.K1);
  SP3_CHECK(d.splitk == 1 || d.epi == SP3_EPI_PARTIAL, "sp3_gemm: splitk > 1 needs the PARTIAL epilogue");
  if (d.loader == SP3_LOAD_PLAIN && !d.a_packed) {
    SP3_CHECK(d.lda % aalign == 0 && d.lda >= d.K1, "sp3_gemm: lda=%lld must be >= K and keep 16-byte rows", (long long)d.lda);
  } else if (d.loader == SP3_LOAD_PLAIN) {
    /* packed A: geometry is implied by M, K */
  } else if (d.loader == SP3_LOAD_SOFTMAX && d.a_packed) {
    // probability form (long-bank read, lean tile 46): A = fragment-order bf16 p~ with the group scales in sm_stats, split-K partials
    SP3_CHECK(d.a_bf16 && !d.A2 && d.epi == SP3_EPI_PARTIAL && d.batch == 1 && !d.ln_stats && !d.relu_in && d.sm_stats && d.K % 4 == 0,
              "sp3_gemm: the softmax loader's probability form takes fragment-order bf16 p~, group scales (sm_stats) and the PARTIAL epilogue");
  } else if (d.loader == SP3_LOAD_SOFTMAX) {
    SP3_CHECK(!d.a_bf16 && !d.a_packed && !d.A2 && d.epi == SP3_EPI_PLAIN && d.splitk == 1 && d.batch == 1 && !d.ln_stats && !d.relu_in,
              "sp3_gemm: the softmax loader takes row-major fp32 scores, plain epilogue, one problem, no split-K");
    SP3_CHECK(d.sm_stats && d.sm_nt == (d.K + 31) / 32 && d.lda % 4 == 0 && d.lda >= (d.K + 15) / 16 * 16 && d.K % 4 == 0,
              "sp3_gemm: softmax loader needs sm_stats[M][ceil(K/32)][2], K %% 4 == 0 and score rows padded to 16 (lda=%lld K=%d)",
              (long long)d.lda, d.K);
    SP3_CHECK(d.tile <= 1 || d.tile == 44, "sp3_gemm: the softmax loader runs on the 16x64 tile (tile <= 0), the 32x32 one (tile 1) or the lean instance (44)");
  } else if (d.loader == SP3_LOAD_CONV3X3) {
    SP3_CHECK(d.conv_C % 16 == 0, "sp3_gemm: conv Cin=%d must be a multiple of 16", d.conv_C);
    SP3_CHECK(d.K == 9 * d.conv_C, "sp3_gemm: conv K=%d != 9*Cin", d.K);
    SP3_CHECK(d.conv_stride == 1 || d.conv_stride == 2, "sp3_gemm: conv stride %d", d.conv_stride);
    SP3_CHECK(d.M % (d.conv_OH * d.conv_OW) == 0, "sp3_gemm: conv M=%d not a multiple of OH*OW", d.M);
  } else {
    SP3_CHECK(false, "sp3_gemm: bad loader %d", d.loader);
  }
  if (d.epi == SP3_EPI_ROPE_VT) {
    SP3_CHECK(d.rope_cols % 64 == 0 && d.N % 64 == 0, "sp3_gemm: ROPE_VT needs 64-wide heads (N=%d rope_cols=%d)", d.N, d.rope_cols);
    SP3_CHECK(d.rope_cols == 0 || (d.rope_cos && d.rope_sin && d.pos), "sp3_gemm: ROPE_VT needs tables and positions");
    SP3_CHECK(d.rope_cols == d.N || (d.vt && d.tokens > 0 && d.heads > 0 && d.vt_ld >= d.tokens), "sp3_gemm: ROPE_VT needs vt/tokens/heads");
    SP3_CHECK(!d.qkv_packed || (d.wdtype == SP3_BF16 && d.tokens > 0 && d.vt_ld % 64 == 0 && d.vt_ld >= d.tokens),
              "sp3_gemm: qkv_packed needs bf16, tokens and vt_ld (padded tokens per image, multiple of 64)");
  } else if (d.epi == SP3_EPI_PIXSHUF) {
    SP3_CHECK(d.ps_k > 0 && d.ps_C > 0 && d.ps_C % 4 == 0 && d.N == d.ps_k * d.ps_k * d.ps_C && d.M % (d.ps_H * d.ps_W) == 0,
              "sp3_gemm: bad PIXSHUF geometry");
  } else if (d.epi == SP3_EPI_PARTIAL) {
    SP3_CHECK(d.batch == 1 && d.ldc % 4 == 0 && d.ldc >= d.N, "sp3_gemm: PARTIAL needs batch 1 and ldc %% 4 == 0");
  } else {
    SP3_CHECK(d.epi == SP3_EPI_PLAIN, "sp3_gemm: bad epilogue %d", d.epi);
  }
  int tile = d.tile;
  if (tile < 0) {
    const long sk = d.splitk;
    const long t64 = (long)((d.M + 63) / 64) * ((d.N + 63) / 64) * d.batch * sk;
    const long t128 = (long)((d.M + 63) / 64) * ((d.N + 127) / 128) * d.batch;
    const bool lds_ok = d.loader == SP3_LOAD_PLAIN && !d.sm_stats_out && sk == 1 && d.a_packed && d.w_packed && d.a_bf16 && !d.A2 &&
                        d.epi != SP3_EPI_PARTIAL && d.K % 64 == 0;
    constexpr bool pipe_on = true;
    // (split-K partials too: the PARTIAL epilogue is tile-agnostic; every K slice must hold whole k-blocks)
    const bool pipe_ok = d.loader == SP3_LOAD_PLAIN && !d.sm_stats_out && d.a_packed && d.w_packed && d.a_bf16 && !d.A2 && d.K % 64 == 0 &&
                         (sk == 1 || d.epi == SP3_EPI_PARTIAL);
    // In the model (profiles/r03_a_*): the RoPE / V^T epilogue and the GELU -> packed epilogue of the WIDE many-row GEMMs (q/k/v
    // projection, fc1: N >= 2304) run faster on the 128-row LDS tiles 5 / 6 (their epilogue walks 4 row steps per thread, the
    // 256 x 128 tile 16: 39.7 vs 29.2 us for the encoder's q/k/v), so those keep the round-2 choice; everything else with
    // many rows (N < 2304: proj, fc2, key MLPs, decoder side GEMMs; split-K partials; very wide score GEMMs of the long-bank
    // read) goes to the pipelined tiles.
    const bool wide_legacy = lds_ok && d.M >= 1024 && d.N >= 2304 && d.N % 128 == 0 && d.N < 16384;
    if (pipe_ok && pipe_on && d.M >= 512 && d.epi != SP3_EPI_ROPE_VT && !wide_legacy) {
      // the largest pipelined tile that still gives most CUs a workgroup (tools/bench_gemm2.py --big on MI355X,
      // profiles/r03_gemm_manyrow_tile_sweep.txt)
      const long mt256 = (d.M + 255) / 256, mt128 = (d.M + 127) / 128, nt128 = (d.N + 127) / 128, nt64 = (d.N + 63) / 64;
      if (mt256 * nt128 * d.batch * sk >= 144) tile = 20;
      else if (mt128 * nt128 * d.batch * sk >= 200) tile = 21;
      // fewer than ~200 workgroups of 128 x 64 (the 1024-row proj / fc2 GEMMs of a 512 x 512 frame): the 64 x 32 tile's 768 / 512
      // workgroups finish first (10.9 vs 13.5 us, 28.2 vs 36.7 us at K = 4096; the 1960-row encoder GEMMs stay at 128 x 64)
      else if (mt128 * nt64 * d.batch * sk <= 192) tile = 24;
      else tile = 22;
    } else if (lds_ok && d.M >= 1024 && d.N >= 2304 && d.N % 128 == 0) {
      // LDS-staged operands (tools/bench_gemm2.py --big, HBM-cold weights, also grouped launches): 128x128 for large grids
      // / N multiple of 4096, else 128x64
      tile = (d.N % 4096 == 0 || (long)((d.M + 127) / 128) * (d.N / 128) * d.batch >= 1024) ? 5 : 6;
    } else if (d.loader == SP3_LOAD_SOFTMAX) {
      tile = 0;
    } else if (d.loader != SP3_LOAD_CONV3X3 && d.M >= 1024) {
      tile = 1;                                           // many rows, narrow N (also split-K partials): 64x64 register tiles
      // fp32 operands (the fp32 / f32x6 / f16x3 modes' whole-sequence encoder): the register-ring tiles are bound by L2 -> CU bytes,
      // the 64 x 128 tile moves a quarter fewer per product -- wide N only (f16x3 at 1960 rows: q/k/v 85.5 -> 72.4 us, fc1 113.5 ->
      // 86.4; N = 1024 leaves 248 workgroups and loses: 102.8 -> 122.4 us; profiles/r06_f16x3_manyrow_tiles.txt)
      if (d.wdtype == SP3_F32 && !d.a_bf16 && sk == 1 && d.epi != SP3_EPI_PARTIAL && d.N % 128 == 0 && t128 >= 512) tile = 2;
    } else if (lds_ok && d.M <= 224 && d.M > 112 && d.epi == SP3_EPI_PLAIN && d.N >= 3072 && d.K <= 1024 &&
               (!d.ln_stats || d.ln_nt <= 32)) {
      // the widest 196-row GEMMs (fc1 of the decoder pair and of the value encoder): 112x64 role tile -- consumers keep
      // their weights in a register ring, loader waves DMA the activations (10.6 vs 14.1 us, 11.4 vs 12.0 us cold)
      tile = 13;
    } else if (d.loader == SP3_LOAD_CONV3X3 || d.M > 2048) {
      if (t128 >= 1024 && d.N % 128 == 0) tile = 2;
      else if (t64 >= 512) tile = 1;
      else tile = 0;
    } else {
      // weight-streaming shapes (M ~ 196): measured on MI355X (tools/bench_gemm.py, packed operands) the 32x32 tile
      // with K split over the 4 waves wins on every hot-path shape: ~3 us fixed cost vs ~5 us for the 64x64 tile
      tile = 0;
      (void)t64;
    }
  }
  SP3_CHECK(!d.sm_stats_out || tile <= 3 || tile >= 20, "sp3_gemm: sm_stats_out needs a register-ring tile (0-3) or a pipelined one (20-25)");
  tile_out = tile;
  return 0;
}

static int gemm_dispatch(const sp3_gemm_desc& d, int tile, hipStream_t stream) {
  if (d.loader == SP3_LOAD_SOFTMAX) {
    // 16 x 64 tile, K over the 4 waves: the probabilities of a row are rebuilt by every column tile, so few wide ones
    // (32 x 32: 16.6 us at 196 x 1764, half of it exponentials)
    if (d.tile == 1) {
      if (d.wdtype == SP3_BF16) return launch<float, __bf16, SP3_LOAD_SOFTMAX, 2, 2, 1, 1, 4, 3>(d, stream);
      return launch<float, float, SP3_LOAD_SOFTMAX, 2, 2, 1, 1, 4, 3>(d, stream);
    }
    if (d.wdtype == SP3_BF16) return launch<float, __bf16, SP3_LOAD_SOFTMAX, 1, 4, 1, 1, 4, 3>(d, stream);
    return launch<float, float, SP3_LOAD_SOFTMAX, 1, 4, 1, 1, 4, 3>(d, stream);
  }
  if (d.wdtype == SP3_BF16) {
    if (d.a_bf16) {
      if (d.loader == SP3_LOAD_PLAIN) return dispatch_tile<__bf16, __bf16, SP3_LOAD_PLAIN>(d, tile, stream);
      return dispatch_tile<__bf16, __bf16, SP3_LOAD_CONV3X3>(d, tile, stream);
    }
    if (d.loader == SP3_LOAD_PLAIN) return dispatch_tile<float, __bf16, SP3_LOAD_PLAIN>(d, tile, stream);
    return dispatch_tile<float, __bf16, SP3_LOAD_CONV3X3>(d, tile, stream);
  } else {
    if (d.loader == SP3_LOAD_PLAIN) return dispatch_tile<float, float, SP3_LOAD_PLAIN>(d, tile, stream);
    return dispatch_tile<float, float, SP3_LOAD_CONV3X3>(d, tile, stream);
  }
}

extern "C" int sp3_gemm(const sp3_gemm_desc* dp, void* stream_) {
  SP3_CHECK(dp != nullptr, "sp3_gemm: null descriptor");
  sp3_gemm_desc d = *dp;
  int tile = -1;
  if (int rc = gemm_prepare(d, tile)) return rc;
  // the 196-row weight-streaming shapes of the per-frame step: lean shape-specialised instances (gemm_sm.hip, tiles 30..)
  if (d.tile >= 30 || (d.tile < 0 && sp3_gemm_sm_tile(d) >= 0)) return sp3_gemm_sm_launch(d, nullptr, reinterpret_cast<hipStream_t>(stream_));
  SP3_CHECK(!(d.a_packed && d.A2), "sp3_gemm: a fragment-order A split along K runs on the lean instances only (no instance for M=%d N=%d K=%d)", d.M, d.N, d.K);
  SP3_CHECK(!(d.res_bf16 && (d.res1 || d.res2)), "sp3_gemm: bf16 residual maps are read by the lean small-map convolutions only (res_bf16 = out_bf16 = a_bf16; M=%d N=%d K=%d)", d.M, d.N, d.K);
  SP3_CHECK(!d.dyn_n, "sp3_gemm: a device-side extent (dyn_n) is served by the lean memory-read instances only (tiles 43 / 44; M=%d N=%d K=%d)", d.M, d.N, d.K);
  return gemm_dispatch(d, tile, reinterpret_cast<hipStream_t>(stream_));
}

// The tile sp3_gemm would run this descriptor on when desc.tile < 0 (the lean instances 30.. included); < 0: invalid descriptor
// (sp3_last_error).  Hosts use it to label profiles and to decide whether two launches can be paired.
extern "C" int sp3_gemm_plan(const sp3_gemm_desc* dp) {
  if (!dp) return -1;
  sp3_gemm_desc d = *dp;
  d.tile = -1;
  int tile = -1;
  if (gemm_prepare(d, tile)) return -1;
  const int lean = sp3_gemm_sm_tile(d);
  return lean >= 0 ? lean : tile;
}

extern "C" int sp3_gemm2(const sp3_gemm_desc* ap, const sp3_gemm_desc* bp, void* stream_) {
  SP3_CHECK(ap != nullptr && bp != nullptr, "sp3_gemm2: null descriptor");
  sp3_gemm_desc a = *ap, b = *bp;
  int ta = -1, tb = -1;
  {
    // both groups on one lean instance (gemm_sm.hip): decided before the legacy tiles are consulted
    sp3_gemm_desc a0 = a, b0 = b;
    int t0 = -1, t1 = -1;
    const int want = a0.tile;
    a0.tile = b0.tile = -1;
    if (gemm_prepare(a0, t0) == 0 && gemm_prepare(b0, t1) == 0) {
      const int la = sp3_gemm_sm_tile(a0), lb = sp3_gemm_sm_tile(b0);
      // (only the q/k/v projections' ROPE instances take a second group; two STREAM / PACKED problems that share an instance
      //  pair on the general tiles below, as they did before the lean families existed)
      if (la >= 30 && la == lb && sp3_gemm_sm_pairs(a0) && (want < 0 || want == la)) return sp3_gemm_sm_launch(a0, &b0, reinterpret_cast<hipStream_t>(stream_));
    }
    SP3_CHECK(want < 30, "sp3_gemm2: tile %d asks for a lean instance that does not take this pair", want);
  }
  if (int rc = gemm_prepare(a, ta)) return rc;
  if (b.tile < 0) b.tile = ta;                   // the second group runs on the first one's kernel instance
  if (int rc = gemm_prepare(b, tb)) return rc;
  SP3_CHECK(!(a.res_bf16 && (a.res1 || a.res2)) && !(b.res_bf16 && (b.res1 || b.res2)), "sp3_gemm2: bf16 residual maps are read by the lean small-map convolutions only");
  SP3_CHECK(ta == tb && a.wdtype == b.wdtype && a.a_bf16 == b.a_bf16 && a.loader == b.loader && a.loader != SP3_LOAD_SOFTMAX,
            "sp3_gemm2: both groups must use one kernel instance (tile %d / %d, same dtypes and loader)", ta, tb);
  SP3_CHECK(a.splitk == 1 && b.splitk == 1 && !a.trace && !b.trace && !a.dyn_n && !b.dyn_n, "sp3_gemm2: no split-K, no trace, no device-side extent");
  g_pair = &b;
  const int rc = gemm_dispatch(a, ta, reinterpret_cast<hipStream_t>(stream_));
  g_pair = nullptr;
  return rc;
}
