// sp3_attention: softmax(q k^T * scale) v for head_dim 64 on gfx950, flash-style (online softmax),
// one wave per 16 query rows, all operands streamed straight from global/L2 into MFMA registers.
//
// Layout trick (DESIGN.md §"Attention"): compute S^T = K.Q^T and O^T = V^T.P^T, i.e. keys / head-dim
// on the MFMA row axis and the 16 query rows on the MFMA column axis (lane&15).  Then
//   * a lane owns ONE query row q = lane&15 in both S^T and O^T: row max, row sum and the online
//     rescale are lane-local (plus two shuffles across the four 16-lane groups);
//   * the exponentiated probabilities a lane holds (keys 16t+4g+r, g = lane>>4) are exactly the
//     B-operand slots of the P^T operand of the second MFMA: P never leaves registers;
//   * V is consumed as V^T[d][key], which the QKV GEMM epilogue writes directly (per head, key axis
//     contiguous, zero padded), so the A operand of the second MFMA is contiguous 8/16-byte loads.
// bf16: v_mfma_f32_16x16x32_bf16; fp32: v_mfma_f32_16x16x4_f32 (exact fp32, parity mode).
#include "common.h"
#include <math.h>
#include <cstdlib>

namespace {

template <typename T> struct AttnT;

template <> struct AttnT<__bf16> {
  using Store = __bf16;
  // ---- S^T block: 16 keys x 16 queries, contraction over d = 64 in two MFMAs
  struct QReg { bf16x8 v[2]; };
  static __device__ __forceinline__ void loadQ(QReg& r, const __bf16* row, int g) {
    r.v[0] = *reinterpret_cast<const bf16x8*>(row + 8 * g);
    r.v[1] = *reinterpret_cast<const bf16x8*>(row + 32 + 8 * g);
  }
  // (the operand row is loaded and multiplied in two steps: the kernels issue the loads of a whole 64-row tile before the first MFMA)
  struct KReg { bf16x8 v[2]; };
  static __device__ __forceinline__ KReg loadK(const __bf16* krow, int g) {
    KReg k;
    k.v[0] = *reinterpret_cast<const bf16x8*>(krow + 8 * g);
    k.v[1] = *reinterpret_cast<const bf16x8*>(krow + 32 + 8 * g);
    return k;
  }
  static __device__ __forceinline__ f32x4 qk(const KReg& k, const QReg& q) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k.v[0], q.v[0], s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k.v[1], q.v[1], s, 0, 0, 0);
    return s;
  }
  static __device__ __forceinline__ f32x4 qk(const __bf16* krow, int g, const QReg& q) { return qk(loadK(krow, g), q); }
  // ---- O^T += V^T . P^T over the 64 keys of a tile: p[t][r] is key 16t+4g+r of query lane&15
  static __device__ __forceinline__ void pv(f32x4 (&o)[4], const __bf16* vt_head, int64_t vt_ld, int kbase, int g,
                                            int dl, const f32x4 (&p)[4]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bf16x8 pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = (__bf16)p[2 * u + (j >> 2)][j & 3];
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const __bf16* vr = vt_head + (int64_t)(db * 16 + dl) * vt_ld + kbase + 32 * u + 4 * g;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vr);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vr + 16);
        bf16x8 va;
        va[0] = lo[0]; va[1] = lo[1]; va[2] = lo[2]; va[3] = lo[3];
        va[4] = hi[0]; va[5] = hi[1]; va[6] = hi[2]; va[7] = hi[3];
        o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o[db], 0, 0, 0);
      }
    }
  }
};

template <> struct AttnT<float> {
  using Store = float;
  struct QReg { float4 v[4]; };
  static __device__ __forceinline__ void loadQ(QReg& r, const float* row, int g) {
    const float4* p = reinterpret_cast<const float4*>(row + 16 * g);
    r.v[0] = p[0]; r.v[1] = p[1]; r.v[2] = p[2]; r.v[3] = p[3];
  }
  struct KReg { float4 v[4]; };
  static __device__ __forceinline__ KReg loadK(const float* krow, int g) {
    const float4* p = reinterpret_cast<const float4*>(krow + 16 * g);
    KReg k;
    k.v[0] = p[0]; k.v[1] = p[1]; k.v[2] = p[2]; k.v[3] = p[3];
    return k;
  }
  static __device__ __forceinline__ f32x4 qk(const float* krow, int g, const QReg& q) { return qk(loadK(krow, g), q); }
  static __device__ __forceinline__ f32x4 qk(const KReg& k, const QReg& q) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 kv = k.v[i];
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.x, q.v[i].x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.y, q.v[i].y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.z, q.v[i].z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.w, q.v[i].w, s, 0, 0, 0);
    }
    return s;
  }
  static __device__ __forceinline__ void pv(f32x4 (&o)[4], const float* vt_head, int64_t vt_ld, int kbase, int g, int dl,
                                            const f32x4 (&p)[4]) {
    float4 vv[4][4];                                   // all 16 loads of the tile first: one round trip, not sixteen
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int db = 0; db < 4; ++db) vv[t][db] = *reinterpret_cast<const float4*>(vt_head + (int64_t)(db * 16 + dl) * vt_ld + kbase + 16 * t + 4 * g);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const float4 va = vv[t][db];
        o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.x, p[t][0], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.y, p[t][1], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.z, p[t][2], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.w, p[t][3], o[db], 0, 0, 0);
      }
    }
  }
};

// One workgroup = 16 query rows of one head; its 4 waves split the 64-key tiles (tile t goes to wave t & 3), each with its own
// online softmax; the partial (m, l, O^T) states are merged through LDS (wave w finishes d-block w) -- at N = 196 every wave
// has ONE tile, so the dependent chain is 1 tile instead of 4 (fp32 mode: 21 -> 8 us per launch).
// fp32 operands, every product through three bf16 MFMAs of a (hi, lo) split (the model's "f32x3" precision, see gemm.hip)
struct F32X3 {};
__device__ __forceinline__ void split8f(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)x[e];
    hi[e] = h;
    lo[e] = (__bf16)(x[e] - (float)h);
  }
}
__device__ __forceinline__ f32x4 mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
template <> struct AttnT<F32X3> {
  using Store = float;
  struct QReg { bf16x8 h[2], l[2]; };                       // d = 16g .. 16g+15 of the head, split once per workgroup
  static __device__ __forceinline__ void loadQ(QReg& r, const float* row, int g) {
    const float4* p = reinterpret_cast<const float4*>(row + 16 * g);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 a = p[2 * u], b = p[2 * u + 1];
      const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      split8f(x, r.h[u], r.l[u]);
    }
  }
  struct KReg { float4 v[4]; };
  static __device__ __forceinline__ KReg loadK(const float* krow, int g) {
    const float4* p = reinterpret_cast<const float4*>(krow + 16 * g);
    KReg k;
    k.v[0] = p[0]; k.v[1] = p[1]; k.v[2] = p[2]; k.v[3] = p[3];
    return k;
  }
  static __device__ __forceinline__ f32x4 qk(const float* krow, int g, const QReg& q) { return qk(loadK(krow, g), q); }
  static __device__ __forceinline__ f32x4 qk(const KReg& k, const QReg& q) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 a = k.v[2 * u], b = k.v[2 * u + 1];
      const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      bf16x8 kh, kl;
      split8f(x, kh, kl);
      s = mfma3(kh, kl, q.h[u], q.l[u], s);
    }
    return s;
  }
  // O^T += V^T . P^T over the 64 keys of a tile, 32 keys per MFMA: lane g supplies keys 32u + 4g + (0..3) and 32u + 16 + 4g + (0..3)
  static __device__ __forceinline__ void pv(f32x4 (&o)[4], const float* vt_head, int64_t vt_ld, int kbase, int g, int dl,
                                            const f32x4 (&p)[4]) {
    float4 vlo[2][4], vhi[2][4];                       // all 16 loads of the tile first: one round trip, not eight
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const float* vr = vt_head + (int64_t)(db * 16 + dl) * vt_ld + kbase + 32 * u + 4 * g;
        vlo[u][db] = *reinterpret_cast<const float4*>(vr);
        vhi[u][db] = *reinterpret_cast<const float4*>(vr + 16);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float px[8] = {p[2 * u][0], p[2 * u][1], p[2 * u][2], p[2 * u][3], p[2 * u + 1][0], p[2 * u + 1][1], p[2 * u + 1][2], p[2 * u + 1][3]};
      bf16x8 ph, pl;
      split8f(px, ph, pl);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const float4 lo4 = vlo[u][db], hi4 = vhi[u][db];
        const float vx[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
        bf16x8 vh, vl;
        split8f(vx, vh, vl);
        o[db] = mfma3(vh, vl, ph, pl, o[db]);
      }
    }
  }
};

// fp32 operands, fp16 two-way split x = h + l * 2^-11 (22 operand bits; the model's "f16x3" precision, see gemm.hip): three fp16 MFMAs per
// product, exact in fp32; the cross terms are summed separately and folded in with 2^-11
struct F16X3 {};
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ void split8h(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)x[e];
    hi[e] = h;
    lo[e] = (_Float16)((x[e] - (float)h) * 2048.0f);
  }
}
__device__ __forceinline__ f32x4 mfma3h(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x4 c) {
  f32x4 x = {0.f, 0.f, 0.f, 0.f};
  x = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, x, 0, 0, 0);
  x = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, x, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = fmaf(x[r], 1.0f / 2048.0f, c[r]);
  return c;
}
template <> struct AttnT<F16X3> {
  using Store = float;
  struct QReg { f16x8 h[2], l[2]; };
  static __device__ __forceinline__ void loadQ(QReg& r, const float* row, int g) {
    const float4* p = reinterpret_cast<const float4*>(row + 16 * g);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 a = p[2 * u], b = p[2 * u + 1];
      const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      split8h(x, r.h[u], r.l[u]);
    }
  }
  struct KReg { float4 v[4]; };
  static __device__ __forceinline__ KReg loadK(const float* krow, int g) {
    const float4* p = reinterpret_cast<const float4*>(krow + 16 * g);
    KReg k;
    k.v[0] = p[0]; k.v[1] = p[1]; k.v[2] = p[2]; k.v[3] = p[3];
    return k;
  }
  static __device__ __forceinline__ f32x4 qk(const float* krow, int g, const QReg& q) { return qk(loadK(krow, g), q); }
  static __device__ __forceinline__ f32x4 qk(const KReg& k, const QReg& q) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 a = k.v[2 * u], b = k.v[2 * u + 1];
      const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      f16x8 kh, kl;
      split8h(x, kh, kl);
      s = mfma3h(kh, kl, q.h[u], q.l[u], s);
    }
    return s;
  }
  static __device__ __forceinline__ void pv(f32x4 (&o)[4], const float* vt_head, int64_t vt_ld, int kbase, int g, int dl,
                                            const f32x4 (&p)[4]) {
    float4 vlo[2][4], vhi[2][4];                       // all 16 loads of the tile first: one round trip, not eight
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const float* vr = vt_head + (int64_t)(db * 16 + dl) * vt_ld + kbase + 32 * u + 4 * g;
        vlo[u][db] = *reinterpret_cast<const float4*>(vr);
        vhi[u][db] = *reinterpret_cast<const float4*>(vr + 16);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float px[8] = {p[2 * u][0], p[2 * u][1], p[2 * u][2], p[2 * u][3], p[2 * u + 1][0], p[2 * u + 1][1], p[2 * u + 1][2], p[2 * u + 1][3]};
      f16x8 ph, pl;
      split8h(px, ph, pl);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const float4 lo4 = vlo[u][db], hi4 = vhi[u][db];
        const float vx[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
        f16x8 vh, vl;
        split8h(vx, vh, vl);
        o[db] = mfma3h(vh, vl, ph, pl, o[db]);
      }
    }
  }
};

// fp32 operands rounded to bf16 on the way into ONE bf16 MFMA per product (training in "bf16" precision: what bf16 autocast computes)
struct BF16X1 {};
template <> struct AttnT<BF16X1> {
  using Store = float;
  struct QReg { bf16x8 v[2]; };
  static __device__ __forceinline__ void loadQ(QReg& r, const float* row, int g) {
    const float4* p = reinterpret_cast<const float4*>(row + 16 * g);
    r.v[0] = cvt8(p[0], p[1]);
    r.v[1] = cvt8(p[2], p[3]);
  }
  struct KReg { float4 v[4]; };
  static __device__ __forceinline__ KReg loadK(const float* krow, int g) {
    const float4* p = reinterpret_cast<const float4*>(krow + 16 * g);
    KReg k;
    k.v[0] = p[0]; k.v[1] = p[1]; k.v[2] = p[2]; k.v[3] = p[3];
    return k;
  }
  static __device__ __forceinline__ f32x4 qk(const float* krow, int g, const QReg& q) { return qk(loadK(krow, g), q); }
  static __device__ __forceinline__ f32x4 qk(const KReg& k, const QReg& q) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cvt8(k.v[0], k.v[1]), q.v[0], s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cvt8(k.v[2], k.v[3]), q.v[1], s, 0, 0, 0);
    return s;
  }
  static __device__ __forceinline__ void pv(f32x4 (&o)[4], const float* vt_head, int64_t vt_ld, int kbase, int g, int dl,
                                            const f32x4 (&p)[4]) {
    float4 vlo[2][4], vhi[2][4];                       // all 16 loads of the tile first: one round trip, not eight
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const float* vr = vt_head + (int64_t)(db * 16 + dl) * vt_ld + kbase + 32 * u + 4 * g;
        vlo[u][db] = *reinterpret_cast<const float4*>(vr);
        vhi[u][db] = *reinterpret_cast<const float4*>(vr + 16);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bf16x8 pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = (__bf16)p[2 * u + (j >> 2)][j & 3];
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cvt8(vlo[u][db], vhi[u][db]), pb, o[db], 0, 0, 0);
    }
  }
};

// Workgroup -> (query tile, head, batch).  xcd_map = heads x B (0: the plain (query tiles, heads, B) grid).  Workgroup L of the dispatch
// order runs on XCD L % 8: hand every XCD WHOLE (head, batch) pairs {x, x + 8, ..} with all their query tiles, so a pair's K / V^T are
// fetched by one L2 instead of by all eight (the plain grid spreads the query tiles of a pair over the XCDs).  The grid is then (query
// tiles, pairs rounded up to a multiple of 8); the surplus pairs exit (returns false).  Measured with cold operands
// (tools/bench_attention_long.py, packed bf16 kernel): 2 x 12 heads x 196 tokens 5.09 -> 3.85 us, 10 x 16 x 196 14.97 -> 11.04,
// 2 x 16 x 1024 23.3 -> 21.8, 16 x 16 x 1024 142 -> 131 us.
__device__ __forceinline__ bool attn_tile_of(int xcd_map, int heads, int& qt, int& h, int& b) {
  qt = blockIdx.x; h = blockIdx.y; b = blockIdx.z;
  if (xcd_map) {
    const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, xc = L & 7, jj = L >> 3, pq = jj / gridDim.x;
    const unsigned pair = xc + 8 * pq;
    if (pair >= (unsigned)xcd_map) return false;
    qt = (int)(jj - pq * gridDim.x);
    b = (int)(pair / (unsigned)heads);
    h = (int)(pair - (unsigned)b * heads);
  }
  return true;
}
// the XCD map needs its (head, batch) pairs, padded to a multiple of 8, in grid.y (<= 65535); beyond that the plain (q tiles, heads, B)
// grid serves (heads and B are checked against 65535 separately)
static bool attn_xcd_ok(int heads, int B) { return ((long)heads * B + 7) / 8 * 8 <= 65535; }

template <typename TT>
__global__ __launch_bounds__(256) void attention_kernel(const typename AttnT<TT>::Store* __restrict__ Q, int64_t sq, int64_t ldq,
                                                       const typename AttnT<TT>::Store* __restrict__ K, int64_t sk, int64_t ldk,
                                                       const typename AttnT<TT>::Store* __restrict__ VT, int64_t vt_ld, void* __restrict__ O,
                                                       int64_t ldo, int out_bf16, int out_packed, int heads, int Nq, int Nk,
                                                       float scale, float* __restrict__ lse, int xcd_map) {
  using A = AttnT<TT>;
  using T = typename A::Store;
  __shared__ float sh_o[4][4][64][4];   // [wave][db][lane][r]
  __shared__ float sh_m[4][64], sh_l[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  int qt, h, b;
  if (!attn_tile_of(xcd_map, heads, qt, h, b)) return;
  const int q0 = qt * 16;
  int qrow = q0 + ql;
  qrow = qrow < Nq ? qrow : Nq - 1;
  typename A::QReg qreg;
  A::loadQ(qreg, Q + (int64_t)b * sq + (int64_t)qrow * ldq + h * 64, g);
  const T* kbase_ptr = K + (int64_t)b * sk + h * 64;
  const T* vt_head = VT + (int64_t)(b * heads + h) * 64 * vt_ld;

  f32x4 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  for (int kb = wave * 64; kb < Nk; kb += 256) {
    f32x4 s[4];
    typename A::KReg kr[4];                      // the tile's four key blocks are requested before the first product
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int krow = kb + 16 * t + ql;               // A-operand row = key
      krow = krow < Nk ? krow : Nk - 1;
      kr[t] = A::loadK(kbase_ptr + (int64_t)krow * ldk, g);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = A::qk(kr[t], qreg);
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb + 16 * t + 4 * g + r;
        const float v = key < Nk ? s[t][r] * scale : -INFINITY;
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);          // finite: every tile has >= 1 valid key
    const float alpha = expf(m_run - m_new);       // exp(-inf) = 0 on the first tile
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(s[t][r] - m_new);
        s[t][r] = e;
        ps += e;
      }
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[db][r] *= alpha;
    A::pv(o, vt_head, vt_ld, kb, g, ql, s);
  }
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  // ---- merge the four per-wave states (running max m, sum l, O^T[d = 16*db + 4*g + r][q = lane&15])
  sh_m[wave][lane] = m_run;
  sh_l[wave][lane] = l_run;
#pragma unroll
  for (int db = 0; db < 4; ++db)
    *reinterpret_cast<float4*>(&sh_o[wave][db][lane][0]) = make_float4(o[db][0], o[db][1], o[db][2], o[db][3]);
  __syncthreads();
  float M = sh_m[0][lane];
#pragma unroll
  for (int w = 1; w < 4; ++w) M = fmaxf(M, sh_m[w][lane]);
  float L = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int db = wave;                       // this wave finishes d-block `wave`
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float sc = expf(sh_m[w][lane] - M);             // 0 for a wave that saw no tile (m = -inf)
    L += sh_l[w][lane] * sc;
    const float4 ow = *reinterpret_cast<const float4*>(&sh_o[w][db][lane][0]);
    acc.x += ow.x * sc; acc.y += ow.y * sc; acc.z += ow.z * sc; acc.w += ow.w * sc;
  }
  const float inv = 1.0f / L;
  // training: the backward recomputes P = exp(s - M) / L.  (M, 1 / L) are kept apart: folded into one log-sum-exp, the fp32 rounding of a
  // large M would rescale the whole row of recomputed probabilities by 1 +- |M| * 2^-24 -- an error the forward never made.
  if (lse && wave == 0 && g == 0 && q0 + ql < Nq)
    *reinterpret_cast<float2*>(lse + 2 * ((int64_t)(b * heads + h) * Nq + q0 + ql)) = make_float2(M, inv);
  if (q0 + ql < Nq) {
    const int row = b * Nq + q0 + ql;
    const int col = h * 64 + db * 16 + 4 * g;
    const int64_t off = out_packed ? packed_off(row, col, heads * 64, out_bf16 != 0) : (int64_t)row * ldo + col;
    if (out_bf16) {
      bf16x4 ob;
      ob[0] = (__bf16)(acc.x * inv); ob[1] = (__bf16)(acc.y * inv); ob[2] = (__bf16)(acc.z * inv); ob[3] = (__bf16)(acc.w * inv);
      st_out(reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(O) + off), ob);
    } else {
      st_out(reinterpret_cast<float4*>(reinterpret_cast<float*>(O) + off), make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
    }
  }
}


// ------------------------------------------------------------------ bf16, fragment-order operands, prefetching
struct KFrag { bf16x8 v[2]; };

__device__ __forceinline__ KFrag load_frag(const __bf16* base, int row, int col, int K) {
  // lane (g, r) of a 16-row block reads its 16 elements (d = 16g .. 16g+15 of the head at `col`): two 16-byte halves,
  // one in each 1 KB piece of the fragment block
  const __bf16* p = base + packed_off(row, col, K, true);
  KFrag f;
  f.v[0] = *reinterpret_cast<const bf16x8*>(p);
  f.v[1] = *reinterpret_cast<const bf16x8*>(p + 64 * 8);
  return f;
}

// ---- the cross-attention q projection INSIDE the attention launch (round 6; croco/models/blocks.py:149-169: q = projq(norm2(x)))
// A decoder layer's cross-attention reads q = RoPE(LN(x) Wq^T + b) of its own side; as a launch of its own that 196 x 768 x 768 GEMM
// is 6.7 us in front of a 5 us attention launch.  Here every attention workgroup (16 query rows x one head) computes its own
// [16 x 64] q tile first: K = 768 over the 4 waves exactly as sm_kernel<2, 2, 4, 12> splits it (wave w owns k-blocks w, w + 4, w + 8,
// every operand byte requested up front), partial tiles meet in LDS, then folded LayerNorm (row statistics of the producer's
// epilogue), bias and 2-D RoPE with the partner column in lane ^ 16 -- the arithmetic of the SM_ROPE epilogue, operation for
// operation -- straight into the MFMA fragment the attention loop wants.  Costs 96 KB of L2-resident weight reads per workgroup.
struct QProjArgs {
  const __bf16* X; int64_t x_gs;                     // fragment-order [rows][D] per group (decoder side), elements between groups
  const float* st; int64_t st_gs;                    // LayerNorm partials [rows][D / 32][2], floats between groups
  const __bf16* W; int64_t w_gs;                     // fragment-order [D][D]
  const float* ln_s; const float* bias; int64_t v_gs;
  const int* pos; const float* cos; const float* sin;
  float eps;
  int group_imgs, img_tokens;
};

template <int NKB>
__device__ __forceinline__ KFrag attn_qproj(const QProjArgs& qa, float* sh, int q0, int Nq, int h, int b, int lane, int wave, int g, int ql) {
  constexpr int D = NKB * 64, NL4 = NKB / 4, KW = NKB / 4, LD = 68;
  static_assert(NKB % 4 == 0 && NKB <= 16, "K over the 4 waves; LayerNorm partials: NKB / 4 float4 per lane of a 4-lane row team");
  const int grp = b / qa.group_imgs, img = b - grp * qa.group_imgs;
  int tok = q0 + ql;
  tok = tok < Nq ? tok : Nq - 1;                    // query rows past the image: re-read the last one (their results are dropped)
  const int xrow = img * qa.img_tokens + tok;
  const __bf16* X = qa.X + grp * qa.x_gs;
  const __bf16* W = qa.W + grp * qa.w_gs;
  // epilogue operands first: LayerNorm partials of this lane's row (the 4 lanes g of a row share them), column sums / bias / RoPE table
  // rows of this lane's 16 head dimensions d = 16 g .. 16 g + 15
  const float4* ps = reinterpret_cast<const float4*>(qa.st + grp * qa.st_gs + (int64_t)xrow * (2 * NKB) * 2);
  float4 lp[NL4];
#pragma unroll
  for (int q = 0; q < NL4; ++q) lp[q] = ps[g + 4 * q];
  const float* sv = qa.ln_s + grp * qa.v_gs + h * 64 + 16 * g;
  const float* bv = qa.bias + grp * qa.v_gs + h * 64 + 16 * g;
  const int p = qa.pos[(int64_t)xrow * 2 + (g >> 1)];                       // axis: y for d < 32, x above
  float4 s4[4], b4[4], c4[4], n4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s4[j] = *reinterpret_cast<const float4*>(sv + 4 * j);
    b4[j] = *reinterpret_cast<const float4*>(bv + 4 * j);
    c4[j] = *reinterpret_cast<const float4*>(qa.cos + p * 16 + 4 * j);
    n4[j] = *reinterpret_cast<const float4*>(qa.sin + p * 16 + 4 * j);
  }
  KFrag xf[KW], wf[KW][4];
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int kb = wave + 4 * i;
    xf[i] = load_frag(X, xrow, kb * 64 + 16 * g, D);
#pragma unroll
    for (int t = 0; t < 4; ++t) wf[i][t] = load_frag(W, h * 64 + 16 * t + ql, kb * 64 + 16 * g, D);
  }
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KW; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i].v[0], wf[i][t].v[0], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i].v[1], wf[i][t].v[1], acc[t], 0, 0, 0);
    }
  // partial tiles -> LDS (C layout: column = lane & 15 = head dimension 16 t + ql, row = 4 (lane >> 4) + reg = token)
  {
    float* slab = sh + wave * 16 * LD;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(4 * g + r) * LD + 16 * t + ql] = acc[t][r];
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < NL4; ++q) { s1 += lp[q].x + lp[q].z; s2 += lp[q].y + lp[q].w; }
  s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
  s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
  const float mean = s1 * (1.0f / (float)D);
  const float var = fmaxf(s2 * (1.0f / (float)D) - mean * mean, 0.f);
  const float rstd = 1.0f / sqrtf(var + qa.eps), rm = rstd * mean;
  __syncthreads();
  float v[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 t4 = *reinterpret_cast<const float4*>(sh + ql * LD + 16 * g + 4 * j);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 u = *reinterpret_cast<const float4*>(sh + w * 16 * LD + ql * LD + 16 * g + 4 * j);
      t4.x += u.x; t4.y += u.y; t4.z += u.z; t4.w += u.w;
    }
    v[4 * j] = rstd * t4.x - rm * s4[j].x; v[4 * j + 1] = rstd * t4.y - rm * s4[j].y;
    v[4 * j + 2] = rstd * t4.z - rm * s4[j].z; v[4 * j + 3] = rstd * t4.w - rm * s4[j].w;
    v[4 * j] += b4[j].x; v[4 * j + 1] += b4[j].y; v[4 * j + 2] += b4[j].z; v[4 * j + 3] += b4[j].w;
  }
  const bool second = (g & 1) != 0;                                        // dimension d & 16: the y of its (x, y) pair
  KFrag qf;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float pv = __shfl_xor(v[j], 16);                                 // partner dimension d ^ 16: lane g ^ 1 of the same row
    const float4 cq = c4[j >> 2], sq = n4[j >> 2];
    const float cs = (j & 3) == 0 ? cq.x : (j & 3) == 1 ? cq.y : (j & 3) == 2 ? cq.z : cq.w;
    const float sn = (j & 3) == 0 ? sq.x : (j & 3) == 1 ? sq.y : (j & 3) == 2 ? sq.z : sq.w;
    const float o = second ? (v[j] * cs + pv * sn) : (v[j] * cs - pv * sn);
    qf.v[j >> 3][j & 7] = (__bf16)o;
  }
  __syncthreads();                                   // the partial tiles' LDS is the merge scratch of the attention loop below
  return qf;
}

// One workgroup = QB blocks of 16 query rows of one head; its 4 waves split the key tiles (tile t goes to wave t & 3), each running
// an independent online softmax per query block; the four partial (m, l, O^T) states are merged through LDS (wave w merges d-block w).
// At N = 196 (4 tiles) every wave handles ONE tile: the dependent chain is 1 tile instead of 4.
//
// Round 5 (profiles/r05_attn_long_pmc.md, r05_attn_long_variants.txt): the round-4 loop was VALU-bound -- 385 VALU instructions per 64-key
// tile next to 16 MFMAs (libm-style exp2f with its denormal fix-up, a key mask on every tile, 64-bit fragment addressing, a 32-register
// copy of the prefetched K fragments, 16-deep max / sum chains), 48 % of the wave cycles in issue stalls at 4 waves per SIMD.  Now:
//   * QB = 2 query blocks per wave from 512 query rows on: the K / V^T fragments of a tile are loaded once for both (half the vector-
//     memory traffic per query) and the two softmax chains are independent work the compiler interleaves under each other's latencies;
//   * bare v_exp_f32 (probabilities below 2^-126 flush to zero), the scale folded into the exponent's FMA (max on the raw scores),
//     the key mask on the ragged last tile only, tree-shaped reductions (four chains of four);
//   * one pointer per lane with constant strides per tile; K and V^T of a tile requested together at its top (no register copy: the
//     other resident waves cover the latency -- the explicit prefetch cost more registers than it hid).
// 2 x 16 heads x 1024 tokens: 28.4 -> 21.5 us; 2 x 12 x 1024: 22.3 -> 16.7; 2 x 12 x 196: 4.26 -> 3.93.  Against the round-4 kernel the
// outputs differ by <= 2.5e-4 of their maximum (the last bit of a probability before its bf16 rounding).
template <int QB, int MINW, int QNKB = 0>
__global__ __launch_bounds__(256, MINW) void attention_packed_kernel(const __bf16* __restrict__ QP, int q_cols, int q_col0, int npad_q,
                                                                     const __bf16* __restrict__ KP, int k_cols, int k_col0, int npad_k,
                                                                     const __bf16* __restrict__ VTP, void* __restrict__ O, int64_t ldo,
                                                                     int out_bf16, int out_packed, int heads, int Nq, int Nk, float scale,
                                                                     int o_group, int o_group_rows, int xcd_map, const QProjArgs qa) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* sh_o = sh;                                  // [wave][qb][db][lane][4]
  float* sh_m = sh + 4 * QB * 4 * 64 * 4;            // [wave][qb][lane]
  float* sh_l = sh_m + 4 * QB * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  int qt, h, b;
  if (!attn_tile_of(xcd_map, heads, qt, h, b)) return;
  const int q0 = qt * 16 * QB;
  const int last_qrow = npad_q - 16;                 // query blocks past the padded rows re-read the last block (their results are dropped)
  KFrag qf[QB];
  if constexpr (QNKB == 0) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int r0 = q0 + 16 * qb < last_qrow ? q0 + 16 * qb : last_qrow;
      qf[qb] = load_frag(QP, b * npad_q + r0 + ql, q_col0 + h * 64 + 16 * g, q_cols);
    }
  }
  const int ntiles = (Nk + 63) >> 6;
  const float sl2 = scale * 1.4426950408889634f;     // softmax in base 2: exp(x) = exp2(x * log2 e), v_exp_f32
  // K fragment of key block t of a tile: rows krow0 + 64 tile + 16 t + ql; a 16-row block's column blocks are contiguous, so one
  // pointer per lane and a constant stride per 16-row block
  const __bf16* kbase = KP + packed_off(b * npad_k + ql, k_col0 + h * 64 + 16 * g, k_cols, true);
  const int64_t kblk = (int64_t)((k_cols + 63) >> 6) * 1024;        // elements per 16-row block
  const int64_t nU = npad_k >> 5;
  const __bf16* vbase = VTP + ((int64_t)(b * heads + h) * nU * 4 * 64 + lane) * 8;

  f32x4 o[QB][4];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY; l_run[qb] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[qb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  KFrag kc[4];
  bf16x8 vv[2][4];
  auto load_tile = [&](int tile) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const __bf16* p = kbase + (int64_t)(4 * tile + t) * kblk;
      kc[t].v[0] = *reinterpret_cast<const bf16x8*>(p);
      kc[t].v[1] = *reinterpret_cast<const bf16x8*>(p + 64 * 8);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int db = 0; db < 4; ++db)
        vv[u][db] = *reinterpret_cast<const bf16x8*>(vbase + (((int64_t)(2 * tile + u)) * 4 + db) * 64 * 8);
  };
  if constexpr (QNKB > 0) {
    // the q projection in front of the loop: this wave's first key tile is requested BEFORE it (independent of q), so the K / V^T
    // round trip and the projection's operand round trip overlap
    static_assert(QB == 1, "the fused q projection serves one query block per workgroup");
    load_tile(wave < ntiles ? wave : ntiles - 1);
    qf[0] = attn_qproj<QNKB>(qa, sh, q0, Nq, h, b, lane, wave, g, ql);
  }
  if (wave < ntiles) {
    for (int tile = wave; tile < ntiles; tile += 4) {
      const int kb = tile << 6;
      if (QNKB == 0 || tile != wave) load_tile(tile);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        f32x4 s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[t].v[0], qf[qb].v[0], s[t], 0, 0, 0);
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[t].v[1], qf[qb].v[1], s[t], 0, 0, 0);
        }
        if (kb + 64 > Nk) {                    // the ragged last tile (wave-uniform): keys past Nk drop out of the softmax
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kb + 16 * t + 4 * g + r >= Nk) s[t][r] = -INFINITY;
        }
        float mt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) mt[t] = fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3]));
        float mx = fmaxf(fmaxf(mt[0], mt[1]), fmaxf(mt[2], mt[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run[qb], mx * sl2);          // (scale > 0: the max of the raw scores is the max of the scaled ones)
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        float pt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) s[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], sl2, -m_new));
          pt[t] = (s[t][0] + s[t][1]) + (s[t][2] + s[t][3]);
        }
        l_run[qb] = l_run[qb] * alpha + ((pt[0] + pt[1]) + (pt[2] + pt[3]));
        m_run[qb] = m_new;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[qb][db][r] *= alpha;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          bf16x8 pb;
#pragma unroll
          for (int j = 0; j < 8; ++j) pb[j] = (__bf16)s[2 * u + (j >> 2)][j & 3];
#pragma unroll
          for (int db = 0; db < 4; ++db) o[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vv[u][db], pb, o[qb][db], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      l_run[qb] += __shfl_xor(l_run[qb], 16);
      l_run[qb] += __shfl_xor(l_run[qb], 32);
    }
  }
  // ---- merge the four per-wave states of every query block (base-2 running max m, sum l, O^T): wave w merges d-block w
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    sh_m[(wave * QB + qb) * 64 + lane] = m_run[qb];
    sh_l[(wave * QB + qb) * 64 + lane] = l_run[qb];
#pragma unroll
    for (int db = 0; db < 4; ++db)
      *reinterpret_cast<float4*>(sh_o + ((((wave * QB + qb) * 4 + db) * 64 + lane) << 2)) = make_float4(o[qb][db][0], o[qb][db][1], o[qb][db][2], o[qb][db][3]);
  }
  __syncthreads();
  const int db = wave;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float M = sh_m[(0 * QB + qb) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = fmaxf(M, sh_m[(w * QB + qb) * 64 + lane]);
    float L = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float sc = __builtin_amdgcn_exp2f(sh_m[(w * QB + qb) * 64 + lane] - M);     // 0 for a wave that saw no tile (m = -inf)
      L += sh_l[(w * QB + qb) * 64 + lane] * sc;
      const float4 ow = *reinterpret_cast<const float4*>(sh_o + ((((w * QB + qb) * 4 + db) * 64 + lane) << 2));
      acc.x += ow.x * sc; acc.y += ow.y * sc; acc.z += ow.z * sc; acc.w += ow.w * sc;
    }
    const float inv = 1.0f / L;
    const int q = q0 + 16 * qb + ql;
    if (q < Nq) {
      // output row: images are consecutive, except that every o_group of them may start at a multiple of o_group_rows
      // (grouped launches keep each problem's packed rows 16-aligned)
      const int row = o_group > 0 ? (b / o_group) * o_group_rows + (b % o_group) * Nq + q : b * Nq + q;
      const int col = h * 64 + db * 16 + 4 * g;
      const int64_t off = out_packed ? packed_off(row, col, heads * 64, out_bf16 != 0) : (int64_t)row * ldo + col;
      if (out_bf16) {
        bf16x4 ob;
        ob[0] = (__bf16)(acc.x * inv); ob[1] = (__bf16)(acc.y * inv); ob[2] = (__bf16)(acc.z * inv); ob[3] = (__bf16)(acc.w * inv);
        st_out(reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(O) + off), ob);
      } else {
        st_out(reinterpret_cast<float4*>(reinterpret_cast<float*>(O) + off), make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
      }
    }
  }
}


// ------------------------------------------------------------------ training: the backward of softmax(q k^T * scale) v
// Same register layout as the forward (a lane owns one row of the 16-row block its workgroup works on; the other axis is streamed
// in 64-row tiles, one tile per wave per round, partial sums merged through LDS), probabilities recomputed from the saved
// softmax statistics:   P = exp(S - max) / sum,  dP = dO V^T,  dS = P (.) (dP - D) * scale,  D = rowsum(dO (.) O),
//   dq = dS k   (workgroup = 16 queries; streams k, v rows and k^T),
//   dk = dS^T q, dv = P^T dO   (workgroup = 16 keys; streams q, dO rows and q^T, dO^T).
// Every product is one of the forward's two MFMA forms (AttnT::qk: rows x registers contracted over the head dimension;
// AttnT::pv: transposed stream x register tile contracted over the streamed axis), so the precision modes are the forward's.
struct AttnBwdArgs {
  const float *q, *k, *v, *dout;               // [B][N][..]: element (b, n, h, d) at p + b*s + n*ld + h*64 + d
  int64_t sq, ldq, sk, ldk, sv, ldv, sdo, lddo;
  const float *qT, *kT, *doT;                  // [B*heads][64][ldT]: per-head transposes, zero padded to a multiple of 64 rows of the other axis
  int64_t ldTq, ldTk;
  const float* lse;                            // [B*heads][Nq][2]: (row max, 1 / row sum) of the forward's softmax
  float* D;                                    // [B*heads][Nq] (written by the dq kernel, read by the dk / dv kernel)
  float *dq, *dk, *dv;
  int64_t sdq, lddq, sdk, lddk, sdv, lddv;
  int heads, Nq, Nk;
  float scale;
};

template <typename TT>
__global__ __launch_bounds__(256) void attention_bwd_dq_kernel(AttnBwdArgs a) {
  using A = AttnT<TT>;
  __shared__ float sh_o[4][4][64][4];
  __shared__ double sh_d[4][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * 16;
  const int Nq = a.Nq, Nk = a.Nk;
  const int qrow = min(q0 + ql, Nq - 1);
  const int64_t bh = (int64_t)b * a.heads + h;
  typename A::QReg qreg, doreg;
  A::loadQ(qreg, a.q + (int64_t)b * a.sq + (int64_t)qrow * a.ldq + h * 64, g);
  A::loadQ(doreg, a.dout + (int64_t)b * a.sdo + (int64_t)qrow * a.lddo + h * 64, g);
  const float2 lq = *reinterpret_cast<const float2*>(a.lse + 2 * (bh * Nq + qrow));          // (row max, 1 / row sum) of the forward
  const float* kbase = a.k + (int64_t)b * a.sk + h * 64;
  const float* vbase = a.v + (int64_t)b * a.sv + h * 64;
  const float* kT = a.kT + bh * 64 * a.ldTk;
  // (p, dp) of a 64-key tile: p[t][r] = P[query ql][key kb + 16t + 4g + r], dp likewise
  auto tile = [&](int kb, f32x4 (&p)[4], f32x4 (&dp)[4]) {
    typename A::KReg kr[4], vr[4];               // 32 independent 16-byte loads in flight before the first product
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int krow = min(kb + 16 * t + ql, Nk - 1);
      kr[t] = A::loadK(kbase + (int64_t)krow * a.ldk, g);
      vr[t] = A::loadK(vbase + (int64_t)krow * a.ldv, g);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 s = A::qk(kr[t], qreg);
      dp[t] = A::qk(vr[t], doreg);
#pragma unroll
      for (int r = 0; r < 4; ++r) p[t][r] = (kb + 16 * t + 4 * g + r) < Nk ? expf(s[r] * a.scale - lq.x) * lq.y : 0.f;
    }
  };
  // pass 1: D = sum_j P_j dP_j from the SAME rounded (P, dP) the second pass uses -- the rows of dS then sum to zero exactly as in
  // the textbook softmax backward (rowsum(dO (.) O) is the same number only in exact arithmetic; in fp32 the difference leaks a
  // common-mode term into every layer's dq and compounds with depth).  The wave's first tile stays in registers for pass 2.
  f32x4 p0[4], dp0[4];
  double dsum = 0.0;
  for (int kb = wave * 64; kb < Nk; kb += 256) {
    f32x4 p[4], dp[4];
    tile(kb, p, dp);
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) part += p[t][r] * dp[t][r];
    dsum += (double)part;
    if (kb == wave * 64) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { p0[t] = p[t]; dp0[t] = dp[t]; }
    }
  }
  dsum += __shfl_xor(dsum, 16);
  dsum += __shfl_xor(dsum, 32);
  if (g == 0) sh_d[wave][ql] = dsum;
  __syncthreads();
  const float Dq = (float)((sh_d[0][ql] + sh_d[1][ql]) + (sh_d[2][ql] + sh_d[3][ql]));
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kb = wave * 64; kb < Nk; kb += 256) {
    f32x4 p[4], dp[4], ds[4];
    if (kb == wave * 64) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { p[t] = p0[t]; dp[t] = dp0[t]; }
    } else {
      tile(kb, p, dp);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[t][r] = p[t][r] * (dp[t][r] - Dq) * a.scale;
    A::pv(acc, kT, a.ldTk, kb, g, ql, ds);
  }
#pragma unroll
  for (int db = 0; db < 4; ++db)
    *reinterpret_cast<float4*>(&sh_o[wave][db][lane][0]) = make_float4(acc[db][0], acc[db][1], acc[db][2], acc[db][3]);
  __syncthreads();
  const int db = wave;
  float4 sum = *reinterpret_cast<const float4*>(&sh_o[0][db][lane][0]);
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float4 x = *reinterpret_cast<const float4*>(&sh_o[w][db][lane][0]);
    sum.x += x.x; sum.y += x.y; sum.z += x.z; sum.w += x.w;
  }
  if (q0 + ql < Nq) {
    *reinterpret_cast<float4*>(a.dq + (int64_t)b * a.sdq + (int64_t)(q0 + ql) * a.lddq + h * 64 + db * 16 + 4 * g) = sum;
    if (wave == 0 && g == 0) a.D[bh * Nq + q0 + ql] = Dq;
  }
}

template <typename TT>
__global__ __launch_bounds__(256) void attention_bwd_dkv_kernel(AttnBwdArgs a) {
  using A = AttnT<TT>;
  __shared__ float sh_k[4][4][64][4], sh_v[4][4][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, kl = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z, k0 = blockIdx.x * 16;
  const int Nq = a.Nq, Nk = a.Nk;
  const int krow = min(k0 + kl, Nk - 1);
  const int64_t bh = (int64_t)b * a.heads + h;
  typename A::QReg kreg, vreg;
  A::loadQ(kreg, a.k + (int64_t)b * a.sk + (int64_t)krow * a.ldk + h * 64, g);
  A::loadQ(vreg, a.v + (int64_t)b * a.sv + (int64_t)krow * a.ldv + h * 64, g);
  const float* qbase = a.q + (int64_t)b * a.sq + h * 64;
  const float* dobase = a.dout + (int64_t)b * a.sdo + h * 64;
  const float* qT = a.qT + bh * 64 * a.ldTq;
  const float* doT = a.doT + bh * 64 * a.ldTq;
  const float2* lse = reinterpret_cast<const float2*>(a.lse) + bh * Nq;
  const float* Dv = a.D + bh * Nq;
  f32x4 acck[4], accv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { acck[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int qb = wave * 64; qb < Nq; qb += 256) {
    f32x4 pt[4], ds[4];
    typename A::KReg qr[4], dr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int qrow = min(qb + 16 * t + kl, Nq - 1);
      qr[t] = A::loadK(qbase + (int64_t)qrow * a.ldq, g);
      dr[t] = A::loadK(dobase + (int64_t)qrow * a.lddo, g);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 s = A::qk(qr[t], kreg);          // [query 16t+4g+r][key kl]
      const f32x4 dp = A::qk(dr[t], vreg);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = qb + 16 * t + 4 * g + r;
        const int qc = min(qi, Nq - 1);
        const float2 st = lse[qc];
        const float p = qi < Nq ? expf(s[r] * a.scale - st.x) * st.y : 0.f;
        pt[t][r] = p;
        ds[t][r] = p * (dp[r] - Dv[qc]) * a.scale;
      }
    }
    A::pv(accv, doT, a.ldTq, qb, g, kl, pt);       // dv^T[d][key] += dO^T[d][query] P[query][key]
    A::pv(acck, qT, a.ldTq, qb, g, kl, ds);        // dk^T[d][key] += q^T[d][query] dS[query][key]
  }
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    *reinterpret_cast<float4*>(&sh_k[wave][db][lane][0]) = make_float4(acck[db][0], acck[db][1], acck[db][2], acck[db][3]);
    *reinterpret_cast<float4*>(&sh_v[wave][db][lane][0]) = make_float4(accv[db][0], accv[db][1], accv[db][2], accv[db][3]);
  }
  __syncthreads();
  const int db = wave;
  float4 sk_ = *reinterpret_cast<const float4*>(&sh_k[0][db][lane][0]), sv_ = *reinterpret_cast<const float4*>(&sh_v[0][db][lane][0]);
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float4 x = *reinterpret_cast<const float4*>(&sh_k[w][db][lane][0]), y = *reinterpret_cast<const float4*>(&sh_v[w][db][lane][0]);
    sk_.x += x.x; sk_.y += x.y; sk_.z += x.z; sk_.w += x.w;
    sv_.x += y.x; sv_.y += y.y; sv_.z += y.z; sv_.w += y.w;
  }
  if (k0 + kl < Nk) {
    *reinterpret_cast<float4*>(a.dk + (int64_t)b * a.sdk + (int64_t)(k0 + kl) * a.lddk + h * 64 + db * 16 + 4 * g) = sk_;
    *reinterpret_cast<float4*>(a.dv + (int64_t)b * a.sdv + (int64_t)(k0 + kl) * a.lddv + h * 64 + db * 16 + 4 * g) = sv_;
  }
}


}  // namespace

extern "C" int sp3_attention_ex(const void* q, int64_t sq, int64_t ldq, const void* k, int64_t sk, int64_t ldk, const void* vt,
                                int64_t vt_ld, void* out, int64_t ldo, int out_bf16, int out_packed, int B, int heads, int Nq,
                                int Nk, float scale, int dtype, void* stream) {
  SP3_CHECK(q && k && vt && out, "sp3_attention: null pointer");
  SP3_CHECK(B > 0 && heads > 0 && Nq > 0 && Nk > 0, "sp3_attention: bad shape");
  SP3_CHECK(vt_ld >= ((Nk + 63) / 64) * 64 && vt_ld % 8 == 0, "sp3_attention: vt_ld=%lld must be >= Nk padded to 64", (long long)vt_ld);
  SP3_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && (out_packed || ldo % 4 == 0), "sp3_attention: row strides must keep 16-byte alignment");
  SP3_CHECK(dtype == SP3_F32 || dtype == SP3_BF16 || dtype == 2 || dtype == 3,
            "sp3_attention: bad dtype %d (0 fp32, 1 bf16, 2 / 3: fp32 operands with bf16x3 / fp16x3 split products)", dtype);
  const int xcd_map = attn_xcd_ok(heads, B) ? heads * B : 0;
  dim3 grid((Nq + 15) / 16, xcd_map ? (heads * B + 7) / 8 * 8 : heads, xcd_map ? 1 : B);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SP3_BF16)
    hipLaunchKernelGGL(attention_kernel<__bf16>, grid, dim3(256), 0, st, reinterpret_cast<const __bf16*>(q), sq, ldq,
                       reinterpret_cast<const __bf16*>(k), sk, ldk, reinterpret_cast<const __bf16*>(vt), vt_ld, out, ldo,
                       out_bf16, out_packed, heads, Nq, Nk, scale, (float*)nullptr, xcd_map);
  else if (dtype == 3)
    hipLaunchKernelGGL(attention_kernel<F16X3>, grid, dim3(256), 0, st, reinterpret_cast<const float*>(q), sq, ldq,
                       reinterpret_cast<const float*>(k), sk, ldk, reinterpret_cast<const float*>(vt), vt_ld, out, ldo, out_bf16,
                       out_packed, heads, Nq, Nk, scale, (float*)nullptr, xcd_map);
  else if (dtype == 2)
    hipLaunchKernelGGL(attention_kernel<F32X3>, grid, dim3(256), 0, st, reinterpret_cast<const float*>(q), sq, ldq,
                       reinterpret_cast<const float*>(k), sk, ldk, reinterpret_cast<const float*>(vt), vt_ld, out, ldo, out_bf16,
                       out_packed, heads, Nq, Nk, scale, (float*)nullptr, xcd_map);
  else
    hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(256), 0, st, reinterpret_cast<const float*>(q), sq, ldq,
                       reinterpret_cast<const float*>(k), sk, ldk, reinterpret_cast<const float*>(vt), vt_ld, out, ldo, out_bf16,
                       out_packed, heads, Nq, Nk, scale, (float*)nullptr, xcd_map);
  SP3_LAUNCH_CHECK("sp3_attention");
  return 0;
}

extern "C" int sp3_attention(const void* q, int64_t sq, int64_t ldq, const void* k, int64_t sk, int64_t ldk, const void* vt,
                             int64_t vt_ld, void* out, int64_t ldo, int out_bf16, int B, int heads, int Nq, int Nk, float scale,
                             int dtype, void* stream) {
  return sp3_attention_ex(q, sq, ldq, k, sk, ldk, vt, vt_ld, out, ldo, out_bf16, 0, B, heads, Nq, Nk, scale, dtype, stream);
}

extern "C" int sp3_attention_packed(const void* qp, int q_cols, int q_col0, int npad_q, const void* kp, int k_cols, int k_col0,
                                    int npad_k, const void* vtp, void* out, int64_t ldo, int out_bf16, int out_packed, int B,
                                    int heads, int Nq, int Nk, float scale, int o_group, int o_group_rows, void* stream) {
  SP3_CHECK(qp && kp && vtp && out, "sp3_attention_packed: null pointer");
  SP3_CHECK(B > 0 && heads > 0 && Nq > 0 && Nk > 0, "sp3_attention_packed: bad shape");
  SP3_CHECK(npad_q % 16 == 0 && npad_q >= ((Nq + 15) / 16) * 16 && npad_k % 64 == 0 && npad_k >= ((Nk + 63) / 64) * 64,
            "sp3_attention_packed: npad_q=%d / npad_k=%d do not cover Nq=%d / Nk=%d", npad_q, npad_k, Nq, Nk);
  SP3_CHECK(q_cols % 64 == 0 && k_cols % 64 == 0 && q_col0 % 64 == 0 && k_col0 % 64 == 0, "sp3_attention_packed: column geometry");
  SP3_CHECK(out_packed || ldo % 4 == 0, "sp3_attention_packed: ldo");
  SP3_CHECK(o_group == 0 || (o_group > 0 && B % o_group == 0 && o_group_rows >= o_group * Nq), "sp3_attention_packed: output grouping");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int np = heads * B, xcd_map = attn_xcd_ok(heads, B) ? np : 0;
  const dim3 tail = xcd_map ? dim3((np + 7) / 8 * 8, 1) : dim3(heads, B);
  // two query blocks per workgroup once the grid still fills the chip that way (profiles/r05_attn_long_variants.txt)
  if (Nq >= 512) {
    constexpr int QB = 2, lds = (4 * QB * 4 * 64 * 4 + 2 * 4 * QB * 64) * 4;
    hipLaunchKernelGGL((attention_packed_kernel<QB, 3>), dim3((Nq + 16 * QB - 1) / (16 * QB), tail.x, tail.y), dim3(256), lds, st,
                       reinterpret_cast<const __bf16*>(qp), q_cols, q_col0, npad_q, reinterpret_cast<const __bf16*>(kp), k_cols, k_col0,
                       npad_k, reinterpret_cast<const __bf16*>(vtp), out, ldo, out_bf16, out_packed, heads, Nq, Nk, scale, o_group, o_group_rows,
                       xcd_map, QProjArgs{});
  } else {
    constexpr int QB = 1, lds = (4 * QB * 4 * 64 * 4 + 2 * 4 * QB * 64) * 4;
    hipLaunchKernelGGL((attention_packed_kernel<QB, 4>), dim3((Nq + 15) / 16, tail.x, tail.y), dim3(256), lds, st,
                       reinterpret_cast<const __bf16*>(qp), q_cols, q_col0, npad_q, reinterpret_cast<const __bf16*>(kp), k_cols, k_col0,
                       npad_k, reinterpret_cast<const __bf16*>(vtp), out, ldo, out_bf16, out_packed, heads, Nq, Nk, scale, o_group, o_group_rows,
                       xcd_map, QProjArgs{});
  }
  SP3_LAUNCH_CHECK("sp3_attention_packed");
  return 0;
}

extern "C" int sp3_attention_packed_qproj(const sp3_attn_qproj_desc* dp, void* stream) {
  SP3_CHECK(dp != nullptr, "sp3_attention_packed_qproj: null descriptor");
  const sp3_attn_qproj_desc& d = *dp;
  SP3_CHECK(d.x_packed && d.ln_stats && d.w_packed && d.ln_s && d.bias && d.pos && d.rope_cos && d.rope_sin && d.kp && d.vtp && d.out,
            "sp3_attention_packed_qproj: null pointer");
  SP3_CHECK(d.D == 768 && d.heads * 64 == d.D, "sp3_attention_packed_qproj: the decoder width (D = 768 = heads x 64) only, got D=%d heads=%d", d.D, d.heads);
  SP3_CHECK(d.B > 0 && d.Nq > 0 && d.Nk > 0 && d.Nq < 512, "sp3_attention_packed_qproj: bad shape (one query block per workgroup: Nq < 512)");
  SP3_CHECK(d.npad_k % 64 == 0 && d.npad_k >= ((d.Nk + 63) / 64) * 64 && d.k_cols % 64 == 0 && d.k_col0 % 64 == 0, "sp3_attention_packed_qproj: key geometry");
  SP3_CHECK(d.out_packed || d.ldo % 4 == 0, "sp3_attention_packed_qproj: ldo");
  SP3_CHECK(d.o_group > 0 && d.B % d.o_group == 0 && d.o_group_rows >= d.o_group * d.Nq, "sp3_attention_packed_qproj: groups (images per decoder side)");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  SP3_CHECK(al16(d.x_packed) && al16(d.w_packed) && al16(d.ln_stats) && al16(d.ln_s) && al16(d.bias) && al16(d.rope_cos) && al16(d.rope_sin) &&
            (d.x_group_stride & 7) == 0 && (d.w_group_stride & 7) == 0 && (d.stats_group_stride & 3) == 0 && (d.vec_group_stride & 3) == 0,
            "sp3_attention_packed_qproj: 16-byte alignment");
  QProjArgs qa;
  qa.X = reinterpret_cast<const __bf16*>(d.x_packed); qa.x_gs = d.x_group_stride;
  qa.st = d.ln_stats; qa.st_gs = d.stats_group_stride;
  qa.W = reinterpret_cast<const __bf16*>(d.w_packed); qa.w_gs = d.w_group_stride;
  qa.ln_s = d.ln_s; qa.bias = d.bias; qa.v_gs = d.vec_group_stride;
  qa.pos = d.pos; qa.cos = d.rope_cos; qa.sin = d.rope_sin;
  qa.eps = d.ln_eps; qa.group_imgs = d.o_group; qa.img_tokens = d.Nq;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int np = d.heads * d.B, xcd_map = attn_xcd_ok(d.heads, d.B) ? np : 0;
  const dim3 tail = xcd_map ? dim3((np + 7) / 8 * 8, 1) : dim3(d.heads, d.B);
  constexpr int QB = 1, lds = (4 * QB * 4 * 64 * 4 + 2 * 4 * QB * 64) * 4;
  static_assert(lds >= 4 * 16 * 68 * 4, "the q projection's partial tiles fit the merge scratch");
  hipLaunchKernelGGL((attention_packed_kernel<QB, 2, 12>), dim3((d.Nq + 15) / 16, tail.x, tail.y), dim3(256), lds, st,
                     (const __bf16*)nullptr, 0, 0, 16, reinterpret_cast<const __bf16*>(d.kp), d.k_cols, d.k_col0, d.npad_k,
                     reinterpret_cast<const __bf16*>(d.vtp), d.out, d.ldo, d.out_bf16, d.out_packed, d.heads, d.Nq, d.Nk, d.scale, d.o_group,
                     d.o_group_rows, xcd_map, qa);
  SP3_LAUNCH_CHECK("sp3_attention_packed_qproj");
  return 0;
}

extern "C" int sp3_attention_train_fwd(const float* q, int64_t sq, int64_t ldq, const float* k, int64_t sk, int64_t ldk, const float* vt,
                                       int64_t vt_ld, float* out, int64_t ldo, float* lse, int B, int heads, int Nq, int Nk, float scale,
                                       int bf16_products, void* stream) {
  SP3_CHECK(q && k && vt && out && lse, "sp3_attention_train_fwd: null pointer");
  SP3_CHECK(B > 0 && heads > 0 && Nq > 0 && Nk > 0 && B <= 65535 && heads <= 65535, "sp3_attention_train_fwd: bad shape");
  SP3_CHECK(vt_ld >= ((Nk + 63) / 64) * 64 && vt_ld % 8 == 0, "sp3_attention_train_fwd: vt_ld=%lld must be >= Nk padded to 64", (long long)vt_ld);
  SP3_CHECK(ldq % 4 == 0 && ldk % 4 == 0 && ldo % 4 == 0 && sq % 4 == 0 && sk % 4 == 0, "sp3_attention_train_fwd: strides must keep 16-byte alignment");
  const int xcd_map = attn_xcd_ok(heads, B) ? heads * B : 0;
  dim3 grid((Nq + 15) / 16, xcd_map ? (heads * B + 7) / 8 * 8 : heads, xcd_map ? 1 : B);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bf16_products)
    hipLaunchKernelGGL(attention_kernel<BF16X1>, grid, dim3(256), 0, st, q, sq, ldq, k, sk, ldk, vt, vt_ld, (void*)out, ldo, 0, 0, heads, Nq, Nk, scale, lse, xcd_map);
  else
    hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(256), 0, st, q, sq, ldq, k, sk, ldk, vt, vt_ld, (void*)out, ldo, 0, 0, heads, Nq, Nk, scale, lse, xcd_map);
  SP3_LAUNCH_CHECK("sp3_attention_train_fwd");
  return 0;
}

extern "C" int sp3_attention_train_bwd(const sp3_attn_bwd_desc* d, void* stream) {
  SP3_CHECK(d && d->q && d->k && d->v && d->dout && d->qT && d->kT && d->doT && d->lse && d->D && d->dq && d->dk && d->dv,
            "sp3_attention_train_bwd: null pointer");
  SP3_CHECK(d->B > 0 && d->heads > 0 && d->Nq > 0 && d->Nk > 0 && d->B <= 65535 && d->heads <= 65535, "sp3_attention_train_bwd: bad shape");
  SP3_CHECK(d->ldTq >= ((d->Nq + 63) / 64) * 64 && d->ldTk >= ((d->Nk + 63) / 64) * 64 && d->ldTq % 4 == 0 && d->ldTk % 4 == 0,
            "sp3_attention_train_bwd: the transposed operands must be padded to a multiple of 64 (ldTq=%lld ldTk=%lld)", (long long)d->ldTq, (long long)d->ldTk);
  const int64_t strides[] = {d->sq, d->ldq, d->sk, d->ldk, d->sv, d->ldv, d->sdo, d->lddo, d->sdq, d->lddq, d->sdk, d->lddk, d->sdv, d->lddv};
  for (int64_t x : strides) SP3_CHECK(x % 4 == 0, "sp3_attention_train_bwd: strides must keep 16-byte alignment");
  AttnBwdArgs a;
  a.q = d->q; a.k = d->k; a.v = d->v; a.dout = d->dout;
  a.sq = d->sq; a.ldq = d->ldq; a.sk = d->sk; a.ldk = d->ldk; a.sv = d->sv; a.ldv = d->ldv; a.sdo = d->sdo; a.lddo = d->lddo;
  a.qT = d->qT; a.kT = d->kT; a.doT = d->doT; a.ldTq = d->ldTq; a.ldTk = d->ldTk;
  a.lse = d->lse; a.D = d->D; a.dq = d->dq; a.dk = d->dk; a.dv = d->dv;
  a.sdq = d->sdq; a.lddq = d->lddq; a.sdk = d->sdk; a.lddk = d->lddk; a.sdv = d->sdv; a.lddv = d->lddv;
  a.heads = d->heads; a.Nq = d->Nq; a.Nk = d->Nk; a.scale = d->scale;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 gq((d->Nq + 15) / 16, d->heads, d->B), gk((d->Nk + 15) / 16, d->heads, d->B);
  if (d->bf16_products) {
    hipLaunchKernelGGL(attention_bwd_dq_kernel<BF16X1>, gq, dim3(256), 0, st, a);
    hipLaunchKernelGGL(attention_bwd_dkv_kernel<BF16X1>, gk, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(attention_bwd_dq_kernel<float>, gq, dim3(256), 0, st, a);
    hipLaunchKernelGGL(attention_bwd_dkv_kernel<float>, gk, dim3(256), 0, st, a);
  }
  SP3_LAUNCH_CHECK("sp3_attention_train_bwd");
  return 0;
}
