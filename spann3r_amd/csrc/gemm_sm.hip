// sp3_gemm, the LEAN small-M instances ("tiles" 30..): the 196-row weight-streaming GEMMs of the per-frame step
// (croco/models/blocks.py:73-79,94-112,149-169 at B = 1: every Linear of a decoder / value-encoder block).
//
// Why a second kernel family (measured, tools/ubench/gemm_sm.hip, profiles/r04_gemm_small_m_*): at 196 rows a launch is
// ~2 us of launch boundary + one cold pass over its weights + an epilogue; the generic gemm_kernel spends another 2.5-5 us
// per launch on things that are not the GEMM -- a ~60-field run-time descriptor, a tile map with run-time divisions, a
// masked K tail path, epilogues that branch on a dozen options.  Here everything a shape fixes is a template constant:
//   * K (k-blocks), tile and wave count are compile-time; the grid is 3-D (x = XCD, y = M-tile, z = group x N-tile / 8), so the
//     tile map needs no division and the M-tiles that share a weight panel still land on one XCD, adjacent in dispatch order;
//   * K is split over ALL waves of the workgroup (4..16) and a wave requests every operand byte it will ever need before its
//     first MFMA (or keeps a RING of k-blocks in flight when K is long): one cold round trip per launch;
//   * the tile is chosen per shape so that the launch is ONE round of <= ~2 workgroups per CU with the fewest operand bytes
//     through the CUs' vector caches (the K loop is bound by L2 -> CU bytes at ~35 B/clk/CU, not by HBM or the MFMAs);
//   * every epilogue operand (bias, LayerNorm column sums and row statistics, residual rows, RoPE positions and table rows)
//     is requested before the K loop; partial tiles meet in LDS once; three epilogues only:
//       SM_PACKED  bias (+ folded LayerNorm) (+ GELU) -> fragment-order bf16           (fc1, key MLP hidden)
//       SM_STREAM  bias (+ folded LayerNorm) (+ residual) -> fp32 rows (+ per-32-column statistics + packed bf16 copy)
//       SM_ROPE    folded LayerNorm + bias + 2-D RoPE -> fragment-order q / k, V in PV-operand order  (q/k/v projections)
//       SM_SCORE   alpha x, folded LayerNorm, bias -> fp32 rows with ANY N (a multiple of 4: the bank grows by a frame per step) + the
//                  (max, sum exp) of every 32-column group: the score GEMM of the spatial-memory read (spann3r/model.py:159-166)
//       SM_PROB    (many-row family only, round 6) the same scores, never stored: exp(s - max of the row's 64-key group) -> fragment-order
//                  bf16 (the A operand of the P.V GEMM, pvs_kernel) + (max, sum) per group: the long-bank read without a score matrix
// Operands: A and W bf16 in fragment order (include/spann3r_hip.h a_packed / w_packed), fp32 accumulation.  Same arithmetic per
// output element as gemm_kernel up to the summation order over K (WK partial sums instead of 4).
#include "common.h"
#include "gemm_sm.h"
#include <cstdlib>
#include <type_traits>

namespace {

enum { SM_PACKED = 0, SM_STREAM = 1, SM_ROPE = 2, SM_SCORE = 3, SM_PROB = 4 };

struct SmOp {
  const char* A; const char* A2; const char* W; char* C;     // A2: second fragment-order source for k-blocks >= nkb1 (split A), else = A
  const float* bias; const float* res1; const float* ln_stats; const float* ln_s;
  float* stats_out; char* c2; char* vt;
  long gA, gA2, gW, gC, gbias, gres, gstats, gs, gso, gc2, gvt;      // byte strides per group (problem) of a grouped launch
  int N, ntz, ldc, rope_cols, act, nkb1;
  int nt;                           // N-tiles per problem (bm_kernel's tile maps)
  int ngrp, nzb;                    // bm_kernel: problems of this op (1 or 2), z-slots of its default map = ceil(ngrp x nt / 8)
  float alpha;                      // SM_SCORE: scale of the accumulator (the other epilogues serve alpha = 1 only)
};

struct SmArgs {
  SmOp op[2];                       // op[1]: second group of problems of a paired launch (blockIdx.z >= z1)
  const float* cos; const float* sin; const int* pos;
  int M, rb_max, z1;
  int mblk;                         // bm_kernel default map: 1 = blocks of 8 M-tiles x all z-slots in the XCD's dispatch order, see bm_kernel
  int xm;                           // bm_kernel: 1 = M-tiles across the XCDs (grid = (8, N-tiles x groups, ceil(M-tiles / 8))), see bm_kernel
  int tokens, heads, vt_ld;
  unsigned tok_magic;               // floor(2^32 / tokens) + 1: gm / tokens by one multiply-high (gm < 65536)
  float ln_eps;
  const int* dyn;                   // SM_SCORE: if set, N = dyn[0] (the bank's token count lives on the device; op[0].N bounds the grid)
};

// grid = (8, mt, nz): linear workgroup id = x + 8 (y + mt z) -> XCD x; tile_m = y; z = group * ntz + zt, tile_n = 8 zt + x
// SPLIT: A = [A | A2] along K (torch.cat(..., dim=-1) without the copy, spann3r/model.py:300): k-blocks [0, nkb1) come from A (a
// fragment-order [M, 64 nkb1] matrix), the rest from A2 ([M, 64 (NKB - nkb1)])
template <int MF, int NF, int WK, int NKB, int RING, int EPI, bool SPLIT = false>
__global__ __launch_bounds__(64 * WK) void sm_kernel(const SmArgs a) {
  constexpr int BM = MF * 16, BN = NF * 16, NT = 64 * WK;
  constexpr int NKW = NKB / WK;                               // k-blocks per wave
  constexpr int R = (RING == 0 || RING > NKW) ? NKW : RING;   // k-blocks in flight per wave
  constexpr int LD = BN + 4, CG = BN / 4;
  constexpr int IT = (BM * CG + NT - 1) / NT;                 // epilogue iterations per thread
  constexpr int K = NKB * 64;
  constexpr bool LNOK = (NKB % 4 == 0) && NKB <= 16 && BM * 4 <= NT;   // folded LayerNorm over K = 32 * (2 NKB) columns
  constexpr int NL4 = LNOK ? NKB / 4 : 1;                     // float4 (= two partials) per thread of a 4-thread row team
  static_assert(NKB % WK == 0, "k-blocks must divide over the waves");
  static_assert(NT % CG == 0, "a thread keeps one column group through the epilogue");
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [WK][BM][LD] partial tiles, then rowstat [BM][2]
  float* rowstat = smem + (size_t)WK * BM * LD;

  const int tid = threadIdx.x, lane = tid & 63, wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool second = EPI == SM_ROPE && (int)blockIdx.z >= a.z1;
#define OPF(f) (second ? a.op[1].f : a.op[0].f)
  int z = (int)blockIdx.z - (second ? a.z1 : 0);
  const int ntz = OPF(ntz);
  const int grp = z >= ntz ? 1 : 0;
  z -= grp * ntz;
  const int tile_m = blockIdx.y, tile_n = z * 8 + blockIdx.x;
  int N = OPF(N);
  if constexpr (EPI == SM_SCORE) {
    if (a.dyn) N = __builtin_amdgcn_readfirstlane(*a.dyn);     // (uniform: a scalar load; the grid was sized for op[0].N >= this)
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (n0 >= N) return;

  // ---- epilogue operands first (every vector of a launch is cold): bias / LayerNorm column sums of this thread's columns
  const int ec4 = (tid % CG) * 4;
  const int row_e0 = tid / CG;                                // epilogue row of iteration 0; iteration i: + i * NT / CG
  const float* lnst = OPF(ln_stats);
  const bool ln = lnst != nullptr;
  const float* biasp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(OPF(bias)) + grp * OPF(gbias));
  // (SM_SCORE: N is any multiple of 4 -- column groups past it re-read the last one, masked at the store)
  const int ecl = EPI == SM_SCORE ? ((n0 + ec4 < N - 4) ? n0 + ec4 : N - 4) : n0 + ec4;
  const float4 b4 = *reinterpret_cast<const float4*>(biasp + ecl);
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 lp[NL4];
  if constexpr (LNOK) {
    if (ln) {
      s4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(OPF(ln_s)) + grp * OPF(gs)) + ecl);
      const int srow = tid >> 2, sj = tid & 3;
      if (srow < BM) {
        int gm = m0 + srow;
        gm = gm < a.M ? gm : a.M - 1;
        const float4* ps = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lnst) + grp * OPF(gstats)) + (long)gm * NKB;
#pragma unroll
        for (int q = 0; q < NL4; ++q) lp[q] = ps[sj + 4 * q];
      }
    }
  }
  float4 pre_r[IT];
  float4 pre_cs[IT], pre_sn[IT];
  if constexpr (EPI == SM_STREAM) {
    const float* res = OPF(res1);
    if (res) {
      res = reinterpret_cast<const float*>(reinterpret_cast<const char*>(res) + grp * OPF(gres));
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        int gm = m0 + row_e0 + it * (NT / CG);
        gm = gm < a.M ? gm : a.M - 1;
        pre_r[it] = *reinterpret_cast<const float4*>(res + (long)gm * N + n0 + ec4);
      }
    } else {
#pragma unroll
      for (int it = 0; it < IT; ++it) pre_r[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const int rope_cols = EPI == SM_ROPE ? OPF(rope_cols) : 0;
  if constexpr (EPI == SM_ROPE) {
    if (n0 < rope_cols) {
      const int hc = (n0 + ec4) & 63, axis = hc >> 5, i0 = hc & 15;
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        int gm = m0 + row_e0 + it * (NT / CG);
        gm = gm < a.M ? gm : a.M - 1;
        const int p = a.pos[(long)gm * 2 + axis];
        pre_cs[it] = *reinterpret_cast<const float4*>(a.cos + p * 16 + i0);
        pre_sn[it] = *reinterpret_cast<const float4*>(a.sin + p * 16 + i0);
      }
    }
  }

  // ---- operands: wave wk owns k-blocks wk, wk + WK, ...
  const char* ap[MF];
  const char* ap2[SPLIT ? MF : 1];
  const char* wp[NF];
  const int nkb1 = SPLIT ? OPF(nkb1) : NKB;
  {
    const char* A = OPF(A) + grp * OPF(gA);
    const char* W = OPF(W) + grp * OPF(gW);
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      int rb = tile_m * MF + m;
      rb = rb < a.rb_max ? rb : a.rb_max;                      // row blocks past M: re-read the last one (masked at the store)
      ap[m] = A + ((long)rb * nkb1 + wk) * 2048 + lane * 16;
      if constexpr (SPLIT) ap2[m] = OPF(A2) + grp * OPF(gA2) + ((long)rb * (NKB - nkb1) + wk - nkb1) * 2048 + lane * 16;
    }
    const int nb_max = ((N + 15) >> 4) - 1;                   // (the last 16-row block of W may be partial: zero rows in the packed layout)
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      int nb = tile_n * NF + n;
      nb = nb < nb_max ? nb : nb_max;
      wp[n] = W + ((long)nb * NKB + wk) * 2048 + lane * 16;
    }
  }
  bf16x8 av[R][MF][2], wv[R][NF][2];
  auto load = [&](int slot, int i) {
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      wv[slot][n][0] = *reinterpret_cast<const bf16x8*>(wp[n] + (long)i * WK * 2048);
      wv[slot][n][1] = *reinterpret_cast<const bf16x8*>(wp[n] + (long)i * WK * 2048 + 1024);
    }
    const bool from2 = SPLIT && (wk + i * WK) >= nkb1;         // wave-uniform
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const char* q = (SPLIT && from2 ? ap2[m] : ap[m]) + (long)i * WK * 2048;
      av[slot][m][0] = *reinterpret_cast<const bf16x8*>(q);
      av[slot][m][1] = *reinterpret_cast<const bf16x8*>(q + 1024);
    }
  };
  f32x4 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < R; ++i) load(i, i);
#pragma unroll
  for (int i = 0; i < NKW; ++i) {
    const int s = i % R;
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][m][0], wv[s][n][0], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][m][1], wv[s][n][1], acc[m][n], 0, 0, 0);
      }
    if (i + R < NKW) load(s, i + R);
  }

  // ---- partial tiles -> LDS (C layout: col = lane & 15, row = 4 (lane >> 4) + reg); folded LayerNorm: mean / rstd of the rows
  {
    float* slab = smem + (size_t)wk * BM * LD;
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(m * 16 + 4 * g + r) * LD + n * 16 + c] = acc[m][n][r];
  }
  if constexpr (LNOK) {
    if (ln) {
      const int srow = tid >> 2;
      if (srow < BM) {                                         // (teams of 4 lanes never straddle the BM boundary)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NL4; ++q) { s1 += lp[q].x + lp[q].z; s2 += lp[q].y + lp[q].w; }
        s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
        s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
        if ((tid & 3) == 0) {
          const float mean = s1 * (1.0f / (float)K);
          const float var = fmaxf(s2 * (1.0f / (float)K) - mean * mean, 0.f);
          rowstat[2 * srow] = mean;
          rowstat[2 * srow + 1] = 1.0f / sqrtf(var + a.ln_eps);
        }
      }
    }
  }
  __syncthreads();

  // sum of the WK partial tiles of (row, 4 columns); y = rstd * acc - rstd * mean * s + bias for a folded LayerNorm
  auto finish4 = [&](int row, int c4, const float4& bb, const float4& ss, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(smem + row * LD + c4);
#pragma unroll
    for (int s = 1; s < WK; ++s) {
      const float4 u = *reinterpret_cast<const float4*>(smem + (size_t)s * BM * LD + row * LD + c4);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    if constexpr (EPI == SM_SCORE) {
      const float al = OPF(alpha);
      v[0] *= al; v[1] *= al; v[2] *= al; v[3] *= al;
    }
    if (LNOK && ln) {
      const float mean = rowstat[2 * row], rstd = rowstat[2 * row + 1], rm = rstd * mean;
      v[0] = rstd * v[0] - rm * ss.x; v[1] = rstd * v[1] - rm * ss.y;
      v[2] = rstd * v[2] - rm * ss.z; v[3] = rstd * v[3] - rm * ss.w;
    }
    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
  };

  if constexpr (EPI == SM_ROPE) {
    if (n0 >= rope_cols) {
      // ---- V columns: PV-operand order [(b,h)][key/32][d/16][lane = 16 (key%16 / 4) + d%16][8]: 4 consecutive tokens of one
      // column per thread are one 8-byte store (tokens % 4 == 0)
      __bf16* vt = reinterpret_cast<__bf16*>(OPF(vt) + grp * OPF(gvt));
      const float* lns = reinterpret_cast<const float*>(reinterpret_cast<const char*>(OPF(ln_s)) + grp * OPF(gs));
      for (int idx = tid; idx < (BM / 4) * BN; idx += NT) {
        const int rq = idx % (BM / 4), col = idx / (BM / 4);
        const int gm = m0 + 4 * rq, gn = n0 + col;
        if (gm >= a.M) continue;
        const float bias = biasp[gn];
        const float sn_ = ln ? lns[gn] : 0.f;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 4 * rq + i;
          float x = smem[row * LD + col];
#pragma unroll
          for (int s_ = 1; s_ < WK; ++s_) x += smem[(size_t)s_ * BM * LD + row * LD + col];
          if (LNOK && ln) { const float mean = rowstat[2 * row], rstd = rowstat[2 * row + 1]; x = rstd * x - rstd * mean * sn_; }
          v[i] = x + bias;
        }
        const int vc = gn - rope_cols;
        const int h = vc >> 6, dd = vc & 63;
        const int b = (int)__umulhi((unsigned)gm, a.tok_magic), n = gm - b * a.tokens;
        const int u = n >> 5, kk = n & 31, w16 = kk & 15;
        const int e = 4 * (kk >> 4), lane_ = (w16 >> 2) * 16 + (dd & 15);
        const long nU = a.vt_ld >> 5;
        const long off = ((((long)(b * a.heads + h) * nU + u) * 4 + (dd >> 4)) * 64 + lane_) * 8 + e;
        bf16x4 ob;
        ob[0] = (__bf16)v[0]; ob[1] = (__bf16)v[1]; ob[2] = (__bf16)v[2]; ob[3] = (__bf16)v[3];
        st_out(reinterpret_cast<bf16x4*>(vt + off), ob);
      }
      return;
    }
    // ---- q / k columns: bias, RoPE with the partner column (col ^ 16 inside the 64-wide head), fragment-order store with the
    // rows padded per image to vt_ld tokens.  The partner group's bias / column sums sit 4 lanes away.
    const float4 pb4 = make_float4(__shfl_xor(b4.x, 4), __shfl_xor(b4.y, 4), __shfl_xor(b4.z, 4), __shfl_xor(b4.w, 4));
    const float4 ps4 = make_float4(__shfl_xor(s4.x, 4), __shfl_xor(s4.y, 4), __shfl_xor(s4.z, 4), __shfl_xor(s4.w, 4));
    __bf16* out = reinterpret_cast<__bf16*>(OPF(C) + grp * OPF(gC));
    const int is_v = (((n0 + ec4) & 63) >> 4) & 1;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int row = row_e0 + it * (NT / CG), gm = m0 + row, gn = n0 + ec4;
      if (row >= BM || gm >= a.M) continue;
      float v[4], pv[4];
      finish4(row, ec4, b4, s4, v);
      finish4(row, ec4 ^ 16, pb4, ps4, pv);
      const float cs[4] = {pre_cs[it].x, pre_cs[it].y, pre_cs[it].z, pre_cs[it].w};
      const float sn[4] = {pre_sn[it].x, pre_sn[it].y, pre_sn[it].z, pre_sn[it].w};
      bf16x4 ob;
#pragma unroll
      for (int e = 0; e < 4; ++e) ob[e] = (__bf16)(is_v ? (v[e] * cs[e] + pv[e] * sn[e]) : (v[e] * cs[e] - pv[e] * sn[e]));
      const int b = (int)__umulhi((unsigned)gm, a.tok_magic), n = gm - b * a.tokens;
      st_out(reinterpret_cast<bf16x4*>(out + packed_off(b * a.vt_ld + n, gn, rope_cols, true)), ob);
    }
    return;
  } else if constexpr (EPI == SM_PACKED) {
    __bf16* out = reinterpret_cast<__bf16*>(OPF(C) + grp * OPF(gC));
    const bool gelu = OPF(act) == SP3_ACT_GELU;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int row = row_e0 + it * (NT / CG), gm = m0 + row, gn = n0 + ec4;
      if (row >= BM || gm >= a.M) continue;
      float v[4];
      finish4(row, ec4, b4, s4, v);
      if (gelu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
      }
      bf16x4 ob;
      ob[0] = (__bf16)v[0]; ob[1] = (__bf16)v[1]; ob[2] = (__bf16)v[2]; ob[3] = (__bf16)v[3];
      st_out(reinterpret_cast<bf16x4*>(out + packed_off(gm, gn, N, true)), ob);
    }
  } else if constexpr (EPI == SM_SCORE) {
    // scores of the memory read: fp32 rows, and per row and 32-column group (max, sum exp(x - max)) over the columns < N --
    // the softmax statistics leave with the scores (the P.V launch merges the groups; no pass over the score matrix)
    static_assert(BN == 32, "one statistics group per tile row");
    float* out = reinterpret_cast<float*>(OPF(C));
    float2* so = reinterpret_cast<float2*>(OPF(stats_out));
    const int ldc = OPF(ldc), ng = (N + 31) >> 5;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int row = row_e0 + it * (NT / CG), gm = m0 + row, gn = n0 + ec4;
      if (row >= BM || gm >= a.M) continue;                   // (the 8 lanes of a group share the row)
      float v[4];
      finish4(row, ec4, b4, s4, v);
      const int nvalid = N - gn;                              // >= 4, or <= 0 (N % 4 == 0)
      float mx = -INFINITY;
      if (nvalid > 0) mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
      for (int o_ = 1; o_ < 8; o_ <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o_));
      float se = 0.f;
      if (nvalid > 0) se = (__expf(v[0] - mx) + __expf(v[1] - mx)) + (__expf(v[2] - mx) + __expf(v[3] - mx));
#pragma unroll
      for (int o_ = 1; o_ < 8; o_ <<= 1) se += __shfl_xor(se, o_);
      if (((gn >> 2) & 7) == 0) so[(long)gm * ng + (gn >> 5)] = make_float2(mx, se);
      if (nvalid > 0) *reinterpret_cast<float4*>(out + (long)gm * ldc + gn) = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
    float* out = reinterpret_cast<float*>(OPF(C) + grp * OPF(gC));
    float* so = OPF(stats_out);
    __bf16* c2 = reinterpret_cast<__bf16*>(OPF(c2));
    if (so) so = reinterpret_cast<float*>(reinterpret_cast<char*>(so) + grp * OPF(gso));
    if (c2) c2 = reinterpret_cast<__bf16*>(reinterpret_cast<char*>(c2) + grp * OPF(gc2));
    const int ldc = OPF(ldc);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int row = row_e0 + it * (NT / CG), gm = m0 + row, gn = n0 + ec4;
      // (the 8 lanes of a 32-column group share a row: they enter or skip together, the shuffles below stay inside the group)
      if (row >= BM || gm >= a.M) continue;
      float v[4];
      finish4(row, ec4, b4, s4, v);
      v[0] += pre_r[it].x; v[1] += pre_r[it].y; v[2] += pre_r[it].z; v[3] += pre_r[it].w;
      if (so) {
        float s1 = (v[0] + v[1]) + (v[2] + v[3]);
        float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
        for (int o_ = 1; o_ < 8; o_ <<= 1) { s1 += __shfl_xor(s1, o_); s2 += __shfl_xor(s2, o_); }
        if (((gn >> 2) & 7) == 0) st_out(reinterpret_cast<float2*>(so) + ((long)gm * (N >> 5) + (gn >> 5)), make_float2(s1, s2));
      }
      if (c2) {
        bf16x4 ob;
        ob[0] = (__bf16)v[0]; ob[1] = (__bf16)v[1]; ob[2] = (__bf16)v[2]; ob[3] = (__bf16)v[3];
        st_out(reinterpret_cast<bf16x4*>(c2 + packed_off(gm, gn, N, true)), ob);
      }
      st_out(reinterpret_cast<float4*>(out + (long)gm * ldc + gn), make_float4(v[0], v[1], v[2], v[3]));
    }
  }
#undef OPF
}

// ------------------------------------------------------------------------------------------------ many rows (M > 256)
// The same three epilogues for the many-row Linears (whole-sequence encoder: M = frames x 196; a 512x512 frame: M = 1024; their
// K = 768 .. 4096): bm_kernel.  Both operand tiles arrive per workgroup by global_load_lds into a ring of fragment-order stages
// (the pipelined loop of gemm_kernel<LOOP = -1>: a slot is refilled right behind the barrier that publishes the next stage, two
// fragment sets per wave so the ds_reads of one half fly under the MFMAs of the other), every wave owns a 64 x (16 NF) output
// tile over the whole K -- and the MFMA operands are SWAPPED (D = W . A^T): a lane then holds 4 consecutive COLUMNS of one output
// row, so bias / LayerNorm fold / GELU / residual / RoPE partner (column ^ 16 = the neighbouring fragment of the same lane) /
// per-32-column statistics / fp32 and fragment-order bf16 stores all happen in registers on 8- and 16-byte row segments: no
// accumulator round trip through LDS (the general kernel's epilogue cost 12-18 k clocks per 256 x 128 workgroup, half of a
// K = 1024 launch).  Only the V columns of a q/k/v projection cross LDS once (their PV-operand layout wants 4 consecutive
// TOKENS per store).
__device__ __forceinline__ void bm_glds16(const char* gsrc, char* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
#else
  (void)gsrc; (void)lds;
#endif
}

// SPLIT (the key MLP's first layer at > 256 rows, spann3r/model.py:300): A = [A | A2] along K as two fragment-order matrices -- the A
// pieces of k-block kb come from A (kb < nkb1, nkb1 k-blocks per row block) or from A2 (NKB - nkb1 per row block).
template <int WM, int WN, int NF, int NKB, int NST, int EPI, bool SPLIT = false>
__global__ __launch_bounds__(64 * WM * WN) void bm_kernel(const SmArgs a) {
  constexpr int MF = 4, NW = WM * WN, NT = 64 * NW;
  constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
  constexpr int NBLK = BM / 16 + BN / 16, STAGE_BYTES = NBLK * 2048, NINSTR = 2 * NBLK, PER = (NINSTR + NW - 1) / NW;
  constexpr int K = NKB * 64;
  constexpr bool LNOK = NKB % 4 == 0 && NKB <= 16;
  static_assert(NST >= 2 && NST <= 4 && PER * (NST - 1) < 64 && NST <= NKB, "stage ring: 2..4 stages, vmcnt is a 6-bit count");
  static_assert(NF % 2 == 0, "RoPE partner / statistics pairs: an even number of column fragments per wave");
  extern __shared__ __attribute__((aligned(16))) char lds_b[];
  const int tid = threadIdx.x, lane = tid & 63, wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave_u % WN, wm = wave_u / WN;
  const int g = lane >> 4, r16 = lane & 15;
  const bool second = EPI == SM_ROPE && (int)blockIdx.z >= a.z1;
#define OPF(f) (second ? a.op[1].f : a.op[0].f)
  // Tile map (block b runs on XCD b % 8 = blockIdx.x).  Default: XCD x owns the N-tiles {x, x + 8, ..} of every M-tile -- its L2
  // fetches A once and 1/8 of W.  a.xm (host: A larger than W, i.e. more rows than columns -- the encoder's proj / fc2): XCD x
  // owns the M-tiles {x, x + 8, ..} of every N-tile -- 1/8 of A and all of W (PMC: 158 MB -> fetched per fc2 launch with the
  // default map, the activation panel once per XCD).
  int tile_m, tile_n, grp;
  if (a.xm) {
    const int nt = OPF(nt);
    int y = blockIdx.y, zm = blockIdx.z;
    if (a.xm == 2) {
      // many M-tiles AND many N-tiles (the 512 x 512 whole-sequence encoder: 64 x 32 tiles of 256 x 128): the 32 workgroups an XCD runs
      // at a time were the 32 N-tiles of ONE M-tile -- the whole W (8 MB at N = 4096, twice the L2) re-streamed per M-tile.  Blocks of
      // 4 M-tiles x 8 N-tiles instead (workgroup l = y + NT z of the XCD's dispatch order -> block l / 32): 2 MB of A + 2 MB of W
      // resident, 4 MB fetched per 32 tiles instead of 8.5.  Host: NT % 8 == 0 and gridDim.z % 4 == 0.
      const int NT = gridDim.y, l = y + NT * zm, b = l >> 5, r = l & 31, nb8 = NT >> 3;
      y = (b % nb8) * 8 + (r & 7);
      zm = (b / nb8) * 4 + (r >> 3);
    }
    grp = y >= nt ? 1 : 0;
    tile_n = y - grp * nt;
    tile_m = zm * 8 + blockIdx.x;
    if (tile_m * BM >= a.M) return;
  } else {
    // the (problem, N-tile) pairs of an op are numbered through (problem 0's tiles, then problem 1's) and dealt to the XCDs round robin:
    // pair p = 8 z + x.  (Per-problem z-slots left the XCDs 0 .. nt % 8 - 1 with one more N-tile PER PROBLEM: at N = 2304 / 1536 / 768 --
    // 18 / 12 / 12 tiles -- two XCDs ran 40 workgroups of a q/k/v + cross-k/v pair on their 32 CUs while four ran 24.)
    int z = (int)blockIdx.z - (second ? a.z1 : 0);
    tile_m = blockIdx.y;
    if (a.mblk) {
      // one op, >= 16 M-tiles, <= 4 z-slots (the 512 x 512 whole-sequence encoder's q/k/v: 64 M-tiles x 3 slots per XCD): the XCD's
      // workgroups in flight were 32 M-tiles of ONE N-tile, every A tile (512 KB) fetched again per N-tile.  Blocks of 8 M-tiles x all
      // its z-slots instead: an A tile is fetched once per XCD.  Host: gridDim.y % 8 == 0.
      const int nz = gridDim.z, l = tile_m + (int)gridDim.y * z, per = 8 * nz, b = l / per, r = l - b * per;
      tile_m = b * 8 + (r & 7);
      z = r >> 3;
    }
    const int nt = OPF(nt), p = z * 8 + (int)blockIdx.x;
    if (p >= nt * OPF(ngrp)) return;
    grp = p >= nt ? 1 : 0;
    tile_n = p - grp * nt;
  }
  int N = OPF(N);
  if constexpr (EPI == SM_PROB) {
    if (a.dyn) N = __builtin_amdgcn_readfirstlane(*a.dyn);        // the bank's token count lives on the device; op[0].N sized the grid
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (n0 >= N) return;

  // ---- DMA pieces of this wave: A row blocks first, then W column blocks (surplus slots repeat the last piece)
  const char* src[PER];
  const char* src2[SPLIT ? PER : 1];
  int dst[PER];
  const int nkb1 = SPLIT ? OPF(nkb1) : NKB;
  {
    const char* A = OPF(A) + grp * OPF(gA);
    const char* W = OPF(W) + grp * OPF(gW);
    const int nb_max = ((N + 15) >> 4) - 1;                       // (N % 16 != 0 only for SM_PROB: the bank's last, partly filled row block)
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int j = wave_u + i * NW;
      j = j < NINSTR ? j : NINSTR - 1;
      const int blk = j >> 1;
      const char* base;
      if constexpr (SPLIT) src2[i] = nullptr;
      if (blk < BM / 16) {
        int rb = (m0 >> 4) + blk;
        rb = rb < a.rb_max ? rb : a.rb_max;                   // rows past M: re-read the last block (masked at the store)
        base = A + (long)rb * nkb1 * 2048;
        if constexpr (SPLIT) src2[i] = OPF(A2) + grp * OPF(gA2) + ((long)rb * (NKB - nkb1) - nkb1) * 2048 + (j & 1) * 1024 + lane * 16;
      } else {
        int nb = (n0 >> 4) + blk - BM / 16;
        nb = nb < nb_max ? nb : nb_max;
        base = W + (long)nb * NKB * 2048;
      }
      src[i] = base + (j & 1) * 1024 + lane * 16;
      dst[i] = j * 1024;
    }
  }
  auto issue = [&](int slot, int kb) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const char* p = src[i];
      if constexpr (SPLIT) p = (kb >= nkb1 && src2[i]) ? src2[i] : p;     // (wave-uniform: a piece is an A piece or a W piece for all lanes)
      bm_glds16(p + (long)kb * 2048, lds_b + slot * STAGE_BYTES + dst[i]);
    }
  };
  auto wait_pending = [&](int pend) {
    if (pend >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
    else if (pend == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (pend == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // ---- per-row / per-column epilogue operands of this lane (rows m0 + wm*64 + m*16 + r16, columns cw0 + 16 n .. +3), requested
  // BEFORE the first DMA stages: the LayerNorm partials are reduced right behind the DMA prologue (in-order returns: that wait
  // retires only these older loads, not the stages), bias / column sums / RoPE positions stay in registers through the loop
  const float* lnst = OPF(ln_stats);
  const bool ln = LNOK && lnst != nullptr;
  float mean[MF], rstd[MF];
  int grow[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    grow[m] = m0 + wm * 64 + m * 16 + r16;
    mean[m] = 0.f; rstd[m] = 1.f;
  }
  constexpr int NL4 = LNOK ? NKB / 4 : 1;
  float4 lp[LNOK ? MF : 1][NL4];
  if constexpr (LNOK) {
    if (ln) {
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        const int gm = grow[m] < a.M ? grow[m] : a.M - 1;
        const float4* ps = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lnst) + grp * OPF(gstats)) + (long)gm * NKB;
#pragma unroll
        for (int q = 0; q < NL4; ++q) lp[m][q] = ps[g + 4 * q];
      }
    }
  }
  const int cw0 = n0 + wn * NF * 16 + 4 * g;
  const float* biasp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(OPF(bias)) + grp * OPF(gbias));
  float4 pb4[NF], ps4[NF];
#pragma unroll
  for (int n = 0; n < NF; ++n) {
    pb4[n] = *reinterpret_cast<const float4*>(biasp + cw0 + n * 16);
    ps4[n] = ln ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(OPF(ln_s)) + grp * OPF(gs)) + cw0 + n * 16)
                : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int ppos[EPI == SM_ROPE ? MF : 1][2];
  if constexpr (EPI == SM_ROPE) {
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int gm = grow[m] < a.M ? grow[m] : a.M - 1;
      ppos[m][0] = a.pos[(long)gm * 2];
      ppos[m][1] = a.pos[(long)gm * 2 + 1];
    }
  }
#pragma unroll
  for (int s_ = 0; s_ < NST; ++s_) issue(s_, s_);
  if constexpr (LNOK) {
    if (ln) {
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NL4; ++q) { s1 += lp[m][q].x + lp[m][q].z; s2 += lp[m][q].y + lp[m][q].w; }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        mean[m] = s1 * (1.0f / (float)K);
        const float var = fmaxf(s2 * (1.0f / (float)K) - mean[m] * mean[m], 0.f);
        rstd[m] = 1.0f / sqrtf(var + a.ln_eps);
      }
    }
  }

  typedef bf16x8 V16;
  V16 fa[2][MF], fw[2][NF];
  f32x4 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto read_half = [&](auto buf_tag, int slot, int half) {
    constexpr int BUF = decltype(buf_tag)::value;
    const char* st = lds_b + slot * STAGE_BYTES + half * 1024 + lane * 16;
#pragma unroll
    for (int m = 0; m < MF; ++m) fa[BUF][m] = *reinterpret_cast<const V16*>(st + (wm * MF + m) * 2048);
#pragma unroll
    for (int n = 0; n < NF; ++n) fw[BUF][n] = *reinterpret_cast<const V16*>(st + (BM / 16 + wn * NF + n) * 2048);
  };
  // swapped operands: D = W_frag . A_frag^T -> lane (r16, g) holds output row r16 of row block m, columns 4g .. 4g+3 of column block n
  auto mma_first = [&](auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[BUF][0], fa[BUF][0], acc[0][0], 0, 0, 0);
  };
  auto mma_rest = [&](auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n)
        if (m + n > 0) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[BUF][n], fa[BUF][m], acc[m][n], 0, 0, 0);
  };
  auto sync_stage = [&](int s) {
    const int newer = NKB - 1 - s;
    wait_pending(newer < NST - 2 ? newer : NST - 2);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (s + NST - 1 < NKB) issue((s + NST - 1) % NST, s + NST - 1);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  wait_pending(NST - 1);
  asm volatile("s_barrier" ::: "memory");
  int slot = 0;
  read_half(B0{}, 0, 0);
  for (int i = 0; i < NKB; ++i) {
    mma_first(B0{});
    __builtin_amdgcn_sched_barrier(0);
    read_half(B1{}, slot, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_rest(B0{});
    mma_first(B1{});
    __builtin_amdgcn_sched_barrier(0);
    slot = slot + 1 == NST ? 0 : slot + 1;
    if (i + 1 < NKB) {
      sync_stage(i + 1);
      read_half(B0{}, slot, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma_rest(B1{});
  }

  // ---- epilogue, in registers.  Column group of (n, this lane): cw0 + 16 n .. +3
  // y = rstd * acc - rstd * mean * s + bias (folded LayerNorm), else acc + bias
  auto finish = [&](int m, int n, float (&v)[4]) {
    const float4 b4 = pb4[n];
    v[0] = acc[m][n][0]; v[1] = acc[m][n][1]; v[2] = acc[m][n][2]; v[3] = acc[m][n][3];
    if constexpr (EPI == SM_PROB) {
      const float al = OPF(alpha);
      v[0] *= al; v[1] *= al; v[2] *= al; v[3] *= al;
    }
    if (ln) {
      const float4 s4 = ps4[n];
      const float rm = rstd[m] * mean[m];
      v[0] = rstd[m] * v[0] - rm * s4.x; v[1] = rstd[m] * v[1] - rm * s4.y;
      v[2] = rstd[m] * v[2] - rm * s4.z; v[3] = rstd[m] * v[3] - rm * s4.w;
    }
    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
  };

  if constexpr (EPI == SM_ROPE) {
    const int rope_cols = OPF(rope_cols);
    const int c_wave = n0 + wn * NF * 16;                         // first column of this wave (a multiple of 32)
    if (c_wave >= rope_cols) {
      // ---- V columns: PV-operand order wants 4 consecutive TOKENS of one column per store -> this wave's tile crosses LDS once
      __syncthreads();                                            // (the stage ring is dead; every wave of the workgroup gets here: n0 is workgroup-uniform, c_wave is not)
      constexpr int LDV = NF * 16 + 4;
      float* tl = reinterpret_cast<float*>(lds_b) + (size_t)wave_u * 64 * LDV;
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          float v[4];
          finish(m, n, v);
          *reinterpret_cast<float4*>(tl + (m * 16 + r16) * LDV + n * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // own tile only: no barrier
      __bf16* vt = reinterpret_cast<__bf16*>(OPF(vt) + grp * OPF(gvt));
      for (int idx = lane; idx < 16 * NF * 16; idx += 64) {       // 16 row quads x NF*16 columns
        const int rq = idx & 15, col = idx >> 4;
        const int gm = m0 + wm * 64 + 4 * rq, gn = c_wave + col;
        if (gm >= a.M) continue;
        const int vc = gn - rope_cols;
        const int h = vc >> 6, dd = vc & 63;
        const int b = (int)__umulhi((unsigned)gm, a.tok_magic), n_ = gm - b * a.tokens;
        const int u = n_ >> 5, kk = n_ & 31, w16 = kk & 15;
        const int e = 4 * (kk >> 4), lane_ = (w16 >> 2) * 16 + (dd & 15);
        const long nU = a.vt_ld >> 5;
        const long off = ((((long)(b * a.heads + h) * nU + u) * 4 + (dd >> 4)) * 64 + lane_) * 8 + e;
        bf16x4 ob;
#pragma unroll
        for (int i = 0; i < 4; ++i) ob[i] = (__bf16)tl[(4 * rq + i) * LDV + col];
        st_out(reinterpret_cast<bf16x4*>(vt + off), ob);
      }
      return;
    }
    if (n0 + BN > rope_cols) __syncthreads();                     // pairs with the V waves' barrier in a tile that straddles rope_cols
    // ---- q / k columns: the RoPE partner of column block n is block n ^ 1 (column ^ 16), same lane
    __bf16* out = reinterpret_cast<__bf16*>(OPF(C) + grp * OPF(gC));
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int gm = grow[m];
      if (gm >= a.M) continue;
      const int b = (int)__umulhi((unsigned)gm, a.tok_magic), n_ = gm - b * a.tokens;
      const int prow = b * a.vt_ld + n_;
      const int py = ppos[m][0], px = ppos[m][1];
#pragma unroll
      for (int n = 0; n < NF; n += 2) {
        const int hc = (c_wave + n * 16) & 63;                    // 0 or 32: axis
        const int p = (hc >> 5) ? px : py;
        const float4 cs4 = *reinterpret_cast<const float4*>(a.cos + p * 16 + 4 * g);
        const float4 sn4 = *reinterpret_cast<const float4*>(a.sin + p * 16 + 4 * g);
        const float cs[4] = {cs4.x, cs4.y, cs4.z, cs4.w}, sn[4] = {sn4.x, sn4.y, sn4.z, sn4.w};
        float v0[4], v1[4];
        finish(m, n, v0);
        finish(m, n + 1, v1);
        bf16x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] = (__bf16)(v0[e] * cs[e] - v1[e] * sn[e]);       // first half of the pair: x cos - y sin
          o1[e] = (__bf16)(v1[e] * cs[e] + v0[e] * sn[e]);       // second half:           y cos + x sin
        }
        st_out(reinterpret_cast<bf16x4*>(out + packed_off(prow, cw0 + n * 16, rope_cols, true)), o0);
        st_out(reinterpret_cast<bf16x4*>(out + packed_off(prow, cw0 + n * 16 + 16, rope_cols, true)), o1);
      }
    }
  } else if constexpr (EPI == SM_PACKED) {
    __bf16* out = reinterpret_cast<__bf16*>(OPF(C) + grp * OPF(gC));
    const bool gelu = OPF(act) == SP3_ACT_GELU;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int gm = grow[m];
      if (gm >= a.M) continue;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        float v[4];
        finish(m, n, v);
        if (gelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        bf16x4 ob;
        ob[0] = (__bf16)v[0]; ob[1] = (__bf16)v[1]; ob[2] = (__bf16)v[2]; ob[3] = (__bf16)v[3];
        st_out(reinterpret_cast<bf16x4*>(out + packed_off(gm, cw0 + n * 16, N, true)), ob);
      }
    }
  } else if constexpr (EPI == SM_PROB) {
    // The long-bank memory read without a score matrix (spann3r/model.py:159-183 at attn_thresh = 0): a wave's 64 columns are one
    // 64-key GROUP of its rows.  Per row: group maximum m_g, p~ = exp(s - m_g) as bf16 in fragment order [rows][ldc keys] -- the A
    // operand of pvs_kernel, which rescales each group's partial product by exp(m_g - m_row) / Z_row (sp3_prob_merge) -- and
    // (m_g, sum of p~) into stats[group][row] (rows contiguous: the merge and the P.V stage's DMA read them along the rows).
    static_assert(NF * 16 == 64, "SM_PROB: one 64-key group per wave");
    __bf16* out = reinterpret_cast<__bf16*>(OPF(C));
    float2* so = reinterpret_cast<float2*>(OPF(stats_out));
    const int ldk = OPF(ldc);
    const long rows_pad = (long)((a.M + 255) & ~255);
    const int grp64 = (n0 + wn * 64) >> 6;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int gm = grow[m];
      const bool ok = gm < a.M;
      float vv[NF][4];
      float mx = -INFINITY;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        finish(m, n, vv[n]);
        if (cw0 + n * 16 >= N) { vv[n][0] = vv[n][1] = vv[n][2] = vv[n][3] = -INFINITY; }     // keys past the bank's end (N % 4 == 0)
        mx = fmaxf(mx, fmaxf(fmaxf(vv[n][0], vv[n][1]), fmaxf(vv[n][2], vv[n][3])));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mref = mx == -INFINITY ? 0.f : mx;               // (a group entirely past the end: p~ = 0, statistics (-inf, 0))
      float se = 0.f;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        bf16x4 ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = __expf(vv[n][e] - mref);
          se += pe;
          ob[e] = (__bf16)pe;
        }
        if (ok) st_out(reinterpret_cast<bf16x4*>(out + packed_off(gm, cw0 + n * 16, ldk, true)), ob);
      }
      se += __shfl_xor(se, 16);
      se += __shfl_xor(se, 32);
      if (ok && g == 0) st_out(so + ((long)grp64 * rows_pad + gm), make_float2(mx, se));
    }
  } else {
    float* out = reinterpret_cast<float*>(OPF(C) + grp * OPF(gC));
    float* so = OPF(stats_out);
    __bf16* c2 = reinterpret_cast<__bf16*>(OPF(c2));
    const float* res = OPF(res1);
    if (so) so = reinterpret_cast<float*>(reinterpret_cast<char*>(so) + grp * OPF(gso));
    if (c2) c2 = reinterpret_cast<__bf16*>(reinterpret_cast<char*>(c2) + grp * OPF(gc2));
    if (res) res = reinterpret_cast<const float*>(reinterpret_cast<const char*>(res) + grp * OPF(gres));
    const int ldc = OPF(ldc);
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int gm = grow[m];
      const bool ok = gm < a.M;
      const int gmc = ok ? gm : a.M - 1;
      float4 r4[NF];
#pragma unroll
      for (int n = 0; n < NF; ++n) r4[n] = res ? *reinterpret_cast<const float4*>(res + (long)gmc * N + cw0 + n * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < NF; n += 2) {
        float v0[4], v1[4];
        finish(m, n, v0);
        finish(m, n + 1, v1);
        v0[0] += r4[n].x; v0[1] += r4[n].y; v0[2] += r4[n].z; v0[3] += r4[n].w;
        v1[0] += r4[n + 1].x; v1[1] += r4[n + 1].y; v1[2] += r4[n + 1].z; v1[3] += r4[n + 1].w;
        if (so) {
          // per-32-column (sum, sum of squares): column blocks n, n+1 of the 4 lanes that share the row (all lanes take part)
          float s1 = ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
          float s2 = ((v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3])) + ((v1[0] * v1[0] + v1[1] * v1[1]) + (v1[2] * v1[2] + v1[3] * v1[3]));
          s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
          if (ok && g == 0) st_out(reinterpret_cast<float2*>(so) + ((long)gm * (N >> 5) + ((cw0 + n * 16) >> 5)), make_float2(s1, s2));
        }
        if (ok) {
          if (c2) {
            bf16x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = (__bf16)v0[e]; o1[e] = (__bf16)v1[e]; }
            st_out(reinterpret_cast<bf16x4*>(c2 + packed_off(gm, cw0 + n * 16, N, true)), o0);
            st_out(reinterpret_cast<bf16x4*>(c2 + packed_off(gm, cw0 + n * 16 + 16, N, true)), o1);
          }
          st_out(reinterpret_cast<float4*>(out + (long)gm * ldc + cw0 + n * 16), make_float4(v0[0], v0[1], v0[2], v0[3]));
          st_out(reinterpret_cast<float4*>(out + (long)gm * ldc + cw0 + n * 16 + 16), make_float4(v1[0], v1[1], v1[2], v1[3]));
        }
      }
    }
  }
#undef OPF
}

// (round 5 also built this shell on v_mfma_f32_32x32x16_bf16 fragments -- bm32_kernel: loop 7 % faster in the probe, kernel 1-4 % slower in the
//  model, profiles/r05_bm32_in_model_ab.txt; removed in round 6, the probe keeps its loop: tools/ubench/gemm_bm.hip)

template <int WM, int WN, int NF, int NKB, int NST, int EPI, bool SPLIT = false>
int bm_launch(const SmArgs& a, int mt, int nz, hipStream_t stream) {
  constexpr int BM = WM * 64, BN = WN * NF * 16;
  constexpr size_t ring = (size_t)NST * (BM / 16 + BN / 16) * 2048;
  constexpr size_t vtile = (size_t)WM * WN * 64 * (NF * 16 + 4) * sizeof(float);     // the V columns' transposition (ROPE epilogue)
  constexpr size_t lds = ring > vtile ? ring : vtile;
  static_assert(lds <= 160 * 1024, "stage ring must fit the LDS");
  auto kern = bm_kernel<WM, WN, NF, NKB, NST, EPI, SPLIT>;
  if (lds > 64 * 1024) {
    static bool raised = false;
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { sp3_set_error("sp3_gemm (lean, many rows): cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return 2; }
      raised = true;
    }
  }
  const dim3 grid = a.xm ? dim3(8, a.op[0].nt * a.op[0].ngrp, (mt + 7) / 8) : dim3(8, mt, nz);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, stream, a);
  SP3_LAUNCH_CHECK("sp3_gemm (lean, many rows)");
  return 0;
}

// ------------------------------------------------------------------------------------------------ memory read, second launch
// out = softmax_thresh(S) . V_hat + q of the spatial-memory read (spann3r/model.py:159-183) for short banks (every per-frame read of
// the 224 x 224 demo): A is the fp32 score matrix the SCORE instance above left, W = V_hat^T in fragment order with K = bank tokens --
// a RUN-TIME k-block count.  Same shell as sm_kernel (3-D grid, K over all waves, every load of a wave's first R k-blocks requested
// up front, one LDS reduction); per workgroup 16 query rows x 64 output columns:
//   * the (max, sum exp) partials of the rows' 32-key groups (<= 8 per thread, 32 threads per row) are merged to (m, Z) while the
//     first operand loads fly;
//   * a lane turns the 8 fp32 scores behind its MFMA operand into p = exp2(s log2e - m log2e - log2 Z), drops p < thresh and keys
//     past the bank's end, adds the kept mass (the renormalisation of model.py:170-172) -- no probability matrix in memory;
//   * epilogue: sum of the WK partial tiles / kept mass + q -> fp32 rows (+ fragment-order bf16 copy), (kept mass, m, 1/Z) per row
//     for the column sums that follow (sp3_colsum_softmax).
struct PvArgs {
  const float* S; const float2* stats; const char* W; float* out; const float* res; char* c2; float* zout;
  int M, Mk, N, ld, ng, nkbw, ldc, ldr;
  float thr;
  const int* dyn;                   // if set: Mk = dyn[0], ng = ceil(Mk / 32) (the bank's token count lives on the device)
};

template <int NF, int WK, int R>
__global__ __launch_bounds__(64 * WK) void pv_kernel(const PvArgs a) {
  constexpr int BM = 16, BN = NF * 16, NT = 64 * WK, LD = BN + 4, TPR = NT / BM;    // TPR threads merge one row's statistics
  constexpr float L2E = 1.44269504088896341f;
  static_assert(TPR == 32 || TPR == 64, "statistics merge: 32 or 64 threads per row");
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [WK][BM][LD] partial tiles (kept mass in column BN), then crow[BM][4]
  float* crow = smem + (size_t)WK * BM * LD;
  const int tid = threadIdx.x, lane = tid & 63, wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int tile_m = blockIdx.y, tile_n = blockIdx.z * 8 + blockIdx.x;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (n0 >= a.N) return;
  const int Mk = a.dyn ? __builtin_amdgcn_readfirstlane(*a.dyn) : a.Mk;
  const int ng = a.dyn ? (Mk + 31) >> 5 : a.ng;
  const int nkb = (Mk + 63) >> 6;
  const int nkw = wk < nkb ? (nkb - wk + WK - 1) / WK : 0;       // k-blocks of this wave: wk, wk + WK, ...
  // ---- operand loads of the first R k-blocks
  int row = m0 + r16;
  row = row < a.M ? row : a.M - 1;
  const float* srow = a.S + (long)row * a.ld + g * 16;
  const char* wp[NF];
#pragma unroll
  for (int n = 0; n < NF; ++n) wp[n] = a.W + ((long)((n0 >> 4) + n) * a.nkbw) * 2048 + lane * 16;
  float4 sv[R][2][2];
  bf16x8 wv[R][NF][2];
  auto load = [&](int slot, int kb) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      sv[slot][h][0] = *reinterpret_cast<const float4*>(srow + kb * 64 + h * 8);
      sv[slot][h][1] = *reinterpret_cast<const float4*>(srow + kb * 64 + h * 8 + 4);
    }
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      wv[slot][n][0] = *reinterpret_cast<const bf16x8*>(wp[n] + (long)kb * 2048);
      wv[slot][n][1] = *reinterpret_cast<const bf16x8*>(wp[n] + (long)kb * 2048 + 1024);
    }
  };
  // ---- the 32-key groups' (max, sum exp) of this thread's row FIRST: the merge below then waits for these (older) loads only,
  // not for the operand loads behind them
  const int rowi = tid / TPR, st_t = tid % TPR;
  float2 stv[8];
  {
    int gm = m0 + rowi;
    gm = gm < a.M ? gm : a.M - 1;
    const float2* st = a.stats + (long)gm * ng;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = st_t + q * TPR;
      stv[q] = j < ng ? st[j] : make_float2(-INFINITY, 0.f);
    }
  }
#pragma unroll
  for (int s_ = 0; s_ < R; ++s_)
    if (s_ < nkw) load(s_, wk + s_ * WK);
  // epilogue operand: the residual (q) quad of this thread
  constexpr int CG = BN / 4;
  const int erow = tid / CG, ec4 = (tid % CG) * 4;
  float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < BM * CG && a.res) {
    int gm = m0 + erow;
    gm = gm < a.M ? gm : a.M - 1;
    r4 = *reinterpret_cast<const float4*>(a.res + (long)gm * a.ldr + n0 + ec4);
  }
  // ---- (m, Z) of the rows
  {
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < 8; ++q) mx = fmaxf(mx, stv[q].x);
#pragma unroll
    for (int o_ = 1; o_ < TPR; o_ <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o_));
    float z = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) z += stv[q].y * __expf(stv[q].x - mx);          // (empty groups: 0 * exp(-inf) = 0)
#pragma unroll
    for (int o_ = 1; o_ < TPR; o_ <<= 1) z += __shfl_xor(z, o_);
    if (st_t == 0) {
      crow[4 * rowi] = -mx * L2E - __log2f(z);
      crow[4 * rowi + 1] = mx;
      crow[4 * rowi + 2] = 1.0f / z;
    }
  }
  __syncthreads();
  const float c = crow[4 * r16];
  // ---- K loop
  f32x4 acc[NF];
#pragma unroll
  for (int n = 0; n < NF; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float zk = 0.f;
  for (int i0 = 0; i0 < nkw; i0 += R) {
#pragma unroll
    for (int s_ = 0; s_ < R; ++s_) {
      const int i = i0 + s_;
      if (i < nkw) {
        const int kb = wk + i * WK;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int left = Mk - (kb * 64 + g * 16 + h * 8);
          float p[8];
          const float4 s0 = sv[s_][h][0], s1 = sv[s_][h][1];
          const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float q = __builtin_amdgcn_exp2f(fmaf(sc[e], L2E, c));
            p[e] = (q < a.thr || e >= left) ? 0.f : q;
            zk += p[e];
          }
          const bf16x8 af = cvt8(make_float4(p[0], p[1], p[2], p[3]), make_float4(p[4], p[5], p[6], p[7]));
#pragma unroll
          for (int n = 0; n < NF; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wv[s_][n][h], acc[n], 0, 0, 0);
        }
        if (i + R < nkw) load(s_, kb + R * WK);
      }
    }
  }
  // ---- partial tiles and kept mass -> LDS (C layout: col = lane & 15, row = 4 (lane >> 4) + reg)
  {
    float* slab = smem + (size_t)wk * BM * LD;
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(4 * g + r) * LD + n * 16 + r16] = acc[n][r];
    zk += __shfl_xor(zk, 16);
    zk += __shfl_xor(zk, 32);
    if (g == 0) slab[r16 * LD + BN] = zk;
  }
  __syncthreads();
  if (tid < BM * CG) {
    const int gm = m0 + erow, gn = n0 + ec4;
    if (gm < a.M) {
      float4 t = *reinterpret_cast<const float4*>(smem + erow * LD + ec4);
      float zs = smem[erow * LD + BN];
#pragma unroll
      for (int s_ = 1; s_ < WK; ++s_) {
        const float4 u = *reinterpret_cast<const float4*>(smem + (size_t)s_ * BM * LD + erow * LD + ec4);
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        zs += smem[(size_t)s_ * BM * LD + erow * LD + BN];
      }
      const float izs = 1.0f / zs;
      const float4 o = make_float4(t.x * izs + r4.x, t.y * izs + r4.y, t.z * izs + r4.z, t.w * izs + r4.w);
      st_out(reinterpret_cast<float4*>(a.out + (long)gm * a.ldc + gn), o);
      if (a.c2) {
        bf16x4 ob;
        ob[0] = (__bf16)o.x; ob[1] = (__bf16)o.y; ob[2] = (__bf16)o.z; ob[3] = (__bf16)o.w;
        st_out(reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.c2) + packed_off(gm, gn, a.N, true)), ob);
      }
      if (a.zout && gn == 0) *reinterpret_cast<float4*>(a.zout + 4 * (long)gm) = make_float4(zs, crow[4 * erow + 1], crow[4 * erow + 2], 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------ long-bank read, P.V stage (round 6)
// out_partial[s] = sum over the key groups of slice s of  scale[row, group] * (P~[rows, group] . V_hat[group, :])   -- split K over S_k
// workgroup slices of the bank (sp3_reduce_ln adds the slices and q).  A = the fragment-order bf16 p~ = exp(s - m_group) that the
// SM_PROB score stage left, W = V_hat^T in fragment order (both with the bank's CAPACITY as their k-extent: one k-block = one 64-key
// group), scale[group][row] = exp(m_group - m_row) / Z_row from sp3_prob_merge.  The loop is bm_kernel's (LDS ring filled by
// global_load_lds, two fragment sets per wave, one barrier per k-block, swapped operands), with two differences: the k-block range
// [kb0, kb1) of a slice is a RUN-TIME quantity (the bank's token count may live on the device: dyn), and every k-block's product
// goes through a temporary accumulator (its first MFMA takes a zero C) that is folded into the running sum with the row's scale of
// that group -- 64 FMAs per wave next to 32 MFMAs.  The 256 scales of a stage (one per row of the tile) are its 49th DMA piece.
// No score or probability matrix in fp32 ever exists; the bf16 p~ is written once and read once per column tile.
struct PvsArgs {
  const char* A; const char* W; const float* scale; float* part;
  int M, N, Mk, nkbc, S_k, ldc;
  long rows_pad;
  const int* dyn;
};

template <int WM, int WN, int NF, int NST>
__global__ __launch_bounds__(64 * WM * WN) void pvs_kernel(const PvsArgs a) {
  constexpr int MF = 4, NW = WM * WN;
  constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
  constexpr int NBLK = BM / 16 + BN / 16, STAGE_BYTES = NBLK * 2048 + BM * 4, NINSTR = 2 * NBLK + BM / 256, PER = (NINSTR + NW - 1) / NW;
  static_assert(BM == 256, "the stage's scale piece is one 1 KB DMA: 256 rows");
  static_assert(NST >= 2 && NST <= 4 && PER * (NST - 1) < 64, "stage ring: 2..4 stages, vmcnt is a 6-bit count");
  extern __shared__ __attribute__((aligned(16))) char lds_b[];
  const int tid = threadIdx.x, lane = tid & 63, wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave_u % WN, wm = wave_u / WN;
  const int g = lane >> 4, r16 = lane & 15;
  // grid = (8, M-tiles x N-tiles, S_k / 8): XCD x owns the slices {x, x + 8, ..} with all their tiles (a slice's operand slabs enter one L2)
  const int slice = blockIdx.z * 8 + blockIdx.x;
  const int mt = (a.M + BM - 1) / BM;
  const int tile_m = (int)blockIdx.y % mt, tile_n = (int)blockIdx.y / mt;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Mk = a.dyn ? __builtin_amdgcn_readfirstlane(*a.dyn) : a.Mk;
  const int nkb_all = (Mk + 63) >> 6, per = (nkb_all + a.S_k - 1) / a.S_k;
  const int kb0 = slice * per;
  int nkb = nkb_all - kb0;
  nkb = nkb < per ? nkb : per;
  float* part = a.part + ((long)slice * a.M) * a.ldc;
  const int cw0 = n0 + wn * NF * 16 + 4 * g;
  int grow[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) grow[m] = m0 + wm * 64 + m * 16 + r16;
  if (nkb <= 0) {                                                 // an empty slice (short bank, many slices): its partial is zero
#pragma unroll
    for (int m = 0; m < MF; ++m)
      if (grow[m] < a.M)
#pragma unroll
        for (int n = 0; n < NF; ++n) st_out(reinterpret_cast<float4*>(part + (long)grow[m] * a.ldc + cw0 + n * 16), make_float4(0.f, 0.f, 0.f, 0.f));
    return;
  }
  // ---- DMA pieces of this wave: A row blocks, W column blocks, the scale piece (surplus slots repeat the last piece)
  const char* src[PER];
  int dst[PER];
  long kstep[PER];
  {
    const int rb_max = ((a.M + 15) >> 4) - 1;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int j = wave_u + i * NW;
      j = j < NINSTR ? j : NINSTR - 1;
      if (j < 2 * NBLK) {
        const int blk = j >> 1;
        const char* base;
        if (blk < BM / 16) {
          int rb = (m0 >> 4) + blk;
          rb = rb < rb_max ? rb : rb_max;
          base = a.A + (long)rb * a.nkbc * 2048;
        } else {
          base = a.W + (long)((n0 >> 4) + blk - BM / 16) * a.nkbc * 2048;
        }
        src[i] = base + (j & 1) * 1024 + lane * 16;
        dst[i] = j * 1024;
        kstep[i] = 2048;
      } else {
        src[i] = reinterpret_cast<const char*>(a.scale + m0) + lane * 16;      // scale[group][rows_pad]: the tile's 256 rows of one group
        dst[i] = NBLK * 2048;
        kstep[i] = a.rows_pad * 4;
      }
    }
  }
  auto issue = [&](int slot, int kb) {                            // kb: absolute k-block (= key group); clamped, so every call issues PER pieces
    const int kc = kb < kb0 + nkb ? kb : kb0 + nkb - 1;
#pragma unroll
    for (int i = 0; i < PER; ++i) bm_glds16(src[i] + kc * kstep[i], lds_b + slot * STAGE_BYTES + dst[i]);
  };
  auto wait_pending = [&](int pend) {
    if (pend >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
    else if (pend == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (pend == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
#pragma unroll
  for (int s_ = 0; s_ < NST; ++s_) issue(s_, kb0 + s_);           // (always NST stages: the counts below do not depend on nkb)

  typedef bf16x8 V16;
  V16 fa[2][MF], fw[2][NF];
  f32x4 acc[MF][NF], tmp[MF][NF];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[m][n] = zero4;
  auto read_half = [&](auto buf_tag, int slot, int half) {
    constexpr int BUF = decltype(buf_tag)::value;
    const char* st = lds_b + slot * STAGE_BYTES + half * 1024 + lane * 16;
#pragma unroll
    for (int m = 0; m < MF; ++m) fa[BUF][m] = *reinterpret_cast<const V16*>(st + (wm * MF + m) * 2048);
#pragma unroll
    for (int n = 0; n < NF; ++n) fw[BUF][n] = *reinterpret_cast<const V16*>(st + (BM / 16 + wn * NF + n) * 2048);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  auto sync_stage = [&](int s) {                                  // stage s (slice-relative) becomes readable; its predecessor's slot is refilled
    const int newer = NST - 1;                                    // (NST stages were always issued ahead: see issue())
    (void)newer;
    wait_pending(NST - 2);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue((s + NST - 1) % NST, kb0 + s + NST - 1);
  };
  wait_pending(NST - 1);
  asm volatile("s_barrier" ::: "memory");
  int slot = 0;
  read_half(B0{}, 0, 0);
  for (int i = 0; i < nkb; ++i) {
    // the row scales of this group (4 rows of this lane), from the stage's scale piece
    float sc[MF];
    {
      const float* sp = reinterpret_cast<const float*>(lds_b + slot * STAGE_BYTES + NBLK * 2048) + wm * 64 + r16;
#pragma unroll
      for (int m = 0; m < MF; ++m) sc[m] = sp[m * 16];
    }
    // first k16-pair of the block: C = 0
    tmp[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][0], fa[0][0], zero4, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_half(B1{}, slot, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n)
        if (m + n > 0) tmp[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][n], fa[0][m], zero4, 0, 0, 0);
    tmp[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][0], fa[1][0], tmp[0][0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    slot = slot + 1 == NST ? 0 : slot + 1;
    if (i + 1 < nkb) {
      sync_stage(i + 1);
      read_half(B0{}, slot, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n)
        if (m + n > 0) tmp[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][n], fa[1][m], tmp[m][n], 0, 0, 0);
    // fold the group's product into the running sum with the rows' scales
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[m][n][e] = fmaf(sc[m], tmp[m][n][e], acc[m][n][e]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the clamped surplus stages of a short slice are still landing in LDS)
#pragma unroll
  for (int m = 0; m < MF; ++m)
    if (grow[m] < a.M)
#pragma unroll
      for (int n = 0; n < NF; ++n)
        st_out(reinterpret_cast<float4*>(part + (long)grow[m] * a.ldc + cw0 + n * 16), make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]));
}

// loader SOFTMAX descriptors in their PROBABILITY form (tile 46): A = fragment-order bf16 p~ [M][ldw keys], sm_stats = the group scales
// [ldw / 64][M rounded up to 256], PARTIAL epilogue with splitk slices (a multiple of 8), W = V_hat^T [N][ldw]
bool pvs_ok(const sp3_gemm_desc& d) {
  return d.loader == SP3_LOAD_SOFTMAX && d.wdtype == SP3_BF16 && d.w_packed && d.a_bf16 && d.a_packed && d.sm_stats && d.epi == SP3_EPI_PARTIAL &&
         !d.out_bf16 && !d.out_packed && !d.bias && !d.res1 && !d.res2 && d.act == SP3_ACT_NONE && d.alpha == 1.0f && d.batch <= 1 &&
         d.splitk >= 8 && d.splitk % 8 == 0 && !d.ln_stats && !d.stats_out && !d.c2 && !d.trace && !d.sm_stats_out && !d.A2 && d.N % 128 == 0 &&
         d.K % 4 == 0 && d.K >= 4 && d.M >= 1 && d.M < 65536 && d.ldw % 64 == 0 && d.ldw >= d.K && (d.ldc & 3) == 0 && d.ldc >= d.N;
}

int pvs_dispatch(const sp3_gemm_desc& d, hipStream_t stream) {
  PvsArgs a;
  a.A = reinterpret_cast<const char*>(d.A); a.W = reinterpret_cast<const char*>(d.W); a.scale = d.sm_stats; a.part = reinterpret_cast<float*>(d.C);
  a.M = d.M; a.N = d.N; a.Mk = d.K; a.nkbc = (int)(d.ldw / 64); a.S_k = d.splitk; a.ldc = (int)d.ldc;
  a.rows_pad = (long)((d.M + 255) & ~255);
  a.dyn = d.dyn_n;
  constexpr int WM = 4, WN = 2, NF = 4, NST = 3, BM = 256, BN = 128;
  constexpr size_t lds = (size_t)NST * ((BM / 16 + BN / 16) * 2048 + BM * 4);
  static_assert(lds <= 160 * 1024, "stage ring must fit the LDS");
  auto kern = pvs_kernel<WM, WN, NF, NST>;
  static bool raised = false;
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { sp3_set_error("sp3_gemm (long-bank P.V): cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return 2; }
    raised = true;
  }
  const int mt = (d.M + BM - 1) / BM, nt = d.N / BN;
  hipLaunchKernelGGL(kern, dim3(8, mt * nt, d.splitk / 8), dim3(64 * WM * WN), lds, stream, a);
  SP3_LAUNCH_CHECK("sp3_gemm (long-bank P.V)");
  return 0;
}

// loader SOFTMAX descriptors this kernel serves (tile 44); anything else stays on the general tiles
bool pv_ok(const sp3_gemm_desc& d) {
  return d.loader == SP3_LOAD_SOFTMAX && d.wdtype == SP3_BF16 && d.w_packed && !d.a_bf16 && !d.a_packed && d.sm_stats && d.epi == SP3_EPI_PLAIN &&
         !d.out_bf16 && !d.out_packed && !d.bias && !d.res2 && d.act == SP3_ACT_NONE && d.alpha == 1.0f && d.batch <= 1 && d.splitk <= 1 &&
         !d.ln_stats && !d.stats_out && !d.trace && !d.sm_stats_out && !d.A2 && d.N % 64 == 0 && d.K % 4 == 0 && d.K >= 4 &&
         d.sm_nt == (d.K + 31) / 32 && d.sm_nt <= 8 * 32 && d.M >= 1 && d.M < 65536 * 16 && d.ldw % 64 == 0 && d.ldw >= d.K && (d.lda & 3) == 0 &&
         d.lda >= ((d.K + 63) & ~63) && (d.ldc & 3) == 0 && d.ldc >= d.N && (!d.res1 || (d.ldr1 & 3) == 0) && (!d.c2 || d.N % 64 == 0);
}

template <int NF, int WK, int R>
int pv_launch(const PvArgs& a, int M, int N, hipStream_t stream) {
  constexpr size_t lds = ((size_t)WK * 16 * (NF * 16 + 4) + 16 * 4) * sizeof(float);
  static_assert(lds <= 160 * 1024, "partial tiles must fit the LDS");
  auto kern = pv_kernel<NF, WK, R>;
  if (lds > 64 * 1024) {
    static bool raised = false;
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { sp3_set_error("sp3_gemm (lean, softmax P.V): cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return 2; }
      raised = true;
    }
  }
  const int nt = N / (NF * 16);
  hipLaunchKernelGGL(kern, dim3(8, (M + 15) / 16, (nt + 7) / 8), dim3(64 * WK), lds, stream, a);
  SP3_LAUNCH_CHECK("sp3_gemm (lean, softmax P.V)");
  return 0;
}

int pv_dispatch(const sp3_gemm_desc& d, hipStream_t stream) {
  PvArgs a;
  a.S = reinterpret_cast<const float*>(d.A); a.stats = reinterpret_cast<const float2*>(d.sm_stats); a.W = reinterpret_cast<const char*>(d.W);
  a.out = reinterpret_cast<float*>(d.C); a.res = d.res1; a.c2 = reinterpret_cast<char*>(d.c2); a.zout = d.sm_zout;
  a.M = d.M; a.Mk = d.K; a.N = d.N; a.ld = (int)d.lda; a.ng = d.sm_nt; a.nkbw = (int)(d.ldw / 64); a.ldc = (int)d.ldc; a.ldr = (int)d.ldr1;
  a.thr = d.sm_thresh;
  a.dyn = d.dyn_n;
  return pv_launch<4, 8, 2>(a, d.M, d.N, stream);      // (16 x 32 tiles, 16 waves, deeper rings: measured slower, profiles/r05_memread_short_bank_launch_breakdown.txt)
}

// ------------------------------------------------------------------------------------------------ small-map 3x3 convolutions
// The DPT heads' 3x3 convolutions on maps of <= 1024 pixels (croco/models/dpt_block.py:33-75,95-113: layer_rn, the
// ResidualConvUnits of refinenet 2-4, act_postprocess[3] at 7x7 .. 28x28 of a 224x224 frame): M = pixels is tiny, K = 9 Cin is
// long (2304 .. 6912), N = 256 / 768.  The general kernel ran them as split-K x 4-8 over 32x32 tiles plus a reduce launch
// (7.7 + 4.7 us at 14x14); here a workgroup owns a SMALL output tile (16x16: 208 workgroups at 196 pixels), splits K over its 12
// waves, gathers the im2col rows straight from the fp32 NHWC map (ReLU + bf16 rounding on the way into the MFMA, exactly the
// general loader's arithmetic) and finishes bias / ReLU / both residuals itself: one launch, no partial sums in memory.
struct ConvSmArgs {
  const void* x; const char* W; void* out;          // x / res1 / res2 / out: NHWC maps in the map dtype TM (fp32, or bf16 in bf16 mode)
  const float* bias; const void* res1; const void* res2;
  int M, N, H, Wd, Cin, OH, OW, stride, nkb, relu_in, act;
  unsigned cin_magic, per_magic, ow_magic;        // floor(2^32 / d) + 1: n / d by one multiply-high (n < 65536)
};

typedef __attribute__((ext_vector_type(4))) unsigned int csm_u32x4;

// one lane's 16 consecutive input channels of one tap, as the two 16-byte MFMA operand halves (ReLU, zero padding, bf16 rounding)
template <typename TM> struct CsmA;
template <> struct CsmA<float> {
  float4 v[4];
  __device__ __forceinline__ void load(const void* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
  }
  __device__ __forceinline__ void halves(unsigned mask, bool relu, bf16x8& h0, bf16x8& h1) const {
    float4 t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      t[q].x = __uint_as_float(__float_as_uint(v[q].x) & mask); t[q].y = __uint_as_float(__float_as_uint(v[q].y) & mask);
      t[q].z = __uint_as_float(__float_as_uint(v[q].z) & mask); t[q].w = __uint_as_float(__float_as_uint(v[q].w) & mask);
      if (relu) t[q] = relu4(t[q]);
    }
    h0 = cvt8(t[0], t[1]);
    h1 = cvt8(t[2], t[3]);
  }
};
template <> struct CsmA<__bf16> {
  bf16x8 v[2];
  __device__ __forceinline__ void load(const void* p) {
    const bf16x8* q = reinterpret_cast<const bf16x8*>(p);
    v[0] = q[0]; v[1] = q[1];
  }
  __device__ __forceinline__ void halves(unsigned mask, bool relu, bf16x8& h0, bf16x8& h1) const {
    csm_u32x4 a = __builtin_bit_cast(csm_u32x4, v[0]), b = __builtin_bit_cast(csm_u32x4, v[1]);
    a &= mask; b &= mask;
    h0 = __builtin_bit_cast(bf16x8, a);
    h1 = __builtin_bit_cast(bf16x8, b);
    if (relu) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        h0[i] = (float)h0[i] > 0.f ? h0[i] : (__bf16)0.f;
        h1[i] = (float)h1[i] > 0.f ? h1[i] : (__bf16)0.f;
      }
    }
  }
};
__device__ __forceinline__ float4 csm_load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 csm_load4(const __bf16* p) {
  const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
}
__device__ __forceinline__ void csm_store4(float* p, const float (&v)[4]) { st_out(reinterpret_cast<float4*>(p), make_float4(v[0], v[1], v[2], v[3])); }
__device__ __forceinline__ void csm_store4(__bf16* p, const float (&v)[4]) {
  bf16x4 t;
  t[0] = (__bf16)v[0]; t[1] = (__bf16)v[1]; t[2] = (__bf16)v[2]; t[3] = (__bf16)v[3];
  st_out(reinterpret_cast<bf16x4*>(p), t);
}

template <typename TM, int MF, int NF, int WK, int R>
__global__ __launch_bounds__(64 * WK) void conv_sm_kernel(const ConvSmArgs a) {
  constexpr int BM = MF * 16, BN = NF * 16, NT = 64 * WK, LD = BN + 4, CG = BN / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_m = blockIdx.y, tile_n = blockIdx.z * 8 + blockIdx.x;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (n0 >= a.N) return;
  const int g = lane >> 4, r16 = lane & 15;
  const TM* X = reinterpret_cast<const TM*>(a.x);

  // epilogue operands of this thread's (row, 4 columns) first
  const int ec4 = (tid % CG) * 4, erow = tid / CG;
  const bool eact = erow < BM && (m0 + erow) < a.M;
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = b4, r2 = b4;
  if (eact) {
    const long o = (long)(m0 + erow) * a.N + n0 + ec4;
    if (a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + n0 + ec4);
    if (a.res1) r1 = csm_load4(reinterpret_cast<const TM*>(a.res1) + o);
    if (a.res2) r2 = csm_load4(reinterpret_cast<const TM*>(a.res2) + o);
  }

  // this lane's output pixels (one per row block)
  const TM* img[MF];
  int iy0[MF], ix0[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    int r = m0 + m * 16 + r16;
    r = r < a.M ? r : a.M - 1;
    const int per = a.OH * a.OW;
    const int b = (int)__umulhi((unsigned)r, a.per_magic);
    const int rem = r - b * per;
    const int oy = (int)__umulhi((unsigned)rem, a.ow_magic), ox = rem - oy * a.OW;
    img[m] = X + (long)b * a.H * a.Wd * a.Cin;
    iy0[m] = oy * a.stride - 1;
    ix0[m] = ox * a.stride - 1;
  }
  const char* wp[NF];
#pragma unroll
  for (int n = 0; n < NF; ++n) wp[n] = a.W + ((long)(tile_n * NF + n) * a.nkb) * 2048 + lane * 16;

  CsmA<TM> av[R][MF];
  bf16x8 wv[R][NF][2];
  unsigned am[R][MF];
  auto load = [&](int slot, int kb) {
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      wv[slot][n][0] = *reinterpret_cast<const bf16x8*>(wp[n] + (long)kb * 2048);
      wv[slot][n][1] = *reinterpret_cast<const bf16x8*>(wp[n] + (long)kb * 2048 + 1024);
    }
    const int k0 = kb * 64 + g * 16;
    const int tap = (int)__umulhi((unsigned)k0, a.cin_magic), ci = k0 - tap * a.Cin;
    const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int iy = iy0[m] + dy, ix = ix0[m] + dx;
      const bool inb = iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd;
      am[slot][m] = inb ? 0xffffffffu : 0u;                 // unconditional load, masked at consume time (zero padding)
      av[slot][m].load(inb ? img[m] + ((long)iy * a.Wd + ix) * a.Cin + ci : X);
    }
  };
  f32x4 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nkw = (a.nkb - wk + WK - 1) / WK;              // k-blocks of this wave: wk, wk + WK, ...
#pragma unroll
  for (int i = 0; i < R; ++i)
    if (i < nkw) load(i, wk + i * WK);
  const bool relu = a.relu_in != 0;
  for (int i0 = 0; i0 < nkw; i0 += R) {
#pragma unroll
    for (int s = 0; s < R; ++s) {
      const int i = i0 + s;
      if (i < nkw) {
        bf16x8 ab[MF][2];
#pragma unroll
        for (int m = 0; m < MF; ++m) av[s][m].halves(am[s][m], relu, ab[m][0], ab[m][1]);
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int n = 0; n < NF; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[m][0], wv[s][n][0], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[m][1], wv[s][n][1], acc[m][n], 0, 0, 0);
          }
        if (i + R < nkw) load(s, wk + (i + R) * WK);
      }
    }
  }
  {
    float* slab = smem + (size_t)wk * BM * LD;
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(m * 16 + 4 * g + r) * LD + n * 16 + r16] = acc[m][n][r];
  }
  __syncthreads();
  static_assert(BM * CG <= NT, "one epilogue item per thread");
  if (eact) {
    float4 t = *reinterpret_cast<const float4*>(smem + erow * LD + ec4);
#pragma unroll
    for (int s = 1; s < WK; ++s) {
      const float4 u = *reinterpret_cast<const float4*>(smem + (size_t)s * BM * LD + erow * LD + ec4);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    float v[4] = {t.x + b4.x, t.y + b4.y, t.z + b4.z, t.w + b4.w};
    if (a.act == SP3_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    v[0] += r1.x + r2.x; v[1] += r1.y + r2.y; v[2] += r1.z + r2.z; v[3] += r1.w + r2.w;
    csm_store4(reinterpret_cast<TM*>(a.out) + (long)(m0 + erow) * a.N + n0 + ec4, v);
  }
}

template <typename TM, int MF, int NF, int WK, int R>
int conv_sm_launch(const ConvSmArgs& a, hipStream_t stream) {
  constexpr int BM = MF * 16, BN = NF * 16;
  constexpr size_t lds = (size_t)WK * BM * (BN + 4) * sizeof(float);
  static_assert(lds <= 64 * 1024, "partial tiles: default dynamic LDS limit");
  const int mt = (a.M + BM - 1) / BM, ntz = (a.N / BN + 7) / 8;
  hipLaunchKernelGGL((conv_sm_kernel<TM, MF, NF, WK, R>), dim3(8, mt, ntz), dim3(64 * WK), lds, stream, a);
  SP3_LAUNCH_CHECK("sp3_gemm (lean conv3x3)");
  return 0;
}

// tile 40: 16x16 outputs per workgroup, K over 12 waves (maps of <= 256 pixels); tile 41: 32x32, K over 8 waves (<= 2048 pixels).
// The map dtype is fp32, or bf16 for input AND output (bf16 mode of the DPT heads: a_bf16 = out_bf16 = 1; residuals in the same dtype)
int conv_sm_tile(const sp3_gemm_desc& d) {
  if (d.loader != SP3_LOAD_CONV3X3 || d.wdtype != SP3_BF16 || !d.w_packed || (d.a_bf16 != 0) != (d.out_bf16 != 0) || d.out_packed || d.epi != SP3_EPI_PLAIN ||
      d.batch > 1 || d.splitk > 1 || d.alpha != 1.0f || d.ln_stats || d.stats_out || d.c2 || d.trace || d.sm_stats_out || d.f32x3)
    return -1;
  if (d.K % 64 || d.conv_C % 16 || d.K != 9 * d.conv_C || d.K >= 65536 || d.M > 2048 || d.M < 1 || d.act == SP3_ACT_GELU) return -1;
  if (d.ldc != d.N || (d.res1 && d.ldr1 != d.N) || (d.res2 && d.ldr2 != d.N) || (d.ldw > 0 && d.ldw != d.K)) return -1;
  // the instances read res1 / res2 in the map dtype: a descriptor that says otherwise (fp32 residuals next to a bf16 map, the
  // contract the general tiles serve) is not theirs
  if ((d.res1 || d.res2) && (d.res_bf16 != 0) != (d.out_bf16 != 0)) return -1;
  // the kernel divides by conv_C, OH * OW and OW with floor(2^32 / d) + 1 multiply-highs: d = 1 overflows the 32-bit magic
  if (d.conv_OW < 2 || d.conv_OH * d.conv_OW < 2 || d.conv_C < 2) return -1;
  if (d.M <= 256) return d.N % 16 == 0 ? 40 : -1;
  return d.N % 32 == 0 ? 41 : -1;
}

int conv_sm_dispatch(const sp3_gemm_desc& d, int tile, hipStream_t stream) {
  ConvSmArgs a;
  a.x = d.A; a.W = reinterpret_cast<const char*>(d.W); a.out = d.C;
  a.bias = d.bias; a.res1 = d.res1; a.res2 = d.res2;
  a.M = d.M; a.N = d.N; a.H = d.conv_H; a.Wd = d.conv_W; a.Cin = d.conv_C; a.OH = d.conv_OH; a.OW = d.conv_OW;
  a.stride = d.conv_stride; a.nkb = d.K / 64; a.relu_in = d.relu_in; a.act = d.act;
  auto magic = [](int v) { return (unsigned)((1ull << 32) / (unsigned)v + 1); };
  a.cin_magic = magic(d.conv_C); a.per_magic = magic(d.conv_OH * d.conv_OW); a.ow_magic = magic(d.conv_OW);
  if (d.a_bf16) {
    if (tile == 40) return conv_sm_launch<__bf16, 1, 1, 12, 3>(a, stream);
    return conv_sm_launch<__bf16, 2, 2, 8, 3>(a, stream);
  }
  if (tile == 40) return conv_sm_launch<float, 1, 1, 12, 3>(a, stream);
  return conv_sm_launch<float, 2, 2, 8, 3>(a, stream);
}

// ------------------------------------------------------------------------------------------------ host side
struct SmInst {
  int tile, epi, K, MF, NF, WK;     // small-M family: tile = 16 MF x 16 NF, K over WK waves; many-row family: see bm / bn
  bool split;                       // serves descriptors with a second A source (and only those)
  int min_n;                        // N range this instance is the choice for (per problem)
  int max_n;
  int (*launch)(const SmArgs&, int mt, int nz, hipStream_t);
  int min_m = 1, max_m = 256;       // row range (small-M family: M <= 256)
  int bm = 0, bn = 0;               // many-row family (bm_kernel): workgroup tile
  int min_mb = 0, max_mb = 1 << 30; // range of M x batch (rows of all groups of the launch): 128-row tiles while they fit ONE round of workgroups
  int tile_m() const { return bm ? bm : MF * 16; }
  int tile_n() const { return bn ? bn : NF * 16; }
};

template <int MF, int NF, int WK, int NKB, int RING, int EPI, bool SPLIT = false>
int sm_launch(const SmArgs& a, int mt, int nz, hipStream_t stream) {
  constexpr int BM = MF * 16, BN = NF * 16;
  constexpr size_t lds = ((size_t)WK * BM * (BN + 4) + 2 * BM) * sizeof(float);
  static_assert(lds <= 160 * 1024, "partial tiles must fit the LDS");
  auto kern = sm_kernel<MF, NF, WK, NKB, RING, EPI, SPLIT>;
  if (lds > 64 * 1024) {
    static bool raised = false;
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { sp3_set_error("sp3_gemm (lean): cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return 2; }
      raised = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(8, mt, nz), dim3(64 * WK), lds, stream, a);
  SP3_LAUNCH_CHECK("sp3_gemm (lean)");
  return 0;
}

// The instances, chosen per shape from tools/ubench/gemm_sm.hip on MI355X (profiles/r04_gemm_small_m_tile_sweep.txt).
// tile ids 30.. are what sp3_gemm_desc.tile / the profiles call them.
const SmInst kInst[] = {
    // ROPE (q/k/v projections): BN % 32 == 0 (the RoPE partner column lies in the tile)
    {30, SM_ROPE, 1024, 3, 4, 8, false, 0, 1 << 30, sm_launch<3, 4, 8, 16, 0, SM_ROPE>},      // val / enc-step qkv: 48x64, K over 8 waves
    {31, SM_ROPE, 768, 2, 2, 4, false, 0, 1 << 30, sm_launch<2, 2, 4, 12, 0, SM_ROPE>},       // decoder qkv + ckv pair, cross q: 32x32 k4
    // PACKED (fc1 + GELU, key MLP hidden)
    {32, SM_PACKED, 1024, 4, 4, 8, false, 0, 1 << 30, sm_launch<4, 4, 8, 16, 0, SM_PACKED>},  // val fc1: 64x64 k8
    {33, SM_PACKED, 768, 2, 4, 4, false, 0, 1 << 30, sm_launch<2, 4, 4, 12, 0, SM_PACKED>},   // dec fc1 x2: 32x64 k4
    {42, SM_PACKED, 1792, 4, 2, 7, true, 0, 1 << 30, sm_launch<4, 2, 7, 28, 0, SM_PACKED, true>},   // key MLP hidden x2 (split A: feat | dec[-1]): 64x32 k7
    // STREAM (output projections onto the residual stream)
    {39, SM_STREAM, 1024, 3, 2, 8, false, 0, 768, sm_launch<3, 2, 8, 16, 0, SM_STREAM>},      // decoder_embed x2 (N = 768): 48x32 k8
    {34, SM_STREAM, 1024, 2, 2, 8, false, 769, 1 << 30, sm_launch<2, 2, 8, 16, 0, SM_STREAM>},  // val proj, value_out: 32x32 k8
    {35, SM_STREAM, 4096, 2, 2, 8, false, 0, 1 << 30, sm_launch<2, 2, 8, 64, 4, SM_STREAM>},   // val fc2: 32x32 k8, ring of 4
    {36, SM_STREAM, 768, 3, 2, 6, false, 0, 1 << 30, sm_launch<3, 2, 6, 12, 0, SM_STREAM>},   // dec proj / cproj x2, pos patch embed: 48x32 k6
    {37, SM_STREAM, 3072, 3, 2, 8, false, 0, 1 << 30, sm_launch<3, 2, 8, 48, 3, SM_STREAM>},  // dec fc2 x2: 48x32 k8, ring of 3
    {38, SM_STREAM, 1792, 4, 2, 7, false, 0, 1 << 30, sm_launch<4, 2, 7, 28, 0, SM_STREAM>},  // key MLP out x2: 64x32 k7
    // SCORE (the memory read's S = LN_q(q) . K_hat^T / 32 with the softmax statistics of its 32-key groups; N = bank tokens, any multiple of 4)
    {43, SM_SCORE, 1024, 2, 2, 8, false, 0, 1 << 30, sm_launch<2, 2, 8, 16, 0, SM_SCORE>},
    // PROB (round 6: the long-bank read's score stage for > 256 query rows -- a 512 x 512 frame -- without a score matrix)
    {45, SM_PROB, 1024, 4, 4, 1, false, 0, 1 << 30, bm_launch<4, 2, 4, 16, 3, SM_PROB>, 257, 1 << 30, 256, 128},
    // ---- many rows (bm_kernel<WM, WN, NF, NKB, NST, EPI>): 256x128 (8 waves) from 1536 rows on, else 128x128 / 128x64 (4 waves)
    {50, SM_ROPE, 1024, 4, 4, 1, false, 0, 1 << 30, bm_launch<4, 2, 4, 16, 3, SM_ROPE>, 1536, 1 << 30, 256, 128},     // encoder q/k/v (M = frames x 196)
    {51, SM_ROPE, 1024, 4, 4, 1, false, 0, 1 << 30, bm_launch<2, 2, 4, 16, 3, SM_ROPE>, 257, 1535, 128, 128},
    // K = 768 (decoder at 512x512: M = 1024 rows per side, both sides in one launch): the 128 x 128 tiles of a q/k/v + cross-k/v pair
    // are 480 workgroups of 96 KB LDS -- two rounds at one workgroup per CU; 256 x 128 makes it one (profiles/r05_gemm_manyrow_config3_tiles.txt:
    // 20.3 -> 14.4 us for the q/k/v half, fc1 24.3 -> 19.2).  The rule looks at rows x groups only, so both groups of a pair agree.
    {52, SM_ROPE, 768, 4, 4, 1, false, 0, 1 << 30, bm_launch<2, 2, 4, 12, 3, SM_ROPE>, 257, 1 << 30, 128, 128, 0, 2047},
    {63, SM_ROPE, 768, 4, 4, 1, false, 1536, 1 << 30, bm_launch<4, 2, 4, 12, 3, SM_ROPE>, 257, 1 << 30, 256, 128, 2048, 1 << 30},   // decoder q/k/v + cross k/v at 512x512 (N = 2304 / 1536)
    {65, SM_ROPE, 768, 4, 2, 1, false, 0, 1535, bm_launch<2, 2, 2, 12, 3, SM_ROPE>, 257, 1 << 30, 128, 64, 2048, 1 << 30},         // its cross-attention q projection (N = 768: 192 workgroups instead of 48)
    {53, SM_PACKED, 1024, 4, 4, 1, false, 0, 1 << 30, bm_launch<4, 2, 4, 16, 3, SM_PACKED>, 1536, 1 << 30, 256, 128},  // encoder fc1
    {54, SM_PACKED, 1024, 4, 4, 1, false, 0, 1 << 30, bm_launch<2, 2, 4, 16, 3, SM_PACKED>, 257, 1535, 128, 128},
    {55, SM_PACKED, 768, 4, 4, 1, false, 0, 1 << 30, bm_launch<2, 2, 4, 12, 3, SM_PACKED>, 257, 1 << 30, 128, 128, 0, 2047},
    {64, SM_PACKED, 768, 4, 4, 1, false, 0, 1 << 30, bm_launch<4, 2, 4, 12, 3, SM_PACKED>, 257, 1 << 30, 256, 128, 2048, 1 << 30},   // decoder fc1 at 512x512
    {61, SM_STREAM, 1024, 4, 4, 1, false, 0, 1 << 30, bm_launch<4, 2, 4, 16, 3, SM_STREAM>, 4096, 1 << 30, 256, 128},   // 512x512 whole-sequence encoder (M = 16 x 1024)
    {62, SM_STREAM, 4096, 4, 4, 1, false, 0, 1 << 30, bm_launch<4, 2, 4, 64, 3, SM_STREAM>, 4096, 1 << 30, 256, 128},
    {66, SM_PACKED, 1792, 4, 4, 1, true, 0, 1 << 30, bm_launch<2, 2, 4, 28, 3, SM_PACKED, true>, 257, 1 << 30, 128, 128, 0, 2047},      // key MLP hidden (split A) above 256 rows: batch 4, 512x512
    {67, SM_PACKED, 1792, 4, 4, 1, true, 0, 1 << 30, bm_launch<4, 2, 4, 28, 3, SM_PACKED, true>, 257, 1 << 30, 256, 128, 2048, 1 << 30},
    {56, SM_STREAM, 1024, 4, 2, 1, false, 0, 1 << 30, bm_launch<2, 2, 2, 16, 3, SM_STREAM>, 257, 1 << 30, 128, 64},    // encoder proj
    {57, SM_STREAM, 4096, 4, 2, 1, false, 0, 1 << 30, bm_launch<2, 2, 2, 64, 3, SM_STREAM>, 257, 1 << 30, 128, 64},    // encoder fc2
    {58, SM_STREAM, 768, 4, 2, 1, false, 0, 1 << 30, bm_launch<2, 2, 2, 12, 3, SM_STREAM>, 257, 1 << 30, 128, 64},
    {59, SM_STREAM, 3072, 4, 2, 1, false, 0, 1 << 30, bm_launch<2, 2, 2, 48, 4, SM_STREAM>, 257, 1 << 30, 128, 64},    // (ring of 4: 22.1 -> 20.9 us at 1024 x 2 x 768)
    {60, SM_STREAM, 1792, 4, 2, 1, false, 0, 1 << 30, bm_launch<2, 2, 2, 28, 3, SM_STREAM>, 257, 1 << 30, 128, 64},
};
bool sm_enabled() {
  static const bool on = [] { const char* e = getenv("SP3_LEAN_GEMM"); return !(e && e[0] == '0'); }();
  return on;
}

int sm_kind(const sp3_gemm_desc& d) {
  if (d.sm_stats_out) return (d.out_packed && d.out_bf16) ? SM_PROB : SM_SCORE;
  if (d.epi == SP3_EPI_ROPE_VT) return SM_ROPE;
  if (d.out_packed) return SM_PACKED;
  return SM_STREAM;
}

const SmInst* sm_find(const sp3_gemm_desc& d) {
  if (!sm_enabled()) return nullptr;
  if (d.wdtype != SP3_BF16 || !d.a_bf16 || !d.a_packed || !d.w_packed || d.loader != SP3_LOAD_PLAIN || d.res2 || d.relu_in ||
      d.trace || d.sm_stats || (d.alpha != 1.0f && !d.sm_stats_out) || d.f32x3)
    return nullptr;
  if (d.splitk > 1 || d.epi == SP3_EPI_PARTIAL || d.epi == SP3_EPI_PIXSHUF) return nullptr;
  if (d.batch > 2 || d.M < 1 || d.M >= 65536 || (d.ldw > 0 && d.ldw != d.K) || !d.bias) return nullptr;
  const int kind = sm_kind(d);
  if (d.dyn_n && kind != SM_SCORE && kind != SM_PROB) return nullptr;        // (device-side extent: the memory read's score instances only)
  if (kind == SM_PROB) {
    // C = fragment-order bf16 [M][ldc keys] (ldc = the bank's capacity: the layout must not move as the bank grows), statistics
    // [ldc / 64][M rounded up to 256] float2
    if (d.epi != SP3_EPI_PLAIN || d.batch > 1 || d.res1 || d.stats_out || d.c2 || d.act != SP3_ACT_NONE || d.N % 4 || d.N < 4 ||
        d.ldc % 64 || d.ldc < d.N || !d.ln_stats)
      return nullptr;
  } else
  if (kind == SM_SCORE) {
    // (sp3_gemm's own checks: plain fp32 epilogue, N % 4 == 0, one problem)
    if (d.epi != SP3_EPI_PLAIN || d.out_bf16 || d.out_packed || d.batch > 1 || d.res1 || d.stats_out || d.c2 || d.act != SP3_ACT_NONE ||
        d.N % 4 || d.N < 4 || (d.ldc & 3) || d.ldc < d.N || d.M > 256)
      return nullptr;
  } else if (kind == SM_ROPE) {
    if (!d.qkv_packed || d.rope_cols % 64 || d.N % 64 || (d.tokens & 3) || d.tokens <= 0 || d.tokens >= 65536 || d.vt_ld % 64) return nullptr;
    if (d.rope_cols < d.N && !d.vt) return nullptr;
  } else if (kind == SM_PACKED) {
    if (!d.out_bf16 || d.res1 || d.stats_out || d.c2 || d.act == SP3_ACT_RELU) return nullptr;
  } else {
    if (d.out_bf16 || d.out_packed || d.act != SP3_ACT_NONE || (d.ldc & 3) || d.ldc < d.N) return nullptr;
    if (d.res1 && d.ldr1 != d.N) return nullptr;
    if ((d.stats_out || d.c2) && d.N % 32) return nullptr;
  }
  if (d.ln_stats && (d.ln_C != d.K || d.K > 1024 || (d.K / 64) % 4)) return nullptr;
  const bool split = d.A2 != nullptr;
  if (split && (d.K1 % 64 || d.K1 <= 0 || d.K1 >= d.K)) return nullptr;
  for (const SmInst& s : kInst) {
    if (s.epi != kind || s.K != d.K || s.split != split || d.M < s.min_m || d.M > s.max_m) continue;
    const long mb = (long)d.M * (d.batch > 1 ? d.batch : 1);
    if (mb < s.min_mb || mb > s.max_mb) continue;
    if ((kind != SM_SCORE && kind != SM_PROB && d.N % s.tile_n()) || d.N < s.min_n || d.N > s.max_n) continue;
    if (!s.bm && d.ln_stats && s.MF * 16 * 4 > 64 * s.WK) continue;
    if (s.bm && kind == SM_ROPE && d.rope_cols % 32) continue;
    return &s;
  }
  return nullptr;
}

void sm_fill(SmOp& o, const sp3_gemm_desc& d, const SmInst& s) {
  const long G = d.batch > 1 ? 1 : 0;
  o.A = reinterpret_cast<const char*>(d.A);
  o.A2 = d.A2 ? reinterpret_cast<const char*>(d.A2) : o.A;
  o.nkb1 = d.A2 ? d.K1 / 64 : d.K / 64;
  o.gA2 = G * d.sb_A2;
  o.W = reinterpret_cast<const char*>(d.W);
  o.C = reinterpret_cast<char*>(d.C);
  o.bias = d.bias; o.res1 = d.res1; o.ln_stats = d.ln_stats; o.ln_s = d.ln_s;
  o.stats_out = d.stats_out; o.c2 = reinterpret_cast<char*>(d.c2); o.vt = reinterpret_cast<char*>(d.vt);
  o.gA = G * d.strideA * 2; o.gW = G * d.strideW * 2;
  o.gC = G * d.strideC * (s.epi == SM_STREAM ? 4 : 2);
  o.gbias = G * d.sb_bias; o.gres = G * (long)d.M * d.ldr1 * 4; o.gstats = G * d.sb_ln_stats; o.gs = G * d.sb_ln_s;
  o.gso = G * d.sb_stats_out; o.gc2 = G * d.sb_c2; o.gvt = G * d.sb_vt;
  o.N = d.N;
  o.alpha = d.alpha;
  if (s.epi == SM_SCORE || s.epi == SM_PROB) o.stats_out = d.sm_stats_out;
  o.nt = (d.N + s.tile_n() - 1) / s.tile_n();
  o.ntz = (o.nt + 7) / 8;
  o.ngrp = d.batch > 1 ? d.batch : 1;
  o.nzb = (o.ngrp * o.nt + 7) / 8;
  o.ldc = (int)d.ldc;
  o.rope_cols = d.rope_cols;
  o.act = d.act;
}

}  // namespace

int sp3_gemm_sm_tile(const sp3_gemm_desc& d) {
  if (d.loader == SP3_LOAD_CONV3X3) return sm_enabled() ? conv_sm_tile(d) : -1;
  if (d.loader == SP3_LOAD_SOFTMAX) return !sm_enabled() ? -1 : pv_ok(d) ? 44 : pvs_ok(d) ? 46 : -1;
  const SmInst* s = sm_find(d);
  return s ? s->tile : -1;
}

bool sp3_gemm_sm_pairs(const sp3_gemm_desc& d) {
  if (d.loader != SP3_LOAD_PLAIN) return false;
  const SmInst* s = sm_find(d);
  return s && s->epi == SM_ROPE;
}

int sp3_gemm_sm_launch(const sp3_gemm_desc& d, const sp3_gemm_desc* pair, hipStream_t stream) {
  if (d.loader == SP3_LOAD_SOFTMAX && sm_enabled() && pvs_ok(d) && !pair && (d.tile < 30 || d.tile == 46)) return pvs_dispatch(d, stream);
  if (d.loader == SP3_LOAD_SOFTMAX) {
    if (!sm_enabled() || !pv_ok(d) || pair || (d.tile >= 30 && d.tile != 44)) {
      sp3_set_error("sp3_gemm: no lean softmax-loader instance for this descriptor (tile %d, M=%d N=%d K=%d)", d.tile, d.M, d.N, d.K);
      return 1;
    }
    return pv_dispatch(d, stream);
  }
  if (d.loader == SP3_LOAD_CONV3X3) {
    const int t = sm_enabled() ? conv_sm_tile(d) : -1;
    if (t < 0 || pair || (d.tile >= 30 && d.tile != t)) {
      sp3_set_error("sp3_gemm: no lean conv3x3 instance for this descriptor (tile %d, M=%d N=%d K=%d)", d.tile, d.M, d.N, d.K);
      return 1;
    }
    return conv_sm_dispatch(d, t, stream);
  }
  const SmInst* s = sm_find(d);
  if (!s || (d.tile >= 30 && d.tile != s->tile)) {
    sp3_set_error("sp3_gemm: no lean small-M instance for this descriptor (tile %d, M=%d N=%d K=%d epi=%d)", d.tile, d.M, d.N, d.K, d.epi);
    return 1;
  }
  SmArgs a;
  sm_fill(a.op[0], d, *s);
  a.op[1] = a.op[0];
  const int G0 = d.batch > 1 ? d.batch : 1;
  int nz = s->bm ? a.op[0].nzb : a.op[0].ntz * G0;        // (many-row family: z-slots over the op's (problem, N-tile) pairs, see bm_kernel)
  a.z1 = nz;
  if (pair) {
    const SmInst* s2 = sm_find(*pair);
    if (s2 != s || pair->M != d.M || pair->tokens != d.tokens || pair->heads != d.heads || pair->vt_ld != d.vt_ld || pair->pos != d.pos ||
        pair->rope_cos != d.rope_cos || pair->rope_sin != d.rope_sin || pair->ln_eps != d.ln_eps || s->epi != SM_ROPE) {
      sp3_set_error("sp3_gemm2: the two groups do not share a lean instance");
      return 1;
    }
    sm_fill(a.op[1], *pair, *s);
    nz += s->bm ? a.op[1].nzb : a.op[1].ntz * (pair->batch > 1 ? pair->batch : 1);
  }
  a.cos = d.rope_cos; a.sin = d.rope_sin; a.pos = d.pos;
  a.M = d.M;
  a.xm = (s->bm && !pair && d.M > d.N) ? 1 : 0;     // more rows than columns: the activation panel is the larger operand
  if (a.xm) {
    a.z1 = 1 << 30;                                  // (one op: no workgroup belongs to a second group, whatever its blockIdx.z)
    const int NT = a.op[0].nt * a.op[0].ngrp, mt_ = (d.M + s->tile_m() - 1) / s->tile_m(), nzm = (mt_ + 7) / 8;
    if (NT % 8 == 0 && NT > 8 && nzm % 4 == 0) a.xm = 2;                 // (N-tiles, M-tile slots) blocks of 8 x 4 per XCD, see bm_kernel
  }
  a.mblk = 0;
  if (s->bm && !a.xm && !pair) {
    const int mt_ = (d.M + s->tile_m() - 1) / s->tile_m();
    if (mt_ >= 16 && mt_ % 8 == 0 && nz >= 2 && nz <= 4) a.mblk = 1;
  }
  a.rb_max = (d.M + 15) / 16 - 1;
  a.tokens = d.tokens > 0 ? d.tokens : 1;
  a.heads = d.heads;
  a.vt_ld = (int)d.vt_ld;
  a.tok_magic = (unsigned)((1ull << 32) / (unsigned)a.tokens + 1);
  a.ln_eps = d.ln_eps;
  a.dyn = d.dyn_n;
  const int mt = (d.M + s->tile_m() - 1) / s->tile_m();
  return s->launch(a, mt, nz, stream);
}
