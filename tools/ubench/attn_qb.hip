// Long-sequence packed attention: several 16-query blocks per workgroup (NEXT-ROUND CANDIDATE, not part of libspann3r_hip.so).
//
// Why: attention_packed_kernel (spann3r_amd/csrc/attention.hip) gives a workgroup 16 query rows of one head and streams that head's
// whole K and V^T through its four waves.  At 1024 tokens (config 3: 2 x 16 heads x 64 query blocks = 2048 workgroups x 256 KB) that
// is 524 MB through the CUs' vector-memory paths per launch = 2 MB per CU, ~29 us at the ~70 KB/us per CU the lean GEMMs measured --
// the launch takes 27.7 us: it is bound by the K / V bytes per query, not by MFMA (3.4 us) or the softmax VALU work (~10 us).
// Here a workgroup owns QB query blocks; a wave keeps its K and V^T fragments of a 64-key tile in registers and runs them against
// all QB blocks (one block's scores at a time: 16 registers), so the K / V bytes per query drop by QB.  Every (wave, query block)
// pair does exactly the arithmetic of the product kernel in the same order: the outputs must be BIT-IDENTICAL to it, which is
// what this program checks before it times both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ispann3r_amd/csrc -Iinclude tools/ubench/attn_qb.hip spann3r_amd/csrc/error.cpp \
//         -Xclang -target-feature -Xclang -packed-fp32-ops -o tools/ubench/attn_qb.bin && tools/ubench/attn_qb.bin
#include "../../spann3r_amd/csrc/attention.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {

template <int QB>
__global__ __launch_bounds__(256) void attention_packed_qb_kernel(const __bf16* __restrict__ QP, int q_cols, int q_col0, int npad_q,
                                                                  const __bf16* __restrict__ KP, int k_cols, int k_col0, int npad_k,
                                                                  const __bf16* __restrict__ VTP, void* __restrict__ O, int64_t ldo,
                                                                  int out_bf16, int out_packed, int heads, int Nq, int Nk, float scale,
                                                                  int o_group, int o_group_rows) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* sh_o = sh;                                  // [wave][qb][db][lane][4]
  float* sh_m = sh + 4 * QB * 4 * 64 * 4;            // [wave][qb][lane]
  float* sh_l = sh_m + 4 * QB * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * 16 * QB;
  const int last_qrow = npad_q - 16;                 // blocks past the padded rows re-read the last block (their results are dropped)
  KFrag qf[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int r0 = q0 + 16 * qb < last_qrow ? q0 + 16 * qb : last_qrow;
    qf[qb] = load_frag(QP, b * npad_q + r0 + ql, q_col0 + h * 64 + 16 * g, q_cols);
  }
  const int krow0 = b * npad_k;
  const int kcol = k_col0 + h * 64 + 16 * g;
  const int64_t nU = npad_k >> 5;
  const __bf16* vbase = VTP + ((int64_t)(b * heads + h) * nU * 4 * 64 + lane) * 8;
  const int ntiles = (Nk + 63) >> 6;
  const float sl2 = scale * 1.4426950408889634f;

  f32x4 o[QB][4];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY; l_run[qb] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[qb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  if (wave < ntiles) {
    KFrag kc[4], kn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) kc[t] = load_frag(KP, krow0 + (wave << 6) + 16 * t + ql, kcol, k_cols);
    for (int tile = wave; tile < ntiles; tile += 4) {
      const int kb = tile << 6;
      bf16x8 vv[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int db = 0; db < 4; ++db)
          vv[u][db] = *reinterpret_cast<const bf16x8*>(vbase + (((int64_t)(2 * tile + u)) * 4 + db) * 64 * 8);
      const int nkb = (tile + 4 < ntiles ? tile + 4 : tile) << 6;
#pragma unroll
      for (int t = 0; t < 4; ++t) kn[t] = load_frag(KP, krow0 + nkb + 16 * t + ql, kcol, k_cols);

#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        f32x4 s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[t].v[0], qf[qb].v[0], s[t], 0, 0, 0);
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[t].v[1], qf[qb].v[1], s[t], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kb + 16 * t + 4 * g + r;
            const float v = key < Nk ? s[t][r] * sl2 : -INFINITY;
            s[t][r] = v;
            mx = fmaxf(mx, v);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run[qb], mx);
        const float alpha = exp2f(m_run[qb] - m_new);
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = exp2f(s[t][r] - m_new);
            s[t][r] = e;
            ps += e;
          }
        l_run[qb] = l_run[qb] * alpha + ps;
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
#pragma unroll
      for (int t = 0; t < 4; ++t) kc[t] = kn[t];
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      l_run[qb] += __shfl_xor(l_run[qb], 16);
      l_run[qb] += __shfl_xor(l_run[qb], 32);
    }
  }
  // ---- merge the four per-wave states of every query block (wave w merges d-block w), as the product kernel does for its one block
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
      const float sc = exp2f(sh_m[(w * QB + qb) * 64 + lane] - M);
      L += sh_l[(w * QB + qb) * 64 + lane] * sc;
      const float4 ow = *reinterpret_cast<const float4*>(sh_o + ((((w * QB + qb) * 4 + db) * 64 + lane) << 2));
      acc.x += ow.x * sc; acc.y += ow.y * sc; acc.z += ow.z * sc; acc.w += ow.w * sc;
    }
    const float inv = 1.0f / L;
    const int q = q0 + 16 * qb + ql;
    if (q < Nq) {
      const int row = o_group > 0 ? (b / o_group) * o_group_rows + (b % o_group) * Nq + q : b * Nq + q;
      const int col = h * 64 + db * 16 + 4 * g;
      const int64_t off = out_packed ? packed_off(row, col, heads * 64, out_bf16 != 0) : (int64_t)row * ldo + col;
      if (out_bf16) {
        bf16x4 ob;
        ob[0] = (__bf16)(acc.x * inv); ob[1] = (__bf16)(acc.y * inv); ob[2] = (__bf16)(acc.z * inv); ob[3] = (__bf16)(acc.w * inv);
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(O) + off) = ob;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(O) + off) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
      }
    }
  }
}

// v2 of the product kernel (same work split: 16 queries per workgroup, waves split the key tiles) with the softmax's VALU work cut
// down -- the ISA of the product loop is 16 MFMAs next to ~285 VALU instructions per 64-key tile (17 exp2f() calls expand to
// v_exp_f32 + range test + two selects + v_ldexp for denormal results, 32 compare / select pairs mask keys >= Nk on EVERY tile,
// 16 multiplies apply the scale): here exp2 is the bare v_exp_f32 (probabilities below 2^-126 flush to zero), the key mask runs on
// the last tile only, and the scale rides in the exponent's FMA (max taken on the raw scores: scale > 0).
// PIN = 1 (v3): a scheduling barrier behind the tile's loads.  The ISA of the product loop (and of v2) shows that the compiler SINKS the
// next tile's K loads from the top of the iteration to its end -- after loop rotation they sit right in front of the MFMAs that consume
// them (`s_waitcnt vmcnt(7)` 50 instructions behind the loads): the software prefetch the source spells out does not exist in the
// binary, every tile starts by waiting one L2 round trip for its K fragments.  The barrier keeps the loads where the source has them.
template <int PIN>
__global__ __launch_bounds__(256) void attention_packed_v2_kernel(const __bf16* __restrict__ QP, int q_cols, int q_col0, int npad_q,
                                                                  const __bf16* __restrict__ KP, int k_cols, int k_col0, int npad_k,
                                                                  const __bf16* __restrict__ VTP, void* __restrict__ O, int64_t ldo,
                                                                  int out_bf16, int out_packed, int heads, int Nq, int Nk, float scale,
                                                                  int o_group, int o_group_rows) {
  __shared__ float sh_o[4][4][64][4];
  __shared__ float sh_m[4][64], sh_l[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * 16;
  const KFrag qf = load_frag(QP, b * npad_q + q0 + ql, q_col0 + h * 64 + 16 * g, q_cols);
  const int krow0 = b * npad_k;
  const int kcol = k_col0 + h * 64 + 16 * g;
  const int64_t nU = npad_k >> 5;
  const __bf16* vbase = VTP + ((int64_t)(b * heads + h) * nU * 4 * 64 + lane) * 8;
  const int ntiles = (Nk + 63) >> 6;
  const float sl2 = scale * 1.4426950408889634f;
  f32x4 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  if (wave < ntiles) {
    KFrag kc[4], kn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) kc[t] = load_frag(KP, krow0 + (wave << 6) + 16 * t + ql, kcol, k_cols);
    for (int tile = wave; tile < ntiles; tile += 4) {
      const int kb = tile << 6;
      bf16x8 vv[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int db = 0; db < 4; ++db)
          vv[u][db] = *reinterpret_cast<const bf16x8*>(vbase + (((int64_t)(2 * tile + u)) * 4 + db) * 64 * 8);
      const int nkb = (tile + 4 < ntiles ? tile + 4 : tile) << 6;
#pragma unroll
      for (int t = 0; t < 4; ++t) kn[t] = load_frag(KP, krow0 + nkb + 16 * t + ql, kcol, k_cols);
      if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
      f32x4 s[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[t].v[0], qf.v[0], s[t], 0, 0, 0);
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[t].v[1], qf.v[1], s[t], 0, 0, 0);
      }
      if (kb + 64 > Nk) {                      // the ragged last tile (wave-uniform): keys past Nk drop out of the softmax
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kb + 16 * t + 4 * g + r >= Nk) s[t][r] = -INFINITY;
      }
      float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
      for (int t = 1; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx * sl2);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], sl2, -m_new));
          s[t][r] = e;
          ps += e;
        }
      l_run = l_run * alpha + ps;
      m_run = m_new;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[db][r] *= alpha;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x8 pb;
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[j] = (__bf16)s[2 * u + (j >> 2)][j & 3];
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vv[u][db], pb, o[db], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) kc[t] = kn[t];
    }
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
  }
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
  const int db = wave;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float sc = __builtin_amdgcn_exp2f(sh_m[w][lane] - M);
    L += sh_l[w][lane] * sc;
    const float4 ow = *reinterpret_cast<const float4*>(&sh_o[w][db][lane][0]);
    acc.x += ow.x * sc; acc.y += ow.y * sc; acc.z += ow.z * sc; acc.w += ow.w * sc;
  }
  const float inv = 1.0f / L;
  if (q0 + ql < Nq) {
    const int row = o_group > 0 ? (b / o_group) * o_group_rows + (b % o_group) * Nq + q0 + ql : b * Nq + q0 + ql;
    const int col = h * 64 + db * 16 + 4 * g;
    const int64_t off = out_packed ? packed_off(row, col, heads * 64, out_bf16 != 0) : (int64_t)row * ldo + col;
    if (out_bf16) {
      bf16x4 ob;
      ob[0] = (__bf16)(acc.x * inv); ob[1] = (__bf16)(acc.y * inv); ob[2] = (__bf16)(acc.z * inv); ob[3] = (__bf16)(acc.w * inv);
      *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(O) + off) = ob;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(O) + off) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
  }
}


// v5: what the counters of profiles/r05_attn_long_pmc.md ask for -- more independent work per wave at (about) the product kernel's
// register budget: QB query blocks per wave on the same K / V^T registers (independent softmax chains the compiler interleaves), v2's
// trimmed softmax, tree-shaped max / sum reductions (4 chains of 4 instead of one of 16), and the two K buffers used alternately by
// a loop unrolled x2 instead of a 32-register copy per tile.  MINW: minimum waves per SIMD for the register allocator; PF: prefetch
// the next tile's K fragments (0: request K and V of a tile together at its top and rely on the other waves).
template <int QB, int MINW, int PF>
__global__ __launch_bounds__(256, MINW) void attention_packed_v5_kernel(const __bf16* __restrict__ QP, int q_cols, int q_col0, int npad_q,
                                                                        const __bf16* __restrict__ KP, int k_cols, int k_col0, int npad_k,
                                                                        const __bf16* __restrict__ VTP, void* __restrict__ O, int64_t ldo,
                                                                        int out_bf16, int out_packed, int heads, int Nq, int Nk, float scale,
                                                                        int o_group, int o_group_rows) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* sh_o = sh;                                  // [wave][qb][db][lane][4]
  float* sh_m = sh + 4 * QB * 4 * 64 * 4;            // [wave][qb][lane]
  float* sh_l = sh_m + 4 * QB * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ql = lane & 15;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * 16 * QB;
  const int last_qrow = npad_q - 16;
  KFrag qf[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int r0 = q0 + 16 * qb < last_qrow ? q0 + 16 * qb : last_qrow;
    qf[qb] = load_frag(QP, b * npad_q + r0 + ql, q_col0 + h * 64 + 16 * g, q_cols);
  }
  const int ntiles = (Nk + 63) >> 6;
  const float sl2 = scale * 1.4426950408889634f;
  // K fragment of key block t of tile `tile`: rows krow0 + 64 tile + 16 t + ql -> fragment blocks are 16 rows x 64 columns = 2 KB, a row block's
  // column blocks are contiguous: one pointer per lane, constant strides per tile / per key block
  const int nkbk = (k_cols + 63) >> 6;
  const __bf16* kbase = KP + packed_off(b * npad_k + ql, k_col0 + h * 64 + 16 * g, k_cols, true);
  const int64_t kblk = (int64_t)nkbk * 1024;          // elements per 16-row block
  const int64_t nU = npad_k >> 5;
  const __bf16* vbase = VTP + ((int64_t)(b * heads + h) * nU * 4 * 64 + lane) * 8;
  auto load_k = [&](KFrag (&kf)[4], int tile) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const __bf16* p = kbase + (int64_t)(4 * tile + t) * kblk;
      kf[t].v[0] = *reinterpret_cast<const bf16x8*>(p);
      kf[t].v[1] = *reinterpret_cast<const bf16x8*>(p + 64 * 8);
    }
  };
  f32x4 o[QB][4];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY; l_run[qb] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[qb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto tile_body = [&](const KFrag (&kc)[4], const bf16x8 (&vv)[2][4], int tile) {
    const int kb = tile << 6;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      f32x4 s[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[t].v[0], qf[qb].v[0], s[t], 0, 0, 0);
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[t].v[1], qf[qb].v[1], s[t], 0, 0, 0);
      }
      if (kb + 64 > Nk) {
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
      const float m_new = fmaxf(m_run[qb], mx * sl2);
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
  };
  auto load_v = [&](bf16x8 (&vv)[2][4], int tile) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int db = 0; db < 4; ++db) vv[u][db] = *reinterpret_cast<const bf16x8*>(vbase + (((int64_t)(2 * tile + u)) * 4 + db) * 64 * 8);
  };
  if (wave < ntiles) {
    if constexpr (PF) {
      KFrag ka[4], kb_[4];
      load_k(ka, wave);
      for (int tile = wave; tile < ntiles; tile += 8) {
        {
          bf16x8 vv[2][4];
          load_v(vv, tile);
          load_k(kb_, tile + 4 < ntiles ? tile + 4 : tile);
          tile_body(ka, vv, tile);
        }
        if (tile + 4 < ntiles) {
          bf16x8 vv[2][4];
          load_v(vv, tile + 4);
          load_k(ka, tile + 8 < ntiles ? tile + 8 : tile + 4);
          tile_body(kb_, vv, tile + 4);
        }
      }
    } else {
      for (int tile = wave; tile < ntiles; tile += 4) {
        KFrag kc[4];
        bf16x8 vv[2][4];
        load_k(kc, tile);
        load_v(vv, tile);
        tile_body(kc, vv, tile);
      }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      l_run[qb] += __shfl_xor(l_run[qb], 16);
      l_run[qb] += __shfl_xor(l_run[qb], 32);
    }
  }
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
      const float sc = __builtin_amdgcn_exp2f(sh_m[(w * QB + qb) * 64 + lane] - M);
      L += sh_l[(w * QB + qb) * 64 + lane] * sc;
      const float4 ow = *reinterpret_cast<const float4*>(sh_o + ((((w * QB + qb) * 4 + db) * 64 + lane) << 2));
      acc.x += ow.x * sc; acc.y += ow.y * sc; acc.z += ow.z * sc; acc.w += ow.w * sc;
    }
    const float inv = 1.0f / L;
    const int q = q0 + 16 * qb + ql;
    if (q < Nq) {
      const int row = o_group > 0 ? (b / o_group) * o_group_rows + (b % o_group) * Nq + q : b * Nq + q;
      const int col = h * 64 + db * 16 + 4 * g;
      const int64_t off = out_packed ? packed_off(row, col, heads * 64, out_bf16 != 0) : (int64_t)row * ldo + col;
      if (out_bf16) {
        bf16x4 ob;
        ob[0] = (__bf16)(acc.x * inv); ob[1] = (__bf16)(acc.y * inv); ob[2] = (__bf16)(acc.z * inv); ob[3] = (__bf16)(acc.w * inv);
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(O) + off) = ob;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(O) + off) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
      }
    }
  }
}

template <int QB, int MINW, int PF>
void launch_v5(const __bf16* q, const __bf16* k, const __bf16* vt, void* o, int cols, int npad_q, int npad_k, int B, int heads, int Nq, int Nk,
               int out_bf16, int out_packed) {
  const int lds = (4 * QB * 4 * 64 * 4 + 2 * 4 * QB * 64) * 4;
  auto kern = attention_packed_v5_kernel<QB, MINW, PF>;
  static bool raised = false;
  if (!raised && lds > 64 * 1024) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); raised = true; }
  hipLaunchKernelGGL(kern, dim3((Nq + 16 * QB - 1) / (16 * QB), heads, B), dim3(256), lds, 0, q, cols, 0, npad_q, k, cols, 0, npad_k, vt, o,
                     (int64_t)cols, out_bf16, out_packed, heads, Nq, Nk, 0.125f, 0, 0);
}

template <int PIN>
void launch_v2(const __bf16* q, const __bf16* k, const __bf16* vt, void* o, int cols, int npad_q, int npad_k, int B, int heads, int Nq, int Nk,
               int out_bf16, int out_packed) {
  hipLaunchKernelGGL(attention_packed_v2_kernel<PIN>, dim3((Nq + 15) / 16, heads, B), dim3(256), 0, 0, q, cols, 0, npad_q, k, cols, 0, npad_k, vt, o,
                     (int64_t)cols, out_bf16, out_packed, heads, Nq, Nk, 0.125f, 0, 0);
}

template <int QB>
void launch_qb(const __bf16* q, const __bf16* k, const __bf16* vt, void* o, int cols, int npad_q, int npad_k, int B, int heads, int Nq, int Nk,
               int out_bf16, int out_packed) {
  const int lds = (4 * QB * 4 * 64 * 4 + 2 * 4 * QB * 64) * 4;
  auto kern = attention_packed_qb_kernel<QB>;
  static bool raised = false;
  if (!raised && lds > 64 * 1024) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); raised = true; }
  hipLaunchKernelGGL(kern, dim3((Nq + 16 * QB - 1) / (16 * QB), heads, B), dim3(256), lds, 0, q, cols, 0, npad_q, k, cols, 0, npad_k, vt, o,
                     (int64_t)cols, out_bf16, out_packed, heads, Nq, Nk, 0.125f, 0, 0);
}

void launch_ref(const __bf16* q, const __bf16* k, const __bf16* vt, void* o, int cols, int npad_q, int npad_k, int B, int heads, int Nq, int Nk,
                int out_bf16, int out_packed) {
  // (the product entry point: since round 5 it runs the v5 loop -- QB = 2 from 512 query rows on; the round-4 loop survives here as v2's ancestor)
  sp3_attention_packed(q, cols, 0, npad_q, k, cols, 0, npad_k, vt, o, (int64_t)cols, out_bf16, out_packed, B, heads, Nq, Nk, 0.125f, 0, 0, nullptr);
}

template <typename F>
float time_us(F&& f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  return best * 1000.f / reps;
}

}  // namespace

int main() {
  // B, heads, Nq (= Nk): config 3 encoder (2 x 16 x 1024 in the grouped launches), its decoder (12 heads), ragged lengths, config 2
  const int cases[][3] = {{2, 16, 1024}, {2, 12, 1024}, {1, 16, 1000}, {2, 12, 777}, {2, 12, 196}, {10, 16, 196}};
  printf("%-18s %10s %10s %10s %10s %10s %10s   v3 TFLOP/s   QB bit-identical to the product kernel   v3 max |diff| / max |out| (fp32 rows)\n", "B,heads,tokens",
         "product us", "QB=2 us", "QB=3 us", "QB=4 us", "v2 us", "v3 us");
  for (auto& c : cases) {
    const int B = c[0], heads = c[1], N = c[2];
    const int npad_q = (N + 15) / 16 * 16, npad_k = (N + 63) / 64 * 64, cols = heads * 64;
    const size_t nq = (size_t)B * npad_q * cols, nk = (size_t)B * npad_k * cols, nv = (size_t)B * heads * (npad_k / 32) * 4 * 64 * 8;
    const size_t no = (size_t)B * npad_q * cols + 64 * cols;
    std::vector<unsigned short> hq(nq), hk(nk), hv(nv);
    srand(1234 + N);
    auto rnd = [] { return (unsigned short)(((rand() & 1) << 15) | (0x3c00 + (rand() % 0x380))); };   // bf16 magnitudes in [0.0078, ~1)
    for (auto& x : hq) x = rnd();
    for (auto& x : hk) x = rnd();
    for (auto& x : hv) x = rnd();
    __bf16 *q, *k, *vt;
    float *o_ref, *o_new;
    CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&k, nk * 2)); CK(hipMalloc(&vt, nv * 2)); CK(hipMalloc(&o_ref, no * 4)); CK(hipMalloc(&o_new, no * 4));
    CK(hipMemcpy(q, hq.data(), nq * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(k, hk.data(), nk * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(vt, hv.data(), nv * 2, hipMemcpyHostToDevice));
    bool same = true;
    std::vector<float> a(no), b2(no);
    for (int mode = 0; mode < 2; ++mode) {            // fp32 row-major output, then packed bf16 output
      const int obf = mode, opk = mode;
      CK(hipMemset(o_ref, 0, no * 4));
      launch_ref(q, k, vt, o_ref, cols, npad_q, npad_k, B, heads, N, N, obf, opk);
      CK(hipMemcpy(a.data(), o_ref, no * 4, hipMemcpyDeviceToHost));
      for (int qb = 2; qb <= 4; ++qb) {
        CK(hipMemset(o_new, 0, no * 4));
        if (qb == 2) launch_qb<2>(q, k, vt, o_new, cols, npad_q, npad_k, B, heads, N, N, obf, opk);
        if (qb == 3) launch_qb<3>(q, k, vt, o_new, cols, npad_q, npad_k, B, heads, N, N, obf, opk);
        if (qb == 4) launch_qb<4>(q, k, vt, o_new, cols, npad_q, npad_k, B, heads, N, N, obf, opk);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(b2.data(), o_new, no * 4, hipMemcpyDeviceToHost));
        if (memcmp(a.data(), b2.data(), no * 4) != 0) { same = false; printf("  MISMATCH: QB=%d mode=%d\n", qb, mode); }
      }
    }
    // v2 against the product kernel, fp32 rows: not bit-identical by design (FMA in the exponent, flushed denormal probabilities)
    CK(hipMemset(o_ref, 0, no * 4)); CK(hipMemset(o_new, 0, no * 4));
    launch_ref(q, k, vt, o_ref, cols, npad_q, npad_k, B, heads, N, N, 0, 0);
    launch_v2<1>(q, k, vt, o_new, cols, npad_q, npad_k, B, heads, N, N, 0, 0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(a.data(), o_ref, no * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b2.data(), o_new, no * 4, hipMemcpyDeviceToHost));
    double dmax = 0.0, amax = 0.0;
    for (size_t i = 0; i < (size_t)B * N * cols; ++i) {
      const double d = fabs((double)a[i] - (double)b2[i]);
      dmax = d > dmax ? d : dmax;
      amax = fabs((double)a[i]) > amax ? fabs((double)a[i]) : amax;
      if (!(b2[i] == b2[i])) dmax = 1e30;
    }
    const int reps = 20;
    const float t0 = time_us([&] { launch_ref(q, k, vt, o_ref, cols, npad_q, npad_k, B, heads, N, N, 1, 1); }, reps);
    const float t2 = time_us([&] { launch_qb<2>(q, k, vt, o_new, cols, npad_q, npad_k, B, heads, N, N, 1, 1); }, reps);
    const float t3 = time_us([&] { launch_qb<3>(q, k, vt, o_new, cols, npad_q, npad_k, B, heads, N, N, 1, 1); }, reps);
    const float t4 = time_us([&] { launch_qb<4>(q, k, vt, o_new, cols, npad_q, npad_k, B, heads, N, N, 1, 1); }, reps);
    const float t5 = time_us([&] { launch_v2<0>(q, k, vt, o_new, cols, npad_q, npad_k, B, heads, N, N, 1, 1); }, reps);
    const float t6 = time_us([&] { launch_v2<1>(q, k, vt, o_new, cols, npad_q, npad_k, B, heads, N, N, 1, 1); }, reps);
    {
      struct V { const char* n; void (*f)(const __bf16*, const __bf16*, const __bf16*, void*, int, int, int, int, int, int, int, int, int); };
      const V vs[] = {{"v5 qb1 w4 pf1", launch_v5<1, 4, 1>}, {"v5 qb1 w3 pf1", launch_v5<1, 3, 1>}, {"v5 qb1 w4 pf0", launch_v5<1, 4, 0>},
                      {"v5 qb2 w2 pf1", launch_v5<2, 2, 1>}, {"v5 qb2 w3 pf1", launch_v5<2, 3, 1>}, {"v5 qb2 w4 pf1", launch_v5<2, 4, 1>},
                      {"v5 qb2 w3 pf0", launch_v5<2, 3, 0>}, {"v5 qb2 w4 pf0", launch_v5<2, 4, 0>}, {"v5 qb4 w2 pf0", launch_v5<4, 2, 0>}};
      for (const V& v : vs) {
        CK(hipMemset(o_new, 0, no * 4));
        v.f(q, k, vt, o_new, cols, npad_q, npad_k, B, heads, N, N, 0, 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(b2.data(), o_new, no * 4, hipMemcpyDeviceToHost));
        double dm = 0.0;
        for (size_t i = 0; i < (size_t)B * N * cols; ++i) { const double d = fabs((double)a[i] - (double)b2[i]); dm = d > dm ? d : dm; if (!(b2[i] == b2[i])) dm = 1e30; }
        const float t = time_us([&] { v.f(q, k, vt, o_new, cols, npad_q, npad_k, B, heads, N, N, 1, 1); }, reps);
        printf("    %-16s %8.2f us  (%5.1f TFLOP/s)  max |diff| / max |out| %.2e\n", v.n, t, 4.0 * B * heads * (double)N * N * 64 / t / 1e6, dm / amax);
      }
    }
    char name[64];
    snprintf(name, sizeof name, "%d,%d,%d", B, heads, N);
    printf("%-18s %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f   %8.1f     %s                                    %.2e\n", name, t0, t2, t3, t4, t5, t6,
           4.0 * B * heads * (double)N * N * 64 / t6 / 1e6, same ? "yes" : "NO", dmax / amax);
    CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(vt)); CK(hipFree(o_ref)); CK(hipFree(o_new));
  }
  return 0;
}
