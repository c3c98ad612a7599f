// Spatial-memory kernels (reference: spann3r/model.py:97-210).
//
// The bank is a static-capacity arena owned by the host side (spann3r_amd/model.py):
//   K_raw [cap,1024] fp32, V_raw [cap,1024] fp32 (what the reference calls mem_k / mem_v),
//   K_hat' = gamma_q (.) LN_k(K_raw) as a FRAGMENT-ORDER [cap, 1024] matrix (the W operand of the S GEMM) and
//   V_hat^T = LN_v(V_raw)^T as a fragment-order [1024, cap] matrix (the W operand of the P.V GEMM), in the MFMA dtype,
//   s_bank / b_bank [cap] fp32: the per-token constants that fold LN_q into the S GEMM, mem_attn / mem_count [cap] fp32.
// LayerNorm is row-wise, so normalising ONCE at write time (sp3_bank_write, one launch per stored frame) is exactly what
// the reference recomputes over the whole bank at every read (model.py:154,174).  A read is then
//   S = LN_q(q) . K_hat^T / 32   (sp3_gemm: raw q in fragment order, LN_q folded through s_bank / b_bank, alpha)
//   P = softmax / threshold / renormalise   (sp3_softmax_thresh: fragment-order copy for the next GEMM)
//   out = P . V_hat + q          (sp3_gemm, residual)   ;  mem_attn += colsum(P)  (sp3_colsum_accum / _packed)
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float block_reduce_max(float v, float* sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, sh[i]);
  return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;
}

// One block per row.  softmax over [0,M); optional threshold + renormalise.  Outputs, both optional: P fp32 row-major
// (zero-filled over [M,Mpad)) and Pk = the same probabilities as a fragment-order operand [rows, Kp] (Kp = M rounded up to
// 64, zero filled) in bf16 or fp32: 8 (4) consecutive probabilities = one 16-byte store.
template <typename TP>
__global__ __launch_bounds__(256) void softmax_thresh_kernel(const float* __restrict__ S, float* __restrict__ P, int64_t ld,
                                                             int64_t strideS, int M, int Mpad, float thresh,
                                                             TP* __restrict__ Pk, int Kp, int64_t stridePk) {
  __shared__ float sh[8];
  const float* s = S + (int64_t)blockIdx.y * strideS + (int64_t)blockIdx.x * ld;
  // one pass for max and sum (online softmax per thread, merged across the block): a long bank row (196 KB at 49152
  // tokens) does not stay in cache between passes
  float mx = -INFINITY, sum = 0.f;
  for (int j = threadIdx.x; j < M; j += 256) {
    const float v = s[j];
    if (v > mx) { sum = sum * expf(mx - v) + 1.0f; mx = v; }
    else sum += expf(v - mx);
  }
  const float gmx = block_reduce_max(mx, sh);
  sum = block_reduce_sum(mx == -INFINITY ? 0.f : sum * expf(mx - gmx), sh);
  mx = gmx;
  float inv = 1.0f / sum, rk = 1.0f;
  if (thresh > 0.f) {
    float kept = 0.f;
    for (int j = threadIdx.x; j < M; j += 256) {
      float v = expf(s[j] - mx) * inv;
      v = v < thresh ? 0.f : v;
      kept += v;
    }
    rk = block_reduce_sum(kept, sh);
  }
  auto prob = [&](int j) -> float {
    float v = expf(s[j] - mx) * inv;
    if (thresh > 0.f) v = (v < thresh ? 0.f : v) / rk;
    return v;
  };
  if (P) {
    float* p = P + (int64_t)blockIdx.y * strideS + (int64_t)blockIdx.x * ld;
    for (int j = threadIdx.x; j < M; j += 256) p[j] = prob(j);
    for (int j = M + threadIdx.x; j < Mpad; j += 256) p[j] = 0.f;
  }
  if (Pk) {
    constexpr int E = 16 / (int)sizeof(TP);               // elements per 16-byte store
    TP* pk = Pk + (int64_t)blockIdx.y * stridePk;
    const int row = blockIdx.x;
    for (int j0 = threadIdx.x * E; j0 < Kp; j0 += 256 * E) {
      TP o[E];
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = (TP)((j0 + e) < M ? prob(j0 + e) : 0.f);
      typedef __attribute__((ext_vector_type(4))) unsigned int u4;
      *reinterpret_cast<u4*>(pk + packed_off(row, j0, Kp, sizeof(TP) == 2)) = *reinterpret_cast<const u4*>(o);
    }
  }
}

// ---- long banks: the probabilities of a read as the P.V GEMM's fragment-order operand, in two streaming launches.
// softmax_thresh_kernel above walks a row three to four times with libm expf and scatters 16-byte pieces 256 bytes apart; at
// 1024 queries x 50176 bank tokens that was 342 us, the largest launch of the read.  Here: (1) one workgroup per row reduces
// (max, 1 / sum exp, 1 / kept mass) with float4 loads and v_exp_f32; (2) a workgroup owns two 2 KB fragment blocks (16 rows x
// one k-block each): a thread reads the 32 bytes of scores behind its 16-byte piece, and a wave stores ONE CONTIGUOUS KILOBYTE.
__global__ __launch_bounds__(256) void softmax_rowstat_kernel(const float* __restrict__ S, int64_t ld, int64_t strideS, int rows, int M, float thresh,
                                                              float4* __restrict__ rowstat) {
  __shared__ float sh[8];
  const float* s = S + (int64_t)blockIdx.y * strideS + (int64_t)blockIdx.x * ld;
  const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(S) & 15) == 0) && ((strideS & 3) == 0);
  const int M4 = vec ? (M & ~3) : 0;
  float mx = -INFINITY, sum = 0.f;
  auto upd = [&](float v) {
    if (v > mx) { sum = sum * __expf(mx - v) + 1.0f; mx = v; }
    else sum += __expf(v - mx);
  };
  for (int j = threadIdx.x * 4; j < M4; j += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(s + j);
    const float m4 = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    if (m4 > mx) { sum *= __expf(mx - m4); mx = m4; }                // (first element: 0 * exp(-inf) = 0)
    sum += (__expf(v.x - mx) + __expf(v.y - mx)) + (__expf(v.z - mx) + __expf(v.w - mx));
  }
  for (int j = M4 + threadIdx.x; j < M; j += 256) upd(s[j]);
  const float gmx = block_reduce_max(mx, sh);
  sum = block_reduce_sum(mx == -INFINITY ? 0.f : sum * __expf(mx - gmx), sh);
  const float inv = 1.0f / sum;
  float rk = 1.0f;
  if (thresh > 0.f) {
    float kept = 0.f;
    for (int j = threadIdx.x * 4; j < M4; j += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(s + j);
      const float p[4] = {__expf(v.x - gmx) * inv, __expf(v.y - gmx) * inv, __expf(v.z - gmx) * inv, __expf(v.w - gmx) * inv};
#pragma unroll
      for (int e = 0; e < 4; ++e) kept += p[e] < thresh ? 0.f : p[e];
    }
    for (int j = M4 + threadIdx.x; j < M; j += 256) { const float p = __expf(s[j] - gmx) * inv; kept += p < thresh ? 0.f : p; }
    rk = 1.0f / block_reduce_sum(kept, sh);
  }
  if (threadIdx.x == 0) rowstat[(int64_t)blockIdx.y * rows + blockIdx.x] = make_float4(gmx, inv, rk, 0.f);
}

template <typename TP>
__global__ __launch_bounds__(256) void softmax_pack_kernel(const float* __restrict__ S, int64_t ld, int64_t strideS, int rows, int M, int Kp,
                                                           float thresh, const float4* __restrict__ rowstat, TP* __restrict__ Pk,
                                                           int64_t stridePk) {
  constexpr int E = 16 / (int)sizeof(TP), KB = 8 * E;
  const int t = threadIdx.x & 127, kb = blockIdx.x * 2 + (threadIdx.x >> 7), nkb = Kp / KB;
  if (kb >= nkb) return;
  const int h = t >> 6, g = (t >> 4) & 3, r = t & 15;
  const int row = blockIdx.y * 16 + r;
  const int k0 = kb * KB + g * 2 * E + h * E;
  float p[E];
#pragma unroll
  for (int e = 0; e < E; ++e) p[e] = 0.f;
  if (row < rows && k0 < M) {
    const float* s = S + (int64_t)blockIdx.z * strideS + (int64_t)row * ld + k0;
    const float4 st = rowstat[(int64_t)blockIdx.z * rows + row];
    float v[E];
    if (k0 + E <= M && ((reinterpret_cast<uintptr_t>(s) & 15) == 0)) {
#pragma unroll
      for (int q = 0; q < E / 4; ++q) {
        const float4 x = reinterpret_cast<const float4*>(s)[q];
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = (k0 + e) < M ? s[e] : -INFINITY;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float q = __expf(v[e] - st.x) * st.y;
      if (thresh > 0.f) q = (q < thresh ? 0.f : q) * st.z;
      p[e] = (k0 + e) < M ? q : 0.f;
    }
  }
  TP o[E];
#pragma unroll
  for (int e = 0; e < E; ++e) o[e] = (TP)p[e];
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  TP* dst = Pk + (int64_t)blockIdx.z * stridePk + (((int64_t)blockIdx.y * nkb + kb) * 128 + t) * E;
  *reinterpret_cast<u4*>(dst) = *reinterpret_cast<const u4*>(o);
}

// mem_attn[j] += sum_r P[r, j] from the fragment-order probabilities [rows, Kp]: one workgroup per 64-column k-block
// walks the row blocks (2 KB each, contiguous), every thread keeps the partial sums of its 16 bytes, the 16 rows x 2
// halves of a column meet in LDS in a fixed order (one writer per column: deterministic, no atomics).
template <typename TP>
__global__ __launch_bounds__(256) void colsum_packed_kernel(const TP* __restrict__ Pk, int rows, int M, int Kp,
                                                            float* __restrict__ mem_attn) {
  constexpr int E = 16 / (int)sizeof(TP), KB = 8 * E, TPB = 2048 / 16;      // 128 threads cover one 2 KB block
  __shared__ float sh[2][KB][17];
  const int kb = blockIdx.x, nkb = Kp / KB, nrb = (rows + 15) / 16;
  const int t = threadIdx.x & (TPB - 1), half_wg = threadIdx.x / TPB;       // two row blocks in flight per iteration
  // thread t of a block holds 16 bytes: piece h = t / 64, lane l = t % 64 -> g = l / 16, r = l % 16, k = g*2E + h*E + e
  const int h = t >> 6, l = t & 63, g = l >> 4, r = l & 15;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  for (int rb = half_wg; rb < nrb; rb += 2) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    const u4 raw = *reinterpret_cast<const u4*>(reinterpret_cast<const char*>(Pk) + ((int64_t)rb * nkb + kb) * 2048 + t * 16);
    const TP* v = reinterpret_cast<const TP*>(&raw);
    const bool ok = rb * 16 + r < rows;
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] += ok ? (float)v[e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < E; ++e) sh[half_wg][g * 2 * E + h * E + e][r] = acc[e];
  __syncthreads();
  if (threadIdx.x < KB) {
    const int j = kb * KB + threadIdx.x;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) sum += sh[q][threadIdx.x][rr];
    if (j < M) mem_attn[j] += sum;
  }
}

// Column sums of the thresholded, renormalised probabilities straight from the scores (the two-launch memory read keeps no
// probability matrix): mem_attn[j] += sum_r [p >= thr] p / Z'_r with p = exp(S[r,j] - m_r) / Z_r; rowz[r] = (Z', m, 1/Z, -)
// as the P.V launch left them.  One workgroup per 64 columns, its 16 waves take the rows r = w, w+16, ... with 8 loads in
// flight each; the sixteen partial sums meet in LDS in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void colsum_softmax_kernel(const float* __restrict__ S, int64_t ld, int rows, int M,
                                                              const float4* __restrict__ rowz, float thr, float* __restrict__ mem_attn,
                                                              float* __restrict__ mem_count, int app_P) {
  __shared__ float sh[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const int jc = j < M ? j : M - 1;
  float acc = 0.f;
  for (int r0 = w; r0 < rows && blockIdx.x * 64 < M; r0 += 16 * 8) {
    float sv[8];
    float4 rz[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int r = r0 + 16 * c, rc = r < rows ? r : rows - 1;
      sv[c] = S[(int64_t)rc * ld + jc];
      rz[c] = rowz[rc];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float p = __expf(sv[c] - rz[c].y) * rz[c].z;
      acc += (p < thr || r0 + 16 * c >= rows) ? 0.f : p / rz[c].x;
    }
  }
  sh[w][lane] = acc;
  __syncthreads();
  if (w == 0 && j < M) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][lane];
    mem_attn[j] += t;
  }
  // the bookkeeping of the append that follows the read (sp3_mem_append), when the caller commits the frame in the same launch
  if (w == 0 && app_P > 0) {
    if (j < M) mem_count[j] += 1.0f;
    else if (j < M + app_P) { mem_count[j] = 0.f; mem_attn[j] = 0.f; }
  }
}

// mem_attn[j] += sum_r P[r, j]: one workgroup per 64 columns, its 4 waves take the rows r = w, w+4, ... (coalesced 256-byte
// row segments), partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void colsum_accum_kernel(const float* __restrict__ P, int64_t ld, int rows, int M,
                                                           float* __restrict__ mem_attn) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f;
  if (j < M) {
    int r = w;
    for (; r + 4 < rows; r += 8) { s0 += P[(int64_t)r * ld + j]; s1 += P[(int64_t)(r + 4) * ld + j]; }
    if (r < rows) s0 += P[(int64_t)r * ld + j];
  }
  sh[w][lane] = s0 + s1;
  __syncthreads();
  if (w == 0 && j < M) mem_attn[j] += (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}

// cos_sim, stage 1: one wave per (stored frame t, patch p) pair -> cosv[t*P + p] = cos(k[p], wm[t,p]).
// state != nullptr (sp3_cos_sim_state): wm is the bank's k_raw, the frames are its last state[1] ones (rows [M - T P, M), M = state[0])
__global__ __launch_bounds__(256) void cos_pair_kernel(const float* __restrict__ k, const float* __restrict__ wm, int TP, int P, int C,
                                                       float* __restrict__ cosv, const int* __restrict__ state) {
  const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (state) {
    const int M = state[0], T = state[1];
    TP = T * P < TP ? T * P : TP;
    wm += (int64_t)(M - TP) * C;
  }
  if (j >= TP) return;
  const float* a = k + (int64_t)(j % P) * C;
  const float* b = wm + (int64_t)j * C;
  float dot = 0.f, na = 0.f, nb = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const float4 x = *reinterpret_cast<const float4*>(a + c);
    const float4 y = *reinterpret_cast<const float4*>(b + c);
    dot += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    na += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    nb += (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
  }
  dot = wave_sum(dot); na = wave_sum(na); nb = wave_sum(nb);
  // F.normalize(p=2, eps=1e-12): x / max(||x||, eps)
  if (lane == 0) cosv[j] = dot / (fmaxf(sqrtf(na), 1e-12f) * fmaxf(sqrtf(nb), 1e-12f));
}

// cos_sim, stage 2: one block per stored frame t: mean over patches, fixed reduction order.
__global__ __launch_bounds__(256) void cos_mean_kernel(const float* __restrict__ cosv, int P, float* __restrict__ score, const int* __restrict__ state) {
  __shared__ float sh[4];
  const int t = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (state && t >= state[1]) return;
  float acc = 0.f;
  for (int p = threadIdx.x; p < P; p += 256) acc += cosv[(int64_t)t * P + p];
  acc = wave_sum(acc);
  if (lane == 0) sh[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) score[t] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / (float)P;
}

// ------------------------------------------------------------------ long-bank read without a score matrix (round 6), the two small kernels
// between / behind its GEMM stages (gemm_sm.hip: SM_PROB score stage, pvs_kernel P.V stage).
// prob_merge: stats[group][rows_pad] = (m_g, sum_g exp(s - m_g)) of every row's 64-key groups -> scale[group][row] =
// exp(m_g - m_row) / Z_row with m_row = max_g m_g, Z_row = sum_g sum_g exp(m_g - m_row): softmax(s)[row, key] = p~[row, key] * scale[key / 64][row]
// (spann3r/model.py:160 at attn_thresh = 0).  One workgroup per 16 rows, 64 lanes over the groups; rows >= rows get scale 0.
__global__ __launch_bounds__(1024) void prob_merge_kernel(const float2* __restrict__ stats, float* __restrict__ scale, int rows, long rows_pad,
                                                          int Mk, const int* __restrict__ dyn) {
  __shared__ float red[64][17];
  __shared__ float rowm[16], rowiz[16];
  const int r = threadIdx.x & 15, gl = threadIdx.x >> 4, row = blockIdx.x * 16 + r;
  if (dyn) Mk = dyn[0];
  const int ng = (Mk + 63) >> 6;
  const float2* st = stats + row;
  float mx = -INFINITY;
  for (int gq = gl; gq < ng; gq += 64) mx = fmaxf(mx, st[(long)gq * rows_pad].x);
  red[gl][r] = mx;
  __syncthreads();
  if (gl == 0) {
    float m = red[0][r];
#pragma unroll 8
    for (int j = 1; j < 64; ++j) m = fmaxf(m, red[j][r]);
    rowm[r] = m;
  }
  __syncthreads();
  const float m = rowm[r];
  float z = 0.f;
  for (int gq = gl; gq < ng; gq += 64) {
    const float2 v = st[(long)gq * rows_pad];
    z += v.y * __expf(v.x - m);                                   // (an empty group: 0 * exp(-inf) = 0)
  }
  __syncthreads();
  red[gl][r] = z;
  __syncthreads();
  if (gl == 0) {
    float t = 0.f;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) t += red[j][r];                  // fixed order
    rowiz[r] = 1.0f / t;
  }
  __syncthreads();
  const float iz = row < rows ? rowiz[r] : 0.f;
  for (int gq = gl; gq < ng; gq += 64) scale[(long)gq * rows_pad + row] = row < rows ? __expf(st[(long)gq * rows_pad].x - m) * iz : 0.f;
}

// colsum_prob: mem_attn[key] += sum_rows p~[row, key] * scale[key / 64][row]  (spann3r/model.py:180-181) from the fragment-order bf16 p~
// [rows][cap keys]: one workgroup per 64-key group, a wave walks every fourth 16-row block (two 1 KB pieces each), fixed order.
__global__ __launch_bounds__(256) void colsum_prob_kernel(const __bf16* __restrict__ P, const float* __restrict__ scale, int rows, long rows_pad,
                                                          int nkbc, int Mk, const int* __restrict__ dyn, float* __restrict__ mem_attn) {
  __shared__ float part[4][64];
  if (dyn) Mk = dyn[0];
  const int kb = blockIdx.x;
  if (kb * 64 >= Mk) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, r = lane & 15;
  const int nrb = (rows + 15) >> 4;
  float sum[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) sum[e] = 0.f;
  for (int rb = w; rb < nrb; rb += 4) {
    const __bf16* piece = P + ((long)rb * nkbc + kb) * 1024 + lane * 8;      // 2 halves x 512 bf16 each
    const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(piece), p1 = *reinterpret_cast<const bf16x8*>(piece + 512);
    const float sc = scale[(long)kb * rows_pad + rb * 16 + r];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sum[e] = fmaf(sc, (float)p0[e], sum[e]);
      sum[8 + e] = fmaf(sc, (float)p1[e], sum[8 + e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float v = sum[e];
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    if (r == 0) part[w][g * 16 + e] = v;                          // key g*16 + (half e >> 3)*8 + (e & 7) = g*16 + e
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int key = kb * 64 + threadIdx.x;
    if (key < Mk) mem_attn[key] += (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
  }
}

__global__ void bank_state_kernel(int* __restrict__ state, int M, int wm) {
  if (threadIdx.x == 0) { state[0] = M; state[1] = wm; }
}

__global__ __launch_bounds__(256) void mem_append_kernel(float* __restrict__ count, float* __restrict__ attn, int M, int P) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < M) count[j] += 1.0f;
  else if (j < M + P) { count[j] = 0.f; attn[j] = 0.f; }
}

// up to 8 contiguous device-to-device copies in ONE launch (the per-frame bookkeeping of the sequence loop: decoder hook
// outputs into their sequence slots, the frame's results into the caller's buffers, the next pair of encoder features)
struct CopyMultiArgs { const char* src[8]; char* dst[8]; int64_t bytes[8]; };
__global__ __launch_bounds__(256) void copy_multi_kernel(const CopyMultiArgs a) {
  const int e = blockIdx.y;
  const int64_t n16 = a.bytes[e] >> 4;
  const uint4* s = reinterpret_cast<const uint4*>(a.src[e]);
  uint4* d = reinterpret_cast<uint4*>(a.dst[e]);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}

// Single-block bitonic sort of (weight, index): weight descending, index ascending on ties.
constexpr int PRUNE_MAX = 16384;      // 4000 + 8 * P tokens for P up to 1536 (512x768); 128 KB of LDS
__global__ __launch_bounds__(1024) void prune_select_kernel(const float* __restrict__ attn, const float* __restrict__ count,
                                                            int M, float protect, int top_k, int32_t* __restrict__ sel) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  float* w = reinterpret_cast<float*>(raw);
  int32_t* id = reinterpret_cast<int32_t*>(raw + PRUNE_MAX * sizeof(float));
  int n = 1;
  while (n < M) n <<= 1;
  for (int j = threadIdx.x; j < n; j += 1024) {
    if (j < M) {
      const float c = count[j];
      w[j] = c < protect ? 1e8f : attn[j] / c;
      id[j] = j;
    } else {
      w[j] = -INFINITY;
      id[j] = 0x7fffffff;
    }
  }
  __syncthreads();
  for (int k = 2; k <= n; k <<= 1) {
    for (int s = k >> 1; s > 0; s >>= 1) {
      for (int j = threadIdx.x; j < n; j += 1024) {
        const int l = j ^ s;
        if (l > j) {
          const float wj = w[j], wl = w[l];
          const int ij = id[j], il = id[l];
          // "j before l" in the target order: larger weight first, smaller index first on ties
          const bool j_first = (wj > wl) || (wj == wl && ij < il);
          const bool up = (j & k) == 0;
          if (up != j_first) { w[j] = wl; w[l] = wj; id[j] = il; id[l] = ij; }
        }
      }
      __syncthreads();
    }
  }
  for (int j = threadIdx.x; j < top_k; j += 1024) sel[j] = id[j];
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                          const int32_t* __restrict__ sel, int C) {
  const int i = blockIdx.x;
  const T* s = src + (int64_t)sel[i] * C;
  T* d = dst + (int64_t)i * C;
  for (int c = threadIdx.x; c < C; c += 256) d[c] = s[c];
}

template <typename T>
__global__ __launch_bounds__(256) void gather_cols_kernel(const T* __restrict__ src, int64_t ld_src, T* __restrict__ dst,
                                                          int64_t ld_dst, const int32_t* __restrict__ sel, int n_sel, int n_fill) {
  const int c = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_sel) dst[(int64_t)c * ld_dst + i] = src[(int64_t)c * ld_src + sel[i]];
  else if (i < n_fill) dst[(int64_t)c * ld_dst + i] = (T)0.f;
}

__global__ __launch_bounds__(256) void gather_1d_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        const int32_t* __restrict__ sel, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[sel[i]];
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, float v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ s, int64_t lds, float* __restrict__ d, int64_t ldd,
                                                     int cols4, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int64_t r = i / cols4;
  const int c = (int)(i - r * cols4);
  reinterpret_cast<float4*>(d + r * ldd)[c] = reinterpret_cast<const float4*>(s + r * lds)[c];
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ s, __bf16* __restrict__ d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = (__bf16)s[i];
}


// ------------------------------------------------------------------ memory write (spann3r/model.py:80-95 + the LayerNorms of :154,174)
// One launch per stored frame.  A workgroup owns TG consecutive bank tokens (TG = 8 bf16 / 4 fp32 = the tokens that share
// a 16-byte piece of the fragment-order V^T), aligned to the absolute token index; it has 2 TG waves, ONE per (token, key | value) row
// -- a frame is only 25 workgroups, so the launch is as long as one wave's dependent chain: with 4 waves taking two tokens' key
// and value rows in turn it was 15 us, one row per wave brings it to a third (round 5):
//   k row: raw copy; k_hat = LN_k(k); K' = k_hat (.) gamma_q stored in fragment order (row = token);
//          s = alpha * sum_c K'_c (of the ROUNDED operand), b = alpha * sum_c beta_q,c k_hat_c  -> fold LN_q into the S GEMM
//   v row: raw copy; v_hat = LN_v(v) staged in LDS, then written TRANSPOSED in fragment order (row = channel, k = token):
//          the TG tokens of a channel are one 16-byte store.
struct BankWriteArgs {
  const float *fk, *fv;
  float *k_raw, *v_raw;
  void *k_hat, *v_hat_t;
  float *s_bank, *b_bank;
  const float *gk, *bk, *gv, *bv, *gq, *bq;
  float eps, alpha;
  int M, P, C, cap;
  const int* state;                                         // if set: the first row comes from state[0] (device-resident fill level)
};

template <typename TW>
__global__ __launch_bounds__(128 * (16 / (int)sizeof(TW))) void bank_write_kernel(const BankWriteArgs a) {
  constexpr int TG = 16 / (int)sizeof(TW), NTH = 128 * TG;
  constexpr bool BF = sizeof(TW) == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char raw_[];
  TW* stage = reinterpret_cast<TW*>(raw_);                 // [TG][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int M0 = a.state ? __builtin_amdgcn_readfirstlane(a.state[0]) : a.M;    // first row of the frame (device-resident fill level, or the launch argument)
  const int t0 = (M0 / TG + blockIdx.x) * TG;              // first token of this group (absolute bank row)
  const int C = a.C;
  {
    const int j = wave % TG;
    const bool is_key = wave < TG;                         // waves [0, TG): key rows, [TG, 2 TG): value rows
    const int t = t0 + j;
    const bool live = t >= M0 && t < M0 + a.P;             // wave-uniform
    const int pr = t - M0;
    // ---- key row
    if (live && is_key) {
      const float* x = a.fk + (int64_t)pr * C;
      float s1 = 0.f, s2 = 0.f;
      for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + c);
        *reinterpret_cast<float4*>(a.k_raw + (int64_t)t * C + c) = v;
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
      s1 = wave_sum(s1); s2 = wave_sum(s2);
      const float mean = s1 / (float)C;
      const float rstd = 1.0f / sqrtf(fmaxf(s2 / (float)C - mean * mean, 0.f) + a.eps);
      float ss = 0.f, sb = 0.f;
      for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + c);
        const float4 g = *reinterpret_cast<const float4*>(a.gk + c), be = *reinterpret_cast<const float4*>(a.bk + c);
        const float4 gq = *reinterpret_cast<const float4*>(a.gq + c), bq = *reinterpret_cast<const float4*>(a.bq + c);
        const float kh[4] = {(v.x - mean) * rstd * g.x + be.x, (v.y - mean) * rstd * g.y + be.y,
                             (v.z - mean) * rstd * g.z + be.z, (v.w - mean) * rstd * g.w + be.w};
        const float gqa[4] = {gq.x, gq.y, gq.z, gq.w}, bqa[4] = {bq.x, bq.y, bq.z, bq.w};
        TW w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          w[e] = (TW)(kh[e] * gqa[e]);
          ss += (float)w[e];
          sb += bqa[e] * kh[e];
        }
        TW* dst = reinterpret_cast<TW*>(a.k_hat) + packed_off(t, c, C, BF);
        if constexpr (BF) {
          bf16x4 o4;
          o4[0] = w[0]; o4[1] = w[1]; o4[2] = w[2]; o4[3] = w[3];
          *reinterpret_cast<bf16x4*>(dst) = o4;                      // (4 consecutive k stay contiguous in the fragment order)
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e] = w[e];
        }
      }
      ss = wave_sum(ss); sb = wave_sum(sb);
      if (lane == 0) { a.s_bank[t] = a.alpha * ss; a.b_bank[t] = a.alpha * sb; }
    }
    // ---- value row
    if (live && !is_key) {
      const float* x = a.fv + (int64_t)pr * C;
      float s1 = 0.f, s2 = 0.f;
      for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + c);
        *reinterpret_cast<float4*>(a.v_raw + (int64_t)t * C + c) = v;
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
      s1 = wave_sum(s1); s2 = wave_sum(s2);
      const float mean = s1 / (float)C;
      const float rstd = 1.0f / sqrtf(fmaxf(s2 / (float)C - mean * mean, 0.f) + a.eps);
      for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + c);
        const float4 g = *reinterpret_cast<const float4*>(a.gv + c), be = *reinterpret_cast<const float4*>(a.bv + c);
        TW* d = stage + j * C + c;
        d[0] = (TW)((v.x - mean) * rstd * g.x + be.x); d[1] = (TW)((v.y - mean) * rstd * g.y + be.y);
        d[2] = (TW)((v.z - mean) * rstd * g.z + be.z); d[3] = (TW)((v.w - mean) * rstd * g.w + be.w);
      }
    }
  }
  __syncthreads();
  // ---- transposed store of the staged value rows: channel c, tokens t0 .. t0+TG-1 = one 16-byte piece
  const bool whole = t0 >= M0 && t0 + TG <= M0 + a.P;
  for (int c = threadIdx.x; c < C; c += NTH) {
    TW* dst = reinterpret_cast<TW*>(a.v_hat_t) + packed_off(c, t0, a.cap, BF);
    if (whole) {
      TW o[TG];
#pragma unroll
      for (int j = 0; j < TG; ++j) o[j] = stage[j * C + c];
      typedef __attribute__((ext_vector_type(4))) unsigned int u4;
      *reinterpret_cast<u4*>(dst) = *reinterpret_cast<const u4*>(o);
    } else {
#pragma unroll
      for (int j = 0; j < TG; ++j)
        if (t0 + j >= M0 && t0 + j < M0 + a.P) dst[j] = stage[j * C + c];
    }
  }
}

// Fragment-order copy + per-32-column (sum, sum of squares) partials of a row-major fp32 matrix: what a producer GEMM's
// stats_out / c2 options write, as a stand-alone launch (the reference-shaped eager path hands memory_read a plain tensor).
template <typename TW>
__global__ __launch_bounds__(256) void pack_stats_kernel(const float* __restrict__ x, int64_t ldx, int rows, int C,
                                                         TW* __restrict__ packed, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  for (int c = lane * 4; c < C; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)row * ldx + c);
    TW* d = packed + packed_off(row, c, C, sizeof(TW) == 2);
    d[0] = (TW)v.x; d[1] = (TW)v.y; d[2] = (TW)v.z; d[3] = (TW)v.w;
    float s1 = (v.x + v.y) + (v.z + v.w), s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
    for (int o_ = 1; o_ < 8; o_ <<= 1) { s1 += __shfl_xor(s1, o_); s2 += __shfl_xor(s2, o_); }
    if ((lane & 7) == 0) reinterpret_cast<float2*>(stats)[(int64_t)row * (C >> 5) + (c >> 5)] = make_float2(s1, s2);
  }
}

// prune (spann3r/model.py:195-201) on the fragment-order banks.  Rows of K' (row = token): 16-byte pieces move whole.
__global__ __launch_bounds__(256) void gather_packed_rows_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                                 const int32_t* __restrict__ sel, int C, int esz) {
  const int i = blockIdx.x, si = sel[i];
  const int per = 16 / esz;                                // elements per 16-byte piece
  const bool bf = esz == 2;
  for (int c = threadIdx.x * per; c < C; c += 256 * per) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    *reinterpret_cast<u4*>(dst + packed_off(i, c, C, bf) * esz) = *reinterpret_cast<const u4*>(src + packed_off(si, c, C, bf) * esz);
  }
}

// Columns of V^T (k = token): destination piece (channel c, tokens 8q..8q+7) gathers its elements one by one.
template <typename TW>
__global__ __launch_bounds__(256) void gather_packed_cols_kernel(const TW* __restrict__ src, TW* __restrict__ dst,
                                                                 const int32_t* __restrict__ sel, int n_sel, int n_fill, int cap) {
  constexpr int TG = 16 / (int)sizeof(TW);
  const int c = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;            // piece index along the token axis
  if (q * TG >= n_fill) return;
  TW o[TG];
#pragma unroll
  for (int j = 0; j < TG; ++j) {
    const int i = q * TG + j;
    o[j] = i < n_sel ? src[packed_off(c, sel[i], cap, sizeof(TW) == 2)] : (TW)0.f;
  }
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  *reinterpret_cast<u4*>(dst + packed_off(c, q * TG, cap, sizeof(TW) == 2)) = *reinterpret_cast<const u4*>(o);
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int sp3_softmax_thresh(const float* S, float* P, int64_t ld, int64_t strideS, int rows, int M, int Mpad,
                                  float thresh, int batch, void* P_packed, int64_t stride_packed, int packed_bf16, void* stream) {
  SP3_CHECK(S && (P || P_packed) && rows > 0 && M > 0 && Mpad >= M && ld >= Mpad, "sp3_softmax_thresh: bad arguments");
  const dim3 grid(rows, batch > 0 ? batch : 1);
  if (packed_bf16) {
    const int Kp = (M + 63) / 64 * 64;
    hipLaunchKernelGGL(softmax_thresh_kernel<__bf16>, grid, dim3(256), 0, ST(stream), S, P, ld, strideS, M, Mpad, thresh,
                       reinterpret_cast<__bf16*>(P_packed), Kp, stride_packed);
  } else {
    const int Kp = (M + 31) / 32 * 32;
    hipLaunchKernelGGL(softmax_thresh_kernel<float>, grid, dim3(256), 0, ST(stream), S, P, ld, strideS, M, Mpad, thresh,
                       reinterpret_cast<float*>(P_packed), Kp, stride_packed);
  }
  SP3_LAUNCH_CHECK("sp3_softmax_thresh");
  return 0;
}

extern "C" int sp3_softmax_pack(const float* S, int64_t ld, int64_t strideS, int rows, int M, float thresh, int batch, void* P_packed,
                                int64_t stride_packed, int packed_bf16, float* rowstat_ws, void* stream) {
  SP3_CHECK(S && P_packed && rowstat_ws && rows > 0 && M > 0 && ld >= M && batch > 0, "sp3_softmax_pack: bad arguments");
  SP3_CHECK((reinterpret_cast<uintptr_t>(P_packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(rowstat_ws) & 15) == 0 && rows <= 65535 * 16,
            "sp3_softmax_pack: P_packed / rowstat_ws must be 16-byte aligned");
  float4* rs = reinterpret_cast<float4*>(rowstat_ws);
  hipLaunchKernelGGL(softmax_rowstat_kernel, dim3(rows, batch), dim3(256), 0, ST(stream), S, ld, strideS, rows, M, thresh, rs);
  const int KB = packed_bf16 ? 64 : 32, Kp = (M + KB - 1) / KB * KB, nkb = Kp / KB, nrb = (rows + 15) / 16;
  const dim3 grid((nkb + 1) / 2, nrb, batch);
  if (packed_bf16) hipLaunchKernelGGL(softmax_pack_kernel<__bf16>, grid, dim3(256), 0, ST(stream), S, ld, strideS, rows, M, Kp, thresh, rs,
                                      reinterpret_cast<__bf16*>(P_packed), stride_packed);
  else hipLaunchKernelGGL(softmax_pack_kernel<float>, grid, dim3(256), 0, ST(stream), S, ld, strideS, rows, M, Kp, thresh, rs,
                          reinterpret_cast<float*>(P_packed), stride_packed);
  SP3_LAUNCH_CHECK("sp3_softmax_pack");
  return 0;
}

extern "C" int sp3_colsum_packed(const void* P_packed, int packed_bf16, int rows, int M, float* mem_attn, void* stream) {
  SP3_CHECK(P_packed && mem_attn && rows > 0 && M > 0, "sp3_colsum_packed: bad arguments");
  if (packed_bf16) {
    const int Kp = (M + 63) / 64 * 64;
    hipLaunchKernelGGL(colsum_packed_kernel<__bf16>, dim3(Kp / 64), dim3(256), 0, ST(stream), (const __bf16*)P_packed, rows, M, Kp, mem_attn);
  } else {
    const int Kp = (M + 31) / 32 * 32;
    hipLaunchKernelGGL(colsum_packed_kernel<float>, dim3(Kp / 32), dim3(256), 0, ST(stream), (const float*)P_packed, rows, M, Kp, mem_attn);
  }
  SP3_LAUNCH_CHECK("sp3_colsum_packed");
  return 0;
}

extern "C" int sp3_colsum_softmax(const float* S, int64_t ld, int rows, int M, const float* rowz, float thresh, float* mem_attn,
                                  float* mem_count, int append_P, void* stream) {
  SP3_CHECK(S && rowz && mem_attn && rows > 0 && M > 0 && ld >= M, "sp3_colsum_softmax: bad arguments");
  SP3_CHECK((reinterpret_cast<uintptr_t>(rowz) & 15) == 0, "sp3_colsum_softmax: rowz must be 16-byte aligned");
  SP3_CHECK(append_P == 0 || (append_P > 0 && mem_count), "sp3_colsum_softmax: append needs mem_count");
  hipLaunchKernelGGL(colsum_softmax_kernel, dim3((M + append_P + 63) / 64), dim3(1024), 0, ST(stream), S, ld, rows, M,
                     reinterpret_cast<const float4*>(rowz), thresh, mem_attn, mem_count, append_P);
  SP3_LAUNCH_CHECK("sp3_colsum_softmax");
  return 0;
}

extern "C" int sp3_bank_write(const sp3_bank_write_desc* dp, void* stream) {
  SP3_CHECK(dp != nullptr, "sp3_bank_write: null descriptor");
  const sp3_bank_write_desc& d = *dp;
  SP3_CHECK(d.feat_k && d.feat_v && d.k_raw && d.v_raw && d.k_hat && d.v_hat_t && d.s_bank && d.b_bank, "sp3_bank_write: null pointer");
  SP3_CHECK(d.gamma_k && d.beta_k && d.gamma_v && d.beta_v && d.gamma_q && d.beta_q, "sp3_bank_write: null LayerNorm parameter");
  SP3_CHECK(d.P > 0 && d.C > 0 && d.C % 256 == 0 && d.cap % 64 == 0 && (d.state ? d.P <= d.cap : (d.M >= 0 && d.M + d.P <= d.cap)),
            "sp3_bank_write: bad geometry M=%d P=%d C=%d cap=%d", d.M, d.P, d.C, d.cap);
  SP3_CHECK(d.wdtype == SP3_F32 || d.wdtype == SP3_BF16, "sp3_bank_write: bad wdtype %d", d.wdtype);
  BankWriteArgs a{d.feat_k, d.feat_v, d.k_raw, d.v_raw, d.k_hat, d.v_hat_t, d.s_bank, d.b_bank, d.gamma_k, d.beta_k,
                  d.gamma_v, d.beta_v, d.gamma_q, d.beta_q, d.eps, d.alpha, d.M, d.P, d.C, d.cap, d.state};
  const int TG = d.wdtype == SP3_BF16 ? 8 : 4;
  // (device-resident first row: the frame may start anywhere inside a token group -- one group more than the aligned case; idle if not needed)
  const int groups = d.state ? (d.P + TG - 1) / TG + 1 : (d.M + d.P + TG - 1) / TG - d.M / TG;
  const size_t lds = (size_t)TG * d.C * (d.wdtype == SP3_BF16 ? 2 : 4);
  if (d.wdtype == SP3_BF16) hipLaunchKernelGGL(bank_write_kernel<__bf16>, dim3(groups), dim3(1024), lds, ST(stream), a);
  else hipLaunchKernelGGL(bank_write_kernel<float>, dim3(groups), dim3(512), lds, ST(stream), a);
  SP3_LAUNCH_CHECK("sp3_bank_write");
  return 0;
}

extern "C" int sp3_pack_stats(const float* x, int64_t ldx, int rows, int C, void* packed, int packed_bf16, float* stats, void* stream) {
  SP3_CHECK(x && packed && stats && rows > 0 && C > 0 && C % 256 == 0 && ldx % 4 == 0, "sp3_pack_stats: bad arguments");
  if (packed_bf16) hipLaunchKernelGGL(pack_stats_kernel<__bf16>, dim3((rows + 3) / 4), dim3(256), 0, ST(stream), x, ldx, rows, C, (__bf16*)packed, stats);
  else hipLaunchKernelGGL(pack_stats_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, ST(stream), x, ldx, rows, C, (float*)packed, stats);
  SP3_LAUNCH_CHECK("sp3_pack_stats");
  return 0;
}

extern "C" int sp3_gather_packed_rows(const void* src, void* dst, const int32_t* sel, int n_sel, int C, int elem_size, void* stream) {
  SP3_CHECK(src && dst && sel && n_sel > 0 && C > 0 && C % 64 == 0 && (elem_size == 2 || elem_size == 4), "sp3_gather_packed_rows: bad arguments");
  hipLaunchKernelGGL(gather_packed_rows_kernel, dim3(n_sel), dim3(256), 0, ST(stream), (const char*)src, (char*)dst, sel, C, elem_size);
  SP3_LAUNCH_CHECK("sp3_gather_packed_rows");
  return 0;
}

extern "C" int sp3_gather_packed_cols(const void* src, void* dst, const int32_t* sel, int n_sel, int n_fill, int C, int cap,
                                      int elem_size, void* stream) {
  SP3_CHECK(src && dst && sel && n_sel > 0 && C > 0 && C % 16 == 0 && cap % 64 == 0 && n_fill <= cap && n_sel <= n_fill,
            "sp3_gather_packed_cols: bad arguments");
  const int TG = 16 / elem_size;
  dim3 grid((n_fill / TG + 255) / 256, C);
  if (elem_size == 2)
    hipLaunchKernelGGL(gather_packed_cols_kernel<__bf16>, grid, dim3(256), 0, ST(stream), (const __bf16*)src, (__bf16*)dst, sel, n_sel, n_fill, cap);
  else if (elem_size == 4)
    hipLaunchKernelGGL(gather_packed_cols_kernel<float>, grid, dim3(256), 0, ST(stream), (const float*)src, (float*)dst, sel, n_sel, n_fill, cap);
  else SP3_CHECK(false, "sp3_gather_packed_cols: elem_size %d", elem_size);
  SP3_LAUNCH_CHECK("sp3_gather_packed_cols");
  return 0;
}

extern "C" int sp3_colsum_accum(const float* P, int64_t ld, int rows, int M, float* mem_attn, void* stream) {
  SP3_CHECK(P && mem_attn && rows > 0 && M > 0, "sp3_colsum_accum: bad arguments");
  hipLaunchKernelGGL(colsum_accum_kernel, dim3((M + 63) / 64), dim3(256), 0, ST(stream), P, ld, rows, M, mem_attn);
  SP3_LAUNCH_CHECK("sp3_colsum_accum");
  return 0;
}

extern "C" int sp3_cos_sim(const float* k, const float* wm, int T, int P, int C, float* scratch, float* score, void* stream) {
  SP3_CHECK(k && wm && scratch && score && T > 0 && P > 0 && C > 0 && C % 4 == 0, "sp3_cos_sim: bad arguments");
  hipLaunchKernelGGL(cos_pair_kernel, dim3((T * P + 3) / 4), dim3(256), 0, ST(stream), k, wm, T * P, P, C, scratch, (const int*)nullptr);
  hipLaunchKernelGGL(cos_mean_kernel, dim3(T), dim3(256), 0, ST(stream), scratch, P, score, (const int*)nullptr);
  SP3_LAUNCH_CHECK("sp3_cos_sim");
  return 0;
}

extern "C" int sp3_cos_sim_state(const float* k, const float* k_raw, int Tmax, int P, int C, const int32_t* state, float* scratch, float* score,
                                 void* stream) {
  SP3_CHECK(k && k_raw && state && scratch && score && Tmax > 0 && P > 0 && C > 0 && C % 4 == 0, "sp3_cos_sim_state: bad arguments");
  hipLaunchKernelGGL(cos_pair_kernel, dim3((Tmax * P + 3) / 4), dim3(256), 0, ST(stream), k, k_raw, Tmax * P, P, C, scratch, state);
  hipLaunchKernelGGL(cos_mean_kernel, dim3(Tmax), dim3(256), 0, ST(stream), scratch, P, score, state);
  SP3_LAUNCH_CHECK("sp3_cos_sim_state");
  return 0;
}

extern "C" int sp3_prob_merge(const float* stats, float* scale, int rows, int M, int cap, const int32_t* dyn_n, void* stream) {
  SP3_CHECK(stats && scale && rows > 0 && M > 0 && cap % 64 == 0 && M <= cap, "sp3_prob_merge: bad arguments");
  const long rows_pad = (long)((rows + 255) & ~255);
  hipLaunchKernelGGL(prob_merge_kernel, dim3((unsigned)(rows_pad / 16)), dim3(1024), 0, ST(stream), reinterpret_cast<const float2*>(stats), scale, rows,
                     rows_pad, M, dyn_n);
  SP3_LAUNCH_CHECK("sp3_prob_merge");
  return 0;
}

extern "C" int sp3_colsum_prob(const void* P_packed, const float* scale, int rows, int M, int cap, const int32_t* dyn_n, float* mem_attn, void* stream) {
  SP3_CHECK(P_packed && scale && mem_attn && rows > 0 && M > 0 && cap % 64 == 0 && M <= cap, "sp3_colsum_prob: bad arguments");
  const long rows_pad = (long)((rows + 255) & ~255);
  hipLaunchKernelGGL(colsum_prob_kernel, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, ST(stream), reinterpret_cast<const __bf16*>(P_packed), scale, rows,
                     rows_pad, cap / 64, M, dyn_n, mem_attn);
  SP3_LAUNCH_CHECK("sp3_colsum_prob");
  return 0;
}

extern "C" int sp3_bank_state_set(int32_t* state, int M, int wm, void* stream) {
  SP3_CHECK(state && M >= 0 && wm >= 0, "sp3_bank_state_set: bad arguments");
  hipLaunchKernelGGL(bank_state_kernel, dim3(1), dim3(64), 0, ST(stream), state, M, wm);
  SP3_LAUNCH_CHECK("sp3_bank_state_set");
  return 0;
}

extern "C" int sp3_copy_multi(int n, const void* const* src, void* const* dst, const int64_t* bytes, void* stream) {
  SP3_CHECK(n >= 1 && n <= 8 && src && dst && bytes, "sp3_copy_multi: 1..8 copies");
  CopyMultiArgs a;
  int64_t mx = 0;
  for (int i = 0; i < 8; ++i) {
    const int j = i < n ? i : 0;
    SP3_CHECK(src[j] && dst[j] && bytes[j] > 0 && bytes[j] % 16 == 0 &&
              ((reinterpret_cast<uintptr_t>(src[j]) | reinterpret_cast<uintptr_t>(dst[j])) & 15) == 0,
              "sp3_copy_multi: copy %d must be non-empty, 16-byte aligned and a multiple of 16 bytes", j);
    a.src[i] = reinterpret_cast<const char*>(src[j]); a.dst[i] = reinterpret_cast<char*>(dst[j]); a.bytes[i] = bytes[j];
    mx = bytes[j] > mx ? bytes[j] : mx;
  }
  int64_t bx = (mx / 16 + 256 * 4 - 1) / (256 * 4);           // ~4 vectors per thread for the largest copy
  bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
  hipLaunchKernelGGL(copy_multi_kernel, dim3((unsigned)bx, n), dim3(256), 0, ST(stream), a);
  SP3_LAUNCH_CHECK("sp3_copy_multi");
  return 0;
}

extern "C" int sp3_mem_append(float* count, float* attn, int M, int P, void* stream) {
  SP3_CHECK(count && attn && M >= 0 && P > 0, "sp3_mem_append: bad arguments");
  hipLaunchKernelGGL(mem_append_kernel, dim3((M + P + 255) / 256), dim3(256), 0, ST(stream), count, attn, M, P);
  SP3_LAUNCH_CHECK("sp3_mem_append");
  return 0;
}

extern "C" int sp3_prune_select(const float* attn, const float* count, int M, float protect, int top_k, int32_t* sel,
                                void* stream) {
  SP3_CHECK(attn && count && sel, "sp3_prune_select: null pointer");
  SP3_CHECK(M > 0 && M <= PRUNE_MAX && top_k > 0 && top_k <= M, "sp3_prune_select: M=%d top_k=%d (M <= %d)", M, top_k, PRUNE_MAX);
  static bool raised = false;         // one-time opt-in to > 64 KiB of dynamic LDS
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(prune_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PRUNE_MAX * 8);
    SP3_CHECK(e == hipSuccess, "sp3_prune_select: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    raised = true;
  }
  hipLaunchKernelGGL(prune_select_kernel, dim3(1), dim3(1024), PRUNE_MAX * 8, ST(stream), attn, count, M, protect, top_k, sel);
  SP3_LAUNCH_CHECK("sp3_prune_select");
  return 0;
}

extern "C" int sp3_gather_rows(const void* src, void* dst, const int32_t* sel, int n_sel, int C, int elem_size, void* stream) {
  SP3_CHECK(src && dst && sel && n_sel > 0 && C > 0, "sp3_gather_rows: bad arguments");
  if (elem_size == 4)
    hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(n_sel), dim3(256), 0, ST(stream), (const float*)src, (float*)dst, sel, C);
  else if (elem_size == 2)
    hipLaunchKernelGGL(gather_rows_kernel<uint16_t>, dim3(n_sel), dim3(256), 0, ST(stream), (const uint16_t*)src, (uint16_t*)dst, sel, C);
  else SP3_CHECK(false, "sp3_gather_rows: elem_size %d", elem_size);
  SP3_LAUNCH_CHECK("sp3_gather_rows");
  return 0;
}

extern "C" int sp3_gather_cols(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const int32_t* sel, int n_sel,
                               int n_fill, int C, int elem_size, void* stream) {
  SP3_CHECK(src && dst && sel && n_sel > 0 && C > 0, "sp3_gather_cols: bad arguments");
  const int n = n_fill > n_sel ? n_fill : n_sel;
  dim3 grid((n + 255) / 256, C);
  if (elem_size == 4)
    hipLaunchKernelGGL(gather_cols_kernel<float>, grid, dim3(256), 0, ST(stream), (const float*)src, ld_src, (float*)dst, ld_dst, sel, n_sel, n);
  else if (elem_size == 2)
    hipLaunchKernelGGL(gather_cols_kernel<__bf16>, grid, dim3(256), 0, ST(stream), (const __bf16*)src, ld_src, (__bf16*)dst, ld_dst, sel, n_sel, n);
  else SP3_CHECK(false, "sp3_gather_cols: elem_size %d", elem_size);
  SP3_LAUNCH_CHECK("sp3_gather_cols");
  return 0;
}

extern "C" int sp3_gather_1d(const float* src, float* dst, const int32_t* sel, int n_sel, void* stream) {
  SP3_CHECK(src && dst && sel && n_sel > 0, "sp3_gather_1d: bad arguments");
  hipLaunchKernelGGL(gather_1d_kernel, dim3((n_sel + 255) / 256), dim3(256), 0, ST(stream), src, dst, sel, n_sel);
  SP3_LAUNCH_CHECK("sp3_gather_1d");
  return 0;
}

// A kernel of known length for calibrating host-side timers: spins `cycles` shader clocks and reports its own duration
// (first to last instruction) in ticks of the constant 100 MHz counter (s_memrealtime).
__global__ void spin_kernel(int64_t cycles, int64_t* ticks) {
  const int64_t t0 = wall_clock64();
  const int64_t c0 = clock64();
  while (clock64() - c0 < cycles) __builtin_amdgcn_s_sleep(2);
  if (threadIdx.x == 0) ticks[0] = wall_clock64() - t0;
}

extern "C" int sp3_spin(int64_t cycles, int64_t* ticks_100mhz, void* stream) {
  SP3_CHECK(ticks_100mhz && cycles >= 0, "sp3_spin: bad arguments");
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, ST(stream), cycles, ticks_100mhz);
  SP3_LAUNCH_CHECK("sp3_spin");
  return 0;
}

extern "C" int sp3_fill_f32(float* p, float v, int64_t n, void* stream) {
  SP3_CHECK(p && n > 0, "sp3_fill_f32: bad arguments");
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(stream), p, v, n);
  SP3_LAUNCH_CHECK("sp3_fill_f32");
  return 0;
}

extern "C" int sp3_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  SP3_CHECK(src && dst && n > 0, "sp3_cast_f32_to_bf16: bad arguments");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(stream), src, (__bf16*)dst, n);
  SP3_LAUNCH_CHECK("sp3_cast_f32_to_bf16");
  return 0;
}

extern "C" int sp3_copy2d_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int rows, int cols, void* stream) {
  SP3_CHECK(src && dst && rows > 0 && cols > 0 && cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0, "sp3_copy2d_f32: bad arguments");
  const int64_t total4 = (int64_t)rows * (cols / 4);
  hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, ST(stream), src, lds, dst, ldd, cols / 4, total4);
  SP3_LAUNCH_CHECK("sp3_copy2d_f32");
  return 0;
}
