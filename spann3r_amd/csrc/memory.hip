// Spatial-memory kernels (reference: spann3r/model.py:97-210).
//
// The bank is a static-capacity arena owned by the host side (spann3r_amd/model.py):
//   K_raw [cap,1024] fp32, V_raw [cap,1024] fp32 (what the reference calls mem_k / mem_v),
//   K_hat [cap,1024] = LN_k(K_raw) and V_hat^T [1024,cap] = LN_v(V_raw)^T in the MFMA dtype,
//   mem_attn [cap], mem_count [cap] fp32.
// LayerNorm is row-wise, so normalising ONCE at write time is exactly what the reference recomputes
// over the whole bank at every read (model.py:154,174).  A read is then
//   S = LN_q(q) . K_hat^T / 32   (sp3_gemm, alpha)      -> sp3_softmax_thresh -> P
//   out = P . V_hat + q          (sp3_gemm, residual)   ;  mem_attn += colsum(P)  (sp3_colsum_accum)
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

// One block per row.  softmax over [0,M); optional threshold + renormalise; zero-fill [M,Mpad).
__global__ __launch_bounds__(256) void softmax_thresh_kernel(const float* __restrict__ S, float* __restrict__ P, int64_t ld,
                                                             int64_t strideS, int M, int Mpad, float thresh,
                                                             __bf16* __restrict__ Pk, int Kp, int64_t stridePk) {
  __shared__ float sh[8];
  const float* s = S + (int64_t)blockIdx.y * strideS + (int64_t)blockIdx.x * ld;
  float* p = P + (int64_t)blockIdx.y * strideS + (int64_t)blockIdx.x * ld;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < M; j += 256) mx = fmaxf(mx, s[j]);
  mx = block_reduce_max(mx, sh);
  float sum = 0.f;
  for (int j = threadIdx.x; j < M; j += 256) sum += expf(s[j] - mx);
  sum = block_reduce_sum(sum, sh);
  const float inv = 1.0f / sum;
  if (thresh > 0.f) {
    float kept = 0.f;
    for (int j = threadIdx.x; j < M; j += 256) {
      float v = expf(s[j] - mx) * inv;
      v = v < thresh ? 0.f : v;
      kept += v;
    }
    kept = block_reduce_sum(kept, sh);
    for (int j = threadIdx.x; j < M; j += 256) {
      float v = expf(s[j] - mx) * inv;
      v = v < thresh ? 0.f : v;
      p[j] = v / kept;
    }
  } else {
    for (int j = threadIdx.x; j < M; j += 256) p[j] = expf(s[j] - mx) * inv;
  }
  for (int j = M + threadIdx.x; j < Mpad; j += 256) p[j] = 0.f;
  if (Pk) {
    // second copy for the P.V GEMM: bf16, fragment order [rows, Kp] (Kp = M rounded up to 64, zero filled): 8 consecutive
    // probabilities = one 16-byte store.  Reads back this block's own fp32 row (visible after the barrier).
    __syncthreads();
    __bf16* pk = Pk + (int64_t)blockIdx.y * stridePk;
    const int row = blockIdx.x;
    for (int j8 = threadIdx.x * 8; j8 < Kp; j8 += 256 * 8) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)((j8 + e) < M ? p[j8 + e] : 0.f);
      *reinterpret_cast<bf16x8*>(pk + packed_off(row, j8, Kp, true)) = o;
    }
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
__global__ __launch_bounds__(256) void cos_pair_kernel(const float* __restrict__ k, const float* __restrict__ wm, int TP, int P, int C,
                                                       float* __restrict__ cosv) {
  const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
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
__global__ __launch_bounds__(256) void cos_mean_kernel(const float* __restrict__ cosv, int P, float* __restrict__ score) {
  __shared__ float sh[4];
  const int t = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc = 0.f;
  for (int p = threadIdx.x; p < P; p += 256) acc += cosv[(int64_t)t * P + p];
  acc = wave_sum(acc);
  if (lane == 0) sh[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) score[t] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / (float)P;
}

__global__ __launch_bounds__(256) void mem_append_kernel(float* __restrict__ count, float* __restrict__ attn, int M, int P) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < M) count[j] += 1.0f;
  else if (j < M + P) { count[j] = 0.f; attn[j] = 0.f; }
}

// Single-block bitonic sort of (weight, index): weight descending, index ascending on ties.
constexpr int PRUNE_MAX = 8192;
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

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int sp3_softmax_thresh(const float* S, float* P, int64_t ld, int64_t strideS, int rows, int M, int Mpad,
                                  float thresh, int batch, void* P_packed, int64_t stride_packed, void* stream) {
  SP3_CHECK(S && P && rows > 0 && M > 0 && Mpad >= M && ld >= Mpad, "sp3_softmax_thresh: bad arguments");
  const int Kp = (M + 63) / 64 * 64;
  hipLaunchKernelGGL(softmax_thresh_kernel, dim3(rows, batch > 0 ? batch : 1), dim3(256), 0, ST(stream), S, P, ld, strideS, M,
                     Mpad, thresh, reinterpret_cast<__bf16*>(P_packed), Kp, stride_packed);
  SP3_LAUNCH_CHECK("sp3_softmax_thresh");
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
  hipLaunchKernelGGL(cos_pair_kernel, dim3((T * P + 3) / 4), dim3(256), 0, ST(stream), k, wm, T * P, P, C, scratch);
  hipLaunchKernelGGL(cos_mean_kernel, dim3(T), dim3(256), 0, ST(stream), scratch, P, score);
  SP3_LAUNCH_CHECK("sp3_cos_sim");
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
