// Backward of the spatial-memory read in its training form (reference forward: spann3r/model.py:145-183 with
// attn_thresh = 0 and mem_dropout):   out = (dropout(softmax(LN_q(q) LN_k(K)^T / sqrt(C))) LN_v(V)) + q.
// The four P x T x C products of the backward are sp3_gemm launches (fp32 or bf16 MFMA); this file holds what sits between
// them: transposes (the GEMM computes A[M,K] . W[N,K]^T only), the softmax / dropout backward, the LayerNorm backward with
// deterministic parameter gradients, and the dropout multiply of the forward.
#include "common.h"
#include <math.h>

namespace {

// dst[c * ldd + r] = src[r * lds + c], 32 x 32 tiles through LDS; columns [rows, ldd) of dst are left untouched
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd,
                                                        int rows, int cols, int64_t bs_src, int64_t bs_dst) {
  __shared__ float t[32][33];
  src += (int64_t)blockIdx.z * bs_src;
  dst += (int64_t)blockIdx.z * bs_dst;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const int r = r0 + ty + k, c = c0 + tx;
    t[ty + k][tx] = (r < rows && c < cols) ? src[(int64_t)r * lds_ + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const int c = c0 + ty + k, r = r0 + tx;
    if (c < cols && r < rows) dst[(int64_t)c * ldd + r] = t[tx][ty + k];
  }
}

// exact-erf GELU (nn.GELU default, croco/models/blocks.py:73-79) and its derivative: Phi(x) + x phi(x)
__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float v = x[i]; y[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float v = x[i];
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * v * v);
    dx[i] = dy[i] * (cdf + v * pdf);
  }
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) o[i] = a[i] * b[i];
}

// dS[r, :] = alpha * A[r, :] (.) (dA[r, :] - sum_j dA[r, j] A[r, j]),  dA = dAd (.) mask (mask may be null): one workgroup per row
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ A, const float* __restrict__ dAd, const float* __restrict__ mask,
                                                          float* __restrict__ dS, int64_t ld, int T, float alpha) {
  __shared__ double sh[4];
  const int64_t o = (int64_t)blockIdx.x * ld;
  double dot = 0.0;
  for (int j = threadIdx.x; j < T; j += 256) {
    const float da = dAd[o + j] * (mask ? mask[o + j] : 1.0f);
    dot += (double)(da * A[o + j]);
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) dot += __shfl_xor(dot, s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
  __syncthreads();
  const float tot = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
  for (int j = threadIdx.x; j < T; j += 256) {
    const float da = dAd[o + j] * (mask ? mask[o + j] : 1.0f);
    dS[o + j] = alpha * A[o + j] * (da - tot);
  }
}

// LayerNorm backward, one wave per row (C <= 4096, C % 4 == 0): xh = (x - mean) rstd,
//   dx = rstd (dy g - mean(dy g) - xh mean(dy g xh)) (+ dx_add), and this row's share of dgamma / dbeta goes to
//   part[row_block][2][C] (row blocks of 4 rows = one workgroup; summed in fixed order by ln_param_reduce).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, int64_t ldy, const float* __restrict__ dx_add,
                                                            int64_t lda, float* __restrict__ dx, int64_t ldo, float* __restrict__ part,
                                                            int rows, int C, float eps) {
  extern __shared__ float sh[];                       // [4][2][C]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  float* mine = sh + (size_t)w * 2 * C;
  if (row < rows) {
    const float* xr = x + (int64_t)row * ldx;
    const float* dr = dy + (int64_t)row * ldy;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = xr[c]; s1 += v; s2 += v * v; }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const float mean = s1 / C, rstd = 1.0f / sqrtf(fmaxf(s2 / C - mean * mean, 0.f) + eps);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float xh = (xr[c] - mean) * rstd, dg = dr[c] * gamma[c];
      a += dg; b += dg * xh;
    }
    a = wave_sum(a) / C; b = wave_sum(b) / C;
    for (int c = lane; c < C; c += 64) {
      const float xh = (xr[c] - mean) * rstd, d = dr[c];
      float g = rstd * (d * gamma[c] - a - xh * b);
      if (dx_add) g += dx_add[(int64_t)row * lda + c];
      dx[(int64_t)row * ldo + c] = g;
      mine[c] = d * xh; mine[C + c] = d;
    }
  } else {
    for (int c = lane; c < 2 * C; c += 64) mine[c] = 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256)
    part[(int64_t)blockIdx.x * 2 * C + c] = (sh[c] + sh[2 * C + c]) + (sh[4 * C + c] + sh[6 * C + c]);
}

// dgamma / dbeta [2][C] (+= if accumulate) = sum over the row blocks, fixed order
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 2 * C) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)part[(int64_t)b * 2 * C + c];
  float* o = c < C ? dgamma + c : dbeta + (c - C);
  *o = (accumulate ? *o : 0.f) + (float)s;
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int sp3_transpose(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int rows, int cols, void* stream) {
  SP3_CHECK(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "sp3_transpose: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, 1), dim3(256), 0, ST(stream), src, ld_src, dst, ld_dst, rows,
                     cols, (int64_t)0, (int64_t)0);
  SP3_LAUNCH_CHECK("sp3_transpose");
  return 0;
}

extern "C" int sp3_transpose_batched(const float* src, int64_t ld_src, int64_t stride_src, float* dst, int64_t ld_dst, int64_t stride_dst,
                                     int rows, int cols, int batch, void* stream) {
  SP3_CHECK(src && dst && rows > 0 && cols > 0 && batch > 0 && batch <= 65535 && ld_src >= cols && ld_dst >= rows, "sp3_transpose_batched: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, batch), dim3(256), 0, ST(stream), src, ld_src, dst, ld_dst, rows,
                     cols, stride_src, stride_dst);
  SP3_LAUNCH_CHECK("sp3_transpose_batched");
  return 0;
}

extern "C" int sp3_gelu(const float* x, float* y, int64_t n, void* stream) {
  SP3_CHECK(x && y && n > 0, "sp3_gelu: bad arguments");
  hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(stream), x, y, n);
  SP3_LAUNCH_CHECK("sp3_gelu");
  return 0;
}

extern "C" int sp3_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  SP3_CHECK(x && dy && dx && n > 0, "sp3_gelu_bwd: bad arguments");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(stream), x, dy, dx, n);
  SP3_LAUNCH_CHECK("sp3_gelu_bwd");
  return 0;
}

extern "C" int sp3_mul(const float* a, const float* b, float* out, int64_t n, void* stream) {
  SP3_CHECK(a && b && out && n > 0, "sp3_mul: bad arguments");
  hipLaunchKernelGGL(mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(stream), a, b, out, n);
  SP3_LAUNCH_CHECK("sp3_mul");
  return 0;
}

extern "C" int sp3_softmax_bwd(const float* A, const float* dAd, const float* mask, float* dS, int64_t ld, int rows, int T, float alpha,
                               void* stream) {
  SP3_CHECK(A && dAd && dS && rows > 0 && T > 0 && ld >= T, "sp3_softmax_bwd: bad arguments");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(rows), dim3(256), 0, ST(stream), A, dAd, mask, dS, ld, T, alpha);
  SP3_LAUNCH_CHECK("sp3_softmax_bwd");
  return 0;
}

extern "C" int sp3_layernorm_bwd(const float* x, int64_t ldx, const float* gamma, const float* dy, int64_t ldy, const float* dx_add,
                                 int64_t ld_add, float* dx, int64_t ld_dx, float* dgamma, float* dbeta, int accumulate, float* scratch,
                                 int rows, int C, float eps, void* stream) {
  SP3_CHECK(x && gamma && dy && dx && dgamma && dbeta && scratch, "sp3_layernorm_bwd: null pointer");
  SP3_CHECK(rows > 0 && C > 0 && C % 4 == 0 && C <= 4096, "sp3_layernorm_bwd: rows=%d C=%d", rows, C);
  const int nblk = (rows + 3) / 4;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblk), dim3(256), (size_t)8 * C * sizeof(float), ST(stream), x, ldx, gamma, dy, ldy, dx_add,
                     ld_add, dx, ld_dx, scratch, rows, C, eps);
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, ST(stream), scratch, nblk, C, dgamma, dbeta, accumulate);
  SP3_LAUNCH_CHECK("sp3_layernorm_bwd");
  return 0;
}
