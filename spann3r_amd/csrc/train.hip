// Backward of the spatial-memory read in its training form (reference forward: spann3r/model.py:145-183 with
// attn_thresh = 0 and mem_dropout):   out = (dropout(softmax(LN_q(q) LN_k(K)^T / sqrt(C))) LN_v(V)) + q.
// The four P x T x C products of the backward are sp3_gemm launches (fp32 or bf16 MFMA); this file holds what sits between
// them: transposes (the GEMM computes A[M,K] . W[N,K]^T only), the softmax / dropout backward, the LayerNorm backward with
// deterministic parameter gradients, and the dropout multiply of the forward.
#include "common.h"
#include <math.h>

namespace {

// dst[c * ldd + r] = src[r * lds + c], 32 x 32 tiles through LDS; columns [rows, pad_to) of dst are zeroed, [pad_to, ldd) left untouched
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd,
                                                        int rows, int cols, int64_t bs_src, int64_t bs_dst, int pad_to) {
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
    if (c < cols && r < pad_to) dst[(int64_t)c * ldd + r] = t[tx][ty + k];          // (rows >= `rows` were loaded as zeros)
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
                                                          float* __restrict__ dS, int64_t ld, int T, float alpha, int Tpad) {
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
  for (int j = T + threadIdx.x; j < Tpad; j += 256) dS[o + j] = 0.f;
}

// LayerNorm backward, one wave per row (C <= 2048: 8*C*4 bytes of dynamic LDS stay under the 64 KB default; C % 4 == 0): xh = (x - mean) rstd,
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

// Same arithmetic with the row held in registers (V float4 per lane: C <= 256 V; 16-byte aligned rows): x, dy and gamma are read ONCE
// with 16-byte loads instead of three passes of 4-byte ones, a wave takes RPW consecutive rows and keeps its dgamma / dbeta share in
// registers, so a workgroup (4 RPW rows) writes one partial block: fewer, fuller blocks for ln_param_reduce.
template <int V, int RPW>
__global__ __launch_bounds__(256) void layernorm_bwd_reg_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                                const float* __restrict__ dy, int64_t ldy, const float* __restrict__ dx_add,
                                                                int64_t lda, float* __restrict__ dx, int64_t ldo, float* __restrict__ part,
                                                                int rows, int C, float eps) {
  extern __shared__ float sh[];                       // [4][2][C]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float4 gm[V], ag[V], ab[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c = (lane + 64 * j) * 4;
    gm[j] = c < C ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int row0 = (blockIdx.x * 4 + w) * RPW;
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int row = row0 + i;
    if (row >= rows) break;
    float4 xv[V], dv[V];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int c = (lane + 64 * j) * 4;
      const bool in = c < C;
      xv[j] = in ? *reinterpret_cast<const float4*>(x + (int64_t)row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      dv[j] = in ? *reinterpret_cast<const float4*>(dy + (int64_t)row * ldy + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      s1 += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
      s2 += (xv[j].x * xv[j].x + xv[j].y * xv[j].y) + (xv[j].z * xv[j].z + xv[j].w * xv[j].w);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const float mean = s1 / C, rstd = 1.0f / sqrtf(fmaxf(s2 / C - mean * mean, 0.f) + eps);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (c < C) {
        const float xs[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w}, ds[4] = {dv[j].x, dv[j].y, dv[j].z, dv[j].w};
        const float gs[4] = {gm[j].x, gm[j].y, gm[j].z, gm[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xs[e] - mean) * rstd, dg = ds[e] * gs[e];
          a += dg; b += dg * xh;
        }
      }
    }
    a = wave_sum(a) / C; b = wave_sum(b) / C;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (c < C) {
        const float xs[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w}, ds[4] = {dv[j].x, dv[j].y, dv[j].z, dv[j].w};
        const float gs[4] = {gm[j].x, gm[j].y, gm[j].z, gm[j].w};
        float o[4], pg[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xs[e] - mean) * rstd;
          o[e] = rstd * (ds[e] * gs[e] - a - xh * b);
          pg[e] = ds[e] * xh;
        }
        if (dx_add) {
          const float4 r = *reinterpret_cast<const float4*>(dx_add + (int64_t)row * lda + c);
          o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
        }
        *reinterpret_cast<float4*>(dx + (int64_t)row * ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
        ag[j].x += pg[0]; ag[j].y += pg[1]; ag[j].z += pg[2]; ag[j].w += pg[3];
        ab[j].x += ds[0]; ab[j].y += ds[1]; ab[j].z += ds[2]; ab[j].w += ds[3];
      }
    }
  }
  float* mine = sh + (size_t)w * 2 * C;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c = (lane + 64 * j) * 4;
    if (c < C) {
      *reinterpret_cast<float4*>(mine + c) = ag[j];
      *reinterpret_cast<float4*>(mine + C + c) = ab[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256)
    part[(int64_t)blockIdx.x * 2 * C + c] = (sh[c] + sh[2 * C + c]) + (sh[4 * C + c] + sh[6 * C + c]);
}

// dgamma / dbeta [2][C] (+= if accumulate) = sum over the row blocks, fixed order: a workgroup owns 64 of the 2C columns, its 4
// waves take the row blocks b = w, w+4, ... (two independent chains each), the partial sums meet in LDS in wave order
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
  __shared__ double sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double s0 = 0.0, s1 = 0.0;
  if (c < 2 * C) {
    int b = w;
    for (; b + 4 < nblk; b += 8) { s0 += (double)part[(int64_t)b * 2 * C + c]; s1 += (double)part[(int64_t)(b + 4) * 2 * C + c]; }
    if (b < nblk) s0 += (double)part[(int64_t)b * 2 * C + c];
  }
  sh[w][lane] = s0 + s1;
  __syncthreads();
  if (w == 0 && c < 2 * C) {
    const double s = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
    float* o = c < C ? dgamma + c : dbeta + (c - C);
    *o = (accumulate ? *o : 0.f) + (float)s;
  }
}

// ---- DPT head pieces (croco/models/dpt_block.py, dust3r/heads/postprocess.py), NHWC fp32 ----
// col[(b, oy, ox), (ky*3 + kx)*C + c] = x[b, oy*s - 1 + ky, ox*s - 1 + kx, c] (0 outside): the 3x3 / pad 1 convolution as a GEMM
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x, float* __restrict__ col, int H, int W, int C, int OH,
                                                        int OW, int stride, int64_t total4) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int c4 = C >> 2;
  const int cc = (int)(idx % c4);
  int64_t r = idx / c4;
  const int tap = (int)(r % 9);
  r /= 9;
  const int ox = (int)(r % OW);
  r /= OW;
  const int oy = (int)(r % OH);
  const int b = (int)(r / OH);
  const int iy = oy * stride - 1 + tap / 3, ix = ox * stride - 1 + tap % 3;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = reinterpret_cast<const float4*>(x)[(((int64_t)b * H + iy) * W + ix) * c4 + cc];
  reinterpret_cast<float4*>(col)[idx] = v;
}

// the adjoint, gather form: dx[b, iy, ix, c] = sum over taps of dcol[(b, oy, ox), tap, c] with oy*s - 1 + ky == iy (fixed order)
__global__ __launch_bounds__(256) void col2im3x3_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int H, int W, int C, int OH,
                                                        int OW, int stride, int64_t total4) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int c4 = C >> 2;
  const int cc = (int)(idx % c4);
  int64_t r = idx / c4;
  const int ix = (int)(r % W);
  r /= W;
  const int iy = (int)(r % H);
  const int b = (int)(r / H);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    if (ty < 0 || ty % stride) continue;
    const int oy = ty / stride;
    if (oy >= OH) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      if (tx < 0 || tx % stride) continue;
      const int ox = tx / stride;
      if (ox >= OW) continue;
      const float4 v = reinterpret_cast<const float4*>(dcol)[((((int64_t)b * OH + oy) * OW + ox) * 9 + ky * 3 + kx) * c4 + cc];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  reinterpret_cast<float4*>(dx)[idx] = a;
}

__global__ __launch_bounds__(256) void relu_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = fmaxf(x[i], 0.f);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx[i] = x[i] > 0.f ? dy[i] : 0.f;
}

// adjoint of upsample2x_kernel (dpt.hip: bilinear x2, align_corners=True, optional crop to outH x outW), gather form
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C, int outH,
                                                             int outW, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  int64_t r = idx / C;
  const int ix = (int)(r % W);
  r /= W;
  const int iy = (int)(r % H);
  const int b = (int)(r / H);
  const int OH = 2 * H, OW = 2 * W;
  const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  auto weight = [](int o, float sc, int n, int i) -> float {        // weight of source index i in output index o, as the forward computes it
    const float f = sc * (float)o;
    int i0 = (int)f;
    i0 = i0 < n - 1 ? i0 : n - 1;
    const int i1 = i0 < n - 1 ? i0 + 1 : i0;
    const float l = f - (float)i0;
    return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
  };
  const int oy_lo = sh > 0.f ? max(0, (int)floorf((iy - 1) / sh) - 1) : 0, oy_hi = sh > 0.f ? min(outH - 1, (int)ceilf((iy + 1) / sh) + 1) : outH - 1;
  const int ox_lo = sw > 0.f ? max(0, (int)floorf((ix - 1) / sw) - 1) : 0, ox_hi = sw > 0.f ? min(outW - 1, (int)ceilf((ix + 1) / sw) + 1) : outW - 1;
  float a = 0.f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float wy = weight(oy, sh, H, iy);
    if (wy == 0.f) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float wx = weight(ox, sw, W, ix);
      if (wx != 0.f) a += wy * wx * dy[(((int64_t)b * outH + oy) * outW + ox) * C + c];
    }
  }
  dx[idx] = a;
}

// dust3r/heads/postprocess.py:10-58 (depth 'exp', conf 'exp' + 1): raw [M, 4] -> pts [M, 3] = xyz / |xyz| * expm1(|xyz|), conf = 1 + exp(c)
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ raw, float* __restrict__ pts, float* __restrict__ conf, int64_t M) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const float4 v = reinterpret_cast<const float4*>(raw)[i];
  const float d = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  const float s = expm1f(d) / fmaxf(d, 1e-8f);
  pts[3 * i] = v.x * s; pts[3 * i + 1] = v.y * s; pts[3 * i + 2] = v.z * s;
  conf[i] = 1.0f + __expf(v.w);
}
__global__ __launch_bounds__(256) void postprocess_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ dpts, const float* __restrict__ dconf,
                                                              float* __restrict__ draw, int64_t M) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const float4 v = reinterpret_cast<const float4*>(raw)[i];
  const float d = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z), dc = fmaxf(d, 1e-8f);
  // pts = xyz * g(d), g = expm1(d) / d:  d pts = g dxyz + xyz g'(d) (xhat . dxyz), g' = (exp(d) d - expm1(d)) / d^2
  const float g = expm1f(d) / dc, gp = d > 1e-8f ? (__expf(d) * d - expm1f(d)) / (d * d) : 0.5f;
  const float gx = dpts[3 * i], gy = dpts[3 * i + 1], gz = dpts[3 * i + 2];
  const float dot = (v.x * gx + v.y * gy + v.z * gz) * gp / dc;
  float4 o;
  o.x = g * gx + v.x * dot; o.y = g * gy + v.y * dot; o.z = g * gz + v.z * dot;
  o.w = dconf[i] * __expf(v.w);
  reinterpret_cast<float4*>(draw)[i] = o;
}

// torch.optim.AdamW (the optimizer of spann3r/training.py:327), one parameter tensor per launch, decoupled weight decay:
//   p *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float rsbc2,
                                                    float gscale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  float pi = p[i] * (1.0f - lr * wd);
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  pi -= (lr / bc1) * mi / (sqrtf(vi) * rsbc2 + eps);
  p[i] = pi; m[i] = mi; v[i] = vi;
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int sp3_transpose(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int rows, int cols, void* stream) {
  SP3_CHECK(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "sp3_transpose: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, 1), dim3(256), 0, ST(stream), src, ld_src, dst, ld_dst, rows,
                     cols, (int64_t)0, (int64_t)0, rows);
  SP3_LAUNCH_CHECK("sp3_transpose");
  return 0;
}

extern "C" int sp3_transpose_batched(const float* src, int64_t ld_src, int64_t stride_src, float* dst, int64_t ld_dst, int64_t stride_dst,
                                     int rows, int cols, int batch, void* stream) {
  SP3_CHECK(src && dst && rows > 0 && cols > 0 && batch > 0 && batch <= 65535 && ld_src >= cols && ld_dst >= rows, "sp3_transpose_batched: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, batch), dim3(256), 0, ST(stream), src, ld_src, dst, ld_dst, rows,
                     cols, stride_src, stride_dst, rows);
  SP3_LAUNCH_CHECK("sp3_transpose_batched");
  return 0;
}

extern "C" int sp3_transpose_pad(const float* src, int64_t ld_src, int64_t stride_src, float* dst, int64_t ld_dst, int64_t stride_dst,
                                 int rows, int cols, int batch, int pad_to, void* stream) {
  SP3_CHECK(src && dst && rows > 0 && cols > 0 && batch > 0 && batch <= 65535 && ld_src >= cols && ld_dst >= pad_to && pad_to >= rows &&
            pad_to <= (rows + 31) / 32 * 32, "sp3_transpose_pad: bad arguments (rows=%d pad_to=%d ld_dst=%lld)", rows, pad_to, (long long)ld_dst);
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, batch), dim3(256), 0, ST(stream), src, ld_src, dst, ld_dst, rows,
                     cols, stride_src, stride_dst, pad_to);
  SP3_LAUNCH_CHECK("sp3_transpose_pad");
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
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(rows), dim3(256), 0, ST(stream), A, dAd, mask, dS, ld, T, alpha, T);
  SP3_LAUNCH_CHECK("sp3_softmax_bwd");
  return 0;
}

extern "C" int sp3_softmax_bwd_pad(const float* A, const float* dAd, const float* mask, float* dS, int64_t ld, int rows, int T, int Tpad,
                                   float alpha, void* stream) {
  SP3_CHECK(A && dAd && dS && rows > 0 && T > 0 && Tpad >= T && ld >= Tpad, "sp3_softmax_bwd_pad: bad arguments");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(rows), dim3(256), 0, ST(stream), A, dAd, mask, dS, ld, T, alpha, Tpad);
  SP3_LAUNCH_CHECK("sp3_softmax_bwd_pad");
  return 0;
}

extern "C" int sp3_layernorm_bwd(const float* x, int64_t ldx, const float* gamma, const float* dy, int64_t ldy, const float* dx_add,
                                 int64_t ld_add, float* dx, int64_t ld_dx, float* dgamma, float* dbeta, int accumulate, float* scratch,
                                 int rows, int C, float eps, void* stream) {
  SP3_CHECK(x && gamma && dy && dx && dgamma && dbeta && scratch, "sp3_layernorm_bwd: null pointer");
  SP3_CHECK(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, "sp3_layernorm_bwd: rows=%d C=%d (C <= 2048, C %% 4 == 0)", rows, C);
  int nblk = (rows + 3) / 4;
  const bool al = (((ldx | ldy | ld_dx | (dx_add ? ld_add : 0)) & 3) == 0) &&
                  (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dx_add) |
                     reinterpret_cast<uintptr_t>(gamma)) & 15) == 0);
  const int V = (C + 255) / 256;
  if (al && V <= 8) {
    constexpr int RPW = 2;
    nblk = (rows + 4 * RPW - 1) / (4 * RPW);
    const size_t lds = (size_t)8 * C * sizeof(float);
#define SP3_LNB(v) hipLaunchKernelGGL((layernorm_bwd_reg_kernel<v, RPW>), dim3(nblk), dim3(256), lds, ST(stream), x, ldx, gamma, dy, ldy, dx_add, ld_add, dx, ld_dx, scratch, rows, C, eps)
    switch (V) {
      case 1: SP3_LNB(1); break;
      case 2: SP3_LNB(2); break;
      case 3: SP3_LNB(3); break;
      case 4: SP3_LNB(4); break;
      case 5: case 6: SP3_LNB(6); break;
      default: SP3_LNB(8); break;
    }
#undef SP3_LNB
  } else {
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblk), dim3(256), (size_t)8 * C * sizeof(float), ST(stream), x, ldx, gamma, dy, ldy, dx_add,
                       ld_add, dx, ld_dx, scratch, rows, C, eps);
  }
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * C + 63) / 64), dim3(256), 0, ST(stream), scratch, nblk, C, dgamma, dbeta, accumulate);
  SP3_LAUNCH_CHECK("sp3_layernorm_bwd");
  return 0;
}


extern "C" int sp3_im2col3x3(const float* x, float* col, int B, int H, int W, int C, int stride, void* stream) {
  SP3_CHECK(x && col && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2), "sp3_im2col3x3: bad arguments");
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int64_t total4 = (int64_t)B * OH * OW * 9 * (C / 4);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, ST(stream), x, col, H, W, C, OH, OW, stride, total4);
  SP3_LAUNCH_CHECK("sp3_im2col3x3");
  return 0;
}

extern "C" int sp3_col2im3x3(const float* dcol, float* dx, int B, int H, int W, int C, int stride, void* stream) {
  SP3_CHECK(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2), "sp3_col2im3x3: bad arguments");
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int64_t total4 = (int64_t)B * H * W * (C / 4);
  hipLaunchKernelGGL(col2im3x3_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, ST(stream), dcol, dx, H, W, C, OH, OW, stride, total4);
  SP3_LAUNCH_CHECK("sp3_col2im3x3");
  return 0;
}

extern "C" int sp3_relu(const float* x, float* y, int64_t n, void* stream) {
  SP3_CHECK(x && y && n > 0, "sp3_relu: bad arguments");
  hipLaunchKernelGGL(relu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(stream), x, y, n);
  SP3_LAUNCH_CHECK("sp3_relu");
  return 0;
}

extern "C" int sp3_relu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  SP3_CHECK(x && dy && dx && n > 0, "sp3_relu_bwd: bad arguments");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(stream), x, dy, dx, n);
  SP3_LAUNCH_CHECK("sp3_relu_bwd");
  return 0;
}

extern "C" int sp3_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, int outH, int outW, void* stream) {
  SP3_CHECK(dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && outH > 0 && outH <= 2 * H && outW > 0 && outW <= 2 * W, "sp3_upsample2x_bwd: bad arguments");
  const int64_t total = (int64_t)B * H * W * C;
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST(stream), dy, dx, H, W, C, outH, outW, total);
  SP3_LAUNCH_CHECK("sp3_upsample2x_bwd");
  return 0;
}

extern "C" int sp3_postprocess(const float* raw, float* pts, float* conf, int64_t M, void* stream) {
  SP3_CHECK(raw && pts && conf && M > 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0, "sp3_postprocess: bad arguments");
  hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ST(stream), raw, pts, conf, M);
  SP3_LAUNCH_CHECK("sp3_postprocess");
  return 0;
}

extern "C" int sp3_postprocess_bwd(const float* raw, const float* dpts, const float* dconf, float* draw, int64_t M, void* stream) {
  SP3_CHECK(raw && dpts && dconf && draw && M > 0, "sp3_postprocess_bwd: bad arguments");
  hipLaunchKernelGGL(postprocess_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ST(stream), raw, dpts, dconf, draw, M);
  SP3_LAUNCH_CHECK("sp3_postprocess_bwd");
  return 0;
}


extern "C" int sp3_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                         int step, float grad_scale, void* stream) {
  SP3_CHECK(p && g && m && v && n > 0 && step >= 1, "sp3_adamw: bad arguments");
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(stream), p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                     bc1, 1.0f / sqrtf(bc2), grad_scale);
  SP3_LAUNCH_CHECK("sp3_adamw");
  return 0;
}
