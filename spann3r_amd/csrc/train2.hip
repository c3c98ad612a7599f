// Training step, bf16 mode (spann3r/training.py:170-259): the pieces that turn the fp32 autograd tape into
// bf16 MFMA work and the optimizer into a handful of bucket-wide launches.
//  * sp3_pack_bf16   : fp32 row-major [rows, cols] -> the GEMM's bf16 fragment order (common.h packed_off), as [rows, cols]
//                      and / or as the TRANSPOSE [cols, rows] in the same pass.  Every product of a Linear's backward is then
//                      an A . W^T launch on packed operands (dX = dY . (W^T)^T, dW = dY^T . (X^T)^T) -- no fp32 transposes.
//  * sp3_sumsq_partial / sp3_clip_coef : global gradient norm of the flat gradient buckets in a fixed order (deterministic)
//                      and torch.nn.utils.clip_grad_norm_'s coefficient, left ON THE DEVICE for the update kernel
//                      (croco/utils/misc.py:262-288 calls it with clip_grad = 1.0, training.py:227-228).
//  * sp3_adamw_flat  : torch.optim.AdamW over a whole flat bucket (parameters, gradients, moments share one element layout);
//                      weight decay / lr scale / skip are looked up per 1024-element chunk (parameters start on chunk
//                      boundaries: runner.GradReducer), the gradient scale is read from device memory.
#include "common.h"
#include <math.h>

namespace {

// one workgroup = a 64 x 64 tile of the source through LDS; both outputs are written as whole 16-byte fragment pieces,
// 16 consecutive lanes = 256 contiguous bytes.  Elements outside [rows, cols] are written as zeros (the contraction pad of
// the GEMM must be zero); row blocks beyond the allocation are skipped.
// partial (nullable): the column sums of src (a Linear's bias gradient, autograd's dY.sum(0)) ride along: every workgroup leaves the sums
// of its 64 x 64 tile in partial[row tile][ceil64(cols)]; colsum_rows_finish_kernel adds the row tiles in order (13 values per column at
// 784 rows instead of a second pass over dY).  (A last-arriving-workgroup reduction inside this launch was measured: the release fence
// every workgroup then needs costs more than the extra launch.)
// conv.C != 0: src is an NHWC map [B, H, W, C] and the matrix being packed is its 3x3 / pad 1 im2col [(b, oy, ox), (ky*3 + kx)*C + c]
// (sp3_im2col3x3's layout), gathered on the fly: the column matrix (9x the map) is never written or read.
struct PackConv { int H, W, C, OH, OW, stride, act; };      // act: 0 none, 1 exact-erf GELU, 2 ReLU applied to every element as it is loaded

__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src, int64_t ld, int rows, int cols,
                                                        __bf16* __restrict__ dst, __bf16* __restrict__ dstT, float* __restrict__ partial,
                                                        PackConv conv) {
  __shared__ float t[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tid = threadIdx.x;
  const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  const int cl0 = (tid & 15) * 4;
  const int tap = conv.C ? (c0 + cl0) / conv.C : 0, ci = conv.C ? (c0 + cl0) - tap * conv.C : 0;      // (C % 4 == 0: a float4 stays inside one tap)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = (tid >> 4) + 16 * i, cl = (tid & 15) * 4;
    const int r = r0 + rl, c = c0 + cl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (conv.C) {
      if (r < rows && c < cols) {
        const int ox = r % conv.OW, q = r / conv.OW, oy = q % conv.OH, b = q / conv.OH;
        const int iy = oy * conv.stride - 1 + tap / 3, ix = ox * conv.stride - 1 + tap % 3;
        if (iy >= 0 && iy < conv.H && ix >= 0 && ix < conv.W)
          v = *reinterpret_cast<const float4*>(src + (((int64_t)b * conv.H + iy) * conv.W + ix) * conv.C + ci);
      }
    } else if (r < rows) {
      const float* p = src + (int64_t)r * ld + c;
      if (vec && c + 3 < cols) v = *reinterpret_cast<const float4*>(p);
      else {
        if (c < cols) v.x = p[0];
        if (c + 1 < cols) v.y = p[1];
        if (c + 2 < cols) v.z = p[2];
        if (c + 3 < cols) v.w = p[3];
      }
    }
    if (conv.act == 1) {                               // the producer's activation rides in the load (gelu_kernel's / relu_kernel's arithmetic)
      v.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752f)); v.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752f));
      v.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752f)); v.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752f));
    } else if (conv.act == 2) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    t[rl][cl] = v.x; t[rl][cl + 1] = v.y; t[rl][cl + 2] = v.z; t[rl][cl + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = tid + 256 * i;                       // piece: (mbl, h, g, r): 16 consecutive r are contiguous in memory
    const int mbl = p >> 7, h = (p >> 6) & 1, g = (p >> 4) & 3, r = p & 15;
    if (dst) {
      const int mb = (r0 >> 4) + mbl, nb = (rows + 15) >> 4, nkb = (cols + 63) >> 6;
      if (mb < nb) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)t[mbl * 16 + r][g * 16 + h * 8 + e];
        const int64_t off = (((((int64_t)mb * nkb + blockIdx.x) * 2 + h) * 4 + g) * 16 + r) * 8;
        *reinterpret_cast<bf16x8*>(dst + off) = o;
      }
    }
    if (dstT) {
      const int mb = (c0 >> 4) + mbl, nb = (cols + 15) >> 4, nkb = (rows + 63) >> 6;
      if (mb < nb) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)t[g * 16 + h * 8 + e][mbl * 16 + r];
        const int64_t off = (((((int64_t)mb * nkb + blockIdx.y) * 2 + h) * 4 + g) * 16 + r) * 8;
        *reinterpret_cast<bf16x8*>(dstT + off) = o;
      }
    }
  }
  if (partial && tid < 64) {                                      // column sums of this tile (rows beyond `rows` were loaded as zeros)
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) s += t[r][tid];
    partial[(int64_t)blockIdx.y * ((int64_t)gridDim.x * 64) + c0 + tid] = s;
  }
}

// Head shuffle of multi-head attention in training (croco/models/blocks.py:100-108, :160-166: the reshape / permute / RoPE between
// the projections and the per-head products).  A part moves one [B, N, H, hd] tensor between two stridings, optionally rotating
// it by RoPE2D (fwd = +1) or its transpose (fwd = -1, the backward), and optionally also writes the per-head TRANSPOSE
// [B*H, hd, r8(N)] (pad columns zero) the A . W^T GEMMs need for the products contracted over tokens.  Up to three parts (q, k, v or
// dq, dk, dv) per launch: what were ~11 ATen copies, two rope launches and a transpose per attention is one launch.
// Workgroup = 64 tokens x hd (<= 64) of one (b, h) through LDS.
struct HeadShuffleArgs { sp3_head_part p[3]; int B, H, hd; float base; const float *cos_tab, *sin_tab; int tab_len; };   // tables [tab_len][hd / 4]

__global__ __launch_bounds__(256) void head_shuffle_kernel(HeadShuffleArgs a) {
  __shared__ float t[64][65];
  const sp3_head_part& P = a.p[blockIdx.z];
  const int N = P.N, n0 = blockIdx.x * 64, tid = threadIdx.x, hd = a.hd;
  if (n0 >= N) return;
  const int b = blockIdx.y / a.H, h = blockIdx.y - b * a.H;
  const float* src = P.src + (int64_t)b * P.s_b + (int64_t)h * P.s_h;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int tl = (tid >> 4) + 16 * i, d = (tid & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 + tl < N && d < hd) v = *reinterpret_cast<const float4*>(src + (int64_t)(n0 + tl) * P.s_n + d);
    t[tl][d] = v.x; t[tl][d + 1] = v.y; t[tl][d + 2] = v.z; t[tl][d + 3] = v.w;
  }
  __syncthreads();
  if (P.pos) {                                       // pair i of a token: (du, du + Q), du = axis * 2Q + f (norm_rope.hip rope2d_kernel)
    const int Q = hd >> 2;
    for (int idx = tid; idx < 64 * 2 * Q; idx += 256) {
      const int tl = idx / (2 * Q), i = idx - tl * 2 * Q;
      if (n0 + tl >= N) continue;
      const int axis = i / Q, f = i - axis * Q;
      const int64_t pi = P.pos[((int64_t)b * N + n0 + tl) * 2 + axis];
      float cs, sn;
      if (pi >= 0 && pi < a.tab_len) {               // the caller's table (positions are small grid coordinates): no libm calls per pair
        cs = a.cos_tab[pi * Q + f];
        sn = P.fwd * a.sin_tab[pi * Q + f];
      } else {
        const float ang = (float)pi * (P.fwd / powf(a.base, (float)f / (float)Q));
        cs = cosf(ang); sn = sinf(ang);
      }
      const int du = axis * 2 * Q + f, dv = du + Q;
      const float u = t[tl][du], v = t[tl][dv];
      t[tl][du] = u * cs - v * sn;
      t[tl][dv] = v * cs + u * sn;
    }
    __syncthreads();
  }
  if (P.dst) {
    float* dst = P.dst + (int64_t)b * P.d_b + (int64_t)h * P.d_h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tl = (tid >> 4) + 16 * i, d = (tid & 15) * 4;
      if (n0 + tl < N && d < hd)
        *reinterpret_cast<float4*>(dst + (int64_t)(n0 + tl) * P.d_n + d) = make_float4(t[tl][d], t[tl][d + 1], t[tl][d + 2], t[tl][d + 3]);
    }
  }
  if (P.dstT) {
    const int64_t Np = P.ldT > 0 ? P.ldT : (int64_t)((N + 7) & ~7);
    float* dT = P.dstT + (int64_t)blockIdx.y * hd * Np;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = (tid >> 4) + 16 * i, tl = (tid & 15) * 4;
      if (d < hd && n0 + tl < Np)                    // (tokens >= N were loaded as zeros: the pad columns)
        *reinterpret_cast<float4*>(dT + d * Np + n0 + tl) = make_float4(t[tl][d], t[tl + 1][d], t[tl + 2][d], t[tl + 3][d]);
    }
  }
}

// column sums of a tall matrix (the bias gradient db = sum_r dY[r, :]) in two deterministic stages: a workgroup takes 64 columns of a
// 256-row chunk (its 4 waves interleave the rows, coalesced 256-byte segments), partial[chunk][col]; then one thread per column
// adds the chunks in order
constexpr int kColChunk = 256;

__global__ __launch_bounds__(256) void colsum_rows_partial_kernel(const float* __restrict__ x, int64_t ld, int rows, int N, float* __restrict__ partial) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * kColChunk, r1 = (r0 + kColChunk) < rows ? (r0 + kColChunk) : rows;
  float s0 = 0.f, s1 = 0.f;
  if (j < N) {
    int r = r0 + w;
    for (; r + 4 < r1; r += 8) { s0 += x[(int64_t)r * ld + j]; s1 += x[(int64_t)(r + 4) * ld + j]; }
    if (r < r1) s0 += x[(int64_t)r * ld + j];
  }
  sh[w][lane] = s0 + s1;
  __syncthreads();
  if (w == 0 && j < N) partial[(int64_t)blockIdx.y * N + j] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}

__global__ __launch_bounds__(256) void colsum_rows_finish_kernel(const float* __restrict__ partial, int nchunk, int N, float* __restrict__ out, int accumulate,
                                                                 int ldp) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  double s0 = 0.0, s1 = 0.0;
  int c = 0;
  for (; c + 1 < nchunk; c += 2) { s0 += (double)partial[(int64_t)c * ldp + j]; s1 += (double)partial[(int64_t)(c + 1) * ldp + j]; }
  if (c < nchunk) s0 += (double)partial[(int64_t)c * ldp + j];
  out[j] = (accumulate ? out[j] : 0.f) + (float)(s0 + s1);
}

// short matrices (a few thousand rows: the token GEMMs' bias gradients): ONE launch, a 1024-thread workgroup per 64 columns, its 16
// waves interleave the rows with four independent chains each (64 rows in flight per column group), partials meet in LDS in wave order
__global__ __launch_bounds__(1024) void colsum_rows_single_kernel(const float* __restrict__ x, int64_t ld, int rows, int N, float* __restrict__ out,
                                                                  int accumulate) {
  __shared__ float sh[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < N) {
    int r = w;
    for (; r + 48 < rows; r += 64) {
      s0 += x[(int64_t)r * ld + j];
      s1 += x[(int64_t)(r + 16) * ld + j];
      s2 += x[(int64_t)(r + 32) * ld + j];
      s3 += x[(int64_t)(r + 48) * ld + j];
    }
    for (; r < rows; r += 16) s0 += x[(int64_t)r * ld + j];
  }
  sh[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && j < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][lane];
    out[j] = (accumulate ? out[j] : 0.f) + t;
  }
}

constexpr int kSumChunk = 16384;                      // elements per workgroup of the norm's first stage

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ partial) {
  __shared__ double sh[4];
  const int64_t b0 = (int64_t)blockIdx.x * kSumChunk;
  double s = 0.0;
  for (int64_t i = b0 + threadIdx.x * 4; i < b0 + kSumChunk && i < n; i += 1024) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    } else {
      for (int64_t j = i; j < n; ++j) s += (double)g[j] * g[j];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// out[0] = extra_scale * min(1, max_norm / (norm + 1e-6)), out[1] = norm   (max_norm <= 0: no clipping, out[0] = extra_scale)
__global__ __launch_bounds__(256) void clip_coef_kernel(const double* __restrict__ partial, int count, float max_norm, float extra_scale,
                                                        float* __restrict__ out, int* __restrict__ step_ctr) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) s += partial[i];       // fixed assignment, fixed order: deterministic
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(sh[0]) * fabsf(extra_scale);          // the norm of the gradients as the optimizer sees them
    float coef = 1.0f;
    if (max_norm > 0.f) coef = fminf(1.0f, max_norm / (norm + 1e-6f));
    out[0] = extra_scale * coef;
    out[1] = norm;
    if (step_ctr) step_ctr[0] += 1;                     // the optimizer's step count lives on the device (hipGraph replays)
  }
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, int64_t n, const float2* __restrict__ chunk, float lr,
                                                         float b1, float b2, float eps, float bc1, float rsbc2, const float* __restrict__ gscale_p,
                                                         float gscale_c, const int* __restrict__ step_dev, const float* __restrict__ lr_dev) {
  const float2 cw = chunk[blockIdx.x];                 // (weight decay, lr scale; < 0: leave the chunk alone)
  if (cw.y < 0.f) return;
  if (step_dev) {                                      // bias corrections from the device-side step count
    const float st = (float)step_dev[0];
    bc1 = 1.0f - powf(b1, st);
    rsbc2 = 1.0f / sqrtf(1.0f - powf(b2, st));
  }
  if (lr_dev) lr = lr_dev[0];
  const float gs = gscale_c * (gscale_p ? gscale_p[0] : 1.0f);
  const float lr_ = lr * cw.y;
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  if (i >= n) return;
  float4 pv = *reinterpret_cast<float4*>(p + i), mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
  const float4 gv = *reinterpret_cast<const float4*>(g + i);
  float pp[4] = {pv.x, pv.y, pv.z, pv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w}, vs[4] = {vv.x, vv.y, vv.z, vv.w};
  const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gi = gg[e] * gs;
    float pi = pp[e] * (1.0f - lr_ * cw.x);
    mm[e] = b1 * mm[e] + (1.0f - b1) * gi;
    vs[e] = b2 * vs[e] + (1.0f - b2) * gi * gi;
    pi -= (lr_ / bc1) * mm[e] / (sqrtf(vs[e]) * rsbc2 + eps);
    pp[e] = pi;
  }
  *reinterpret_cast<float4*>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
  *reinterpret_cast<float4*>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
  *reinterpret_cast<float4*>(v + i) = make_float4(vs[0], vs[1], vs[2], vs[3]);
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int sp3_pack_bf16(const float* src, int64_t ld, int rows, int cols, void* dst, void* dstT, void* stream) {
  SP3_CHECK(src && (dst || dstT) && rows > 0 && cols > 0 && ld >= cols, "sp3_pack_bf16: bad arguments (rows=%d cols=%d ld=%lld)", rows, cols, (long long)ld);
  SP3_CHECK(((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(dstT)) & 15) == 0, "sp3_pack_bf16: outputs must be 16-byte aligned");
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  SP3_CHECK(grid.y <= 65535, "sp3_pack_bf16: too many rows for one launch (%d)", rows);
  hipLaunchKernelGGL(pack_bf16_kernel, grid, dim3(256), 0, ST(stream), src, ld, rows, cols, reinterpret_cast<__bf16*>(dst), reinterpret_cast<__bf16*>(dstT),
                     (float*)nullptr, PackConv{0, 0, 0, 0, 0, 0, 0});
  SP3_LAUNCH_CHECK("sp3_pack_bf16");
  return 0;
}

extern "C" int sp3_pack_bf16_act(const float* src, int64_t ld, int rows, int cols, void* dst, void* dstT, int act, void* stream) {
  SP3_CHECK(src && (dst || dstT) && rows > 0 && cols > 0 && ld >= cols && act >= 0 && act <= 2, "sp3_pack_bf16_act: bad arguments (rows=%d cols=%d ld=%lld act=%d)", rows, cols, (long long)ld, act);
  SP3_CHECK(((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(dstT)) & 15) == 0, "sp3_pack_bf16_act: outputs must be 16-byte aligned");
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  SP3_CHECK(grid.y <= 65535, "sp3_pack_bf16_act: too many rows for one launch (%d)", rows);
  hipLaunchKernelGGL(pack_bf16_kernel, grid, dim3(256), 0, ST(stream), src, ld, rows, cols, reinterpret_cast<__bf16*>(dst), reinterpret_cast<__bf16*>(dstT),
                     (float*)nullptr, PackConv{0, 0, 0, 0, 0, 0, act});
  SP3_LAUNCH_CHECK("sp3_pack_bf16_act");
  return 0;
}

extern "C" int sp3_pack_bf16_conv3x3(const float* x, int B, int H, int W, int C, int stride, int act, void* dst, void* dstT, void* stream) {
  SP3_CHECK(x && (dst || dstT) && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2) && act >= 0 && act <= 2 &&
            (reinterpret_cast<uintptr_t>(x) & 15) == 0, "sp3_pack_bf16_conv3x3: bad arguments");
  SP3_CHECK(((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(dstT)) & 15) == 0, "sp3_pack_bf16_conv3x3: outputs must be 16-byte aligned");
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int64_t rows64 = (int64_t)B * OH * OW;
  SP3_CHECK(rows64 <= 65535LL * 64, "sp3_pack_bf16_conv3x3: too many output pixels for one launch (%lld)", (long long)rows64);
  const int rows = (int)rows64, cols = 9 * C;
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(pack_bf16_kernel, grid, dim3(256), 0, ST(stream), x, (int64_t)cols, rows, cols, reinterpret_cast<__bf16*>(dst),
                     reinterpret_cast<__bf16*>(dstT), (float*)nullptr, PackConv{H, W, C, OH, OW, stride, act});
  SP3_LAUNCH_CHECK("sp3_pack_bf16_conv3x3");
  return 0;
}

extern "C" int sp3_pack_bf16_colsum(const float* src, int64_t ld, int rows, int cols, void* dst, void* dstT, float* colsum, int accumulate,
                                    float* partial_ws, void* stream) {
  SP3_CHECK(src && (dst || dstT) && colsum && partial_ws && rows > 0 && cols > 0 && ld >= cols,
            "sp3_pack_bf16_colsum: bad arguments (rows=%d cols=%d ld=%lld)", rows, cols, (long long)ld);
  SP3_CHECK(((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(dstT)) & 15) == 0, "sp3_pack_bf16_colsum: outputs must be 16-byte aligned");
  SP3_CHECK(rows <= 8192, "sp3_pack_bf16_colsum: rows=%d > 8192 (taller matrices: sp3_pack_bf16 + sp3_colsum_rows)", rows);
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(pack_bf16_kernel, grid, dim3(256), 0, ST(stream), src, ld, rows, cols, reinterpret_cast<__bf16*>(dst), reinterpret_cast<__bf16*>(dstT),
                     partial_ws, PackConv{0, 0, 0, 0, 0, 0, 0});
  // partial_ws is [grid.y][grid.x * 64]: the finish kernel walks it with that row length
  hipLaunchKernelGGL(colsum_rows_finish_kernel, dim3((cols + 255) / 256), dim3(256), 0, ST(stream), partial_ws, (int)grid.y, cols, colsum, accumulate,
                     (int)grid.x * 64);
  SP3_LAUNCH_CHECK("sp3_pack_bf16_colsum");
  return 0;
}

extern "C" int64_t sp3_colsum_rows_ws(int rows, int N) { return (int64_t)((rows + kColChunk - 1) / kColChunk) * N; }

extern "C" int sp3_colsum_rows(const float* x, int64_t ld, int rows, int N, float* out, int accumulate, float* scratch, void* stream) {
  SP3_CHECK(x && out && scratch && rows > 0 && N > 0 && ld >= N, "sp3_colsum_rows: bad arguments");
  if (rows <= 8192) {
    hipLaunchKernelGGL(colsum_rows_single_kernel, dim3((N + 63) / 64), dim3(1024), 0, ST(stream), x, ld, rows, N, out, accumulate);
    SP3_LAUNCH_CHECK("sp3_colsum_rows");
    return 0;
  }
  const int nchunk = (rows + kColChunk - 1) / kColChunk;
  SP3_CHECK(nchunk <= 65535, "sp3_colsum_rows: too many rows (%d)", rows);
  hipLaunchKernelGGL(colsum_rows_partial_kernel, dim3((N + 63) / 64, nchunk), dim3(256), 0, ST(stream), x, ld, rows, N, scratch);
  hipLaunchKernelGGL(colsum_rows_finish_kernel, dim3((N + 255) / 256), dim3(256), 0, ST(stream), scratch, nchunk, N, out, accumulate, N);
  SP3_LAUNCH_CHECK("sp3_colsum_rows");
  return 0;
}

extern "C" int64_t sp3_sumsq_blocks(int64_t n) { return (n + kSumChunk - 1) / kSumChunk; }

extern "C" int sp3_sumsq_partial(const float* g, int64_t n, double* partial, void* stream) {
  SP3_CHECK(g && partial && n > 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0, "sp3_sumsq_partial: bad arguments");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)sp3_sumsq_blocks(n)), dim3(256), 0, ST(stream), g, n, partial);
  SP3_LAUNCH_CHECK("sp3_sumsq_partial");
  return 0;
}

extern "C" int sp3_clip_coef(const double* partial, int count, float max_norm, float extra_scale, float* out, int* step_counter, void* stream) {
  SP3_CHECK(partial && out && count > 0, "sp3_clip_coef: bad arguments");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, ST(stream), partial, count, max_norm, extra_scale, out, step_counter);
  SP3_LAUNCH_CHECK("sp3_clip_coef");
  return 0;
}

extern "C" int sp3_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, const float* chunk_table, float lr, float beta1,
                              float beta2, float eps, int step, const float* grad_scale_dev, float grad_scale, const int* step_dev,
                              const float* lr_dev, void* stream) {
  SP3_CHECK(p && g && m && v && chunk_table && n > 0 && n % 1024 == 0 && (step >= 1 || step_dev), "sp3_adamw_flat: bad arguments (n=%lld must be a multiple of 1024)", (long long)n);
  const int st_ = step >= 1 ? step : 1;
  const float bc1 = 1.0f - powf(beta1, (float)st_), bc2 = 1.0f - powf(beta2, (float)st_);
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)(n / 1024)), dim3(256), 0, ST(stream), p, g, m, v, n,
                     reinterpret_cast<const float2*>(chunk_table), lr, beta1, beta2, eps, bc1, 1.0f / sqrtf(bc2), grad_scale_dev, grad_scale,
                     step_dev, lr_dev);
  SP3_LAUNCH_CHECK("sp3_adamw_flat");
  return 0;
}

extern "C" int sp3_head_shuffle(const sp3_head_part* parts, int nparts, int B, int H, int hd, float base, const float* cos_tab,
                                const float* sin_tab, int tab_len, void* stream) {
  SP3_CHECK(parts && nparts >= 1 && nparts <= 3 && B > 0 && H > 0 && hd > 0 && hd <= 64 && hd % 4 == 0 && (int64_t)B * H <= 65535,
            "sp3_head_shuffle: bad arguments (nparts=%d B=%d H=%d hd=%d)", nparts, B, H, hd);
  HeadShuffleArgs a;
  a.B = B; a.H = H; a.hd = hd; a.base = base;
  a.cos_tab = cos_tab; a.sin_tab = sin_tab; a.tab_len = (cos_tab && sin_tab) ? tab_len : 0;
  int nmax = 0;
  for (int i = 0; i < nparts; ++i) {
    const sp3_head_part& p = parts[i];
    SP3_CHECK(p.src && (p.dst || p.dstT) && p.N > 0, "sp3_head_shuffle: part %d: null pointer or N <= 0", i);
    SP3_CHECK(((p.s_b | p.s_n | p.s_h) & 3) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 15) == 0, "sp3_head_shuffle: part %d: source not 16-byte aligned", i);
    SP3_CHECK(!p.dst || ((((p.d_b | p.d_n | p.d_h) & 3) == 0) && (reinterpret_cast<uintptr_t>(p.dst) & 15) == 0),
              "sp3_head_shuffle: part %d: destination not 16-byte aligned", i);
    SP3_CHECK(!p.dstT || ((reinterpret_cast<uintptr_t>(p.dstT) & 15) == 0 && (p.ldT == 0 || (p.ldT % 4 == 0 && p.ldT >= p.N && p.ldT <= (p.N + 63) / 64 * 64))),
              "sp3_head_shuffle: part %d: transposed destination (16-byte aligned, N <= ldT <= N rounded up to 64, ldT %% 4 == 0)", i);
    SP3_CHECK(!p.pos || (p.fwd == 1.0f || p.fwd == -1.0f), "sp3_head_shuffle: part %d: fwd must be +1 or -1", i);
    a.p[i] = p;
    nmax = p.N > nmax ? p.N : nmax;
  }
  for (int i = nparts; i < 3; ++i) a.p[i] = parts[0];
  hipLaunchKernelGGL(head_shuffle_kernel, dim3((nmax + 63) / 64, B * H, nparts), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  SP3_LAUNCH_CHECK("sp3_head_shuffle");
  return 0;
}
