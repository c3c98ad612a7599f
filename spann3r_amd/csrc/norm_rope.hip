// LayerNorm (row-major and transposed-store) and the stand-alone 2-D RoPE kernel (curope drop-in).
#include "common.h"
#include <math.h>

namespace {

constexpr int LN_MAX_V4 = 16;   // C <= 64 lanes * 16 float4 * 4 = 4096

// One wave per row; the row lives in registers; two-pass mean / variance in fp32
// (torch.nn.functional.layer_norm numerics: var = mean((x-mean)^2), biased).
__device__ __forceinline__ void ln_row(const float* __restrict__ xr, int C, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float eps, int lane,
                                       float4 (&v)[LN_MAX_V4], int& nv) {
  const int c4 = C >> 2;
  nv = 0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int j = lane + i * 64;
    if (j < c4) {
      v[i] = reinterpret_cast<const float4*>(xr)[j];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      nv = i + 1;
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int j = lane + i * 64;
    if (j < c4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int j = lane + i * 64;
    if (j < c4) {
      const float4 gm = reinterpret_cast<const float4*>(gamma)[j];
      const float4 bt = reinterpret_cast<const float4*>(beta)[j];
      v[i].x = (v[i].x - mean) * rstd * gm.x + bt.x;
      v[i].y = (v[i].y - mean) * rstd * gm.y + bt.y;
      v[i].z = (v[i].z - mean) * rstd * gm.z + bt.z;
      v[i].w = (v[i].w - mean) * rstd * gm.w + bt.w;
    }
  }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, void* __restrict__ out, int64_t ldo, int out_bf16,
                                                        int out_packed, int rows, int C, __bf16* __restrict__ dual = nullptr,
                                                        int group_rows = 0, int group_pad = 0) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 v[LN_MAX_V4];
  int nv;
  ln_row(x + (int64_t)row * ldx, C, gamma, beta, eps, lane, v, nv);
  const int c4 = C >> 2;
  if (dual) {
    // second output: bf16 fragment-order copy, every group of group_rows rows starting on a 16-row boundary (group_pad)
    const int gi = row / group_rows, prow = gi * group_pad + (row - gi * group_rows);
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
      const int j = lane + i * 64;
      if (j < c4) {
        bf16x4 o;
        o[0] = (__bf16)v[i].x; o[1] = (__bf16)v[i].y; o[2] = (__bf16)v[i].z; o[3] = (__bf16)v[i].w;
        *reinterpret_cast<bf16x4*>(dual + packed_off(prow, 4 * j, C, true)) = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int j = lane + i * 64;
    if (j < c4) {
      const int64_t off = out_packed ? packed_off(row, 4 * j, C, out_bf16 != 0) : (int64_t)row * ldo + 4 * j;
      if (out_bf16) {
        bf16x4 o;
        o[0] = (__bf16)v[i].x; o[1] = (__bf16)v[i].y; o[2] = (__bf16)v[i].z; o[3] = (__bf16)v[i].w;
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(out) + off) = o;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + off) = v[i];
      }
    }
  }
}

// Transposed store: a block normalises 8 consecutive rows (4 waves x 2 rows), parks them in LDS and
// writes out[c*ldo + row0 .. row0+7] as 8-element contiguous runs.
constexpr int LNT_ROWS = 8;    // 1024 x 9 floats = 36 KB of LDS
__global__ __launch_bounds__(256) void layernorm_t_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, void* __restrict__ out, int64_t ldo, int out_bf16,
                                                          int rows, int C) {
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [C][LNT_ROWS+1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * LNT_ROWS;
  const int c4 = C >> 2;
  for (int rr = 0; rr < LNT_ROWS / 4; ++rr) {
    const int lr = wave * (LNT_ROWS / 4) + rr;
    const int row = row0 + lr;
    if (row < rows) {
      float4 v[LN_MAX_V4];
      int nv;
      ln_row(x + (int64_t)row * ldx, C, gamma, beta, eps, lane, v, nv);
#pragma unroll
      for (int i = 0; i < LN_MAX_V4; ++i) {
        const int j = lane + i * 64;
        if (j < c4) {
          tile[(j * 4 + 0) * (LNT_ROWS + 1) + lr] = v[i].x;
          tile[(j * 4 + 1) * (LNT_ROWS + 1) + lr] = v[i].y;
          tile[(j * 4 + 2) * (LNT_ROWS + 1) + lr] = v[i].z;
          tile[(j * 4 + 3) * (LNT_ROWS + 1) + lr] = v[i].w;
        }
      }
    }
  }
  __syncthreads();
  const int nr = (rows - row0) < LNT_ROWS ? (rows - row0) : LNT_ROWS;
  for (int idx = threadIdx.x; idx < C * LNT_ROWS; idx += 256) {
    const int c = idx / LNT_ROWS, r = idx % LNT_ROWS;
    if (r < nr) {
      const float val = tile[c * (LNT_ROWS + 1) + r];
      if (out_bf16) reinterpret_cast<__bf16*>(out)[(int64_t)c * ldo + row0 + r] = (__bf16)val;
      else reinterpret_cast<float*>(out)[(int64_t)c * ldo + row0 + r] = val;
    }
  }
}

// ------------------------------------------------------------------ split-K finish + residual + LayerNorm(s)
// One wave per row.  x = sum_s partial[s][row] + bias + res[row]; optional x_out; up to two LayerNorms of x
// (they share mean / rstd).  Same two-pass fp32 statistics as layernorm_kernel.
struct ReduceLnArgs { sp3_reduce_ln_desc d; };

// store 4 consecutive columns (4j .. 4j+3) of `row`; packed != 0 -> fragment order of a [rows, C] GEMM operand
__device__ __forceinline__ void store_row4(void* out, int64_t ld, int bf, int packed, int C, int row, int j, float4 v) {
  const int64_t off = packed ? packed_off(row, 4 * j, C, bf != 0) : (int64_t)row * ld + 4 * j;
  if (bf) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(out) + off) = o;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + off) = v;
  }
}

__global__ __launch_bounds__(256) void reduce_ln_kernel(const ReduceLnArgs args) {
  const sp3_reduce_ln_desc& d = args.d;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= d.rows) return;
  const int C = d.C, c4 = C >> 2;
  float4 v[LN_MAX_V4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int j = lane + i * 64;
    if (j < c4) {
      float4 x = reinterpret_cast<const float4*>(d.partial + (int64_t)row * C)[j];
      for (int sp = 1; sp < d.splits; ++sp) {
        const float4 t = reinterpret_cast<const float4*>(d.partial + sp * d.split_stride + (int64_t)row * C)[j];
        x.x += t.x; x.y += t.y; x.z += t.z; x.w += t.w;
      }
      if (d.bias) {
        const float4 b = reinterpret_cast<const float4*>(d.bias)[j];
        x.x += b.x; x.y += b.y; x.z += b.z; x.w += b.w;
      }
      if (d.act == SP3_ACT_RELU) x = relu4(x);
      if (d.res) {
        const float4 r = reinterpret_cast<const float4*>(d.res + (int64_t)row * d.ldres)[j];
        x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w;
      }
      if (d.res2) {
        const float4 r = reinterpret_cast<const float4*>(d.res2 + (int64_t)row * d.ldres2)[j];
        x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w;
      }
      if (d.x_out) reinterpret_cast<float4*>(d.x_out + (int64_t)row * d.ldx)[j] = x;
      v[i] = x;
      s += (x.x + x.y) + (x.z + x.w);
    }
  }
  if (!d.out1 && !d.out2) return;
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int j = lane + i * 64;
    if (j < c4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
      q += (a * a + b * b) + (c * c + e * e);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + d.eps);
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int j = lane + i * 64;
    if (j < c4) {
      float4 n;
      n.x = (v[i].x - mean) * rstd; n.y = (v[i].y - mean) * rstd; n.z = (v[i].z - mean) * rstd; n.w = (v[i].w - mean) * rstd;
      if (d.out1) {
        const float4 gm = reinterpret_cast<const float4*>(d.g1)[j], bt = reinterpret_cast<const float4*>(d.b1)[j];
        store_row4(d.out1, d.ld1, d.out1_bf16, d.out1_packed, C, row, j, make_float4(n.x * gm.x + bt.x, n.y * gm.y + bt.y, n.z * gm.z + bt.z, n.w * gm.w + bt.w));
      }
      if (d.out2) {
        const float4 gm = reinterpret_cast<const float4*>(d.g2)[j], bt = reinterpret_cast<const float4*>(d.b2)[j];
        store_row4(d.out2, d.ld2, d.out2_bf16, d.out2_packed, C, row, j, make_float4(n.x * gm.x + bt.x, n.y * gm.y + bt.y, n.z * gm.z + bt.z, n.w * gm.w + bt.w));
      }
    }
  }
}

// ------------------------------------------------------------------ stand-alone RoPE (curope drop-in)
// One wave handles one token (b, n) for all heads.  Lane i < D/2 owns the pair (u_i, v_i):
// i in [0,Q) -> Y quarter pair (d=i, d=i+Q); i in [Q,2Q) -> X pair (d=2Q+i-Q, d=3Q+i-Q).
template <typename T>
__global__ __launch_bounds__(256) void rope2d_kernel(T* __restrict__ tokens, int B, int N, int H, int D, int64_t sB,
                                                     int64_t sN, int64_t sH, const int64_t* __restrict__ pos,
                                                     float base, float fwd) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= B * N) return;
  const int b = tok / N, n = tok - b * N;
  const int Q = D >> 2;
  T* t0 = tokens + (int64_t)b * sB + (int64_t)n * sN;
  for (int i = lane; i < 2 * Q; i += 64) {
    const int axis = i / Q, f = i - axis * Q;
    const float p = (float)pos[(int64_t)tok * 2 + axis];
    const float inv_freq = fwd / powf(base, (float)f / (float)Q);
    const float ang = p * inv_freq;
    const float cs = cosf(ang), sn = sinf(ang);
    const int du = axis * 2 * Q + f, dv = du + Q;
    for (int h = 0; h < H; ++h) {
      T* th = t0 + (int64_t)h * sH;
      const float u = (float)th[du], v = (float)th[dv];
      th[du] = (T)(u * cs - v * sn);
      th[dv] = (T)(v * cs + u * sn);
    }
  }
}

}  // namespace

static int ln_check(const float* x, int64_t ldx, const float* gamma, const float* beta, void* out, int rows, int C) {
  SP3_CHECK(x && gamma && beta && out, "sp3_layernorm: null pointer");
  SP3_CHECK(rows > 0 && C > 0 && C % 4 == 0 && C <= 64 * 4 * LN_MAX_V4, "sp3_layernorm: bad rows=%d C=%d", rows, C);
  SP3_CHECK(ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "sp3_layernorm: x must be 16-byte aligned rows");
  return 0;
}

extern "C" int sp3_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* out,
                             int64_t ldo, int out_bf16, int rows, int C, void* stream) {
  if (ln_check(x, ldx, gamma, beta, out, rows, C)) return 1;
  SP3_CHECK(ldo % 4 == 0, "sp3_layernorm: ldo must be a multiple of 4");
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx,
                     gamma, beta, eps, out, ldo, out_bf16, 0, rows, C);
  SP3_LAUNCH_CHECK("sp3_layernorm");
  return 0;
}

extern "C" int sp3_layernorm_packed(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* out,
                                    int out_bf16, int rows, int C, void* stream) {
  if (ln_check(x, ldx, gamma, beta, out, rows, C)) return 1;
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx,
                     gamma, beta, eps, out, (int64_t)0, out_bf16, 1, rows, C);
  SP3_LAUNCH_CHECK("sp3_layernorm_packed");
  return 0;
}

extern "C" int sp3_layernorm_dual(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, float* out, int64_t ldo,
                                  void* out_packed_bf16, int rows, int C, int group_rows, int group_rows_pad, void* stream) {
  if (ln_check(x, ldx, gamma, beta, out, rows, C)) return 1;
  SP3_CHECK(ldo % 4 == 0 && out_packed_bf16, "sp3_layernorm_dual: ldo must be a multiple of 4, the packed output non-null");
  SP3_CHECK(group_rows > 0 && group_rows_pad >= group_rows && group_rows_pad % 16 == 0, "sp3_layernorm_dual: bad row groups %d -> %d",
            group_rows, group_rows_pad);
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx,
                     gamma, beta, eps, static_cast<void*>(out), ldo, 0, 0, rows, C, reinterpret_cast<__bf16*>(out_packed_bf16), group_rows,
                     group_rows_pad);
  SP3_LAUNCH_CHECK("sp3_layernorm_dual");
  return 0;
}

extern "C" int sp3_reduce_ln(const sp3_reduce_ln_desc* dp, void* stream) {
  SP3_CHECK(dp && dp->partial, "sp3_reduce_ln: null descriptor / partial");
  const sp3_reduce_ln_desc& d = *dp;
  SP3_CHECK(d.rows > 0 && d.C > 0 && d.C % 4 == 0 && d.C <= 64 * 4 * LN_MAX_V4 && d.splits >= 1, "sp3_reduce_ln: bad rows=%d C=%d splits=%d", d.rows, d.C, d.splits);
  SP3_CHECK(!d.out1 || (d.g1 && d.b1 && (d.out1_packed || d.ld1 % 4 == 0)), "sp3_reduce_ln: LayerNorm 1 needs gamma/beta");
  SP3_CHECK(!d.out2 || (d.g2 && d.b2 && (d.out2_packed || d.ld2 % 4 == 0)), "sp3_reduce_ln: LayerNorm 2 needs gamma/beta");
  SP3_CHECK(!d.res || d.ldres % 4 == 0, "sp3_reduce_ln: ldres must be a multiple of 4");
  SP3_CHECK(!d.x_out || d.ldx % 4 == 0, "sp3_reduce_ln: ldx must be a multiple of 4");
  ReduceLnArgs a;
  a.d = d;
  hipLaunchKernelGGL(reduce_ln_kernel, dim3((d.rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  SP3_LAUNCH_CHECK("sp3_reduce_ln");
  return 0;
}

extern "C" int sp3_layernorm_t(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* out,
                               int64_t ldo, int out_bf16, int rows, int C, void* stream) {
  if (ln_check(x, ldx, gamma, beta, out, rows, C)) return 1;
  const size_t lds = (size_t)C * (LNT_ROWS + 1) * sizeof(float);
  SP3_CHECK(lds <= 160 * 1024, "sp3_layernorm_t: C=%d too large for LDS", C);
  hipLaunchKernelGGL(layernorm_t_kernel, dim3((rows + LNT_ROWS - 1) / LNT_ROWS), dim3(256), lds,
                     reinterpret_cast<hipStream_t>(stream), x, ldx, gamma, beta, eps, out, ldo, out_bf16, rows, C);
  SP3_LAUNCH_CHECK("sp3_layernorm_t");
  return 0;
}

extern "C" int sp3_rope_2d(void* tokens, int dtype, int B, int N, int H, int D, int64_t sB, int64_t sN, int64_t sH,
                           const int64_t* positions, float base, float fwd, void* stream) {
  // same argument checks as curope.cpp:54-59 / kernels.cu:91-94
  SP3_CHECK(tokens && positions, "rope_2d: null pointer");
  SP3_CHECK(B > 0 && N > 0 && H > 0 && D > 0, "rope_2d: bad shape");
  SP3_CHECK(D % 4 == 0, "token dim must be multiple of 4");
  SP3_CHECK(dtype == SP3_F32 || dtype == SP3_BF16, "rope_2d: bad dtype %d", dtype);
  const int blocks = (B * N + 3) / 4;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SP3_F32)
    hipLaunchKernelGGL(rope2d_kernel<float>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<float*>(tokens), B, N, H, D,
                       sB, sN, sH, positions, base, fwd);
  else
    hipLaunchKernelGGL(rope2d_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<__bf16*>(tokens), B, N, H,
                       D, sB, sN, sH, positions, base, fwd);
  SP3_LAUNCH_CHECK("sp3_rope_2d");
  return 0;
}
