// sp3_conv3x3_tile : 3x3 / stride 1 / pad 1 convolution on NHWC maps as an LDS-tiled implicit GEMM (bf16 MFMA).
//
// Why a second conv path (sp3_gemm's LOAD_CONV3X3 stays for fp32 mode, stride 2 and odd channel counts): the generic
// loader re-reads every input pixel from L2 once per tap and per output-channel tile (9 x Cout/32 times) with one cache
// line per lane quad, so the DPT head convolutions were L2/TA-bound at ~110 TFLOP/s.  Here a workgroup owns an 8 x 8
// output tile x 64 output channels:
//   * the (8+2) x (8+2) input halo tile, all Cin channels, is fetched ONCE with coalesced 16-byte loads, gets the
//     optional input ReLU and the bf16 rounding once, and lives in LDS (pixel stride Cin*2 + 16 bytes: conflict-free
//     ds_read_b128 for the 16 pixels of an MFMA row block);
//   * K = 9 * Cin is cut into units of 32 (one v_mfma_f32_16x16x32_bf16 deep), interleaved over the 4 waves; a wave
//     multiplies the full 64 x 64 tile for its units (16 accumulator fragments), reading A fragments from LDS and
//     streaming its W fragments straight from the fragment-order weight (each W byte is loaded once per workgroup);
//   * the 4 partial tiles are summed through LDS (re-using the halo bytes) by the coalesced epilogue
//     (bias, ReLU, up to two residual maps, fp32 or bf16 store).
// Reference ops: croco/models/dpt_block.py:33-75 (ResidualConvUnit), 95-113 (head convs), 180-188 (layer_rn).
#include "common.h"
#include <type_traits>

namespace {

constexpr int TH = 8, TW = 8;                 // output tile (pixels)
constexpr int HP = (TH + 2) * (TW + 2);       // halo pixels
constexpr int BM = TH * TW, BN = 64;
constexpr int MF = BM / 16, NF = BN / 16;
constexpr int SLAB_LD = BN + 4;
constexpr int DEPTH = 3;                      // W units in flight per wave

struct ConvArgs {
  const void* x; const __bf16* w; const float* bias; const float* res1; const float* res2; void* out;
  int B, H, W, Cin, Cout, tiles_x, tiles_y, relu_in, act, out_bf16;
};

__device__ __forceinline__ bf16x8 ld_frag_lds(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

template <typename TIN>
__global__ __launch_bounds__(256) void conv3x3_tile_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
  const int Cin = a.Cin;
  const int PS = Cin * 2 + 16;                // halo pixel stride in bytes
  int t = blockIdx.x;
  const int tx = t % a.tiles_x; t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = blockIdx.y * BN;

  // ---- W stream set-up: unit u = (tap, 32-channel group); fragment-order weight [Cout/16][K/64][2 halves][64 lanes][8]
  const int upt = Cin >> 5;                   // units per tap
  const int U = 9 * upt;
  const int nkb = (9 * Cin) >> 6;
  const __bf16* wb[NF];
#pragma unroll
  for (int n = 0; n < NF; ++n) wb[n] = a.w + ((int64_t)((n0 >> 4) + n) * nkb * 128 + lane) * 8;
  const int nu = (U - wave + 3) >> 2;         // units of this wave: wave, wave + 4, ...
  auto w_off = [&](int i) -> int64_t {        // element offset of unit #i of this wave inside a 16-row fragment panel
    int u = wave + 4 * i;
    u = u < U ? u : U - 1;                    // clamped: the tail re-loads the last unit instead of branching
    return (int64_t)(u >> 1) * (64 * 16) + (u & 1) * (64 * 8);     // K-block (u >> 1) [u counts 32-deep halves], 1 KB half u & 1
  };
  bf16x8 wq[DEPTH][NF];
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) {
    const int64_t o = w_off(s);
#pragma unroll
    for (int n = 0; n < NF; ++n) wq[s][n] = *reinterpret_cast<const bf16x8*>(wb[n] + o);
  }

  // ---- halo tile -> LDS (ReLU + bf16 rounding once per element; out-of-image pixels are zeros)
  {
    constexpr int CHK = 16 / (int)sizeof(TIN);          // elements per 16-byte chunk
    const int cpp = Cin / CHK;                          // chunks per pixel
    const TIN* xin = reinterpret_cast<const TIN*>(a.x) + (int64_t)b * a.H * a.W * Cin;
    using Chunk = typename std::conditional<sizeof(TIN) == 4, float4, bf16x8>::type;
    auto src_of = [&](int p, int c, bool& ok) -> const Chunk* {
      const int hy = p / (TW + 2), hx = p - hy * (TW + 2);
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      ok = p < HP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const int cy = ok ? iy : 0, cx = ok ? ix : 0;
      return reinterpret_cast<const Chunk*>(xin + ((int64_t)cy * a.W + cx) * Cin + c);
    };
    auto put = [&](int p, int c, Chunk v, bool ok) {
      char* dst = lds + p * PS + c * 2;
      if constexpr (sizeof(TIN) == 4) {
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.relu_in) v = relu4(v);
        bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(dst) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[e];
          v[e] = (ok && (!a.relu_in || f > 0.f)) ? v[e] : (__bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(dst) = v;
      }
    };
    if ((256 % cpp) == 0) {
      // the hot shapes: a thread keeps its channel chunk and walks the pixels; 8 loads are in flight per thread
      const int c = (tid % cpp) * CHK, pstep = 256 / cpp;
      constexpr int NB = 8;
      for (int p0 = tid / cpp; p0 < HP; p0 += NB * pstep) {
        Chunk buf[NB];
        bool ok[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) buf[q] = *src_of(p0 + q * pstep, c, ok[q]);
#pragma unroll
        for (int q = 0; q < NB; ++q)
          if (p0 + q * pstep < HP) put(p0 + q * pstep, c, buf[q], ok[q]);
      }
    } else {
      for (int i = tid; i < HP * cpp; i += 256) {
        const int p = i / cpp, c = (i - p * cpp) * CHK;
        bool ok;
        const Chunk v = *src_of(p, c, ok);
        put(p, c, v, ok);
      }
    }
  }
  __syncthreads();

  // ---- K loop: A fragments from the halo tile, W fragments from the register queue
  f32x4 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // lane (g, r) of row block m is output pixel p = 16 m + r -> (p / TW, p % TW); its halo pixel for tap (ky, kx) is
  // (py + ky, px + kx); channels 32 (u % upt) + 8 g .. + 7 ... matched to the W lane: K-block half h holds k = 16 g + 8 h + j
  int abase[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    const int p = 16 * m + r;
    abase[m] = ((p / TW) * (TW + 2) + (p % TW)) * PS;
  }
  auto a_off = [&](int i) -> int {
    int u = wave + 4 * i;
    u = u < U ? u : U - 1;
    const int tap = u / upt, c32 = u - tap * upt;
    const int ky = tap / 3, kx = tap - 3 * ky;
    // channel base inside the tap: the 64-block (c32 >> 1) * 64, lane group g -> + 16 g, half (c32 & 1) -> + 8
    return (ky * (TW + 2) + kx) * PS + (((c32 >> 1) << 6) + 16 * g + ((c32 & 1) << 3)) * 2;
  };
  bf16x8 af[2][MF];
  {
    const int o = a_off(0);
#pragma unroll
    for (int m = 0; m < MF; ++m) af[0][m] = ld_frag_lds(lds + abase[m] + o);
  }
  for (int i0 = 0; i0 < nu; i0 += 2 * DEPTH) {
#pragma unroll
    for (int s2 = 0; s2 < 2 * DEPTH; ++s2) {
      const int i = i0 + s2;
      if (i < nu) {
        const int s = s2 % DEPTH, cur = s2 & 1;
        {   // A of the next unit
          const int o = a_off(i + 1);
#pragma unroll
          for (int m = 0; m < MF; ++m) af[cur ^ 1][m] = ld_frag_lds(lds + abase[m] + o);
        }
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int n = 0; n < NF; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][m], wq[s][n], acc[m][n], 0, 0, 0);
        {   // refill this queue slot with unit i + DEPTH
          const int64_t o = w_off(i + DEPTH);
#pragma unroll
          for (int n = 0; n < NF; ++n) wq[s][n] = *reinterpret_cast<const bf16x8*>(wb[n] + o);
        }
      }
    }
  }
  __syncthreads();                            // every wave is done with the halo bytes: re-use them for the partial tiles

  // ---- partial tiles -> LDS (C layout: col = lane & 15, row = 4 g + reg), then the coalesced epilogue
  float* slab = reinterpret_cast<float*>(lds) + (size_t)wave * BM * SLAB_LD;
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) slab[(m * 16 + 4 * g + q) * SLAB_LD + n * 16 + r] = acc[m][n][q];
  __syncthreads();
  const float* sm = reinterpret_cast<const float*>(lds);
  for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
    const int row = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
    const int oy = y0 + row / TW, ox = x0 + row % TW;
    if (oy >= a.H || ox >= a.W) continue;
    float4 v = *reinterpret_cast<const float4*>(sm + row * SLAB_LD + c4);
#pragma unroll
    for (int s = 1; s < 4; ++s) {
      const float4 t4 = *reinterpret_cast<const float4*>(sm + (size_t)s * BM * SLAB_LD + row * SLAB_LD + c4);
      v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
    }
    const int gn = n0 + c4;
    if (a.bias) { const float4 b4 = *reinterpret_cast<const float4*>(a.bias + gn); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
    if (a.act == SP3_ACT_RELU) v = relu4(v);
    const int64_t off = (((int64_t)b * a.H + oy) * a.W + ox) * a.Cout + gn;
    if (a.out_bf16 & 2) {                     // residual maps stored as bf16 (bf16 mode of the DPT heads)
      if (a.res1) { const bf16x4 q = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(a.res1) + off); v.x += (float)q[0]; v.y += (float)q[1]; v.z += (float)q[2]; v.w += (float)q[3]; }
      if (a.res2) { const bf16x4 q = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(a.res2) + off); v.x += (float)q[0]; v.y += (float)q[1]; v.z += (float)q[2]; v.w += (float)q[3]; }
    } else {
      if (a.res1) { const float4 q = *reinterpret_cast<const float4*>(a.res1 + off); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
      if (a.res2) { const float4 q = *reinterpret_cast<const float4*>(a.res2 + off); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    }
    if (a.out_bf16 & 1) {
      bf16x4 o;
      o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
      *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.out) + off) = o;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + off) = v;
    }
  }
}

}  // namespace

extern "C" int sp3_conv3x3_tile(const void* x, int in_bf16, const void* w_packed, const float* bias, const float* res1,
                                const float* res2, void* out, int out_bf16, int B, int H, int W, int Cin, int Cout,
                                int relu_in, int act, void* stream) {
  SP3_CHECK(x && w_packed && out, "sp3_conv3x3_tile: null pointer");
  SP3_CHECK(B > 0 && H > 0 && W > 0, "sp3_conv3x3_tile: bad shape");
  SP3_CHECK(Cin % 64 == 0 && Cin >= 64 && Cout % 64 == 0 && Cout >= 64, "sp3_conv3x3_tile: Cin=%d / Cout=%d must be multiples of 64", Cin, Cout);
  SP3_CHECK(act == SP3_ACT_NONE || act == SP3_ACT_RELU, "sp3_conv3x3_tile: act %d", act);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  SP3_CHECK(al16(x) && al16(w_packed) && al16(bias) && al16(res1) && al16(res2) && al16(out), "sp3_conv3x3_tile: 16-byte alignment");
  ConvArgs a;
  a.x = x; a.w = reinterpret_cast<const __bf16*>(w_packed); a.bias = bias; a.res1 = res1; a.res2 = res2; a.out = out;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
  a.relu_in = relu_in; a.act = act; a.out_bf16 = out_bf16;
  const size_t halo = (size_t)HP * (Cin * 2 + 16), slabs = (size_t)4 * BM * SLAB_LD * sizeof(float);
  const size_t lds = halo > slabs ? halo : slabs;
  SP3_CHECK(lds <= 160 * 1024, "sp3_conv3x3_tile: Cin=%d needs %zu bytes of LDS", Cin, lds);
  dim3 grid(a.tiles_x * a.tiles_y * B, Cout / BN);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  auto launch = [&](auto kern) -> int {
    static size_t raised = 0;                 // per instantiation: opt in to > 64 KiB of dynamic LDS once per size
    if (lds > 64 * 1024 && lds > raised) {
      raised = lds;
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { sp3_set_error("sp3_conv3x3_tile: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return 2; }
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    return 0;
  };
  const int rc = in_bf16 ? launch(conv3x3_tile_kernel<__bf16>) : launch(conv3x3_tile_kernel<float>);
  if (rc) return rc;
  SP3_LAUNCH_CHECK("sp3_conv3x3_tile");
  return 0;
}
