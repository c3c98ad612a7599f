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
#include <cstdlib>

namespace {

constexpr int TH = 8, TW = 8;                 // output tile (pixels)
constexpr int HP = (TH + 2) * (TW + 2);       // halo pixels
constexpr int BM = TH * TW, BN = 64;
constexpr int MF = BM / 16, NF = BN / 16;
constexpr int SLAB_LD = BN + 4;
constexpr int DEPTH = 3;                      // W units in flight per wave
constexpr int SP3_CONV_WIDE_MIN_WGS = 256;    // 8 x 16 x 64-channel tile from this many workgroups up, the 32-channel instance from 80 (measured: profiles/r04_conv_tile_sweep.txt)

struct ConvArgs {
  const void* x; const __bf16* w; const float* bias; const float* res1; const float* res2; void* out;
  int B, H, W, Cin, Cout, tiles_x, tiles_y, relu_in, act, out_bf16;
  int xcd_nb;      // > 0: 1-D grid, XCD x (= workgroup id % 8) owns the pixel tiles {x, x + 8, ..} with all their xcd_nb output-channel blocks
};

// workgroup -> (pixel tile, output-channel block).  Plain grid: x = pixel tile, y = channel block -- the channel blocks of a pixel tile
// are gridDim.x apart in dispatch order and land on different XCDs, so every L2 fetches the tile's halo (PMC: 2.6 x the algorithmic
// bytes per launch).  xcd_nb: the channel blocks of a tile are consecutive workgroups of ONE XCD; returns false for the padding
// workgroups of the last group of 8 tiles.
__device__ __forceinline__ bool conv_tile_of(const ConvArgs& a, int& t, int& nblk) {
  t = blockIdx.x; nblk = blockIdx.y;
  if (a.xcd_nb) {
    const unsigned L = blockIdx.x, xc = L & 7, jj = L >> 3, q = jj / (unsigned)a.xcd_nb;
    nblk = (int)(jj - q * a.xcd_nb);
    t = (int)(q * 8 + xc);
    if (t >= a.tiles_x * a.tiles_y * a.B) return false;
  }
  return true;
}

__device__ __forceinline__ bf16x8 ld_frag_lds(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

template <typename TIN>
__global__ __launch_bounds__(256) void conv3x3_tile_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
  const int Cin = a.Cin;
  const int PS = Cin * 2 + 16;                // halo pixel stride in bytes
  int t, nblk;
  if (!conv_tile_of(a, t, nblk)) return;
  const int tx = t % a.tiles_x; t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = nblk * BN;

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
      // the hot shapes: a thread keeps its channel chunk and walks the pixels; 13 loads are in flight per thread (bf16 maps with
      // 256 channels: the whole halo in one batch, i.e. one memory latency -- the wide kernel's ablation shows what batches cost)
      const int c = (tid % cpp) * CHK, pstep = 256 / cpp;
      constexpr int NB = 13;
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
      st_out(reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.out) + off), o);
    } else {
      st_out(reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + off), v);
    }
  }
}

// ---- the wide variant: 8 rows x 16 pixels x 64 output channels per workgroup -------------------------------------------------
// Every W fragment a workgroup multiplies has to come through its CU's vector-memory path once (it is used by one wave only),
// so the W bytes per flop are 1 / (pixels per tile): with 64 pixels the kernel above is bound by that stream (2 x 295 KB per CU
// and round at Cin = 256, ~70 KB/us per CU), not by MFMA.  Here the tile holds 128 pixels -- half the W bytes per flop -- and
// the LDS foot-print stays at two workgroups per CU because
//   * the halo tile is staged 128 input channels at a time (10 x 18 pixels x 272 bytes = 48 KB); K runs chunk by chunk, inside a
//     chunk wave w owns the 32-deep channel unit w of every tap (9 units, 288 MFMAs per chunk), so each wave's LDS reads and
//     its W stream are fixed-stride and fully unrolled; the W queue runs across chunk boundaries;
//   * A fragments are single-buffered: fragment m is re-loaded for the next tap right behind its last MFMA of this one;
//   * the four partial tiles are summed through LDS in two passes of 64 pixels (4 x 64 x 68 floats = 68 KB).
// 128 accumulator registers per lane; __launch_bounds__(256, 2) keeps two waves per SIMD.
namespace wide {
constexpr int TH = 8, TW = 16, HW = TW + 2, HP = (TH + 2) * HW;   // 180 halo pixels
constexpr int CH = 128, PS = CH * 2 + 16;                          // channels per staged chunk, halo pixel stride (bytes)
constexpr int MF = 8, DEPTH = 3;
constexpr int HALO_BYTES = HP * PS;
template <int NF> constexpr int lds_bytes() {                     // halo chunk, re-used by the 4 partial tiles of one 64-pixel pass
  return HALO_BYTES > 4 * 64 * (16 * NF + 4) * 4 ? HALO_BYTES : 4 * 64 * (16 * NF + 4) * 4;
}

// ABL (tools/ubench/conv_wide.hip only; 0 in the library): ablation bits -- 1 no W refills, 2 no A re-loads per tap, 4 no halo loads
// from memory, 8 no epilogue (a dummy consumer keeps the MFMAs alive)
// NF = 16-channel column blocks per workgroup: 4 (64 output channels), or 2 for maps whose grid of 64-channel tiles leaves most CUs idle
// (batch-1 56 x 56: 112 workgroups) -- twice the workgroups, each with half the W stream, at twice the LDS reads per MFMA.
template <typename TIN, int NF = 4, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_wide_kernel(const ConvArgs a) {
  constexpr int BN = 16 * NF, SLAB_LD = BN + 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
  const int Cin = a.Cin;
  int t, nblk;
  if (!conv_tile_of(a, t, nblk)) return;
  const int tx = t % a.tiles_x; t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = nblk * BN;
  const int nchunk = Cin / CH, upt = Cin >> 5;                     // chunks; 32-deep units per tap in the weight's K order
  const int total = 9 * nchunk;                                    // units of this wave: (chunk, tap), tap fastest

  // ---- W stream: unit (chunk c, tap) of wave w is the global 32-deep unit tap * upt + 4 c + w.  Buffer loads: the lane part of the
  // address is one register (lane * 16 bytes + the wave's 32-deep half), unit and 16-row panel go into the scalar offset
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  const int nkb = (9 * Cin) >> 6;
  const uint32_t panel_bytes = (uint32_t)nkb * 128 * 8 * 2;        // one 16-row fragment panel of the packed weight
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.w)) + (int64_t)(n0 >> 4) * panel_bytes, 0, (int)(NF * panel_bytes), 0x00020000);
  const uint32_t wlane = (uint32_t)lane * 16 + (uint32_t)(wave & 1) * 1024;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);         // (uniform by construction; lets the unit offset live in SGPRs)
  auto w_off = [&](int seq) -> uint32_t {                          // byte offset of a unit inside a panel
    seq = seq < total ? seq : total - 1;                           // the tail re-loads the last unit instead of branching
    const int c = seq / 9, tap = seq - 9 * c;
    return (uint32_t)((tap * upt + 4 * c + wave_u) >> 1) * 2048;
  };
  auto w_load = [&](int seq, bf16x8* dst) {
    const uint32_t o = w_off(seq);
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, o + n * panel_bytes, 0);
      dst[n] = *reinterpret_cast<const bf16x8*>(&v);
    }
  };
  bf16x8 wq[DEPTH][NF];
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) w_load(s, wq[s]);

  f32x4 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // lane (g, r) of row block m is output pixel (m, r); its halo pixel for tap (ky, kx) is (m + ky, r + kx); the wave's channel
  // unit inside the chunk: 64-block wave >> 1, lane group g -> + 16 g, half wave & 1 -> + 8 (the W lane's k = 16 g + 8 h + j)
  const char* abase = lds + r * PS + (((wave >> 1) << 6) + 16 * g + ((wave & 1) << 3)) * 2;

  constexpr int CHK = 16 / (int)sizeof(TIN);                       // elements per 16-byte chunk
  constexpr int CPP = CH / CHK;                                    // 16-byte chunks per halo pixel: 16 (bf16) or 32 (fp32)
  // loads in flight per thread: bf16 maps -- the whole chunk (12 x 16 bytes, one memory latency per chunk; the A fragments are not
  // live yet, so accumulators + W queue + these fit the 256 registers); fp32 maps -- 6 of its 23
  constexpr int PSTEP = 256 / CPP, NB = sizeof(TIN) == 4 ? 6 : 12;
  // buffer loads off a per-image descriptor: one 32-bit offset register per load in flight, and halo pixels outside the image get
  // an out-of-range offset, for which the hardware returns zeros (an image is < 2 GiB: checked by the host)
  const uint32_t img_bytes = (uint32_t)(a.H * a.W * Cin) * (uint32_t)sizeof(TIN);
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.x)) + (int64_t)b * img_bytes, 0, (int)img_bytes, 0x00020000);
  const uint32_t xlane = (uint32_t)((tid % CPP) * CHK * sizeof(TIN));
  char* hdst = lds + (tid % CPP) * CHK * 2;

  // halo chunk -> LDS in two steps: `issue` requests NB pixels' 16-byte pieces of chunk c from memory, `commit` applies the input
  // ReLU / the bf16 rounding and writes them to LDS (out-of-image pixels arrive as zeros).  bf16 maps: one batch holds the whole
  // chunk, i.e. one memory latency per chunk.  (Requesting the NEXT chunk's batch behind a wave's last tap, in flight across the
  // barrier, was tried: with 128 accumulator + 48 W-queue registers the 48 buffer registers do not fit next to the loop-invariant
  // offsets, the compiler spilled 87 registers and serialised the loads.)
  u32x4 buf[NB];
  auto issue = [&](int c, int p0) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int p = p0 + q * PSTEP;
      const int hy = p / HW, hx = p - hy * HW;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      const bool ok = p < HP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const uint32_t off = ok ? (uint32_t)((iy * a.W + ix) * Cin + c * CH) * (uint32_t)sizeof(TIN) + xlane : 0x80000000u;
      if constexpr (ABL & 4) buf[q] = u32x4{0u, 0u, 0u, 0u};
      else buf[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    }
  };
  auto commit = [&](int p0) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int p = p0 + q * PSTEP;
      if (p >= HP) continue;
      char* dst = hdst + p * PS;
      if constexpr (sizeof(TIN) == 4) {
        float4 v = make_float4(__uint_as_float(buf[q][0]), __uint_as_float(buf[q][1]), __uint_as_float(buf[q][2]), __uint_as_float(buf[q][3]));
        if (a.relu_in) v = relu4(v);
        bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(dst) = o;
      } else {
        u32x4 v = buf[q];
        if (a.relu_in) {
          // two bf16 per dword: a negative half (sign bit set) becomes +0
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t neg = v[e] & 0x80008000u;                 // sign bits of both halves
            v[e] &= ~((neg >> 15) * 0xffffu);                        // 0x0001 / 0x00010000 per negative half -> 16-bit masks
          }
        }
        *reinterpret_cast<u32x4*>(dst) = v;
      }
    }
  };
  for (int c = 0; c < nchunk; ++c) {
    if (c) __syncthreads();                                        // every wave is done with the previous chunk's halo bytes
    for (int p0 = tid / CPP; p0 < HP; p0 += NB * PSTEP) {          // (bf16 maps: one trip)
      issue(c, p0);
      commit(p0);
    }
    __syncthreads();

    // ---- the 9 taps of this chunk
    bf16x8 af[MF];
#pragma unroll
    for (int m = 0; m < MF; ++m) af[m] = ld_frag_lds(abase + m * HW * PS);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = tap % DEPTH;
      const int nt = tap + 1, nky = nt / 3, nkx = nt - 3 * nky;
#pragma unroll
      for (int m = 0; m < MF; ++m) {
#pragma unroll
        for (int n = 0; n < NF; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], wq[s][n], acc[m][n], 0, 0, 0);
        if constexpr (!(ABL & 2))
          if (tap < 8) af[m] = ld_frag_lds(abase + ((m + nky) * HW + nkx) * PS);
      }
      // refill this queue slot with the unit DEPTH ahead (9 % DEPTH == 0: the slot of a tap is the same in every chunk)
      if constexpr (!(ABL & 1)) w_load(9 * c + tap + DEPTH, wq[s]);
    }
  }

  // ---- partial tiles -> LDS in two passes of 64 pixels (C layout: col = lane & 15, row = 4 g + reg), coalesced epilogue
  float* slab = reinterpret_cast<float*>(lds) + (size_t)wave * 64 * SLAB_LD;
  const float* sm = reinterpret_cast<const float*>(lds);
  if constexpr (ABL & 8) {
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (sum == 12345.678f) reinterpret_cast<float*>(a.out)[tid] = sum;
    return;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();                                               // halo bytes (h = 0) / the previous pass's sums are consumed
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) slab[(m * 16 + 4 * g + q) * SLAB_LD + n * 16 + r] = acc[4 * h + m][n][q];
    __syncthreads();
    for (int idx = tid; idx < 64 * (BN / 4); idx += 256) {
      const int row = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
      const int oy = y0 + 4 * h + (row >> 4), ox = x0 + (row & 15);
      if (oy >= a.H || ox >= a.W) continue;
      float4 v = *reinterpret_cast<const float4*>(sm + row * SLAB_LD + c4);
#pragma unroll
      for (int s = 1; s < 4; ++s) {
        const float4 t4 = *reinterpret_cast<const float4*>(sm + (size_t)s * 64 * SLAB_LD + row * SLAB_LD + c4);
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
        st_out(reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.out) + off), o);
      } else {
        st_out(reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + off), v);
      }
    }
  }
}
}  // namespace wide

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
  a.relu_in = relu_in; a.act = act; a.out_bf16 = out_bf16 & 3;
  a.xcd_nb = 0;
  constexpr bool xcd_on = true;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    // tile choice (out_bf16 bits 2-3: 0 = by size, 1 = 8 x 8 pixels, 2 = 8 x 16 pixels x 64 channels, 3 = 8 x 16 pixels x 32 channels): the
    // wide tile halves the W stream per flop and wins once its own grid covers most of the 256 CUs (profiles/r04_conv_tile_sweep.txt);
    // when the 64-channel grid is too small for that but the 32-channel one is not, the narrow-N instance takes over
    const int want = (out_bf16 >> 2) & 3;
    const int wtx = (W + wide::TW - 1) / wide::TW, wty = (H + wide::TH - 1) / wide::TH;
    const int64_t wgs = (int64_t)wtx * wty * B * (Cout / 64);
    const bool can = Cin % wide::CH == 0;
    SP3_CHECK(want < 2 || can, "sp3_conv3x3_tile: the 8 x 16 tile needs Cin %% 128 == 0 (Cin=%d)", Cin);
    SP3_CHECK((int64_t)H * W * Cin * (in_bf16 ? 2 : 4) < (1ll << 31), "sp3_conv3x3_tile: one image of the map must stay below 2 GiB");
    const int nf = want == 2 ? 4 : want == 3 ? 2 : (want == 0 && can) ? (wgs >= SP3_CONV_WIDE_MIN_WGS ? 4 : wgs >= 80 ? 2 : 0) : 0;
    if (nf) {
      a.tiles_x = wtx; a.tiles_y = wty;
      dim3 grid(wtx * wty * B, Cout / (16 * nf));
      if (xcd_on) { a.xcd_nb = Cout / (16 * nf); grid = dim3((unsigned)(((wtx * wty * B + 7) / 8) * 8 * a.xcd_nb), 1); }
      auto launch = [&](auto kern, int lds_bytes) -> int {
        static bool raised = false;               // per instantiation: opt in to > 64 KiB of dynamic LDS once
        if (!raised && lds_bytes > 64 * 1024) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
          if (e != hipSuccess) { sp3_set_error("sp3_conv3x3_tile: cannot raise dynamic LDS to %d: %s", lds_bytes, hipGetErrorString(e)); return 2; }
          raised = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
        return 0;
      };
      int rc;
      if (nf == 4) rc = in_bf16 ? launch(wide::conv3x3_wide_kernel<__bf16, 4>, wide::lds_bytes<4>()) : launch(wide::conv3x3_wide_kernel<float, 4>, wide::lds_bytes<4>());
      else         rc = in_bf16 ? launch(wide::conv3x3_wide_kernel<__bf16, 2>, wide::lds_bytes<2>()) : launch(wide::conv3x3_wide_kernel<float, 2>, wide::lds_bytes<2>());
      if (rc) return rc;
      SP3_LAUNCH_CHECK("sp3_conv3x3_tile");
      return 0;
    }
  }
  a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
  const size_t halo = (size_t)HP * (Cin * 2 + 16), slabs = (size_t)4 * BM * SLAB_LD * sizeof(float);
  const size_t lds = halo > slabs ? halo : slabs;
  SP3_CHECK(lds <= 160 * 1024, "sp3_conv3x3_tile: Cin=%d needs %zu bytes of LDS", Cin, lds);
  dim3 grid(a.tiles_x * a.tiles_y * B, Cout / BN);
  if (xcd_on) { a.xcd_nb = Cout / BN; grid = dim3((unsigned)(((a.tiles_x * a.tiles_y * B + 7) / 8) * 8 * a.xcd_nb), 1); }
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
