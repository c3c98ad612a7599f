// Structure probe for the MANY-ROW GEMMs of the whole-sequence encoder (croco/models/blocks.py:73-79,94-112 at M = frames x 196):
// 1960 x 4096 x 1024 (fc1), 1960 x 3072 x 1024 (q/k/v), 1960 x 1024 x 4096 (fc2) on bf16 fragment-order operands, HBM-cold weights
// (every launch of the timed hipGraph reads its own copy of W, as every encoder layer does).  Main-loop structures side by side,
// each with shader-clock stamps per workgroup (entry / first stage landed / K loop done / end), so that a single run says where
// a launch's time goes:
//   k16  the product's bm_kernel loop: 8 waves x (64 x 64), v_mfma_f32_16x16x32_bf16, LDS ring filled by global_load_lds
//   k32  v_mfma_f32_32x32x16_bf16 on the SAME fragment-order bytes (a 32-row operand = two 16-row blocks, lanes 16..31 / 48..63
//        read the second block: still one conflict-free ds_read_b128 per operand), wave tile (32 TM) x (32 TN):
//          8 waves x (64 x 64) | 4 waves x (128 x 64) self-loading | 4 compute waves x (128 x 64) + 4 loader waves
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/gemm_bm.hip -o tools/ubench/gemm_bm.bin && tools/ubench/gemm_bm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include <type_traits>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __host__ inline long packed_off(int row, int k, int K) {
  const int nkb = K >> 6, kb = k >> 6, kk = k & 63, g = kk >> 4, e = kk & 15, h = e >> 3, eh = e & 7;
  return (((((long)(row >> 4) * nkb + kb) * 2 + h) * 4 + g) * 16 + (row & 15)) * 8 + eh;
}

__device__ __forceinline__ float erf_fast(float z) {
  const float a = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float y = fmaf(-p * t, __expf(-a * a), 1.0f);
  return copysignf(y, z);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

struct BArgs {
  const char* A; const char* W; const float* bias; char* C;
  int M, N, rb_max;
  unsigned long long* trace;      // [wg][8]: 4 shader-clock stamps, 2 wall-clock stamps
};

__device__ __forceinline__ void glds16(const char* gsrc, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#define STAMP(i) do { if (a.trace && threadIdx.x == 0) a.trace[(size_t)wg_id * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#define WSTAMP(i) do { if (a.trace && threadIdx.x == 0) a.trace[(size_t)wg_id * 8 + (i)] = wall_clock64(); } while (0)

// ------------------------------------------------------------------------------------------------ k16: the product loop
template <int WM, int WN, int NF, int NKB, int NST, int EPI>
__global__ __launch_bounds__(64 * WM * WN) void k16(const BArgs a) {
  constexpr int MF = 4, NW = WM * WN;
  constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
  constexpr int NBLK = BM / 16 + BN / 16, STAGE_BYTES = NBLK * 2048, NINSTR = 2 * NBLK, PER = (NINSTR + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) char lds_b[];
  const int tid = threadIdx.x, lane = tid & 63, wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg_id = blockIdx.x + 8 * (blockIdx.y + gridDim.y * blockIdx.z);
  WSTAMP(4);
  STAMP(0);
  const int wn = wave_u % WN, wm = wave_u / WN;
  const int g = lane >> 4, r16 = lane & 15;
  const int tile_m = blockIdx.y, tile_n = blockIdx.z * 8 + blockIdx.x;
  const int N = a.N;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (n0 >= N) return;
  const char* src[PER];
  int dst[PER];
  {
    const int nb_max = (N >> 4) - 1;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int j = wave_u + i * NW;
      j = j < NINSTR ? j : NINSTR - 1;
      const int blk = j >> 1;
      const char* base;
      if (blk < BM / 16) {
        int rb = (m0 >> 4) + blk;
        rb = rb < a.rb_max ? rb : a.rb_max;
        base = a.A + (long)rb * NKB * 2048;
      } else {
        int nb = (n0 >> 4) + blk - BM / 16;
        nb = nb < nb_max ? nb : nb_max;
        base = a.W + (long)nb * NKB * 2048;
      }
      src[i] = base + (j & 1) * 1024 + lane * 16;
      dst[i] = j * 1024;
    }
  }
  auto issue = [&](int slot, int kb) {
#pragma unroll
    for (int i = 0; i < PER; ++i) glds16(src[i] + (long)kb * 2048, lds_b + slot * STAGE_BYTES + dst[i]);
  };
  auto wait_pending = [&](int pend) {
    if (pend >= 3) wait_vm<3 * PER>();
    else if (pend == 2) wait_vm<2 * PER>();
    else if (pend == 1) wait_vm<PER>();
    else wait_vm<0>();
  };
  int grow[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) grow[m] = m0 + wm * 64 + m * 16 + r16;
  const int cw0 = n0 + wn * NF * 16 + 4 * g;
  float4 pb4[NF];
#pragma unroll
  for (int n = 0; n < NF; ++n) pb4[n] = *reinterpret_cast<const float4*>(a.bias + cw0 + n * 16);
#pragma unroll
  for (int s_ = 0; s_ < NST; ++s_) issue(s_, s_);

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
  STAMP(1);
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
  if (a.trace) {   // make the stamp wait for the accumulators
    float t = 0.f;
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) t += acc[m][n][0];
    if (t == 1.2345e33f) a.C[1] = 1;
  }
  STAMP(2);
  if constexpr (EPI == 0) {
    float t = 0.f;
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) t += (acc[m][n][0] + acc[m][n][1]) + (acc[m][n][2] + acc[m][n][3]);
    if (t == 1.2345e33f) a.C[0] = 1;
  } else {
    __bf16* out = reinterpret_cast<__bf16*>(a.C);
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int gm = grow[m];
      if (gm >= a.M) continue;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        float v[4] = {acc[m][n][0] + pb4[n].x, acc[m][n][1] + pb4[n].y, acc[m][n][2] + pb4[n].z, acc[m][n][3] + pb4[n].w};
        if constexpr (EPI == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        bf16x4 ob;
        ob[0] = (__bf16)v[0]; ob[1] = (__bf16)v[1]; ob[2] = (__bf16)v[2]; ob[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(out + packed_off(gm, cw0 + n * 16, N)) = ob;
      }
    }
  }
  STAMP(3);
  WSTAMP(5);
}

// ------------------------------------------------------------------------------------------------ k32: 32x32x16 fragments
// compute waves WM x WN, each (32 TM) x (32 TN); LOADERS: the same number of extra waves that do nothing but the stage DMA.
// HS: the ring holds HALF k-blocks (piece h of every 16-row block: 1 KB per block, the k's {16 g + 8 h + e}); a 32x32x16 operand
//     of step t takes the lane groups g = 2 t + (lane >> 5) of that piece.  Twice the stages in the same LDS: more bytes in flight.
// KS: k16-steps per register set ("unit"): the wait for a set sits in front of its first MFMA, the next set's reads behind it.
// MODE 1: loads only (DMA ring + barriers, no LDS reads, no MFMAs): the fill floor of the ring.  MODE 2: no DMA inside the loop
// (LDS reads + MFMAs + barriers on whatever the prologue staged).  MODE 3: DMA + MFMAs + barriers, no LDS reads in the loop.
// ISS (self-loading): 0 = every wave issues its refill pieces right behind the barrier; 1 = waves of the first half there, the
//     second half (the SIMD partners) in the middle of the k-block; 2 = first half at the barrier, second half one unit later.
//     LOADERS: ISS = 1 raises the loader waves' priority (s_setprio 3).
// EPI: 0 none, 1 bias, 2 bias + GELU (A&S 7.1.26 erf, the product's), 3 bias + GELU (A&S 26.2.17 normal tail, 12 VALU ops),
//      4 = 3 with write-through (sc1) stores
__device__ __forceinline__ float gelu_tail(float x) {
  // x Phi(x) = max(x, 0) - |x| Q(|x|), Q(a) = phi(a) (b1 t + .. + b5 t^5), t = 1 / (1 + 0.2316419 a)   (|error of Q| < 7.5e-8)
  const float a = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.2316419f, a, 1.0f));
  float p = fmaf(1.330274429f, t, -1.821255978f);
  p = fmaf(p, t, 1.781477937f);
  p = fmaf(p, t, -0.356563782f);
  p = fmaf(p, t, 0.319381530f);
  // phi(a) = exp(-a^2 / 2) / sqrt(2 pi): exp2(-0.72134752 a^2) * 0.39894228
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * a * a);
  const float q = (p * t) * (e * 0.3989422804014327f);
  return fmaf(-a, q, fmaxf(x, 0.f));
}

template <int WM, int WN, int TM, int TN, int KS, bool HS, int NKB, int NST, int EPI, bool LOADERS, int MODE, bool READS_FIRST = false, int ISS = 0>
__global__ __launch_bounds__(64 * WM * WN * (LOADERS ? 2 : 1)) void k32(const BArgs a) {
  constexpr int NC = WM * WN;                     // compute waves
  constexpr int NL = NC;                          // waves that issue DMA (the loader waves, or the compute waves themselves)
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int NBLK = BM / 16 + BN / 16;
  constexpr int PIECES = HS ? NBLK : 2 * NBLK;    // 1 KB pieces per stage
  constexpr int STAGE_BYTES = PIECES * 1024, PER = (PIECES + NL - 1) / NL;
  constexpr int BLK = HS ? 1024 : 2048;           // bytes of one 16-row block inside a stage
  constexpr int SPS = HS ? 2 : 4;                 // k16-steps per stage
  constexpr int NSTAGE = NKB * 4 / SPS;           // stages of the whole K
  constexpr int U = SPS / KS;                     // units per stage
  constexpr int TU = NSTAGE * U;
  static_assert(PER * (NST - 1) < 64, "vmcnt is a 6-bit count");
  static_assert(U >= 1 && TU % 2 == 0, "units");
  extern __shared__ __attribute__((aligned(16))) char lds_b[];
  const int tid = threadIdx.x, lane = tid & 63, wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg_id = blockIdx.x + 8 * (blockIdx.y + gridDim.y * blockIdx.z);
  WSTAMP(4);
  STAMP(0);
  const bool is_loader = LOADERS && wave_u >= NC;
  const int wl = LOADERS ? (wave_u >= NC ? wave_u - NC : 0) : wave_u;       // DMA lane of this wave
  const int wc = wave_u % NC;
  const int wn = wc % WN, wm = wc / WN;
  const int tile_m = blockIdx.y, tile_n = blockIdx.z * 8 + blockIdx.x;
  const int N = a.N;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (n0 >= N) return;
  const char* src[PER];
  int dst[PER];
  {
    const int nb_max = (N >> 4) - 1;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int j = wl + i * NL;
      j = j < PIECES ? j : PIECES - 1;
      const int blk = HS ? j : j >> 1;
      const char* base;
      if (blk < BM / 16) {
        int rb = (m0 >> 4) + blk;
        rb = rb < a.rb_max ? rb : a.rb_max;
        base = a.A + (long)rb * NKB * 2048;
      } else {
        int nb = (n0 >> 4) + blk - BM / 16;
        nb = nb < nb_max ? nb : nb_max;
        base = a.W + (long)nb * NKB * 2048;
      }
      src[i] = base + (HS ? 0 : (j & 1) * 1024) + lane * 16;
      dst[i] = j * 1024;
    }
  }
  // stage s: HS: k-block s >> 1, piece s & 1 (+ 1024 bytes); else k-block s
  auto issue = [&](int slot, int s) {
    const long off = HS ? (long)(s >> 1) * 2048 + (s & 1) * 1024 : (long)s * 2048;
#pragma unroll
    for (int i = 0; i < PER; ++i) glds16(src[i] + off, lds_b + slot * STAGE_BYTES + dst[i]);
  };
  auto wait_pending = [&](int pend) {
#define WP(k) case k: wait_vm<((k) * PER < 64 ? (k) * PER : 63)>(); break;
    switch (pend) { WP(0) WP(1) WP(2) WP(3) WP(4) WP(5) WP(6) WP(7) WP(8) WP(9) WP(10) default: wait_vm<(11 * PER < 64 ? 11 * PER : 63)>(); break; }
#undef WP
  };

  if (LOADERS && is_loader) {
    // ---- loader waves: keep NST - 1 stages in flight; one barrier per stage with the compute waves
    if (ISS == 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int s_ = 0; s_ < NST; ++s_) issue(s_, s_);
    wait_pending(NST - 1);
    asm volatile("s_barrier" ::: "memory");
    for (int s = 1; s < NSTAGE; ++s) {
      const int newer = NSTAGE - 1 - s;
      wait_pending(newer < NST - 2 ? newer : NST - 2);
      asm volatile("s_barrier" ::: "memory");
      if (s + NST - 1 < NSTAGE) issue((s + NST - 1) % NST, s + NST - 1);
    }
    return;
  }

  // ---- compute waves
  const int ml = lane & 31, hi = lane >> 5;
  const int wave_m0 = m0 + wm * TM * 32, wave_n0 = n0 + wn * TN * 32;
  if (!LOADERS) {
#pragma unroll
    for (int s_ = 0; s_ < NST; ++s_) issue(s_, s_);
  }
  float4 pb4[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int q = 0; q < 4; ++q) pb4[tn][q] = *reinterpret_cast<const float4*>(a.bias + wave_n0 + tn * 32 + 8 * q + 4 * hi);
  typedef bf16x8 V16;
  V16 fa[2][KS][TM], fw[2][KS][TN];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  // lanes 0..15 rows of block b, lanes 16..31 rows of block b + 1; lanes >= 32: the other 8 k of the step
  const int lane_off = ((lane >> 4) & 1) * BLK + (HS ? hi * 256 : hi * 1024) + (lane & 15) * 16;
  auto read_unit = [&](auto buf_tag, int slot, int u) {
    constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const char* st = lds_b + slot * STAGE_BYTES + (u * KS + ks) * (HS ? 512 : 256) + lane_off;
#pragma unroll
      for (int m = 0; m < TM; ++m) fa[BUF][ks][m] = *reinterpret_cast<const V16*>(st + (wm * TM + m) * 2 * BLK);
#pragma unroll
      for (int n = 0; n < TN; ++n) fw[BUF][ks][n] = *reinterpret_cast<const V16*>(st + (BM / 16 + (wn * TN + n) * 2) * BLK);
    }
  };
  auto mma_first = [&](auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[BUF][0][0], fa[BUF][0][0], acc[0][0], 0, 0, 0);
  };
  auto mma_rest = [&](auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
          if (ks + m + n > 0) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[BUF][ks][n], fa[BUF][ks][m], acc[m][n], 0, 0, 0);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  if (!LOADERS) wait_pending(NST - 1);
  asm volatile("s_barrier" ::: "memory");
  STAMP(1);
  int slot = 0;
  int late_slot = -1, late_stage = 0;             // ISS: the refill the second half of the waves still owes
  const bool late_wave = ISS != 0 && !LOADERS && wave_u >= NC / 2;
  if (MODE != 1) read_unit(B0{}, 0, 0);
  if (MODE == 3) read_unit(B1{}, 0, 0);
  auto unit = [&](auto cur_tag, auto nxt_tag, int ui) {
    const int st = ui / U, u = ui % U;
    if (READS_FIRST && u < U - 1) {
      if (MODE == 0 || MODE == 2) read_unit(nxt_tag, slot, u + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE != 1) mma_first(cur_tag);
    __builtin_amdgcn_sched_barrier(0);
    if (late_wave && late_slot >= 0 && u == (ISS == 1 ? U / 2 - 1 : 0)) {
      issue(late_slot, late_stage);
      late_slot = -1;
    }
    if (READS_FIRST && u < U - 1) {
    } else if (u < U - 1) {
      if (MODE == 0 || MODE == 2) read_unit(nxt_tag, slot, u + 1);
    } else {
      // last unit of the stage: publish the next stage, refill the slot every wave has finished reading
      const int cur = slot;
      slot = slot + 1 == NST ? 0 : slot + 1;
      if (st + 1 < NSTAGE) {
        const int s = st + 1;
        if (!LOADERS && MODE != 2) {
          const int newer = NSTAGE - 1 - s;
          wait_pending(newer < NST - 2 ? newer : NST - 2);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!LOADERS && MODE != 2 && s + NST - 1 < NSTAGE) {
          if (late_wave) { late_slot = cur; late_stage = s + NST - 1; }
          else issue(cur, s + NST - 1);
        }
        if (MODE == 0 || MODE == 2) read_unit(nxt_tag, slot, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 1) mma_rest(cur_tag);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int ui = 0; ui < TU; ui += 2) {
    unit(B0{}, B1{}, ui);
    unit(B1{}, B0{}, ui + 1);
  }
  if (a.trace) {
    float t = 0.f;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < TN; ++n) t += acc[m][n][0];
    if (t == 1.2345e33f) a.C[1] = 1;
  }
  STAMP(2);
  if constexpr (EPI == 0) {
    float t = 0.f;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[m][n][r];
    if (t == 1.2345e33f) a.C[0] = 1;
  } else {
    // D = W_frag . A_frag^T: lane (ml, hi) holds output row ml of row tile tm, columns 8 q + 4 hi + (0..3) of column tile tn (q = reg >> 2)
    __bf16* out = reinterpret_cast<__bf16*>(a.C);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int gm = wave_m0 + tm * 32 + ml;
      if (gm >= a.M) continue;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = pb4[tn][q];
          float v[4] = {acc[tm][tn][4 * q] + b4.x, acc[tm][tn][4 * q + 1] + b4.y, acc[tm][tn][4 * q + 2] + b4.z, acc[tm][tn][4 * q + 3] + b4.w};
          if constexpr (EPI == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
          }
          if constexpr (EPI >= 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_tail(v[e]);
          }
          bf16x4 ob;
          ob[0] = (__bf16)v[0]; ob[1] = (__bf16)v[1]; ob[2] = (__bf16)v[2]; ob[3] = (__bf16)v[3];
          __bf16* dstp = out + packed_off(gm, wave_n0 + tn * 32 + 8 * q + 4 * hi, N);
          if constexpr (EPI == 4)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(dstp), __builtin_bit_cast(unsigned long long, ob), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            *reinterpret_cast<bf16x4*>(dstp) = ob;
        }
    }
  }
  STAMP(3);
  WSTAMP(5);
}

// ------------------------------------------------------------------------------------------------ host
static float bf2f(__bf16 x) { return (float)x; }

struct Problem {
  int M, N, K, ncopy, Mp;
  char *A, *W, *C;
  float* bias;
  unsigned long long* trace;
  std::vector<__bf16> hA, hW0;
  std::vector<float> hb;
};

static Problem make(int M, int N, int K, size_t mb) {
  Problem p;
  p.M = M; p.N = N; p.K = K;
  const int Mp = (M + 15) / 16 * 16;
  p.Mp = Mp;
  const size_t wbytes = (size_t)N * K * 2;
  p.ncopy = (int)std::max<size_t>(1, std::min<size_t>(64, mb * (1 << 20) / wbytes));
  p.hA.assign((size_t)Mp * K, (__bf16)0.f);
  p.hW0.resize((size_t)N * K);
  p.hb.resize(N);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  std::vector<__bf16> pa((size_t)Mp * K), pw((size_t)N * K);
  for (int r = 0; r < Mp; ++r)
    for (int k = 0; k < K; ++k) { __bf16 v = (__bf16)(r < M ? rnd() : 0.f); p.hA[(size_t)r * K + k] = v; pa[packed_off(r, k, K)] = v; }
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) { __bf16 v = (__bf16)(rnd() * 0.1f); p.hW0[(size_t)n * K + k] = v; pw[packed_off(n, k, K)] = v; }
  for (int n = 0; n < N; ++n) p.hb[n] = rnd();
  CK(hipMalloc(&p.A, pa.size() * 2));
  CK(hipMemcpy(p.A, pa.data(), pa.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&p.W, wbytes * p.ncopy));
  for (int c = 0; c < p.ncopy; ++c) CK(hipMemcpy(p.W + c * wbytes, pw.data(), wbytes, hipMemcpyHostToDevice));
  CK(hipMalloc(&p.bias, N * 4));
  CK(hipMemcpy(p.bias, p.hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&p.C, (size_t)Mp * N * 2));
  CK(hipMalloc(&p.trace, 4096 * 8 * 8));
  return p;
}
static void release(Problem& p) { for (void* q : {(void*)p.A, (void*)p.W, (void*)p.C, (void*)p.bias, (void*)p.trace}) CK(hipFree(q)); }

template <typename F>
static float time_graph(int nlaunch, F&& launch_i) {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < nlaunch; ++i) launch_i(i, st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
  return best * 1e3f / nlaunch;
}

template <typename K>
static void run_kernel(Problem& p, const char* tag, K kern, int BM, int BN, int threads, size_t lds, int epi) {
  if (p.N % BN != 0) return;
  const int mt = (p.M + BM - 1) / BM, nt = p.N / BN, nz = (nt + 7) / 8;
  if (lds > 160 * 1024) { printf("  %-44s LDS %zu too large\n", tag, lds); return; }
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)kern));
  const size_t wbytes = (size_t)p.N * p.K * 2;
  BArgs a{p.A, p.W, p.bias, p.C, p.M, p.N, (p.M + 15) / 16 - 1, nullptr};
  const dim3 grid(8, mt, nz);
  const int nwg = 8 * mt * nz;
  // correctness (bias / bias + gelu epilogues)
  double worst = -1;
  if (epi > 0) {
    CK(hipMemset(p.C, 0, (size_t)p.Mp * p.N * 2));
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<__bf16> hc((size_t)p.Mp * p.N);
    CK(hipMemcpy(hc.data(), p.C, hc.size() * 2, hipMemcpyDeviceToHost));
    worst = 0;
    unsigned s = 777u;
    for (int t = 0; t < 600; ++t) {
      s = s * 1664525u + 1013904223u;
      const int r = (t < 8) ? p.M - 1 - t : (t < 16 ? t - 8 : (int)((s >> 8) % p.M));
      s = s * 1664525u + 1013904223u;
      const int c = (int)((s >> 8) % p.N);
      double ref = p.hb[c];
      for (int k = 0; k < p.K; ++k) ref += (double)bf2f(p.hA[(size_t)r * p.K + k]) * (double)bf2f(p.hW0[(size_t)c * p.K + k]);
      if (epi >= 2) ref = 0.5 * ref * (1.0 + std::erf(ref * 0.70710678118654752440));
      const double got = bf2f(hc[packed_off(r, c, p.N)]);
      worst = std::max(worst, std::fabs(ref - got) / (std::fabs(ref) + 0.05));
    }
  }
  const int nl = std::max(40, p.ncopy);
  const float us = time_graph(nl, [&](int i, hipStream_t st) {
    BArgs b = a;
    b.W = p.W + (size_t)(i % p.ncopy) * wbytes;
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, st, b);
  });
  // traced launch (cold W copy), stamps of every workgroup
  CK(hipMemset(p.trace, 0, 4096 * 8 * 8));
  BArgs t = a;
  t.trace = p.trace;
  t.W = p.W + (size_t)(p.ncopy - 1) * wbytes;
  hipLaunchKernelGGL(kern, grid, dim3(threads), lds, 0, t);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> ht((size_t)nwg * 8);
  CK(hipMemcpy(ht.data(), p.trace, ht.size() * 8, hipMemcpyDeviceToHost));
  double s01 = 0, s12 = 0, s23 = 0, mx = 0;
  unsigned long long w0 = ~0ull, w1 = 0, wlast_start = 0;
  int cnt = 0;
  for (int w = 0; w < nwg; ++w) {
    const unsigned long long* q = &ht[(size_t)w * 8];
    if (!q[3]) continue;
    ++cnt;
    s01 += (double)(q[1] - q[0]); s12 += (double)(q[2] - q[1]); s23 += (double)(q[3] - q[2]);
    mx = std::max(mx, (double)(q[3] - q[0]));
    w0 = std::min(w0, q[4]); w1 = std::max(w1, q[5]); wlast_start = std::max(wlast_start, q[4]);
  }
  const double flop = 2.0 * p.M * p.N * p.K;
  printf("  %-44s %4d wgs x %4d thr  vgpr %3d lds %6zu  %7.2f us %6.0f TF | clk: prologue %6.0f  loop %6.0f (%5.0f / k-block)  epilogue %6.0f  max life %6.0f | wall %5.2f us, starts within %4.2f us | err %.1e\n",
         tag, nwg, threads, fa.numRegs, lds, us, flop / us * 1e-6, s01 / cnt, s12 / cnt, s12 / cnt / (p.K / 64), s23 / cnt, mx,
         (double)(w1 - w0) * 0.01, (double)(wlast_start - w0) * 0.01, worst);
  fflush(stdout);
}

template <int WM, int WN, int NF, int NKB, int NST, int EPI>
static void run16(Problem& p, const char* tag) {
  if (p.K != NKB * 64) return;
  constexpr int BM = WM * 64, BN = WN * NF * 16;
  run_kernel(p, tag, k16<WM, WN, NF, NKB, NST, EPI>, BM, BN, 64 * WM * WN, (size_t)NST * (BM / 16 + BN / 16) * 2048, EPI);
}
template <int WM, int WN, int TM, int TN, int KS, bool HS, int NKB, int NST, int EPI, bool LOADERS, int MODE = 0, bool RF = false, int ISS = 0>
static void run32(Problem& p, const char* tag) {
  if (p.K != NKB * 64) return;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  run_kernel(p, tag, k32<WM, WN, TM, TN, KS, HS, NKB, NST, EPI, LOADERS, MODE, RF, ISS>, BM, BN, 64 * WM * WN * (LOADERS ? 2 : 1),
             (size_t)NST * (BM / 16 + BN / 16) * (HS ? 1024 : 2048), MODE ? 0 : EPI);
}

template <int NKB>
static void suite(Problem& p) {
  run16<4, 2, 4, NKB, 3, 0>(p, "k16 256x128 8w(64x64) 3x48K epi none");
  run32<4, 2, 2, 2, 2, false, NKB, 3, 0, false, 0>(p, "k32 8w ks2 3x48K full");
  run32<4, 2, 2, 2, 2, false, NKB, 3, 0, false, 0, false, 1>(p, "k32 8w ks2 3x48K issue split mid");
  run32<4, 2, 2, 2, 2, false, NKB, 3, 0, false, 0, false, 2>(p, "k32 8w ks2 3x48K issue split +1 unit");
  run32<4, 2, 2, 2, 1, false, NKB, 3, 0, false, 0, false, 1>(p, "k32 8w ks1 3x48K issue split mid");
  run32<4, 2, 2, 2, 1, false, NKB, 3, 0, false, 0, false, 2>(p, "k32 8w ks1 3x48K issue split +1 unit");
  run32<4, 2, 2, 2, 1, false, NKB, 3, 0, false, 3, false, 1>(p, "k32 8w ks1 3x48K issue split mid, no reads");
  run32<4, 2, 2, 2, 2, true, NKB, 6, 0, false, 0, false, 1>(p, "k32 8w ks2 6x24K issue split mid");
  run32<2, 2, 4, 2, 1, false, NKB, 3, 0, true, 0>(p, "k32 4w+4L ks1 3x48K full");
  run32<2, 2, 4, 2, 1, false, NKB, 3, 0, true, 0, false, 1>(p, "k32 4w+4L ks1 3x48K loaders prio 3");
  run32<2, 2, 4, 2, 1, false, NKB, 3, 0, true, 3, false, 1>(p, "k32 4w+4L ks1 3x48K loaders prio 3, no reads");
  run32<2, 2, 4, 2, 1, true, NKB, 6, 0, true, 0, false, 1>(p, "k32 4w+4L ks1 6x24K loaders prio 3");
  run32<2, 2, 4, 2, 1, false, NKB, 3, 2, true, 0, false, 1>(p, "k32 4w+4L ks1 3x48K loaders prio 3 epi gelu");
  run32<4, 2, 2, 2, 2, false, NKB, 3, 2, false, 0, false, 1>(p, "k32 8w ks2 3x48K issue split mid epi gelu");
}

int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : "";
  auto want = [&](const char* n) { return !only[0] || strstr(only, n); };
  if (want("fc1")) {
    Problem p = make(1960, 4096, 1024, 330);
    printf("enc fc1 1960x4096x1024 (%d weight copies)\n", p.ncopy);
    suite<16>(p);
    release(p);
  }
  if (want("qkv")) {
    Problem p = make(1960, 3072, 1024, 330);
    printf("enc qkv 1960x3072x1024 (%d weight copies)\n", p.ncopy);
    suite<16>(p);
    release(p);
  }
  if (want("fc2")) {
    Problem p = make(1960, 1024, 4096, 330);
    printf("enc fc2 1960x1024x4096 (%d weight copies)\n", p.ncopy);
    run16<2, 2, 2, 64, 3, 1>(p, "k16 128x64 4w(64x32) 3x24K epi bias");
    run32<2, 2, 2, 1, 1, false, 64, 3, 0, false, 1>(p, "LOADS ONLY 128x64 4w 3x24K");
    run32<2, 2, 2, 1, 1, true, 64, 6, 0, false, 1>(p, "LOADS ONLY 128x64 4w 6x12K");
    run32<2, 2, 2, 1, 1, true, 64, 12, 0, false, 1>(p, "LOADS ONLY 128x64 4w 12x12K");
    run32<2, 2, 2, 1, 2, false, 64, 3, 1, false>(p, "k32 128x64 4w(64x32) ks2 3x24K epi bias");
    run32<2, 2, 2, 1, 1, true, 64, 6, 1, false>(p, "k32 128x64 4w(64x32) ks1 6x12K epi bias");
    run32<2, 2, 2, 1, 1, true, 64, 12, 1, false>(p, "k32 128x64 4w(64x32) ks1 12x12K epi bias");
    run32<4, 2, 1, 1, 1, true, 64, 12, 1, false>(p, "k32 128x64 8w(32x32) ks1 12x12K epi bias");
    release(p);
  }
  if (want("c3")) {
    // config 3 (512 x 512 frame: M = 1024 rows per side, two sides grouped -> emulated as one problem with 2 N): what the ring depth /
    // workgroups per CU do to the 128-row tiles
    { Problem p = make(1024, 4608, 768, 330);
      printf("c3 dec qkv 1024x(2x2304)x768 (%d weight copies)\n", p.ncopy);
      run16<2, 2, 4, 12, 3, 1>(p, "k16 128x128 4w(64x64) 3x32K epi bias");
      run16<2, 2, 4, 12, 2, 1>(p, "k16 128x128 4w(64x64) 2x32K epi bias");
      run16<4, 2, 4, 12, 3, 1>(p, "k16 256x128 8w(64x64) 3x48K epi bias");
      run16<4, 2, 4, 12, 2, 1>(p, "k16 256x128 8w(64x64) 2x48K epi bias");
      run32<2, 2, 2, 2, 2, false, 12, 3, 1, false>(p, "k32 128x128 4w(64x64) ks2 3x32K epi bias");
      run32<2, 2, 2, 2, 2, false, 12, 2, 1, false>(p, "k32 128x128 4w(64x64) ks2 2x32K epi bias");
      run32<4, 2, 2, 2, 2, false, 12, 3, 1, false>(p, "k32 256x128 8w(64x64) ks2 3x48K epi bias");
      release(p); }
    { Problem p = make(1024, 6144, 768, 330);
      printf("c3 dec fc1 1024x(2x3072)x768 (%d weight copies)\n", p.ncopy);
      run16<2, 2, 4, 12, 3, 2>(p, "k16 128x128 4w(64x64) 3x32K epi gelu");
      run16<2, 2, 4, 12, 2, 2>(p, "k16 128x128 4w(64x64) 2x32K epi gelu");
      run16<4, 2, 4, 12, 3, 2>(p, "k16 256x128 8w(64x64) 3x48K epi gelu");
      run32<2, 2, 2, 2, 2, false, 12, 2, 3, false>(p, "k32 128x128 4w(64x64) ks2 2x32K epi gelu");
      run32<4, 2, 2, 2, 2, false, 12, 3, 3, false>(p, "k32 256x128 8w(64x64) ks2 3x48K epi gelu");
      release(p); }
    { Problem p = make(1024, 1536, 3072, 330);
      printf("c3 dec fc2 1024x(2x768)x3072 (%d weight copies)\n", p.ncopy);
      run16<2, 2, 2, 48, 3, 1>(p, "k16 128x64 4w(64x32) 3x24K epi bias");
      run16<2, 2, 2, 48, 4, 1>(p, "k16 128x64 4w(64x32) 4x24K epi bias");
      run16<2, 2, 2, 48, 2, 1>(p, "k16 128x64 4w(64x32) 2x24K epi bias");
      run16<2, 2, 4, 48, 2, 1>(p, "k16 128x128 4w(64x64) 2x32K epi bias");
      run32<2, 2, 2, 1, 2, false, 48, 3, 1, false>(p, "k32 128x64 4w(64x32) ks2 3x24K epi bias");
      release(p); }
    { Problem p = make(1024, 4096, 1024, 330);
      printf("c3 enc fc1 1024x4096x1024 (%d weight copies)\n", p.ncopy);
      run16<2, 2, 4, 16, 3, 2>(p, "k16 128x128 4w(64x64) 3x32K epi gelu");
      run16<2, 2, 4, 16, 2, 2>(p, "k16 128x128 4w(64x64) 2x32K epi gelu");
      run16<4, 2, 4, 16, 3, 2>(p, "k16 256x128 8w(64x64) 3x48K epi gelu");
      run32<2, 2, 2, 2, 2, false, 16, 2, 3, false>(p, "k32 128x128 4w(64x64) ks2 2x32K epi gelu");
      release(p); }
  }
  if (want("big")) {
    Problem p = make(4096, 4096, 4096, 330);
    printf("4096^3 (%d weight copies)\n", p.ncopy);
    run16<4, 2, 4, 64, 3, 1>(p, "k16 256x128 8w(64x64) 3x48K epi bias");
    run32<4, 2, 2, 2, 1, true, 64, 6, 1, false>(p, "k32 256x128 8w(64x64) ks1 6x24K epi bias");
    run32<2, 2, 4, 2, 1, true, 64, 6, 1, true>(p, "k32 256x128 4w(128x64)+4 loaders ks1 6x24K epi bias");
    release(p);
  }
  return 0;
}
