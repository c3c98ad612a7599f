// Structure probe for the 196-row weight-streaming GEMMs (bf16 fragment-order operands, HBM-cold weights): a LEAN K-split
// kernel (compile-time shape, 3-D grid instead of a run-time tile map, every load of a wave requested up front, one LDS
// reduction, bias + packed bf16 store) over tile shapes / wave counts / ring depths, next to the floors of the launch
// structure itself (empty kernel, load-only kernel).  Every launch of the timed hipGraph reads its own copy of W.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/gemm_sm.hip -o tools/ubench/gemm_sm.bin && tools/ubench/gemm_sm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __host__ inline long packed_off(int row, int k, int K) {
  const int nkb = K >> 6, kb = k >> 6, kk = k & 63, g = kk >> 4, e = kk & 15, h = e >> 3, eh = e & 7;
  return (((((long)(row >> 4) * nkb + kb) * 2 + h) * 4 + g) * 16 + (row & 15)) * 8 + eh;
}

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }

struct Args {
  const __bf16* A; const __bf16* W; const float* bias; __bf16* C; float* Cf;
  int M, N, rb_max, mode;   // mode 0: full; 1: loads only (no MFMA, no epilogue)
  int ntg; long gA, gW, gC, gb;   // grouped launch: blockIdx.z = group * ntg + zt; byte strides per group
};

// grid = (8, mt, ntg): linear workgroup id = x + 8 (y + mt z) -> XCD x; tile_m = y, tile_n = 8 z + x (the M-tiles that share a
// W panel sit on one XCD, adjacent in dispatch order)
template <int MF, int NF, int WK, int NKB, int RING>
__global__ __launch_bounds__(64 * WK) void k_ksplit(const Args a) {
  constexpr int BM = MF * 16, BN = NF * 16, NT = 64 * WK, K = NKB * 64;
  constexpr int NKW = NKB / WK;                        // k-blocks per wave (NKB % WK == 0)
  constexpr int R = (RING == 0 || RING > NKW) ? NKW : RING;
  constexpr int LD = BN + 4;
  static_assert(NKB % WK == 0, "k-blocks must divide over the waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = blockIdx.z / a.ntg, zt = blockIdx.z - grp * a.ntg;
  const int tile_m = blockIdx.y, tile_n = zt * 8 + blockIdx.x;
  if (tile_n * BN >= a.N) return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  // epilogue operands first: bias of this thread's column group
  constexpr int CG = BN / 4;
  const int ec4 = (tid % CG) * 4;
  const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.bias) + grp * a.gb);
  float4 b4 = *reinterpret_cast<const float4*>(bias + n0 + ec4);

  const char* ap[MF];
  const char* wp[NF];
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    int rb = tile_m * MF + m;
    rb = rb < a.rb_max ? rb : a.rb_max;
    ap[m] = reinterpret_cast<const char*>(a.A) + grp * a.gA + ((long)rb * NKB + wk) * 2048 + lane * 16;
  }
#pragma unroll
  for (int n = 0; n < NF; ++n) wp[n] = reinterpret_cast<const char*>(a.W) + grp * a.gW + ((long)(tile_n * NF + n) * NKB + wk) * 2048 + lane * 16;

  bf16x8 av[R][MF][2], wv[R][NF][2];
  auto load = [&](int slot, int i) {
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      wv[slot][n][0] = *reinterpret_cast<const bf16x8*>(wp[n] + (long)i * WK * 2048);
      wv[slot][n][1] = *reinterpret_cast<const bf16x8*>(wp[n] + (long)i * WK * 2048 + 1024);
    }
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      av[slot][m][0] = *reinterpret_cast<const bf16x8*>(ap[m] + (long)i * WK * 2048);
      av[slot][m][1] = *reinterpret_cast<const bf16x8*>(ap[m] + (long)i * WK * 2048 + 1024);
    }
  };
  f32x4 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < R; ++i) load(i, i);
  if (a.mode == 1) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 x = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < R; ++i) {
#pragma unroll
      for (int m = 0; m < MF; ++m) { x ^= __builtin_bit_cast(u32x4, av[i][m][0]); x ^= __builtin_bit_cast(u32x4, av[i][m][1]); }
#pragma unroll
      for (int n = 0; n < NF; ++n) { x ^= __builtin_bit_cast(u32x4, wv[i][n][0]); x ^= __builtin_bit_cast(u32x4, wv[i][n][1]); }
    }
    if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345678u) a.C[0] = (__bf16)1.f;
    return;
  }
#pragma unroll
  for (int i = 0; i < NKW; ++i) {
    const int s = i % R;
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][m][0], wv[s][n][0], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][m][1], wv[s][n][1], acc[m][n], 0, 0, 0);
      }
    if (i + R < NKW) load(s, i + R);
  }
  // accumulators -> LDS slab of this wave (C layout: col = lane & 15, row = 4 (lane >> 4) + reg)
  float* slab = smem + (size_t)wk * BM * LD;
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(m * 16 + 4 * g + r) * LD + n * 16 + c] = acc[m][n][r];
  __syncthreads();
  for (int idx = tid; idx < BM * CG; idx += NT) {
    const int row = idx / CG, c4 = (idx % CG) * 4;
    const int gm = m0 + row;
    if (gm >= a.M) continue;
    float4 v = *reinterpret_cast<const float4*>(smem + row * LD + c4);
#pragma unroll
    for (int s = 1; s < WK; ++s) {
      const float4 t = *reinterpret_cast<const float4*>(smem + (size_t)s * BM * LD + row * LD + c4);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if constexpr (NT % CG != 0) b4 = *reinterpret_cast<const float4*>(bias + n0 + c4);
    v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(a.C) + grp * a.gC + 2 * packed_off(gm, n0 + c4, a.N)) = o;
    if (a.Cf) *reinterpret_cast<float4*>(a.Cf + (long)gm * a.N + n0 + c4) = v;
  }
}

// ------------------------------------------------------------------------------------------------ host
static float bf2f(__bf16 x) { return (float)x; }

struct Problem {
  int M, N, K, ncopy, G;
  __bf16 *A, *W, *C;
  float *bias, *Cf;
  std::vector<__bf16> hA, hW0;
  std::vector<float> hb;
};

static Problem make(int M, int N, int K, size_t mb, int G = 1) {
  Problem p;
  p.M = M; p.N = N; p.K = K; p.G = G;
  const int Mp = (M + 15) / 16 * 16;
  const size_t wbytes = (size_t)N * K * 2;
  p.ncopy = (int)std::max<size_t>(1, std::min<size_t>(64, mb * (1 << 20) / (wbytes * G)));
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
  CK(hipMalloc(&p.A, pa.size() * 2 * G));
  for (int g = 0; g < G; ++g) CK(hipMemcpy(reinterpret_cast<char*>(p.A) + g * pa.size() * 2, pa.data(), pa.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&p.W, wbytes * p.ncopy * G));
  for (int c = 0; c < p.ncopy * G; ++c) CK(hipMemcpy(reinterpret_cast<char*>(p.W) + c * wbytes, pw.data(), wbytes, hipMemcpyHostToDevice));
  CK(hipMalloc(&p.bias, N * 4));
  CK(hipMemcpy(p.bias, p.hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&p.C, (size_t)Mp * N * 2 * G));
  CK(hipMalloc(&p.Cf, (size_t)Mp * N * 4));
  return p;
}

static void release(Problem& p) { for (void* q : {(void*)p.A, (void*)p.W, (void*)p.C, (void*)p.Cf, (void*)p.bias}) CK(hipFree(q)); }

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

template <int MF, int NF, int WK, int NKB, int RING>
static void run(Problem& p, const char* tag) {
  constexpr int BM = MF * 16, BN = NF * 16;
  if (p.K != NKB * 64 || p.N % BN != 0) return;
  const int mt = (p.M + BM - 1) / BM, nt = p.N / BN, ntg = (nt + 7) / 8;
  const size_t lds = (size_t)WK * BM * (BN + 4) * 4;
  if (lds > 160 * 1024) { printf("  %-28s LDS %zu too large\n", tag, lds); return; }
  auto kern = k_ksplit<MF, NF, WK, NKB, RING>;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)kern));
  const size_t wbytes = (size_t)p.N * p.K * 2;
  const int Mp = (p.M + 15) / 16 * 16;
  Args a{p.A, p.W, p.bias, p.C, p.Cf, p.M, p.N, (p.M + 15) / 16 - 1, 0, ntg, (long)Mp * p.K * 2, (long)wbytes, (long)Mp * p.N * 2, 0};
  // correctness once
  CK(hipMemset(p.Cf, 0, (size_t)p.M * p.N * 4));
  hipLaunchKernelGGL(kern, dim3(8, mt, ntg * p.G), dim3(64 * WK), lds, 0, a);
  CK(hipDeviceSynchronize());
  std::vector<float> hc((size_t)p.M * p.N);
  CK(hipMemcpy(hc.data(), p.Cf, hc.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  unsigned s = 777u;
  for (int t = 0; t < 400; ++t) {
    s = s * 1664525u + 1013904223u;
    const int r = (t < 8) ? p.M - 1 - t : (int)((s >> 8) % p.M);
    s = s * 1664525u + 1013904223u;
    const int c = (int)((s >> 8) % p.N);
    double ref = p.hb[c];
    for (int k = 0; k < p.K; ++k) ref += (double)bf2f(p.hA[(size_t)r * p.K + k]) * (double)bf2f(p.hW0[(size_t)c * p.K + k]);
    worst = std::max(worst, std::fabs(ref - hc[(size_t)r * p.N + c]));
  }
  a.Cf = nullptr;
  const int nl = std::max(48, p.ncopy);
  Args a1 = a;
  float us[2];
  for (int mode = 0; mode < 2; ++mode) {
    a1.mode = mode;
    us[mode] = time_graph(nl, [&](int i, hipStream_t st) {
      Args b = a1;
      b.W = reinterpret_cast<const __bf16*>(reinterpret_cast<const char*>(p.W) + (size_t)(i % p.ncopy) * wbytes * p.G);
      hipLaunchKernelGGL(kern, dim3(8, mt, ntg * p.G), dim3(64 * WK), lds, st, b);
    });
  }
  printf("  %-28s %4d wgs x %2d waves  vgpr %3d lds %6zu  %7.2f us  (loads only %6.2f)  err %.1e\n", tag, mt * nt * p.G, WK, fa.numRegs, lds,
         us[0], us[1], worst);
}

int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : "";
  // floor: dependent empty launches
  {
    int* d;
    CK(hipMalloc(&d, 4));
    for (int wgs : {256, 672}) {
      float us = time_graph(64, [&](int, hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, st, d); });
      printf("empty kernel, %d wgs: %.2f us per dependent launch\n", wgs, us);
    }
    CK(hipFree(d));
  }
  auto want = [&](const char* n) { return !only[0] || strstr(only, n); };
  if (want("qkv")) {
    Problem p = make(196, 3072, 1024, 330);
    printf("val qkv 196x3072x1024 (%d weight copies)\n", p.ncopy);
    run<2, 2, 4, 16, 3>(p, "32x32 k4 ring3");
    run<2, 2, 4, 16, 0>(p, "32x32 k4 all");
    run<2, 2, 8, 16, 0>(p, "32x32 k8 all");
    run<4, 2, 4, 16, 0>(p, "64x32 k4 all");
    run<4, 2, 4, 16, 2>(p, "64x32 k4 ring2");
    run<4, 2, 8, 16, 0>(p, "64x32 k8 all");
    run<2, 4, 4, 16, 0>(p, "32x64 k4 all");
    run<2, 4, 8, 16, 0>(p, "32x64 k8 all");
    run<4, 3, 4, 16, 0>(p, "64x48 k4 all");
    run<4, 3, 8, 16, 0>(p, "64x48 k8 all");
    run<4, 4, 4, 16, 0>(p, "64x64 k4 all");
    run<4, 4, 4, 16, 2>(p, "64x64 k4 ring2");
    run<4, 4, 8, 16, 0>(p, "64x64 k8 all");
    run<3, 3, 4, 16, 0>(p, "48x48 k4 all");
    run<3, 3, 8, 16, 0>(p, "48x48 k8 all");
    run<3, 2, 8, 16, 0>(p, "48x32 k8 all");
    run<7, 2, 8, 16, 0>(p, "112x32 k8 all");
    run<7, 2, 16, 16, 0>(p, "112x32 k16 all");
    run<7, 4, 16, 16, 0>(p, "112x64 k16 all");
    release(p);
  }
  if (want("qkx")) {
    Problem p = make(196, 3072, 1024, 330);
    printf("val qkv 196x3072x1024, more tiles (%d weight copies)\n", p.ncopy);
    run<5, 2, 8, 16, 0>(p, "80x32 k8 all");
    run<6, 2, 8, 16, 0>(p, "96x32 k8 all");
    run<13, 2, 8, 16, 0>(p, "208x32 k8 all");
    run<3, 4, 8, 16, 0>(p, "48x64 k8 all");
    run<5, 4, 8, 16, 0>(p, "80x64 k8 all");
    run<7, 4, 8, 16, 0>(p, "112x64 k8 all");
    run<4, 4, 16, 16, 0>(p, "64x64 k16 all");
    run<4, 2, 16, 16, 0>(p, "64x32 k16 all");
    run<2, 2, 16, 16, 0>(p, "32x32 k16 all");
    run<4, 6, 8, 16, 0>(p, "64x96 k8 all");
    release(p);
  }
  if (want("dec")) {
    { Problem p = make(196, 768, 768, 330, 2);
      printf("dec proj 196x768x768 x2 (%d weight copies)\n", p.ncopy);
      run<2, 2, 4, 12, 0>(p, "32x32 k4 all");
      run<2, 2, 6, 12, 0>(p, "32x32 k6 all");
      run<2, 2, 12, 12, 0>(p, "32x32 k12 all");
      run<3, 2, 6, 12, 0>(p, "48x32 k6 all");
      run<3, 2, 12, 12, 0>(p, "48x32 k12 all");
      run<2, 3, 6, 12, 0>(p, "32x48 k6 all");
      run<4, 2, 6, 12, 0>(p, "64x32 k6 all");
      run<2, 1, 6, 12, 0>(p, "32x16 k6 all");
      release(p); }
    { Problem p = make(196, 2304, 768, 330, 2);
      printf("dec qkv 196x2304x768 x2 (%d weight copies)\n", p.ncopy);
      run<2, 2, 4, 12, 0>(p, "32x32 k4 all");
      run<4, 2, 6, 12, 0>(p, "64x32 k6 all");
      run<4, 4, 6, 12, 0>(p, "64x64 k6 all");
      run<4, 4, 12, 12, 0>(p, "64x64 k12 all");
      run<7, 4, 6, 12, 0>(p, "112x64 k6 all");
      run<7, 4, 12, 12, 0>(p, "112x64 k12 all");
      run<7, 2, 12, 12, 0>(p, "112x32 k12 all");
      run<4, 6, 6, 12, 0>(p, "64x96 k6 all");
      release(p); }
    { Problem p = make(196, 3072, 768, 330, 2);
      printf("dec fc1 196x3072x768 x2 (%d weight copies)\n", p.ncopy);
      run<2, 2, 4, 12, 0>(p, "32x32 k4 all");
      run<4, 4, 6, 12, 0>(p, "64x64 k6 all");
      run<4, 4, 12, 12, 0>(p, "64x64 k12 all");
      run<7, 4, 6, 12, 0>(p, "112x64 k6 all");
      run<7, 4, 12, 12, 0>(p, "112x64 k12 all");
      run<4, 6, 6, 12, 0>(p, "64x96 k6 all");
      run<4, 6, 12, 12, 0>(p, "64x96 k12 all");
      release(p); }
    { Problem p = make(196, 768, 3072, 330, 2);
      printf("dec fc2 196x768x3072 x2 (%d weight copies)\n", p.ncopy);
      run<2, 2, 4, 48, 3>(p, "32x32 k4 ring3");
      run<2, 2, 8, 48, 3>(p, "32x32 k8 ring3");
      run<2, 2, 16, 48, 0>(p, "32x32 k16 all");
      run<3, 2, 16, 48, 0>(p, "48x32 k16 all");
      run<2, 3, 16, 48, 0>(p, "32x48 k16 all");
      run<3, 2, 8, 48, 3>(p, "48x32 k8 ring3");
      run<2, 3, 8, 48, 3>(p, "32x48 k8 ring3");
      release(p); }
    { Problem p = make(196, 1024, 1792, 330, 2);
      printf("key2 196x1024x1792 x2 (%d weight copies)\n", p.ncopy);
      run<2, 2, 4, 28, 3>(p, "32x32 k4 ring3");
      run<2, 2, 7, 28, 0>(p, "32x32 k7 all");
      run<2, 2, 14, 28, 0>(p, "32x32 k14 all");
      run<4, 2, 14, 28, 0>(p, "64x32 k14 all");
      run<4, 2, 7, 28, 0>(p, "64x32 k7 all");
      release(p); }
  }
  if (want("dc2")) {
    { Problem p = make(196, 2304, 768, 330, 2);
      printf("dec qkv 196x2304x768 x2, more tiles (%d weight copies)\n", p.ncopy);
      run<2, 2, 4, 12, 0>(p, "32x32 k4 all");
      run<2, 2, 6, 12, 0>(p, "32x32 k6 all");
      run<2, 2, 12, 12, 0>(p, "32x32 k12 all");
      run<2, 4, 4, 12, 0>(p, "32x64 k4 all");
      run<2, 4, 6, 12, 0>(p, "32x64 k6 all");
      run<2, 4, 12, 12, 0>(p, "32x64 k12 all");
      run<3, 2, 6, 12, 0>(p, "48x32 k6 all");
      run<3, 2, 12, 12, 0>(p, "48x32 k12 all");
      run<3, 4, 6, 12, 0>(p, "48x64 k6 all");
      run<3, 4, 12, 12, 0>(p, "48x64 k12 all");
      run<1, 4, 12, 12, 0>(p, "16x64 k12 all");
      run<1, 2, 12, 12, 0>(p, "16x32 k12 all");
      release(p); }
    { Problem p = make(196, 3072, 768, 330, 2);
      printf("dec fc1 196x3072x768 x2, more tiles (%d weight copies)\n", p.ncopy);
      run<2, 2, 4, 12, 0>(p, "32x32 k4 all");
      run<2, 2, 6, 12, 0>(p, "32x32 k6 all");
      run<2, 4, 4, 12, 0>(p, "32x64 k4 all");
      run<2, 4, 6, 12, 0>(p, "32x64 k6 all");
      run<2, 4, 12, 12, 0>(p, "32x64 k12 all");
      run<3, 2, 6, 12, 0>(p, "48x32 k6 all");
      run<3, 4, 6, 12, 0>(p, "48x64 k6 all");
      run<3, 4, 12, 12, 0>(p, "48x64 k12 all");
      run<4, 2, 6, 12, 0>(p, "64x32 k6 all");
      release(p); }
    { Problem p = make(196, 768, 3072, 330, 2);
      printf("dec fc2 196x768x3072 x2, more tiles (%d weight copies)\n", p.ncopy);
      run<3, 2, 8, 48, 3>(p, "48x32 k8 ring3");
      run<3, 2, 8, 48, 2>(p, "48x32 k8 ring2");
      run<3, 2, 12, 48, 0>(p, "48x32 k12 all");
      run<3, 2, 12, 48, 2>(p, "48x32 k12 ring2");
      run<3, 2, 16, 48, 0>(p, "48x32 k16 all");
      run<3, 2, 6, 48, 4>(p, "48x32 k6 ring4");
      run<3, 2, 6, 48, 3>(p, "48x32 k6 ring3");
      run<2, 2, 12, 48, 0>(p, "32x32 k12 all");
      run<2, 2, 12, 48, 2>(p, "32x32 k12 ring2");
      run<4, 2, 8, 48, 3>(p, "64x32 k8 ring3");
      run<4, 2, 12, 48, 2>(p, "64x32 k12 ring2");
      run<3, 1, 8, 48, 3>(p, "48x16 k8 ring3");
      release(p); }
    { Problem p = make(196, 1024, 4096, 330);
      printf("val fc2 196x1024x4096, more tiles (%d weight copies)\n", p.ncopy);
      run<2, 2, 16, 64, 0>(p, "32x32 k16 all");
      run<3, 2, 8, 64, 3>(p, "48x32 k8 ring3");
      run<3, 2, 16, 64, 0>(p, "48x32 k16 all");
      run<3, 2, 16, 64, 2>(p, "48x32 k16 ring2");
      run<2, 2, 8, 64, 4>(p, "32x32 k8 ring4");
      run<4, 2, 16, 64, 2>(p, "64x32 k16 ring2");
      run<4, 1, 16, 64, 0>(p, "64x16 k16 all");
      release(p); }
    { Problem p = make(196, 1792, 1792, 330, 2);
      printf("key0 196x1792x1792 x2 (%d weight copies)\n", p.ncopy);
      run<2, 2, 4, 28, 3>(p, "32x32 k4 ring3");
      run<2, 2, 7, 28, 0>(p, "32x32 k7 all");
      run<4, 2, 7, 28, 0>(p, "64x32 k7 all");
      run<4, 4, 7, 28, 0>(p, "64x64 k7 all");
      run<4, 4, 14, 28, 0>(p, "64x64 k14 all");
      run<3, 4, 14, 28, 0>(p, "48x64 k14 all");
      run<3, 4, 7, 28, 0>(p, "48x64 k7 all");
      run<2, 4, 7, 28, 0>(p, "32x64 k7 all");
      run<2, 4, 14, 28, 0>(p, "32x64 k14 all");
      release(p); }
    { Problem p = make(196, 768, 1024, 330, 2);
      printf("dec_embed 196x768x1024 x2 (%d weight copies)\n", p.ncopy);
      run<2, 2, 8, 16, 0>(p, "32x32 k8 all");
      run<3, 2, 8, 16, 0>(p, "48x32 k8 all");
      run<2, 2, 16, 16, 0>(p, "32x32 k16 all");
      release(p); }
  }
  if (want("fc1")) {
    Problem p = make(196, 4096, 1024, 330);
    printf("val fc1 196x4096x1024 (%d weight copies)\n", p.ncopy);
    run<2, 2, 4, 16, 3>(p, "32x32 k4 ring3");
    run<2, 2, 4, 16, 0>(p, "32x32 k4 all");
    run<4, 2, 8, 16, 0>(p, "64x32 k8 all");
    run<4, 4, 4, 16, 0>(p, "64x64 k4 all");
    run<4, 4, 8, 16, 0>(p, "64x64 k8 all");
    run<2, 4, 8, 16, 0>(p, "32x64 k8 all");
    run<3, 4, 8, 16, 0>(p, "48x64 k8 all");
    release(p);
  }
  if (want("proj")) {
    Problem p = make(196, 1024, 1024, 330);
    printf("val proj 196x1024x1024 (%d weight copies)\n", p.ncopy);
    run<2, 2, 4, 16, 3>(p, "32x32 k4 ring3");
    run<2, 2, 4, 16, 0>(p, "32x32 k4 all");
    run<2, 2, 8, 16, 0>(p, "32x32 k8 all");
    run<2, 2, 16, 16, 0>(p, "32x32 k16 all");
    run<2, 1, 8, 16, 0>(p, "32x16 k8 all");
    run<1, 2, 8, 16, 0>(p, "16x32 k8 all");
    run<3, 2, 8, 16, 0>(p, "48x32 k8 all");
    run<4, 1, 8, 16, 0>(p, "64x16 k8 all");
    release(p);
  }
  if (want("fc2")) {
    Problem p = make(196, 1024, 4096, 330);
    printf("val fc2 196x1024x4096 (%d weight copies)\n", p.ncopy);
    run<2, 2, 4, 64, 3>(p, "32x32 k4 ring3");
    run<2, 2, 8, 64, 3>(p, "32x32 k8 ring3");
    run<2, 2, 8, 64, 4>(p, "32x32 k8 ring4");
    run<2, 2, 16, 64, 0>(p, "32x32 k16 all");
    run<2, 2, 16, 64, 2>(p, "32x32 k16 ring2");
    run<3, 2, 16, 64, 0>(p, "48x32 k16 all");
    run<2, 1, 16, 64, 0>(p, "32x16 k16 all");
    release(p);
  }
  return 0;
}
