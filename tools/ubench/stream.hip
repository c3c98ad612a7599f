// Ceiling probe for the operand paths of the 196-row GEMMs on MI355X: how many bytes per clock a CU can pull
//   (a) into LDS with global_load_lds_dwordx4 (1 KB per wave instruction), (b) into registers with global_load_dwordx4,
// as a function of waves per workgroup, loads in flight per wave, and where the data lives (one 400 KB panel shared by
// every workgroup = L2-hot activations; a private slab per workgroup of a 2 GB buffer = HBM-cold weights).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream.hip -o tools/ubench/stream.bin && tools/ubench/stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int D>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory"); }

// each wave: `iters` rounds of G loads (G KB), ring of D rounds in flight.  src advances by `stride` bytes per round per wave.
template <int G, int D, bool LDS>
__global__ __launch_bounds__(1024) void stream_kernel(const char* __restrict__ buf, size_t wg_stride, size_t wave_stride,
                                                       size_t wrap, int iters, unsigned* sink, size_t base0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t base = base0 + (size_t)blockIdx.x * wg_stride + (size_t)wave * wave_stride;
  u32x4 acc = {0, 0, 0, 0};
  if constexpr (LDS) {
    char* ring = smem + wave * (D * G * 1024);
    auto issue = [&](int it) {
      const size_t off = (base + (size_t)it * G * 1024) % wrap;
#pragma unroll
      for (int g = 0; g < G; ++g)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + off + g * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(ring + ((it % D) * G + g) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < D - 1; ++p) issue(p);
    for (int it = 0; it < iters; ++it) {
      if (it + D - 1 < iters) { issue(it + D - 1); wait_vm<(D - 1) * G>(); }
      else wait_vm<0>();
      // touch the landed round so the loads cannot be dropped
      acc.x ^= *reinterpret_cast<const unsigned*>(ring + (it % D) * G * 1024 + lane * 4);
    }
  } else {
    u32x4 r[D][G];
    auto issue = [&](int slot, int it) {
      const size_t off = (base + (size_t)it * G * 1024) % wrap;
#pragma unroll
      for (int g = 0; g < G; ++g) r[slot][g] = *reinterpret_cast<const u32x4*>(buf + off + g * 1024 + lane * 16);
    };
#pragma unroll
    for (int p = 0; p < D - 1; ++p) issue(p, p);
    for (int it0 = 0; it0 < iters; it0 += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const int it = it0 + s;
        if (it + D - 1 < iters) issue((s + D - 1) % D, it + D - 1);
#pragma unroll
        for (int g = 0; g < G; ++g) acc ^= r[s][g];
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int G, int D, bool LDS>
void run(const char* name, const char* buf, size_t bufsz, int nwg, int nw, bool hot, int iters, unsigned* sink) {
  const size_t round = (size_t)G * 1024;
  size_t wg_stride, wave_stride, wrap;
  if (hot) { wg_stride = 0; wave_stride = round; wrap = 448 * 1024; }           // every WG sweeps the same 448 KB panel
  else { wave_stride = round * iters; wg_stride = wave_stride * nw; wrap = bufsz; }   // private contiguous slab per wave
  const size_t lds = LDS ? (size_t)nw * D * G * 1024 : 0;
  if (lds > 160 * 1024) return;
  auto k = stream_kernel<G, D, LDS>;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(nwg), dim3(64 * nw), lds, 0, buf, wg_stride, wave_stride, wrap, iters, sink,
                       hot ? (size_t)0 : ((size_t)rep * 700 * 1024 * 1024) % (bufsz - 700ull * 1024 * 1024));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)nwg * nw * iters * round;
  const int cus = nwg < 256 ? nwg : 256;
  printf("%-4s %-4s G=%d D=%-2d waves=%-2d wgs=%-4d  %8.1f us  %7.2f TB/s  %6.1f B/clk/CU(2.4GHz)  in-flight/CU %4d KB\n", name, hot ? "hot" : "cold",
         G, D, nw, nwg, best * 1e3, bytes / best / 1e9, bytes / (best * 1e-3) / cus / 2.4e9, (int)((size_t)(nwg > 256 ? nwg / 256 : 1) * nw * (D - 1) * G));
}

int main() {
  const size_t bufsz = (size_t)3 << 30;
  char* buf; unsigned* sink;
  CK(hipMalloc(&buf, bufsz)); CK(hipMemset(buf, 1, bufsz)); CK(hipMalloc(&sink, 4));
  for (int hot = 1; hot >= 0; --hot) {
    // per-wave bytes fixed at 256 KB (hot) / 128 KB (cold: 256 WGs x 16 waves x 128 KB = 512 MB per launch)
    for (int nwg : {256, 512}) {
      for (int nw : {4, 8, 16}) {
        const int kb_per_wave = hot ? 256 : 128;
#define RUN(G, D, L) run<G, D, L>(L ? "lds" : "reg", buf, bufsz, nwg, nw, hot, kb_per_wave / G, sink)
        RUN(1, 2, true); RUN(1, 4, true); RUN(1, 8, true); RUN(2, 4, true); RUN(2, 8, true); RUN(4, 4, true); RUN(4, 8, true);
        RUN(1, 2, false); RUN(1, 4, false); RUN(2, 4, false); RUN(4, 4, false); RUN(8, 3, false);
      }
    }
  }
  return 0;
}
