// What does a device-wide barrier cost inside a persistent kernel on MI355X (8 XCDs, per-XCD L2)?  Each phase: every
// workgroup writes 4 KB, release fence, arrives on a global counter, spins until all arrived, acquire fence, reads the
// 4 KB another workgroup (on another XCD) wrote.  Compared with the same phases as separate kernel launches in a hipGraph.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gridbar.hip -o tools/ubench/gridbar.bin && tools/ubench/gridbar.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();                                            // release: this workgroup's stores are visible device-wide
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 22)) { *fail = 1; ok = false; break; }     // never hang the GPU
    }
    __threadfence();
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void persistent(float* buf, unsigned* counter, unsigned* fail, int phases, float* out) {
  const int nb = gridDim.x, b = blockIdx.x;
  float acc = 0.f;
  for (int p = 0; p < phases; ++p) {
    float* mine = buf + ((size_t)(p & 1) * nb + b) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) mine[threadIdx.x + 256 * i] = acc + (float)(p + i);
    if (!grid_barrier(counter, (unsigned)(p + 1) * nb, fail)) return;
    const float* other = buf + ((size_t)(p & 1) * nb + (b + 1 + 8 * (p % 7)) % nb) * 1024;    // a neighbour on another XCD
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += __builtin_nontemporal_load(other + threadIdx.x + 256 * i);
  }
  out[b * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void phase_kernel(float* buf, int p, float* out) {
  const int nb = gridDim.x, b = blockIdx.x;
  float acc = out[b * 256 + threadIdx.x];
  if (p > 0) {
    const float* other = buf + ((size_t)((p - 1) & 1) * nb + (b + 1 + 8 * ((p - 1) % 7)) % nb) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += other[threadIdx.x + 256 * i];
  }
  float* mine = buf + ((size_t)(p & 1) * nb + b) * 1024;
#pragma unroll
  for (int i = 0; i < 4; ++i) mine[threadIdx.x + 256 * i] = acc + (float)(p + i);
  out[b * 256 + threadIdx.x] = acc;
}

int main() {
  const int phases = 200;
  for (int nb : {256, 512, 768}) {
    float *buf, *out; unsigned *ctr, *fail;
    CK(hipMalloc(&buf, (size_t)2 * nb * 4096)); CK(hipMalloc(&out, (size_t)nb * 1024)); CK(hipMalloc(&ctr, 8)); fail = ctr + 1;
    CK(hipMemset(out, 0, (size_t)nb * 1024));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemsetAsync(ctr, 0, 8, st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(persistent, dim3(nb), dim3(256), 0, st, buf, ctr, fail, phases, out);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    unsigned h[2]; CK(hipMemcpy(h, ctr, 8, hipMemcpyDeviceToHost));
    printf("persistent  wgs=%4d: %7.2f us per phase (write 4 KB + device barrier + read 4 KB)%s\n", nb, best * 1e3f / phases, h[1] ? "  [TIMED OUT]" : "");
    // the same as a graph of `phases` dependent launches
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(nb), dim3(256), 0, st, buf, p, out);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("graph       wgs=%4d: %7.2f us per launch (same phase as its own kernel)\n", nb, best * 1e3f / phases);
    CK(hipFree(buf)); CK(hipFree(out)); CK(hipFree(ctr));
  }
  return 0;
}
