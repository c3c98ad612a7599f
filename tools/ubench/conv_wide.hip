// Ablation of the wide 3x3 convolution tile (spann3r_amd/csrc/conv.hip, wide::conv3x3_wide_kernel): the full kernel next to
// variants without the W refills / the per-tap A re-loads from LDS / the halo loads from memory / the epilogue, to see which
// stream the launch waits for.  bf16 maps in and out, HIP events around 10 launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ispann3r_amd/csrc -Iinclude tools/ubench/conv_wide.hip spann3r_amd/csrc/error.cpp \
//         -Xclang -target-feature -Xclang -packed-fp32-ops -o tools/ubench/conv_wide.bin && tools/ubench/conv_wide.bin
#include "../../spann3r_amd/csrc/conv.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ABL>
static float run(const ConvArgs& a, dim3 grid, int reps) {
  auto kern = wide::conv3x3_wide_kernel<__bf16, 4, ABL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, wide::lds_bytes<4>()));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), wide::lds_bytes<4>(), 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  return best * 1000.f / reps;
}

int main() {
  const int shapes[][5] = {{9, 112, 112, 256, 256}, {1, 112, 112, 256, 256}, {1, 256, 256, 256, 256}, {9, 224, 224, 128, 128}};
  printf("%-24s %9s %9s %9s %9s %9s %9s %9s  (us per launch; TFLOP/s of the full kernel)\n", "B,H,W,Cin,Cout", "full", "noW", "noA", "noHalo",
         "noEpi", "noW+A", "mfma");
  for (auto& sh : shapes) {
    const int B = sh[0], H = sh[1], W = sh[2], Cin = sh[3], Cout = sh[4];
    const size_t nx = (size_t)B * H * W * Cin, no = (size_t)B * H * W * Cout, nw = (size_t)Cout * 9 * Cin;
    std::vector<unsigned short> hx(nx), hw(nw);
    for (size_t i = 0; i < nx; ++i) hx[i] = 0x3c00 + (rand() & 0xff);          // bf16 values around 0.008
    for (size_t i = 0; i < nw; ++i) hw[i] = 0x3a00 + (rand() & 0xff);
    void *x, *w, *o;
    float* bias;
    CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&o, no * 2)); CK(hipMalloc(&bias, Cout * 4));
    CK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, Cout * 4));
    ConvArgs a;
    a.x = x; a.w = reinterpret_cast<const __bf16*>(w); a.bias = bias; a.res1 = nullptr; a.res2 = nullptr; a.out = o;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.tiles_x = (W + wide::TW - 1) / wide::TW; a.tiles_y = (H + wide::TH - 1) / wide::TH;
    a.relu_in = 1; a.act = SP3_ACT_NONE; a.out_bf16 = 1; a.xcd_nb = 0;
    dim3 grid(a.tiles_x * a.tiles_y * B, Cout / 64);
    const int reps = 10;
    const float t0 = run<0>(a, grid, reps), t1 = run<1>(a, grid, reps), t2 = run<2>(a, grid, reps), t4 = run<4>(a, grid, reps),
                t8 = run<8>(a, grid, reps), t3 = run<3>(a, grid, reps), t15 = run<15>(a, grid, reps);
    char name[64];
    snprintf(name, sizeof name, "%d,%d,%d,%d,%d", B, H, W, Cin, Cout);
    printf("%-24s %9.2f %9.2f %9.2f %9.2f %9.2f %9.2f %9.2f  %.0f TFLOP/s, %d workgroups\n", name, t0, t1, t2, t4, t8, t3, t15,
           2.0 * B * H * W * Cout * 9.0 * Cin / t0 / 1e6, grid.x * grid.y);
    CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(o)); CK(hipFree(bias));
  }
  return 0;
}
