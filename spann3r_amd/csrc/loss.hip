// Training criterion: ConfLoss_t(Regr3D_t(L21, norm_mode='avg_dis', fix_first), alpha) forward and
// backward (reference: spann3r/loss.py:20-84,129-285, dust3r/losses.py:52-56).  Everything the reference does with ~40
// masked torch ops per frame pair (boolean indexing, norms, means) is five small launches over stacked buffers:
//   P  [E, B, HW, 3]  predicted pointmaps, one slab per loss entry e in the reference's order L0, L1, R1, ..., R_{n-1}
//   Cf [E, B, HW]     their confidences           G [n, B, HW, 3]  ground-truth points (world)     V [n, B, HW] u8 valid
//   pose0 [B, 16]     camera pose of view 1 (its inverse brings G into view 1's camera)
// frame(e) = (e + 1) / 2.  The prediction that normalises frame i (get_norm_factor over pts_l + [pts_r[-1]]) is entry
// 2i - (i > 0) for i < n-1 and entry E-1 for i = n-1.  All sums are accumulated in double, one workgroup per (slab, b) in
// a fixed order: deterministic.
#include "common.h"
#include <math.h>

namespace {

constexpr int NT = 1024;

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < NT / 64; ++i) r += sh[i];
  return r;
}

// inverse of an AFFINE 4x4 (camera poses: last row 0 0 0 1; the reference calls torch.linalg.inv and geotrf then uses
// Trf[:3,:3] and Trf[:3,3], dust3r/utils/geometry.py:66-68) -> R [9], t [3]
__device__ void inv_affine(const float* __restrict__ m, float* R, float* t) {
  const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
  const double A = e * i - f * h, Bc = -(d * i - f * g), Cc = d * h - e * g;
  const double det = a * A + b * Bc + c * Cc, id = 1.0 / det;
  const double r[9] = {A * id, -(b * i - c * h) * id, (b * f - c * e) * id,
                       Bc * id, (a * i - c * g) * id, -(a * f - c * d) * id,
                       Cc * id, -(a * h - b * g) * id, (a * e - b * d) * id};
  const double tx = m[3], ty = m[7], tz = m[11];
  for (int k = 0; k < 9; ++k) R[k] = (float)r[k];
  t[0] = (float)-(r[0] * tx + r[1] * ty + r[2] * tz);
  t[1] = (float)-(r[3] * tx + r[4] * ty + r[5] * tz);
  t[2] = (float)-(r[6] * tx + r[7] * ty + r[8] * tz);
}

__device__ __forceinline__ int norm_entry(int frame, int n) { return frame < n - 1 ? (frame == 0 ? 0 : 2 * frame - 1) : 2 * (n - 1) - 1; }

// stage 1: per (frame i, b): sum of ||p|| over the valid pixels for the normalising prediction and for the ground truth
// (in view 1's camera), and the number of valid pixels.  sums [n][B][3] (double)
__global__ __launch_bounds__(NT) void loss_norm_sums(const float* __restrict__ P, const float* __restrict__ G, const uint8_t* __restrict__ V,
                                                     const float* __restrict__ pose0, int n, int B, int HW, double* __restrict__ sums) {
  __shared__ double sh[NT / 64];
  __shared__ float Rt[12];
  const int i = blockIdx.x, b = blockIdx.y;
  if (threadIdx.x == 0) inv_affine(pose0 + b * 16, Rt, Rt + 9);
  __syncthreads();
  const float* p = P + ((int64_t)norm_entry(i, n) * B + b) * HW * 3;
  const float* g = G + ((int64_t)i * B + b) * HW * 3;
  const uint8_t* v = V + ((int64_t)i * B + b) * HW;
  double sp = 0.0, sg = 0.0, cnt = 0.0;
  for (int x = threadIdx.x; x < HW; x += NT) {
    if (!v[x]) continue;
    const float px = p[3 * x], py = p[3 * x + 1], pz = p[3 * x + 2];
    sp += (double)sqrtf(px * px + py * py + pz * pz);
    const float gx = g[3 * x], gy = g[3 * x + 1], gz = g[3 * x + 2];
    const float tx = Rt[0] * gx + Rt[1] * gy + Rt[2] * gz + Rt[9];
    const float ty = Rt[3] * gx + Rt[4] * gy + Rt[5] * gz + Rt[10];
    const float tz = Rt[6] * gx + Rt[7] * gy + Rt[8] * gz + Rt[11];
    sg += (double)sqrtf(tx * tx + ty * ty + tz * tz);
    cnt += 1.0;
  }
  sp = block_sum(sp, sh); sg = block_sum(sg, sh); cnt = block_sum(cnt, sh);
  if (threadIdx.x == 0) { double* o = sums + ((int64_t)i * B + b) * 3; o[0] = sp; o[1] = sg; o[2] = cnt; }
}

// stage 2 (one thread per b): fp_b, fg_b (clipped at 1e-8), the factor loss and d(factor loss)/d fp_b.
// fac [B][4] = fp, fg, dfactor/dfp, (pred factor clipped ? 0 : 1 / (N_total + 1e-8)) ; scal[0] = factor loss
__global__ void loss_factors(const double* __restrict__ sums, int n, int B, int fix_first, float* __restrict__ fac, float* __restrict__ scal) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int nf = fix_first ? 1 : n;
  double ntot = 0.0;
  for (int i = 0; i < nf; ++i)
    for (int b = 0; b < B; ++b) ntot += sums[((int64_t)i * B + b) * 3 + 2];
  const double den = ntot + 1e-8;
  int k = 0;
  for (int b = 0; b < B; ++b) {
    double sp = 0.0, sg = 0.0;
    for (int i = 0; i < nf; ++i) { sp += sums[((int64_t)i * B + b) * 3]; sg += sums[((int64_t)i * B + b) * 3 + 1]; }
    const float fp = (float)(sp / den), fg = (float)(sg / den);
    fac[4 * b] = fmaxf(fp, 1e-8f);
    fac[4 * b + 1] = fmaxf(fg, 1e-8f);
    fac[4 * b + 3] = fp > 1e-8f ? (float)(1.0 / den) : 0.f;
  }
  // filter_factor = pr_factor[pr_factor > gt_factor]; (filter_factor - gt_factor).abs().mean() broadcasts k x B
  for (int b = 0; b < B; ++b) k += fac[4 * b] > fac[4 * b + 1];
  double fl = 0.0;
  for (int j = 0; j < B; ++j) {
    double dj = 0.0;
    if (fac[4 * j] > fac[4 * j + 1]) {
      for (int b = 0; b < B; ++b) {
        const float df = fac[4 * j] - fac[4 * b + 1];
        fl += fabs((double)df);
        dj += df > 0.f ? 1.0 : (df < 0.f ? -1.0 : 0.0);
      }
    }
    fac[4 * j + 2] = k ? (float)(dj / ((double)k * B)) : 0.f;
  }
  scal[0] = k ? (float)(fl / ((double)k * B)) : 0.f;
}

// stage 3: per (entry e, b): over the valid pixels x = p / fp - g' / fg, d = ||x||:
//   terms [E][B][6] (double) = sum(d * c - alpha * log c), sum d, sum c, count, sum c * (x . p) / (d * fp^2)  [-> d loss / d fp], -
__global__ __launch_bounds__(NT) void loss_terms(const float* __restrict__ P, const float* __restrict__ Cf, const float* __restrict__ G,
                                                 const uint8_t* __restrict__ V, const float* __restrict__ pose0,
                                                 const float* __restrict__ fac, int B, int HW, float alpha, double* __restrict__ terms) {
  __shared__ double sh[NT / 64];
  __shared__ float Rt[12];
  const int e = blockIdx.x, b = blockIdx.y, i = (e + 1) >> 1;
  if (threadIdx.x == 0) inv_affine(pose0 + b * 16, Rt, Rt + 9);
  __syncthreads();
  const float ifp = 1.0f / fac[4 * b], ifg = 1.0f / fac[4 * b + 1];
  const float* p = P + ((int64_t)e * B + b) * HW * 3;
  const float* c = Cf + ((int64_t)e * B + b) * HW;
  const float* g = G + ((int64_t)i * B + b) * HW * 3;
  const uint8_t* v = V + ((int64_t)i * B + b) * HW;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
  for (int x = threadIdx.x; x < HW; x += NT) {
    if (!v[x]) continue;
    const float px = p[3 * x], py = p[3 * x + 1], pz = p[3 * x + 2];
    const float gx = g[3 * x], gy = g[3 * x + 1], gz = g[3 * x + 2];
    const float dx = px * ifp - (Rt[0] * gx + Rt[1] * gy + Rt[2] * gz + Rt[9]) * ifg;
    const float dy = py * ifp - (Rt[3] * gx + Rt[4] * gy + Rt[5] * gz + Rt[10]) * ifg;
    const float dz = pz * ifp - (Rt[6] * gx + Rt[7] * gy + Rt[8] * gz + Rt[11]) * ifg;
    const float d = sqrtf(dx * dx + dy * dy + dz * dz), cc = c[x];
    s0 += (double)(d * cc - alpha * logf(cc));
    s1 += (double)d; s2 += (double)cc; s3 += 1.0;
    if (d > 0.f) s4 += (double)(cc * (dx * px + dy * py + dz * pz) / d) * (double)(ifp * ifp);
  }
  s0 = block_sum(s0, sh); s1 = block_sum(s1, sh); s2 = block_sum(s2, sh); s3 = block_sum(s3, sh); s4 = block_sum(s4, sh);
  if (threadIdx.x == 0) { double* o = terms + ((int64_t)e * B + b) * 6; o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = s4; o[5] = 0.0; }
}

// stage 4 (one thread): scal[1] = loss, scal[2..] = details (conf_loss per entry e at 8+e is not exported; see below);
//   went [E] = 2 / (E * count_e) (the weight of one pixel of entry e in the loss), dfp [B] = d loss / d fp_b,
//   ent [E][2] = (mean error, conf loss) per entry
// scal: [0] factor loss, [1] loss, [2] conf_loss_1, [3] conf_loss2, [4] conf_mean, [5] pts3d_1, [6] pts3d_2
__global__ void loss_finish(const double* __restrict__ terms, const float* __restrict__ fac, int E, int B, float* __restrict__ scal,
                            float* __restrict__ went, float* __restrict__ dfp, float* __restrict__ ent) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double loss = 0.0, conf_sum = 0.0;
  for (int e = 0; e < E; ++e) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (int b = 0; b < B; ++b) { const double* t = terms + ((int64_t)e * B + b) * 6; s0 += t[0]; s1 += t[1]; s2 += t[2]; cnt += t[3]; }
    const double cl = cnt > 0.0 ? 2.0 * s0 / cnt : 0.0;
    loss += cl;
    conf_sum += s2 / cnt;                              // (an empty entry makes the reference's monitoring value NaN too)
    went[e] = cnt > 0.0 ? (float)(2.0 / ((double)E * cnt)) : 0.f;
    ent[2 * e] = (float)(s1 / cnt);                    // mean Euclidean error of the entry (monitoring)
    ent[2 * e + 1] = (float)cl;
    if (e == 0) { scal[2] = (float)cl; scal[5] = (float)(s1 / cnt); }
    if (e == 1) { scal[3] = (float)cl; scal[6] = (float)(s1 / cnt); }
  }
  scal[1] = (float)(loss / E);
  scal[4] = (float)(conf_sum / E);
  for (int b = 0; b < B; ++b) {
    double gsum = 0.0;
    for (int e = 0; e < E; ++e) gsum -= (double)went[e] * terms[((int64_t)e * B + b) * 6 + 4];
    dfp[b] = (float)gsum;                              // d loss / d fp_b (the factor loss' share is fac[4 b + 2])
  }
}

// backward: gradients of gscale[0] * loss + gscale[1] * factor loss w.r.t. P and Cf
__global__ __launch_bounds__(256) void loss_backward(const float* __restrict__ P, const float* __restrict__ Cf, const float* __restrict__ G,
                                                     const uint8_t* __restrict__ V, const float* __restrict__ pose0,
                                                     const float* __restrict__ fac, const float* __restrict__ went,
                                                     const float* __restrict__ dfp, const float* __restrict__ gscale, int n, int B, int HW,
                                                     float alpha, int fix_first, float* __restrict__ dP, float* __restrict__ dC) {
  __shared__ float Rt[12];
  const int e = blockIdx.y, b = blockIdx.z, i = (e + 1) >> 1;
  if (threadIdx.x == 0) inv_affine(pose0 + b * 16, Rt, Rt + 9);
  __syncthreads();
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= HW) return;
  const int64_t o = ((int64_t)e * B + b) * HW + x;
  float gx_ = 0.f, gy_ = 0.f, gz_ = 0.f, gc = 0.f;
  if (V[((int64_t)i * B + b) * HW + x]) {
    const float gs = gscale[0], w = went[e] * gs;
    const float ifp = 1.0f / fac[4 * b], ifg = 1.0f / fac[4 * b + 1];
    const float px = P[3 * o], py = P[3 * o + 1], pz = P[3 * o + 2];
    const float* g = G + (((int64_t)i * B + b) * HW + x) * 3;
    const float dx = px * ifp - (Rt[0] * g[0] + Rt[1] * g[1] + Rt[2] * g[2] + Rt[9]) * ifg;
    const float dy = py * ifp - (Rt[3] * g[0] + Rt[4] * g[1] + Rt[5] * g[2] + Rt[10]) * ifg;
    const float dz = pz * ifp - (Rt[6] * g[0] + Rt[7] * g[1] + Rt[8] * g[2] + Rt[11]) * ifg;
    const float d = sqrtf(dx * dx + dy * dy + dz * dz), cc = Cf[o];
    if (d > 0.f) { const float s = w * cc * ifp / d; gx_ = s * dx; gy_ = s * dy; gz_ = s * dz; }
    gc = w * (d - alpha / cc);
    // this prediction also sets the scale of its batch element: d fp_b / d p = p / (||p|| (N + 1e-8))
    if (e == norm_entry(i, n) && (!fix_first || i == 0)) {
      const float nrm = sqrtf(px * px + py * py + pz * pz);
      if (nrm > 0.f) { const float s = (gs * dfp[b] + gscale[1] * fac[4 * b + 2]) * fac[4 * b + 3] / nrm; gx_ += s * px; gy_ += s * py; gz_ += s * pz; }
    }
  }
  dP[3 * o] = gx_; dP[3 * o + 1] = gy_; dP[3 * o + 2] = gz_;
  dC[o] = gc;
}

}  // namespace

// ws: device scratch of sp3_conf_loss_ws_bytes(n, B) bytes (kept between forward and backward).
extern "C" int64_t sp3_conf_loss_ws_bytes(int n, int B) {
  const int E = 2 * (n - 1);
  return (int64_t)sizeof(double) * ((int64_t)n * B * 3 + (int64_t)E * B * 6) + (int64_t)sizeof(float) * (4 * B + 16 + E + B + 2 * E) + 64;
}

struct LossWs { double* sums; double* terms; float* fac; float* scal; float* went; float* dfp; float* ent; };
static LossWs loss_ws(void* ws, int n, int B) {
  const int E = 2 * (n - 1);
  LossWs w;
  w.sums = reinterpret_cast<double*>(ws);
  w.terms = w.sums + (int64_t)n * B * 3;
  w.fac = reinterpret_cast<float*>(w.terms + (int64_t)E * B * 6);
  w.scal = w.fac + 4 * B;
  w.went = w.scal + 16;
  w.dfp = w.went + E;
  w.ent = w.dfp + B;
  return w;
}

extern "C" int sp3_conf_loss_forward(const float* P, const float* Cf, const float* G, const uint8_t* V, const float* pose0, int n, int B,
                                     int HW, float alpha, int fix_first, void* ws, float* out7, float* ent_out, void* stream) {
  SP3_CHECK(P && Cf && G && V && pose0 && ws && out7, "sp3_conf_loss_forward: null pointer");
  SP3_CHECK(n >= 2 && B > 0 && HW > 0 && alpha > 0.f, "sp3_conf_loss_forward: bad geometry n=%d B=%d HW=%d alpha=%g", n, B, HW, (double)alpha);
  SP3_CHECK((reinterpret_cast<uintptr_t>(ws) & 7) == 0, "sp3_conf_loss_forward: workspace must be 8-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int E = 2 * (n - 1);
  LossWs w = loss_ws(ws, n, B);
  hipLaunchKernelGGL(loss_norm_sums, dim3(n, B), dim3(NT), 0, st, P, G, V, pose0, n, B, HW, w.sums);
  hipLaunchKernelGGL(loss_factors, dim3(1), dim3(64), 0, st, w.sums, n, B, fix_first, w.fac, w.scal);
  hipLaunchKernelGGL(loss_terms, dim3(E, B), dim3(NT), 0, st, P, Cf, G, V, pose0, w.fac, B, HW, alpha, w.terms);
  hipLaunchKernelGGL(loss_finish, dim3(1), dim3(64), 0, st, w.terms, w.fac, E, B, w.scal, w.went, w.dfp, w.ent);
  SP3_LAUNCH_CHECK("sp3_conf_loss_forward");
  if (hipMemcpyAsync(out7, w.scal, 7 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess ||
      (ent_out && hipMemcpyAsync(ent_out, w.ent, 2 * E * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)) {
    sp3_set_error("sp3_conf_loss_forward: copy of the results failed");
    return 2;
  }
  return 0;
}

extern "C" int sp3_conf_loss_backward(const float* P, const float* Cf, const float* G, const uint8_t* V, const float* pose0, int n, int B,
                                      int HW, float alpha, int fix_first, const void* ws, const float* grad_scale2, float* dP, float* dC,
                                      void* stream) {
  SP3_CHECK(P && Cf && G && V && pose0 && ws && grad_scale2 && dP && dC, "sp3_conf_loss_backward: null pointer");
  SP3_CHECK(n >= 2 && B > 0 && HW > 0, "sp3_conf_loss_backward: bad geometry");
  const int E = 2 * (n - 1);
  LossWs w = loss_ws(const_cast<void*>(ws), n, B);
  hipLaunchKernelGGL(loss_backward, dim3((HW + 255) / 256, E, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P, Cf, G, V, pose0,
                     w.fac, w.went, w.dfp, grad_scale2, n, B, HW, alpha, fix_first, dP, dC);
  SP3_LAUNCH_CHECK("sp3_conf_loss_backward");
  return 0;
}

// =====================================================================================================================
// Regr3D_t_ScaleShiftInv(L21, gt_scale) (spann3r/loss.py:292-368; the validation criterion of spann3r/training.py:39,152):
// avg_dis normalisation of the predictions (and of the ground truth unless gt_scale), joint median-depth shift, joint
// median-centre / median-norm scale, then the sum over the entries of the mean Euclidean error.  Forward only (it runs under
// torch.no_grad).  The medians (torch.nanmedian: the LOWER middle element) are radix selections on the device: 4 passes of
// 8 bits over the order-preserving integer image of the floats, integer histograms (atomics commute: deterministic).
namespace {

__device__ __forceinline__ unsigned fkey(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// state[row] = {prefix, k, done-mask, count}; hist[row][256]
__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ vals, int64_t L, const unsigned* __restrict__ state,
                                                          unsigned* __restrict__ hist, int pass) {
  const int row = blockIdx.y;
  const unsigned prefix = state[4 * row], mask = pass == 0 ? 0u : (0xffffffffu << (32 - 8 * pass));
  const int shift = 24 - 8 * pass;
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const float* v = vals + (int64_t)row * L;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < L; i += (int64_t)gridDim.x * 256) {
    const float f = v[i];
    if (f != f) continue;                                   // NaN = invalid point
    const unsigned k = fkey(f);
    if ((k & mask) == (prefix & mask)) atomicAdd(&h[(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[row * 256 + threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void select_pick_kernel(unsigned* __restrict__ state, unsigned* __restrict__ hist, int pass, float* __restrict__ out) {
  const int row = blockIdx.x;
  __shared__ unsigned h[256];
  h[threadIdx.x] = hist[row * 256 + threadIdx.x];
  hist[row * 256 + threadIdx.x] = 0;                        // ready for the next pass
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned k = state[4 * row + 1];
    if (pass == 0) {
      unsigned tot = 0;
      for (int d = 0; d < 256; ++d) tot += h[d];
      state[4 * row + 3] = tot;
      k = tot ? (tot - 1) / 2 : 0;                          // torch.nanmedian: the lower of the two middle elements
    }
    unsigned cum = 0;
    int d = 0;
    for (; d < 255; ++d) {
      if (cum + h[d] > k) break;
      cum += h[d];
    }
    state[4 * row] |= (unsigned)d << (24 - 8 * pass);
    state[4 * row + 1] = k - cum;
    if (pass == 3) out[row] = state[4 * row + 3] ? fkey_inv(state[4 * row]) : __uint_as_float(0x7fc00000u);
  }
}

struct SsiArgs {
  const float* P; const float* G; const uint8_t* V; const float* pose0; const float* fac;   // fac [B][4]: fp, fg, ...
  const float* med;                                                                          // medians so far (layout below)
  float* vals;                                                                               // [rows][n*HW]
  int n, B, HW, gt_scale, stage;
};
// med layout: [0, 2B): shift z (gt rows 0..B-1, pred B..2B-1); [2B, 8B): centres (gt x, y, z, pred x, y, z; B each); [8B, 10B): scales
__device__ __forceinline__ void ssi_points(const SsiArgs& a, const float* Rt, int i, int b, int x, float (&g)[3], float (&p)[3]) {
  const float* gp = a.G + (((int64_t)i * a.B + b) * a.HW + x) * 3;
  const float ifg = a.gt_scale ? 1.0f : 1.0f / a.fac[4 * b + 1], ifp = 1.0f / a.fac[4 * b];
  g[0] = (Rt[0] * gp[0] + Rt[1] * gp[1] + Rt[2] * gp[2] + Rt[9]) * ifg;
  g[1] = (Rt[3] * gp[0] + Rt[4] * gp[1] + Rt[5] * gp[2] + Rt[10]) * ifg;
  g[2] = (Rt[6] * gp[0] + Rt[7] * gp[1] + Rt[8] * gp[2] + Rt[11]) * ifg;
  const float* pp = a.P + (((int64_t)norm_entry(i, a.n) * a.B + b) * a.HW + x) * 3;
  p[0] = pp[0] * ifp; p[1] = pp[1] * ifp; p[2] = pp[2] * ifp;
}

// stage 0: z of gt / pred (2B rows); stage 1: shifted x, y, z of gt / pred (6B rows); stage 2: |p - centre| (2B rows)
__global__ __launch_bounds__(256) void ssi_values_kernel(const SsiArgs a) {
  __shared__ float Rt[12];
  const int i = blockIdx.y, b = blockIdx.z;
  if (threadIdx.x == 0) inv_affine(a.pose0 + b * 16, Rt, Rt + 9);
  __syncthreads();
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= a.HW) return;
  const int64_t L = (int64_t)a.n * a.HW, o = (int64_t)i * a.HW + x;
  const bool ok = a.V[((int64_t)i * a.B + b) * a.HW + x] != 0;
  const float qnan = __uint_as_float(0x7fc00000u);
  float g[3], p[3];
  ssi_points(a, Rt, i, b, x, g, p);
  const int B = a.B;
  if (a.stage == 0) {
    a.vals[(int64_t)b * L + o] = ok ? g[2] : qnan;
    a.vals[(int64_t)(B + b) * L + o] = ok ? p[2] : qnan;
    return;
  }
  g[2] -= a.med[b]; p[2] -= a.med[B + b];
  if (a.stage == 1) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a.vals[(int64_t)(c * B + b) * L + o] = ok ? g[c] : qnan;
      a.vals[(int64_t)((3 + c) * B + b) * L + o] = ok ? p[c] : qnan;
    }
    return;
  }
  float dg = 0.f, dp = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float u = g[c] - a.med[2 * B + c * B + b], w = p[c] - a.med[2 * B + (3 + c) * B + b];
    dg += u * u; dp += w * w;
  }
  a.vals[(int64_t)b * L + o] = ok ? sqrtf(dg) : qnan;
  a.vals[(int64_t)(B + b) * L + o] = ok ? sqrtf(dp) : qnan;
}

// per (entry e, b): sum over the valid pixels of |pred' - gt'| and the count -> terms [E][B][2] (double)
__global__ __launch_bounds__(NT) void ssi_terms_kernel(const SsiArgs a, double* __restrict__ terms) {
  __shared__ double sh[NT / 64];
  __shared__ float Rt[12];
  const int e = blockIdx.x, b = blockIdx.y, i = (e + 1) >> 1, B = a.B;
  if (threadIdx.x == 0) inv_affine(a.pose0 + b * 16, Rt, Rt + 9);
  __syncthreads();
  const float ifg = a.gt_scale ? 1.0f : 1.0f / a.fac[4 * b + 1], ifp = 1.0f / a.fac[4 * b];
  const float sg = a.med[8 * B + b], sp = fminf(fmaxf(a.med[9 * B + b], 1e-3f), 1e3f);
  const float mp = a.gt_scale ? sg / sp : sp / sg, mg = a.gt_scale ? 1.0f : sg / sp;
  const float zg = a.med[b], zp = a.med[B + b];
  const float* p = a.P + ((int64_t)e * B + b) * a.HW * 3;
  const float* gq = a.G + ((int64_t)i * B + b) * a.HW * 3;
  const uint8_t* v = a.V + ((int64_t)i * B + b) * a.HW;
  double s = 0.0, c = 0.0;
  for (int x = threadIdx.x; x < a.HW; x += NT) {
    if (!v[x]) continue;
    const float gx = gq[3 * x], gy = gq[3 * x + 1], gz = gq[3 * x + 2];
    const float tx = (Rt[0] * gx + Rt[1] * gy + Rt[2] * gz + Rt[9]) * ifg * mg;
    const float ty = (Rt[3] * gx + Rt[4] * gy + Rt[5] * gz + Rt[10]) * ifg * mg;
    const float tz = ((Rt[6] * gx + Rt[7] * gy + Rt[8] * gz + Rt[11]) * ifg - zg) * mg;
    const float dx = p[3 * x] * ifp * mp - tx, dy = p[3 * x + 1] * ifp * mp - ty, dz = (p[3 * x + 2] * ifp - zp) * mp - tz;
    s += (double)sqrtf(dx * dx + dy * dy + dz * dz);
    c += 1.0;
  }
  s = block_sum(s, sh); c = block_sum(c, sh);
  if (threadIdx.x == 0) { terms[((int64_t)e * B + b) * 2] = s; terms[((int64_t)e * B + b) * 2 + 1] = c; }
}

// out: [0] loss (sum over entries of the mean error), [1] factor loss, [2] gt_shift_z mean, [3] pred_shift_z mean, [4] gt_scale mean,
// [5] pred_scale mean (clipped), [6 + e] mean error of entry e
__global__ void ssi_finish_kernel(const double* __restrict__ terms, const float* __restrict__ med, const float* __restrict__ scal, int E, int B,
                                  int gt_scale, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double loss = 0.0;
  for (int e = 0; e < E; ++e) {
    double s = 0.0, c = 0.0;
    for (int b = 0; b < B; ++b) { s += terms[((int64_t)e * B + b) * 2]; c += terms[((int64_t)e * B + b) * 2 + 1]; }
    out[6 + e] = (float)(s / c);
    loss += s / c;
  }
  out[0] = (float)loss;
  out[1] = gt_scale ? 0.f : scal[0];
  double m[4] = {0, 0, 0, 0};
  for (int b = 0; b < B; ++b) {
    m[0] += med[b]; m[1] += med[B + b]; m[2] += med[8 * B + b];
    m[3] += fminf(fmaxf(med[9 * B + b], 1e-3f), 1e3f);
  }
  for (int k = 0; k < 4; ++k) out[2 + k] = (float)(m[k] / B);
}

}  // namespace

// floats of device scratch the forward needs
extern "C" int64_t sp3_ssi_loss_ws_bytes(int n, int B, int HW) {
  const int64_t rows = 6 * (int64_t)B;
  return (int64_t)sizeof(double) * ((int64_t)n * B * 3 + (int64_t)2 * (n - 1) * B * 2) + 4 * (rows * (int64_t)n * HW + rows * 256 + rows * 4 + 10 * B + 4 * B + 16) + 256;
}

extern "C" int sp3_ssi_loss_forward(const float* P, const float* G, const uint8_t* V, const float* pose0, int n, int B, int HW, int fix_first,
                                    int gt_scale, void* ws, float* out, void* stream) {
  SP3_CHECK(P && G && V && pose0 && ws && out && n >= 2 && B > 0 && HW > 0, "sp3_ssi_loss_forward: bad arguments");
  SP3_CHECK((reinterpret_cast<uintptr_t>(ws) & 7) == 0 && 6 * B <= 65535, "sp3_ssi_loss_forward: workspace alignment / batch");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int E = 2 * (n - 1);
  const int64_t L = (int64_t)n * HW, rows = 6 * (int64_t)B;
  double* sums = reinterpret_cast<double*>(ws);
  double* terms = sums + (int64_t)n * B * 3;
  float* vals = reinterpret_cast<float*>(terms + (int64_t)E * B * 2);
  unsigned* hist = reinterpret_cast<unsigned*>(vals + rows * L);
  unsigned* state = hist + rows * 256;
  float* med = reinterpret_cast<float*>(state + rows * 4);
  float* fac = med + 10 * B;
  float* scal = fac + 4 * B;
  hipLaunchKernelGGL(loss_norm_sums, dim3(n, B), dim3(NT), 0, st, P, G, V, pose0, n, B, HW, sums);
  hipLaunchKernelGGL(loss_factors, dim3(1), dim3(64), 0, st, sums, n, B, fix_first, fac, scal);
  SsiArgs a{P, G, V, pose0, fac, med, vals, n, B, HW, gt_scale, 0};
  const int nrows[3] = {2 * B, 6 * B, 2 * B};
  const int moff[3] = {0, 2 * B, 8 * B};
  const int hb = (int)((L + 256 * 16 - 1) / (256 * 16)) < 1 ? 1 : (int)((L + 256 * 16 - 1) / (256 * 16));
  for (int stage = 0; stage < 3; ++stage) {
    a.stage = stage;
    hipLaunchKernelGGL(ssi_values_kernel, dim3((HW + 255) / 256, n, B), dim3(256), 0, st, a);
    if (hipMemsetAsync(hist, 0, (size_t)(rows * 256 + rows * 4) * 4, st) != hipSuccess) { sp3_set_error("sp3_ssi_loss_forward: memset failed"); return 2; }
    for (int pass = 0; pass < 4; ++pass) {
      hipLaunchKernelGGL(select_hist_kernel, dim3(hb, nrows[stage]), dim3(256), 0, st, vals, L, state, hist, pass);
      hipLaunchKernelGGL(select_pick_kernel, dim3(nrows[stage]), dim3(256), 0, st, state, hist, pass, med + moff[stage]);
    }
  }
  hipLaunchKernelGGL(ssi_terms_kernel, dim3(E, B), dim3(NT), 0, st, a, terms);
  hipLaunchKernelGGL(ssi_finish_kernel, dim3(1), dim3(64), 0, st, terms, med, scal, E, B, gt_scale, out);
  SP3_LAUNCH_CHECK("sp3_ssi_loss_forward");
  return 0;
}
