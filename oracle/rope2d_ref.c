/* CPU restatement (plain C) of the reference's one native routine: 2-D RoPE.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows the CPU twin the reference ships next to its CUDA kernel:
 *   /root/reference/croco/models/curope/curope.cpp:11-47 (rope_2d_cpu)
 * tokens [B,N,H,D] fp32 contiguous, positions int64 [B,N,2] (y, x), in place:
 *   Q = D/4; for axis X in {0 (y), 1 (x)}: u = tok[d + X*2Q], v = tok[d + Q + X*2Q], d in [0,Q)
 *   angle = fwd * pos / base^(d/Q);  u' = u cos - v sin;  v' = v cos + u sin
 * Built by oracle/Makefile into oracle/_ref/librope2d_ref.so and used by tests/test_rope_cref.py to pin both the
 * torch oracle (oracle/spann3r_oracle.py:rope2d) and, on the GPU box, sp3_rope_2d.
 */
#include <math.h>
#include <stdint.h>

void rope2d_ref(float* tok, const int64_t* pos, int B, int N, int H, int D, float base, float fwd) {
  const int Q = D / 4;
  for (int b = 0; b < B; ++b)
    for (int x = 0; x < 2; ++x)
      for (int n = 0; n < N; ++n) {
        const int p = (int)pos[((int64_t)b * N + n) * 2 + x];
        for (int h = 0; h < H; ++h) {
          float* t = tok + (((int64_t)b * N + n) * H + h) * D;
          for (int d = 0; d < Q; ++d) {
            const float u = t[d + x * 2 * Q], v = t[d + Q + x * 2 * Q];
            const float ang = fwd * p / powf(base, d / (float)Q);
            const float c = cosf(ang), s = sinf(ang);
            t[d + x * 2 * Q] = u * c - v * s;
            t[d + Q + x * 2 * Q] = v * c + u * s;
          }
        }
      }
}
