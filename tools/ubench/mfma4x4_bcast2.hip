// Probe: v_mfma_f32_4x4x1_16B_f32 with cbsz = 2 (A broadcast inside groups of four blocks).  Expectation:
//   D[blk][i][j] += A[(blk & ~3) + abid][i] * B[blk][j],  abid = 0..3
// (used by the "light" wave of the recurrence, which spreads K over the four block groups).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma4x4_bcast2.hip -o tools/ubench/build/mfma4x4_bcast2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int ABID>
__device__ void one(float a, float b, float* out, int l) {
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 2, ABID, 0);
  for (int r = 0; r < 4; ++r) out[(ABID * 64 + l) * 4 + r] = acc[r];
}
__global__ void k(float* out) {
  const int l = threadIdx.x;
  const float a = 1.0f + (l & 3) + 10.0f * (l >> 2);           // A[block][i] = 1 + i + 10*block
  const float b = 100.0f * (1 + (l & 3)) + 0.001f * (l >> 2);  // B[block][j]
  one<0>(a, b, out, l);
  one<1>(a, b, out, l);
  one<2>(a, b, out, l);
  one<3>(a, b, out, l);
}
int main() {
  float* d;
  (void)hipMalloc(&d, 4 * 64 * 4 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  static float h[4 * 64 * 4];
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int abid = 0; abid < 4; ++abid) {
    int ok = 1;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int blk = l >> 2, j = l & 3, src = (blk & ~3) + abid;
        const float want = (1.0f + r + 10.0f * src) * (100.0f * (1 + j) + 0.001f * blk);
        if (fabsf(h[(abid * 64 + l) * 4 + r] - want) > 1e-3f * fabsf(want)) ok = 0;
      }
    printf("cbsz=2 abid=%d: D[blk][i][j] = A[(blk&~3)+abid][i]*B[blk][j] %s\n", abid, ok ? "CONFIRMED" : "NO");
  }
  return 0;
}
