// Probe: A-operand broadcast of v_mfma_f32_4x4x1_16B_f32 (CBSZ / ABID modifiers).
// Expectation (CDNA ISA): with cbsz = 4 all 16 blocks take their 4x1 A vector from block ABID:
//   D[blk][i][j] += A[abid][i] * B[blk][j]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ABID>
__device__ void one(float a, float b, float* out, int l) {
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 4, ABID, 0);
  for (int r = 0; r < 4; ++r) out[(ABID * 64 + l) * 4 + r] = acc[r];
}
__global__ void k(float* out, long long* cyc) {
  const int l = threadIdx.x;
  float a = 1.0f + (l & 3) + 10.0f * (l >> 2);           // A[block][i] = 1 + i + 10*block
  float b = 100.0f * (1 + (l & 3)) + 0.001f * (l >> 2);  // B[block][j]
  one<0>(a, b, out, l);
  one<1>(a, b, out, l);
  one<5>(a, b, out, l);
  one<15>(a, b, out, l);
  // issue rate with broadcast, 2 accumulators alternating
  f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int i = 0; i < 500; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 3, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 7, 0);
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 11, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 12, 0);
  }
  long long t1 = clock64();
  if (l == 0) cyc[0] = t1 - t0;
  out[16 * 64 * 4 + l] = c0[0] + c1[1];
}
int main() {
  float* d;
  long long* c;
  hipMalloc(&d, (16 * 64 * 4 + 64) * 4);
  hipMalloc(&c, 16);
  hipMemset(d, 0, (16 * 64 * 4 + 64) * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c);
  static float h[16 * 64 * 4];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  long long hc;
  hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
  for (int abid : {0, 1, 5, 15}) {
    int ok = 1;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int blk = l >> 2, j = l & 3;
        const float want = (1.0f + r + 10.0f * abid) * (100.0f * (1 + j) + 0.001f * blk);
        if (fabsf(h[(abid * 64 + l) * 4 + r] - want) > 1e-3f * fabsf(want)) ok = 0;
      }
    printf("cbsz=4 abid=%2d : D[blk][i][j] = A[abid][i]*B[blk][j]  %s   (lane 9: %.3f %.3f %.3f %.3f)\n", abid, ok ? "CONFIRMED" : "NO",
           h[(abid * 64 + 9) * 4], h[(abid * 64 + 9) * 4 + 1], h[(abid * 64 + 9) * 4 + 2], h[(abid * 64 + 9) * 4 + 3]);
  }
  printf("ticks per broadcast MFMA (2 accumulators): %.2f\n", hc / 2000.0);
  return 0;
}
