// layout probe for v_mfma_f32_4x4x1_16b_f32: D[block][i][j] += A[block][i] * B[block][j]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, long long* cyc) {
  const int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  float a = 1.0f + (l & 3) + 10.0f * (l >> 2);   // A[block][i] = 1 + i + 10*block
  float b = 100.0f * (1 + (l & 3)) + 0.001f * (l >> 2);  // B[block][j]
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
  // issue-rate probe: 2 independent accumulators, 2000 MFMAs
  f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0,0,0,0}, c3 = {0,0,0,0};
  long long t0 = clock64();
  for (int i = 0; i < 500; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, b, c3, 0, 0, 0);
  }
  long long t1 = clock64();
  f32x4 d0 = {0, 0, 0, 0};
  long long t2 = clock64();
  for (int i = 0; i < 2000; ++i) d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
  long long t3 = clock64();
  if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
  out[256 + l] = c0[0] + c1[1] + c2[2] + c3[3] + d0[0];
}
int main() {
  float* d; long long* c; hipMalloc(&d, 1024 * 4); hipMalloc(&c, 16);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 4, 5, 63}) {
    printf("lane %2d:", l);
    for (int r = 0; r < 4; ++r) printf(" %.3f", h[l * 4 + r]);
    printf("\n");
  }
  int ok = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int blk = l >> 2, j = l & 3;
    float want = (1.0f + r + 10.0f * blk) * (100.0f * (1 + j) + 0.001f * blk);
    if (fabsf(h[l * 4 + r] - want) > 1e-3f * fabsf(want)) ok = 0;
  }
  printf("layout D[lane=4*blk+j][reg=i]: %s\n", ok ? "CONFIRMED" : "NO");
  printf("ticks per MFMA: 4 independent accs %.2f, dependent chain %.2f\n", hc[0] / 2000.0, hc[1] / 2000.0);
  return 0;
}
