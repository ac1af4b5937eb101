// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal INPUTS (needed for the hi/lo split of fp32 operands)?
// A = subnormal half 2^-20 in k = 0, B = 1.0 -> D should be 2^-20 if subnormals are kept, 0 if flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  if (threadIdx.x < 32) { a[0] = (_Float16)9.5367431640625e-07f; b[0] = (_Float16)1.0f; }   // 2^-20 : subnormal in f16
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
  // subnormal * subnormal-scale product: 2^-20 * 2^-14 = 2^-34 (fits fp32)
  if (threadIdx.x < 32) b[0] = (_Float16)6.103515625e-05f;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[2] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 16);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  printf("a = %.9g (f16 subnormal 2^-20)\nmfma(a, 1.0)    = %.9g  (expected 9.53674316e-07 if subnormal inputs are honoured)\nmfma(a, 2^-14)  = %.9g  (expected 5.82076609e-11)\n", h[1], h[0], h[2]);
  return 0;
}
