// Which VALU instruction classes overlap with a stream of fp32 MFMAs issued by ANOTHER wave of the same SIMD?
// Workgroup of 8 waves on one CU: waves 0-3 (one per SIMD) issue NM back-to-back MFMAs on 4 accumulators, waves 4-7
// issue NV instructions of class X on 8 independent registers.  Prints solo and combined times; "sum" means the
// two streams serialise on the SIMD's issue/execute path, "max" means they overlap.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/build/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int X>
__device__ __forceinline__ void valu_block(float (&r)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (X == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
    if (X == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
    if (X == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
    if (X == 3) asm volatile("v_add_u32 %0, %0, %0" : "+v"(r[i]));
    if (X == 4) asm volatile("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
    if (X == 5) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(r[i]));
    if (X == 6) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[i]));
    if (X == 7) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(r[i]));
  }
}

// WHO: 1 = MFMA waves only, 2 = VALU waves only, 3 = both.  SHAPE 0: 4x4x1, 1: 32x32x2
template <int X, int WHO, int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16 big[2] = {};
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = a + i;
  __syncthreads();
  const long long t0 = clock64();
  if (wave < 4) {
    if (WHO & 1)
      for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i & 3], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i], 0, 0, 0);
        }
      }
  } else {
    if (WHO & 2)
      for (int it = 0; it < iters; ++it) {
        valu_block<X>(r);
        valu_block<X>(r);
      }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  float s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  out[threadIdx.x] = s + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + big[0][0] + big[1][1];
}

template <int X, int WHO, int SHAPE>
static double run(float* d, long long* c, int iters) {
  hipLaunchKernelGGL((k<X, WHO, SHAPE>), dim3(1), dim3(512), 0, 0, d, c, iters);
  hipLaunchKernelGGL((k<X, WHO, SHAPE>), dim3(1), dim3(512), 0, 0, d, c, iters);
  long long h = 0;
  (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  return (double)h / iters;
}
template <int X>
static void report(const char* name, float* d, long long* c) {
  const int iters = 2000;
  // per iteration: MFMA wave 16 x 4x4x1 (= 128 pipe cycles) or 2 x 32x32x2 (= 128 pipe cycles); VALU wave 16 instructions
  const double m0 = run<X, 1, 0>(d, c, iters), v = run<X, 2, 0>(d, c, iters), b0 = run<X, 3, 0>(d, c, iters);
  const double m1 = run<X, 1, 1>(d, c, iters), b1 = run<X, 3, 1>(d, c, iters);
  printf("%-14s valu alone %6.1f | 4x4x1: mfma %6.1f both %6.1f (sum %6.1f) | 32x32x2: mfma %6.1f both %6.1f (sum %6.1f)  ticks/iter\n", name, v, m0, b0,
         m0 + v, m1, b1, m1 + v);
}
int main() {
  float* d;
  long long* c;
  (void)hipMalloc(&d, 4096);
  (void)hipMalloc(&c, 64);
  report<0>("v_fma_f32", d, c);
  report<1>("v_exp_f32", d, c);
  report<2>("v_rcp_f32", d, c);
  report<3>("v_add_u32", d, c);
  report<4>("v_mov_dpp", d, c);
  report<5>("v_cndmask", d, c);
  report<6>("v_mul_f32", d, c);
  report<7>("v_cvt_f16_f32", d, c);
  return 0;
}
