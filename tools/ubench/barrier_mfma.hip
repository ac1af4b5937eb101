// micro-benchmarks: cost of s_barrier with 8 waves, MFMA 16x16x4 f32 issue interval, with/without partner VALU
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* cyc, int iters) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  float v = a;
  lds[threadIdx.x] = a;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // barrier only
      __syncthreads();
    } else if (MODE == 1) {  // waves 0-3: 100 MFMA; waves 4-7: idle; barrier
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 25; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
      }
      __syncthreads();
    } else if (MODE == 2) {  // waves 0-3: 100 MFMA; waves 4-7: 200 VALU fma + 30 exp
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 25; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 200; ++i) v = fmaf(v, 1.0001f, 0.5f);
#pragma unroll
        for (int i = 0; i < 30; ++i) v = __expf(v * 1e-3f);
      }
      __syncthreads();
    } else if (MODE == 3) {  // no barrier: waves 0-3 100 MFMA
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 25; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
      }
    } else if (MODE == 4) {  // ping-pong alternate roles each iteration, barrier
      if ((wave < 4) == ((it & 1) == 0)) {
#pragma unroll
        for (int i = 0; i < 25; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 200; ++i) v = fmaf(v, 1.0001f, 0.5f);
#pragma unroll
        for (int i = 0; i < 30; ++i) v = __expf(v * 1e-3f);
      }
      __syncthreads();
    } else if (MODE == 6) {  // VALU-only partner (no MFMA anywhere), barrier
      if (wave >= 4) {
#pragma unroll
        for (int i = 0; i < 200; ++i) v = fmaf(v, 1.0001f, 0.5f);
#pragma unroll
        for (int i = 0; i < 30; ++i) v = __expf(v * 1e-3f);
      }
      __syncthreads();
    } else if (MODE == 7) {  // MFMA waves + partner with 4 independent VALU chains (same op count)
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 25; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
      } else {
        float v1 = v + 1, v2 = v + 2, v3 = v + 3;
#pragma unroll
        for (int i = 0; i < 50; ++i) { v = fmaf(v, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { v = __expf(v * 1e-3f); v1 = __expf(v1 * 1e-3f); v2 = __expf(v2 * 1e-3f); v3 = __expf(v3 * 1e-3f); }
        v += v1 + v2 + v3;
      }
      __syncthreads();
    } else if (MODE == 8) {  // partner 4 independent chains, no MFMA
      if (wave >= 4) {
        float v1 = v + 1, v2 = v + 2, v3 = v + 3;
#pragma unroll
        for (int i = 0; i < 50; ++i) { v = fmaf(v, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { v = __expf(v * 1e-3f); v1 = __expf(v1 * 1e-3f); v2 = __expf(v2 * 1e-3f); v3 = __expf(v3 * 1e-3f); }
        v += v1 + v2 + v3;
      }
      __syncthreads();
    } else if (MODE == 5) {  // all 8 waves MFMA (2 per SIMD), no barrier
#pragma unroll
      for (int i = 0; i < 25; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + v;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 512 * 512 * 4); hipMalloc(&cyc, 512 * 8);
  const int iters = 2000;
  for (int mode = 0; mode < 9; ++mode) {
    for (int grid : {70}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, out, cyc, iters); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, out, cyc, iters); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, out, cyc, iters); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, out, cyc, iters); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, out, cyc, iters); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(512), 0, 0, out, cyc, iters); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(512), 0, 0, out, cyc, iters); break;
        case 7: hipLaunchKernelGGL(k<7>, dim3(grid), dim3(512), 0, 0, out, cyc, iters); break;
        case 8: hipLaunchKernelGGL(k<8>, dim3(grid), dim3(512), 0, 0, out, cyc, iters); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("mode %d grid %3d: %.1f us/iter-1000 = %.3f us per iter, clock64 %.0f ticks/iter\n", mode, grid, ms * 1000 / iters * 1000, ms * 1000.0 / iters, (double)c / iters);
    }
  }
  return 0;
}
