// What does a wave64 VALU instruction cost when part of the wave is masked off?
// One wave per SIMD runs a chain of independent v_fma_f32 (8 accumulators, no dependency stalls) with EXEC limited to the
// first N lanes (N = 64, 48, 32, 16, 1): cycles per instruction from s_memtime.  If the hardware skips the 16-lane passes
// whose EXEC bits are all zero, a 30-wide CTC beam (beam.hip: lane = beam slot) pays for two passes instead of four.
// A second wave of full-width MFMAs on the same SIMD shows what a masked VALU wave takes away from the matrix pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void valu_chain(float* out, long long* cyc, int n_active, int iters) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  const float m = 1.0001f, c = 0.0003f;
  long long t0 = 0, t1 = 0;
  if ((int)threadIdx.x < n_active) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    }
    t1 = __builtin_readcyclecounter();
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// wave 0: masked VALU chain; wave 1..: fp32 MFMA chain; both timed.  256 threads = one wave per SIMD: the VALU wave and the
// MFMA wave of interest share SIMD 0 only when the block has 5+ waves, so launch 512 threads: waves 0 and 4 share a SIMD.
__global__ __launch_bounds__(512) void valu_next_to_mfma(float* out, long long* cyc, int n_active, int iters, int valu_on) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = lane * 0.001f + i;
  const float m = 1.0001f, c = 0.0003f;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (valu_on && lane < n_active) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[i + 4], acc[i], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  float* d;
  long long* c;
  hipMalloc(&d, 512 * 4 * 1024);
  hipMalloc(&c, 8 * 8 * 1024);
  const int iters = 2000;
  const int ns[] = {64, 48, 33, 32, 17, 16, 1};
  printf("-- one wave alone: %d x 64 independent v_fma_f32\n", iters);
  for (int n : ns) {
    hipLaunchKernelGGL(valu_chain, dim3(1), dim3(64), 0, 0, d, c, n, iters);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("active lanes %2d: %.2f cycles (s_memtime units of 100 MHz -> x clock ratio) per instruction, raw %lld\n", n, (double)h / (iters * 64.0), h);
  }
  printf("-- next to an fp32 MFMA wave on the same SIMD (8 MFMA 16x16x4 per VALU-wave 64 fma)\n");
  for (int on = 0; on < 2; ++on)
    for (int n : ns) {
      if (!on && n != 64) continue;
      hipLaunchKernelGGL(valu_next_to_mfma, dim3(1), dim3(512), 0, 0, d, c, n, iters, on);
      hipDeviceSynchronize();
      long long h[8];
      hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
      printf("valu %s active %2d: valu wave0 %lld  mfma wave4 %lld wave5 %lld\n", on ? "on " : "off", n, h[0], h[4], h[5]);
    }
  return 0;
}
