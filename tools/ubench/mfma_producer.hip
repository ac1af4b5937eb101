// Does a wave that ONLY issues LDS-DMA loads slow down a partner wave on the same SIMD that ONLY issues
// v_mfma_f32_32x32x2_f32?  (Would a dedicated producer wave make the GEMM's DMA issue free?)
// 8 waves per workgroup, one workgroup per CU: waves 0-3 run the MFMA stream (timed with clock64), waves 4-7 issue
// ND global_load_lds_dwordx4 per "chunk" (64 partner MFMAs = 4096 cycles) in mode 1, or spin on s_sleep in mode 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int ND>
__global__ __launch_bounds__(512, 1) void k(float* buf, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 1024];
  __shared__ volatile int done;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (iters < 0) lds[threadIdx.x] = 0.f;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  if (wave < 4) {
    f32x16 acc[4];
    for (int g = 0; g < 4; ++g)
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0;
    for (int g = 0; g < 4; ++g) s += acc[g][0];
    if (s == 12345.678f) buf[0] = s;
    if (lane == 0) {
      cyc[blockIdx.x * 4 + wave] = t1 - t0;
      atomicAdd((int*)&done, 1);
    }
  } else {
    const float* gp = buf + ((long)blockIdx.x * 512 + threadIdx.x) * 4;
    float* lp = lds + (wave - 4) * 1024;
    // issue ND DMAs, then idle for the rest of a 4096-cycle chunk, until the MFMA waves are done
    while (done < 4) {
      const long long c0 = clock64();
      if (ND > 0) {
#pragma unroll
        for (int i = 0; i < ND; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(gp + (i & 3) * 64 * 1024), (lptr_t)lp, 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)");
      }
      while (clock64() - c0 < 4096) __builtin_amdgcn_s_sleep(2);
    }
  }
}

template <int ND>
void run(float* buf, long long* cyc) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<ND>), dim3(256), dim3(512), 0, 0, buf, cyc, iters);
  hipDeviceSynchronize();
  static long long h[1024];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < 1024; ++i) avg += h[i];
  avg /= 1024;
  printf("partner wave issues %2d DMA per 4096 cycles: MFMA wave takes %.0f cycles per 64 MFMAs (ideal 4096) -> %.1f %% of peak\n", ND,
         avg / iters, 100.0 * 4096 * iters / avg);
}

int main() {
  float* buf;
  long long* cyc;
  hipMalloc(&buf, 256 * 512 * 16 + (4 << 20));
  hipMalloc(&cyc, 1024 * 8);
  hipMemset(buf, 0, 256 * 512 * 16 + (4 << 20));
  run<0>(buf, cyc);
  run<8>(buf, cyc);
  run<16>(buf, cyc);
  run<32>(buf, cyc);
  return 0;
}
