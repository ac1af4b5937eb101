// Issue rate of v_mfma_f32_4x4x1_16B_f32 as the recurrence uses it: a dependent accumulator chain with the A-broadcast
// (cbsz = 4, abid = k & 15), 1 / 2 / 4 waves per SIMD, one or two accumulators.  Wall time (HIP events) over all CUs,
// reported as core clocks per MFMA and SIMD at the clock rocm-smi reports under load (2.4 GHz nominal).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma4x4_rate.hip -o tools/ubench/build/mfma4x4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool BCAST>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  float w[100];
#pragma unroll
  for (int i = 0; i < 100; ++i) w[i] = threadIdx.x * 0.001f + i;
  float hv[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) hv[q] = threadIdx.x * 0.01f + q;
  f32x4 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) acc[a] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 100; ++kk) {
      if (BCAST) {
        switch (kk & 15) {
#define C(B) case B: acc[kk % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(hv[kk >> 4], w[kk], acc[kk % NACC], 4, B, 0); break;
          C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
        }
      } else {
        acc[kk % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(hv[kk >> 4], w[kk], acc[kk % NACC], 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) hv[q] += acc[0][q & 3] * 1e-30f;   // carry a dependence into the next iteration
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int a = 1; a < NACC; ++a) s += acc[a];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC, bool BCAST>
static void run(int waves, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, BCAST>), dim3(256), dim3(64 * waves), 0, 0, d, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, BCAST>), dim3(256), dim3(64 * waves), 0, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (waves + 3) / 4;   // waves on the fullest SIMD
  printf("acc %d bcast %d waves/WG %2d: %.3f ms  -> %.2f clocks per MFMA and SIMD at 2.4 GHz (%.2f per wave)\n", NACC, (int)BCAST, waves, ms,
         ms * 1e-3 * 2.4e9 / (iters * 100.0 * per_simd), ms * 1e-3 * 2.4e9 / (iters * 100.0));
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  for (int waves : {4, 8, 16}) {
    run<1, true>(waves, d);
    run<2, true>(waves, d);
    run<4, true>(waves, d);
    run<1, false>(waves, d);
    run<2, false>(waves, d);
  }
  return 0;
}
