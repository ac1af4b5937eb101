// What does an LDS-DMA instruction (buffer_load_dwordx4 ... lds) cost inside a back-to-back fp32 MFMA stream when its 64 lanes
// gather 16-byte slots of ROWS that lie 1 KB apart (channels-last activations, 256 floats per position), as a function of the
// row piece: 1024 B (fully contiguous), 128 B (8 lanes per row: gemm.hip's chunk of 32 floats), 64 B (4 lanes per row: wino.hip's
// F(4,3) chunk of 16 floats), 32 B (2 lanes per row: a chunk of 8 floats)?  2 waves per SIMD, 8 DMA per 4096 MFMA-cycles and wave.
// The source region per workgroup is 512 KB (L2 resident after the first pass).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef void __attribute__((address_space(3)))* lptr_t;

template <int ROWB, int NV>
__global__ __launch_bounds__(512, 1) void k(float* buf, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 8 * 256 + 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (iters < 0) lds[threadIdx.x] = 0.f;
  f32x16 acc[4];
  for (int g = 0; g < 4; ++g)
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  float* lp = lds + wave * 8 * 256;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7fffffff, 0x00027000);
  // lane -> (row, slot): ROWB / 16 lanes per row, rows 1 KB apart; a wave's 8 pieces per iteration walk along the row (chunks)
  constexpr int LPR = ROWB / 16;                      // lanes per row
  const int row = lane / LPR, slot = lane % LPR;
  const int base = (int)(((long)blockIdx.x * 8 + wave) * 64 * 1024) + row * (ROWB == 1024 ? 1024 : 1024) + slot * 16;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
      if (i < NV) {
        const int off = ROWB == 1024 ? base + (i & 7) * 65536 / 8 : base + ((i * ROWB) & 1023);   // next chunk of the same rows
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lp + (i & 7) * 256), 16, off, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)");
  }
  float s = 0.f;
  for (int g = 0; g < 4; ++g) s += acc[g][0];
  if (s == 12345.678f) buf[0] = s;
}

template <int ROWB, int NV>
void run(float* buf) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<ROWB, NV>), dim3(256), dim3(512), 0, 0, buf, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float m;
    hipEventElapsedTime(&m, e0, e1);
    if (m < ms) ms = m;
  }
  const double ideal_us = iters * 4096.0 * 2 / 2400.0;
  printf("row piece %4d B, %2d DMA per 4096 MFMA-cycles and wave: %.3f of the MFMA-only time (+%.0f cycles per SIMD and iteration = %.0f per DMA instruction)\n", ROWB, NV,
         ms * 1e3 / ideal_us, (ms * 1e3 - ideal_us) * 2400.0 / iters, NV ? (ms * 1e3 - ideal_us) * 2400.0 / iters / (2.0 * NV) : 0.0);
}

int main() {
  float* buf;
  const size_t bytes = (size_t)256 * 8 * 64 * 1024 + (1 << 20);
  hipMalloc(&buf, bytes);
  hipMemset(buf, 0, bytes);
  run<1024, 0>(buf);
  run<1024, 8>(buf);
  run<128, 8>(buf);
  run<64, 8>(buf);
  run<32, 8>(buf);
  run<128, 16>(buf);
  run<64, 16>(buf);
  run<32, 16>(buf);
  return 0;
}
