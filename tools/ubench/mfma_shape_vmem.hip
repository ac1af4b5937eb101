// Cost of DMA / LDS-read instructions interleaved into a back-to-back fp32 MFMA stream, for the two fp32 shapes:
//   32x32x2 (16 passes = 64 cycles per instruction) vs 16x16x4 (8 passes = 32 cycles); same FLOP rate.
// 2 waves per SIMD (8 per workgroup, one workgroup per CU); per iteration 4096 cycles of MFMA per wave + NV DMA loads
// spread evenly + ND ds_read_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE, int NV, int ND>
__global__ __launch_bounds__(512, 1) void k(float* buf, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 8 * 256 + 4096];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (iters < 0) lds[threadIdx.x] = 0.f;
  f32x16 acc[4];
  f32x4 acs[16];
  for (int g = 0; g < 4; ++g)
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  for (int g = 0; g < 16; ++g) acs[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  const float* gp = buf + ((long)blockIdx.x * 512 + threadIdx.x) * 4;
  float* lp = lds + wave * 8 * 256;
  f32x4 sink = {0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7fffffff, 0x00027000);
  const int voff = (int)(((long)blockIdx.x * 512 + threadIdx.x) * 16);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (SHAPE == 32 || SHAPE == 33) {
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
      } else {
#pragma unroll
        for (int g = 0; g < 8; ++g) acs[(g + 8 * (i & 1))] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acs[(g + 8 * (i & 1))], 0, 0, 0);
      }
      if (i < NV) {
        if (SHAPE == 33)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lp + (i & 7) * 256), 16, voff + (i & 7) * 32768, 0, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((gptr_t)(gp + (i & 7) * 8192), (lptr_t)(lp + (i & 7) * 256), 16, 0, 0);
      }
      if (i < ND) asm volatile("ds_read_b128 %0, %1" : "=v"(sink) : "v"((int)(size_t)(lptr_t)(lds + 16384 + lane * 4)));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  }
  float s = sink[0];
  for (int g = 0; g < 4; ++g) s += acc[g][0];
  for (int g = 0; g < 16; ++g) s += acs[g][0];
  if (s == 12345.678f) buf[0] = s;
}

template <int SHAPE, int NV, int ND>
void run(float* buf) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<SHAPE, NV, ND>), dim3(256), dim3(512), 0, 0, buf, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float m;
    hipEventElapsedTime(&m, e0, e1);
    if (m < ms) ms = m;
  }
  const double ideal_us = iters * 4096.0 * 2 / 2400.0;
  printf("mfma %2dx%2d  %2d DMA + %2d ds_read per 4096 MFMA-cycles: %.3f of ideal time  (+%.0f cycles per SIMD and iteration)\n", SHAPE, SHAPE, NV, ND,
         ms * 1e3 / ideal_us, (ms * 1e3 - ideal_us) * 2400.0 / iters);
}

int main() {
  float* buf;
  hipMalloc(&buf, 256 * 512 * 16 + (1 << 20));
  hipMemset(buf, 0, 256 * 512 * 16 + (1 << 20));
  run<32, 0, 0>(buf);
  run<16, 0, 0>(buf);
  run<32, 8, 0>(buf);
  run<16, 8, 0>(buf);
  run<32, 0, 16>(buf);
  run<16, 0, 16>(buf);
  run<32, 8, 16>(buf);
  run<16, 8, 16>(buf);
  run<33, 8, 0>(buf);
  run<33, 8, 16>(buf);
  return 0;
}
