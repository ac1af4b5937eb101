// How fast can ONE wave issue LDS-DMA loads?  cycles per global_load_lds_dwordx4 when 16 are issued back to back and then
// waited for, for (a) a fresh M0 per load, (b) one M0 and growing immediate offsets, (c) like (a) with dword-sized loads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int MODE>
__global__ __launch_bounds__(64) void k(const float* buf, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 256 + 4096];
  const int lane = threadIdx.x;
  if (iters < 0) lds[lane] = 0.f;
  const float* gp = buf + ((long)blockIdx.x * 64 + lane) * 4;
  // GEMM-like: 8 rows of 128 bytes per instruction, row stride 1 KB (activations) or 3 KB (weights, shared by all waves)
  const float* gr = buf + (long)blockIdx.x * 128 * 256 + (lane >> 3) * 256 + (lane & 7) * 4;
  const float* gw = buf + (lane >> 3) * 768 + (lane & 7) * 4;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (MODE == 0) __builtin_amdgcn_global_load_lds((gptr_t)(gp + q * 4096), (lptr_t)(lds + q * 256), 16, 0, 0);
      if (MODE == 1) {
        const float* g4 = gp + (q >> 2) * 4096;
        float* l4 = lds + (q >> 2) * 1024;
        switch (q & 3) {
          case 0: __builtin_amdgcn_global_load_lds((gptr_t)g4, (lptr_t)l4, 16, 0, 0); break;
          case 1: __builtin_amdgcn_global_load_lds((gptr_t)g4, (lptr_t)l4, 16, 1024, 0); break;
          case 2: __builtin_amdgcn_global_load_lds((gptr_t)g4, (lptr_t)l4, 16, 2048, 0); break;
          default: __builtin_amdgcn_global_load_lds((gptr_t)g4, (lptr_t)l4, 16, 3072, 0); break;
        }
      }
      if (MODE == 2) __builtin_amdgcn_global_load_lds((gptr_t)(gp + q * 4096), (lptr_t)(lds + q * 256), 4, 0, 0);
      if (MODE == 3) __builtin_amdgcn_global_load_lds((gptr_t)(gr + q * 8 * 256), (lptr_t)(lds + q * 256), 16, 0, 0);
      if (MODE == 4) __builtin_amdgcn_global_load_lds((gptr_t)(gw + q * 8 * 768), (lptr_t)(lds + q * 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const float* buf, long long* cyc, int blocks) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, buf, cyc, iters);
  hipDeviceSynchronize();
  static long long h[512];
  hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < blocks; ++i) avg += h[i];
  printf("%-44s %3d waves on the chip: %.0f cycles per DMA instruction (incl. the wait after each 16)\n", name, blocks, avg / blocks / iters / 16);
}

int main() {
  float* buf;
  long long* cyc;
  hipMalloc(&buf, 64 << 20);
  hipMalloc(&cyc, 512 * 8);
  hipMemset(buf, 0, 64 << 20);
  for (int blocks : {1, 256, 512}) {
    run<0>("dwordx4, new M0 per load", buf, cyc, blocks);
    run<1>("dwordx4, one M0 per 4 loads (imm offsets)", buf, cyc, blocks);
    run<2>("dword,   new M0 per load", buf, cyc, blocks);
    run<3>("dwordx4, 8 rows x 128 B, 1 KB stride", buf, cyc, blocks);
    run<4>("dwordx4, 8 rows x 128 B, 3 KB stride, shared", buf, cyc, blocks);
  }
  return 0;
}
