// Checks the gfx950 lane-swap 4x4 transpose used by lstm.hip (gate_transpose): element (r, g) starts in register r of
// lane group g (16 lanes each) and must end in register g of lane group r.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/permlane_transpose.hip -o tools/ubench/build/permlane_transpose
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  const unsigned l = threadIdx.x, g = l >> 4, u = l & 15;
  unsigned a = 0 * 1000 + g * 100 + u, b = 1 * 1000 + g * 100 + u, c = 2 * 1000 + g * 100 + u, d = 3 * 1000 + g * 100 + u;
  u32x2 t;
  t = __builtin_amdgcn_permlane32_swap(a, c, false, false), a = t[0], c = t[1];
  t = __builtin_amdgcn_permlane32_swap(b, d, false, false), b = t[0], d = t[1];
  t = __builtin_amdgcn_permlane16_swap(a, b, false, false), a = t[0], b = t[1];
  t = __builtin_amdgcn_permlane16_swap(c, d, false, false), c = t[0], d = t[1];
  o[l * 4 + 0] = a, o[l * 4 + 1] = b, o[l * 4 + 2] = c, o[l * 4 + 3] = d;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (unsigned l = 0; l < 64; ++l)
    for (unsigned q = 0; q < 4; ++q) ok &= h[l * 4 + q] == (l >> 4) * 1000 + q * 100 + (l & 15);
  printf("lane-swap 4x4 transpose: %s (lane 17: %u %u %u %u)\n", ok ? "CONFIRMED" : "WRONG", h[68], h[69], h[70], h[71]);
  return !ok;
}
