// Where do the 7 waves of two co-resident 448-thread workgroups (the recurrence's shape: 128 VGPRs, 2 per CU) land?
// Prints, for a few CUs, the SIMD id of every (workgroup, wave).  HW_ID: wave_id[3:0] simd_id[5:4] cu_id[11:8] sh_id[12] se_id[15:13]
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/wave_placement.hip -o tools/ubench/build/wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <string>
#include <vector>
__global__ __launch_bounds__(448, 4) void k(unsigned* out, int spin) {
  __shared__ float lds[512];
  float r[100];
  for (int i = 0; i < 100; ++i) r[i] = threadIdx.x + i;  // hold ~100 VGPRs like W_hh
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float acc = 0;
  for (int it = 0; it < spin; ++it) {
#pragma unroll
    for (int i = 0; i < 100; ++i) acc = fmaf(r[i], acc, 1.0f);
    lds[threadIdx.x] = acc;
    __syncthreads();
    acc += lds[(threadIdx.x + 1) % 448];
  }
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 7 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 7 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
  if (acc == 12345.f) out[0] = 0;
}
int main() {
  const int nwg = 512;
  unsigned* d;
  (void)hipMalloc(&d, nwg * 7 * 8);
  hipLaunchKernelGGL(k, dim3(nwg), dim3(448), 0, 0, d, 2000);
  std::vector<unsigned> h(nwg * 7 * 2);
  (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cus;  // physical CU -> workgroups
  for (int w = 0; w < nwg; ++w) {
    const unsigned hw = h[w * 14], xcc = h[w * 14 + 1] & 0xf;
    const unsigned cu = (xcc << 16) | (hw & 0xff00);
    cus[cu].push_back(w);
  }
  int shown = 0, hist[8] = {0};
  for (auto& kv : cus) {
    int load[4] = {0, 0, 0, 0};
    for (int w : kv.second)
      for (int i = 0; i < 7; ++i) ++load[(h[(w * 7 + i) * 2] >> 4) & 3];
    int mx = 0;
    for (int s = 0; s < 4; ++s) mx = load[s] > mx ? load[s] : mx;
    ++hist[mx < 8 ? mx : 7];
    if (shown < 6) {
      printf("cu %06x:", kv.first);
      for (int w : kv.second) {
        printf("  wg %3d simd", w);
        for (int i = 0; i < 7; ++i) printf(" %u", (h[(w * 7 + i) * 2] >> 4) & 3);
        printf(" slot");
        for (int i = 0; i < 7; ++i) printf(" %u", h[(w * 7 + i) * 2] & 15);
      }
      printf("   loads %d %d %d %d\n", load[0], load[1], load[2], load[3]);
      ++shown;
    }
  }
  // histogram of the SIMD sequences (wave 0..6) of the workgroups sharing a CU
  std::map<std::string, int> pat;
  for (auto& kv : cus) {
    std::string key;
    for (int w : kv.second) {
      for (int i = 0; i < 7; ++i) key += char('0' + ((h[(w * 7 + i) * 2] >> 4) & 3));
      key += '|';
    }
    ++pat[key];
  }
  for (auto& kv : pat) printf("pattern %s : %d CUs\n", kv.first.c_str(), kv.second);
  // the recurrence's role rule (lstm.hip): light = first wave of the workgroup whose slot is 3, else wave 6
  std::map<std::string, int> heavy;
  for (auto& kv : cus) {
    int hl[4] = {0, 0, 0, 0}, ll[4] = {0, 0, 0, 0};
    for (int w : kv.second) {
      int lw = 6;
      for (int i = 6; i >= 0; --i)
        if ((h[(w * 7 + i) * 2] & 15) == 3) lw = i;
      for (int i = 0; i < 7; ++i) ++(i == lw ? ll : hl)[(h[(w * 7 + i) * 2] >> 4) & 3];
    }
    char key[64];
    snprintf(key, sizeof key, "heavy %d %d %d %d light %d %d %d %d", hl[0], hl[1], hl[2], hl[3], ll[0], ll[1], ll[2], ll[3]);
    ++heavy[key];
  }
  for (auto& kv : heavy) printf("roles: %s : %d CUs\n", kv.first.c_str(), kv.second);
  printf("%zu CUs used; CUs by waves on their fullest SIMD:", cus.size());
  for (int i = 0; i < 8; ++i)
    if (hist[i]) printf("  %d waves: %d", i, hist[i]);
  printf("\n");
  return 0;
}
