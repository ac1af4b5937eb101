// Times the conv-shaped launches of gemm.hip in isolation (conv2a: K = 256, conv2b: 3 taps, K = 768,
// M = 1100*400, N = 256).  The kernel source is textually included so that experimental variants can be
// compiled with -DGEMM_SRC=\"...\":
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I chiron_amd/csrc -DGEMM_SRC='"../../chiron_amd/csrc/gemm.hip"' \
//         tools/ubench/gemm_probe.hip -o tools/ubench/build/gemm_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifndef GEMM_SRC
#define GEMM_SRC "../../chiron_amd/csrc/gemm.hip"
#endif
#include GEMM_SRC

using namespace chiron;

int main(int argc, char** argv) {
  const int B = getenv("PROBE_B") ? atoi(getenv("PROBE_B")) : 1100, T = 400, C = 256;
  const long M = (long)B * T;
  const int LD = C + (getenv("PROBE_LDPAD") ? atoi(getenv("PROBE_LDPAD")) : 0);  // row stride of the activations
  float *act, *out, *wt, *shift, *zero;
  hipMalloc(&act, M * LD * 4);
  hipMalloc(&out, M * LD * 4);
  hipMalloc(&wt, 2048 * 256 * 4);
  hipMalloc(&shift, 256 * 4);
  hipMalloc(&zero, 4096);
  hipMemset(zero, 0, 4096);
  std::vector<float> h(M * LD);
  unsigned s = 12345;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
  }
  hipMemcpy(act, h.data(), M * LD * 4, hipMemcpyHostToDevice);
  hipMemcpy(wt, h.data(), 2048 * 256 * 4, hipMemcpyHostToDevice);
  hipMemcpy(shift, h.data(), 256 * 4, hipMemcpyHostToDevice);

  for (int ntap = 1; ntap <= (getenv("PROBE_MAXTAP") ? atoi(getenv("PROBE_MAXTAP")) : 3); ntap += (getenv("PROBE_MAXTAP") ? 1 : 2)) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.B = B;
    g.BP = (B + 3) / 4 * 4;
    g.N = C;
    g.K = ntap * C;
    g.Wt = wt;
    g.shift = shift;
    g.z_dirs_total = 2;
   
    g.M = (int)M;
    g.T_out = T;
    g.nseg = ntap;
    for (int j = 0; j < ntap; ++j) g.seg[j] = GemmSeg{act, LD, 0, C, C, T, 1, getenv("PROBE_SHIFT0") ? 0 : j - ntap / 2, 0};  // PROBE_SHIFT0: every tap reads the same rows
    g.relu = 1;
    g.out = out;
    g.ldo = LD;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_gemm(g, 0);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch_gemm(g, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    double sum = 0;
    hipMemcpy(h.data(), out, 1 << 20, hipMemcpyDeviceToHost);
    for (int i = 0; i < (1 << 18); ++i) sum += h[i];
    printf("%s K=%d: %.4f ms  %.1f TFLOP/s  (checksum %.6e)\n", argc > 1 ? argv[1] : "", g.K, ms, 2.0 * M * C * g.K / ms * 1e-9, sum);
  }
  // ---- LSTM x-projection: A = time-major [T][BP][200] (K = 200, padded 224), N = 800, z layout of lstm.hip
  if (getenv("PROBE_ZOUT")) {
    const int BP = (B + 3) / 4 * 4, H2 = 200, N = 800, Kp = 224;
    float *la, *w2, *z, *sh2;
    int* seq;
    hipMalloc(&la, (long)T * BP * H2 * 4);
    hipMalloc(&w2, (long)896 * Kp * 4);
    hipMalloc(&z, (long)T * BP * N * 4);
    hipMalloc(&sh2, 1024 * 4);
    hipMalloc(&seq, BP * 4);
    hipMemcpy(la, h.data(), (long)T * BP * H2 * 4, hipMemcpyHostToDevice);
    hipMemset(w2, 0, (long)896 * Kp * 4);
    hipMemcpy(w2, h.data(), (long)N * Kp * 4, hipMemcpyHostToDevice);
    hipMemset(sh2, 0, 1024 * 4);
    {
      std::vector<int> hs(BP, T);
      hipMemcpy(seq, hs.data(), BP * 4, hipMemcpyHostToDevice);
    }
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.B = B; g.BP = BP; g.N = N; g.K = Kp; g.Wt = w2; g.shift = sh2; g.z_dirs_total = 2;
    g.M = T * BP; g.T_out = T; g.m_time_major = 1; g.nseg = 1;
    g.seg[0] = GemmSeg{la, H2, 0, H2, Kp, T, 1, 0, 1};
    g.z_seq_len = seq;
    g.out = z; g.out_mode = 1; g.z_cols = 400; g.z_ndir = 2; g.z_dir0 = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_gemm(g, 0);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch_gemm(g, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%s zproj K=200 N=800: %.4f ms  %.1f TFLOP/s\n", argc > 1 ? argv[1] : "", ms, 2.0 * T * B * N * H2 / ms * 1e-9);
  }
  return 0;
}