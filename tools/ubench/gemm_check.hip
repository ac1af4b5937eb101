// Correctness probe for gemm.hip in isolation: conv-shaped (3 taps, SAME padding, ReLU) and z-layout launches at small
// ragged shapes against a plain CPU loop.  usage: gemm_check B T BP
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../chiron_amd/csrc/gemm.hip"

using namespace chiron;

static float frand(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 21, T = argc > 2 ? atoi(argv[2]) : 400, BP = argc > 3 ? atoi(argv[3]) : 64;
  const int C = 256;
  const long M = (long)B * T;
  unsigned s = 7;
  std::vector<float> hA(M * C), hW(896 * 768), hS(1024);
  for (auto& v : hA) v = frand(s);
  for (auto& v : hW) v = frand(s) * 0.1f;
  for (auto& v : hS) v = frand(s);
  float *act, *out, *wt, *shift, *zero, *z;
  hipMalloc(&act, M * C * 4);
  hipMalloc(&out, M * C * 4);
  hipMalloc(&wt, hW.size() * 4);
  hipMalloc(&shift, 4096);
  hipMalloc(&zero, 4096);
  hipMemset(zero, 0, 4096);
  hipMemcpy(act, hA.data(), M * C * 4, hipMemcpyHostToDevice);
  hipMemcpy(wt, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(shift, hS.data(), 4096, hipMemcpyHostToDevice);
  // ---- conv: 3 taps
  {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.B = B; g.BP = BP; g.N = C; g.K = 3 * C; g.Wt = wt; g.shift = shift; g.z_dirs_total = 2;
    g.M = (int)M; g.T_out = T; g.nseg = 3;
    for (int j = 0; j < 3; ++j) g.seg[j] = GemmSeg{act, C, 0, C, C, T, 1, j - 1, 0};
    g.relu = 1; g.out = out; g.ldo = C;
    hipMemset(out, 0xff, M * C * 4);
    launch_gemm(g, 0);
    hipDeviceSynchronize();
    std::vector<float> ho(M * C);
    hipMemcpy(ho.data(), out, M * C * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    long wm = -1;
    for (long m = 0; m < M; m += 1) {
      const int b = m / T, t = m % T;
      for (int n = 0; n < C; n += 37) {
        double acc = hS[n];
        for (int j = 0; j < 3; ++j) {
          const int tt = t + j - 1;
          if (tt < 0 || tt >= T) continue;
          const float* a = &hA[((long)b * T + tt) * C];
          const float* w = &hW[(long)n * 768 + j * C];
          for (int k = 0; k < C; ++k) acc += (double)a[k] * w[k];
        }
        const double ref = acc > 0 ? acc : 0;
        const double e = fabs(ref - ho[m * C + n]);
        if (!(e <= worst)) { worst = e; wm = m * C + n; }
      }
    }
    printf("conv  B=%d T=%d: max err %.3e at m=%ld n=%ld\n", B, T, worst, wm / C, wm % C);
  }
  // ---- z layout: A = time-major [T][BP][200], K = 200 (pad 224), N = 800 (six full tiles + one narrow)
  {
    const int H2 = 200, N = 800, Kp = 224;
    std::vector<float> hL((long)T * BP * H2), hW2((long)N * Kp, 0.f);
    for (auto& v : hL) v = frand(s);
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < H2; ++k) hW2[(long)n * Kp + k] = frand(s) * 0.1f;
    float *la, *w2;
    hipMalloc(&la, hL.size() * 4);
    hipMalloc(&w2, hW2.size() * 4);
    hipMalloc(&z, (long)T * BP * N * 4);
    hipMemcpy(la, hL.data(), hL.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w2, hW2.data(), hW2.size() * 4, hipMemcpyHostToDevice);
    hipMemset(z, 0xff, (long)T * BP * N * 4);
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.B = B; g.BP = BP; g.N = N; g.K = Kp; g.Wt = w2; g.shift = shift; g.z_dirs_total = 2;
    g.M = T * BP; g.T_out = T; g.m_time_major = 1; g.nseg = 1;
    g.seg[0] = GemmSeg{la, H2, 0, H2, Kp, T, 1, 0, 1};
    int* seq;
    hipMalloc(&seq, BP * 4);
    {
      std::vector<int> hs(BP, T);
      hipMemcpy(seq, hs.data(), BP * 4, hipMemcpyHostToDevice);
    }
    g.z_seq_len = seq;  // all rows full length: the backward half is stored at step T-1-t
    g.out = z; g.out_mode = 1; g.z_cols = 400; g.z_ndir = 2; g.z_dir0 = 0;
    launch_gemm(g, 0);
    hipDeviceSynchronize();
    std::vector<float> hz((long)T * BP * N);
    hipMemcpy(hz.data(), z, hz.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    long wt_ = -1, wb = -1, wn = -1;
    for (int t = 0; t < T; t += 3)
      for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; n += 29) {
          double acc = hS[n];
          for (int k = 0; k < H2; ++k) acc += (double)hL[((long)t * BP + b) * H2 + k] * hW2[(long)n * Kp + k];
          const int dir = n / 400, nl = n % 400;
          const int ts = dir == 0 ? t : T - 1 - t;
          const float got = hz[((((long)ts * (BP / 4) + (b >> 2)) * 2 + dir) * 400 + nl) * 4 + (b & 3)];
          const double e = fabs(acc - got);
          if (!(e <= worst)) { worst = e; wt_ = t; wb = b; wn = n; }
        }
    printf("zproj B=%d T=%d BP=%d: max err %.3e at t=%ld b=%ld n=%ld\n", B, T, BP, worst, wt_, wb, wn);
  }
  return 0;
}
