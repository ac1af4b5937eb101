// res_layer1 / branch2 up to conv2b as ONE memory-bound pass for gfx950 (population BN).
//
// The first residual block sees a ONE-channel input (cnn.py:380-389 DNA_model1, :234-262 residual_layer): conv2a is
// a 1x1 convolution of the raw signal, so with BN folded its output is a1[pos][c] = relu(s[pos]*a[c] + b[c]) -- every
// one of the 256 channels is a function of the single scalar s[pos].  conv2b (1 x k over a1, BN, ReLU) is therefore
//     y[pos][n] = relu( sh[n] + sum_tap f[tap][n]( s[pos*stride + tap - left] ) ),
//     f[tap][n](s) = sum_c W2b'[tap][c][n] * relu(s*a[c] + b[c]),
// and each f is PIECEWISE LINEAR in s with at most C breakpoints s_c = -b[c]/a[c] (the same for every tap and n).
// The engine tabulates, on the host and in float64, the slope alpha and the value f(ref) at a reference point ref of
// every interval between consecutive breakpoints (f = alpha*(s - ref) + f(ref), ref = the interval's point nearest to 0: |s - ref| <= |s|, no cancellation); this kernel finds the interval of each sample (binary search in LDS) and evaluates k taps
// with one 8-byte table read each -- instead of materialising a1 (450 MB) and running a K = k*256 GEMM over it
// (173 GFLOP per batch for DNA_default).  Same function, re-associated: the table sums in float64, the result differs
// from the GEMM form by fp32 rounding only (tests/test_gpu_parity.py holds the 1e-4 logit bound against the oracle).
#include "kernels.h"

namespace chiron {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int PWL_TP = 32;        // output positions per workgroup
constexpr int PWL_MAX_BP = 512;   // breakpoints (= channels of conv2a) held in LDS
constexpr int PWL_MAX_S = 512;    // samples a workgroup looks at: (PWL_TP - 1) * stride + k

template <int FMT>
__global__ __launch_bounds__(256) void pwl_conv_kernel(const PwlConvParams p) {
  __builtin_amdgcn_s_setprio(3);   // a short kernel between the other batches' persistent GEMM waves
  __shared__ float bps[PWL_MAX_BP];
  __shared__ float refs[PWL_MAX_BP + 1];
  __shared__ float ss[PWL_MAX_S];
  __shared__ int ks[PWL_MAX_S];
  const int tid = threadIdx.x;
  const int tiles_per_row = (p.T_out + PWL_TP - 1) / PWL_TP;
  const int b = blockIdx.x / tiles_per_row;
  const int t0 = (blockIdx.x - b * tiles_per_row) * PWL_TP;
  const int np = min(PWL_TP, p.T_out - t0);
  for (int i = tid; i < p.nbp; i += 256) bps[i] = p.bp[i];
  for (int i = tid; i <= p.nbp; i += 256) refs[i] = p.ref[i];
  __syncthreads();
  const int nsamp = (np - 1) * p.stride + p.k;
  for (int i = tid; i < nsamp; i += 256) {
    const int idx = t0 * p.stride - p.left + i;
    const bool valid = idx >= 0 && idx < p.L;  // SAME padding pads a1 with zeros: such taps contribute nothing
    const float s = valid ? p.sig[(long)b * p.L + idx] : 0.f;
    int lo = 0, hi = p.nbp;  // interval = number of breakpoints below s
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (bps[mid] < s) lo = mid + 1; else hi = mid;
    }
    // the table holds f at the interval's reference point (the point of the interval nearest to 0, engine.hip)
    ss[i] = s - refs[lo];
    ks[i] = valid ? lo : -1;
  }
  __syncthreads();
  // thread = (position mod 4, four adjacent channels): two 16-byte table reads cover (alpha, beta) of four channels
  const int C = p.C, k = p.k;
  const int par = tid >> 6, t4 = tid & 63;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  for (int n = 4 * t4; n < C; n += 256) {
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n);
    const float* tab = reinterpret_cast<const float*>(p.tab + n);
    for (int pp = par; pp < np; pp += 4) {
      f32x4 acc = sh;
      for (int tap = 0; tap < k; ++tap) {
        const int i = pp * p.stride + tap;
        const int kk = ks[i];
        if (kk >= 0) {
          const float* q = tab + ((long)kk * k + tap) * C * 2;
          const f32x4 ab0 = *reinterpret_cast<const f32x4*>(q), ab1 = *reinterpret_cast<const f32x4*>(q + 4);
          const float sv = ss[i];
          acc[0] += fmaf(ab0[0], sv, ab0[1]);
          acc[1] += fmaf(ab0[2], sv, ab0[3]);
          acc[2] += fmaf(ab1[0], sv, ab1[1]);
          acc[3] += fmaf(ab1[2], sv, ab1[3]);
        }
      }
      f32x4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = fmaxf(acc[j], 0.f);
      const long pos = (long)b * p.T_out + t0 + pp;
      if (FMT == 0) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + pos * C + n) = y;
      } else {
        f16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hi[j] = (_Float16)y[j];
          lo[j] = (_Float16)(y[j] - (float)hi[j]);
        }
        if (FMT == 1) {
          *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(p.out) + pos * C + n) = hi;
        } else {  // split: per 32-element block 32 hi halves, then 32 lo halves
          _Float16* o = reinterpret_cast<_Float16*>(p.out) + (pos * C + (n >> 5) * 32) * 2 + (n & 31);
          *reinterpret_cast<f16x4*>(o) = hi;
          *reinterpret_cast<f16x4*>(o + 32) = lo;
        }
      }
    }
  }
}

bool launch_pwl_conv(const PwlConvParams& p, hipStream_t stream) {
  if (p.nbp > PWL_MAX_BP || (PWL_TP - 1) * p.stride + p.k > PWL_MAX_S || (p.C & 3)) return false;
  const int tiles_per_row = (p.T_out + PWL_TP - 1) / PWL_TP;
  const dim3 grid(p.B * tiles_per_row), block(256);
  if (p.fmt == 0)
    hipLaunchKernelGGL(pwl_conv_kernel<0>, grid, block, 0, stream, p);
  else if (p.fmt == 1)
    hipLaunchKernelGGL(pwl_conv_kernel<1>, grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL(pwl_conv_kernel<2>, grid, block, 0, stream, p);
  return true;
}

}  // namespace chiron
