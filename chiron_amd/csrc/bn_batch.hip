// Batch-statistics BatchNorm for the HEAD code path of the reference: chiron/cnn.py:166-188 simple_global_bn,
//   mean, variance = tf.nn.moments(inp, [0, 1, 2]);  tf.nn.batch_normalization(inp, mean, variance, scale, offset, 1e-5)
// i.e. per-channel moments over every (row, position) of THIS batch, biased variance.  The shipped checkpoints use
// population statistics (folded into the GEMM weights, engine.hip); this mode cannot be folded because the
// statistics exist only after the convolution has run: raw GEMM -> bn_stats -> bn_apply (HBM-bound passes).
#include "kernels.h"

namespace chiron {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// sums[c] += sum_m x[m][c], sums[C + c] += sum_m x[m][c]^2 (double: 440 000 terms per channel at the bench size)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long M, int C, double* __restrict__ sums) {
  __shared__ double red[2][256][4];
  const int c4n = C / 4;                    // float4 groups per row
  const int cg = threadIdx.x % c4n;         // this thread's channel group
  const int rsub = threadIdx.x / c4n;       // row sub-index inside the block
  const int rper = blockDim.x / c4n;        // rows per block iteration
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  for (long m = (long)blockIdx.x * rper + rsub; m < M; m += (long)gridDim.x * rper) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * C + cg * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] += (double)v[j];
      q[j] += (double)v[j] * (double)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][threadIdx.x][j] = s[j];
    red[1][threadIdx.x][j] = q[j];
  }
  __syncthreads();
  if (rsub == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a = 0, b = 0;
      for (int r = 0; r < rper; ++r) {
        a += red[0][r * c4n + cg][j];
        b += red[1][r * c4n + cg][j];
      }
      atomicAdd(&sums[cg * 4 + j], a);
      atomicAdd(&sums[C + cg * 4 + j], b);
    }
  }
}

__device__ __forceinline__ void bn_coeff(const double* sums, const float* scale, const float* offset, long M, int C, int c,
                                         float& inv, float& sh) {
  const double mean = sums[c] / (double)M;
  const double var = fmax(sums[C + c] / (double)M - mean * mean, 0.0);
  // association order of tf.nn.batch_normalization: inv = rsqrt(var + eps) * scale; y = x*inv + (offset - mean*inv)
  inv = (1.0f / sqrtf((float)var + 1e-5f)) * scale[c];
  sh = offset[c] - (float)mean * inv;
}

// x <- act( bn(x) [+ (add_bn ? bn'(add) : add)] ), in place on x.
__global__ __launch_bounds__(256) void bn_apply_kernel(float* __restrict__ x, const double* __restrict__ sums, const float* __restrict__ scale,
                                                       const float* __restrict__ offset, long M, int C, int relu, const float* __restrict__ add,
                                                       const double* __restrict__ add_sums, const float* __restrict__ add_scale,
                                                       const float* __restrict__ add_offset) {
  const int c4n = C / 4;
  const int cg = threadIdx.x % c4n;
  const int rsub = threadIdx.x / c4n;
  const int rper = blockDim.x / c4n;
  float inv[4], sh[4], ainv[4] = {1.f, 1.f, 1.f, 1.f}, ash[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bn_coeff(sums, scale, offset, M, C, cg * 4 + j, inv[j], sh[j]);
    if (add && add_sums) bn_coeff(add_sums, add_scale, add_offset, M, C, cg * 4 + j, ainv[j], ash[j]);
  }
  for (long m = (long)blockIdx.x * rper + rsub; m < M; m += (long)gridDim.x * rper) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + m * C + cg * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] * inv[j] + sh[j];
    if (add) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(add + m * C + cg * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (a[j] * ainv[j] + ash[j]) + v[j];
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    *reinterpret_cast<f32x4*>(x + m * C + cg * 4) = v;
  }
}

// 1x1 convolution of the one-channel signal (res_layer1 conv2a / branch1 conv1): out[(b, t)][c] = sig[b][t*stride] * w[c]
__global__ __launch_bounds__(256) void rank1_conv_kernel(const float* __restrict__ sig, const float* __restrict__ w, float* __restrict__ out,
                                                         long n_pos, int T_out, int L, int stride, int C) {
  const int c4n = C / 4;
  const long total = n_pos * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pos = i / c4n;
    const int c0 = (int)(i - pos * c4n) * 4;
    const long b = pos / T_out;
    const int t = (int)(pos - b * T_out);
    const float xv = sig[b * L + (long)t * stride];
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c0);
    *reinterpret_cast<f32x4*>(out + pos * C + c0) = wv * xv;
  }
}

// Column sums of a channels-last activation tensor of HALVES (the f16 engine's calibration pass, engine.hip: chiron_engine_calibrate):
// sums[c] += sum over rows m of x[m * ld + col0 + c], c < cin; rows of a time-major tensor (m = t * BP + b) count only for b < B.
__global__ __launch_bounds__(256) void colsum_f16_kernel(const _Float16* __restrict__ x, long rows, int ld, int col0, int cin, int BP, int B,
                                                         double* __restrict__ sums) {
  const int c = threadIdx.x;
  if (c >= cin) return;
  double a = 0;
  const long per = (rows + gridDim.x - 1) / gridDim.x;
  const long m0 = (long)blockIdx.x * per, m1 = m0 + per < rows ? m0 + per : rows;
  for (long m = m0; m < m1; ++m) {
    if (BP > 0 && (int)(m % BP) >= B) continue;
    a += (double)(float)x[m * ld + col0 + c];
  }
  atomicAdd(&sums[c], a);
}
void launch_colsum_f16(const void* x, long rows, int ld, int col0, int cin, int BP, int B, double* sums, hipStream_t stream) {
  hipLaunchKernelGGL(colsum_f16_kernel, dim3(512), dim3(256), 0, stream, reinterpret_cast<const _Float16*>(x), rows, ld, col0, cin, BP, B, sums);
}

void launch_bn_stats(const float* x, long M, int C, double* sums, hipStream_t stream) {
  hipMemsetAsync(sums, 0, 2 * (size_t)C * sizeof(double), stream);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(1024), dim3(256), 0, stream, x, M, C, sums);
}
void launch_bn_apply(float* x, const double* sums, const float* scale, const float* offset, long M, int C, int relu, const float* add,
                     const double* add_sums, const float* add_scale, const float* add_offset, hipStream_t stream) {
  hipLaunchKernelGGL(bn_apply_kernel, dim3(2048), dim3(256), 0, stream, x, sums, scale, offset, M, C, relu, add, add_sums, add_scale,
                     add_offset);
}
void launch_rank1_conv(const float* sig, const float* w, float* out, long n_pos, int T_out, int L, int stride, int C, hipStream_t stream) {
  hipLaunchKernelGGL(rank1_conv_kernel, dim3(2048), dim3(256), 0, stream, sig, w, out, n_pos, T_out, L, stride, C);
}

}  // namespace chiron
