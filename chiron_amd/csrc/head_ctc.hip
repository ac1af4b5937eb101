// FC head, path_prob, greedy CTC decode and SparseTensor construction for gfx950.
//
//   fc_kernel        rnn.py:72-96  -- HBM-bound: reads lasth once.
//   greedy_kernel    tf.nn.ctc_greedy_decoder(merge_repeated=True) (chiron_eval.py:485-487) and
//                    path_prob (chiron_eval.py:116-136); one wave per segment, ballot/popcount
//                    compaction.
//   scan/scatter     the (indices, values, dense_shape) SparseTensor the reference dequeues
//                    (chiron_eval.py:403-409), built on device.
#include "kernels.h"

#include <cstdlib>

namespace chiron {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// --------------------------------------------------------------------------------------------
// FC head.  logits[b][t][k] = sum_u (h_fw[u]*w[0][u] + h_bw[u]*w[1][u] + bias[u]) * wc[u][k] + bc[k]
// HBM-bound (reads lasth once).  16 lanes per position (t, b), four positions -- four consecutive batch rows of one
// frame, 4 x 800 contiguous bytes of the time-major lasth -- per wave and iteration: lane sl of a group owns the
// float4 column groups sl, sl+16, sl+32, sl+48 (< 2H/4) of its row, the 16-lane sums are four DPP row rotations per
// class (a 64-lane butterfly per position cost 6 cross-lane steps per class and made the kernel issue-bound).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}

template <bool RAW16>
__global__ __launch_bounds__(256) void fc_kernel(const FcParams p) {
  __builtin_amdgcn_s_setprio(3);   // short kernels between the other batches' persistent GEMM waves: see DESIGN 3.2 (small kernels in the mix)
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 4, sl = lane & 15;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int H = p.H, K = p.K;
  const int nl = (2 * H) / 4;  // float4 column groups of a row (50 for H=100; <= 64)

  // per-lane folded weights: cw[s][j][k] = w[dir][u] * wc[u][k] for column 4*(sl + 16 s) + j
  float cw[4][4][CHIRON_KMAX];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cidx = 4 * (sl + 16 * s) + j;  // column in [0, 2H)
      const bool ok = sl + 16 * s < nl;
      const int d = ok ? cidx / H : 0;
      const int u = ok ? cidx - d * H : 0;
      const float wd = ok ? p.w[d * H + u] : 0.f;
#pragma unroll
      for (int k = 0; k < CHIRON_KMAX; ++k) cw[s][j][k] = (ok && k < K) ? wd * p.wc[u * K + k] : 0.f;
    }
  // constant term: sum_u bias[u]*wc[u][k] + bc[k]  (same for every position)
  float cst[CHIRON_KMAX];
#pragma unroll
  for (int k = 0; k < CHIRON_KMAX; ++k) {
    float s = 0.f;
    if (k < K) {
      for (int u = lane; u < H; u += 64) s += p.bias[u] * p.wc[u * K + k];
    }
    s = wave_sum(s);
    cst[k] = (k < K) ? s + p.bc[k] : 0.f;
  }

  const int nb4 = (p.B + 3) >> 2;
  const int nunits = p.T * nb4;  // (frame, 4-row group)
  auto load_unit = [&](int q, f32x4 (&h)[4]) {
    const int t = q / nb4;
    const int b = (q - t * nb4) * 4 + sub;
    const long row = (long)t * p.BP + (b < p.B ? b : 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      h[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int f = sl + 16 * s;
      if (f < nl) {
        if (p.split) {
          typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
          const int col = 4 * f;  // 4 consecutive columns never straddle a 32-element block
          const _Float16* hq = reinterpret_cast<const _Float16*>(p.lasth) + (row * p.ld + (col >> 5) * 32) * 2 + (col & 31);
          const f16x4 hi = *reinterpret_cast<const f16x4*>(hq), lo = *reinterpret_cast<const f16x4*>(hq + 32);
          h[s] = (f32x4){(float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]};
        } else if (p.f16) {
          typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
          const f16x4 hh = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(p.lasth) + row * 2 * H + 4 * f);
          h[s] = (f32x4){(float)hh[0], (float)hh[1], (float)hh[2], (float)hh[3]};
        } else {
          h[s] = *reinterpret_cast<const f32x4*>(p.lasth + row * 2 * H + 4 * f);
        }
      }
    }
  };
  // fp16 engine: the prefetched unit stays in the halves it was loaded as and is converted when it is consumed -- converting
  // in load_unit makes the conversion wait for the load, i.e. no prefetch at all (0.49 ms at B = 4096 for half the bytes the
  // fp32 engine reads in 0.33 ms)
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  constexpr bool raw16 = RAW16;   // p.f16 && !p.split (launch_fc)
  auto load_raw16 = [&](int q, f16x4 (&r)[4]) {
    const int t = q / nb4;
    const int b = (q - t * nb4) * 4 + sub;
    const long row = (long)t * p.BP + (b < p.B ? b : 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      r[s] = (f16x4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      const int f = sl + 16 * s;
      if (f < nl) r[s] = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(p.lasth) + row * 2 * H + 4 * f);
    }
  };
  f32x4 hn[4];
  f16x4 rn[4];
  if (wave < nunits) {
    if constexpr (raw16) load_raw16(wave, rn);
    else load_unit(wave, hn);
  }
  for (int q = wave; q < nunits; q += nwaves) {
    const int t = q / nb4;
    const int b = (q - t * nb4) * 4 + sub;
    const bool live = b < p.B;
    f32x4 h[4];
    if constexpr (raw16) {
#pragma unroll
      for (int s = 0; s < 4; ++s) h[s] = (f32x4){(float)rn[s][0], (float)rn[s][1], (float)rn[s][2], (float)rn[s][3]};
      if (q + nwaves < nunits) load_raw16(q + nwaves, rn);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) h[s] = hn[s];
      if (q + nwaves < nunits) load_unit(q + nwaves, hn);  // the next unit's rows are in flight during this one's arithmetic
    }
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < CHIRON_KMAX; ++k) {
      if (k >= K) break;
      float a = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) a = fmaf(h[s][j], cw[s][j][k], a);
      a = row16_sum(a);
      if (sl == k) v = a + cst[k];
    }
    if (live && sl < K) p.logits[((long)b * p.T + t) * K + sl] = v;
  }
}

// The fp32 / split engines' form: 32 lanes per position, two positions per wave and iteration, lane sl of a half owns the
// column groups sl and sl + 32.  Half the folded weights per lane: 76 registers instead of 166 -- with three batches in
// flight the kernel runs in whatever the other batches' persistent GEMM workgroups leave of a SIMD's 512 registers (two
// GEMM waves: 208), and the number of its waves that fit decides how long a batch waits in it (0.73 ms in the mix for a
// kernel that takes 0.13 ms alone, profiles/r03_overlap_slots3.txt).
__device__ __forceinline__ float row32_sum(float v) {   // the sum of a 32-lane half, valid in its upper 16 lanes
  v = row16_sum(v);
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1, 3
}

__global__ __launch_bounds__(256) void fc32_kernel(const FcParams p) {
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 5, sl = lane & 31;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int H = p.H, K = p.K;
  const int nl = (2 * H) / 4;  // float4 column groups of a row (50 for H=100; <= 64)

  float cw[2][4][CHIRON_KMAX];   // cw[s][j][k] = w[dir][u] * wc[u][k] for column 4*(sl + 32 s) + j
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cidx = 4 * (sl + 32 * s) + j;
      const bool ok = sl + 32 * s < nl;
      const int d = ok ? cidx / H : 0;
      const int u = ok ? cidx - d * H : 0;
      const float wd = ok ? p.w[d * H + u] : 0.f;
#pragma unroll
      for (int k = 0; k < CHIRON_KMAX; ++k) cw[s][j][k] = (ok && k < K) ? wd * p.wc[u * K + k] : 0.f;
    }
  float cst = 0.f;               // lane 16 + k of a half: sum_u bias[u]*wc[u][k] + bc[k]
#pragma unroll
  for (int k = 0; k < CHIRON_KMAX; ++k) {
    float s = 0.f;
    if (k < K) {
      for (int u = lane; u < H; u += 64) s += p.bias[u] * p.wc[u * K + k];
    }
    s = wave_sum(s);
    if (k < K && sl == 16 + k) cst = s + p.bc[k];
  }

  const int nb2 = (p.B + 1) >> 1;
  const int nunits = p.T * nb2;  // (frame, 2-row group)
  auto load_unit = [&](int q, f32x4 (&h)[2]) {
    const int t = q / nb2;
    const int b = (q - t * nb2) * 2 + sub;
    const long row = (long)t * p.BP + (b < p.B ? b : 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      h[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int f = sl + 32 * s;
      if (f < nl) {
        if (p.split) {
          typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
          const int col = 4 * f;  // 4 consecutive columns never straddle a 32-element block
          const _Float16* hq = reinterpret_cast<const _Float16*>(p.lasth) + (row * p.ld + (col >> 5) * 32) * 2 + (col & 31);
          const f16x4 hi = *reinterpret_cast<const f16x4*>(hq), lo = *reinterpret_cast<const f16x4*>(hq + 32);
          h[s] = (f32x4){(float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]};
        } else {
          h[s] = *reinterpret_cast<const f32x4*>(p.lasth + row * 2 * H + 4 * f);
        }
      }
    }
  };
  f32x4 hn[2];
  if (wave < nunits) load_unit(wave, hn);
  for (int q = wave; q < nunits; q += nwaves) {
    const int t = q / nb2;
    const int b = (q - t * nb2) * 2 + sub;
    const bool live = b < p.B;
    f32x4 h[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) h[s] = hn[s];
    if (q + nwaves < nunits) load_unit(q + nwaves, hn);  // the next unit's rows are in flight during this one's arithmetic
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < CHIRON_KMAX; ++k) {
      if (k >= K) break;
      float a = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) a = fmaf(h[s][j], cw[s][j][k], a);
      a = row32_sum(a);
      if (sl == 16 + k) v = a + cst;
    }
    if (live && sl >= 16 && sl < 16 + K) p.logits[((long)b * p.T + t) * K + (sl - 16)] = v;
  }
}

void launch_fc(const FcParams& p, hipStream_t stream) {
  static const bool wide16 = getenv("CHIRON_FC16") != nullptr;   // A/B switch: the 16-lane form for every dtype
  if (p.f16 && !p.split)
    hipLaunchKernelGGL(fc_kernel<true>, dim3(256 * 8), dim3(256), 0, stream, p);
  else if (wide16)
    hipLaunchKernelGGL(fc_kernel<false>, dim3(256 * 8), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(fc32_kernel, dim3(256 * 16), dim3(256), 0, stream, p);
}

// --------------------------------------------------------------------------------------------
// Greedy CTC + path_prob: one wave per segment.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void greedy_kernel(const GreedyParams p) {
  __builtin_amdgcn_s_setprio(3);   // short kernels between the other batches' persistent GEMM waves: see DESIGN 3.2 (small kernels in the mix)
  const int lane = threadIdx.x & 63;
  const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (b >= p.B) return;
  const int K = p.K, T = p.T;
  const int blank = K - 1;
  const int len = min(max(p.seq_len[b], 0), T);
  const float* lg = p.logits + (long)b * T * K;
  uint8_t* out = p.labels + (long)b * T;

  float negsum = 0.f, diffsum = 0.f;
  int base = 0;
  int carry_k = -1;  // argmax of the last frame of the previous 64-frame group
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    int k = -1;
    float m1 = 0.f, m2 = 0.f;
    if (t < T) {
      // first-max argmax (Eigen maxCoeff tie rule) and runner-up value
      m1 = lg[t * K];
      k = 0;
      m2 = -INFINITY;
      for (int j = 1; j < K; ++j) {
        const float v = lg[t * K + j];
        if (v > m1) {
          m2 = m1;
          m1 = v;
          k = j;
        } else if (v > m2) {
          m2 = v;
        }
      }
      diffsum += m1 - m2;
      if (t < len) negsum -= m1;
    }
    int prev = __shfl_up(k, 1);
    if (lane == 0) prev = carry_k;
    carry_k = __shfl(k, 63);
    const bool emit = (t < len) && (k != blank) && (k != prev);
    const unsigned long long mask = __ballot(emit);
    if (emit) {
      const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
      out[pos] = (uint8_t)k;
    }
    base += __popcll(mask);
  }
  negsum = wave_sum(negsum);
  diffsum = wave_sum(diffsum);
  if (lane == 0) {
    p.count[b] = base;
    p.log_prob[b] = negsum;
    if (p.prob_logits) p.prob_logits[b] = diffsum / (float)T;
  }
}

void launch_greedy(const GreedyParams& p, hipStream_t stream) {
  const int blocks = (p.B + 3) / 4;
  hipLaunchKernelGGL(greedy_kernel, dim3(blocks), dim3(256), 0, stream, p);
}

__global__ __launch_bounds__(256) void path_prob_kernel(const PathProbParams p) {
  __builtin_amdgcn_s_setprio(3);   // short kernels between the other batches' persistent GEMM waves: see DESIGN 3.2 (small kernels in the mix)
  const int lane = threadIdx.x & 63;
  const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (b >= p.B) return;
  const float* lg = p.logits + (long)b * p.T * p.K;
  float s = 0.f;
  for (int t = lane; t < p.T; t += 64) {
    float m1 = lg[t * p.K], m2 = -INFINITY;
    for (int j = 1; j < p.K; ++j) {
      const float v = lg[t * p.K + j];
      if (v > m1) {
        m2 = m1;
        m1 = v;
      } else if (v > m2) {
        m2 = v;
      }
    }
    s += m1 - m2;
  }
  s = wave_sum(s);
  if (lane == 0) p.prob_logits[b] = s / (float)p.T;
}

void launch_path_prob(const PathProbParams& p, hipStream_t stream) {
  hipLaunchKernelGGL(path_prob_kernel, dim3((p.B + 3) / 4), dim3(256), 0, stream, p);
}

// --------------------------------------------------------------------------------------------
// SparseTensor build: exclusive scan of the per-row counts (one workgroup), then scatter.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_kernel(const SparseParams p) {
  __builtin_amdgcn_s_setprio(3);   // short kernels between the other batches' persistent GEMM waves: see DESIGN 3.2 (small kernels in the mix)
  __shared__ long part[1024];
  __shared__ int pmax[1024];
  const int tid = threadIdx.x;
  const int chunk = (p.B + 1023) / 1024;
  const int lo = min(tid * chunk, p.B), hi = min(lo + chunk, p.B);
  long s = 0;
  int mx = 0;
  for (int i = lo; i < hi; ++i) {
    s += p.count[i];
    mx = max(mx, p.count[i]);
  }
  part[tid] = s;
  pmax[tid] = mx;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    long v = 0;
    int m = 0;
    if (tid >= o) {
      v = part[tid - o];
      m = pmax[tid - o];
    }
    __syncthreads();
    part[tid] += v;
    pmax[tid] = max(pmax[tid], m);
    __syncthreads();
  }
  long run = part[tid] - s;  // exclusive prefix of this thread's chunk
  for (int i = lo; i < hi; ++i) {
    p.offsets[i] = run;
    run += p.count[i];
  }
  if (tid == 1023) {
    p.offsets[p.B] = part[1023];
    p.meta[0] = part[1023];
    p.meta[1] = p.B;
    p.meta[2] = pmax[1023];
  }
}

__global__ __launch_bounds__(64) void scatter_kernel(const SparseParams p) {
  __builtin_amdgcn_s_setprio(3);   // short kernels between the other batches' persistent GEMM waves: see DESIGN 3.2 (small kernels in the mix)
  const int b = blockIdx.x;
  const long off = p.offsets[b];
  const int n = p.count[b];
  const uint8_t* src = p.labels + (long)b * p.T;
  for (int j = threadIdx.x; j < n; j += 64) {
    p.indices[(off + j) * 2] = b;
    p.indices[(off + j) * 2 + 1] = j;
    p.values[off + j] = src[j];
  }
}

void launch_sparse(const SparseParams& p, hipStream_t stream) {
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, p);
  hipLaunchKernelGGL(scatter_kernel, dim3(p.B), dim3(64), 0, stream, p);
}

}  // namespace chiron
