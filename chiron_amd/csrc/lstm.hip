// LSTM recurrence for gfx950: the TF while_loop of LSTMCell(100) under dynamic_rnn with
// sequence_length (chiron/rnn.py:49-65 / :140-145; op composition recorded in the shipped .meta
// graphs) as ONE persistent launch per layer.
//
//   workgroup = 4*NG batch rows x 1 direction, resident for all T steps (rows never interact, so no
//   grid-wide sync exists).  7 waves; wave w owns hidden units [16w, 16w+16) for all four gates = 64
//   matrix columns = exactly the N extent of ONE v_mfma_f32_4x4x1_16B_f32 (16 blocks of a 4x4 outer
//   product: M = 4 batch rows, N = 64 columns, K = 1).  Its slice of W_hh lives in VGPRs for the whole
//   sequence (100 registers per lane; W_hh = 160 KB fp32 = the entire LDS, spread over 7 register files).
//   The 4-row MFMA shape is what lets B = 1100 fill the chip: 16-row tiles give 138 workgroups for 256
//   CUs, 4-row groups give 550 half-CU workgroups; the instruction runs at the same 64 FLOP/clk/SIMD.
//   (v_mfma_f32_* shares the fp32 VALU datapath -- tools/ubench/barrier_mfma.hip -- so the gate math can
//   not hide behind it; what counts is total issue time, and this layout needs ONE cell per lane.)
//
//   per step and row group:  acc[4 rows] (lane = gate*16 + unit)  <-  h_{t-1} . W_hh      100 MFMAs
//                            + z_t (x-projection, produced by gemm.hip in exactly this fragment order)
//                            4x4 transpose through LDS: lane (row, unit) gets i, j, f, o of its cell
//                            c = sig(f)*c + sig(i)*tanh(j);  h = sig(o)*tanh(c)   (forget bias folded in z)
//   h is exchanged through a double-buffered [4 rows][100] LDS tile per group, one barrier per step.
//   Masking: rows with t >= seq_len emit 0 and carry (c,h); the backward direction walks
//   t = seq_len-1-s per row (tf.reverse_sequence folded into index arithmetic, no copy).
#include "kernels.h"

// Every multiply-add below is written out (fmaf or separate ops) so that a row's result does not depend on
// which register slot / row position it occupies: batches can be re-packed without changing a bit.
#pragma clang fp contract(off)

namespace chiron {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HS4 = 104;  // LDS row stride (floats) of an h row: 16-byte aligned rows for ds_read_b128

__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // 1 - 2/(e^{2x}+1); saturates correctly at +-inf, |abs err| ~1e-7
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f), 1.0f);
}

template <int NG, bool UNIFORM>
__device__ __forceinline__ void lstm_loop(const LstmParams& p, const float (&w)[LSTM_K], float* hbuf, float* tbuf,
                                          const int (&len4)[NG][4], const int (&lenr)[NG], int maxlen, int dir, int g0,
                                          int wave, int lane) {
  const int unit = wave * 16 + (lane & 15);
  const int row = lane >> 4;         // after the transpose: this lane's batch row within the group
  const bool live = unit < p.H;
  const int nb4 = p.BP >> 2;
  const int outw = p.ndir * p.H;
  const long zstep = (long)nb4 * p.ndir * LSTM_ZCOLS * 4;  // floats between consecutive t
  const float* zb = p.z + (((long)g0 * p.ndir + dir) * LSTM_ZCOLS + wave * 64 + lane) * 4;
  const long zgrp = (long)p.ndir * LSTM_ZCOLS * 4;         // floats between consecutive row groups
  float* tw = tbuf + wave * NG * 256;                      // this wave's transpose scratch

  float c[NG], hprev[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) c[g] = hprev[g] = 0.f;

  int cur = 0;
  for (int s = 0; s < maxlen; ++s) {
    // ---- z_t for the 4 rows of each group (one 16-byte load per lane when t is uniform)
    f32x4 z[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (UNIFORM) {
        const int t = dir == 0 ? s : maxlen - 1 - s;
        z[g] = *reinterpret_cast<const f32x4*>(zb + t * zstep + g * zgrp);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool a = s < len4[g][r];
          const int t = dir == 0 ? s : (a ? len4[g][r] - 1 - s : 0);
          z[g][r] = zb[t * zstep + g * zgrp + r];
        }
      }
    }
    // ---- acc = h_{s-1} . W_hh : A[block][i] = h[row i][k] for every block, B = this lane's column of W_hh
    constexpr int NA = NG == 1 ? 2 : 1;  // a lone group alternates two accumulators (dependent MFMA = 12 cycles)
    f32x4 acc[NG][NA];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int a = 0; a < NA; ++a) acc[g][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* hb = hbuf + cur * (NG * 4 * HS4) + (lane & 3) * HS4;
#pragma unroll
    for (int q = 0; q < LSTM_K / 4; ++q) {
      f32x4 a4[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) a4[g] = *reinterpret_cast<const f32x4*>(hb + g * 4 * HS4 + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < NG; ++g)
          acc[g][(q * 4 + j) % NA] =
              __builtin_amdgcn_mfma_f32_4x4x1f32(a4[g][j], w[4 * q + j], acc[g][(q * 4 + j) % NA], 0, 0, 0);
    }
    // ---- gates: transpose (lane = gate*16+unit, reg = row) -> (lane = row*16+unit, reg = gate)
    float* hn = hbuf + (cur ^ 1) * (NG * 4 * HS4);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      f32x4 v = acc[g][0];
      if (NA == 2) v += acc[g][NA - 1];
      v += z[g];
      float* ts = tw + g * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) ts[(r * 16 + (lane & 15)) * 4 + (lane >> 4)] = v[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(tw + g * 256 + lane * 4);  // i, j, f, o of (row, unit)
      const bool act = UNIFORM ? true : (s < lenr[g]);
      const float cn = fmaf(fast_sigmoid(q[2]), c[g], fast_sigmoid(q[0]) * fast_tanh(q[1]));
      const float hnew = fast_sigmoid(q[3]) * fast_tanh(cn);
      c[g] = act ? cn : c[g];
      hprev[g] = act ? hnew : hprev[g];
      if (live) {
        hn[(g * 4 + row) * HS4 + unit] = hprev[g];
        const int to = dir == 0 ? s : (UNIFORM ? maxlen - 1 - s : (act ? lenr[g] - 1 - s : s));
        p.out[((long)to * p.BP + (g0 + g) * 4 + row) * outw + dir * p.H + unit] = act ? hnew : 0.f;
      }
    }
    cur ^= 1;
    __syncthreads();
  }
}

template <int NG>
__global__ __launch_bounds__(64 * LSTM_NW, NG == 1 ? 4 : 2) void lstm_kernel(const LstmParams p) {
  __shared__ __attribute__((aligned(16))) float hbuf[2 * NG * 4 * HS4];
  __shared__ __attribute__((aligned(16))) float tbuf[LSTM_NW * NG * 256];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x % p.ndir;
  const int g0 = (blockIdx.x / p.ndir) * NG;  // first 4-row group of this workgroup

  // ---- this wave's 64 columns of the recurrent weights (fragment order prepared on the host)
  float w[LSTM_K];
  {
    const float* wf = p.wfrag + ((long)dir * LSTM_NW + wave) * LSTM_K * 64 + lane;
#pragma unroll
    for (int k = 0; k < LSTM_K; ++k) w[k] = wf[k * 64];
  }
  for (int i = tid; i < 2 * NG * 4 * HS4; i += 64 * LSTM_NW) hbuf[i] = 0.f;

  int len4[NG][4], lenr[NG];
  int maxlen = 0, minlen = 1 << 30;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      len4[g][r] = min(p.seq_len[(g0 + g) * 4 + r], p.T);
      maxlen = max(maxlen, len4[g][r]);
      minlen = min(minlen, len4[g][r]);
    }
    lenr[g] = len4[g][0];
#pragma unroll
    for (int r = 1; r < 4; ++r) lenr[g] = (lane >> 4) == r ? len4[g][r] : lenr[g];
  }
  __syncthreads();

  if (minlen == maxlen)
    lstm_loop<NG, true>(p, w, hbuf, tbuf, len4, lenr, maxlen, dir, g0, wave, lane);
  else
    lstm_loop<NG, false>(p, w, hbuf, tbuf, len4, lenr, maxlen, dir, g0, wave, lane);

  // ---- frames past the longest row of this workgroup read back as zeros (dynamic_rnn semantics)
  const int outw = p.ndir * p.H;
  for (int s = maxlen; s < p.T; ++s) {
    for (int i = tid; i < NG * 4 * p.H; i += 64 * LSTM_NW) {
      const int row = i / p.H;
      const int unit = i - row * p.H;
      p.out[((long)s * p.BP + g0 * 4 + row) * outw + dir * p.H + unit] = 0.f;
    }
  }
}

void launch_lstm(const LstmParams& p, hipStream_t stream) {
  // hidden = 100 is the only size the reference's shipped models use (rnn.py:23 hidden_num=100)
  const int groups = p.BP / 4;
  const int ng = p.rows_per_wg / 4;
  const dim3 block(64 * LSTM_NW);
  if (ng == 1)
    hipLaunchKernelGGL(lstm_kernel<1>, dim3(groups * p.ndir), block, 0, stream, p);
  else if (ng == 2)
    hipLaunchKernelGGL(lstm_kernel<2>, dim3(groups / 2 * p.ndir), block, 0, stream, p);
  else
    hipLaunchKernelGGL(lstm_kernel<4>, dim3(groups / 4 * p.ndir), block, 0, stream, p);
}

}  // namespace chiron
