// LSTM recurrence for gfx950: the TF while_loop of LSTMCell(100) under dynamic_rnn with
// sequence_length (chiron/rnn.py:49-65 / :140-145; op composition recorded in the shipped .meta
// graphs) as ONE persistent launch per layer.
//
//   workgroup = 16 batch rows x 1 direction, resident for all T steps (rows never interact, so no
//   grid-wide sync exists).  4 waves, one per SIMD; wave w owns hidden units [32w, 32w+32) for all
//   four gates = 8 accumulator tiles of v_mfma_f32_16x16x4_f32 (M = batch rows, N = units,
//   K = previous hidden).  The recurrent weights W_hh (100x400 fp32 = 160 KB, the size of the whole
//   LDS) live in VGPRs for the entire sequence: 8 tiles x 25 k-steps = 200 registers per lane.
//   h_{t-1} is exchanged through a double-buffered 16x100 LDS tile (one barrier per step);
//   the x-projection z_t (+bias, forget bias folded in) was produced by gemm.hip directly in this
//   kernel's accumulator-fragment order, so each tile is one coalesced 16-byte load per lane.
//   Masking: rows with t >= seq_len emit 0 and carry (c,h); the backward direction walks
//   t = seq_len-1-s per row (tf.reverse_sequence folded into index arithmetic, no copy).
#include "kernels.h"

// Every multiply-add below is written out (fmaf or separate ops) so that a row's result does not depend on
// which register slot / row position it occupies: batches can be re-packed without changing a bit.
#pragma clang fp contract(off)

namespace chiron {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HS = 102;  // LDS row stride of the h tile: rows*6 mod 32 distinct -> conflict-free reads

__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // 1 - 2/(e^{2x}+1); saturates correctly at +-inf, |abs err| ~1e-7
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f), 1.0f);
}

template <int KS>
__global__ __launch_bounds__(256, 1) void lstm_kernel(const LstmParams p) {
  __shared__ float hbuf[2][LSTM_ROWS * HS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 15;
  const int rg = lane >> 4;  // row group: rows rg*4 .. rg*4+3
  const int dir = blockIdx.x % p.ndir;
  const int btile = blockIdx.x / p.ndir;
  const int b0 = btile * LSTM_ROWS;
  const int nbt = p.BP / LSTM_ROWS;
  const int ubz = p.hpz >> 4;  // unit blocks present in z

  // ---- recurrent weights into registers (fragment order prepared on the host)
  float w[8][KS];
  {
    const float* wf = p.wfrag + (((long)dir * LSTM_WAVES + wave) * 8) * KS * 64 + lane;
#pragma unroll
    for (int ti = 0; ti < 8; ++ti)
#pragma unroll
      for (int s = 0; s < KS; ++s) w[ti][s] = wf[(ti * KS + s) * 64];
  }

  for (int i = tid; i < 2 * LSTM_ROWS * HS; i += 256) (&hbuf[0][0])[i] = 0.f;

  int len[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) len[r] = p.seq_len[b0 + rg * 4 + r];
  int maxlen = max(max(len[0], len[1]), max(len[2], len[3]));
  int minlen = min(min(len[0], len[1]), min(len[2], len[3]));
  maxlen = max(maxlen, __shfl_xor(maxlen, 16));
  maxlen = max(maxlen, __shfl_xor(maxlen, 32));
  minlen = min(minlen, __shfl_xor(minlen, 16));
  minlen = min(minlen, __shfl_xor(minlen, 32));
  maxlen = min(maxlen, p.T);
  const bool uniform = (minlen == maxlen) || dir == 0;  // all rows share t at every active step

  float c[2][4], hreg[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[u][r] = hreg[u][r] = 0.f;

  const int outw = p.ndir * p.H;
  __syncthreads();

  int cur = 0;
  for (int s = 0; s < maxlen; ++s) {
    // ---- z_t loads (consumed after the MFMA chain; their latency hides behind it)
    f32x4 z[8];
    int tr[4];
    bool act[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      act[r] = s < len[r];
      tr[r] = dir == 0 ? s : (act[r] ? len[r] - 1 - s : 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ub = wave * 2 + u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ub < ubz) {
          const int tile = g * ubz + ub;
          if (uniform) {
            const int t = dir == 0 ? s : maxlen - 1 - s;
            const long base = ((((long)t * nbt + btile) * p.ndir + dir) * p.tiles + tile) * 256;
            v = *reinterpret_cast<const f32x4*>(p.z + base + rg * 64 + col * 4);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const long base = ((((long)tr[r] * nbt + btile) * p.ndir + dir) * p.tiles + tile) * 256;
              v[r] = p.z[base + rg * 64 + col * 4 + r];
            }
          }
        }
        z[u * 4 + g] = v;
      }
    }

    // ---- h_{t-1} A fragments: A[i = lane&15][k = 4s + (lane>>4)]
    float a[KS];
    const float* hb = &hbuf[cur][col * HS + rg];
#pragma unroll
    for (int k = 0; k < KS; ++k) a[k] = hb[4 * k];

    f32x4 acc[8];
#pragma unroll
    for (int ti = 0; ti < 8; ++ti) acc[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int ti = 0; ti < 8; ++ti)
        acc[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], w[ti][k], acc[ti], 0, 0, 0);

    // ---- gates (lane-local: accumulator (row, unit) coincides for the four gate tiles)
    float* hn = &hbuf[cur ^ 1][0];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int unit = (wave * 2 + u) * 16 + col;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gi = acc[u * 4 + 0][r] + z[u * 4 + 0][r];
        const float gj = acc[u * 4 + 1][r] + z[u * 4 + 1][r];
        const float gf = acc[u * 4 + 2][r] + z[u * 4 + 2][r];
        const float go = acc[u * 4 + 3][r] + z[u * 4 + 3][r];
        const float cn = fmaf(fast_sigmoid(gf), c[u][r], fast_sigmoid(gi) * fast_tanh(gj));
        const float hnew = fast_sigmoid(go) * fast_tanh(cn);
        if (act[r]) {
          c[u][r] = cn;
          hreg[u][r] = hnew;
        }
        if (unit < p.H) {
          const int row = rg * 4 + r;
          hn[row * HS + unit] = hreg[u][r];
          const int to = act[r] ? tr[r] : s;
          p.out[((long)to * p.BP + b0 + row) * outw + dir * p.H + unit] = act[r] ? hnew : 0.f;
        }
      }
    }
    cur ^= 1;
    __syncthreads();
  }

  // ---- frames past the longest row of this tile read back as zeros (dynamic_rnn semantics)
  for (int s = maxlen; s < p.T; ++s) {
    for (int i = tid; i < LSTM_ROWS * p.H; i += 256) {
      const int row = i / p.H;
      const int unit = i - row * p.H;
      p.out[((long)s * p.BP + b0 + row) * outw + dir * p.H + unit] = 0.f;
    }
  }
}

int lstm_ksteps(int H) { return (H + 3) / 4; }

void launch_lstm(const LstmParams& p, hipStream_t stream) {
  const int grid = (p.BP / LSTM_ROWS) * p.ndir;
  // hidden = 100 is the only size the reference's shipped models use (rnn.py:23 hidden_num=100)
  hipLaunchKernelGGL(lstm_kernel<25>, dim3(grid), dim3(256), 0, stream, p);
}

}  // namespace chiron
