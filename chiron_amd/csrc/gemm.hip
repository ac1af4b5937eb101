// Fused conv / LSTM-projection GEMM for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak).
//
// Replaces the tf.nn.conv2d + batch_normalization + relu (+ add) chains of chiron/cnn.py:15-83,
// :234-262 and the x-part of the LSTMCell MatMul (rnn.py:49-65) of the reference.
//
//   block tile 128(M) x 128(N) x 32(K), 256 threads = 4 waves, each wave 64x64 = 2x2 MFMA tiles.
//   A and B tiles are register-staged into LDS ([rows][36] floats: conflict-free ds_read_b128 /
//   ds_write_b128), global loads of chunk k+1 are in flight while chunk k is on the matrix pipe.
//   K order inside a chunk is permuted (lanes 0-31 take k=8g+j, lanes 32-63 take k=8g+4+j) so every
//   operand fetch is one ds_read_b128 feeding four MFMAs; A and B use the same permutation.
#include "kernels.h"

namespace chiron {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LDS_LD = GEMM_BK + 4;  // 36 floats: 16B-aligned rows, conflict-free b128 access

__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) float As[GEMM_BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[GEMM_BN * LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: the dispatcher places block id on XCD id%8; all N-blocks of one M-block
  // are consecutive on the same XCD so the A panel is fetched from HBM once and re-read from that L2.
  const int nblocks_n = (p.N + GEMM_BN - 1) / GEMM_BN;
  const int mblocks = (p.M + GEMM_BM - 1) / GEMM_BM;
  const int id = blockIdx.x;
  const int xcd = id & 7;
  const int slot = id >> 3;
  const int mblk = (slot / nblocks_n) * 8 + xcd;
  const int nblk = slot % nblocks_n;
  if (mblk >= mblocks) return;
  const int m0 = mblk * GEMM_BM;
  const int n0 = nblk * GEMM_BN;

  // ---- per-thread loader coordinates: 4 A rows + 4 B rows, one float4 (4 k) each
  const int kq = tid & 7;
  const int lr = tid >> 3;  // 0..31
  int rb[4], rt[4];
  bool rvalid[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + lr + 32 * i;
    bool v = m < p.M;
    int b, t;
    if (p.m_time_major) {
      t = m / p.BP;
      b = m - t * p.BP;
      v = v && (b < p.B);
    } else {
      b = m / p.T_out;
      t = m - b * p.T_out;
    }
    rb[i] = b;
    rt[i] = t;
    rvalid[i] = v;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  f32x4 ra[4], rbv[4];
  const int nk = p.K / GEMM_BK;

  auto load_chunk = [&](int kc) {
    // locate the K-segment of this chunk (wave-uniform)
    const int k0 = kc * GEMM_BK;
    int s = 0, kofs = 0;
    while (s + 1 < p.nseg && k0 >= kofs + p.seg[s].kpad) {
      kofs += p.seg[s].kpad;
      ++s;
    }
    const GemmSeg& sg = p.seg[s];
    const int kk = k0 - kofs + 4 * kq;  // channel within the segment
    const bool kin = kk < sg.cin;
    if (sg.src != nullptr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int in_t = rt[i] * sg.stride + sg.shift;
        const bool ok = rvalid[i] && kin && in_t >= 0 && in_t < sg.w_in;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
          const long row = sg.time_major ? ((long)in_t * p.BP + rb[i]) : ((long)rb[i] * sg.w_in + in_t);
          v = *reinterpret_cast<const f32x4*>(sg.src + row * sg.lda + sg.col0 + kk);
        }
        ra[i] = v;
      }
    } else {
      f32x4 la = {0.f, 0.f, 0.f, 0.f}, lb = {0.f, 0.f, 0.f, 0.f};
      if (kin) {
        la = *reinterpret_cast<const f32x4*>(p.lift_a + kk);
        lb = *reinterpret_cast<const f32x4*>(p.lift_b + kk);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int in_t = rt[i] * sg.stride + sg.shift;
        const bool ok = rvalid[i] && kin && in_t >= 0 && in_t < sg.w_in;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
          const float x = p.sig[(long)rb[i] * p.L + in_t];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(x, la[j], lb[j]), 0.f);
        }
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rbv[i] = *reinterpret_cast<const f32x4*>(p.Wt + (long)(n0 + lr + 32 * i) * p.K + k0 + 4 * kq);
    }
  };

  load_chunk(0);
  const int li = lane & 31;
  const int kh = lane >> 5;

  for (int kc = 0; kc < nk; ++kc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(&As[(lr + 32 * i) * LDS_LD + 4 * kq]) = ra[i];
      *reinterpret_cast<f32x4*>(&Bs[(lr + 32 * i) * LDS_LD + 4 * kq]) = rbv[i];
    }
    __syncthreads();
    if (kc + 1 < nk) load_chunk(kc + 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        a[mi] = *reinterpret_cast<const f32x4*>(&As[(wm * 64 + mi * 32 + li) * LDS_LD + 8 * g + 4 * kh]);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        b[ni] = *reinterpret_cast<const f32x4*>(&Bs[(wn * 64 + ni * 32 + li) * LDS_LD + 8 * g + 4 * kh]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue.  C layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wn * 64 + ni * 32 + li;
    if (n >= p.N) continue;
    const float sh = p.shift ? p.shift[n] : 0.f;
    const float resa = p.res_a ? p.res_a[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mq = m0 + wm * 64 + mi * 32 + 8 * q + 4 * kh;  // first of 4 consecutive rows
        if (p.out_mode == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mq + r;
            if (m >= p.M) continue;
            float v = acc[mi][ni][4 * q + r] + sh;
            if (p.res_a) {
              const int b = m / p.T_out;
              const int t = m - b * p.T_out;
              v = fmaf(p.sig[(long)b * p.L + (long)t * p.res_stride], resa, v);
            }
            if (p.relu) v = fmaxf(v, 0.f);
            p.out[(long)m * p.ldo + n] = v;
          }
        } else {
          if (mq >= p.M) continue;
          const int t = mq / p.BP;
          const int b = mq - t * p.BP;  // multiple of 4
          const int nbt = p.BP >> 4;
          const int zcols = p.z_tiles * 16;
          const int dir = p.z_dir0 + n / zcols;
          const int nl = n % zcols;
          const long tile = (((long)t * nbt + (b >> 4)) * p.z_dirs_total + dir) * p.z_tiles + (nl >> 4);
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][4 * q + r] + sh;
          *reinterpret_cast<f32x4*>(p.out + tile * 256 + ((b & 15) >> 2) * 64 + (nl & 15) * 4) = v;
        }
      }
    }
  }
}

void launch_gemm(const GemmParams& p, hipStream_t stream) {
  const int nblocks_n = (p.N + GEMM_BN - 1) / GEMM_BN;
  const int mblocks = (p.M + GEMM_BM - 1) / GEMM_BM;
  const int mgroups = (mblocks + 7) / 8;
  const int grid = mgroups * nblocks_n * 8;
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(grid), dim3(256), 0, stream, p);
}

}  // namespace chiron
