// Fused conv / LSTM-projection GEMM for gfx950 (exact fp32 on v_mfma_f32_32x32x2_f32, 157 TF peak; the f16 and
// split dtypes run the same kernel on v_mfma_f32_32x32x16_f16).
//
// Replaces the tf.nn.conv2d + batch_normalization + relu (+ add) chains of chiron/cnn.py:15-83,
// :234-262 and the x-part of the LSTMCell MatMul (rnn.py:49-65) of the reference.
//
//   block tile 128(M) x 128(N) x 32(K), 256 threads = 4 waves, each wave 64x64 = 2x2 MFMA tiles (projections on the DMA
//   kernel: 128 x 160, waves 4 x 1, each 32 x 160 = 1x5 tiles, so that N = 8H = 800 is five whole tiles), persistent
//   workgroups over XCD-aware tile ids.  K order inside a chunk is permuted (lanes 0-31 take k=8g+j, lanes
//   32-63 take k=8g+4+j) so every operand fetch is one ds_read_b128 feeding four MFMAs; A and B use the same
//   permutation.  A is a sequence of K-segments (conv taps / fused inputs): per-row source pointers and
//   validity are computed once per segment.
//
//   gemm_f32_dma_kernel  every launch whose A operand is a tensor in HBM: LDS-DMA staging (see below).
//   gemm_f32_kernel      register-staged ([rows][36] LDS image, masked loads): the lifted first convolution, whose A
//                        operand relu(sig*a[c]+b[c]) is computed in the loader, and shapes the DMA kernel is not
//                        instantiated for.
#include "kernels.h"
#include "timing_variants.h"

#include <cstdlib>
#include <type_traits>

namespace chiron {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int LDS_LD = GEMM_BK + 4;  // 36 floats: 16B-aligned rows, conflict-free b128 access
constexpr int TILE_F = GEMM_BM * LDS_LD;

// One 4-row group of the x-projection goes to z[t][group][dir][col][4 rows].  The backward direction is stored
// by STEP instead of by frame: row r lands at s = seq_len[r] - 1 - t (frames past its length are never consumed),
// so that lstm.hip reads one 16-byte vector per step for either direction, whatever the lengths are.
struct ZGroup {
  unsigned base;   // float index of (t = 0, group, dir 0, col 0)
  unsigned per_t;  // floats per step
  int t, l0, l1, l2, l3;
  __device__ __forceinline__ ZGroup(const GemmParams& p, int t_, int b) : t(t_) {
    per_t = (unsigned)(p.BP >> 2) * p.z_dirs_total * p.z_cols * 4;
    base = (unsigned)(b >> 2) * p.z_dirs_total * p.z_cols * 4;
    const int4 len = *reinterpret_cast<const int4*>(p.z_seq_len + b);
    l0 = min(len.x, p.T_out), l1 = min(len.y, p.T_out), l2 = min(len.z, p.T_out), l3 = min(len.w, p.T_out);
  }
  __device__ __forceinline__ void store(const GemmParams& p, int dir, int nl, f32x4 v) const {
    const unsigned col = base + ((unsigned)dir * p.z_cols + nl) * 4;
    const bool uniform = l0 == l1 && l1 == l2 && l2 == l3;
    if (p.z_f16) {
      _Float16* o = reinterpret_cast<_Float16*>(p.out) + col;
      const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
      if (dir == 0) {
        *reinterpret_cast<f16x4*>(o + (unsigned)t * per_t) = h;
      } else if (uniform) {
        if (t < l0) *reinterpret_cast<f16x4*>(o + (unsigned)(l0 - 1 - t) * per_t) = h;
      } else {
        if (t < l0) o[(unsigned)(l0 - 1 - t) * per_t + 0] = h[0];
        if (t < l1) o[(unsigned)(l1 - 1 - t) * per_t + 1] = h[1];
        if (t < l2) o[(unsigned)(l2 - 1 - t) * per_t + 2] = h[2];
        if (t < l3) o[(unsigned)(l3 - 1 - t) * per_t + 3] = h[3];
      }
      return;
    }
    float* o = p.out + col;
    if (dir == 0) {
      *reinterpret_cast<f32x4*>(o + (unsigned)t * per_t) = v;
    } else if (uniform) {
      if (t < l0) *reinterpret_cast<f32x4*>(o + (unsigned)(l0 - 1 - t) * per_t) = v;
    } else {
      if (t < l0) o[(unsigned)(l0 - 1 - t) * per_t + 0] = v[0];
      if (t < l1) o[(unsigned)(l1 - 1 - t) * per_t + 1] = v[1];
      if (t < l2) o[(unsigned)(l2 - 1 - t) * per_t + 2] = v[2];
      if (t < l3) o[(unsigned)(l3 - 1 - t) * per_t + 3] = v[3];
    }
  }
};

// (b, t) of the rows of one 128-row tile without a per-row integer division: the quotient of the tile's first
// row is wave-uniform, every other row is at most two wrap-arounds away (inner >= 64).
struct RowSplit {
  int q0, r0, inner;  // m0 = q0 * inner + r0
  __device__ __forceinline__ RowSplit(int m0, int inner_) : inner(inner_) {
    q0 = m0 / inner_;
    r0 = m0 - q0 * inner_;
  }
  __device__ __forceinline__ void split(int dm, int& q, int& r) const {  // row m0 + dm, 0 <= dm < 128
    r = r0 + dm;
    q = q0;
    if (inner >= 64) {
      if (r >= inner) { r -= inner; ++q; }
      if (r >= inner) { r -= inner; ++q; }
    } else {
      const int e = r / inner;
      q += e;
      r -= e * inner;
    }
  }
};

// Epilogue of the DMA kernel for a tile that lies completely inside N: no per-store bounds checks or branches,
// one base address per row (the stores use immediate offsets), ReLU as one v_med3 per value, shift already in
// the accumulators.  ~100 VALU instructions per tile instead of ~350 -- they are all paid in matrix-pipe time.
template <bool RES, bool RELU, int OUTFMT = 0>
__device__ __forceinline__ void gemm_epilogue_lean(const GemmParams& p, f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn,
                                                   int li, int kh) {
  {
    const RowSplit rs(m0, p.T_out);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + wm * 64 + mi * 32 + li;
      if (m >= p.M) continue;
      float sv = 0.f;
      if (RES) {
        int b, t;
        rs.split(wm * 64 + mi * 32 + li, b, t);
        sv = p.sig[(long)b * p.L + (long)t * p.res_stride];
      }
      // f16 output: ldo counts 4-byte units, a row holds 2*ldo halves
      float* orow = p.out + (long)m * p.ldo + (OUTFMT == 1 ? (n0 + wn * 64 + 4 * kh) / 2 : OUTFMT == 2 ? n0 + wn * 64 : n0 + wn * 64 + 4 * kh);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][4 * q + r];
          if (RES) {
            const f32x4 ra4 = *reinterpret_cast<const f32x4*>(p.res_a + n0 + wn * 64 + ni * 32 + 8 * q + 4 * kh);
            const f32x4 rb4 = *reinterpret_cast<const f32x4*>(p.res_b + n0 + wn * 64 + ni * 32 + 8 * q + 4 * kh);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (OUTFMT == 1) v[r] = fmaf(sv, ra4[r], v[r]);   // f16 engines: res_b = 0 (shift folded, engine.hip), one fused rounding
              else v[r] += fmaf(sv, ra4[r], rb4[r]);
            }
          }
          if (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], 0.f, INFINITY);
          }
          if (OUTFMT == 1) {
            f16x4 hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (_Float16)v[r];
            *reinterpret_cast<f16x4*>(orow + (ni * 32 + 8 * q) / 2) = hv;
          } else if (OUTFMT == 2) {
            // split format: per 32-element block 64 B of hi halves then 64 B of lo halves (x = hi + lo to 2^-22);
            // this lane's 4 columns are elements e .. e+3 of block ni, e = 8q + 4kh
            f16x4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              hi[r] = (_Float16)v[r];
              lo[r] = (_Float16)(v[r] - (float)hi[r]);
            }
            _Float16* blk = reinterpret_cast<_Float16*>(orow + ni * 32) + 8 * q + 4 * kh;
            *reinterpret_cast<f16x4*>(blk) = hi;
            *reinterpret_cast<f16x4*>(blk + 32) = lo;
          } else {
            *reinterpret_cast<f32x4*>(orow + ni * 32 + 8 * q) = v;
          }
        }
      }
    }
  }
}

// Epilogue of the projection (ZOUT) layout of the DMA kernel: a 128 x 160 tile whose four waves split the ROWS (wave w
// owns rows 32w .. 32w+31 and all 160 columns = five 32x32 accumulators), so that N = 800 = 8H is five whole tiles.
// A lane holds 4 consecutive rows (one 4-row group of lstm.hip) of one column per register quad: one 16-byte store.
// Every per-store instruction is paid NNI * 4 times per lane and tile in matrix-pipe time, so the three common cases
// are decided once per wave / per 4-row group: forward-only tiles (no lengths, no predicates, five stores off one
// pointer), backward-only tiles whose groups have equal lengths (one predicate per group), and the general path.
template <int NNI, bool ZF16>
__device__ __forceinline__ void gemm_epilogue_zrows_t(const GemmParams& p, f32x16 (&acc)[1][NNI], int m0, int n0, int wave, int li, int kh) {
  typedef typename std::conditional<ZF16, _Float16, float>::type elem_t;
  typedef typename std::conditional<ZF16, f16x4, f32x4>::type vec_t;
  const RowSplit rs(m0, p.BP);
  const int zcols = p.z_cols;
  const unsigned gstride = (unsigned)p.z_dirs_total * zcols * 4;  // elements per 4-row group
  const unsigned per_t = (unsigned)(p.BP >> 2) * gstride;         // elements per step
  elem_t* const out0 = reinterpret_cast<elem_t*>(p.out) + (unsigned)(p.z_dir0 * zcols + n0 + li) * 4;  // this lane's column of block ni = 0
  const bool full_n = n0 + NNI * 32 <= p.N;
  const bool all_fwd = full_n && p.z_dir0 == 0 && n0 + NNI * 32 <= zcols;  // wave-uniform
  const bool all_bwd = full_n && (p.z_dir0 > 0 || n0 >= zcols);
  auto vec = [&](int ni, int q) {
    vec_t v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (elem_t)acc[0][ni][4 * q + r];
    return v;
  };
  // z is written once and read once, a whole projection later (1.4 GB per launch: no cache holds it): the whole-vector stores carry
  // the non-temporal hint.  +0.2 .. 0.4 % headline in three same-box A/B runs (profiles/r04_gemm_epilogue_ab.txt 11); the same hint
  // on the convolutions' outputs, the recurrence's z loads and outputs, Winograd's and the streaming kernel's stores costs 0 .. 0.8 %
  // (their consumers find them in the L2 / MALL).
  auto zst = [&](elem_t* ptr, const vec_t& v) { __builtin_nontemporal_store(v, reinterpret_cast<vec_t*>(ptr)); };
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // one register quad at a time: with the four quads' addresses and conversions scheduled together the epilogue, not the
    // main loop (162), set the kernel's register count (186 .. 205) -- and 176 is what fits next to two recurrence waves
    __builtin_amdgcn_sched_barrier(0);
    const int dm = wave * 32 + 8 * q + 4 * kh;  // first of 4 consecutive rows
    if (m0 + dm >= p.M) continue;
    int t, b;
    rs.split(dm, t, b);
    elem_t* const og = out0 + (unsigned)(b >> 2) * gstride;
    if (all_fwd) {
      elem_t* const o = og + (unsigned)t * per_t;
#pragma unroll
      for (int ni = 0; ni < NNI; ++ni) zst(o + ni * 128, vec(ni, q));
      continue;
    }
    const int4 len = *reinterpret_cast<const int4*>(p.z_seq_len + b);
    const int l0 = min(len.x, p.T_out), l1 = min(len.y, p.T_out), l2 = min(len.z, p.T_out), l3 = min(len.w, p.T_out);
    const bool uniform = l0 == l1 && l1 == l2 && l2 == l3;
    if (all_bwd && uniform) {
      if (t < l0) {
        elem_t* const o = og + (unsigned)(l0 - 1 - t) * per_t;
#pragma unroll
        for (int ni = 0; ni < NNI; ++ni) zst(o + ni * 128, vec(ni, q));
      }
      continue;
    }
#pragma unroll
    for (int ni = 0; ni < NNI; ++ni) {
      const int n = n0 + ni * 32 + li;
      if (n >= p.N) continue;
      elem_t* const o = og + ni * 128;
      const vec_t v = vec(ni, q);
      if (p.z_dir0 == 0 && n < zcols) {
        *reinterpret_cast<vec_t*>(o + (unsigned)t * per_t) = v;
      } else if (uniform) {
        if (t < l0) *reinterpret_cast<vec_t*>(o + (unsigned)(l0 - 1 - t) * per_t) = v;
      } else {
        if (t < l0) o[(unsigned)(l0 - 1 - t) * per_t + 0] = v[0];
        if (t < l1) o[(unsigned)(l1 - 1 - t) * per_t + 1] = v[1];
        if (t < l2) o[(unsigned)(l2 - 1 - t) * per_t + 2] = v[2];
        if (t < l3) o[(unsigned)(l3 - 1 - t) * per_t + 3] = v[3];
      }
    }
  }
}
template <int NNI>
__device__ __forceinline__ void gemm_epilogue_zrows(const GemmParams& p, f32x16 (&acc)[1][NNI], int m0, int n0, int wave, int li, int kh) {
  if (p.z_f16)
    gemm_epilogue_zrows_t<NNI, true>(p, acc, m0, n0, wave, li, kh);
  else
    gemm_epilogue_zrows_t<NNI, false>(p, acc, m0, n0, wave, li, kh);
}

// Stores LO <= idx < HI of the 16 16-byte stores a lane owns (idx = (outer*2 + inner)*4 + q): the DMA kernel
// spreads one tile's stores over the first chunks of the next tile.
template <bool ZOUT, bool RES, int LO = 0, int HI = 16, bool ADD_SHIFT = true>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn,
                                              int li, int kh) {
    // ---- epilogue.  32x32 accumulator layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Conv launches accumulate C^T (operands swapped), so a lane owns ONE output row m and four
  // consecutive columns n per register quad: 16-byte stores instead of 4-byte ones (the 4-byte
  // epilogue was store-issue-bound: 64 store instructions per lane per tile).
  if (!ZOUT) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      if (mi * 8 >= HI || mi * 8 + 8 <= LO) continue;
      const int m = m0 + wm * 64 + mi * 32 + li;
      if (m >= p.M) continue;
      float sv = 0.f;
      if (RES) {
        const int b = m / p.T_out;
        const int t = m - b * p.T_out;
        sv = p.sig[(long)b * p.L + (long)t * p.res_stride];
      }
      float* orow = p.out + (long)m * p.ldo;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if ((mi * 2 + ni) * 4 + q < LO || (mi * 2 + ni) * 4 + q >= HI) continue;
          const int n = n0 + wn * 64 + ni * 32 + 8 * q + 4 * kh;  // 4 consecutive columns
          if (n >= p.N) continue;
          f32x4 v;
          if (ADD_SHIFT) {
            const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][4 * q + r] + sh[r];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][4 * q + r];
          }
          if (RES) {
            const f32x4 ra4 = *reinterpret_cast<const f32x4*>(p.res_a + n);
            const f32x4 rb4 = *reinterpret_cast<const f32x4*>(p.res_b + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += fmaf(sv, ra4[r], rb4[r]);
          }
          if (p.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          *reinterpret_cast<f32x4*>(orow + n) = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      if (ni * 8 >= HI || ni * 8 + 8 <= LO) continue;
      const int n = n0 + wn * 64 + ni * 32 + li;
      if (n >= p.N) continue;
      const float sh = ADD_SHIFT ? p.shift[n] : 0.f;
      const int zcols = p.z_cols;
      const int dir = p.z_dir0 + n / zcols;
      const int nl = n % zcols;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if ((ni * 2 + mi) * 4 + q < LO || (ni * 2 + mi) * 4 + q >= HI) continue;
          const int mq = m0 + wm * 64 + mi * 32 + 8 * q + 4 * kh;  // first of 4 consecutive rows
          if (mq >= p.M) continue;
          const int t = mq / p.BP;
          const int b = mq - t * p.BP;  // multiple of 4: one 4-row group of lstm.hip
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ADD_SHIFT ? acc[mi][ni][4 * q + r] + sh : acc[mi][ni][4 * q + r];
          ZGroup(p, t, b).store(p, dir, nl, v);
        }
      }
    }
  }
}

template <bool LIFT, bool ZOUT, bool RES>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) float lds[4 * TILE_F];  // A0 A1 B0 B1
  float* const As = lds;
  float* const Bs = lds + 2 * TILE_F;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int kq = tid & 7;
  const int lr = tid >> 3;  // 0..31
  const int li = lane & 31;
  const int kh = lane >> 5;

  // Persistent workgroups: the grid is 2 blocks per CU and every block walks tile ids
  // blockIdx.x, +gridDim.x, ... (gridDim.x is a multiple of 8, so a block stays on "its" XCD id%8).
  // XCD-aware tile order: all N-blocks of one M-block are consecutive ids on the same XCD, so the A
  // panel is fetched from HBM once and re-read from that XCD's L2.  The chunk pipeline (LDS double
  // buffer + loads in flight) runs straight across tile boundaries and the epilogue stores of tile i
  // drain underneath the MFMAs of tile i+1.
  const int nblocks_n = (p.N + GEMM_BN - 1) / GEMM_BN;
  const int mblocks = (p.M + GEMM_BM - 1) / GEMM_BM;
  const int total_ids = ((mblocks + 7) / 8) * 8 * nblocks_n;
  const int nk = p.K / GEMM_BK;

  auto tile_of = [&](int id, int& m0, int& n0) -> bool {
    const int xcd = id & 7;
    const int slot = id >> 3;
    const int mblk = (slot / nblocks_n) * 8 + xcd;
    m0 = mblk * GEMM_BM;
    n0 = (slot % nblocks_n) * GEMM_BN;
    return mblk < mblocks;
  };

  // ---- loader state machine (runs up to two chunks ahead of the MFMA side)
  int l_id = (int)blockIdx.x - (int)gridDim.x;  // tile id being loaded
  int l_kc = nk;                                // chunk within that tile (nk => advance to next tile)
  bool l_done = false;
  int rb[4], rt[4];
  bool rvalid[4];
  int seg = -1, seg_left = 0, kk = 0, cin = 0;  // kk: channel of this thread's float4 within the segment
  const float* aptr[4];                        // per-row source pointer of the current segment (clamped)
  bool aok[4];
  float axv[4];                                // LIFT: signal sample of the row for the current tap
  const float* bptr = p.Wt;
  const long bstep = (long)32 * p.K;
  f32x4 ra[4], rbv[4];
  bool rmask[4];   // validity of the staged rows, applied when the chunk is written to LDS so the
  f32x4 la, lb;    // global loads stay in flight across the MFMA phase

  auto next_tile = [&]() {
    int m0 = 0, n0 = 0;
    do {
      l_id += gridDim.x;
      if (l_id >= total_ids) {
        l_done = true;
        return;
      }
    } while (!tile_of(l_id, m0, n0));
    l_kc = 0;
    seg = -1;
    seg_left = 0;
    bptr = p.Wt + (long)(n0 + lr) * p.K + 4 * kq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + lr + 32 * i;
      bool v = m < p.M;
      int b, t;
      if (ZOUT) {  // LSTM projection: m = t*BP + b
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
  };

  auto next_segment = [&]() {
    ++seg;
    const GemmSeg& sg = p.seg[seg];
    seg_left = sg.kpad / GEMM_BK;
    kk = 4 * kq;
    cin = sg.cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int in_t = rt[i] * sg.stride + sg.shift;
      const bool ok = rvalid[i] && in_t >= 0 && in_t < sg.w_in;
      aok[i] = ok;
      if (LIFT) {
        axv[i] = p.sig[ok ? ((long)rb[i] * p.L + in_t) : 0];
      } else {
        const long row = sg.time_major ? ((long)in_t * p.BP + rb[i]) : ((long)rb[i] * sg.w_in + in_t);
        aptr[i] = sg.src + (ok ? row * sg.lda : 0) + sg.col0 + 4 * kq;
      }
    }
  };

  // returns false when this block has no more chunks to load
  auto load_chunk = [&]() -> bool {
    if (l_kc == nk) next_tile();
    if (l_done) return false;
    if (seg_left == 0) next_segment();
    const bool kin = kk < cin;
    if (LIFT) {
      const int kc = kin ? kk : 0;
      la = *reinterpret_cast<const f32x4*>(p.lift_a + kc);
      lb = *reinterpret_cast<const f32x4*>(p.lift_b + kc);
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i][0] = axv[i];
    } else {
      const int ko = kin ? kk - 4 * kq : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(aptr[i] + ko);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) rmask[i] = aok[i] && kin;
#pragma unroll
    for (int i = 0; i < 4; ++i) rbv[i] = *reinterpret_cast<const f32x4*>(bptr + i * bstep);
    bptr += GEMM_BK;
    kk += GEMM_BK;
    --seg_left;
    ++l_kc;
    return true;
  };

  auto store_chunk = [&](int buf) {
    float* a = As + buf * TILE_F + lr * LDS_LD + 4 * kq;
    float* b = Bs + buf * TILE_F + lr * LDS_LD + 4 * kq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v;
      if (LIFT) {
        const float x = ra[i][0];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rmask[i] ? fmaxf(fmaf(x, la[j], lb[j]), 0.f) : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rmask[i] ? ra[i][j] : 0.f;
      }
      *reinterpret_cast<f32x4*>(a + 32 * i * LDS_LD) = v;
      *reinterpret_cast<f32x4*>(b + 32 * i * LDS_LD) = rbv[i];
    }
  };

  // ---- pipeline prologue: chunk 0 -> LDS buffer 0, chunk 1 in flight
  bool staged = load_chunk();   // registers hold a chunk not yet written to LDS
  if (!staged) return;
  store_chunk(0);
  staged = load_chunk();
  int buf = 0;

  for (int c_id = blockIdx.x; c_id < total_ids; c_id += gridDim.x) {
    int m0, n0;
    if (!tile_of(c_id, m0, n0)) continue;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    for (int kc = 0; kc < nk; ++kc) {
      __syncthreads();  // LDS buffer `buf` complete; every wave is done reading buffer buf^1
      const float* a0 = As + buf * TILE_F + (wm * 64 + li) * LDS_LD + 4 * kh;
      const float* b0 = Bs + buf * TILE_F + (wn * 64 + li) * LDS_LD + 4 * kh;
      f32x4 a[4][2], b[4][2];
      // The first fragment group is fetched BEFORE the next chunk is staged: the LDS queue serves it
      // first, so the matrix pipe restarts ~one ds_read latency after the barrier while the eight
      // ds_write_b128 + eight global loads of the staging drain underneath the MFMAs.
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[0][mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * LDS_LD);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[0][ni] = *reinterpret_cast<const f32x4*>(b0 + ni * 32 * LDS_LD);
      __builtin_amdgcn_sched_barrier(0);
      if (staged) {
        store_chunk(buf ^ 1);
        staged = load_chunk();
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 1; g < 4; ++g) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[g][mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * LDS_LD + 8 * g);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[g][ni] = *reinterpret_cast<const f32x4*>(b0 + ni * 32 * LDS_LD + 8 * g);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = ZOUT ? __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][mi][j], b[g][ni][j], acc[mi][ni], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x2f32(b[g][ni][j], a[g][mi][j], acc[mi][ni], 0, 0, 0);
      }
      buf ^= 1;
    }

    gemm_epilogue<ZOUT, RES>(p, acc, m0, n0, wm, wn, li, kh);
  }
}

// ---------------------------------------------------------------------------------------------
// DMA-staged variant (every launch whose A operand is a tensor in HBM):
// buffer_load_dwordx4 ... lds copies 16 bytes per lane straight from HBM/L2 into LDS -- no staging
// VGPRs, no ds_write, no masking VALU.  That matters more than usual here: v_mfma_f32_* runs at the
// fp32 VECTOR rate and does not overlap VALU work issued on the same SIMD (tools/ubench), so every
// VALU instruction in the K loop is paid in matrix-pipe time.
//   * LDS image: [rows][32 floats] unpadded (the DMA destination must be lane-linear); the 16-byte
//     k-slot of a row is XOR-swizzled with (row>>1)&7 on the SOURCE side and on the fragment reads, which
//     makes the ds_read_b128 of 16 different rows conflict-free.
//   * zero padding (conv edges, rows past M, K tails) is an out-of-range buffer offset (reads zeros) instead of a select.
//   * one barrier per chunk; the DMA of chunk k+1 is issued right after barrier k and has a whole
//     chunk of MFMAs to land (hipcc drains vmcnt before the next barrier because an LDS-DMA is pending).
// ---------------------------------------------------------------------------------------------
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;
constexpr int DTILE_F = GEMM_BM * GEMM_BK;  // 4096 floats = 16 KB per operand tile

constexpr int DMA_MAX_N = 1024;  // output columns whose shift fits the LDS table
constexpr int DMA_PAD_F = 256;  // floats in front of the tiles: the instruction offset (<= 896 B) is also
                                 // added to the LDS address and is compensated in M0, which must stay >= 0

// The DMA instruction is the RAW-BUFFER form (buffer_load_dwordx4 ... lds): resource descriptor in SGPRs + a 32-bit
// byte offset per lane.  It costs the issuing wave about half of what global_load_lds_dwordx4 with 64-bit per-lane
// addresses does (tools/ubench/mfma_shape_vmem.hip: 8 per 4096 MFMA-cycles: +3.6 % vs +10 %), and an offset past
// num_records reads zeros -- which is exactly what SAME padding, rows past M and K tails need, so no zero page.
constexpr unsigned DMA_OOB = 0xFFFF0000u;      // >= num_records with any immediate offset added: reads zeros
constexpr unsigned DMA_RECORDS = 0xFFFE0000u;  // every tensor of the engine is smaller than this many bytes
struct DmaSrc {
  unsigned aoff[4];  // per piece: byte offset (from the segment's tensor) of this lane's 16 bytes for chunk 0, or DMA_OOB
  unsigned boff[5];  // same for the weights (5 pieces of 8 rows per wave in the 160-column projection layout)
  __amdgpu_buffer_rsrc_t ra, rb;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, DMA_RECORDS, 0x00027000);
}

// Issue one chunk.  The chunk index C inside the segment is a compile-time constant and travels in the
// instruction's immediate offset, so stepping through a segment costs NO address arithmetic at all: the
// per-lane offsets stay fixed for the whole segment.
template <int C>
__device__ __forceinline__ void dma_piece(const DmaSrc& L, int q, float* a_dst, float* b_dst) {
  if (q < 4)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(L.ra, (lptr_t)(a_dst + q * 256 - C * GEMM_BK), 16, L.aoff[q], 0, C * GEMM_BK * 4, 0);
  else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(L.rb, (lptr_t)(b_dst + (q - 4) * 256 - C * GEMM_BK), 16, L.boff[q - 4], 0,
                                             C * GEMM_BK * 4, 0);
}
template <int C, int NP>
__device__ __forceinline__ void dma_issue(const DmaSrc& L, float* a_dst, float* b_dst) {
#pragma unroll
  for (int q = 0; q < NP; ++q) dma_piece<C>(L, q, a_dst, b_dst);
}

// CPS = chunks per K-segment (every segment of a launch has the same width), TAIL = the segment's channel
// count is not a multiple of 32 (its last chunk selects per lane against the K tail).
// MODE 0: fp32 operands (v_mfma_f32_32x32x2_f32).  1: IEEE halves (v_mfma_f32_32x32x16_f16).  2: fp32 values carried as
// exact hi + lo half pairs ("split" format), three f16 MFMAs per product term set: hi*hi + hi*lo + lo*hi, fp32 accumulate.
// ONESEG: the launch has a single K-segment (every fp32 projection of the DNA stack): known at compile time, the per-row
// (batch, frame, valid) triples of a tile die as soon as its DMA offsets are built instead of living through the tile for a
// next segment that never comes -- part of getting the projection under 176 registers.
template <bool ZOUT, bool RES, int CPS, bool TAIL, int MODE = 0, bool ONESEG = false>
__global__ __launch_bounds__(256, 2) void gemm_f32_dma_kernel(const GemmParams p) {
  constexpr bool F16 = MODE == 1;
  constexpr bool SPLIT = MODE == 2;
  // Wave layout.  Convolutions: 128 x 128 tile, waves 2 x 2, each 64 x 64 (2 x 2 accumulators).  Projections (ZOUT):
  // 128 x 160 tile, waves 4 x 1, each 32 rows x 160 columns (1 x 5 accumulators) -- N = 8H = 800 is then five whole
  // tiles instead of 6.25 tiles of 128.
  constexpr int BN = ZOUT ? 160 : GEMM_BN;      // tile columns = rows of the weight tile
  constexpr int NMI = ZOUT ? 1 : 2, NNI = ZOUT ? 5 : 2;
  constexpr int NBP = BN / 32;                  // DMA pieces (8 rows each) of the weight tile per wave and chunk
  constexpr int NP = 4 + NBP;                   // DMA instructions per wave and chunk
  constexpr int BTILE_F = BN * GEMM_BK;         // floats per weight tile
  __shared__ __attribute__((aligned(16))) float lds[DMA_PAD_F + 2 * DTILE_F + 2 * BTILE_F + DMA_MAX_N];  // pad A0 A1 B0 B1 shift
  float* const As = lds + DMA_PAD_F;
  float* const Bs = As + 2 * DTILE_F;
  float* const shl = Bs + 2 * BTILE_F;  // per-column shift (folded BN offset / LSTM bias), zero past N

  const int tid = threadIdx.x;
  // The tiles are only ever written by the DMA engine, which the optimiser does not see as a store to `lds`;
  // one (never taken) ordinary store keeps it from reasoning about a never-written array.
  if (p.K < 0) lds[tid] = 0.f;
  for (int n = tid; n < DMA_MAX_N; n += 256) shl[n] = n < p.N ? p.shift[n] : 0.f;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = ZOUT ? wave : wave >> 1, wn = ZOUT ? 0 : wave & 1;
  const int li = lane & 31;
  const int kh = lane >> 5;

  const int nblocks_n = (p.N + BN - 1) / BN;
  const int mblocks = (p.M + GEMM_BM - 1) / GEMM_BM;
  const int total_ids = ((mblocks + 7) / 8) * 8 * nblocks_n;
  const int nseg = ONESEG ? 1 : p.nseg;

  auto tile_of = [&](int id, int& m0, int& n0) -> bool {
    const int xcd = id & 7;
    const int slot = id >> 3;
    const int mblk = (slot / nblocks_n) * 8 + xcd;
    m0 = mblk * GEMM_BM;
    n0 = (slot % nblocks_n) * BN;
    return mblk < mblocks;
  };
  auto next_valid = [&](int id) -> int {  // next tile id of this workgroup after `id`, or total_ids
    int m0, n0;
    do id += gridDim.x;
    while (id < total_ids && !tile_of(id, m0, n0));
    return id;
  };

  // ---- DMA geometry: piece j (0..3) of wave w covers tile rows w*32 + j*8 .. +7; lane -> (row, slot).  The physical
  //      16-byte slot of logical slot s in tile row r is s ^ ((r >> 1) & 7).
  const int drow = wave * 32 + (lane >> 3);           // + 8*j
  int dslot[4];                                       // logical k-slot fetched into physical slot lane&7
#pragma unroll
  for (int j = 0; j < 4; ++j) dslot[j] = ((lane & 7) ^ ((j * 4 + (lane >> 4)) & 7)) * 4;  // in floats
  //      weight tile: wave w covers its rows w*BN/4 + j*8 .. +7, j < NBP (BN/4 = 32: same slots as A; 40: shifted by 4w)
  const int bdrow = wave * (BN / 4) + (lane >> 3);    // + 8*j
  int bslot[NBP];
#pragma unroll
  for (int j = 0; j < NBP; ++j) bslot[j] = ((lane & 7) ^ (((wave * (BN / 4) + j * 8 + (lane >> 3)) >> 1) & 7)) * 4;
  // fragment reads: physical slot of logical slot (2g+kh) for this lane's row
  int fslot[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) fslot[g] = ((2 * g + kh) ^ ((li >> 1) & 7)) * 4;

  // ---- loader: always exactly one chunk ahead of the MFMA loop
  DmaSrc L;
  L.rb = dma_rsrc(p.Wt);
  L.ra = L.rb;
  int rb[4], rt[4];
  bool rvalid[4];
  unsigned brow[NBP];
  auto load_tile = [&](int id) {
    int m0, n0;
    tile_of(id, m0, n0);
    const RowSplit rs(m0, ZOUT ? p.BP : p.T_out);
#pragma unroll
    for (int j = 0; j < NBP; ++j) brow[j] = (unsigned)((n0 + bdrow + 8 * j) * p.K + bslot[j]) * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + drow + 8 * j;
      bool v = m < p.M;
      int b, t;
      if (ZOUT) {
        rs.split(drow + 8 * j, t, b);
        v = v && (b < p.B);
      } else {
        rs.split(drow + 8 * j, b, t);
      }
      rb[j] = b;
      rt[j] = t;
      rvalid[j] = v;
    }
  };
  auto load_segment = [&](const GemmSeg& sg, int k0) {
    L.ra = dma_rsrc(sg.src);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int in_t = rt[j] * sg.stride + sg.shift;
      const bool ok = rvalid[j] && in_t >= 0 && in_t < sg.w_in;
      const long row = sg.time_major ? ((long)in_t * p.BP + rb[j]) : ((long)rb[j] * sg.w_in + in_t);
      L.aoff[j] = ok ? (unsigned)((row * sg.lda + sg.col0 + dslot[j]) * 4) : DMA_OOB;
    }
#pragma unroll
    for (int j = 0; j < NBP; ++j) L.boff[j] = brow[j] + (unsigned)k0 * 4u;
  };
  // chunk c of a segment whose channel count is not a multiple of 32: per-lane select against the K tail
  auto tail_piece = [&](int c, int cin, int q, float* a_dst, float* b_dst) {
    if (q < 4) {
      const bool ok = L.aoff[q] != DMA_OOB && (c * GEMM_BK + dslot[q] < cin);
      const unsigned off = ok ? L.aoff[q] + (unsigned)c * GEMM_BK * 4u : DMA_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(L.ra, (lptr_t)(a_dst + q * 256), 16, off, 0, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(L.rb, (lptr_t)(b_dst + (q - 4) * 256), 16, L.boff[q - 4] + (unsigned)c * GEMM_BK * 4u, 0, 0, 0);
    }
  };
  auto dma_issue_tail = [&](int c, int cin, float* a_dst, float* b_dst) {
#pragma unroll
    for (int q = 0; q < NP; ++q) tail_piece(c, cin, q, a_dst, b_dst);
  };

  // ---- tile ids: static shares (stride gridDim.x) or, with counters, the next slot of this XCD (GemmParams::tile_ctr)
  __shared__ int sh_next;
  const bool dyn = p.tile_ctr != nullptr && CPS >= 3;   // the number is fetched in chunk 0, published in chunk 1, read in chunk 2
  const int my_xcd = blockIdx.x & 7;
  const unsigned slots_per_xcd = (unsigned)(total_ids >> 3);
  unsigned long long grabbed = 0;   // thread 0: the counter value behind the next tile
  auto issue_grab = [&]() {
    if (tid == 0) grabbed = atomicAdd(p.tile_ctr + my_xcd, 1ull);
  };
  auto decode_grab = [&]() -> int {   // thread 0: the tile id behind `grabbed` (slots past the last row block are skipped)
    for (;;) {
      const unsigned long long v = grabbed - p.tile_base;
      if (v >= slots_per_xcd) return total_ids;
      const int id = (int)((unsigned)v << 3) | my_xcd;
      int mm, nn;
      if (tile_of(id, mm, nn)) return id;
      grabbed = atomicAdd(p.tile_ctr + my_xcd, 1ull);
    }
  };
  int c_id = blockIdx.x;
  if (dyn) {
    issue_grab();
    if (tid == 0) sh_next = decode_grab();
    __syncthreads();
    c_id = sh_next;
    if (c_id >= total_ids) return;
  } else {
    int m0, n0;
    if (c_id >= total_ids) return;
    if (!tile_of(c_id, m0, n0)) c_id = next_valid(c_id);
    if (c_id >= total_ids) return;
  }
  load_tile(c_id);
  load_segment(p.seg[0], 0);
  if (TAIL && !SPLIT && CPS == 1)
    dma_issue_tail(0, p.seg[0].cin, As + wave * 1024, Bs + wave * (BTILE_F / 4));
  else
    dma_issue<0, NP>(L, As + wave * 1024, Bs + wave * (BTILE_F / 4));
  int buf = 0;
  bool have_prev = false;
  int pm0 = 0, pn0 = 0;
  f32x16 acc[NMI][NNI];
  const bool relu = p.relu != 0;
  // real work of the last chunk of a segment: 8-column groups (fp32 / f16 units), or 16-element k-steps (split)
  const int tail_groups = TAIL ? (p.seg[0].cin - (CPS - 1) * GEMM_BK + (SPLIT ? 15 : 7)) / (SPLIT ? 16 : 8) : 4;
  auto epilogue = [&](int em0, int en0) {
#if (CHIRON_SENS & 32) && defined(__HIP_DEVICE_COMPILE__)
    if (ZOUT) {   // timing experiment: the projection without its epilogue (the asm keeps every accumulator, and so every MFMA, alive)
#pragma unroll
      for (int ni = 0; ni < NNI; ++ni) asm volatile("" ::"v"(acc[0][ni]));
      return;
    }
#endif
#if (CHIRON_SENS & 64) && defined(__HIP_DEVICE_COMPILE__)
    if (!ZOUT) {  // timing experiment: the convolution form without its epilogue
#pragma unroll
      for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NNI; ++ni) asm volatile("" ::"v"(acc[mi][ni]));
      return;
    }
#endif
    if constexpr (SPLIT) {
      // the weight rows (and the shift the accumulators started from) carry a power of two: take it out, exactly (GemmParams::descale)
      if (p.descale != nullptr) {
#pragma unroll
        for (int ni = 0; ni < NNI; ++ni) {
          if constexpr (ZOUT) {
            const float ds = p.descale[en0 + ni * 32 + li];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][ni][r] *= ds;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 ds = *reinterpret_cast<const f32x4*>(p.descale + en0 + wn * 64 + ni * 32 + 8 * q + 4 * kh);
#pragma unroll
              for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ni][4 * q + r] *= ds[r];
            }
          }
        }
      }
    }
    if constexpr (ZOUT) {
      gemm_epilogue_zrows<NNI>(p, acc, em0, en0, wave, li, kh);
    } else {
      if (F16 || SPLIT || en0 + GEMM_BN <= p.N) {  // f16 / split conv: N % 128 == 0 (launch_gemm)
        if (relu)
          gemm_epilogue_lean<RES, true, MODE>(p, acc, em0, en0, wm, wn, li, kh);
        else
          gemm_epilogue_lean<RES, false, MODE>(p, acc, em0, en0, wm, wn, li, kh);
      } else {
        gemm_epilogue<false, RES, 0, 16, false>(p, acc, em0, en0, wm, wn, li, kh);
      }
    }
  };

  // Every non-MFMA instruction in this loop is paid in matrix-pipe time (tools/ubench/mfma_vmem.hip: ~11 pipe
  // cycles per VALU instruction with two waves per SIMD), so the per-tile bookkeeping is kept off the VALU:
  // the accumulators are never zeroed and the shift is never added -- the first MFMA of a tile takes its C
  // operand from a register vector holding the shift (read from LDS), D = acc.
#if CHIRON_SENS & 1024
  // timing experiment: where a wave's cycles go (s_memtime): the barrier in front of each chunk of a tile's first segment, the
  // epilogue, whole tiles; wave 0 of four workgroups prints (tools/gemm_clk.py condenses)
  unsigned long long clk_bar[8] = {0, 0, 0, 0, 0, 0, 0, 0}, clk_epi = 0, clk_tiles = 0;
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime();
#endif
  while (c_id < total_ids) {
    int m0, n0;
    tile_of(c_id, m0, n0);
    int n_id = dyn ? total_ids : next_valid(c_id);
    int k0 = 0;

    auto segment = [&](auto first_tag, int sgi) {
      constexpr bool FS = decltype(first_tag)::value;  // first K-segment of the tile
      k0 += CPS * GEMM_BK;
      const bool last_seg = sgi + 1 == nseg;

      // one K-chunk: barrier, start the DMA of the chunk after it, run this one on the matrix pipe
      auto chunk = [&](auto cc) {
        constexpr int C = decltype(cc)::value;
#if CHIRON_SENS & 1024
        const unsigned long long clk_b0 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();  // chunk in `buf` has landed (vmcnt drained before the barrier); buf^1 is free
#if CHIRON_SENS & 1024
        if (FS) clk_bar[C] += __builtin_amdgcn_s_memtime() - clk_b0;
#endif
        if (FS && dyn) {   // the next tile's number: requested, published one barrier later (it has returned by then), read after the next
          if (C == 0) issue_grab();
          if (C == 1 && tid == 0) sh_next = decode_grab();
          if (C == 2) n_id = sh_next;
        }
        float* a_dst = As + (buf ^ 1) * DTILE_F + wave * 1024;  // wave-uniform bases; lanes land at +16 B each
        float* b_dst = Bs + (buf ^ 1) * BTILE_F + wave * (BTILE_F / 4);
        // The NP (8 or 9) DMA instructions of the next chunk are interleaved with the first MFMAs below instead of
        // being issued in a burst (all eight waves of the CU share one texture-address unit).
        bool go = true;
        if (C + 1 == CPS) {  // first chunk of the next segment / of the next tile: new per-lane pointers
          int nsg = sgi + 1, nk0 = k0;
          if (last_seg) {
            nsg = 0;
            nk0 = 0;
            go = n_id < total_ids;
            if (go) load_tile(n_id);
          }
          if (go) load_segment(p.seg[nsg], nk0);
        }
        auto piece = [&](int q) {
          if (!go) return;
          if (C + 1 < CPS) {
            if (TAIL && !SPLIT && C + 2 == CPS)  // split rows are stored zero padded to whole blocks: nothing to mask
              tail_piece(C + 1, p.seg[sgi].cin, q, a_dst, b_dst);
            else
              dma_piece<(C + 1 < CPS ? C + 1 : 0)>(L, q, a_dst, b_dst);
          } else {
            if (TAIL && !SPLIT && CPS == 1)
              tail_piece(0, p.seg[0].cin, q, a_dst, b_dst);
            else
              dma_piece<0>(L, q, a_dst, b_dst);
          }
        };
        if (FS && C == 0) {
          // The previous tile's epilogue is issued HERE, after this tile's first barrier: the barrier's
          // vmcnt(0) then only ever waits for DMA issued a whole chunk earlier, never for fresh stores.
          // (Round 3: BEFORE the shift vectors below are built, so that these take the registers the epilogue has just
          // released -- built first, the projection's 5 x 16 of them sat next to its 80 accumulators: 226 .. 237 registers,
          // and no such wave fits next to two recurrence waves on a SIMD; now it does.)
#if CHIRON_SENS & 1024
          const unsigned long long clk_e0 = __builtin_amdgcn_s_memtime();
#endif
          if (have_prev) epilogue(pm0, pn0);
          asm volatile("" ::: "memory");
#if CHIRON_SENS & 1024
          clk_epi += __builtin_amdgcn_s_memtime() - clk_e0;
#endif
          // C operand of the tile's first MFMAs: this lane's 16 shift values per 32-column block, written INTO the accumulators
          // (a separate vector is not coalesced with them by the register allocator: 80 more live registers in the projection)
#pragma unroll
          for (int ni = 0; ni < NNI; ++ni) {
            if (ZOUT) {
              const float sh = shl[n0 + ni * 32 + li];
#pragma unroll
              for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = sh;
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 sh = *reinterpret_cast<const f32x4*>(shl + n0 + wn * 64 + ni * 32 + 8 * q + 4 * kh);
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                  for (int r = 0; r < 4; ++r) acc[mi][ni][4 * q + r] = sh[r];
              }
            }
          }
        }
        {
          const float* a0 = As + buf * DTILE_F + (wm * (NMI * 32) + li) * GEMM_BK;
          const float* b0 = Bs + buf * BTILE_F + (wn * (NNI * 32) + li) * GEMM_BK;
          // fragment slot of k-group g: fslot[g] = fslot[0] ^ 8 g (floats).  Derived per chunk from ONE opaque register: hoisted,
          // the eight (operand, g) addresses of both buffers are sixteen registers held for the whole kernel, and the projection
          // has to come down to 176 to fit next to two recurrence waves on a SIMD
          int fs0 = fslot[0];
          asm volatile("" : "+v"(fs0));
          auto fsl = [&](int g) -> int { return fs0 ^ (8 * g); };
          if (SPLIT) {
            // A 128-byte row chunk = 32 elements: slots 0-3 hold the hi halves of elements 0-7 / 8-15 / 16-23 / 24-31,
            // slots 4-7 the lo halves.  k-step s (16 elements) uses hi slot 2s+kh and lo slot 4+2s+kh, i.e. the fp32
            // kernel's fragment addresses fslot[s] and fslot[s+2].
#pragma unroll
            for (int st = 0; st < 2; ++st) {
#pragma unroll
              for (int j = 0; j < (st == 1 ? NP - 4 : 4); ++j) piece(st * 4 + j);
              if (TAIL && C + 1 == CPS && st >= tail_groups) continue;  // tail_groups counts 16-element k-steps here
              f32x4 ah[NMI], al[NMI], bh[NNI], bl[NNI];
#pragma unroll
              for (int mi = 0; mi < NMI; ++mi) {
                ah[mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * GEMM_BK + fsl(st));
                al[mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * GEMM_BK + fsl(st + 2));
              }
#pragma unroll
              for (int ni = 0; ni < NNI; ++ni) {
                bh[ni] = *reinterpret_cast<const f32x4*>(b0 + ni * 32 * GEMM_BK + fsl(st));
                bl[ni] = *reinterpret_cast<const f32x4*>(b0 + ni * 32 * GEMM_BK + fsl(st + 2));
              }
#pragma unroll
              for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NNI; ++ni) {
                  const f16x8 xah = __builtin_bit_cast(f16x8, ah[mi]), xal = __builtin_bit_cast(f16x8, al[mi]);
                  const f16x8 xbh = __builtin_bit_cast(f16x8, bh[ni]), xbl = __builtin_bit_cast(f16x8, bl[ni]);
                  f32x16 c = acc[mi][ni];
                  if (ZOUT) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(xal, xbh, c, 0, 0, 0);   // small terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbh, c, 0, 0, 0);
                  } else {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(xbh, xal, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(xbl, xah, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(xbh, xah, c, 0, 0, 0);
                  }
                  acc[mi][ni] = c;
                }
            }
          } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            // K tail: the last chunk of a segment holds only tail_groups * 8 real columns, the rest multiplies zeros
            if (TAIL && C + 1 == CPS && g >= tail_groups) {  // nothing but zeros to multiply: only the DMA pieces this group carries
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (g * 4 + j < NP) piece(g * 4 + j);
              continue;
            }
            f32x4 a[NMI], b[NNI];
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * GEMM_BK + fsl(g));
#pragma unroll
            for (int ni = 0; ni < NNI; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(b0 + ni * 32 * GEMM_BK + fsl(g));
            if (F16) {
              // 16 bytes = the 8 halves of one k-step of v_mfma_f32_32x32x16_f16 (lanes 0-31: k 0-7, lanes 32-63: k 8-15)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (g * 4 + j < NP) piece(g * 4 + j);
#pragma unroll
              for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NNI; ++ni) {
                  const f32x16 c = acc[mi][ni];
                  const f16x8 ah = __builtin_bit_cast(f16x8, a[mi]), bh = __builtin_bit_cast(f16x8, b[ni]);
                  acc[mi][ni] = ZOUT ? __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, c, 0, 0, 0);
                }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (g * 4 + j < NP) piece(g * 4 + j);
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
                  for (int ni = 0; ni < NNI; ++ni) {
                    const f32x16 c = acc[mi][ni];
#if CHIRON_SENS & 16
                    if (j & 1) { acc[mi][ni][0] += a[mi][j] * b[ni][j]; continue; }   // timing experiment: half of the MFMAs
#endif
                    acc[mi][ni] = ZOUT ? __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], c, 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni][j], a[mi][j], c, 0, 0, 0);
                  }
              }
#if CHIRON_SENS & 4096
              // timing experiment (with bit 32: no epilogue): what a second accumulator set would buy -- the previous tile's 20 stores
              // trickle through this tile's main loop, one per k-group, instead of standing between two tiles (the data is a fragment
              // register: garbage, same instruction stream and the same z traffic)
              if (ZOUT && FS && have_prev && C * 4 + g < 4 * NNI) {
                const int sq = (C * 4 + g) / NNI, sni = (C * 4 + g) % NNI;
                const unsigned gstr = (unsigned)p.z_dirs_total * p.z_cols * 4;
                float* o = p.out + (unsigned)((pm0 + wave * 32 + 8 * sq + 4 * kh) >> 2) * gstr + (unsigned)(p.z_dir0 * p.z_cols + pn0 + li) * 4 + sni * 128;
                if (pm0 + wave * 32 + 8 * sq + 4 * kh < p.M) *reinterpret_cast<f32x4*>(o) = a[0];
              }
#endif
            }
          }
          }
        }
        buf ^= 1;
      };
      chunk(std::integral_constant<int, 0>{});
      if (CPS > 1) chunk(std::integral_constant<int, 1>{});
      if (CPS > 2) chunk(std::integral_constant<int, 2>{});
      if (CPS > 3) chunk(std::integral_constant<int, 3>{});
      if (CPS > 4) chunk(std::integral_constant<int, 4>{});
      if (CPS > 5) chunk(std::integral_constant<int, 5>{});
      if (CPS > 6) chunk(std::integral_constant<int, 6>{});
      if (CPS > 7) chunk(std::integral_constant<int, 7>{});
    };
    segment(std::true_type{}, 0);
    for (int sgi = 1; sgi < nseg; ++sgi) segment(std::false_type{}, sgi);

    have_prev = true;
    pm0 = m0;
    pn0 = n0;
    c_id = n_id;
#if CHIRON_SENS & 1024
    ++clk_tiles;
#endif
  }
  if (have_prev) epilogue(pm0, pn0);
#if CHIRON_SENS & 1024
  if (tid == 0 && blockIdx.x < 4 && clk_tiles > 0)
    printf("gemm_clk zout %d K %d block %d: tiles %llu, cycles per tile: total %llu, epilogue %llu, barriers %llu %llu %llu %llu %llu %llu %llu %llu\n",
           (int)ZOUT, p.K, (int)blockIdx.x, clk_tiles, (__builtin_amdgcn_s_memtime() - clk_t0) / clk_tiles, clk_epi / clk_tiles, clk_bar[0] / clk_tiles,
           clk_bar[1] / clk_tiles, clk_bar[2] / clk_tiles, clk_bar[3] / clk_tiles, clk_bar[4] / clk_tiles, clk_bar[5] / clk_tiles, clk_bar[6] / clk_tiles,
           clk_bar[7] / clk_tiles);
#endif
}

// The unrolled DMA loop is instantiated for the K-segment widths of the shipped topologies (256 channels;
// LSTM inputs of 200 = 2H and 100 = H); anything else takes the register-staged kernel.
template <bool ZOUT, bool RES>
static bool launch_dma(const GemmParams& p, dim3 grid, dim3 block, hipStream_t stream) {
  const int kpad = p.seg[0].kpad, cin = p.seg[0].cin;
  constexpr int BN = ZOUT ? 160 : GEMM_BN;
  if (((p.N + BN - 1) / BN) * BN > DMA_MAX_N) return false;
  if (ZOUT && p.z_cols < BN) return false;  // a projection tile may span at most two directions
  for (int s = 1; s < p.nseg; ++s)
    if (p.seg[s].kpad != kpad || p.seg[s].cin != cin) return false;
  const bool tail = cin != kpad;
  if (p.f16 == 2) {
    // split format: 4-byte units = elements, rows stored zero padded to whole 32-element blocks
    if (!ZOUT && p.N % GEMM_BN) return false;
    if constexpr (ZOUT) {
      if (p.nseg == 1 && kpad == 256 && !tail) {
        hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 8, false, 2, true>), grid, block, 0, stream, p);
        return true;
      }
      if (p.nseg == 1 && kpad == 224 && tail) {
        hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 7, true, 2, true>), grid, block, 0, stream, p);
        return true;
      }
    }
    if (kpad == 256 && !tail) {
      hipLaunchKernelGGL((gemm_f32_dma_kernel<ZOUT, RES, 8, false, 2>), grid, block, 0, stream, p);
      return true;
    }
    if constexpr (ZOUT) {
      if (kpad == 224 && tail) {
        hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 7, true, 2>), grid, block, 0, stream, p);
        return true;
      }
      if (kpad == 128 && tail) {
        hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 4, true, 2>), grid, block, 0, stream, p);
        return true;
      }
    }
    return false;
  }
  if (p.f16) {
    // 4-byte units: 256 halves = 128 units = 4 chunks; LSTM inputs of 200 / 100 halves are padded to 256 / 128 halves
    if (!ZOUT && p.N % GEMM_BN) return false;
    if (kpad == 128 && !tail) {
      hipLaunchKernelGGL((gemm_f32_dma_kernel<ZOUT, RES, 4, false, 1>), grid, block, 0, stream, p);
      return true;
    }
    if constexpr (ZOUT) {
      if (kpad == 128 && tail) {
        hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 4, true, 1>), grid, block, 0, stream, p);
        return true;
      }
      if (kpad == 64 && tail) {
        hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 2, true, 1>), grid, block, 0, stream, p);
        return true;
      }
    }
    return false;
  }
  if constexpr (ZOUT) {
    if (p.nseg == 1) {
      if (kpad == 256 && !tail) {
        hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 8, false, 0, true>), grid, block, 0, stream, p);
        return true;
      }
      if (kpad == 224 && tail) {
        hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 7, true, 0, true>), grid, block, 0, stream, p);
        return true;
      }
    }
  }
  if (kpad == 256 && !tail) {
    hipLaunchKernelGGL((gemm_f32_dma_kernel<ZOUT, RES, 8, false>), grid, block, 0, stream, p);
    return true;
  }
  if constexpr (ZOUT) {
    if (kpad == 224 && tail) {
      hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 7, true>), grid, block, 0, stream, p);
      return true;
    }
    if (kpad == 128 && tail) {
      hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 4, true>), grid, block, 0, stream, p);
      return true;
    }
  }
  return false;
}

bool launch_gemm(const GemmParams& p0, hipStream_t stream) {
  GemmParams p = p0;
  const int nblocks_n = (p.N + GEMM_BN - 1) / GEMM_BN;
  const int mblocks = (p.M + GEMM_BM - 1) / GEMM_BM;
  const int total_ids = ((mblocks + 7) / 8) * 8 * nblocks_n;
  const int n_cu = current_device_cus();
  // Workgroups per CU of the persistent launches: two are resident at a time (69 KB LDS each; 77 KB for the projection layout).
  // With tiles by counter a launch may hold MORE workgroups than fit: the extra ones start when a CU frees up -- behind another
  // batch's recurrence or Winograd workgroup -- and take whatever tiles are left (or leave at once).  CHIRON_GEMM_WGS_PER_CU is
  // the A/B knob (default 2; only meaningful with dynamic tiles).
  static const int wgs_per_cu = [] {
    const char* v = getenv("CHIRON_GEMM_WGS_PER_CU");
    const int n = v ? atoi(v) : 2;
    return n >= 1 && n <= 8 ? n : 2;
  }();
  int g = wgs_per_cu * n_cu;
  g = (g / 8) * 8;
  if (g > total_ids) g = total_ids;
  const dim3 grid(g), block(256);
  // projections run 128 x 160 tiles on the DMA kernel
  const int total_z = ((mblocks + 7) / 8) * 8 * ((p.N + 159) / 160);
  const dim3 grid_z(std::min((wgs_per_cu * n_cu / 8) * 8, total_z));
  // dynamic tile numbers (GemmParams::tile_ctr): the DMA kernel uses them when a segment has three chunks or more; a launch
  // takes slots-per-XCD + workgroups-per-XCD numbers from every counter
  const bool zout = p.out_mode == 1;
  const bool dyn = p.tile_ctr != nullptr && p.tile_base_host != nullptr && p.seg[0].src != nullptr && p.seg[0].kpad / GEMM_BK >= 3 &&
                   (zout ? grid_z.x : grid.x) % 8 == 0;
  if (dyn) p.tile_base = *p.tile_base_host;
  else p.tile_ctr = nullptr;
  auto advance = [&](bool launched) {
    if (launched && dyn) *p.tile_base_host += (unsigned long long)((zout ? total_z : total_ids) / 8 + (zout ? grid_z.x : grid.x) / 8);
    return launched;
  };
  if (p.f16) {  // halves: only the DMA kernels exist
    if (p.seg[0].src == nullptr) return false;
    if (p.out_mode == 1) return advance(launch_dma<true, false>(p, grid_z, block, stream));
    if (p.res_a != nullptr) return advance(launch_dma<false, true>(p, grid, block, stream));
    return advance(launch_dma<false, false>(p, grid, block, stream));
  }
  if (p.seg[0].src == nullptr) {  // lifted signal: A is computed in the loader
    hipLaunchKernelGGL((gemm_f32_kernel<true, false, false>), grid, block, 0, stream, p);
  } else if (p.out_mode == 1) {
    if (!advance(launch_dma<true, false>(p, grid_z, block, stream)))
      hipLaunchKernelGGL((gemm_f32_kernel<false, true, false>), grid, block, 0, stream, p);
  } else if (p.res_a != nullptr) {
    if (!advance(launch_dma<false, true>(p, grid, block, stream)))
      hipLaunchKernelGGL((gemm_f32_kernel<false, false, true>), grid, block, 0, stream, p);
  } else {
    if (!advance(launch_dma<false, false>(p, grid, block, stream)))
      hipLaunchKernelGGL((gemm_f32_kernel<false, false, false>), grid, block, 0, stream, p);
  }
  return true;
}

}  // namespace chiron
