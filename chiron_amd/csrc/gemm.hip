// Fused conv / LSTM-projection GEMM for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak).
//
// Replaces the tf.nn.conv2d + batch_normalization + relu (+ add) chains of chiron/cnn.py:15-83,
// :234-262 and the x-part of the LSTMCell MatMul (rnn.py:49-65) of the reference.
//
//   block tile 128(M) x 128(N) x 32(K), 256 threads = 4 waves, each wave 64x64 = 2x2 MFMA tiles.
//   A and B tiles are register-staged into a DOUBLE-BUFFERED LDS image ([rows][36] floats:
//   conflict-free ds_read_b128 / ds_write_b128), one barrier per K-chunk: while chunk k is on the
//   matrix pipe, chunk k+1 is written to the other LDS buffer and the global loads of chunk k+2 are
//   in flight.  K order inside a chunk is permuted (lanes 0-31 take k=8g+j, lanes 32-63 take
//   k=8g+4+j) so every operand fetch is one ds_read_b128 feeding four MFMAs; A and B use the same
//   permutation.  The A loader is a small state machine over K-segments (conv taps / fused inputs):
//   per-row source pointers and validity are computed once per segment, every load is
//   unconditional from a clamped address and masked afterwards (no divergent branches in the loop).
#include "kernels.h"

namespace chiron {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LDS_LD = GEMM_BK + 4;  // 36 floats: 16B-aligned rows, conflict-free b128 access
constexpr int TILE_F = GEMM_BM * LDS_LD;

template <bool ZOUT, bool RES>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn,
                                              int li, int kh) {
    // ---- epilogue.  32x32 accumulator layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Conv launches accumulate C^T (operands swapped), so a lane owns ONE output row m and four
  // consecutive columns n per register quad: 16-byte stores instead of 4-byte ones (the 4-byte
  // epilogue was store-issue-bound: 64 store instructions per lane per tile).
  if (!ZOUT) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
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
          const int n = n0 + wn * 64 + ni * 32 + 8 * q + 4 * kh;  // 4 consecutive columns
          if (n >= p.N) continue;
          const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n);
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][4 * q + r] + sh[r];
          if (RES) {
            const f32x4 ra4 = *reinterpret_cast<const f32x4*>(p.res_a + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(sv, ra4[r], v[r]);
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
      const int n = n0 + wn * 64 + ni * 32 + li;
      if (n >= p.N) continue;
      const float sh = p.shift[n];
      const int nb4 = p.BP >> 2;
      const int zcols = p.z_cols;
      const int dir = p.z_dir0 + n / zcols;
      const int nl = n % zcols;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int mq = m0 + wm * 64 + mi * 32 + 8 * q + 4 * kh;  // first of 4 consecutive rows
          if (mq >= p.M) continue;
          const int t = mq / p.BP;
          const int b = mq - t * p.BP;  // multiple of 4: one 4-row group of lstm.hip
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][4 * q + r] + sh;
          *reinterpret_cast<f32x4*>(p.out + ((((long)t * nb4 + (b >> 2)) * p.z_dirs_total + dir) * zcols + nl) * 4) = v;
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
// DMA-staged variant (every launch whose A operand is a tensor in HBM, i.e. all but the lifted conv2b):
// global_load_lds_dwordx4 copies 16 bytes per lane straight from HBM/L2 into LDS -- no staging
// VGPRs, no ds_write, no masking VALU.  That matters more than usual here: v_mfma_f32_* runs at the
// fp32 VECTOR rate and does not overlap VALU work issued on the same SIMD (tools/ubench), so every
// VALU instruction in the K loop is paid in matrix-pipe time.
//   * LDS image: [rows][32 floats] unpadded (the DMA destination must be lane-linear); the 16-byte
//     k-slot of a row is XOR-swizzled with (row>>1)&7 on the SOURCE side and on the fragment reads, which
//     makes the ds_read_b128 of 16 different rows conflict-free.
//   * zero padding (conv edges, rows past M, K tails) is a pointer to a zero page instead of a select.
//   * one barrier per chunk; the DMA of chunk k+1 is issued right after barrier k and has a whole
//     chunk of MFMAs to land (hipcc drains vmcnt before the next barrier because an LDS-DMA is pending).
// ---------------------------------------------------------------------------------------------
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;
constexpr int DTILE_F = GEMM_BM * GEMM_BK;  // 4096 floats = 16 KB per operand tile

template <bool ZOUT, bool RES>
__global__ __launch_bounds__(256, 2) void gemm_f32_dma_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) float lds[4 * DTILE_F];  // A0 A1 B0 B1
  float* const As = lds;
  float* const Bs = lds + 2 * DTILE_F;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31;
  const int kh = lane >> 5;

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

  // ---- DMA geometry: piece j (0..3) of wave w covers tile rows w*32 + j*8 .. +7; lane -> (row, slot)
  const int drow = wave * 32 + (lane >> 3);           // + 8*j
  int dslot[4];                                       // logical k-slot fetched into physical slot lane&7
#pragma unroll
  for (int j = 0; j < 4; ++j) dslot[j] = ((lane & 7) ^ ((j * 4 + (lane >> 4)) & 7)) * 4;  // in floats
  // fragment reads: physical slot of logical slot (2g+kh) for this lane's row
  int fslot[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) fslot[g] = ((2 * g + kh) ^ ((li >> 1) & 7)) * 4;

  // ---- loader state machine
  int l_id = (int)blockIdx.x - (int)gridDim.x;
  int l_kc = nk;
  bool l_done = false;
  int rb[4], rt[4];
  bool rvalid[4];
  int seg = -1, seg_left = 0, kk = 0, cin = 0;
  const float* aptr[4];
  bool aok[4];
  const float* bptr[4];

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
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bptr[j] = p.Wt + (long)(n0 + drow + 8 * j) * p.K + dslot[j];
      const int m = m0 + drow + 8 * j;
      bool v = m < p.M;
      int b, t;
      if (ZOUT) {
        t = m / p.BP;
        b = m - t * p.BP;
        v = v && (b < p.B);
      } else {
        b = m / p.T_out;
        t = m - b * p.T_out;
      }
      rb[j] = b;
      rt[j] = t;
      rvalid[j] = v;
    }
  };

  auto next_segment = [&]() {
    ++seg;
    const GemmSeg& sg = p.seg[seg];
    seg_left = sg.kpad / GEMM_BK;
    kk = 0;
    cin = sg.cin;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int in_t = rt[j] * sg.stride + sg.shift;
      const bool ok = rvalid[j] && in_t >= 0 && in_t < sg.w_in;
      aok[j] = ok;
      const long row = sg.time_major ? ((long)in_t * p.BP + rb[j]) : ((long)rb[j] * sg.w_in + in_t);
      aptr[j] = sg.src + (ok ? row * sg.lda : 0) + sg.col0 + dslot[j];
    }
  };

  // issue the DMA of the next chunk into LDS buffer `buf`; false when nothing is left
  auto dma_chunk = [&](int buf) -> bool {
    if (l_kc == nk) next_tile();
    if (l_done) return false;
    if (seg_left == 0) next_segment();
    float* a_dst = As + buf * DTILE_F + wave * 1024;   // wave-uniform bases; lanes land at +16 B each
    float* b_dst = Bs + buf * DTILE_F + wave * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = aok[j] && (kk + dslot[j] < cin);
      const float* src = ok ? aptr[j] + kk : p.zero_page + (lane & 7) * 4;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a_dst + j * 256), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)bptr[j], (lptr_t)(b_dst + j * 256), 16, 0, 0);
      bptr[j] += GEMM_BK;
    }
    kk += GEMM_BK;
    --seg_left;
    ++l_kc;
    return true;
  };

  bool more = dma_chunk(0);
  if (!more) return;
  int buf = 0;
  f32x16 acc[2][2];
  bool have_prev = false;
  int pm0 = 0, pn0 = 0;

  for (int c_id = blockIdx.x; c_id < total_ids; c_id += gridDim.x) {
    int m0, n0;
    if (!tile_of(c_id, m0, n0)) continue;
    for (int kc = 0; kc < nk; ++kc) {
      __syncthreads();  // chunk in `buf` has landed (vmcnt drained before the barrier); buf^1 is free
      if (more) more = dma_chunk(buf ^ 1);
      if (kc == 0) {
        // The previous tile's epilogue is issued HERE, after this tile's first barrier: the barrier's
        // vmcnt(0) then only ever waits for DMA issued a whole chunk earlier, never for fresh stores.
        if (have_prev) gemm_epilogue<ZOUT, RES>(p, acc, pm0, pn0, wm, wn, li, kh);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      }
      const float* a0 = As + buf * DTILE_F + (wm * 64 + li) * GEMM_BK;
      const float* b0 = Bs + buf * DTILE_F + (wn * 64 + li) * GEMM_BK;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * GEMM_BK + fslot[g]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(b0 + ni * 32 * GEMM_BK + fslot[g]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = ZOUT ? __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni][j], a[mi][j], acc[mi][ni], 0, 0, 0);
      }
      buf ^= 1;
    }
    have_prev = true;
    pm0 = m0;
    pn0 = n0;
  }
  if (have_prev) gemm_epilogue<ZOUT, RES>(p, acc, pm0, pn0, wm, wn, li, kh);
}

void launch_gemm(const GemmParams& p, hipStream_t stream) {
  const int nblocks_n = (p.N + GEMM_BN - 1) / GEMM_BN;
  const int mblocks = (p.M + GEMM_BM - 1) / GEMM_BM;
  const int total_ids = ((mblocks + 7) / 8) * 8 * nblocks_n;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  int g = 2 * n_cu;            // two resident workgroups per CU (73.7 KB LDS, <=168 VGPRs each)
  g = (g / 8) * 8;
  if (g > total_ids) g = total_ids;
  const dim3 grid(g), block(256);
  const bool lift = p.seg[0].src == nullptr;
  if (lift)
    hipLaunchKernelGGL((gemm_f32_kernel<true, false, false>), grid, block, 0, stream, p);
  else if (p.out_mode == 1)
    hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false>), grid, block, 0, stream, p);
  else if (p.res_a != nullptr)
    hipLaunchKernelGGL((gemm_f32_dma_kernel<false, true>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((gemm_f32_dma_kernel<false, false>), grid, block, 0, stream, p);
}

}  // namespace chiron
