// 1 x 3, stride-1 convolution over 256-wide channels-last activations (cnn.py:234-262: branch2/conv2b of the residual
// blocks that follow the first one) in Winograd F(2,3) form on v_mfma_f32_32x32x2_f32.
//
// Two neighbouring output positions 2p, 2p+1 of one segment share their four input positions d0..d3 = x[2p-1 .. 2p+2]
// (SAME padding: x[-1] = x[T] = 0).  With the filter taps g0, g1, g2 (BN scale folded in):
//     m0 = (d0 - d2) g0            m1 = (d1 + d2) (g0 + g1 + g2)/2
//     m2 = (d2 - d1) (g0 - g1 + g2)/2            m3 = (d1 - d3) g2
//     y[2p] = m0 + m1 + m2         y[2p+1] = m1 - m2 - m3
// -- four C x C products per position pair instead of six: 2/3 of the multiply-adds of the direct form.  Every GEMM of
// the engine runs at the fp32 matrix-pipe limit (64 FLOP/clk/SIMD, the vector rate), so the only way to make this
// layer faster is to execute fewer of them; the transforms cost two v_pk_add_f32 per operand quad next to 16 MFMAs.
//
//   workgroup tile: 128 position pairs (256 output rows) x 64 output channels, 8 waves as 4 x 2, each 32 pairs x 32
//   channels x 4 products = 4 accumulators of 16 registers.  One persistent workgroup per CU (133 KB of LDS).
//   Staging: LDS-DMA (buffer_load_dwordx4 ... lds, as gemm.hip), double buffered over the 8 channel chunks of 32.
//   The four operands of a pair are rows of TWO halo tiles, not four: E[i] = x[even position of pair m0 + i] and
//   O[i] = x[odd position of pair m0 - 1 + i], 129 rows each, so d1 = E[i], d3 = E[i+1], d0 = O[i], d2 = O[i+1]
//   (the LDS-staged stencil: every input row is fetched once per tile and read by both pairs that need it).  A pair at
//   the start / end of its segment must see the SAME padding zero instead of its neighbour segment's row: those two
//   operands are read from a row of zeros (an address chosen once per tile, nothing per step).  Four filter tiles U_j
//   complete a chunk: 34 + 32 one-KB DMA pieces per 512 MFMAs.
//   Rows keep the XOR swizzle of gemm.hip (physical 16-byte slot = logical slot ^ ((row >> 1) & 7), applied on the
//   source side), so a fragment read is one conflict-free ds_read_b128 feeding four MFMAs.
//   Epilogue: the lane that owns pair p holds 16 channels of all four products: y[2p], y[2p+1] = combinations + shift,
//   ReLU, two 16-byte stores per channel quad.
// Differs from the direct form by fp32 rounding only (the transforms re-associate the sum over taps).
#include "kernels.h"
#ifndef CHIRON_WINO_ROWMAJOR_STORES
#define CHIRON_WINO_ROWMAJOR_STORES 1   // store order of the F(4,3) epilogue: 1 = row-major (round 6, WRITE_SIZE 556 -> 453 MB per launch for 450.6 stored, same time: profiles/r06_wino_store_ab.txt); 0 = round 5s (A/B: tools/variants.sh --product wino CHIRON_WINO_ROWMAJOR_STORES 0)
#endif
#include "timing_variants.h"

#include <cstdlib>

namespace chiron {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// a * b + c on two fp32 values at once (v_pk_fma_f32)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
typedef void __attribute__((address_space(3)))* lptr_t;

namespace {

constexpr int WP = 128;                 // position pairs per tile
constexpr int WN = 64;                  // output channels per tile
constexpr int WK = 32;                  // channels per chunk (128 bytes per row)
constexpr int HALO_ROWS = 136;          // 129 rows used, 17 DMA pieces of 8 rows
constexpr int HALO_F = HALO_ROWS * WK;  // floats per halo tile
constexpr int U_F = WN * WK;            // floats per filter sub-tile
constexpr int BUF_F = 2 * HALO_F + 4 * U_F;   // E, O, U0..U3: 66 KB per buffer
constexpr unsigned OOB = 0xFFFF0000u;   // byte offset past num_records (+ any chunk offset): reads zeros
constexpr unsigned RECORDS = 0xFFFE0000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, RECORDS, 0x00027000);
}

}  // namespace

__global__ __launch_bounds__(512, 2) void wino_conv3_kernel(const WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][BUF_F] tiles, one row of zeros, shift[N]
  float* const zrow = lds + 2 * BUF_F;
  float* const shl = zrow + WK;
  const int tid = threadIdx.x;
  if (p.C < 0) lds[tid] = 0.f;   // the tiles are only ever written by the DMA engine (see gemm.hip)
  if (tid < WK) zrow[tid] = 0.f;
  for (int n = tid; n < p.N; n += 512) shl[n] = p.shift[n];
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;

  const int half_t = p.T >> 1;                      // pairs per segment
  const int mp = p.B * half_t;                      // pair rows in total
  const int mblocks = (mp + WP - 1) / WP;
  const int nblocks = p.N / WN;
  const int total_ids = ((mblocks + 7) / 8) * 8 * nblocks;
  const int chunks = p.C / WK;

  auto tile_of = [&](int id, int& m0, int& n0) -> bool {   // XCD-aware: the N tiles of an M block share one XCD's L2
    const int xcd = id & 7, slot = id >> 3;
    const int mblk = (slot / nblocks) * 8 + xcd;
    m0 = mblk * WP;
    n0 = (slot % nblocks) * WN;
    return mblk < mblocks;
  };
  auto next_valid = [&](int id) -> int {
    int m0, n0;
    do id += gridDim.x;
    while (id < total_ids && !tile_of(id, m0, n0));
    return id;
  };

  // ---- loader: a piece = 8 rows x 128 bytes; lane -> (row 8*piece + lane>>3, physical slot lane&7).
  //      wave w loads pieces 2w, 2w+1 of E and of O, the 17th piece of E (wave 0) / O (wave 1), and pieces 4w .. 4w+3 of
  //      the 32 filter pieces (sub-tile j = piece >> 3).
  const __amdgpu_buffer_rsrc_t ra = rsrc_of(p.src), rb = rsrc_of(p.U);
  unsigned eoff[3], ooff[3], boff[4];   // byte offsets of this lane's 16 bytes in chunk 0, or OOB
  auto halo_off = [&](int m0, int row, int odd) -> unsigned {
    const int slot = ((lane & 7) ^ ((row >> 1) & 7)) * 4;   // logical k-slot (floats) fetched into physical slot lane&7
    const int pr = m0 + row - odd;                          // E[row] <- pair m0 + row (even position), O[row] <- pair m0 - 1 + row (odd)
    if (pr < 0 || pr >= mp) return OOB;
    const int b = pr / half_t, pp = pr - b * half_t;
    return (unsigned)((((long)b * p.T + 2 * pp + odd) * p.lda + slot) * 4);
  };
  auto load_tile = [&](int id) {
    int m0, n0;
    tile_of(id, m0, n0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int piece = i < 2 ? 2 * wave + i : 16;
      const int row = 8 * piece + (lane >> 3);
      eoff[i] = halo_off(m0, row, 0);
      ooff[i] = halo_off(m0, row, 1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = 4 * wave + i, j = piece >> 3;
      const int row = 8 * (piece & 7) + (lane >> 3);
      const int slot = ((lane & 7) ^ ((row >> 1) & 7)) * 4;
      boff[i] = (unsigned)((((long)j * p.N + n0 + row) * p.C + slot) * 4);
    }
  };
  // this wave's pieces of chunk c into buffer `dst`, in four parts so that they can be spread over the chunk's MFMA groups
  // (a burst of nine DMA instructions stalls the wave that issues it; all waves of the CU share one address unit)
  auto issue_part = [&](int part, int c, float* dst) {
    const unsigned ck = (unsigned)(c * WK * 4);
    if (part < 2) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(dst + (2 * wave + part) * 256), 16, eoff[part] + ck, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(dst + HALO_F + (2 * wave + part) * 256), 16, ooff[part] + ck, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(dst + 2 * HALO_F + (4 * wave + part) * 256), 16, boff[part] + ck, 0, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(dst + 2 * HALO_F + (4 * wave + part) * 256), 16, boff[part] + ck, 0, 0, 0);
      if (part == 2 && wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(dst + 16 * 256), 16, eoff[2] + ck, 0, 0, 0);
      if (part == 2 && wave == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(dst + HALO_F + 16 * 256), 16, ooff[2] + ck, 0, 0, 0);
    }
  };
  auto issue = [&](int c, float* dst) {
#pragma unroll
    for (int part = 0; part < 4; ++part) issue_part(part, c, dst);
  };

  // fragment slots of this lane's rows: row r = wm*32 + li (E[r] = d1, O[r] = d0) and r + 1 (E = d3, O = d2)
  int fs0[4], fs1[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    fs0[g] = ((2 * g + kh) ^ ((li >> 1) & 7)) * 4;
    fs1[g] = ((2 * g + kh) ^ (((li + 1) >> 1) & 7)) * 4;
  }

  int c_id = blockIdx.x;
  {
    int m0, n0;
    if (c_id >= total_ids) return;
    if (!tile_of(c_id, m0, n0)) c_id = next_valid(c_id);
    if (c_id >= total_ids) return;
  }
  load_tile(c_id);
  issue(0, lds);
  int buf = 0;
  f32x16 acc[4];
  bool have_prev = false;
  int prev_pr = 0, prev_n0 = 0;
  // y[2p], y[2p+1] of the finished tile: this lane owns pair `pr` and the channels n0 + wn*32 + 8q + 4kh + r (q, r = 0..3)
  auto epilogue = [&](int pr, int n0) {
    if (pr >= mp) return;
    const int pb = pr / half_t, pp = pr - pb * half_t;
    float* o0 = p.out + ((long)pb * p.T + 2 * pp) * p.ldo + n0 + wn * 32 + 4 * kh;
    float* o1 = o0 + p.ldo;
    const float* sh = shl + n0 + wn * 32 + 4 * kh;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(sh + 8 * q);
      f32x4 y0, y1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a0 = acc[0][4 * q + r], a1 = acc[1][4 * q + r], a2 = acc[2][4 * q + r], a3 = acc[3][4 * q + r];
        y0[r] = ((a0 + a1) + a2) + s4[r];
        y1[r] = ((a1 - a2) - a3) + s4[r];
        if (p.relu) {
          y0[r] = __builtin_amdgcn_fmed3f(y0[r], 0.f, INFINITY);
          y1[r] = __builtin_amdgcn_fmed3f(y1[r], 0.f, INFINITY);
        }
      }
      *reinterpret_cast<f32x4*>(o0 + 8 * q) = y0;
      *reinterpret_cast<f32x4*>(o1 + 8 * q) = y1;
    }
  };
  while (c_id < total_ids) {
    int m0, n0;
    tile_of(c_id, m0, n0);
    const int n_id = next_valid(c_id);
    // this lane's pair: SAME padding at the ends of its segment comes from the row of zeros
    const int pr = m0 + wm * 32 + li;
    const int pp = pr % half_t;
    const bool first = pp == 0, last = pp == half_t - 1;
    for (int c = 0; c < chunks; ++c) {
      __syncthreads();   // chunk c has landed in `buf` (vmcnt is drained before the barrier); buf ^ 1 is free
      float* const nxt = lds + (buf ^ 1) * BUF_F;
      const bool more = c + 1 < chunks;
      const bool go = more || n_id < total_ids;
      if (!more && go) load_tile(n_id);
      const int nc = more ? c + 1 : 0;
      if (c == 0) {
        // the previous tile's stores are issued HERE, behind this tile's first barrier, so that no barrier ever waits for
        // fresh stores; then the accumulators start from zero
        if (have_prev) epilogue(prev_pr, prev_n0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
      }
      const float* t0 = lds + buf * BUF_F;
      const int r0 = (wm * 32 + li) * WK;
      const float* e0 = t0 + r0;                                   // d1
      const float* e1 = last ? zrow - fs1[0] : t0 + r0 + WK;       // d3 (the zero row is read at offset 0..28: any slot is zero)
      const float* o0 = first ? zrow - fs0[0] : t0 + HALO_F + r0;  // d0
      const float* o1 = t0 + HALO_F + r0 + WK;                     // d2
      const float* brow = t0 + 2 * HALO_F + (wn * 32 + li) * WK;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(e0 + fs0[g]);
        const f32x4 d3 = *reinterpret_cast<const f32x4*>(e1 + (last ? fs1[0] : fs1[g]));
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(o0 + (first ? fs0[0] : fs0[g]));
        const f32x4 d2 = *reinterpret_cast<const f32x4*>(o1 + fs1[g]);
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(brow + 0 * U_F + fs0[g]);
        const f32x4 u1 = *reinterpret_cast<const f32x4*>(brow + 1 * U_F + fs0[g]);
        const f32x4 u2 = *reinterpret_cast<const f32x4*>(brow + 2 * U_F + fs0[g]);
        const f32x4 u3 = *reinterpret_cast<const f32x4*>(brow + 3 * U_F + fs0[g]);
        const f32x4 v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3;
        if (go) issue_part(g, nc, nxt);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0[j], v0[j], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1[j], v1[j], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2[j], v2[j], acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(u3[j], v3[j], acc[3], 0, 0, 0);
        }
      }
      buf ^= 1;
    }
    have_prev = true;
    prev_pr = pr;
    prev_n0 = n0;
    c_id = n_id;
  }
  if (have_prev) epilogue(prev_pr, prev_n0);
}

// ---------------------------------------------------------------------------------------------------------
// F(4,3): four neighbouring outputs 4q .. 4q+3 share six inputs d0..d5 = x[4q-1 .. 4q+4]: SIX C x C products per output
// quad instead of twelve -- half of the direct form's multiply-adds (F(2,3): two thirds).  Transforms (Lavin & Gray):
//     v0 = 4 d0 - 5 d2 + d4            v1 = (d4 - 4 d2) + (d3 - 4 d1)      v2 = (d4 - 4 d2) - (d3 - 4 d1)
//     v3 = (d4 - d2) + 2 (d3 - d1)     v4 = (d4 - d2) - 2 (d3 - d1)        v5 = 4 d1 - 5 d3 + d5
//     U  = G g,  G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]   (float64, engine creation)
//     y0 = m0+m1+m2+m3+m4   y1 = (m1-m2) + 2 (m3-m4)   y2 = (m1+m2) + 4 (m3+m4)   y3 = (m1-m2) + 8 (m3-m4) + m5
// In float32 the result is as close to the float64 oracle as the F(2,3) and direct forms on this network (measured with the
// numpy oracle: logits 6.5e-6 / 1.1e-5 / 5.7e-6 max deviation), so the 1e-4 bound keeps its margin.
//
//   workgroup tile: 128 quads (512 output rows) x 64 channels, 8 waves as 4 x 2, each 32 quads x 32 channels x 6 products.
//   Channel chunks of 16 (64-byte rows) so that four halo tiles -- P_r[i] = x[4 (q0 + i) + r] for r = 0, 1, 2 and
//   P_3[i] = x[4 (q0 - 1 + i) + 3], 129 rows each: d0 = P3[i], d1 = P0[i], d2 = P1[i], d3 = P2[i], d4 = P3[i+1],
//   d5 = P0[i+1] -- and six filter tiles fit twice into the LDS (116 KB).  A DMA piece is 16 rows x 64 bytes; the physical
//   16-byte slot of logical slot s in row r is s ^ ((r >> 2) & 3), which keeps every ds_read_b128 group on distinct banks.
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int Q4 = 128;                 // quads per tile
constexpr int K4 = 16;                  // channels per chunk (64 bytes per row)
constexpr int P_ROWS = 144;             // 129 rows used, 9 DMA pieces of 16 rows
constexpr int P_F = P_ROWS * K4;        // floats per halo tile
constexpr int U4_F = WN * K4;           // floats per filter sub-tile
constexpr int BUF4_F = 4 * P_F + 6 * U4_F;   // 15360 floats = 60 KB per buffer
}  // namespace

__global__ __launch_bounds__(512, 2) void wino_conv3_f4_kernel(const WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][BUF4_F] tiles, one row of zeros, shift[N]
  float* const zrow = lds + 2 * BUF4_F;
  float* const shl = zrow + K4;
  const int tid = threadIdx.x;
  if (p.C < 0) lds[tid] = 0.f;   // the tiles are only ever written by the DMA engine (see gemm.hip)
  if (tid < K4) zrow[tid] = 0.f;
  for (int n = tid; n < p.N; n += 512) shl[n] = p.shift[n];
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;

  const int qpw = p.T >> 2;                         // quads per segment
  const int mq = p.B * qpw;                         // quad rows in total
  const int mblocks = (mq + Q4 - 1) / Q4;
  const int nblocks = p.N / WN;
  const int total_ids = ((mblocks + 7) / 8) * 8 * nblocks;
  const int chunks = p.C / K4;

  auto tile_of = [&](int id, int& m0, int& n0) -> bool {
    const int xcd = id & 7, slot = id >> 3;
    const int mblk = (slot / nblocks) * 8 + xcd;
    m0 = mblk * Q4;
    n0 = (slot % nblocks) * WN;
    return mblk < mblocks;
  };
  auto next_valid = [&](int id) -> int {
    int m0, n0;
    do id += gridDim.x;
    while (id < total_ids && !tile_of(id, m0, n0));
    return id;
  };

  // ---- loader: a piece = 16 rows x 64 bytes; lane -> (row 16*piece + lane>>2, physical slot lane&3).
  //      wave w loads pieces 4(w&1) .. +3 of halo tile w>>1, the ninth piece of tile w>>1 if w is even, and filter pieces
  //      3w .. 3w+2 of the 24 (sub-tile j = piece >> 2).
  const __amdgpu_buffer_rsrc_t ra = rsrc_of(p.src), rb = rsrc_of(p.U);
  unsigned aoff[5], boff[3];
  const int ptile = wave >> 1;                      // halo tile this wave loads: position class r = ptile
  auto halo_off = [&](int m0, int row) -> unsigned {
    const int slot = ((lane & 3) ^ ((row >> 2) & 3)) * 4;   // logical k-slot (floats) fetched into physical slot lane&3
    const int qg = m0 + row - (ptile == 3 ? 1 : 0);         // P3[row] belongs to quad m0 - 1 + row
    if (qg < 0 || qg >= mq) return OOB;
    const int b = qg / qpw, qq = qg - b * qpw;
    return (unsigned)((((long)b * p.T + 4 * qq + ptile) * p.lda + slot) * 4);
  };
  auto load_tile = [&](int id) {
    int m0, n0;
    tile_of(id, m0, n0);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int piece = i < 4 ? 4 * (wave & 1) + i : 8;
      aoff[i] = halo_off(m0, 16 * piece + (lane >> 2));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int piece = 3 * wave + i, j = piece >> 2;
      const int row = 16 * (piece & 3) + (lane >> 2);
      const int slot = ((lane & 3) ^ ((row >> 2) & 3)) * 4;
      boff[i] = (unsigned)((((long)j * p.N + n0 + row) * p.C + slot) * 4);
    }
  };
  auto issue_part = [&](int part, int c, float* dst) {   // parts 0..1, one per k-group of the chunk
    const unsigned ck = (unsigned)(c * K4 * 4);
    float* const pt = dst + ptile * P_F;
    float* const ut = dst + 4 * P_F;
    if (part == 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(pt + (4 * (wave & 1) + 0) * 256), 16, aoff[0] + ck, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(pt + (4 * (wave & 1) + 1) * 256), 16, aoff[1] + ck, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(ut + (3 * wave + 0) * 256), 16, boff[0] + ck, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(ut + (3 * wave + 1) * 256), 16, boff[1] + ck, 0, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(pt + (4 * (wave & 1) + 2) * 256), 16, aoff[2] + ck, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(pt + (4 * (wave & 1) + 3) * 256), 16, aoff[3] + ck, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(ut + (3 * wave + 2) * 256), 16, boff[2] + ck, 0, 0, 0);
      if (!(wave & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(pt + 8 * 256), 16, aoff[4] + ck, 0, 0, 0);
    }
  };

  // fragment slots of this lane's rows: row r = wm*32 + li and r + 1
  int fs0[2], fs1[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    fs0[g] = ((2 * g + kh) ^ ((li >> 2) & 3)) * 4;
    fs1[g] = ((2 * g + kh) ^ (((li + 1) >> 2) & 3)) * 4;
  }

  // tile ids: fixed shares (stride gridDim.x) or the next slot of this workgroup's XCD from the stream's counters (see
  // GemmParams::tile_ctr in kernels.h: a persistent workgroup that starts late no longer sets the end of the launch)
  __shared__ int sh_next;
  const bool dyn = p.tile_ctr != nullptr && chunks >= 3;
  const int my_xcd = blockIdx.x & 7;
  const unsigned slots_per_xcd = (unsigned)(total_ids >> 3);
  unsigned long long grabbed = 0;
  auto issue_grab = [&]() {
    if (tid == 0) grabbed = atomicAdd(p.tile_ctr + my_xcd, 1ull);
  };
  auto decode_grab = [&]() -> int {   // thread 0
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
  issue_part(0, 0, lds);
  issue_part(1, 0, lds);
  int buf = 0;
  f32x16 acc[6];
  bool have_prev = false;
  int prev_q = 0, prev_n0 = 0;
  auto epilogue = [&](int qg, int n0) {   // this lane owns quad qg and the channels n0 + wn*32 + 8q + 4kh + r
    if (qg >= mq) return;
    const int b = qg / qpw, qq = qg - b * qpw;
    float* o0 = p.out + ((long)b * p.T + 4 * qq) * p.ldo + n0 + wn * 32 + 4 * kh;
    const float* sh = shl + n0 + wn * 32 + 4 * kh;
#if CHIRON_WINO_ROWMAJOR_STORES
    f32x4 yy[4][4];
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(sh + 8 * q);
      f32x4 y0, y1, y2, y3;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = 4 * q + r;
        const float m0_ = acc[0][e], m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e], m5 = acc[5][e];
        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
#if CHIRON_SENS & 4
        y0[r] = m0_ + m4, y1[r] = m1, y2[r] = m2 + m5, y3[r] = m3 + s4[r];   // timing experiment: no output transform
#else
        y0[r] = ((m0_ + s12) + s34) + s4[r];
        y1[r] = fmaf(2.0f, d34, d12) + s4[r];
        y2[r] = fmaf(4.0f, s34, s12) + s4[r];
        y3[r] = (fmaf(8.0f, d34, d12) + m5) + s4[r];
#endif
        if (p.relu) {
          y0[r] = __builtin_amdgcn_fmed3f(y0[r], 0.f, INFINITY);
          y1[r] = __builtin_amdgcn_fmed3f(y1[r], 0.f, INFINITY);
          y2[r] = __builtin_amdgcn_fmed3f(y2[r], 0.f, INFINITY);
          y3[r] = __builtin_amdgcn_fmed3f(y3[r], 0.f, INFINITY);
        }
      }
#if CHIRON_WINO_ROWMAJOR_STORES
      yy[0][q] = y0, yy[1][q] = y1, yy[2][q] = y2, yy[3][q] = y3;
    }
    // (round-5 review, item 7; measured: profiles/r06_wino_store_ab.txt) the four 32-byte pieces that complete one 128-byte line of an output row leave in four
    // CONSECUTIVE store instructions (row-major) instead of every fourth -- does the L2 merge them better (WRITE_SIZE 553 MB for 450 stored)?
#pragma unroll
    for (int row = 0; row < 4; ++row)
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(o0 + row * p.ldo + 8 * q) = yy[row][q];
  };
#else
      *reinterpret_cast<f32x4*>(o0 + 8 * q) = y0;
      *reinterpret_cast<f32x4*>(o0 + p.ldo + 8 * q) = y1;
      *reinterpret_cast<f32x4*>(o0 + 2 * p.ldo + 8 * q) = y2;
      *reinterpret_cast<f32x4*>(o0 + 3 * p.ldo + 8 * q) = y3;
    }
  };
#endif
#if CHIRON_SENS & 2048
  // timing experiment: cycles per tile in the chunk barriers and in the epilogue (s_memtime), printed by wave 0 of four workgroups
  unsigned long long clk_bar = 0, clk_epi = 0, clk_tiles = 0;
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime();
#endif
  while (c_id < total_ids) {
    int m0, n0;
    tile_of(c_id, m0, n0);
    int n_id = dyn ? total_ids : next_valid(c_id);
    const int qg = m0 + wm * 32 + li;
    const int qq = qg % qpw;
    const bool first = qq == 0, last = qq == qpw - 1;   // SAME padding: x[-1] and x[T] come from the row of zeros
    for (int c = 0; c < chunks; ++c) {
#if CHIRON_SENS & 2048
      const unsigned long long clk_b0 = __builtin_amdgcn_s_memtime();
#endif
      __syncthreads();   // chunk c has landed in `buf`; buf ^ 1 is free
#if CHIRON_SENS & 2048
      clk_bar += __builtin_amdgcn_s_memtime() - clk_b0;
#endif
      if (dyn) {   // the next tile's number: requested in chunk 0, published in chunk 1, read in chunk 2
        if (c == 0) issue_grab();
        if (c == 1 && tid == 0) sh_next = decode_grab();
        if (c == 2) n_id = sh_next;
      }
      float* const nxt = lds + (buf ^ 1) * BUF4_F;
      const bool more = c + 1 < chunks;
      const bool go = more || n_id < total_ids;
      if (!more && go) load_tile(n_id);
      const int nc = more ? c + 1 : 0;
      if (c == 0) {
#if CHIRON_SENS & 2048
        const unsigned long long clk_e0 = __builtin_amdgcn_s_memtime();
#endif
        if (have_prev) epilogue(prev_q, prev_n0);
#if CHIRON_SENS & 2048
        asm volatile("" ::: "memory");
        clk_epi += __builtin_amdgcn_s_memtime() - clk_e0;
#endif
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
      }
      const float* t0 = lds + buf * BUF4_F;
      const int r0 = (wm * 32 + li) * K4;
      const float* p0 = t0 + r0;                 // d1 = P0[i];  d5 = P0[i+1]
      const float* p1 = t0 + P_F + r0;           // d2
      const float* p2 = t0 + 2 * P_F + r0;       // d3
      const float* p3 = t0 + 3 * P_F + r0;       // d0 = P3[i];  d4 = P3[i+1]
      const float* brow = t0 + 4 * P_F + (wn * 32 + li) * K4;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        // the two padded inputs are chosen by ADDRESS (one select each), not by value (four each)
        const float* a0 = first ? zrow : p3 + fs0[g];
        const float* a5 = last ? zrow : p0 + K4 + fs1[g];
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(a0);
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(p0 + fs0[g]);
        const f32x4 d2 = *reinterpret_cast<const f32x4*>(p1 + fs0[g]);
        const f32x4 d3 = *reinterpret_cast<const f32x4*>(p2 + fs0[g]);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(p3 + K4 + fs1[g]);
        const f32x4 d5 = *reinterpret_cast<const f32x4*>(a5);
        f32x4 u[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) u[j] = *reinterpret_cast<const f32x4*>(brow + j * U4_F + fs0[g]);
        // input transform v = B^T d on PACKED fp32 (v_pk_fma_f32 / v_pk_add_f32: two values per instruction; every VALU
        // instruction of this loop is paid in matrix-pipe time): 12 packed operations per pair instead of 24 scalar ones
        f32x4 v[6];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 e0 = {d0[2 * h], d0[2 * h + 1]}, e1 = {d1[2 * h], d1[2 * h + 1]}, e2 = {d2[2 * h], d2[2 * h + 1]};
          const f32x2 e3 = {d3[2 * h], d3[2 * h + 1]}, e4 = {d4[2 * h], d4[2 * h + 1]}, e5 = {d5[2 * h], d5[2 * h + 1]};
          const f32x2 c4 = {4.0f, 4.0f}, c5 = {5.0f, 5.0f}, c2 = {2.0f, 2.0f};
          const f32x2 a = pk_fma(-c4, e2, e4), b = pk_fma(-c4, e1, e3), cc = e4 - e2, e = e3 - e1;
          const f32x2 w0 = pk_fma(c4, e0, pk_fma(-c5, e2, e4)), w1 = a + b, w2 = a - b, w3 = pk_fma(c2, e, cc), w4 = pk_fma(-c2, e, cc),
                      w5 = pk_fma(c4, e1, pk_fma(-c5, e3, e5));
          v[0][2 * h] = w0[0], v[0][2 * h + 1] = w0[1];
          v[1][2 * h] = w1[0], v[1][2 * h + 1] = w1[1];
          v[2][2 * h] = w2[0], v[2][2 * h + 1] = w2[1];
          v[3][2 * h] = w3[0], v[3][2 * h + 1] = w3[1];
          v[4][2 * h] = w4[0], v[4][2 * h + 1] = w4[1];
          v[5][2 * h] = w5[0], v[5][2 * h + 1] = w5[1];
        }
#if CHIRON_SENS & 2
        v[0] = d0, v[1] = d1, v[2] = d2, v[3] = d3, v[4] = d4, v[5] = d5;   // timing experiment: no input transform
#endif
        if (go) issue_part(g, nc, nxt);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int q = 0; q < 6; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[q][j], v[q][j], acc[q], 0, 0, 0);
        }
      }
      buf ^= 1;
    }
    have_prev = true;
    prev_q = qg;
    prev_n0 = n0;
    c_id = n_id;
#if CHIRON_SENS & 2048
    ++clk_tiles;
#endif
  }
  if (have_prev) epilogue(prev_q, prev_n0);
#if CHIRON_SENS & 2048
  if (tid == 0 && blockIdx.x < 4 && clk_tiles > 0)
    printf("wino_clk block %d: tiles %llu, cycles per tile: total %llu, barriers %llu (%d chunks), epilogue %llu\n", (int)blockIdx.x, clk_tiles,
           (__builtin_amdgcn_s_memtime() - clk_t0) / clk_tiles, clk_bar / clk_tiles, chunks, clk_epi / clk_tiles);
#endif
}

bool launch_wino_conv3(const WinoParams& p, hipStream_t stream) {
  if (p.T <= 0 || (p.T & 1) || p.C <= 0 || (p.C % WK) || p.N <= 0 || (p.N % WN) || p.N > 1024) return false;
  if ((size_t)p.B * p.T * (size_t)p.lda * 4 > RECORDS) return false;
  const int n_cu = current_device_cus();
  static char attr_set[CHIRON_MAX_DEVICES] = {};   // the dynamic-LDS opt-in is a per-device function attribute
  const int dev = current_device_index();
  if (!__atomic_load_n(&attr_set[dev], __ATOMIC_ACQUIRE)) {
    // what the launches below ask for at most (N <= 1024); the limit counts the kernels' static LDS too, so not "all 160 KB"
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (2 * BUF_F + WK + 1024) * 4) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv3_f4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (2 * BUF4_F + K4 + 1024) * 4) != hipSuccess) {
      (void)hipGetLastError();   // not sticky: the caller takes the GEMM form
      return false;
    }
    __atomic_store_n(&attr_set[dev], (char)1, __ATOMIC_RELEASE);
  }
  // one workgroup per CU is resident (121 KB of LDS); with tiles by counter a launch may ask for more: the extra workgroups start
  // when a CU frees up and take what is left (CHIRON_WINO_WGS_PER_CU: A/B knob, default 1)
  static const int wgs_per_cu = [] {
    const char* v = getenv("CHIRON_WINO_WGS_PER_CU");
    const int n = v ? atoi(v) : 1;
    return n >= 1 && n <= 4 ? n : 1;
  }();
  int g = (wgs_per_cu * n_cu / 8) * 8;
  if (p.f4) {   // F(4,3): U holds six transformed filters, T is a multiple of 4
    if (p.T % 4) return false;
    const int mblocks = (p.B * (p.T / 4) + Q4 - 1) / Q4;
    const int total_ids = ((mblocks + 7) / 8) * 8 * (p.N / WN);
    if (g > total_ids) g = total_ids;
    WinoParams q = p;
    const bool dyn = q.tile_ctr != nullptr && q.tile_base_host != nullptr && p.C / K4 >= 3 && g % 8 == 0;
    if (dyn) q.tile_base = *q.tile_base_host;
    else q.tile_ctr = nullptr;
    hipLaunchKernelGGL(wino_conv3_f4_kernel, dim3(g), dim3(512), (size_t)(2 * BUF4_F + K4 + p.N) * 4, stream, q);
    if (dyn) *q.tile_base_host += (unsigned long long)(total_ids / 8 + g / 8);   // what the launch takes from each counter
    return true;
  }
  const int half_t = p.T / 2;
  const int mblocks = (p.B * half_t + WP - 1) / WP;
  const int total_ids = ((mblocks + 7) / 8) * 8 * (p.N / WN);
  if (g > total_ids) g = total_ids;
  hipLaunchKernelGGL(wino_conv3_kernel, dim3(g), dim3(512), (size_t)(2 * BUF_F + WK + p.N) * 4, stream, p);
  return true;
}

}  // namespace chiron
