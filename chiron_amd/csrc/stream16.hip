// fp16 engine: the 1 x 1 convolutions of a residual block (cnn.py:234-262: branch2/conv2a, and branch2/conv2c fused
// with the branch1 convolution) as a STREAMING kernel: weights in registers, the LDS as a deep FIFO of activation rows.
//
// Why.  At f16 these launches carry 8x less matrix-pipe work per byte than at fp32 (BASELINE configs[4], B = 4096: 0.84 GB in,
// 0.84 GB out, 0.2 ms of MFMA), and the tiled GEMM (gemm.hip) spends its LDS on double-buffered A AND weight tiles: one 32 KB
// chunk in flight per workgroup, 64 KB per CU, ~3.5 us per chunk -- it runs at DMA latency, 2.3 TB/s of HBM.  Here
//   * the whole weight matrix lives in VGPRs for the lifetime of the (persistent, one per CU) workgroup: wave w owns output
//     columns 32w .. 32w+31 for every k -- 64 registers per 256-channel K-segment (A operand of v_mfma_f32_32x32x16_f16,
//     the product is computed transposed, D = W^T x^T, so that a lane ends up with 4 x 4 CONSECUTIVE columns of one row);
//   * the LDS holds nothing but activation tiles (32 rows x 256 channels = 16 KB per K-segment, four to six deep) and the
//     staging tiles the results leave through; every wave is loader, multiplier and storer (the round-2 form had two producer
//     waves: see the note in front of the kernel for why they went).
// Per tile a wave issues 16 (32) MFMAs of 32 cycles: the kernel is HBM-bound by construction.  NSEG = 2 is conv2c + branch1:
// two K-segments (the block's conv2b output and the block's input) into one accumulator.  Applicable when the engine is f16,
// C_in = C_out = 256, stride 1 (res_layer2 / res_layer3 of DNA_default and their RNA counterparts); every other shape keeps
// gemm.hip.
#include "kernels.h"
#include "timing_variants.h"

#include <algorithm>

namespace chiron {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3)))* lptr_t;

constexpr unsigned S_OOB = 0xFFFF0000u;      // byte offset past num_records: the DMA writes zeros
constexpr unsigned S_RECORDS = 0xFFFE0000u;  // every tensor of the engine is smaller than this many bytes
constexpr int S_ROWS = 32;                   // rows per tile
constexpr int S_C = 256;                     // channels per K-segment = output columns
constexpr int S_TILE_H = S_ROWS * S_C;       // halves per segment tile (16 KB)
constexpr int S_NW = 8;                      // waves per workgroup

// The s_barrier of a tile.  The compiler does not know that the DMA engine writes the tiles, so it
// must not move an LDS read of the next tile above the barrier (or keep one below it alive across it): the empty asm
// statements are compiler-only fences, the hardware ordering is the barrier itself (no s_waitcnt vmcnt here -- a
// __syncthreads() would make every compute wave wait for its own output stores once per tile).
static __device__ __forceinline__ void tile_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t s_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, S_RECORDS, 0x00027000);
}

// CHIRON_S16_VARIANT (timing_variants.h): instrumented builds of these kernels, 0 in the product
// Round 3: both sides of the kernel move WHOLE 512-byte rows per instruction.  Round 2 fetched 32 bytes of each of 32 rows per
// DMA instruction (octet-major tiles) and stored 16 bytes of each of 32 rows per store instruction: instrumented builds at
// B = 4096 (0.84 GB in, 0.84 GB out): 0.418 ms, without the stores 0.219, without the loads 0.257, without both 0.194 -- the
// two directions queued behind each other in the L2 request path, neither near the HBM rate.  Now
//   * input tiles are ROW-major [32 rows][32 octets][8 halves], one DMA instruction = two rows (as in conv3 below); logical
//     octet o of row r sits at position o ^ (r & 15), applied to the GLOBAL address of the fetching lane (the LDS side of a
//     DMA is lane-linear), so that the 16 rows of one ds_read_b128 pass fall on 16 different 16-byte bank groups;
//   * the f16 results of a tile go to a staging tile of the same layout (two of them: tile j is written while the copy of
//     tile j - 1 is read) and leave one barrier later as 16-byte pieces of whole rows: 2 store instructions per wave and
//     tile, each two complete rows.
constexpr int S_STAGE_H = S_ROWS * S_C;   // halves per staging tile (16 KB)
//   * no producer waves any more (round 3): with ten waves a SIMD may hold three, i.e. 168 registers per wave, and the
//     two-segment form (128 weight registers) had one 4-register B fragment in flight -- read, wait, multiply, 32 times per
//     tile -- and spilled.  Eight waves own 256 registers each: every wave issues the DMA of ITS four rows of tile j + D - 1
//     right after the barrier of tile j, reads the B fragments of a K-segment sixteen deep, and waits for its own rows of
//     tile j + 1 before the next barrier.  A wave's vmcnt then counts its DMA loads AND its output stores; loads retire in
//     order among themselves, so "at most (D - 2) x (loads per tile) outstanding" proves the loads of tile j + 1 complete
//     whatever the stores do (it may also wait for a store that is one or two tiles old).
template <int NSEG>
__global__ __launch_bounds__(64 * S_NW, 1) void conv1x1_f16_stream_kernel(const GemmParams p) {
  // NSEG = 2 (32 KB per tile): four tiles and two whole staging tiles do not fit; there every wave stages its own 32 x 32
  // block privately (2 KB, same interval, no second buffer) and stores 64-byte row segments -- 16 rows per instruction
  constexpr bool PRIV = NSEG == 2;
  constexpr int D = NSEG == 1 ? 6 : 4;                 // tiles resident in the LDS; D - 1 in flight behind the one being consumed
  constexpr int IPW = 2 * NSEG;                        // DMA instructions per wave and tile (two rows each)
  extern __shared__ __attribute__((aligned(16))) _Float16 tiles[];   // [D][NSEG][32 rows][32 octets][8 halves], then the staging tile(s)
  _Float16* const stage = tiles + D * NSEG * S_TILE_H;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  __shared__ __attribute__((aligned(16))) float shl[S_C];   // folded BN offset per output column
  __shared__ __attribute__((aligned(16))) float rsl[S_C];   // res_layer1: folded scale of the branch1 convolution of the signal
  if (p.M < 0) tiles[tid] = (_Float16)0.f;   // the tiles are only ever written by the DMA engine (see gemm.hip)
  if (tid < S_C) shl[tid] = p.shift ? p.shift[tid] : 0.f;
  if (tid < S_C) rsl[tid] = p.res_a ? p.res_a[tid] : 0.f;
  __syncthreads();

  const int ntiles = (p.M + S_ROWS - 1) / S_ROWS;
  const int first = blockIdx.x, step = gridDim.x;
  const int mine = first < ntiles ? (ntiles - first + step - 1) / step : 0;   // tiles of this workgroup
  if (mine == 0) return;

  // ---------------- input: wave w fetches rows 4 w .. 4 w + 3 of every tile
  // (an array of __amdgpu_buffer_rsrc_t with a template-dependent bound makes hipcc 7.2 drop the kernel's host stub)
  __amdgpu_buffer_rsrc_t rs[2];
  rs[0] = s_rsrc(p.seg[0].src);
  rs[1] = s_rsrc(p.seg[NSEG - 1].src);
  auto issue = [&](int j) {   // tile number j of this workgroup (past the last one: zeros, no memory traffic)
    const int m0 = (first + j * step) * S_ROWS;
    _Float16* base = tiles + (j % D) * NSEG * S_TILE_H;
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = 2 * (wave * 2 + i) + kh;               // row of the tile this lane fetches a piece of
#if CHIRON_S16_VARIANT & 2
        const bool ok = false;
#else
        const bool ok = j < mine && m0 + r < p.M;
#endif
        const unsigned off = ok ? (unsigned)(((long)(m0 + r) * p.seg[sg].lda + p.seg[sg].col0) * 2 + ((li ^ (r & 15)) * 16)) : S_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[sg], (lptr_t)(base + sg * S_TILE_H + (wave * 2 + i) * 2 * S_C), 16, off, 0, 0, 0);
      }
    }
  };

  // ---------------- weights: wave w owns output columns 32 w .. 32 w + 31
  f16x8 wr[NSEG][16];
  {
    const _Float16* wt = reinterpret_cast<const _Float16*>(p.Wt) + (long)(32 * wave + li) * p.K + 8 * kh;
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) wr[sg][ks] = *reinterpret_cast<const f16x8*>(wt + sg * S_C + 16 * ks);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the weights are in: from here on vmcnt holds tile rows and output stores only
#pragma unroll
  for (int j = 0; j < D - 1; ++j) issue(j);

  // D[i = column][j = row]: lane (row li), registers r -> column 32 w + 8 (r / 4) + 4 kh + r % 4
  const float* const bias = shl + 32 * wave + 4 * kh;   // + 8 q: the four columns of register group q
  const float* const resa = rsl + 32 * wave + 4 * kh;
  const bool res = p.res_a != nullptr;                  // + sig[b][t * res_stride] * res_a[column] (gemm.hip gemm_epilogue_lean<RES>)
  const float relu_lo = p.relu != 0 ? 0.f : -INFINITY;
  // B operand of k-step ks: logical octet 2 ks + kh of row li, at position (2 ks + kh) ^ (li & 15): one XOR of the lane's base
  const unsigned rd0 = (unsigned)li * (S_C * 2) + ((((unsigned)li & 14u) << 4) | (((unsigned)kh ^ ((unsigned)li & 1u)) << 4));   // bytes
  // staging write: 8 bytes (columns 32 w + 8 q + 4 kh ..) = half kh of piece 4 w + q of row li
  const unsigned wr0 = (unsigned)li * (S_C * 2) + (unsigned)kh * 8u;
  // copy-out: store i of this wave = rows 4 w + 2 i + kh, piece li
  auto copy_out = [&](int jt, f16x8* v) {   // reads the staging tile of tile jt (issued early, consumed by store_out)
    const char* sb = reinterpret_cast<const char*>(stage + (jt & 1) * S_STAGE_H);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned r = 4 * wave + 2 * i + kh;
      v[i] = *reinterpret_cast<const f16x8*>(sb + r * (S_C * 2) + (((unsigned)li ^ (r & 15u)) << 4));
    }
  };
  auto store_out = [&](int jt, const f16x8* v) {
    const int m0 = (first + jt * step) * S_ROWS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + 4 * wave + 2 * i + kh;
#if CHIRON_S16_VARIANT & 1
      if (v[i][0] == (_Float16)123.25f)
#endif
      if (m < p.M) *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(p.out) + (long)m * p.ldo + 8 * li) = v[i];
    }
  };
  // The barrier of a tile: this wave's rows of the tile have landed (vmcnt, see above), its LDS writes to the staging tile are
  // complete (lgkmcnt), then s_barrier.  The compiler does not know that the DMA engine writes the tiles: tile_barrier()'s
  // fences keep every LDS read of a tile behind the barrier that publishes it.
  auto stage_barrier = [&]() {
    constexpr int PENDING = (D - 2) * IPW;
    __builtin_amdgcn_s_waitcnt(0x0070 | (PENDING & 15) | ((PENDING >> 4) << 14));   // vmcnt(PENDING) lgkmcnt(0)
    tile_barrier();
  };
  for (int j = 0; j < mine; ++j) {
    stage_barrier();                       // tile j has landed; every wave has finished tile j - 1 (and staged its columns)
    issue(j + D - 1);                      // into the buffer of tile j - 1
    f16x8 ov[2];
    if (!PRIV && j > 0) copy_out(j - 1, ov);
    const char* base = reinterpret_cast<const char*>(tiles + (j % D) * NSEG * S_TILE_H);
    // the accumulator starts from the folded BN offsets of its columns (read from the LDS: no VALU; every VALU instruction of
    // a wave64 costs 4 SIMD cycles and the two waves of a SIMD run their epilogues at the same time, after their products)
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 8 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[4 * q + r] = b4[r];
    }
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
      f16x8 xb[16];   // the whole K-segment of this lane's row: sixteen reads in flight, the products follow as they arrive
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const f16x8*>(base + sg * (S_TILE_H * 2) + (rd0 ^ (unsigned)(ks * 32)));
      asm volatile("" ::: "memory");   // all sixteen reads are issued before the first product waits for one (the scheduler pairs them up otherwise)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[sg][ks], xb[ks], acc, 0, 0, 0);
    }
    if (!PRIV && j > 0) store_out(j - 1, ov);
    const int m = (first + j * step) * S_ROWS + li;
    if (res) {   // res_layer1: + sig[b][t * res_stride] * res_a[column]
      float sv = 0.f;
      if (m < p.M) {
        const int b = m / p.T_out, t = m - b * p.T_out;
        sv = p.sig[(long)b * p.L + (long)t * p.res_stride];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(resa + 8 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[4 * q + r] = fmaf(sv, r4[r], acc[4 * q + r]);   // f16 engines: the branch's shift is folded into
                                                                                         // the accumulator's start (engine.hip: res_b = 0), ONE fused rounding as in round 4
      }
    }
    char* sw = reinterpret_cast<char*>(stage + (j & 1) * S_STAGE_H) + wr0;
    char* const pbase = reinterpret_cast<char*>(stage) + wave * 2048;      // PRIV: [32 rows][4 pieces of 16 bytes], piece q of row r at q ^ (r / 4 % 4)
    char* const pw0 = pbase + (unsigned)li * 64u + (unsigned)kh * 8u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f16x4 hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hv[r] = (_Float16)__builtin_amdgcn_fmed3f(acc[4 * q + r], relu_lo, INFINITY);   // one VALU instruction, ReLU or not
      }
      if (PRIV) *reinterpret_cast<f16x4*>(pw0 + (((unsigned)q ^ (((unsigned)li >> 2) & 3u)) << 4)) = hv;
      else *reinterpret_cast<f16x4*>(sw + ((((unsigned)(4 * wave + q)) ^ ((unsigned)li & 15u)) << 4)) = hv;
    }
    if (PRIV) {   // this wave's block: rows 16 i + lane / 4, 16-byte piece lane % 4 of its 64-byte row segment
      const int m0 = (first + j * step) * S_ROWS;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned r = 16 * i + (lane >> 2), pc = lane & 3;
        const f16x8 v = *reinterpret_cast<const f16x8*>(pbase + r * 64 + ((pc ^ ((r >> 2) & 3u)) << 4));
#if CHIRON_S16_VARIANT & 1
        if (v[0] == (_Float16)123.25f)
#endif
        if (m0 + (int)r < p.M) *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(p.out) + (long)(m0 + r) * p.ldo + 32 * wave + 8 * pc) = v;
      }
    }
  }
  if (!PRIV) {
    stage_barrier();
    f16x8 ov[2];
    copy_out(mine - 1, ov);
    store_out(mine - 1, ov);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the tiles past the last one (zeros) before the workgroup's LDS is released
}


// ---------------------------------------------------------------------------------------------------------
// The 1 x 3 convolution (branch2/conv2b, stride 1, 256 -> 256) in the same style.  Its weights are 384 KB of halves -- more
// than a CU's registers can spare -- so the output columns are split over TWO workgroups (128 columns each: 96 weight
// registers per lane) that stream the same rows; workgroups i and i + 8 form such a pair, i.e. they sit on the same XCD
// and the second reader of a row finds it in that XCD's L2.  v_mfma_f32_16x16x32_f16, transposed product: wave w owns 16
// columns (A operand = its W^T slice for the three taps, 3 x 8 k-steps x 4 registers), B = 16 rows x 32 k from the LDS.
// Round 3 -- shift the PRODUCTS, not the inputs.  y[r] = W0^T x[r-1] + W1^T x[r] + W2^T x[r+1]: round 2 read the B fragment
// of rows r-1, r, r+1 separately (48 ds_read_b128 per wave and tile, 384 KB per workgroup and tile: 3072 LDS cycles against
// 1536 matrix-pipe cycles per SIMD -- the kernel ran at LDS bandwidth, 0.86 ms per launch at B = 4096 for 0.26 ms of MFMA).
// Now every fragment is read ONCE and multiplied by the three taps' weights into three accumulators P0, P1, P2 (per 16-row
// block); rows are lanes of the accumulator (lane & 15), so the shift by one row is a DPP rotate within the 16-lane row:
//   tile = 32 consecutive rows (slots 0..31 = rows m0 - 1 .. m0 + 30), two 16-row blocks, 30 valid outputs (slots 1..30);
//   block 0: y[n] = ror(P0_0)[n] + P1_0[n] + (n == 15 ? rol(P2_1) : rol(P2_0))[n]       (lane 0 = slot 0: not an output)
//   block 1: y[n] = (n == 0 ? ror(P0_0) : ror(P0_1))[n] + P1_1[n] + rol(P2_1)[n]        (lane 15 = slot 31: not an output)
//   (ror: lane n receives lane n - 1 mod 16; the wrapped lane is exactly the neighbour block's edge row);
//   a row at the first / last position of its sequence drops the tap-0 / tap-2 term (SAME padding).
// 16 reads and 48 MFMAs per wave and tile (2 of 32 rows are recomputed by the next tile: 6 %); tiles, DMA and the staged
// whole-row stores as in the 1 x 1 kernel above (row-major XOR-swizzled tiles, eight self-feeding waves).
// ---------------------------------------------------------------------------------------------------------
constexpr int S3_VALID = S_ROWS - 2;             // output rows per tile
constexpr int S3_D = 6;                          // tiles resident
constexpr int S3_STAGE_H = S_ROWS * (S_C / 2);   // halves per staging tile: 32 rows x 128 columns (8 KB)

template <int CTRL>
static __device__ __forceinline__ f32x4 row_shift(f32x4 old, f32x4 v) {   // 0x111: lane n <- n - 1, 0x101: lane n <- n + 1; the lane without a source keeps `old`
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old[i]), __float_as_int(v[i]), CTRL, 0xF, 0xF, false));
  return r;
}
template <int CTRL>
static __device__ __forceinline__ f32x4 row_rotate(f32x4 v) {   // DPP within each 16-lane row; 0x121: lane n <- n - 1, 0x12F: lane n <- n + 1
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), CTRL, 0xF, 0xF, false));
  return r;
}

__global__ __launch_bounds__(64 * S_NW, 1) void conv3_f16_stream_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) _Float16 tiles[];   // [S3_D][32 slots][32 octets][8 halves], then [2] staging tiles
  _Float16* const stage = tiles + S3_D * S_TILE_H;
  __shared__ __attribute__((aligned(16))) float shl[S_C];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (p.M < 0) tiles[tid] = (_Float16)0.f;
  if (tid < S_C) shl[tid] = p.shift ? p.shift[tid] : 0.f;
  __syncthreads();

  const int half = (blockIdx.x >> 3) & 1;                          // column half of this workgroup
  const int stream_id = (blockIdx.x & 7) | ((blockIdx.x >> 4) << 3);   // which of the gridDim.x / 2 row streams
  const int nstreams = gridDim.x >> 1;
  const int ntiles = (p.M + S3_VALID - 1) / S3_VALID;
  const int mine = stream_id < ntiles ? (ntiles - stream_id + nstreams - 1) / nstreams : 0;
  if (mine == 0) return;

  // ---------------- input: wave w fetches slots 4 w .. 4 w + 3 of every tile (two DMA instructions, two whole rows each)
  const __amdgpu_buffer_rsrc_t rs = s_rsrc(p.seg[0].src);
  const int li = lane & 31, kh = lane >> 5;
  auto issue = [&](int j) {
    const int mfirst = (stream_id + j * nstreams) * S3_VALID - 1;   // row of slot 0
    _Float16* base = tiles + (j % S3_D) * S_TILE_H;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 2 * (wave * 2 + i) + kh;
      const int m = mfirst + r;
#if CHIRON_S16_VARIANT & 2
      const bool ok = false;
#else
      const bool ok = j < mine && m >= 0 && m < p.M;
#endif
      const unsigned off = ok ? (unsigned)(((long)m * p.seg[0].lda + p.seg[0].col0) * 2 + ((li ^ (r & 15)) * 16)) : S_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(base + (wave * 2 + i) * 2 * S_C), 16, off, 0, 0, 0);
    }
  };

  // ---------------- weights: wave w owns output columns 128 half + 16 w .. + 15
  const int n = lane & 15, kg = lane >> 4;
  const int col0 = 128 * half + 16 * wave;
  f16x8 wr[3][8];
  {
    const _Float16* wt = reinterpret_cast<const _Float16*>(p.Wt) + (long)(col0 + n) * p.K + 8 * kg;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) wr[tap][ks] = *reinterpret_cast<const f16x8*>(wt + tap * S_C + 32 * ks);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the weights are in: from here on vmcnt holds tile rows and output stores only
#pragma unroll
  for (int j = 0; j < S3_D - 1; ++j) issue(j);

  // D[i = column][row]: lane (row n of the block), registers r -> column col0 + 4 kg + r
  const f32x4 bias = *reinterpret_cast<const f32x4*>(shl + col0 + 4 * kg);
  const float relu_lo = p.relu != 0 ? 0.f : -INFINITY;
  const int T = p.T_out;
  // B fragment of k-step ks, block h: logical octet 4 ks + kg of slot 16 h + n, at position (4 ks + kg) ^ n
  const unsigned rd0 = (unsigned)n * (S_C * 2) + ((((unsigned)n & 12u) << 4) | ((((unsigned)kg) ^ ((unsigned)n & 3u)) << 4));   // bytes; + h * 8192, ^ ks * 64
  // staging write: slot 16 h + n, 8 bytes = half (kg & 1) of piece 2 w + kg / 2 (of 16 per 128-column row), at position piece ^ n
  const unsigned wr0 = (unsigned)n * 256u + ((((unsigned)(2 * wave + (kg >> 1))) ^ (unsigned)n) << 4) + ((unsigned)kg & 1u) * 8u;
  // copy-out: thread -> slot tid / 16, piece tid % 16
  const unsigned crow = tid >> 4, cpc = tid & 15;
  const unsigned crd = crow * 256u + ((cpc ^ (crow & 15u)) << 4);
  auto stage_barrier = [&]() {
    constexpr int PENDING = (S3_D - 2) * 2;
    __builtin_amdgcn_s_waitcnt(0x0070 | (PENDING & 15) | ((PENDING >> 4) << 14));   // vmcnt(PENDING) lgkmcnt(0)
    tile_barrier();
  };
  auto store_out = [&](int jt, const f16x8& v) {
    const int m = (stream_id + jt * nstreams) * S3_VALID - 1 + (int)crow;
#if CHIRON_S16_VARIANT & 1
    if (v[0] == (_Float16)123.25f)
#endif
    if (crow >= 1 && crow <= (unsigned)S3_VALID && m < p.M)
      *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(p.out) + (long)m * p.ldo + 128 * half + 8 * cpc) = v;
  };

  // sequence position of slot 0 of this workgroup's tile j, carried from tile to tile (scalar arithmetic: no per-lane division)
  int tpos0 = (int)(((long)stream_id * S3_VALID - 1 + T) % T);
  const int tstep = (int)(((long)nstreams * S3_VALID) % T);

  // ---- finishing a tile: rows are lanes, so the tap-0 / tap-2 products move by one row with DPP.  A = ror(P0 of block 0): lane
  //      n <- n - 1, lane 0 <- 15; block 1's tap-0 term is shr(P0 of block 1) with lane 0 KEEPING A's value (DPP leaves a lane
  //      without a source at `old`): exactly block 0's row 15.  Mirror image for tap 2.  A row at the first / last position of
  //      its sequence multiplies the tap-0 / tap-2 term by 0 (SAME padding).  Branch-free on purpose: 16 DPP moves, 16 FMAs,
  //      8 med3, 4 packed converts per wave and tile.
  auto finish = [&](const f32x4 (&Q)[2][3], int tp0, int jt) {
    const f32x4 A = row_rotate<0x121>(Q[0][0]), E = row_rotate<0x12F>(Q[1][2]);
    const f32x4 t01 = row_shift<0x111>(A, Q[1][0]), t20 = row_shift<0x101>(E, Q[0][2]);
    char* sw = reinterpret_cast<char*>(stage + (jt & 1) * S3_STAGE_H) + wr0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int tp = tp0 + 16 * h + n;       // a tile spans 32 rows < T: at most one wrap inside it
      tp = tp >= T ? tp - T : tp;
      const float k0 = tp == 0 ? 0.f : 1.f, k2 = tp == T - 1 ? 0.f : 1.f;
      const f32x4 t0 = h == 0 ? A : t01, t2 = h == 0 ? t20 : E;
      f16x4 hv;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        hv[r] = (_Float16)__builtin_amdgcn_fmed3f(fmaf(t2[r], k2, fmaf(t0[r], k0, Q[h][1][r])), relu_lo, INFINITY);   // ReLU or not: one instruction
      *reinterpret_cast<f16x4*>(sw + h * 4096) = hv;
    }
  };
  // ---- one interval: the products of tile j (matrix pipe) and, in the same basic block, the finishing of tile j - 1 (VALU, LDS
  //      write) -- two waves share a SIMD in lockstep, so a wave's own VALU work has to sit under its own MFMAs; the copy of tile
  //      j - 2 leaves the staging tile for HBM.  (Round 3 first form: products, THEN finishing: 1.65 us per tile for 0.64 us of MFMA.)
  auto interval = [&](int j, f32x4 (&P)[2][3], const f32x4 (&Q)[2][3]) {
    stage_barrier();                       // tile j has landed; every wave has finished tile j - 1 and staged tile j - 2
    issue(j + S3_D - 1);
    const f16x8 ov = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(stage + (j & 1) * S3_STAGE_H) + crd);   // tile j - 2
    const char* tb = reinterpret_cast<const char*>(tiles + (j % S3_D) * S_TILE_H);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) P[h][tap] = tap == 1 ? bias : (f32x4){0.f, 0.f, 0.f, 0.f};
      f16x8 xb[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) xb[ks] = *reinterpret_cast<const f16x8*>(tb + h * 8192 + (rd0 ^ (unsigned)(ks * 64)));
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#if CHIRON_S16_VARIANT & 4
          P[h][tap][0] += (float)wr[tap][ks][0] * (float)xb[ks][0];
#else
          P[h][tap] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[tap][ks], xb[ks], P[h][tap], 0, 0, 0);
#endif
    }
    int tpp = tpos0 - tstep;               // slot 0 of tile j - 1
    tpp = tpp < 0 ? tpp + T : tpp;
    finish(Q, tpp, j - 1);                 // j = 0: zeros into staging tile 1, never copied out
    if (j >= 2) store_out(j - 2, ov);
    tpos0 += tstep;
    tpos0 = tpos0 >= T ? tpos0 - T : tpos0;
  };
  f32x4 PA[2][3], PB[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) PB[h][tap] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < mine; j += 2) {
    interval(j, PA, PB);
    if (j + 1 < mine) interval(j + 1, PB, PA);
  }
  // ---- drain: tile mine - 1 is finished, tiles mine - 2 and mine - 1 are copied out
  stage_barrier();
  {
    const f16x8 ov = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(stage + (mine & 1) * S3_STAGE_H) + crd);   // tile mine - 2
    int tpp = tpos0 - tstep;
    tpp = tpp < 0 ? tpp + T : tpp;
    if (mine & 1) finish(PA, tpp, mine - 1);
    else finish(PB, tpp, mine - 1);
    if (mine >= 2) store_out(mine - 2, ov);
  }
  stage_barrier();
  {
    const f16x8 ov = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(stage + ((mine - 1) & 1) * S3_STAGE_H) + crd);
    store_out(mine - 1, ov);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the tiles past the last one (zeros) before the workgroup's LDS is released
}

// GemmParams in ELEMENT units (halves).  false: shape not covered, the caller takes gemm.hip.
bool launch_stream16(const GemmParams& p, hipStream_t stream) {
  if (p.out_mode != 0 || p.N != S_C || p.m_time_major || (p.res_a != nullptr && (p.sig == nullptr || p.nseg != 1))) return false;
  for (int i = 0; i < p.nseg; ++i)
    if (p.seg[i].src == nullptr) return false;   // lifted segments (A computed from the signal) stay with gemm.hip
  if (p.nseg < 1 || p.nseg > 3 || p.K != p.nseg * S_C) return false;
  const bool taps = p.nseg == 3;   // conv2b: the three taps of one tensor
  if (taps && p.T_out < S_ROWS) return false;   // conv3's position bookkeeping assumes at most one sequence boundary per 32-row tile
  for (int i = 0; i < p.nseg; ++i) {
    const GemmSeg& s = p.seg[i];
    if (s.src == nullptr || s.cin != S_C || s.kpad != S_C || s.stride != 1 || s.time_major || s.w_in != p.T_out) return false;
    if (s.shift != (taps ? i - 1 : 0) || (taps && (s.src != p.seg[0].src || s.lda != p.seg[0].lda || s.col0 != p.seg[0].col0))) return false;
    if ((size_t)p.M * (size_t)s.lda * 2 > S_RECORDS) return false;
  }
  const int n_cu = current_device_cus();
  static char attr_state[CHIRON_MAX_DEVICES] = {};   // 0 unknown, 1 opted in, 2 refused -- per device
  const int dev = current_device_index();
  char st = __atomic_load_n(&attr_state[dev], __ATOMIC_ACQUIRE);
  if (st == 0) {
    const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_f16_stream_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (6 * 1 * S_TILE_H + 2 * S_STAGE_H) * 2) == hipSuccess &&
              hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_f16_stream_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (4 * 2 * S_TILE_H + S_STAGE_H) * 2) == hipSuccess &&
              hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_f16_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (S3_D * S_TILE_H + 2 * S3_STAGE_H) * 2) == hipSuccess;
    if (!attr_ok) (void)hipGetLastError();   // not sticky: the caller falls back to gemm.hip
    st = attr_ok ? 1 : 2;
    __atomic_store_n(&attr_state[dev], st, __ATOMIC_RELEASE);
  }
  if (st != 1) return false;
  const int ntiles = (p.M + S_ROWS - 1) / S_ROWS;
  if (ntiles <= 0) return true;   // nothing to do
  const int grid = std::min(n_cu, ntiles);
  if (taps) {
    // pairs of workgroups (i, i + 8) share a row stream: the grid is a whole number of 16-workgroup blocks
    const int ntiles3 = (p.M + S3_VALID - 1) / S3_VALID;
    const int g3 = std::max(16, (std::min(n_cu, 2 * ntiles3 + 15) / 16) * 16);
    hipLaunchKernelGGL(conv3_f16_stream_kernel, dim3(g3), dim3(64 * S_NW), (size_t)(S3_D * S_TILE_H + 2 * S3_STAGE_H) * 2, stream, p);
    return true;
  }
  if (p.nseg == 1)
    hipLaunchKernelGGL(conv1x1_f16_stream_kernel<1>, dim3(grid), dim3(64 * S_NW), (size_t)(6 * 1 * S_TILE_H + 2 * S_STAGE_H) * 2, stream, p);
  else
    hipLaunchKernelGGL(conv1x1_f16_stream_kernel<2>, dim3(grid), dim3(64 * S_NW), (size_t)(4 * 2 * S_TILE_H + S_STAGE_H) * 2, stream, p);
  return true;
}

}  // namespace chiron
