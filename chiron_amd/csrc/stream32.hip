// fp32 engine: the 256 -> 256 channel 1 x 1 convolutions of the residual blocks (cnn.py:234-262: branch2/conv2a of
// res_layer2 / res_layer3, branch2/conv2c of res_layer1 with its signal-synthesised branch1) as a WEIGHT-STATIONARY
// streaming kernel -- the structure of stream16.hip at exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Why.  PMC of the tiled GEMM (gemm.hip, round 2): 0.146 LDS-DMA instructions and 0.44 VALU instructions per MFMA; a DMA
// instruction issued by a wave that shares its SIMD with fp32 MFMAs costs about 70 matrix-pipe cycles
// (tools/ubench/mfma_vmem.hip), and half of those instructions re-fetch the WEIGHT tile, 256 KB that never changes.  Here
//   * the whole weight matrix lives in VGPRs for the lifetime of the persistent workgroup (one per CU, eight waves): wave w
//     owns output columns 32w .. 32w+31 for every k -- 128 registers (A operand: lane (column li, kh) holds
//     W[32w + li][8g + 4kh + j], g = 0..31, j = 0..3); the product is computed transposed, D = W^T x^T, so that a lane ends up
//     with 4 x 4 CONSECUTIVE output channels of one position and stores 16-byte pieces of a channels-last row;
//   * the LDS holds nothing but activation tiles: 32 positions x 256 channels = 32 KB, laid out [16-byte piece q = k / 4]
//     [position][4 floats] = the B-operand order (lane (position li, kh) reads piece 2g + kh: lane-linear 16-byte reads, each
//     feeding four MFMAs), FOUR tiles deep (96 KB in flight behind the one being consumed);
//   * every wave issues its four LDS-DMA instructions of tile j + 3 right after the barrier of tile j: 0.03 per MFMA (the
//     tiled kernel: 0.15) -- and keeps TWO accumulators, even and odd 8-channel groups of k, so that consecutive MFMAs of a
//     wave never depend on each other.
// Measured: 0.407 ms per 1100-window batch in the steady state (142 TFLOP/s = 0.90 of the fp32 MFMA peak; the
// tiled kernel: 0.45) plus 0.04 ms per launch (256 workgroups x 256 KB of weights through the L2, pipeline fill):
// conv2a 0.450 ms instead of 0.465, res_layer1's conv2c 0.463 instead of 0.505.  Two forms that were built and dropped:
// dedicated producer waves as in stream16.hip (10 waves: three on two SIMDs, 168 registers, ONE accumulator chain: 0.483 ms),
// and the x-projections on the same scheme (N = 800 = 25 column groups over three workgroup types that share the activation
// tiles, the 25th group K-split over eight waves: 1.53 / 1.28 ms against the tiled kernel's 1.42 / 1.16 -- a single type runs
// at the convolution's rate, the shared tiles and the uneven types cost 7 - 17 %).
// Arithmetic: the same products as gemm.hip, accumulated over k in another order -- results agree to rounding (tests hold
// both to the float64 oracle at 1e-4).
#include "kernels.h"

#include <algorithm>

namespace chiron {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3)))* lptr_t;

constexpr unsigned T_OOB = 0xFFFF0000u;      // byte offset past num_records: the DMA writes zeros
constexpr unsigned T_RECORDS = 0xFFFE0000u;  // every tensor of the engine is smaller than this many bytes
constexpr int T_ROWS = 32;                   // positions per tile
constexpr int T_C = 256;                     // channels in = output columns
constexpr int T_TILE_F = T_ROWS * T_C;       // floats per tile (32 KB)
constexpr int T_D = 4;                       // tiles resident in the LDS
constexpr int T_NW = 8;                      // waves per workgroup

// s_barrier between producer and consumer waves (see stream16.hip): compiler-only fences around the hardware barrier; no
// s_waitcnt vmcnt, which would make every compute wave wait for its own output stores once per tile.
static __device__ __forceinline__ void tile_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t t_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, T_RECORDS, 0x00027000);
}

// vmcnt counts a wave's loads and stores together and the two kinds may retire out of order, so "tile j has landed" is
// waited for conservatively: at most 8 operations outstanding (the loads of tiles j + 1 and j + 2 are the 8 youngest loads;
// loads retire in order).
template <bool RES>
__global__ __launch_bounds__(64 * T_NW, 1) void conv1x1_f32_stream_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float tiles[];   // [T_D][64 pieces][32 positions][4 floats]
  __shared__ __attribute__((aligned(16))) float shl[T_C];
  __shared__ __attribute__((aligned(16))) float rsl[T_C];
  __shared__ __attribute__((aligned(16))) float rbl[T_C];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  if (p.M < 0) tiles[tid] = 0.f;
  if (tid < T_C) shl[tid] = p.shift ? p.shift[tid] : 0.f;
  if (tid < T_C) rsl[tid] = (RES && p.res_a) ? p.res_a[tid] : 0.f;
  if (tid < T_C) rbl[tid] = (RES && p.res_b) ? p.res_b[tid] : 0.f;
  __syncthreads();

  const int ntiles = (p.M + T_ROWS - 1) / T_ROWS;
  const int first = blockIdx.x, step = gridDim.x;
  // Tile ids.  Fixed shares: position j of this workgroup is tile first + j * step.  With the stream's counter
  // (GemmParams::tile_ctr; one counter: rows have no XCD affinity here, the weights sit in registers) every position's tile is
  // the next number of the counter.  Thread 0 requests the number of position j + 4 at the top of iteration j and publishes it
  // in an LDS ring at the END of the same iteration -- behind the iteration's DMA and stores, so that the wait the compiler puts
  // in front of the use is a vmcnt(8) that the request has long satisfied (consumed in the NEXT iteration the value is copied
  // at the loop's back edge and the wait lands right behind the request: +9 % per launch) -- and everybody reads it after the
  // next barrier, when the DMA of position j + 4 is due.  A workgroup stops at its first number past the last tile and has
  // then taken exactly four such numbers: a launch moves the counter by ntiles + 4 * gridDim.x.
  __shared__ int ring[8];
  const bool dyn = p.tile_ctr != nullptr;
  auto take = [&]() -> unsigned long long { return atomicAdd(p.tile_ctr, 1ull); };
  auto tile_id = [&](unsigned long long v) -> int {
    const unsigned long long k = v - p.tile_base;
    return k >= (unsigned long long)ntiles ? ntiles : (int)k;
  };
  int cur[4];   // tiles of positions j .. j + 3 (wave-uniform)
  if (dyn) {
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) ring[k] = tile_id(take());
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) cur[k] = __builtin_amdgcn_readfirstlane(ring[k]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) cur[k] = first + k * step < ntiles ? first + k * step : ntiles;
  }
  if (cur[0] >= ntiles) return;

  const __amdgpu_buffer_rsrc_t rs = t_rsrc(p.seg[0].src);
  auto issue = [&](int j, int tile) {   // this wave's 8 of the 64 pieces of position j's tile (past the last tile: zeros, no memory traffic)
    const int m = tile * T_ROWS + li;
    const bool ok = tile < ntiles && m < p.M;
    float* base = tiles + (j % T_D) * T_TILE_F;
    const unsigned off = ok ? (unsigned)(((long)m * p.seg[0].lda + p.seg[0].col0) * 4 + kh * 16) : T_OOB;
#pragma unroll
    for (int o = 0; o < 8; o += 2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(base + (wave * 8 + o) * T_ROWS * 4), 16, off + (unsigned)(wave * 8 + o) * 16u, 0, 0, 0);
  };

  f32x4 wr[32];
  {
    const float* wt = p.Wt + (long)(32 * wave + li) * p.K + 4 * kh;
#pragma unroll
    for (int g = 0; g < 32; ++g) wr[g] = *reinterpret_cast<const f32x4*>(wt + 8 * g);
  }
  const float* const bias = shl + 32 * wave + 4 * kh;
  const float* const resa = rsl + 32 * wave + 4 * kh;
  const float* const resb = rbl + 32 * wave + 4 * kh;
  float* const outp = p.out + 32 * wave + 4 * kh;
  const bool relu = p.relu != 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weights are in: from here on vmcnt counts tiles and stores only
#pragma unroll
  for (int j = 0; j < T_D - 1; ++j) issue(j, cur[j]);

  for (int j = 0; cur[0] < ntiles; ++j) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's pieces of tile j have landed (thread 0's counter request makes
                                                        // its wave's wait stricter, never looser)
    tile_barrier();                                     // everybody's have; everybody has finished tile j - 1
    unsigned long long req;   // (left undefined for the other threads on purpose: an initial value makes the compiler copy the result -- and wait for it -- inside the branch)
    if (dyn) {
      // (plain LDS accesses between the barrier's compiler fences: through a volatile pointer they become FLAT instructions, which
      //  count in vmcnt as well and drag a vmcnt(0) into every wave's iteration)
      if (j > 0) cur[3] = __builtin_amdgcn_readfirstlane(ring[(j + 3) & 7]);
      // position j + 4.  Hand-issued: hipcc's atomic optimiser turns a builtin atomic in a divergent branch into one that needs its
      // result at once (readfirstlane), i.e. an s_waitcnt vmcnt(0) right behind the request, in a wave with three tiles in flight
      if (tid == 0) {
        const unsigned zero = 0;
        const unsigned long long one = 1;
        asm volatile("global_atomic_add_x2 %0, %1, %2, %3 sc0" : "=v"(req) : "v"(zero), "v"(one), "s"(p.tile_ctr) : "memory");
      }
    } else if (j > 0) {
      cur[3] = first + (j + 3) * step < ntiles ? first + (j + 3) * step : ntiles;
    }
    issue(j + T_D - 1, cur[3]);                         // ... whose buffer tile j + 3 goes to
    const float* base = tiles + (j % T_D) * T_TILE_F + (kh * T_ROWS + li) * 4;
    f32x16 acc0, acc1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 8 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc0[4 * q + r] = b4[r];
    }
#pragma unroll
    for (int g = 0; g < 32; g += 2) {
      const f32x4 xa = *reinterpret_cast<const f32x4*>(base + g * 2 * T_ROWS * 4);
      const f32x4 xb = *reinterpret_cast<const f32x4*>(base + (g + 1) * 2 * T_ROWS * 4);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[g][jj], xa[jj], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[g + 1][jj], xb[jj], acc1, 0, 0, 0);
      }
    }
    const int m = cur[0] * T_ROWS + li;
    cur[0] = cur[1], cur[1] = cur[2], cur[2] = cur[3];
    if (m < p.M) {
      float* o = outp + (long)m * p.ldo;
      float sv = 0.f;
      if (RES) {
        const int b = m / p.T_out, t = m - b * p.T_out;
        sv = p.sig[(long)b * p.L + (long)t * p.res_stride];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc0[4 * q] + acc1[4 * q], acc0[4 * q + 1] + acc1[4 * q + 1], acc0[4 * q + 2] + acc1[4 * q + 2], acc0[4 * q + 3] + acc1[4 * q + 3]};
        if (RES) {
          // the signal branch as ONE fmaf (exact value small: sv*a cancels against b), added to the finished sum (kernels.h res_b)
          const f32x4 r4 = *reinterpret_cast<const f32x4*>(resa + 8 * q);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(resb + 8 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += fmaf(sv, r4[r], b4[r]);
        }
        if (relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], 0.f, INFINITY);
        }
        *reinterpret_cast<f32x4*>(o + 8 * q) = v;
      }
    }
    if (dyn && tid == 0) {
      // the request is older than this iteration's four DMA instructions (and its stores): at most four operations outstanding
      // means it has returned (returning operations retire in order)
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(req) : : "memory");
      ring[(j + 4) & 7] = tile_id(req);
    }
  }
}

bool launch_stream32(const GemmParams& p, hipStream_t stream) {
  // 256 -> 256 channel 1 x 1 convolution, fp32, channels-last in and out, no K tail
  if (p.f16 != 0 || p.out_mode != 0 || p.m_time_major || p.N != T_C || p.K != T_C || p.nseg != 1 || p.M <= 0) return false;
  const GemmSeg& s = p.seg[0];
  if (s.src == nullptr || s.cin != T_C || s.kpad != T_C || s.stride != 1 || s.shift != 0 || s.time_major || s.w_in != p.T_out) return false;
  if ((s.lda & 3) || (s.col0 & 3) || (p.ldo & 3)) return false;
  if ((size_t)p.M * (size_t)s.lda * 4 > T_RECORDS) return false;
  const bool res = p.res_a != nullptr;
  if (res && (p.sig == nullptr || p.T_out <= 0)) return false;
  const int n_cu = current_device_cus();
  static char attr_state[CHIRON_MAX_DEVICES] = {};   // 0 unknown, 1 opted in, 2 refused -- per device
  const int dev = current_device_index();
  char st = __atomic_load_n(&attr_state[dev], __ATOMIC_ACQUIRE);
  const size_t lds = (size_t)T_D * T_TILE_F * 4;
  if (st == 0) {
    const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_f32_stream_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_f32_stream_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!ok) (void)hipGetLastError();   // not sticky: the caller falls back to gemm.hip
    st = ok ? 1 : 2;
    __atomic_store_n(&attr_state[dev], st, __ATOMIC_RELEASE);
  }
  if (st != 1) return false;
  const int ntiles = (p.M + T_ROWS - 1) / T_ROWS;
  const int grid = std::min(n_cu, ntiles);
  GemmParams q = p;
  const bool dyn = q.tile_ctr != nullptr && q.tile_base_host != nullptr;
  if (dyn) q.tile_base = *q.tile_base_host;
  else q.tile_ctr = nullptr;
  if (res)
    hipLaunchKernelGGL(conv1x1_f32_stream_kernel<true>, dim3(grid), dim3(64 * T_NW), lds, stream, q);
  else
    hipLaunchKernelGGL(conv1x1_f32_stream_kernel<false>, dim3(grid), dim3(64 * T_NW), lds, stream, q);
  if (dyn) *q.tile_base_host += (unsigned long long)ntiles + 4ull * (unsigned long long)grid;   // what the launch takes from its counter
  return true;
}

}  // namespace chiron
