// LSTM recurrence for gfx950: the TF while_loop of LSTMCell(100) under dynamic_rnn with
// sequence_length (chiron/rnn.py:49-65 / :140-145; op composition recorded in the shipped .meta
// graphs) as ONE persistent launch per layer.
//
//   workgroup = 4 or 8 batch rows (one or two 4-row groups) x 1 direction, resident for all T steps (rows never
//   interact, so no grid-wide sync exists).  7 waves per group; heavy wave w owns hidden units [16w, 16w+16) for all
//   four gates = 64 matrix columns = exactly the N extent of ONE v_mfma_f32_4x4x1_16B_f32 (16 blocks of a 4x4 outer
//   product: M = 4 batch rows, N = 64 columns, K = 1).  Its slice of W_hh lives in VGPRs for the whole sequence
//   (100 registers per lane; W_hh = 160 KB fp32 = the entire LDS, spread over the waves' register files).
//   The 4-row MFMA shape is what lets B = 1100 fill the chip: 16-row tiles give 138 workgroups for 256 CUs, 4-row
//   groups give 550 group-directions.  The price: a 4x4x1 MFMA occupies the matrix pipe for 10 cycles, not the 8 of its
//   two passes (tools/ubench/mfma4x4_rate.hip, mfma_valu_overlap.hip): 51 instead of 64 FLOP/clk/SIMD.
//   (v_mfma_f32_* shares the fp32 VALU datapath -- tools/ubench/barrier_mfma.hip -- so the gate math can
//   not hide behind it; what counts is total issue time, and this layout needs ONE cell per lane.)
//
//   per step and row group:  acc[4 rows] (lane = gate*16 + unit)  <-  h_{t-1} . W_hh      100 MFMAs
//                            + z_t (x-projection; gemm.hip stores it [col = gate*H + unit][4 rows], one 16-byte load per lane)
//                            4x4 register transpose (lane swaps): lane (row, unit) gets i, j, f, o of its cell
//                            c = sig(f)*c + sig(i)*tanh(j);  h = sig(o)*tanh(c)   (forget bias folded in z)
//   h is exchanged through a double-buffered 2 KB LDS tile per group (A-operand order, see HG), one barrier per step.
//   Masking: rows with t >= seq_len emit 0 and carry (c,h); the backward direction walks
//   t = seq_len-1-s per row (tf.reverse_sequence folded into index arithmetic, no copy): its z is stored by step
//   by the projection GEMM, its output frame index is per-lane arithmetic.
#include "kernels.h"
#include "timing_variants.h"

#include <algorithm>

// Every multiply-add below is written out (fmaf or separate ops) so that a row's result does not depend on
// which register slot / row position it occupies: batches can be re-packed without changing a bit.
#pragma clang fp contract(off)

namespace chiron {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// h_{t-1} of one 4-row group in LDS, in the order the MFMA A operand wants it: lane (blk = lane>>2, row = lane&3)
// keeps h[row][16q + blk], q = 0..6.  The 4x4x1 MFMA for k = 16q + blk then BROADCASTS block blk's A vector to all 16
// blocks (cbsz = 4, abid = blk; semantics checked in tools/ubench/mfma4x4_bcast.hip), so a wave reads 2 KB of h per
// step instead of 25.6 KB.  Layout [q][lane]: the tile of k-group q is 64 consecutive floats indexed by the READING
// lane, so a read is lane-linear (seven 4-byte reads, paired by the compiler into ds_read2_b32) and the cell of
// (row, unit = 16q + blk), which sits in lane row*16 + blk of wave q, writes float q*64 + blk*4 + row: the 64 writers
// of a wave hit 64 different banks (the [lane][q] order of round 1 put them 8 floats apart: an 8-way conflict on
// every write, SQ_LDS_BANK_CONFLICT 0.73 of the LDS cycles).
constexpr int HG = 8 * 64;  // floats per group and buffer (q = 7 is never used: K = 100 < 112)

// Gate math.  Three forms, chosen at compile time (CHIRON_GATE_MATH; tools/variants.sh --product builds the others for A/B runs
// and for the per-stage error budget of tools/parity_budget.py):
//   0  hardware exp2 / rcp, tanh(x) = 1 - 2 / (2^(2 x log2 e) + 1): the cheapest form (rounds 1 .. 3).  Its tanh carries an
//      ABSOLUTE error of an ulp of 1 .. 2 (1.2e-7 .. 2.4e-7) whatever |x| is, where libm's is relative to tanh(x);
//   1  the same exp2 / rcp with one Newton step on every reciprocal, and tanh through the odd, cancellation-free form
//      tanh|x| = (1 - e) / (1 + e), e = 2^(-2 |x| log2 e) (1 - e is exact for e >= 1/2 and never loses more than an ulp of e);
//   2  libm: expf / tanhf and IEEE division -- what a float32 numpy restatement computes; the yardstick.
//   3  form 0 with the odd tanh of form 1 and no Newton steps: two VALU instructions more per tanh than form 0.
// The four pre-activations are scaled two at a time (v_pk_mul_f32) and the "+ 1" of the denominators added two at a time
// (v_pk_add_f32): every VALU instruction of the step is paid in matrix-pipe time (tools/ubench/mfma_valu_overlap.hip).
//   sigmoid(x) = 1 / (1 + 2^(-x log2 e))   (saturates correctly at +-inf in every form)
#ifndef CHIRON_GATE_MATH
#define CHIRON_GATE_MATH 0
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float LOG2E = 1.4426950408889634f;
#if CHIRON_GATE_MATH == 3
__device__ __forceinline__ float sym_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(__builtin_fabsf(x) * (-2.0f * LOG2E));      // (0, 1]
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return __builtin_copysignf(t, x);
}
#endif
#if CHIRON_GATE_MATH == 1
__device__ __forceinline__ float rcp_newton(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  // r' = r + r (1 - d r); d = inf gives r = 0 and 1 - inf * 0 = NaN, so the correction is dropped there (v_cndmask on a class test
  // would cost more than guarding with the finite test the compiler turns into v_cmp_class)
  const float e = fmaf(-d, r, 1.0f);
  const float r2 = fmaf(r, e, r);
  return __builtin_isfinite(d) ? r2 : r;
}
__device__ __forceinline__ float sym_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(__builtin_fabsf(x) * (-2.0f * LOG2E));      // (0, 1]
  const float t = (1.0f - e) * rcp_newton(1.0f + e);
  return __builtin_copysignf(t, x);
}
#endif
__device__ __forceinline__ float fast_tanh(float x) {
#if CHIRON_GATE_MATH == 2
  return tanhf(x);
#elif CHIRON_GATE_MATH == 1 || CHIRON_GATE_MATH == 3
  return sym_tanh(x);
#else
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * (2.0f * LOG2E)) + 1.0f), 1.0f);
#endif
}
// q = (i, j, f, o) pre-activations, c = previous cell state  ->  new cell state; *h_out = new output
__device__ __forceinline__ float lstm_cell(f32x4 q, float c, float* h_out) {
#if CHIRON_GATE_MATH == 2
  const float si = 1.0f / (1.0f + expf(-q[0])), sf = 1.0f / (1.0f + expf(-q[2])), so = 1.0f / (1.0f + expf(-q[3]));
  const float tj = tanhf(q[1]);
  const float cn = fmaf(sf, c, si * tj);
  *h_out = so * tanhf(cn);
  return cn;
#elif CHIRON_GATE_MATH == 1
  const f32x2 e_if = (f32x2){q[0], q[2]} * (f32x2){-LOG2E, -LOG2E};
  const f32x2 d_if = (f32x2){__builtin_amdgcn_exp2f(e_if[0]), __builtin_amdgcn_exp2f(e_if[1])} + (f32x2){1.0f, 1.0f};
  const float d_o = __builtin_amdgcn_exp2f(q[3] * -LOG2E) + 1.0f;
  const float si = rcp_newton(d_if[0]), sf = rcp_newton(d_if[1]), so = rcp_newton(d_o);
  const float tj = sym_tanh(q[1]);
  const float cn = fmaf(sf, c, si * tj);
  *h_out = so * sym_tanh(cn);
  return cn;
#elif CHIRON_GATE_MATH == 3
  const f32x2 e_if = (f32x2){q[0], q[2]} * (f32x2){-LOG2E, -LOG2E};
  const f32x2 d_if = (f32x2){__builtin_amdgcn_exp2f(e_if[0]), __builtin_amdgcn_exp2f(e_if[1])} + (f32x2){1.0f, 1.0f};
  const float d_o = __builtin_amdgcn_exp2f(q[3] * -LOG2E) + 1.0f;
  const float si = __builtin_amdgcn_rcpf(d_if[0]), sf = __builtin_amdgcn_rcpf(d_if[1]), so = __builtin_amdgcn_rcpf(d_o);
  const float tj = sym_tanh(q[1]);
  const float cn = fmaf(sf, c, si * tj);
  *h_out = so * sym_tanh(cn);
  return cn;
#else
  // the pairs are the register-adjacent ones, (i, j) and (f, o): the accumulators of the 16-row forms deliver the four gates in
  // consecutive registers, and pairing (i, f) / (o, j) cost five moves per cell to build the packed operands
  const f32x2 e_ij = (f32x2){q[0], q[1]} * (f32x2){-LOG2E, 2.0f * LOG2E};
  const f32x2 e_fo = (f32x2){q[2], q[3]} * (f32x2){-LOG2E, -LOG2E};
  const f32x2 d_ij = (f32x2){__builtin_amdgcn_exp2f(e_ij[0]), __builtin_amdgcn_exp2f(e_ij[1])} + (f32x2){1.0f, 1.0f};
  const f32x2 d_fo = (f32x2){__builtin_amdgcn_exp2f(e_fo[0]), __builtin_amdgcn_exp2f(e_fo[1])} + (f32x2){1.0f, 1.0f};
  const float si = __builtin_amdgcn_rcpf(d_ij[0]), sf = __builtin_amdgcn_rcpf(d_fo[0]);
  const float so = __builtin_amdgcn_rcpf(d_fo[1]);
  const float tj = fmaf(-2.0f, __builtin_amdgcn_rcpf(d_ij[1]), 1.0f);
  const float cn = fmaf(sf, c, si * tj);
  *h_out = so * fast_tanh(cn);
  return cn;
#endif
}

// 4x4 transpose between the register index and lane bits [5:4], in registers: (lane = gate*16 + unit, reg = row) ->
// (lane = row*16 + unit, reg = gate).  Two gfx950 lane-swap instructions per stage (v_permlane32_swap exchanges the
// upper 32 lanes of one register with the lower 32 of another, v_permlane16_swap the odd 16-lane rows of one with the
// even rows of another; checked in tools/ubench/permlane_transpose.hip) instead of a round trip through LDS.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 gate_transpose(f32x4 v) {
  unsigned a = __float_as_uint(v[0]), b = __float_as_uint(v[1]), c = __float_as_uint(v[2]), d = __float_as_uint(v[3]);
  u32x2 t;
  t = __builtin_amdgcn_permlane32_swap(a, c, false, false), a = t[0], c = t[1];
  t = __builtin_amdgcn_permlane32_swap(b, d, false, false), b = t[0], d = t[1];
  t = __builtin_amdgcn_permlane16_swap(a, b, false, false), a = t[0], b = t[1];
  t = __builtin_amdgcn_permlane16_swap(c, d, false, false), c = t[0], d = t[1];
  return (f32x4){__uint_as_float(a), __uint_as_float(b), __uint_as_float(c), __uint_as_float(d)};
}

// ---------------------------------------------------------------------------------------------------------
// The fp32 recurrence.  One workgroup = NGRP (1 or 2) 4-row groups of one direction, 7 waves per group:
//   SIX heavy waves: wave w owns hidden units [16w, 16w + 16) for all four gates = 64 matrix columns = the N extent of
//     one v_mfma_f32_4x4x1_16B_f32; 100 MFMAs per step, its slice of W_hh (100 registers per lane) resident for the
//     whole sequence;
//   ONE light wave for units 96..99: their 16 columns (4 gates x 4 units) occupy four MFMA blocks, and cbsz = 2
//     broadcasts a different A block to each group of four blocks, so the four block groups take four different k
//     (tools/ubench/mfma4x4_bcast2.hip): 28 MFMAs cover K = 100 instead of 100 MFMAs with 48 of 64 columns multiplying
//     zeros.  The four partial sums of a column are joined through 1 KB of LDS inside the wave (no extra barrier) in a
//     fixed order, then the same register transpose and cell update as the heavy waves.
// NGRP = 2 ("paired", 14 waves = a whole CU, opt-in): fp32 MFMA and VALU share a SIMD's pipe, so what sets the step time is
// the fullest SIMD.  Two 7-wave workgroups put 4, 4, 3, 3 waves on the four SIMDs; one 14-wave workgroup's waves go to the
// SIMDs in cyclic order, so with the two light waves LAST the loads are 3 heavy + light, 3 heavy + light, 3 heavy, 3 heavy
// on every CU.  Measured it buys nothing (0.767 against 0.763 ms per resident round at T = 400): the single barrier now
// spans both groups and the latency two independent workgroups hide in each other comes back.  NGRP = 1 balances most CUs
// by choosing its light wave from the SIMD slot ids (below).  launch_lstm uses the paired form, when asked, for as many
// groups as fit one round (256 CUs x 8 rows = 1024 rows) and NGRP = 1 for the rest.  Both forms do the same arithmetic for
// a row, so a row's result does not depend on where in the batch it sits.
// ---------------------------------------------------------------------------------------------------------
constexpr int REC_HEAVY = 6;                   // heavy waves per group: wave w owns units [16w, 16w + 16)
constexpr int LIGHT_MF = 28;                   // MFMAs of a light wave per step: q = 0..6 times abid = 0..3
static_assert(REC_HEAVY + 1 == LSTM_NW, "7 waves per group");

template <int NGRP>
__global__ __launch_bounds__(64 * LSTM_NW * NGRP, 4) void lstm_kernel(const LstmParams p) {
  __shared__ __attribute__((aligned(16))) float hbuf[2 * NGRP * HG];    // [buffer][group][HG]
  __shared__ __attribute__((aligned(16))) float part[NGRP * 256];       // light waves: [group][lane][4 rows] K-split partial sums

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x % p.ndir;
  const int gbase = p.group0 + (blockIdx.x / p.ndir) * NGRP;            // first 4-row group of this workgroup
  // Which wave takes the light role.  fp32 MFMA and VALU share a SIMD's pipe, so the step time is set by the fullest
  // SIMD.  The dispatcher deals a workgroup's waves to the SIMDs cyclically and starts the second 7-wave workgroup of a
  // CU one SIMD further (tools/ubench/wave_placement.hip: 0213021|2130213), which leaves 4, 4, 3, 3 waves on the four
  // SIMDs; with the light role fixed to wave 6 one of the two light waves sits on a 3-wave SIMD and the other 4-wave SIMD
  // carries four heavy waves.  A wave's slot index inside its SIMD (HW_ID.wave_id) says how full that SIMD was when the
  // wave arrived: the first wave that got slot 3 sits on a full SIMD and takes the light role; without one (first
  // workgroup on the CU) wave 6 does, which is on the other SIMD that fills up.  Loads become 3 heavy (+ light) on all
  // four SIMDs.  Roles only say which wave computes which columns: results do not depend on them.
  int lw = NGRP * REC_HEAVY;
  if constexpr (NGRP == 1) {
    __shared__ int simd_slot[LSTM_NW];
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (lane == 0) simd_slot[wave] = (int)(hwid & 15u);
    __syncthreads();
#pragma unroll
    for (int i = LSTM_NW - 1; i >= 0; --i)
      if (simd_slot[i] == 3 && !p.fixed_roles) lw = i;
    lw = __builtin_amdgcn_readfirstlane(lw);
  }
  const bool light = NGRP == 1 ? wave == lw : wave >= NGRP * REC_HEAVY;
  const int hwave = (NGRP == 1 && wave > lw) ? wave - 1 : wave;          // heavy waves numbered 0 .. 6 NGRP - 1 in wave order
  const int grp = light ? (NGRP == 1 ? 0 : wave - NGRP * REC_HEAVY) : hwave / REC_HEAVY;   // group inside the workgroup
  const int hq = light ? 6 : hwave % REC_HEAVY;                          // k-group (tile q of the h buffer) this wave's cells write
  const int g0 = gbase + grp;                                            // this wave's 4-row group

  // ---- recurrent weights of this wave: 100 registers (heavy: its 64 columns for every k) or 28 (light: K-split)
  float w[LSTM_K];
  if (!light) {
    const float* wf = p.wfrag + ((long)dir * LSTM_NW + hq) * LSTM_K * 64 + lane;
#pragma unroll
    for (int k = 0; k < LSTM_K; ++k) w[k] = wf[k * 64];
  } else {
    const float* wl = p.wlight + (long)dir * LIGHT_MF * 64 + lane;
#pragma unroll
    for (int m = 0; m < LIGHT_MF; ++m) w[m] = wl[m * 64];
  }
  for (int i = tid; i < 2 * NGRP * HG; i += 64 * LSTM_NW * NGRP) hbuf[i] = 0.f;

  // the cell this lane owns after the transpose: heavy (row = lane >> 4, unit = 16 hq + lane & 15); light: lanes hold
  // the 16 cells (row = (lane & 15) >> 2, unit = 96 + (lane & 3)) four times over, lanes 0..15 store
  const int row = light ? (lane & 15) >> 2 : lane >> 4;
  const int unit = light ? 96 + (lane & 3) : hq * 16 + (lane & 15);
  const int lenr = min(p.seq_len[g0 * 4 + row], p.T);
  int maxlen = 0;   // longest row of the WORKGROUP: every wave runs the same number of steps (one barrier each)
#pragma unroll
  for (int r = 0; r < 4 * NGRP; ++r) maxlen = max(maxlen, min(p.seq_len[gbase * 4 + r], p.T));
  __syncthreads();

  const unsigned outw = p.ndir * p.H;
  const unsigned zcols = 4 * p.H;
  const unsigned zstep = (p.BP >> 2) * p.ndir * zcols * 4;   // floats between consecutive steps
  // heavy: lane = gate*16 + unit reads its column's 4 rows (16 bytes); light: lane = gate*16 + row*4 + j reads one float
  const unsigned zlane_b = light ? (((g0 * p.ndir + dir) * zcols + (lane >> 4) * p.H + 96 + (lane & 3)) * 4 + ((lane >> 2) & 3)) * 4
                                 : ((g0 * p.ndir + dir) * zcols + (lane >> 4) * p.H + hq * 16 + (lane & 15)) * 16;
  const unsigned ostep = p.BP * outw;
  const unsigned olane = (g0 * 4 + row) * outw + dir * p.H + unit;
  const int hw = light ? 6 * 64 + (lane & 3) * 4 + row : hq * 64 + (lane & 15) * 4 + row;   // [q][blk][row]

  float c = 0.f, hprev = 0.f;
  int cur = 0;
  // Two loops, one per role, with the same trip count and one s_barrier per step each: the hardware barrier counts
  // arrivals, so heavy and light waves meet at it from different program points.  (One loop with a role branch inside
  // makes the register allocator keep both roles' values live and spills.)
  if (!light) {
    for (int s = 0; s < maxlen; ++s) {
      const float* zs = p.z + (size_t)s * zstep;   // wave-uniform
      f32x4 z4;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(z4) : "v"(zlane_b), "s"(zs) : "memory");
      const float* hb = hbuf + (cur * NGRP + grp) * HG + lane;
      float hv[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) hv[q] = hb[q * 64];
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#define CHIRON_MF(B)                                                                                       \
  if (16 * q + (B) < LSTM_K) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(hv[q], w[(16 * q + (B)) % LSTM_K], acc, 4, B, 0);
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        CHIRON_MF(0) CHIRON_MF(1) CHIRON_MF(2) CHIRON_MF(3) CHIRON_MF(4) CHIRON_MF(5) CHIRON_MF(6) CHIRON_MF(7)
        CHIRON_MF(8) CHIRON_MF(9) CHIRON_MF(10) CHIRON_MF(11) CHIRON_MF(12) CHIRON_MF(13) CHIRON_MF(14) CHIRON_MF(15)
      }
#undef CHIRON_MF
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(z4), "+v"(acc));   // tied to the accumulators: not ahead of the MFMAs
      const f32x4 gates = gate_transpose(acc + z4);                // i, j, f, o of (row, unit)
      const bool act = s < lenr;
      float hnew;
      const float cn = lstm_cell(gates, c, &hnew);
      c = act ? cn : c;
      hprev = act ? hnew : hprev;
      hbuf[((cur ^ 1) * NGRP + grp) * HG + hw] = hprev;
      const unsigned to = (dir == 0 || !act) ? s : lenr - 1 - s;
      const unsigned ob = (to * ostep + olane) * 4u;
      *reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + ob) = act ? hnew : 0.f;
      cur ^= 1;
      __syncthreads();
    }
  } else {
    float* const mypart = part + grp * 256;
    for (int s = 0; s < maxlen; ++s) {
      const float* zs = p.z + (size_t)s * zstep;
      float z1;
      asm volatile("global_load_dword %0, %1, %2" : "=v"(z1) : "v"(zlane_b), "s"(zs) : "memory");
      const float* hb = hbuf + (cur * NGRP + grp) * HG + lane;
      float hv[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) hv[q] = hb[q * 64];
      // block group kg = lane >> 4 takes k = 16 q + 4 kg + abid: D[lane = kg*16 + gate*4 + j][reg = row] = partial sum
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(hv[q], w[q * 4 + 0], acc, 2, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(hv[q], w[q * 4 + 1], acc, 2, 1, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(hv[q], w[q * 4 + 2], acc, 2, 2, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(hv[q], w[q * 4 + 3], acc, 2, 3, 0);
      }
      // join the four k-subsets inside the wave: lane (gate, row, j) adds partial[kg][gate][j][row], kg = 0..3 in order
      *reinterpret_cast<f32x4*>(mypart + lane * 4) = acc;
      __builtin_amdgcn_wave_barrier();
      const float* src = mypart + (((lane >> 4) * 4 + (lane & 3)) * 4 + ((lane >> 2) & 3));
      float sum = src[0];
      sum += src[64];
      sum += src[128];
      sum += src[192];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(z1), "+v"(sum));
      const float v = sum + z1;
      const f32x4 gates = gate_transpose((f32x4){v, v, v, v});   // every lane (., r*4 + j): i, j, f, o of cell (r, 96 + j)
      const bool act = s < lenr;
      float hnew;
      const float cn = lstm_cell(gates, c, &hnew);
      c = act ? cn : c;
      hprev = act ? hnew : hprev;
      if (lane < 16) {
        hbuf[((cur ^ 1) * NGRP + grp) * HG + hw] = hprev;
        const unsigned to = (dir == 0 || !act) ? s : lenr - 1 - s;
        const unsigned ob = (to * ostep + olane) * 4u;
        *reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + ob) = act ? hnew : 0.f;
      }
      cur ^= 1;
      __syncthreads();
    }
  }

  // ---- frames past the longest row of the workgroup read back as zeros (dynamic_rnn semantics)
  for (int s = maxlen; s < p.T; ++s)
    for (int i = tid; i < 4 * NGRP * p.H; i += 64 * LSTM_NW * NGRP) {
      const int r = i / p.H;
      const int u = i - r * p.H;
      p.out[((long)s * p.BP + gbase * 4 + r) * outw + dir * p.H + u] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// f16 variant (engine dtype CHIRON_F16): h and W_hh are IEEE halves on v_mfma_f32_4x4x4_16B_f16 (K = 4 per
// instruction: 25 MFMAs per step instead of 100, W_hh in 50 registers), accumulation, z, gates and the cell state
// stay fp32; lasth is written as halves for the next layer's f16 GEMM.  Same workgroup shape and A-broadcast
// scheme: lane (blk, row) keeps h[row][4*(blk + 16q) .. +3], q = 0,1 -- one ds_read_b128 per step.
// ---------------------------------------------------------------------------------------------------------
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int HG16 = 16 * 4 * 8;  // halves per group and buffer

__global__ __launch_bounds__(64 * LSTM_NW, 4) void lstm16_kernel(const LstmParams p) {
  __shared__ __attribute__((aligned(16))) _Float16 hbuf[2 * HG16];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x % p.ndir;
  const int g0 = p.group0 + blockIdx.x / p.ndir;  // 4-row group of this workgroup

  f16x4 w[LSTM_KSTEPS16];
  {
    const f16x4* wf = reinterpret_cast<const f16x4*>(p.wfrag) + ((long)dir * LSTM_NW + wave) * LSTM_KSTEPS16 * 64 + lane;
#pragma unroll
    for (int j = 0; j < LSTM_KSTEPS16; ++j) w[j] = wf[j * 64];
  }
  for (int i = tid; i < 2 * HG16; i += 64 * LSTM_NW) hbuf[i] = (_Float16)0.f;

  const int unit = wave * 16 + (lane & 15);
  const int row = lane >> 4;
  const bool live = unit < p.H;
  const int lenr = min(p.seq_len[g0 * 4 + row], p.T);
  int maxlen = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) maxlen = max(maxlen, min(p.seq_len[g0 * 4 + r], p.T));
  __syncthreads();

  const unsigned outw = p.ndir * p.H;
  const unsigned zcols = 4 * p.H;
  const unsigned zstep = (p.BP >> 2) * p.ndir * zcols * 4;
  const unsigned zlane_b = ((g0 * p.ndir + dir) * zcols + (lane >> 4) * p.H + min(wave * 16 + (lane & 15), p.H - 1)) * 8;   // 4 halves per lane
  const unsigned ostep = p.BP * outw;
  const unsigned olane = (g0 * 4 + row) * outw + dir * p.H + unit;
  _Float16* outh = reinterpret_cast<_Float16*>(p.out);
  // this lane's cell writes h[row][unit]: k-step j = unit/4 = blk + 16q
  const int hw = ((((unit >> 2) & 15) * 4 + row) * 2 + (unit >> 6)) * 4 + (unit & 3);

  float c = 0.f, hprev = 0.f;
  int cur = 0;
  for (int s = 0; s < maxlen; ++s) {
    // z arrives as halves (gemm.hip ZGroup with z_f16): 8 bytes per lane and step
    f16x4 zh;
    {
      const _Float16* zs = reinterpret_cast<const _Float16*>(p.z) + (size_t)s * zstep;
      asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(zh) : "v"(zlane_b), "s"(zs) : "memory");
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const f16x8 hv = *reinterpret_cast<const f16x8*>(hbuf + cur * HG16 + lane * 8);
    const f16x4 h0 = {hv[0], hv[1], hv[2], hv[3]}, h1 = {hv[4], hv[5], hv[6], hv[7]};
#define CHIRON_MF16(J)                                                                                     \
  if ((J) < LSTM_KSTEPS16) {                                                                                \
    if ((J) & 1)                                                                                            \
      acc1 = __builtin_amdgcn_mfma_f32_4x4x4f16((J) < 16 ? h0 : h1, w[(J) % LSTM_KSTEPS16], acc1, 4, (J) % 16, 0); \
    else                                                                                                    \
      acc0 = __builtin_amdgcn_mfma_f32_4x4x4f16((J) < 16 ? h0 : h1, w[(J) % LSTM_KSTEPS16], acc0, 4, (J) % 16, 0); \
  }
    CHIRON_MF16(0) CHIRON_MF16(1) CHIRON_MF16(2) CHIRON_MF16(3) CHIRON_MF16(4) CHIRON_MF16(5) CHIRON_MF16(6) CHIRON_MF16(7)
    CHIRON_MF16(8) CHIRON_MF16(9) CHIRON_MF16(10) CHIRON_MF16(11) CHIRON_MF16(12) CHIRON_MF16(13) CHIRON_MF16(14) CHIRON_MF16(15)
    CHIRON_MF16(16) CHIRON_MF16(17) CHIRON_MF16(18) CHIRON_MF16(19) CHIRON_MF16(20) CHIRON_MF16(21) CHIRON_MF16(22) CHIRON_MF16(23)
    CHIRON_MF16(24)
#undef CHIRON_MF16
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(zh), "+v"(acc0), "+v"(acc1));  // tied to the accumulators: not ahead of the MFMAs
    const f32x4 q = gate_transpose(acc0 + acc1 + (f32x4){(float)zh[0], (float)zh[1], (float)zh[2], (float)zh[3]});  // i, j, f, o of (row, unit)
    const bool act = s < lenr;
    float hnew;
    const float cn = lstm_cell(q, c, &hnew);
    c = act ? cn : c;
    hprev = act ? hnew : hprev;
    if (live) {
      hbuf[(cur ^ 1) * HG16 + hw] = (_Float16)hprev;
      const unsigned to = (dir == 0 || !act) ? s : lenr - 1 - s;
      if (p.out_f32) p.out[to * ostep + olane] = act ? hnew : 0.f;
        else outh[to * ostep + olane] = (_Float16)(act ? hnew : 0.f);
    }
    cur ^= 1;
    __syncthreads();
  }
  for (int s = maxlen; s < p.T; ++s)
    for (int i = tid; i < 4 * p.H; i += 64 * LSTM_NW) {
      const int r = i / p.H;
      const int u = i - r * p.H;
      if (p.out_f32) p.out[((long)s * p.BP + g0 * 4 + r) * outw + dir * p.H + u] = 0.f; else outh[((long)s * p.BP + g0 * 4 + r) * outw + dir * p.H + u] = (_Float16)0.f;
    }
}

// res_layer1/conv2a (the 1x1 convolution of the one-channel signal + BN + ReLU) materialised: a1[pos][c] = relu(sig[pos]*a[c]
// + b[c]).  Folding it into the loader of conv2b (gemm_f32_kernel<LIFT>) saves the 450 MB round trip but forces the
// register-staged GEMM, which is slower than this pass plus the DMA GEMM.
template <int FMT>  // 0 fp32, 1 halves, 2 split hi/lo
__global__ __launch_bounds__(256) void lift_kernel(const float* __restrict__ sig, const float* __restrict__ a, const float* __restrict__ b,
                                                   void* __restrict__ outv, long n_pos, int C) {
  const int c8 = C / 8;  // 8 channels per thread
  const long total = n_pos * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pos = i / c8;
    const int c0 = (int)(i - pos * c8) * 8;
    const float x = sig[pos];
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = fmaxf(fmaf(x, a[c0 + j], b[c0 + j]), 0.f);
    if (FMT == 0) {
      float* o = reinterpret_cast<float*>(outv) + pos * C + c0;
      *reinterpret_cast<f32x4*>(o) = (f32x4){y[0], y[1], y[2], y[3]};
      *reinterpret_cast<f32x4*>(o + 4) = (f32x4){y[4], y[5], y[6], y[7]};
    } else {
      f16x8 v, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = (_Float16)y[j];
        lo[j] = (_Float16)(y[j] - (float)v[j]);
      }
      _Float16* out = reinterpret_cast<_Float16*>(outv);
      if (FMT == 2) {
        _Float16* o = out + (pos * C + (c0 >> 5) * 32) * 2 + (c0 & 31);
        *reinterpret_cast<f16x8*>(o) = v;
        *reinterpret_cast<f16x8*>(o + 32) = lo;
      } else {
        *reinterpret_cast<f16x8*>(out + pos * C + c0) = v;
      }
    }
  }
}

void launch_lift(const float* sig, const float* a, const float* b, void* out, long n_pos, int C, int fmt, hipStream_t stream) {
  const dim3 grid(256 * 16), block(256);
  if (fmt == 0)
    hipLaunchKernelGGL(lift_kernel<0>, grid, block, 0, stream, sig, a, b, out, n_pos, C);
  else if (fmt == 1)
    hipLaunchKernelGGL(lift_kernel<1>, grid, block, 0, stream, sig, a, b, out, n_pos, C);
  else
    hipLaunchKernelGGL(lift_kernel<2>, grid, block, 0, stream, sig, a, b, out, n_pos, C);
}

// lasth of the fp32 recurrence -> split hi/lo format for the next layer's projection GEMM (dtype fp32-split).  A separate
// HBM-bound pass: the recurrence has no registers or issue slots to spare for the conversion.
__global__ __launch_bounds__(256) void split_convert_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long rows, int cols, int ld,
                                                            int half, int half_dst) {
  const int c4n = cols / 4;
  const long total = rows * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c4n;
    const int c0 = (int)(i - r * c4n) * 4;   // 4 consecutive columns never straddle a 32-element block
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * cols + c0);
    f16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hi[j] = (_Float16)v[j];
      lo[j] = (_Float16)(v[j] - (float)hi[j]);
    }
    const int cd = (half > 0 && c0 >= half) ? c0 - half + half_dst : c0;   // half is a multiple of 4: a group stays on one side
    _Float16* o = dst + (r * ld + (cd >> 5) * 32) * 2 + (cd & 31);
    *reinterpret_cast<f16x4*>(o) = hi;
    *reinterpret_cast<f16x4*>(o + 32) = lo;
  }
}

// Stem convolution on the one-channel signal (HEAD RNA_model2 / RNA_model3): K = k taps only, so it is an elementwise
// kernel, 8 output channels per thread; BN (population) is folded into w / shift by the engine.
template <int FMT>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ sig, const float* __restrict__ w, const float* __restrict__ shift,
                                                        void* __restrict__ outv, int B, int L, int T_out, int k, int stride, int left, int C,
                                                        int relu) {
  const int c8 = C / 8;
  const long total = (long)B * T_out * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pos = i / c8;
    const int c0 = (int)(i - pos * c8) * 8;
    const long b = pos / T_out;
    const int t = (int)(pos - b * T_out);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int tap = 0; tap < k; ++tap) {
      const int it = t * stride + tap - left;
      if (it < 0 || it >= L) continue;
      const float x = sig[b * L + it];
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + tap * C + c0), w1 = *reinterpret_cast<const f32x4*>(w + tap * C + c0 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = fmaf(x, w0[j], acc[j]);
        acc[4 + j] = fmaf(x, w1[j], acc[4 + j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[j] += shift[c0 + j];
      if (relu) acc[j] = fmaxf(acc[j], 0.f);
    }
    if (FMT == 0) {
      float* o = reinterpret_cast<float*>(outv) + pos * C + c0;
      *reinterpret_cast<f32x4*>(o) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<f32x4*>(o + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
    } else {
      f16x8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        hi[j] = (_Float16)acc[j];
        lo[j] = (_Float16)(acc[j] - (float)hi[j]);
      }
      _Float16* oh = reinterpret_cast<_Float16*>(outv);
      if (FMT == 1) {
        *reinterpret_cast<f16x8*>(oh + pos * C + c0) = hi;
      } else {
        _Float16* o = oh + (pos * C + (c0 >> 5) * 32) * 2 + (c0 & 31);
        *reinterpret_cast<f16x8*>(o) = hi;
        *reinterpret_cast<f16x8*>(o + 32) = lo;
      }
    }
  }
}

void launch_stem_conv(const float* sig, const float* w, const float* shift, void* out, int B, int L, int T_out, int k, int stride, int left,
                      int C, int fmt, int relu, hipStream_t stream) {
  const dim3 grid(256 * 16), block(256);
  if (fmt == 0)
    hipLaunchKernelGGL(stem_conv_kernel<0>, grid, block, 0, stream, sig, w, shift, out, B, L, T_out, k, stride, left, C, relu);
  else if (fmt == 1)
    hipLaunchKernelGGL(stem_conv_kernel<1>, grid, block, 0, stream, sig, w, shift, out, B, L, T_out, k, stride, left, C, relu);
  else
    hipLaunchKernelGGL(stem_conv_kernel<2>, grid, block, 0, stream, sig, w, shift, out, B, L, T_out, k, stride, left, C, relu);
}

void launch_split_convert(const float* src, void* dst, long rows, int cols, int ld, int half, int half_dst, hipStream_t stream) {
  hipLaunchKernelGGL(split_convert_kernel, dim3(256 * 16), dim3(256), 0, stream, src, reinterpret_cast<_Float16*>(dst), rows, cols, ld, half,
                     half_dst);
}


// ---------------------------------------------------------------------------------------------------------
// f16, wide form: one workgroup = SIXTEEN batch rows x one direction on v_mfma_f32_16x16x16_f16 (M = 16 rows, N = 16
// columns = 4 units x 4 gates, K = 16; 8 cycles, the full f16 rate -- the 4x4x4 form above pays 10 cycles for a
// sixteenth of the work).  B = 4096 (BASELINE configs[4]) gives 512 workgroups = 2 per CU; what remains is the gate math
// (one cell per lane and column tile).  8 waves; wave w owns the column tiles 3w .. 3w+2 (wave 7: 21 .. 24) of the 25
// (H = 100 = 25 x 4 units), their W_hh slices (7 k-steps x 2 registers per tile) resident in VGPRs.
//   A = h_{t-1} [16 rows][K = 112 padded]: lane (row = lane & 15, kq = lane >> 4) reads h[row][16 s + 4 kq .. +3] for
//     k-step s -- the LDS tile is laid out [unit tile = 4 s + kq][row][4 units], i.e. lane-linear 8-byte reads, and the
//     64 cells of a column tile write 128 contiguous bytes;
//   D: lane (col = lane & 15 = 4 u + gate, q = lane >> 4) holds rows 4q .. 4q+3 of its column: z (stored [col][4 rows] by
//     the projection, one 8-byte load per lane and tile) is added in place, then a 4 x 4 transpose inside each lane quad
//     gives lane (u, r) the four gates of cell (row 4q + r, unit 4 tile + u).  The kernel is VALU-bound (gate math of four
//     cells per lane and step), so the transpose goes through 1.1 KB of wave-private LDS -- one ds_write_b128, two
//     ds_read2_b32, bank-conflict free with 16 bytes of padding per 16 lanes -- instead of 4 DPP moves + 8 selects per tile.
//     (The transposed product D = W^T h^T delivers the four gates of a cell to one lane directly, but then z needs four
//     2-byte loads per lane and tile: measured 1.40 ms against 1.12 ms per launch at B = 4096.)
// Same cell arithmetic as lstm16_kernel; the k order of the fp32 accumulation differs (16 per MFMA instead of 4), so
// the two forms agree to rounding, not bit for bit; which one runs depends only on the padded batch size.
// ---------------------------------------------------------------------------------------------------------
constexpr int W16_NW = 8;       // waves per workgroup
constexpr int W16_NT = 4;       // column-tile slots per wave (3 used by waves 0..6)
constexpr int W16_KS = 7;       // k-steps of 16: K = 100 padded to 112
constexpr int HW16 = 28 * 64;   // halves per h buffer: 28 unit tiles x 16 rows x 4 units (tiles 25..27 stay zero)

constexpr int W16_XF = 288;     // floats of transpose scratch per wave and tile slot (64 x 4 + 4 per 16 lanes of padding)

__global__ __launch_bounds__(64 * W16_NW, 2) void lstm16w_kernel(const LstmParams p) {
  __shared__ __attribute__((aligned(16))) _Float16 hbuf[2 * HW16];
  __shared__ __attribute__((aligned(16))) float xf[W16_NW * W16_NT * W16_XF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x % p.ndir;
  const int g16 = blockIdx.x / p.ndir;                 // 16-row group = the 4-row groups 4 g16 .. 4 g16 + 3
  const int nt = wave == W16_NW - 1 ? 4 : 3;           // column tiles of this wave
  const int tile0 = 3 * wave;

  f16x4 w[W16_NT][W16_KS];
  {
    const f16x4* wf = reinterpret_cast<const f16x4*>(p.wwide) + ((long)dir * W16_NW + wave) * W16_NT * W16_KS * 64 + lane;
#pragma unroll
    for (int n = 0; n < W16_NT; ++n)
#pragma unroll
      for (int ks = 0; ks < W16_KS; ++ks) w[n][ks] = wf[(n * W16_KS + ks) * 64];
  }
  for (int i = tid; i < 2 * HW16; i += 64 * W16_NW) hbuf[i] = (_Float16)0.f;

  const int q = lane >> 4, u = (lane >> 2) & 3, gp = lane & 3;   // before the transpose: column 4u + gp, rows 4q .. 4q+3
  const int row = 4 * q + gp;                                      // after it: this lane's cell is (row, unit 4 tile + u)
  const int brow = g16 * 16 + row;
  const int lenr = min(p.seq_len[brow], p.T);
  int maxlen = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) maxlen = max(maxlen, min(p.seq_len[g16 * 16 + r], p.T));
  __syncthreads();

  const unsigned outw = p.ndir * p.H;
  const unsigned zcols = 4 * p.H;
  const unsigned zstep = (p.BP >> 2) * p.ndir * zcols * 4;        // halves between consecutive steps
  // byte offset of this lane's 4 halves (rows 4q..4q+3 of column gate*H + unit) for tile slot 0; + 32 bytes per tile
  const unsigned zlane_b = ((((g16 * 4 + q) * p.ndir + dir) * zcols + gp * p.H + 4 * tile0 + u) * 4) * 2;
  const unsigned ostep = p.BP * outw;
  const unsigned olane = brow * outw + dir * p.H + 4 * tile0 + u;  // + 4 per tile
  const int hw = tile0 * 64 + row * 4 + u;                          // + 64 per tile
  _Float16* outh = reinterpret_cast<_Float16*>(p.out);

  // transpose scratch: lane l = (q, u, gp) stores its 4 rows at float 4 l + 4 q; lane (q, u, r) then reads gate g at
  // float 4 (16 q + 4 u + g) + 4 q + r: bank 16 u + 4 g + 4 q + r, distinct over the wave for every g
  float* const xw = xf + wave * W16_NT * W16_XF + 4 * lane + 4 * q;
  const float* const xr = xf + wave * W16_NT * W16_XF + 16 * (4 * q + u) + 4 * q + gp;

  float c[W16_NT] = {0.f, 0.f, 0.f, 0.f}, hprev[W16_NT] = {0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  for (int s = 0; s < maxlen; ++s) {
    const _Float16* zs = reinterpret_cast<const _Float16*>(p.z) + (size_t)s * zstep;   // wave-uniform
    f16x4 zh[W16_NT];
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(zh[0]) : "v"(zlane_b), "s"(zs) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:32" : "=v"(zh[1]) : "v"(zlane_b), "s"(zs) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:64" : "=v"(zh[2]) : "v"(zlane_b), "s"(zs) : "memory");
    if (nt == 4) asm volatile("global_load_dwordx2 %0, %1, %2 offset:96" : "=v"(zh[3]) : "v"(zlane_b), "s"(zs) : "memory");
    else zh[3] = (f16x4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    const f16x4* hb = reinterpret_cast<const f16x4*>(hbuf + cur * HW16) + lane;
    f16x4 hv[W16_KS];
#pragma unroll
    for (int ks = 0; ks < W16_KS; ++ks) hv[ks] = hb[ks * 64];
    f32x4 acc[W16_NT];
#pragma unroll
    for (int n = 0; n < W16_NT; ++n) {
      acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (n < nt) {
#pragma unroll
        for (int ks = 0; ks < W16_KS; ++ks) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(hv[ks], w[n][ks], acc[n], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(zh[0]), "+v"(zh[1]), "+v"(zh[2]), "+v"(zh[3]), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    const bool act = s < lenr;
    const unsigned to = (dir == 0 || !act) ? s : lenr - 1 - s;
#pragma unroll
    for (int n = 0; n < W16_NT; ++n) {
      if (n < nt) {
        *reinterpret_cast<f32x4*>(xw + n * W16_XF) = acc[n] + (f32x4){(float)zh[n][0], (float)zh[n][1], (float)zh[n][2], (float)zh[n][3]};
      }
    }
    __builtin_amdgcn_wave_barrier();   // wave-private scratch: LDS operations of a wave execute in order
#pragma unroll
    for (int n = 0; n < W16_NT; ++n) {
      if (n < nt) {
        const float* xs = xr + n * W16_XF;
        const f32x4 gates = {xs[0], xs[4], xs[8], xs[12]};   // i, j, f, o of (row, unit 4 (tile0 + n) + u)
        float hnew;
        const float cn = lstm_cell(gates, c[n], &hnew);
        c[n] = act ? cn : c[n];
        hprev[n] = act ? hnew : hprev[n];
        hbuf[(cur ^ 1) * HW16 + hw + 64 * n] = (_Float16)hprev[n];
        if (p.out_f32) p.out[to * ostep + olane + 4 * n] = act ? hnew : 0.f;
        else outh[to * ostep + olane + 4 * n] = (_Float16)(act ? hnew : 0.f);
      }
    }
    cur ^= 1;
    __syncthreads();
  }

  // ---- frames past the longest row of the workgroup read back as zeros (dynamic_rnn semantics)
  for (int s = maxlen; s < p.T; ++s)
    for (int i = tid; i < 16 * p.H; i += 64 * W16_NW) {
      const int r = i / p.H;
      const int uu = i - r * p.H;
      if (p.out_f32) p.out[((long)s * p.BP + g16 * 16 + r) * outw + dir * p.H + uu] = 0.f; else outh[((long)s * p.BP + g16 * 16 + r) * outw + dir * p.H + uu] = (_Float16)0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// dtype fp32-split: the recurrence on the f16 matrix pipe with fp32 VALUES.  Same workgroup shape and operand layouts as
// lstm16w_kernel (sixteen batch rows x one direction, eight waves, wave w owns the column tiles 3w .. 3w+2 (wave 7: four) of 4 units x
// 4 gates, v_mfma_f32_16x16x16_f16), but h and W_hh are carried as exact hi + lo half pairs (x = hi + lo to 2^-22, the format the
// split engine's GEMMs already use) and a tile's product is hi*hi + hi*lo + lo*hi with fp32 accumulation: 21 MFMAs of 16 cycles per
// tile and step instead of the 25 x 32 cycles of v_mfma_f32_16x16x4_f32 -- the recurrence was 4.4 of the split engine's 8.1 ms per batch
// while it ran the fp32 kernel (plus a conversion pass per layer).  z (fp32, the projection's layout), gates, cell state and the output
// (fp32 lasth, what the fp32 kernels write) are unchanged, so the engine swaps the kernel and nothing else.  A row's bits do not depend
// on the batch it travels in (every output element of a 16 x 16 tile depends on its own row and column only).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * W16_NW, 1) void lstm32s_kernel(const LstmParams p) {
  __shared__ __attribute__((aligned(16))) _Float16 hhi[2 * HW16];
  __shared__ __attribute__((aligned(16))) _Float16 hlo[2 * HW16];
  __shared__ __attribute__((aligned(16))) float xf[W16_NW * W16_NT * W16_XF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x % p.ndir;
  const int g16 = blockIdx.x / p.ndir;
  const int nt = wave == W16_NW - 1 ? 4 : 3;
  const int tile0 = 3 * wave;

  // W_hh fragments: [hi | lo][dir][wave][slot][k-step][lane] f16x4, the order of LstmParams::wwide
  f16x4 wh[W16_NT][W16_KS], wl[W16_NT][W16_KS];
  {
    const long half = (long)p.ndir * W16_NW * W16_NT * W16_KS * 64;
    const f16x4* wf = reinterpret_cast<const f16x4*>(p.wsplit) + ((long)dir * W16_NW + wave) * W16_NT * W16_KS * 64 + lane;
#pragma unroll
    for (int n = 0; n < W16_NT; ++n)
#pragma unroll
      for (int ks = 0; ks < W16_KS; ++ks) {
        wh[n][ks] = wf[(n * W16_KS + ks) * 64];
        wl[n][ks] = wf[half + (n * W16_KS + ks) * 64];
      }
  }
  for (int i = tid; i < 2 * HW16; i += 64 * W16_NW) hhi[i] = (_Float16)0.f, hlo[i] = (_Float16)0.f;

  const int q = lane >> 4, u = (lane >> 2) & 3, gp = lane & 3;   // before the transpose: column 4u + gp, rows 4q .. 4q+3
  const int row = 4 * q + gp;                                      // after it: this lane's cell is (row, unit 4 tile + u)
  const int brow = g16 * 16 + row;
  const int lenr = min(p.seq_len[brow], p.T);
  int maxlen = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) maxlen = max(maxlen, min(p.seq_len[g16 * 16 + r], p.T));
  __syncthreads();

  const unsigned outw = p.ndir * p.H;
  const unsigned zcols = 4 * p.H;
  const unsigned zstep = (p.BP >> 2) * p.ndir * zcols * 4;        // floats between consecutive steps
  // byte offset of this lane's 4 floats (rows 4q..4q+3 of column gate*H + unit) for tile slot 0; + 64 bytes per tile
  const unsigned zlane_b = ((((g16 * 4 + q) * p.ndir + dir) * zcols + gp * p.H + 4 * tile0 + u) * 4) * 4;
  const unsigned ostep = p.BP * outw;
  const unsigned olane = brow * outw + dir * p.H + 4 * tile0 + u;  // + 4 per tile
  const int hw = tile0 * 64 + row * 4 + u;                          // + 64 per tile
  float* outf = p.out;
  // split-format destination (layers that feed another projection): element column of this lane's unit in tile slot n, and the
  // half index of its hi value inside a row (the lo value sits 32 halves further)
  _Float16* const outs = reinterpret_cast<_Float16*>(p.out_split);
  const unsigned srow = 2u * p.split_ld;                             // halves per row
  // the copy-out thread's (tile = tid / 16, row = tid % 16): its row's length and the half index of the tile's four hi values in a row
  const int olens = (outs && tid < 25 * 16) ? min(p.seq_len[g16 * 16 + (tid & 15)], p.T) : 0;
  const unsigned ocol0 = (dir == 0 ? 0u : (unsigned)p.split_bw0) + 4u * (tid >> 4);
  const unsigned ocol = (ocol0 >> 5) * 64 + (ocol0 & 31);

  float* const xw = xf + wave * W16_NT * W16_XF + 4 * lane + 4 * q;
  const float* const xr = xf + wave * W16_NT * W16_XF + 16 * (4 * q + u) + 4 * q + gp;

  // Split-format output leaves from the completed h tiles: thread (tile, row) copies the 4 units of its tile -- 8 bytes of hi halves
  // and 8 bytes of lo halves, exactly the pair the next step multiplies -- instead of two 2-byte stores per cell (1600 x 2 per step:
  // 0.83 ms per launch against 0.61 with fp32 output).  A row past its length holds its carried state there and is written as zeros.
  // Step sp's h is the INPUT tile of step sp + 1 (stable for that whole step), and the copy is issued there BEHIND the wait for z:
  // vmcnt counts stores too, and a store issued in front of that wait makes every step wait for its own write acknowledgements.
  auto copy_out = [&](int sp, int buf) {
    if (outs && sp >= 0 && tid < 25 * 16) {
      const int ot = tid >> 4, orow = tid & 15;
      const bool oact = sp < olens;
      const unsigned oto = (dir == 0 || !oact) ? sp : olens - 1 - sp;
      f16x4 vh = *reinterpret_cast<const f16x4*>(hhi + buf * HW16 + ot * 64 + orow * 4);
      f16x4 vl = *reinterpret_cast<const f16x4*>(hlo + buf * HW16 + ot * 64 + orow * 4);
      if (!oact) vh = vl = (f16x4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      _Float16* o = outs + (size_t)(oto * p.BP + g16 * 16 + orow) * srow + ocol;
      *reinterpret_cast<f16x4*>(o) = vh;
      *reinterpret_cast<f16x4*>(o + 32) = vl;
    }
  };
  float c[W16_NT] = {0.f, 0.f, 0.f, 0.f}, hprev[W16_NT] = {0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  for (int s = 0; s < maxlen; ++s) {
    const float* zs = p.z + (size_t)s * zstep;   // wave-uniform
    f32x4 zv[W16_NT];
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(zv[0]) : "v"(zlane_b), "s"(zs) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(zv[1]) : "v"(zlane_b), "s"(zs) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:128" : "=v"(zv[2]) : "v"(zlane_b), "s"(zs) : "memory");
    if (nt == 4) asm volatile("global_load_dwordx4 %0, %1, %2 offset:192" : "=v"(zv[3]) : "v"(zlane_b), "s"(zs) : "memory");
    else zv[3] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16x4* hbh = reinterpret_cast<const f16x4*>(hhi + cur * HW16) + lane;
    const f16x4* hbl = reinterpret_cast<const f16x4*>(hlo + cur * HW16) + lane;
    f16x4 hvh[W16_KS], hvl[W16_KS];
#pragma unroll
    for (int ks = 0; ks < W16_KS; ++ks) hvh[ks] = hbh[ks * 64], hvl[ks] = hbl[ks * 64];
    f32x4 acc[W16_NT];
#pragma unroll
    for (int n = 0; n < W16_NT; ++n) {
      acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (n < nt) {
        // the two cross terms first (2^-11 of the main term each), then hi * hi: the small addends are not rounded against a large sum
#pragma unroll
        for (int ks = 0; ks < W16_KS; ++ks) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(hvl[ks], wh[n][ks], acc[n], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < W16_KS; ++ks) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(hvh[ks], wl[n][ks], acc[n], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < W16_KS; ++ks) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(hvh[ks], wh[n][ks], acc[n], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(zv[0]), "+v"(zv[1]), "+v"(zv[2]), "+v"(zv[3]), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    copy_out(s - 1, cur);
    const bool act = s < lenr;
    const unsigned to = (dir == 0 || !act) ? s : lenr - 1 - s;
#pragma unroll
    for (int n = 0; n < W16_NT; ++n)
      if (n < nt) *reinterpret_cast<f32x4*>(xw + n * W16_XF) = acc[n] + zv[n];
    __builtin_amdgcn_wave_barrier();   // wave-private scratch: LDS operations of a wave execute in order
#pragma unroll
    for (int n = 0; n < W16_NT; ++n) {
      if (n < nt) {
        const float* xs = xr + n * W16_XF;
        const f32x4 gates = {xs[0], xs[4], xs[8], xs[12]};   // i, j, f, o of (row, unit 4 (tile0 + n) + u)
        float hnew;
        const float cn = lstm_cell(gates, c[n], &hnew);
        c[n] = act ? cn : c[n];
        hprev[n] = act ? hnew : hprev[n];
        const _Float16 hi = (_Float16)hprev[n];
        hhi[(cur ^ 1) * HW16 + hw + 64 * n] = hi;
        hlo[(cur ^ 1) * HW16 + hw + 64 * n] = (_Float16)(hprev[n] - (float)hi);
        if (!outs) outf[to * ostep + olane + 4 * n] = act ? hnew : 0.f;
      }
    }
    cur ^= 1;
    __syncthreads();
  }
  copy_out(maxlen - 1, cur);   // the last step's h

  // ---- frames past the longest row of the workgroup read back as zeros (dynamic_rnn semantics)
  for (int s = maxlen; s < p.T; ++s)
    for (int i = tid; i < 16 * p.H; i += 64 * W16_NW) {
      const int r = i / p.H;
      const int uu = i - r * p.H;
      if (outs) {
        const unsigned col = (dir == 0 ? 0u : (unsigned)p.split_bw0) + uu;
        _Float16* o = outs + ((size_t)s * p.BP + g16 * 16 + r) * srow + (col >> 5) * 64 + (col & 31);
        o[0] = (_Float16)0.f;
        o[32] = (_Float16)0.f;
      } else {
        outf[((long)s * p.BP + g16 * 16 + r) * outw + dir * p.H + uu] = 0.f;
      }
    }
}

// ---------------------------------------------------------------------------------------------------------
// dtype f16-w2 (CHIRON_F16_W2): lstm16w_kernel's recurrence -- h as halves, the same workgroup shape, operand layouts, transpose and
// cell -- against EXACT recurrent weights: W_hh as hi + lo half pairs (lstm32s_kernel's fragments), h*lo then h*hi per tile and step
// (14 MFMAs of 16 cycles), z read as fp32 (the projection of this dtype does not round it), the output written as halves for the
// next layer's projection / the FC head.  A row's bits do not depend on the batch it travels in.
// ---------------------------------------------------------------------------------------------------------
template <bool ZH>   // ZH: z arrives as halves (A/B switch CHIRON_W2_ZF16: half the z traffic, z rounded to 11 bits)
__global__ __launch_bounds__(64 * W16_NW, 1) void lstm16w2_kernel(const LstmParams p) {
  __shared__ __attribute__((aligned(16))) _Float16 hbuf[2 * HW16];
  __shared__ __attribute__((aligned(16))) float xf[W16_NW * W16_NT * W16_XF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x % p.ndir;
  const int g16 = blockIdx.x / p.ndir;
  const int nt = wave == W16_NW - 1 ? 4 : 3;
  const int tile0 = 3 * wave;

  f16x4 wh[W16_NT][W16_KS], wl[W16_NT][W16_KS];
  {
    const long half = (long)p.ndir * W16_NW * W16_NT * W16_KS * 64;
    const f16x4* wf = reinterpret_cast<const f16x4*>(p.wsplit) + ((long)dir * W16_NW + wave) * W16_NT * W16_KS * 64 + lane;
#pragma unroll
    for (int n = 0; n < W16_NT; ++n)
#pragma unroll
      for (int ks = 0; ks < W16_KS; ++ks) {
        wh[n][ks] = wf[(n * W16_KS + ks) * 64];
        wl[n][ks] = wf[half + (n * W16_KS + ks) * 64];
      }
  }
  for (int i = tid; i < 2 * HW16; i += 64 * W16_NW) hbuf[i] = (_Float16)0.f;

  const int q = lane >> 4, u = (lane >> 2) & 3, gp = lane & 3;   // before the transpose: column 4u + gp, rows 4q .. 4q+3
  const int row = 4 * q + gp;                                      // after it: this lane's cell is (row, unit 4 tile + u)
  const int brow = g16 * 16 + row;
  const int lenr = min(p.seq_len[brow], p.T);
  int maxlen = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) maxlen = max(maxlen, min(p.seq_len[g16 * 16 + r], p.T));
  __syncthreads();

  const unsigned outw = p.ndir * p.H;
  const unsigned zcols = 4 * p.H;
  const unsigned zstep = (p.BP >> 2) * p.ndir * zcols * 4;        // elements between consecutive steps
  const unsigned zlane_b = ((((g16 * 4 + q) * p.ndir + dir) * zcols + gp * p.H + 4 * tile0 + u) * 4) * (ZH ? 2 : 4);   // bytes; + 64 (32) per tile
  const unsigned ostep = p.BP * outw;
  const unsigned olane = brow * outw + dir * p.H + 4 * tile0 + u;  // + 4 per tile
  const int hw = tile0 * 64 + row * 4 + u;                          // + 64 per tile
  _Float16* outh = reinterpret_cast<_Float16*>(p.out);

  float* const xw = xf + wave * W16_NT * W16_XF + 4 * lane + 4 * q;
  const float* const xr = xf + wave * W16_NT * W16_XF + 16 * (4 * q + u) + 4 * q + gp;

  float c[W16_NT] = {0.f, 0.f, 0.f, 0.f}, hprev[W16_NT] = {0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  for (int s = 0; s < maxlen; ++s) {
    f32x4 zv[W16_NT];
    f16x4 zh[W16_NT];
    if (ZH) {
      const _Float16* zs = reinterpret_cast<const _Float16*>(p.z) + (size_t)s * zstep;   // wave-uniform
      asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(zh[0]) : "v"(zlane_b), "s"(zs) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, %2 offset:32" : "=v"(zh[1]) : "v"(zlane_b), "s"(zs) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, %2 offset:64" : "=v"(zh[2]) : "v"(zlane_b), "s"(zs) : "memory");
      if (nt == 4) asm volatile("global_load_dwordx2 %0, %1, %2 offset:96" : "=v"(zh[3]) : "v"(zlane_b), "s"(zs) : "memory");
      else zh[3] = (f16x4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    } else {
      const float* zs = p.z + (size_t)s * zstep;   // wave-uniform
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(zv[0]) : "v"(zlane_b), "s"(zs) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(zv[1]) : "v"(zlane_b), "s"(zs) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:128" : "=v"(zv[2]) : "v"(zlane_b), "s"(zs) : "memory");
      if (nt == 4) asm volatile("global_load_dwordx4 %0, %1, %2 offset:192" : "=v"(zv[3]) : "v"(zlane_b), "s"(zs) : "memory");
      else zv[3] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f16x4* hb = reinterpret_cast<const f16x4*>(hbuf + cur * HW16) + lane;
    f16x4 hv[W16_KS];
#pragma unroll
    for (int ks = 0; ks < W16_KS; ++ks) hv[ks] = hb[ks * 64];
    f32x4 acc[W16_NT];
#pragma unroll
    for (int n = 0; n < W16_NT; ++n) {
      acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (n < nt) {
        // the small term first: it is not rounded against a large sum
#pragma unroll
        for (int ks = 0; ks < W16_KS; ++ks) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(hv[ks], wl[n][ks], acc[n], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < W16_KS; ++ks) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(hv[ks], wh[n][ks], acc[n], 0, 0, 0);
      }
    }
    if (ZH) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(zh[0]), "+v"(zh[1]), "+v"(zh[2]), "+v"(zh[3]), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
      for (int n = 0; n < W16_NT; ++n) zv[n] = (f32x4){(float)zh[n][0], (float)zh[n][1], (float)zh[n][2], (float)zh[n][3]};
    } else {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(zv[0]), "+v"(zv[1]), "+v"(zv[2]), "+v"(zv[3]), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    }
    const bool act = s < lenr;
    const unsigned to = (dir == 0 || !act) ? s : lenr - 1 - s;
#pragma unroll
    for (int n = 0; n < W16_NT; ++n)
      if (n < nt) *reinterpret_cast<f32x4*>(xw + n * W16_XF) = acc[n] + zv[n];
    __builtin_amdgcn_wave_barrier();   // wave-private scratch: LDS operations of a wave execute in order
#pragma unroll
    for (int n = 0; n < W16_NT; ++n) {
      if (n < nt) {
        const float* xs = xr + n * W16_XF;
        const f32x4 gates = {xs[0], xs[4], xs[8], xs[12]};   // i, j, f, o of (row, unit 4 (tile0 + n) + u)
        float hnew;
        const float cn = lstm_cell(gates, c[n], &hnew);
        c[n] = act ? cn : c[n];
        hprev[n] = act ? hnew : hprev[n];
        hbuf[(cur ^ 1) * HW16 + hw + 64 * n] = (_Float16)hprev[n];
        if (p.out_f32) p.out[to * ostep + olane + 4 * n] = act ? hnew : 0.f;
        else outh[to * ostep + olane + 4 * n] = (_Float16)(act ? hnew : 0.f);
      }
    }
    cur ^= 1;
    __syncthreads();
  }

  // ---- frames past the longest row of the workgroup read back as zeros (dynamic_rnn semantics)
  for (int s = maxlen; s < p.T; ++s)
    for (int i = tid; i < 16 * p.H; i += 64 * W16_NW) {
      const int r = i / p.H;
      const int uu = i - r * p.H;
      if (p.out_f32) p.out[((long)s * p.BP + g16 * 16 + r) * outw + dir * p.H + uu] = 0.f; else outh[((long)s * p.BP + g16 * 16 + r) * outw + dir * p.H + uu] = (_Float16)0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// f16, wide AND fused with the x-projection: the same 16-row workgroups as lstm16w_kernel, but z is never materialised.
// At B = 4096 the projection GEMM writes 2.6 GB of z per layer and the recurrence reads it back -- 5.2 GB of HBM traffic and
// a 1.25 ms launch per layer for a product the recurrence's idle matrix pipe can do itself: per step the workgroup gathers
// its 16 input rows x_t (8 KB; the backward direction walks t = seq_len - 1 - s per row, so this is a per-row gather, not a
// tile of a GEMM), and every wave multiplies them by ITS columns of W_x, resident in registers next to W_hh (KSX k-steps of
// 16 x 2 registers per column tile; wave 7 with four tiles holds 184 weight registers: one workgroup per CU, 8 waves).
//   MFMA: v_mfma_f32_16x16x32_f16 (gfx950's double-K form: the 16x16x16 instruction still takes 16 cycles on this chip, so
//   it runs at half the f16 rate; measured 1.92 -> see DESIGN 3.5).  x and h tiles in LDS: [k octet][row][8 halves], the
//   A-operand order (lane (row, kg) reads k = 32 s + 8 kg .. +7: lane-linear 16-byte reads); thread (row = tid / 32,
//   j = tid % 32) fetches the 16 bytes k = 8j .. 8j+7 of its row one step ahead: exactly one octet.
//   K = 200 (layers 1, 2) is padded to 224 (7 k-steps): the octets past the 25th are never written and stay zero;
//   the hidden state's K = 100 is padded to 128 (4 k-steps).
// ---------------------------------------------------------------------------------------------------------
constexpr int F16_KS = 4;        // fused kernel: k-steps of 32 over the hidden state (K = 100 padded to 128)
constexpr int HF16 = 16 * 128;   // halves per h buffer of the fused kernel: [16 k octets][16 rows][8 halves]
// The x tiles of lstm16f_kernel live in DYNAMIC shared memory: with a static array the compiler proves that the LDS-DMA of a
// step and the tile reads of the same step touch one object and puts an s_waitcnt vmcnt(0) between them -- every step would wait
// for the prefetch it has just issued.  The dynamic region starts where the static one ends; its address is taken from
// __builtin_amdgcn_groupstaticsize() (an `extern __shared__` array anywhere near the template makes hipcc 7.2 drop the kernel's
// host stub: "Declaration may not be in a Comdat").
static __device__ __forceinline__ _Float16* lstm16f_xtiles() {
  const unsigned base = (__builtin_amdgcn_groupstaticsize() + 15u) & ~15u;
  return (_Float16*)(__attribute__((address_space(3))) _Float16*)(unsigned long)base;
}
// The LDS-DMA goes through this plain function: the builtin written inside the template kernel makes hipcc 7.2 drop the
// kernel's host stub as well (same diagnostic).
static __device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, _Float16* dst, unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
}
// CHIRON_F16F_VARIANT (timing_variants.h): instrumented builds of this kernel, 0 in the product
template <int KSX, int NG, bool OUT32 = false>
__global__ __launch_bounds__(64 * W16_NW, 1) void lstm16f_kernel(const LstmParams p) {
  // NG = 2: one workgroup carries TWO 16-row groups through the same weight registers (every weight fragment feeds two
  // MFMAs): B = 4096 is then exactly one workgroup per CU, and a wave has two independent chains per step to hide the
  // LDS / barrier / transcendental latencies that a single 8-wave workgroup per CU leaves exposed.
  constexpr int XQ = KSX * 4;                       // k octets (8 halves) of the x tile: KSX k-steps of 32
  constexpr int NTR = NG == 2 ? 3 : 4;              // column-tile slots whose W_x lives in registers (NG = 2: wave 7's fourth in LDS)
  __shared__ __attribute__((aligned(16))) _Float16 hbuf[2 * NG * HF16];
  // OUT32 (the last layer, LstmParams::out_f32): the cells' fp32 h next to its half, same [octet][row][8] positions -- the flush copies
  // 16-byte pieces (4 units) of THIS tile to an fp32 output instead of 8-byte pieces of the h tile
  __shared__ __attribute__((aligned(16))) float hout32[OUT32 ? 2 * NG * HF16 : 4];
  _Float16* const xbuf = lstm16f_xtiles();   // [3][NG][XQ octets][16 rows][8 halves]: three deep, the pieces of step s + 2 fly while s is consumed
  __shared__ __attribute__((aligned(16))) f32x4 biasq[W16_NW * W16_NT * 4];             // [wave][tile slot][unit of the tile] -> (i, j, f, o)
  __shared__ __attribute__((aligned(16))) _Float16 wx3[NG == 2 ? KSX * 64 * 8 : 8];   // wave 7, tile slot 3: W_x fragments

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x % p.ndir;
  const int g0 = (blockIdx.x / p.ndir) * NG;        // first 16-row group of this workgroup
  const int nt = wave == W16_NW - 1 ? 4 : 3;
  const int tile0 = 3 * wave;

  f16x8 wh[W16_NT][F16_KS], wx[NTR][KSX];
  {
    const f16x8* wf = reinterpret_cast<const f16x8*>(p.whfused) + ((long)dir * W16_NW + wave) * W16_NT * F16_KS * 64 + lane;
#pragma unroll
    for (int n = 0; n < W16_NT; ++n)
#pragma unroll
      for (int ks = 0; ks < F16_KS; ++ks) wh[n][ks] = wf[(n * F16_KS + ks) * 64];
    const f16x8* xwf = reinterpret_cast<const f16x8*>(p.wxwide) + ((long)dir * W16_NW + wave) * W16_NT * KSX * 64 + lane;
#pragma unroll
    for (int n = 0; n < NTR; ++n)
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) wx[n][ks] = xwf[(n * KSX + ks) * 64];
    if (NG == 2 && wave == W16_NW - 1) {
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) reinterpret_cast<f16x8*>(wx3)[ks * 64 + lane] = xwf[(3 * KSX + ks) * 64];
    }
  }
  for (int i = tid; i < 2 * NG * HF16; i += 64 * W16_NW) hbuf[i] = (_Float16)0.f;
  for (int i = tid; i < 3 * NG * XQ * 128; i += 64 * W16_NW) xbuf[i] = (_Float16)0.f;

  // The product is computed TRANSPOSED (round 3): the weight fragments are the A operand, the x / h tiles the B operand --
  // both operands have the same lane layout, so it is a swap of the two arguments -- and D = (x W)^T arrives as lane
  // (batch row = lane & 15, unit of the tile = lane >> 4) holding rows 4u .. 4u+3 of D^T: the gates i, j, f, o of ONE
  // cell in the lane's four registers.  No transpose at all (round 2: 1.1 KB of wave-private LDS per tile, one
  // ds_write_b128 + four ds_read_b32 per tile and step, 0.14 of the 1.48 ms launch at B = 4096).
  const int row = lane & 15, u = lane >> 4;
  int maxlen = 0;
  for (int r = 0; r < 16 * NG; ++r) maxlen = max(maxlen, min(p.seq_len[g0 * 16 + r], p.T));

  // bias (+ forget bias) of a cell's four gates: the C operand the tile's first MFMA starts from, read from the LDS every
  // step (16 registers per lane otherwise; the kernel sits at 256)
  if (tid < W16_NW * W16_NT * 4) {
    const int bw = tid >> 4, bn = (tid >> 2) & 3, bu = tid & 3;
    const int unit = 4 * (3 * bw + bn) + bu;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (bn < (bw == W16_NW - 1 ? 4 : 3) && unit < p.H)
      for (int gt = 0; gt < 4; ++gt) b[gt] = p.xbias[dir * 4 * p.H + gt * p.H + unit];
    biasq[tid] = b;
  }

  // ---- x loader (round 3): LDS-DMA, TWO steps ahead.  One instruction per wave, group and step: wave w fetches octets
  //      4w .. 4w+3 of all 16 rows (lane -> octet 4w + (lane >> 4), row lane & 15: the tile's [octet][row][8 halves] order is
  //      lane-linear in exactly that numbering), straight into the buffer of step s + 2 -- no staging registers, no ds_write,
  //      and a row's piece has two whole steps to arrive (it is gathered from HBM: the backward direction walks
  //      t = seq_len - 1 - s per row; round 2 fetched one step ahead into registers and stalled on the gather every step).
  const int xr = lane & 15, xj = 4 * wave + (lane >> 4);
  const bool xwave = 4 * wave < XQ;                  // K = 200: 28 octets, wave 7 has none
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xsrc), 0, 0xFFFE0000u, 0x00027000);
  int xlen[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) xlen[g] = min(p.seq_len[(g0 + g) * 16 + xr], p.T);
  auto x_offset = [&](int g, int s) -> unsigned {   // byte offset of the piece for step s (any valid frame for a finished row)
    int xro = xr;
    asm volatile("" : "+v"(xro));   // recomputed every step: hoisted, the per-group row bases are spilled (256 VGPRs) and every
                                    // reload brings an s_waitcnt vmcnt(0) -- a wait for the prefetch just issued
    const int xb = (g0 + g) * 16 + xro;
    int t = dir == 0 ? s : xlen[g] - 1 - s;
    t = min(max(t, 0), p.T - 1);
    // rows past the submitted batch have length 0 and are never consumed; they read row B - 1 (the feature tensor holds B rows)
    const unsigned r = p.x_time_major ? (unsigned)t * p.BP + xb : (unsigned)min(xb, p.B - 1) * p.T + t;
    return 8 * xj < p.xK ? (r * p.xld + 8 * xj) * 2u : 0xFFFF0000u;   // octets past K read zeros (offset past num_records)
  };
  auto x_issue = [&](int s, int buf) {
    if (!xwave) return;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      lds_dma16(xrs, xbuf + ((buf * NG + g) * XQ + 4 * wave) * 128, x_offset(g, s));
  };
  // barrier of this kernel's loop: LDS writes of the step (h, scratch) complete, then s_barrier -- and NO vmcnt wait (a
  // __syncthreads() drains vmcnt while an LDS-DMA is pending, which would put the prefetch back to one step)
  auto step_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  __syncthreads();   // the zero fill above is complete before the first pieces land
  if (maxlen > 0) {
    x_issue(0, 0);
    x_issue(1, 1);
  }
  // (the builtin, not inline asm: the compiler's own wait-count bookkeeping has to see that nothing is pending at loop entry,
  //  or it keeps a vmcnt(0) for the pre-loop loads INSIDE the loop, where it would wait for every step's prefetch)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0); expcnt / lgkmcnt untouched
  __syncthreads();

  const unsigned outw = p.ndir * p.H;
  const unsigned ostep = p.BP * outw;
  _Float16* outh = reinterpret_cast<_Float16*>(p.out);
  // ---- outputs (round 3): the h tile every wave completed in step s - 1 is the output of that step; it is copied out at the
  //      top of step s in 8-byte pieces (4 units): 16 rows x 25 pieces x NG groups = 50 NG per wave -- NG store instructions per
  //      wave and step, each covering whole 200-byte output rows, instead of one 2-byte store per cell (6 .. 8 instructions
  //      touching 16 rows each).  The cell phase then knows nothing about sequence lengths: a finished row keeps computing
  //      (rows are independent columns of the product; its state is never read again) and its outputs are written as zeros here.
  unsigned f_lds[NG], f_out[NG];
  int f_len[NG];
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    const int P = min(50 * NG * wave + 50 * k + lane, 400 * NG - 1);   // piece of this lane (lanes 50 .. 63 idle)
    const int fg = P / 400, rem = P - 400 * fg, fr = rem / 25, fq = rem - 25 * fr;
    f_lds[k] = fg * HF16 + ((fq >> 1) * 16 + fr) * 8 + 4 * (fq & 1);
    f_out[k] = ((g0 + fg) * 16 + fr) * outw + dir * p.H + 4 * fq;
    f_len[k] = min(p.seq_len[(g0 + fg) * 16 + fr], p.T);
  }
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
  auto flush = [&](int sp, int buf) {   // the outputs of step sp, h tile `buf`
    if (lane < 50) {
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const bool act = sp < f_len[k];
        const unsigned to = (dir == 0 || !act) ? sp : f_len[k] - 1 - sp;
        if (OUT32) {
          f32x4 v = *reinterpret_cast<const f32x4*>(hout32 + buf * NG * HF16 + f_lds[k]);
          if (!act) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(p.out + (to * ostep + f_out[k])) = v;
        } else {
          u32x2v v = *reinterpret_cast<const u32x2v*>(hbuf + buf * NG * HF16 + f_lds[k]);
          if (!act) v = (u32x2v){0u, 0u};
          *reinterpret_cast<u32x2v*>(outh + (to * ostep + f_out[k])) = v;
        }
      }
    }
  };
  // cell (row, unit 4 T + u) of column tile T: octet T / 2, element 4 (T % 2) + u  ->  + 4 per tile, + 120 more every second
  auto h_pos = [&](int n) -> int { const int T = tile0 + n; return ((T >> 1) * 16 + row) * 8 + 4 * (T & 1) + u; };
  const f32x4* const biasl = biasq + (wave * W16_NT) * 4 + u;

  float c[NG][W16_NT];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int n = 0; n < W16_NT; ++n) c[g][n] = 0.f;
  int cur = 0, xcur = 0;   // h buffer / x buffer of this step
  for (int s = 0; s < maxlen; ++s) {
#if !(CHIRON_F16F_VARIANT & 8)
    x_issue(s + 2, xcur == 0 ? 2 : xcur - 1);   // into the buffer consumed in step s - 1 (every wave is past that step's barrier)
#endif
#if !(CHIRON_F16F_VARIANT & 4)
    if (s > 0) flush(s - 1, cur);
#endif
    f32x4 acc[NG][W16_NT];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int n = 0; n < W16_NT; ++n) acc[g][n] = n < nt ? biasl[4 * n] : (f32x4){0.f, 0.f, 0.f, 0.f};
#if CHIRON_F16F_VARIANT == 0
    // One branch-free sequence of KSX + 4 k-steps for the three tile slots every wave has, the B fragments of k-step k + 1 read
    // before the MFMAs of k-step k are issued (with the fourth slot tested inside the loop every k-step ended in a branch, and
    // the compiler put each k-step's tile reads, their wait and its MFMAs strictly behind one another); wave 7's fourth tile
    // follows on its own and reads the fragments again.
    {
      auto frag = [&](int k, int g) -> f16x8 {   // k < KSX: x tile, else h tile
        return k < KSX ? reinterpret_cast<const f16x8*>(xbuf + (xcur * NG + g) * XQ * 128)[k * 64 + lane]
                       : reinterpret_cast<const f16x8*>(hbuf + (cur * NG + g) * HF16)[(k - KSX) * 64 + lane];
      };
      f16x8 fb[2][NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) fb[0][g] = frag(0, g);
#pragma unroll
      for (int k = 0; k < KSX + F16_KS; ++k) {
        if (k + 1 < KSX + F16_KS) {
#pragma unroll
          for (int g = 0; g < NG; ++g) fb[(k + 1) & 1][g] = frag(k + 1, g);
        }
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          const f16x8 wv = k < KSX ? wx[n][k < KSX ? k : 0] : wh[n][k < KSX ? 0 : k - KSX];
#pragma unroll
          for (int g = 0; g < NG; ++g) acc[g][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, fb[k & 1][g], acc[g][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // one k-step of look-ahead, not eleven (the kernel sits at 246 registers)
      }
      if (nt == 4) {   // wave-uniform
#pragma unroll
        for (int k = 0; k < KSX + F16_KS; ++k) {
          f16x8 wv;
          if (k < KSX) {
            if (3 < NTR) wv = wx[3 < NTR ? 3 : 0][k < KSX ? k : 0];
            else wv = reinterpret_cast<const f16x8*>(wx3)[(k < KSX ? k : 0) * 64 + lane];
          } else {
            wv = wh[3][k < KSX ? 0 : k - KSX];
          }
#pragma unroll
          for (int g = 0; g < NG; ++g) acc[g][3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, frag(k, g), acc[g][3], 0, 0, 0);
          if (k & 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#else
#pragma unroll
    for (int ks = 0; ks < KSX; ++ks) {
      f16x8 xa[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#if CHIRON_F16F_VARIANT & 16
        xa[g] = wh[0][g];
#else
        xa[g] = reinterpret_cast<const f16x8*>(xbuf + (xcur * NG + g) * XQ * 128)[ks * 64 + lane];
#endif
#pragma unroll
      for (int n = 0; n < W16_NT; ++n)
        if (n < nt) {
          f16x8 wv;
          if (n < NTR) wv = wx[n < NTR ? n : 0][ks];
          else wv = reinterpret_cast<const f16x8*>(wx3)[ks * 64 + lane];
#pragma unroll
          for (int g = 0; g < NG; ++g)
#if CHIRON_F16F_VARIANT & 2
            acc[g][n][0] += (float)xa[g][0] * (float)wv[0];
#else
            acc[g][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xa[g], acc[g][n], 0, 0, 0);
#endif
        }
    }
#pragma unroll
    for (int ks = 0; ks < F16_KS; ++ks) {
      f16x8 ha[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#if CHIRON_F16F_VARIANT & 32
        ha[g] = wh[1][g];
#else
        ha[g] = reinterpret_cast<const f16x8*>(hbuf + (cur * NG + g) * HF16)[ks * 64 + lane];
#endif
#pragma unroll
      for (int n = 0; n < W16_NT; ++n)
        if (n < nt) {
#pragma unroll
          for (int g = 0; g < NG; ++g)
#if CHIRON_F16F_VARIANT & 2
            acc[g][n][0] += (float)ha[g][0] * (float)wh[n][ks][0];
#else
            acc[g][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n][ks], ha[g], acc[g][n], 0, 0, 0);
#endif
        }
    }
#endif
    // the pieces of step s + 1 (issued during step s - 1) have landed when at most this step's NG instructions are
    // outstanding.  The wait sits HERE, behind the products and before this step's output stores: vmcnt counts loads and stores
    // together, and behind the stores (round 2) it made every step wait for its own stores' write acknowledgements.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(NG == 2 ? 0x0F74 : 0x0F72);   // vmcnt(2 NG)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int n = 0; n < W16_NT; ++n) {
        if (n < nt) {
          const f32x4 gates = acc[g][n];
          float hnew;
#if CHIRON_F16F_VARIANT & 1
          const float cn = gates[0] + gates[1] + gates[2] + gates[3] + c[g][n];
          hnew = 0.5f * cn;
#else
          const float cn = lstm_cell(gates, c[g][n], &hnew);
#endif
          c[g][n] = cn;
#if !(CHIRON_F16F_VARIANT & 64)
          hbuf[((cur ^ 1) * NG + g) * HF16 + h_pos(n)] = (_Float16)hnew;
          if (OUT32) hout32[((cur ^ 1) * NG + g) * HF16 + h_pos(n)] = hnew;
#else
          c[g][n] += hnew;
#endif
        }
      }
    }
    cur ^= 1;
    xcur = xcur == 2 ? 0 : xcur + 1;
    step_barrier();
  }
  if (maxlen > 0) flush(maxlen - 1, cur);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last two prefetches (never consumed) before the workgroup may retire

  for (int s = maxlen; s < p.T; ++s)
    for (int i = tid; i < 16 * NG * p.H; i += 64 * W16_NW) {
      const int r = i / p.H;
      const int uu = i - r * p.H;
      if (OUT32) p.out[((long)s * p.BP + g0 * 16 + r) * outw + dir * p.H + uu] = 0.f;
      else outh[((long)s * p.BP + g0 * 16 + r) * outw + dir * p.H + uu] = (_Float16)0.f;
    }
}


// ---------------------------------------------------------------------------------------------------------
// fp32, wide form (round 3): one workgroup = SIXTEEN batch rows x one direction on v_mfma_f32_16x16x4_f32
// (M = 16 rows, N = 16 columns = 4 units x 4 gates, K = 4; 8 passes = 32 cycles for 2048 FLOP: the full 64 FLOP/clk/SIMD,
// where the 4x4x1 form of lstm_kernel pays 10 cycles for 512 FLOP).  B = 1100 gives 69 x 2 = 138 workgroups of 8 waves,
// one per CU: on its own such a launch uses 54 % of the CUs and takes about as long as the 550 four-row workgroups that
// fill the chip twice over -- but the engine keeps three batches in flight, and what counts there is CU-time: a layer's
// recurrence costs 138 CUs x 1.1 ms instead of 256 CUs x 1.1 ms, and the other batches' GEMMs run on the CUs it leaves
// alone (the 4-row form's two workgroups per CU hold 496 of a SIMD's 512 registers: nothing co-resides with them, while
// their matrix pipes are 41 % busy).  DESIGN 3.2.
//   wave w owns the column tiles 3w .. 3w+2 (wave 7: 21 .. 24) of the 25 (H = 100 = 25 x 4 units); their W_hh slices
//   (25 k-steps x 1 register per tile) stay in VGPRs for the whole sequence;
//   A = h_{t-1} [16 rows][K = 100]: lane (row = lane & 15, kq = lane >> 4) reads h[row][4 ks + kq] for k-step ks; the LDS
//     tile is [ks][lane]: reads are lane-linear, and the 64 cells of a column tile write 64 consecutive floats;
//   D: lane (col = lane & 15 = 4 u + gate, q = lane >> 4) holds rows 4q .. 4q+3 of its column; z arrives in the projection's
//     layout [4-row group][dir][col = gate*H + unit][4 rows] -- the 16 bytes this lane needs -- and is added in place, then
//     the 4 x 4 transpose inside each lane quad goes through wave-private LDS (as in lstm16w_kernel) so that lane (q, u, r)
//     holds i, j, f, o of cell (row 4q + r, unit 4 tile + u).
// Every row of a batch takes this kernel whatever the batch size (partial last groups included), so a window's result
// does not depend on the batch it travels in; the accumulation order differs from lstm_kernel's (4 k per MFMA), so the two
// forms agree to rounding (CHIRON_LSTM_WIDE=1 / =0 selects the form: A/B switch and test partner).
// ---------------------------------------------------------------------------------------------------------
constexpr int W32_KS = 25;      // k-steps of 4: K = 100
constexpr int HW32 = W32_KS * 64;   // floats per h buffer: [ks][kq*16 + row]

__global__ __launch_bounds__(64 * W16_NW, 1) void lstm32w_kernel(const LstmParams p) {
  __shared__ __attribute__((aligned(16))) float hbuf[2 * HW32];
  __shared__ __attribute__((aligned(16))) float xf[W16_NW * W16_NT * W16_XF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x % p.ndir;
  const int g16 = blockIdx.x / p.ndir;                 // 16-row group = the 4-row groups 4 g16 .. 4 g16 + 3
  const int nt = wave == W16_NW - 1 ? 4 : 3;           // column tiles of this wave
  const int tile0 = 3 * wave;

  float w[W16_NT][W32_KS];
  {
    const float* wf = p.wwide32 + ((long)dir * W16_NW + wave) * W16_NT * W32_KS * 64 + lane;
#pragma unroll
    for (int n = 0; n < W16_NT; ++n)
#pragma unroll
      for (int ks = 0; ks < W32_KS; ++ks) w[n][ks] = wf[(n * W32_KS + ks) * 64];
  }
  for (int i = tid; i < 2 * HW32; i += 64 * W16_NW) hbuf[i] = 0.f;

  const int q = lane >> 4, u = (lane >> 2) & 3, gp = lane & 3;   // before the transpose: column 4u + gp, rows 4q .. 4q+3
  const int row = 4 * q + gp;                                      // after it: this lane's cell is (row, unit 4 tile + u)
  const int brow = g16 * 16 + row;
  const int lenr = min(p.seq_len[brow], p.T);
  int maxlen = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) maxlen = max(maxlen, min(p.seq_len[g16 * 16 + r], p.T));
  __syncthreads();

  const unsigned outw = p.ndir * p.H;
  const unsigned zcols = 4 * p.H;
  const unsigned zstep = (p.BP >> 2) * p.ndir * zcols * 4;        // floats between consecutive steps
  // byte offset of this lane's 16 bytes (rows 4q..4q+3 of column gate*H + unit) for tile slot 0; + 64 bytes per tile
  const unsigned zlane_b = ((((g16 * 4 + q) * p.ndir + dir) * zcols + gp * p.H + 4 * tile0 + u) * 4) * 4;
  const unsigned ostep = p.BP * outw;
  const unsigned olane = brow * outw + dir * p.H + 4 * tile0 + u;  // + 4 per tile
  const int hw = tile0 * 64 + u * 16 + row;                         // + 64 per tile: [ks = tile][kq = u][row]

  // transpose scratch: lane l = (q, u, gp) stores its 4 rows at float 4 l + 4 q; lane (q, u, r) then reads gate g at
  // float 4 (16 q + 4 u + g) + 4 q + r: bank 16 u + 4 g + 4 q + r, distinct over the wave for every g
  float* const xw = xf + wave * W16_NT * W16_XF + 4 * lane + 4 * q;
  const float* const xr = xf + wave * W16_NT * W16_XF + 16 * (4 * q + u) + 4 * q + gp;

  float c[W16_NT] = {0.f, 0.f, 0.f, 0.f}, hprev[W16_NT] = {0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  for (int s = 0; s < maxlen; ++s) {
    const float* zs = p.z + (size_t)s * zstep;   // wave-uniform
    f32x4 z4[W16_NT];
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(z4[0]) : "v"(zlane_b), "s"(zs) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(z4[1]) : "v"(zlane_b), "s"(zs) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:128" : "=v"(z4[2]) : "v"(zlane_b), "s"(zs) : "memory");
    if (nt == 4) asm volatile("global_load_dwordx4 %0, %1, %2 offset:192" : "=v"(z4[3]) : "v"(zlane_b), "s"(zs) : "memory");
    else z4[3] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* hb = hbuf + cur * HW32 + lane;
    float hv[W32_KS];
#pragma unroll
    for (int ks = 0; ks < W32_KS; ++ks) hv[ks] = hb[ks * 64];
    f32x4 acc[W16_NT];
#pragma unroll
    for (int n = 0; n < W16_NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if CHIRON_W32_VARIANT == 2
#pragma unroll
    for (int n = 0; n < W16_NT; ++n) acc[n][0] = hv[n] + w[n][0] + hv[n + 8] + hv[24];
#else
#pragma unroll
    for (int ks = 0; ks < W32_KS; ++ks) {
#pragma unroll
      for (int n = 0; n < W16_NT; ++n)
        if (n < nt) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[ks], w[n][ks], acc[n], 0, 0, 0);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(z4[0]), "+v"(z4[1]), "+v"(z4[2]), "+v"(z4[3]), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    const bool act = s < lenr;
    const unsigned to = (dir == 0 || !act) ? s : lenr - 1 - s;
#if CHIRON_W32_VARIANT != 3
#pragma unroll
    for (int n = 0; n < W16_NT; ++n)
      if (n < nt) *reinterpret_cast<f32x4*>(xw + n * W16_XF) = acc[n] + z4[n];
    __builtin_amdgcn_wave_barrier();   // wave-private scratch: LDS operations of a wave execute in order
#endif
#pragma unroll
    for (int n = 0; n < W16_NT; ++n) {
      if (n < nt) {
        const float* xs = xr + n * W16_XF;
#if CHIRON_W32_VARIANT == 3
        const f32x4 gates = acc[n] + z4[n];
#else
        const f32x4 gates = {xs[0], xs[4], xs[8], xs[12]};   // i, j, f, o of (row, unit 4 (tile0 + n) + u)
#endif
        float hnew;
#if CHIRON_W32_VARIANT == 1
        const float cn = gates[0] + gates[1] * 0.001f;
        hnew = gates[2] * 0.001f + gates[3] * 0.002f;
#else
        const float cn = lstm_cell(gates, c[n], &hnew);
#endif
        c[n] = act ? cn : c[n];
        hprev[n] = act ? hnew : hprev[n];
        hbuf[(cur ^ 1) * HW32 + hw + 64 * n] = hprev[n];
        const unsigned ob = (to * ostep + olane + 4 * n) * 4u;
        *reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + ob) = act ? hnew : 0.f;
      }
    }
    cur ^= 1;
    __syncthreads();
  }

  // ---- frames past the longest row of the workgroup read back as zeros (dynamic_rnn semantics)
  for (int s = maxlen; s < p.T; ++s)
    for (int i = tid; i < 16 * p.H; i += 64 * W16_NW) {
      const int r = i / p.H;
      const int uu = i - r * p.H;
      p.out[((long)s * p.BP + g16 * 16 + r) * outw + dir * p.H + uu] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// fp32 wide form, balanced (lstm32w2_kernel; CHIRON_LSTM_WIDE=2).  lstm32w_kernel gives wave 7 four of the 25 column tiles:
// its SIMD carries 7 tiles x 25 k-steps x 32 cycles = 5600 matrix-pipe cycles per step and 7 cells of gate math, the others
// 4800 and 6.  Here every wave owns THREE tiles (0 .. 23) and the 25th (units 96..99) is K-split:
//   waves 0, 1, 2 each multiply 8 of its first 24 k-steps (partial 16 x 16 products, handed to wave 7 through LDS);
//   wave 7 (whose SIMD partner, wave 3, takes no partial) adds the three partials, its own product of the LAST k-step and z,
//   and does the tile's gate math.
// The last k-step is k = 96..99 -- the units of that very tile -- so the step is split in two: k-steps 0..23 need only
// h[0..95], which every wave has written before barrier A; wave 7 finishes tile 24 of the PREVIOUS step while the others
// are already multiplying, writes h[96..99], barrier B, then everybody issues the 25th k-step.  Per SIMD: 156 MFMAs (5000
// cycles) and 6 cells of gate math, wave 7's seventh overlapping the other SIMDs' products.
// Same arithmetic per row as lstm32w_kernel for tiles 0..23; tile 24 sums its partial products in the fixed order
// ((p0 + p1) + p2) + (p24 + z): agrees with the other forms to rounding, and a row's bits do not depend on its batch.
// ---------------------------------------------------------------------------------------------------------
constexpr int W32_PK = 8;   // k-steps of tile 24 per partial wave (waves 0, 1, 2)
// Round 3, second pass (what the f16 fused kernel taught): on a SIMD that runs fp32 MFMAs, every VALU instruction of a
// wave64 costs 4 cycles ON TOP of the matrix time, and this kernel spent 340 of them per wave and step.  Now
//   * the product is computed TRANSPOSED (weight fragment = A operand, h = B operand: a swap of the two arguments): lane
//     (batch row = lane & 15, unit of the tile = lane >> 4) receives the gates i, j, f, o of ONE cell in its four accumulator
//     registers -- no transpose through the LDS (3 ds_write_b128 + 12 ds_read_b32 per wave and step) -- and writes h back at
//     tile * 64 + lane: lane-linear;
//   * z is read as four dwords per tile (gate g of the lane's cell: [4-row group][dir][g * H + unit][row % 4], 64-byte segments)
//     instead of one 16-byte column of four rows;
//   * outputs leave from the completed h tile, 16 bytes (4 units) per thread and step, and the cell phase knows nothing about
//     sequence lengths (a finished row keeps computing -- rows are independent columns of the product -- and is written as
//     zeros): no selects, no per-cell address arithmetic, no 4-byte stores;
//   * tile 24's partial products sit under a wave-uniform branch per 8-k-step chunk, not per k-step.
#define W32_ZLOAD(dst, vo, off) asm volatile("global_load_dword %0, %1, %2 offset:" #off : "=v"(dst) : "v"(vo), "s"(zs) : "memory")

__device__ __forceinline__ float sens_cell(f32x4 q, float c, float* h_out) {
#if CHIRON_SENS & 1
  const float cn = 0.25f * (q[0] + q[1]) + 0.5f * c;
  *h_out = 0.25f * (q[2] + q[3]) + 0.1f * cn;
  return cn;
#else
  return lstm_cell(q, c, h_out);
#endif
}
__global__ __launch_bounds__(64 * W16_NW, 1) void lstm32w2_kernel(const LstmParams p) {
  __shared__ __attribute__((aligned(16))) float hbuf[2 * HW32];
  __shared__ __attribute__((aligned(16))) float part[4 * 256];    // [partial wave 0..2 | wave 7's own last-k-step product + z][lane][4 gates]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x % p.ndir;
  const int g16 = blockIdx.x / p.ndir;
  const int tile0 = 3 * wave;
  const bool partial = wave < 3, owner24 = wave == W16_NW - 1;

  float w[3][W32_KS], wp[W32_PK];
  {
    // a wave's own tiles: wave w, slots 0..2 of the fragment array; tile 24 = wave 7, slot 3
    const float* wf = p.wwide32 + ((long)dir * W16_NW + wave) * W16_NT * W32_KS * 64 + lane;
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int ks = 0; ks < W32_KS; ++ks) w[n][ks] = wf[(n * W32_KS + ks) * 64];
    const float* w24 = p.wwide32 + (((long)dir * W16_NW + 7) * W16_NT + 3) * W32_KS * 64 + lane;
#pragma unroll
    for (int j = 0; j < W32_PK; ++j) wp[j] = partial ? w24[(W32_PK * wave + j) * 64] : (owner24 && j == 0 ? w24[24 * 64] : 0.f);
  }
  for (int i = tid; i < 2 * HW32; i += 64 * W16_NW) hbuf[i] = 0.f;

  const int row = lane & 15, q = lane >> 4;          // accumulator lane: batch row, unit of the tile
  int maxlen = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) maxlen = max(maxlen, min(p.seq_len[g16 * 16 + r], p.T));
  __syncthreads();

  const unsigned outw = p.ndir * p.H;
  const unsigned zcols = 4 * p.H;
  const unsigned zstep = (p.BP >> 2) * p.ndir * zcols * 4;
  // gate g of cell (row, unit 4 T + q): + g * H * 16 bytes (H = 100: 1600), + 64 bytes per tile; the 13-bit immediate reaches
  // gates 0, 1 from zlo and 2, 3 from zhi
  const unsigned zlo = (((((g16 * 4 + (row >> 2)) * p.ndir + dir) * zcols + 4 * tile0 + q) * 4) + (row & 3)) * 4;
  const unsigned zhi = zlo + 2 * 1600;
  // (tile 24 belongs to wave 7, tile0 = 21: its z is 3 tiles = 192 bytes behind the wave's first tile)
  const unsigned ostep = p.BP * outw;

  // ---- outputs: the h tile completed in step s - 1 is copied out in step s, 16 bytes (units 4 P .. 4 P + 3 of one row) per
  //      thread: lane -> row lane & 15 (consecutive lanes read consecutive floats of the [k-step][unit % 4][row] tile), piece
  //      4 wave + lane / 16: a store instruction covers 64 contiguous bytes of each of 16 rows; pieces 0..24: waves 0..5, a
  //      quarter of wave 6
  //      (row, piece and the two offsets are recomputed every step from the lane number: the kernel has to stay at 168 registers
  //      for a conv GEMM wave of another batch to fit next to two of its waves on a SIMD; only the row's length is kept)
  const int f_len = min(p.seq_len[g16 * 16 + (lane & 15)], p.T);
  auto flush = [&](int sp, int buf) {
    int lo = lane;
    asm volatile("" : "+v"(lo));   // not hoisted out of the step loop
    const int fr = lo & 15, fq = 4 * wave + (lo >> 4);
    if (fq < 25) {
      const bool act = sp < f_len;
      const unsigned to = (dir == 0 || !act) ? sp : f_len - 1 - sp;
      const float* hp = hbuf + buf * HW32 + fq * 64 + fr;   // + 16 per unit
      f32x4 v = {hp[0], hp[16], hp[32], hp[48]};
      if (!act) v = (f32x4){0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(p.out + (to * ostep + (g16 * 16 + fr) * outw + dir * p.H + 4 * fq)) = v;
    }
  };

  float c[3] = {0.f, 0.f, 0.f};
  float c24 = 0.f;
  int cur = 0;

  // wave 7: tile 24 of the previous step (its partial products are in `part`, the last k-step's product + z in acc24)
  auto finish24 = [&](int buf) {
    const f32x4* pp = reinterpret_cast<const f32x4*>(part) + lane;
    f32x4 sum = pp[0] + pp[64];
    sum = sum + pp[128];
    float hnew;
    c24 = sens_cell(sum + pp[192], c24, &hnew);   // pp[192]: wave 7's own last-k-step product + z (kept in the LDS, not in registers)
    hbuf[buf * HW32 + 24 * 64 + lane] = hnew;
  };

  for (int s = 0; s < maxlen; ++s) {
    const float* zs = p.z + (size_t)s * zstep;
    f32x4 z4[3], z24 = {0.f, 0.f, 0.f, 0.f};
    W32_ZLOAD(z4[0][0], zlo, 0);   W32_ZLOAD(z4[0][1], zlo, 1600);   W32_ZLOAD(z4[0][2], zhi, 0);   W32_ZLOAD(z4[0][3], zhi, 1600);
    W32_ZLOAD(z4[1][0], zlo, 64);  W32_ZLOAD(z4[1][1], zlo, 1664);   W32_ZLOAD(z4[1][2], zhi, 64);  W32_ZLOAD(z4[1][3], zhi, 1664);
    W32_ZLOAD(z4[2][0], zlo, 128); W32_ZLOAD(z4[2][1], zlo, 1728);   W32_ZLOAD(z4[2][2], zhi, 128); W32_ZLOAD(z4[2][3], zhi, 1728);
    if (owner24) {
      W32_ZLOAD(z24[0], zlo, 192); W32_ZLOAD(z24[1], zlo, 1792); W32_ZLOAD(z24[2], zhi, 192); W32_ZLOAD(z24[3], zhi, 1792);
    }
    // ---- wave 7 first completes tile 24 of the previous step: h[96..99] of that step is what the last k-step below needs
    if (owner24 && s > 0) finish24(cur);
    // ---- k-steps 0 .. 23: h[0..95] of the previous step (complete since barrier A), eight at a time, the next eight in flight
    const float* hb = hbuf + cur * HW32 + lane;
    // (the first product of a chain takes a literal zero as its C operand: no accumulator is cleared with moves)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[3], accp = zero4;
    float hv[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) hv[0][j] = hb[j * 64];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      if (ch < 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[(ch + 1) & 1][j] = hb[(8 * (ch + 1) + j) * 64];
      }
      if (partial && ch == wave) {   // (wave-uniform: one branch per chunk)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
          for (int n = 0; n < 3; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][8 * ch + j], hv[ch & 1][j], (ch == 0 && j == 0) ? zero4 : acc[n], 0, 0, 0);
          accp = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[j], hv[ch & 1][j], j == 0 ? zero4 : accp, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int n = 0; n < 3; ++n) {
#if CHIRON_SENS & 8
            if (j & 1) { acc[n][0] += w[n][8 * ch + j] * hv[ch & 1][j]; continue; }   // timing experiment: half of the MFMAs
#endif
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][8 * ch + j], hv[ch & 1][j], (ch == 0 && j == 0) ? zero4 : acc[n], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // barrier B: h[96..99] of the previous step is in place
    // ---- the 25th k-step
    const float h24 = hb[(W32_KS - 1) * 64];
#pragma unroll
    for (int n = 0; n < 3; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][W32_KS - 1], h24, acc[n], 0, 0, 0);
    f32x4 acc24 = {0.f, 0.f, 0.f, 0.f};
    if (owner24) acc24 = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[0], h24, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    // z has arrived (vmcnt also counts the previous step's output stores: a step old); this step's output stores go out BEHIND
    // this wait
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(z4[0]), "+v"(z4[1]), "+v"(z4[2]), "+v"(z24) : : "memory");
    if (owner24) reinterpret_cast<f32x4*>(part)[192 + lane] = acc24 + z24;
    if (s > 0) flush(s - 1, cur);
    if (partial) reinterpret_cast<f32x4*>(part)[wave * 64 + lane] = accp;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      float hnew;
      c[n] = sens_cell(acc[n] + z4[n], c[n], &hnew);
      hbuf[(cur ^ 1) * HW32 + (tile0 + n) * 64 + lane] = hnew;
    }
    cur ^= 1;
    __syncthreads();   // barrier A: h[0..95] of this step and the partial products of tile 24 are in place
  }
  if (maxlen > 0) {
    if (owner24) finish24(cur);
    __syncthreads();
    flush(maxlen - 1, cur);
  }

  for (int s = maxlen; s < p.T; ++s)
    for (int i = tid; i < 16 * p.H; i += 64 * W16_NW) {
      const int r = i / p.H;
      const int uu = i - r * p.H;
      p.out[((long)s * p.BP + g16 * 16 + r) * outw + dir * p.H + uu] = 0.f;
    }
}

static int lstm_cu_count() { return current_device_cus(); }

void launch_lstm(const LstmParams& p0, hipStream_t stream) {
  // hidden = 100 is the only size the reference's shipped models use (rnn.py:23 hidden_num=100)
  LstmParams p = p0;
  p.group0 = 0;
  const int groups = p.BP / 4;
  if (p.f16 && p.xsrc) {   // fused with the x-projection: the engine asks for it only when whole 16-row groups cover the batch
    // two 16-row groups per workgroup when the padded batch is whole 32-row pairs (4096: one workgroup per CU)
    const int g16 = p.BP / 16;
    const dim3 blk(64 * W16_NW);
    if (p.out_f32) {   // the OUT32 forms hold 58 KB of static LDS next to 25 .. 49 KB of dynamic x tiles: ask for the dynamic part explicitly, once
      static const bool attr_ok = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(lstm16f_kernel<8, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * 32 * 128 * 2 + 16) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(lstm16f_kernel<7, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * 28 * 128 * 2 + 16) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(lstm16f_kernel<8, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32 * 128 * 2 + 16) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(lstm16f_kernel<7, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 28 * 128 * 2 + 16) == hipSuccess;
      }();
      (void)attr_ok;
    }
    if (g16 % 2 == 0 && p.fused_pair) {
      const dim3 grd((g16 / 2) * p.ndir);
      if (p.xK > 224) {
        if (p.out_f32) hipLaunchKernelGGL((lstm16f_kernel<8, 2, true>), grd, blk, 3 * 2 * 32 * 128 * 2 + 16, stream, p);
        else hipLaunchKernelGGL((lstm16f_kernel<8, 2>), grd, blk, 3 * 2 * 32 * 128 * 2 + 16, stream, p);
      } else {
        if (p.out_f32) hipLaunchKernelGGL((lstm16f_kernel<7, 2, true>), grd, blk, 3 * 2 * 28 * 128 * 2 + 16, stream, p);
        else hipLaunchKernelGGL((lstm16f_kernel<7, 2>), grd, blk, 3 * 2 * 28 * 128 * 2 + 16, stream, p);
      }
    } else {
      const dim3 grd(g16 * p.ndir);
      if (p.xK > 224) {
        if (p.out_f32) hipLaunchKernelGGL((lstm16f_kernel<8, 1, true>), grd, blk, 3 * 32 * 128 * 2 + 16, stream, p);
        else hipLaunchKernelGGL((lstm16f_kernel<8, 1>), grd, blk, 3 * 32 * 128 * 2 + 16, stream, p);
      } else {
        if (p.out_f32) hipLaunchKernelGGL((lstm16f_kernel<7, 1, true>), grd, blk, 3 * 28 * 128 * 2 + 16, stream, p);
        else hipLaunchKernelGGL((lstm16f_kernel<7, 1>), grd, blk, 3 * 28 * 128 * 2 + 16, stream, p);
      }
    }
    return;
  }
  if (p.f16 && p.w2) {   // dtype f16-w2: exact recurrent weights (hi + lo), sixteen-row workgroups for every row of the padded batch
    if (p.w2 == 2)
      hipLaunchKernelGGL(lstm16w2_kernel<true>, dim3((p.BP / 16) * p.ndir), dim3(64 * W16_NW), 0, stream, p);
    else
      hipLaunchKernelGGL(lstm16w2_kernel<false>, dim3((p.BP / 16) * p.ndir), dim3(64 * W16_NW), 0, stream, p);
    return;
  }
  if (p.f16) {
    // sixteen-row workgroups for whole 16-row groups, 4-row workgroups for what is left of the padded batch
    // (138 wide workgroups of B = 1100 leave half the CUs idle: 0.68 ms against 0.53 ms for 550 narrow ones)
    const int wide = (p.wwide && !p.narrow16 && (p.BP / 16) * p.ndir >= lstm_cu_count()) ? p.BP / 16 : 0;
    if (wide > 0) hipLaunchKernelGGL(lstm16w_kernel, dim3(wide * p.ndir), dim3(64 * W16_NW), 0, stream, p);
    p.group0 = 4 * wide;
    if (groups > p.group0) hipLaunchKernelGGL(lstm16_kernel, dim3((groups - p.group0) * p.ndir), dim3(64 * LSTM_NW), 0, stream, p);
    return;
  }
  if (p.wsplit) {   // dtype fp32-split: the recurrence on the f16 pipe (hi + lo pairs), sixteen-row workgroups for every row of the padded batch
    hipLaunchKernelGGL(lstm32s_kernel, dim3((p.BP / 16) * p.ndir), dim3(64 * W16_NW), 0, stream, p);
    return;
  }
  if (p.wwide32 && p.form32 > 0 && !p.paired) {   // sixteen-row workgroups for every row of the padded batch
    if (p.form32 == 2)
      hipLaunchKernelGGL(lstm32w2_kernel, dim3((p.BP / 16) * p.ndir), dim3(64 * W16_NW), 0, stream, p);
    else
      hipLaunchKernelGGL(lstm32w_kernel, dim3((p.BP / 16) * p.ndir), dim3(64 * W16_NW), 0, stream, p);
    return;
  }
  // Paired workgroups (one per CU) for as many groups as fit ONE resident round, 7-wave workgroups for the rest: a
  // paired workgroup fills its CU, so a second round of them would cost a whole round however few there are.
  const int n_cu = lstm_cu_count();
  const int pairs = p.paired ? std::min(groups / 2, n_cu / p.ndir) : 0;
  if (pairs > 0) hipLaunchKernelGGL(lstm_kernel<2>, dim3(pairs * p.ndir), dim3(64 * LSTM_NW * 2), 0, stream, p);
  p.group0 = 2 * pairs;
  if (groups > p.group0) hipLaunchKernelGGL(lstm_kernel<1>, dim3((groups - p.group0) * p.ndir), dim3(64 * LSTM_NW), 0, stream, p);
}

}  // namespace chiron
