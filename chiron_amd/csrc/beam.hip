// CTC prefix beam search on device: tf.nn.ctc_beam_search_decoder(merge_repeated=False, top_paths=1)
// as the reference calls it (chiron_eval.py:489-492).  Follows TF 1.15
// core/util/ctc/ctc_beam_search.h CTCBeamSearchDecoder<>::Step / TopPaths *sequentially*, because the
// decoder's result depends on its processing order: while it grows new leaves (branches visited in
// descending old probability against a moving top-N bottom) it resets the old probability of any child
// it rejects -- and that child may itself be a branch still waiting in the loop, which then never
// spawns children.  A set-based "top-N of all candidates" formulation decodes differently on flat
// posteriors, so the grow phase below is the literal sequential walk (oracle/ctc_oracle.py,
// oracle/chiron_oracle.c restate the same thing on the CPU).
//
//   one wave per segment.  Per frame:
//     P1 (lane-parallel)  every beam entry extends itself: blank / repeat / parent hand-over (TF
//                         "Step", first loop); its 4 child node ids and their beam slots are gathered
//                         from the trie in HBM into LDS so that ...
//     P2 (wave-uniform)   ... the sequential grow walk touches LDS only; the top-N "bottom" is a wave
//                         arg-min, recomputed only after an insertion.
//     P3 (lane-parallel)  leaves are ranked (descending total) into the next beam, evicted nodes lose
//                         their slot, surviving new children are materialised as trie nodes.
//   The prefix trie (parent / label / children / slot / depth) lives in a per-segment slab of HBM with
//   1 + beam*T nodes (only leaves that survive a frame are materialised).
#include "kernels.h"
#include "ctc_math.h"

#include <cstdlib>

namespace chiron {

constexpr int BEAM_MAX = 256;
constexpr float NEG_INF = -INFINITY;

struct __attribute__((aligned(32))) BeamNode {
  int parent;
  int label;
  int child[4];
  int slot;   // index in the current beam (at frame start), -1 when inactive
  int depth;  // number of labels on the path from the root
};

// ctc_loss_util.h LogSumExp and the per-frame log-softmax run on the operation-exact exp / log pair of ctc_math.h
// (no libm, no hardware exp2 / log2): the decoder's results are then a pure function of the logits' bits, the same
// in this kernel, in the sequential kernel below and in the C oracle the tests compare against bit for bit.
__device__ __forceinline__ float log_sum_exp(float a, float b) { return ctc_log_sum_exp(a, b); }
// wave priority of the register kernels (s_setprio): 3 = ahead of the other batches' persistent GEMM waves (the product since round 3);
// 0 = the round-5 review's "decoder behind the next batch's network" (A/B: tools/variants.sh --product beam CHIRON_BEAM_PRIO 0;
// profiles/r06_beam_priority_ab.txt)
#ifndef CHIRON_BEAM_PRIO
#define CHIRON_BEAM_PRIO 3
#endif
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__global__ __launch_bounds__(64) void beam_kernel(const BeamParams p, int node_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = p.beam;
  // current beam (old probabilities), sorted by descending total
  float* E_tot = reinterpret_cast<float*>(smem);
  float* E_blk = E_tot + W;
  float* E_lab = E_blk + W;
  int* E_node = reinterpret_cast<int*>(E_lab + W);
  int* E_lc = E_node + W;  // label of the entry's node
  // leaves being built for the next frame
  float* L_tot = reinterpret_cast<float*>(E_lc + W);
  float* L_blk = L_tot + W;
  float* L_lab = L_blk + W;
  int* L_node = reinterpret_cast<int*>(L_lab + W);
  int* L_par = L_node + W;   // parent node id (children inserted in this frame), -1 for carried entries
  int* L_lc = L_par + W;     // label of the leaf's node
  int* L_orig = L_lc + W;    // index in E for carried entries, -1 for inserted children
  int* ch_node = L_orig + W;     // [W][4] child node id (-1: not materialised)
  int* ch_orig = ch_node + 4 * W;  // [W][4] index in E of the child if it is in the beam, else -1
  int* alive = ch_orig + 4 * W;  // [W] carried entry still in the leaves
  int* odead = alive + W;        // [W] TF reset this entry's old probability (it will not spawn children)
  __shared__ int s_nodes;

  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int K = p.K, blank = p.K - 1, T = p.T;
  const int len = min(max(p.seq_len[b], 0), T);
  BeamNode* nodes = reinterpret_cast<BeamNode*>(p.workspace) + (long)b * node_cap;
  const float* lg = p.logits + (long)b * T * K;

  if (lane == 0) {
    BeamNode r;
    r.parent = -1;
    r.label = -1;
    r.child[0] = r.child[1] = r.child[2] = r.child[3] = -1;
    r.slot = 0;
    r.depth = 0;
    nodes[0] = r;
    E_tot[0] = 0.f;  // root: newp.total = newp.blank = 0, newp.label = -inf
    E_blk[0] = 0.f;
    E_lab[0] = NEG_INF;
    E_node[0] = 0;
    s_nodes = 1;
  }
  __syncthreads();
  int nb = 1;

  for (int t = 0; t < len; ++t) {
    // log-softmax of the frame (TF normalises inside Step())
    float logp[CHIRON_KMAX];
    ctc_log_softmax<CHIRON_KMAX>(lg + t * K, logp);   // K == CHIRON_KMAX (checked at launch)
    // ---- P1: carried entries
    for (int i = lane; i < nb; i += 64) {
      const int n = E_node[i];
      const BeamNode nd = nodes[n];
      float n_label = E_lab[i];
      if (nd.parent >= 0) {
        const BeamNode pa = nodes[nd.parent];
        if (pa.slot >= 0) {  // parent active: hand over its mass
          const float prev = (nd.label == pa.label) ? E_blk[pa.slot] : E_tot[pa.slot];
          n_label = log_sum_exp(n_label, prev);
        }
        n_label += logp[nd.label];
      }
      const float n_blank = E_tot[i] + logp[blank];
      L_tot[i] = log_sum_exp(n_blank, n_label);
      L_blk[i] = n_blank;
      L_lab[i] = n_label;
      L_node[i] = n;
      L_par[i] = -1;
      L_lc[i] = nd.label;
      L_orig[i] = i;
      E_lc[i] = nd.label;
      alive[i] = 1;
      odead[i] = 0;
      for (int c = 0; c < 4; ++c) {
        const int ch = nd.child[c];
        ch_node[4 * i + c] = ch;
        ch_orig[4 * i + c] = ch >= 0 ? nodes[ch].slot : -1;
      }
    }
    __syncthreads();

    // ---- P2: grow new leaves -- TF's second loop, literally and in order (wave-uniform control flow)
    int nL = nb;
    bool bot_valid = false;
    float bot_val = NEG_INF;
    int bot_idx = 0;
    auto find_bottom = [&]() {
      float v = INFINITY;
      int ix = 1 << 30;
      for (int s = lane; s < nL; s += 64) {
        const float x = L_tot[s];
        if (x < v) {
          v = x;
          ix = s;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(ix, o);
        if (ov < v || (ov == v && oi < ix)) {
          v = ov;
          ix = oi;
        }
      }
      bot_val = v;
      bot_idx = ix;
      bot_valid = true;
    };
    for (int i = 0; i < nb; ++i) {
      if (uni(odead[i])) continue;
      const float ot = uni(E_tot[i]);
      if (!(ot > NEG_INF)) continue;
      if (nL >= W) {  // is_candidate(b->oldp)
        if (!bot_valid) find_bottom();
        // branches come in descending old total and the bottom never decreases: once one fails, all
        // later ones fail too (TF would `continue` through every one of them to the same effect)
        if (!(ot > bot_val)) break;
      }
      const float oblk = uni(E_blk[i]);
      const int lc = uni(E_lc[i]);
      const int pnode = uni(E_node[i]);
      for (int c = 0; c < 4; ++c) {
        const int co = uni(ch_orig[4 * i + c]);
        if (co >= 0 && uni(alive[co])) continue;  // child Active(): already carried
        const float prev = (c == lc) ? oblk : ot;
        const float tot = logp[c] + prev;
        bool cand = tot > NEG_INF;
        if (cand && nL >= W) {
          if (!bot_valid) find_bottom();
          cand = tot > bot_val;
        }
        if (cand) {
          int slot;
          if (nL >= W) {  // beam full: the bottom leaves the search
            slot = bot_idx;
            const int jo = uni(L_orig[slot]);
            if (jo >= 0 && lane == 0) alive[jo] = 0;
          } else {
            slot = nL++;
          }
          if (lane == 0) {
            L_tot[slot] = tot;
            L_blk[slot] = NEG_INF;
            L_lab[slot] = tot;
            L_node[slot] = ch_node[4 * i + c];
            L_par[slot] = pnode;
            L_lc[slot] = c;
            L_orig[slot] = -1;
          }
          bot_valid = false;
          __syncthreads();
        } else if (co >= 0) {
          // TF: c.oldp.Reset() -- if that child is a branch still waiting in this loop it is now dead
          if (lane == 0) odead[co] = 1;
          __syncthreads();
        }
      }
    }
    __syncthreads();

    // ---- P3: next beam = leaves ranked by descending total; trie bookkeeping
    for (int j = lane; j < nb; j += 64)
      if (!alive[j]) nodes[E_node[j]].slot = -1;
    float r_tot[BEAM_MAX / 64], r_blk[BEAM_MAX / 64], r_lab[BEAM_MAX / 64];
    int r_node[BEAM_MAX / 64], r_rank[BEAM_MAX / 64];
#pragma unroll
    for (int q = 0; q < BEAM_MAX / 64; ++q) {
      const int s = lane + 64 * q;
      r_rank[q] = -1;
      if (s < nL) {
        const float ts = L_tot[s];
        int r = 0;
        for (int k = 0; k < nL; ++k) {
          const float tk = L_tot[k];
          r += (tk > ts) || (tk == ts && k < s);
        }
        r_rank[q] = r;
        r_tot[q] = ts;
        r_blk[q] = L_blk[s];
        r_lab[q] = L_lab[s];
        int n = L_node[s];
        const int par = L_par[s];
        if (par >= 0) {
          if (n < 0) {  // materialise the surviving child
            n = atomicAdd(&s_nodes, 1);
            BeamNode nd;
            nd.parent = par;
            nd.label = L_lc[s];
            nd.child[0] = nd.child[1] = nd.child[2] = nd.child[3] = -1;
            nd.slot = r;
            nd.depth = nodes[par].depth + 1;
            nodes[n] = nd;
            nodes[par].child[L_lc[s]] = n;
          } else {
            nodes[n].slot = r;
          }
        } else {
          nodes[n].slot = r;
        }
        r_node[q] = n;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BEAM_MAX / 64; ++q) {
      if (r_rank[q] >= 0) {
        const int r = r_rank[q];
        E_tot[r] = r_tot[q];
        E_blk[r] = r_blk[q];
        E_lab[r] = r_lab[q];
        E_node[r] = r_node[q];
      }
    }
    nb = nL;
    __threadfence_block();
    __syncthreads();
  }

  // ---- TopPaths(1): the leaf with the largest total; labels root -> leaf, no merging
  if (lane == 0) {
    int n = E_node[0];
    float best = E_tot[0];
    for (int i = 1; i < nb; ++i)
      if (E_tot[i] > best) {
        best = E_tot[i];
        n = E_node[i];
      }
    const int depth = nodes[n].depth;
    uint8_t* out = p.labels + (long)b * T;
    while (n > 0) {
      const BeamNode nd = nodes[n];
      out[nd.depth - 1] = (uint8_t)nd.label;
      n = nd.parent;
    }
    p.count[b] = depth;
    p.log_prob[b] = best;
  }
}

// ---------------------------------------------------------------------------------------------------------
// beam <= 64: the whole beam lives in the registers of one wave (lane = beam slot).
//
// The generic kernel above pays an LDS (or HBM) round trip for every scalar the sequential grow walk
// looks at.  Here the walk is event driven instead: in any state (top-N bottom, alive set, reset set) all
// (branch, child) pairs are classified by every lane at once; only pairs that CHANGE the state (an insertion,
// or TF's oldp.Reset() of a branch still waiting) are events, and the first event in TF's visiting order is
// found with one ballot.  Events are applied one at a time with v_readlane / predicated moves, the bottom is
// a DPP min + ballot.  The result is, by construction, the same sequence of state changes as the literal walk
// (tests compare the two kernels bit for bit).
//   Every entry carries the beam SLOT of its parent and of its four children (or "not in the beam") next to their node
//   ids; the references follow the entries through each frame's re-ranking (old slot -> new rank, a 64-byte table), new
//   children report their rank to the branch that spawned them, and a node that re-enters the beam finds its relatives by
//   comparing node ids across the wave (rare).  So "is my parent in the beam? which of my children are?" costs no look-up
//   at all, and the kernel's LDS is 4.4 KB whatever beam * T is -- it used to keep a node id -> slot byte table of
//   1 + beam * T entries (12 .. 20 KB), and with the other batches' persistent GEMM workgroups holding 120 .. 133 KB of a
//   CU's LDS that footprint decided how many windows a CU takes and how long the next GEMM launch waits for its CUs:
//   measured with three batches in flight, every extra 16 KB per window cost 0.7 ms per 1100-window batch.
//   The trie in HBM is write-mostly (read back only for a re-inserted node's child ids and for the final back-trace).
//   Log-softmax of 64 frames at a time is held lane-distributed in registers and broadcast by v_readlane.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rlf(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int rli(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp_f<0xB1>(v));   // quad_perm [1,0,3,2]
  v = fminf(v, dpp_f<0x4E>(v));   // quad_perm [2,3,0,1]
  v = fminf(v, dpp_f<0x141>(v));  // row_half_mirror
  v = fminf(v, dpp_f<0x140>(v));  // row_mirror: every lane of a 16-lane row holds the row minimum
  // (the same reduction with v_permlane16_swap + v_permlane32_swap instead of the four v_readlane was measured in round 5: 2.84 against
  //  2.75 ms alone at W = 50, the same in the mix -- the wave-uniform result in an SGPR is worth more than the two instructions)
  return fminf(fminf(rlf(v, 0), rlf(v, 16)), fminf(rlf(v, 32), rlf(v, 48)));
}

constexpr int B64_FIELDS = 11;

// One wave per workgroup: LDS operations of the wave execute in order, so "synchronising" only has to stop the compiler
// from reordering around it and drain lgkmcnt.  __syncthreads() would also wait for every outstanding global store
// (s_waitcnt vmcnt(0)) -- the trie writes of each frame, a few thousand cycles of HBM write latency per frame.
__device__ __forceinline__ void lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Rank of a leaf = number of leaves that sort before it (larger total, or equal total and lower slot).  The totals of the live
// quads lie in the LDS (entries past the leaves hold -inf and count for nobody).  The read of quad q + 1 is issued before quad q is
// compared: the plain loop paid one LDS round trip per iteration (8 x ~100 cycles per frame at W = 30).  (All reads up front in a
// register array is what one would write first; it took the kernels from 81 / 95 to 216 / 140 registers, and between the other
// batches' GEMM waves the decoder's register footprint decides where its waves fit: +0.6 ms per batch in the mix.)
__device__ __forceinline__ int rank_in_block(const int* totals, int nq, float mine, int slot) {
  int r = 0;
  int4 nx = *reinterpret_cast<const int4*>(totals);
  for (int q = 0; q < nq; ++q) {
    const int4 t4 = nx;
    if (q + 1 < nq) nx = *reinterpret_cast<const int4*>(totals + 4 * (q + 1));   // nq is wave-uniform
    const float tk0 = __int_as_float(t4.x), tk1 = __int_as_float(t4.y), tk2 = __int_as_float(t4.z), tk3 = __int_as_float(t4.w);
    const int k0 = 4 * q;
    r += (tk0 > mine) || (tk0 == mine && k0 < slot);
    r += (tk1 > mine) || (tk1 == mine && k0 + 1 < slot);
    r += (tk2 > mine) || (tk2 == mine && k0 + 2 < slot);
    r += (tk3 > mine) || (tk3 == mine && k0 + 3 < slot);
  }
  return r;
}

__global__ __launch_bounds__(64) void beam64_kernel(const BeamParams p, int node_cap) {
  __shared__ __attribute__((aligned(16))) int scratch[B64_FIELDS * 64];   // permutation buffer: [field][rank]
  __shared__ __attribute__((aligned(16))) int chupd[64 * 4];              // [branch slot][label] node id of the child materialised this frame
  __shared__ unsigned csu[64];          // [branch slot] four bytes: rank + 1 of the child spawned this frame under that label (0: none)
  __shared__ unsigned csr[64];          // [slot of a re-inserted node] the same for its children that are in the beam
  __shared__ unsigned char rmap[64];    // old slot -> rank + 1 of the carried entry if it is still a leaf (0: it left the beam)
  // One latency-bound wave per window next to the other batches' GEMM waves: with equal priority a saturated MFMA wave
  // leaves a VALU wave of its SIMD about one issue slot per MFMA (tools/ubench/valu_exec_mask.hip).  Measured same-box
  // with three batches in flight: 11.81 / 11.83 ms per beam-30 batch against 11.87 / 11.95.
  __builtin_amdgcn_s_setprio(CHIRON_BEAM_PRIO);

  const int W = p.beam;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int K = p.K, T = p.T;
  const int len = min(max(p.seq_len[b], 0), T);
  BeamNode* nodes = reinterpret_cast<BeamNode*>(p.workspace) + (long)b * node_cap;
  const float* lg = p.logits + (long)b * T * K;

  // beam entry of this lane (valid for lane < nb), ranked by descending total
  float e_tot = 0.f, e_blk = 0.f, e_lab = NEG_INF;
  int e_node = 0, e_par = -1, e_lc = -1, e_depth = 0;
  int e_ch[4] = {-1, -1, -1, -1};
  int e_ps = -1;                       // beam slot of the parent, -1: not in the beam
  int e_cs[4] = {-1, -1, -1, -1};      // beam slots of the children, -1: not in the beam (or not in the trie)
  if (lane == 0) {
    BeamNode r;
    r.parent = -1;
    r.label = -1;
    r.child[0] = r.child[1] = r.child[2] = r.child[3] = -1;
    r.slot = 0;
    r.depth = 0;
    nodes[0] = r;
  }
  int nb = 1;
  int n_nodes = 1;

  for (int t0 = 0; t0 < len; t0 += 64) {
    // ---- log-softmax of frames t0 .. t0+63, one frame per lane (TF normalises inside Step())
    float lpk[CHIRON_KMAX];
    {
      const int t = min(t0 + lane, len - 1);
      float x[CHIRON_KMAX];
#pragma unroll
      for (int k = 0; k < CHIRON_KMAX; ++k) x[k] = lg[t * K + k];   // K == CHIRON_KMAX (checked at launch)
      ctc_log_softmax<CHIRON_KMAX>(x, lpk);
    }
    const int tn = min(64, len - t0);
    for (int tt = 0; tt < tn; ++tt) {
      float logp[CHIRON_KMAX];
#pragma unroll
      for (int k = 0; k < CHIRON_KMAX; ++k) logp[k] = rlf(lpk[k], tt);
      const float lp_blank = logp[CHIRON_KMAX - 1];  // K == 5 (checked at launch)

      // ---- P1: every entry extends itself; leaf registers start as the carried entries
      const bool inb = lane < nb;
      float l_tot = INFINITY, l_blk = NEG_INF, l_lab = NEG_INF;
      int chs[4] = {-1, -1, -1, -1};
      float cand[4];
      {
        const int lc = max(e_lc, 0);
        const float lp_lc = lc == 0 ? logp[0] : lc == 1 ? logp[1] : lc == 2 ? logp[2] : logp[3];
        const int pslot = (inb && e_ps >= 0) ? e_ps : 255;
        const int src = pslot & 63;
        const float p_tot = __shfl(e_tot, src), p_blk = __shfl(e_blk, src);
        const int p_lc = __shfl(e_lc, src);
        float n_label = e_lab;
        if (e_par >= 0) {
          if (pslot != 255) n_label = log_sum_exp(n_label, (e_lc == p_lc) ? p_blk : p_tot);
          n_label += lp_lc;
        }
        const float n_blank = e_tot + lp_blank;
        if (inb) {
          l_tot = log_sum_exp(n_blank, n_label);
          l_blk = n_blank;
          l_lab = n_label;
#pragma unroll
          for (int c = 0; c < 4; ++c) chs[c] = e_cs[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) cand[c] = logp[c] + (c == e_lc ? e_blk : e_tot);
      }
      int l_node = e_node, l_par = -1, l_lc = e_lc, l_depth = e_depth, l_orig = inb ? lane : -1, l_pi = -1;

      // ---- P2: grow new leaves, event by event, in TF's visiting order (branch-major, label-minor).
      // Per lane (= branch) four bit sets over its children c = 0..3 carry the walk's state, so one iteration costs a few
      // dozen instructions and an event updates only what it touches:
      //   pend  pairs (lane, c) the walk has not passed yet          act    child c is an alive carried entry
      //   cdead TF reset child c's old probability already           passed this lane has been entered as a branch
      //   dead  TF reset THIS lane's old probability (as somebody's child) before the walk reached it
      int nL = nb;
      float botv = NEG_INF;
      int boti = 0;
      bool full = nL >= W;
      if (full) {
        botv = wave_min(lane < nL ? l_tot : INFINITY);
        boti = __builtin_ctzll(__ballot(lane < nL && l_tot == botv));
      }
      const bool has_old = inb && e_tot > NEG_INF;
      unsigned pend = has_old ? 0xFu : 0u, act = 0u, cdead = 0u, cfin = 0u, cafter = 0u;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (chs[c] >= 0) act |= 1u << c;         // every carried entry is alive at the start of the frame
        if (cand[c] > NEG_INF) cfin |= 1u << c;   // finite candidate
        if (chs[c] > lane) cafter |= 1u << c;     // the child is a branch the walk reaches later
      }
      bool passed = false, dead = false;
      bool any_event = false;
      while (true) {
        const bool br = passed || (!dead && (!full || e_tot > botv));
        unsigned cnd = cfin;                      // candidates that would enter the leaves now
        if (full) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (!(cand[c] > botv)) cnd &= ~(1u << c);
        }
        // a rejected child that sits in the old beam loses its old probability; that only matters if it is a branch the
        // walk has not reached yet
        const unsigned rst = ~cnd & cafter & ~cdead;
        const unsigned ev = br ? (pend & ~act & (cnd | rst)) : 0u;
        const unsigned long long evm = __ballot(ev != 0u);
        if (evm == 0ull) break;
        any_event = true;
        const int evc = __builtin_ctz(ev | 16u);
        const int i = __builtin_ctzll(evm);
        const int c = rli(evc, i);
        const bool ins = (rli((int)cnd, i) >> c) & 1;
        if (ins) {
          // Only the new leaf's TOTAL is needed inside the walk (it decides the next bottom).  Its node id, parent node and depth
          // are facts about the branch (i, c) that spawned it and are fetched ONCE per frame, after the walk, by the lane that
          // still owns the leaf then (l_pi, l_lc): three broadcasts per event used to carry them, although a slot may be
          // overwritten again within the frame.
          const float sel_tot = c == 0 ? cand[0] : c == 1 ? cand[1] : c == 2 ? cand[2] : cand[3];   // c is wave-uniform
          const float tot = rlf(sel_tot, i);
          int slot;
          if (full) {  // the bottom leaves the search
            slot = boti;
            const int jo = rli(l_orig, slot);
            if (jo >= 0) {  // a carried entry: whoever has it as a child sees it inactive from now on
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (chs[k] == jo) act &= ~(1u << k);
            }
          } else {
            slot = nL++;
          }
          if (lane == slot) {
            l_tot = tot;
            l_blk = NEG_INF;
            l_lab = tot;
            l_lc = c;
            l_orig = -1;
            l_pi = i;
          }
          full = nL >= W;
          if (full) {
            botv = wave_min(lane < nL ? l_tot : INFINITY);
            boti = __builtin_ctzll(__ballot(lane < nL && l_tot == botv));
          }
        } else {
          const int sel_co = c == 0 ? chs[0] : c == 1 ? chs[1] : c == 2 ? chs[2] : chs[3];
          const int co = rli(sel_co, i);
          if (lane == co) dead = true;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (chs[k] == co) cdead |= 1u << k;
        }
        // the walk is now past (i, c)
        if (lane < i) pend = 0u;
        if (lane == i) {
          pend &= ~((2u << c) - 1u);
          passed = true;
        }
      }

      // ---- a QUIET frame: the walk had no event (no leaf inserted, no old probability reset) and the carried entries are still in
      //      descending order (equal totals keep their slots: the ranking below breaks ties by slot).  Then the ranking is the
      //      identity, nobody's relatives move, no node is created: the next beam is this one with its three probabilities updated.
      //      On a trained model's blank-dominated posteriors that is most frames (84 % quiet, 61 % also in order at W = 30); on flat
      //      posteriors it costs one neighbour shuffle and a ballot per frame.
      if (!any_event) {
        const float nxt = __shfl(l_tot, (lane + 1) & 63);
        if (__ballot(lane + 1 < nL && !(l_tot >= nxt)) == 0ull) {
          if (lane < nb) e_tot = l_tot, e_blk = l_blk, e_lab = l_lab;
          continue;
        }
      }
      // ---- P3: trie bookkeeping, rank the leaves (descending total), permute into rank order
      *reinterpret_cast<int4*>(chupd + 4 * lane) = make_int4(-1, -1, -1, -1);
      csu[lane] = 0u;
      csr[lane] = 0u;
      const bool isleaf = lane < nL;
      {   // the inserted leaves fetch what their branch knows about them (see the walk): six shuffles per frame
        const bool got = isleaf && l_pi >= 0;
        const int src = got ? l_pi : lane;
        const int g_node = __shfl(e_node, src), g_depth = __shfl(e_depth, src);
        const int g0 = __shfl(e_ch[0], src), g1 = __shfl(e_ch[1], src), g2 = __shfl(e_ch[2], src), g3 = __shfl(e_ch[3], src);
        if (got) {
          l_par = g_node;
          l_depth = g_depth + 1;
          l_node = l_lc == 0 ? g0 : l_lc == 1 ? g1 : l_lc == 2 ? g2 : g3;
        }
      }
      const bool inserted = isleaf && l_par >= 0;     // otherwise a leaf is this lane's carried entry (an evicted one's slot holds an inserted child)
      const bool fresh = inserted && l_node < 0;
      const bool reins = inserted && !fresh;
      const unsigned long long fm = __ballot(fresh);
      int f_ch[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) f_ch[c] = e_ch[c];
      int f_par = e_par;
      if (inserted) f_par = l_par;
      if (fresh) {
        l_node = n_nodes + __popcll(fm & ((1ull << lane) - 1ull));
        BeamNode nd;
        nd.parent = l_par;
        nd.label = l_lc;
        nd.child[0] = nd.child[1] = nd.child[2] = nd.child[3] = -1;
        nd.slot = 0;
        nd.depth = l_depth;
        nodes[l_node] = nd;
        nodes[l_par].child[l_lc] = l_node;
        f_ch[0] = f_ch[1] = f_ch[2] = f_ch[3] = -1;
      }
      n_nodes += __popcll(fm);
      // Only a re-inserted node reads the trie back (its children may have been linked by the stores just above), and
      // only then the wave waits for its own global stores.
      const unsigned long long reins_m = __ballot(reins);
      if (reins_m) __threadfence_block();
      lds_sync();
      if (reins) {
        // a node that was in the beam before (possibly evicted earlier in this very frame, after it had
        // spawned children -- hence after the stores above): its children keep their identity
        const int4 ch = *reinterpret_cast<const int4*>(nodes[l_node].child);
        f_ch[0] = ch.x;
        f_ch[1] = ch.y;
        f_ch[2] = ch.z;
        f_ch[3] = ch.w;
      }
      if (inserted) chupd[4 * l_pi + l_lc] = l_node;
      lds_sync();
      if (isleaf && !inserted) {
        const int4 u = *reinterpret_cast<const int4*>(chupd + 4 * lane);
        f_ch[0] = u.x >= 0 ? u.x : f_ch[0];
        f_ch[1] = u.y >= 0 ? u.y : f_ch[1];
        f_ch[2] = u.z >= 0 ? u.z : f_ch[2];
        f_ch[3] = u.w >= 0 ? u.w : f_ch[3];
      }
      // rank = number of leaves that sort before this one (larger total, or equal total and lower slot).  The totals go
      // through the LDS once (field 0 of the permutation buffer, about to be overwritten anyway) and come back four per
      // broadcast read: a v_readlane per leaf cost 93 cycles each, 2800 of the 4200 cycles P3 took per frame at W = 30.
      scratch[lane] = __float_as_int(lane < nL ? l_tot : NEG_INF);
      lds_sync();
      const int r = rank_in_block(scratch, (nL + 3) >> 2, l_tot, lane);
      // Slot references into the next beam (all encoded + 1, 0 = not in the beam).  A carried entry that is still a leaf
      // publishes its rank under its old slot; whoever referred to that slot follows it, whoever referred to a slot whose
      // entry left the beam reads 0.  A surviving new child reports its rank to the branch that spawned it.
      rmap[lane] = (isleaf && !inserted) ? (unsigned char)(r + 1) : (unsigned char)0;
      if (inserted) reinterpret_cast<unsigned char*>(csu)[4 * l_pi + l_lc] = (unsigned char)(r + 1);
      lds_sync();
      unsigned n_ps = 0u, n_cs = 0u;
      if (isleaf && !inserted) {
        if (e_ps >= 0) n_ps = rmap[e_ps];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (e_cs[c] >= 0) n_cs |= (unsigned)rmap[e_cs[c]] << (8 * c);
        const unsigned u = csu[lane];     // a label's old child (if any) has left the beam when a new one could be spawned under it
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (u & (0xFFu << (8 * c))) n_cs = (n_cs & ~(0xFFu << (8 * c))) | (u & (0xFFu << (8 * c)));
      } else if (inserted) {
        n_ps = rmap[l_pi];                // the branch that spawned it, if that is still a leaf
      }
      // A node that was in the beam before and comes back (possibly within this very frame) may have relatives here that
      // lost sight of it: every leaf whose parent is that node points to it again, and tells it where it is.
      if (reins_m) {
        unsigned long long rm = reins_m;
        while (rm) {
          const int i = __builtin_ctzll(rm);
          rm &= rm - 1ull;
          const int n = rli(l_node, i), ri = rli(r, i);
          if (isleaf && f_par == n) {
            n_ps = (unsigned)(ri + 1);
            reinterpret_cast<unsigned char*>(csr)[4 * i + l_lc] = (unsigned char)(r + 1);
          }
        }
        lds_sync();
        if (reins) n_cs = csr[lane];
      }
      if (isleaf) {
        scratch[0 * 64 + r] = __float_as_int(l_tot);
        scratch[1 * 64 + r] = __float_as_int(l_blk);
        scratch[2 * 64 + r] = __float_as_int(l_lab);
        scratch[3 * 64 + r] = l_node;
        scratch[4 * 64 + r] = f_par;
        scratch[5 * 64 + r] = (l_lc + 1) | (l_depth << 3) | (int)(n_ps << 16);   // label + 1 (3 bits), depth (13 bits: T < 8192 at launch), parent slot + 1
        scratch[6 * 64 + r] = (int)n_cs;                                          // child slots + 1, one byte per label
        scratch[7 * 64 + r] = f_ch[0];
        scratch[8 * 64 + r] = f_ch[1];
        scratch[9 * 64 + r] = f_ch[2];
        scratch[10 * 64 + r] = f_ch[3];
      }
      lds_sync();
      nb = nL;
      if (lane < nb) {
        e_tot = __int_as_float(scratch[0 * 64 + lane]);
        e_blk = __int_as_float(scratch[1 * 64 + lane]);
        e_lab = __int_as_float(scratch[2 * 64 + lane]);
        e_node = scratch[3 * 64 + lane];
        e_par = scratch[4 * 64 + lane];
        const int pa = scratch[5 * 64 + lane];
        const unsigned pb = (unsigned)scratch[6 * 64 + lane];
        e_lc = (pa & 7) - 1;
        e_depth = (pa >> 3) & 0x1FFF;
        e_ps = (pa >> 16) - 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) e_cs[c] = (int)((pb >> (8 * c)) & 0xFFu) - 1;
        e_ch[0] = scratch[7 * 64 + lane];
        e_ch[1] = scratch[8 * 64 + lane];
        e_ch[2] = scratch[9 * 64 + lane];
        e_ch[3] = scratch[10 * 64 + lane];
      }
      lds_sync();
    }
  }

  // ---- TopPaths(1): rank 0 is the leaf with the largest total; labels root -> leaf, no merging
  __threadfence_block();  // the back-trace reads the trie this wave wrote
  if (lane == 0) {
    int n = e_node;
    uint8_t* out = p.labels + (long)b * T;
    p.count[b] = e_depth;
    p.log_prob[b] = e_tot;
    while (n > 0) {
      const BeamNode nd = nodes[n];
      out[nd.depth - 1] = (uint8_t)nd.label;
      n = nd.parent;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// beam <= 32: TWO windows per wave (lanes 0..31 / 32..63, lane & 31 = beam slot).
//
// The hardware does not skip the empty 16-lane passes of a partly masked wave (tools/ubench/valu_exec_mask.hip), so a
// 30-wide beam in a 64-wide wave pays for the idle half -- and next to the other batches' GEMM waves what the decoder costs
// is its issue slots.  Here the lane-parallel phases (P1, P3) serve both windows with the same instructions; the event
// walk (P2) advances both windows in lockstep, one event of each per iteration, with everything that was wave-uniform in
// beam64_kernel (counts, the top-N bottom, the event's branch and label) uniform per HALF instead: held in VGPRs, broadcast
// inside a half through the LDS crossbar (ds_bpermute) instead of v_readlane.  Same sequence of state changes per window,
// so the same results bit for bit (tests compare the two kernels and the oracle).  A window that is shorter than its
// neighbour idles through the extra frames with its entries carried unchanged (re-ranking equal totals is the identity).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_min(float v) {   // minimum over the 32 lanes of this lane's half, in every lane
  v = fminf(v, dpp_f<0xB1>(v));
  v = fminf(v, dpp_f<0x4E>(v));
  v = fminf(v, dpp_f<0x141>(v));
  v = fminf(v, dpp_f<0x140>(v));   // every lane of a 16-lane row holds its row's minimum
  // the neighbouring row's value without the LDS crossbar (round 5: was __shfl_xor(v, 16), a ds_bpermute round trip in every bottom()):
  // v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second -- with both operands v the results
  // are (row0, row0, row2, row2) and (row1, row1, row3, row3)
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fminf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}

__global__ __launch_bounds__(64) void beam32x2_kernel(const BeamParams p, int node_cap) {
  __shared__ __attribute__((aligned(16))) int scratch[B64_FIELDS * 64];
  __shared__ __attribute__((aligned(16))) int chupd[64 * 4];
  __shared__ unsigned csu[64];
  __shared__ unsigned csr[64];
  __shared__ unsigned char rmap[64];
  __builtin_amdgcn_s_setprio(CHIRON_BEAM_PRIO);

  const int W = p.beam;
  const int lane = threadIdx.x;
  const int hl = lane & 31, hb = lane & 32;
  const bool h = lane >= 32;
  const int b = 2 * blockIdx.x + (h ? 1 : 0);
  const bool valid = b < p.B;                      // odd batch: the last wave's upper half has no window
  const int bc = valid ? b : p.B - 1;
  const int K = p.K, T = p.T;
  const int len = valid ? min(max(p.seq_len[bc], 0), T) : 0;
  const int lenmax = max(rli(len, 0), rli(len, 32));
  BeamNode* nodes = reinterpret_cast<BeamNode*>(p.workspace) + (long)bc * node_cap;
  const float* lg = p.logits + (long)bc * T * K;
  auto pick = [&](unsigned lo, unsigned hi) -> unsigned { return h ? hi : lo; };   // a per-half scalar into the lanes of its half

  float e_tot = 0.f, e_blk = 0.f, e_lab = NEG_INF;
  int e_node = 0, e_par = -1, e_lc = -1, e_depth = 0;
  int e_ch[4] = {-1, -1, -1, -1};
  int e_ps = -1;
  int e_cs[4] = {-1, -1, -1, -1};
  if (hl == 0 && valid) {
    BeamNode r;
    r.parent = -1;
    r.label = -1;
    r.child[0] = r.child[1] = r.child[2] = r.child[3] = -1;
    r.slot = 0;
    r.depth = 0;
    nodes[0] = r;
  }
  int nb = 1;        // per half
  int n_nodes = 1;   // per half

  for (int t0 = 0; t0 < lenmax; t0 += 32) {
    float lpk[CHIRON_KMAX];
    {
      const int t = max(min(t0 + hl, len - 1), 0);
      float x[CHIRON_KMAX];
#pragma unroll
      for (int k = 0; k < CHIRON_KMAX; ++k) x[k] = lg[t * K + k];
      ctc_log_softmax<CHIRON_KMAX>(x, lpk);
    }
    const int tn = min(32, lenmax - t0);
    for (int tt = 0; tt < tn; ++tt) {
      const bool live = t0 + tt < len;
      float logp[CHIRON_KMAX];
#pragma unroll
      for (int k = 0; k < CHIRON_KMAX; ++k) logp[k] = __shfl(lpk[k], hb + tt);
      const float lp_blank = logp[CHIRON_KMAX - 1];

      // ---- P1
      const bool have = hl < nb;
      const bool inb = live && have;
      float l_tot = INFINITY, l_blk = NEG_INF, l_lab = NEG_INF;
      int chs[4] = {-1, -1, -1, -1};
      float cand[4];
      {
        const int lc = max(e_lc, 0);
        const float lp_lc = lc == 0 ? logp[0] : lc == 1 ? logp[1] : lc == 2 ? logp[2] : logp[3];
        const int pslot = (inb && e_ps >= 0) ? e_ps : 255;
        const int src = hb + (pslot & 31);
        const float p_tot = __shfl(e_tot, src), p_blk = __shfl(e_blk, src);
        const int p_lc = __shfl(e_lc, src);
        float n_label = e_lab;
        if (e_par >= 0) {
          if (pslot != 255) n_label = log_sum_exp(n_label, (e_lc == p_lc) ? p_blk : p_tot);
          n_label += lp_lc;
        }
        const float n_blank = e_tot + lp_blank;
        if (inb) {
          l_tot = log_sum_exp(n_blank, n_label);
          l_blk = n_blank;
          l_lab = n_label;
#pragma unroll
          for (int c = 0; c < 4; ++c) chs[c] = e_cs[c];
        } else if (have) {   // a window past its end: the entry is carried as it is
          l_tot = e_tot;
          l_blk = e_blk;
          l_lab = e_lab;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) cand[c] = logp[c] + (c == e_lc ? e_blk : e_tot);
      }
      int l_node = e_node, l_par = -1, l_lc = e_lc, l_depth = e_depth, l_orig = inb ? hl : -1, l_pi = -1;

      // ---- P2 (see beam64_kernel): nL, full, botv, boti are uniform per half
      int nL = nb;
      float botv = NEG_INF;
      int boti = 0;
      bool full = nL >= W;
      auto bottom = [&](bool want) {   // the halves that `want` take their new bottom
        const float v = half_min(hl < nL ? l_tot : INFINITY);
        const unsigned long long m = __ballot(hl < nL && l_tot == v);
        const int bi = __builtin_ctz(pick((unsigned)m, (unsigned)(m >> 32)) | 0x80000000u);
        if (want) {
          botv = v;
          boti = bi;
        }
      };
      if (__ballot(full)) bottom(full);
      const bool has_old = inb && e_tot > NEG_INF;
      unsigned pend = has_old ? 0xFu : 0u, act = 0u, cdead = 0u, cfin = 0u, cafter = 0u;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (chs[c] >= 0) act |= 1u << c;
        if (cand[c] > NEG_INF) cfin |= 1u << c;
        if (chs[c] > hl) cafter |= 1u << c;
      }
      bool passed = false, dead = false;
      bool any_event = false;   // of either window
      while (true) {
        const bool br = passed || (!dead && (!full || e_tot > botv));
        unsigned cnd = cfin;
        if (full) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (!(cand[c] > botv)) cnd &= ~(1u << c);
        }
        const unsigned rst = ~cnd & cafter & ~cdead;
        const unsigned ev = br ? (pend & ~act & (cnd | rst)) : 0u;
        const unsigned long long evm = __ballot(ev != 0u);
        if (evm == 0ull) break;
        any_event = true;
        const unsigned evh = pick((unsigned)evm, (unsigned)(evm >> 32));
        const bool hev = evh != 0u;                                  // this half has an event in this iteration
        const int i = __builtin_ctz(evh | 0x80000000u);              // its branch (slot in the half)
        const int src = hb + i;
        // every lane prepares what it would report as the event's branch -- its first pending label, whether that is an insertion, the
        // candidate's total and the child's slot -- BEFORE the broadcast: three independent ds_bpermute instead of two dependent rounds
        // (label first, then the values selected with it) plus a third for the candidate set (round 5)
        const int c_own = __builtin_ctz(ev | 16u) & 3;
        const int pk_own = c_own | ((int)((cnd >> c_own) & 1u) << 2);
        const float tot_own = c_own == 0 ? cand[0] : c_own == 1 ? cand[1] : c_own == 2 ? cand[2] : cand[3];
        const int co_own = c_own == 0 ? chs[0] : c_own == 1 ? chs[1] : c_own == 2 ? chs[2] : chs[3];
        const int pk = __shfl(pk_own, src);
        const float tot = __shfl(tot_own, src);     // node id, parent node and depth of a new leaf: once per frame, after the walk (beam64_kernel)
        const int co = __shfl(co_own, src);
        const int c = pk & 3;
        const bool ins = hev && ((pk >> 2) & 1);
        const bool rs = hev && !ins;
        // insertion: the bottom leaves the search (full) or the leaves grow by one
        const int slot = full ? boti : nL;
        const int jo = __shfl(l_orig, hb + (slot & 31));
        if (ins && full && jo >= 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (chs[k] == jo) act &= ~(1u << k);
        }
        if (ins && !full) nL += 1;
        if (ins && hl == slot) {
          l_tot = tot;
          l_blk = NEG_INF;
          l_lab = tot;
          l_lc = c;
          l_orig = -1;
          l_pi = i;
        }
        if (ins) full = nL >= W;
        if (__ballot(ins && full)) bottom(ins && full);
        // rejection of a child that is a branch still waiting: TF resets its old probability
        if (rs) {
          if (hl == co) dead = true;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (chs[k] == co) cdead |= 1u << k;
        }
        // the walk is now past (i, c)
        if (hev) {
          if (hl < i) pend = 0u;
          if (hl == i) {
            pend &= ~((2u << c) - 1u);
            passed = true;
          }
        }
      }

      // ---- a quiet frame of BOTH windows (see beam64_kernel): no event, carried entries still in order -> only the probabilities move
      if (!any_event) {
        const float nxt = __shfl(l_tot, hb + ((hl + 1) & 31));
        if (__ballot(hl + 1 < nL && !(l_tot >= nxt)) == 0ull) {
          if (hl < nb) e_tot = l_tot, e_blk = l_blk, e_lab = l_lab;
          continue;
        }
      }
      // ---- P3
      *reinterpret_cast<int4*>(chupd + 4 * lane) = make_int4(-1, -1, -1, -1);
      csu[lane] = 0u;
      csr[lane] = 0u;
      const bool isleaf = hl < nL;
      {
        const bool got = isleaf && l_pi >= 0;
        const int src = got ? hb + l_pi : lane;
        const int g_node = __shfl(e_node, src), g_depth = __shfl(e_depth, src);
        const int g0 = __shfl(e_ch[0], src), g1 = __shfl(e_ch[1], src), g2 = __shfl(e_ch[2], src), g3 = __shfl(e_ch[3], src);
        if (got) {
          l_par = g_node;
          l_depth = g_depth + 1;
          l_node = l_lc == 0 ? g0 : l_lc == 1 ? g1 : l_lc == 2 ? g2 : g3;
        }
      }
      const bool inserted = isleaf && l_par >= 0;
      const bool fresh = inserted && l_node < 0;
      const bool reins = inserted && !fresh;
      const unsigned long long fm = __ballot(fresh);
      const unsigned fmh = pick((unsigned)fm, (unsigned)(fm >> 32));
      int f_ch[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) f_ch[c] = e_ch[c];
      int f_par = e_par;
      if (inserted) f_par = l_par;
      if (fresh) {
        l_node = n_nodes + __popc(fmh & ((1u << hl) - 1u));
        BeamNode nd;
        nd.parent = l_par;
        nd.label = l_lc;
        nd.child[0] = nd.child[1] = nd.child[2] = nd.child[3] = -1;
        nd.slot = 0;
        nd.depth = l_depth;
        nodes[l_node] = nd;
        nodes[l_par].child[l_lc] = l_node;
        f_ch[0] = f_ch[1] = f_ch[2] = f_ch[3] = -1;
      }
      n_nodes += __popc(fmh);
      const unsigned long long reins_m = __ballot(reins);
      if (reins_m) __threadfence_block();
      lds_sync();
      if (reins) {
        const int4 ch = *reinterpret_cast<const int4*>(nodes[l_node].child);
        f_ch[0] = ch.x;
        f_ch[1] = ch.y;
        f_ch[2] = ch.z;
        f_ch[3] = ch.w;
      }
      if (inserted) chupd[4 * (hb + l_pi) + l_lc] = l_node;
      lds_sync();
      if (isleaf && !inserted) {
        const int4 u = *reinterpret_cast<const int4*>(chupd + 4 * lane);
        f_ch[0] = u.x >= 0 ? u.x : f_ch[0];
        f_ch[1] = u.y >= 0 ? u.y : f_ch[1];
        f_ch[2] = u.z >= 0 ? u.z : f_ch[2];
        f_ch[3] = u.w >= 0 ? u.w : f_ch[3];
      }
      scratch[lane] = __float_as_int(isleaf ? l_tot : NEG_INF);
      lds_sync();
      const int r = rank_in_block(scratch + hb, (max(rli(nL, 0), rli(nL, 32)) + 3) >> 2, l_tot, hl);   // a half's entries past its own leaves hold -inf
      rmap[lane] = (isleaf && !inserted) ? (unsigned char)(r + 1) : (unsigned char)0;
      if (inserted) reinterpret_cast<unsigned char*>(csu)[4 * (hb + l_pi) + l_lc] = (unsigned char)(r + 1);
      lds_sync();
      unsigned n_ps = 0u, n_cs = 0u;
      if (isleaf && !inserted) {
        if (e_ps >= 0) n_ps = rmap[hb + e_ps];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (e_cs[c] >= 0) n_cs |= (unsigned)rmap[hb + e_cs[c]] << (8 * c);
        const unsigned u = csu[lane];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (u & (0xFFu << (8 * c))) n_cs = (n_cs & ~(0xFFu << (8 * c))) | (u & (0xFFu << (8 * c)));
      } else if (inserted) {
        n_ps = rmap[hb + l_pi];
      }
      if (reins_m) {
        unsigned long long rm = reins_m;
        while (rm) {
          const int i = __builtin_ctzll(rm);   // a lane of the wave: only its own half can hold its relatives
          rm &= rm - 1ull;
          const int n = rli(l_node, i), ri = rli(r, i);
          if ((lane >> 5) == (i >> 5) && isleaf && f_par == n) {
            n_ps = (unsigned)(ri + 1);
            reinterpret_cast<unsigned char*>(csr)[4 * i + l_lc] = (unsigned char)(r + 1);
          }
        }
        lds_sync();
        if (reins) n_cs = csr[lane];
      }
      if (isleaf) {
        const int o = hb + r;
        scratch[0 * 64 + o] = __float_as_int(l_tot);
        scratch[1 * 64 + o] = __float_as_int(l_blk);
        scratch[2 * 64 + o] = __float_as_int(l_lab);
        scratch[3 * 64 + o] = l_node;
        scratch[4 * 64 + o] = f_par;
        scratch[5 * 64 + o] = (l_lc + 1) | (l_depth << 3) | (int)(n_ps << 16);
        scratch[6 * 64 + o] = (int)n_cs;
        scratch[7 * 64 + o] = f_ch[0];
        scratch[8 * 64 + o] = f_ch[1];
        scratch[9 * 64 + o] = f_ch[2];
        scratch[10 * 64 + o] = f_ch[3];
      }
      lds_sync();
      nb = nL;
      if (hl < nb) {
        e_tot = __int_as_float(scratch[0 * 64 + lane]);
        e_blk = __int_as_float(scratch[1 * 64 + lane]);
        e_lab = __int_as_float(scratch[2 * 64 + lane]);
        e_node = scratch[3 * 64 + lane];
        e_par = scratch[4 * 64 + lane];
        const int pa = scratch[5 * 64 + lane];
        const unsigned pb = (unsigned)scratch[6 * 64 + lane];
        e_lc = (pa & 7) - 1;
        e_depth = (pa >> 3) & 0x1FFF;
        e_ps = (pa >> 16) - 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) e_cs[c] = (int)((pb >> (8 * c)) & 0xFFu) - 1;
        e_ch[0] = scratch[7 * 64 + lane];
        e_ch[1] = scratch[8 * 64 + lane];
        e_ch[2] = scratch[9 * 64 + lane];
        e_ch[3] = scratch[10 * 64 + lane];
      }
      lds_sync();
    }
  }

  __threadfence_block();
  if (hl == 0 && valid) {
    int n = e_node;
    uint8_t* out = p.labels + (long)b * T;
    p.count[b] = e_depth;
    p.log_prob[b] = e_tot;
    while (n > 0) {
      const BeamNode nd = nodes[n];
      out[nd.depth - 1] = (uint8_t)nd.label;
      n = nd.parent;
    }
  }
}

static size_t beam_smem_bytes(int W) { return (size_t)W * 4 * (5 + 7 + 8 + 2); }

size_t beam_workspace_bytes(int B, int T, int beam) {
  return (size_t)B * (size_t)(1 + (size_t)beam * T) * sizeof(BeamNode);
}

int launch_beam(const BeamParams& p, hipStream_t stream) {
  if (p.beam < 1 || p.beam > BEAM_MAX || p.K != 5) return -1;
  const int node_cap = 1 + p.beam * p.T;
  if ((size_t)p.B * node_cap * sizeof(BeamNode) > p.workspace_bytes) return -2;
  // CHIRON_BEAM_GENERIC=1 forces the literal sequential kernel (the tests use it to cross-check the two)
  const char* fg = getenv("CHIRON_BEAM_GENERIC");
  const bool force_generic = fg && fg[0] == '1';
  // Two windows per wave pay when the windows outnumber the chip's SIMDs by enough for issue slots to count; a small batch
  // is latency-bound and finishes sooner with a wave per window (1.55 against 1.99 ms for any batch up to 1024 windows alone
  // on the chip).  CHIRON_BEAM_SINGLE=1: always one window per wave, =0: two whenever the width allows (A/B switch, tests).
  const char* f1 = getenv("CHIRON_BEAM_SINGLE");
  const bool single = f1 ? f1[0] == '1' : p.B < 512;
  if (p.beam <= 32 && p.T < 8192 && !force_generic && !single)
    hipLaunchKernelGGL(beam32x2_kernel, dim3((p.B + 1) / 2), dim3(64), 0, stream, p, node_cap);
  else if (p.beam <= 64 && p.T < 8192 && !force_generic)   // T: the depth field of the register kernel's packed entry
    hipLaunchKernelGGL(beam64_kernel, dim3(p.B), dim3(64), 0, stream, p, node_cap);
  else
    hipLaunchKernelGGL(beam_kernel, dim3(p.B), dim3(64), beam_smem_bytes(p.beam), stream, p, node_cap);
  return 0;
}

}  // namespace chiron
