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

__device__ __forceinline__ float log_sum_exp(float a, float b) {  // ctc_loss_util.h LogSumExp
  if (a == NEG_INF) return b;
  if (b == NEG_INF) return a;
  return a > b ? a + log1pf(expf(b - a)) : b + log1pf(expf(a - b));
}
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
    {
      float mx = lg[t * K];
      for (int k = 1; k < K; ++k) mx = fmaxf(mx, lg[t * K + k]);
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += expf(lg[t * K + k] - mx);
      const float lse = logf(s);
      for (int k = 0; k < K; ++k) logp[k] = (lg[t * K + k] - mx) - lse;
    }
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

static size_t beam_smem_bytes(int W) { return (size_t)W * 4 * (5 + 7 + 8 + 2); }

size_t beam_workspace_bytes(int B, int T, int beam) {
  return (size_t)B * (size_t)(1 + (size_t)beam * T) * sizeof(BeamNode);
}

int launch_beam(const BeamParams& p, hipStream_t stream) {
  if (p.beam < 1 || p.beam > BEAM_MAX || p.K != 5) return -1;
  const int node_cap = 1 + p.beam * p.T;
  if ((size_t)p.B * node_cap * sizeof(BeamNode) > p.workspace_bytes) return -2;
  hipLaunchKernelGGL(beam_kernel, dim3(p.B), dim3(64), beam_smem_bytes(p.beam), stream, p, node_cap);
  return 0;
}

}  // namespace chiron
