// Internal launch interface between the engine (engine.hip) and the gfx950 kernels.
// Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CHIRON_KMAX 5  // classes upper bound (reference: class_n = 5, rnn.py:25)

namespace chiron {

// Per-device launch facts.  One process may hold engines on several devices (include/chiron_amd.h: engines on different
// devices are independent), so nothing a launcher caches may be process-wide: the CU count and the "dynamic LDS opt-in
// done" flags are indexed by the current device.
constexpr int CHIRON_MAX_DEVICES = 64;
inline int current_device_index() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CHIRON_MAX_DEVICES) dev = 0;
  return dev;
}
inline int current_device_cus() {
  static int n_cu[CHIRON_MAX_DEVICES] = {};   // racing first calls store the same value
  const int dev = current_device_index();
  int n = __atomic_load_n(&n_cu[dev], __ATOMIC_RELAXED);
  if (n == 0) {
    hipDeviceProp_t prop;
    n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
    if (n <= 0) n = 256;
    __atomic_store_n(&n_cu[dev], n, __ATOMIC_RELAXED);
  }
  return n;
}

// ---------------------------------------------------------------------------------------------
// Fused conv / projection GEMM on v_mfma_f32_32x32x2_f32  (gemm.hip)
//
//   out[row(m)][n] = act( sum_k A(m,k) * Wt[n][k] + shift[n] + residual(m,n) )
//
// A(m,k) is assembled on the fly from up to GEMM_MAX_SEG K-segments (conv taps, or the fused
// branch1 input), each a strided/shifted view of an activation tensor -- or synthesised from the raw
// signal ("lift": the 1->C conv + BN + ReLU of res_layer1/branch2/conv2a is never materialised).
// ---------------------------------------------------------------------------------------------
constexpr int GEMM_MAX_SEG = 16;
constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_BK = 32;

struct GemmSeg {
  const float* src;  // activation tensor [rows][lda]; nullptr => lift from the signal
  int lda;           // row stride of src in floats
  int col0;          // first column of src used by this segment
  int cin;           // channels taken (K extent of the segment, before padding to GEMM_BK)
  int kpad;          // cin rounded up to GEMM_BK: K extent occupied in Wt
  int w_in;          // positions per sequence in src (conv mode)
  int stride;        // conv stride
  int shift;         // tap offset: in_t = t_out*stride + shift
  int time_major;    // src rows are (t*BP + b) instead of (b*w_in + in_t)
};

struct GemmParams {
  int M;        // output rows (conv: B*T_out; projection: T*BP)
  int T_out;    // conv mode: positions per sequence of the output
  int B;        // valid batch rows
  int BP;       // batch padded to 16 (projection / time-major modes)
  int m_time_major;  // 1: m = t*BP + b (LSTM projection), 0: m = b*T_out + t (conv)
  int N;        // valid output columns
  int K;        // total padded K (sum of kpad)
  int nseg;
  GemmSeg seg[GEMM_MAX_SEG];
  const float* Wt;     // [Npad][K], k contiguous, zero padded; Npad multiple of GEMM_BN
  const float* shift;  // [Npad] added before activation (folded BN offset / LSTM bias), may be null
  // dtype fp32-split only (nullptr otherwise): row n of Wt and shift[n] are stored multiplied by a power of two 2^s[n] chosen so that
  // the row's largest weight lies in [2^12, 2^13) -- the `lo` halves of a split weight are then normal halves (a weight of 0.03 has
  // lo = 1e-5, a SUBNORMAL half: 8 bits instead of 11, and being a constant its error is coherent over all positions; trained-like
  // weights lost 3 .. 8 x at the logits to that, profiles/r06_split_*) -- and descale[n] = 2^-s[n] is multiplied into the finished
  // accumulator (exact) before the epilogue.
  const float* descale;
  int relu;
  // lift (segments with src == nullptr): A = relu(sig[b][in_t]*lift_a[c] + lift_b[c]), 0 outside
  const float* sig;    // [B][L]
  int L;
  const float* lift_a;
  const float* lift_b;
  // residual synthesised from the signal (res_layer1/branch1: 1x1 conv on the signal + BN):
  //   + (sig[b][t_out*res_stride] * res_a[n] + res_b[n])
  // res_b = that branch's own folded BN shift (offset - mean*inv, of the order of -500 * res_a for a raw signal around 500).  It is
  // NOT folded into shift[]: the branch is evaluated as ONE fmaf whose exact value is small, and added to the finished
  // accumulator -- inside shift[] it would ride through the whole K chain and every step would round at ITS magnitude
  // (round 5: 1.3e-6 -> 0.8e-6 rms born in res_layer1 on trained-like weights, profiles/r05_parity_error_structure.txt).
  const float* res_a;
  const float* res_b;
  int res_stride;
  // output
  float* out;
  int ldo;         // row stride (out_mode 0)
  int out_mode;    // 0: out[m*ldo + n];  1: LSTM z layout [t][4-row group][dir][col][4 rows] (see lstm.hip)
  int z_cols;      // out_mode 1: z columns per direction (4*H, TF gate-major order i, j, f, o)
  int z_ndir;      // out_mode 1: directions interleaved in N (N = z_ndir * z_cols)
  int z_dir0;      // out_mode 1: first direction index written by this launch
  int z_dirs_total;  // out_mode 1: directions in the z buffer (2)
  const int32_t* z_seq_len;  // out_mode 1: [BP] sequence lengths; direction 1 is stored time-reversed per row:
                             //   z[s][b][dir 1] = x-projection of frame seq_len[b]-1-s (tf.reverse_sequence folded in)
  // f16 = 1 (engine dtype CHIRON_F16): A, Wt and the conv output hold IEEE halves; lda / col0 / cin / kpad / K and
  // ldo are then counted in 4-BYTE UNITS (two halves) so that the loader geometry is the fp32 one; accumulation,
  // shift and the z output stay fp32.
  int f16;
  int z_f16;  // out_mode 1: z is written as halves (f16 engine: the recurrence converts back; halves the z traffic)
  // Dynamic tile scheduling of the DMA kernel (round 3): eight counters in device memory, one per XCD (a workgroup's XCD is
  // blockIdx.x & 7, as in the static mapping).  A workgroup takes its next tile slot with one atomic add instead of striding
  // by gridDim.x: with three batches in flight part of a launch's persistent workgroups find their CU occupied (a recurrence
  // workgroup of another batch leaves room for one GEMM workgroup, not two) and start late -- with fixed shares the launch
  // ends when the last of them has worked off its 34 tiles.  The counters are never reset: a launch takes exactly
  // (slots per XCD + workgroups per XCD) numbers from each, so the host knows where the next launch of that stream starts.
  unsigned long long* tile_ctr;        // device, [8]; nullptr: static shares
  unsigned long long tile_base;        // first number of this launch (filled by launch_gemm)
  unsigned long long* tile_base_host;  // host counter of the owning stream, advanced by launch_gemm
};

bool launch_gemm(const GemmParams& p, hipStream_t stream);  // false: no kernel for this shape / dtype
// fp16 engine, 1 x 1 convolutions with 256 channels in and out (stream16.hip): weights in registers, activations streamed through
// the LDS.  Takes GemmParams in ELEMENT units (before launch()'s conversion to 4-byte units); false: shape not covered.
bool launch_stream16(const GemmParams& p, hipStream_t stream);
// fp32 engine, 1 x 1 convolutions with 256 channels in and out (stream32.hip): the whole weight matrix in registers, activations
// streamed through the LDS by producer waves.  false: shape not covered (the caller takes launch_gemm).
bool launch_stream32(const GemmParams& p, hipStream_t stream);


// 1 x 3 stride-1 convolution in Winograd F(2,3) form (wino.hip): fp32, channels-last, T even, C % 32 == 0, N % 64 == 0
struct WinoParams {
  const float* src;    // [B*T][lda] activations
  const float* U;      // F(2,3): [4][N][C] transformed filters (BN scale folded): g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2;
                       // F(4,3) (f4 = 1): [6][N][C] = G g
  const float* shift;  // [N] folded BN offset
  float* out;          // [B*T][ldo]
  int B, T, C, N, lda, ldo, relu;
  int f4;              // 1: F(4,3) (T % 4 == 0), 0: F(2,3) (T % 2 == 0)
  // dynamic tile numbers of the F(4,3) kernel: as GemmParams::tile_ctr (per-XCD counters of the launching stream)
  unsigned long long* tile_ctr;
  unsigned long long tile_base;
  unsigned long long* tile_base_host;
};
bool launch_wino_conv3(const WinoParams& p, hipStream_t stream);  // false: shape not covered

// ---------------------------------------------------------------------------------------------
// LSTM recurrence (lstm.hip): one workgroup = 4*NG batch rows x one direction x all T steps.
// ---------------------------------------------------------------------------------------------
constexpr int LSTM_K = 100;      // hidden size the kernel is built for (rnn.py:23 hidden_num=100)
constexpr int LSTM_NW = 7;       // waves per 4-row group: six own hidden units [16w, 16w+16), the seventh units 96..99 (fp32) / [96, 112) (f16)

struct LstmParams {
  const float* z;        // [T][BP/4][ndir][4*H][4 rows]  x-projection + bias, column = gate*H + unit; direction 1 is
                         //   indexed by STEP (frame seq_len-1-s), direction 0 by frame
  const float* wfrag;    // [ndir][LSTM_NW][LSTM_K][64 lanes] recurrent weights, fragment order
  const float* wlight;   // [ndir][28][64 lanes] units 96..99 in the K-split order of the light wave (fp32 kernel):
                         //   entry m = 4q + a, lane = kg*16 + gate*4 + j  ->  W_hh[k = 16q + 4kg + a][gate*H + 96 + j]
  const float* wwide32;  // fp32 wide form (lstm32w_kernel): [ndir][8 waves][4 tile slots][25 k-steps][64 lanes], lane = kq*16 + 4u + gate ->
                         //   W_hh[k = 4 ks + kq][gate*H + 4 tile + u], tile = 3 wave + slot (zero for the slots a wave does not use)
  int out_f32;           // f16 engines, LAST layer (round 6): the layer's output -- what the FC head reads -- is written as fp32, from the cell's
                         //   own fp32 h, instead of as halves (the recurrent operand h stays a half).  tools/f16_sites.py: under a head with
                         //   weights of order 100 this ONE rounding flips more windows than all other activation roundings together
  const void* wsplit;    // dtype fp32-split (lstm32s_kernel): W_hh as hi + lo half pairs, [hi | lo] x the order of `wwide`; nullptr: the fp32 kernels
  void* out_split;       // lstm32s_kernel: when set, the output goes HERE in the split hi / lo format the next layer's projection reads
                         //   ([T * BP rows][split_ld], per 32-element block 32 hi halves then 32 lo halves) instead of fp32 to `out`
  int split_ld;          //   elements per row; split_bw0: first element of the backward direction's H columns (H: the directions are
  int split_bw0;         //   contiguous; roundup(H, 32): MultiRNN, each direction a K-segment of its own)
  int form32;            // fp32: 0 = the 4-row kernels (lstm_kernel), 1 = lstm32w_kernel, 2 = lstm32w2_kernel (CHIRON_LSTM_WIDE)
  const void* wwide;     // f16 wide form: [ndir][8 waves][4 tile slots][7 k-steps][64 lanes][4 halves], lane = kq*16 + 4u + gate ->
                         //   W_hh[k = 16 ks + 4 kq + e][gate*H + 4 tile + u], tile = 3 wave + slot (zero past K or the wave's tiles)
  // f16, fused with the x-projection (lstm16f_kernel; xsrc == nullptr: z comes from the projection GEMM as above)
  const void* xsrc;      // layer input, halves: [B][T][xld] (x_time_major 0: CNN features) or [T][BP][xld] (1: previous lasth)
  const void* wxwide;    // [ndir][8 waves][4 tile slots][KSX k-steps of 32][64 lanes][8 halves]: W_x, lane = kg*16 + 4u + gate ->
                         //   W_x[k = 32 ks + 8 kg + e][gate*H + 4 tile + u]; KSX = 8 (K = 256) or 7 (K = 200)
  const void* whfused;   // W_hh in the same order, 4 k-steps of 32 (K = 100 padded to 128)
  int fused_pair;        // 1: two 16-row groups per workgroup (A/B switch CHIRON_LSTM16_PAIR)
  const float* xbias;    // [ndir][4H] bias + forget bias, column = gate*H + unit (the projection GEMM's shift vector)
  int xK, xld, x_time_major;
  const int32_t* seq_len;  // [BP] (0 for padded rows)
  float* out;            // lasth [T][BP][ndir*H] time major
  int T, B, BP, H;
  int ndir;              // 2
  int paired;            // fp32: 1 = 14-wave workgroups (two groups per CU) for the part of the batch that fits one resident round
  int fixed_roles;       // fp32: 1 = wave 6 is always the light wave (A/B switch CHIRON_LSTM_FIXED_ROLES; results are the same)
  int group0;            // first 4-row group this launch covers (blockIdx 0); launch_lstm splits a batch into a paired part
                         //   and a remainder
  int narrow16;          // f16: 1 = 4-row workgroups only (A/B switch CHIRON_LSTM16_NARROW)
  int w2;                // f16 engine with hi + lo weights (CHIRON_F16_W2): wsplit holds W_hh, z arrives as fp32, out as halves (lstm16w2_kernel)
  int f16;               // 1: wfrag holds halves in 4x4x4 fragment order, out (lasth) is written as halves; z stays fp32
};
void launch_lstm(const LstmParams& p, hipStream_t stream);
constexpr int LSTM_KSTEPS16 = 25;  // k-steps of v_mfma_f32_4x4x4_16B_f16 covering K = 100

// relu(sig*a[c] + b[c]) [B*L][C]: res_layer1/conv2a materialised in the engine's activation format
// Stem convolution of HEAD's RNA models (cnn.py:454-476): out[(b,t)][c] = act(sum_tap sig[b][t*stride + tap - left] * w[tap][c]
// + shift[c]) with TF SAME padding; fmt 0 fp32, 1 halves, 2 split hi/lo; relu = 0 leaves the raw convolution (batch-BN mode).
void launch_stem_conv(const float* sig, const float* w, const float* shift, void* out, int B, int L, int T_out, int k, int stride, int left,
                      int C, int fmt, int relu, hipStream_t stream);
// fp32 [rows][cols] -> split hi/lo format [rows][ld] (ld a multiple of 32; padding columns are left untouched).
// half > 0: the row is two halves of `half` columns (fw | bw) and the second one starts at column half_dst of the
// destination (a 32-element block boundary), so that each direction can be read as a K-segment of its own.
void launch_split_convert(const float* src, void* dst, long rows, int cols, int ld, int half, int half_dst, hipStream_t stream);
// res_layer1 conv2a + conv2b as a piecewise-linear table (pwl.hip)
struct PwlConvParams {
  const float* sig;   // [B][L]
  const float* bp;    // [nbp] breakpoints of relu(s*a[c] + b[c]), ascending
  const float* ref;   // [nbp + 1] reference point of every interval (its point nearest to 0)
  const float2* tab;  // [nbp + 1][k][C] (alpha, f(ref)): f[tap][n](s) = alpha*(s - ref) + f(ref) on the interval
  const float* shift; // [C] folded BN offset of conv2b
  void* out;          // [B*T_out][C] in the engine's activation format
  int B, L, T_out, k, stride, left, C, nbp;
  int fmt;            // 0 fp32, 1 halves, 2 split
};
bool launch_pwl_conv(const PwlConvParams& p, hipStream_t stream);

void launch_lift(const float* sig, const float* a, const float* b, void* out, long n_pos, int C, int fmt /* 0 fp32, 1 halves, 2 split */,
                 hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// FC head + CTC (head_ctc.hip)
// ---------------------------------------------------------------------------------------------
struct FcParams {
  const float* lasth;  // [T][BP][2H] time major
  const float* w;      // [2][H]
  const float* bias;   // [H]
  const float* wc;     // [H][K]
  const float* bc;     // [K]
  float* logits;       // [B][T][K]
  int T, B, BP, H, K;
  int f16;  // lasth holds halves
  int split;  // lasth in the split hi/lo format, rows of ld elements
  int ld;
};
void launch_fc(const FcParams& p, hipStream_t stream);

struct GreedyParams {
  const float* logits;      // [B][T][K]
  const int32_t* seq_len;   // [B]
  uint8_t* labels;          // [B][T] decoded labels (dense, first count[b] valid)
  int32_t* count;           // [B]
  float* log_prob;          // [B] -sum of max logits over t < seq_len
  float* prob_logits;       // [B] path_prob (mean over ALL T of top1-top2), or nullptr
  int B, T, K;
};
void launch_greedy(const GreedyParams& p, hipStream_t stream);

struct SparseParams {
  const uint8_t* labels;  // [B][T]
  const int32_t* count;   // [B]
  int64_t* offsets;       // [B+1] scratch
  int64_t* indices;       // [nnz][2]
  int64_t* values;        // [nnz]
  int64_t* meta;          // [3]: nnz, batch, max_len
  int B, T;
};
void launch_sparse(const SparseParams& p, hipStream_t stream);

struct BeamParams {
  const float* logits;     // [B][T][K]
  const int32_t* seq_len;  // [B]
  uint8_t* labels;         // [B][T]
  int32_t* count;          // [B]
  float* log_prob;         // [B]
  void* workspace;         // trie nodes
  size_t workspace_bytes;
  int B, T, K, beam;
};
size_t beam_workspace_bytes(int B, int T, int beam);
int launch_beam(const BeamParams& p, hipStream_t stream);  // returns 0 on success

struct PathProbParams {
  const float* logits;
  float* prob_logits;
  int B, T, K;
};
void launch_path_prob(const PathProbParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// batch-statistics BatchNorm (bn_batch.hip): cnn.py:166-188 simple_global_bn
// ---------------------------------------------------------------------------------------------
void launch_bn_stats(const float* x, long M, int C, double* sums /* [2][C] */, hipStream_t stream);
// sums[c] += sum_m x[m * ld + col0 + c] over `rows` rows of halves (BP > 0: time-major rows m = t * BP + b, only b < B count)
void launch_colsum_f16(const void* x, long rows, int ld, int col0, int cin, int BP, int B, double* sums /* [cin], zeroed by the caller */, hipStream_t stream);
void launch_bn_apply(float* x, const double* sums, const float* scale, const float* offset, long M, int C, int relu, const float* add,
                     const double* add_sums, const float* add_scale, const float* add_offset, hipStream_t stream);
void launch_rank1_conv(const float* sig, const float* w, float* out, long n_pos, int T_out, int L, int stride, int C, hipStream_t stream);

}  // namespace chiron
