// libchiron_amd.so -- engine: C ABI (include/chiron_amd.h), weight preparation, workspace, and the
// launch sequence that replaces chiron_model.inference (chiron_model.py:134-172) + the decode
// sub-graph (chiron_eval.py:465-492) of the reference.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/chiron_amd.h"
#include "kernels.h"
#include <dlfcn.h>

// profiling buckets (chiron_engine_profile_read) = names of the roctx ranges (CHIRON_ROCTX); order = the PN_* enum below
static const char* const kProfNames[] = {"conv_dma", "lstm_proj_dma", "lstm_recurrence", "fc_head", "ctc_greedy", "ctc_beam", "sparse_build",
                                         "path_prob", "conv_lift", "conv_res", "conv1_pwl", "conv_wino", "lstm_proj0_dma", "conv2a"};

using namespace chiron;

// ----------------------------------------------------------------------------------------------
// error plumbing: never throw across the ABI
// ----------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static chiron_status fail(chiron_status st, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return st;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(CHIRON_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                  __LINE__);                                                                       \
  } while (0)

namespace chiron {
chiron_status set_error(chiron_status st, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return st;
}
}  // namespace chiron

extern "C" const char* chiron_last_error(void) { return g_err; }
extern "C" int32_t chiron_abi_version(void) { return CHIRON_ABI_VERSION; }

// Objects built with CHIRON_TIMING_BUILD (timing_variants.h: parts of a kernel switched off, garbage results) register here from
// a static constructor; the flag is constant-initialised, so the order of static initialisation does not matter.
static int g_timing_build = 0;
static char g_timing_source[96] = "";
namespace chiron {
void mark_timing_build(const char* source) {
  g_timing_build = 1;
  snprintf(g_timing_source, sizeof(g_timing_source), "%s", source ? source : "?");
}
}  // namespace chiron
extern "C" uint32_t chiron_build_flags(void) { return g_timing_build ? CHIRON_BUILD_TIMING : 0u; }

extern "C" chiron_status chiron_device_pci_bus_id(int32_t device_id, char* out, size_t cap) {
  if (!out || cap < 16) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_device_pci_bus_id: buffer of at least 16 bytes needed");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return chiron::set_error(CHIRON_ERR_DEVICE, "no HIP device");
  if (device_id < 0 || device_id >= n) return chiron::set_error(CHIRON_ERR_INVALID, "device %d of %d", device_id, n);
  hipError_t err = hipDeviceGetPCIBusId(out, (int)cap, device_id);
  if (err != hipSuccess) return chiron::set_error(CHIRON_ERR_DEVICE, "hipDeviceGetPCIBusId: %s", hipGetErrorString(err));
  for (char* c = out; *c; ++c)
    if (*c >= 'A' && *c <= 'F') *c += 'a' - 'A';   // sysfs spells PCI addresses in lower case
  return CHIRON_OK;
}

static int roundup(int v, int m) { return (v + m - 1) / m * m; }

// TF 'SAME' padding: out = ceil(W/s), pad_total = max((out-1)s + k - W, 0), left = total/2
static void same_pad(int w, int k, int s, int* out, int* left) {
  *out = (w + s - 1) / s;
  int tot = (*out - 1) * s + k - w;
  if (tot < 0) tot = 0;
  *left = tot / 2;
}

// ----------------------------------------------------------------------------------------------
// model description helpers
// ----------------------------------------------------------------------------------------------
static chiron_status validate_desc(const chiron_model_desc* d) {
  if (!d) return fail(CHIRON_ERR_INVALID, "null model descriptor");
  if (d->n_blocks < 1 || d->n_blocks > CHIRON_MAX_BLOCKS) return fail(CHIRON_ERR_INVALID, "n_blocks %d out of range", d->n_blocks);
  for (int i = 0; i < d->n_blocks; ++i) {
    const chiron_res_block& b = d->blocks[i];
    const int want_in = i == 0 ? (d->stem_k > 0 ? d->stem_channels : 1) : d->blocks[i - 1].out_channels;
    if (b.in_channels != want_in) return fail(CHIRON_ERR_INVALID, "block %d: in_channels %d, expected %d", i, b.in_channels, want_in);
    if (b.out_channels < 4 || b.out_channels % 4) return fail(CHIRON_ERR_INVALID, "block %d: out_channels must be a multiple of 4", i);
    if (b.k < 1 || b.k > GEMM_MAX_SEG) return fail(CHIRON_ERR_INVALID, "block %d: conv2b width %d unsupported (1..%d)", i, b.k, GEMM_MAX_SEG);
    if (b.stride < 1) return fail(CHIRON_ERR_INVALID, "block %d: stride %d", i, b.stride);
  }
  if (d->stem_k < 0 || d->stem_k > 64 || (d->stem_k > 0 && (d->stem_stride < 1 || d->stem_channels < 8 || d->stem_channels % 8)))
    return fail(CHIRON_ERR_INVALID, "stem: k %d stride %d channels %d", d->stem_k, d->stem_stride, d->stem_channels);
  if (d->rnn_kind != CHIRON_RNN_STACK && d->rnn_kind != CHIRON_RNN_MULTI) return fail(CHIRON_ERR_INVALID, "rnn_kind %d", d->rnn_kind);
  if (d->rnn_layers < 1 || d->rnn_layers > 8) return fail(CHIRON_ERR_INVALID, "rnn_layers %d unsupported (1..8)", d->rnn_layers);
  if (d->hidden < 4 || d->hidden > 100 || d->hidden % 4) return fail(CHIRON_ERR_INVALID, "hidden %d unsupported (multiple of 4, <= 100)", d->hidden);
  if (d->classes < 2 || d->classes > CHIRON_KMAX) return fail(CHIRON_ERR_INVALID, "classes %d unsupported (2..%d)", d->classes, CHIRON_KMAX);
  if (d->bn_mode != CHIRON_BN_POPULATION && d->bn_mode != CHIRON_BN_BATCH) return fail(CHIRON_ERR_INVALID, "bn_mode %d", d->bn_mode);
  return CHIRON_OK;
}

static int lstm_in_width(const chiron_model_desc* d, int layer) {
  if (layer == 0) return d->blocks[d->n_blocks - 1].out_channels;
  return d->rnn_kind == CHIRON_RNN_STACK ? 2 * d->hidden : d->hidden;
}

extern "C" chiron_status chiron_weights_size(const chiron_model_desc* d, size_t* n_floats) {
  chiron_status st = validate_desc(d);
  if (st) return st;
  if (!n_floats) return fail(CHIRON_ERR_INVALID, "null n_floats");
  size_t n = 0;
  if (d->stem_k > 0) n += (size_t)d->stem_k * d->stem_channels + 4 * (size_t)d->stem_channels;
  for (int i = 0; i < d->n_blocks; ++i) {
    const chiron_res_block& b = d->blocks[i];
    const size_t ci = b.in_channels, co = b.out_channels;
    n += ci * co + (b.i_bn ? 4 * co : 0);  // branch1
    n += ci * co + 4 * co;                 // conv2a
    n += (size_t)b.k * co * co + 4 * co;   // conv2b
    n += co * co + 4 * co;                 // conv2c
  }
  const size_t H = d->hidden;
  for (int l = 0; l < d->rnn_layers; ++l) n += 2 * ((lstm_in_width(d, l) + H) * 4 * H + 4 * H);
  n += 2 * H + H + H * d->classes + d->classes;
  *n_floats = n;
  return CHIRON_OK;
}

// ----------------------------------------------------------------------------------------------
// engine state
// ----------------------------------------------------------------------------------------------
// Default form of the fp32 recurrence (DESIGN 3.2 has the same-box A/B figures behind the choice)
#define CHIRON_LSTM_WIDE_DEFAULT 2

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct ConvGemmPlan {
  // device weights for one fused GEMM
  float* Wt = nullptr;
  float* shift = nullptr;
  float* descale = nullptr;   // dtype fp32-split: 2^-s[n] of the power-of-two row scaling (GemmParams::descale)
  int N = 0, Npad = 0, K = 0;
};

// f16 engine, calibration (chiron_engine_calibrate): what a plan's weights lost when they were rounded to halves.
// y[n] = sum_k x[k] f16(W[n][k]) = sum_k x[k] W[n][k] + sum_k x[k] dW[n][k]; the second sum's MEAN over the data, sum_k E[x_k] dW[n][k],
// is a constant per output channel and is taken out of the plan's shift once the input channels' means are known.
struct PlanHost {
  std::vector<float> dW;       // [Npad][K]: (float)(_Float16)W - W of the BN-folded fp32 weight (0 in the K / N padding)
  std::vector<float> shift0;   // the uncorrected shift
  int N = 0, Npad = 0, K = 0;
};
struct CalibRecord {           // one measured input: columns [k0, k0 + cin) of the plan behind `shift`, or rows of an LSTM kernel
  const float* shift;          // device shift array of the plan (key into plan_host), or the projection's for an LSTM layer
  int k0, cin;
  int lstm_layer, lstm_dir, lstm_part;   // lstm_layer >= 0: part 0 = x rows [k0, k0 + cin) of the TF kernel, part 1 = its h rows; dir -1 = both
  double rows;                 // rows summed
  size_t slot;                 // index of the record's [256] doubles in the device sum buffer
};
struct CalibCtx {
  double* dev_sums = nullptr;  // [max_records][256]
  size_t max_records = 0;
  std::vector<CalibRecord> rec;
  int skipped = 0;             // inputs that could not be measured (more than 256 channels, or more records than max_records)
};

struct BlockPlan {
  bool lift = false;
  int c_in = 0, c = 0, k = 0, stride = 1, left = 0;
  int t_in = 0, t_out = 0;
  float *lift_a = nullptr, *lift_b = nullptr;  // lift: conv2a folded scale/shift
  float *res_a = nullptr, *res_b = nullptr;    // lift: branch1 folded scale and its own folded shift (kernels.h res_b)
  // lift, population BN: conv2a + conv2b as a piecewise-linear table of the signal value (pwl.hip)
  float *pwl_bp = nullptr, *pwl_ref = nullptr, *pwl_tab = nullptr, *pwl_shift = nullptr;
  int pwl_nbp = 0;
  ConvGemmPlan ga, gb, gc;                     // conv2a (non-lift), conv2b, conv2c(+conv1)
  float* wino_u = nullptr;                     // conv2b in Winograd form (wino.hip): transformed filters [4 or 6][C][C], or null
  int wino_f4 = 0;                             // 1: F(4,3) (six filters, length % 4 == 0), 0: F(2,3)
  // bn_mode = batch (cnn.py:166-188): the GEMM weights above are raw, gc holds conv2c alone, g1 the 1x1 branch1 conv;
  // scale / offset of the four BN sites (conv1 only when i_bn)
  ConvGemmPlan g1;
  bool i_bn = false;
  float *bn_scale[4] = {nullptr, nullptr, nullptr, nullptr}, *bn_offset[4] = {nullptr, nullptr, nullptr, nullptr};  // conv1, 2a, 2b, 2c
};

struct LstmPlan {
  int in_w = 0;
  ConvGemmPlan proj[2];  // STACK / layer 0: proj[0] covers both directions; MULTI l>0: one per dir
  int nproj = 1;
  float* wfrag = nullptr;
  void* wwide = nullptr;    // f16: recurrent weights in the 16x16x16 B-operand order of lstm16w_kernel
  void* whfused = nullptr;  // f16: W_hh in the 16x16x32 order of lstm16f_kernel
  void* wxwide = nullptr;   // f16: input weights in that order (lstm16f_kernel: projection fused into the recurrence)
  int wx_ksteps = 0;        //      its k-steps of 16 (16: K = 256, 13: K = 200)
  void* wsplit = nullptr;   // fp32-split: W_hh as hi + lo half pairs in the order of wwide (lstm32s_kernel)
  float* wwide32 = nullptr; // fp32: recurrent weights in the 16x16x4 B-operand order of lstm32w_kernel
  float* wlight = nullptr;  // K-split fragment of units 96..99 for the paired recurrence (fp32, H = 100)
};

struct ProfEvent {
  int name_id;
  hipEvent_t a, b;
  double flops, bytes;
};

struct Slot {
  hipStream_t stream = nullptr;
  unsigned long long* tile_ctr = nullptr;   // device [16]: [0..7] per-XCD tile counters of the DMA GEMM / Winograd kernels (GemmParams::tile_ctr),
                                            // [8] the streaming 1 x 1 kernel's; never reset
  unsigned long long tile_base = 0;         // host: the number the stream's next GEMM / Winograd launch starts at
  unsigned long long tile_base32 = 0;       // host: the same for counter [8]
  float* sig = nullptr;
  int32_t* seq = nullptr;
  float* act[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // 3 in population mode, 5 in batch-BN mode
  double* bn_sums = nullptr;  // batch-BN mode: two sets of [2][C] sums
  float* z = nullptr;
  float* lasth[2] = {nullptr, nullptr};
  float* lasth_f32 = nullptr;  // dtype fp32-split: the recurrence writes plain fp32 here, a pass converts it for the next GEMM
  float* logits = nullptr;
  uint8_t* labels = nullptr;
  int32_t* count = nullptr;
  float* log_prob = nullptr;
  float* prob = nullptr;
  int64_t* offsets = nullptr;
  int64_t* indices = nullptr;
  int64_t* values = nullptr;
  int64_t* meta = nullptr;
  void* beam_ws = nullptr;
  size_t beam_ws_bytes = 0;
  // pinned host
  float* h_sig = nullptr;
  int32_t* h_seq = nullptr;
  int64_t* h_indices = nullptr;
  int64_t* h_values = nullptr;
  int64_t* h_meta = nullptr;
  float* h_log_prob = nullptr;
  float* h_prob = nullptr;
  float* h_logits = nullptr;
  uint8_t* h_labels = nullptr;   // CHIRON_COMPACT_DECODE: the device's per-row label strings [B][T] ...
  int32_t* h_count = nullptr;    // ... and their lengths [B], copied in-stream behind the decode
  uint8_t* h_flat = nullptr;     // the rows' labels back to back, built by collect (plain host memory)
  // state of the in-flight batch
  // 0 = idle, 1 = a batch is in flight (between a successful submit / decode and its collect).  Atomic: submit and
  // collect of one slot may come from different threads (one producer, one consumer per engine; include/chiron_amd.h).
  struct State {
    std::atomic<int> v{0};
    State() {}
    State(const State& o) : v(o.v.load()) {}
    State& operator=(const State& o) {
      v.store(o.v.load());
      return *this;
    }
  } state;
  int batch = 0;
  uint32_t flags = 0;
  const float* sig_used = nullptr;
  // what chiron_engine_features / chiron_engine_rnn_output may hand out: the batch size of the last NETWORK batch of this slot
  // (0 after a decode-only batch or a failed submit: `batch` alone also counts chiron_engine_decode, which runs no network)
  int net_batch = 0;
  const float* rnn_out = nullptr;   // the last layer's lasth (time-major [T][BP][lasth_ld]; rnn_out_f32 = false: halves)
  bool rnn_out_f32 = true;
  std::vector<ProfEvent> events;
};

struct chiron_engine {
  chiron_model_desc desc;
  chiron_engine_opts opts;
  int L = 0, T = 0, C = 0, H = 0, K = 0;
  int maxB = 0, BP = 0;
  bool stream32 = true;           // fp32: 1 x 1 convolutions on the weight-stationary streaming kernel (CHIRON_NO_STREAM32=1: gemm.hip, A/B switch)
  bool stream16 = true;           // f16: 1 x 1 convolutions on the streaming kernel (CHIRON_NO_STREAM16=1: gemm.hip, A/B switch)
  bool dyn_tiles = true;          // DMA GEMM: tiles handed out by per-XCD counters instead of fixed shares (CHIRON_STATIC_TILES=1: off)
  int lstm16_pair = -1;           // f16 fused recurrence, two 16-row groups per workgroup: 1 always, 0 never (CHIRON_LSTM16_PAIR), -1 when single groups would not fit one round
  bool lstm16_fused = false;      // f16: x-projection inside the recurrence (whole 16-row groups that fill the CUs)
  bool lstm16_narrow = false;     // A/B switch: f16 recurrence on 4-row workgroups only
  bool lstm_fixed_roles = false;  // A/B switch: light role always on wave 6
  int lstm_form = 0;              // fp32 recurrence: 0 = 4-row kernels (lstm_kernel), 1 = lstm32w_kernel, 2 = lstm32w2_kernel
  bool lstm_paired = false;  // fp32 recurrence: 14-wave workgroups for the part of a batch that fits one resident round
  bool bn_batch = false;  // desc.bn_mode == CHIRON_BN_BATCH
  bool f16 = false;    // opts.dtype == CHIRON_F16: halves for activations / weights, fp32 accumulate, z, gates, logits
  bool w2 = false;     // opts.dtype == CHIRON_F16_W2: an f16 engine (f16 is set too) whose weights are exact hi + lo half pairs -- every GEMM
                       // runs its K-segments twice ([x, x] . [W_hi; W_lo], tiled DMA GEMMs only), z stays fp32, the recurrence is lstm16w2_kernel
  bool lasth32 = true;    // f16 engines: the LAST recurrent layer writes its output (what the FC head reads) as fp32 (LstmParams::out_f32);
                          // CHIRON_F16_LASTH16=1: halves as rounds 2 .. 5 (A/B)
  bool w2_zf16 = false;   // f16-w2, A/B switch CHIRON_W2_ZF16=1: z as halves between projection and recurrence
  bool split = false;  // opts.dtype == CHIRON_F32_SPLIT: fp32 values as hi/lo half pairs on the f16 matrix cores (GEMMs only)
  int lasth_ld = 0;    // elements per lasth row (2H; split: rounded up to whole 32-element blocks)
  int kq = GEMM_BK;    // K padding quantum in elements: one LDS chunk = 128 bytes per row (32 floats / 64 halves)
  // stem (HEAD RNA_model2 / RNA_model3): folded filter [k][C], shift [C]; batch-BN mode: raw filter + scale / offset
  int stem_k = 0, stem_stride = 1, stem_left = 0, stem_t = 0, stem_c = 0;
  float *stem_w = nullptr, *stem_shift = nullptr, *stem_scale = nullptr, *stem_offset = nullptr;
  std::vector<BlockPlan> blocks;
  std::vector<LstmPlan> lstm;
  float *fc_w = nullptr, *fc_b = nullptr, *fc_wc = nullptr, *fc_bc = nullptr;
  std::vector<Slot> slots;
  std::map<const float*, PlanHost> plan_host;   // f16 engine: keyed by the plan's device shift pointer
  std::vector<float> host_weights;              // f16 engine: the caller's blob (LSTM kernels are read back from it in calibration)
  std::vector<size_t> lstm_kernel_off[2];       // offset of layer l's kernel of direction d in host_weights
  CalibCtx* calib = nullptr;                    // non-null while chiron_engine_calibrate runs the network
  int calibrated = 0;                           // iterations applied so far
  std::vector<void*> owned;  // device allocations freed on destroy
  bool profiling = false;
  std::vector<std::string> prof_names;
  std::mutex mu;
};

static chiron_status dev_alloc(chiron_engine* e, void** p, size_t bytes, bool zero) {
  HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
  e->owned.push_back(*p);
  if (zero) HIP_TRY(hipMemset(*p, 0, bytes ? bytes : 16));
  return CHIRON_OK;
}
template <typename Tp>
static chiron_status dev_upload(chiron_engine* e, Tp** p, const std::vector<Tp>& h) {
  chiron_status st = dev_alloc(e, (void**)p, h.size() * sizeof(Tp), false);
  if (st) return st;
  HIP_TRY(hipMemcpy(*p, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice));
  return CHIRON_OK;
}

// folded BN (cnn.py:125-163 population branch; association order of the .meta graph):
//   inv = rsqrt(var + 1e-5) * scale ; y = x*inv + (offset - mean*inv)
struct BnFold {
  std::vector<float> inv, sh;
};
static BnFold fold_bn(const float* scale, const float* offset, const float* mean, const float* var, int n) {
  BnFold f;
  f.inv.resize(n);
  f.sh.resize(n);
  for (int i = 0; i < n; ++i) {
    const float inv = (1.0f / sqrtf(var[i] + 1e-5f)) * scale[i];
    f.inv[i] = inv;
    f.sh[i] = offset[i] - mean[i] * inv;
  }
  return f;
}

static chiron_status upload_gemm(chiron_engine* e, ConvGemmPlan* g, const std::vector<float>& Wt, const std::vector<float>& shift, int N, int Npad, int K) {
  g->N = N;
  g->Npad = Npad;
  g->K = K;
  chiron_status st;
  if (e->split) {
    // per 32-element block of a row: 32 hi halves then 32 lo halves (K is a multiple of 32).  Row n is scaled by 2^s[n] so that its
    // largest weight lies in [2^12, 2^13): hi <= 8192 is far from a half's 65504, and lo = O(2^-11 w) is a NORMAL half for every
    // weight down to 2^-15 of the row's largest (GemmParams::descale).  CHIRON_SPLIT_NO_ROW_SCALE=1: the unscaled format of round 5 (A/B).
    static const bool row_scale = getenv("CHIRON_SPLIT_NO_ROW_SCALE") == nullptr;
    std::vector<_Float16> h(2 * Wt.size());
    std::vector<float> sh2(shift), ds((size_t)Npad + 192, 1.0f);
    for (int n = 0; n < Npad; ++n) {
      float mx = 0.f;
      for (int k = 0; k < K; ++k) mx = std::max(mx, fabsf(Wt[(size_t)n * K + k]));
      int s = 0;
      if (row_scale && mx > 0.f && std::isfinite(mx)) {
        int ex;
        frexpf(mx, &ex);                       // mx = f * 2^ex, 0.5 <= f < 1
        s = std::min(60, std::max(-60, 13 - ex));
        if (n < (int)sh2.size() && !std::isfinite(ldexpf(sh2[n], s))) s = 0;
      }
      ds[n] = ldexpf(1.0f, -s);
      if (n < (int)sh2.size()) sh2[n] = ldexpf(sh2[n], s);
      for (int k = 0; k < K; ++k) {
        const size_t i = (size_t)n * K + k;
        const float v = ldexpf(Wt[i], s);
        const _Float16 hi = (_Float16)v;
        const size_t blk = i / 32, el = i % 32;
        h[blk * 64 + el] = hi;
        h[blk * 64 + 32 + el] = (_Float16)(v - (float)hi);
      }
    }
    _Float16* d = nullptr;
    if ((st = dev_upload(e, &d, h))) return st;
    g->Wt = reinterpret_cast<float*>(d);
    if ((st = dev_upload(e, &g->descale, ds))) return st;
    return dev_upload(e, &g->shift, sh2);
  } else if (e->w2) {
    // every row [K hi halves | K lo halves]: launch() runs the K-segments of a GEMM twice, the second time against the lo columns
    std::vector<_Float16> h(2 * Wt.size());
    for (int n = 0; n < Npad; ++n)
      for (int k = 0; k < K; ++k) {
        const float v = Wt[(size_t)n * K + k];
        const _Float16 hi = (_Float16)v;
        h[(size_t)n * 2 * K + k] = hi;
        h[(size_t)n * 2 * K + K + k] = (_Float16)(v - (float)hi);
      }
    _Float16* d = nullptr;
    if ((st = dev_upload(e, &d, h))) return st;
    g->Wt = reinterpret_cast<float*>(d);
  } else if (e->f16) {
    std::vector<_Float16> h(Wt.size());
    for (size_t i = 0; i < Wt.size(); ++i) h[i] = (_Float16)Wt[i];
    _Float16* d = nullptr;
    if ((st = dev_upload(e, &d, h))) return st;
    g->Wt = reinterpret_cast<float*>(d);
    if ((st = dev_upload(e, &g->shift, shift))) return st;
    PlanHost& ph = e->plan_host[g->shift];
    ph.dW.resize(Wt.size());
    for (size_t i = 0; i < Wt.size(); ++i) ph.dW[i] = (float)h[i] - Wt[i];
    ph.shift0 = shift;
    ph.N = N, ph.Npad = Npad, ph.K = K;
    return CHIRON_OK;
  } else if ((st = dev_upload(e, &g->Wt, Wt))) {
    return st;
  }
  return dev_upload(e, &g->shift, shift);
}

static chiron_status build_plans(chiron_engine* e, const float* w) {
  const chiron_model_desc& d = e->desc;
  const bool batch = d.bn_mode == CHIRON_BN_BATCH;
  e->bn_batch = batch;
  if (batch && (e->f16 || e->split)) return fail(CHIRON_ERR_INVALID, "bn_mode=batch is implemented for dtype f32 only");
  if (e->f16 || e->split) {
    for (int bi = 0; bi < d.n_blocks; ++bi)
      if (d.blocks[bi].out_channels % GEMM_BN || (d.blocks[bi].in_channels != 1 && d.blocks[bi].in_channels % 64))
        return fail(CHIRON_ERR_INVALID, "dtype f16: block %d has %d -> %d channels; the f16 kernels need multiples of 64 / 128", bi,
                    d.blocks[bi].in_channels, d.blocks[bi].out_channels);
  }
  int t = e->L;
  const float* p = w;
  if (d.stem_k > 0) {
    const int k = d.stem_k, co = d.stem_channels;
    const float* Ws = p;  // [k][1][co]
    p += (size_t)k * co;
    std::vector<float> wf((size_t)k * co), sh(co, 0.f);
    chiron_status st;
    if (batch) {
      std::vector<float> sc(p, p + co), of(p + co, p + 2 * co);
      for (size_t i = 0; i < wf.size(); ++i) wf[i] = Ws[i];
      if ((st = dev_upload(e, &e->stem_scale, sc))) return st;
      if ((st = dev_upload(e, &e->stem_offset, of))) return st;
    } else {
      const BnFold f = fold_bn(p, p + co, p + 2 * co, p + 3 * co, co);
      for (int tap = 0; tap < k; ++tap)
        for (int c = 0; c < co; ++c) wf[(size_t)tap * co + c] = Ws[(size_t)tap * co + c] * f.inv[c];
      sh = f.sh;
    }
    p += 4 * co;
    if ((st = dev_upload(e, &e->stem_w, wf))) return st;
    if ((st = dev_upload(e, &e->stem_shift, sh))) return st;
    e->stem_k = k;
    e->stem_stride = d.stem_stride;
    e->stem_c = co;
    same_pad(t, k, d.stem_stride, &e->stem_t, &e->stem_left);
    t = e->stem_t;
  }
  for (int bi = 0; bi < d.n_blocks; ++bi) {
    const chiron_res_block& b = d.blocks[bi];
    BlockPlan bp;
    bp.lift = b.in_channels == 1;
    bp.c_in = b.in_channels;
    bp.c = b.out_channels;
    bp.k = b.k;
    bp.stride = b.stride;
    bp.t_in = t;
    same_pad(t, b.k, b.stride, &bp.t_out, &bp.left);
    const int ci = b.in_channels, co = b.out_channels;
    chiron_status st;
    bp.i_bn = b.i_bn != 0;
    // one BN site: population statistics fold into the weights; batch statistics leave the weights raw and keep
    // scale / offset for bn_batch.hip
    auto bn_site = [&](int site, BnFold* f) -> chiron_status {
      if (batch) {
        f->inv.assign(co, 1.0f);
        f->sh.assign(co, 0.0f);
        std::vector<float> sc(p, p + co), of(p + co, p + 2 * co);
        chiron_status r = dev_upload(e, &bp.bn_scale[site], sc);
        if (r == CHIRON_OK) r = dev_upload(e, &bp.bn_offset[site], of);
        p += 4 * co;
        return r;
      }
      *f = fold_bn(p, p + co, p + 2 * co, p + 3 * co, co);
      p += 4 * co;
      return CHIRON_OK;
    };
    const float* W1 = p;
    p += (size_t)ci * co;
    BnFold f1;
    if (b.i_bn) {
      if ((st = bn_site(0, &f1))) return st;
    } else {
      f1.inv.assign(co, 1.0f);
      f1.sh.assign(co, 0.0f);
    }
    const float* W2a = p;
    p += (size_t)ci * co;
    BnFold f2a, f2b, f2c;
    if ((st = bn_site(1, &f2a))) return st;
    const float* W2b = p;
    p += (size_t)b.k * co * co;
    if ((st = bn_site(2, &f2b))) return st;
    const float* W2c = p;
    p += (size_t)co * co;
    if ((st = bn_site(3, &f2c))) return st;

    const int Npad = roundup(co, GEMM_BN);
    const int cop = roundup(co, e->kq);
    // conv2b: Wt[n][tap*cop + c] = W2b[tap][c][n] * inv2b[n]
    {
      const int K = b.k * cop;
      std::vector<float> Wt((size_t)Npad * K, 0.f), sh(Npad, 0.f);
      for (int n = 0; n < co; ++n) {
        for (int tap = 0; tap < b.k; ++tap)
          for (int c = 0; c < co; ++c) Wt[(size_t)n * K + tap * cop + c] = W2b[((size_t)tap * co + c) * co + n] * f2b.inv[n];
        sh[n] = f2b.sh[n];
      }
      if ((st = upload_gemm(e, &bp.gb, Wt, sh, co, Npad, K))) return st;
      // 1 x 3, stride 1 over C = co channels, fp32, population BN, even length: Winograd F(2,3) (wino.hip) -- four
      // products per output pair instead of six.  U_j[n][c] in float64 from the folded taps g_tap = W2b[tap][c][n]*inv[n].
      if (!bp.lift && !batch && !e->f16 && !e->split && b.k == 3 && b.stride == 1 && (t % 2) == 0 && co % 64 == 0 && ci == co &&
          getenv("CHIRON_NO_WINOGRAD") == nullptr) {
        // F(4,3) from 256 frames per window on (round 6): its rounding error is correlated over the four frames of a quad and, measured
        // against the float32 ensembles of tests/golden/parity_dist, costs the short strided topology more than it saves -- RNA_default
        // (T = 100): typical window 1.32 .. 1.45 -> 1.13 .. 1.20 x the ensemble's median, tail 7 .. 9 % -> 2 .. 5 % with F(2,3), for 0.04 ms
        // of its 1.7 ms batch; DNA_default (T = 400): parity within the noise of F(2,3)'s, F(4,3) worth 4.1 % of the headline.
        // CHIRON_WINOGRAD_F4=1 / CHIRON_WINOGRAD_F2=1 force either form.
        const bool f4 = (t % 4) == 0 && getenv("CHIRON_WINOGRAD_F2") == nullptr && (t >= 256 || getenv("CHIRON_WINOGRAD_F4") != nullptr);
        const int nu = f4 ? 6 : 4;
        // F(4,3): U = G g;  F(2,3): g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2
        static const double G4[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                        {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
        static const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        std::vector<float> U((size_t)nu * co * co);
        for (int n = 0; n < co; ++n)
          for (int c = 0; c < co; ++c) {
            double gt[3];
            for (int tap = 0; tap < 3; ++tap) gt[tap] = (double)W2b[((size_t)tap * co + c) * co + n] * f2b.inv[n];
            for (int j = 0; j < nu; ++j) {
              const double* gj = f4 ? G4[j] : G2[j];
              U[((size_t)j * co + n) * co + c] = (float)(gj[0] * gt[0] + gj[1] * gt[1] + gj[2] * gt[2]);
            }
          }
        bp.wino_f4 = f4 ? 1 : 0;
        if ((st = dev_upload(e, &bp.wino_u, U))) return st;
      }
    }
    if (bp.lift) {
      std::vector<float> la(cop, 0.f), lb(cop, 0.f), ra(Npad, 0.f), rb(Npad, 0.f);
      for (int c = 0; c < co; ++c) {
        la[c] = W2a[c] * f2a.inv[c];
        lb[c] = f2a.sh[c];
        ra[c] = W1[c] * f1.inv[c];
        rb[c] = e->f16 ? 0.f : f1.sh[c];   // the f16 engines keep round 4's arithmetic (shift folded): halves' rounding dominates there
      }
      if ((st = dev_upload(e, &bp.res_b, rb))) return st;
      if ((st = dev_upload(e, &bp.lift_a, la))) return st;
      if ((st = dev_upload(e, &bp.lift_b, lb))) return st;
      if ((st = dev_upload(e, &bp.res_a, ra))) return st;
      if (!batch && getenv("CHIRON_NO_PWL") == nullptr) {
        // f[tap][n](s) = sum_c W2b'[tap][c][n] * relu(s*a[c] + b[c]) is piecewise linear in the signal value s:
        // tabulate (alpha, beta) per interval between consecutive breakpoints -b[c]/a[c] (pwl.hip).  float64 sums.
        std::vector<std::pair<double, int>> brk;  // (breakpoint, channel)
        for (int c = 0; c < co; ++c)
          if (la[c] != 0.f) brk.emplace_back(-(double)lb[c] / (double)la[c], c);
        std::sort(brk.begin(), brk.end());
        const int nb = (int)brk.size(), kk = b.k;
        std::vector<double> al((size_t)kk * co, 0.0), be((size_t)kk * co, 0.0);
        auto toggle = [&](int c, double sign) {
          for (int tap = 0; tap < kk; ++tap)
            for (int n = 0; n < co; ++n) {
              const double wv = (double)(W2b[((size_t)tap * co + c) * co + n] * f2b.inv[n]) * sign;
              al[(size_t)tap * co + n] += wv * (double)la[c];
              be[(size_t)tap * co + n] += wv * (double)lb[c];
            }
        };
        // s -> -inf: a channel is active iff a < 0, or a == 0 and b > 0
        for (int c = 0; c < co; ++c)
          if (la[c] < 0.f || (la[c] == 0.f && lb[c] > 0.f)) toggle(c, 1.0);
        std::vector<float> tab((size_t)(nb + 1) * kk * co * 2), bpf(std::max(nb, 1), 0.f), reff(nb + 1, 0.f);
        for (int iv = 0; iv < nb; ++iv) bpf[iv] = (float)brk[iv].first;   // may round to +-inf: such a breakpoint is simply never crossed
        for (int iv = 0; iv <= nb; ++iv) {
          if (iv > 0) {
            const int c = brk[iv - 1].second;
            toggle(c, la[c] > 0.f ? 1.0 : -1.0);  // crossing its breakpoint upwards switches a channel on (a > 0) or off (a < 0)
          }
          // The table stores the slope and the value at a reference point of the interval, f = alpha*(s - ref) + f(ref).
          // ref = the point of the interval nearest to 0 (0 itself when the interval contains it): |s - ref| <= |s| for
          // every s the interval can receive, so a breakpoint far outside the signal range (a near-dead channel: tiny
          // folded scale, breakpoint at -1e6 or beyond float range) never makes alpha*(s - ref) cancel against f(ref).
          const double lower = iv > 0 ? (double)bpf[iv - 1] : -(double)INFINITY, upper = iv < nb ? (double)bpf[iv] : (double)INFINITY;
          double ref = std::min(std::max(0.0, lower), upper);
          ref = std::min(std::max(ref, -(double)FLT_MAX), (double)FLT_MAX);
          reff[iv] = (float)ref;
          for (size_t i = 0; i < (size_t)kk * co; ++i) {
            tab[((size_t)iv * kk * co + i) * 2] = (float)al[i];
            tab[((size_t)iv * kk * co + i) * 2 + 1] = (float)(al[i] * (double)reff[iv] + be[i]);
          }
        }
        std::vector<float> sh2(f2b.sh.begin(), f2b.sh.begin() + co);
        bp.pwl_nbp = nb;
        if ((st = dev_upload(e, &bp.pwl_bp, bpf))) return st;
        if ((st = dev_upload(e, &bp.pwl_ref, reff))) return st;
        if ((st = dev_upload(e, &bp.pwl_tab, tab))) return st;
        if ((st = dev_upload(e, &bp.pwl_shift, sh2))) return st;
      }
      // conv2c; the signal branch (scale res_a, shift res_b) is evaluated by the epilogue as one fmaf and added to the finished sum
      // (batch-statistics BN: both fold to 1 / 0 here and the branch is normalised by bn_batch.hip)
      const int K = cop;
      std::vector<float> Wt((size_t)Npad * K, 0.f), sh(Npad, 0.f);
      for (int n = 0; n < co; ++n) {
        for (int c = 0; c < co; ++c) Wt[(size_t)n * K + c] = W2c[(size_t)c * co + n] * f2c.inv[n];
        sh[n] = e->f16 ? f2c.sh[n] + f1.sh[n] : f2c.sh[n];
      }
      if ((st = upload_gemm(e, &bp.gc, Wt, sh, co, Npad, K))) return st;
    } else {
      const int cip = roundup(ci, e->kq);
      {
        const int K = cip;
        std::vector<float> Wt((size_t)Npad * K, 0.f), sh(Npad, 0.f);
        for (int n = 0; n < co; ++n) {
          for (int c = 0; c < ci; ++c) Wt[(size_t)n * K + c] = W2a[(size_t)c * co + n] * f2a.inv[n];
          sh[n] = f2a.sh[n];
        }
        if ((st = upload_gemm(e, &bp.ga, Wt, sh, co, Npad, K))) return st;
      }
      if (batch) {
        // separate GEMMs: each branch is normalised with its own batch statistics before the add
        {
          const int K = cop;
          std::vector<float> Wt((size_t)Npad * K, 0.f), sh(Npad, 0.f);
          for (int n = 0; n < co; ++n)
            for (int c = 0; c < co; ++c) Wt[(size_t)n * K + c] = W2c[(size_t)c * co + n];
          if ((st = upload_gemm(e, &bp.gc, Wt, sh, co, Npad, K))) return st;
        }
        {
          const int K = cip;
          std::vector<float> Wt((size_t)Npad * K, 0.f), sh(Npad, 0.f);
          for (int n = 0; n < co; ++n)
            for (int c = 0; c < ci; ++c) Wt[(size_t)n * K + c] = W1[(size_t)c * co + n];
          if ((st = upload_gemm(e, &bp.g1, Wt, sh, co, Npad, K))) return st;
        }
      } else {
        // conv2c and branch1/conv1 fused along K: [conv2b output | block input]
        const int K = cop + cip;
        std::vector<float> Wt((size_t)Npad * K, 0.f), sh(Npad, 0.f);
        for (int n = 0; n < co; ++n) {
          for (int c = 0; c < co; ++c) Wt[(size_t)n * K + c] = W2c[(size_t)c * co + n] * f2c.inv[n];
          for (int c = 0; c < ci; ++c) Wt[(size_t)n * K + cop + c] = W1[(size_t)c * co + n] * f1.inv[n];
          sh[n] = f2c.sh[n] + f1.sh[n];
        }
        if ((st = upload_gemm(e, &bp.gc, Wt, sh, co, Npad, K))) return st;
      }
    }
    e->blocks.push_back(bp);
    t = bp.t_out;
  }
  e->T = t;
  e->C = d.blocks[d.n_blocks - 1].out_channels;

  // ---- LSTM layers.  z column n of a direction: gate = n / H, unit = n % H (the order of the TF kernel's columns).
  const int H = d.hidden;
  const int zc = 4 * H;
  for (int l = 0; l < d.rnn_layers; ++l) {
    LstmPlan lp;
    lp.in_w = lstm_in_width(&d, l);
    const float* kern[2];
    const float* bias[2];
    for (int dir = 0; dir < 2; ++dir) {
      kern[dir] = p;
      e->lstm_kernel_off[dir].push_back((size_t)(p - w));
      p += (size_t)(lp.in_w + H) * 4 * H;
      bias[dir] = p;
      p += 4 * H;
    }
    const bool split = d.rnn_kind == CHIRON_RNN_MULTI && l > 0;
    lp.nproj = split ? 2 : 1;
    const int Kp = roundup(lp.in_w, e->kq);
    chiron_status st;
    for (int pj = 0; pj < lp.nproj; ++pj) {
      const int ndir = split ? 1 : 2;
      const int N = ndir * zc;
      const int Npad = std::max(roundup(N, GEMM_BN), roundup(N, 160));  // the DMA kernel reads whole 160-row weight tiles
      std::vector<float> Wt((size_t)Npad * Kp, 0.f), sh(Npad, 0.f);
      for (int n = 0; n < N; ++n) {
        const int dir = split ? pj : n / zc;
        const int nl = n % zc;
        const int g = nl / H, unit = nl % H;
        for (int k = 0; k < lp.in_w; ++k) Wt[(size_t)n * Kp + k] = kern[dir][(size_t)k * 4 * H + g * H + unit];
        // forget_bias = 1.0 (TF LSTMCell default; Add(+1.0) const in the .meta while-body) folded here
        sh[n] = bias[dir][g * H + unit] + (g == 2 ? 1.0f : 0.0f);
      }
      if ((st = upload_gemm(e, &lp.proj[pj], Wt, sh, N, Npad, Kp))) return st;
    }
    // recurrent weights in MFMA B-operand order: [dir][wave][k][lane], lane = gate*16 + (unit & 15)
    std::vector<float> wf((size_t)2 * LSTM_NW * LSTM_K * 64, 0.f);
    for (int dir = 0; dir < 2; ++dir)
      for (int wv = 0; wv < LSTM_NW; ++wv)
        for (int k = 0; k < LSTM_K; ++k)
          for (int lane = 0; lane < 64; ++lane) {
            const int g = lane >> 4, unit = 16 * wv + (lane & 15);
            float v = 0.f;
            if (k < H && unit < H) v = kern[dir][(size_t)(lp.in_w + k) * 4 * H + g * H + unit];
            wf[(((size_t)dir * LSTM_NW + wv) * LSTM_K + k) * 64 + lane] = v;
          }
    if (e->f16) {
      // v_mfma_f32_4x4x4_16B_f16 B-operand order: [dir][wave][k-step j][lane][4 halves], k = 4j .. 4j+3
      std::vector<_Float16> wh((size_t)2 * LSTM_NW * LSTM_KSTEPS16 * 64 * 4, (_Float16)0.f);
      for (int dir = 0; dir < 2; ++dir)
        for (int wv = 0; wv < LSTM_NW; ++wv)
          for (int j = 0; j < LSTM_KSTEPS16; ++j)
            for (int lane = 0; lane < 64; ++lane)
              for (int q = 0; q < 4; ++q) {
                const int k = 4 * j + q, g = lane >> 4, unit = 16 * wv + (lane & 15);
                if (k < H && unit < H)
                  wh[((((size_t)dir * LSTM_NW + wv) * LSTM_KSTEPS16 + j) * 64 + lane) * 4 + q] =
                      (_Float16)kern[dir][(size_t)(lp.in_w + k) * 4 * H + g * H + unit];
              }
      _Float16* d16 = nullptr;
      if ((st = dev_upload(e, &d16, wh))) return st;
      lp.wfrag = reinterpret_cast<float*>(d16);
      if (H == 100) {
        // lstm16w_kernel: [dir][wave 8][slot 4][k-step 7][lane][4 halves]; lane = kq*16 + 4u + gate, tile = 3 wave + slot
        std::vector<_Float16> ww((size_t)2 * 8 * 4 * 7 * 64 * 4, (_Float16)0.f);
        for (int dir = 0; dir < 2; ++dir)
          for (int wv = 0; wv < 8; ++wv)
            for (int slot = 0; slot < (wv == 7 ? 4 : 3); ++slot)
              for (int ks = 0; ks < 7; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                  for (int q = 0; q < 4; ++q) {
                    const int k = 16 * ks + 4 * (lane >> 4) + q, g = lane & 3, unit = 4 * (3 * wv + slot) + ((lane >> 2) & 3);
                    if (k < H && unit < H)
                      ww[(((((size_t)dir * 8 + wv) * 4 + slot) * 7 + ks) * 64 + lane) * 4 + q] =
                          (_Float16)kern[dir][(size_t)(lp.in_w + k) * 4 * H + g * H + unit];
                  }
        _Float16* dw = nullptr;
        if ((st = dev_upload(e, &dw, ww))) return st;
        lp.wwide = dw;
        if (lp.nproj == 1 && (lp.in_w == 256 || lp.in_w == 200)) {
          // lstm16f_kernel (v_mfma_f32_16x16x32_f16): W_x and W_hh as [dir][wave][slot][k-step of 32][lane][8 halves],
          // lane = kg*16 + 4u + gate -> k = 32 ks + 8 kg + e, column gate*H + 4 (3 wave + slot) + u; zero past the width
          const int ksx = lp.in_w == 256 ? 8 : 7;
          auto frag = [&](int ksteps, int k_off, int width) {
            std::vector<_Float16> v((size_t)2 * 8 * 4 * ksteps * 64 * 8, (_Float16)0.f);
            for (int dir = 0; dir < 2; ++dir)
              for (int wv = 0; wv < 8; ++wv)
                for (int slot = 0; slot < (wv == 7 ? 4 : 3); ++slot)
                  for (int ks = 0; ks < ksteps; ++ks)
                    for (int lane = 0; lane < 64; ++lane)
                      for (int q = 0; q < 8; ++q) {
                        const int k = 32 * ks + 8 * (lane >> 4) + q, g = lane & 3, unit = 4 * (3 * wv + slot) + ((lane >> 2) & 3);
                        if (k < width && unit < H)
                          v[(((((size_t)dir * 8 + wv) * 4 + slot) * ksteps + ks) * 64 + lane) * 8 + q] =
                              (_Float16)kern[dir][(size_t)(k_off + k) * 4 * H + g * H + unit];
                      }
            return v;
          };
          _Float16 *dx = nullptr, *dh = nullptr;
          if ((st = dev_upload(e, &dx, frag(ksx, 0, lp.in_w)))) return st;
          if ((st = dev_upload(e, &dh, frag(4, lp.in_w, H)))) return st;
          lp.wxwide = dx;
          lp.whfused = dh;
          lp.wx_ksteps = ksx;
        }
      }
    } else if ((st = dev_upload(e, &lp.wfrag, wf))) {
      return st;
    }
    if (e->w2 && H != 100) return fail(CHIRON_ERR_INVALID, "dtype f16-w2: the recurrence kernel is built for hidden=100");
    if ((e->w2 || (e->split && getenv("CHIRON_SPLIT_REC32") == nullptr)) && H == 100) {
      // lstm32s_kernel: [hi | lo][dir][wave 8][slot 4][k-step 7][lane][4 halves]; lane = kq*16 + 4u + gate, tile = 3 wave + slot (the order
      // of lstm16w_kernel's fragments), every weight as an exact hi + lo half pair
      const size_t half = (size_t)2 * 8 * 4 * 7 * 64 * 4;
      std::vector<_Float16> ws(2 * half, (_Float16)0.f);
      for (int dir = 0; dir < 2; ++dir)
        for (int wv = 0; wv < 8; ++wv)
          for (int slot = 0; slot < (wv == 7 ? 4 : 3); ++slot)
            for (int ks = 0; ks < 7; ++ks)
              for (int lane = 0; lane < 64; ++lane)
                for (int q = 0; q < 4; ++q) {
                  const int k = 16 * ks + 4 * (lane >> 4) + q, g = lane & 3, unit = 4 * (3 * wv + slot) + ((lane >> 2) & 3);
                  if (k < H && unit < H) {
                    const float wv32 = kern[dir][(size_t)(lp.in_w + k) * 4 * H + g * H + unit];
                    const _Float16 hi = (_Float16)wv32;
                    const size_t at = (((((size_t)dir * 8 + wv) * 4 + slot) * 7 + ks) * 64 + lane) * 4 + q;
                    ws[at] = hi;
                    ws[half + at] = (_Float16)(wv32 - (float)hi);
                  }
                }
      _Float16* dws = nullptr;
      if ((st = dev_upload(e, &dws, ws))) return st;
      lp.wsplit = dws;
    }
    if (!e->f16) {
      // light-wave fragment of lstm_pair_kernel: [dir][m = 4q + a][lane = kg*16 + gate*4 + j] = W_hh[16q + 4kg + a][gate*H + 96 + j]
      std::vector<float> wl((size_t)2 * 28 * 64, 0.f);
      for (int dir = 0; dir < 2; ++dir)
        for (int m = 0; m < 28; ++m)
          for (int lane = 0; lane < 64; ++lane) {
            const int q = m >> 2, a = m & 3, kg = lane >> 4, g = (lane >> 2) & 3, j = lane & 3;
            const int k = 16 * q + 4 * kg + a;
            if (k < H) wl[((size_t)dir * 28 + m) * 64 + lane] = kern[dir][(size_t)(lp.in_w + k) * 4 * H + g * H + 96 + j];
          }
      if ((st = dev_upload(e, &lp.wlight, wl))) return st;
      if (H == 100) {
        // lstm32w_kernel: [dir][wave 8][slot 4][k-step 25][lane]; lane = kq*16 + 4u + gate, tile = 3 wave + slot
        std::vector<float> ww((size_t)2 * 8 * 4 * 25 * 64, 0.f);
        for (int dir = 0; dir < 2; ++dir)
          for (int wv = 0; wv < 8; ++wv)
            for (int slot = 0; slot < (wv == 7 ? 4 : 3); ++slot)
              for (int ks = 0; ks < 25; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                  const int k = 4 * ks + (lane >> 4), g = lane & 3, unit = 4 * (3 * wv + slot) + ((lane >> 2) & 3);
                  ww[((((size_t)dir * 8 + wv) * 4 + slot) * 25 + ks) * 64 + lane] = kern[dir][(size_t)(lp.in_w + k) * 4 * H + g * H + unit];
                }
        if ((st = dev_upload(e, &lp.wwide32, ww))) return st;
      }
    }
    e->lstm.push_back(lp);
  }
  // ---- FC head (raw)
  {
    chiron_status st;
    std::vector<float> a(p, p + 2 * H);
    p += 2 * H;
    std::vector<float> b(p, p + H);
    p += H;
    std::vector<float> c(p, p + (size_t)H * d.classes);
    p += (size_t)H * d.classes;
    std::vector<float> dd(p, p + d.classes);
    p += d.classes;
    if ((st = dev_upload(e, &e->fc_w, a))) return st;
    if ((st = dev_upload(e, &e->fc_b, b))) return st;
    if ((st = dev_upload(e, &e->fc_wc, c))) return st;
    if ((st = dev_upload(e, &e->fc_bc, dd))) return st;
  }
  return CHIRON_OK;
}

// ---- sizes of an engine, computable without a GPU: the frame count, the largest tensor a kernel addresses with a
// 32-bit byte offset, the device memory of a slot.  chiron_engine_create refuses what this refuses.
// Limits (kernels.h / gemm.hip): the DMA GEMMs read their A operand through a raw-buffer descriptor with
// num_records = 0xFFFE0000 and 32-bit per-lane byte offsets, and the recurrence writes lasth through 32-bit byte
// offsets, so every activation tensor and every lasth tensor must stay below 0xFFFE0000 bytes; row counts are ints.
static const uint64_t TENSOR_LIMIT = 0xFFFE0000ull;

static chiron_status plan_sizes(const chiron_model_desc* d, const chiron_engine_opts* o, chiron_engine_sizes* out) {
  chiron_status st = validate_desc(d);
  if (st) return st;
  if (!o) return fail(CHIRON_ERR_INVALID, "null opts");
  if (o->max_batch < 1 || o->segment_len < 1) return fail(CHIRON_ERR_INVALID, "max_batch/segment_len must be positive");
  if (o->dtype != CHIRON_F32 && o->dtype != CHIRON_F16 && o->dtype != CHIRON_F32_SPLIT && o->dtype != CHIRON_F16_W2)
    return fail(CHIRON_ERR_INVALID, "dtype %d unknown (0 = f32, 1 = f16, 2 = f32 as hi/lo half pairs, 3 = f16 activations against hi/lo weights)", o->dtype);
  if (o->max_beam < 0) return fail(CHIRON_ERR_INVALID, "max_beam %d", o->max_beam);
  const bool f16 = o->dtype == CHIRON_F16 || o->dtype == CHIRON_F16_W2, split = o->dtype == CHIRON_F32_SPLIT, bn_batch = d->bn_mode == CHIRON_BN_BATCH;
  const uint64_t B = (uint64_t)o->max_batch, BP = (uint64_t)roundup(o->max_batch, 16), L = (uint64_t)o->segment_len, H = d->hidden, K = d->classes;
  int t = o->segment_len, left = 0;
  uint64_t tmax = 0, cmax = 0;
  if (d->stem_k > 0) same_pad(t, d->stem_k, d->stem_stride, &t, &left);
  for (int i = 0; i < d->n_blocks; ++i) {
    tmax = std::max<uint64_t>(tmax, (uint64_t)t);
    same_pad(t, d->blocks[i].k, d->blocks[i].stride, &t, &left);
    tmax = std::max<uint64_t>(tmax, (uint64_t)t);
    cmax = std::max<uint64_t>(cmax, (uint64_t)d->blocks[i].out_channels);
  }
  const uint64_t T = (uint64_t)t;
  const uint64_t lasth_ld = !split ? 2 * H : d->rnn_kind == CHIRON_RNN_MULTI ? 2 * (uint64_t)roundup(d->hidden, 32) : (uint64_t)roundup(2 * d->hidden, 32);
  const uint64_t act = B * tmax * cmax * (f16 ? 2 : 4), lasth = T * BP * lasth_ld * 4, z = T * BP * 2 * 4 * H * 4;
  if (B * tmax >= (1ull << 31) || T * BP >= (1ull << 31))
    return fail(CHIRON_ERR_OVERFLOW, "max_batch %d x %llu frames does not fit the kernels' int row index", o->max_batch, (unsigned long long)tmax);
  if (act > TENSOR_LIMIT)
    return fail(CHIRON_ERR_OVERFLOW, "max_batch %d: an activation tensor [%d x %llu x %llu] needs %llu bytes, the kernels address at most %llu "
                "per tensor (about %llu windows at this segment length and dtype): use a smaller max_batch",
                o->max_batch, o->max_batch, (unsigned long long)tmax, (unsigned long long)cmax, (unsigned long long)act,
                (unsigned long long)TENSOR_LIMIT, (unsigned long long)(TENSOR_LIMIT / (tmax * cmax * (f16 ? 2 : 4))));
  if (T * BP * lasth_ld * (f16 ? 2 : 4) > TENSOR_LIMIT)   // the f16 engine keeps lasth as halves (the allocation stays 4 bytes per element)
    return fail(CHIRON_ERR_OVERFLOW, "max_batch %d: the recurrent output [%llu x %llu x %llu] needs %llu bytes, the kernels address at most %llu per tensor",
                o->max_batch, (unsigned long long)T, (unsigned long long)BP, (unsigned long long)lasth_ld, (unsigned long long)lasth,
                (unsigned long long)TENSOR_LIMIT);
  uint64_t slot = B * L * 4 + BP * 4 + (bn_batch ? 5 : 3) * act + (bn_batch ? 2 * 2 * cmax * 8 : 0) + z + 2 * lasth + (split ? T * BP * 2 * H * 4 : 0);
  slot += B * T * K * 4 + B * T + 3 * B * 4 + (B + 1) * 8 + B * T * 16 + B * T * 8 + 3 * 8;
  if (o->max_beam > 0) slot += beam_workspace_bytes((int)B, (int)T, o->max_beam);
  if (out) {
    out->T = (int32_t)T;
    out->ratio = (double)o->segment_len / (double)T;
    out->largest_tensor_bytes = std::max(act, T * BP * lasth_ld * (f16 ? 2 : 4));
    out->tensor_limit_bytes = TENSOR_LIMIT;
    out->slot_bytes = slot;
    out->total_bytes = slot * (uint64_t)std::max(1, o->n_slots);
  }
  return CHIRON_OK;
}

extern "C" chiron_status chiron_engine_plan(const chiron_model_desc* desc, const chiron_engine_opts* opts, chiron_engine_sizes* out) {
  if (!out) return fail(CHIRON_ERR_INVALID, "null out");
  memset(out, 0, sizeof(*out));
  return plan_sizes(desc, opts, out);
}

static chiron_status alloc_slot(chiron_engine* e, Slot* s) {
  HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  const size_t B = e->maxB, BP = e->BP, L = e->L, T = e->T, H = e->H, K = e->K;
  size_t tmax = 0, cmax = 0;
  for (const BlockPlan& b : e->blocks) {
    tmax = std::max<size_t>(tmax, b.t_in);  // the lifted conv2a is materialised at input resolution
    tmax = std::max<size_t>(tmax, b.t_out);
    cmax = std::max<size_t>(cmax, b.c);
  }
  chiron_status st;
  if ((st = dev_alloc(e, (void**)&s->tile_ctr, 16 * sizeof(unsigned long long), true))) return st;
  if ((st = dev_alloc(e, (void**)&s->sig, B * L * 4, true))) return st;
  if ((st = dev_alloc(e, (void**)&s->seq, BP * 4, true))) return st;
  for (int i = 0; i < (e->bn_batch ? 5 : 3); ++i)
    if ((st = dev_alloc(e, (void**)&s->act[i], B * tmax * cmax * (e->f16 ? 2 : 4), false))) return st;
  if (e->bn_batch && (st = dev_alloc(e, (void**)&s->bn_sums, 2 * 2 * cmax * sizeof(double), true))) return st;
  if ((st = dev_alloc(e, (void**)&s->z, (size_t)T * BP * 2 * 4 * H * 4, true))) return st;
  for (int i = 0; i < 2; ++i)
    if ((st = dev_alloc(e, (void**)&s->lasth[i], T * BP * (size_t)e->lasth_ld * 4, true))) return st;
  if (e->split && (st = dev_alloc(e, (void**)&s->lasth_f32, T * BP * 2 * H * 4, true))) return st;
  if ((st = dev_alloc(e, (void**)&s->logits, B * T * K * 4, false))) return st;
  if ((st = dev_alloc(e, (void**)&s->labels, B * T, false))) return st;
  if ((st = dev_alloc(e, (void**)&s->count, B * 4, true))) return st;
  if ((st = dev_alloc(e, (void**)&s->log_prob, B * 4, true))) return st;
  if ((st = dev_alloc(e, (void**)&s->prob, B * 4, true))) return st;
  if ((st = dev_alloc(e, (void**)&s->offsets, (B + 1) * 8, true))) return st;
  if ((st = dev_alloc(e, (void**)&s->indices, B * T * 16, false))) return st;
  if ((st = dev_alloc(e, (void**)&s->values, B * T * 8, false))) return st;
  if ((st = dev_alloc(e, (void**)&s->meta, 3 * 8, true))) return st;
  if (e->opts.max_beam > 0) {
    s->beam_ws_bytes = beam_workspace_bytes((int)B, (int)T, e->opts.max_beam);
    if ((st = dev_alloc(e, &s->beam_ws, s->beam_ws_bytes, false))) return st;
  }
  HIP_TRY(hipHostMalloc((void**)&s->h_sig, B * L * 4, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&s->h_seq, BP * 4, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&s->h_indices, B * T * 16, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&s->h_values, B * T * 8, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&s->h_meta, 3 * 8, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&s->h_log_prob, B * 4, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&s->h_prob, B * 4, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&s->h_logits, B * T * K * 4, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&s->h_labels, B * T, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&s->h_count, B * 4, hipHostMallocDefault));
  s->h_flat = (uint8_t*)malloc(B * T ? B * T : 1);
  if (!s->h_flat) return fail(CHIRON_ERR_DEVICE, "out of host memory");
  memset(s->h_prob, 0, B * 4);
  return CHIRON_OK;
}

extern "C" chiron_status chiron_engine_create(const chiron_model_desc* desc, const float* weights, size_t n_floats,
                                              const chiron_engine_opts* opts, chiron_engine** out) {
  if (!out) return fail(CHIRON_ERR_INVALID, "null out");
  *out = nullptr;
  chiron_status st = validate_desc(desc);
  if (st) return st;
  if (!weights || !opts) return fail(CHIRON_ERR_INVALID, "null weights/opts");
  size_t want = 0;
  chiron_weights_size(desc, &want);
  if (want != n_floats) return fail(CHIRON_ERR_INVALID, "weight blob has %zu floats, descriptor needs %zu", n_floats, want);
  {
    chiron_engine_sizes sz;
    if ((st = plan_sizes(desc, opts, &sz))) return st;   // argument checks and the 32-bit addressing limits, before any GPU call
  }
  if (desc->hidden != 100) return fail(CHIRON_ERR_INVALID, "hidden=%d: the recurrence kernel is built for hidden=100 (both shipped models)", desc->hidden);
  // CHIRON_TRACE_CREATE=1: one line on stderr with where the time of this call went (tools/cold_start.py: the start-up budget of a
  // rank of `chiron call`; at 1250 reads per GPU the compute is ~3 s, so start-up decides the end-to-end figure of configs[3])
  const bool trace = getenv("CHIRON_TRACE_CREATE") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_enter = now();
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(CHIRON_ERR_DEVICE, "no HIP device visible: libchiron_amd has no CPU fallback");
  if (opts->device_id < 0 || opts->device_id >= ndev) return fail(CHIRON_ERR_INVALID, "device_id %d out of range (%d devices)", opts->device_id, ndev);
  HIP_TRY(hipSetDevice(opts->device_id));
  if (trace) HIP_TRY(hipFree(nullptr));     // forces the context (and the code objects of this library) onto the device now, so that the stage below is not charged for it
  const double t_runtime = now();

  chiron_engine* e = new chiron_engine();
  e->desc = *desc;
  e->opts = *opts;
  if (e->opts.n_slots < 1) e->opts.n_slots = 1;
  e->L = opts->segment_len;
  e->H = desc->hidden;
  e->K = desc->classes;
  e->maxB = opts->max_batch;
  e->BP = roundup(opts->max_batch, 16);
  e->w2 = opts->dtype == CHIRON_F16_W2;
  e->w2_zf16 = e->w2 && getenv("CHIRON_W2_ZF16") != nullptr;
  e->lasth32 = getenv("CHIRON_F16_LASTH16") == nullptr;
  e->f16 = opts->dtype == CHIRON_F16 || e->w2;
  e->split = opts->dtype == CHIRON_F32_SPLIT;
  e->kq = e->f16 ? 2 * GEMM_BK : GEMM_BK;
  // split: rows hold whole 32-element blocks; MultiRNN reads each direction as its own K-segment, so the backward half
  // starts on a block boundary (fw at 0, bw at roundup(H, 32))
  e->lasth_ld = !e->split ? 2 * desc->hidden
                : desc->rnn_kind == CHIRON_RNN_MULTI ? 2 * roundup(desc->hidden, 32) : roundup(2 * desc->hidden, 32);
  // Tuning knob, off by default; results do not depend on it (same arithmetic per row).  Paired workgroups are 14 %
  // faster per resident round, but they fill their CU: with several batches in flight the 7-wave form, which shares a CU
  // with another slot's GEMM workgroups, gives the higher throughput (3905 vs 4004 kbases/s on one box).
  e->lstm_paired = getenv("CHIRON_LSTM_PAIR") != nullptr;
  e->lstm_fixed_roles = getenv("CHIRON_LSTM_FIXED_ROLES") != nullptr;
  // fp32 recurrence form: CHIRON_LSTM_WIDE=1 -> lstm32w_kernel (16 rows per workgroup), =0 -> lstm_kernel (4 rows); read once here
  {
    const char* wv = getenv("CHIRON_LSTM_WIDE");
    // (fp32-split: the GEMMs take a third of their fp32 time, the recurrence dominates and nothing fills the CUs a wide
    // launch leaves alone -- the 4-row form, 1.10 against 1.29 ms per launch, is the faster one there)
    e->lstm_form = wv ? atoi(wv) : (e->split ? 0 : CHIRON_LSTM_WIDE_DEFAULT);
    if (e->lstm_form < 0 || e->lstm_form > 2) e->lstm_form = CHIRON_LSTM_WIDE_DEFAULT;
  }
  e->lstm16_narrow = getenv("CHIRON_LSTM16_NARROW") != nullptr;
  e->lstm16_pair = getenv("CHIRON_LSTM16_PAIR") ? atoi(getenv("CHIRON_LSTM16_PAIR")) : -1;
  e->stream16 = getenv("CHIRON_NO_STREAM16") == nullptr && opts->dtype != CHIRON_F16_W2;   // (the streaming kernels hold ONE half per weight in registers)
  e->stream32 = getenv("CHIRON_NO_STREAM32") == nullptr;
  e->dyn_tiles = getenv("CHIRON_STATIC_TILES") == nullptr;   // A/B switch: fixed tile shares per workgroup
  {
    // f16 engines run the x-projection inside the recurrence (lstm16f_kernel) from 64 sixteen-row workgroups up (B >= 512:
    // a workgroup alone on its CU takes 0.69 ms for T = 400 whatever the batch, the projection GEMMs + the 4-row recurrence
    // take 0.66 ms at B = 512, 1.18 at 1100, 1.9 at 2048; measured per batch 2.04 vs 2.18 / 3.36 vs 4.12 / 5.33 vs 6.84 ms).
    // CHIRON_LSTM16_UNFUSED=1 keeps the projection GEMM + z (A/B switch), CHIRON_LSTM16_FUSED_MIN=<workgroups> moves the threshold.
    const int min_groups = getenv("CHIRON_LSTM16_FUSED_MIN") ? atoi(getenv("CHIRON_LSTM16_FUSED_MIN")) : 64;
    e->lstm16_fused = e->f16 && !e->w2 && !e->lstm16_narrow && desc->hidden == 100 && (e->BP / 16) * 2 >= min_groups && getenv("CHIRON_LSTM16_UNFUSED") == nullptr;
  }
  if (e->f16 && !e->w2) e->host_weights.assign(weights, weights + n_floats);
  st = build_plans(e, weights);
  const double t_plans = now();
  if (st == CHIRON_OK) {
    e->slots.resize(e->opts.n_slots);
    for (Slot& s : e->slots)
      if ((st = alloc_slot(e, &s))) break;
  }
  const double t_slots = now();
  if (st == CHIRON_OK && hipDeviceSynchronize() != hipSuccess) st = fail(CHIRON_ERR_DEVICE, "device sync after setup failed");
  if (trace)
    fprintf(stderr, "chiron_engine_create: {\"hip_runtime_and_context_ms\": %.1f, \"weights_prepared_and_uploaded_ms\": %.1f, \"slot_buffers_ms\": %.1f, "
                    "\"final_sync_ms\": %.1f, \"slots\": %d}\n", t_runtime - t_enter, t_plans - t_runtime, t_slots - t_plans, now() - t_slots, e->opts.n_slots);
  if (st) {
    chiron_engine_destroy(e);
    return st;
  }
  e->prof_names.assign(std::begin(kProfNames), std::end(kProfNames));
  *out = e;
  return CHIRON_OK;
}

extern "C" void chiron_engine_destroy(chiron_engine* e) {
  if (!e) return;
  hipSetDevice(e->opts.device_id);
  hipDeviceSynchronize();
  for (Slot& s : e->slots) {
    for (ProfEvent& ev : s.events) {
      hipEventDestroy(ev.a);
      hipEventDestroy(ev.b);
    }
    if (s.stream) hipStreamDestroy(s.stream);
    void* hp[] = {s.h_sig, s.h_seq, s.h_indices, s.h_values, s.h_meta, s.h_log_prob, s.h_prob, s.h_logits, s.h_labels, s.h_count};
    free(s.h_flat);
    for (void* q : hp)
      if (q) hipHostFree(q);
  }
  for (void* q : e->owned) hipFree(q);
  delete e;
}

extern "C" chiron_status chiron_engine_dims(const chiron_engine* e, int32_t* out_T, double* out_ratio) {
  if (!e) return fail(CHIRON_ERR_INVALID, "null engine");
  if (out_T) *out_T = e->T;
  if (out_ratio) *out_ratio = (double)e->L / (double)e->T;  // chiron_model.py:151-152 (true division)
  return CHIRON_OK;
}

// ----------------------------------------------------------------------------------------------
// launch sequence
// ----------------------------------------------------------------------------------------------
enum { PN_CONV = 0, PN_PROJ, PN_REC, PN_FC, PN_GREEDY, PN_BEAM, PN_SPARSE, PN_PATHPROB, PN_LIFT, PN_RES, PN_PWL, PN_WINO, PN_PROJ0, PN_CONV2A };

// roctx ranges around every stage's launches (CHIRON_ROCTX=1): a `rocprofv3 --kernel-trace --marker-trace` run then names the stages
// by themselves -- "conv_wino", "lstm_recurrence", ... -- instead of through bench.py's bucket -> kernel symbol table.  The marker
// library is looked up at run time (librocprofiler-sdk-roctx.so, else libroctx64.so): the engine links only the HIP runtime, and
// without the switch, or without the library, nothing is called.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* v = getenv("CHIRON_ROCTX");
    if (!v || !*v || *v == '0') return;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (push && pop) return;
      push = nullptr, pop = nullptr;
    }
    fprintf(stderr, "chiron_amd: CHIRON_ROCTX is set but no roctx library could be loaded: no ranges\n");
  }
};
static const Roctx& roctx() {
  static const Roctx r;
  return r;
}

struct Prof {
  chiron_engine* e;
  Slot* s;
  ProfEvent ev;
  bool on;
  bool range;
  Prof(chiron_engine* e_, Slot* s_, int name_id, double flops, double bytes) : e(e_), s(s_), on(e_->profiling), range(roctx().push != nullptr) {
    if (range) roctx().push(kProfNames[name_id]);
    if (!on) return;
    ev.name_id = name_id;
    ev.flops = flops;
    ev.bytes = bytes;
    hipEventCreate(&ev.a);
    hipEventCreate(&ev.b);
    hipEventRecord(ev.a, s->stream);
  }
  ~Prof() {
    if (on) {
      hipEventRecord(ev.b, s->stream);
      s->events.push_back(ev);
    }
    if (range) roctx().pop();
  }
};

static void init_gemm(GemmParams* g, const chiron_engine* e, const ConvGemmPlan& w, int B) {
  memset(g, 0, sizeof(*g));
  g->B = B;
  g->BP = e->BP;
  g->N = w.N;
  g->K = w.K;
  g->Wt = w.Wt;
  g->shift = w.shift;
  g->descale = w.descale;
  g->z_dirs_total = 2;
}

// Segments are filled in ELEMENTS; the f16 kernels address in 4-byte units (see GemmParams::f16).
// Calibration pass of the f16 engine: before a convolution GEMM runs, the column sums of every K-segment's source.
static void calib_measure(chiron_engine* e, const float* shift, int k0, const void* src, long rows, int ld, int col0, int cin, int BP, int B,
                          int lstm_layer, int lstm_dir, int lstm_part, hipStream_t stream) {
  CalibCtx* c = e->calib;
  if (!c) return;
  if (c->rec.size() >= c->max_records || cin > 256) {   // a partially applied correction must not pass silently (advisor, round 4)
    ++c->skipped;
    return;
  }
  CalibRecord r{shift, k0, cin, lstm_layer, lstm_dir, lstm_part, BP > 0 ? (double)(rows / BP) * B : (double)rows, c->rec.size()};
  launch_colsum_f16(src, rows, ld, col0, cin, BP, B, c->dev_sums + r.slot * 256, stream);
  c->rec.push_back(r);
}

static bool launch(chiron_engine* e, GemmParams& g, hipStream_t stream) {
  if (e->calib && e->f16 && g.out_mode == 0) {       // (the LSTM projections are measured in run_rnn, fused or not)
    int k0 = 0;
    for (int i = 0; i < g.nseg; ++i) {
      const GemmSeg& sg = g.seg[i];
      if (sg.src) calib_measure(e, g.shift, k0, sg.src, (long)g.B * sg.w_in, sg.lda, sg.col0, sg.cin, 0, g.B, -1, 0, 0, stream);
      k0 += sg.kpad;
    }
  }
  Slot* own = nullptr;   // the slot this stream belongs to: its tile counters (dynamic tile scheduling)
  if (e->dyn_tiles)
    for (Slot& sl : e->slots)
      if (sl.stream == stream) own = &sl;
  if (!e->f16 && !e->split && e->stream32) {   // fp32 256 -> 256 channel 1 x 1 convolutions
    if (own) g.tile_ctr = own->tile_ctr + 8, g.tile_base_host = &own->tile_base32;
    if (launch_stream32(g, stream)) return true;
  }
  g.tile_ctr = own ? own->tile_ctr : nullptr;
  g.tile_base_host = own ? &own->tile_base : nullptr;
  if (e->split) g.f16 = 2;  // same 4-byte element units as fp32; only the content of the 128-byte blocks differs
  if (e->f16) {
    if (e->stream16 && launch_stream16(g, stream)) return true;   // 256 -> 256 channel 1 x 1 convolutions: streaming kernel
    g.f16 = 1;
    for (int i = 0; i < g.nseg; ++i) {
      g.seg[i].lda /= 2;
      g.seg[i].col0 /= 2;
      g.seg[i].cin /= 2;
      g.seg[i].kpad /= 2;
    }
    g.K /= 2;
    if (g.out_mode == 0) g.ldo /= 2;
    if (e->w2) {   // [x, x] . [W_hi; W_lo]: the same A segments again, against the lo columns of the weight rows (upload_gemm)
      if (2 * g.nseg > GEMM_MAX_SEG) return false;
      for (int i = 0; i < g.nseg; ++i) g.seg[g.nseg + i] = g.seg[i];
      g.nseg *= 2;
      g.K *= 2;
    }
  }
  return launch_gemm(g, stream);
}

// bn_mode = batch: every BN site is raw conv -> batch moments -> normalise (cnn.py:166-188), nothing folds.
static bool run_cnn_batch_bn(chiron_engine* e, Slot* s, int B, const float* sig) {
  bool ok = true;
  int xi = -1;
  const float* x = nullptr;
  if (e->stem_k > 0) {
    const long M0 = (long)B * e->stem_t;
    launch_stem_conv(sig, e->stem_w, e->stem_shift, s->act[4], B, e->L, e->stem_t, e->stem_k, e->stem_stride, e->stem_left, e->stem_c, 0, 0,
                     s->stream);
    launch_bn_stats(s->act[4], M0, e->stem_c, s->bn_sums, s->stream);
    launch_bn_apply(s->act[4], s->bn_sums, e->stem_scale, e->stem_offset, M0, e->stem_c, 1, nullptr, nullptr, nullptr, nullptr, s->stream);
    x = s->act[4];
    xi = 4;
  }
  for (const BlockPlan& b : e->blocks) {
    float* buf[4];
    for (int i = 0, j = 0; i < 5 && j < 4; ++i)
      if (i != xi) buf[j++] = s->act[i];
    float *A = buf[0], *Bf = buf[1], *Cf = buf[2], *D = buf[3];
    const int cop = roundup(b.c, e->kq), C = b.c;
    const long Min = (long)B * b.t_in, Mout = (long)B * b.t_out;
    double* s0 = s->bn_sums;
    double* s1 = s->bn_sums + 2 * C;
    GemmParams g;
    Prof pr(e, s, PN_CONV, 2.0 * ((double)Min * b.c_in * C + (double)Mout * (b.k * C + C + b.c_in) * C), 4.0 * (Min + 3.0 * Mout) * C * 3);
    // conv2a + BN + ReLU
    if (b.lift) {
      launch_rank1_conv(sig, b.lift_a, A, Min, b.t_in, e->L, 1, C, s->stream);
    } else {
      init_gemm(&g, e, b.ga, B);
      g.M = (int)Min;
      g.T_out = b.t_in;
      g.nseg = 1;
      g.seg[0] = GemmSeg{x, b.c_in, 0, b.c_in, roundup(b.c_in, e->kq), b.t_in, 1, 0, 0};
      g.out = A;
      g.ldo = C;
      ok &= launch(e, g, s->stream);
    }
    launch_bn_stats(A, Min, C, s0, s->stream);
    launch_bn_apply(A, s0, b.bn_scale[1], b.bn_offset[1], Min, C, 1, nullptr, nullptr, nullptr, nullptr, s->stream);
    // conv2b + BN + ReLU
    init_gemm(&g, e, b.gb, B);
    g.M = (int)Mout;
    g.T_out = b.t_out;
    g.nseg = b.k;
    for (int j = 0; j < b.k; ++j) g.seg[j] = GemmSeg{A, C, 0, C, cop, b.t_in, b.stride, j - b.left, 0};
    g.out = Bf;
    g.ldo = C;
    ok &= launch(e, g, s->stream);
    launch_bn_stats(Bf, Mout, C, s0, s->stream);
    launch_bn_apply(Bf, s0, b.bn_scale[2], b.bn_offset[2], Mout, C, 1, nullptr, nullptr, nullptr, nullptr, s->stream);
    // conv2c (+ BN), branch1 conv1 (+ BN iff i_bn), add, ReLU
    init_gemm(&g, e, b.gc, B);
    g.M = (int)Mout;
    g.T_out = b.t_out;
    g.nseg = 1;
    g.seg[0] = GemmSeg{Bf, C, 0, C, cop, b.t_out, 1, 0, 0};
    g.out = Cf;
    g.ldo = C;
    ok &= launch(e, g, s->stream);
    launch_bn_stats(Cf, Mout, C, s0, s->stream);
    if (b.lift) {
      launch_rank1_conv(sig, b.res_a, D, Mout, b.t_out, e->L, b.stride, C, s->stream);
    } else {
      init_gemm(&g, e, b.g1, B);
      g.M = (int)Mout;
      g.T_out = b.t_out;
      g.nseg = 1;
      g.seg[0] = GemmSeg{x, b.c_in, 0, b.c_in, roundup(b.c_in, e->kq), b.t_in, b.stride, 0, 0};
      g.out = D;
      g.ldo = C;
      ok &= launch(e, g, s->stream);
    }
    if (b.i_bn) launch_bn_stats(D, Mout, C, s1, s->stream);
    launch_bn_apply(Cf, s0, b.bn_scale[3], b.bn_offset[3], Mout, C, 1, D, b.i_bn ? s1 : nullptr, b.bn_scale[0], b.bn_offset[0], s->stream);
    x = Cf;
    for (int i = 0; i < 5; ++i)
      if (s->act[i] == Cf) xi = i;
  }
  s->sig_used = x;
  return ok;
}

static bool run_cnn(chiron_engine* e, Slot* s, int B, const float* sig) {
  if (e->bn_batch) return run_cnn_batch_bn(e, s, B, sig);
  bool ok = true;
  float* x = nullptr;  // block input (channels-last [B*T][C])
  int xi = -1;         // which act buffer holds x
  if (e->stem_k > 0) {
    Prof pr(e, s, PN_LIFT, 2.0 * B * e->stem_t * (double)e->stem_k * e->stem_c, 4.0 * B * e->L + 4.0 * B * e->stem_t * e->stem_c);
    launch_stem_conv(sig, e->stem_w, e->stem_shift, s->act[2], B, e->L, e->stem_t, e->stem_k, e->stem_stride, e->stem_left, e->stem_c,
                     e->f16 ? 1 : e->split ? 2 : 0, 1, s->stream);
    x = s->act[2];
    xi = 2;
  }
  for (const BlockPlan& b : e->blocks) {
    const int cop = roundup(b.c, e->kq);
    // pick two scratch buffers different from x
    int ia = (xi + 1) % 3, ib = (xi + 2) % 3;
    if (xi < 0) {
      ia = 0;
      ib = 1;
    }
    float* bufA = s->act[ia];
    float* bufB = s->act[ib];
    GemmParams g;
    if (b.lift) {
      // conv2b over the lifted signal: A(m, tap*C + c) = relu(sig*a[c] + b[c])
      init_gemm(&g, e, b.gb, B);
      g.M = B * b.t_out;
      g.T_out = b.t_out;
      g.nseg = b.k;
      g.relu = 1;
      g.out = bufB;
      g.ldo = b.c;
      bool done_pwl = false;
      if (b.pwl_tab != nullptr) {
        // conv2a + conv2b in one memory-bound pass over a piecewise-linear table of the signal value (pwl.hip)
        PwlConvParams q;
        q.sig = sig, q.bp = b.pwl_bp, q.ref = b.pwl_ref, q.tab = reinterpret_cast<const float2*>(b.pwl_tab), q.shift = b.pwl_shift, q.out = bufB;
        q.B = B, q.L = e->L, q.T_out = b.t_out, q.k = b.k, q.stride = b.stride, q.left = b.left, q.C = b.c, q.nbp = b.pwl_nbp;
        q.fmt = e->f16 ? 1 : e->split ? 2 : 0;
        Prof pr(e, s, PN_PWL, 2.0 * B * b.t_out * (double)b.k * b.c, 4.0 * B * e->L + (e->f16 ? 2.0 : 4.0) * B * b.t_out * b.c);
        done_pwl = launch_pwl_conv(q, s->stream);
      }
      if (!done_pwl) {
        // conv2a of the signal is materialised (one HBM-bound pass), conv2b is then an ordinary DMA launch
        {
          Prof pr(e, s, PN_LIFT, 2.0 * B * b.t_in * b.c, 4.0 * B * b.t_in + (e->f16 ? 2.0 : 4.0) * B * b.t_in * b.c);
          launch_lift(sig, b.lift_a, b.lift_b, bufA, (long)B * b.t_in, b.c, e->f16 ? 1 : e->split ? 2 : 0, s->stream);
        }
        for (int j = 0; j < b.k; ++j) g.seg[j] = GemmSeg{bufA, b.c, 0, b.c, cop, b.t_in, b.stride, j - b.left, 0};
        Prof pr(e, s, PN_CONV, 2.0 * B * b.t_out * (double)b.k * b.c * b.c, (e->f16 ? 2.0 : 4.0) * B * (b.t_in + b.t_out) * b.c);
        ok &= launch(e, g, s->stream);
      }
      // conv2c + lifted branch1 + ReLU
      init_gemm(&g, e, b.gc, B);
      g.M = B * b.t_out;
      g.T_out = b.t_out;
      g.nseg = 1;
      g.seg[0] = GemmSeg{bufB, b.c, 0, b.c, cop, b.t_out, 1, 0, 0};
      g.relu = 1;
      g.sig = sig;
      g.L = e->L;
      g.res_a = b.res_a;
      g.res_b = b.res_b;
      g.res_stride = b.stride;
      g.out = bufA;
      g.ldo = b.c;
      {
        Prof pr(e, s, PN_RES, 2.0 * B * b.t_out * (double)b.c * b.c, 8.0 * B * b.t_out * b.c);
        ok &= launch(e, g, s->stream);
      }
      x = bufA;
      xi = ia;
    } else {
      const int cip = roundup(b.c_in, e->kq);
      // conv2a
      init_gemm(&g, e, b.ga, B);
      g.M = B * b.t_in;
      g.T_out = b.t_in;
      g.nseg = 1;
      g.seg[0] = GemmSeg{x, b.c_in, 0, b.c_in, cip, b.t_in, 1, 0, 0};
      g.relu = 1;
      g.out = bufA;
      g.ldo = b.c;
      {
        // conv2a (K = c_in) has its own bucket: on the fp32 engine it runs on the streaming kernel (stream32.hip)
        Prof pr(e, s, PN_CONV2A, 2.0 * B * b.t_in * (double)b.c_in * b.c, 4.0 * B * b.t_in * (b.c_in + b.c));
        ok &= launch(e, g, s->stream);
      }
      // conv2b
      init_gemm(&g, e, b.gb, B);
      g.M = B * b.t_out;
      g.T_out = b.t_out;
      g.nseg = b.k;
      for (int j = 0; j < b.k; ++j) g.seg[j] = GemmSeg{bufA, b.c, 0, b.c, cop, b.t_in, b.stride, j - b.left, 0};
      g.relu = 1;
      g.out = bufB;
      g.ldo = b.c;
      {
        bool done = false;
        if (b.wino_u != nullptr) {
          WinoParams wq;
          wq.src = bufA, wq.U = b.wino_u, wq.shift = b.gb.shift, wq.out = bufB;
          wq.B = B, wq.T = b.t_in, wq.C = b.c, wq.N = b.c, wq.lda = b.c, wq.ldo = b.c, wq.relu = 1, wq.f4 = b.wino_f4;
          wq.tile_ctr = e->dyn_tiles ? s->tile_ctr : nullptr, wq.tile_base = 0, wq.tile_base_host = &s->tile_base;
          // FLOPs of the convolution as the reference defines it (2 * 3 taps * C * C per position); 2/3 of them are executed
          Prof pr(e, s, PN_WINO, 2.0 * B * b.t_out * (double)b.k * b.c * b.c, 4.0 * B * (b.t_in + b.t_out) * b.c);
          done = launch_wino_conv3(wq, s->stream);
        }
        if (!done) {
          Prof pr(e, s, PN_CONV, 2.0 * B * b.t_out * (double)b.k * b.c * b.c, 4.0 * B * (b.t_in + b.t_out) * b.c);
          ok &= launch(e, g, s->stream);
        }
      }
      // conv2c + branch1/conv1 fused along K, + ReLU
      init_gemm(&g, e, b.gc, B);
      g.M = B * b.t_out;
      g.T_out = b.t_out;
      g.nseg = 2;
      g.seg[0] = GemmSeg{bufB, b.c, 0, b.c, cop, b.t_out, 1, 0, 0};
      g.seg[1] = GemmSeg{x, b.c_in, 0, b.c_in, cip, b.t_in, b.stride, 0, 0};
      g.relu = 1;
      g.out = bufA;
      g.ldo = b.c;
      {
        Prof pr(e, s, PN_CONV, 2.0 * B * b.t_out * (double)(b.c + b.c_in) * b.c, 4.0 * B * b.t_out * (2.0 * b.c + b.c_in));
        ok &= launch(e, g, s->stream);
      }
      x = bufA;
      xi = ia;
    }
  }
  s->sig_used = x;  // CNN feature [B*T][C]
  return ok;
}

static bool run_rnn(chiron_engine* e, Slot* s, int B) {
  bool ok = true;
  bool last_out_f32 = false;
  const float* fea = s->sig_used;
  const int T = e->T, H = e->H, BP = e->BP;
  const int zc = 4 * H;
  const float* prev = nullptr;
  for (size_t l = 0; l < e->lstm.size(); ++l) {
    const LstmPlan& lp = e->lstm[l];
    float* outbuf = s->lasth[l & 1];
    // f16, whole 16-row groups filling the CUs: the projection runs inside the recurrence (lstm16f_kernel), no z
    const bool fused = e->lstm16_fused && lp.wxwide != nullptr;
    for (int pj = 0; pj < (fused ? 0 : lp.nproj); ++pj) {
      GemmParams g;
      init_gemm(&g, e, lp.proj[pj], B);
      g.M = T * BP;
      g.T_out = T;
      g.m_time_major = 1;
      g.nseg = 1;
      const int Kp = roundup(lp.in_w, e->kq);
      if (l == 0)
        g.seg[0] = GemmSeg{fea, e->C, 0, e->C, Kp, T, 1, 0, 0};
      else if (lp.nproj == 1)
        g.seg[0] = GemmSeg{prev, e->lasth_ld, 0, 2 * H, Kp, T, 1, 0, 1};
      else
        g.seg[0] = GemmSeg{prev, e->lasth_ld, e->split ? pj * roundup(H, 32) : pj * H, H, Kp, T, 1, 0, 1};
      g.out = s->z;
      g.out_mode = 1;
      g.z_cols = zc;
      g.z_ndir = lp.nproj == 1 ? 2 : 1;
      g.z_dir0 = lp.nproj == 1 ? 0 : pj;
      g.z_seq_len = s->seq;
      g.z_f16 = (e->f16 && (!e->w2 || e->w2_zf16)) ? 1 : 0;
      const double ndir = lp.nproj == 1 ? 2.0 : 1.0;
      // layer 0 reads the CNN features (K = 256: its own kernel instantiation), the other layers the recurrent output
      Prof pr(e, s, l == 0 ? PN_PROJ0 : PN_PROJ, 2.0 * B * T * (double)lp.in_w * 4 * H * ndir, 4.0 * B * T * (lp.in_w + ndir * zc));
      ok &= launch(e, g, s->stream);
    }
    LstmParams r;
    r.z = s->z;
    r.wfrag = lp.wfrag;
    r.wlight = lp.wlight;
    r.wwide = lp.wwide;
    r.wwide32 = lp.wwide32;
    r.wsplit = lp.wsplit;
    r.form32 = e->lstm_form;
    r.narrow16 = e->lstm16_narrow ? 1 : 0;
    r.xsrc = nullptr;
    r.wxwide = nullptr;
    r.whfused = nullptr;
    r.fused_pair = 0;
    r.xbias = nullptr;
    r.xK = r.xld = r.x_time_major = 0;
    if (fused) {
      r.xsrc = l == 0 ? (const void*)fea : (const void*)prev;
      r.wxwide = lp.wxwide;
      r.whfused = lp.whfused;
      // one 8-wave workgroup per CU either way (registers): 16-row workgroups that need a second round (B = 4096: 512) take 2 x 0.66 ms,
      // 256 32-row workgroups 1.24 ms in one; below one round the 16-row form finishes in 0.66 (B = 2048)
      r.fused_pair = e->lstm16_pair >= 0 ? (e->lstm16_pair ? 1 : 0) : ((BP / 16) * 2 > current_device_cus() ? 1 : 0);
      r.xbias = lp.proj[0].shift;
      r.xK = lp.in_w;
      r.xld = l == 0 ? e->C : e->lasth_ld;
      r.x_time_major = l == 0 ? 0 : 1;
    }
    r.seq_len = s->seq;
    r.out = e->split ? s->lasth_f32 : outbuf;
    // f16 engines: the last layer's output goes to the FC head as fp32 (not while the calibration pass measures the layers' outputs as halves)
    r.out_f32 = (e->f16 && e->lasth32 && e->calib == nullptr && l + 1 == e->lstm.size()) ? 1 : 0;
    last_out_f32 = r.out_f32 != 0;
    // fp32-split with its own recurrence (lstm32s_kernel): a layer that feeds another projection writes the hi / lo format that
    // projection reads straight into the layer's lasth buffer -- no fp32 copy, no conversion pass; the last layer writes fp32 for the FC head
    const bool direct_split = e->split && lp.wsplit != nullptr && l + 1 < e->lstm.size();
    r.out_split = direct_split ? (void*)outbuf : nullptr;
    r.split_ld = e->lasth_ld;
    r.split_bw0 = e->desc.rnn_kind == CHIRON_RNN_MULTI ? roundup(H, 32) : H;
    r.T = T;
    r.B = B;
    r.BP = BP;
    r.H = H;
    r.ndir = 2;
    r.paired = e->lstm_paired ? 1 : 0;
    r.fixed_roles = e->lstm_fixed_roles ? 1 : 0;
    r.group0 = 0;
    r.f16 = e->f16 ? 1 : 0;
    r.w2 = e->w2 ? (e->w2_zf16 ? 2 : 1) : 0;
    if (e->calib && e->f16) {
      // x rows of the layer's kernels: the means of the layer's input (features: batch-major rows; lasth: time-major)
      for (int pj = 0; pj < lp.nproj; ++pj) {
        const int dir = lp.nproj == 1 ? -1 : pj;
        if (l == 0)
          calib_measure(e, lp.proj[pj].shift, 0, fea, (long)B * T, e->C, 0, e->C, 0, B, (int)l, dir, 0, s->stream);
        else
          calib_measure(e, lp.proj[pj].shift, 0, prev, (long)T * BP, e->lasth_ld, lp.nproj == 1 ? 0 : pj * H, lp.in_w, BP, B, (int)l, dir, 0, s->stream);
      }
    }
    {
      Prof pr(e, s, PN_REC, 2.0 * 2.0 * B * T * (double)H * 4 * H + (fused ? 2.0 * B * T * (double)lp.in_w * 4 * H * 2.0 : 0.0),
              fused ? (e->f16 ? 2.0 : 4.0) * B * T * (2.0 * lp.in_w + 2.0 * H) : 4.0 * B * T * 2.0 * (zc + H));
      launch_lstm(r, s->stream);
    }
    if (e->calib && e->f16)      // h rows: the means of the layer's own output, per direction
      for (int dir = 0; dir < 2; ++dir)
        calib_measure(e, lp.proj[lp.nproj == 1 ? 0 : dir].shift, 0, outbuf, (long)T * BP, e->lasth_ld, dir * H, H, BP, B, (int)l, dir, 1, s->stream);
    prev = outbuf;
    if (e->split) {
      if (direct_split) {
        // nothing to convert
      } else if (l + 1 < e->lstm.size()) {
        Prof pr(e, s, PN_REC, 0.0, (4.0 + 4.0) * T * BP * 2.0 * H);
        const bool multi = e->desc.rnn_kind == CHIRON_RNN_MULTI;
        launch_split_convert(s->lasth_f32, outbuf, (long)T * BP, 2 * H, e->lasth_ld, multi ? H : 0, multi ? roundup(H, 32) : 0, s->stream);
      } else {
        prev = s->lasth_f32;   // the FC head reads the last layer's fp32 output directly
      }
    }
  }
  s->rnn_out = prev;
  FcParams f;
  f.lasth = prev;
  f.w = e->fc_w;
  f.bias = e->fc_b;
  f.wc = e->fc_wc;
  f.bc = e->fc_bc;
  f.logits = s->logits;
  f.T = T;
  f.B = B;
  f.BP = BP;
  f.H = H;
  f.K = e->K;
  f.f16 = (e->f16 && !last_out_f32) ? 1 : 0;
  f.split = 0;
  f.ld = e->lasth_ld;
  s->rnn_out_f32 = !e->f16 || last_out_f32;
  {
    Prof pr(e, s, PN_FC, 2.0 * B * T * (2.0 * H + (double)H * e->K), 4.0 * B * T * (2.0 * H + e->K));
    launch_fc(f, s->stream);
  }
  return ok;
}

static chiron_status enqueue_decode(chiron_engine* e, Slot* s, int B, int beam_width, uint32_t flags) {
  const int T = e->T, K = e->K;
  if (beam_width == 0) {
    GreedyParams g;
    g.logits = s->logits;
    g.seq_len = s->seq;
    g.labels = s->labels;
    g.count = s->count;
    g.log_prob = s->log_prob;
    g.prob_logits = (flags & CHIRON_WANT_PROB) ? s->prob : nullptr;
    g.B = B;
    g.T = T;
    g.K = K;
    Prof pr(e, s, PN_GREEDY, 0, 4.0 * B * T * K);
    launch_greedy(g, s->stream);
  } else {
    if (flags & CHIRON_WANT_PROB) {
      PathProbParams pp{s->logits, s->prob, B, T, K};
      Prof pr(e, s, PN_PATHPROB, 0, 4.0 * B * T * K);
      launch_path_prob(pp, s->stream);
    }
    BeamParams bp;
    bp.logits = s->logits;
    bp.seq_len = s->seq;
    bp.labels = s->labels;
    bp.count = s->count;
    bp.log_prob = s->log_prob;
    bp.workspace = s->beam_ws;
    bp.workspace_bytes = s->beam_ws_bytes;
    bp.B = B;
    bp.T = T;
    bp.K = K;
    bp.beam = beam_width;
    Prof pr(e, s, PN_BEAM, 0, 4.0 * B * T * K);
    if (launch_beam(bp, s->stream) != 0) return fail(CHIRON_ERR_INVALID, "beam search launch rejected (beam_width %d)", beam_width);
  }
  {
    SparseParams sp{s->labels, s->count, s->offsets, s->indices, s->values, s->meta, B, T};
    Prof pr(e, s, PN_SPARSE, 0, 0);
    launch_sparse(sp, s->stream);
  }
  HIP_TRY(hipMemcpyAsync(s->h_meta, s->meta, 3 * 8, hipMemcpyDeviceToHost, s->stream));
  if (flags & CHIRON_COMPACT_DECODE) {
    // fixed-size copies, known at submit time: no second round trip in collect (the SparseTensor's size is only known once the
    // decode has run, so its copy is issued BY collect: sync, read nnz, copy nnz * 24 bytes, sync again)
    HIP_TRY(hipMemcpyAsync(s->h_labels, s->labels, (size_t)B * T, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(s->h_count, s->count, (size_t)B * 4, hipMemcpyDeviceToHost, s->stream));
  }
  HIP_TRY(hipMemcpyAsync(s->h_log_prob, s->log_prob, (size_t)B * 4, hipMemcpyDeviceToHost, s->stream));
  if (flags & CHIRON_WANT_PROB) HIP_TRY(hipMemcpyAsync(s->h_prob, s->prob, (size_t)B * 4, hipMemcpyDeviceToHost, s->stream));
  if (flags & CHIRON_WANT_LOGITS)
    HIP_TRY(hipMemcpyAsync(s->h_logits, s->logits, (size_t)B * T * K * 4, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipGetLastError());
  return CHIRON_OK;
}

// After a failed submit / collect the slot's stream is drained; a launch that never ran has not taken its numbers from the
// tile counters while the host side already counted them (GemmParams::tile_base), and every later workgroup would read
// "no tile left".  Both sides restart from zero.
static void resync_tile_counters(Slot* s) {
  hipStreamSynchronize(s->stream);
  if (s->tile_ctr && hipMemsetAsync(s->tile_ctr, 0, 16 * sizeof(unsigned long long), s->stream) == hipSuccess) hipStreamSynchronize(s->stream);
  s->tile_base = s->tile_base32 = 0;
  (void)hipGetLastError();
}

static chiron_status submit_impl(chiron_engine* e, int32_t slot, const float* x, const float* const* pieces, const int32_t* piece_rows,
                                 const int64_t* piece_row_stride, int32_t n_pieces, const int32_t* seq_len, int32_t batch, int32_t beam_width,
                                 uint32_t flags);

extern "C" chiron_status chiron_engine_submit(chiron_engine* e, int32_t slot, const float* x, const int32_t* seq_len,
                                              int32_t batch, int32_t beam_width, uint32_t flags) {
  if (!x) return fail(CHIRON_ERR_INVALID, "null x/seq_len");
  return submit_impl(e, slot, x, nullptr, nullptr, nullptr, 0, seq_len, batch, beam_width, flags);
}

extern "C" chiron_status chiron_engine_submit_pieces(chiron_engine* e, int32_t slot, const float* const* pieces, const int32_t* piece_rows,
                                                     const int64_t* piece_row_stride, int32_t n_pieces, const int32_t* seq_len, int32_t batch,
                                                     int32_t beam_width, uint32_t flags) {
  if (!pieces || !piece_rows || n_pieces < 1) return fail(CHIRON_ERR_INVALID, "null / empty piece list");
  if (flags & CHIRON_X_ON_DEVICE) return fail(CHIRON_ERR_INVALID, "chiron_engine_submit_pieces takes host pieces");
  long rows = 0;
  for (int i = 0; i < n_pieces; ++i) {
    if (!pieces[i] || piece_rows[i] < 0) return fail(CHIRON_ERR_INVALID, "piece %d: null pointer or negative row count", i);
    if (piece_row_stride && piece_row_stride[i] < 1) return fail(CHIRON_ERR_INVALID, "piece %d: row stride %lld", i, (long long)piece_row_stride[i]);
    rows += piece_rows[i];
  }
  if (rows != batch) return fail(CHIRON_ERR_INVALID, "the pieces hold %ld rows, batch is %d", rows, batch);
  return submit_impl(e, slot, nullptr, pieces, piece_rows, piece_row_stride, n_pieces, seq_len, batch, beam_width, flags);
}

static chiron_status submit_impl(chiron_engine* e, int32_t slot, const float* x, const float* const* pieces, const int32_t* piece_rows,
                                 const int64_t* piece_row_stride, int32_t n_pieces, const int32_t* seq_len, int32_t batch, int32_t beam_width,
                                 uint32_t flags) {
  if (!e) return fail(CHIRON_ERR_INVALID, "null engine");
  if (slot < 0 || slot >= (int)e->slots.size()) return fail(CHIRON_ERR_STATE, "slot %d out of range (%zu slots)", slot, e->slots.size());
  if (!seq_len) return fail(CHIRON_ERR_INVALID, "null x/seq_len");
  if (batch < 1 || batch > e->maxB) return fail(CHIRON_ERR_OVERFLOW, "batch %d exceeds max_batch %d", batch, e->maxB);
  if (beam_width < 0) return fail(CHIRON_ERR_INVALID, "beam_width %d", beam_width);
  if (beam_width > e->opts.max_beam) return fail(CHIRON_ERR_OVERFLOW, "beam_width %d exceeds max_beam %d given at create", beam_width, e->opts.max_beam);
  Slot* s = &e->slots[slot];
  if (s->state.v.load(std::memory_order_acquire) != 0)
    return fail(CHIRON_ERR_STATE, "slot %d still holds an uncollected batch: collect it before the next submit", slot);
  HIP_TRY(hipSetDevice(e->opts.device_id));
  const int B = batch;
  // Everything from here on puts work on the slot's stream that reads the slot's pinned staging buffers.  If any
  // step fails the stream is drained before returning, so a later submit can never overwrite h_sig / h_seq under a
  // copy that is still running; the slot stays idle.
  auto enqueue = [&]() -> chiron_status {
    const float* sig;
    HIP_TRY(hipMemsetAsync(s->seq, 0, e->BP * 4, s->stream));
    if (flags & CHIRON_X_ON_DEVICE) {
      sig = x;
      HIP_TRY(hipMemcpyAsync(s->seq, seq_len, (size_t)B * 4, hipMemcpyDeviceToDevice, s->stream));
    } else {
      if (x) {
        memcpy(s->h_sig, x, (size_t)B * e->L * 4);
      } else {   // cross-read packing (chiron_eval.py:321-334) straight into the staging buffer: the batch is never assembled anywhere else
        size_t row = 0;
        for (int i = 0; i < n_pieces; ++i) {
          const int64_t stride = piece_row_stride ? piece_row_stride[i] : e->L;
          if (stride == e->L) {
            memcpy(s->h_sig + row * e->L, pieces[i], (size_t)piece_rows[i] * e->L * 4);
          } else {   // windows of one signal buffer: row r starts r * jump samples in and overlaps its neighbour (chiron_input.py:276-286)
            for (int r = 0; r < piece_rows[i]; ++r) memcpy(s->h_sig + (row + r) * e->L, pieces[i] + (size_t)r * stride, (size_t)e->L * 4);
          }
          row += piece_rows[i];
        }
      }
      memcpy(s->h_seq, seq_len, (size_t)B * 4);
      HIP_TRY(hipMemcpyAsync(s->sig, s->h_sig, (size_t)B * e->L * 4, hipMemcpyHostToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(s->seq, s->h_seq, (size_t)B * 4, hipMemcpyHostToDevice, s->stream));
      sig = s->sig;
    }
    if (!run_cnn(e, s, B, sig) || !run_rnn(e, s, B))
      return fail(CHIRON_ERR_INVALID, "a GEMM of this topology has no kernel for the engine's dtype");
    chiron_status st = enqueue_decode(e, s, B, beam_width, flags);
    if (st) return st;
    HIP_TRY(hipGetLastError());
    return CHIRON_OK;
  };
  const chiron_status st = enqueue();
  if (st) {
    s->net_batch = 0;
    resync_tile_counters(s);
    return st;
  }
  s->batch = B;
  s->net_batch = B;
  s->flags = flags;
  s->state.v.store(1, std::memory_order_release);
  return CHIRON_OK;
}

// Decode-only entry: run the decode sub-graph (chiron_eval.py:465-492) on caller-supplied logits.
extern "C" chiron_status chiron_engine_decode(chiron_engine* e, int32_t slot, const float* logits, const int32_t* seq_len,
                                              int32_t batch, int32_t beam_width, uint32_t flags) {
  if (!e) return fail(CHIRON_ERR_INVALID, "null engine");
  if (slot < 0 || slot >= (int)e->slots.size()) return fail(CHIRON_ERR_STATE, "slot %d out of range (%zu slots)", slot, e->slots.size());
  if (!logits || !seq_len) return fail(CHIRON_ERR_INVALID, "null logits/seq_len");
  if (batch < 1 || batch > e->maxB) return fail(CHIRON_ERR_OVERFLOW, "batch %d exceeds max_batch %d", batch, e->maxB);
  if (beam_width < 0) return fail(CHIRON_ERR_INVALID, "beam_width %d", beam_width);
  if (beam_width > e->opts.max_beam) return fail(CHIRON_ERR_OVERFLOW, "beam_width %d exceeds max_beam %d given at create", beam_width, e->opts.max_beam);
  Slot* s = &e->slots[slot];
  if (s->state.v.load(std::memory_order_acquire) != 0)
    return fail(CHIRON_ERR_STATE, "slot %d still holds an uncollected batch: collect it before the next decode", slot);
  HIP_TRY(hipSetDevice(e->opts.device_id));
  const size_t nlog = (size_t)batch * e->T * e->K * 4;
  auto enqueue = [&]() -> chiron_status {
    HIP_TRY(hipMemsetAsync(s->seq, 0, e->BP * 4, s->stream));
    const hipMemcpyKind kind = (flags & CHIRON_X_ON_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const float* src = logits;
    const int32_t* sl = seq_len;
    if (!(flags & CHIRON_X_ON_DEVICE)) {
      memcpy(s->h_logits, logits, nlog);
      memcpy(s->h_seq, seq_len, (size_t)batch * 4);
      src = s->h_logits;
      sl = s->h_seq;
    }
    HIP_TRY(hipMemcpyAsync(s->logits, src, nlog, kind, s->stream));
    HIP_TRY(hipMemcpyAsync(s->seq, sl, (size_t)batch * 4, kind, s->stream));
    return enqueue_decode(e, s, batch, beam_width, flags & ~CHIRON_WANT_LOGITS);
  };
  const chiron_status st = enqueue();
  s->net_batch = 0;     // the slot's activations no longer belong to the batch it reports
  if (st) {
    hipStreamSynchronize(s->stream);
    return st;
  }
  s->batch = batch;
  s->flags = flags & ~CHIRON_WANT_LOGITS;
  s->state.v.store(1, std::memory_order_release);
  return CHIRON_OK;
}

extern "C" chiron_status chiron_engine_collect(chiron_engine* e, int32_t slot, chiron_decoded* out) {
  if (!e || !out) return fail(CHIRON_ERR_INVALID, "null engine/out");
  if (slot < 0 || slot >= (int)e->slots.size()) return fail(CHIRON_ERR_STATE, "slot %d out of range", slot);
  Slot* s = &e->slots[slot];
  if (s->state.v.load(std::memory_order_acquire) != 1) return fail(CHIRON_ERR_STATE, "collect on slot %d without a submitted batch", slot);
  HIP_TRY(hipSetDevice(e->opts.device_id));
  // A HIP error below loses the batch, not the slot: the stream is drained and the slot returns to idle, as a failed
  // submit leaves it (otherwise every later submit would be refused with "still holds an uncollected batch").
  auto drain = [&]() -> chiron_status {
    HIP_TRY(hipStreamSynchronize(s->stream));
    const int64_t n = s->h_meta[0];
    if (!(s->flags & (CHIRON_NO_DECODE_COPY | CHIRON_COMPACT_DECODE)) && n > 0) {
      HIP_TRY(hipMemcpyAsync(s->h_indices, s->indices, (size_t)n * 16, hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipMemcpyAsync(s->h_values, s->values, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return CHIRON_OK;
  };
  const chiron_status dst = drain();
  if (dst) {
    resync_tile_counters(s);
    s->state.v.store(0, std::memory_order_release);
    return dst;
  }
  const int64_t nnz = s->h_meta[0];
  if (!(s->flags & CHIRON_WANT_PROB)) memset(s->h_prob, 0, (size_t)s->batch * 4);
  out->nnz = nnz;
  const bool compact = (s->flags & CHIRON_COMPACT_DECODE) != 0;
  out->indices = compact ? nullptr : s->h_indices;
  out->values = compact ? nullptr : s->h_values;
  out->flat_labels = nullptr;
  out->row_counts = nullptr;
  if (compact) {
    // the per-read regroup of chiron_eval.py:403-446 needs, per run of rows, the rows' label strings back to back and their lengths:
    // exactly this, in row order, built here in one pass over what the decoder wrote (labels[b][0 .. count[b]))
    const int T = e->T;
    int64_t off = 0;
    for (int b = 0; b < s->batch; ++b) {
      const int c = s->h_count[b];
      if (c < 0 || c > T || off + c > nnz) {
        s->state.v.store(0, std::memory_order_release);
        return fail(CHIRON_ERR_DEVICE, "decoder wrote an impossible row length %d (row %d)", c, b);
      }
      memcpy(s->h_flat + off, s->h_labels + (size_t)b * T, (size_t)c);
      off += c;
    }
    if (off != nnz) {
      s->state.v.store(0, std::memory_order_release);
      return fail(CHIRON_ERR_DEVICE, "row lengths sum to %lld, the SparseTensor holds %lld", (long long)off, (long long)nnz);
    }
    out->flat_labels = s->h_flat;
    out->row_counts = s->h_count;
  }
  out->dense_shape[0] = s->h_meta[1];
  out->dense_shape[1] = s->h_meta[2];
  out->log_prob = s->h_log_prob;
  out->prob_logits = s->h_prob;
  out->logits = (s->flags & CHIRON_WANT_LOGITS) ? s->h_logits : nullptr;
  out->batch = s->batch;
  out->T = e->T;
  s->state.v.store(0, std::memory_order_release);
  return CHIRON_OK;
}

extern "C" chiron_status chiron_engine_sync(chiron_engine* e) {
  if (!e) return fail(CHIRON_ERR_INVALID, "null engine");
  HIP_TRY(hipSetDevice(e->opts.device_id));
  for (Slot& s : e->slots) HIP_TRY(hipStreamSynchronize(s.stream));
  return CHIRON_OK;
}

extern "C" chiron_status chiron_engine_device_results(chiron_engine* e, int32_t slot, const float** logits,
                                                      const int64_t** indices, const int64_t** values,
                                                      const int64_t** nnz_and_shape) {
  if (!e) return fail(CHIRON_ERR_INVALID, "null engine");
  if (slot < 0 || slot >= (int)e->slots.size()) return fail(CHIRON_ERR_STATE, "slot %d out of range", slot);
  Slot* s = &e->slots[slot];
  if (logits) *logits = s->logits;
  if (indices) *indices = s->indices;
  if (values) *values = s->values;
  if (nnz_and_shape) *nnz_and_shape = s->meta;
  return CHIRON_OK;
}

// getcnnfeature (cnn.py:334-371): the [batch, T, C] feature tensor the CNN handed to the recurrent layers for the batch
// most recently run on an idle slot.
extern "C" chiron_status chiron_engine_features(chiron_engine* e, int32_t slot, float* out, size_t cap_floats, int32_t* out_batch,
                                                int32_t* out_channels) {
  if (!e) return fail(CHIRON_ERR_INVALID, "null engine");
  if (slot < 0 || slot >= (int)e->slots.size()) return fail(CHIRON_ERR_STATE, "slot %d out of range", slot);
  Slot* s = &e->slots[slot];
  if (s->state.v.load(std::memory_order_acquire) != 0) return fail(CHIRON_ERR_STATE, "slot %d holds an uncollected batch", slot);
  if (s->net_batch < 1 || s->sig_used == nullptr)
    return fail(CHIRON_ERR_STATE, "slot %d holds no network batch (nothing submitted yet, or its last batch was decode-only)", slot);
  if (e->split) return fail(CHIRON_ERR_INVALID, "chiron_engine_features: dtype fp32-split keeps features as hi/lo half pairs; not exported");
  const size_t n = (size_t)s->net_batch * e->T * e->C;
  if (out_batch) *out_batch = s->net_batch;
  if (out_channels) *out_channels = e->C;
  if (!out || cap_floats < n) return fail(CHIRON_ERR_OVERFLOW, "features need %zu floats, capacity %zu", n, cap_floats);
  HIP_TRY(hipSetDevice(e->opts.device_id));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (e->f16) {
    std::vector<_Float16> h(n);
    HIP_TRY(hipMemcpy(h.data(), s->sig_used, n * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) out[i] = (float)h[i];
  } else {
    HIP_TRY(hipMemcpy(out, s->sig_used, n * 4, hipMemcpyDeviceToHost));
  }
  return CHIRON_OK;
}

// Bias correction of the f16 engine (post-training-quantisation style, data dependent only through per-channel MEANS of a
// calibration batch): run the network, measure the mean of every input channel of every f16 weight matrix, move
// sum_k E[x_k] (f16(W) - W)[n][k] out of the shift / LSTM bias of output n.  Upstream corrections move downstream means a little,
// hence `iterations` (2 is enough).  tools/f16_study.py: on trained-checkpoint-like weights the mean term is most of what rounding
// the weights to halves costs.
extern "C" chiron_status chiron_engine_calibrate(chiron_engine* e, const float* x, const int32_t* seq_len, int32_t batch, int32_t iterations) {
  if (!e) return fail(CHIRON_ERR_INVALID, "null engine");
  if (iterations < 0 || iterations > 16) return fail(CHIRON_ERR_INVALID, "iterations %d", iterations);
  if (iterations > 0) {     // iterations = 0 only restores the uncorrected shifts: it reads no calibration data
    if (!x || !seq_len) return fail(CHIRON_ERR_INVALID, "null x/seq_len");
    if (batch < 1 || batch > e->maxB) return fail(CHIRON_ERR_OVERFLOW, "batch %d exceeds max_batch %d", batch, e->maxB);
  }
  if (!e->f16 || e->w2) return CHIRON_OK;      // fp32 / fp32-split / hi + lo weights are not rounded: nothing to correct
  Slot* s = &e->slots[0];
  for (Slot& sl : e->slots)
    if (sl.state.v.load(std::memory_order_acquire) != 0) return fail(CHIRON_ERR_STATE, "chiron_engine_calibrate needs every slot idle");
  HIP_TRY(hipSetDevice(e->opts.device_id));
  const int B = batch, H = e->H;
  CalibCtx ctx;
  ctx.max_records = 256;
  HIP_TRY(hipMalloc((void**)&ctx.dev_sums, ctx.max_records * 256 * sizeof(double)));
  struct FreeSums {            // every exit path below, early ones included
    double* p;
    ~FreeSums() { (void)hipFree(p); }
  } free_sums{ctx.dev_sums};
  chiron_status st = CHIRON_OK;
  std::vector<double> sums(ctx.max_records * 256);
  for (int it = 0; it < iterations && st == CHIRON_OK; ++it) {
    ctx.rec.clear();
    ctx.skipped = 0;
    auto run = [&]() -> chiron_status {
      HIP_TRY(hipMemsetAsync(ctx.dev_sums, 0, ctx.max_records * 256 * sizeof(double), s->stream));
      HIP_TRY(hipMemsetAsync(s->seq, 0, e->BP * 4, s->stream));
      memcpy(s->h_sig, x, (size_t)B * e->L * 4);
      memcpy(s->h_seq, seq_len, (size_t)B * 4);
      HIP_TRY(hipMemcpyAsync(s->sig, s->h_sig, (size_t)B * e->L * 4, hipMemcpyHostToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(s->seq, s->h_seq, (size_t)B * 4, hipMemcpyHostToDevice, s->stream));
      e->calib = &ctx;
      const bool ok = run_cnn(e, s, B, s->sig) && run_rnn(e, s, B);
      e->calib = nullptr;
      if (!ok) return fail(CHIRON_ERR_INVALID, "a GEMM of this topology has no kernel for the engine's dtype");
      HIP_TRY(hipStreamSynchronize(s->stream));
      HIP_TRY(hipMemcpy(sums.data(), ctx.dev_sums, ctx.rec.size() * 256 * sizeof(double), hipMemcpyDeviceToHost));
      return CHIRON_OK;
    };
    st = run();
    s->net_batch = 0;
    if (st == CHIRON_OK && ctx.skipped)
      st = fail(CHIRON_ERR_OVERFLOW, "calibration could not measure %d of the network's inputs (more than 256 channels, or more than %zu "
                "measured inputs): no correction applied.  Run this model uncorrected (`chiron call --no-calibration`, "
                "Engine(calibrate=False), `serve.py --no-calibration`) or with dtype fp16-w2, whose weights are exact", ctx.skipped, ctx.max_records);
    if (st) {
      resync_tile_counters(s);
      break;
    }
    // corrections, always from the ORIGINAL shifts: correction[n] = - sum over the records of the plan of sum_c mean[c] * dW[n][k0 + c]
    std::map<const float*, std::vector<double>> corr;
    const int parts = getenv("CHIRON_CALIB_PARTS") ? atoi(getenv("CHIRON_CALIB_PARTS")) : 7;   // diagnostics: 1 convolutions, 2 LSTM x rows, 4 LSTM h rows
    for (const CalibRecord& r : ctx.rec) {
      auto ph = e->plan_host.find(r.shift);
      if (ph == e->plan_host.end() || r.rows <= 0) continue;
      if (!(parts & (r.lstm_layer < 0 ? 1 : r.lstm_part == 0 ? 2 : 4))) continue;
      std::vector<double>& c = corr[r.shift];
      c.resize(ph->second.Npad, 0.0);
      const double* sm = &sums[r.slot * 256];
      if (r.lstm_layer < 0) {
        const PlanHost& P = ph->second;
        for (int n = 0; n < P.N; ++n) {
          double a = 0;
          for (int k = 0; k < r.cin; ++k) a += sm[k] * (double)P.dW[(size_t)n * P.K + r.k0 + k];
          c[n] -= a / r.rows;
        }
      } else {
        // rows of the TF kernel [(in_w + H)][4H] of layer l, direction d: x rows (part 0) or h rows (part 1); output column n of the
        // plan = dir * 4H + gate * H + unit (one direction per plan in the MultiRNN layers above 0)
        const LstmPlan& lp = e->lstm[r.lstm_layer];
        const int zc = 4 * H;
        for (int dir = 0; dir < 2; ++dir) {
          if (r.lstm_dir >= 0 && dir != r.lstm_dir) continue;
          const float* kern = e->host_weights.data() + e->lstm_kernel_off[dir][r.lstm_layer];
          const int row0 = r.lstm_part == 0 ? 0 : lp.in_w;
          const int nbase = lp.nproj == 1 ? dir * zc : 0;
          for (int col = 0; col < zc; ++col) {
            double a = 0;
            for (int k = 0; k < r.cin; ++k) {
              const float wv = kern[(size_t)(row0 + k) * zc + col];
              a += sm[k] * ((double)(float)(_Float16)wv - (double)wv);
            }
            c[nbase + col] -= a / r.rows;
          }
        }
      }
    }
    for (auto& kv : corr) {
      const PlanHost& P = e->plan_host[kv.first];
      std::vector<float> sh(P.shift0);
      for (int n = 0; n < P.Npad; ++n) sh[n] = (float)((double)P.shift0[n] + kv.second[n]);
      if (hipMemcpy(const_cast<float*>(kv.first), sh.data(), sh.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        st = fail(CHIRON_ERR_DEVICE, "uploading a corrected shift failed");
        break;
      }
    }
    if (st == CHIRON_OK) e->calibrated = it + 1;
  }
  if (iterations == 0) {   // back to the uncorrected engine
    for (auto& kv : e->plan_host)
      if (hipMemcpy(const_cast<float*>(kv.first), kv.second.shift0.data(), kv.second.shift0.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        st = fail(CHIRON_ERR_DEVICE, "restoring a shift failed");
    e->calibrated = 0;
  }
  return st;
}

// The recurrent stack's output (rnn.py:63-65 / :140-145, the tensor the FC head of rnn.py:72-96 reads), re-ordered from the
// engine's time-major [T][BP][2H] to the reference's [batch, T, 2H].
extern "C" chiron_status chiron_engine_rnn_output(chiron_engine* e, int32_t slot, float* out, size_t cap_floats, int32_t* out_batch,
                                                  int32_t* out_width) {
  if (!e) return fail(CHIRON_ERR_INVALID, "null engine");
  if (slot < 0 || slot >= (int)e->slots.size()) return fail(CHIRON_ERR_STATE, "slot %d out of range", slot);
  Slot* s = &e->slots[slot];
  if (s->state.v.load(std::memory_order_acquire) != 0) return fail(CHIRON_ERR_STATE, "slot %d holds an uncollected batch", slot);
  if (s->net_batch < 1 || s->rnn_out == nullptr)
    return fail(CHIRON_ERR_STATE, "slot %d holds no network batch (nothing submitted yet, or its last batch was decode-only)", slot);
  if (e->split) return fail(CHIRON_ERR_INVALID, "chiron_engine_rnn_output: not exported for dtype fp32-split");
  const int B = s->net_batch, T = e->T, W = 2 * e->H, BP = e->BP, ld = e->lasth_ld;
  const size_t n = (size_t)B * T * W;
  if (out_batch) *out_batch = B;
  if (out_width) *out_width = W;
  if (!out || cap_floats < n) return fail(CHIRON_ERR_OVERFLOW, "the recurrent output needs %zu floats, capacity %zu", n, cap_floats);
  HIP_TRY(hipSetDevice(e->opts.device_id));
  HIP_TRY(hipStreamSynchronize(s->stream));
  const size_t all = (size_t)T * BP * ld;
  if (!s->rnn_out_f32) {
    std::vector<_Float16> h(all);
    HIP_TRY(hipMemcpy(h.data(), s->rnn_out, all * 2, hipMemcpyDeviceToHost));
    for (int t = 0; t < T; ++t)
      for (int b = 0; b < B; ++b)
        for (int k = 0; k < W; ++k) out[((size_t)b * T + t) * W + k] = (float)h[((size_t)t * BP + b) * ld + k];
  } else {
    std::vector<float> h(all);
    HIP_TRY(hipMemcpy(h.data(), s->rnn_out, all * 4, hipMemcpyDeviceToHost));
    for (int t = 0; t < T; ++t)
      for (int b = 0; b < B; ++b) memcpy(out + ((size_t)b * T + t) * W, h.data() + ((size_t)t * BP + b) * ld, (size_t)W * 4);
  }
  return CHIRON_OK;
}

extern "C" chiron_status chiron_engine_profile(chiron_engine* e, int32_t enable) {
  if (!e) return fail(CHIRON_ERR_INVALID, "null engine");
  chiron_status st = chiron_engine_sync(e);
  if (st) return st;
  for (Slot& s : e->slots) {
    for (ProfEvent& ev : s.events) {
      hipEventDestroy(ev.a);
      hipEventDestroy(ev.b);
    }
    s.events.clear();
  }
  e->profiling = enable != 0;
  return CHIRON_OK;
}

extern "C" chiron_status chiron_engine_profile_read(chiron_engine* e, chiron_kernel_stat* stats, int32_t max_stats,
                                                    int32_t* n_stats) {
  if (!e || !stats || !n_stats) return fail(CHIRON_ERR_INVALID, "null argument");
  chiron_status st = chiron_engine_sync(e);
  if (st) return st;
  std::vector<chiron_kernel_stat> acc(e->prof_names.size());
  for (size_t i = 0; i < acc.size(); ++i) {
    memset(&acc[i], 0, sizeof(acc[i]));
    snprintf(acc[i].name, sizeof(acc[i].name), "%s", e->prof_names[i].c_str());
  }
  for (Slot& s : e->slots)
    for (ProfEvent& ev : s.events) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev.a, ev.b) != hipSuccess) continue;
      chiron_kernel_stat& k = acc[ev.name_id];
      k.total_ms += ms;
      k.launches += 1;
      k.flops += ev.flops;
      k.bytes += ev.bytes;
    }
  int n = 0;
  for (size_t i = 0; i < acc.size() && n < max_stats; ++i)
    if (acc[i].launches > 0) stats[n++] = acc[i];
  *n_stats = n;
  return CHIRON_OK;
}
