/*
 * chiron_amd.h -- C ABI of libchiron_amd.so, the MI355X (gfx950) basecalling
 * inference engine that replaces the TensorFlow session behind the reference's
 * `chiron call` hot path.
 *
 * The reference has no FFI layer; its seam is the pair of sess.run calls in
 * chiron/chiron_eval.py (feed: :335-342, drain: :403-409), formalised by the
 * SavedModel PREDICT signature in chiron/export_test.py:103-112
 *   (x, seq_len) -> (indices, values, dense_shape, logits, prob_logits, log_prob).
 * Every entry point below cites the reference interface it replaces.
 *
 * Conventions: plain pointers and sizes only; every function returns a
 * chiron_status (0 = OK) and never throws; chiron_last_error() returns a
 * thread-local description of the last failure.  Host buffers belong to the
 * caller; device buffers, streams and weights belong to the engine; result
 * pointers stay valid until the next submit on the same slot.
 */
#ifndef CHIRON_AMD_H
#define CHIRON_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHIRON_ABI_VERSION 7
#define CHIRON_MAX_BLOCKS 8
#define CHIRON_CLASSES 5 /* A,C,G,T,blank (rnn.py:25 class_n=5) */

typedef enum {
  CHIRON_OK = 0,
  CHIRON_ERR_INVALID = 1,   /* bad argument / unsupported topology          */
  CHIRON_ERR_DEVICE = 2,    /* HIP runtime failure (no GPU, OOM, launch)     */
  CHIRON_ERR_STATE = 3,     /* collect without submit, submit on a slot whose batch was not collected, slot out of range */
  CHIRON_ERR_OVERFLOW = 4   /* batch > max_batch, beam > limit, a tensor beyond the kernels' 32-bit addressing */
} chiron_status;

/* One residual block, chiron/cnn.py:234-262 residual_layer():
 *   branch1 = conv 1x1 (stride) [+BN iff i_bn];  branch2 = 1x1+BN+ReLU ->
 *   1xk (stride)+BN+ReLU -> 1x1+BN;  out = ReLU(branch1 + branch2).          */
typedef struct {
  int32_t in_channels;  /* 1 for the first block (raw signal)                 */
  int32_t out_channels; /* 256                                                */
  int32_t k;            /* width of conv2b: 3 (DNA), 13 (shipped RNA block 1) */
  int32_t stride;       /* stride of conv2b and branch1/conv1                 */
  int32_t i_bn;         /* BN on branch1 (only res_layer1, cnn.py:382-384)    */
} chiron_res_block;

typedef enum {
  CHIRON_RNN_STACK = 0, /* rnn.py:20-97  stack_bidirectional_dynamic_rnn (DNA) */
  CHIRON_RNN_MULTI = 1  /* rnn.py:99-174 bidirectional_dynamic_rnn(MultiRNNCell) (RNA) */
} chiron_rnn_kind;

typedef enum {
  CHIRON_BN_POPULATION = 0, /* cnn.py:125-163 batchnorm() inference branch (shipped checkpoints) */
  CHIRON_BN_BATCH = 1       /* cnn.py:166-188 simple_global_bn (HEAD code)                        */
} chiron_bn_mode;

/* Topology descriptor: what chiron_model.read_config (chiron_model.py:37-48)
 * plus the checkpoint's variable shapes determine.                           */
typedef struct {
  int32_t n_blocks;
  chiron_res_block blocks[CHIRON_MAX_BLOCKS];
  int32_t rnn_kind;   /* chiron_rnn_kind                                      */
  int32_t rnn_layers; /* 3                                                    */
  int32_t hidden;     /* 100                                                  */
  int32_t classes;    /* 5                                                    */
  int32_t bn_mode;    /* chiron_bn_mode                                       */
  /* Optional stem in front of the residual blocks: HEAD's RNA_model2 / RNA_model3 (cnn.py:454-476) start with
   *   conv_layer(net, [1, k, 1, C], 'SAME', strides = s) + BN + ReLU  (k, s = 9, 5 / 14, 7; C = 256)
   * under the variable scope conv_layer/conv1.  stem_k = 0: no stem, blocks[0].in_channels must be 1 (DNA_model1 and
   * the shipped RNA graph).  With a stem, blocks[0].in_channels = stem_channels.                                 */
  int32_t stem_k, stem_stride, stem_channels;
} chiron_model_desc;

/* Weight blob layout (float32, little endian), in this order:
 *  if stem_k: conv_layer/conv1/weights [k][1][C], conv1_bn scale, offset, pop_mean, pop_var   4 x [C]
 *  for each block b:
 *     branch1/conv1/weights            [1][in][out]        (TF HWIO, H squeezed)
 *     if i_bn: conv1_bn scale, offset, pop_mean, pop_var   4 x [out]
 *     branch2/conv2a/weights           [1][in][out]
 *     conv2a_bn scale, offset, pop_mean, pop_var           4 x [out]
 *     branch2/conv2b/weights           [k][out][out]
 *     conv2b_bn ...                                        4 x [out]
 *     branch2/conv2c/weights           [1][out][out]
 *     conv2c_bn ...                                        4 x [out]
 *  for each rnn layer l, for dir in (fw, bw):
 *     lstm_cell/kernel                 [(in_l + H)][4H]    columns i|j|f|o
 *     lstm_cell/bias                   [4H]
 *       in_l: layer 0 -> out_channels of the last block;
 *             STACK l>0 -> 2H ; MULTI l>0 -> H
 *  rnn_fnn_layer/weights [2][H], bias [H], weights_class [H][K], bias_class [K]
 * (in CHIRON_BN_BATCH mode the pop_mean/pop_var slots are present but unused.) */
chiron_status chiron_weights_size(const chiron_model_desc* desc, size_t* n_floats);

/* CHIRON_F32: exact-fp32 path (fp32 MFMA), the parity path: logits within 1e-4 of the reference arithmetic.
 * CHIRON_F16: activations, weights and the recurrent h as IEEE halves on the f16 MFMA instructions; accumulation,
 *             the LSTM pre-activations z, gates, cell state, logits and both CTC decoders stay fp32 (BASELINE
 *             configs[4]); the LAST recurrent layer's output -- what the FC head reads -- leaves the recurrence as fp32 (round 6).
 *             Logits stay within 0.08 of the fp32 engine on the synthetic weights (tests/test_gpu_parity.py); identical-window rates
 *             against the fp32 engine per regime: profiles/r06_f16_frontier.json.
 * CHIRON_F32_SPLIT: fp32 VALUES, carried between kernels as hi + lo half pairs (x = hi + lo to 2^-22: 22-bit operands) and multiplied
 *             on the f16 matrix cores as hi*hi + hi*lo + lo*hi with fp32 accumulation (the f16 MFMA rate is 16x the fp32 one on
 *             gfx950); GEMM weight rows are stored scaled by a power of two so that their lo halves are normal halves; gates, z, logits
 *             and CTC are the fp32 code.  Opt-in.  What it meets (tests): the 1e-4 logits bound against the oracle on the seeded
 *             synthetic weights, as CHIRON_F32 does; on trained-checkpoint-like weights, where no float32 pipeline meets 1e-4, it is
 *             judged against the same ensemble of float32 realisations as CHIRON_F32 with wider bars (bulk rms <= 1.5 x the ensemble's
 *             p90, typical window <= 1.75 x its median; measured 0.75 .. 1.4 and 1.0 .. 1.6 -- CHIRON_F32: 0.63 .. 1.15 and 1.13 .. 1.23);
 *             greedy strings at basecalling density as CHIRON_F32.  It is not fp32 MFMA arithmetic, so the headline benchmark stays
 *             on CHIRON_F32.  Population BN only.
 * CHIRON_F16_W2: CHIRON_F16's activations (halves) against EXACT weights: every weight is carried as a hi + lo half pair
 *             (W = hi + lo to 2^-22) and every product is x*lo + x*hi on the f16 matrix cores, fp32 accumulation; the LSTM
 *             pre-activations z stay fp32 in memory.  What half precision costs this network is mostly the WEIGHTS'
 *             rounding (tools/f16_study.py: 10 x the activations'), and no calibration data is needed to avoid it:
 *             the mode for trained checkpoints when CHIRON_F16's accuracy is not enough.  Population BN only.     */
typedef enum { CHIRON_F32 = 0, CHIRON_F16 = 1, CHIRON_F32_SPLIT = 2, CHIRON_F16_W2 = 3 } chiron_dtype;

typedef struct {
  int32_t device_id;    /* HIP device ordinal                                 */
  int32_t max_batch;    /* FLAGS.batch_size (chiron_eval.py:248)              */
  int32_t segment_len;  /* FLAGS.segment_len                                  */
  int32_t n_slots;      /* in-flight batches (>=1); each has its own stream   */
  int32_t dtype;        /* chiron_dtype; CHIRON_F32 is the parity path        */
  int32_t max_beam;     /* largest beam_width that will be requested (0=greedy only) */
} chiron_engine_opts;

typedef struct chiron_engine chiron_engine;

/* Replaces build_eval_graph + Saver.restore (chiron_eval.py:244-276).        */
chiron_status chiron_engine_create(const chiron_model_desc* desc, const float* weights, size_t n_floats,
                                   const chiron_engine_opts* opts, chiron_engine** out);
void chiron_engine_destroy(chiron_engine* e);

/* Sizes of the engine chiron_engine_create would build for (desc, opts), WITHOUT touching a GPU: frame count and ratio,
 * the largest tensor the kernels address with 32-bit byte offsets against their limit, device bytes per slot and in
 * total.  Returns CHIRON_ERR_OVERFLOW (with the largest max_batch that fits in the message) when a tensor would pass
 * the limit -- e.g. fp32, segment_len 400, 256 channels: max_batch > 10485 -- and chiron_engine_create refuses the
 * same configurations with the same status instead of reading zeros past the descriptor's range.                  */
typedef struct {
  int32_t T;
  double ratio;
  uint64_t largest_tensor_bytes;
  uint64_t tensor_limit_bytes;
  uint64_t slot_bytes;
  uint64_t total_bytes;
} chiron_engine_sizes;
chiron_status chiron_engine_plan(const chiron_model_desc* desc, const chiron_engine_opts* opts, chiron_engine_sizes* out);

/* T = number of logits frames per segment and ratio = segment_len / T
 * (chiron_model.py:151-152).                                                 */
chiron_status chiron_engine_dims(const chiron_engine* e, int32_t* out_T, double* out_ratio);

/* Threading: one producer thread (submit / decode) and one consumer thread (collect) per engine may run concurrently;
 * a slot alternates strictly submit -> collect (its state is an atomic, a second submit before the collect returns
 * CHIRON_ERR_STATE and changes nothing).  Engines on different devices are independent.  A submit that fails after
 * it has started to enqueue work drains the slot's stream before returning; the slot stays free.
 *
 * flags for submit */
#define CHIRON_X_ON_DEVICE 1u   /* x / seq_len are device pointers on opts.device_id.  The slot streams are non-blocking
                                   streams and are NOT ordered against the caller's: the buffers must be complete before
                                   the call (synchronise the producing stream) and untouched until the collect          */
#define CHIRON_WANT_PROB 2u     /* compute prob_logits = path_prob (chiron_eval.py:116-136, -e fastq) */
#define CHIRON_WANT_LOGITS 4u   /* copy logits [B,T,K] back on collect                */
#define CHIRON_NO_DECODE_COPY 8u /* leave decoded sparse tensor on the device (bench)  */
#define CHIRON_COMPACT_DECODE 16u /* the host wants the decode in the per-row form of the regroup step (chiron_eval.py:403-446): collect
                                    fills flat_labels / row_counts and does NOT copy indices / values (NULL; nnz and dense_shape are
                                    still reported).  One fixed-size copy enqueued with the batch instead of a second round trip
                                    for nnz * 24 bytes of int64 pairs that the host would only scan for row boundaries.           */

/* Replaces sess.run(logits_enqueue, feed_dict) (chiron_eval.py:335-342) plus the
 * decode sub-graph (chiron_eval.py:465-492).  x: float32 [batch, segment_len]
 * row-major; seq_len: int32 [batch], ALREADY divided by ratio and rounded
 * half-even by the caller (chiron_eval.py:337).  beam_width 0 = greedy
 * (merge_repeated=True), >0 = CTC beam search (merge_repeated=False, top_paths=1).
 * Asynchronous: work is enqueued on the slot's stream.                         */
chiron_status chiron_engine_submit(chiron_engine* e, int32_t slot, const float* x, const int32_t* seq_len,
                                   int32_t batch, int32_t beam_width, uint32_t flags);

/* Decode-only: the decode sub-graph of the reference (decoding_queue, chiron_eval.py:465-492: path_prob +
 * ctc_greedy_decoder / ctc_beam_search_decoder) on caller-supplied logits float32 [batch, T, K] (T from
 * chiron_engine_dims).  Same flags, slot and collect protocol as chiron_engine_submit.                  */
chiron_status chiron_engine_decode(chiron_engine* e, int32_t slot, const float* logits, const int32_t* seq_len,
                                   int32_t batch, int32_t beam_width, uint32_t flags);

/* The reference's decoded tuple (chiron_eval.py:403-409): SparseTensor
 * (indices, values, dense_shape) + log_prob + prob_logits (+ logits).          */
typedef struct {
  int64_t nnz;
  const int64_t* indices;   /* [nnz,2] (row, position), row-major sorted        */
  const int64_t* values;    /* [nnz] in 0..3                                    */
  int64_t dense_shape[2];   /* [batch, max decoded length]                      */
  const float* log_prob;    /* [batch,1] greedy: -sum max logit; beam: log p    */
  const float* prob_logits; /* [batch,1] path_prob, or zeros without WANT_PROB  */
  const float* logits;      /* [batch,T,K] or NULL                              */
  int32_t batch;
  int32_t T;
  /* CHIRON_COMPACT_DECODE only (else NULL): the rows' decoded labels 0..3 back to back in row order [nnz], and the number of
   * labels of every row [batch] -- the same decode as (indices, values): row b owns the next row_counts[b] entries.       */
  const uint8_t* flat_labels;
  const int32_t* row_counts;
} chiron_decoded;

/* chiron_engine_submit with the batch given as `n_pieces` host arrays of whole rows (piece i: piece_rows[i] rows of segment_len
 * floats, rows summing to `batch`): the cross-read packing of chiron_eval.py:321-334 -- the tail of one read, whole reads, the
 * head of the next -- copied straight into the slot's pinned staging buffer, so the caller never assembles the [batch,
 * segment_len] array (1.76 MB per 1100-window batch on the host's main thread otherwise).  piece_row_stride[i] (floats; NULL =
 * segment_len everywhere) is the distance between consecutive rows of piece i: with stride = jump the rows ARE the windows
 * signal[r*jump : r*jump + segment_len] of one zero-padded signal buffer (chiron_input.py:276-286) and the host never
 * materialises the windowed read either.  Host pointers only.                                                            */
chiron_status chiron_engine_submit_pieces(chiron_engine* e, int32_t slot, const float* const* pieces, const int32_t* piece_rows,
                                          const int64_t* piece_row_stride, int32_t n_pieces, const int32_t* seq_len, int32_t batch,
                                          int32_t beam_width, uint32_t flags);

/* Replaces sess.run(decode dequeue) (chiron_eval.py:403-409).  Blocks until the
 * slot's work is complete, then fills *out with host pointers owned by the slot. */
chiron_status chiron_engine_collect(chiron_engine* e, int32_t slot, chiron_decoded* out);

/* Blocks until every slot's stream is idle. */
chiron_status chiron_engine_sync(chiron_engine* e);

/* Device pointers of a slot's most recent results (valid after collect/sync):
 * logits [B,T,K] f32, and the decoded sparse tensor left on device.            */
chiron_status chiron_engine_device_results(chiron_engine* e, int32_t slot, const float** logits,
                                           const int64_t** indices, const int64_t** values,
                                           const int64_t** nnz_and_shape /* [3]: nnz,batch,maxlen */);

/* getcnnfeature (cnn.py:334-371): the CNN feature tensor [batch, T, C] (float32; an f16 engine's halves are widened)
 * of the batch most recently run on `slot`, which must be idle (collected).  Copied into out [cap_floats];
 * *out_batch / *out_channels receive batch and C.  CHIRON_ERR_OVERFLOW when cap_floats is too small (the sizes are
 * still reported), CHIRON_ERR_STATE before the first batch or with a batch in flight.  Stage-level parity checks use
 * it (tests); dtype CHIRON_F32_SPLIT does not export its hi/lo pairs (CHIRON_ERR_INVALID).                       */
chiron_status chiron_engine_features(chiron_engine* e, int32_t slot, float* out, size_t cap_floats, int32_t* out_batch,
                                     int32_t* out_channels);

/* The recurrent stack's output, rnn.py:63-65 (DNA: stack_bidirectional_dynamic_rnn) / rnn.py:140-145 (RNA: MultiRNNCell inside
 * bidirectional_dynamic_rnn) -- `lasth`, the tensor the FC head of rnn.py:72-96 reads: [batch, T, 2 * hidden] float32 ([..., :H]
 * forward, [..., H:] backward; frames at or past a row's seq_len are 0), of the batch most recently run on `slot` (idle).  Same
 * protocol and status codes as chiron_engine_features.  With a model descriptor of 1 or 2 rnn_layers it is the output of that
 * layer: the per-stage error budget of the parity tests (tools/parity_budget.py) is built on it.                        */
chiron_status chiron_engine_rnn_output(chiron_engine* e, int32_t slot, float* out, size_t cap_floats, int32_t* out_batch,
                                       int32_t* out_width);

/* f16 engines only (CHIRON_F16; a no-op returning CHIRON_OK for the other dtypes): bias correction for the weights' rounding to
 * halves.  The reference has no counterpart (it computes in fp32); this is what lets the f16 engine be used on a trained checkpoint
 * whose BN sites cancel large convolution means (tools/f16_study.py: there most of what half-precision WEIGHTS cost is a constant
 * per output channel, sum_k E[x_k] (f16(W) - W)[n][k]).  The call runs `batch` calibration windows (same x / seq_len convention as
 * chiron_engine_submit, host pointers) through the network `iterations` times (2 is enough; upstream corrections move downstream
 * means slightly), measures the mean of every input channel of every f16 weight matrix -- convolutions, LSTM input and recurrent
 * kernels -- and moves that constant out of the folded BN shift / LSTM bias.  No run-time cost afterwards.  Every slot must be idle.
 * iterations = 0 restores the uncorrected shifts (x, seq_len and batch are then ignored and may be NULL / 0).  An input that cannot
 * be measured (more than 256 channels; more than 256 measured inputs) is CHIRON_ERR_OVERFLOW with nothing applied.  The correction depends on the calibration windows through per-channel MEANS
 * only; `chiron call --dtype fp16` calibrates on a fixed synthetic squiggle, so every rank of a sharded run holds the same engine. */
chiron_status chiron_engine_calibrate(chiron_engine* e, const float* x, const int32_t* seq_len, int32_t batch, int32_t iterations);

/* Per-kernel timing with HIP events on the engine's own streams (bench.py
 * roofline).  Enable, run, sync, then read.                                    */
typedef struct {
  char name[48];
  double total_ms;     /* sum of hipEventElapsedTime over launches              */
  int64_t launches;
  double flops;        /* algorithmic FLOPs summed over those launches          */
  double bytes;        /* algorithmic HBM bytes summed over those launches      */
} chiron_kernel_stat;
chiron_status chiron_engine_profile(chiron_engine* e, int32_t enable);
chiron_status chiron_engine_profile_read(chiron_engine* e, chiron_kernel_stat* stats, int32_t max_stats,
                                         int32_t* n_stats);

/* Overlap-consensus vote, chiron/utils/easy_assembler.py:
 *   glue_kernal :276-294, stick_kernal :296-300, simple_assembly_kernal :212-250 (difflib.SequenceMatcher matching
 *   blocks + offset prior; `error_rate`, `jump_step_ratio` as in simple_assembly(bpreads, jump_step_ratio, error_rate,
 *   kernal) :302), simple_assembly(_qs) :302-335 / :393-432, add_count(_qs) :381-387 / :435-442, and the argmax of
 *   chiron_eval.py:457.
 * bases: concatenated segments as 0..3; seg_off [n_seg+1] prefix offsets; seg_qs [n_seg] per-segment quality (may be
 * NULL); kernal: one of CHIRON_KERNAL_*; error_rate / jump_step_ratio are read by the simple kernel only.
 * Outputs (caller-allocated, capacity cap columns): counts [4][cap] float64, qs_sum [4][cap] float64 (if seg_qs),
 * *out_len = consensus length.  Returns CHIRON_ERR_OVERFLOW if cap is too small (then *out_len = needed).
 * Pure host code: callable without a GPU and from several threads at once.                                      */
#define CHIRON_KERNAL_GLUE 1
#define CHIRON_KERNAL_STICK 2
#define CHIRON_KERNAL_SIMPLE 3
chiron_status chiron_assemble(const uint8_t* bases, const int64_t* seg_off, int64_t n_seg, const double* seg_qs,
                              int32_t kernal, double error_rate, double jump_step_ratio, double* counts, double* qs_sum,
                              int64_t cap, int64_t* out_len);

/* One read from its decoded windows to its files (chiron_eval.py:446-462 + write_output :176-228) in one call that never
 * enters the interpreter: index2base of every window, the vote above, np.argmax (:457), qs (:152-174; seg_qs = NULL or
 * fastq = 0: no quality, FASTA), then result_path = "@name\nSEQ\n+\nQUAL\n" (FASTQ) or ">name\nSEQ" (FASTA, no final
 * newline; rna != 0 writes U for T in the consensus, :204-205) and -- unless segments_path is NULL (--concise) --
 * ">name<k>\nSEGMENT\n" per window.  *consensus_len receives len(SEQ) (the caller writes meta/<name>.meta from it) and
 * consensus_out [consensus_cap], when given, SEQ itself (no terminator; the decoded base count is always enough room).
 * The folders must exist.  Equal, byte for byte, to the Python writers of chiron_amd/eval.py (tests).               */
chiron_status chiron_finish_read(const uint8_t* bases, const int64_t* seg_off, int64_t n_seg, const double* seg_qs,
                                 int32_t kernal, double error_rate, double jump_step_ratio, const char* name,
                                 const char* result_path, const char* segments_path, int32_t fastq, int32_t rna,
                                 char* consensus_out, int64_t consensus_cap, int64_t* consensus_len);

/* One displacement of the vote above: where `cur` starts relative to the start of `prev` (the return value of
 * glue_kernal / stick_kernal / simple_assembly_kernal; for the simple kernel *log_px receives its second return
 * value, the score of the chosen offset; 0 otherwise; log_px may be NULL).                                       */
chiron_status chiron_overlap_displacement(const uint8_t* cur, int64_t n, const uint8_t* prev, int64_t prev_n,
                                          int32_t kernal, double error_rate, double jump_step_ratio, int64_t* disp,
                                          double* log_px);

/* The same consensus on the device, for one read (very long reads): glue / stick displacements of every
 * consecutive segment pair, running start columns, the vote, and per consensus column the winning base
 * (np.argmax(consensus, axis=0), chiron_eval.py:457: first maximum) and, when seg_qs is given, the three numbers
 * qs() (chiron_eval.py:152-174) reads of a column: n1 and n2, the two largest vote counts, and q_top, the summed segment
 * quality behind the winning base (the last of equal maxima) -- i.e. chiron_assemble + argmax + the inputs of qs without
 * the [4][len] matrices ever leaving the GPU.  All of it is integer work or ordered double sums: identical to the host
 * vote.  The Phred character q = int(10 log10((n1+1)/(n2+1)) + q_top/n1/ln 10) is left to the caller's own formula
 * (chiron_amd.eval.qs_from_votes), because truncation turns a last-bit difference between two log10 implementations
 * into a different character.  Host pointers in and out; the call owns a stream and stream-ordered device buffers
 * (thread-safe, independent of any engine, no device-wide synchronisation).  consensus [cap] receives 0..3; n1, n2
 * [cap] int32 and q_top [cap] float64 come together or are all NULL; *out_len the length; CHIRON_ERR_OVERFLOW if cap is
 * too small (then *out_len = needed).  kernal: CHIRON_KERNAL_GLUE or _STICK.                                       */
chiron_status chiron_consensus_device(int32_t device_id, const uint8_t* bases, const int64_t* seg_off, int64_t n_seg,
                                      const double* seg_qs, int32_t kernal, uint8_t* consensus, int32_t* n1, int32_t* n2,
                                      double* q_top, int64_t cap, int64_t* out_len);

/* Host-side reader of the reference's raw-signal text format: chiron_input.py:527-539 read_signal() --
 * `f.read().split()` converted to float32 -- whitespace/newline separated numbers.  Each token is parsed as a
 * C double (like Python's float()) and then narrowed to float32, so the values equal numpy's conversion.
 * out holds up to cap values; *n_out receives the number parsed.  A token that is not a number ->
 * CHIRON_ERR_INVALID (the reference raises ValueError); more than cap values -> CHIRON_ERR_OVERFLOW.
 * Pure host code: callable without a GPU and from several threads at once (it does not touch Python). */
chiron_status chiron_parse_signal_text(const char* text, size_t len, float* out, size_t cap, size_t* n_out);

/* fast5 (HDF5) reader without libhdf5 / h5py: what chiron/utils/extract_sig_ref.py:149-193 (extract_file for single-read
 * files: the first group under /Raw/Reads; extract_file_v2 for multi-read files: one read per top-level group, sorted by
 * name) takes from a file -- raw signal, read_id attribute, reference FASTQ / FASTA if present.  Replaces, together with
 * chiron_input.py:541-555 read_signal_fast5, the extract -> `.signal` text -> read_signal round trip of `chiron call`
 * (SURVEY 8(f)1).  Host code (zlib for the deflate filter): no GPU, thread-safe, every handle owns its file image.
 * An unsupported or damaged file gives CHIRON_ERR_INVALID with the reason in chiron_last_error(); the caller logs and
 * skips it, like the reference (extract_sig_ref.py:97-117).                                                        */
typedef struct chiron_fast5 chiron_fast5;
chiron_status chiron_fast5_open(const char* path, chiron_fast5** out);
void chiron_fast5_close(chiron_fast5* f);
int32_t chiron_fast5_read_count(const chiron_fast5* f);
/* read i: suffix ("" for a single-read file, the read's group name in a multi-read file: the .signal name is
 * <file stem><suffix>, extract_sig_ref.py:128-144), read_id, number of samples, length of the reference text (0: none) */
chiron_status chiron_fast5_read_info(const chiron_fast5* f, int32_t i, char* suffix, size_t suffix_cap, char* read_id,
                                     size_t id_cap, int64_t* n_samples, int64_t* fastq_len);
/* the raw signal of read i as float32 (np.float32 of the DAC counts, chiron_input.py:536); reverse != 0 stores it
 * back to front (RNA: extract_sig_ref.py:165 / chiron_input.py:269-272) */
chiron_status chiron_fast5_signal(const chiron_fast5* f, int32_t i, float* out, int64_t cap, int32_t reverse);
chiron_status chiron_fast5_fastq(const chiron_fast5* f, int32_t i, char* out, int64_t cap);
/* extract_sig_ref.py:122-123: delimiter.join(str(v) for v in raw_signal) written to `path` for integer-valued samples
 * (what `chiron call` extracts: unit = False, entry.py:36); a non-integer sample is refused (CHIRON_ERR_INVALID). */
chiron_status chiron_write_signal_text(const char* path, const float* v, int64_t n, const char* delimiter);

/* The host side of `chiron call` on its direct fast5 path as ONE native call (ABI 7): reader threads (fast5 -> samples, raw/<name>.signal
 * and reference/<stem>_ref.fastq as extract_sig_ref.py:92-147 writes them, windows as rows of one zero-padded buffer at stride `jump`:
 * chiron_input.py:276-286), the CALLING thread packing batches across reads in file order (chiron_eval.py:321-334, seq_len rounded half
 * even :337), submitting them to the engine's slots and regrouping the compact decode per read (:403-446), finisher threads running
 * chiron_finish_read + meta/<name>.meta (:446-462, :176-242).  It replaces the Python thread pools of chiron_amd/eval.py:evaluation --
 * same files, byte for byte, except the timings inside meta/ -- for: fast5 input, bn_mode population (a partial last batch is submitted
 * as it is), host-side vote for every read; `.signal` text files are taken as well (name_root).  The folders raw/ reference/ result/
 * segments/ meta/ under `output` must exist (sub-folders of a recursive `.signal` input are created on demand).
 * null_engine != 0: no engine call is made (e may be NULL): collect() is replaced by a canned decode of ~44 bases per window -- the host
 * pipeline's own ceiling (tools/host_ceiling.py).  Unreadable files are skipped and reported in stats->messages (the reference logs and
 * skips them, extract_sig_ref.py:97-117); a failed engine call or file write ends the run with its status.                           */
typedef struct {
  int32_t batch_size, segment_len, jump, start;   /* FLAGS.batch_size / segment_len / jump / start                                    */
  int32_t beam;                                    /* 0 = greedy                                                                       */
  int32_t fastq, concise, rna, no_raw;             /* -e fastq; --concise; --mode rna (signal reversed, U for T); --no-raw             */
  int32_t n_threads;                               /* reader threads = finisher threads                                                */
  int32_t n_slots;                                 /* the engine's slots (chiron_engine_opts.n_slots)                                  */
  int32_t null_engine;
  double null_ratio;                               /* null engine only: segment_len / T                                                */
  const char* output;                              /* FLAGS.output                                                                     */
  const char* delimiter;                           /* of raw/<name>.signal ("\n": HEAD; NULL = "\n")                                   */
  const char* input_name;                          /* meta/<name>.meta: FLAGS.input, FLAGS.model                                       */
  const char* model_name;
  const char* name_root;                           /* `.signal` inputs (a path ending in ".signal" is parsed as text: chiron_input.py:527-539, and
                                                      nothing is written to raw/ or reference/): a read is named by its path relative to this
                                                      folder, sub-folders included (chiron_eval.py:277-293); NULL: by its file name          */
} chiron_pipeline_opts;
typedef struct {
  int64_t reads, reads_finished, windows, batches, consensus_bases, files_failed;
  double seconds;
  char messages[4096];                             /* one line per skipped file / first failure                                        */
} chiron_pipeline_stats;
chiron_status chiron_pipeline_run(chiron_engine* e, const char* const* fast5_paths, int64_t n_paths, const chiron_pipeline_opts* opts,
                                  chiron_pipeline_stats* stats);

/* CRC-32C (Castagnoli) of a byte range: the checksum TF's tensor-bundle checkpoints record per tensor
 * (BundleEntryProto.crc32c holds its masked form; tensor_bundle.cc verifies it in Saver.restore, chiron_eval.py:276).
 * Host code, used by the checkpoint reader.                                                                        */
chiron_status chiron_crc32c(const void* data, size_t len, uint32_t* out);

/* PCI address ("0000:c1:00.0", lower case as sysfs spells it) of HIP device `device_id`: what `chiron call --gpus N` needs to put
 * rank r on the cores of the NUMA node its GPU hangs off (/sys/bus/pci/devices/<address>/numa_node; chiron_amd/shard.py).
 * The reference has no counterpart: it is one process on one device (chiron_eval.py:255-262).  cap >= 16.            */
chiron_status chiron_device_pci_bus_id(int32_t device_id, char* out, size_t cap);

const char* chiron_last_error(void);
int32_t chiron_abi_version(void);
/* What kind of build this library is.  CHIRON_BUILD_TIMING: at least one object was compiled as a timing-only kernel variant
 * (csrc/timing_variants.h: parts of a kernel switched off to measure what they cost) -- its results are GARBAGE; the Python
 * binding refuses such a library unless CHIRON_ALLOW_TIMING_BUILD=1.  0 for the product.                               */
#define CHIRON_BUILD_TIMING 1u
uint32_t chiron_build_flags(void);

#ifdef __cplusplus
}
#endif
#endif /* CHIRON_AMD_H */
