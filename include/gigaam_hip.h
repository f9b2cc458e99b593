/* gigaam_hip.h -- C ABI of libgigaam_hip.so: the MI355X (gfx950) inference path for
 * GigaAM's log-mel frontend, Conformer encoder and CTC / RNN-T greedy decoders.
 *
 * This is the drop-in boundary of SURVEY.md §8b.  The reference has no FFI of its own
 * (it is pure Python on torch); what it has is three operator slots instantiated from
 * the checkpoint config (reference gigaam/model.py:24-25,93-94) with plain-tensor call
 * contracts.  Each entry point below replaces exactly one of those calls and is what a
 * ctypes binding on the reference side would load (INTEGRATION.md shows the stub):
 *
 *   gam_frontend     <- FeatureExtractor.forward          gigaam/preprocess.py:94-98
 *   gam_encode       <- ConformerEncoder.forward          gigaam/encoder.py:605-647
 *   gam_ctc_head     <- CTCHead.forward                   gigaam/decoder.py:18-21
 *   gam_ctc_greedy   <- CTCGreedyDecoding.decode          gigaam/decoding.py:56-96
 *   gam_rnnt_greedy  <- RNNTGreedyDecoding.decode         gigaam/decoding.py:128-207
 *                        (+ RNNTDecoder.predict decoder.py:85-102, RNNTJoint.joint :41-47)
 *   gam_emo_probs    <- GigaAMEmo.get_probs (pool+head)    gigaam/model.py:272-285
 *   gam_set_weight   <- nn.Module.load_state_dict         gigaam/__init__.py:185
 *   gam_create       <- hydra.utils.instantiate(cfg.*)    gigaam/model.py:24-25,93-94
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  All tensor pointers are DEVICE
 *     pointers unless the parameter says "host".  The caller owns every input/output
 *     buffer; the library owns weights, position tables and a grow-only workspace.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and performs no host synchronisation, except that a workspace
 *     growth (first call at a larger shape) allocates.
 *   - return value 0 = success, negative = error; gam_last_error() has the message.
 *   - one handle per device; a handle is not thread-safe; distinct handles are independent.
 *   - storage and accumulation are fp32 end to end (the parity target is the reference's
 *     fp32 CPU path); see gam_set_gemm_mode for how the dense contractions are evaluated.
 */
#ifndef GIGAAM_HIP_H
#define GIGAAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAM_ABI_VERSION 1

typedef struct gam_handle gam_handle;

enum { GAM_SUBS_CONV2D = 0, GAM_SUBS_CONV1D = 1 };
enum { GAM_ATT_ROTARY = 0, GAM_ATT_REL_POS = 1 };
enum { GAM_NORM_BATCH = 0, GAM_NORM_LAYER = 1 };
enum { GAM_HEAD_NONE = 0, GAM_HEAD_CTC = 1, GAM_HEAD_RNNT = 2, GAM_HEAD_EMO = 3 };
enum { GAM_DTYPE_F32 = 0, GAM_DTYPE_F16 = 1, GAM_DTYPE_BF16 = 2, GAM_DTYPE_F64 = 3, GAM_DTYPE_I64 = 4 };

/* POD mirror of the four cfg sub-trees of a GigaAM checkpoint. */
typedef struct gam_config {
  /* cfg.preprocessor -- FeatureExtractor(sample_rate, features, **kwargs), preprocess.py:60-65 */
  int32_t sample_rate, n_mels, hop_length, win_length, n_fft, center;
  /* cfg.encoder -- ConformerEncoder(...), encoder.py:510-526 */
  int32_t feat_in, n_layers, d_model, subsampling, subs_kernel_size, subsampling_factor;
  int32_t ff_expansion_factor, self_attention_model, n_heads, pos_emb_max_len;
  int32_t conv_norm_type, conv_kernel_size;
  /* cfg.head -- CTCHead(feat_in, num_classes) decoder.py:12-16 | RNNTHead(decoder, joint) :146-149 |
   * emotion model: a Linear(d_model, num_classes) (model.py:267-270; keys head.weight / head.bias) */
  int32_t head_type, num_classes, pred_hidden, pred_rnn_layers, joint_hidden;
} gam_config;

int gam_abi_version(void);

/* Construct the operators for one device.  Weights arrive through gam_set_weight. */
int gam_create(const gam_config* cfg, int device_id, gam_handle** out);
void gam_destroy(gam_handle* h);

/* Stage one state_dict entry (HOST pointer, copied).  `key` is the reference's
 * state_dict key ("encoder.layers.0.self_attn.linear_q.weight", ...).
 * Unknown keys are accepted and ignored (e.g. num_batches_tracked). */
int gam_set_weight(gam_handle* h, const char* key, const void* host_ptr, int dtype,
                   const int64_t* shape, int ndim);

/* Validate the key set, re-lay weights for the kernels (fused q|k|v, channels-last conv
 * taps, folded BatchNorm, DFT basis, rotary table, LSTM input table) and upload. */
int gam_finalize(gam_handle* h);

/* Shape helpers (host arithmetic): mel frames for L samples (preprocess.py:78-92) and
 * encoder frames for T mel frames (encoder.py:77-90). */
int64_t gam_feat_frames(const gam_handle* h, int64_t n_samples);
int64_t gam_enc_frames(const gam_handle* h, int64_t n_feat_frames);

/* FeatureExtractor.forward: wav f32 [B,L], len i64 [B] -> feat f32 [B,n_mels,T], feat_len i64 [B];
 * T = gam_feat_frames(L). */
int gam_frontend(gam_handle* h, const float* wav, const int64_t* wav_len, int B, int64_t L,
                 float* feat, int64_t* feat_len, void* stream);

/* ConformerEncoder.forward: feat f32 [B,feat_in,T], feat_len i64 [B] ->
 * encoded f32 [B,d_model,T'], enc_len i32 [B]; T' = gam_enc_frames(T). */
int gam_encode(gam_handle* h, const float* feat, const int64_t* feat_len, int B, int64_t T,
               float* encoded, int32_t* enc_len, void* stream);

/* Test hook: as gam_encode but stops after `n_layers_run` Conformer layers (0 = after
 * pre_encode, <0 = all) and, if tokens_out != NULL, also writes the token-major
 * activations f32 [B,T',d_model] at that point. */
int gam_encode_ex(gam_handle* h, const float* feat, const int64_t* feat_len, int B, int64_t T,
                  float* encoded, int32_t* enc_len, int n_layers_run, float* tokens_out, void* stream);

/* gam_encode / gam_encode_ex for a RAGGED batch whose lengths the caller also knows on the host (r06): feat_len_host[b] >= the device
 * value feat_len[b] (the same numbers in practice; NULL = exactly gam_encode_ex).  The Conformer layers then run on the batch's valid
 * frames only ("packed rows", gigaam_amd/csrc/gam_pack.h; the reference's counterpart is its optional flash-attn varlen attention,
 * gigaam/utils.py:103-155) instead of B x T'max rows, whenever that drops >= 3 % of the rows; outputs are the same as gam_encode's on every valid
 * frame (bit-identical at batch sizes without split-K), and frames behind an utterance's end in `encoded` are zero.  The host array is read
 * before the call returns; still no host synchronisation.  A device length above the host's is reported through the range-flag word (bit 1,
 * gam_range_flag / gam_range_flag_fetch): that batch's results are then incomplete.  n_layers_run / tokens_out as gam_encode_ex (-1, NULL). */
int gam_encode_varlen(gam_handle* h, const float* feat, const int64_t* feat_len, const int64_t* feat_len_host, int B, int64_t T,
                      float* encoded, int32_t* enc_len, int n_layers_run, float* tokens_out, void* stream);

/* Token rows the Conformer layers of the LAST gam_encode / _ex / _varlen call of this handle ran on; *rows_padded (may be NULL) = B x Ta, what
 * the padded layout takes.  Smaller than *rows_padded exactly when that call used packed rows. */
int gam_last_encode_rows(gam_handle* h, int* rows_padded);

/* CTCHead.forward: encoded f32 [B,d_model,T'] -> log_probs f32 [B,T',V]. */
int gam_ctc_head(gam_handle* h, const float* encoded, int B, int64_t Tp, float* log_probs, void* stream);

/* CTCGreedyDecoding.decode: -> ids i32 [B,T'], frames i32 [B,T'] (first counts[b] valid), counts i32 [B]. */
int gam_ctc_greedy(gam_handle* h, const float* encoded, const int32_t* enc_len, int B, int64_t Tp,
                   int32_t* ids, int32_t* frames, int32_t* counts, void* stream);

/* RNNTGreedyDecoding.decode: ids/frames i32 [B, T'*max_symbols], counts i32 [B].
 * Optional dump of the log-softmax of every joint evaluation, in order, per utterance:
 * logits_dump f32 [B,dump_cap,V] (may be NULL), dump_count i32 [B] (may be NULL). */
int gam_rnnt_greedy(gam_handle* h, const float* encoded, const int32_t* enc_len, int B, int64_t Tp,
                    int max_symbols, int32_t* ids, int32_t* frames, int32_t* counts,
                    float* logits_dump, int32_t* dump_count, int dump_cap, void* stream);

/* Workgroups per utterance of the cluster decode kernel behind gam_rnnt_greedy: -1 = as many as the device holds at once
 * (the default: the decode has the GPU to itself), 0 = the one-workgroup-per-utterance kernel, 1..8 = at most that many.
 * A caller that runs the decode of batch n on a side stream BESIDE the encoder of batch n+1 (the product's RNN-T
 * pipelines do, r05: model.launch_batch) asks for small clusters so that the latency-bound decode holds few CUs.  Every
 * setting holds the reference's bars (ids / frames / step counts exact on the fixtures, log-probs <= 1e-3; tests: C in
 * {0, 1, 2, 3, 5, 8}) and is bit-reproducible run to run; the setting fixes how a member partitions its sums, so two
 * DIFFERENT settings may resolve a near-tie (top-1 / top-2 margin ~1e-5) differently.  gam_rnnt_greedy is safe to run
 * concurrently with gam_frontend / gam_encode of the SAME handle on another stream (it shares no scratch with them).
 * Decode-class calls of one handle (gam_ctc_head / gam_ctc_greedy / gam_rnnt_greedy / gam_rnnt_joint) share one set of
 * scratch buffers: the library orders them itself -- a call on a stream other than the previous decode-class call's first
 * waits for that call's completion event (r06) -- so the caller may put them on any streams; they never run concurrently
 * with each other.  (Environment GAM_RNNT_CLUSTER sets the initial value, clamped to -1..8.)
 * Which stream to hand over for a decode that is to run BESIDE an encoder (r06, tools/queue_probe.py): HIP maps the streams of one
 * priority level onto four hardware queues in creation order and two streams on one queue serialise -- one stream in four created with
 * hipStreamCreate shares the null stream's queue.  Create the side stream with hipStreamCreateWithPriority at the HIGH priority (its level
 * has its own queues); the Python layer does (engine.HipEngine.aux_streams). */
int gam_set_rnnt_cluster(gam_handle* h, int workgroups_per_utterance);
/* The setting in force (-1 auto, 0 .. 8); -2 for a NULL handle. */
int gam_get_rnnt_cluster(gam_handle* h);

/* Debug aid (r05): FNV-1a hash over one of the decode's scratch buffers as it sits in device memory (synchronises the
 * device).  which: 0 = token-major copy of the encoder output, 1 = encoder projection, 2 = hand-off granules, 3 = CTC
 * logits.  (r05's overlap investigation used it to show that nothing but the decode writes these buffers.) */
int gam_debug_buffer_hash(gam_handle* h, int which, uint64_t* hash_out, int64_t* floats_out);

/* The RNN-T head taken apart (r04): the per-step entry points the reference exposes as sub-modules.  The greedy decode above
 * never calls them; they exist for callers that drive their own search or export the head.
 * gam_rnnt_predict replaces RNNTDecoder.predict (gigaam/decoder.py:85-102) for ONE step of B samples: labels i32 [B] (a value
 * < 0 is the reference's x = None: zero embedding), h_in / c_in f32 [L, B, pred_hidden] (both NULL: zero state) ->
 * g_out f32 [B, pred_hidden] (the top layer's new hidden state = the predictor output), h_out / c_out f32 [L, B, pred_hidden].
 * gam_rnnt_joint replaces RNNTJoint.joint (gigaam/decoder.py:41-47): enc f32 [B, T, d_model], dec f32 [B, U, pred_hidden]
 * -> log_probs f32 [B, T, U, num_classes] = log_softmax(W_out relu(W_enc enc + W_pred dec)).  Exact-fp32 arithmetic. */
int gam_rnnt_predict(gam_handle* h, const int32_t* labels, const float* h_in, const float* c_in, int B, float* g_out,
                     float* h_out, float* c_out, void* stream);
int gam_rnnt_joint(gam_handle* h, const float* enc, const float* dec, int B, int T, int U, float* log_probs, void* stream);

/* GigaAMEmo.get_probs after the encoder (model.py:277-283): mean over time of encoded f32 [B,d_model,T'],
 * Linear, softmax -> probs f32 [B,num_classes].  enc_len i32 [B] restricts the mean to the valid frames of
 * each utterance; NULL = all T' frames (the reference pools its single unpadded file over the whole axis). */
int gam_emo_probs(gam_handle* h, const float* encoded, const int32_t* enc_len, int B, int64_t Tp, float* probs,
                  void* stream);

/* Arithmetic of the dense contractions (every other kernel is plain fp32):
 *   GAM_GEMM_F32   -- v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fmaf chain.
 *   GAM_GEMM_F16X3 -- three-term split on v_mfma_f32_32x32x16_f16 with fp32 accumulation
 *                     (a = a_hi + a_lo, w = w_hi + w_lo; the a_lo.w_lo term, ~2^-22 relative,
 *                     is dropped): fp32-equivalent accuracy at several times the rate.
 *   GAM_GEMM_F16   -- OPT-IN speed mode (r04), never the default: ONE fp16 MFMA per product on plain-fp16 operands (the
 *                     producing kernels store each activation rounded once to fp16, "format 2"; the weights' hi plane;
 *                     a ~ fp16(a), w ~ fp16(w): 11 significant bits each), fp32 accumulation, fp32 softmax /
 *                     LayerNorm / residual stream -- the arithmetic contract of the reference's own GPU default (fp16
 *                     autocast + half() encoder, gigaam/model.py:34-37, gigaam/__init__.py:188-189), NOT that of its CPU
 *                     path: results differ from GAM_GEMM_F16X3 at the 1e-3 .. 1e-2 level in the encoder output.  Covers
 *                     the encoder GEMMs, the stem convolution and the attention products; the range guard below applies.
 *                     (Format 2 needs d_model and the FFN width to be multiples of 64; a model where they are only
 *                     multiples of 32 keeps the three-term GEMM kernels in this mode -- gam_encode and gam_op_gemm alike.)
 * Default: GAM_GEMM_F16X3 (environment GAM_GEMM_MODE=f32 | f16 selects another at gam_create).
 * The CTC / RNN-T head GEMMs and the windowed DFT always use GAM_GEMM_F32; gam_op_gemm follows the mode. */
enum { GAM_GEMM_F32 = 0, GAM_GEMM_F16X3 = 1, GAM_GEMM_F16 = 2 };
int gam_set_gemm_mode(gam_handle* h, int mode);
/* Range guard of GAM_GEMM_F16X3 / GAM_GEMM_F16.  LayerNorm-produced operands carry a per-row power-of-two scale and cannot leave
 * fp16's range; the other split-fp16 operands (FFN hidden, conv-module output, stem image and Conv2d#2 output, and the
 * attention's q / k / v -- its context is a convex combination of v) are split as they are, and a value beyond +-60000
 * there sets a device flag instead of silently becoming inf.  gam_range_flag copies the flag
 * accumulated since the last call to *flag_host (host int), clears it, and SYNCHRONISES `stream`; a caller that
 * sees 1 should repeat the batch under GAM_GEMM_F32 (the Python shim does).  Always 0 under GAM_GEMM_F32. */
int gam_range_flag(gam_handle* h, int* flag_host, void* stream);
/* The same without the host round trip: copies the accumulated flag into *flag_dev (a DEVICE int32) and clears it,
 * asynchronously on `stream`.  A caller that brings the decode counts to the host anyway (every transcribe path does)
 * appends one int to that buffer and reads both in ONE copy (gigaam_amd/engine.py does). */
int gam_range_flag_fetch(gam_handle* h, int32_t* flag_dev, void* stream);
int gam_get_gemm_mode(const gam_handle* h);

/* Raw GEMM entry for kernel-level tests and the roofline bench (arithmetic = current mode):
 * C[M,N] = act(A[M,K] . W[N,K]^T + bias) (act: 0 none, 1 SiLU, 2 ReLU); K % 32 == 0. */
int gam_op_gemm(gam_handle* h, const float* A, const float* W, const float* bias, float* C,
                int M, int N, int K, int act, void* stream);

/* Raw attention entry for kernel-level tests: q, k, v, ctx f32 [B*T, H*48] token-major
 * (head h = columns 48h..48h+47), lens i32 [B] valid keys per utterance or NULL (no mask);
 * ctx = softmax(q.k^T / sqrt(48) over keys < len).v per head (arithmetic = current mode). */
int gam_op_attention(gam_handle* h, const float* q, const float* k, const float* v, float* ctx,
                     const int32_t* lens, int B, int T, int H, void* stream);

/* Tuning hook of the large-M GEMM (tools/smallm_sweep.py): force the tile shape (mt in 2..4 rows of 64, nw in {2, 4}
 * columns of 64) and / or the split-K factor of every following launch in this process; 0 = planned per launch (default). */
int gam_tune_sp(int mt, int nw, int splitk);
/* The plan a launch of C[M,N] = A[M,K].W[N,K]^T would get on a device with n_cu compute units (host-side only: no GPU
 * needed; K % 32 == 0): tile rows = 64 * mt, tile columns = 64 * nw, split-K slices. */
int gam_plan_sp(int M, int N, int K, int n_cu, int* mt, int* nw, int* splitk);
/* The same with the fourth plan dimension (r04): the number of LDS stages of the kernel's operand pipeline -- 2 (the k-tile
 * after next lands while the current one is multiplied) or 3 (two k-tiles in flight: the 4-wave tiles of small grids, whose
 * k-tile is shorter than its fetch latency).  gam_tune_sp_stages forces it (0 = planned, 2, 3); tiles without a three-stage
 * build keep 2. */
int gam_plan_sp_ex(int M, int N, int K, int n_cu, int* mt, int* nw, int* splitk, int* stages);
int gam_tune_sp_stages(int stages);

/* Per-kernel-class HIP-event timing on the launch stream (bench.py's roofline leg).
 * gam_profile_enable(h,1) starts collecting for every launch, (h,2) for the GEMM family only
 * (fewer event packets inside a timed region), (h,0) stops; gam_profile_read synchronises the events and
 * returns, for class `cls`, the summed milliseconds, launch count and algorithmic work
 * (FLOP for GEMM/attention classes, bytes for the HBM-bound classes); it then resets. */
enum { GAM_PF_GEMM = 0, GAM_PF_CONV2 = 1, GAM_PF_ATTN = 2, GAM_PF_NORM = 3, GAM_PF_CONVMOD = 4,
       GAM_PF_STEM = 5, GAM_PF_FRONTEND = 6, GAM_PF_DECODE = 7, GAM_PF_MISC = 8, GAM_PF_NCLASS = 9 };
int gam_profile_enable(gam_handle* h, int on);
int gam_profile_read(gam_handle* h, int cls, double* ms, int64_t* launches, double* work);
/* Switch the collection level (0 / 1 / 2) WITHOUT resetting what was collected: bench.py samples every 4th timed step (an
 * event pair costs ~3 us on this runtime: 0.9 ms of a fully instrumented 33 ms step). */
int gam_profile_pause(gam_handle* h, int on);
/* algorithmic (unique operand + result) bytes of the timed launches of a GEMM class */
int gam_profile_read_bytes(gam_handle* h, int cls, double* bytes);

/* ---- multi-GPU: the path's ONE exchange -------------------------------------------------
 * Utterances are independent, so ranks (one process per GPU) decode disjoint shards with replicated weights
 * and never talk during the encoder; what is exchanged at the end are the fixed-shape decode buffers.  The
 * reference has no multi-device inference loop (gigaam/model.py:219-258 runs on one device); a binder that
 * shards that loop calls these three entry points.  Transport: RCCL (ncclAllGather over xGMI), resolved at
 * run time with dlopen("librccl.so.1") -- inside a torch process that is the RCCL torch already loaded.
 *
 *   gam_comm_unique_id   rank 0 makes the 128-byte RCCL id; the caller ships it to the other ranks by any
 *                        host channel it has (a file, MPI, torch.distributed's store, an environment variable)
 *   gam_comm_create      every rank: ncclCommInitRank on `device_id`
 *   gam_gather_ids       all-gather of index i32 [rows], counts i32 [rows], ids i32 [rows,cap], frames i32
 *                        [rows,cap] (DEVICE pointers; every rank passes the same rows / cap) into
 *                        all_* [world*rows ...], rank-major; one grouped RCCL call, asynchronous on `stream`.
 *                        `index` carries each row's global utterance number (or -1 for an unused row) so the
 *                        caller can restore its order; it may be NULL (then all_index must be NULL too).
 */
typedef struct gam_comm gam_comm;
#define GAM_COMM_ID_BYTES 128
int gam_comm_unique_id(char id_out[GAM_COMM_ID_BYTES]);
int gam_comm_create(const char id[GAM_COMM_ID_BYTES], int rank, int world, int device_id, gam_comm** out);
int gam_comm_world(const gam_comm* c);
int gam_gather_ids(gam_comm* c, const int32_t* index, const int32_t* counts, const int32_t* ids, const int32_t* frames,
                   int rows, int cap, int32_t* all_index, int32_t* all_counts, int32_t* all_ids, int32_t* all_frames,
                   void* stream);
const char* gam_comm_last_error(const gam_comm* c);
void gam_comm_destroy(gam_comm* c);

const char* gam_last_error(const gam_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* GIGAAM_HIP_H */
