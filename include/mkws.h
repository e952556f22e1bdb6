/*
 * mkws.h -- C-ABI of libmkws_hip.so: the MI355X (gfx950) hot path of multilingual_kws
 * (micro-frontend features -> EfficientNet-B0 embedding -> few-shot head).
 *
 * The reference (harvard-edge/multilingual_kws) is pure Python on TensorFlow and has no FFI; its
 * boundary for this path is a Python function surface.  Each entry point below names the reference
 * call it replaces (file:line relative to the reference repo).  INTEGRATION.md shows the ctypes
 * binding a reference maintainer would add.
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++/torch types.  Returns MKWS_OK (0) or a negative mkws_status;
 *    never throws.  mkws_last_error() returns a thread-local message for the last failure.
 *  - every `d_*` pointer is caller-owned DEVICE memory (e.g. torch tensor.data_ptr()), contiguous,
 *    row-major.  `h_*` pointers are host memory.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All *_forward / *_step
 *    calls are asynchronous on that stream, allocate nothing and never synchronise the device, so
 *    they are hipGraph-capturable.  Handles own only tables, weights and a workspace sized at create.
 *  - a handle belongs to the device that was current at create; not thread-safe; distinct handles are.
 *  - there is no CPU fallback: with no usable GPU, create fails with MKWS_ERR_NO_DEVICE.
 */
#ifndef MKWS_H_
#define MKWS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MKWS_ABI_VERSION 5

typedef enum mkws_status {
  MKWS_OK = 0,
  MKWS_ERR_INVALID_ARG = -1,   /* NULL pointer, negative size, batch > max_batch, ... */
  MKWS_ERR_UNSUPPORTED = -2,   /* configuration outside what the kernels implement */
  MKWS_ERR_NO_DEVICE = -3,     /* no HIP device / not gfx950 */
  MKWS_ERR_HIP = -4,           /* a HIP runtime call failed; see mkws_last_error() */
  MKWS_ERR_ALLOC = -5,
  MKWS_ERR_BAD_WEIGHTS = -6,   /* weight blob does not match the architecture */
  MKWS_ERR_EXCHANGE = -7       /* an in-kernel exchange of the paired whole-block kernel failed in an EARLIER forward of this
                                  handle (its embeddings are NaN); the handle has switched plans: repeat the call */
} mkws_status;

int mkws_abi_version(void);
const char* mkws_last_error(void);
/* Name of the architecture the device code was compiled for ("gfx950"). */
const char* mkws_build_arch(void);

/* ------------------------------------------------------------------------------------------------
 * Micro-frontend.  Replaces the AudioMicrofrontend op call in
 *   multilingual_kws/embedding/input_data.py:19-35  (to_micro_spectrogram)
 * including the `audio * 32768 -> int16` cast (:23) and the `* 10/256` scaling (:34), batched.
 * Field defaults = input_data.py:25-33 plus the TF Python wrapper's own defaults.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mkws_frontend_cfg {
  int32_t sample_rate;          /* 16000 */
  int32_t window_size_ms;       /* 30  (model_settings["window_size_samples"]*1000/sample_rate) */
  int32_t window_step_ms;       /* 20 */
  int32_t num_channels;         /* 40  (model_settings["fingerprint_width"]) */
  float upper_band_limit;       /* 7500 */
  float lower_band_limit;       /* 125 */
  int32_t smoothing_bits;       /* 10 */
  float even_smoothing;         /* 0.025 */
  float odd_smoothing;          /* 0.06 */
  float min_signal_remaining;   /* 0.05 */
  int32_t enable_pcan;          /* 1 */
  float pcan_strength;          /* 0.95 */
  float pcan_offset;            /* 80 */
  int32_t gain_bits;            /* 21 */
  int32_t enable_log;           /* 1 */
  int32_t scale_shift;          /* 6 */
} mkws_frontend_cfg;

typedef struct mkws_frontend mkws_frontend;

/* Fills *cfg with the defaults above. */
void mkws_frontend_default_cfg(mkws_frontend_cfg* cfg);

/* Host-only (no GPU needed): builds the integer tables for `cfg` exactly as the upstream C library
 * does and copies table `which` into dst (at most cap_bytes).  Returns the table's size in bytes,
 * or a negative mkws_status.  `which`: */
enum {
  MKWS_FT_WINDOW_COEF = 0,     /* int16[window_size] */
  MKWS_FT_TWIDDLES = 1,        /* int16[2*ncfft]  (r,i) pairs of the ncfft-point complex FFT */
  MKWS_FT_SUPER_TWIDDLES = 2,  /* int16[2*(ncfft/2)] */
  MKWS_FT_FB_WEIGHTS = 3,      /* int16[num_weights] */
  MKWS_FT_FB_UNWEIGHTS = 4,    /* int16[num_weights] */
  MKWS_FT_FB_FREQ_STARTS = 5,  /* int16[num_channels+1] */
  MKWS_FT_FB_WEIGHT_STARTS = 6,/* int16[num_channels+1] */
  MKWS_FT_FB_WIDTHS = 7,       /* int16[num_channels+1] */
  MKWS_FT_PCAN_LUT = 8,        /* int16[125] */
  MKWS_FT_LOG_LUT = 9,         /* uint16[130] */
  MKWS_FT_SCALARS = 10         /* int32[8]: window_size, window_step, fft_size, start_index, end_index,
                                  num_weights, snr_shift, correction_bits */
};
int mkws_frontend_host_table(const mkws_frontend_cfg* cfg, int which, void* dst, size_t cap_bytes);

/* Number of frames the op emits for n_samples of audio (0 if shorter than one window). */
int mkws_frontend_num_frames(const mkws_frontend_cfg* cfg, int n_samples);

/* Builds the tables on the host, uploads them once.  max_samples bounds n_samples of later calls. */
int mkws_frontend_create(const mkws_frontend_cfg* cfg, int max_samples, mkws_frontend** out);
void mkws_frontend_destroy(mkws_frontend* fe);

/* d_audio float32 [B, n_samples] in [-1,1]  ->  d_spec float32 [B, frames, channels]
 * (= raw uint16 * 10/256, what to_micro_spectrogram returns).  d_raw (optional, may be NULL) receives
 * the op's raw uint16 [B, frames, channels] for bit-exact checks. */
int mkws_frontend_forward_f32(mkws_frontend* fe, const float* d_audio, int B, int n_samples,
                              float* d_spec, uint16_t* d_raw, void* stream);
/* Same with int16 PCM input (what decode_wav holds before the /32768, input_data.py:41-45). */
int mkws_frontend_forward_i16(mkws_frontend* fe, const int16_t* d_audio, int B, int n_samples,
                              float* d_spec, uint16_t* d_raw, void* stream);

/* Streaming form of batch_streaming_analysis.py:99-117: one long recording d_audio [n_samples]
 * (float32), windows of `window_samples` every `hop_samples`; window w covers samples
 * [w*hop, w*hop + window_samples).  Emits d_spec [num_windows, frames, channels] with
 * num_windows = 1 + (n_samples - window_samples) / hop_samples... (0 if too short).  Requires
 * hop_samples to be a multiple of the frame step so per-frame FFT/filterbank work is shared
 * across overlapping windows; the noise-reduction/PCAN recurrence restarts per window exactly as
 * the reference's per-window op call does.  Returns num_windows or a negative mkws_status. */
int mkws_frontend_stream_f32(mkws_frontend* fe, const float* d_audio, int n_samples,
                             int window_samples, int hop_samples, float* d_spec, uint16_t* d_raw,
                             int max_windows, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embedding model.  Replaces `embedding.predict(x)` on the Keras model
 *   EfficientNetB0(include_top=False, weights=None, input_shape=(49,40,1)) -> GAP ->
 *   Dense 2048 relu -> Dense 2048 relu -> Dense 1024 selu ("dense_2")
 * defined at multilingual_kws/train_multilingual_embedding.py:58-83 and cut at dense_2 by
 * multilingual_kws/embedding/transfer_learning.py:36-43 / distance_filtering.py:18-27.
 * Weights arrive as one host blob of float32 tensors in Keras order/layout (HWIO conv kernels,
 * [in,out] dense kernels, BN gamma/beta/mean/var); multilingual_kws_amd/weights.py writes it.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mkws_embed mkws_embed;

/* Number of float32 values the weight blob must hold (12 967 004 params + BN statistics). */
size_t mkws_embed_weight_count(void);
/* Host-only: writes a JSON manifest (tensor name, shape, offset into the blob) into dst; returns
 * bytes needed (call with cap 0 to size). */
int mkws_embed_weight_manifest(char* dst, size_t cap_bytes);

int mkws_embed_create(const float* h_weights, size_t n_floats, int max_batch, mkws_embed** out);
void mkws_embed_destroy(mkws_embed* em);
/* d_spec float32 [B,49,40,1] (NHWC, i.e. the frontend's output) -> d_emb float32 [B,1024]. */
int mkws_embed_forward(mkws_embed* em, const float* d_spec, int B, float* d_emb, void* stream);
/* Failure contract of the paired whole-block kernel ("fuse_pair", below) and of the cluster kernel ("fuse_cluster").  Its two workgroups find each other through the
 * GPU's dispatch order (workgroups are dealt round-robin over the 8 XCDs, so linear ids b and b^8 run on one XCD, adjacent in
 * its queue).  That order is observed, probed at create and re-checked by every pair at run time, but it is NOT a documented
 * guarantee (CU masking, CPX/DPX partitions, a second process holding CUs can break it).  When a pair finds itself on two XCDs,
 * or a half waits ~0.5 s for a partner that is not resident, the launch fills its outputs with NaN and records the failure; all
 * later paired launches of the handle poison without exchanging, and the last layer of every forward that ran after the failure
 * stores NaN for EVERY embedding (NaN inside the network would not survive the ReLUs of the dense layers).  The next mkws_embed_forward / _forward_tap / _profile on the
 * handle notices (a host-mapped word, no synchronisation), switches the handle to the one-workgroup-per-4-clips kernel for
 * good ("fuse_pair" = 0, mkws_embed_get_option("pair_degraded") counts it) and returns MKWS_ERR_EXCHANGE: the earlier result is
 * invalid, the repeated call is correct.  A captured hipGraph keeps replaying the paired launch: graph users poll
 * mkws_embed_get_option(em, "exchange_error") (nonzero = a launch that already executed recorded a failure; host-mapped word, no
 * synchronisation) before a replay, and on a hit make one eager mkws_embed_forward (which heals the handle and returns
 * MKWS_ERR_EXCHANGE), repeat it, and re-capture -- multilingual_kws_amd/embedding/batch_streaming_analysis.py does exactly that --
 * or capture with "fuse_pair" = 0.  The exchange kernels assume that no OTHER kernel competes for the XCD's CUs while they spin:
 * the paired kernel tolerates concurrent launches (at most one unmatched half per launch and XCD), the cluster kernel (10-14 members)
 * does not, so handles that run concurrently (serving lanes) must not use "fuse_cluster".
 *
 * Execution options (A/B switches for measurement; results are equal up to fp32 rounding):
 *   "fuse_front" (default 1): expand 1x1 conv + depthwise conv in one kernel (expanded tensor stays in LDS);
 *                 0 = separate GEMM and depthwise kernels.
 *   "fuse_block" (default 2): blocks with 4x3 and 2x2 images (4b..7a) run expand -> depthwise -> SE -> project as
 *                 ONE kernel, 4 clips per workgroup, activations resident in LDS; 1 = only the 2x2 blocks
 *                 (6b..7a); 0 = the multi-kernel path everywhere.
 *   "fuse_chain" (default 1; needs "fuse_block" = 2; 2 / 3 = only the first / second of the two chains): the stride-1 2x2-image blocks
 *                 (6b, 6c, 6d, 7a) run as ONE paired launch (exchange 2 carries all projection tiles and both halves finish them), and
 *                 consecutive 4x3-image blocks (4b, 4c, 5a, 5b, 5c, 6a) run as ONE launch: a workgroup
 *                 keeps its clips' activations in LDS from block to block (a block's projection writes the next block's MFMA operand
 *                 fragments), the residual rides in registers, the next block's weight ring is requested during the current block's
 *                 projection.  Bit-identical to one whole-block launch per block (0).  Not used by handles that plan the cluster kernel.
 *   "fuse_top" (default 1; needs "fuse_chain" 1 or 3 and "fuse_gap"): the top conv + BN + swish + global average pool run as the LAST PHASE of
 *                 the paired chain (both halves hold block 7a's output as fragments; they split the 80 output tiles, nothing is exchanged)
 *                 instead of a launch of their own.  Bit-identical pooled features.
 *   "fuse_pair" (default 1 when the probe at create passed; needs "fuse_block"): the stride-1 2x2-image blocks (6b, 6c, 6d, 7a) run on
 *                 the PAIRED whole-block kernel: two workgroups on two CUs of one XCD share 8 clips and split the expanded channels,
 *                 so each CU streams half of the block's weights; two small in-kernel exchanges through L2.  0 = one workgroup per 4 clips.
 *                 mkws_embed_create turns it on only after a probe launch has shown that workgroups b and b^8 share an XCD on this device.
 *                 Handles with max_batch <= 512 pair 4 clips (and give the 4x3-image whole-block kernels 2 clips per workgroup) so that
 *                 every CU still gets a workgroup.
 *   "fuse_cluster" (default 1 for handles with max_batch <= 32 when the probe at create passed; settable up to 64): the tiny-image
 *                 blocks (4b..7a) run on the CLUSTER kernel: P = 10 / 14 / 12 workgroups on CUs of one XCD share one 16-row tile (one
 *                 clip of a 4x3 image, four of a 2x2 image) and split the block's expanded channels, so that a single live window no
 *                 longer waits for one CU to pull a whole block's weights (batch-1 embedding 0.44 -> 0.30 ms); two in-kernel exchanges
 *                 with generation flags (graph-replayable).  Same failure contract as the paired kernel (above).
 *   "fuse_cluster_chain" (default 1 for one-clip handles, max_batch == 1, that run "fuse_cluster"; refused on larger handles): a live window's
 *                 blocks 4b..7a run as ONE launch of the cluster kernel's body: the 120 members of all ten blocks (block k on XCD k % 8) start
 *                 together and request every weight they will need, then wait for the previous block's `done` generations; a block hands its
 *                 output on with write-through stores + one generation word per member, the next reads it with L1-bypassing loads (placement-
 *                 free); every block owns an exchange slot.  Bit-identical to one cluster launch per block (0); batch-1 embedding 0.283 ->
 *                 0.255 ms.  Taps inside the range run launch by launch.  Same failure contract; several one-clip handles may run at once
 *                 (measured up to six on six streams), but on a chip crowded by OTHER kernels members can wait for CUs until the bounded
 *                 polls give up -- then the contract above applies.
 *   "fuse_back" (default 1): blocks 2a, 2b, 3b (and 3a / 4a on handles that do not run them on the whole-block kernel: one-clip handles,
 *                 "fuse_mid" = 0) run squeeze-excite + gated projection as
 *                 ONE kernel behind the fused expand+depthwise kernel (the clip's depthwise output is staged in LDS once);
 *                 0 = se_reduce + se_expand + projection GEMM launches.
 *   "fuse_mid" (default 1): blocks with big images run expand -> depthwise -> SE ->
 *                 project as ONE kernel per block (depthwise output of all channels resident in LDS): 1 = blocks 2b, 3a and 4a
 *                 (where it measured faster than the front / back kernel pair), 2 = all of 2a..4a, 3 = 3a and 4a only, 0 = never.
 *   "fuse_rows" (default 0; 1 =): the stride-1 big-image blocks 2b and 3b run as ONE kernel per block in which a wave owns a 16-row tile
 *                 of a clip from the expand to the projection and the depthwise output stays in its registers (mbconv_rows_kernel,
 *                 csrc/mkws_embed_rows.hip: 43 / 57 KB of LDS per workgroup, so two or three workgroups share a CU); 0 = "fuse_mid" /
 *                 the front + back pair decide for those blocks as before round 6.
 *   "fuse_gemv" (default 1): handles planned for at most 4 rows (live serving: max_batch <= 4 for the dense layers, a one-clip handle for the
 *                 top conv + pool) run those layers as matrix-vector products, ONE launch per layer (a 16-wave workgroup per 16-column tile,
 *                 K split over its waves, partial columns folded in LDS in a fixed order); 0 = the MFMA GEMM + split-K fold launches.
 *   "fuse_se4" (default 1): the 4x3-image blocks (4b..6a, whole-block and chain kernels) run their squeeze-excite FCs on the 4x4x1 matrix
 *                 instruction -- a workgroup's 4 clips are exactly its N -- and the lanes that compute a gate multiply the clip's depthwise rows
 *                 by it on the spot (no gate buffer, no gate pass); 0 = the 16x16x4 weight streams (clips = 4 of 16 columns) + a gate pass.
 *                 Another summation order inside the two FCs: results agree at fp32 round-off (1e-6 of the block outputs), bit-identical per setting.
 *   "fuse_walk" (default 1): block 2a's expand + depthwise kernel runs ONE workgroup per clip that walks the clip's three 32-channel blocks
 *                 (the input is read from HBM once); 0 = one workgroup per (clip, channel block).  Bit-identical either way.
 *   "block_tiles" (set: 0 = the rule of the handle's max_batch, 1 / 2 / 3 = one / two / four clips per workgroup; get: the value in force):
 *                 row tiles per workgroup of the 4x3-image whole-block / chain kernels (blocks 4b..6a).  The rule picks the largest that
 *                 still gives every CU a workgroup: 3 from 1 024 clips, 2 for 512, 1 for 129..256-clip handles (round 6).  Another tile shape
 *                 = another summation order inside the SE sums: results agree at fp32 round-off, bit-identical per handle.
 *   "fuse_gap" (default 1): global average pool fused into the top conv epilogue (its [B*4,1280] output is never stored).
 *   "fuse_stem" (default 1): stem conv + the whole of block 1a in one kernel (persistent workgroups, one per CU, each walking
 *                 clips blockIdx, blockIdx + grid, ... with the next clip's spectrogram prefetched; both 25x20x32 activations stay
 *                 in LDS); 0 = separate kernels.
 *   "pair_fault" / "inject_exchange_error" (test hooks): force the exchange kernels' failure paths / leave the error words as a failed
 *                 exchange of an earlier launch would (tests/test_embedding_gpu.py, tests/test_streaming.py).
 *   "big_tiles" (A/B aid, default 0): 8-clip pairs and 4-clip 4x3 workgroups whatever max_batch is (what handles above 512 clips use).
 *   "plan_batch" (default max_batch; set: a clip count >= max_batch, 0 = back to max_batch): the workgroup shapes of the tiny-image kernels
 *                 (blocks 4b..7a: clips per workgroup / per pair) are those of a handle of this many clips.  For callers that run L handles
 *                 CONCURRENTLY on L streams (serving lanes): pass the clips they hold together and each of those launches takes 1 / L of the chip
 *                 instead of one clip per CU -- four 256-clip handles on four hardware queues then serve 1.03 M clips/s where one handle serves
 *                 0.56 M (profiles/r06_notes.md section 8).  A lone call on such a handle is SLOWER (0.46 -> 0.64 ms at 256 clips): leave the
 *                 option alone on handles that run by themselves.  The caller keeps lanes x 2 x ceil(max_batch / 8) within the CU count: the paired
 *                 kernels hold their CU while they wait for their partner (failure contract below).  Results: as "block_tiles". */
int mkws_embed_set_option(mkws_embed* em, const char* name, int value);
/* Current value of an option above, of "exchange_error" (see the failure contract), or of "pair_degraded" (times the handle left the paired kernel after a failed exchange) /
 * "max_batch"; negative mkws_status for an unknown name.  ("pair_fault" is a write-only test hook that forces those failures.)
 * Guard-band mode (test aid; MKWS_EMBED_GUARD=<floats> in the environment when the handle is created): the workspace starts as a NaN canary
 * pattern and every sub-buffer carved from it is followed -- the first one also preceded -- by that many floats no kernel may touch.
 * "guard_floats" / "guard_bands" = size and number of the bands (0 = mode off); "guard_violations" = SYNCHRONISES the device and counts the
 * guard words that no longer hold the canary (an out-of-range store of any kernel of the handle). */
int mkws_embed_get_option(const mkws_embed* em, const char* name);

/* Measurement aid (NOT capturable: it records a hipEvent pair around every kernel launch and
 * synchronises the stream after each of the `reps` passes).  Writes one line per launch,
 * "<stage>\t<kernel name as rocprofv3 shows it>\t<average ms>\n", into dst; returns the bytes needed. */
int mkws_embed_profile(mkws_embed* em, const float* d_spec, int B, int reps, float* d_emb, char* dst,
                       size_t cap_bytes, void* stream);

/* Debug/parity tap: runs the forward pass up to and including `stage` ("stem", "block2a_expand",
 * "block2a_dw", "block2a_gate", "block2a", ..., "top", "gap", "dense", "dense_1", "dense_2") and copies
 * that stage's output (float32, NHWC) into d_dst; returns its element count or a negative status. */
int mkws_embed_forward_tap(mkws_embed* em, const float* d_spec, int B, const char* stage, float* d_dst,
                           size_t cap_floats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Few-shot head.  Replaces Dense(18,tanh) -> Dense(3,softmax) + SparseCategoricalCrossentropy +
 * Adam of multilingual_kws/embedding/transfer_learning.py:47-59 and the per-step work of
 * xfer.fit (:86-93) on the frozen embedding.
 * Parameter vector layout (float32, P = in*hid + hid + hid*cls + cls = 18 507 for 1024/18/3):
 *   W1[in,hid] | b1[hid] | W2[hid,cls] | b2[cls]     (Keras [in,out] kernels)
 * ---------------------------------------------------------------------------------------------- */
typedef struct mkws_head mkws_head;

int mkws_head_create(int in_dim, int hidden, int classes, int max_batch, mkws_head** out);
void mkws_head_destroy(mkws_head* hd);
int mkws_head_param_count(const mkws_head* hd);
/* Device pointers into the handle's own state (valid until destroy): params, grads, Adam m, Adam v.
 * Exposed so the data-parallel host code can all-reduce the flat gradient with RCCL in place.
 * The gradient buffer holds mkws_head_grad_count() = P + 2 floats: the P gradients followed by the two
 * statistics of the last mkws_head_loss_grad call {sum of per-row loss, number of correct argmax rows}, so
 * that one data-parallel step is ONE all-reduce(sum) over that range. */
float* mkws_head_params(mkws_head* hd);
float* mkws_head_grads(mkws_head* hd);
int mkws_head_grad_count(const mkws_head* hd);
/* The handle's whole optimizer state is ONE contiguous range starting at mkws_head_params(): params | grads (+ 2 statistics) | Adam m |
 * Adam v, each padded to the same length; this is the length of the range in floats (snapshot / restore around a warm-up step). */
int mkws_head_state_floats(const mkws_head* hd);
/* Uploads h_params (host) and zeroes the gradient and Adam state, ordered on `stream`; returns after the
 * stream has drained (h_params may be pageable). */
int mkws_head_set_params(mkws_head* hd, const float* h_params, int n, void* stream);
int mkws_head_get_params(mkws_head* hd, float* h_params, int n, void* stream);
/* d_emb [B,in] -> d_probs [B,classes] (softmax probabilities, what model.predict returns). */
int mkws_head_forward(mkws_head* hd, const float* d_emb, int B, float* d_probs, void* stream);
/* Multi-keyword serving (batch_streaming_analysis.py runs one full model per keyword; here N keywords share one
 * embedding pass): n_heads heads of equal dimensions over the same d_emb [B,in] in ONE launch (per 64 heads)
 * -> d_probs [n_heads, B, classes].  `heads` is a host array of handles. */
int mkws_heads_forward(mkws_head* const* heads, int n_heads, const float* d_emb, int B, float* d_probs, void* stream);
/* Forward + mean sparse-CE loss + backward into the handle's grad buffer (gradient of the MEAN loss
 * over these B rows).  d_labels int32 [B].  d_stats float32[2] (optional, may be NULL) receives {sum of
 * per-row loss, number of correct argmax predictions} for these B rows; the same two values are always
 * written behind the gradient (mkws_head_grads()[P], [P+1]). */
int mkws_head_loss_grad(mkws_head* hd, const float* d_emb, const int32_t* d_labels, int B,
                        float* d_stats, void* stream);
/* After mkws_head_loss_grad on the same B rows: d(mean loss)/d(embedding rows) -> d_dx [B, in]  (what
 * backprop_into_embedding=True propagates into the embedding network, transfer_learning.py:94-112). */
int mkws_head_input_grad(mkws_head* hd, float* d_dx, int B, void* stream);
/* Keras Adam update from the grad buffer (grad is multiplied by grad_scale first, e.g. 1/world_size
 * after a sum all-reduce): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps). */
int mkws_head_adam_step(mkws_head* hd, float lr, float beta1, float beta2, float eps, int step_t,
                        float grad_scale, void* stream);
/* The same update with the step index t read from device memory (*d_step >= 1, see mkws_op_step_inc): a captured hipGraph
 * replays ONE launch for every step, so t cannot be a launch argument there. */
int mkws_head_adam_step_dev(mkws_head* hd, float lr, float beta1, float beta2, float eps, const int* d_step,
                            float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training-batch assembly.  Replaces the per-clip tf.data map of AudioDataset.augment /
 * random_timeshift / random_background_sample / add_background
 * (multilingual_kws/embedding/input_data.py:141-157,227-304) and spec_augment (:306-369).
 * The random draws are made by the host (one item per clip); the sample-level work runs on device.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mkws_augment_item {
  int32_t mode;     /* 0: out = shift(fg)            (random_timeshift, :245-268)
                       1: out = bg_slice * bg_vol    (silence branch, :284-287 / :227-243)
                       2: out = clip(shift(fg) + bg_slice * (rms(fg)/rms(bg) or 0) * bg_vol, -1, 1)
                          (add_background, :141-157) */
  int32_t bank;     /* foreground bank: 0 = d_bank0 (target clips), 1 = d_bank1 (unknown-word clips) */
  int32_t src;      /* row of the foreground clip in that bank */
  int32_t shift;    /* out[t] = fg[t - shift], zero-filled (positive = delay) */
  int32_t bg_idx;   /* background track */
  int32_t bg_off;   /* first sample of the n_samples-long background slice */
  float bg_vol;
  int32_t reserved;
} mkws_augment_item;

/* d_bank0 [n0, n_samples], d_bank1 [n1, n_samples] (may be NULL), d_bg [tracks, bg_stride] float32;
 * d_items [B]; d_out [B, n_samples]. */
int mkws_augment_batch(const float* d_bank0, const float* d_bank1, const float* d_bg, int64_t bg_stride,
                       const mkws_augment_item* d_items, int B, int n_samples, float* d_out, void* stream);
/* SpecAugment masking in place on d_spec [B, frames, channels]; d_masks int32 [B,8] =
 * {freq0 start, size, freq1 start, size, time0 start, size, time1 start, size}; size 0 = no mask. */
int mkws_specaug_apply(float* d_spec, const int32_t* d_masks, int B, int frames, int channels, void* stream);
/* The same with ANY number of masks per axis (the reference loops frequency_n / time_n times for whatever SpecAugParams says,
 * input_data.py:317-362): d_masks int32 [B, 2*(n_freq + n_time)] = n_freq x {channel start, size} then n_time x {frame start, size}. */
int mkws_specaug_apply_n(float* d_spec, const int32_t* d_masks, int n_freq, int n_time, int B, int frames, int channels, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training operators for `backprop_into_embedding=True` (multilingual_kws/embedding/transfer_learning.py:94-112).
 * The reference un-freezes the whole nested EfficientNet, so Keras runs xfer.fit with training=True through
 * keras/applications/efficientnet.py: BatchNormalization on batch statistics (+ moving-average updates), per-block
 * drop-connect, gradients into every kernel / bias / gamma / beta, Adam.  Keras owns the graph there; here the tape
 * is host code (multilingual_kws_amd/embedding_trainer.py) and each numerical operator is one entry point below.
 * All tensors are device float32, NHWC viewed as row-major [M, C]; parameters keep their Keras layouts (1x1 conv
 * kernel = [K, N], depthwise [kh, kw, C], dense [in, out]).  `act`: 0 none, 1 swish, 2 relu, 3 selu, 4 sigmoid.
 * Cross-workgroup reductions are two-level and FIXED-ORDER (partial sums in a caller-provided scratch arena, folded in index
 * order by a second launch): no atomics, a step with the same inputs is bit-reproducible.
 * ---------------------------------------------------------------------------------------------- */
/* State of the training operators = a CONTEXT: the scratch arena (device memory, caller-owned) for those partial sums and the queue of
 * deferred second stages (mkws_op_fold_defer).  16 Mi floats cover every layer of the network at any batch (the largest user is the
 * split reduction of mkws_op_gemm, which splits only as far as the arena reaches); ops that need the arena fail with
 * MKWS_ERR_INVALID_ARG when it is missing or too small.  A context is used in stream order (ops on ONE stream share it safely; give
 * concurrent streams separate contexts).
 *   mkws_train_ctx_create / _destroy   a context handle owning nothing but its bookkeeping (the arena stays the caller's)
 *   mkws_train_ctx_bind(ctx)           every mkws_op_* call of THIS host thread uses ctx until another bind; NULL = the thread's default
 *                                      context.  Distinct contexts are independent: two trainers on one thread each bind their own before
 *                                      their calls, a trainer handed to another thread binds there and finds its queue as it left it.
 *   mkws_op_set_scratch                the thread's DEFAULT context (rounds 2-3 interface): sets its arena and binds it. */
typedef struct mkws_train_ctx mkws_train_ctx;
int mkws_train_ctx_create(float* d_scratch, size_t floats, mkws_train_ctx** out);
void mkws_train_ctx_destroy(mkws_train_ctx* ctx);
int mkws_train_ctx_bind(mkws_train_ctx* ctx);
int mkws_op_set_scratch(float* d_scratch, size_t floats);
/* C[M,N] (+)= op(A)[M,K] . op(B)[K,N] on the fp32 MFMA; op(A)(m,k) = transA ? A[k*lda+m] : A[m*lda+k], likewise B.
 * ksplit > 1 splits K over workgroups (slice sums go to the scratch arena, a second launch folds them in order into C);
 * ksplit = 0 picks the split from the shapes and the arena size (small grids with a long K: ~512 workgroups). */
int mkws_op_gemm(const float* d_A, const float* d_B, float* d_C, int M, int N, int K, int lda, int ldb, int ldc, int transA, int transB,
                 int accumulate, int ksplit, void* stream);
/* Process-wide switches of the training operators (ABI 5; A/B aids, results equal up to fp32 summation order):
 *   "gemm_ring"    (default 1; initial value from MKWS_TRAIN_GEMM2): NN / NT GEMMs whose K, N and leading dimensions are multiples of 4 run on the
 *                  register-ring kernel (operands global / L2 -> registers through buffer loads, no LDS, no barriers: the scheme of the inference
 *                  GEMM); 0 = the LDS-staged kernel for everything.
 *   "gemm_ring_tn" (default 0; MKWS_TRAIN_GEMM_TN2): the same for weight gradients (transA = 1): 0 none, 1 small outputs over >= 4096 rows, 2 all.
 *                  Faster per launch and in a single-stream step, slower inside the trainer's two-stream step (profiles/r05_notes.md).
 * mkws_op_get_option returns the value, or a negative mkws_status for an unknown name. */
int mkws_op_set_option(const char* name, int value);
int mkws_op_get_option(const char* name);
/* Stream ordering for a trainer that spreads its launches over two streams (weight gradients next to the input-gradient chain): everything queued
 * on signalling_stream so far happens before whatever is queued on waiting_stream afterwards.  Events belong to the bound context.  Capturable. */
int mkws_op_stream_wait(void* waiting_stream, void* signalling_stream);
/* Deferred second stages.  With enable = 1 the fixed-order folds whose results only the optimizer (or a gradient all-reduce) reads -- the
 * split reduction of a weight-gradient GEMM (transA = 1), the bias gradient of mkws_op_bias_act_bwd, the weight gradients of
 * mkws_op_dwconv_bwd / mkws_op_stem_bwd_weight / mkws_op_se_wgrad (batches above 64 rows) -- are QUEUED (their partial sums stay in the scratch arena, handed out from a bump pointer)
 * and run as ONE launch at mkws_op_fold_flush, at enable = 0, or when the queue (24 entries) or the arena is full.  Same sums in the same
 * order: bit-identical results, ~90 launches fewer per training step.  Until the flush those outputs are not final.  Per host thread, like
 * the scratch arena; capturable. */
int mkws_op_fold_defer(int enable, void* stream);
int mkws_op_fold_flush(void* stream);
/* Dense / SE layer forward in one call: Z = X[M,K] . W[K,N] (kept: the backward pass differentiates the activation at Z + bias) and
 * A = act(Z + bias), the epilogue fused into the GEMM (or into its split-reduction fold). */
int mkws_op_dense_fwd(const float* d_X, const float* d_W, const float* d_bias, int act, float* d_Z, float* d_A, int M, int N, int K, void* stream);
/* Batch statistics of Z [M,C] per channel: mean, biased variance (per-chunk mean / M2, combined with Chan's update). */
int mkws_op_bn_stats(const float* d_Z, int M, int C, float* d_mean, float* d_var, void* stream);
/* Training-mode BatchNormalization forward in two launches: chunk statistics, then (fold of the chunks +) moving-average update
 * (moving = momentum * moving + (1 - momentum) * batch, variance Bessel-corrected) + A = act(gamma * xhat + beta).
 * d_mean / d_var receive the batch statistics the backward pass needs. */
int mkws_op_bn_train_fwd(const float* d_Z, int M, int C, const float* d_gamma, const float* d_beta, float eps, int act, float momentum,
                         float* d_moving_mean, float* d_moving_var, float* d_mean, float* d_var, float* d_A, void* stream);
/* The same with the residual branch of an MBConv block in the second launch (reference: the `layers.add([x, inputs])` behind Keras'
 * drop-connect Dropout(noise_shape=(None,1,1,1)), train_multilingual_embedding.py:58-83 via keras/applications/efficientnet.py):
 * A[r] = keep[r / group] * act(BN(Z)[r]) + d_res[r];  d_row_scale may be NULL (keep = 1), d_res NULL = plain mkws_op_bn_train_fwd. */
int mkws_op_bn_train_fwd_res(const float* d_Z, int M, int C, const float* d_gamma, const float* d_beta, float eps, int act, float momentum,
                             float* d_moving_mean, float* d_moving_var, float* d_mean, float* d_var, float* d_A, const float* d_res,
                             const float* d_row_scale, int group, void* stream);
/* A = act(gamma * (Z - mean) / sqrt(var + eps) + beta) */
int mkws_op_bn_act_fwd(const float* d_Z, const float* d_mean, const float* d_var, const float* d_gamma, const float* d_beta, float eps, int act,
                       float* d_A, int M, int C, void* stream);
/* Backward of the above through the batch statistics: d_dA [M,C] holds dLoss/dA on entry and dLoss/dZ on return;
 * d_dgamma / d_dbeta [C] are written; d_scratch: 2*C floats. */
int mkws_op_bn_act_bwd(const float* d_Z, const float* d_mean, const float* d_var, const float* d_gamma, const float* d_beta, float eps, int act,
                       float* d_dA, float* d_dgamma, float* d_dbeta, float* d_scratch, int M, int C, void* stream);
/* The same with the incoming gradient assembled on the fly (saves the launches that used to build it):
 *   dLoss/dA[r] = d_src[r] * d_row_scale[r / group] + d_bcast[r / group][:] * bscale      (each term optional; at least one of src / bcast)
 * e.g. the drop-connect scale of a residual block (src = gradient of the block output, kept intact for the shortcut), or the squeeze-excite
 * mean's gradient broadcast over the pixels (src = d_dA, bcast = dLoss/dmean, bscale = 1 / HW).  d_dA receives dLoss/dZ. */
int mkws_op_bn_act_bwd_ex(const float* d_Z, const float* d_mean, const float* d_var, const float* d_gamma, const float* d_beta, float eps, int act,
                          float* d_dA, const float* d_src, const float* d_row_scale, const float* d_bcast, float bscale, int group, float* d_dgamma,
                          float* d_dbeta, int M, int C, void* stream);
/* moving = momentum * moving + (1 - momentum) * batch; the variance enters Bessel-corrected (M/(M-1)) as in Keras' fused BN. */
int mkws_op_bn_update_moving(float* d_moving_mean, float* d_moving_var, const float* d_mean, const float* d_var, float momentum, int M, int C, void* stream);
/* 1x1 convolution Z [M,N] = X [M,K] W [K,N] followed by training-mode BatchNorm (+ activation, + residual branch) -- mkws_op_gemm +
 * mkws_op_bn_train_fwd_res as ONE operator, so that the GEMM's epilogue can leave the BatchNorm's chunk statistics (one chunk per 64-row tile)
 * when it is unsplit and has at most 160 row tiles: two launches instead of three and one pass less over Z.  Same arguments and results
 * (statistics up to summation order) as the two calls; Z keeps the raw convolution output for the backward pass. */
int mkws_op_conv_bn_fwd(const float* d_X, const float* d_W, float* d_Z, int M, int N, int K, const float* d_gamma, const float* d_beta, float eps, int act,
                        float momentum, float* d_moving_mean, float* d_moving_var, float* d_mean, float* d_var, float* d_A, const float* d_res,
                        const float* d_row_scale, int group, void* stream);
/* The same for the depthwise convolution (mkws_op_dwconv_fwd + mkws_op_bn_train_fwd): up to 256 chunks of 128 output rows the convolution launch
 * leaves the chunk statistics itself. */
int mkws_op_dwconv_bn_fwd(const float* d_X, const float* d_W, float* d_Z, int B, int H, int W, int C, int k, int stride, int pad_top, int pad_left, int Ho, int Wo,
                          const float* d_gamma, const float* d_beta, float eps, int act, float momentum, float* d_moving_mean, float* d_moving_var, float* d_mean,
                          float* d_var, float* d_A, void* stream);
/* Depthwise k x k conv (k = 3, 5; stride 1, 2) with explicit top / left padding (Keras "same" or correct_pad), raw output. */
int mkws_op_dwconv_fwd(const float* d_X, const float* d_W, float* d_Z, int B, int H, int W, int C, int k, int stride, int pad_top, int pad_left, int Ho,
                       int Wo, void* stream);
/* d_dX (may be NULL) <- input gradient, d_dW [k,k,C] (may be NULL; not both) <- weight gradient: two calls may split them over two streams. */
int mkws_op_dwconv_bwd(const float* d_X, const float* d_W, const float* d_dZ, float* d_dX, float* d_dW, int B, int H, int W, int C, int k, int stride,
                       int pad_top, int pad_left, int Ho, int Wo, void* stream);
/* Stem: Rescaling(1/255) + Normalization + ZeroPadding2D(((1,1),(0,1))) + Conv2D(32,3,s2,valid) on [B,49,40] -> raw [B,25,20,32]. */
int mkws_op_stem_fwd(const float* d_spec, const float* d_W, float norm_mean, float norm_std, float* d_Z, int B, void* stream);
int mkws_op_stem_bwd_weight(const float* d_spec, const float* d_dZ, float norm_mean, float norm_std, float* d_dW, int B, void* stream);
/* mean over HW: A [B,HW,C] -> [B,C] */
int mkws_op_pool_hw(const float* d_A, float* d_mean, int B, int HW, int C, void* stream);
/* out[b,hw,c] = A[b,hw,c] * g[b,c]   (SE excite) */
int mkws_op_scale_channels(const float* d_A, const float* d_g, float* d_out, int B, int HW, int C, void* stream);
/* backward of the excite multiply: dA = dOut * g,  dg[b,c] = sum_hw dOut * A */
int mkws_op_se_bwd(const float* d_A, const float* d_g, const float* d_dOut, float* d_dA, float* d_dg, int B, int HW, int C, void* stream);
/* The squeeze-excite branch of one MBConv block (Keras: GlobalAveragePooling2D -> Conv2D(se, swish) -> Conv2D(C, sigmoid) -> multiply) in two
 * launches, one workgroup per (clip, 128-channel slab):  mean [B,C] = pool(A [B,HW,C]);  Yr [B,se] = mean Wr + br;  R = swish(Yr);
 * G [B,C] = sigmoid(R We + be);  out = A * G.  Wr [C,se], We [se,C] are the Keras 1x1 kernels.  C % 4 == 0, C <= 1152, se <= 48.
 * d_work: B * ceil(C / 128) * se floats of scratch (the per-slab partial sums; folded in slab order).  Same results as pool_hw + dense_fwd x 2 +
 * scale_channels up to summation order. */
int mkws_op_se_fwd(const float* d_A, const float* d_Wr, const float* d_br, const float* d_We, const float* d_be, float* d_mean, float* d_Yr, float* d_R, float* d_G,
                   float* d_out, float* d_work, int B, int HW, int C, int se, void* stream);
/* Its backward in three launches: given dOut = dLoss/d(out):  dA = dOut * G (the multiply's direct path; the squeeze's path comes back as dmean [B,C]
 * = dLoss/d(mean), which mkws_op_bn_act_bwd_ex spreads over the pixels), and the gradients of the four parameter tensors (written, not accumulated;
 * batch sums in row order).  dYg [B,C] / dYr [B,se] are outputs too (the pre-activation gradients); d_work as above. */
int mkws_op_se_bwd_fused(const float* d_A, const float* d_G, const float* d_dOut, const float* d_mean, const float* d_Yr, const float* d_R, const float* d_Wr,
                         const float* d_We, float* d_dA, float* d_dmean, float* d_dYg, float* d_dYr, float* d_dWr, float* d_dbr, float* d_dWe, float* d_dbe,
                         float* d_work, int B, int HW, int C, int se, void* stream);
/* The parameter-gradient launch alone (mkws_op_se_bwd_fused with the four gradient pointers NULL skips it): nothing downstream waits for it, so a
 * trainer may run it on a second stream next to the input-gradient chain.  Batches above 64 rows are summed in chunks of 64 rows whose partial
 * sums (context scratch arena) are folded in chunk order -- deferrable like the other weight-gradient folds (mkws_op_fold_defer). */
int mkws_op_se_wgrad(const float* d_mean, const float* d_R, const float* d_dYg, const float* d_dYr, float* d_dWr, float* d_dbr, float* d_dWe, float* d_dbe, int B,
                     int C, int se, void* stream);
/* X[b,hw,c] += v[b,c] * scale   (backward of a mean over HW) */
int mkws_op_add_bcast(float* d_X, const float* d_v, float scale, int B, int HW, int C, void* stream);
/* A = act(Z + bias);  backward: d_dA <- dA * act'(Z + bias) in place, d_dbias [N] <- its column sums */
int mkws_op_bias_act_fwd(const float* d_Z, const float* d_bias, int act, float* d_A, int M, int N, void* stream);
int mkws_op_bias_act_bwd(const float* d_Z, const float* d_bias, int act, float* d_dA, float* d_dbias, int M, int N, void* stream);
/* out[b,:] = a[b,:] * row_scale[b] (+ c[b,:] if c != NULL): drop-connect (keep / (1 - rate)) + residual add, and its backward */
int mkws_op_row_scale_add(const float* d_a, const float* d_row_scale, const float* d_c, float* d_out, int B, int64_t per_row, void* stream);
int mkws_op_axpy(float* d_y, const float* d_x, float alpha, int64_t n, void* stream);
/* Keras Adam over a flat buffer (same arithmetic as mkws_head_adam_step). */
int mkws_op_adam(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2, float eps, int step_t,
                 float grad_scale, void* stream);
/* Sparse categorical cross-entropy from logits over N classes (the classifier the reference trains the embedding with,
 * train_multilingual_embedding.py:84-93): d_logits [B,N] is overwritten with d(mean loss)/d(logits); d_rowstat [B,2] receives
 * {-log softmax(z)[y], argmax(z) == y} per row and d_stats [2] their sums (rows folded in index order). */
int mkws_op_softmax_ce(float* d_logits, const int32_t* d_labels, int B, int N, float* d_rowstat, float* d_stats, void* stream);
/* Graph-replayable form: the step index lives in device memory.  mkws_op_step_inc adds 1 to *d_step (once per optimizer step,
 * before the Adam launches that share the counter); mkws_op_adam_dev computes lr_t from *d_step on the device. */
int mkws_op_step_inc(int* d_step, void* stream);
int mkws_op_adam_dev(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2, float eps,
                     const int* d_step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MKWS_H_ */
