/*
 * lipreading_hip.h — C ABI of the MI355X (gfx950) hot path of joseph-zhong/LipReading:
 * landmark step -> recurrent sequence encoder -> CTC loss / greedy decode.
 *
 * The reference has no FFI of its own (it is pure Python over stock torch ops), so this
 * ABI is the build's design; every entry point names the reference symbol (file:line under
 * the reference root) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - every function only ENQUEUES work on `stream` (a hipStream_t passed as void*); it never
 *     allocates, frees, synchronises or copies to the host, so a whole training step can be
 *     captured in one hipGraph.  TWO exceptions, both outside any captured region:
 *       (1) the per-device fault words (16 bytes: {recurrence timed out, 3 spare}) are
 *           hipMalloc'ed + hipMemset the FIRST time any entry point needs them on a device;
 *           call lr_fault_words_ptr() once per device before capturing (lipreading_amd.train's
 *           StepGraphs does: its first steps of a shape run eagerly) — afterwards nothing
 *           allocates again for the life of the process;
 *       (2) the host-side readers lr_rnn_pair_errors / lr_profile_read
 *           synchronise the device and copy a few words back: diagnostics, never on a step;
 *   - scratch comes from the caller: ask `*_workspace_bytes`, pass the buffer back in;
 *   - return value: LR_OK (0) or a negative lr_status; no C++ exception crosses the boundary;
 *   - all floating point is IEEE fp32 (subnormals kept), all indices int32 unless noted.
 */
#ifndef LIPREADING_HIP_H
#define LIPREADING_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lr_stream_t; /* hipStream_t */

typedef enum lr_status {
  LR_OK = 0,
  LR_ERR_INVALID_ARG = -1, /* null pointer, negative size, unsupported shape            */
  LR_ERR_WORKSPACE = -2,   /* caller workspace smaller than *_workspace_bytes()          */
  LR_ERR_LAUNCH = -3,      /* hipLaunchKernel / hipGetLastError reported a failure       */
  LR_ERR_UNSUPPORTED = -4, /* valid request outside what the kernels implement           */
  LR_ERR_NO_DEVICE = -5    /* no gfx950 device visible                                   */
} lr_status;

typedef enum lr_rnn_mode {
  LR_RNN_GRU = 0, /* torch.nn.GRU  gate order r,z,n   (3 gates) */
  LR_RNN_LSTM = 1, /* torch.nn.LSTM gate order i,f,g,o (4 gates) */
  LR_RNN_TANH = 2, /* torch.nn.RNN (nonlinearity='tanh'): h' = tanh(W_ih x + b_ih + W_hh h + b_hh) (1 "gate");
                      better_model.py:9 allows rnn_type='RNN' */
  LR_RNN_CELL_MASK = 0xff,
  /* Optional flags OR-ed into `mode` of lr_rnn_layer_* / lr_rnn_*_bytes (build-defined, used by
   * the pixel regime only; the reference-faithful path passes none and stays on exact fp32 MFMA):
   * contract the layer's INPUT projection (x W_ih^T, and its two gradients) on the bf16 matrix
   * cores with every fp32 operand split into bf16 hi + lo terms (~1e-5 relative, lr_xgemm). */
  LR_RNN_PROJ_BF16X3 = 0x100,
  /* with LR_RNN_PROJ_BF16X3: the layer input is the bf16 conv frontend's output — every element
   * of x is already a bf16 value, so x needs no lo term, and dx goes to a bf16 consumer (the
   * frontend's backward), so dx is contracted from the hi terms only. */
  LR_RNN_INPUT_BF16_EXACT = 0x200,
  /* (0x400: rounds 1-4's single-plane bf16 recurrence, LR_RNN_RECUR_BF16 — removed in round 5: no shipped
   * configuration ran it once the fp32-faithful one-launch kernels below covered every size) */
  /* with LR_RNN_PROJ_BF16X3 | LR_RNN_INPUT_BF16_EXACT and I % 8 == 0: x (and dx) are STORED as bf16
   * matrices [B*T][I] — the conv frontend's features as they are; x is then the hi plane of the
   * projection's operand and is never converted or packed. */
  LR_RNN_INPUT_STORED_BF16 = 0x800,
  /* The recurrence of a supported layer (lr_rnn_pair_supported != 0: GRU or LSTM, H % 4 == 0, H up to the largest
   * cluster — the reference's config/defaults.txt:19-21, config/train/attn/attention_type:16-19, config/train/micro:6-8
   * and the decoders behind them, better_model.py:134-148) as ONE launch per pass and fp32-FAITHFUL: W_hh and the
   * carried state are split into bf16 hi + lo planes, all four cross terms accumulate in fp32 on the bf16 matrix cores
   * (~1e-6 of the exact fp32 product).  W_hh stays in the registers + LDS of a cluster of ceil(H / 32) (past 864 /
   * 768 units: ceil(H / 16)) compute units per (direction, 8 samples), which exchange their slices of the state
   * (forward) / their partial state gradients (backward) once per step (lr_rnn_cluster.hip).  What the
   * reference-faithful regime runs by default where it is supported; the decoder loop (lr_decoder_forward /
   * _backward) takes the same cluster kernels, started from the encoder's final state, when every step is
   * teacher forced. */
  LR_RNN_RECUR_SPLIT = 0x1000,
  /* with LR_RNN_PROJ_BF16X3: keep only the bf16 hi plane of EVERY operand of the input projection, its two
   * gradients and dW_hh — one product each instead of two or three (weights and gate gradients rounded to bf16
   * once per step, ~2^-9 relative).  An experiment of the build-defined pixel regime (encoder.input_projection =
   * 'bf16x1'); measured against the oracle in bench.py's parity block, not a default. */
  LR_RNN_PROJ_BF16X1 = 0x2000
} lr_rnn_mode;

typedef enum lr_ctc_reduction {
  LR_CTC_SUM = 0, /* reference reduction='sum'  (src/train/ctc_loss.py:114) */
  LR_CTC_MEAN = 1 /* reference reduction='mean' incl. its run-weighting quirk (:74,:103-105) */
} lr_ctc_reduction;

/* ---- library ------------------------------------------------------------------------- */

/* ABI version: major*10000 + minor*100 + patch. */
int lr_version(void);
/* Static string for a status code (never NULL). */
const char* lr_status_string(int status);
/* Number of visible HIP devices (0 when none); does not create a context. */
int lr_device_count(void);

/* ---- A1: batch collation — src/data/data_loader.py:117-152 (_collate_fn, _pad) --------- */

/* Zero-pad ragged rows to the batch maximum on the device.
 *   packed   [sum(lens), feat]   rows of all samples back to back (sample b starts at
 *                                row offsets[b])
 *   offsets  [B] int64, lens [B] int32
 *   out      [B, t_max, feat]    out[b,t,:] = t < lens[b] ? packed[offsets[b]+t,:] : 0
 * Replaces the per-sample torch.Tensor(seq) copy loop of _pad (:139-141). */
int lr_collate_pad_f32(const float* packed, const int64_t* offsets, const int32_t* lens,
                       float* out, int B, int t_max, int feat, lr_stream_t stream);

/* ---- A7: landmark step — src/utils/data/face.py:76-90,164-175 -------------------------- */

/* _applyPadding (face.py:76-90) for n rectangles, integer arithmetic identical to the
 * reference: left = max(0, left - int(padding*box_w)) ... ; rect layout (left,right,top,bottom).
 *   rects_in/out [n,4] int32, dims [n,2] int32 = (img_h, img_w). */
int lr_lmk_apply_padding(const int32_t* rects_in, const int32_t* dims, int32_t* rects_out,
                         int n, float padding, lr_stream_t stream);

/* getFace (face.py:164-175): out[i,p,0] = in[i,p,0]-left_i; out[i,p,1] = in[i,p,1]-top_i;
 * z untouched.  lmk [n, npts, 3] f32, rects [n,4] int32 (left,right,top,bottom).  in==out ok. */
int lr_lmk_translate(const float* lmk_in, const int32_t* rects, float* lmk_out, int n,
                     int npts, lr_stream_t stream);

/* PRNet crop geometry (src/models/face/prnet.py:112-119,136-140, image_info = the UNPADDED face rect
 * as generate_dataview.py:62 passes it): old = (r-l+b-t)/2, centre of the box, size = int(old*1.6),
 * similarity transform of the crop square onto resolution x resolution (what the reference obtains
 * from skimage.transform.estimate_transform('similarity', ...)).
 *   rects [n,4] int32 (left,right,top,bottom) -> tform [n,3,3] float64 row-major, sizes [n] int32 (NULL ok) */
int lr_lmk_crop_transform(const int32_t* rects, double* tform, int32_t* sizes, int n, int resolution,
                          lr_stream_t stream);

/* Restore (prnet.py:150-156): the network's position map of the crop -> image coordinates:
 * z /= tform[0][0]; [x y] = (tform^-1 [x y 1])[0:2].
 *   cropped_pos [n, npix, 3] f32 (PosPrediction.predict's output, prnet.py:304-309), tform [n,3,3] f64
 *   -> pos [n, npix, 3] f64 (np.dot's result type). */
int lr_lmk_restore(const float* cropped_pos, const double* tform, double* pos, int n, int64_t npix,
                   lr_stream_t stream);

/* get_landmarks (prnet.py:162-170): kpt[i,k,:] = pos[i, uv[1][k], uv[0][k], :]; with `rects` non-NULL
 * also getFace's translation by (left, top) of rects[i] (face.py:164-175).
 *   pos [n,res,res,3] f64, uv_kpt_ind [2,K] int32, rects [n,4] int32 or NULL -> kpt [n,K,3] f64 */
int lr_lmk_gather(const double* pos, const int32_t* uv_kpt_ind, const int32_t* rects, double* kpt, int n,
                  int resolution, int K, lr_stream_t stream);

/* The arithmetic of _gen_data (src/scripts/generate_dataview.py:58-64) around the two networks, in one
 * launch: padded rect = _applyPadding(dims, rect, padding) (extractFace, :61); the UNPADDED rect
 * defines the PRNet crop (:62); the K landmark points are gathered from the position map, restored
 * to image coordinates and translated by the PADDED rect (:64).
 *   cropped_pos [n,res,res,3] f32, rects [n,4], dims [n,2] = (img_h, img_w), uv_kpt_ind [2,K]
 *   -> lmk_f64 [n,K,3] and/or lmk_f32 [n,K,3] (either may be NULL), rects_padded [n,4] (NULL ok) */
int lr_lmk_landmarks(const float* cropped_pos, const int32_t* rects, const int32_t* dims, double padding,
                     const int32_t* uv_kpt_ind, double* lmk_f64, float* lmk_f32, int32_t* rects_padded,
                     int n, int resolution, int K, lr_stream_t stream);

/* A9 (BUILD-DEFINED, no reference symbol: face.py:21 defines `_mouth = slice(48,68)` and never uses
 * it): crop the mouth region and resample it to S x S.  Per frame: bounding box of landmarks
 * [lo,hi) (x = lmk[..,0], y = lmk[..,1], image pixels) -> centre, side = max(w,h)*(1+2*margin)
 * (>= 2 px) -> bilinear resize with half-pixel centres and edge clamping, rounded to uint8.
 *   frames uint8 [n][3][H][W], lmk f32 [n][npts][3], out uint8 [n][3][S][S]. */
int lr_lip_crop_u8(const void* frames, const float* lmk, void* out, int n, int H, int W, int S, int npts,
                   int lo, int hi, float margin, lr_stream_t stream);

/* ---- dense fp32 contraction (MFMA f32 16x16x4 / 32x32x2), used by A3 ------------------- */

/* C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C + bias[N]   (row-major, fp32)
 *   transA=0: A is [M,K] lda>=K ; transA=1: A is [K,M] lda>=M   (same for B with N)
 *   bias may be NULL.  beta==0 never reads C.
 *   row_shift/period: when period>0, row k of the K dimension of op(B) (transB=0 storage
 *   [K,N]) is taken from row k+row_shift if 0 <= (k % period)+row_shift < period, else it
 *   is zero — the "previous hidden state" view of a [B*T, H] matrix without a copy.
 *   workspace: lr_sgemm_workspace_bytes(M,N,K) bytes enable a deterministic split along K
 *   when M*N alone cannot fill the chip (weight gradients: K = B*T); NULL/0 = no split. */
size_t lr_sgemm_workspace_bytes(int M, int N, int K);
int lr_sgemm(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
             const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
             int row_shift, int period, void* workspace, size_t workspace_bytes,
             lr_stream_t stream);

/* ---- A3: recurrent layer — src/models/lipreader/better_model.py:47-49,64-89 ------------ */

/* One (bi)directional GRU/LSTM layer with torch packed-sequence semantics
 * (pack_padded_sequence -> nn.GRU/LSTM -> pad_packed_sequence, better_model.py:68-78),
 * WITHOUT the length sort: every sample stops at its own length, the reverse direction
 * starts at each sample's last valid frame, positions past a length are zero in y and the
 * final state is the state at each sample's last valid step.
 *
 *   x      [B,T,I]        layer input, batch-first, contiguous
 *   lens   [B] int32      1 <= lens[b] <= T
 *   w_ih   D pointers -> [G*H, I]   (torch weight_ih_l{k}[_reverse])
 *   w_hh   D pointers -> [G*H, H]
 *   b_ih, b_hh  D pointers -> [G*H]
 *   y      [B,T,D*H]      outputs (direction d in columns [d*H,(d+1)*H))
 *   h_n    [D,B,H]        final hidden state; c_n [D,B,H] (LSTM; may be NULL for GRU)
 *   reserve               saved activations for the backward pass (lr_rnn_reserve_bytes)
 * Pointer arrays (w_ih ...) are HOST arrays of D device pointers. */
int lr_rnn_pair_supported(int mode, int B, int T, int I, int H, int D);         /* see LR_RNN_RECUR_SPLIT: 2 or 0 */
/* ... and why not: 0 = the layer has a one-launch recurrence on this device now; 1 = no kernel for the shape; 2 =
 * switched off by lr_rnn_one_launch_enable(0); 3 = switched off by the test hook lr_rnn_debug_disable_cluster; 4 = the
 * device has too few compute units for a launch's clusters.  (What a caller says when it falls back.) */
int lr_rnn_one_launch_status(int mode, int B, int T, int I, int H, int D);
/* Recurrence launches ONE pass (forward or backward) of the layer takes on this device: 1 where every (direction,
 * 8 samples) cluster of the batch fits one launch — up to 8 * floor(32 / members) clusters, members = ceil(H / 32):
 * BiGRU-256 up to B = 128, BiLSTM-512 up to B = 64 (BASELINE configs[3]'s whole batch on one GPU); clusters of 17 .. 24
 * members (one per XCD: BiLSTM-700 / 768) take SIXTEEN samples each once the batch needs more than eight of them:
 * BiLSTM-768 one launch up to B = 64, two at the ecd files' B = 128 —, more where the batch is cut into several
 * launches, T where the layer runs one launch per time step; LSTM past 1152 units: one launch per direction and 64
 * samples. */
int lr_rnn_pass_launches(int mode, int B, int T, int I, int H, int D);
/* LR_RNN_RECUR_SPLIT's workgroups of a pair / cluster must be resident together (lr_rnn_pair_supported checks the
 * device's compute-unit count); their waits are bounded, and a member that gave up leaves garbage and raises the
 * device-side FAULT WORD.  The reference's contract for a batch it cannot use is assert / `None` => skip
 * (src/train/train_better_model.py:46-50); here, with no host round trip:
 *   fault words  int32[2] on the device, {pending, total}.  A recurrence that timed out ORs 1 into `pending`;
 *                lr_ctc_reduce then reports the batch as skipped (loss 0, status 2, zero gradient weights) and
 *                lr_adam_step leaves parameters, moments and step count untouched, exactly as for skip != 0;
 *                lr_step_begin (the top of the next step) moves `pending` into `total`.
 * lr_fault_words_ptr   device pointer to the two words (allocated on the first call — never call it first under
 *                      stream capture), NULL if that failed.
 * lr_rnn_pair_errors   pending + total since the last call, and clears both; synchronises the device (once per
 *                      epoch in train(), tests, bench).
 * lr_step_begin        zeroes the flat gradient buffer grad[0..n) (16-byte aligned; what opt.zero_grad() does,
 *                      train_better_model.py:67) and rolls the fault words; also_zero (may be NULL): TWO more words
 *                      to clear (lr_clip_adam_step's `sumsq`: the accumulator and its ticket); one launch.
 * lr_rnn_debug_drop_member   TEST HOOK: member `m` (>= 0) of every pair / cluster returns at once, so its partners
 *                      time out (a few tenths of a second) and raise the fault; -1 (default) = off.
 * lr_rnn_debug_disable_cluster   TEST HOOK: bit 0 makes lr_rnn_pair_supported answer 0 for the cluster shapes
 *                      (and the decoder loop take its step kernels), bit 3 puts the first pixel-regime layer's dW_ih back on the
 *                      packed lr_xgemm path (round 5's A/B against lr_fgemm), bit 2 keeps the weight gradients of
 *                      LR_RNN_RECUR_SPLIT layers on the fp32 grouped GEMM, so the paths can be compared on one
 *                      model.  Size queries depend on it: set it BEFORE the forward whose backward it should cover.
 * lr_fault_export / lr_fault_import   data parallel (lipreading_amd/distributed.py): out2 = {status[0] (0 when
 *                      status is NULL), -(pending != 0)} for ONE MIN all-reduce over the ranks; import writes
 *                      in2[0] back to status[0] (skip the batch only if every rank skipped it) and raises the local
 *                      fault when in2[1] < 0 (ANY rank timed out: its garbage gradient is in the sum, so every rank
 *                      skips the update and the ranks stay identical). */
/* lr_rnn_one_launch_enable   PRODUCT switch: 0 = every recurrence (encoder layers and the decoder loop) takes the
 *                      per-step kernels from the next forward on, 1 (default) = the one-launch kernels where supported.
 *                      lipreading_amd.train switches it off — once, with a warning — after three consecutive steps
 *                      whose one-launch recurrence timed out (e.g. a device whose compute units are held by another
 *                      process): a slow run instead of one that skips every batch (the reference's contract is skip a
 *                      bad batch and keep training, src/train/train_better_model.py:49-50).  Size queries depend on it:
 *                      switch between steps.
 * lr_debug_busy        TEST HOOK: `workgroups` workgroups that each hold lds_bytes of LDS (up to a whole CU's 160 KB)
 *                      and spin for `microseconds` — a stand-in for a foreign kernel (an RCCL ring kernel on another
 *                      stream) that occupies compute units across the launch of a cluster recurrence. */
void lr_rnn_one_launch_enable(int on);
int lr_rnn_one_launch_enabled(void);
int lr_debug_busy(int workgroups, int lds_bytes, int microseconds, lr_stream_t stream);
int lr_rnn_pair_errors(void);
int lr_fault_export(const int32_t* status, int32_t* out2, lr_stream_t stream);
int lr_fault_import(const int32_t* in2, int32_t* status, lr_stream_t stream);
/* The same two facts as FLOATS that the gradient all-reduce itself can sum (round 5: no second collective): out2 = {1 if
 * status[0] != 0 else 0, 1 if a recurrence of this step timed out else 0}.  lipreading_amd.distributed writes them
 * into the two spare words at the front of the flat gradient buffer; lr_adam_step / lr_clip_adam_step take the summed
 * words as `dist_words` with the number of ranks: the batch is skipped if every rank skipped it (words[0] > world - 0.5),
 * nobody updates if any rank timed out (words[1] > 0.5). */
int lr_fault_export_f32(const int32_t* status, float* out2, lr_stream_t stream);
void* lr_fault_words_ptr(void);
int lr_step_begin(float* grad, int64_t n, float* also_zero, lr_stream_t stream);
/* lr_step_begin and lr_ctc_prepare_i64 (below) in ONE launch: the two first launches of a training step
 * (train_better_model.py:31-32 and :67) depend on nothing and not on each other. */
int lr_step_begin_ctc(float* grad, int64_t n, float* also_zero, const int64_t* chars, int64_t chars_stride,
                      const int64_t* frame_lens, const int64_t* char_lens, int32_t* labels_p1, int32_t* frame_lens32,
                      int32_t* label_lens32, int B, int L, lr_stream_t stream);
void lr_rnn_debug_drop_member(int member);
void lr_rnn_debug_disable_cluster(int off);   /* bit 0: no one-launch recurrence; 2: fp32 weight gradients + projection; 3: packed dW_ih; 4: always 8 samples per cluster */
/* TUNING HOOK of the one-launch recurrences' exchange polling: which = 0 forward / 1 backward cluster kernels; 2 .. 5 the
 * four gathers of the grid recurrence (LSTM past 1152 units): forward h, forward partial sums, backward partial dh,
 * backward dG; first_poll_delay = 64-clock sleeps between a member's publish and its first poll of the others (default 0),
 * round_sleep = sleeps between two poll rounds (default 1). */
void lr_rnn_debug_tune(int which, int first_poll_delay, int round_sleep);
size_t lr_rnn_reserve_bytes(int mode, int B, int T, int I, int H, int D);
size_t lr_rnn_workspace_bytes(int mode, int B, int T, int I, int H, int D);

int lr_rnn_layer_forward(int mode, const float* x, const int32_t* lens,
                         const float* const* w_ih_host, const float* const* w_hh_host,
                         const float* const* b_ih_host, const float* const* b_hh_host, float* y,
                         float* h_n, float* c_n, void* reserve, size_t reserve_bytes, int B,
                         int T, int I, int H, int D, lr_stream_t stream);

/* Backward of the layer above.
 *   dy [B,T,D*H]; dh_n, dc_n [D,B,H] (may be NULL = zero)
 *   dx [B,T,I] (may be NULL when the input needs no gradient: layer 0)
 *   dw_ih/dw_hh/db_ih/db_hh: HOST arrays of D device pointers; overwritten with the grads, or
 *   (accumulate != 0) added to — the reference step runs two backward passes over the same
 *   encoder graph and lets the gradients accumulate (train_better_model.py:69,74).
 *   workspace: lr_rnn_workspace_bytes.
 *   reserve: the forward's, at the SAME address (const for the caller's purposes: the gates are only read).  On the
 *   one-launch recurrence the forward's prologue launch also left the backward's W_hh fragments and cleared exchange
 *   words in it (round 5); the first backward over a forward's reserve uses them (and dirties the exchange words), a
 *   second one over the same forward re-packs with a launch of its own — the library keeps track by the address. */
int lr_rnn_layer_backward(int mode, const float* x, const int32_t* lens,
                          const float* const* w_ih_host, const float* const* w_hh_host,
                          const float* const* b_ih_host, const float* const* b_hh_host,
                          const float* y, const float* dy, const float* dh_n, const float* dc_n,
                          float* dx, float* const* dw_ih_host, float* const* dw_hh_host,
                          float* const* db_ih_host, float* const* db_hh_host,
                          const void* reserve, size_t reserve_bytes, void* workspace,
                          size_t workspace_bytes, int accumulate, int B, int T, int I, int H, int D,
                          lr_stream_t stream);

/* The same backward in two calls that may sit on two streams (build-defined, LR_RNN_PROJ_BF16X3
 * layers only; LR_ERR_UNSUPPORTED otherwise unless parts == 3):
 *   parts & 1: the recurrence and the data gradient dx (what the layer below waits for);
 *   parts & 2: the weight and bias gradients, from the gate gradients the parts & 1 call left in
 *              `workspace` — the caller orders it after that call (an event) and keeps the workspace
 *              alive; it can then overlap the layer below's latency-bound recurrence.
 * parts == 3 is lr_rnn_layer_backward. */
int lr_rnn_layer_backward_parts(int mode, const float* x, const int32_t* lens,
                                const float* const* w_ih_host, const float* const* w_hh_host,
                                const float* const* b_ih_host, const float* const* b_hh_host,
                                const float* y, const float* dy, const float* dh_n, const float* dc_n,
                                float* dx, float* const* dw_ih_host, float* const* dw_hh_host,
                                float* const* db_ih_host, float* const* db_hh_host,
                                const void* reserve, size_t reserve_bytes, void* workspace,
                                size_t workspace_bytes, int accumulate, int B, int T, int I, int H, int D,
                                int parts, lr_stream_t stream);

/* Instrumentation for the roofline leg of bench.py (the only entry points that touch the host
 * clock).  While enabled, every lr_rnn_layer_forward / _backward call issues ONE of its step
 * launches (the middle one) with a hipEvent pair that stamps that dispatch's begin and end on
 * the stream the kernel runs on.
 * The conv entry points do the same for every call.  lr_profile_read(which) WAITS for the recorded
 * events, returns the summed elapsed milliseconds and the number of samples in HOST memory, and
 * clears the ring (1024 samples per slot; later samples are dropped).  Slots: 0/1 recurrent
 * forward/backward step kernel; 2,3,4 conv1..3 forward; 5,6 conv2/conv3 data gradient; 7,8,9
 * conv1..3 weight gradient; 10 CTC alpha/beta kernel (lr_ctc_nll); 11 CTC gradient rows kernel
 * (lr_ctc_grad). */
int lr_profile_enable(int on);
int lr_profile_read(int which, float* total_ms_host, int* samples_host);

/* fp32 GEMM on the bf16 matrix cores by operand splitting (hi = bf16(x), lo = bf16(x - hi);
 * a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulate): same conventions as lr_sgemm; a_exact /
 * b_exact declare an operand whose elements are bf16 values already (no lo term).  Build-defined
 * (no reference counterpart); ~1e-5 relative accuracy. */
size_t lr_xgemm_workspace_bytes(int transA, int transB, int M, int N, int K);
int lr_xgemm(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
             const float* B, int ldb, float beta, float* C, int ldc, const float* bias, int a_exact,
             int b_exact, void* workspace, size_t workspace_bytes, lr_stream_t stream);

/* ---- A3 tail: output_proj + masked_log_softmax — better_model.py:92-93 ------------------ */

/* log_probs[r,:] = log_softmax(hidden[r,:] @ W^T + bias + log(mask + 1e-45))
 *   hidden [R,K] (R = B*T rows, padded rows included exactly as the reference does),
 *   W [C,K], bias [C], mask [C] (0/1 floats; masked classes end ~103.28 below, finite),
 *   log_probs [R,C].  C <= 256.  workspace: lr_proj_workspace_bytes(R,K,C). */
size_t lr_proj_workspace_bytes(int R, int K, int C);
int lr_proj_logsoftmax_forward(const float* hidden, const float* W, const float* bias,
                               const float* mask, float* log_probs, void* workspace,
                               size_t workspace_bytes, int R, int K, int C, lr_stream_t stream);

/* dlogits[r,c] = g[r,c] - exp(log_probs[r,c]) * sum_c g[r,c]; then
 * dhidden = dlogits @ W, dW = dlogits^T @ hidden, dbias = colsum(dlogits).
 *   dlogits [R,C] is caller scratch (also an output).  dhidden may be NULL.
 *   accumulate != 0: dW and dbias are added to instead of overwritten. */
int lr_proj_logsoftmax_backward(const float* g, const float* log_probs, const float* hidden,
                                const float* W, float* dlogits, float* dhidden, float* dW,
                                float* dbias, void* workspace, size_t workspace_bytes, int accumulate,
                                int R, int K, int C, lr_stream_t stream);

/* ---- N1: attention character decoder — better_model.py:124-235, train_better_model.py:54-65 - */

typedef enum lr_attention_type {
  LR_ATT_NONE = 0,    /* output_proj(rnn state)                                   (:228)      */
  LR_ATT_DOT = 1,     /* logit[t] = h . enc[t]                                     (:206-208)  */
  LR_ATT_GENERAL = 2, /* logit[t] = (W_g h + b_g) . enc[t]                         (:191-205)  */
  LR_ATT_1LNN = 3,    /* logit[t] = w . [enc[t]; h] + b                            (:185-190)  */
  LR_ATT_CONCAT = 4   /* logit[t] = w2 . tanh(W1 [enc[t]; h] + b1) + b2            (:209-215)  */
} lr_attention_type;

/* Device pointers to the parameters of CharDecodingStep (torch layouts, fp32). */
typedef struct lr_decoder_params {
  const float* emb;     /* embedding.weight      [V][Cd]                                        */
  const float* w_ih;    /* rnn.weight_ih_l0      [G*Hd][Cd]                                     */
  const float* w_hh;    /* rnn.weight_hh_l0      [G*Hd][Hd]                                     */
  const float* b_ih;    /* rnn.bias_ih_l0        [G*Hd]                                         */
  const float* b_hh;    /* rnn.bias_hh_l0        [G*Hd]                                         */
  const float* attn_w1; /* general: W_g [Hd][Hd]; 1_layer_nn: w [1][2Hd]; concat: W1 [A][2Hd]   */
  const float* attn_b1; /* general: b_g [Hd];     1_layer_nn: b [1];      concat: b1 [A]        */
  const float* attn_w2; /* concat: attn_proj_layer2.weight [1][A]                               */
  const float* attn_b2; /* concat: attn_proj_layer2.bias   [1]                                  */
  const float* w_c;     /* concat_layer.weight   [Hd][2Hd]   (unused for LR_ATT_NONE)           */
  const float* b_c;     /* concat_layer.bias     [Hd]                                           */
  const float* w_o;     /* output_proj.weight    [V][Hd]                                        */
  const float* b_o;     /* output_proj.bias      [V]                                            */
  const float* out_mask;/* [V] 0/1 floats: PAD and BOS masked (better_model.py:142-144)         */
} lr_decoder_params;

/* Gradients, same layouts; overwritten, or added to when `accumulate` != 0. */
typedef struct lr_decoder_grads {
  float* emb; float* w_ih; float* w_hh; float* b_ih; float* b_hh;
  float* attn_w1; float* attn_b1; float* attn_w2; float* attn_b2;
  float* w_c; float* b_c; float* w_o; float* b_o;
  int emb_padding_idx;  /* nn.Embedding(padding_idx): that row receives no gradient; -1 = none    */
} lr_decoder_grads;

/* Layers 1 .. num_layers-1 of the decoder's RNN stack (better_model.py:136,147-148: the decoder takes
 * the encoder's num_layers; nn.GRU/LSTM/RNN(char_dim, Hd, num_layers) semantics: layer k's input is
 * layer k-1's output of the same step).  Entry k-1 holds layer k: rnn.weight_ih_l{k} [G*Hd][Hd],
 * rnn.weight_hh_l{k} [G*Hd][Hd], rnn.bias_ih_l{k}, rnn.bias_hh_l{k} [G*Hd].  drop_mask (NULL = none):
 * [num_layers-1][B][L][Hd] multipliers (0 or 1/(1-p)) applied to the outputs of every layer but the
 * last — the inter-layer dropout of nn.GRU(dropout=p) in training mode, drawn by the caller.
 * Passing NULL for the struct = a single-layer decoder (every shipped config). */
#define LR_DEC_MAX_LAYERS 8
typedef struct lr_decoder_upper {
  int num_layers;
  const float* w_ih[LR_DEC_MAX_LAYERS - 1];
  const float* w_hh[LR_DEC_MAX_LAYERS - 1];
  const float* b_ih[LR_DEC_MAX_LAYERS - 1];
  const float* b_hh[LR_DEC_MAX_LAYERS - 1];
  const float* drop_mask;
} lr_decoder_upper;
typedef struct lr_decoder_upper_grads {
  float* w_ih[LR_DEC_MAX_LAYERS - 1];
  float* w_hh[LR_DEC_MAX_LAYERS - 1];
  float* b_ih[LR_DEC_MAX_LAYERS - 1];
  float* b_hh[LR_DEC_MAX_LAYERS - 1];
} lr_decoder_upper_grads;

/* The whole decoder loop of one batch (max_label_len = L steps): an RNN stack (mode = LR_RNN_GRU /
 * LR_RNN_LSTM / LR_RNN_TANH, num_layers layers) of hidden size Hd started from the encoder's final
 * state (h0 [num_layers][B][Hd], c0 likewise for the LSTM); the attention and the output head read the
 * top layer's state.
 *   tokens [B][L] int32    teacher inputs chars[:, i] (device)
 *   teacher_forced_host[L] HOST bytes: 1 = step i is fed tokens[:, i], 0 = fed the previous step's
 *                          multinomial sample (train_better_model.py:57-58; step 0 always uses tokens)
 *   enc [B][T][Hd], enc_lens [B] int32; step_lens [B] int32, every entry = L
 *   log_probs [B][L][V]    masked log-softmax outputs of every step (better_model.py:229)
 *   sampled [B][L] int32   multinomial(exp(log_probs)) of every step (counter-based RNG on `seed`;
 *                          equal to torch's sampler in distribution only)
 * The caller derives the loss (nll_loss, ignore_index = PAD, train_better_model.py:62) and the
 * accuracy counts (:131-133) from log_probs / sampled.  h_n / c_n [num_layers][B][Hd] (may be NULL)
 * receive the RNN state after the last step (the final_state a single reference step returns).   */
size_t lr_decoder_reserve_bytes(int mode, int attn_type, int num_layers, int B, int L, int T, int Hd, int Cd,
                                int V, int A);
size_t lr_decoder_workspace_bytes(int mode, int attn_type, int num_layers, int B, int L, int T, int Hd, int Cd,
                                  int V, int A);
int lr_decoder_forward(int mode, int attn_type, const lr_decoder_params* params_host,
                       const lr_decoder_upper* upper_host, const int32_t* tokens,
                       const uint8_t* teacher_forced_host, const float* enc, const int32_t* enc_lens,
                       const float* h0, const float* c0, const int32_t* step_lens, uint64_t seed,
                       float* log_probs, int32_t* sampled, float* h_n, float* c_n, void* reserve,
                       size_t reserve_bytes, int B, int L, int T, int Hd, int Cd, int V, int A,
                       lr_stream_t stream);
/* Backward of the loop: d_log_probs [B][L][V] (+ dh_n / dc_n [num_layers][B][Hd], gradient arriving
 * through the returned final state; may be NULL) -> d_enc [B][T][Hd] (overwritten), dh0 / dc0
 * [num_layers][B][Hd] (gradient into the encoder's final state) and every parameter gradient
 * (upper_grads_host may be NULL for a single layer).                                                 */
int lr_decoder_backward(int mode, int attn_type, const lr_decoder_params* params_host,
                        const lr_decoder_upper* upper_host, const lr_decoder_grads* grads_host,
                        const lr_decoder_upper_grads* upper_grads_host, const float* enc, const int32_t* enc_lens,
                        const float* h0, const float* c0, const int32_t* step_lens, const float* log_probs,
                        const float* d_log_probs, const float* dh_n, const float* dc_n, float* d_enc,
                        float* dh0, float* dc0, const void* reserve, size_t reserve_bytes, void* workspace,
                        size_t workspace_bytes, int accumulate, int B, int L, int T, int Hd, int Cd, int V,
                        int A, lr_stream_t stream);
/* ... in two halves (as lr_rnn_layer_backward_parts): parts 1 = the data half — d_enc, dh0, dc0: what the encoder's
 * backward waits for —, 2 = every parameter gradient from what part 1 left in `workspace` (same arguments, any stream
 * that waits for part 1; the workspace must live until it has run), 3 = both.  Separable for a single-layer loop
 * (lr_decoder_backward_splittable: every flag file the reference ships); otherwise only parts = 3 is accepted. */
int lr_decoder_backward_splittable(int attn_type, int num_layers);
int lr_decoder_backward_parts(int mode, int attn_type, const lr_decoder_params* params_host,
                              const lr_decoder_upper* upper_host, const lr_decoder_grads* grads_host,
                              const lr_decoder_upper_grads* upper_grads_host, const float* enc, const int32_t* enc_lens,
                              const float* h0, const float* c0, const int32_t* step_lens, const float* log_probs,
                              const float* d_log_probs, const float* dh_n, const float* dc_n, float* d_enc,
                              float* dh0, float* dc0, const void* reserve, size_t reserve_bytes, void* workspace,
                              size_t workspace_bytes, int accumulate, int B, int L, int T, int Hd, int Cd, int V,
                              int A, int parts, lr_stream_t stream);

/* The decoder loss of the train loop (train_better_model.py:62,65): over the R = B*L (sample, step) rows,
 * -sum_r log_probs[r][label_r] for label_r != ignore_index (F.nll_loss(ignore_index=PAD, reduction='sum') summed
 * over the steps) divided by the number of such rows ((labels != PAD).sum()).  labels: int64, row (b, i) at
 * labels[b * label_stride + i] (a view of chars[:, 1:]).  out2[0] = loss, out2[1] = count (kept for the backward).
 * Backward: d_log_probs [R][V] (every element written) from the upstream scalar gradient on the device. */
int lr_nll_mean_forward(const float* log_probs, const int64_t* labels, int64_t label_stride, int L, int ignore_index,
                        float* out2, int R, int V, lr_stream_t stream);
/* ... the same with out3 = {loss, count, the un-divided sum}: eval keeps sum and count apart
 * (train_better_model.py:127,138: nll sums added over the batches, divided by the total count at the end). */
int lr_nll_forward3(const float* log_probs, const int64_t* labels, int64_t label_stride, int L, int ignore_index,
                    float* out3, int R, int V, lr_stream_t stream);
int lr_nll_mean_backward(const int64_t* labels, int64_t label_stride, int L, int ignore_index, const float* fwd_out2,
                         const float* grad_out, float* d_log_probs, int R, int V, lr_stream_t stream);

/* ---- A10 (build-defined): transformer encoder blocks — no reference symbol ------ *
 * torch.nn.TransformerEncoderLayer(norm_first=False, activation=relu, dropout=0) arithmetic; the
 * host composition is lipreading_amd/transformer.py -> lr_tfm_* below; the pieces are entry points of their own. */

/* batch_outer x batch_inner independent fp32 products C_z = alpha op(A_z) op(B_z) + beta C_z, problem
 * z = (o, i) at element offsets o*s?_outer + i*s?_inner from the base pointers (0: shared operand).
 * Per (sample, head) attention products read Q/K/V straight out of the fused [R][3D] projection. */
int lr_sgemm_batched(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                     int64_t sA_outer, int64_t sA_inner, const float* B, int ldb, int64_t sB_outer,
                     int64_t sB_inner, float beta, float* C, int ldc, int64_t sC_outer, int64_t sC_inner,
                     int batch_outer, int batch_inner, lr_stream_t stream);
/* y = LayerNorm(x + residual) * gamma + beta over rows of D (residual may be NULL); stats [R][2] =
 * (mean, rstd) for the backward.  Backward: dx is the gradient of BOTH x and residual; dgamma / dbeta
 * overwritten or (accumulate != 0) added to. */
int lr_layernorm_forward(const float* x, const float* residual, const float* gamma, const float* beta,
                         float* y, float* stats, int R, int D, float eps, lr_stream_t stream);
size_t lr_layernorm_workspace_bytes(int D);
int lr_layernorm_backward(const float* x, const float* residual, const float* gamma, const float* stats,
                          const float* dy, float* dx, float* dgamma, float* dbeta, void* workspace,
                          size_t workspace_bytes, int accumulate, int R, int D, lr_stream_t stream);
/* In place: scores [B][Hh][T][T] -> softmax over keys of scale*scores with keys >= key_lens[b] masked
 * (probability 0).  Backward, in place over dprobs: dscores = scale * P * (dP - sum_k dP P). */
int lr_attn_softmax_forward(float* scores, const int32_t* key_lens, float scale, int B, int Hh, int T,
                            lr_stream_t stream);
int lr_attn_softmax_backward(const float* probs, float* dprobs, float scale, int B, int Hh, int T,
                             lr_stream_t stream);
/* Fused self-attention core on the bf16 matrix cores (lr_attention.hip): per (sample, head)
 * out = softmax_keys(scale * Q K^T, keys >= key_lens[b] masked) V, with Q / K / V read in place from the fused
 * projection qkv [B][T][3*nhead*dh] (head h at columns h*dh of each third), out [B][T][nhead*dh]; backward:
 * dout -> dqkv (same layout as qkv; every element written).  One workgroup per (sample, head), T <= 96, dh in
 * {32, 64} (lr_attn_fused_supported); the T x T probabilities never leave the chip (the backward recomputes
 * them).  bf16 operands, fp32 accumulation and softmax. */
int lr_attn_fused_supported(int T, int dh);
int lr_attn_fused_forward(const float* qkv, const int32_t* key_lens, float* out, float scale, int B, int T,
                          int nhead, int dh, lr_stream_t stream);
int lr_attn_fused_backward(const float* qkv, const int32_t* key_lens, const float* dout, float* dqkv, float scale,
                           int B, int T, int nhead, int dh, lr_stream_t stream);
/* Products straight from the tensors as they lie in memory (lr_fgemm.hip; BUILD-DEFINED, no reference symbol):
 *   C[M][N] = epilogue(sum_k op(A)[m][k] op(B)[k][n]),  form NT: A [M][K], B [N][K];  NN: A [M][K], B [K][N];
 *   TN: A [K][M], B [K][N].  prec LR_FGEMM_X3: fp32 operands split into bf16 hi + lo on the way into LDS, three
 *   products on the bf16 matrix cores (~1e-5 relative); LR_FGEMM_F32: exact fp32 matrix-core products.  a_bf16 (NT
 *   only) / b_bf16 (TN only): that operand is stored as bf16.  Epilogue per job: out = alpha * acc + bias[n] +
 *   addend[(m % add_period) * ldadd + n]; LR_FGEMM_RELU; mask (out = mask[m * ldmask + n] > 0 ? out : 0); + beta * C;
 *   LR_FGEMM_C_BF16: C is a bf16 matrix (beta must be 0).  colsum (TN only): colsum[m] = beta * colsum[m] + sum_k A[k][m] — the
 *   bias gradient of a weight-gradient product.  splits > 1: K is cut into that many ranges whose partial sums (and
 *   partial column sums) go through `slabs` (lr_fgemm_slab_floats(M, N, splits) floats) and one combine launch for all
 *   the launch's split jobs (lr_fgemm_splits suggests a count).  Up to 20 jobs of one form share a launch.  Only enqueues. */
#define LR_FGEMM_X3 0
#define LR_FGEMM_F32 1
#define LR_FGEMM_NT 0
#define LR_FGEMM_NN 1
#define LR_FGEMM_TN 2
#define LR_FGEMM_RELU 1
#define LR_FGEMM_C_BF16 2
typedef struct {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  const float* addend;
  const float* mask;
  float* colsum;
  float* slabs;
  int32_t M, N, K, lda, ldb, ldc, ldadd, add_period, ldmask, flags, splits;
  float alpha, beta;
  int32_t b_shift, b_period;   /* NN / TN, fp32 B: row k of B is read from row k + b_shift, as zeros where (k % b_period) +
                                  b_shift leaves [0, b_period) — the previous / next time step of a [B*T][N] sequence tensor
                                  (a recurrent layer's dW_hh = dG^T . h_prev straight from y); b_period 0: off */
} lr_fgemm_job;
int lr_fgemm(int prec, int form, int a_bf16, int b_bf16, const lr_fgemm_job* jobs, int njobs, lr_stream_t stream);
int lr_fgemm_splits(int M, int N, int K);
long long lr_fgemm_slab_floats(int M, int N, int splits);

/* The whole encoder stack of lipreading_amd/transformer.py as three enqueues (lr_transformer.hip):
 *   h0 = x W_p^T + b_p + pe[t];  per layer (post-LN, ReLU feed-forward, dropout 0 — torch.nn.TransformerEncoderLayer):
 *   qkv = h W_qkv^T + b;  a = attention(qkv, key_lens);  h1 = LN1(a W_o^T + b_o + h);
 *   h2 = LN2(relu(h1 W_1^T + b_1) W_2^T + b_2 + h1).
 * mode: LR_TFM_X3 (projections as split-bf16 products, else exact fp32) | LR_TFM_X_BF16 (x is stored as bf16) |
 *   LR_TFM_DX_BF16 (dx is written as bf16) | LR_TFM_ATTN_FUSED (lr_attn_fused_*; else fp32 batched products + softmax) |
 *   LR_TFM_ROWBLOCK (below).
 * weights: 2 + 12 * nlayers device pointers in torch's registration order: input_proj.weight [Dm][I], .bias, then per
 *   layer in_proj_weight [3Dm][Dm], in_proj_bias, out_proj.weight [Dm][Dm], .bias, linear1.weight [F][Dm], .bias,
 *   linear2.weight [Dm][F], .bias, norm1.weight, .bias, norm2.weight, .bias.  pe: [>= T][Dm].
 * forward: x [B*T][I] -> h_out [B*T][Dm]; `reserve` (lr_tfm_reserve_bytes) keeps what the backward reads.
 * backward_data: dh_out -> dx (NULL: not wanted) and every layer's pre-activation gradients, kept in `workspace`
 *   (lr_tfm_workspace_bytes); backward_weights (any stream that waits for backward_data) turns them into the 2 + 12 *
 *   nlayers gradients: every weight gradient of the stack with its bias gradient in ONE launch (two when x is bf16),
 *   the LayerNorm parameter gradients in one more; accumulate != 0 adds to `grads`. */
#define LR_TFM_X3 1
#define LR_TFM_X_BF16 2
#define LR_TFM_DX_BF16 4
#define LR_TFM_ATTN_FUSED 8
#define LR_TFM_ROWBLOCK 16   /* out-projection .. LN2 (and their backward) as ONE launch per layer and direction over 32-row
                                blocks (lr_tfm_rowblock.hip); needs LR_TFM_X3 and lr_tfm_rowblock_supported(): d_model 256,
                                feed-forward width 256, 512, 1024 or 2048, at most 8 layers */
int lr_tfm_rowblock_supported(int B, int T, int Dm, int F, int nlayers);
size_t lr_tfm_reserve_bytes(int mode, int B, int T, int I, int Dm, int nhead, int F, int nlayers);
size_t lr_tfm_workspace_bytes(int mode, int B, int T, int I, int Dm, int nhead, int F, int nlayers);
int lr_tfm_forward(int mode, const void* x, const int32_t* key_lens, const float* const* weights, const float* pe,
                   float* h_out, void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes, int B, int T,
                   int I, int Dm, int nhead, int F, int nlayers, float eps, lr_stream_t stream);
int lr_tfm_backward_data(int mode, const int32_t* key_lens, const float* const* weights, const float* dh_out, void* dx,
                         void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes, int B, int T, int I,
                         int Dm, int nhead, int F, int nlayers, lr_stream_t stream);
int lr_tfm_backward_weights(int mode, const void* x, float* const* grads, int accumulate, void* reserve,
                            size_t reserve_bytes, void* workspace, size_t workspace_bytes, int B, int T, int I, int Dm,
                            int nhead, int F, int nlayers, lr_stream_t stream);

/* ---- A4: CTC loss — src/train/ctc_loss.py:28-114 ---------------------------------------- */

/* Per-sample CTC negative log-likelihood and its gradient, blank = 0, the same recursion as
 * torch.nn.functional.ctc_loss which the reference calls at ctc_loss.py:85.
 *   log_probs  element (b,t,c) at log_probs[b*stride_b + t*stride_t + c]; so both the
 *              reference's (B,T,V') tensor and its transposed (T,B,V') view are accepted
 *   labels     [B, label_stride] int32, ALREADY shifted by +1 (ctc_loss.py:80), PAD after
 *              label_lens[b]
 *   frame_lens [B] int32 (input lengths), label_lens [B] int32
 *   nll        [B]  out: -log p(labels | log_probs); +inf when no valid alignment
 *   grad       out, same addressing as log_probs, or NULL for loss only.  grad[b,t,c] =
 *              grad_weight[b] * (exp(lp) - exp(log sum_{s:l'_s=c} alpha_t(s) beta_t(s) + nll - lp))
 *              for t < frame_lens[b], 0 after — torch's ctc_loss backward formula.
 *              grad_weight may be NULL (= 1).  Samples whose nll is inf or whose
 *              grad_weight is 0 get an all-zero gradient.
 *   workspace  lr_ctc_workspace_bytes(B, T, C, max_label_len) bytes; lr_ctc_grad must get the
 *              workspace lr_ctc_nll filled (alpha/beta tables and label chains), unmodified.
 * T is the padded time extent (number of t rows addressed), C = V'+... number of classes,
 * max_label_len <= 256 bounds label_lens (ctc_loss.py:46). */
size_t lr_ctc_workspace_bytes(int B, int T, int C, int max_label_len);

int lr_ctc_nll(const float* log_probs, int64_t stride_b, int64_t stride_t, const int32_t* labels,
               int label_stride, const int32_t* frame_lens, const int32_t* label_lens, float* nll,
               void* workspace, size_t workspace_bytes, int B, int T, int C, int max_label_len,
               lr_stream_t stream);

int lr_ctc_grad(const float* log_probs, int64_t stride_b, int64_t stride_t, const int32_t* labels,
                int label_stride, const int32_t* frame_lens, const int32_t* label_lens,
                const float* nll, const float* grad_weight, float* grad, void* workspace,
                size_t workspace_bytes, int B, int T, int C, int max_label_len,
                lr_stream_t stream);

/* lr_ctc_grad with every grad_weight[b] multiplied by the DEVICE scalar grad_scale[0] (NULL = 1): the incoming
 * gradient of the reduced loss in an autograd backward (_CTCLossFunction.backward), without an elementwise launch
 * to fold it into the weights first. */
int lr_ctc_grad_scaled(const float* log_probs, int64_t stride_b, int64_t stride_t, const int32_t* labels,
                       int label_stride, const int32_t* frame_lens, const int32_t* label_lens,
                       const float* nll, const float* grad_weight, const float* grad_scale, float* grad,
                       void* workspace, size_t workspace_bytes, int B, int T, int C, int max_label_len,
                       lr_stream_t stream);

/* The reference's batch reduction (ctc_loss.py:46-114) evaluated on the device from the
 * per-sample nll, so no host round trip is needed to apply it:
 *   - samples with label_lens > 256 are dropped (:46);
 *   - the (ascending) batch is cut into runs of equal frame_len (:64-65);
 *   - a run containing an inf nll drops those samples (:87-101); a run with none left is
 *     skipped WITHOUT advancing the run start (:92 `continue` before :107), so its samples
 *     are re-examined as part of the next run;
 *   - 'mean': each run's torch-'mean' (mean of nll/clamp(label_len,1)) is multiplied by
 *     `minibatch_size`, which the reference reads BEFORE slicing (:74) — the whole batch for
 *     the first run, the previous run's size afterwards — and `count` accumulates the same
 *     number (:103-105); 'sum': plain sum.
 *   out_loss[0]   the loss (total/count or total); 0 when status != 0
 *   out_status[0] 0 = ok, 1 = reference would return None (everything dropped / total == 0)
 *   grad_weight   [B] d out_loss / d nll[b]  (0 for dropped samples) -> lr_ctc_grad */
int lr_ctc_reduce(const float* nll, const int32_t* frame_lens, const int32_t* label_lens,
                  int reduction, float* out_loss, int32_t* out_status, float* grad_weight, int B,
                  lr_stream_t stream);

/* lr_ctc_nll followed by lr_ctc_reduce with the same arguments, as ONE launch when every label has at most 31
 * characters (the one-wave recursion kernel: its last workgroup to finish runs the batch reduction — round 5; the
 * train step's path, lipreading_amd/ctc.py), as the two launches otherwise.  Launches of this entry must not overlap
 * in time within a process (one completion counter; launches on one stream never do). */
int lr_ctc_nll_reduce(const float* log_probs, int64_t stride_b, int64_t stride_t, const int32_t* labels,
                      int label_stride, const int32_t* frame_lens, const int32_t* label_lens, float* nll,
                      void* workspace, size_t workspace_bytes, int reduction, float* out_loss, int32_t* out_status,
                      float* grad_weight, int B, int T, int C, int max_label_len, lr_stream_t stream);

/* The train loop's label plumbing in ONE launch (train_better_model.py:31-32 labels = chars[:, 1:], label_lens =
 * char_lens - 1; ctc_loss.py:42,80 int32 integers, labels moved up by one so that index 0 is the blank):
 *   labels_p1[b][l]  = (int32) chars[b * chars_stride + 1 + l] + 1     l < L (= chars' width - 1)
 *   frame_lens32[b]  = (int32) frame_lens[b];   label_lens32[b] = (int32) char_lens[b] - 1
 * (as torch ops these are five elementwise launches at the head of every step). */
int lr_ctc_prepare_i64(const int64_t* chars, int64_t chars_stride, const int64_t* frame_lens, const int64_t* char_lens,
                       int32_t* labels_p1, int32_t* frame_lens32, int32_t* label_lens32, int B, int L,
                       lr_stream_t stream);

/* ---- A6: CTC greedy decode — src/models/lipreader/decoder.py:165-197 -------------------- */

/* argmax over classes (first maximum, as torch.max), then per sample for t < sizes[b]:
 * drop blanks, drop a frame whose argmax equals the previous FRAME's argmax (:168-173).
 *   probs    element (b,t,c) at probs[b*stride_b + t*stride_t + c]
 *   sizes    [B] int32 or NULL (= T)
 *   class_map [C] int32 or NULL: the reference compares CHARACTERS (labels[i] == labels[j]),
 *            so with duplicate label strings the caller passes the canonical index per class
 *   out_ids  [B,T] int32 kept (mapped) class ids, out_offsets [B,T] int32 frame index of each kept id,
 *   out_lens [B] int32 number kept.  Entries past out_lens[b] are -1. */
int lr_ctc_greedy_decode(const float* probs, int64_t stride_b, int64_t stride_t,
                         const int32_t* sizes, const int32_t* class_map, int32_t* out_ids,
                         int32_t* out_offsets, int32_t* out_lens, int B, int T, int C, int blank,
                         lr_stream_t stream);

/* ---- A8 (BUILD-DEFINED, no reference symbol): 3-D conv frontend on bf16 MFMA -------------- */
/* The reference has no conv frontend (src/models/lipreader/model.py:122,153-156 are comments, the
 * `ced` configs are empty); BASELINE.json's north_star asks for one ("im2col + MFMA GEMM for the 3D
 * convs").  The specification is this build's; the oracle is torch conv3d on the CPU.
 * Activations are channels-last bf16 [B][T][H][W][C]; accumulation is fp32.                      */

/* clips [frames][3][H][W] (uint8, scaled by 1/255, or fp32 taken as is) -> [frames][H][W][4] bf16. */
int lr_clip_to_ndhwc_bf16(const void* clips, int is_u8, void* out, int64_t frames, int H, int W,
                          lr_stream_t stream);

/* fp32 torch Conv3d weight [Cout][Cin_real][KT][KH][KW] -> bf16 operand of the implicit GEMM:
 *   dgrad & 1 == 0: out[Cout][taps][Cin_pad]          (forward; channels >= Cin_real are zero)
 *   dgrad & 1 == 1: out[Cin_real][taps][Cout], taps flipped (data gradient of a stride-1 "same" conv
 *                   becomes lr_conv3d_forward on dZ with this operand).
 *   dgrad & 2 / dgrad & 4: the same elements in the fragment-major order of a patch-resident kernel
 *              (every 32-output-channel x 16-k, resp. 16-output-channel x 32-k, MFMA operand 1 KB
 *              contiguous); the bit is what lr_conv3d_patch_supported() returns for the layer and
 *              is passed on to lr_conv3d_forward in `flags`. */
int lr_conv3d_pack_weights(const float* W, void* out, int Cout, int Cin_real, int Cin_pad, int KT,
                           int KH, int KW, int dgrad, lr_stream_t stream);
/* n <= 8 operands in ONE launch (HOST arrays of n entries each, meaning as above): a training step packs the
 * forward operand of every layer and the data-gradient operand of the upper layers together. */
int lr_conv3d_pack_weights_multi(int n, const float* const* W, void* const* out, const int* Cout, const int* Cin_real,
                                 const int* Cin_pad, const int* KT, const int* KH, const int* KW, const int* dgrad,
                                 lr_stream_t stream);

/* Y[b,t,ho,wo,n] = act( sum_{kt,kh,kw,c} X[b,t+kt-pt,ho*s+kh-ph,wo*s+kw-pw,c] * Wp[n][(kt,kh,kw)][c]
 *                       + bias[n] ),  zero padding, temporal stride 1 and KT = 2*pt+1, spatial stride s.
 *   X bf16 [B][T][Hin][Win][Cin] (Cin % 4 == 0), Wp from lr_conv3d_pack_weights, bias fp32 or NULL,
 *   Y bf16 [B][T][Ho][Wo][Cout] (Cout in {32,64,96}); flags & 1 applies max(.,0) (ReLU);
 *   flags & 2 / flags & 4: Wp is fragment-major (lr_conv3d_pack_weights dgrad & 2 / & 4);
 *   flags & 8 (first STCNN layer only, LR_ERR_UNSUPPORTED elsewhere): X is the raw clip, uint8
 *   [B][T][3][Hin][Win], scaled by 1/255 on the way into LDS exactly as lr_clip_to_ndhwc_bf16
 *   would have — no bf16 copy of the clip is made.
 *   The first STCNN layer's geometry (Cin = 4, Cout = 32, taps 3x5x5, stride 2, padding 1,2,2) is a 3-channel
 *   layer padded to four: its kernel contracts channels 0..2 only — X[...,3] and Wp[...][3] are padding (zero, as
 *   lr_clip_to_ndhwc_bf16 and lr_conv3d_pack_weights with Cin_real = 3 write them) and are not read.  */
int lr_conv3d_forward(const void* X, const void* Wp, const float* bias, void* Y, int B, int T, int Hin,
                      int Win, int Cin, int Cout, int KT, int KH, int KW, int stride, int pt, int ph,
                      int pw, int flags, lr_stream_t stream);
/* Forward with the ReLU -> MaxPool3d((1,2,2)) that follows it fused into the epilogue, for layers
 * with lr_conv3d_pool_fusion_supported() == 1: P bf16 [B][T][Ho/2][Wo/2][Cout] receives the pooled
 * activation and code (uint8, same shape) each window's CODE: the position 0..3 (row-major in the 2x2
 * window) of its first maximum, or 4 when the pooled activation is 0 (ReLU blocks the window's gradient);
 * the full-resolution activation is never written.  Backward of the pair: lr_unpool_code_bf16 (materialises
 * dZ), or — never writing dZ — lr_conv3d_dgrad_pooled / lr_conv3d_wgrad_pooled. */
int lr_conv3d_pool_fusion_supported(int Hin, int Win, int Cin, int Cout, int KT, int KH, int KW, int stride,
                                    int pt, int ph, int pw);
int lr_conv3d_forward_pooled(const void* X, const void* Wp, const float* bias, void* P, void* code, int B,
                             int T, int Hin, int Win, int Cin, int Cout, int KT, int KH, int KW, int stride,
                             int pt, int ph, int pw, int flags, lr_stream_t stream);
/* Data gradient of a stride-1 "same" layer whose forward fused ReLU + MaxPool, taken straight from the pooled
 * gradient dP bf16 [B][T][Ho/2][Wo/2][Cout] and the window codes (uint8, same shape): dX bf16 [B][T][Ho][Wo][Cin]
 * = conv(un-pool(dP, code), flipped / channel-transposed weights).  The kernel rebuilds its dZ patch on the way
 * into LDS; dZ itself (Ho x Wo, 75 % zeros) is never written.  Wd: lr_conv3d_pack_weights with the dgrad flag and
 * the fragment bit `_supported` returns (0: not supported — use lr_unpool_code_bf16 + lr_conv3d_forward).
 * Build-defined like the rest of the frontend (no reference symbol). */
int lr_conv3d_dgrad_pooled_supported(int Ho, int Wo, int Cout, int Cin, int KT, int KH, int KW, int pt, int ph, int pw);
int lr_conv3d_dgrad_pooled(const void* dP, const void* code, const void* Wd, void* dX, int B, int T, int Ho, int Wo,
                           int Cout, int Cin, int KT, int KH, int KW, int pt, int ph, int pw, lr_stream_t stream);
/* Non-zero when the layer (as lr_conv3d_forward sees it: Cin = contraction channels, Cout = output
 * channels) has a patch-resident kernel — the input patch of an output tile is loaded into LDS
 * once and all taps run out of LDS, instead of re-gathering it per tap.  The value is the weight
 * packing bit that kernel reads: 2 (24x24, taps 3x5x5) or 4 (12x12, taps 3x3x3). */
int lr_conv3d_patch_supported(int Hin, int Win, int Cin, int Cout, int KT, int KH, int KW, int stride,
                              int pt, int ph, int pw);

/* dW[Cout][Cin_real][KT][KH][KW] (fp32) (+)= sum_pixels dZ[pixel][n] * im2col(X)[pixel][(tap,c)];
 * dbias[n] (+)= sum_pixels dZ[pixel][n] (NULL to skip).  dZ bf16 [B][T][Ho][Wo][Cout].          */
size_t lr_conv3d_wgrad_workspace_bytes(int Cout, int Cin_pad, int KT, int KH, int KW);
int lr_conv3d_wgrad(const void* X, const void* dZ, float* dW, float* dbias, void* workspace,
                    size_t workspace_bytes, int accumulate, int B, int T, int Hin, int Win, int Cin_pad,
                    int Cin_real, int Cout, int KT, int KH, int KW, int stride, int pt, int ph, int pw,
                    lr_stream_t stream);

/* The same weight (and bias) gradient for a layer whose forward fused ReLU + MaxPool
 * (lr_conv3d_forward_pooled), taken straight from the pooled gradient dP and the window codes: the
 * un-pooled dZ is rebuilt tile by tile inside the kernel and never touches memory.  Build-defined like the rest of the frontend (the reference has no conv
 * stage; its only trace is the commented-out stack at src/models/lipreader/model.py:122,153-156).
 * `_supported` is non-zero for the layers that have this kernel: 1 = the first STCNN layer (flags bit 0: X is the raw
 * clip, uint8 [B][T][3][Hin][Win], scaled by 1/255 on the way into LDS as lr_clip_to_ndhwc_bf16 would, not bf16 NDHWC);
 * 2 = the stride-1 layers 2 and 3 (LR_ERR_UNSUPPORTED for a frame count whose tile table does not fit the kernel's LDS:
 * use lr_unpool_code_bf16 + lr_conv3d_wgrad then).  The codes carry the ReLU mask — code 4 — so `pooled` is not read
 * and may be NULL; dbias = column sums of dP over the windows ReLU did not block.  Workspace as for lr_conv3d_wgrad. */
int lr_conv3d_wgrad_pooled_supported(int Hin, int Win, int Cin_pad, int Cin_real, int Cout, int KT, int KH,
                                     int KW, int stride, int pt, int ph, int pw);
/* The same answer for a given frame count B * T: 0 where lr_conv3d_wgrad_pooled would return LR_ERR_UNSUPPORTED for it
 * (the stride-1 layers' kernel keeps a tile table in LDS that has to fit). */
int lr_conv3d_wgrad_pooled_supported_frames(int frames, int Hin, int Win, int Cin_pad, int Cin_real, int Cout, int KT,
                                            int KH, int KW, int stride, int pt, int ph, int pw);
int lr_conv3d_wgrad_pooled(const void* X, const void* pooled, const void* code, const void* dP, float* dW,
                           float* dbias, void* workspace, size_t workspace_bytes, int accumulate, int B, int T,
                           int Hin, int Win, int Cin_pad, int Cin_real, int Cout, int KT, int KH, int KW,
                           int stride, int pt, int ph, int pw, int flags, lr_stream_t stream);

/* MaxPool3d((1,2,2)) on channels-last bf16, and the backward of ReLU -> that pool: the gradient of
 * a window goes to its FIRST maximum (row-major, torch's rule) if the activation there is > 0.   */
int lr_maxpool_hw2_bf16(const void* in, void* out, int64_t frames, int H, int W, int C,
                        lr_stream_t stream);
/* dbias (may be NULL) receives the layer's bias gradient sum_pos dZ[pos][n] (fp32 [C]; added to
 * when accumulate != 0), computed in the same pass; it needs lr_unpool_workspace_bytes(C) bytes. */
size_t lr_unpool_workspace_bytes(int C);
int lr_unpool_relu_mask_bf16(const void* act, const void* dP, void* dZ, float* dbias, int accumulate,
                             void* workspace, size_t workspace_bytes, int64_t frames, int H, int W, int C,
                             lr_stream_t stream);
/* the same from the pooled activation and the window codes of lr_conv3d_forward_pooled (H, W: the
 * UN-pooled extent of dZ) */
int lr_unpool_code_bf16(const void* pooled, const void* code, const void* dP, void* dZ, float* dbias,
                        int accumulate, void* workspace, size_t workspace_bytes, int64_t frames, int H, int W,
                        int C, lr_stream_t stream);
int lr_bf16_to_f32(const void* in, float* out, int64_t n, lr_stream_t stream);
int lr_f32_to_bf16(const float* in, void* out, int64_t n, lr_stream_t stream);

/* Inter-layer dropout of nn.GRU / nn.LSTM(dropout = p) in training mode (better_model.py:47-49 hands rnn_dropout to
 * torch; every shipped config sets 0): mask[i] = 0 with probability p, else 1 / (1 - p) (Philox4x32-10, key = seed,
 * counter = i / 4: the same mask for a (seed, n) whatever the launch geometry), y = x * mask (x == y == NULL: only
 * the mask is written — the decoder loop applies it inside its own kernels).  Backward: dx = dy * mask
 * (lr_mul_f32).  lr_cat_directions: (D, B, H) -> (B, D*H), forward direction first (better_model.py:98-112
 * _cat_directions), for one or two tensors (h, c) in one launch; inverse != 0: the way back (its gradient). */
int lr_dropout_forward(const float* x, float* y, float* mask, int64_t n, float p, uint64_t seed, lr_stream_t stream);
int lr_mul_f32(const float* a, const float* b, float* out, int64_t n, lr_stream_t stream);
int lr_cat_directions(const float* in0, float* out0, const float* in1, float* out1, int B, int H, int D, int inverse,
                      lr_stream_t stream);

/* ---- A5 tail: optimiser side of the reference step — train_better_model.py:78-80 -------- */

/* sum of squares of n floats accumulated into out[0] (caller zeroes it); used for
 * clip_grad_norm_ (train_better_model.py:78) over a flat gradient buffer. */
int lr_sumsq(const float* x, int64_t n, float* out, lr_stream_t stream);

/* Fused clip + Adam step over a flat parameter buffer (torch.optim.Adam defaults as used at
 * src/scripts/train.py:280: betas (0.9,0.999), eps 1e-8, no weight decay, no amsgrad).
 *   grad_scale multiplies the gradient first (1/world_size after an all-reduce sum);
 *   max_norm > 0 applies clip_grad_norm_ (train_better_model.py:78) with the total norm taken
 *   from sumsq[0] (lr_sumsq over the same gradient buffer): coef = min(1, max_norm/(norm+1e-6)).
 *   step_count  [2] int32 DEVICE counters: [0] updates taken, incremented here (bias correction uses the new
 *               value), so a captured hipGraph replays correct corrections; [1] steps skipped (skip != 0 or a
 *               pending recurrence fault), so the caller can report them without a per-step host read;
 *   skip        [1] int32 DEVICE flag or NULL: non-zero = the reference skipped this batch
 *               (`continue`, train_better_model.py:49-50): nothing is updated, step_count stays;
 *   scratch     [4] floats of device scratch;
 *   dist_words  NULL, or (data parallel) the two summed words of lr_fault_export_f32 with `world` = the number of
 *               ranks: they then decide skip / fault instead of `skip` (a local fault still counts), and skip[0]
 *               is overwritten with the ranks' verdict (1 = every rank skipped the batch, else 0). */
int lr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                 const float* sumsq, float max_norm, float grad_scale, float lr, float beta1,
                 float beta2, float eps, int32_t* step_count, int32_t* skip, float* scratch,
                 const float* dist_words, float world, lr_stream_t stream);
/* lr_sumsq + lr_adam_step with max_norm > 0 in TWO launches instead of three: the sum-of-squares kernel's last workgroup
 * (a ticket in sumsq[1]) derives the step's coefficients.  sumsq [2]: {accumulator, ticket}, both zero before the first
 * call and put back to zero by every call (lr_step_begin's also_zero clears them as well: a launch that was torn down
 * half-way leaves nothing behind); scratch8: EIGHT floats of device scratch: [0..3] the step's coefficients, [5] the sum
 * of squares of this step; n_sumsq (<= n): the sum of squares is taken over grad[0 .. n_sumsq) only — the rest of the
 * buffer's was added to sumsq[0] by lr_sumsq calls earlier in the step (FusedAdam.sum_squares_early: the encoder's share,
 * beside the conv backward); n_sumsq = n: over everything; everything else as lr_adam_step. */
int lr_clip_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* sumsq,
                      float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps,
                      int32_t* step_count, int32_t* skip, float* scratch8, const float* dist_words, float world,
                      int64_t n_sumsq, lr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LIPREADING_HIP_H */
