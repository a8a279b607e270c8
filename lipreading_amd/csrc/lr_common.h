// lr_common.h — shared helpers for the gfx950 kernels (internal; the public ABI is
// include/lipreading_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/lipreading_hip.h"

#define LR_WAVE 64

#define LR_CHECK_ARG(cond)                 \
  do {                                     \
    if (!(cond)) return LR_ERR_INVALID_ARG; \
  } while (0)

// Launch check: hipGetLastError is cheap and does not synchronise.  The runtime keeps the last
// error of ANY earlier call on this thread (torch leaves benign ones such as hipErrorNotReady
// behind), so every entry point clears it before launching (lr_clear_error) and only then
// reads it back.
static inline void lr_clear_error() { (void)hipGetLastError(); }
static inline int lr_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? LR_OK : LR_ERR_LAUNCH;
}
#define LR_LAUNCH(kernel, grid, block, lds, stream, ...)                          \
  do {                                                                            \
    lr_clear_error();                                                             \
    hipLaunchKernelGGL(kernel, grid, block, lds, (hipStream_t)(stream), __VA_ARGS__); \
  } while (0)

static inline size_t lr_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#define LR_NEG_INF (-__builtin_inff())

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it
// waits for every outstanding GLOBAL store of the wave; inside a T-step recursion that streams
// its rows out to HBM that wait (~1 us) would sit on the critical path of every step.
__device__ __forceinline__ void lr_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// log(exp(a)+exp(b)+exp(c)) with the -inf convention of torch's CTC kernels
// (aten/native/LossCTC.cpp: lamax == -inf -> 0).
__device__ __forceinline__ float lr_lse3(float a, float b, float c) {
  float m = fmaxf(fmaxf(a, b), c);
  if (m == LR_NEG_INF) m = 0.f;
  return logf(expf(a - m) + expf(b - m) + expf(c - m)) + m;
}
__device__ __forceinline__ float lr_lse2(float a, float b) {
  float m = fmaxf(a, b);
  if (m == LR_NEG_INF) m = 0.f;
  return logf(expf(a - m) + expf(b - m)) + m;
}

__device__ __forceinline__ float lr_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// 64-lane butterfly reductions (wave = 64 on gfx950).
__device__ __forceinline__ float lr_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float lr_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- two-stage deterministic column sums (lr_misc.hip) -----------------------------------------
// stage 1: partial[rs][col] = sum over the rs-th row slice of x[row][col]   (LR_COLSUM_SPLITS slices)
// stage 2 (caller specific): out[col] = sum_rs partial[rs][col] in fixed order.
constexpr int LR_COLSUM_SPLITS = 32;
int lr_colsum_partial(const float* x, int ld, int rows, int ncol, float* partial, hipStream_t stream);
// stage 1 as a device function (256 threads = 4 row lanes x 64 columns; bx = 64-column block, by = row slice): the
// kernel of lr_colsum_partial, and the RIDER blocks of a grouped GEMM launch (LrRnnBiasJob below)
__device__ __forceinline__ void lr_colsum_partial_body(const float* __restrict__ x, int ld, int rows, int ncol,
                                                       float* __restrict__ partial, int bx, int by) {
  __shared__ float lr_part_[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = bx * 64 + cl;
  const int per = (rows + LR_COLSUM_SPLITS - 1) / LR_COLSUM_SPLITS;
  const int r0 = by * per;
  const int r1 = min(rows, r0 + per);
  float s = 0.f;
  if (col < ncol)
    for (int r = r0 + rl; r < r1; r += 4) s += x[(int64_t)r * ld + col];
  lr_part_[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && col < ncol)
    partial[(int64_t)by * ncol + col] = lr_part_[0][cl] + lr_part_[1][cl] + lr_part_[2][cl] + lr_part_[3][cl];
}
// The bias gradients of a recurrent layer — column sums of dG [rows][D][4][H] scattered into db_ih / db_hh (GRU: the
// input side's n gate is slot 2, the recurrent side's slot 3) — as a RIDER of the layer's grouped weight-gradient
// launch: the partial sums are extra workgroups of the GEMM launch, the fixed-order finish extra workgroups of its
// combine launch (round 5: two launches of a regime-R step instead of four).
struct LrRnnBiasJob {
  const float* dG;
  float* partial;      // [LR_COLSUM_SPLITS][D * 4 * H]
  float* db_ih[2];
  float* db_hh[2];
  int ld, rows, H, D, G, accumulate;
};
__device__ __forceinline__ void lr_rnn_bias_final_body(const LrRnnBiasJob& p, int col) {
  const int ncol = p.D * 4 * p.H, H = p.H;
  if (col >= ncol) return;
  float s = 0.f;
  for (int r = 0; r < LR_COLSUM_SPLITS; ++r) s += p.partial[(int64_t)r * ncol + col];
  const int d = col / (4 * H), slot = (col / H) & 3, j = col % H;
  float* ih = nullptr;
  float* hh = nullptr;
  if (p.G == 1) {
    if (slot == 0) {
      ih = p.db_ih[d] + j;
      hh = p.db_hh[d] + j;
    }
  } else if (p.G == 4) {
    ih = p.db_ih[d] + slot * H + j;
    hh = p.db_hh[d] + slot * H + j;
  } else {
    if (slot < 3) ih = p.db_ih[d] + slot * H + j;
    if (slot != 2) hh = p.db_hh[d] + (slot == 3 ? 2 : slot) * H + j;
  }
  if (ih) *ih = p.accumulate ? *ih + s : s;
  if (hh) *hh = p.accumulate ? *hh + s : s;
}

// ---- optional instrumentation (bench.py roofline leg; lr_misc.hip) ------------------------------
// While enabled, selected launches are issued through hipExtLaunchKernelGGL with a hipEvent pair
// that stamps the dispatch's own begin/end on the stream it runs on (what rocprofv3's kernel trace
// reports).  lr_prof_next hands out the pair for the next sample of a slot, or returns false.
enum {
  LR_PROF_RNN_FWD = 0, LR_PROF_RNN_BWD = 1,
  LR_PROF_CONV1_FWD = 2, LR_PROF_CONV2_FWD = 3, LR_PROF_CONV3_FWD = 4,
  LR_PROF_CONV2_DGRAD = 5, LR_PROF_CONV3_DGRAD = 6,
  LR_PROF_CONV1_WGRAD = 7, LR_PROF_CONV2_WGRAD = 8, LR_PROF_CONV3_WGRAD = 9,
  LR_PROF_CTC_ALPHA_BETA = 10, LR_PROF_CTC_GRAD = 11,
  LR_PROF_SLOTS = 12
};
bool lr_prof_next(int slot, hipEvent_t* start, hipEvent_t* stop);
// LR_LAUNCH that becomes a sampled (event-stamped) launch of `slot` while profiling is enabled;
// needs <hip/hip_ext.h> in the including file
#define LR_LAUNCH_PROF(slot, kernel, grid, block, lds, stream, ...)                                       \
  do {                                                                                                    \
    hipEvent_t lr_e0_, lr_e1_;                                                                            \
    lr_clear_error();                                                                                     \
    if (lr_prof_next(slot, &lr_e0_, &lr_e1_))                                                             \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, (hipStream_t)(stream), lr_e0_, lr_e1_, 0, __VA_ARGS__); \
    else                                                                                                  \
      hipLaunchKernelGGL(kernel, grid, block, lds, (hipStream_t)(stream), __VA_ARGS__);                   \
  } while (0)

// ---- conv frontend, first layer: patch-resident forward and weight gradient (lr_conv1.hip) -----------------
constexpr int LR_CONV1_WGRAD_WGS = 256;   // persistent workgroups of the first layer's weight gradient (one per CU), one slab each
int lr_conv1_forward(bool pool, bool u8, const void* X, const void* Wp, const float* bias, void* Y, unsigned char* code,
                     int frames, int T, int Hin, int Win, int Ho, int Wo, int relu, bool sample, hipEvent_t e0,
                     hipEvent_t e1, hipStream_t stream);
int lr_conv1_wgrad(bool pooled, bool u8, const void* X, const void* dZ, const void* code, float* slabs, float* bias_part,
                   int frames, int T, int Hin, int Win, int Ho, int Wo, bool sample, hipEvent_t e0, hipEvent_t e1,
                   hipStream_t stream);

// ---- conv frontend, patch-resident forward / data gradient of the 24-wide (3,5,5) layer (lr_conv_patch.hip) ----
int lr_conv_patch24(bool fwd, bool unpool, const void* X, const void* Wf, const float* bias, void* Y, unsigned char* code,
                    int F, int T, int Hin, int relu, bool sample, hipEvent_t e0, hipEvent_t e1, hipStream_t stream);

// ---- conv frontend, weight gradient of the stride-1 layers, second form (lr_conv_wgrad.hip) ----------
constexpr int LR_CONV_TR2_SLOTS = 85;   // slots per temporal tap: 3 x 85 = 255 workgroups, one slab each
int lr_conv_wgrad_tr2_supported(int layer, int F, int H);   // 160 KB of LDS hold the two tile buffers + the tile table
int lr_conv_wgrad_tr2(int layer, const void* X, const void* dZ, const void* code, float* slabs, int F, int T, int H,
                      bool sample, hipEvent_t e0, hipEvent_t e1, hipStream_t stream);

// ---- device-side fault words {pending, total} of the one-launch recurrences (lr_misc.hip; see
// include/lipreading_hip.h lr_fault_words_ptr).  NULL only when the allocation failed.
int32_t* lr_fault_words();
int lr_device_cus();                // compute units of the current device, 0 without one
int lr_debug_drop_member_value();
int lr_debug_cluster_disabled();    // test hook (lr_rnn_debug_disable_cluster, bit 0): lr_rnn_cluster_supported answers 0
int lr_debug_tune_value(int which);  // lr_rnn_debug_tune: exchange polling knobs of the cluster recurrence (0 forward, 1 backward)
int lr_debug_ns8();                 // (bit 4): the cluster recurrence keeps 8 samples per cluster at every batch (round 6's A/B)
int lr_debug_dwih_packed();         // (bit 3): the stored-bf16 first layer's dW_ih on the packed lr_xgemm path (round 5's A/B) instead of lr_fgemm
int lr_debug_wgrad_f32();           // (bit 2): LR_RNN_RECUR_SPLIT layers keep their weight gradients on the fp32 grouped GEMM   // test hook (lr_rnn_debug_drop_member): that member of every cluster / pair exits at once

// ---- recurrent layer pieces shared with lr_decoder.hip (implemented in lr_rnn.hip) --------------------
size_t lr_rnn_packed_w_floats(int G, int H);
size_t lr_rnn_packed_state_floats(int B, int H);
int lr_rnn_fold_bias(const float* b_ih, const float* b_hh, float* out, int G, int H, hipStream_t stream);
int lr_rnn_pack_w(const float* W, float* out, int G, int H, int transposed, hipStream_t stream);
int lr_rnn_pack_state(const float* h, float* slot, int B, int H, hipStream_t stream);
int lr_rnn_step_fwd(int G, float* gates, float* extra, float* y, float* hp, const int32_t* lens,
                    const float* wp, const float* b_hh, const float* h0, const float* c0, int B, int T,
                    int H, int step, hipStream_t stream);
int lr_rnn_step_bwd(int G, const float* gates, const float* extra, const float* y, const float* dy,
                    const float* dh_n, const float* dc_n, float* dG, float* dcar, float* dgp, const int32_t* lens,
                    const float* wpT, const float* h0, const float* c0, int B, int T, int H, int step,
                    hipStream_t stream);
int lr_rnn_dh0(int G, const float* dcar, const float* dgp_slot, const float* wpT, float* dh0, float* dc0, int B,
               int T, int H, hipStream_t stream);
int lr_rnn_bias_grads(const float* dG, float* partial, float* db_ih, float* db_hh, int rows, int H, int G,
                      int accumulate, hipStream_t stream);
int lr_sgemm_impl(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                  const float* B, int ldb, float beta, float* C, int ldc, const float* bias, int row_shift,
                  int period, void* workspace, size_t workspace_bytes, hipStream_t stream);
int lr_sgemm_batched_impl(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                          int64_t sA, const float* B, int ldb, int64_t sB, float beta, float* C, int ldc,
                          int64_t sC, const float* bias, int batch, hipStream_t stream);
int lr_sgemm_batched_bias_impl(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                               int64_t sA, const float* B, int ldb, int64_t sB, float beta, float* C, int ldc,
                               int64_t sC, const float* bias, int64_t sBias, int batch, hipStream_t stream);
extern "C" size_t lr_sgemm_workspace_bytes(int M, int N, int K);
// up to 8 products C_i = A_i^T B_i (+ beta C_i) in one launch + one fixed-order combine (lr_gemm.hip)
size_t lr_sgemm_grouped_workspace_bytes(int n, const int* M, const int* N, const int* K);
int lr_sgemm_grouped_tn_impl(int n, const int* M, const int* N, const int* K, const float* const* A, const int* lda,
                             const float* const* B, const int* ldb, float* const* C, const int* ldc, float beta,
                             const int* row_shift, const int* period, void* workspace, size_t workspace_bytes,
                             hipStream_t stream, const LrRnnBiasJob* rider = nullptr);
// lr_fgemm.hip: products straight from the tensors as they lie in memory (include/lipreading_hip.h lr_fgemm)
int lr_fgemm_launch(int prec, int form, int a_bf16, int b_bf16, const lr_fgemm_job* jobs, int njobs, hipStream_t stream);
int lr_fgemm_want_splits(int M, int N, int K);
size_t lr_fgemm_slab_floats_impl(int M, int N, int splits);
// lr_rnn_cluster.hip: the GRU / LSTM recurrence as one launch per layer pass, fp32-faithful (W_hh sliced over a
// cluster of ceil(H / 32) CUs per (direction, 8 samples), bf16 hi + lo planes, self-tagged 4-byte exchange words);
// G = 3 (GRU) or 4 (LSTM); h0 / c0 (may be NULL) = the state before the first step, [D][B][H]; dh0 / dc0 (may be
// NULL) receive the gradient into it
int lr_rnn_cluster_supported(int G, int B, int H);
int lr_rnn_cluster_cus(int G, int H);   // compute units the SMALLEST launch needs resident together; 0: no kernel for the shape
int lr_rnn_cluster_launches(int G, int B, int H, int D);   // recurrence launches per layer pass on this device
size_t lr_rnn_cluster_pack_bytes(int G, int H, int D, int backward);
size_t lr_rnn_cluster_xch_bytes(int B, int H, int D, int backward);
// prologue_done: lr_rnn_cluster_prologue already ran for this pass (W_hh packed into wpack, the first launch's
// exchange words cleared) — one launch that also folds the layer's biases into bias_out [D][G*H] for the input projection
int lr_rnn_cluster_forward(int G, float* gates, float* extra, float* y, const float* const* w_hh, const float* const* b_hh,
                           const float* h0, const float* c0, const int32_t* lens, void* wpack, void* xch, int B, int T,
                           int D, int H, hipStream_t stream, int prologue_done = 0);
// wpack_b / xch_b (may be NULL): the BACKWARD pass's fragments (lr_rnn_cluster_pack_bytes(.., 1)) and exchange words
// (lr_rnn_cluster_xch_bytes(.., 1)) prepared by the same launch -> lr_rnn_cluster_backward(.., pack_done = 1)
int lr_rnn_cluster_prologue(int G, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                            float* bias_out, void* wpack, void* xch, int B, int D, int H, hipStream_t stream,
                            void* wpack_b = nullptr, void* xch_b = nullptr);
int lr_rnn_cluster_backward(int G, const float* gates, const float* extra, const float* y, const float* dy,
                            const float* dh_n, const float* dc_n, float* dG, float* dh0, float* dc0, const float* h0,
                            const float* c0, const float* const* w_hh, const int32_t* lens, void* wpack, void* xch, int B,
                            int T, int D, int H, hipStream_t stream, int pack_done = 0);
// lr_rnn_grid.hip: the LSTM recurrence for 1152 < H <= 1536 (the 1400 / 1536-unit decoders behind BiLSTM-700 / 768) as one
// launch per pass on a 24 x 8 grid of 192 compute units; reached through lr_rnn_cluster_* above, which delegate
int lr_rnn_grid_shape(int G, int H);   // 1: this file's kernels cover the shape
int lr_rnn_grid_cus();
int lr_rnn_grid_launches(int B, int D);
size_t lr_rnn_grid_pack_bytes(int D);
size_t lr_rnn_grid_xch_bytes(int B, int backward);
int lr_rnn_grid_prologue(const float* const* w_hh, const float* const* b_ih, const float* const* b_hh, float* bias_out,
                         void* wpack, void* xch, int B, int D, int H, hipStream_t stream, void* wpack_b, void* xch_b);
int lr_rnn_grid_forward(float* gates, float* extra, float* y, const float* const* w_hh, const float* h0, const float* c0,
                        const int32_t* lens, void* wpack, void* xch, int B, int T, int D, int H, hipStream_t stream,
                        int prologue_done);
int lr_rnn_grid_backward(const float* gates, const float* extra, const float* dy, const float* dh_n, const float* dc_n,
                         float* dG, float* dh0, float* dc0, const float* c0, const float* const* w_hh, const int32_t* lens,
                         void* wpack, void* xch, int B, int T, int D, int H, hipStream_t stream, int pack_done);
// lr_xgemm.hip: fp32 GEMM on the bf16 matrix cores by hi/lo operand splitting (same operand
// conventions; a_exact / b_exact: the operand's elements are bf16 values already)
int lr_xgemm_impl(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                  const float* B, int ldb, float beta, float* C, int ldc, const float* bias, int a_exact,
                  int b_exact, void* workspace, size_t workspace_bytes, hipStream_t stream);
extern "C" size_t lr_xgemm_workspace_bytes(int transA, int transB, int M, int N, int K);
// the three products of a recurrent layer's input projection with all D directions in one
// contraction each (gates / dG hold the directions side by side in a row; dstride = floats between
// the directions' blocks of a dG row)
size_t lr_xproj_workspace_bytes(int R, int I, int GH, int D, int H);
// (one_product: every operand as its bf16 hi plane only, LR_RNN_PROJ_BF16X1)
int lr_xproj_forward(const float* x, int R, int I, const float* const* w_ih, int GH, int D, const float* bias,
                     float* gates, int x_exact, int x_bf16, void* workspace, size_t workspace_bytes,
                     hipStream_t stream, int one_product = 0);
int lr_xproj_dw(const float* dG, int ldg, int dstride, const float* x, int R, int I, int GH, int D,
                float* const* dw_ih, float beta, int x_exact, int x_bf16, void* workspace, size_t workspace_bytes,
                hipStream_t stream, int one_product = 0);
int lr_xproj_dx(const float* dG, int ldg, int dstride, const float* const* w_ih, int R, int I, int GH, int D,
                float* dx, int hi_only, int dx_bf16, void* workspace, size_t workspace_bytes, hipStream_t stream);
// dW_ih and dW_hh from ONE pack of dG (layers whose recurrent side reads dG slots 0..G-1: LSTM, tanh RNN)
size_t lr_xproj_dw_both_workspace_bytes(int R, int I, int GH, int H, int D);
int lr_xproj_dw_both(const float* dG, int ldg, int dstride, const float* x, const float* y, int ldy, int R, int T, int I,
                     int H, int GH, int D, float* const* dw_ih, float* const* dw_hh, float beta, void* workspace,
                     size_t workspace_bytes, hipStream_t stream);
int lr_xproj_dwhh(const float* dG, int ldg, const float* y, int ldy, int R, int T, int H, int G, int D,
                  float* const* dw_hh, float beta, void* workspace, size_t workspace_bytes, hipStream_t stream,
                  int one_product = 0);

// ---- lr_tfm_rowblock.hip: the row-wise half of an encoder layer as one launch per direction (LR_TFM_ROWBLOCK) ----
int lr_tfm_rb_supported(int Dm, int F, int nlayers);
size_t lr_tfm_rb_plane_elems(int F);   // bf16 elements of one layer's weight planes
int lr_tfm_rb_pack(const float* const* weights, void* planes, int F, int nlayers, hipStream_t st);
int lr_tfm_rb_forward(const void* planes, int l, const float* const* W, const float* a, const float* h, float* s1,
                      float* st1, float* h1, float* f1, float* s2, float* st2, float* h2, const float* const* Wnext,
                      float* qkv_next, int R, int F, float eps, hipStream_t st);
int lr_tfm_rb_backward(const void* planes, int l, const float* const* W, const float* dh2, const float* dqkv_up,
                       const float* ds1_up, const float* s2, const float* st2, const float* f1, const float* s1,
                       const float* st1, float* ds2, float* df1, float* ds1, float* da, float* lnp2, float* lnp1,
                       int lnblocks, int R, int F, hipStream_t st);
