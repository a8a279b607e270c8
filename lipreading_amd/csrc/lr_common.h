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

// ---- optional instrumentation (bench.py roofline leg; lr_misc.hip) ------------------------------
// While enabled, selected launches are issued through hipExtLaunchKernelGGL with a hipEvent pair
// that stamps the dispatch's own begin/end on the stream it runs on (what rocprofv3's kernel trace
// reports).  lr_prof_next hands out the pair for the next sample of a slot, or returns false.
enum {
  LR_PROF_RNN_FWD = 0, LR_PROF_RNN_BWD = 1,
  LR_PROF_CONV1_FWD = 2, LR_PROF_CONV2_FWD = 3, LR_PROF_CONV3_FWD = 4,
  LR_PROF_CONV2_DGRAD = 5, LR_PROF_CONV3_DGRAD = 6,
  LR_PROF_CONV1_WGRAD = 7, LR_PROF_CONV2_WGRAD = 8, LR_PROF_CONV3_WGRAD = 9,
  LR_PROF_SLOTS = 10
};
bool lr_prof_next(int slot, hipEvent_t* start, hipEvent_t* stop);
