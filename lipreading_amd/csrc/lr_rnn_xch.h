// lr_rnn_xch.h — what the one-launch recurrences share (lr_rnn_cluster.hip: clusters of ceil(H / 32) or ceil(H / 16)
// compute units per (direction, 8 samples); lr_rnn_grid.hip: one 24 x 8 grid of 192 compute units for 1152 < H <= 1536):
// the bf16 hi + lo split, the fast gate non-linearities, the self-tagged 32-bit exchange words and the loads / stores
// that move them between compute units, the inline-asm MFMA forms.  Moved out of lr_rnn_cluster.hip unchanged (round 6).
#pragma once
#include "lr_common.h"

namespace lrx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;
typedef unsigned u32;

__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}
// two values -> {hi0 | hi1 << 16}, {lo0 | lo1 << 16}
__device__ __forceinline__ void split_bf16_pair(float a, float b, u32& hi, u32& lo) {
  const bf16x2 h = __builtin_convertvector((f32x2){a, b}, bf16x2);
  const float ha = (float)h[0], hb = (float)h[1];
  const bf16x2 l = __builtin_convertvector((f32x2){a - ha, b - hb}, bf16x2);
  hi = __builtin_bit_cast(u32, h);
  lo = __builtin_bit_cast(u32, l);
}

// ---- gate non-linearities --------------------------------------------------------------------------------------
// The cell runs ONCE per thread and step, on the critical path of the step chain, and its transcendental functions
// were a sixth of a GRU-256 step: expf / tanhf of the device library cost ~10 / ~30+ instructions (range
// reduction, fix-ups, an IEEE division).  Here: v_exp_f32 and v_rcp_f32 (1 ulp each) — sigmoid to ~3e-7 relative,
// tanh to ~2e-7 ABSOLUTE (1 - 2 / (1 + e^2x): exact saturation at both ends, cancellation only where |tanh| is
// small), an order of magnitude inside the 2^-18 of the hi + lo operand split these kernels already work with.
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// ---- exchange words ------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 tag_of(int step) { return 1u + (u32)((step >> 1) % 3); }
__device__ __forceinline__ u32 xword(float v, u32 tag) {
  return ((__builtin_bit_cast(u32, v) + 2u) & ~3u) | tag;      // round to 22 mantissa bits, tag in the low two
}
__device__ __forceinline__ float xval(u32 w) { return __builtin_bit_cast(float, w & ~3u); }
__device__ __forceinline__ void publish(u32* p, u32 w, bool local) {
  // workgroup scope = the ISA's `sc0`: through the CU's write-through L1 into the XCD's L2, where it STAYS (an
  // agent-scope `sc1` store writes through and drops the line: every reader then goes to the fabric).  Only other
  // CUs of the same XCD are guaranteed to see it there — used when the cluster verified that it shares one XCD.
  if (local) __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32 peek(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// four words, L1-bypassing (`sc1`: served by the L2).  The compiler does not see the load: LR_VM_DRAIN + LR_TOUCH
// before the first use.
__device__ __forceinline__ u32x4 peek4(const u32* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
#define LR_VM_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define LR_TOUCH(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ int xcc_id() {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 0xf;
}

#define LR_MFMA_A0(acc, a, w) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(w))
#define LR_MFMA_A(acc, a, w) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w))
#define LR_MFMA_V(acc, a, w) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(w))
// The asm MFMAs are opaque to the compiler's hazard recogniser: let the last ones retire before any VALU instruction
// reads an accumulator — and TIE every accumulator to a statement behind the wait (LR_ACC_READY), or the reads are free
// to move in front of it: they are plain register arithmetic, which a volatile asm with a "memory" clobber does not
// order.  (Round 4 found <3,2>'s ISA adding acc0 + acc1 of two registers BETWEEN the last two MFMAs: the 2-member
// clusters were off by 1e-4 — the lo-plane product of the last k step — in half of the samples; every other
// instantiation happened to be scheduled the other way round.)
#define LR_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 3" ::: "memory")
#define LR_ACC_READY(acc) asm volatile("" : "+v"(acc))
// lanes 0-31: x + (x of lane + 32); lanes 32-63: y + (y of lane - 32).  (Inline asm: this ROCm's
// __builtin_amdgcn_permlane32_swap folds its two results into one register.  s_nop: the wait states a lane-crossing
// VALU read wants behind a VALU write of its operands, which the hazard recogniser cannot place around an asm.)
__device__ __forceinline__ float fold32(float x, float y) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  return x + y;
}


}  // namespace lrx
