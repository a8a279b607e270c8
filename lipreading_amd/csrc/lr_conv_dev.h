// Device-side helpers shared by the conv frontend's translation units (lr_conv.hip, lr_conv1.hip, lr_conv_patch.hip).
// Build-defined subsystem: the reference has no conv frontend (regime X).
#pragma once
#include "lr_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;  // storage type

__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;  // round to nearest even (v_cvt_pk_bf16_f32 on gfx950)
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf2f(bf16_t v) {
  return __builtin_bit_cast(float, (unsigned)v << 16);
}

// ReLU -> 2x2 max-pool of one window in the forward epilogues: the bf16 value that is stored and the window's CODE:
// the position (row-major scan) of its FIRST maximum, torch's rule, decided on the values that would have been stored
// — or 4 when that maximum is 0, i.e. when ReLU blocks the window's gradient (no position matches 4: the backward
// kernels that un-pool on the way into LDS, unpool8 below, then need neither the pooled activation nor a compare).  The
// epilogues are VALU-bound (the layer-2 forward spends 6.6 k cycles of a tile's 81 k here, 48 windows per lane:
// s_memtime, round 3), so this is written to be few instructions: the four a + bias are converted two per
// instruction; ReLU is a packed signed-integer max with 0 on the bf16 BIT PATTERNS (a negative float — and -0 — is a
// negative int16; the non-negative ones order as integers the way they order as numbers); the window's maximum and
// its first position come out of ONE unsigned maximum over the keys (pattern << 16 | 3 - position): equal patterns
// leave the decision to the low bits, the smaller position wins.  17 VALU instead of 30 (round 2: float ReLU,
// sign-bit masks, a compare-and-select chain); results identical bit for bit.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void relu_pool4(float a0, float a1, float a2, float a3, float bias, bf16_t& best, int& arg) {
  const f32x2_t lo = {a0 + bias, a1 + bias}, hi = {a2 + bias, a3 + bias};
  const s16x2_t zero = {0, 0};
  const s16x2_t q01 = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, __builtin_convertvector(lo, bf16x2_t)), zero);
  const s16x2_t q23 = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, __builtin_convertvector(hi, bf16x2_t)), zero);
  const unsigned p01 = __builtin_bit_cast(unsigned, q01), p23 = __builtin_bit_cast(unsigned, q23);
  const unsigned k0 = (p01 << 16) | 3u, k1 = (p01 & 0xffff0000u) | 2u, k2 = (p23 << 16) | 1u, k3 = p23 & 0xffff0000u;
  const unsigned k01 = k0 > k1 ? k0 : k1, k23 = k2 > k3 ? k2 : k3;
  const unsigned k = k01 > k23 ? k01 : k23;
  best = (bf16_t)(k >> 16);
  arg = k < 0x10000u ? 4 : (int)(~k & 3u);
}

// The backward of ReLU -> MaxPool((1,2,2)) for 8 channels of one window, in registers: d = the window's pooled
// gradient (8 bf16), cc = its 8 codes (relu_pool4: 0..3 = position of the maximum, 4 = blocked by ReLU); o[j] = the
// gradient of position j (row-major in the window): d where the code is j, zero elsewhere.  Byte arithmetic on four
// codes at a time: x = codes ^ jjjj is zero exactly in the matching bytes, and with every byte of x <= 7 the
// byte-wise 8 - x (no borrows) has bit 3 set exactly there; (m << 5) - (m >> 3) spreads each such bit into 0xff of
// its byte; v_perm_b32 doubles the bytes into 16-bit lane masks.  20 VALU per position.
__device__ __forceinline__ void unpool8(const uint4 d, const uint2 cc, uint4 (&o)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned x0 = cc.x ^ (0x01010101u * (unsigned)j), x1 = cc.y ^ (0x01010101u * (unsigned)j);
    const unsigned m0 = (0x08080808u - x0) & 0x08080808u, m1 = (0x08080808u - x1) & 0x08080808u;
    const unsigned b0 = (m0 << 5) - (m0 >> 3), b1 = (m1 << 5) - (m1 >> 3);
    o[j] = make_uint4(d.x & __builtin_amdgcn_perm(b0, b0, 0x01010000u), d.y & __builtin_amdgcn_perm(b0, b0, 0x03030202u),
                      d.z & __builtin_amdgcn_perm(b1, b1, 0x01010000u), d.w & __builtin_amdgcn_perm(b1, b1, 0x03030202u));
  }
}

// Two ds_read_b64_tr_b16 (gfx950 LDS transpose read) -> one MFMA operand.  16 lanes read a
// [4 k][16 columns] block, 8 bytes each (lane s: row s>>2, columns 4(s&3)..4(s&3)+3), and lane L gets
// column L of the 4 rows; a0 / a1 address the lane's 8 bytes of k 0..3 / 4..7.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_tr_pair(const unsigned char* lds, int a0, int a1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}


}  // namespace
