// lr_tfm_rowblock.hip — the ROW-WISE half of a transformer encoder layer as one launch per direction (round 6).
//
// BUILD-DEFINED like the rest of the stage (lr_transformer.hip has the specification).  Everything
// of a post-LN encoder layer but the attention is row-wise: out-projection + residual + LayerNorm, feed-forward
// (ReLU) + residual + LayerNorm.  lr_transformer.hip ran those as five launches per layer and direction — products of
// 6-42 us at M = 2400 whose time is a launch's fixed latency, not its flops (VERDICT r5
// item 6) — with every intermediate [R][256] / [R][1024] tensor written and read back between them.  Here a
// workgroup owns 32 ROWS and walks the whole chain with the rows' activations in LDS:
//
//   forward :  a -> s1 = a Wo^T + bo + h -> h1 = LN1(s1) -> f1 = relu(h1 W1^T + b1) (256 hidden columns at a time)
//              -> s2 = f1 W2^T + b2 + h1 -> h2 = LN2(s2)   [-> the NEXT layer's qkv = h2 Wqkv'^T + bqkv']
//   backward:  [dh2 = dqkv' Wqkv' + ds1' of the layer ABOVE ->] ds2 = LN2'(dh2) -> df1 = (ds2 W2) . [f1 > 0]
//              (256 columns at a time) -> dh1 = df1 W1 + ds2 -> ds1 = LN1'(dh1) -> da = ds1 Wo
//
// and writes what lr_tfm_backward_weights and the other direction read (s1, stats, h1, f1, s2, stats / ds2, df1,
// ds1, da, LayerNorm partials) — the same tensors as before, so nothing else of the stack changes.  The bracketed
// products are the neighbouring layer's QKV projection and its data gradient (template argument X): they are row-wise
// too, and riding along saves their launches (18 / 35 us each for + 8 / + 14 us here).
//
// Shape of the kernel.  512 threads = 8 waves; wave w owns output columns 32 w .. 32 w + 31 of every product, so a
// product of the chain is 4 "groups" of 64 k: 48 v_mfma_f32_32x32x16_bf16 (hi hi + hi lo + lo hi, the stage's X3
// arithmetic) on one 32 x 32 accumulator.  The activation operand comes out of LDS (bf16 hi / lo planes [32][264],
// ds_read_b128, conflict-free at 528-byte rows); the WEIGHT operand never touches LDS: a lane's fragment — 8
// consecutive k of one output column — is a contiguous run of a [N][K] weight matrix, so lr_tfm_rb_pack (hi + lo of
// every weight and of its transpose, once per forward call) writes the planes in MFMA FRAGMENT ORDER — blocks of
// (32-column tile, 64-k group), inside [k16 step][lane][8 values] — and one load instruction of a wave reads 1 KB of
// consecutive bytes.  Nothing is shared between waves, so the weight stream (2.36 MB per workgroup and layer, out of
// the L2: every workgroup reads the same planes at about the same time) runs three groups ahead of the MFMAs through
// a ring of 96 registers, across barriers and epilogues.  One workgroup per CU, 75 of them at B x T = 2400.  Bound
// (stamps and switches below): that stream at ~52 B/clk per CU with the matrix pipe, two waves per
// SIMD, within a quarter of it.
#include "lr_common.h"
#include "lr_rnn_xch.h"

namespace {

using lrx::bf16_t;
using lrx::bf16x8;
using lrx::u32;
using lrx::u32x4;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RB = 32;          // rows of a workgroup
constexpr int DM = 256;         // model width = 8 waves x 32 columns
constexpr int NW = 8;
constexpr int PLD = DM + 8;     // bf16 elements per plane row (528 bytes: 16-byte slots of consecutive rows differ)
constexpr int SLD = DM + 4;     // floats per stage row
constexpr int PLANE = RB * PLD * 2;              // bytes of one bf16 plane
constexpr int OFF_P0 = 0;                        // planes 0: the input rows, later hidden chunk 0 (hi, lo)
constexpr int OFF_P1 = 2 * PLANE;                // planes 1: h1 / ds2 (the operand of the chunk products)
constexpr int OFF_P2 = 4 * PLANE;                // planes 2: hidden chunk 1
constexpr int OFF_ST = 6 * PLANE;                // fp32 stage [32][260]
constexpr int OFF_PAR = OFF_ST + RB * SLD * 4;   // 134 656: the layer's bias / LayerNorm vectors (fp32), staged once
// forward: bo, b2, g1, be1, g2, be2 (256 each), b1 (F); backward: g2, g1
constexpr int PAR_BO = 0, PAR_B2 = 256, PAR_G1 = 512, PAR_BE1 = 768, PAR_G2 = 1024, PAR_BE2 = 1280, PAR_B1 = 1536;
constexpr int lds_bytes(int F) { return OFF_PAR + (PAR_B1 + F + 768) * 4; }   // (+ bqkv behind b1)

struct PackJob {
  const float* src;
  bf16_t* hi;
  bf16_t* lo;
  int N, K, tile0, transpose;    // the operand M [N][K] = src [N][K], or (transpose) src^T with src [K][N]
};
constexpr int PACK_MAX = 64;
struct PackJobs {
  PackJob j[PACK_MAX];
  int n;
};

// fp32 weight -> bf16 hi + lo planes in FRAGMENT ORDER.  A plane is a sequence of (32-column tile nt, 64-k group g)
// blocks of 4 KB, tile-major; inside a block [k16 step s][lane l][8 values]: lane l's MFMA fragment of step s = column
// nt 32 + l % 32, k = 64 g + 32 (l / 32) + 8 s + 0..7.  So one load instruction of the chain kernels (a wave, one step)
// reads 1 KB of consecutive bytes: full cache lines.  (The first cut read row-major planes, 16 bytes per lane at a
// 512-byte pitch — 32 lines touched for 1 KB — and ran at the L1's line rate: 30 GB/s per CU, 79 us per launch.)
// One block per workgroup; the source tile goes through LDS so that both the straight and the transposed operand are
// read along the source's rows.
__global__ __launch_bounds__(256) void tfm_rb_pack_kernel(const PackJobs jobs) {
  __shared__ float t[32][65];
  int ji = 0;
  for (int i = 1; i < jobs.n; ++i)
    if ((int)blockIdx.x >= jobs.j[i].tile0) ji = i;
  const float* src = jobs.j[ji].src;
  const int N = jobs.j[ji].N, K = jobs.j[ji].K, tr = jobs.j[ji].transpose;
  const int blk = blockIdx.x - jobs.j[ji].tile0, gk = K / 64;
  const int n0 = (blk / gk) * 32, k0 = (blk % gk) * 64;
  if (!tr) {
    const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[r4 + 4 * i][c] = src[(int64_t)(n0 + r4 + 4 * i) * K + k0 + c];
  } else {
    const int r = threadIdx.x & 31, c8 = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[r][c8 + 8 * i] = src[(int64_t)(k0 + c8 + 8 * i) * N + n0 + r];
  }
  __syncthreads();
  const int st = threadIdx.x >> 6, l = threadIdx.x & 63;
  const float* row = &t[l & 31][32 * (l >> 5) + 8 * st];
  u32 h[4], lo4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) lrx::split_bf16_pair(row[2 * e], row[2 * e + 1], h[e], lo4[e]);
  const int64_t o = (int64_t)blk * 2048 + (st * 64 + l) * 8;
  *reinterpret_cast<u32x4*>(jobs.j[ji].hi + o) = u32x4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<u32x4*>(jobs.j[ji].lo + o) = u32x4{lo4[0], lo4[1], lo4[2], lo4[3]};
}

struct RbW {   // one weight of the chain as planes [N][K]
  const bf16_t* hi;
  const bf16_t* lo;
};

struct RbFwdArgs {
  const float* a;     // [R][256] attention output
  const float* h;     // [R][256] the layer's input (residual)
  RbW wo, w1, w2;     // [256][256], [F][256], [256][F]
  const float *bo, *b1, *b2, *g1, *be1, *g2, *be2;
  float *s1, *st1, *h1, *f1, *s2, *st2, *h2;
  // QKV: the NEXT layer's fused projection rides at the end of the chain: qkv = h2 Wqkv^T + bqkv
  RbW wq;             // [768][256]
  const float* bq;
  float* qkv;         // [R][768]
  int R, F, blk0;     // (blk0: set by the launcher, always 0 here)
  float eps;
};

struct RbBwdArgs {
  const float* dh2;   // [R][256] gradient of the layer's output
  const float *s2, *st2, *g2, *f1, *s1, *st1, *g1;
  RbW w2t, w1t, wot;  // W2^T [F][256], W1^T [256][F], Wo^T [256][256]
  float *ds2, *df1, *ds1, *da;
  float *lnp2, *lnp1;   // [kLnBlocks][2][256] partial sums of the LayerNorm parameter gradients
  // DQ: the chain starts one product earlier — dh2 = dqkv Wqkv + ds1 of the layer ABOVE (its input gradient)
  const float* dqkv_up;   // [R][768]
  const float* ds1_up;    // [R][256]
  RbW wqt;                // Wqkv^T [256][768] of the layer above
  int R, F, lnblocks, blk0;   // blk0: the first row block of this launch
};

// -DLR_RB_TIMING (tools/build_variant.sh; never in the product build): lane 0 of every wave of workgroup 0 stamps the
// clock at the phase boundaries; tools/probes/rb_timing.py reads them back through lr_tfm_rb_debug_times
#ifdef LR_RB_TIMING
__device__ long long g_rb_times[2][8][32];
#define LR_RB_T(dir, k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_rb_times[dir][threadIdx.x >> 6][k] = wall_clock64(); } while (0)
#else
#define LR_RB_T(dir, k) do { } while (0)
#endif

// ---- memory through buffer resources: a row past R (or a lane that has nothing to store) gets bit 31 set in its
// offset, is out of the resource's range and reads zeros / is dropped — no branch anywhere in the kernels, which are
// straight lines (the chunk loop is unrolled: NCH is a template argument), so that the compiler counts the weight
// stream's loads exactly (`s_waitcnt vmcnt(n)`; with branches around guarded stores it fell back to vmcnt(0) in
// front of every product and the stream ran one group at a time: 84 us per launch) ----
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, 0x7fffffff, 0x00020000);
}
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ float ld_f32(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ float4 ld_f32x4(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ void st_f32(rsrc_t r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);
}
__device__ __forceinline__ void st_f32x4(rsrc_t r, unsigned off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, 0);
}

// (barriers: lr_lds_barrier, which waits for LDS traffic only — __syncthreads would drain the stream's loads in flight)

// ---- the weight stream of a wave ----------------------------------------------------------------------------------
// group gi of the chain: phase gi >> 2 (one product of 256 k), quarter gi & 3 (64 k) = one 4 KB block of the hi plane
// and one of the lo plane (tfm_rb_pack_kernel's fragment order).  A lane holds column n = lane % 32 of the wave's 32
// and the k half kh = lane / 32; its fragment of k16 step s is the 16 bytes at block + 1024 s + 16 lane.
//   forward : phase 0 = Wo, then per chunk c: W1 rows 256 c + n, W2 columns 256 c + k
//   backward: per chunk c: W2^T rows 256 c + n, W1^T columns 256 c + k; the last phase = Wo^T
// The ring: THREE groups (96 registers) — the one being consumed and two in flight; a group's slot is refilled with
// the group three ahead as soon as its MFMAs are issued.  Everything is unrolled, so a group's slot (GI % 3) is static.
struct Ring {
  u32x4 hi[3][4], lo[3][4];
};
// VMEM instructions stay where they are written (the scheduler otherwise sinks the stream's loads down to their use to
// save registers, or lifts the MFMAs of the groups in between over them — either way the prefetch distance becomes
// zero): VMEM and MFMA instructions do not cross; VALU, SALU and LDS reads may
#define LR_RB_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x106)
struct Stream {
  rsrc_t first_hi, first_lo, a_hi, a_lo, b_hi, b_lo, c_hi, c_lo;   // planes of the chain's operands
  unsigned voff_a, voff_b, voff_c;   // lane * 16 + the wave's tile row in a plane with 4 / F / 64 / (backward) 12 groups per tile
};
// (measured and dropped, round 6: every wave taking a group's four k16 steps in its own rotated order, so that the
// eight waves and the launch's workgroups do not ask for the same 1 KB quarter of their 4 KB blocks at once — no
// change in the products' time, eight more live registers: 34.7 -> 36.0 us.)

// The chain as a sequence of phases (one product of 256 k each), NCH = hidden chunks, X = the extra projection:
//   forward : [Wo] [W1 c0] [W2 c0] [W1 c1] ... [W2 c(NCH-1)]  X: [Wqkv' 0] [Wqkv' 1] [Wqkv' 2]   (the next layer's)
//   backward: X: [Wqkv'^T k0] [k1] [k2]   [W2^T c0] [W1^T c0] ... [W1^T c(NCH-1)] [Wo^T]           (the layer above's)
// kind of a phase: 0 `first` (256 x 256), 1 `a` ([F][256]: tile = 8 c + wave), 2 `b` ([256][F]: group = 4 c + quarter),
// 3 `c` (forward [768][256]: tile = 8 j + wave; backward [256][768]: group = 4 j + quarter)
template <int NCH, bool BWD, bool X>
struct Chain {
  static constexpr int NPH = 1 + 2 * NCH + (X ? 3 : 0);
  static constexpr int kind(int p) {
    if (BWD) {
      if (X && p < 3) return 3;
      const int q = p - (X ? 3 : 0);
      return q >= 2 * NCH ? 0 : ((q & 1) == 0 ? 1 : 2);
    }
    if (p == 0) return 0;
    if (p > 2 * NCH) return 3;
    return ((p - 1) & 1) == 0 ? 1 : 2;
  }
  static constexpr int block(int p, int kq) {   // the group's block index, less the wave's share (in the lane offset)
    if (BWD) {
      if (X && p < 3) return 4 * p + kq;
      const int q = p - (X ? 3 : 0);
      return q >= 2 * NCH ? kq : ((q & 1) == 0 ? (q >> 1) * 32 + kq : (q >> 1) * 4 + kq);
    }
    if (p == 0) return kq;
    if (p > 2 * NCH) return (p - 2 * NCH - 1) * 32 + kq;
    return ((p - 1) & 1) == 0 ? ((p - 1) >> 1) * 32 + kq : ((p - 1) >> 1) * 4 + kq;
  }
};
// block (tile nt, group g) of a plane with GK groups per tile starts at (nt GK + g) 4096 bytes; step s at + 1024 s
template <int NCH, bool BWD, bool X, int GI>
__device__ __forceinline__ void ring_issue(Ring& ring, const Stream& sm) {
  typedef Chain<NCH, BWD, X> C;
  constexpr int P = GI >> 2, KQ = GI & 3, SLOT = GI % 3;
#ifdef LR_RB_EXP_NOSTREAM   // timing experiment: the chain without its weight stream (results are garbage)
  if constexpr (GI >= 3) return;
#endif
  if constexpr (P < C::NPH) {
    constexpr int K = C::kind(P), so = C::block(P, KQ) * 4096;
    const rsrc_t rh = K == 0 ? sm.first_hi : K == 1 ? sm.a_hi : K == 2 ? sm.b_hi : sm.c_hi;
    const rsrc_t rl = K == 0 ? sm.first_lo : K == 1 ? sm.a_lo : K == 2 ? sm.b_lo : sm.c_lo;
    const int vo = (int)(K == 2 ? sm.voff_b : K == 3 ? sm.voff_c : sm.voff_a);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ring.hi[SLOT][s] = __builtin_amdgcn_raw_buffer_load_b128(rh, vo + s * 1024, so, 0);
      ring.lo[SLOT][s] = __builtin_amdgcn_raw_buffer_load_b128(rl, vo + s * 1024, so, 0);
    }
  }
}
template <bool BWD>
__device__ __forceinline__ Stream make_stream(const RbW& first, const RbW& a, const RbW& b, const RbW& c, int F, int wave,
                                              int lane) {
  Stream sm;
  sm.first_hi = make_rsrc(first.hi); sm.first_lo = make_rsrc(first.lo);
  sm.a_hi = make_rsrc(a.hi); sm.a_lo = make_rsrc(a.lo);
  sm.b_hi = make_rsrc(b.hi); sm.b_lo = make_rsrc(b.lo);
  sm.c_hi = make_rsrc(c.hi); sm.c_lo = make_rsrc(c.lo);
  sm.voff_a = (unsigned)(lane * 16 + wave * 4 * 4096);
  sm.voff_b = (unsigned)(lane * 16 + wave * (F / 64) * 4096);
  sm.voff_c = BWD ? (unsigned)(lane * 16 + wave * 12 * 4096) : sm.voff_a;
  return sm;
}

// One product of the chain on a wave's accumulator: acc += A[32][256] (planes at `pa`, hi then lo) x the wave's 32
// columns of the phase's weight; behind a group's MFMAs the loads of the group three ahead go into its ring slot.
template <int NCH, bool BWD, bool X, int PHASE>
__device__ __forceinline__ void product(f32x16& acc, Ring& ring, const Stream& sm, const unsigned char* lds, int pa,
                                        int lane) {
  const int abase = pa + (lane & 31) * (PLD * 2) + (lane >> 5) * 64;
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) {
    const int slot = (PHASE * 4 + kq) % 3;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#ifdef LR_RB_EXP_NOLDS      // timing experiment: no LDS reads of the activation operand
      const u32x4 ah = u32x4{(u32)lane, 1u, 2u, 3u}, al = u32x4{4u, 5u, (u32)lane, 7u};
#else
      const u32x4 ah = *reinterpret_cast<const u32x4*>(lds + abase + kq * 128 + s * 16);
      const u32x4 al = *reinterpret_cast<const u32x4*>(lds + abase + PLANE + kq * 128 + s * 16);
#endif
      const bf16x8 a_hi = __builtin_bit_cast(bf16x8, ah), a_lo = __builtin_bit_cast(bf16x8, al);
      const bf16x8 b_hi = __builtin_bit_cast(bf16x8, ring.hi[slot][s]), b_lo = __builtin_bit_cast(bf16x8, ring.lo[slot][s]);
#ifdef LR_RB_EXP_X1         // timing experiment: one MFMA per step instead of three (the loads stay: lo joins by xor)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah ^ al), __builtin_bit_cast(bf16x8, ring.hi[slot][s] ^ ring.lo[slot][s]), acc, 0, 0, 0);
#else
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc, 0, 0, 0);
#endif
    }
    LR_RB_PIN_VMEM();
    if (kq == 0) ring_issue<NCH, BWD, X, PHASE * 4 + 3>(ring, sm);
    if (kq == 1) ring_issue<NCH, BWD, X, PHASE * 4 + 4>(ring, sm);
    if (kq == 2) ring_issue<NCH, BWD, X, PHASE * 4 + 5>(ring, sm);
    if (kq == 3) ring_issue<NCH, BWD, X, PHASE * 4 + 6>(ring, sm);
    LR_RB_PIN_VMEM();
  }
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// a wave's accumulator tile -> bf16 hi / lo planes at `pp` (columns 32 wave + lane % 32)
__device__ __forceinline__ void tile_to_planes(const f32x16& v, unsigned char* lds, int pp, int wave, int lane) {
  const int col = wave * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    bf16_t h, l;
    lrx::split_bf16(v[r], h, l);
    const int o = pp + (acc_row(r, lane) * PLD + col) * 2;
    *reinterpret_cast<bf16_t*>(lds + o) = h;
    *reinterpret_cast<bf16_t*>(lds + o + PLANE) = l;
  }
}

// byte offset of (row0 + row, col) in a [R][ld] fp32 tensor, out of range past R
__device__ __forceinline__ unsigned roff(int row0, int row, int R, int ld, int col) {
  return row0 + row < R ? (unsigned)((row0 + row) * ld + col) * 4u : OOB;
}

// rows [row0, row0 + 32) of x [R][256] fp32 -> planes at `pp` (zeros past R); 512 threads, 4 float4 each
__device__ __forceinline__ void rows_to_planes(const float* __restrict__ x, int row0, int R, int ld, int c0,
                                               unsigned char* lds, int pp) {
  const rsrc_t rx = make_rsrc(x);
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = threadIdx.x + 512 * i;
    v[i] = ld_f32x4(rx, roff(row0, u >> 6, R, ld, c0 + (u & 63) * 4));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = threadIdx.x + 512 * i, r = u >> 6, c4 = (u & 63) * 4;
    u32 h0, l0, h1, l1;
    lrx::split_bf16_pair(v[i].x, v[i].y, h0, l0);
    lrx::split_bf16_pair(v[i].z, v[i].w, h1, l1);
    const int o = pp + (r * PLD + c4) * 2;
    *reinterpret_cast<uint2*>(lds + o) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(lds + o + PLANE) = make_uint2(l0, l1);
  }
}

// four 64-lane sums at once, on DPP: two quad permutes, half-row mirror, row mirror (every lane of a 16-lane row then
// holds the row's sum), the four rows through v_readlane.  (__shfl_xor is ds_bpermute: six dependent LDS-crossbar
// round trips per sum — 1.9 us of a LayerNorm's 2.3, round 6 stamps.)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void wave_sum4(float (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = dpp_add<0xb1>(v[i]);    // quad_perm [1, 0, 3, 2]
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = dpp_add<0x4e>(v[i]);    // quad_perm [2, 3, 0, 1]
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = dpp_add<0x141>(v[i]);   // row_half_mirror
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = dpp_add<0x140>(v[i]);   // row_mirror
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = __builtin_bit_cast(int, v[i]);
    v[i] = (__builtin_bit_cast(float, __builtin_amdgcn_readlane(w, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(w, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(w, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(w, 48)));
  }
}

// LayerNorm of the stage's rows (wave w: rows 4 w .. 4 w + 3; a lane: 4 consecutive columns): y = LN(stage) gamma +
// beta -> `y` (global), stats; and, when `pp` >= 0, back into the stage and into the planes at pp
__device__ __forceinline__ void ln_rows(unsigned char* lds, int par_gamma, int par_beta,
                                        float* __restrict__ y, float* __restrict__ stats, int row0, int R, float eps,
                                        int pp, int wave, int lane) {
  float* stage = reinterpret_cast<float*>(lds + OFF_ST);
  const float* par = reinterpret_cast<const float*>(lds + OFF_PAR);
  const rsrc_t ry = make_rsrc(y), rs = make_rsrc(stats);
  const float4 g = *reinterpret_cast<const float4*>(par + par_gamma + 4 * lane);
  const float4 b = *reinterpret_cast<const float4*>(par + par_beta + 4 * lane);
  // the wave's four rows side by side: four independent reduction chains (one row after the other is four times the
  // shuffle latency: 2.3 us per LayerNorm, round 6 stamps)
  float4 v[4];
  float mean[4], var[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = *reinterpret_cast<const float4*>(stage + (4 * wave + i) * SLD + 4 * lane);
    mean[i] = (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  wave_sum4(mean);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    mean[i] *= 1.f / DM;
    v[i] = make_float4(v[i].x - mean[i], v[i].y - mean[i], v[i].z - mean[i], v[i].w - mean[i]);
    var[i] = (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  wave_sum4(var);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * wave + i;
    const float rstd = rsqrtf(var[i] * (1.f / DM) + eps);
    const float4 d = v[i];
    const float4 o = make_float4(d.x * rstd * g.x + b.x, d.y * rstd * g.y + b.y, d.z * rstd * g.z + b.z, d.w * rstd * g.w + b.w);
    st_f32x4(ry, roff(row0, r, R, DM, 4 * lane), o);
    st_f32(rs, lane < 2 ? roff(row0, r, R, 2, lane) : OOB, lane == 0 ? mean[i] : rstd);   // lanes 0, 1: (mean, rstd)
    if (pp >= 0) {
      *reinterpret_cast<float4*>(stage + r * SLD + 4 * lane) = o;
      u32 h0, l0, h1, l1;
      lrx::split_bf16_pair(o.x, o.y, h0, l0);
      lrx::split_bf16_pair(o.z, o.w, h1, l1);
      const int a = pp + (r * PLD + 4 * lane) * 2;
      *reinterpret_cast<uint2*>(lds + a) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(lds + a + PLANE) = make_uint2(l0, l1);
    }
  }
}

template <int NCH, bool X, int C>
__device__ __forceinline__ void fwd_chunks(const RbFwdArgs& p, Ring& ring, const Stream& sm, unsigned char* lds,
                                           f32x16& acc2, rsrc_t rf1, int row0, int wave, int lane) {
  if constexpr (C < NCH) {
    constexpr int hp = (C & 1) ? OFF_P2 : OFF_P0;
    const int col = wave * 32 + (lane & 31);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    product<NCH, false, X, 1 + 2 * C>(acc, ring, sm, lds, OFF_P1, lane);
    LR_RB_T(0, 8 + 4 * C);
    {
      const float bias = reinterpret_cast<const float*>(lds + OFF_PAR)[PAR_B1 + C * 256 + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] = fmaxf(acc[r] + bias, 0.f);
        st_f32(rf1, roff(row0, acc_row(r, lane), p.R, 256 * NCH, C * 256 + col), acc[r]);
      }
      tile_to_planes(acc, lds, hp, wave, lane);
    }
    LR_RB_T(0, 9 + 4 * C);
    lr_lds_barrier();
    LR_RB_T(0, 10 + 4 * C);
    product<NCH, false, X, 2 + 2 * C>(acc2, ring, sm, lds, hp, lane);
    LR_RB_T(0, 11 + 4 * C);
    fwd_chunks<NCH, X, C + 1>(p, ring, sm, lds, acc2, rf1, row0, wave, lane);
  }
}

// the extra projection of the forward chain: qkv[:, 256 j ..] = h2 Wqkv'[256 j ..]^T + bqkv'
template <int NCH, int J>
__device__ __forceinline__ void fwd_qkv(const RbFwdArgs& p, Ring& ring, const Stream& sm, unsigned char* lds, rsrc_t rq,
                                        int row0, int wave, int lane) {
  if constexpr (J < 3) {
    const int col = wave * 32 + (lane & 31);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    product<NCH, false, true, 1 + 2 * NCH + J>(acc, ring, sm, lds, OFF_P1, lane);
    const float bias = reinterpret_cast<const float*>(lds + OFF_PAR)[PAR_B1 + 256 * NCH + J * 256 + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) st_f32(rq, roff(row0, acc_row(r, lane), p.R, 768, J * 256 + col), acc[r] + bias);
    fwd_qkv<NCH, J + 1>(p, ring, sm, lds, rq, row0, wave, lane);
  }
}

template <int NCH, bool X>
__global__ __launch_bounds__(512, 1) void tfm_rb_fwd_kernel(const RbFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* stage = reinterpret_cast<float*>(lds + OFF_ST);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = blockIdx.x * RB, R = p.R;
  const int col = wave * 32 + (lane & 31);
  float* par = reinterpret_cast<float*>(lds + OFF_PAR);
  LR_RB_T(0, 0);
  // the layer's small vectors -> LDS, asked for IN FRONT of the weight stream: a load in the middle of the chain would
  // have to wait for every older load, i.e. drain the stream
  float pv[3 + NCH / 2 + 1], pq[2];
  {
    const float* src[3] = {threadIdx.x < 256 ? p.bo : p.b2, threadIdx.x < 256 ? p.g1 : p.be1, threadIdx.x < 256 ? p.g2 : p.be2};
#pragma unroll
    for (int i = 0; i < 3; ++i) pv[i] = src[i][threadIdx.x & 255];
#pragma unroll
    for (int i = 0; i < (256 * NCH + 511) / 512; ++i) {
      const int c = threadIdx.x + 512 * i;
      pv[3 + i] = c < 256 * NCH ? p.b1[c] : 0.f;
    }
    if constexpr (X) {
      pq[0] = p.bq[threadIdx.x];
      pq[1] = threadIdx.x < 256 ? p.bq[512 + threadIdx.x] : 0.f;
    }
  }
  LR_RB_PIN_VMEM();
  const Stream sm = make_stream<false>(p.wo, p.w1, p.w2, p.wq, 256 * NCH, wave, lane);
  Ring ring;
  ring_issue<NCH, false, X, 0>(ring, sm);
  ring_issue<NCH, false, X, 1>(ring, sm);
  ring_issue<NCH, false, X, 2>(ring, sm);
  LR_RB_PIN_VMEM();
  if constexpr (X) {
    par[PAR_B1 + 256 * NCH + threadIdx.x] = pq[0];
    if (threadIdx.x < 256) par[PAR_B1 + 256 * NCH + 512 + threadIdx.x] = pq[1];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) par[512 * i + threadIdx.x] = pv[i];
#pragma unroll
  for (int i = 0; i < (256 * NCH + 511) / 512; ++i)
    if (threadIdx.x + 512 * i < 256 * NCH) par[PAR_B1 + threadIdx.x + 512 * i] = pv[3 + i];
  rows_to_planes(p.a, row0, R, DM, 0, lds, OFF_P0);
  LR_RB_T(0, 1);
  lr_lds_barrier();
  LR_RB_T(0, 2);

  // s1 = a Wo^T + bo + h
  f32x16 acc, pre;   // pre: what an epilogue reads from memory, asked for in front of the product
  {
    const rsrc_t rh = make_rsrc(p.h);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] = 0.f;
      pre[r] = ld_f32(rh, roff(row0, acc_row(r, lane), R, DM, col));
    }
  }
  product<NCH, false, X, 0>(acc, ring, sm, lds, OFF_P0, lane);
  LR_RB_T(0, 3);
  {
    const rsrc_t rs1 = make_rsrc(p.s1);
    const float bias = par[PAR_BO + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, lane);
      const float v = acc[r] + bias + pre[r];
      st_f32(rs1, roff(row0, row, R, DM, col), v);
      stage[row * SLD + col] = v;
    }
  }
  LR_RB_T(0, 4);
  lr_lds_barrier();
  LR_RB_T(0, 5);
  ln_rows(lds, PAR_G1, PAR_BE1, p.h1, p.st1, row0, R, p.eps, OFF_P1, wave, lane);
  LR_RB_T(0, 6);
  lr_lds_barrier();
  LR_RB_T(0, 7);

  f32x16 acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  fwd_chunks<NCH, X, 0>(p, ring, sm, lds, acc2, make_rsrc(p.f1), row0, wave, lane);
  {
    const rsrc_t rs2 = make_rsrc(p.s2);
    const float bias = par[PAR_B2 + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, lane);
      const float v = acc2[r] + bias + stage[row * SLD + col];   // + h1
      st_f32(rs2, roff(row0, row, R, DM, col), v);
      stage[row * SLD + col] = v;
    }
  }
  LR_RB_T(0, 24);
  lr_lds_barrier();
  LR_RB_T(0, 25);
  ln_rows(lds, PAR_G2, PAR_BE2, p.h2, p.st2, row0, R, p.eps, X ? OFF_P1 : -1, wave, lane);
  LR_RB_T(0, 26);
  if constexpr (X) {
    lr_lds_barrier();
    fwd_qkv<NCH, 0>(p, ring, sm, lds, make_rsrc(p.qkv), row0, wave, lane);
    LR_RB_T(0, 27);
  }
}

// LayerNorm backward of the stage's rows: stage holds dy; x rows (the LayerNorm's input) and stats from memory.
// dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma -> `dx` (global), the stage and the planes at pp; the
// rows' share of dgamma / dbeta is summed over the workgroup's rows into partial[0 / 1][256].
__device__ __forceinline__ void ln_rows_bwd(unsigned char* lds, const float* __restrict__ x, const float* __restrict__ stats,
                                            int par_gamma, float* __restrict__ dx,
                                            float* __restrict__ partial, bool first, int row0, int R, int pp, int wave,
                                            int lane, float* red) {
  float* stage = reinterpret_cast<float*>(lds + OFF_ST);
  const rsrc_t rx = make_rsrc(x), rst = make_rsrc(stats), rdx = make_rsrc(dx);
  const float4 g = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(lds + OFF_PAR) + par_gamma + 4 * lane);
  float4 sxh = make_float4(0.f, 0.f, 0.f, 0.f), sdy = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 xv[4];
  float mean[4], rstd[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // (rows past R: x, mean and rstd read as zeros, dy is zero in the stage)
    const int r = 4 * wave + i;
    xv[i] = ld_f32x4(rx, roff(row0, r, R, DM, 4 * lane));
    mean[i] = ld_f32(rst, roff(row0, r, R, 2, 0));
    rstd[i] = ld_f32(rst, roff(row0, r, R, 2, 1));
  }
  float4 dy[4], xh[4], gg[4];
  float sg[4], sgx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dy[i] = *reinterpret_cast<const float4*>(stage + (4 * wave + i) * SLD + 4 * lane);
    const float mu = mean[i], rs = rstd[i];
    xh[i] = make_float4((xv[i].x - mu) * rs, (xv[i].y - mu) * rs, (xv[i].z - mu) * rs, (xv[i].w - mu) * rs);
    gg[i] = make_float4(dy[i].x * g.x, dy[i].y * g.y, dy[i].z * g.z, dy[i].w * g.w);
    sg[i] = (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
    sgx[i] = (gg[i].x * xh[i].x + gg[i].y * xh[i].y) + (gg[i].z * xh[i].z + gg[i].w * xh[i].w);
    sxh.x += dy[i].x * xh[i].x; sxh.y += dy[i].y * xh[i].y; sxh.z += dy[i].z * xh[i].z; sxh.w += dy[i].w * xh[i].w;
    sdy.x += dy[i].x; sdy.y += dy[i].y; sdy.z += dy[i].z; sdy.w += dy[i].w;
  }
  wave_sum4(sg);
  wave_sum4(sgx);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * wave + i;
    const float rs = rstd[i], a0 = sg[i] * (1.f / DM), a1 = sgx[i] * (1.f / DM);
    const float4 o = make_float4(rs * (gg[i].x - a0 - xh[i].x * a1), rs * (gg[i].y - a0 - xh[i].y * a1),
                                 rs * (gg[i].z - a0 - xh[i].z * a1), rs * (gg[i].w - a0 - xh[i].w * a1));
    st_f32x4(rdx, roff(row0, r, R, DM, 4 * lane), o);
    *reinterpret_cast<float4*>(stage + r * SLD + 4 * lane) = o;
    u32 h0, l0, h1, l1;
    lrx::split_bf16_pair(o.x, o.y, h0, l0);
    lrx::split_bf16_pair(o.z, o.w, h1, l1);
    const int a = pp + (r * PLD + 4 * lane) * 2;
    *reinterpret_cast<uint2*>(lds + a) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(lds + a + PLANE) = make_uint2(l0, l1);
  }
  // the workgroup's partial sums: waves in fixed order through `red` [8][2][256]
  *reinterpret_cast<float4*>(red + (wave * 2 + 0) * DM + 4 * lane) = sxh;
  *reinterpret_cast<float4*>(red + (wave * 2 + 1) * DM + 4 * lane) = sdy;
  lr_lds_barrier();
  {
    const int c = threadIdx.x;   // 512 threads = 2 x 256 columns
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * 2 * DM + c];
    // (a later launch over the next row blocks adds to the row; the first reads a zero from out of range)
    partial[c] = ld_f32(make_rsrc(partial), first ? OOB : (unsigned)c * 4u) + s;
  }
}

template <int NCH, bool X, int C>
__device__ __forceinline__ void bwd_chunks(const RbBwdArgs& p, Ring& ring, const Stream& sm, unsigned char* lds,
                                           f32x16& acc2, rsrc_t rf1, rsrc_t rdf1, int row0, int wave, int lane) {
  if constexpr (C < NCH) {
    constexpr int hp = (C & 1) ? OFF_P2 : OFF_P0;
    const int col = wave * 32 + (lane & 31);
    f32x16 acc, pre;   // pre: the chunk's f1 (the ReLU mask), asked for in front of the product
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] = 0.f;
      pre[r] = ld_f32(rf1, roff(row0, acc_row(r, lane), p.R, 256 * NCH, C * 256 + col));
    }
    product<NCH, true, X, (X ? 3 : 0) + 2 * C>(acc, ring, sm, lds, OFF_P1, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] = pre[r] > 0.f ? acc[r] : 0.f;
      st_f32(rdf1, roff(row0, acc_row(r, lane), p.R, 256 * NCH, C * 256 + col), acc[r]);
    }
    tile_to_planes(acc, lds, hp, wave, lane);
    lr_lds_barrier();
    product<NCH, true, X, (X ? 3 : 0) + 2 * C + 1>(acc2, ring, sm, lds, hp, lane);
    bwd_chunks<NCH, X, C + 1>(p, ring, sm, lds, acc2, rf1, rdf1, row0, wave, lane);
  }
}

template <int NCH, bool X>
__global__ __launch_bounds__(512, 1) void tfm_rb_bwd_kernel(const RbBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* stage = reinterpret_cast<float*>(lds + OFF_ST);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int R = p.R, row0 = (p.blk0 + blockIdx.x) * RB;
  const bool first = p.blk0 == 0;
  const int col = wave * 32 + (lane & 31);
  float* red = reinterpret_cast<float*>(lds + OFF_P2);   // [8][2][256] floats = 16 KB: inside planes 2 while they are idle
  const float pv = (threadIdx.x < 256 ? p.g2 : p.g1)[threadIdx.x & 255];   // (in front of the stream, as in the forward)
  LR_RB_PIN_VMEM();
  const Stream sm = make_stream<true>(p.wot, p.w2t, p.w1t, p.wqt, 256 * NCH, wave, lane);
  Ring ring;
  ring_issue<NCH, true, X, 0>(ring, sm);
  ring_issue<NCH, true, X, 1>(ring, sm);
  ring_issue<NCH, true, X, 2>(ring, sm);
  LR_RB_PIN_VMEM();
  // the partial rows nobody writes stay zero (the first workgroups clear the unused tail).  More than `lnblocks` row
  // blocks (8192 rows): the host launches the kernel again for the next lnblocks of them (blk0 > 0), and workgroup i
  // ADDS to partial row i — launches are ordered, so the sum's order is fixed
  for (int b = gridDim.x + blockIdx.x; b < (first ? p.lnblocks : 0); b += gridDim.x) {
    p.lnp2[(int64_t)b * 2 * DM + threadIdx.x] = 0.f;
    p.lnp1[(int64_t)b * 2 * DM + threadIdx.x] = 0.f;
  }
  reinterpret_cast<float*>(lds + OFF_PAR)[threadIdx.x] = pv;   // g2 at 0, g1 at 256
  if constexpr (X) {
    // dy = dqkv Wqkv + ds1 of the layer above: its 32 x 768 rows as three plane sets (all three are free here), three
    // products on one accumulator
    {
      const rsrc_t rx = make_rsrc(p.dqkv_up);
      float4 v[3][4];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int u = threadIdx.x + 512 * i;
          v[j][i] = ld_f32x4(rx, roff(row0, u >> 6, R, 768, 256 * j + (u & 63) * 4));
        }
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int u = threadIdx.x + 512 * i, r = u >> 6, c4 = (u & 63) * 4;
          u32 h0, l0, h1, l1;
          lrx::split_bf16_pair(v[j][i].x, v[j][i].y, h0, l0);
          lrx::split_bf16_pair(v[j][i].z, v[j][i].w, h1, l1);
          const int o = (j == 0 ? OFF_P0 : j == 1 ? OFF_P2 : OFF_P1) + (r * PLD + c4) * 2;
          *reinterpret_cast<uint2*>(lds + o) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(lds + o + PLANE) = make_uint2(l0, l1);
        }
    }
    f32x16 dacc, dpre;
    {
      const rsrc_t ru = make_rsrc(p.ds1_up);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dacc[r] = 0.f;
        dpre[r] = ld_f32(ru, roff(row0, acc_row(r, lane), R, DM, col));
      }
    }
    lr_lds_barrier();
    product<NCH, true, true, 0>(dacc, ring, sm, lds, OFF_P0, lane);
    product<NCH, true, true, 1>(dacc, ring, sm, lds, OFF_P2, lane);
    product<NCH, true, true, 2>(dacc, ring, sm, lds, OFF_P1, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[acc_row(r, lane) * SLD + col] = dacc[r] + dpre[r];
  } else {   // dy rows -> stage
    const rsrc_t rdy = make_rsrc(p.dh2);
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = threadIdx.x + 512 * i;
      v[i] = ld_f32x4(rdy, roff(row0, u >> 6, R, DM, (u & 63) * 4));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = threadIdx.x + 512 * i;
      *reinterpret_cast<float4*>(stage + (u >> 6) * SLD + (u & 63) * 4) = v[i];
    }
  }
  lr_lds_barrier();
  ln_rows_bwd(lds, p.s2, p.st2, 0, p.ds2, p.lnp2 + (int64_t)blockIdx.x * 2 * DM, first, row0, R, OFF_P1, wave, lane, red);
  lr_lds_barrier();   // planes 1 = ds2 (and the stage, fp32); `red` is free again

  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  bwd_chunks<NCH, X, 0>(p, ring, sm, lds, acc2, make_rsrc(p.f1), make_rsrc(p.df1), row0, wave, lane);
  // dh1 = df1 W1 + ds2 -> stage
#pragma unroll
  for (int r = 0; r < 16; ++r) stage[acc_row(r, lane) * SLD + col] += acc2[r];
  lr_lds_barrier();   // (every wave is past its last read of planes 2: `red` may be written)
  ln_rows_bwd(lds, p.s1, p.st1, 256, p.ds1, p.lnp1 + (int64_t)blockIdx.x * 2 * DM, first, row0, R, OFF_P1, wave, lane, red);
  lr_lds_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  product<NCH, true, X, (X ? 3 : 0) + 2 * NCH>(acc, ring, sm, lds, OFF_P1, lane);
  {
    const rsrc_t rda = make_rsrc(p.da);
#pragma unroll
    for (int r = 0; r < 16; ++r) st_f32(rda, roff(row0, acc_row(r, lane), R, DM, col), acc[r]);
  }
}

bool g_attr_set[2][8] = {};

}  // namespace

// ---- host side (declared in lr_common.h; lr_transformer.hip composes the stack) -------------------------------------
int lr_tfm_rb_supported(int Dm, int F, int nlayers) {
  return Dm == DM && (F == 256 || F == 512 || F == 1024 || F == 2048) && nlayers >= 1 && 8 * nlayers <= PACK_MAX;
}
// bf16 elements of one layer's planes: hi + lo of Wo, W1, W2, Wqkv and of their transposes
size_t lr_tfm_rb_plane_elems(int F) { return (size_t)4 * (4 * (size_t)DM * DM + 2 * (size_t)DM * F); }

// planes of layer l inside `planes`: [Wo | W1 | W2 | Wqkv | Wo^T | W1^T | W2^T | Wqkv^T] each as hi then lo
static void rb_layer_planes(void* planes, int l, int F, RbW out[8]) {
  bf16_t* b = (bf16_t*)planes + (size_t)l * lr_tfm_rb_plane_elems(F);
  const size_t sz[4] = {(size_t)DM * DM, (size_t)DM * F, (size_t)DM * F, 3 * (size_t)DM * DM};
  for (int i = 0; i < 8; ++i) {
    out[i].hi = b;
    out[i].lo = b + sz[i & 3];
    b += 2 * sz[i & 3];
  }
}

// weights: the stack's pointer table (2 + 12 per layer, torch's registration order)
int lr_tfm_rb_pack(const float* const* weights, void* planes, int F, int nlayers, hipStream_t st) {
  PackJobs jobs;
  int n = 0, tiles = 0;
  for (int l = 0; l < nlayers; ++l) {
    const float* const* W = weights + 2 + 12 * l;
    RbW pl[8];
    rb_layer_planes(planes, l, F, pl);
    const float* src[4] = {W[2], W[4], W[6], W[0]};
    const int Ns[4] = {DM, F, DM, 3 * DM}, Ks[4] = {DM, DM, F, DM};
    for (int t = 0; t < 2; ++t)
      for (int i = 0; i < 4; ++i) {
        PackJob& j = jobs.j[n++];
        j.src = src[i];
        j.hi = const_cast<bf16_t*>(pl[4 * t + i].hi);
        j.lo = const_cast<bf16_t*>(pl[4 * t + i].lo);
        j.N = t ? Ks[i] : Ns[i]; j.K = t ? Ns[i] : Ks[i]; j.tile0 = tiles; j.transpose = t;   // the OPERAND's shape
        tiles += (j.N / 32) * (j.K / 64);
      }
  }
  jobs.n = n;
  LR_LAUNCH(tfm_rb_pack_kernel, dim3(tiles), dim3(256), 0, st, jobs);
  return lr_launch_status();
}

static int rb_attr(const void* fn, int dir, int idx) {
  if (g_attr_set[dir][idx]) return LR_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(2048)) != hipSuccess) return LR_ERR_LAUNCH;
  g_attr_set[dir][idx] = true;
  return LR_OK;
}
#ifdef LR_RB_TIMING
extern "C" int lr_tfm_rb_debug_times(long long* out_host) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_rb_times), sizeof(long long) * 2 * 8 * 32) == hipSuccess ? 0 : -1;
}
#endif
// the kernels are instantiated for 1, 2, 4, 8 chunks of 256 hidden columns, with and without the extra projection
#define LR_RB_LAUNCH_(KERNEL, DIR, NCH_, X_, IDX_)                                                         \
  do {                                                                                                     \
    const int rc_ = rb_attr((const void*)KERNEL<NCH_, X_>, DIR, IDX_);                                     \
    if (rc_ != LR_OK) return rc_;                                                                          \
    const int nblk_ = (R + RB - 1) / RB;                                                                   \
    for (int b0_ = 0; b0_ < nblk_; b0_ += cap_) {                                                          \
      p.blk0 = b0_;                                                                                        \
      LR_LAUNCH((KERNEL<NCH_, X_>), dim3(nblk_ - b0_ > cap_ ? cap_ : nblk_ - b0_), dim3(512), lds_bytes(F), st, p); \
    }                                                                                                      \
  } while (0)
#define LR_RB_DISPATCH(KERNEL, DIR, XFLAG, CAP)                                                            \
  do {                                                                                                     \
    const int nch_ = F / 256, cap_ = (CAP);                                                                              \
    if (XFLAG) {                                                                                           \
      if (nch_ == 1) LR_RB_LAUNCH_(KERNEL, DIR, 1, true, 4);                                               \
      else if (nch_ == 2) LR_RB_LAUNCH_(KERNEL, DIR, 2, true, 5);                                          \
      else if (nch_ == 4) LR_RB_LAUNCH_(KERNEL, DIR, 4, true, 6);                                          \
      else LR_RB_LAUNCH_(KERNEL, DIR, 8, true, 7);                                                         \
    } else {                                                                                               \
      if (nch_ == 1) LR_RB_LAUNCH_(KERNEL, DIR, 1, false, 0);                                              \
      else if (nch_ == 2) LR_RB_LAUNCH_(KERNEL, DIR, 2, false, 1);                                         \
      else if (nch_ == 4) LR_RB_LAUNCH_(KERNEL, DIR, 4, false, 2);                                         \
      else LR_RB_LAUNCH_(KERNEL, DIR, 8, false, 3);                                                        \
    }                                                                                                      \
  } while (0)

// W: the layer's 12 weight pointers; the tensors are the layer's block of the reserve (lr_transformer.hip).
// Wnext / qkv_next (or NULL): the NEXT layer's weights and its qkv buffer — its fused projection rides along.
int lr_tfm_rb_forward(const void* planes, int l, const float* const* W, const float* a, const float* h, float* s1,
                      float* st1, float* h1, float* f1, float* s2, float* st2, float* h2, const float* const* Wnext,
                      float* qkv_next, int R, int F, float eps, hipStream_t st) {
  RbW pl[8], pn[8];
  rb_layer_planes(const_cast<void*>(planes), l, F, pl);
  rb_layer_planes(const_cast<void*>(planes), l + 1, F, pn);   // (only addresses; used when Wnext)
  RbFwdArgs p;
  p.a = a; p.h = h;
  p.wo = pl[0]; p.w1 = pl[1]; p.w2 = pl[2];
  p.bo = W[3]; p.b1 = W[5]; p.b2 = W[7]; p.g1 = W[8]; p.be1 = W[9]; p.g2 = W[10]; p.be2 = W[11];
  p.s1 = s1; p.st1 = st1; p.h1 = h1; p.f1 = f1; p.s2 = s2; p.st2 = st2; p.h2 = h2;
  p.wq = Wnext ? pn[3] : pl[3]; p.bq = Wnext ? Wnext[1] : W[1]; p.qkv = qkv_next;
  p.R = R; p.F = F; p.eps = eps;
  LR_RB_DISPATCH(tfm_rb_fwd_kernel, 0, Wnext != nullptr, 0x7fffffff);
  return lr_launch_status();
}

// dqkv_up / ds1_up (or NULL): the gradients of the layer ABOVE — then dh2 is not read: the chain starts with
// dh2 = dqkv_up Wqkv(l + 1) + ds1_up
int lr_tfm_rb_backward(const void* planes, int l, const float* const* W, const float* dh2, const float* dqkv_up,
                       const float* ds1_up, const float* s2, const float* st2, const float* f1, const float* s1,
                       const float* st1, float* ds2, float* df1, float* ds1, float* da, float* lnp2, float* lnp1,
                       int lnblocks, int R, int F, hipStream_t st) {
  RbW pl[8], pn[8];
  rb_layer_planes(const_cast<void*>(planes), l, F, pl);
  rb_layer_planes(const_cast<void*>(planes), l + 1, F, pn);
  RbBwdArgs p;
  p.dh2 = dh2; p.s2 = s2; p.st2 = st2; p.g2 = W[10]; p.f1 = f1; p.s1 = s1; p.st1 = st1; p.g1 = W[8];
  p.wot = pl[4]; p.w1t = pl[5]; p.w2t = pl[6];
  p.ds2 = ds2; p.df1 = df1; p.ds1 = ds1; p.da = da; p.lnp2 = lnp2; p.lnp1 = lnp1;
  p.dqkv_up = dqkv_up; p.ds1_up = ds1_up; p.wqt = dqkv_up ? pn[7] : pl[7];
  p.R = R; p.F = F; p.lnblocks = lnblocks;
  LR_RB_DISPATCH(tfm_rb_bwd_kernel, 1, dqkv_up != nullptr, lnblocks);   // (more row blocks than partial rows: more launches)
  return lr_launch_status();
}
