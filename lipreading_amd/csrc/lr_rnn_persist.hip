// lr_rnn_persist.hip — the GRU-256 recurrence as ONE launch per layer pass (pixel regime).
//
// The step-per-launch recurrence of lr_rnn.hip is bound by launch boundaries and memory round trips
// (~5 us per step whatever the arithmetic), because fp32 W_hh (768 KB per direction) has to be
// re-streamed from the fabric every step.  BASELINE.json's configs[1] (the pixel regime: conv
// frontend + BiGRU-256, "bf16") allows a different trade: with W_hh rounded to bf16 (384 KB per
// direction) the whole matrix fits ONE compute unit — as MFMA operand fragments held in the registers
// of 4 waves (240 of each wave's 256 AGPRs + 96-144 VGPRs; the forward kernel keeps its last k step,
// 48 KB, in LDS) — and batch rows
// are independent, so every (sample, direction) gets its own workgroup that runs all T steps with
// nothing but one workgroup barrier per step:
//   state (bf16, double-buffered in LDS, row 0 of the 16-row MFMA operand) x W_hh^T on
//   v_mfma_f32_16x16x32_bf16 (96 per wave per step), fp32 accumulation, fp32 gate math and fp32
//   carried state in registers.  The backward kernel is the mirror image (dGh x W_hh over K = 768).
// The AGPR-resident fragments are read by the matrix core directly: the MFMAs are inline asm with an
// "a" operand (the compiler's own MFMA selection first copies AGPR-resident operands to VGPRs — 57
// cycles per MFMA instead of ~17).  Fragments are pre-packed (gru256_pack_whh*_kernel) so the prologue
// is 1 KB-per-wave loads: reading row-major fp32 W_hh in the prologue cost 85 us per launch.
// The interface buffers are those of the step kernels (gates in/out, extra, y, dG: fp32), so the
// one-shot GEMMs around the recurrence (input projection, weight gradients) are unchanged, and so
// is the reference-faithful fp32 path, which never comes here.  No reference counterpart beyond
// better_model.py:74 (nn.GRU); precision: bf16 operands in the recurrent product only.
//
// STATUS: what PixelLipReader runs (VideoEncoder.recurrence = 'bf16'; tests/test_gpu_encoder.py
// checks it against the step kernels, ragged lengths included).  MI355X, B = 32, T = 75: 104 us per
// forward layer pass (1.4 us per step), 127 us per backward pass (1.7 us), against 5.4 / 4.8 us per
// step for the step kernels.  History: 16 samples per workgroup 6.7 us/step; 4 samples 4.0; AGPR
// operands 3.4; one sample per workgroup 2.7; pre-packed fragments 1.5; operands fetched two steps
// ahead and (nearly) all fragments in registers 1.4.
#include "lr_common.h"
#include <hip/hip_ext.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

constexpr int PH = 256;            // hidden size this kernel is built for
constexpr int PKS = PH / 32;       // k steps of 32 (forward: k = previous state)
constexpr int PKS_A = 5;           // k steps whose weight fragments live in AGPRs (12 x 5 x 4 = 240 of the 256)
constexpr int PKS_REG = 8;         // k steps held in registers (AGPR + VGPR); any rest would sit in LDS (none: all 96 fragments fit)
constexpr int PNT = 12;            // 16-unit column tiles per wave: 3 gates x 4
constexpr int PHLD = PH + 8;       // bf16 per LDS row of the state (528 B: conflict-free b128 rows)
constexpr int PBH = 1;             // samples per workgroup (rows >= PBH of the 16-row MFMA operand are zero)
constexpr size_t PFWD_LDS = (size_t)2 * 16 * PHLD * 2 + (size_t)4 * PNT * (PKS - PKS_REG) * 1024 +
                            (size_t)3 * PBH * PH * 4;

__device__ __forceinline__ bf16x8 pack8(const float4& a, const float4& b) {
  bf16x8 v;
  v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
  v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
  return v;
}
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}

// W_hh [3*256][256] fp32 of each direction -> bf16 MFMA B fragments in the order the persistent kernel
// consumes them: out[((d*4 + wave)*PNT + tl)*PKS + ks][lane] (8 bf16 = k 32ks + 8kg .. +7 of unit
// 64 wave + 16 (tl&3) + col, gate tl>>2; lane = kg*16 + col).
__global__ void gru256_pack_whh_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                       bf16x8* __restrict__ out, int D) {
  const int total = D * 4 * PNT * PKS * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, ks = (i >> 6) % PKS, tl = (i / (64 * PKS)) % PNT, wave = (i / (64 * PKS * PNT)) & 3;
    const int d = i / (64 * PKS * PNT * 4);
    const int col = lane & 15, kg = lane >> 4, g = tl >> 2, nt = tl & 3;
    const float* row = (d ? w1 : w0) + ((int64_t)g * PH + 64 * wave + 16 * nt + col) * PH + ks * 32 + kg * 8;
    out[i] = pack8(*reinterpret_cast<const float4*>(row), *reinterpret_cast<const float4*>(row + 4));
  }
}

// Twelve v_mfma_f32_16x16x32_bf16 (one per column tile) sharing the A fragment, weights named with
// register-class constraint WC ("a": AGPR, read by the matrix core directly; "v": VGPR).
#define LR_MFMA12(acc, a, WC, w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11)                                  \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %12, %13, %0\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %1, %12, %14, %1\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %2, %12, %15, %2\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %3, %12, %16, %3\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %4, %12, %17, %4\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %5, %12, %18, %5\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %6, %12, %19, %6\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %7, %12, %20, %7\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %8, %12, %21, %8\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %9, %12, %22, %9\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %10, %12, %23, %10\n\t"                                                 \
               "v_mfma_f32_16x16x32_bf16 %11, %12, %24, %11"                                                      \
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), \
                 "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11])                           \
               : "v"(a), WC(w0), WC(w1), WC(w2), WC(w3), WC(w4), WC(w5), WC(w6), WC(w7), WC(w8), WC(w9), WC(w10),  \
                 WC(w11))
// first k step: C = 0 (inline constant), accumulators are pure outputs; weights W[tl][ks] in AGPRs
#define LR_MFMA12_FIRST(acc, a, W, ks)                                                                            \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %12, %13, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %1, %12, %14, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %2, %12, %15, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %3, %12, %16, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %4, %12, %17, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %5, %12, %18, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %6, %12, %19, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %7, %12, %20, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %8, %12, %21, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %9, %12, %22, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %10, %12, %23, 0\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %11, %12, %24, 0"                                                        \
               : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]), "=&v"(acc[4]), "=&v"(acc[5]),         \
                 "=&v"(acc[6]), "=&v"(acc[7]), "=&v"(acc[8]), "=&v"(acc[9]), "=&v"(acc[10]), "=&v"(acc[11])       \
               : "v"(a), "a"(W[0][ks]), "a"(W[1][ks]), "a"(W[2][ks]), "a"(W[3][ks]), "a"(W[4][ks]), "a"(W[5][ks]), \
                 "a"(W[6][ks]), "a"(W[7][ks]), "a"(W[8][ks]), "a"(W[9][ks]), "a"(W[10][ks]), "a"(W[11][ks]))

// grid (sample groups of PBH, directions); 256 threads.  Product phase: wave w owns hidden units
// [64w, 64w + 64) as column tiles tl = gate * 4 + nt (units 64w + 16nt .. +15 of that gate); its
// results (rows = the group's samples) go through LDS so that the gate phase runs one hidden unit
// per THREAD over the group's samples, with every global access a contiguous 1 KB per wave.
__global__ __launch_bounds__(256, 1) void gru256_fwd_persist_kernel(float* __restrict__ gates,
                                                                    float* __restrict__ extra,
                                                                    float* __restrict__ y,
                                                                    const bf16x8* __restrict__ wpk,
                                                                    const float* __restrict__ bhh0,
                                                                    const float* __restrict__ bhh1,
                                                                    const int32_t* __restrict__ lens, int B, int T,
                                                                    int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* hS = reinterpret_cast<bf16_t*>(smem);                                      // [2][16][PHLD]
  bf16x8* Wl = reinterpret_cast<bf16x8*>(smem + (size_t)2 * 16 * PHLD * 2);          // [4][PNT][PKS-PKS_REG][64]
  float* S = reinterpret_cast<float*>(smem + (size_t)2 * 16 * PHLD * 2 +
                                      (size_t)4 * PNT * (PKS - PKS_REG) * 1024);     // [3][PBH][PH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * PBH, d = blockIdx.y;
  const float* bhh = d ? bhh1 : bhh0;
  const int col = lane & 15, kg = lane >> 4;

  // ---- W_hh -> bf16 MFMA B fragments: lane (column = unit, k group) holds 8 consecutive k ---------
  // k steps 0..PKS_A-1 in AGPRs (read by the MFMAs directly: the inline asm below names them with the
  // "a" constraint; the compiler's own MFMA selection only ever copies AGPR-resident values back to
  // VGPRs first, which made a step's 96 MFMAs cost 57 cycles apiece), PKS_A..PKS_REG-1 in VGPRs
  bf16x8 Wa[PNT][PKS_A], Wv[PNT][PKS_REG - PKS_A];
  // fragments come pre-packed (gru256_pack_whh_kernel): 1 KB contiguous per wave load
  const bf16x8* wsrc = wpk + ((int64_t)(d * 4 + wave) * PNT * PKS) * 64 + lane;
#pragma unroll
  for (int tl = 0; tl < PNT; ++tl) {
#pragma unroll
    for (int ks = 0; ks < PKS; ++ks) {
      const bf16x8 f = wsrc[(tl * PKS + ks) * 64];
      if (ks < PKS_A) Wa[tl][ks] = f;
      else if (ks < PKS_REG) Wv[tl][ks - PKS_A] = f;
      else Wl[((wave * PNT + tl) * (PKS - PKS_REG) + (ks - PKS_REG)) * 64 + lane] = f;
    }
  }
  for (int i = tid; i < 2 * 16 * PHLD; i += 256) hS[i] = 0;

  // gate phase: thread = hidden unit `tid`, loop over the group's samples
  const float bhn = bhh[2 * PH + tid];
  int len[PBH];
  float hreg[PBH];
  struct Gx { float v[3][PBH]; };
  Gx gxA, gxB;   // pre-activations of steps s (even) / s (odd): each set is fetched TWO steps ahead
#pragma unroll
  for (int k = 0; k < PBH; ++k) {
    len[k] = b0 + k < B ? lens[b0 + k] : 0;
    hreg[k] = 0.f;
  }
  // input-projection pre-activations, fetched one step ahead
  auto time_of = [&](int s) {   // time index of step s (clamped to the last step)
    const int sc = s < T ? s : T - 1;
    return d == 0 ? sc : T - 1 - sc;
  };
  auto fetch_gx = [&](Gx& gx, int t, int k) {
    const int b = b0 + k < B ? b0 + k : 0;
    const float* gp = gates + (((int64_t)b * T + t) * D + d) * (3 * PH) + tid;
    gx.v[0][k] = gp[0];
    gx.v[1][k] = gp[PH];
    gx.v[2][k] = gp[2 * PH];
  };
#pragma unroll
  for (int k = 0; k < PBH; ++k) {
    fetch_gx(gxA, time_of(0), k);
    fetch_gx(gxB, time_of(1), k);
  }
  __syncthreads();

  auto step = [&](int s, Gx& gx) {
    const int t = time_of(s);
    const int tnext = time_of(s + 2);
    const bf16_t* hcur = hS + (s & 1) * 16 * PHLD;
    bf16_t* hnxt = hS + ((s + 1) & 1) * 16 * PHLD;
    f32x4 acc[PNT];
    bf16x8 a_next = *reinterpret_cast<const bf16x8*>(hcur + col * PHLD + kg * 8);   // row = sample
#pragma unroll
    for (int ks = 0; ks < PKS; ++ks) {
      const bf16x8 a = a_next;   // the next k step's state fragment is read while this one multiplies
      if (ks + 1 < PKS) a_next = *reinterpret_cast<const bf16x8*>(hcur + col * PHLD + (ks + 1) * 32 + kg * 8);
      if (ks == 0) {
        LR_MFMA12_FIRST(acc, a, Wa, 0);
      } else if (ks < PKS_A) {
        LR_MFMA12(acc, a, "a", Wa[0][ks], Wa[1][ks], Wa[2][ks], Wa[3][ks], Wa[4][ks], Wa[5][ks], Wa[6][ks], Wa[7][ks],
                  Wa[8][ks], Wa[9][ks], Wa[10][ks], Wa[11][ks]);
      } else if (ks < PKS_REG) {
        const int j = ks - PKS_A;
        LR_MFMA12(acc, a, "v", Wv[0][j], Wv[1][j], Wv[2][j], Wv[3][j], Wv[4][j], Wv[5][j], Wv[6][j], Wv[7][j],
                  Wv[8][j], Wv[9][j], Wv[10][j], Wv[11][j]);
      } else {
        bf16x8 wl[PNT];
#pragma unroll
        for (int tl = 0; tl < PNT; ++tl)
          wl[tl] = Wl[((wave * PNT + tl) * (PKS - PKS_REG) + (ks - PKS_REG)) * 64 + lane];
        LR_MFMA12(acc, a, "v", wl[0], wl[1], wl[2], wl[3], wl[4], wl[5], wl[6], wl[7], wl[8], wl[9], wl[10], wl[11]);
      }
    }
    // the asm MFMAs are opaque to the compiler's hazard recogniser: let the last ones retire before
    // any VALU instruction reads an accumulator
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    // (... and tie the accumulators to a statement behind the wait: their reads are plain register arithmetic,
    // which nothing else keeps behind it — see LR_ACC_READY in lr_rnn_cluster.hip)
#pragma unroll
    for (int tl = 0; tl < PNT; ++tl) asm volatile("" : "+v"(acc[tl]));
    // rows 4 kg + i of the result tile: the group's samples are rows 0..PBH-1, i.e. lanes with kg == 0
    if (kg == 0) {
#pragma unroll
      for (int tl = 0; tl < PNT; ++tl)
#pragma unroll
        for (int i = 0; i < PBH; ++i)
          S[((tl >> 2) * PBH + i) * PH + 64 * wave + 16 * (tl & 3) + col] = acc[tl][i];
    }
    // wave w produced units 64w .. 64w+63 and its own threads consume them: in-order LDS, no barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- gate math (fast exp / reciprocal forms: ~1e-6 relative, inside the bf16 operands' rounding) ----
#pragma unroll
    for (int k = 0; k < PBH; ++k) {
      const int b = b0 + k;
      const bool row_ok = b < B;
      const bool live = row_ok && t < len[k];
      const float hn = S[(2 * PBH + k) * PH + tid] + bhn;
      const float r = __frcp_rn(1.f + __expf(-(gx.v[0][k] + S[(0 * PBH + k) * PH + tid])));
      const float z = __frcp_rn(1.f + __expf(-(gx.v[1][k] + S[(1 * PBH + k) * PH + tid])));
      const float n = 2.f * __frcp_rn(1.f + __expf(-2.f * (gx.v[2][k] + r * hn))) - 1.f;
      fetch_gx(gx, tnext, k);
      const float h = live ? (1.f - z) * n + z * hreg[k] : 0.f;
      hreg[k] = h;
      hnxt[k * PHLD + tid] = f2bf(h);
      if (row_ok) {
        const int64_t bt = (int64_t)b * T + t;
        y[bt * (D * PH) + d * PH + tid] = h;
        extra[(bt * D + d) * PH + tid] = live ? hn : 0.f;
        if (live) {
          float* go = gates + (bt * D + d) * (3 * PH) + tid;
          go[0] = r;
          go[PH] = z;
          go[2 * PH] = n;
        }
      }
    }
    lr_lds_barrier();   // hnxt complete, S free again
  };
  for (int s = 0; s < T; s += 2) {
    step(s, gxA);
    if (s + 1 < T) step(s + 1, gxB);
  }
}


// ---------------------------------------------------------------------------------------------
// backward recurrence, one launch per layer pass
// ---------------------------------------------------------------------------------------------
// dh_t[j] = dy_t[j] + dh_{t'} z_{t'} + sum_kappa dGh_{t'}[kappa] W_hh[kappa][j]   (t' = the step
// processed just before, kappa = (gate, k) over 3*256), then the gate gradients of step t — the
// arithmetic of rnn_bwd_step_kernel<3> with the product's operands in bf16.  One workgroup per
// (sample, direction); wave w owns output units 64w .. 64w+63 as 4 column tiles x 24 k steps = 96
// weight fragments (k steps 0-14 in AGPRs, 15-23 in VGPRs, none in LDS); dGh of the previous step
// (768 bf16, row 0 of the A operand) goes from the gate phase to the product through LDS.
constexpr int BKS = 3 * PH / 32;   // 24 k steps
constexpr int BKS_A = 15, BKS_REG = 24;
constexpr int BGLD = 3 * PH + 8;   // bf16 per dGh buffer
constexpr size_t PBWD_LDS = (size_t)2 * BGLD * 2 + 16 + (size_t)4 * 4 * (BKS - BKS_REG) * 1024 + (size_t)PH * 4;

#define LR_MFMA4(acc, a, WC, w0, w1, w2, w3)                                                                  \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %2, %4, %7, %2\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %3, %4, %8, %3"                                                        \
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                         \
               : "v"(a), WC(w0), WC(w1), WC(w2), WC(w3))
#define LR_MFMA4_FIRST(acc, a, w0, w1, w2, w3)                                                                \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %1, %4, %6, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %2, %4, %7, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %3, %4, %8, 0"                                                         \
               : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3])                                     \
               : "v"(a), "a"(w0), "a"(w1), "a"(w2), "a"(w3))

// W_hh -> bf16 B fragments of the backward product: out[((d*4 + wave)*4 + nt)*BKS + ks][lane] holds
// W_hh[kappa = 32ks + 8kg + e][j = 64 wave + 16 nt + col], e = 0..7
__global__ void gru256_pack_whh_t_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                         bf16x8* __restrict__ out, int D) {
  const int total = D * 4 * 4 * BKS * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, ks = (i >> 6) % BKS, nt = (i / (64 * BKS)) & 3, wave = (i / (64 * BKS * 4)) & 3;
    const int d = i / (64 * BKS * 16);
    const int col = lane & 15, kg = lane >> 4;
    const float* src = (d ? w1 : w0) + (int64_t)(ks * 32 + kg * 8) * PH + 64 * wave + 16 * nt + col;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)src[(int64_t)e * PH];
    out[i] = v;
  }
}

__global__ __launch_bounds__(256, 1) void gru256_bwd_persist_kernel(
    const float* __restrict__ gates, const float* __restrict__ extra, const float* __restrict__ y,
    const float* __restrict__ dy, const float* __restrict__ dh_n, float* __restrict__ dG,
    const bf16x8* __restrict__ wpk, const int32_t* __restrict__ lens, int B, int T, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* gS = reinterpret_cast<bf16_t*>(smem);                                        // [2][BGLD] dGh (row 0)
  bf16x8* Wl = reinterpret_cast<bf16x8*>(smem + (size_t)2 * BGLD * 2 + 16);            // [4][4][BKS-BKS_REG][64]
  float* S = reinterpret_cast<float*>(smem + (size_t)2 * BGLD * 2 + 16 + (size_t)16 * (BKS - BKS_REG) * 1024);   // [PH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, d = blockIdx.y;
  const int col = lane & 15, kg = lane >> 4;
  const int DH = D * PH;

  bf16x8 Wa[4][BKS_A], Wv[4][BKS_REG - BKS_A];
  const bf16x8* wsrc = wpk + ((int64_t)(d * 4 + wave) * 4 * BKS) * 64 + lane;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int ks = 0; ks < BKS; ++ks) {
      const bf16x8 f = wsrc[(nt * BKS + ks) * 64];
      if (ks < BKS_A) Wa[nt][ks] = f;
      else if (ks < BKS_REG) Wv[nt][ks - BKS_A] = f;
      else Wl[((wave * 4 + nt) * (BKS - BKS_REG) + (ks - BKS_REG)) * 64 + lane] = f;
    }
  }
  for (int i = tid; i < 2 * BGLD; i += 256) gS[i] = 0;

  // gate phase: thread = hidden unit `tid` of sample b
  const int len = lens[b];
  const float inj = dh_n ? dh_n[((int64_t)d * B + b) * PH + tid] : 0.f;
  float car = 0.f;   // dh_{t'} * z_{t'}
  struct In { float dy, r, z, n, hn, hp; };
  In inA, inB;   // operands of even / odd steps: each set is fetched TWO steps ahead
  auto time_of = [&](int s) {   // time index of step s (clamped to the last step)
    const int sc = s < T ? s : T - 1;
    return d == 0 ? T - 1 - sc : sc;
  };
  auto fetch = [&](In& in, int t) {
    const int tp = d == 0 ? t - 1 : t + 1;
    const int64_t bt = (int64_t)b * T + t;
    in.dy = dy[bt * DH + d * PH + tid];
    const float* gi = gates + (bt * D + d) * (int64_t)(3 * PH) + tid;
    in.r = gi[0];
    in.z = gi[PH];
    in.n = gi[2 * PH];
    in.hn = extra[(bt * D + d) * PH + tid];
    in.hp = (tp >= 0 && tp < T) ? y[((int64_t)b * T + tp) * DH + d * PH + tid] : 0.f;
  };
  fetch(inA, time_of(0));
  fetch(inB, time_of(1));
  __syncthreads();

  auto step = [&](int s, In& in) {
    const int t = time_of(s);
    const int tnext = time_of(s + 2);
    const bf16_t* gcur = gS + (s & 1) * BGLD;      // dGh of the step processed before this one
    bf16_t* gnxt = gS + ((s + 1) & 1) * BGLD;
    // ---- product: rows other than 0 of the A operand are zero ----------------------------------
    f32x4 acc0[4], acc1[4];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto afrag = [&](int ks) -> bf16x8 {
      bf16x8 v = zero8;
      if (col == 0) v = *reinterpret_cast<const bf16x8*>(gcur + ks * 32 + kg * 8);
      return v;
    };
    // a k step is only 4 MFMAs (64 cycles), less than an LDS round trip: keep the A fragments of the
    // next three k steps in flight
    bf16x8 aring[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) aring[i] = afrag(i);
#pragma unroll
    for (int ks = 0; ks < BKS; ++ks) {
      const bf16x8 a = aring[ks & 3];
      if (ks + 3 < BKS) aring[(ks + 3) & 3] = afrag(ks + 3);
      // two accumulator sets (even / odd k steps): dependent MFMAs are 8 issues apart
      if (ks == 0) {
        LR_MFMA4_FIRST(acc0, a, Wa[0][0], Wa[1][0], Wa[2][0], Wa[3][0]);
      } else if (ks == 1) {
        LR_MFMA4_FIRST(acc1, a, Wa[0][1], Wa[1][1], Wa[2][1], Wa[3][1]);
      } else if (ks < BKS_A) {
        if (ks & 1) LR_MFMA4(acc1, a, "a", Wa[0][ks], Wa[1][ks], Wa[2][ks], Wa[3][ks]);
        else LR_MFMA4(acc0, a, "a", Wa[0][ks], Wa[1][ks], Wa[2][ks], Wa[3][ks]);
      } else if (ks < BKS_REG) {
        const int j = ks - BKS_A;
        if (ks & 1) LR_MFMA4(acc1, a, "v", Wv[0][j], Wv[1][j], Wv[2][j], Wv[3][j]);
        else LR_MFMA4(acc0, a, "v", Wv[0][j], Wv[1][j], Wv[2][j], Wv[3][j]);
      } else {
        bf16x8 wl[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wl[nt] = Wl[((wave * 4 + nt) * (BKS - BKS_REG) + (ks - BKS_REG)) * 64 + lane];
        if (ks & 1) LR_MFMA4(acc1, a, "v", wl[0], wl[1], wl[2], wl[3]);
        else LR_MFMA4(acc0, a, "v", wl[0], wl[1], wl[2], wl[3]);
      }
    }
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {   // (reads stay behind the wait: see the forward kernel)
      asm volatile("" : "+v"(acc0[nt]));
      asm volatile("" : "+v"(acc1[nt]));
    }
    if (kg == 0) {   // row 0 = the sample: lanes 0..15, register 0
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) S[64 * wave + 16 * nt + col] = acc0[nt][0] + acc1[nt][0];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local exchange: units 64w .. 64w+63
    // ---- gate gradients of step t -------------------------------------------------------------------
    {
      const float r = in.r, z = in.z, n = in.n, hn = in.hn, hp = in.hp;
      float dh = in.dy + S[tid] + car;
      const bool is_last = d == 0 ? (t == len - 1) : (t == 0);
      if (is_last) dh += inj;
      fetch(in, tnext);
      float dr_pre = 0.f, dz_pre = 0.f, dn_pre = 0.f, dnr = 0.f;
      car = 0.f;
      if (t < len) {
        dn_pre = dh * (1.f - z) * (1.f - n * n);
        dr_pre = dn_pre * hn * r * (1.f - r);
        dz_pre = dh * (hp - n) * z * (1.f - z);
        dnr = dn_pre * r;
        car = dh * z;
      }
      float* dgo = dG + (((int64_t)b * T + t) * D + d) * (int64_t)(4 * PH) + tid;
      dgo[0] = dr_pre;
      dgo[PH] = dz_pre;
      dgo[2 * PH] = dn_pre;
      dgo[3 * PH] = dnr;
      gnxt[tid] = f2bf(dr_pre);
      gnxt[PH + tid] = f2bf(dz_pre);
      gnxt[2 * PH + tid] = f2bf(dnr);
    }
    lr_lds_barrier();   // gnxt complete
  };
  for (int s = 0; s < T; s += 2) {
    step(s, inA);
    if (s + 1 < T) step(s + 1, inB);
  }
}

}  // namespace

int lr_gru256_persist_supported(int G, int B, int H) { return G == 3 && H == PH && B >= 1 ? 1 : 0; }

// bytes of packed-weight workspace lr_gru256_persist_forward needs (bf16 fragments of every direction)
size_t lr_gru256_persist_pack_bytes(int D) { return (size_t)D * 4 * PNT * PKS * 64 * sizeof(bf16x8); }

int lr_gru256_persist_forward(float* gates, float* extra, float* y, const float* const* w_hh, const float* const* b_hh,
                              const int32_t* lens, void* wpack, int B, int T, int D, hipStream_t stream) {
  static bool attr_set = false;
  lr_clear_error();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gru256_fwd_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)PFWD_LDS) != hipSuccess)
      return LR_ERR_LAUNCH;
    attr_set = true;
  }
  LR_LAUNCH(gru256_pack_whh_kernel, dim3(192), dim3(256), 0, stream, w_hh[0], w_hh[D - 1], (bf16x8*)wpack, D);
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  const dim3 grid((B + PBH - 1) / PBH, D);
  hipEvent_t e0, e1;
  if (lr_prof_next(LR_PROF_RNN_FWD, &e0, &e1))
    hipExtLaunchKernelGGL(gru256_fwd_persist_kernel, grid, dim3(256), PFWD_LDS, stream, e0, e1, 0, gates, extra, y,
                          (const bf16x8*)wpack, b_hh[0], b_hh[D - 1], lens, B, T, D);
  else
    hipLaunchKernelGGL(gru256_fwd_persist_kernel, grid, dim3(256), PFWD_LDS, stream, gates, extra, y,
                       (const bf16x8*)wpack, b_hh[0], b_hh[D - 1], lens, B, T, D);
  return lr_launch_status();
}

size_t lr_gru256_persist_bwd_pack_bytes(int D) { return (size_t)D * 16 * BKS * 64 * sizeof(bf16x8); }

int lr_gru256_persist_backward(const float* gates, const float* extra, const float* y, const float* dy,
                               const float* dh_n, float* dG, const float* const* w_hh, const int32_t* lens,
                               void* wpack, int B, int T, int D, hipStream_t stream) {
  static bool attr_set = false;
  lr_clear_error();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gru256_bwd_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)PBWD_LDS) != hipSuccess)
      return LR_ERR_LAUNCH;
    attr_set = true;
  }
  LR_LAUNCH(gru256_pack_whh_t_kernel, dim3(192), dim3(256), 0, stream, w_hh[0], w_hh[D - 1], (bf16x8*)wpack, D);
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  const dim3 grid(B, D);
  hipEvent_t e0, e1;
  if (lr_prof_next(LR_PROF_RNN_BWD, &e0, &e1))
    hipExtLaunchKernelGGL(gru256_bwd_persist_kernel, grid, dim3(256), PBWD_LDS, stream, e0, e1, 0, gates, extra, y, dy,
                          dh_n, dG, (const bf16x8*)wpack, lens, B, T, D);
  else
    hipLaunchKernelGGL(gru256_bwd_persist_kernel, grid, dim3(256), PBWD_LDS, stream, gates, extra, y, dy, dh_n, dG,
                       (const bf16x8*)wpack, lens, B, T, D);
  return lr_launch_status();
}
