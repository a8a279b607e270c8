// lr_rnn_persist.hip — the GRU-256 recurrence as ONE launch per layer pass (pixel regime).
//
// The step-per-launch recurrence of lr_rnn.hip is bound by launch boundaries and memory round trips
// (~6 us per step whatever the arithmetic), because fp32 W_hh (768 KB per direction) has to be
// re-streamed from the fabric every step.  BASELINE.json's configs[1] (the pixel regime: conv
// frontend + BiGRU-256, "bf16") allows a different trade: with W_hh rounded to bf16 (384 KB per
// direction) the whole matrix fits ONE compute unit — 288 KB as MFMA operand fragments held in
// the registers of 4 waves, 96 KB in LDS — and batch rows are independent, so a direction's 32
// samples split into two groups of 16 that never communicate.  A layer pass is then 2 x D
// workgroups that each run all T steps with nothing but a workgroup barrier per step:
//   state (16 x 256, bf16, double-buffered in LDS) x W_hh^T on v_mfma_f32_16x16x32_bf16 (96 per wave
//   per step), fp32 accumulation, fp32 gate math and fp32 carried state in registers.
// The interface buffers are those of the step kernels (gates in/out, extra, y, dG: fp32), so the
// one-shot GEMMs around the recurrence (input projection, weight gradients) are unchanged, and so
// is the reference-faithful fp32 path, which never comes here.  No reference counterpart beyond
// better_model.py:74 (nn.GRU); precision: bf16 operands in the recurrent product only.
//
// STATUS: opt-in (VideoEncoder.recurrence = 'bf16'), correct (tests/test_gpu_encoder.py), measured
// on MI355X at B = 32, T = 75: 4.0 us per step (a layer pass 301 us) against ~5.3-6.0 us for the step
// kernels — the pixel step only moves from 5.32 to ~5.2 ms with both forward passes on it, because a
// workgroup's product phase is ~2.3 us (96 v_mfma_f32_16x16x32_bf16 with half of their B operands
// staged out of AGPRs, ~57 cycles apiece) and the gate phase + interface traffic another ~1.7 us.
// Sample groups of 4 (not 16) per workgroup keep the gate phase and the per-CU interface traffic
// small: the first cut with 16 samples per workgroup ran 6.7 us per step.  Not enabled by default:
// a matching persistent backward would be needed for a ~6 % gain, at bf16 recurrent precision.
#include "lr_common.h"
#include <hip/hip_ext.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

constexpr int PH = 256;            // hidden size this kernel is built for
constexpr int PKS = PH / 32;       // k steps of 32 (forward: k = previous state)
constexpr int PKS_REG = 6;         // forward k steps whose weight fragments stay in registers
constexpr int PNT = 12;            // 16-unit column tiles per wave: 3 gates x 4
constexpr int PHLD = PH + 8;       // bf16 per LDS row of the state (528 B: conflict-free b128 rows)
constexpr int PBH = 4;             // samples per workgroup (rows >= PBH of the 16-row MFMA operand are zero)
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

// grid (sample groups of PBH, directions); 256 threads.  Product phase: wave w owns hidden units
// [64w, 64w + 64) as column tiles tl = gate * 4 + nt (units 64w + 16nt .. +15 of that gate); its
// results (rows = the group's samples) go through LDS so that the gate phase runs one hidden unit
// per THREAD over the group's samples, with every global access a contiguous 1 KB per wave.
__global__ __launch_bounds__(256, 1) void gru256_fwd_persist_kernel(float* __restrict__ gates,
                                                                    float* __restrict__ extra,
                                                                    float* __restrict__ y, const float* __restrict__ w0,
                                                                    const float* __restrict__ w1,
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
  const float* W = d ? w1 : w0;
  const float* bhh = d ? bhh1 : bhh0;
  const int col = lane & 15, kg = lane >> 4;

  // ---- W_hh -> bf16 MFMA B fragments: lane (column = unit, k group) holds 8 consecutive k ---------
  bf16x8 Wr[PNT][PKS_REG];
#pragma unroll
  for (int tl = 0; tl < PNT; ++tl) {
    const int g = tl >> 2, nt = tl & 3;
    const float* row = W + ((int64_t)g * PH + 64 * wave + 16 * nt + col) * PH + kg * 8;
#pragma unroll
    for (int ks = 0; ks < PKS; ++ks) {
      const float4 lo = *reinterpret_cast<const float4*>(row + ks * 32);
      const float4 hi = *reinterpret_cast<const float4*>(row + ks * 32 + 4);
      const bf16x8 f = pack8(lo, hi);
      if (ks < PKS_REG) Wr[tl][ks] = f;
      else Wl[((wave * PNT + tl) * (PKS - PKS_REG) + (ks - PKS_REG)) * 64 + lane] = f;
    }
  }
  for (int i = tid; i < 2 * 16 * PHLD; i += 256) hS[i] = 0;

  // gate phase: thread = hidden unit `tid`, loop over the group's samples
  const float bhn = bhh[2 * PH + tid];
  int len[PBH];
  float hreg[PBH], gx[3][PBH];
#pragma unroll
  for (int k = 0; k < PBH; ++k) {
    len[k] = b0 + k < B ? lens[b0 + k] : 0;
    hreg[k] = 0.f;
  }
  // input-projection pre-activations, fetched one step ahead
  auto fetch_gx = [&](int t, int k) {
    const int b = b0 + k < B ? b0 + k : 0;
    const float* gp = gates + (((int64_t)b * T + t) * D + d) * (3 * PH) + tid;
    gx[0][k] = gp[0];
    gx[1][k] = gp[PH];
    gx[2][k] = gp[2 * PH];
  };
#pragma unroll
  for (int k = 0; k < PBH; ++k) fetch_gx(d == 0 ? 0 : T - 1, k);
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? s : T - 1 - s;
    const int tnext = d == 0 ? (s + 1 < T ? s + 1 : s) : (s + 1 < T ? T - 2 - s : 0);
    const bf16_t* hcur = hS + (s & 1) * 16 * PHLD;
    bf16_t* hnxt = hS + ((s + 1) & 1) * 16 * PHLD;
    f32x4 acc[PNT];
#pragma unroll
    for (int tl = 0; tl < PNT; ++tl) acc[tl] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a_next = *reinterpret_cast<const bf16x8*>(hcur + col * PHLD + kg * 8);   // row = sample
#pragma unroll
    for (int ks = 0; ks < PKS; ++ks) {
      const bf16x8 a = a_next;   // the next k step's state fragment is read while this one multiplies
      if (ks + 1 < PKS) a_next = *reinterpret_cast<const bf16x8*>(hcur + col * PHLD + (ks + 1) * 32 + kg * 8);
#pragma unroll
      for (int tl = 0; tl < PNT; ++tl) {
        const bf16x8 w = ks < PKS_REG ? Wr[tl][ks < PKS_REG ? ks : 0]
                                      : Wl[((wave * PNT + tl) * (PKS - PKS_REG) + (ks - PKS_REG)) * 64 + lane];
        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w, acc[tl], 0, 0, 0);
      }
    }
    // rows 4 kg + i of the result tile: the group's samples are rows 0..PBH-1, i.e. lanes with kg == 0
    if (kg == 0) {
#pragma unroll
      for (int tl = 0; tl < PNT; ++tl)
#pragma unroll
        for (int i = 0; i < PBH; ++i)
          S[((tl >> 2) * PBH + i) * PH + 64 * wave + 16 * (tl & 3) + col] = acc[tl][i];
    }
    __syncthreads();
    // ---- gate math (fast exp / reciprocal forms: ~1e-6 relative, inside the bf16 operands' rounding) ----
#pragma unroll
    for (int k = 0; k < PBH; ++k) {
      const int b = b0 + k;
      const bool row_ok = b < B;
      const bool live = row_ok && t < len[k];
      const float hn = S[(2 * PBH + k) * PH + tid] + bhn;
      const float r = __frcp_rn(1.f + __expf(-(gx[0][k] + S[(0 * PBH + k) * PH + tid])));
      const float z = __frcp_rn(1.f + __expf(-(gx[1][k] + S[(1 * PBH + k) * PH + tid])));
      const float n = 2.f * __frcp_rn(1.f + __expf(-2.f * (gx[2][k] + r * hn))) - 1.f;
      fetch_gx(tnext, k);
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
    __syncthreads();   // hnxt complete, S free again
  }
}

}  // namespace

int lr_gru256_persist_supported(int G, int B, int H) { return G == 3 && H == PH && B >= 1 && B <= 64 ? 1 : 0; }

int lr_gru256_persist_forward(float* gates, float* extra, float* y, const float* const* w_hh, const float* const* b_hh,
                              const int32_t* lens, int B, int T, int D, hipStream_t stream) {
  static bool attr_set = false;
  lr_clear_error();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gru256_fwd_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)PFWD_LDS) != hipSuccess)
      return LR_ERR_LAUNCH;
    attr_set = true;
  }
  const dim3 grid((B + PBH - 1) / PBH, D);
  hipEvent_t e0, e1;
  if (lr_prof_next(LR_PROF_RNN_FWD, &e0, &e1))
    hipExtLaunchKernelGGL(gru256_fwd_persist_kernel, grid, dim3(256), PFWD_LDS, stream, e0, e1, 0, gates, extra, y,
                          w_hh[0], w_hh[D - 1], b_hh[0], b_hh[D - 1], lens, B, T, D);
  else
    hipLaunchKernelGGL(gru256_fwd_persist_kernel, grid, dim3(256), PFWD_LDS, stream, gates, extra, y, w_hh[0],
                       w_hh[D - 1], b_hh[0], b_hh[D - 1], lens, B, T, D);
  return lr_launch_status();
}
