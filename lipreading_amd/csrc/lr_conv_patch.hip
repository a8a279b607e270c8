// gfx950 HIP kernels of the build-defined conv frontend: the patch-resident forward / data-gradient kernel of the
// 24-wide stride-1 (3,5,5) layer (layer 2 of the frontend), in a translation unit of its own (its pinned schedules
// take minutes to compile).  No reference file: the reference has no conv frontend (regime X).
#include "lr_common.h"
#include "lr_conv_dev.h"
#include <hip/hip_ext.h>

namespace {

// ---------------------------------------------------------------------------------------------
// patch-resident forward / data-gradient kernel for the 24-wide stride-1 (3,5,5) layer
// ---------------------------------------------------------------------------------------------
// The implicit-GEMM kernel above re-gathers every input element once per tap it feeds (75x for this
// layer): 6.6 GB of L2 -> LDS traffic and ~30 VALU operations per 16-byte unit.  Here a workgroup
// loads the input patch of its output tile ONCE (4 frames x 8 rows x 24 columns of outputs, i.e.
// 6 x 12 x 28 input positions of 32 channels = 126 KB of LDS) and then runs all 75 taps out of LDS:
// an A fragment (32 positions x 16 channels of one tap) is one ds_read_b128 per lane at
// patch[position + tap offset]; B fragments (32 output channels x 16 channels of one tap) come
// straight from global memory in the fragment-major packing above, 1 KB contiguous per wave load,
// prefetched one tap ahead.  Inputs with 64 channels (the data gradient: dZ has 64) take two passes
// of 32 channels with the accumulators kept.
//   Wave w owns output frame f0 + w: 6 MFMA row tiles of 8 rows x 4 columns x NT column tiles ->
//   6*NT accumulators; taps reaching across a clip boundary are skipped (wave-uniform).
//   LDS layout: patch row (slot*12 + row) starts at byte (slot*12 + row) * P2_RS, P2_RS = 28 * 64 (a multiple of the
//   256-byte bank line); position col of it holds its 32 channels as 4 chunks of 16 bytes, chunk c stored at
//   c ^ (row & 3).  ds_read_b128 serves a wave in four FIXED lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19,
//   28-31} and the same + 32): with the 8 x 4 pixel block below each group reads two columns of four rows whose
//   indices are distinct mod 4, twice, so the row-only swizzle puts its 16 lanes on 16 different 16-byte slots
//   for every tap shift (checked exhaustively).  Unlike round 1's swizzle on the position index, the column
//   part of a tap shift is a plain byte offset: the reads of a row of five taps share TWO address registers (one
//   per k chunk; they differ in bit 5) and the instruction's immediate does the rest.  (A padded row stride
//   without swizzle — tried first — is 2-way conflicted for these lane groups whatever the padding:
//   SQ_LDS_BANK_CONFLICT went from 0 to 83 % of the LDS cycles.)
// (Measured and dropped, round 2: half-WIDTH tiles — 4 frames x 8 rows x 12 columns, 73 KB patch, 3 x NT
// accumulators, TWO workgroups per CU so that one's MFMAs cover the other's patch fill and epilogue — are bit-
// identical but 7 % (forward) / 19 % (data gradient) SLOWER: a weight fragment then feeds 3 MFMAs instead of 6
// and the doubled fragment traffic from L1/L2 costs more than the overlap returns.)
constexpr int P2_TT = 4, P2_TH = 8, P2_W = 24, P2_PW = P2_W + 4, P2_PH = P2_TH + 4, P2_SLOTS = P2_TT + 2;
constexpr int P2_RS = P2_PW * 64;                  // bytes per patch row: 28 positions x 64 B = 7 bank lines
constexpr int P2_LDS = P2_SLOTS * P2_PH * P2_RS;   // 129,024 bytes
constexpr int P2_STAGE = 6144;                     // per-wave output staging area behind the patch (epilogue)
constexpr int P2_BDIST = 2;                        // taps between a B fragment's load and its use

constexpr int kChipCUs = 256;                      // one workgroup of this kernel per CU

// -DLRP_TIMING (tools/build_variant.sh; never in the product build): thread 0 of every workgroup adds up the shader
// clock between the phase boundaries of its tile: [0] fill (until the barrier behind it), [1] taps, [2] epilogue, [3] whole
// tile, [4] tiles; tools/probes/conv_patch_timing.py reads them back through lr_conv_patch_debug_times
#ifdef LRP_TIMING
__device__ long long g_lrp_times[4096][8];
#define LRP_T0() long long lrp_t_ = clock64(), lrp_t0_ = lrp_t_
#define LRP_T(k) do { const long long n_ = clock64(); if (threadIdx.x == 0) g_lrp_times[blockIdx.x & 4095][k] += n_ - lrp_t_; lrp_t_ = n_; } while (0)
#define LRP_TEND() do { if (threadIdx.x == 0) { g_lrp_times[blockIdx.x & 4095][3] += clock64() - lrp_t0_; g_lrp_times[blockIdx.x & 4095][4] += 1; } } while (0)
#else
#define LRP_T0() do { } while (0)
#define LRP_T(k) do { } while (0)
#define LRP_TEND() do { } while (0)
#endif

// One tile.  SPLIT = false: the 4-frame tile described above, wave w = frame f0 + w with all 6 x NT accumulators.
// SPLIT = true (the launch's LAST tiles, see conv_patch_kernel): a tile of FT = 1 (NT = 2) or 2 (NT = 1) frames
// whose accumulator tiles are dealt to the four waves — three row tiles x one column tile each — so that it takes
// 1/4 (1/2) of a whole tile's time; same patch layout, same per-accumulator MFMA order (results are bit-identical).
// UNPOOL (data gradient of a layer whose forward fused ReLU + MaxPool): X is the POOLED gradient dP [F][H/2][12][C]
// and `code` the windows' codes (relu_pool4); the fill rebuilds the full-resolution dZ patch on the way into LDS —
// a thread loads 16 bytes of dP and 8 codes and stores the window's four 16-byte units (unpool8) — so dZ is never
// written to or read from memory (round 3: a 114 us HBM-bound un-pooling kernel in front of this one, 177 MB written
// and read back, 75 % of it zeros), and the fill loads 3/8 of the bytes with 14 load instructions per thread, not 36.
template <int CG, int NT, bool POOL, bool SPLIT, bool UNPOOL>
__device__ __forceinline__ void conv_patch_tile(unsigned char* __restrict__ patch, const bf16_t* __restrict__ X,
                                                const bf16_t* __restrict__ Wf, const float* __restrict__ bias,
                                                bf16_t* __restrict__ Y, unsigned char* __restrict__ code, int F, int T,
                                                int H, int relu, int f0, int h0) {
  constexpr int C = 32 * CG, N = 32 * NT, TAPS = 75;
  constexpr int FT = SPLIT ? (NT == 2 ? 1 : 2) : P2_TT;   // frames of this tile
  constexpr int MTW = SPLIT ? 3 : 6, NTW = SPLIT ? 1 : NT;   // accumulator tiles of one wave: MTW x NTW
  constexpr int NPASS = (FT + 2) * (P2_PH / 2);           // patch fill passes (two patch rows each)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, kg = lane >> 5;
  // frame of the tile, first row tile (in threes) and column tile this wave computes
  const int fi = !SPLIT ? wave : (NT == 2 ? 0 : wave >> 1);
  const int g0 = !SPLIT ? 0 : (NT == 2 ? wave >> 1 : wave & 1);
  const int j0 = !SPLIT ? 0 : (NT == 2 ? wave & 1 : 0);
  const int f = f0 + fi;
  const bool fvalid = f < F;
  const int t = f % T;
  // temporal taps reaching across the clip boundary are skipped: this wave runs tap rows
  // (dt, dh) = row0 .. row0 + nrow - 1, whole temporal offsets only (wave-uniform, in SGPRs)
  const int dt_lo = t == 0 ? 1 : 0, dt_hi = t == T - 1 ? 1 : 2;
  const int row0 = 5 * dt_lo;
  const int nrow = fvalid ? 5 * (dt_hi - dt_lo + 1) : 0;

  f32x16 acc[MTW][NTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // A row i = lane & 31 of an m-tile is pixel (h, w) = (b1 + 2 b3 + 4 b4, b0 + 2 b2) of the 8 x 4 block
  // (b_k = bit k of i): the four C/D rows r & 3 a lane holds per register group r >> 2 are then the
  // 2 x 2 pooling window (h = 2(r>>2) + {0,1}, w = 2 kg + {0,1}) in scan order, so the pooled
  // epilogue is lane-local.  Lanes that ds_read_b128 serves together still touch 16 different 16-byte
  // slots of a 256-byte bank line: same-w lanes of a group sit on rows {0,1,6,7} or {2,3,4,5},
  // 3*row mod 4 all different.
  const int a_h = ((lr >> 1) & 1) + 2 * (lr >> 3), a_w = (lr & 1) + 2 * ((lr >> 2) & 1);
  // byte offset of this lane's pixel in its first row tile (chunk bits added per tap row)
  const int base_b = (fi * P2_PH + a_h) * P2_RS + a_w * 64 + g0 * (3 * 256);
  if constexpr (UNPOOL) {
    // the halo columns of every patch row are zeros for every channel group: written once (the fill below only
    // touches positions 2 .. 25)
    for (int u = tid; u < (FT + 2) * P2_PH * 16; u += 256) {
      const int q = (u >> 2) & 3;
      *reinterpret_cast<uint4*>(patch + (u >> 4) * P2_RS + (q < 2 ? q : P2_W + q) * 64 + (u & 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  LRP_T0();
  for (int cg = 0; cg < CG; ++cg) {
    if (cg > 0) __syncthreads();   // every wave is done with the previous channel group's patch
    LRP_T(1);
    // ---- load the patch -------------------------------------------------------------------------
    // 224 threads cover two patch rows (28 positions x 4 sixteen-byte chunks each) per pass, 36
    // passes: slot and row of a pass are compile-time, so a unit costs a handful of VALU operations
    // (decoding a flat unit index cost ~60 and a third of the kernel's VALU work).  Batches of 18
    // loads, each issued completely before its first store.
    // (One batch of 36 — every load of a fill in flight before the first store — measured the same forward and 1 %
    // slower in the data gradient, round 3: the fill runs at the rate of the CU's outstanding misses either way.)
    if constexpr (UNPOOL) {
      // pooled units of the patch: (FT + 2) slots x 6 pooled rows x 12 pooled columns x 4 chunks of 8 channels, dealt
      // flat over the 256 threads (unit u = tid + 256 i: row = u / 48 counts slots and pooled rows together, so the
      // two patch rows of a window are 2 row, 2 row + 1); the halo columns (positions 0, 1, 26, 27) were zeroed once
      constexpr int UROWS = (FT + 2) * (P2_PH / 2), UNITS = UROWS * 48, UI = (UNITS + 255) / 256;
      const int Hp = H >> 1, hp0 = (h0 >> 1) - 1;
      uint4 dv[UI];
      uint2 cv[UI];
      int row = tid / 48, rem = tid - 48 * row;
      int orow[UI], orem[UI];
#pragma unroll
      for (int i = 0; i < UI; ++i) {
        orow[i] = row;
        orem[i] = rem;
        const int s = (row * 43) >> 8, pr = row - 6 * s;   // slot, pooled row of the slot (row < 48)
        const int ff = f0 - 1 + s, hp = hp0 + pr;
        dv[i] = make_uint4(0u, 0u, 0u, 0u);
        cv[i] = make_uint2(0u, 0u);
        if (row < UROWS && ff >= 0 && ff < F && hp >= 0 && hp < Hp) {
          const int64_t pi = (((int64_t)ff * Hp + hp) * (P2_W / 2) + (rem >> 2)) * C + cg * 32 + (rem & 3) * 8;
          dv[i] = *reinterpret_cast<const uint4*>(X + pi);
          cv[i] = *reinterpret_cast<const uint2*>(code + pi);
        }
        rem += 256 % 48;
        row += 256 / 48;
        if (rem >= 48) {
          rem -= 48;
          row += 1;
        }
      }
#pragma unroll
      for (int i = 0; i < UI; ++i) {
        if (orow[i] < UROWS) {
          uint4 o[4];
          unpool8(dv[i], cv[i], o);
          const int c = orem[i] & 3, sw = (c ^ (2 * (orow[i] & 1))) << 4;   // swizzle key of patch row 2 row (+ 1: ^ 16)
          unsigned char* dst = patch + (2 * orow[i]) * P2_RS + (2 * (orem[i] >> 2) + 2) * 64;
          *reinterpret_cast<uint4*>(dst + sw) = o[0];
          *reinterpret_cast<uint4*>(dst + 64 + sw) = o[1];
          *reinterpret_cast<uint4*>(dst + P2_RS + (sw ^ 16)) = o[2];
          *reinterpret_cast<uint4*>(dst + P2_RS + 64 + (sw ^ 16)) = o[3];
        }
      }
    } else {
      const int rp = tid >= 112 ? 1 : 0, un = tid - 112 * rp;   // row of the pair, unit in the row
      const int pw = un >> 2, c = un & 3;
      const bool tvalid = tid < 224 && pw >= 2 && pw < P2_W + 2;
      const bf16_t* xt = X + ((int64_t)(pw - 2)) * C + cg * 32 + c * 8;
#pragma unroll
      for (int half = 0; half < (NPASS + 17) / 18; ++half) {
        uint4 v[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) {
          const int k = half * 18 + i;            // pass: patch rows 2k, 2k+1
          if (k < NPASS) {
            const int s = k / 6, ph = 2 * (k % 6) + rp;
            const int ff = f0 - 1 + s, hh = h0 - 2 + ph;
            v[i] = make_uint4(0u, 0u, 0u, 0u);
            if (tvalid && ff >= 0 && ff < F && hh >= 0 && hh < H)
              v[i] = *reinterpret_cast<const uint4*>(xt + ((int64_t)ff * H + hh) * (P2_W * C));
          }
        }
        if (tid < 224) {
#pragma unroll
          for (int i = 0; i < 18; ++i) {
            const int k = half * 18 + i;
            if (k < NPASS)
              *reinterpret_cast<uint4*>(patch + (2 * k + rp) * P2_RS + pw * 64 + ((c ^ ((2 * k + rp) & 3)) << 4)) = v[i];
          }
        }
      }
    }
    __syncthreads();
    LRP_T(0);
    // ---- the taps out of LDS ---------------------------------------------------------------------
    // One wave per SIMD: nothing hides a latency unless the code does.  The loop body is one row of
    // five taps (dt, dh fixed; 25 taps per temporal offset, so no remainder and no branch), software
    // pipelined with static register indices: A fragments are double-buffered per HALF tap (m-tiles
    // 0-2 / 3-5: one half's reads fly during the other half's 6*NT MFMAs), B fragments sit in a ring
    // of five taps and are loaded P2_BDIST taps ahead of their use.
    if (nrow > 0) {
      const bf16_t* wfb = Wf + ((int64_t)cg * TAPS + 5 * row0) * (2 * NT) * 512 + j0 * 512 + lane * 8;
      const int last = 5 * nrow - 1;
      bf16x8 bq[5][2][NTW];
      bf16x8 a0[6], a1[6];
      auto load_b = [&](bf16x8 (&bb)[2][NTW], int i) {
        const int ii = i < last ? i : last;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            bb[kc][j] = *reinterpret_cast<const bf16x8*>(wfb + ((int64_t)ii * 2 * NT + kc * NT + j) * 512);
      };
      // xb = lane base + row offset + swizzled chunk of k chunk 0 (bytes; k chunk 1 is xb ^ 32), imm = 64 dw
      auto load_a = [&](bf16x8 (&aa)[6], int xb, int imm, int grp) {
        const int xb1 = xb ^ 32;
#pragma unroll
        for (int w3 = 0; w3 < 3; ++w3)
#pragma unroll
          for (int kc = 0; kc < 2; ++kc)
            aa[w3 * 2 + kc] = *reinterpret_cast<const bf16x8*>(patch + (kc ? xb1 : xb) + (imm + (3 * grp + w3) * 256));
      };
      // k chunk outermost: consecutive MFMAs never share an accumulator (with NT = 1 the w3-outer order issued
      // the two k chunks of a tile back to back, each waiting for the other's result)
      auto run = [&](const bf16x8 (&aa)[6], const bf16x8 (&bb)[2][NTW], int grp) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
          for (int w3 = 0; w3 < 3; ++w3)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
              acc[3 * grp + w3][j] =
                  __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[w3 * 2 + kc], bb[kc][j], acc[3 * grp + w3][j], 0, 0, 0);
      };
      // byte offset of tap row r (r = dt*5 + dh) + this lane's chunk of k chunk 0 in that patch row: the patch row
      // is slot*12 + a_h + dh, so its swizzle key is (a_h + dh) & 3
      auto row_off = [&](int r) {
        const int rr = row0 + (r < nrow ? r : nrow - 1);
        const int dh = rr % 5;
        return ((rr / 5) * P2_PH + dh) * P2_RS + ((kg ^ ((a_h + dh) & 3)) << 4);
      };
#pragma unroll
      for (int d = 0; d < P2_BDIST; ++d) {
        load_b(bq[d], d);
        // Keep the prologue's fragment loads in PROGRAM order.  Left alone hipcc issues them youngest tap first, so on
        // the loop's entry path the first tap's first fragment is the LAST load in the queue — and the wait in front of
        // the row's first MFMA, merged over both paths into the loop header, becomes vmcnt(1) on EVERY row: it then also
        // waits for the fragments of the row's second tap (the taps of a row after the second have their proper
        // vmcnt(6..8)).  MEASURED (round 4, same box): data gradient 345.4 -> 334.7 us, forward unchanged (316.7 / 316.6).
        __builtin_amdgcn_sched_barrier(0);
      }
      load_a(a0, base_b + row_off(0), 0, 0);
      if constexpr (!SPLIT) {
#pragma unroll 1
        for (int r = 0; r < nrow; ++r) {
          const int x0 = base_b + row_off(r), xn = base_b + row_off(r + 1);
#pragma unroll
          for (int dw = 0; dw < 5; ++dw) {
            load_b(bq[(dw + P2_BDIST) % 5], 5 * r + dw + P2_BDIST);
            load_a(a1, x0, 64 * dw, 1);
            run(a0, bq[dw], 0);
            if (dw < 4) load_a(a0, x0, 64 * (dw + 1), 0);
            else load_a(a0, xn, 0, 0);
            run(a1, bq[dw], 1);
          }
          // The machine scheduler would sink every load next to its use (lowest register pressure);
          // pin the interleave instead: 12 x {1 LDS read, NT MFMAs} per tap, with the tap's 2 NT weight-fragment
          // loads ONE AT A TIME in front of every (6 / NT)-th of them.  (Round 2 issued them back to back at the
          // start of the tap.  The four waves run in step and load the same fragments: 16 KB at once through the
          // CU's 64 B/clk vector memory path, 256 cycles during which a wave that is waiting for its load to be
          // taken issues no MFMA — the weight-gradient kernel lost half its MFMA rate in exactly this way until its
          // loads were spread out, lr_conv_wgrad.hip.)
#pragma unroll
          for (int st = 0; st < 10; ++st) {
#pragma unroll
            for (int g = 0; g < 6; ++g) {
              if (((st & 1) * 6 + g) % (6 / NT) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // DS read
              __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);   // MFMA
            }
          }
        }
      } else {
        // three row tiles per wave: the whole tap's A fragments are double-buffered, tap by tap (five taps per
        // row: the buffers swap roles once per row, a register copy)
#pragma unroll 1
        for (int r = 0; r < nrow; ++r) {
          const int x0 = base_b + row_off(r), xn = base_b + row_off(r + 1);
#pragma unroll
          for (int dw = 0; dw < 5; ++dw) {
            load_b(bq[(dw + P2_BDIST) % 5], 5 * r + dw + P2_BDIST);
            if (dw & 1) {
              load_a(a0, x0, 64 * (dw + 1), 0);
              run(a1, bq[dw], 0);
            } else {
              if (dw < 4) load_a(a1, x0, 64 * (dw + 1), 0);
              else load_a(a1, xn, 0, 0);
              run(a0, bq[dw], 0);
            }
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) a0[i] = a1[i];
#pragma unroll
          for (int st = 0; st < 5; ++st) {
#pragma unroll
            for (int g = 0; g < 6; ++g) {
              if (g % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // DS read
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
            }
          }
        }
      }
    }
  }
  LRP_T(1);
  // ---- epilogue: C/D layout col = lane&31, row i = (r&3) + 8*(r>>2) + 4*kg -> pixel (h, w) above ----
  // A lane holds ONE channel of 48 (pooled) or 96 pixels: written straight from the accumulators that is 96 two-
  // and one-byte stores per wave, 64 B contiguous at best — 9-10 k cycles of a whole tile's 86 k (s_memtime, round 3)
  // with the MFMA pipe idle.  The wave's frame of the tile is CONTIGUOUS in the channels-last output (the tile spans
  // the full width), so a whole tile goes through a wave-private staging area behind the patch instead: element by
  // element into LDS in output order (no barrier: a wave's LDS instructions execute in order), out again 16 bytes
  // per lane, 1 KB contiguous per store instruction, P2_STAGE bytes at a time.
  if constexpr (!SPLIT) {
    if (!fvalid) return;
    unsigned char* stage = patch + P2_LDS + wave * P2_STAGE;
    if (POOL) {
      // ReLU -> MaxPool((1,2,2)) in registers: registers 4g .. 4g+3 of an accumulator are the window
      // (h = 2g + {0,1}, w = 2kg + {0,1}) in row-major scan order.  Output: pooled value + position of
      // the first maximum (torch's rule), compared on the values that would have been stored.
      static_assert(!POOL || (P2_TH / 2) * (P2_W / 2) * N * 2 == P2_STAGE, "a wave's pooled frame tile fills the staging area");
      const int Hp = H >> 1;
      unsigned char args[NTW][MTW][4];
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int n = j * 32 + lr;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int wb = 0; wb < MTW; ++wb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16_t best;
            int arg;
            relu_pool4(acc[wb][j][4 * g], acc[wb][j][4 * g + 1], acc[wb][j][4 * g + 2], acc[wb][j][4 * g + 3], bv, best, arg);
            args[j][wb][g] = (unsigned char)arg;
            reinterpret_cast<bf16_t*>(stage)[(g * (P2_W / 2) + 2 * wb + kg) * N + n] = best;
          }
      }
      const int64_t o = (((int64_t)f * Hp + (h0 >> 1)) * (P2_W / 2)) * N;   // first element of the wave's tile
#pragma unroll
      for (int i = 0; i < P2_STAGE / 1024; ++i)
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(Y + o) + (i * 64 + lane) * 16) =
            *reinterpret_cast<const uint4*>(stage + (i * 64 + lane) * 16);
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int wb = 0; wb < MTW; ++wb)
#pragma unroll
          for (int g = 0; g < 4; ++g) stage[(g * (P2_W / 2) + 2 * wb + kg) * N + j * 32 + lr] = args[j][wb][g];
#pragma unroll
      for (int i = 0; i < P2_STAGE / 2048; ++i)
        *reinterpret_cast<uint4*>(code + o + (i * 64 + lane) * 16) = *reinterpret_cast<const uint4*>(stage + (i * 64 + lane) * 16);
      LRP_T(2);
      LRP_TEND();
      return;
    }
    // full-resolution output: RPC rows of the tile per pass
    constexpr int RPC = P2_STAGE / (P2_W * N * 2);
    static_assert(RPC >= 2 && RPC % 2 == 0 && P2_TH % RPC == 0, "whole pairs of tile rows per pass");
#pragma unroll
    for (int q = 0; q < P2_TH / RPC; ++q) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int n = j * 32 + lr;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int wb = 0; wb < MTW; ++wb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int hh = ((r >> 1) & 1) + 2 * (r >> 2), ww = 4 * wb + (r & 1) + 2 * kg;
            if (hh / RPC != q) continue;
            float v = acc[wb][j][r] + bv;
            if (relu) v = fmaxf(v, 0.f);
            reinterpret_cast<bf16_t*>(stage)[((hh - q * RPC) * P2_W + ww) * N + n] = f2bf(v);
          }
      }
      const int64_t o = (((int64_t)f * H + h0 + q * RPC) * P2_W) * N;
#pragma unroll
      for (int i = 0; i < P2_STAGE / 1024; ++i)
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(Y + o) + (i * 64 + lane) * 16) =
            *reinterpret_cast<const uint4*>(stage + (i * 64 + lane) * 16);
    }
    LRP_T(2);
    LRP_TEND();
    return;
  }
  if (POOL) {
    if (!fvalid) return;
    const int Hp = H >> 1;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = (j0 + j) * 32 + lr;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int wb = 0; wb < MTW; ++wb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16_t best;
          int arg;
          relu_pool4(acc[wb][j][4 * g], acc[wb][j][4 * g + 1], acc[wb][j][4 * g + 2], acc[wb][j][4 * g + 3], bv, best, arg);
          const int64_t o = ((((int64_t)f * Hp + (h0 >> 1) + g) * (P2_W / 2)) + 2 * (3 * g0 + wb) + kg) * N + n;
          Y[o] = best;
          code[o] = (unsigned char)arg;
        }
    }
    return;
  }
  if (!fvalid) return;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int n = (j0 + j) * 32 + lr;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int wb = 0; wb < MTW; ++wb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int hh = h0 + ((r >> 1) & 1) + 2 * (r >> 2), ww = 4 * (3 * g0 + wb) + (r & 1) + 2 * kg;
        float v = acc[wb][j][r] + bv;
        if (relu) v = fmaxf(v, 0.f);
        Y[(((int64_t)f * H + hh) * P2_W + ww) * N + n] = f2bf(v);
      }
  }
}

// Blocks 0 .. nfull-1 run whole tiles; the blocks after them run the remaining tiles split by frame
// (P2_TT / FT blocks per tile).  The host splits the tiles left over after the last FULL round of one workgroup per
// CU when they are few (1800 tiles on 256 CUs: 7 rounds + 8 tiles, which used to cost an eighth round).
// (256 PERSISTENT workgroups looping over their seven tiles instead of a workgroup per tile: the same within 1 %,
// round 3 — the dispatcher's relaunch of a 150 KB-LDS workgroup is not what the rounds lose.)
template <int CG, int NT, bool POOL, bool UNPOOL>
__global__ __launch_bounds__(256, 1) void conv_patch_kernel(const bf16_t* __restrict__ X,
                                                            const bf16_t* __restrict__ Wf,
                                                            const float* __restrict__ bias,
                                                            bf16_t* __restrict__ Y, unsigned char* __restrict__ code,
                                                            int F, int T, int H, int relu, int nfull) {
  extern __shared__ __attribute__((aligned(16))) unsigned char patch[];
  const int htiles = H / P2_TH;
  if ((int)blockIdx.x < nfull) {
    // XCD-aware tile order: block b runs on XCD b % 8 (each XCD has its own L2), so XCD x takes a CONTIGUOUS run of
    // tiles (h-tiles of a frame tile, then the next frame tile): the tiles resident together on an XCD are
    // neighbours in time and height and find each other's halo rows in that XCD's L2 instead of re-fetching them.
    const int xcd = blockIdx.x & 7, per_xcd = nfull >> 3, rem_xcd = nfull & 7;
    const int tile = xcd * per_xcd + (xcd < rem_xcd ? xcd : rem_xcd) + (int)(blockIdx.x >> 3);
    conv_patch_tile<CG, NT, POOL, false, UNPOOL>(patch, X, Wf, bias, Y, code, F, T, H, relu, (tile / htiles) * P2_TT,
                                         (tile % htiles) * P2_TH);
  } else {
    constexpr int FT = NT == 2 ? 1 : 2, PER = P2_TT / FT;
    const int sb = (int)blockIdx.x - nfull;
    const int tile = nfull + sb / PER;
    conv_patch_tile<CG, NT, POOL, true, UNPOOL>(patch, X, Wf, bias, Y, code, F, T, H, relu,
                                        (tile / htiles) * P2_TT + (sb % PER) * FT, (tile % htiles) * P2_TH);
  }
}

}  // namespace

#ifdef LRP_TIMING
extern "C" int lr_conv_patch_debug_times(long long* out_host, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_lrp_times), sizeof(long long) * 4096 * 8) != hipSuccess) return -1;
  if (reset) {
    static long long zeros[4096][8];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lrp_times), zeros, sizeof(zeros)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// fwd: 32 -> 64 channels (code != nullptr: fused ReLU + 2x2 max-pool epilogue); else the data gradient, 64 -> 32
// (unpool: X is the pooled gradient [F][Hin/2][12][64] and code the windows' codes, see conv_patch_tile).
// Wf: fragment-major weights (lr_conv3d_pack_weights, frag = 2).  H % 8 == 0, width 24.
int lr_conv_patch24(bool fwd, bool unpool, const void* X, const void* Wf, const float* bias, void* Y, unsigned char* code,
                    int F, int T, int Hin, int relu, bool sample, hipEvent_t e0, hipEvent_t e1, hipStream_t stream) {
  const bf16_t* x = (const bf16_t*)X;
  const bf16_t* w = (const bf16_t*)Wf;
  bf16_t* y = (bf16_t*)Y;
  // tiles left over after the last full round of one workgroup per CU, when they are few, run split by frame
  // (4 blocks per tile forward, 2 for the data gradient) so that they do not cost a round of their own
  const int ntiles = ((F + P2_TT - 1) / P2_TT) * (Hin / P2_TH);
  const int left = ntiles % kChipCUs;
  const int nsplit = ntiles > kChipCUs && left > 0 && left <= kChipCUs / 4 ? left : 0;
  const int nfull = ntiles - nsplit;
  const dim3 pgrid((unsigned)(nfull + nsplit * (fwd ? 4 : 2)));
  static bool attr_set[4] = {false, false, false, false};
  lr_clear_error();
#define LR_PATCH(IDX, ...)                                                                                     \
  do {                                                                                                        \
    if (!attr_set[IDX]) {                                                                                     \
      if (hipFuncSetAttribute((const void*)conv_patch_kernel<__VA_ARGS__>,                                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS + 4 * P2_STAGE) != hipSuccess)              \
        return LR_ERR_LAUNCH;                                                                                 \
      attr_set[IDX] = true;                                                                                   \
    }                                                                                                         \
    if (sample) hipExtLaunchKernelGGL((conv_patch_kernel<__VA_ARGS__>), pgrid, dim3(256), P2_LDS + 4 * P2_STAGE, stream, e0, e1, \
                                      0, x, w, bias, y, code, F, T, Hin, relu, nfull);                        \
    else hipLaunchKernelGGL((conv_patch_kernel<__VA_ARGS__>), pgrid, dim3(256), P2_LDS + 4 * P2_STAGE, stream, x, w, bias, y, \
                            code, F, T, Hin, relu, nfull);                                                    \
  } while (0)
  if (unpool && (fwd || !code)) return LR_ERR_UNSUPPORTED;
  if (fwd && code) LR_PATCH(2, 1, 2, true, false);
  else if (fwd) LR_PATCH(0, 1, 2, false, false);
  else if (unpool) LR_PATCH(3, 2, 1, false, true);
  else LR_PATCH(1, 2, 1, false, false);
#undef LR_PATCH
  return lr_launch_status();
}
