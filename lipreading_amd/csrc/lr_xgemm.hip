// lr_xgemm.hip — fp32 GEMM contracted on the gfx950 bf16 matrix cores by operand splitting.
//
// C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C + bias[N], fp32 in memory, where each fp32
// operand element x is split into two bf16 terms, hi = bf16(x) and lo = bf16(x - hi) (16 mantissa
// bits together), and the product is accumulated in fp32 as
//     a_hi b_hi + a_hi b_lo + a_lo b_hi          (the lo*lo term, 2^-16 of the result, is dropped)
// with v_mfma_f32_32x32x16_bf16: 3 MFMAs at the bf16 rate (2.5 PF dense) instead of one at the fp32
// rate (157 TF) — a 5x higher ceiling at ~1e-5 relative accuracy.  An operand the caller declares
// bf16-exact (every element already a bf16 value, e.g. features produced by the bf16 conv frontend)
// has no lo term: 2 MFMAs.
//
// Used ONLY where the caller asks for it (lr_rnn_layer_* with LR_RNN_PROJ_BF16X3: the input
// projection of the recurrent layers in the pixel regime, whose input is the bf16 frontend's
// output).  The reference-faithful landmark regime keeps the exact fp32 MFMA path (lr_gemm.hip)
// for its 1e-4 loss parity.  No reference counterpart: the reference's input projection is inside
// nn.GRU/nn.LSTM (better_model.py:74).
//
// Two passes.  (1) pack: each operand is split ONCE into bf16 planes in the caller's workspace,
// K-contiguous ([rows][K]; operands that are row-contiguous in memory are transposed on the way,
// 32x32 tiles through LDS).  Splitting inside the contraction instead costs ~8 VALU lane-ops per
// element per tile that uses it — measured 2x the MFMA time at 128x128 tiles.  (2) contract: 256
// threads = 2x2 waves, workgroup tile 128x128, wave tile 64x64 (2x2 MFMA tiles), BK = 64 per LDS
// stage; plane tiles are staged as bf16 [row][64 + 8] (144-byte rows: conflict-free for the fragment
// reads' lane groups); one stage of 16-byte global loads is in flight per workgroup (issued right behind
// the previous stage's LDS write) behind LDS-only barriers; K is split over workgroups
// until ~2 are resident per CU, partial sums reduced in fixed order (deterministic); tiles are
// ordered so that the ~64 resident on one XCD share operand panels in its L2.
//
// Where it stands (round 3, tools/bench_xgemm.py, contraction alone): 0.5-0.75 PF/s on the steps' shapes (K >= 1536),
// i.e. 20-30 % of the bf16 peak.  Measured and dropped: two LDS stage buffers with ONE barrier per stage instead of
// store / barrier / read / barrier on one buffer — no change (+-3 % on every shape), so the barriers are not what
// bounds it.  What does: a 128 x 128 tile pulls 3-4 plane tiles (24-32 KB) through the CU's vector L1 per stage of
// 512-768 MFMA clocks, ~47 B/clk with two workgroups per CU — the L1's rate.
// Measured and dropped as well (round 3): a 256 x 256 tile on 8 waves (wave tile 128 x 64: 6-9 fragment reads for 8-16
// MFMAs, two LDS buffers with one barrier per stage, loads two stages ahead) for the products with at most three
// planes.  Bit-compatible results, and SLOWER on the steps' shapes: 173 us instead of 122 for the first layer's dW_ih
// (1536 x 3456 x 2400), 93 instead of 79 for its data gradient — these outputs are 84-135 tiles of 256 x 256, so the
// chip is only filled by splitting K three ways, and the slabs' write + combine pass costs more than the larger tile
// saves; a tile this size wants outputs of >= 256 tiles (4096 x 4096).
#include "lr_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short bf16_t;

// 64-k stages — half the barriers and fragment-read restarts per MFMA of a 32-k stage — with ONE stage of loads in
// flight: the next stage's loads are issued right after this stage's LDS write and land under its 32-48 MFMAs per wave
// (two register sets of 64-k stages do not fit two workgroups per CU: 232 + 64 registers).  Rows of 64 + 8 elements (144
// bytes) are as conflict-free for the fragment reads' lane groups as the 80-byte rows of a 32-k stage.
// MEASURED (round 4, same box) against rounds 1-3's 32-k stages behind a three-slot register ring (whose first slot's LDS
// write drained the two younger stages as well — hipcc's wait counts for loads issued and consumed under conditions):
// pixel step 2.59 -> 2.55 ms; K = 3456 projection 102.9 -> 99.3 us, its dx 79.3 -> 73.5, the other shapes within 1 %.
constexpr int XBM = 128, XBN = 128, XBK = 64, XLD = XBK + 8;
constexpr int XDEPTH = 1;                     // stages of global loads in flight
constexpr int XUPR = XBK / 8;                 // 16-byte units per plane-tile row
constexpr int XUSH = 3;                       // log2 of it
constexpr int XNU = XBM * XUPR / 256;         // units per thread and plane tile
constexpr int XGROUP_M = 4;   // m-panels per group of the tile order

struct XArgs {
  const bf16_t* Ah;   // [M][K] planes, leading dimension ldp (multiple of 8: 16-byte rows)
  const bf16_t* Al;
  const bf16_t* Bh;   // [N][K]
  const bf16_t* Bl;
  float* C;
  float* C1;          // rows >= split_row go to C1 + (row - split_row) * ldc (two stacked outputs)
  int split_row;
  const float* bias;
  int M, N, K, ldp, ldc;
  float alpha, beta;
  int k_chunk;     // K range per split (multiple of XBK)
  float* slabs;    // split-K partial sums [splits][M][N], or nullptr
  int nx, ny, splits, per_xcd;   // tile grid and the number of tiles each XCD takes
  // batch (blockIdx.y): independent products of one shape sharing a launch
  int64_t a_bstride, b_bstride;  // bf16 elements between the batches' planes (hi and lo alike)
  int64_t slab_bstride;          // floats between the batches' split-K slabs
  float* Cb;                     // C of batch 1 (batch 0 writes C); no C1 split when batched
  int c_bf16;                    // C is a bf16 matrix (ldc in elements; beta must be 0, no C1 split)
};

// ---- pass 1: fp32 [rows][cols] (or its transpose) -> bf16 hi (+ lo) planes [rows'][ldp] -------------
// transpose == 0: out[r][c] = split(in[r][c]);  transpose == 1: out[c][r] = split(in[r][c]).
// Output columns [0, owidth) are written, zero beyond the logical width (the contraction reads whole
// 16-byte units up to ldp); hi / lo may point into the middle of a larger plane (an operand assembled
// from several blocks).  32x32 tiles, 256 threads; the grid covers the OUTPUT extent.
// shift / period: input row r is read from row r + shift, or as zero when (r % period) + shift leaves
// [0, period) — the previous / next time step of a [B*T][cols] sequence tensor (period = T).
__global__ __launch_bounds__(256) void xpack_kernel(const float* __restrict__ in, int ld_in, int rows, int cols,
                                                    int transpose, bf16_t* __restrict__ hi,
                                                    bf16_t* __restrict__ lo, int ldp, int owidth, int shift,
                                                    int period) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  // tile origin in OUTPUT coordinates (orow, ocol); input origin is the same or swapped
  const int or0 = blockIdx.y * 32, oc0 = blockIdx.x * 32;
  const int ir0 = transpose ? oc0 : or0, ic0 = transpose ? or0 : oc0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = ir0 + ty + 8 * i, c = ic0 + tx;
    bool ok = r < rows && c < cols;
    if (ok && shift != 0) {
      const int tt = r % period + shift;
      ok = tt >= 0 && tt < period;
    }
    tile[ty + 8 * i][tx] = ok ? in[(int64_t)(r + shift) * ld_in + c] : 0.f;
  }
  __syncthreads();
  const int orows = transpose ? cols : rows;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int orow = or0 + ty + 8 * i, ocol = oc0 + tx;
    if (orow >= orows || ocol >= owidth) continue;
    const float v = transpose ? tile[tx][ty + 8 * i] : tile[ty + 8 * i][tx];
    const __bf16 h = (__bf16)v;
    hi[(int64_t)orow * ldp + ocol] = __builtin_bit_cast(bf16_t, h);
    if (lo) {
      const __bf16 l = (__bf16)(v - (float)h);
      lo[(int64_t)orow * ldp + ocol] = __builtin_bit_cast(bf16_t, l);
    }
  }
}

// Several pack blocks in one launch (blockIdx.z): the operands of one product — and of the products
// of both directions — are each a handful of small blocks, and a launch per block cost ~5 us apiece.
struct XPackBlock {
  const float* in;
  bf16_t* hi;
  bf16_t* lo;
  int ld_in, rows, cols, transpose, ldp, owidth, shift, period;
  int in_bf16;   // the source is a bf16 matrix (ld_in in elements), e.g. the conv frontend's features
};
constexpr int XPACK_MAX = 8;
struct XPackArgs {
  XPackBlock b[XPACK_MAX];
};
// 64 x 64 tiles, 16 bytes per lane on both sides where the block's addresses allow: a lane loads four consecutive
// input columns (float4, or four bf16) and stores EIGHT consecutive plane elements (one uint4 per plane) that it
// collects from the LDS tile — down a tile column when the block is transposed; tile rows of 65 floats make both
// directions conflict-free.  (Round 2's form — 32 x 32 tiles, one element per lane, 2-byte stores — moved ~0.8 TB/s:
// the eight pack launches of a pixel step were 150 us, 60 of them on the critical path.)
__global__ __launch_bounds__(256) void xpack_multi_kernel(XPackArgs a) {
  __shared__ float tile[64][65];
  const XPackBlock& k = a.b[blockIdx.z];
  const int orows = k.transpose ? k.cols : k.rows;
  const int or0 = blockIdx.y * 64, oc0 = blockIdx.x * 64;
  if (or0 >= orows || oc0 >= k.owidth) return;   // workgroup-uniform
  const int t = threadIdx.x;
  const int ir0 = k.transpose ? oc0 : or0, ic0 = k.transpose ? or0 : oc0;
  const bool in_vec = (k.ld_in & 3) == 0 && (reinterpret_cast<uintptr_t>(k.in) & (k.in_bf16 ? 7 : 15)) == 0;
  const bool out_vec = (k.ldp & 7) == 0 && (k.owidth & 7) == 0 && (reinterpret_cast<uintptr_t>(k.hi) & 15) == 0 &&
                       (!k.lo || (reinterpret_cast<uintptr_t>(k.lo) & 15) == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = (t >> 4) + 16 * i, cl = 4 * (t & 15);
    const int r = ir0 + rl, c = ic0 + cl;
    bool ok = r < k.rows;
    if (ok && k.shift != 0) {
      const int tt = r % k.period + k.shift;
      ok = tt >= 0 && tt < k.period;
    }
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      const int64_t idx = (int64_t)(r + k.shift) * k.ld_in + c;
      if (in_vec && c + 3 < k.cols) {
        if (k.in_bf16) {
          const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(k.in) + idx);
          v[0] = __builtin_bit_cast(float, u.x << 16); v[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
          v[2] = __builtin_bit_cast(float, u.y << 16); v[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
        } else {
          const float4 f = *reinterpret_cast<const float4*>(k.in + idx);
          v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < k.cols)
            v[j] = k.in_bf16 ? __builtin_bit_cast(float, (unsigned)reinterpret_cast<const bf16_t*>(k.in)[idx + j] << 16)
                             : k.in[idx + j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[rl][cl + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int orl = (t >> 3) + 32 * i, ocl = 8 * (t & 7);
    const int orow = or0 + orl, ocol = oc0 + ocl;
    if (orow >= orows || ocol >= k.owidth) continue;
    unsigned hw[4], lw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v0 = k.transpose ? tile[ocl + 2 * j][orl] : tile[orl][ocl + 2 * j];
      const float v1 = k.transpose ? tile[ocl + 2 * j + 1][orl] : tile[orl][ocl + 2 * j + 1];
      const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
      hw[j] = (unsigned)__builtin_bit_cast(bf16_t, h0) | ((unsigned)__builtin_bit_cast(bf16_t, h1) << 16);
      const __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
      lw[j] = (unsigned)__builtin_bit_cast(bf16_t, l0) | ((unsigned)__builtin_bit_cast(bf16_t, l1) << 16);
    }
    const int64_t o = (int64_t)orow * k.ldp + ocol;
    if (out_vec) {
      *reinterpret_cast<uint4*>(k.hi + o) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      if (k.lo) *reinterpret_cast<uint4*>(k.lo + o) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (ocol + j >= k.owidth) continue;
        k.hi[o + j] = (bf16_t)(hw[j >> 1] >> (16 * (j & 1)));
        if (k.lo) k.lo[o + j] = (bf16_t)(lw[j >> 1] >> (16 * (j & 1)));
      }
    }
  }
}

// ---- pass 2: C = sum over the retained (hi, lo) products of A_plane . B_plane^T ------------------------
// AX / BX: operand is bf16-exact, no lo plane.
template <bool AX, bool BX>
__global__ __launch_bounds__(256) void xgemm_kernel(XArgs g) {
  __shared__ __attribute__((aligned(16))) bf16_t Ah[XBM * XLD];
  __shared__ __attribute__((aligned(16))) bf16_t Al[AX ? 8 : XBM * XLD];
  __shared__ __attribute__((aligned(16))) bf16_t Bh[XBN * XLD];
  __shared__ __attribute__((aligned(16))) bf16_t Bl[BX ? 8 : XBN * XLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order.  Workgroup b runs on XCD b % 8 (observed placement; only speed depends
  // on it): XCD x takes the contiguous range [x * per_xcd, (x + 1) * per_xcd) of an order in which
  // neighbouring tiles share operand panels (split slowest; groups of XGROUP_M m-panels, n across a
  // group, m fastest).
  const int t = (int)(blockIdx.x & 7) * g.per_xcd + (int)(blockIdx.x >> 3);
  const int tiles = g.nx * g.ny;
  if ((int)(blockIdx.x >> 3) >= g.per_xcd || t >= tiles * g.splits) return;   // workgroup-uniform
  const int zs = t / tiles, tt = t - zs * tiles;
  const int grp = tt / (XGROUP_M * g.nx), first_m = grp * XGROUP_M;
  const int gm = min(XGROUP_M, g.ny - first_m);
  const int in_grp = tt - grp * XGROUP_M * g.nx;
  const int m0 = (first_m + in_grp % gm) * XBM, n0 = (in_grp / gm) * XBN;
  const int kbeg = zs * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
  const int bz = blockIdx.y;
  if (bz) {   // workgroup-uniform
    g.Ah += (int64_t)bz * g.a_bstride;
    if (!AX) g.Al += (int64_t)bz * g.a_bstride;
    g.Bh += (int64_t)bz * g.b_bstride;
    if (!BX) g.Bl += (int64_t)bz * g.b_bstride;
    if (g.slabs) g.slabs += (int64_t)bz * g.slab_bstride;
    g.C = g.Cb;
    g.C1 = g.Cb;
  }

  // a plane tile is 128 rows x 64 k = 1024 16-byte units: four per thread (row = e / 8, unit = e % 8).  Rows past
  // the matrix edge are CLAMPED, not zeroed (their products land in accumulator rows / columns that are never
  // stored, and the planes hold finite numbers); only the K tail needs zeros, and only in a split's last stage,
  // so the loads of every other stage are unconditional (a per-load predicate makes hipcc branch around each load).
  constexpr int NPL = 2 + (AX ? 0 : 1) + (BX ? 0 : 1);
  // (element offsets of the thread's two units inside a plane — a plane stays below 2^31 elements —, added to the
  // uniform plane pointer + k0: scalar base + 32-bit lane offset addressing, no 64-bit address registers)
  unsigned offa[XNU], offb[XNU];
#pragma unroll
  for (int i = 0; i < XNU; ++i) {
    const int e = tid + i * 256, row = e >> XUSH, ku = 8 * (e & (XUPR - 1));
    offa[i] = (unsigned)min(m0 + row, g.M - 1) * (unsigned)g.ldp + ku;
    offb[i] = (unsigned)min(n0 + row, g.N - 1) * (unsigned)g.ldp + ku;
  }
  uint4 rr[XDEPTH][NPL][XNU];
  auto load = [&](int slot, int k0) {
    const bf16_t* const pl_ptr[4] = {g.Ah + k0, AX ? nullptr : g.Al + k0, g.Bh + k0, BX ? nullptr : g.Bl + k0};
    if (k0 + XBK <= kend) {   // workgroup-uniform
#pragma unroll
      for (int i = 0; i < XNU; ++i) {
        int pl = 0;
        rr[slot][pl++][i] = *reinterpret_cast<const uint4*>(pl_ptr[0] + offa[i]);
        if (!AX) rr[slot][pl++][i] = *reinterpret_cast<const uint4*>(pl_ptr[1] + offa[i]);
        rr[slot][pl++][i] = *reinterpret_cast<const uint4*>(pl_ptr[2] + offb[i]);
        if (!BX) rr[slot][pl++][i] = *reinterpret_cast<const uint4*>(pl_ptr[3] + offb[i]);
      }
    } else {                  // the K tail: units at or past kend are zero (planes are zero-padded to ldp only)
#pragma unroll
      for (int i = 0; i < XNU; ++i) {
        const bool ok = k0 + 8 * ((tid + i * 256) & (XUPR - 1)) < kend;
        const int back = ok ? 0 : k0 - kbeg;   // a unit past the end reads the split's first stage instead
        int pl = 0;
        auto get = [&](const bf16_t* p, unsigned off) {
          uint4 v = *reinterpret_cast<const uint4*>(p + off - back);
          v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
          return v;
        };
        rr[slot][pl++][i] = get(pl_ptr[0], offa[i]);
        if (!AX) rr[slot][pl++][i] = get(pl_ptr[1], offa[i]);
        rr[slot][pl++][i] = get(pl_ptr[2], offb[i]);
        if (!BX) rr[slot][pl++][i] = get(pl_ptr[3], offb[i]);
      }
    }
  };
  auto store = [&](int slot) {
#pragma unroll
    for (int i = 0; i < XNU; ++i) {
      const int e = tid + i * 256, o = (e >> XUSH) * XLD + 8 * (e & (XUPR - 1));
      int pl = 0;
      *reinterpret_cast<uint4*>(&Ah[o]) = rr[slot][pl++][i];
      if (!AX) *reinterpret_cast<uint4*>(&Al[o]) = rr[slot][pl++][i];
      *reinterpret_cast<uint4*>(&Bh[o]) = rr[slot][pl++][i];
      if (!BX) *reinterpret_cast<uint4*>(&Bl[o]) = rr[slot][pl++][i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lk = lane >> 5;
  const int fa = (wm * 64 + lr) * XLD + lk * 8, fb = (wn * 64 + lr) * XLD + lk * 8;   // fragment offsets (elements)
  constexpr int NFRAG = 2 * NPL;                   // fragment reads per k16 step
  constexpr int NMMA = 4 * (1 + (AX ? 0 : 1) + (BX ? 0 : 1));
#pragma unroll
  for (int sl = 0; sl < XDEPTH; ++sl)
    if (kbeg + sl * XBK < kend) load(sl, kbeg + sl * XBK);
  for (int kb = kbeg; kb < kend; kb += XDEPTH * XBK) {
#pragma unroll
    for (int sl = 0; sl < XDEPTH; ++sl) {   // fully unrolled: the stage registers are static
      const int k0 = kb + sl * XBK;
      if (k0 < kend) {                     // workgroup-uniform
        // LDS-only barriers: __syncthreads() would also drain vmcnt (harmless with one stage in flight, whose
        // loads were consumed by store() just above, but it is the same instruction count)
        lr_lds_barrier();
        store(sl);
        lr_lds_barrier();
        if (k0 + XDEPTH * XBK < kend) load(sl, k0 + XDEPTH * XBK);
        // both k16 steps' fragments are read up front (left alone, hipcc puts every read next to its first use
        // and each group of MFMAs waits out an LDS round trip), then the MFMAs run term by term over the four
        // accumulators: small terms first, so they are not absorbed by a large partial sum, and consecutive
        // MFMAs never wait for each other's result
        auto half = [&](auto HF) {   // 32 k at a time
          constexpr int hf = decltype(HF)::value;
          bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
  #pragma unroll
          for (int ks = 0; ks < 2; ++ks)
  #pragma unroll
            for (int i = 0; i < 2; ++i) {
              ah[ks][i] = *reinterpret_cast<const bf16x8*>(&Ah[fa + i * 32 * XLD + hf * 32 + ks * 16]);
              if (!AX) al[ks][i] = *reinterpret_cast<const bf16x8*>(&Al[fa + i * 32 * XLD + hf * 32 + ks * 16]);
              bh[ks][i] = *reinterpret_cast<const bf16x8*>(&Bh[fb + i * 32 * XLD + hf * 32 + ks * 16]);
              if (!BX) bl[ks][i] = *reinterpret_cast<const bf16x8*>(&Bl[fb + i * 32 * XLD + hf * 32 + ks * 16]);
            }
  #pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            if (!BX) {
  #pragma unroll
              for (int i = 0; i < 2; ++i)
  #pragma unroll
                for (int j = 0; j < 2; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
            }
            if (!AX) {
  #pragma unroll
              for (int i = 0; i < 2; ++i)
  #pragma unroll
                for (int j = 0; j < 2; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
            }
  #pragma unroll
            for (int i = 0; i < 2; ++i)
  #pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
          }
          // pinned order: the first step's reads, then one of the second step's reads behind each of the first
          // MFMAs, then the remaining MFMAs
          __builtin_amdgcn_sched_group_barrier(0x100, NFRAG, 0);
  #pragma unroll
          for (int q = 0; q < NFRAG; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * NMMA - NFRAG, 0);
        };
        half(std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{});
      }
    }
  }
  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + lr;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= g.M) continue;
        if (g.slabs) {
          g.slabs[((int64_t)zs * g.M + row) * g.N + col] = acc[i][j][r];
          continue;
        }
        float out = g.alpha * acc[i][j][r] + bv;
        if (g.c_bf16) {
          const __bf16 hb = (__bf16)out;
          reinterpret_cast<bf16_t*>(g.C)[(int64_t)row * g.ldc + col] = __builtin_bit_cast(bf16_t, hb);
          continue;
        }
        float* c = (row < g.split_row ? g.C + (int64_t)row * g.ldc : g.C1 + (int64_t)(row - g.split_row) * g.ldc) + col;
        if (g.beta != 0.f) out += g.beta * *c;
        *c = out;
      }
    }
}

// deterministic split-K combine (fixed order over the slabs), then alpha / beta / bias.  Four consecutive
// elements per thread (16-byte loads of every slab in flight together) when N % 4 == 0; this pass is pure HBM
// streaming of (splits + 1) * M * N floats.
__global__ __launch_bounds__(256) void xsplitk_reduce_kernel(const float* __restrict__ slabs, int splits,
                                                             float* __restrict__ C, float* __restrict__ C1,
                                                             int split_row, int ldc, const float* __restrict__ bias,
                                                             int M, int N, float alpha, float beta,
                                                             float* __restrict__ Cb, int64_t slab_bstride, int c_bf16) {
  const int64_t total = (int64_t)M * N;
  if (blockIdx.y) {   // batch 1
    slabs += slab_bstride;
    C = Cb;
    C1 = Cb;
  }
  auto finish = [&](int64_t i, float s) {
    const int row = (int)(i / N), col = (int)(i - (int64_t)row * N);
    float out = alpha * s;
    if (bias) out += bias[col];
    if (c_bf16) {
      const __bf16 hb = (__bf16)out;
      reinterpret_cast<bf16_t*>(C)[(int64_t)row * ldc + col] = __builtin_bit_cast(bf16_t, hb);
      return;
    }
    float* c = (row < split_row ? C + (int64_t)row * ldc : C1 + (int64_t)(row - split_row) * ldc) + col;
    if (beta != 0.f) out += beta * *c;
    *c = out;
  };
  if ((N & 3) == 0) {
    const int64_t quads = total >> 2;
    const bool vec_out = !c_bf16 && (ldc & 3) == 0 &&
                         ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(C1) |
                           (bias ? reinterpret_cast<uintptr_t>(bias) : 0)) & 15) == 0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
      const float4* src = reinterpret_cast<const float4*>(slabs) + q;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      int z = 0;
      for (; z + 4 <= splits; z += 4) {   // four slabs' loads in flight; summed in slab order
        const float4 v0 = src[(int64_t)z * quads], v1 = src[(int64_t)(z + 1) * quads];
        const float4 v2 = src[(int64_t)(z + 2) * quads], v3 = src[(int64_t)(z + 3) * quads];
        s.x = (((s.x + v0.x) + v1.x) + v2.x) + v3.x;
        s.y = (((s.y + v0.y) + v1.y) + v2.y) + v3.y;
        s.z = (((s.z + v0.z) + v1.z) + v2.z) + v3.z;
        s.w = (((s.w + v0.w) + v1.w) + v2.w) + v3.w;
      }
      for (; z < splits; ++z) {
        const float4 v = src[(int64_t)z * quads];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int64_t i = q << 2;
      if (vec_out) {   // plain fp32 output with 16-byte aligned rows: one store (and one read for beta) per quad
        const int row = (int)(i / N), col = (int)(i - (int64_t)row * N);
        float4* c = reinterpret_cast<float4*>((row < split_row ? C + (int64_t)row * ldc : C1 + (int64_t)(row - split_row) * ldc) + col);
        float4 o = make_float4(alpha * s.x, alpha * s.y, alpha * s.z, alpha * s.w);
        if (bias) {
          const float4 b = *reinterpret_cast<const float4*>(bias + col);
          o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
        }
        if (beta != 0.f) {
          const float4 p = *c;
          o.x += beta * p.x; o.y += beta * p.y; o.z += beta * p.z; o.w += beta * p.w;
        }
        *c = o;
        continue;
      }
      finish(i, s.x);
      finish(i + 1, s.y);
      finish(i + 2, s.z);
      finish(i + 3, s.w);
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(int64_t)z * total + i];
    finish(i, s);
  }
}

size_t pad64(size_t n) { return (n + 63) / 64 * 64; }
int ldp_of(int K) { return (K + 7) / 8 * 8; }
// floats of workspace taken by one operand's planes ([rows][ldp] bf16, hi + optionally lo)
size_t plane_floats(int rows, int K, bool exact) {
  return pad64(((size_t)rows * ldp_of(K) * (exact ? 1 : 2) + 1) / 2);
}

// Split K so that the workgroups fill the 512 resident slots (2 per CU) in as few rounds as possible:
// cost(s) = rounds(s) * (K / s + fixed per-workgroup overhead) + the slabs: every split writes a 128 x 128
// fp32 tile that the combine pass reads back (2 x 64 KB of HBM traffic per tile and split ~ 2.2 k-units, the
// unit being what one k element of a tile round costs) — without that term a 324-tile, K = 2400 product (the
// first layer's dW_ih) was split three ways and its combine pass (101 us) cost more than the product.
// >= 4 stages per split.
int want_splits(int M, int N, int K) {
  const long tiles = (long)((M + XBM - 1) / XBM) * ((N + XBN - 1) / XBN);
  if (tiles >= 384) return 1;
  long max_split = K / (4 * XBK);
  if (max_split > 16) max_split = 16;
  long best = 1, best_cost = -1;
  for (long sp = 1; sp <= max_split; ++sp) {
    const long rounds = (tiles * sp + 511) / 512;
    const long cost = rounds * (K / sp + 96) + (sp > 1 ? (sp * tiles * 22) / 10 : 0);
    if (best_cost < 0 || cost < best_cost) { best = sp; best_cost = cost; }
  }
  return (int)best;
}

// planes [orows][owidth of ldp] of an operand block stored [rows][cols] (transpose: planes are
// [cols][rows]); hi / lo may be offset into a larger plane
int pack_operand(const float* in, int ld_in, int rows, int cols, int transpose, bf16_t* hi, bf16_t* lo, int ldp,
                 int owidth, hipStream_t stream, int shift = 0, int period = 1) {
  const int orows = transpose ? cols : rows;
  LR_LAUNCH(xpack_kernel, dim3((owidth + 31) / 32, (orows + 31) / 32), dim3(256), 0, stream, in, ld_in, rows, cols,
            transpose, hi, lo, ldp, owidth, shift, period);
  return lr_launch_status();
}

// a list of pack blocks launched together
struct PackList {
  XPackArgs a;
  int n = 0, gx = 0, gy = 0;
  void add(const float* in, int ld_in, int rows, int cols, int transpose, bf16_t* hi, bf16_t* lo, int ldp, int owidth,
           int shift = 0, int period = 1, int in_bf16 = 0) {
    XPackBlock& k = a.b[n++];
    k.in = in; k.hi = hi; k.lo = lo;
    k.ld_in = ld_in; k.rows = rows; k.cols = cols; k.transpose = transpose; k.ldp = ldp; k.owidth = owidth;
    k.shift = shift; k.period = period; k.in_bf16 = in_bf16;
    const int orows = transpose ? cols : rows;
    if ((owidth + 63) / 64 > gx) gx = (owidth + 63) / 64;
    if ((orows + 63) / 64 > gy) gy = (orows + 63) / 64;
  }
  int launch(hipStream_t stream) {
    LR_LAUNCH(xpack_multi_kernel, dim3(gx, gy, n), dim3(256), 0, stream, a);
    return lr_launch_status();
  }
};

// C (+ C1 below split_row) = alpha * A_planes . B_planes^T + beta * C + bias; slabs: split-K workspace.
// nbatch = 2: a second product of the same shape in the same launch (planes a_bstride / b_bstride
// bf16 elements further, output Cb, its slabs behind the first batch's).
int contract(const bf16_t* Ahp, const bf16_t* Alp, const bf16_t* Bhp, const bf16_t* Blp, int M, int N, int K,
             float alpha, float beta, float* C, float* C1, int split_row, int ldc, const float* bias, float* slabs,
             size_t slab_floats, hipStream_t stream, int nbatch = 1, int64_t a_bstride = 0, int64_t b_bstride = 0,
             float* Cb = nullptr, int c_bf16 = 0) {
  const int ldp = ldp_of(K);
  // split-K as far as the remaining workspace allows (none: a single pass, just slower)
  int splits = want_splits(M, N * nbatch, K);
  while (splits > 1 && (size_t)splits * M * N * nbatch > slab_floats) --splits;
  int chunk = (K + splits - 1) / splits;
  chunk = (chunk + XBK - 1) / XBK * XBK;
  splits = (K + chunk - 1) / chunk;
  XArgs g;
  g.Ah = Ahp; g.Al = Alp; g.Bh = Bhp; g.Bl = Blp;
  g.C = C; g.C1 = C1 ? C1 : C; g.split_row = C1 ? split_row : M; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.ldp = ldp; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta;
  g.k_chunk = chunk;
  g.slabs = splits > 1 ? slabs : nullptr;
  g.nx = (N + XBN - 1) / XBN;
  g.ny = (M + XBM - 1) / XBM;
  g.splits = splits;
  g.a_bstride = a_bstride; g.b_bstride = b_bstride;
  g.slab_bstride = (int64_t)splits * M * N;
  g.Cb = Cb ? Cb : C;
  g.c_bf16 = c_bf16;
  const int ntile = g.nx * g.ny * splits;
  g.per_xcd = (ntile + 7) / 8;
  dim3 grid(8 * g.per_xcd, nbatch);
  lr_clear_error();
  const bool ax = Alp == nullptr, bx = Blp == nullptr;
  if (ax && bx) hipLaunchKernelGGL((xgemm_kernel<true, true>), grid, dim3(256), 0, stream, g);
  else if (ax) hipLaunchKernelGGL((xgemm_kernel<true, false>), grid, dim3(256), 0, stream, g);
  else if (bx) hipLaunchKernelGGL((xgemm_kernel<false, true>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((xgemm_kernel<false, false>), grid, dim3(256), 0, stream, g);
  int st = lr_launch_status();
  if (st != LR_OK || splits == 1) return st;
  const int64_t total = (int64_t)M * N;
  int blocks = (int)(((N & 3) == 0 ? total / 4 : total) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  LR_LAUNCH(xsplitk_reduce_kernel, dim3(blocks, nbatch), dim3(256), 0, stream, (const float*)g.slabs, splits, C, g.C1,
            g.split_row, ldc, bias, M, N, alpha, beta, g.Cb, g.slab_bstride, c_bf16);
  return lr_launch_status();
}

// carve the two operands' planes out of a workspace; returns false when it is too small
struct Planes {
  bf16_t *Ah, *Al, *Bh, *Bl;
  float* slabs;
  size_t slab_floats;
};
bool carve(void* workspace, size_t workspace_bytes, int M, int N, int K, bool a_exact, bool b_exact, Planes* p) {
  const int ldp = ldp_of(K);
  const size_t fa = plane_floats(M, K, a_exact), fb = plane_floats(N, K, b_exact);
  const size_t avail = workspace_bytes / sizeof(float);
  if (avail < fa + fb) return false;
  float* ws = (float*)workspace;
  p->Ah = (bf16_t*)ws;
  p->Al = a_exact ? nullptr : p->Ah + (size_t)M * ldp;
  p->Bh = (bf16_t*)(ws + fa);
  p->Bl = b_exact ? nullptr : p->Bh + (size_t)N * ldp;
  p->slabs = ws + fa + fb;
  p->slab_floats = avail - fa - fb;
  return true;
}

}  // namespace

// bytes of workspace lr_xgemm needs: the operands' bf16 planes + split-K slabs
extern "C" size_t lr_xgemm_workspace_bytes(int transA, int transB, int M, int N, int K) {
  (void)transA;
  (void)transB;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  size_t f = plane_floats(M, K, false) + plane_floats(N, K, false);
  const int sp = want_splits(M, N, K);
  if (sp > 1) f += pad64((size_t)sp * M * N);
  return f * sizeof(float);
}

// Internal entry: same operand conventions as lr_sgemm_impl (transA: A stored [K][M]; transB: B
// stored [N][K]).
int lr_xgemm_impl(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                  const float* B, int ldb, float beta, float* C, int ldc, const float* bias, int a_exact,
                  int b_exact, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  LR_CHECK_ARG(A && B && C && workspace);
  LR_CHECK_ARG(M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N);
  const int ldp = ldp_of(K);
  Planes pl;
  if (!carve(workspace, workspace_bytes, M, N, K, a_exact != 0, b_exact != 0, &pl)) return LR_ERR_WORKSPACE;
  // A planes [M][K]: stored [K][M] when transA (rows = K, cols = M, transposed), else [M][K]
  int st = transA ? pack_operand(A, lda, K, M, 1, pl.Ah, pl.Al, ldp, ldp, stream)
                  : pack_operand(A, lda, M, K, 0, pl.Ah, pl.Al, ldp, ldp, stream);
  if (st != LR_OK) return st;
  // B planes [N][K]: stored [N][K] when transB, else [K][N] (transposed)
  st = transB ? pack_operand(B, ldb, N, K, 0, pl.Bh, pl.Bl, ldp, ldp, stream)
              : pack_operand(B, ldb, K, N, 1, pl.Bh, pl.Bl, ldp, ldp, stream);
  if (st != LR_OK) return st;
  return contract(pl.Ah, pl.Al, pl.Bh, pl.Bl, M, N, K, alpha, beta, C, nullptr, M, ldc, bias, pl.slabs,
                  pl.slab_floats, stream);
}

// ---- the products of a recurrent layer around its recurrence, all directions in ONE contraction -------
// (lr_rnn.hip, LR_RNN_PROJ_BF16X3).  R = B*T rows, I input features, GH gate rows per direction, D
// directions; gates / dG keep the directions side by side in a row (leading dimensions ldgates, ldg).
// Every product packs all of its operand blocks in one launch (PackList).
size_t lr_xproj_workspace_bytes(int R, int I, int GH, int D, int H) {
  if (R <= 0 || I <= 0 || GH <= 0 || D <= 0 || H <= 0) return 0;
  size_t a = lr_xgemm_workspace_bytes(0, 1, R, D * GH, I);
  size_t b = lr_xgemm_workspace_bytes(1, 0, D * GH, I, R);
  if (b > a) a = b;
  b = lr_xgemm_workspace_bytes(0, 0, R, I, D * GH);
  if (b > a) a = b;
  // recurrent weight gradient: D batched products GH x H over K = R (lr_xproj_dwhh) and their split-K slabs
  b = ((size_t)D * (plane_floats(GH, R, false) + plane_floats(H, R, false)) +
       pad64((size_t)want_splits(GH, H * D, R) * GH * H * D)) * sizeof(float);
  if (b > a) a = b;
  return a;
}

// gates[R][D*GH] = x[R][I] . [W_ih[0]; W_ih[1]]^T + bias[D*GH]
// x_bf16: x is stored as a bf16 matrix [R][I] (the conv frontend's features): with I % 8 == 0 it IS
// the hi plane of the A operand and is not packed at all.
// one_product (LR_RNN_PROJ_BF16X1): every operand as its bf16 hi plane only — ONE product instead of two or three
int lr_xproj_forward(const float* x, int R, int I, const float* const* w_ih, int GH, int D, const float* bias,
                     float* gates, int x_exact, int x_bf16, void* workspace, size_t workspace_bytes,
                     hipStream_t stream, int one_product) {
  const int N = D * GH, ldp = ldp_of(I);
  if (1 + D > XPACK_MAX || (x_bf16 && !x_exact)) return LR_ERR_UNSUPPORTED;
  Planes pl;
  if (!carve(workspace, workspace_bytes, R, N, I, x_exact != 0 || one_product, one_product != 0, &pl)) return LR_ERR_WORKSPACE;
  PackList pk;
  const bool direct = x_bf16 && ldp == I;
  if (direct) pl.Ah = (bf16_t*)x;
  else pk.add(x, I, R, I, 0, pl.Ah, pl.Al, ldp, ldp, 0, 1, x_bf16);
  for (int d = 0; d < D; ++d)
    pk.add(w_ih[d], I, GH, I, 0, pl.Bh + (size_t)d * GH * ldp, pl.Bl ? pl.Bl + (size_t)d * GH * ldp : nullptr, ldp, ldp);
  int st = pk.launch(stream);
  if (st != LR_OK) return st;
  return contract(pl.Ah, pl.Al, pl.Bh, pl.Bl, R, N, I, 1.f, 0.f, gates, nullptr, R, N, bias, pl.slabs, pl.slab_floats,
                  stream);
}

// dW_ih[d][GH][I] (beta) = dG[:, d, :GH]^T . x   — rows d*GH.. of one (D*GH) x I product over K = R
int lr_xproj_dw(const float* dG, int ldg, int dstride, const float* x, int R, int I, int GH, int D,
                float* const* dw_ih, float beta, int x_exact, int x_bf16, void* workspace, size_t workspace_bytes,
                hipStream_t stream, int one_product) {
  const int M = D * GH, ldp = ldp_of(R);
  if (1 + D > XPACK_MAX) return LR_ERR_UNSUPPORTED;
  Planes pl;
  if (!carve(workspace, workspace_bytes, M, I, R, one_product != 0, x_exact != 0 || one_product, &pl)) return LR_ERR_WORKSPACE;
  PackList pk;
  for (int d = 0; d < D; ++d)
    pk.add(dG + (size_t)d * dstride, ldg, R, GH, 1, pl.Ah + (size_t)d * GH * ldp,
           pl.Al ? pl.Al + (size_t)d * GH * ldp : nullptr, ldp, ldp);
  pk.add(x, I, R, I, 1, pl.Bh, pl.Bl, ldp, ldp, 0, 1, x_bf16);
  int st = pk.launch(stream);
  if (st != LR_OK) return st;
  return contract(pl.Ah, pl.Al, pl.Bh, pl.Bl, M, I, R, 1.f, beta, dw_ih[0], D > 1 ? dw_ih[1] : nullptr, GH, I, nullptr,
                  pl.slabs, pl.slab_floats, stream);
}

// dx[R][I] = sum_d dG[:, d, :GH] . W_ih[d]   — one product over K = D*GH
int lr_xproj_dx(const float* dG, int ldg, int dstride, const float* const* w_ih, int R, int I, int GH, int D,
                float* dx, int hi_only, int dx_bf16, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const int K = D * GH, ldp = ldp_of(K);
  if (2 * D > XPACK_MAX) return LR_ERR_UNSUPPORTED;
  Planes pl;
  if (!carve(workspace, workspace_bytes, R, I, K, hi_only != 0, hi_only != 0, &pl)) return LR_ERR_WORKSPACE;
  PackList pk;
  for (int d = 0; d < D; ++d) {
    const int ow = d == D - 1 ? ldp - d * GH : GH;
    pk.add(dG + (size_t)d * dstride, ldg, R, GH, 0, pl.Ah + d * GH, pl.Al ? pl.Al + d * GH : nullptr, ldp, ow);
    pk.add(w_ih[d], I, GH, I, 1, pl.Bh + d * GH, pl.Bl ? pl.Bl + d * GH : nullptr, ldp, ow);
  }
  int st = pk.launch(stream);
  if (st != LR_OK) return st;
  return contract(pl.Ah, pl.Al, pl.Bh, pl.Bl, R, I, K, 1.f, 0.f, dx, nullptr, R, I, nullptr, pl.slabs, pl.slab_floats,
                  stream, 1, 0, 0, nullptr, dx_bf16);
}

// dW_hh[d][G*H][H] (beta) = dGh[:, d]^T . h_prev[:, d], h_prev[b,t] = y[b,t-1] (d = 0) / y[b,t+1] (d = 1), zero
// across sequence ends.  dG rows hold 4 slots of H per direction; the recurrent side reads slots
// (0, 1, 3) for the GRU (dr, dz, d(W_hn h + b_hn)) and (0..3) for the LSTM.  The directions are two
// batches of one launch (operands packed together, one contraction, one split-K combine).
int lr_xproj_dwhh(const float* dG, int ldg, const float* y, int ldy, int R, int T, int H, int G, int D,
                  float* const* dw_hh, float beta, void* workspace, size_t workspace_bytes, hipStream_t stream,
                  int one_product) {
  const int GH = G * H, ldp = ldp_of(R);
  if (D > 2 || 3 * D > XPACK_MAX) return LR_ERR_UNSUPPORTED;
  const size_t fa = plane_floats(GH, R, false), fb = plane_floats(H, R, false);
  const size_t avail = workspace_bytes / sizeof(float);
  if (avail < D * (fa + fb)) return LR_ERR_WORKSPACE;
  float* ws = (float*)workspace;
  bf16_t* A0 = (bf16_t*)ws;
  bf16_t* B0 = (bf16_t*)(ws + D * fa);
  float* slabs = ws + D * (fa + fb);
  const size_t slab_floats = avail - D * (fa + fb);
  PackList pk;
  for (int d = 0; d < D; ++d) {
    const float* g = dG + (size_t)d * 4 * H;
    bf16_t* Ah = A0 + (size_t)d * fa * 2;           // fa floats = 2 fa bf16 per batch
    bf16_t* Al = one_product ? nullptr : Ah + (size_t)GH * ldp;
    bf16_t* Bh = B0 + (size_t)d * fb * 2;
    bf16_t* Bl = one_product ? nullptr : Bh + (size_t)H * ldp;
    if (G == 3) {
      pk.add(g, ldg, R, 2 * H, 1, Ah, Al, ldp, ldp);
      pk.add(g + 3 * H, ldg, R, H, 1, Ah + (size_t)2 * H * ldp, Al ? Al + (size_t)2 * H * ldp : nullptr, ldp, ldp);
    } else {
      pk.add(g, ldg, R, GH, 1, Ah, Al, ldp, ldp);
    }
    pk.add(y + (size_t)d * H, ldy, R, H, 1, Bh, Bl, ldp, ldp, d == 0 ? -1 : 1, T);
  }
  int st = pk.launch(stream);
  if (st != LR_OK) return st;
  return contract(A0, one_product ? nullptr : A0 + (size_t)GH * ldp, B0, one_product ? nullptr : B0 + (size_t)H * ldp, GH, H,
                  R, 1.f, beta, dw_hh[0], nullptr, GH, H, nullptr, slabs, slab_floats, stream, D, (int64_t)fa * 2,
                  (int64_t)fb * 2, D > 1 ? dw_hh[1] : nullptr);
}

// dW_ih AND dW_hh of a layer whose recurrent side reads the same dG slots as its input side (LSTM: i, f, g, o; the
// tanh RNN) from ONE pack of dG: the transposed hi / lo planes of dG [D*GH][R] are the A operand of both products
// (lr_xproj_dw + lr_xproj_dwhh pack them twice: 37 us each at BiLSTM-768).  One pack launch (dG blocks, x, the
// time-shifted y blocks), two contractions.
size_t lr_xproj_dw_both_workspace_bytes(int R, int I, int GH, int H, int D) {
  if (R <= 0 || I <= 0 || GH <= 0 || H <= 0 || D <= 0) return 0;
  size_t slab = (size_t)want_splits(D * GH, I, R) * D * GH * I;
  const size_t s2 = (size_t)want_splits(GH, H * D, R) * GH * H * D;
  if (s2 > slab) slab = s2;
  return (plane_floats(D * GH, R, false) + plane_floats(I, R, false) + (size_t)D * plane_floats(H, R, false) + pad64(slab)) *
         sizeof(float);
}
int lr_xproj_dw_both(const float* dG, int ldg, int dstride, const float* x, const float* y, int ldy, int R, int T, int I,
                     int H, int GH, int D, float* const* dw_ih, float* const* dw_hh, float beta, void* workspace,
                     size_t workspace_bytes, hipStream_t stream) {
  const int M = D * GH, ldp = ldp_of(R);
  if (D > 2 || 2 * D + 1 > XPACK_MAX) return LR_ERR_UNSUPPORTED;
  const size_t fa = plane_floats(M, R, false), fx = plane_floats(I, R, false), fy = plane_floats(H, R, false);
  const size_t avail = workspace_bytes / sizeof(float);
  if (avail < fa + fx + D * fy) return LR_ERR_WORKSPACE;
  float* ws = (float*)workspace;
  bf16_t* Ah = (bf16_t*)ws;
  bf16_t* Al = Ah + (size_t)M * ldp;
  bf16_t* Xh = (bf16_t*)(ws + fa);
  bf16_t* Xl = Xh + (size_t)I * ldp;
  bf16_t* Y0 = (bf16_t*)(ws + fa + fx);
  float* slabs = ws + fa + fx + D * fy;
  const size_t slab_floats = avail - fa - fx - D * fy;
  PackList pk;
  for (int d = 0; d < D; ++d) {
    pk.add(dG + (size_t)d * dstride, ldg, R, GH, 1, Ah + (size_t)d * GH * ldp, Al + (size_t)d * GH * ldp, ldp, ldp);
    bf16_t* Yh = Y0 + (size_t)d * fy * 2;
    pk.add(y + (size_t)d * H, ldy, R, H, 1, Yh, Yh + (size_t)H * ldp, ldp, ldp, d == 0 ? -1 : 1, T);
  }
  pk.add(x, I, R, I, 1, Xh, Xl, ldp, ldp);
  int st = pk.launch(stream);
  if (st != LR_OK) return st;
  st = contract(Ah, Al, Xh, Xl, M, I, R, 1.f, beta, dw_ih[0], D > 1 ? dw_ih[1] : nullptr, GH, I, nullptr, slabs, slab_floats,
                stream);
  if (st != LR_OK) return st;
  return contract(Ah, Al, Y0, Y0 + (size_t)H * ldp, GH, H, R, 1.f, beta, dw_hh[0], nullptr, GH, H, nullptr, slabs, slab_floats,
                  stream, D, (int64_t)GH * ldp, (int64_t)fy * 2, D > 1 ? dw_hh[1] : nullptr);
}

extern "C" int lr_xgemm(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                        const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                        int a_exact, int b_exact, void* workspace, size_t workspace_bytes, lr_stream_t stream) {
  return lr_xgemm_impl(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, a_exact, b_exact,
                       workspace, workspace_bytes, (hipStream_t)stream);
}
