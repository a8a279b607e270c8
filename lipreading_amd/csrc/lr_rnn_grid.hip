// lr_rnn_grid.hip — the LSTM recurrence for 1152 < H <= 1536 as ONE launch per layer pass, fp32-faithful: the decoder
// RNN of the reference's dominant configuration family.  CharDecodingStep takes hidden = directions x encoder hidden
// (better_model.py:134-148,181): behind BiLSTM-768 (config/archive/experiments/e*/*/*: 52 of the reference's 56 sized flag
// files, batch 128) that is LSTM-1536, behind BiLSTM-700 LSTM-1400 — past lr_rnn_cluster.hip's 1152-unit ceiling, and
// until round 6 on the step kernels (62 launches per decoder loop, W_hh's 37.7 MB re-streamed by each).
//
// W_hh (4 x 1536 x 1536 as bf16 hi + lo planes: 37.7 MB) is spread over ONE grid of 24 x 8 = 192 compute units, 196 KB
// each, all of it in the registers of the member's four waves (48 MFMA fragments per wave, AGPRs) — see
// lr_rnn_grid_map.h for who holds what and why the split is two-dimensional (a 1-D split moves W_hh's own size in
// exchange words per step).  A launch serves up to 2 blocks of 32 samples (B = 128, the family's own batch: two launches): the
// samples are the N dimension of the MFMAs (two 16-sample sub-blocks per block), the weights the A operand, so a
// lane's four accumulator registers are the four gates of one (sample, unit) — one 16-byte exchange item.
//
// Per step, forward:  gather h[K_c] from the column group (23 x 1 KB per block, across XCDs) -> 192 MFMAs per wave and
// block (all four cross terms of (W_hi + W_lo)(h_hi + h_lo), fp32 accumulate) -> publish the partial gate sums to the
// seven row-group partners (4 KB each per block, inside the XCD) -> gather the seven addressed to this member, add its
// own, run the cell (nn.LSTM's arithmetic, better_model.py:47-49,74), publish h.  Backward mirrors it: partial dh
// reduce-scatter inside the column group, cell backward, dG all-gather inside the row group, W^T dG.  The exchange uses
// lr_rnn_xch.h's self-tagged words (fp32 rounded to 22 mantissa bits + a 2-bit step tag; two parity slots); waits are
// bounded and a member that gave up raises the device-side fault word, exactly as in lr_rnn_cluster.hip.
//
// Interface: the same buffers as the step kernels and the cluster kernels (gates in / out, extra = c, y, dG), reached
// through lr_rnn_cluster_* (lr_rnn_cluster.hip delegates the shapes this file covers).
#include "lr_common.h"
#include "lr_rnn_grid_map.h"
#include "lr_rnn_xch.h"
#include <hip/hip_ext.h>

namespace {

using namespace lrx;
using namespace lrg;

constexpr int SPIN_LIMIT = 1 << 18;
constexpr int HLD = KC + 8;   // bf16 per row of the state slice in LDS (100 dwords = 4 x odd: conflict-free ds_read_b128)
constexpr int GLD = GR + 8;   // ... of the row group's dG (132 dwords)

// weights as the A operand out of AGPRs, the state / dG as B
#define LRG_MFMA0(acc, w, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(b))
#define LRG_MFMA(acc, w, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b))

__device__ __forceinline__ void store4(u32* p, u32x4 w, bool local) {
  // workgroup scope (`sc0`) keeps the line in this XCD's L2 — only where the readers were verified to sit on it.
  // s_nop: a store of more than 8 bytes reads its data registers late; a VALU write to one of them in the next wait state
  // changes what some lanes store.  hipcc's hazard recogniser places the wait state behind stores it can see, not behind
  // an asm: the first cut of the backward kernel re-used the second data register in the very next instruction, and lanes
  // 12-15 of every row of 16 published an UNTAGGED second word (found with tools/probes/grid_xch_dump.cpp: the row
  // group waited for it until the time-out).
  if (local) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ u32x4 xword4(f32x4 v, u32 tg) {
  return (u32x4){xword(v[0], tg), xword(v[1], tg), xword(v[2], tg), xword(v[3], tg)};
}
__device__ __forceinline__ bool tags_ok(u32x4 v, u32 tg) {
  return (((v[0] ^ tg) | (v[1] ^ tg) | (v[2] ^ tg) | (v[3] ^ tg)) & 3u) == 0u;
}

// N 16-byte loads at base + i * stride (words), the ones whose bit is set in `want`; polled until every word carries
// `tg`.  Every load that is still missing a word is asked for again IN PARALLEL (lr_rnn_cluster.hip's scheme).  A thread
// that has given up (`bad`) issues nothing.
// `bad` collects WHICH wait gave up (the fault word is tested for != 0 everywhere; the bits are a diagnostic):
//   1 handshake | 2 forward h gather | 4 forward partial sums | 8 backward partial dh | 16 backward dG gather
template <int N>
__device__ __forceinline__ void gather(u32x4 (&g)[N], const u32* base, int stride, unsigned want, u32 tg, int& bad, int tune, int code) {
  for (int w = 0; w < (tune & 0xff); ++w) __builtin_amdgcn_s_sleep(1);   // (tuning knob: sleeps before the first poll)
#pragma unroll
  for (int i = 0; i < N; ++i) {
    g[i] = (u32x4){0u, 0u, 0u, 0u};
    if (((want >> i) & 1u) && !bad) g[i] = peek4(base + i * stride);
  }
  unsigned pend = want;
  for (int round = 0; pend && !bad; ++round) {
    LR_VM_DRAIN();
#pragma unroll
    for (int i = 0; i < N; ++i) LR_TOUCH(g[i]);
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (((pend >> i) & 1u) && tags_ok(g[i], tg)) pend &= ~(1u << i);
    if (!pend) break;
    if (round > SPIN_LIMIT) {
      bad |= code;
      break;
    }
    for (int w = 0; w < ((tune >> 8) & 0xff); ++w) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int i = 0; i < N; ++i)
      if ((pend >> i) & 1u) g[i] = peek4(base + i * stride);
  }
}

// -DLRG_TIMING (tools/build_variant.sh; never in the product build): thread 0 of every member stamps the shader clock at
// the phase boundaries of every step; tools/probes/grid_timing.py reads them back through lr_rnn_grid_debug_times
#ifdef LRG_TIMING
__device__ long long g_lrg_times[2][NM][64][8];
#define LRG_T(pass, k) do { if (threadIdx.x == 0 && s < 64) g_lrg_times[pass][blockIdx.x][s][k] = clock64(); } while (0)
#else
#define LRG_T(pass, k) do { } while (0)
#endif

// do the members first, first + stride, ... (count of them: this member's column group or row group) sit on one XCD?
// (a speed matter: see lr_rnn_cluster.hip xcd_handshake)
__device__ __forceinline__ void group_handshake(u32* xid, int m, int first, int stride, int count, int tid, int* s_local, int& bad) {
  if (tid == 0) {
    *s_local = 1;
    publish(xid + m, 0x100u | (u32)xcc_id(), false);
  }
  __syncthreads();
  if (tid < count) {
    u32 g = peek(xid + first + stride * tid);
    int n = 0;
    while (!(g & 0x100u) && n++ < SPIN_LIMIT) {
      __builtin_amdgcn_s_sleep(2);
      g = peek(xid + first + stride * tid);
    }
    if (!(g & 0x100u)) bad = 1;
    if ((int)(g & 0xf) != xcc_id() || bad) *s_local = 0;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// weight packing: W_hh [4H][H] fp32 of each direction -> the forward's and the backward's bf16 hi / lo fragments in the
// order a wave loads them (lr_rnn_grid_map.h), both planes from one read.  The same launch clears the exchange words
// of the first launch of each pass and folds the layer's biases for the input projection (b_ih + b_hh).
// ---------------------------------------------------------------------------------------------------------------
struct GridFold {
  const float* b_ih[2];
  const float* b_hh[2];
  float* out;   // [D][4H] or nullptr
};
__global__ void rnng_pack_kernel(const float* __restrict__ w0, const float* __restrict__ w1, bf16x8* __restrict__ out_f,
                                 bf16x8* __restrict__ out_b, int D, int H, u32* __restrict__ xch_f, long nzero_f,
                                 u32* __restrict__ xch_b, long nzero_b, GridFold fold) {
  const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long)gridDim.x * blockDim.x;
  if (xch_f)
    for (long i = gtid; i < nzero_f; i += gsz) xch_f[i] = 0u;
  if (xch_b)
    for (long i = gtid; i < nzero_b; i += gsz) xch_b[i] = 0u;
  if (fold.out)
    for (long i = gtid; i < (long)D * 4 * H; i += gsz) {
      const int d = (int)(i / (4 * H)), j = (int)(i - (long)d * 4 * H);
      fold.out[i] = fold.b_ih[d][j] + fold.b_hh[d][j];
    }
  // one thread per (direction, member, wave, tile, k step, lane) of each pass: eight elements, both planes
  const long per_dir = FRAGS_PER_DIR / 2;
  for (int pass = 0; pass < 2; ++pass) {
    bf16x8* out = pass ? out_b : out_f;
    if (!out) continue;
    for (long i = gtid; i < (long)D * per_dir; i += gsz) {
      const int d = (int)(i / per_dir);
      long j = i - (long)d * per_dir;
      const int lane = (int)(j & 63);
      j >>= 6;
      const int nq = pass ? BQ : FQ, nt = pass ? 3 : 4;
      const int q = (int)(j % nq);
      j /= nq;
      const int tt = (int)(j % nt);
      j /= nt;
      const int wave = (int)(j & 3), m = (int)(j >> 2);
      const int r = m / C, c = m % C;
      const float* w = d ? w1 : w0;
      bf16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int gate, uo, ui;
        if (pass) bwd_w_elem(r, c, 3 * wave + tt, q, lane, e, gate, uo, ui);
        else fwd_w_elem(r, c, 4 * wave + tt, q, lane, e, gate, uo, ui);
        const float v = (uo < H && ui < H) ? w[((long)gate * H + uo) * H + ui] : 0.f;
        bf16_t h16, l16;
        split_bf16(v, h16, l16);
        hi[e] = __builtin_bit_cast(__bf16, h16);
        lo[e] = __builtin_bit_cast(__bf16, l16);
      }
      const long o = (long)d * FRAGS_PER_DIR + (pass ? bwd_frag_index(m, wave, tt, q, 0, lane) : fwd_frag_index(m, wave, tt, q, 0, lane));
      out[o] = hi;
      out[o + 64] = lo;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward recurrence.  grid: 192 workgroups x 256 threads, block b = member (r = b / 8, c = b % 8) of direction d;
// samples b0 .. b0 + 32 NSB - 1.  Cell role of thread tid: item = tid = sample (tid >> 3) of each block, unit tid & 7.
// ---------------------------------------------------------------------------------------------------------------
template <int NSB>
__global__ __launch_bounds__(256, 1) void rnng_fwd_kernel(
    float* __restrict__ gates, float* __restrict__ extra, float* __restrict__ y, const bf16x8* __restrict__ wpk,
    const float* __restrict__ h0, const float* __restrict__ c0, const int32_t* __restrict__ lens, u32* __restrict__ xch,
    int32_t* __restrict__ fault, int drop, int tune, int tune2, int b0, int d, int B, int T, int D, int H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* hS = reinterpret_cast<bf16_t*>(smem);                                            // [NSB][2][32][HLD]
  float* Pown = reinterpret_cast<float*>(smem + (size_t)NSB * 2 * SB * HLD * 2);          // [NSB][256][4]
  __shared__ int s_local;
  // Placement, forward: the ROW group on one XCD.  The dispatcher places blocks x, x + 8, ... on XCD x (observed): block
  // 8 j + x is member (r = 3 x + j / 8, c = j % 8), so XCD x holds row groups 3 x .. 3 x + 2 whole.  What crosses XCDs is
  // then the h all-gather (1 KB written, 23 KB read per member and block) and what stays inside one is the partial-sum
  // exchange (28 KB written, 28 KB read).  The first cut had the column group on an XCD instead, as the backward does:
  // every member then wrote its 28 KB through to the fabric in the same microsecond (5.4 MB chip-wide per step), and the
  // partial-sum gather took 3.0 us of an 8.2 us step.
  const int r = 3 * (blockIdx.x & 7) + (blockIdx.x >> 6), c = (blockIdx.x >> 3) & 7, m = r * C + c;
  if (m == drop) return;   // test hook (lr_rnn_debug_drop_member): the others must time out and report
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;

  // ---- weights: 48 fragments per wave -----------------------------------------------------------------------------
  bf16x8 Wf[4][FQ][2];
  {
    const bf16x8* wsrc = wpk + (long)d * FRAGS_PER_DIR + fwd_frag_index(m, wave, 0, 0, 0, lane);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int q = 0; q < FQ; ++q)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) Wf[tt][q][pl] = wsrc[((tt * FQ + q) * 2 + pl) * 64];
  }
  // ---- the state before step 0 (zero, or h0): this column group's 192 units of every sample -----------------------
  for (int i = tid; i < NSB * 2 * SB * HLD; i += 256) hS[i] = 0;
  __syncthreads();
  if (h0)
    for (int i = tid; i < NSB * SB * KC; i += 256) {
      const int sb = i / (SB * KC), rem = i - sb * (SB * KC), s_ = rem / KC, kk = rem - s_ * KC;
      const int unit = kc_unit(c, kk), bb = b0 + SB * sb + s_;
      if (unit < H && bb < B) {
        bf16_t hi, lo;
        split_bf16(h0[((long)d * B + bb) * H + unit], hi, lo);
        hS[((sb * 2 + 0) * SB + s_) * HLD + kk] = hi;
        hS[((sb * 2 + 1) * SB + s_) * HLD + kk] = lo;
      }
    }

  // ---- cell role ---------------------------------------------------------------------------------------------------
  const int s_ = tid >> 3, u8 = tid & 7;
  const int unit = own_unit(r, c, u8);
  bool alive[NSB];
  int len[NSB];
  float creg[NSB];
  struct Gx { float v[4]; };
  Gx gx[NSB];
  auto time_of = [&](int s) {
    const int sc = s < T ? s : T - 1;
    return d == 0 ? sc : T - 1 - sc;
  };
  auto fetch_gx = [&](int t) {
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      if (!alive[sb]) continue;
      const int bb = b0 + SB * sb + s_;
      const float* gp = gates + (((long)bb * T + t) * D + d) * (long)(4 * H) + unit;
#pragma unroll
      for (int g = 0; g < 4; ++g) gx[sb].v[g] = gp[(long)g * H];
    }
  };
#pragma unroll
  for (int sb = 0; sb < NSB; ++sb) {
    const int bb = b0 + SB * sb + s_;
    alive[sb] = bb < B && unit < H;
    len[sb] = alive[sb] ? lens[bb] : 0;
    creg[sb] = (alive[sb] && c0) ? c0[((long)d * B + bb) * H + unit] : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) gx[sb].v[g] = 0.f;
  }
  fetch_gx(time_of(0));

  // ---- exchange ----------------------------------------------------------------------------------------------------
  u32* HX = xch;
  u32* PX = xch + hx_words(NSB);
  u32* XID = PX + px_words(NSB);
  const int hslot = (int)hx_index(NSB, 1, 0, 0, 0), pslot = (int)px_index(NSB, 1, 0, 0, 0, 0);   // words between the parity slots
  u32* hmine = HX + hx_index(NSB, 0, c, 0, r) + tid;                  // + sb * R * 256
  const u32* hin = HX + hx_index(NSB, 0, c, 0, 0) + 4 * tid;          // load i: + i * 1024: item 256 i + tid of the column group's sweep
  u32* pout = PX + px_index(NSB, 0, r, 0, 0, c);                      // + (cd * NSB + sb) * C * 1024 + item * 4
  const u32* pin = PX + px_index(NSB, 0, r, c, 0, 0) + 4 * tid;       // load i = sb * 8 + cs: + i * 1024
  // the gather sweep of h: load i covers block sb = i / 6, source row 4 (i % 6) + wave, item lane
  unsigned hwant = 0, pwant = 0;
#pragma unroll
  for (int i = 0; i < 6 * NSB; ++i)
    if (4 * (i % 6) + wave != r) hwant |= 1u << i;
#pragma unroll
  for (int i = 0; i < 8 * NSB; ++i)
    if ((i & 7) != c) pwant |= 1u << i;
  int bad = 0;
  group_handshake(XID, m, r * C, 1, C, tid, &s_local, bad);   // the row group's 8 members (also: hS complete)
  const bool local = s_local != 0;

  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the loads in front of the loop (lr_rnn_cluster.hip explains)
  for (int s = 0; s < T; ++s) {
    const int t = time_of(s);
    int hoff = __builtin_amdgcn_readfirstlane(((s - 1) & 1) * hslot), poff = __builtin_amdgcn_readfirstlane((s & 1) * pslot);
    asm volatile("" : "+s"(hoff), "+s"(poff));
    LRG_T(0, 0);
    // ---- (1) the column group's h_{s-1}: 23 x 64 items of four units per block -> hS ---------------------------------
    if (s > 0) {
      u32x4 g[6 * NSB];
      gather<6 * NSB>(g, hin + hoff, 1024, hwant, tag_of(s - 1), bad, tune, 2);
#pragma unroll
      for (int i = 0; i < 6 * NSB; ++i) {
        if (!((hwant >> i) & 1u)) continue;
        const int sb = i / 6, rs = 4 * (i % 6) + wave;
        u32 hi0, lo0, hi1, lo1;
        split_bf16_pair(xval(g[i][0]), xval(g[i][1]), hi0, lo0);
        split_bf16_pair(xval(g[i][2]), xval(g[i][3]), hi1, lo1);
        bf16_t* dst = hS + ((sb * 2) * SB + h_gather_sample(lane)) * HLD + h_gather_kk(rs, lane);
        *reinterpret_cast<uint2*>(dst) = make_uint2(hi0, hi1);
        *reinterpret_cast<uint2*>(dst + SB * HLD) = make_uint2(lo0, lo1);
      }
    }
    LRG_T(0, 1);
    lr_lds_barrier();   // hS complete (the own member's columns were written by the cell of step s - 1)
    LRG_T(0, 2);
    // ---- (2) the product, block by block; (3) its partial sums go out as soon as a block is done -----------------------
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      f32x4 acc[4][2];
      const bf16_t* hb = hS + (sb * 2) * SB * HLD + col * HLD + kg * 8;
      // the B operands of k step q + 1 are read while the 32 MFMAs of k step q issue (left to itself hipcc re-used one
      // register set and put every k step's four ds_read_b128 in front of its MFMAs: their latency six times per block,
      // a third of the product's time in the first cut)
      bf16x8 bv[2][2], bn[2][2];   // [plane][sub-block]
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int sbb = 0; sbb < 2; ++sbb) bv[pl][sbb] = *reinterpret_cast<const bf16x8*>(hb + (pl * SB + 16 * sbb) * HLD);
#pragma unroll
      for (int q = 0; q < FQ; ++q) {
        if (q + 1 < FQ) {
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int sbb = 0; sbb < 2; ++sbb)
              bn[pl][sbb] = *reinterpret_cast<const bf16x8*>(hb + (pl * SB + 16 * sbb) * HLD + 32 * (q + 1));
        }
#pragma unroll
        for (int wp = 0; wp < 2; ++wp)
#pragma unroll
          for (int hp = 0; hp < 2; ++hp)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
              for (int sbb = 0; sbb < 2; ++sbb) {
                if (q == 0 && wp == 0 && hp == 0) LRG_MFMA0(acc[tt][sbb], Wf[tt][0][0], bv[0][sbb]);
                else LRG_MFMA(acc[tt][sbb], Wf[tt][q][wp], bv[hp][sbb]);
              }
        if (q + 1 < FQ) {
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int sbb = 0; sbb < 2; ++sbb) bv[pl][sbb] = bn[pl][sbb];
        }
      }
      LR_MFMA_DRAIN();
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int sbb = 0; sbb < 2; ++sbb) LR_ACC_READY(acc[tt][sbb]);
      const u32 tg = tag_of(s);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int sbb = 0; sbb < 2; ++sbb) {
          int cd, item;
          fwd_acc_dest(4 * wave + tt, sbb, lane, cd, item);   // (cd is the same for the whole wave)
          if (cd == c) *reinterpret_cast<f32x4*>(Pown + (sb * 256 + item) * 4) = acc[tt][sbb];
          else store4(pout + poff + (cd * NSB + sb) * (C * 1024) + item * 4, xword4(acc[tt][sbb], tg), local);
        }
    }
    LRG_T(0, 3);
    // ---- (4) the seven partial sums addressed to this member, its own, the cell ----------------------------------------
    {
      u32x4 g[8 * NSB];
      gather<8 * NSB>(g, pin + poff, 1024, pwant, tag_of(s), bad, tune2, 4);
      LRG_T(0, 4);
      lr_lds_barrier();   // Pown complete; every wave is done with hS
      LRG_T(0, 5);
      const int tnext = time_of(s + 1);
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
        const f32x4 own = *reinterpret_cast<const f32x4*>(Pown + (sb * 256 + tid) * 4);
#pragma unroll
        for (int cs = 0; cs < C; ++cs) {   // fixed order: column group 0 .. 7
#pragma unroll
          for (int k = 0; k < 4; ++k) sum[k] += cs == c ? own[k] : xval(g[sb * 8 + cs][k]);
        }
        const bool live = alive[sb] && t < len[sb];
        // nn.LSTM's cell, torch gate order i, f, g, o (lr_rnn_cluster.hip's arithmetic)
        const float ig = fast_sigmoid(gx[sb].v[0] + sum[0]);
        const float fg = fast_sigmoid(gx[sb].v[1] + sum[1]);
        const float gg = fast_tanh(gx[sb].v[2] + sum[2]);
        const float og = fast_sigmoid(gx[sb].v[3] + sum[3]);
        const float cn = live ? fg * creg[sb] + ig * gg : 0.f;
        float h = live ? og * fast_tanh(cn) : 0.f;
        creg[sb] = cn;
        const u32 w = xword(h, tag_of(s));
        {   // first: the column group (on the other XCDs) is waiting for it.  Four units of a sample sit in the four lanes
            // of a quad: the quad's first lane collects them (DPP quad_perm) and writes ONE 16-byte item — a scalar
            // write-through store is one fabric write each (MI355X_MICROARCH.md: a dword costs ~6x a dwordx4 per byte)
          const u32x4 w4 = {w, (u32)__builtin_amdgcn_mov_dpp((int)w, 0x55, 0xf, 0xf, true), (u32)__builtin_amdgcn_mov_dpp((int)w, 0xaa, 0xf, 0xf, true),
                            (u32)__builtin_amdgcn_mov_dpp((int)w, 0xff, 0xf, 0xf, true)};
          if ((tid & 3) == 0) store4(hmine + (s & 1) * hslot + sb * (R * 256), w4, false);
        }
        h = xval(w);                                                   // the state everyone uses, this member included
        bf16_t hi, lo;
        split_bf16(h, hi, lo);
        hS[((sb * 2 + 0) * SB + s_) * HLD + 8 * r + u8] = hi;
        hS[((sb * 2 + 1) * SB + s_) * HLD + 8 * r + u8] = lo;
        if (alive[sb]) {
          const long bt = (long)(b0 + SB * sb + s_) * T + t;
          y[bt * ((long)D * H) + d * H + unit] = h;
          extra[(bt * D + d) * H + unit] = cn;
          if (live) {
            float* gout = gates + (bt * D + d) * (long)(4 * H) + unit;
            gout[0] = ig;
            gout[(long)H] = fg;
            gout[(long)2 * H] = gg;
            gout[(long)3 * H] = og;
          }
        }
      }
      if (s + 1 < T) fetch_gx(tnext);
      LRG_T(0, 6);
    }
  }
  if (bad && fault) atomicOr(fault, bad);
}

// ---------------------------------------------------------------------------------------------------------------
// backward recurrence (rnn_bwd_step_kernel<4>'s arithmetic per (sample, unit); lr_rnn_cluster.hip rnnc_bwd_kernel)
// ---------------------------------------------------------------------------------------------------------------
template <int NSB>
__global__ __launch_bounds__(256, 1) void rnng_bwd_kernel(
    const float* __restrict__ gates, const float* __restrict__ extra, const float* __restrict__ dy,
    const float* __restrict__ dh_n, const float* __restrict__ dc_n, float* __restrict__ dG, float* __restrict__ dh0,
    float* __restrict__ dc0, const float* __restrict__ c0, const bf16x8* __restrict__ wpk,
    const int32_t* __restrict__ lens, u32* __restrict__ xch, int32_t* __restrict__ fault, int drop, int tune, int tune2, int b0,
    int d, int B, int T, int D, int H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* gS = reinterpret_cast<bf16_t*>(smem);                                                   // [NSB][2][32][GLD]
  float* red = reinterpret_cast<float*>(smem + (size_t)NSB * 2 * SB * GLD * 2);                  // [NSB][4][256]
  float* ownD = red + NSB * 4 * 256;                                                             // [NSB][256]
  __shared__ int s_local;
  // Placement, backward: the COLUMN group on one XCD (block b = member (r = b / 8, c = b % 8) on XCD b % 8): the partial dh
  // reduce-scatter (23 KB written and read per member and block) stays inside it, the dG all-gather (4 KB written, 28 KB
  // read) crosses — the light writer crosses in both passes.
  const int m = blockIdx.x, c = m % C, r = m / C;
  if (m == drop) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const long DH = (long)D * H;

  bf16x8 Wb[3][BQ][2];
  {
    const bf16x8* wsrc = wpk + (long)d * FRAGS_PER_DIR + bwd_frag_index(m, wave, 0, 0, 0, lane);
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
#pragma unroll
      for (int q = 0; q < BQ; ++q)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) Wb[jj][q][pl] = wsrc[((jj * BQ + q) * 2 + pl) * 64];
  }
  for (int i = tid; i < NSB * 2 * SB * GLD; i += 256) gS[i] = 0;
  for (int i = tid; i < NSB * 256; i += 256) ownD[i] = 0.f;

  const int s_ = tid >> 3, u8 = tid & 7;
  const int unit = own_unit(r, c, u8);
  const int ridx = (s_ * 2 + (u8 >> 2)) * 4 + (u8 & 3);   // this (sample, unit) in a 1 KB dh block
  bool alive[NSB];
  int len[NSB];
  float car[NSB], inj_h[NSB], inj_c[NSB];
  struct In { float dy, g[4], ex, prev; };
  In in[NSB];
  auto time_of = [&](int s) {
    const int sc = s < T ? s : T - 1;
    return d == 0 ? T - 1 - sc : sc;
  };
  auto fetch = [&](int t) {
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      in[sb].dy = in[sb].ex = in[sb].prev = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) in[sb].g[g] = 0.f;
      if (!alive[sb]) continue;
      const int bb = b0 + SB * sb + s_;
      const int tp = d == 0 ? t - 1 : t + 1;
      const long bt = (long)bb * T + t;
      in[sb].dy = dy[bt * DH + d * H + unit];
      const float* gi = gates + (bt * D + d) * (long)(4 * H) + unit;
#pragma unroll
      for (int g = 0; g < 4; ++g) in[sb].g[g] = gi[(long)g * H];
      in[sb].ex = extra[(bt * D + d) * H + unit];
      if (tp >= 0 && tp < T) in[sb].prev = extra[(((long)bb * T + tp) * D + d) * H + unit];
      else if (c0) in[sb].prev = c0[((long)d * B + bb) * H + unit];   // the state before the first step (decoder loop), else zero
    }
  };
#pragma unroll
  for (int sb = 0; sb < NSB; ++sb) {
    const int bb = b0 + SB * sb + s_;
    alive[sb] = bb < B && unit < H;
    len[sb] = alive[sb] ? lens[bb] : 0;
    car[sb] = 0.f;
    inj_h[sb] = (alive[sb] && dh_n) ? dh_n[((long)d * B + bb) * H + unit] : 0.f;
    inj_c[sb] = (alive[sb] && dc_n) ? dc_n[((long)d * B + bb) * H + unit] : 0.f;
  }
  fetch(time_of(0));

  u32* GX = xch;
  u32* DX = xch + gx_words(NSB);
  u32* XID = DX + dx_words(NSB);
  const int gslot = (int)gx_index(NSB, 1, 0, 0, 0), dslot = (int)dx_index(NSB, 1, 0, 0, 0, 0);
  u32* gmine = GX + gx_index(NSB, 0, r, 0, c) + 4 * tid;               // + sb * C * 1024
  const u32* gin = GX + gx_index(NSB, 0, r, 0, 0) + 4 * tid;           // load i = sb * 8 + cs: + i * 1024
  u32* dout = DX + dx_index(NSB, 0, c, 0, 0, r);                       // + (rd * NSB + sb) * R * 256 + item * 4
  const u32* din = DX + dx_index(NSB, 0, c, r, 0, 0) + 4 * tid;        // load i: + i * 1024 (block i / 6, source row 4 (i % 6) + wave)
  unsigned dwant = 0, gwant = 0;
#pragma unroll
  for (int i = 0; i < 6 * NSB; ++i)
    if (4 * (i % 6) + wave != r) dwant |= 1u << i;
#pragma unroll
  for (int i = 0; i < 8 * NSB; ++i)
    if ((i & 7) != c) gwant |= 1u << i;
  int bad = 0;
  group_handshake(XID, m, c, C, R, tid, &s_local, bad);   // the column group's 24 members (also: gS / ownD cleared)
  const bool local = s_local != 0;

  const int nsteps = dh0 ? T + 1 : T;   // with dh0: one more reduce-scatter after the last step
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int s = 0; s < nsteps; ++s) {
    const int t = time_of(s);
    int doff_in = __builtin_amdgcn_readfirstlane(((s - 1) & 1) * dslot), goff = __builtin_amdgcn_readfirstlane((s & 1) * gslot),
        doff_out = __builtin_amdgcn_readfirstlane((s & 1) * dslot);
    asm volatile("" : "+s"(doff_in), "+s"(goff), "+s"(doff_out));
    LRG_T(1, 0);
    // ---- (1) W_hh^T dG of the step before, for this member's units: 23 partial sums per block + its own ---------------
    float prod[NSB];
    if (s > 0) {
      u32x4 g[6 * NSB];
      gather<6 * NSB>(g, din + doff_in, 1024, dwant, tag_of(s - 1), bad, tune, 8);
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;   // this thread's sources of the block, in FIXED order
        if (!bad) {
#pragma unroll
          for (int i = 6 * sb; i < 6 * sb + 6; ++i)
            if ((dwant >> i) & 1u) {
              p0 += xval(g[i][0]);
              p1 += xval(g[i][1]);
              p2 += xval(g[i][2]);
              p3 += xval(g[i][3]);
            }
        }
        *reinterpret_cast<float4*>(red + (sb * 4 + wave) * 256 + 4 * lane) = make_float4(p0, p1, p2, p3);
      }
    }
    LRG_T(1, 1);
    lr_lds_barrier();   // `red`, ownD complete; every wave is done with gS
    LRG_T(1, 2);
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      prod[sb] = 0.f;
      if (s > 0) {
        prod[sb] = ownD[sb * 256 + ridx];
#pragma unroll
        for (int w = 0; w < 4; ++w) prod[sb] += red[(sb * 4 + w) * 256 + ridx];
      }
    }
    if (s == T) {   // past the last step (only with dh0): the gradient into the initial state (lr_rnn_dh0's arithmetic)
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb)
        if (alive[sb]) {
          const long o = ((long)d * B + b0 + SB * sb + s_) * H + unit;
          dh0[o] = prod[sb];
          if (dc0) dc0[o] = car[sb];
        }
      break;
    }
    // ---- (2) the cell backward; dG goes to memory, to the row group, and (own) into gS ----------------------------------
    const u32 tg = tag_of(s);
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      const In& q_ = in[sb];
      float dh = q_.dy + prod[sb];
      const bool is_last = d == 0 ? (t == len[sb] - 1) : (t == 0);   // where the final state was read
      if (is_last) dh += inj_h[sb];
      const bool live = alive[sb] && t < len[sb];
      const float ig = q_.g[0], fg = q_.g[1], gg = q_.g[2], og = q_.g[3], ct = q_.ex, cp = q_.prev;
      float dc = car[sb];
      if (is_last) dc += inj_c[sb];
      f32x4 dg = {0.f, 0.f, 0.f, 0.f};
      car[sb] = 0.f;
      if (live) {
        const float tc = fast_tanh(ct);
        dc += dh * og * (1.f - tc * tc);
        dg[0] = dc * gg * ig * (1.f - ig);
        dg[1] = dc * cp * fg * (1.f - fg);
        dg[2] = dc * ig * (1.f - gg * gg);
        dg[3] = dh * tc * og * (1.f - og);
        car[sb] = dc * fg;
      }
      const u32x4 w = xword4(dg, tg);
      store4(gmine + goff + sb * (C * 1024), w, false);   // first: the row group is waiting for it
      u32 hi0, lo0, hi1, lo1;   // what everyone contracts: the rounded values, this member included
      split_bf16_pair(xval(w[0]), xval(w[1]), hi0, lo0);
      split_bf16_pair(xval(w[2]), xval(w[3]), hi1, lo1);
      bf16_t* dst = gS + ((sb * 2) * SB + s_) * GLD + dg_kidx(c, u8, 0);
      *reinterpret_cast<uint2*>(dst) = make_uint2(hi0, hi1);
      *reinterpret_cast<uint2*>(dst + SB * GLD) = make_uint2(lo0, lo1);
      if (alive[sb]) {
        float* dgo = dG + (((long)(b0 + SB * sb + s_) * T + t) * D + d) * (long)(4 * H) + unit;
#pragma unroll
        for (int k = 0; k < 4; ++k) dgo[(long)k * H] = dg[k];
      }
    }
    if (s + 1 < T) fetch(time_of(s + 1));
    LRG_T(1, 3);
    // ---- (3) the row group's dG -> gS ---------------------------------------------------------------------------------
    {
      u32x4 g[8 * NSB];
      gather<8 * NSB>(g, gin + goff, 1024, gwant, tg, bad, tune2, 16);
#pragma unroll
      for (int i = 0; i < 8 * NSB; ++i) {
        if (!((gwant >> i) & 1u)) continue;
        const int sb = i >> 3, cs = i & 7;
        u32 hi0, lo0, hi1, lo1;
        split_bf16_pair(xval(g[i][0]), xval(g[i][1]), hi0, lo0);
        split_bf16_pair(xval(g[i][2]), xval(g[i][3]), hi1, lo1);
        bf16_t* dst = gS + ((sb * 2) * SB + s_) * GLD + dg_kidx(cs, u8, 0);
        *reinterpret_cast<uint2*>(dst) = make_uint2(hi0, hi1);
        *reinterpret_cast<uint2*>(dst + SB * GLD) = make_uint2(lo0, lo1);
      }
    }
    LRG_T(1, 4);
    lr_lds_barrier();   // gS complete; `red` and ownD free again
    LRG_T(1, 5);
    // ---- (4) partial dh of this column group's 192 units, published to their owners ------------------------------------
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      f32x4 acc[3][2];
      const bf16_t* gb = gS + (sb * 2) * SB * GLD + col * GLD + kg * 8;
      bf16x8 bv[2][2], bn[2][2];   // (operands of k step q + 1 read under the MFMAs of k step q: see the forward kernel)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int sbb = 0; sbb < 2; ++sbb) bv[pl][sbb] = *reinterpret_cast<const bf16x8*>(gb + (pl * SB + 16 * sbb) * GLD);
#pragma unroll
      for (int q = 0; q < BQ; ++q) {
        if (q + 1 < BQ) {
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int sbb = 0; sbb < 2; ++sbb)
              bn[pl][sbb] = *reinterpret_cast<const bf16x8*>(gb + (pl * SB + 16 * sbb) * GLD + 32 * (q + 1));
        }
#pragma unroll
        for (int wp = 0; wp < 2; ++wp)
#pragma unroll
          for (int hp = 0; hp < 2; ++hp)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
#pragma unroll
              for (int sbb = 0; sbb < 2; ++sbb) {
                if (q == 0 && wp == 0 && hp == 0) LRG_MFMA0(acc[jj][sbb], Wb[jj][0][0], bv[0][sbb]);
                else LRG_MFMA(acc[jj][sbb], Wb[jj][q][wp], bv[hp][sbb]);
              }
        if (q + 1 < BQ) {
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int sbb = 0; sbb < 2; ++sbb) bv[pl][sbb] = bn[pl][sbb];
        }
      }
      LR_MFMA_DRAIN();
#pragma unroll
      for (int jj = 0; jj < 3; ++jj)
#pragma unroll
        for (int sbb = 0; sbb < 2; ++sbb) LR_ACC_READY(acc[jj][sbb]);
#pragma unroll
      for (int jj = 0; jj < 3; ++jj)
#pragma unroll
        for (int sbb = 0; sbb < 2; ++sbb) {
          int rd, item;
          bwd_acc_dest(3 * wave + jj, sbb, lane, rd, item);
          if (rd == r) *reinterpret_cast<f32x4*>(ownD + sb * 256 + item * 4) = acc[jj][sbb];
          else store4(dout + doff_out + (rd * NSB + sb) * (R * 256) + item * 4, xword4(acc[jj][sbb], tg), local);
        }
    }
    LRG_T(1, 6);
  }
  if (bad && fault) atomicOr(fault, bad);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// sample blocks per launch: 1 or 2 (64 samples).  Four blocks fit the LDS (static_assert below) but not the registers as
// the step is written — the 24 + 32 landing registers per block of the two gathers spill past two blocks (hipcc: 332 / 724
// bytes of scratch per lane); B = 128 runs as two launches of 64.
constexpr int MAXSB = 2;
inline int nsb_of(int samples) { return samples <= SB ? 1 : 2; }
template <int NSB> constexpr size_t fwd_lds() { return (size_t)NSB * 2 * SB * HLD * 2 + (size_t)NSB * 256 * 4 * 4; }
template <int NSB> constexpr size_t bwd_lds() { return (size_t)NSB * 2 * SB * GLD * 2 + (size_t)NSB * 4 * 256 * 4 + (size_t)NSB * 256 * 4; }
static_assert(fwd_lds<4>() <= 160 * 1024 && bwd_lds<4>() <= 160 * 1024, "(four sample blocks would fit one compute unit's LDS)");

template <int NSB>
int fwd_launch1(float* gates, float* extra, float* y, const void* wpack, const float* h0, const float* c0, const int32_t* lens,
                void* xch, int b0, int d, int B, int T, int D, int H, hipStream_t stream, bool prof) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rnng_fwd_kernel<NSB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds<NSB>()) != hipSuccess)
      return LR_ERR_LAUNCH;
    attr_set = true;
  }
  int32_t* fault = lr_fault_words();
  const int drop = lr_debug_drop_member_value(), tune = lr_debug_tune_value(2), tune2 = lr_debug_tune_value(3);
  hipEvent_t e0, e1;
  if (prof && lr_prof_next(LR_PROF_RNN_FWD, &e0, &e1))
    hipExtLaunchKernelGGL((rnng_fwd_kernel<NSB>), dim3(NM), dim3(256), fwd_lds<NSB>(), stream, e0, e1, 0, gates, extra, y,
                          (const bf16x8*)wpack, h0, c0, lens, (u32*)xch, fault, drop, tune, tune2, b0, d, B, T, D, H);
  else
    hipLaunchKernelGGL((rnng_fwd_kernel<NSB>), dim3(NM), dim3(256), fwd_lds<NSB>(), stream, gates, extra, y, (const bf16x8*)wpack,
                       h0, c0, lens, (u32*)xch, fault, drop, tune, tune2, b0, d, B, T, D, H);
  return lr_launch_status();
}

template <int NSB>
int bwd_launch1(const float* gates, const float* extra, const float* dy, const float* dh_n, const float* dc_n, float* dG,
                float* dh0, float* dc0, const float* c0, const void* wpack, const int32_t* lens, void* xch, int b0, int d, int B,
                int T, int D, int H, hipStream_t stream, bool prof) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rnng_bwd_kernel<NSB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds<NSB>()) != hipSuccess)
      return LR_ERR_LAUNCH;
    attr_set = true;
  }
  int32_t* fault = lr_fault_words();
  const int drop = lr_debug_drop_member_value(), tune = lr_debug_tune_value(4), tune2 = lr_debug_tune_value(5);
  hipEvent_t e0, e1;
  if (prof && lr_prof_next(LR_PROF_RNN_BWD, &e0, &e1))
    hipExtLaunchKernelGGL((rnng_bwd_kernel<NSB>), dim3(NM), dim3(256), bwd_lds<NSB>(), stream, e0, e1, 0, gates, extra, dy, dh_n,
                          dc_n, dG, dh0, dc0, c0, (const bf16x8*)wpack, lens, (u32*)xch, fault, drop, tune, tune2, b0, d, B, T, D, H);
  else
    hipLaunchKernelGGL((rnng_bwd_kernel<NSB>), dim3(NM), dim3(256), bwd_lds<NSB>(), stream, gates, extra, dy, dh_n, dc_n, dG, dh0,
                       dc0, c0, (const bf16x8*)wpack, lens, (u32*)xch, fault, drop, tune, tune2, b0, d, B, T, D, H);
  return lr_launch_status();
}

}  // namespace

#ifdef LRG_TIMING
extern "C" int lr_rnn_grid_debug_times(long long* out_host) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_lrg_times), sizeof(long long) * 2 * NM * 64 * 8) == hipSuccess ? 0 : -1;
}
#endif

// shapes this file covers: LSTM (G = 4), 1152 < H <= 1536
int lr_rnn_grid_shape(int G, int H) { return G == 4 && H > 1152 && H <= HP && H % 4 == 0 ? 1 : 0; }
int lr_rnn_grid_cus() { return NM; }
// launches per layer pass: every direction on its own, 64 samples per launch
int lr_rnn_grid_launches(int B, int D) { return D * ((B + MAXSB * SB - 1) / (MAXSB * SB)); }
size_t lr_rnn_grid_pack_bytes(int D) { return (size_t)D * FRAGS_PER_DIR * sizeof(bf16x8); }
size_t lr_rnn_grid_xch_bytes(int B, int backward) {
  return (size_t)xch_words(nsb_of(B), backward) * sizeof(u32);
}

// W_hh -> fragments of the forward (wpack) and, where wpack_b is given, of the backward; the exchange words of each
// pass's first launch cleared; biases folded (b_ih may be NULL)
int lr_rnn_grid_prologue(const float* const* w_hh, const float* const* b_ih, const float* const* b_hh, float* bias_out,
                         void* wpack, void* xch, int B, int D, int H, hipStream_t stream, void* wpack_b, void* xch_b) {
  GridFold fold;
  for (int d = 0; d < 2; ++d) {
    fold.b_ih[d] = b_ih ? b_ih[d < D ? d : 0] : nullptr;
    fold.b_hh[d] = b_hh ? b_hh[d < D ? d : 0] : nullptr;
  }
  fold.out = b_ih ? bias_out : nullptr;
  const int nsb = nsb_of(B);
  LR_LAUNCH(rnng_pack_kernel, dim3(2048), dim3(256), 0, stream, w_hh[0], w_hh[D - 1], (bf16x8*)wpack, (bf16x8*)wpack_b, D, H,
            (u32*)xch, wpack ? xch_words(nsb, 0) : 0L, (u32*)xch_b, wpack_b ? xch_words(nsb, 1) : 0L, fold);
  return lr_launch_status();
}

int lr_rnn_grid_forward(float* gates, float* extra, float* y, const float* const* w_hh, const float* h0, const float* c0,
                        const int32_t* lens, void* wpack, void* xch, int B, int T, int D, int H, hipStream_t stream,
                        int prologue_done) {
  lr_clear_error();
  int st = LR_OK;
  if (!prologue_done) {
    st = lr_rnn_grid_prologue(w_hh, nullptr, nullptr, nullptr, wpack, xch, B, D, H, stream, nullptr, nullptr);
    if (st != LR_OK) return st;
  }
  bool first = true;
  for (int d = 0; d < D; ++d)
    for (int b0 = 0; b0 < B; b0 += MAXSB * SB) {
      const int nsb = nsb_of(B - b0);
      // (the first launch's words were cleared by the prologue)
      if (!first && hipMemsetAsync(xch, 0, (size_t)xch_words(nsb, 0) * sizeof(u32), stream) != hipSuccess) return LR_ERR_LAUNCH;
      if (nsb == 1) st = fwd_launch1<1>(gates, extra, y, wpack, h0, c0, lens, xch, b0, d, B, T, D, H, stream, first);
      else st = fwd_launch1<2>(gates, extra, y, wpack, h0, c0, lens, xch, b0, d, B, T, D, H, stream, first);
      if (st != LR_OK) return st;
      first = false;
    }
  return LR_OK;
}

int lr_rnn_grid_backward(const float* gates, const float* extra, const float* dy, const float* dh_n, const float* dc_n,
                         float* dG, float* dh0, float* dc0, const float* c0, const float* const* w_hh, const int32_t* lens,
                         void* wpack, void* xch, int B, int T, int D, int H, hipStream_t stream, int pack_done) {
  lr_clear_error();
  int st = LR_OK;
  if (!pack_done) {
    st = lr_rnn_grid_prologue(w_hh, nullptr, nullptr, nullptr, nullptr, nullptr, B, D, H, stream, wpack, xch);
    if (st != LR_OK) return st;
  }
  bool first = true;
  for (int d = 0; d < D; ++d)
    for (int b0 = 0; b0 < B; b0 += MAXSB * SB) {
      const int nsb = nsb_of(B - b0);
      if (!first && hipMemsetAsync(xch, 0, (size_t)xch_words(nsb, 1) * sizeof(u32), stream) != hipSuccess) return LR_ERR_LAUNCH;
      if (nsb == 1) st = bwd_launch1<1>(gates, extra, dy, dh_n, dc_n, dG, dh0, dc0, c0, wpack, lens, xch, b0, d, B, T, D, H, stream, first);
      else st = bwd_launch1<2>(gates, extra, dy, dh_n, dc_n, dG, dh0, dc0, c0, wpack, lens, xch, b0, d, B, T, D, H, stream, first);
      if (st != LR_OK) return st;
      first = false;
    }
  return LR_OK;
}
