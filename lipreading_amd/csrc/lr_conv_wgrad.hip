// gfx950 HIP kernels of the build-defined conv frontend: the stride-1 layers' weight gradient, second form
// (lr_conv.hip holds the slab reduction and the host dispatch).
// No reference file: the reference has no conv frontend (regime X).
#include "lr_common.h"
#include <hip/hip_ext.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short bf16_t;  // storage type

// Two ds_read_b64_tr_b16 (gfx950 LDS transpose read) -> one MFMA operand (see lr_conv_dev.h).
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_tr_pair(const unsigned char* lds, int a0, int a1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// Weight gradient of the frontend's stride-1 layers:  dW[n][kt][kh][kw][c] = sum_pos dZ[pos][n] * X[pos + tap][c].
// The contraction runs over POSITIONS, the slow axis of both channels-last operands, while an MFMA operand wants 8
// consecutive k per lane: gfx950's ds_read_b64_tr_b16 does that transpose on the way out of LDS (16 lanes read a
// [4 positions][16 channels] block, 8 bytes each, and lane L receives channel L of the 4 positions; two reads = one
// operand), so dZ and X stay channels-last in LDS, as planes of [position][32 channels].  A workgroup = (temporal tap
// kt, slot) keeps dW[:, kt] — all (tap, 32-channel plane) units x row tiles — in accumulators while it walks its share
// of (2-frame x TH-row) tiles; the X patch is loaded already shifted by kt - 1 frames.  Partial results go to slabs,
// reduced in fixed order (conv_wgrad_slab_reduce_kernel, lr_conv.hip).
// This is the kernel's SECOND form (round 3; the first, rounds 1-2, is in the history), rebuilt around what the first
// spent outside its MFMAs — 34 % of a tile's 12.2 k cycles at layer 2 (r03_pixels_pmc_SQ_pass1: MFMA busy 0.63):
//   * 85 slots per temporal tap instead of 80: 255 workgroups (the first used 240 of the 256 CUs).  Blocks 0..239
//     keep the three kt siblings of a slot on one XCD; the last 15 blocks are slots 80..84, siblings side by side.
//   * the (tap, plane) units that do not divide by four waves are SPLIT BY M TILE instead of being repeated: every
//     wave runs FULL whole units plus a partial slot of PART row tiles of one left-over unit (layer 2: 6 units + one
//     of unit 24's two row tiles = 13 MFMAs per k16 step instead of 14; layer 3: 4 units + two of a left-over
//     unit's three = 14 instead of 15).  A wave addresses the dZ row tiles ROTATED by the first tile of its partial
//     slot, so that the slot's operands are local tiles 0.. for every wave and the register indices stay static.
//   * a k16 step takes its two position groups from frame 0 (lanes 0-31) and the same two groups from frame 1 (lanes
//     32-63): the frame is a per-lane constant folded into the lane's base address and everything else about a
//     fragment address is an instruction immediate — no VALU instruction in the MFMA loop (the first form computed
//     two position groups per lane and step: 205 VALU per tile in the loop).
//   * global -> register loads are buffer loads: a unit that is padding (halo column, row outside the frame, frame
//     outside the clip) gets bit 31 set in its offset, is out of the resource's range and returns zeros.  No select on
//     the way into LDS, two VALU per unit and tile on the way out of memory (the first form: ~10 + 4).  The
//     resources are re-based per tile (scalar arithmetic), so a lane's offsets are tile-invariant.
//   * the next tile's loads are issued behind the first MFMAs of the current tile and its LDS stores behind the
//     MFMAs of its last steps but one (one per MFMA, pinned), instead of in phases of their own in front of and
//     behind the MFMA loop; the tile loop is ONE basic block (the round after a workgroup's last tile moves zeros),
//     and a tile's scalars (frame, row band, time in the clip) advance by constants instead of being divided out.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kTr2Slots = LR_CONV_TR2_SLOTS;

// LDS image of one tile: the X patch planes, then the dZ planes, each rounded up to whole rounds of 256 sixteen-byte
// units (a thread moves units tid + 256 i; the units past the end of a region are loaded as zeros and land in its
// padding, so that no load and no store of the round needs a predicate).
template <int CIN, int MT, int KH, int KW, int W, int TT, int TH>
struct Tr2 {
  static constexpr int CH = CIN / 32, PH = TH + KH - 1, PW = W + KW - 1;
  static constexpr int XPOS = TT * PH * PW, ZPOS = TT * TH * W;
  static constexpr int XUNITS = CH * XPOS * 4, ZUNITS = MT * ZPOS * 4;
  static constexpr int XI = (XUNITS + 255) / 256, ZI = (ZUNITS + 255) / 256, UPT = XI + ZI;
  static constexpr int XREG = XI * 4096, ZREG = ZI * 4096, BUF = XREG + ZREG;
  static constexpr int LDS_BYTES = 2 * BUF, TAB_BYTES = 160 * 1024 - LDS_BYTES;   // tile table behind the buffers
  // UNPOOL: the dZ planes are rebuilt from POOLED units (one window x 8 channels: 16 bytes of dP + 8 codes -> the
  // window's four dZ units)
  static constexpr int PPOS = TT * (TH / 2) * (W / 2), PZU = MT * PPOS * 4, PZ = (PZU + 255) / 256;
};

// UNPOOL (the layer's forward fused ReLU + MaxPool): dZ is the POOLED gradient dP [F][H/2][W/2][COUT] and `code` the
// windows' codes (relu_pool4, lr_conv_dev.h); the dZ planes of a tile are rebuilt on the way into LDS — a thread loads
// 16 bytes of dP and 8 codes per pooled unit and stores the window's four units (unpool8's arithmetic, cut into slices
// of <= 8 VALU instructions that ride in the MFMA gaps like the stores do) — so the full-resolution dZ is never written
// to or read from memory.
template <int CIN, int MT, int KH, int KW, int W, int TT, int TH, bool UNPOOL>
__global__ __launch_bounds__(256, 1) void conv_wgrad_tr2_kernel(const bf16_t* __restrict__ X,
                                                                const bf16_t* __restrict__ dZ,
                                                                const unsigned char* __restrict__ code,
                                                                float* __restrict__ slabs, int F, int T, int H) {
  static_assert(TT == 2, "lanes 0-31 / 32-63 of a fragment read take frame 0 / 1 of the tile");
  static_assert(!UNPOOL || (TH % 2 == 0 && W % 2 == 0), "whole pooling windows per tile");
  typedef Tr2<CIN, MT, KH, KW, W, TT, TH> G;
  constexpr int CH = G::CH, PH = G::PH, PW = G::PW, XPOS = G::XPOS, ZPOS = G::ZPOS;
  constexpr int XUNITS = G::XUNITS, ZUNITS = G::ZUNITS, XI = G::XI, XREG = G::XREG, BUF = G::BUF;
  constexpr int PZ = G::PZ, PZU = G::PZU, PPOS = G::PPOS;
  // loads of a tile: the X units, then (UNPOOL) PZ 16-byte loads of dP and PZ 8-byte loads of codes, else the dZ units;
  // store-phase events: the X units' stores, then (UNPOOL) three slices per (pooled unit, window position) — the
  // third one stores —, else the dZ units' stores
  constexpr int UPT = UNPOOL ? XI + 2 * PZ : G::UPT;
  constexpr int NEV = UNPOOL ? XI + 12 * PZ : G::UPT;
  constexpr int NPRE = UNPOOL ? XI + PZ : G::UPT;
  constexpr int W4 = W / 4, GPS = TH * W4;   // position groups (4 columns) of one frame's rows
  constexpr int STEPS = GPS / 2;             // k16 steps per tile: 2 groups x 2 frames each
  constexpr int NU = KH * KW * CH, COUT = MT * 32;
  constexpr int FULL = NU / 4, LEFT = NU - 4 * FULL;             // whole units per wave, left-over units
  constexpr int PART = LEFT ? (LEFT * MT + 3) / 4 : 0;           // row tiles in a wave's partial slot
  constexpr int CHUNKS = PART ? (MT + PART - 1) / PART : 0;      // a left-over unit's row tiles in chunks of PART
  constexpr int UPW = FULL + (PART ? 1 : 0);
  constexpr int NM = FULL * MT + PART, ND = 2 * (MT + UPW);      // MFMAs / LDS reads per step
  // MFMA "gaps" g = st * NM + m of a tile: the next tile's table entry in step 1, its UPT loads one every LSTR gaps
  // from LG0, its UPT LDS stores one every SSTR gaps from MID (see the loop)
  constexpr int GAPS = STEPS * NM, SETUP = 1, LG0 = 2 * NM, MID = (LG0 + GAPS) / 2;
  constexpr int LSTR = (MID - LG0) / UPT, SSTR = (GAPS - 4 - MID) / NEV;
  static_assert(W % 4 == 0 && GPS % 2 == 0, "tile must be a whole number of k16 steps");
  static_assert(LEFT * CHUNKS <= 4, "one chunk of a left-over unit per wave");
  static_assert(ND >= NM && NM >= 13 && NPRE <= 31 && LSTR >= 1 && SSTR >= 1, "placement of the next tile's loads / stores");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int kt, slot;
  if (blockIdx.x < 240) {   // block b -> XCD b % 8; the kt siblings of a slot are blocks 8(3i + kt) + x
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    kt = kq % 3;
    slot = (kq / 3) * 8 + xcd;
  } else {
    const int r = blockIdx.x - 240;
    kt = r % 3;
    slot = 80 + r / 3;
  }
  const int htiles = H / TH;
  const int ntile = ((F + TT - 1) / TT) * htiles;

  f32x16 acc[NM];   // m = j * MT + i' for the whole units, FULL * MT + i' for the partial slot
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  // the partial slot of this wave: unit, first row tile, row tiles that are real (the rest is computed and dropped)
  int p_unit = NU - 1, p_m0 = 0, p_cnt = 0;
  if (PART) {
    const int lu = wave % (LEFT ? LEFT : 1), ck = wave / (LEFT ? LEFT : 1);
    if (ck < CHUNKS) {
      p_unit = 4 * FULL + lu;
      p_m0 = ck * PART;
      p_cnt = MT - p_m0 < PART ? MT - p_m0 : PART;
    }
  }
  const int sl = lane & 15, kg = lane >> 5;
  const int laneoff = ((lane >> 4) & 1) * 32 + (sl & 3) * 8 + (sl >> 2) * 64;
  // lane bases inside a buffer: X operand of unit slot j (tap shift + plane + this lane's frame), dZ row tile i'
  int xbase[UPW], zbase[MT];
#pragma unroll
  for (int j = 0; j < UPW; ++j) {
    const int u = j < FULL ? wave + 4 * j : p_unit;
    const int tap = u / CH, plane = u - tap * CH;
    xbase[j] = plane * XPOS * 64 + ((tap / KW) * PW + tap % KW) * 64 + kg * (PH * PW * 64) + laneoff;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int gi = i + p_m0;
    gi = gi >= MT ? gi - MT : gi;
    zbase[i] = XREG + gi * ZPOS * 64 + kg * (TH * W * 64) + laneoff;
  }

  // A thread moves the same 16-byte units of every tile (units tid + 256 i: XI rounds of the X patch, then the
  // rounds of dZ).  voff = byte offset from the tile's resource base; bit i of a mask: unit i is never data /
  // a row above the tile / a row below it / an X (dZ) unit of frame slot sf.
  const int negx = ((H + (KH - 1) / 2) * W + (KW - 1) / 2) * CIN;   // elements the X resource starts before the tile
  unsigned voff[NPRE];
  int zdst[UNPOOL ? PZ : 1];   // UNPOOL: byte offset (inside a buffer) of the window's first dZ unit
  unsigned m_nok = 0, m_top = 0, m_bot = 0, m_xf0 = 0, m_xf1 = 0, m_zf0 = 0, m_zf1 = 0;
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    voff[i] = 0;
    if (i < XI) {
      const int u = tid + 256 * i;
      if (u < XUNITS) {
        const int plane = u / (XPOS * 4), rem = u - plane * (XPOS * 4);
        const int pos = rem >> 2, c8 = rem & 3;
        const int sf = pos / (PH * PW), r2 = pos - sf * (PH * PW);
        const int ph = r2 / PW, pw = r2 - ph * PW;
        const int dh = ph - (KH - 1) / 2, w = pw - (KW - 1) / 2;
        voff[i] = (unsigned)(((((sf + kt - 1) * H + dh) * W + w) * CIN + plane * 32 + c8 * 8 + negx) * 2);
        if (w < 0 || w >= W) m_nok |= 1u << i;
        if (dh < 0) m_top |= 1u << i;
        if (dh >= TH) m_bot |= 1u << i;
        m_xf0 |= sf == 0 ? 1u << i : 0u;
        m_xf1 |= sf == 1 ? 1u << i : 0u;
      } else {
        m_nok |= 1u << i;
      }
    } else if constexpr (UNPOOL) {
      const int up = tid + 256 * (i - XI);
      zdst[i - XI] = XREG;
      if (up < PZU) {
        const int mt = up / (PPOS * 4), rem = up - mt * (PPOS * 4);
        const int ppos = rem >> 2, c8 = rem & 3;
        const int sf = ppos / ((TH / 2) * (W / 2)), r2 = ppos - sf * ((TH / 2) * (W / 2));
        const int hp = r2 / (W / 2), wp = r2 - hp * (W / 2);
        // element offset in dP (and byte offset in the codes) from the tile's pooled origin; x2 = bytes in dP
        voff[i] = (unsigned)((((sf * (H >> 1) + hp) * (W / 2) + wp) * COUT + mt * 32 + c8 * 8) * 2);
        zdst[i - XI] = XREG + ((mt * ZPOS + sf * (TH * W) + 2 * hp * W + 2 * wp) * 4 + c8) * 16;
        m_zf0 |= sf == 0 ? 1u << i : 0u;
        m_zf1 |= sf == 1 ? 1u << i : 0u;
      } else {
        m_nok |= 1u << i;
      }
    } else {
      const int uz = tid + 256 * (i - XI);
      if (uz < ZUNITS) {
        const int mt = uz / (ZPOS * 4), rem = uz - mt * (ZPOS * 4);
        const int pos = rem >> 2, c8 = rem & 3;
        const int sf = pos / (TH * W), r2 = pos - sf * (TH * W);
        const int h = r2 / W, w = r2 - h * W;
        voff[i] = (unsigned)((((sf * H + h) * W + w) * COUT + mt * 32 + c8 * 8) * 2);
        m_zf0 |= sf == 0 ? 1u << i : 0u;
        m_zf1 |= sf == 1 ? 1u << i : 0u;
      } else {
        m_nok |= 1u << i;
      }
    }
  }
  // A workgroup's tiles are slot, slot + 85, ...  What the loads of a tile need from its index — the two resource
  // bases and seven facts about its padding — is worked out ONCE, one tile per thread, into a table behind the two
  // LDS buffers (32 bytes per tile: X base, dZ base, flags), so that the tile loop has no scalar arithmetic worth
  // the name (divisions by the row-band count and the clip length, two 64-bit products and ~20 compares per tile:
  // ~95 scalar instructions in one dependent chain, which no scheduler spread over the MFMAs).  Entry `mine` is the
  // tile of zeros that is moved after the last real one, so that the loop body is the same straight line for every
  // tile.  flags: bit 0 no tile, 1 top row band, 2 bottom row band, 3 + sf dZ frame sf missing, 5 + sf X frame sf
  // missing (outside the clip once shifted by kt - 1).
  const int mine = slot < ntile ? (ntile - slot + kTr2Slots - 1) / kTr2Slots : 0;
  unsigned char* tab = lds + 2 * BUF;
  for (int e = tid; e <= mine; e += 256) {
    const int tile = slot + e * kTr2Slots;
    unsigned fl = 0x7f;
    uint64_t xa = (uint64_t)X, za = (uint64_t)dZ, ca = (uint64_t)code;
    if (e < mine) {
      const int ft = tile / htiles, hb = tile - ft * htiles;
      const int f0 = ft * TT;
      fl = (hb == 0 ? 2u : 0u) | (hb == htiles - 1 ? 4u : 0u);
#pragma unroll
      for (int sf = 0; sf < TT; ++sf) {
        const int f = f0 + sf, tt = f % T + kt - 1;
        fl |= f < F ? 0u : 8u << sf;
        fl |= (f < F && tt >= 0 && tt < T) ? 0u : 32u << sf;
      }
      const int64_t org = ((int64_t)f0 * H + hb * TH) * W;
      xa = (uint64_t)(X + (org * CIN - negx));
      za = (uint64_t)(dZ + org * COUT);
      if constexpr (UNPOOL) {
        const int64_t porg = ((int64_t)f0 * (H >> 1) + hb * (TH / 2)) * (W / 2) * COUT;
        za = (uint64_t)(dZ + porg);
        ca = (uint64_t)(code + porg);
      }
    }
    *reinterpret_cast<u32x4_t*>(tab + e * 32) = u32x4_t{(unsigned)xa, (unsigned)(xa >> 32), (unsigned)za, (unsigned)(za >> 32)};
    *reinterpret_cast<unsigned*>(tab + e * 32 + 16) = fl;
    if constexpr (UNPOOL) *reinterpret_cast<uint2*>(tab + e * 32 + 24) = make_uint2((unsigned)ca, (unsigned)(ca >> 32));
  }
  __syncthreads();
  u32x4_t pre[NPRE];
  uint2 prc[UNPOOL ? PZ : 1];   // UNPOOL: the codes of the pooled units
  u32x4_t ent;                // the table entry of the tile whose units are being loaded (the same in every lane)
  uint2 entc = make_uint2(0u, 0u);
  unsigned efl = 0, bad = 0;
  const bf16_t *xp = X, *zp = dZ;
  const unsigned char* cp = code;
  auto entry_read = [&](int e) {
    ent = *reinterpret_cast<const u32x4_t*>(tab + e * 32);
    efl = *reinterpret_cast<const unsigned*>(tab + e * 32 + 16);
    if constexpr (UNPOOL) entc = *reinterpret_cast<const uint2*>(tab + e * 32 + 24);
  };
  auto entry_bases = [&]() {
    const unsigned x0 = __builtin_amdgcn_readfirstlane(ent.x), x1 = __builtin_amdgcn_readfirstlane(ent.y);
    const unsigned z0 = __builtin_amdgcn_readfirstlane(ent.z), z1 = __builtin_amdgcn_readfirstlane(ent.w);
    xp = (const bf16_t*)(((uint64_t)x1 << 32) | x0);
    zp = (const bf16_t*)(((uint64_t)z1 << 32) | z0);
    if constexpr (UNPOOL) {
      const unsigned c0 = __builtin_amdgcn_readfirstlane(entc.x), c1 = __builtin_amdgcn_readfirstlane(entc.y);
      cp = (const unsigned char*)(((uint64_t)c1 << 32) | c0);
    }
  };
  // term k of the padding mask: a per-lane unit mask, taken when bit k of the flags is set
  auto entry_bad = [&](int k) {
    const unsigned m = k == 0 ? ~0u : k == 1 ? m_top : k == 2 ? m_bot : k == 3 ? m_zf0 : k == 4 ? m_zf1 : k == 5 ? m_xf0 : m_xf1;
    const unsigned on = (unsigned)__builtin_amdgcn_sbfe((int)efl, k, 1);   // 0 or ~0
    bad = (k == 0 ? m_nok : bad) | (m & on);
  };
  auto issue_unit = [&](int i) {   // global -> registers: one 16-byte unit (8 codes), zeros when it is padding
    if constexpr (UNPOOL) {
      if (i >= NPRE) {   // the codes of pooled unit i - NPRE: a byte per element, so half the dP offset
        const int p = i - NPRE;
        const __amdgpu_buffer_rsrc_t r =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(cp), (short)0, 0x7fffffff, 0x00020000);
        const unsigned o = (voff[XI + p] >> 1) | (((bad >> (XI + p)) & 1u) << 31);
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)o, 0, 0);
        prc[p] = make_uint2(v[0], v[1]);
        return;
      }
    }
    const __amdgpu_buffer_rsrc_t r =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(i < XI ? xp : zp), (short)0, 0x7fffffff, 0x00020000);
    const unsigned o = voff[i] | (((bad >> i) & 1u) << 31);
    pre[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)o, 0, 0);
  };
  // store-phase event e of the tile whose units are in pre / prc, into the buffer at `dep0` (a thread's X units sit at
  // tid * 16 + i * 4096): e < XI the X unit e; UNPOOL then (pooled unit p, window position j, slice): 0 = the bytes
  // that match position j, 1 = their byte masks, 2 = mask, select and store (unpool8's arithmetic, lr_conv_dev.h);
  // else the dZ unit e
  unsigned ua = 0, ub = 0;
  auto store_event = [&](unsigned char* buf, int e) {
    if (e < XI || !UNPOOL) {
      *reinterpret_cast<u32x4_t*>(buf + tid * 16 + e * 4096) = pre[e];
      return;
    }
    const int q = e - XI, p = q / 12, j = (q % 12) / 3, slice = q % 3;
    if (slice == 0) {
      ua = (0x08080808u - (prc[p].x ^ (0x01010101u * (unsigned)j))) & 0x08080808u;
      ub = (0x08080808u - (prc[p].y ^ (0x01010101u * (unsigned)j))) & 0x08080808u;
    } else if (slice == 1) {
      ua = (ua << 5) - (ua >> 3);
      ub = (ub << 5) - (ub >> 3);
    } else {
      const u32x4_t d = pre[XI + p];
      const u32x4_t o = {d.x & __builtin_amdgcn_perm(ua, ua, 0x01010000u), d.y & __builtin_amdgcn_perm(ua, ua, 0x03030202u),
                         d.z & __builtin_amdgcn_perm(ub, ub, 0x01010000u), d.w & __builtin_amdgcn_perm(ub, ub, 0x03030202u)};
      // (the last round's units past the end are no units: nothing to store — unlike the whole-round regions of the
      // X patch, the dZ planes have no padding for them to land in)
      if (p < PZ - 1 || tid < PZU - 256 * (PZ - 1))
        *reinterpret_cast<u32x4_t*>(buf + zdst[p] + (j >> 1) * (W * 64) + (j & 1) * 64) = o;
    }
  };

  entry_read(0);
  entry_bases();
#pragma unroll
  for (int k = 0; k < 7; ++k) entry_bad(k);
#pragma unroll
  for (int i = 0; i < UPT; ++i) issue_unit(i);
#pragma unroll
  for (int e = 0; e < NEV; ++e) store_event(lds, e);
  __syncthreads();
  for (int it = 0; it < mine; ++it) {
    const int cur = it & 1;
    unsigned char* dep = lds + (cur ^ 1) * BUF;   // where the next tile goes
    int xb[UPW], zb[MT];
#pragma unroll
    for (int j = 0; j < UPW; ++j) xb[j] = xbase[j] + cur * BUF;
#pragma unroll
    for (int i = 0; i < MT; ++i) zb[i] = zbase[i] + cur * BUF;
    // One wave per SIMD: a fragment read next to its use is an LDS round trip that nothing covers.  The tile is one
    // straight line, fully unrolled, WRITTEN in the order it is meant to run with a scheduling fence behind every
    // MFMA's group: the fragments of step st + 1 (double buffered by step parity) are read one per MFMA of step st;
    // the next tile's table entry is read and expanded among the MFMAs of step 1; its buffer loads ride behind every
    // LSTR-th MFMA of the first half of the tile and its LDS stores behind every SSTR-th of the second half.  SPREAD
    // OUT: the four waves run in step, so a load in one gap is 4 KB through the CU's 64 B/clk vector memory path — 64
    // cycles, two MFMAs' worth — and with one load (one store) behind EVERY MFMA of two steps those steps ran at 64-82
    // (56) cycles per MFMA instead of 34 (s_memtime stamps, round 3: a layer-2 tile = 390 cycles from the barrier
    // to the first MFMA's group + 234 MFMAs at 34.5-36 + the barrier, 9.0 k cycles; the first form 12.2 k).
    // Measured and dropped: reading step 0 of the NEXT tile behind the MFMAs of the last step (barrier moved in
    // front of that step, with a counted lgkmcnt so that it waits for the stores only) hides those 390 cycles and
    // changed nothing (layer 2 340.6 vs 339.9 us, layer 3 105.6 vs 105.6 on one box, tools/bench_conv_wgrad.py): at
    // 1.24 PFLOP/s on random operands the kernel runs against the chip's power budget, where a saved cycle comes back
    // as clock (MI355X_MICROARCH.md, DVFS give-back; the guide's best 8192^3 bf16 GEMMs reach 1.16-1.22 PFLOP/s).  (sched_group_barrier
    // pins, which the first form uses, lost the MFMAs of the store steps to the end of the tile; and the stores have
    // to be WRITTEN among the reads in any case: the compiler cannot know that the two LDS buffers do not alias and
    // keeps every read that precedes a store in the source in front of it.)
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 raw[2][ND];   // reads 2f, 2f + 1 = fragment f: dZ row tiles 0..MT-1, then the X operands of the unit slots
    auto read_one = [&](int st, int idx) {
      const int g = 2 * st + (idx & 1);
      const int h = g / W4, w0 = 4 * (g - h * W4);
      const int f = idx >> 1;
      const int a = f < MT ? zb[f] + (h * W + w0) * 64 : xb[f - MT] + (h * PW + w0) * 64;
      raw[st & 1][idx] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a));
    };
    auto frag = [&](int p, int f) {
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 v = __builtin_shufflevector(raw[p][2 * f], raw[p][2 * f + 1], 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(bf16x8, v);
    };
#pragma unroll
    for (int idx = 0; idx < ND; ++idx) read_one(0, idx);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      if (st + 1 < STEPS) {
#pragma unroll
        for (int idx = 0; idx < ND - NM; ++idx) read_one(st + 1, idx);
      }
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int j = m < FULL * MT ? m / MT : FULL, i = m < FULL * MT ? m % MT : m - FULL * MT;
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(st & 1, i), frag(st & 1, MT + j), acc[m], 0, 0, 0);
        if (st + 1 < STEPS) read_one(st + 1, ND - NM + m);
        const int gap = st * NM + m;
        if (gap >= LG0 && (gap - LG0) % LSTR == 0 && (gap - LG0) / LSTR < UPT) issue_unit((gap - LG0) / LSTR);
        if (gap >= MID && (gap - MID) % SSTR == 0 && (gap - MID) / SSTR < NEV) store_event(dep, (gap - MID) / SSTR);
        if (st == SETUP && m == 0) entry_read(it + 1);
        if (st == SETUP && m == 5) entry_bases();
        if (st == SETUP && m >= 6 && m < 13) entry_bad(m - 6);
        __builtin_amdgcn_sched_barrier(0);   // nothing moves across: the line runs as written
      }
    }
    __syncthreads();   // buffer `cur` is free again; the next tile is in place
  }
  // partial result of this workgroup: slabs[slot*3 + kt][tap][n][c]
  const int lr = lane & 31, lk = lane >> 5;
  float* out = slabs + (int64_t)(slot * 3 + kt) * (KH * KW) * COUT * CIN;
  auto store_tile = [&](const f32x16& a, int u, int gi) {
    const int tap = u / CH, plane = u - tap * CH;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = gi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      out[((int64_t)tap * COUT + n) * CIN + plane * 32 + lr] = a[r];
    }
  };
#pragma unroll
  for (int j = 0; j < FULL; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int gi = i + p_m0;
      gi = gi >= MT ? gi - MT : gi;
      store_tile(acc[j * MT + i], wave + 4 * j, gi);
    }
#pragma unroll
  for (int i = 0; i < PART; ++i)
    if (i < p_cnt) store_tile(acc[FULL * MT + i], p_unit, p_m0 + i);
}

}  // namespace

// The tile table has to fit behind the two LDS buffers: (tiles of a workgroup + 1) x 32 bytes.  Layer 2 takes 6-row
// tiles where the height allows, 4-row tiles otherwise.
static int tr2_tile_rows(int layer, int H) { return H % 6 == 0 ? 6 : (layer == 2 && H % 4 == 0 ? 4 : 0); }
static int tr2_table_bytes(int F, int H, int th) {
  const int ntile = ((F + 1) / 2) * (H / th);
  return ((ntile + kTr2Slots - 1) / kTr2Slots + 1) * 32;
}
int lr_conv_wgrad_tr2_supported(int layer, int F, int H) {
  if ((layer != 2 && layer != 3) || H <= 0 || F <= 0) return 0;
  const int th = tr2_tile_rows(layer, H);
  if (!th) return 0;
  const int room = layer == 3 ? Tr2<64, 3, 3, 3, 12, 2, 6>::TAB_BYTES
                              : (th == 6 ? Tr2<32, 2, 5, 5, 24, 2, 6>::TAB_BYTES : Tr2<32, 2, 5, 5, 24, 2, 4>::TAB_BYTES);
  return tr2_table_bytes(F, H, th) <= room;
}

// layer: 2 (24 wide, 32 -> 64 channels, 3x5x5; H % 6 == 0 or H % 4 == 0) or 3 (12 wide, 64 -> 96, 3x3x3; H % 6 == 0).
// slabs: 3 x LR_CONV_TR2_SLOTS partial results [slot * 3 + kt][tap][n][c].
// code != nullptr: dZ is the POOLED gradient [F][H/2][W/2][Cout] and code the windows' codes (the kernel un-pools).
int lr_conv_wgrad_tr2(int layer, const void* X, const void* dZ, const void* code, float* slabs, int F, int T, int H,
                      bool sample, hipEvent_t e0, hipEvent_t e1, hipStream_t stream) {
  static bool attr_set[6] = {false, false, false, false, false, false};
  const bf16_t* x = (const bf16_t*)X;
  const bf16_t* dz = (const bf16_t*)dZ;
  const unsigned char* cd = (const unsigned char*)code;
  if (!lr_conv_wgrad_tr2_supported(layer, F, H) || T <= 0) return LR_ERR_UNSUPPORTED;
  const int th = tr2_tile_rows(layer, H);
  const int tabb = tr2_table_bytes(F, H, th);
  lr_clear_error();
#define LR_WGTR2(IDX, UNP, ...)                                                                                 \
  do {                                                                                                      \
    constexpr int LDSMAX = 160 * 1024;                                                                      \
    const int LDSB = Tr2<__VA_ARGS__>::LDS_BYTES + tabb;                                                    \
    if (!attr_set[IDX]) {                                                                                   \
      if (hipFuncSetAttribute((const void*)conv_wgrad_tr2_kernel<__VA_ARGS__, UNP>,                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDSMAX) != hipSuccess)            \
        return LR_ERR_LAUNCH;                                                                               \
      attr_set[IDX] = true;                                                                                 \
    }                                                                                                       \
    if (sample) hipExtLaunchKernelGGL((conv_wgrad_tr2_kernel<__VA_ARGS__, UNP>), dim3(3 * kTr2Slots), dim3(256), \
                                      LDSB, stream, e0, e1, 0, x, dz, cd, slabs, F, T, H);                  \
    else hipLaunchKernelGGL((conv_wgrad_tr2_kernel<__VA_ARGS__, UNP>), dim3(3 * kTr2Slots), dim3(256), LDSB, \
                            stream, x, dz, cd, slabs, F, T, H);                                             \
  } while (0)
#define LR_WGTR2_KERNEL(IDX, ...)                                         \
  do {                                                                   \
    if (cd) LR_WGTR2(2 * (IDX) + 1, true, __VA_ARGS__);                   \
    else LR_WGTR2(2 * (IDX), false, __VA_ARGS__);                         \
  } while (0)
  if (cd && (H & 1)) return LR_ERR_UNSUPPORTED;
  if (layer == 2 && th == 6) LR_WGTR2_KERNEL(0, 32, 2, 5, 5, 24, 2, 6);
  else if (layer == 2) LR_WGTR2_KERNEL(2, 32, 2, 5, 5, 24, 2, 4);
  else LR_WGTR2_KERNEL(1, 64, 3, 3, 3, 12, 2, 6);
#undef LR_WGTR2_KERNEL
#undef LR_WGTR2
  return lr_launch_status();
}
