// gfx950 HIP kernels of the build-defined conv frontend: the first layer (4-channel padded RGB or raw uint8 input,
// 3x5x5 taps, spatial stride 2, 32 output channels), forward and weight gradient, in a translation unit of their own.
// No reference file: the reference has no conv frontend (regime X).
#include "lr_common.h"
#include "lr_conv_dev.h"
#include <hip/hip_ext.h>
#include <type_traits>

namespace {

// ---------------------------------------------------------------------------------------------
// first layer (4-channel padded RGB input, spatial stride 2): patch-resident kernels
// ---------------------------------------------------------------------------------------------
// With Cin = 4 an im2col row is 75 separate 8-byte gathers and neighbouring rows overlap almost
// completely, so the generic implicit GEMM is gather-bound (82 TF/s forward, 36 TF/s weight
// gradient).  Here a workgroup owns a 16x16 tile of output pixels of one frame and loads the
// input patch it needs ONCE into LDS: 3 frames x 35 x 35 pixels.  Every operand of the forward
// product and of the weight gradient is then an LDS read at (pixel offset + tap offset): no im2col
// staging, no barrier inside the K loop.
// This first part — the 4-channel patch (8 bytes per pixel), c1_* — serves the WEIGHT GRADIENT (and the tile walk both
// kernels share); the forward keeps a 3-channel patch of its own, f1_* below.
constexpr int C1_T = 16;                       // output tile edge
constexpr int C1_P = 2 * C1_T + 3;             // patch edge (stride 2, 5x5 taps): 35
constexpr int C1_PIX = C1_T * C1_T;            // 256 output pixels = 8 MFMA row tiles
constexpr int C1_FPIX = C1_P * C1_P;           // pixels of one patch frame (1225)
// The weight gradient's patch is a RING of frames: a workgroup walks the tiles (t = 0, 1, 2, ...) of one spatial
// window of one clip in order, so tile t only has to bring frame t+1 — frames t-1 and t are already in LDS
// (FETCH_SIZE with a fresh 3-frame patch per tile: 3.4x the input; with the ring ~1.25x).  Four slots: frame ti of
// the clip lives in slot (ti + 1) & 3, i.e. temporal tap kt of tile t in slot (t + kt) & 3, and the fourth slot takes
// frame t+2 while tile t is being contracted.
// LDS layout of a slot: 44-pixel rows (352 B), 1568 pixels (12,544 B) per slot.  A transpose read of the patch takes,
// per 32-lane group, 4 pixels x 8 consecutive taps, i.e. up to three patch rows: with the rows packed (280 B) two of
// them share banks — two-way conflicts on most reads, 44 % of the kernel's LDS cycles in round 3
// (SQ_LDS_BANK_CONFLICT) — with 352-byte rows the three rows sit 24 banks apart (modelled over every tap group and
// ring rotation: 2.11 -> 1.02 cycles per lane group; measured: 11 % of the LDS cycles are conflicts now, the dZ
// tile's stores included).  Patch column p (input x = 2 x0 - 2 + p) sits at pixel column p + C1_XOFF of its row: the
// dword fill below stores from two pixels further left.
constexpr int C1_RING = 4;
constexpr int C1_RS = 44;                      // pixels between patch rows in LDS
constexpr int C1_FS = 1568;                    // pixels between the ring's frame slots
constexpr int C1_XOFF = 2;

// Frame kt (0..2) of tile (f = b*T + t, rows 2*y0-2.., cols 2*x0-2..): 1225 8-byte pixels, 5 per
// thread.  Split into "issue every global load" and "write to LDS" so that the NEXT tile's new frame
// (kt = 2) flies while the current tile's MFMAs run; only 5 staging registers stay live across the
// MFMA loop (15 for a whole patch cost a wave of occupancy).
constexpr int C1_NPU = (C1_FPIX + 255) / 256;   // 5
// U8: X is the raw clip, uint8 planar [frame][3][Hin][Win] — the three bytes of a pixel are loaded as
// they are (rp.x, rp.y, rb) and turned into the bf16 pixel (value / 255, 4th channel 0) when they are
// written to LDS, exactly as lr_clip_to_ndhwc_bf16 would have: no bf16 copy of the clip exists.
// Branch-free: every pixel is loaded — pixel (0, 0) of the frame (of frame f itself when the temporal tap leaves
// the clip) where the patch hangs over the edge — and bit 31 of rb[i] says whether it is real; c1_frame_store
// writes zeros for the others.  (`if (inside) load` made hipcc branch around each of the 15 byte loads of a tile:
// ~40 exec-mask branches per tile in kernels that are bound by their scalar / vector issue.)
// The kernels are bound by the instructions around their MFMAs (38 MFMAs per tile and wave against, in round 2,
// ~600 vector and ~300 scalar instructions), so what does not depend on the tile is worked out ONCE per thread:
// C1Pix = row and column of the thread's five pixels inside the 35 x 35 patch frame (a pixel index >= 1225 gets a
// row that no tile can reach).  Per tile a pixel then costs two adds, two unsigned compares, one multiply-add and
// one select (round 2: a division by 35, four signed compares, two clamps and 64-bit address arithmetic each).
struct C1Pix { int py[C1_NPU], px[C1_NPU]; };
__device__ __forceinline__ C1Pix c1_pix(int tid) {
  C1Pix r;
#pragma unroll
  for (int i = 0; i < C1_NPU; ++i) {
    const int e = tid + i * 256;
    r.py[i] = e < C1_FPIX ? e / C1_P : (1 << 20);
    r.px[i] = e < C1_FPIX ? e % C1_P : 0;
  }
  return r;
}
// (BRANCHFREE false: predicated loads, zeros for the rest — the weight-gradient kernel, whose loads sit in front of a
// long MFMA phase, measures 4-7 % slower with the branch-free form in both rounds, the forward 6 % faster.)
template <bool U8, bool BRANCHFREE = true>
__device__ __forceinline__ void c1_frame_issue(const bf16_t* __restrict__ X, const C1Pix& pm, uint2 (&rp)[C1_NPU],
                                               unsigned (&rb)[C1_NPU], int f, int t, int T, int Hin, int Win, int y0,
                                               int x0, int kt) {
  const int ti = t + kt - 1;
  const bool frame_ok = ti >= 0 && ti < T;
  const int64_t fr = frame_ok ? f + kt - 1 : f;
  const int yb = 2 * y0 - 2, xb = 2 * x0 - 2;
  const int64_t plane = (int64_t)Hin * Win;
  const unsigned char* p0 = reinterpret_cast<const unsigned char*>(X) + fr * 3 * plane;
  const unsigned char *p1 = p0 + plane, *p2 = p1 + plane;
  const bf16_t* pf = X + fr * plane * 4;
#pragma unroll
  for (int i = 0; i < C1_NPU; ++i) {
    const int yi = pm.py[i] + yb, xi = pm.px[i] + xb;
    const bool ok = frame_ok && (unsigned)yi < (unsigned)Hin && (unsigned)xi < (unsigned)Win;
    const int off = ok ? yi * Win + xi : 0;
    if (!BRANCHFREE) {
      rp[i] = make_uint2(0u, 0u);
      rb[i] = 0x80000000u;
      if (ok) {
        if (U8) {
          rp[i].x = p0[off];
          rp[i].y = p1[off];
          rb[i] = 0x80000000u | p2[off];
        } else {
          rp[i] = *reinterpret_cast<const uint2*>(pf + off * 4);
        }
      }
      continue;
    }
    if (U8) {
      rp[i].x = p0[off];
      rp[i].y = p1[off];
      rb[i] = (unsigned)p2[off] | (ok ? 0x80000000u : 0u);
    } else {
      rp[i] = *reinterpret_cast<const uint2*>(pf + off * 4);
      rb[i] = ok ? 0x80000000u : 0u;
    }
  }
}
template <bool U8>
__device__ __forceinline__ void c1_frame_store(bf16_t* Ps, const C1Pix& pm, const uint2 (&rp)[C1_NPU],
                                               const unsigned (&rb)[C1_NPU], int tid, int t, int kt) {
  bf16_t* slot = Ps + (((t + kt) & (C1_RING - 1)) * C1_FS + C1_XOFF) * 4;
#pragma unroll
  for (int i = 0; i < C1_NPU; ++i) {
    const int e = tid + i * 256;
    if (e >= C1_FPIX) continue;
    const bool ok = (rb[i] >> 31) != 0;
    uint2 v = rp[i];
    if (U8) {
      v.x = (unsigned)f2bf((float)rp[i].x * (1.f / 255.f)) | ((unsigned)f2bf((float)rp[i].y * (1.f / 255.f)) << 16);
      v.y = (unsigned)f2bf((float)(rb[i] & 0xffu) * (1.f / 255.f));
    }
    v.x = ok ? v.x : 0u;
    v.y = ok ? v.y : 0u;
    *reinterpret_cast<uint2*>(&slot[(pm.py[i] * C1_RS + pm.px[i]) * 4]) = v;
  }
}
// first tile of a walk: frames t-1 and t are fetched synchronously (once per ~28 tiles)
template <bool U8>
__device__ __forceinline__ void c1_walk_start(const bf16_t* __restrict__ X, const C1Pix& pm, bf16_t* Ps, int f, int t,
                                              int T, int Hin, int Win, int y0, int x0, int tid) {
#pragma unroll 1
  for (int kt = 0; kt < 2; ++kt) {
    uint2 r[C1_NPU];
    unsigned r3[C1_NPU];
    c1_frame_issue<U8>(X, pm, r, r3, f, t, T, Hin, Win, y0, x0, kt);
    c1_frame_store<U8>(Ps, pm, r, r3, tid, t, kt);
  }
}

// Walk order of the persistent first-layer kernels: tile q = seq*T + t with seq = (clip, ty, tx);
// workgroup w owns the contiguous range [w*N/G, (w+1)*N/G).  The first tile of a range is decoded with
// divisions, every later one is the previous one advanced (t, then tx, ty, clip: a few scalar instructions).
struct C1Tile { int f, t, y0, x0; };
__device__ __forceinline__ C1Tile c1_tile(int64_t q64, int T, int tiles_x, int tiles_y) {
  C1Tile r;
  const unsigned q = (unsigned)q64;   // tile counts stay far below 2^31: 32-bit divisions
  const unsigned seq = q / (unsigned)T;
  r.t = (int)(q - seq * (unsigned)T);
  const int tx = (int)(seq % (unsigned)tiles_x), ty = (int)((seq / (unsigned)tiles_x) % (unsigned)tiles_y);
  r.f = (int)(seq / (unsigned)(tiles_x * tiles_y)) * T + r.t;
  r.y0 = ty * C1_T;
  r.x0 = tx * C1_T;
  return r;
}
__device__ __forceinline__ C1Tile c1_next(const C1Tile& c, int T, int tiles_x, int tiles_y) {
  C1Tile n = c;
  n.t = c.t + 1;
  n.f = c.f + 1;
  if (n.t == T) {   // next spatial window of the clip, or the first one of the next clip
    n.t = 0;
    n.x0 = c.x0 + C1_T;
    const bool wrap_x = n.x0 == tiles_x * C1_T;
    n.x0 = wrap_x ? 0 : n.x0;
    n.y0 = wrap_x ? c.y0 + C1_T : c.y0;
    const bool wrap_y = n.y0 == tiles_y * C1_T;
    n.y0 = wrap_y ? 0 : n.y0;
    n.f = wrap_y ? c.f + 1 : c.f + 1 - T;
  }
  return n;
}

// ---------------------------------------------------------------------------------------------
// forward: a patch of THREE channels per pixel
// ---------------------------------------------------------------------------------------------
// The fourth channel is padding (always zero: the raw clip has three planes, lr_clip_to_ndhwc_bf16 writes a zero,
// lr_conv3d_pack_weights zero weights), and with it in the patch a quarter of the forward's MFMAs and LDS reads are
// spent on zeros — in a kernel that is bound by what it reads out of LDS (19 k steps x (1 KB of weights + 2 KB of
// pixels) per wave and tile = 228 KB per tile and workgroup, 1.8 k cycles of the LDS against 1.2 k of the MFMAs).
// The forward therefore keeps its own patch with 6 bytes per pixel: the 5 taps x 3 channels of one (kt, kh) row of
// the filter are 15 CONTIGUOUS bf16 behind pixel (2y + kh, 2x), one k step of 16 with one zero weight — 15 k steps
// instead of 19 (30 MFMAs per wave and tile instead of 38, 180 KB out of LDS instead of 228), and a fragment address
// is a per-kernel lane base + the tile's slot offset + an immediate (round 2 / 3: a select and an add per read).
// Rows are 36 pixels: the 36th is only ever multiplied by the zero weight.  Pixels go in as PAIRS (the
// patch's x origin is even): one 16-bit load per plane and pair from the raw clip, three dwords per pair into LDS.
// (The weight gradient contracts over pixels with LDS transpose reads of 8-byte units and keeps the 4-channel patch.)
// The forward keeps its weight fragments in REGISTERS (60 VGPRs: one LDS read in five less, three waves per SIMD instead
// of four).  MEASURED (round 4, same box): 112.4 -> 107.5 us against reading them from LDS at every k step.
constexpr int LR_C1_FWD_WGS = 768;
constexpr int F1_RW = C1_P + 1;                  // pixels per patch row
// bytes per patch row: 36 pixels = 216, padded to 224.  A fragment read is two ds_read2_b32 (4-byte alignment is all a
// 6-byte pixel gives), i.e. dword accesses banked (address / 4) mod 32 over lanes 0-31 = two output rows x 16 columns:
// the columns are 3 dwords apart (conflict-free on their own), and the two rows land on disjoint banks only when two
// patch rows are 16 dwords (mod 32) apart: row stride = 32 (mod 64) bytes.  (216: SQ_LDS_BANK_CONFLICT = 43 % of the
// kernel's LDS cycles and the LDS array busy for two thirds of its duration, profiles/r03_pixels_pmc_SQ_pass2/3.)
constexpr int F1_RB = 224;
static_assert(F1_RB >= F1_RW * 6 && F1_RB % 64 == 32 && (4 * F1_RB + 12) / 4 <= 255, "row stride: banks and ds_read2_b32 offsets");
constexpr int F1_SLOT = C1_P * F1_RB;            // bytes per ring slot (one frame): 7840
constexpr int F1_PATCH = 3 * F1_SLOT;            // 23,520 bytes
constexpr int F1_PPR = F1_RW / 2;                // pixel pairs per row (18)
constexpr int F1_PAIRS = C1_P * F1_PPR;          // pixel pairs per frame (630)
constexpr int F1_NPU = (F1_PAIRS + 255) / 256;   // pairs per thread (3)
constexpr int F1_KS = 15;                        // k steps: (kt, kh) rows of the filter
constexpr int F1_WLD = F1_KS * 16 + 8;           // weight row in LDS (496 B: conflict-free b128 reads)

// row and (even) column of the thread's pairs inside the patch frame, and py * Win + px
struct F1Pix { int py[F1_NPU], px[F1_NPU], lin[F1_NPU]; };
__device__ __forceinline__ F1Pix f1_pix(int tid, int Win) {
  F1Pix r;
#pragma unroll
  for (int i = 0; i < F1_NPU; ++i) {
    // past the frame's last pair a thread repeats its own previous pair (same load, same LDS write): a branch around
    // the LDS write would leave the pair's loads "pending" on the path that skips it, and hipcc then waits for ALL
    // outstanding memory operations (vmcnt) at the next instruction that redefines one of those registers — here in
    // front of the next tile's loads (the previous tile's stores) and in front of the MFMA loop (the loads just issued)
    const int e0 = tid + i * 256, e = e0 < F1_PAIRS ? e0 : e0 - 256;
    r.py[i] = e / F1_PPR;
    r.px[i] = 2 * (e % F1_PPR);
    r.lin[i] = r.py[i] * Win + r.px[i];
  }
  return r;
}
// One frame of the patch in flight: raw words per pair (U8: the pair's two bytes of each plane; else the two 4-channel
// bf16 pixels) and, in the MASKED forms, two validity bits per pair in `ok` (bit 2i: left pixel, bit 2i + 1: right
// pixel): a pixel outside the frame / the clip loads pixel (0, 0) and is written as zeros (branch-free as in
// c1_frame_issue).
// EVENW: Win is even, so a pair (its left column is even) is inside or outside the frame as a whole and is ONE aligned
// load per plane (U8) or ONE 16-byte load; odd widths load the two pixels on their own.
// U8 && EVENW (the product's path) needs no mask at all: the three planes are read through buffer descriptors of
// exactly one plane each (none of it when the frame lies outside the clip), so rows above / below the frame are out
// of the descriptor's range and come back as zeros — which bf16(0 / 255) is; columns left / right of the frame get an
// out-of-range offset.  A pair then costs 4 vector instructions to address (offset = the thread's py * Win + px, a
// kernel constant, + the tile's scalar; column test; select) instead of ~14 (two coordinates, four compares, a 64-bit
// multiply-add, three 64-bit adds, the mask bits), and nothing to mask when it is written.
template <bool U8, bool EVENW> struct F1Stage {
  static constexpr bool MASKED = !(U8 && EVENW);
  unsigned w[F1_NPU][U8 ? 3 : 4];
  unsigned ok;
};
template <bool U8, bool EVENW>
__device__ __forceinline__ void f1_frame_issue(const bf16_t* __restrict__ X, const F1Pix& pm, F1Stage<U8, EVENW>& s, int f,
                                               int t, int T, int Hin, int Win, int y0, int x0, int kt) {
  const int ti = t + kt - 1;
  const bool frame_ok = ti >= 0 && ti < T;
  const int64_t fr = frame_ok ? f + kt - 1 : f;
  const int yb = 2 * y0 - 2, xb = 2 * x0 - 2;
  const int64_t plane = (int64_t)Hin * Win;
  const unsigned char* p0 = reinterpret_cast<const unsigned char*>(X) + fr * 3 * plane;
  const unsigned char *p1 = p0 + plane, *p2 = p1 + plane;
  const bf16_t* pf = X + fr * plane * 4;
  s.ok = 0u;
  if constexpr (U8 && EVENW) {
    const int records = frame_ok ? (int)plane : 0;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p0), (short)0, records, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p1), (short)0, records, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p2), (short)0, records, 0x00020000);
    const int sb = yb * Win + xb;
#pragma unroll
    for (int i = 0; i < F1_NPU; ++i) {
      const bool col_ok = (unsigned)(pm.px[i] + xb) < (unsigned)Win;
      const int off = col_ok ? pm.lin[i] + sb : (int)0x80000000;
      s.w[i][0] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r0, off, 0, 0);
      s.w[i][1] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r1, off, 0, 0);
      s.w[i][2] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r2, off, 0, 0);
    }
    return;
  } else {
#pragma unroll
    for (int i = 0; i < F1_NPU; ++i) {
      const int yi = pm.py[i] + yb, xi = pm.px[i] + xb;
      const bool row_ok = frame_ok && (unsigned)yi < (unsigned)Hin;
      const bool ok0 = row_ok && (unsigned)xi < (unsigned)Win;
      const bool ok1 = EVENW ? ok0 : (row_ok && (unsigned)(xi + 1) < (unsigned)Win);
      const int off0 = ok0 ? yi * Win + xi : 0;
      s.ok |= (ok0 ? 1u : 0u) << (2 * i) | (ok1 ? 2u : 0u) << (2 * i);
      if constexpr (EVENW) {   // bf16 pixels (U8 && EVENW returned above)
        const uint4 v = *reinterpret_cast<const uint4*>(pf + (int64_t)off0 * 4);
        s.w[i][0] = v.x, s.w[i][1] = v.y, s.w[i][2] = v.z, s.w[i][3] = v.w;
      } else {
        const int off1 = ok1 ? yi * Win + xi + 1 : 0;
        if constexpr (U8) {
          s.w[i][0] = (unsigned)p0[off0] | ((unsigned)p0[off1] << 8);
          s.w[i][1] = (unsigned)p1[off0] | ((unsigned)p1[off1] << 8);
          s.w[i][2] = (unsigned)p2[off0] | ((unsigned)p2[off1] << 8);
        } else {
          const uint2 a = *reinterpret_cast<const uint2*>(pf + (int64_t)off0 * 4);
          const uint2 b = *reinterpret_cast<const uint2*>(pf + (int64_t)off1 * 4);
          s.w[i][0] = a.x, s.w[i][1] = a.y, s.w[i][2] = b.x, s.w[i][3] = b.y;
        }
      }
    }
  }
}
// bf16 of two bytes of the clip, exactly as lr_clip_to_ndhwc_bf16 makes them (byte / 255 in fp32, round to nearest
// even), as ONE packed multiply and ONE packed conversion: {lo: byte `bl` of wl, hi: byte `bh` of wh}.  (Written value
// by value hipcc spends cvt + mul + cvt + shift + or on each: 28 vector instructions per pair of pixels against 15.)
template <int bl, int bh>
__device__ __forceinline__ unsigned f1_bf_pair(unsigned wl, unsigned wh) {
  const f32x2_t v = f32x2_t{(float)((wl >> (8 * bl)) & 0xffu), (float)((wh >> (8 * bh)) & 0xffu)} * (1.f / 255.f);
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
template <bool U8, bool EVENW>
__device__ __forceinline__ void f1_frame_store(unsigned char* PsB, const F1Pix& pm, const F1Stage<U8, EVENW>& s, int tid,
                                               int t, int kt) {
  unsigned char* slot = PsB + ((t + kt) % 3) * F1_SLOT;
#pragma unroll
  for (int i = 0; i < F1_NPU; ++i) {
    unsigned d0, d1, d2;   // {c0 c1} {c2 | c0'} {c1' c2'} of the pair's left and right (') pixel
    if constexpr (U8) {
      // planes 0 and 1 in one word (bytes: a a' b b') so that every byte is a v_cvt_f32_ubyteN away
      const unsigned ab = s.w[i][0] | (s.w[i][1] << 16), c = s.w[i][2];
      d0 = f1_bf_pair<0, 2>(ab, ab);
      d1 = f1_bf_pair<0, 1>(c, ab);
      d2 = f1_bf_pair<3, 1>(ab, c);
    } else {
      d0 = s.w[i][0];
      d1 = (s.w[i][1] & 0xffffu) | (s.w[i][2] << 16);
      d2 = (s.w[i][2] >> 16) | (s.w[i][3] << 16);
    }
    if (F1Stage<U8, EVENW>::MASKED) {
      const bool ok0 = (s.ok >> (2 * i)) & 1u, ok1 = (s.ok >> (2 * i + 1)) & 1u;
      d0 = ok0 ? d0 : 0u;
      d1 = (ok0 ? d1 & 0xffffu : 0u) | (ok1 ? d1 & 0xffff0000u : 0u);
      d2 = ok1 ? d2 : 0u;
    }
    unsigned* dst = reinterpret_cast<unsigned*>(slot + pm.py[i] * F1_RB + pm.px[i] * 6);
    dst[0] = d0;
    dst[1] = d1;
    dst[2] = d2;
  }
}
// first tile of a walk: frames t-1 and t are fetched synchronously (once per ~28 tiles)
template <bool U8, bool EVENW>
__device__ __forceinline__ void f1_walk_start(const bf16_t* __restrict__ X, const F1Pix& pm, unsigned char* PsB, int f,
                                              int t, int T, int Hin, int Win, int y0, int x0, int tid) {
#pragma unroll 1
  for (int kt = 0; kt < 2; ++kt) {
    F1Stage<U8, EVENW> s;
    f1_frame_issue<U8, EVENW>(X, pm, s, f, t, T, Hin, Win, y0, x0, kt);
    f1_frame_store<U8, EVENW>(PsB, pm, s, tid, t, kt);
  }
}

// Persistent workgroups: the 32 x 225 weights are staged once, then tiles are streamed.
// POOL: the epilogue applies ReLU -> MaxPool((1,2,2)) in registers (a lane holds all four pixels of
// its windows) and writes the pooled activation + the 2-bit position of the window's first maximum
// (row-major scan, torch's rule) instead of the full-resolution activation.
template <bool POOL, bool U8, bool EVENW>
__global__ __launch_bounds__(256) void conv1_fwd_patch_kernel(const bf16_t* __restrict__ X,
                                                              const bf16_t* __restrict__ Wp,  // [32][75 taps][4]
                                                              const float* __restrict__ bias,
                                                              bf16_t* __restrict__ Y, unsigned char* __restrict__ code,
                                                              int frames, int T,
                                                              int Hin, int Win, int Ho, int Wo, int relu) {
  __shared__ __attribute__((aligned(16))) unsigned char PsB[F1_PATCH + 8];
  __shared__ __attribute__((aligned(16))) bf16_t Ws[32 * F1_WLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lk = lane >> 5;
  const int tiles_x = (Wo + C1_T - 1) / C1_T, tiles_y = (Ho + C1_T - 1) / C1_T;
  const int64_t ntiles = (int64_t)frames * tiles_x * tiles_y;
  // weights: row n, k step s = kt*5 + kh, element j = kw*3 + c (j = 15 and the row's tail: zeros)
  for (int e = tid; e < 32 * F1_WLD; e += 256) {
    const int n = e / F1_WLD, u = e - n * F1_WLD;
    const int st = u >> 4, j = u & 15;
    Ws[e] = (st < F1_KS && j < 15) ? Wp[n * 300 + (st * 5 + j / 3) * 4 + j % 3] : (bf16_t)0;
  }

  // wave w: row tiles 2w, 2w+1; row tile m covers pixels (y = 2m + (lr>>4), x = lr & 15); a lane's fragment of a
  // (kt, kh) step = 16 bytes at its pixel's (2y + kh, 2x) + 16 lk
  int pixbase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int yl = 2 * (2 * wave + i) + (lr >> 4), xl = lr & 15;
    pixbase[i] = (2 * yl) * F1_RB + (2 * xl) * 6 + lk * 16;
  }
  const float bv = bias ? bias[lr] : 0.f;
  const F1Pix pm = f1_pix(tid, Win);
  bf16x8 wreg[F1_KS];
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < F1_KS; ++ks) wreg[ks] = *reinterpret_cast<const bf16x8*>(&Ws[lr * F1_WLD + ks * 16 + lk * 8]);
  const bool exact = Ho % C1_T == 0 && Wo % C1_T == 0;   // no tile hangs over the output's edge
  F1Stage<U8, EVENW> stg;              // the new frame (temporal tap 2) of the NEXT tile, in flight during a tile's MFMAs
  const int64_t q_end = ntiles * (blockIdx.x + 1) / gridDim.x;
  const int64_t q_begin = ntiles * blockIdx.x / gridDim.x;
  if (q_begin >= q_end) return;
  C1Tile c = c1_tile(q_begin, T, tiles_x, tiles_y);
  // the first tile's patch, synchronously
  f1_walk_start<U8, EVENW>(X, pm, PsB, c.f, c.t, T, Hin, Win, c.y0, c.x0, tid);
  f1_frame_issue<U8, EVENW>(X, pm, stg, c.f, c.t, T, Hin, Win, c.y0, c.x0, 2);
  f1_frame_store<U8, EVENW>(PsB, pm, stg, tid, c.t, 2);
  __syncthreads();
  // A tile: issue the next tile's frame -> MFMAs -> barrier -> that frame into the ring slot the MFMAs no longer read
  // -> barrier -> epilogue.  The frame's loads are consumed BEFORE this tile's output stores are issued: vmcnt counts
  // loads and stores in one queue, and with the stores in front of the wait (round 2 / 3 had the LDS write at the top
  // of the next tile) every tile waited for its predecessor's stores to be acknowledged.
  for (int64_t q = q_begin; q < q_end; ++q) {
    const int f = c.f, y0 = c.y0, x0 = c.x0;
    const C1Tile n = c1_next(c, T, tiles_x, tiles_y);
    const bool more = q + 1 < q_end;
    if (more) f1_frame_issue<U8, EVENW>(X, pm, stg, n.f, n.t, T, Hin, Win, n.y0, n.x0, 2);
    f32x16 acc[2];
    // (Measured and dropped with the 4-channel patch, round 3: a K order in which the two lane halves of a step are a
    // CONSTANT distance apart so that a fragment address is a lane base plus an immediate — bit-identical results and
    // 150-166 us instead of 134; with the interleave written out behind scheduling fences: 144.  The 3-channel patch
    // gets the immediates for free: a step is one contiguous row.)
    // 15 k steps of 16 (one (kt, kh) row: 5 taps x 3 channels + a zero), fully unrolled; the fragments of step
    // ks+1 are read during the MFMAs of step ks.
    int abase[3][2];   // ring slot of temporal tap kt + the lane's pixel
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        abase[kt][i] = pixbase[i] + ((c.t + kt) % 3) * F1_SLOT;
        // six registers of their own: left to itself hipcc keeps ONE base and re-derives the other five in front of
        // every read (45 vector adds per tile), because a ds_read2_b32 offset only reaches 1020 bytes
        asm volatile("" : "+v"(abase[kt][i]));
      }
    uint4 af[2][2];
    bf16x8 bw[2];
    auto load_k = [&](int ks, uint4 (&a)[2], bf16x8& b) {
      const int kt = ks / 5, kh = ks % 5;
      b = wreg[ks];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned* p = reinterpret_cast<const unsigned*>(PsB + abase[kt][i] + kh * F1_RB);
        a[i] = make_uint4(p[0], p[1], p[2], p[3]);
      }
    };
    load_k(0, af[0], bw[0]);
#pragma unroll
    for (int ks = 0; ks < F1_KS; ++ks) {
      if (ks + 1 < F1_KS) load_k(ks + 1, af[(ks + 1) & 1], bw[(ks + 1) & 1]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ks & 1][i]), bw[ks & 1],
                                                         ks == 0 ? zero : acc[i], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
    for (int ks = 0; ks + 1 < F1_KS; ++ks) {
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __syncthreads();            // this tile's fragments are no longer being read
    if (more) {
      if (n.t == 0) f1_walk_start<U8, EVENW>(X, pm, PsB, n.f, n.t, T, Hin, Win, n.y0, n.x0, tid);   // a new walk: whole patch
      f1_frame_store<U8, EVENW>(PsB, pm, stg, tid, n.t, 2);
    }
    __syncthreads();
    if (POOL) {
      // row tile = 2 output rows x 16 columns: registers r and r + 8 are vertical neighbours,
      // r and r + 1 (r even) horizontal ones
      const int Hp = Ho >> 1, Wp2 = Wo >> 1;
      if (exact) {   // every window is inside the output: one base per pooled row, the rest is an immediate
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int64_t ob = (((int64_t)f * Hp + (y0 >> 1) + 2 * wave + i) * Wp2 + (x0 >> 1) + 2 * lk) * 32 + lr;
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
              const int r0 = 4 * a + 2 * pp;
              bf16_t best;
              int arg;
              relu_pool4(acc[i][r0], acc[i][r0 + 1], acc[i][r0 + 8], acc[i][r0 + 9], bv, best, arg);
              Y[ob + (pp + 4 * a) * 32] = best;
              code[ob + (pp + 4 * a) * 32] = (unsigned char)arg;
            }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
              const int r0 = 4 * a + 2 * pp;
              const int yp = (y0 >> 1) + 2 * wave + i, xp = (x0 >> 1) + pp + 4 * a + 2 * lk;
              if (yp >= Hp || xp >= Wp2) continue;
              bf16_t best;
              int arg;
              relu_pool4(acc[i][r0], acc[i][r0 + 1], acc[i][r0 + 8], acc[i][r0 + 9], bv, best, arg);
              const int64_t o = (((int64_t)f * Hp + yp) * Wp2 + xp) * 32 + lr;
              Y[o] = best;
              code[o] = (unsigned char)arg;
            }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int prow = (r & 3) + 8 * (r >> 2) + 4 * lk;         // pixel index inside the row tile
          const int y = y0 + 2 * (2 * wave + i) + (prow >> 4), x = x0 + (prow & 15);
          if (y >= Ho || x >= Wo) continue;
          float v = acc[i][r] + bv;
          if (relu) v = fmaxf(v, 0.f);
          Y[(((int64_t)f * Ho + y) * Wo + x) * 32 + lr] = f2bf(v);
        }
    }
    c = n;
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: slab[wg][n][k] = sum over the workgroup's tiles of dZ[pix][n] * patch(pix, k)
// ---------------------------------------------------------------------------------------------
// POOLED: the layer's forward fused ReLU + MaxPool (lr_conv3d_forward_pooled), and instead of a materialised dZ the
// kernel takes the pooled gradient dP and the window codes and rebuilds its dZ tile on the way into LDS (a window's
// gradient goes to position `code`; code 4 — ReLU blocked the window — to none): the 354 MB dZ of this layer is never
// written or read, and neither is the pooled activation (round 3 read it for the ReLU mask the codes carry: 88 MB).
// The bias gradient (column sums of dZ) falls out of the same pass: bias_part[wg][32].
// Both operands are read with LDS transpose reads: the contraction runs over PIXELS, the slow axis of the channels-
// last dZ tile and of the patch.  dZ tile: [pixel][32 channels], 64 B per pixel (4 pixels = 256 contiguous bytes per
// read).  im2col column (tap, c): a lane's 8 bytes are the 4 channels of one tap at one pixel, so the 16 source
// lanes of a read cover 4 pixels x 4 taps.
//
// Three ROLES in one 768-thread workgroup per CU, a wave of each on every SIMD:
//   waves 0..3  (contract) run the MFMAs of tile q out of one dZ buffer and three slots of the patch ring, while
//   waves 4..7  (frame)    bring tile q + 1's new frame of the ring from the raw clip and
//   waves 8..11 (dZ)       un-pool tile q + 1's gradient into the other dZ buffer;
// ONE barrier per tile hands tile q + 1 over and tile q - 1's buffers back.  (Rounds 2-4 alternated a fill phase —
// ~300 vector instructions per wave — and a contraction phase — 48 MFMAs — in every wave, two barriers per tile,
// two 256-thread workgroups per CU: 144 us, matrix pipes busy 45 % of the time, a third workgroup per CU slower.
// With the roles the fill's vector instructions issue beside the contraction's MFMAs: 114 us in the step.  What is
// left: the matrix pipe's 1.28 k cycles per tile and the fills' ~1.1 k cycles of vector issue per SIMD add up rather
// than overlap — period 2.45 k cycles, SQ counters of the round — so the kernel is bound by the vector instructions
// of the fills; measured and dropped on that road: un-pooling by scatter (zero the window, eight 2-byte stores at
// the position the code names: a quarter fewer vector instructions, 5x the LDS bank conflicts, 133 us), static
// wave priorities either way (no change / slower), loads four tiles ahead with masks at the store (fills alone 15 %
// faster, kernel 4 % slower).)
//   LDS: a ring of FOUR frame slots (tile t reads frames t-1, t, t+1; the frame role writes frame t+2) + TWO dZ tiles
//        = 4 x 12,544 + 2 x 16,384 = 82,944 bytes.
//   A walk start (a new spatial window: all three frames are new) cannot be staged under the previous tile — it
//   would overwrite slots that tile reads — so the roles take one extra barrier there (once per 75 tiles).
// The contraction splits K as well as N: wave w takes pixel rows 8 (w >> 1) .. + 7 of the tile (8 k steps) and column
// tiles 5 (w & 1) .. + 4 — 40 MFMAs on every wave (with 3 / 3 / 2 / 2 column tiles per wave: 48 on two waves) and
// 192 instead of 224 transpose-read pairs per tile; the two k halves are added once, through LDS, when the workgroup
// has walked all its tiles.  Columns of taps 75..79 (the im2col matrix has 80 tap columns for 75 taps) are never
// read back: their lanes read tap 74's pixels — a broadcast — where round 3 kept a zone of zeros for them, and
// column tiles 10, 11 are not computed at all.
// The raw clip in DWORDS.  A patch row starts at x = 2 x0 - 2 = 2 mod 4 (x0 is a multiple of 16), so the ten aligned
// dwords from x = 2 x0 - 4 cover it (40 pixels for 35; the 44-pixel LDS rows hold them: columns 0, 1 and 37..39 are
// never read).  A frame is 35 x 10 = 350 units of (3 planes x 4 pixels): three dword loads, v_cvt_f32_ubyte0..3
// straight out of them, two 16-byte LDS stores — against 4 x 3 byte loads, per-pixel address arithmetic and four
// 8-byte stores for the same pixels (fill role alone, loads only: 72 us with byte loads).  Needs Win % 4 == 0 and a
// 4-byte aligned clip (the host picks the byte path otherwise).
constexpr int C1W_UD = 10, C1W_UNITS = C1_P * C1W_UD, C1W_NU = (C1W_UNITS + 255) / 256;   // 350 units, 2 per thread
// Units outside the image, the clip or the frame's 350 load nothing and convert their zeros to zeros: no validity mask
// reaches the store.  (The loads therefore sit in a branch, and hipcc, which cannot count loads issued in a branch,
// waits for every load in flight at the first use of a staging register: the fill roles run one tile ahead of their
// loads.  Unconditional loads, four tiles ahead, with a mask applied at the store — measured — run the fills alone
// 15 % faster and the kernel 4 % slower.)
struct C1Units { unsigned a[C1W_NU], b[C1W_NU], c[C1W_NU]; };
__device__ __forceinline__ void c1w_units_issue(const bf16_t* __restrict__ X, C1Units& r, int tid, int f, int t, int T,
                                                int Hin, int Win, int y0, int x0, int kt) {
  const int ti = t + kt - 1;
  const bool frame_ok = ti >= 0 && ti < T;
  const int64_t fr = frame_ok ? f + kt - 1 : f;
  const int yb = 2 * y0 - 2, xd = 2 * x0 - 4;
  const int64_t plane = (int64_t)Hin * Win;
  const unsigned char* p0 = reinterpret_cast<const unsigned char*>(X) + fr * 3 * plane;
#pragma unroll
  for (int i = 0; i < C1W_NU; ++i) {
    const int u = tid + 256 * i, ur = u / C1W_UD, ud = u - ur * C1W_UD;
    const int y = yb + ur, x = xd + 4 * ud;
    r.a[i] = r.b[i] = r.c[i] = 0u;
    if (frame_ok && u < C1W_UNITS && (unsigned)y < (unsigned)Hin && (unsigned)x < (unsigned)Win) {
      const unsigned char* p = p0 + y * Win + x;
      r.a[i] = *reinterpret_cast<const unsigned*>(p);
      r.b[i] = *reinterpret_cast<const unsigned*>(p + plane);
      r.c[i] = *reinterpret_cast<const unsigned*>(p + 2 * plane);
    }
  }
}
__device__ __forceinline__ uint2 c1w_pixel(float r, float g, float b) {   // value / 255 as lr_clip_to_ndhwc_bf16 has it
  return make_uint2((unsigned)f2bf(r * (1.f / 255.f)) | ((unsigned)f2bf(g * (1.f / 255.f)) << 16),
                    (unsigned)f2bf(b * (1.f / 255.f)));
}
__device__ __forceinline__ void c1w_units_store(bf16_t* Ps, const C1Units& r, int tid, int t, int kt) {
  unsigned char* slot = reinterpret_cast<unsigned char*>(Ps) + ((t + kt) & (C1_RING - 1)) * (C1_FS * 8);
#pragma unroll
  for (int i = 0; i < C1W_NU; ++i) {
    const int u = tid + 256 * i, ur = u / C1W_UD, ud = u - ur * C1W_UD;
    if (u < C1W_UNITS) {
      const unsigned a = r.a[i], b = r.b[i], c = r.c[i];
      const uint2 p0 = c1w_pixel((float)(a & 0xffu), (float)(b & 0xffu), (float)(c & 0xffu));
      const uint2 p1 = c1w_pixel((float)((a >> 8) & 0xffu), (float)((b >> 8) & 0xffu), (float)((c >> 8) & 0xffu));
      const uint2 p2 = c1w_pixel((float)((a >> 16) & 0xffu), (float)((b >> 16) & 0xffu), (float)((c >> 16) & 0xffu));
      const uint2 p3 = c1w_pixel((float)(a >> 24), (float)(b >> 24), (float)(c >> 24));
      uint4* dst = reinterpret_cast<uint4*>(slot + (ur * C1_RS + 4 * ud) * 8);
      dst[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
      dst[1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
    }
  }
}
constexpr int C1W_ZT = C1_PIX * 32;                                   // bf16 elements of one dZ tile
constexpr int C1W_LDS = (C1_RING * C1_FS * 4 + 2 * C1W_ZT) * 2;      // bytes: 82,944
template <bool POOLED, int UM>   // UM: X is 0 = bf16 NDHWC (4 channels), 1 = the raw uint8 clip read in bytes, 2 = in dwords
__global__ __launch_bounds__(768) void conv1_wgrad_roles_kernel(const bf16_t* __restrict__ X,
                                                                const bf16_t* __restrict__ dZ,
                                                                const unsigned char* __restrict__ code,
                                                                float* __restrict__ slabs,
                                                                float* __restrict__ bias_part, int frames, int T,
                                                                int Hin, int Win, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c1w_lds[];
  bf16_t* Ps = reinterpret_cast<bf16_t*>(c1w_lds);
  bf16_t* Zs = Ps + C1_RING * C1_FS * 4;
  constexpr int ZLD = 32;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tiles_x = (Wo + C1_T - 1) / C1_T, tiles_y = (Ho + C1_T - 1) / C1_T;
  const int64_t ntiles = (int64_t)frames * tiles_x * tiles_y;
  const int64_t q_end = ntiles * (blockIdx.x + 1) / gridDim.x;
  const int64_t q_begin = ntiles * blockIdx.x / gridDim.x;
  float* out = slabs + (int64_t)blockIdx.x * 32 * 320;
  float* red = reinterpret_cast<float*>(c1w_lds);   // end of the walk: 2 x 5 x 16 x 64 partial sums, then 256 x 8 bias sums
  constexpr int RED_ACC = 2 * 5 * 16 * 64;

  if (wave < 4) {
    // ---------------- contract ----------------
    const int lane = threadIdx.x & 63;
    const int lr = lane & 31, lk = lane >> 5;
    const int sl = lane & 15, colhalf = (lane >> 4) & 1;
    const int khalf = wave >> 1, ct0 = (wave & 1) * 5;
    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned char* PsB = reinterpret_cast<const unsigned char*>(Ps);
    const unsigned char* ZsB = reinterpret_cast<const unsigned char*>(Zs);
    // this lane's taps (columns of taps 75..79 are never read back: their lanes read tap 74's pixels, a broadcast)
    int tap_kt[5], tap_in[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      int tap = (ct0 + j) * 8 + 4 * colhalf + (sl & 3);
      tap = tap < 75 ? tap : 74;
      tap_kt[j] = tap / 25;
      tap_in[j] = ((((tap / 5) % 5) * C1_RS + tap % 5 + C1_XOFF) * 4) * 2 + (lk * 8 + (sl >> 2)) * 16 +
                  khalf * 8 * (2 * C1_RS * 8);
    }
    const int zlane = (lk * 8 + (sl >> 2)) * 64 + colhalf * 32 + (sl & 3) * 8 + khalf * 8 * 1024;
    // Software-pipelined across tiles: the barrier that hands tile q + 1 over sits between the READS of tile q's last
    // k step and its MFMAs, and the first reads of tile q + 1 are issued right behind it — those MFMAs cover their
    // latency (at the top of the tile loop the twelve reads of a tile's first k step were exposed: the contraction
    // alone took 1.85 k cycles per tile for 1.28 k of MFMAs).
    int64_t q = q_begin;
    C1Tile c = c1_tile(q < q_end ? q : 0, T, tiles_x, tiles_y);
    int pbase[5], zbase = 0;
    bf16x8 fa[2], fb[2][5];
    auto bases = [&](const C1Tile& tl, int64_t qq) {
#pragma unroll
      for (int j = 0; j < 5; ++j) pbase[j] = ((tl.t + tap_kt[j]) & 3) * (C1_FS * 8) + tap_in[j];
      zbase = zlane + (int)(qq & 1) * (C1W_ZT * 2);
    };
    auto load_k = [&](int ks, bf16x8& a, bf16x8 (&b)[5]) {
      a = lds_tr_pair(ZsB, zbase + ks * 1024, zbase + ks * 1024 + 256);
#pragma unroll
      for (int j = 0; j < 5; ++j)
        b[j] = lds_tr_pair(PsB, pbase[j] + ks * (2 * C1_RS * 8), pbase[j] + ks * (2 * C1_RS * 8) + 64);
    };
    if (q < q_end) {
      bases(c, q);
      __syncthreads();   // the first tile is staged
      load_k(0, fa[0], fb[0]);
    }
    auto steps_0_to_6 = [&]() {
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) {
        load_k(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
        for (int j = 0; j < 5; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1], fb[ks & 1][j], acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) {
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    };
    auto step_7 = [&]() {
#pragma unroll
      for (int j = 0; j < 5; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1][j], acc[j], 0, 0, 0);
    };
    // (the last tile is peeled off: with "is there a next tile" as a branch inside the loop, the paths merge in front of
    // step 7's MFMAs and hipcc makes them wait for the NEW tile's reads)
    for (; q + 1 < q_end; ++q) {
      const C1Tile n = c1_next(c, T, tiles_x, tiles_y);
      steps_0_to_6();
      bases(n, q + 1);
      if (n.t == 0) __syncthreads();   // walk start: the frame role waits for this tile's reads
      __syncthreads();                 // tile q + 1 is staged (and this tile's reads are done)
      load_k(0, fa[0], fb[0]);
      step_7();
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
      c = n;
    }
    if (q < q_end) {
      steps_0_to_6();
      step_7();
    }
    __syncthreads();   // every read of the walk is done: LDS becomes the reduction buffer
    if (khalf == 1) {
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((wave & 1) * 5 + j) * 16 + r) * 64 + lane] = acc[j][r];
    }
    __syncthreads();
    if (khalf == 0) {
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          out[((r & 3) + 8 * (r >> 2) + 4 * lk) * 320 + (ct0 + j) * 32 + lr] =
              acc[j][r] + red[(((wave & 1) * 5 + j) * 16 + r) * 64 + lane];
    }
    return;
  }

  // ---------------- fill ----------------
  // Two fill roles: waves 4..7 bring the patch ring's new frame, waves 8..11 the dZ tile (each alone would take longer
  // than the contraction: 95 us for both in one role with the contraction switched off, against its 78).  Two sets of
  // staging registers, the tile loop unrolled by two: a tile's loads are issued when the tile before it has been staged.
  constexpr bool U8 = UM != 0;
  const bool frame_role = wave < 8;
  const int tid = threadIdx.x - (frame_role ? 256 : 512);
  int64_t q = q_begin;
  C1Tile c0 = c1_tile(q < q_end ? q : 0, T, tiles_x, tiles_y);
  C1Tile c1 = c1_next(c0, T, tiles_x, tiles_y);
  // the tile loop both fill roles run: stage(S, tile, q) writes tile q out of the staging registers S into LDS,
  // issue(S, tile) loads a tile into them
  auto walk = [&](auto& S0, auto& S1, auto&& issue, auto&& stage) {
    if (q < q_end) issue(S0, c0);
    if (q + 1 < q_end) issue(S1, c1);
    for (; q < q_end; q += 2) {
      const C1Tile c2 = c1_next(c1, T, tiles_x, tiles_y);
      stage(S0, c0, q);
      if (q + 2 < q_end) issue(S0, c2);
      __syncthreads();   // tile q is staged (and the contraction is done with tile q - 1)
      if (q + 1 < q_end) {
        const C1Tile c3 = c1_next(c2, T, tiles_x, tiles_y);
        stage(S1, c1, q + 1);
        if (q + 3 < q_end) issue(S1, c3);
        __syncthreads();
        c0 = c2;
        c1 = c3;
      }
    }
    __syncthreads();
  };
  if (frame_role) {
    struct Frame {
      uint2 rp[UM == 2 ? 1 : C1_NPU];
      unsigned rb[UM == 2 ? 1 : C1_NPU];
      C1Units un;
    };
    const C1Pix pm = c1_pix(tid);
    Frame S0, S1;
    walk(S0, S1,
         [&](Frame& S, const C1Tile& c) {
           if constexpr (UM == 2) c1w_units_issue(X, S.un, tid, c.f, c.t, T, Hin, Win, c.y0, c.x0, 2);
           else c1_frame_issue<U8, false>(X, pm, S.rp, S.rb, c.f, c.t, T, Hin, Win, c.y0, c.x0, 2);
         },
         [&](const Frame& S, const C1Tile& c, int64_t qq) {
           if (qq == q_begin || c.t == 0) {
             if (qq != q_begin) __syncthreads();   // the previous tile still reads the slots a walk start rewrites
             if constexpr (UM == 2) {
#pragma unroll 1
               for (int kt = 0; kt < 2; ++kt) {
                 C1Units w;
                 c1w_units_issue(X, w, tid, c.f, c.t, T, Hin, Win, c.y0, c.x0, kt);
                 c1w_units_store(Ps, w, tid, c.t, kt);
               }
             } else {
               c1_walk_start<U8>(X, pm, Ps, c.f, c.t, T, Hin, Win, c.y0, c.x0, tid);
             }
           }
           if constexpr (UM == 2) c1w_units_store(Ps, S.un, tid, c.t, 2);
           else c1_frame_store<U8>(Ps, pm, S.rp, S.rb, tid, c.t, 2);
         });
    __syncthreads();
    return;
  }
  struct Grad {
    uint4 rz[POOLED ? 1 : 4];   // POOLED: the pooled gradient of this thread's window (8 channels)
    uint2 rc;
  };
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int wpy = tid >> 5, wpx = (tid >> 2) & 7, wcg = tid & 3;   // POOLED: window (py, px) of the 8 x 8, 8 channels
  Grad G0, G1;
  walk(G0, G1,
       [&](Grad& S, const C1Tile& c) {
         const int f = c.f, y0 = c.y0, x0 = c.x0;
         if (POOLED) {
           const int Hp = Ho >> 1, Wp = Wo >> 1;
           const int yp = (y0 >> 1) + wpy, xp = (x0 >> 1) + wpx;
           S.rz[0] = make_uint4(0u, 0u, 0u, 0u);
           S.rc = make_uint2(0u, 0u);
           if (yp < Hp && xp < Wp) {
             const int64_t pi = (((int64_t)f * Hp + yp) * Wp + xp) * 32 + wcg * 8;
             S.rz[0] = *reinterpret_cast<const uint4*>(dZ + pi);       // dP
             S.rc = *reinterpret_cast<const uint2*>(code + pi);
           }
         } else {
#pragma unroll
           for (int i = 0; i < 4; ++i) {   // dZ tile: 256 pixels x 32 channels in 16-byte units
             const int e = tid + i * 256;
             const int pix = e >> 2, u = e & 3;
             const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
             S.rz[i] = make_uint4(0u, 0u, 0u, 0u);
             if (y < Ho && x < Wo)
               S.rz[i] = *reinterpret_cast<const uint4*>(dZ + (((int64_t)f * Ho + y) * Wo + x) * 32 + u * 8);
           }
         }
       },
       [&](const Grad& S, const C1Tile& c, int64_t qq) {
         if (qq != q_begin && c.t == 0) __syncthreads();   // (the frame role's walk start)
         bf16_t* Zq = Zs + (int)(qq & 1) * C1W_ZT;
         if (POOLED) {
           // rebuild the window's four dZ units (8 channels each): the gradient goes to position `code` (code 4 —
           // ReLU blocked the window — to none); at most one unit holds it, so their OR is what the bias gradient sums
           uint4 o[4];
           unpool8(S.rz[0], S.rc, o);
           const unsigned gw[4] = {o[0].x | o[1].x | o[2].x | o[3].x, o[0].y | o[1].y | o[2].y | o[3].y,
                                   o[0].z | o[1].z | o[2].z | o[3].z, o[0].w | o[1].w | o[2].w | o[3].w};
#pragma unroll
           for (int wd = 0; wd < 4; ++wd) {
             bsum[2 * wd] += __uint_as_float(gw[wd] << 16);
             bsum[2 * wd + 1] += __uint_as_float(gw[wd] & 0xffff0000u);
           }
           // store s writes position s of the even windows and s ^ 1 of the odd ones: the two windows of an 8-lane
           // store group then land on different halves of the 32 banks (pixels 2 apart are 128 bytes apart: the same
           // banks)
           const bool oddw = (wpx & 1) != 0;
#pragma unroll
           for (int sidx = 0; sidx < 4; ++sidx) {
             const uint4 a = o[sidx], b = o[sidx ^ 1];
             const uint4 v = make_uint4(oddw ? b.x : a.x, oddw ? b.y : a.y, oddw ? b.z : a.z, oddw ? b.w : a.w);
             const int j = sidx ^ (wpx & 1);
             const int pix = (2 * wpy + (j >> 1)) * 16 + 2 * wpx + (j & 1);
             *reinterpret_cast<uint4*>(&Zq[pix * ZLD + wcg * 8]) = v;
           }
         } else {
#pragma unroll
           for (int i = 0; i < 4; ++i) {
             const int e = tid + i * 256;
             *reinterpret_cast<uint4*>(&Zq[(e >> 2) * ZLD + (e & 3) * 8]) = S.rz[i];
           }
         }
       });
  if (POOLED && bias_part) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[RED_ACC + tid * 8 + e] = bsum[e];
  }
  __syncthreads();
  if (POOLED && bias_part && tid < 32) {   // column sums of this workgroup's dZ tiles, fixed order
    const int gq = tid >> 3, e = tid & 7;
    float sacc = 0.f;
    for (int w = 0; w < 64; ++w) sacc += red[RED_ACC + (w * 4 + gq) * 8 + e];
    bias_part[(int64_t)blockIdx.x * 32 + tid] = sacc;
  }
}

}  // namespace

// pool: fused ReLU + 2x2 max-pool epilogue (Y = pooled activation, code = window positions); u8: X is the raw clip.
int lr_conv1_forward(bool pool, bool u8, const void* X, const void* Wp, const float* bias, void* Y, unsigned char* code,
                     int frames, int T, int Hin, int Win, int Ho, int Wo, int relu, bool sample, hipEvent_t e0,
                     hipEvent_t e1, hipStream_t stream) {
  const bf16_t* x = (const bf16_t*)X;
  const bf16_t* w = (const bf16_t*)Wp;
  bf16_t* y = (bf16_t*)Y;
  int tiles = frames * ((Ho + C1_T - 1) / C1_T) * ((Wo + C1_T - 1) / C1_T);
  if (tiles > LR_C1_FWD_WGS) tiles = LR_C1_FWD_WGS;   // persistent: 4 workgroups per CU, each streams its share of tiles
  lr_clear_error();
#define LR_C1B(POOLV, U8V, EV)                                                                                    \
  do {                                                                                                           \
    if (sample) hipExtLaunchKernelGGL((conv1_fwd_patch_kernel<POOLV, U8V, EV>), dim3(tiles), dim3(256), 0, stream, \
                                      e0, e1, 0, x, w, bias, y, code, frames, T, Hin, Win, Ho, Wo, relu);         \
    else hipLaunchKernelGGL((conv1_fwd_patch_kernel<POOLV, U8V, EV>), dim3(tiles), dim3(256), 0, stream, x, w,    \
                            bias, y, code, frames, T, Hin, Win, Ho, Wo, relu);                                   \
  } while (0)
  // pair loads: an even width AND a base the pair's load is aligned at (2 bytes of a plane, 16 bytes of two bf16 pixels)
  const bool pairs = Win % 2 == 0 && (reinterpret_cast<uintptr_t>(X) & (u8 ? 1u : 15u)) == 0;
#define LR_C1(POOLV, U8V)                                                                                         \
  do {                                                                                                           \
    if (pairs) LR_C1B(POOLV, U8V, true);                                                                         \
    else LR_C1B(POOLV, U8V, false);                                                                              \
  } while (0)
  if (pool && u8) LR_C1(true, true);
  else if (pool) LR_C1(true, false);
  else if (u8) LR_C1(false, true);
  else LR_C1(false, false);
#undef LR_C1
#undef LR_C1B
  return lr_launch_status();
}

// pooled: dZ is the POOLED gradient and the window codes rebuild the full-resolution one on the fly; bias_part (may
// be null) receives LR_CONV1_WGRAD_WGS x 32 column sums.  slabs: LR_CONV1_WGRAD_WGS x 32 x 320 partial results.
int lr_conv1_wgrad(bool pooled, bool u8, const void* X, const void* dZ, const void* code, float* slabs, float* bias_part,
                   int frames, int T, int Hin, int Win, int Ho, int Wo, bool sample, hipEvent_t e0, hipEvent_t e1,
                   hipStream_t stream) {
  const int nwg = LR_CONV1_WGRAD_WGS;
  lr_clear_error();
  static bool attr[4] = {false, false, false, false};
#define LR_C1W(IDX, PV, UMV)                                                                                      \
  do {                                                                                                           \
    if (!attr[IDX]) {                                                                                            \
      if (hipFuncSetAttribute((const void*)conv1_wgrad_roles_kernel<PV, UMV>,                                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, C1W_LDS) != hipSuccess)                \
        return LR_ERR_LAUNCH;                                                                                    \
      attr[IDX] = true;                                                                                          \
    }                                                                                                            \
    if (sample) hipExtLaunchKernelGGL((conv1_wgrad_roles_kernel<PV, UMV>), dim3(nwg), dim3(768), C1W_LDS, stream, \
                                      e0, e1, 0, (const bf16_t*)X, (const bf16_t*)dZ, (const unsigned char*)code, \
                                      slabs, bias_part, frames, T, Hin, Win, Ho, Wo);                            \
    else hipLaunchKernelGGL((conv1_wgrad_roles_kernel<PV, UMV>), dim3(nwg), dim3(768), C1W_LDS, stream,           \
                            (const bf16_t*)X, (const bf16_t*)dZ, (const unsigned char*)code, slabs, bias_part,   \
                            frames, T, Hin, Win, Ho, Wo);                                                        \
  } while (0)
  // the raw clip in dwords: rows of whole dwords at an aligned base (tiles start at multiples of 16 output columns)
  const bool dwords = Win % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 3u) == 0;
  if (pooled && u8 && dwords) LR_C1W(0, true, 2);
  else if (pooled && u8) LR_C1W(1, true, 1);
  else if (pooled) LR_C1W(2, true, 0);
  else if (!u8) LR_C1W(3, false, 0);
  else return LR_ERR_UNSUPPORTED;
#undef LR_C1W
  return lr_launch_status();
}
