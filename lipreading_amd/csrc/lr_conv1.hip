// gfx950 HIP kernels of the build-defined conv frontend: the first layer (4-channel padded RGB or raw uint8 input,
// 3x5x5 taps, spatial stride 2, 32 output channels), forward and weight gradient, in a translation unit of their own.
// No reference file: the reference has no conv frontend (regime X).
#include "lr_common.h"
#include "lr_conv_dev.h"
#include <hip/hip_ext.h>

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
// This first part — the 4-channel patch (8 bytes per pixel, 29 kB), c1_* — serves the WEIGHT GRADIENT
// (and the tile walk both kernels share); the forward keeps a 3-channel patch of its own, f1_* below.
constexpr int C1_T = 16;                       // output tile edge
constexpr int C1_P = 2 * C1_T + 3;             // patch edge (stride 2, 5x5 taps): 35
constexpr int C1_PIX = C1_T * C1_T;            // 256 output pixels = 8 MFMA row tiles
constexpr int C1_PATCH = 3 * C1_P * C1_P * 4;  // bf16 elements of the patch

constexpr int C1_FPIX = C1_P * C1_P;          // pixels of one patch frame (1225)

// The patch is a RING of three frames: a workgroup walks the tiles (t = 0, 1, 2, ...) of one spatial
// window of one clip in order, so tile t only has to bring frame t+1 — frames t-1 and t are already
// in LDS (FETCH_SIZE with a fresh 3-frame patch per tile: 3.4x the input; with the ring ~1.25x).
// Frame ti of the clip lives in slot (ti + 1) % 3, i.e. temporal tap kt of tile t in slot (t + kt) % 3.
// (Round 2's counters show two-way LDS bank conflicts in both kernels, 37 % / 44 % of their LDS cycles: an operand read
// takes two output rows x 16 columns, i.e. every second pixel of input rows two apart, and both rows land on the same
// half of the banks.  Shifting every second PAIR of patch rows by one pixel — 36-pixel rows — removed them and
// changed neither kernel's time early in round 3, with this 4-channel patch in both.  The rebuilt forward is another
// matter: there the LDS array WAS the bound and the conflict-free row stride gave 16 %, see F1_RB.)
// tap = (kt*5 + kh)*5 + kw  ->  element offset inside the patch [3 slots][35][35][4]
__device__ __forceinline__ int c1_tap_off(int tap, int t) {
  if (tap >= 75) return -1;
  const int kw = tap % 5, kh = (tap / 5) % 5, kt = tap / 25;
  return ((((t + kt) % 3) * C1_P + kh) * C1_P + kw) * 4;
}

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
__device__ __forceinline__ void c1_frame_store(bf16_t* Ps, const uint2 (&rp)[C1_NPU], const unsigned (&rb)[C1_NPU],
                                               int tid, int t, int kt) {
  bf16_t* slot = Ps + ((t + kt) % 3) * C1_FPIX * 4;
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
    *reinterpret_cast<uint2*>(&slot[e * 4]) = v;
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
    c1_frame_store<U8>(Ps, r, r3, tid, t, kt);
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

// weight gradient of the first layer: slab[wg][n][k] = sum over the workgroup's tiles of
// dZ[pix][n] * patch(pix, k).  Wave w owns column tiles w, w+4, w+8 of the 10 (320 columns).
// POOLED: the layer's forward fused ReLU + MaxPool (lr_conv3d_forward_pooled), and instead of a
// materialised dZ the kernel takes the pooled gradient dP, the pooled activation and the window
// codes and rebuilds its dZ tile on the way into LDS (a window's gradient goes to position `code` if
// the pooled activation is > 0) — the 354 MB dZ of this layer is never written or read.  The bias
// gradient (column sums of dZ) falls out of the same pass: bias_part[wg][32].
template <bool POOLED, bool U8>
__global__ __launch_bounds__(256) void conv1_wgrad_patch_kernel(const bf16_t* __restrict__ X,
                                                                const bf16_t* __restrict__ dZ,
                                                                const bf16_t* __restrict__ pooled,
                                                                const unsigned char* __restrict__ code,
                                                                float* __restrict__ slabs,
                                                                float* __restrict__ bias_part, int frames,
                                                                int T, int Hin, int Win, int Ho, int Wo) {
  // Both operands are read with LDS transpose reads: the contraction runs over PIXELS, the slow axis
  // of the channels-last dZ tile and of the patch.  dZ tile: [pixel][32 channels], 64 B per pixel (4
  // pixels = 256 contiguous bytes per read).  im2col column (tap, c): a lane's 8 bytes are the 4
  // channels of one tap at one pixel, so the 16 source lanes of a read cover 4 pixels x 4 taps.
  constexpr int ZLD = 32;
  // the patch, then a zone of zeros the padded columns read: wide enough for the largest immediate of a k step
  constexpr int ZERO_ELEMS = (15 * 2 * C1_P * 8 + 64 + 8 + 15) / 16 * 8;
  __shared__ __attribute__((aligned(16))) bf16_t Ps[C1_PATCH + ZERO_ELEMS];
  __shared__ __attribute__((aligned(16))) bf16_t Zs[C1_PIX * ZLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lk = lane >> 5;
  const int sl = lane & 15, colhalf = (lane >> 4) & 1;
  const int tiles_x = (Wo + C1_T - 1) / C1_T, tiles_y = (Ho + C1_T - 1) / C1_T;
  const int64_t ntiles = (int64_t)frames * tiles_x * tiles_y;
  // column tile jt = wave + 4j covers taps 8jt..8jt+7; this lane sources tap 8jt + 4 colhalf + (sl & 3)
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int e = tid; e < ZERO_ELEMS; e += 256) Ps[C1_PATCH + e] = 0;

  uint2 rp[C1_NPU];
  unsigned rb[C1_NPU];
  uint4 rz[POOLED ? 2 : 4];   // POOLED: pooled gradient and activation of this thread's window
  uint2 rc = make_uint2(0u, 0u);
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int wpy = tid >> 5, wpx = (tid >> 2) & 7, wcg = tid & 3;   // POOLED: window (py, px) of the 8 x 8, 8 channels
  const C1Pix pm = c1_pix(tid);
  auto issue = [&](const C1Tile& c) {
    const int f = c.f, y0 = c.y0, x0 = c.x0;
    c1_frame_issue<U8, false>(X, pm, rp, rb, f, c.t, T, Hin, Win, y0, x0, 2);
    if (POOLED) {
      const int Hp = Ho >> 1, Wp = Wo >> 1;
      const int yp = (y0 >> 1) + wpy, xp = (x0 >> 1) + wpx;
      rz[0] = rz[1] = make_uint4(0u, 0u, 0u, 0u);
      rc = make_uint2(0u, 0u);
      if (yp < Hp && xp < Wp) {
        const int64_t pi = (((int64_t)f * Hp + yp) * Wp + xp) * 32 + wcg * 8;
        rz[0] = *reinterpret_cast<const uint4*>(dZ + pi);       // dP
        rz[1] = *reinterpret_cast<const uint4*>(pooled + pi);
        rc = *reinterpret_cast<const uint2*>(code + pi);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // dZ tile: 256 pixels x 32 channels in 16-byte units
        const int e = tid + i * 256;
        const int pix = e >> 2, u = e & 3;
        const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
        rz[i] = make_uint4(0u, 0u, 0u, 0u);
        if (y < Ho && x < Wo) rz[i] = *reinterpret_cast<const uint4*>(dZ + (((int64_t)f * Ho + y) * Wo + x) * 32 + u * 8);
      }
    }
  };
  const unsigned char* PsB = reinterpret_cast<const unsigned char*>(Ps);
  const unsigned char* ZsB = reinterpret_cast<const unsigned char*>(Zs);
  const int64_t q_end = ntiles * (blockIdx.x + 1) / gridDim.x;
  const int64_t q_begin = ntiles * blockIdx.x / gridDim.x;
  int64_t q = q_begin;
  C1Tile c = c1_tile(q < q_end ? q : 0, T, tiles_x, tiles_y);
  if (q < q_end) issue(c);
  for (; q < q_end; ++q) {
    const C1Tile n = c1_next(c, T, tiles_x, tiles_y);
    __syncthreads();
    if (q == q_begin || c.t == 0) c1_walk_start<U8>(X, pm, Ps, c.f, c.t, T, Hin, Win, c.y0, c.x0, tid);
    c1_frame_store<U8>(Ps, rp, rb, tid, c.t, 2);
    if (POOLED) {
      // rebuild the window's four dZ units (8 channels each): gradient at position `code`, if the
      // pooled activation is positive
      const unsigned gw[4] = {rz[0].x, rz[0].y, rz[0].z, rz[0].w}, pw_[4] = {rz[1].x, rz[1].y, rz[1].z, rz[1].w};
      unsigned o[4][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int wd = e >> 1, sh = (e & 1) * 16;
        const float act = bf2f((bf16_t)((pw_[wd] >> sh) & 0xffffu));
        const int arg = (int)(((e < 4 ? rc.x : rc.y) >> (8 * (e & 3))) & 3u);
        const unsigned g = act > 0.f ? ((gw[wd] >> sh) & 0xffffu) : 0u;
        bsum[e] += bf2f((bf16_t)g);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j == arg) o[j][wd] |= g << sh;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pix = (2 * wpy + (j >> 1)) * 16 + 2 * wpx + (j & 1);
        *reinterpret_cast<uint4*>(&Zs[pix * ZLD + wcg * 8]) = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tid + i * 256;
        *reinterpret_cast<uint4*>(&Zs[(e >> 2) * ZLD + (e & 3) * 8]) = rz[i];
      }
    }
    // byte offset of this lane's tap per column tile, and a mask that drops the pixel offset for the
    // padded columns (taps >= 75, and the third tile of waves 2 and 3, which run it on zeros rather
    // than branch: they would wait at the tile barrier anyway) so those read the zero zone
    // A fragment address = lane base + an immediate: a k step is one tile row of 16 pixels (ks * 2 patch rows, ks *
    // 1 KB of the dZ tile) and a lane's two 4-pixel groups are 4 pixels apart, so only the base depends on the lane
    // (round 2 recomputed pixel and tap offsets per read: 212 vector instructions around the 48 MFMAs of a tile).
    int pbase[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int tap = (wave + 4 * j) * 8 + 4 * colhalf + (sl & 3);
      const int o = (wave + 4 * j) < 10 ? c1_tap_off(tap, c.t) : -1;
      pbase[j] = o >= 0 ? o * 2 + (lk * 8 + (sl >> 2)) * 16 : C1_PATCH * 2;
    }
    __syncthreads();
    if (q + 1 < q_end) issue(n);
    // 16 k steps (16 pixels each), fully unrolled; the transpose reads of step ks+1 fly during the
    // MFMAs of step ks (fragments double buffered by parity, interleave pinned below)
    const int zbase = (lk * 8 + (sl >> 2)) * 64 + colhalf * 32 + (sl & 3) * 8;
    bf16x8 fa[2], fb[2][3];
    auto load_k = [&](int ks, bf16x8& a, bf16x8 (&b)[3]) {
      // this lane's source pixels of the two 4-pixel groups: k = lk*8 + {0..3 | 4..7} of tile row ks
      a = lds_tr_pair(ZsB, zbase + ks * 1024, zbase + ks * 1024 + 256);
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = lds_tr_pair(PsB, pbase[j] + ks * (2 * C1_P * 8), pbase[j] + ks * (2 * C1_P * 8) + 64);
    };
    constexpr int KS = C1_PIX / 16;
    load_k(0, fa[0], fb[0]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) load_k(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
      for (int j = 0; j < 3; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1], fb[ks & 1][j], acc[j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int ks = 0; ks + 1 < KS; ++ks) {
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    c = n;
  }
  float* out = slabs + (int64_t)blockIdx.x * 32 * 320;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (wave + 4 * j < 10) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        out[((r & 3) + 8 * (r >> 2) + 4 * lk) * 320 + (wave + 4 * j) * 32 + lr] = acc[j][r];
    }
  }
  if (POOLED && bias_part) {   // column sums of this workgroup's dZ tiles, fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(Zs);   // 256 x 8 floats
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 32) {
      const int gq = tid >> 3, e = tid & 7;
      float sacc = 0.f;
      for (int w = 0; w < 64; ++w) sacc += red[(w * 4 + gq) * 8 + e];
      bias_part[(int64_t)blockIdx.x * 32 + tid] = sacc;
    }
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

// pooled: dZ is the POOLED gradient and (pooled_act, code) rebuild the full-resolution one on the fly; bias_part (may
// be null) receives LR_CONV1_WGRAD_WGS x 32 column sums.  slabs: LR_CONV1_WGRAD_WGS x 32 x 320 partial results.
int lr_conv1_wgrad(bool pooled, bool u8, const void* X, const void* dZ, const void* pooled_act, const void* code,
                   float* slabs, float* bias_part, int frames, int T, int Hin, int Win, int Ho, int Wo, bool sample,
                   hipEvent_t e0, hipEvent_t e1, hipStream_t stream) {
  const int nwg = LR_CONV1_WGRAD_WGS;
  lr_clear_error();
#define LR_C1W(PV, U8V)                                                                                           \
  do {                                                                                                           \
    if (sample) hipExtLaunchKernelGGL((conv1_wgrad_patch_kernel<PV, U8V>), dim3(nwg), dim3(256), 0, stream, e0,   \
                                      e1, 0, (const bf16_t*)X, (const bf16_t*)dZ, (const bf16_t*)pooled_act,      \
                                      (const unsigned char*)code, slabs, bias_part, frames, T, Hin, Win, Ho, Wo); \
    else hipLaunchKernelGGL((conv1_wgrad_patch_kernel<PV, U8V>), dim3(nwg), dim3(256), 0, stream,                 \
                            (const bf16_t*)X, (const bf16_t*)dZ, (const bf16_t*)pooled_act,                       \
                            (const unsigned char*)code, slabs, bias_part, frames, T, Hin, Win, Ho, Wo);           \
  } while (0)
  if (pooled && u8) LR_C1W(true, true);
  else if (pooled) LR_C1W(true, false);
  else if (!u8) LR_C1W(false, false);
  else return LR_ERR_UNSUPPORTED;
#undef LR_C1W
  return lr_launch_status();
}
