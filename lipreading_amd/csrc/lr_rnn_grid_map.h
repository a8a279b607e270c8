// lr_rnn_grid_map.h — index algebra of the GRID recurrence (lr_rnn_grid.hip): who owns which unit, which element of
// W_hh sits in which lane of which MFMA fragment, where an accumulator register goes in the exchange.  Plain integer
// functions, compiled by hipcc into the kernels AND by g++ into oracle/grid_map_check.cpp (tests/test_grid_map.py), which
// plays one forward and one backward step through them on the CPU and compares with the plain matrix products: the
// layout is checked in the build container, before a kernel ever runs.
//
// Geometry (LSTM, 1152 < H <= 1536; the decoder behind a BiLSTM-700 / 768 encoder, better_model.py:134-148):
//   HP = 1536 padded units; 192 members = R x C = 24 row groups x 8 column groups, one workgroup (compute unit) each;
//   block b -> column group c = b % 8 (the dispatcher places b, b + 8, ... on XCD b % 8: a column group = one XCD),
//   row group r = b / 8.
//   member (r, c) OWNS units 64 r + 8 c + u8, u8 = 0..7: it runs their cell, keeps their c state, publishes their h;
//   it HOLDS the block of W_hh with rows = the 4 x 64 gate rows of row group r's units [64 r, 64 r + 64) and columns =
//   the 192 units of column group c, K_c = { 64 r' + 8 c + u8 : r' = 0..23, u8 = 0..7 } in the order kk = 8 r' + u8
//   (= the order in which the h blocks of the column group's members arrive): 256 x 192 x (hi + lo) bf16 = 196 KB =
//   192 MFMA fragments, 48 per wave, all in registers.
//
//   forward   gates[rows of r] = sum over c of W[rows of r, K_c] h[K_c]: (1) all-gather h[K_c] inside the column group
//             (23 x 1 KB per 32 samples, same XCD), (2) 256 x 192 x 32 product, (3) reduce-scatter of the partial
//             gate sums inside the row group (member (r, c') needs the 32 gate rows of ITS 8 units from all 8 column
//             groups: 7 x 4 KB per 32 samples, across XCDs), (4) cell.
//   backward  dh[K_c] = sum over r of W[rows of r, K_c]^T dG[rows of r]: (1) reduce-scatter of the partial dh inside the
//             column group (23 x 1 KB), (2) cell backward, (3) all-gather dG inside the row group (7 x 4 KB), (4) product.
//   A 1-D split (every member all of h, or all of dh) would move 196 KB per member and step: 37.7 MB per step chip-wide,
//   the same bytes as re-streaming W_hh.  The 2-D split moves ~80 KB per member.
//
// MFMA v_mfma_f32_16x16x32_bf16, D[16 x 16] += A[16 x 32] B[32 x 16]: lane l holds A[l % 16][8 (l / 16) + e], B[8 (l / 16) + e][l % 16],
// e = 0..7, and D[4 (l / 16) + i][l % 16], i = 0..3.  A = the weights (rows = gate rows or units), B = the state or dG
// (columns = samples): a lane's four accumulator registers are one 16-byte exchange item.
#pragma once

#if defined(__HIPCC__)
#define LRG_HD __host__ __device__ __forceinline__
#else
#define LRG_HD inline
#endif

namespace lrg {

constexpr int R = 24, C = 8, NM = R * C;   // row groups, column groups, members
constexpr int UM = 8;                      // units a member owns
constexpr int HP = R * C * UM;             // 1536
constexpr int KC = HP / C;                 // 192 units of a column group = K of the forward product
constexpr int GR = 4 * (HP / R);           // 256 gate rows of a row group = K of the backward product
constexpr int SB = 32;                     // samples per block
constexpr int FT = GR / 16, FQ = KC / 32;  // forward: 16 row tiles x 6 k steps (4 tiles per wave)
constexpr int BT = KC / 16, BQ = GR / 32;  // backward: 12 row tiles x 8 k steps (3 tiles per wave)
constexpr int FFRAG = (FT / 4) * FQ * 2, BFRAG = (BT / 4) * BQ * 2;   // fragments per wave: 48, 48
static_assert(FFRAG == 48 && BFRAG == 48, "48 fragments (192 registers) per wave");

LRG_HD int own_unit(int r, int c, int u8) { return 64 * r + 8 * c + u8; }
LRG_HD int kc_unit(int c, int kk) { return 64 * (kk >> 3) + 8 * c + (kk & 7); }   // kk = 8 r' + u8 of column group c

// ---- forward -----------------------------------------------------------------------------------------------------
// element e of lane `lane` of the A fragment (tile 0..15 of the row group, k step q): W_hh[gate * H + uo][ui]
LRG_HD void fwd_w_elem(int r, int c, int tile, int q, int lane, int e, int& gate, int& uo, int& ui) {
  const int m = lane & 15;               // row of the tile = 4 * (unit of the tile) + gate
  gate = m & 3;
  uo = 64 * r + 4 * tile + (m >> 2);
  ui = kc_unit(c, 32 * q + 8 * (lane >> 4) + e);
}
// the accumulator of (tile, 16-sample sub-block sbb) in lane `lane`: registers 0..3 = gates i, f, g, o of ONE (sample,
// unit) -> destination column group cd (the member (r, cd) that owns the unit) and its item = sample * 8 + unit of the member
LRG_HD void fwd_acc_dest(int tile, int sbb, int lane, int& cd, int& item) {
  cd = tile >> 1;
  item = (16 * sbb + (lane & 15)) * 8 + 4 * (tile & 1) + (lane >> 4);
}
// cell thread <-> item: sample = item >> 3, unit of the member = item & 7; the h word of an item sits at word `item` of
// the member's 1 KB block.  Gather item it (0..63) of source row r': words 4 it .. 4 it + 3 = sample it >> 1, units
// 4 (it & 1) .. + 3 of that member -> state columns kk = 8 r' + 4 (it & 1) .. + 3
LRG_HD int h_gather_sample(int it) { return it >> 1; }
LRG_HD int h_gather_kk(int rs, int it) { return 8 * rs + 4 * (it & 1); }

// ---- backward ----------------------------------------------------------------------------------------------------
// the row group's dG in k order: kidx = 32 * (source column group cs) + 4 * u8 + gate
LRG_HD int dg_kidx(int cs, int u8, int gate) { return 32 * cs + 4 * u8 + gate; }
// element e of lane `lane` of the A fragment (tile 0..11 of the column group's units, k step q): W_hh[gate * H + uo][ui]
LRG_HD void bwd_w_elem(int r, int c, int tile, int q, int lane, int e, int& gate, int& uo, int& ui) {
  ui = kc_unit(c, 16 * tile + (lane & 15));
  const int kidx = 32 * q + 8 * (lane >> 4) + e;
  gate = kidx & 3;
  uo = 64 * r + 8 * (kidx >> 5) + ((kidx & 31) >> 2);
}
// the accumulator of (tile, sub-block) in lane `lane`: registers 0..3 = the partial dh of units 4 half .. 4 half + 3 of
// member (rd, c) for ONE sample -> destination row rd and its item = sample * 2 + half
LRG_HD void bwd_acc_dest(int tile, int sbb, int lane, int& rd, int& item) {
  const int kg = lane >> 4;
  rd = 2 * tile + (kg >> 1);
  item = (16 * sbb + (lane & 15)) * 2 + (kg & 1);
}

// ---- fragment order in memory (bf16x8 elements) ---------------------------------------------------------------------
// forward:  ((((member * 4 + wave) * 4 + tt) * FQ + q) * 2 + plane) * 64 + lane,  tile = 4 wave + tt
// backward: ((((member * 4 + wave) * 3 + jj) * BQ + q) * 2 + plane) * 64 + lane,  tile = 3 wave + jj
LRG_HD long fwd_frag_index(int member, int wave, int tt, int q, int plane, int lane) {
  return ((((long)(member * 4 + wave) * 4 + tt) * FQ + q) * 2 + plane) * 64 + lane;
}
LRG_HD long bwd_frag_index(int member, int wave, int jj, int q, int plane, int lane) {
  return ((((long)(member * 4 + wave) * 3 + jj) * BQ + q) * 2 + plane) * 64 + lane;
}
constexpr long FRAGS_PER_DIR = (long)NM * 4 * 48 * 64;   // bf16x8 elements: 37.7 MB

// ---- exchange areas (32-bit words), nsb = sample blocks of the launch -------------------------------------------------
// forward   HX [slot][c][sb][r][256]                the column group's h blocks (a reader sweeps 24 KB)
//           PX [slot][r][cd][sb][cs][1024]          partial gate sums for member (r, cd) from column group cs
// backward  GX [slot][r][sb][cs][1024]              the row group's dG blocks
//           DX [slot][c][rd][sb][rs][256]           partial dh for member (rd, c) from row group rs
// then NM words for the XCC-id handshake
LRG_HD long hx_index(int nsb, int slot, int c, int sb, int r) { return ((((long)slot * C + c) * nsb + sb) * R + r) * 256; }
LRG_HD long hx_words(int nsb) { return (long)2 * C * nsb * R * 256; }
LRG_HD long px_index(int nsb, int slot, int r, int cd, int sb, int cs) {
  return (((((long)slot * R + r) * C + cd) * nsb + sb) * C + cs) * 1024;
}
LRG_HD long px_words(int nsb) { return (long)2 * R * C * nsb * C * 1024; }
LRG_HD long gx_index(int nsb, int slot, int r, int sb, int cs) { return ((((long)slot * R + r) * nsb + sb) * C + cs) * 1024; }
LRG_HD long gx_words(int nsb) { return (long)2 * R * nsb * C * 1024; }
LRG_HD long dx_index(int nsb, int slot, int c, int rd, int sb, int rs) {
  return (((((long)slot * C + c) * R + rd) * nsb + sb) * R + rs) * 256;
}
LRG_HD long dx_words(int nsb) { return (long)2 * C * R * nsb * R * 256; }
LRG_HD long xch_words(int nsb, int backward) {
  return (backward ? gx_words(nsb) + dx_words(nsb) : hx_words(nsb) + px_words(nsb)) + NM;
}

}  // namespace lrg
