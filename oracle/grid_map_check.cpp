// grid_map_check.cpp — TEST INFRASTRUCTURE (never linked into the product): plays one forward and one backward step of
// the grid recurrence (lipreading_amd/csrc/lr_rnn_grid.hip) on the CPU through the kernel's OWN index functions
// (lr_rnn_grid_map.h) — fragment packing, the 16x16x32 MFMA's lane layout, publish, gather — and compares what every
// member's cell threads end up with against the plain products  gates = W_hh h  and  dh = W_hh^T dG
// (nn.LSTM's recurrent half, better_model.py:47-49).  Built and run by tests/test_grid_map.py with g++.
//   usage: grid_map_check H   (1152 < H <= 1536)   -> prints the two largest errors, exit code 0 when both are ~0
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../lipreading_amd/csrc/lr_rnn_grid_map.h"

using namespace lrg;

static unsigned long long rng_state = 88172645463325252ull;
static double rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return ((double)(rng_state % 17) - 8.0) / 8.0;   // exact in float: the comparison below is exact
}

int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 1400;
  if (H <= 0 || H > HP) return 2;
  std::vector<double> W((size_t)4 * H * H), h((size_t)SB * H), dG((size_t)SB * 4 * H);
  for (auto& v : W) v = rnd();
  for (auto& v : h) v = rnd();
  for (auto& v : dG) v = rnd();
  auto w_at = [&](int gate, int uo, int ui) { return (uo < H && ui < H) ? W[((size_t)gate * H + uo) * H + ui] : 0.0; };

  // ---- fragments, exactly as the pack kernels index them (one plane: the layout is what is checked) ---------------
  std::vector<float> ff((size_t)FRAGS_PER_DIR * 8, 0.f), bf((size_t)FRAGS_PER_DIR * 8, 0.f);   // (plane 0 slots only)
  for (int m = 0; m < NM; ++m) {
    const int r = m / C, c = m % C;
    for (int wave = 0; wave < 4; ++wave)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          for (int tt = 0; tt < 4; ++tt)
            for (int q = 0; q < FQ; ++q) {
              int gate, uo, ui;
              fwd_w_elem(r, c, 4 * wave + tt, q, lane, e, gate, uo, ui);
              ff[(size_t)fwd_frag_index(m, wave, tt, q, 0, lane) * 8 + e] = (float)w_at(gate, uo, ui);
            }
          for (int jj = 0; jj < 3; ++jj)
            for (int q = 0; q < BQ; ++q) {
              int gate, uo, ui;
              bwd_w_elem(r, c, 3 * wave + jj, q, lane, e, gate, uo, ui);
              bf[(size_t)bwd_frag_index(m, wave, jj, q, 0, lane) * 8 + e] = (float)w_at(gate, uo, ui);
            }
        }
  }

  // ---- forward step ------------------------------------------------------------------------------------------------
  // publish: member (r, c)'s h block, word `item` = sample * 8 + u8
  std::vector<double> HX((size_t)C * R * 256), PX((size_t)R * C * C * 1024, 0.0);
  for (int c = 0; c < C; ++c)
    for (int r = 0; r < R; ++r)
      for (int item = 0; item < 256; ++item) {
        const int u = own_unit(r, c, item & 7);
        HX[((size_t)c * R + r) * 256 + item] = u < H ? h[(size_t)(item >> 3) * H + u] : 0.0;
      }
  for (int m = 0; m < NM; ++m) {
    const int r = m / C, c = m % C;
    // gather: 24 sources x 64 items of four words -> the state slice [sample][kk]
    std::vector<double> hS((size_t)SB * KC);
    for (int rs = 0; rs < R; ++rs)
      for (int it = 0; it < 64; ++it)
        for (int j = 0; j < 4; ++j)
          hS[(size_t)h_gather_sample(it) * KC + h_gather_kk(rs, it) + j] = HX[((size_t)c * R + rs) * 256 + 4 * it + j];
    for (int wave = 0; wave < 4; ++wave)
      for (int tt = 0; tt < 4; ++tt)
        for (int sbb = 0; sbb < 2; ++sbb) {
          double D[16][16] = {};
          for (int q = 0; q < FQ; ++q)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 8; ++e) {
                const double a = ff[(size_t)fwd_frag_index(m, wave, tt, q, 0, lane) * 8 + e];   // A[lane % 16][8 (lane / 16) + e]
                const int k = 32 * q + 8 * (lane >> 4) + e;
                for (int n = 0; n < 16; ++n) D[lane & 15][n] += a * hS[(size_t)(16 * sbb + n) * KC + k];   // B[k][n]
              }
          for (int lane = 0; lane < 64; ++lane) {
            int cd, item;
            fwd_acc_dest(4 * wave + tt, sbb, lane, cd, item);
            for (int i = 0; i < 4; ++i) PX[(((size_t)r * C + cd) * C + c) * 1024 + item * 4 + i] = D[4 * (lane >> 4) + i][lane & 15];
          }
        }
  }
  double err_f = 0.0;
  for (int r = 0; r < R; ++r)
    for (int cd = 0; cd < C; ++cd)
      for (int item = 0; item < 256; ++item)
        for (int g = 0; g < 4; ++g) {
          double sum = 0.0;
          for (int cs = 0; cs < C; ++cs) sum += PX[(((size_t)r * C + cd) * C + cs) * 1024 + item * 4 + g];
          const int s = item >> 3, uo = own_unit(r, cd, item & 7);
          double ref = 0.0;
          if (uo < H)
            for (int ui = 0; ui < H; ++ui) ref += W[((size_t)g * H + uo) * H + ui] * h[(size_t)s * H + ui];
          err_f = std::fmax(err_f, std::fabs(sum - ref));
        }

  // ---- backward step -----------------------------------------------------------------------------------------------
  std::vector<double> GX((size_t)R * C * 1024), DX((size_t)C * R * R * 256, 0.0);
  for (int r = 0; r < R; ++r)
    for (int cs = 0; cs < C; ++cs)
      for (int item = 0; item < 256; ++item)
        for (int g = 0; g < 4; ++g) {
          const int u = own_unit(r, cs, item & 7);
          GX[((size_t)r * C + cs) * 1024 + item * 4 + g] = u < H ? dG[((size_t)(item >> 3) * 4 + g) * H + u] : 0.0;
        }
  for (int m = 0; m < NM; ++m) {
    const int r = m / C, c = m % C;
    std::vector<double> gS((size_t)SB * GR);
    for (int cs = 0; cs < C; ++cs)
      for (int item = 0; item < 256; ++item)
        for (int g = 0; g < 4; ++g)
          gS[(size_t)(item >> 3) * GR + dg_kidx(cs, item & 7, g)] = GX[((size_t)r * C + cs) * 1024 + item * 4 + g];
    for (int wave = 0; wave < 4; ++wave)
      for (int jj = 0; jj < 3; ++jj)
        for (int sbb = 0; sbb < 2; ++sbb) {
          double D[16][16] = {};
          for (int q = 0; q < BQ; ++q)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 8; ++e) {
                const double a = bf[(size_t)bwd_frag_index(m, wave, jj, q, 0, lane) * 8 + e];
                const int k = 32 * q + 8 * (lane >> 4) + e;
                for (int n = 0; n < 16; ++n) D[lane & 15][n] += a * gS[(size_t)(16 * sbb + n) * GR + k];
              }
          for (int lane = 0; lane < 64; ++lane) {
            int rd, item;
            bwd_acc_dest(3 * wave + jj, sbb, lane, rd, item);
            for (int i = 0; i < 4; ++i) DX[(((size_t)c * R + rd) * R + r) * 256 + item * 4 + i] = D[4 * (lane >> 4) + i][lane & 15];
          }
        }
  }
  double err_b = 0.0;
  for (int c = 0; c < C; ++c)
    for (int rd = 0; rd < R; ++rd)
      for (int s = 0; s < SB; ++s)
        for (int u8 = 0; u8 < 8; ++u8) {
          double sum = 0.0;
          for (int rs = 0; rs < R; ++rs) sum += DX[(((size_t)c * R + rd) * R + rs) * 256 + (s * 2 + (u8 >> 2)) * 4 + (u8 & 3)];
          const int ui = own_unit(rd, c, u8);
          double ref = 0.0;
          if (ui < H)
            for (int g = 0; g < 4; ++g)
              for (int uo = 0; uo < H; ++uo) ref += W[((size_t)g * H + uo) * H + ui] * dG[((size_t)s * 4 + g) * H + uo];
          err_b = std::fmax(err_b, std::fabs(sum - ref));
        }
  printf("H=%d forward max |err| %.3g, backward max |err| %.3g\n", H, err_f, err_b);
  return (err_f < 1e-9 && err_b < 1e-9) ? 0 : 1;
}
