// Probe (not product): run the grid recurrence's backward for a few steps on zero inputs and dump which exchange words
// carry which tag.  hipcc -O2 --offload-arch=gfx950 -I lipreading_amd/csrc tools/probes/grid_xch_dump.cpp -L lipreading_amd/_lib -llipreading_hip -o /tmp/grid_xch_dump
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "lr_common.h"
#include "lr_rnn_grid_map.h"
extern "C" int lr_rnn_pair_errors(void);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int H = 1536, B = argc > 1 ? atoi(argv[1]) : 4, T = argc > 2 ? atoi(argv[2]) : 1, D = 1;
  const int backward = argc > 3 ? atoi(argv[3]) : 1;
  float *W, *gates, *extra, *y, *dy, *dG;
  int32_t* lens;
  CK(hipMalloc(&W, (size_t)4 * H * H * 4));
  CK(hipMemset(W, 0, (size_t)4 * H * H * 4));
  const size_t n4 = (size_t)B * T * D * 4 * H, n1 = (size_t)B * T * D * H;
  CK(hipMalloc(&gates, n4 * 4)); CK(hipMemset(gates, 0, n4 * 4));
  CK(hipMalloc(&dG, n4 * 4)); CK(hipMemset(dG, 0, n4 * 4));
  CK(hipMalloc(&extra, n1 * 4)); CK(hipMemset(extra, 0, n1 * 4));
  CK(hipMalloc(&y, n1 * 4)); CK(hipMemset(y, 0, n1 * 4));
  CK(hipMalloc(&dy, n1 * 4)); CK(hipMemset(dy, 0, n1 * 4));
  std::vector<int32_t> hl(B, T);
  CK(hipMalloc(&lens, B * 4)); CK(hipMemcpy(lens, hl.data(), B * 4, hipMemcpyHostToDevice));
  void *wp, *xch;
  const size_t pb = lr_rnn_grid_pack_bytes(D), xb = lr_rnn_grid_xch_bytes(B, backward);
  CK(hipMalloc(&wp, pb)); CK(hipMalloc(&xch, xb));
  CK(hipMemset(xch, 0xff, xb));   // the prologue must clear it
  const float* whh[1] = {W};
  int st;
  if (backward) st = lr_rnn_grid_backward(gates, extra, dy, nullptr, nullptr, dG, nullptr, nullptr, nullptr, whh, lens, wp, xch, B, T, D, H, 0, 0);
  else st = lr_rnn_grid_forward(gates, extra, y, whh, nullptr, nullptr, lens, wp, xch, B, T, D, H, 0, 0);
  CK(hipDeviceSynchronize());
  printf("%s status %d, fault word %d, xch bytes %zu\n", backward ? "backward" : "forward", st, lr_rnn_pair_errors(), xb);
  std::vector<unsigned> h(xb / 4);
  CK(hipMemcpy(h.data(), xch, xb, hipMemcpyDeviceToHost));
  const int nsb = B <= 32 ? 1 : 2;
  using namespace lrg;
  // first area: GX (backward) / HX (forward): per 1 KB or 4 KB block, how many words carry tag 0/1/2/3
  const long a_words = backward ? gx_words(nsb) : hx_words(nsb), b_words = backward ? dx_words(nsb) : px_words(nsb);
  long cnt[2][4] = {{0}};
  for (long i = 0; i < a_words; ++i) cnt[0][h[i] & 3]++;
  for (long i = 0; i < b_words; ++i) cnt[1][h[a_words + i] & 3]++;
  printf("area A (%ld words) tags 0..3: %ld %ld %ld %ld\n", a_words, cnt[0][0], cnt[0][1], cnt[0][2], cnt[0][3]);
  printf("area B (%ld words) tags 0..3: %ld %ld %ld %ld\n", b_words, cnt[1][0], cnt[1][1], cnt[1][2], cnt[1][3]);
  if (backward) {   // GX [slot][r][sb][cs][1024]: untagged words per (slot 0, r, cs)
    for (int r = 0; r < R; ++r) {
      printf("r=%2d:", r);
      for (int cs = 0; cs < C; ++cs) {
        int z = 0;
        const long o = gx_index(nsb, 0, r, 0, cs);
        for (int k = 0; k < 1024; ++k) z += (h[o + k] & 3) == 0;
        printf(" %4d", z);
      }
      printf("\n");
    }
  }
  if (backward) {
    for (int blk = 0; blk < 3; ++blk) {
      printf("untagged items of GX block (r=%d, cs=%d):", blk, blk);
      const long o = gx_index(nsb, 0, blk, 0, blk);
      for (int it = 0; it < 256; ++it) {
        int z = 0;
        for (int k = 0; k < 4; ++k) z += (h[o + 4 * it + k] & 3) == 0;
        if (z && it < 32) printf(" %d(%d)", it, z);
      }
      printf("\n  items 10..17 words:");
      for (int it = 10; it < 18; ++it) printf(" [%x %x %x %x]", h[o + 4 * it], h[o + 4 * it + 1], h[o + 4 * it + 2], h[o + 4 * it + 3]);
      printf("\n");
    }
  }
  long xid0 = a_words + b_words;
  printf("xid:");
  for (int m = 0; m < 16; ++m) printf(" %x", h[xid0 + m]);
  printf("\n");
  return 0;
}
