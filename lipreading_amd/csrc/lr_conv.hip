// lr_conv.hip — 3-D convolution frontend (pixels -> per-frame features) on the gfx950 bf16
// matrix cores.  BUILD-DEFINED: the reference has no conv frontend (its conv stack is commented
// out, src/models/lipreader/model.py:122,153-156; the `ced` configs are empty files), so there is
// no reference arithmetic to match — the specification is this repo's and the
// oracle is torch.nn.functional.conv3d / max_pool3d on the CPU.  BASELINE.json's north_star asks for
// "im2col + MFMA GEMM for the 3D convs": the im2col matrix is never materialised (6.6 GB for the
// second layer at B=32); each workgroup gathers its (128 output pixels x 32 k) slice of it
// straight into LDS — an implicit GEMM — and multiplies with v_mfma_f32_32x32x16_bf16.
//
// Layouts: activations are channels-last bf16, [B][T][H][W][C] (C = 4 for the padded RGB input,
// else the channel count, a multiple of 32), so the K axis of the implicit GEMM, k = (tap, c)
// with tap = (kt,kh,kw), reads C contiguous values per tap.  Weights are repacked per step from
// the fp32 torch layout [Cout][Cin][kt][kh][kw] into bf16 [N][taps][Cpad] (forward) and into the
// flipped/transposed form that turns the data gradient of a stride-1 convolution into the same
// forward kernel.  Accumulation is fp32; bias + ReLU are fused into the forward epilogue.
//
// Weight gradient: dW[n][k] = sum_m dZ[m][n] * im2col[m][k] contracts over PIXELS, which are the
// slow axis of both operands; tiles are staged pixel-major in LDS and the MFMA fragments are read
// down the columns (ds_read_u16).  K is split over workgroups into fp32 slabs that a second kernel
// reduces in fixed order (deterministic) into the torch-layout gradient.
//
// This file: the generic implicit-GEMM kernels, layer 3's patch-resident kernel, the elementwise stages, weight
// packing, the slab reduction and the C ABI's dispatch.  The kernels the metric's geometry actually runs have
// translation units of their own: lr_conv1.hip (layer 1), lr_conv_patch.hip (layer 2 forward / data gradient),
// lr_conv_wgrad.hip (weight gradient of layers 2 and 3); lr_conv_dev.h holds what they share.
#include "lr_common.h"
#include "lr_conv_dev.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include <cstdlib>

namespace {

// sum_{z<n} p[z*stride] with 8 loads in flight; the 8 partial sums are combined in a fixed order, so
// the result is deterministic (it does not depend on scheduling).
__device__ __forceinline__ float strided_sum8(const float* __restrict__ p, int n, int64_t stride) {
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int z = 0;
  for (; z + 8 <= n; z += 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] += p[(int64_t)(z + i) * stride];
  }
  for (; z < n; ++z) a[z & 7] += p[(int64_t)z * stride];
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

struct ConvGeom {
  int B, T, Hin, Win, Cin;   // Cin = stored (padded) input channels
  int Ho, Wo, Cout;
  int KT, KH, KW;            // taps
  int stride, pt, ph, pw;    // spatial stride, paddings (temporal stride is 1)
  int Ktot;                  // KT*KH*KW*Cin
  int64_t M;                 // B*T*Ho*Wo
};


// Gather one 8-byte unit (4 consecutive k) of im2col row `m` at k index `k` (k % 4 == 0).
// The row's output coordinates are pre-decoded into (b*T + t, hi0, wi0).
struct RowCoord {
  int bt, t, hi0, wi0;
  bool ok;
};
__device__ __forceinline__ RowCoord decode_row(const ConvGeom& g, int64_t m) {
  RowCoord r;
  r.ok = m < g.M;
  const int64_t mm = r.ok ? m : 0;
  const int wo = (int)(mm % g.Wo);
  const int64_t q = mm / g.Wo;
  const int ho = (int)(q % g.Ho);
  r.bt = (int)(q / g.Ho);
  r.t = r.bt % g.T;
  r.hi0 = ho * g.stride - g.ph;
  r.wi0 = wo * g.stride - g.pw;
  return r;
}
__device__ __forceinline__ uint2 gather_unit(const ConvGeom& g, const bf16_t* __restrict__ X,
                                             const RowCoord& r, int k) {
  uint2 v = make_uint2(0u, 0u);
  if (!r.ok || k >= g.Ktot) return v;
  const int tap = k / g.Cin, c = k - tap * g.Cin;
  const int kw = tap % g.KW;
  const int kh = (tap / g.KW) % g.KH;
  const int kt = tap / (g.KW * g.KH);
  const int ti = r.t + kt - g.pt, hi = r.hi0 + kh, wi = r.wi0 + kw;
  if (ti < 0 || ti >= g.T || hi < 0 || hi >= g.Hin || wi < 0 || wi >= g.Win) return v;
  const int64_t off = ((((int64_t)(r.bt + kt - g.pt)) * g.Hin + hi) * g.Win + wi) * g.Cin + c;
  return *reinterpret_cast<const uint2*>(X + off);
}

// ---------------------------------------------------------------------------------------------
// forward / data-gradient implicit GEMM:  Y[m][n] = act( sum_k im2col(X)[m][k] * Wp[n][k] + b[n] )
// ---------------------------------------------------------------------------------------------
// Workgroup tile: 256 output pixels x Cout (all of N: Cout <= 96), K staged 64 at a time.
// 4 waves, each 64 pixels (two 32-row MFMA tiles) so every weight fragment read from LDS feeds
// two MFMAs; a 64-k stage is 8*NT MFMAs per wave between barriers.  The im2col slice is gathered
// in 16-byte units (8 channels of one tap; 8-byte units for the 4-channel input layer), lanes
// running along k so a row's slice is one contiguous 128-byte read when the stage stays inside
// a tap.  Tap -> (dt,dh,dw) comes from a small LDS table instead of divisions.  LDS rows are
// padded to 144 B, which makes the ds_read_b128 fragment reads conflict-free (36-dword stride
// covers all 64 banks within each 16-lane group).
constexpr int IG_BM = 256;
constexpr int IG_BK = 64;
constexpr int IG_LD = IG_BK + 8;   // bf16 per LDS row (144 B)

template <int CIN, int NT>
__global__ __launch_bounds__(256) void conv3d_igemm_kernel(ConvGeom g, const bf16_t* __restrict__ X,
                                                           const bf16_t* __restrict__ Wp,
                                                           const float* __restrict__ bias,
                                                           bf16_t* __restrict__ Y, int relu) {
  constexpr int UE = CIN >= 8 ? 8 : 4;               // bf16 elements per gather unit (16 B / 8 B)
  constexpr int UPR = IG_BK / UE;                    // units per row per stage (8 or 16)
  constexpr int A_PER = IG_BM * UPR / 256;           // units per thread per stage (8 or 16)
  constexpr int ROWS_PER_PASS = 256 / UPR;           // rows covered by one pass of the workgroup
  constexpr int B_UNITS = NT * 32 * (IG_BK / 8);     // 16-byte units of the weight stage
  constexpr int B_PER = (B_UNITS + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf16_t As[IG_BM * IG_LD];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[NT * 32 * IG_LD];
  __shared__ int tap_dt[128], tap_dh[128], tap_dw[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * IG_BM;
  const int taps = g.KT * g.KH * g.KW;
  for (int tp = tid; tp < taps; tp += 256) {
    tap_dw[tp] = tp % g.KW;
    tap_dh[tp] = (tp / g.KW) % g.KH;
    tap_dt[tp] = tp / (g.KW * g.KH) - g.pt;
  }

  // this thread's rows: row = tid / UPR + ROWS_PER_PASS * i, unit column = tid % UPR
  const int ucol = tid % UPR;
  const int row0 = tid / UPR;
  constexpr int NROW = A_PER;
  int r_bt[NROW], r_t[NROW], r_hi[NROW], r_wi[NROW];
#pragma unroll
  for (int i = 0; i < NROW; ++i) {
    const RowCoord rc = decode_row(g, m0 + row0 + ROWS_PER_PASS * i);
    r_bt[i] = rc.ok ? rc.bt : -1;
    r_t[i] = rc.t;
    r_hi[i] = rc.hi0;
    r_wi[i] = rc.wi0;
  }
  __syncthreads();

  typedef typename std::conditional<UE == 8, uint4, uint2>::type unit_t;
  unit_t ra[A_PER];
  uint4 rb[B_PER];
  auto load_stage = [&](int k0) {
    const int k = k0 + ucol * UE;
    const int tap = k / CIN, c = k - tap * CIN;
    const bool kok = k < g.Ktot;
    const int dt = kok ? tap_dt[tap] : 0, dh = kok ? tap_dh[tap] : 0, dw = kok ? tap_dw[tap] : 0;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      unit_t v;
      __builtin_memset(&v, 0, sizeof(v));
      const int ti = r_t[i] + dt, hi = r_hi[i] + dh, wi = r_wi[i] + dw;
      if (kok && r_bt[i] >= 0 && ti >= 0 && ti < g.T && hi >= 0 && hi < g.Hin && wi >= 0 && wi < g.Win)
        v = *reinterpret_cast<const unit_t*>(X + ((((int64_t)(r_bt[i] + dt)) * g.Hin + hi) * g.Win + wi) * CIN + c);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int e = tid + i * 256;
      rb[i] = make_uint4(0u, 0u, 0u, 0u);
      if (e < B_UNITS) {
        const int n = e / (IG_BK / 8), kk = k0 + (e % (IG_BK / 8)) * 8;
        if (kk < g.Ktot) rb[i] = *reinterpret_cast<const uint4*>(Wp + (int64_t)n * g.Ktot + kk);
      }
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      *reinterpret_cast<unit_t*>(&As[(row0 + ROWS_PER_PASS * i) * IG_LD + ucol * UE]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int e = tid + i * 256;
      if (e < B_UNITS) *reinterpret_cast<uint4*>(&Bs[(e / (IG_BK / 8)) * IG_LD + (e % (IG_BK / 8)) * 8]) = rb[i];
    }
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lk = lane >> 5;
  load_stage(0);
  for (int k0 = 0; k0 < g.Ktot; k0 += IG_BK) {
    __syncthreads();
    store_stage();
    __syncthreads();
    if (k0 + IG_BK < g.Ktot) load_stage(k0 + IG_BK);
#pragma unroll
    for (int kk = 0; kk < IG_BK; kk += 16) {
      bf16x8 a[2], b[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(&As[(wave * 64 + i * 32 + lr) * IG_LD + kk + lk * 8]);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(&Bs[(j * 32 + lr) * IG_LD + kk + lk * 8]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 32 + lr;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m >= g.M) continue;
        float v = acc[i][j][r] + bv;
        if (relu) v = fmaxf(v, 0.f);
        Y[m * g.Cout + n] = f2bf(v);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  slab[split][n][k] = sum_{m in split} dZ[m][n] * im2col(X)[m][k]
// ---------------------------------------------------------------------------------------------
// One workgroup = one (32*NTW)-wide k tile x one pixel range; all Cout rows.  Every stage brings
// 128 pixels of dZ and of the im2col slice into LDS (pixel-major, as they lie in memory); wave w
// contracts pixels [32w, 32w+32) of the stage into its own MT x NTW accumulator tiles (each dZ
// fragment is reused NTW times, each im2col fragment MT times), and the four waves' partial
// tiles are summed through LDS at the end in fixed order.  The contraction runs over PIXELS, the
// slow axis of both operands, so fragments are read down LDS columns (ds_read_u16).
constexpr int WG_PIX = 128;

template <int CIN, int MT, int NTW>
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(ConvGeom g, const bf16_t* __restrict__ X,
                                                           const bf16_t* __restrict__ dZ,
                                                           float* __restrict__ slabs,
                                                           int64_t pix_per_split) {
  constexpr int NC = NTW * 32;               // k columns per workgroup
  constexpr int UE = CIN >= 8 ? 8 : 4;       // bf16 per gather unit
  constexpr int UPR = NC / UE;               // units per pixel row
  constexpr int XLD = NC + 8;                // bf16 per LDS row, rows 16-byte aligned
  constexpr int ZLD = MT * 32 + 8;
  constexpr int ZU = MT * 4;                 // 16-byte units per dZ row
  __shared__ __attribute__((aligned(16))) bf16_t Xs[WG_PIX * XLD];
  __shared__ __attribute__((aligned(16))) bf16_t Zs[WG_PIX * ZLD];
  __shared__ int rbt[WG_PIX], rt[WG_PIX], rhi[WG_PIX], rwi[WG_PIX];
  __shared__ int tap_dt[128], tap_dh[128], tap_dw[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = blockIdx.x * NC;
  const int64_t mbeg = (int64_t)blockIdx.y * pix_per_split;
  const int64_t mend = min(g.M, mbeg + pix_per_split);
  const int lr = lane & 31, lk = lane >> 5;
  const int taps = g.KT * g.KH * g.KW;
  for (int tp = tid; tp < taps; tp += 256) {
    tap_dw[tp] = tp % g.KW;
    tap_dh[tp] = (tp / g.KW) % g.KH;
    tap_dt[tp] = tp / (g.KW * g.KH) - g.pt;
  }
  typedef typename std::conditional<UE == 8, uint4, uint2>::type unit_t;

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int64_t m0 = mbeg; m0 < mend; m0 += WG_PIX) {
    __syncthreads();   // previous stage consumed (and the tap table is visible)
    if (tid < WG_PIX) {
      const int64_t m = m0 + tid;
      const RowCoord rc = decode_row(g, m < mend ? m : g.M);
      rbt[tid] = rc.ok ? rc.bt : -1;
      rt[tid] = rc.t;
      rhi[tid] = rc.hi0;
      rwi[tid] = rc.wi0;
    }
    // dZ rows: the 128 x Cout tile is one contiguous block of memory
    for (int e = tid; e < WG_PIX * ZU; e += 256) {
      const int row = e / ZU, u = e - row * ZU;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (m0 + row < mend) v = *reinterpret_cast<const uint4*>(dZ + (m0 + row) * g.Cout + u * 8);
      *reinterpret_cast<uint4*>(&Zs[row * ZLD + u * 8]) = v;
    }
    __syncthreads();   // row coordinates ready
    for (int e = tid; e < WG_PIX * UPR; e += 256) {
      const int row = e / UPR, u = e - row * UPR;
      const int k = k0 + u * UE;
      unit_t v;
      __builtin_memset(&v, 0, sizeof(v));
      if (k < g.Ktot && rbt[row] >= 0) {
        const int tap = k / CIN, c = k - tap * CIN;
        const int dt = tap_dt[tap];
        const int ti = rt[row] + dt, hi = rhi[row] + tap_dh[tap], wi = rwi[row] + tap_dw[tap];
        if (ti >= 0 && ti < g.T && hi >= 0 && hi < g.Hin && wi >= 0 && wi < g.Win)
          v = *reinterpret_cast<const unit_t*>(X + ((((int64_t)(rbt[row] + dt)) * g.Hin + hi) * g.Win + wi) * CIN + c);
      }
      *reinterpret_cast<unit_t*>(&Xs[row * XLD + u * UE]) = v;
    }
    __syncthreads();
    // wave w contracts pixels [32w, 32w+32): two MFMA k-steps of 16 pixels
#pragma unroll
    for (int kk = 0; kk < 32; kk += 16) {
      const int p0 = wave * 32 + kk + lk * 8;
      bf16x8 a[MT], b[NTW];
#pragma unroll
      for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          a[j][i] = __builtin_bit_cast(__bf16, Zs[(p0 + i) * ZLD + j * 32 + lr]);   // A[row=n][k=pixel]
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          b[j][i] = __builtin_bit_cast(__bf16, Xs[(p0 + i) * XLD + j * 32 + lr]);   // B[k=pixel][col]
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // combine the four waves' partial tiles (fixed order) and write the slab
  float* red = reinterpret_cast<float*>(Xs);   // 4 x 32 x 33 floats = 16.9 kB <= sizeof(Xs)
  static_assert(sizeof(bf16_t) * WG_PIX * XLD >= 4 * 32 * 33 * sizeof(float), "reduction scratch");
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r)
        red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * 33 + lr] = acc[i][j][r];
      __syncthreads();
      for (int e = tid; e < 32 * 32; e += 256) {
        const int row = e >> 5, col = e & 31;
        const float sum = red[row * 33 + col] + red[(32 + row) * 33 + col] + red[(64 + row) * 33 + col] +
                          red[(96 + row) * 33 + col];
        const int k = k0 + j * 32 + col;
        if (k < g.Ktot) slabs[((int64_t)blockIdx.y * g.Cout + i * 32 + row) * g.Ktot + k] = sum;
      }
    }
}

// ---------------------------------------------------------------------------------------------
// tap-stationary weight gradient for stride-1 "same" layers with Cin in {32,64} (layers 2 and 3)
// ---------------------------------------------------------------------------------------------
// dW[n][tap][c] = sum_pixels dZ[pix][n] * X[pix + tap][c].  Instead of building im2col rows (each
// input value would be gathered once per tap), a workgroup keeps the accumulators of ALL taps of
// one temporal offset kt resident in registers and walks over spatial tiles: for every tile it
// brings the dZ tile and ONE input patch (tile + halo, frame t+kt-pt) into LDS; every tap's
// operand is then just a shifted window of that patch.  Work split: unit = (tap, 32-channel block
// of Cin); wave w owns units w, w+4, ... with all MT row tiles, so each dZ fragment read from LDS
// feeds UPW*MT MFMAs.  Tiles are TY rows x the full (8-padded) width, pixels row-major, so an
// 8-pixel k group never straddles a row.  The next tile is prefetched into registers while the
// current one is contracted.  Per-workgroup partials go to slabs, reduced in fixed order.
template <int CIN, int MT, int UPW>
__global__ __launch_bounds__(256) void conv3d_wgrad_ts_kernel(ConvGeom g, const bf16_t* __restrict__ X,
                                                              const bf16_t* __restrict__ dZ,
                                                              float* __restrict__ slabs, int TY, int TXP,
                                                              int wgs_per_kt) {
  constexpr int NTC = CIN / 32;
  constexpr int ZLD = MT * 32 + 8;
  constexpr int PLD = CIN + 8;
  constexpr int ZU = MT * 4;        // 16-byte units per dZ pixel
  constexpr int PU = CIN / 8;       // 16-byte units per patch pixel
  constexpr int MAXU = 6;           // prefetch registers per operand (uint4 each)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int NPIX = TY * TXP;
  const int PH = TY + g.KH - 1, PW = TXP + g.KW - 1;
  bf16_t* Zs = reinterpret_cast<bf16_t*>(smem_raw);              // [NPIX][ZLD]
  bf16_t* Ps = Zs + (size_t)NPIX * ZLD;                           // [PH][PW][PLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lk = lane >> 5;
  const int kt = blockIdx.x % g.KT, slot = blockIdx.x / g.KT;
  const int H = g.Ho, W = g.Wo;
  const int RB = (H + TY - 1) / TY;
  const int64_t ntiles = (int64_t)g.B * g.T * RB;
  const int khw = g.KH * g.KW, units = khw * NTC;
  const int nz_units = NPIX * ZU, np_units = PH * PW * PU;

  // this wave's units: (kh, kw, nt) -> patch offset of pixel (0,0) and column block
  int u_off[UPW], u_tap[UPW], u_nt[UPW];
#pragma unroll
  for (int j = 0; j < UPW; ++j) {
    const int u = wave + 4 * j;
    const int tap = u < units ? u / NTC : 0;
    u_tap[j] = tap;
    u_nt[j] = u % NTC;
    u_off[j] = ((tap / g.KW) * PW + (tap % g.KW)) * PLD + u_nt[j] * 32 + lr;
  }
  f32x16 acc[UPW][MT];
#pragma unroll
  for (int j = 0; j < UPW; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  uint4 rz[MAXU], rp[MAXU];
  auto tile_valid = [&](int64_t q) -> bool {
    const int f = (int)(q / RB);
    const int ti = f % g.T + kt - g.pt;
    return ti >= 0 && ti < g.T;
  };
  auto load_tile = [&](int64_t q) {
    const int f = (int)(q / RB), y0 = (int)(q % RB) * TY;
    const int fi = f + kt - g.pt;
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
      const int e = tid + i * 256;
      rz[i] = make_uint4(0u, 0u, 0u, 0u);
      if (e < nz_units) {
        const int pix = e / ZU, u = e - pix * ZU;
        const int yl = pix / TXP, x = pix - yl * TXP;
        if (y0 + yl < H && x < W)
          rz[i] = *reinterpret_cast<const uint4*>(dZ + (((int64_t)f * H + y0 + yl) * W + x) * g.Cout + u * 8);
      }
      rp[i] = make_uint4(0u, 0u, 0u, 0u);
      if (e < np_units) {
        const int pp = e / PU, u = e - pp * PU;
        const int r = pp / PW, cx = pp - r * PW;
        const int yi = y0 + r - g.ph, xi = cx - g.pw;
        if (yi >= 0 && yi < H && xi >= 0 && xi < W)
          rp[i] = *reinterpret_cast<const uint4*>(X + (((int64_t)fi * H + yi) * W + xi) * CIN + u * 8);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
      const int e = tid + i * 256;
      if (e < nz_units) *reinterpret_cast<uint4*>(&Zs[(e / ZU) * ZLD + (e % ZU) * 8]) = rz[i];
      if (e < np_units) *reinterpret_cast<uint4*>(&Ps[(e / PU) * PLD + (e % PU) * 8]) = rp[i];
    }
  };

  // first valid tile of this workgroup
  int64_t q = slot;
  while (q < ntiles && !tile_valid(q)) q += wgs_per_kt;
  if (q < ntiles) load_tile(q);
  while (q < ntiles) {
    __syncthreads();          // everyone is done reading the previous tile
    store_tile();
    __syncthreads();
    int64_t qn = q + wgs_per_kt;
    while (qn < ntiles && !tile_valid(qn)) qn += wgs_per_kt;
    if (qn < ntiles) load_tile(qn);   // in flight while the MFMAs below run
    const int nks = NPIX >> 4;
    for (int ks = 0; ks < nks; ++ks) {
      const int p0 = ks * 16 + lk * 8;
      const int yl = p0 / TXP, x0 = p0 - yl * TXP;
      bf16x8 a[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          a[i][e] = __builtin_bit_cast(__bf16, Zs[(p0 + e) * ZLD + i * 32 + lr]);
      const int pbase = (yl * PW + x0) * PLD;
#pragma unroll
      for (int j = 0; j < UPW; ++j) {
        if (wave + 4 * j < units) {
          bf16x8 b;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            b[e] = __builtin_bit_cast(__bf16, Ps[pbase + e * PLD + u_off[j]]);
#pragma unroll
          for (int i = 0; i < MT; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b, acc[j][i], 0, 0, 0);
        }
      }
    }
    q = qn;
  }
  // partial result of this workgroup: slabs[wg][tap][n][c]
  float* out = slabs + (int64_t)blockIdx.x * khw * g.Cout * CIN;
#pragma unroll
  for (int j = 0; j < UPW; ++j) {
    if (wave + 4 * j < units) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          out[((int64_t)u_tap[j] * g.Cout + n) * CIN + u_nt[j] * 32 + lr] = acc[j][i][r];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// patch-resident forward / data-gradient kernel for the 12x12 stride-1 (3,3,3) layer
// ---------------------------------------------------------------------------------------------
// Same idea as conv_patch_kernel, shaped for 12x12 frames: a wave owns a whole frame as nine
// 4x4-position row tiles of v_mfma_f32_16x16x32_bf16 (16 rows x 32 k: one MFMA consumes a whole
// 32-channel group of a tap), a workgroup 4 consecutive frames; the 6 x 14 x 14-position patch of
// one 32-channel group sits in LDS as rows of 14 positions x 64 B + 16 B of padding (P3_RS = 912 B); every
// fragment address of the kernel is then ONE lane base + an immediate offset (< 64 KB), no per-read swizzle
// arithmetic.  With ds_read_b128's fixed lane groups ({0-3, 12-15, 20-27}, ...) this layout is 2-way bank
// conflicted (no row padding avoids that); the conflict-free row swizzle c ^ (2 (row & 1)) on 1024-byte rows
// was measured too and is SLOWER here (forward 85 -> 95 us: the LDS pipe is not what bounds this kernel, the
// second address register and the larger patch are paid for nothing), so the padded rows stay.  Inputs
// with 64 / 96 channels take 2 / 3 passes.  B fragments (16 output channels x 32 k) come from
// global in the 16-column fragment-major packing.
// A workgroup covers P3_TT = 2 consecutive frames with TWO waves per frame (each takes half of the
// output channels: NT16 sixteen-column tiles per wave), so its patch is 4 slots = 57 KB and two
// workgroups share a CU: 8 waves per CU overlap each other's load phases and epilogues, and 1200
// half-size tiles fill 256 CUs in 2.5 tile-times where 600 four-frame tiles took 3 (2.34 rounds).
constexpr int P3_TT = 2, P3_NSPL = 4 / P3_TT, P3_H = 12, P3_W = 12, P3_PH = P3_H + 2, P3_SLOTS = P3_TT + 2;
constexpr int P3_RS = (P3_W + 2) * 64 + 16;        // bytes per patch row
constexpr int P3_ROWS = P3_SLOTS * P3_PH;          // 56 patch rows: one per wave and pass, 14 passes
constexpr int P3_LDS = P3_ROWS * P3_RS;            // 51,072 bytes
static_assert(P3_ROWS % 28 == 0, "patch rows are loaded in batches of 7 passes x 4 waves");

// (Measured and dropped, round 4: three workgroups per CU again after the fragment ring took the forward's registers
// from 161 to 185 — 17 spilled registers in its patch fill, forward 86.9 -> 95.0 us.  The ring itself against the
// form before it, same visit: forward 84.7 -> 86.9 us, data gradient 99.2 -> 91.6: kept.)
// UNPOOL (data gradient of a layer whose forward fused ReLU + MaxPool): X is the POOLED gradient dP [F][6][6][C] and
// `code` the windows' codes; the fill rebuilds the dZ patch on the way into LDS (unpool8; see conv_patch_tile in
// lr_conv_patch.hip): no un-pooling kernel in front of this one, no dZ in memory.
template <int CG, int NT16, bool POOL, bool UNPOOL>
__global__ __launch_bounds__(256, 2) void conv_patch16_kernel(const bf16_t* __restrict__ X,
                                                              const bf16_t* __restrict__ Wf,
                                                              const float* __restrict__ bias,
                                                              bf16_t* __restrict__ Y, unsigned char* __restrict__ code,
                                                              int F, int T, int relu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char patch[];
  constexpr int NTT = NT16 * P3_NSPL;   // sixteen-column tiles of the whole layer
  constexpr int C = 32 * CG, N = 16 * NTT, TAPS = 27;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fw = wave % P3_TT, nh = wave / P3_TT;   // frame of the tile, slice of the output channels
  // XCD-aware tile order (as conv_patch_kernel): XCD b % 8 takes a contiguous run of frame tiles, whose temporal
  // halo frames it then finds in its own L2
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3, rem_xcd = gridDim.x & 7;
  const int f0 = (xcd * per_xcd + (xcd < rem_xcd ? xcd : rem_xcd) + (int)(blockIdx.x >> 3)) * P3_TT;
  const int rl = lane & 15, kg = lane >> 4;
  const int f = f0 + fw;
  const bool fvalid = f < F;
  const int t = f % T;

  f32x4 acc[9][NT16];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < NT16; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // row tile mb = (hb, wb); A row i = lane & 15 is pixel (h, w) = (4hb + b1 + 2 b3, 4wb + b0 + 2 b2)
  // (b_k = bit k of i), so the four D rows 4kg .. 4kg+3 a lane holds are the 2 x 2 pooling window
  // (h = 2(kg>>1) + {0,1}, w = 2(kg&1) + {0,1}) in scan order and the pooled epilogue is lane-local.
  // The 16 lanes ds_read_b128 serves together still cover the 4 x 4 block once, and same-column
  // lanes land on 4 different chunks for any tap shift.
  const int base_b = (fw * P3_PH + ((rl >> 1) & 1) + 2 * (rl >> 3)) * P3_RS + ((rl & 1) + 2 * ((rl >> 2) & 1)) * 64 + kg * 16;
  // B fragments: a ring of THREE taps' fragments, indexed by dw (compile-time), running through the channel passes as
  // one sequence g = cg * 27 + tap: the fragments of g + 3 are loaded into the registers of g right after g's MFMAs.
  // (Round 2 / 3 loaded tap + 1 during tap and copied next -> current at the end of the tap: the copy READS the
  // registers the load is filling, so the wave waited for the load inside the tap it was issued in — 288 (data
  // gradient) / 432 (forward) MFMA cycles of cover for an L2 round trip, and a cold start after every patch fill:
  // SQ_WAIT_ANY = 51 % / 37 % of the kernels' wave cycles, profiles/r03_pixels_pmc_SQ_pass3.txt.)  The first three
  // taps are in flight before the first patch fill, a pass's last row loads the next pass's first taps.
  const bf16_t* wfl = Wf + (int64_t)nh * NT16 * 512 + lane * 8;   // + (g * NTT + j) * 512
  bf16x8 br[3][NT16];
#pragma unroll
  for (int dw = 0; dw < 3; ++dw)
#pragma unroll
    for (int j = 0; j < NT16; ++j) br[dw][j] = *reinterpret_cast<const bf16x8*>(wfl + (dw * NTT + j) * 512);
  int g = 0;
  if constexpr (UNPOOL) {
    // halo rows (0 and 13 of every slot) and halo columns (positions 0 and 13) are zeros for every channel group
    for (int u = tid; u < P3_ROWS * (P3_W + 2) * 4; u += 256) {
      const int R = u / ((P3_W + 2) * 4), q = u - R * ((P3_W + 2) * 4), pos = q >> 2, ph = R % P3_PH;
      if (ph == 0 || ph == P3_PH - 1 || pos == 0 || pos == P3_W + 1)
        *reinterpret_cast<uint4*>(patch + R * P3_RS + pos * 64 + (q & 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  for (int cg = 0; cg < CG; ++cg) {
    if (cg > 0) __syncthreads();
    // patch load: a wave covers one 16-position patch row (64 sixteen-byte units) per pass, so the
    // row / slot decode is wave-uniform scalar work; batches of 7 passes, each fully in flight
    if constexpr (UNPOOL) {
      // pooled units of the patch: 4 slots x 6 x 6 windows x 4 chunks of 8 channels = 576, dealt flat over the threads;
      // a window's four units go to patch rows 2 hp + 1, 2 hp + 2, positions 2 wp + 1, 2 wp + 2 (the halo rows and
      // columns were zeroed once)
      constexpr int UNITS = P3_SLOTS * 36 * 4, UI = (UNITS + 255) / 256;
      uint4 dv[UI];
      uint2 cv[UI];
#pragma unroll
      for (int i = 0; i < UI; ++i) {
        const int u = tid + 256 * i, s = u / 144, r = u - 144 * s;
        const int ff = f0 - 1 + s;
        dv[i] = make_uint4(0u, 0u, 0u, 0u);
        cv[i] = make_uint2(0u, 0u);
        if (u < UNITS && ff >= 0 && ff < F) {
          const int64_t pi = ((int64_t)ff * 36 + (r >> 2)) * C + cg * 32 + (r & 3) * 8;
          dv[i] = *reinterpret_cast<const uint4*>(X + pi);
          cv[i] = *reinterpret_cast<const uint2*>(code + pi);
        }
      }
#pragma unroll
      for (int i = 0; i < UI; ++i) {
        const int u = tid + 256 * i, s = u / 144, r = u - 144 * s;
        if (u < UNITS) {
          uint4 o[4];
          unpool8(dv[i], cv[i], o);
          const int win = r >> 2, hp = win / 6, wp = win - 6 * hp;
          unsigned char* dst = patch + (s * P3_PH + 2 * hp + 1) * P3_RS + (2 * wp + 1) * 64 + (r & 3) * 16;
          *reinterpret_cast<uint4*>(dst) = o[0];
          *reinterpret_cast<uint4*>(dst + 64) = o[1];
          *reinterpret_cast<uint4*>(dst + P3_RS) = o[2];
          *reinterpret_cast<uint4*>(dst + P3_RS + 64) = o[3];
        }
      }
    } else {
      const int pw = (tid & 63) >> 2, c = tid & 3;
      const bool tvalid = pw >= 1 && pw <= P3_W;
      const bf16_t* xt = X + ((int64_t)(pw - 1)) * C + cg * 32 + c * 8;
#pragma unroll
      for (int part = 0; part < P3_ROWS / 28; ++part) {
        uint4 v[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int R = 4 * (part * 7 + i) + wave;          // patch row 0..83
          const int s = R / P3_PH, ph = R - s * P3_PH;
          const int ff = f0 - 1 + s, hh = ph - 1;
          v[i] = make_uint4(0u, 0u, 0u, 0u);
          if (tvalid && ff >= 0 && ff < F && hh >= 0 && hh < P3_H)
            v[i] = *reinterpret_cast<const uint4*>(xt + ((int64_t)ff * P3_H + hh) * (P3_W * C));
        }
        if (pw < P3_W + 2) {   // 14 positions per row (lanes of positions 14, 15 idle)
#pragma unroll
          for (int i = 0; i < 7; ++i)
            *reinterpret_cast<uint4*>(patch + (4 * (part * 7 + i) + wave) * P3_RS + pw * 64 + c * 16) = v[i];
        }
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int dt = 0; dt < 3; ++dt) {
      const bool valid = fvalid && t + dt - 1 >= 0 && t + dt - 1 < T;   // wave-uniform
#pragma unroll 1
      for (int dh = 0; dh < 3; ++dh) {   // (unrolled further, the 243 fragment addresses get hoisted and spilled)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw, ++g) {
          if (valid) {
            const int to = (dt * P3_PH + dh) * P3_RS + dw * 64;
#pragma unroll
            for (int mb = 0; mb < 9; ++mb) {
              const bf16x8 a = *reinterpret_cast<const bf16x8*>(patch + base_b + (to + (4 * (mb / 3)) * P3_RS + 4 * (mb % 3) * 64));
#pragma unroll
              for (int j = 0; j < NT16; ++j)
                acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, br[dw][j], acc[mb][j], 0, 0, 0);
            }
          }
          const int gn = g + 3 < CG * TAPS ? g + 3 : CG * TAPS - 1;   // (the last three reload the last tap: no branch)
#pragma unroll
          for (int j = 0; j < NT16; ++j)
            br[dw][j] = *reinterpret_cast<const bf16x8*>(wfl + ((int64_t)gn * NTT + j) * 512);
        }
      }
    }
  }
  // ---- epilogue: D layout of 16x16: col = lane & 15, rows 4kg + i -> pixel (h, w) of the mapping above
  if (!fvalid) return;
  if (POOL) {
    // ReLU -> MaxPool((1,2,2)) in registers: the lane's four rows are one window in scan order
#pragma unroll
    for (int j = 0; j < NT16; ++j) {
      const int n = (nh * NT16 + j) * 16 + rl;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int mb = 0; mb < 9; ++mb) {
        bf16_t best;
        int arg;
        relu_pool4(acc[mb][j][0], acc[mb][j][1], acc[mb][j][2], acc[mb][j][3], bv, best, arg);
        const int hp = 2 * (mb / 3) + (kg >> 1), wp = 2 * (mb % 3) + (kg & 1);
        const int64_t o = (((int64_t)f * (P3_H / 2) + hp) * (P3_W / 2) + wp) * N + n;
        Y[o] = best;
        code[o] = (unsigned char)arg;
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NT16; ++j) {
    const int n = (nh * NT16 + j) * 16 + rl;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int mb = 0; mb < 9; ++mb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int hh = 4 * (mb / 3) + 2 * (kg >> 1) + (i >> 1), ww = 4 * (mb % 3) + 2 * (kg & 1) + (i & 1);
        float v = acc[mb][j][i] + bv;
        if (relu) v = fmaxf(v, 0.f);
        Y[(((int64_t)f * P3_H + hh) * P3_W + ww) * N + n] = f2bf(v);
      }
  }
}

// (The weight gradient of the stride-1 layers at the frontend's sizes — LDS transpose reads, persistent workgroups —
// is lr_conv_wgrad.hip; the slab reduction below serves it as well.)

// Reduction of the per-workgroup weight-gradient slabs into dW (fp32, torch layout [n][c][kt][kh][kw]).
// Lanes run over CONSECUTIVE slab elements, so every slab read is a coalesced 256-byte wave load
// (indexing the threads in dW order instead makes neighbouring lanes 8 kB apart: 5x the bytes and
// 7x the time, measured with FETCH_SIZE); the scatter is on the small dW side.  blockDim = (64, P):
// wave y sums slabs y, y+P, ... (8 loads in flight), the P partial sums are combined through LDS in
// a fixed order, so the result is deterministic.  Slab element e maps to dW as
//   ts   = 1:  e = ((kt*khw + tap)*Cout + n)*Cin + c           (tap-stationary / transpose-read kernels)
//   ts   = 0:  e = n*pitch + tap*Cin + c, tap = kt*khw + ...   (im2col-split and conv1 kernels; k >= taps*Cin
//              and c >= Cin_real are padding and are skipped)
__global__ __launch_bounds__(1024) void conv_wgrad_slab_reduce_kernel(const float* __restrict__ slabs, int nslabs,
                                                                      int64_t slab_elems, float* __restrict__ dW,
                                                                      int ts, int Cout, int Cin, int Cin_real,
                                                                      int taps, int khw, int pitch,
                                                                      int accumulate) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x, y = threadIdx.y, P = blockDim.y;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (e < slab_elems) {
    const int mine = (nslabs - y + P - 1) / P;   // slabs y, y+P, ...
    s = strided_sum8(slabs + (int64_t)y * slab_elems + e, mine, (int64_t)P * slab_elems);
  }
  part[y][lane] = s;
  __syncthreads();
  if (y != 0 || e >= slab_elems) return;
  float tot = part[0][lane];
  for (int i = 1; i < P; ++i) tot += part[i][lane];
  int64_t o;
  if (ts) {
    const int c = (int)(e % Cin);
    const int n = (int)((e / Cin) % Cout);
    const int tap = (int)(e / ((int64_t)Cin * Cout));       // kt*khw + t2: dW's own tap order
    o = ((int64_t)n * Cin + c) * taps + tap;
  } else {
    const int k = (int)(e % pitch);
    const int n = (int)(e / pitch);
    const int c = k % Cin, tap = k / Cin;
    if (tap >= taps || c >= Cin_real) return;
    o = ((int64_t)n * Cin_real + c) * taps + tap;
  }
  dW[o] = accumulate ? dW[o] + tot : tot;
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
// forward:  Wp[n][tap][c]  = W[n][c][tap]                      (c >= Cin_real -> 0)
// dgrad:    Wd[c][tap'][n] = W[n][c][flip(tap')]               rows = Cin_real (multiple of 32)
// Both are [N_out][taps][K_ch] (N_out = GEMM output channel, K_ch = contraction channel).  With
// frag != 0 the same elements are written fragment-major for conv_patch_kernel,
//   Wf[cg][tap][kc][nt][lane = kg*32 + n%32][j] = Wp[nt*32 + n%32][tap][cg*32 + kc*16 + kg*8 + j],
// i.e. every MFMA B fragment (32 output channels x 16 k) is 1 KB contiguous, 16 bytes per lane.
__device__ __forceinline__ void conv3d_pack_weights_body(const float* __restrict__ W, bf16_t* __restrict__ out,
                                                         int Cout, int Cin_real, int Cin_pad, int KT, int KH, int KW,
                                                         int dgrad, int frag) {
  const int taps = KT * KH * KW;
  const int Nout = dgrad ? Cin_real : Cout, Kch = dgrad ? Cout : Cin_pad;
  const int64_t total = (int64_t)Nout * taps * Kch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int kch = (int)(i % Kch);
    const int tp = (int)((i / Kch) % taps);
    const int nout = (int)(i / ((int64_t)Kch * taps));
    float v = 0.f;
    if (!dgrad) {
      if (kch < Cin_real) v = W[((int64_t)nout * Cin_real + kch) * taps + tp];
    } else {
      const int tap = taps - 1 - tp;  // flip kt, kh and kw together
      v = W[((int64_t)kch * Cin_real + nout) * taps + tap];
    }
    int64_t dst = i;
    if (frag == 1) {
      const int cg = kch >> 5, kc = (kch >> 4) & 1, kg = (kch >> 3) & 1, j = kch & 7;
      const int nt = nout >> 5, nl = nout & 31, NT = Nout >> 5;
      dst = ((((((int64_t)cg * taps + tp) * 2 + kc) * NT + nt) * 64 + kg * 32 + nl) << 3) + j;
    } else if (frag == 2) {
      // 16-column fragments of v_mfma_f32_16x16x32_bf16: Wf[cg][tap][n16][lane = kg*16 + n%16][j],
      // k = cg*32 + kg*8 + j  (conv_patch16_kernel)
      const int cg = kch >> 5, kg = (kch >> 3) & 3, j = kch & 7;
      const int n16 = nout >> 4, nl = nout & 15, NT16 = Nout >> 4;
      dst = (((((int64_t)cg * taps + tp) * NT16 + n16) * 64 + kg * 16 + nl) << 3) + j;
    }
    out[dst] = f2bf(v);
  }
}
__global__ void conv3d_pack_weights_kernel(const float* __restrict__ W, bf16_t* __restrict__ out, int Cout,
                                           int Cin_real, int Cin_pad, int KT, int KH, int KW,
                                           int dgrad, int frag) {
  conv3d_pack_weights_body(W, out, Cout, Cin_real, Cin_pad, KT, KH, KW, dgrad, frag);
}
// several operands in one launch (blockIdx.y = item): a training step packs the forward operand of every layer
// and the data-gradient operand of the upper layers — five ~4 us launches otherwise
constexpr int kPackMax = 8;
struct PackItems {
  const float* W[kPackMax];
  bf16_t* out[kPackMax];
  int cout[kPackMax], cin_real[kPackMax], cin_pad[kPackMax], kt[kPackMax], kh[kPackMax], kw[kPackMax];
  int flip[kPackMax], frag[kPackMax];
};
__global__ void conv3d_pack_weights_multi_kernel(PackItems p) {
  const int z = blockIdx.y;
  conv3d_pack_weights_body(p.W[z], p.out[z], p.cout[z], p.cin_real[z], p.cin_pad[z], p.kt[z], p.kh[z], p.kw[z],
                           p.flip[z], p.frag[z]);
}

// ---------------------------------------------------------------------------------------------
// elementwise stages
// ---------------------------------------------------------------------------------------------
// clips [B*T][3][H][W] (u8 scaled by 1/255, or f32 as is) -> [B*T][H][W][4] bf16 (4th channel 0)
__global__ void clip_to_ndhwc_kernel(const void* __restrict__ clips, int is_u8, bf16_t* __restrict__ out,
                                     int64_t frames, int H, int W) {
  const int64_t total = frames * H * W;
  const int64_t hw = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = i / hw, p = i - f * hw;
    float c[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const int64_t src = (f * 3 + ch) * hw + p;
      c[ch] = is_u8 ? (float)((const unsigned char*)clips)[src] * (1.f / 255.f) : ((const float*)clips)[src];
    }
    uint2 v;
    v.x = (unsigned)f2bf(c[0]) | ((unsigned)f2bf(c[1]) << 16);
    v.y = (unsigned)f2bf(c[2]);
    *reinterpret_cast<uint2*>(out + i * 4) = v;
  }
}

// MaxPool3d((1,2,2)) on channels-last bf16: [F][H][W][C] -> [F][H/2][W/2][C]
__global__ void maxpool_hw2_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int64_t frames,
                                   int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = frames * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t q = i / C;
    const int wo = (int)(q % Wo);
    q /= Wo;
    const int ho = (int)(q % Ho);
    const int64_t f = q / Ho;
    const bf16_t* p = in + (((f * H + 2 * ho) * W) + 2 * wo) * C + c;
    const float a = bf2f(p[0]), b = bf2f(p[C]), d = bf2f(p[(int64_t)W * C]), e = bf2f(p[(int64_t)W * C + C]);
    out[i] = f2bf(fmaxf(fmaxf(a, b), fmaxf(d, e)));
  }
}

// Backward of ReLU -> MaxPool((1,2,2)):  dZ[pos] = dP[window] if pos is the FIRST maximum of its
// window (row-major scan, torch's max_pool backward) and the activation there is > 0, else 0.
// One thread = 8 channels (16 bytes) of one pooled element.  The bias gradient of the layer,
// dbias[n] = sum_pos dZ[pos][n], is the sum of the routed gradients, so it is accumulated here per
// thread (the launch keeps gridDim * 256 a multiple of C/8, so a thread's channel group is fixed),
// reduced per workgroup in fixed order and written as one partial row per workgroup.
constexpr int kUnpoolBlocks = 768;   // 3 per CU; a multiple of 3, so 768 * 256 % (C/8) == 0 for C in {32,64,96}
__global__ __launch_bounds__(256) void unpool_relu_mask_kernel(const bf16_t* __restrict__ act,
                                                               const bf16_t* __restrict__ dP,
                                                               bf16_t* __restrict__ dZ, int64_t frames, int H, int W,
                                                               int C, float* __restrict__ partial) {
  __shared__ float red[256][9];
  const int Ho = H / 2, Wo = W / 2, G = C / 8;
  const int64_t total = frames * Ho * Wo * G;
  float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % G);
    int64_t q = i / G;
    const int wo = (int)(q % Wo);
    q /= Wo;
    const int ho = (int)(q % Ho);
    const int64_t f = q / Ho;
    const int64_t base = (((f * H + 2 * ho) * W) + 2 * wo) * C + cg * 8;
    const int64_t offs[4] = {0, C, (int64_t)W * C, (int64_t)W * C + C};
    uint4 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const uint4*>(act + base + offs[j]);
    const uint4 gp = *reinterpret_cast<const uint4*>(dP + (q * Wo + wo) * C + cg * 8);
    const unsigned gw[4] = {gp.x, gp.y, gp.z, gp.w};
    unsigned o[4][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int wd = e >> 1, sh = (e & 1) * 16;
      float best = -__builtin_inff();
      int arg = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned wv = wd == 0 ? a[j].x : (wd == 1 ? a[j].y : (wd == 2 ? a[j].z : a[j].w));
        const float v = bf2f((bf16_t)((wv >> sh) & 0xffffu));
        if (v > best) { best = v; arg = j; }
      }
      const unsigned g = best > 0.f ? ((gw[wd] >> sh) & 0xffffu) : 0u;
      sum[e] += bf2f((bf16_t)g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j == arg) o[j][wd] |= g << sh;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<uint4*>(dZ + base + offs[j]) = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
  }
  if (!partial) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = sum[e];
  __syncthreads();
  if ((int)threadIdx.x < C) {
    const int c = threadIdx.x, gq = c >> 3, e = c & 7;
    const int b0 = (int)(((int64_t)blockIdx.x * blockDim.x) % G);
    float s = 0.f;
    for (int tt = ((gq - b0) % G + G) % G; tt < 256; tt += G) s += red[tt][e];   // threads of channel group gq
    partial[(int64_t)blockIdx.x * C + c] = s;
  }
}

// The same backward for layers whose forward fused the pooling (lr_conv3d_forward_pooled): the
// window's gradient goes to position code (0..3, row-major) if the pooled activation is > 0.
__global__ __launch_bounds__(256) void unpool_code_kernel(const bf16_t* __restrict__ pooled,
                                                          const unsigned char* __restrict__ code,
                                                          const bf16_t* __restrict__ dP, bf16_t* __restrict__ dZ,
                                                          int64_t frames, int H, int W, int C,
                                                          float* __restrict__ partial) {
  __shared__ float red[256][9];
  const int Ho = H / 2, Wo = W / 2, G = C / 8;
  const int64_t total = frames * Ho * Wo * G;
  float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % G);
    int64_t q = i / G;
    const int wo = (int)(q % Wo);
    q /= Wo;
    const int ho = (int)(q % Ho);
    const int64_t f = q / Ho;
    const int64_t base = (((f * H + 2 * ho) * W) + 2 * wo) * C + cg * 8;
    const int64_t offs[4] = {0, C, (int64_t)W * C, (int64_t)W * C + C};
    const int64_t pi = (q * Wo + wo) * C + cg * 8;
    const uint4 pv = *reinterpret_cast<const uint4*>(pooled + pi);
    const uint4 gp = *reinterpret_cast<const uint4*>(dP + pi);
    const uint2 cv = *reinterpret_cast<const uint2*>(code + pi);
    const unsigned pw_[4] = {pv.x, pv.y, pv.z, pv.w}, gw[4] = {gp.x, gp.y, gp.z, gp.w};
    unsigned o[4][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int wd = e >> 1, sh = (e & 1) * 16;
      const float act = bf2f((bf16_t)((pw_[wd] >> sh) & 0xffffu));
      const int arg = (int)(((e < 4 ? cv.x : cv.y) >> (8 * (e & 3))) & 3u);
      const unsigned g = act > 0.f ? ((gw[wd] >> sh) & 0xffffu) : 0u;
      sum[e] += bf2f((bf16_t)g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j == arg) o[j][wd] |= g << sh;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<uint4*>(dZ + base + offs[j]) = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
  }
  if (!partial) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = sum[e];
  __syncthreads();
  if ((int)threadIdx.x < C) {
    const int c = threadIdx.x, gq = c >> 3, e = c & 7;
    const int b0 = (int)(((int64_t)blockIdx.x * blockDim.x) % G);
    float s = 0.f;
    for (int tt = ((gq - b0) % G + G) % G; tt < 256; tt += G) s += red[tt][e];
    partial[(int64_t)blockIdx.x * C + c] = s;
  }
}

__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = bf2f(in[i]);
}
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = f2bf(in[i]);
}

// column sums of a bf16 [M][C] matrix (C % 8 == 0), two-stage (deterministic).  Stage 1: every
// thread owns 8 channels (one 16-byte load per row) and strides over the rows of its workgroup's
// slice; the workgroup combines its row lanes through LDS and writes partial[blockIdx.x][C].
// code != nullptr: x is a POOLED gradient and code the windows' codes (relu_pool4): elements whose code is 4 (blocked
// by ReLU) are left out — the column sums of the un-pooled dZ, i.e. the bias gradient, without dZ.
__global__ void colsum_bf16_partial_kernel(const bf16_t* __restrict__ x, int64_t M, int C,
                                           float* __restrict__ partial, int splits,
                                           const unsigned char* __restrict__ code) {
  __shared__ float red[256][9];
  const int tpr = C >> 3;                    // threads per row
  const int rl = threadIdx.x / tpr, cg = threadIdx.x - rl * tpr;
  const int rows_per_iter = 256 / tpr;
  const int64_t per = (M + splits - 1) / splits;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(M, r0 + per);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < rows_per_iter) {
    for (int64_t r = r0 + rl; r < r1; r += rows_per_iter) {
      uint4 v = *reinterpret_cast<const uint4*>(x + r * C + cg * 8);
      if (code) {
        const uint2 cc = *reinterpret_cast<const uint2*>(code + r * C + cg * 8);
        // bit 2 of a code byte = blocked: spread it into a byte mask, double the bytes into 16-bit lanes
        const unsigned k0 = (cc.x >> 2) & 0x01010101u, k1 = (cc.y >> 2) & 0x01010101u;
        const unsigned b0 = (k0 << 8) - k0, b1 = (k1 << 8) - k1;
        v.x &= ~__builtin_amdgcn_perm(b0, b0, 0x01010000u);
        v.y &= ~__builtin_amdgcn_perm(b0, b0, 0x03030202u);
        v.z &= ~__builtin_amdgcn_perm(b1, b1, 0x01010000u);
        v.w &= ~__builtin_amdgcn_perm(b1, b1, 0x03030202u);
      }
      acc[0] += bf2f((bf16_t)(v.x & 0xffffu)); acc[1] += bf2f((bf16_t)(v.x >> 16));
      acc[2] += bf2f((bf16_t)(v.y & 0xffffu)); acc[3] += bf2f((bf16_t)(v.y >> 16));
      acc[4] += bf2f((bf16_t)(v.z & 0xffffu)); acc[5] += bf2f((bf16_t)(v.z >> 16));
      acc[6] += bf2f((bf16_t)(v.w & 0xffffu)); acc[7] += bf2f((bf16_t)(v.w >> 16));
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[i];
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x, g = c >> 3, i = c & 7;
    float s = 0.f;
    for (int r = 0; r < rows_per_iter; ++r) s += red[r * tpr + g][i];
    partial[(int64_t)blockIdx.x * C + c] = s;
  }
}
__global__ void colsum_final_acc_kernel(const float* __restrict__ partial, int splits, float* __restrict__ out,
                                        int C, int accumulate) {
  // blockDim / C row lanes per channel, each a fixed subsequence of the partial rows; the lanes'
  // sums are combined in lane order (deterministic)
  __shared__ float red[1024];
  const int RL = blockDim.x / C, c = threadIdx.x % C, rl = threadIdx.x / C;
  float s = 0.f;
  if (rl < RL) s = strided_sum8(partial + (int64_t)rl * C + c, (splits - rl + RL - 1) / RL, (int64_t)RL * C);
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int r = 0; r < RL; ++r) t += red[r * C + c];
    out[c] = accumulate ? out[c] + t : t;
  }
}

inline int grid1d(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

bool fill_geom(ConvGeom* g, int B, int T, int Hin, int Win, int Cin, int Cout, int KT, int KH, int KW,
               int stride, int pt, int ph, int pw) {
  if (B <= 0 || T <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cout <= 0 || stride <= 0) return false;
  if (Cin % 4 != 0 || Cout % 32 != 0 || Cout > 96 || KT <= 0 || KH <= 0 || KW <= 0) return false;
  g->B = B; g->T = T; g->Hin = Hin; g->Win = Win; g->Cin = Cin; g->Cout = Cout;
  g->KT = KT; g->KH = KH; g->KW = KW; g->stride = stride; g->pt = pt; g->ph = ph; g->pw = pw;
  g->Ho = (Hin + 2 * ph - KH) / stride + 1;
  g->Wo = (Win + 2 * pw - KW) / stride + 1;
  g->Ktot = KT * KH * KW * Cin;
  g->M = (int64_t)B * T * g->Ho * g->Wo;
  if (KT != 2 * pt + 1) return false;  // "same" in time, temporal stride 1
  return g->Ho > 0 && g->Wo > 0;
}

// pixel splits of the weight gradient: enough workgroups to fill the chip for each k-tile count
static int wgrad_ntw(int Cout) { return Cout == 32 ? 5 : (Cout == 64 ? 4 : 2); }
static int wgrad_splits(int Cout, int Ktot) {
  const int coltiles = (Ktot + wgrad_ntw(Cout) * 32 - 1) / (wgrad_ntw(Cout) * 32);
  int s = 1024 / coltiles;
  if (s < 16) s = 16;
  if (s > 512) s = 512;
  return s;
}
constexpr int kColsumSplits = 256;
constexpr int kTsWgsPerKt = LR_CONV_TR2_SLOTS;   // 3 temporal offsets x 85 = 255 resident workgroups

}  // namespace

extern "C" int lr_clip_to_ndhwc_bf16(const void* clips, int is_u8, void* out, int64_t frames, int H,
                                     int W, lr_stream_t stream) {
  LR_CHECK_ARG(clips && out && frames > 0 && H > 0 && W > 0);
  LR_LAUNCH(clip_to_ndhwc_kernel, dim3(grid1d(frames * H * W)), dim3(256), 0, stream, clips, is_u8,
            (bf16_t*)out, frames, H, W);
  return lr_launch_status();
}

extern "C" int lr_conv3d_pack_weights(const float* W, void* out, int Cout, int Cin_real, int Cin_pad,
                                      int KT, int KH, int KW, int dgrad, lr_stream_t stream) {
  LR_CHECK_ARG(W && out && Cout > 0 && Cin_real > 0 && Cin_pad >= Cin_real);
  const int flip = dgrad & 1, frag = (dgrad & 2) ? 1 : ((dgrad & 4) ? 2 : 0);
  // fragment-major needs whole 32-channel groups on both axes
  if (frag && ((flip ? Cin_real : Cout) % 32 != 0 || (flip ? Cout : Cin_pad) % 32 != 0)) return LR_ERR_UNSUPPORTED;
  const int64_t total = flip ? (int64_t)Cin_real * KT * KH * KW * Cout
                             : (int64_t)Cout * KT * KH * KW * Cin_pad;
  LR_LAUNCH(conv3d_pack_weights_kernel, dim3(grid1d(total)), dim3(256), 0, stream, W, (bf16_t*)out,
            Cout, Cin_real, Cin_pad, KT, KH, KW, flip, frag);
  return lr_launch_status();
}

extern "C" int lr_conv3d_pack_weights_multi(int n, const float* const* W, void* const* out, const int* Cout,
                                            const int* Cin_real, const int* Cin_pad, const int* KT, const int* KH,
                                            const int* KW, const int* dgrad, lr_stream_t stream) {
  LR_CHECK_ARG(n >= 1 && n <= kPackMax && W && out && Cout && Cin_real && Cin_pad && KT && KH && KW && dgrad);
  PackItems p;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    LR_CHECK_ARG(W[i] && out[i] && Cout[i] > 0 && Cin_real[i] > 0 && Cin_pad[i] >= Cin_real[i]);
    const int flip = dgrad[i] & 1, frag = (dgrad[i] & 2) ? 1 : ((dgrad[i] & 4) ? 2 : 0);
    if (frag && ((flip ? Cin_real[i] : Cout[i]) % 32 != 0 || (flip ? Cout[i] : Cin_pad[i]) % 32 != 0))
      return LR_ERR_UNSUPPORTED;
    p.W[i] = W[i]; p.out[i] = (bf16_t*)out[i];
    p.cout[i] = Cout[i]; p.cin_real[i] = Cin_real[i]; p.cin_pad[i] = Cin_pad[i];
    p.kt[i] = KT[i]; p.kh[i] = KH[i]; p.kw[i] = KW[i]; p.flip[i] = flip; p.frag[i] = frag;
    const int64_t total = flip ? (int64_t)Cin_real[i] * KT[i] * KH[i] * KW[i] * Cout[i]
                               : (int64_t)Cout[i] * KT[i] * KH[i] * KW[i] * Cin_pad[i];
    if (total > most) most = total;
  }
  for (int i = n; i < kPackMax; ++i) {
    p.W[i] = nullptr; p.out[i] = nullptr;
    p.cout[i] = p.cin_real[i] = p.cin_pad[i] = p.kt[i] = p.kh[i] = p.kw[i] = p.flip[i] = p.frag[i] = 0;
  }
  LR_LAUNCH(conv3d_pack_weights_multi_kernel, dim3(grid1d(most), n), dim3(256), 0, stream, p);
  return lr_launch_status();
}

extern "C" int lr_conv3d_patch_supported(int Hin, int Win, int Cin, int Cout, int KT, int KH, int KW,
                                         int stride, int pt, int ph, int pw) {
  const bool shape2 = KT == 3 && KH == 5 && KW == 5 && stride == 1 && pt == 1 && ph == 2 && pw == 2 &&
                      Win == 24 && Hin > 0 && Hin % 8 == 0;   // lr_conv_patch.hip: P2_W, P2_TH
  if (shape2 && ((Cin == 32 && Cout == 64) || (Cin == 64 && Cout == 32))) return 2;   // 32-column fragments
  const bool shape3 = KT == 3 && KH == 3 && KW == 3 && stride == 1 && pt == 1 && ph == 1 && pw == 1 &&
                      Win == P3_W && Hin == P3_H;
  if (shape3 && ((Cin == 64 && Cout == 96) || (Cin == 96 && Cout == 64))) return 4;   // 16-column fragments
  return 0;
}

extern "C" int lr_conv3d_pool_fusion_supported(int Hin, int Win, int Cin, int Cout, int KT, int KH, int KW,
                                               int stride, int pt, int ph, int pw) {
  const bool first = Cin == 4 && Cout == 32 && KT == 3 && KH == 5 && KW == 5 && stride == 2 && pt == 1 && ph == 2 &&
                     pw == 2 && Hin % 4 == 0 && Win % 4 == 0;
  const bool second = Cin == 32 && Cout == 64 &&
                      lr_conv3d_patch_supported(Hin, Win, Cin, Cout, KT, KH, KW, stride, pt, ph, pw);
  const bool third = Cin == 64 && Cout == 96 &&
                     lr_conv3d_patch_supported(Hin, Win, Cin, Cout, KT, KH, KW, stride, pt, ph, pw);
  return first || second || third ? 1 : 0;
}

// code == nullptr: Y = full-resolution activation; else Y = ReLU -> MaxPool((1,2,2)) of it and code =
// position of each window's first maximum (layers with lr_conv3d_pool_fusion_supported only)
// ucode != nullptr (data gradients through the patch-resident kernels only): X is the POOLED gradient
// [B][T][Hin/2][Win/2][Cin] and ucode the windows' codes; the kernel un-pools on the way into LDS.
static int conv_forward_impl(const void* X, const void* Wp, const float* bias, void* Y, unsigned char* code, int B,
                             int T, int Hin, int Win, int Cin, int Cout, int KT, int KH, int KW, int stride, int pt,
                             int ph, int pw, int flags, lr_stream_t stream, const unsigned char* ucode = nullptr) {
  LR_CHECK_ARG(X && Wp && Y);
  const int relu = flags & 1;
  if (ucode && (code || relu || bias || !(flags & 6))) return LR_ERR_UNSUPPORTED;
  const bool u8 = (flags & 8) != 0;   // X is the raw uint8 planar clip: first-layer patch kernel only
  if (u8 && !(Cin == 4 && Cout == 32 && KT == 3 && KH == 5 && KW == 5 && stride == 2 && pt == 1 && ph == 2 && pw == 2))
    return LR_ERR_UNSUPPORTED;
  ConvGeom g;
  if (!fill_geom(&g, B, T, Hin, Win, Cin, Cout, KT, KH, KW, stride, pt, ph, pw)) return LR_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((g.M + IG_BM - 1) / IG_BM));
  const bf16_t* x = (const bf16_t*)X;
  const bf16_t* w = (const bf16_t*)Wp;
  bf16_t* y = (bf16_t*)Y;
  if (code && (!relu || !lr_conv3d_pool_fusion_supported(Hin, Win, Cin, Cout, KT, KH, KW, stride, pt, ph, pw)))
    return LR_ERR_UNSUPPORTED;
  if (flags & 4) {
    // 16-column fragment-major weights: the 12x12 patch-resident kernel
    if (lr_conv3d_patch_supported(Hin, Win, Cin, Cout, KT, KH, KW, stride, pt, ph, pw) != 4) return LR_ERR_UNSUPPORTED;
    const int F = B * T;
    const dim3 pgrid((unsigned)((F + P3_TT - 1) / P3_TT));
    hipEvent_t e0, e1;
    const bool fwd = Cin == 64;
    const bool sample = lr_prof_next(fwd ? LR_PROF_CONV3_FWD : LR_PROF_CONV3_DGRAD, &e0, &e1);
    static bool attr16[4] = {false, false, false, false};
    lr_clear_error();
#define LR_PATCH16(IDX, ...)                                                                                    \
  do {                                                                                                         \
    if (!attr16[IDX]) {                                                                                        \
      if (hipFuncSetAttribute((const void*)conv_patch16_kernel<__VA_ARGS__>,                                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, P3_LDS) != hipSuccess)               \
        return LR_ERR_LAUNCH;                                                                                  \
      attr16[IDX] = true;                                                                                      \
    }                                                                                                          \
    if (sample) hipExtLaunchKernelGGL((conv_patch16_kernel<__VA_ARGS__>), pgrid, dim3(256), P3_LDS,             \
                                      (hipStream_t)stream, e0, e1, 0, x, w, bias, y, code, F, T, relu);        \
    else hipLaunchKernelGGL((conv_patch16_kernel<__VA_ARGS__>), pgrid, dim3(256), P3_LDS, (hipStream_t)stream,  \
                            x, w, bias, y, code, F, T, relu);                                                  \
  } while (0)
    if (ucode && fwd) return LR_ERR_UNSUPPORTED;
    if (ucode) code = const_cast<unsigned char*>(ucode);
    if (fwd && code) LR_PATCH16(2, 2, 6 / P3_NSPL, true, false);
    else if (fwd) LR_PATCH16(0, 2, 6 / P3_NSPL, false, false);
    else if (ucode) LR_PATCH16(3, 3, 4 / P3_NSPL, false, true);
    else LR_PATCH16(1, 3, 4 / P3_NSPL, false, false);
#undef LR_PATCH16
    return lr_launch_status();
  }
  if (flags & 2) {
    // fragment-major weights: the patch-resident kernel (no other kernel reads that packing)
    if (lr_conv3d_patch_supported(Hin, Win, Cin, Cout, KT, KH, KW, stride, pt, ph, pw) != 2) return LR_ERR_UNSUPPORTED;
    hipEvent_t e0, e1;
    const bool fwd = Cin == 32;
    const bool sample = lr_prof_next(fwd ? LR_PROF_CONV2_FWD : LR_PROF_CONV2_DGRAD, &e0, &e1);
    if (ucode && fwd) return LR_ERR_UNSUPPORTED;
    return lr_conv_patch24(fwd, ucode != nullptr, x, w, bias, y, ucode ? const_cast<unsigned char*>(ucode) : code, B * T, T,
                           Hin, relu, sample, e0, e1, (hipStream_t)stream);
  }
  if (code && Cin != 4) return LR_ERR_UNSUPPORTED;   // the second layer's fused pooling lives in the patch kernel
  if (Cin == 4 && Cout == 32 && KT == 3 && KH == 5 && KW == 5 && stride == 2 && pt == 1 && ph == 2 && pw == 2) {
    // first layer: patch-resident kernel (one 16x16 output tile of one frame per workgroup), lr_conv1.hip
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool sample = relu && lr_prof_next(LR_PROF_CONV1_FWD, &e0, &e1);
    return lr_conv1_forward(code != nullptr, u8, x, w, bias, y, code, B * T, T, Hin, Win, g.Ho, g.Wo, relu, sample, e0,
                            e1, (hipStream_t)stream);
  }
  // instrumentation slot: forward layers by input channels, data gradients by (Cin, relu == 0)
  int slot = -1;
  if (relu) slot = Cin == 4 ? LR_PROF_CONV1_FWD : (Cin == 32 ? LR_PROF_CONV2_FWD : LR_PROF_CONV3_FWD);
  else if (!bias) slot = Cin == 64 ? LR_PROF_CONV2_DGRAD : (Cin == 96 ? LR_PROF_CONV3_DGRAD : -1);
  hipEvent_t e0, e1;
  const bool sample = lr_prof_next(slot, &e0, &e1);
#define LR_IGEMM(CI, NTT)                                                                              \
  do {                                                                                                 \
    lr_clear_error();                                                                                  \
    if (sample) hipExtLaunchKernelGGL((conv3d_igemm_kernel<CI, NTT>), grid, dim3(256), 0,               \
                                      (hipStream_t)stream, e0, e1, 0, g, x, w, bias, y, relu);         \
    else hipLaunchKernelGGL((conv3d_igemm_kernel<CI, NTT>), grid, dim3(256), 0, (hipStream_t)stream, g, \
                            x, w, bias, y, relu);                                                      \
  } while (0)
  const int nt = Cout / 32;
  if (Cin == 4 && nt == 1) LR_IGEMM(4, 1);
  else if (Cin == 32 && nt == 2) LR_IGEMM(32, 2);
  else if (Cin == 64 && nt == 3) LR_IGEMM(64, 3);
  else if (Cin == 64 && nt == 1) LR_IGEMM(64, 1);     // data gradient of layer 2
  else if (Cin == 96 && nt == 2) LR_IGEMM(96, 2);     // data gradient of layer 3
  else if (Cin == 32 && nt == 1) LR_IGEMM(32, 1);
  else if (Cin == 32 && nt == 3) LR_IGEMM(32, 3);
  else if (Cin == 64 && nt == 2) LR_IGEMM(64, 2);
  else if (Cin == 96 && nt == 1) LR_IGEMM(96, 1);
  else if (Cin == 96 && nt == 3) LR_IGEMM(96, 3);
  else return LR_ERR_UNSUPPORTED;
#undef LR_IGEMM
  return lr_launch_status();
}

extern "C" int lr_conv3d_forward(const void* X, const void* Wp, const float* bias, void* Y, int B,
                                 int T, int Hin, int Win, int Cin, int Cout, int KT, int KH, int KW,
                                 int stride, int pt, int ph, int pw, int flags, lr_stream_t stream) {
  return conv_forward_impl(X, Wp, bias, Y, nullptr, B, T, Hin, Win, Cin, Cout, KT, KH, KW, stride, pt, ph, pw, flags,
                           stream);
}

extern "C" int lr_conv3d_dgrad_pooled_supported(int Ho, int Wo, int Cout, int Cin, int KT, int KH, int KW, int pt, int ph,
                                                int pw) {
  // the data gradient of a stride-1 "same" convolution is the forward kernel on dZ (Cout channels) with the flipped,
  // channel-transposed weights: supported where that product has a patch-resident kernel
  if (Ho <= 0 || Wo <= 0 || (Ho & 1) || (Wo & 1)) return 0;
  return lr_conv3d_patch_supported(Ho, Wo, Cout, Cin, KT, KH, KW, 1, pt, ph, pw);
}

extern "C" int lr_conv3d_dgrad_pooled(const void* dP, const void* code, const void* Wd, void* dX, int B, int T, int Ho,
                                      int Wo, int Cout, int Cin, int KT, int KH, int KW, int pt, int ph, int pw,
                                      lr_stream_t stream) {
  LR_CHECK_ARG(dP && code && Wd && dX);
  const int frag = lr_conv3d_dgrad_pooled_supported(Ho, Wo, Cout, Cin, KT, KH, KW, pt, ph, pw);
  if (!frag) return LR_ERR_UNSUPPORTED;
  return conv_forward_impl(dP, Wd, nullptr, dX, nullptr, B, T, Ho, Wo, Cout, Cin, KT, KH, KW, 1, pt, ph, pw, frag, stream,
                           (const unsigned char*)code);
}

extern "C" int lr_conv3d_forward_pooled(const void* X, const void* Wp, const float* bias, void* P, void* code,
                                        int B, int T, int Hin, int Win, int Cin, int Cout, int KT, int KH, int KW,
                                        int stride, int pt, int ph, int pw, int flags, lr_stream_t stream) {
  LR_CHECK_ARG(code);
  return conv_forward_impl(X, Wp, bias, P, (unsigned char*)code, B, T, Hin, Win, Cin, Cout, KT, KH, KW, stride, pt,
                           ph, pw, flags | 1, stream);
}

extern "C" size_t lr_conv3d_wgrad_workspace_bytes(int Cout, int Cin_pad, int KT, int KH, int KW) {
  if (Cout <= 0 || Cin_pad <= 0) return 0;
  const int Ktot = KT * KH * KW * Cin_pad;
  size_t slab = (size_t)wgrad_splits(Cout, Ktot) * Cout * Ktot;
  const size_t ts = (size_t)KT * kTsWgsPerKt * KH * KW * Cout * Cin_pad;   // tap-stationary path
  if (ts > slab) slab = ts;
  if (slab < (size_t)LR_CONV1_WGRAD_WGS * 32 * 320) slab = (size_t)LR_CONV1_WGRAD_WGS * 32 * 320;   // first-layer patch kernel
  const size_t colparts = kColsumSplits > LR_CONV1_WGRAD_WGS ? kColsumSplits : LR_CONV1_WGRAD_WGS;
  return (slab + colparts * Cout) * sizeof(float);
}

// The stride-1 layers' weight gradient with LDS transpose reads (lr_conv_wgrad.hip): which layer (2 / 3) the geometry
// is, 0 if neither; and the run: kernel, slab reduction, bias gradient.  code != nullptr: dZ is the POOLED gradient
// [F][Ho/2][Wo/2][Cout] and code the windows' codes — the kernel un-pools on the way into LDS and the bias gradient is
// the column sum of dP over the windows ReLU did not block.
static int wgrad_tr2_layer(int F, int Hin, int Win, int Cin_pad, int Cin_real, int Cout, int KT, int KH, int KW, int stride,
                           int pt, int ph, int pw) {
  const bool l2 = Cin_pad == 32 && Cout == 64 && KH == 5 && KW == 5 && Win == 24 && Hin % 4 == 0;
  const bool l3 = Cin_pad == 64 && Cout == 96 && KH == 3 && KW == 3 && Win == 12 && Hin % 6 == 0;
  if (stride == 1 && KT == 3 && pt == 1 && 2 * ph + 1 == KH && 2 * pw + 1 == KW && Cin_real == Cin_pad && (l2 || l3) &&
      lr_conv_wgrad_tr2_supported(l2 ? 2 : 3, F, Hin))
    return l2 ? 2 : 3;
  return 0;
}
static int wgrad_tr2_run(const void* X, const void* dZ, const void* code, float* dW, float* dbias, void* workspace,
                         int accumulate, int F, int T, int Hin, int Win, int Cin_pad, int Cout, int KT, int KH, int KW,
                         bool sample, hipEvent_t e0, hipEvent_t e1, hipStream_t stream) {
  const int layer = Cin_pad == 32 ? 2 : 3;
  float* slabs = (float*)workspace;
  const int nslots = LR_CONV_TR2_SLOTS;
  int st = lr_conv_wgrad_tr2(layer, X, dZ, code, slabs, F, T, Hin, sample, e0, e1, stream);
  if (st != LR_OK) return st;
  {
    const int64_t se = (int64_t)Cout * Cin_pad * KT * KH * KW;
    LR_LAUNCH(conv_wgrad_slab_reduce_kernel, dim3((unsigned)((se + 63) / 64)), dim3(64, 4), 0, stream,
              (const float*)slabs, nslots, se, dW, 1, Cout, Cin_pad, Cin_pad, KT * KH * KW, KH * KW, 0,
              accumulate);
  }
  st = lr_launch_status();
  if (st != LR_OK || !dbias) return st;
  float* cpart2 = slabs + (size_t)3 * kTsWgsPerKt * KH * KW * Cout * Cin_pad;
  // rows of dZ (stride-1 "same" layers: output extent = input extent), or of the pooled gradient
  const int64_t rows = code ? (int64_t)F * (Hin / 2) * (Win / 2) : (int64_t)F * Hin * Win;
  LR_LAUNCH(colsum_bf16_partial_kernel, dim3(kColsumSplits), dim3(256), 0, stream, (const bf16_t*)dZ, rows, Cout, cpart2,
            kColsumSplits, (const unsigned char*)code);
  LR_LAUNCH(colsum_final_acc_kernel, dim3(1), dim3(1024), 0, stream, (const float*)cpart2, kColsumSplits, dbias,
            Cout, accumulate);
  return lr_launch_status();
}

extern "C" int lr_conv3d_wgrad_pooled_supported(int Hin, int Win, int Cin_pad, int Cin_real, int Cout, int KT,
                                               int KH, int KW, int stride, int pt, int ph, int pw) {
  const int Ho = (Hin + 2 * ph - KH) / (stride > 0 ? stride : 1) + 1, Wo = (Win + 2 * pw - KW) / (stride > 0 ? stride : 1) + 1;
  if (Cin_pad == 4 && Cin_real == 3 && Cout == 32 && KT == 3 && KH == 5 && KW == 5 && stride == 2 && pt == 1 &&
      ph == 2 && pw == 2 && Ho % 2 == 0 && Wo % 2 == 0)
    return 1;
  // 2: the stride-1 layers' transpose-read kernel un-pools on the fly (any frame count that fits its tile table: the
  // caller asks again through lr_conv3d_wgrad_pooled, which answers LR_ERR_UNSUPPORTED when B * T does not)
  return wgrad_tr2_layer(1, Hin, Win, Cin_pad, Cin_real, Cout, KT, KH, KW, stride, pt, ph, pw) && Hin % 2 == 0 ? 2 : 0;
}

// ... with the frame count: 0 where lr_conv3d_wgrad_pooled would answer LR_ERR_UNSUPPORTED for B * T = frames (the
// transpose-read kernel's tile table has to fit its LDS), so that a caller can choose its path — and allocate the
// un-pooled gradient of the other one — before it enqueues anything
extern "C" int lr_conv3d_wgrad_pooled_supported_frames(int frames, int Hin, int Win, int Cin_pad, int Cin_real, int Cout,
                                                      int KT, int KH, int KW, int stride, int pt, int ph, int pw) {
  const int kind = lr_conv3d_wgrad_pooled_supported(Hin, Win, Cin_pad, Cin_real, Cout, KT, KH, KW, stride, pt, ph, pw);
  if (kind == 2 && !wgrad_tr2_layer(frames, Hin, Win, Cin_pad, Cin_real, Cout, KT, KH, KW, stride, pt, ph, pw)) return 0;
  return frames > 0 ? kind : 0;
}

extern "C" int lr_conv3d_wgrad_pooled(const void* X, const void* pooled, const void* code, const void* dP, float* dW,
                                      float* dbias, void* workspace, size_t workspace_bytes, int accumulate, int B,
                                      int T, int Hin, int Win, int Cin_pad, int Cin_real, int Cout, int KT, int KH,
                                      int KW, int stride, int pt, int ph, int pw, int flags, lr_stream_t stream) {
  LR_CHECK_ARG(X && code && dP && dW && workspace);
  const bool u8 = (flags & 1) != 0;   // X is the raw uint8 planar clip
  const int kind = lr_conv3d_wgrad_pooled_supported(Hin, Win, Cin_pad, Cin_real, Cout, KT, KH, KW, stride, pt, ph, pw);
  if (!kind) return LR_ERR_UNSUPPORTED;
  ConvGeom g;
  if (!fill_geom(&g, B, T, Hin, Win, Cin_pad, Cout, KT, KH, KW, stride, pt, ph, pw)) return LR_ERR_UNSUPPORTED;
  const size_t need = lr_conv3d_wgrad_workspace_bytes(Cout, Cin_pad, KT, KH, KW);
  if (workspace_bytes < need) return LR_ERR_WORKSPACE;
  if (kind == 2) {   // layers 2 / 3 (the codes carry the ReLU mask: `pooled` is not read)
    if (u8 || !wgrad_tr2_layer(B * T, Hin, Win, Cin_pad, Cin_real, Cout, KT, KH, KW, stride, pt, ph, pw))
      return LR_ERR_UNSUPPORTED;
    hipEvent_t e0, e1;
    const bool sample = lr_prof_next(Cin_pad == 32 ? LR_PROF_CONV2_WGRAD : LR_PROF_CONV3_WGRAD, &e0, &e1);
    return wgrad_tr2_run(X, dP, code, dW, dbias, workspace, accumulate, B * T, T, Hin, Win, Cin_pad, Cout, KT, KH, KW,
                         sample, e0, e1, (hipStream_t)stream);
  }
  (void)pooled;   // the codes carry the ReLU mask (code 4) for the first layer as well
  float* slabs = (float*)workspace;
  float* bpart = (float*)((char*)workspace + need) - (size_t)LR_CONV1_WGRAD_WGS * Cout;
  hipEvent_t e0, e1;
  const bool sample = lr_prof_next(LR_PROF_CONV1_WGRAD, &e0, &e1);
  const int nwg = LR_CONV1_WGRAD_WGS;
  {
    const int st1 = lr_conv1_wgrad(true, u8, X, dP, code, slabs, dbias ? bpart : (float*)nullptr, B * T, T, Hin, Win,
                                   g.Ho, g.Wo, sample, e0, e1, (hipStream_t)stream);
    if (st1 != LR_OK) return st1;
  }
  LR_LAUNCH(conv_wgrad_slab_reduce_kernel, dim3(32 * 320 / 64), dim3(64, 16), 0, stream, (const float*)slabs, nwg,
            (int64_t)32 * 320, dW, 0, 32, 4, 3, 75, 25, 320, accumulate);
  int st = lr_launch_status();
  if (st != LR_OK || !dbias) return st;
  LR_LAUNCH(colsum_final_acc_kernel, dim3(1), dim3(1024), 0, stream, (const float*)bpart, nwg, dbias, Cout,
            accumulate);
  return lr_launch_status();
}

extern "C" int lr_conv3d_wgrad(const void* X, const void* dZ, float* dW, float* dbias, void* workspace,
                               size_t workspace_bytes, int accumulate, int B, int T, int Hin, int Win,
                               int Cin_pad, int Cin_real, int Cout, int KT, int KH, int KW, int stride,
                               int pt, int ph, int pw, lr_stream_t stream) {
  LR_CHECK_ARG(X && dZ && dW && workspace && Cin_real > 0 && Cin_real <= Cin_pad);
  ConvGeom g;
  if (!fill_geom(&g, B, T, Hin, Win, Cin_pad, Cout, KT, KH, KW, stride, pt, ph, pw)) return LR_ERR_UNSUPPORTED;
  if (workspace_bytes < lr_conv3d_wgrad_workspace_bytes(Cout, Cin_pad, KT, KH, KW)) return LR_ERR_WORKSPACE;
  const int max_splits = wgrad_splits(Cout, g.Ktot);
  float* slabs = (float*)workspace;
  float* cpart = (float*)((char*)workspace + lr_conv3d_wgrad_workspace_bytes(Cout, Cin_pad, KT, KH, KW)) -
                 (size_t)kColsumSplits * Cout;
  const bf16_t* x = (const bf16_t*)X;
  const bf16_t* dz = (const bf16_t*)dZ;
  hipEvent_t e0, e1;
  const bool sample = lr_prof_next(Cin_pad == 4 ? LR_PROF_CONV1_WGRAD
                                                : (Cin_pad == 32 ? LR_PROF_CONV2_WGRAD : LR_PROF_CONV3_WGRAD),
                                   &e0, &e1);
  if (Cin_pad == 4 && Cin_real == 3 && Cout == 32 && KT == 3 && KH == 5 && KW == 5 && stride == 2 && pt == 1 &&
      ph == 2 && pw == 2) {
    const int nwg = LR_CONV1_WGRAD_WGS;   // persistent workgroups, partial sums reduced in fixed order
    int st = lr_conv1_wgrad(false, false, x, dz, nullptr, slabs, nullptr, B * T, T, Hin, Win, g.Ho, g.Wo, sample,
                            e0, e1, (hipStream_t)stream);
    if (st != LR_OK) return st;
    LR_LAUNCH(conv_wgrad_slab_reduce_kernel, dim3(32 * 320 / 64), dim3(64, 16), 0, stream, (const float*)slabs, nwg,
              (int64_t)32 * 320, dW, 0, 32, 4, 3, 75, 25, 320, accumulate);
    st = lr_launch_status();
    if (st != LR_OK || !dbias) return st;
    LR_LAUNCH(colsum_bf16_partial_kernel, dim3(kColsumSplits), dim3(256), 0, stream, dz, g.M, Cout, cpart,
              kColsumSplits, (const unsigned char*)nullptr);
    LR_LAUNCH(colsum_final_acc_kernel, dim3(1), dim3(1024), 0, stream, (const float*)cpart, kColsumSplits, dbias,
              Cout, accumulate);
    return lr_launch_status();
  }
  // stride-1 layers of the frontend: LDS-transpose-read kernel (dZ and X stay channels-last in LDS)
  if (wgrad_tr2_layer(B * T, Hin, Win, Cin_pad, Cin_real, Cout, KT, KH, KW, stride, pt, ph, pw))
    return wgrad_tr2_run(X, dZ, nullptr, dW, dbias, workspace, accumulate, B * T, T, Hin, Win, Cin_pad, Cout, KT, KH, KW,
                         sample, e0, e1, (hipStream_t)stream);
  const bool ts_path = stride == 1 && (Cin_pad == 32 || Cin_pad == 64) && Cin_real == Cin_pad &&
                       2 * ph + 1 == KH && 2 * pw + 1 == KW && KH * KW * (Cin_pad / 32) <= 28;
  if (ts_path) {
    // tile: TY rows x the 8-padded width, at most 192 pixels (12 MFMA k steps) and <= 64 kB of LDS
    const int TXP = (g.Wo + 7) / 8 * 8;
    const int MTv = Cout / 32;
    int TY = 192 / TXP;
    if (TY > g.Ho) TY = g.Ho;
    auto fits = [&](int ty) {
      const size_t bytes = ((size_t)ty * TXP * (MTv * 32 + 8) + (size_t)(ty + KH - 1) * (TXP + KW - 1) * (Cin_pad + 8)) * 2;
      return (ty * TXP) % 16 == 0 && bytes <= 60 * 1024 && ty * TXP * MTv * 4 <= 6 * 256 &&
             (ty + KH - 1) * (TXP + KW - 1) * (Cin_pad / 8) <= 6 * 256;
    };
    while (TY > 1 && !fits(TY)) --TY;
    const int npix = TY * TXP;
    const size_t lds = ((size_t)npix * (MTv * 32 + 8) + (size_t)(TY + KH - 1) * (TXP + KW - 1) * (Cin_pad + 8)) * 2;
    const int zu = npix * MTv * 4, pu = (TY + KH - 1) * (TXP + KW - 1) * (Cin_pad / 8);
    const int units = KH * KW * (Cin_pad / 32);
    const int upw = (units + 3) / 4;
    if (fits(TY) && zu <= 6 * 256 && pu <= 6 * 256 && upw <= 7) {
      const dim3 grid(KT * kTsWgsPerKt);
#define LR_WGTS(CI, MTT, UP)                                                                                 \
  do {                                                                                                       \
    lr_clear_error();                                                                                        \
    if (sample) hipExtLaunchKernelGGL((conv3d_wgrad_ts_kernel<CI, MTT, UP>), grid, dim3(256), lds,            \
                                      (hipStream_t)stream, e0, e1, 0, g, x, dz, slabs, TY, TXP, kTsWgsPerKt); \
    else hipLaunchKernelGGL((conv3d_wgrad_ts_kernel<CI, MTT, UP>), grid, dim3(256), lds, (hipStream_t)stream, \
                            g, x, dz, slabs, TY, TXP, kTsWgsPerKt);                                          \
  } while (0)
      bool launched = true;
      if (Cin_pad == 32 && MTv == 2 && upw == 7) LR_WGTS(32, 2, 7);
      else if (Cin_pad == 64 && MTv == 3 && upw == 5) LR_WGTS(64, 3, 5);
      else if (Cin_pad == 32 && MTv == 2 && upw == 3) LR_WGTS(32, 2, 3);
      else launched = false;
#undef LR_WGTS
      if (launched) {
        int st = lr_launch_status();
        if (st != LR_OK) return st;
        {
          const int64_t se = (int64_t)Cout * Cin_pad * KT * KH * KW;
          LR_LAUNCH(conv_wgrad_slab_reduce_kernel, dim3((unsigned)((se + 63) / 64)), dim3(64, 4), 0, stream,
                    (const float*)slabs, kTsWgsPerKt, se, dW, 1, Cout, Cin_pad, Cin_pad, KT * KH * KW, KH * KW, 0,
                    accumulate);
        }
        st = lr_launch_status();
        if (st != LR_OK || !dbias) return st;
        LR_LAUNCH(colsum_bf16_partial_kernel, dim3(kColsumSplits), dim3(256), 0, stream, dz, g.M, Cout, cpart,
                  kColsumSplits, (const unsigned char*)nullptr);
        LR_LAUNCH(colsum_final_acc_kernel, dim3(1), dim3(1024), 0, stream, (const float*)cpart, kColsumSplits,
                  dbias, Cout, accumulate);
        return lr_launch_status();
      }
    }
  }
  int64_t per = (g.M + max_splits - 1) / max_splits;
  per = (per + WG_PIX - 1) / WG_PIX * WG_PIX;
  const int splits = (int)((g.M + per - 1) / per);
  const int nc = wgrad_ntw(Cout) * 32;
  const dim3 grid((g.Ktot + nc - 1) / nc, splits);
#define LR_WGRAD(CI, MTT, NTT)                                                                          \
  do {                                                                                                  \
    lr_clear_error();                                                                                   \
    if (sample) hipExtLaunchKernelGGL((conv3d_wgrad_kernel<CI, MTT, NTT>), grid, dim3(256), 0,           \
                                      (hipStream_t)stream, e0, e1, 0, g, x, dz, slabs, per);            \
    else hipLaunchKernelGGL((conv3d_wgrad_kernel<CI, MTT, NTT>), grid, dim3(256), 0, (hipStream_t)stream, \
                            g, x, dz, slabs, per);                                                      \
  } while (0)
  if (Cin_pad == 4 && Cout == 32) LR_WGRAD(4, 1, 5);
  else if (Cin_pad == 32 && Cout == 64) LR_WGRAD(32, 2, 4);
  else if (Cin_pad == 64 && Cout == 96) LR_WGRAD(64, 3, 2);
  else if (Cin_pad == 32 && Cout == 32) LR_WGRAD(32, 1, 5);
  else if (Cin_pad == 64 && Cout == 64) LR_WGRAD(64, 2, 4);
  else if (Cin_pad == 32 && Cout == 96) LR_WGRAD(32, 3, 2);
  else if (Cin_pad == 64 && Cout == 32) LR_WGRAD(64, 1, 5);
  else return LR_ERR_UNSUPPORTED;
#undef LR_WGRAD
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  const int taps = KT * KH * KW;
  {
    const int64_t se = (int64_t)Cout * g.Ktot;
    LR_LAUNCH(conv_wgrad_slab_reduce_kernel, dim3((unsigned)((se + 63) / 64)), dim3(64, 4), 0, stream,
              (const float*)slabs, splits, se, dW, 0, Cout, Cin_pad, Cin_real, taps, KH * KW, g.Ktot, accumulate);
  }
  st = lr_launch_status();
  if (st != LR_OK || !dbias) return st;
  LR_LAUNCH(colsum_bf16_partial_kernel, dim3(kColsumSplits), dim3(256), 0, stream, dz, g.M, Cout, cpart,
            kColsumSplits, (const unsigned char*)nullptr);
  LR_LAUNCH(colsum_final_acc_kernel, dim3(1), dim3(1024), 0, stream, (const float*)cpart, kColsumSplits,
            dbias, Cout, accumulate);
  return lr_launch_status();
}

extern "C" int lr_maxpool_hw2_bf16(const void* in, void* out, int64_t frames, int H, int W, int C,
                                   lr_stream_t stream) {
  LR_CHECK_ARG(in && out && frames > 0 && H > 1 && W > 1 && C > 0 && H % 2 == 0 && W % 2 == 0);
  LR_LAUNCH(maxpool_hw2_kernel, dim3(grid1d(frames * (H / 2) * (W / 2) * C)), dim3(256), 0, stream,
            (const bf16_t*)in, (bf16_t*)out, frames, H, W, C);
  return lr_launch_status();
}

extern "C" size_t lr_unpool_workspace_bytes(int C) {
  return C > 0 ? (size_t)kUnpoolBlocks * C * sizeof(float) : 0;
}

extern "C" int lr_unpool_relu_mask_bf16(const void* act, const void* dP, void* dZ, float* dbias, int accumulate,
                                        void* workspace, size_t workspace_bytes, int64_t frames, int H, int W,
                                        int C, lr_stream_t stream) {
  LR_CHECK_ARG(act && dP && dZ && frames > 0 && H > 1 && W > 1 && C > 0 && H % 2 == 0 && W % 2 == 0);
  if (C % 8 != 0 || C > 256 || (kUnpoolBlocks * 256) % (C / 8) != 0) return LR_ERR_UNSUPPORTED;
  if (dbias && (!workspace || workspace_bytes < lr_unpool_workspace_bytes(C))) return LR_ERR_WORKSPACE;
  LR_LAUNCH(unpool_relu_mask_kernel, dim3(kUnpoolBlocks), dim3(256), 0, stream, (const bf16_t*)act,
            (const bf16_t*)dP, (bf16_t*)dZ, frames, H, W, C, dbias ? (float*)workspace : (float*)nullptr);
  int st = lr_launch_status();
  if (st != LR_OK || !dbias) return st;
  LR_LAUNCH(colsum_final_acc_kernel, dim3(1), dim3(1024), 0, stream, (const float*)workspace, kUnpoolBlocks, dbias, C,
            accumulate);
  return lr_launch_status();
}

extern "C" int lr_unpool_code_bf16(const void* pooled, const void* code, const void* dP, void* dZ, float* dbias,
                                   int accumulate, void* workspace, size_t workspace_bytes, int64_t frames, int H,
                                   int W, int C, lr_stream_t stream) {
  LR_CHECK_ARG(pooled && code && dP && dZ && frames > 0 && H > 1 && W > 1 && C > 0 && H % 2 == 0 && W % 2 == 0);
  if (C % 8 != 0 || C > 256 || (kUnpoolBlocks * 256) % (C / 8) != 0) return LR_ERR_UNSUPPORTED;
  if (dbias && (!workspace || workspace_bytes < lr_unpool_workspace_bytes(C))) return LR_ERR_WORKSPACE;
  LR_LAUNCH(unpool_code_kernel, dim3(kUnpoolBlocks), dim3(256), 0, stream, (const bf16_t*)pooled,
            (const unsigned char*)code, (const bf16_t*)dP, (bf16_t*)dZ, frames, H, W, C,
            dbias ? (float*)workspace : (float*)nullptr);
  int st = lr_launch_status();
  if (st != LR_OK || !dbias) return st;
  LR_LAUNCH(colsum_final_acc_kernel, dim3(1), dim3(1024), 0, stream, (const float*)workspace, kUnpoolBlocks, dbias, C,
            accumulate);
  return lr_launch_status();
}

extern "C" int lr_bf16_to_f32(const void* in, float* out, int64_t n, lr_stream_t stream) {
  LR_CHECK_ARG(in && out && n > 0);
  LR_LAUNCH(bf16_to_f32_kernel, dim3(grid1d(n)), dim3(256), 0, stream, (const bf16_t*)in, out, n);
  return lr_launch_status();
}

extern "C" int lr_f32_to_bf16(const float* in, void* out, int64_t n, lr_stream_t stream) {
  LR_CHECK_ARG(in && out && n > 0);
  LR_LAUNCH(f32_to_bf16_kernel, dim3(grid1d(n)), dim3(256), 0, stream, in, (bf16_t*)out, n);
  return lr_launch_status();
}
