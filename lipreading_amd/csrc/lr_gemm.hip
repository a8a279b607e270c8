// lr_gemm.hip — fp32 dense contraction on the gfx950 matrix cores.
//
// C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C + bias[N], row-major fp32, computed with
// v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate: bitwise a k-ordered fmaf chain, so the result
// is in the same round-off class as the reference's CPU sgemm).  Used for the parts of the
// encoder that are one-shot contractions (reference better_model.py:74 nn.GRU/LSTM input
// projection and :92 output_proj, plus every weight gradient); the T-step recurrence itself is
// in lr_rnn.hip.
//
// Tiling: 256 threads = 2x2 waves; workgroup tile BM x BN (128x128 or 64x64), wave tile
// (BM/2)x(BN/2) made of 32x32 MFMA tiles, BK = 32 per LDS stage.  Operands are staged through
// LDS in the orientation that keeps BOTH the global loads coalesced and the per-lane MFMA operand
// reads bank-conflict free:
//   K-contiguous operand  -> tile[row][BK+1]   (lane reads row*(BK+1)+k, 33 is odd: no conflict)
//   M/N-contiguous operand -> tile[k][BM+4]    (lane reads k*(BM+4)+row, consecutive lanes)
// Small M*N with long K (the weight gradients: K = B*T) is split along K into slabs in the
// caller's workspace and reduced deterministically (fixed order) by a second kernel.
#include "lr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;   // k per LDS stage: 128 B per operand row per stage, 2x the MFMA work per barrier pair

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;
  int lda, ldb, ldc;
  float alpha, beta;
  int row_shift, period;  // B-row remap along K (transB == 0 only)
  int k_chunk;            // K range per blockIdx.z (multiple of BK)
  float* slabs;           // split-K partial sums [splits][M][N] or nullptr
  bool vecA, vecB;        // operand base + leading dimension allow 16-byte loads
  int batch;              // > 1: blockIdx.z indexes independent problems (no split-K)
  int64_t sA, sB, sC;     // element strides between the problems of a batch (outer index)
  int batch_inner;        // problems per outer index (blockIdx.z = outer * batch_inner + inner)
  int64_t sA2, sB2, sC2;  // element strides of the inner index
  int64_t sBias;          // element stride of `bias` between the problems of a batch (outer index)
};

template <int BM, int BN, bool TA, bool TB>
__device__ __forceinline__ void sgemm_tile(GemmArgs g, const int bx, const int by, const int bz) {
  if (g.batch > 1) {   // batched: one problem per blockIdx.z, the whole K range
    const int zo = bz / g.batch_inner, zi = bz - zo * g.batch_inner;
    g.A += (int64_t)zo * g.sA + (int64_t)zi * g.sA2;
    g.B += (int64_t)zo * g.sB + (int64_t)zi * g.sB2;
    g.C += (int64_t)zo * g.sC + (int64_t)zi * g.sC2;
    if (g.bias) g.bias += (int64_t)zo * g.sBias;
  }
  constexpr int WM = BM / 64;  // 32x32 tiles per wave along M
  constexpr int WN = BN / 64;
  constexpr int LDA_S = TA ? (BM + 4) : (BK + 1);
  constexpr int LDB_S = TB ? (BK + 1) : (BN + 4);
  constexpr int A_ELEMS = TA ? BK * LDA_S : BM * LDA_S;
  constexpr int B_ELEMS = TB ? BN * LDB_S : BK * LDB_S;
  __shared__ __attribute__((aligned(16))) float smem[A_ELEMS + B_ELEMS];
  float* As = smem;
  float* Bs = smem + A_ELEMS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = by * BM;
  const int n0 = bx * BN;
  const int kbeg = g.batch > 1 ? 0 : bz * g.k_chunk;
  const int kend = min(g.K, kbeg + g.k_chunk);

  // per-thread staging registers: BM*BK/1024 float4 of A, BN*BK/1024 float4 of B
  constexpr int A_V4 = BM * BK / 1024;
  constexpr int B_V4 = BN * BK / 1024;
  float4 ra[A_V4], rb[B_V4];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- global -> registers: every thread moves float4s along the operand's contiguous axis ----
  // K-contiguous operand: float4 e -> row = e / (BK/4), k = 4*(e % (BK/4))   (4 lanes per row)
  // M-contiguous operand: float4 e -> k = e / (BM/4),  row = 4*(e % (BM/4))  (lanes along m)
  // `vec` = base pointer and leading dimension 16-byte aligned; otherwise (and at ragged
  // edges) the four elements are fetched one by one with bounds checks.
  auto fetch4 = [&](const float* base, int64_t ld, bool contig_is_minor, int major, int minor,
                    int major_lim, int minor_lim, bool vec) -> float4 {
    // element (major, minor + i) at base[major*ld + minor + i]
    (void)contig_is_minor;
    if (major >= major_lim) return zero4;
    const float* ptr = base + (int64_t)major * ld + minor;
    if (vec && minor + 3 < minor_lim) return *reinterpret_cast<const float4*>(ptr);
    float4 v = zero4;
    if (minor < minor_lim) v.x = ptr[0];
    if (minor + 1 < minor_lim) v.y = ptr[1];
    if (minor + 2 < minor_lim) v.z = ptr[2];
    if (minor + 3 < minor_lim) v.w = ptr[3];
    return v;
  };
  const bool vecA = g.vecA, vecB = g.vecB;
  auto load_a = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_V4; ++i) {
      const int e = tid + i * 256;
      if (TA) {  // stored [K][M]: major = k, minor = m
        const int k = e / (BM / 4), row = 4 * (e % (BM / 4));
        ra[i] = fetch4(g.A, g.lda, true, k0 + k, m0 + row, kend, g.M, vecA);
      } else {   // stored [M][K]: major = m, minor = k
        const int row = e / (BK / 4), k = 4 * (e % (BK / 4));
        ra[i] = fetch4(g.A, g.lda, true, m0 + row, k0 + k, g.M, kend, vecA);
      }
    }
  };
  auto load_b = [&](int k0) {
#pragma unroll
    for (int i = 0; i < B_V4; ++i) {
      const int e = tid + i * 256;
      if (TB) {  // stored [N][K]: major = n, minor = k
        const int col = e / (BK / 4), k = 4 * (e % (BK / 4));
        rb[i] = fetch4(g.B, g.ldb, true, n0 + col, k0 + k, g.N, kend, vecB);
      } else {   // stored [K][N]: major = k (with the optional row remap), minor = n
        const int k = e / (BN / 4), col = 4 * (e % (BN / 4));
        const int gk = k0 + k;
        int src = gk;
        bool ok = gk < kend;
        if (ok && g.period > 0) {
          const int t = gk % g.period + g.row_shift;
          ok = t >= 0 && t < g.period;
          src = gk + g.row_shift;
        }
        rb[i] = ok ? fetch4(g.B, g.ldb, true, src, n0 + col, 1 << 30, g.N, vecB) : zero4;
      }
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < A_V4; ++i) {
      const int e = tid + i * 256;
      if (TA) {
        *reinterpret_cast<float4*>(&As[(e / (BM / 4)) * LDA_S + 4 * (e % (BM / 4))]) = ra[i];
      } else {
        float* dst = &As[(e / (BK / 4)) * LDA_S + 4 * (e % (BK / 4))];
        dst[0] = ra[i].x; dst[1] = ra[i].y; dst[2] = ra[i].z; dst[3] = ra[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < B_V4; ++i) {
      const int e = tid + i * 256;
      if (TB) {
        float* dst = &Bs[(e / (BK / 4)) * LDB_S + 4 * (e % (BK / 4))];
        dst[0] = rb[i].x; dst[1] = rb[i].y; dst[2] = rb[i].z; dst[3] = rb[i].w;
      } else {
        *reinterpret_cast<float4*>(&Bs[(e / (BN / 4)) * LDB_S + 4 * (e % (BN / 4))]) = rb[i];
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31;  // row (A) / col (B) inside a 32x32 tile
  const int lk = lane >> 5;  // k slot 0/1

  if (kbeg < kend) {
    load_a(kbeg);
    load_b(kbeg);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile fully consumed
    store_tiles();
    __syncthreads();
    if (k0 + BK < kend) {  // prefetch the next stage while the MFMAs run
      load_a(k0 + BK);
      load_b(k0 + BK);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float af[WM], bf[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int row = wm * (BM / 2) + i * 32 + lr;
        af[i] = TA ? As[(kk + lk) * LDA_S + row] : As[row * LDA_S + kk + lk];
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = wn * (BN / 2) + j * 32 + lr;
        bf[j] = TB ? Bs[col * LDB_S + kk + lk] : Bs[(kk + lk) * LDB_S + col];
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -------------
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 32 + lr;
      if (col >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= g.M) continue;
        const float v = acc[i][j][r];
        if (g.slabs) {
          g.slabs[((int64_t)bz * g.M + row) * g.N + col] = v;
        } else {
          float out = g.alpha * v;
          if (g.bias) out += g.bias[col];
          float* c = g.C + (int64_t)row * g.ldc + col;
          if (g.beta != 0.f) out += g.beta * *c;
          *c = out;
        }
      }
    }
}

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void sgemm_kernel(GemmArgs g) {
  sgemm_tile<BM, BN, TA, TB>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- grouped launch: several independent C_i = op(A_i) op(B_i) problems of ONE operand orientation in one
// grid (the weight gradients of a recurrent layer: 4-6 small-M*N, long-K products that each fill a sixth
// of the chip on their own).  Every problem is split along K into slabs; one grouped combine follows.
constexpr int kMaxGroup = 8;
struct GroupItem {
  const float* A;
  const float* B;
  float* C;
  float* slabs;
  int M, N, K, lda, ldb, ldc;
  float beta;
  int row_shift, period;
  int tiles_n, tiles, splits, k_chunk, block_begin;
};
struct GroupArgs {
  GroupItem it[kMaxGroup];
  int n;
  // rider (lr_common.h LrRnnBiasJob): workgroups [rider_begin, ...) of the GEMM launch take the partial column sums of
  // dG, row blockIdx.y == n of the combine launch finishes them; rider_begin < 0: none
  int rider_begin, rider_nxb;
  LrRnnBiasJob rider;
};

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void sgemm_grouped_kernel(GroupArgs ga) {
  if (ga.rider_begin >= 0 && (int)blockIdx.x >= ga.rider_begin) {   // workgroup-uniform
    const int rb = blockIdx.x - ga.rider_begin;
    lr_colsum_partial_body(ga.rider.dG, ga.rider.ld, ga.rider.rows, ga.rider.D * 4 * ga.rider.H, ga.rider.partial,
                           rb % ga.rider_nxb, rb / ga.rider_nxb);
    return;
  }
  int i = 0;
#pragma unroll
  for (int k = 1; k < kMaxGroup; ++k)
    if (k < ga.n && (int)blockIdx.x >= ga.it[k].block_begin) i = k;
  const GroupItem& it = ga.it[i];
  const int local = blockIdx.x - it.block_begin;
  const int split = local / it.tiles, tile = local - split * it.tiles;
  GemmArgs g;
  g.A = it.A; g.B = it.B; g.C = it.C; g.bias = nullptr;
  g.M = it.M; g.N = it.N; g.K = it.K;
  g.lda = it.lda; g.ldb = it.ldb; g.ldc = it.ldc;
  g.alpha = 1.f; g.beta = it.beta;
  g.row_shift = it.row_shift; g.period = it.period;
  g.k_chunk = it.k_chunk;
  g.slabs = it.slabs;
  g.vecA = (reinterpret_cast<uintptr_t>(it.A) & 15) == 0 && (it.lda & 3) == 0;
  g.vecB = (reinterpret_cast<uintptr_t>(it.B) & 15) == 0 && (it.ldb & 3) == 0;
  g.batch = 1; g.sA = g.sB = g.sC = 0;
  g.batch_inner = 1; g.sA2 = g.sB2 = g.sC2 = 0; g.sBias = 0;
  sgemm_tile<BM, BN, TA, TB>(g, tile % it.tiles_n, tile / it.tiles_n, split);
}

// grid (ceil(max M*N / 256), items): C_i = sum over the item's slabs in fixed order (+ beta * C_i)
__global__ void grouped_reduce_kernel(GroupArgs ga) {
  if ((int)blockIdx.y == ga.n) {   // the rider's finish (only launched with one)
    const int ncol = ga.rider.D * 4 * ga.rider.H;
    for (int col = blockIdx.x * blockDim.x + threadIdx.x; col < ncol; col += gridDim.x * blockDim.x)
      lr_rnn_bias_final_body(ga.rider, col);
    return;
  }
  const GroupItem& it = ga.it[blockIdx.y];
  if (!it.slabs) return;   // a problem that was not split wrote C itself
  const int64_t total = (int64_t)it.M * it.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < it.splits; ++z) s += it.slabs[(int64_t)z * total + i];
    const int row = (int)(i / it.N), col = (int)(i - (int64_t)row * it.N);
    float* c = it.C + (int64_t)row * it.ldc + col;
    *c = it.beta != 0.f ? s + it.beta * *c : s;
  }
}

// Deterministic split-K combine: fixed summation order over the slabs, then alpha/beta/bias.
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, float* __restrict__ C,
                                     int ldc, const float* __restrict__ bias, int M, int N,
                                     float alpha, float beta) {
  const int64_t total = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(int64_t)z * total + i];
    const int row = (int)(i / N), col = (int)(i - (int64_t)row * N);
    float out = alpha * s;
    if (bias) out += bias[col];
    float* c = C + (int64_t)row * ldc + col;
    if (beta != 0.f) out += beta * *c;
    *c = out;
  }
}

template <int BM, int BN>
void launch_tile(int transA, int transB, const GemmArgs& g, dim3 grid, hipStream_t stream) {
  lr_clear_error();
  if (!transA && !transB) hipLaunchKernelGGL((sgemm_kernel<BM, BN, false, false>), grid, dim3(256), 0, stream, g);
  else if (!transA && transB) hipLaunchKernelGGL((sgemm_kernel<BM, BN, false, true>), grid, dim3(256), 0, stream, g);
  else if (transA && !transB) hipLaunchKernelGGL((sgemm_kernel<BM, BN, true, false>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((sgemm_kernel<BM, BN, true, true>), grid, dim3(256), 0, stream, g);
}

struct GemmPlan {
  bool big;
  int splits;
  int k_chunk;
};

GemmPlan plan_gemm(int M, int N, int K, size_t ws_bytes) {
  GemmPlan p;
  const long tiles_big = (long)((M + 127) / 128) * ((N + 127) / 128);
  const long tiles_small = (long)((M + 63) / 64) * ((N + 63) / 64);
  // how far split-K may go: >= 64 k per split, slabs must fit the caller's workspace
  long max_split = K / (2 * BK);
  const long max_by_ws = (long)(ws_bytes / ((size_t)M * N * sizeof(float)));
  if (max_split > max_by_ws) max_split = max_by_ws;
  if (max_split > 64) max_split = 64;
  if (max_split < 1) max_split = 1;
  // 128x128 tiles (4 MFMA tiles per wave) are ~2x as efficient per workgroup as 64x64, but only
  // pay off when tiles x splits can cover the 256 CUs; otherwise take the finer tiling.
  p.big = tiles_big * max_split >= 200;
  const long tiles = p.big ? tiles_big : tiles_small;
  p.splits = 1;
  if (tiles < 256 && max_split >= 2) {
    long want = (512 + tiles - 1) / tiles;            // aim at ~2 workgroups per CU
    if (want > max_split) want = max_split;
    if (want >= 2) p.splits = (int)want;
  }
  int chunk = (K + p.splits - 1) / p.splits;
  chunk = (chunk + BK - 1) / BK * BK;
  if (chunk < BK) chunk = BK;
  p.k_chunk = chunk;
  p.splits = (K + chunk - 1) / chunk;
  if (p.splits < 1) p.splits = 1;
  return p;
}

}  // namespace

// Internal entry (lr_rnn.hip, lr_proj.hip): same contract as lr_sgemm.
int lr_sgemm_impl(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                  const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                  int row_shift, int period, void* workspace, size_t workspace_bytes,
                  hipStream_t stream) {
  LR_CHECK_ARG(A && B && C);
  LR_CHECK_ARG(M > 0 && N > 0 && K >= 0 && lda > 0 && ldb > 0 && ldc >= N);
  LR_CHECK_ARG(period >= 0 && (period == 0 || !transB));
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta;
  g.row_shift = row_shift; g.period = period;
  g.batch = 1; g.sA = g.sB = g.sC = 0;
  g.batch_inner = 1; g.sA2 = g.sB2 = g.sC2 = 0; g.sBias = 0;
  g.vecA = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0;
  g.vecB = (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (ldb & 3) == 0;
  const GemmPlan p = plan_gemm(M, N, K, workspace ? workspace_bytes : 0);
  g.k_chunk = p.k_chunk;
  g.slabs = p.splits > 1 ? (float*)workspace : nullptr;
  const int bm = p.big ? 128 : 64;
  dim3 grid((N + bm - 1) / bm, (M + bm - 1) / bm, p.splits);
  if (p.big) launch_tile<128, 128>(transA, transB, g, grid, stream);
  else launch_tile<64, 64>(transA, transB, g, grid, stream);
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  if (p.splits > 1) {
    const int64_t total = (int64_t)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    LR_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)workspace,
              p.splits, C, ldc, bias, M, N, alpha, beta);
    st = lr_launch_status();
  }
  return st;
}

// Grouped products with A transposed (the weight-gradient orientation: C_i [M_i x N_i] = A_i^T B_i, A_i stored
// [K][M], B_i stored [K][N] with the optional row remap of lr_sgemm): one GEMM launch + one combine for all of
// them.  workspace: lr_sgemm_grouped_workspace_bytes of the same problem list.
namespace {
// tile edge of a group: 128 when the group is large enough to fill the chip with 128 x 128 tiles (the LSTM-768
// layer: 1536 tiles of 64 x 64 — a 128-tile moves half the operand bytes per flop through LDS), else 64
int group_tile(int n, const int* M, const int* N) {
  long t128 = 0;
  for (int i = 0; i < n; ++i) {
    if (M[i] % 128 != 0 || N[i] < 128) return 64;
    t128 += (long)(M[i] / 128) * ((N[i] + 127) / 128);
  }
  return t128 >= 256 ? 128 : 64;
}
int plan_group(int n, const int* M, const int* N, const int* K, int* splits, int* k_chunk) {
  const int T = group_tile(n, M, N);
  long tiles = 0;
  for (int i = 0; i < n; ++i) tiles += (long)((M[i] + T - 1) / T) * ((N[i] + T - 1) / T);
  const long slots = T == 128 ? 512 : 768;                 // ~2 (128-tiles) / ~3 (64-tiles) workgroups per CU
  long want = tiles > 0 ? (slots + tiles - 1) / tiles : 1;
  for (int i = 0; i < n; ++i) {
    long sp = want, max_split = K[i] / (2 * BK);
    if (sp > max_split) sp = max_split;
    if (sp > 32) sp = 32;
    if (sp < 1) sp = 1;
    int chunk = (int)((K[i] + sp - 1) / sp);
    chunk = (chunk + BK - 1) / BK * BK;
    if (chunk < BK) chunk = BK;
    k_chunk[i] = chunk;
    splits[i] = (K[i] + chunk - 1) / chunk;
    if (splits[i] < 1) splits[i] = 1;
  }
  return 0;
}
}  // namespace

size_t lr_sgemm_grouped_workspace_bytes(int n, const int* M, const int* N, const int* K) {
  if (n <= 0 || n > kMaxGroup) return 0;
  int splits[kMaxGroup], chunk[kMaxGroup];
  plan_group(n, M, N, K, splits, chunk);
  size_t floats = 0;
  for (int i = 0; i < n; ++i)
    if (splits[i] > 1) floats += ((size_t)splits[i] * M[i] * N[i] + 63) / 64 * 64;
  return (floats + 64) * sizeof(float);
}

int lr_sgemm_grouped_tn_impl(int n, const int* M, const int* N, const int* K, const float* const* A, const int* lda,
                             const float* const* B, const int* ldb, float* const* C, const int* ldc, float beta,
                             const int* row_shift, const int* period, void* workspace, size_t workspace_bytes,
                             hipStream_t stream, const LrRnnBiasJob* rider) {
  LR_CHECK_ARG(n > 0 && n <= kMaxGroup && M && N && K && A && lda && B && ldb && C && ldc && workspace);
  if (workspace_bytes < lr_sgemm_grouped_workspace_bytes(n, M, N, K)) return LR_ERR_WORKSPACE;
  int splits[kMaxGroup], chunk[kMaxGroup];
  plan_group(n, M, N, K, splits, chunk);
  GroupArgs ga;
  ga.n = n;
  const int T = group_tile(n, M, N);
  int blocks = 0;
  int64_t biggest = 0;
  float* slab = (float*)workspace;
  for (int i = 0; i < n; ++i) {
    LR_CHECK_ARG(A[i] && B[i] && C[i] && M[i] > 0 && N[i] > 0 && K[i] > 0 && ldc[i] >= N[i]);
    GroupItem& it = ga.it[i];
    it.A = A[i]; it.B = B[i]; it.C = C[i]; it.slabs = splits[i] > 1 ? slab : nullptr;
    it.M = M[i]; it.N = N[i]; it.K = K[i]; it.lda = lda[i]; it.ldb = ldb[i]; it.ldc = ldc[i];
    it.beta = beta;
    it.row_shift = row_shift ? row_shift[i] : 0;
    it.period = period ? period[i] : 0;
    it.tiles_n = (N[i] + T - 1) / T;
    it.tiles = it.tiles_n * ((M[i] + T - 1) / T);
    it.splits = splits[i];
    it.k_chunk = chunk[i];
    it.block_begin = blocks;
    blocks += it.tiles * it.splits;
    if (splits[i] > 1) {
      slab += ((size_t)splits[i] * M[i] * N[i] + 63) / 64 * 64;
      if ((int64_t)M[i] * N[i] > biggest) biggest = (int64_t)M[i] * N[i];
    }
  }
  for (int i = n; i < kMaxGroup; ++i) ga.it[i] = ga.it[0];
  ga.rider_begin = -1;
  ga.rider_nxb = 1;
  if (rider) {
    LR_CHECK_ARG(rider->dG && rider->partial && rider->rows > 0 && rider->H > 0 && (rider->D == 1 || rider->D == 2));
    ga.rider = *rider;
    ga.rider_begin = blocks;
    ga.rider_nxb = (rider->D * 4 * rider->H + 63) / 64;
    blocks += ga.rider_nxb * LR_COLSUM_SPLITS;
  } else {
    ga.rider.dG = nullptr; ga.rider.partial = nullptr;
    ga.rider.db_ih[0] = ga.rider.db_ih[1] = ga.rider.db_hh[0] = ga.rider.db_hh[1] = nullptr;
    ga.rider.ld = ga.rider.rows = ga.rider.H = ga.rider.D = ga.rider.G = ga.rider.accumulate = 0;
  }
  lr_clear_error();
  if (T == 128) hipLaunchKernelGGL((sgemm_grouped_kernel<128, 128, true, false>), dim3(blocks), dim3(256), 0, stream, ga);
  else hipLaunchKernelGGL((sgemm_grouped_kernel<64, 64, true, false>), dim3(blocks), dim3(256), 0, stream, ga);
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  if (biggest == 0 && !rider) return LR_OK;   // nothing was split
  int rb = (int)((biggest + 255) / 256);
  if (rb > 512) rb = 512;
  if (rb < 8) rb = 8;
  LR_LAUNCH(grouped_reduce_kernel, dim3(rb, rider ? n + 1 : n), dim3(256), 0, stream, ga);
  return lr_launch_status();
}

// batch_outer x batch_inner independent products C_z = alpha * op(A_z) op(B_z) + beta * C_z + bias, problem
// z = (o, i) at element offsets o * s?_outer + i * s?_inner (0 = shared operand); 64x64 tiles, no split-K.
int lr_sgemm_batched2_impl(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                           int64_t sA, int64_t sA2, const float* B, int ldb, int64_t sB, int64_t sB2, float beta,
                           float* C, int ldc, int64_t sC, int64_t sC2, const float* bias, int batch_outer,
                           int batch_inner, hipStream_t stream, int64_t sBias) {
  LR_CHECK_ARG(A && B && C);
  LR_CHECK_ARG(M > 0 && N > 0 && K >= 0 && lda > 0 && ldb > 0 && ldc >= N && batch_outer > 0 && batch_inner > 0 &&
               (int64_t)batch_outer * batch_inner <= 65535);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta;
  g.row_shift = 0; g.period = 0;
  const int batch = batch_outer * batch_inner;
  g.batch = batch > 1 ? batch : 1; g.sA = sA; g.sB = sB; g.sC = sC;
  g.batch_inner = batch_inner; g.sA2 = sA2; g.sB2 = sB2; g.sC2 = sC2; g.sBias = sBias;
  g.vecA = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0 && (sA & 3) == 0 && (sA2 & 3) == 0;
  g.vecB = (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (ldb & 3) == 0 && (sB & 3) == 0 && (sB2 & 3) == 0;
  g.k_chunk = (K + BK - 1) / BK * BK;
  if (g.k_chunk < BK) g.k_chunk = BK;
  g.slabs = nullptr;
  // 128 x 128 tiles when they alone fill the chip (the LSTM-768 input projection: 2 x 19 x 24 of them), else 64 x 64
  const long big = (long)((N + 127) / 128) * ((M + 127) / 128) * (batch > 1 ? batch : 1);
  if (big >= 512) {
    dim3 grid((N + 127) / 128, (M + 127) / 128, batch);
    launch_tile<128, 128>(transA, transB, g, grid, stream);
  } else {
    dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
    launch_tile<64, 64>(transA, transB, g, grid, stream);
  }
  return lr_launch_status();
}

int lr_sgemm_batched_impl(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                          int64_t sA, const float* B, int ldb, int64_t sB, float beta, float* C, int ldc,
                          int64_t sC, const float* bias, int batch, hipStream_t stream) {
  return lr_sgemm_batched2_impl(transA, transB, M, N, K, alpha, A, lda, sA, 0, B, ldb, sB, 0, beta, C, ldc, sC, 0,
                                bias, batch, 1, stream, 0);
}

// the same with a per-problem bias (bias + z * sBias): the two directions of a recurrent layer's input projection
int lr_sgemm_batched_bias_impl(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                               int64_t sA, const float* B, int ldb, int64_t sB, float beta, float* C, int ldc,
                               int64_t sC, const float* bias, int64_t sBias, int batch, hipStream_t stream) {
  return lr_sgemm_batched2_impl(transA, transB, M, N, K, alpha, A, lda, sA, 0, B, ldb, sB, 0, beta, C, ldc, sC, 0,
                                bias, batch, 1, stream, sBias);
}

extern "C" int lr_sgemm_batched(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                                int64_t sA_outer, int64_t sA_inner, const float* B, int ldb, int64_t sB_outer,
                                int64_t sB_inner, float beta, float* C, int ldc, int64_t sC_outer, int64_t sC_inner,
                                int batch_outer, int batch_inner, lr_stream_t stream) {
  return lr_sgemm_batched2_impl(transA, transB, M, N, K, alpha, A, lda, sA_outer, sA_inner, B, ldb, sB_outer,
                                sB_inner, beta, C, ldc, sC_outer, sC_inner, nullptr, batch_outer, batch_inner,
                                (hipStream_t)stream, 0);
}

extern "C" size_t lr_sgemm_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  // room for the largest split count plan_gemm may choose
  const GemmPlan p = plan_gemm(M, N, K, (size_t)64 * M * N * sizeof(float));
  return p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
}

extern "C" int lr_sgemm(int transA, int transB, int M, int N, int K, float alpha, const float* A,
                        int lda, const float* B, int ldb, float beta, float* C, int ldc,
                        const float* bias, int row_shift, int period, void* workspace,
                        size_t workspace_bytes, lr_stream_t stream) {
  return lr_sgemm_impl(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias,
                       row_shift, period, workspace, workspace_bytes, (hipStream_t)stream);
}
