// lr_proj.hip — output projection + masked log-softmax (and its backward) for gfx950.
//
// Reference arithmetic replaced here:
//   src/models/lipreader/better_model.py:92  output_logits = self.output_proj(hidden_states)
//   src/models/lipreader/better_model.py:93  masked_log_softmax(logits, output_mask)
//     = log_softmax(logits + log(mask + 1e-45))   (allennlp.nn.util, unpinned third party)
// 1e-45 rounds to the smallest fp32 subnormal (1.4e-45); its log is -103.2789, so masked classes
// stay FINITE.  gfx950 keeps fp32 subnormals (hipcc default float_denorm_mode_32 = 3), which is
// what makes this match the CPU value instead of producing -inf.
//
// The contraction (R x K)·(K x C), C = V+1 = 65, goes through the fp32-MFMA GEMM with the mask
// term folded into the bias; the row-wise log-softmax is one wave per row (C <= 256).
#include "lr_common.h"

int lr_sgemm_impl(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                  const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                  int row_shift, int period, void* workspace, size_t workspace_bytes,
                  hipStream_t stream);
extern "C" size_t lr_sgemm_workspace_bytes(int M, int N, int K);

namespace {

constexpr int kMaxClasses = 256;

__global__ void mask_bias_kernel(const float* __restrict__ bias, const float* __restrict__ mask,
                                 float* __restrict__ out, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  out[c] = bias[c] + logf(mask[c] + 1e-45f);
}

// in place: x[r,:] <- x[r,:] - logsumexp(x[r,:]); one wave per row, 4 rows per workgroup.
__global__ void log_softmax_rows_kernel(float* __restrict__ x, int R, int C) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float* row = x + (int64_t)r * C;
  float v[kMaxClasses / 64];
  float m = LR_NEG_INF;
#pragma unroll
  for (int i = 0; i < kMaxClasses / 64; ++i) {
    const int c = lane + i * 64;
    v[i] = c < C ? row[c] : LR_NEG_INF;
    m = fmaxf(m, v[i]);
  }
  m = lr_wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxClasses / 64; ++i) s += (lane + i * 64 < C) ? expf(v[i] - m) : 0.f;
  s = lr_wave_sum(s);
  const float lse = m + logf(s);
#pragma unroll
  for (int i = 0; i < kMaxClasses / 64; ++i) {
    const int c = lane + i * 64;
    if (c < C) row[c] = v[i] - lse;
  }
}

// dlogits[r,c] = g[r,c] - exp(lp[r,c]) * sum_c g[r,c]
__global__ void log_softmax_bwd_rows_kernel(const float* __restrict__ g, const float* __restrict__ lp,
                                            float* __restrict__ dlogits, int R, int C) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* gr = g + (int64_t)r * C;
  const float* lr = lp + (int64_t)r * C;
  float gv[kMaxClasses / 64];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxClasses / 64; ++i) {
    const int c = lane + i * 64;
    gv[i] = c < C ? gr[c] : 0.f;
    s += gv[i];
  }
  s = lr_wave_sum(s);
#pragma unroll
  for (int i = 0; i < kMaxClasses / 64; ++i) {
    const int c = lane + i * 64;
    if (c < C) dlogits[(int64_t)r * C + c] = gv[i] - expf(lr[c]) * s;
  }
}

// The whole forward head in ONE launch (K % 16 == 0): log_probs[r,:] = log_softmax(hidden[r,:] W^T + bias +
// log(mask + 1e-45)).  The product is tiny (2400 x 65 x 512 at the bench shape) and the unfused chain — folded-bias
// kernel, split-K GEMM, combine pass, row log-softmax — is four launches of ~5-10 us each, which is what it costs.
// A workgroup owns 16 rows; wave w owns classes [16w, 16w + 16) (ceil(C / 16) waves, C <= 256) and runs the exact-
// fp32 v_mfma_f32_16x16x4_f32 over all of K: a lane's float4 of hidden[row][k0 + 4q ..] / W[class][k0 + 4q ..]
// (q = lane / 16) feeds four MFMAs, which contract k in the order k0 + 4q + j — the same permutation on both
// operands.  The 16 x C logits meet in LDS, where sixteen lanes per row reduce max and sum.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void proj_logsoftmax_fused_kernel(const float* __restrict__ hidden,
                                                                     const float* __restrict__ W,
                                                                     const float* __restrict__ bias,
                                                                     const float* __restrict__ mask,
                                                                     float* __restrict__ log_probs, int R, int K,
                                                                     int C) {
  __shared__ float logits[16][kMaxClasses + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int row = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * 16;
  const int rr = min(r0 + row, R - 1);               // rows past the end compute a copy of the last row (never stored)
  const int cls = min(16 * wave + row, C - 1);       // classes past C likewise
  const float4* ap = reinterpret_cast<const float4*>(hidden + (int64_t)rr * K) + q;
  const float4* bp = reinterpret_cast<const float4*>(W + (int64_t)cls * K) + q;
  // two accumulators (even / odd float4 of a group: dependent MFMAs are 4 issues apart), groups of four float4
  // per operand with the NEXT group's eight loads in flight behind the current group's sixteen MFMAs (a single
  // load ahead left every step waiting out an L2 round trip: 50 us at K = 1536)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int steps = K / 16;
  constexpr int GR = 4;
  float4 a[GR], b[GR], an[GR], bn[GR];
#pragma unroll
  for (int i = 0; i < GR; ++i) {
    const int s = i < steps ? i : steps - 1;
    a[i] = ap[s * 4];
    b[i] = bp[s * 4];
  }
  for (int s0 = 0; s0 < steps; s0 += GR) {
#pragma unroll
    for (int i = 0; i < GR; ++i) {
      const int s = s0 + GR + i < steps ? s0 + GR + i : steps - 1;
      an[i] = ap[s * 4];
      bn[i] = bp[s * 4];
    }
#pragma unroll
    for (int i = 0; i < GR; ++i) {
      if (s0 + i < steps) {   // workgroup-uniform
        f32x4& acc = (i & 1) ? acc1 : acc0;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i].w, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < GR; ++i) {
      a[i] = an[i];
      b[i] = bn[i];
    }
  }
  f32x4 acc;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = acc0[i] + acc1[i];
  // D layout: lane holds rows 4q .. 4q + 3 of column (class) 16 wave + (lane & 15)
  {
    const int c = 16 * wave + row;
    const float add = c < C ? bias[c] + logf(mask[c] + 1e-45f) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) logits[4 * q + i][c] = acc[i] + add;
  }
  __syncthreads();
  // sixteen lanes per row, four rows per wave and pass
  for (int rl = wave * 4 + q; rl < 16; rl += nw * 4) {
    const int r = r0 + rl;
    float m = LR_NEG_INF;
    for (int c = row; c < C; c += 16) m = fmaxf(m, logits[rl][c]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int c = row; c < C; c += 16) sum += expf(logits[rl][c] - m);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float lse = m + logf(sum);
    if (r < R)
      for (int c = row; c < C; c += 16) log_probs[(int64_t)r * C + c] = logits[rl][c] - lse;
  }
}

// out[c] = fixed-order sum of the LR_COLSUM_SPLITS partial column sums (lr_colsum_partial).
__global__ void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int C,
                                    int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < LR_COLSUM_SPLITS; ++r) s += partial[(int64_t)r * C + c];
  out[c] = accumulate ? out[c] + s : s;
}

}  // namespace

// sum_{z<n} p[z*stride] with 8 loads in flight; the 8 partial sums are combined in a fixed order (deterministic)
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

// The whole backward head in TWO launches (C <= 128): the unfused chain — row kernel, two split-K GEMMs, their
// combine pass, two column-sum stages — is six launches of 5-10 us each for 0.3 GFLOP, a tenth of the landmark
// regime's step.  (1) A workgroup owns 16 rows x 64 k columns (the grid's second dimension): it forms the rows'
// dlogits in LDS (written out by the first column block), multiplies them with W into its dhidden tile, and contracts
// them with its `hidden` tile over the 16 rows into a private dW slab (+ a dbias row) — both on the exact-fp32
// v_mfma_f32_16x16x4_f32, like the forward head: a wave owns one 16-column tile; dhidden = 17 k steps over the
// classes (eight k steps' W loads in flight), the slab = 4 k steps over the rows per class tile.  The hidden tile is
// loaded before the dlogits phase so that its latency runs under it.  (2) The slabs are summed in workgroup order
// (deterministic) into dW / dbias.
// Measured at R = 2400, K = 512, C = 65 (fused + reduce): scalar FMAs with dlogits broadcast out of LDS, 16 x 256
// tiles: 22.7 + 7.1 us; MFMAs, 16 x 256: 19.4 + 7.5; 16 x 128: 17.7 + 7.4; 16 x 64 (this): 17.2 + 7.4; more rows per
// workgroup write fewer slabs (150 slabs of C x K floats = 20 MB) but run longer serial chains on fewer workgroups:
// 32 x 256: 24.2 + 5.1, 64 x 256: 38.2 + 4.6, 32 x 64: 20.4 + 5.1, 64 x 64: 27.7 + 4.6.  LR_PB_RT / LR_PB_CT: the
// row tiles per workgroup / column tiles per wave of those variants.
#ifndef LR_PB_RT
#define LR_PB_RT 1
#endif
#ifndef LR_PB_CT
#define LR_PB_CT 1
#endif
constexpr int kPbRT = LR_PB_RT, kPbRows = 16 * kPbRT, kPbCT = LR_PB_CT, kPbCols = 64 * kPbCT, kPbThreads = 256, kPbMaxC = 128, kPbLd = kPbMaxC + 4;
__global__ __launch_bounds__(kPbThreads) void proj_bwd_fused_kernel(const float* __restrict__ g,
                                                                    const float* __restrict__ lp,
                                                                    const float* __restrict__ hidden,
                                                                    const float* __restrict__ W,
                                                                    float* __restrict__ dlogits,
                                                                    float* __restrict__ dhidden,
                                                                    float* __restrict__ slabs, int R, int K, int C) {
  __shared__ __attribute__((aligned(16))) float dl[kPbRows][kPbLd];   // [row][class], zeros past C up to kPbMaxC (and past R)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * kPbRows;
  const int Cp = (C + 7) & ~7;        // slab rows
  const int C16 = (C + 15) & ~15;     // classes in LDS: whole MFMA row tiles
  const int nrow = min(kPbRows, R - row0);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int colb = blockIdx.y * kPbCols;     // first k column of the workgroup
  // this wave's four column tiles: columns colb + 16 (wave + 4 q) + l15.  The hidden tile does not depend on dlogits:
  // it is loaded FIRST, so that the loads' latency runs under the dlogits phase.
  int col[kPbCT];
#pragma unroll
  for (int q = 0; q < kPbCT; ++q) col[q] = colb + 16 * (wave + 4 * q) + l15;
  float hb[kPbCT][4 * kPbRT];   // [column tile][k step over the rows]
#pragma unroll
  for (int q = 0; q < kPbCT; ++q)
#pragma unroll
    for (int ks = 0; ks < 4 * kPbRT; ++ks) {
      const int r = 4 * ks + l4;
      hb[q][ks] = (r < nrow && col[q] < K) ? hidden[(int64_t)(row0 + r) * K + col[q]] : 0.f;
    }
  // dlogits[r,c] = g[r,c] - exp(lp[r,c]) * sum_c g[r,c]: a wave per row, 16 rows per wave (four at a time)
#pragma unroll 4
  for (int rr = wave; rr < kPbRows; rr += kPbThreads / 64) {
    const int r = row0 + rr;
    float gv[kPbMaxC / 64], s = 0.f;
#pragma unroll
    for (int i = 0; i < kPbMaxC / 64; ++i) {
      const int c = lane + i * 64;
      gv[i] = (r < R && c < C) ? g[(int64_t)r * C + c] : 0.f;
      s += gv[i];
    }
    s = lr_wave_sum(s);
#pragma unroll
    for (int i = 0; i < kPbMaxC / 64; ++i) {
      const int c = lane + i * 64;
      float d = 0.f;
      if (r < R && c < C) {
        d = gv[i] - expf(lp[(int64_t)r * C + c]) * s;
        if (blockIdx.y == 0) dlogits[(int64_t)r * C + c] = d;
      }
      dl[rr][c] = d;
    }
  }
  __syncthreads();
  float* slab = slabs + (int64_t)blockIdx.x * ((int64_t)Cp * K + Cp);
  if (dhidden) {   // dhidden[r,k] = sum_c dlogits[r,c] W[c,k]: A = dl (row 16 rt + l15, class c0 + l4), B = W (class c0 + l4, column)
    f32x4 acc[kPbRT][kPbCT];
#pragma unroll
    for (int rt = 0; rt < kPbRT; ++rt)
#pragma unroll
      for (int q = 0; q < kPbCT; ++q) acc[rt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8 / kPbCT;   // k steps whose loads are in flight together (8 loads a lane)
    for (int c0 = 0; c0 < C; c0 += 4 * U) {
      float bw[U][kPbCT];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + 4 * u + l4;
#pragma unroll
        for (int q = 0; q < kPbCT; ++q) bw[u][q] = (c < C && col[q] < K) ? W[(int64_t)c * K + col[q]] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int rt = 0; rt < kPbRT; ++rt) {
          const float a = dl[16 * rt + l15][c0 + 4 * u + l4];
#pragma unroll
          for (int q = 0; q < kPbCT; ++q) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[u][q], acc[rt][q], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rt = 0; rt < kPbRT; ++rt)
#pragma unroll
      for (int q = 0; q < kPbCT; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // D: column l15, rows 16 rt + 4 l4 + i
          const int r = 16 * rt + 4 * l4 + i;
          if (r < nrow && col[q] < K) dhidden[(int64_t)(row0 + r) * K + col[q]] = acc[rt][q][i];
        }
  }
  // slab[c,k] = sum_r dlogits[r,c] hidden[r,k]: A = dl^T (class mt*16 + l15, row 4 ks + l4), B = hidden (row 4 ks + l4, column)
  for (int mt = 0; mt < C16 / 16; ++mt) {
    float af[4 * kPbRT];
#pragma unroll
    for (int ks = 0; ks < 4 * kPbRT; ++ks) af[ks] = dl[4 * ks + l4][mt * 16 + l15];
#pragma unroll
    for (int q = 0; q < kPbCT; ++q) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4 * kPbRT; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], hb[q][ks], acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // D: column l15, classes mt*16 + 4 l4 + i
        const int c = mt * 16 + 4 * l4 + i;
        if (c < Cp && col[q] < K) slab[(int64_t)c * K + col[q]] = acc[i];
      }
    }
  }
  if (blockIdx.y == 0 && tid < Cp) {   // the rows' column sums (dbias)
    float sacc = 0.f;
    for (int r = 0; r < kPbRows; ++r) sacc += dl[r][tid];
    slab[(int64_t)Cp * K + tid] = sacc;
  }
}

// out[i] (+)= sum over the workgroups' slabs; i < C*K: dW, then C elements of dbias.  blockDim = (64, 4): thread
// row y sums slabs y, y + 4, ... (8 loads in flight), the four partial sums meet in LDS in a fixed order.
__global__ void proj_bwd_reduce_kernel(const float* __restrict__ slabs, int nslab, int64_t n_dw, int C,
                                       float* __restrict__ dW, float* __restrict__ dbias, int accumulate) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x, y = threadIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const int Cp = (C + 7) & ~7;
  const int64_t K = n_dw / C, per = (int64_t)Cp * K + Cp;   // a slab: [Cp][K] + [Cp]
  float s = 0.f;
  if (i < n_dw + C) {
    const int mine = (nslab - y + 3) / 4;
    s = strided_sum8(slabs + (int64_t)y * per + (i < n_dw ? i : (int64_t)Cp * K + (i - n_dw)), mine, 4 * per);
  }
  part[y][lane] = s;
  __syncthreads();
  if (y != 0 || i >= n_dw + C) return;
  const float tot = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
  float* out = i < n_dw ? dW + i : dbias + (i - n_dw);
  *out = accumulate ? *out + tot : tot;
}

// workspace head: [C] folded bias (forward) or [LR_COLSUM_SPLITS][C] partial sums (backward)
static size_t proj_head_bytes(int C) {
  return lr_align_up((size_t)LR_COLSUM_SPLITS * C * sizeof(float), 256);
}

extern "C" size_t lr_proj_workspace_bytes(int R, int K, int C) {
  if (R <= 0 || K <= 0 || C <= 0) return 0;
  size_t a = lr_sgemm_workspace_bytes(C, K, R);  // dW = dlogits^T @ hidden
  size_t b = lr_sgemm_workspace_bytes(R, K, C);  // dhidden
  size_t c = lr_sgemm_workspace_bytes(R, C, K);  // forward
  size_t m = a > b ? a : b;
  if (c > m) m = c;
  const size_t Cp = ((size_t)C + 7) / 8 * 8;
  const size_t fused = C <= kPbMaxC ? (size_t)((R + kPbRows - 1) / kPbRows) * (Cp * K + Cp) * sizeof(float) : 0;
  if (fused > m) m = fused;   // the fused backward's per-workgroup slabs
  return m + proj_head_bytes(C);
}

extern "C" int lr_proj_logsoftmax_forward(const float* hidden, const float* W, const float* bias,
                                          const float* mask, float* log_probs, void* workspace,
                                          size_t workspace_bytes, int R, int K, int C,
                                          lr_stream_t stream_) {
  LR_CHECK_ARG(hidden && W && bias && mask && log_probs && workspace);
  LR_CHECK_ARG(R > 0 && K > 0 && C > 0);
  if (C > kMaxClasses) return LR_ERR_UNSUPPORTED;
  if (workspace_bytes < lr_proj_workspace_bytes(R, K, C)) return LR_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  if (K % 16 == 0 && ((reinterpret_cast<uintptr_t>(hidden) | reinterpret_cast<uintptr_t>(W)) & 15) == 0) {
    const int nw = (C + 15) / 16;
    LR_LAUNCH(proj_logsoftmax_fused_kernel, dim3((R + 15) / 16), dim3(64 * nw), 0, stream, hidden, W, bias, mask,
              log_probs, R, K, C);
    return lr_launch_status();
  }
  float* mb = (float*)workspace;  // [C] bias + log(mask + 1e-45)
  char* gws = (char*)workspace + proj_head_bytes(C);
  const size_t gws_bytes = workspace_bytes - proj_head_bytes(C);
  LR_LAUNCH(mask_bias_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, bias, mask, mb, C);
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  st = lr_sgemm_impl(0, 1, R, C, K, 1.f, hidden, K, W, K, 0.f, log_probs, C, mb, 0, 0, gws,
                     gws_bytes, stream);
  if (st != LR_OK) return st;
  LR_LAUNCH(log_softmax_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, log_probs, R, C);
  return lr_launch_status();
}

extern "C" int lr_proj_logsoftmax_backward(const float* g, const float* log_probs,
                                           const float* hidden, const float* W, float* dlogits,
                                           float* dhidden, float* dW, float* dbias, void* workspace,
                                           size_t workspace_bytes, int accumulate, int R, int K,
                                           int C, lr_stream_t stream_) {
  LR_CHECK_ARG(g && log_probs && hidden && W && dlogits && dW && dbias && workspace);
  LR_CHECK_ARG(R > 0 && K > 0 && C > 0);
  if (C > kMaxClasses) return LR_ERR_UNSUPPORTED;
  if (workspace_bytes < lr_proj_workspace_bytes(R, K, C)) return LR_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  char* gws = (char*)workspace + proj_head_bytes(C);
  const size_t gws_bytes = workspace_bytes - proj_head_bytes(C);
  if (C <= kPbMaxC) {
    const int nblk = (R + kPbRows - 1) / kPbRows;
    float* slabs = (float*)gws;
    LR_LAUNCH(proj_bwd_fused_kernel, dim3(nblk, (K + kPbCols - 1) / kPbCols), dim3(kPbThreads), 0, stream, g, log_probs, hidden, W, dlogits, dhidden,
              slabs, R, K, C);
    int st0 = lr_launch_status();
    if (st0 != LR_OK) return st0;
    const int64_t n_dw = (int64_t)C * K;
    LR_LAUNCH(proj_bwd_reduce_kernel, dim3((unsigned)((n_dw + C + 63) / 64)), dim3(64, 4), 0, stream,
              (const float*)slabs, nblk, n_dw, C, dW, dbias, accumulate);
    return lr_launch_status();
  }
  LR_LAUNCH(log_softmax_bwd_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, g, log_probs,
            dlogits, R, C);
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  if (dhidden) {
    st = lr_sgemm_impl(0, 0, R, K, C, 1.f, dlogits, C, W, K, 0.f, dhidden, K, nullptr, 0, 0, gws,
                       gws_bytes, stream);
    if (st != LR_OK) return st;
  }
  st = lr_sgemm_impl(1, 0, C, K, R, 1.f, dlogits, C, hidden, K, accumulate ? 1.f : 0.f, dW, K, nullptr, 0, 0, gws,
                     gws_bytes, stream);
  if (st != LR_OK) return st;
  st = lr_colsum_partial(dlogits, C, R, C, (float*)workspace, stream);
  if (st != LR_OK) return st;
  LR_LAUNCH(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, stream,
            (const float*)workspace, dbias, C, accumulate);
  return lr_launch_status();
}
