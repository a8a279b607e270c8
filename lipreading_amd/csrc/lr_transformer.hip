// lr_transformer.hip — the row-wise pieces of a transformer encoder layer on gfx950.
//
// BUILD-DEFINED: the reference has no transformer encoder; BASELINE.json's
// configs[4] names one ("transformer encoder over per-frame conv features (self-attn MFMA path) + CTC").
// The specification is this repo's (lipreading_amd/transformer.py): post-LayerNorm encoder layers
// with ReLU feed-forward, exactly torch.nn.TransformerEncoderLayer(norm_first=False, dropout=0),
// which is also the CPU oracle.  The contractions (QKV / output / feed-forward projections,
// QK^T and PV per (sample, head)) run on the fp32 matrix cores through lr_sgemm / lr_sgemm_batched;
// this file holds what is left: LayerNorm (+ residual), the key-masked softmax of the attention
// scores, ReLU and the positional-encoding add, forward and backward.  All are one pass over
// rows of a few hundred floats: HBM-bound, one wave per row, lanes along the row.
#include "lr_common.h"

namespace {

constexpr int kLnBlocks = 256;   // partial rows of the gamma/beta gradients

// y = LayerNorm(x + residual) * gamma + beta; stats[r] = (mean, rstd).  4 rows per workgroup.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ residual,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ stats, int R, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* xr = x + (int64_t)row * D;
  const float* rr = residual ? residual + (int64_t)row * D : nullptr;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += xr[c] + (rr ? rr[c] : 0.f);
  const float mean = lr_wave_sum(s) / D;
  float v = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float d = xr[c] + (rr ? rr[c] : 0.f) - mean;
    v += d * d;
  }
  const float rstd = rsqrtf(lr_wave_sum(v) / D + eps);   // biased variance, as torch
  for (int c = lane; c < D; c += 64)
    y[(int64_t)row * D + c] = (xr[c] + (rr ? rr[c] : 0.f) - mean) * rstd * gamma[c] + beta[c];
  if (lane == 0) {
    stats[2 * (int64_t)row] = mean;
    stats[2 * (int64_t)row + 1] = rstd;
  }
}

// dx (= d residual) = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;
// partial[block][0][c] = sum_rows dy * xhat, partial[block][1][c] = sum_rows dy (this block's rows).
// One wave per row, a workgroup walks rows blockIdx.x*4 + wave, + 4*gridDim.x, ...
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ residual,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ dy, float* __restrict__ dx,
                                                            float* __restrict__ partial, int R, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);   // [4 waves][2][D]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* mine = red + (size_t)wave * 2 * D;
  for (int c = lane; c < D; c += 64) { mine[c] = 0.f; mine[D + c] = 0.f; }
  for (int row = blockIdx.x * 4 + wave; row < R; row += 4 * gridDim.x) {
    const float* xr = x + (int64_t)row * D;
    const float* rr = residual ? residual + (int64_t)row * D : nullptr;
    const float* dr = dy + (int64_t)row * D;
    const float mean = stats[2 * (int64_t)row], rstd = stats[2 * (int64_t)row + 1];
    float sg = 0.f, sgx = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float xh = (xr[c] + (rr ? rr[c] : 0.f) - mean) * rstd;
      const float g = dr[c] * gamma[c];
      sg += g;
      sgx += g * xh;
      mine[c] += dr[c] * xh;       // a lane owns its columns: no race
      mine[D + c] += dr[c];
    }
    sg = lr_wave_sum(sg) / D;
    sgx = lr_wave_sum(sgx) / D;
    for (int c = lane; c < D; c += 64) {
      const float xh = (xr[c] + (rr ? rr[c] : 0.f) - mean) * rstd;
      dx[(int64_t)row * D + c] = rstd * (dr[c] * gamma[c] - sg - xh * sgx);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256)
    partial[(int64_t)blockIdx.x * 2 * D + c] = red[c] + red[2 * D + c] + red[4 * D + c] + red[6 * D + c];
}

// out[c] (+)= fixed-order sum over blocks of partial[block][c]
__global__ void partial_sum_kernel(const float* __restrict__ partial, int blocks, int ld, int n,
                                   float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += partial[(int64_t)b * ld + c];
  out[c] = accumulate ? out[c] + s : s;
}

// In-place softmax over the keys of scores [B][Hh][T(query)][T(key)] * scale, keys >= key_lens[b]
// masked out (probability exactly 0, as an additive -inf mask gives).  One wave per query row.
__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(float* __restrict__ scores,
                                                               const int32_t* __restrict__ key_lens, float scale,
                                                               int rows, int Hh, int T) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int b = row / (Hh * T);
  const int len = min(max(key_lens[b], 1), T);
  float* s = scores + (int64_t)row * T;
  float m = LR_NEG_INF;
  for (int k = lane; k < len; k += 64) m = fmaxf(m, s[k] * scale);
  m = lr_wave_max(m);
  float z = 0.f;
  for (int k = lane; k < len; k += 64) z += expf(s[k] * scale - m);
  z = lr_wave_sum(z);
  for (int k = lane; k < T; k += 64) s[k] = k < len ? expf(s[k] * scale - m) / z : 0.f;
}

// d scores = scale * P * (dP - sum_k dP P), in place over dP (masked keys have P = 0)
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const float* __restrict__ probs,
                                                               float* __restrict__ dprobs, float scale, int rows,
                                                               int T) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = probs + (int64_t)row * T;
  float* d = dprobs + (int64_t)row * T;
  float s = 0.f;
  for (int k = lane; k < T; k += 64) s += d[k] * p[k];
  s = lr_wave_sum(s);
  for (int k = lane; k < T; k += 64) d[k] = scale * p[k] * (d[k] - s);
}

__global__ void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = fmaxf(x[i], 0.f);
}
__global__ void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
// x[b][t][:] += pe[t][:]
__global__ void add_rows_kernel(float* __restrict__ x, const float* __restrict__ pe, int64_t n, int64_t period) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] += pe[i % period];
}

inline int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  return g < 1 ? 1 : (int)g;
}

}  // namespace

extern "C" int lr_layernorm_forward(const float* x, const float* residual, const float* gamma, const float* beta,
                                    float* y, float* stats, int R, int D, float eps, lr_stream_t stream) {
  LR_CHECK_ARG(x && gamma && beta && y && stats && R > 0 && D > 0);
  LR_LAUNCH(layernorm_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, x, residual, gamma, beta, y, stats, R, D,
            eps);
  return lr_launch_status();
}

extern "C" size_t lr_layernorm_workspace_bytes(int D) {
  return D > 0 ? (size_t)kLnBlocks * 2 * D * sizeof(float) : 0;
}

extern "C" int lr_layernorm_backward(const float* x, const float* residual, const float* gamma, const float* stats,
                                     const float* dy, float* dx, float* dgamma, float* dbeta, void* workspace,
                                     size_t workspace_bytes, int accumulate, int R, int D, lr_stream_t stream) {
  LR_CHECK_ARG(x && gamma && stats && dy && dx && dgamma && dbeta && workspace && R > 0 && D > 0);
  if (workspace_bytes < lr_layernorm_workspace_bytes(D)) return LR_ERR_WORKSPACE;
  const size_t lds = (size_t)4 * 2 * D * sizeof(float);
  if (lds > 60 * 1024) return LR_ERR_UNSUPPORTED;
  float* partial = (float*)workspace;
  LR_LAUNCH(layernorm_bwd_kernel, dim3(kLnBlocks), dim3(256), lds, stream, x, residual, gamma, stats, dy, dx, partial,
            R, D);
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  LR_LAUNCH(partial_sum_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, (const float*)partial, kLnBlocks, 2 * D, D,
            dgamma, accumulate);
  LR_LAUNCH(partial_sum_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, (const float*)(partial + D), kLnBlocks,
            2 * D, D, dbeta, accumulate);
  return lr_launch_status();
}

extern "C" int lr_attn_softmax_forward(float* scores, const int32_t* key_lens, float scale, int B, int Hh, int T,
                                       lr_stream_t stream) {
  LR_CHECK_ARG(scores && key_lens && B > 0 && Hh > 0 && T > 0);
  const int rows = B * Hh * T;
  LR_LAUNCH(attn_softmax_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, scores, key_lens, scale, rows, Hh, T);
  return lr_launch_status();
}

extern "C" int lr_attn_softmax_backward(const float* probs, float* dprobs, float scale, int B, int Hh, int T,
                                        lr_stream_t stream) {
  LR_CHECK_ARG(probs && dprobs && B > 0 && Hh > 0 && T > 0);
  const int rows = B * Hh * T;
  LR_LAUNCH(attn_softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, probs, dprobs, scale, rows, T);
  return lr_launch_status();
}

extern "C" int lr_relu_forward(const float* x, float* y, int64_t n, lr_stream_t stream) {
  LR_CHECK_ARG(x && y && n > 0);
  LR_LAUNCH(relu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n);
  return lr_launch_status();
}

extern "C" int lr_relu_backward(const float* y, const float* dy, float* dx, int64_t n, lr_stream_t stream) {
  LR_CHECK_ARG(y && dy && dx && n > 0);
  LR_LAUNCH(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, y, dy, dx, n);
  return lr_launch_status();
}

extern "C" int lr_add_rows(float* x, const float* pe, int B, int T, int D, lr_stream_t stream) {
  LR_CHECK_ARG(x && pe && B > 0 && T > 0 && D > 0);
  const int64_t n = (int64_t)B * T * D;
  LR_LAUNCH(add_rows_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, pe, n, (int64_t)T * D);
  return lr_launch_status();
}
