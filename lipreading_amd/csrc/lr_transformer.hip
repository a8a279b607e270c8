// lr_transformer.hip — the row-wise pieces of a transformer encoder layer on gfx950.
//
// BUILD-DEFINED: the reference has no transformer encoder; BASELINE.json's
// configs[4] names one ("transformer encoder over per-frame conv features (self-attn MFMA path) + CTC").
// The specification is this repo's (lipreading_amd/transformer.py): post-LayerNorm encoder layers
// with ReLU feed-forward, exactly torch.nn.TransformerEncoderLayer(norm_first=False, dropout=0),
// which is also the CPU oracle.  This file holds the row-wise kernels — LayerNorm, the key-masked softmax of
// the unfused attention — and, since round 5, the STACK itself (lr_tfm_forward / backward_data / backward_weights):
// the whole encoder as three enqueues from C++, every projection a lr_fgemm.hip product that takes its operands as
// they lie in memory and carries bias / residual / positional table / ReLU / ReLU mask in its epilogue, every weight
// gradient of every layer (with its bias gradient) in ONE launch, LayerNorm parameter gradients in one more.
// Round 4 composed the same arithmetic per operation from Python: 375 launches per step at the bench shape, 58 of
// them ATen adds; this is 29 forward + 30 backward launches.  Round 6 (mode bit LR_TFM_ROWBLOCK, lr_tfm_rowblock.hip):
// out-projection .. LN2, and the neighbouring layer's QKV product, as ONE launch per layer and direction over 32-row
// blocks — 12 forward + 10 backward launches for the four-layer stack (input projection + its K-split combine, the
// weight pack, layer 0's QKV, four x (attention, row block); four x (row block, attention), layer 0's input gradient, dx).
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

// LayerNorm parameter gradients of up to LN_MAX_JOBS LayerNorms in one launch: out[job][c] (+)= fixed-order sum over the
// kLnBlocks partial rows of partial[job][block][c], c in [0, 2 D) (dgamma | dbeta side by side in a partial row).
// grid (ceil(2 D / 64), jobs), 256 threads = 4 block groups x 64 columns.
constexpr int LN_MAX_JOBS = 32;
struct LnJobs {
  const float* partial[LN_MAX_JOBS];
  float* dgamma[LN_MAX_JOBS];
  float* dbeta[LN_MAX_JOBS];
};
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const LnJobs jobs, int D, int accumulate) {
  __shared__ float red[4][64];
  const int job = blockIdx.y, cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cg;
  const float* p = jobs.partial[job];
  float s = 0.f;
  if (c < 2 * D)
    for (int b = rg; b < kLnBlocks; b += 4) s += p[(int64_t)b * 2 * D + c];
  red[rg][cg] = s;
  __syncthreads();
  if (rg == 0 && c < 2 * D) {
    s = ((red[0][cg] + red[1][cg]) + red[2][cg]) + red[3][cg];
    float* out = c < D ? jobs.dgamma[job] + c : jobs.dbeta[job] + (c - D);
    *out = accumulate ? *out + s : s;
  }
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
  LnJobs jobs;
  jobs.partial[0] = partial;
  jobs.dgamma[0] = dgamma;
  jobs.dbeta[0] = dbeta;
  LR_LAUNCH(ln_param_reduce_kernel, dim3((2 * D + 63) / 64, 1), dim3(256), 0, stream, jobs, D, accumulate);
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


// ---------------------------------------------------------------------------------------------------------------------
// The encoder stack (include/lipreading_hip.h lr_tfm_*).  Row count R = B * T; every tensor [R][width] fp32, contiguous.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

inline size_t al64(size_t n) { return (n + 63) / 64 * 64; }

// what the forward keeps for the backward (float offsets into `reserve`)
struct TfmReserve {
  size_t h0, layer0, per_layer, total;
  size_t qkv, a, s1, st1, h1, f1, s2, st2, h2, probs;   // inside a layer's block
  size_t planes;                                        // LR_TFM_ROWBLOCK: the bf16 weight planes of every layer
};
TfmReserve tfm_reserve(int mode, int B, int T, int Dm, int nh, int F, int nl) {
  TfmReserve r;
  const size_t R = (size_t)B * T;
  r.h0 = 0;
  r.layer0 = al64(R * Dm);
  size_t p = 0;
  r.qkv = p; p += al64(R * 3 * Dm);
  r.a = p; p += al64(R * Dm);
  r.s1 = p; p += al64(R * Dm);
  r.st1 = p; p += al64(2 * R);
  r.h1 = p; p += al64(R * Dm);
  r.f1 = p; p += al64(R * F);
  r.s2 = p; p += al64(R * Dm);
  r.st2 = p; p += al64(2 * R);
  r.h2 = p; p += al64(R * Dm);
  r.probs = p; p += (mode & LR_TFM_ATTN_FUSED) ? 0 : al64((size_t)B * nh * T * T);
  r.per_layer = p;
  r.planes = r.layer0 + (size_t)nl * p;
  r.total = r.planes + ((mode & LR_TFM_ROWBLOCK) ? al64((size_t)nl * lr_tfm_rb_plane_elems(F) / 2) : 0);
  return r;
}

// the backward's buffers: a layer's pre-activation gradients live until backward_weights has contracted them
struct TfmWs {
  size_t layer0, per_layer, ds2, df1, ds1, dqkv, lnp1, lnp2;   // per layer
  size_t dh1, da, dhA, dhB, dP, slabs, slab_floats, total;
};
TfmWs tfm_ws(int mode, int B, int T, int I, int Dm, int nh, int F, int nl) {
  TfmWs w;
  const size_t R = (size_t)B * T, lnp = (size_t)kLnBlocks * 2 * Dm;
  size_t p = 0;
  w.ds2 = p; p += al64(R * Dm);
  w.df1 = p; p += al64(R * F);
  w.ds1 = p; p += al64(R * Dm);
  w.dqkv = p; p += al64(R * 3 * Dm);
  w.lnp1 = p; p += al64(lnp);
  w.lnp2 = p; p += al64(lnp);
  w.per_layer = p;
  w.layer0 = 0;
  size_t o = (size_t)nl * p;
  w.dh1 = o; o += al64(R * Dm);
  w.da = o; o += al64(R * Dm);
  w.dhA = o; o += al64(R * Dm);
  w.dhB = o; o += al64(R * Dm);
  w.dP = o; o += (mode & LR_TFM_ATTN_FUSED) ? 0 : al64((size_t)B * nh * T * T);
  w.slabs = o;
  w.slab_floats = al64(lr_fgemm_slab_floats_impl((int)R, Dm, lr_fgemm_want_splits((int)R, Dm, I)));
  o += w.slab_floats;
  w.total = o;
  return w;
}

bool tfm_dims_ok(int mode, int B, int T, int I, int Dm, int nh, int F, int nl) {
  return B > 0 && T > 0 && I > 0 && Dm > 0 && nh > 0 && F > 0 && nl > 0 && Dm % nh == 0 && Dm % 4 == 0 && F % 4 == 0 &&
         (Dm / nh) % 4 == 0 && 2 * nl <= LN_MAX_JOBS && (size_t)4 * 2 * Dm * sizeof(float) <= 60 * 1024 &&
         (mode & ~(LR_TFM_X3 | LR_TFM_X_BF16 | LR_TFM_DX_BF16 | LR_TFM_ATTN_FUSED | LR_TFM_ROWBLOCK)) == 0 &&
         (!(mode & LR_TFM_ATTN_FUSED) || lr_attn_fused_supported(T, Dm / nh)) &&
         (!(mode & LR_TFM_ROWBLOCK) || ((mode & LR_TFM_X3) && lr_tfm_rb_supported(Dm, F, nl)));
}

lr_fgemm_job job(const void* A, int lda, const void* Bm, int ldb, void* C, int ldc, int M, int N, int K) {
  lr_fgemm_job j;
  j.A = A; j.B = Bm; j.C = C;
  j.bias = nullptr; j.addend = nullptr; j.mask = nullptr; j.colsum = nullptr; j.slabs = nullptr;
  j.M = M; j.N = N; j.K = K; j.lda = lda; j.ldb = ldb; j.ldc = ldc;
  j.ldadd = 0; j.add_period = 0; j.ldmask = 0; j.flags = 0; j.splits = 1;
  j.alpha = 1.f; j.beta = 0.f;
  j.b_shift = 0; j.b_period = 0;
  return j;
}

#define LR_TRY_(expr)              \
  do {                             \
    const int lr_st_ = (expr);     \
    if (lr_st_ != LR_OK) return lr_st_; \
  } while (0)

// per (sample, head): P = softmax(scale Q K^T), a = P V on the fp32 matrix cores (probs kept for the backward)
int attention_f32_forward(const float* qkv, const int32_t* lens, float* probs, float* out, int B, int T, int nh, int dh,
                          hipStream_t st) {
  const int D = nh * dh, D3 = 3 * D;
  const float scale = 1.f / sqrtf((float)dh);
  LR_TRY_(lr_sgemm_batched(0, 1, T, T, dh, 1.f, qkv, D3, (int64_t)T * D3, dh, qkv + D, D3, (int64_t)T * D3, dh, 0.f, probs, T,
                           (int64_t)nh * T * T, (int64_t)T * T, B, nh, st));
  LR_TRY_(lr_attn_softmax_forward(probs, lens, scale, B, nh, T, st));
  return lr_sgemm_batched(0, 0, T, dh, T, 1.f, probs, T, (int64_t)nh * T * T, (int64_t)T * T, qkv + 2 * D, D3, (int64_t)T * D3, dh,
                          0.f, out, D, (int64_t)T * D, dh, B, nh, st);
}
int attention_f32_backward(const float* qkv, const float* probs, const float* dout, float* dP, float* dqkv, int B, int T,
                           int nh, int dh, hipStream_t st) {
  const int D = nh * dh, D3 = 3 * D;
  const float scale = 1.f / sqrtf((float)dh);
  const int64_t PS = (int64_t)nh * T * T, PI = (int64_t)T * T;
  const float *q = qkv, *k = qkv + D, *v = qkv + 2 * D;
  float *dq = dqkv, *dk = dqkv + D, *dv = dqkv + 2 * D;
  // dP = dO V^T ; dV = P^T dO ; dS = softmax backward ; dQ = dS K ; dK = dS^T Q
  LR_TRY_(lr_sgemm_batched(0, 1, T, T, dh, 1.f, dout, D, (int64_t)T * D, dh, v, D3, (int64_t)T * D3, dh, 0.f, dP, T, PS, PI, B, nh, st));
  LR_TRY_(lr_sgemm_batched(1, 0, T, dh, T, 1.f, probs, T, PS, PI, dout, D, (int64_t)T * D, dh, 0.f, dv, D3, (int64_t)T * D3, dh, B, nh, st));
  LR_TRY_(lr_attn_softmax_backward(probs, dP, scale, B, nh, T, st));
  LR_TRY_(lr_sgemm_batched(0, 0, T, dh, T, 1.f, dP, T, PS, PI, k, D3, (int64_t)T * D3, dh, 0.f, dq, D3, (int64_t)T * D3, dh, B, nh, st));
  return lr_sgemm_batched(1, 0, T, dh, T, 1.f, dP, T, PS, PI, q, D3, (int64_t)T * D3, dh, 0.f, dk, D3, (int64_t)T * D3, dh, B, nh, st);
}

int ln_forward(const float* s, const float* gamma, const float* beta, float* y, float* stats, int R, int D, float eps,
               hipStream_t st) {
  LR_LAUNCH(layernorm_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, st, s, (const float*)nullptr, gamma, beta, y, stats, R, D, eps);
  return lr_launch_status();
}
int ln_backward(const float* s, const float* gamma, const float* stats, const float* dy, float* ds, float* partial, int R,
                int D, hipStream_t st) {
  LR_LAUNCH(layernorm_bwd_kernel, dim3(kLnBlocks), dim3(256), (size_t)4 * 2 * D * sizeof(float), st, s, (const float*)nullptr,
            gamma, stats, dy, ds, partial, R, D);
  return lr_launch_status();
}

}  // namespace

extern "C" size_t lr_tfm_reserve_bytes(int mode, int B, int T, int I, int Dm, int nhead, int F, int nlayers) {
  if (!tfm_dims_ok(mode, B, T, I, Dm, nhead, F, nlayers)) return 0;
  return tfm_reserve(mode, B, T, Dm, nhead, F, nlayers).total * sizeof(float);
}
extern "C" size_t lr_tfm_workspace_bytes(int mode, int B, int T, int I, int Dm, int nhead, int F, int nlayers) {
  if (!tfm_dims_ok(mode, B, T, I, Dm, nhead, F, nlayers)) return 0;
  return tfm_ws(mode, B, T, I, Dm, nhead, F, nlayers).total * sizeof(float);
}

extern "C" int lr_tfm_forward(int mode, const void* x, const int32_t* key_lens, const float* const* weights, const float* pe,
                              float* h_out, void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                              int B, int T, int I, int Dm, int nhead, int F, int nlayers, float eps, lr_stream_t stream_) {
  LR_CHECK_ARG(tfm_dims_ok(mode, B, T, I, Dm, nhead, F, nlayers));
  LR_CHECK_ARG(x && key_lens && weights && pe && h_out && reserve && workspace);
  for (int i = 0; i < 2 + 12 * nlayers; ++i) LR_CHECK_ARG(weights[i]);
  const TfmReserve r = tfm_reserve(mode, B, T, Dm, nhead, F, nlayers);
  const TfmWs w = tfm_ws(mode, B, T, I, Dm, nhead, F, nlayers);
  if (reserve_bytes < r.total * sizeof(float) || workspace_bytes < w.total * sizeof(float)) return LR_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream_;
  const int prec = (mode & LR_TFM_X3) ? LR_FGEMM_X3 : LR_FGEMM_F32;
  const int xbf = (mode & LR_TFM_X_BF16) ? 1 : 0;
  const int R = B * T, dh = Dm / nhead;
  float* base = (float*)reserve;
  float* wsb = (float*)workspace;
  // h0 = x W_p^T + b_p + pe[t]   (K = I may be long and the product has few tiles: split K)
  float* h = base + r.h0;
  {
    lr_fgemm_job j = job(x, I, weights[0], I, h, Dm, R, Dm, I);
    j.bias = weights[1];
    j.addend = pe; j.ldadd = Dm; j.add_period = T;
    j.splits = lr_fgemm_want_splits(R, Dm, I);
    j.slabs = wsb + w.slabs;
    LR_TRY_(lr_fgemm_launch(prec, LR_FGEMM_NT, xbf, 0, &j, 1, st));
  }
  const bool rowblock = (mode & LR_TFM_ROWBLOCK) != 0;
  if (rowblock) LR_TRY_(lr_tfm_rb_pack(weights, base + r.planes, F, nlayers, st));
  for (int l = 0; l < nlayers; ++l) {
    const float* const* W = weights + 2 + 12 * l;
    float* L = base + r.layer0 + (size_t)l * r.per_layer;
    float* h2 = l == nlayers - 1 ? h_out : L + r.h2;
    lr_fgemm_job j = job(h, Dm, W[0], Dm, L + r.qkv, 3 * Dm, R, 3 * Dm, Dm);
    j.bias = W[1];
    if (!(rowblock && l > 0))   // (row blocks: layer l - 1's launch wrote this layer's qkv on its way out)
      LR_TRY_(lr_fgemm_launch(prec, LR_FGEMM_NT, 0, 0, &j, 1, st));
    if (mode & LR_TFM_ATTN_FUSED)
      LR_TRY_(lr_attn_fused_forward(L + r.qkv, key_lens, L + r.a, 1.f / sqrtf((float)dh), B, T, nhead, dh, st));
    else
      LR_TRY_(attention_f32_forward(L + r.qkv, key_lens, L + r.probs, L + r.a, B, T, nhead, dh, st));
    if (rowblock) {   // out-projection .. LN2 in one launch (lr_tfm_rowblock.hip)
      const bool more = l + 1 < nlayers;   // the next layer's qkv = h2 W_qkv'^T + b' rides at the end of the chain
      LR_TRY_(lr_tfm_rb_forward(base + r.planes, l, W, L + r.a, h, L + r.s1, L + r.st1, L + r.h1, L + r.f1, L + r.s2,
                                L + r.st2, h2, more ? W + 12 : nullptr, more ? L + r.per_layer + r.qkv : nullptr, R, F, eps,
                                st));
      h = h2;
      continue;
    }
    j = job(L + r.a, Dm, W[2], Dm, L + r.s1, Dm, R, Dm, Dm);      // s1 = a W_o^T + b_o + h
    j.bias = W[3];
    j.addend = h; j.ldadd = Dm; j.add_period = R;
    LR_TRY_(lr_fgemm_launch(prec, LR_FGEMM_NT, 0, 0, &j, 1, st));
    LR_TRY_(ln_forward(L + r.s1, W[8], W[9], L + r.h1, L + r.st1, R, Dm, eps, st));
    j = job(L + r.h1, Dm, W[4], Dm, L + r.f1, F, R, F, Dm);        // f1 = relu(h1 W_1^T + b_1)
    j.bias = W[5];
    j.flags = LR_FGEMM_RELU;
    LR_TRY_(lr_fgemm_launch(prec, LR_FGEMM_NT, 0, 0, &j, 1, st));
    j = job(L + r.f1, F, W[6], F, L + r.s2, Dm, R, Dm, F);         // s2 = f1 W_2^T + b_2 + h1
    j.bias = W[7];
    j.addend = L + r.h1; j.ldadd = Dm; j.add_period = R;
    LR_TRY_(lr_fgemm_launch(prec, LR_FGEMM_NT, 0, 0, &j, 1, st));
    LR_TRY_(ln_forward(L + r.s2, W[10], W[11], h2, L + r.st2, R, Dm, eps, st));
    h = h2;
  }
  return LR_OK;
}

extern "C" int lr_tfm_backward_data(int mode, const int32_t* key_lens, const float* const* weights, const float* dh_out,
                                    void* dx, void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                                    int B, int T, int I, int Dm, int nhead, int F, int nlayers, lr_stream_t stream_) {
  LR_CHECK_ARG(tfm_dims_ok(mode, B, T, I, Dm, nhead, F, nlayers));
  LR_CHECK_ARG(key_lens && weights && dh_out && reserve && workspace);
  const TfmReserve r = tfm_reserve(mode, B, T, Dm, nhead, F, nlayers);
  const TfmWs w = tfm_ws(mode, B, T, I, Dm, nhead, F, nlayers);
  if (reserve_bytes < r.total * sizeof(float) || workspace_bytes < w.total * sizeof(float)) return LR_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream_;
  const int prec = (mode & LR_TFM_X3) ? LR_FGEMM_X3 : LR_FGEMM_F32;
  const int R = B * T, dh = Dm / nhead;
  float* base = (float*)reserve;
  float* wsb = (float*)workspace;
  const float* dh_cur = dh_out;
  for (int l = nlayers - 1; l >= 0; --l) {
    const float* const* W = weights + 2 + 12 * l;
    float* L = base + r.layer0 + (size_t)l * r.per_layer;
    float* G = wsb + w.layer0 + (size_t)l * w.per_layer;
    // the gradient of layer l's input: layer 0's lands in dhA (backward_weights reads it there)
    float* dh_prev = wsb + ((l & 1) ? w.dhB : w.dhA);
    lr_fgemm_job j;
    if (mode & LR_TFM_ROWBLOCK) {   // LN2' .. the out-projection's data gradient in one launch (lr_tfm_rowblock.hip)
      // (below the top layer the chain starts with the input gradient of the layer above: dqkv' W_qkv' + ds1')
      const bool up = l + 1 < nlayers;
      LR_TRY_(lr_tfm_rb_backward(base + r.planes, l, W, dh_cur, up ? G + w.per_layer + w.dqkv : nullptr,
                                 up ? G + w.per_layer + w.ds1 : nullptr, L + r.s2, L + r.st2, L + r.f1, L + r.s1,
                                 L + r.st1, G + w.ds2, G + w.df1, G + w.ds1, wsb + w.da, G + w.lnp2, G + w.lnp1, kLnBlocks, R,
                                 F, st));
    } else {
      LR_TRY_(ln_backward(L + r.s2, W[10], L + r.st2, dh_cur, G + w.ds2, G + w.lnp2, R, Dm, st));
      j = job(G + w.ds2, Dm, W[6], F, G + w.df1, F, R, F, Dm);      // df1 = (ds2 W_2) where f1 > 0
      j.mask = L + r.f1; j.ldmask = F;
      LR_TRY_(lr_fgemm_launch(prec, LR_FGEMM_NN, 0, 0, &j, 1, st));
      j = job(G + w.df1, F, W[4], Dm, wsb + w.dh1, Dm, R, Dm, F);                // dh1 = df1 W_1 + ds2
      j.addend = G + w.ds2; j.ldadd = Dm; j.add_period = R;
      LR_TRY_(lr_fgemm_launch(prec, LR_FGEMM_NN, 0, 0, &j, 1, st));
      LR_TRY_(ln_backward(L + r.s1, W[8], L + r.st1, wsb + w.dh1, G + w.ds1, G + w.lnp1, R, Dm, st));
      j = job(G + w.ds1, Dm, W[2], Dm, wsb + w.da, Dm, R, Dm, Dm);               // da = ds1 W_o
      LR_TRY_(lr_fgemm_launch(prec, LR_FGEMM_NN, 0, 0, &j, 1, st));
    }
    if (mode & LR_TFM_ATTN_FUSED)
      LR_TRY_(lr_attn_fused_backward(L + r.qkv, key_lens, wsb + w.da, G + w.dqkv, 1.f / sqrtf((float)dh), B, T, nhead, dh, st));
    else
      LR_TRY_(attention_f32_backward(L + r.qkv, L + r.probs, wsb + w.da, wsb + w.dP, G + w.dqkv, B, T, nhead, dh, st));
    j = job(G + w.dqkv, 3 * Dm, W[0], Dm, dh_prev, Dm, R, Dm, 3 * Dm);         // dh = dqkv W_qkv + ds1
    j.addend = G + w.ds1; j.ldadd = Dm; j.add_period = R;
    if (!((mode & LR_TFM_ROWBLOCK) && l > 0))   // (row blocks: the layer below takes it in at the start of its chain)
      LR_TRY_(lr_fgemm_launch(prec, LR_FGEMM_NN, 0, 0, &j, 1, st));
    dh_cur = dh_prev;
  }
  if (dx) {   // dx = dh0 W_p
    lr_fgemm_job j = job(dh_cur, Dm, weights[0], I, dx, I, R, I, Dm);
    if (mode & LR_TFM_DX_BF16) j.flags = LR_FGEMM_C_BF16;
    LR_TRY_(lr_fgemm_launch(prec, LR_FGEMM_NN, 0, 0, &j, 1, st));
  }
  return LR_OK;
}

extern "C" int lr_tfm_backward_weights(int mode, const void* x, float* const* grads, int accumulate, void* reserve,
                                       size_t reserve_bytes, void* workspace, size_t workspace_bytes, int B, int T, int I,
                                       int Dm, int nhead, int F, int nlayers, lr_stream_t stream_) {
  LR_CHECK_ARG(tfm_dims_ok(mode, B, T, I, Dm, nhead, F, nlayers));
  LR_CHECK_ARG(x && grads && reserve && workspace);
  for (int i = 0; i < 2 + 12 * nlayers; ++i) LR_CHECK_ARG(grads[i]);
  const TfmReserve r = tfm_reserve(mode, B, T, Dm, nhead, F, nlayers);
  const TfmWs w = tfm_ws(mode, B, T, I, Dm, nhead, F, nlayers);
  if (reserve_bytes < r.total * sizeof(float) || workspace_bytes < w.total * sizeof(float)) return LR_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream_;
  const int prec = (mode & LR_TFM_X3) ? LR_FGEMM_X3 : LR_FGEMM_F32;
  const int xbf = (mode & LR_TFM_X_BF16) ? 1 : 0;
  const int R = B * T;
  float* base = (float*)reserve;
  float* wsb = (float*)workspace;
  const float beta = accumulate ? 1.f : 0.f;
  // dW = dy^T x with db = column sums of dy, for every Linear of the stack: jobs of ONE launch (20 at a time)
  lr_fgemm_job jobs[20];
  int n = 0;
  auto flush = [&](int b_bf16) -> int {
    if (n == 0) return LR_OK;
    const int st_ = lr_fgemm_launch(prec, LR_FGEMM_TN, 0, b_bf16, jobs, n, st);
    n = 0;
    return st_;
  };
  auto add = [&](const float* dy, int ldy, const void* xin, int ldx, float* dW, float* db, int M, int N) {
    lr_fgemm_job j = job(dy, ldy, xin, ldx, dW, N, M, N, R);
    j.colsum = db;
    j.beta = beta;
    jobs[n++] = j;
  };
  for (int l = 0; l < nlayers; ++l) {
    float* const* Gd = grads + 2 + 12 * l;
    float* L = base + r.layer0 + (size_t)l * r.per_layer;
    float* G = wsb + w.layer0 + (size_t)l * w.per_layer;
    const float* h_in = l == 0 ? base + r.h0 : base + r.layer0 + (size_t)(l - 1) * r.per_layer + r.h2;
    if (n + 4 > 20) LR_TRY_(flush(0));
    add(G + w.dqkv, 3 * Dm, h_in, Dm, Gd[0], Gd[1], 3 * Dm, Dm);
    add(G + w.ds1, Dm, L + r.a, Dm, Gd[2], Gd[3], Dm, Dm);
    add(G + w.df1, F, L + r.h1, Dm, Gd[4], Gd[5], F, Dm);
    add(G + w.ds2, Dm, L + r.f1, F, Gd[6], Gd[7], Dm, F);
  }
  if (xbf) {
    LR_TRY_(flush(0));
    add(wsb + w.dhA, Dm, x, I, grads[0], grads[1], Dm, I);
    LR_TRY_(flush(1));
  } else {
    if (n + 1 > 20) LR_TRY_(flush(0));
    add(wsb + w.dhA, Dm, x, I, grads[0], grads[1], Dm, I);
    LR_TRY_(flush(0));
  }
  LnJobs lj;
  for (int l = 0; l < nlayers; ++l) {
    float* const* Gd = grads + 2 + 12 * l;
    float* G = wsb + w.layer0 + (size_t)l * w.per_layer;
    lj.partial[2 * l] = G + w.lnp1; lj.dgamma[2 * l] = Gd[8]; lj.dbeta[2 * l] = Gd[9];
    lj.partial[2 * l + 1] = G + w.lnp2; lj.dgamma[2 * l + 1] = Gd[10]; lj.dbeta[2 * l + 1] = Gd[11];
  }
  LR_LAUNCH(ln_param_reduce_kernel, dim3((2 * Dm + 63) / 64, 2 * nlayers), dim3(256), 0, st, lj, Dm, accumulate);
  return lr_launch_status();
}

extern "C" int lr_tfm_rowblock_supported(int B, int T, int Dm, int F, int nlayers) {
  return B > 0 && T > 0 && lr_tfm_rb_supported(Dm, F, nlayers);
}
