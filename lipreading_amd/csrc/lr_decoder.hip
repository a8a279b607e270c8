// lr_decoder.hip — the attention character decoder loop on gfx950.
//
// Reference arithmetic replaced here (paths under the reference root):
//   src/models/lipreader/better_model.py:161-231  CharDecodingStep.forward: embedding -> 1-step RNN
//       -> attention over the encoder states (none | dot | general | 1_layer_nn | concat)
//       -> masked_softmax -> context -> concat_layer + tanh -> output_proj -> masked_log_softmax
//   src/train/train_better_model.py:54-65 (train) / :121-135 (eval): the loop over max_label_len
//       steps with teacher forcing or the previous step's multinomial sample as input.
//
// Structure.  The loop is a strictly sequential chain of small ops at batch 32; the whole of it
// (all L steps, forward or backward) is enqueued by ONE C call, so the host pays one ctypes call
// per pass and the chain is hipGraph-capturable when the teacher-forcing pattern is fixed.
//   * embedding + input projection collapse into a V x (G*Hd) table EW = E W_ih^T + b, one GEMM
//     per pass (V = 64 tokens); a step's gate pre-activations are a row gather by token id;
//   * the recurrent cell is the encoder's fused step kernel (lr_rnn.hip: packed W_hh, MFMA
//     16x16x4 f32, one memory round trip), started from the encoder's final state;
//   * attention runs one workgroup per sample: logits (lanes along the hidden axis, one wave per
//     encoder frame), allennlp masked_softmax, context; the parts of each attention type that do
//     not depend on the step (W_g applied to the encoder states, w_e . enc, W1e enc + b1) are
//     computed once per pass as GEMMs;
//   * concat_layer is an M = B GEMM (split-K); tanh, output_proj (V = 64), masked log-softmax and
//     the multinomial draw are one workgroup per sample.
// Backward walks the chain in reverse with per-step kernels for everything on the dependency
// chain and defers every weight gradient to one batched GEMM over all (step, sample) rows.
//
// The multinomial draw uses a counter-based hash (seed, step, sample); it matches torch's sampler
// in distribution only (the reference's draws are RNG-dependent as well).
#include "lr_common.h"

namespace {

enum { ATT_NONE = 0, ATT_DOT = 1, ATT_GENERAL = 2, ATT_1LNN = 3, ATT_CONCAT = 4 };

__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = lr_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += scratch[w];
  return s;
}

// gates[b][i][:] = EW[id][:], id = teacher-forced token or the previous step's sample
__global__ void dec_gather_kernel(const float* __restrict__ EW, const int32_t* __restrict__ tokens,
                                  const int32_t* __restrict__ sampled, int32_t* __restrict__ ids_used,
                                  float* __restrict__ gates, int L, int GH, int V, int i, int teacher) {
  const int b = blockIdx.x;
  int id = (teacher || i == 0) ? tokens[(int64_t)b * L + i] : sampled[(int64_t)b * L + i - 1];
  if (id < 0 || id >= V) id = 0;
  if (threadIdx.x == 0) ids_used[(int64_t)b * L + i] = id;
  const float* src = EW + (int64_t)id * GH;
  float* dst = gates + ((int64_t)b * L + i) * GH;
  for (int c = threadIdx.x; c < GH; c += blockDim.x) dst[c] = src[c];
}

struct AttnAux {
  const float* src;     // dot: enc; general: GE = enc W_g            [B][T][Hd]
  const float* cterm;   // general: cE = enc . b_g ; 1_layer_nn: se = enc . w_e   [B][T]
  const float* wvec;    // 1_layer_nn: w_h [Hd] ; concat: W1 (row a: [2Hd], the h part starts at Hd)
  const float* w2;      // concat: w2 [A]
  const float* PE;      // concat: W1e enc + b1   [B][T][A]
  float bias;           // 1_layer_nn: b ; concat: b2   (read on the host? no: passed as pointers)
  const float* bias_p;  // pointer to that scalar
};

// One workgroup per sample: attention of step i.
//   logits_out [B][T] (raw, before masking), cat_out [B][2Hd] = (context | h), ph_out [B][A] (concat)
__global__ __launch_bounds__(256) void dec_attn_fwd_kernel(int type, const float* __restrict__ hs,
                                                           const float* __restrict__ enc,
                                                           const int32_t* __restrict__ enc_lens, AttnAux a,
                                                           float* __restrict__ logits_out,
                                                           float* __restrict__ cat_out,
                                                           float* __restrict__ ph_out, int L, int T, int Hd,
                                                           int A, int i) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* h = reinterpret_cast<float*>(smem_raw);   // [Hd]
  float* lg = h + Hd;                              // [T]
  float* ph = lg + T;                              // [A]
  __shared__ float scratch[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const float* hrow = hs + ((int64_t)b * L + i) * Hd;
  for (int k = tid; k < Hd; k += blockDim.x) h[k] = hrow[k];
  __syncthreads();
  const float* encb = enc + (int64_t)b * T * Hd;
  if (type == ATT_DOT || type == ATT_GENERAL) {
    const float* srcb = a.src + (int64_t)b * T * Hd;
    for (int t = wave; t < T; t += nw) {
      float s = 0.f;
      for (int k = lane; k < Hd; k += 64) s += h[k] * srcb[(int64_t)t * Hd + k];
      s = lr_wave_sum(s);
      if (lane == 0) lg[t] = s + (type == ATT_GENERAL ? a.cterm[(int64_t)b * T + t] : 0.f);
    }
  } else if (type == ATT_1LNN) {
    float s = 0.f;
    for (int k = tid; k < Hd; k += blockDim.x) s += h[k] * a.wvec[k];
    const float sh = block_sum(s, scratch) + a.bias_p[0];
    for (int t = tid; t < T; t += blockDim.x) lg[t] = a.cterm[(int64_t)b * T + t] + sh;
  } else {  // ATT_CONCAT
    for (int r = wave; r < A; r += nw) {            // ph[r] = W1h[r] . h
      const float* wrow = a.wvec + (int64_t)r * 2 * Hd + Hd;
      float s = 0.f;
      for (int k = lane; k < Hd; k += 64) s += wrow[k] * h[k];
      s = lr_wave_sum(s);
      if (lane == 0) { ph[r] = s; ph_out[(int64_t)b * A + r] = s; }
    }
    __syncthreads();
    const float* peb = a.PE + (int64_t)b * T * A;
    for (int t = wave; t < T; t += nw) {
      float s = 0.f;
      for (int r = lane; r < A; r += 64) s += a.w2[r] * tanhf(peb[(int64_t)t * A + r] + ph[r]);
      s = lr_wave_sum(s);
      if (lane == 0) lg[t] = s + a.bias_p[0];
    }
  }
  __syncthreads();
  // allennlp masked_softmax: softmax(logits * mask) * mask / (sum + 1e-13)
  const int len = enc_lens[b];
  float mx = LR_NEG_INF;
  for (int t = tid; t < T; t += blockDim.x) {
    logits_out[(int64_t)b * T + t] = lg[t];
    const float x = t < len ? lg[t] : 0.f;
    mx = fmaxf(mx, x);
  }
  mx = lr_wave_max(mx);
  __syncthreads();
  if (lane == 0) scratch[wave] = mx;
  __syncthreads();
  mx = scratch[0];
  for (int w = 1; w < nw; ++w) mx = fmaxf(mx, scratch[w]);
  float se = 0.f, sv = 0.f;
  for (int t = tid; t < T; t += blockDim.x) {
    const float e = expf((t < len ? lg[t] : 0.f) - mx);
    se += e;
    if (t < len) sv += e;
  }
  const float Z = block_sum(se, scratch);
  const float S = block_sum(sv, scratch) / Z;       // sum of the masked probabilities
  __syncthreads();
  for (int t = tid; t < T; t += blockDim.x)
    lg[t] = t < len ? (expf(lg[t] - mx) / Z) / (S + 1e-13f) : 0.f;   // attention weights
  __syncthreads();
  float* cat = cat_out + (int64_t)b * 2 * Hd;
  for (int k = tid; k < Hd; k += blockDim.x) {
    float c = 0.f;
    for (int t = 0; t < len && t < T; ++t) c += lg[t] * encb[(int64_t)t * Hd + k];
    cat[k] = c;
    cat[Hd + k] = h[k];
  }
}

__device__ __forceinline__ float hash_uniform(uint64_t seed, uint32_t step, uint32_t sample) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)step * 0x100000001B3ull + sample + 1);
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (float)(x >> 40) * (1.f / 16777216.f);   // 24 random bits -> [0,1)
}

// One workgroup per sample: new_h = tanh(pre) (stored back), logits = W_o new_h + b_o,
// masked log-softmax, multinomial draw.  With has_attn == 0 the input is h itself (no tanh).
__global__ __launch_bounds__(256) void dec_out_fwd_kernel(float* __restrict__ nh, const float* __restrict__ w_o,
                                                          const float* __restrict__ b_o,
                                                          const float* __restrict__ mask,
                                                          float* __restrict__ log_probs,
                                                          int32_t* __restrict__ sampled, int L, int Hd, int V,
                                                          int i, int has_attn, uint64_t seed) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* x = reinterpret_cast<float*>(smem_raw);   // [Hd]
  float* lg = x + Hd;                              // [V]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  float* row = nh + (int64_t)b * Hd;
  for (int k = tid; k < Hd; k += blockDim.x) {
    float v = row[k];
    if (has_attn) { v = tanhf(v); row[k] = v; }
    x[k] = v;
  }
  __syncthreads();
  for (int v = wave; v < V; v += nw) {
    const float* wrow = w_o + (int64_t)v * Hd;
    float s = 0.f;
    for (int k = lane; k < Hd; k += 64) s += wrow[k] * x[k];
    s = lr_wave_sum(s);
    if (lane == 0) lg[v] = s + b_o[v] + logf(mask[v] + 1e-45f);
  }
  __syncthreads();
  if (wave == 0) {
    float m = LR_NEG_INF;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, lg[v]);
    m = lr_wave_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(lg[v] - m);
    s = lr_wave_sum(s);
    const float lse = m + logf(s);
    float* out = log_probs + ((int64_t)b * L + i) * V;
    for (int v = lane; v < V; v += 64) { lg[v] -= lse; out[v] = lg[v]; }
  }
  __syncthreads();
  if (tid == 0) {   // multinomial(1) over exp(log_probs)
    const float u = hash_uniform(seed, (uint32_t)i, (uint32_t)b);
    float total = 0.f;
    for (int v = 0; v < V; ++v) total += expf(lg[v]);
    float cum = 0.f;
    int pick = V - 1;
    for (int v = 0; v < V; ++v) {
      cum += expf(lg[v]);
      if (u * total < cum) { pick = v; break; }
    }
    sampled[(int64_t)b * L + i] = pick;
  }
}

// backward of dec_out_fwd: dlogits = g - exp(lp) sum(g); d new_h = dlogits W_o; dpre = d new_h (1 - new_h^2)
__global__ __launch_bounds__(256) void dec_out_bwd_kernel(const float* __restrict__ g_lp,
                                                          const float* __restrict__ log_probs,
                                                          const float* __restrict__ nh,
                                                          const float* __restrict__ w_o,
                                                          float* __restrict__ dlogits, float* __restrict__ dpre,
                                                          int L, int Hd, int V, int i, int has_attn) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* dl = reinterpret_cast<float*>(smem_raw);   // [V]
  __shared__ float scratch[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* g = g_lp + ((int64_t)b * L + i) * V;
  const float* lp = log_probs + ((int64_t)b * L + i) * V;
  float s = 0.f;
  for (int v = tid; v < V; v += blockDim.x) s += g[v];
  const float gs = block_sum(s, scratch);
  for (int v = tid; v < V; v += blockDim.x) {
    const float d = g[v] - expf(lp[v]) * gs;
    dl[v] = d;
    dlogits[(int64_t)b * V + v] = d;
  }
  __syncthreads();
  for (int k = tid; k < Hd; k += blockDim.x) {
    float acc = 0.f;
    for (int v = 0; v < V; ++v) acc += dl[v] * w_o[(int64_t)v * Hd + k];
    if (has_attn) {
      const float y = nh[(int64_t)b * Hd + k];
      acc *= 1.f - y * y;
    }
    dpre[(int64_t)b * Hd + k] = acc;
  }
}

struct AttnGrad {
  float* d_enc;   // [B][T][Hd]  accumulated over steps
  float* d_src;   // general: dGE [B][T][Hd]
  float* d_cterm; // general: dcE [B][T]; 1_layer_nn: dse [B][T]
  float* d_sh;    // 1_layer_nn: [B] for this step (dsh)
  float* d_ph;    // concat: [B][A] for this step
  float* d_PE;    // concat: [B][T][A] accumulated
  float* d_w2;    // concat: [B][A] per-sample partial of dw2 for this step
  float* d_b2;    // concat / 1_layer_nn bias: [B] per-sample partial (sum_t dlogit)
};

// One workgroup per sample: backward of the attention of step i.
//   in: dcat [B][2Hd] (d context | d h through concat_layer); out: dy [B][L][Hd] row i = total external dh
__global__ __launch_bounds__(256) void dec_attn_bwd_kernel(int type, const float* __restrict__ hs,
                                                           const float* __restrict__ enc,
                                                           const int32_t* __restrict__ enc_lens, AttnAux a,
                                                           const float* __restrict__ logits_in,
                                                           const float* __restrict__ ph_in,
                                                           const float* __restrict__ dcat, AttnGrad gr,
                                                           float* __restrict__ dy, int L, int T, int Hd, int A,
                                                           int i) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* h = reinterpret_cast<float*>(smem_raw);   // [Hd]
  float* dctx = h + Hd;                            // [Hd]
  float* wt = dctx + Hd;                           // [T] attention weights
  float* pt = wt + T;                              // [T] full softmax p
  float* dlg = pt + T;                             // [T] d logits
  float* ph = dlg + T;                             // [A]
  float* dph = ph + A;                             // [A]
  __shared__ float scratch[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int len = min(enc_lens[b], T);
  const float* hrow = hs + ((int64_t)b * L + i) * Hd;
  const float* dc = dcat + (int64_t)b * 2 * Hd;
  for (int k = tid; k < Hd; k += blockDim.x) { h[k] = hrow[k]; dctx[k] = dc[k]; }
  if (type == ATT_CONCAT)
    for (int r = tid; r < A; r += blockDim.x) { ph[r] = ph_in[(int64_t)b * A + r]; dph[r] = 0.f; }
  // recompute p (softmax over all T of logits*mask), S and the weights
  const float* lgin = logits_in + (int64_t)b * T;
  float mx = LR_NEG_INF;
  for (int t = tid; t < T; t += blockDim.x) mx = fmaxf(mx, t < len ? lgin[t] : 0.f);
  mx = lr_wave_max(mx);
  __syncthreads();
  if (lane == 0) scratch[wave] = mx;
  __syncthreads();
  mx = scratch[0];
  for (int w = 1; w < nw; ++w) mx = fmaxf(mx, scratch[w]);
  float se = 0.f, sv = 0.f;
  for (int t = tid; t < T; t += blockDim.x) {
    const float e = expf((t < len ? lgin[t] : 0.f) - mx);
    pt[t] = e;
    se += e;
    if (t < len) sv += e;
  }
  const float Z = block_sum(se, scratch);
  const float S = block_sum(sv, scratch) / Z;
  __syncthreads();
  for (int t = tid; t < T; t += blockDim.x) {
    pt[t] /= Z;
    wt[t] = t < len ? pt[t] / (S + 1e-13f) : 0.f;
  }
  __syncthreads();
  // d weights: dw[t] = dctx . enc[b,t];   d_enc[b,t] += w[t] * dctx
  const float* encb = enc + (int64_t)b * T * Hd;
  float* dencb = gr.d_enc + (int64_t)b * T * Hd;
  for (int t = wave; t < T; t += nw) {
    float s = 0.f;
    if (t < len) {
      for (int k = lane; k < Hd; k += 64) {
        s += dctx[k] * encb[(int64_t)t * Hd + k];
        dencb[(int64_t)t * Hd + k] += wt[t] * dctx[k];
      }
    }
    s = lr_wave_sum(s);
    if (lane == 0) dlg[t] = s;   // holds dw[t] for now
  }
  __syncthreads();
  // masked_softmax backward
  float s1 = 0.f;
  for (int t = tid; t < T; t += blockDim.x) s1 += dlg[t] * wt[t];
  const float dot_w = block_sum(s1, scratch);
  __syncthreads();
  float s2 = 0.f;
  for (int t = tid; t < T; t += blockDim.x) {
    const float dp = t < len ? (dlg[t] - dot_w) / (S + 1e-13f) : 0.f;   // d p[t] (through r = p*m)
    dlg[t] = dp;
    s2 += dp * pt[t];
  }
  const float dot_p = block_sum(s2, scratch);
  __syncthreads();
  float s3 = 0.f;
  for (int t = tid; t < T; t += blockDim.x) {
    const float dx = pt[t] * (dlg[t] - dot_p);
    const float d = t < len ? dx : 0.f;     // x = logits * mask
    dlg[t] = d;
    s3 += d;
  }
  const float dsum = block_sum(s3, scratch);   // sum_t dlogit[t]
  __syncthreads();
  // back through the logits
  float* dyrow = dy + ((int64_t)b * L + i) * Hd;
  if (type == ATT_DOT || type == ATT_GENERAL) {
    const float* srcb = a.src + (int64_t)b * T * Hd;
    float* dsrcb = (type == ATT_GENERAL ? gr.d_src : gr.d_enc) + (int64_t)b * T * Hd;
    for (int k = tid; k < Hd; k += blockDim.x) {
      float acc = 0.f;
      for (int t = 0; t < len; ++t) {
        acc += dlg[t] * srcb[(int64_t)t * Hd + k];
        dsrcb[(int64_t)t * Hd + k] += dlg[t] * h[k];
      }
      dyrow[k] = dc[Hd + k] + acc;
    }
    if (type == ATT_GENERAL)
      for (int t = tid; t < len; t += blockDim.x) gr.d_cterm[(int64_t)b * T + t] += dlg[t];
  } else if (type == ATT_1LNN) {
    for (int t = tid; t < len; t += blockDim.x) gr.d_cterm[(int64_t)b * T + t] += dlg[t];
    if (tid == 0) gr.d_sh[b] = dsum;   // also the bias gradient of this (step, sample)
    for (int k = tid; k < Hd; k += blockDim.x) dyrow[k] = dc[Hd + k] + dsum * a.wvec[k];
  } else {  // ATT_CONCAT
    const float* peb = a.PE + (int64_t)b * T * A;
    float* dpeb = gr.d_PE + (int64_t)b * T * A;
    for (int r = tid; r < A; r += blockDim.x) {
      float accp = 0.f, accw = 0.f;
      const float w2r = a.w2[r];
      for (int t = 0; t < len; ++t) {
        const float u = tanhf(peb[(int64_t)t * A + r] + ph[r]);
        const float du = dlg[t] * w2r * (1.f - u * u);
        dpeb[(int64_t)t * A + r] += du;
        accp += du;
        accw += dlg[t] * u;
      }
      dph[r] = accp;
      gr.d_ph[(int64_t)b * A + r] = accp;
      gr.d_w2[(int64_t)b * A + r] = accw;
    }
    if (tid == 0) gr.d_b2[b] = dsum;
    __syncthreads();
    for (int k = tid; k < Hd; k += blockDim.x) {   // dh += W1h^T dph
      float acc = 0.f;
      for (int r = 0; r < A; ++r) acc += a.wvec[(int64_t)r * 2 * Hd + Hd + k] * dph[r];
      dyrow[k] = dc[Hd + k] + acc;
    }
  }
}

// dEW[v][:] = sum over (b,i) with ids_used == v of dG[b][i][slot(c)][j]   (fixed order: deterministic)
__global__ void dec_scatter_dew_kernel(const float* __restrict__ dG, const int32_t* __restrict__ ids,
                                       float* __restrict__ dEW, int rows, int G, int Hd) {
  const int v = blockIdx.x;
  const int GH = G * Hd;
  for (int c = threadIdx.x; c < GH; c += blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r)
      if (ids[r] == v) s += dG[(int64_t)r * 4 * Hd + c];   // slots 0..G-1 are the first G*Hd columns
    dEW[(int64_t)v * GH + c] = s;
  }
}

__global__ void zero_row_kernel(float* __restrict__ x, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = 0.f;
}

// out[i] (+)= sum_r x[r*ld + i] for i < n   (small: one block)
__global__ void rowsum_acc_kernel(const float* __restrict__ x, int rows, int ld, int n, float* __restrict__ out,
                                  int accumulate) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += x[(int64_t)r * ld + i];
    out[i] = accumulate ? out[i] + s : s;
  }
}

struct Sizes {
  int B, L, T, Hd, Cd, V, A, G, type;
};

// ---- reserve (forward -> backward) layout, in floats ------------------------------------------------
struct Res {
  size_t EW, biasf, gates, extra, hs, hp, wp, ids, logits, cat, pre, aux1, aux2, ph, gemm, total;
  size_t hp_slot, gemm_bytes;
};
Res res_layout(const Sizes& z) {
  Res r;
  const size_t GH = (size_t)z.G * z.Hd, BL = (size_t)z.B * z.L, BT = (size_t)z.B * z.T;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += (n + 63) / 64 * 64; return at; };
  r.EW = take((size_t)z.V * GH);
  r.biasf = take(GH);
  r.gates = take(BL * GH);
  r.extra = take(BL * z.Hd);
  r.hs = take(BL * z.Hd);
  r.hp_slot = lr_rnn_packed_state_floats(z.B, z.Hd);
  r.hp = take(2 * r.hp_slot);
  r.wp = take(lr_rnn_packed_w_floats(z.G, z.Hd));
  r.ids = take(BL);
  r.logits = take((size_t)z.L * z.B * z.T);
  r.cat = take(BL * 2 * z.Hd);
  r.pre = take(BL * z.Hd);
  r.aux1 = take(z.type == ATT_GENERAL ? BT * z.Hd : (z.type == ATT_CONCAT ? BT * z.A : 0));   // GE | PE
  r.aux2 = take((z.type == ATT_GENERAL || z.type == ATT_1LNN) ? BT : 0);                       // cE | se
  r.ph = take(z.type == ATT_CONCAT ? BL * z.A : 0);
  size_t gb = lr_sgemm_workspace_bytes(z.B, z.Hd, 2 * z.Hd);
  size_t g2 = lr_sgemm_workspace_bytes(z.V, (int)GH, z.Cd);
  if (g2 > gb) gb = g2;
  g2 = lr_sgemm_workspace_bytes((int)BT, z.Hd, z.Hd);
  if (g2 > gb) gb = g2;
  r.gemm_bytes = gb;
  r.gemm = take((gb + 3) / 4);
  r.total = o;
  return r;
}

struct Wsp {
  size_t wpT, dG, dcar, dgp, dy, dlogits, dpre, dcat, dEW, dsrc, dcterm, dsh, dPE, dph, dw2, db2, colsum, gemm, total;
  size_t dgp_slot, gemm_bytes;
};
Wsp ws_layout(const Sizes& z) {
  Wsp w;
  const size_t GH = (size_t)z.G * z.Hd, BL = (size_t)z.B * z.L, BT = (size_t)z.B * z.T;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += (n + 63) / 64 * 64; return at; };
  w.wpT = take(lr_rnn_packed_w_floats(z.G, z.Hd));
  w.dG = take(BL * 4 * z.Hd);
  w.dcar = take(BL * z.Hd);
  w.dgp_slot = (size_t)((z.B + 15) / 16) * z.G * ((z.Hd + 15) / 16) * 256;
  w.dgp = take(2 * w.dgp_slot);
  w.dy = take(BL * z.Hd);
  w.dlogits = take(BL * z.V);
  w.dpre = take(BL * z.Hd);
  w.dcat = take((size_t)z.B * 2 * z.Hd);
  w.dEW = take((size_t)z.V * GH);
  w.dsrc = take(z.type == ATT_GENERAL ? BT * z.Hd : 0);
  w.dcterm = take((z.type == ATT_GENERAL || z.type == ATT_1LNN) ? BT : 0);
  w.dsh = take(z.type == ATT_1LNN ? BL : 0);
  w.dPE = take(z.type == ATT_CONCAT ? BT * z.A : 0);
  w.dph = take(z.type == ATT_CONCAT ? BL * z.A : 0);
  w.dw2 = take(z.type == ATT_CONCAT ? BL * z.A : 0);
  w.db2 = take((z.type == ATT_CONCAT || z.type == ATT_1LNN) ? BL : 0);
  w.colsum = take((size_t)LR_COLSUM_SPLITS * 4 * z.Hd);
  size_t gb = 0;
  const int dims[][3] = {{z.B, 2 * z.Hd, z.Hd}, {z.Hd, 2 * z.Hd, (int)BL}, {z.V, z.Hd, (int)BL},
                         {(int)GH, z.Hd, (int)BL}, {(int)GH, z.Cd, z.V}, {z.V, z.Cd, (int)GH},
                         {(int)BT, z.Hd, z.Hd}, {z.Hd, z.Hd, (int)BT}, {z.A > 0 ? z.A : 1, z.Hd, (int)BT},
                         {(int)BT, z.Hd, z.A > 0 ? z.A : 1}, {z.A > 0 ? z.A : 1, z.Hd, (int)BL}};
  for (auto& d : dims) {
    const size_t g = lr_sgemm_workspace_bytes(d[0], d[1], d[2]);
    if (g > gb) gb = g;
  }
  w.gemm_bytes = gb;
  w.gemm = take((gb + 3) / 4);
  w.total = o;
  return w;
}

bool sizes_ok(int mode, int type, int B, int L, int T, int Hd, int Cd, int V, int A) {
  return (mode == LR_RNN_GRU || mode == LR_RNN_LSTM) && type >= ATT_NONE && type <= ATT_CONCAT && B > 0 &&
         L > 0 && T > 0 && Hd > 0 && Hd % 4 == 0 && Cd > 0 && V > 0 && V <= 1024 &&
         (type != ATT_CONCAT || A > 0);
}

#define LR_TRY(expr)              \
  do {                            \
    const int st__ = (expr);      \
    if (st__ != LR_OK) return st__; \
  } while (0)

}  // namespace

extern "C" size_t lr_decoder_reserve_bytes(int mode, int attn_type, int B, int L, int T, int Hd, int Cd, int V,
                                           int A) {
  if (!sizes_ok(mode, attn_type, B, L, T, Hd, Cd, V, A)) return 0;
  const Sizes z = {B, L, T, Hd, Cd, V, A, mode == LR_RNN_GRU ? 3 : 4, attn_type};
  return res_layout(z).total * sizeof(float);
}

extern "C" size_t lr_decoder_workspace_bytes(int mode, int attn_type, int B, int L, int T, int Hd, int Cd, int V,
                                             int A) {
  if (!sizes_ok(mode, attn_type, B, L, T, Hd, Cd, V, A)) return 0;
  const Sizes z = {B, L, T, Hd, Cd, V, A, mode == LR_RNN_GRU ? 3 : 4, attn_type};
  return ws_layout(z).total * sizeof(float);
}

extern "C" int lr_decoder_forward(int mode, int attn_type, const lr_decoder_params* p, const int32_t* tokens,
                                  const uint8_t* teacher_forced_host, const float* enc, const int32_t* enc_lens,
                                  const float* h0, const float* c0, const int32_t* step_lens, uint64_t seed,
                                  float* log_probs, int32_t* sampled, float* h_n, float* c_n, void* reserve,
                                  size_t reserve_bytes, int B, int L, int T, int Hd, int Cd, int V, int A,
                                  lr_stream_t stream_) {
  LR_CHECK_ARG(sizes_ok(mode, attn_type, B, L, T, Hd, Cd, V, A));
  LR_CHECK_ARG(p && tokens && teacher_forced_host && enc && enc_lens && h0 && step_lens && log_probs && sampled &&
               reserve);
  LR_CHECK_ARG(p->emb && p->w_ih && p->w_hh && p->b_ih && p->b_hh && p->w_o && p->b_o && p->out_mask);
  LR_CHECK_ARG(attn_type == ATT_NONE || (p->w_c && p->b_c));
  LR_CHECK_ARG(mode == LR_RNN_GRU || c0);
  const Sizes z = {B, L, T, Hd, Cd, V, A, mode == LR_RNN_GRU ? 3 : 4, attn_type};
  const Res r = res_layout(z);
  if (reserve_bytes < r.total * sizeof(float)) return LR_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* base = (float*)reserve;
  const int G = z.G, GH = G * Hd;
  float* EW = base + r.EW;
  float* gates = base + r.gates;
  float* hs = base + r.hs;
  float* hp = base + r.hp;
  void* gws = base + r.gemm;
  int32_t* ids = (int32_t*)(base + r.ids);

  // table of input projections: EW = emb @ W_ih^T + folded bias
  LR_TRY(lr_rnn_fold_bias(p->b_ih, p->b_hh, base + r.biasf, G, Hd, stream));
  LR_TRY(lr_sgemm_impl(0, 1, V, GH, Cd, 1.f, p->emb, Cd, p->w_ih, Cd, 0.f, EW, GH, base + r.biasf, 0, 0, gws,
                       r.gemm_bytes, stream));
  LR_TRY(lr_rnn_pack_w(p->w_hh, base + r.wp, G, Hd, 0, stream));
  lr_clear_error();
  if (hipMemsetAsync(hp, 0, 2 * r.hp_slot * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;
  LR_TRY(lr_rnn_pack_state(h0, hp + r.hp_slot, B, Hd, stream));   // step 0 reads parity (0+1)&1 = 1

  AttnAux aux = {};
  const int R = B * T;
  if (attn_type == ATT_DOT) {
    aux.src = enc;
  } else if (attn_type == ATT_GENERAL) {
    LR_CHECK_ARG(p->attn_w1 && p->attn_b1);
    // GE = enc @ W_g (so that (W_g h + b_g) . enc = h . GE + enc . b_g)
    LR_TRY(lr_sgemm_impl(0, 0, R, Hd, Hd, 1.f, enc, Hd, p->attn_w1, Hd, 0.f, base + r.aux1, Hd, nullptr, 0, 0, gws,
                         r.gemm_bytes, stream));
    LR_TRY(lr_sgemm_impl(0, 1, R, 1, Hd, 1.f, enc, Hd, p->attn_b1, Hd, 0.f, base + r.aux2, 1, nullptr, 0, 0,
                         nullptr, 0, stream));
    aux.src = base + r.aux1;
    aux.cterm = base + r.aux2;
  } else if (attn_type == ATT_1LNN) {
    LR_CHECK_ARG(p->attn_w1 && p->attn_b1);
    LR_TRY(lr_sgemm_impl(0, 1, R, 1, Hd, 1.f, enc, Hd, p->attn_w1, 2 * Hd, 0.f, base + r.aux2, 1, nullptr, 0, 0,
                         nullptr, 0, stream));   // se = enc . w_e  (w_e = first Hd entries)
    aux.cterm = base + r.aux2;
    aux.wvec = p->attn_w1 + Hd;                  // w_h
    aux.bias_p = p->attn_b1;
  } else if (attn_type == ATT_CONCAT) {
    LR_CHECK_ARG(p->attn_w1 && p->attn_b1 && p->attn_w2 && p->attn_b2);
    LR_TRY(lr_sgemm_impl(0, 1, R, A, Hd, 1.f, enc, Hd, p->attn_w1, 2 * Hd, 0.f, base + r.aux1, A, p->attn_b1, 0, 0,
                         gws, r.gemm_bytes, stream));   // PE = enc @ W1e^T + b1
    aux.PE = base + r.aux1;
    aux.wvec = p->attn_w1;
    aux.w2 = p->attn_w2;
    aux.bias_p = p->attn_b2;
  }
  const size_t attn_lds = ((size_t)Hd + T + (A > 0 ? A : 0) + 8) * sizeof(float);
  const size_t out_lds = ((size_t)Hd + V + 8) * sizeof(float);
  if (attn_lds > 60 * 1024 || out_lds > 60 * 1024) return LR_ERR_UNSUPPORTED;

  for (int i = 0; i < L; ++i) {
    LR_LAUNCH(dec_gather_kernel, dim3(B), dim3(256), 0, stream, (const float*)EW, tokens, (const int32_t*)sampled,
              ids, gates, L, GH, V, i, (int)teacher_forced_host[i]);
    LR_TRY(lr_launch_status());
    LR_TRY(lr_rnn_step_fwd(G, gates, base + r.extra, hs, hp, step_lens, base + r.wp, p->b_hh, h0, c0, B, L, Hd, i,
                           stream));
    float* nh = base + r.pre + (size_t)i * B * Hd;
    if (attn_type != ATT_NONE) {
      float* cat = base + r.cat + (size_t)i * B * 2 * Hd;
      LR_LAUNCH(dec_attn_fwd_kernel, dim3(B), dim3(256), attn_lds, stream, attn_type, (const float*)hs, enc,
                enc_lens, aux, base + r.logits + (size_t)i * B * T, cat,
                attn_type == ATT_CONCAT ? base + r.ph + (size_t)i * B * A : (float*)nullptr, L, T, Hd, A, i);
      LR_TRY(lr_launch_status());
      LR_TRY(lr_sgemm_impl(0, 1, B, Hd, 2 * Hd, 1.f, cat, 2 * Hd, p->w_c, 2 * Hd, 0.f, nh, Hd, p->b_c, 0, 0, gws,
                           r.gemm_bytes, stream));
    } else {
      // no attention: output_proj acts on the RNN state itself (better_model.py:228)
      lr_clear_error();
      if (hipMemcpy2DAsync(nh, (size_t)Hd * sizeof(float), hs + (size_t)i * Hd, (size_t)L * Hd * sizeof(float),
                           (size_t)Hd * sizeof(float), B, hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return LR_ERR_LAUNCH;
    }
    LR_LAUNCH(dec_out_fwd_kernel, dim3(B), dim3(256), out_lds, stream, nh, p->w_o, p->b_o, p->out_mask, log_probs,
              sampled, L, Hd, V, i, attn_type != ATT_NONE ? 1 : 0, seed);
    LR_TRY(lr_launch_status());
  }
  // state after the last step (what the reference's step returns as final_state, better_model.py:181)
  lr_clear_error();
  if (h_n && hipMemcpy2DAsync(h_n, (size_t)Hd * sizeof(float), hs + (size_t)(L - 1) * Hd,
                              (size_t)L * Hd * sizeof(float), (size_t)Hd * sizeof(float), B,
                              hipMemcpyDeviceToDevice, stream) != hipSuccess)
    return LR_ERR_LAUNCH;
  if (c_n && G == 4 && hipMemcpy2DAsync(c_n, (size_t)Hd * sizeof(float), base + r.extra + (size_t)(L - 1) * Hd,
                                        (size_t)L * Hd * sizeof(float), (size_t)Hd * sizeof(float), B,
                                        hipMemcpyDeviceToDevice, stream) != hipSuccess)
    return LR_ERR_LAUNCH;
  return LR_OK;
}

extern "C" int lr_decoder_backward(int mode, int attn_type, const lr_decoder_params* p, const lr_decoder_grads* g,
                                   const float* enc, const int32_t* enc_lens, const float* h0, const float* c0,
                                   const int32_t* step_lens, const float* log_probs, const float* d_log_probs,
                                   const float* dh_n, const float* dc_n, float* d_enc, float* dh0, float* dc0,
                                   const void* reserve, size_t reserve_bytes,
                                   void* workspace, size_t workspace_bytes, int accumulate, int B, int L, int T,
                                   int Hd, int Cd, int V, int A, lr_stream_t stream_) {
  LR_CHECK_ARG(sizes_ok(mode, attn_type, B, L, T, Hd, Cd, V, A));
  LR_CHECK_ARG(p && g && enc && enc_lens && h0 && step_lens && log_probs && d_log_probs && d_enc && dh0 &&
               reserve && workspace);
  LR_CHECK_ARG(g->emb && g->w_ih && g->w_hh && g->b_ih && g->b_hh && g->w_o && g->b_o);
  LR_CHECK_ARG(mode == LR_RNN_GRU || (c0 && dc0));
  const Sizes z = {B, L, T, Hd, Cd, V, A, mode == LR_RNN_GRU ? 3 : 4, attn_type};
  const Res r = res_layout(z);
  const Wsp w = ws_layout(z);
  if (reserve_bytes < r.total * sizeof(float)) return LR_ERR_WORKSPACE;
  if (workspace_bytes < w.total * sizeof(float)) return LR_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const float* rb = (const float*)reserve;
  float* wb = (float*)workspace;
  const int G = z.G, GH = G * Hd, R = B * T, BL = B * L;
  const float beta = accumulate ? 1.f : 0.f;
  void* gws = wb + w.gemm;
  const float* hs = rb + r.hs;
  const int32_t* ids = (const int32_t*)(rb + r.ids);
  float* dG = wb + w.dG;
  float* dy = wb + w.dy;

  LR_TRY(lr_rnn_pack_w(p->w_hh, wb + w.wpT, G, Hd, 1, stream));
  lr_clear_error();
  if (hipMemsetAsync(wb + w.dgp, 0, 2 * w.dgp_slot * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;
  if (hipMemsetAsync(d_enc, 0, (size_t)R * Hd * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;
  AttnAux aux = {};
  AttnGrad gr = {};
  gr.d_enc = d_enc;
  if (attn_type == ATT_DOT) {
    aux.src = enc;
  } else if (attn_type == ATT_GENERAL) {
    aux.src = rb + r.aux1;
    aux.cterm = rb + r.aux2;
    gr.d_src = wb + w.dsrc;
    gr.d_cterm = wb + w.dcterm;
    if (hipMemsetAsync(gr.d_src, 0, (size_t)R * Hd * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;
    if (hipMemsetAsync(gr.d_cterm, 0, (size_t)R * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;
  } else if (attn_type == ATT_1LNN) {
    aux.cterm = rb + r.aux2;
    aux.wvec = p->attn_w1 + Hd;
    aux.bias_p = p->attn_b1;
    gr.d_cterm = wb + w.dcterm;
    if (hipMemsetAsync(gr.d_cterm, 0, (size_t)R * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;
  } else if (attn_type == ATT_CONCAT) {
    aux.PE = rb + r.aux1;
    aux.wvec = p->attn_w1;
    aux.w2 = p->attn_w2;
    aux.bias_p = p->attn_b2;
    gr.d_PE = wb + w.dPE;
    if (hipMemsetAsync(gr.d_PE, 0, (size_t)R * A * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;
  }
  const size_t attn_lds = ((size_t)2 * Hd + 3 * T + 2 * (A > 0 ? A : 0) + 8) * sizeof(float);
  const size_t out_lds = ((size_t)V + 8) * sizeof(float);
  if (attn_lds > 60 * 1024) return LR_ERR_UNSUPPORTED;

  for (int i = L - 1; i >= 0; --i) {
    const float* nh = rb + r.pre + (size_t)i * B * Hd;
    float* dpre = wb + w.dpre + (size_t)i * B * Hd;
    LR_LAUNCH(dec_out_bwd_kernel, dim3(B), dim3(256), out_lds, stream, d_log_probs, log_probs, nh, p->w_o,
              wb + w.dlogits + (size_t)i * B * V, dpre, L, Hd, V, i, attn_type != ATT_NONE ? 1 : 0);
    LR_TRY(lr_launch_status());
    if (attn_type != ATT_NONE) {
      // d(cat) = dpre @ W_c
      LR_TRY(lr_sgemm_impl(0, 0, B, 2 * Hd, Hd, 1.f, dpre, Hd, p->w_c, 2 * Hd, 0.f, wb + w.dcat, 2 * Hd, nullptr, 0,
                           0, gws, w.gemm_bytes, stream));
      if (attn_type == ATT_1LNN) gr.d_sh = wb + w.dsh + (size_t)i * B;
      if (attn_type == ATT_CONCAT) {
        gr.d_ph = wb + w.dph + (size_t)i * B * A;
        gr.d_w2 = wb + w.dw2 + (size_t)i * B * A;
        gr.d_b2 = wb + w.db2 + (size_t)i * B;
      }
      LR_LAUNCH(dec_attn_bwd_kernel, dim3(B), dim3(256), attn_lds, stream, attn_type, hs, enc, enc_lens, aux,
                rb + r.logits + (size_t)i * B * T,
                attn_type == ATT_CONCAT ? rb + r.ph + (size_t)i * B * A : (const float*)nullptr,
                (const float*)(wb + w.dcat), gr, dy, L, T, Hd, A, i);
      LR_TRY(lr_launch_status());
    } else {
      lr_clear_error();
      if (hipMemcpy2DAsync(dy + (size_t)i * Hd, (size_t)L * Hd * sizeof(float), dpre, (size_t)Hd * sizeof(float),
                           (size_t)Hd * sizeof(float), B, hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return LR_ERR_LAUNCH;
    }
    LR_TRY(lr_rnn_step_bwd(G, rb + r.gates, rb + r.extra, hs, dy, dh_n, dc_n, dG, wb + w.dcar, wb + w.dgp,
                           step_lens, wb + w.wpT, h0, c0, B, L, Hd, L - 1 - i, stream));
  }
  // gradient into the initial state (the encoder's final state)
  LR_TRY(lr_rnn_dh0(G, wb + w.dcar, wb + w.dgp + (size_t)((L - 1) & 1) * w.dgp_slot, wb + w.wpT, dh0, dc0, B, L, Hd,
                    stream));

  // ---- deferred weight gradients: one GEMM each over all (sample, step) rows ----------------------
  const int ldg = 4 * Hd;
  // W_hh: h_prev of step t is hs[b][t-1]; step 0 used h0
  if (G == 3) {
    LR_TRY(lr_sgemm_impl(1, 0, 2 * Hd, Hd, BL, 1.f, dG, ldg, hs, Hd, beta, g->w_hh, Hd, nullptr, -1, L, gws,
                         w.gemm_bytes, stream));
    LR_TRY(lr_sgemm_impl(1, 0, Hd, Hd, BL, 1.f, dG + 3 * Hd, ldg, hs, Hd, beta, g->w_hh + (size_t)2 * Hd * Hd, Hd,
                         nullptr, -1, L, gws, w.gemm_bytes, stream));
    LR_TRY(lr_sgemm_impl(1, 0, 2 * Hd, Hd, B, 1.f, dG, L * ldg, h0, Hd, 1.f, g->w_hh, Hd, nullptr, 0, 0, nullptr, 0,
                         stream));
    LR_TRY(lr_sgemm_impl(1, 0, Hd, Hd, B, 1.f, dG + 3 * Hd, L * ldg, h0, Hd, 1.f, g->w_hh + (size_t)2 * Hd * Hd, Hd,
                         nullptr, 0, 0, nullptr, 0, stream));
  } else {
    LR_TRY(lr_sgemm_impl(1, 0, GH, Hd, BL, 1.f, dG, ldg, hs, Hd, beta, g->w_hh, Hd, nullptr, -1, L, gws,
                         w.gemm_bytes, stream));
    LR_TRY(lr_sgemm_impl(1, 0, GH, Hd, B, 1.f, dG, L * ldg, h0, Hd, 1.f, g->w_hh, Hd, nullptr, 0, 0, nullptr, 0,
                         stream));
  }
  LR_TRY(lr_rnn_bias_grads(dG, wb + w.colsum, g->b_ih, g->b_hh, BL, Hd, G, accumulate, stream));
  // embedding / W_ih through the table: dEW[v] = sum of the dG_x rows that used token v
  LR_LAUNCH(dec_scatter_dew_kernel, dim3(V), dim3(256), 0, stream, (const float*)dG, ids, wb + w.dEW, BL, G, Hd);
  LR_TRY(lr_launch_status());
  if (g->emb_padding_idx >= 0 && g->emb_padding_idx < V) {
    // nn.Embedding(padding_idx): that row receives no gradient (its W_ih contribution is x = 0 anyway)
    LR_LAUNCH(zero_row_kernel, dim3(1), dim3(256), 0, stream, wb + w.dEW + (size_t)g->emb_padding_idx * GH, GH);
    LR_TRY(lr_launch_status());
  }
  LR_TRY(lr_sgemm_impl(1, 0, GH, Cd, V, 1.f, wb + w.dEW, GH, p->emb, Cd, beta, g->w_ih, Cd, nullptr, 0, 0, gws,
                       w.gemm_bytes, stream));
  LR_TRY(lr_sgemm_impl(0, 0, V, Cd, GH, 1.f, wb + w.dEW, GH, p->w_ih, Cd, beta, g->emb, Cd, nullptr, 0, 0, gws,
                       w.gemm_bytes, stream));

  // output projection and concat layer
  LR_TRY(lr_sgemm_impl(1, 0, V, Hd, BL, 1.f, wb + w.dlogits, V, rb + r.pre, Hd, beta, g->w_o, Hd, nullptr, 0, 0, gws,
                       w.gemm_bytes, stream));
  LR_LAUNCH(rowsum_acc_kernel, dim3(1), dim3(256), 0, stream, (const float*)(wb + w.dlogits), BL, V, V, g->b_o,
            accumulate);
  LR_TRY(lr_launch_status());
  if (attn_type != ATT_NONE) {
    LR_CHECK_ARG(g->w_c && g->b_c);
    LR_TRY(lr_sgemm_impl(1, 0, Hd, 2 * Hd, BL, 1.f, wb + w.dpre, Hd, rb + r.cat, 2 * Hd, beta, g->w_c, 2 * Hd,
                         nullptr, 0, 0, gws, w.gemm_bytes, stream));
    LR_LAUNCH(rowsum_acc_kernel, dim3((Hd + 255) / 256), dim3(256), 0, stream, (const float*)(wb + w.dpre), BL, Hd,
              Hd, g->b_c, accumulate);
    LR_TRY(lr_launch_status());
  }
  // attention parameters and the step-independent parts of d_enc
  if (attn_type == ATT_GENERAL) {
    LR_CHECK_ARG(g->attn_w1 && g->attn_b1);
    // GE = enc @ W_g: dW_g = enc^T @ dGE, d_enc += dGE @ W_g^T; cE = enc . b_g: db_g = enc^T dcE, d_enc += dcE b_g^T
    LR_TRY(lr_sgemm_impl(1, 0, Hd, Hd, R, 1.f, enc, Hd, wb + w.dsrc, Hd, beta, g->attn_w1, Hd, nullptr, 0, 0, gws,
                         w.gemm_bytes, stream));
    LR_TRY(lr_sgemm_impl(0, 1, R, Hd, Hd, 1.f, wb + w.dsrc, Hd, p->attn_w1, Hd, 1.f, d_enc, Hd, nullptr, 0, 0, gws,
                         w.gemm_bytes, stream));
    LR_TRY(lr_sgemm_impl(1, 0, Hd, 1, R, 1.f, enc, Hd, wb + w.dcterm, 1, beta, g->attn_b1, 1, nullptr, 0, 0, nullptr,
                         0, stream));
    LR_TRY(lr_sgemm_impl(0, 0, R, Hd, 1, 1.f, wb + w.dcterm, 1, p->attn_b1, Hd, 1.f, d_enc, Hd, nullptr, 0, 0, nullptr,
                         0, stream));
  } else if (attn_type == ATT_1LNN) {
    LR_CHECK_ARG(g->attn_w1 && g->attn_b1);
    // w = [w_e | w_h]: dw_e = enc^T dse, dw_h = sum_{i,b} dsh[i][b] hs[b][i], db = sum dsh; d_enc += dse w_e^T
    LR_TRY(lr_sgemm_impl(1, 0, Hd, 1, R, 1.f, enc, Hd, wb + w.dcterm, 1, beta, g->attn_w1, 1, nullptr, 0, 0, nullptr,
                         0, stream));
    LR_TRY(lr_sgemm_impl(0, 1, R, Hd, 1, 1.f, wb + w.dcterm, 1, p->attn_w1, 1, 1.f, d_enc, Hd, nullptr, 0, 0, nullptr,
                         0, stream));
    // dsh is stored [L][B]; hs is [B][L][Hd]: per step a (Hd x B)·(B x 1) product, accumulated
    for (int i = 0; i < L; ++i)
      LR_TRY(lr_sgemm_impl(1, 0, Hd, 1, B, 1.f, hs + (size_t)i * Hd, L * Hd, wb + w.dsh + (size_t)i * B, 1,
                           (i == 0 ? beta : 1.f), g->attn_w1 + Hd, 1, nullptr, 0, 0, nullptr, 0, stream));
    LR_LAUNCH(rowsum_acc_kernel, dim3(1), dim3(64), 0, stream, (const float*)(wb + w.dsh), BL, 1, 1, g->attn_b1,
              accumulate);
    LR_TRY(lr_launch_status());
  } else if (attn_type == ATT_CONCAT) {
    LR_CHECK_ARG(g->attn_w1 && g->attn_b1 && g->attn_w2 && g->attn_b2);
    // W1 = [W1e | W1h] (A x 2Hd): dW1e = dPE^T enc, dW1h = sum_i dph_i^T hs_i, db1 = colsum(dPE)
    LR_TRY(lr_sgemm_impl(1, 0, A, Hd, R, 1.f, wb + w.dPE, A, enc, Hd, beta, g->attn_w1, 2 * Hd, nullptr, 0, 0, gws,
                         w.gemm_bytes, stream));
    for (int i = 0; i < L; ++i)
      LR_TRY(lr_sgemm_impl(1, 0, A, Hd, B, 1.f, wb + w.dph + (size_t)i * B * A, A, hs + (size_t)i * Hd, L * Hd,
                           (i == 0 ? beta : 1.f), g->attn_w1 + Hd, 2 * Hd, nullptr, 0, 0, nullptr, 0, stream));
    LR_TRY(lr_sgemm_impl(0, 0, R, Hd, A, 1.f, wb + w.dPE, A, p->attn_w1, 2 * Hd, 1.f, d_enc, Hd, nullptr, 0, 0, gws,
                         w.gemm_bytes, stream));
    LR_LAUNCH(rowsum_acc_kernel, dim3(1), dim3(256), 0, stream, (const float*)(wb + w.dPE), R, A, A, g->attn_b1,
              accumulate);
    LR_LAUNCH(rowsum_acc_kernel, dim3(1), dim3(256), 0, stream, (const float*)(wb + w.dw2), BL, A, A, g->attn_w2,
              accumulate);
    LR_LAUNCH(rowsum_acc_kernel, dim3(1), dim3(64), 0, stream, (const float*)(wb + w.db2), BL, 1, 1, g->attn_b2,
              accumulate);
    LR_TRY(lr_launch_status());
  }
  return LR_OK;
}
