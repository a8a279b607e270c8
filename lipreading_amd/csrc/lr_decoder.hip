// lr_decoder.hip — the attention character decoder loop on gfx950.
//
// Reference arithmetic replaced here (paths under the reference root):
//   src/models/lipreader/better_model.py:161-231  CharDecodingStep.forward: embedding -> 1-step RNN
//       -> attention over the encoder states (none | dot | general | 1_layer_nn | concat)
//       -> masked_softmax -> context -> concat_layer + tanh -> output_proj -> masked_log_softmax
//   src/train/train_better_model.py:54-65 (train) / :121-135 (eval): the loop over max_label_len
//       steps with teacher forcing or the previous step's multinomial sample as input.
//
// Structure.  In the reference's step only the RNN state is carried from step to step: the
// attention, concat_layer and output_proj of step i read h_i and feed nothing back (the next
// input is a token id).  So the recurrence that has to run step by step is just the RNN cell —
// the encoder's fused step kernel (lr_rnn.hip: packed W_hh, MFMA 16x16x4 f32, one memory round
// trip per step) started from the encoder's final state — and everything else is batched over
// all (sample, step) rows, row = b*L + i:
//   * embedding + input projection collapse into a V x (G*Hd) table EW = E W_ih^T + b (one GEMM);
//     a step's gate pre-activations are a row gather by token id;
//   * attention logits / context are per-sample GEMMs (L x Hd)(Hd x T) and (L x T)(T x Hd) on the
//     matrix cores (lr_sgemm_batched_impl); the step-independent halves of each attention type
//     (W_g applied to the encoder states, w_e . enc, W1e enc + b1) are computed once per pass;
//   * concat_layer is two (B*L) x Hd x Hd GEMMs (context half, state half: no concatenated copy);
//   * tanh, output_proj (V = 64), masked log-softmax and the multinomial draw are one kernel,
//     8 rows per workgroup so W_o is read once per 8 rows.
// A step that is NOT teacher forced needs the previous step's sample: the batched head is then
// flushed for the steps finished so far before that step's gather (teacher_forcing_ratio = 1, the
// shipped configs and eval, is a single flush).  Backward has no such dependency at all: the head,
// the attention and every weight gradient are batched; only the RNN backward walks the steps.
//
// The multinomial draw uses a counter-based hash (seed, step, sample); it matches torch's sampler
// in distribution only (the reference's draws are RNG-dependent as well).
#include "lr_common.h"

namespace {

enum { ATT_NONE = 0, ATT_DOT = 1, ATT_GENERAL = 2, ATT_1LNN = 3, ATT_CONCAT = 4 };
constexpr int OUT_ROWS = 8;   // rows per workgroup of dec_out_fwd_kernel

// gates[b][i][:] = EW[id][:] for steps i0 + blockIdx.x; id = teacher-forced token or the previous
// step's sample (step 0 always reads tokens: the reference feeds BOS)
__global__ __launch_bounds__(256) void dec_gather_kernel(const float* __restrict__ EW,
                                                         const int32_t* __restrict__ tokens,
                                                         const int32_t* __restrict__ sampled,
                                                         int32_t* __restrict__ ids_used,
                                                         float* __restrict__ gates, int L, int GH, int V, int i0,
                                                         int teacher) {
  const int i = i0 + blockIdx.x, b = blockIdx.y;
  int id = (teacher || i == 0) ? tokens[(int64_t)b * L + i] : sampled[(int64_t)b * L + i - 1];
  if (id < 0 || id >= V) id = 0;
  if (threadIdx.x == 0) ids_used[(int64_t)b * L + i] = id;
  const float4* src = reinterpret_cast<const float4*>(EW + (int64_t)id * GH);
  float4* dst = reinterpret_cast<float4*>(gates + ((int64_t)b * L + i) * GH);
  for (int c = threadIdx.x; c < GH / 4; c += blockDim.x) dst[c] = src[c];
}

// concat attention: logits[b][i][t] = w2 . tanh(PE[b][t][:] + ph[b][i][:]) + b2.  grid (steps, B)
__global__ __launch_bounds__(256) void dec_concat_logits_kernel(const float* __restrict__ PE,
                                                                const float* __restrict__ ph,
                                                                const float* __restrict__ w2,
                                                                const float* __restrict__ b2,
                                                                float* __restrict__ logits, int L, int T, int A,
                                                                int i0) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* phs = reinterpret_cast<float*>(smem_raw);   // [A]
  float* w2s = phs + A;                              // [A]
  const int i = i0 + blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = (int64_t)b * L + i;
  for (int r = tid; r < A; r += 256) { phs[r] = ph[row * A + r]; w2s[r] = w2[r]; }
  __syncthreads();
  const float* peb = PE + (int64_t)b * T * A;
  for (int t = wave; t < T; t += 4) {
    float s = 0.f;
    for (int r = lane; r < A; r += 64) s += w2s[r] * tanhf(peb[(int64_t)t * A + r] + phs[r]);
    s = lr_wave_sum(s);
    if (lane == 0) logits[row * T + t] = s + b2[0];
  }
}

// One wave per (sample, step) row: finish the raw logits (1_layer_nn: se[b][t] + w_h . h + b;
// general: + cE[b][t]), then allennlp masked_softmax: softmax(logits * mask) * mask / (sum + 1e-13).
// grid (steps, B), 64 threads.
__global__ __launch_bounds__(64) void dec_attn_softmax_kernel(int type, const float* __restrict__ hs,
                                                              const int32_t* __restrict__ enc_lens,
                                                              const float* __restrict__ cterm,
                                                              const float* __restrict__ wvec,
                                                              const float* __restrict__ bias_p,
                                                              float* __restrict__ logits,
                                                              float* __restrict__ wts, int L, int T, int Hd,
                                                              int i0) {
  const int i = i0 + blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int64_t row = (int64_t)b * L + i;
  float* lg = logits + row * T;
  if (type == ATT_1LNN) {
    const float* h = hs + row * Hd;
    float s = 0.f;
    for (int k = lane; k < Hd; k += 64) s += h[k] * wvec[k];
    const float sh = lr_wave_sum(s) + bias_p[0];
    for (int t = lane; t < T; t += 64) lg[t] = cterm[(int64_t)b * T + t] + sh;
  } else if (type == ATT_GENERAL) {
    for (int t = lane; t < T; t += 64) lg[t] += cterm[(int64_t)b * T + t];
  }
  // (each lane re-reads only the entries it wrote)
  const int len = min(enc_lens[b], T);
  float mx = LR_NEG_INF;
  for (int t = lane; t < T; t += 64) mx = fmaxf(mx, t < len ? lg[t] : 0.f);
  mx = lr_wave_max(mx);
  float se = 0.f, sv = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float e = expf((t < len ? lg[t] : 0.f) - mx);
    se += e;
    if (t < len) sv += e;
  }
  const float Z = lr_wave_sum(se);
  const float S = lr_wave_sum(sv) / Z;   // sum of the masked probabilities
  float* w = wts + row * T;
  for (int t = lane; t < T; t += 64) w[t] = t < len ? (expf(lg[t] - mx) / Z) / (S + 1e-13f) : 0.f;
}

__device__ __forceinline__ float hash_uniform(uint64_t seed, uint32_t step, uint32_t sample) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)step * 0x100000001B3ull + sample + 1);
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (float)(x >> 40) * (1.f / 16777216.f);   // 24 random bits -> [0,1)
}

// `nrows` (<= OUT_ROWS) rows per workgroup: new_h = tanh(pre) (stored back), logits = W_o new_h +
// b_o + log(mask + 1e-45), log-softmax, multinomial draw.  With has_attn == 0 the input rows are
// the RNN states themselves (no tanh).  Rows of the segment [i0, i0 + n): q -> (b = q / n, i = i0 + q % n).
__global__ __launch_bounds__(256) void dec_out_fwd_kernel(float* __restrict__ nh, const float* __restrict__ w_o,
                                                          const float* __restrict__ b_o,
                                                          const float* __restrict__ mask,
                                                          float* __restrict__ log_probs,
                                                          int32_t* __restrict__ sampled, int L, int Hd, int V,
                                                          int i0, int n, int total, int nrows, int has_attn,
                                                          uint64_t seed) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* x = reinterpret_cast<float*>(smem_raw);   // [nrows][Hd]
  float* lg = x + (size_t)nrows * Hd;              // [nrows][V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.x * nrows;
  const int H4 = Hd >> 2;
  for (int idx = tid; idx < nrows * H4; idx += 256) {
    const int r = idx / H4, k4 = idx - r * H4;
    const int q = q0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < total) {
      const int64_t row = (int64_t)(q / n) * L + i0 + q % n;
      float4* src = reinterpret_cast<float4*>(nh + row * Hd) + k4;
      v = *src;
      if (has_attn) {
        v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
        *src = v;
      }
    }
    *reinterpret_cast<float4*>(&x[(size_t)r * Hd + 4 * k4]) = v;
  }
  __syncthreads();
  // 4 lanes per output class, each a quarter of the k range in 16-byte pieces; all loads of a
  // class row are independent, so they are in flight together
  const int q4 = tid & 3;
  for (int vc = 0; vc < V; vc += 64) {
    const int v = vc + (tid >> 2);
    float acc[OUT_ROWS];
#pragma unroll
    for (int r = 0; r < OUT_ROWS; ++r) acc[r] = 0.f;
    if (v < V) {
      const float* wrow = w_o + (int64_t)v * Hd;
      for (int k = q4 * 4; k < Hd; k += 16) {
        const float4 w4 = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
        for (int r = 0; r < OUT_ROWS; ++r) {
          if (r < nrows) {
            const float4 x4 = *reinterpret_cast<const float4*>(&x[(size_t)r * Hd + k]);
            acc[r] += w4.x * x4.x + w4.y * x4.y + w4.z * x4.z + w4.w * x4.w;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < OUT_ROWS; ++r) {
      acc[r] += __shfl_xor(acc[r], 1, 64);
      acc[r] += __shfl_xor(acc[r], 2, 64);
    }
    if (q4 == 0 && v < V) {
      const float add = b_o[v] + logf(mask[v] + 1e-45f);
#pragma unroll
      for (int r = 0; r < OUT_ROWS; ++r)
        if (r < nrows) lg[(size_t)r * V + v] = acc[r] + add;
    }
  }
  __syncthreads();
  for (int r = wave; r < nrows; r += 4) {
    const int q = q0 + r;
    if (q >= total) continue;   // wave-uniform
    const int b = q / n, i = i0 + q % n;
    const int64_t row = (int64_t)b * L + i;
    float* l = lg + (size_t)r * V;
    float m = LR_NEG_INF;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, l[v]);
    m = lr_wave_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(l[v] - m);
    s = lr_wave_sum(s);
    const float lse = m + logf(s);
    float* out = log_probs + row * V;
    float tot = 0.f;
    for (int v = lane; v < V; v += 64) {
      const float lp = l[v] - lse;
      l[v] = lp;
      out[v] = lp;
      tot += expf(lp);
    }
    tot = lr_wave_sum(tot);
    // multinomial(1) over exp(log_probs): first class whose cumulative mass exceeds u * total
    const float u = hash_uniform(seed, (uint32_t)i, (uint32_t)b);
    float running = 0.f;
    int pick = V - 1;
    bool found = false;
    for (int vc = 0; vc < V; vc += 64) {
      const int v = vc + lane;
      float c = v < V ? expf(l[v]) : 0.f;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const float t = __shfl_up(c, d, 64);
        if (lane >= d) c += t;
      }
      const unsigned long long hit = __ballot(v < V && u * tot < running + c);
      if (!found && hit) {
        pick = vc + __ffsll((long long)hit) - 1;
        found = true;
      }
      running += __shfl(c, 63, 64);
    }
    if (lane == 0) sampled[row] = pick;
  }
}

// one wave per row: dlogits = g - exp(lp) * sum(g)   (backward of log_softmax)
__global__ __launch_bounds__(256) void dec_out_bwd_rows_kernel(const float* __restrict__ g_lp,
                                                               const float* __restrict__ log_probs,
                                                               float* __restrict__ dlogits, int rows, int V) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* g = g_lp + (int64_t)row * V;
  const float* lp = log_probs + (int64_t)row * V;
  float s = 0.f;
  for (int v = lane; v < V; v += 64) s += g[v];
  s = lr_wave_sum(s);
  for (int v = lane; v < V; v += 64) dlogits[(int64_t)row * V + v] = g[v] - expf(lp[v]) * s;
}

// d(pre) = d(new_h) * (1 - new_h^2), in place
__global__ void dec_tanh_bwd_kernel(float* __restrict__ d, const float* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<float4*>(d)[i];
    const float4 t = reinterpret_cast<const float4*>(y)[i];
    a.x *= 1.f - t.x * t.x; a.y *= 1.f - t.y * t.y; a.z *= 1.f - t.z * t.z; a.w *= 1.f - t.w * t.w;
    reinterpret_cast<float4*>(d)[i] = a;
  }
}

// One wave per (sample, step) row: backward of masked_softmax.  in: dlg = d weights (d ctx . enc[t]);
// out: dlg = d logits (0 past the sample's length), dsum[row] = sum_t d logits.
__global__ __launch_bounds__(256) void dec_attn_softmax_bwd_kernel(const float* __restrict__ logits,
                                                                   const float* __restrict__ wts,
                                                                   const int32_t* __restrict__ enc_lens,
                                                                   float* __restrict__ dlg,
                                                                   float* __restrict__ dsum, int rows, int L,
                                                                   int T) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int len = min(enc_lens[row / L], T);
  const float* lg = logits + (int64_t)row * T;
  const float* w = wts + (int64_t)row * T;
  float* d = dlg + (int64_t)row * T;
  float mx = LR_NEG_INF;
  for (int t = lane; t < T; t += 64) mx = fmaxf(mx, t < len ? lg[t] : 0.f);
  mx = lr_wave_max(mx);
  float se = 0.f, sv = 0.f, s1 = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float e = expf((t < len ? lg[t] : 0.f) - mx);
    se += e;
    if (t < len) { sv += e; s1 += d[t] * w[t]; }
  }
  const float Z = lr_wave_sum(se);
  const float S = lr_wave_sum(sv) / Z;
  const float dot_w = lr_wave_sum(s1);
  float s2 = 0.f;
  for (int t = lane; t < len; t += 64) {
    const float p = expf(lg[t] - mx) / Z;
    s2 += (d[t] - dot_w) / (S + 1e-13f) * p;   // d p[t] (through r = p * mask), times p[t]
  }
  const float dot_p = lr_wave_sum(s2);
  float s3 = 0.f;
  for (int t = lane; t < T; t += 64) {
    float dx = 0.f;
    if (t < len) {
      const float p = expf(lg[t] - mx) / Z;
      dx = p * ((d[t] - dot_w) / (S + 1e-13f) - dot_p);   // x = logits * mask
    }
    d[t] = dx;
    s3 += dx;
  }
  s3 = lr_wave_sum(s3);
  if (lane == 0) dsum[row] = s3;
}

// out[b][t] = sum_i x[b][i][t]   (step-independent logit terms: cE of 'general', se of '1_layer_nn')
__global__ void dec_sum_steps_kernel(const float* __restrict__ x, float* __restrict__ out, int L, int T) {
  const int b = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    float s = 0.f;
    for (int i = 0; i < L; ++i) s += x[((int64_t)b * L + i) * T + t];
    out[(int64_t)b * T + t] = s;
  }
}

// concat attention backward through u = tanh(PE[b][t][r] + ph[b][i][r]), logit = w2 . u + b2:
//   dph[b][i][r] = sum_t du, dPE[b][t][r] = sum_i du, dw2p[b][r] = sum_{i,t} dlogit u,
//   du = dlogit[b][i][t] w2[r] (1 - u^2).  grid (ceil(A/64), B), one lane per r; tanh is recomputed
//   in a second sweep rather than holding L (or T) accumulators per lane.
__global__ __launch_bounds__(64) void dec_concat_bwd_kernel(const float* __restrict__ PE,
                                                            const float* __restrict__ ph,
                                                            const float* __restrict__ w2,
                                                            const float* __restrict__ dlg,
                                                            const int32_t* __restrict__ enc_lens,
                                                            float* __restrict__ dPE, float* __restrict__ dph,
                                                            float* __restrict__ dw2p, int L, int T, int A) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* dl = reinterpret_cast<float*>(smem_raw);   // [L][T]
  const int b = blockIdx.y, r = blockIdx.x * 64 + threadIdx.x;
  const float* src = dlg + (int64_t)b * L * T;
  for (int e = threadIdx.x; e < L * T; e += 64) dl[e] = src[e];
  __syncthreads();
  if (r >= A) return;
  const int len = min(enc_lens[b], T);
  const float w2r = w2[r];
  const float* peb = PE + (int64_t)b * T * A + r;
  const float* phb = ph + (int64_t)b * L * A + r;
  float accw = 0.f;
  for (int i = 0; i < L; ++i) {
    const float phv = phb[(int64_t)i * A];
    float acc = 0.f;
    for (int t = 0; t < len; ++t) {
      const float u = tanhf(peb[(int64_t)t * A] + phv);
      const float g = dl[i * T + t];
      acc += g * w2r * (1.f - u * u);
      accw += g * u;
    }
    dph[((int64_t)b * L + i) * A + r] = acc;
  }
  for (int t = 0; t < T; ++t) {
    float acc = 0.f;
    if (t < len) {
      const float pe = peb[(int64_t)t * A];
      for (int i = 0; i < L; ++i) {
        const float u = tanhf(pe + phb[(int64_t)i * A]);
        acc += dl[i * T + t] * w2r * (1.f - u * u);
      }
    }
    dPE[((int64_t)b * T + t) * A + r] = acc;
  }
  dw2p[(int64_t)b * A + r] = accw;
}

// dEW[v][c] = sum over the rows (b,i) that used token v of dG[row][c], rows in ascending order
// (deterministic).  grid (V, ceil(G*Hd / 256)); the matching rows of each 256-row chunk are compacted
// in order into LDS, then every lane adds its column over that list.
__global__ __launch_bounds__(256) void dec_scatter_dew_kernel(const float* __restrict__ dG,
                                                              const int32_t* __restrict__ ids,
                                                              float* __restrict__ dEW, int rows, int G, int Hd) {
  __shared__ int list[256];
  __shared__ int wcount[4];
  const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int GH = G * Hd;
  const int c = blockIdx.y * 256 + tid;
  float s = 0.f;
  for (int r0 = 0; r0 < rows; r0 += 256) {
    const int r = r0 + tid;
    const bool hit = r < rows && ids[r] == v;
    const unsigned long long m = __ballot(hit);
    __syncthreads();   // previous chunk's list fully consumed
    if (lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wcount[w];
    if (hit) list[base + __popcll(m & ((1ull << lane) - 1ull))] = r;
    __syncthreads();
    const int cnt = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    if (c < GH)
      for (int j = 0; j < cnt; ++j) s += dG[(int64_t)list[j] * 4 * Hd + c];   // slots 0..G-1 = first G*Hd columns
  }
  if (c < GH) dEW[(int64_t)v * GH + c] = s;
}

__global__ void zero_row_kernel(float* __restrict__ x, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = 0.f;
}

// out[c] (+)= fixed-order sum of the LR_COLSUM_SPLITS partial column sums (lr_colsum_partial)
__global__ void dec_colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int C,
                                        int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < LR_COLSUM_SPLITS; ++r) s += partial[(int64_t)r * C + c];
  out[c] = accumulate ? out[c] + s : s;
}

struct Sizes {
  int B, L, T, Hd, Cd, V, A, G, type, NL;
};
constexpr int MAXL = LR_DEC_MAX_LAYERS;
inline int gates_of_mode(int mode) { return mode == LR_RNN_GRU ? 3 : (mode == LR_RNN_LSTM ? 4 : 1); }

size_t max_gemm_ws(const int (*dims)[3], int n) {
  size_t gb = 0;
  for (int i = 0; i < n; ++i) {
    const size_t g = lr_sgemm_workspace_bytes(dims[i][0], dims[i][1], dims[i][2]);
    if (g > gb) gb = g;
  }
  return gb;
}

// ---- reserve (forward -> backward) layout, in floats; per-(sample, step) buffers are [B][L][...] ------
// Per layer k of the decoder's RNN stack: gate buffer, extra (GRU W_hn h + b_hn / LSTM c), output
// states hs, two packed-state parity slots, packed W_hh, folded bias.  The head and the attention read
// the TOP layer's states: hs = hsl[NL-1].
struct Res {
  size_t EW, ids, logits, wts, ctx, pre, aux1, aux2, ph, gemm, total;
  size_t biasf[MAXL], gates[MAXL], extra[MAXL], hsl[MAXL], hp[MAXL], wp[MAXL], xm[MAXL];
  size_t hs;
  size_t hp_slot, gemm_bytes;
  size_t xch, xch_bytes;   // exchange words of the one-launch recurrence (lr_rnn_cluster.hip), when it covers (G, B, Hd)
};
// the packed recurrent weights of a layer: the step kernels' fragment order, or the cluster recurrence's
size_t packed_w_floats(const Sizes& z, int backward) {
  size_t n = lr_rnn_packed_w_floats(z.G, z.Hd);
  if (z.G != 1 && lr_rnn_cluster_supported(z.G, z.B, z.Hd)) {
    const size_t c = (lr_rnn_cluster_pack_bytes(z.G, z.Hd, 1, backward) + 3) / 4;
    if (c > n) n = c;
  }
  return n;
}
Res res_layout(const Sizes& z) {
  Res r;
  const size_t GH = (size_t)z.G * z.Hd, BL = (size_t)z.B * z.L, BT = (size_t)z.B * z.T;
  const bool attn = z.type != ATT_NONE;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += (n + 63) / 64 * 64; return at; };
  r.EW = take((size_t)z.V * GH);
  r.hp_slot = lr_rnn_packed_state_floats(z.B, z.Hd);
  for (int k = 0; k < MAXL; ++k) r.biasf[k] = r.gates[k] = r.extra[k] = r.hsl[k] = r.hp[k] = r.wp[k] = r.xm[k] = 0;
  for (int k = 0; k < z.NL; ++k) {
    r.biasf[k] = take(GH);
    r.gates[k] = take(BL * GH);
    r.extra[k] = take(BL * z.Hd);
    r.hsl[k] = take(BL * z.Hd);
    r.hp[k] = take(2 * r.hp_slot);
    r.wp[k] = take(packed_w_floats(z, 0));
    r.xm[k] = take(k + 1 < z.NL ? BL * z.Hd : 0);   // dropout-masked copy of hsl[k] (input of layer k+1)
  }
  r.hs = r.hsl[z.NL - 1];
  r.ids = take(BL);
  r.logits = take(attn ? BL * z.T : 0);    // raw attention logits
  r.wts = take(attn ? BL * z.T : 0);       // attention weights
  r.ctx = take(attn ? BL * z.Hd : 0);      // context vectors
  r.pre = take(attn ? BL * z.Hd : 0);      // concat_layer output, overwritten by its tanh
  r.aux1 = take(z.type == ATT_GENERAL ? BT * z.Hd : (z.type == ATT_CONCAT ? BT * z.A : 0));   // GE | PE
  r.aux2 = take((z.type == ATT_GENERAL || z.type == ATT_1LNN) ? BT : 0);                       // cE | se
  r.ph = take(z.type == ATT_CONCAT ? BL * z.A : 0);
  const int a = z.A > 0 ? z.A : 1;
  const int dims[][3] = {{z.V, (int)GH, z.Cd}, {(int)BT, z.Hd, z.Hd}, {(int)BT, a, z.Hd}, {(int)BL, z.Hd, z.Hd},
                         {(int)BL, a, z.Hd}, {(int)BL, (int)GH, z.Hd}};
  r.gemm_bytes = max_gemm_ws(dims, 6);
  r.gemm = take((r.gemm_bytes + 3) / 4);
  r.xch_bytes = (z.G != 1 && lr_rnn_cluster_supported(z.G, z.B, z.Hd)) ? lr_rnn_cluster_xch_bytes(z.B, z.Hd, 1, 0) : 0;
  r.xch = take((r.xch_bytes + 3) / 4);
  r.total = o;
  return r;
}

// A WIDE decoder's recurrent weight gradients (G * Hd >= 1536: the 1024-unit decoder of config/train/attn/attention_type,
// the 1400 / 1536-unit ones of config/defaults.txt-style and ecd flag files, better_model.py:134-148) are split-bf16
// products straight from dG and the states (lr_fgemm.hip, TN form, ~1e-5 relative — the encoder's rule for its own
// wide layers, lr_rnn.hip wgrad_split) instead of the fp32-MFMA grouped GEMM: at LSTM-1536, B = 32 that GEMM was
// 295 us of a 2.68 ms step at 40 % of the fp32 matrix peak.
inline bool wide_wgrad(int G, int Hd) { return G * Hd >= 1536 && !lr_debug_wgrad_f32(); }

struct Wsp {
  size_t wpT, dG, dcar, dgp, dy, dlogits, dpre, dctx, dlg, dsum, dEW, dsrc, dcterm, dPE, dph, dw2p, colsum, gemm,
      hprev, total;
  size_t dgp_slot, gemm_bytes;
  size_t xch, xch_bytes;
};
Wsp ws_layout(const Sizes& z) {
  Wsp w;
  const size_t GH = (size_t)z.G * z.Hd, BL = (size_t)z.B * z.L, BT = (size_t)z.B * z.T;
  const bool attn = z.type != ATT_NONE;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += (n + 63) / 64 * 64; return at; };
  w.wpT = take(packed_w_floats(z, 1));
  w.xch_bytes = (z.G != 1 && lr_rnn_cluster_supported(z.G, z.B, z.Hd)) ? lr_rnn_cluster_xch_bytes(z.B, z.Hd, 1, 1) : 0;
  w.xch = take((w.xch_bytes + 3) / 4);
  w.dG = take(BL * 4 * z.Hd);
  w.dcar = take(BL * z.Hd);
  w.dgp_slot = (size_t)((z.B + 15) / 16) * z.G * ((z.Hd + 15) / 16) * 256;
  w.dgp = take(2 * w.dgp_slot);
  w.dy = take(BL * z.Hd);
  w.dlogits = take(BL * z.V);
  w.dpre = take(attn ? BL * z.Hd : 0);
  w.dctx = take(attn ? BL * z.Hd : 0);
  w.dlg = take(attn ? BL * z.T : 0);
  w.dsum = take(attn ? BL : 0);
  w.dEW = take((size_t)z.V * GH);
  w.dsrc = take(z.type == ATT_GENERAL ? BT * z.Hd : 0);
  w.dcterm = take((z.type == ATT_GENERAL || z.type == ATT_1LNN) ? BT : 0);
  w.dPE = take(z.type == ATT_CONCAT ? BT * z.A : 0);
  w.dph = take(z.type == ATT_CONCAT ? BL * z.A : 0);
  w.dw2p = take(z.type == ATT_CONCAT ? (size_t)z.B * z.A : 0);
  size_t widest = (size_t)4 * z.Hd;
  if ((size_t)z.V > widest) widest = z.V;
  if ((size_t)z.A > widest) widest = z.A;
  w.colsum = take((size_t)LR_COLSUM_SPLITS * widest);
  const int a = z.A > 0 ? z.A : 1;
  const int dims[][3] = {{(int)BL, z.Hd, z.V}, {(int)BL, z.Hd, z.Hd}, {z.Hd, z.Hd, (int)BL}, {z.V, z.Hd, (int)BL},
                         {(int)GH, z.Hd, (int)BL}, {(int)GH, z.Cd, z.V}, {z.V, z.Cd, (int)GH},
                         {(int)BT, z.Hd, z.Hd}, {z.Hd, z.Hd, (int)BT}, {a, z.Hd, (int)BT}, {(int)BT, z.Hd, a},
                         {a, z.Hd, (int)BL}, {(int)BL, z.Hd, a}, {z.Hd, 1, (int)BL}, {z.Hd, 1, (int)BT},
                         {(int)GH, z.Hd, (int)BL}, {(int)BL, z.Hd, (int)GH}};
  w.gemm_bytes = max_gemm_ws(dims, 17);
  {   // grouped weight-gradient launches of lr_decoder_backward: a layer's W_hh (+ W_ih) products, and the head's
    const int bl = (int)BL, gh = (int)GH;
    int M1[3], N1[3], K1[3], n1 = 0;
    if (z.G == 3) {
      M1[n1] = 2 * z.Hd; N1[n1] = z.Hd; K1[n1] = bl; ++n1;
      M1[n1] = z.Hd; N1[n1] = z.Hd; K1[n1] = bl; ++n1;
    } else {
      M1[n1] = gh; N1[n1] = z.Hd; K1[n1] = bl; ++n1;
    }
    M1[n1] = gh; N1[n1] = z.Hd; K1[n1] = bl; ++n1;     // an upper layer's W_ih
    const size_t g1 = lr_sgemm_grouped_workspace_bytes(n1, M1, N1, K1);
    int Mh[4] = {gh, z.V, z.Hd, z.Hd}, Nh[4] = {z.Cd, z.Hd, z.Hd, z.Hd}, Kh[4] = {z.V, bl, bl, bl};
    const size_t g2 = lr_sgemm_grouped_workspace_bytes(4, Mh, Nh, Kh);
    if (g1 > w.gemm_bytes) w.gemm_bytes = g1;
    if (g2 > w.gemm_bytes) w.gemm_bytes = g2;
  }
  w.gemm = take((w.gemm_bytes + 3) / 4);
  w.hprev = take(wide_wgrad(z.G, z.Hd) ? BL * z.Hd : 0);    // (see lr_decoder_backward: h_{t-1} of every (sample, step) row)
  w.total = o;
  return w;
}

bool sizes_ok(int mode, int type, int B, int L, int T, int Hd, int Cd, int V, int A, int NL = 1) {
  return (mode == LR_RNN_GRU || mode == LR_RNN_LSTM || mode == LR_RNN_TANH) && type >= ATT_NONE &&
         type <= ATT_CONCAT && B > 0 && B <= 65535 && L > 0 && L <= 65535 && T > 0 && Hd > 0 && Hd % 4 == 0 &&
         Cd > 0 && V > 0 && V <= 1024 && (type != ATT_CONCAT || A > 0) && NL >= 1 && NL <= MAXL;
}
inline int layers_of(const lr_decoder_upper* up) { return up ? up->num_layers : 1; }
// rows [i0, i1) of every sample: out = x * mask (inter-layer dropout multipliers), 4 floats per thread
__global__ void dec_mask_rows_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                     float* __restrict__ out, int L, int Hd, int i0, int n) {
  const int b = blockIdx.y;
  const int64_t base = ((int64_t)b * L + i0) * Hd;
  const int64_t total4 = (int64_t)n * Hd / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x + base)[i];
    const float4 m = reinterpret_cast<const float4*>(mask + base)[i];
    reinterpret_cast<float4*>(out + base)[i] = make_float4(a.x * m.x, a.y * m.y, a.z * m.z, a.w * m.w);
  }
}
// out[b][t][:] = t ? hs[b][t - 1][:] : h0[b][:] — the state every step of the loop STARTED from (4 floats per thread)
__global__ void dec_hprev_kernel(const float* __restrict__ hs, const float* __restrict__ h0, float* __restrict__ out, int L,
                                 int Hd) {
  const int b = blockIdx.y;
  const int q = Hd / 4;
  const int64_t total4 = (int64_t)L * q;
  const float4* hs4 = reinterpret_cast<const float4*>(hs + (int64_t)b * L * Hd);
  const float4* h04 = reinterpret_cast<const float4*>(h0 + (int64_t)b * Hd);
  float4* o4 = reinterpret_cast<float4*>(out + (int64_t)b * L * Hd);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x)
    o4[i] = i < q ? h04[i] : hs4[i - q];
}

#define LR_TRY(expr)              \
  do {                            \
    const int st__ = (expr);      \
    if (st__ != LR_OK) return st__; \
  } while (0)

// two-stage deterministic column sum of x [rows][ld] -> out[ncol] (+= when accumulate)
int colsum_into(const float* x, int ld, int rows, int ncol, float* scratch, float* out, int accumulate,
                hipStream_t stream) {
  LR_TRY(lr_colsum_partial(x, ld, rows, ncol, scratch, stream));
  LR_LAUNCH(dec_colsum_final_kernel, dim3((ncol + 255) / 256), dim3(256), 0, stream, (const float*)scratch, out, ncol,
            accumulate);
  return lr_launch_status();
}

}  // namespace

extern "C" size_t lr_decoder_reserve_bytes(int mode, int attn_type, int num_layers, int B, int L, int T, int Hd,
                                           int Cd, int V, int A) {
  if (!sizes_ok(mode, attn_type, B, L, T, Hd, Cd, V, A, num_layers)) return 0;
  const Sizes z = {B, L, T, Hd, Cd, V, A, gates_of_mode(mode), attn_type, num_layers};
  return res_layout(z).total * sizeof(float);
}

extern "C" size_t lr_decoder_workspace_bytes(int mode, int attn_type, int num_layers, int B, int L, int T, int Hd,
                                             int Cd, int V, int A) {
  if (!sizes_ok(mode, attn_type, B, L, T, Hd, Cd, V, A, num_layers)) return 0;
  const Sizes z = {B, L, T, Hd, Cd, V, A, gates_of_mode(mode), attn_type, num_layers};
  return ws_layout(z).total * sizeof(float);
}

extern "C" int lr_decoder_forward(int mode, int attn_type, const lr_decoder_params* p, const lr_decoder_upper* up,
                                  const int32_t* tokens,
                                  const uint8_t* teacher_forced_host, const float* enc, const int32_t* enc_lens,
                                  const float* h0, const float* c0, const int32_t* step_lens, uint64_t seed,
                                  float* log_probs, int32_t* sampled, float* h_n, float* c_n, void* reserve,
                                  size_t reserve_bytes, int B, int L, int T, int Hd, int Cd, int V, int A,
                                  lr_stream_t stream_) {
  const int NL = layers_of(up);
  LR_CHECK_ARG(sizes_ok(mode, attn_type, B, L, T, Hd, Cd, V, A, NL));
  LR_CHECK_ARG(p && tokens && teacher_forced_host && enc && enc_lens && h0 && step_lens && log_probs && sampled &&
               reserve);
  LR_CHECK_ARG(p->emb && p->w_ih && p->w_hh && p->b_ih && p->b_hh && p->w_o && p->b_o && p->out_mask);
  LR_CHECK_ARG(attn_type == ATT_NONE || (p->w_c && p->b_c));
  LR_CHECK_ARG(mode != LR_RNN_LSTM || c0);
  for (int k = 1; k < NL; ++k) LR_CHECK_ARG(up->w_ih[k - 1] && up->w_hh[k - 1] && up->b_ih[k - 1] && up->b_hh[k - 1]);
  const Sizes z = {B, L, T, Hd, Cd, V, A, gates_of_mode(mode), attn_type, NL};
  const Res r = res_layout(z);
  if (reserve_bytes < r.total * sizeof(float)) return LR_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* base = (float*)reserve;
  const int G = z.G, GH = G * Hd, BL = B * L, R = B * T;
  const bool attn = attn_type != ATT_NONE;
  float* EW = base + r.EW;
  float* hs = base + r.hs;          // the top layer's states
  const size_t state = (size_t)B * Hd;   // floats per layer of h0 / c0 / h_n / c_n
  const float* w_hh[MAXL];
  const float* w_ihu[MAXL];
  const float* b_ihl[MAXL];
  const float* b_hhl[MAXL];
  for (int k = 0; k < NL; ++k) {
    w_hh[k] = k == 0 ? p->w_hh : up->w_hh[k - 1];
    w_ihu[k] = k == 0 ? nullptr : up->w_ih[k - 1];
    b_ihl[k] = k == 0 ? p->b_ih : up->b_ih[k - 1];
    b_hhl[k] = k == 0 ? p->b_hh : up->b_hh[k - 1];
  }
  const float* drop = (up && NL > 1) ? up->drop_mask : nullptr;   // [NL-1][B][L][Hd] or NULL
  float* logits = base + r.logits;
  float* wts = base + r.wts;
  float* ctx = base + r.ctx;
  float* pre = base + r.pre;
  float* ph = base + r.ph;
  void* gws = base + r.gemm;
  int32_t* ids = (int32_t*)(base + r.ids);

  // every input token known up front (teacher forcing on all steps) and a shape the cluster recurrence covers:
  // each layer's L steps go out as one launch
  bool one_launch = r.xch_bytes != 0;
  for (int i = 1; i < L; ++i) one_launch = one_launch && teacher_forced_host[i];
  // table of input projections of layer 0: EW = emb @ W_ih^T + folded bias
  for (int k = 0; k < NL; ++k) {
    LR_TRY(lr_rnn_fold_bias(b_ihl[k], b_hhl[k], base + r.biasf[k], G, Hd, stream));
    if (one_launch) continue;   // the cluster recurrence packs W_hh itself and reads h0 / c0 directly
    LR_TRY(lr_rnn_pack_w(w_hh[k], base + r.wp[k], G, Hd, 0, stream));
    lr_clear_error();
    if (hipMemsetAsync(base + r.hp[k], 0, 2 * r.hp_slot * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;
    // step 0 reads parity (0+1)&1 = 1
    LR_TRY(lr_rnn_pack_state(h0 + k * state, base + r.hp[k] + r.hp_slot, B, Hd, stream));
  }
  LR_TRY(lr_sgemm_impl(0, 1, V, GH, Cd, 1.f, p->emb, Cd, p->w_ih, Cd, 0.f, EW, GH, base + r.biasf[0], 0, 0, gws,
                       r.gemm_bytes, stream));

  // step-independent halves of the attention logits
  const float* src = nullptr;     // dot: enc; general: GE = enc W_g        [B][T][Hd]
  const float* cterm = nullptr;   // general: cE = enc . b_g; 1_layer_nn: se = enc . w_e   [B][T]
  if (attn_type == ATT_DOT) {
    src = enc;
  } else if (attn_type == ATT_GENERAL) {
    LR_CHECK_ARG(p->attn_w1 && p->attn_b1);
    // (W_g h + b_g) . enc[t] = h . (enc[t] W_g) + enc[t] . b_g
    LR_TRY(lr_sgemm_impl(0, 0, R, Hd, Hd, 1.f, enc, Hd, p->attn_w1, Hd, 0.f, base + r.aux1, Hd, nullptr, 0, 0, gws,
                         r.gemm_bytes, stream));
    LR_TRY(lr_sgemm_impl(0, 1, R, 1, Hd, 1.f, enc, Hd, p->attn_b1, Hd, 0.f, base + r.aux2, 1, nullptr, 0, 0,
                         nullptr, 0, stream));
    src = base + r.aux1;
    cterm = base + r.aux2;
  } else if (attn_type == ATT_1LNN) {
    LR_CHECK_ARG(p->attn_w1 && p->attn_b1);
    LR_TRY(lr_sgemm_impl(0, 1, R, 1, Hd, 1.f, enc, Hd, p->attn_w1, 2 * Hd, 0.f, base + r.aux2, 1, nullptr, 0, 0,
                         nullptr, 0, stream));   // se = enc . w_e  (w_e = first Hd entries of w)
    cterm = base + r.aux2;
  } else if (attn_type == ATT_CONCAT) {
    LR_CHECK_ARG(p->attn_w1 && p->attn_b1 && p->attn_w2 && p->attn_b2);
    LR_TRY(lr_sgemm_impl(0, 1, R, A, Hd, 1.f, enc, Hd, p->attn_w1, 2 * Hd, 0.f, base + r.aux1, A, p->attn_b1, 0, 0,
                         gws, r.gemm_bytes, stream));   // PE = enc @ W1e^T + b1
  }
  int out_rows = OUT_ROWS;
  while (out_rows > 1 && (size_t)out_rows * (Hd + V) * sizeof(float) > 60 * 1024) out_rows >>= 1;
  const size_t out_lds = (size_t)out_rows * (Hd + V) * sizeof(float);
  if (out_lds > 60 * 1024 || (size_t)2 * A * sizeof(float) > 60 * 1024) return LR_ERR_UNSUPPORTED;

  // rows (b, i), i in [i0, i1), of a [B][L][ld] buffer times a torch-layout weight [N][K] (+ bias)
  auto rows_gemm = [&](int N, int K, const float* Am, int lda, const float* W, int ldw, float beta, float* C,
                       int ldc, const float* bias, int i0, int i1) -> int {
    if (i0 == 0 && i1 == L)
      return lr_sgemm_impl(0, 1, BL, N, K, 1.f, Am, lda, W, ldw, beta, C, ldc, bias, 0, 0, gws, r.gemm_bytes, stream);
    return lr_sgemm_batched_impl(0, 1, i1 - i0, N, K, 1.f, Am + (size_t)i0 * lda, lda, (int64_t)L * lda, W, ldw, 0,
                                 beta, C + (size_t)i0 * ldc, ldc, (int64_t)L * ldc, bias, B, stream);
  };
  // attention + concat_layer + output head of steps [i0, i1), whose RNN states are in hs
  auto flush = [&](int i0, int i1) -> int {
    const int n = i1 - i0;
    if (n <= 0) return LR_OK;
    if (attn) {
      if (attn_type == ATT_DOT || attn_type == ATT_GENERAL) {
        // logits[b] (n x T) = hs[b] (n x Hd) . src[b]^T
        LR_TRY(lr_sgemm_batched_impl(0, 1, n, T, Hd, 1.f, hs + (size_t)i0 * Hd, Hd, (int64_t)L * Hd, src, Hd,
                                     (int64_t)T * Hd, 0.f, logits + (size_t)i0 * T, T, (int64_t)L * T, nullptr, B,
                                     stream));
      } else if (attn_type == ATT_CONCAT) {
        LR_TRY(rows_gemm(A, Hd, hs, Hd, p->attn_w1 + Hd, 2 * Hd, 0.f, ph, A, nullptr, i0, i1));   // ph = W1h h
        LR_LAUNCH(dec_concat_logits_kernel, dim3(n, B), dim3(256), (size_t)2 * A * sizeof(float), stream,
                  (const float*)(base + r.aux1), (const float*)ph, p->attn_w2, p->attn_b2, logits, L, T, A, i0);
        LR_TRY(lr_launch_status());
      }
      LR_LAUNCH(dec_attn_softmax_kernel, dim3(n, B), dim3(64), 0, stream, attn_type, (const float*)hs, enc_lens, cterm,
                attn_type == ATT_1LNN ? p->attn_w1 + Hd : (const float*)nullptr,
                attn_type == ATT_1LNN ? p->attn_b1 : (const float*)nullptr, logits, wts, L, T, Hd, i0);
      LR_TRY(lr_launch_status());
      // ctx[b] (n x Hd) = wts[b] (n x T) . enc[b] (T x Hd)
      LR_TRY(lr_sgemm_batched_impl(0, 0, n, Hd, T, 1.f, wts + (size_t)i0 * T, T, (int64_t)L * T, enc, Hd,
                                   (int64_t)T * Hd, 0.f, ctx + (size_t)i0 * Hd, Hd, (int64_t)L * Hd, nullptr, B,
                                   stream));
      // concat_layer([ctx; h]) = W_c[:, :Hd] ctx + W_c[:, Hd:] h + b_c
      LR_TRY(rows_gemm(Hd, Hd, ctx, Hd, p->w_c, 2 * Hd, 0.f, pre, Hd, p->b_c, i0, i1));
      LR_TRY(rows_gemm(Hd, Hd, hs, Hd, p->w_c + Hd, 2 * Hd, 1.f, pre, Hd, nullptr, i0, i1));
    }
    const int total = n * B;
    LR_LAUNCH(dec_out_fwd_kernel, dim3((total + out_rows - 1) / out_rows), dim3(256), out_lds, stream,
              attn ? pre : hs, p->w_o, p->b_o, p->out_mask, log_probs, sampled, L, Hd, V, i0, n, total, out_rows,
              attn ? 1 : 0, seed);
    return lr_launch_status();
  };

  // A run of steps whose input tokens are known goes through the stack LAYER by layer (a layer's input
  // projection over the run is one GEMM on the layer below's outputs, nn.GRU/LSTM(num_layers) semantics:
  // better_model.py:147-148,181); a sampled-input step is a run of one.
  int done = 0;   // steps whose head has been computed
  for (int i = 0; i < L;) {
    int run = 1;
    if (teacher_forced_host[i] || i == 0) {
      while (i + run < L && teacher_forced_host[i + run]) ++run;
      LR_LAUNCH(dec_gather_kernel, dim3(run, B), dim3(256), 0, stream, (const float*)EW, tokens,
                (const int32_t*)sampled, ids, base + r.gates[0], L, GH, V, i, 1);
    } else {
      LR_TRY(flush(done, i));   // the input of step i is the sample drawn from step i-1's output
      done = i;
      LR_LAUNCH(dec_gather_kernel, dim3(1, B), dim3(256), 0, stream, (const float*)EW, tokens,
                (const int32_t*)sampled, ids, base + r.gates[0], L, GH, V, i, 0);
    }
    LR_TRY(lr_launch_status());
    for (int k = 0; k < NL; ++k) {
      if (k > 0) {
        const float* x = base + r.hsl[k - 1];
        if (drop) {   // nn.GRU/LSTM(dropout=p): the outputs of every layer but the last are dropped out
          int gx_ = (int)(((int64_t)run * Hd / 4 + 255) / 256);
          if (gx_ > 64) gx_ = 64;
          LR_LAUNCH(dec_mask_rows_kernel, dim3(gx_, B), dim3(256), 0, stream, x, drop + (size_t)(k - 1) * BL * Hd,
                    base + r.xm[k - 1], L, Hd, i, run);
          LR_TRY(lr_launch_status());
          x = base + r.xm[k - 1];
        }
        LR_TRY(rows_gemm(GH, Hd, x, Hd, w_ihu[k], Hd, 0.f, base + r.gates[k], GH, base + r.biasf[k], i, i + run));
      }
      if (one_launch) {
        // the whole loop is one run (teacher_forcing_ratio = 1: the shipped configs and eval): the layer's L steps as
        // ONE launch, fp32-faithful (lr_rnn_cluster.hip), started from the encoder's final state
        const float* whh1[1] = {w_hh[k]};
        const float* bhh1[1] = {b_hhl[k]};
        LR_TRY(lr_rnn_cluster_forward(G, base + r.gates[k], base + r.extra[k], base + r.hsl[k], whh1, bhh1,
                                      h0 + k * state, c0 ? c0 + k * state : nullptr, step_lens, base + r.wp[k],
                                      base + r.xch, B, L, 1, Hd, stream));
        continue;
      }
      for (int s = i; s < i + run; ++s)
        LR_TRY(lr_rnn_step_fwd(G, base + r.gates[k], base + r.extra[k], base + r.hsl[k], base + r.hp[k], step_lens,
                               base + r.wp[k], b_hhl[k], h0 + k * state, c0 ? c0 + k * state : nullptr, B, L, Hd, s,
                               stream));
    }
    i += run;
  }
  LR_TRY(flush(done, L));
  // state after the last step (what the reference's step returns as final_state, better_model.py:181)
  lr_clear_error();
  for (int k = 0; k < NL; ++k) {
    if (h_n && hipMemcpy2DAsync(h_n + k * state, (size_t)Hd * sizeof(float), base + r.hsl[k] + (size_t)(L - 1) * Hd,
                                (size_t)L * Hd * sizeof(float), (size_t)Hd * sizeof(float), B,
                                hipMemcpyDeviceToDevice, stream) != hipSuccess)
      return LR_ERR_LAUNCH;
    if (c_n && G == 4 && hipMemcpy2DAsync(c_n + k * state, (size_t)Hd * sizeof(float),
                                          base + r.extra[k] + (size_t)(L - 1) * Hd, (size_t)L * Hd * sizeof(float),
                                          (size_t)Hd * sizeof(float), B, hipMemcpyDeviceToDevice, stream) != hipSuccess)
      return LR_ERR_LAUNCH;
  }
  return LR_OK;
}

// parts: 1 = the data half (everything the encoder's backward waits for: d_enc, dh0, dc0; leaves dG, dlogits in the
// workspace), 2 = the weight half (every parameter gradient, from what part 1 left), 3 = both.  The two halves are
// separable for a single-layer loop (lr_decoder_backward_splittable), any attention type: every shipped flag file.
extern "C" int lr_decoder_backward_splittable(int attn_type, int num_layers) {
  (void)attn_type;
  return num_layers == 1;   // (layers above the first overwrite the gate-gradient buffer their weight products read)
}
extern "C" int lr_decoder_backward_parts(int mode, int attn_type, const lr_decoder_params* p, const lr_decoder_upper* up,
                                         const lr_decoder_grads* g, const lr_decoder_upper_grads* gup,
                                         const float* enc, const int32_t* enc_lens, const float* h0, const float* c0,
                                         const int32_t* step_lens, const float* log_probs, const float* d_log_probs,
                                         const float* dh_n, const float* dc_n, float* d_enc, float* dh0, float* dc0,
                                         const void* reserve, size_t reserve_bytes,
                                         void* workspace, size_t workspace_bytes, int accumulate, int B, int L, int T,
                                         int Hd, int Cd, int V, int A, int parts, lr_stream_t stream_) {
  const int NL = layers_of(up);
  LR_CHECK_ARG(parts == 3 || ((parts == 1 || parts == 2) && lr_decoder_backward_splittable(attn_type, NL)));
  const bool do_data = (parts & 1) != 0, do_weights = (parts & 2) != 0;
  LR_CHECK_ARG(sizes_ok(mode, attn_type, B, L, T, Hd, Cd, V, A, NL));
  LR_CHECK_ARG(p && g && enc && enc_lens && h0 && step_lens && log_probs && d_log_probs && d_enc && dh0 &&
               reserve && workspace);
  LR_CHECK_ARG(g->emb && g->w_ih && g->w_hh && g->b_ih && g->b_hh && g->w_o && g->b_o);
  LR_CHECK_ARG(mode != LR_RNN_LSTM || (c0 && dc0));
  LR_CHECK_ARG(NL == 1 || gup);
  for (int k = 1; k < NL; ++k)
    LR_CHECK_ARG(up->w_ih[k - 1] && up->w_hh[k - 1] && gup->w_ih[k - 1] && gup->w_hh[k - 1] && gup->b_ih[k - 1] &&
                 gup->b_hh[k - 1]);
  const Sizes z = {B, L, T, Hd, Cd, V, A, gates_of_mode(mode), attn_type, NL};
  const Res r = res_layout(z);
  const Wsp w = ws_layout(z);
  if (reserve_bytes < r.total * sizeof(float)) return LR_ERR_WORKSPACE;
  if (workspace_bytes < w.total * sizeof(float)) return LR_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const float* rb = (const float*)reserve;
  float* wb = (float*)workspace;
  const int G = z.G, GH = G * Hd, R = B * T, BL = B * L;
  const bool attn = attn_type != ATT_NONE;
  const float beta = accumulate ? 1.f : 0.f;
  void* gws = wb + w.gemm;
  const float* hs = rb + r.hs;
  const float* wts = rb + r.wts;
  const float* ctx = rb + r.ctx;
  const float* nh = attn ? rb + r.pre : hs;   // what output_proj was applied to
  const int32_t* ids = (const int32_t*)(rb + r.ids);
  float* dG = wb + w.dG;
  float* dy = wb + w.dy;
  float* dlogits = wb + w.dlogits;
  float* dpre = wb + w.dpre;
  float* dctx = wb + w.dctx;
  float* dlg = wb + w.dlg;
  float* dsum = wb + w.dsum;
  float* colsum = wb + w.colsum;
  if ((size_t)L * T * sizeof(float) > 60 * 1024 && attn_type == ATT_CONCAT) return LR_ERR_UNSUPPORTED;

  const size_t state = (size_t)B * Hd;
  const float* drop = (up && NL > 1) ? up->drop_mask : nullptr;

  // ---- output head, all (sample, step) rows at once -------------------------------------------------
  if (do_data) {
    LR_LAUNCH(dec_out_bwd_rows_kernel, dim3((BL + 3) / 4), dim3(256), 0, stream, d_log_probs, log_probs, dlogits, BL, V);
    LR_TRY(lr_launch_status());
    // d(new_h) = dlogits @ W_o; without attention new_h is the RNN state itself
    LR_TRY(lr_sgemm_impl(0, 0, BL, Hd, V, 1.f, dlogits, V, p->w_o, Hd, 0.f, attn ? dpre : dy, Hd, nullptr, 0, 0, gws,
                         w.gemm_bytes, stream));
  }
  if (!attn) {
    lr_clear_error();
    if (do_data && hipMemsetAsync(d_enc, 0, (size_t)R * Hd * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;
  } else {   // (attention: every launch below belongs to ONE half — data D, weights W)
    if (do_data) {   // D: through the concat layer and the attention weights into dctx, dlg, d_enc, dy
    const int64_t n4 = (int64_t)BL * Hd / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    LR_LAUNCH(dec_tanh_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dpre, nh, n4);
    LR_TRY(lr_launch_status());
    // concat_layer: d ctx = dpre @ W_c[:, :Hd];  dh (direct part) = dpre @ W_c[:, Hd:]
    LR_TRY(lr_sgemm_impl(0, 0, BL, Hd, Hd, 1.f, dpre, Hd, p->w_c, 2 * Hd, 0.f, dctx, Hd, nullptr, 0, 0, gws,
                         w.gemm_bytes, stream));
    LR_TRY(lr_sgemm_impl(0, 0, BL, Hd, Hd, 1.f, dpre, Hd, p->w_c + Hd, 2 * Hd, 0.f, dy, Hd, nullptr, 0, 0, gws,
                         w.gemm_bytes, stream));
    // ---- attention, per sample: d weights = dctx[b] (L x Hd) . enc[b]^T -> masked_softmax backward ----
    LR_TRY(lr_sgemm_batched_impl(0, 1, L, T, Hd, 1.f, dctx, Hd, (int64_t)L * Hd, enc, Hd, (int64_t)T * Hd, 0.f, dlg,
                                 T, (int64_t)L * T, nullptr, B, stream));
    LR_LAUNCH(dec_attn_softmax_bwd_kernel, dim3((BL + 3) / 4), dim3(256), 0, stream, rb + r.logits, wts, enc_lens,
              dlg, dsum, BL, L, T);
    LR_TRY(lr_launch_status());
    // d_enc[b] (T x Hd) = wts[b]^T (T x L) . dctx[b] (L x Hd)
    LR_TRY(lr_sgemm_batched_impl(1, 0, T, Hd, L, 1.f, wts, T, (int64_t)L * T, dctx, Hd, (int64_t)L * Hd, 0.f, d_enc,
                                 Hd, (int64_t)T * Hd, nullptr, B, stream));
    if (attn_type == ATT_DOT || attn_type == ATT_GENERAL) {
      const float* src = attn_type == ATT_DOT ? enc : rb + r.aux1;
      float* dsrc = attn_type == ATT_DOT ? d_enc : wb + w.dsrc;
      // logit[b] = hs[b] . src[b]^T:  dh += dlg[b] (L x T) . src[b];  dsrc[b] (+)= dlg[b]^T . hs[b]
      LR_TRY(lr_sgemm_batched_impl(0, 0, L, Hd, T, 1.f, dlg, T, (int64_t)L * T, src, Hd, (int64_t)T * Hd, 1.f, dy, Hd,
                                   (int64_t)L * Hd, nullptr, B, stream));
      LR_TRY(lr_sgemm_batched_impl(1, 0, T, Hd, L, 1.f, dlg, T, (int64_t)L * T, hs, Hd, (int64_t)L * Hd,
                                   attn_type == ATT_DOT ? 1.f : 0.f, dsrc, Hd, (int64_t)T * Hd, nullptr, B, stream));
    }
    if (attn_type == ATT_GENERAL || attn_type == ATT_1LNN) {
      LR_LAUNCH(dec_sum_steps_kernel, dim3(B), dim3(128), 0, stream, (const float*)dlg, wb + w.dcterm, L, T);
      LR_TRY(lr_launch_status());
    }
    }   // (do_data)
    if (attn_type == ATT_GENERAL) {
      LR_CHECK_ARG(g->attn_w1 && g->attn_b1);
      // GE = enc @ W_g: dW_g = enc^T @ dGE, d_enc += dGE @ W_g^T; cE = enc . b_g: db_g = enc^T dcE, d_enc += dcE b_g^T
      if (do_weights)
        LR_TRY(lr_sgemm_impl(1, 0, Hd, Hd, R, 1.f, enc, Hd, wb + w.dsrc, Hd, beta, g->attn_w1, Hd, nullptr, 0, 0, gws,
                             w.gemm_bytes, stream));
      if (do_data)
        LR_TRY(lr_sgemm_impl(0, 1, R, Hd, Hd, 1.f, wb + w.dsrc, Hd, p->attn_w1, Hd, 1.f, d_enc, Hd, nullptr, 0, 0, gws,
                             w.gemm_bytes, stream));
      if (do_weights)
        LR_TRY(lr_sgemm_impl(1, 0, Hd, 1, R, 1.f, enc, Hd, wb + w.dcterm, 1, beta, g->attn_b1, 1, nullptr, 0, 0, gws,
                             w.gemm_bytes, stream));
      if (do_data)
        LR_TRY(lr_sgemm_impl(0, 0, R, Hd, 1, 1.f, wb + w.dcterm, 1, p->attn_b1, Hd, 1.f, d_enc, Hd, nullptr, 0, 0,
                             nullptr, 0, stream));
    } else if (attn_type == ATT_1LNN) {
      LR_CHECK_ARG(g->attn_w1 && g->attn_b1);
      // logit = w_e . enc[t] + w_h . h + b:  dw_e = enc^T dse, d_enc += dse w_e^T,
      // dh += dsum w_h, dw_h = hs^T dsum, db = sum dsum
      if (do_weights)
        LR_TRY(lr_sgemm_impl(1, 0, Hd, 1, R, 1.f, enc, Hd, wb + w.dcterm, 1, beta, g->attn_w1, 1, nullptr, 0, 0, gws,
                             w.gemm_bytes, stream));
      if (do_data) {
        LR_TRY(lr_sgemm_impl(0, 1, R, Hd, 1, 1.f, wb + w.dcterm, 1, p->attn_w1, 1, 1.f, d_enc, Hd, nullptr, 0, 0,
                             nullptr, 0, stream));
        LR_TRY(lr_sgemm_impl(0, 0, BL, Hd, 1, 1.f, dsum, 1, p->attn_w1 + Hd, Hd, 1.f, dy, Hd, nullptr, 0, 0, nullptr, 0,
                             stream));
      }
      if (do_weights) {
        LR_TRY(lr_sgemm_impl(1, 0, Hd, 1, BL, 1.f, hs, Hd, dsum, 1, beta, g->attn_w1 + Hd, 1, nullptr, 0, 0, gws,
                             w.gemm_bytes, stream));
        LR_TRY(colsum_into(dsum, 1, BL, 1, colsum, g->attn_b1, accumulate, stream));
      }
    } else if (attn_type == ATT_CONCAT) {
      LR_CHECK_ARG(g->attn_w1 && g->attn_b1 && g->attn_w2 && g->attn_b2);
      if (do_data) {
        LR_LAUNCH(dec_concat_bwd_kernel, dim3((A + 63) / 64, B), dim3(64), (size_t)L * T * sizeof(float), stream,
                  rb + r.aux1, rb + r.ph, p->attn_w2, (const float*)dlg, enc_lens, wb + w.dPE, wb + w.dph, wb + w.dw2p,
                  L, T, A);
        LR_TRY(lr_launch_status());
        // W1 = [W1e | W1h] (A x 2Hd): PE = enc W1e^T + b1, ph = hs W1h^T
        LR_TRY(lr_sgemm_impl(0, 0, BL, Hd, A, 1.f, wb + w.dph, A, p->attn_w1 + Hd, 2 * Hd, 1.f, dy, Hd, nullptr, 0, 0,
                             gws, w.gemm_bytes, stream));                                  // dh += dph W1h
        LR_TRY(lr_sgemm_impl(0, 0, R, Hd, A, 1.f, wb + w.dPE, A, p->attn_w1, 2 * Hd, 1.f, d_enc, Hd, nullptr, 0, 0, gws,
                             w.gemm_bytes, stream));                                       // d_enc += dPE W1e
      }
      if (do_weights) {
        LR_TRY(lr_sgemm_impl(1, 0, A, Hd, BL, 1.f, wb + w.dph, A, hs, Hd, beta, g->attn_w1 + Hd, 2 * Hd, nullptr, 0, 0,
                             gws, w.gemm_bytes, stream));                                  // dW1h = dph^T hs
        LR_TRY(lr_sgemm_impl(1, 0, A, Hd, R, 1.f, wb + w.dPE, A, enc, Hd, beta, g->attn_w1, 2 * Hd, nullptr, 0, 0, gws,
                             w.gemm_bytes, stream));                                       // dW1e = dPE^T enc
        LR_TRY(colsum_into(wb + w.dPE, A, R, A, colsum, g->attn_b1, accumulate, stream));
        LR_TRY(colsum_into(wb + w.dw2p, A, B, A, colsum, g->attn_w2, accumulate, stream));
        LR_TRY(colsum_into(dsum, 1, BL, 1, colsum, g->attn_b2, accumulate, stream));
      }
    }
  }

  // ---- the only sequential part: the RNN backward over the L steps, top layer first -----------------
  const int ldg = 4 * Hd;
  for (int k = NL - 1; k >= 0; --k) {
    const float* w_hh_k = k == 0 ? p->w_hh : up->w_hh[k - 1];
    float* gw_hh = k == 0 ? g->w_hh : gup->w_hh[k - 1];
    float* gb_ih = k == 0 ? g->b_ih : gup->b_ih[k - 1];
    float* gb_hh = k == 0 ? g->b_hh : gup->b_hh[k - 1];
    const float* hs_k = rb + r.hsl[k];
    const float* h0_k = h0 + k * state;
    const float* c0_k = c0 ? c0 + k * state : nullptr;
    if (!do_data) {
      // (the weight half of a split call: dG is in the workspace)
    } else if (w.xch_bytes) {
      // the L reverse steps AND the gradient into the initial state as one launch (lr_rnn_cluster.hip)
      const float* whh1[1] = {w_hh_k};
      LR_TRY(lr_rnn_cluster_backward(G, rb + r.gates[k], rb + r.extra[k], hs_k, dy, dh_n ? dh_n + k * state : nullptr,
                                     dc_n ? dc_n + k * state : nullptr, dG, dh0 + k * state,
                                     dc0 ? dc0 + k * state : nullptr, h0_k, c0_k, whh1, step_lens, wb + w.wpT, wb + w.xch,
                                     B, L, 1, Hd, stream));
    } else {
      LR_TRY(lr_rnn_pack_w(w_hh_k, wb + w.wpT, G, Hd, 1, stream));
      lr_clear_error();
      if (hipMemsetAsync(wb + w.dgp, 0, 2 * w.dgp_slot * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;
      for (int s2 = 0; s2 < L; ++s2)
        LR_TRY(lr_rnn_step_bwd(G, rb + r.gates[k], rb + r.extra[k], hs_k, dy, dh_n ? dh_n + k * state : nullptr,
                               dc_n ? dc_n + k * state : nullptr, dG, wb + w.dcar, wb + w.dgp, step_lens, wb + w.wpT,
                               h0_k, c0_k, B, L, Hd, s2, stream));
      // gradient into the initial state (the encoder's final state of this layer)
      LR_TRY(lr_rnn_dh0(G, wb + w.dcar, wb + w.dgp + (size_t)((L - 1) & 1) * w.dgp_slot, wb + w.wpT, dh0 + k * state,
                        dc0 ? dc0 + k * state : nullptr, B, L, Hd, stream));
    }
    if (!do_weights) continue;   // (splittable: a single layer — nothing below waits for this layer's dy)
    // W_hh (h_prev of step t is hs[b][t-1]; step 0 used h0) and, for an upper layer, W_ih (x = the (dropped-out)
    // states of the layer below)
    if (wide_wgrad(G, Hd)) {
      // split-bf16 products of ONE launch: h_{t-1} of every (sample, step) row is gathered once (the initial state in the
      // rows of step 0), so the W_hh product is a plain dG^T . h_prev over all B * L rows — no row shift, no separate
      // product for the initial state
      float* hprev = wb + w.hprev;
      {
        int gx_ = (int)(((int64_t)L * Hd / 4 + 255) / 256);
        if (gx_ > 64) gx_ = 64;
        LR_LAUNCH(dec_hprev_kernel, dim3(gx_, B), dim3(256), 0, stream, hs_k, h0_k, hprev, L, Hd);
        LR_TRY(lr_launch_status());
      }
      lr_fgemm_job jobs[3];
      int n = 0;
      auto add = [&](int M, const float* A, const float* Bm, float* C) {
        lr_fgemm_job& j = jobs[n++];
        j.A = A; j.B = Bm; j.C = C;
        j.bias = nullptr; j.addend = nullptr; j.mask = nullptr; j.colsum = nullptr; j.slabs = nullptr;
        j.M = M; j.N = Hd; j.K = BL; j.lda = ldg; j.ldb = Hd; j.ldc = Hd;
        j.ldadd = 0; j.add_period = 0; j.ldmask = 0; j.flags = 0; j.splits = 1;
        j.alpha = 1.f; j.beta = beta;
        j.b_shift = 0; j.b_period = 0;
      };
      if (G == 3) {
        add(2 * Hd, dG, hprev, gw_hh);
        add(Hd, dG + 3 * Hd, hprev, gw_hh + (size_t)2 * Hd * Hd);
      } else {
        add(GH, dG, hprev, gw_hh);
      }
      if (k > 0) add(GH, dG, drop ? rb + r.xm[k - 1] : rb + r.hsl[k - 1], gup->w_ih[k - 1]);
      LR_TRY(lr_fgemm_launch(LR_FGEMM_X3, LR_FGEMM_TN, 0, 0, jobs, n, stream));
    } else {
      // ONE grouped fp32 launch + one combine (lr_gemm.hip), then the initial state's rows
      int Ms[3], Ns[3], Ks[3], ldas[3], ldbs[3], ldcs[3], shifts[3], periods[3], n = 0;
      const float* As[3];
      const float* Bs[3];
      float* Cs[3];
      auto add = [&](int M, const float* A, const float* Bm, float* C, int shift, int period) {
        Ms[n] = M; Ns[n] = Hd; Ks[n] = BL; As[n] = A; ldas[n] = ldg; Bs[n] = Bm; ldbs[n] = Hd; Cs[n] = C; ldcs[n] = Hd;
        shifts[n] = shift; periods[n] = period;
        ++n;
      };
      if (G == 3) {
        add(2 * Hd, dG, hs_k, gw_hh, -1, L);
        add(Hd, dG + 3 * Hd, hs_k, gw_hh + (size_t)2 * Hd * Hd, -1, L);
      } else {
        add(GH, dG, hs_k, gw_hh, -1, L);
      }
      if (k > 0) add(GH, dG, drop ? rb + r.xm[k - 1] : rb + r.hsl[k - 1], gup->w_ih[k - 1], 0, 0);
      LR_TRY(lr_sgemm_grouped_tn_impl(n, Ms, Ns, Ks, As, ldas, Bs, ldbs, Cs, ldcs, beta, shifts, periods, gws,
                                      w.gemm_bytes, stream));
      if (G == 3) {
        LR_TRY(lr_sgemm_impl(1, 0, 2 * Hd, Hd, B, 1.f, dG, L * ldg, h0_k, Hd, 1.f, gw_hh, Hd, nullptr, 0, 0, nullptr, 0,
                             stream));
        LR_TRY(lr_sgemm_impl(1, 0, Hd, Hd, B, 1.f, dG + 3 * Hd, L * ldg, h0_k, Hd, 1.f, gw_hh + (size_t)2 * Hd * Hd, Hd,
                             nullptr, 0, 0, nullptr, 0, stream));
      } else {
        LR_TRY(lr_sgemm_impl(1, 0, GH, Hd, B, 1.f, dG, L * ldg, h0_k, Hd, 1.f, gw_hh, Hd, nullptr, 0, 0, nullptr, 0,
                             stream));
      }
    }
    LR_TRY(lr_rnn_bias_grads(dG, colsum, gb_ih, gb_hh, BL, Hd, G, accumulate, stream));
    if (k > 0) {
      // ... and the gradient that reaches the layer below through it replaces dy
      LR_TRY(lr_sgemm_impl(0, 0, BL, Hd, GH, 1.f, dG, ldg, up->w_ih[k - 1], Hd, 0.f, dy, Hd, nullptr, 0, 0, gws,
                           w.gemm_bytes, stream));
      if (drop) {
        int gx_ = (int)(((int64_t)L * Hd / 4 + 255) / 256);
        if (gx_ > 64) gx_ = 64;
        LR_LAUNCH(dec_mask_rows_kernel, dim3(gx_, B), dim3(256), 0, stream, (const float*)dy,
                  drop + (size_t)(k - 1) * BL * Hd, dy, L, Hd, 0, L);
        LR_TRY(lr_launch_status());
      }
    }
  }
  if (!do_weights) return LR_OK;
  // (the loop ends on layer 0: dG now holds layer 0's gate gradients for the embedding path below)
  // embedding / W_ih through the table: dEW[v] = sum of the dG_x rows that used token v
  LR_LAUNCH(dec_scatter_dew_kernel, dim3(V, (GH + 255) / 256), dim3(256), 0, stream, (const float*)dG, ids,
            wb + w.dEW, BL, G, Hd);
  LR_TRY(lr_launch_status());
  if (g->emb_padding_idx >= 0 && g->emb_padding_idx < V) {
    // nn.Embedding(padding_idx): that row receives no gradient (its W_ih contribution is x = 0 anyway)
    LR_LAUNCH(zero_row_kernel, dim3(1), dim3(256), 0, stream, wb + w.dEW + (size_t)g->emb_padding_idx * GH, GH);
    LR_TRY(lr_launch_status());
  }
  LR_TRY(lr_sgemm_impl(0, 0, V, Cd, GH, 1.f, wb + w.dEW, GH, p->w_ih, Cd, beta, g->emb, Cd, nullptr, 0, 0, gws,
                       w.gemm_bytes, stream));
  // layer 0's W_ih (through the table), the output projection and the two halves of the concat layer: one
  // grouped launch + one combine
  {
    if (attn) LR_CHECK_ARG(g->w_c && g->b_c);
    int Ms[4] = {GH, V, Hd, Hd}, Ns[4] = {Cd, Hd, Hd, Hd}, Ks[4] = {V, BL, BL, BL};
    int ldas[4] = {GH, V, Hd, Hd}, ldbs[4] = {Cd, Hd, Hd, Hd}, ldcs[4] = {Cd, Hd, 2 * Hd, 2 * Hd};
    const float* As[4] = {wb + w.dEW, dlogits, dpre, dpre};
    const float* Bs[4] = {p->emb, nh, ctx, hs};
    float* Cs[4] = {g->w_ih, g->w_o, attn ? g->w_c : nullptr, attn ? g->w_c + Hd : nullptr};
    LR_TRY(lr_sgemm_grouped_tn_impl(attn ? 4 : 2, Ms, Ns, Ks, As, ldas, Bs, ldbs, Cs, ldcs, beta, nullptr, nullptr, gws,
                                    w.gemm_bytes, stream));
  }
  LR_TRY(colsum_into(dlogits, V, BL, V, colsum, g->b_o, accumulate, stream));
  if (attn) LR_TRY(colsum_into(dpre, Hd, BL, Hd, colsum, g->b_c, accumulate, stream));
  return LR_OK;
}

extern "C" int lr_decoder_backward(int mode, int attn_type, const lr_decoder_params* p, const lr_decoder_upper* up,
                                   const lr_decoder_grads* g, const lr_decoder_upper_grads* gup,
                                   const float* enc, const int32_t* enc_lens, const float* h0, const float* c0,
                                   const int32_t* step_lens, const float* log_probs, const float* d_log_probs,
                                   const float* dh_n, const float* dc_n, float* d_enc, float* dh0, float* dc0,
                                   const void* reserve, size_t reserve_bytes,
                                   void* workspace, size_t workspace_bytes, int accumulate, int B, int L, int T,
                                   int Hd, int Cd, int V, int A, lr_stream_t stream_) {
  return lr_decoder_backward_parts(mode, attn_type, p, up, g, gup, enc, enc_lens, h0, c0, step_lens, log_probs, d_log_probs,
                                   dh_n, dc_n, d_enc, dh0, dc0, reserve, reserve_bytes, workspace, workspace_bytes,
                                   accumulate, B, L, T, Hd, Cd, V, A, 3, stream_);
}
