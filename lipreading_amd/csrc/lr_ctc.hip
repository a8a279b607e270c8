// lr_ctc.hip — CTC alpha/beta forward-backward, the reference's batch reduction, and greedy
// decode for gfx950.
//
// Reference arithmetic replaced here:
//   src/train/ctc_loss.py:85      F.ctc_loss(log_probs(T,N,C), concat targets, blank=0)
//   src/train/ctc_loss.py:46-114  length filter, equal-length runs, inf fallback, weighting
//   src/models/lipreader/decoder.py:165-197  GreedyDecoder.process_string / decode
//
// Layout: the recursions run one workgroup per sample.  A sample's lattice is (T x C) fp32
// log-probs (19.5 kB at T=75, C=65), read ONCE from HBM with coalesced loads into LDS; its 2L+1
// CTC states live one per thread, alpha and beta advance concurrently on the two halves of the
// workgroup, the previous row is exchanged through a double-buffered LDS row (one s_barrier per
// time step) and both tables go to the caller's workspace (L2 resident).  The gradient rows are
// independent once alpha and beta are known, so they run one wave per (sample, frame) across the
// whole chip with lanes along the class axis (coalesced stores of the (T x C) gradient).
#include "lr_common.h"
#include <hip/hip_ext.h>

namespace {

constexpr int kMaxLabelLen = 256;  // src/train/ctc_loss.py:46 (CuDNN-era limit the reference keeps)

__host__ __device__ inline int ctc_state_stride(int max_label_len) {
  return ((2 * max_label_len + 1) + 63) / 64 * 64;
}

// workspace: alpha [B][T][sst] | beta [B][T][sst] | nxt int[B][Lmax] | first int[B][C]
struct CtcWs {
  float* alpha;
  float* beta;
  int* nxt;
  int* first;
};
__host__ __device__ inline size_t ctc_ws_floats(int B, int T, int C, int max_label_len) {
  const size_t sst = ctc_state_stride(max_label_len);
  return 2 * (size_t)B * T * sst + (size_t)B * max_label_len + (size_t)B * C;
}
inline CtcWs ctc_ws_carve(void* ws, int B, int T, int C, int max_label_len) {
  const size_t sst = ctc_state_stride(max_label_len);
  CtcWs w;
  w.alpha = (float*)ws;
  w.beta = w.alpha + (size_t)B * T * sst;
  w.nxt = (int*)(w.beta + (size_t)B * T * sst);
  w.first = w.nxt + (size_t)B * max_label_len;
  return w;
}

// ---------------------------------------------------------------------------------------
// alpha and beta recursions, concurrently, + nll + label tables
// ---------------------------------------------------------------------------------------
// One workgroup per sample.  Threads [0,sst) own the alpha state s = tid, threads [sst,2*sst)
// the beta state s = tid - sst (when 2*sst fits a workgroup; otherwise the two recursions run
// one after the other on the same threads).  The sample's (T x C) lattice is staged into LDS
// once with coalesced loads — the recursions then gather lp[t][l'_s] from LDS instead of paying
// one global round trip per time step.
template <bool LATTICE_IN_LDS>
__global__ void ctc_alpha_beta_kernel(const float* __restrict__ lp, int64_t stride_b,
                                      int64_t stride_t, const int32_t* __restrict__ labels,
                                      int label_stride, const int32_t* __restrict__ frame_lens,
                                      const int32_t* __restrict__ label_lens,
                                      float* __restrict__ nll, CtcWs ws, int T, int C, int sst,
                                      int max_label_len, int concurrent) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* rowa = reinterpret_cast<float*>(smem_raw);        // [2][sst] alpha exchange rows
  float* rowb = rowa + 2 * sst;                            // [2][sst] beta exchange rows
  int* lab_s = reinterpret_cast<int*>(rowb + 2 * sst);     // [max_label_len]
  float* lat = reinterpret_cast<float*>(lab_s + max_label_len);  // [T][C] when LATTICE_IN_LDS

  const int b = blockIdx.x;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int L = label_lens[b];
  int Tb = frame_lens[b];
  if (Tb > T) Tb = T;
  if (L > max_label_len || L < 0 || Tb <= 0) {
    // dropped by lr_ctc_reduce when L > 256; otherwise a caller error: poison the loss.
    if (tid == 0) nll[b] = __builtin_inff();
    return;
  }
  const int S = 2 * L + 1;
  const int32_t* lab = labels + (int64_t)b * label_stride;
  const float* lpb = lp + (int64_t)b * stride_b;
  float* al = ws.alpha + ((int64_t)b * T) * sst;
  float* be = ws.beta + ((int64_t)b * T) * sst;

  // ---- stage labels (+ lattice) --------------------------------------------------------------
  for (int i = tid; i < L; i += nthr) {
    int c = lab[i];
    if (c < 0 || c >= C) c = 0;  // out-of-range ids are rejected by the host; stay in bounds
    lab_s[i] = c;
  }
  for (int c = tid; c < C; c += nthr) ws.first[(int64_t)b * C + c] = -1;
  if (LATTICE_IN_LDS) {
    const int n = Tb * C;
    for (int i = tid; i < n; i += nthr) {
      const int t = i / C, c = i - t * C;
      lat[i] = lpb[(int64_t)t * stride_t + c];
    }
  }
  __syncthreads();
  // next / first occurrence chains per class (for the gradient kernel)
  for (int i = tid; i < L; i += nthr) {
    const int c = lab_s[i];
    int n = -1;
    for (int j = i + 1; j < L; ++j)
      if (lab_s[j] == c) { n = j; break; }
    ws.nxt[(int64_t)b * max_label_len + i] = n;
    bool first = true;
    for (int j = 0; j < i; ++j)
      if (lab_s[j] == c) { first = false; break; }
    if (first) ws.first[(int64_t)b * C + c] = i;
  }

  auto LP = [&](int t, int c) -> float {
    return LATTICE_IN_LDS ? lat[t * C + c] : lpb[(int64_t)t * stride_t + c];
  };

  const bool is_beta_half = concurrent && tid >= sst;
  const int s = is_beta_half ? tid - sst : tid;
  const bool live = s < S && s < sst;
  int cls = 0;
  bool skip_a = false, skip_b = false;
  if (live && (s & 1)) {
    cls = lab_s[s >> 1];
    if (s >= 3) skip_a = lab_s[(s >> 1) - 1] != cls;       // alpha_t(s) <- alpha_{t-1}(s-2)
    if (s + 2 < S) skip_b = lab_s[(s >> 1) + 1] != cls;    // beta_t(s)  <- beta_{t+1}(s+2)
  }

  // one pass = the alpha recursion (threads < sst) and, when concurrent, the beta recursion on
  // the other half at the same time; otherwise a second pass does beta.
  for (int pass = 0; pass < (concurrent ? 1 : 2); ++pass) {
    const bool do_beta = concurrent ? is_beta_half : (pass == 1);
    float* row0 = do_beta ? rowb : rowa;
    float v = LR_NEG_INF;
    if (!do_beta) {
      if (live && s < 2) v = LP(0, cls);
      if (live) al[s] = v;
    } else {
      if (live && s >= S - 2) v = LP(Tb - 1, cls);
      if (live) be[(int64_t)(Tb - 1) * sst + s] = v;
    }
    for (int i = 1; i < Tb; ++i) {
      const int t = do_beta ? Tb - 1 - i : i;
      float* row = row0 + (i & 1) * sst;
      if (s < sst) row[s] = v;
      const float lp_t = live ? LP(t, cls) : 0.f;
      lr_lds_barrier();   // LDS only: the alpha/beta rows streaming out to HBM stay in flight
      if (live) {
        if (!do_beta) {
          const float a2 = s >= 1 ? row[s - 1] : LR_NEG_INF;
          const float a3 = skip_a ? row[s - 2] : LR_NEG_INF;
          v = lr_lse3(v, a2, a3) + lp_t;
          al[(int64_t)t * sst + s] = v;
        } else {
          const float b2 = s + 1 < S ? row[s + 1] : LR_NEG_INF;
          const float b3 = skip_b ? row[s + 2] : LR_NEG_INF;
          v = lr_lse3(v, b2, b3) + lp_t;
          be[(int64_t)t * sst + s] = v;
        }
      }
    }
    if (!do_beta) {
      float* row = row0 + (Tb & 1) * sst;
      if (s < sst) row[s] = v;
    }
    __syncthreads();
    if (!do_beta && tid == 0) {
      const float* row = rowa + (Tb & 1) * sst;
      const float l1 = row[S - 1];
      const float l2 = S > 1 ? row[S - 2] : LR_NEG_INF;
      nll[b] = -lr_lse2(l1, l2);
    }
    __syncthreads();
  }
}

// log(exp(a)+exp(b)+exp(c)) with the fast exp/log forms (v_exp_f32 / v_log_f32 based, ~1-2 ulp):
// the recursion is a chain of T dependent evaluations and the libm sequences are several times
// longer; the parity tests (loss 1e-4, gradient 1e-4 of max) hold with them.
__device__ __forceinline__ float lse3_fast(float a, float b, float c) {
  float m = fmaxf(fmaxf(a, b), c);
  if (m == LR_NEG_INF) m = 0.f;
  return __logf(__expf(a - m) + __expf(b - m) + __expf(c - m)) + m;
}

// The same recursions when all 2L+1 states fit ONE wave (L <= 31: every shipped caption config at
// the bench's label length): wave 0 runs alpha, wave 1 beta, a state per lane, and the previous
// row's neighbours come from lane shuffles — no LDS exchange row and no barrier per time step
// (the T-step chain is then ~T x (2 shuffles + one log-sum-exp) instead of T barriers).
__device__ __forceinline__ void ctc_alpha_beta_wave_body(char* smem_raw, const float* __restrict__ lp, int64_t stride_b,
                                                         int64_t stride_t, const int32_t* __restrict__ labels,
                                                         int label_stride, const int32_t* __restrict__ frame_lens,
                                                         const int32_t* __restrict__ label_lens,
                                                         float* __restrict__ nll, CtcWs ws, int T, int C,
                                                         int max_label_len) {
  constexpr int sst = 64;
  int* lab_s = reinterpret_cast<int*>(smem_raw);                  // [32]
  float* lat = reinterpret_cast<float*>(lab_s + 32);              // [T][C]
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = label_lens[b];
  int Tb = frame_lens[b];
  if (Tb > T) Tb = T;
  if (L > max_label_len || L < 0 || L > 31 || Tb <= 0) {
    if (tid == 0) nll[b] = __builtin_inff();
    return;
  }
  const int S = 2 * L + 1;
  const int32_t* lab = labels + (int64_t)b * label_stride;
  const float* lpb = lp + (int64_t)b * stride_b;
  float* al = ws.alpha + ((int64_t)b * T) * sst;
  float* be = ws.beta + ((int64_t)b * T) * sst;
  for (int i = tid; i < L; i += 128) {
    int c = lab[i];
    if (c < 0 || c >= C) c = 0;
    lab_s[i] = c;
  }
  for (int c = tid; c < C; c += 128) ws.first[(int64_t)b * C + c] = -1;
  const int n = Tb * C;
  // The LDS image is shifted by the sample's misalignment (in floats) so that global and LDS 16-byte
  // boundaries coincide: the dense lattice then moves as 16-byte loads, all of a thread's loads in
  // flight at once (a scalar load -> store loop serialises ~38 memory round trips here).
  const int mis = stride_t == C ? (int)((reinterpret_cast<uintptr_t>(lpb) >> 2) & 3) : 0;
  float* latp = lat + mis;
  if (stride_t == C) {
    const float4* g4 = reinterpret_cast<const float4*>(lpb - mis);   // aligned; quads below qlo are not touched
    const int total = n + mis, qlo = (mis + 3) >> 2, qhi = total >> 2;
    for (int e = mis + tid; e < 4 * qlo && e < total; e += 128) lat[e] = lpb[e - mis];   // head
    for (int q0 = qlo; q0 < qhi; q0 += 128 * 10) {   // (qhi > qlo here, so qhi - 1 is a valid quad)
      float4 v4[10];
#pragma unroll
      for (int u = 0; u < 10; ++u) {   // unconditional (clamped) loads: a load under a branch is waited for
        const int q = q0 + u * 128 + tid;  // inside its branch, which would serialise the ten round trips
        v4[u] = g4[q < qhi ? q : qhi - 1];
      }
      // The values are only used under the `q < qhi` of the store loop below, so hipcc would SINK each load into its
      // store's branch after all — ten times {branch, global_load_dwordx4, vmcnt(0), ds_write_b128}, the serialised
      // round trips the clamped loads were written to avoid.  Two empty asm statements that take the registers as
      // operands keep five loads each in front of a single wait.  MEASURED (round 4): 23.6 -> 21.8 us (regime R).
#define LR_PIN4(u) "+v"(v4[u].x), "+v"(v4[u].y), "+v"(v4[u].z), "+v"(v4[u].w)
      asm volatile("" : LR_PIN4(0), LR_PIN4(1), LR_PIN4(2), LR_PIN4(3), LR_PIN4(4));
      asm volatile("" : LR_PIN4(5), LR_PIN4(6), LR_PIN4(7), LR_PIN4(8), LR_PIN4(9));
#undef LR_PIN4
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        const int q = q0 + u * 128 + tid;
        if (q < qhi) reinterpret_cast<float4*>(lat)[q] = v4[u];
      }
    }
    for (int e = (qhi > qlo ? 4 * qhi : 4 * qlo) + tid; e < total; e += 128) lat[e] = lpb[e - mis];   // tail
  } else {
    for (int i = tid; i < n; i += 128) {
      const int t = i / C, c = i - t * C;
      latp[i] = lpb[(int64_t)t * stride_t + c];
    }
  }
  __syncthreads();
  for (int i = tid; i < L; i += 128) {   // next / first occurrence chains per class (gradient kernel)
    const int c = lab_s[i];
    int nx = -1;
    for (int j = i + 1; j < L; ++j)
      if (lab_s[j] == c) { nx = j; break; }
    ws.nxt[(int64_t)b * max_label_len + i] = nx;
    bool first = true;
    for (int j = 0; j < i; ++j)
      if (lab_s[j] == c) { first = false; break; }
    if (first) ws.first[(int64_t)b * C + c] = i;
  }
  const int s = lane;
  const bool live = s < S;
  int cls = 0;
  bool skip_a = false, skip_b = false;
  if (live && (s & 1)) {
    cls = lab_s[s >> 1];
    if (s >= 3) skip_a = lab_s[(s >> 1) - 1] != cls;
    if (s + 2 < S) skip_b = lab_s[(s >> 1) + 1] != cls;
  }
  float v = LR_NEG_INF;
  if (wave == 0) {
    if (live && s < 2) v = latp[cls];
    if (live) al[s] = v;
    for (int t = 1; t < Tb; ++t) {
      const float lp_t = latp[t * C + cls];
      const float up1 = __shfl_up(v, 1, 64), up2 = __shfl_up(v, 2, 64);
      const float a2 = s >= 1 ? up1 : LR_NEG_INF;                          // alpha_{t-1}(s-1)
      const float a3 = skip_a ? up2 : LR_NEG_INF;                          // alpha_{t-1}(s-2)
      v = live ? lse3_fast(v, a2, a3) + lp_t : LR_NEG_INF;
      if (live) al[(int64_t)t * sst + s] = v;
    }
    const float l1 = __shfl(v, S - 1, 64);
    const float l2 = S > 1 ? __shfl(v, S - 2, 64) : LR_NEG_INF;
    if (lane == 0) nll[b] = -lr_lse2(l1, l2);
  } else {
    if (live && s >= S - 2) v = latp[(Tb - 1) * C + cls];
    if (live) be[(int64_t)(Tb - 1) * sst + s] = v;
    for (int t = Tb - 2; t >= 0; --t) {
      const float lp_t = latp[t * C + cls];
      const float dn1 = __shfl_down(v, 1, 64), dn2 = __shfl_down(v, 2, 64);
      const float b2 = s + 1 < S ? dn1 : LR_NEG_INF;                     // beta_{t+1}(s+1)
      const float b3 = skip_b ? dn2 : LR_NEG_INF;                         // beta_{t+1}(s+2)
      v = live ? lse3_fast(v, b2, b3) + lp_t : LR_NEG_INF;
      if (live) be[(int64_t)t * sst + s] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------
// gradient rows: one wave per (sample, frame), lanes along the class axis
// ---------------------------------------------------------------------------------------
__global__ void ctc_grad_rows_kernel(const float* __restrict__ lp, int64_t stride_b,
                                     int64_t stride_t, const int32_t* __restrict__ frame_lens,
                                     const int32_t* __restrict__ label_lens,
                                     const float* __restrict__ nll,
                                     const float* __restrict__ grad_weight,
                                     const float* __restrict__ grad_scale,
                                     float* __restrict__ grad, CtcWs ws, int T, int C, int sst,
                                     int max_label_len) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= T) return;
  const int L = label_lens[b];
  int Tb = frame_lens[b];
  if (Tb > T) Tb = T;
  const float w = (grad_weight ? grad_weight[b] : 1.f) * (grad_scale ? grad_scale[0] : 1.f);
  const float nll_b = nll[b];
  float* grow = grad + (int64_t)b * stride_b + (int64_t)t * stride_t;
  const bool dead = (L > max_label_len) || L < 0 || Tb <= 0 || w == 0.f || isinf(nll_b) ||
                    isnan(nll_b);
  if (dead || t >= Tb) {
    for (int c = lane; c < C; c += 64) grow[c] = 0.f;
    return;
  }
  const int S = 2 * L + 1;
  const float* ar = ws.alpha + ((int64_t)b * T + t) * sst;
  const float* br = ws.beta + ((int64_t)b * T + t) * sst;
  const float* lpr = lp + (int64_t)b * stride_b + (int64_t)t * stride_t;
  const int* nxt = ws.nxt + (int64_t)b * max_label_len;
  const int* first = ws.first + (int64_t)b * C;
  // blank: log-sum-exp over the even states
  float m = LR_NEG_INF;
  for (int s = 2 * lane; s < S; s += 128) m = fmaxf(m, ar[s] + br[s]);
  m = lr_wave_max(m);
  float blank_lcab = LR_NEG_INF;
  if (m != LR_NEG_INF) {
    float acc = 0.f;
    for (int s = 2 * lane; s < S; s += 128) acc += expf(ar[s] + br[s] - m);
    acc = lr_wave_sum(acc);
    blank_lcab = logf(acc) + m;
  }
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    if (c < C) {
      float v;
      if (c == 0) {
        v = blank_lcab;
      } else {
        v = LR_NEG_INF;
        int i = first[c];
        if (i >= 0) {
          v = ar[2 * i + 1] + br[2 * i + 1];
          for (i = nxt[i]; i >= 0; i = nxt[i]) v = lr_lse2(v, ar[2 * i + 1] + br[2 * i + 1]);
        }
      }
      const float l = lpr[c];
      // torch ctc_loss backward: (exp(lp) - exp(lcab + nll - lp)) * grad_out
      grow[c] = w * (expf(l) - expf(v + nll_b - l));
    }
  }
}

// ---------------------------------------------------------------------------------------
// the reference's batch reduction, on the device
// ---------------------------------------------------------------------------------------
constexpr int kReduceMaxB = 2048;

// (the body of the batch reduction; s_nll / s_fl / s_ll / s_w: B-entry LDS arrays of the caller.  nll is read with
// agent-scope atomic loads: in the fused form below other workgroups of the same launch wrote it)
__device__ __forceinline__ void ctc_reduce_body(const float* nll, const int32_t* __restrict__ frame_lens,
                                                const int32_t* __restrict__ label_lens, int reduction,
                                                float* __restrict__ out_loss, int32_t* __restrict__ out_status,
                                                float* __restrict__ grad_weight, int B, const int32_t* __restrict__ fault,
                                                float* s_nll, int* s_fl, int* s_ll, float* s_w) {
  // fault[0] != 0: a one-launch recurrence that produced these log-probs gave up waiting for a partner workgroup
  // (lr_common.h lr_fault_words): the numbers are garbage, so the batch is reported like one the reference
  // skips (loss 0, status 2, zero gradient) — train_better_model.py:49-50.
  if (fault && fault[0] != 0) {
    if (threadIdx.x == 0) {
      out_loss[0] = 0.f;
      out_status[0] = 2;
    }
    for (int i = threadIdx.x; i < B; i += blockDim.x) grad_weight[i] = 0.f;
    return;
  }
  // compact away the label_len > 256 samples (ctc_loss.py:46-56) while staging into LDS;
  // done serially by thread 0 after a parallel copy so the order is preserved.
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    s_nll[i] = __hip_atomic_load(nll + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_fl[i] = frame_lens[i];
    s_ll[i] = label_lens[i];
    s_w[i] = 0.f;
  }
  __shared__ int s_plain;
  if (threadIdx.x == 0) s_plain = 1;
  __syncthreads();
  // The common batch — every label fits, one frame length, no infinite loss — is ONE run of the reference's
  // loop (ctc_loss.py:64-105): its terms are formed by all threads and only the ordered sum is serial
  // (the general scan below costs 14 us at B = 32 on the critical path between the CTC passes; this 3).
  for (int i = threadIdx.x; i < B; i += blockDim.x)
    if (s_ll[i] > kMaxLabelLen || s_fl[i] != s_fl[0] || isinf(s_nll[i])) s_plain = 0;
  __syncthreads();
  if (s_plain) {
    const float fb = (float)B;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
      const float l = (float)(s_ll[i] < 1 ? 1 : s_ll[i]);
      if (reduction == LR_CTC_MEAN) {
        s_w[i] = fb / (fb * l);      // mb / (m * l) with mb = m = B, as the general path forms it
        s_nll[i] = s_nll[i] / l;
      } else {
        s_w[i] = 1.f;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float run = 0.f;
      for (int k = 0; k < B; ++k) run += s_nll[k];
      float total = run, inv = 1.f;
      if (reduction == LR_CTC_MEAN) total = run / fb * fb;
      const bool none = total == 0.f;   // ctc_loss.py:110-112
      float loss = total;
      if (!none && reduction == LR_CTC_MEAN) {
        inv = 1.f / fb;
        loss = total / fb;
      }
      out_loss[0] = none ? 0.f : loss;
      out_status[0] = none ? 1 : 0;
      s_nll[0] = none ? 0.f : inv;
    }
    __syncthreads();
    const float inv = s_nll[0];
    for (int i = threadIdx.x; i < B; i += blockDim.x) grad_weight[i] = s_w[i] * inv;
    return;
  }
  if (threadIdx.x == 0) {
    // kept list = indices with label_len <= 256; walk it through an index array stored in
    // place of s_fl's upper half is not possible at B = max, so re-scan with a cursor.
    int n = 0;
    for (int i = 0; i < B; ++i) n += (s_ll[i] <= kMaxLabelLen);
    float total = 0.f;
    float count = 0.f;
    bool any = false;
    if (n > 0) {
      // positions below are positions in the kept list; kept(i) maps through a cursor.
      // Run boundaries (change points, ctc_loss.py:64-65) are detected on the kept list.
      int cur_len = n;      // len(frame_lens) as the reference's loop sees it (:74)
      int prev_i = -1;      // raw index of the first kept sample of the pending slice
      // first kept raw index
      int i = 0;
      while (i < B && s_ll[i] > kMaxLabelLen) ++i;
      prev_i = i;
      int slice_cnt = 0;    // kept samples in [prev_i, cursor)
      int cursor = i;
      while (cursor < B) {
        // advance cursor over one run of equal frame_len (kept samples only)
        const int fl = s_fl[cursor];
        int j = cursor;
        int last_kept = cursor;
        while (j < B) {
          if (s_ll[j] <= kMaxLabelLen) {
            if (s_fl[j] != fl) break;
            ++slice_cnt;
            last_kept = j;
          }
          ++j;
        }
        cursor = j;  // first kept sample of the next run, or B
        // slice = kept samples in [prev_i, cursor)
        int mb = cur_len;
        cur_len = slice_cnt;
        int m = 0;
        bool has_inf = false;
        for (int k = prev_i; k < cursor; ++k)
          if (s_ll[k] <= kMaxLabelLen) {
            if (isinf(s_nll[k])) has_inf = true; else ++m;
          }
        if (has_inf) {
          if (m == 0) continue;  // ctc_loss.py:92 — prev_change_point NOT advanced
          cur_len = m;
          mb = m;
        } else {
          m = slice_cnt;
        }
        float run = 0.f;
        if (reduction == LR_CTC_MEAN) {
          for (int k = prev_i; k < cursor; ++k)
            if (s_ll[k] <= kMaxLabelLen && !isinf(s_nll[k])) {
              const float l = (float)(s_ll[k] < 1 ? 1 : s_ll[k]);
              run += s_nll[k] / l;
              s_w[k] = (float)mb / ((float)m * l);
            }
          run = run / (float)m * (float)mb;
          count += (float)mb;
        } else {
          for (int k = prev_i; k < cursor; ++k)
            if (s_ll[k] <= kMaxLabelLen && !isinf(s_nll[k])) {
              run += s_nll[k];
              s_w[k] = 1.f;
            }
        }
        total += run;
        any = true;
        prev_i = cursor;
        slice_cnt = 0;
        (void)last_kept;
      }
    }
    const bool none = !any || total == 0.f;  // ctc_loss.py:110-112
    float inv = 1.f;
    float loss = total;
    if (!none && reduction == LR_CTC_MEAN) {
      inv = 1.f / count;
      loss = total / count;
    }
    out_loss[0] = none ? 0.f : loss;
    out_status[0] = none ? 1 : 0;
    s_nll[0] = none ? 0.f : inv;  // broadcast slot
  }
  __syncthreads();
  const float inv = s_nll[0];
  for (int i = threadIdx.x; i < B; i += blockDim.x) grad_weight[i] = s_w[i] * inv;
}

__global__ void ctc_reduce_kernel(const float* __restrict__ nll,
                                  const int32_t* __restrict__ frame_lens,
                                  const int32_t* __restrict__ label_lens, int reduction,
                                  float* __restrict__ out_loss, int32_t* __restrict__ out_status,
                                  float* __restrict__ grad_weight, int B, const int32_t* __restrict__ fault) {
  __shared__ float s_nll[kReduceMaxB];
  __shared__ int s_fl[kReduceMaxB];
  __shared__ int s_ll[kReduceMaxB];
  __shared__ float s_w[kReduceMaxB];
  ctc_reduce_body(nll, frame_lens, label_lens, reduction, out_loss, out_status, grad_weight, B, fault, s_nll, s_fl, s_ll, s_w);
}

// What the batch reduction writes, riding the alpha/beta launch (lr_ctc_nll_reduce): out_loss == NULL: no rider.
struct CtcTail {
  int reduction;
  float* out_loss;
  int32_t* out_status;
  float* grad_weight;
  const int32_t* fault;
};
// workgroups of the current alpha/beta launch that have stored their sample's nll; the last one resets it.  The word
// is the THIRD of the device's fault words (lr_common.h lr_fault_words: {pending, total, ctc ticket, spare}), which
// lr_step_begin / lr_step_begin_ctc put back to 0 at the top of every step: a launch that was torn down half-way
// cannot leave a count behind that no later launch would ever complete (round 5 kept it in a __device__ global that
// nothing reset).  One word per device: launches of lr_ctc_nll_reduce must not overlap in time (launches on one
// stream never do; a caller with two losses in flight on two streams uses lr_ctc_nll + lr_ctc_reduce).
// g_ctc_ticket: only where the fault words could not be allocated.
__device__ unsigned g_ctc_ticket = 0u;

// One workgroup per sample as above; with a rider (round 5) the LAST workgroup to finish runs the reference's batch
// reduction (ctc_loss.py:64-112) in the same launch: the 5 us ctc_reduce launch between the recursions and the gradient
// rows leaves a training step's critical path.  Its B-entry arrays lie over the lattice image, which is dead by then.
__global__ __launch_bounds__(128) void ctc_alpha_beta_wave_kernel(const float* __restrict__ lp, int64_t stride_b,
                                                                  int64_t stride_t,
                                                                  const int32_t* __restrict__ labels,
                                                                  int label_stride,
                                                                  const int32_t* __restrict__ frame_lens,
                                                                  const int32_t* __restrict__ label_lens,
                                                                  float* __restrict__ nll, CtcWs ws, int T, int C,
                                                                  int max_label_len, CtcTail tail) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  ctc_alpha_beta_wave_body(smem_raw, lp, stride_b, stride_t, labels, label_stride, frame_lens, label_lens, nll, ws, T, C,
                           max_label_len);
  if (!tail.out_loss) return;
  __shared__ int s_last;
  __syncthreads();                      // (every exit of the body is workgroup-uniform; thread 0 stored nll[b])
  if (threadIdx.x == 0) {
    __threadfence();
    unsigned* ticket = tail.fault ? reinterpret_cast<unsigned*>(const_cast<int32_t*>(tail.fault) + 2) : &g_ctc_ticket;
    const bool last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    if (last) atomicExch(ticket, 0u);
    s_last = last ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const int B = (int)gridDim.x;
  float* s_nll = reinterpret_cast<float*>(smem_raw);
  int* s_fl = reinterpret_cast<int*>(s_nll + B);
  int* s_ll = s_fl + B;
  float* s_w = reinterpret_cast<float*>(s_ll + B);
  ctc_reduce_body(nll, frame_lens, label_lens, tail.reduction, tail.out_loss, tail.out_status, tail.grad_weight, B,
                  tail.fault, s_nll, s_fl, s_ll, s_w);
}

// ---------------------------------------------------------------------------------------
// greedy decode
// ---------------------------------------------------------------------------------------
__global__ void ctc_greedy_kernel(const float* __restrict__ probs, int64_t stride_b,
                                  int64_t stride_t, const int32_t* __restrict__ sizes,
                                  const int32_t* __restrict__ class_map,
                                  int32_t* __restrict__ out_ids, int32_t* __restrict__ out_off,
                                  int32_t* __restrict__ out_lens, int T, int C, int blank) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int* amax = reinterpret_cast<int*>(smem_raw);  // [T]
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nwave = blockDim.x >> 6;
  int n = sizes ? sizes[b] : T;
  if (n > T) n = T;
  if (n < 0) n = 0;
  const float* pb = probs + (int64_t)b * stride_b;
  // argmax per frame: lanes along the class axis, first maximum wins (torch.max on CPU).
  for (int t = wave; t < n; t += nwave) {
    const float* pr = pb + (int64_t)t * stride_t;
    float best = LR_NEG_INF;
    int bi = C;  // sentinel larger than any index
    for (int c = lane; c < C; c += 64) {
      const float v = pr[c];
      if (bi == C || v > best) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) amax[t] = class_map ? class_map[bi] : bi;
  }
  __syncthreads();
  // collapse: ordered compaction by wave 0, 64 frames per pass.
  if (wave == 0) {
    int base = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
      const int t = t0 + lane;
      bool keep = false;
      int id = -1;
      if (t < n) {
        id = amax[t];
        keep = (id != blank) && (t == 0 || id != amax[t - 1]);
      }
      const unsigned long long mask = __ballot(keep);
      const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
      if (keep) {
        out_ids[(int64_t)b * T + pos] = id;
        out_off[(int64_t)b * T + pos] = t;
      }
      base += __popcll(mask);
    }
    for (int p = base + lane; p < T; p += 64) {
      out_ids[(int64_t)b * T + p] = -1;
      out_off[(int64_t)b * T + p] = -1;
    }
    if (lane == 0) out_lens[b] = base;
  }
}

}  // namespace

extern "C" size_t lr_ctc_workspace_bytes(int B, int T, int C, int max_label_len) {
  if (B <= 0 || T <= 0 || C <= 0 || max_label_len < 0) return 0;
  if (max_label_len > kMaxLabelLen) max_label_len = kMaxLabelLen;
  return ctc_ws_floats(B, T, C, max_label_len) * sizeof(float);
}

namespace {
// *fused (may be NULL): set to 1 when the launch took the rider `tail` along (the one-wave kernel), else 0
int ctc_nll_impl(const float* log_probs, int64_t stride_b, int64_t stride_t,
                          const int32_t* labels, int label_stride, const int32_t* frame_lens,
                          const int32_t* label_lens, float* nll, void* workspace,
                          size_t workspace_bytes, int B, int T, int C, int max_label_len,
                          lr_stream_t stream, CtcTail tail, int* fused) {
  if (fused) *fused = 0;
  LR_CHECK_ARG(log_probs && labels && frame_lens && label_lens && nll && workspace);
  LR_CHECK_ARG(B > 0 && T > 0 && C > 0 && max_label_len >= 0 && label_stride >= 0);
  if (max_label_len > kMaxLabelLen) max_label_len = kMaxLabelLen;
  if (workspace_bytes < lr_ctc_workspace_bytes(B, T, C, max_label_len)) return LR_ERR_WORKSPACE;
  const int sst = ctc_state_stride(max_label_len);
  const CtcWs ws = ctc_ws_carve(workspace, B, T, C, max_label_len);
  const int concurrent = 2 * sst <= 1024 ? 1 : 0;
  const int nthr = concurrent ? 2 * sst : sst;
  const size_t base = 4 * (size_t)sst * sizeof(float) + (size_t)max_label_len * sizeof(int);
  const size_t lat = (size_t)T * C * sizeof(float);
  if (sst == 64 && max_label_len <= 31 && 32 * sizeof(int) + lat + 16 <= 60 * 1024) {
    // every state fits one wave: shuffle recursion, no per-step barrier
    size_t lds = 32 * sizeof(int) + lat + 16;
    if (tail.out_loss && (size_t)B * 16 <= 60 * 1024) {
      if (lds < (size_t)B * 16) lds = (size_t)B * 16;
      if (fused) *fused = 1;
    } else {
      tail.out_loss = nullptr;
    }
    LR_LAUNCH_PROF(LR_PROF_CTC_ALPHA_BETA, ctc_alpha_beta_wave_kernel, dim3(B), dim3(128), lds, stream, log_probs, stride_b,
              stride_t, labels, label_stride, frame_lens, label_lens, nll, ws, T, C, max_label_len, tail);
  } else if (base + lat <= 60 * 1024) {
    LR_LAUNCH_PROF(LR_PROF_CTC_ALPHA_BETA, ctc_alpha_beta_kernel<true>, dim3(B), dim3(nthr), base + lat, stream, log_probs,
              stride_b, stride_t, labels, label_stride, frame_lens, label_lens, nll, ws, T, C, sst,
              max_label_len, concurrent);
  } else {
    LR_LAUNCH_PROF(LR_PROF_CTC_ALPHA_BETA, ctc_alpha_beta_kernel<false>, dim3(B), dim3(nthr), base, stream, log_probs, stride_b,
              stride_t, labels, label_stride, frame_lens, label_lens, nll, ws, T, C, sst,
              max_label_len, concurrent);
  }
  return lr_launch_status();
}
}  // namespace

extern "C" int lr_ctc_nll(const float* log_probs, int64_t stride_b, int64_t stride_t,
                          const int32_t* labels, int label_stride, const int32_t* frame_lens,
                          const int32_t* label_lens, float* nll, void* workspace,
                          size_t workspace_bytes, int B, int T, int C, int max_label_len,
                          lr_stream_t stream) {
  CtcTail none = {0, nullptr, nullptr, nullptr, nullptr};
  return ctc_nll_impl(log_probs, stride_b, stride_t, labels, label_stride, frame_lens, label_lens, nll, workspace,
                      workspace_bytes, B, T, C, max_label_len, stream, none, nullptr);
}

extern "C" int lr_ctc_reduce(const float* nll, const int32_t* frame_lens,
                             const int32_t* label_lens, int reduction, float* out_loss,
                             int32_t* out_status, float* grad_weight, int B, lr_stream_t stream);

// lr_ctc_nll then lr_ctc_reduce, as ONE launch where the one-wave recursion kernel applies (labels of <= 31 characters:
// every shipped caption configuration), two otherwise
extern "C" int lr_ctc_nll_reduce(const float* log_probs, int64_t stride_b, int64_t stride_t,
                                 const int32_t* labels, int label_stride, const int32_t* frame_lens,
                                 const int32_t* label_lens, float* nll, void* workspace, size_t workspace_bytes,
                                 int reduction, float* out_loss, int32_t* out_status, float* grad_weight, int B, int T,
                                 int C, int max_label_len, lr_stream_t stream) {
  LR_CHECK_ARG(out_loss && out_status && grad_weight);
  LR_CHECK_ARG(reduction == LR_CTC_SUM || reduction == LR_CTC_MEAN);
  if (B > kReduceMaxB) return LR_ERR_UNSUPPORTED;
  CtcTail tail = {reduction, out_loss, out_status, grad_weight, (const int32_t*)lr_fault_words()};
  int fused = 0;
  int st = ctc_nll_impl(log_probs, stride_b, stride_t, labels, label_stride, frame_lens, label_lens, nll, workspace,
                        workspace_bytes, B, T, C, max_label_len, stream, tail, &fused);
  if (st != LR_OK || fused) return st;
  return lr_ctc_reduce(nll, frame_lens, label_lens, reduction, out_loss, out_status, grad_weight, B, stream);
}

extern "C" int lr_ctc_grad_scaled(const float* log_probs, int64_t stride_b, int64_t stride_t,
                                  const int32_t* labels, int label_stride, const int32_t* frame_lens,
                                  const int32_t* label_lens, const float* nll, const float* grad_weight,
                                  const float* grad_scale, float* grad, void* workspace, size_t workspace_bytes,
                                  int B, int T, int C, int max_label_len, lr_stream_t stream) {
  LR_CHECK_ARG(log_probs && labels && frame_lens && label_lens && nll && grad && workspace);
  LR_CHECK_ARG(B > 0 && T > 0 && C > 0 && max_label_len >= 0 && label_stride >= 0);
  if (max_label_len > kMaxLabelLen) max_label_len = kMaxLabelLen;
  if (workspace_bytes < lr_ctc_workspace_bytes(B, T, C, max_label_len)) return LR_ERR_WORKSPACE;
  const int sst = ctc_state_stride(max_label_len);
  const CtcWs ws = ctc_ws_carve(workspace, B, T, C, max_label_len);
  LR_LAUNCH_PROF(LR_PROF_CTC_GRAD, ctc_grad_rows_kernel, dim3((T + 3) / 4, B), dim3(256), 0, stream, log_probs, stride_b,
            stride_t, frame_lens, label_lens, nll, grad_weight, grad_scale, grad, ws, T, C, sst, max_label_len);
  return lr_launch_status();
}

extern "C" int lr_ctc_grad(const float* log_probs, int64_t stride_b, int64_t stride_t,
                           const int32_t* labels, int label_stride, const int32_t* frame_lens,
                           const int32_t* label_lens, const float* nll, const float* grad_weight,
                           float* grad, void* workspace, size_t workspace_bytes, int B, int T,
                           int C, int max_label_len, lr_stream_t stream) {
  return lr_ctc_grad_scaled(log_probs, stride_b, stride_t, labels, label_stride, frame_lens, label_lens, nll,
                            grad_weight, nullptr, grad, workspace, workspace_bytes, B, T, C, max_label_len, stream);
}

extern "C" int lr_ctc_reduce(const float* nll, const int32_t* frame_lens,
                             const int32_t* label_lens, int reduction, float* out_loss,
                             int32_t* out_status, float* grad_weight, int B, lr_stream_t stream) {
  LR_CHECK_ARG(nll && frame_lens && label_lens && out_loss && out_status && grad_weight);
  LR_CHECK_ARG(B > 0 && (reduction == LR_CTC_SUM || reduction == LR_CTC_MEAN));
  if (B > kReduceMaxB) return LR_ERR_UNSUPPORTED;
  LR_LAUNCH(ctc_reduce_kernel, dim3(1), dim3(256), 0, stream, nll,
                     frame_lens, label_lens, reduction, out_loss, out_status, grad_weight, B,
                     (const int32_t*)lr_fault_words());
  return lr_launch_status();
}

extern "C" int lr_ctc_greedy_decode(const float* probs, int64_t stride_b, int64_t stride_t,
                                    const int32_t* sizes, const int32_t* class_map,
                                    int32_t* out_ids, int32_t* out_offsets, int32_t* out_lens,
                                    int B, int T, int C, int blank, lr_stream_t stream) {
  LR_CHECK_ARG(probs && out_ids && out_offsets && out_lens);
  LR_CHECK_ARG(B > 0 && T > 0 && C > 0);
  const size_t lds = (size_t)T * sizeof(int);
  if (lds > 64 * 1024) return LR_ERR_UNSUPPORTED;
  LR_LAUNCH(ctc_greedy_kernel, dim3(B), dim3(256), lds, stream, probs,
                     stride_b, stride_t, sizes, class_map, out_ids, out_offsets, out_lens, T, C, blank);
  return lr_launch_status();
}
