// lr_rnn.hip — (bi)directional GRU / LSTM layer with torch packed-sequence semantics on gfx950.
//
// Reference arithmetic replaced here (paths under the reference root):
//   src/models/lipreader/better_model.py:47-49   self.rnn = nn.{LSTM,GRU}(..., batch_first=True)
//   src/models/lipreader/better_model.py:64-78   sort -> pack_padded_sequence -> rnn -> pad_packed
//   src/models/lipreader/better_model.py:84-89   undo the sort (not needed here: nothing is sorted)
// Cell equations (torch.nn.GRU / LSTM, gate order r,z,n / i,f,g,o):
//   GRU : r=s(Wir x+bir+Whr h+bhr)  z=s(...)  n=tanh(Win x+bin + r*(Whn h+bhn))  h'=(1-z)n+z h
//   LSTM: i,f,o=s(...)  g=tanh(...)  c'=f c+i g  h'=o tanh(c')
//
// Structure (MI355X-first, not a translation of torch's cuDNN/MIOpen path):
//   1. ONE fp32-MFMA GEMM per direction computes the input projection for all T frames
//      (lr_gemm.hip), biases folded in, straight into the gate buffer.
//   2. The T-step recurrence is a chain of small (B x H)·(H x G*H) products.  Each step is one
//      launch covering BOTH directions; a workgroup owns a 16(batch) x 16(hidden unit) tile of all
//      G gates, its 4 waves split the K=H reduction, operands go global/L2 -> VGPR as float4
//      (W_hh rows and h rows are both K-contiguous, so no LDS staging is needed for the MFMA
//      fragments), partial tiles are combined through LDS and the gate non-linearities are fused
//      into the same kernel.  W_hh (0.75 MiB/dir at GRU-256, 9 MiB/dir at LSTM-768) stays resident
//      in the per-XCD L2s across steps because a workgroup's tile -> XCD mapping is the same at
//      every step.  A kernel boundary per step is the cheapest grid-wide hand-off on this chip
//      (MI355X_MICROARCH.md price list: boundary ~1.5 us vs 4-5 us for an in-kernel grid barrier).
//   3. The previous hidden state is read directly from y[b, t-1] (t+1 for the reverse direction):
//      positions past a sample's length hold zeros, which is exactly the packed-sequence
//      semantics (the reverse direction starts from zero state at each sample's own last frame).
//   4. Backward: the same chain in reverse with dG·W_hh as the per-step product (W_hh transposed
//      once per call so its rows are again K-contiguous), then one GEMM each for dW_ih, dW_hh
//      (with the "previous hidden state" row-shift view of y), dx, and a column-sum for the biases.
#include <mutex>

#include "lr_common.h"
#include <hip/hip_ext.h>

int lr_sgemm_impl(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                  const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                  int row_shift, int period, void* workspace, size_t workspace_bytes,
                  hipStream_t stream);
extern "C" size_t lr_sgemm_workspace_bytes(int M, int N, int K);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = 16;      // batch rows and hidden units per workgroup (MFMA 16x16x4)
constexpr int RED_LD = 17;    // padded row of the cross-wave reduction buffer

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Fragment-major ("packed") operands.  An MFMA 16x16x4 f32 operand fragment for 16 rows x 16 k
// is 64 lanes x float4: lane l holds row (l & 15), k = 4*(l >> 4) + {0,1,2,3}.  Storing operands
// in exactly that order makes every operand load of the recurrence ONE fully coalesced 1 KiB
// read per wave instruction (instead of 16 row segments of 64 B whose row stride H*4 B lands on a
// quarter of the memory channels):
//   Wp  [unit tile][gate][chunk][64 lanes][4]     packed once per call from W_hh
//   hp  [parity][dir][batch tile][chunk][64][4]   written by each step's epilogue for the next
// Rows/k past B or H are zero (hp is memset per call, Wp is zero-filled by the pack kernel).
struct StepPtrs {
  const float* w[2];   // per direction: packed W_hh (forward) / packed W_hh^T (backward)
  const float* b[2];   // per direction: b_hh [G*H]
  // optional initial state [D][B][H] (decoder: the encoder's final state, better_model.py:181):
  // step 0 then reads its "previous" h / c from here (and the packed copy already sits in hp).
  const float* h0;
  const float* c0;
};
constexpr int FRAG = 256;  // floats per 16x16 operand fragment

// ---------------------------------------------------------------------------------------------
// step kernels
// ---------------------------------------------------------------------------------------------
// A step is latency-bound, not FLOP-bound: after a kernel boundary every operand comes from the
// Infinity Cache / remote L2 (~1 us per dependent round trip), so the kernels are written to
// make exactly ONE round trip: every wave issues all of its float4 operand loads for the
// product up front (NW waves x UF chunks x (1+G) loads in flight), the 256 epilogue threads
// issue their gate / length / previous-state loads before the product starts, and only then do
// the MFMAs, the LDS combine and the stores run.
constexpr int NW = 8;        // waves per workgroup (512 threads): K is split NW ways
constexpr int UF_FWD = 6;    // chunks (16 k each) a wave keeps in flight, forward
constexpr int UF_BWD_GRU = 12;   // backward has one accumulator and 2 loads per fragment:
constexpr int UF_BWD_LSTM = 12;  // (24 in flight measured slower: 11.5 vs 10.6 us at LSTM-768)

#define LR_MFMA4(accv, av, wv)                                                 \
  accv = __builtin_amdgcn_mfma_f32_16x16x4f32((av).x, (wv).x, accv, 0, 0, 0); \
  accv = __builtin_amdgcn_mfma_f32_16x16x4f32((av).y, (wv).y, accv, 0, 0, 0); \
  accv = __builtin_amdgcn_mfma_f32_16x16x4f32((av).z, (wv).z, accv, 0, 0, 0); \
  accv = __builtin_amdgcn_mfma_f32_16x16x4f32((av).w, (wv).w, accv, 0, 0, 0)

template <int G>
__global__ __launch_bounds__(NW * 64) void rnn_fwd_step_kernel(float* gates, float* extra, float* y,
                                                              float* hp,
                                                              const int32_t* __restrict__ lens,
                                                              StepPtrs p, int B, int T, int H, int D,
                                                              int step) {
  __shared__ float red[NW * G * TILE * RED_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = blockIdx.z;
  const int t = d == 0 ? step : T - 1 - step;
  const int tp = d == 0 ? t - 1 : t + 1;
  const bool in_seq = tp >= 0 && tp < T;
  const bool has_prev = in_seq || (p.h0 != nullptr && step == 0);
  const int j0 = blockIdx.x * TILE, b0 = blockIdx.y * TILE;
  const int DH = D * H;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- epilogue operands first (threads 0..255 own one (batch row, hidden unit) each) -------
  const int bl = (tid >> 4) & 15, jl = tid & 15;
  const int b = b0 + bl, j = j0 + jl;
  const bool epi = tid < 256 && b < B && j < H;
  const int64_t bt = (int64_t)b * T + t, btp = (int64_t)b * T + tp;
  float gx[G];
  float prev_own = 0.f, bhn = 0.f;
  int len_b = 0;
  float* go = gates + (bt * D + d) * (int64_t)(G * H) + j;
  if (epi) {
    len_b = lens[b];
#pragma unroll
    for (int g = 0; g < G; ++g) gx[g] = go[g * H];
    if (G == 1) {
      // tanh RNN: the previous state only enters through the recurrent product
    } else if (G == 3) {
      bhn = p.b[d][2 * H + j];
      if (in_seq) prev_own = y[btp * DH + d * H + j];              // h_{t-1}
      else if (has_prev) prev_own = p.h0[((int64_t)d * B + b) * H + j];
    } else {
      if (in_seq) prev_own = extra[(btp * D + d) * H + j];         // c_{t-1}
      else if (has_prev && p.c0) prev_own = p.c0[((int64_t)d * B + b) * H + j];
    }
  }

  // ---- recurrent product: acc[g] (16 batch x 16 units) += h_prev (16 x K) . W_g^T (K x 16) ---
  if (has_prev) {
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int rowi = lane & 15, kq = lane >> 4;
    const int nchunk = (H + 15) >> 4;
    const int nbt = gridDim.y;
    // previous step's packed state: parity (step-1)&1
    const float* hsrc = hp + ((((int64_t)((step + 1) & 1) * D + d) * nbt + blockIdx.y) * nchunk) * FRAG + lane * 4;
    const float* wsrc = p.w[d] + ((int64_t)blockIdx.x * G * nchunk) * FRAG + lane * 4;
    for (int c0 = wave; c0 < nchunk; c0 += NW * UF_FWD) {
      float4 a[UF_FWD], w[UF_FWD][G];
#pragma unroll
      for (int u = 0; u < UF_FWD; ++u) {
        const int c = c0 + u * NW;
        const bool ok = c < nchunk;
        a[u] = ok ? ld4(hsrc + (int64_t)c * FRAG) : zero4;
#pragma unroll
        for (int g = 0; g < G; ++g) w[u][g] = ok ? ld4(wsrc + ((int64_t)g * nchunk + c) * FRAG) : zero4;
      }
#pragma unroll
      for (int u = 0; u < UF_FWD; ++u) {
        if (c0 + u * NW < nchunk) {
#pragma unroll
          for (int g = 0; g < G; ++g) { LR_MFMA4(acc[g], a[u], w[u][g]); }
        }
      }
    }
    // C/D layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[((wave * G + g) * TILE + kq * 4 + r) * RED_LD + rowi] = acc[g][r];
  }
  __syncthreads();
  if (!epi) return;

  float s[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    s[g] = 0.f;
    if (has_prev) {
#pragma unroll
      for (int w = 0; w < NW; ++w) s[g] += red[((w * G + g) * TILE + bl) * RED_LD + jl];
    }
  }
  float* yo = y + bt * DH + d * H + j;
  float* eo = extra + (bt * D + d) * H + j;
  // this step's packed state for the next launch: fragment (batch tile, chunk = unit tile),
  // lane = row + 16*(k/4), element k%4
  float* ho = hp + ((((int64_t)(step & 1) * D + d) * gridDim.y + blockIdx.y) * ((H + 15) >> 4) + blockIdx.x) * FRAG +
              (bl + 16 * (jl >> 2)) * 4 + (jl & 3);
  if (t >= len_b) {  // padded position: zero output, zero carried state
    *yo = 0.f;
    if (G != 1) *eo = 0.f;
    *ho = 0.f;
    return;
  }
  if (G == 1) {
    // nn.RNN (tanh): h' = tanh(W_ih x + b_ih + W_hh h + b_hh); the gate buffer keeps h' for backward
    const float h = tanhf(gx[0] + s[0]);
    go[0] = h;
    *yo = h;
    *ho = h;
  } else if (G == 3) {
    const float hn = s[2] + bhn;
    const float r = lr_sigmoid(gx[0] + s[0]);
    const float z = lr_sigmoid(gx[1] + s[1]);
    const float n = tanhf(gx[2] + r * hn);
    const float h = (1.f - z) * n + z * prev_own;
    go[0] = r;
    go[H] = z;
    go[2 * H] = n;
    *eo = hn;
    *yo = h;
    *ho = h;
  } else {
    const float ig = lr_sigmoid(gx[0] + s[0]);
    const float fg = lr_sigmoid(gx[1] + s[1]);
    const float gg = tanhf(gx[2] + s[2]);
    const float og = lr_sigmoid(gx[G - 1] + s[G - 1]);
    const float c = fg * prev_own + ig * gg;
    go[0] = ig;
    go[H] = fg;
    go[2 * H] = gg;
    go[3 * H] = og;
    const float h = og * tanhf(c);
    *eo = c;
    *yo = h;
    *ho = h;
  }
}

// dG has four slots per (b,t,d): GRU [dr_pre, dz_pre, dn_pre, dhn_lin], LSTM [di,df,dg,do]_pre.
// The recurrent product uses slots (0,1,3) for the GRU (d/d(W_hh h + b_hh)) and (0,1,2,3) for
// the LSTM; dW_ih / dx use slots (0,1,2) / (0,1,2,3).
template <int G>
__global__ __launch_bounds__(NW * 64) void rnn_bwd_step_kernel(
    const float* __restrict__ gates, const float* __restrict__ extra, const float* __restrict__ y,
    const float* __restrict__ dy, const float* __restrict__ dh_n, const float* __restrict__ dc_n,
    float* dG, float* dcar, float* dgp, const int32_t* __restrict__ lens, StepPtrs p, int B, int T,
    int H, int D, int step) {
  __shared__ float red[NW * TILE * RED_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = blockIdx.z;
  const int t = d == 0 ? T - 1 - step : step;   // reverse of the forward order
  const int tn = d == 0 ? t + 1 : t - 1;        // step processed just before this one
  const int tp = d == 0 ? t - 1 : t + 1;        // where h_prev / c_prev of step t live
  const bool has_next = tn >= 0 && tn < T;
  const bool has_prev = tp >= 0 && tp < T;
  const bool init_prev = !has_prev && p.h0 != nullptr;   // step 0 of a layer with an initial state
  const int j0 = blockIdx.x * TILE, b0 = blockIdx.y * TILE;
  const int DH = D * H, GH = G * H;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- epilogue operands first ----------------------------------------------------------------
  const int bl = (tid >> 4) & 15, jl = tid & 15;
  const int b = b0 + bl, j = j0 + jl;
  const bool epi = tid < 256 && b < B && j < H;
  const int64_t bt = (int64_t)b * T + t, btn = (int64_t)b * T + tn, btp = (int64_t)b * T + tp;
  int len_b = 0;
  float dh = 0.f, car = 0.f, inj_h = 0.f, inj_c = 0.f, gv[4] = {0.f, 0.f, 0.f, 0.f};
  float ex = 0.f, prev = 0.f;
  if (epi) {
    len_b = lens[b];
    dh = dy[bt * DH + d * H + j];
    if (has_next) car = dcar[(btn * D + d) * H + j];
    const float* gi = gates + (bt * D + d) * (int64_t)GH + j;
#pragma unroll
    for (int g = 0; g < G; ++g) gv[g] = gi[g * H];
    if (G != 1) ex = extra[(bt * D + d) * H + j];           // GRU: W_hn h + b_hn ; LSTM: c_t
    if (G == 1) {
    } else if (has_prev) prev = G == 3 ? y[btp * DH + d * H + j] : extra[(btp * D + d) * H + j];
    else if (init_prev) {
      const float* src = G == 3 ? p.h0 : p.c0;
      if (src) prev = src[((int64_t)d * B + b) * H + j];
    }
    if (dh_n) inj_h = dh_n[((int64_t)d * B + b) * H + j];
    if (G == 4 && dc_n) inj_c = dc_n[((int64_t)d * B + b) * H + j];
  }

  // ---- recurrent product: dh[b][j] += sum_{g,k} dG_h[b][tn][g][k] * W_hh[g*H+k][j] -------------
  constexpr int UF_BWD = G == 4 ? UF_BWD_LSTM : UF_BWD_GRU;
  if (has_next) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int rowi = lane & 15, kq = lane >> 4;
    const int nchunk = (H + 15) >> 4;
    const int total = G * nchunk;   // K = (gate, k) in fragments of 16
    const float* asrc = dgp + ((((int64_t)((step + 1) & 1) * D + d) * gridDim.y + blockIdx.y) * total) * FRAG + lane * 4;
    const float* wsrc = p.w[d] + ((int64_t)blockIdx.x * total) * FRAG + lane * 4;
    for (int f0 = wave; f0 < total; f0 += NW * UF_BWD) {
      float4 a[UF_BWD], w[UF_BWD];
#pragma unroll
      for (int u = 0; u < UF_BWD; ++u) {
        const int f = f0 + u * NW;
        const bool ok = f < total;
        a[u] = ok ? ld4(asrc + (int64_t)f * FRAG) : zero4;
        w[u] = ok ? ld4(wsrc + (int64_t)f * FRAG) : zero4;
      }
#pragma unroll
      for (int u = 0; u < UF_BWD; ++u) {
        if (f0 + u * NW < total) { LR_MFMA4(acc, a[u], w[u]); }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * TILE + kq * 4 + r) * RED_LD + rowi] = acc[r];
  }
  __syncthreads();
  if (!epi) return;

  float* dgo = dG + (bt * D + d) * (int64_t)(4 * H) + j;
  float* dco = dcar + (bt * D + d) * H + j;
  // packed copy of d/d(W_hh h + b_hh) for the next launch's product: fragment f = g*nchunk + chunk
  const int nchunk_e = (H + 15) >> 4;
  float* po = dgp + ((((int64_t)(step & 1) * D + d) * gridDim.y + blockIdx.y) * (G * nchunk_e) + blockIdx.x) * FRAG +
              (bl + 16 * (jl >> 2)) * 4 + (jl & 3);
  const int64_t pstride = (int64_t)nchunk_e * FRAG;  // gate g -> + g * pstride
  if (t >= len_b) {  // padded position: contributes nothing, carries nothing
    dgo[0] = 0.f;
    dgo[H] = 0.f;
    dgo[2 * H] = 0.f;
    dgo[3 * H] = 0.f;
    *dco = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) po[g * pstride] = 0.f;
    return;
  }
  if (has_next) {
#pragma unroll
    for (int w = 0; w < NW; ++w) dh += red[(w * TILE + bl) * RED_LD + jl];
  }
  const bool is_last = d == 0 ? (t == len_b - 1) : (t == 0);  // where the final state was read
  if (is_last) dh += inj_h;
  if (G == 1) {
    const float h = gv[0];
    const float dpre = dh * (1.f - h * h);
    dgo[0] = dpre;
    dgo[H] = 0.f;
    dgo[2 * H] = 0.f;
    dgo[3 * H] = 0.f;
    *dco = 0.f;
    po[0] = dpre;
  } else if (G == 3) {
    dh += car;  // dh_{t+1} * z_{t+1}
    const float r = gv[0], z = gv[1], n = gv[2], hn = ex, hp = prev;
    const float dn_pre = dh * (1.f - z) * (1.f - n * n);
    const float dr_pre = dn_pre * hn * r * (1.f - r);
    const float dz_pre = dh * (hp - n) * z * (1.f - z);
    dgo[0] = dr_pre;
    dgo[H] = dz_pre;
    dgo[2 * H] = dn_pre;
    dgo[3 * H] = dn_pre * r;
    *dco = dh * z;
    po[0] = dr_pre;
    po[pstride] = dz_pre;
    po[2 * pstride] = dn_pre * r;   // recurrent path of the n gate: d/d(W_hn h + b_hn)
  } else {
    float dc = car;  // dc_{t+1} * f_{t+1}
    if (is_last) dc += inj_c;
    const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3], cp = prev;
    const float tc = tanhf(ex);
    dc += dh * og * (1.f - tc * tc);
    const float di = dc * gg * ig * (1.f - ig);
    const float df = dc * cp * fg * (1.f - fg);
    const float dg_ = dc * ig * (1.f - gg * gg);
    const float do_ = dh * tc * og * (1.f - og);
    dgo[0] = di;
    dgo[H] = df;
    dgo[2 * H] = dg_;
    dgo[3 * H] = do_;
    *dco = dc * fg;
    po[0] = di;
    po[pstride] = df;
    po[2 * pstride] = dg_;
    po[(G - 1) * pstride] = do_;
  }
}

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
// bias folded into the input projection: b_ih + b_hh, except the GRU n-gate whose b_hn sits
// inside r*(W_hn h + b_hn).
__global__ void fold_bias_kernel(const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                 float* __restrict__ out, int G, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * H) return;
  float v = b_ih[i];
  if (G != 3 || i < 2 * H) v += b_hh[i];
  out[i] = v;
}
// both directions of a layer in one launch (blockIdx.y = direction; out[d] = out + d * G * H)
__global__ void fold_bias2_kernel(const float* __restrict__ b_ih0, const float* __restrict__ b_hh0,
                                  const float* __restrict__ b_ih1, const float* __restrict__ b_hh1,
                                  float* __restrict__ out, int G, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, d = blockIdx.y;
  if (i >= G * H) return;
  const float* b_ih = d ? b_ih1 : b_ih0;
  const float* b_hh = d ? b_hh1 : b_hh0;
  float v = b_ih[i];
  if (G != 3 || i < 2 * H) v += b_hh[i];
  out[(size_t)d * G * H + i] = v;
}

// W_hh [G*H][H] -> fragment-major operand for the FORWARD product (rows = hidden units, K = h):
//   out[((jt*G + g)*nchunk + c)*256 + l*4 + q] = W_hh[g*H + jt*16 + (l&15)][c*16 + 4*(l>>4) + q]
// and for the BACKWARD product (rows = hidden units j, K = (gate, k) recurrent outputs):
//   out[((jt*G + g)*nchunk + c)*256 + l*4 + q] = W_hh[g*H + c*16 + 4*(l>>4) + q][jt*16 + (l&15)]
// zero where the unit or k index is >= H.
__global__ void pack_w_kernel(const float* __restrict__ W, float* __restrict__ out, int G, int H,
                              int transposed) {
  const int nchunk = (H + 15) >> 4;
  const int64_t total = (int64_t)nchunk * G * nchunk * FRAG;  // unit tiles == chunks
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i & 3), l = (int)((i >> 2) & 63);
    int64_t r = i >> 8;
    const int c = (int)(r % nchunk);
    r /= nchunk;
    const int g = (int)(r % G);
    const int jt = (int)(r / G);
    const int unit = jt * 16 + (l & 15), k = c * 16 + 4 * (l >> 4) + q;
    float v = 0.f;
    if (unit < H && k < H)
      v = transposed ? W[((int64_t)g * H + k) * H + unit] : W[((int64_t)g * H + unit) * H + k];
    out[i] = v;
  }
}

__global__ void final_state_kernel(const float* __restrict__ y, const float* __restrict__ extra,
                                   const int32_t* __restrict__ lens, float* __restrict__ h_n,
                                   float* __restrict__ c_n, int B, int T, int H, int D) {
  const int64_t total = (int64_t)D * B * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % H);
    const int b = (int)((i / H) % B);
    const int d = (int)(i / ((int64_t)H * B));
    int tl = d == 0 ? lens[b] - 1 : 0;
    if (tl < 0) tl = 0;
    if (tl >= T) tl = T - 1;
    const int64_t bt = (int64_t)b * T + tl;
    h_n[i] = y[bt * D * H + d * H + j];
    if (c_n) c_n[i] = extra[(bt * D + d) * H + j];
  }
}

// Bias gradients: fixed-order sum of the LR_COLSUM_SPLITS partial column sums of
// dG [rows][D][4][H] (lr_colsum_partial), scattered into db_ih / db_hh.
__global__ void bias_grad_final_kernel(LrRnnBiasJob p) {
  lr_rnn_bias_final_body(p, blockIdx.x * blockDim.x + threadIdx.x);
}

// h [B][H] -> one packed state slot [batch tile][chunk][64][4] (rows past B / k past H stay zero)
__global__ void pack_state_kernel(const float* __restrict__ h, float* __restrict__ slot, int B, int H) {
  const int nchunk = (H + 15) >> 4;
  const int64_t total = (int64_t)B * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / H), k = (int)(i - (int64_t)b * H);
    const int kk = k & 15;
    slot[(((int64_t)(b >> 4) * nchunk + (k >> 4)) * 64 + (b & 15) + 16 * (kk >> 2)) * 4 + (kk & 3)] = h[i];
  }
}

// Gradient that leaves a layer through its initial state (single direction, forward in time):
//   dh0[b][j] = (GRU: dh_0 * z_0) + sum_{g,k} dG_h[b][t=0][g][k] * W_hh[g*H+k][j],   dc0 = dc_0 * f_0 (LSTM)
// i.e. one more recurrent product after the last backward step, without a gate stage.
template <int G>
__global__ __launch_bounds__(NW * 64) void rnn_dh0_kernel(const float* __restrict__ dcar,
                                                         const float* __restrict__ dgp_slot,
                                                         const float* __restrict__ wpT, float* __restrict__ dh0,
                                                         float* __restrict__ dc0, int B, int T, int H) {
  __shared__ float red[NW * TILE * RED_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nchunk = (H + 15) >> 4, total = G * nchunk;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* asrc = dgp_slot + ((int64_t)blockIdx.y * total) * FRAG + lane * 4;
  const float* wsrc = wpT + ((int64_t)blockIdx.x * total) * FRAG + lane * 4;
  for (int f0 = wave; f0 < total; f0 += NW * UF_BWD_GRU) {
    float4 a[UF_BWD_GRU], w[UF_BWD_GRU];
#pragma unroll
    for (int u = 0; u < UF_BWD_GRU; ++u) {
      const int f = f0 + u * NW;
      const bool ok = f < total;
      a[u] = ok ? ld4(asrc + (int64_t)f * FRAG) : zero4;
      w[u] = ok ? ld4(wsrc + (int64_t)f * FRAG) : zero4;
    }
#pragma unroll
    for (int u = 0; u < UF_BWD_GRU; ++u) {
      if (f0 + u * NW < total) { LR_MFMA4(acc, a[u], w[u]); }
    }
  }
  const int rowi = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * TILE + kq * 4 + r) * RED_LD + rowi] = acc[r];
  __syncthreads();
  const int bl = (tid >> 4) & 15, jl = tid & 15;
  const int b = blockIdx.y * TILE + bl, j = blockIdx.x * TILE + jl;
  if (tid >= 256 || b >= B || j >= H) return;
  float dh = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) dh += red[(w * TILE + bl) * RED_LD + jl];
  const float car = dcar[((int64_t)b * T) * H + j];   // t = 0, D = 1
  if (G == 1) {
    dh0[(int64_t)b * H + j] = dh;
  } else if (G == 3) {
    dh0[(int64_t)b * H + j] = dh + car;
  } else {
    dh0[(int64_t)b * H + j] = dh;
    if (dc0) dc0[(int64_t)b * H + j] = car;
  }
}

struct Layout {
  size_t gates, extra, bias, wp, hp, gemm, xch, wpb, xchb, total;  // float offsets / total floats
  size_t wp_per_dir, hp_floats, gemm_bytes, xch_bytes, wpb_bytes, xchb_bytes;
};
Layout reserve_layout(int G, int B, int T, int I, int H, int D, bool x3 = false) {
  Layout l;
  const size_t nchunk = (H + 15) / 16, nbt = (B + 15) / 16;
  l.gates = 0;
  l.extra = l.gates + (size_t)B * T * D * G * H;
  l.bias = l.extra + (size_t)B * T * D * H;
  l.wp = (l.bias + (size_t)D * G * H + 63) / 64 * 64;       // 256-byte aligned
  l.wp_per_dir = nchunk * G * nchunk * FRAG;
  if (lr_rnn_cluster_supported(G, B, H)) {   // the cluster recurrence's fragment order pads H and the GRU's gate tiles
    const size_t cf = (lr_rnn_cluster_pack_bytes(G, H, 1, 0) / sizeof(float) + 63) / 64 * 64;
    if (cf > l.wp_per_dir) l.wp_per_dir = cf;
  }
  l.hp = l.wp + (size_t)D * l.wp_per_dir;
  l.hp_floats = 2 * (size_t)D * nbt * nchunk * FRAG;
  l.gemm = l.hp + l.hp_floats;                               // split-K slabs of the input projection
  l.gemm_bytes = x3 ? lr_xproj_workspace_bytes(B * T, I, G * H, D, H) : lr_sgemm_workspace_bytes(B * T, G * H, I);
  l.xch = (l.gemm + (l.gemm_bytes + 3) / 4 + 63) / 64 * 64;   // exchange words of the cluster recurrence
  l.xch_bytes = lr_rnn_cluster_supported(G, B, H) ? lr_rnn_cluster_xch_bytes(B, H, D, 0) : 0;
  // the cluster recurrence's BACKWARD fragments of W_hh and first exchange words: prepared by the forward's prologue
  // launch (round 5), kept with the gates for the backward
  l.wpb = (l.xch + (l.xch_bytes + 3) / 4 + 63) / 64 * 64;
  l.wpb_bytes = lr_rnn_cluster_supported(G, B, H) ? lr_rnn_cluster_pack_bytes(G, H, D, 1) : 0;
  l.xchb = (l.wpb + (l.wpb_bytes + 3) / 4 + 63) / 64 * 64;
  l.xchb_bytes = lr_rnn_cluster_supported(G, B, H) ? lr_rnn_cluster_xch_bytes(B, H, D, 1) : 0;
  l.total = l.xchb + (l.xchb_bytes + 3) / 4;
  return l;
}

// Which reserves hold backward fragments that no backward has consumed yet (host side, keyed by the reserve's address):
// set by a forward whose prologue packed them, taken by the first backward over that reserve, which then skips its own
// pack launch.  A second backward over the same forward (retain_graph), or one whose forward ran with the one-launch
// recurrence switched off, finds no entry and packs (and clears the exchange words) itself, as rounds 1-4 always did.
// Under stream capture the decision is taken once, at capture: the graph replays forward-then-backward as captured.
struct FreshPacks {
  static constexpr int N = 64;
  std::mutex mu;
  const void* key[N] = {};
  int next = 0;
  void put(const void* p) {
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < N; ++i)
      if (key[i] == p) return;
    key[next] = p;
    next = (next + 1) % N;
  }
  bool take(const void* p) {
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < N; ++i)
      if (key[i] == p) { key[i] = nullptr; return true; }
    return false;
  }
};
FreshPacks g_fresh_packs;
// A layer's weight half as split-K lr_fgemm jobs (what a regime-R layer on the one-launch recurrence runs, see
// rnn_layer_backward_impl): K = B * T rows is long and the products are a few dozen tiles, so K is cut until the launch
// is ~1.5 workgroups per compute unit.  Returns the K split; *slab_floats = what the jobs' partial sums take.
int wgrad_fgemm_plan(int G, int R, int I, int H, int D, size_t* slab_floats) {
  const int GH = G * H;
  int Ms[3] = {GH, G == 3 ? 2 * H : GH, H}, Ns[3] = {I, H, H};
  const int nj = G == 3 ? 3 : 2;
  long tiles = 0;
  for (int j = 0; j < nj; ++j) tiles += (long)((Ms[j] + 127) / 128) * ((Ns[j] + 127) / 128);
  tiles *= D;
  const long stages = (R + 31) / 32;
  long sp = tiles > 0 ? 384 / tiles : 1;
  if (sp > stages / 8) sp = stages / 8;
  if (sp > 16) sp = 16;
  if (sp < 1) sp = 1;
  size_t f = 0;
  for (int j = 0; j < nj; ++j) f += (lr_fgemm_slab_floats_impl(Ms[j], Ns[j], (int)sp) + 63) / 64 * 64;
  *slab_floats = f * D;
  return (int)sp;
}

struct WsLayout {
  size_t dG, dcar, wT, dgp, colsum, gemm, xch, total;  // float offsets
  size_t gemm_bytes, wp_per_dir, dgp_floats, xch_bytes;
};
WsLayout ws_layout(int G, int B, int T, int I, int H, int D) {
  WsLayout l;
  const size_t nchunk = (H + 15) / 16, nbt = (B + 15) / 16;
  l.dG = 0;
  l.dcar = l.dG + (size_t)B * T * D * 4 * H;
  l.wT = (l.dcar + (size_t)B * T * D * H + 63) / 64 * 64;   // packed W_hh^T, 256-byte aligned
  l.wp_per_dir = nchunk * G * nchunk * FRAG;
  if (lr_rnn_cluster_supported(G, B, H)) {
    const size_t cf = (lr_rnn_cluster_pack_bytes(G, H, 1, 1) / sizeof(float) + 63) / 64 * 64;
    if (cf > l.wp_per_dir) l.wp_per_dir = cf;
  }
  l.dgp = l.wT + (size_t)D * l.wp_per_dir;                  // packed dG_h, two parities
  l.dgp_floats = 2 * (size_t)D * nbt * G * nchunk * FRAG;
  l.colsum = l.dgp + l.dgp_floats;
  l.gemm = l.colsum + (size_t)LR_COLSUM_SPLITS * D * 4 * H;
  size_t gb = lr_sgemm_workspace_bytes(G * H, I, B * T);
  size_t g2 = lr_sgemm_workspace_bytes(G * H, H, B * T);
  if (g2 > gb) gb = g2;
  g2 = lr_sgemm_workspace_bytes(2 * H, H, B * T);
  if (g2 > gb) gb = g2;
  g2 = lr_sgemm_workspace_bytes(H, H, B * T);
  if (g2 > gb) gb = g2;
  g2 = lr_sgemm_workspace_bytes(B * T, I, G * H);
  if (g2 > gb) gb = g2;
  {   // the layer's weight gradients as ONE grouped launch (rnn_layer_backward_impl)
    int Ms[8], Ns[8], Ks[8], n = 0;
    for (int d = 0; d < D; ++d) {
      Ms[n] = G * H; Ns[n] = I; Ks[n] = B * T; ++n;
      if (G == 3) {
        Ms[n] = 2 * H; Ns[n] = H; Ks[n] = B * T; ++n;
        Ms[n] = H; Ns[n] = H; Ks[n] = B * T; ++n;
      } else {
        Ms[n] = G * H; Ns[n] = H; Ks[n] = B * T; ++n;
      }
    }
    g2 = lr_sgemm_grouped_workspace_bytes(n, Ms, Ns, Ks);
    if (g2 > gb) gb = g2;
  }
  {
    size_t sf = 0;
    wgrad_fgemm_plan(G, B * T, I, H, D, &sf);
    if (sf * sizeof(float) > gb) gb = sf * sizeof(float);
  }
  l.gemm_bytes = gb;
  l.xch = (l.gemm + (gb + 3) / 4 + 63) / 64 * 64;
  l.xch_bytes = lr_rnn_cluster_supported(G, B, H) ? lr_rnn_cluster_xch_bytes(B, H, D, 1) : 0;
  l.total = l.xch + (l.xch_bytes + 3) / 4;
  return l;
}

inline int cell_of(int mode) { return mode & LR_RNN_CELL_MASK; }
inline int gates_of(int mode) { return cell_of(mode) == LR_RNN_GRU ? 3 : (cell_of(mode) == LR_RNN_LSTM ? 4 : 1); }
inline bool proj_x3(int mode) { return (mode & LR_RNN_PROJ_BF16X3) != 0; }
inline bool x_exact(int mode) { return (mode & LR_RNN_INPUT_BF16_EXACT) != 0; }
inline bool recur_split(int mode) { return (mode & LR_RNN_RECUR_SPLIT) != 0; }
inline bool x_stored_bf16(int mode) { return (mode & LR_RNN_INPUT_STORED_BF16) != 0; }
inline bool proj_x1(int mode) { return (mode & LR_RNN_PROJ_BF16X1) != 0; }
bool dims_ok(int mode, int B, int T, int I, int H, int D) {
  return (cell_of(mode) == LR_RNN_GRU || cell_of(mode) == LR_RNN_LSTM || cell_of(mode) == LR_RNN_TANH) &&
         (mode & ~(LR_RNN_CELL_MASK | LR_RNN_PROJ_BF16X3 | LR_RNN_INPUT_BF16_EXACT | LR_RNN_INPUT_STORED_BF16 |
                   LR_RNN_RECUR_SPLIT | LR_RNN_PROJ_BF16X1)) == 0 &&
         (!proj_x1(mode) || proj_x3(mode)) &&
         // a bf16-stored input only makes sense on the split-bf16 projection, as an exact operand
         (!x_stored_bf16(mode) || (proj_x3(mode) && x_exact(mode) && I % 8 == 0)) &&
         B > 0 && T > 0 && I > 0 && H > 0 && (D == 1 || D == 2);
}
// extra workspace floats of the bf16x3 input projection's backward (operand planes + split-K slabs
// of the larger of its products, all directions in one contraction)
size_t x3_ws_floats(int G, int B, int T, int I, int H, int D) {
  size_t b = lr_xproj_workspace_bytes(B * T, I, G * H, D, H);
  const size_t b2 = lr_xproj_dw_both_workspace_bytes(B * T, I, G * H, H, D);
  if (b2 > b) b = b2;
  return (b / sizeof(float) + 63) / 64 * 64;
}
// Rounds 3-4: the weight gradients of a LARGE layer whose recurrence runs fp32-faithful on the bf16 matrix cores
// (LR_RNN_RECUR_SPLIT, G * H >= 1536) as packed split-bf16 products (lr_xgemm.hip: one pack of dG, two contractions, a
// combine) instead of the fp32-MFMA grouped GEMM (LSTM-768: 1.218 -> 1.155 ms per step then).  Round 5: EVERY such layer,
// large or small, takes the one-launch lr_fgemm weight half (rnn_layer_backward_impl; BiLSTM-768 1.01 -> 0.87 ms); this
// predicate now only sizes the workspace of the packed path, which test hook bit 3 still selects for the A/B.
bool wgrad_split(int mode, int G, int H) {
  return recur_split(mode) && !proj_x3(mode) && G * H >= 1536 && !lr_debug_wgrad_f32();
}

}  // namespace

extern "C" int lr_rnn_pair_supported(int mode, int B, int T, int I, int H, int D) {
  return lr_rnn_one_launch_status(mode, B, T, I, H, D) == 0 ? 2 : 0;
}

// why (not): 0 = the layer has a one-launch recurrence on this device now; 1 = no kernel for the shape (H % 4 != 0, H
// past the largest cluster, a tanh RNN); 2 = switched off by lr_rnn_one_launch_enable(0) (lipreading_amd.train after
// repeated time-outs); 3 = switched off by the test hook lr_rnn_debug_disable_cluster; 4 = the device has too few
// compute units for a launch's clusters
extern "C" int lr_rnn_one_launch_status(int mode, int B, int T, int I, int H, int D) {
  if (!dims_ok(mode, B, T, I, H, D) || H % 4 != 0) return 1;   // (H % 4: lr_rnn_layer_forward's own requirement)
  const int need = lr_rnn_cluster_cus(gates_of(mode), H);
  if (need == 0) return 1;
  if (!lr_rnn_one_launch_enabled()) return 2;
  if (lr_debug_cluster_disabled()) return 3;
  return lr_device_cus() >= need ? 0 : 4;
}

// recurrence launches ONE pass of the layer takes on this device: 1 where every (direction, 8 samples) cluster of the
// batch fits one launch (round 6: up to 8 floor(32 / members) clusters per launch — GRU-256 up to B = 128, LSTM-512 up
// to B = 64; 24-member LSTM-768 clusters: 8 of sixteen samples each, i.e. B = 64 bidirectional), T where the layer runs the
// step kernels
extern "C" int lr_rnn_pass_launches(int mode, int B, int T, int I, int H, int D) {
  if (!dims_ok(mode, B, T, I, H, D)) return 0;
  if (lr_rnn_one_launch_status(mode, B, T, I, H, D) != 0) return T;
  return lr_rnn_cluster_launches(gates_of(mode), B, H, D);
}

extern "C" size_t lr_rnn_reserve_bytes(int mode, int B, int T, int I, int H, int D) {
  if (!dims_ok(mode, B, T, I, H, D)) return 0;
  return reserve_layout(gates_of(mode), B, T, I, H, D, proj_x3(mode)).total * sizeof(float);
}

extern "C" size_t lr_rnn_workspace_bytes(int mode, int B, int T, int I, int H, int D) {
  if (!dims_ok(mode, B, T, I, H, D)) return 0;
  const int G = gates_of(mode);
  return ((ws_layout(G, B, T, I, H, D).total + 63) / 64 * 64 +
          ((proj_x3(mode) || wgrad_split(mode, G, H)) ? x3_ws_floats(G, B, T, I, H, D) : 0)) *
         sizeof(float);
}

extern "C" int lr_rnn_layer_forward(int mode, const float* x, const int32_t* lens,
                                    const float* const* w_ih, const float* const* w_hh,
                                    const float* const* b_ih, const float* const* b_hh, float* y,
                                    float* h_n, float* c_n, void* reserve, size_t reserve_bytes,
                                    int B, int T, int I, int H, int D, lr_stream_t stream_) {
  LR_CHECK_ARG(dims_ok(mode, B, T, I, H, D));
  LR_CHECK_ARG(x && lens && w_ih && w_hh && b_ih && b_hh && y && reserve);
  if (H % 4 != 0) return LR_ERR_UNSUPPORTED;  // float4 operand loads along K
  const int G = gates_of(mode);
  LR_CHECK_ARG(G != 4 || c_n || !h_n);   // h_n == NULL: the caller does not want the final states (one launch less)
  for (int d = 0; d < D; ++d) LR_CHECK_ARG(w_ih[d] && w_hh[d] && b_ih[d] && b_hh[d]);
  const Layout l = reserve_layout(G, B, T, I, H, D, proj_x3(mode));
  if (reserve_bytes < l.total * sizeof(float)) return LR_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* base = (float*)reserve;
  float* gates = base + l.gates;
  float* extra = base + l.extra;
  float* bias = base + l.bias;
  const int GH = G * H;

  // the cluster recurrence's prologue does three launches' work in one: biases folded, W_hh packed into its fragment
  // order, exchange words cleared (none of it depends on the input projection below)
  const bool cluster = recur_split(mode) && lr_rnn_cluster_supported(G, B, H);
  if (cluster) {
    if ((size_t)D * l.wp_per_dir * sizeof(float) < lr_rnn_cluster_pack_bytes(G, H, D, 0)) return LR_ERR_WORKSPACE;
    int st = lr_rnn_cluster_prologue(G, w_hh, b_ih, b_hh, bias, base + l.wp, base + l.xch, B, D, H, stream, base + l.wpb,
                                     base + l.xchb);
    if (st != LR_OK) return st;
    g_fresh_packs.put(reserve);
  } else {
    g_fresh_packs.take(reserve);
    LR_LAUNCH(fold_bias2_kernel, dim3((GH + 255) / 256, D), dim3(256), 0, stream, b_ih[0], b_hh[0], b_ih[D - 1],
              b_hh[D - 1], bias, G, H);
    int st = lr_launch_status();
    if (st != LR_OK) return st;
  }
  if (proj_x3(mode) && !x_stored_bf16(mode) && !x_exact(mode) && !proj_x1(mode) && I <= 1024 && !lr_debug_dwih_packed()) {
    // an UPPER layer of the pixel regime (fp32 input, short K): gates[:, d, :] = x . W_ih[d]^T + folded bias straight from
    // x and the weights (lr_fgemm.hip, NT form, bias in the epilogue), the directions as two jobs of one launch — no
    // pack launch (round 5, bench shape: 37 us of pack + contraction -> one launch)
    lr_fgemm_job jobs[2];
    for (int d = 0; d < D; ++d) {
      lr_fgemm_job& j = jobs[d];
      j.A = x; j.B = w_ih[d]; j.C = gates + (size_t)d * GH;
      j.bias = bias + (size_t)d * GH; j.addend = nullptr; j.mask = nullptr; j.colsum = nullptr; j.slabs = nullptr;
      j.M = B * T; j.N = GH; j.K = I; j.lda = I; j.ldb = I; j.ldc = D * GH;
      j.ldadd = 0; j.add_period = 0; j.ldmask = 0; j.flags = 0; j.splits = 1;
      j.alpha = 1.f; j.beta = 0.f;
      j.b_shift = 0; j.b_period = 0;
    }
    int st = lr_fgemm_launch(LR_FGEMM_X3, LR_FGEMM_NT, 0, 0, jobs, D, stream);
    if (st != LR_OK) return st;
  } else if (proj_x3(mode)) {
    // gates[b,t,:,:] = x[b,t,:] @ [W_ih[0]; W_ih[1]]^T + folded bias: both directions in one product
    int st = lr_xproj_forward(x, B * T, I, w_ih, GH, D, bias, gates, x_exact(mode) ? 1 : 0, x_stored_bf16(mode) ? 1 : 0,
                              l.gemm_bytes ? (void*)(base + l.gemm) : nullptr, l.gemm_bytes, stream, proj_x1(mode) ? 1 : 0);
    if (st != LR_OK) return st;
  } else if (I % 4 == 0 && D <= 2 && !lr_debug_dwih_packed()) {
    // exact fp32 (regime R, recurrence 'f32'): gates[:, d, :] = x . W_ih[d]^T + folded bias as lr_fgemm jobs on the fp32
    // matrix cores, the directions as two jobs of one launch (round 5: 128 x 128 tiles with double-buffered stages in
    // place of lr_gemm's 64 x 64 ones: 35.5 -> see profiles/r05_variants_ab.txt at B*T = 2400, I = 204, G*H = 768)
    lr_fgemm_job jobs[2];
    for (int d = 0; d < D; ++d) {
      lr_fgemm_job& j = jobs[d];
      j.A = x; j.B = w_ih[d]; j.C = gates + (size_t)d * GH;
      j.bias = bias + (size_t)d * GH; j.addend = nullptr; j.mask = nullptr; j.colsum = nullptr; j.slabs = nullptr;
      j.M = B * T; j.N = GH; j.K = I; j.lda = I; j.ldb = I; j.ldc = D * GH;
      j.ldadd = 0; j.add_period = 0; j.ldmask = 0; j.flags = 0; j.splits = 1;
      j.alpha = 1.f; j.beta = 0.f;
      j.b_shift = 0; j.b_period = 0;
    }
    // Round 6: the LARGE layers of the one-launch recurrence (G * H >= 1536: the reference's own sizes — LSTM-512 / 700 /
    // 768, GRU-800 — whose weight gradients have been split-bf16 products since round 5) take the projection as three
    // bf16 products too (~1e-5 relative; the recurrence it feeds contracts hi + lo planes itself): at M = 2400, N = 6144,
    // K = 204 the exact-fp32 MFMA form runs at 65-73 TF/s (93 us at B = 32, 329 us at the ecd family's own B = 128).
    // Small layers (BiGRU-256: 31 us) and recurrence = 'f32' stay exact.
    const bool x3 = cluster && G * H >= 1536 && !lr_debug_wgrad_f32();
    int st = lr_fgemm_launch(x3 ? LR_FGEMM_X3 : LR_FGEMM_F32, LR_FGEMM_NT, 0, 0, jobs, D, stream);
    if (st != LR_OK) return st;
  } else if (D == 2 && (((w_ih[1] - w_ih[0]) & 3) == 0)) {
    // gates[b,t,d,:] = x[b,t,:] @ W_ih[d]^T + folded bias: both directions as one batched launch (x shared,
    // W_ih[d] / bias[d] / the direction's column block of `gates` a fixed stride apart)
    int st = lr_sgemm_batched_bias_impl(0, 1, B * T, GH, I, 1.f, x, I, 0, w_ih[0], I, (int64_t)(w_ih[1] - w_ih[0]), 0.f,
                                        gates, D * GH, GH, bias, GH, D, stream);
    if (st != LR_OK) return st;
  } else {
    for (int d = 0; d < D; ++d) {
      // gates[b,t,d,:] = x[b,t,:] @ W_ih[d]^T + folded bias
      int st = lr_sgemm_impl(0, 1, B * T, GH, I, 1.f, x, I, w_ih[d], I, 0.f, gates + (size_t)d * GH, D * GH,
                             bias + (size_t)d * GH, 0, 0, l.gemm_bytes ? (void*)(base + l.gemm) : nullptr,
                             l.gemm_bytes, stream);
      if (st != LR_OK) return st;
    }
  }
  if (recur_split(mode)) {
    // one launch for all T steps, fp32-faithful (lr_rnn_cluster.hip); same interface buffers as the step kernels;
    // the step kernels' packed-W_hh area of the reserve holds the bf16 hi/lo fragments instead
    int st;
    if (lr_rnn_cluster_supported(G, B, H)) {
      if ((size_t)D * l.wp_per_dir * sizeof(float) < lr_rnn_cluster_pack_bytes(G, H, D, 0)) return LR_ERR_WORKSPACE;
      st = lr_rnn_cluster_forward(G, gates, extra, y, w_hh, b_hh, nullptr, nullptr, lens, base + l.wp, base + l.xch, B, T,
                                  D, H, stream, 1);
    } else {
      return LR_ERR_UNSUPPORTED;
    }
    if (st != LR_OK) return st;
    if (!h_n) return LR_OK;
    const int64_t total = (int64_t)D * B * H;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    LR_LAUNCH(final_state_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)y, (const float*)extra, lens, h_n,
              G == 4 ? c_n : (float*)nullptr, B, T, H, D);
    return lr_launch_status();
  }
  StepPtrs p;
  p.h0 = nullptr;
  p.c0 = nullptr;
  float* wp = base + l.wp;
  float* hp = base + l.hp;
  for (int d = 0; d < D; ++d) {
    float* out = wp + (size_t)d * l.wp_per_dir;
    int blocks = (int)((l.wp_per_dir + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    LR_LAUNCH(pack_w_kernel, dim3(blocks), dim3(256), 0, stream, w_hh[d], out, G, H, 0);
    p.w[d] = out;
    p.b[d] = b_hh[d];
  }
  if (D == 1) { p.w[1] = p.w[0]; p.b[1] = p.b[0]; }
  lr_clear_error();
  if (hipMemsetAsync(hp, 0, l.hp_floats * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;
  const dim3 grid((H + TILE - 1) / TILE, (B + TILE - 1) / TILE, D);
  for (int s = 0; s < T; ++s) {
    hipEvent_t e0, e1;
    if (s == T / 2 && lr_prof_next(LR_PROF_RNN_FWD, &e0, &e1)) {
      // sampled launch: the events carry the dispatch's own begin/end timestamps
      lr_clear_error();
      if (G == 3) hipExtLaunchKernelGGL(rnn_fwd_step_kernel<3>, grid, dim3(NW * 64), 0, stream, e0, e1, 0, gates, extra, y, hp, lens, p, B, T, H, D, s);
      else if (G == 4) hipExtLaunchKernelGGL(rnn_fwd_step_kernel<4>, grid, dim3(NW * 64), 0, stream, e0, e1, 0, gates, extra, y, hp, lens, p, B, T, H, D, s);
      else hipExtLaunchKernelGGL(rnn_fwd_step_kernel<1>, grid, dim3(NW * 64), 0, stream, e0, e1, 0, gates, extra, y, hp, lens, p, B, T, H, D, s);
      continue;
    }
    if (G == 3) LR_LAUNCH(rnn_fwd_step_kernel<3>, grid, dim3(NW * 64), 0, stream, gates, extra, y, hp, lens, p, B, T, H, D, s);
    else if (G == 4) LR_LAUNCH(rnn_fwd_step_kernel<4>, grid, dim3(NW * 64), 0, stream, gates, extra, y, hp, lens, p, B, T, H, D, s);
    else LR_LAUNCH(rnn_fwd_step_kernel<1>, grid, dim3(NW * 64), 0, stream, gates, extra, y, hp, lens, p, B, T, H, D, s);
  }
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  if (!h_n) return LR_OK;
  const int64_t total = (int64_t)D * B * H;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  LR_LAUNCH(final_state_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)y,
            (const float*)extra, lens, h_n, G == 4 ? c_n : (float*)nullptr, B, T, H, D);
  return lr_launch_status();
}

// parts & 1: the recurrence (dG into the workspace) and the data gradient dx;  parts & 2: weight and
// bias gradients from the dG a parts & 1 call left in the same workspace (split calls: only on the
// LR_RNN_PROJ_BF16X3 path, where the two halves touch disjoint outputs and may run on two streams)
static int rnn_layer_backward_impl(int mode, const float* x, const int32_t* lens,
                                   const float* const* w_ih, const float* const* w_hh,
                                   const float* const* b_ih, const float* const* b_hh,
                                   const float* y, const float* dy, const float* dh_n,
                                   const float* dc_n, float* dx, float* const* dw_ih,
                                   float* const* dw_hh, float* const* db_ih, float* const* db_hh,
                                   const void* reserve, size_t reserve_bytes, void* workspace,
                                   size_t workspace_bytes, int accumulate, int B, int T, int I,
                                   int H, int D, int parts, lr_stream_t stream_) {
  LR_CHECK_ARG(dims_ok(mode, B, T, I, H, D));
  LR_CHECK_ARG((parts & ~3) == 0 && parts != 0);
  if (parts != 3 && !proj_x3(mode)) return LR_ERR_UNSUPPORTED;
  LR_CHECK_ARG(x && lens && w_ih && w_hh && y && dy && reserve && workspace);
  const float wbeta = accumulate ? 1.f : 0.f;
  LR_CHECK_ARG(dw_ih && dw_hh && db_ih && db_hh);
  if (H % 4 != 0) return LR_ERR_UNSUPPORTED;
  const int G = gates_of(mode);
  for (int d = 0; d < D; ++d)
    LR_CHECK_ARG(w_ih[d] && w_hh[d] && dw_ih[d] && dw_hh[d] && db_ih[d] && db_hh[d]);
  (void)b_ih;
  (void)b_hh;
  const Layout rl = reserve_layout(G, B, T, I, H, D, proj_x3(mode));
  if (reserve_bytes < rl.total * sizeof(float)) return LR_ERR_WORKSPACE;
  const WsLayout wl = ws_layout(G, B, T, I, H, D);
  const bool x3 = proj_x3(mode);
  const bool wx = wgrad_split(mode, G, H);   // weight gradients as split-bf16 products (see wgrad_split)
  const size_t xws_off = (wl.total + 63) / 64 * 64;   // floats; keeps the bf16 planes 16-byte aligned
  if (workspace_bytes < ((x3 || wx) ? xws_off + x3_ws_floats(G, B, T, I, H, D) : wl.total) * sizeof(float))
    return LR_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const float* rbase = (const float*)reserve;
  const float* gates = rbase + rl.gates;
  const float* extra = rbase + rl.extra;
  float* wbase = (float*)workspace;
  float* dG = wbase + wl.dG;
  float* dcar = wbase + wl.dcar;
  float* wT = wbase + wl.wT;
  void* gws = wbase + wl.gemm;
  const int GH = G * H;

  int st = LR_OK;
  if (!(parts & 1)) {
    // dG is already in the workspace
  } else if (recur_split(mode) && lr_rnn_cluster_supported(G, B, H)) {
    if ((size_t)D * wl.wp_per_dir * sizeof(float) < lr_rnn_cluster_pack_bytes(G, H, D, 1)) return LR_ERR_WORKSPACE;
    // fragments of W_hh and exchange words: in the reserve, left there by the forward's prologue unless an earlier
    // backward used them up (g_fresh_packs)
    float* rw = const_cast<float*>(rbase);
    st = lr_rnn_cluster_backward(G, gates, extra, y, dy, dh_n, dc_n, dG, nullptr, nullptr, nullptr, nullptr, w_hh, lens,
                                 rw + rl.wpb, rw + rl.xchb, B, T, D, H, stream, g_fresh_packs.take(reserve) ? 1 : 0);
    if (st != LR_OK) return st;
  } else {
    StepPtrs p;
    p.h0 = nullptr;
    p.c0 = nullptr;
    float* dgp = wbase + wl.dgp;
    for (int d = 0; d < D; ++d) {
      float* out = wT + (size_t)d * wl.wp_per_dir;
      int blocks = (int)((wl.wp_per_dir + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      LR_LAUNCH(pack_w_kernel, dim3(blocks), dim3(256), 0, stream, w_hh[d], out, G, H, 1);
      p.w[d] = out;
      p.b[d] = nullptr;
    }
    if (D == 1) { p.w[1] = p.w[0]; p.b[1] = nullptr; }
    st = lr_launch_status();
    if (st != LR_OK) return st;
    if (hipMemsetAsync(dgp, 0, wl.dgp_floats * sizeof(float), stream) != hipSuccess) return LR_ERR_LAUNCH;

    const dim3 grid((H + TILE - 1) / TILE, (B + TILE - 1) / TILE, D);
    for (int s = 0; s < T; ++s) {
      hipEvent_t e0, e1;
      if (s == T / 2 && lr_prof_next(LR_PROF_RNN_BWD, &e0, &e1)) {
        lr_clear_error();
        if (G == 3) hipExtLaunchKernelGGL(rnn_bwd_step_kernel<3>, grid, dim3(NW * 64), 0, stream, e0, e1, 0, gates, extra, y, dy, dh_n, dc_n, dG, dcar, dgp, lens, p, B, T, H, D, s);
        else if (G == 4) hipExtLaunchKernelGGL(rnn_bwd_step_kernel<4>, grid, dim3(NW * 64), 0, stream, e0, e1, 0, gates, extra, y, dy, dh_n, dc_n, dG, dcar, dgp, lens, p, B, T, H, D, s);
        else hipExtLaunchKernelGGL(rnn_bwd_step_kernel<1>, grid, dim3(NW * 64), 0, stream, e0, e1, 0, gates, extra, y, dy, dh_n, dc_n, dG, dcar, dgp, lens, p, B, T, H, D, s);
        continue;
      }
      if (G == 3) LR_LAUNCH(rnn_bwd_step_kernel<3>, grid, dim3(NW * 64), 0, stream, gates, extra, y, dy, dh_n, dc_n, dG, dcar, dgp, lens, p, B, T, H, D, s);
      else if (G == 4) LR_LAUNCH(rnn_bwd_step_kernel<4>, grid, dim3(NW * 64), 0, stream, gates, extra, y, dy, dh_n, dc_n, dG, dcar, dgp, lens, p, B, T, H, D, s);
      else LR_LAUNCH(rnn_bwd_step_kernel<1>, grid, dim3(NW * 64), 0, stream, gates, extra, y, dy, dh_n, dc_n, dG, dcar, dgp, lens, p, B, T, H, D, s);
    }
    st = lr_launch_status();
    if (st != LR_OK) return st;
  }

  const int R = B * T;
  const int ldg = D * 4 * H;
  void* xws = wbase + xws_off;   // bf16x3 input projection: operand planes + split-K slabs
  const size_t xws_bytes = (x3 || wx) ? x3_ws_floats(G, B, T, I, H, D) * sizeof(float) : 0;
  if (x3) {
    // input projection: all directions in one contraction per product (lr_xgemm.hip)
    if ((parts & 1) && dx) {
      // a bf16 input's gradient goes to a bf16 consumer (the conv frontend's backward): hi terms only
      st = lr_xproj_dx(dG, ldg, 4 * H, w_ih, R, I, GH, D, dx, (x_exact(mode) || proj_x1(mode)) ? 1 : 0,
                       x_stored_bf16(mode) ? 1 : 0, xws, xws_bytes, stream);
      if (st != LR_OK) return st;
    }
    if (!(parts & 2)) return LR_OK;
    if (!proj_x1(mode) && !lr_debug_dwih_packed()) {
      // The layer's whole weight half straight from dG, x and y (lr_fgemm.hip, TN form: ds_read_b64_tr_b16 delivers both
      // operands' K-major fragments, nothing is packed or transposed in memory):
      //   dW_ih[d] = dG[:, d, :GH]^T . x                        (x fp32, or the stored bf16 features: their own hi plane)
      //   dW_hh[d] = dGh^T . h_prev, h_prev[b,t] = y[b,t-1] (forward) / y[b,t+1] (reverse), zero across sequence ends: B
      //              read through a row shift; the GRU's n-gate rows take slot 3 (d/d(W_hn h + b_hn)) instead of slot 2
      //   db_ih / db_hh = the column sums of those products' A operands, emitted by their first column tiles
      // ONE launch (two when x is stored as bf16: that operand form is its own instantiation) of 24-350 workgroups in
      // place of 2 pack launches + 2 contractions + 2 split-K combines + 2 bias-gradient launches that each swept the
      // chip.  Round 5, same-visit A/B at the bench shape (profiles/r05_variants_ab.txt): dW_ih of the first layer alone
      // (M 1536, N 3456, K 2400) took the pixel step from 2.517 / 2.477 ms to 2.397 / 2.393, the whole gradient
      // BIT-identical (same products, same order).  lr_rnn_debug_disable_cluster bit 3 = the packed path, for that A/B.
      lr_fgemm_job jobs[8];
      int n = 0;
      auto add = [&](const float* A, const void* Bm, int ldb, float* C, float* bias_out, int M, int N, int shift, int period) {
        lr_fgemm_job& j = jobs[n++];
        j.A = A; j.B = Bm; j.C = C;
        j.bias = nullptr; j.addend = nullptr; j.mask = nullptr; j.colsum = bias_out; j.slabs = nullptr;
        j.M = M; j.N = N; j.K = R; j.lda = ldg; j.ldb = ldb; j.ldc = N;
        j.ldadd = 0; j.add_period = 0; j.ldmask = 0; j.flags = 0; j.splits = 1;
        j.alpha = 1.f; j.beta = wbeta;
        j.b_shift = shift; j.b_period = period;
      };
      const bool xbf = x_stored_bf16(mode);
      if (xbf) {
        for (int d = 0; d < D; ++d) add(dG + (size_t)d * 4 * H, x, I, dw_ih[d], db_ih[d], GH, I, 0, 0);
        st = lr_fgemm_launch(LR_FGEMM_X3, LR_FGEMM_TN, 0, 1, jobs, n, stream);
        if (st != LR_OK) return st;
        n = 0;
      }
      for (int d = 0; d < D; ++d) {
        const float* dGd = dG + (size_t)d * 4 * H;
        const float* yd = y + (size_t)d * H;
        const int shift = d == 0 ? -1 : 1;
        if (!xbf) add(dGd, x, I, dw_ih[d], db_ih[d], GH, I, 0, 0);
        if (G == 3) {
          add(dGd, yd, D * H, dw_hh[d], db_hh[d], 2 * H, H, shift, T);
          add(dGd + 3 * H, yd, D * H, dw_hh[d] + (size_t)2 * H * H, db_hh[d] + 2 * H, H, H, shift, T);
        } else {
          add(dGd, yd, D * H, dw_hh[d], db_hh[d], GH, H, shift, T);
        }
      }
      return lr_fgemm_launch(LR_FGEMM_X3, LR_FGEMM_TN, 0, 0, jobs, n, stream);
    }
    st = lr_xproj_dw(dG, ldg, 4 * H, x, R, I, GH, D, dw_ih, wbeta, x_exact(mode) ? 1 : 0, x_stored_bf16(mode) ? 1 : 0,
                     xws, xws_bytes, stream, proj_x1(mode) ? 1 : 0);
    if (st != LR_OK) return st;
    // recurrent weight gradient on the same split-bf16 path (one contraction per direction)
    st = lr_xproj_dwhh(dG, ldg, y, D * H, R, T, H, G, D, dw_hh, wbeta, xws, xws_bytes, stream, proj_x1(mode) ? 1 : 0);
    if (st != LR_OK) return st;
  }
  // (round 5: the lr_fgemm weight half below serves the large layers too; test hook bit 3 keeps them on lr_xgemm's
  // packed contractions, for the A/B)
  const bool wx_packed = wx && lr_debug_dwih_packed();
  if (wx_packed && G != 3) {
    // fp32-faithful on the bf16 matrix cores, like the recurrence that produced dG (lr_xgemm.hip); both products
    // read the same dG slots: ONE pack of dG
    st = lr_xproj_dw_both(dG, ldg, 4 * H, x, y, D * H, R, T, I, H, GH, D, dw_ih, dw_hh, wbeta, xws, xws_bytes, stream);
    if (st != LR_OK) return st;
  } else if (wx_packed) {
    // (GRU: the recurrent side reads slot 3 where the input side reads slot 2)
    st = lr_xproj_dw(dG, ldg, 4 * H, x, R, I, GH, D, dw_ih, wbeta, 0, 0, xws, xws_bytes, stream);
    if (st != LR_OK) return st;
    st = lr_xproj_dwhh(dG, ldg, y, D * H, R, T, H, G, D, dw_hh, wbeta, xws, xws_bytes, stream);
    if (st != LR_OK) return st;
  }
  LrRnnBiasJob bias_job;
  bias_job.dG = dG; bias_job.partial = wbase + wl.colsum;
  for (int d = 0; d < 2; ++d) {
    bias_job.db_ih[d] = db_ih[d < D ? d : 0];
    bias_job.db_hh[d] = db_hh[d < D ? d : 0];
  }
  bias_job.ld = ldg; bias_job.rows = R; bias_job.H = H; bias_job.D = D; bias_job.G = G; bias_job.accumulate = accumulate;
  bool bias_done = false;
  if (!x3 && !wx_packed && recur_split(mode) && !lr_debug_wgrad_f32()) {
    // A layer on the one-launch recurrence, exact-fp32 projection (regime R; round 5): the weight half as split-bf16
    // products straight from dG, x and y
    // (lr_fgemm.hip, TN form, K cut into `sp` ranges; ~1e-5 relative, the recurrence that produced dG is itself a
    // split-bf16 product) with the bias gradients as the products' column sums — one launch + one combine in place of
    // the exact-fp32 grouped GEMM (64 us at BiGRU-256, B = 32: the fp32 matrix cores' rate), its combine and two
    // bias-gradient launches.  recurrence = 'f32' (the per-step kernels) keeps every product exact fp32.
    size_t sf = 0;
    const int sp = wgrad_fgemm_plan(G, R, I, H, D, &sf);
    if (wl.gemm_bytes < sf * sizeof(float)) return LR_ERR_WORKSPACE;
    float* slab = (float*)gws;
    lr_fgemm_job jobs[8];
    int n = 0;
    auto add = [&](const float* A, const float* Bm, int ldb, float* C, float* bias_out, int M, int N, int shift, int period) {
      lr_fgemm_job& j = jobs[n++];
      j.A = A; j.B = Bm; j.C = C;
      j.bias = nullptr; j.addend = nullptr; j.mask = nullptr; j.colsum = bias_out;
      j.slabs = sp > 1 ? slab : nullptr;
      slab += (lr_fgemm_slab_floats_impl(M, N, sp) + 63) / 64 * 64;
      j.M = M; j.N = N; j.K = R; j.lda = ldg; j.ldb = ldb; j.ldc = N;
      j.ldadd = 0; j.add_period = 0; j.ldmask = 0; j.flags = 0; j.splits = sp;
      j.alpha = 1.f; j.beta = wbeta;
      j.b_shift = shift; j.b_period = period;
    };
    for (int d = 0; d < D; ++d) {
      const float* dGd = dG + (size_t)d * 4 * H;
      const float* yd = y + (size_t)d * H;
      const int shift = d == 0 ? -1 : 1;
      add(dGd, x, I, dw_ih[d], db_ih[d], GH, I, 0, 0);
      if (G == 3) {
        add(dGd, yd, D * H, dw_hh[d], db_hh[d], 2 * H, H, shift, T);
        add(dGd + 3 * H, yd, D * H, dw_hh[d] + (size_t)2 * H * H, db_hh[d] + 2 * H, H, H, shift, T);
      } else {
        add(dGd, yd, D * H, dw_hh[d], db_hh[d], GH, H, shift, T);
      }
    }
    st = lr_fgemm_launch(LR_FGEMM_X3, LR_FGEMM_TN, 0, 0, jobs, n, stream);
    if (st != LR_OK) return st;
    bias_done = true;
  } else if (!x3 && !wx_packed) {
    // every weight gradient of the layer in ONE grouped launch + one combine (lr_gemm.hip): each of these
    // small-M*N, K = B*T products fills a sixth of the chip on its own.
    //   dW_ih[d] (G*H x I) = dGx^T (slots 0..G-1 are contiguous rows) @ x
    //   dW_hh[d] = dGh^T @ h_prev, h_prev[b,t] = y[b,t-1] (forward dir) / y[b,t+1] (reverse dir); the GRU's
    //   n-gate rows take slot 3 (d/d(W_hn h + b_hn)) instead of slot 2
    int Ms[8], Ns[8], Ks[8], ldas[8], ldbs[8], ldcs[8], shifts[8], periods[8], n = 0;
    const float* As[8];
    const float* Bs[8];
    float* Cs[8];
    auto add = [&](int M, int N, const float* A, const float* Bm, int ldb, float* C, int shift, int period) {
      Ms[n] = M; Ns[n] = N; Ks[n] = R; As[n] = A; ldas[n] = ldg; Bs[n] = Bm; ldbs[n] = ldb; Cs[n] = C; ldcs[n] = N;
      shifts[n] = shift; periods[n] = period;
      ++n;
    };
    for (int d = 0; d < D; ++d) {
      const float* dGd = dG + (size_t)d * 4 * H;
      const float* yd = y + (size_t)d * H;
      const int shift = d == 0 ? -1 : 1;
      add(GH, I, dGd, x, I, dw_ih[d], 0, 0);
      if (G == 3) {
        add(2 * H, H, dGd, yd, D * H, dw_hh[d], shift, T);
        add(H, H, dGd + 3 * H, yd, D * H, dw_hh[d] + (size_t)2 * H * H, shift, T);
      } else {
        add(GH, H, dGd, yd, D * H, dw_hh[d], shift, T);
      }
    }
    // (the bias gradients ride along: their partial column sums are extra workgroups of this launch, the finish extra
    // workgroups of its combine launch)
    st = lr_sgemm_grouped_tn_impl(n, Ms, Ns, Ks, As, ldas, Bs, ldbs, Cs, ldcs, wbeta, shifts, periods, gws,
                                  wl.gemm_bytes, stream, &bias_job);
    if (st != LR_OK) return st;
    bias_done = true;
  }
  for (int d = 0; d < D && !x3; ++d) {
    const float* dGd = dG + (size_t)d * 4 * H;
    if (dx && !x3) {
      st = lr_sgemm_impl(0, 0, R, I, GH, 1.f, dGd, ldg, w_ih[d], I, d == 0 ? 0.f : 1.f, dx, I,
                         nullptr, 0, 0, gws, wl.gemm_bytes, stream);
      if (st != LR_OK) return st;
    }
  }
  if (bias_done) return LR_OK;
  st = lr_colsum_partial(dG, ldg, R, ldg, bias_job.partial, stream);
  if (st != LR_OK) return st;
  LR_LAUNCH(bias_grad_final_kernel, dim3((ldg + 255) / 256), dim3(256), 0, stream, bias_job);
  return lr_launch_status();
}

extern "C" int lr_rnn_layer_backward(int mode, const float* x, const int32_t* lens,
                                     const float* const* w_ih, const float* const* w_hh,
                                     const float* const* b_ih, const float* const* b_hh,
                                     const float* y, const float* dy, const float* dh_n,
                                     const float* dc_n, float* dx, float* const* dw_ih,
                                     float* const* dw_hh, float* const* db_ih, float* const* db_hh,
                                     const void* reserve, size_t reserve_bytes, void* workspace,
                                     size_t workspace_bytes, int accumulate, int B, int T, int I,
                                     int H, int D, lr_stream_t stream) {
  return rnn_layer_backward_impl(mode, x, lens, w_ih, w_hh, b_ih, b_hh, y, dy, dh_n, dc_n, dx, dw_ih, dw_hh, db_ih,
                                 db_hh, reserve, reserve_bytes, workspace, workspace_bytes, accumulate, B, T, I, H, D,
                                 3, stream);
}

extern "C" int lr_rnn_layer_backward_parts(int mode, const float* x, const int32_t* lens,
                                           const float* const* w_ih, const float* const* w_hh,
                                           const float* const* b_ih, const float* const* b_hh,
                                           const float* y, const float* dy, const float* dh_n,
                                           const float* dc_n, float* dx, float* const* dw_ih,
                                           float* const* dw_hh, float* const* db_ih, float* const* db_hh,
                                           const void* reserve, size_t reserve_bytes, void* workspace,
                                           size_t workspace_bytes, int accumulate, int B, int T, int I,
                                           int H, int D, int parts, lr_stream_t stream) {
  return rnn_layer_backward_impl(mode, x, lens, w_ih, w_hh, b_ih, b_hh, y, dy, dh_n, dc_n, dx, dw_ih, dw_hh, db_ih,
                                 db_hh, reserve, reserve_bytes, workspace, workspace_bytes, accumulate, B, T, I, H, D,
                                 parts, stream);
}

// ---- internal entry points for lr_decoder.hip (single direction, one step per call) -------------
size_t lr_rnn_packed_w_floats(int G, int H) {
  const size_t nchunk = (H + 15) / 16;
  return nchunk * G * nchunk * FRAG;
}
size_t lr_rnn_packed_state_floats(int B, int H) {   // one parity slot, one direction
  return (size_t)((B + 15) / 16) * ((H + 15) / 16) * FRAG;
}
int lr_rnn_fold_bias(const float* b_ih, const float* b_hh, float* out, int G, int H, hipStream_t stream) {
  LR_LAUNCH(fold_bias_kernel, dim3((G * H + 255) / 256), dim3(256), 0, stream, b_ih, b_hh, out, G, H);
  return lr_launch_status();
}
int lr_rnn_pack_w(const float* W, float* out, int G, int H, int transposed, hipStream_t stream) {
  int blocks = (int)((lr_rnn_packed_w_floats(G, H) + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  LR_LAUNCH(pack_w_kernel, dim3(blocks), dim3(256), 0, stream, W, out, G, H, transposed);
  return lr_launch_status();
}
int lr_rnn_pack_state(const float* h, float* slot, int B, int H, hipStream_t stream) {
  int blocks = (int)(((int64_t)B * H + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  LR_LAUNCH(pack_state_kernel, dim3(blocks), dim3(256), 0, stream, h, slot, B, H);
  return lr_launch_status();
}
int lr_rnn_step_fwd(int G, float* gates, float* extra, float* y, float* hp, const int32_t* lens,
                    const float* wp, const float* b_hh, const float* h0, const float* c0, int B, int T,
                    int H, int step, hipStream_t stream) {
  StepPtrs p;
  p.w[0] = p.w[1] = wp;
  p.b[0] = p.b[1] = b_hh;
  p.h0 = h0;
  p.c0 = c0;
  const dim3 grid((H + TILE - 1) / TILE, (B + TILE - 1) / TILE, 1);
  if (G == 3) LR_LAUNCH(rnn_fwd_step_kernel<3>, grid, dim3(NW * 64), 0, stream, gates, extra, y, hp, lens, p, B, T, H, 1, step);
  else if (G == 4) LR_LAUNCH(rnn_fwd_step_kernel<4>, grid, dim3(NW * 64), 0, stream, gates, extra, y, hp, lens, p, B, T, H, 1, step);
  else LR_LAUNCH(rnn_fwd_step_kernel<1>, grid, dim3(NW * 64), 0, stream, gates, extra, y, hp, lens, p, B, T, H, 1, step);
  return lr_launch_status();
}
int lr_rnn_step_bwd(int G, const float* gates, const float* extra, const float* y, const float* dy,
                    const float* dh_n, const float* dc_n, float* dG, float* dcar, float* dgp, const int32_t* lens,
                    const float* wpT, const float* h0, const float* c0, int B, int T, int H, int step,
                    hipStream_t stream) {
  StepPtrs p;
  p.w[0] = p.w[1] = wpT;
  p.b[0] = p.b[1] = nullptr;
  p.h0 = h0;
  p.c0 = c0;
  const dim3 grid((H + TILE - 1) / TILE, (B + TILE - 1) / TILE, 1);
  if (G == 3) LR_LAUNCH(rnn_bwd_step_kernel<3>, grid, dim3(NW * 64), 0, stream, gates, extra, y, dy, dh_n, dc_n, dG, dcar, dgp, lens, p, B, T, H, 1, step);
  else if (G == 4) LR_LAUNCH(rnn_bwd_step_kernel<4>, grid, dim3(NW * 64), 0, stream, gates, extra, y, dy, dh_n, dc_n, dG, dcar, dgp, lens, p, B, T, H, 1, step);
  else LR_LAUNCH(rnn_bwd_step_kernel<1>, grid, dim3(NW * 64), 0, stream, gates, extra, y, dy, dh_n, dc_n, dG, dcar, dgp, lens, p, B, T, H, 1, step);
  return lr_launch_status();
}
int lr_rnn_dh0(int G, const float* dcar, const float* dgp_slot, const float* wpT, float* dh0, float* dc0, int B,
               int T, int H, hipStream_t stream) {
  const dim3 grid((H + TILE - 1) / TILE, (B + TILE - 1) / TILE, 1);
  if (G == 3) LR_LAUNCH(rnn_dh0_kernel<3>, grid, dim3(NW * 64), 0, stream, dcar, dgp_slot, wpT, dh0, dc0, B, T, H);
  else if (G == 4) LR_LAUNCH(rnn_dh0_kernel<4>, grid, dim3(NW * 64), 0, stream, dcar, dgp_slot, wpT, dh0, dc0, B, T, H);
  else LR_LAUNCH(rnn_dh0_kernel<1>, grid, dim3(NW * 64), 0, stream, dcar, dgp_slot, wpT, dh0, dc0, B, T, H);
  return lr_launch_status();
}
int lr_rnn_bias_grads(const float* dG, float* partial, float* db_ih, float* db_hh, int rows, int H, int G,
                      int accumulate, hipStream_t stream) {
  const int ldg = 4 * H;
  int st = lr_colsum_partial(dG, ldg, rows, ldg, partial, stream);
  if (st != LR_OK) return st;
  LrRnnBiasJob bp;
  bp.dG = dG; bp.partial = partial;
  bp.db_ih[0] = bp.db_ih[1] = db_ih;
  bp.db_hh[0] = bp.db_hh[1] = db_hh;
  bp.ld = ldg; bp.rows = rows; bp.H = H; bp.D = 1; bp.G = G; bp.accumulate = accumulate;
  LR_LAUNCH(bias_grad_final_kernel, dim3((ldg + 255) / 256), dim3(256), 0, stream, bp);
  return lr_launch_status();
}
