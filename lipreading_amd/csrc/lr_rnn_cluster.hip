// lr_rnn_cluster.hip — the GRU / LSTM recurrence of the REFERENCE-FAITHFUL regime as ONE launch per layer pass,
// fp32-faithful, for the hidden sizes the reference's own configs use: LSTM-700 (config/defaults.txt:19-21),
// LSTM-512 (config/train/attn/attention_type:16-19), GRU-800 (config/train/micro:6-8), LSTM-768
// (config/archive/experiments/ecd/*, BASELINE configs[2]) and the unidirectional decoder RNN started from the
// encoder's final state (better_model.py:134-148,181: hidden = D * H_enc, e.g. GRU-512 behind a BiGRU-256).
// Reference arithmetic: better_model.py:47-49,74 (nn.GRU / nn.LSTM, fp32) — the same cells as lr_rnn.hip's step
// kernels, whose interface buffers (gates in/out, extra = c or W_hn h + b_hn, y, dG) these kernels share.
//
// The step kernels re-stream W_hh from the fabric at every one of the 2T steps of a training step (LSTM-768:
// 9.4 MB per direction and step, 8.8-9.4 us per step).  Here W_hh never moves: it is split into bf16 hi + lo
// planes (the fp32 product to ~1e-6: all four cross terms of (h_hi + h_lo)(W_hi + W_lo), fp32 accumulation) and
// SLICED over a cluster of CC = ceil(H / 32) compute units per (direction, group of 8 samples).  Member c owns
// hidden units [32c, 32c + 32): its G x 32 gate rows of W_hh x HP (= 32 CC) columns x 2 planes stay in the
// registers of its four waves (60 MFMA B fragments per wave in AGPRs that the matrix core reads directly, 28 in
// VGPRs) and, past that, in LDS.  The 8 samples of a group ride in rows 0-7 (state hi) and 8-15 (state lo) of the
// 16-row MFMA A operand: rows r and r + 8 of the accumulator sum to the full product of sample r.
//
//   forward   gates[c's rows] += W_hh[c's rows][:] h_{t-1}: every member needs the WHOLE previous state — each
//             step a member publishes its 32 x 8 new state values and gathers the other members' (an all-gather).
//   backward  dh_{t-1} = W_hh^T dG_t is ROW-split: a member contracts its OWN G x 32 gate rows (its own dG:
//             nothing to gather first) into a partial dh for all HP units, publishes the partial sums the other
//             members own and gathers the CC - 1 partials of its own units, summed in a fixed order (a
//             reduce-scatter: 1/G of what a column split would have to gather).
//
// Exchange (round 3).  A value crosses as ONE self-tagged 32-bit word: the fp32 value rounded to 22 mantissa
// bits with a 2-bit tag in the two freed low bits (tag = 1 + (step / 2) mod 3; the buffer is zeroed per launch,
// so tag 0 = nothing yet).  A word is valid on its own — no ordering between words is needed, 4-byte stores are
// single-copy atomic — so the forward all-gather reads FOUR values per 16-byte load (6-7 loads per thread and
// step where round 2's 8-byte {value, tag} granules took 23) and the backward's stores and loads move half the
// bytes (round 2: 609 MB written per LSTM-768 backward launch).  The owner keeps the SAME rounded value as its
// state (y, its own k step), so every member sees one state; the rounding (2^-22 relative) is 16x finer than the
// hi + lo operand split it feeds.  Two parity slots: a member overwrites slot s & 1 at step s + 2 only after it
// consumed every other member's step-(s + 1) data, each of which was published after its author read slot s & 1;
// the stale content of a slot is always the step two back, whose tag differs.
//
// Placement.  Block b -> cluster b & 7, member b >> 3: the dispatcher places blocks k, k + 8, ... on ONE XCD
// (observed, never guaranteed — a speed matter only).  Every member publishes its XCC id (HW_REG_XCC_ID) once; when
// all members of a cluster share an XCD the words are stored at workgroup scope (`sc0`: they stay in that XCD's
// L2, where the other members' L1-bypassing loads find them) instead of agent-scope `sc1` stores that drop the
// line and send every reader to the fabric.  8 clusters x CC members <= 256 workgroups, one per CU, which must be
// resident together: lr_rnn_cluster_supported checks the device's CU count.  Waits are bounded; a member that
// gave up ORs 1 into the device-side fault word (lr_common.h lr_fault_words), which makes lr_ctc_reduce and
// lr_adam_step skip the batch — the reference's contract for a bad batch (train_better_model.py:49-50).
//
// Initial state (h0 / c0: the decoder loop, better_model.py:181) enters as the state "before step 0"; the backward
// kernel then also returns dh0 / dc0 (one more partial product + reduce-scatter after the last step).
//
// MEASURED: see DESIGN.md section 4 (round 2, 8-byte granules, LSTM-768 only: forward 4.6 us, backward 5.5 us per
// step against 9.4 / 10.7 for the step kernels).
#include "lr_common.h"
#include "lr_rnn_xch.h"
#include <hip/hip_ext.h>

namespace {

using namespace lrx;

constexpr int NS = 8;              // samples per cluster (rows 0-7 hi, 8-15 lo of the A operand) — the default form
// Round 6: SIXTEEN samples per cluster (template argument NSV = 16) for the shapes whose clusters fill an XCD (32-unit
// members, CC >= 17: LSTM-576 .. 768, GRU-576 .. 864 — BiLSTM-700 / 768 of the reference's own flag files), used when the
// batch needs more than one launch of 8-sample clusters: BiLSTM-768 at the ecd family's own B = 128 is 32 clusters of 24
// CUs, 8 per launch = FOUR launches per pass with 64 CUs idle in each.  With 16 samples the state hi and lo planes are two
// A operands (rows 0-15 each) instead of the two row halves of one: twice the MFMAs of a product that is a quarter of a
// step, no hi / lo fold of the accumulator rows, twice the exchange words — and half the launches.
constexpr int SPIN_LIMIT = 1 << 18;

constexpr int imin(int a, int b) { return a < b ? a : b; }
constexpr int imax(int a, int b) { return a > b ? a : b; }

// U = hidden units per member: 32 (a cluster of ceil(H / 32) CUs, at most one XCD's worth: the sizes up to 864 / 768),
// or 16 (round 4: twice the members, half the slice of W_hh each — LSTM-800 and the 1024-unit decoder behind a
// BiLSTM-512 encoder, config/train/attn/attention_type:16-19 with better_model.py:134-148; a cluster then spans two
// XCDs and exchanges at agent scope).  Everything below is written in these terms:
//   UW    units a wave owns (8 | 4); its gate columns are NTILE MFMA column tiles of GPT gates x UW units
//   KS    k steps of 32 state columns (CC | CC / 2: with 16-unit members a k step is a PAIR of members)
//   XB    exchange words a member publishes per step (forward) / per destination and step (backward): NS x U
//   TPB   16-byte gather items per XB block: a thread's item g = sweep * 256 + tid belongs to member g / TPB
template <int G, int CC, int U, int NSV = 8>
struct Cfg {
  static_assert(G == 3 || G == 4, "GRU or LSTM");
  static_assert(NSV == 8 || (NSV == 16 && U == 32), "16 samples per cluster: 32-unit members only");
  static constexpr int NS = NSV, NJ = NSV / 8;   // samples; (sample, unit) pairs a cell thread runs
  static_assert(U == 32 || (U == 16 && CC % 2 == 0), "16-unit members come in pairs (one k step of 32)");
  static constexpr int UW = U / 4, NTILE = U / 16, GPT = 16 / UW;
  static constexpr int HP = U * CC;                // padded hidden size
  static constexpr int KS = HP / 32;
  static constexpr int CLD = HP + 8;               // bf16 per LDS row of the state
  static constexpr int XB = NS * U, TPB = XB / 4;
  static constexpr int NL = (CC * TPB + 255) / 256;   // 16-byte gather loads per thread and step
  // clusters per launch: block b -> cluster b % ncl, member b / ncl, with ncl a multiple of XS chosen per launch.  XS = 8
  // for 32-unit members: the dispatcher places blocks k, k + 8, ... on ONE XCD (observed), and since 8 | ncl every
  // member of cluster k sits on XCD k % 8.  Rounds 2-5 launched 8 clusters whatever CC was: GRU-256 (CC = 8) at B = 64
  // became two serial launches on 64 CUs each.  An XCD has 32 CUs, so it holds floor(32 / CC) whole clusters: round 6
  // launches up to MAXCL = 8 floor(32 / CC) of them (GRU-256: 32 clusters = B 128 in one launch, LSTM-512: 16 = B 64).
  static constexpr int XS = CC <= 32 ? 8 : (CC <= 64 ? 4 : 2);
  static constexpr int MAXCL = (U == 32 && CC <= 16) ? XS * (32 / CC) : XS;
  // forward: CF fragments per tile: f = 2 * local k step + plane
  static constexpr int CF = 2 * KS;
  static constexpr int CF_A = imin(CF, 60 / NTILE);             // f < CF_A in AGPRs (60 fragments per wave)
  static constexpr int CF_REG = imin(CF, CF_A + (NSV == 16 ? 24 : 28) / NTILE);    // f < CF_REG in registers; the rest in LDS
                                                                // (16 samples: 16 VGPRs less — the gather lands 12 loads, not 6)
  static constexpr int CF_L = CF - CF_REG;
  static constexpr size_t FWD_LDS = (size_t)2 * 2 * NS * CLD * 2 + (size_t)4 * NTILE * CF_L * 1024 + (size_t)4 * NTILE * 256 * 4;
  // backward: K = the member's own dG in KB k steps of 32 (U = 32: one per gate — LSTM i, f, g, o; GRU dr, dz,
  // d(W_hn h); U = 16: two gates x 16 units per k step); the HP output units are column tiles of 16; wave w owns the
  // TPD tiles of every destination member w, w + 4, ... (a wave's 2 TPD accumulator values per lane and destination
  // are ONE store, and the wave's 64 lanes write that destination's whole XB block); BFW fragments per tile: f = 2 *
  // k step + plane
  static constexpr int KB = U == 32 ? G : 2;
  static constexpr int TPD = U / 16, ND = (CC + 3) / 4;
  static constexpr int NT = ND * TPD;
  static constexpr int VPL = (NSV / 4) * TPD;      // values per lane and destination (8 samples: two per tile behind the hi / lo fold; 16: four)
  static constexpr int BFW = 2 * KB;
  static constexpr int BFW_A = imin(BFW, 60 / NT);
  static constexpr int BFW_REG = imin(BFW, BFW_A + imax(1, 14 / NT));
  static constexpr int BFW_L = BFW - BFW_REG;
  static constexpr int BKLD = KB * 32 + 8;
  static constexpr int NRED = 256 / TPB;           // partial sums of the gather: one per (wave, block of the sweep)
  static constexpr size_t BWD_LDS = (size_t)2 * 2 * NS * BKLD * 2 + (size_t)4 * NT * BFW_L * 1024 + (size_t)(NRED + 1) * XB * 4;
  static constexpr size_t FWD_PACK = (size_t)CC * 4 * NTILE * CF * 64 * sizeof(bf16x8);     // per direction
  static constexpr size_t BWD_PACK = (size_t)CC * 4 * NT * BFW * 64 * sizeof(bf16x8);
};

// do all members of this cluster sit on one XCD?  Every member publishes its XCC id (agent scope) and reads all
// CC; the verdict is the same on every member because it is computed from the same CC words.  Returns through
// *s_local (LDS); *bad is set on the threads that gave up waiting.
template <int CC>
__device__ __forceinline__ void xcd_handshake(u32* xid, int c, int tid, int* s_local, int& bad) {
  if (tid == 0) {
    *s_local = 1;
    publish(xid + c, 0x100u | (u32)xcc_id(), false);
  }
  __syncthreads();
  if (tid < CC) {
    u32 g = peek(xid + tid);
    int n = 0;
    while (!(g & 0x100u) && n++ < SPIN_LIMIT) {
      __builtin_amdgcn_s_sleep(2);
      g = peek(xid + tid);
    }
    if (!(g & 0x100u)) bad = 1;
    if ((int)(g & 0xf) != xcc_id() || bad) *s_local = 0;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------------------------
// W_hh [G*H][H] fp32 of each direction -> bf16 hi/lo MFMA B fragments of the forward product:
// out[((((d*CC + c)*4 + wave)*NTILE + t)*CF + f)*64 + lane] = plane f & 1 of W_hh[gate*H + unit][k .. k+7],
// gate = GPT t + col / UW, unit = U c + UW wave + col % UW, k = 32 ((own k step + (f >> 1)) % KS) + 8 kg (LOCAL k order:
// the own member's k step — c, or c / 2 with 16-unit members — first); zero where gate >= G, unit >= H or k >= H.
// The same launch clears the first launch's exchange words and (FoldPtrs given) folds the layer's biases for the
// input projection (lr_rnn.hip fold_bias2_kernel's arithmetic: b_ih + b_hh, except the GRU's n gate, whose b_hn sits
// inside r * (W_hn h + b_hn)) — one launch where there were three (a dependent launch costs ~1.5-5 us here).
struct FoldPtrs {
  const float* b_ih[2];
  const float* b_hh[2];
  float* out;      // [D][G*H], or nullptr: nothing to fold
};
// backward: out[((((d*CC + c)*4 + wave)*NT + tile)*BFW + f)*64 + lane] = plane f & 1 of W_hh[gate * H + unit + e][j],
// e = 0..7: entry 8 kg + e of k step f >> 1 of the member's own dG — U = 32: gate = the k step, unit = 32 c + 8 kg;
// U = 16: gate = 2 (k step) + (kg >> 1), unit = 16 c + 8 (kg & 1) — and j = U (wave + 4 (tile / TPD)) + 16 (tile % TPD) + col
// (the units of destination member wave + 4 (tile / TPD)); zero where gate >= G, unit + e >= H or j >= H.
template <int G, int CC, int U>
__device__ __forceinline__ void pack_bwd_body(const float* __restrict__ w0, const float* __restrict__ w1,
                                              bf16x8* __restrict__ out, int D, int H, u32* __restrict__ xch, int nzero) {
  using C = Cfg<G, CC, U>;
  constexpr int NT = C::NT, BFW = C::BFW, TPD = C::TPD;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nzero; i += gridDim.x * blockDim.x) xch[i] = 0u;
  const int64_t total = (int64_t)D * CC * 4 * NT * BFW * 64;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), f = (int)((i >> 6) % BFW), tile = (int)((i / (64 * BFW)) % NT);
    const int wave = (int)((i / (64 * BFW * NT)) & 3), c = (int)((i / (64 * BFW * NT * 4)) % CC);
    const int d = (int)(i / ((int64_t)64 * BFW * NT * 4 * CC));
    const int col = lane & 15, kg = lane >> 4, kb = f >> 1, plane = f & 1;
    const int j = U * (wave + 4 * (tile / TPD)) + 16 * (tile % TPD) + col;
    const int gate = U == 32 ? kb : 2 * kb + (kg >> 1);
    const int u0 = U == 32 ? U * c + 8 * kg : U * c + 8 * (kg & 1);
    const float* src = (d ? w1 : w0) + ((int64_t)gate * H + u0) * H + j;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bf16_t hi, lo;
      split_bf16(gate < G && j < H && u0 + e < H ? src[(int64_t)e * H] : 0.f, hi, lo);
      v[e] = __builtin_bit_cast(__bf16, plane ? lo : hi);
    }
    out[i] = v;
  }
}

// out_b != NULL (a training layer of lr_rnn.hip, round 5): the BACKWARD pass's fragments of the same W_hh and its first
// launch's exchange words as well — the weights are the ones the backward differentiates through, and the backward loses
// a ~5 us launch from its critical path
template <int G, int CC, int U>
__global__ void rnnc_pack_fwd_kernel(const float* __restrict__ w0, const float* __restrict__ w1, bf16x8* __restrict__ out,
                                     int D, int H, u32* __restrict__ xch, int nzero, FoldPtrs fold,
                                     bf16x8* __restrict__ out_b, u32* __restrict__ xch_b, int nzero_b) {
  using C = Cfg<G, CC, U>;
  constexpr int CF = C::CF, NTILE = C::NTILE, UW = C::UW, GPT = C::GPT, KS = C::KS;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nzero; i += gridDim.x * blockDim.x) xch[i] = 0u;
  if (fold.out)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D * G * H; i += gridDim.x * blockDim.x) {
      const int d = i / (G * H), j = i - d * (G * H);
      float v = fold.b_ih[d][j];
      if (G != 3 || j < 2 * H) v += fold.b_hh[d][j];
      fold.out[i] = v;
    }
  const int64_t total = (int64_t)D * CC * 4 * NTILE * CF * 64;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), f = (int)((i >> 6) % CF), t = (int)((i / (64 * CF)) % NTILE);
    const int wave = (int)((i / (64 * CF * NTILE)) & 3), c = (int)((i / (64 * CF * NTILE * 4)) % CC);
    const int d = (int)(i / ((int64_t)64 * CF * NTILE * 4 * CC));
    const int col = lane & 15, kg = lane >> 4, q = f >> 1, plane = f & 1;
    const int gate = GPT * t + col / UW, unit = U * c + UW * wave + col % UW;
    const int k = 32 * (((U == 32 ? c : c >> 1) + q) % KS) + 8 * kg;
    const float* row = (d ? w1 : w0) + ((int64_t)gate * H + unit) * H + k;
    const bool rok = gate < G && unit < H;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bf16_t hi, lo;
      split_bf16(rok && k + e < H ? row[e] : 0.f, hi, lo);
      v[e] = __builtin_bit_cast(__bf16, plane ? lo : hi);
    }
    out[i] = v;
  }
  if (out_b) pack_bwd_body<G, CC, U>(w0, w1, out_b, D, H, xch_b, nzero_b);
}

template <int G, int CC, int U>
__global__ void rnnc_pack_bwd_kernel(const float* __restrict__ w0, const float* __restrict__ w1, bf16x8* __restrict__ out,
                                     int D, int H, u32* __restrict__ xch, int nzero) {
  pack_bwd_body<G, CC, U>(w0, w1, out, D, H, xch, nzero);
}

// ---------------------------------------------------------------------------------------------------------------
// forward recurrence
// ---------------------------------------------------------------------------------------------------------------
// grid: ncl * CC workgroups x 256 threads; block b -> cluster b % ncl (U = 32: ncl = 8, 16, ... <= MAXCL, cluster k on XCD
// k % 8; U = 16: 4 or 2, a cluster spans XCDs k, k + ncl, ...), member b / ncl.
// cluster k -> (sample group g0 + k / D, direction k % D).  Gate phase: lane = sample * UW + unit of the wave: it
// consumes its own wave's results (wave-local LDS exchange, no workgroup barrier); with 4-unit waves lanes 32-63 idle.
// Exchange layout: [slot][cluster][member][wave][sample][unit of the wave] words — a wave publishes NS * UW consecutive
// words with one store instruction; a reading thread takes four consecutive units of one (member, wave, sample).
template <int G, int CC, int U, int NSV>
__global__ __launch_bounds__(256, 1) void rnnc_fwd_kernel(
    float* __restrict__ gates, float* __restrict__ extra, float* __restrict__ y, const bf16x8* __restrict__ wpk,
    const float* __restrict__ bhh0, const float* __restrict__ bhh1, const float* __restrict__ h0,
    const float* __restrict__ c0, const int32_t* __restrict__ lens, u32* __restrict__ xch, int32_t* __restrict__ fault,
    int drop, int tune, int g0, int nclusters, int ncl, int B, int T, int D, int H) {
  using C = Cfg<G, CC, U, NSV>;
  constexpr int NS = C::NS, NJ = C::NJ, HSZ = 2 * C::NS * C::CLD;   // (this kernel's samples per cluster; bf16 per parity buffer of the state)
  constexpr int CLD = C::CLD, CF = C::CF, CF_A = C::CF_A, CF_REG = C::CF_REG, CF_L = C::CF_L, NL = C::NL, HP = C::HP;
  constexpr int NTILE = C::NTILE, UW = C::UW, GPT = C::GPT, KS = C::KS, XB = C::XB, TPB = C::TPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* hS = reinterpret_cast<bf16_t*>(smem);                                                    // [2][2 NS][CLD]: rows 0 .. NS-1 hi, NS .. 2 NS-1 lo
  bf16x8* Wl = reinterpret_cast<bf16x8*>(smem + (size_t)2 * HSZ * 2);                              // [4][NTILE][CF_L][64]
  float* S = reinterpret_cast<float*>(smem + (size_t)2 * HSZ * 2 + (size_t)4 * NTILE * CF_L * 1024);   // [4][NTILE][16][16]
  __shared__ int s_local;
  const int cluster = blockIdx.x % ncl, c = blockIdx.x / ncl;
  const int ks_own = U == 32 ? c : c >> 1;   // the k step this member's units sit in
  if (cluster >= nclusters) return;     // whole clusters leave together
  if (c == drop) return;                // test hook (lr_rnn_debug_drop_member): the others must time out and report
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = cluster % D, group = g0 + cluster / D;
  const int col = lane & 15, kg = lane >> 4;

  // ---- weights: NTILE * CF fragments per wave ------------------------------------------------------------------
  bf16x8 Wa[NTILE][CF_A], Wv[NTILE][imax(1, CF_REG - CF_A)];
  const bf16x8* wsrc = wpk + ((int64_t)((d * CC + c) * 4 + wave) * NTILE * CF) * 64 + lane;
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
#pragma unroll
    for (int f = 0; f < CF; ++f) {
      const bf16x8 w = wsrc[(t * CF + f) * 64];
      if (f < CF_A) Wa[t][f] = w;
      else if (f < CF_REG) Wv[t][f - CF_A] = w;
      else Wl[((wave * NTILE + t) * CF_L + (f - CF_REG)) * 64 + lane] = w;
    }
  }
  // ---- the state before step 0 (zero, or h0): hS[0], hi in rows 0-7, lo in rows 8-15, LOCAL k order ------------
  for (int i = tid; i < HSZ; i += 256) hS[HSZ + i] = 0;
  for (int i = tid; i < NS * CLD; i += 256) {
    const int r = i / CLD, pos = i - r * CLD;
    float v = 0.f;
    if (h0 && pos < HP) {
      int j = ks_own + (pos >> 5);
      if (j >= KS) j -= KS;
      const int k = 32 * j + (pos & 31), bb = group * NS + r;
      if (k < H && bb < B) v = h0[((int64_t)d * B + bb) * H + k];
    }
    bf16_t hi, lo;
    split_bf16(v, hi, lo);
    hS[r * CLD + pos] = hi;
    hS[(r + NS) * CLD + pos] = lo;
  }

  // ---- gate-phase role: one (sample, unit) per thread ----------------------------------------------------------
  // (16 samples per cluster: a lane runs NJ = 2 cells, samples sl0 and sl0 + 8 of its unit)
  const bool active = lane < 8 * UW;                // (4-unit waves: the upper half of the wave has no (sample, unit))
  const int sl0 = (lane / UW) & 7, u8 = lane % UW;  // sample within the group (+ 8 j), unit within the wave
  const int ul = UW * wave + u8;                    // member-local unit
  const int unit = U * c + ul;
  const int own_pos = ((U * c) & 31) + ul;          // its column in the own k step (local k step 0)
  int b[NJ], len[NJ];
  bool alive[NJ];
  float sreg[NJ];                                   // carried fp32 state of this (sample, unit): GRU h, LSTM c
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    b[j] = group * NS + sl0 + 8 * j;
    alive[j] = active && b[j] < B && unit < H;
    len[j] = alive[j] ? lens[b[j]] : 0;
    sreg[j] = 0.f;
    if (alive[j]) {
      if (G == 3 && h0) sreg[j] = h0[((int64_t)d * B + b[j]) * H + unit];
      if (G == 4 && c0) sreg[j] = c0[((int64_t)d * B + b[j]) * H + unit];
    }
  }
  const float bhn = (G == 3 && active && unit < H) ? (d ? bhh1 : bhh0)[2 * H + unit] : 0.f;
  struct Gx { float v[NJ][G]; };
  Gx gxA, gxB;   // pre-activations of even / odd steps, fetched TWO steps ahead
  auto time_of = [&](int s) {
    const int sc = s < T ? s : T - 1;
    return d == 0 ? sc : T - 1 - sc;
  };
  // (Measured and dropped, round 4: threads without a (sample, unit) fetching element (0, 0) instead of nothing, so that
  // the prefetch has counted waits only — <3,8> forward 111.0 -> 108.6 us, <4,24> 199.7 -> 202.1, backward 153.6 -> 153.9
  // / 277.1 -> 279.3: inside the pool's noise, and it costs a mask operand in the backward cell.)
  auto fetch_gx = [&](Gx& gx, int t) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (!alive[j]) continue;
      const float* gp = gates + (((int64_t)b[j] * T + t) * D + d) * (int64_t)(G * H) + unit;
#pragma unroll
      for (int g = 0; g < G; ++g) gx.v[j][g] = gp[(int64_t)g * H];
    }
  };
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int g = 0; g < G; ++g) gxA.v[j][g] = gxB.v[j][g] = 0.f;
  fetch_gx(gxA, time_of(0));
  fetch_gx(gxB, time_of(1));
  const int xcluster = CC * XB, xslot = nclusters * xcluster;       // words (32-bit: scalar multiplies)
  u32* xmine = xch + cluster * xcluster + c * XB + wave * (NS * UW) + lane;   // word (wave, sample, unit of the wave); + 64 j
  // what this thread gathers in sweep i: item g = 256 i + tid of the cluster's CC * TPB 16-byte items, i.e. member
  // g / TPB (32-unit members: 4 i + wave; 16-unit members: 8 i + 2 wave + (lane >> 5)), its words 4 w .. 4 w + 3 with
  // w = g % TPB: wave w / (2 NS) ... of the source, sample rsmp, member-local units rpos .. rpos + 3
  const int gw = tid % TPB, gj0 = tid / TPB;                            // item inside a block; member of sweep 0
  const u32* xbase = xch + cluster * xcluster + gj0 * XB + 4 * gw;      // + sweep * (256 / TPB) * XB
  const int rsmp = ((4 * gw) % (NS * UW)) / UW, rpos = UW * ((4 * gw) / (NS * UW)) + (4 * gw) % UW;
  unsigned pend0 = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int j = (256 / TPB) * i + gj0;
    if (j < CC && j != c) pend0 |= 1u << i;
  }
  int bad = 0;
  xcd_handshake<CC>(xch + 2 * xslot + cluster * CC, c, tid, &s_local, bad);   // (also: hS complete)
  const bool local = s_local != 0;
  const bool has_h0 = h0 != nullptr;

  auto step = [&](int s, Gx& gx) __attribute__((always_inline)) {
    const int t = time_of(s);
    const int tnext = time_of(s + 2);
    bf16_t* hcur = hS + (s & 1) * HSZ;
    bf16_t* hnxt = hS + ((s + 1) & 1) * HSZ;
    float sum[NJ][G];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < G; ++g) sum[j][g] = 0.f;
    if (s > 0 || has_h0) {
      f32x4 acc0[NTILE], acc1[NTILE];   // hi / lo weight plane: four accumulation chains per wave (eight — even / odd k
                                  // steps apart — measured 4 % slower: the chains are not what the product waits for)
      // ---- own member's k step (local q = 0): its operands are already in LDS (32-unit members; with 16-unit
      // members half of that k step is the partner's and arrives with the gather) ------------------------------------
      // (16 samples: `col` is the SAMPLE, the hi plane in rows 0-15 and the lo plane in rows 16-31 of the state; every
      // product below is issued for both, into the same accumulator)
      constexpr int LO = 16 * CLD;
      if constexpr (U == 32) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(hcur + col * CLD + kg * 8);
        LR_MFMA_A0(acc0[0], a, Wa[0][0]);
        LR_MFMA_A0(acc0[1], a, Wa[1][0]);
        LR_MFMA_A0(acc1[0], a, Wa[0][1]);
        LR_MFMA_A0(acc1[1], a, Wa[1][1]);
        if constexpr (NSV == 16) {
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(hcur + LO + col * CLD + kg * 8);
          LR_MFMA_A(acc0[0], al, Wa[0][0]);
          LR_MFMA_A(acc0[1], al, Wa[1][0]);
          LR_MFMA_A(acc1[0], al, Wa[0][1]);
          LR_MFMA_A(acc1[1], al, Wa[1][1]);
        }
      }
      if (s > 0) {
        // ---- the other members' h_{s-1} (tag of step s-1, slot (s-1) & 1): asked for once the own k step is under
        // way; every sweep that is still missing a word is asked for again IN PARALLEL (a serial re-poll costs a
        // memory round trip each).  (opaque slot offset: otherwise the addresses of BOTH parity slots are hoisted
        // out of the step loop and held in registers the weight fragments want)
        int slot_off = __builtin_amdgcn_readfirstlane(((s - 1) & 1) * xslot);
        asm volatile("" : "+s"(slot_off));
        const u32* xp = xbase + slot_off;
        const u32 tg = tag_of(s - 1);
        // (a thread that has given up issues nothing: peek4's loads are invisible to the compiler, and the loop
        // below — the only place that drains them — does not run for it)
        // (tuning knobs, lr_rnn_debug_tune: sleeps before the first poll / between rounds.  Swept on the MI355X: a
        // first-poll delay only costs — the first poll is usually answered —, one sleep between rounds is best by 1-3 %.
        // Also measured and dropped: issuing the step's HBM traffic (next step's pre-activation loads, y / extra /
        // gates stores) right AFTER the gather instead of before the publish, so that the polls would not queue
        // behind it in the in-order return path — GRU-256 got 11 % slower: that traffic hides in the exchange wait
        // where it is, and its address arithmetic then sits between the gather and the product.)
        for (int w = 0; w < (tune & 0xff); ++w) __builtin_amdgcn_s_sleep(1);
        u32x4 g[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          g[i] = (u32x4){0u, 0u, 0u, 0u};
          if (((pend0 >> i) & 1u) && !bad) g[i] = peek4(xp + i * (256 / TPB) * XB);
        }
        unsigned pend = pend0;
        for (int round = 0; pend && !bad; ++round) {
          LR_VM_DRAIN();
#pragma unroll
          for (int i = 0; i < NL; ++i) LR_TOUCH(g[i]);
#pragma unroll
          for (int i = 0; i < NL; ++i) {
            if ((pend >> i) & 1u) {
              const u32x4 v = g[i];
              if ((((v[0] ^ tg) | (v[1] ^ tg) | (v[2] ^ tg) | (v[3] ^ tg)) & 3u) == 0u) {
                // where member j's units sit in the LOCAL k order: k step (U j) / 32 relative to the own one
                const int j = (256 / TPB) * i + gj0;
                int q = ((U * j) >> 5) - ks_own;
                if (q < 0) q += KS;
                const int pos = 32 * q + ((U * j) & 31) + rpos;
                u32 hi0, lo0, hi1, lo1;
                split_bf16_pair(xval(v[0]), xval(v[1]), hi0, lo0);
                split_bf16_pair(xval(v[2]), xval(v[3]), hi1, lo1);
                *reinterpret_cast<uint2*>(hcur + rsmp * CLD + pos) = make_uint2(hi0, hi1);
                *reinterpret_cast<uint2*>(hcur + (rsmp + NS) * CLD + pos) = make_uint2(lo0, lo1);
                pend &= ~(1u << i);
              }
            }
          }
          if (!pend) break;
          if (round > SPIN_LIMIT) {
            bad = 1;
            break;
          }
          for (int w = 0; w < ((tune >> 8) & 0xff); ++w) __builtin_amdgcn_s_sleep(1);
#pragma unroll
          for (int i = 0; i < NL; ++i)
            if ((pend >> i) & 1u) g[i] = peek4(xp + i * (256 / TPB) * XB);
        }
        lr_lds_barrier();
      }
      if constexpr (NSV == 8) {
        constexpr int Q0 = U == 32 ? 1 : 0;    // first k step behind the gather
        bf16x8 a_next = *reinterpret_cast<const bf16x8*>(hcur + col * CLD + 32 * Q0 + kg * 8);
  #pragma unroll
        for (int q = Q0; q < KS; ++q) {
          const bf16x8 a = a_next;
          if (q + 1 < KS) a_next = *reinterpret_cast<const bf16x8*>(hcur + col * CLD + (q + 1) * 32 + kg * 8);
          const int f0 = 2 * q, f1 = 2 * q + 1;
          if (U == 16 && q == 0) {   // (16-unit members: nothing was accumulated in front of the gather)
  #pragma unroll
            for (int t = 0; t < NTILE; ++t) {
              LR_MFMA_A0(acc0[t], a, Wa[t][0]);
              LR_MFMA_A0(acc1[t], a, Wa[t][1]);
            }
          } else if (f1 < CF_A) {
  #pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_A(acc0[t], a, Wa[t][f0]);
  #pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_A(acc1[t], a, Wa[t][f1]);
          } else if (f1 < CF_REG) {
  #pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc0[t], a, Wv[t][f0 - CF_A]);
  #pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc1[t], a, Wv[t][f1 - CF_A]);
          } else {
            bf16x8 w0[NTILE], w1[NTILE];
  #pragma unroll
            for (int t = 0; t < NTILE; ++t) {
              w0[t] = Wl[((wave * NTILE + t) * CF_L + (f0 - CF_REG)) * 64 + lane];
              w1[t] = Wl[((wave * NTILE + t) * CF_L + (f1 - CF_REG)) * 64 + lane];
            }
  #pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc0[t], a, w0[t]);
  #pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc1[t], a, w1[t]);
          }
        }
      } else {
        // 16 samples: every fragment meets the hi rows and the lo rows of the state (four MFMAs on other accumulators
        // between two on the same one)
        bf16x8 ah_next = *reinterpret_cast<const bf16x8*>(hcur + col * CLD + 32 + kg * 8);
        bf16x8 al_next = *reinterpret_cast<const bf16x8*>(hcur + LO + col * CLD + 32 + kg * 8);
#pragma unroll
        for (int q = 1; q < KS; ++q) {
          const bf16x8 ah = ah_next, al = al_next;
          if (q + 1 < KS) {
            ah_next = *reinterpret_cast<const bf16x8*>(hcur + col * CLD + (q + 1) * 32 + kg * 8);
            al_next = *reinterpret_cast<const bf16x8*>(hcur + LO + col * CLD + (q + 1) * 32 + kg * 8);
          }
          const int f0 = 2 * q, f1 = 2 * q + 1;
          if (f1 < CF_A) {
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_A(acc0[t], ah, Wa[t][f0]);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_A(acc1[t], ah, Wa[t][f1]);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_A(acc0[t], al, Wa[t][f0]);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_A(acc1[t], al, Wa[t][f1]);
          } else if (f1 < CF_REG) {
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc0[t], ah, Wv[t][f0 - CF_A]);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc1[t], ah, Wv[t][f1 - CF_A]);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc0[t], al, Wv[t][f0 - CF_A]);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc1[t], al, Wv[t][f1 - CF_A]);
          } else {
            bf16x8 w0[NTILE], w1[NTILE];
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
              w0[t] = Wl[((wave * NTILE + t) * CF_L + (f0 - CF_REG)) * 64 + lane];
              w1[t] = Wl[((wave * NTILE + t) * CF_L + (f1 - CF_REG)) * 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc0[t], ah, w0[t]);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc1[t], ah, w1[t]);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc0[t], al, w0[t]);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) LR_MFMA_V(acc1[t], al, w1[t]);
          }
        }
      }
      LR_MFMA_DRAIN();
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        LR_ACC_READY(acc0[t]);
        LR_ACC_READY(acc1[t]);
      }
      // tile (wave, t): S[row 4 kg + r][col] ; rows 0-7 = state hi of samples 0-7, rows 8-15 = state lo (16 samples: row = sample)
      float* Sw = S + wave * NTILE * 256;
#pragma unroll
      for (int t2 = 0; t2 < NTILE; ++t2)
#pragma unroll
        for (int r = 0; r < 4; ++r) Sw[t2 * 256 + (4 * kg + r) * 16 + col] = acc0[t2][r] + acc1[t2][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local exchange
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float* Sg = Sw + (g / GPT) * 256 + (g % GPT) * UW + u8;
        if constexpr (NSV == 8) {
          sum[0][g] = Sg[sl0 * 16] + Sg[(sl0 + 8) * 16];
        } else {
#pragma unroll
          for (int j = 0; j < NJ; ++j) sum[j][g] = Sg[(sl0 + 8 * j) * 16];
        }
      }
    }
    // ---- the cell (torch gate order: GRU r, z, n; LSTM i, f, g, o) ---------------------------------------------
    float hv[NJ], exv[NJ];
    float go[NJ][G];
    bool livev[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool live = alive[j] && t < len[j];
      livev[j] = live;
      float h, ex;
      if (G == 3) {
        const float hn = sum[j][2] + bhn;
        const float r = fast_sigmoid(gx.v[j][0] + sum[j][0]);
        const float z = fast_sigmoid(gx.v[j][1] + sum[j][1]);
        const float n = fast_tanh(gx.v[j][2] + r * hn);
        h = live ? (1.f - z) * n + z * sreg[j] : 0.f;
        go[j][0] = r;
        go[j][1] = z;
        go[j][2] = n;
        ex = live ? hn : 0.f;
      } else {
        const float ig = fast_sigmoid(gx.v[j][0] + sum[j][0]);
        const float fg = fast_sigmoid(gx.v[j][1] + sum[j][1]);
        const float gg = fast_tanh(gx.v[j][2] + sum[j][2]);
        const float og = fast_sigmoid(gx.v[j][G - 1] + sum[j][G - 1]);
        const float cn = live ? fg * sreg[j] + ig * gg : 0.f;
        h = live ? og * fast_tanh(cn) : 0.f;
        go[j][0] = ig;
        go[j][1] = fg;
        go[j][2] = gg;
        go[j][G - 1] = og;
        ex = cn;
        sreg[j] = cn;
      }
      hv[j] = h;
      exv[j] = ex;
    }
    fetch_gx(gx, tnext);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const u32 w = xword(hv[j], tag_of(s));
      if (active) publish(xmine + (s & 1) * xslot + 64 * j, w, local);      // first: the other members are waiting for it
      const float h = xval(w);                         // the state everyone uses, this member included
      hv[j] = h;
      if (G == 3) sreg[j] = h;
      bf16_t hi, lo;
      split_bf16(h, hi, lo);
      if (active) {
        hnxt[(sl0 + 8 * j) * CLD + own_pos] = hi;      // local k position of the own member: q = 0
        hnxt[(sl0 + 8 * j + NS) * CLD + own_pos] = lo;
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (alive[j]) {
        const int64_t bt = (int64_t)b[j] * T + t;
        y[bt * ((int64_t)D * H) + d * H + unit] = hv[j];
        extra[(bt * D + d) * H + unit] = exv[j];
        if (livev[j]) {
          float* gout = gates + (bt * D + d) * (int64_t)(G * H) + unit;
#pragma unroll
          for (int g = 0; g < G; ++g) gout[(int64_t)g * H] = go[j][g];
        }
      }
    }
    lr_lds_barrier();   // hnxt's own k step complete; hcur free for the next gather
  };
  // One full vmcnt wait in front of the step loop.  The loads before the loop that the loop's body uses (bhn, sreg:
  // issued under `if (alive)`) are never provably complete for hipcc's wait-count pass, so the FIRST use in every
  // iteration — `sum[2] + bhn` in the even step's cell — got a vmcnt(0), right BEHIND the even step's prefetch of the
  // pre-activations of step s + 2 (hipcc hoists those loads above the cell): every second step waited out a fresh HBM /
  // MALL load before it published its state, i.e. before the other members of the cluster could go on.  With the wait
  // here the prefetch's latency runs under the exchange.  MEASURED (round 4, same box, layer pass): <3,8> 119.8 -> 111.0
  // us in the pixel step, 114.1 -> 104.6 in regime R; <4,24> 207.3 -> 199.7.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), lgkmcnt / expcnt untouched
  for (int s = 0; s < T; s += 2) {
    step(s, gxA);
    if (s + 1 < T) step(s + 1, gxB);
  }
  if (bad && fault) atomicOr(fault, 1);
}

// ---------------------------------------------------------------------------------------------------------------
// backward recurrence (row-split, see the file header)
// ---------------------------------------------------------------------------------------------------------------
// Per step a member contracts its OWN dG (KB blocks x 32 units x 8 samples, bf16 hi/lo in rows 0-7 / 8-15 of the A
// operand, one k step per block) against its rows of W_hh for ALL HP output units (2 CC column tiles; wave w owns
// those of the destination members w, w + 4, ...), folds the hi/lo rows, and publishes the partial dh of a
// destination's 8 x 32 (sample, unit) values as ONE store instruction: a lane's four values (two tiles x two rows)
// are 16 contiguous bytes of the 1 KB block [slot][cluster][dst member][src member][lane][tile half][row].  The
// gather mirrors the forward's: thread (wave ww, lane l) takes the 16 bytes of lane l from the blocks of source
// members ww, ww + 4, ... and adds them up (fixed order); the four waves' sums and the member's own partial meet in
// LDS, where each thread picks up the five values of its own (sample, unit) and runs the cell backward
// (rnn_bwd_step_kernel's arithmetic).
template <int G, int CC, int U, int NSV>
__global__ __launch_bounds__(256, 1) void rnnc_bwd_kernel(
    const float* __restrict__ gates, const float* __restrict__ extra, const float* __restrict__ y,
    const float* __restrict__ dy, const float* __restrict__ dh_n, const float* __restrict__ dc_n, float* __restrict__ dG,
    float* __restrict__ dh0, float* __restrict__ dc0, const float* __restrict__ h0, const float* __restrict__ c0,
    const bf16x8* __restrict__ wpk, const int32_t* __restrict__ lens, u32* __restrict__ xch, int32_t* __restrict__ fault,
    int drop, int tune, int g0, int nclusters, int ncl, int B, int T, int D, int H) {
  using C = Cfg<G, CC, U, NSV>;
  constexpr int NS = C::NS, NJ = C::NJ, GSZ = 2 * C::NS * C::BKLD;   // (samples per cluster; cells per thread; bf16 per parity buffer of dG)
  constexpr int KB = C::KB, NT = C::NT, NL = C::NL, BFW = C::BFW, BFW_A = C::BFW_A, BFW_REG = C::BFW_REG, BFW_L = C::BFW_L,
                BKLD = C::BKLD;
  constexpr int XB = C::XB, TPB = C::TPB, TPD = C::TPD, ND = C::ND, VPL = C::VPL, NRED = C::NRED;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* gS = reinterpret_cast<bf16_t*>(smem);                                                    // [2][2 NS][BKLD]: rows 0 .. NS-1 hi, NS .. 2 NS-1 lo
  bf16x8* Wl = reinterpret_cast<bf16x8*>(smem + (size_t)2 * GSZ * 2);                              // [4][NT][BFW_L][64]
  float* red = reinterpret_cast<float*>(smem + (size_t)2 * GSZ * 2 + (size_t)4 * NT * BFW_L * 1024);   // [NRED + 1][XB]: the sums of remote partials per (wave, block of the sweep), then this member's own
  __shared__ int s_local;
  const int cluster = blockIdx.x % ncl, c = blockIdx.x / ncl;
  if (cluster >= nclusters) return;
  if (c == drop) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = cluster % D, group = g0 + cluster / D;
  const int col = lane & 15, kg = lane >> 4;
  const int64_t DH = (int64_t)D * H;

  bf16x8 Wa[NT][imax(1, BFW_A)], Wv[NT][imax(1, BFW_REG - BFW_A)];
  const bf16x8* wsrc = wpk + ((int64_t)((d * CC + c) * 4 + wave) * NT * BFW) * 64 + lane;
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) {
#pragma unroll
    for (int f = 0; f < BFW; ++f) {
      const bf16x8 w = wsrc[(tl * BFW + f) * 64];
      if (f < BFW_A) Wa[tl][f] = w;
      else if (f < BFW_REG) Wv[tl][f - BFW_A] = w;
      else Wl[((wave * NT + tl) * BFW_L + (f - BFW_REG)) * 64 + lane] = w;
    }
  }
  for (int i = tid; i < 2 * GSZ; i += 256) gS[i] = 0;

  // ---- one (sample, unit) per thread: sample = tid / U, unit = tid % U of this member (16-unit members: threads
  // 128-255 have none) ---------------------------------------------------------------------------------------------
  // (16 samples per cluster: a thread runs NJ = 2 cells, samples sl0 and sl0 + 8 of its unit)
  const bool active = tid < 8 * U;
  const int sl0 = (tid / U) & 7, ul = tid % U;
  const int unit = U * c + ul;
  int b[NJ], len[NJ];
  bool alive[NJ];
  float inj_h[NJ], inj_c[NJ], car[NJ];   // car: GRU dh_{t'} * z_{t'}; LSTM dc_{t'} * f_{t'} of the step processed before
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    b[j] = group * NS + sl0 + 8 * j;
    alive[j] = active && b[j] < B && unit < H;
    len[j] = alive[j] ? lens[b[j]] : 0;
    inj_h[j] = (alive[j] && dh_n) ? dh_n[((int64_t)d * B + b[j]) * H + unit] : 0.f;
    inj_c[j] = (G == 4 && alive[j] && dc_n) ? dc_n[((int64_t)d * B + b[j]) * H + unit] : 0.f;
    car[j] = 0.f;
  }
  struct In1 { float dy, g[G], ex, prev; };
  struct In { In1 v[NJ]; };
  In inA, inB;       // operands of even / odd steps, fetched TWO steps ahead
  auto time_of = [&](int s) {
    const int sc = s < T ? s : T - 1;
    return d == 0 ? T - 1 - sc : sc;
  };
  auto fetch = [&](In& inn, int t) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      In1& in = inn.v[j];
      in.dy = in.ex = in.prev = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) in.g[g] = 0.f;
      if (!alive[j]) continue;
      const int tp = d == 0 ? t - 1 : t + 1;
      const int64_t bt = (int64_t)b[j] * T + t;
      in.dy = dy[bt * DH + d * H + unit];
      const float* gi = gates + (bt * D + d) * (int64_t)(G * H) + unit;
#pragma unroll
      for (int g = 0; g < G; ++g) in.g[g] = gi[(int64_t)g * H];
      in.ex = extra[(bt * D + d) * H + unit];
      if (tp >= 0 && tp < T) {
        const int64_t btp = (int64_t)b[j] * T + tp;
        in.prev = G == 3 ? y[btp * DH + d * H + unit] : extra[(btp * D + d) * H + unit];
      } else {
        const float* src = G == 3 ? h0 : c0;     // the state before the first step (decoder loop), else zero
        if (src) in.prev = src[((int64_t)d * B + b[j]) * H + unit];
      }
    }
  };
  fetch(inA, time_of(0));
  fetch(inB, time_of(1));
  const int xdst = CC * XB, xcluster = CC * xdst, xslot = nclusters * xcluster;   // words (32-bit: scalar multiplies)
  u32* xout = xch + cluster * xcluster + c * XB + VPL * lane;         // + dst * xdst
  // gather item g = 256 i + tid of the CC * TPB 16-byte items addressed to this member: source member g / TPB
  const int gw = tid % TPB, gj0 = tid / TPB;
  const u32* xin = xch + cluster * xcluster + c * xdst + gj0 * XB + 4 * gw;    // + sweep * (256 / TPB) * XB
  // where this thread's (sample sl, unit ul) sits in a block: lane kg * 16 + col, word 2 * (tile of the destination) + row
  // (16 samples: the accumulator rows ARE the samples — lane (sl >> 2) * 16 + col, word 4 * tile + (sl & 3))
  int rpos[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int sl = sl0 + 8 * j;
    rpos[j] = NSV == 8 ? ((((sl >> 1) & 1) * 2 + (sl >> 2)) * 16 + (ul & 15)) * VPL + (ul >> 4) * 2 + (sl & 1)
                       : ((sl >> 2) * 16 + (ul & 15)) * VPL + (ul >> 4) * 4 + (sl & 3);
  }
  unsigned pend0 = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int j = (256 / TPB) * i + gj0;
    if (j < CC && j != c) pend0 |= 1u << i;
  }
  int bad = 0;
  xcd_handshake<CC>(xch + 2 * xslot + cluster * CC, c, tid, &s_local, bad);   // (also: gS cleared)
  const bool local = s_local != 0;

  // W_hh^T dG of the step processed before step s, for this thread's (sample, unit): dG sits in gS[s & 1]
  auto reduce = [&](int s, float (&prod)[NJ]) __attribute__((always_inline)) {
    const bf16_t* gcur = gS + (s & 1) * GSZ;    // rows 0-7 hi, 8-15 lo (16 samples: rows 0-15 hi, 16-31 lo)
    f32x4 acc[NT];
    bf16x8 a_next = *reinterpret_cast<const bf16x8*>(gcur + col * BKLD + kg * 8);
    [[maybe_unused]] bf16x8 al_next = a_next;
    if constexpr (NSV == 16) al_next = *reinterpret_cast<const bf16x8*>(gcur + (16 + col) * BKLD + kg * 8);
#pragma unroll
    for (int ks = 0; ks < KB; ++ks) {
      const bf16x8 a = a_next;
      [[maybe_unused]] const bf16x8 al = al_next;
      if (ks + 1 < KB) {
        a_next = *reinterpret_cast<const bf16x8*>(gcur + col * BKLD + (ks + 1) * 32 + kg * 8);
        if constexpr (NSV == 16) al_next = *reinterpret_cast<const bf16x8*>(gcur + (16 + col) * BKLD + (ks + 1) * 32 + kg * 8);
      }
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const int f = 2 * ks + pl;
        // (16 samples: every fragment meets the hi rows and then the lo rows of dG, NT MFMAs apart on an accumulator)
        if (f == 0) {
#pragma unroll
          for (int tl = 0; tl < NT; ++tl) LR_MFMA_A0(acc[tl], a, Wa[tl][0]);
          if constexpr (NSV == 16) {
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) LR_MFMA_A(acc[tl], al, Wa[tl][0]);
          }
        } else if (f < BFW_A) {
#pragma unroll
          for (int tl = 0; tl < NT; ++tl) LR_MFMA_A(acc[tl], a, Wa[tl][f]);
          if constexpr (NSV == 16) {
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) LR_MFMA_A(acc[tl], al, Wa[tl][f]);
          }
        } else if (f < BFW_REG) {
#pragma unroll
          for (int tl = 0; tl < NT; ++tl) LR_MFMA_V(acc[tl], a, Wv[tl][f - BFW_A]);
          if constexpr (NSV == 16) {
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) LR_MFMA_V(acc[tl], al, Wv[tl][f - BFW_A]);
          }
        } else {   // LDS-resident fragments, up to six tiles at a time (all at once costs 4 NT registers)
#pragma unroll
          for (int t0 = 0; t0 < NT; t0 += 6) {
            bf16x8 wl[6];
#pragma unroll
            for (int i = 0; i < 6; ++i)
              if (t0 + i < NT) wl[i] = Wl[((wave * NT + t0 + i) * BFW_L + (f - BFW_REG)) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 6; ++i)
              if (t0 + i < NT) LR_MFMA_V(acc[t0 + i], a, wl[i]);
            if constexpr (NSV == 16) {
#pragma unroll
              for (int i = 0; i < 6; ++i)
                if (t0 + i < NT) LR_MFMA_V(acc[t0 + i], al, wl[i]);
            }
          }
        }
      }
    }
    LR_MFMA_DRAIN();
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) LR_ACC_READY(acc[tl]);
    // rows 4 kg + r: rows 0-7 (kg 0, 1) came from the hi plane of dG, rows 8-15 (kg 2, 3) from the lo plane of the
    // same samples: fold across lanes +-32.  Afterwards both halves hold sample 4 (kg & 1) + r of the tile's unit
    // col; the lower half keeps r = 0, 1, the upper half r = 2, 3: sample 4 (kg & 1) + 2 (kg >> 1) + row.
    int slot_off = __builtin_amdgcn_readfirstlane(((s - 1) & 1) * xslot);
    asm volatile("" : "+s"(slot_off));
    u32* xo = xout + slot_off;
    const u32 tg = tag_of(s - 1);
#pragma unroll
    for (int m = 0; m < ND; ++m) {
      const int dstm = wave + 4 * m;
      float v[VPL];
#pragma unroll
      for (int th = 0; th < TPD; ++th) {
        // v_permlane32_swap (gfx950) trades the upper half of one register for the lower half of another: with a[0] and
        // a[2] that leaves {a0.lo, a2.lo} and {a0.hi, a2.hi}, whose sum IS the fold — lanes 0-31 hold a[0] + the a[0] of
        // lane + 32, lanes 32-63 a[2] + the a[2] of lane - 32 — in two VALU instructions per kept value (rounds 2-3: four
        // ds_bpermute + four adds + two selects per tile through the LDS pipe, half of them for values nobody keeps)
        const f32x4 a = acc[TPD * m + th];
        if constexpr (NSV == 8) {
          v[2 * th] = fold32(a[0], a[2]);
          v[2 * th + 1] = fold32(a[1], a[3]);
        } else {   // 16 samples: rows 4 kg + r ARE samples 4 kg + r; nothing to fold
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * th + r] = a[r];
        }
      }
      if (dstm < CC) {
        if (dstm == c) {
#pragma unroll
          for (int e = 0; e < VPL; ++e) red[NRED * XB + VPL * lane + e] = v[e];
        } else {
          // one store per lane: the wave writes the destination's whole block.  Each WORD is valid on its own, so it
          // does not matter whether the bytes land together.  (s_nop behind the 16-byte stores: the wait state a VALU
          // write to a store's data registers needs behind it, which hipcc cannot place behind an asm — lr_rnn_grid.hip
          // store4 tells how that was found.)
          u32* p = xo + dstm * xdst;
          if constexpr (VPL == 4 || VPL == 8) {
#pragma unroll
            for (int q4 = 0; q4 < VPL / 4; ++q4) {
              const u32x4 w = {xword(v[4 * q4], tg), xword(v[4 * q4 + 1], tg), xword(v[4 * q4 + 2], tg), xword(v[4 * q4 + 3], tg)};
              if (local) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" ::"v"(p + 4 * q4), "v"(w) : "memory");   // workgroup scope
              else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p + 4 * q4), "v"(w) : "memory");   // agent scope
            }
          } else {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 w = {xword(v[0], tg), xword(v[1], tg)};
            if (local) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(w) : "memory");
            else asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
          }
        }
      }
    }
    // ---- gather: the 16 bytes of this lane from the blocks of source members wave, wave + 4, ... ------------------
    const u32* xp = xin + slot_off;
    for (int w = 0; w < (tune & 0xff); ++w) __builtin_amdgcn_s_sleep(1);
    u32x4 g[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      g[i] = (u32x4){0u, 0u, 0u, 0u};
      if (((pend0 >> i) & 1u) && !bad) g[i] = peek4(xp + i * (256 / TPB) * XB);   // (see the forward kernel)
    }
    unsigned pend = pend0;
    for (int round = 0; pend && !bad; ++round) {
      LR_VM_DRAIN();
#pragma unroll
      for (int i = 0; i < NL; ++i) LR_TOUCH(g[i]);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        if ((pend >> i) & 1u) {
          const u32x4 v = g[i];
          if ((((v[0] ^ tg) | (v[1] ^ tg) | (v[2] ^ tg) | (v[3] ^ tg)) & 3u) == 0u) pend &= ~(1u << i);
        }
      }
      if (!pend) break;
      if (round > SPIN_LIMIT) {
        bad = 1;
        break;
      }
      for (int w = 0; w < ((tune >> 8) & 0xff); ++w) __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int i = 0; i < NL; ++i)
        if ((pend >> i) & 1u) g[i] = peek4(xp + i * (256 / TPB) * XB);
    }
    {   // this thread's sum over the source members of its sweeps, in FIXED order; absent sweeps hold zeros
      float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
      if (!bad) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          if ((pend0 >> i) & 1u) {
            p0 += xval(g[i][0]);
            p1 += xval(g[i][1]);
            p2 += xval(g[i][2]);
            p3 += xval(g[i][3]);
          }
        }
      }
      *reinterpret_cast<float4*>(red + gj0 * XB + 4 * gw) = make_float4(p0, p1, p2, p3);
    }
    lr_lds_barrier();     // `red` complete
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      prod[j] = red[NRED * XB + rpos[j]];
#pragma unroll
      for (int w = 0; w < NRED; ++w) prod[j] += red[w * XB + rpos[j]];
    }
  };

  auto step = [&](int s, In& inn) __attribute__((always_inline)) {
    const int t = time_of(s);
    const int tnext = time_of(s + 2);
    bf16_t* gnxt = gS + ((s + 1) & 1) * GSZ;
    float prod[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) prod[j] = 0.f;
    if (s > 0) reduce(s, prod);
    if (s == T) {   // past the last step (only with dh0): the gradient into the initial state (lr_rnn_dh0's arithmetic)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (alive[j]) {
          dh0[((int64_t)d * B + b[j]) * H + unit] = G == 3 ? prod[j] + car[j] : prod[j];
          if (G == 4 && dc0) dc0[((int64_t)d * B + b[j]) * H + unit] = car[j];
        }
      return;
    }
    float kv[NJ][G], dgv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const In1& in = inn.v[j];
      float dh = in.dy + prod[j];
      const bool is_last = d == 0 ? (t == len[j] - 1) : (t == 0);   // where the final state was read
      if (is_last) dh += inj_h[j];
      const bool live = alive[j] && t < len[j];
      if (G == 3) {
        // rnn_bwd_step_kernel<3>
        dh += car[j];   // dh_{t+1} * z_{t+1}
        const float r = in.g[0], z = in.g[1], n = in.g[2], hn = in.ex, hp = in.prev;
        const float dn_pre = live ? dh * (1.f - z) * (1.f - n * n) : 0.f;
        const float dr_pre = dn_pre * hn * r * (1.f - r);
        const float dz_pre = live ? dh * (hp - n) * z * (1.f - z) : 0.f;
        car[j] = live ? dh * z : 0.f;
        dgv[j][0] = dr_pre;
        dgv[j][1] = dz_pre;
        dgv[j][2] = dn_pre;
        dgv[j][3] = dn_pre * r;
        kv[j][0] = dr_pre;
        kv[j][1] = dz_pre;
        kv[j][2] = dn_pre * r;    // recurrent path of the n gate: d/d(W_hn h + b_hn)
      } else {
        // rnn_bwd_step_kernel<4>
        const float ig = in.g[0], fg = in.g[1], gg = in.g[2], og = in.g[G - 1], ct = in.ex, cp = in.prev;
        float dc = car[j];
        if (is_last) dc += inj_c[j];
        float di = 0.f, df = 0.f, dg_ = 0.f, do_ = 0.f;
        car[j] = 0.f;
        if (live) {
          const float tc = fast_tanh(ct);
          dc += dh * og * (1.f - tc * tc);
          di = dc * gg * ig * (1.f - ig);
          df = dc * cp * fg * (1.f - fg);
          dg_ = dc * ig * (1.f - gg * gg);
          do_ = dh * tc * og * (1.f - og);
          car[j] = dc * fg;
        }
        dgv[j][0] = di;
        dgv[j][1] = df;
        dgv[j][2] = dg_;
        dgv[j][3] = do_;
        kv[j][0] = di;
        kv[j][1] = df;
        kv[j][2] = dg_;
        kv[j][G - 1] = do_;
      }
    }
    fetch(inn, tnext);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int sl = sl0 + 8 * j;
      if (active) {
#pragma unroll
        for (int k = 0; k < G; ++k) {   // gate k's entry of this (sample, unit): k step (k U) / 32 of the member's dG
          bf16_t hi, lo;
          split_bf16(kv[j][k], hi, lo);
          gnxt[sl * BKLD + k * U + ul] = hi;
          gnxt[(sl + NS) * BKLD + k * U + ul] = lo;
        }
      }
      if (alive[j]) {
        float* dgo = dG + (((int64_t)b[j] * T + t) * D + d) * (int64_t)(4 * H) + unit;
#pragma unroll
        for (int k = 0; k < 4; ++k) dgo[(int64_t)k * H] = dgv[j][k];
      }
    }
    lr_lds_barrier();   // gnxt complete; `red` free again
  };
  const int nsteps = dh0 ? T + 1 : T;   // with dh0: one more product + reduce-scatter after the last step
  for (int s = 0; s < nsteps; s += 2) {
    step(s, inA);
    if (s + 1 < nsteps) step(s + 1, inB);
  }
  if (bad && fault) atomicOr(fault, 1);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
size_t xch_words(int CC, int U, int ns, int nclusters, int backward) {
  const size_t xb = (size_t)ns * U;
  const size_t per = backward ? (size_t)CC * CC * xb : (size_t)CC * xb;
  return (size_t)2 * nclusters * per + (size_t)nclusters * CC;
}

// clusters a launch is laid out for (its grid is ncl * CC workgroups): the smallest multiple of XS that holds `need`
// clusters, at most MAXCL, and no more than the device has compute units for (all of them must be resident together)
template <int G, int CC, int U>
inline int launch_clusters(int need) {
  using C = Cfg<G, CC, U>;
  int m = (need + C::XS - 1) / C::XS * C::XS;
  if (m > C::MAXCL) m = C::MAXCL;
  const int cus = lr_device_cus();
  while (m > C::XS && cus > 0 && m * CC > cus) m -= C::XS;
  return m;
}
inline int max_clusters(int CC, int U) {   // Cfg::MAXCL without the template
  const int xs = CC <= 32 ? 8 : (CC <= 64 ? 4 : 2);
  return (U == 32 && CC <= 16) ? xs * (32 / CC) : xs;
}
// samples per cluster of a pass: 16 where the shape has the form (32-unit members, CC >= 17: one cluster per XCD) AND the
// batch would take more than one launch of 8-sample clusters; 8 otherwise (and under the test hook, bit 4 of
// lr_rnn_debug_disable_cluster, for the A/B)
constexpr bool has_ns16(int CC, int U) { return U == 32 && CC >= 17 && CC <= 24; }   // (25 .. 27 GRU members: the state of 16 samples does not fit the LDS beside the fragments)
inline int pick_ns(int CC, int U, int B, int D) {
  return has_ns16(CC, U) && !lr_debug_ns8() && (B + 7) / 8 * D > max_clusters(CC, U) ? 16 : 8;
}

// words of the first launch's exchange area
inline int first_xch_words(int CC, int U, int ns, int maxcl, int B, int D, int backward) {
  const int groups = (B + ns - 1) / ns, gchunk = maxcl / D;
  return (int)xch_words(CC, U, ns, (groups < gchunk ? groups : gchunk) * D, backward);
}

// the layer's prologue: W_hh -> fragments, exchange words of the first launch cleared, biases folded (b_ih may be
// NULL: nothing to fold)
template <int G, int CC, int U>
int fwd_prologue(const float* const* w_hh, const float* const* b_ih, const float* const* b_hh, float* bias_out, void* wpack,
                 void* xch, int B, int D, int H, hipStream_t stream, void* wpack_b = nullptr, void* xch_b = nullptr) {
  FoldPtrs fold;
  for (int d = 0; d < 2; ++d) {
    fold.b_ih[d] = b_ih ? b_ih[d < D ? d : 0] : nullptr;
    fold.b_hh[d] = b_hh ? b_hh[d < D ? d : 0] : nullptr;
  }
  fold.out = b_ih ? bias_out : nullptr;
  const int ns = pick_ns(CC, U, B, D), ncl = launch_clusters<G, CC, U>((B + ns - 1) / ns * D);
  LR_LAUNCH((rnnc_pack_fwd_kernel<G, CC, U>), dim3(1024), dim3(256), 0, stream, w_hh[0], w_hh[D - 1], (bf16x8*)wpack, D, H,
            (u32*)xch, first_xch_words(CC, U, ns, ncl, B, D, 0), fold, (bf16x8*)wpack_b, (u32*)xch_b,
            wpack_b ? first_xch_words(CC, U, ns, ncl, B, D, 1) : 0);
  return lr_launch_status();
}

template <int G, int CC, int U, int NSV>
int fwd_launch_ns(float* gates, float* extra, float* y, const float* const* b_hh, const float* h0, const float* c0,
                  const int32_t* lens, void* wpack, void* xch, int B, int T, int D, int H, hipStream_t stream) {
  using C = Cfg<G, CC, U, NSV>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rnnc_fwd_kernel<G, CC, U, NSV>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)C::FWD_LDS) != hipSuccess)
      return LR_ERR_LAUNCH;
    attr_set = true;
  }
  int st = LR_OK;
  int32_t* fault = lr_fault_words();
  const int drop = lr_debug_drop_member_value();
  const int tune = lr_debug_tune_value(0);
  const int groups = (B + NSV - 1) / NSV, ncl = launch_clusters<G, CC, U>(groups * D), gchunk = ncl / D;   // sample groups per launch
  for (int g0 = 0; g0 < groups; g0 += gchunk) {
    const int ng = groups - g0 < gchunk ? groups - g0 : gchunk, nclusters = ng * D;
    // (the first launch's words were cleared by the prologue)
    if (g0 > 0 && hipMemsetAsync(xch, 0, xch_words(CC, U, NSV, nclusters, 0) * sizeof(u32), stream) != hipSuccess) return LR_ERR_LAUNCH;
    const dim3 grid(ncl * CC);
    hipEvent_t e0, e1;
    if (g0 == 0 && lr_prof_next(LR_PROF_RNN_FWD, &e0, &e1))
      hipExtLaunchKernelGGL((rnnc_fwd_kernel<G, CC, U, NSV>), grid, dim3(256), C::FWD_LDS, stream, e0, e1, 0, gates, extra, y,
                            (const bf16x8*)wpack, b_hh[0], b_hh[D - 1], h0, c0, lens, (u32*)xch, fault, drop, tune, g0,
                            nclusters, ncl, B, T, D, H);
    else
      hipLaunchKernelGGL((rnnc_fwd_kernel<G, CC, U, NSV>), grid, dim3(256), C::FWD_LDS, stream, gates, extra, y,
                         (const bf16x8*)wpack, b_hh[0], b_hh[D - 1], h0, c0, lens, (u32*)xch, fault, drop, tune, g0, nclusters,
                         ncl, B, T, D, H);
    st = lr_launch_status();
    if (st != LR_OK) return st;
  }
  return LR_OK;
}

template <int G, int CC, int U>
int fwd_launch(float* gates, float* extra, float* y, const float* const* w_hh, const float* const* b_hh, const float* h0,
               const float* c0, const int32_t* lens, void* wpack, void* xch, int B, int T, int D, int H,
               hipStream_t stream, int prologue_done) {
  lr_clear_error();
  if (!prologue_done) {
    const int st = fwd_prologue<G, CC, U>(w_hh, nullptr, nullptr, nullptr, wpack, xch, B, D, H, stream);
    if (st != LR_OK) return st;
  }
  if constexpr (has_ns16(CC, U)) {
    if (pick_ns(CC, U, B, D) == 16)
      return fwd_launch_ns<G, CC, U, 16>(gates, extra, y, b_hh, h0, c0, lens, wpack, xch, B, T, D, H, stream);
  }
  return fwd_launch_ns<G, CC, U, 8>(gates, extra, y, b_hh, h0, c0, lens, wpack, xch, B, T, D, H, stream);
}

template <int G, int CC, int U, int NSV>
int bwd_launch_ns(const float* gates, const float* extra, const float* y, const float* dy, const float* dh_n, const float* dc_n,
                  float* dG, float* dh0, float* dc0, const float* h0, const float* c0, const int32_t* lens, void* wpack,
                  void* xch, int B, int T, int D, int H, hipStream_t stream) {
  using C = Cfg<G, CC, U, NSV>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rnnc_bwd_kernel<G, CC, U, NSV>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)C::BWD_LDS) != hipSuccess)
      return LR_ERR_LAUNCH;
    attr_set = true;
  }
  int st = LR_OK;
  int32_t* fault = lr_fault_words();
  const int drop = lr_debug_drop_member_value();
  const int tune = lr_debug_tune_value(1);
  const int groups = (B + NSV - 1) / NSV, ncl = launch_clusters<G, CC, U>(groups * D), gchunk = ncl / D;
  for (int g0 = 0; g0 < groups; g0 += gchunk) {
    const int ng = groups - g0 < gchunk ? groups - g0 : gchunk, nclusters = ng * D;
    if (g0 > 0 && hipMemsetAsync(xch, 0, xch_words(CC, U, NSV, nclusters, 1) * sizeof(u32), stream) != hipSuccess) return LR_ERR_LAUNCH;
    const dim3 grid(ncl * CC);
    hipEvent_t e0, e1;
    if (g0 == 0 && lr_prof_next(LR_PROF_RNN_BWD, &e0, &e1))
      hipExtLaunchKernelGGL((rnnc_bwd_kernel<G, CC, U, NSV>), grid, dim3(256), C::BWD_LDS, stream, e0, e1, 0, gates, extra, y, dy,
                            dh_n, dc_n, dG, dh0, dc0, h0, c0, (const bf16x8*)wpack, lens, (u32*)xch, fault, drop, tune, g0,
                            nclusters, ncl, B, T, D, H);
    else
      hipLaunchKernelGGL((rnnc_bwd_kernel<G, CC, U, NSV>), grid, dim3(256), C::BWD_LDS, stream, gates, extra, y, dy, dh_n, dc_n,
                         dG, dh0, dc0, h0, c0, (const bf16x8*)wpack, lens, (u32*)xch, fault, drop, tune, g0, nclusters, ncl, B, T,
                         D, H);
    st = lr_launch_status();
    if (st != LR_OK) return st;
  }
  return LR_OK;
}

template <int G, int CC, int U>
int bwd_launch(const float* gates, const float* extra, const float* y, const float* dy, const float* dh_n, const float* dc_n,
               float* dG, float* dh0, float* dc0, const float* h0, const float* c0, const float* const* w_hh,
               const int32_t* lens, void* wpack, void* xch, int B, int T, int D, int H, hipStream_t stream, int pack_done) {
  lr_clear_error();
  const int ns = pick_ns(CC, U, B, D);
  if (!pack_done) {   // (else: the layer's forward prologue left the fragments and the cleared words in wpack / xch)
    LR_LAUNCH((rnnc_pack_bwd_kernel<G, CC, U>), dim3(1024), dim3(256), 0, stream, w_hh[0], w_hh[D - 1], (bf16x8*)wpack, D,
              H, (u32*)xch, first_xch_words(CC, U, ns, launch_clusters<G, CC, U>((B + ns - 1) / ns * D), B, D, 1));
    const int st = lr_launch_status();
    if (st != LR_OK) return st;
  }
  if constexpr (has_ns16(CC, U)) {
    if (ns == 16)
      return bwd_launch_ns<G, CC, U, 16>(gates, extra, y, dy, dh_n, dc_n, dG, dh0, dc0, h0, c0, lens, wpack, xch, B, T, D, H, stream);
  }
  return bwd_launch_ns<G, CC, U, 8>(gates, extra, y, dy, dh_n, dc_n, dG, dh0, dc0, h0, c0, lens, wpack, xch, B, T, D, H, stream);
}

// the instantiated (gates, members, units per member) triples — EVERY hidden size the kernels' storage holds:
//   32-unit members (a cluster = ceil(H / 32) CUs of one XCD)
//     GRU   1 .. 27 members  H <= 864   (GRU-800: config/train/micro; the decoder GRU-512 behind a BiGRU-256)
//     LSTM  1 .. 24 members  H <= 768   (LSTM-700: config/defaults.txt; LSTM-512: config/train/attn/attention_type;
//                                        LSTM-768: config/archive/experiments/ecd)
//   What stops there is the member's slice of W_hh (G x 32 rows x HP columns x hi + lo planes), which must stay in the
//   registers and LDS of ONE compute unit: 28 GRU members would take 164 KB of LDS for the forward fragments, 25 LSTM
//   members 186 KB for the backward ones (the backward keeps 60 fragments per wave in AGPRs, 14 in VGPRs — 198 of 256
//   VGPRs at 24 members —, the rest in LDS).
//   16-unit members (round 4; an even number of them, 50 .. 72: a cluster spans two or more XCDs, agent-scope exchange)
//     LSTM  768 < H <= 1152  (LSTM-800; LSTM-1024 = the decoder of config/train/attn/attention_type's BiLSTM-512,
//                             better_model.py:134-148: with the reference's batch of 4 ONE cluster of 64 CUs)
//     GRU   864 < H <= 1152
//   Larger layers (the 1400 / 1536-unit decoders behind BiLSTM-700 / 768 encoders) exceed a CU's storage with 16-unit
//   members too (the backward's accumulators and VGPR-resident fragments): they run the step kernels (lr_rnn.hip), and
//   lipreading_amd.encoder / attention_decoder say so once.
#define LR_CLUSTER_CC_COMMON(X, g) \
  X(g, 1, 32) X(g, 2, 32) X(g, 3, 32) X(g, 4, 32) X(g, 5, 32) X(g, 6, 32) X(g, 7, 32) X(g, 8, 32) X(g, 9, 32) X(g, 10, 32) \
  X(g, 11, 32) X(g, 12, 32) X(g, 13, 32) X(g, 14, 32) X(g, 15, 32) X(g, 16, 32) X(g, 17, 32) X(g, 18, 32) X(g, 19, 32)     \
  X(g, 20, 32) X(g, 21, 32) X(g, 22, 32) X(g, 23, 32) X(g, 24, 32)
#define LR_CLUSTER_CC16_COMMON(X, g) \
  X(g, 56, 16) X(g, 58, 16) X(g, 60, 16) X(g, 62, 16) X(g, 64, 16) X(g, 66, 16) X(g, 68, 16) X(g, 70, 16) X(g, 72, 16)
#define LR_CLUSTER_SHAPES(X)                                                                                     \
  LR_CLUSTER_CC_COMMON(X, 3) X(3, 25, 32) X(3, 26, 32) X(3, 27, 32) LR_CLUSTER_CC_COMMON(X, 4)                    \
  LR_CLUSTER_CC16_COMMON(X, 3) X(4, 50, 16) X(4, 52, 16) X(4, 54, 16) LR_CLUSTER_CC16_COMMON(X, 4)
#define X(g, c, u)                                                                                               \
  static_assert(Cfg<g, c, u>::FWD_LDS <= 160 * 1024 && Cfg<g, c, u>::BWD_LDS <= 160 * 1024, "a member's fragments fit one CU"); \
  static_assert(!has_ns16(c, u) || (Cfg<g, c, 32, 16>::FWD_LDS <= 160 * 1024 && Cfg<g, c, 32, 16>::BWD_LDS <= 160 * 1024), \
                "... with the state / dG of 16 samples too");
LR_CLUSTER_SHAPES(X)
#undef X

// (members, units per member) of a hidden size; false: no cluster recurrence
bool resolve_shape(int G, int H, int* cc, int* u) {
  const int c32 = (H + 31) / 32;
  if (c32 <= (G == 3 ? 27 : 24)) {
    *cc = c32;
    *u = 32;
    return true;
  }
  int c16 = (H + 15) / 16;
  c16 += c16 & 1;
  if (c16 >= (G == 3 ? 56 : 50) && c16 <= 72) {
    *cc = c16;
    *u = 16;
    return true;
  }
  return false;
}

}  // namespace

// all ncl * CC workgroups of a launch must be resident together, one per compute unit; the smallest launch (XS
// clusters) is what a device must hold — larger batches get as many clusters per launch as it has room for
int lr_rnn_cluster_cus(int G, int H) {
  int cc, u;
  if (lr_rnn_grid_shape(G, H)) return lr_rnn_grid_cus();   // 1152 < H <= 1536, LSTM: one grid of 192 (lr_rnn_grid.hip)
  if (H < 1 || (G != 3 && G != 4) || !resolve_shape(G, H, &cc, &u)) return 0;
  return (cc <= 32 ? 8 : (cc <= 64 ? 4 : 2)) * cc;
}

// recurrence launches one pass of a layer takes (both directions ride in one launch): ceil(sample groups / groups a
// launch's clusters hold); 0: no kernel for the shape
int lr_rnn_cluster_launches(int G, int B, int H, int D) {
  int cc, u;
  if (B >= 1 && D >= 1 && lr_rnn_grid_shape(G, H)) return lr_rnn_grid_launches(B, D);
  if (B < 1 || D < 1 || !resolve_shape(G, H, &cc, &u)) return 0;
  int ncl = 0;
  const int ns = pick_ns(cc, u, B, D);
#define X(g, c, uu) \
  if (G == g && cc == c && u == uu) ncl = launch_clusters<g, c, uu>((B + ns - 1) / ns * D);
  LR_CLUSTER_SHAPES(X)
#undef X
  if (ncl < D) return 0;
  const int groups = (B + ns - 1) / ns, gchunk = ncl / D;
  return (groups + gchunk - 1) / gchunk;
}

int lr_rnn_cluster_supported(int G, int B, int H) {
  if (B < 1 || lr_debug_cluster_disabled()) return 0;
  const int need = lr_rnn_cluster_cus(G, H);
  return need > 0 && lr_device_cus() >= need ? 1 : 0;
}

size_t lr_rnn_cluster_pack_bytes(int G, int H, int D, int backward) {
  int cc, u;
  if (lr_rnn_grid_shape(G, H)) return lr_rnn_grid_pack_bytes(D);
  if (!resolve_shape(G, H, &cc, &u)) return 0;
#define X(g, c, uu) \
  if (G == g && cc == c && u == uu) return (size_t)D * (backward ? Cfg<g, c, uu>::BWD_PACK : Cfg<g, c, uu>::FWD_PACK);
  LR_CLUSTER_SHAPES(X)
#undef X
  return 0;
}

size_t lr_rnn_cluster_xch_bytes(int B, int H, int D, int backward) {
  int cc, u;
  if (lr_rnn_grid_shape(4, H)) return lr_rnn_grid_xch_bytes(B, backward);   // (a GRU of this size has no one-launch kernel)
  // (the exchange area does not depend on the gate count: resolve as the LSTM, whose 32-unit range is the smaller)
  if (!resolve_shape(4, H, &cc, &u) && !resolve_shape(3, H, &cc, &u)) return 0;
  // (the largest of: 8-sample clusters, and 16-sample ones where the shape and the batch take them — the test hook may
  // switch between the two after the buffers were sized)
  auto words = [&](int c_, int u_) {
    size_t best = 0;
    for (int ns = 8; ns <= (has_ns16(c_, u_) && (B + 7) / 8 * D > max_clusters(c_, u_) ? 16 : 8); ns += 8) {
      const int maxcl = max_clusters(c_, u_);
      int clusters = (B + ns - 1) / ns * D;
      if (clusters > maxcl) clusters = maxcl;
      const size_t w_ = xch_words(c_, u_, ns, clusters, backward);
      if (w_ > best) best = w_;
    }
    return best;
  };
  size_t w = words(cc, u);
  int cc3, u3;
  if (resolve_shape(3, H, &cc3, &u3) && (cc3 != cc || u3 != u)) {   // a GRU of this size takes the other form: the larger
    const size_t w3 = words(cc3, u3);
    if (w3 > w) w = w3;
  }
  return w * sizeof(u32);
}

int lr_rnn_cluster_forward(int G, float* gates, float* extra, float* y, const float* const* w_hh, const float* const* b_hh,
                           const float* h0, const float* c0, const int32_t* lens, void* wpack, void* xch, int B, int T,
                           int D, int H, hipStream_t stream, int prologue_done) {
  int cc, u;
  if (lr_rnn_grid_shape(G, H))
    return lr_rnn_grid_forward(gates, extra, y, w_hh, h0, c0, lens, wpack, xch, B, T, D, H, stream, prologue_done);
  if (!resolve_shape(G, H, &cc, &u)) return LR_ERR_UNSUPPORTED;
#define X(g, c, uu)                 \
  if (G == g && cc == c && u == uu) \
    return fwd_launch<g, c, uu>(gates, extra, y, w_hh, b_hh, h0, c0, lens, wpack, xch, B, T, D, H, stream, prologue_done);
  LR_CLUSTER_SHAPES(X)
#undef X
  return LR_ERR_UNSUPPORTED;
}

int lr_rnn_cluster_prologue(int G, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                            float* bias_out, void* wpack, void* xch, int B, int D, int H, hipStream_t stream,
                            void* wpack_b, void* xch_b) {
  int cc, u;
  lr_clear_error();
  if (lr_rnn_grid_shape(G, H)) return lr_rnn_grid_prologue(w_hh, b_ih, b_hh, bias_out, wpack, xch, B, D, H, stream, wpack_b, xch_b);
  if (!resolve_shape(G, H, &cc, &u)) return LR_ERR_UNSUPPORTED;
#define X(g, c, uu) \
  if (G == g && cc == c && u == uu) \
    return fwd_prologue<g, c, uu>(w_hh, b_ih, b_hh, bias_out, wpack, xch, B, D, H, stream, wpack_b, xch_b);
  LR_CLUSTER_SHAPES(X)
#undef X
  return LR_ERR_UNSUPPORTED;
}

int lr_rnn_cluster_backward(int G, const float* gates, const float* extra, const float* y, const float* dy,
                            const float* dh_n, const float* dc_n, float* dG, float* dh0, float* dc0, const float* h0,
                            const float* c0, const float* const* w_hh, const int32_t* lens, void* wpack, void* xch, int B,
                            int T, int D, int H, hipStream_t stream, int pack_done) {
  int cc, u;
  if (lr_rnn_grid_shape(G, H))
    return lr_rnn_grid_backward(gates, extra, dy, dh_n, dc_n, dG, dh0, dc0, c0, w_hh, lens, wpack, xch, B, T, D, H, stream, pack_done);
  if (!resolve_shape(G, H, &cc, &u)) return LR_ERR_UNSUPPORTED;
#define X(g, c, uu)                                                                                                      \
  if (G == g && cc == c && u == uu)                                                                                      \
    return bwd_launch<g, c, uu>(gates, extra, y, dy, dh_n, dc_n, dG, dh0, dc0, h0, c0, w_hh, lens, wpack, xch, B, T, D, H, \
                                stream, pack_done);
  LR_CLUSTER_SHAPES(X)
#undef X
  return LR_ERR_UNSUPPORTED;
}
