// lr_rnn_cluster.hip — the LSTM-768 recurrence of the REFERENCE-FAITHFUL regime as ONE launch per layer
// pass, fp32-faithful (config/archive/experiments/ecd/*: BiLSTM, hidden 768; BASELINE configs[2]).
// Reference arithmetic: better_model.py:47-49,74 (nn.LSTM fp32), the same cell as lr_rnn.hip's step
// kernels, whose interface buffers (gates in/out, extra = c, y, dG) it shares.
//
// The step kernels re-stream W_hh (9.4 MB per direction, fp32) from the fabric at every one of the 150
// steps of a training step (1.42 GB, 8.8-9.4 us per step).  Here W_hh never moves: it is split into bf16
// hi + lo planes (the fp32 product to ~1e-6, all four cross terms of (h_hi + h_lo)(W_hi + W_lo) with fp32
// accumulation, as in lr_rnn_pair.hip) and SLICED over a cluster of 24 compute units per (direction, group
// of 8 samples): member c owns hidden units [32c, 32c + 32) — 128 gate rows of W_hh x 768 x 2 planes =
// 384 KB = 96 MFMA B fragments per wave, 60 in AGPRs the matrix core reads directly, 24-36 in VGPRs, the
// rest in LDS.  The 8 samples of a group ride in rows 0-7 (state hi) and 8-15 (state lo) of the 16-row
// MFMA A operand: rows r and r + 8 of the accumulator sum to the full product of sample r.
//
//   forward   gates[c's rows] += W_hh[c's rows][:] h_{t-1}: every member needs the WHOLE previous state:
//             each step a member publishes its 32 x 8 new state values and gathers the other 23 members'
//             (47 KB of {value, tag} granules per member per step).
//   backward  dh_{t-1} = W_hh^T dG_t is computed ROW-split: a member contracts its OWN 128 gate rows (its
//             own dG: nothing to gather first) into a partial dh for all 768 units, publishes the 23 x 256
//             partial sums the other members own and gathers the 23 partials of its own units — the same
//             47 KB per member per step instead of the 188 KB a column split would gather.
// Granules are 8-byte {fp32 value, tag = step + 1} written by ONE agent-scope store and polled by
// agent-scope loads (MI355X_MICROARCH.md "handoff-1to1"/"allgather": data-tagged granules need no fence);
// a cluster is blocks {k, k + 8, k + 16, ...}, which the dispatcher places on ONE XCD (a speed matter
// only).  Two parity slots: a member overwrites slot s & 1 at step s + 2 only after it consumed every
// other member's step-(s + 1) data, each of which was published after its author read slot s & 1.
// 8 clusters x 24 members = 192 workgroups per launch, one per CU; waits are bounded (lr_rnn_pair_errors).
//
// XCD-local exchange: every member publishes its XCC id (HW_REG_XCC_ID) once; when all 24 members of a cluster
// share an XCD (the dispatcher's placement of blocks k, k + 8, ...), granules are stored at workgroup scope
// (`sc0`: they stay in that XCD's L2, where the other members' agent-scope loads find them) instead of
// agent-scope `sc1` stores that drop the line and send every reader to the fabric.
//
// MEASURED (round 2, MI355X, B = 32, T = 75, BiLSTM-768): forward 344 us per layer pass (4.6 us per step; step
// kernels 9.4 us per step), backward 414 us (5.5 us per step; step kernels 10.7) — the training step 2.19 ->
// 1.59 ms, both passes 1e-6 of the step kernels.  Where a step's time goes (forward, pieces switched off):
// ~3 us is the exchange itself (8-byte granule reads), ~1.1 us waiting for tags, 0.8 us of MFMAs.
#include "lr_common.h"
#include <hip/hip_ext.h>

namespace {

__device__ int g_cluster_err;       // members that gave up waiting, since lr_cluster_errors() last read it

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;
typedef unsigned long long u64;

constexpr int CH = 768;            // hidden size
constexpr int CG = 4;              // gates i, f, g, o
constexpr int CC = 24;             // members per cluster
constexpr int CU_ = 32;            // hidden units per member
constexpr int NS = 8;              // samples per cluster (rows 0-7 hi, 8-15 lo of the A operand)
constexpr int CKS = CH / 32;       // 24 k steps of 32 = one per source member; LOCAL order: own member first, then c+1, ...
constexpr int CLD = CH + 8;        // bf16 per LDS row of the state
constexpr int SPIN_LIMIT = 1 << 18;
constexpr int MAX_CLUSTERS = 8;    // per launch: 192 workgroups

constexpr int CF = 2 * CKS;        // 48 fragments per column tile: f = 2 * local k step + plane
constexpr int CF_A = 30;           // f < 30 in AGPRs (2 tiles x 30 = 60 fragments)
constexpr int CF_REG = 44;         // f < 44 in registers (2 x 14 = 28 fragments in VGPRs); 44..47 in LDS (32 KB)
constexpr size_t CFWD_LDS = (size_t)2 * 16 * CLD * 2 + (size_t)4 * 2 * (CF - CF_REG) * 1024 + (size_t)4 * 2 * 256 * 4;

__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}
__device__ __forceinline__ void publish(u64* p, float v, int tag) {
  __hip_atomic_store(p, ((u64)(unsigned)tag << 32) | (u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// the same granule with a store that stops at the writer's XCD L2 and stays there (a write-through `sc1` store
// drops the line, so every reader then goes to the fabric).  Only other CUs of the SAME XCD are guaranteed to see
// it there — used when a cluster verified at run time that all its members share one XCD.  (Workgroup scope is
// the ISA's `sc0`: through the CU's write-through L1 into L2.  NOT a `volatile` store: the compiler makes that a
// system-scope store followed by s_waitcnt vmcnt(0) — 24 serialised round trips per step in the backward kernel.)
__device__ __forceinline__ void publish_local(u64* p, float v, int tag) {
  __hip_atomic_store(p, ((u64)(unsigned)tag << 32) | (u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int xcc_id() {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 0xf;
}
__device__ __forceinline__ u64 peek(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float await(const u64* p, u64 first, int tag, int* err) {
  u64 g = first;
  int n = 0;
  if (*err) return 0.f;
  while ((int)(g >> 32) != tag) {
    __builtin_amdgcn_s_sleep(1);
    g = peek(p);
    if (++n > SPIN_LIMIT) {
      *err = 1;
      return 0.f;
    }
  }
  return __builtin_bit_cast(float, (unsigned)(g & 0xffffffffu));
}

// W_hh [4*768][768] fp32 of each direction -> bf16 hi/lo MFMA B fragments of the forward product:
// out[((((d*CC + c)*4 + wave)*2 + t)*CF + f)*64 + lane] = plane f & 1 of W_hh[gate*768 + unit][k .. k+7],
// gate = 2t + (col >> 3), unit = 32c + 8 wave + (col & 7), k = 32 ((c + (f >> 1)) % 24) + 8 kg.
__global__ void lstm768_pack_whh_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                        bf16x8* __restrict__ out, int D) {
  const int64_t total = (int64_t)D * CC * 4 * 2 * CF * 64;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), f = (int)((i >> 6) % CF), t = (int)((i / (64 * CF)) & 1);
    const int wave = (int)((i / (64 * CF * 2)) & 3), c = (int)((i / (64 * CF * 8)) % CC), d = (int)(i / ((int64_t)64 * CF * 8 * CC));
    const int col = lane & 15, kg = lane >> 4, q = f >> 1, plane = f & 1;
    const int gate = 2 * t + (col >> 3), unit = CU_ * c + 8 * wave + (col & 7);
    const int k = 32 * ((c + q) % CC) + 8 * kg;
    const float* row = (d ? w1 : w0) + ((int64_t)gate * CH + unit) * CH + k;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bf16_t hi, lo;
      split_bf16(row[e], hi, lo);
      v[e] = __builtin_bit_cast(__bf16, plane ? lo : hi);
    }
    out[i] = v;
  }
}

#define LR_CMFMA2_FIRST(acc, a, w0, w1)                                                                         \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, 0\n\t"                                                     \
               "v_mfma_f32_16x16x32_bf16 %1, %2, %4, 0"                                                         \
               : "=&v"(acc[0]), "=&v"(acc[1])                                                                   \
               : "v"(a), "a"(w0), "a"(w1))
#define LR_CMFMA2(acc, a, WC, w0, w1)                                                                           \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %1, %2, %4, %1"                                                        \
               : "+v"(acc[0]), "+v"(acc[1])                                                                     \
               : "v"(a), WC(w0), WC(w1))

// grid: 8 * CC workgroups x 256 threads; block b -> cluster b % 8 (= its XCD), member b / 8.
// cluster k -> (sample group k / D, direction k % D).  Wave w owns the member's units 8w .. 8w+7 as two
// column tiles (t = 0: gates i, f; t = 1: gates g, o; column = (gate & 1) * 8 + unit).  Gate phase: lane
// = sample * 8 + unit: every lane is busy and consumes its own wave's results (wave-local LDS exchange).
__global__ __launch_bounds__(256, 1) void lstm768_fwd_cluster_kernel(
    float* __restrict__ gates, float* __restrict__ extra, float* __restrict__ y, const bf16x8* __restrict__ wpk,
    const int32_t* __restrict__ lens, u64* __restrict__ xch, int g0, int nclusters, int B, int T, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* hS = reinterpret_cast<bf16_t*>(smem);                                                   // [2][16][CLD]
  bf16x8* Wl = reinterpret_cast<bf16x8*>(smem + (size_t)2 * 16 * CLD * 2);                        // [4][2][CF-CF_REG][64]
  float* S = reinterpret_cast<float*>(smem + (size_t)2 * 16 * CLD * 2 + (size_t)4 * 2 * (CF - CF_REG) * 1024);   // [4][2][16][16]
  const int cluster = blockIdx.x & 7, c = blockIdx.x >> 3;
  if (cluster >= nclusters) return;     // whole clusters leave together
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = cluster % D, group = g0 + cluster / D;
  const int col = lane & 15, kg = lane >> 4;

  // ---- weights: 96 fragments per wave ----------------------------------------------------------------------
  bf16x8 Wa[2][CF_A], Wv[2][CF_REG - CF_A];
  const bf16x8* wsrc = wpk + ((int64_t)((d * CC + c) * 4 + wave) * 2 * CF) * 64 + lane;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int f = 0; f < CF; ++f) {
      const bf16x8 w = wsrc[(t * CF + f) * 64];
      if (f < CF_A) Wa[t][f] = w;
      else if (f < CF_REG) Wv[t][f - CF_A] = w;
      else Wl[((wave * 2 + t) * (CF - CF_REG) + (f - CF_REG)) * 64 + lane] = w;
    }
  }
  for (int i = tid; i < 2 * 16 * CLD; i += 256) hS[i] = 0;

  // ---- gate-phase role: one (sample, unit) per thread ---------------------------------------------------
  const int sl = lane >> 3, u8 = lane & 7;          // sample within the group, unit within the wave
  const int ul = 8 * wave + u8;                     // member-local unit
  const int unit = CU_ * c + ul;
  const int b = group * NS + sl;
  const bool alive = b < B;
  const int len = alive ? lens[b] : 0;
  float creg = 0.f;
  struct Gx { float v[4]; };
  Gx gxA, gxB;   // pre-activations of even / odd steps, fetched TWO steps ahead
  auto time_of = [&](int s) {
    const int sc = s < T ? s : T - 1;
    return d == 0 ? sc : T - 1 - sc;
  };
  auto fetch_gx = [&](Gx& gx, int t) {
    if (!alive) return;
    const float* gp = gates + (((int64_t)b * T + t) * D + d) * (CG * CH) + unit;
#pragma unroll
    for (int g = 0; g < CG; ++g) gx.v[g] = gp[g * CH];
  };
#pragma unroll
  for (int g = 0; g < CG; ++g) gxA.v[g] = gxB.v[g] = 0.f;
  fetch_gx(gxA, time_of(0));
  fetch_gx(gxB, time_of(1));
  // exchange: [slot][cluster][member][sample][unit 32]; a thread publishes index sl*32 + ul of its member and
  // reads index tid of every other member (-> sample tid >> 5, unit tid & 31)
  const int64_t xmember = NS * CU_, xcluster = (int64_t)CC * xmember, xslot = (int64_t)nclusters * xcluster;
  u64* xmine = xch + cluster * xcluster + c * xmember + sl * CU_ + ul;
  const u64* xbase = xch + cluster * xcluster + tid;
  const int rs = tid >> 5, ru = tid & 31;           // what this thread gathers: sample row, unit of the source member
  int bad = 0;
  // ---- do all 24 members of this cluster sit on one XCD?  (observed: block b runs on XCD b % 8 — never
  // guaranteed.)  Every member publishes its XCC id (agent scope, tag 1) and reads all 24; the verdict is the
  // same on every member because it is computed from the same 24 words.
  __shared__ int s_local;
  {
    u64* xid = xch + 2 * xslot + (int64_t)cluster * CC;     // after the two parity slots
    if (tid == 0) {
      s_local = 1;
      publish(xid + c, __builtin_bit_cast(float, xcc_id()), 1);
    }
    __syncthreads();
    if (tid < CC) {
      u64 g = peek(xid + tid);
      int n = 0;
      while ((int)(g >> 32) != 1 && n++ < SPIN_LIMIT) {
        __builtin_amdgcn_s_sleep(2);
        g = peek(xid + tid);
      }
      if ((int)(g >> 32) != 1) bad = 1;
      if ((int)(g & 0xf) != xcc_id() || bad) s_local = 0;
    }
  }
  __syncthreads();
  const bool local = s_local != 0;

  auto step = [&](int s, Gx& gx) {
    const int t = time_of(s);
    const int tnext = time_of(s + 2);
    bf16_t* hcur = hS + (s & 1) * 16 * CLD;
    bf16_t* hnxt = hS + ((s + 1) & 1) * 16 * CLD;
    float sum[CG] = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      // (opaque slot offset: otherwise the 23 per-lane gather addresses of BOTH parity slots are hoisted out of the
      // step loop and held in registers — the registers CF_REG wants for weight fragments)
      int64_t slot_off = ((s - 1) & 1) * xslot;
      asm volatile("" : "+s"(slot_off));
      const u64* xp = xbase + slot_off;
      f32x4 acc0[2], acc1[2];     // hi / lo weight plane
      // ---- own member's k step (local q = 0): its operands are already in LDS ------------------------------------
      {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(hcur + col * CLD + kg * 8);
        LR_CMFMA2_FIRST(acc0, a, Wa[0][0], Wa[1][0]);
        LR_CMFMA2_FIRST(acc1, a, Wa[0][1], Wa[1][1]);
      }
      // the other 23 members' h_{s-1} (tag s, slot (s-1) & 1): asked for once the own k step is under way —
      // a poll issued right behind the publish finds nothing yet and only costs a round
      u64 g[CC - 1];
#pragma unroll
      for (int q = 1; q < CC; ++q) {
        int j = c + q;
        if (j >= CC) j -= CC;
        g[q - 1] = peek(xp + j * xmember);
      }
      // ---- the rest of the state -> LDS rows rs (hi) / rs + 8 (lo), local position 32 q + ru ------------------
      // rounds of polls: every granule that is still missing is asked for again IN PARALLEL (a serial
      // re-poll per granule costs a memory round trip each: 7 us per step instead of ~3)
      unsigned pend = (1u << (CC - 1)) - 1;
      for (int round = 0; pend && !bad; ++round) {
#pragma unroll
        for (int q = 1; q < CC; ++q) {
          if ((pend >> (q - 1)) & 1u) {
            const u64 v = g[q - 1];
            if ((int)(v >> 32) == s) {
              bf16_t hi, lo;
              split_bf16(__builtin_bit_cast(float, (unsigned)(v & 0xffffffffu)), hi, lo);
              hcur[rs * CLD + 32 * q + ru] = hi;
              hcur[(rs + 8) * CLD + 32 * q + ru] = lo;
              pend &= ~(1u << (q - 1));
            }
          }
        }
        if (!pend) break;
        if (round > SPIN_LIMIT) {
          bad = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
#pragma unroll
        for (int q = 1; q < CC; ++q) {
          if ((pend >> (q - 1)) & 1u) {
            int j = c + q;
            if (j >= CC) j -= CC;
            g[q - 1] = peek(xp + j * xmember);
          }
        }
      }
      lr_lds_barrier();
      bf16x8 a_next = *reinterpret_cast<const bf16x8*>(hcur + col * CLD + 32 + kg * 8);
#pragma unroll
      for (int q = 1; q < CKS; ++q) {
        const bf16x8 a = a_next;
        if (q + 1 < CKS) a_next = *reinterpret_cast<const bf16x8*>(hcur + col * CLD + (q + 1) * 32 + kg * 8);
        const int f0 = 2 * q, f1 = 2 * q + 1;
        if (f1 < CF_A) {
          LR_CMFMA2(acc0, a, "a", Wa[0][f0], Wa[1][f0]);
          LR_CMFMA2(acc1, a, "a", Wa[0][f1], Wa[1][f1]);
        } else if (f1 < CF_REG) {
          LR_CMFMA2(acc0, a, "v", Wv[0][f0 - CF_A], Wv[1][f0 - CF_A]);
          LR_CMFMA2(acc1, a, "v", Wv[0][f1 - CF_A], Wv[1][f1 - CF_A]);
        } else {
          const bf16x8 w00 = Wl[((wave * 2 + 0) * (CF - CF_REG) + (f0 - CF_REG)) * 64 + lane];
          const bf16x8 w10 = Wl[((wave * 2 + 1) * (CF - CF_REG) + (f0 - CF_REG)) * 64 + lane];
          const bf16x8 w01 = Wl[((wave * 2 + 0) * (CF - CF_REG) + (f1 - CF_REG)) * 64 + lane];
          const bf16x8 w11 = Wl[((wave * 2 + 1) * (CF - CF_REG) + (f1 - CF_REG)) * 64 + lane];
          LR_CMFMA2(acc0, a, "v", w00, w10);
          LR_CMFMA2(acc1, a, "v", w01, w11);
        }
      }
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
      // tile (wave, t): S[row 4 kg + r][col] ; rows 0-7 = state hi of samples 0-7, rows 8-15 = state lo
      float* Sw = S + wave * 2 * 256;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 4; ++r) Sw[t2 * 256 + (4 * kg + r) * 16 + col] = acc0[t2][r] + acc1[t2][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local exchange
#pragma unroll
      for (int g = 0; g < CG; ++g) {
        const float* Sg = Sw + (g >> 1) * 256 + (g & 1) * 8 + u8;
        sum[g] = Sg[sl * 16] + Sg[(sl + 8) * 16];
      }
    }
    // ---- LSTM cell (torch gate order i, f, g, o) ----------------------------------------------------------------
    const bool live = alive && t < len;
    const float ig = lr_sigmoid(gx.v[0] + sum[0]);
    const float fg = lr_sigmoid(gx.v[1] + sum[1]);
    const float gg = tanhf(gx.v[2] + sum[2]);
    const float og = lr_sigmoid(gx.v[3] + sum[3]);
    fetch_gx(gx, tnext);
    const float cn = live ? fg * creg + ig * gg : 0.f;
    const float h = live ? og * tanhf(cn) : 0.f;
    creg = cn;
    if (local) publish_local(xmine + (s & 1) * xslot, h, s + 1);   // first: 23 members are waiting for it
    else publish(xmine + (s & 1) * xslot, h, s + 1);
    bf16_t hi, lo;
    split_bf16(h, hi, lo);
    hnxt[sl * CLD + ul] = hi;                       // local k position of the own member: q = 0
    hnxt[(sl + 8) * CLD + ul] = lo;
    if (alive) {
      const int64_t bt = (int64_t)b * T + t;
      y[bt * (D * CH) + d * CH + unit] = h;
      extra[(bt * D + d) * CH + unit] = cn;
      if (live) {
        float* go = gates + (bt * D + d) * (CG * CH) + unit;
        go[0] = ig;
        go[CH] = fg;
        go[2 * CH] = gg;
        go[3 * CH] = og;
      }
    }
    lr_lds_barrier();   // hnxt's own k step complete; hcur free for the next gather
  };
  for (int s = 0; s < T; s += 2) {
    step(s, gxA);
    if (s + 1 < T) step(s + 1, gxB);
  }
  if (bad) atomicAdd(&g_cluster_err, 1);
}


// ---------------------------------------------------------------------------------------------
// backward recurrence (row-split, see the file header)
// ---------------------------------------------------------------------------------------------
// Per step a member contracts its OWN dG (4 gates x 32 units x 8 samples, bf16 hi/lo in rows 0-7 / 8-15 of the A
// operand, K = 128 = one k step per gate) against its 128 rows of W_hh for ALL 768 output units (48 column tiles:
// wave w owns units 192w .. 192w+191), folds the hi/lo rows, and publishes the partial dh of every unit to the
// member that owns it; each thread then gathers the 23 remote partials of its own (sample, unit) — fixed
// summation order — adds its own and runs the LSTM cell backward (rnn_bwd_step_kernel<4>'s arithmetic).
constexpr int BT12 = 12;            // column tiles per wave
constexpr int BFW = 8;              // fragments per tile: f = 2 * gate + plane
constexpr int BFW_A = 5;            // f < 5 in AGPRs (12 x 5 = 60 fragments)
constexpr int BFW_REG = 6;          // f == 5 in VGPRs (12 fragments); f = 6, 7 in LDS (24 per wave)
constexpr int BKLD = CG * CU_ + 8;  // bf16 per row of the dG operand buffer (128 own kappa)
constexpr size_t CBWD_LDS = (size_t)2 * 16 * BKLD * 2 + (size_t)4 * BT12 * (BFW - BFW_REG) * 1024 + (size_t)NS * CU_ * 4;

// out[((((d*CC + c)*4 + wave)*BT12 + tile)*BFW + f)*64 + lane] = plane f & 1 of W_hh[kappa + e][j], e = 0..7,
// kappa = (f >> 1) * 768 + 32 c + 8 kg, j = 192 wave + 16 tile + col
__global__ void lstm768_pack_whh_rows_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                             bf16x8* __restrict__ out, int D) {
  const int64_t total = (int64_t)D * CC * 4 * BT12 * BFW * 64;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), f = (int)((i >> 6) % BFW), tile = (int)((i / (64 * BFW)) % BT12);
    const int wave = (int)((i / (64 * BFW * BT12)) & 3), c = (int)((i / (64 * BFW * BT12 * 4)) % CC);
    const int d = (int)(i / ((int64_t)64 * BFW * BT12 * 4 * CC));
    const int col = lane & 15, kg = lane >> 4, gate = f >> 1, plane = f & 1;
    const int j = 192 * wave + 16 * tile + col;
    const float* src = (d ? w1 : w0) + ((int64_t)gate * CH + CU_ * c + 8 * kg) * CH + j;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bf16_t hi, lo;
      split_bf16(src[(int64_t)e * CH], hi, lo);
      v[e] = __builtin_bit_cast(__bf16, plane ? lo : hi);
    }
    out[i] = v;
  }
}

#define LR_CMFMA12(acc, a, WC, w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11)                                  \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %12, %13, %0\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %1, %12, %14, %1\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %2, %12, %15, %2\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %3, %12, %16, %3\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %4, %12, %17, %4\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %5, %12, %18, %5\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %6, %12, %19, %6\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %7, %12, %20, %7\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %8, %12, %21, %8\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %9, %12, %22, %9\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %10, %12, %23, %10\n\t"                                                 \
               "v_mfma_f32_16x16x32_bf16 %11, %12, %24, %11"                                                      \
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), \
                 "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11])                           \
               : "v"(a), WC(w0), WC(w1), WC(w2), WC(w3), WC(w4), WC(w5), WC(w6), WC(w7), WC(w8), WC(w9), WC(w10),  \
                 WC(w11))
#define LR_CMFMA12_FIRST(acc, a, W)                                                                               \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %12, %13, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %1, %12, %14, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %2, %12, %15, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %3, %12, %16, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %4, %12, %17, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %5, %12, %18, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %6, %12, %19, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %7, %12, %20, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %8, %12, %21, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %9, %12, %22, 0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %10, %12, %23, 0\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %11, %12, %24, 0"                                                        \
               : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]), "=&v"(acc[4]), "=&v"(acc[5]),         \
                 "=&v"(acc[6]), "=&v"(acc[7]), "=&v"(acc[8]), "=&v"(acc[9]), "=&v"(acc[10]), "=&v"(acc[11])       \
               : "v"(a), "a"(W[0][0]), "a"(W[1][0]), "a"(W[2][0]), "a"(W[3][0]), "a"(W[4][0]), "a"(W[5][0]),       \
                 "a"(W[6][0]), "a"(W[7][0]), "a"(W[8][0]), "a"(W[9][0]), "a"(W[10][0]), "a"(W[11][0]))

__global__ __launch_bounds__(256, 1) void lstm768_bwd_cluster_kernel(
    const float* __restrict__ gates, const float* __restrict__ extra, const float* __restrict__ dy,
    const float* __restrict__ dh_n, const float* __restrict__ dc_n, float* __restrict__ dG,
    const bf16x8* __restrict__ wpk, const int32_t* __restrict__ lens, u64* __restrict__ xch, int g0, int nclusters, int B,
    int T, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* gS = reinterpret_cast<bf16_t*>(smem);                                                    // [2][16][BKLD]
  bf16x8* Wl = reinterpret_cast<bf16x8*>(smem + (size_t)2 * 16 * BKLD * 2);                        // [4][BT12][2][64]
  float* own = reinterpret_cast<float*>(smem + (size_t)2 * 16 * BKLD * 2 + (size_t)4 * BT12 * (BFW - BFW_REG) * 1024);   // [NS][CU_]
  const int cluster = blockIdx.x & 7, c = blockIdx.x >> 3;
  if (cluster >= nclusters) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = cluster % D, group = g0 + cluster / D;
  const int col = lane & 15, kg = lane >> 4;
  const int DH = D * CH;

  bf16x8 Wa[BT12][BFW_A], Wv[BT12];
  const bf16x8* wsrc = wpk + ((int64_t)((d * CC + c) * 4 + wave) * BT12 * BFW) * 64 + lane;
#pragma unroll
  for (int tl = 0; tl < BT12; ++tl) {
#pragma unroll
    for (int f = 0; f < BFW; ++f) {
      const bf16x8 w = wsrc[(tl * BFW + f) * 64];
      if (f < BFW_A) Wa[tl][f] = w;
      else if (f < BFW_REG) Wv[tl] = w;
      else Wl[((wave * BT12 + tl) * (BFW - BFW_REG) + (f - BFW_REG)) * 64 + lane] = w;
    }
  }
  for (int i = tid; i < 2 * 16 * BKLD; i += 256) gS[i] = 0;

  // ---- one (sample, unit) per thread: sample = tid >> 5, unit = tid & 31 of this member ----------------------
  const int sl = tid >> 5, ul = tid & 31;
  const int unit = CU_ * c + ul;
  const int b = group * NS + sl;
  const bool alive = b < B;
  const int len = alive ? lens[b] : 0;
  const float inj_h = (alive && dh_n) ? dh_n[((int64_t)d * B + b) * CH + unit] : 0.f;
  const float inj_c = (alive && dc_n) ? dc_n[((int64_t)d * B + b) * CH + unit] : 0.f;
  float car = 0.f;   // dc_{t'} * f_{t'} of the step processed before
  struct In { float dy, g[4], c, cp; };
  In inA, inB;       // operands of even / odd steps, fetched TWO steps ahead
  auto time_of = [&](int s) {
    const int sc = s < T ? s : T - 1;
    return d == 0 ? T - 1 - sc : sc;
  };
  auto fetch = [&](In& in, int t) {
    in.dy = in.c = in.cp = 0.f;
    in.g[0] = in.g[1] = in.g[2] = in.g[3] = 0.f;
    if (!alive) return;
    const int tp = d == 0 ? t - 1 : t + 1;
    const int64_t bt = (int64_t)b * T + t;
    in.dy = dy[bt * DH + d * CH + unit];
    const float* gi = gates + (bt * D + d) * (int64_t)(CG * CH) + unit;
#pragma unroll
    for (int g = 0; g < CG; ++g) in.g[g] = gi[g * CH];
    in.c = extra[(bt * D + d) * CH + unit];
    if (tp >= 0 && tp < T) in.cp = extra[(((int64_t)b * T + tp) * D + d) * CH + unit];
  };
  fetch(inA, time_of(0));
  fetch(inB, time_of(1));
  // exchange: [slot][cluster][dst member][src member][sample][unit 32]
  const int64_t xsrc = NS * CU_, xdst = (int64_t)CC * xsrc, xcluster = (int64_t)CC * xdst, xslot = (int64_t)nclusters * xcluster;
  u64* xout = xch + cluster * xcluster + c * xsrc;                 // + dst * xdst + sample * 32 + unit
  const u64* xin = xch + cluster * xcluster + c * xdst + tid;      // + src * xsrc   (tid = sample * 32 + unit)
  int bad = 0;
  __shared__ int s_local;
  {
    u64* xid = xch + 2 * xslot + (int64_t)cluster * CC;
    if (tid == 0) {
      s_local = 1;
      publish(xid + c, __builtin_bit_cast(float, xcc_id()), 1);
    }
    __syncthreads();
    if (tid < CC) {
      u64 g = peek(xid + tid);
      int n = 0;
      while ((int)(g >> 32) != 1 && n++ < SPIN_LIMIT) {
        __builtin_amdgcn_s_sleep(2);
        g = peek(xid + tid);
      }
      if ((int)(g >> 32) != 1) bad = 1;
      if ((int)(g & 0xf) != xcc_id() || bad) s_local = 0;
    }
  }
  __syncthreads();
  const bool local = s_local != 0;

  auto step = [&](int s, In& in) {
    const int t = time_of(s);
    const int tnext = time_of(s + 2);
    bf16_t* gcur = gS + (s & 1) * 16 * BKLD;        // dG of the step processed before: rows 0-7 hi, 8-15 lo
    bf16_t* gnxt = gS + ((s + 1) & 1) * 16 * BKLD;
    float prod = 0.f;
    if (s > 0) {
      // ---- partial dh of all 768 units from this member's own dG ---------------------------------------
      f32x4 acc[BT12];
      bf16x8 a_next = *reinterpret_cast<const bf16x8*>(gcur + col * BKLD + kg * 8);
#pragma unroll
      for (int ks = 0; ks < CG; ++ks) {
        const bf16x8 a = a_next;
        if (ks + 1 < CG) a_next = *reinterpret_cast<const bf16x8*>(gcur + col * BKLD + (ks + 1) * 32 + kg * 8);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const int f = 2 * ks + pl;
          if (f == 0) {
            LR_CMFMA12_FIRST(acc, a, Wa);
          } else if (f < BFW_A) {
            LR_CMFMA12(acc, a, "a", Wa[0][f], Wa[1][f], Wa[2][f], Wa[3][f], Wa[4][f], Wa[5][f], Wa[6][f], Wa[7][f], Wa[8][f],
                       Wa[9][f], Wa[10][f], Wa[11][f]);
          } else if (f < BFW_REG) {
            LR_CMFMA12(acc, a, "v", Wv[0], Wv[1], Wv[2], Wv[3], Wv[4], Wv[5], Wv[6], Wv[7], Wv[8], Wv[9], Wv[10], Wv[11]);
          } else {   // LDS-resident fragments, six tiles at a time (twelve at once cost 48 registers)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              bf16x8 wl[6];
#pragma unroll
              for (int i = 0; i < 6; ++i)
                wl[i] = Wl[((wave * BT12 + 6 * half + i) * (BFW - BFW_REG) + (f - BFW_REG)) * 64 + lane];
              asm volatile("v_mfma_f32_16x16x32_bf16 %0, %6, %7, %0\n\t"
                           "v_mfma_f32_16x16x32_bf16 %1, %6, %8, %1\n\t"
                           "v_mfma_f32_16x16x32_bf16 %2, %6, %9, %2\n\t"
                           "v_mfma_f32_16x16x32_bf16 %3, %6, %10, %3\n\t"
                           "v_mfma_f32_16x16x32_bf16 %4, %6, %11, %4\n\t"
                           "v_mfma_f32_16x16x32_bf16 %5, %6, %12, %5"
                           : "+v"(acc[6 * half]), "+v"(acc[6 * half + 1]), "+v"(acc[6 * half + 2]), "+v"(acc[6 * half + 3]),
                             "+v"(acc[6 * half + 4]), "+v"(acc[6 * half + 5])
                           : "v"(a), "v"(wl[0]), "v"(wl[1]), "v"(wl[2]), "v"(wl[3]), "v"(wl[4]), "v"(wl[5]));
            }
          }
        }
      }
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
      // rows 4 kg + r: rows 0-7 (kg 0, 1) came from the hi plane of dG, rows 8-15 (kg 2, 3) from the lo plane of the
      // same samples: fold across lanes +-32.  Afterwards both halves hold sample 4 (kg & 1) + r of unit 192 wave +
      // 16 tile + col; the lower half publishes r = 0, 1, the upper half r = 2, 3.
      // (the slot offset goes through an opaque register: otherwise the compiler hoists the 24 + 23 per-lane
      // addresses of BOTH parity slots out of the step loop and keeps ~190 registers of addresses alive)
      int64_t slot_off = ((s - 1) & 1) * xslot;
      asm volatile("" : "+s"(slot_off));
      u64* xo = xout + slot_off;
#pragma unroll
      for (int tl = 0; tl < BT12; ++tl) {
        const int j = 192 * wave + 16 * tl + col, dstm = j >> 5, ju = j & 31;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tl][r] += __shfl_xor(acc[tl][r], 32, 64);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int r = 2 * (kg >> 1) + rr, smp = 4 * (kg & 1) + r;
          const float v = (kg >> 1) ? (rr ? acc[tl][3] : acc[tl][2]) : (rr ? acc[tl][1] : acc[tl][0]);
          if (dstm == c) {
            own[smp * CU_ + ju] = v;
          } else {
            u64* p = xo + dstm * xdst + smp * CU_ + ju;
            if (local) publish_local(p, v, s);
            else publish(p, v, s);
          }
        }
      }
      // ---- gather the 23 remote partials of this thread's (sample, unit): tag s, slot (s-1) & 1 ---------
      const u64* xp = xin + slot_off;
      u64 g[CC - 1];
#pragma unroll
      for (int q = 1; q < CC; ++q) {
        int jm = c + q;
        if (jm >= CC) jm -= CC;
        g[q - 1] = peek(xp + jm * xsrc);
      }
      lr_lds_barrier();     // `own` complete
      unsigned pend = (1u << (CC - 1)) - 1;
      for (int round = 0; pend && !bad; ++round) {
#pragma unroll
        for (int q = 1; q < CC; ++q)
          if (((pend >> (q - 1)) & 1u) && (int)(g[q - 1] >> 32) == s) pend &= ~(1u << (q - 1));
        if (!pend) break;
        if (round > SPIN_LIMIT) {
          bad = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
#pragma unroll
        for (int q = 1; q < CC; ++q) {
          if ((pend >> (q - 1)) & 1u) {
            int jm = c + q;
            if (jm >= CC) jm -= CC;
            g[q - 1] = peek(xp + jm * xsrc);
          }
        }
      }
      prod = own[sl * CU_ + ul];
      if (!bad) {   // every g[] now holds its granule: sum in FIXED order (rotated by the member index)
#pragma unroll
        for (int q = 1; q < CC; ++q) prod += __builtin_bit_cast(float, (unsigned)(g[q - 1] & 0xffffffffu));
      }
    }
    // ---- LSTM cell backward of step t (rnn_bwd_step_kernel<4>) --------------------------------------------------
    const float ig = in.g[0], fg = in.g[1], gg = in.g[2], og = in.g[3], ct = in.c, cp = in.cp;
    float dh = in.dy + prod;
    const bool is_last = d == 0 ? (t == len - 1) : (t == 0);
    if (is_last) dh += inj_h;
    float dc = car;
    if (is_last) dc += inj_c;
    fetch(in, tnext);
    float di = 0.f, df = 0.f, dg_ = 0.f, do_ = 0.f;
    car = 0.f;
    if (alive && t < len) {
      const float tc = tanhf(ct);
      dc += dh * og * (1.f - tc * tc);
      di = dc * gg * ig * (1.f - ig);
      df = dc * cp * fg * (1.f - fg);
      dg_ = dc * ig * (1.f - gg * gg);
      do_ = dh * tc * og * (1.f - og);
      car = dc * fg;
    }
    bf16_t hi, lo;
    split_bf16(di, hi, lo);
    gnxt[sl * BKLD + ul] = hi;
    gnxt[(sl + 8) * BKLD + ul] = lo;
    split_bf16(df, hi, lo);
    gnxt[sl * BKLD + CU_ + ul] = hi;
    gnxt[(sl + 8) * BKLD + CU_ + ul] = lo;
    split_bf16(dg_, hi, lo);
    gnxt[sl * BKLD + 2 * CU_ + ul] = hi;
    gnxt[(sl + 8) * BKLD + 2 * CU_ + ul] = lo;
    split_bf16(do_, hi, lo);
    gnxt[sl * BKLD + 3 * CU_ + ul] = hi;
    gnxt[(sl + 8) * BKLD + 3 * CU_ + ul] = lo;
    if (alive) {
      float* dgo = dG + (((int64_t)b * T + t) * D + d) * (int64_t)(4 * CH) + unit;
      dgo[0] = di;
      dgo[CH] = df;
      dgo[2 * CH] = dg_;
      dgo[3 * CH] = do_;
    }
    lr_lds_barrier();   // gnxt complete; `own` free again
  };
  for (int s = 0; s < T; s += 2) {
    step(s, inA);
    if (s + 1 < T) step(s + 1, inB);
  }
  if (bad) atomicAdd(&g_cluster_err, 1);
}

}  // namespace

// read-and-clear of the error word (lr_rnn_pair_errors adds it to the pair kernels' count)
int lr_cluster_errors() {
  int v = 0, zero = 0;
  lr_clear_error();
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_cluster_err), sizeof(int)) != hipSuccess) return -1;
  if (v && hipMemcpyToSymbol(HIP_SYMBOL(g_cluster_err), &zero, sizeof(int)) != hipSuccess) return -1;
  return v;
}

int lr_lstm768_cluster_supported(int G, int B, int H) { return G == 4 && H == CH && B >= 1 ? 1 : 0; }
size_t lr_lstm768_cluster_pack_bytes(int D) { return (size_t)D * CC * 4 * 2 * CF * 64 * sizeof(bf16x8); }
size_t lr_lstm768_cluster_xch_bytes(int B, int D, int backward) {
  int clusters = (B + NS - 1) / NS * D;
  if (clusters > MAX_CLUSTERS) clusters = MAX_CLUSTERS;
  const size_t per = backward ? (size_t)CC * CC * NS * CU_ : (size_t)CC * NS * CU_;
  return ((size_t)2 * clusters * per + (size_t)clusters * CC) * sizeof(u64);
}
size_t lr_lstm768_cluster_bwd_pack_bytes(int D) { return (size_t)D * CC * 4 * BT12 * BFW * 64 * sizeof(bf16x8); }

int lr_lstm768_cluster_forward(float* gates, float* extra, float* y, const float* const* w_hh, const int32_t* lens,
                               void* wpack, void* xch, int B, int T, int D, hipStream_t stream) {
  static bool attr_set = false;
  lr_clear_error();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)lstm768_fwd_cluster_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)CFWD_LDS) != hipSuccess)
      return LR_ERR_LAUNCH;
    attr_set = true;
  }
  LR_LAUNCH(lstm768_pack_whh_kernel, dim3(1024), dim3(256), 0, stream, w_hh[0], w_hh[D - 1], (bf16x8*)wpack, D);
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  const int groups = (B + NS - 1) / NS, gchunk = MAX_CLUSTERS / D;   // sample groups per launch
  for (int g0 = 0; g0 < groups; g0 += gchunk) {
    const int ng = groups - g0 < gchunk ? groups - g0 : gchunk, nclusters = ng * D;
    if (hipMemsetAsync(xch, 0, ((size_t)2 * nclusters * CC * NS * CU_ + (size_t)nclusters * CC) * sizeof(u64), stream) != hipSuccess)
      return LR_ERR_LAUNCH;
    const dim3 grid(8 * CC);
    hipEvent_t e0, e1;
    if (g0 == 0 && lr_prof_next(LR_PROF_RNN_FWD, &e0, &e1))
      hipExtLaunchKernelGGL(lstm768_fwd_cluster_kernel, grid, dim3(256), CFWD_LDS, stream, e0, e1, 0, gates, extra, y,
                            (const bf16x8*)wpack, lens, (u64*)xch, g0, nclusters, B, T, D);
    else
      hipLaunchKernelGGL(lstm768_fwd_cluster_kernel, grid, dim3(256), CFWD_LDS, stream, gates, extra, y,
                         (const bf16x8*)wpack, lens, (u64*)xch, g0, nclusters, B, T, D);
    st = lr_launch_status();
    if (st != LR_OK) return st;
  }
  return LR_OK;
}

int lr_lstm768_cluster_backward(const float* gates, const float* extra, const float* dy, const float* dh_n,
                                const float* dc_n, float* dG, const float* const* w_hh, const int32_t* lens, void* wpack,
                                void* xch, int B, int T, int D, hipStream_t stream) {
  static bool attr_set = false;
  lr_clear_error();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)lstm768_bwd_cluster_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)CBWD_LDS) != hipSuccess)
      return LR_ERR_LAUNCH;
    attr_set = true;
  }
  LR_LAUNCH(lstm768_pack_whh_rows_kernel, dim3(1024), dim3(256), 0, stream, w_hh[0], w_hh[D - 1], (bf16x8*)wpack, D);
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  const int groups = (B + NS - 1) / NS, gchunk = MAX_CLUSTERS / D;
  for (int g0 = 0; g0 < groups; g0 += gchunk) {
    const int ng = groups - g0 < gchunk ? groups - g0 : gchunk, nclusters = ng * D;
    if (hipMemsetAsync(xch, 0, ((size_t)2 * nclusters * CC * CC * NS * CU_ + (size_t)nclusters * CC) * sizeof(u64),
                       stream) != hipSuccess)
      return LR_ERR_LAUNCH;
    const dim3 grid(8 * CC);
    hipEvent_t e0, e1;
    if (g0 == 0 && lr_prof_next(LR_PROF_RNN_BWD, &e0, &e1))
      hipExtLaunchKernelGGL(lstm768_bwd_cluster_kernel, grid, dim3(256), CBWD_LDS, stream, e0, e1, 0, gates, extra, dy, dh_n,
                            dc_n, dG, (const bf16x8*)wpack, lens, (u64*)xch, g0, nclusters, B, T, D);
    else
      hipLaunchKernelGGL(lstm768_bwd_cluster_kernel, grid, dim3(256), CBWD_LDS, stream, gates, extra, dy, dh_n, dc_n, dG,
                         (const bf16x8*)wpack, lens, (u64*)xch, g0, nclusters, B, T, D);
    st = lr_launch_status();
    if (st != LR_OK) return st;
  }
  return LR_OK;
}
