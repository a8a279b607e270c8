// lr_rnn_pair.hip — the GRU-256 recurrence of the REFERENCE-FAITHFUL regime as ONE launch per layer
// pass, fp32-faithful: W_hh and the carried state are split into bf16 hi + lo planes and the product
// runs on the bf16 matrix cores with fp32 accumulation — (h_hi + h_lo)(W_hi + W_lo), all four cross
// terms — which reproduces the fp32 product to ~1e-6 (the step kernels' exact-fp32 MFMA: ~1e-7; the
// single-plane bf16 kernel of lr_rnn_persist.hip: ~1e-3).  Replaces nothing new in the reference: it is
// better_model.py:74 (`self.rnn(packed)`, nn.GRU fp32) like lr_rnn.hip's step kernels, whose interface
// buffers (gates in/out, extra, y, dG) it shares.
//
// Why a PAIR of compute units.  The two planes of one direction's W_hh are 768 KB — more than the
// registers + LDS of a CU (512 KB + 160 KB) — so each (sample, direction) gets TWO workgroups on two
// CUs: member m owns hidden units [128m, 128m + 128) (all three gates: 384 gate columns x 256 k x 2
// planes = 384 KB = 96 MFMA B fragments per wave, 60 in AGPRs read by the matrix core directly, 24-36
// in VGPRs, the rest in LDS — the budget of the single-plane kernel).  The hi and lo planes of the
// STATE ride in rows 0 and 1 of the 16-row MFMA A operand, so a (k step, weight plane) costs one MFMA
// per column tile and rows 0 + 1 of the accumulator sum to the full product: 96 MFMAs per wave per
// step, exactly the single-plane kernel's count.
//
// The price is one exchange per step: a member needs the other member's 128 new state values (forward;
// 384 gate gradients backward).  They travel as 8-byte {value, tag = step + 1} granules written with
// ONE store each and polled with agent-scope loads (MI355X_MICROARCH.md "handoff-1to1": ~1 us idle,
// data-tagged granules need no fence); both members sit on the same XCD (blocks i and i + 8), verify it
// (HW_REG_XCC_ID) and then store at workgroup scope, so the granule stays in that XCD's L2 (an agent-scope
// store writes through and drops the line: forward 1.93 -> 1.74 us per step, backward 2.27 -> 2.11).
// The k loop runs the member's OWN half first (its operands are local), so about a third of the
// exchange latency hides behind MFMAs.  Two parity slots suffice: a producer overwrites slot s & 1 at
// step s + 2 only after it consumed the partner's step-(s + 1) data, which the partner published
// after reading slot s & 1.  Spins are bounded: a partner that never shows (it was not resident) makes
// the kernel finish with garbage and raise a device-side error word instead of hanging the GPU.
#include "lr_common.h"
#include <hip/hip_ext.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;
typedef unsigned long long u64;

constexpr int PH = 256;            // hidden size
constexpr int HALF = 128;          // hidden units per pair member
constexpr int PHLD = PH + 8;       // bf16 per LDS row of the state
constexpr int SPIN_LIMIT = 1 << 18;   // polls before a member gives up on its partner (~0.1 s), once per launch

// ---- forward geometry --------------------------------------------------------------------------------
constexpr int FNT = 6;             // column tiles per wave: 3 gates x 2 (wave w owns units 32w .. 32w+31 of the member)
constexpr int FKS = PH / 32;       // 8 k steps of 32; LOCAL k order: the member's own 128 units first
constexpr int FF = 2 * FKS;        // fragments per tile: f = 2 * kstep + plane (0 = W_hi, 1 = W_lo)
constexpr int FF_A = 10;           // f < 10 in AGPRs (6 x 10 = 60 fragments = 240 registers)
constexpr int FF_REG = 16;         // f < 16 in registers (6 x 6 = 36 fragments in VGPRs): all of them, none in LDS
constexpr size_t FWD_LDS = (size_t)2 * 16 * PHLD * 2 + (size_t)4 * FNT * (FF - FF_REG) * 1024 + (size_t)3 * HALF * 4;

__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
  return __builtin_bit_cast(float, (unsigned)b << 16);
}
// fp32 -> bf16 hi + bf16 lo (x ~= hi + lo to 2^-17 relative)
__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}
__device__ __forceinline__ u64 granule(float v, int tag) {
  return ((u64)(unsigned)tag << 32) | (u64)__builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void publish(u64* p, float v, int tag) {
  __hip_atomic_store(p, granule(v, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 peek(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same granule with a store that stops at the writer's XCD L2 and stays there (the agent-scope `sc1` store
// writes through and drops the line, so the partner's poll goes to the fabric).  Only the other CUs of the SAME
// XCD are guaranteed to see it there: used when the two members verified at run time (HW_REG_XCC_ID) that they
// share an XCD — the usual placement of blocks i and i + 8.  (As lr_rnn_cluster.hip.)
__device__ __forceinline__ void publish_local(u64* p, float v, int tag) {
  __hip_atomic_store(p, granule(v, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int xcc_id() {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 0xf;
}
// one-time handshake of a pair: each member publishes its XCC id (tag 1) at xid[member] and reads the partner's;
// true when both run on the same XCD.  A partner that never answers leaves `false` (the per-step waits raise the
// error word).
__device__ __forceinline__ bool pair_shares_xcd(u64* xid, int m, int tid) {
  __shared__ int s_local;
  if (tid == 0) {
    publish(xid + m, __builtin_bit_cast(float, xcc_id()), 1);
    u64 g = peek(xid + (1 - m));
    int n = 0;
    while ((int)(g >> 32) != 1 && n++ < SPIN_LIMIT) {
      __builtin_amdgcn_s_sleep(2);
      g = peek(xid + (1 - m));
    }
    s_local = ((int)(g >> 32) == 1 && (int)(g & 0xf) == xcc_id()) ? 1 : 0;
  }
  __syncthreads();
  return s_local != 0;
}
// wait until the granule carries `tag`; returns its value (0 and *err set after SPIN_LIMIT polls)
__device__ __forceinline__ float await(const u64* p, u64 first, int tag, int* err) {
  u64 g = first;
  int n = 0;
  if (*err) return 0.f;      // gave up before: do not wait again (the results are garbage already)
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

// block -> (pair, member): blocks i and i + 8 of a group of 16 form a pair (the dispatcher places block
// b on XCD b % 8, so the two members share an L2 — a speed matter only, never correctness)
__device__ __forceinline__ void pair_of(int block, int& pair, int& member) {
  const int g = block >> 4, r = block & 15;
  member = r >> 3;
  pair = g * 8 + (r & 7);
}

// W_hh [3*256][256] fp32 of each direction -> bf16 hi/lo MFMA B fragments in the order the forward
// kernel consumes them: out[((((d*2 + m)*4 + wave)*FNT + tl)*FF + f)*64 + lane] (8 bf16 = plane f & 1 of
// W_hh[gate*256 + unit][k .. k+7], gate = tl >> 1, unit = 128m + 32 wave + 16 (tl & 1) + col, k = the
// member-local k step f >> 1 (own half first) + 8 kg; lane = kg*16 + col).
// (also clears the first launch's exchange buffer, `nzero` granules: one launch less than a memset node beside it)
__global__ void gru256_pair_pack_whh_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                            bf16x8* __restrict__ out, int D, u64* __restrict__ xch, int nzero) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nzero; i += gridDim.x * blockDim.x) xch[i] = 0;
  const int total = D * 2 * 4 * FNT * FF * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, f = (i >> 6) % FF, tl = (i / (64 * FF)) % FNT, wave = (i / (64 * FF * FNT)) & 3;
    const int m = (i / (64 * FF * FNT * 4)) & 1, d = i / (64 * FF * FNT * 8);
    const int col = lane & 15, kg = lane >> 4, g = tl >> 1, nt = tl & 1, ks = f >> 1, plane = f & 1;
    const int unit = HALF * m + 32 * wave + 16 * nt + col;
    const int kbase = (ks < 4 ? HALF * m + 32 * ks : HALF * (1 - m) + 32 * (ks - 4)) + 8 * kg;
    const float* row = (d ? w1 : w0) + ((int64_t)g * PH + unit) * PH + kbase;
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

// six v_mfma_f32_16x16x32_bf16 (one per column tile) sharing the A fragment; weights in register class WC
#define LR_MFMA6(acc, a, WC, w0, w1, w2, w3, w4, w5)                                                          \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %6, %7, %0\n\t"                                                  \
               "v_mfma_f32_16x16x32_bf16 %1, %6, %8, %1\n\t"                                                  \
               "v_mfma_f32_16x16x32_bf16 %2, %6, %9, %2\n\t"                                                  \
               "v_mfma_f32_16x16x32_bf16 %3, %6, %10, %3\n\t"                                                 \
               "v_mfma_f32_16x16x32_bf16 %4, %6, %11, %4\n\t"                                                 \
               "v_mfma_f32_16x16x32_bf16 %5, %6, %12, %5"                                                     \
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5])          \
               : "v"(a), WC(w0), WC(w1), WC(w2), WC(w3), WC(w4), WC(w5))
#define LR_MFMA6_FIRST(acc, a, w0, w1, w2, w3, w4, w5)                                                        \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %6, %7, 0\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %1, %6, %8, 0\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %2, %6, %9, 0\n\t"                                                   \
               "v_mfma_f32_16x16x32_bf16 %3, %6, %10, 0\n\t"                                                  \
               "v_mfma_f32_16x16x32_bf16 %4, %6, %11, 0\n\t"                                                  \
               "v_mfma_f32_16x16x32_bf16 %5, %6, %12, 0"                                                      \
               : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]), "=&v"(acc[4]), "=&v"(acc[5])    \
               : "v"(a), "a"(w0), "a"(w1), "a"(w2), "a"(w3), "a"(w4), "a"(w5))

// grid: 16 * ceil(pairs / 8) workgroups of 256 threads; pair p = (sample b0 + p / D, direction p % D).
// Product phase: wave w owns the member's units 32w .. 32w+31 as column tiles tl = gate*2 + nt.  Gate
// phase: lanes 0..31 of wave w run units 32w + lane (the wave's own results: wave-local LDS exchange);
// lanes 32..63 fetch the partner's state of the same unit index.
__global__ __launch_bounds__(256, 1) void gru256_fwd_pair_kernel(float* __restrict__ gates, float* __restrict__ extra,
                                                                 float* __restrict__ y, const bf16x8* __restrict__ wpk,
                                                                 const float* __restrict__ bhh0,
                                                                 const float* __restrict__ bhh1,
                                                                 const int32_t* __restrict__ lens, u64* __restrict__ xch,
                                                                 int32_t* __restrict__ fault, int drop,
                                                                 int b0, int npairs, int B, int T, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* hS = reinterpret_cast<bf16_t*>(smem);                                                  // [2][16][PHLD]
  bf16x8* Wl = reinterpret_cast<bf16x8*>(smem + (size_t)2 * 16 * PHLD * 2);                      // [4][FNT][FF-FF_REG][64]
  float* S = reinterpret_cast<float*>(smem + (size_t)2 * 16 * PHLD * 2 + (size_t)4 * FNT * (FF - FF_REG) * 1024);   // [3][HALF]
  int pair, m;
  pair_of(blockIdx.x, pair, m);
  if (pair >= npairs) return;      // both members of a pair beyond the range leave together
  if (m == drop) return;           // test hook (lr_rnn_debug_drop_member): the partner must time out and report
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = b0 + pair / D, d = pair % D;
  const float* bhh = d ? bhh1 : bhh0;
  const int col = lane & 15, kg = lane >> 4;

  // ---- weights: 96 fragments per wave ----------------------------------------------------------------
  bf16x8 Wa[FNT][FF_A], Wv[FNT][FF_REG - FF_A];
  const bf16x8* wsrc = wpk + ((int64_t)((d * 2 + m) * 4 + wave) * FNT * FF) * 64 + lane;
#pragma unroll
  for (int tl = 0; tl < FNT; ++tl) {
#pragma unroll
    for (int f = 0; f < FF; ++f) {
      const bf16x8 w = wsrc[(tl * FF + f) * 64];
      if (f < FF_A) Wa[tl][f] = w;
      else if (f < FF_REG) Wv[tl][f - FF_A] = w;
      else Wl[((wave * FNT + tl) * (FF - FF_REG) + (f - FF_REG)) * 64 + lane] = w;
    }
  }
  for (int i = tid; i < 2 * 16 * PHLD; i += 256) hS[i] = 0;

  // ---- per-thread roles ----------------------------------------------------------------------------------
  const bool gate_lane = lane < 32;           // lanes 0..31: gate math of unit ul; lanes 32..63: the partner's unit ul
  const int ul = 32 * wave + (lane & 31);     // member-local unit index
  const int unit = HALF * m + ul;             // hidden unit
  const float bhn = bhh[2 * PH + unit];
  const int len = lens[b];
  float hreg = 0.f;
  struct Gx { float v[3]; };
  Gx gxA, gxB;   // pre-activations of even / odd steps, fetched TWO steps ahead
  auto time_of = [&](int s) {
    const int sc = s < T ? s : T - 1;
    return d == 0 ? sc : T - 1 - sc;
  };
  auto fetch_gx = [&](Gx& gx, int t) {
    const float* gp = gates + (((int64_t)b * T + t) * D + d) * (3 * PH) + unit;
    gx.v[0] = gp[0];
    gx.v[1] = gp[PH];
    gx.v[2] = gp[2 * PH];
  };
  if (gate_lane) {
    fetch_gx(gxA, time_of(0));
    fetch_gx(gxB, time_of(1));
  }
  // exchange buffers: [slot][pair][member][HALF]
  u64* xmine = xch + ((int64_t)pair * 2 + m) * HALF + ul;
  const u64* xtheirs = xch + ((int64_t)pair * 2 + (1 - m)) * HALF + ul;
  const int64_t xslot = (int64_t)npairs * 2 * HALF;
  int bad = 0;
  const bool local = pair_shares_xcd(xch + 2 * xslot + (int64_t)pair * 2, m, tid);   // (a barrier inside)

  auto step = [&](int s, Gx& gx) {
    const int t = time_of(s);
    const int tnext = time_of(s + 2);
    bf16_t* hcur = hS + (s & 1) * 16 * PHLD;
    bf16_t* hnxt = hS + ((s + 1) & 1) * 16 * PHLD;
    f32x4 acc[FNT];
    if (s > 0) {
      // the partner's h_{s-1} (tag s, slot (s-1) & 1): ask for it now, look at it after the own-half MFMAs
      const u64* xp = xtheirs + ((s - 1) & 1) * xslot;
      u64 first = 0;
      if (!gate_lane) first = peek(xp);
      // ---- own half: local k steps 0..3 = fragments f 0..7 (all in AGPRs) --------------------------
      bf16x8 a_next = *reinterpret_cast<const bf16x8*>(hcur + col * PHLD + kg * 8);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = a_next;
        if (ks + 1 < 4) a_next = *reinterpret_cast<const bf16x8*>(hcur + col * PHLD + (ks + 1) * 32 + kg * 8);
        if (ks == 0) LR_MFMA6_FIRST(acc, a, Wa[0][0], Wa[1][0], Wa[2][0], Wa[3][0], Wa[4][0], Wa[5][0]);
        else LR_MFMA6(acc, a, "a", Wa[0][2 * ks], Wa[1][2 * ks], Wa[2][2 * ks], Wa[3][2 * ks], Wa[4][2 * ks], Wa[5][2 * ks]);
        LR_MFMA6(acc, a, "a", Wa[0][2 * ks + 1], Wa[1][2 * ks + 1], Wa[2][2 * ks + 1], Wa[3][2 * ks + 1],
                 Wa[4][2 * ks + 1], Wa[5][2 * ks + 1]);
      }
      // ---- the partner's half of the state -> LDS rows 0 (hi) / 1 (lo), local positions 128 + ul ----
      if (!gate_lane) {
        const float hv = await(xp, first, s, &bad);
        bf16_t hi, lo;
        split_bf16(hv, hi, lo);
        hcur[HALF + ul] = hi;
        hcur[PHLD + HALF + ul] = lo;
      }
      lr_lds_barrier();
      // ---- partner half: local k steps 4..7 = fragments f 8..15 ---------------------------------------------
      a_next = *reinterpret_cast<const bf16x8*>(hcur + col * PHLD + 4 * 32 + kg * 8);
#pragma unroll
      for (int ks = 4; ks < FKS; ++ks) {
        const bf16x8 a = a_next;
        if (ks + 1 < FKS) a_next = *reinterpret_cast<const bf16x8*>(hcur + col * PHLD + (ks + 1) * 32 + kg * 8);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const int f = 2 * ks + pl;
          if (f < FF_A) {
            LR_MFMA6(acc, a, "a", Wa[0][f], Wa[1][f], Wa[2][f], Wa[3][f], Wa[4][f], Wa[5][f]);
          } else if (f < FF_REG) {
            const int j = f - FF_A;
            LR_MFMA6(acc, a, "v", Wv[0][j], Wv[1][j], Wv[2][j], Wv[3][j], Wv[4][j], Wv[5][j]);
          } else {
            bf16x8 wl[FNT];
#pragma unroll
            for (int tl = 0; tl < FNT; ++tl) wl[tl] = Wl[((wave * FNT + tl) * (FF - FF_REG) + (f - FF_REG)) * 64 + lane];
            LR_MFMA6(acc, a, "v", wl[0], wl[1], wl[2], wl[3], wl[4], wl[5]);
          }
        }
      }
      // the asm MFMAs are opaque to the compiler's hazard recogniser: let the last ones retire before any
      // VALU instruction reads an accumulator
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
      // (... and tie the accumulators to a statement behind the wait: their reads are plain register arithmetic,
      // which nothing else keeps behind it — see LR_ACC_READY in lr_rnn_cluster.hip)
#pragma unroll
      for (int tl = 0; tl < FNT; ++tl) asm volatile("" : "+v"(acc[tl]));
      // rows 0 (state hi) + 1 (state lo) of the tile = the full product; they sit in lanes kg == 0, regs 0, 1
      if (kg == 0) {
#pragma unroll
        for (int tl = 0; tl < FNT; ++tl) S[(tl >> 1) * HALF + 32 * wave + 16 * (tl & 1) + col] = acc[tl][0] + acc[tl][1];
      }
    } else {
      if (kg == 0) {
#pragma unroll
        for (int tl = 0; tl < FNT; ++tl) S[(tl >> 1) * HALF + 32 * wave + 16 * (tl & 1) + col] = 0.f;
      }
    }
    // wave w produced units 32w .. 32w+31 and its own lanes consume them: in-order LDS, no barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (gate_lane) {
      const bool live = t < len;
      const float hn = S[2 * HALF + ul] + bhn;
      const float r = lr_sigmoid(gx.v[0] + S[ul]);
      const float z = lr_sigmoid(gx.v[1] + S[HALF + ul]);
      const float n = tanhf(gx.v[2] + r * hn);
      fetch_gx(gx, tnext);
      const float h = live ? (1.f - z) * n + z * hreg : 0.f;
      hreg = h;
      if (local) publish_local(xmine + (s & 1) * xslot, h, s + 1);      // first: the partner is waiting for it
      else publish(xmine + (s & 1) * xslot, h, s + 1);
      bf16_t hi, lo;
      split_bf16(h, hi, lo);
      hnxt[ul] = hi;
      hnxt[PHLD + ul] = lo;
      const int64_t bt = (int64_t)b * T + t;
      y[bt * (D * PH) + d * PH + unit] = h;
      extra[(bt * D + d) * PH + unit] = live ? hn : 0.f;
      if (live) {
        float* go = gates + (bt * D + d) * (3 * PH) + unit;
        go[0] = r;
        go[PH] = z;
        go[2 * PH] = n;
      }
    }
    lr_lds_barrier();   // hnxt's own half complete, S free again
  };
  for (int s = 0; s < T; s += 2) {
    step(s, gxA);
    if (s + 1 < T) step(s + 1, gxB);
  }
  if (bad && fault) atomicOr(fault, 1);   // device-side: this step is skipped (lr_common.h lr_fault_words)
}

// ---------------------------------------------------------------------------------------------
// backward recurrence
// ---------------------------------------------------------------------------------------------
// dh_t[j] = dy_t[j] + dh_{t'} z_{t'} + sum_kappa dGh_{t'}[kappa] W_hh[kappa][j]  (t' = the step processed
// just before; kappa = (gate, k) over 3 x 256), then the gate gradients of step t — rnn_bwd_step_kernel<3>'s
// arithmetic.  Member m owns OUTPUT units j in [128m, 128m + 128) (wave w: 32w .. 32w+31 = 2 column tiles)
// over all 768 kappa = 24 k steps x 2 planes = 96 fragments per wave; member-local kappa order: the 384
// kappa whose unit k is the member's own first (its own gate gradients), then the partner's 384.
constexpr int BNT = 2;
constexpr int BKS = 3 * PH / 32;    // 24
constexpr int BF = 2 * BKS;         // 48 fragments per tile
constexpr int BF_A = 30;            // f < 30 in AGPRs (2 x 30 = 60 fragments); the other 36 in VGPRs
constexpr int BOWN = 3 * HALF;      // 384 own kappa
constexpr int BGLD = 2 * BOWN + 8;  // bf16 per row of the dGh buffer
constexpr size_t BWD_LDS = (size_t)2 * 2 * BGLD * 2 + 16 + (size_t)HALF * 4;

// out[((((d*2 + m)*4 + wave)*BNT + nt)*BF + f)*64 + lane]: plane f & 1 of W_hh[kappa + e][j], j = 128m + 32 wave +
// 16 nt + col, kappa = local k step f >> 1 -> (gate = (ks % 12) / 4, unit k = 128 (own ? m : 1-m) + 32 (ks % 4) + 8 kg)
__global__ void gru256_pair_pack_whh_t_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                              bf16x8* __restrict__ out, int D, u64* __restrict__ xch, int nzero) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nzero; i += gridDim.x * blockDim.x) xch[i] = 0;
  const int total = D * 2 * 4 * BNT * BF * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, f = (i >> 6) % BF, nt = (i / (64 * BF)) & 1, wave = (i / (64 * BF * 2)) & 3;
    const int m = (i / (64 * BF * 8)) & 1, d = i / (64 * BF * 16);
    const int col = lane & 15, kg = lane >> 4, ks = f >> 1, plane = f & 1;
    const int own = ks < 12, kl = ks % 12, g = kl >> 2;
    const int k = HALF * (own ? m : 1 - m) + 32 * (kl & 3) + 8 * kg;
    const float* src = (d ? w1 : w0) + ((int64_t)g * PH + k) * PH + HALF * m + 32 * wave + 16 * nt + col;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bf16_t hi, lo;
      split_bf16(src[(int64_t)e * PH], hi, lo);
      v[e] = __builtin_bit_cast(__bf16, plane ? lo : hi);
    }
    out[i] = v;
  }
}

#define LR_MFMA2_FIRST(acc, a, w0, w1)                                                                          \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, 0\n\t"                                                     \
               "v_mfma_f32_16x16x32_bf16 %1, %2, %4, 0"                                                         \
               : "=&v"(acc[0]), "=&v"(acc[1])                                                                   \
               : "v"(a), "a"(w0), "a"(w1))
#define LR_MFMA2(acc, a, WC, w0, w1)                                                                            \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\t"                                                    \
               "v_mfma_f32_16x16x32_bf16 %1, %2, %4, %1"                                                        \
               : "+v"(acc[0]), "+v"(acc[1])                                                                     \
               : "v"(a), WC(w0), WC(w1))

__global__ __launch_bounds__(256, 1) void gru256_bwd_pair_kernel(
    const float* __restrict__ gates, const float* __restrict__ extra, const float* __restrict__ y,
    const float* __restrict__ dy, const float* __restrict__ dh_n, float* __restrict__ dG,
    const bf16x8* __restrict__ wpk, const int32_t* __restrict__ lens, u64* __restrict__ xch, int32_t* __restrict__ fault,
    int drop, int b0, int npairs, int B, int T, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* gS = reinterpret_cast<bf16_t*>(smem);                                        // [2 parity][2 rows][BGLD]
  float* S = reinterpret_cast<float*>(smem + (size_t)2 * 2 * BGLD * 2 + 16);           // [HALF]
  int pair, m;
  pair_of(blockIdx.x, pair, m);
  if (pair >= npairs) return;
  if (m == drop) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = b0 + pair / D, d = pair % D;
  const int col = lane & 15, kg = lane >> 4;
  const int DH = D * PH;

  bf16x8 Wa[BNT][BF_A], Wv[BNT][BF - BF_A];
  const bf16x8* wsrc = wpk + ((int64_t)((d * 2 + m) * 4 + wave) * BNT * BF) * 64 + lane;
#pragma unroll
  for (int nt = 0; nt < BNT; ++nt) {
#pragma unroll
    for (int f = 0; f < BF; ++f) {
      const bf16x8 w = wsrc[(nt * BF + f) * 64];
      if (f < BF_A) Wa[nt][f] = w;
      else Wv[nt][f - BF_A] = w;
    }
  }
  for (int i = tid; i < 2 * 2 * BGLD; i += 256) gS[i] = 0;

  const bool gate_lane = lane < 32;
  const int ul = 32 * wave + (lane & 31);
  const int unit = HALF * m + ul;
  const int len = lens[b];
  const float inj = dh_n ? dh_n[((int64_t)d * B + b) * PH + unit] : 0.f;
  float car = 0.f;   // dh_{t'} * z_{t'}
  struct In { float dy, r, z, n, hn, hp; };
  In inA, inB;       // operands of even / odd steps: each set is fetched TWO steps ahead
  auto time_of = [&](int s) {
    const int sc = s < T ? s : T - 1;
    return d == 0 ? T - 1 - sc : sc;
  };
  auto fetch = [&](In& in, int t) {
    const int tp = d == 0 ? t - 1 : t + 1;
    const int64_t bt = (int64_t)b * T + t;
    in.dy = dy[bt * DH + d * PH + unit];
    const float* gi = gates + (bt * D + d) * (int64_t)(3 * PH) + unit;
    in.r = gi[0];
    in.z = gi[PH];
    in.n = gi[2 * PH];
    in.hn = extra[(bt * D + d) * PH + unit];
    in.hp = (tp >= 0 && tp < T) ? y[((int64_t)b * T + tp) * DH + d * PH + unit] : 0.f;
  };
  if (gate_lane) {
    fetch(inA, time_of(0));
    fetch(inB, time_of(1));
  }
  // exchange buffers: [slot][pair][member][3 gates][HALF]
  u64* xmine = xch + ((int64_t)pair * 2 + m) * BOWN + ul;
  const u64* xtheirs = xch + ((int64_t)pair * 2 + (1 - m)) * BOWN + ul;
  const int64_t xslot = (int64_t)npairs * 2 * BOWN;
  int bad = 0;
  const bool local = pair_shares_xcd(xch + 2 * xslot + (int64_t)pair * 2, m, tid);   // (a barrier inside)

  auto step = [&](int s, In& in) {
    const int t = time_of(s);
    const int tnext = time_of(s + 2);
    bf16_t* gcur = gS + (s & 1) * 2 * BGLD;        // dGh of the step processed before this one: rows hi, lo
    bf16_t* gnxt = gS + ((s + 1) & 1) * 2 * BGLD;
    float prod = 0.f;
    if (s > 0) {
      const u64* xp = xtheirs + ((s - 1) & 1) * xslot;
      u64 first[3] = {0, 0, 0};
      if (!gate_lane) {
#pragma unroll
        for (int g = 0; g < 3; ++g) first[g] = peek(xp + g * HALF);
      }
      f32x4 acc0[BNT], acc1[BNT];   // hi / lo weight plane: dependent MFMAs are 4 issues apart
      const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
      auto afrag = [&](int ks) -> bf16x8 {   // rows 0 (hi) and 1 (lo) of the A operand; the other rows are zero
        bf16x8 v = zero8;
        if (col < 2) v = *reinterpret_cast<const bf16x8*>(gcur + col * BGLD + ks * 32 + kg * 8);
        return v;
      };
      // a k step is 4 MFMAs, less than an LDS round trip: keep the next three A fragments in flight
      bf16x8 aring[4];
#define LR_PAIR_KSTEPS(K0, K1)                                                                      \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) aring[((K0) + i) & 3] = afrag((K0) + i);        \
      _Pragma("unroll") for (int ks = (K0); ks < (K1); ++ks) {                                      \
        const bf16x8 a = aring[ks & 3];                                                             \
        if (ks + 3 < (K1)) aring[(ks + 3) & 3] = afrag(ks + 3);                                     \
        const int f0 = 2 * ks, f1 = 2 * ks + 1;                                                     \
        if (ks == 0) {                                                                              \
          LR_MFMA2_FIRST(acc0, a, Wa[0][0], Wa[1][0]);                                              \
          LR_MFMA2_FIRST(acc1, a, Wa[0][1], Wa[1][1]);                                              \
        } else if (f1 < BF_A) {                                                                     \
          LR_MFMA2(acc0, a, "a", Wa[0][f0], Wa[1][f0]);                                             \
          LR_MFMA2(acc1, a, "a", Wa[0][f1], Wa[1][f1]);                                             \
        } else {                                                                                    \
          LR_MFMA2(acc0, a, "v", Wv[0][f0 - BF_A], Wv[1][f0 - BF_A]);                               \
          LR_MFMA2(acc1, a, "v", Wv[0][f1 - BF_A], Wv[1][f1 - BF_A]);                               \
        }                                                                                           \
      }
      LR_PAIR_KSTEPS(0, 12)           // the member's own gate gradients
      if (!gate_lane && !bad) {       // the partner's: LDS positions BOWN + g*128 + ul, rows hi / lo
        // rounds of polls: whatever is still missing is asked for again IN PARALLEL (a serial re-poll per
        // granule would cost a memory round trip each)
        unsigned pend = 7u;
        for (int round = 0; pend; ++round) {
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            if (((pend >> g) & 1u) && (int)(first[g] >> 32) == s) {
              bf16_t hi, lo;
              split_bf16(__builtin_bit_cast(float, (unsigned)(first[g] & 0xffffffffu)), hi, lo);
              gcur[BOWN + g * HALF + ul] = hi;
              gcur[BGLD + BOWN + g * HALF + ul] = lo;
              pend &= ~(1u << g);
            }
          }
          if (!pend) break;
          if (round > SPIN_LIMIT) {
            bad = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
#pragma unroll
          for (int g = 0; g < 3; ++g)
            if ((pend >> g) & 1u) first[g] = peek(xp + g * HALF);
        }
      }
      lr_lds_barrier();
      LR_PAIR_KSTEPS(12, BKS)
#undef LR_PAIR_KSTEPS
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
      for (int nt = 0; nt < BNT; ++nt) {   // (reads stay behind the wait: see the forward kernel)
        asm volatile("" : "+v"(acc0[nt]));
        asm volatile("" : "+v"(acc1[nt]));
      }
      if (kg == 0) {   // rows 0 + 1: lanes 0..15, registers 0 and 1
#pragma unroll
        for (int nt = 0; nt < BNT; ++nt)
          S[32 * wave + 16 * nt + col] = (acc0[nt][0] + acc1[nt][0]) + (acc0[nt][1] + acc1[nt][1]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local exchange: units 32w .. 32w+31
      if (gate_lane) prod = S[ul];
    }
    if (gate_lane) {
      const float r = in.r, z = in.z, n = in.n, hn = in.hn, hp = in.hp;
      float dh = in.dy + prod + car;
      const bool is_last = d == 0 ? (t == len - 1) : (t == 0);
      if (is_last) dh += inj;
      fetch(in, tnext);
      float dr_pre = 0.f, dz_pre = 0.f, dn_pre = 0.f, dnr = 0.f;
      car = 0.f;
      if (t < len) {
        dn_pre = dh * (1.f - z) * (1.f - n * n);
        dr_pre = dn_pre * hn * r * (1.f - r);
        dz_pre = dh * (hp - n) * z * (1.f - z);
        dnr = dn_pre * r;
        car = dh * z;
      }
      u64* xo = xmine + (s & 1) * xslot;
      if (local) {
        publish_local(xo, dr_pre, s + 1);
        publish_local(xo + HALF, dz_pre, s + 1);
        publish_local(xo + 2 * HALF, dnr, s + 1);
      } else {
        publish(xo, dr_pre, s + 1);
        publish(xo + HALF, dz_pre, s + 1);
        publish(xo + 2 * HALF, dnr, s + 1);
      }
      bf16_t hi, lo;
      split_bf16(dr_pre, hi, lo);
      gnxt[ul] = hi;
      gnxt[BGLD + ul] = lo;
      split_bf16(dz_pre, hi, lo);
      gnxt[HALF + ul] = hi;
      gnxt[BGLD + HALF + ul] = lo;
      split_bf16(dnr, hi, lo);
      gnxt[2 * HALF + ul] = hi;
      gnxt[BGLD + 2 * HALF + ul] = lo;
      float* dgo = dG + (((int64_t)b * T + t) * D + d) * (int64_t)(4 * PH) + unit;
      dgo[0] = dr_pre;
      dgo[PH] = dz_pre;
      dgo[2 * PH] = dn_pre;
      dgo[3 * PH] = dnr;
    }
    lr_lds_barrier();   // gnxt's own half complete
  };
  for (int s = 0; s < T; s += 2) {
    step(s, inA);
    if (s + 1 < T) step(s + 1, inB);
  }
  if (bad && fault) atomicOr(fault, 1);   // device-side: this step is skipped (lr_common.h lr_fault_words)
}

constexpr int MAX_PAIRS = 64;   // per launch: 128 workgroups, one per CU, on a 256-CU chip (both members
                                // of every pair must be resident at once)
}  // namespace

// (the two members of a pair are blocks i and i + 8 of a group of 16 consecutive blocks: they are resident together
// on any device with at least 16 compute units)
int lr_gru256_pair_supported(int G, int B, int H) {
  return G == 3 && H == PH && B >= 1 && lr_device_cus() >= 16 && !lr_debug_pair_disabled() ? 1 : 0;
}

size_t lr_gru256_pair_pack_bytes(int D) { return (size_t)D * 2 * 4 * FNT * FF * 64 * sizeof(bf16x8); }
size_t lr_gru256_pair_bwd_pack_bytes(int D) { return (size_t)D * 2 * 4 * BNT * BF * 64 * sizeof(bf16x8); }
// exchange area: two parity slots of {value, tag} granules for min(B*D, MAX_PAIRS) pairs
size_t lr_gru256_pair_xch_bytes(int B, int D, int backward) {
  int pairs = B * D;
  if (pairs > MAX_PAIRS) pairs = MAX_PAIRS / D * D;
  return ((size_t)2 * pairs * 2 * (backward ? BOWN : HALF) + (size_t)pairs * 2) * sizeof(u64);   // two slots + the XCC ids
}

int lr_gru256_pair_forward(float* gates, float* extra, float* y, const float* const* w_hh, const float* const* b_hh,
                           const int32_t* lens, void* wpack, void* xch, int B, int T, int D, hipStream_t stream) {
  static bool attr_set = false;
  lr_clear_error();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gru256_fwd_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)FWD_LDS) != hipSuccess)
      return LR_ERR_LAUNCH;
    attr_set = true;
  }
  const int chunk = MAX_PAIRS / D;   // samples per launch
  {
    const int np0 = (B < chunk ? B : chunk) * D;   // the pack kernel clears the first launch's exchange buffer
    LR_LAUNCH(gru256_pair_pack_whh_kernel, dim3(192), dim3(256), 0, stream, w_hh[0], w_hh[D - 1], (bf16x8*)wpack, D,
              (u64*)xch, 2 * np0 * 2 * HALF + np0 * 2);
  }
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk, npairs = nb * D;
    if (b0 > 0 &&
        hipMemsetAsync(xch, 0, ((size_t)2 * npairs * 2 * HALF + (size_t)npairs * 2) * sizeof(u64), stream) != hipSuccess)
      return LR_ERR_LAUNCH;
    const dim3 grid(16 * ((npairs + 7) / 8));
    hipEvent_t e0, e1;
    if (b0 == 0 && lr_prof_next(LR_PROF_RNN_FWD, &e0, &e1))
      hipExtLaunchKernelGGL(gru256_fwd_pair_kernel, grid, dim3(256), FWD_LDS, stream, e0, e1, 0, gates, extra, y,
                            (const bf16x8*)wpack, b_hh[0], b_hh[D - 1], lens, (u64*)xch, lr_fault_words(),
                            lr_debug_drop_member_value(), b0, npairs, B, T, D);
    else
      hipLaunchKernelGGL(gru256_fwd_pair_kernel, grid, dim3(256), FWD_LDS, stream, gates, extra, y,
                         (const bf16x8*)wpack, b_hh[0], b_hh[D - 1], lens, (u64*)xch, lr_fault_words(),
                            lr_debug_drop_member_value(), b0, npairs, B, T, D);
    st = lr_launch_status();
    if (st != LR_OK) return st;
  }
  return LR_OK;
}

int lr_gru256_pair_backward(const float* gates, const float* extra, const float* y, const float* dy, const float* dh_n,
                            float* dG, const float* const* w_hh, const int32_t* lens, void* wpack, void* xch, int B,
                            int T, int D, hipStream_t stream) {
  static bool attr_set = false;
  lr_clear_error();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gru256_bwd_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)BWD_LDS) != hipSuccess)
      return LR_ERR_LAUNCH;
    attr_set = true;
  }
  const int chunk = MAX_PAIRS / D;
  {
    const int np0 = (B < chunk ? B : chunk) * D;
    LR_LAUNCH(gru256_pair_pack_whh_t_kernel, dim3(192), dim3(256), 0, stream, w_hh[0], w_hh[D - 1], (bf16x8*)wpack, D,
              (u64*)xch, 2 * np0 * 2 * BOWN + np0 * 2);
  }
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk, npairs = nb * D;
    if (b0 > 0 &&
        hipMemsetAsync(xch, 0, ((size_t)2 * npairs * 2 * BOWN + (size_t)npairs * 2) * sizeof(u64), stream) != hipSuccess)
      return LR_ERR_LAUNCH;
    const dim3 grid(16 * ((npairs + 7) / 8));
    hipEvent_t e0, e1;
    if (b0 == 0 && lr_prof_next(LR_PROF_RNN_BWD, &e0, &e1))
      hipExtLaunchKernelGGL(gru256_bwd_pair_kernel, grid, dim3(256), BWD_LDS, stream, e0, e1, 0, gates, extra, y, dy,
                            dh_n, dG, (const bf16x8*)wpack, lens, (u64*)xch, lr_fault_words(), lr_debug_drop_member_value(), b0,
                            npairs, B, T, D);
    else
      hipLaunchKernelGGL(gru256_bwd_pair_kernel, grid, dim3(256), BWD_LDS, stream, gates, extra, y, dy, dh_n, dG,
                         (const bf16x8*)wpack, lens, (u64*)xch, lr_fault_words(), lr_debug_drop_member_value(), b0, npairs,
                         B, T, D);
    st = lr_launch_status();
    if (st != LR_OK) return st;
  }
  return LR_OK;
}
