// lr_fgemm.hip — fp32 products straight from the tensors as they lie in memory: no packed operand planes.
//
// BUILD-DEFINED (no reference counterpart: the transformer encoder of BASELINE configs[4], SURVEY.md A10, whose CPU
// oracle is torch.nn.TransformerEncoder).  lr_xgemm.hip contracts PACKED bf16 hi / lo planes, K-contiguous: every
// product is preceded by pack launches — transposed ones for the weight gradients, whose contraction runs over the rows
// of both operands — and a transformer layer of 2400 x 256 activations is a dozen products of a few microseconds each,
// so round 4's stage was 102 pack launches + 51 contractions + 17 one-column "bias gradient" GEMMs per step.  Here a
// workgroup takes its operand tiles from the fp32 (or bf16) tensors themselves and splits them on the way into LDS:
//
//   C[M][N] = epilogue( sum_k op(A)[m][k] * op(B)[k][n] ),   fp32 accumulation, three operand forms in one kernel
//     NT  A [M][K], B [N][K]      x . W^T                  (forward of a Linear)
//     NN  A [M][K], B [K][N]      dy . W                   (its data gradient)
//     TN  A [K][M], B [K][N]      dy^T . x                 (its weight gradient; the column sums of dy — the bias
//                                                           gradient — fall out of the A tiles the workgroup stages)
//   precision X3: each fp32 element -> bf16 hi + lo, a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_bf16
//                 (~1e-5 relative, lr_xgemm.hip's arithmetic); an operand STORED as bf16 is its own hi plane
//   precision F32: v_mfma_f32_32x32x2_f32 on the fp32 tiles (exact fp32 products; the landmark regimes' choice)
//   epilogue: alpha, + bias[n], + addend[(m % period)][n] (residual / positional table), ReLU, mask (out = mask > 0 ?
//             out : 0: the ReLU gradient), beta * C, fp32 or bf16 output; split-K through slabs + one combine launch
//
// An operand whose K axis is the SLOW one in memory (B of NN, both of TN) stays K-major in LDS — planes of [k][32
// columns], 64-byte rows — and gfx950's ds_read_b64_tr_b16 delivers the K-contiguous MFMA fragments (16 lanes read a
// [4 k][16 columns] block and lane L receives column L: two reads = one operand; lr_conv_wgrad.hip uses the same
// read).  So every tensor of a layer is split where it is consumed and nothing is ever packed or transposed in memory.
//
// Tiles: 128 x 128 per workgroup (2 x 2 waves of 64 x 64 = 2 x 2 MFMA tiles), 32 k per stage, one stage of global loads
// in flight under the MFMAs of the previous one.  Several products of one form share a launch (a job table in the
// kernel arguments: the four weight gradients of every layer are ONE launch of ~200 workgroups).
#include "lr_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;

constexpr int FBM = 128, FBN = 128, FBK = 32;
constexpr int KC_LD = FBK + 8;        // bf16 per row of a K-contiguous plane tile (80-byte rows)
constexpr int KC_LDF = FBK + 1;       // floats per row of a K-contiguous fp32 tile
constexpr int KS_LDF = 128;           // floats per k row of a K-strided fp32 tile
// bytes of one operand's LDS image (the largest of its forms)
constexpr int OP_BYTES_X3 = 2 * FBM * KC_LD * 2;     // hi + lo planes, K-contiguous form: 20480 (K-strided: 16384)
constexpr int OP_BYTES_F32 = FBM * KC_LDF * 4;       // 16896 (K-strided: 16384)

enum { PREC_X3 = 0, PREC_F32 = 1 };

struct FJob {
  const void* A;
  const void* B;
  float* C;              // fp32 [M][ldc], or bf16 with LR_FGEMM_C_BF16
  const float* bias;     // [N] or nullptr
  const float* addend;   // out += addend[(row % add_period) * ldadd + col], or nullptr
  const float* mask;     // out = mask[row * ldmask + col] > 0 ? out : 0, or nullptr
  float* colsum;         // TN only: colsum[m] (+)= sum_k A[k][m], or nullptr
  float* slabs;          // split-K partial sums [splits][M][N], or nullptr
  int M, N, K, lda, ldb, ldc, ldadd, add_period, ldmask;
  int flags, splits, k_chunk, tile0, ny, tiles;
  int vec_a, vec_b;      // 0: element loads; 1: 16-byte (fp32) / 8-byte (bf16) loads where a quad is whole; 2: whole quads only
  float alpha, beta;
};
constexpr int FMAXJOBS = 20;
struct FArgs {
  FJob j[FMAXJOBS];
  int njobs;
};

__device__ __forceinline__ float bf2f(bf16_t b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// four consecutive elements along the operand's contiguous axis; `nvalid` of them exist (<= 0: none)
template <bool BF>
__device__ __forceinline__ float4 ld4(const void* base, int64_t off, int nvalid, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid <= 0) return v;
  if (BF) {
    const bf16_t* p = reinterpret_cast<const bf16_t*>(base) + off;
    if (vec && nvalid >= 4) {
      const uint2 u = *reinterpret_cast<const uint2*>(p);
      v.x = __builtin_bit_cast(float, u.x << 16);
      v.y = __builtin_bit_cast(float, u.x & 0xffff0000u);
      v.z = __builtin_bit_cast(float, u.y << 16);
      v.w = __builtin_bit_cast(float, u.y & 0xffff0000u);
    } else {
      v.x = bf2f(p[0]);
      if (nvalid > 1) v.y = bf2f(p[1]);
      if (nvalid > 2) v.z = bf2f(p[2]);
      if (nvalid > 3) v.w = bf2f(p[3]);
    }
  } else {
    const float* p = reinterpret_cast<const float*>(base) + off;
    if (vec && nvalid >= 4) {
      v = *reinterpret_cast<const float4*>(p);
    } else {
      v.x = p[0];
      if (nvalid > 1) v.y = p[1];
      if (nvalid > 2) v.z = p[2];
      if (nvalid > 3) v.w = p[3];
    }
  }
  return v;
}

// four values -> bf16 hi (8 bytes) and lo (8 bytes)
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  const bf16x2 h0 = __builtin_convertvector((f32x2){v.x, v.y}, bf16x2);
  const bf16x2 h1 = __builtin_convertvector((f32x2){v.z, v.w}, bf16x2);
  const bf16x2 l0 = __builtin_convertvector((f32x2){v.x - (float)h0[0], v.y - (float)h0[1]}, bf16x2);
  const bf16x2 l1 = __builtin_convertvector((f32x2){v.z - (float)h1[0], v.w - (float)h1[1]}, bf16x2);
  hi = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
  lo = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}

// One operand's tile: 128 rows (m or n) x 32 k.
//   KS == false: stored [row][k] in memory (K contiguous).  Thread unit e = tid + 256 i: row e >> 3, k quad e & 7.
//   KS == true:  stored [k][row] (K strided).  Wave w moves the 32 columns of plane w: k = 8 i + (lane >> 3), column
//                quad lane & 7 — a wave's store of one pass is 512 contiguous bytes of its plane.
template <int PREC, bool KS, bool BF>
struct Tile {
  // ---- global -> registers --------------------------------------------------------------------------------------
  // `fast` (workgroup-uniform, the common case): whole aligned quads only — every load is UNCONDITIONAL and nothing is
  // selected behind it: a quad outside the matrix is read from a CLAMPED address instead (rows / columns past the edge
  // feed accumulators that are never stored; the K tail is zeroed on the way into LDS, store() below), so an operand's
  // four loads are in flight together and stay in flight under the previous stage's MFMAs.  A predicate around each load
  // — the guarded path, for leading dimensions or extents that are not multiples of four — makes hipcc close every
  // branch with a wait for its load (measured: 5.6 us per 32-k stage, ten times the stage's MFMAs), and a select right
  // behind an unconditional load puts the wait there as well.
  static __device__ __forceinline__ float4 quad(const void* base, int64_t off) {
    float4 v;
    if (BF) {
      const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(base) + off);
      v.x = __builtin_bit_cast(float, u.x << 16);
      v.y = __builtin_bit_cast(float, u.x & 0xffff0000u);
      v.z = __builtin_bit_cast(float, u.y << 16);
      v.w = __builtin_bit_cast(float, u.y & 0xffff0000u);
    } else {
      v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
    }
    return v;
  }
  // K: the operand's whole contraction extent (clamp limit); kend: where this workgroup's K range ends
  static __device__ __forceinline__ void load(float4 (&r)[4], const void* base, int ld, int row0, int rows, int k0,
                                              int kend, int K, int how, int tid) {
    const bool fast = how == 2, vec = how != 0;
    if (KS) {
      const int lane = tid & 63, col = row0 + 32 * (tid >> 6) + 4 * (lane & 7);
      if (fast) {
        const int cc = min(col, rows - 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = quad(base, (int64_t)min(k0 + 8 * i + (lane >> 3), K - 1) * ld + cc);
        return;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + 8 * i + (lane >> 3);
        r[i] = ld4<BF>(base, (int64_t)k * ld + col, k < kend ? rows - col : 0, vec);
      }
    } else {
      if (fast) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = tid + 256 * i;
          r[i] = quad(base, (int64_t)min(row0 + (e >> 3), rows - 1) * ld + min(k0 + 4 * (e & 7), K - 4));
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, row = row0 + (e >> 3), k = k0 + 4 * (e & 7);
        r[i] = ld4<BF>(base, (int64_t)row * ld + k, row < rows ? kend - k : 0, vec);
      }
    }
  }
  // the K tail of a `fast` operand (only a K range's last stage has one: workgroup-uniform): quads at or past kend -> 0
  static __device__ __forceinline__ void zero_tail(float4 (&r)[4], int k0, int kend, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = KS ? k0 + 8 * i + ((tid & 63) >> 3) : k0 + 4 * ((tid + 256 * i) & 7);
      if (k >= kend) r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // ---- registers -> LDS -----------------------------------------------------------------------------------------
  static __device__ __forceinline__ void store(const float4 (&r)[4], unsigned char* lds, int tid) {
    if (PREC == PREC_X3) {
      bf16_t* hi = reinterpret_cast<bf16_t*>(lds);
      bf16_t* lo = hi + (KS ? 4 * FBK * 32 : FBM * KC_LD);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int o;
        if (KS) {
          const int lane = tid & 63;
          o = (tid >> 6) * (FBK * 32) + (8 * i + (lane >> 3)) * 32 + 4 * (lane & 7);
        } else {
          const int e = tid + 256 * i;
          o = (e >> 3) * KC_LD + 4 * (e & 7);
        }
        uint2 h, l;
        split4(r[i], h, l);
        *reinterpret_cast<uint2*>(hi + o) = h;
        if (!BF) *reinterpret_cast<uint2*>(lo + o) = l;
      }
    } else {
      float* f = reinterpret_cast<float*>(lds);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KS) {
          const int lane = tid & 63;
          *reinterpret_cast<float4*>(f + (8 * i + (lane >> 3)) * KS_LDF + 32 * (tid >> 6) + 4 * (lane & 7)) = r[i];
        } else {
          const int e = tid + 256 * i;
          float* p = f + (e >> 3) * KC_LDF + 4 * (e & 7);
          p[0] = r[i].x; p[1] = r[i].y; p[2] = r[i].z; p[3] = r[i].w;
        }
      }
    }
  }
  // ---- LDS -> MFMA fragments ------------------------------------------------------------------------------------
  // X3: the bf16x8 operand of k16 step `ks` (0, 1) for the 32-row sub-tile `sub` (0..3) of plane `pl` (0 hi, 1 lo)
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* lds, int sub, int ks, int pl, int lane) {
    if (KS) {
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const int g = lane >> 4, i = lane & 15;
      const unsigned char* p = lds + pl * (4 * FBK * 64) + sub * (FBK * 64) +
                               (16 * ks + 8 * (g >> 1) + (i >> 2)) * 64 + (16 * (g & 1) + 4 * (i & 3)) * 2;
      const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
      const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 256));
      return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    }
    const bf16_t* p = reinterpret_cast<const bf16_t*>(lds) + pl * (FBM * KC_LD) + (sub * 32 + (lane & 31)) * KC_LD +
                      16 * ks + 8 * (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(p);
  }
  // F32: the operand of k2 step s (0..15): row lane & 31 of the sub-tile, k = 2 s + (lane >> 5)
  static __device__ __forceinline__ float fragf(const unsigned char* lds, int sub, int s, int lane) {
    const float* f = reinterpret_cast<const float*>(lds);
    if (KS) return f[(2 * s + (lane >> 5)) * KS_LDF + sub * 32 + (lane & 31)];
    return f[(sub * 32 + (lane & 31)) * KC_LDF + 2 * s + (lane >> 5)];
  }
};

template <int PREC, bool TA, bool TB, bool ABF, bool BBF>
__global__ __launch_bounds__(256, 2) void fgemm_kernel(const FArgs args) {
  constexpr int OPB = PREC == PREC_X3 ? OP_BYTES_X3 : OP_BYTES_F32;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OPB];
  unsigned char* As = smem;
  unsigned char* Bs = smem + OPB;
  typedef Tile<PREC, TA, ABF> TileA;      // A^T is stored [K][M]: K strided
  typedef Tile<PREC, !TB, BBF> TileB;     // B stored [K][N] unless TB ([N][K])
  int ji = 0;
  for (int q = 1; q < args.njobs; ++q)
    if ((int)blockIdx.x >= args.j[q].tile0) ji = q;      // workgroup-uniform
  const FJob& g = args.j[ji];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int t = (int)blockIdx.x - g.tile0;
  const int zs = t / g.tiles, tt = t - zs * g.tiles;
  const int m0 = (tt % g.ny) * FBM, n0 = (tt / g.ny) * FBN;
  const int kbeg = zs * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
  const bool do_colsum = TA && g.colsum != nullptr && n0 == 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);

  // the K loop in two instantiations: FAST (both operands in whole aligned quads: the straight-line loop with no
  // predicate around any load) and the guarded one; the choice is workgroup-uniform
  auto k_loop = [&](auto fast_tag) {
  constexpr bool FAST = decltype(fast_tag)::value;
  const int how_a = FAST ? 2 : (g.vec_a == 2 ? 1 : g.vec_a), how_b = FAST ? 2 : (g.vec_b == 2 ? 1 : g.vec_b);
  float4 ra[4], rb[4];
  TileA::load(ra, g.A, g.lda, m0, g.M, kbeg, kend, g.K, how_a, tid);
  TileB::load(rb, g.B, g.ldb, n0, g.N, kbeg, kend, g.K, how_b, tid);
  for (int k0 = kbeg; k0 < kend; k0 += FBK) {
    lr_lds_barrier();                       // every wave has read the previous stage
    if (FAST && k0 + FBK > kend) {          // (workgroup-uniform) the K tail of operands loaded without predicates
      TileA::zero_tail(ra, k0, kend, tid);
      TileB::zero_tail(rb, k0, kend, tid);
    }
    TileA::store(ra, As, tid);
    TileB::store(rb, Bs, tid);
    if (do_colsum) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        cs.x += ra[i].x; cs.y += ra[i].y; cs.z += ra[i].z; cs.w += ra[i].w;
      }
    }
    lr_lds_barrier();
    if (k0 + FBK < kend) {                  // the next stage's loads fly under this stage's MFMAs
      TileA::load(ra, g.A, g.lda, m0, g.M, k0 + FBK, kend, g.K, how_a, tid);
      TileB::load(rb, g.B, g.ldb, n0, g.N, k0 + FBK, kend, g.K, how_b, tid);
    }
    if (PREC == PREC_X3) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ah[i] = TileA::frag(As, 2 * wm + i, ks, 0, lane);
          if (!ABF) al[i] = TileA::frag(As, 2 * wm + i, ks, 1, lane);
          bh[i] = TileB::frag(Bs, 2 * wn + i, ks, 0, lane);
          if (!BBF) bl[i] = TileB::frag(Bs, 2 * wn + i, ks, 1, lane);
        }
        // small terms first, so that they are not absorbed by a large partial sum
        if (!BBF) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        }
        if (!ABF) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < FBK / 2; ++s) {
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = TileA::fragf(As, 2 * wm + i, s, lane);
          b[i] = TileB::fragf(Bs, 2 * wn + i, s, lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  };
  if (g.vec_a == 2 && g.vec_b == 2) k_loop(std::true_type{});
  else k_loop(std::false_type{});

  // ---- column sums of A (the bias gradient of a weight-gradient product): thread (wave w, lane) holds columns
  // 32 w + 4 (lane & 7) .. + 3 of its k rows; fold the 8 row groups of a wave in a fixed order -------------------------
  if (do_colsum) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      cs.x += __shfl_xor(cs.x, o, 64);
      cs.y += __shfl_xor(cs.y, o, 64);
      cs.z += __shfl_xor(cs.z, o, 64);
      cs.w += __shfl_xor(cs.w, o, 64);
    }
    if (lane < 8) {
      const int c = m0 + 32 * wave + 4 * lane;
      const float v[4] = {cs.x, cs.y, cs.z, cs.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < g.M) g.colsum[c + e] = g.beta != 0.f ? g.beta * g.colsum[c + e] + v[e] : v[e];
    }
  }

  // ---- epilogue: C/D layout of a 32 x 32 tile: column = lane & 31, row of register r = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  // What the epilogue reads — addend, mask, C for beta — is fetched for eight registers at a time, from clamped rows
  // and through 32-bit element offsets from the (uniform) base pointers, so that the loads are in flight together
  // (one load, one wait, one store per element is a memory round trip for each of a thread's 64 outputs).
  const int lr = lane & 31, lk = lane >> 5;
  const bool wrap = g.addend && g.add_period < g.M;    // workgroup-uniform: a periodic table (else a residual: row itself)
  const bool has_add = g.addend != nullptr, has_mask = g.mask != nullptr, has_beta = g.beta != 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + lr;
      const bool cok = col < g.N;
      const int ccol = cok ? col : 0;
      const int rbase = m0 + wm * 64 + i * 32 + 4 * lk;
      const float bv = (g.bias && !g.slabs && cok) ? g.bias[ccol] : 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float add[8], mk[8], old[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = 8 * h + q;
          const int row = min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1);
          add[q] = 0.f; mk[q] = 1.f; old[q] = 0.f;
          if (g.slabs) continue;
          if (has_add) add[q] = g.addend[(unsigned)((wrap ? row % g.add_period : row) * g.ldadd + ccol)];
          if (has_mask) mk[q] = g.mask[(unsigned)(row * g.ldmask + ccol)];
          if (has_beta) old[q] = g.C[(unsigned)(row * g.ldc + ccol)];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = 8 * h + q;
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          const bool ok = cok && row < g.M;
          if (g.slabs) {
            if (ok) g.slabs[((int64_t)zs * g.M + row) * g.N + col] = acc[i][j][r];
            continue;
          }
          float out = g.alpha * acc[i][j][r] + bv + add[q];
          if (g.flags & LR_FGEMM_RELU) out = fmaxf(out, 0.f);
          out = mk[q] > 0.f ? out : 0.f;
          if (g.flags & LR_FGEMM_C_BF16) {
            const __bf16 hb = (__bf16)out;
            if (ok) reinterpret_cast<bf16_t*>(g.C)[(unsigned)(row * g.ldc + col)] = __builtin_bit_cast(bf16_t, hb);
          } else if (ok) {
            g.C[(unsigned)(row * g.ldc + col)] = out + g.beta * old[q];
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // eight outputs' operands at a time (a whole tile's at once spills)
      }
    }
}

// split-K combine (fixed order over the slabs) + the epilogue
__global__ __launch_bounds__(256) void fgemm_reduce_kernel(const FJob g) {
  const int64_t total = (int64_t)g.M * g.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < g.splits; ++z) s += g.slabs[(int64_t)z * total + i];
    const int row = (int)(i / g.N), col = (int)(i - (int64_t)row * g.N);
    float out = g.alpha * s + (g.bias ? g.bias[col] : 0.f);
    if (g.addend) out += g.addend[(int64_t)(row % g.add_period) * g.ldadd + col];
    if (g.flags & LR_FGEMM_RELU) out = fmaxf(out, 0.f);
    if (g.mask) out = g.mask[(int64_t)row * g.ldmask + col] > 0.f ? out : 0.f;
    if (g.flags & LR_FGEMM_C_BF16) {
      const __bf16 hb = (__bf16)out;
      reinterpret_cast<bf16_t*>(g.C)[(int64_t)row * g.ldc + col] = __builtin_bit_cast(bf16_t, hb);
      continue;
    }
    float* c = g.C + (int64_t)row * g.ldc + col;
    if (g.beta != 0.f) out += g.beta * *c;
    *c = out;
  }
}

bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

}  // namespace

// ---- host side (internal C++ interface, lr_common.h) ------------------------------------------------------------------
size_t lr_fgemm_slab_floats(int M, int N, int splits) { return splits > 1 ? (size_t)splits * M * N : 0; }

// K split so that a product of few tiles and a long K (the K = 3456 input projection: 38 tiles) fills the chip;
// >= 8 stages per split, never more splits than bring the launch to ~2 workgroups per CU
int lr_fgemm_want_splits(int M, int N, int K) {
  const long tiles = (long)((M + FBM - 1) / FBM) * ((N + FBN - 1) / FBN);
  const long stages = (K + FBK - 1) / FBK;
  if (tiles >= 192 || stages < 32) return 1;
  long sp = 512 / tiles;
  if (sp > stages / 8) sp = stages / 8;
  if (sp > 16) sp = 16;
  return sp < 1 ? 1 : (int)sp;
}

int lr_fgemm_launch(int prec, int form, int a_bf16, int b_bf16, const lr_fgemm_job* jobs, int njobs, hipStream_t stream) {
  LR_CHECK_ARG(jobs && njobs >= 1 && njobs <= FMAXJOBS && (prec == 0 || prec == 1) && form >= 0 && form <= 2);
  // instantiated: NT with either A form, NN, TN with either B form
  if ((a_bf16 && form != LR_FGEMM_NT) || (b_bf16 && form != LR_FGEMM_TN)) return LR_ERR_UNSUPPORTED;
  FArgs a;
  a.njobs = njobs;
  int tile0 = 0;
  for (int q = 0; q < njobs; ++q) {
    const lr_fgemm_job& s = jobs[q];
    FJob& g = a.j[q];
    LR_CHECK_ARG(s.A && s.B && s.C && s.M > 0 && s.N > 0 && s.K > 0 && s.lda > 0 && s.ldb > 0 && s.ldc >= s.N);
    LR_CHECK_ARG(!s.addend || (s.add_period > 0 && s.ldadd >= s.N));
    LR_CHECK_ARG(!s.mask || s.ldmask >= s.N);
    LR_CHECK_ARG(!s.colsum || (form == LR_FGEMM_TN && s.splits <= 1));
    LR_CHECK_ARG(s.splits <= 1 || s.slabs);
    LR_CHECK_ARG(!(s.flags & LR_FGEMM_C_BF16) || s.beta == 0.f);
    g.A = s.A; g.B = s.B; g.C = (float*)s.C; g.bias = s.bias; g.addend = s.addend; g.mask = s.mask; g.colsum = s.colsum;
    g.M = s.M; g.N = s.N; g.K = s.K; g.lda = s.lda; g.ldb = s.ldb; g.ldc = s.ldc; g.ldadd = s.ldadd;
    g.add_period = s.add_period > 0 ? s.add_period : 1; g.ldmask = s.ldmask; g.flags = s.flags;
    g.alpha = s.alpha; g.beta = s.beta;
    int splits = s.splits > 1 ? s.splits : 1;
    int chunk = ((s.K + splits - 1) / splits + FBK - 1) / FBK * FBK;
    splits = (s.K + chunk - 1) / chunk;
    g.splits = splits; g.k_chunk = chunk;
    g.slabs = splits > 1 ? s.slabs : nullptr;
    g.ny = (s.M + FBM - 1) / FBM;
    g.tiles = g.ny * ((s.N + FBN - 1) / FBN);
    g.tile0 = tile0;
    tile0 += g.tiles * splits;
    const size_t ea = a_bf16 ? 2 : 4, eb = b_bf16 ? 2 : 4;
    g.vec_a = (s.lda % 4 == 0) && aligned_to(s.A, 4 * ea);
    g.vec_b = (s.ldb % 4 == 0) && aligned_to(s.B, 4 * eb);
    // whole quads only: the contiguous extent of the operand is a multiple of four as well (A: K, or M when stored
    // [K][M]; B: K when stored [N][K], else N)
    if (g.vec_a && (form == LR_FGEMM_TN ? s.M : s.K) % 4 == 0) g.vec_a = 2;
    if (g.vec_b && (form == LR_FGEMM_NT ? s.K : s.N) % 4 == 0) g.vec_b = 2;
  }
  lr_clear_error();
  const dim3 grid(tile0), block(256);
#define LR_FG(P, TA_, TB_, AB, BB) hipLaunchKernelGGL((fgemm_kernel<P, TA_, TB_, AB, BB>), grid, block, 0, stream, a)
  if (prec == PREC_X3) {
    if (form == LR_FGEMM_NT) { if (a_bf16) LR_FG(PREC_X3, false, true, true, false); else LR_FG(PREC_X3, false, true, false, false); }
    else if (form == LR_FGEMM_NN) LR_FG(PREC_X3, false, false, false, false);
    else { if (b_bf16) LR_FG(PREC_X3, true, false, false, true); else LR_FG(PREC_X3, true, false, false, false); }
  } else {
    if (form == LR_FGEMM_NT) { if (a_bf16) LR_FG(PREC_F32, false, true, true, false); else LR_FG(PREC_F32, false, true, false, false); }
    else if (form == LR_FGEMM_NN) LR_FG(PREC_F32, false, false, false, false);
    else { if (b_bf16) LR_FG(PREC_F32, true, false, false, true); else LR_FG(PREC_F32, true, false, false, false); }
  }
#undef LR_FG
  int st = lr_launch_status();
  if (st != LR_OK) return st;
  for (int q = 0; q < njobs; ++q) {
    const FJob& g = a.j[q];
    if (g.splits <= 1) continue;
    FJob r = g;
    r.bias = jobs[q].bias;
    const int64_t total = (int64_t)g.M * g.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    LR_LAUNCH(fgemm_reduce_kernel, dim3(blocks), dim3(256), 0, stream, r);
    st = lr_launch_status();
    if (st != LR_OK) return st;
  }
  return LR_OK;
}

extern "C" int lr_fgemm(int prec, int form, int a_bf16, int b_bf16, const lr_fgemm_job* jobs, int njobs,
                        lr_stream_t stream) {
  return lr_fgemm_launch(prec, form, a_bf16, b_bf16, jobs, njobs, (hipStream_t)stream);
}
extern "C" int lr_fgemm_splits(int M, int N, int K) { return lr_fgemm_want_splits(M, N, K); }
