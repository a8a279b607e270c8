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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;

constexpr int FBM = 128, FBN = 128, FBK = 32;
constexpr int KC_LD = FBK + 8;        // bf16 per row of a K-contiguous plane tile (80-byte rows)
constexpr int KC_LDF = FBK + 1;       // floats per row of a K-contiguous fp32 tile
constexpr int KS_LDF = 128;           // floats per k row of a K-strided fp32 tile
// bytes of one operand's LDS image (the largest of its forms)
constexpr int OP_BYTES_X3 = 2 * FBM * KC_LD * 2;     // hi + lo planes, K-contiguous form: 20480 (K-strided: 16384)
constexpr int OP_BYTES_F32 = FBM * KC_LDF * 4;       // 16896 (K-strided: 16384)
constexpr int C_LDS = FBN + 4;        // floats per row of the epilogue's staging tile

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
  int vec_c;             // the epilogue's row-contiguous accesses (C, slabs, bias, addend, mask) are whole aligned quads
  int b_shift, b_period; // B stored [K][N]: row k is read from row k + b_shift, as zeros where (k % b_period) + b_shift
                         // leaves [0, b_period) (B already points b_shift rows off); b_period 0: off
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
  // ---- global -> registers, FAST form: buffer loads (see the kernel) ------------------------------------------------
  static constexpr int ES = BF ? 2 : 4;     // bytes per element
  // byte offset of each of the thread's four quads from (matrix base + the stage's scalar offset); bit 31: the quad's row
  // (K contiguous) / columns (K strided) lie past the matrix edge — never in range, always zeros
  static __device__ __forceinline__ void offsets(unsigned (&voff)[4], int ld, int row0, int rows, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (KS) {
        const int lane = tid & 63, col = row0 + 32 * (tid >> 6) + 4 * (lane & 7), kr = 8 * i + (lane >> 3);
        voff[i] = col < rows ? (unsigned)((kr * ld + col) * ES) : 0x80000000u;
      } else {
        const int e = tid + 256 * i, row = row0 + (e >> 3);
        voff[i] = row < rows ? (unsigned)((row * ld + 4 * (e & 7)) * ES) : 0x80000000u;
      }
    }
  }
  static __device__ __forceinline__ unsigned stage_offset(int k0, int ld) { return (unsigned)(KS ? k0 * ld : k0) * ES; }
  // krem = kend - k0: a quad whose k (relative to the stage) is at or past it is the K tail -> bit 31 -> zeros.  Only a K
  // range's last stages have one (workgroup-uniform branch); every other stage's loads have no VALU instruction at all.
  // kill: bit i = quad i is zeros whatever its address (a row whose shifted neighbour lies across a sequence end)
  static __device__ __forceinline__ void loadf(float4 (&r)[4], __amdgpu_buffer_rsrc_t rsrc, const unsigned (&voff)[4],
                                               unsigned soff, int krem, int tid, unsigned kill = 0u) {
    unsigned o[4] = {voff[0], voff[1], voff[2], voff[3]};
    if (kill) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((kill >> i) & 1u) o[i] = 0x80000000u;
    }
    if (krem < FBK) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = KS ? 8 * i + ((tid & 63) >> 3) : 4 * ((tid + 256 * i) & 7);
        if (k >= krem) o[i] = 0x80000000u;
      }
      if (krem <= 0) soff = 0;     // (a stage past the end of the K range: nothing but zeros, whatever its offset)
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (BF) {
        const u32x2 u = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)o[i], (int)soff, 0));
        r[i].x = __builtin_bit_cast(float, u[0] << 16);
        r[i].y = __builtin_bit_cast(float, u[0] & 0xffff0000u);
        r[i].z = __builtin_bit_cast(float, u[1] << 16);
        r[i].w = __builtin_bit_cast(float, u[1] & 0xffff0000u);
      } else {
        // (the WHOLE vector is reinterpreted: taking the builtin's result apart element by element makes this hipcc
        // narrow the load to one dword and use it for all four — seen in the ISA, not in any diagnostic)
        const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o[i], (int)soff, 0));
        r[i] = make_float4(f[0], f[1], f[2], f[3]);
      }
    }
  }
  // ---- global -> registers, guarded form: element (or whole-quad) loads under predicates ----------------------------
  static __device__ __forceinline__ void load(float4 (&r)[4], const void* base, int ld, int row0, int rows, int k0,
                                              int kend, bool vec, int tid, int shift = 0, int period = 0) {
    if (KS) {
      const int lane = tid & 63, col = row0 + 32 * (tid >> 6) + 4 * (lane & 7);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + 8 * i + (lane >> 3);
        bool ok = k < kend;
        if (period > 0) {
          const int tt = k % period + shift;
          ok = ok && tt >= 0 && tt < period;
        }
        r[i] = ld4<BF>(base, (int64_t)k * ld + col, ok ? rows - col : 0, vec);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, row = row0 + (e >> 3), k = k0 + 4 * (e & 7);
        r[i] = ld4<BF>(base, (int64_t)row * ld + k, row < rows ? kend - k : 0, vec);
      }
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
__global__ __launch_bounds__(256, 1) void fgemm_kernel(const FArgs args) {
  constexpr int OPB = PREC == PREC_X3 ? OP_BYTES_X3 : OP_BYTES_F32;
  // two stage buffers (A + B each); the epilogue stages 64 x 128 outputs at a time through the same memory
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * OPB];
  static_assert(4 * OPB >= 64 * C_LDS * 4, "the epilogue's staging tile fits the stage buffers");
  typedef Tile<PREC, TA, ABF> TileA;      // A^T is stored [K][M]: K strided
  typedef Tile<PREC, !TB, BBF> TileB;     // B stored [K][N] unless TB ([N][K])
  int ji = 0;
  for (int q = 1; q < args.njobs; ++q)
    if ((int)blockIdx.x >= args.j[q].tile0) ji = q;      // workgroup-uniform
  // The job's fields as LOCAL scalars, read once.  Through a reference into the kernel arguments hipcc re-loads a field
  // at every use behind a global store (1003 scalar loads + waits in the first cut, ~6 per output element of the
  // epilogue).
  struct Job {
    const void *A, *B;
    float* C;
    const float *bias, *addend, *mask;
    float *colsum, *slabs;
    int M, N, K, lda, ldb, ldc, ldadd, add_period, ldmask, flags, k_chunk, tile0, ny, tiles, vec_a, vec_b, vec_c, b_shift,
        b_period, splits;
    float alpha, beta;
  } g;
  {
    const FJob& j = args.j[ji];
    g.A = j.A; g.B = j.B; g.C = j.C; g.bias = j.bias; g.addend = j.addend; g.mask = j.mask; g.colsum = j.colsum;
    g.slabs = j.slabs; g.M = j.M; g.N = j.N; g.K = j.K; g.lda = j.lda; g.ldb = j.ldb; g.ldc = j.ldc; g.ldadd = j.ldadd;
    g.add_period = j.add_period; g.ldmask = j.ldmask; g.flags = j.flags; g.k_chunk = j.k_chunk; g.tile0 = j.tile0;
    g.ny = j.ny; g.tiles = j.tiles; g.vec_a = j.vec_a; g.vec_b = j.vec_b; g.vec_c = j.vec_c; g.alpha = j.alpha;
    g.b_shift = j.b_shift; g.b_period = j.b_period; g.splits = j.splits;
    g.beta = j.beta;
  }
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

  // one stage's contraction out of the stage buffer at `buf`
  auto contract = [&](const unsigned char* buf) __attribute__((always_inline)) {
    const unsigned char* As = buf;
    const unsigned char* Bs = buf + OPB;
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
  };
  auto add_colsum = [&](const float4 (&r)[4]) __attribute__((always_inline)) {
    if (do_colsum) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        cs.x += r[i].x; cs.y += r[i].y; cs.z += r[i].z; cs.w += r[i].w;
      }
    }
  };

  if (g.vec_a == 2 && g.vec_b == 2) {
    // ---- FAST (workgroup-uniform: both operands in whole aligned quads) ------------------------------------------------
    // These products leave one workgroup per compute unit at best (38 .. 152 tiles on 256 CUs): one wave per SIMD with
    // nothing to switch to, so everything a stage does besides its 24 MFMAs has to ride UNDER them in the same wave.
    //   * loads are buffer loads from a per-thread byte offset fixed at the start + the stage's SCALAR offset: no
    //     address arithmetic in the loop; a quad outside the matrix (rows / columns past the edge, the K tail) has bit 31
    //     set in its offset, is out of the resource's range and comes back as zeros — no predicate, no select;
    //   * two register sets, loads issued two stages ahead (the wait in front of a stage's LDS write is a counted one);
    //   * two stage buffers in LDS: while the MFMAs of stage s read buffer s & 1, the split (fp32 -> bf16 hi + lo) and the
    //     LDS writes of stage s + 1 go to the other one in the same basic block; ONE barrier per stage.
    // Measured on the way here (M 2400, N 768): one buffer + one register set 1.17 us per 32-k stage; two register
    // sets 0.95; a launch's fixed cost 27 us -> 19 us when the job's fields moved out of the kernel arguments' memory,
    // and the rest of it went with the unrolled per-register epilogue (see below).
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), (short)0, 0x7fffffff, 0x00020000);
    unsigned offa[4], offb[4];
    TileA::offsets(offa, g.lda, m0, g.M, tid);
    TileB::offsets(offb, g.ldb, n0, g.N, tid);
    float4 ra[2][4], rb[2][4];
    // (B read through a row shift: the phase of each of the thread's four rows within its period, advanced by a stage
    // per load_stage call — the calls come in stage order —, no division in the loop)
    int bt[4] = {0, 0, 0, 0};
    if (!TB && g.b_period > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bt[i] = (kbeg + 8 * i + (lane >> 3)) % g.b_period;
    }
    auto load_stage = [&](int q, int k0) __attribute__((always_inline)) {
      unsigned kill = 0u;
      if (!TB && g.b_period > 0) {     // workgroup-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int tt = bt[i] + g.b_shift;
          if (tt < 0 || tt >= g.b_period) kill |= 1u << i;
          bt[i] += FBK;
          if (bt[i] >= g.b_period) bt[i] -= g.b_period;
          if (bt[i] >= g.b_period) bt[i] %= g.b_period;
        }
      }
      TileA::loadf(ra[q], ra_rsrc, offa, TileA::stage_offset(k0, g.lda), kend - k0, tid);
      TileB::loadf(rb[q], rb_rsrc, offb, TileB::stage_offset(k0, g.ldb), kend - k0, tid, kill);
    };
    load_stage(0, kbeg);
    load_stage(1, kbeg + FBK);
    TileA::store(ra[0], smem, tid);
    TileB::store(rb[0], smem + OPB, tid);
    add_colsum(ra[0]);
    load_stage(0, kbeg + 2 * FBK);
    lr_lds_barrier();
    for (int kq = kbeg; kq < kend; kq += 2 * FBK) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {          // (fully unrolled: buffers and register sets are static)
        // stage kq + q * FBK: contract buffer q; registers of the NEXT stage (set q ^ 1) -> buffer q ^ 1
        contract(smem + q * 2 * OPB);
        TileA::store(ra[q ^ 1], smem + (q ^ 1) * 2 * OPB, tid);
        TileB::store(rb[q ^ 1], smem + (q ^ 1) * 2 * OPB + OPB, tid);
        add_colsum(ra[q ^ 1]);
        load_stage(q ^ 1, kq + (q + 3) * FBK);
        lr_lds_barrier();
      }
    }
  } else {
    // ---- guarded: leading dimensions or extents that are not multiples of four — element loads under predicates, one
    // stage buffer, one register set (hipcc closes every predicated load with a wait: a memory round trip each) ---------
    unsigned char* As = smem;
    unsigned char* Bs = smem + OPB;
    const int how_a = g.vec_a == 2 ? 1 : g.vec_a, how_b = g.vec_b == 2 ? 1 : g.vec_b;
    float4 ra[4], rb[4];
    TileA::load(ra, g.A, g.lda, m0, g.M, kbeg, kend, how_a != 0, tid);
    TileB::load(rb, g.B, g.ldb, n0, g.N, kbeg, kend, how_b != 0, tid, g.b_shift, TB ? 0 : g.b_period);
    for (int k0 = kbeg; k0 < kend; k0 += FBK) {
      lr_lds_barrier();                       // every wave has read the previous stage
      TileA::store(ra, As, tid);
      TileB::store(rb, Bs, tid);
      add_colsum(ra);
      lr_lds_barrier();
      if (k0 + FBK < kend) {
        TileA::load(ra, g.A, g.lda, m0, g.M, k0 + FBK, kend, how_a != 0, tid);
        TileB::load(rb, g.B, g.ldb, n0, g.N, k0 + FBK, kend, how_b != 0, tid, g.b_shift, TB ? 0 : g.b_period);
      }
      contract(smem);
    }
    lr_lds_barrier();
  }

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
      for (int e = 0; e < 4; ++e) {
        if (c + e >= g.M) continue;
        if (g.slabs) g.slabs[(size_t)g.splits * g.M * g.N + (size_t)zs * g.M + c + e] = v[e];   // this K range's share
        else g.colsum[c + e] = g.beta != 0.f ? g.beta * g.colsum[c + e] + v[e] : v[e];
      }
    }
  }

  // ---- epilogue, staged through LDS 64 rows at a time.  The accumulators leave the registers with 64 LDS stores per
  // wave half (C/D layout of a 32 x 32 tile: column = lane & 31, row of register r = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
  // and everything else is a LOOP over row-contiguous quads: 16-byte loads of bias / addend / mask / C and 16-byte
  // stores, a few dozen instructions executed eight times.  The first cut applied the epilogue per accumulator register,
  // fully unrolled: ~20 KB of straight-line code that every workgroup ran once — fetched cold, it was most of a
  // launch's 19 us fixed cost at every K.
  float* Cs = reinterpret_cast<float*>(smem);
  const int lr = lane & 31, lk = lane >> 5;
  const bool wrap = g.addend && g.add_period < g.M;    // a periodic table (else a residual: the row itself)
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            Cs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * C_LDS + wn * 64 + j * 32 + lr] = acc[i][j][r];
    }
    lr_lds_barrier();
#pragma unroll 2
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + 256 * u, rl = idx >> 5, c4 = 4 * (idx & 31);
      const int row = m0 + 64 * half + rl, col = n0 + c4;
      if (row >= g.M || col >= g.N) continue;
      const float4 a4 = *reinterpret_cast<const float4*>(Cs + rl * C_LDS + c4);
      float v[4] = {a4.x, a4.y, a4.z, a4.w};
      const int nv = min(4, g.N - col);                 // (4 when the row-contiguous accesses are whole quads: vec_c)
      if (g.slabs) {
        float* sp = g.slabs + ((int64_t)zs * g.M + row) * g.N + col;
        if (g.vec_c) *reinterpret_cast<float4*>(sp) = a4;
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nv) sp[e] = v[e];
        }
        continue;
      }
      float bv[4] = {0.f, 0.f, 0.f, 0.f}, add[4] = {0.f, 0.f, 0.f, 0.f}, mk[4] = {1.f, 1.f, 1.f, 1.f}, old[4] = {0.f, 0.f, 0.f, 0.f};
      const int arow = wrap ? row % g.add_period : row;
      if (g.vec_c) {
        if (g.bias) { const float4 q = *reinterpret_cast<const float4*>(g.bias + col); bv[0] = q.x; bv[1] = q.y; bv[2] = q.z; bv[3] = q.w; }
        if (g.addend) { const float4 q = *reinterpret_cast<const float4*>(g.addend + (int64_t)arow * g.ldadd + col); add[0] = q.x; add[1] = q.y; add[2] = q.z; add[3] = q.w; }
        if (g.mask) { const float4 q = *reinterpret_cast<const float4*>(g.mask + (int64_t)row * g.ldmask + col); mk[0] = q.x; mk[1] = q.y; mk[2] = q.z; mk[3] = q.w; }
        if (g.beta != 0.f) { const float4 q = *reinterpret_cast<const float4*>(g.C + (int64_t)row * g.ldc + col); old[0] = q.x; old[1] = q.y; old[2] = q.z; old[3] = q.w; }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (e >= nv) continue;
          if (g.bias) bv[e] = g.bias[col + e];
          if (g.addend) add[e] = g.addend[(int64_t)arow * g.ldadd + col + e];
          if (g.mask) mk[e] = g.mask[(int64_t)row * g.ldmask + col + e];
          if (g.beta != 0.f && !(g.flags & LR_FGEMM_C_BF16)) old[e] = g.C[(int64_t)row * g.ldc + col + e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float out = g.alpha * v[e] + bv[e] + add[e];
        if (g.flags & LR_FGEMM_RELU) out = fmaxf(out, 0.f);
        out = mk[e] > 0.f ? out : 0.f;
        v[e] = out + g.beta * old[e];
      }
      if (g.flags & LR_FGEMM_C_BF16) {
        bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (int64_t)row * g.ldc + col;
        const bf16x2 p0 = __builtin_convertvector((f32x2){v[0], v[1]}, bf16x2), p1 = __builtin_convertvector((f32x2){v[2], v[3]}, bf16x2);
        if (g.vec_c) *reinterpret_cast<uint2*>(cp) = make_uint2(__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1));
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nv) { const __bf16 hb = (__bf16)v[e]; cp[e] = __builtin_bit_cast(bf16_t, hb); }
        }
      } else {
        float* cp = g.C + (int64_t)row * g.ldc + col;
        if (g.vec_c) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nv) cp[e] = v[e];
        }
      }
    }
    lr_lds_barrier();
  }
}

// split-K combine of every split job of a launch (grid (blocks, jobs); fixed order over the slabs) + the epilogue, and
// the jobs' column sums (TN with colsum): the K ranges' shares in fixed order
__global__ __launch_bounds__(256) void fgemm_reduce_kernel(const FArgs args) {
  const FJob& g = args.j[blockIdx.y];
  if (g.splits <= 1) return;
  const int64_t total = (int64_t)g.M * g.N;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g.colsum) {
    const float* part = g.slabs + (size_t)g.splits * total;
    for (int64_t i = i0; i < g.M; i += stride) {
      float s = 0.f;
      for (int z = 0; z < g.splits; ++z) s += part[(int64_t)z * g.M + i];
      g.colsum[i] = g.beta != 0.f ? g.beta * g.colsum[i] + s : s;
    }
  }
  for (int64_t i = i0; i < total; i += stride) {
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
// [splits][M][N] partial products, then [splits][M] partial column sums (TN jobs with colsum)
size_t lr_fgemm_slab_floats_impl(int M, int N, int splits) { return splits > 1 ? (size_t)splits * ((size_t)M * N + M) : 0; }

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
    LR_CHECK_ARG(!s.colsum || form == LR_FGEMM_TN);
    LR_CHECK_ARG(s.b_period >= 0 && (s.b_period == 0 || (form != LR_FGEMM_NT && !b_bf16)));
    LR_CHECK_ARG(s.splits <= 1 || s.slabs);
    LR_CHECK_ARG(!(s.flags & LR_FGEMM_C_BF16) || s.beta == 0.f);
    g.A = s.A; g.B = s.B; g.C = (float*)s.C;
    g.b_shift = s.b_period > 0 ? s.b_shift : 0; g.b_period = s.b_period;
    // (row k of B comes from row k + b_shift: the base moves, the rows whose neighbour does not exist are never read)
    if (g.b_period > 0) g.B = reinterpret_cast<const float*>(s.B) + (int64_t)g.b_shift * s.ldb;
    g.bias = s.bias; g.addend = s.addend; g.mask = s.mask; g.colsum = s.colsum;
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
    // ... and the matrix stays below 2^31 bytes (the buffer loads' offsets)
    const int64_t rows_a = form == LR_FGEMM_TN ? s.K : s.M, rows_b = form == LR_FGEMM_NT ? s.N : s.K;
    if (g.vec_a && (form == LR_FGEMM_TN ? s.M : s.K) % 4 == 0 && rows_a * s.lda * (int64_t)ea < (1ll << 31)) g.vec_a = 2;
    if (g.vec_b && (form == LR_FGEMM_NT ? s.K : s.N) % 4 == 0 && rows_b * s.ldb * (int64_t)eb < (1ll << 31)) g.vec_b = 2;
    const size_t ec = (s.flags & LR_FGEMM_C_BF16) ? 2 : 4;
    g.vec_c = s.N % 4 == 0 && s.ldc % 4 == 0 && aligned_to(s.C, 4 * ec) && (!s.bias || aligned_to(s.bias, 16)) &&
              (!s.addend || (s.ldadd % 4 == 0 && aligned_to(s.addend, 16))) &&
              (!s.mask || (s.ldmask % 4 == 0 && aligned_to(s.mask, 16))) && (splits <= 1 || aligned_to(s.slabs, 16));
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
  int64_t biggest = 0;
  for (int q = 0; q < njobs; ++q) {
    FJob& g = a.j[q];
    if (g.splits <= 1) continue;
    g.slabs = jobs[q].slabs;      // (the combine reads slabs, bias, colsum of the job as given)
    if ((int64_t)g.M * g.N > biggest) biggest = (int64_t)g.M * g.N;
  }
  if (biggest == 0) return LR_OK;
  int blocks = (int)((biggest + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  LR_LAUNCH(fgemm_reduce_kernel, dim3(blocks, njobs), dim3(256), 0, stream, a);
  return lr_launch_status();
}

extern "C" int lr_fgemm(int prec, int form, int a_bf16, int b_bf16, const lr_fgemm_job* jobs, int njobs,
                        lr_stream_t stream) {
  return lr_fgemm_launch(prec, form, a_bf16, b_bf16, jobs, njobs, (hipStream_t)stream);
}
extern "C" int lr_fgemm_splits(int M, int N, int K) { return lr_fgemm_want_splits(M, N, K); }
extern "C" long long lr_fgemm_slab_floats(int M, int N, int splits) { return (long long)lr_fgemm_slab_floats_impl(M, N, splits); }
