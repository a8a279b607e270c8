// lr_misc.hip — library entry points plus the bandwidth-trivial kernels either side of the
// encoder: batch collation (A1), the landmark step (A7), and the optimiser tail of the
// reference step (A5).
//
// Reference arithmetic replaced here:
//   src/data/data_loader.py:130-142   _pad (zero-pad ragged sequences to the batch max)
//   src/utils/data/face.py:76-90      _applyPadding (integer rectangle padding)
//   src/utils/data/face.py:164-175    getFace (translate x,y by the padded rect origin)
//   src/train/train_better_model.py:78-80  clip_grad_norm_ + Adam.step
#include "lr_common.h"
#include <hip/hip_ext.h>

extern "C" int lr_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* lr_status_string(int status) {
  switch (status) {
    case LR_OK: return "ok";
    case LR_ERR_INVALID_ARG: return "invalid argument";
    case LR_ERR_WORKSPACE: return "workspace too small";
    case LR_ERR_LAUNCH: return "kernel launch failed";
    case LR_ERR_UNSUPPORTED: return "unsupported configuration";
    case LR_ERR_NO_DEVICE: return "no gfx950 device";
    default: return "unknown status";
  }
}

extern "C" int lr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

namespace {

// One workgroup-row per (b,t): feat floats copied or zeroed; lanes along feat.
__global__ void collate_pad_kernel(const float* __restrict__ packed,
                                   const int64_t* __restrict__ offsets,
                                   const int32_t* __restrict__ lens, float* __restrict__ out,
                                   int B, int t_max, int feat) {
  const int64_t total = (int64_t)B * t_max * feat;
  const int64_t row_elems = (int64_t)t_max * feat;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / row_elems);
    const int64_t r = i - (int64_t)b * row_elems;
    const int t = (int)(r / feat);
    const int f = (int)(r - (int64_t)t * feat);
    float v = 0.f;
    if (t < lens[b]) v = packed[(offsets[b] + t) * feat + f];
    out[i] = v;
  }
}

__global__ void lmk_apply_padding_kernel(const int32_t* __restrict__ rin,
                                         const int32_t* __restrict__ dims,
                                         int32_t* __restrict__ rout, int n, float padding) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int img_h = dims[2 * i], img_w = dims[2 * i + 1];
  int left = rin[4 * i], right = rin[4 * i + 1], top = rin[4 * i + 2], bottom = rin[4 * i + 3];
  const int box_h = bottom - top, box_w = right - left;
  // Python: int(padding * box_w) — a double product truncated toward zero.
  const int pw = (int)((double)padding * (double)box_w);
  const int ph = (int)((double)padding * (double)box_h);
  left = max(0, left - pw);
  right = min(img_w, right + pw);
  top = max(0, top - ph);
  bottom = min(img_h, bottom + ph);
  rout[4 * i] = left;
  rout[4 * i + 1] = right;
  rout[4 * i + 2] = top;
  rout[4 * i + 3] = bottom;
}

__global__ void lmk_translate_kernel(const float* __restrict__ in, const int32_t* __restrict__ rects,
                                     float* __restrict__ out, int64_t total, int per_frame) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t frame = i / per_frame;
    const int comp = (int)((i - frame * per_frame) % 3);
    float v = in[i];
    if (comp == 0) v -= (float)rects[4 * frame];          // left
    else if (comp == 1) v -= (float)rects[4 * frame + 2];  // top
    out[i] = v;
  }
}

__global__ void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) acc += x[i] * x[i];
  acc = lr_wave_sum(acc);
  __shared__ float part[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += part[w];
    atomicAdd(out, s);
  }
}

// One thread: advance the device-side step counter (unless the batch is skipped), derive the
// bias corrections and the clip coefficient.  Keeping the counter on the device makes the
// whole optimiser step replayable from a hipGraph (host-computed corrections would be frozen).
//   st[0] = active (0/1), st[1] = lr / (1 - beta1^t), st[2] = 1 / sqrt(1 - beta2^t),
//   st[3] = gradient scale (grad_scale x clip coefficient)
__global__ void adam_prepare_kernel(int32_t* __restrict__ step_count,
                                    const int32_t* __restrict__ skip,
                                    const float* __restrict__ sumsq, float max_norm,
                                    float grad_scale, float lr, float beta1, float beta2,
                                    float* __restrict__ st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const bool active = !(skip && skip[0] != 0);
  int t = step_count[0];
  if (active) step_count[0] = ++t;
  if (t < 1) t = 1;
  const float bc1 = 1.f - powf(beta1, (float)t);
  const float bc2 = 1.f - powf(beta2, (float)t);
  float scale = grad_scale;
  if (max_norm > 0.f && sumsq) {
    // clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    const float norm = sqrtf(sumsq[0]) * fabsf(grad_scale);
    const float coef = max_norm / (norm + 1e-6f);
    if (coef < 1.f) scale *= coef;
  }
  st[0] = active ? 1.f : 0.f;
  st[1] = lr / bc1;
  st[2] = 1.f / sqrtf(bc2);
  st[3] = scale;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, int64_t n,
                            const float* __restrict__ st, float beta1, float beta2, float eps) {
  if (st[0] == 0.f) return;  // skipped batch: parameters and moments untouched
  const float step = st[1], inv_sqrt_bc2 = st[2], scale = st[3];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * scale;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] -= step * (mi / denom);
  }
}

// 256 threads = 4 row lanes x 64 columns; blockIdx.y = row slice.
__global__ void colsum_partial_kernel(const float* __restrict__ x, int ld, int rows, int ncol,
                                      float* __restrict__ partial) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int per = (rows + LR_COLSUM_SPLITS - 1) / LR_COLSUM_SPLITS;
  const int r0 = blockIdx.y * per;
  const int r1 = min(rows, r0 + per);
  float s = 0.f;
  if (col < ncol)
    for (int r = r0 + rl; r < r1; r += 4) s += x[(int64_t)r * ld + col];
  part[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && col < ncol)
    partial[(int64_t)blockIdx.y * ncol + col] = part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl];
}

inline int grid_for(int64_t n, int block) {
  int64_t g = (n + block - 1) / block;
  if (g > 2048) g = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int lr_collate_pad_f32(const float* packed, const int64_t* offsets,
                                  const int32_t* lens, float* out, int B, int t_max, int feat,
                                  lr_stream_t stream) {
  LR_CHECK_ARG(packed && offsets && lens && out);
  LR_CHECK_ARG(B > 0 && t_max > 0 && feat > 0);
  const int64_t total = (int64_t)B * t_max * feat;
  LR_LAUNCH(collate_pad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, packed, offsets, lens, out, B, t_max, feat);
  return lr_launch_status();
}

extern "C" int lr_lmk_apply_padding(const int32_t* rects_in, const int32_t* dims,
                                    int32_t* rects_out, int n, float padding,
                                    lr_stream_t stream) {
  LR_CHECK_ARG(rects_in && dims && rects_out && n > 0);
  LR_LAUNCH(lmk_apply_padding_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, rects_in, dims, rects_out, n, padding);
  return lr_launch_status();
}

extern "C" int lr_lmk_translate(const float* lmk_in, const int32_t* rects, float* lmk_out, int n,
                                int npts, lr_stream_t stream) {
  LR_CHECK_ARG(lmk_in && rects && lmk_out && n > 0 && npts > 0);
  const int64_t total = (int64_t)n * npts * 3;
  LR_LAUNCH(lmk_translate_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, lmk_in, rects, lmk_out, total, npts * 3);
  return lr_launch_status();
}

extern "C" int lr_sumsq(const float* x, int64_t n, float* out, lr_stream_t stream) {
  LR_CHECK_ARG(x && out && n >= 0);
  if (n == 0) return LR_OK;
  LR_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  int g = grid_for((n + 3) / 4, 256);
  if (g > 256) g = 256;  // one atomic per workgroup
  LR_LAUNCH(sumsq_kernel, dim3(g), dim3(256), 0, stream, x, n, out);
  return lr_launch_status();
}

extern "C" int lr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                            int64_t n, const float* sumsq, float max_norm, float grad_scale,
                            float lr, float beta1, float beta2, float eps, int32_t* step_count,
                            const int32_t* skip, float* scratch, lr_stream_t stream) {
  LR_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step_count && scratch && n >= 0);
  LR_LAUNCH(adam_prepare_kernel, dim3(1), dim3(64), 0, stream, step_count, skip, sumsq, max_norm,
            grad_scale, lr, beta1, beta2, scratch);
  int st = lr_launch_status();
  if (st != LR_OK || n == 0) return st;
  LR_LAUNCH(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, param, grad, exp_avg,
            exp_avg_sq, n, (const float*)scratch, beta1, beta2, eps);
  return lr_launch_status();
}

int lr_colsum_partial(const float* x, int ld, int rows, int ncol, float* partial, hipStream_t stream) {
  LR_LAUNCH(colsum_partial_kernel, dim3((ncol + 63) / 64, LR_COLSUM_SPLITS), dim3(256), 0, stream, x,
            ld, rows, ncol, partial);
  return lr_launch_status();
}

// ---- A9 (BUILD-DEFINED): mouth crop + bilinear resize ------------------------------------------------
// The reference defines `_mouth = slice(48, 68)` (src/utils/data/face.py:21) and never uses it; it
// has no pixel lip crop.  Specification of this build: per frame, the bounding box
// of landmarks [lo,hi) in image coordinates -> centre (cx,cy), side = max(w,h)*(1+2*margin), at least
// 2 px -> square window resampled to S x S with bilinear interpolation, half-pixel centres, source
// coordinates clamped to the image (edge replication), result rounded to nearest uint8.
__global__ void lip_crop_kernel(const unsigned char* __restrict__ frames, const float* __restrict__ lmk,
                                unsigned char* __restrict__ out, int H, int W, int S, int npts, int lo,
                                int hi, float margin) {
  __shared__ float box[4];
  const int n = blockIdx.x;
  const float* L = lmk + (int64_t)n * npts * 3;
  if (threadIdx.x == 0) {
    float x0 = L[lo * 3], x1 = x0, y0 = L[lo * 3 + 1], y1 = y0;
    for (int p = lo + 1; p < hi; ++p) {
      const float x = L[p * 3], y = L[p * 3 + 1];
      x0 = fminf(x0, x); x1 = fmaxf(x1, x); y0 = fminf(y0, y); y1 = fmaxf(y1, y);
    }
    float side = fmaxf(x1 - x0, y1 - y0) * (1.f + 2.f * margin);
    side = fmaxf(side, 2.f);
    box[0] = 0.5f * (x0 + x1) - 0.5f * side;   // left
    box[1] = 0.5f * (y0 + y1) - 0.5f * side;   // top
    box[2] = side / (float)S;                  // source pixels per output pixel
  }
  __syncthreads();
  const float left = box[0], top = box[1], scale = box[2];
  const unsigned char* F = frames + (int64_t)n * 3 * H * W;
  unsigned char* O = out + (int64_t)n * 3 * S * S;
  for (int i = threadIdx.x; i < 3 * S * S; i += blockDim.x) {
    const int c = i / (S * S), r = i - c * S * S;
    const int oy = r / S, ox = r - oy * S;
    float sx = left + ((float)ox + 0.5f) * scale - 0.5f;
    float sy = top + ((float)oy + 0.5f) * scale - 0.5f;
    sx = fminf(fmaxf(sx, 0.f), (float)(W - 1));
    sy = fminf(fmaxf(sy, 0.f), (float)(H - 1));
    const int ix = (int)floorf(sx), iy = (int)floorf(sy);
    const int ix1 = min(ix + 1, W - 1), iy1 = min(iy + 1, H - 1);
    const float fx = sx - (float)ix, fy = sy - (float)iy;
    const unsigned char* P = F + (int64_t)c * H * W;
    const float a = (float)P[iy * W + ix], b = (float)P[iy * W + ix1];
    const float d = (float)P[iy1 * W + ix], e = (float)P[iy1 * W + ix1];
    const float top_v = a + (b - a) * fx, bot_v = d + (e - d) * fx;
    const float v = top_v + (bot_v - top_v) * fy;
    O[i] = (unsigned char)fminf(fmaxf(floorf(v + 0.5f), 0.f), 255.f);
  }
}

extern "C" int lr_lip_crop_u8(const void* frames, const float* lmk, void* out, int n, int H, int W, int S,
                              int npts, int lo, int hi, float margin, lr_stream_t stream) {
  LR_CHECK_ARG(frames && lmk && out && n > 0 && H > 0 && W > 0 && S > 0);
  LR_CHECK_ARG(npts > 0 && lo >= 0 && hi > lo && hi <= npts && margin >= 0.f);
  LR_LAUNCH(lip_crop_kernel, dim3(n), dim3(256), 0, stream, (const unsigned char*)frames, lmk,
            (unsigned char*)out, H, W, S, npts, lo, hi, margin);
  return lr_launch_status();
}

// ---- instrumentation ----------------------------------------------------------------------------
namespace {
constexpr int kProfRing = 1024;
struct ProfSlot {
  hipEvent_t start[kProfRing], stop[kProfRing];
  int count;
  bool ready;
};
ProfSlot g_prof[LR_PROF_SLOTS];
bool g_prof_on = false;
}  // namespace

bool lr_prof_next(int slot, hipEvent_t* start, hipEvent_t* stop) {
  if (!g_prof_on || slot < 0 || slot >= LR_PROF_SLOTS) return false;
  ProfSlot& p = g_prof[slot];
  if (!p.ready || p.count >= kProfRing) return false;
  *start = p.start[p.count];
  *stop = p.stop[p.count];
  ++p.count;
  return true;
}

extern "C" int lr_profile_enable(int on) {
  if (on && !g_prof_on) {
    for (int w = 0; w < LR_PROF_SLOTS; ++w) {
      ProfSlot& p = g_prof[w];
      if (!p.ready) {
        for (int i = 0; i < kProfRing; ++i) {
          if (hipEventCreate(&p.start[i]) != hipSuccess || hipEventCreate(&p.stop[i]) != hipSuccess) {
            (void)hipGetLastError();
            return LR_ERR_NO_DEVICE;
          }
        }
        p.ready = true;
      }
      p.count = 0;
    }
  }
  g_prof_on = on != 0;
  return LR_OK;
}

extern "C" int lr_profile_read(int which, float* total_ms_host, int* samples_host) {
  LR_CHECK_ARG(which >= 0 && which < LR_PROF_SLOTS && total_ms_host && samples_host);
  ProfSlot& p = g_prof[which];
  float total = 0.f;
  int n = 0;
  for (int i = 0; i < p.count; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(p.stop[i]) == hipSuccess &&
        hipEventElapsedTime(&ms, p.start[i], p.stop[i]) == hipSuccess) {
      total += ms;
      ++n;
    }
  }
  (void)hipGetLastError();
  p.count = 0;
  *total_ms_host = total;
  *samples_host = n;
  return LR_OK;
}
