/*
 * lr_oracle.c — plain-C restatement of the video->characters hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (lipreading_amd/) never does.  It exists so that the parity claim does not rest
 * on torch's CPU kernels alone: the cell equations, the packed-sequence masking, the masked
 * log-softmax, the CTC recursion and the greedy collapse are written out here from the
 * reference's definitions, in double precision where it is cheap, and pinned against the
 * vectors captured from the reference itself (tests/golden, tests/test_c_oracle.py).
 *
 * Each function cites the reference file:line it follows (paths under the reference root).
 * Scalar, single-threaded, no dependencies beyond libm.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

/* ------------------------------------------------------------------------------------------
 * A3: one (bi)directional recurrent layer with packed-sequence semantics.
 * src/models/lipreader/better_model.py:64-78 (sort -> pack -> nn.GRU/LSTM -> pad_packed) —
 * restated without the sort: sample b runs for lens[b] steps; the reverse direction starts at
 * its last valid frame; outputs past lens[b] are zero; the final state is the state after the
 * sample's own last step.  Gate order and equations: torch.nn.GRU (r,z,n) / torch.nn.LSTM
 * (i,f,g,o) as documented.
 *   mode 0 = GRU, 1 = LSTM;  x [B,T,I];  w_ih [D][G*H,I];  w_hh [D][G*H,H];  b_ih,b_hh [D][G*H]
 *   y [B,T,D*H];  h_n [D,B,H];  c_n [D,B,H] (LSTM)
 * ------------------------------------------------------------------------------------------ */
int oracle_rnn_layer(int mode, const float* x, const int32_t* lens, const float* w_ih,
                     const float* w_hh, const float* b_ih, const float* b_hh, float* y, float* h_n,
                     float* c_n, int B, int T, int I, int H, int D) {
  const int G = mode == 0 ? 3 : 4;
  double* h = (double*)malloc(sizeof(double) * H);
  double* c = (double*)malloc(sizeof(double) * H);
  double* gi = (double*)malloc(sizeof(double) * G * H);
  double* gh = (double*)malloc(sizeof(double) * G * H);
  double* hn = (double*)malloc(sizeof(double) * H);
  if (!h || !c || !gi || !gh || !hn) return -1;
  memset(y, 0, sizeof(float) * (size_t)B * T * D * H);
  for (int d = 0; d < D; ++d) {
    const float* Wi = w_ih + (size_t)d * G * H * I;
    const float* Wh = w_hh + (size_t)d * G * H * H;
    const float* bi = b_ih + (size_t)d * G * H;
    const float* bh = b_hh + (size_t)d * G * H;
    for (int b = 0; b < B; ++b) {
      const int n = lens[b];
      for (int j = 0; j < H; ++j) h[j] = c[j] = 0.0;
      for (int s = 0; s < n; ++s) {
        const int t = d == 0 ? s : n - 1 - s;
        const float* xt = x + ((size_t)b * T + t) * I;
        for (int r = 0; r < G * H; ++r) {
          double a = bi[r], q = bh[r];
          for (int k = 0; k < I; ++k) a += (double)Wi[(size_t)r * I + k] * xt[k];
          for (int k = 0; k < H; ++k) q += (double)Wh[(size_t)r * H + k] * h[k];
          gi[r] = a;
          gh[r] = q;
        }
        for (int j = 0; j < H; ++j) {
          if (mode == 0) {
            const double r = sigmoid_d(gi[j] + gh[j]);
            const double z = sigmoid_d(gi[H + j] + gh[H + j]);
            const double nn = tanh(gi[2 * H + j] + r * gh[2 * H + j]);
            hn[j] = (1.0 - z) * nn + z * h[j];
          } else {
            const double ig = sigmoid_d(gi[j] + gh[j]);
            const double fg = sigmoid_d(gi[H + j] + gh[H + j]);
            const double gg = tanh(gi[2 * H + j] + gh[2 * H + j]);
            const double og = sigmoid_d(gi[3 * H + j] + gh[3 * H + j]);
            c[j] = fg * c[j] + ig * gg;
            hn[j] = og * tanh(c[j]);
          }
        }
        for (int j = 0; j < H; ++j) {
          h[j] = hn[j];
          y[((size_t)b * T + t) * D * H + (size_t)d * H + j] = (float)h[j];
        }
      }
      for (int j = 0; j < H; ++j) {
        h_n[((size_t)d * B + b) * H + j] = (float)h[j];
        if (mode == 1 && c_n) c_n[((size_t)d * B + b) * H + j] = (float)c[j];
      }
    }
  }
  free(h); free(c); free(gi); free(gh); free(hn);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * A3 tail: output_proj + masked_log_softmax — better_model.py:92-93.
 * log_softmax(hidden @ W^T + bias + log(mask + 1e-45)); the mask term is evaluated in fp32 as
 * the reference does (1e-45 rounds to the smallest fp32 subnormal; its log is -103.2789).
 * ------------------------------------------------------------------------------------------ */
int oracle_proj_logsoftmax(const float* hidden, const float* W, const float* bias,
                           const float* mask, float* log_probs, int R, int K, int C) {
  double* z = (double*)malloc(sizeof(double) * C);
  if (!z) return -1;
  for (int r = 0; r < R; ++r) {
    double m = -INFINITY;
    for (int c = 0; c < C; ++c) {
      double a = bias[c];
      for (int k = 0; k < K; ++k) a += (double)W[(size_t)c * K + k] * hidden[(size_t)r * K + k];
      const float mterm = logf(mask[c] + 1e-45f);
      z[c] = a + (double)mterm;
      if (z[c] > m) m = z[c];
    }
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += exp(z[c] - m);
    const double lse = m + log(s);
    for (int c = 0; c < C; ++c) log_probs[(size_t)r * C + c] = (float)(z[c] - lse);
  }
  free(z);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * A4: CTC negative log-likelihood and gradient for one batch, blank = 0 — the quantity
 * F.ctc_loss computes at src/train/ctc_loss.py:85 (Graves et al. 2006, eq. 6-8, 10-11, 16),
 * with torch's gradient convention  grad = exp(lp) - exp(log sum alpha beta + nll - lp).
 *   lp [B,T,C]; labels [B,label_stride] already shifted by +1 (ctc_loss.py:80);
 *   nll [B] (+inf when infeasible); grad [B,T,C] or NULL (unweighted, zero past frame_lens
 *   and zero for infeasible samples).
 * ------------------------------------------------------------------------------------------ */
static double lse2_d(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const double m = a > b ? a : b;
  return m + log(exp(a - m) + exp(b - m));
}

int oracle_ctc(const float* lp, const int32_t* labels, int label_stride, const int32_t* frame_lens,
               const int32_t* label_lens, float* nll, float* grad, int B, int T, int C) {
  for (int b = 0; b < B; ++b) {
    const int L = label_lens[b], Tb = frame_lens[b], S = 2 * L + 1;
    const float* lpb = lp + (size_t)b * T * C;
    const int32_t* lab = labels + (size_t)b * label_stride;
    double* al = (double*)malloc(sizeof(double) * (size_t)Tb * S);
    double* be = (double*)malloc(sizeof(double) * (size_t)Tb * S);
    if (!al || !be) return -1;
#define CLS(s) (((s) & 1) ? lab[(s) >> 1] : 0)
    for (int s = 0; s < S; ++s) al[s] = s < 2 ? (double)lpb[CLS(s)] : -INFINITY;
    for (int t = 1; t < Tb; ++t)
      for (int s = 0; s < S; ++s) {
        double a = al[(size_t)(t - 1) * S + s];
        if (s >= 1) a = lse2_d(a, al[(size_t)(t - 1) * S + s - 1]);
        if ((s & 1) && s >= 3 && CLS(s) != CLS(s - 2)) a = lse2_d(a, al[(size_t)(t - 1) * S + s - 2]);
        al[(size_t)t * S + s] = a + (double)lpb[(size_t)t * C + CLS(s)];
      }
    double ll = al[(size_t)(Tb - 1) * S + S - 1];
    if (S > 1) ll = lse2_d(ll, al[(size_t)(Tb - 1) * S + S - 2]);
    nll[b] = (float)(-ll);
    if (grad) {
      float* gb = grad + (size_t)b * T * C;
      memset(gb, 0, sizeof(float) * (size_t)T * C);
      if (ll != -INFINITY) {
        for (int s = 0; s < S; ++s)
          be[(size_t)(Tb - 1) * S + s] = s >= S - 2 ? (double)lpb[(size_t)(Tb - 1) * C + CLS(s)] : -INFINITY;
        for (int t = Tb - 2; t >= 0; --t)
          for (int s = 0; s < S; ++s) {
            double a = be[(size_t)(t + 1) * S + s];
            if (s + 1 < S) a = lse2_d(a, be[(size_t)(t + 1) * S + s + 1]);
            if ((s & 1) && s + 2 < S && CLS(s) != CLS(s + 2)) a = lse2_d(a, be[(size_t)(t + 1) * S + s + 2]);
            be[(size_t)t * S + s] = a + (double)lpb[(size_t)t * C + CLS(s)];
          }
        double* acc = (double*)malloc(sizeof(double) * C);
        for (int t = 0; t < Tb; ++t) {
          for (int c = 0; c < C; ++c) acc[c] = -INFINITY;
          for (int s = 0; s < S; ++s)
            acc[CLS(s)] = lse2_d(acc[CLS(s)], al[(size_t)t * S + s] + be[(size_t)t * S + s]);
          for (int c = 0; c < C; ++c) {
            const double l = lpb[(size_t)t * C + c];
            gb[(size_t)t * C + c] = (float)(exp(l) - exp(acc[c] - ll - l));
          }
        }
        free(acc);
      }
    }
#undef CLS
    free(al);
    free(be);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * A4: the reference's batch reduction — src/train/ctc_loss.py:46-114 — on per-sample nll.
 * Returns 0 and writes *loss, or returns 1 where the reference returns None.
 * weight[b] = d loss / d nll[b].
 * ------------------------------------------------------------------------------------------ */
int oracle_ctc_reduce(const float* nll, const int32_t* frame_lens, const int32_t* label_lens,
                      int mean, float* loss, float* weight, int B) {
  int* kept = (int*)malloc(sizeof(int) * (B > 0 ? B : 1));
  int n = 0;
  for (int i = 0; i < B; ++i) {
    weight[i] = 0.f;
    if (label_lens[i] <= 256) kept[n++] = i; /* :46 */
  }
  *loss = 0.f;
  if (n == 0) { free(kept); return 1; }
  float total = 0.f, count = 0.f;
  int any = 0, prev = 0, cur_len = n;
  for (int k = 1; k <= n; ++k) {
    if (k < n && frame_lens[kept[k]] == frame_lens[kept[k - 1]]) continue; /* change points :64 */
    const int cp = k;
    int mb = cur_len; /* len(frame_lens) read BEFORE the slice, :74 */
    cur_len = cp - prev;
    int m = 0, has_inf = 0;
    for (int q = prev; q < cp; ++q) {
      if (isinf(nll[kept[q]])) has_inf = 1; else ++m;
    }
    if (has_inf) {
      if (m == 0) continue; /* :92 — prev_change_point not advanced */
      cur_len = mb = m;     /* :96-101 */
    }
    float run = 0.f;
    for (int q = prev; q < cp; ++q) {
      const int i = kept[q];
      if (isinf(nll[i])) continue;
      if (mean) {
        const float l = (float)(label_lens[i] < 1 ? 1 : label_lens[i]);
        run += nll[i] / l;
        weight[i] = (float)mb / ((float)m * l);
      } else {
        run += nll[i];
        weight[i] = 1.f;
      }
    }
    if (mean) { run = run / (float)m * (float)mb; count += (float)mb; } /* :103-105 */
    total += run;
    any = 1;
    prev = cp; /* :107 */
  }
  free(kept);
  if (!any || total == 0.f) { /* :110-112 */
    for (int i = 0; i < B; ++i) weight[i] = 0.f;
    return 1;
  }
  if (mean) {
    for (int i = 0; i < B; ++i) weight[i] /= count;
    *loss = total / count;
  } else {
    *loss = total;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * A6: greedy decode — src/models/lipreader/decoder.py:165-197: argmax (first maximum), drop
 * blanks, drop a frame equal to the previous frame's argmax.  out_ids/out_off [B,T], out_lens [B].
 * ------------------------------------------------------------------------------------------ */
int oracle_greedy(const float* probs, const int32_t* sizes, int32_t* out_ids, int32_t* out_off,
                  int32_t* out_lens, int B, int T, int C, int blank) {
  for (int b = 0; b < B; ++b) {
    const int n = sizes ? sizes[b] : T;
    int prev = -1, cnt = 0;
    for (int t = 0; t < T; ++t) out_ids[(size_t)b * T + t] = out_off[(size_t)b * T + t] = -1;
    for (int t = 0; t < n; ++t) {
      const float* p = probs + ((size_t)b * T + t) * C;
      int best = 0;
      for (int c = 1; c < C; ++c)
        if (p[c] > p[best]) best = c;
      if (best != blank && !(t > 0 && best == prev)) {
        out_ids[(size_t)b * T + cnt] = best;
        out_off[(size_t)b * T + cnt] = t;
        ++cnt;
      }
      prev = best;
    }
    out_lens[b] = cnt;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * A7: landmark step — src/utils/data/face.py:76-90 (_applyPadding) and :164-175 (getFace).
 * ------------------------------------------------------------------------------------------ */
int oracle_apply_padding(const int32_t* rects_in, const int32_t* dims, int32_t* rects_out, int n,
                         double padding) {
  for (int i = 0; i < n; ++i) {
    const int img_h = dims[2 * i], img_w = dims[2 * i + 1];
    const int left = rects_in[4 * i], right = rects_in[4 * i + 1];
    const int top = rects_in[4 * i + 2], bottom = rects_in[4 * i + 3];
    const int pw = (int)(padding * (double)(right - left)); /* Python int(): toward zero */
    const int ph = (int)(padding * (double)(bottom - top));
    rects_out[4 * i] = left - pw > 0 ? left - pw : 0;
    rects_out[4 * i + 1] = right + pw < img_w ? right + pw : img_w;
    rects_out[4 * i + 2] = top - ph > 0 ? top - ph : 0;
    rects_out[4 * i + 3] = bottom + ph < img_h ? bottom + ph : img_h;
  }
  return 0;
}

int oracle_get_face(const float* lmk, const int32_t* rects, float* out, int n, int npts) {
  for (int i = 0; i < n; ++i)
    for (int p = 0; p < npts; ++p) {
      const size_t o = ((size_t)i * npts + p) * 3;
      out[o] = lmk[o] - (float)rects[4 * i];
      out[o + 1] = lmk[o + 1] - (float)rects[4 * i + 2];
      out[o + 2] = lmk[o + 2];
    }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * A1: collation — src/data/data_loader.py:130-142 (_pad).
 * ------------------------------------------------------------------------------------------ */
int oracle_collate_pad(const float* packed, const int64_t* offsets, const int32_t* lens, float* out,
                       int B, int t_max, int feat) {
  memset(out, 0, sizeof(float) * (size_t)B * t_max * feat);
  for (int b = 0; b < B; ++b)
    memcpy(out + (size_t)b * t_max * feat, packed + (size_t)offsets[b] * feat,
           sizeof(float) * (size_t)lens[b] * feat);
  return 0;
}
