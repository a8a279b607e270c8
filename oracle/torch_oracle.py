"""CPU oracle for the video->characters hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (lipreading_amd/) never does and fails loudly without its HIP library.

This is a restatement of the reference's algorithm with the same stock torch CPU ops the
reference itself bottoms out in (nn.GRU/nn.LSTM on a PackedSequence, nn.Linear, log_softmax,
F.ctc_loss), each function citing the reference file:line it follows (paths relative to the
reference root).  It is pinned by tests/golden/*.npz, which were produced by importing the
reference itself (tests/golden/make_golden.py):
  - ctc_* fixtures: reference src/train/ctc_loss.py imported as-is            -> CLEAN pin
  - greedy_* fixtures: reference decoder.py with an inert Levenshtein stub     -> CLEAN pin
  - enc_* / step_* fixtures: reference VideoEncoder with a stand-in for the absent, unpinned
    third-party `allennlp.nn.util` (masked_log_softmax / sort_batch_by_length) -> SHIMMED pin;
    parity at that boundary is otherwise unpinned.
  - lmk_* fixtures: reference face.py (_applyPadding, extractFace, getFace) imported with inert
    `dlib` / `src.models.face.prnet` stubs (neither is on the arithmetic path)   -> CLEAN pin
  - prn_* fixtures: the reference's PRN.process / get_landmarks bodies (prnet.py:112-170) run with the
    network replaced by a fixed position map; `skimage.transform.estimate_transform` (scikit-image
    0.14.1, absent) stood in by umeyama_similarity below                        -> SHIMMED pin
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# src/data/data_loader.py:29-35
PAD, BOS, EOS, UNK = '<PAD>', '<BOS>', '<EOS>', '<UNK>'
MARKERS = {PAD: 0, BOS: 1, EOS: 2, UNK: 3}
DEFAULT_LABELS = list(" !\"#$%&'()*+,-./0123456789:;<>?@[]abcdefghijklmnopqrstuvwxyz")


def default_char2idx():
  """build_vocab fallback path — src/data/data_loader.py:100-115."""
  c2i = dict(MARKERS)
  for ch in DEFAULT_LABELS:
    c2i[ch] = len(c2i)
  return c2i


# ------------------------------------------------------------------------------------------
# A1: collation — src/data/data_loader.py:117-152
# ------------------------------------------------------------------------------------------
def collate(frames_list, captions_list):
  """Zero-pad ragged (len_i,68,3) float and (cap_i,) int sequences to the batch maximum."""
  def pad(seqs, dtype):
    lens = torch.tensor([len(s) for s in seqs], dtype=torch.long)
    tail = tuple(np.asarray(seqs[0]).shape[1:])
    out = torch.zeros((len(seqs), int(lens.max())) + tail, dtype=dtype)
    for i, s in enumerate(seqs):
      out[i, :len(s)] = torch.as_tensor(np.asarray(s)).to(dtype)
    return out, lens
  f, fl = pad(frames_list, torch.float32)
  c, cl = pad(captions_list, torch.long)
  return f, fl, c, cl


# ------------------------------------------------------------------------------------------
# A7: landmark step — src/utils/data/face.py:76-90 and :164-175
# ------------------------------------------------------------------------------------------
def apply_padding(dims, rect, padding):
  """face.py:76-90; rect = (left, right, top, bottom), dims = (img_h, img_w[, c])."""
  img_h, img_w = dims[0], dims[1]
  left, right, top, bottom = rect
  box_h, box_w = bottom - top, right - left
  return (max(0, left - int(padding * box_w)), min(img_w, right + int(padding * box_w)),
          max(0, top - int(padding * box_h)), min(img_h, bottom + int(padding * box_h)))


def get_face(lmks, rect):
  """face.py:164-175: translate x by rect.left and y by rect.top."""
  out = np.array(lmks, copy=True)
  out[:, 0] -= rect[0]
  out[:, 1] -= rect[2]
  return out


def umeyama_similarity(src, dst):
  """`skimage.transform.estimate_transform('similarity', src, dst).params` (prnet.py:140) — the
  reference pins scikit-image==0.14.1 (requirements.txt:31), which is NOT under /root/reference
  and not installed here, so this is a restatement of the published algorithm that function
  implements: S. Umeyama, "Least-squares estimation of transformation parameters between two point
  patterns", IEEE TPAMI 13(4), 1991 (eq. 34-43) — demean, covariance A = dst_c^T src_c / n, SVD,
  reflection guard d, R = U diag(d) V^T, scale = sum(S*d) / var(src), t = mean_dst - scale R mean_src.
  PARITY UNPINNED at this boundary (no reference test vectors; the library is absent)."""
  src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
  num, dim = src.shape
  src_mean, dst_mean = src.mean(axis=0), dst.mean(axis=0)
  src_c, dst_c = src - src_mean, dst - dst_mean
  A = dst_c.T @ src_c / num
  d = np.ones(dim)
  if np.linalg.det(A) < 0:
    d[dim - 1] = -1
  T = np.eye(dim + 1)
  U, S, Vt = np.linalg.svd(A)
  rank = np.linalg.matrix_rank(A)
  if rank == 0:
    return np.nan * T
  if rank == dim - 1:
    if np.linalg.det(U) * np.linalg.det(Vt) > 0:
      T[:dim, :dim] = U @ Vt
    else:
      s = d[dim - 1]
      d[dim - 1] = -1
      T[:dim, :dim] = U @ np.diag(d) @ Vt
      d[dim - 1] = s
  else:
    T[:dim, :dim] = U @ np.diag(d) @ Vt
  scale = 1.0 / src_c.var(axis=0).sum() * (S @ d)
  T[:dim, dim] = dst_mean - scale * (T[:dim, :dim] @ src_mean.T)
  T[:dim, :dim] *= scale
  return T


def prn_crop_transform(rect, resolution=256):
  """prnet.py:112-119,136-140 with image_info = the (unpadded) face rect (left, right, top, bottom):
  crop square of side int(1.6 * mean box side) about the box centre -> similarity transform onto
  the resolution x resolution network input.  Returns (tform 3x3 float64, size)."""
  left, right, top, bottom = [int(v) for v in rect]
  old_size = (right - left + bottom - top) / 2
  center = np.array([right - (right - left) / 2.0, bottom - (bottom - top) / 2.0])
  size = int(old_size * 1.6)
  src_pts = np.array([[center[0] - size / 2, center[1] - size / 2], [center[0] - size / 2, center[1] + size / 2],
                      [center[0] + size / 2, center[1] - size / 2]])
  dst_pts = np.array([[0, 0], [0, resolution - 1], [resolution - 1, 0]])
  return umeyama_similarity(src_pts, dst_pts), size


def prn_restore(cropped_pos, tform):
  """prnet.py:150-156: position map of the crop -> image coordinates.  z /= tform[0,0];
  [x, y] = (tform^-1 [x, y, 1])[:2].  float64 like np.dot's result (the reference's numpy 1.15 keeps
  z at float32 precision — array/np.float64-scalar stays float32 there; numpy >= 2 promotes: the two
  differ by one float32 rounding of z, below what the dataview's float32 consumer can see)."""
  res = cropped_pos.shape[0]
  v = np.reshape(cropped_pos, [-1, 3]).T.astype(np.float64)
  z = v[2, :].copy() / tform[0, 0]
  v[2, :] = 1
  out = np.dot(np.linalg.inv(tform), v)
  out = np.vstack((out[:2, :], z))
  return np.reshape(out.T, [res, res, 3])


def prn_get_landmarks(pos, uv_kpt_ind):
  """prnet.py:162-170: kpt = pos[uv_kpt_ind[1], uv_kpt_ind[0], :] (68 gathers, row index second)."""
  return pos[uv_kpt_ind[1, :], uv_kpt_ind[0, :], :]


def landmark_step(cropped_pos, rect, dims, uv_kpt_ind, padding=0.3):
  """generate_dataview.py:58-64 minus the two networks: the UNPADDED dlib rect goes to the PRNet crop
  (:62), the PADDED one (extractFace(padding=0.3), :61) is what getFace translates by (:64).
  Returns (face landmarks (K,3) float64, padded rect)."""
  padded = apply_padding(dims, rect, padding)
  tform, _ = prn_crop_transform(rect, cropped_pos.shape[0])
  kpt = prn_get_landmarks(prn_restore(cropped_pos, tform), uv_kpt_ind)
  return get_face(kpt, padded), padded


# ------------------------------------------------------------------------------------------
# A3: encoder — src/models/lipreader/better_model.py:13-112
# ------------------------------------------------------------------------------------------
def masked_log_softmax(vector, mask, dim=-1):
  """The allennlp.nn.util function the reference calls at better_model.py:93, as published
  (allennlp 0.7-0.9): log_softmax(vector + log(mask + 1e-45)), mask broadcast from the left."""
  mask = mask.float()
  while mask.dim() < vector.dim():
    mask = mask.unsqueeze(1)
  return F.log_softmax(vector + (mask + 1e-45).log(), dim=dim)


class OracleVideoEncoder(nn.Module):
  """Same parameters and state_dict keys as the reference VideoEncoder (better_model.py:14-51)."""

  def __init__(self, frame_dim, hidden_size, rnn_type='LSTM', num_layers=1, bidirectional=True,
               rnn_dropout=0, enable_ctc=False, vocab_size=-1, char2idx=None):
    super().__init__()
    self.frame_dim, self.hidden_size = frame_dim, hidden_size
    self.rnn_type, self.num_layers, self.bidirectional = rnn_type, num_layers, bidirectional
    self.enable_ctc = enable_ctc
    self.rnn = getattr(nn, rnn_type)(frame_dim, hidden_size, num_layers=num_layers,
                                     bidirectional=bidirectional, batch_first=True,
                                     dropout=rnn_dropout)
    if enable_ctc:
      self.adj_vocab_size = vocab_size + 1                       # :38
      mask = torch.ones(self.adj_vocab_size)
      mask[char2idx[PAD] + 1] = 0                                # :44
      mask[char2idx[BOS] + 1] = 0                                # :45
      self.output_mask = mask
      dirs = 2 if bidirectional else 1
      self.output_proj = nn.Linear(dirs * hidden_size, self.adj_vocab_size)  # :51

  def forward(self, frames, frame_lens):
    B = frames.shape[0]
    x = frames.reshape(B, frames.shape[1], -1)                   # :61
    # :64-70 — descending-length sort + pack; enforce_sorted=False performs the same sort
    packed = nn.utils.rnn.pack_padded_sequence(x, frame_lens.cpu(), batch_first=True,
                                               enforce_sorted=False)
    packed_out, final = self.rnn(packed)                         # :74
    hidden, _ = nn.utils.rnn.pad_packed_sequence(packed_out, batch_first=True)  # :78

    def cat_dirs(s):                                             # :98-112
      return torch.cat([s[0::2], s[1::2]], dim=2)
    if self.bidirectional:
      final = tuple(cat_dirs(s) for s in final) if isinstance(final, tuple) else cat_dirs(final)
    if not self.enable_ctc:
      return hidden, final
    logits = self.output_proj(hidden)                            # :92
    log_probs = masked_log_softmax(logits, self.output_mask.expand(B, self.adj_vocab_size))  # :93
    return log_probs, hidden, final


# ------------------------------------------------------------------------------------------
# A4: CTC loss — src/train/ctc_loss.py:28-114
# ------------------------------------------------------------------------------------------
def ctc_nll_per_sample(log_probs, labels_plus1, frame_lens, label_lens):
  """-log p(label | lattice) per sample with torch's CPU ctc_loss, blank = 0 (ctc_loss.py:85)."""
  tgt = torch.cat([labels_plus1[i, :int(label_lens[i])] for i in range(len(label_lens))])
  return F.ctc_loss(log_probs.transpose(0, 1), tgt.to(torch.int32),
                    frame_lens.to(torch.int32), label_lens.to(torch.int32), blank=0,
                    reduction='none')


def ctc_loss(encoder_outputs, labels, frame_lens, label_lens, reduction):
  """The reference's reduction over a batch, restated on per-sample losses.

  Returns a differentiable 0-dim tensor or None, exactly where the reference returns None.
  Follows ctc_loss.py: ascending-length assert :39; label_len>256 filter :46-56; equal-length
  runs :64-65; labels+1 :80; inf fallback :87-101 (a fully-inf run is skipped BEFORE
  prev_change_point advances, :92 vs :107); 'mean' weighting with minibatch_size read before
  the slice :74,:103-105; None when the total is 0 :110-112.
  """
  frame_lens = frame_lens.to(torch.int64).cpu()
  label_lens = label_lens.to(torch.int64).cpu()
  assert bool((frame_lens[1:] - frame_lens[:-1] >= 0).all())
  keep = [i for i in range(len(label_lens)) if int(label_lens[i]) <= 256]
  if not keep:
    return None
  keep_t = torch.tensor(keep, dtype=torch.long)
  lp = encoder_outputs.cpu().index_select(0, keep_t)
  lab = labels.cpu().index_select(0, keep_t).to(torch.int64) + 1
  fl, ll = frame_lens[keep_t], label_lens[keep_t]
  # one detached call decides which samples are inf (the reference's fallback probes them one
  # by one, :4-13); the differentiable call then sees only the finite ones, as at :95-101,
  # so an inf sample never enters the graph.  Runs only change the weighting.
  n = len(keep)
  with torch.no_grad():
    nll_probe = ctc_nll_per_sample(lp, lab, fl, ll)
  finite = [k for k in range(n) if not math.isinf(float(nll_probe[k]))]
  pos = {k: j for j, k in enumerate(finite)}
  if finite:
    ft = torch.tensor(finite, dtype=torch.long)
    nll_f = ctc_nll_per_sample(lp.index_select(0, ft), lab[ft], fl[ft], ll[ft])
  bounds = [k for k in range(1, n) if int(fl[k]) != int(fl[k - 1])] + [n]
  total, count, prev, cur_len = None, 0, 0, n
  for cp in bounds:
    mb = cur_len
    idx = list(range(prev, cp))
    cur_len = len(idx)
    if any(k not in pos for k in idx):
      idx = [k for k in idx if k in pos]
      if not idx:
        continue
      cur_len = mb = len(idx)
    sel = nll_f[torch.tensor([pos[k] for k in idx])]
    if reduction == 'mean':
      run = (sel / ll[torch.tensor(idx)].clamp(min=1).to(sel.dtype)).mean() * mb
      count += mb
    else:
      run = sel.sum()
    total = run if total is None else total + run
    prev = cp
  if total is None or float(total.detach()) == 0.0:
    return None
  return total / count if reduction == 'mean' else total


# ------------------------------------------------------------------------------------------
# A6: greedy decode — src/models/lipreader/decoder.py:165-197
# ------------------------------------------------------------------------------------------
def ctc_labels(char2idx=None):
  """The build-defined label list for the live model's V'=V+1 layout (the reference has no
  live caller that defines one): index 0 = blank '_', index i+1 = idx2char[i]."""
  c2i = char2idx or default_char2idx()
  inv = {v: k for k, v in c2i.items()}
  return ['_'] + [inv[i] for i in range(len(inv))]


def greedy_decode(probs, sizes, labels, blank_index=0):
  """argmax -> drop blanks -> drop frames equal to the previous frame's argmax."""
  best = torch.max(probs, 2)[1]
  strings, offsets = [], []
  for b in range(best.shape[0]):
    n = int(sizes[b]) if sizes is not None else best.shape[1]
    out, off = '', []
    for t in range(n):
      ch = labels[int(best[b, t])]
      if ch == labels[blank_index]:
        continue
      if t != 0 and ch == labels[int(best[b, t - 1])]:
        continue
      out += ch
      off.append(t)
    strings.append([out])
    offsets.append([torch.tensor(off, dtype=torch.int)])
  return strings, offsets


def edit_distance(a, b):
  """Levenshtein distance (what decoder.py:44-73 delegates to the Levenshtein package)."""
  prev = list(range(len(b) + 1))
  for i, ca in enumerate(a, 1):
    cur = [i]
    for j, cb in enumerate(b, 1):
      cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
    prev = cur
  return prev[-1]


# ------------------------------------------------------------------------------------------
# A5: the encoder+CTC part of one optimisation step — src/train/train_better_model.py:24-80
# ------------------------------------------------------------------------------------------
def encoder_ctc_step(encoder, opt, frames, frame_lens, chars, char_lens, grad_norm=None,
                     reduction='mean'):
  """labels = chars[:,1:] (:31-32) -> encoder (:46) -> ctc_loss 'mean' (:48) -> backward (:74)
  -> clip (:78) -> opt.step (:80).  The attention decoder loop (:56-65) is the N1 'next' row
  and not part of this oracle step.  Returns the loss (None if the batch is skipped)."""
  labels = chars[:, 1:]
  label_lens = char_lens - 1
  log_probs, _, _ = encoder(frames, frame_lens)
  loss = ctc_loss(log_probs, labels, frame_lens, label_lens, reduction)
  if loss is None:
    return None
  opt.zero_grad()
  loss.backward()
  if grad_norm is not None:
    torch.nn.utils.clip_grad_norm_(encoder.parameters(), grad_norm)
  opt.step()
  return loss.detach()


# ------------------------------------------------------------------------------------------
# A8: 3-D conv frontend — BUILD-DEFINED, no reference symbol.
# The specification is lipreading_amd/frontend.py's; this is the same arithmetic on stock torch
# CPU ops.  `emulate_bf16` rounds the input, the weights and each layer's post-ReLU activation
# to bfloat16 (straight-through for gradients), which is what the HIP path stores, so that the
# comparison isolates accumulation-order differences.
# ------------------------------------------------------------------------------------------
CONV_LAYERS = ((3, 32, (3, 5, 5), 2, (1, 2, 2)), (32, 64, (3, 5, 5), 1, (1, 2, 2)),
               (64, 96, (3, 3, 3), 1, (1, 1, 1)))


def _bf16_ste(x):
  return x + (x.to(torch.bfloat16).to(torch.float32) - x).detach()


def conv_frontend(clips, params, emulate_bf16=True):
  """clips (B,T,3,H,W) uint8 or float; params = [w1,b1,w2,b2,w3,b3] in torch Conv3d layout.
  Returns features (B, T, h*w*96) in (h, w, c) order."""
  x = clips.float() / 255.0 if clips.dtype == torch.uint8 else clips.float()
  x = x.permute(0, 2, 1, 3, 4)                        # (B, 3, T, H, W)
  rnd = _bf16_ste if emulate_bf16 else (lambda v: v)
  x = rnd(x)
  for li, (_, _, _, stride, pad) in enumerate(CONV_LAYERS):
    w, b = params[2 * li], params[2 * li + 1]
    x = F.conv3d(x, rnd(w), b, stride=(1, stride, stride), padding=pad)
    x = rnd(torch.relu(x))
    x = F.max_pool3d(x, (1, 2, 2))
  B, C, T, h, w_ = x.shape
  return x.permute(0, 2, 3, 4, 1).reshape(B, T, h * w_ * C)


# ------------------------------------------------------------------------------------------
# A9: mouth crop — BUILD-DEFINED (the reference defines `_mouth = slice(48, 68)` at
# src/utils/data/face.py:21 and never uses it).  Same formula as lr_lip_crop_u8, in float32 numpy.
# ------------------------------------------------------------------------------------------
def lip_crop(frames, lmks, size=96, margin=0.3, lo=48, hi=68):
  frames = np.asarray(frames)
  lmks = np.asarray(lmks, dtype=np.float32)
  n, _, H, W = frames.shape
  out = np.zeros((n, 3, size, size), dtype=np.uint8)
  f32 = np.float32
  for i in range(n):
    x, y = lmks[i, lo:hi, 0], lmks[i, lo:hi, 1]
    x0, x1, y0, y1 = x.min(), x.max(), y.min(), y.max()
    side = max(f32(max(x1 - x0, y1 - y0)) * f32(f32(1) + f32(2) * f32(margin)), f32(2))
    left = f32(0.5) * (x0 + x1) - f32(0.5) * side
    top = f32(0.5) * (y0 + y1) - f32(0.5) * side
    scale = f32(side / f32(size))
    o = np.arange(size, dtype=np.float32)
    sx = np.clip(left + (o + f32(0.5)) * scale - f32(0.5), 0, W - 1).astype(np.float32)
    sy = np.clip(top + (o + f32(0.5)) * scale - f32(0.5), 0, H - 1).astype(np.float32)
    ix, iy = np.floor(sx).astype(int), np.floor(sy).astype(int)
    ix1, iy1 = np.minimum(ix + 1, W - 1), np.minimum(iy + 1, H - 1)
    fx, fy = (sx - ix).astype(np.float32), (sy - iy).astype(np.float32)
    img = frames[i].astype(np.float32)
    a, b = img[:, iy][:, :, ix], img[:, iy][:, :, ix1]
    d, e = img[:, iy1][:, :, ix], img[:, iy1][:, :, ix1]
    tv = a + (b - a) * fx[None, None, :]
    bv = d + (e - d) * fx[None, None, :]
    v = tv + (bv - tv) * fy[None, :, None]
    out[i] = np.clip(np.floor(v + f32(0.5)), 0, 255).astype(np.uint8)
  return out


# ------------------------------------------------------------------------------------------
# N1: attention decoder — src/models/lipreader/better_model.py:124-235 (CharDecodingStep) and
# the decoder loop of src/train/train_better_model.py:54-65 (train) / :121-135 (eval).
# masked_softmax is allennlp.nn.util's published (non memory-efficient) definition, the same
# unpinned third-party boundary as masked_log_softmax (SHIMMED pin, see header).
# ------------------------------------------------------------------------------------------
def masked_softmax(vector, mask, dim=-1):
  mask = mask.float()
  while mask.dim() < vector.dim():
    mask = mask.unsqueeze(1)
  result = F.softmax(vector * mask, dim=dim) * mask
  return result / (result.sum(dim=dim, keepdim=True) + 1e-13)


class OracleCharDecodingStep(nn.Module):
  """Same parameters / state_dict keys as the reference CharDecodingStep (better_model.py:125-159)."""

  def __init__(self, hidden_size, rnn_type, num_layers, char_dim, vocab_size, char2idx,
               attention_type='none', attn_hidden_size=-1, rnn_dropout=0):
    super().__init__()
    self.hidden_size, self.rnn_type, self.num_layers = hidden_size, rnn_type, num_layers
    self.char_dim, self.vocab_size, self.attention_type = char_dim, vocab_size, attention_type
    mask = torch.ones(vocab_size)
    mask[char2idx[PAD]] = 0                                        # :143
    mask[char2idx[BOS]] = 0                                        # :144
    self.output_mask = mask
    self.embedding = nn.Embedding(vocab_size, char_dim, padding_idx=char2idx[PAD])   # :145
    self.rnn = getattr(nn, rnn_type)(char_dim, hidden_size, num_layers=num_layers, batch_first=True,
                                     dropout=rnn_dropout)         # :146-147
    if attention_type == '1_layer_nn':
      self.attn_proj_1_layer_nn = nn.Linear(2 * hidden_size, 1)    # :149
    elif attention_type == 'general':
      self.attn_proj_general = nn.Linear(hidden_size, hidden_size)  # :151
    elif attention_type == 'concat':
      self.attn_proj_layer1 = nn.Linear(2 * hidden_size, attn_hidden_size)   # :153
      self.attn_proj_layer2 = nn.Linear(attn_hidden_size, 1)       # :154
    self.concat_layer = nn.Linear(2 * hidden_size, hidden_size)    # :155
    self.output_proj = nn.Linear(hidden_size, vocab_size)          # :156

  def forward(self, input_, previous_state, encoder_lens, enc):
    B, T = input_.shape[0], enc.shape[1]
    enc_mask = torch.arange(T).expand(B, T) < encoder_lens.unsqueeze(1)       # :173
    emb = self.embedding(input_).unsqueeze(1)                                 # :175-177
    hidden_state, final_state = self.rnn(emb, previous_state)                  # :181
    h = hidden_state.squeeze(1)
    he = hidden_state.expand_as(enc)                                           # :184
    at = self.attention_type
    if at == '1_layer_nn':
      logits = self.attn_proj_1_layer_nn(torch.cat([enc, he], dim=2)).squeeze(-1)      # :186-190
    elif at == 'general':
      logits = (self.attn_proj_general(he) * enc).sum(-1)                      # :191-205
    elif at == 'dot':
      logits = (he * enc).sum(-1)                                              # :206-208
    elif at == 'concat':
      logits = self.attn_proj_layer2(self.attn_proj_layer1(torch.cat([enc, he], dim=2)).tanh()).squeeze(-1)  # :209-215
    if at != 'none':
      w = masked_softmax(logits, enc_mask, dim=-1).unsqueeze(1)                # :219
      ctx = w.bmm(enc).squeeze(1)                                              # :221
      new_h = self.concat_layer(torch.cat([ctx, h], dim=1)).tanh()             # :223-224
      out = self.output_proj(new_h)                                            # :226
    else:
      out = self.output_proj(h)                                                # :228
    return masked_log_softmax(out, self.output_mask.expand(B, self.vocab_size)), final_state   # :229


def decoder_loop(dec, chars, char_lens, enc, enc_lens, prev_state, pad=0):
  """Teacher-forced decoder loop (train_better_model.py:56-65 with teacher_forcing_ratio=1; the
  multinomial draws of :63 do not influence the loss then).  Returns (decoder_loss, log_probs)."""
  labels = chars[:, 1:]
  max_label_len = int((char_lens - 1).max())
  loss, outs = 0, []
  for i in range(max_label_len):
    lp, prev_state = dec(chars[:, i], prev_state, enc_lens, enc)
    loss = loss + F.nll_loss(lp, labels[:, i], ignore_index=pad, reduction='sum')
    outs.append(lp)
  loss = loss / (labels != pad).sum()
  return loss, torch.stack(outs, 1)


# ---- A10 (build-defined): transformer encoder — NO reference symbol --------------------
class OracleTransformerEncoder(nn.Module):
  """CPU oracle of lipreading_amd/transformer.py: Linear + sinusoidal positions -> torch's own
  nn.TransformerEncoder (post-LN, ReLU, dropout 0, key padding mask) -> the reference's CTC head
  (better_model.py:92-93).  Parity unpinned: nothing in the reference to match."""

  def __init__(self, frame_dim, d_model, nhead, num_layers, dim_feedforward, vocab_size, char2idx, max_len=512):
    super().__init__()
    import math
    self.input_proj = nn.Linear(frame_dim, d_model)
    layer = nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout=0.0, activation='relu',
                                       batch_first=True, norm_first=False)
    self.encoder = nn.TransformerEncoder(layer, num_layers, enable_nested_tensor=False)
    self.layers = self.encoder.layers          # same state_dict names as the HIP-backed module
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    self.register_buffer("pe", pe, persistent=False)
    self.output_proj = nn.Linear(d_model, vocab_size + 1)
    mask = torch.ones(vocab_size + 1)
    mask[char2idx[PAD] + 1] = 0
    mask[char2idx[BOS] + 1] = 0
    self.register_buffer("output_mask", mask, persistent=False)

  def forward(self, frames, frame_lens):
    B, T = frames.shape[0], int(frame_lens.max())
    x = frames.reshape(B, frames.shape[1], -1)[:, :T]
    h = self.input_proj(x) + self.pe[:T]
    pad = torch.arange(T).unsqueeze(0) >= frame_lens.unsqueeze(1)
    h = self.encoder(h, src_key_padding_mask=pad)
    logits = self.output_proj(h)
    return masked_log_softmax(logits, self.output_mask.expand_as(logits)), h
