"""Batch collation on the device — drop-in for `src/data/data_loader.py:117 _collate_fn`.

The reference zero-pads on the host with one `torch.Tensor(seq)` copy per sample
(data_loader.py:139-141) and ships the padded batch; here the ragged landmark rows are
uploaded once, back to back, and padded in HBM by lr_collate_pad_f32.
"""
import numpy as np
import torch

from . import _C

# src/data/data_loader.py:29-35
BOS = '<BOS>'
EOS = '<EOS>'
PAD = '<PAD>'
UNK = '<UNK>'
_markers2Id = {PAD: 0, BOS: 1, EOS: 2, UNK: 3}
_labels = list(" !\"#$%&'()*+,-./0123456789:;<>?@[]abcdefghijklmnopqrstuvwxyz")


def default_char2idx():
  """The vocabulary `build_vocab` falls back to (data_loader.py:100-115): 4 markers + 60 chars."""
  c2i = dict(_markers2Id)
  for ch in _labels:
    c2i[ch] = len(c2i)
  return c2i


def pad_frames(seqs, device):
  """(len_i, ...) float sequences -> ((B, Tmax, ...) float32 on `device`, lens int64)."""
  assert len(seqs) > 0
  arrs = [np.asarray(s, dtype=np.float32) for s in seqs]
  tail = arrs[0].shape[1:]
  assert all(a.shape[1:] == tail for a in arrs)  # data_loader.py:132
  lens = np.array([len(a) for a in arrs], dtype=np.int64)
  feat = int(np.prod(tail)) if tail else 1
  offsets = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
  packed = torch.from_numpy(np.concatenate([a.reshape(len(a), feat) for a in arrs], axis=0))
  B, t_max = len(arrs), int(lens.max())
  dev = torch.device(device)
  if dev.type != "cuda":
    raise _C.LipReadingHipError("collation runs on the MI355X only (no CPU fallback)")
  packed_d = packed.to(dev, non_blocking=True)
  off_d = torch.from_numpy(offsets).to(dev, non_blocking=True)
  lens_d = torch.from_numpy(lens.astype(np.int32)).to(dev, non_blocking=True)
  out = torch.empty((B, t_max, feat), dtype=torch.float32, device=dev)
  with torch.cuda.device(dev):
    _C.check(_C.lib().lr_collate_pad_f32(packed_d.data_ptr(), off_d.data_ptr(), lens_d.data_ptr(),
                                         out.data_ptr(), B, t_max, feat, _C.stream_handle()),
             "lr_collate_pad_f32")
  return out.reshape((B, t_max) + tuple(tail)), torch.from_numpy(lens)


def make_collate_fn(device):
  """Returns a `_collate_fn(batch)` with the reference's contract (data_loader.py:117-152):
  batch = [(frames (len,68,3), caption (n,))] -> (frames f32 (B,Tmax,68,3), frame_lens i64,
  chars i64 (B,Cmax) PAD=0, char_lens i64); frames live on `device`, chars and both lens on the host."""
  def _collate_fn(batch):
    assert all(len(x) == 2 for x in batch)
    frames, captions = zip(*batch)
    src, src_lens = pad_frames(frames, device)
    caps = [np.asarray(c, dtype=np.int64) for c in captions]
    tgt_lens = torch.tensor([len(c) for c in caps], dtype=torch.long)
    tgt = torch.zeros((len(caps), int(tgt_lens.max())), dtype=torch.long)
    for i, c in enumerate(caps):
      tgt[i, :len(c)] = torch.from_numpy(c)
    # chars stay on the host like the lengths (the reference's collate returns host tensors and train()
    # uploads them): the framing asserts of train_better_model.py:26-33 then need no device read
    return src, src_lens, tgt, tgt_lens
  return _collate_fn


def make_pixel_collate_fn(device, size=96, margin=0.3):
  """BUILD-DEFINED (the pixel regime of BASELINE configs[1]/[4]; the reference has no pixel path): batch =
  [((frames u8 (len,3,H,W), landmarks (len,68,3) in those frames' pixels), caption (n,))] ->
  (clips u8 (B,Tmax,3,size,size), frame_lens i64, chars i64 (B,Cmax) PAD=0, char_lens i64) — the tuple train() /
  eval() / greedy_cer take, with mouth-crop clips where the landmark regime has landmarks.  Frames and landmarks
  are uploaded ragged, back to back; every sample's frames are cropped about their own mouth landmarks and
  resampled to size x size straight into their rows of the zero-padded batch (lr_lip_crop_u8, landmarks.lip_crop's
  arithmetic); chars and both lengths stay on the host like make_collate_fn's."""
  from .landmarks import _mouth
  dev = torch.device(device)
  if dev.type != "cuda":
    raise _C.LipReadingHipError("collation runs on the MI355X only (no CPU fallback)")

  def _collate_fn(batch):
    assert all(len(x) == 2 and len(x[0]) == 2 for x in batch)
    pairs, captions = zip(*batch)
    pix = [np.ascontiguousarray(p[0]) for p in pairs]
    lmk = [np.asarray(p[1], dtype=np.float32) for p in pairs]
    H, W = pix[0].shape[2], pix[0].shape[3]
    assert all(p.dtype == np.uint8 and p.shape[1:] == (3, H, W) for p in pix)
    assert all(l.shape[0] == p.shape[0] and l.shape[1:] == (68, 3) for l, p in zip(lmk, pix))
    lens = np.array([len(p) for p in pix], dtype=np.int64)
    B, t_max = len(pix), int(lens.max())
    frames_d = torch.from_numpy(np.concatenate(pix, axis=0)).to(dev, non_blocking=True)
    lmk_d = torch.from_numpy(np.concatenate(lmk, axis=0)).to(dev, non_blocking=True)
    clips = torch.zeros((B, t_max, 3, size, size), dtype=torch.uint8, device=dev)
    L = _C.lib()
    off = 0
    with torch.cuda.device(dev):
      for b in range(B):
        n = int(lens[b])
        _C.check(L.lr_lip_crop_u8(frames_d[off:off + n].data_ptr(), lmk_d[off:off + n].data_ptr(), clips[b].data_ptr(),
                                  n, H, W, size, 68, _mouth.start, _mouth.stop, float(margin), _C.stream_handle()),
                 "lr_lip_crop_u8")
        off += n
    caps = [np.asarray(c, dtype=np.int64) for c in captions]
    tgt_lens = torch.tensor([len(c) for c in caps], dtype=torch.long)
    tgt = torch.zeros((len(caps), int(tgt_lens.max())), dtype=torch.long)
    for i, c in enumerate(caps):
      tgt[i, :len(c)] = torch.from_numpy(c)
    return clips, torch.from_numpy(lens), tgt, tgt_lens
  return _collate_fn
