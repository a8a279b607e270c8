"""On-disk formats and dataset invariants either side of the hot path — the
host-side mirror of `src/data/data_loader.py` (split_dataset :49, load_dataset :64,
filter_occlusions :80, sort_by_seqlen :93, build_vocab :100, FrameCaptionDataset :154).

Formats (written by the reference's `generate_dataview.py:229-233`, read at
`data_loader.py:213-257`):

  <root>/data/datasets/<name>/<vid>/s_e.npy           object array of (start, end) seconds
  <root>/data/datasets/<name>/<vid>/face_lmk_seq.npy  object array of (len_i, 68, 3) float arrays
  <root>/data/datasets/<name>/<vid>/cap.npy           object array of caption strings
  <root>/data/pickles/<name>/{sentence,non-sentence}/<split>/{char2idx,frames,captions}.pkl

Invariants the hot path relies on: samples are globally sorted by frame count (so batches have
ascending lengths, `ctc_loss.py:39`), a caption plus BOS/EOS is shorter than its frame sequence
(`data_loader.py:88`, otherwise CTC is infeasible), captions are framed with BOS/EOS.

`write_synthetic_dataview` produces dataviews in exactly this layout so the nano/micro/small
configurations can run without the YouTube corpus (which is not shipped, `.gitignore:4`).

Pixel regime (BUILD-DEFINED, BASELINE configs[1]/[4]: "3Dconv + ..."; the reference's dataview keeps only the
landmarks).  What `generate_dataview.py:58-76` has in hand per frame is the video frame AND its landmarks; a dataview
written with `frames=True` keeps both:
  <root>/data/datasets/<name>/<vid>/face_frames_seq.npy  object array of (len_i, 3, H, W) uint8 frames, aligned with
                                                         face_lmk_seq row by row (landmarks in those frames' pixels)
`FrameCaptionDataset(pixels=True)` then yields ((frames u8, landmarks), caption ids) under the SAME filter / sort /
vocabulary as the landmark dataset (its cache is <split>/face_frames.pkl beside the reference's three pickles), and
`data.make_pixel_collate_fn` turns a batch into mouth-crop clips (B, Tmax, 3, 96, 96) on the device (lr_lip_crop_u8).
"""
import glob
import json
import os
import pickle

import numpy as np

from .data import BOS, EOS, PAD, UNK, _labels, _markers2Id  # noqa: F401


def datasets_path(root, *rel):
  return os.path.join(root, "data", "datasets", *rel)     # utility.py:61-62


def pickles_path(root, *rel):
  return os.path.join(root, "data", "pickles", *rel)      # utility.py:64-65


def raw_path(root, *rel):
  return os.path.join(root, "data", "raw", *rel)          # utility.py:49-50


def gen_vid_ids(root, dataset_name, rand=None):
  """data_loader.py:37-47: sorted video directories, shuffled with the caller's RandomState."""
  vid_ids = glob.glob(os.path.join(datasets_path(root, dataset_name), '*/'))
  assert len(vid_ids) > 0, "No video ids found: '%s'" % datasets_path(root, dataset_name)
  vid_ids.sort()
  (rand if rand is not None else np.random).shuffle(vid_ids)
  return vid_ids


def split_dataset(root, dataset_name, train_split=0.8, rand=None):
  """data_loader.py:49-62: train_split / half of the rest / the remainder, by video id."""
  vid_ids = gen_vid_ids(root, dataset_name, rand=rand)
  train_idx = int(train_split * len(vid_ids))
  val_test_size = len(vid_ids) - train_idx
  val_idx = train_idx + val_test_size // 2
  return vid_ids[:train_idx], vid_ids[train_idx:val_idx], vid_ids[val_idx:]


def keep_and_order(frames, captions, start_ends, fps=29.97, threshold=0.8):
  """The indices filter_occlusions keeps, in sort_by_seqlen's order: for data that must stay aligned with the
  landmark sequences (the pixel frames of a `frames=True` dataview)."""
  kept = [i for i, (f, c, (start, end)) in enumerate(zip(frames, captions, start_ends))
          if (end - start) * fps * threshold <= len(f) and len(c) + 2 < len(f)]
  order = np.argsort([frames[i].shape[0] for i in kept])
  return [kept[i] for i in order]


def filter_occlusions(frames, captions, start_ends, fps=29.97, threshold=0.8):
  """data_loader.py:80-91: keep a sample iff at least `threshold` of the window's frames were
  captured AND len(caption) + 2 < len(frames) (room for BOS/EOS, CTC feasibility)."""
  kept_f, kept_c = [], []
  for f, c, (start, end) in zip(frames, captions, start_ends):
    if (end - start) * fps * threshold <= len(f) and len(c) + 2 < len(f):
      kept_f.append(f)
      kept_c.append(c)
  return kept_f, kept_c


def sort_by_seqlen(frames, captions):
  """data_loader.py:93-98: ascending frame count (np.argsort, i.e. not stable for ties)."""
  order = np.argsort([x.shape[0] for x in frames])
  return [frames[i] for i in order], [captions[i] for i in order]


def build_vocab(root, dataset_name, labels='labels.json'):
  """data_loader.py:100-115: 4 markers, then the characters of labels.json (or the 60-character
  fallback when the file cannot be read)."""
  try:
    with open(os.path.join(raw_path(root, dataset_name), labels)) as f:
      chars = str(''.join(json.load(f)))
  except Exception:
    chars = _labels
  char2idx = dict(_markers2Id)
  for ch in chars:
    char2idx[ch] = len(char2idx)
  return char2idx


def parse_caption(char2idx, cap):
  """data_loader.py:283-289: BOS + per-character ids (UNK when unknown) + EOS."""
  ids = [_markers2Id[BOS]] + [char2idx.get(ch, _markers2Id[UNK]) for ch in list(cap)] + [_markers2Id[EOS]]
  ids = np.array(ids)
  assert len(ids) > 2 and ids[0] == _markers2Id[BOS] and ids[-1] == _markers2Id[EOS]
  return ids


def load_pickles(pickle_dir, out_ext='.pkl'):
  """data_loader.py:64-78."""
  out = []
  for base in ('char2idx', 'frames', 'captions'):
    path = os.path.join(pickle_dir, base + out_ext)
    assert os.path.isfile(path), "File not found: '{}'".format(path)
    with open(path, 'rb') as f:
      out.append(pickle.load(f))
  return tuple(out)


def _lmk_fingerprint(lmks):
  """(frames, crc32 of the landmark bytes) of one sample: what ties a row of the pixel cache to its landmarks."""
  import zlib
  a = np.ascontiguousarray(np.asarray(lmks, dtype=np.float32))
  return (int(a.shape[0]), int(zlib.crc32(a.tobytes())))


class FrameCaptionDataset(object):
  """data_loader.py:154-257.  `__getitem__` returns (frames (len,68,3) float, caption ids)."""

  def __init__(self, root, dataset_name, split_name, vid_ids, labels='labels.json', start_end='s_e',
               threshold=0.8, fps=29.97, cap='cap', frame_type='face_lmk_seq',
               sentence_dataset=False, in_ext='.npy', out_ext='.pkl', refresh=False, pixels=False):
    assert all(os.path.isdir(x) for x in vid_ids)
    assert frame_type in ('face_lmk_seq', 'face_vtx_seq')
    if sentence_dataset:
      raise NotImplementedError("sentence re-segmentation needs spaCy (data_loader.py:260-281); "
                                "out of scope for the hot path")
    pickle_dir = pickles_path(root, dataset_name, 'non-sentence', split_name)
    if refresh or not os.path.isdir(pickle_dir):
      char2idx, frames, captions = self.construct_dataset(
          root, dataset_name, pickle_dir, vid_ids, labels=labels, start_end=start_end, cap=cap,
          frame_type=frame_type, in_ext=in_ext, out_ext=out_ext, fps=fps, threshold=threshold)
    else:
      char2idx, frames, captions = load_pickles(pickle_dir, out_ext=out_ext)
    assert len(frames) == len(captions) > 0
    self.char2idx = char2idx
    self.idx2char = {v: k for k, v in char2idx.items()}
    self.frames, self.captions = frames, captions
    self.num_elements = len(captions)
    self.frame_type = frame_type
    self.pixels = None
    if pixels:   # build-defined: the u8 frames of a `frames=True` dataview, same filter and order as the landmarks
      ppath = os.path.join(pickle_dir, 'face_frames' + out_ext)
      cached = None
      if not refresh and os.path.isfile(ppath):
        with open(ppath, 'rb') as f:
          cached = pickle.load(f)
      # The landmark / caption pickles may come from an earlier run (other vid_ids, seed or train_split) while this
      # cache is rebuilt from the CURRENT vid_ids: equal-length clips would then pair pixels with another sample's
      # landmarks and caption without an error.  Every row carries the fingerprint of the landmarks it was cut with.
      want = [_lmk_fingerprint(f) for f in self.frames]
      if not (isinstance(cached, dict) and cached.get("lmk_fingerprint") == want):
        cached = self.construct_pixels(vid_ids, ppath, start_end=start_end, cap=cap, frame_type=frame_type,
                                       in_ext=in_ext, fps=fps, threshold=threshold)
      assert cached["lmk_fingerprint"] == want, \
          ("the pixel frames under %s do not belong to the cached landmarks in %s (built from other videos or another "
           "split): pass refresh=True" % (os.path.dirname(vid_ids[0]), pickle_dir))
      self.pixels = cached["pixels"]
      assert len(self.pixels) == len(self.frames)
      assert all(p.shape[0] == f.shape[0] and p.dtype == np.uint8 and p.ndim == 4 and p.shape[1] == 3
                 for p, f in zip(self.pixels, self.frames))

  def __len__(self):
    return self.num_elements

  def __getitem__(self, index):
    frames = self.frames[index]
    assert len(frames.shape) == 3
    if self.pixels is not None:
      return (self.pixels[index], frames), parse_caption(self.char2idx, self.captions[index])
    return frames, parse_caption(self.char2idx, self.captions[index])

  @staticmethod
  def construct_pixels(vid_ids, out_path, start_end='s_e', cap='cap', frame_type='face_lmk_seq',
                       pixel_type='face_frames_seq', in_ext='.npy', fps=29.97, threshold=0.8):
    def load(base):
      rows = []
      for vid in vid_ids:
        path = os.path.join(vid, base + in_ext)
        assert os.path.isfile(path), "%s (a dataview written with frames=True has it)" % path
        rows.extend(list(np.load(path, allow_pickle=True)))
      return rows
    lmks, captions, start_ends, pix = load(frame_type), load(cap), load(start_end), load(pixel_type)
    assert len(pix) == len(lmks) == len(captions) == len(start_ends)
    order = keep_and_order(lmks, captions, start_ends, fps=fps, threshold=threshold)
    out = {"pixels": [np.ascontiguousarray(pix[i]) for i in order],
           "lmk_fingerprint": [_lmk_fingerprint(lmks[i]) for i in order]}
    with open(out_path, 'wb') as f:
      pickle.dump(out, f)
    return out

  @staticmethod
  def construct_dataset(root, dataset_name, pickle_dir, vid_ids, labels='labels.json',
                        start_end='s_e', cap='cap', frame_type='face_lmk_seq', in_ext='.npy',
                        out_ext='.pkl', fps=29.97, threshold=0.8):
    def load(base):
      rows = []
      for vid in vid_ids:
        path = os.path.join(vid, base + in_ext)
        assert os.path.isfile(path), path
        rows.extend(list(np.load(path, allow_pickle=True)))   # object arrays (numpy >= 1.16.3)
      return rows
    frames, captions, start_ends = load(frame_type), load(cap), load(start_end)
    assert len(captions) == len(frames) == len(start_ends)
    assert all(len(x.shape) == 3 for x in frames)
    assert all(isinstance(x, str) for x in captions)
    frames, captions = filter_occlusions(frames, captions, start_ends, fps=fps, threshold=threshold)
    frames, captions = sort_by_seqlen(frames, captions)
    char2idx = build_vocab(root, dataset_name, labels)
    os.makedirs(pickle_dir, exist_ok=True)
    for base, obj in (('char2idx', char2idx), ('frames', frames), ('captions', captions)):
      with open(os.path.join(pickle_dir, base + out_ext), 'wb') as f:
        pickle.dump(obj, f)
    return char2idx, frames, captions


def iterate_batches(dataset, batch_size, collate_fn):
  """What `DataLoader(dataset, batch_size, collate_fn=..., shuffle=False)` yields
  (src/scripts/train.py:209-211): consecutive, un-shuffled batches of the sorted dataset."""
  for lo in range(0, len(dataset), batch_size):
    yield collate_fn([dataset[i] for i in range(lo, min(lo + batch_size, len(dataset)))])


class BatchLoader(object):
  """`DataLoader(dataset, batch_size, collate_fn=..., shuffle=False)` (src/scripts/train.py:209-211):
  len() = number of batches; every iteration collates its batches afresh, so only the batch in
  flight lives on the device (not the whole train/val/test set for the life of the run)."""

  def __init__(self, dataset, batch_size, collate_fn):
    self.dataset, self.batch_size, self.collate_fn = dataset, batch_size, collate_fn

  def __len__(self):
    return (len(self.dataset) + self.batch_size - 1) // self.batch_size

  def __iter__(self):
    return iterate_batches(self.dataset, self.batch_size, self.collate_fn)


def make_loader(dataset, batch_size, collate_fn):
  return BatchLoader(dataset, batch_size, collate_fn)


_WORDS = ("the quick brown fox jumps over a lazy dog and then it went home to see what was going on "
          "tonight we have a great show for you folks thank you very much please welcome").split()


def write_synthetic_dataview(root, dataset_name, n_videos, captions_per_video=8, seed=123456,
                             fps=29.97, min_seconds=1.0, max_seconds=3.0, frames=False, frame_hw=96):
  """Synthetic dataviews in the reference's on-disk format.  Landmarks are smooth random walks in
  a ~200-pixel face box (x, y in pixels, z relative depth), i.e. the unnormalised magnitudes the
  reference feeds its encoder.

  frames=True (build-defined pixel regime): every caption also gets `face_frames_seq` — (len, 3, frame_hw,
  frame_hw) uint8 frames (a fixed random texture per caption whose brightness follows the mouth opening, plus
  noise) — and the landmarks live in those frames' pixel coordinates, the mouth points (48..67) clustered in the
  lower middle so that the lip crop has something to frame."""
  rng = np.random.RandomState(seed)
  if frames:
    return _write_synthetic_pixel_dataview(root, dataset_name, n_videos, captions_per_video, rng, fps, min_seconds,
                                           max_seconds, int(frame_hw))
  for v in range(n_videos):
    vid_dir = datasets_path(root, dataset_name, "vid%04d" % v)
    os.makedirs(vid_dir, exist_ok=True)
    s_e, lmks, caps = [], [], []
    t = 0.0
    for _ in range(captions_per_video):
      dur = rng.uniform(min_seconds, max_seconds)
      n = int(round(dur * fps))
      base = np.stack([rng.uniform(40, 160, 68), rng.uniform(40, 160, 68), rng.uniform(-60, 60, 68)], 1)
      walk = np.cumsum(rng.randn(n, 68, 3) * 0.5, axis=0)
      words = []
      while len(' '.join(words)) < max(3, n // 3):
        words.append(_WORDS[rng.randint(len(_WORDS))])
      cap = ' '.join(words)[:max(3, n - 3)]
      s_e.append((t, t + dur))
      lmks.append((base[None] + walk).astype(np.float64))
      caps.append(cap)
      t += dur
    for base, rows in (("s_e", s_e), ("face_lmk_seq", lmks), ("cap", caps)):
      arr = np.empty(len(rows), dtype=object)
      for i, r in enumerate(rows):
        arr[i] = r
      np.save(os.path.join(vid_dir, base + ".npy"), arr, allow_pickle=True)
  return datasets_path(root, dataset_name)


def _write_synthetic_pixel_dataview(root, dataset_name, n_videos, captions_per_video, rng, fps, min_seconds,
                                    max_seconds, hw):
  for v in range(n_videos):
    vid_dir = datasets_path(root, dataset_name, "vid%04d" % v)
    os.makedirs(vid_dir, exist_ok=True)
    s_e, lmks, caps, pix = [], [], [], []
    t = 0.0
    for _ in range(captions_per_video):
      dur = rng.uniform(min_seconds, max_seconds)
      n = int(round(dur * fps))
      base = np.stack([rng.uniform(0.2, 0.8, 68) * hw, rng.uniform(0.2, 0.8, 68) * hw, rng.uniform(-30, 30, 68)], 1)
      # mouth points: a cluster in the lower middle of the frame, opening and closing over time
      cx, cy = rng.uniform(0.4, 0.6) * hw, rng.uniform(0.6, 0.75) * hw
      ang = np.linspace(0, 2 * np.pi, 20, endpoint=False)
      openness = 0.5 + 0.5 * np.sin(np.linspace(0, rng.uniform(2, 6) * np.pi, n))
      seq = np.repeat(base[None], n, axis=0) + np.cumsum(rng.randn(n, 68, 3) * 0.1, axis=0)
      seq[:, 48:68, 0] = cx + 0.12 * hw * np.cos(ang)[None]
      seq[:, 48:68, 1] = cy + (0.03 + 0.06 * openness[:, None]) * hw * np.sin(ang)[None]
      tex = rng.randint(0, 256, (3, hw, hw)).astype(np.float32)
      fr = tex[None] * (0.6 + 0.4 * openness[:, None, None, None]) + rng.randn(n, 3, hw, hw).astype(np.float32) * 4
      words = []
      while len(' '.join(words)) < max(3, n // 3):
        words.append(_WORDS[rng.randint(len(_WORDS))])
      cap = ' '.join(words)[:max(3, n - 3)]
      s_e.append((t, t + dur))
      lmks.append(seq.astype(np.float64))
      caps.append(cap)
      pix.append(np.clip(fr, 0, 255).astype(np.uint8))
      t += dur
    for base_name, rows in (("s_e", s_e), ("face_lmk_seq", lmks), ("cap", caps), ("face_frames_seq", pix)):
      arr = np.empty(len(rows), dtype=object)
      for i, r in enumerate(rows):
        arr[i] = r
      np.save(os.path.join(vid_dir, base_name + ".npy"), arr, allow_pickle=True)
  return datasets_path(root, dataset_name)
