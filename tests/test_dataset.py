"""N3: on-disk formats and dataset invariants (CPU).  The last test lets the REFERENCE's own
loader read what this repo writes, when the reference is present (build container only)."""
import os
import pickle
import sys
import types

import numpy as np
import pytest

from lipreading_amd import dataset as DS
from lipreading_amd.data import default_char2idx


@pytest.fixture()
def root(tmp_path):
  DS.write_synthetic_dataview(str(tmp_path), "synthetic/nano", n_videos=5, captions_per_video=6)
  return str(tmp_path)


def test_dataview_layout_and_split(root):
  d = DS.datasets_path(root, "synthetic/nano")
  vids = sorted(os.listdir(d))
  assert len(vids) == 5
  for base in ("s_e", "face_lmk_seq", "cap"):
    arr = np.load(os.path.join(d, vids[0], base + ".npy"), allow_pickle=True)
    assert arr.dtype == object and len(arr) == 6
  lm = np.load(os.path.join(d, vids[0], "face_lmk_seq.npy"), allow_pickle=True)[0]
  assert lm.shape[1:] == (68, 3)
  tr, va, te = DS.split_dataset(root, "synthetic/nano", 0.8, np.random.RandomState(123456))
  assert (len(tr), len(va), len(te)) == (4, 0, 1)            # data_loader.py:53-58 arithmetic
  tr2, _, _ = DS.split_dataset(root, "synthetic/nano", 0.8, np.random.RandomState(123456))
  assert tr == tr2


def test_dataset_invariants_and_pickle_cache(root):
  tr, _, te = DS.split_dataset(root, "synthetic/nano", 0.8, np.random.RandomState(1))
  ds = DS.FrameCaptionDataset(root, "synthetic/nano", "train", tr)
  lens = [ds[i][0].shape[0] for i in range(len(ds))]
  assert lens == sorted(lens)                                  # sort_by_seqlen
  for i in range(len(ds)):
    frames, cap = ds[i]
    assert cap[0] == 1 and cap[-1] == 2 and len(cap) < len(frames) + 1   # BOS/EOS, CTC feasible
    assert frames.shape[1:] == (68, 3)
  assert ds.char2idx == default_char2idx()                     # no labels.json -> fallback vocab
  pdir = DS.pickles_path(root, "synthetic/nano", "non-sentence", "train")
  assert sorted(os.listdir(pdir)) == ["captions.pkl", "char2idx.pkl", "frames.pkl"]
  ds2 = DS.FrameCaptionDataset(root, "synthetic/nano", "train", tr)     # now from the cache
  assert len(ds2) == len(ds)
  for i in range(len(ds)):
    np.testing.assert_array_equal(ds[i][0], ds2[i][0])
    np.testing.assert_array_equal(ds[i][1], ds2[i][1])


def test_pixel_dataview_keeps_frames_aligned_with_landmarks(tmp_path):
  """frames=True (build-defined pixel regime): every caption also carries its uint8 frames; the dataset applies
  the SAME occlusion filter, sort and vocabulary to them as to the landmarks, caches them beside the reference's
  three pickles, and leaves those three readable as before."""
  root = str(tmp_path)
  DS.write_synthetic_dataview(root, "synthetic/px", n_videos=4, captions_per_video=3, frames=True, frame_hw=48,
                              min_seconds=0.6, max_seconds=1.2)
  d = DS.datasets_path(root, "synthetic/px")
  vid = sorted(os.listdir(d))[0]
  fr = np.load(os.path.join(d, vid, "face_frames_seq.npy"), allow_pickle=True)
  lm = np.load(os.path.join(d, vid, "face_lmk_seq.npy"), allow_pickle=True)
  assert fr[0].dtype == np.uint8 and fr[0].shape[1:] == (3, 48, 48) and fr[0].shape[0] == lm[0].shape[0]
  assert (lm[0][..., :2] >= 0).all() and (lm[0][..., :2] < 48).all()         # landmarks in the frames' pixels
  ids = DS.gen_vid_ids(root, "synthetic/px", np.random.RandomState(0))
  px = DS.FrameCaptionDataset(root, "synthetic/px", "train", ids, pixels=True)
  plain = DS.FrameCaptionDataset(root, "synthetic/px", "train", ids)         # the landmark view of the same cache
  assert len(px) == len(plain) > 0
  lens = []
  for i in range(len(px)):
    (frames, lmk), cap = px[i]
    np.testing.assert_array_equal(lmk, plain[i][0])
    np.testing.assert_array_equal(cap, plain[i][1])
    assert frames.shape[0] == lmk.shape[0] and frames.shape[1:] == (3, 48, 48)
    lens.append(frames.shape[0])
  assert lens == sorted(lens)
  pdir = DS.pickles_path(root, "synthetic/px", "non-sentence", "train")
  assert sorted(os.listdir(pdir)) == ["captions.pkl", "char2idx.pkl", "face_frames.pkl", "frames.pkl"]
  again = DS.FrameCaptionDataset(root, "synthetic/px", "train", ids, pixels=True)    # from the cache
  np.testing.assert_array_equal(again[0][0][0], px[0][0][0])
  # a landmark-only dataview has no frames to offer: loud, not silent
  DS.write_synthetic_dataview(root, "synthetic/lm", n_videos=2, captions_per_video=2)
  with pytest.raises(AssertionError, match="frames=True"):
    DS.FrameCaptionDataset(root, "synthetic/lm", "train", DS.gen_vid_ids(root, "synthetic/lm"), pixels=True)


def test_driver_flags_of_the_pixel_regime_and_archived_defaults(tmp_path):
  from lipreading_amd import driver
  f = driver.parse_flags(["--frontend=conv3d", "--encoder=transformer", "--hidden_size=64"])
  assert f["frontend"] == "conv3d" and f["encoder"] == "transformer" and f["ctc_only"] and f["enable_ctc"]
  with pytest.raises(SystemExit):
    driver.parse_flags(["--frontend=resnet"])
  # the archived trainer's defaults (CTC-only, bidirectional) apply only where the file / command line is silent
  cfg = tmp_path / "micro"
  cfg.write_text("--dataset=x\n--batch=5\n--hidden_size=800\n--hidden_layers=5\n--rnn_type=gru\n")
  f = driver.parse_flags([str(cfg)])
  assert f["ctc_only"] and f["bidirectional"] and f["num_layers"] == 5 and f["rnn_type"] == "GRU"
  f = driver.parse_flags([str(cfg), "--bidirectional=False"])
  assert f["ctc_only"] and f["bidirectional"] is False


def test_filter_occlusions_and_unknown_characters():
  f = [np.zeros((30, 68, 3)), np.zeros((10, 68, 3)), np.zeros((30, 68, 3))]
  c = ["short", "this caption is far too long", "ok cap"]
  se = [(0.0, 1.0), (0.0, 0.3), (0.0, 2.0)]                    # third: 2 s * 29.97 * 0.8 > 30 frames
  kf, kc = DS.filter_occlusions(f, c, se)
  assert kc == ["short"]
  ids = DS.parse_caption(default_char2idx(), "a~b")            # '~' is not in the vocabulary
  assert ids.tolist() == [1, default_char2idx()['a'], 3, default_char2idx()['b'], 2]


def test_labels_json_vocab(tmp_path):
  os.makedirs(DS.raw_path(str(tmp_path), "x"))
  with open(os.path.join(DS.raw_path(str(tmp_path), "x"), "labels.json"), "w") as f:
    f.write('["a", "b", " "]')
  v = DS.build_vocab(str(tmp_path), "x")
  assert v == {'<PAD>': 0, '<BOS>': 1, '<EOS>': 2, '<UNK>': 3, 'a': 4, 'b': 5, ' ': 6}


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference not present")
def test_reference_loader_reads_our_pickle_cache(root, monkeypatch):
  """The reference's FrameCaptionDataset (cache branch, data_loader.py:175-176) and
  load_dataset (:64-78) read the pickles this repo writes and yield identical samples."""
  tr, _, _ = DS.split_dataset(root, "synthetic/nano", 0.8, np.random.RandomState(1))
  ours = DS.FrameCaptionDataset(root, "synthetic/nano", "train", tr)
  monkeypatch.setenv("LIP_READING_WS_PATH", root)
  monkeypatch.syspath_prepend("/root/reference")
  sys.modules.setdefault("spacy", types.ModuleType("spacy"))   # inert: only sentence splitting uses it
  import src.utils.utility as ref_util
  monkeypatch.setattr(ref_util, "_ws_dir", None, raising=False)
  import src.data.data_loader as ref_dl
  theirs = ref_dl.FrameCaptionDataset("synthetic/nano", "train", tr)
  assert len(theirs) == len(ours) and theirs.char2idx == ours.char2idx
  for i in range(len(ours)):
    f_r, c_r = theirs[i]
    np.testing.assert_array_equal(f_r, ours[i][0])
    np.testing.assert_array_equal(c_r, ours[i][1])
