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
