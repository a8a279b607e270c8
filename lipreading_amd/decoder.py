"""CTC decoders — drop-in for `src/models/lipreader/decoder.py` (Decoder :23, GreedyDecoder :146).

GreedyDecoder.decode keeps the reference's contract — `(strings, offsets)` with
`strings[b] == [str]` and `offsets[b] == [IntTensor]` — but the per-frame Python loop with one
`.item()` per element (decoder.py:168-169) becomes one kernel (argmax + collapse + ordered
compaction) and a single device->host copy of the kept ids.
"""
import torch

from . import _C


def _edit_distance(a, b):
  """Levenshtein distance; the reference delegates to the `Levenshtein` package (decoder.py:18)."""
  prev = list(range(len(b) + 1))
  for i, ca in enumerate(a, 1):
    cur = [i]
    for j, cb in enumerate(b, 1):
      cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
    prev = cur
  return prev[-1]


class Decoder(object):
  """decoder.py:23-88: label bookkeeping plus WER/CER helpers."""

  def __init__(self, labels, blank_index=0):
    self.labels = labels
    self.int_to_char = dict([(i, c) for (i, c) in enumerate(labels)])
    self.blank_index = blank_index
    space_index = len(labels)  # out-of-bounds sentinel when there is no ' ' (decoder.py:38)
    if ' ' in labels:
      space_index = labels.index(' ')
    self.space_index = space_index

  def wer(self, s1, s2):
    """Word-level edit distance (decoder.py:44-62)."""
    b = set(s1.split() + s2.split())
    word2char = dict(zip(b, range(len(b))))
    w1 = [chr(word2char[w]) for w in s1.split()]
    w2 = [chr(word2char[w]) for w in s2.split()]
    return _edit_distance(''.join(w1), ''.join(w2))

  def cer(self, s1, s2):
    """Character-level edit distance with spaces removed (decoder.py:64-73)."""
    s1, s2 = s1.replace(' ', ''), s2.replace(' ', '')
    return _edit_distance(s1, s2)

  def decode(self, probs, sizes=None):
    raise NotImplementedError


class GreedyDecoder(Decoder):
  def __init__(self, labels, blank_index=0):
    super(GreedyDecoder, self).__init__(labels, blank_index)
    # the reference compares characters, not indices (decoder.py:167-171): classes that share
    # a label string collapse together.  canonical index = first class with that string.
    first = {}
    canon = [first.setdefault(c, i) for i, c in enumerate(labels)]
    self._canon = canon if canon != list(range(len(labels))) else None
    self._canon_dev = {}

  def decode_ids(self, probs, sizes=None):
    """Device part: returns (ids (B,T) int32, offsets (B,T) int32, lens (B,) int32) on the GPU."""
    _C.require_cuda(probs, sizes)
    L = _C.lib()
    if probs.dtype != torch.float32:
      probs = probs.float()
    if probs.stride(2) != 1:
      probs = probs.contiguous()
    B, T, C = probs.shape
    if C > len(self.labels):
      raise KeyError("probs has %d classes but only %d labels" % (C, len(self.labels)))
    dev = probs.device
    ids = torch.empty((B, T), dtype=torch.int32, device=dev)
    off = torch.empty((B, T), dtype=torch.int32, device=dev)
    lens = torch.empty((B,), dtype=torch.int32, device=dev)
    sz = None if sizes is None else sizes.to(device=dev, dtype=torch.int32).contiguous()
    cmap = None
    if self._canon is not None:
      cmap = self._canon_dev.get(dev)
      if cmap is None:
        cmap = torch.tensor(self._canon, dtype=torch.int32, device=dev)
        self._canon_dev[dev] = cmap
    blank = self.blank_index if self._canon is None else self._canon[self.blank_index]
    _C.check(L.lr_ctc_greedy_decode(probs.data_ptr(), probs.stride(0), probs.stride(1),
                                    _C.ptr(sz), _C.ptr(cmap), ids.data_ptr(), off.data_ptr(),
                                    lens.data_ptr(), B, T, C, blank, _C.stream_handle()),
             "lr_ctc_greedy_decode")
    return ids, off, lens

  def decode(self, probs, sizes=None):
    """decoder.py:182-197: probs (B,T,C) -> ([[str]], [[IntTensor offsets]])."""
    ids, off, lens = self.decode_ids(probs, sizes)
    ids, off, lens = ids.cpu(), off.cpu(), lens.cpu().tolist()
    strings, offsets = [], []
    for b, n in enumerate(lens):
      chars = [self.int_to_char[i] for i in ids[b, :n].tolist()]
      if chars and self.space_index >= len(self.labels):
        # decoder.py:174 indexes labels[space_index] for every kept character
        raise IndexError("list index out of range")
      strings.append([''.join(chars)])
      offsets.append([off[b, :n].clone().to(torch.int)])
    return strings, offsets


def ctc_labels(char2idx):
  """Label list for the live model's V'=V+1 class layout (better_model.py:38,43-45): index 0 is
  the CTC blank '_', index i+1 the character with id i.  The reference defines none because its
  greedy decoder has no live caller."""
  inv = {v: k for k, v in char2idx.items()}
  return ['_'] + [inv[i] for i in range(len(inv))]
