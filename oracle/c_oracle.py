"""ctypes wrapper of oracle/lr_oracle.c (plain-C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblr_oracle.so")
_lib = None


def build(force=False):
  src = os.path.join(_HERE, "lr_oracle.c")
  if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
    subprocess.run(["make", "-C", _HERE, "-B", "liblr_oracle.so"], check=True, capture_output=True)
  return _SO


def lib():
  global _lib
  if _lib is None:
    _lib = ctypes.CDLL(build())
  return _lib


def _p(a):
  return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
  return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
  return np.ascontiguousarray(a, dtype=np.int32)


def rnn_layer(mode, x, lens, w_ih, w_hh, b_ih, b_hh):
  """mode 'GRU'/'LSTM'; w_* are lists over directions.  Returns y, h_n, c_n (None for GRU)."""
  m = 0 if mode == "GRU" else 1
  x = _f(x)
  B, T, I = x.shape
  D = len(w_ih)
  H = w_hh[0].shape[1]
  wi, wh = _f(np.stack(w_ih)), _f(np.stack(w_hh))
  bi, bh = _f(np.stack(b_ih)), _f(np.stack(b_hh))
  lens = _i(lens)
  y = np.zeros((B, T, D * H), np.float32)
  h_n = np.zeros((D, B, H), np.float32)
  c_n = np.zeros((D, B, H), np.float32) if m == 1 else None
  rc = lib().oracle_rnn_layer(m, _p(x), _p(lens), _p(wi), _p(wh), _p(bi), _p(bh), _p(y), _p(h_n),
                              _p(c_n), B, T, I, H, D)
  assert rc == 0
  return y, h_n, c_n


def proj_logsoftmax(hidden, W, bias, mask):
  hidden = _f(hidden)
  shape = hidden.shape[:-1]
  K = hidden.shape[-1]
  R = int(np.prod(shape))
  C = W.shape[0]
  out = np.zeros((R, C), np.float32)
  rc = lib().oracle_proj_logsoftmax(_p(hidden.reshape(R, K)), _p(_f(W)), _p(_f(bias)), _p(_f(mask)),
                                    _p(out), R, K, C)
  assert rc == 0
  return out.reshape(shape + (C,))


def ctc(lp, labels_p1, frame_lens, label_lens, want_grad=True):
  lp = _f(lp)
  B, T, C = lp.shape
  lab = _i(labels_p1)
  nll = np.zeros(B, np.float32)
  grad = np.zeros_like(lp) if want_grad else None
  rc = lib().oracle_ctc(_p(lp), _p(lab), lab.shape[1], _p(_i(frame_lens)), _p(_i(label_lens)),
                        _p(nll), _p(grad), B, T, C)
  assert rc == 0
  return nll, grad


def ctc_reduce(nll, frame_lens, label_lens, reduction):
  nll = _f(nll)
  B = len(nll)
  loss = np.zeros(1, np.float32)
  w = np.zeros(B, np.float32)
  none = lib().oracle_ctc_reduce(_p(nll), _p(_i(frame_lens)), _p(_i(label_lens)),
                                 1 if reduction == "mean" else 0, _p(loss), _p(w), B)
  return (None if none else float(loss[0])), w


def greedy(probs, sizes, blank=0):
  probs = _f(probs)
  B, T, C = probs.shape
  ids = np.zeros((B, T), np.int32)
  off = np.zeros((B, T), np.int32)
  lens = np.zeros(B, np.int32)
  sz = None if sizes is None else _i(sizes)
  lib().oracle_greedy(_p(probs), _p(sz), _p(ids), _p(off), _p(lens), B, T, C, blank)
  return ids, off, lens


def apply_padding(rects, dims, padding):
  rects = _i(rects)
  out = np.zeros_like(rects)
  lib().oracle_apply_padding.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_double]
  lib().oracle_apply_padding(_p(rects), _p(_i(np.asarray(dims)[:, :2])), _p(out), len(rects),
                             float(padding))
  return out


def get_face(lmk, rects):
  lmk = _f(lmk)
  out = np.zeros_like(lmk)
  lib().oracle_get_face(_p(lmk), _p(_i(rects)), _p(out), lmk.shape[0], lmk.shape[1])
  return out
