"""Sweep of the grid recurrence's polling knobs (lr_rnn_debug_tune which = 2..5): pass times of one LSTM-1536 layer.
python tools/probes/grid_tune.py [B T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lipreading_amd import _C  # noqa: E402
from lipreading_amd.data import default_char2idx  # noqa: E402
from lipreading_amd.encoder import VideoEncoder  # noqa: E402

B, T = (int(v) for v in (sys.argv[1:3] + ["32", "31"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
L = _C.lib()
torch.manual_seed(1)
enc = VideoEncoder(64, 1536, rnn_type="LSTM", num_layers=1, bidirectional=False, enable_ctc=True, vocab_size=64,
                   char2idx=default_char2idx()).to(dev)
x = torch.randn(B, T, 64, 1, device=dev)
lens = torch.full((B,), T)


def measure(reps=8):
  tf = tb = 0.0
  for i in range(reps + 2):
    enc.zero_grad()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    lp, hid, fin = enc(x, lens, max_len=T)
    e1.record()
    (hid.pow(2).sum()).backward()
    e2.record()
    torch.cuda.synchronize()
    if i >= 2:
      tf += e0.elapsed_time(e1)
      tb += e1.elapsed_time(e2)
  return tf / reps * 1e3, tb / reps * 1e3


def setall(vals):
  for which, (d, r) in vals.items():
    L.lr_rnn_debug_tune(which, d, r)


base = {2: (0, 1), 3: (0, 1), 4: (0, 1), 5: (0, 1)}
setall(base)
print("default                 fwd %.0f us  bwd %.0f us (whole layer passes incl. projections)" % measure())
names = {2: "fwd h gather", 3: "fwd partial gather", 4: "bwd dh gather", 5: "bwd dG gather"}
for which in (2, 3, 4, 5):
  for d, r in ((4, 1), (8, 1), (12, 1), (16, 1), (24, 1), (0, 0), (0, 2), (0, 4), (8, 0), (8, 4)):
    v = dict(base)
    v[which] = (d, r)
    setall(v)
    f, b = measure()
    print("%-20s delay %2d sleep %d   fwd %.0f us  bwd %.0f us" % (names[which], d, r, f, b))
setall(base)
print("default again           fwd %.0f us  bwd %.0f us" % measure())
print("fault word", L.lr_rnn_pair_errors())
