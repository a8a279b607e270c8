"""Probe: where does the one-launch recurrence differ from the step kernels?  python tools/probes/cluster_diff.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lipreading_amd import _C
from lipreading_amd.data import default_char2idx
from lipreading_amd.encoder import VideoEncoder

dev = torch.device("cuda:0")
for rnn_type, H, B, T, bi in (("GRU", 40, 11, 6, True), ("GRU", 64, 8, 6, False), ("GRU", 32, 8, 6, False), ("GRU", 96, 8, 6, False),
                              ("LSTM", 12, 3, 5, True), ("LSTM", 32, 3, 5, True), ("LSTM", 64, 8, 6, False), ("GRU", 48, 17, 21, False)):
  torch.manual_seed(41)
  enc = VideoEncoder(64, H, rnn_type=rnn_type, num_layers=1, bidirectional=bi, enable_ctc=True, vocab_size=64,
                     char2idx=default_char2idx()).to(dev)
  g = torch.Generator().manual_seed(42)
  x = torch.randn(B, T, 64, 1, generator=g).to(dev)
  lens = torch.full((B,), T)
  out = {}
  for mode in ("f32", "split"):
    enc.recurrence = mode
    with torch.no_grad():
      lp, hid, fin = enc(x, lens, max_len=T)
    out[mode] = hid.cpu()
  d = (out["f32"] - out["split"]).abs()
  D = 2 if bi else 1
  print(rnn_type, H, "B", B, "T", T, "bi", bi, "errs", _C.lib().lr_rnn_pair_errors(), "max", float(d.max()),
        "| per t:", ["%.1e" % float(d[:, t].max()) for t in range(T)])
  dd = d.reshape(B, T, D, H)
  print("    per sample:", ["%.1e" % float(dd[b].max()) for b in range(B)])
  print("    per unit (dir 0):", ["%.0e" % float(dd[:, :, 0, u].max()) for u in range(min(H, 48))])
