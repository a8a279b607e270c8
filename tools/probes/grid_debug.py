"""Probe of the grid recurrence (lr_rnn_grid.hip): forward and backward of one LSTM layer against the step kernels,
with the fault word read after each pass (its bits say which wait gave up).  python tools/probes/grid_debug.py [H B T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lipreading_amd import _C  # noqa: E402
from lipreading_amd.data import default_char2idx  # noqa: E402
from lipreading_amd.encoder import VideoEncoder  # noqa: E402

H, B, T = (int(v) for v in (sys.argv[1:4] + ["1536", "32", "6"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
L = _C.lib()
torch.manual_seed(1)
enc = VideoEncoder(64, H, rnn_type="LSTM", num_layers=1, bidirectional=False, enable_ctc=True, vocab_size=64,
                   char2idx=default_char2idx()).to(dev)
g = torch.Generator().manual_seed(2)
x = torch.randn(B, T, 64, 1, generator=g).to(dev)
lens = torch.full((B,), T)
print("status", L.lr_rnn_one_launch_status(1, B, T, 64, H, 1), "launches", L.lr_rnn_pass_launches(1, B, T, 64, H, 1))
res = {}
for mode in ("f32", "split"):
  enc.recurrence = mode
  enc.zero_grad()
  L.lr_rnn_pair_errors()
  t0 = time.time()
  lp, hid, fin = enc(x, lens, max_len=T)
  torch.cuda.synchronize()
  print(mode, "forward %.1f ms, fault word %d" % ((time.time() - t0) * 1e3, L.lr_rnn_pair_errors()))
  t0 = time.time()
  (hid.pow(2).sum() + lp.sum() * 0.01 + sum(f.pow(2).sum() for f in fin)).backward()
  torch.cuda.synchronize()
  print(mode, "backward %.1f ms, fault word %d" % ((time.time() - t0) * 1e3, L.lr_rnn_pair_errors()))
  res[mode] = [hid.detach().cpu()] + [f.detach().cpu() for f in fin] + [p.grad.cpu().clone() for p in enc.parameters()]
names = ["hid", "h_n", "c_n"] + [k for k, _ in enc.named_parameters()]
for n, a, b in zip(names, res["f32"], res["split"]):
  print("%-28s rel %.3g  max|a| %.3g  max|d| %.3g" % (n, float((a - b).norm()) / max(1e-9, float(a.norm())), float(a.abs().max()),
                                                     float((a - b).abs().max())))
if os.environ.get("GRID_PER_STEP"):
  a, b = res["f32"][0], res["split"][0]
  for t in range(T):
    print("t=%d rel %.3g" % (t, float((a[:, t] - b[:, t]).norm()) / max(1e-9, float(a[:, t].norm()))))
