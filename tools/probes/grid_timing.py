"""Where a step of the grid recurrence (lr_rnn_grid.hip) spends its time: needs the LRG_TIMING variant library
(VARIANT_DEFS=-DLRG_TIMING tools/build_variant.sh gridtime lipreading_amd/csrc/lr_rnn_grid.hip; LIPREADING_HIP_LIB=.../alt/gridtime.so).
python tools/probes/grid_timing.py [B T]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lipreading_amd import _C  # noqa: E402
from lipreading_amd.data import default_char2idx  # noqa: E402
from lipreading_amd.encoder import VideoEncoder  # noqa: E402

B, T = (int(v) for v in (sys.argv[1:3] + ["32", "31"][len(sys.argv) - 1:]))
H = 1536
dev = torch.device("cuda:0")
L = _C.lib()
fn = ctypes.CDLL(os.environ["LIPREADING_HIP_LIB"]).lr_rnn_grid_debug_times
torch.manual_seed(1)
enc = VideoEncoder(64, H, rnn_type="LSTM", num_layers=1, bidirectional=False, enable_ctc=True, vocab_size=64,
                   char2idx=default_char2idx()).to(dev)
x = torch.randn(B, T, 64, 1, device=dev)
lens = torch.full((B,), T)
for _ in range(3):
  enc.zero_grad()
  lp, hid, fin = enc(x, lens, max_len=T)
  (hid.pow(2).sum() + lp.sum() * 0.01).backward()
torch.cuda.synchronize()
buf = np.zeros((2, 192, 64, 8), dtype=np.int64)
assert fn(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))) == 0
print("fault word", L.lr_rnn_pair_errors())
names = {0: ["h gather", "barrier", "product + publish", "partial gather", "barrier", "cell + publish h", "(loop edge)"],
         1: ["dh gather", "barrier", "cell + publish dG", "dG gather", "barrier", "product + publish", "(loop edge)"]}
for p, label in ((0, "forward"), (1, "backward")):
  t = buf[p][:, 1:T, :7].astype(np.float64)            # steps 1 .. T-1 (step 0 has no gather)
  per_step = (buf[p][:, 2:T, 0] - buf[p][:, 1:T - 1, 0]).astype(np.float64)
  total = (buf[p][:, T - 1, 6] - buf[p][:, 0, 0]).astype(np.float64)
  print("%s: clock ticks per step: mean %.0f (min member %.0f, max %.0f); whole loop %.0f ticks" %
        (label, per_step.mean(), per_step.mean(axis=1).min(), per_step.mean(axis=1).max(), total.mean()))
  d = np.diff(t, axis=2)                                # (members, steps, 6)
  edge = (buf[p][:, 2:T, 0] - buf[p][:, 1:T - 1, 6]).astype(np.float64)
  for k in range(6):
    print("   %-22s mean %7.0f   by member: min %7.0f max %7.0f   share %.2f" %
          (names[p][k], d[:, :, k].mean(), d[:, :, k].mean(axis=1).min(), d[:, :, k].mean(axis=1).max(), d[:, :, k].mean() / per_step.mean()))
  print("   %-22s mean %7.0f" % ("loop edge", edge.mean()))
