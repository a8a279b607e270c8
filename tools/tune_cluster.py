#!/usr/bin/env python
"""Sweep of the cluster recurrence's polling knobs (lr_rnn_debug_tune) on one GPU box:
   python tools/tune_cluster.py          # LSTM-768 and GRU-256, B = 32, T = 75, bidirectional
Times encoder forward + backward (input projection, recurrence both ways, weight gradients) with events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lipreading_amd import _C  # noqa: E402
from lipreading_amd.data import default_char2idx  # noqa: E402
from lipreading_amd.encoder import VideoEncoder  # noqa: E402


def main():
  dev = torch.device("cuda:0")
  L = _C.lib()
  B, T = 32, 75
  x = torch.randn(B, T, 68, 3, device=dev)
  lens = torch.full((B,), T, device=dev, dtype=torch.int64)
  for rnn_type, H in (("LSTM", 768), ("GRU", 256), ("GRU", 800)):
    torch.manual_seed(1)
    enc = VideoEncoder(204, H, rnn_type=rnn_type, bidirectional=True, enable_ctc=True, vocab_size=64,
                       char2idx=default_char2idx()).to(dev).train()
    w = torch.randn(B, T, 65, device=dev)

    def run(n):
      for _ in range(n):
        enc.zero_grad()
        lp, _, _ = enc(x, lens, max_len=T, need_final_state=False)
        (lp * w).sum().backward()

    cfgs = ((0, 1, 0, 1), (0, 2, 0, 2), (3, 1, 3, 1), (0, 0, 0, 0))
    if len(sys.argv) > 1 and sys.argv[1] == "full":
      cfgs = ((0, 2, 0, 2), (3, 2, 3, 2), (6, 2, 6, 2), (9, 2, 9, 2), (12, 2, 12, 2), (16, 2, 16, 2), (0, 1, 0, 1),
              (0, 0, 0, 0), (6, 1, 6, 1), (6, 0, 6, 0), (9, 0, 9, 0), (0, 4, 0, 4))
    for cfg in cfgs:
      L.lr_rnn_debug_tune(0, cfg[0], cfg[1])
      L.lr_rnn_debug_tune(1, cfg[2], cfg[3])
      run(3)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      run(20)
      e1.record()
      torch.cuda.synchronize()
      print("%s-%d first-poll delay %2d, round sleep %d: %.1f us per fwd+bwd; faults %d"
            % (rnn_type, H, cfg[0], cfg[1], e0.elapsed_time(e1) * 1000 / 20, L.lr_rnn_pair_errors()), flush=True)
    L.lr_rnn_debug_tune(0, 0, 1)
    L.lr_rnn_debug_tune(1, 0, 1)


if __name__ == "__main__":
  main()
