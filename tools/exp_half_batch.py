#!/usr/bin/env python
"""TIMING-ONLY experiment: the pixel step with the batch cut in two halves whose forward and
backward passes run on TWO streams, so that one half's conv kernels can use the compute units the other half's
latency-bound recurrences leave idle.

  python tools/exp_half_batch.py [--steps 20]          (under rocprofv3 --kernel-trace for the timeline: anchor adam_kernel)

What it is NOT: a training step.  Both halves' backward passes add into the SAME gradient buffers (beta = 1
read-modify-write launches from two streams: a race), the two CTC launches share lr_ctc_nll_reduce's one completion
counter, and each half takes the reference's batch reduction over its own 16 samples.  The numbers the step leaves in
the weights are wrong; only the clock and the kernel timeline are read.  Three variants, same model, same inputs:
  full      forward + CTC + backward of the 32-sample batch, then clip + Adam (eager launches: the product's ctc_step body)
  serial    the two halves one after the other on one stream, then clip + Adam (what cutting the batch costs by itself)
  streams   the two halves on two streams — issue order fwd A, fwd B, bwd A, bwd B — then clip + Adam
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lipreading_amd  # noqa: E402,F401  (GPU_MAX_HW_QUEUES before the first HIP call)
from lipreading_amd import _C  # noqa: E402
from lipreading_amd.ctc import ctc_loss_with_status  # noqa: E402
from lipreading_amd.data import default_char2idx  # noqa: E402
from lipreading_amd.encoder import VideoEncoder  # noqa: E402
from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader, feature_dim  # noqa: E402
from lipreading_amd.optim import FlatParameters, FusedAdam  # noqa: E402

B, T, IMG, VOCAB, LABEL_LEN = 32, 75, 96, 64, 30


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--only", default="full,serial,streams")
  ap.add_argument("--graph", action="store_true", help="capture each variant's step as ONE hipGraph and time replays "
                  "(takes Python's launch time out of the comparison)")
  args = ap.parse_args()
  dev = torch.device("cuda:0")
  torch.manual_seed(123456)
  c2i = default_char2idx()
  enc = VideoEncoder(feature_dim(IMG, IMG), 256, rnn_type="GRU", num_layers=2, bidirectional=True, enable_ctc=True,
                     vocab_size=VOCAB, char2idx=c2i)
  model = PixelLipReader(enc, ConvFrontend3D()).to(dev).train()
  opt = FusedAdam(FlatParameters(model), lr=1e-4)
  g = torch.Generator().manual_seed(7)
  clips = torch.randint(0, 256, (B, T, 3, IMG, IMG), generator=g, dtype=torch.uint8).to(dev)
  lens = torch.full((B,), T, dtype=torch.int64, device=dev)
  labels = torch.randint(4, VOCAB, (B, LABEL_LEN), generator=g).to(dev)
  label_lens = torch.full((B,), LABEL_LEN, dtype=torch.int64, device=dev)
  one = torch.ones((), device=dev)
  halves = [slice(0, B // 2), slice(B // 2, B)]
  sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

  def fwd(sl):
    lp, _, _ = model(clips[sl], lens[sl], max_len=T, need_final_state=False)
    loss, status, _ = ctc_loss_with_status(lp, labels[sl], lens[sl], label_lens[sl], 'mean')
    return loss

  def full():
    opt.zero_grad()
    fwd(slice(0, B)).backward(one)
    opt.step(grad_norm=5.0, grad_scale=1.0)

  def serial():
    opt.zero_grad()
    for sl in halves:
      fwd(sl).backward(one)
    opt.step(grad_norm=5.0, grad_scale=1.0)

  def streams():
    opt.zero_grad()
    cur = torch.cuda.current_stream()
    losses = []
    for s, sl in zip((sA, sB), halves):
      s.wait_stream(cur)
      with torch.cuda.stream(s):
        losses.append(fwd(sl))
    for s, loss in zip((sA, sB), losses):
      with torch.cuda.stream(s):
        loss.backward(one)
    for s in (sA, sB):
      cur.wait_stream(s)
    opt.step(grad_norm=5.0, grad_scale=1.0)

  out = {}
  for name, fn in (("full", full), ("serial", serial), ("streams", streams)):
    if name not in args.only.split(","):
      continue
    for _ in range(3):
      fn()
    torch.cuda.synchronize()
    if args.graph:
      side = torch.cuda.Stream()
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):     # (capture wants the allocator warmed up on a side stream)
        fn()
      torch.cuda.current_stream().wait_stream(side)
      torch.cuda.synchronize()
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        fn()
      fn = graph.replay
      fn()
      torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(3):
      e0.record()
      for _ in range(args.steps):
        fn()
      e1.record()
      torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / args.steps
      best = ms if best is None else min(best, ms)
    out[name] = round(best, 4)
    print("%-8s %s %.4f ms per step (best of 3 x %d steps; recurrence time-outs so far: %d)"
          % (name, "graph" if args.graph else "eager", best, args.steps, _C.lib().lr_rnn_pair_errors()), flush=True)
  print(out)


if __name__ == "__main__":
  main()
