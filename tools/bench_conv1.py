#!/usr/bin/env python
"""Stand-alone timing of the frontend's first layer at the metric's batch (32 x 75 raw uint8 frames of 96 x 96): forward
with the fused ReLU + max-pool epilogue (3 -> 32 channels, taps 3x5x5, stride 2) out of the raw clip, and the weight
gradient taken from the pooled gradient; HIP events around
each call, median / min over `reps` calls, and a checksum of the outputs so that two builds (LIPREADING_HIP_LIB) can be
compared bit for bit.  usage: python tools/bench_conv1.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lipreading_amd import _C


def main(reps):
  L = _C.lib()
  dev = torch.device("cuda:0")
  st = _C.stream_handle()
  bf = torch.bfloat16
  B, T, h, w, cin, cin_p, cout = 32, 75, 96, 96, 3, 4, 32
  kt, kh, kw, pt, ph, pw = 3, 5, 5, 1, 2, 2
  g = torch.Generator().manual_seed(0)
  clips = torch.randint(0, 256, (B * T, 3, h, w), generator=g, dtype=torch.uint8).to(dev)
  weight = (torch.randn(cout, cin, kt, kh, kw, generator=g) * 0.05).to(dev)
  bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
  wp = torch.empty((cout, kt * kh * kw, cin_p), dtype=bf, device=dev)
  _C.check(L.lr_conv3d_pack_weights(weight.data_ptr(), wp.data_ptr(), cout, cin, cin_p, kt, kh, kw, 0, st))
  F = B * T
  ho = h // 2
  pooled = torch.empty((F, ho // 2, ho // 2, cout), dtype=bf, device=dev)
  code = torch.empty(pooled.shape, dtype=torch.uint8, device=dev)
  flops = 2.0 * F * ho * ho * cout * cin * kt * kh * kw

  def fwd():
    _C.check(L.lr_conv3d_forward_pooled(clips.data_ptr(), wp.data_ptr(), bias.data_ptr(), pooled.data_ptr(), code.data_ptr(),
                                        B, T, h, w, cin_p, cout, kt, kh, kw, 2, pt, ph, pw, 1 | 8, st))

  # weight + bias gradient straight from the pooled gradient (lr_conv3d_wgrad_pooled: kernel + slab reduction)
  dP = (torch.randn(pooled.shape, generator=g) * 0.1).to(dev).bfloat16()
  wb = L.lr_conv3d_wgrad_workspace_bytes(cout, cin_p, kt, kh, kw)
  ws = torch.empty(wb, dtype=torch.uint8, device=dev)
  dw = torch.empty((cout, cin, kt, kh, kw), device=dev)
  db = torch.empty(cout, device=dev)

  def wgrad():
    _C.check(L.lr_conv3d_wgrad_pooled(clips.data_ptr(), pooled.data_ptr(), code.data_ptr(), dP.data_ptr(), dw.data_ptr(),
                                      db.data_ptr(), ws.data_ptr(), wb, 0, B, T, h, w, cin_p, cin, cout, kt, kh, kw, 2, pt,
                                      ph, pw, 1, st))

  for name, fn in (("layer1 forward+pool (u8)", fwd), ("layer1 weight gradient from the pooled one (u8)", wgrad)):
    times = []
    for rep in range(reps + 3):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      fn()
      e1.record()
      torch.cuda.synchronize()
      if rep >= 3:
        times.append(e0.elapsed_time(e1) * 1e3)
    times.sort()
    med = times[len(times) // 2]
    print("%s: median %.1f us  min %.1f us  (%.0f TFLOP/s at the median)" % (name, med, times[0], flops / med / 1e6))
  print("checksum pooled %.6f code %d dW %.9e |dW| %.9e db %.9e" % (
    float(pooled.float().double().sum()), int(code.long().sum()), float(dw.double().sum()), float(dw.double().abs().sum()),
    float(db.double().sum())))


if __name__ == "__main__":
  main(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
