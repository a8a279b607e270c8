#!/usr/bin/env python
"""Stand-alone timing of the frontend's layer-2 kernels at the metric's batch (32 x 75 frames of 24 x 24): forward with
the fused ReLU + max-pool epilogue (32 -> 64 channels, taps 3x5x5) and the data gradient (64 -> 32); HIP events
around each call, median / min over `reps` calls.  usage: python tools/bench_conv_patch.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lipreading_amd import _C


def main(reps, zeros=False):
  L = _C.lib()
  dev = torch.device("cuda:0")
  st = _C.stream_handle()
  bf = torch.bfloat16
  B, T, h, w, cin, cout = 32, 75, 24, 24, 32, 64
  kt, kh, kw, pt, ph, pw = 3, 5, 5, 1, 2, 2
  x = (torch.randn(B * T, h, w, cin, device=dev) * 0.5).clamp_min(0).to(bf)
  dz = (torch.randn(B * T, h, w, cout, device=dev) * 0.1).to(bf)
  dz = dz * (torch.rand(B * T, h, w, cout, device=dev) < 0.25)   # max-pool backward: one in four is non-zero
  weight = torch.randn(cout, cin, kt, kh, kw, device=dev) * 0.02
  bias = torch.randn(cout, device=dev) * 0.1
  if zeros:   # the same instruction stream on all-zero operands: what the chip's power budget takes from the dense kernels
    x.zero_(); dz.zero_(); weight.zero_(); bias.zero_()
  wp = torch.empty((cout, kt * kh * kw, cin), dtype=bf, device=dev)
  wd = torch.empty((cin, kt * kh * kw, cout), dtype=bf, device=dev)
  _C.check(L.lr_conv3d_pack_weights(weight.data_ptr(), wp.data_ptr(), cout, cin, cin, kt, kh, kw, 2, st))
  _C.check(L.lr_conv3d_pack_weights(weight.data_ptr(), wd.data_ptr(), cout, cin, cin, kt, kh, kw, 1 | 2, st))
  F = B * T
  pooled = torch.empty((F, h // 2, w // 2, cout), dtype=bf, device=dev)
  code = torch.empty(pooled.shape, dtype=torch.uint8, device=dev)
  dx = torch.empty((F, h, w, cin), dtype=bf, device=dev)
  flops = 2.0 * F * h * w * cout * cin * kt * kh * kw

  def fwd():
    _C.check(L.lr_conv3d_forward_pooled(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), pooled.data_ptr(), code.data_ptr(),
                                        B, T, h, w, cin, cout, kt, kh, kw, 1, pt, ph, pw, 1 | 2, st))

  def dgrad():
    _C.check(L.lr_conv3d_forward(dz.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), B, T, h, w, cout, cin, kt, kh, kw, 1,
                                 pt, ph, pw, 2, st))

  for name, fn in (("layer2 forward+pool", fwd), ("layer2 data gradient", dgrad)):
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
    print("%s%s: median %.1f us  min %.1f us  (%.0f TFLOP/s at the median)" % (name, " [zero operands]" if zeros else "", med, times[0], flops / med / 1e6))


if __name__ == "__main__":
  main(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
  if len(sys.argv) > 2 and sys.argv[2] == "zeros":
    main(int(sys.argv[1]), zeros=True)
