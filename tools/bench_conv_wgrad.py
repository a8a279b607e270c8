#!/usr/bin/env python
"""Stand-alone timing of lr_conv3d_wgrad on the frontend's stride-1 layers at the metric's batch (32 x 75 frames):
layer 2 (24 x 24, 32 -> 64, taps 3x5x5) and layer 3 (12 x 12, 64 -> 96, taps 3x3x3); HIP events around each call
(kernel + slab reduction), median / min over `reps` calls.  usage: python tools/bench_conv_wgrad.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lipreading_amd import _C


def main(reps):
  L = _C.lib()
  dev = torch.device("cuda:0")
  st = _C.stream_handle()
  B, T = 32, 75
  for name, hw, cin, cout, k in (("layer2", 24, 32, 64, (3, 5, 5)), ("layer3", 12, 64, 96, (3, 3, 3))):
    x = torch.randn(B * T, hw, hw, cin, device=dev).clamp_min(0).bfloat16()
    dz = (torch.randn(B * T, hw, hw, cout, device=dev) * 0.1).bfloat16()
    wb = L.lr_conv3d_wgrad_workspace_bytes(cout, cin, *k)
    ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    dw = torch.empty((cout, cin) + k, device=dev)
    pads = (1, k[1] // 2, k[2] // 2)
    times = []
    for rep in range(reps + 3):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      _C.check(L.lr_conv3d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), wb, 0, B, T, hw, hw,
                                 cin, cin, cout, *k, 1, *pads, st))
      e1.record()
      torch.cuda.synchronize()
      if rep >= 3:
        times.append(e0.elapsed_time(e1) * 1e3)
    times.sort()
    flops = 2.0 * B * T * hw * hw * cout * cin * k[0] * k[1] * k[2]
    med = times[len(times) // 2]
    print("%s: median %.1f us  min %.1f us  (%.0f TFLOP/s at the median, incl. the slab reduction)" %
          (name, med, times[0], flops / med / 1e6))


if __name__ == "__main__":
  main(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
