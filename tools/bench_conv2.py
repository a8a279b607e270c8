#!/usr/bin/env python
"""Micro-benchmark of the second conv layer's kernels at the bench shape (B=32, T=75, 24x24, 32 -> 64 channels,
taps (3,5,5)): forward (+ReLU+pool epilogue), data gradient, weight gradient — through the C ABI, timed with
events on the launch stream, with a checksum of every output so that two builds can be compared bit for bit.
  usage:  python tools/bench_conv2.py [--reps 20] [--B 32]
"""
import argparse
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lipreading_amd import _C  # noqa: E402


def digest(t):
  return hashlib.sha1(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]


def timed(fn, reps):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--reps", type=int, default=20)
  ap.add_argument("--B", type=int, default=32)
  ap.add_argument("--lib", default=None, help="a variant library (tools/build_variant.sh) instead of the in-tree build")
  args = ap.parse_args()
  if args.lib:
    _C.lib_path = lambda: os.path.abspath(args.lib)
  L = _C.lib()
  st = _C.stream_handle()
  dev = torch.device("cuda", 0)
  bf = torch.bfloat16
  B, T, h, w, cin, cout = args.B, 75, 24, 24, 32, 64
  kt, kh, kw, pt, ph, pw = 3, 5, 5, 1, 2, 2
  frames = B * T
  g = torch.Generator().manual_seed(11)
  x = (torch.randn(frames, h, w, cin, generator=g) * 0.5).clamp_min(0).to(bf).to(dev)
  weight = (torch.randn(cout, cin, kt, kh, kw, generator=g) * 0.02).to(dev)
  bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
  out = {}

  # forward (+ ReLU + 2x2 max-pool)
  frag = L.lr_conv3d_patch_supported(h, w, cin, cout, kt, kh, kw, 1, pt, ph, pw)
  wp = torch.empty((cout, kt * kh * kw, cin), dtype=bf, device=dev)
  _C.check(L.lr_conv3d_pack_weights(weight.data_ptr(), wp.data_ptr(), cout, cin, cin, kt, kh, kw, frag, st), "pack")
  pooled = torch.empty((frames, h // 2, w // 2, cout), dtype=bf, device=dev)
  code = torch.empty(pooled.shape, dtype=torch.uint8, device=dev)

  def fwd():
    _C.check(L.lr_conv3d_forward_pooled(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), pooled.data_ptr(), code.data_ptr(),
                                        B, T, h, w, cin, cout, kt, kh, kw, 1, pt, ph, pw, 1 | frag, st), "fwd")
  out["fwd_us"] = round(timed(fwd, args.reps), 1)
  out["fwd_sha"] = digest(pooled) + "/" + digest(code)

  # data gradient: the forward kernel on dZ (64 channels) with flipped, transposed weights
  dZ = (torch.randn(frames, h, w, cout, generator=g) * 0.1).to(bf).to(dev)
  fragd = L.lr_conv3d_patch_supported(h, w, cout, cin, kt, kh, kw, 1, pt, ph, pw)
  wd = torch.empty((cin, kt * kh * kw, cout), dtype=bf, device=dev)
  _C.check(L.lr_conv3d_pack_weights(weight.data_ptr(), wd.data_ptr(), cout, cin, cin, kt, kh, kw, 1 | fragd, st), "pack t")
  dX = torch.empty((frames, h, w, cin), dtype=bf, device=dev)

  def dgrad():
    _C.check(L.lr_conv3d_forward(dZ.data_ptr(), wd.data_ptr(), None, dX.data_ptr(), B, T, h, w, cout, cin, kt, kh, kw,
                                 1, pt, ph, pw, fragd, st), "dgrad")
  out["dgrad_us"] = round(timed(dgrad, args.reps), 1)
  out["dgrad_sha"] = digest(dX)

  # weight gradient
  wbytes = L.lr_conv3d_wgrad_workspace_bytes(cout, cin, kt, kh, kw)
  ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
  dW = torch.empty_like(weight)

  def wgrad():
    _C.check(L.lr_conv3d_wgrad(x.data_ptr(), dZ.data_ptr(), dW.data_ptr(), None, ws.data_ptr(), wbytes, 0, B, T, h, w,
                               cin, cin, cout, kt, kh, kw, 1, pt, ph, pw, st), "wgrad")
  out["wgrad_us"] = round(timed(wgrad, args.reps), 1)
  out["wgrad_sha"] = digest(dW)
  flops = 2.0 * frames * h * w * cout * cin * kt * kh * kw
  for k in ("fwd", "dgrad", "wgrad"):
    out[k + "_tflops"] = round(flops / (out[k + "_us"] * 1e-6) / 1e12, 1)
  out["lib"] = args.lib or "in-tree"
  print(json.dumps(out))


if __name__ == "__main__":
  main()
