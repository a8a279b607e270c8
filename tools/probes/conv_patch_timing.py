"""Where a tile of the layer-2 patch kernel spends its cycles (fill / taps / epilogue): needs the LRP_TIMING variant
(VARIANT_DEFS=-DLRP_TIMING tools/build_variant.sh patchtime lipreading_amd/csrc/lr_conv_patch.hip) and
LIPREADING_HIP_LIB pointing at it."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lipreading_amd import _C  # noqa: E402

L = _C.lib()
fn = ctypes.CDLL(os.environ["LIPREADING_HIP_LIB"]).lr_conv_patch_debug_times
dev = torch.device("cuda:0")
st = _C.stream_handle()
bf = torch.bfloat16
B, T, h, w, cin, cout = 32, 75, 24, 24, 32, 64
kt, kh, kw, pt, ph, pw = 3, 5, 5, 1, 2, 2
x = (torch.randn(B * T, h, w, cin, device=dev) * 0.5).clamp_min(0).to(bf)
weight = torch.randn(cout, cin, kt, kh, kw, device=dev) * 0.02
bias = torch.randn(cout, device=dev) * 0.1
wp = torch.empty((cout, kt * kh * kw, cin), dtype=bf, device=dev)
wd = torch.empty((cin, kt * kh * kw, cout), dtype=bf, device=dev)
_C.check(L.lr_conv3d_pack_weights(weight.data_ptr(), wp.data_ptr(), cout, cin, cin, kt, kh, kw, 2, st))
_C.check(L.lr_conv3d_pack_weights(weight.data_ptr(), wd.data_ptr(), cout, cin, cin, kt, kh, kw, 1 | 2, st))
F = B * T
pooled = torch.empty((F, h // 2, w // 2, cout), dtype=bf, device=dev)
code = torch.empty(pooled.shape, dtype=torch.uint8, device=dev)
dp = (torch.randn(F, h // 2, w // 2, cout, device=dev) * 0.1).to(bf)
dx = torch.empty((F, h, w, cin), dtype=bf, device=dev)


def fwd():
  _C.check(L.lr_conv3d_forward_pooled(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), pooled.data_ptr(), code.data_ptr(),
                                      B, T, h, w, cin, cout, kt, kh, kw, 1, pt, ph, pw, 1 | 2, st))


def dgrad():
  _C.check(L.lr_conv3d_dgrad_pooled(dp.data_ptr(), code.data_ptr(), wd.data_ptr(), dx.data_ptr(), B, T, h, w, cout, cin,
                                    kt, kh, kw, pt, ph, pw, st))


buf = np.zeros((4096, 8), dtype=np.int64)
for name, f in (("forward+pool", fwd), ("data gradient (unpool)", dgrad)):
  try:
    f()
  except Exception as e:   # (the entry point's name differs: fall back to what bench_conv_patch.py calls)
    print(name, "skipped:", e)
    continue
  fn(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), 1)
  for _ in range(3):
    f()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  fn(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), 1)
  e0.record()
  f()
  e1.record()
  torch.cuda.synchronize()
  fn(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), 1)
  full = buf[buf[:, 4] > 0]
  n = full[:, 4].sum()
  tot = full[:, 3].sum() / n
  print("%s: %.1f us; %d tiles stamped; per tile: fill %.0f, taps %.0f, epilogue %.0f, whole %.0f ticks (shares %.2f / %.2f / %.2f)"
        % (name, e0.elapsed_time(e1) * 1e3, n, full[:, 0].sum() / n, full[:, 1].sum() / n, full[:, 2].sum() / n, tot,
           full[:, 0].sum() / n / tot, full[:, 1].sum() / n / tot, full[:, 2].sum() / n / tot))
