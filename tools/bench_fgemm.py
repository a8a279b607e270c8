#!/usr/bin/env python
"""Launch time of lr_fgemm at the transformer stage's shapes (HIP events over many launches): fixed cost per launch and
cost per 32-k stage.   python tools/bench_fgemm.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lipreading_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
L = _C.lib()


def run(form, M, N, K, prec=0, reps=50, relu=False, addend=False, mask=False):
  g = torch.Generator().manual_seed(1)
  if form == 2:
    A, B = torch.randn(K, M, generator=g).to(dev), torch.randn(K, N, generator=g).to(dev)
  elif form == 1:
    A, B = torch.randn(M, K, generator=g).to(dev), torch.randn(K, N, generator=g).to(dev)
  else:
    A, B = torch.randn(M, K, generator=g).to(dev), torch.randn(N, K, generator=g).to(dev)
  C = torch.empty(M, N, device=dev)
  bias = torch.randn(N, generator=g).to(dev)
  add = torch.randn(M, N, generator=g).to(dev)
  j = _C.FgemmJob()
  j.A, j.B, j.C, j.bias = A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr()
  j.addend = add.data_ptr() if addend else None
  j.mask = add.data_ptr() if mask else None
  j.colsum = j.slabs = None
  j.M, j.N, j.K, j.lda, j.ldb, j.ldc = M, N, K, A.stride(0), B.stride(0), N
  j.ldadd, j.add_period, j.ldmask = N, M, N
  j.flags, j.splits, j.alpha, j.beta = (1 if relu else 0), 1, 1.0, 0.0
  st = _C.stream_handle()
  for _ in range(5):
    _C.check(L.lr_fgemm(prec, form, 0, 0, ctypes.byref(j), 1, st))
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  e0.record()
  for _ in range(reps):
    _C.check(L.lr_fgemm(prec, form, 0, 0, ctypes.byref(j), 1, st))
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


for form, name in ((0, "NT"), (1, "NN"), (2, "TN")):
  for (M, N) in ((2400, 768), (2400, 256), (128, 128), (2400, 1024)):
    row = []
    for K in (32, 64, 256, 1024, 2400):
      row.append("K=%d %.1f" % (K, run(form, M, N, K)))
    print(name, M, N, " ".join(row), flush=True)
print("NT 2400x256 K=1024 addend", run(0, 2400, 256, 1024, addend=True), "mask", run(0, 2400, 256, 1024, mask=True))
print("F32 NT 2400x768 K=256", run(0, 2400, 768, 256, prec=1))
