#!/usr/bin/env python
"""Split-bf16 GEMM (lr_xgemm: pack + contract) on the shapes the steps use; checks against an fp64 product.
   python tools/bench_xgemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lipreading_amd import _C  # noqa: E402

SHAPES = [  # name, transA, transB, M, N, K, a_exact, b_exact
    ("pixel fwd proj L0   x[2400,3456] W^T", 0, 1, 2400, 1536, 3456, 1, 0),
    ("pixel fwd proj L1   y[2400,512]  W^T", 0, 1, 2400, 1536, 512, 0, 0),
    ("pixel dW_ih L0      dG^T x", 1, 0, 1536, 3456, 2400, 0, 1),
    ("pixel dx L0         dG W (hi only)", 0, 0, 2400, 3456, 1536, 1, 1),
    ("lstm768 dW_hh       dG^T y", 1, 0, 3072, 768, 2400, 0, 0),
    ("lstm768 dW_ih       dG^T x", 1, 0, 6144, 204, 2400, 0, 0),
    ("tfm ff1             h[2400,256] W^T", 0, 1, 2400, 1024, 256, 0, 0),
]


def main():
  dev = torch.device("cuda:0")
  L = _C.lib()
  for name, ta, tb, M, N, K, ax, bx in SHAPES:
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    if ax:
      A = A.bfloat16().float()
    if bx:
      B = B.bfloat16().float()
    Ad, Bd = A.to(dev), B.to(dev)
    C = torch.empty(M, N, device=dev)
    wsb = L.lr_xgemm_workspace_bytes(ta, tb, M, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def run():
      _C.check(L.lr_xgemm(ta, tb, M, N, K, 1.0, Ad.data_ptr(), Ad.shape[1], Bd.data_ptr(), Bd.shape[1], 0.0, C.data_ptr(), N,
                          None, ax, bx, ws.data_ptr(), wsb, _C.stream_handle()), "lr_xgemm")
    for _ in range(3):
      run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
      run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    terms = 3 - ax - bx if not (ax and bx) else 1
    want = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    err = float((C.cpu().double() - want).norm() / want.norm())
    print("%-40s %4dx%4dx%4d terms %d: %7.1f us (pack + contract), %6.1f TF/s of bf16 MFMA work, rel err %.2e"
          % (name, M, N, K, terms, us, 2.0 * M * N * K * terms / us / 1e6, err), flush=True)


if __name__ == "__main__":
  main()
