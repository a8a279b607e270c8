"""What the library GEMM (hipBLASLt / rocBLAS behind torch.matmul, bf16 in, fp32 accumulate) makes of the pixel regime's
projection shapes — the bar for lr_xgemm's products (tools/bench_xgemm.py).  One bf16 product each; random operands."""
import torch

dev = torch.device("cuda:0")
SHAPES = [("pixel fwd proj L0  x[2400,3456] W^T[3456,1536]", 2400, 1536, 3456),
          ("pixel dx L0        dG[2400,1536] W[1536,3456]", 2400, 3456, 1536),
          ("pixel dW_ih L0     dG^T[1536,2400] x[2400,3456]", 1536, 3456, 2400),
          ("pixel fwd proj L1  y[2400,512] W^T[512,1536]", 2400, 1536, 512),
          ("pixel dx L1        dG[2400,1536] W[1536,512]", 2400, 512, 1536)]
for name, M, N, K in SHAPES:
  A = torch.randn(M, K, device=dev).bfloat16()
  B = torch.randn(K, N, device=dev).bfloat16()
  for _ in range(5):
    C = A @ B
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20):
    C = A @ B
  e1.record()
  torch.cuda.synchronize()
  us = e0.elapsed_time(e1) * 1000 / 20
  print("%-52s %7.1f us  %6.0f TF/s (one bf16 product)" % (name, us, 2.0 * M * N * K / us / 1e6), flush=True)
