"""Where a launch of the row-block encoder-layer kernel (lr_tfm_rowblock.hip, forward) spends its time: needs the
LR_RB_TIMING variant library (VARIANT_DEFS=-DLR_RB_TIMING tools/build_variant.sh rbtime lipreading_amd/csrc/lr_tfm_rowblock.hip;
LIPREADING_HIP_LIB=.../alt/rbtime.so).  Stamps are the 100 MHz wall clock (10 ns ticks) of workgroup 0's eight waves."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lipreading_amd.data import default_char2idx  # noqa: E402
from lipreading_amd.transformer import TransformerVideoEncoder  # noqa: E402

dev = torch.device("cuda:0")
fn = ctypes.CDLL(os.environ["LIPREADING_HIP_LIB"]).lr_tfm_rb_debug_times
torch.manual_seed(1)
enc = TransformerVideoEncoder(96, 256, 4, 1, 1024, enable_ctc=True, vocab_size=64, char2idx=default_char2idx()).to(dev)
enc.input_projection = 'bf16x3'
enc.attention = 'bf16'
x = torch.randn(32, 75, 96, device=dev)
lens = torch.full((32,), 75)
for _ in range(5):
  enc.zero_grad()
  lp, h, _ = enc(x, lens, max_len=75)
  (h.pow(2).sum() + lp.sum() * 0.01).backward()
torch.cuda.synchronize()
buf = np.zeros((2, 8, 32), dtype=np.int64)
assert fn(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))) == 0
names = ["params + ring issue + rows->planes", "barrier", "product Wo", "epilogue s1", "barrier", "LN1", "barrier"]
for c in range(4):
  names += ["product W1 c%d" % c, "epilogue f1 c%d" % c, "barrier", "product W2 c%d" % c]
names += ["epilogue s2", "barrier", "LN2"]
t = buf[0].astype(np.float64) * 10.0   # ns
print("forward launch, workgroup 0: %.2f us from the first stamp to the last (mean over waves)" % ((t[:, 26] - t[:, 0]).mean() / 1e3))
d = np.diff(t[:, :27], axis=1)
for k, n in enumerate(names):
  print("  %-36s mean %7.0f ns   min wave %7.0f  max wave %7.0f" % (n, d[:, k].mean(), d[:, k].min(), d[:, k].max()))
