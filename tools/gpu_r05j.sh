#!/usr/bin/env bash
# round 5 A/B in one visit (in-tree first and last): the stored-bf16 first layer's dW_ih through lr_fgemm (RNN_DEBUG=8),
# the upper layer's weight half behind the lower layer's recurrence (DEFER_UPPER=1), both
set -u
OUT=gpurun_out; mkdir -p $OUT
line() {   # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 python bench.py --regime pixels --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$tag', d['ms_per_step'], d['timing']['ms_per_step_min'], 'loss', d['final_loss'])"
}
timeout 600 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_encoder.py -q -x -k "pixel or oracle or bf16x3" 2>&1 | tail -2
timeout 300 python - <<'PY'
import torch
from lipreading_amd import _C
from lipreading_amd.ctc import ctc_loss_with_status
from lipreading_amd.data import default_char2idx
from lipreading_amd.encoder import VideoEncoder
from lipreading_amd import encoder as E
from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader, feature_dim
from lipreading_amd.optim import FlatParameters
dev = torch.device("cuda:0")
torch.manual_seed(3)
enc = VideoEncoder(feature_dim(96, 96), 256, rnn_type='GRU', num_layers=2, bidirectional=True, enable_ctc=True, vocab_size=64, char2idx=default_char2idx())
model = PixelLipReader(enc, ConvFrontend3D()).to(dev).train()
flat = FlatParameters(model)
g = torch.Generator().manual_seed(4)
B, T = 8, 75
clips = torch.randint(0, 256, (B, T, 3, 96, 96), generator=g, dtype=torch.uint8).to(dev)
lens = torch.full((B,), T, device=dev)
labels = torch.randint(4, 64, (B, 12), generator=g).to(dev); ll = torch.full((B,), 12, device=dev)
res = {}
for bit in (0, 8):
  _C.lib().lr_rnn_debug_disable_cluster(bit)
  flat.zero_grad()
  lp, _, _ = model(clips, lens, max_len=T)
  loss, st, _ = ctc_loss_with_status(lp, labels, lens, ll, 'mean')
  loss.backward(); E.flush_deferred(); torch.cuda.synchronize()
  res[bit] = flat.grad.clone()
_C.lib().lr_rnn_debug_disable_cluster(0)
d = float((res[0] - res[8]).norm()) / float(res[0].norm())
w = enc.rnn.weight_ih_l0.grad
print("dW_ih through lr_fgemm vs lr_xgemm: relative difference of the whole gradient %.3g" % d)
assert d < 2e-5
PY
line default A=1
line dwih_fgemm LIPREADING_RNN_DEBUG=8
line defer_upper LIPREADING_DEFER_UPPER=1
line both LIPREADING_RNN_DEBUG=8 LIPREADING_DEFER_UPPER=1
line default_again A=1
