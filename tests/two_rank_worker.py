"""One rank of tests/test_gpu_two_ranks.py: two processes share cuda:0, talk over gloo and run the PRODUCT's
optimisation step (lipreading_amd.train.ctc_step) on their shard of the batch, with the gradient exchange of
lipreading_amd.distributed.GradSync between backward and the optimiser.

  python tests/two_rank_worker.py <scenario> <rank> <world> <port> <out_dir>

Also imported by the test itself for `build()` / `batch()` (the single-process full-batch twin)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

B_FULL, T, L = 8, 12, 5
STEPS = {"eager": 2, "hold": 2, "graph": 4, "skip": 2}


def build(kind, dev):
  """Same seed -> same initial weights on every rank and in the twin."""
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  torch.manual_seed(77)
  if kind == "pixels":
    from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader, feature_dim
    enc = VideoEncoder(feature_dim(32, 32), 16, rnn_type='GRU', num_layers=2, bidirectional=True, enable_ctc=True,
                       vocab_size=64, char2idx=default_char2idx())
    return PixelLipReader(enc, ConvFrontend3D()).to(dev).train(), enc
  # BiGRU-256: the one-launch cluster recurrence (the default path of the reference-faithful regime)
  enc = VideoEncoder(204, 256, rnn_type='GRU', num_layers=1, bidirectional=True, enable_ctc=True, vocab_size=64,
                     char2idx=default_char2idx())
  return enc.to(dev).train(), enc


def batch(kind, dev, lo=0, hi=B_FULL, broken_from=None):
  """Rows [lo, hi) of the full batch.  broken_from: rows >= it get captions longer than their clips — ctc_loss
  then skips them (inf), and a shard made only of such rows is a skipped batch (status 1)."""
  g = torch.Generator().manual_seed(78)
  if kind == "pixels":
    x = torch.randint(0, 256, (B_FULL, T, 3, 32, 32), generator=g, dtype=torch.uint8)
  else:
    x = torch.randn(B_FULL, T, 68, 3, generator=g)
  lens = torch.full((B_FULL,), T, dtype=torch.long)
  width = L + 2 if broken_from is None else T + 4
  chars = torch.zeros(B_FULL, width, dtype=torch.long)
  char_lens = torch.full((B_FULL,), L + 2, dtype=torch.long)
  body = torch.randint(4, 64, (B_FULL, T + 4), generator=g)     # (same labels whether or not some rows are broken)
  for b in range(B_FULL):
    n = L + 2 if (broken_from is None or b < broken_from) else T + 4
    chars[b, 0], chars[b, 1:n - 1], chars[b, n - 1] = 1, body[b, 1:n - 1], 2
    char_lens[b] = n
  return tuple(t[lo:hi].to(dev) for t in (x, lens, chars, char_lens))


def main():
  scenario, rank, world, port, out_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
  kind, mode = scenario.split("_")
  import torch.distributed as dist
  from lipreading_amd import train as T_
  from lipreading_amd.distributed import GradSync, shard_batch
  from lipreading_amd.optim import FlatParameters, FusedAdam
  os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
  dev = torch.device("cuda:0")          # both ranks on the one GPU
  torch.cuda.set_device(dev)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    model, enc = build(kind, dev)
    if rank == 1:                       # ranks start from different weights: the broadcast fixes it
      with torch.no_grad():
        for p in model.parameters():
          p.add_(0.25)
    flat = FlatParameters(model)
    opt = FusedAdam(flat, lr=1e-3)
    # pixels: [conv], [encoder] — the encoder's bucket goes out behind the step's last recurrence
    groups = GradSync.groups_for_pixel_model(model, flat) if kind == "pixels" else GradSync.groups_for_encoder(enc, flat)
    sync = GradSync(flat, groups=groups, overlap=True)
    sync.broadcast_parameters(0)
    lo, hi = shard_batch(B_FULL, rank, world)
    x, lens, chars, char_lens = batch(kind, dev, lo, hi, broken_from=(B_FULL // 2 if mode == "skip" else None))
    graphs = T_.StepGraphs() if mode == "graph" else None
    losses, statuses = [], []
    for _ in range(STEPS[mode]):
      if mode == "hold":
        with sync.hold():                # every bucket goes out at sync() instead of from the hooks
          loss, status = T_.ctc_step(model, opt, x, lens, chars, char_lens, grad_norm=50, max_len=T, grad_sync=sync)
      else:
        loss, status = T_.ctc_step(model, opt, x, lens, chars, char_lens, grad_norm=50, max_len=T, grad_sync=sync,
                                   graphs=graphs)
      losses.append(float(loss))
      statuses.append(int(status))
    torch.cuda.synchronize()
    from lipreading_amd import _C
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), data=flat.data.detach().cpu().numpy(),
             grad=flat.grad.detach().cpu().numpy(), losses=np.array(losses), statuses=np.array(statuses),
             norm=float(opt.total_norm()), steps=int(opt.step_count[0]), skipped=int(opt.step_count[1]),
             captures=(graphs.captures if graphs else 0), replays=(graphs.replays if graphs else 0),
             faults=int(_C.lib().lr_rnn_pair_errors()))
    sync.close()
  finally:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
