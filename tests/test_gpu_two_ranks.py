"""Two ranks, the REAL HIP step: shard -> lipreading_amd.train.ctc_step -> GradSync -> clip after the reduce -> Adam.

Only one GPU is available to this build, and gloo accepts device tensors: two processes share cuda:0, exchange
over gloo, and their result is compared with a single process that steps on the full batch (the data-parallel
contract of SURVEY.md section 8e; the reference is single-process, src/scripts/train.py:200-203).  What this cannot
show is a scaling curve — two ranks time-share one GPU here."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
  assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
  return torch.device("cuda:0")


def run_ranks(scenario, out_dir, world=2):
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
  env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_worker.py"), scenario, str(r),
                             str(world), port, str(out_dir)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                            text=True) for r in range(world)]
  outs = []
  try:
    for p in procs:
      outs.append(p.communicate(timeout=600)[0])
  finally:
    for p in procs:          # by PID: a rank left waiting in a collective must not outlive the test
      if p.poll() is None:
        p.kill()
  for r, p in enumerate(procs):
    assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
  return [np.load(os.path.join(str(out_dir), "rank%d.npz" % r)) for r in range(world)]


def twin(kind, dev, steps, lo=0, hi=None, grad_scale=1.0):
  """The single-process step on rows [lo, hi) of the same batch from the same initial weights."""
  from lipreading_amd.optim import FlatParameters, FusedAdam
  from lipreading_amd.ctc import ctc_loss_with_status
  from tests import two_rank_worker as W
  model, _ = W.build(kind, dev)
  flat = FlatParameters(model)
  init = flat.data.detach().cpu().numpy().copy()
  opt = FusedAdam(flat, lr=1e-3)
  x, lens, chars, char_lens = W.batch(kind, dev, lo, W.B_FULL if hi is None else hi)
  losses = []
  for _ in range(steps):
    opt.zero_grad()
    lp, _, _ = model(x, lens, max_len=W.T, need_final_state=False)
    loss, status, _ = ctc_loss_with_status(lp, chars[:, 1:], lens, char_lens - 1, 'mean')
    loss.backward()
    opt.step(grad_norm=50, grad_scale=grad_scale, skip=status)
    losses.append(float(loss))
  torch.cuda.synchronize()
  return flat.data.detach().cpu().numpy(), flat.grad.detach().cpu().numpy(), losses, float(opt.total_norm()), init


def updates_agree(got, want, init, rel):
  """Adam divides by sqrt(v): where a gradient entry is within rounding of zero, a different summation order (shard
  sums + all-reduce vs one full-batch sum) moves that entry's update by a visible fraction of lr, so entries are
  compared on the UPDATE as a whole: |got - want| small against |want - init|."""
  upd = float(np.linalg.norm(want - init))
  assert upd > 0
  assert float(np.linalg.norm(got - want)) <= rel * upd, (float(np.linalg.norm(got - want)), upd)


@pytest.mark.parametrize("scenario", ["landmarks_eager", "landmarks_hold", "landmarks_graph", "pixels_eager", "pixels_hold"])
def test_two_ranks_equal_the_full_batch_step(dev, tmp_path, scenario):
  """eager: buckets all-reduced from the gradient-ready hooks while backward is still running; hold: every
  bucket exchanged at sync(); graph: forward+backward replayed from a hipGraph (StepGraphs holds the exchange back:
  a collective is never captured), the exchange and the optimiser after it.  Every variant: both ranks end with
  IDENTICAL weights, equal to the full-batch step's; the mean of the shard losses is the full-batch loss; the summed
  gradient / world is the full-batch gradient, so the clip (computed after the reduce, train_better_model.py:78) sees
  the same norm."""
  from tests import two_rank_worker as W
  kind, mode = scenario.split("_")
  ranks = run_ranks(scenario, tmp_path)
  steps = W.STEPS[mode]
  data, grad, losses, norm, init = twin(kind, dev, steps)
  np.testing.assert_array_equal(ranks[0]["data"], ranks[1]["data"])       # the ranks never diverge
  assert int(ranks[0]["faults"]) == 0 and int(ranks[1]["faults"]) == 0
  assert list(ranks[0]["statuses"]) == [0] * steps and int(ranks[0]["steps"]) == steps
  if mode == "graph":
    assert int(ranks[0]["captures"]) == 1 and int(ranks[0]["replays"]) >= 1
  # bf16 conv stack: shard and full-batch runs round the same values at the same points — only the fp32 weight-gradient
  # sums differ in order
  tol = 1e-6 if kind == "landmarks" else 2e-5
  mean_loss = (ranks[0]["losses"] + ranks[1]["losses"]) / 2
  np.testing.assert_allclose(mean_loss, np.array(losses), rtol=10 * tol, atol=1e-6)
  scale = float(np.abs(grad).max())
  assert np.abs(ranks[0]["grad"] / 2 - grad).max() <= 20 * tol * scale      # last step's reduced gradient / world
  assert abs(float(ranks[0]["norm"]) / 2 - norm) <= 20 * tol * norm          # (sumsq is taken on the SUM: norm x world)
  updates_agree(ranks[0]["data"], data, init, 2e-3 if kind == "landmarks" else 2e-2)


def test_a_rank_whose_batch_is_skipped_contributes_nothing(dev, tmp_path):
  """Rank 1's shard holds only captions longer than their clips: its CTC loss is the reference's `None` (status 1,
  zero gradient).  The step is skipped only if EVERY rank skipped (MIN over ranks): both ranks update with rank 0's
  gradient / world — identical weights, equal to a single process stepping on rank 0's shard with that scale."""
  from tests import two_rank_worker as W
  ranks = run_ranks("landmarks_skip", tmp_path)
  np.testing.assert_array_equal(ranks[0]["data"], ranks[1]["data"])
  assert list(ranks[0]["statuses"]) == [0, 0] and list(ranks[1]["statuses"]) == [0, 0]   # the ranks' verdict (lr_clip_adam_step)
  assert float(ranks[1]["losses"][0]) == 0.0 and float(ranks[0]["losses"][0]) > 0
  assert int(ranks[0]["steps"]) == 2 and int(ranks[1]["steps"]) == 2
  data, grad, losses, _, init = twin("landmarks", dev, 2, 0, W.B_FULL // 2, grad_scale=0.5)
  np.testing.assert_allclose(ranks[0]["losses"], np.array(losses), rtol=1e-5, atol=1e-6)
  # (the two words in front of the gradients travelled with them: one rank skipped its batch, nobody timed out)
  assert float(ranks[0]["grad"][0]) == 1.0 and float(ranks[0]["grad"][1]) == 0.0 and float(grad[0]) == 0.0
  assert np.abs(ranks[0]["grad"][64:] - grad[64:]).max() <= 2e-5 * float(np.abs(grad).max())
  updates_agree(ranks[0]["data"], data, init, 2e-3)
