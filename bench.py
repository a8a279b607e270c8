#!/usr/bin/env python
"""Benchmark of the video->characters hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--model gru256|lstm768] [--batch B]

One "step" = one pass of the hot path over one batch of synthetic landmark clips already resident
in HBM: VideoEncoder forward (input-projection GEMM, T-step recurrent chain, output projection,
masked log-softmax) -> CTC 'mean' loss -> backward -> clip_grad_norm_(50) -> Adam(1e-4) step —
the encoder+CTC part of the reference step (src/train/train_better_model.py:46-48,74,78,80).
Rank 0 prints ONE JSON line (contract in the task statement): value = whole-job frames/s.

Regime: (R) reference-faithful input — landmarks (B,75,68,3) f32 — the only regime with a
reference oracle (the reference has no pixel path and no conv
frontend; the (B,75,3,96,96) pixel regime of BASELINE.json is build-defined and not built yet).

Extra objects on the JSON line:
  roofline     for the dominant kernel (the recurrent step kernel): algorithmic bytes per launch
               = W_hh re-streamed once per step for both directions, D*G*H*H*4 B,
               / average launch duration measured live with hipEvent pairs on the launch stream
               (lr_profile_enable / lr_profile_read).
  cpu_baseline the oracle (oracle/torch_oracle.py: the reference's own op sequence on stock torch
               CPU ops) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

MODELS = {
    # name: (rnn_type, hidden, layers, bidirectional)
    "gru256": ("GRU", 256, 1, True),     # LipNet-style BiGRU-256 (BASELINE configs[1] encoder)
    "lstm768": ("LSTM", 768, 1, True),   # config/archive/experiments/ecd/* shape (configs[2])
}
T_FRAMES, N_LMK, LMK_DIM, VOCAB, LABEL_LEN = 75, 68, 3, 64, 30
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3


def synth_batch(B, seed, device=None):
  """Synthetic batch of SURVEY.md 8d: frames ~ N(0,1) (B,75,68,3), all lengths 75, labels L=30
  uniform in [4,64) + EOS, framed with BOS."""
  import torch
  g = torch.Generator().manual_seed(seed)
  frames = torch.randn(B, T_FRAMES, N_LMK, LMK_DIM, generator=g)
  frame_lens = torch.full((B,), T_FRAMES, dtype=torch.long)
  chars = torch.zeros(B, LABEL_LEN + 2, dtype=torch.long)
  chars[:, 0] = 1
  chars[:, 1:LABEL_LEN + 1] = torch.randint(4, VOCAB, (B, LABEL_LEN), generator=g)
  chars[:, LABEL_LEN + 1] = 2
  char_lens = torch.full((B,), LABEL_LEN + 2, dtype=torch.long)
  if device is not None:
    frames, frame_lens, chars, char_lens = (x.to(device) for x in (frames, frame_lens, chars, char_lens))
  return frames, frame_lens, chars, char_lens


def cpu_baseline(model, B, budget_s=20.0):
  """The oracle's encoder+CTC step on this host's cores, bounded to ~budget_s of CPU work."""
  import torch
  from oracle import torch_oracle as O   # checker/baseline only — never on the product path
  rnn_type, H, layers, bi = MODELS[model]
  torch.manual_seed(123456)
  enc = O.OracleVideoEncoder(N_LMK * LMK_DIM, H, rnn_type=rnn_type, num_layers=layers,
                             bidirectional=bi, enable_ctc=True, vocab_size=VOCAB,
                             char2idx=O.default_char2idx()).train()
  opt = torch.optim.Adam(enc.parameters(), lr=1e-4)
  frames, frame_lens, chars, char_lens = synth_batch(B, 123456)
  O.encoder_ctc_step(enc, opt, frames, frame_lens, chars, char_lens, grad_norm=50)  # warm-up
  t0 = time.perf_counter()
  n = 0
  while True:
    O.encoder_ctc_step(enc, opt, frames, frame_lens, chars, char_lens, grad_norm=50)
    n += 1
    el = time.perf_counter() - t0
    if el >= budget_s or n >= 50:
      break
  return {"value": round(n * B * T_FRAMES / el, 1), "unit": "frames/s",
          "cores": torch.get_num_threads(), "kind": "port",
          "sample": "%d steps of the same workload (B=%d,T=%d, %s) through oracle/torch_oracle.py "
                    "(stock torch CPU ops in the reference's order), %d intra-op threads of %d host cores, "
                    "%.1f s" % (n, B, T_FRAMES, model, torch.get_num_threads(), os.cpu_count(), el)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=30)
  ap.add_argument("--warmup", type=int, default=5)
  ap.add_argument("--model", choices=sorted(MODELS), default="gru256")
  ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
  ap.add_argument("--regime", choices=["landmarks", "pixels"], default="landmarks",
                  help="landmarks: reference-faithful (B,75,68,3) input; pixels: build-defined "
                       "(B,75,3,96,96) uint8 clips through the 3-D conv frontend")
  ap.add_argument("--layers", type=int, default=None, help="recurrent layers (default 1; 2 for pixels)")
  ap.add_argument("--no-graph", action="store_true",
                  help="launch every kernel eagerly instead of replaying a captured hipGraph")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--cpu-budget", type=float, default=20.0)
  args = ap.parse_args()

  import torch
  import torch.distributed as dist
  from lipreading_amd import _C
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.optim import FlatParameters, FusedAdam
  from lipreading_amd.train import ctc_step

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if args.gpus > 1 and world != args.gpus:
    sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
             % (args.gpus, args.gpus))
  assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the product path)"
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="nccl", device_id=dev)

  rnn_type, H, layers, bi = MODELS[args.model]
  pixels = args.regime == "pixels"
  if args.layers is not None:
    layers = args.layers
  elif pixels:
    layers = 2      # LipNet-style: STCNN x3 -> 2 x BiGRU-256 -> CTC
  D, G = (2 if bi else 1), (3 if rnn_type == "GRU" else 4)
  B = args.batch
  torch.manual_seed(123456)
  frame_dim = N_LMK * LMK_DIM
  if pixels:
    from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader, feature_dim
    frame_dim = feature_dim(96, 96)
  enc = VideoEncoder(frame_dim, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bi,
                     enable_ctc=True, vocab_size=VOCAB, char2idx=default_char2idx())
  model = PixelLipReader(enc, ConvFrontend3D()) if pixels else enc
  model = model.to(dev).train()
  enc = model.encoder if pixels else model
  flat = FlatParameters(model)
  opt = FusedAdam(flat, lr=1e-4)
  use_graph = not args.no_graph
  sync = None
  if world > 1:
    from lipreading_amd.distributed import GradSync
    # eager: all-reduce each bucket from its autograd hook, overlapped with the rest of backward.
    # graph: forward+backward replay as one hipGraph, the exchange follows it (hooks do not fire
    # on replay, and no collective is ever captured).
    groups = GradSync.groups_for_encoder(enc, flat)
    if pixels:   # conv parameters come first in the flat buffer: one more bucket
      first = min(min(g) for g in groups)
      groups = [list(range(first))] + groups
    sync = GradSync(flat, groups=groups, overlap=not use_graph)
    sync.broadcast_parameters(0)
  # every rank gets its own shard of the global batch (weak scaling: B per GPU)
  frames, frame_lens, chars, char_lens = synth_batch(B, 123456 + rank, dev)
  labels, label_lens = chars[:, 1:], char_lens - 1
  if pixels:   # uint8 clips (B,75,3,96,96), resident in HBM
    gen = torch.Generator().manual_seed(123456 + rank)
    frames = torch.randint(0, 256, (B, T_FRAMES, 3, 96, 96), generator=gen, dtype=torch.uint8).to(dev)

  def fwd_bwd():
    # train_better_model.py:46-48,67,74 — everything up to and including backward
    from lipreading_amd.ctc import ctc_loss_with_status
    opt.zero_grad()
    log_probs, _, _ = model(frames, frame_lens, max_len=T_FRAMES)
    loss, status, _ = ctc_loss_with_status(log_probs, labels, frame_lens, label_lens, 'mean')
    loss.backward()
    return loss.detach(), status

  graph = None
  if use_graph:
    # one hipGraph for the ~400 launches of forward+backward: the T-step recurrent chains are
    # launch-bound from Python (MI355X_MICROARCH.md: eager goes host-bound below ~3 us/kernel)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(2):
        fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      g_loss, g_status = fwd_bwd()

  def step():
    if graph is not None:
      graph.replay()
      loss, status = g_loss, g_status
    else:
      loss, status = fwd_bwd()
    scale = sync(status) if sync is not None else 1.0          # RCCL all-reduce of the flat grads
    opt.step(grad_norm=50, grad_scale=scale, skip=status)      # :78 clip + :80 Adam
    return loss, status

  def fence():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  L = _C.lib()
  for _ in range(args.warmup):
    loss, status = step()
  fence()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    loss, status = step()
  fence()
  elapsed = time.perf_counter() - t0

  # roofline leg (after the timed region, same process, same tensors): the same steps issued
  # eagerly with one step-kernel launch per layer call stamped by a hipEvent pair on its stream
  # (a hipGraph replay does not re-run the host code that records events).
  _C.check(L.lr_profile_enable(1), "lr_profile_enable")
  for _ in range(min(args.steps, 20)):
    fwd_bwd()
  torch.cuda.synchronize()
  L.lr_profile_enable(0)

  el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
  if world > 1:
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
  elapsed = float(el.item())
  loss_v, status_v = float(loss.item()), int(status.item())

  # roofline leg: live hipEvent timing of the recurrent step kernels
  import ctypes
  prof = {}
  for which, name in ((0, "rnn_fwd_step_kernel"), (1, "rnn_bwd_step_kernel")):
    ms, n = ctypes.c_float(0), ctypes.c_int(0)
    L.lr_profile_read(which, ctypes.byref(ms), ctypes.byref(n))
    prof[name] = (ms.value / n.value * 1e3) if n.value else None   # us per launch
  if rank == 0:
    frames_per_step = world * B * T_FRAMES
    ms_per_step = elapsed / args.steps * 1e3
    bytes_per_launch = D * G * H * H * 4          # W_hh streamed once per step, both directions
    flops_per_launch = 2.0 * B * D * G * H * H    # (B x H)·(H x G*H) per direction
    dom = max((k for k in prof if prof[k]), key=lambda k: prof[k], default=None)
    roofline = None
    traffic, traffic_src = None, None
    try:   # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside bench.py)
      with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
        pmc = json.load(f)
      if dom and B == 32:
        traffic = pmc[args.model][dom]["traffic_bytes"]
        traffic_src = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 fetch correction)"
    except Exception:
      pass
    if dom:
      us = prof[dom]
      ach = bytes_per_launch / (us * 1e-6) / 1e9
      roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                  "traffic_source": traffic_src,
                  "avg_launch_us": round(us, 3),
                  "algorithmic_bytes_per_launch": bytes_per_launch,
                  "launches_per_step": 2 * T_FRAMES * layers,
                  "avg_launch_us_by_kernel": {k: (round(v, 3) if v else None) for k, v in prof.items()},
                  "mfma_f32_tflops": round(flops_per_launch / (us * 1e-6) / 1e12, 2),
                  "mfma_f32_frac": round(flops_per_launch / (us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
    out = {
        "metric": "training frames/sec at (B,75,3,96,96) + CTC-loss parity, 1/2/4/8 MI355X",
        "value": round(frames_per_step * args.steps / elapsed, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "regime R (reference-faithful): landmarks (B=%d,T=75,68,3) f32 -> "
                               "%d-layer Bi%s-%d -> Linear(%d,65) -> masked log-softmax -> CTC 'mean' "
                               "(L=30+EOS) -> backward -> clip_grad_norm 50 -> Adam 1e-4; "
                               "pixel regime (B,75,3,96,96)+conv3d has no reference and is not built"
                               % (B, layers, rnn_type, H, D * H),
                   "model": args.model, "per_gpu_batch": B, "global_batch": world * B,
                   "seq_len": T_FRAMES, "parallelism": "dp%d" % world,
                   "launch": "hipGraph replay of forward+backward" if use_graph else "eager"},
        "final_loss": round(loss_v, 6), "skipped_last": status_v,
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
      out["cpu_baseline"] = cpu_baseline(args.model, B, args.cpu_budget)
    print(json.dumps(out))
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
