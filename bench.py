#!/usr/bin/env python
"""Benchmark of the video->characters hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--regime pixels|landmarks|landmarks_attn|all]
                  [--model gru256|lstm768|lstm700|lstm512|gru800] [--batch B]

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
(frontend ->) VideoEncoder forward -> CTC 'mean' loss -> backward -> clip_grad_norm_(50) ->
Adam(1e-4) step — the encoder+CTC part of the reference step
(src/train/train_better_model.py:46-48,74,78,80).  Rank 0 prints ONE JSON line.

Two input regimes (SURVEY.md section 8d says they must be reported separately):
  pixels     BASELINE.json's metric shape: uint8 clips (B,75,3,96,96) -> build-defined 3-D conv
             frontend (bf16 MFMA implicit GEMM; the reference has NO conv frontend, so this stage
             has no reference parity) -> 2-layer BiGRU-256 -> CTC.  This is the headline `value`.
  landmarks  reference-faithful: landmarks (B,75,68,3) f32 -> 1-layer BiGRU-256 (or BiLSTM-768) ->
             CTC; the only regime whose every stage is pinned to the reference.  Reported under
             "regimes" on the same line (and as the headline with --regime landmarks).
  pixels_tfm (opt-in) BASELINE configs[4]: the conv frontend feeding a 4-layer transformer encoder (256, 4
             heads) + CTC; frontend and encoder are both build-defined (no reference symbol).
  landmarks_attn  the reference's WHOLE train step (train_better_model.py:46-80): the landmarks
             regime plus the CharDecodingStep loop (char_dim 300, '1_layer_nn' attention,
             teacher_forcing_ratio 1, L=31 steps), decoder NLL + CTC, per-module clip, Adam.

Extra objects: `roofline` for the regime's dominant kernel (average launch duration measured live
with hipEvent pairs that stamp the dispatch on its own stream, in an eager pass right after the
timed region — hipGraph replays do not re-run the host code that records events) and
`cpu_baseline` (the oracle, i.e. the reference's op sequence on stock torch CPU ops, timed on this
host on a bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
import lipreading_amd  # noqa: E402,F401  (sets GPU_MAX_HW_QUEUES before the HIP runtime initialises: see its __init__)

MODELS = {
    # name: (rnn_type, hidden, bidirectional)
    "gru256": ("GRU", 256, True),     # LipNet-style BiGRU-256 (BASELINE configs[1] encoder)
    "lstm768": ("LSTM", 768, True),   # config/archive/experiments/ecd/* shape (configs[2])
    # the hidden sizes of the reference's live config files (one-launch cluster recurrence, lr_rnn_cluster.hip)
    "lstm700": ("LSTM", 700, True),   # config/defaults.txt:19-21 (hidden_size 700, LSTM)
    "lstm512": ("LSTM", 512, True),   # config/train/attn/attention_type:16-19 (BiLSTM-512 + CTC)
    "gru800": ("GRU", 800, True),     # config/train/micro:6-8 (GRU-800; the archived trainer stacks 5 layers: --layers 5)
    "lstm700uni": ("LSTM", 700, False),   # config/defaults.txt:19-26 as shipped: LSTM-700, bidirectional=False (decoder LSTM-700)
}
T_FRAMES, N_LMK, LMK_DIM, VOCAB, LABEL_LEN, IMG = 75, 68, 3, 64, 30, 96
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3      # fp32-input MFMA
MFMA_BF16_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA
PARITY_TOL_LANDMARKS = 1e-4       # north_star: CTC loss within 1e-4 of the CPU reference (fp32, absolute)
PARITY_TOL_PIXELS = 1e-4          # north_star's bar at (B=32,T=75,96x96); measured 3.1e-5 (bf16 conv stack and recurrent operands
                                  # vs the oracle's bf16-storage conv + fp32 reference tail)
# the arithmetic the path computes in (not a precision claim).  The recurrence and the projections are NOT plain fp32:
# fp32 values are split into bf16 hi + lo planes and multiplied on the bf16 matrix cores with fp32 accumulation
# (lr_rnn_cluster.hip:28-34: all four cross terms, exchanged state rounded to 22 mantissa bits, ~1e-6 of the fp32
# product; LR_RNN_PROJ_BF16X3 / lr_fgemm X3: three cross terms, ~1e-5); gate math, CTC and Adam are fp32.
DTYPE_SPLIT = "fp32-faithful split-bf16 (hi+lo planes on the bf16 matrix cores, fp32 accumulate) recurrence + projections"
DTYPE_NOTE = {
    "pixels": "bf16 (conv frontend, fp32 accumulate) + " + DTYPE_SPLIT + " + f32 (gate math, CTC, Adam)",
    "pixels_tfm": "bf16 (conv frontend, fused attention; fp32 accumulate) + split-bf16 hi+lo (transformer projections, fp32 "
                  "accumulate) + f32 (layer norm, CTC, Adam)",
    "landmarks": DTYPE_SPLIT + " (the first layer's forward projection: exact-fp32 MFMA) + f32 (gate math, head, CTC, Adam)",
    "landmarks_attn": DTYPE_SPLIT + " + f32 (gate math, attention, heads, CTC / NLL, Adam)",
}
# one store -> L2 -> load round trip between two compute units of an XCD, which a step of the one-launch recurrence
# cannot overlap with anything (MI355X_MICROARCH.md price list, handoff-1to1 idle: 0.8-1.0 us)
EXCHANGE_FLOOR_US = 0.75
# lr_profile_read slots (include/lipreading_hip.h)
SLOTS = {0: "rnn_fwd_step_kernel", 1: "rnn_bwd_step_kernel", 2: "conv1_fwd", 3: "conv2_fwd",
         4: "conv3_fwd", 5: "conv2_dgrad", 6: "conv3_dgrad", 7: "conv1_wgrad",
         8: "conv2_wgrad", 9: "conv3_wgrad", 10: "ctc_alpha_beta", 11: "ctc_grad_rows"}


# multi-rank code path on: WORLD_SIZE > 1, or LIPREADING_BENCH_FORCE_DIST=1 to run the very same path
# (process group, broadcast, bucketed all-reduce, barriers) as a 1-rank RCCL group on a single GPU
DIST_ON = int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("LIPREADING_BENCH_FORCE_DIST", "0") == "1"


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


def synth_clips(B, seed, device=None):
  import torch
  g = torch.Generator().manual_seed(seed)
  clips = torch.randint(0, 256, (B, T_FRAMES, 3, IMG, IMG), generator=g, dtype=torch.uint8)
  return clips if device is None else clips.to(device)


def pmc_traffic():
  """(dict, path) of the newest committed profiles/rNN_pmc_traffic.json (HBM bytes per launch from
  separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; rocprofv3 cannot run inside bench.py)."""
  import glob
  cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
  if not cands:
    return None, None
  with open(cands[-1]) as f:
    pmc = json.load(f)
  # the passes are figures of the kernels they ran: quoted only when this build is of the same sources and flags
  # (lipreading_amd/_build.py's fingerprint, written beside the passes on the GPU box); anything else is stale
  from lipreading_amd import _build
  if pmc.get("source_fingerprint") != _build._fingerprint():
    raise LookupError("%s was taken with other kernel sources (fingerprint %s..., this build %s...)"
                      % (os.path.relpath(cands[-1], ROOT), str(pmc.get("source_fingerprint"))[:12], _build._fingerprint()[:12]))
  return pmc, os.path.relpath(cands[-1], ROOT)


def conv_flops(B):
  """Algorithmic flops per launch of each conv kernel at (B,75,3,96,96) (frontend.LAYERS)."""
  from lipreading_amd.frontend import LAYERS
  out, h = {}, IMG
  for i, (cin, cout, (kt, kh, kw), stride, (pt, ph, pw)) in enumerate(LAYERS, 1):
    ho = (h + 2 * ph - kh) // stride + 1
    f = 2.0 * B * T_FRAMES * ho * ho * cout * cin * kt * kh * kw
    out["conv%d_fwd" % i] = f
    out["conv%d_wgrad" % i] = f
    if i > 1:
      out["conv%d_dgrad" % i] = f
    h = ho // 2
  return out


def cpu_baseline(regime, model, layers, B, budget_s=20.0, warmup=3, min_steps=10, char_dim=300, attention='1_layer_nn',
                 use_ctc=True):
  """The oracle's step on this host's cores: `warmup` untimed steps, then >= `min_steps` timed ones
  (BASELINE.md section 3), each timed on its own; median and min reported.  The sample is bounded:
  when a full-batch step would blow the budget (the pixel regime's conv frontend on the CPU), the
  same workload runs at a smaller batch, stated in `sample`."""
  import statistics
  import torch
  from oracle import torch_oracle as O   # checker/baseline only — never on the product path
  rnn_type, H, bi = MODELS[model]
  torch.manual_seed(123456)
  tfm = regime == "pixels_tfm"
  pixels = regime == "pixels" or tfm
  B_cpu = min(B, 4) if pixels else B     # ~1 s per step either way on a 100+-thread host
  full_steps = min_steps                 # timed steps at the GPU line's batch when the sample's batch is smaller (BASELINE.md
                                         # section 3: >= 10 timed steps after the warm-ups, at the stated batch; ~2.2 s each for pixels)
  frame_dim = 96 * (IMG // 16) ** 2 if pixels else N_LMK * LMK_DIM
  if tfm:
    tenc = O.OracleTransformerEncoder(frame_dim, 256, 4, 4, 1024, VOCAB, O.default_char2idx()).train()
    enc = lambda x, lens: tenc(x, lens) + (None,)   # (log_probs, hidden, final_state)
    enc.parameters = tenc.parameters
  else:
    enc = O.OracleVideoEncoder(frame_dim, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bi,
                               enable_ctc=use_ctc, vocab_size=VOCAB, char2idx=O.default_char2idx()).train()
  params = list(enc.parameters())
  convs = []
  if pixels:
    for (cin, cout, k, stride, pad) in O.CONV_LAYERS:
      c = torch.nn.Conv3d(cin, cout, k, stride=(1, stride, stride), padding=pad)
      convs += [c.weight, c.bias]
    params = convs + params
  attn = regime == "landmarks_attn"
  dec = None
  if attn:
    dec = O.OracleCharDecodingStep(H * (2 if bi else 1), rnn_type, 1, char_dim, VOCAB, O.default_char2idx(),
                                   attention_type=attention).train()
  opt = torch.optim.Adam(params + (list(dec.parameters()) if attn else []), lr=1e-4)
  data = {}

  def batch_of(n):
    if n not in data:
      data[n] = synth_batch(n, 123456) + (synth_clips(n, 123456) if pixels else None,)
    return data[n]

  def step(n=B_cpu):
    frames, frame_lens, chars, char_lens, clips = batch_of(n)
    x = frames
    if pixels:
      feats = O.conv_frontend(clips, convs, emulate_bf16=False)
      x = feats.reshape(n, T_FRAMES, -1, 1)
    if use_ctc:
      lp, hid, st = enc(x, frame_lens)
      loss = O.ctc_loss(lp, chars[:, 1:], frame_lens, char_lens - 1, 'mean')
    else:   # (enable_ctc=False, config/defaults.txt:12: the encoder returns (hidden, final_state), train_better_model.py:52)
      hid, st = enc(x, frame_lens)
      loss = None
    opt.zero_grad()
    if attn:   # train_better_model.py:56-74, both backward calls as the reference makes them
      dl, _ = O.decoder_loop(dec, chars, char_lens, hid, frame_lens, st)
      dl.backward(retain_graph=loss is not None)
    if loss is not None:
      loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 50)
    if attn:
      torch.nn.utils.clip_grad_norm_(dec.parameters(), 50)
    opt.step()

  # Intra-op threads: torch's default on this 256-core host is 128, the oracle's WORST count for these shapes (round 4:
  # BiGRU-256 B=32 at 905 ms per step where 8 threads take 170-280) — a thread's share of a (32 x 204) x
  # (204 x 768) product is too small to pay for the fork/join.  Sweep, keep the fastest, time the real sample there.
  host = os.cpu_count() or 1
  default_threads = torch.get_num_threads()
  sweep = {}
  for nt in [n for n in (8, 16, 32, 64, 128) if n <= host] or [default_threads]:
    torch.set_num_threads(nt)
    for _ in range(1 if pixels else 2):
      step()
    ts = []
    for _ in range(2 if pixels else 3):
      t0 = time.perf_counter()
      step()
      ts.append(time.perf_counter() - t0)
    sweep[nt] = statistics.median(ts)
  threads_best = min(sweep, key=sweep.get)
  torch.set_num_threads(threads_best)
  for _ in range(warmup):
    step()
  times = []
  t_all = time.perf_counter()
  while len(times) < min_steps or (time.perf_counter() - t_all < budget_s and len(times) < 50):
    t0 = time.perf_counter()
    step()
    times.append(time.perf_counter() - t0)
  med, best = statistics.median(times), min(times)
  per = B_cpu * T_FRAMES
  full = None
  if B_cpu != B:
    # the thread sweep runs at a smaller batch than the GPU line; `value` is then taken from `full_steps` (>= 10) timed
    # steps at the GPU line's OWN batch, at the sweep's best thread count (two untimed steps first; median)
    for _ in range(min(warmup, 2)):
      step(B)
    dts = []
    for _ in range(full_steps):
      t0 = time.perf_counter()
      step(B)
      dts.append(time.perf_counter() - t0)
    dt = statistics.median(dts)
    full = {"batch": B, "steps": full_steps, "ms_per_step": round(dt * 1e3, 1), "ms_per_step_min": round(min(dts) * 1e3, 1),
            "value": round(B * T_FRAMES / dt, 1), "unit": "frames/s"}
  used = torch.get_num_threads()
  torch.set_num_threads(default_threads)
  # `value` is quoted at the GPU line's own batch whenever a step at that batch was timed
  return {"value": full["value"] if full else round(per / med, 1), "unit": "frames/s",
          "value_sample": round(per / med, 1), "value_best": round(per / best, 1), "full_batch_step": full,
          "ms_per_step_median": round(med * 1e3, 2), "ms_per_step_min": round(best * 1e3, 2),
          "steps": len(times), "warmup": warmup,
          "cores": used, "threads_best": threads_best, "host_cores": os.cpu_count(), "kind": "port",
          "thread_sweep_ms_per_step": {str(k): round(v * 1e3, 1) for k, v in sweep.items()},
          "sample": "%d timed steps after %d warm-ups of the same workload (%s regime, B=%d%s, T=%d, %s x%d) through "
                    "oracle/torch_oracle.py (stock torch CPU ops in the reference's order%s), %d intra-op threads "
                    "of %d host cores (the fastest of the sweep beside it); value = frames / median step, at the GPU "
                    "line's batch where full_batch_step exists"
                    % (len(times), warmup, regime, B_cpu,
                       " — bounded sample, the GPU line is B=%d" % B if B_cpu != B else "", T_FRAMES, model, layers,
                       "; conv frontend = F.conv3d/max_pool3d fp32" if pixels else "",
                       used, os.cpu_count())}


def _frame_flips(lp_hip, lp_ref, lens):
  """Per-frame comparison of two (B,T,C) log-prob lattices over the valid frames: how many argmaxes differ, and how
  close to a tie the ORACLE is where they do (its top-1 minus top-2 log-prob at those frames).  An argmax can only
  flip where that margin is below twice the largest log-prob difference."""
  import torch
  lp_hip, lp_ref = lp_hip.detach().float().cpu(), lp_ref.detach().float().cpu()
  T = lp_ref.shape[1]
  valid = torch.arange(T).unsqueeze(0) < lens.cpu().unsqueeze(1)
  top2 = lp_ref.topk(2, dim=-1).values
  margin = (top2[..., 0] - top2[..., 1])
  flip = (lp_hip.argmax(-1) != lp_ref.argmax(-1)) & valid
  n, nf = int(valid.sum()), int(flip.sum())
  d = float(((lp_hip - lp_ref).abs() * valid.unsqueeze(-1)).max())
  return {"frames": n, "argmax_flips": nf, "flip_fraction": float("%.3g" % (nf / max(n, 1))),
          "max_oracle_margin_at_flips": float("%.3g" % float(margin[flip].max())) if nf else 0.0,
          "median_oracle_margin": float("%.3g" % float(margin[valid].median())),
          "max_abs_log_prob_diff": float("%.3g" % d),
          "flips_outside_2x_diff": int((flip & (margin > 2 * d)).sum())}


def parity_block(regime, model_name, layers, B, dev, train_steps=50):
  """The metric's "+ CTC-loss parity": the HIP path and the oracle on IDENTICAL inputs and weights, one
  forward + CTC 'mean' loss each (the caller contract of train_better_model.py:46-48), at the bench
  shape.  Landmarks regime: every stage is pinned to the reference, tolerance 1e-4 absolute (fp32).
  Pixels regime: the conv stage is build-defined; the oracle is F.conv3d with this repo's bf16 storage
  points emulated, then the reference's encoder and CTC — tolerance stated in DESIGN.md section 7; the loss of
  the plain fp32 F.conv3d oracle (no bf16 anywhere) is reported beside it, the one-product input projection
  ('bf16x1') is compared with the default, and the comparison is
  repeated on the weights after `train_steps` optimisation steps of the HIP path (peaked log-probs: a random
  initialisation's lattice is nearly flat, so its argmaxes sit within a rounding of a tie)."""
  import torch
  from oracle import torch_oracle as O   # the checker
  from lipreading_amd.ctc import ctc_loss_with_status
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  rnn_type, H, bi = MODELS[model_name]
  tfm = regime == "pixels_tfm"
  pixels = regime == "pixels" or tfm
  torch.manual_seed(123456)
  frame_dim = 96 * (IMG // 16) ** 2 if pixels else N_LMK * LMK_DIM
  if tfm:   # configs[4]: both stages build-defined; the oracle is torch's own nn.TransformerEncoder behind the same conv oracle
    from lipreading_amd.transformer import TransformerVideoEncoder
    ref = O.OracleTransformerEncoder(frame_dim, 256, 4, 4, 1024, VOCAB, O.default_char2idx()).eval()
    enc = TransformerVideoEncoder(frame_dim, d_model=256, nhead=4, num_layers=4, dim_feedforward=1024,
                                  enable_ctc=True, vocab_size=VOCAB, char2idx=default_char2idx())
    enc.load_state_dict({k: v for k, v in ref.state_dict().items() if not k.startswith("encoder.")})
  else:
    ref = O.OracleVideoEncoder(frame_dim, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bi,
                               enable_ctc=True, vocab_size=VOCAB, char2idx=O.default_char2idx()).eval()
    enc = VideoEncoder(frame_dim, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bi,
                       enable_ctc=True, vocab_size=VOCAB, char2idx=default_char2idx())
    enc.load_state_dict(ref.state_dict())
  frames, frame_lens, chars, char_lens = synth_batch(B, 123456)
  labels, label_lens = chars[:, 1:], char_lens - 1
  lens_d = frame_lens.to(dev)

  def hip_loss(lp):
    loss, status, _ = ctc_loss_with_status(lp, labels.to(dev), lens_d, label_lens.to(dev), 'mean')
    return float(loss.item()), int(status.item())

  extra = {}
  with torch.no_grad():
    if pixels:
      from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader
      fe = ConvFrontend3D()
      model = PixelLipReader(enc, fe).to(dev).eval()
      clips = synth_clips(B, 123456)
      clips_d = clips.to(dev)

      def oracle_lp(emulate):
        convs = [p.detach().cpu().clone() for p in fe.parameters_in_order()]
        feats = O.conv_frontend(clips, convs, emulate_bf16=emulate)
        return ref(feats.reshape(B, T_FRAMES, -1, 1), frame_lens)[0]

      lp_hip, _, _ = model(clips_d, lens_d, max_len=T_FRAMES)
      lp_ref = oracle_lp(True)
      tol = PARITY_TOL_PIXELS
      if tfm:
        note = ("same uint8 clips and weights; oracle = F.conv3d/max_pool3d with bf16 rounding at this repo's storage "
                "points -> torch.nn.TransformerEncoder (fp32, CPU) -> reference ctc_loss; HIP = PixelLipReader over "
                "TransformerVideoEncoder defaults (split-bf16 projections, fused bf16 attention).  Both stages are "
                "build-defined: no reference parity, the tolerance is north_star's 1e-4 on the loss")
      else:
        note = ("same uint8 clips and weights; oracle = F.conv3d/max_pool3d with bf16 rounding at this repo's "
                "storage points -> reference VideoEncoder (fp32) -> reference ctc_loss; HIP = PixelLipReader "
                "defaults (%s recurrence, %s input projection)" % (enc.recurrence, enc.input_projection))
      # the same lattice from the other recurrences / projections of the HIP path, and the un-shaped oracle (fp32
      # conv, no bf16 anywhere)
      default_rec, default_proj = getattr(enc, "recurrence", None), getattr(enc, "input_projection", None)
      l_ref = float(O.ctc_loss(lp_ref, labels, frame_lens, label_lens, 'mean'))
      s_ref0 = GreedyStrings.oracle(lp_ref, frame_lens)

      def variant(rec, proj):
        enc.recurrence, enc.input_projection = rec, proj
        lp_v, _, _ = model(clips_d, lens_d, max_len=T_FRAMES)
        enc.recurrence, enc.input_projection = default_rec, default_proj
        l_v, _ = hip_loss(lp_v)
        return dict(_frame_flips(lp_v, lp_ref, frame_lens), loss_hip=round(l_v, 7), abs_diff=float("%.3g" % abs(l_v - l_ref)),
                    greedy_strings_equal=GreedyStrings.hip(lp_v, lens_d) == s_ref0)

      if not tfm:
        extra["other_paths_vs_the_same_oracle"] = {
            "recurrence 'split' (fp32-faithful)": variant("split", default_proj),
            "recurrence 'split' + input projection 'bf16x1' (ONE bf16 product: W_ih rounded to bf16)": variant("split", "bf16x1")}
      lp_ref32 = oracle_lp(False)
      l_ref32 = float(O.ctc_loss(lp_ref32, labels, frame_lens, label_lens, 'mean'))
      extra["loss_oracle_fp32conv"] = round(l_ref32, 7)
      extra["abs_diff_vs_fp32conv_oracle"] = float("%.3g" % abs(hip_loss(lp_hip)[0] - l_ref32))
      extra["vs_fp32conv_oracle"] = _frame_flips(lp_hip, lp_ref32, frame_lens)
    else:
      model = enc.to(dev).eval()
      lp_hip, _, _ = model(frames.to(dev), lens_d, max_len=T_FRAMES)
      lp_ref, _, _ = ref(frames, frame_lens)
      tol = PARITY_TOL_LANDMARKS
      note = "same landmarks and weights; oracle = reference VideoEncoder + ctc_loss on stock torch CPU ops (fp32)"
    lh, st = hip_loss(lp_hip)
    loss_ref = O.ctc_loss(lp_ref, labels, frame_lens, label_lens, 'mean')
    s_hip = GreedyStrings.hip(lp_hip, lens_d)
    s_ref = GreedyStrings.oracle(lp_ref, frame_lens)
  lr = float(loss_ref.item())
  out = {"regime": regime, "loss_hip": round(lh, 7), "loss_oracle": round(lr, 7), "abs_diff": float("%.3g" % abs(lh - lr)),
         "tol": tol, "ok": bool(abs(lh - lr) <= tol and st == 0),
         "greedy_strings_equal": s_hip == s_ref, "batch": B, "what": note}
  out.update(_frame_flips(lp_hip, lp_ref, frame_lens))
  out["string_tolerance"] = ("greedy strings may differ from the oracle's only at frames whose ORACLE top-1/top-2 log-prob "
                             "margin is below 2 x max_abs_log_prob_diff (flips_outside_2x_diff must be 0); at a random "
                             "initialisation the lattice is nearly flat (median margin above), after training it is not")
  out.update(extra)
  if train_steps and train_steps > 0:
    # ... and once more on TRAINED weights: `train_steps` optimisation steps of the HIP path (the product's ctc_step),
    # the weights copied into the oracle, one forward each
    from lipreading_amd import train as T
    from lipreading_amd.optim import FlatParameters, FusedAdam
    model.train()
    opt = FusedAdam(FlatParameters(model), lr=1e-3)
    x_d = clips_d if pixels else frames.to(dev)
    for _ in range(train_steps):
      T.ctc_step(model, opt, x_d, lens_d, chars.to(dev), char_lens.to(dev), grad_norm=50, max_len=T_FRAMES)
    model.eval()
    with torch.no_grad():
      ref.load_state_dict({k: v.detach().cpu() for k, v in enc.state_dict().items()})
      lp_hip2, _, _ = model(x_d, lens_d, max_len=T_FRAMES)
      lp_ref2 = oracle_lp(True) if pixels else ref(frames, frame_lens)[0]
      lh2, _ = hip_loss(lp_hip2)
      lr2 = float(O.ctc_loss(lp_ref2, labels, frame_lens, label_lens, 'mean'))
      out["after_training"] = dict(_frame_flips(lp_hip2, lp_ref2, frame_lens), steps=train_steps, lr=1e-3,
                                   loss_hip=round(lh2, 7), loss_oracle=round(lr2, 7),
                                   abs_diff=float("%.3g" % abs(lh2 - lr2)),
                                   greedy_strings_equal=GreedyStrings.hip(lp_hip2, lens_d) == GreedyStrings.oracle(lp_ref2, frame_lens))
  return out


def parity_block_attn(model_name, B, dev, char_dim, attention, use_ctc):
  """The reference's WHOLE forward (train_better_model.py:46-65) on identical inputs and weights: encoder (+ CTC 'mean'
  when the encoder has it) and the teacher-forced CharDecodingStep loop with its NLL / non-PAD count — HIP path against
  the oracle (stock torch CPU ops in the reference's order).  Tolerance 1e-4 absolute on both losses (fp32)."""
  import torch
  from oracle import torch_oracle as O   # the checker
  from lipreading_amd import train as T
  from lipreading_amd.attention_decoder import CharDecodingStep
  from lipreading_amd.ctc import ctc_loss_with_status
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  rnn_type, H, bi = MODELS[model_name]
  D = 2 if bi else 1
  torch.manual_seed(123456)
  ref = O.OracleVideoEncoder(N_LMK * LMK_DIM, H, rnn_type=rnn_type, num_layers=1, bidirectional=bi, enable_ctc=use_ctc,
                             vocab_size=VOCAB, char2idx=O.default_char2idx()).eval()
  rdec = O.OracleCharDecodingStep(D * H, rnn_type, 1, char_dim, VOCAB, O.default_char2idx(), attention_type=attention).eval()
  enc = VideoEncoder(N_LMK * LMK_DIM, H, rnn_type=rnn_type, num_layers=1, bidirectional=bi, enable_ctc=use_ctc,
                     vocab_size=VOCAB, char2idx=default_char2idx())
  dec = CharDecodingStep(enc, char_dim=char_dim, vocab_size=VOCAB, char2idx=default_char2idx(), attention_type=attention)
  enc.load_state_dict(ref.state_dict())
  dec.load_state_dict(rdec.state_dict())
  enc, dec = enc.to(dev).eval(), dec.to(dev).eval()
  frames, frame_lens, chars, char_lens = synth_batch(B, 123456)
  labels, label_lens = chars[:, 1:], char_lens - 1
  L = int(label_lens.max())
  out = {"regime": "landmarks_attn", "batch": B, "tol": PARITY_TOL_LANDMARKS,
         "what": "same landmarks, captions and weights; oracle = reference VideoEncoder%s + CharDecodingStep loop (teacher "
                 "forced) + NLL / non-PAD count on stock torch CPU ops (fp32)" % (" + ctc_loss" if use_ctc else "")}
  with torch.no_grad():
    if use_ctc:
      lp_r, hid_r, st_r = ref(frames, frame_lens)
      lp_h, hid_h, st_h = enc(frames.to(dev), frame_lens.to(dev), max_len=T_FRAMES)
      l_r = float(O.ctc_loss(lp_r, labels, frame_lens, label_lens, 'mean'))
      l_h, status, _ = ctc_loss_with_status(lp_h, labels.to(dev), frame_lens.to(dev), label_lens.to(dev), 'mean')
      out.update(ctc_loss_hip=round(float(l_h), 7), ctc_loss_oracle=round(l_r, 7), ctc_abs_diff=float("%.3g" % abs(float(l_h) - l_r)),
                 greedy_strings_equal=GreedyStrings.hip(lp_h, frame_lens.to(dev)) == GreedyStrings.oracle(lp_r, frame_lens))
    else:
      hid_r, st_r = ref(frames, frame_lens)
      hid_h, st_h = enc(frames.to(dev), frame_lens.to(dev), max_len=T_FRAMES)
    dl_r, dlp_r = O.decoder_loop(rdec, chars, char_lens, hid_r, frame_lens, st_r)
    dlp_h, _, _ = dec.decode_sequence(chars[:, :L].to(dev), st_h, frame_lens.to(dev), hid_h, teacher_forced=(True,) * L)
    dl_h = float(T.decoder_nll(dlp_h, labels.to(dev), 0))
  dd = abs(dl_h - float(dl_r))
  out.update(decoder_loss_hip=round(dl_h, 7), decoder_loss_oracle=round(float(dl_r), 7), decoder_abs_diff=float("%.3g" % dd),
             max_abs_decoder_log_prob_diff=float("%.3g" % float((dlp_h.cpu() - dlp_r).abs().max())),
             abs_diff=float("%.3g" % max(dd, out.get("ctc_abs_diff", 0.0))))
  out["ok"] = bool(out["abs_diff"] <= PARITY_TOL_LANDMARKS)
  return out


class GreedyStrings(object):
  @staticmethod
  def hip(lp, lens):
    from lipreading_amd.data import default_char2idx
    from lipreading_amd.decoder import GreedyDecoder, ctc_labels
    return GreedyDecoder(ctc_labels(default_char2idx())).decode(lp, lens)[0]

  @staticmethod
  def oracle(lp, lens):
    from oracle import torch_oracle as O
    return O.greedy_decode(lp, lens, O.ctc_labels())[0]


def launch_ranks(n, argv=None, timeout=None):
  """`python bench.py --gpus N` without a launcher around it: re-execute this script as N ranks, one
  process per GPU, with the environment torch.distributed.run would set (RANK, LOCAL_RANK,
  WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT = a free port).  Rank 0's stdout is this process's
  stdout (the ONE JSON line); the other ranks' stdout goes to stderr.  Returns the worst exit code;
  if a rank dies the others are terminated (by PID) instead of hanging in a collective."""
  import socket
  import subprocess
  argv = list(sys.argv[1:] if argv is None else argv)
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  procs = []
  for r in range(n):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LIPREADING_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver (RCCL needs it)
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                  stdout=None if r == 0 else sys.stderr))
  t0 = time.time()
  rc = 0
  live = list(procs)
  while live:
    for p in list(live):
      code = p.poll()
      if code is None:
        continue
      live.remove(p)
      if code != 0:
        rc = rc or code
        for q in live:        # a dead rank would leave the others waiting in a collective forever
          q.terminate()
    if timeout is not None and time.time() - t0 > timeout:
      for q in live:
        q.terminate()
      rc = rc or 124
      timeout = None
    time.sleep(0.05)
  return rc


def probe_main(args):
  """LIPREADING_BENCH_PROBE=1: the rank plumbing of launch_ranks without a GPU — every rank joins a
  gloo group from the environment it was given and rank 0 prints what the group saw
  (tests/test_distributed_cpu.py)."""
  import torch
  import torch.distributed as dist
  world, rank, local = (int(os.environ[k]) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"))
  dist.init_process_group(backend="gloo", rank=rank, world_size=world)
  seen = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
  dist.all_gather(seen, torch.tensor([rank, local]))
  total = torch.tensor([rank + 1.0])
  dist.all_reduce(total)
  dist.barrier()
  dist.destroy_process_group()
  if rank == 0:
    print(json.dumps({"probe": True, "n_gpus": args.gpus, "world": world, "ranks": [int(t[0]) for t in seen],
                      "local_ranks": [int(t[1]) for t in seen], "sum_rank_plus_1": float(total),
                      "master": "%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"])}), flush=True)


def run_regime(args, regime, world, rank, dev, recurrence=None):
  """Builds the model, times `args.steps` steps, measures the dominant kernels.  Returns a dict."""
  import torch
  import torch.distributed as dist
  from lipreading_amd import _C
  from lipreading_amd.ctc import ctc_loss_with_status
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.optim import FlatParameters, FusedAdam

  rnn_type, H, bi = MODELS[args.model]
  if os.environ.get("LIPREADING_RNN_DEBUG"):
    # experiment switch (include/lipreading_hip.h lr_rnn_debug_disable_cluster): 1 = no cluster recurrence,
    # 2 = no pair recurrence (GRU-256 then takes the 8-member cluster kernels)
    _C.lib().lr_rnn_debug_disable_cluster(int(os.environ["LIPREADING_RNN_DEBUG"]))
  if os.environ.get("LIPREADING_RNN_TUNE"):
    # experiment: "fwd_delay,fwd_sleep,bwd_delay,bwd_sleep" (lr_rnn_debug_tune)
    t_ = [int(v) for v in os.environ["LIPREADING_RNN_TUNE"].split(",")]
    _C.lib().lr_rnn_debug_tune(0, t_[0], t_[1])
    _C.lib().lr_rnn_debug_tune(1, t_[2], t_[3])
  tfm = regime == "pixels_tfm"     # BASELINE configs[4]: conv features -> transformer encoder -> CTC (build-defined)
  pixels = regime == "pixels" or tfm
  attn = regime == "landmarks_attn"
  layers = args.layers if args.layers is not None else (2 if pixels else 1)
  if attn:
    layers = 1   # better_model.py:134-136: the decoder takes the encoder's layer count; shipped configs use 1
  D, G = (2 if bi else 1), (3 if rnn_type == "GRU" else 4)
  B = args.batch
  torch.manual_seed(123456)
  frame_dim = N_LMK * LMK_DIM
  if pixels:
    from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader, feature_dim
    frame_dim = feature_dim(IMG, IMG)
  if tfm:
    from lipreading_amd.transformer import TransformerVideoEncoder
    enc = TransformerVideoEncoder(frame_dim, d_model=256, nhead=4, num_layers=4, dim_feedforward=1024,
                                  enable_ctc=True, vocab_size=VOCAB, char2idx=default_char2idx())
  else:
    # (the attention regime follows the reference's config file: --no-ctc = enable_ctc False, config/defaults.txt:12)
    enc = VideoEncoder(frame_dim, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bi,
                       rnn_dropout=(args.rnn_dropout if attn else 0),
                       enable_ctc=not (attn and args.no_ctc), vocab_size=VOCAB, char2idx=default_char2idx())
  model = PixelLipReader(enc, ConvFrontend3D()) if pixels else enc
  # experiment switch: 'f32' = step kernels instead of the one-launch cluster recurrence
  if os.environ.get("LIPREADING_RECURRENCE") and hasattr(enc, "recurrence"):
    enc.recurrence = os.environ["LIPREADING_RECURRENCE"]
  if os.environ.get("LIPREADING_OVERLAP_WGRAD") and pixels:
    # experiment: 0 = the recurrent layers' weight-gradient GEMMs stay on the main stream
    from lipreading_amd import encoder as _enc_mod
    _enc_mod.overlap_weight_grads = os.environ["LIPREADING_OVERLAP_WGRAD"] != "0"
  if os.environ.get("LIPREADING_INPUT_PROJECTION") and pixels:
    # experiment: 'bf16x1' = one bf16 product per GEMM of the recurrent layers' projections (LR_RNN_PROJ_BF16X1)
    enc.input_projection = os.environ["LIPREADING_INPUT_PROJECTION"]
  if recurrence is not None and hasattr(enc, "recurrence"):
    enc.recurrence = recurrence
  model = model.to(dev).train()
  enc = model.encoder if pixels else model
  flat = FlatParameters(model)
  opt = FusedAdam(flat, lr=1e-4)
  if pixels and not DIST_ON and os.environ.get("LIPREADING_SUMSQ_EARLY", "1") != "0":
    opt.sum_squares_early(enc)     # the encoder's share of the clip's sum of squares beside the conv backward
  dec = dec_opt = dec_sync = None
  if attn:
    import torch.nn.functional as F
    from lipreading_amd.attention_decoder import CharDecodingStep
    dec = CharDecodingStep(enc, char_dim=args.char_dim, vocab_size=VOCAB, char2idx=default_char2idx(),
                           rnn_dropout=args.rnn_dropout, attention_type=args.attention).to(dev).train()
    dec_flat = FlatParameters(dec)
    dec_opt = FusedAdam(dec_flat, lr=1e-4)
  use_graph = not args.no_graph
  if DIST_ON and pixels:
    # the pixel step is GPU-bound either way (eager and hipGraph replay agree within 0.2 %), and eager
    # launches let the gradient buckets go out from the gradient-ready hooks: the big bucket (first
    # recurrent layer, 21 MB) is final before the conv backward starts and rides under it on the side
    # stream.  A hipGraph replay would put the whole exchange after backward.
    use_graph = False
  sync = None
  if DIST_ON:
    from lipreading_amd.distributed import GradSync
    # eager: all-reduce each bucket the moment its gradients are final, overlapped with the rest
    # of backward.  graph: forward+backward replay as one hipGraph and the exchange follows it
    # (hooks do not fire on replay, and no collective is ever captured).
    groups = None   # one bucket
    if pixels and not tfm:
      # [conv], [encoder]: the encoder's bucket goes out behind the last recurrence of the step and rides under the
      # conv backward (distributed.GradSync.groups_for_pixel_model)
      groups = GradSync.groups_for_pixel_model(model, flat)
    elif not tfm:
      groups = GradSync.groups_for_encoder(enc, flat)
    sync = GradSync(flat, groups=groups, overlap=not use_graph)
    sync.broadcast_parameters(0)
    if attn:
      dec_sync = GradSync(dec_flat, overlap=False)   # one bucket, exchanged after backward
      dec_sync.broadcast_parameters(0)
  # every rank gets its own shard of the global batch (weak scaling: B per GPU)
  frames, frame_lens, chars, char_lens = synth_batch(B, 123456 + rank, dev)
  labels, label_lens = chars[:, 1:], char_lens - 1
  if pixels:
    frames = synth_clips(B, 123456 + rank, dev)
  # The step is the PRODUCT's: lipreading_amd.train.ctc_step (encoder+CTC; train_better_model.py:46-48,74,78,80)
  # or train.decoder_step (the whole reference step with the attention decoder loop), and the hipGraph
  # capture is the product's too (train.StepGraphs: one graph per batch shape, replayed).
  from lipreading_amd import train as T
  graphs = T.StepGraphs(enabled=use_graph)
  flags = (True,) * (LABEL_LEN + 1)     # teacher_forcing_ratio 1: every step of the decoder loop is teacher forced

  def step(graphs=graphs, sync=None, dec_sync=None):
    if attn:
      dloss, loss, status = T.decoder_step(enc, dec, (opt, dec_opt), frames, frame_lens, chars, char_lens, flags, 1, 0,
                                           grad_norm=50, max_len=T_FRAMES, grad_sync=(sync, dec_sync), graphs=graphs)
      if loss is None:   # enable_ctc False: the decoder loss is the step's loss, nothing can be skipped
        return dloss, _zero_status
      return loss, status
    return T.ctc_step(model, opt, frames, frame_lens, chars, char_lens, grad_norm=50, max_len=T_FRAMES,
                      grad_sync=sync, graphs=graphs)

  def fence():
    if DIST_ON:
      dist.barrier()
    torch.cuda.synchronize()

  _zero_status = torch.zeros(1, dtype=torch.int32, device=dev)
  graph_note = None
  step_sync = dict(sync=sync, dec_sync=dec_sync)
  # untimed priming, before the W warm-up steps: a shape's first steps run eagerly and the next one is
  # captured (StepGraphs), so that warm-up and the timed region see nothing but replays
  for _ in range(graphs.warmup + 1 if use_graph else 0):
    loss, status = step(**step_sync)
  if use_graph:
    ok = torch.tensor([1 if graphs.captures > 0 else 0], device=dev)
    if DIST_ON:   # every rank must be on the same path
      dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
      graphs.enabled, use_graph = False, False
      graph_note = "hipGraph capture failed; eager launches"

  # GPU-bound steps (the pixel regimes: ~100 long kernels, two streams) gain nothing from a graph and may lose
  # a little (the captured side-stream branches overlap less well than eager launches do): time both forms on
  # a few untimed steps and keep the faster one.  So do the decoder regimes since round 6: an EAGER step runs the
  # loop's weight half and the decoder's optimiser on a side stream (attention_decoder.overlap_weight_half), a
  # captured one keeps them on one queue.  Launch-bound steps (landmarks, encoder + CTC only) always replay.
  launch_choice = None
  if use_graph and (pixels or attn) and not DIST_ON:
    def probe(g):
      fence()
      t0 = time.perf_counter()
      for _ in range(6):
        step(graphs=g, **{k: v for k, v in step_sync.items()})
      fence()
      return (time.perf_counter() - t0) / 6
    probe(None)
    t_replay, t_eager = probe(graphs), probe(None)
    launch_choice = {"replay_ms": round(t_replay * 1e3, 4), "eager_ms": round(t_eager * 1e3, 4)}
    if t_eager < 0.995 * t_replay:
      step_sync["graphs"] = None
      use_graph = False
      graph_note = ("eager launches (auto: %.3f ms/step against %.3f for the hipGraph replay of the same step)"
                    % (t_eager * 1e3, t_replay * 1e3))
  # The synthetic batch is resident in HBM: like a loader that stages its batches straight into the graph's input
  # buffers (StepGraphs.staging_buffers), hand the step those buffers — a replay then has no per-step staging copy
  # of the (unchanged) inputs in front of it.
  if use_graph and step_sync.get("graphs", graphs) is not None and graphs.staging_buffers() is not None:
    st_frames, st_lens, st_chars, st_clens = graphs.staging_buffers()
    st_frames.copy_(frames); st_lens.copy_(frame_lens); st_chars.copy_(chars); st_clens.copy_(char_lens)
    frames, frame_lens, chars, char_lens = st_frames, st_lens, st_chars, st_clens
  L = _C.lib()
  for _ in range(args.warmup):
    loss, status = step(**step_sync)
  # the timed region — EXACTLY args.steps steps between two fences — is repeated args.repeats times
  # back to back; `value` comes from the median repeat, the fastest is reported beside it
  elapsed_all = []
  for _ in range(max(1, args.repeats)):
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      loss, status = step(**step_sync)
    fence()
    elapsed_all.append(time.perf_counter() - t0)

  # roofline leg (after the timed region, same process, same tensors).  Per-kernel durations are taken with the conv
  # backward on ONE stream: in the timed step a layer's weight gradient runs on a side stream beside its data
  # gradient (frontend._WGRAD_SIDE_STREAM) — the two then share the chip and each one's duration says nothing about
  # the kernel itself.
  from lipreading_amd import frontend as _FE
  from lipreading_amd import encoder as _ENC
  _side_saved, _FE._WGRAD_SIDE_STREAM = _FE._WGRAD_SIDE_STREAM, False
  # ... and the recurrent layers' weight-gradient GEMMs too (encoder.overlap_weight_grads puts them on a side stream,
  # where they ran beside layer 3's backward: round 3's conv3_wgrad read 84 us on one visit and 199 on the next)
  _ovl_saved, _ENC.overlap_weight_grads = _ENC.overlap_weight_grads, False
  _C.check(L.lr_profile_enable(1), "lr_profile_enable")
  n_prof = min(args.steps, 20 if not pixels else 5)
  for _ in range(n_prof):    # eager launches of the same step (graph replays do not re-run the host code that records events)
    if sync is not None:
      with sync.hold():       # no bucket goes out from the hooks: this leg only samples kernel durations
        step(graphs=None)
    else:
      step(graphs=None)
  torch.cuda.synchronize()
  L.lr_profile_enable(0)
  _FE._WGRAD_SIDE_STREAM = _side_saved
  _ENC.overlap_weight_grads = _ovl_saved
  prof = {}
  for which, name in SLOTS.items():
    ms, n = ctypes.c_float(0), ctypes.c_int(0)
    L.lr_profile_read(which, ctypes.byref(ms), ctypes.byref(n))
    if n.value:
      prof[name] = (ms.value / n.value * 1e3, n.value / n_prof)   # us per launch, samples per step

  el = torch.tensor(elapsed_all, dtype=torch.float64, device=dev)
  if DIST_ON:
    dist.all_reduce(el, op=dist.ReduceOp.MAX)     # every repeat: the slowest rank's time
  elapsed_all = sorted(float(v) for v in el.tolist())
  elapsed = elapsed_all[(len(elapsed_all) - 1) // 2]   # median (lower middle for an even count)
  elapsed_min = elapsed_all[0]

  # the data-parallel exchange on its own: each bucket's all-reduce timed on the side stream it runs on
  bucket_us = None
  if sync is not None:
    bucket_us = []
    for lo, hi in sync.bounds:
      buf = flat.grad[lo:hi]
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      for _ in range(3):
        dist.all_reduce(buf)
      fence()
      e0.record()
      for _ in range(10):
        dist.all_reduce(buf)
      e1.record()
      torch.cuda.synchronize()
      bucket_us.append({"bytes": int((hi - lo) * 4), "all_reduce_us": round(e0.elapsed_time(e1) * 100.0, 1)})
    flat.grad.zero_()
  res = {"regime": regime, "elapsed": elapsed, "loss": float(loss.item()), "skipped": int(status.item()),
         "layers": layers, "use_graph": use_graph, "graph_note": graph_note,
         "rccl_ranks": dist.get_world_size() if DIST_ON else 1, "all_reduce_buckets": bucket_us,
         "launch_choice": launch_choice}
  if sync is not None:
    sync.close()
  if dec_sync is not None:
    dec_sync.close()
  if rank != 0:
    return res

  frames_per_step = world * B * T_FRAMES
  res["value"] = round(frames_per_step * args.steps / elapsed, 1)
  res["ms_per_step"] = round(elapsed / args.steps * 1e3, 4)
  res["ms_per_step_min"] = round(elapsed_min / args.steps * 1e3, 4)
  res["repeats"] = len(elapsed_all)
  # CTC kernels (HBM-bound by the survey's accounting, latency-bound in fact): 3*T*V'*4 B per sample
  ctc_us = sum(prof[k][0] for k in ("ctc_alpha_beta", "ctc_grad_rows") if k in prof)
  if "ctc_alpha_beta" in prof and "ctc_grad_rows" in prof:
    ctc_bytes = 3 * T_FRAMES * (VOCAB + 1) * 4 * B
    res["ctc"] = {"kernels": {"ctc_alpha_beta": round(prof["ctc_alpha_beta"][0], 2),
                              "ctc_grad_rows": round(prof["ctc_grad_rows"][0], 2)},
                  "unit": "us per launch", "algorithmic_bytes": ctc_bytes,
                  "achieved_GBps": round(ctc_bytes / (ctc_us * 1e-6) / 1e9, 1),
                  "frac_of_hbm_peak": round(ctc_bytes / (ctc_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                  "note": "one workgroup per sample: %d workgroups on 256 CUs — latency-bound, ~1%% of the step" % B}
  by_kernel = {k: round(v[0], 3) for k, v in prof.items()}
  # which recurrence ran: 'f32' one launch per step | 'split' one launch per layer pass on clusters of ceil(H / 32)
  # (past 864 / 768 units: ceil(H / 16)) CUs, bf16 hi + lo planes (lr_rnn_cluster.hip)
  rec, rec_bwd = "f32", "f32"
  kind = 0
  members = (H + 31) // 32
  if not tfm:
    mode_id = {"GRU": 0, "LSTM": 1, "RNN": 2}[rnn_type]
    want = getattr(enc, "recurrence", "f32")
    if want in ("auto", "split"):
      kind = L.lr_rnn_pair_supported(mode_id, B, T_FRAMES, frame_dim, H, D)
      rec = rec_bwd = "split" if kind else "f32"
  res["recurrence"] = rec if rec == rec_bwd else "%s forward / %s backward" % (rec, rec_bwd)
  pass_kernel = None
  if rec == "split":
    pass_kernel = "rnnc_%%s_kernel<%d,%d>" % (G, members)
  pass_names = {}
  for slot, which in (("rnn_fwd_step_kernel", "fwd"), ("rnn_bwd_step_kernel", "bwd")):
    if pass_kernel and slot in by_kernel:
      # this slot carries ONE launch per layer pass (all 75 steps), not a step
      pass_names[slot] = pass_kernel % which + " (layer pass)"
      by_kernel[pass_names[slot]] = by_kernel.pop(slot)
  res["pair_errors"] = int(L.lr_rnn_pair_errors()) if "split" in (rec, rec_bwd) else 0
  roofline = None
  if pixels:
    flops = conv_flops(B)
    # dominant = largest time per step among the conv kernels (one launch of each per step)
    cand = {k: v for k, v in prof.items() if k in flops}
    if cand:
      dom = max(cand, key=lambda k: cand[k][0])
      us = cand[dom][0]
      ach = flops[dom] / (us * 1e-6) / 1e12
      traffic, traffic_src, mfma_busy = None, None, None
      try:   # HBM bytes per launch from the newest committed PMC passes
        pmc, pmc_path = pmc_traffic()
        if B == 32:
          traffic = pmc["pixels"][dom]["traffic_bytes"]
          mfma_busy = pmc["pixels"][dom].get("mfma_busy")
          traffic_src = ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                         "64-byte gathers: fetch not doubled, see its note; source fingerprint %s = this build's)"
                         % (pmc_path, pmc["source_fingerprint"][:12]))
      except LookupError as e:
        traffic_src = "none: %s" % e
      except Exception:
        pass
      roofline = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                  "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
                  "traffic_source": traffic_src,
                  # matrix-pipe busy fraction of this kernel from the committed SQ counter pass
                  # (SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES)); real HBM bytes / duration / HBM peak
                  "mfma_busy": mfma_busy,
                  "traffic_frac": round(traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                  "avg_launch_us": round(us, 1), "algorithmic_flops_per_launch": flops[dom],
                  "note": "kernel durations of this block: eager steps with the conv backward on one stream (stand-alone "
                          "kernels); the timed step runs each layer's weight gradient on a side stream beside its data "
                          "gradient" if (_side_saved and not tfm) else None,
                  "avg_launch_us_by_kernel": by_kernel,
                  "tflops_by_kernel": {k: round(flops[k] / (v[0] * 1e-6) / 1e12, 1) for k, v in cand.items()}}
  else:
    bytes_per_launch = D * G * H * H * 4          # W_hh streamed once per step, both directions
    flops_per_launch = 2.0 * B * D * G * H * H    # (B x H)·(H x G*H) per direction
    cand = {k: v for k, v in prof.items() if k.startswith("rnn_")}
    # per-STEP time of each direction (a pass kernel's launch covers T steps)
    per_step = {k: (v[0] / T_FRAMES if k in pass_names else v[0]) for k, v in cand.items()}
    if cand and max(per_step, key=per_step.get) in pass_names:
      # one launch per layer pass: per-step time = pass / T.  W_hh never leaves the CUs (registers + LDS of
      # a pair / cluster, bf16 hi/lo planes), so the only per-step HBM traffic is the gate pre-activations in
      # and the gates / state out; `achieved` still prices the step against the bytes a per-step launch re-streams
      # (SURVEY section 8d's accounting: D*G*H^2*4 per step), so the figure is comparable across rounds.
      dom = max(per_step, key=per_step.get)
      us_pass = cand[dom][0]
      us = per_step[dom]
      ach = bytes_per_launch / (us * 1e-6) / 1e9
      io_bytes = B * D * H * 4 * (G + G + 2)     # read G gate pre-activations, write G gates + y + extra per (b, d, unit)
      r_dom = rec if dom == "rnn_fwd_step_kernel" else rec_bwd
      pass_traffic, pass_traffic_src = io_bytes * T_FRAMES, "computed: the pass's global loads and stores (W_hh stays on-chip)"
      mfma_busy = None
      try:   # per-launch HBM bytes of this kernel from the newest committed PMC passes, when it has been profiled
        pmc, pmc_path = pmc_traffic()
        if B == 32 and layers == 1:
          pass_traffic = pmc[args.model][pass_names[dom].split(" ")[0]]["traffic_bytes"]
          mfma_busy = pmc[args.model][pass_names[dom].split(" ")[0]].get("mfma_busy")
          pass_traffic_src = ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 fetch correction; "
                              "one launch = %d steps)" % (pmc_path, T_FRAMES))
      except Exception:
        pass
      cus = members if r_dom == "split" else 1
      real_gbs = pass_traffic / (us_pass * 1e-6) / 1e9
      # What bounds a one-launch recurrence is neither peak: W_hh never moves, so the HBM roofline prices bytes the kernel
      # does not stream, and the matrix pipe is a tenth busy.  A step IS one store -> L2 -> load exchange between the
      # cluster's members plus the cell, so the honest figure is us per step against that round trip (EXCHANGE_FLOOR_US).
      # `achieved` / `frac` are the REAL fabric bytes of the pass over its duration against the HBM peak; the notional
      # figure of rounds 1-5 (a step priced against the D*G*H^2*4 bytes a per-step launch would re-stream, SURVEY 8d's
      # accounting, comparable across rounds) is kept under `restream_equivalent`.
      roofline = {"bound": "hbm", "kernel": pass_names[dom],
                  "achieved": round(real_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(real_gbs / HBM_PEAK_GBS, 4),
                  "traffic": pass_traffic, "traffic_source": pass_traffic_src,
                  "traffic_frac": round(real_gbs / HBM_PEAK_GBS, 4),
                  "exchange_latency": {"us_per_step": round(us, 3), "floor_us_per_step": EXCHANGE_FLOOR_US,
                                       "floor_over_measured": round(EXCHANGE_FLOOR_US / us, 3),
                                       "what": "the bound of this kernel: one store -> L2 -> load round trip between the "
                                               "members of a cluster per time step, which nothing overlaps (DESIGN.md 4.1); "
                                               "the rest of a step is the recurrent product's MFMAs, the cell and two barriers"},
                  "restream_equivalent": {"achieved": round(ach, 1), "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                          "algorithmic_bytes_per_step": bytes_per_launch,
                                          "what": "notional: W_hh bytes a per-step launch re-streams / time per step (SURVEY 8d)"},
                  "mfma_busy": mfma_busy,
                  "avg_launch_us": round(us_pass, 1), "us_per_step": round(us, 3), "steps_per_launch": T_FRAMES,
                  "us_per_step_by_direction": {("forward" if k == "rnn_fwd_step_kernel" else "backward"): round(v, 3)
                                               for k, v in per_step.items()},
                  "algorithmic_bytes_per_step": bytes_per_launch,
                  "w_hh_residency": "registers + LDS of %d compute units per %s, bf16 %s; re-streamed bytes per step: 0"
                                    % (cus, "(sample, direction)" if cus <= 2 else "(direction, 8 samples)",
                                       "hi + lo planes (fp32-faithful)" if r_dom == "split" else "single plane"),
                  "launches_per_step": sum(layers if k in pass_names else layers * T_FRAMES for k in cand),
                  "avg_launch_us_by_kernel": by_kernel,
                  "mfma_bf16_tflops": round((4 if r_dom == "split" else 1) * flops_per_launch / (us * 1e-6) / 1e12, 2)}
    elif cand:
      dom = max(cand, key=lambda k: cand[k][0])
      us = cand[dom][0]
      ach = bytes_per_launch / (us * 1e-6) / 1e9
      traffic, traffic_src = None, None
      try:   # HBM bytes per launch from the newest committed PMC passes
        pmc, pmc_path = pmc_traffic()
        if B == 32 and layers == 1:
          traffic = pmc[args.model][dom]["traffic_bytes"]
          traffic_src = ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 fetch "
                         "correction)" % pmc_path)
      except Exception:
        pass
      roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                  "traffic_source": traffic_src, "avg_launch_us": round(us, 3),
                  "algorithmic_bytes_per_launch": bytes_per_launch,
                  "launches_per_step": 2 * T_FRAMES * layers, "avg_launch_us_by_kernel": by_kernel,
                  "mfma_f32_tflops": round(flops_per_launch / (us * 1e-6) / 1e12, 2),
                  "mfma_f32_frac": round(flops_per_launch / (us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
  res["roofline"] = roofline
  if tfm:
    res["workload"] = ("regime X-transformer (BASELINE configs[4]; frontend AND encoder build-defined, no reference "
                       "symbol): uint8 clips (B=%d,T=75,3,96,96) -> STCNN x3 (bf16 MFMA) -> Linear(3456,256) + "
                       "sinusoidal positions -> 4 x TransformerEncoderLayer(256, 4 heads, ff 1024, post-LN; self-attention "
                       "fused on the bf16 matrix cores, projections split-bf16) -> "
                       "Linear(256,65) -> masked log-softmax -> CTC 'mean' (L=30+EOS) -> backward -> clip 50 -> "
                       "Adam 1e-4" % B)
  elif pixels:
    res["workload"] = ("regime X (BASELINE metric shape, frontend build-defined: the reference has no conv "
                       "stage): uint8 clips (B=%d,T=75,3,96,96) -> STCNN x3 (bf16 MFMA implicit GEMM, fp32 "
                       "accumulate) -> %d-layer Bi%s-%d (%s) -> Linear(%d,65) -> masked log-softmax -> CTC "
                       "'mean' (L=30+EOS) -> backward -> clip_grad_norm 50 -> Adam 1e-4"
                       % (B, layers, rnn_type, H,
                          "input projection bf16x3; recurrence in one launch per pass, bf16 hi+lo planes (fp32-faithful)"
                          if rec == "split" else "input projection bf16x3, fp32 recurrence", D * H))
  elif attn:
    res["workload"] = ("regime R+decoder (the reference's whole train step): landmarks (B=%d,T=75,68,3) f32 -> "
                       "1-layer %s%s-%d%s AND CharDecodingStep x31 "
                       "(%s-%d, char_dim %d, %s attention over the 75 encoder states, teacher forced, "
                       "multinomial sample per step) + NLL -> backward -> per-module clip_grad_norm 50 -> Adam 1e-4"
                       % (B, "Bi" if bi else "", rnn_type, H,
                          " -> Linear(%d,65) + CTC 'mean' (L=30+EOS)" % (D * H) if enc.enable_ctc else " (enable_ctc False)",
                          rnn_type, D * H, args.char_dim, args.attention))
    dec_kind = L.lr_rnn_pair_supported({"GRU": 0, "LSTM": 1, "RNN": 2}[rnn_type], B, LABEL_LEN + 1, args.char_dim + D * H, D * H, 1)
    dec_members = (D * H + 31) // 32 if D * H <= (864 if rnn_type == "GRU" else 768) else (D * H + 15) // 16 // 2 * 2 + ((D * H + 15) // 16 % 2) * 2
    dec_launches = L.lr_rnn_pass_launches({"GRU": 0, "LSTM": 1, "RNN": 2}[rnn_type], B, LABEL_LEN + 1, args.char_dim + D * H, D * H, 1)
    if dec_kind == 2 and D * H > 1152:
      res["decoder_recurrence"] = ("%d launch(es) per loop pass: the 24 x 8 grid of 192 CUs (lr_rnn_grid.hip), 64 samples per launch"
                                   % dec_launches)
    else:
      res["decoder_recurrence"] = ("%d launch(es) per loop pass (cluster of %d CUs per 8 samples)" % (dec_launches, dec_members)
                                   if dec_kind == 2
                                   else "one launch per decoder step (no one-launch kernels for %s-%d)" % (rnn_type, D * H))
  else:
    res["workload"] = ("regime R (reference-faithful): landmarks (B=%d,T=75,68,3) f32 -> %d-layer Bi%s-%d (%s) -> "
                       "Linear(%d,65) -> masked log-softmax -> CTC 'mean' (L=30+EOS) -> backward -> "
                       "clip_grad_norm 50 -> Adam 1e-4"
                       % (B, layers, rnn_type, H,
                          {"f32": "recurrence: one fp32-MFMA launch per time step",
                           "split": "recurrence: %d launch(es) per layer pass, W_hh and state as bf16 hi+lo planes held by a "
                                    "cluster of %d CUs per (direction, %d samples), fp32 accumulation — fp32-faithful"
                                    % (L.lr_rnn_pass_launches({"GRU": 0, "LSTM": 1, "RNN": 2}[rnn_type], B, T_FRAMES, frame_dim, H, D),
                                       members, 16 if (17 <= members <= 24 and (B + 7) // 8 * D > 8) else 8)}[rec], D * H))
  return res


def scaling_model(legs, dist_overhead_ms=0.06):
  """MODELLED, NOT MEASURED (gpurun reaches one GPU; the driver's 8-GPU runs are the measurement): what the data-parallel
  pixel step would cost on 2 / 4 / 8 GPUs of one node, from what CAN be measured on one — the per-rank step at the batch
  sizes a weak (B = 32 per GPU) and a strong (configs[3]: 64 clips in total) run put on a rank, the conv backward window
  the large bucket hides under, the one-rank price of the exchange path — and an alpha-beta model of RCCL's ring
  all-reduce over xGMI whose two constants are ASSUMPTIONS, given as a slow and a fast case.

  legs: {B: {"ms": step time, "conv_backward_us": sum of the stand-alone conv backward kernels}}."""
  enc_bytes, conv_bytes = 27.7e6, 1.3e6            # distributed.GradSync.groups_for_pixel_model's two buckets
  cases = {"slow": {"busbw_GBps": 120.0, "alpha_us": {2: 30.0, 4: 45.0, 8: 60.0}},     # one ring, one xGMI link at ~0.8 of 153 GB/s
           "fast": {"busbw_GBps": 300.0, "alpha_us": {2: 20.0, 4: 30.0, 8: 40.0}}}     # several rings over the 7 links
  def t_ar(n, nbytes, case):
    return case["alpha_us"][n] * 1e-3 + 2.0 * (n - 1) / n * nbytes / (case["busbw_GBps"] * 1e9) * 1e3    # ms
  def exposed(n, B, case):
    window = legs[B]["conv_backward_us"] * 1e-3    # the [encoder] bucket leaves when the conv backward starts and hides under it
    return max(0.0, t_ar(n, enc_bytes, case) - window) + t_ar(n, conv_bytes, case) + dist_overhead_ms
  out = {"label": "MODELLED, NOT MEASURED", "regime": "pixels",
         "measured_on_one_gpu": {str(b): {"ms_per_step": round(v["ms"], 4), "conv_backward_window_ms": round(v["conv_backward_us"] * 1e-3, 3)}
                                 for b, v in sorted(legs.items())},
         "assumptions": {
             "all_reduce": "T(n, S) = alpha(n) + 2 (n - 1) / n * S / busbw; constants NOT measured here: " + json.dumps(cases),
             "buckets": "[conv] 1.3 MB leaves at the end of backward (fully exposed); [encoder] 27.7 MB leaves when the conv "
                        "backward starts and is exposed only where it outlasts that window (distributed.groups_for_pixel_model)",
             "exchange_path_overhead_ms": dist_overhead_ms,
             "exchange_path_overhead_source": "LIPREADING_BENCH_FORCE_DIST=1 against the plain step on one GPU "
                                              "(profiles/r05_bench_forcedist.json: +0.04 .. +0.07 ms)",
             "not_modelled": "RCCL's ring kernels sharing compute units and HBM with the conv backward they hide under; "
                             "stragglers; host launch jitter across ranks; the 'mean' quirk of ctc_loss on ragged shards"},
         "weak_efficiency_B32_per_gpu": {}, "strong_efficiency_global_batch_64": {}}
  for n in (2, 4, 8):
    w, st = {}, {}
    for name, case in cases.items():
      w[name] = round(legs[32]["ms"] / (legs[32]["ms"] + exposed(n, 32, case)), 3)
      b = 64 // n
      if b in legs and 64 in legs:
        st[name] = round(legs[64]["ms"] / (n * (legs[b]["ms"] + exposed(n, b, case))), 3)
    out["weak_efficiency_B32_per_gpu"][str(n)] = w
    out["strong_efficiency_global_batch_64"][str(n)] = st or None
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--model", choices=sorted(MODELS), default="gru256")
  ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
  ap.add_argument("--regime", choices=["pixels", "landmarks", "landmarks_attn", "pixels_tfm", "both", "all"], default="all",
                  help="all (default): headline = pixels (the metric's (B,75,3,96,96) shape); the "
                       "reference-faithful landmarks regimes (encoder+CTC, and the whole step with the "
                       "attention decoder) are reported under 'regimes'")
  ap.add_argument("--layers", type=int, default=None, help="recurrent layers (default 1; 2 for pixels)")
  ap.add_argument("--no-graph", action="store_true",
                  help="launch every kernel eagerly instead of replaying a captured hipGraph")
  ap.add_argument("--repeats", type=int, default=5,
                  help="how many times the timed region of exactly --steps steps is repeated (median reported)")
  ap.add_argument("--char-dim", type=int, default=300, help="landmarks_attn: the decoder's character embedding size "
                  "(config/defaults.txt: 300; config/train/attn/attention_type: 256)")
  ap.add_argument("--attention", default="1_layer_nn", help="landmarks_attn: attention_type of CharDecodingStep")
  ap.add_argument("--no-ctc", action="store_true", help="landmarks_attn: enable_ctc False (config/defaults.txt)")
  ap.add_argument("--rnn-dropout", type=float, default=0.0,
                  help="landmarks_attn: rnn_dropout of the flag file (config/archive/experiments/ecd/*: 0.3).  With the one "
                       "recurrent layer every shipped config has, nn.LSTM / nn.GRU apply no dropout (it acts BETWEEN "
                       "layers): accepted so that a config can be quoted as shipped, and passed on to both modules")
  ap.add_argument("--model-scaling", action="store_true",
                  help="one GPU only: also time the pixel step at the per-rank batches of a strong-scaling run (8, 16, 64) and "
                       "print a MODELLED (not measured) 2/4/8-GPU weak and strong efficiency under `scaling_model`")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--cpu-budget", type=float, default=12.0,
                  help="seconds of timed CPU-oracle steps per regime (never fewer than 10 steps)")
  args = ap.parse_args()

  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    # started as plain `python bench.py --gpus N`: become the launcher of N ranks (one per GPU)
    sys.exit(launch_ranks(args.gpus))
  if os.environ.get("LIPREADING_BENCH_PROBE") == "1":
    return probe_main(args)

  import torch
  import torch.distributed as dist

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus and not (args.gpus == 1 and world == 1):
    sys.exit("bench.py --gpus %d was started inside a %d-rank group (WORLD_SIZE); the two must agree"
             % (args.gpus, world))
  assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the product path)"
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  if DIST_ON:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

  order = {"both": ["pixels", "landmarks"],
           "all": ["pixels", "landmarks", "landmarks_attn", "pixels_tfm"]}.get(args.regime, [args.regime])
  results = [run_regime(args, r, world, rank, dev) for r in order]
  # the reference-faithful regime once more on the OTHER recurrence: the per-step fp32 launches (every shape's fallback)
  options = {}
  if "landmarks" in order and args.regime in ("all", "landmarks"):
    options["f32"] = run_regime(args, "landmarks", world, rank, dev, recurrence="f32")
  if rank == 0:
    head = results[0]
    out = {
        "metric": "training frames/sec at (B,75,3,96,96) + CTC-loss parity, 1/2/4/8 MI355X",
        "value": head["value"], "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE_NOTE[head["regime"]] if head.get("recurrence", "split") != "f32" or head["regime"] == "pixels_tfm"
                 else DTYPE_NOTE[head["regime"]].replace(DTYPE_SPLIT, "f32 recurrence (exact-fp32 MFMA, one launch per step)"),
        "data": "synthetic",
        "config": {"workload": head["workload"], "regime": head["regime"], "per_gpu_batch": args.batch,
                   "global_batch": world * args.batch, "seq_len": T_FRAMES, "parallelism": "dp%d" % world,
                   "launch_probe": head.get("launch_choice"),
                   "launch": ("hipGraph replay of the product's step (lipreading_amd.train.StepGraphs): "
                              + ("forward+backward captured, gradient exchange and optimiser launched after the replay"
                                 if DIST_ON else "zero_grad, forward, loss, backward, clip and Adam in one graph"))
                             if head["use_graph"]
                             else (head.get("graph_note") or
                                   ("eager launches; gradient all-reduce overlapped with backward on a side stream"
                                    if DIST_ON else "eager"))},
        "final_loss": round(head["loss"], 6), "skipped_last": head["skipped"],
        "timing": {"repeats": head["repeats"], "ms_per_step_median": head["ms_per_step"],
                   "ms_per_step_min": head["ms_per_step_min"],
                   "note": "the timed region of exactly `steps` steps is repeated back to back; value is the median repeat"},
        "rccl_ranks": head["rccl_ranks"], "all_reduce_buckets": head["all_reduce_buckets"],
        "pair_errors": sum(r.get("pair_errors", 0) for r in results),
        "roofline": head["roofline"], "ctc": head.get("ctc"),
    }
    if head.get("decoder_recurrence"):
      out["config"]["decoder_recurrence"] = head["decoder_recurrence"]
    if len(results) > 1:
      out["regimes"] = {r["regime"]: {"value": r["value"], "unit": "frames/s", "ms_per_step": r["ms_per_step"],
                                      "ms_per_step_min": r["ms_per_step_min"],
                                      "workload": r["workload"], "final_loss": round(r["loss"], 6),
                                      "all_reduce_buckets": r["all_reduce_buckets"],
                                      "roofline": r["roofline"], "ctc": r.get("ctc")} for r in results[1:]}
    if options:
      lm = out["regimes"]["landmarks"] if "landmarks" in out.get("regimes", {}) else (out if head["regime"] == "landmarks" else None)
      if lm is not None:
        base_loss = results[order.index("landmarks")]["loss"]
        notes = {"f32": "VideoEncoder.recurrence = 'f32': one exact-fp32 MFMA launch per time step (the fallback of every shape)"}
        lm["other_recurrences"] = {
            name: {"note": notes[name], "value": o["value"], "unit": "frames/s", "ms_per_step": o["ms_per_step"],
                   "final_loss": round(o["loss"], 6), "final_loss_delta_vs_default": round(o["loss"] - base_loss, 7)}
            for name, o in options.items()}
    if world == 1 and args.model_scaling and not DIST_ON:
      import copy
      legs = {}
      conv_bwd = ("conv2_dgrad", "conv3_dgrad", "conv1_wgrad", "conv2_wgrad", "conv3_wgrad")
      for b in (8, 16, 32, 64):
        if b == args.batch and head["regime"] == "pixels":
          r = head
        else:
          a2 = copy.copy(args)
          a2.batch = b
          r = run_regime(a2, "pixels", world, rank, dev)
        by = (r.get("roofline") or {}).get("avg_launch_us_by_kernel") or {}
        legs[b] = {"ms": r["ms_per_step"], "conv_backward_us": sum(by.get(k, 0.0) for k in conv_bwd)}
      out["scaling_model"] = scaling_model(legs)
    if world == 1 and not args.no_cpu_baseline:
      # parity first (the metric's "+ CTC-loss parity"): HIP vs oracle on identical inputs and weights
      by_regime = {r["regime"]: r for r in results}
      par = {}
      for rg in ("pixels", "landmarks", "pixels_tfm"):
        if rg in by_regime:
          par[rg] = parity_block(rg, args.model, by_regime[rg]["layers"], args.batch, dev,
                                 train_steps=50 if rg != "pixels_tfm" else 0)
      if "landmarks_attn" in by_regime:
        par["landmarks_attn"] = parity_block_attn(args.model, args.batch, dev, args.char_dim, args.attention, not args.no_ctc)
      out["parity"] = par.get(head["regime"])
      for rg, blk in par.items():
        if rg != head["regime"] and "regimes" in out and rg in out["regimes"]:
          out["regimes"][rg]["parity"] = blk
      out["cpu_baseline"] = cpu_baseline(head["regime"], args.model, head["layers"], args.batch, args.cpu_budget,
                                         char_dim=args.char_dim, attention=args.attention,
                                         use_ctc=not (head["regime"] == "landmarks_attn" and args.no_ctc))
      if "regimes" in out and "landmarks" in out["regimes"] and head["regime"] != "landmarks":
        out["regimes"]["landmarks"]["cpu_baseline"] = cpu_baseline(
            "landmarks", args.model, by_regime["landmarks"]["layers"], args.batch, args.cpu_budget)
  if DIST_ON:
    dist.destroy_process_group()
  # the JSON line is the LAST thing on stdout: RCCL's version banner sits in the C stdio buffer until
  # it is flushed, which otherwise happens at exit, after Python's own prints
  sys.stdout.flush()
  try:
    ctypes.CDLL(None).fflush(None)
  except Exception:
    pass
  if rank == 0:
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
  main()
