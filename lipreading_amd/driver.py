"""The caller on the other side of the hot path: what `src/scripts/train.py` does around
`train()` / `eval()`, so the reference's `config/` flag files drive the MI355X path unchanged.

  python -m lipreading_amd.driver config/train/attn/attention_type --data=<name> [--flag=value ...]

Mirrors (reference file:line, read for behaviour only):
  _init_models   train.py:57-79    VideoEncoder(...) + CharDecodingStep(...), same keyword arguments
  restore        train.py:82-132   name-and-shape tolerant checkpoint load
  train          train.py:134-320  flags and defaults :134-167; epoch loop: patience / annealing
                                   (lr /= 5, reload best weights) :256-268, linear teacher-forcing
                                   decay :270-272, Adam re-created every epoch :275-276, eval on
                                   val/train, best_{encoder,decoder}.pth :287-288
  flag files     utils/cmd_line.py: one `--name=value` per line; booleans as `--flag=True|False`
Out of scope here (the reference's control plane, not the path): tensorboard logging, sample
printing and confusion-matrix plots (train.py:296-318), sentence re-segmentation (needs spaCy).
"""
import os
import sys
import time

import numpy as np
import torch

DEFAULTS = dict(  # train.py:134-167
    data="StephenColbert/medium_no_vtx1", labels="labels.json", sentence_dataset=False,
    occlussion_threshold=0.8, train_split=0.8, num_workers=1, refresh=False,
    patience=10, batch_size=4, learning_rate=1e-4, annealings=2, enable_ctc=False, grad_norm=50,
    tr_epochs=50, max_tfr=0.9, min_tfr=0.0,
    num_layers=1, frame_dim=68 * 3, hidden_size=700, char_dim=300,
    rnn_type='LSTM', attention_type='1_layer_nn', attn_hidden_size=-1, bidirectional=False,
    rnn_dropout=0.0, seed=123456, cuda=False,
    # not reference flags: where data/ and weights/ live, a cap for smoke runs, and the encoder+CTC-only
    # loop (no attention decoder; error = greedy-decoded CER) that the archived trainer's flag files select
    root=".", max_epochs=None, ctc_only=False, step_graphs=True,
    # build-defined pixel regime (BASELINE configs[1] "3Dconv+BiGRU+CTC", configs[4] "transformer encoder over
    # per-frame conv features"): --frontend=conv3d reads a dataview written with frames (u8 frames + landmarks),
    # crops the mouth on the device (lr_lip_crop_u8) and puts the STCNN frontend in front of the sequence encoder;
    # --encoder=transformer swaps the recurrent encoder for the transformer (hidden_size = d_model, num_layers
    # layers, --nhead heads).  Either one selects the encoder+CTC loop with greedy CER (no attention decoder).
    frontend="none", encoder="rnn", nhead=4, crop_size=96,
)


def _coerce(name, text, like):
  if isinstance(like, bool):
    if text.lower() in ("true", "1", "yes", ""):
      return True
    if text.lower() in ("false", "0", "no"):
      return False
    raise ValueError("--%s expects True/False, got %r" % (name, text))
  if like is None:
    return None if text.lower() == "none" else int(text)
  if isinstance(like, int):
    return int(float(text)) if text.lower() != "none" else None
  if isinstance(like, float):
    return float(text)
  return text


# Flags of the reference's config/ files that are not parameters of the live train() (train.py:134-167):
#  - `--teacher_forcing_ratio` (config/train/attn/*, config/archive/experiments/ecd/*) predates the
#    max_tfr/min_tfr schedule; it maps to max_tfr.
#  - the archived trainer's set (archive/train_model.py:186-205; config/train/micro and
#    config/train/test_train_nano — BASELINE configs[0]/[1] — are written in it): the ones with a live
#    counterpart are renamed, the rest (SGD momentum, checkpointing, tensorboard) have no meaning on
#    this path and are accepted with a warning.  That trainer is CTC-only, so its files switch enable_ctc on.
RENAMED = {"teacher_forcing_ratio": "max_tfr", "dataset": "data", "batch": "batch_size", "epochs": "max_epochs",
           "hidden_layers": "num_layers", "max_norm": "grad_norm"}
IGNORED = {"momentum", "anneal", "annealing", "checkpoint", "tensorboard", "continue_from", "silent"}
ARCHIVED_ONLY = {"dataset", "batch", "epochs", "hidden_layers", "max_norm"} | IGNORED


def parse_flags(argv, defaults=DEFAULTS):
  """Positional arguments are flag files (one or several `--name=value` tokens per line, as under
  the reference's config/); later `--name=value` arguments override them.  Unknown flags are an
  error, as in the reference's argument parser; deprecated / archived ones (RENAMED, IGNORED) are
  accepted with a warning."""
  tokens = []
  for a in argv:
    if a.startswith("--"):
      tokens.append(a)
    else:
      with open(a) as f:
        tokens += [t for t in f.read().replace("\n", " ").split(" ") if t.startswith("--")]
  out = dict(defaults)
  archived = False
  explicit = set()
  for t in tokens:
    name, _, text = t[2:].partition("=")
    archived = archived or name in ARCHIVED_ONLY
    if name in IGNORED:
      print("warning: --%s has no meaning on this path; ignored" % name, file=sys.stderr)
      continue
    if name in RENAMED:
      print("warning: --%s is read as --%s" % (name, RENAMED[name]), file=sys.stderr)
      name = RENAMED[name]
    if name not in out:
      raise SystemExit("unknown flag --%s" % name)
    out[name] = _coerce(name, text, defaults[name])
    explicit.add(name)
  if isinstance(out.get("rnn_type"), str):
    out["rnn_type"] = out["rnn_type"].upper()          # the archived files write `gru`
  if archived:
    # archive/train_model.py trains a CTC-only model (no attention decoder) and reports greedy CER/WER; its
    # LipReader is bidirectional throughout.  These are that trainer's DEFAULTS: a flag the command line or the
    # file sets itself (e.g. --bidirectional=False) wins.
    for name, value in (("enable_ctc", True), ("ctc_only", True), ("bidirectional", True)):
      if name not in explicit:
        out[name] = value
  if out.get("frontend") not in ("none", "conv3d"):
    raise SystemExit("--frontend must be none or conv3d")
  if out.get("encoder") not in ("rnn", "transformer"):
    raise SystemExit("--encoder must be rnn or transformer")
  if out["frontend"] != "none" or out["encoder"] != "rnn":
    # the build-defined regimes are encoder + CTC (BASELINE configs[1], [4]); no attention decoder behind them
    for name in ("enable_ctc", "ctc_only"):
      if name not in explicit:
        out[name] = True
  return out


def init_models(char2idx, num_layers, frame_dim, hidden_size, char_dim, enable_ctc, rnn_type,
                attention_type, attn_hidden_size, bidirectional, rnn_dropout, device):
  """train.py:57-79 with the HIP-backed modules."""
  from .attention_decoder import CharDecodingStep
  from .encoder import VideoEncoder
  encoder = VideoEncoder(frame_dim, hidden_size, rnn_type=rnn_type, num_layers=num_layers,
                         bidirectional=bidirectional, rnn_dropout=rnn_dropout, enable_ctc=enable_ctc,
                         vocab_size=len(char2idx), char2idx=char2idx, device=device).to(device)
  decoding_step = CharDecodingStep(encoder, char_dim=char_dim, vocab_size=len(char2idx), char2idx=char2idx,
                                   rnn_dropout=rnn_dropout, attention_type=attention_type,
                                   attn_hidden_size=attn_hidden_size, device=device).to(device)
  return encoder, decoding_step


def init_pixel_model(char2idx, f, device):
  """BUILD-DEFINED (no reference symbol): [conv3d frontend ->] recurrent or transformer encoder -> CTC head.
  The sequence encoder keeps the reference's constructor arguments where it has them."""
  from .encoder import VideoEncoder
  pixels = f["frontend"] == "conv3d"
  frame_dim = f["frame_dim"]
  if pixels:
    from .frontend import ConvFrontend3D, PixelLipReader, feature_dim
    frame_dim = feature_dim(f["crop_size"], f["crop_size"])
  if f["encoder"] == "transformer":
    from .transformer import TransformerVideoEncoder
    enc = TransformerVideoEncoder(frame_dim, d_model=f["hidden_size"], nhead=f["nhead"], num_layers=f["num_layers"],
                                  dim_feedforward=4 * f["hidden_size"], enable_ctc=True, vocab_size=len(char2idx),
                                  char2idx=char2idx)
  else:
    enc = VideoEncoder(frame_dim, f["hidden_size"], rnn_type=f["rnn_type"], num_layers=f["num_layers"],
                       bidirectional=f["bidirectional"], rnn_dropout=f["rnn_dropout"], enable_ctc=True,
                       vocab_size=len(char2idx), char2idx=char2idx, device=device)
  model = PixelLipReader(enc, ConvFrontend3D()) if pixels else enc
  return model.to(device)


def restore(net, save_file, verbose=True):
  """Load every tensor of the checkpoint whose name exists in `net` with the same shape; report
  (not raise on) the rest — train.py:82-132.  Accepts the reference's `best_encoder.pth` /
  `best_decoder.pth` state_dicts: the HIP-backed modules keep the reference's key names
  (SURVEY.md 8b).  Returns (restored, ignored, untouched) name lists."""
  own = net.state_dict()
  ckpt = torch.load(save_file, map_location="cpu")
  restored, mismatched = [], []
  with torch.no_grad():
    for name, value in ckpt.items():
      if name not in own:
        continue
      if tuple(own[name].shape) != tuple(value.shape):
        mismatched.append(name)
        if verbose:
          print('\t\tShape mismatch for var', name, 'expected', tuple(own[name].shape), 'got', tuple(value.shape))
        continue
      own[name].copy_(value.data if isinstance(value, torch.nn.Parameter) else value)
      restored.append(name)
  ignored = sorted(set(ckpt.keys()) - set(restored))
  untouched = sorted(set(own.keys()) - set(restored))
  if verbose:
    print('\t\tRestored all variables' if not ignored else '\t\tDid not restore:\n\t' + '\n\t'.join(ignored))
    print('\t\tNo new variables' if not untouched
          else '\t\tInitialized but did not modify:\n\t' + '\n\t'.join(untouched))
    print('\tRestored %s' % save_file)
  return restored, ignored, untouched


def weights_path(root, data):
  """weights/<data>/<n>: a fresh numbered directory per run (utility.py getRelWeightsPath)."""
  base = os.path.join(root, "weights", data)
  os.makedirs(base, exist_ok=True)
  taken = [int(d) for d in os.listdir(base) if d.isdigit()]
  path = os.path.join(base, str(max(taken) + 1 if taken else 0))
  os.makedirs(path)
  return path


def _cer(correct, count):
  return float(count - correct) / count if count else 1.0


def run(**flags):
  """The training loop of train.py:134-320 on one MI355X.  Returns a summary dict."""
  from . import train as T
  from .data import make_collate_fn
  from .dataset import FrameCaptionDataset, make_loader, split_dataset
  from .optim import FlatParameters, FusedAdam
  f = dict(DEFAULTS)
  f.update(flags)
  if f["frontend"] != "none" or f["encoder"] != "rnn":
    # the build-defined regimes are encoder + CTC with greedy CER (parse_flags sets the same for flag files)
    f["enable_ctc"], f["ctc_only"] = flags.get("enable_ctc", True), flags.get("ctc_only", True)
  torch.manual_seed(f["seed"])
  rand = np.random.RandomState(seed=f["seed"])
  assert torch.cuda.is_available(), "the driver runs the HIP path: an MI355X is required (no CPU fallback)"
  device = torch.device("cuda")
  if not f["cuda"]:
    print("note: --cuda=False is ignored: this build has no CPU execution path")
  print("Initializing dataset '{}'".format(f["data"]))
  splits = split_dataset(f["root"], f["data"], f["train_split"], rand=rand)
  pixels = f["frontend"] == "conv3d"
  sets = [FrameCaptionDataset(f["root"], f["data"], name, ids, labels=f["labels"],
                              threshold=f["occlussion_threshold"], sentence_dataset=f["sentence_dataset"],
                              refresh=f["refresh"], pixels=pixels)
          for name, ids in zip(("train", "val", "test"), splits)]
  char2idx = sets[0].char2idx
  if pixels:
    # u8 frames + landmarks -> mouth crops (B, Tmax, 3, S, S) on the GPU (lr_lip_crop_u8)
    from .data import make_pixel_collate_fn
    collate = make_pixel_collate_fn(device, size=f["crop_size"])
  else:
    collate = make_collate_fn(device)   # padded on the GPU (lr_collate_pad_f32); lengths stay on the host
  train_loader, val_loader, test_loader = (make_loader(d, f["batch_size"], collate) for d in sets)
  print("Initializing model")
  ctc_only = bool(f["ctc_only"])
  if pixels or f["encoder"] == "transformer":
    assert ctc_only and f["enable_ctc"], "--frontend=conv3d / --encoder=transformer run the encoder+CTC loop"
    encoder, decoding_step = init_pixel_model(char2idx, f, device), None
  else:
    encoder, decoding_step = init_models(char2idx, f["num_layers"], f["frame_dim"], f["hidden_size"], f["char_dim"],
                                         f["enable_ctc"], f["rnn_type"], f["attention_type"], f["attn_hidden_size"],
                                         f["bidirectional"], f["rnn_dropout"], device)
  if ctc_only:
    assert f["enable_ctc"], "--ctc_only needs --enable_ctc"
    decoding_step = None
    flats = (FlatParameters(encoder),)
  else:
    flats = (FlatParameters(encoder), FlatParameters(decoding_step))
  weights_dir = weights_path(f["root"], f["data"])
  encoder_path = os.path.join(weights_dir, "best_encoder.pth")
  decoder_path = os.path.join(weights_dir, "best_decoder.pth")

  def error_of(loader):
    """The live loop's "CER" is the sampled-token mismatch rate of the attention decoder (train.py:287-288
    via eval's correct/count); without a decoder it is the greedy-decoded CER of the CTC head
    (decoder.py:64-73 on :182-197, as archive/train_model.py:351-357 composes them)."""
    if ctc_only:
      return T.greedy_cer(encoder, loader, device, char2idx)
    _, correct, count, _ = T.eval(encoder, decoding_step, loader, device, char2idx)
    return _cer(correct, count)

  opt = tuple(FusedAdam(fl, lr=f["learning_rate"]) for fl in flats)
  if pixels and hasattr(encoder, "encoder"):
    # (single process, one backward per step: the sequence encoder's share of the clip's sum of squares is taken beside
    # the conv backward, optim.FusedAdam.sum_squares_early)
    opt[0].sum_squares_early(encoder.encoder)
  # the step of every batch shape (B, Tmax, Lmax) is captured once as a hipGraph and replayed — in the landmark
  # regimes, whose steps are launch-bound (0.4-2 ms of 14-100 short launches).  The pixel regimes' steps are GPU-bound
  # (~45 launches, 2.3 ms) and replay no faster than eager launches (2.55 against 2.47 ms at the bench shape, bench.py's
  # launch_probe; with six hardware queues the replay is 3.4 ms, lipreading_amd/__init__.py): eager there.
  graphs = T.StepGraphs(enabled=bool(f["step_graphs"]) and not pixels)

  print("Initial evaluation...")
  val_cer = error_of(val_loader)
  print("\tCER: ", val_cer)
  best_val_cer, best_idx = 1.0, -1
  lr = f["learning_rate"]
  epochs, annealings = 0, 0
  history = []
  t0 = time.time()
  print("Beginning training loop")
  while val_cer < best_val_cer or annealings < f["annealings"]:
    if f["max_epochs"] is not None and epochs >= f["max_epochs"]:
      break
    print("Epoch {}:".format(epochs + 1))
    if epochs - best_idx > f["patience"]:
      annealings += 1
      lr /= 5
      print(f'\tAnnealing to {lr}')
      if os.path.isfile(encoder_path):
        restore(encoder, encoder_path)
        if not ctc_only:
          restore(decoding_step, decoder_path)
      best_idx = epochs
    tfr = max(f["min_tfr"], f["max_tfr"] - epochs / f["tr_epochs"])
    assert 0.0 <= tfr <= 1.0
    print(f'\tCurrent Teacher Forcing Ratio: {tfr}')
    for o in opt:
      o.reset(lr)     # what re-creating Adam every epoch does (:275-276): moments and step count start over
    dec_loss, ctc_loss = T.train(encoder, decoding_step, train_loader, opt[0] if ctc_only else opt, device, char2idx,
                                 teacher_forcing_ratio=tfr, grad_norm=f["grad_norm"], graphs=graphs)
    print(f'\tAVG Decoder Loss: {dec_loss}')
    print(f'\tAVG CTC Loss: {ctc_loss}')
    val_cer, train_cer = error_of(val_loader), error_of(train_loader)
    encoder.save_best_model(val_cer, encoder_path)
    if not ctc_only:
      decoding_step.save_best_model(val_cer, decoder_path)
    test_cer = error_of(test_loader)
    print(f'\tTrain CER: {train_cer}')
    print(f'\tVal CER: {val_cer}')
    print(f'\tTest CER: {test_cer}')
    stats = T.last_epoch_stats or {}
    history.append(dict(epoch=epochs, decoder_loss=dec_loss, ctc_loss=ctc_loss, train_cer=train_cer,
                        val_cer=val_cer, test_cer=test_cer, lr=lr, tfr=tfr,
                        # batches that updated nothing (the reference's `continue`, train_better_model.py:49-50) and how
                        # many of them because a one-launch recurrence timed out
                        skipped_batches=stats.get("skipped", 0), recurrence_faults=stats.get("recurrence_faults", 0)))
    if val_cer < best_val_cer:   # :339-341
      best_val_cer, best_idx = val_cer, epochs
    epochs += 1
  return dict(history=history, weights_dir=weights_dir, seconds=time.time() - t0, epochs=epochs,
              graph_captures=graphs.captures, graph_replays=graphs.replays, encoder=encoder,
              decoding_step=decoding_step, char2idx=char2idx, loaders=(train_loader, val_loader, test_loader))


def main(argv=None):
  flags = parse_flags(sys.argv[1:] if argv is None else argv)
  out = run(**flags)
  print("done: %d epochs in %.1f s, weights in %s" % (out["epochs"], out["seconds"], out["weights_dir"]))


if __name__ == "__main__":
  main()
