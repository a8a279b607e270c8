"""CPU: host-side mirror of the reference interface (no GPU, no kernels)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from lipreading_amd.data import default_char2idx
from lipreading_amd.decoder import Decoder, ctc_labels
from oracle import torch_oracle as O


@pytest.mark.parametrize("rnn_type", ["GRU", "LSTM"])
@pytest.mark.parametrize("layers,bi", [(1, True), (2, True), (2, False)])
def test_encoder_state_dict_and_init_match_torch_rnn(rnn_type, layers, bi):
  """Same key names, shapes and (given the seed) the same initial weights as the reference's
  nn.GRU/nn.LSTM + nn.Linear (better_model.py:47-51)."""
  from lipreading_amd.encoder import VideoEncoder
  torch.manual_seed(123456)
  ref = O.OracleVideoEncoder(204, 20, rnn_type=rnn_type, num_layers=layers, bidirectional=bi,
                             enable_ctc=True, vocab_size=64, char2idx=O.default_char2idx())
  torch.manual_seed(123456)
  enc = VideoEncoder(204, 20, rnn_type=rnn_type, num_layers=layers, bidirectional=bi,
                     enable_ctc=True, vocab_size=64, char2idx=default_char2idx())
  a, b = ref.state_dict(), enc.state_dict()
  assert list(a.keys()) == list(b.keys())
  for k in a:
    assert a[k].shape == b[k].shape, k
    assert torch.equal(a[k], b[k]), k
  assert enc.output_mask.tolist() == ref.output_mask.tolist()
  assert enc.adj_vocab_size == 65 and enc.best_error == 1


def test_encoder_rejects_cpu_tensors():
  from lipreading_amd import _C
  from lipreading_amd.encoder import VideoEncoder
  enc = VideoEncoder(204, 8, rnn_type='GRU', enable_ctc=True, vocab_size=64,
                     char2idx=default_char2idx())
  with pytest.raises(_C.LipReadingHipError):
    enc(torch.zeros(2, 3, 68, 3), torch.tensor([3, 3]))


def test_vocab_and_labels():
  c2i = default_char2idx()
  assert c2i == O.default_char2idx() and len(c2i) == 64
  labels = ctc_labels(c2i)
  assert labels == O.ctc_labels() and len(labels) == 65 and labels[0] == '_' and labels[5] == ' '


def test_wer_cer_edit_distance():
  d = Decoder(ctc_labels(default_char2idx()))
  assert d.cer("hello world", "helo wurld") == 2
  assert d.wer("the cat sat", "the cat sat down") == 1
  assert d.wer("a b c", "a x c") == 1
  assert d.cer("", "abc") == 3
  assert O.edit_distance("kitten", "sitting") == 3


def test_checkpoint_round_trip_with_reference_format(tmp_path):
  """N4: `best_encoder.pth` is a plain state_dict (better_model.py:114-122) that the reference
  restores by name and shape (src/scripts/train.py:95-119).  A checkpoint written by the
  reference-shaped oracle module loads into the HIP-backed module and vice versa."""
  from lipreading_amd.encoder import VideoEncoder
  torch.manual_seed(7)
  ref = O.OracleVideoEncoder(204, 12, rnn_type="LSTM", num_layers=2, bidirectional=True,
                             enable_ctc=True, vocab_size=64, char2idx=O.default_char2idx())
  path = tmp_path / "weights" / "best_encoder.pth"
  path.parent.mkdir()
  torch.save(ref.state_dict(), str(path))
  enc = VideoEncoder(204, 12, rnn_type="LSTM", num_layers=2, bidirectional=True, enable_ctc=True,
                     vocab_size=64, char2idx=default_char2idx())
  # the reference's restore(): copy every entry whose name and shape match
  own = enc.state_dict()
  loaded = torch.load(str(path))
  matched = 0
  for name, param in loaded.items():
    assert name in own and own[name].shape == param.shape, name
    own[name].copy_(param)
    matched += 1
  assert matched == len(own) == 18
  for k, v in enc.state_dict().items():
    assert torch.equal(v, ref.state_dict()[k])
  # and back: save_best_model writes a file the reference-shaped module loads strictly
  enc.best_error = 1
  out = tmp_path / "weights2" / "best_encoder.pth"
  enc.save_best_model(0.5, str(out))
  assert enc.best_error == 0.5
  enc.save_best_model(0.9, str(out))       # worse error: not overwritten, best_error kept
  assert enc.best_error == 0.5
  ref2 = O.OracleVideoEncoder(204, 12, rnn_type="LSTM", num_layers=2, bidirectional=True,
                              enable_ctc=True, vocab_size=64, char2idx=O.default_char2idx())
  ref2.load_state_dict(torch.load(str(out)), strict=True)


@pytest.mark.parametrize("attn,ah", [("none", -1), ("dot", -1), ("general", -1), ("1_layer_nn", -1), ("concat", 10)])
def test_decoder_state_dict_and_init_match_reference_shape(attn, ah):
  """CharDecodingStep: same keys, shapes and (given the seed) initial weights as the reference
  module structure (better_model.py:145-156)."""
  from lipreading_amd.attention_decoder import CharDecodingStep
  from lipreading_amd.encoder import VideoEncoder
  enc = VideoEncoder(204, 8, rnn_type='LSTM', bidirectional=True, enable_ctc=True, vocab_size=64,
                     char2idx=default_char2idx())
  torch.manual_seed(99)
  ref = O.OracleCharDecodingStep(16, 'LSTM', 1, 12, 64, O.default_char2idx(), attention_type=attn,
                                 attn_hidden_size=ah)
  torch.manual_seed(99)
  dec = CharDecodingStep(enc, 12, 64, default_char2idx(), attention_type=attn, attn_hidden_size=ah)
  a, b = ref.state_dict(), dec.state_dict()
  assert list(a.keys()) == list(b.keys())
  for k in a:
    assert torch.equal(a[k], b[k]), k
  assert dec.output_mask.tolist() == ref.output_mask.tolist() and dec.hidden_size == 16


def test_driver_flag_files_and_restore(tmp_path):
  """The reference's config/ flag-file format and its tolerant checkpoint restore (train.py:82-132)."""
  from lipreading_amd import driver
  from lipreading_amd.encoder import VideoEncoder
  cfg = tmp_path / "flags"
  cfg.write_text("--batch_size=16\n--enable_ctc=True\n--rnn_type=GRU --hidden_size=256\n--bidirectional=True\n"
                 "--learning_rate=3e-4\n--attention_type=dot\n")
  f = driver.parse_flags([str(cfg), "--hidden_size=128", "--grad_norm=5"])
  assert (f["batch_size"], f["enable_ctc"], f["rnn_type"], f["hidden_size"], f["bidirectional"]) == \
         (16, True, "GRU", 128, True)
  assert f["learning_rate"] == 3e-4 and f["grad_norm"] == 5 and f["attention_type"] == "dot"
  assert f["char_dim"] == 300 and f["seed"] == 123456          # untouched defaults of train.py:134-167
  with pytest.raises(SystemExit):
    driver.parse_flags(["--no_such_flag=1"])
  # restore(): a checkpoint written by the reference-shaped module (same keys), one tensor with the
  # wrong shape and one unknown name
  torch.manual_seed(0)
  ref = O.OracleVideoEncoder(204, 8, rnn_type='GRU', bidirectional=True, enable_ctc=True, vocab_size=64,
                             char2idx=O.default_char2idx())
  sd = dict(ref.state_dict())
  sd["output_proj.bias"] = torch.zeros(3)
  sd["extra.weight"] = torch.zeros(2)
  path = str(tmp_path / "best_encoder.pth")
  torch.save(sd, path)
  enc = VideoEncoder(204, 8, rnn_type='GRU', bidirectional=True, enable_ctc=True, vocab_size=64,
                     char2idx=default_char2idx())
  before = enc.output_proj.bias.detach().clone()
  restored, ignored, untouched = driver.restore(enc, path, verbose=False)
  assert "extra.weight" in ignored and "output_proj.bias" in ignored and untouched == ["output_proj.bias"]
  assert torch.equal(enc.output_proj.bias, before)
  for k, v in ref.state_dict().items():
    if k != "output_proj.bias":
      assert torch.equal(enc.state_dict()[k], v), k


def test_driver_accepts_the_deprecated_and_archived_flag_files(tmp_path):
  """The flag files under the reference's config/ drive the path unchanged: config/train/attn/* carry
  `--teacher_forcing_ratio` and bare booleans (`--enable_ctc`, `--cuda`); config/train/micro and
  config/train/test_train_nano (BASELINE configs[0]/[1]) are written in the archived trainer's flag set
  (archive/train_model.py:186-205).  The flag NAMES below are those files'."""
  from lipreading_amd import driver
  attn = tmp_path / "attention_type"
  attn.write_text("--data=StephenColbert/small\n--labels=labels.json\n--occlussion_threshold=0.8\n--train_split=0.8\n"
                  "--num_workers=1\n\n--patience=15\n--batch_size=4\n--learning_rate=3e-4\n--enable_ctc\n"
                  "--teacher_forcing_ratio=1.0\n--grad_norm=50\n\n--num_layers=1\n--frame_dim=204\n--hidden_size=512\n"
                  "--char_dim=256\n\n--rnn_type=LSTM\n--bidirectional\n--rnn_dropout=0\n\n--seed=123456\n--cuda\n")
  f = driver.parse_flags([str(attn), "--attention_type=dot"])
  assert f["max_tfr"] == 1.0 and f["enable_ctc"] is True and f["bidirectional"] is True and f["cuda"] is True
  assert (f["hidden_size"], f["char_dim"], f["rnn_type"], f["patience"]) == (512, 256, "LSTM", 15)
  micro = tmp_path / "micro"
  micro.write_text("--dataset=StephenColbert/micro\n--epochs=70\n--batch=5\n--train_split=0.8\n--num_workers=1\n"
                   "--hidden_size=800\n--hidden_layers=5\n--rnn_type=gru\n--cuda\n--learning_rate=3e-4\n--momentum=0.9\n"
                   "--max_norm=400\n--anneal=1.1\n--checkpoint\n--tensorboard\n--continue_from=0\n")
  f = driver.parse_flags([str(micro)])
  assert (f["data"], f["max_epochs"], f["batch_size"], f["num_layers"], f["grad_norm"]) == \
         ("StephenColbert/micro", 70, 5, 5, 400)
  assert f["rnn_type"] == "GRU" and f["enable_ctc"] is True and f["hidden_size"] == 800
  assert f["ctc_only"] is True and f["bidirectional"] is True
  with pytest.raises(SystemExit):
    driver.parse_flags([str(micro), "--still_unknown=1"])


def test_batch_loader_is_lazy_and_counts_batches():
  from lipreading_amd import dataset as DS
  calls = []
  loader = DS.make_loader(list(range(10)), 4, lambda items: calls.append(list(items)) or len(items))
  assert len(loader) == 3 and calls == []            # nothing is collated until iteration
  assert list(loader) == [4, 4, 2] and calls == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]
  assert list(loader) == [4, 4, 2] and len(calls) == 6      # each epoch collates afresh


def test_six_hardware_queues_are_set_for_ranks_only():
  """lipreading_amd/__init__.py: GPU_MAX_HW_QUEUES=6 in a process launched as one of several ranks (or as bench.py's
  one-rank stand-in), HIP's default otherwise (a single process's hipGraph-replayed pixel step takes 3.4 ms with six
  queues, 2.5 with four: profiles/r05_variants_ab.txt); a value the user set is never overwritten."""
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

  def queues(**env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "GPU_MAX_HW_QUEUES", "LIPREADING_BENCH_FORCE_DIST")}
    e.update(env)
    out = subprocess.run([sys.executable, "-c", "import os, lipreading_amd; print(os.environ.get('GPU_MAX_HW_QUEUES'))"],
                         capture_output=True, text=True, env=e, cwd=root, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    return out.stdout.strip().splitlines()[-1]
  assert queues() == "None"
  assert queues(WORLD_SIZE="1") == "None"
  assert queues(WORLD_SIZE="8") == "6"
  assert queues(LIPREADING_BENCH_FORCE_DIST="1") == "6"
  assert queues(WORLD_SIZE="8", GPU_MAX_HW_QUEUES="4") == "4"


REF_CONFIG = "/root/reference/config"


@pytest.mark.skipif(not os.path.isdir(REF_CONFIG), reason="build container only: the reference tree is not on the GPU box")
def test_every_reference_flag_file_parses_to_the_model_it_describes():
  """Every non-empty flag file the reference ships for its trainers — config/defaults.txt, config/train/**,
  config/archive/experiments/** — goes through driver.parse_flags unchanged (the reference reads them through
  src/utils/cmd_line.py:91-141 into src/scripts/train.py:134-167's keyword arguments), and the model the flags imply
  is the one this test reads out of the file with its own three-line tokenizer: encoder cell / width / directions,
  CTC head or not, batch, attention, and the decoder width CharDecodingStep derives (better_model.py:134-148:
  hidden = directions x encoder hidden).  config/gen_dataview/* drive the ETL, which is out of scope."""
  from lipreading_amd import driver
  files = []
  for root, _, names in os.walk(REF_CONFIG):
    if "gen_dataview" in root:
      continue
    files += [os.path.join(root, n) for n in names]
  files = sorted(f for f in files if os.path.getsize(f) > 0)
  assert len(files) >= 61                       # 52 archived experiments + defaults + 6 attention files + micro + nano
  truthy = lambda v: v.lower() not in ("false", "0", "no")
  shapes = {}
  for path in files:
    raw = {}
    for tok in open(path).read().split():
      if tok.startswith("--"):
        name, eq, val = tok[2:].partition("=")
        raw[name] = val if eq else "True"        # a bare flag is a boolean switch (cmd_line.py:110-118)
    f = driver.parse_flags([path])
    rel = os.path.relpath(path, REF_CONFIG)
    archived_trainer = "hidden_layers" in raw or "batch" in raw      # archive/train_model.py's flag names
    if "hidden_size" in raw:
      assert f["hidden_size"] == int(raw["hidden_size"]), rel
    if "rnn_type" in raw:
      assert f["rnn_type"] == raw["rnn_type"].upper(), rel
    layers = raw.get("num_layers", raw.get("hidden_layers"))
    if layers is not None:
      assert f["num_layers"] == int(layers), rel
    batch = raw.get("batch_size", raw.get("batch"))
    if batch is not None:
      assert f["batch_size"] == int(batch), rel
    if "attention_type" in raw:
      assert f["attention_type"] == raw["attention_type"], rel
    if "char_dim" in raw:
      assert f["char_dim"] == int(raw["char_dim"]), rel
    if archived_trainer:                         # CTC-only, bidirectional throughout (archive/train_model.py:186-205)
      assert f["enable_ctc"] is True and f["ctc_only"] is True and f["bidirectional"] is True, rel
    else:
      for flag in ("enable_ctc", "bidirectional"):
        if flag in raw:
          assert f[flag] is truthy(raw[flag]), (rel, flag)
    D = 2 if f["bidirectional"] else 1
    shapes.setdefault((f["rnn_type"], f["hidden_size"], D, f["enable_ctc"], D * f["hidden_size"]), []).append(rel)
  # the families the kernels are sized for: the archived experiments are all BiLSTM-768 with an
  # LSTM-1536 decoder, at batch 128
  ecd = [k for k in shapes if k[:3] == ("LSTM", 768, 2)]
  assert sum(len(shapes[k]) for k in ecd) == 52 and all(k[4] == 1536 for k in ecd)
  assert ("LSTM", 700, 1, False, 700) in shapes and ("LSTM", 512, 2, True, 1024) in shapes
  assert any(k[:2] == ("GRU", 800) for k in shapes)
