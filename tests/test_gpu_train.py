"""GPU parity: fused clip+Adam and the whole encoder+CTC optimisation step."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from tests.test_gpu_encoder import hip_encoder_from, make_pair
from tests.test_oracle_golden import _flatten

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
  assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
  return torch.device("cuda:0")


@pytest.mark.parametrize("max_norm", [None, 0.5, 1e6])
def test_fused_adam_matches_torch_adam(dev, max_norm):
  from lipreading_amd.optim import FlatParameters, FusedAdam
  torch.manual_seed(0)
  ref = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5))
  mod = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5))
  mod.load_state_dict(ref.state_dict())
  mod = mod.to(dev)
  flat = FlatParameters(mod)
  opt = FusedAdam(flat, lr=1e-2)
  ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
  g = torch.Generator().manual_seed(1)
  for it in range(5):
    x = torch.randn(8, 33, generator=g)
    ropt.zero_grad()
    ref(x).pow(2).sum().backward()
    if max_norm is not None:
      torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm)
    ropt.step()
    opt.zero_grad()
    mod(x.to(dev)).pow(2).sum().backward()
    opt.step(grad_norm=max_norm)
  for a, b in zip(mod.parameters(), ref.parameters()):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-5, atol=2e-6)
  assert int(opt.step_count[0].item()) == 5
  # skip flag: nothing moves, the step counter stays (train_better_model.py:49-50 `continue`)
  before = flat.data.clone()
  skip = torch.ones(1, dtype=torch.int32, device=dev)
  opt.step(grad_norm=max_norm, skip=skip)
  assert torch.equal(before, flat.data) and int(opt.step_count[0].item()) == 5
  assert opt.skipped_steps() == 1   # ... and the skip is counted on the device (train() reports it per epoch)
  # grad_scale = 1/world: same as averaging the gradient first
  opt2 = FusedAdam(flat, lr=1e-2)
  flat.grad.mul_(4.0)
  snap = flat.data.clone()
  opt2.step(grad_norm=None, grad_scale=0.25)
  flat.grad.div_(4.0)
  flat.data.copy_(snap)
  opt3 = FusedAdam(flat, lr=1e-2)
  snap2 = flat.data.clone()
  opt3.step()
  # both started from the same weights with the same effective gradient
  assert torch.allclose(snap, snap2)


@pytest.mark.parametrize("name", ["gru", "lstm"])
def test_ctc_step_fused_matches_reference_vectors(golden_step, dev, name):
  """train_better_model.py:31-32,46-48,74,78,80 through ctc_step (FlatParameters + FusedAdam),
  against the step captured from the reference (clip 5.0, Adam 1e-3)."""
  from lipreading_amd.optim import FlatParameters, FusedAdam
  from lipreading_amd.train import ctc_step
  case = golden_step[name]
  enc = hip_encoder_from(case, name.upper(), dev).train()
  opt = FusedAdam(FlatParameters(enc), lr=1e-3)
  lens = torch.tensor(case["lens"], device=dev)
  loss, status = ctc_step(enc, opt, torch.tensor(case["frames"], device=dev), lens,
                          torch.tensor(case["chars"], device=dev),
                          torch.tensor(case["char_lens"], device=dev), grad_norm=5.0)
  assert int(status.item()) == 0
  assert abs(loss.item() - float(case["loss"])) < 1e-4
  assert abs(float(opt.total_norm()) - float(case["total_norm"])) < 1e-3 * float(case["total_norm"])
  sd1 = _flatten(case["sd1"])
  for k, v in enc.state_dict().items():
    np.testing.assert_allclose(v.cpu().numpy(), sd1[k], rtol=1e-3, atol=2e-5, err_msg=k)


def test_training_reduces_loss_and_matches_oracle_trajectory(dev):
  """20 steps at the BASELINE shape family (reduced B): the HIP loss trajectory tracks the
  oracle's (same seed, same data) and goes down."""
  from lipreading_amd.optim import FlatParameters, FusedAdam
  from lipreading_amd.train import ctc_step
  import bench
  ref, enc = make_pair("GRU", 64, 1, True, dev)
  frames, fl, chars, cl = bench.synth_batch(8, 1)
  ropt = torch.optim.Adam(ref.parameters(), lr=1e-3)
  opt = FusedAdam(FlatParameters(enc), lr=1e-3)
  enc.train(); ref.train()
  fd, fld, cd, cld = (x.to(dev) for x in (frames, fl, chars, cl))
  hip, cpu = [], []
  for _ in range(20):
    cpu.append(float(O.encoder_ctc_step(ref, ropt, frames, fl, chars, cl, grad_norm=50)))
    loss, _ = ctc_step(enc, opt, fd, fld, cd, cld, grad_norm=50, max_len=75)
    hip.append(float(loss))
  assert hip[-1] < hip[0]
  np.testing.assert_allclose(hip, cpu, rtol=2e-3)
  assert abs(hip[0] - cpu[0]) < 1e-4 * max(1.0, cpu[0])


def test_train_and_eval_loops(dev):
  from lipreading_amd.data import default_char2idx, make_collate_fn
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.optim import FlatParameters, FusedAdam
  from lipreading_amd import train as T
  c2i = default_char2idx()
  rng = np.random.RandomState(0)
  items = []
  for n in sorted(rng.randint(20, 40, 12)):   # dataset is sorted by length (data_loader.py:93-98)
    cap = np.r_[1, rng.randint(4, 64, 6), 2]
    items.append((rng.randn(n, 68, 3).astype(np.float32), cap))
  collate = make_collate_fn(dev)
  loader = [collate(items[i:i + 4]) for i in range(0, 12, 4)]
  torch.manual_seed(0)
  enc = VideoEncoder(204, 32, rnn_type='GRU', bidirectional=True, enable_ctc=True, vocab_size=64,
                     char2idx=c2i).to(dev)
  opt = FusedAdam(FlatParameters(enc), lr=3e-3)
  first = T.train(enc, None, loader, opt, dev, c2i, grad_norm=50)[1]
  for _ in range(15):
    last = T.train(enc, None, loader, opt, dev, c2i, grad_norm=50)[1]
  assert last < first
  _, _, _, ev = T.eval(enc, None, loader, dev, c2i)
  assert np.isfinite(ev) and ev > 0
  cer = T.greedy_cer(enc, loader, dev, c2i)
  assert 0.0 <= cer <= 2.0


def test_nano_dataset_end_to_end(dev, tmp_path):
  """BASELINE configs[0] plumbing on the GPU path: synthetic dataview in the reference's on-disk
  format -> FrameCaptionDataset -> device collation -> train()/eval()/greedy CER."""
  from lipreading_amd import dataset as DS
  from lipreading_amd import train as T
  from lipreading_amd.data import make_collate_fn
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.optim import FlatParameters, FusedAdam
  root = str(tmp_path)
  DS.write_synthetic_dataview(root, "synthetic/nano", n_videos=3, captions_per_video=6, seed=1)
  tr, va, te = DS.split_dataset(root, "synthetic/nano", 0.8, np.random.RandomState(123456))
  ds = DS.FrameCaptionDataset(root, "synthetic/nano", "train", tr)
  loader = DS.make_loader(ds, 4, make_collate_fn(dev))
  assert len(loader) == (len(ds) + 3) // 4
  torch.manual_seed(123456)
  enc = VideoEncoder(204, 48, rnn_type='LSTM', bidirectional=True, enable_ctc=True,
                     vocab_size=len(ds.char2idx), char2idx=ds.char2idx).to(dev)
  opt = FusedAdam(FlatParameters(enc), lr=2e-3)
  losses = [T.train(enc, None, loader, opt, dev, ds.char2idx, grad_norm=50)[1] for _ in range(12)]
  assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[0], losses
  assert np.isfinite(T.eval(enc, None, loader, dev, ds.char2idx)[3])
  assert 0.0 <= T.greedy_cer(enc, loader, dev, ds.char2idx) <= 2.0


def test_driver_trains_from_a_flag_file_on_a_synthetic_dataview(dev, tmp_path):
  """train.py's loop (flag file -> datasets -> models -> epochs -> best_*.pth) on the HIP path,
  over a synthetic dataview in the reference's on-disk format."""
  from lipreading_amd import driver
  from lipreading_amd.dataset import write_synthetic_dataview
  root = str(tmp_path)
  write_synthetic_dataview(root, "synth/micro", n_videos=10, captions_per_video=6, seed=7)
  cfg = tmp_path / "flags"
  cfg.write_text("--data=synth/micro\n--batch_size=8\n--enable_ctc=True\n--rnn_type=GRU\n--hidden_size=32\n"
                 "--bidirectional=True\n--char_dim=16\n--attention_type=1_layer_nn\n--learning_rate=3e-3\n"
                 "--grad_norm=50\n--max_tfr=0.9\n--tr_epochs=50\n--cuda=True\n")
  flags = driver.parse_flags([str(cfg), "--root=" + root, "--max_epochs=3"])
  out = driver.run(**flags)
  h = out["history"]
  assert len(h) == 3 and all(np.isfinite([e["decoder_loss"], e["ctc_loss"]]).all() for e in h)
  assert h[-1]["decoder_loss"] < h[0]["decoder_loss"] and h[-1]["ctc_loss"] < h[0]["ctc_loss"]
  assert h[0]["tfr"] == 0.9 and abs(h[2]["tfr"] - (0.9 - 2 / 50)) < 1e-12
  import os
  enc_path = os.path.join(out["weights_dir"], "best_encoder.pth")
  assert os.path.isfile(enc_path) and os.path.isfile(os.path.join(out["weights_dir"], "best_decoder.pth"))
  # the checkpoint is a plain state_dict with the reference's key names
  sd = torch.load(enc_path, map_location="cpu")
  assert {"rnn.weight_ih_l0", "rnn.weight_hh_l0_reverse", "output_proj.weight"} <= set(sd)


@pytest.mark.parametrize("H", [64, 768])
def test_small_config_loss_and_cer_parity_with_the_cpu_path(dev, tmp_path, H):
  """BASELINE configs[2]: "small (23 videos), BiLSTM encoder + CTC, fp32, loss/CER parity vs CPU" — at H = 64 (on the
  per-step fp32 kernels: recurrence = 'f32') and at the config family's own size, BiLSTM-768 (config/archive/experiments/ecd/*), where the
  whole two-epoch loop runs on the one-launch cluster recurrence (lr_rnn_cluster.hip) and the split-bf16 weight
  gradients.
  A 23-video synthetic dataview in the reference's on-disk format; the HIP path (train(): collate ->
  VideoEncoder BiLSTM -> ctc_loss -> clip -> FusedAdam) and the oracle (same op sequence on stock torch
  CPU ops + torch.optim.Adam) start from the same weights and see the same batches for two epochs.
  Stated tolerances: per-epoch mean CTC loss 2e-3 relative (fp32 Adam trajectories drift apart slowly);
  greedy strings — hence the CER, decoder.py:64-73 over :182-197 — IDENTICAL when both sides decode with
  the same weights; the two separately trained models' CERs within 0.02."""
  from lipreading_amd import dataset as DS
  from lipreading_amd import train as T
  from lipreading_amd.data import make_collate_fn
  from lipreading_amd.decoder import GreedyDecoder, ctc_labels
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.optim import FlatParameters, FusedAdam
  root = str(tmp_path)
  DS.write_synthetic_dataview(root, "synthetic/small", n_videos=23, captions_per_video=4, seed=3)
  tr, va, te = DS.split_dataset(root, "synthetic/small", 0.8, np.random.RandomState(123456))
  ds = DS.FrameCaptionDataset(root, "synthetic/small", "train", tr)
  c2i = ds.char2idx
  loader = DS.make_loader(ds, 8, make_collate_fn(dev))
  torch.manual_seed(123456)
  ref = O.OracleVideoEncoder(204, H, rnn_type='LSTM', bidirectional=True, enable_ctc=True, vocab_size=len(c2i),
                             char2idx=c2i).train()
  enc = VideoEncoder(204, H, rnn_type='LSTM', bidirectional=True, enable_ctc=True, vocab_size=len(c2i), char2idx=c2i)
  enc.load_state_dict(ref.state_dict())
  enc = enc.to(dev)
  from lipreading_amd import _C
  assert _C.lib().lr_rnn_pair_supported(1, 8, 60, 204, H, 2) == 2
  if H == 64:
    enc.recurrence = 'f32'   # this case keeps covering the per-step kernels (round 4: every H <= 768 has the cluster kernels)
  _C.lib().lr_rnn_pair_errors()
  lr = 1e-3 if H == 64 else 3e-4
  opt = FusedAdam(FlatParameters(enc), lr=lr)
  ropt = torch.optim.Adam(ref.parameters(), lr=lr)
  inv = {v: k for k, v in c2i.items()}
  labels = ctc_labels(c2i)

  def oracle_cer(model):
    model.eval()
    dist, total, strings = 0, 0, []
    with torch.no_grad():
      for frames, lens, chars, cl in loader:
        lp = model(frames.cpu(), lens)[0]
        out, _ = O.greedy_decode(lp, lens, labels)
        for b in range(len(out)):
          want = ''.join(inv[int(c)] for c in chars[b, 1:int(cl[b]) - 1])
          hyp = out[b][0].replace('<EOS>', '')
          dist += O.edit_distance(hyp.replace(' ', ''), want.replace(' ', ''))
          total += len(want.replace(' ', ''))
          strings.append(hyp)
    model.train()
    return dist / max(total, 1), strings

  for epoch in range(2):
    hip_loss = T.train(enc, None, loader, opt, dev, c2i, grad_norm=50)[1]
    cpu_losses = []
    for frames, lens, chars, cl in loader:
      l = O.encoder_ctc_step(ref, ropt, frames.cpu(), lens, chars, cl, grad_norm=50)
      cpu_losses.append(0.0 if l is None else float(l))
    cpu_loss = float(np.mean(cpu_losses))
    assert abs(hip_loss - cpu_loss) <= 2e-3 * abs(cpu_loss), (epoch, hip_loss, cpu_loss)
  # same weights on both sides: identical greedy strings, identical CER
  assert _C.lib().lr_rnn_pair_errors() == 0 and T.last_epoch_stats["skipped"] == 0
  twin = O.OracleVideoEncoder(204, H, rnn_type='LSTM', bidirectional=True, enable_ctc=True, vocab_size=len(c2i),
                              char2idx=c2i)
  twin.load_state_dict({k: v.cpu() for k, v in enc.state_dict().items()})
  cer_twin, strings_twin = oracle_cer(twin)
  dec = GreedyDecoder(labels, blank_index=0)
  enc.eval()
  strings_hip = []
  with torch.no_grad():
    for frames, lens, chars, cl in loader:
      out, _ = dec.decode(enc(frames, lens.to(dev))[0], lens.to(dev))
      strings_hip += [o[0].replace('<EOS>', '') for o in out]
  assert strings_hip == strings_twin
  cer_hip = T.greedy_cer(enc, loader, dev, c2i)
  assert abs(cer_hip - cer_twin) < 1e-12, (cer_hip, cer_twin)
  cer_cpu, _ = oracle_cer(ref)
  assert abs(cer_hip - cer_cpu) <= 0.02, (cer_hip, cer_cpu)


def test_driver_runs_the_archived_ctc_only_flag_files(dev, tmp_path):
  """config/train/micro and config/train/test_train_nano (BASELINE configs[0]/[1]) are written in the
  archived trainer's flags: they select the encoder+CTC loop with greedy CER as the error."""
  from lipreading_amd import driver
  from lipreading_amd.dataset import write_synthetic_dataview
  root = str(tmp_path)
  write_synthetic_dataview(root, "StephenColbert/micro", n_videos=6, captions_per_video=5, seed=11)
  cfg = tmp_path / "micro"
  cfg.write_text("--dataset=StephenColbert/micro\n--epochs=70\n--batch=5\n--train_split=0.8\n--num_workers=1\n"
                 "--hidden_size=800\n--hidden_layers=5\n--rnn_type=gru\n--cuda\n--learning_rate=3e-4\n--momentum=0.9\n"
                 "--max_norm=400\n--anneal=1.1\n--checkpoint\n--tensorboard\n--continue_from=0\n")
  flags = driver.parse_flags([str(cfg), "--root=" + root, "--max_epochs=2", "--hidden_size=48", "--num_layers=2"])
  out = driver.run(**flags)
  h = out["history"]
  assert len(h) == 2 and all(np.isfinite(e["ctc_loss"]) for e in h) and h[1]["ctc_loss"] < h[0]["ctc_loss"]
  assert all(0.0 <= e["val_cer"] <= 2.0 for e in h)
  import os
  # best_encoder.pth appears once the greedy CER drops below 1 (better_model.py:114-122); never a decoder file
  assert not os.path.exists(os.path.join(out["weights_dir"], "best_decoder.pth"))


@pytest.mark.parametrize("with_decoder", [False, True])
def test_step_graphs_replay_is_bit_identical_to_eager(dev, with_decoder):
  """train(..., graphs=StepGraphs()): every batch shape's step (zero_grad -> forward -> loss -> backward ->
  clip -> Adam) is captured once as a hipGraph and replayed.  Same kernels, same order, same data: the
  weights after three epochs over three batch shapes (two of them repeated, so replays really happen)
  are BIT-identical to eager launches, with and without the attention decoder loop."""
  from lipreading_amd import train as T
  from lipreading_amd.attention_decoder import CharDecodingStep
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.optim import FlatParameters, FusedAdam
  c2i = default_char2idx()
  g = torch.Generator().manual_seed(4)
  batches = []
  for B, Tm, Cm in ((6, 20, 8), (6, 20, 8), (4, 17, 7), (6, 20, 8), (4, 17, 7), (5, 12, 6)):
    lens = torch.sort(torch.randint(Tm // 2, Tm + 1, (B,), generator=g))[0]
    lens[-1] = Tm
    frames = torch.randn(B, Tm, 68, 3, generator=g)
    for b in range(B):
      frames[b, lens[b]:] = 0
    cl = torch.randint(4, Cm + 1, (B,), generator=g)
    cl[0] = Cm
    chars = torch.zeros(B, Cm, dtype=torch.long)
    for b in range(B):
      n = int(cl[b])
      chars[b, 0], chars[b, n - 1] = 1, 2
      chars[b, 1:n - 1] = torch.randint(4, 64, (n - 2,), generator=g)
    batches.append((frames.to(dev), lens, chars, cl))
  results = []
  for use_graphs in (False, True):
    torch.manual_seed(9)
    enc = VideoEncoder(204, 16, rnn_type='GRU', num_layers=1, bidirectional=True, enable_ctc=True,
                       vocab_size=64, char2idx=c2i).to(dev)
    fe = FlatParameters(enc)
    dec = fd = None
    if with_decoder:
      dec = CharDecodingStep(enc, 12, 64, c2i, attention_type='1_layer_nn').to(dev)
      fd = FlatParameters(dec)
      opt = (FusedAdam(fe, lr=1e-3), FusedAdam(fd, lr=1e-3))
    else:
      opt = FusedAdam(fe, lr=1e-3)
    graphs = T.StepGraphs() if use_graphs else None
    torch.manual_seed(10)
    losses = [T.train(enc, dec, batches, opt, dev, c2i, teacher_forcing_ratio=1, grad_norm=5.0, graphs=graphs)
              for _ in range(3)]
    torch.cuda.synchronize()
    if use_graphs:
      assert graphs.captures == 3 and graphs.replays > 3, (graphs.captures, graphs.replays)
    results.append((losses, fe.data.clone(), fd.data.clone() if fd is not None else None))
  assert results[0][0] == results[1][0]                    # epoch averages: identical floats
  assert torch.equal(results[0][1], results[1][1])
  if with_decoder:
    assert torch.equal(results[0][2], results[1][2])


def test_ctc_step_without_final_states_and_with_staged_inputs_is_the_same_step(dev):
  """ctc_step asks the encoder for no final states (need_final_state=False: h_n == NULL in the C ABI) and a
  caller may stage its batches straight into the graph's input buffers (StepGraphs.staging_buffers): the
  encoder returns bit-identical log-probs and hidden states without the final states, and the weights after
  four steps on staged inputs equal those of four steps whose inputs were copied into the graph's buffers."""
  from lipreading_amd import train as T
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.optim import FlatParameters, FusedAdam
  c2i = default_char2idx()
  g = torch.Generator().manual_seed(14)
  B, Tm, Cm = 6, 18, 8
  lens = torch.sort(torch.randint(Tm // 2, Tm + 1, (B,), generator=g))[0]
  lens[-1] = Tm
  frames = torch.randn(B, Tm, 68, 3, generator=g).to(dev)
  cl = torch.full((B,), Cm, dtype=torch.long)
  chars = torch.randint(4, 64, (B, Cm), generator=g)
  chars[:, 0], chars[:, -1] = 1, 2
  lens_d, chars_d, cl_d = lens.to(dev), chars.to(dev), cl.to(dev)

  def fresh():
    torch.manual_seed(3)
    enc = VideoEncoder(204, 16, rnn_type='LSTM', num_layers=2, bidirectional=True, enable_ctc=True,
                       vocab_size=64, char2idx=c2i).to(dev)
    flat = FlatParameters(enc)
    return enc, flat, FusedAdam(flat, lr=1e-3)

  enc, flat, opt = fresh()
  with torch.no_grad():
    lp1, hid1, fin1 = enc(frames, lens_d, max_len=Tm)
    lp0, hid0, fin0 = enc(frames, lens_d, max_len=Tm, need_final_state=False)
  assert fin0 is None and fin1 is not None and torch.equal(lp0, lp1) and torch.equal(hid0, hid1)
  graphs = T.StepGraphs()
  for _ in range(4):
    T.ctc_step(enc, opt, frames, lens_d, chars_d, cl_d, grad_norm=5.0, max_len=Tm, graphs=graphs)
  ref = flat.data.clone()

  enc, flat, opt = fresh()
  graphs = T.StepGraphs()
  assert graphs.staging_buffers() is None
  ins = (frames, lens_d, chars_d, cl_d)
  for i in range(4):
    T.ctc_step(enc, opt, *ins, grad_norm=5.0, max_len=Tm, graphs=graphs)
    if i == 0:   # from now on the batch lives in the graph's own input buffers
      st = graphs.staging_buffers()
      for dst, src in zip(st, ins):
        dst.copy_(src)
      ins = st
  torch.cuda.synchronize()
  assert graphs.captures == 1 and graphs.replays >= 1
  assert torch.equal(flat.data, ref)


@pytest.mark.parametrize("encoder", ["rnn", "transformer"])
def test_pixel_pipeline_end_to_end_on_a_six_video_dataview(dev, tmp_path, encoder):
  """BASELINE configs[1] ("micro (6 videos), LipNet-style 3Dconv+BiGRU+CTC bf16, greedy decode") and the shape of
  configs[4] (transformer encoder over per-frame conv features), composed in the PRODUCT: a dataview that keeps
  what generate_dataview.py:58-76 has in hand per frame (the frame and its landmarks) -> FrameCaptionDataset
  (pixels=True) -> collate: mouth crop on the device (lr_lip_crop_u8) -> PixelLipReader (STCNN frontend ->
  BiGRU or transformer -> CTC head) -> train() / greedy CER, driven by driver.run (--frontend=conv3d
  --encoder=...).  The stages are build-defined (no reference symbol); what is checked is that the product path
  runs end to end, learns, and that its greedy strings / CER equal the ORACLE twin's on the same weights and the
  same clips (oracle: numpy lip crop -> F.conv3d with bf16 storage points -> reference VideoEncoder /
  nn.TransformerEncoder -> reference greedy decode)."""
  from lipreading_amd import driver
  from lipreading_amd import train as T
  from lipreading_amd.dataset import write_synthetic_dataview
  from lipreading_amd.decoder import GreedyDecoder, ctc_labels
  root = str(tmp_path)
  write_synthetic_dataview(root, "synth/micro6", n_videos=6, captions_per_video=3, seed=11, frames=True, frame_hw=64,
                           min_seconds=0.7, max_seconds=1.3)
  out = driver.run(root=root, data="synth/micro6", frontend="conv3d", encoder=encoder, rnn_type="GRU", hidden_size=32,
                   num_layers=1, bidirectional=True, batch_size=4, learning_rate=2e-3, max_epochs=2, train_split=0.7,
                   crop_size=32, nhead=4, cuda=True)
  h = out["history"]
  assert len(h) == 2 and all(np.isfinite(e["ctc_loss"]) for e in h) and h[1]["ctc_loss"] < h[0]["ctc_loss"]
  assert all(0.0 <= e[k] <= 2.0 for e in h for k in ("train_cer", "val_cer", "test_cer"))
  model, c2i, (train_loader, _, _) = out["encoder"], out["char2idx"], out["loaders"]
  # (best_encoder.pth is written when the validation CER drops below 1 — two epochs need not get there; the
  # checkpoint contract is checked directly)
  ckpt = os.path.join(out["weights_dir"], "probe_encoder.pth")
  model.best_error = 1
  model.save_best_model(0.5, ckpt)
  sd = torch.load(ckpt, map_location="cpu")
  assert "frontend.conv1.weight" in sd and any(k.startswith("encoder.") for k in sd) and model.best_error == 0.5
  # ---- the oracle twin on the same weights and the same raw samples -------------------------------------------
  labels = ctc_labels(c2i)
  convs = [p.detach().cpu().clone() for p in model.frontend.parameters_in_order()]
  if encoder == "rnn":
    twin = O.OracleVideoEncoder(96 * 2 * 2, 32, rnn_type="GRU", bidirectional=True, enable_ctc=True, vocab_size=len(c2i),
                                char2idx=c2i)
  else:
    twin = O.OracleTransformerEncoder(96 * 2 * 2, 32, 4, 1, 128, len(c2i), c2i)
  twin.load_state_dict({k: v.detach().cpu() for k, v in model.encoder.state_dict().items()}, strict=False)
  twin.eval()
  model.eval()
  ds = train_loader.dataset
  dec = GreedyDecoder(labels, blank_index=0)
  same = total = 0
  dist_h = dist_o = chars_n = 0
  inv = {v: k for k, v in c2i.items()}
  with torch.no_grad():
    for clips, lens, chars, cl in train_loader:
      lo = total
      s_hip = [o[0] for o in dec.decode(model(clips, lens.to(dev))[0], lens.to(dev))[0]]
      # the oracle's own crop of the raw frames (numpy), its conv stack and encoder
      crops = torch.zeros(clips.shape, dtype=torch.uint8)
      for b in range(clips.shape[0]):
        (frames, lmk), _ = ds[lo + b]
        crops[b, :len(frames)] = torch.from_numpy(O.lip_crop(frames, lmk, size=32))
      assert int((crops.int() - clips.cpu().int()).abs().max()) <= 1          # A9: <= 1 LSB vs the numpy oracle
      feats = O.conv_frontend(clips.cpu(), convs, emulate_bf16=True)
      lp = twin(feats.reshape(clips.shape[0], clips.shape[1], -1, 1), lens)[0]
      s_ref = [o[0] for o in O.greedy_decode(lp, lens, labels)[0]]
      for b in range(len(s_hip)):
        want = ''.join(inv[int(c)] for c in chars[b, 1:int(cl[b]) - 1]).replace(' ', '')
        dist_h += O.edit_distance(s_hip[b].replace('<EOS>', '').replace(' ', ''), want)
        dist_o += O.edit_distance(s_ref[b].replace('<EOS>', '').replace(' ', ''), want)
        chars_n += len(want)
        same += int(s_hip[b] == s_ref[b])
        total += 1
  cer_h, cer_o = dist_h / max(chars_n, 1), dist_o / max(chars_n, 1)
  print("pixel pipeline (%s): greedy strings identical on %d / %d samples, CER %.4f (HIP) vs %.4f (oracle twin)"
        % (encoder, same, total, cer_h, cer_o))
  # bf16 conv stack on both sides, rounded at the same points: a string may differ where a frame's argmax sits
  # within a rounding of a tie — stated tolerance: CER within 0.02, at least 3 of 4 strings identical
  assert abs(cer_h - cer_o) <= 0.02 and same >= 0.75 * total
  assert abs(T.greedy_cer(model, train_loader, dev, c2i) - cer_h) < 1e-12


def test_repeated_recurrence_time_outs_fall_back_to_the_step_kernels(dev, capsys):
  """A device on which the one-launch recurrence cannot run (here: the test hook makes member 1 of every cluster leave
  at once, so its partners time out in EVERY step) must not turn into a run that skips every batch: the reference's
  contract is skip a bad batch and keep training (src/train/train_better_model.py:49-50).  train() watches the
  device-side fault flag without a host round trip (RecurrenceWatch) and after three consecutive faulted steps
  switches every recurrence to the per-step kernels, warns once, drops the captured graphs — the rest of the epoch
  trains.  eval() and greedy_cer() do not score a batch whose recurrence timed out."""
  import warnings
  from lipreading_amd import _C, train as T_
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.optim import FlatParameters, FusedAdam
  L = _C.lib()
  torch.manual_seed(5)
  c2i = default_char2idx()
  enc = VideoEncoder(204, 128, rnn_type='LSTM', bidirectional=True, enable_ctc=True, vocab_size=64, char2idx=c2i).to(dev)
  opt = FusedAdam(FlatParameters(enc), lr=1e-3)
  g = torch.Generator().manual_seed(6)
  B, Tn = 8, 12
  batches = []
  for _ in range(8):
    frames = torch.randn(B, Tn, 68, 3, generator=g)
    chars = torch.zeros(B, 7, dtype=torch.long)
    chars[:, 0], chars[:, 1:6], chars[:, 6] = c2i['<BOS>'], torch.randint(4, 64, (B, 5), generator=g), c2i['<EOS>']
    batches.append((frames, torch.full((B,), Tn), chars, torch.full((B,), 7)))
  assert L.lr_rnn_pair_supported(1, B, Tn, 204, 128, 2) == 2 and L.lr_rnn_one_launch_enabled() == 1
  L.lr_rnn_pair_errors()
  graphs = T_.StepGraphs(warmup=1)
  try:
    L.lr_rnn_debug_drop_member(1)
    # evaluation first: every batch's recurrence times out -> nothing is scored, and it is said
    ctc_avg = T_.eval(enc, None, batches[:2], dev, c2i)[3]
    assert ctc_avg == 0.0 and "dropped after a recurrence time-out: 2 of 2" in capsys.readouterr().out
    before = opt.flat.data.clone()
    with warnings.catch_warnings(record=True) as w:
      warnings.simplefilter("always")
      T_.train(enc, None, batches, opt, dev, c2i, grad_norm=50, graphs=graphs)
    notes = [str(m.message) for m in w if "per-step kernels from here on" in str(m.message)]
    assert len(notes) == 1, [str(m.message) for m in w]
    assert L.lr_rnn_one_launch_enabled() == 0
    st = T_.last_epoch_stats
    # the first steps were lost (three faulted ones are needed to decide, a fourth or fifth may be under way when
    # the flag lands), the rest trained
    assert 3 <= st["skipped"] <= 6 and st["batches"] == 8, st
    assert not torch.equal(before, opt.flat.data) and torch.isfinite(opt.flat.data).all()
    # with the hook still dropping members nothing uses them any more: a whole healthy epoch, CER is computed
    T_.train(enc, None, batches, opt, dev, c2i, grad_norm=50, graphs=graphs)
    assert T_.last_epoch_stats["skipped"] == 0
    assert 0.0 <= T_.greedy_cer(enc, batches[:2], dev, c2i)
  finally:
    L.lr_rnn_debug_drop_member(-1)
    L.lr_rnn_one_launch_enable(1)
    L.lr_rnn_pair_errors()
  # greedy_cer on the one-launch kernels again, one batch's recurrence timing out: that batch is decoded a second
  # time on the step kernels, and the CER equals the undisturbed one
  enc.eval()
  want = T_.greedy_cer(enc, batches[:3], dev, c2i)
  try:
    L.lr_rnn_debug_drop_member(1)
    got = T_.greedy_cer(enc, batches[:3], dev, c2i)
  finally:
    L.lr_rnn_debug_drop_member(-1)
    L.lr_rnn_pair_errors()
  assert got == want


@pytest.mark.parametrize("V", [64, 61, 30, 7])
def test_decoder_nll_for_any_vocabulary_size(dev, V):
  """train_better_model.py:62-65 / :127,138: the decoder loss and its gradient for vocabularies that are not a multiple
  of four as well (round 3 fell back to F.nll_loss there), and eval's (sum, count) pair."""
  import torch.nn.functional as F
  from lipreading_amd import train as T
  g = torch.Generator().manual_seed(V)
  B, L = 5, 9
  lp = torch.log_softmax(torch.randn(B, L, V, generator=g), -1).to(dev).requires_grad_()
  chars = torch.randint(1, V, (B, L + 2), generator=g)
  chars[1, 5:] = 0
  chars[3, 2:] = 0
  chars = chars.to(dev)
  labels = chars[:, 1:]
  loss = T.decoder_nll(lp, labels, 0)
  loss.backward()
  lp_r = lp.detach().clone().requires_grad_()
  want = F.nll_loss(lp_r.reshape(-1, V), labels[:, :L].reshape(-1), ignore_index=0, reduction='sum') / (labels[:, :L] != 0).sum()
  want.backward()
  assert abs(float(loss) - float(want)) < 1e-5 and float((lp.grad - lp_r.grad).abs().max()) < 1e-7
  s, n = T.decoder_nll_sum(lp.detach(), labels, 0)
  assert abs(float(s) - float(want) * float(n)) < 1e-4 and int(n) == int((labels[:, :L] != 0).sum())


@pytest.mark.parametrize("encoder_kind,graph", [("rnn", False), ("rnn", True), ("transformer", False)])
def test_early_sum_of_squares_gives_the_same_clipped_step(dev, encoder_kind, graph):
  """FusedAdam.sum_squares_early (round 5): the sequence encoder's share of clip_grad_norm_'s sum of squares
  (train_better_model.py:78) is taken on the side stream as soon as its last weight gradients exist — beside the conv
  backward — and the launch in front of Adam only sweeps the frontend's gradients.  Same total norm (two partial sums in
  another order: 1e-6 relative), same weights after clipped steps, eager and from a replayed hipGraph."""
  from lipreading_amd import train as T_
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader, feature_dim
  from lipreading_amd.optim import FlatParameters, FusedAdam
  from lipreading_amd.transformer import TransformerVideoEncoder
  c2i = default_char2idx()
  g = torch.Generator().manual_seed(8)
  B, Tn = 4, 12
  clips = torch.randint(0, 256, (B, Tn, 3, 32, 32), generator=g, dtype=torch.uint8).to(dev)
  lens = torch.full((B,), Tn)
  chars = torch.zeros(B, 6, dtype=torch.long)
  chars[:, 0], chars[:, 1:5], chars[:, 5] = c2i['<BOS>'], torch.randint(4, 64, (B, 4), generator=g), c2i['<EOS>']
  char_lens = torch.full((B,), 6)
  out = {}
  for early in (False, True):
    torch.manual_seed(9)
    if encoder_kind == "rnn":
      enc = VideoEncoder(feature_dim(32, 32), 32, rnn_type='GRU', num_layers=2, bidirectional=True, enable_ctc=True,
                         vocab_size=64, char2idx=c2i)
    else:
      enc = TransformerVideoEncoder(feature_dim(32, 32), d_model=64, nhead=4, num_layers=2, dim_feedforward=128,
                                    enable_ctc=True, vocab_size=64, char2idx=c2i)
    model = PixelLipReader(enc, ConvFrontend3D()).to(dev).train()
    opt = FusedAdam(FlatParameters(model), lr=1e-3)
    if early is True:
      opt.sum_squares_early(enc)
    graphs = T_.StepGraphs() if graph else None
    norms = []
    for _ in range(5 if graph else 3):
      loss, status = T_.ctc_step(model, opt, clips, lens.to(dev), chars.to(dev), char_lens.to(dev), grad_norm=0.05,
                                 max_len=Tn, graphs=graphs)
      norms.append(float(opt.total_norm()))
      assert int(status) == 0
      if len(norms) == 1:
        first = opt.flat.data.detach().cpu().numpy().copy()
    if graph:
      assert graphs.replays >= 1
    out[early] = (opt.flat.data.detach().cpu().numpy().copy(), norms, first)
  assert all(n > 0.05 for n in out[False][1])                       # the clip was active in every step
  # the first step's norm is the same sum in another order; later steps start from weights that differ by that rounding
  np.testing.assert_allclose(out[True][1][0], out[False][1][0], rtol=2e-6)
  np.testing.assert_allclose(out[True][1], out[False][1], rtol=5e-3)   # (five clipped Adam steps amplify that rounding)
  # (not even the same run twice gives the same bits: the sum of squares adds its workgroups' shares with float atomics)
  # Weights after the FIRST clipped step: the same gradients scaled by coefficients that differ by that rounding
  np.testing.assert_allclose(out[True][2], out[False][2], rtol=1e-4, atol=1e-6)
  # ... and after the last one: Adam divides by sqrt(v), so an element whose gradient is ~1e-9 (a conv tap that only
  # sees border pixels) moves by ~lr per step in a direction the LAST BIT of the previous step decides, and five steps
  # amplify it; a few percent of the elements may differ, none by more than lr * steps
  steps, lr = len(out[False][1]), 1e-3
  a, b = out[True][0], out[False][0]
  off = ~np.isclose(a, b, rtol=1e-3, atol=1e-5)
  assert off.mean() < 5e-2 and float(np.abs(a - b).max()) <= lr * steps
