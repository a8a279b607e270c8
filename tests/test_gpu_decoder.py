"""GPU parity: the attention decoder loop (N1) through the C ABI against the vectors captured from
the reference's CharDecodingStep (five attention types) and against the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_oracle as O
from tests.test_oracle_golden import DEC2_CASES, DEC_CASES, _flatten, build_oracle_pair, dec_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
  assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
  return torch.device("cuda:0")


def hip_pair(case, rnn_type, attn, dev):
  from lipreading_amd.attention_decoder import CharDecodingStep
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  H, bi, char_dim, ah, layers = dec_cfg(case)
  enc = VideoEncoder(204, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bool(bi), enable_ctc=True,
                     vocab_size=64, char2idx=default_char2idx())
  dec = CharDecodingStep(enc, char_dim=char_dim, vocab_size=64, char2idx=default_char2idx(),
                         attention_type=attn, attn_hidden_size=ah)
  r1 = enc.load_state_dict({k: torch.tensor(v) for k, v in _flatten(case["enc_sd"]).items()})
  r2 = dec.load_state_dict({k: torch.tensor(v) for k, v in _flatten(case["dec_sd"]).items()})
  assert not (r1.missing_keys or r1.unexpected_keys or r2.missing_keys or r2.unexpected_keys)
  return enc.to(dev).train(), dec.to(dev).train()


def grads_close(mod, want, tol=3e-4):
  for k, p in mod.named_parameters():
    ref = want[k]
    got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
    scale = max(1e-4, float(np.abs(ref).max()))
    assert np.abs(got - ref).max() / scale < tol, (k, np.abs(got - ref).max(), scale)


@pytest.mark.parametrize("name", sorted(DEC_CASES) + sorted(DEC2_CASES))
def test_decoder_matches_reference_vectors(golden_dec, golden_dec2, dev, name):
  """encoder -> decoder loop at teacher_forcing_ratio 1 + CTC, decoder_loss.backward(retain_graph)
  then ctc_loss.backward() (train_better_model.py:46-74).  dec2 cases: decoders with 2 / 3 layers
  (better_model.py:136,147-148) and rnn_type='RNN'."""
  from lipreading_amd.ctc import ctc_loss
  case = (golden_dec2 if name in DEC2_CASES else golden_dec)[name]
  enc, dec = hip_pair(case, *{**DEC_CASES, **DEC2_CASES}[name], dev)
  lens = torch.tensor(case["lens"])
  chars = torch.tensor(case["chars"], device=dev)
  char_lens = torch.tensor(case["char_lens"])
  labels, label_lens = chars[:, 1:], char_lens - 1
  L = int(label_lens.max())
  lp_enc, hid, final = enc(torch.tensor(case["frames"], device=dev), lens)
  ctc = ctc_loss(lp_enc, labels, lens.to(dev), label_lens.to(dev), 'mean', dev)
  lp, sampled, _ = dec.decode_sequence(chars[:, :L], final, lens, hid)
  dec_loss = F.nll_loss(lp.reshape(-1, 64), labels[:, :L].reshape(-1), ignore_index=0, reduction='sum')
  dec_loss = dec_loss / (labels != 0).sum()
  np.testing.assert_allclose(lp.detach().cpu().numpy(), case["dec_log_probs"], rtol=1e-4, atol=2e-5)
  assert abs(dec_loss.item() - float(case["dec_loss"])) < 1e-4
  assert abs(ctc.item() - float(case["ctc_loss"])) < 1e-4
  dec_loss.backward(retain_graph=True)
  ctc.backward()
  grads_close(dec, _flatten(case["dec_grad"]))
  grads_close(enc, _flatten(case["enc_grad"]))
  # samples come from exp(log_probs): never the masked PAD/BOS classes
  s = sampled.cpu().numpy()
  assert s.shape == (4, L) and ((s >= 2) & (s < 64)).all()


@pytest.mark.parametrize("name", ["gru_1layernn", "lstm_concat", "gru_l2_1layernn", "lstm_l3_concat"])
def test_single_step_api_equals_fused_loop(golden_dec, golden_dec2, dev, name):
  """CharDecodingStep.forward (the reference's per-step contract) chained L times gives the fused
  decode_sequence result, values and gradients."""
  case = (golden_dec2 if name in DEC2_CASES else golden_dec)[name]
  enc, dec = hip_pair(case, *{**DEC_CASES, **DEC2_CASES}[name], dev)
  lens = torch.tensor(case["lens"])
  chars = torch.tensor(case["chars"], device=dev)
  L = int(case["char_lens"].max()) - 1
  outs = {}
  for mode in ("fused", "steps"):
    enc.zero_grad(); dec.zero_grad()
    _, hid, state = enc(torch.tensor(case["frames"], device=dev), lens)
    if mode == "fused":
      lp, _, _ = dec.decode_sequence(chars[:, :L], state, lens, hid)
    else:
      rows = []
      for i in range(L):
        o, state = dec(chars[:, i], state, lens.to(dev), hid)
        rows.append(o)
      lp = torch.stack(rows, 1)
    (lp * torch.linspace(-1, 1, lp.numel(), device=dev).reshape(lp.shape)).sum().backward()
    outs[mode] = (lp.detach().cpu().numpy(),
                  {k: p.grad.detach().cpu().numpy().copy() for k, p in list(dec.named_parameters()) + list(enc.named_parameters()) if p.grad is not None})
  np.testing.assert_allclose(outs["steps"][0], outs["fused"][0], rtol=1e-5, atol=1e-6)
  for k, g in outs["fused"][1].items():
    scale = max(1e-5, float(np.abs(g).max()))
    assert np.abs(outs["steps"][1][k] - g).max() / scale < 2e-4, k


def test_sampled_inputs_and_bench_sizes_match_oracle(dev):
  """BASELINE-size decoder (B=32, T=75, Hd=512, L=31) with a mixed teacher-forcing pattern: the
  HIP loop's own samples are replayed through the oracle step by step."""
  from lipreading_amd.attention_decoder import CharDecodingStep
  from lipreading_amd.data import default_char2idx

  class Enc:   # only what CharDecodingStep reads from the encoder (better_model.py:134-136)
    hidden_size, bidirectional, rnn_type, num_layers = 256, True, 'GRU', 1

  torch.manual_seed(5)
  ref = O.OracleCharDecodingStep(512, 'GRU', 1, 300, 64, O.default_char2idx(), attention_type='1_layer_nn')
  dec = CharDecodingStep(Enc(), 300, 64, default_char2idx(), attention_type='1_layer_nn')
  dec.load_state_dict(ref.state_dict())
  dec = dec.to(dev)
  g = torch.Generator().manual_seed(6)
  B, T, L = 32, 75, 31
  enc = torch.randn(B, T, 512, generator=g) * 0.5
  lens = torch.sort(torch.randint(40, T + 1, (B,), generator=g))[0]
  h0 = torch.randn(1, B, 512, generator=g) * 0.5
  chars = torch.randint(4, 64, (B, L), generator=g)
  tf = [True] + [bool(i % 3) for i in range(1, L)]
  encd = enc.to(dev).requires_grad_(True)
  lp, sampled, _ = dec.decode_sequence(chars.to(dev), h0.to(dev), lens, encd, teacher_forced=tf, seed=11)
  s = sampled.cpu().long()
  encr = enc.clone().requires_grad_(True)
  state, rows = h0, []
  for i in range(L):
    inp = chars[:, i] if tf[i] else s[:, i - 1]
    o, state = ref(inp, state, lens, encr)
    rows.append(o)
  want = torch.stack(rows, 1)
  np.testing.assert_allclose(lp.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-4, atol=5e-5)
  wgt = torch.randn(B, L, 64, generator=g) / 100
  (lp * wgt.to(dev)).sum().backward()
  (want * wgt).sum().backward()
  a, b = encd.grad.cpu().numpy(), encr.grad.numpy()
  assert np.abs(a - b).max() / np.abs(b).max() < 5e-4
  gr = dict(ref.named_parameters())
  for k, p in dec.named_parameters():
    r = gr[k].grad.numpy()
    # (the score bias of '1_layer_nn' has a mathematically zero gradient - softmax is shift
    #  invariant - so both sides hold rounding noise there; the absolute floor covers it)
    assert np.abs(p.grad.cpu().numpy() - r).max() / max(1e-4, np.abs(r).max()) < 5e-4, k


@pytest.mark.parametrize("Henc,attn,char_dim,B", [(512, "1_layer_nn", 256, 4), (512, "1_layer_nn", 256, 32),
                                                 (768, "none", 256, 8), (768, "none", 256, 32), (768, "1_layer_nn", 256, 32),
                                                 (700, "1_layer_nn", 300, 32), (768, "none", 256, 72)])
def test_shipped_decoder_configs_match_oracle(dev, Henc, attn, char_dim, B):
  """The decoders of the reference's SHIPPED flag files against the ORACLE (round 4 checked them inside bench.py only):
  config/train/attn/attention_type:8-19 — BiLSTM-512 encoder, so an LSTM-1024 decoder (better_model.py:134), char_dim 256,
  1_layer_nn attention, at the config's batch of 4 and at 32 — and the ecd family (config/archive/experiments/ecd/*:
  BiLSTM-768 -> LSTM-1536, attention none).  Every step teacher forced (teacher_forcing_ratio 1.0 in those files): the
  loop's RNN runs as ONE launch: 1024 on 16-unit cluster members, 1400 / 1536 (round 6) on the 192-CU grid recurrence
  (lr_rnn_grid.hip; 72 samples: two launches) — and nothing announces a fallback to the step kernels."""
  import warnings
  from lipreading_amd import encoder as E
  E._fallback_noted.clear()
  from lipreading_amd import _C
  from lipreading_amd.attention_decoder import CharDecodingStep
  from lipreading_amd.data import default_char2idx

  class Enc:   # only what CharDecodingStep reads from the encoder (better_model.py:134-136)
    hidden_size, bidirectional, rnn_type, num_layers = Henc, True, 'LSTM', 1

  Hd = 2 * Henc
  torch.manual_seed(15)
  ref = O.OracleCharDecodingStep(Hd, 'LSTM', 1, char_dim, 64, O.default_char2idx(), attention_type=attn)
  dec = CharDecodingStep(Enc(), char_dim, 64, default_char2idx(), attention_type=attn)
  dec.load_state_dict(ref.state_dict())
  dec = dec.to(dev)
  g = torch.Generator().manual_seed(16)
  T, L = 75, 31
  enc = torch.randn(B, T, Hd, generator=g) * 0.5
  lens = torch.sort(torch.randint(40, T + 1, (B,), generator=g))[0]
  h0, c0 = torch.randn(1, B, Hd, generator=g) * 0.5, torch.randn(1, B, Hd, generator=g) * 0.5
  chars = torch.randint(4, 64, (B, L), generator=g)
  assert _C.lib().lr_rnn_one_launch_status(1, B, L, Hd, Hd, 1) == 0      # the one-launch path is what runs
  _C.lib().lr_rnn_pair_errors()
  encd = enc.to(dev).requires_grad_(True)
  h0d, c0d = h0.to(dev).requires_grad_(True), c0.to(dev).requires_grad_(True)
  with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter("always")
    lp, _, _ = dec.decode_sequence(chars.to(dev), (h0d, c0d), lens, encd, seed=11)
  assert not [w for w in caught if "one launch per time step" in str(w.message)], [str(w.message) for w in caught]
  encr, h0r, c0r = enc.clone().requires_grad_(True), h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
  state, rows = (h0r, c0r), []
  for i in range(L):
    o, state = ref(chars[:, i], state, lens, encr)
    rows.append(o)
  want = torch.stack(rows, 1)
  np.testing.assert_allclose(lp.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-4, atol=5e-5)
  wgt = torch.randn(B, L, 64, generator=g) / 100
  (lp * wgt.to(dev)).sum().backward()
  (want * wgt).sum().backward()
  assert _C.lib().lr_rnn_pair_errors() == 0
  pairs = [(h0d.grad, h0r.grad), (c0d.grad, c0r.grad)]
  if attn != "none":
    pairs.append((encd.grad, encr.grad))
  for a, b in pairs:
    assert float((a.cpu() - b).abs().max()) <= 5e-4 * float(b.abs().max())
  gr = dict(ref.named_parameters())
  for k, p in dec.named_parameters():
    if gr[k].grad is None:      # (attention 'none' never touches concat_layer: better_model.py:226-227)
      assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
      continue
    r = gr[k].grad.numpy()
    assert np.abs(p.grad.cpu().numpy() - r).max() / max(1e-4, np.abs(r).max()) < 5e-4, k


@pytest.mark.parametrize("rnn_type,layers,attn", [("LSTM", 2, "general"), ("GRU", 3, "dot"), ("RNN", 2, "1_layer_nn")])
def test_multilayer_decoder_with_sampled_inputs_matches_oracle(dev, rnn_type, layers, attn):
  """Decoder stacks deeper than one layer (better_model.py:136,147-148) under a MIXED teacher-forcing
  pattern: a sampled-input step needs the top layer's output of the step before, so the stack runs
  layer-major inside each run of known inputs.  The HIP loop's own samples are replayed through the
  oracle (nn.GRU/LSTM/RNN(num_layers)) step by step; values and every gradient are compared."""
  from lipreading_amd.attention_decoder import CharDecodingStep
  from lipreading_amd.data import default_char2idx
  Hd, Cd = 48, 20

  class Enc:   # only what CharDecodingStep reads from the encoder (better_model.py:134-136)
    hidden_size, bidirectional, num_layers = Hd // 2, True, layers
  Enc.rnn_type = rnn_type
  torch.manual_seed(15)
  ref = O.OracleCharDecodingStep(Hd, rnn_type, layers, Cd, 64, O.default_char2idx(), attention_type=attn)
  dec = CharDecodingStep(Enc(), Cd, 64, default_char2idx(), attention_type=attn)
  dec.load_state_dict(ref.state_dict())
  dec = dec.to(dev)
  g = torch.Generator().manual_seed(16)
  B, T, L = 7, 23, 9
  enc = torch.randn(B, T, Hd, generator=g) * 0.5
  lens = torch.sort(torch.randint(10, T + 1, (B,), generator=g))[0]
  h0 = torch.randn(layers, B, Hd, generator=g) * 0.5
  c0 = torch.randn(layers, B, Hd, generator=g) * 0.5
  chars = torch.randint(4, 64, (B, L), generator=g)
  tf = [True, True, False, True, False, False, True, True, False]
  encd = enc.to(dev).requires_grad_(True)
  h0d, c0d = h0.to(dev).requires_grad_(True), c0.to(dev).requires_grad_(True)
  state_d = (h0d, c0d) if rnn_type == "LSTM" else h0d
  lp, sampled, fin = dec.decode_sequence(chars.to(dev), state_d, lens, encd, teacher_forced=tf, seed=17)
  s = sampled.cpu().long()
  encr = enc.clone().requires_grad_(True)
  h0r, c0r = h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
  state, rows = ((h0r, c0r) if rnn_type == "LSTM" else h0r), []
  for i in range(L):
    inp = chars[:, i] if tf[i] else s[:, i - 1]
    o, state = ref(inp, state, lens, encr)
    rows.append(o)
  want = torch.stack(rows, 1)
  np.testing.assert_allclose(lp.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-4, atol=2e-5)
  fins_d = fin if isinstance(fin, tuple) else (fin,)
  fins_r = state if isinstance(state, tuple) else (state,)
  for a, b in zip(fins_d, fins_r):
    assert a.shape == (layers, B, Hd)
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-4, atol=2e-5)
  wgt = torch.randn(B, L, 64, generator=g) / 100
  wf = torch.randn(layers, B, Hd, generator=g) / 50
  # the returned final state takes part in the loss: its gradient enters every layer's last step
  ((lp * wgt.to(dev)).sum() + sum((f * wf.to(dev)).sum() for f in fins_d)).backward()
  ((want * wgt).sum() + sum((f * wf).sum() for f in fins_r)).backward()
  pairs = [("enc", encd.grad, encr.grad), ("h0", h0d.grad, h0r.grad)]
  if rnn_type == "LSTM":
    pairs.append(("c0", c0d.grad, c0r.grad))
  gr = dict(ref.named_parameters())
  pairs += [(k, p.grad, gr[k].grad) for k, p in dec.named_parameters()]
  for k, a, b in pairs:
    a, b = a.cpu().numpy(), b.numpy()
    assert np.abs(a - b).max() / max(1e-4, np.abs(b).max()) < 5e-4, k


def test_multilayer_decoder_dropout_is_a_mask_between_layers(dev):
  """nn.GRU(dropout=p) semantics in training mode: the outputs of every layer but the last are dropped
  out before they feed the next layer.  The draw is RNG-dependent (parity in distribution only), so the
  check is structural: p -> 0 reproduces the no-dropout result, eval mode ignores it, and with p > 0 the
  training-mode result differs while gradients stay finite."""
  from lipreading_amd.attention_decoder import CharDecodingStep
  from lipreading_amd.data import default_char2idx

  class Enc:
    hidden_size, bidirectional, rnn_type, num_layers = 16, True, 'GRU', 2
  torch.manual_seed(25)
  dec = CharDecodingStep(Enc(), 12, 64, default_char2idx(), rnn_dropout=0.0, attention_type='dot').to(dev).train()
  g = torch.Generator().manual_seed(26)
  B, T, L = 5, 11, 6
  enc = (torch.randn(B, T, 32, generator=g) * 0.5).to(dev)
  lens = torch.tensor([7, 9, 11, 11, 11])
  h0 = (torch.randn(2, B, 32, generator=g) * 0.5).to(dev)
  chars = torch.randint(4, 64, (B, L), generator=g).to(dev)
  base, _, _ = dec.decode_sequence(chars, h0, lens, enc, seed=1)
  dec.rnn_dropout = 0.5
  dec.eval()
  ev, _, _ = dec.decode_sequence(chars, h0, lens, enc, seed=1)
  assert torch.equal(ev, base)
  dec.train()
  torch.manual_seed(27)
  dr, _, _ = dec.decode_sequence(chars, h0, lens, enc, seed=1)
  assert float((dr - base).abs().max()) > 1e-3
  dec.zero_grad()
  dr.sum().backward()
  assert all(torch.isfinite(p.grad).all() for p in dec.parameters() if p.grad is not None)
  # only layer 0's outputs are dropped: the mask reaches layer 0's weights' gradient but a zero mask
  # row cannot make the top layer's bias gradient vanish
  assert float(dec.rnn.bias_hh_l1.grad.abs().sum()) > 0


@pytest.mark.parametrize("rnn_type,attn", [("GRU", "1_layer_nn"), ("LSTM", "dot")])
def test_train_and_eval_with_decoder_follow_reference_loop(dev, rnn_type, attn):
  """train()/eval() of train_better_model.py with a CharDecodingStep (teacher_forcing_ratio 1,
  grad_norm 5, Adam 1e-3): three batches through the HIP path against the same loop run on the
  oracle modules with torch.optim.Adam and per-module clip_grad_norm_."""
  from lipreading_amd import train as T
  from lipreading_amd.attention_decoder import CharDecodingStep
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.optim import FlatParameters, FusedAdam
  c2i = default_char2idx()
  torch.manual_seed(3)
  renc = O.OracleVideoEncoder(204, 16, rnn_type=rnn_type, num_layers=1, bidirectional=True, enable_ctc=True,
                              vocab_size=64, char2idx=O.default_char2idx())
  rdec = O.OracleCharDecodingStep(32, rnn_type, 1, 12, 64, O.default_char2idx(), attention_type=attn)
  enc = VideoEncoder(204, 16, rnn_type=rnn_type, num_layers=1, bidirectional=True, enable_ctc=True,
                     vocab_size=64, char2idx=c2i)
  dec = CharDecodingStep(enc, 12, 64, c2i, attention_type=attn)
  enc.load_state_dict(renc.state_dict()); dec.load_state_dict(rdec.state_dict())
  enc, dec = enc.to(dev), dec.to(dev)
  g = torch.Generator().manual_seed(4)
  batches = []
  for _ in range(3):
    lens = torch.sort(torch.randint(12, 21, (6,), generator=g))[0]
    frames = torch.randn(6, int(lens.max()), 68, 3, generator=g)
    for b in range(6):
      frames[b, lens[b]:] = 0
    cl = torch.randint(4, 9, (6,), generator=g)
    chars = torch.zeros(6, int(cl.max()), dtype=torch.long)
    for b in range(6):
      n = int(cl[b])
      chars[b, 0], chars[b, n - 1] = 1, 2
      chars[b, 1:n - 1] = torch.randint(4, 64, (n - 2,), generator=g)
    batches.append((frames, lens, chars, cl))
  # reference loop on the oracle (train_better_model.py:46-80)
  ropt = torch.optim.Adam(list(renc.parameters()) + list(rdec.parameters()), lr=1e-3)
  rdl, rcl = 0.0, 0.0
  for frames, lens, chars, cl in batches:
    lp, hid, st = renc(frames, lens)
    ctc = O.ctc_loss(lp, chars[:, 1:], lens, cl - 1, 'mean')
    dl, _ = O.decoder_loop(rdec, chars, cl, hid, lens, st)
    ropt.zero_grad()
    dl.backward(retain_graph=True)
    ctc.backward()
    torch.nn.utils.clip_grad_norm_(renc.parameters(), 5.0)
    torch.nn.utils.clip_grad_norm_(rdec.parameters(), 5.0)
    ropt.step()
    rdl += dl.item() / 3; rcl += ctc.item() / 3
  opt = (FusedAdam(FlatParameters(enc), lr=1e-3), FusedAdam(FlatParameters(dec), lr=1e-3))
  dl, cl_ = T.train(enc, dec, batches, opt, dev, c2i, teacher_forcing_ratio=1, grad_norm=5.0)
  assert abs(dl - rdl) < 2e-4 and abs(cl_ - rcl) < 2e-4
  for mod, ref in ((enc, renc), (dec, rdec)):
    want = ref.state_dict()
    for k, v in mod.state_dict().items():
      if k == "attn_proj_1_layer_nn.bias":
        continue   # zero true gradient (softmax shift invariance): Adam normalises rounding noise
      # Adam divides by sqrt(v): elements whose gradient is ~0 move by lr-sized, noise-signed steps
      np.testing.assert_allclose(v.cpu().numpy(), want[k].numpy(), rtol=2e-4, atol=1.5e-4, err_msg=k)
  # eval (:89-143): teacher-forced NLL / count, CTC 'sum' averaged over batches
  renc.eval(); rdec.eval()
  nll, count, csum = 0.0, 0, 0.0
  with torch.no_grad():
    for frames, lens, chars, cl in batches:
      lp, hid, st = renc(frames, lens)
      csum += O.ctc_loss(lp, chars[:, 1:], lens, cl - 1, 'sum').item()
      l, _ = O.decoder_loop(rdec, chars, cl, hid, lens, st)
      n = int((chars[:, 1:] != 0).sum())
      nll += l.item() * n; count += n
  d, correct, cnt, c = T.eval(enc, dec, batches, dev, c2i)
  assert cnt == count and 0 <= correct <= count
  assert abs(d - nll / count) < 2e-4 and abs(c - csum / 3) < 2e-3


def test_fused_decoder_nll_equals_the_reference_formula(dev):
  """train.decoder_nll (lr_nll_mean_forward/backward: two launches) = sum over steps of F.nll_loss(ignore_index=PAD,
  reduction='sum') / (labels != PAD).sum() (train_better_model.py:62,65), value and gradient; labels are the strided
  view chars[:, 1:] the loop slices."""
  from lipreading_amd.train import decoder_nll
  g = torch.Generator().manual_seed(3)
  B, L, V = 7, 9, 64
  chars = torch.randint(4, V, (B, L + 1), generator=g)
  for b, n in enumerate([9, 3, 5, 9, 1, 7, 2]):
    chars[b, 1 + n:] = 0                       # PAD after each sample's labels
  lp = torch.log_softmax(torch.randn(B, L, V, generator=g), -1)
  labels = chars[:, 1:]
  a = lp.clone().requires_grad_(True)
  want = F.nll_loss(a.reshape(-1, V), labels.reshape(-1), ignore_index=0, reduction='sum') / (labels != 0).sum()
  (want * 1.7).backward()
  x = lp.to(dev).requires_grad_(True)
  got = decoder_nll(x, chars.to(dev)[:, 1:], 0)
  (got * 1.7).backward()
  assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want))
  np.testing.assert_allclose(x.grad.cpu().numpy(), a.grad.numpy(), rtol=1e-6, atol=0)


@pytest.mark.parametrize("rnn_type,Hd,layers,attn,B,L", [("GRU", 512, 1, "1_layer_nn", 32, 31), ("LSTM", 512, 2, "general", 32, 31),
                                                         ("GRU", 256, 2, "none", 32, 31), ("LSTM", 704, 1, "dot", 32, 31),
                                                         ("LSTM", 256, 1, "concat", 5, 3), ("GRU", 500, 1, "dot", 13, 1),
                                                         # round 6: the 1400 / 1536-unit decoders behind BiLSTM-700 / 768 (config/defaults.txt
                                                         # sizes made bidirectional; config/archive/experiments/e*: 52 of 56 sized files) on
                                                         # the 192-CU grid recurrence (lr_rnn_grid.hip): one block of 32 samples, a partial
                                                         # block, two blocks, two launches (70 samples), two layers, a single step
                                                         ("LSTM", 1536, 1, "none", 32, 31), ("LSTM", 1400, 1, "1_layer_nn", 32, 31),
                                                         ("LSTM", 1536, 1, "dot", 5, 3), ("LSTM", 1536, 1, "none", 40, 7),
                                                         ("LSTM", 1400, 2, "none", 70, 4), ("LSTM", 1156, 1, "none", 13, 1),
                                                         # sixteen samples per cluster with an initial state and its gradient (the
                                                         # backward's extra step): 22 / 24-member decoders past 64 samples
                                                         ("LSTM", 704, 1, "dot", 72, 5), ("GRU", 768, 1, "none", 100, 3)])
def test_decoder_loop_on_the_cluster_recurrence_equals_the_step_kernels(dev, rnn_type, Hd, layers, attn, B, L):
  """With every step teacher forced (the shipped configs and eval) the decoder's RNN — unidirectional, started from
  the encoder's final state (better_model.py:134-148,181) — runs each layer's L steps as ONE launch of the cluster
  recurrence (lr_rnn_cluster.hip), and its backward returns the gradient into that initial state from the same
  launch.  Against the per-step kernels on the same weights and inputs (test hook lr_rnn_debug_disable_cluster):
  log-probs, final states and every gradient incl. d(h0), d(c0), d(enc)."""
  from lipreading_amd import _C
  from lipreading_amd.attention_decoder import CharDecodingStep
  from lipreading_amd.data import default_char2idx
  L_ = _C.lib()

  class Enc:   # only what CharDecodingStep reads from the encoder (better_model.py:134-136)
    hidden_size, bidirectional, num_layers = Hd // 2, True, layers
  Enc.rnn_type = rnn_type
  torch.manual_seed(21)
  dec = CharDecodingStep(Enc(), 300, 64, default_char2idx(), attention_type=attn,
                         attn_hidden_size=(64 if attn == "concat" else -1)).to(dev)
  g = torch.Generator().manual_seed(22)
  T = 75 if B == 32 else 9     # (ragged sample groups, a single step: B % 8 != 0, L = 1)
  import warnings
  from lipreading_amd import encoder as E
  E._fallback_noted.clear()
  enc = (torch.randn(B, T, Hd, generator=g) * 0.5).to(dev)
  lens = torch.sort(torch.randint(max(1, T // 2), T + 1, (B,), generator=g))[0]
  h0 = (torch.randn(layers, B, Hd, generator=g) * 0.5).to(dev)
  c0 = (torch.randn(layers, B, Hd, generator=g) * 0.5).to(dev)
  chars = torch.randint(4, 64, (B, L), generator=g).to(dev)
  wgt = (torch.randn(B, L, 64, generator=g) / 100).to(dev)
  wf = (torch.randn(layers, B, Hd, generator=g) / 50).to(dev)
  mode = {"GRU": 0, "LSTM": 1}[rnn_type]
  assert L_.lr_rnn_pair_supported(mode, B, L, Hd, Hd, 1) in (1, 2)   # (GRU-256: the encoder would take the pair kernels)
  if Hd > 1152:
    assert L_.lr_rnn_one_launch_status(mode, B, L, Hd, Hd, 1) == 0
    assert L_.lr_rnn_pass_launches(mode, B, L, Hd, Hd, 1) == (B + 63) // 64     # 64 samples per launch of the grid
  L_.lr_rnn_pair_errors()
  out = {}
  try:
    for name, off in (("cluster", 0), ("steps", 1)):
      L_.lr_rnn_debug_disable_cluster(off)
      dec.zero_grad()
      e, h, c = enc.clone().requires_grad_(True), h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
      with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        lp, _, fin = dec.decode_sequence(chars, (h, c) if rnn_type == "LSTM" else h, lens, e, seed=5)
      if name == "cluster":     # the one-launch path announces no fallback (rounds 1-5: "LSTM-1536 runs one launch per time step")
        assert not [w for w in caught if "one launch per time step" in str(w.message)], [str(w.message) for w in caught]
      fins = fin if isinstance(fin, tuple) else (fin,)
      ((lp * wgt).sum() + sum((f * wf).sum() for f in fins)).backward()
      egrad = e.grad if e.grad is not None else torch.zeros_like(e)    # (attention 'none' never reads the encoder states)
      out[name] = [lp.detach().cpu()] + [f.detach().cpu() for f in fins] + [egrad.cpu(), h.grad.cpu()] + \
                  ([c.grad.cpu()] if rnn_type == "LSTM" else []) + \
                  [(p.grad if p.grad is not None else torch.zeros_like(p)).cpu().clone() for p in dec.parameters()]
  finally:
    L_.lr_rnn_debug_disable_cluster(0)
  assert L_.lr_rnn_pair_errors() == 0
  if L > 1:
    assert float((out["cluster"][0] - out["steps"][0]).abs().max()) > 0     # the other path really ran
  worst = 0.0
  for i, (a, b) in enumerate(zip(out["steps"], out["cluster"])):
    # (floor: the score bias of '1_layer_nn' has a mathematically zero gradient — rounding noise on both sides)
    rel = float((a - b).norm()) / max(1e-4, float(a.norm()))
    worst = max(worst, rel)
    assert rel < 5e-5, (i, rel, tuple(a.shape))
  print("decoder %s-%d x%d: cluster vs step kernels, worst relative norm difference %.3g" % (rnn_type, Hd, layers, worst))


@pytest.mark.parametrize("rnn_type,Hd,B,attn", [("LSTM", 1536, 32, "none"), ("GRU", 512, 8, "none"), ("LSTM", 1024, 4, "1_layer_nn"),
                                                ("GRU", 512, 8, "general"), ("LSTM", 512, 8, "concat"), ("GRU", 512, 8, "dot")])
def test_loop_weight_half_on_the_side_stream_is_bit_identical(dev, rnn_type, Hd, B, attn):
  """A single-layer loop (every flag file the reference ships) splits its backward
  (lr_decoder_backward_parts): the data half on the caller's stream, every parameter gradient on the decoder module's own side
  stream, joined when the backward pass ends.  Same kernels, same arguments: the gradients are the unsplit call's, bit for
  bit, and whoever reads them after backward() finds them complete."""
  import lipreading_amd.attention_decoder as AD
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.optim import FlatParameters

  class Enc:
    hidden_size, bidirectional, rnn_type, num_layers = Hd // 2, True, None, 1
  Enc.rnn_type = rnn_type
  torch.manual_seed(21)
  dec = AD.CharDecodingStep(Enc(), 64, 64, default_char2idx(), attention_type=attn,
                            attn_hidden_size=96 if attn == "concat" else -1).to(dev)
  FlatParameters(dec)                      # dense .grad buffers: the loop accumulates into them directly
  g = torch.Generator().manual_seed(22)
  T, L = 40, 19
  enc = (torch.randn(B, T, Hd, generator=g) * 0.5).to(dev)
  lens = torch.full((B,), T)
  h0 = (torch.randn(1, B, Hd, generator=g) * 0.5).to(dev)
  c0 = (torch.randn(1, B, Hd, generator=g) * 0.5).to(dev)
  chars = torch.randint(4, 64, (B, L), generator=g).to(dev)
  wgt = (torch.randn(B, L, 64, generator=g) / 100).to(dev)
  res = {}
  for overlap in (False, True):
    AD.overlap_weight_half = overlap
    try:
      AD.take_split_flag()
      for p in dec.parameters():
        p.grad.zero_()
      h0d = h0.clone().requires_grad_(True)
      encd = enc.clone().requires_grad_(True)
      state = (h0d, c0.clone().requires_grad_(True)) if rnn_type == "LSTM" else h0d
      lp, _, _ = dec.decode_sequence(chars, state, lens, encd, seed=5)
      (lp * wgt).sum().backward()
      assert AD.take_split_flag() == overlap          # the split path is what ran (and only then)
      # (no synchronisation of our own: reading on the current stream must be enough)
      res[overlap] = [h0d.grad.clone(), encd.grad.clone()] + [p.grad.clone() for p in dec.parameters()]
    finally:
      AD.overlap_weight_half = True
  for a, b in zip(res[False], res[True]):
    assert torch.equal(a, b)
