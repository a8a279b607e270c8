"""The oracle (oracle/torch_oracle.py) against the vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only.  This is what pins the oracle; the GPU parity tests
then compare the HIP path with the oracle and with the same vectors."""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as O

CTC_CASES = ["equal", "runs112", "runs233", "repeats", "toolong", "alltoolong", "infsample",
             "infrun", "inffirst", "allinf", "bench", "mixed75"]


@pytest.mark.parametrize("name", CTC_CASES)
@pytest.mark.parametrize("red", ["mean", "sum"])
def test_ctc_oracle_matches_reference(golden_ctc, name, red):
  c = golden_ctc[name]
  lp = torch.tensor(c["lp"], requires_grad=True)
  loss = O.ctc_loss(lp, torch.tensor(c["labels"]), torch.tensor(c["frame_lens"]),
                    torch.tensor(c["label_lens"]), red)
  if int(c["none_" + red]):
    assert loss is None
    return
  assert loss is not None
  np.testing.assert_allclose(loss.item(), float(c["loss_" + red]), rtol=1e-6, atol=1e-6)
  loss.backward()
  np.testing.assert_allclose(lp.grad.numpy(), c["grad_" + red], rtol=1e-5, atol=1e-6)


def test_ctc_quirk_is_not_the_sample_mean(golden_ctc):
  """runs (1,1,2) in B=4: (4*m1 + 1*m2 + 1*m3)/6, SURVEY.md A4."""
  c = golden_ctc["runs112"]
  lp = torch.tensor(c["lp"])
  lab = torch.tensor(c["labels"]) + 1
  fl, ll = torch.tensor(c["frame_lens"]), torch.tensor(c["label_lens"])
  nll = O.ctc_nll_per_sample(lp, lab, fl, ll)
  per = nll / ll.float()
  quirk = (4 * per[0] + 1 * per[1] + 1 * per[2:4].mean()) / 6
  np.testing.assert_allclose(quirk.item(), float(c["loss_mean"]), rtol=1e-6)
  assert abs(per.mean().item() - float(c["loss_mean"])) > 1e-3


ENC_CASES = ["gru_bi", "gru_uni", "lstm_bi", "lstm_uni", "gru_bi_l2", "lstm_bi_l2"]
ENC2_CASES = ["rnn_bi", "rnn_uni_l2", "rnn_bi_l2"]      # enc2_cases.npz: rnn_type='RNN' (tanh)


def rnn_type_of(name):
  return {"gru": "GRU", "lstm": "LSTM", "rnn": "RNN"}[name.split("_")[0]]


def build_oracle_encoder(case, rnn_type):
  H, layers, bi = [int(x) for x in case["cfg"]]
  enc = O.OracleVideoEncoder(204, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bool(bi),
                             enable_ctc=True, vocab_size=64, char2idx=O.default_char2idx())
  key = "sd" if "sd" in case else "sd0"
  enc.load_state_dict({k: torch.tensor(v) for k, v in _flatten(case[key]).items()})
  return enc


def _flatten(d, prefix=""):
  out = {}
  for k, v in d.items():
    if isinstance(v, dict):
      out.update(_flatten(v, prefix + k + "/"))
    else:
      out[(prefix + k).replace("/", ".")] = v
  return out


@pytest.mark.parametrize("name", ENC_CASES + ENC2_CASES)
@pytest.mark.parametrize("tag", ["eq", "mix"])
def test_encoder_oracle_matches_reference(golden_enc, golden_enc2, name, tag):
  case = (golden_enc2 if name in ENC2_CASES else golden_enc)[name]
  enc = build_oracle_encoder(case, rnn_type_of(name)).eval()
  io = case[tag]
  with torch.no_grad():
    lp, hid, fin = enc(torch.tensor(io["frames"]), torch.tensor(io["lens"]))
  np.testing.assert_allclose(lp.numpy(), io["log_probs"], rtol=1e-5, atol=1e-5)
  np.testing.assert_allclose(hid.numpy(), io["hidden"], rtol=1e-5, atol=1e-6)
  if isinstance(fin, tuple):
    np.testing.assert_allclose(fin[0].numpy(), io["h_n"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(fin[1].numpy(), io["c_n"], rtol=1e-5, atol=1e-6)
  else:
    np.testing.assert_allclose(fin.numpy(), io["h_n"], rtol=1e-5, atol=1e-6)
  # masked classes (PAD+1, BOS+1) are finite, ~103.28 below
  assert np.isfinite(io["log_probs"]).all()
  assert (io["log_probs"][..., 1:3] < -90).all()


def test_greedy_oracle_matches_reference(golden_greedy):
  g = golden_greedy
  labels = list(g["labels"])
  assert labels == O.ctc_labels()
  strings, offsets = O.greedy_decode(torch.tensor(g["lp"]), torch.tensor(g["sizes"]), labels)
  assert [s[0] for s in strings] == list(g["strings"])
  for b in range(len(strings)):
    np.testing.assert_array_equal(offsets[b][0].numpy(), g["offsets_%d" % b])
  strings2, _ = O.greedy_decode(torch.tensor(g["lp"]), None, labels)
  assert [s[0] for s in strings2] == list(g["strings_nosizes"])


@pytest.mark.parametrize("name", ["gru", "lstm"])
def test_step_oracle_matches_reference(golden_step, name):
  case = golden_step[name]
  enc = build_oracle_encoder(case, name.upper()).train()
  opt = torch.optim.Adam(enc.parameters(), lr=1e-3)
  loss = O.encoder_ctc_step(enc, opt, torch.tensor(case["frames"]), torch.tensor(case["lens"]),
                            torch.tensor(case["chars"]), torch.tensor(case["char_lens"]),
                            grad_norm=5.0)
  np.testing.assert_allclose(loss.item(), float(case["loss"]), rtol=1e-5)
  sd1 = _flatten(case["sd1"])
  for k, v in enc.state_dict().items():
    np.testing.assert_allclose(v.numpy(), sd1[k], rtol=1e-4, atol=1e-6, err_msg=k)


def test_landmark_oracle_matches_reference(golden_lmk):
  """lmk_cases: the reference's _applyPadding / extractFace / getFace were RUN (clean pin)."""
  g = golden_lmk
  for i in range(len(g["rects"])):
    padded = O.apply_padding(tuple(g["dims"][i]), tuple(int(x) for x in g["rects"][i]),
                             float(g["padding"]))
    assert list(padded) == list(g["padded"][i])
    assert (padded[3] - padded[2], padded[1] - padded[0]) == tuple(g["crop_hw"][i])     # extractFace's crop
    np.testing.assert_array_equal(O.get_face(g["lmk"][i], padded), g["face"][i])


def test_prn_oracle_matches_reference(golden_prn):
  """prn_cases: the reference's PRN.process / get_landmarks bodies run on a seeded position map
  (shimmed pin: estimate_transform).  Crop transform, restored map (16-strided sample), gathered
  landmarks and the dataview row (padded-rect translation) — generate_dataview.py:58-64."""
  from tests.golden.make_golden import prn_position_map
  g = golden_prn
  uv = g["uv_kpt_ind"]
  assert uv.shape == (2, 68)
  for i in range(len(g["rects"])):
    rect = tuple(int(v) for v in g["rects"][i])
    cropped = prn_position_map(int(g["seeds"][i]), rect)
    tform, size = O.prn_crop_transform(rect)
    np.testing.assert_allclose(tform, g["tform"][i], rtol=0, atol=1e-12)
    # the closed form the HIP kernel uses: exact similarity of the three corner points
    np.testing.assert_allclose(tform, [[255.0 / size, 0, tform[0, 2]], [0, 255.0 / size, tform[1, 2]], [0, 0, 1]],
                               rtol=1e-13, atol=1e-13)
    pos = O.prn_restore(cropped, tform)
    np.testing.assert_allclose(pos[::16, ::16], g["pos_sub16"][i], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(O.prn_get_landmarks(pos, uv), g["kpt"][i], rtol=1e-12, atol=1e-9)
    face, padded = O.landmark_step(cropped, rect, tuple(g["dims"][i]), uv, 0.3)
    assert list(padded) == list(g["padded"][i])
    np.testing.assert_allclose(face, g["face_lmk"][i], rtol=1e-12, atol=1e-9)


def test_collate_pads_with_zeros():
  rng = np.random.RandomState(0)
  frames = [rng.randn(n, 68, 3) for n in (3, 5, 4)]
  caps = [np.arange(1, n + 1) for n in (2, 4, 3)]
  f, fl, c, cl = O.collate(frames, caps)
  assert f.shape == (3, 5, 68, 3) and f.dtype == torch.float32
  assert fl.tolist() == [3, 5, 4] and cl.tolist() == [2, 4, 3]
  assert float(f[0, 3:].abs().sum()) == 0 and c[0, 2:].tolist() == [0, 0]
  np.testing.assert_allclose(f[2, :4].numpy(), frames[2].astype(np.float32))


DEC_CASES = {"gru_1layernn": ("GRU", "1_layer_nn"), "lstm_dot": ("LSTM", "dot"), "gru_general": ("GRU", "general"),
             "lstm_concat": ("LSTM", "concat"), "gru_none": ("GRU", "none")}


# dec2_cases.npz: decoders with num_layers 2 / 3 (better_model.py:136,147-148) and the tanh RNN
DEC2_CASES = {"gru_l2_1layernn": ("GRU", "1_layer_nn"), "lstm_l2_dot": ("LSTM", "dot"), "rnn_l1_general": ("RNN", "general"),
              "rnn_l3_none": ("RNN", "none"), "lstm_l3_concat": ("LSTM", "concat")}


def dec_cfg(case):
  cfg = [int(x) for x in case["cfg"]]
  return cfg + [1] if len(cfg) == 4 else cfg      # H, bi, char_dim, attn_hidden, layers


def build_oracle_pair(case, rnn_type, attn):
  H, bi, char_dim, ah, layers = dec_cfg(case)
  enc = O.OracleVideoEncoder(204, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bool(bi),
                             enable_ctc=True, vocab_size=64, char2idx=O.default_char2idx())
  enc.load_state_dict({k: torch.tensor(v) for k, v in _flatten(case["enc_sd"]).items()})
  dec = O.OracleCharDecodingStep(H * (2 if bi else 1), rnn_type, layers, char_dim, 64, O.default_char2idx(),
                                 attention_type=attn, attn_hidden_size=ah)
  dec.load_state_dict({k: torch.tensor(v) for k, v in _flatten(case["dec_sd"]).items()})
  return enc.train(), dec.train()


@pytest.mark.parametrize("name", sorted(DEC_CASES) + sorted(DEC2_CASES))
def test_decoder_oracle_matches_reference(golden_dec, golden_dec2, name):
  """Encoder -> attention decoder loop (tfr = 1) + CTC; decoder_loss.backward(retain_graph) then
  ctc_loss.backward(), as train_better_model.py:46-74."""
  case = (golden_dec2 if name in DEC2_CASES else golden_dec)[name]
  enc, dec = build_oracle_pair(case, *{**DEC_CASES, **DEC2_CASES}[name])
  lens = torch.tensor(case["lens"])
  chars, char_lens = torch.tensor(case["chars"]), torch.tensor(case["char_lens"])
  lp_enc, hid, final = enc(torch.tensor(case["frames"]), lens)
  ctc = O.ctc_loss(lp_enc, chars[:, 1:], lens, char_lens - 1, 'mean')
  dec_loss, outs = O.decoder_loop(dec, chars, char_lens, hid, lens, final)
  np.testing.assert_allclose(dec_loss.item(), float(case["dec_loss"]), rtol=1e-5)
  np.testing.assert_allclose(ctc.item(), float(case["ctc_loss"]), rtol=1e-5)
  np.testing.assert_allclose(outs.detach().numpy(), case["dec_log_probs"], rtol=1e-4, atol=1e-5)
  dec_loss.backward(retain_graph=True)
  ctc.backward()
  for mod, key in ((enc, "enc_grad"), (dec, "dec_grad")):
    want = _flatten(case[key])
    for k, p in mod.named_parameters():
      g = p.grad if p.grad is not None else torch.zeros_like(p)
      np.testing.assert_allclose(g.numpy(), want[k], rtol=2e-4, atol=2e-6, err_msg=k)
