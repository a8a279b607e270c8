"""The plain-C oracle (oracle/lr_oracle.c) against the vectors captured from the reference and
against the torch oracle.  CPU only.  Two independently written restatements agreeing with the
reference's own outputs is what the GPU parity tests lean on."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import torch_oracle as O
from tests.test_oracle_golden import CTC_CASES, ENC_CASES, _flatten


@pytest.mark.parametrize("name", ENC_CASES)
@pytest.mark.parametrize("tag", ["eq", "mix"])
def test_c_encoder_matches_reference_vectors(golden_enc, name, tag):
  case = golden_enc[name]
  H, layers, bi = [int(v) for v in case["cfg"]]
  D = 2 if bi else 1
  mode = "GRU" if name.startswith("gru") else "LSTM"
  sd = _flatten(case["sd"])
  io = case[tag]
  x = io["frames"].reshape(io["frames"].shape[0], io["frames"].shape[1], -1)
  x = x[:, :int(io["lens"].max())]
  h_fin, c_fin = [], []
  for layer in range(layers):
    sfx = ["", "_reverse"][:D]
    w = {k: [sd["rnn.%s_l%d%s" % (k, layer, s)] for s in sfx]
         for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
    x, h_n, c_n = CO.rnn_layer(mode, x, io["lens"], w["weight_ih"], w["weight_hh"], w["bias_ih"],
                               w["bias_hh"])
    h_fin.append(np.concatenate(list(h_n), axis=-1))
    if c_n is not None:
      c_fin.append(np.concatenate(list(c_n), axis=-1))
  np.testing.assert_allclose(x, io["hidden"], rtol=1e-5, atol=2e-6)
  np.testing.assert_allclose(np.stack(h_fin), io["h_n"], rtol=1e-5, atol=2e-6)
  if c_fin:
    np.testing.assert_allclose(np.stack(c_fin), io["c_n"], rtol=1e-5, atol=2e-6)
  mask = np.ones(65, np.float32)
  mask[1] = mask[2] = 0
  lp = CO.proj_logsoftmax(x, sd["output_proj.weight"], sd["output_proj.bias"], mask)
  np.testing.assert_allclose(lp, io["log_probs"], rtol=1e-5, atol=2e-5)
  assert abs(float(lp[..., 1].max()) + 103.28) < 20  # finite, about -103 below the rest


@pytest.mark.parametrize("name", CTC_CASES)
@pytest.mark.parametrize("red", ["mean", "sum"])
def test_c_ctc_matches_reference_vectors(golden_ctc, name, red):
  c = golden_ctc[name]
  keep = c["label_lens"] <= 256          # oracle_ctc_reduce drops these; oracle_ctc needs L<=T-ish
  lab = np.minimum(c["labels"] + 1, 64).astype(np.int32)
  ll = np.where(keep, c["label_lens"], 0)
  nll, grad = CO.ctc(c["lp"], lab, c["frame_lens"], ll)
  loss, w = CO.ctc_reduce(nll, c["frame_lens"], c["label_lens"], red)
  if int(c["none_" + red]):
    assert loss is None
    return
  assert loss is not None
  np.testing.assert_allclose(loss, float(c["loss_" + red]), rtol=2e-6, atol=2e-6)
  g = grad * w[:, None, None]
  np.testing.assert_allclose(g, c["grad_" + red], rtol=2e-4, atol=2e-5)  # fp64 here, fp32 in torch


def test_c_ctc_agrees_with_torch_oracle_at_bench_shape():
  g = torch.Generator().manual_seed(123456)
  lp = torch.log_softmax(torch.randn(8, 75, 65, generator=g), -1)
  labels = torch.randint(4, 64, (8, 31), generator=g)
  fl, ll = torch.full((8,), 75), torch.full((8,), 31)
  nll_t = O.ctc_nll_per_sample(lp, labels + 1, fl, ll).numpy()
  nll_c, _ = CO.ctc(lp.numpy(), (labels + 1).numpy(), fl.numpy(), ll.numpy(), want_grad=False)
  np.testing.assert_allclose(nll_c, nll_t, rtol=2e-6)


def test_c_greedy_matches_reference_vectors(golden_greedy):
  g = golden_greedy
  labels = list(g["labels"])
  ids, off, lens = CO.greedy(g["lp"], g["sizes"])
  for b in range(len(lens)):
    s = "".join(labels[i] for i in ids[b, :lens[b]])
    assert s == g["strings"][b]
    np.testing.assert_array_equal(off[b, :lens[b]], g["offsets_%d" % b])


def test_c_landmarks_match_reference(golden_lmk):
  g = golden_lmk
  padded = CO.apply_padding(g["rects"], g["dims"], float(g["padding"]))
  np.testing.assert_array_equal(padded, g["padded"])
  np.testing.assert_array_equal(CO.get_face(g["lmk"], padded), g["face"].astype(np.float32))
