"""GPU parity: CTC loss / gradient / greedy decode / landmark step / collation through the C ABI
against (a) the vectors captured from the reference and (b) the oracle on seeded inputs."""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu

CTC_CASES = ["equal", "runs112", "runs233", "repeats", "toolong", "alltoolong", "infsample",
             "infrun", "inffirst", "allinf", "bench", "mixed75"]
LOSS_TOL = 1e-4   # BASELINE.json north_star: CTC loss within 1e-4 of the CPU reference (fp32)


@pytest.fixture(scope="module")
def dev():
  assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
  return torch.device("cuda:0")


@pytest.mark.parametrize("name", CTC_CASES)
@pytest.mark.parametrize("red", ["mean", "sum"])
def test_ctc_matches_reference_vectors(golden_ctc, dev, name, red):
  from lipreading_amd.ctc import ctc_loss
  c = golden_ctc[name]
  lp = torch.tensor(c["lp"], device=dev, requires_grad=True)
  loss = ctc_loss(lp, torch.tensor(c["labels"], device=dev), torch.tensor(c["frame_lens"], device=dev),
                  torch.tensor(c["label_lens"], device=dev), red, dev)
  if int(c["none_" + red]):
    assert loss is None
    return
  assert loss is not None
  ref = float(c["loss_" + red])
  assert abs(loss.item() - ref) <= LOSS_TOL * max(1.0, abs(ref) / 10), (loss.item(), ref)
  loss.backward()
  np.testing.assert_allclose(lp.grad.cpu().numpy(), c["grad_" + red], rtol=1e-4, atol=2e-6)


def test_ctc_transposed_view_and_second_backward(golden_ctc, dev):
  """The reference hands F.ctc_loss a (T,B,C) transposed view (ctc_loss.py:85); strides are
  part of the ABI.  A retained graph may be traversed twice (train_better_model.py:69,74)."""
  from lipreading_amd.ctc import ctc_loss
  c = golden_ctc["runs233"]
  base = torch.tensor(c["lp"], device=dev).transpose(0, 1).contiguous()   # (T,B,C) storage
  lp = base.transpose(0, 1).requires_grad_(True)                          # (B,T,C) view
  args = (torch.tensor(c["labels"], device=dev), torch.tensor(c["frame_lens"], device=dev),
          torch.tensor(c["label_lens"], device=dev), "mean", dev)
  loss = ctc_loss(lp, *args)
  assert abs(loss.item() - float(c["loss_mean"])) <= LOSS_TOL
  g1, = torch.autograd.grad(loss, lp, retain_graph=True)
  g2, = torch.autograd.grad(loss, lp)
  np.testing.assert_allclose(g1.cpu().numpy(), c["grad_mean"], rtol=1e-4, atol=2e-6)
  np.testing.assert_array_equal(g1.cpu().numpy(), g2.cpu().numpy())


@pytest.mark.parametrize("B,T,L", [(32, 75, 31), (64, 75, 31), (7, 150, 120), (3, 300, 256)])
def test_ctc_matches_oracle_seeded(dev, B, T, L):
  """BASELINE shapes (B=32/64, T=75, L=30+EOS) and the label-length ceiling (2L+1 = 513 states)."""
  from lipreading_amd.ctc import ctc_loss_with_status
  g = torch.Generator().manual_seed(123456)
  lp = torch.log_softmax(torch.randn(B, T, 65, generator=g), -1)
  lens_l = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
  lens_l[-1] = L
  labels = torch.randint(4, 64, (B, L), generator=g)
  for b in range(B):
    labels[b, int(lens_l[b]) - 1] = 2
    labels[b, int(lens_l[b]):] = 0
  fl = torch.sort(torch.randint(min(T, max(L + L // 4, T // 2)), T + 1, (B,), generator=g))[0]
  fl[-1] = T
  for red in ("mean", "sum"):
    x = lp.clone().requires_grad_(True)
    ref = O.ctc_loss(x, labels, fl, lens_l, red)
    xd = lp.to(dev).requires_grad_(True)
    loss, status, nll = ctc_loss_with_status(xd, labels.to(dev), fl.to(dev), lens_l.to(dev), red)
    assert int(status.item()) == (1 if ref is None else 0)
    if ref is None:
      continue
    tol = LOSS_TOL * max(1.0, abs(ref.item()) / 10)
    assert abs(loss.item() - ref.item()) <= tol, (loss.item(), ref.item())
    ref.backward()
    loss.backward()
    got, want = xd.grad.cpu().numpy(), x.grad.numpy()
    if T <= 150:
      np.testing.assert_allclose(got, want, rtol=2e-4, atol=5e-6)
    else:
      # nll ~ 1e3 here: exp(lcab + nll - lp) cancels two ~1e3 fp32 numbers, so BOTH fp32
      # implementations carry ~1e-4 relative noise.  Judge each against an fp64 evaluation of
      # the oracle and require the HIP path to be no worse than the fp32 CPU path (x2 slack).
      x64 = lp.double().requires_grad_(True)
      O.ctc_loss(x64, labels, fl, lens_l, red).backward()
      truth = x64.grad.numpy()
      err_hip = np.abs(got - truth).max()
      err_cpu = np.abs(want - truth).max()
      assert err_hip <= max(2 * err_cpu, 1e-5), (err_hip, err_cpu)


def test_ctc_linearity_in_grad_output(dev):
  """size-independent property: d(k*loss) = k * d(loss)."""
  from lipreading_amd.ctc import ctc_loss_with_status
  g = torch.Generator().manual_seed(7)
  lp = torch.log_softmax(torch.randn(16, 75, 65, generator=g), -1).to(dev)
  labels = torch.randint(4, 64, (16, 31), generator=g).to(dev)
  fl = torch.full((16,), 75, device=dev)
  ll = torch.full((16,), 31, device=dev)
  grads = []
  for k in (1.0, 3.0):
    x = lp.clone().requires_grad_(True)
    loss, _, _ = ctc_loss_with_status(x, labels, fl, ll, "sum")
    (loss * k).backward()
    grads.append(x.grad)
  np.testing.assert_allclose((grads[0] * 3.0).cpu().numpy(), grads[1].cpu().numpy(), rtol=1e-6)
  # posterior occupancies sum to one per valid frame => rows of the torch-style gradient sum to 0
  np.testing.assert_allclose(grads[0].sum(-1).cpu().numpy(), 0.0, atol=1e-4)


def test_greedy_matches_reference_vectors(golden_greedy, dev):
  from lipreading_amd.decoder import GreedyDecoder, ctc_labels
  from lipreading_amd.data import default_char2idx
  g = golden_greedy
  labels = ctc_labels(default_char2idx())
  assert labels == list(g["labels"])
  dec = GreedyDecoder(labels, blank_index=0)
  lp = torch.tensor(g["lp"], device=dev)
  strings, offsets = dec.decode(lp, torch.tensor(g["sizes"], device=dev))
  assert [s[0] for s in strings] == list(g["strings"])
  for b in range(len(strings)):
    np.testing.assert_array_equal(offsets[b][0].numpy(), g["offsets_%d" % b])
    assert offsets[b][0].dtype == torch.int32
  strings2, _ = dec.decode(lp)
  assert [s[0] for s in strings2] == list(g["strings_nosizes"])


def test_greedy_matches_oracle_seeded_and_duplicate_labels(dev):
  from lipreading_amd.decoder import GreedyDecoder
  g = torch.Generator().manual_seed(99)
  B, T, C = 64, 75, 65
  # low-entropy paths so that repeats and blanks are frequent
  path = torch.randint(0, 6, (B, T), generator=g) * 7
  lp = torch.log_softmax(torch.randn(B, T, C, generator=g), -1)
  lp.scatter_(2, path.unsqueeze(-1), 2.0)
  sizes = torch.randint(1, T + 1, (B,), generator=g)
  labels = O.ctc_labels()
  s_ref, o_ref = O.greedy_decode(lp, sizes, labels)
  s, o = GreedyDecoder(labels).decode(lp.to(dev), sizes.to(dev))
  assert s == s_ref
  for b in range(B):
    np.testing.assert_array_equal(o[b][0].numpy(), o_ref[b][0].numpy())
  # duplicate label strings collapse by CHARACTER (decoder.py:167-171)
  dup = list(labels)
  dup[14] = dup[7]
  s_ref, _ = O.greedy_decode(lp, sizes, dup)
  s, _ = GreedyDecoder(dup).decode(lp.to(dev), sizes.to(dev))
  assert s == s_ref


def test_greedy_without_space_label_raises_like_the_reference(dev):
  from lipreading_amd.decoder import GreedyDecoder
  lp = torch.zeros(1, 4, 3, device=dev)
  lp[0, :, 1] = 1.0
  with pytest.raises(IndexError):
    GreedyDecoder(['_', 'a', 'b']).decode(lp)


def test_landmark_step(golden_lmk, dev):
  from lipreading_amd import landmarks as LM
  g = golden_lmk
  padded = LM.apply_padding(torch.tensor(g["dims"], device=dev), torch.tensor(g["rects"], device=dev),
                            float(g["padding"]))
  np.testing.assert_array_equal(padded.cpu().numpy(), g["padded"])
  face = LM.get_face(torch.tensor(g["lmk"], device=dev), padded)
  np.testing.assert_array_equal(face.cpu().numpy(), g["face"].astype(np.float32))   # reference vectors, bit-exact
  # seeded, against the oracle: a 75-frame clip of 68 points
  rng = np.random.RandomState(123456)
  lm = rng.uniform(0, 200, (75, 68, 3)).astype(np.float32)
  rects = rng.randint(0, 300, (75, 4)).astype(np.int32)
  rects[:, 1] += rects[:, 0]
  rects[:, 3] += rects[:, 2]
  dims = np.tile(np.array([[720, 1280]]), (75, 1))
  pd = LM.apply_padding(torch.tensor(dims, device=dev), torch.tensor(rects, device=dev), 0.3)
  exp_pd = np.array([O.apply_padding(tuple(dims[i]), tuple(int(v) for v in rects[i]), 0.3)
                     for i in range(75)])
  np.testing.assert_array_equal(pd.cpu().numpy(), exp_pd)
  out = LM.get_face(torch.tensor(lm, device=dev), pd)
  exp = np.stack([O.get_face(lm[i], exp_pd[i]) for i in range(75)])
  np.testing.assert_array_equal(out.cpu().numpy(), exp)


def test_prnet_crop_restore_gather(golden_prn, dev):
  """A7 remainder through the C ABI against the reference vectors (prn_cases: PRN.process and
  get_landmarks bodies run on a seeded position map, composed as generate_dataview.py:58-64) and, on a
  75-frame clip, against the oracle.  Stated tolerances: integer rect math bit-exact; transform and
  coordinates 1e-9 absolute (float64 closed form vs LAPACK inverse + np.dot; coordinates are ~1e2..1e3)."""
  from lipreading_amd import landmarks as LM
  from tests.golden.make_golden import prn_position_map
  g = golden_prn
  uv = torch.tensor(g["uv_kpt_ind"], device=dev)
  n = len(g["rects"])
  cropped = np.stack([prn_position_map(int(g["seeds"][i]), g["rects"][i]) for i in range(n)])
  rects, dims = torch.tensor(g["rects"], device=dev), torch.tensor(g["dims"], device=dev)
  tform, sizes = LM.crop_transform(rects)
  np.testing.assert_allclose(tform.cpu().numpy(), g["tform"], rtol=0, atol=1e-12)
  pos = LM.restore(torch.tensor(cropped, device=dev), tform)
  assert pos.dtype == torch.float64 and pos.shape == (n, 256, 256, 3)
  np.testing.assert_allclose(pos.cpu().numpy()[:, ::16, ::16], g["pos_sub16"], rtol=0, atol=1e-9)
  np.testing.assert_allclose(LM.get_landmarks(pos, uv).cpu().numpy(), g["kpt"], rtol=0, atol=1e-9)
  padded = LM.apply_padding(dims, rects, 0.3)
  np.testing.assert_array_equal(padded.cpu().numpy(), g["padded"])
  np.testing.assert_allclose(LM.get_landmarks(pos, uv, padded).cpu().numpy(), g["face_lmk"], rtol=0, atol=1e-9)
  # the fused launch: the whole of _gen_data around the two networks
  lmk, padded2 = LM.landmark_step(torch.tensor(cropped, device=dev), rects, dims, uv, padding=0.3)
  np.testing.assert_array_equal(padded2.cpu().numpy(), g["padded"])
  np.testing.assert_allclose(lmk.cpu().numpy(), g["face_lmk"], rtol=0, atol=1e-9)
  lmk32, _ = LM.landmark_step(torch.tensor(cropped, device=dev), rects, dims, uv, padding=0.3, dtype=torch.float32)
  np.testing.assert_array_equal(lmk32.cpu().numpy(), g["face_lmk"].astype(np.float32))   # what _collate_fn makes of the row
  # a 75-frame clip against the oracle
  rng = np.random.RandomState(7)
  F = 75
  r = np.zeros((F, 4), np.int32)
  r[:, 0], r[:, 2] = rng.randint(0, 900, F), rng.randint(0, 500, F)
  r[:, 1], r[:, 3] = r[:, 0] + rng.randint(40, 380, F), r[:, 2] + rng.randint(40, 220, F)
  d = np.tile(np.array([[720, 1280, 3]]), (F, 1))
  cp = rng.uniform(-50, 300, (F, 256, 256, 3)).astype(np.float32)
  got, gp = LM.landmark_step(torch.tensor(cp, device=dev), torch.tensor(r, device=dev), torch.tensor(d, device=dev), uv)
  for i in range(F):
    want, wp = O.landmark_step(cp[i], tuple(int(v) for v in r[i]), tuple(d[i]), g["uv_kpt_ind"], 0.3)
    assert list(wp) == gp[i].tolist()
    np.testing.assert_allclose(got[i].cpu().numpy(), want, rtol=0, atol=1e-9)


def test_collate_matches_oracle(dev):
  from lipreading_amd.data import make_collate_fn
  rng = np.random.RandomState(5)
  lens = [1, 17, 75, 40, 75]
  batch = [(rng.randn(n, 68, 3), np.r_[1, rng.randint(4, 64, 3 + i), 2]) for i, n in enumerate(lens)]
  f, fl, c, cl = make_collate_fn(dev)(batch)
  rf, rfl, rc, rcl = O.collate([b[0] for b in batch], [b[1] for b in batch])
  assert f.shape == rf.shape and f.dtype == torch.float32 and c.dtype == torch.long
  np.testing.assert_array_equal(f.cpu().numpy(), rf.numpy())
  np.testing.assert_array_equal(c.cpu().numpy(), rc.numpy())
  assert fl.tolist() == rfl.tolist() and cl.tolist() == rcl.tolist()


def test_ctc_prepare_equals_the_reference_label_plumbing(dev):
  """lr_ctc_prepare_i64 (one launch) against the reference's own expressions: labels = chars[:, 1:] and
  label_lens = char_lens - 1 (train_better_model.py:31-32), then int32 integers with labels moved up by one
  (ctc_loss.py:42,80) — bit-exact, on a non-contiguous `chars` view too; and the loss through the prepared
  inputs equals the loss through the converting path."""
  from lipreading_amd.ctc import ctc_loss_prepared, ctc_loss_with_status, prepare_ctc_inputs
  g = torch.Generator().manual_seed(12)
  B, Lc, T, C = 7, 9, 20, 65
  wide = torch.randint(0, 64, (B, Lc + 3), generator=g)
  chars = wide[:, :Lc].to(dev)                      # row stride Lc + 3: a view, as collate may hand over
  frame_lens = torch.sort(torch.randint(12, T + 1, (B,), generator=g))[0].to(dev)
  char_lens = torch.randint(3, Lc + 1, (B,), generator=g).to(dev)
  lab, fl, ll = prepare_ctc_inputs(chars, frame_lens, char_lens)
  assert lab.dtype == fl.dtype == ll.dtype == torch.int32
  assert torch.equal(lab.cpu(), (chars[:, 1:].cpu() + 1).to(torch.int32))
  assert torch.equal(fl.cpu(), frame_lens.cpu().to(torch.int32))
  assert torch.equal(ll.cpu(), (char_lens.cpu() - 1).to(torch.int32))
  lp = torch.log_softmax(torch.randn(B, T, C, generator=g), dim=-1).to(dev).requires_grad_(True)
  a, sa, _ = ctc_loss_prepared(lp, lab, fl, ll, 'mean')
  b, sb, _ = ctc_loss_with_status(lp, chars[:, 1:], frame_lens, char_lens - 1, 'mean')
  assert int(sa) == int(sb) and float(a.detach()) == float(b.detach())
  ga, = torch.autograd.grad(a, lp)
  gb, = torch.autograd.grad(b, lp)
  assert torch.equal(ga, gb)
