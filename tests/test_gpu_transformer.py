"""GPU checks of the build-defined transformer encoder (no reference parity: the
oracle is torch.nn.TransformerEncoder on the CPU)."""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
  assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
  return torch.device("cuda:0")


def make_pair(dev, frame_dim=204, d_model=64, nhead=4, layers=2, ff=128, seed=0):
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.transformer import TransformerVideoEncoder
  torch.manual_seed(seed)
  ref = O.OracleTransformerEncoder(frame_dim, d_model, nhead, layers, ff, 64, O.default_char2idx())
  enc = TransformerVideoEncoder(frame_dim, d_model, nhead, layers, ff, enable_ctc=True, vocab_size=64,
                                char2idx=default_char2idx())
  sd = {k: v for k, v in ref.state_dict().items() if not k.startswith("encoder.")}
  res = enc.load_state_dict(sd)
  assert not res.missing_keys and not res.unexpected_keys   # torch's own parameter names
  return ref.train(), enc.to(dev).train()


@pytest.mark.parametrize("lens", [[20, 20, 20, 20], [7, 12, 20, 20]])
def test_transformer_forward_backward_matches_torch(dev, lens):
  ref, enc = make_pair(dev)
  g = torch.Generator().manual_seed(1)
  lens = torch.tensor(lens)
  x = torch.randn(4, 20, 68, 3, generator=g)
  wgt = torch.randn(4, 20, 65, generator=g)
  valid = (torch.arange(20).unsqueeze(0) < lens.unsqueeze(1)).float().unsqueeze(-1)
  lp_r, h_r = ref(x, lens)
  ((lp_r * wgt * valid).sum() + (h_r * valid).pow(2).sum()).backward()
  lp, h, _ = enc(x.to(dev), lens, max_len=20)
  vd = valid.to(dev)
  ((lp * wgt.to(dev) * vd).sum() + (h * vd).pow(2).sum()).backward()
  m = valid.bool().squeeze(-1)
  np.testing.assert_allclose(lp.detach().cpu()[m].numpy(), lp_r.detach()[m].numpy(), rtol=2e-4, atol=2e-5)
  np.testing.assert_allclose(h.detach().cpu()[m].numpy(), h_r.detach()[m].numpy(), rtol=2e-4, atol=2e-5)
  want = dict(ref.named_parameters())   # (the shared layers are listed once, under "encoder.")
  for k, p in enc.named_parameters():
    r = (want[k] if k in want else want["encoder." + k]).grad.numpy()
    assert np.abs(p.grad.cpu().numpy() - r).max() / max(1e-5, np.abs(r).max()) < 3e-4, k


def test_transformer_bf16x3_linears_track_fp32(dev):
  """Pixel-regime option: every projection on the bf16 matrix cores with hi/lo split operands."""
  _, enc = make_pair(dev, frame_dim=864, d_model=128, ff=256, seed=5)
  g = torch.Generator().manual_seed(6)
  x = torch.randn(8, 40, 864, 1, generator=g).to(dev)
  lens = torch.full((8,), 40)
  wgt = torch.randn(8, 40, 65, generator=g).to(dev)
  res = {}
  for mode in ("f32", "bf16x3"):
    enc.input_projection = mode
    enc.zero_grad()
    lp, h, _ = enc(x, lens, max_len=40)
    ((lp * wgt).sum() + h.pow(2).sum()).backward()
    res[mode] = [lp.detach().cpu(), h.detach().cpu()] + [p.grad.cpu().clone() for p in enc.parameters()]
  enc.input_projection = "f32"
  # every product carries ~1e-5 relative error (~1e-3 after 13 projections and 4 LayerNorms); a ReLU
  # whose pre-activation sits within that of zero may switch, which moves single entries of the
  # feed-forward gradients by a sample's worth — so the comparison is norm-wise
  for a, b in zip(res["f32"], res["bf16x3"]):
    assert float((a - b).norm()) / max(1e-6, float(a.norm())) < 3e-2   # (lr_xgemm itself: 3e-5, test_gpu_encoder)


def test_transformer_ctc_training_step_runs_and_learns(dev):
  from lipreading_amd.ctc import ctc_loss_with_status
  from lipreading_amd.optim import FlatParameters, FusedAdam
  _, enc = make_pair(dev, seed=3)
  opt = FusedAdam(FlatParameters(enc), lr=2e-3)
  g = torch.Generator().manual_seed(4)
  x = torch.randn(6, 30, 68, 3, generator=g).to(dev)
  lens = torch.full((6,), 30, device=dev)
  labels = torch.randint(4, 64, (6, 8), generator=g).to(dev)
  ll = torch.full((6,), 8, device=dev)
  losses = []
  for _ in range(30):
    opt.zero_grad()
    lp, _, _ = enc(x, lens, max_len=30)
    loss, status, _ = ctc_loss_with_status(lp, labels, lens, ll, 'mean')
    loss.backward()
    opt.step(grad_norm=50, skip=status)
    losses.append(float(loss.detach()))
  assert np.isfinite(losses).all() and losses[-1] < 0.7 * losses[0], losses
