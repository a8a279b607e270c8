"""GPU parity: fp32-MFMA GEMM, recurrent layer (fwd+bwd), projection + masked log-softmax and
the whole VideoEncoder through the C ABI, against the reference vectors and the oracle."""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from tests.test_oracle_golden import ENC2_CASES, ENC_CASES, _flatten, build_oracle_encoder, rnn_type_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
  assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
  return torch.device("cuda:0")


def hip_encoder_from(case, rnn_type, dev):
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.data import default_char2idx
  H, layers, bi = [int(x) for x in case["cfg"]]
  enc = VideoEncoder(204, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bool(bi),
                     enable_ctc=True, vocab_size=64, char2idx=default_char2idx())
  key = "sd" if "sd" in case else "sd0"
  missing = enc.load_state_dict({k: torch.tensor(v) for k, v in _flatten(case[key]).items()})
  assert not missing.missing_keys and not missing.unexpected_keys   # reference key names
  return enc.to(dev)


# ---- GEMM -----------------------------------------------------------------------------------
def run_sgemm(ta, tb, M, N, K, dev, alpha=1.0, beta=0.0, bias=False, shift=0, period=0, seed=0):
  from lipreading_amd import _C
  L = _C.lib()
  g = torch.Generator().manual_seed(seed)
  A = torch.randn((K, M) if ta else (M, K), generator=g)
  B = torch.randn((N, K) if tb else (K, N), generator=g)
  C0 = torch.randn(M, N, generator=g)
  bv = torch.randn(N, generator=g) if bias else None
  opA = A.t() if ta else A
  opB = B.t() if tb else B
  if period:
    rows = torch.arange(K)
    ok = ((rows % period + shift) >= 0) & ((rows % period + shift) < period)
    src = (rows + shift).clamp(0, K - 1)
    opB = torch.where(ok[:, None], opB[src], torch.zeros_like(opB))
  ref = alpha * (opA.double() @ opB.double()) + beta * C0.double()
  if bias:
    ref = ref + bv.double()
  Ad, Bd, Cd = A.to(dev), B.to(dev), C0.clone().to(dev)
  bd = bv.to(dev) if bias else None
  wsb = L.lr_sgemm_workspace_bytes(M, N, K)
  ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
  _C.check(L.lr_sgemm(int(ta), int(tb), M, N, K, alpha, Ad.data_ptr(), A.shape[1], Bd.data_ptr(),
                      B.shape[1], beta, Cd.data_ptr(), N, _C.ptr(bd), shift, period,
                      ws.data_ptr() if wsb else None, wsb, _C.stream_handle()), "lr_sgemm")
  scale = max(1.0, float(ref.abs().max()))
  err = float((Cd.cpu().double() - ref).abs().max()) / scale
  assert err < 2e-6 * max(1, K) ** 0.5 + 1e-6, err


def run_xgemm(ta, tb, M, N, K, dev, alpha=1.0, beta=0.0, bias=False, a_exact=False, b_exact=False, seed=0):
  """lr_xgemm (fp32 operands split into bf16 hi+lo on the bf16 matrix cores) vs an fp64 product."""
  from lipreading_amd import _C
  L = _C.lib()
  g = torch.Generator().manual_seed(seed)
  A = torch.randn((K, M) if ta else (M, K), generator=g)
  B = torch.randn((N, K) if tb else (K, N), generator=g)
  if a_exact:
    A = A.to(torch.bfloat16).float()
  if b_exact:
    B = B.to(torch.bfloat16).float()
  C0 = torch.randn(M, N, generator=g)
  bv = torch.randn(N, generator=g) if bias else None
  ref = alpha * ((A.t() if ta else A).double() @ (B.t() if tb else B).double()) + beta * C0.double()
  if bias:
    ref = ref + bv.double()
  Ad, Bd, Cd = A.to(dev), B.to(dev), C0.clone().to(dev)
  bd = bv.to(dev) if bias else None
  wsb = L.lr_xgemm_workspace_bytes(int(ta), int(tb), M, N, K)
  ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
  _C.check(L.lr_xgemm(int(ta), int(tb), M, N, K, alpha, Ad.data_ptr(), A.shape[1], Bd.data_ptr(), B.shape[1],
                      beta, Cd.data_ptr(), N, _C.ptr(bd), int(a_exact), int(b_exact),
                      ws.data_ptr() if wsb else None, wsb, _C.stream_handle()), "lr_xgemm")
  # error model: each product carries ~2^-17 relative error (16 mantissa bits per operand), random sign
  scale = max(1.0, float(ref.abs().max()))
  err = float((Cd.cpu().double() - ref).abs().max()) / scale
  assert err < 3e-5, err
  return err


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0)])
@pytest.mark.parametrize("M,N,K", [(37, 65, 204), (2400, 768, 3456), (130, 129, 31), (1, 1, 1)])
def test_xgemm_layouts_and_ragged_edges(dev, ta, tb, M, N, K):
  run_xgemm(ta, tb, M, N, K, dev, seed=M + N + K)


def test_xgemm_epilogue_and_exact_operands(dev):
  run_xgemm(0, 1, 300, 200, 64, dev, alpha=0.5, beta=2.0, bias=True)
  run_xgemm(0, 1, 2400, 768, 3456, dev, a_exact=True, bias=True)       # pixel regime forward projection
  run_xgemm(1, 0, 768, 3456, 2400, dev, b_exact=True, beta=1.0)        # dW_ih
  e3 = run_xgemm(0, 1, 256, 256, 1024, dev, seed=5)
  e1 = run_xgemm(0, 1, 256, 256, 1024, dev, a_exact=True, b_exact=True, seed=5)
  assert e1 < 2e-6 and e3 < 3e-5   # exact operands: only fp32 accumulation error remains


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(37, 65, 204), (2400, 768, 204), (130, 129, 31), (1, 1, 1)])
def test_sgemm_layouts_and_ragged_edges(dev, ta, tb, M, N, K):
  run_sgemm(ta, tb, M, N, K, dev, seed=M + N + K)


def test_sgemm_epilogue_splitk_and_row_shift(dev):
  run_sgemm(0, 1, 300, 200, 64, dev, alpha=0.5, beta=2.0, bias=True)
  run_sgemm(1, 0, 768, 256, 2400, dev)                       # dW_hh shape: split along K
  run_sgemm(1, 0, 65, 512, 2400, dev, alpha=2.0, beta=1.0)   # dW_proj shape, split + beta
  run_sgemm(1, 0, 96, 32, 150, dev, shift=-1, period=75)     # h_prev view, forward direction
  run_sgemm(1, 0, 96, 32, 150, dev, shift=1, period=75)      # h_prev view, reverse direction
  run_sgemm(0, 0, 2400, 3072, 768, dev)                      # 128x128 tiles


# ---- encoder forward against the reference vectors -------------------------------------------
@pytest.mark.parametrize("R,K,C", [(2400, 512, 65), (37, 200, 29), (100, 520, 128), (16, 256, 1), (33, 64, 130)])
def test_output_head_backward_against_torch(dev, R, K, C):
  """lr_proj_logsoftmax_backward (C <= 128: the fused rows kernel + slab reduction of lr_proj.hip, two launches; wider
  heads: the GEMM chain) against autograd of log_softmax(hidden W^T + b): ragged row blocks (R not a multiple of 16),
  K that is not a multiple of the 256 columns a workgroup owns, C at and past the fused path's limit, a single class,
  dhidden = NULL, and accumulate = 1 on top of existing dW / dbias."""
  from lipreading_amd import _C
  L = _C.lib()
  st = _C.stream_handle()
  g = torch.Generator().manual_seed(R + K + C)
  hidden = torch.randn(R, K, generator=g, requires_grad=True)
  W = (torch.randn(C, K, generator=g) / K ** 0.5).requires_grad_(True)
  b = torch.randn(C, generator=g).requires_grad_(True)
  gout = torch.randn(R, C, generator=g)
  lp = torch.log_softmax(hidden @ W.t() + b, dim=1)
  lp.backward(gout)
  hd, Wd, lpd, gd = hidden.detach().to(dev), W.detach().to(dev), lp.detach().to(dev), gout.to(dev)
  wsb = L.lr_proj_workspace_bytes(R, K, C)
  ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
  for accumulate, want_dh in ((0, True), (1, False)):
    dlogits = torch.empty(R, C, device=dev)
    dh = torch.full((R, K), float("nan"), device=dev) if want_dh else None
    dW = torch.full((C, K), 0.5 if accumulate else float("nan"), device=dev)
    db = torch.full((C,), 0.25 if accumulate else float("nan"), device=dev)
    _C.check(L.lr_proj_logsoftmax_backward(gd.data_ptr(), lpd.data_ptr(), hd.data_ptr(), Wd.data_ptr(), dlogits.data_ptr(),
                                           dh.data_ptr() if want_dh else None, dW.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                           wsb, accumulate, R, K, C, st), "lr_proj_logsoftmax_backward")
    scale = float(W.grad.abs().max())
    assert float((dW.cpu() - 0.5 * accumulate - W.grad).abs().max()) <= 2e-5 * max(1.0, scale)
    assert float((db.cpu() - 0.25 * accumulate - b.grad).abs().max()) <= 2e-5 * max(1.0, float(b.grad.abs().max()))
    if want_dh:
      assert float((dh.cpu() - hidden.grad).abs().max()) <= 2e-5 * max(1.0, float(hidden.grad.abs().max()))


@pytest.mark.parametrize("name", ENC_CASES + ENC2_CASES)
@pytest.mark.parametrize("tag", ["eq", "mix"])
def test_encoder_matches_reference_vectors(golden_enc, golden_enc2, dev, name, tag):
  case = (golden_enc2 if name in ENC2_CASES else golden_enc)[name]
  enc = hip_encoder_from(case, rnn_type_of(name), dev).eval()
  io = case[tag]
  with torch.no_grad():
    lp, hid, fin = enc(torch.tensor(io["frames"], device=dev), torch.tensor(io["lens"]))
  np.testing.assert_allclose(lp.cpu().numpy(), io["log_probs"], rtol=1e-4, atol=2e-5)
  np.testing.assert_allclose(hid.cpu().numpy(), io["hidden"], rtol=1e-4, atol=2e-6)
  if isinstance(fin, tuple):
    np.testing.assert_allclose(fin[0].cpu().numpy(), io["h_n"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(fin[1].cpu().numpy(), io["c_n"], rtol=1e-4, atol=2e-6)
  else:
    np.testing.assert_allclose(fin.cpu().numpy(), io["h_n"], rtol=1e-4, atol=2e-6)
  # masked classes: finite and ~103.28 below, not -inf (subnormal 1e-45 kept on gfx950)
  masked = lp[..., 1:3].cpu().numpy()
  assert np.isfinite(masked).all() and (masked < -90).all()
  # padded positions: hidden is exactly zero
  for b, n in enumerate(io["lens"]):
    assert float(hid[b, int(n):].abs().sum()) == 0.0


# ---- encoder forward + backward against the oracle at real sizes ------------------------------
def make_pair(rnn_type, H, layers, bi, dev, seed=123456):
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.data import default_char2idx
  torch.manual_seed(seed)
  ref = O.OracleVideoEncoder(204, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bi,
                             enable_ctc=True, vocab_size=64, char2idx=O.default_char2idx())
  enc = VideoEncoder(204, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bi,
                     enable_ctc=True, vocab_size=64, char2idx=default_char2idx())
  enc.load_state_dict(ref.state_dict())
  return ref, enc.to(dev)


# (the reference's own shapes — LSTM-700: config/defaults.txt:19-21; LSTM-512: config/train/attn/attention_type:16-19;
#  GRU-800: config/train/micro:6-8 — run the one-launch cluster recurrence on the DEFAULT path, like GRU-256 / LSTM-768)
CFG = [("GRU", 256, 1, True, 32, 75), ("LSTM", 768, 1, True, 32, 75), ("GRU", 700, 1, True, 9, 40),
       ("LSTM", 700, 1, True, 32, 75), ("LSTM", 512, 1, True, 32, 75), ("GRU", 800, 2, True, 32, 75),
       ("LSTM", 64, 2, True, 20, 33), ("GRU", 48, 2, False, 17, 21), ("LSTM", 36, 1, False, 5, 12),
       ("RNN", 256, 1, True, 32, 75), ("RNN", 52, 2, False, 11, 19),
       # round 4's 16-unit-member clusters against the ORACLE (they were only compared with the step kernels)
       ("LSTM", 800, 1, True, 32, 75), ("LSTM", 1024, 1, False, 32, 31), ("GRU", 1100, 1, False, 13, 12),
       # round 6 against the ORACLE too: sixteen samples per cluster (BiLSTM-768 / -700 past B = 32: the ecd family's own
       # batch is 128), more than eight clusters per launch (BiLSTM-512 at configs[3]'s B = 64), the 192-CU grid (LSTM-1536)
       ("LSTM", 768, 1, True, 64, 40), ("LSTM", 700, 1, True, 72, 21), ("LSTM", 512, 1, True, 64, 40),
       ("LSTM", 1536, 1, False, 32, 31), ("LSTM", 1400, 1, False, 40, 12)]


@pytest.mark.parametrize("rnn_type,H,layers,bi,B,T", CFG)
def test_encoder_forward_backward_matches_oracle(dev, rnn_type, H, layers, bi, B, T):
  ref, enc = make_pair(rnn_type, H, layers, bi, dev)
  g = torch.Generator().manual_seed(1)
  lens = torch.sort(torch.randint(max(1, T // 2), T + 1, (B,), generator=g))[0]
  lens[-1] = T
  lens[0] = max(1, T // 3)
  frames = torch.randn(B, T, 68, 3, generator=g)
  for b in range(B):
    frames[b, int(lens[b]):] = 0
  w_lp = torch.randn(B, T, 65, generator=g) / 50
  w_h = torch.randn(B, T, (2 if bi else 1) * H, generator=g) / 50

  def loss_of(out):
    lp, hid, fin = out
    l = (lp * w_lp.to(lp.device)).sum() + (hid * w_h.to(hid.device)).sum()
    fins = fin if isinstance(fin, tuple) else (fin,)
    for f in fins:
      l = l + (f * 0.01).sum()
    return l

  from lipreading_amd import _C
  if rnn_type != "RNN" and H in (256, 512, 700, 768, 800, 1024, 1100, 1400, 1536):
    # the default path IS the one-launch recurrence for these
    assert _C.lib().lr_rnn_pair_supported({"GRU": 0, "LSTM": 1}[rnn_type], B, T, 204, H, 2 if bi else 1) == 2
  _C.lib().lr_rnn_pair_errors()
  out_r = ref(frames, lens)
  loss_of(out_r).backward()
  out_g = enc(frames.to(dev), lens)
  loss_of(out_g).backward()
  assert _C.lib().lr_rnn_pair_errors() == 0
  tol = 3e-5 if H <= 256 else 1e-4
  np.testing.assert_allclose(out_g[0].detach().cpu().numpy(), out_r[0].detach().numpy(), rtol=1e-4, atol=tol)
  np.testing.assert_allclose(out_g[1].detach().cpu().numpy(), out_r[1].detach().numpy(), rtol=1e-4, atol=tol)
  gr = dict(ref.named_parameters())
  for k, p in enc.named_parameters():
    a, b_ = p.grad.cpu().numpy(), gr[k].grad.numpy()
    scale = max(1e-3, float(np.abs(b_).max()))
    assert np.abs(a - b_).max() / scale < 2e-4, (k, np.abs(a - b_).max(), scale)


def test_encoder_two_backward_passes_accumulate(dev):
  """train_better_model.py:69,74: decoder_loss.backward(retain_graph=True) then
  ctc_loss_.backward() traverse the same encoder graph; gradients accumulate."""
  ref, enc = make_pair("GRU", 32, 1, True, dev)
  g = torch.Generator().manual_seed(3)
  frames = torch.randn(6, 10, 68, 3, generator=g)
  lens = torch.tensor([4, 6, 10, 10, 10, 10])
  outs = {}
  for name, m, x in (("ref", ref, frames), ("hip", enc, frames.to(dev))):
    lp, hid, fin = m(x, lens)
    hid.sum().backward(retain_graph=True)
    lp[..., 5].sum().backward()
    outs[name] = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
  for k in outs["ref"]:
    np.testing.assert_allclose(outs["hip"][k], outs["ref"][k], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("name", ["gru", "lstm"])
def test_ctc_step_matches_reference_vectors(golden_step, dev, name):
  """One CTC-only optimisation step (train_better_model.py:31-32,46-48,74,78,80)."""
  from lipreading_amd.ctc import ctc_loss
  case = golden_step[name]
  enc = hip_encoder_from(case, name.upper(), dev).train()
  opt = torch.optim.Adam(enc.parameters(), lr=1e-3)
  chars = torch.tensor(case["chars"], device=dev)
  lens = torch.tensor(case["lens"])
  lp, _, _ = enc(torch.tensor(case["frames"], device=dev), lens)
  loss = ctc_loss(lp, chars[:, 1:], lens.to(dev), torch.tensor(case["char_lens"], device=dev) - 1,
                  'mean', dev)
  assert abs(loss.item() - float(case["loss"])) < 1e-4
  opt.zero_grad()
  loss.backward()
  grads = _flatten(case["grad"])
  for k, p in enc.named_parameters():
    ref = grads[k]
    scale = max(1e-4, float(np.abs(ref).max()))
    assert np.abs(p.grad.cpu().numpy() - ref).max() / scale < 2e-4, k
  total = torch.nn.utils.clip_grad_norm_(enc.parameters(), 5.0)
  assert abs(float(total) - float(case["total_norm"])) < 1e-3 * float(case["total_norm"])
  opt.step()
  sd1 = _flatten(case["sd1"])
  for k, v in enc.state_dict().items():
    np.testing.assert_allclose(v.cpu().numpy(), sd1[k], rtol=1e-3, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("rnn_type", ["GRU", "LSTM"])
def test_bf16x3_input_projection_tracks_the_fp32_path(dev, rnn_type):
  """Pixel-regime option (LR_RNN_PROJ_BF16X3): the input projection and its two gradients on the
  bf16 matrix cores with hi/lo split operands stay within 1e-4 of the exact fp32 MFMA path — 2-layer
  encoder, feature-sized K, bf16-exact layer-0 input (what the conv frontend delivers)."""
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  torch.manual_seed(11)
  enc = VideoEncoder(864, 64, rnn_type=rnn_type, num_layers=2, bidirectional=True, enable_ctc=True,
                     vocab_size=64, char2idx=default_char2idx()).to(dev)
  g = torch.Generator().manual_seed(12)
  x = (torch.randn(6, 20, 864, 1, generator=g) * 0.5).to(torch.bfloat16).float()
  lens = torch.tensor([9, 12, 15, 20, 20, 20])
  wgt = torch.randn(6, 20, 65, generator=g).to(dev)
  res = {}
  for mode in ("f32", "bf16x3"):
    enc.input_projection, enc.input_is_bf16 = mode, mode == "bf16x3"
    enc.zero_grad()
    xd = x.to(dev).requires_grad_(True)
    lp, hid, _ = enc(xd, lens, max_len=20)
    ((lp * wgt).sum() + hid.pow(2).sum()).backward()
    res[mode] = [lp.detach().cpu(), hid.detach().cpu(), xd.grad.cpu()] + [p.grad.cpu().clone() for p in enc.parameters()]
  for i, (a, b) in enumerate(zip(res["f32"], res["bf16x3"])):
    scale = max(1e-6, float(a.abs().max()))
    # the gradient w.r.t. a bf16 input is contracted from the hi terms only (its consumer, the conv
    # frontend's backward, stores it as bf16 anyway): bf16-level agreement there, 1e-4 elsewhere
    assert float((a - b).abs().max()) / scale < (1e-2 if i == 2 else 1e-4), i


@pytest.mark.parametrize("B,T,bi,lens,launches", [(32, 75, True, None, 1),
                                                  (40, 30, True, "ragged", 1),     # 10 clusters: a 16-cluster launch (rounds 2-5: two launches of 8)
                                                  (70, 9, False, "ragged", 1),     # unidirectional: 9 clusters
                                                  (64, 75, True, None, 1),         # BASELINE configs[3]'s whole batch on one GPU: 16 clusters on 128 CUs
                                                  (128, 20, True, "ragged", 1),    # 32 clusters x 8 members = every CU of the chip
                                                  (136, 6, True, "ragged", 2),     # 34 clusters: a launch of 32 (16 groups) + the 17th group
                                                  (3, 1, True, None, 1), (5, 2, True, [2, 1, 2, 1, 1], 1)])
def test_split_recurrence_is_fp32_faithful(dev, B, T, bi, lens, launches):
  """LR_RNN_RECUR_SPLIT (VideoEncoder's default where supported) on GRU, H = 256: the whole recurrence of a
  layer pass in one launch, W_hh and the state as bf16 hi + lo planes on the 8-member cluster kernels
  (lr_rnn_cluster.hip; up to 32 clusters per launch, four to an XCD; rounds 2-4 also kept CU-pair kernels for devices too small for a cluster launch: removed in
  round 5, nothing on an MI355X reached them).  Against the exact-fp32 step kernels on the same weights (2 layers: the
  second layer's input is the first's output), forward and backward, ragged lengths, final-state gradient injected:
  agreement to ~1e-5 relative, and no member ever timed out."""
  from lipreading_amd import _C
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  torch.manual_seed(31)
  enc = VideoEncoder(96, 256, rnn_type='GRU', num_layers=2, bidirectional=bi, enable_ctc=True,
                     vocab_size=64, char2idx=default_char2idx()).to(dev)
  g = torch.Generator().manual_seed(32)
  x = torch.randn(B, T, 96, 1, generator=g)
  if lens == "ragged":
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
  elif lens is None:
    lens = torch.full((B,), T)
  else:
    lens = torch.tensor(lens)
  wgt = torch.randn(B, T, 65, generator=g).to(dev)
  valid = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).float().unsqueeze(-1).to(dev)
  assert _C.lib().lr_rnn_pair_supported(0, B, T, 96, 256, 2 if bi else 1) == 2
  # round 6: an XCD holds four 8-member clusters, a launch up to 32 of them — ONE launch per layer pass up to B = 128
  assert _C.lib().lr_rnn_pass_launches(0, B, T, 96, 256, 2 if bi else 1) == launches
  _C.lib().lr_rnn_pair_errors()
  res = {}
  for mode in ("f32", "split"):
    enc.recurrence = mode
    enc.zero_grad()
    lp, hid, fin = enc(x.to(dev), lens, max_len=T)
    ((lp * wgt * valid).sum() + hid.pow(2).sum() + 3.0 * fin.pow(2).sum()).backward()
    res[mode] = [lp.detach().cpu() * valid.cpu(), hid.detach().cpu(), fin.detach().cpu()] + \
                [p.grad.cpu().clone() for p in enc.parameters()]
  enc.recurrence = "auto"
  assert _C.lib().lr_rnn_pair_errors() == 0
  if T > 1:
    assert float((res["f32"][1] - res["split"][1]).abs().max()) > 0    # the other path really ran
  worst = 0.0
  for a, b in zip(res["f32"], res["split"]):
    worst = max(worst, float((a - b).norm()) / max(1e-6, float(a.norm())))
    assert float((a - b).norm()) / max(1e-6, float(a.norm())) < 2e-5
    assert float((a - b).abs().max()) <= 1e-4 * max(1e-6, float(a.abs().max()))
  print("split vs f32 recurrence: worst relative norm difference %.3g" % worst)
  assert float((res["split"][1] * (1 - valid.cpu())).abs().max()) == 0.0     # padded positions exactly zero


def test_split_recurrence_is_the_default_and_meets_the_loss_bar(dev):
  """VideoEncoder() with no switches at the bench shape (landmarks B=32, T=75, BiGRU-256) runs the
  one-launch fp32-faithful recurrence; CTC 'mean' loss within 1e-4 of the ORACLE (north_star), log-probs
  within 3e-5."""
  from lipreading_amd import _C
  from lipreading_amd.ctc import ctc_loss_with_status
  ref, enc = make_pair("GRU", 256, 1, True, dev)
  assert enc.recurrence == 'auto'
  g = torch.Generator().manual_seed(5)
  B, T = 32, 75
  lens = torch.sort(torch.randint(40, T + 1, (B,), generator=g))[0]
  lens[-8:] = T
  frames = torch.randn(B, T, 68, 3, generator=g)
  for b in range(B):
    frames[b, int(lens[b]):] = 0
  labels = torch.randint(4, 64, (B, 30), generator=g)
  ll = torch.full((B,), 30)
  with torch.no_grad():
    lp_r, _, _ = ref(frames, lens)
    lp_h, _, _ = enc(frames.to(dev), lens.to(dev), max_len=T)
    loss_r = O.ctc_loss(lp_r, labels, lens, ll, 'mean')
    loss_h, status, _ = ctc_loss_with_status(lp_h, labels.to(dev), lens.to(dev), ll.to(dev), 'mean')
  assert int(status) == 0 and _C.lib().lr_rnn_pair_errors() == 0
  valid = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(-1)
  d_lp = float(((lp_h.cpu() - lp_r) * valid).abs().max())
  assert abs(float(loss_h) - float(loss_r)) <= 1e-4 and d_lp <= 3e-5, (float(loss_h), float(loss_r), d_lp)


CLUSTER_CASES = [("LSTM", 768, 32, 75, True, None, 1), ("LSTM", 768, 37, 20, True, "ragged", 1),
                 ("LSTM", 768, 13, 6, False, "ragged", 1), ("LSTM", 768, 2, 1, True, None, 1),
                 ("LSTM", 768, 12, 9, True, "ragged", 2), ("LSTM", 768, 70, 5, False, "ragged", 1),
                 ("LSTM", 700, 32, 75, True, "ragged", 1), ("LSTM", 700, 11, 8, False, "ragged", 2),
                 ("LSTM", 512, 32, 75, True, None, 1), ("LSTM", 500, 9, 7, True, "ragged", 1),
                 ("GRU", 800, 32, 75, True, "ragged", 1), ("GRU", 800, 19, 10, False, "ragged", 2),
                 ("GRU", 512, 32, 31, False, None, 1), ("GRU", 780, 5, 4, True, "ragged", 1),
                 ("LSTM", 256, 20, 12, True, "ragged", 1), ("GRU", 704, 70, 5, True, "ragged", 1),
                 # round 4: every member count the kernels' storage holds (1 .. 27 GRU, 1 .. 24 LSTM), not a list of sizes
                 ("LSTM", 128, 32, 75, True, "ragged", 1), ("LSTM", 384, 32, 40, True, None, 1),
                 ("LSTM", 600, 32, 75, True, "ragged", 1), ("LSTM", 640, 9, 7, False, "ragged", 2),
                 ("LSTM", 32, 17, 9, True, "ragged", 1), ("LSTM", 12, 3, 5, True, "ragged", 2),
                 ("GRU", 40, 11, 6, True, "ragged", 1), ("GRU", 96, 32, 20, False, None, 1), ("GRU", 64, 32, 75, True, None, 2),
                 ("GRU", 864, 32, 30, True, "ragged", 1), ("GRU", 832, 10, 6, False, "ragged", 1),
                 ("GRU", 160, 8, 75, True, None, 1), ("LSTM", 200, 5, 11, True, "ragged", 1),
                 # 16-unit members on clusters of 50 .. 72 CUs that span XCDs (768 / 864 < H <= 1152)
                 ("LSTM", 1024, 32, 31, False, None, 1), ("LSTM", 1024, 4, 31, False, "ragged", 1),
                 ("LSTM", 800, 32, 40, True, "ragged", 1), ("LSTM", 800, 11, 9, False, "ragged", 2),
                 ("GRU", 1024, 13, 12, False, "ragged", 1), ("GRU", 896, 32, 20, True, None, 1),
                 ("LSTM", 1152, 9, 6, False, "ragged", 1), ("LSTM", 788, 20, 7, True, "ragged", 1),
                 ("GRU", 1100, 40, 5, False, "ragged", 1),
                 # round 6: more than 8 clusters per launch where an XCD holds several (8 floor(32 / members) of them):
                 # BiLSTM-512 at configs[3]'s B = 64 (16 clusters), 10-member clusters three to an XCD (24), 4-member
                 # ones at B = 130 (33 groups of a unidirectional layer: 64-cluster launches hold them all)
                 ("LSTM", 512, 64, 12, True, "ragged", 1), ("LSTM", 320, 96, 6, True, "ragged", 1),
                 ("GRU", 128, 130, 5, False, "ragged", 2), ("LSTM", 256, 128, 7, True, "ragged", 1),
                 # round 6: LSTM past 1152 units on the 24 x 8 grid of 192 CUs (lr_rnn_grid.hip): one and two blocks of 32
                 # samples, both directions (a launch each), a hidden size that pads (1400 -> 1536), more than 64 samples
                 ("LSTM", 1536, 32, 12, False, "ragged", 1), ("LSTM", 1400, 9, 5, True, "ragged", 1),
                 ("LSTM", 1280, 70, 4, False, "ragged", 1), ("LSTM", 1156, 40, 6, False, None, 2),
                 # round 6: SIXTEEN samples per cluster where a batch would take more than one launch of 8-sample clusters and
                 # the cluster fills an XCD (17 .. 24 members): BiLSTM-768 at B = 64 and 128 (the ecd family's own batch), BiLSTM-700,
                 # 18 GRU members, a partial last group, two layers — beside the (768, 37), (768, 70), (704, 70) cases above,
                 # which now take the 16-sample form too
                 ("LSTM", 768, 64, 12, True, "ragged", 1), ("LSTM", 768, 128, 5, True, "ragged", 1),
                 ("LSTM", 700, 100, 6, True, "ragged", 1), ("GRU", 576, 45, 9, True, "ragged", 2),
                 ("LSTM", 544, 72, 7, False, "ragged", 1)]
# (rnn_type, H, B, bidirectional) -> recurrence launches per layer pass (lr_rnn_pass_launches)
CLUSTER_LAUNCHES = {("LSTM", 512, 64, True): 1, ("LSTM", 320, 96, True): 1, ("GRU", 128, 130, False): 1,
                    ("LSTM", 256, 128, True): 1, ("LSTM", 768, 37, True): 1, ("LSTM", 768, 70, False): 1,
                    ("LSTM", 768, 32, True): 1, ("GRU", 704, 70, True): 2, ("GRU", 1100, 40, False): 3,
                    ("LSTM", 768, 64, True): 1, ("LSTM", 768, 128, True): 2, ("LSTM", 700, 100, True): 2,
                    ("GRU", 576, 45, True): 1, ("LSTM", 544, 72, False): 1,
                    ("LSTM", 1536, 32, False): 1, ("LSTM", 1400, 9, True): 2, ("LSTM", 1280, 70, False): 2,
                    ("LSTM", 1156, 40, False): 1}


def test_one_launch_recurrence_covers_every_hidden_size_its_storage_holds(dev):
  """lr_rnn_pair_supported: any H up to 1152 (multiples of 4), not an enumerated list (round 3 instantiated five sizes):
  32-unit members up to 27 (GRU) / 24 (LSTM) of them, 16-unit members beyond; past 1152 the step kernels run and
  VideoEncoder says so ONCE."""
  import warnings
  from lipreading_amd import _C, encoder as E
  from lipreading_amd.data import default_char2idx
  L = _C.lib()
  for H in (4, 8, 32, 36, 128, 384, 600, 640, 700, 768):
    assert L.lr_rnn_pair_supported(1, 32, 75, 204, H, 2) == 2, H
  for H in (4, 128, 256, 512, 800, 836, 864):
    assert L.lr_rnn_pair_supported(0, 32, 75, 204, H, 2) == 2, H
  for H in (772, 800, 1024, 1100, 1152):
    assert L.lr_rnn_pair_supported(1, 32, 75, 204, H, 1) == 2 and L.lr_rnn_pair_supported(0, 32, 75, 204, max(H, 868), 1) == 2, H
  for H in (1156, 1400, 1536):     # round 6: LSTM past 1152 units on the 24 x 8 grid of 192 CUs (lr_rnn_grid.hip)
    assert L.lr_rnn_pair_supported(1, 32, 75, 204, H, 1) == 2 and L.lr_rnn_pair_supported(1, 128, 31, 204, H, 1) == 2, H
  for mode, H in ((1, 1540), (0, 1156), (0, 1536), (0, 2048), (1, 2048), (1, 30)):
    assert L.lr_rnn_pair_supported(mode, 32, 75, 204, H, 1) == 0, (mode, H)
  enc = E.VideoEncoder(16, 2048, rnn_type='LSTM', bidirectional=False, enable_ctc=True, vocab_size=64,
                       char2idx=default_char2idx()).to(dev)
  x = torch.randn(2, 3, 16, 1, device=dev)
  E._fallback_noted.clear()
  with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    enc(x, torch.tensor([3, 3]), max_len=3)
    enc(x, torch.tensor([3, 3]), max_len=3)
  notes = [str(m.message) for m in w if "one launch per time step" in str(m.message)]
  assert len(notes) == 1 and "LSTM-2048" in notes[0] and "no one-launch kernel for this shape" in notes[0], notes


def test_the_fallback_warning_says_why(dev):
  """lr_rnn_one_launch_status: a layer that runs the step kernels says whether its SHAPE has no one-launch kernel (1),
  the product switched them off after time-outs (2: lr_rnn_one_launch_enable(0), train.RecurrenceWatch), the test hook
  is set (3) or the device is too small (4) — round 4 gave the shape explanation for all of them."""
  import warnings
  from lipreading_amd import _C, encoder as E
  from lipreading_amd.data import default_char2idx
  L = _C.lib()
  assert L.lr_rnn_one_launch_status(1, 32, 75, 204, 768, 2) == 0
  assert L.lr_rnn_one_launch_status(1, 32, 75, 204, 2048, 1) == 1 and L.lr_rnn_one_launch_status(1, 32, 75, 204, 30, 1) == 1
  assert L.lr_rnn_one_launch_status(2, 32, 75, 204, 256, 2) == 1       # tanh RNN
  enc = E.VideoEncoder(16, 64, rnn_type='GRU', bidirectional=True, enable_ctc=True, vocab_size=64,
                       char2idx=default_char2idx()).to(dev)
  x = torch.randn(2, 3, 16, 1, device=dev)

  def notes_of():
    E._fallback_noted.clear()
    with warnings.catch_warnings(record=True) as w:
      warnings.simplefilter("always")
      enc(x, torch.tensor([3, 3]), max_len=3)
    return [str(m.message) for m in w if "one launch per time step" in str(m.message)]
  assert notes_of() == []
  try:
    L.lr_rnn_one_launch_enable(0)
    assert L.lr_rnn_one_launch_status(0, 2, 3, 16, 64, 2) == 2 and L.lr_rnn_pair_supported(0, 2, 3, 16, 64, 2) == 0
    n = notes_of()
    assert len(n) == 1 and "switched off for this process" in n[0] and "no one-launch kernel" not in n[0], n
  finally:
    L.lr_rnn_one_launch_enable(1)
  try:
    L.lr_rnn_debug_disable_cluster(1)
    assert L.lr_rnn_one_launch_status(0, 2, 3, 16, 64, 2) == 3
    n = notes_of()
    assert len(n) == 1 and "test hook" in n[0], n
  finally:
    L.lr_rnn_debug_disable_cluster(0)
  assert L.lr_rnn_one_launch_status(0, 2, 3, 16, 64, 2) == 0 and notes_of() == []


@pytest.mark.parametrize("rnn_type,H,B,T,bi,lens,layers", CLUSTER_CASES)
def test_cluster_recurrence_is_fp32_faithful(dev, rnn_type, H, B, T, bi, lens, layers):
  """LR_RNN_RECUR_SPLIT on the cluster shapes (lr_rnn_cluster.hip: GRU / LSTM, ceil(H / 32) CUs per (direction, 8
  samples); the reference's LSTM-700 / LSTM-512 / GRU-800 / LSTM-768 and sizes that pad up to them): the recurrence
  of a layer pass in ONE launch — W_hh as bf16 hi + lo planes sliced over the cluster; forward: one all-gather of
  the state per step; backward: row-split partial dh, one reduce-scatter per step; self-tagged 4-byte exchange
  words.  Against the exact-fp32 step kernels on the same weights, forward and backward, final-state gradients
  injected, ragged lengths, partial sample groups (B % 8 != 0), hidden sizes that are not a multiple of 32, more
  groups than one launch holds, two stacked layers."""
  from lipreading_amd import _C
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  torch.manual_seed(41)
  enc = VideoEncoder(64, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bi, enable_ctc=True,
                     vocab_size=64, char2idx=default_char2idx()).to(dev)
  g = torch.Generator().manual_seed(42)
  x = torch.randn(B, T, 64, 1, generator=g)
  if lens == "ragged":
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
  else:
    lens = torch.full((B,), T)
  wgt = torch.randn(B, T, 65, generator=g).to(dev)
  valid = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).float().unsqueeze(-1).to(dev)
  assert _C.lib().lr_rnn_pair_supported({"GRU": 0, "LSTM": 1}[rnn_type], B, T, 64, H, 2 if bi else 1) == 2
  if (rnn_type, H, B, bi) in CLUSTER_LAUNCHES:
    assert _C.lib().lr_rnn_pass_launches({"GRU": 0, "LSTM": 1}[rnn_type], B, T, 64, H, 2 if bi else 1) == \
           CLUSTER_LAUNCHES[(rnn_type, H, B, bi)]
  _C.lib().lr_rnn_pair_errors()
  res = {}
  for mode in ("f32", "split"):
    enc.recurrence = mode
    enc.zero_grad()
    lp, hid, fin = enc(x.to(dev), lens, max_len=T)
    fins = fin if isinstance(fin, tuple) else (fin,)
    ((lp * wgt * valid).sum() + hid.pow(2).sum() + sum((3.0 - i) * f.pow(2).sum() for i, f in enumerate(fins))).backward()
    res[mode] = [lp.detach().cpu() * valid.cpu(), hid.detach().cpu()] + [f.detach().cpu() for f in fins] + \
                [p.grad.cpu().clone() for p in enc.parameters()]
  enc.recurrence = "auto"
  assert _C.lib().lr_rnn_pair_errors() == 0
  if T > 1:
    assert float((res["f32"][1] - res["split"][1]).abs().max()) > 0
  worst = 0.0
  for a, b in zip(res["f32"], res["split"]):
    worst = max(worst, float((a - b).norm()) / max(1e-6, float(a.norm())))
    assert float((a - b).norm()) / max(1e-6, float(a.norm())) < 2e-5
  print("%s-%d cluster vs f32 recurrence: worst relative norm difference %.3g" % (rnn_type, H, worst))
  assert float((res["split"][1] * (1 - valid.cpu())).abs().max()) == 0.0


@pytest.mark.parametrize("rnn_type,H", [("GRU", 256), ("LSTM", 512), ("LSTM", 1536)])   # (1536: the 192-CU grid, lr_rnn_grid.hip)
def test_a_recurrence_time_out_skips_the_step_and_is_reported(dev, rnn_type, H, capsys):
  """A member of a cluster that never shows up (here: the test hook makes member 1 return at once) leaves
  its partners waiting; their waits are bounded, what they produce is garbage, and the reference's contract for a
  batch it cannot use is assert / None => skip (src/train/train_better_model.py:46-50).  Here the fault travels on
  the device: lr_ctc_reduce reports status != 0 with loss 0, lr_adam_step leaves the weights alone, train() counts
  the batch as skipped and reports the time-outs — no host round trip inside the step."""
  from lipreading_amd import _C, train as T_
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.optim import FlatParameters, FusedAdam
  L = _C.lib()
  torch.manual_seed(3)
  c2i = default_char2idx()
  enc = VideoEncoder(204, H, rnn_type=rnn_type, bidirectional=True, enable_ctc=True, vocab_size=64, char2idx=c2i).to(dev)
  opt = FusedAdam(FlatParameters(enc), lr=1e-3)
  g = torch.Generator().manual_seed(4)
  B, Tn = 8, 12
  frames = torch.randn(B, Tn, 68, 3, generator=g)
  lens = torch.full((B,), Tn)
  chars = torch.zeros(B, 7, dtype=torch.long)
  chars[:, 0], chars[:, 1:6], chars[:, 6] = c2i['<BOS>'], torch.randint(4, 64, (B, 5), generator=g), c2i['<EOS>']
  char_lens = torch.full((B,), 7)
  batch = [(frames, lens, chars, char_lens)]
  L.lr_rnn_pair_errors()
  # a healthy step first
  before = opt.flat.data.clone()
  loss, status = T_.ctc_step(enc, opt, frames.to(dev), lens.to(dev), chars.to(dev), char_lens.to(dev), grad_norm=50)
  assert int(status) == 0 and float(loss) > 0 and not torch.equal(before, opt.flat.data)
  assert L.lr_rnn_pair_errors() == 0
  try:
    L.lr_rnn_debug_drop_member(1)
    before = opt.flat.data.clone()
    steps_before = int(opt.step_count[0])
    loss, status = T_.ctc_step(enc, opt, frames.to(dev), lens.to(dev), chars.to(dev), char_lens.to(dev), grad_norm=50)
    torch.cuda.synchronize()
    assert int(status) != 0 and float(loss) == 0.0           # reported like a batch the reference skips
    assert torch.equal(before, opt.flat.data)                # nothing was updated
    assert int(opt.step_count[0]) == steps_before and opt.skipped_steps() >= 1
    # the product loop: the epoch runs through, counts the batch and names the cause
    T_.train(enc, None, batch, opt, dev, c2i, grad_norm=50)
    assert T_.last_epoch_stats["skipped"] == 1 and T_.last_epoch_stats["recurrence_faults"] >= 1
    assert "recurrence time-outs" in capsys.readouterr().out
  finally:
    L.lr_rnn_debug_drop_member(-1)
  # and the next healthy step is a normal step again (the fault word was rolled by lr_step_begin)
  try:
    L.lr_rnn_pair_errors()
    before = opt.flat.data.clone()
    loss, status = T_.ctc_step(enc, opt, frames.to(dev), lens.to(dev), chars.to(dev), char_lens.to(dev), grad_norm=50)
    assert int(status) == 0 and float(loss) > 0 and not torch.equal(before, opt.flat.data)
    assert torch.isfinite(opt.flat.data).all() and L.lr_rnn_pair_errors() == 0
  finally:
    L.lr_rnn_debug_disable_cluster(0)


def test_cluster_recurrence_survives_a_foreign_kernel_that_holds_compute_units(dev):
  """What a data-parallel run does to the one-launch recurrence: while the gradient all-reduce of one bucket runs (an
  RCCL ring kernel on a side stream, holding compute units of its own), the next layer's cluster kernels — 8 x 24 = 192
  workgroups that must ALL be resident, one per compute unit — are launched.  A 1-rank all-reduce is a memcpy and gloo
  runs on the host, so the stand-in is lr_debug_busy: 96 workgroups that each hold a whole CU's LDS for 3 ms, on a
  side stream, across the launches of rnnc_fwd_kernel<4,24> and rnnc_bwd_kernel<4,24>.  192 + 96 > 256: members have
  to wait for compute units; their waits are bounded at a few tenths of a second, the foreign kernel is not: no
  time-out, and the same numbers as the undisturbed run."""
  from lipreading_amd import _C
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  L = _C.lib()
  torch.manual_seed(51)
  enc = VideoEncoder(64, 768, rnn_type='LSTM', bidirectional=True, enable_ctc=True, vocab_size=64,
                     char2idx=default_char2idx()).to(dev)
  g = torch.Generator().manual_seed(52)
  B, T = 32, 30
  x = torch.randn(B, T, 64, 1, generator=g).to(dev)
  lens = torch.full((B,), T)
  wgt = torch.randn(B, T, 65, generator=g).to(dev)
  assert L.lr_rnn_pair_supported(1, B, T, 64, 768, 2) == 2
  L.lr_rnn_pair_errors()
  side = torch.cuda.Stream()
  res = {}
  for busy in (False, True, True):
    enc.zero_grad()
    torch.cuda.synchronize()
    if busy:
      with torch.cuda.stream(side):
        _C.check(L.lr_debug_busy(96, 160 * 1024, 3000, _C.stream_handle()), "lr_debug_busy")
    lp, hid, _ = enc(x, lens, max_len=T)
    if busy:   # ... and again across the backward launch
      with torch.cuda.stream(side):
        _C.check(L.lr_debug_busy(96, 160 * 1024, 3000, _C.stream_handle()), "lr_debug_busy")
    ((lp * wgt).sum() + hid.pow(2).sum()).backward()
    torch.cuda.synchronize()
    res[busy] = [lp.detach().clone(), hid.detach().clone()] + [p.grad.clone() for p in enc.parameters()]
  assert L.lr_rnn_pair_errors() == 0
  for a, b in zip(res[False], res[True]):
    assert torch.equal(a, b)


def test_inter_layer_dropout_is_a_philox_mask(dev):
  """rnn_dropout > 0 (better_model.py:47-49 hands it to nn.GRU / nn.LSTM): between stacked layers, training mode only —
  lr_dropout_forward.  torch's mask comes from torch's generator stream, so the mask is tested AS A MASK: values in
  {0, 1 / (1 - p)}, the dropped fraction, determinism in the seed, independence of the launch geometry, the backward
  the same multiply; eval mode and the last layer untouched."""
  import torch.nn.functional as F
  from lipreading_amd import _C, encoder as E
  from lipreading_amd.data import default_char2idx
  L = _C.lib()
  n, p = 1000003, 0.3
  x = torch.randn(n, device=dev)
  y, m = torch.empty_like(x), torch.empty_like(x)
  _C.check(L.lr_dropout_forward(x.data_ptr(), y.data_ptr(), m.data_ptr(), n, p, 12345, _C.stream_handle()), "dropout")
  keep = 1.0 / (1.0 - p)
  assert bool(((m == 0) | ((m - keep).abs() < 1e-6)).all()) and torch.equal(y, x * m)
  frac = float((m == 0).float().mean())
  assert abs(frac - p) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-4, frac
  m2 = torch.empty_like(x)
  _C.check(L.lr_dropout_forward(x.data_ptr(), y.data_ptr(), m2.data_ptr(), n, p, 12345, _C.stream_handle()), "dropout")
  assert torch.equal(m, m2)
  _C.check(L.lr_dropout_forward(x.data_ptr(), y.data_ptr(), m2.data_ptr(), n, p, 12346, _C.stream_handle()), "dropout")
  assert not torch.equal(m, m2) and abs(float(((m == 0) & (m2 == 0)).float().mean()) - p * p) < 0.005   # independent draws
  # a prefix of the same seed is the same mask (counter = element index / 4, not the launch)
  _C.check(L.lr_dropout_forward(x.data_ptr(), y.data_ptr(), m2.data_ptr(), 4099, p, 12345, _C.stream_handle()), "dropout")
  assert torch.equal(m[:4099], m2[:4099])
  # in the encoder: training mode changes the second layer's input, eval mode does not; gradients flow through the mask
  torch.manual_seed(7)
  enc = E.VideoEncoder(64, 32, rnn_type='GRU', num_layers=2, bidirectional=True, rnn_dropout=0.5, enable_ctc=True,
                       vocab_size=64, char2idx=default_char2idx()).to(dev)
  xx = torch.randn(4, 9, 64, 1, device=dev)
  lens = torch.full((4,), 9)
  enc.eval()
  a = enc(xx, lens, max_len=9)[1]
  b = enc(xx, lens, max_len=9)[1]
  assert torch.equal(a, b)
  enc.train()
  c = enc(xx, lens, max_len=9)[1]
  d = enc(xx, lens, max_len=9)[1]
  assert not torch.equal(c, d) and not torch.equal(a, c)
  c.pow(2).sum().backward()
  assert all(p_.grad is not None and torch.isfinite(p_.grad).all() and float(p_.grad.abs().max()) > 0
             for p_ in enc.rnn.parameters())
  # _DropoutFunction's backward is the mask multiply
  z = torch.randn(5, 7, device=dev, requires_grad=True)
  out = E._DropoutFunction.apply(z, 0.25, 99)
  out.backward(torch.ones_like(out))
  assert bool(((z.grad == 0) | ((z.grad - 1 / 0.75).abs() < 1e-6)).all()) and torch.equal(z.grad == 0, out.detach() == 0)


def test_final_states_in_one_launch_equal_the_permuted_copies(dev):
  """_cat_directions (better_model.py:98-112): (D,B,H) -> (B, D*H) for h and c of a layer in one launch each way."""
  from lipreading_amd import encoder as E
  g = torch.Generator().manual_seed(3)
  for D, B, H in ((2, 5, 12), (1, 3, 8), (2, 32, 700)):
    h = torch.randn(D, B, H, generator=g).to(dev).requires_grad_()
    c = torch.randn(D, B, H, generator=g).to(dev).requires_grad_()
    oh, oc = E._CatDirectionsFunction.apply(h, c)
    assert torch.equal(oh, h.detach().permute(1, 0, 2).reshape(B, D * H)) and torch.equal(oc, c.detach().permute(1, 0, 2).reshape(B, D * H))
    wh, wc = torch.randn(B, D * H, generator=g).to(dev), torch.randn(B, D * H, generator=g).to(dev)
    ((oh * wh).sum() + (oc * wc).sum()).backward()
    assert torch.equal(h.grad, wh.reshape(B, D, H).permute(1, 0, 2)) and torch.equal(c.grad, wc.reshape(B, D, H).permute(1, 0, 2))
    h2 = h.detach().clone().requires_grad_()
    o2 = E._CatDirectionsFunction.apply(h2, None)
    (o2 * wh).sum().backward()
    assert torch.equal(o2, oh) and torch.equal(h2.grad, h.grad)
