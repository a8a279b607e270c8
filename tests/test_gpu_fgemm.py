"""GPU checks of lr_fgemm (lr_fgemm.hip): products that take their operands as they lie in memory — fp32 (split into bf16
hi + lo on the way into LDS) or bf16 — in the three forms of a Linear's forward / data gradient / weight gradient, against
an fp64 evaluation of the same formula.  BUILD-DEFINED (the transformer stage of BASELINE configs[4]): no reference
symbol; the tolerances are this repo's: X3 (three bf16 products) 3e-5 in relative norm (lr_xgemm's), F32 (exact fp32
matrix-core products) 2e-6."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

X3, F32 = 0, 1
NT, NN, TN = 0, 1, 2
RELU, C_BF16 = 1, 2


@pytest.fixture(scope="module")
def dev():
  assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
  return torch.device("cuda:0")


def run_job(prec, form, A, Bm, M, N, K, bias=None, addend=None, add_period=0, mask=None, relu=False, c_bf16=False,
            colsum=None, beta=0.0, alpha=1.0, C0=None, splits=1, a_bf16=False, b_bf16=False, b_shift=0, b_period=0):
  from lipreading_amd import _C
  dev = A.device
  C = C0.clone() if C0 is not None else torch.full((M, N), float("nan"), dtype=torch.bfloat16 if c_bf16 else torch.float32,
                                                   device=dev)
  slabs = torch.full((max(1, splits) * (M * N + M),), float("nan"), dtype=torch.float32, device=dev) if splits > 1 else None
  j = _C.FgemmJob()
  j.A, j.B, j.C = A.data_ptr(), Bm.data_ptr(), C.data_ptr()
  j.bias, j.addend, j.mask = _C.ptr(bias), _C.ptr(addend), _C.ptr(mask)
  j.colsum, j.slabs = _C.ptr(colsum), _C.ptr(slabs)
  j.M, j.N, j.K = M, N, K
  j.lda, j.ldb, j.ldc = A.stride(0), Bm.stride(0), C.stride(0)
  j.ldadd = addend.stride(0) if addend is not None else 0
  j.add_period = add_period
  j.ldmask = mask.stride(0) if mask is not None else 0
  j.flags = (RELU if relu else 0) | (C_BF16 if c_bf16 else 0)
  j.splits = splits
  j.alpha, j.beta = alpha, beta
  j.b_shift, j.b_period = b_shift, b_period
  _C.check(_C.lib().lr_fgemm(prec, form, int(a_bf16), int(b_bf16), ctypes.byref(j), 1, _C.stream_handle()), "lr_fgemm")
  torch.cuda.synchronize()
  return C


def operands(form, M, N, K, g, dev, pad_a=0, pad_b=0):
  """A, B as they lie in memory for the form (with optional padding columns: leading dimension > width), and the fp64
  matrices op(A) [M][K], op(B) [K][N]."""
  # (sliced ON the device: a column slice keeps its storage, so the leading dimension is the padded width)
  if form == TN:
    fa = torch.randn(K, M + pad_a, generator=g)
    A, opA = fa.to(dev)[:, :M], fa[:, :M].double().t()
  else:
    fa = torch.randn(M, K + pad_a, generator=g)
    A, opA = fa.to(dev)[:, :K], fa[:, :K].double()
  if form == NT:
    fb = torch.randn(N, K + pad_b, generator=g)
    Bm, opB = fb.to(dev)[:, :K], fb[:, :K].double().t()
  else:
    fb = torch.randn(K, N + pad_b, generator=g)
    Bm, opB = fb.to(dev)[:, :N], fb[:, :N].double()
  return A, Bm, opA, opB


SHAPES = [(80, 192, 64), (300, 130, 100), (128, 128, 32), (2400, 256, 256), (257, 65, 31), (5, 7, 3), (256, 1024, 2400)]


@pytest.mark.parametrize("prec", [X3, F32])
@pytest.mark.parametrize("form", [NT, NN, TN])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_products_match_fp64(dev, prec, form, M, N, K):
  g = torch.Generator().manual_seed(M * 7 + N * 3 + K + form)
  A, Bm, opA, opB = operands(form, M, N, K, g, dev)
  want = opA @ opB
  got = run_job(prec, form, A, Bm, M, N, K).cpu().double()
  tol = 3e-5 if prec == X3 else 2e-6
  assert float((got - want).norm()) <= tol * float(want.norm()), (prec, form, M, N, K)
  assert float((got - want).abs().max()) <= 10 * tol * float(want.abs().max())


@pytest.mark.parametrize("prec", [X3, F32])
@pytest.mark.parametrize("form", [NT, NN, TN])
def test_unaligned_leading_dimensions_take_the_scalar_loads(dev, prec, form):
  """leading dimensions that are not multiples of four elements (and so rows that are not 16-byte aligned)"""
  M, N, K = 150, 70, 45
  g = torch.Generator().manual_seed(99 + form)
  A, Bm, opA, opB = operands(form, M, N, K, g, dev, pad_a=4, pad_b=4)
  assert A.stride(0) % 4 != 0 and Bm.stride(0) % 4 != 0
  want = opA @ opB
  got = run_job(prec, form, A, Bm, M, N, K).cpu().double()
  assert float((got - want).norm()) <= (3e-5 if prec == X3 else 2e-6) * float(want.norm())


@pytest.mark.parametrize("prec", [X3, F32])
def test_epilogues(dev, prec):
  """bias, periodic addend (positional table), residual addend, ReLU, mask, alpha / beta, bf16 output"""
  M, N, K, T = 150, 136, 96, 25
  g = torch.Generator().manual_seed(5)
  A, Bm, opA, opB = operands(NT, M, N, K, g, dev)
  acc = opA @ opB
  bias = torch.randn(N, generator=g)
  pe = torch.randn(T, N, generator=g)
  res = torch.randn(M, N, generator=g)
  msk = torch.randn(M, N, generator=g)
  C0 = torch.randn(M, N, generator=g)
  tol = 3e-5 if prec == X3 else 2e-6

  def close(got, want, t=tol):
    assert float((got.cpu().double() - want).norm()) <= t * float(want.norm())
  close(run_job(prec, NT, A, Bm, M, N, K, bias=bias.to(dev)), acc + bias.double())
  rows = torch.arange(M) % T
  close(run_job(prec, NT, A, Bm, M, N, K, bias=bias.to(dev), addend=pe.to(dev), add_period=T), acc + bias.double() + pe.double()[rows])
  close(run_job(prec, NT, A, Bm, M, N, K, addend=res.to(dev), add_period=M), acc + res.double())
  close(run_job(prec, NT, A, Bm, M, N, K, bias=bias.to(dev), relu=True), (acc + bias.double()).clamp(min=0))
  close(run_job(prec, NT, A, Bm, M, N, K, mask=msk.to(dev)), torch.where(msk > 0, acc, torch.zeros_like(acc)))
  close(run_job(prec, NT, A, Bm, M, N, K, alpha=0.5, beta=2.0, C0=C0.to(dev)), 0.5 * acc + 2.0 * C0.double())
  got = run_job(prec, NT, A, Bm, M, N, K, bias=bias.to(dev), c_bf16=True)
  assert got.dtype == torch.bfloat16
  close(got.float(), acc + bias.double(), 4e-3)   # bf16 rounding of the output


@pytest.mark.parametrize("prec", [X3, F32])
def test_split_k_goes_through_slabs(dev, prec):
  from lipreading_amd import _C
  M, N, K, T = 200, 96, 1000, 40
  g = torch.Generator().manual_seed(8)
  A, Bm, opA, opB = operands(NT, M, N, K, g, dev)
  bias = torch.randn(N, generator=g)
  pe = torch.randn(T, N, generator=g)
  want = opA @ opB + bias.double() + pe.double()[torch.arange(M) % T]
  for splits in (2, 5, _C.lib().lr_fgemm_splits(M, N, K)):
    got = run_job(prec, NT, A, Bm, M, N, K, bias=bias.to(dev), addend=pe.to(dev), add_period=T, splits=splits)
    assert float((got.cpu().double() - want).norm()) <= (3e-5 if prec == X3 else 2e-6) * float(want.norm()), splits
  assert _C.lib().lr_fgemm_splits(2400, 256, 3456) > 1       # the K = 3456 input projection: 38 tiles


@pytest.mark.parametrize("prec", [X3, F32])
def test_weight_gradient_form_emits_the_bias_gradient(dev, prec):
  """TN: dW = dy^T x with colsum = column sums of dy (+ beta * what was there)"""
  R, No, Ki = 333, 200, 72
  g = torch.Generator().manual_seed(12)
  dy = torch.randn(R, No, generator=g)
  x = torch.randn(R, Ki, generator=g)
  want, wsum = dy.double().t() @ x.double(), dy.double().sum(0)
  tol = 3e-5 if prec == X3 else 2e-6
  cs = torch.full((No,), float("nan"), device=dev)
  got = run_job(prec, TN, dy.to(dev), x.to(dev), No, Ki, R, colsum=cs)
  assert float((got.cpu().double() - want).norm()) <= tol * float(want.norm())
  assert float((cs.cpu().double() - wsum).norm()) <= 2e-6 * float(wsum.norm())
  # accumulate (beta = 1): gradients added to what the flat buffer holds
  C0, c0 = torch.randn(No, Ki, generator=g), torch.randn(No, generator=g)
  cs = c0.to(dev)
  got = run_job(prec, TN, dy.to(dev), x.to(dev), No, Ki, R, colsum=cs, beta=1.0, C0=C0.to(dev))
  assert float((got.cpu().double() - (want + C0.double())).norm()) <= tol * float(want.norm())
  assert float((cs.cpu().double() - (wsum + c0.double())).norm()) <= 2e-6 * float(wsum.norm())


@pytest.mark.parametrize("prec", [X3, F32])
def test_bf16_stored_operands(dev, prec):
  """the conv frontend's features are bf16 tensors: A of the input projection (NT), B of its weight gradient (TN)"""
  R, I, Dm = 300, 864, 128
  g = torch.Generator().manual_seed(21)
  feats = torch.randn(R, I, generator=g).to(torch.bfloat16)
  W = torch.randn(Dm, I, generator=g)
  dh = torch.randn(R, Dm, generator=g)
  tol = 3e-5 if prec == X3 else 2e-6
  want = feats.double() @ W.double().t()
  got = run_job(prec, NT, feats.to(dev), W.to(dev), R, Dm, I, a_bf16=True)
  assert float((got.cpu().double() - want).norm()) <= tol * float(want.norm())
  want = dh.double().t() @ feats.double()
  got = run_job(prec, TN, dh.to(dev), feats.to(dev), Dm, I, R, b_bf16=True)
  assert float((got.cpu().double() - want).norm()) <= tol * float(want.norm())


def test_several_products_share_a_launch(dev):
  """the weight gradients of a whole stack are jobs of ONE launch"""
  from lipreading_amd import _C
  g = torch.Generator().manual_seed(33)
  R = 260
  shapes = [(192, 64), (64, 64), (128, 64), (64, 128), (40, 12)] * 3      # 15 jobs
  jobs = (_C.FgemmJob * len(shapes))()
  keep, want = [], []
  for q, (No, Ki) in enumerate(shapes):
    dy, x = torch.randn(R, No, generator=g).to(dev), torch.randn(R, Ki, generator=g).to(dev)
    C, cs = torch.empty(No, Ki, device=dev), torch.empty(No, device=dev)
    keep.append((dy, x, C, cs))
    want.append((dy.double().t() @ x.double(), dy.double().sum(0)))
    j = jobs[q]
    j.A, j.B, j.C, j.colsum = dy.data_ptr(), x.data_ptr(), C.data_ptr(), cs.data_ptr()
    j.bias = j.addend = j.mask = j.slabs = None
    j.M, j.N, j.K, j.lda, j.ldb, j.ldc = No, Ki, R, No, Ki, Ki
    j.ldadd = j.add_period = j.ldmask = j.flags = 0
    j.splits, j.alpha, j.beta = 1, 1.0, 0.0
    j.b_shift = j.b_period = 0
  _C.check(_C.lib().lr_fgemm(X3, TN, 0, 0, jobs, len(shapes), _C.stream_handle()), "lr_fgemm")
  torch.cuda.synchronize()
  for (dy, x, C, cs), (w, ws) in zip(keep, want):
    assert float((C.double() - w).norm()) <= 3e-5 * float(w.norm())
    assert float((cs.double() - ws).norm()) <= 2e-6 * float(ws.norm())


@pytest.mark.parametrize("prec", [X3, F32])
@pytest.mark.parametrize("shift", [-1, 1])
@pytest.mark.parametrize("Bn,T,H,G4", [(32, 75, 256, 512), (5, 7, 36, 72), (3, 1, 64, 128), (2, 40, 30, 90)])
def test_recurrent_weight_gradient_reads_the_neighbouring_time_step(dev, prec, shift, Bn, T, H, G4):
  """TN with b_shift / b_period: dW_hh = dG^T . h_prev straight from y [B*T][H] — h_prev[b,t] = y[b,t-1] (shift -1: the
  forward direction) or y[b,t+1] (+1: the reverse one), zero across the ends of a sequence (lr_rnn.hip's weight half;
  rounds 1-4 packed a shifted, transposed copy of y for it)."""
  g = torch.Generator().manual_seed(Bn * 100 + T + H + shift)
  R = Bn * T
  y = torch.randn(Bn, T, H + 4, generator=g)          # (a direction's columns of a wider [B][T][D*H] tensor)
  dG = torch.randn(R, G4, generator=g)
  hp = torch.zeros(Bn, T, H)
  if shift < 0:
    hp[:, 1:] = y[:, :-1, :H]
  else:
    hp[:, :-1] = y[:, 1:, :H]
  want = dG.double().t() @ hp.reshape(R, H).double()
  yd = y.to(dev).reshape(R, H + 4)[:, :H]
  got = run_job(prec, TN, dG.to(dev), yd, G4, H, R, b_shift=shift, b_period=T)
  scale = max(float(want.norm()), 1e-6)
  assert float((got.cpu().double() - want).norm()) <= (3e-5 if prec == X3 else 2e-6) * scale + (1e-30 if T > 1 else 0)


@pytest.mark.parametrize("prec", [X3, F32])
@pytest.mark.parametrize("splits", [2, 7])
def test_split_weight_gradient_keeps_its_bias_gradient_and_row_shift(dev, prec, splits):
  """a regime-R layer's weight half (lr_rnn.hip, round 5): K = B * T cut into ranges, the column sums of A reduced with
  the slabs, the recurrent product's neighbouring-step read across the range boundaries"""
  Bn, T, H, G4 = 32, 75, 64, 256
  g = torch.Generator().manual_seed(77 + splits)
  R = Bn * T
  y = torch.randn(Bn, T, H, generator=g)
  dG = torch.randn(R, G4, generator=g)
  hp = torch.zeros(Bn, T, H)
  hp[:, 1:] = y[:, :-1]
  want, wsum = dG.double().t() @ hp.reshape(R, H).double(), dG.double().sum(0)
  C0, c0 = torch.randn(G4, H, generator=g), torch.randn(G4, generator=g)
  tol = 3e-5 if prec == X3 else 2e-6
  for beta in (0.0, 1.0):
    cs = c0.to(dev).clone()
    got = run_job(prec, TN, dG.to(dev), y.to(dev).reshape(R, H), G4, H, R, colsum=cs, splits=splits, b_shift=-1, b_period=T,
                  beta=beta, C0=C0.to(dev))
    assert float((got.cpu().double() - (want + beta * C0.double())).norm()) <= tol * float(want.norm()), beta
    assert float((cs.cpu().double() - (wsum + beta * c0.double())).norm()) <= 2e-6 * float(wsum.norm()), beta


def test_split_and_unsplit_products_share_a_launch_and_one_combine(dev):
  from lipreading_amd import _C
  g = torch.Generator().manual_seed(41)
  R = 1200
  shapes = [(192, 64, 4), (64, 64, 1), (128, 72, 3), (40, 12, 4)]
  jobs = (_C.FgemmJob * len(shapes))()
  keep, want = [], []
  for q, (No, Ki, sp) in enumerate(shapes):
    dy, x = torch.randn(R, No, generator=g).to(dev), torch.randn(R, Ki, generator=g).to(dev)
    C, cs = torch.empty(No, Ki, device=dev), torch.empty(No, device=dev)
    slabs = torch.full((_C.lib().lr_fgemm_slab_floats(No, Ki, sp),), float("nan"), device=dev) if sp > 1 else None
    keep.append((dy, x, C, cs, slabs))
    want.append((dy.double().t() @ x.double(), dy.double().sum(0)))
    j = jobs[q]
    j.A, j.B, j.C, j.colsum = dy.data_ptr(), x.data_ptr(), C.data_ptr(), cs.data_ptr()
    j.bias = j.addend = j.mask = None
    j.slabs = _C.ptr(slabs)
    j.M, j.N, j.K, j.lda, j.ldb, j.ldc = No, Ki, R, No, Ki, Ki
    j.ldadd = j.add_period = j.ldmask = j.flags = 0
    j.splits, j.alpha, j.beta = sp, 1.0, 0.0
    j.b_shift = j.b_period = 0
  _C.check(_C.lib().lr_fgemm(X3, TN, 0, 0, jobs, len(shapes), _C.stream_handle()), "lr_fgemm")
  torch.cuda.synchronize()
  for (dy, x, C, cs, _), (w, ws) in zip(keep, want):
    assert float((C.double() - w).norm()) <= 3e-5 * float(w.norm())
    assert float((cs.double() - ws).norm()) <= 2e-6 * float(ws.norm())
