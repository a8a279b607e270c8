"""GPU: the build-defined 3-D conv frontend (A8) against its torch-CPU oracle.  There is no
reference arithmetic for this stage; tolerances are bf16-storage tolerances."""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
  assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
  return torch.device("cuda:0")


def rel_err(a, b):
  return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


def test_single_conv_layer_and_dgrad_against_torch(dev):
  """lr_conv3d_forward on each layer geometry, and its data-gradient use, against F.conv3d."""
  from lipreading_amd import _C
  L = _C.lib()
  g = torch.Generator().manual_seed(0)
  for cin, cout, (kt, kh, kw), stride, (pt, ph, pw), hw in ((3, 32, (3, 5, 5), 2, (1, 2, 2), 20),
                                                             (32, 64, (3, 5, 5), 1, (1, 2, 2), 10),
                                                             (64, 96, (3, 3, 3), 1, (1, 1, 1), 6)):
    B, T = 2, 4
    cin_p = (cin + 3) // 4 * 4
    x = torch.randn(B, T, hw, hw, cin_p, generator=g)
    x[..., cin:] = 0
    w = torch.randn(cout, cin, kt, kh, kw, generator=g) / (cin * kt * kh * kw) ** 0.5
    b = torch.randn(cout, generator=g)
    xb, wb = x.bfloat16(), w.bfloat16()
    ref = torch.nn.functional.conv3d(xb.float()[..., :cin].permute(0, 4, 1, 2, 3), wb.float(), b,
                                     stride=(1, stride, stride), padding=(pt, ph, pw))
    ho = ref.shape[-1]
    xd, wd32, bd = xb.to(dev).contiguous(), w.to(dev), b.to(dev)
    wp = torch.empty((cout, kt * kh * kw, cin_p), dtype=torch.bfloat16, device=dev)
    y = torch.empty((B * T, ho, ho, cout), dtype=torch.bfloat16, device=dev)
    st = _C.stream_handle()
    _C.check(L.lr_conv3d_pack_weights(wd32.data_ptr(), wp.data_ptr(), cout, cin, cin_p, kt, kh, kw, 0, st))
    _C.check(L.lr_conv3d_forward(xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), y.data_ptr(), B, T, hw, hw,
                                 cin_p, cout, kt, kh, kw, stride, pt, ph, pw, 0, st))
    got = y.float().cpu().reshape(B, T, ho, ho, cout).permute(0, 4, 1, 2, 3)
    assert rel_err(got.numpy(), ref.numpy()) < 1e-2, (cin, cout)
    if stride == 1:   # data gradient through the same kernel
      dz = torch.randn(B, T, ho, ho, cout, generator=g).bfloat16()
      xr = xb.float()[..., :cin].permute(0, 4, 1, 2, 3).requires_grad_(True)
      out = torch.nn.functional.conv3d(xr, wb.float(), None, stride=1, padding=(pt, ph, pw))
      out.backward(dz.float().permute(0, 4, 1, 2, 3))
      wdg = torch.empty((cin, kt * kh * kw, cout), dtype=torch.bfloat16, device=dev)
      dx = torch.empty((B * T, hw, hw, cin), dtype=torch.bfloat16, device=dev)
      dzd = dz.to(dev).contiguous()
      _C.check(L.lr_conv3d_pack_weights(wd32.data_ptr(), wdg.data_ptr(), cout, cin, cin_p, kt, kh, kw, 1, st))
      _C.check(L.lr_conv3d_forward(dzd.data_ptr(), wdg.data_ptr(), None, dx.data_ptr(), B, T, ho, ho, cout,
                                   cin, kt, kh, kw, 1, pt, ph, pw, 0, st))
      got = dx.float().cpu().reshape(B, T, hw, hw, cin).permute(0, 4, 1, 2, 3)
      assert rel_err(got.numpy(), xr.grad.numpy()) < 1.5e-2, (cin, cout)


@pytest.mark.parametrize("H,W", [(19, 21), (20, 22), (37, 33)])
@pytest.mark.parametrize("dtype", ["u8", "bf16"])
def test_first_layer_forward_at_odd_and_ragged_sizes_against_torch(dev, dtype, H, W):
  """The first layer's patch kernel at frame sizes the frontend never produces: odd widths (pixel pairs that straddle the
  frame's edge: the kernel's per-pixel loads instead of pair loads), heights and widths whose 16 x 16 output tiles hang
  over the edge, several tiles per frame (37 x 33 -> 19 x 17 outputs), from the raw uint8 clip (flags & 8) and from the
  4-channel bf16 copy, without the pooling epilogue; against F.conv3d on the same bf16 values."""
  from lipreading_amd import _C
  L = _C.lib()
  st = _C.stream_handle()
  g = torch.Generator().manual_seed(H * 100 + W)
  B, T, cin, cin_p, cout, kt, kh, kw = 1, 4, 3, 4, 32, 3, 5, 5
  clips = torch.randint(0, 256, (B * T, 3, H, W), generator=g, dtype=torch.uint8)
  w = torch.randn(cout, cin, kt, kh, kw, generator=g) / (cin * kt * kh * kw) ** 0.5
  b = torch.randn(cout, generator=g)
  xb = (clips.float() * (1.0 / 255.0)).bfloat16()                       # [F][3][H][W], what the kernel makes of a byte
  ref = torch.nn.functional.conv3d(xb.float().reshape(B, T, 3, H, W).permute(0, 2, 1, 3, 4), w.bfloat16().float(), b,
                                   stride=(1, 2, 2), padding=(1, 2, 2))
  ho, wo = ref.shape[-2], ref.shape[-1]
  wp = torch.empty((cout, kt * kh * kw, cin_p), dtype=torch.bfloat16, device=dev)
  _C.check(L.lr_conv3d_pack_weights(w.to(dev).data_ptr(), wp.data_ptr(), cout, cin, cin_p, kt, kh, kw, 0, st))
  y = torch.full((B * T, ho, wo, cout), float("nan"), dtype=torch.bfloat16, device=dev)
  if dtype == "u8":
    x = clips.to(dev)
    flags = 8
  else:
    x = torch.zeros((B * T, H, W, cin_p), dtype=torch.bfloat16)
    x[..., :3] = xb.permute(0, 2, 3, 1)
    x = x.to(dev)
    flags = 0
  _C.check(L.lr_conv3d_forward(x.data_ptr(), wp.data_ptr(), b.to(dev).data_ptr(), y.data_ptr(), B, T, H, W, cin_p, cout,
                               kt, kh, kw, 2, 1, 2, 2, flags, st))
  got = y.float().cpu().reshape(B, T, ho, wo, cout).permute(0, 4, 1, 2, 3)
  assert torch.isfinite(got).all()
  assert rel_err(got.numpy(), ref.numpy()) < 1e-2, (dtype, H, W)


@pytest.mark.parametrize("dtype", ["u8", "f32"])
def test_frontend_forward_backward_matches_oracle(dev, dtype):
  from lipreading_amd.frontend import ConvFrontend3D, feature_dim
  torch.manual_seed(3)
  fe = ConvFrontend3D()
  params_cpu = [p.detach().clone().requires_grad_(True) for p in fe.parameters_in_order()]
  fe = fe.to(dev)
  g = torch.Generator().manual_seed(4)
  B, T, H = 2, 6, 32
  clips = torch.randint(0, 256, (B, T, 3, H, H), generator=g, dtype=torch.uint8)
  if dtype == "f32":
    clips = clips.float() / 255.0
  wgt = torch.randn(B, T, feature_dim(H, H), generator=g)
  ref = O.conv_frontend(clips, params_cpu)
  (ref * wgt).sum().backward()
  out = fe(clips.to(dev))
  assert out.shape == (B, T, feature_dim(H, H)) and out.dtype == torch.float32
  (out * wgt.to(dev)).sum().backward()
  assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 2e-2
  for p, q in zip(fe.parameters_in_order(), params_cpu):
    assert rel_err(p.grad.cpu().numpy(), q.grad.numpy()) < 4e-2, tuple(p.shape)


def test_pixel_lipreader_trains_end_to_end(dev):
  """pixels -> conv frontend -> BiGRU -> CTC -> backward -> fused Adam: loss goes down."""
  from lipreading_amd.ctc import ctc_loss_with_status
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader, feature_dim
  from lipreading_amd.optim import FlatParameters, FusedAdam
  torch.manual_seed(0)
  H = 32
  enc = VideoEncoder(feature_dim(H, H), 32, rnn_type='GRU', bidirectional=True, enable_ctc=True,
                     vocab_size=64, char2idx=default_char2idx())
  model = PixelLipReader(enc, ConvFrontend3D()).to(dev).train()
  opt = FusedAdam(FlatParameters(model), lr=2e-3)
  g = torch.Generator().manual_seed(1)
  clips = torch.randint(0, 256, (4, 12, 3, H, H), generator=g, dtype=torch.uint8).to(dev)
  lens = torch.full((4,), 12, device=dev)
  labels = torch.randint(4, 64, (4, 5), generator=g).to(dev)
  ll = torch.full((4,), 5, device=dev)
  losses = []
  for _ in range(25):
    opt.zero_grad()
    lp, _, _ = model(clips, lens, max_len=12)
    loss, status, _ = ctc_loss_with_status(lp, labels, lens, ll, 'mean')
    loss.backward()
    opt.step(grad_norm=50, skip=status)
    losses.append(float(loss))
  assert np.isfinite(losses).all() and losses[-1] < 0.8 * losses[0], losses


@pytest.mark.parametrize("B,T,H", [(2, 9, 96), (1, 5, 96), (3, 7, 64), (2, 75, 96)])
def test_data_gradients_that_unpool_on_the_fly_equal_the_staged_ones(dev, B, T, H):
  """Layers 2 and 3: lr_conv3d_dgrad_pooled takes the pooled gradient and the window codes and rebuilds its dZ patch
  on the way into LDS, and so does their weight gradient (lr_conv3d_wgrad_pooled); staged reference:
  lr_unpool_code_bf16 materialises dZ and lr_conv3d_forward / lr_conv3d_wgrad (the same kernels) read it.  The same
  bf16 values reach the same MFMAs in the same order: every gradient is bit-identical."""
  from lipreading_amd import frontend as FE
  torch.manual_seed(19)
  fe = FE.ConvFrontend3D().to(dev)
  g = torch.Generator().manual_seed(20)
  clips = torch.randint(0, 256, (B, T, 3, H, H), generator=g, dtype=torch.uint8).to(dev)
  wgt = torch.randn(B, T, FE.feature_dim(H, H), generator=g).to(dev)
  res = {}
  for fused in (2, 1, 0):   # data and weight gradients un-pool on the fly | data gradients only | staged
    FE._FUSE_UNPOOL, FE._FUSE_UNPOOL_WGRAD = fused > 0, fused > 1
    try:
      fe.zero_grad()
      out = fe(clips)
      (out * wgt).sum().backward()
      torch.cuda.synchronize()
      res[fused] = [out.detach().clone()] + [p.grad.detach().clone() for p in fe.parameters_in_order()]
    finally:
      FE._FUSE_UNPOOL = FE._FUSE_UNPOOL_WGRAD = True
  for a, b in zip(res[1], res[0]):
    assert torch.equal(a, b)
  # the weight gradients that un-pool on the fly (lr_conv3d_wgrad_pooled, layers 2 and 3): the same products in the
  # same order — bit-identical; their bias gradients are the same sums in another order
  for i, (a, b) in enumerate(zip(res[2], res[0])):
    if i in (4, 6):   # conv2.bias, conv3.bias
      assert float((a - b).abs().max()) <= 2e-5 * max(1e-6, float(b.abs().max())), i
    else:
      assert torch.equal(a, b), i
  # the codes: 0..3 = position of the window's first maximum, 4 = pooled activation 0 (ReLU blocks the gradient)
  from lipreading_amd import _C
  L = _C.lib()
  x = torch.randn(B * T, 24, 24, 32, generator=g).to(dev).bfloat16()
  w = (torch.randn(64, 32, 3, 5, 5, generator=g) * 0.05).to(dev)
  bias = torch.randn(64, generator=g).to(dev) - 1.0
  wp = torch.empty((64, 75, 32), dtype=torch.bfloat16, device=dev)
  frag = L.lr_conv3d_patch_supported(24, 24, 32, 64, 3, 5, 5, 1, 1, 2, 2)
  _C.check(L.lr_conv3d_pack_weights(w.data_ptr(), wp.data_ptr(), 64, 32, 32, 3, 5, 5, frag, _C.stream_handle()), "pack")
  pooled = torch.empty((B * T, 12, 12, 64), dtype=torch.bfloat16, device=dev)
  code = torch.empty(pooled.shape, dtype=torch.uint8, device=dev)
  _C.check(L.lr_conv3d_forward_pooled(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), pooled.data_ptr(), code.data_ptr(), B, T,
                                      24, 24, 32, 64, 3, 5, 5, 1, 1, 2, 2, 1 | frag, _C.stream_handle()), "fwd")
  assert int(code.max()) == 4 and bool(((code == 4) == (pooled.float() == 0)).all())


@pytest.mark.parametrize("B,T,H,shift", [(2, 9, 96, 0), (2, 80, 96, 0), (3, 7, 64, 0), (2, 9, 96, 1), (1, 40, 96, 2)])
def test_first_layer_fused_paths_equal_the_staged_ones(dev, B, T, H, shift):
  """The first layer's kernels read the raw uint8 clip and its weight gradient rebuilds dZ from the pooled gradient and
  the window codes on the fly (lr_conv3d_forward_pooled flags & 8, lr_conv3d_wgrad_pooled).  Staged reference:
  lr_clip_to_ndhwc_bf16 copy, lr_unpool_code_bf16 + lr_conv3d_wgrad on the materialised dZ.  Same arithmetic on the
  same bf16 values: features and conv1.weight.grad are bit-identical, the bias gradient is the same sum in another
  order.  (2, 80, 96): 1440 tiles, five or six per workgroup of the weight gradient, walk starts (t = 0) in the middle
  of a workgroup's range and a clip boundary; shift: the clip starts 1 / 2 bytes off a dword boundary, where the
  weight gradient reads it in bytes instead of dwords and the forward in bytes instead of pairs."""
  from lipreading_amd import frontend as FE
  torch.manual_seed(9)
  fe = FE.ConvFrontend3D().to(dev)
  g = torch.Generator().manual_seed(10)
  raw = torch.randint(0, 256, (B * T * 3 * H * H + 4,), generator=g, dtype=torch.uint8).to(dev)
  clips = raw[shift:shift + B * T * 3 * H * H].view(B, T, 3, H, H)
  assert clips.data_ptr() % 4 == shift
  wgt = torch.randn(B, T, FE.feature_dim(H, H), generator=g).to(dev)
  res = {}
  for fused in (True, False):
    FE._FUSE_FIRST_LAYER = fused
    try:
      fe.zero_grad()
      out = fe(clips)
      (out * wgt).sum().backward()
      res[fused] = [out.detach().clone()] + [p.grad.detach().clone() for p in fe.parameters_in_order()]
    finally:
      FE._FUSE_FIRST_LAYER = True
  assert torch.equal(res[True][0], res[False][0])
  assert torch.equal(res[True][1], res[False][1])                  # conv1.weight.grad
  assert float((res[True][2] - res[False][2]).abs().max()) <= 1e-5 * float(res[False][2].abs().max())   # conv1.bias.grad
  for a, b in zip(res[True][3:], res[False][3:]):
    assert torch.equal(a, b)


def test_side_stream_weight_gradients_are_the_same_gradients(dev):
  """encoder.overlap_weight_grads: the weight-gradient half of a recurrent layer's backward runs on a
  side stream (lr_rnn_layer_backward_parts) beside the layer below.  Same kernels on the same data:
  every gradient is bit-identical to the single-stream backward."""
  from lipreading_amd import encoder as E
  from lipreading_amd.ctc import ctc_loss_with_status
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader, feature_dim
  from lipreading_amd.optim import FlatParameters
  torch.manual_seed(11)
  H = 96
  enc = E.VideoEncoder(feature_dim(H, H), 256, rnn_type='GRU', num_layers=2, bidirectional=True, enable_ctc=True,
                       vocab_size=64, char2idx=default_char2idx())
  model = PixelLipReader(enc, ConvFrontend3D()).to(dev).train()
  flat = FlatParameters(model)          # dense .grad buffers: gradients are written in place
  g = torch.Generator().manual_seed(12)
  B, T = 4, 10
  clips = torch.randint(0, 256, (B, T, 3, H, H), generator=g, dtype=torch.uint8).to(dev)
  lens = torch.tensor([10, 10, 7, 3], device=dev)
  labels = torch.randint(4, 64, (B, 3), generator=g).to(dev)
  ll = torch.full((B,), 3, device=dev)
  assert E.overlap_weight_grads
  res = {}
  for overlap in (True, False):
    E.overlap_weight_grads = overlap
    try:
      flat.zero_grad()
      lp, _, _ = model(clips, lens, max_len=T)
      loss, _, _ = ctc_loss_with_status(lp, labels, lens, ll, 'mean')
      loss.backward()
      E.flush_deferred()
      torch.cuda.synchronize()
      res[overlap] = flat.grad.clone()
    finally:
      E.overlap_weight_grads = True
  assert float(res[True].abs().max()) > 0
  assert torch.equal(res[True], res[False])


def test_pixel_regime_fast_paths_track_the_plain_ones(dev):
  """PixelLipReader's defaults at the metric's shape family (96x96, BiGRU-256 x2): bf16 features
  handed to the encoder as stored, split-bf16 input projection and recurrent weight gradient, the
  recurrence as one launch per pass with bf16 operands.  Against the same weights on the plain
  paths (fp32 features, fp32 MFMA GEMMs, fp32 step kernels): losses and gradients agree to the
  level of the bf16 operand rounding."""
  from lipreading_amd.ctc import ctc_loss_with_status
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader, feature_dim
  torch.manual_seed(7)
  H = 96
  enc = VideoEncoder(feature_dim(H, H), 256, rnn_type='GRU', num_layers=2, bidirectional=True, enable_ctc=True,
                     vocab_size=64, char2idx=default_char2idx())
  model = PixelLipReader(enc, ConvFrontend3D()).to(dev).train()
  assert enc.recurrence == 'split' and enc.input_projection == 'bf16x3'
  g = torch.Generator().manual_seed(8)
  B, T = 3, 11
  clips = torch.randint(0, 256, (B, T, 3, H, H), generator=g, dtype=torch.uint8).to(dev)
  lens = torch.tensor([11, 9, 6], device=dev)
  labels = torch.randint(4, 64, (B, 3), generator=g).to(dev)
  ll = torch.full((B,), 3, device=dev)
  res = {}
  for fast in (True, False):
    model.zero_grad()
    if fast:
      lp, hid, _ = model(clips, lens, max_len=T)
    else:   # the plain paths, same parameters
      enc.recurrence, enc.input_projection, enc.input_is_bf16 = 'f32', 'f32', False
      try:
        feats = model.frontend(clips)
        assert feats.dtype == torch.float32
        lp, hid, _ = enc(feats.reshape(B, T, -1, 1), lens, max_len=T)
      finally:
        enc.recurrence, enc.input_projection, enc.input_is_bf16 = 'split', 'bf16x3', True
    loss, status, _ = ctc_loss_with_status(lp, labels, lens, ll, 'mean')
    assert int(status) == 0
    loss.backward()
    res[fast] = [loss.detach().cpu().reshape(1), hid.detach().cpu()] + [p.grad.cpu().clone() for p in model.parameters()]
  # measured on MI355X (B = 8, T = 40): loss 4e-6, hidden 7e-4, gradients 4e-4 .. 2.4e-3 relative
  assert abs(float(res[True][0]) - float(res[False][0])) < 1e-3 * abs(float(res[False][0]))
  for a, b in zip(res[True][1:], res[False][1:]):
    assert float((a - b).norm()) <= 1e-2 * max(1e-6, float(b.norm())), tuple(a.shape)
  assert float((res[True][1] - res[False][1]).abs().max()) > 0   # the fast paths really ran


def test_lip_crop_matches_oracle(dev):
  """A9 (build-defined): mouth-landmark bounding box -> square window -> bilinear 96x96."""
  from lipreading_amd.landmarks import lip_crop
  rng = np.random.RandomState(123456)
  n, H, W = 6, 120, 160
  frames = rng.randint(0, 256, (n, 3, H, W)).astype(np.uint8)
  lm = np.zeros((n, 68, 3), np.float32)
  lm[:, :, 0] = rng.uniform(20, 140, (n, 68))
  lm[:, :, 1] = rng.uniform(20, 100, (n, 68))
  lm[0, 48:68, 0] = rng.uniform(60, 100, 20); lm[0, 48:68, 1] = rng.uniform(70, 90, 20)
  lm[1, 48:68, 0] = rng.uniform(-5, 12, 20)                      # window hangs over the left edge
  lm[2, 48:68, :2] = 50.0                                        # degenerate box -> minimum side
  want = O.lip_crop(frames, lm)
  got = lip_crop(torch.tensor(frames, device=dev), torch.tensor(lm, device=dev)).cpu().numpy()
  assert got.shape == (n, 3, 96, 96) and got.dtype == np.uint8
  diff = np.abs(got.astype(int) - want.astype(int))
  assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())
  # a crop of a constant image is constant; a crop feeds the frontend as-is
  flat = lip_crop(torch.full((1, 3, 64, 64), 77, dtype=torch.uint8, device=dev),
                  torch.tensor(lm[:1], device=dev))
  assert int(flat.min()) == int(flat.max()) == 77


def test_patch_resident_layer2_matches_implicit_gemm_and_oracle(dev):
  """At the metric's 96x96 size layer 2 (24x24, 32->64, taps 3x5x5) runs the patch-resident kernels
  (forward and data gradient).  They must agree with the implicit-GEMM kernels they replace (same
  bf16 operands, fp32 accumulation in another order) and with the torch-CPU oracle.  T=7 makes the
  4-frame tiles ragged and puts a clip boundary inside a tile (taps across it are skipped)."""
  from lipreading_amd import frontend as FE
  torch.manual_seed(5)
  fe = FE.ConvFrontend3D().to(dev)
  g = torch.Generator().manual_seed(6)
  B, T, H = 3, 7, 96
  clips = torch.randint(0, 256, (B, T, 3, H, H), generator=g, dtype=torch.uint8)
  wgt = torch.randn(B, T, FE.feature_dim(H, H), generator=g)
  res = {}
  for patch in (True, False):
    FE._PATCH_KERNELS = patch
    try:
      fe.zero_grad()
      out = fe(clips.to(dev))
      (out * wgt.to(dev)).sum().backward()
      res[patch] = [out.detach().cpu().numpy()] + [p.grad.cpu().numpy().copy() for p in fe.parameters_in_order()]
    finally:
      FE._PATCH_KERNELS = True
  for a, b in zip(res[True], res[False]):
    assert rel_err(a, b) < 6e-3     # bf16 rounding of intermediate activations differs by <= 1 ulp
  params_cpu = [p.detach().cpu().clone().requires_grad_(True) for p in fe.parameters_in_order()]
  ref = O.conv_frontend(clips, params_cpu)
  (ref * wgt).sum().backward()
  assert rel_err(res[True][0], ref.detach().numpy()) < 2e-2
  for a, q in zip(res[True][1:], params_cpu):
    assert rel_err(a, q.grad.numpy()) < 4e-2, tuple(q.shape)


def test_layer2_tail_tiles_split_by_frame_are_the_same_numbers(dev):
  """The patch-resident layer-2 kernels run the few tiles left over after the last full round of one workgroup per
  CU split by frame (four waves share a frame's accumulator tiles).  7 clips x 49 frames = 343 frames = 258 tiles:
  tiles 256 and 257 (frames 340-342 + one frame past the end, rows 8-23) take that path.  Every output frame depends
  on its own clip only and each accumulator sums its taps in the same order wherever it is computed, so the 7-clip
  launch must reproduce, bit for bit, a 6-clip launch (222 tiles: no split) and a launch of the last clip alone."""
  from lipreading_amd import _C
  L = _C.lib()
  st = _C.stream_handle()
  bf = torch.bfloat16
  B, T, h, w, cin, cout = 7, 49, 24, 24, 32, 64
  kt, kh, kw, pt, ph, pw = 3, 5, 5, 1, 2, 2
  g = torch.Generator().manual_seed(17)
  x = (torch.randn(B * T, h, w, cin, generator=g) * 0.5).clamp_min(0).to(bf).to(dev)
  dz = (torch.randn(B * T, h, w, cout, generator=g) * 0.1).to(bf).to(dev)
  weight = (torch.randn(cout, cin, kt, kh, kw, generator=g) * 0.02).to(dev)
  bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
  frag = L.lr_conv3d_patch_supported(h, w, cin, cout, kt, kh, kw, 1, pt, ph, pw)
  fragd = L.lr_conv3d_patch_supported(h, w, cout, cin, kt, kh, kw, 1, pt, ph, pw)
  assert frag == 2 and fragd == 2
  wp = torch.empty((cout, kt * kh * kw, cin), dtype=bf, device=dev)
  wd = torch.empty((cin, kt * kh * kw, cout), dtype=bf, device=dev)
  _C.check(L.lr_conv3d_pack_weights(weight.data_ptr(), wp.data_ptr(), cout, cin, cin, kt, kh, kw, frag, st))
  _C.check(L.lr_conv3d_pack_weights(weight.data_ptr(), wd.data_ptr(), cout, cin, cin, kt, kh, kw, 1 | fragd, st))

  def run(first, nclips):
    xs, zs = x[first * T:(first + nclips) * T].contiguous(), dz[first * T:(first + nclips) * T].contiguous()
    F = nclips * T
    pooled = torch.empty((F, h // 2, w // 2, cout), dtype=bf, device=dev)
    code = torch.empty(pooled.shape, dtype=torch.uint8, device=dev)
    full = torch.empty((F, h, w, cout), dtype=bf, device=dev)
    dx = torch.empty((F, h, w, cin), dtype=bf, device=dev)
    _C.check(L.lr_conv3d_forward_pooled(xs.data_ptr(), wp.data_ptr(), bias.data_ptr(), pooled.data_ptr(), code.data_ptr(),
                                        nclips, T, h, w, cin, cout, kt, kh, kw, 1, pt, ph, pw, 1 | frag, st))
    _C.check(L.lr_conv3d_forward(xs.data_ptr(), wp.data_ptr(), bias.data_ptr(), full.data_ptr(), nclips, T, h, w, cin,
                                 cout, kt, kh, kw, 1, pt, ph, pw, 1 | frag, st))
    _C.check(L.lr_conv3d_forward(zs.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), nclips, T, h, w, cout, cin, kt, kh,
                                 kw, 1, pt, ph, pw, fragd, st))
    torch.cuda.synchronize()
    return [t.view(torch.uint8 if t.dtype == torch.uint8 else torch.int16).cpu() for t in (pooled, code, full, dx)]

  whole = run(0, 7)
  head, tail = run(0, 6), run(6, 1)
  for name, a, b, c in zip(("pooled", "code", "full", "dgrad"), whole, head, tail):
    assert torch.equal(a[:6 * T], b), name
    assert torch.equal(a[6 * T:], c), name
  assert int(whole[2][6 * T:].ne(0).sum()) > 0 and int(whole[3][6 * T:].ne(0).sum()) > 0


@pytest.mark.parametrize("layer,B,T,H", [(2, 2, 45, 24), (2, 3, 7, 24), (2, 1, 1, 24), (2, 2, 9, 16), (3, 2, 45, 12),
                                         (3, 5, 9, 12), (3, 1, 2, 12)])
def test_stride1_weight_gradient_against_torch(dev, layer, B, T, H):
  """lr_conv3d_wgrad on the geometries of layers 2 and 3 (the persistent LDS-transpose-read kernel of
  lr_conv_wgrad.hip: 255 workgroups, units split by row tile, padding through out-of-range buffer loads, the
  next tile loaded and stored among the current tile's MFMAs) against torch's fp32 conv3d weight gradient of the
  same bf16 operands.  B*T = 90 frames gives a workgroup up to three tiles (the table, both LDS buffers and the
  tile of zeros after the last one are all exercised); 21, 45 and 2 frames leave workgroups without any tile and
  put clip boundaries and the end of the batch inside a 2-frame tile; T = 1, 2: every temporal tap but the
  middle one(s) leaves the clip; a 16-row layer-2 input takes the 4-row tiles.  Accumulation is fp32 in another
  order: 2e-5 of the largest gradient."""
  from lipreading_amd import _C
  L = _C.lib()
  st = _C.stream_handle()
  cin, cout, k, hw = (32, 64, (3, 5, 5), 24) if layer == 2 else (64, 96, (3, 3, 3), 12)
  pads = (1, k[1] // 2, k[2] // 2)
  g = torch.Generator().manual_seed(100 * layer + T)
  x = (torch.randn(B, T, H, hw, cin, generator=g)).clamp_min(0).bfloat16()
  dz = (torch.randn(B, T, H, hw, cout, generator=g) * 0.1).bfloat16()
  w = torch.zeros(cout, cin, *k, requires_grad=True)
  out = torch.nn.functional.conv3d(x.float().permute(0, 4, 1, 2, 3), w, None, stride=1, padding=pads)
  out.backward(dz.float().permute(0, 4, 1, 2, 3))
  ref_w, ref_b = w.grad.numpy(), dz.float().sum((0, 1, 2, 3)).numpy()
  xd, dzd = x.to(dev).contiguous(), dz.to(dev).contiguous()
  wbytes = L.lr_conv3d_wgrad_workspace_bytes(cout, cin, *k)
  ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
  for accumulate in (0, 1):
    dw = torch.full((cout, cin) + k, 0.5 if accumulate else float("nan"), device=dev)
    db = torch.full((cout,), 0.25 if accumulate else float("nan"), device=dev)
    _C.check(L.lr_conv3d_wgrad(xd.data_ptr(), dzd.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), wbytes,
                               accumulate, B, T, H, hw, cin, cin, cout, *k, 1, *pads, st))
    got_w, got_b = dw.cpu().numpy() - 0.5 * accumulate, db.cpu().numpy() - 0.25 * accumulate
    assert np.abs(got_w - ref_w).max() <= 2e-5 * np.abs(ref_w).max() + 1e-6 * accumulate, (layer, B, T, H, accumulate)
    assert np.abs(got_b - ref_b).max() <= 1e-4 * max(1.0, np.abs(ref_b).max())


# ---- headline regime (BASELINE configs[1]) against the ORACLE, not against this repo's other paths ----
# Stated tolerances, ~3x what was measured: |loss_hip - loss_oracle| <= 1e-4 absolute on the
# 'mean' CTC loss — north_star's bar, the same as bench.PARITY_TOL_PIXELS at B = 32 (round 2: 1e-3 here, with the
# single-plane bf16 recurrence measuring 5-7e-5; the fp32-faithful recurrence is the default now); log-probs 1e-3;
# encoder/CTC-head gradients within 3e-3 of the oracle's norm per tensor (measured 6-8e-4), conv gradients within
# 1e-2 (measured 2-2.5e-3: bf16 activations, a rounding flip moves an activation by 2^-8 relative).
PIXEL_LOSS_TOL = 1e-4


@pytest.mark.parametrize("B,lens", [(8, None), (8, [40, 52, 52, 60, 75, 75, 75, 75]), (32, None)])   # (32: the headline batch)
def test_pixel_regime_defaults_match_the_oracle(dev, B, lens):
  """Same uint8 clips and the same weights through (a) PixelLipReader with its DEFAULTS (bf16 conv
  stack, bf16 features handed over as stored, split-bf16 input projection, one-launch fp32-faithful
  recurrence) -> HIP ctc_loss, and (b) oracle.conv_frontend(emulate_bf16=True) ->
  OracleVideoEncoder (the reference's fp32 nn.GRU path) -> oracle.ctc_loss, at the metric's shape family
  (T = 75, 96x96, 2 x BiGRU-256).  The loss and every gradient are compared."""
  from lipreading_amd.ctc import ctc_loss_with_status
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader, feature_dim
  T, H = 75, 96
  torch.manual_seed(31)
  ref = O.OracleVideoEncoder(feature_dim(H, H), 256, rnn_type='GRU', num_layers=2, bidirectional=True,
                             enable_ctc=True, vocab_size=64, char2idx=O.default_char2idx()).train()
  enc = VideoEncoder(feature_dim(H, H), 256, rnn_type='GRU', num_layers=2, bidirectional=True, enable_ctc=True,
                     vocab_size=64, char2idx=default_char2idx())
  enc.load_state_dict(ref.state_dict())
  fe = ConvFrontend3D()
  convs = [p.detach().clone().requires_grad_(True) for p in fe.parameters_in_order()]
  model = PixelLipReader(enc, fe).to(dev).train()
  assert enc.recurrence == 'split' and enc.input_projection == 'bf16x3'
  g = torch.Generator().manual_seed(32)
  clips = torch.randint(0, 256, (B, T, 3, H, H), generator=g, dtype=torch.uint8)
  lens = torch.tensor(lens) if lens is not None else torch.full((B,), T)
  labels = torch.randint(4, 64, (B, 12), generator=g)
  ll = torch.full((B,), 12)
  # oracle
  feats = O.conv_frontend(clips, convs, emulate_bf16=True)
  lp_r, _, _ = ref(feats.reshape(B, T, -1, 1), lens)
  loss_r = O.ctc_loss(lp_r, labels, lens, ll, 'mean')
  loss_r.backward()
  # HIP
  lp_h, _, _ = model(clips.to(dev), lens.to(dev), max_len=T)
  loss_h, status, _ = ctc_loss_with_status(lp_h, labels.to(dev), lens.to(dev), ll.to(dev), 'mean')
  assert int(status) == 0
  loss_h.backward()
  from lipreading_amd import encoder as E
  E.flush_deferred()
  torch.cuda.synchronize()
  d_loss = abs(float(loss_h) - float(loss_r))
  valid = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(-1)
  d_lp = float(((lp_h.detach().cpu() - lp_r.detach()) * valid).abs().max())
  print("pixel parity: loss hip %.7f oracle %.7f |d| %.3g; max |d log-prob| %.3g" % (float(loss_h), float(loss_r), d_loss, d_lp))
  assert d_loss <= PIXEL_LOSS_TOL, (float(loss_h), float(loss_r))
  assert d_lp <= 1e-3
  ref_grads = dict(ref.named_parameters())
  worst = {}
  for k, p in enc.named_parameters():
    a, b = p.grad.cpu(), ref_grads[k].grad
    worst[k] = float((a - b).norm()) / max(1e-8, float(b.norm()))
    assert worst[k] <= 3e-3, (k, worst[k])
  for i, (p, q) in enumerate(zip(fe.parameters_in_order(), convs)):
    r = float((p.grad.cpu() - q.grad).norm()) / max(1e-8, float(q.grad.norm()))
    worst["conv[%d]" % i] = r
    assert r <= 1e-2, (i, r)
  print("pixel parity: worst relative gradient differences", {k: float("%.2g" % v) for k, v in worst.items()})


@pytest.mark.parametrize("B", [8, 32, 64])
def test_pixel_step_at_the_sharded_batch_shapes(dev, B):
  """BASELINE configs[3] is (B=64,T=75,3,96,96) over 8 GPUs: 8 clips per rank (strong) or 32 per rank
  (weak, the bench's choice); a single GPU must also take the whole B=64.  One full optimisation step
  of PixelLipReader at each of those per-rank batches: finite loss, no skipped batch, weights move."""
  from lipreading_amd.ctc import ctc_loss_with_status
  from lipreading_amd.data import default_char2idx
  from lipreading_amd.encoder import VideoEncoder
  from lipreading_amd.frontend import ConvFrontend3D, PixelLipReader, feature_dim
  from lipreading_amd.optim import FlatParameters, FusedAdam
  torch.manual_seed(41)
  enc = VideoEncoder(feature_dim(96, 96), 256, rnn_type='GRU', num_layers=2, bidirectional=True, enable_ctc=True,
                     vocab_size=64, char2idx=default_char2idx())
  model = PixelLipReader(enc, ConvFrontend3D()).to(dev).train()
  flat = FlatParameters(model)
  opt = FusedAdam(flat, lr=1e-4)
  g = torch.Generator().manual_seed(42)
  clips = torch.randint(0, 256, (B, 75, 3, 96, 96), generator=g, dtype=torch.uint8).to(dev)
  lens = torch.full((B,), 75, device=dev)
  labels = torch.randint(4, 64, (B, 31), generator=g).to(dev)
  ll = torch.full((B,), 31, device=dev)
  before = flat.data.clone()
  losses = []
  for _ in range(2):
    opt.zero_grad()
    lp, _, _ = model(clips, lens, max_len=75)
    loss, status, _ = ctc_loss_with_status(lp, labels, lens, ll, 'mean')
    loss.backward()
    opt.step(grad_norm=50, skip=status)
    assert int(status) == 0
    losses.append(float(loss))
  assert np.isfinite(losses).all() and losses[1] < losses[0], losses
  assert float((flat.data - before).abs().max()) > 0


def test_packing_every_weight_operand_in_one_launch_equals_packing_them_one_by_one(dev):
  """lr_conv3d_pack_weights_multi (what the frontend's forward launches once per step: the forward operand of
  each layer and the data-gradient operands of layers 2, 3) against lr_conv3d_pack_weights item by item:
  bit-identical bf16 operands, fragment-major orders included."""
  import ctypes
  from lipreading_amd import _C
  from lipreading_amd.frontend import LAYERS, _pack_weights, _pad4
  L = _C.lib()
  st = _C.stream_handle()
  g = torch.Generator().manual_seed(21)
  packs, singles = [], []
  h = w = 96
  for li, (cin, cout, (kt, kh, kw), stride, (pt, ph, pw)) in enumerate(LAYERS):
    cin_p = _pad4(cin)
    weight = torch.randn(cout, cin, kt, kh, kw, generator=g).to(dev)
    ho, wo = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
    flags = [L.lr_conv3d_patch_supported(h, w, cin_p, cout, kt, kh, kw, stride, pt, ph, pw)]
    if li > 0:
      flags.append(1 | L.lr_conv3d_patch_supported(ho, wo, cout, cin, kt, kh, kw, 1, pt, ph, pw))
    for f in flags:
      shape = (cin, kt * kh * kw, cout) if f & 1 else (cout, kt * kh * kw, cin_p)
      a = torch.zeros(shape, dtype=torch.bfloat16, device=dev)
      b = torch.zeros(shape, dtype=torch.bfloat16, device=dev)
      packs.append((weight, a, cout, cin, cin_p, kt, kh, kw, f))
      _C.check(L.lr_conv3d_pack_weights(weight.data_ptr(), b.data_ptr(), cout, cin, cin_p, kt, kh, kw, f, st), "single")
      singles.append(b)
    h, w = ho // 2, wo // 2
  assert len(packs) == 5
  _pack_weights(L, packs, st)
  torch.cuda.synchronize()
  for p, b in zip(packs, singles):
    assert torch.equal(p[1].view(torch.int16), b.view(torch.int16))
