"""3-D convolution frontend: pixel clips -> per-frame features for the sequence encoder.

BUILD-DEFINED: the reference has no conv frontend — its conv stack is commented
out (`src/models/lipreader/model.py:122,153-156`) and the `ced` experiment configs are empty
files — but BASELINE.json's north_star and metric name one ("clips of shape (B,75,3,96,96)",
"im2col+MFMA GEMM for the 3D convs").  The specification below is therefore this repo's own,
LipNet-style (three spatio-temporal convolutions, each followed by ReLU and a (1,2,2) max-pool):

    conv1  Conv3d(3, 32,  k=(3,5,5), stride=(1,2,2), padding=(1,2,2))   96x96 -> 48x48 -> pool 24x24
    conv2  Conv3d(32, 64, k=(3,5,5), stride=1,       padding=(1,2,2))   24x24          -> pool 12x12
    conv3  Conv3d(64, 96, k=(3,3,3), stride=1,       padding=(1,1,1))   12x12          -> pool  6x6
    features[b, t] = pooled3[b, t].reshape(6*6*96)   (h, w, c order; 3456 per frame at 96x96)

Parameters are held as nn.Conv3d modules (state_dict keys conv{1,2,3}.{weight,bias}, torch layout
and initialisation) but never called: the arithmetic is the HIP path of lr_conv*.hip — bf16
channels-last activations, implicit-GEMM convolution on v_mfma_f32_32x32x16_bf16 with fp32
accumulation, fused bias+ReLU.  There is NO reference parity for this stage; the oracle is torch
conv3d/max_pool3d on the CPU (oracle/torch_oracle.py conv_frontend).
"""
import ctypes
import os

import torch
import torch.nn as nn

from . import _C
from .encoder import _direct_grads, _notify

# (Cin, Cout, (KT,KH,KW), spatial stride, (pt,ph,pw))
LAYERS = ((3, 32, (3, 5, 5), 2, (1, 2, 2)),
          (32, 64, (3, 5, 5), 1, (1, 2, 2)),
          (64, 96, (3, 3, 3), 1, (1, 1, 1)))


# tests switch this off to compare the patch-resident kernels with the implicit-GEMM kernels
_PATCH_KERNELS = True
# ... and this one to compare the first layer's fused paths (raw uint8 clip into the kernels, weight
# gradient that un-pools on the fly) with the staged ones (bf16 clip copy, materialised dZ)
_FUSE_FIRST_LAYER = True
# ... and this one to compare the data gradients that un-pool on the fly (lr_conv3d_dgrad_pooled: layers 2 and 3 take
# the pooled gradient and the window codes) with the staged form (lr_unpool_code_bf16, then lr_conv3d_forward on dZ)
# (LIPREADING_FUSE_UNPOOL: experiment switch for A/B timing — 0 = staged, 1 = only the data gradients un-pool on the fly)
_FUSE_UNPOOL = os.environ.get("LIPREADING_FUSE_UNPOOL", "2") != "0"
_FUSE_UNPOOL_WGRAD = os.environ.get("LIPREADING_FUSE_UNPOOL", "2") == "2"


def _pad4(c):
  return (c + 3) // 4 * 4


def feature_dim(H, W):
  return 96 * (H // 16) * (W // 16)


# layers 2 and 3: weight gradient on a side stream beside the data gradient.  Measured (B=32, T=75): with the
# recurrent encoder, whose first layer's weight-gradient GEMMs already run beside the conv backward on the encoder's
# side stream, a third queue fills the gaps (pixel step 2.77 -> 2.71 ms); with the transformer encoder — two queues
# only, the weight gradient and the data gradient of a layer simply sharing the chip — it costs 3 % (5.37 -> 5.54
# ms).  So: on exactly when the encoder has deferred work in flight; LIPREADING_CONV_WGRAD_SIDE=0 / 1 forces it.
_WGRAD_SIDE_ENV = os.environ.get("LIPREADING_CONV_WGRAD_SIDE", "auto")
_WGRAD_SIDE_STREAM = _WGRAD_SIDE_ENV != "0"
# the side stream's weight half of a layer starts when the layer's pooled gradient exists (1) — beside the layer's OWN
# data gradient — or (0, rounds 2-4) when that data gradient has finished, i.e. beside the layer BELOW's
_WGRAD_EARLY = os.environ.get("LIPREADING_CONV_WGRAD_EARLY", "1") != "0"
_conv_side = None


def _conv_side_stream(device):
  global _conv_side
  if _conv_side is None or _conv_side.device != device:
    _conv_side = torch.cuda.Stream(device=device)
  return _conv_side


def _pack_weights(L, packs, st):
  """packs: (weight fp32, out bf16, cout, cin, cin_pad, kt, kh, kw, dgrad flags) — one launch for all of them."""
  n = len(packs)
  ptrs = (ctypes.c_void_p * n)(*[p[0].data_ptr() for p in packs])
  outs = (ctypes.c_void_p * n)(*[p[1].data_ptr() for p in packs])
  cols = [(ctypes.c_int * n)(*[p[k] for p in packs]) for k in range(2, 9)]
  _C.check(L.lr_conv3d_pack_weights_multi(n, ptrs, outs, *cols, st), "lr_conv3d_pack_weights_multi")


class _ConvFrontendFunction(torch.autograd.Function):
  @staticmethod
  def forward(ctx, clips, out_bf16, *params):
    L = _C.lib()
    st = _C.stream_handle()
    B, T, C, H, W = clips.shape
    assert C == 3 and H % 16 == 0 and W % 16 == 0
    dev = clips.device
    frames = B * T
    bf = torch.bfloat16
    is_u8 = clips.dtype == torch.uint8
    src = clips if is_u8 else clips.to(torch.float32)
    src = src.contiguous()
    cin0, cout0, (kt0, kh0, kw0), s0, (pt0, ph0, pw0) = LAYERS[0]
    # uint8 clips go straight into the first layer's patch kernels (forward and weight gradient), which
    # scale and convert while they fill LDS: no bf16 copy of the clip is written or kept
    raw_u8 = bool(is_u8 and _PATCH_KERNELS and _FUSE_FIRST_LAYER and L.lr_conv3d_pool_fusion_supported(
        H, W, 4, cout0, kt0, kh0, kw0, s0, pt0, ph0, pw0) and L.lr_conv3d_wgrad_pooled_supported(
        H, W, 4, cin0, cout0, kt0, kh0, kw0, s0, pt0, ph0, pw0))
    if raw_u8:
      x = src.reshape(frames, 3, H, W)
    else:
      x = torch.empty((frames, H, W, 4), dtype=bf, device=dev)
      _C.check(L.lr_clip_to_ndhwc_bf16(src.data_ptr(), 1 if is_u8 else 0, x.data_ptr(), frames, H, W, st),
               "lr_clip_to_ndhwc_bf16")
    saved = [x]
    # every bf16 weight operand of the step in ONE launch: the forward operand of each layer (fragment-major where
    # the layer has a patch-resident kernel) and, when a backward will follow, the flipped / channel-transposed
    # operand of the upper layers' data gradients (kept for the backward)
    packs, fwd_ops, dgrad_ops = [], [], {}
    h, w = H, W
    for li, (cin, cout, (kt, kh, kw), stride, (pt, ph, pw)) in enumerate(LAYERS):
      cin_p = _pad4(cin)
      frag = L.lr_conv3d_patch_supported(h, w, cin_p, cout, kt, kh, kw, stride, pt, ph, pw) if _PATCH_KERNELS else 0
      wp = torch.empty((cout, kt * kh * kw, cin_p), dtype=bf, device=dev)
      packs.append((params[2 * li], wp, cout, cin, cin_p, kt, kh, kw, frag))
      fwd_ops.append((wp, frag))
      ho, wo = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
      if li > 0 and any(ctx.needs_input_grad[2:]):
        fragd = L.lr_conv3d_patch_supported(ho, wo, cout, cin, kt, kh, kw, 1, pt, ph, pw) if _PATCH_KERNELS else 0
        wd = torch.empty((cin, kt * kh * kw, cout), dtype=bf, device=dev)
        packs.append((params[2 * li], wd, cout, cin, cin_p, kt, kh, kw, 1 | fragd))
        dgrad_ops[li] = (wd, fragd)
      h, w = ho // 2, wo // 2
    _pack_weights(L, packs, st)
    ctx.dgrad_ops = dgrad_ops
    h, w = H, W
    for li, (cin, cout, (kt, kh, kw), stride, (pt, ph, pw)) in enumerate(LAYERS):
      weight, bias = params[2 * li], params[2 * li + 1]
      cin_p = _pad4(cin)
      wp, frag = fwd_ops[li]
      ho, wo = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
      pooled = torch.empty((frames, ho // 2, wo // 2, cout), dtype=bf, device=dev)
      fuse = _PATCH_KERNELS and (frag or cin_p == 4) and L.lr_conv3d_pool_fusion_supported(
          h, w, cin_p, cout, kt, kh, kw, stride, pt, ph, pw)
      if fuse:
        # ReLU + max-pool in the conv epilogue: the full-resolution activation is never written; the
        # backward gets the pooled activation and each window's argmax code (uint8) instead
        code = torch.empty(pooled.shape, dtype=torch.uint8, device=dev)
        _C.check(L.lr_conv3d_forward_pooled(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), pooled.data_ptr(),
                                            code.data_ptr(), B, T, h, w, cin_p, cout, kt, kh, kw, stride, pt, ph,
                                            pw, 1 | frag | (8 if (li == 0 and raw_u8) else 0), st),
                 "lr_conv3d_forward_pooled")
        saved += [code, pooled]
      else:
        act = torch.empty((frames, ho, wo, cout), dtype=bf, device=dev)
        _C.check(L.lr_conv3d_forward(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), act.data_ptr(), B, T, h,
                                     w, cin_p, cout, kt, kh, kw, stride, pt, ph, pw, 1 | frag, st),
                 "lr_conv3d_forward")
        _C.check(L.lr_maxpool_hw2_bf16(act.data_ptr(), pooled.data_ptr(), frames, ho, wo, cout, st),
                 "lr_maxpool_hw2_bf16")
        saved += [act, pooled]
      x, h, w = pooled, ho // 2, wo // 2
    if out_bf16:   # the encoder's split-bf16 input projection takes the pooled activation as it is
      feats = x.reshape(B, T, h * w * 96)
    else:
      feats = torch.empty((B, T, h * w * 96), dtype=torch.float32, device=dev)
      _C.check(L.lr_bf16_to_f32(x.data_ptr(), feats.data_ptr(), feats.numel(), st), "lr_bf16_to_f32")
    ctx.save_for_backward(*saved, *params)
    ctx.dims = (B, T, H, W)
    return feats

  @staticmethod
  def backward(ctx, dfeat):
    from . import encoder as _enc
    L = _C.lib()
    st = _C.stream_handle()
    B, T, H, W = ctx.dims
    frames = B * T
    saved = ctx.saved_tensors
    acts = saved[:7]            # x0, act1|code1, pool1, act2|code2, pool2, act3|code3, pool3
    params = saved[7:]
    dev = dfeat.device
    bf = torch.bfloat16
    direct = _direct_grads(params)
    grads = [p.grad for p in params] if direct else [torch.empty_like(p) for p in params]
    if dfeat.dtype == bf:
      dP = dfeat.contiguous().reshape(acts[6].shape)
    else:
      dfeat = dfeat.contiguous().to(torch.float32)
      dP = torch.empty(acts[6].shape, dtype=bf, device=dev)
      _C.check(L.lr_f32_to_bf16(dfeat.data_ptr(), dP.data_ptr(), dfeat.numel(), st), "lr_f32_to_bf16")
    # spatial size of each layer's input
    sizes = [(H, W)]
    for (_, _, (kt, kh, kw), stride, (pt, ph, pw)) in LAYERS:
      h, w = sizes[-1]
      sizes.append((((h + 2 * ph - kh) // stride + 1) // 2, ((w + 2 * pw - kw) // stride + 1) // 2))
    side_keep = []
    for li in (2, 1, 0):
      cin, cout, (kt, kh, kw), stride, (pt, ph, pw) = LAYERS[li]
      cin_p = _pad4(cin)
      x_in, act, pooled = acts[2 * li], acts[2 * li + 1], acts[2 * li + 2]
      h, w = sizes[li]
      ho, wo = 2 * pooled.shape[1], 2 * pooled.shape[2]
      wbytes = max(L.lr_conv3d_wgrad_workspace_bytes(cout, cin_p, kt, kh, kw), L.lr_unpool_workspace_bytes(cout))
      ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
      if act.dtype == torch.uint8 and li == 0 and ((_PATCH_KERNELS and _FUSE_FIRST_LAYER) or x_in.dtype == torch.uint8) and L.lr_conv3d_wgrad_pooled_supported(
          h, w, cin_p, cin, cout, kt, kh, kw, stride, pt, ph, pw):
        # first layer (no data gradient needed): the weight-gradient kernel un-pools on the fly
        _C.check(L.lr_conv3d_wgrad_pooled(x_in.data_ptr(), pooled.data_ptr(), act.data_ptr(), dP.data_ptr(),
                                          grads[0].data_ptr(), grads[1].data_ptr(), ws.data_ptr(), wbytes,
                                          1 if direct else 0, B, T, h, w, cin_p, cin, cout, kt, kh, kw, stride, pt,
                                          ph, pw, 1 if x_in.dtype == torch.uint8 else 0, st), "lr_conv3d_wgrad_pooled")
        continue
      coded = act.dtype == torch.uint8   # the forward fused the pooling: act holds the window codes
      # data gradient of a stride-1 "same" convolution = the forward kernel on dZ with the flipped,
      # channel-transposed weights (packed with the forward operands: nothing updates the weights in between)
      wd = frag = None
      if li > 0:
        if li in ctx.dgrad_ops:
          wd, frag = ctx.dgrad_ops[li]
        else:
          wd = torch.empty((cin, kt * kh * kw, cout), dtype=bf, device=dev)
          frag = L.lr_conv3d_patch_supported(ho, wo, cout, cin, kt, kh, kw, 1, pt, ph, pw) if _PATCH_KERNELS else 0
          _C.check(L.lr_conv3d_pack_weights(params[2 * li].data_ptr(), wd.data_ptr(), cout, cin, cin_p, kt,
                                            kh, kw, 1 | frag, st), "lr_conv3d_pack_weights")
      # ... taken STRAIGHT from the pooled gradient and the window codes where the layer has the kernel: the layer
      # below only waits for this, and the un-pooled dZ (75 % zeros) then exists for the weight gradient alone
      dP_in = dP
      fused_dgrad = bool(li > 0 and coded and _FUSE_UNPOOL and frag and frag == L.lr_conv3d_dgrad_pooled_supported(
          ho, wo, cout, cin, kt, kh, kw, pt, ph, pw))
      on_side = _WGRAD_SIDE_STREAM and direct and li > 0 and (_WGRAD_SIDE_ENV == "1" or _enc._deferred)
      ready = None
      if fused_dgrad:
        if on_side and _WGRAD_EARLY:
          # everything the weight half reads (dP_in, the codes, the layer's input) and its workspace exist HERE
          ready = torch.cuda.Event()
          ready.record(torch.cuda.current_stream())
        dP = torch.empty((frames, h, w, cin), dtype=bf, device=dev)
        _C.check(L.lr_conv3d_dgrad_pooled(dP_in.data_ptr(), act.data_ptr(), wd.data_ptr(), dP.data_ptr(), B, T, ho, wo,
                                          cout, cin, kt, kh, kw, pt, ph, pw, st), "lr_conv3d_dgrad_pooled")

      # (decided — and the other path's full-resolution dZ allocated — HERE, on the stream this backward runs on: the
      # weight half may be enqueued on the side stream, whose allocator pool a per-step dZ should not come from)
      pooled_wgrad = bool(coded and _FUSE_UNPOOL_WGRAD and L.lr_conv3d_wgrad_pooled_supported_frames(
          frames, h, w, cin_p, cin, cout, kt, kh, kw, stride, pt, ph, pw) == 2)
      dZ_fallback = (torch.empty((frames, ho, wo, cout), dtype=bf, device=dev)
                     if fused_dgrad and not pooled_wgrad else None)

      def weight_half(accumulate, stream):
        # dW and the bias gradient straight from the pooled gradient and the codes where the layer has the kernel
        # (no dZ at all then) ...
        if pooled_wgrad:
          _C.check(L.lr_conv3d_wgrad_pooled(x_in.data_ptr(), pooled.data_ptr(), act.data_ptr(), dP_in.data_ptr(),
                                            grads[2 * li].data_ptr(), grads[2 * li + 1].data_ptr(), ws.data_ptr(),
                                            wbytes, accumulate, B, T, h, w, cin_p, cin, cout, kt, kh, kw, stride, pt, ph,
                                            pw, 0, stream), "lr_conv3d_wgrad_pooled")
          return None
        # ... else un-pool (the bias gradient — the sum of the routed gradients — falls out of that pass), then dW
        dZ = dZ_fallback
        if coded:
          _C.check(L.lr_unpool_code_bf16(pooled.data_ptr(), act.data_ptr(), dP_in.data_ptr(), dZ.data_ptr(),
                                         grads[2 * li + 1].data_ptr(), accumulate, ws.data_ptr(), wbytes,
                                         frames, ho, wo, cout, stream), "lr_unpool_code_bf16")
        else:
          _C.check(L.lr_unpool_relu_mask_bf16(act.data_ptr(), dP_in.data_ptr(), dZ.data_ptr(),
                                              grads[2 * li + 1].data_ptr(), accumulate, ws.data_ptr(), wbytes,
                                              frames, ho, wo, cout, stream), "lr_unpool_relu_mask_bf16")
        _C.check(L.lr_conv3d_wgrad(x_in.data_ptr(), dZ.data_ptr(), grads[2 * li].data_ptr(),
                                   None, ws.data_ptr(), wbytes, accumulate,
                                   B, T, h, w, cin_p, cin, cout, kt, kh, kw, stride, pt, ph, pw, stream),
                 "lr_conv3d_wgrad")
        return dZ

      if on_side and fused_dgrad:
        # the whole weight half (un-pooling included) runs on a side stream beside the data gradient, joined at the
        # end of this backward
        side = _conv_side_stream(dev)
        if ready is not None:
          side.wait_event(ready)
        else:
          side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
          dZ = weight_half(1, _C.stream_handle())
        side_keep.append((x_in, dZ, ws, dP_in))
        continue
      if fused_dgrad:
        weight_half(1 if direct else 0, st)
        continue
      dZ = torch.empty((frames, ho, wo, cout), dtype=bf, device=dev)
      # the bias gradient (sum of the routed gradients) falls out of the un-pooling pass
      if coded:
        _C.check(L.lr_unpool_code_bf16(pooled.data_ptr(), act.data_ptr(), dP.data_ptr(), dZ.data_ptr(),
                                       grads[2 * li + 1].data_ptr(), 1 if direct else 0, ws.data_ptr(), wbytes,
                                       frames, ho, wo, cout, st), "lr_unpool_code_bf16")
      else:
        _C.check(L.lr_unpool_relu_mask_bf16(act.data_ptr(), dP.data_ptr(), dZ.data_ptr(),
                                            grads[2 * li + 1].data_ptr(), 1 if direct else 0, ws.data_ptr(), wbytes,
                                            frames, ho, wo, cout, st), "lr_unpool_relu_mask_bf16")
      if on_side:
        # the weight gradient and the data gradient of a layer both read dZ and feed nothing to each other: the
        # weight gradient goes to a side stream (joined at the end of this backward), the data gradient — which
        # the layer below waits for — stays on this one
        side = _conv_side_stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
          _C.check(L.lr_conv3d_wgrad(x_in.data_ptr(), dZ.data_ptr(), grads[2 * li].data_ptr(),
                                     None, ws.data_ptr(), wbytes, 1,
                                     B, T, h, w, cin_p, cin, cout, kt, kh, kw, stride, pt, ph, pw, _C.stream_handle()),
                   "lr_conv3d_wgrad")
        side_keep.append((x_in, dZ, ws))
      else:
        _C.check(L.lr_conv3d_wgrad(x_in.data_ptr(), dZ.data_ptr(), grads[2 * li].data_ptr(),
                                   None, ws.data_ptr(), wbytes, 1 if direct else 0,
                                   B, T, h, w, cin_p, cin, cout, kt, kh, kw, stride, pt, ph, pw, st),
                 "lr_conv3d_wgrad")
      if li > 0:
        dP = torch.empty((frames, h, w, cin), dtype=bf, device=dev)
        _C.check(L.lr_conv3d_forward(dZ.data_ptr(), wd.data_ptr(), None, dP.data_ptr(), B, T, ho, wo, cout,
                                     cin, kt, kh, kw, 1, pt, ph, pw, frag, st), "lr_conv3d_forward(dgrad)")
    if side_keep:
      torch.cuda.current_stream().wait_stream(_conv_side_stream(dev))
      del side_keep[:]
    # the first recurrent layer's weight-gradient GEMMs ran on the side stream beside these kernels
    _enc.flush_deferred()
    if direct:
      _notify(params)
      return (None,) * (2 + len(params))
    return (None, None) + tuple(grads)


class ConvFrontend3D(nn.Module):
  """clips (B, T, 3, H, W) uint8 (scaled by 1/255 on the device) or float -> (B, T, feature_dim)."""

  def __init__(self):
    super().__init__()
    for i, (cin, cout, k, stride, pad) in enumerate(LAYERS, 1):
      setattr(self, "conv%d" % i, nn.Conv3d(cin, cout, k, stride=(1, stride, stride), padding=pad))

  def parameters_in_order(self):
    return [p for i in (1, 2, 3) for p in (getattr(self, "conv%d" % i).weight, getattr(self, "conv%d" % i).bias)]

  def forward(self, clips, out_bf16=False):
    """out_bf16: return the features as the bf16 tensor the last layer produced (no fp32 copy)."""
    _C.require_cuda(clips)
    assert clips.dim() == 5 and clips.shape[2] == 3, "clips must be (B, T, 3, H, W)"
    return _ConvFrontendFunction.apply(clips, bool(out_bf16), *self.parameters_in_order())


class PixelLipReader(nn.Module):
  """frontend + VideoEncoder: the (B,75,3,96,96) regime of BASELINE.json (build-defined)."""

  def __init__(self, encoder, frontend=None):
    super().__init__()
    self.frontend = frontend if frontend is not None else ConvFrontend3D()
    self.encoder = encoder
    self.enable_ctc = encoder.enable_ctc
    # the frontend's features are bf16 values: contract the encoder's input projections
    # (K = 3456 features) on the bf16 matrix cores with hi/lo split fp32 operands
    encoder.input_projection = 'bf16x3'
    encoder.input_is_bf16 = True
    # ... and run the recurrence of every layer that has the kernels as ONE launch per pass, fp32-FAITHFUL
    # ('split': W_hh and the state as bf16 hi + lo planes on the cluster kernels, lr_rnn_cluster.hip).  Rounds 1-4 also
    # carried a single-plane bf16 recurrence (2 % faster, 9 of 2400 greedy argmaxes flipped against the oracle at the
    # bench shape); north_star asks for identical strings, nothing shipped ran it, and round 5 removed it.
    encoder.recurrence = 'split'
    if hasattr(encoder, "attention"):
      encoder.attention = 'bf16'     # transformer encoder: fused attention on the bf16 matrix cores
    self.best_error = 1
    # weight-gradient GEMMs of upper recurrent layers overlap the recurrence of the layer below
    from . import encoder as _enc
    _enc.overlap_weight_grads = True

  def forward(self, clips, frame_lens, max_len=None, need_final_state=True):
    feats = self.frontend(clips, out_bf16=True)
    B, T, F = feats.shape
    return self.encoder(feats.reshape(B, T, F, 1), frame_lens, max_len=max_len, need_final_state=need_final_state)

  def save_best_model(self, error, file_path):
    """VideoEncoder.save_best_model's contract (better_model.py:114-122) for the whole pixel model: the
    state_dict holds `frontend.conv{1,2,3}.*` and `encoder.*`."""
    if error < self.best_error:
      self.best_error = error
      folder = os.path.dirname(file_path)
      if folder and not os.path.exists(folder):
        os.makedirs(folder)
      torch.save(self.state_dict(), file_path)
      print("\tSaving best error '{}' to '{}'".format(self.best_error, file_path))
