"""VideoEncoder on the MI355X — drop-in for `src/models/lipreader/better_model.py:13 VideoEncoder`.

Same constructor signature, attributes, forward contract and state_dict key names
(`rnn.weight_ih_l{k}[_reverse]`, `rnn.weight_hh_l{k}[_reverse]`, `rnn.bias_ih_l{k}[_reverse]`,
`rnn.bias_hh_l{k}[_reverse]`, `output_proj.{weight,bias}`), so the reference's `_init_models`
(src/scripts/train.py:57-79), `restore()` (:82-132) and `save_best_model` keep working.  The
arithmetic is the HIP path behind include/lipreading_hip.h: one fp32-MFMA input-projection GEMM,
a chain of fused recurrent step kernels with packed-sequence masking (no length sort, no
PackedSequence, no host sync), an MFMA output projection and a fused masked log-softmax.
"""
import ctypes
import math
import os

import torch
import torch.nn as nn

from . import _C
from .data import BOS, PAD

_ALLOWED_RNN_TYPES = {'LSTM', 'GRU', 'RNN'}          # better_model.py:9
_ALLOWED_FRAME_PROCESSING = {'flatten'}              # better_model.py:10
_MODES = {'GRU': 0, 'LSTM': 1, 'RNN': 2}
_PROJ_BF16X3, _INPUT_BF16_EXACT, _INPUT_STORED_BF16, _RECUR_SPLIT = 0x100, 0x200, 0x800, 0x1000   # lr_rnn_mode flags
_PROJ_BF16X1 = 0x2000
_GATES = {'GRU': 3, 'LSTM': 4, 'RNN': 1}


def _ptr_array(tensors):
  return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


# Callables invoked with the list of parameters whose .grad was just written IN PLACE by a
# backward kernel (see _direct_grads).  distributed.GradSync registers here, because autograd's
# own post-accumulate hooks do not fire for gradients that never pass through AccumulateGrad.
grad_ready_hooks = []


def _direct_grads(weights):
  """Weight gradients can be accumulated straight into existing .grad buffers (the flat
  gradient buffer of optim.FlatParameters) when every weight already has a dense fp32 .grad:
  the GEMMs then run with beta = 1 and no AccumulateGrad add kernels are launched."""
  for w in weights:
    g = w.grad
    if (g is None or not w.is_leaf or g.dtype != torch.float32 or not g.is_contiguous()
        or g.shape != w.shape or g.device != w.device):
      return False
  return True


def _notify(weights):
  for hook in grad_ready_hooks:
    hook(list(weights))


# Callables invoked (on the stream that produced them) when the LAST gradients of the encoder have been enqueued: the
# weight half of the layer that reads the model's input features (optim.FusedAdam.sum_squares_early registers here).
# Argument: one parameter tensor of that layer, so that a hook can tell whose backward it is.
encoder_grads_complete_hooks = []


# Weight-gradient halves of layer backwards that were put on the side stream (see
# _RNNLayerFunction.backward): the tensors they use, kept alive until flush_deferred() joins the streams.
overlap_weight_grads = False     # PixelLipReader switches it on; needs in-place (.grad) gradients
_side_stream = None
_deferred = []
# (Measured and dropped, round 5: starting an upper layer's weight half BEHIND the lower layer's recurrence — which it
# slows from 131 to 163 us at the bench shape — instead of beside it: 2.500 against 2.517 / 2.477 ms for the default in
# the same visit, 2.393 against 2.397 on top of the lr_fgemm dW_ih: inside the noise.  profiles/r05_variants_ab.txt.)


def _get_side_stream(device):
  global _side_stream
  if _side_stream is None or _side_stream.device != device:
    _side_stream = torch.cuda.Stream(device=device)
  return _side_stream


# other modules' deferred halves on streams of their own (attention_decoder: the loop's weight half): callables that make
# the current stream wait for theirs
deferred_flushers = []


def flush_deferred():
  """Join the side stream: the current stream waits for every deferred weight-gradient half.  Called
  at the end of the next layer's backward, by the conv frontend's backward and by FusedAdam.step; a
  no-op when nothing is pending."""
  for f in deferred_flushers:
    f()
  if not _deferred:
    return
  torch.cuda.current_stream().wait_stream(_side_stream)
  del _deferred[:]   # the tensors the side stream was using may be released now


# (rnn_type, H, why) triples whose fall-back to the per-step kernels has been announced (one warning each)
_fallback_noted = set()
_WHY = {
    1: "there is no one-launch kernel for this shape (GRU / LSTM with a hidden size that is a multiple of 4: up to 1152 "
       "units on clusters, lr_rnn_cluster.hip; LSTM up to 1536 on the 192-CU grid, lr_rnn_grid.hip)",
    2: "the one-launch recurrences were switched off for this process (lr_rnn_one_launch_enable(0): lipreading_amd.train "
       "does that after repeated time-outs, see its own warning)",
    3: "the test hook lr_rnn_debug_disable_cluster is set",
    4: "this device has too few compute units for a launch's clusters, which must be resident together",
}


def _note_step_kernel_fallback(rnn_type, H, why):
  """Says ONCE per (cell, size, reason) that a layer runs one launch per time step, and WHY (lr_rnn_one_launch_status):
  an unsupported shape, the product's switch after time-outs, the test hook, or a small device — and, under
  torch.distributed, on which rank."""
  if (rnn_type, H, why) in _fallback_noted:
    return
  _fallback_noted.add((rnn_type, H, why))
  import warnings
  rank = ""
  try:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
      rank = " [rank %d of %d]" % (dist.get_rank(), dist.get_world_size())
  except Exception:
    pass
  warnings.warn("lipreading_amd%s: %s-%d runs one launch per time step (2-5x slower per pass, same results): %s"
                % (rank, rnn_type, H, _WHY.get(why, "status %d" % why)), stacklevel=3)


class _RNNLayerFunction(torch.autograd.Function):
  """One (bi)directional layer: lr_rnn_layer_forward / lr_rnn_layer_backward."""

  @staticmethod
  def forward(ctx, x, lens, mode, H, need_dx, need_state, *weights):
    L = _C.lib()
    B, T, I = x.shape
    D = len(weights) // 4
    dev = x.device
    w_ih, w_hh = list(weights[0::4]), list(weights[1::4])
    b_ih, b_hh = list(weights[2::4]), list(weights[3::4])
    # unused outputs come back to backward() as None, not as zero tensors (a fill launch each)
    ctx.set_materialize_grads(False)
    y = torch.empty((B, T, D * H), dtype=torch.float32, device=dev)
    # need_state False: the final states are not extracted (h_n == NULL; the CTC-only step never reads them)
    h_n = torch.empty((D, B, H) if need_state else (0,), dtype=torch.float32, device=dev)
    c_n = torch.empty((D, B, H), dtype=torch.float32, device=dev) if ((mode & 0xff) == 1 and need_state) else None
    rbytes = L.lr_rnn_reserve_bytes(mode, B, T, I, H, D)
    reserve = torch.empty(rbytes, dtype=torch.uint8, device=dev)
    _C.check(L.lr_rnn_layer_forward(mode, x.data_ptr(), lens.data_ptr(), _ptr_array(w_ih),
                                    _ptr_array(w_hh), _ptr_array(b_ih), _ptr_array(b_hh),
                                    y.data_ptr(), h_n.data_ptr() if need_state else None, _C.ptr(c_n), reserve.data_ptr(),
                                    rbytes, B, T, I, H, D, _C.stream_handle()),
             "lr_rnn_layer_forward")
    ctx.save_for_backward(x, lens, y, reserve, *weights)
    ctx.cfg = (mode, H, D, need_dx)
    if c_n is None:
      c_n = torch.empty((0,), device=dev)
      ctx.mark_non_differentiable(c_n)
    if not need_state:
      ctx.mark_non_differentiable(h_n)
    return y, h_n, c_n

  @staticmethod
  def backward(ctx, dy, dh_n, dc_n):
    x, lens, y, reserve = ctx.saved_tensors[:4]
    weights = ctx.saved_tensors[4:]
    mode, H, D, need_dx = ctx.cfg
    L = _C.lib()
    B, T, I = x.shape
    dev = x.device
    w_ih, w_hh = list(weights[0::4]), list(weights[1::4])
    b_ih, b_hh = list(weights[2::4]), list(weights[3::4])
    dy = dy.contiguous() if dy is not None else torch.zeros_like(y)
    dh_n = dh_n.contiguous() if dh_n is not None else None
    dc_n = dc_n.contiguous() if ((mode & 0xff) == 1 and dc_n is not None) else None
    direct = _direct_grads(weights)
    grads = [w.grad for w in weights] if direct else [torch.empty_like(w) for w in weights]
    dx = torch.empty_like(x) if need_dx else None
    wbytes = L.lr_rnn_workspace_bytes(mode, B, T, I, H, D)
    ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
    if overlap_weight_grads and direct and need_dx and (mode & _PROJ_BF16X3):
      # An upper layer of the pixel regime: the layer below only waits for dx.  The recurrence and dx
      # stay on this stream; the weight-gradient GEMMs go to a side stream, where they overlap the
      # layer below's recurrence (one workgroup per sample and direction: most of the chip is idle).
      args = (mode, x.data_ptr(), lens.data_ptr(), _ptr_array(w_ih), _ptr_array(w_hh), _ptr_array(b_ih),
              _ptr_array(b_hh), y.data_ptr(), dy.data_ptr(), _C.ptr(dh_n), _C.ptr(dc_n), _C.ptr(dx),
              _ptr_array(grads[0::4]), _ptr_array(grads[1::4]), _ptr_array(grads[2::4]), _ptr_array(grads[3::4]),
              reserve.data_ptr(), reserve.numel(), ws.data_ptr(), wbytes, 1, B, T, I, H, D)
      _C.check(L.lr_rnn_layer_backward_parts(*args, 1, _C.stream_handle()), "lr_rnn_layer_backward_parts(data)")

      def weight_half():
        _C.check(L.lr_rnn_layer_backward_parts(*args, 2, _C.stream_handle()), "lr_rnn_layer_backward_parts(weights)")
        # announced from the side stream: a gradient all-reduce (distributed.GradSync) then waits for
        # THIS stream, i.e. starts as soon as these gradients exist, not when the streams are joined
        _notify(weights)
      flush_deferred()                      # at most one deferred half in flight
      _deferred.append((x, lens, y, dy, reserve, ws, grads, weights, args))
      side = _get_side_stream(dev)
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):
        weight_half()
        if mode & _INPUT_STORED_BF16:         # (the layer that reads the frontend's features: the encoder's last gradients)
          for hook in encoder_grads_complete_hooks:
            hook(weights[0])   # (which encoder: a parameter of the layer that just finished)
      return (dx, None, None, None, None, None) + (None,) * len(weights)
    _C.check(L.lr_rnn_layer_backward(
        mode, x.data_ptr(), lens.data_ptr(), _ptr_array(w_ih), _ptr_array(w_hh), _ptr_array(b_ih),
        _ptr_array(b_hh), y.data_ptr(), dy.data_ptr(), _C.ptr(dh_n), _C.ptr(dc_n), _C.ptr(dx),
        _ptr_array(grads[0::4]), _ptr_array(grads[1::4]), _ptr_array(grads[2::4]),
        _ptr_array(grads[3::4]), reserve.data_ptr(), reserve.numel(), ws.data_ptr(), wbytes,
        1 if direct else 0, B, T, I, H, D, _C.stream_handle()), "lr_rnn_layer_backward")
    flush_deferred()   # the layer above's weight-gradient half overlapped this layer's recurrence
    if direct:
      _notify(weights)
      return (dx, None, None, None, None, None) + (None,) * len(weights)
    return (dx, None, None, None, None, None) + tuple(grads)


class _DropoutFunction(torch.autograd.Function):
  """nn.GRU / nn.LSTM(dropout=p) between stacked layers, training mode: lr_dropout_forward (Philox mask, seed drawn from
  torch's host generator) and the same multiply on the way back."""

  @staticmethod
  def forward(ctx, x, p, seed):
    x = x.contiguous()
    y, mask = torch.empty_like(x), torch.empty_like(x)
    _C.check(_C.lib().lr_dropout_forward(x.data_ptr(), y.data_ptr(), mask.data_ptr(), x.numel(), float(p), int(seed),
                                         _C.stream_handle()), "lr_dropout_forward")
    ctx.save_for_backward(mask)
    return y

  @staticmethod
  def backward(ctx, dy):
    mask, = ctx.saved_tensors
    dy = dy.contiguous()
    dx = torch.empty_like(dy)
    _C.check(_C.lib().lr_mul_f32(dy.data_ptr(), mask.data_ptr(), dx.data_ptr(), dy.numel(), _C.stream_handle()), "lr_mul_f32")
    return dx, None, None


class _CatDirectionsFunction(torch.autograd.Function):
  """(D,B,H) -> (B, D*H), forward direction first (better_model.py:98-112 _cat_directions), for h (and c) of a layer in
  ONE launch each way (torch: permute + reshape copies per tensor, and their gradients' copies back)."""

  @staticmethod
  def forward(ctx, h, c):
    D, B, H = h.shape
    h = h.contiguous()
    out_h = torch.empty((B, D * H), dtype=h.dtype, device=h.device)
    out_c = None
    if c is not None:
      c = c.contiguous()
      out_c = torch.empty_like(out_h)
    _C.check(_C.lib().lr_cat_directions(h.data_ptr(), out_h.data_ptr(), _C.ptr(c), _C.ptr(out_c), B, H, D, 0,
                                        _C.stream_handle()), "lr_cat_directions")
    ctx.dims = (D, B, H, c is not None)
    ctx.set_materialize_grads(False)
    if c is None:
      return out_h
    return out_h, out_c

  @staticmethod
  def backward(ctx, gh, gc=None):
    D, B, H, has_c = ctx.dims
    if gh is None and gc is None:
      return None, None
    ref = gh if gh is not None else gc
    gh = gh.contiguous() if gh is not None else torch.zeros_like(ref)
    if has_c:
      gc = gc.contiguous() if gc is not None else torch.zeros_like(ref)
    dh = torch.empty((D, B, H), dtype=ref.dtype, device=ref.device)
    dc = torch.empty_like(dh) if has_c else None
    _C.check(_C.lib().lr_cat_directions(gh.data_ptr(), dh.data_ptr(), _C.ptr(gc) if has_c else None, _C.ptr(dc), B, H, D, 1,
                                        _C.stream_handle()), "lr_cat_directions")
    return dh, dc


class _ProjLogSoftmaxFunction(torch.autograd.Function):
  """output_proj + masked_log_softmax (better_model.py:92-93)."""

  @staticmethod
  def forward(ctx, hidden, weight, bias, mask):
    L = _C.lib()
    B, T, K = hidden.shape
    C = weight.shape[0]
    R = B * T
    lp = torch.empty((B, T, C), dtype=torch.float32, device=hidden.device)
    wbytes = L.lr_proj_workspace_bytes(R, K, C)
    ws = torch.empty(wbytes, dtype=torch.uint8, device=hidden.device)
    _C.check(L.lr_proj_logsoftmax_forward(hidden.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                          mask.data_ptr(), lp.data_ptr(), ws.data_ptr(), wbytes, R,
                                          K, C, _C.stream_handle()), "lr_proj_logsoftmax_forward")
    ctx.save_for_backward(hidden, weight, lp, bias)
    return lp

  @staticmethod
  def backward(ctx, g):
    hidden, weight, lp, bias = ctx.saved_tensors
    L = _C.lib()
    B, T, K = hidden.shape
    C = weight.shape[0]
    R = B * T
    dev = hidden.device
    g = g.contiguous()
    dlogits = torch.empty((R, C), dtype=torch.float32, device=dev)
    dhidden = torch.empty_like(hidden) if ctx.needs_input_grad[0] else None
    direct = _direct_grads((weight, bias))
    dW = weight.grad if direct else torch.empty_like(weight)
    db = bias.grad if direct else torch.empty((C,), dtype=torch.float32, device=dev)
    wbytes = L.lr_proj_workspace_bytes(R, K, C)
    ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
    _C.check(L.lr_proj_logsoftmax_backward(g.data_ptr(), lp.data_ptr(), hidden.data_ptr(),
                                           weight.data_ptr(), dlogits.data_ptr(), _C.ptr(dhidden),
                                           dW.data_ptr(), db.data_ptr(), ws.data_ptr(), wbytes,
                                           1 if direct else 0, R, K, C, _C.stream_handle()),
             "lr_proj_logsoftmax_backward")
    if direct:
      _notify((weight, bias))
      return dhidden, None, None, None
    return dhidden, dW, db, None


class _RNNParams(nn.Module):
  """Parameter container with torch.nn.RNNBase's names, shapes, registration order and
  initialisation (uniform(-1/sqrt(H), 1/sqrt(H)) drawn in registration order), so that a seed
  produces the same initial weights as the reference's `getattr(nn, rnn_type)(...)`
  (better_model.py:47-49) and `state_dict()` keys are identical."""

  def __init__(self, rnn_type, input_size, hidden_size, num_layers, bidirectional):
    super().__init__()
    G = _GATES[rnn_type]
    D = 2 if bidirectional else 1
    self.names = []
    for layer in range(num_layers):
      layer_in = input_size if layer == 0 else hidden_size * D
      for d in range(D):
        suffix = '_reverse' if d == 1 else ''
        shapes = [('weight_ih', (G * hidden_size, layer_in)), ('weight_hh', (G * hidden_size, hidden_size)),
                  ('bias_ih', (G * hidden_size,)), ('bias_hh', (G * hidden_size,))]
        for base, shape in shapes:
          name = '%s_l%d%s' % (base, layer, suffix)
          self.register_parameter(name, nn.Parameter(torch.empty(shape)))
          self.names.append(name)
    stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
    for p in self.parameters():
      nn.init.uniform_(p, -stdv, stdv)

  def layer_weights(self, layer, D):
    return [getattr(self, self.names[(layer * D + d) * 4 + i]) for d in range(D) for i in range(4)]


class VideoEncoder(nn.Module):
  def __init__(self, frame_dim, hidden_size, frame_processing='flatten',
               rnn_type='LSTM', num_layers=1, bidirectional=True, rnn_dropout=0,
               enable_ctc=False, vocab_size=-1, char2idx=None, device="cpu"):
    """When enable_ctc=True, vocab_size (including the special tokens) and char2idx must be
    provided — better_model.py:14-25."""
    super(VideoEncoder, self).__init__()
    assert frame_processing in _ALLOWED_FRAME_PROCESSING
    assert rnn_type in _ALLOWED_RNN_TYPES
    if enable_ctc:
      assert vocab_size > 0 and char2idx is not None

    self.frame_dim = frame_dim
    self.hidden_size = hidden_size
    self.frame_processing = frame_processing
    self.rnn_type = rnn_type
    self.num_layers = num_layers
    self.bidirectional = bidirectional
    self.rnn_dropout = rnn_dropout
    self.enable_ctc = enable_ctc
    self.best_error = 1
    self.num_dirs = 2 if self.bidirectional else 1
    # not in the reference: 'f32' = exact fp32 MFMA input projection (reference-faithful regime);
    # 'bf16x3' is set by frontend.PixelLipReader for the build-defined pixel regime
    self.input_projection = 'f32'
    self.input_is_bf16 = False
    # how the T-step recurrence of a layer runs (not in the reference):
    #   'f32'    one launch per time step, exact fp32 MFMA — every shape
    #   'split'  ONE launch per layer pass, fp32-faithful (W_hh and the state as bf16 hi + lo planes, ~1e-6
    #            of the fp32 product): a cluster of ceil(H / 32) CUs per (direction, 8 samples) (lr_rnn_cluster.hip) —
    #            every H the clusters hold (16-unit members past 864 / 768); LSTM past 1152 up to 1536 units (the
    #            decoders behind BiLSTM-700 / 768) on one 24 x 8 grid of 192 CUs (lr_rnn_grid.hip); larger layers as
    #            'f32' (announced once)
    #   'auto'   (default) same as 'split': reference-faithful numerics at the one-launch speed
    self.recurrence = 'auto'
    if self.enable_ctc:
      self.vocab_size = vocab_size
      self.adj_vocab_size = self.vocab_size + 1      # idx 0 is reserved for the CTC blank
      self.char2idx = char2idx
      mask = torch.ones(self.adj_vocab_size, device=device)
      mask[self.char2idx[PAD] + 1] = 0
      mask[self.char2idx[BOS] + 1] = 0
      # the reference keeps a plain attribute (not in state_dict); a non-persistent buffer has
      # the same state_dict and additionally follows .to(device)
      self.register_buffer("output_mask", mask, persistent=False)

    self.rnn = _RNNParams(self.rnn_type, self.frame_dim, self.hidden_size, self.num_layers,
                          self.bidirectional)
    if self.enable_ctc:
      self.output_proj = nn.Linear(self.num_dirs * self.hidden_size, self.adj_vocab_size)

  def forward(self, frames, frame_lens, max_len=None, need_final_state=True, head_stream=None):
    """frames (B, seq_len, num_lmks, lmk_dim) f32, frame_lens (B,) -> as better_model.py:53-96:
    (log_probs (B,Tmax,V+1), hidden (B,Tmax,D*H), final_state) if enable_ctc else
    (hidden, final_state); final_state is (h, c) for the LSTM, each (layers, B, D*H).

    `max_len` (optional, not in the reference) = max(frame_lens) when the caller already knows
    it on the host; it avoids the one device->host read that the reference also performs
    (better_model.py:69).  `head_stream` (not in the reference): the CTC head runs on that stream (it waits for this
    one); the caller joins before it reads log_probs on another stream (train.decoder_step: the head and the CTC loss
    beside the decoder loop)."""
    _C.require_cuda(frames)
    if self.frame_processing == 'flatten':
      frames = frames.reshape(frames.shape[0], frames.shape[1], -1)
    B, T, I = frames.shape
    assert I == self.frame_dim
    if max_len is None:
      max_len = int(frame_lens.max())          # host read iff frame_lens lives on the device
    assert 1 <= max_len <= T
    # the pixel regime hands the frontend's bf16 features over as they are (LR_RNN_INPUT_STORED_BF16)
    stored_bf16 = (frames.dtype == torch.bfloat16 and self.input_projection in ('bf16x3', 'bf16x1') and self.input_is_bf16
                   and I % 8 == 0)
    x = frames[:, :max_len].contiguous() if stored_bf16 else frames[:, :max_len].to(torch.float32).contiguous()
    lens = frame_lens.to(device=x.device, dtype=torch.int32).contiguous()
    mode, H, D = _MODES[self.rnn_type], self.hidden_size, self.num_dirs

    h_fin, c_fin = [], []
    for layer in range(self.num_layers):
      weights = self.rnn.layer_weights(layer, D)
      need_dx = layer > 0 or x.requires_grad
      lmode = mode
      if self.input_projection in ('bf16x3', 'bf16x1'):
        # build-defined (pixel regime): input projection on the bf16 matrix cores with hi/lo split
        # operands (include/lipreading_hip.h LR_RNN_PROJ_BF16X3); layer 0's input is bf16-exact
        # when it comes from the bf16 conv frontend
        lmode |= _PROJ_BF16X3 | (_INPUT_BF16_EXACT if (layer == 0 and self.input_is_bf16) else 0)
        if self.input_projection == 'bf16x1':   # experiment: one bf16 product per GEMM (LR_RNN_PROJ_BF16X1)
          lmode |= _PROJ_BF16X1
        if layer == 0 and stored_bf16:
          lmode |= _INPUT_STORED_BF16
      assert self.recurrence in ('auto', 'f32', 'split'), self.recurrence
      if self.recurrence in ('auto', 'split') and mode != 2:
        why = _C.lib().lr_rnn_one_launch_status(mode, B, max_len, x.shape[2], H, D)
        if why == 0:
          lmode |= _RECUR_SPLIT
        else:
          _note_step_kernel_fallback(self.rnn_type, H, why)
      y, h_n, c_n = _RNNLayerFunction.apply(x, lens, lmode, H, need_dx, need_final_state, *weights)
      if need_final_state:
        # (D,B,H) -> (B, D*H): forward direction first, as _cat_directions (better_model.py:98-112); h and c in one launch
        if mode == 1:
          hf, cf = _CatDirectionsFunction.apply(h_n, c_n)
          c_fin.append(cf)
        else:
          hf = _CatDirectionsFunction.apply(h_n, None)
        h_fin.append(hf)
      x = y
      if self.rnn_dropout and self.training and layer + 1 < self.num_layers:
        # nn.GRU/LSTM inter-layer dropout; the seed comes from torch's host generator (no device work)
        x = _DropoutFunction.apply(x, float(self.rnn_dropout), int(torch.randint(0, 2 ** 62, (1,)).item()))
    hidden_states = y
    final_state = None
    if need_final_state:
      # (layers, B, D*H): one layer (every shipped config) is a view, more are stacked
      final_state = h_fin[0].unsqueeze(0) if len(h_fin) == 1 else torch.stack(h_fin, 0)
      if mode == 1:
        final_state = (final_state, c_fin[0].unsqueeze(0) if len(c_fin) == 1 else torch.stack(c_fin, 0))

    if self.enable_ctc:
      if head_stream is not None:
        head_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(head_stream):
          output_log_probs = _ProjLogSoftmaxFunction.apply(hidden_states, self.output_proj.weight,
                                                           self.output_proj.bias, self.output_mask)
      else:
        output_log_probs = _ProjLogSoftmaxFunction.apply(hidden_states, self.output_proj.weight,
                                                         self.output_proj.bias, self.output_mask)
      return output_log_probs, hidden_states, final_state
    return hidden_states, final_state

  def save_best_model(self, error, file_path):
    """better_model.py:114-122."""
    if error < self.best_error:
      self.best_error = error
      folder = os.path.dirname(file_path)
      if folder and not os.path.exists(folder):
        os.makedirs(folder)
      torch.save(self.state_dict(), file_path)
      print("\tSaving best error '{}' to '{}'".format(self.best_error, file_path))
