"""Attention character decoder on the MI355X — drop-in for
`src/models/lipreader/better_model.py:124 CharDecodingStep` plus the fused decoder loop of
`src/train/train_better_model.py:54-65`.

`CharDecodingStep` keeps the reference's constructor signature, attributes and state_dict keys
(`embedding.weight`, `rnn.weight_ih_l0` …, `attn_proj_*`, `concat_layer.*`, `output_proj.*`; same
creation order, so the same seed gives the same initial weights).  `forward(input_, previous_state,
encoder_lens, encoder_hidden_states)` is the reference's single step; `decode_sequence(...)` runs
all `max_label_len` steps in ONE call into the C ABI (lr_decoder_forward / lr_decoder_backward):
the loop is a strictly sequential chain of small kernels, so the host should not sit between them.
"""
import ctypes
import os

import torch
import torch.nn as nn

from . import _C
from .data import BOS, PAD
from .encoder import _RNNParams, _direct_grads, _notify

_ALLOWED_ATTENTION_TYPES = {'none', 'dot', 'general', '1_layer_nn', 'concat'}   # better_model.py:11
_ATT_CODE = {'none': 0, 'dot': 1, 'general': 2, '1_layer_nn': 3, 'concat': 4}
_MODES = {'GRU': 0, 'LSTM': 1, 'RNN': 2}
_FIELDS = ("emb", "w_ih", "w_hh", "b_ih", "b_hh", "attn_w1", "attn_b1", "attn_w2", "attn_b2", "w_c", "b_c",
           "w_o", "b_o")


def _upper_struct(upper, drop_mask):
  """lr_decoder_upper for the layers above the first (upper = flat list w_ih, w_hh, b_ih, b_hh per layer)."""
  nl = 1 + len(upper) // 4
  if nl == 1:
    return None, nl
  st = _C.DecoderUpper()
  st.num_layers = nl
  for k in range(nl - 1):
    for j, name in enumerate(("w_ih", "w_hh", "b_ih", "b_hh")):
      getattr(st, name)[k] = upper[4 * k + j].data_ptr()
  st.drop_mask = _C.ptr(drop_mask)
  return st, nl


_STEP_LENS = {}


def _step_lens(B, L, dev):
  """The (B,) int32 vector of L the C ABI takes as `step_lens`: constant per (B, L, device), so it is built once (a
  fill launch per step otherwise).  Never written by a kernel."""
  key = (B, L, dev.type, dev.index)
  t = _STEP_LENS.get(key)
  if t is None:
    if len(_STEP_LENS) > 64:
      _STEP_LENS.clear()
    t = _STEP_LENS[key] = torch.full((B,), L, dtype=torch.int32, device=dev)
  return t


# test hook / switch: False keeps the loop's weight half on the stream of its data half
overlap_weight_half = True
# The stream whose backward pass put the loop's weight half on the side stream, until train._step_all takes it (or the
# next loop starts): the decoder's optimiser may follow the weight half there only in THAT pass, from THAT stream.
_split_from = None


# The loop's weight half runs on a stream of ITS OWN, not on encoder._side_stream: with the encoder module's side
# stream forked into the captured decoder steps as well as into the captured pixel steps, the -m gpu suite (some forty
# captured graphs in one process) died inside hipGraphLaunch of a later pixel-step graph — reproducibly, in one order of
# the test files; with this stream it does not (round 6, visits r06H .. r06X; the runtime-side cause was not found).
_side = None
_deferred = []     # what the half in flight reads, kept alive until the streams are joined


def side_stream(device=None):
  global _side
  if device is not None and (_side is None or _side.device != device):
    _side = torch.cuda.Stream(device=device)
  return _side


def flush(into=None):
  """`into` (default: the current stream) waits for the loop's deferred weight half; a no-op when none is pending."""
  if _deferred:
    (into if into is not None else torch.cuda.current_stream()).wait_stream(_side)
    del _deferred[:]


def _register():
  from . import encoder as _enc
  if flush not in _enc.deferred_flushers:      # FusedAdam.step and the encoder layers' backward join through there
    _enc.deferred_flushers.append(flush)


_register()


def take_split_flag():
  """True once, if the backward pass that just ended on the current stream split the loop's backward."""
  global _split_from
  s, _split_from = _split_from, None
  return s is not None and s == torch.cuda.current_stream()


class _AttnDecoderFunction(torch.autograd.Function):
  """All L decoder steps: (tokens, teacher-forcing pattern, encoder states, initial state, params)
  -> (log_probs (B,L,V), sampled (B,L), h_n, c_n).  `params` = the 13 tensors of lr_decoder_params
  (None where the attention type has none) followed by 4 tensors per RNN layer above the first."""

  @staticmethod
  def forward(ctx, tokens, teacher_forced, seed, mode, attn_type, attn_hidden, enc, enc_lens, h0, c0,
              out_mask, drop_mask, *params):
    L_ = _C.lib()
    global _split_from
    _split_from = None     # (a flag nobody took — a loop stepped outside train._step_all — does not outlive its step)
    B, L = tokens.shape
    T, Hd = enc.shape[1], enc.shape[2]
    params, upper = params[:13], params[13:]
    emb = params[0]
    V, Cd = emb.shape
    A = max(int(attn_hidden), 0)
    dev = enc.device
    pstruct = _C.DecoderParams(*[_C.ptr(p) for p in params], out_mask.data_ptr())
    ustruct, NL = _upper_struct(upper, drop_mask)
    assert h0.shape == (NL, B, Hd)
    lp = torch.empty((B, L, V), dtype=torch.float32, device=dev)
    sampled = torch.empty((B, L), dtype=torch.int32, device=dev)
    h_n = torch.empty((NL, B, Hd), dtype=torch.float32, device=dev)
    c_n = torch.empty((NL, B, Hd), dtype=torch.float32, device=dev) if mode == 1 else None
    step_lens = _step_lens(B, L, dev)
    rbytes = L_.lr_decoder_reserve_bytes(mode, attn_type, NL, B, L, T, Hd, Cd, V, A)
    reserve = torch.empty(rbytes, dtype=torch.uint8, device=dev)
    tf = (ctypes.c_uint8 * L)(*[1 if f else 0 for f in teacher_forced])
    _C.check(L_.lr_decoder_forward(mode, attn_type, ctypes.byref(pstruct),
                                   ctypes.byref(ustruct) if ustruct is not None else None, tokens.data_ptr(), tf,
                                   enc.data_ptr(), enc_lens.data_ptr(), h0.data_ptr(), _C.ptr(c0),
                                   step_lens.data_ptr(), int(seed) & 0xFFFFFFFFFFFFFFFF, lp.data_ptr(),
                                   sampled.data_ptr(), h_n.data_ptr(), _C.ptr(c_n), reserve.data_ptr(), rbytes,
                                   B, L, T, Hd, Cd, V, A, _C.stream_handle()), "lr_decoder_forward")
    ctx.save_for_backward(enc, enc_lens, h0, c0 if c0 is not None else h0, out_mask, lp, reserve, step_lens,
                          drop_mask if drop_mask is not None else out_mask,
                          *[p for p in params if p is not None], *upper)
    ctx.present = [p is not None for p in params]
    ctx.cfg = (mode, attn_type, A, B, L, T, Hd, Cd, V, c0 is not None, NL, drop_mask is not None)
    ctx.mark_non_differentiable(sampled)
    if c_n is None:
      c_n = torch.zeros((0,), device=dev)
      ctx.mark_non_differentiable(c_n)
    return lp, sampled, h_n, c_n

  @staticmethod
  def backward(ctx, d_lp, _ds, dh_n, dc_n):
    L_ = _C.lib()
    mode, attn_type, A, B, L, T, Hd, Cd, V, has_c, NL, has_drop = ctx.cfg
    saved = ctx.saved_tensors
    enc, enc_lens, h0, c0, out_mask, lp, reserve, step_lens, drop_mask = saved[:9]
    if not has_drop:
      drop_mask = None
    present = ctx.present
    rest = list(saved[9:])
    n_real = sum(present)
    it = iter(rest[:n_real])
    params = [next(it) if pr else None for pr in present]
    upper = rest[n_real:]
    if not has_c:
      c0 = None
    dev = enc.device
    real = [p for p in params if p is not None] + list(upper)
    direct = _direct_grads(real)
    grads = [(p.grad if direct else torch.empty_like(p)) if p is not None else None for p in params]
    ugrads = [(p.grad if direct else torch.empty_like(p)) for p in upper]
    pstruct = _C.DecoderParams(*[_C.ptr(p) for p in params], out_mask.data_ptr())
    ustruct, _ = _upper_struct(upper, drop_mask)
    pad_idx = getattr(params[0], "_lr_padding_idx", -1)
    gstruct = _C.DecoderGrads(*[_C.ptr(g) for g in grads], int(pad_idx))
    gustruct = None
    if NL > 1:
      gustruct = _C.DecoderUpperGrads()
      for k in range(NL - 1):
        for j, name in enumerate(("w_ih", "w_hh", "b_ih", "b_hh")):
          getattr(gustruct, name)[k] = ugrads[4 * k + j].data_ptr()
    d_lp = (d_lp if d_lp is not None else torch.zeros_like(lp)).contiguous()
    dh_n = dh_n.contiguous() if dh_n is not None else None
    dc_n = dc_n.contiguous() if (has_c and dc_n is not None and dc_n.numel()) else None
    d_enc = torch.empty_like(enc)
    dh0 = torch.empty_like(h0)
    dc0 = torch.empty_like(h0) if has_c else None
    wbytes = L_.lr_decoder_workspace_bytes(mode, attn_type, NL, B, L, T, Hd, Cd, V, A)
    ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
    args = (mode, attn_type, ctypes.byref(pstruct), ctypes.byref(ustruct) if ustruct is not None else None,
            ctypes.byref(gstruct), ctypes.byref(gustruct) if gustruct is not None else None,
            enc.data_ptr(), enc_lens.data_ptr(), h0.data_ptr(), _C.ptr(c0), step_lens.data_ptr(), lp.data_ptr(),
            d_lp.data_ptr(), _C.ptr(dh_n), _C.ptr(dc_n), d_enc.data_ptr(), dh0.data_ptr(), _C.ptr(dc0),
            reserve.data_ptr(), reserve.numel(), ws.data_ptr(), wbytes, 1 if direct else 0, B, L, T, Hd, Cd, V, A)
    if overlap_weight_half and direct and L_.lr_decoder_backward_splittable(attn_type, NL):
      # The encoder's backward only waits for dh0 / dc0 (and d_enc): the data half stays on this stream, every parameter
      # gradient of the loop goes to this module's side stream and runs beside the head's and the encoder's backward —
      # whose recurrence leaves a quarter of the chip idle.  Round 6: the ecd family's step 1.81 -> 1.66 ms.
      _C.check(L_.lr_decoder_backward_parts(*args, 1, _C.stream_handle()), "lr_decoder_backward_parts(data)")
      flush()                                # at most one deferred half in flight
      forked_from = torch.cuda.current_stream()
      side = side_stream(dev)
      side.wait_stream(forked_from)
      with torch.cuda.stream(side):
        _C.check(L_.lr_decoder_backward_parts(*args, 2, _C.stream_handle()), "lr_decoder_backward_parts(weights)")
        _notify(real)
      _deferred.append((enc, enc_lens, h0, c0, step_lens, lp, d_lp, dh_n, dc_n, reserve, ws, grads, ugrads, real,
                        pstruct, ustruct, gstruct, gustruct, out_mask, drop_mask))
      # whoever reads the gradients after loss.backward() — an optimiser, a test, .cpu() — finds them complete: the
      # streams are joined when this backward pass ends at the latest.  (The callback may run on an engine thread whose
      # current stream is not this one: it joins the stream the fork came from.)
      torch.autograd.Variable._execution_engine.queue_callback(lambda: flush(forked_from))
      global _split_from
      _split_from = forked_from
      return (None, None, None, None, None, None, d_enc, None, dh0, dc0, None, None) + (None,) * (len(params) + len(upper))
    _C.check(L_.lr_decoder_backward_parts(*args, 3, _C.stream_handle()), "lr_decoder_backward")
    if direct:
      _notify(real)
      pgrads = (None,) * (len(params) + len(upper))
    else:
      pgrads = tuple(grads) + tuple(ugrads)
    return (None, None, None, None, None, None, d_enc, None, dh0, dc0, None, None) + pgrads


class CharDecodingStep(nn.Module):
  def __init__(self, encoder, char_dim, vocab_size, char2idx, rnn_dropout=0, attention_type='none',
               attn_hidden_size=-1, device="cpu"):
    """vocab_size includes all the special tokens — better_model.py:125-159."""
    super(CharDecodingStep, self).__init__()
    assert attention_type in _ALLOWED_ATTENTION_TYPES
    if attention_type == 'concat':
      assert attn_hidden_size > 0
    self.hidden_size = encoder.hidden_size * (2 if encoder.bidirectional else 1)
    self.rnn_type = encoder.rnn_type
    self.num_layers = encoder.num_layers          # better_model.py:136
    assert 1 <= self.num_layers <= _C.DEC_MAX_LAYERS, "the decoder stack holds up to %d layers" % _C.DEC_MAX_LAYERS
    self.rnn_dropout = rnn_dropout
    self.char_dim = char_dim
    self.vocab_size = vocab_size
    self.char2idx = char2idx
    self.attention_type = attention_type
    self.attn_hidden_size = attn_hidden_size

    mask = torch.ones(self.vocab_size, device=device)
    mask[self.char2idx[PAD]] = 0
    mask[self.char2idx[BOS]] = 0
    self.register_buffer("output_mask", mask, persistent=False)   # plain attribute in the reference
    # creation order = the reference's (same seed -> same initial weights)
    self.embedding = nn.Embedding(self.vocab_size, self.char_dim, padding_idx=self.char2idx[PAD])
    self.rnn = _RNNParams(self.rnn_type, self.char_dim, self.hidden_size, self.num_layers, False)
    if attention_type == '1_layer_nn':
      self.attn_proj_1_layer_nn = nn.Linear(2 * self.hidden_size, 1)
    elif attention_type == 'general':
      self.attn_proj_general = nn.Linear(self.hidden_size, self.hidden_size)
    elif attention_type == 'concat':
      self.attn_proj_layer1 = nn.Linear(2 * self.hidden_size, attn_hidden_size)
      self.attn_proj_layer2 = nn.Linear(attn_hidden_size, 1)
    self.concat_layer = nn.Linear(2 * self.hidden_size, self.hidden_size)
    self.output_proj = nn.Linear(self.hidden_size, self.vocab_size)
    self.best_error = 1

  def _params(self):
    at = self.attention_type
    w1 = b1 = w2 = b2 = None
    if at == '1_layer_nn':
      w1, b1 = self.attn_proj_1_layer_nn.weight, self.attn_proj_1_layer_nn.bias
    elif at == 'general':
      w1, b1 = self.attn_proj_general.weight, self.attn_proj_general.bias
    elif at == 'concat':
      w1, b1 = self.attn_proj_layer1.weight, self.attn_proj_layer1.bias
      w2, b2 = self.attn_proj_layer2.weight, self.attn_proj_layer2.bias
    r = self.rnn
    self.embedding.weight._lr_padding_idx = self.embedding.padding_idx
    wc, bc = (self.concat_layer.weight, self.concat_layer.bias) if at != 'none' else (None, None)
    upper = [w for layer in range(1, self.num_layers) for w in r.layer_weights(layer, 1)]
    return [self.embedding.weight, r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0, w1, b1, w2, b2,
            wc, bc, self.output_proj.weight, self.output_proj.bias] + upper

  def decode_sequence(self, tokens, previous_state, encoder_lens, encoder_hidden_states,
                      teacher_forced=None, seed=None):
    """The whole loop of train_better_model.py:56-63 in one call.

    tokens (B, L) = chars[:, :L] (teacher inputs); teacher_forced: L booleans (None = all True;
    entry 0 is always teacher forced — the reference feeds BOS); previous_state: the encoder's
    final state (1, B, Hd) or (h, c).  Returns (log_probs (B,L,V), sampled (B,L) int32, final_state)."""
    _C.require_cuda(tokens, encoder_hidden_states)
    B, L = tokens.shape
    mode = _MODES[self.rnn_type]
    # (num_layers, B, Hd): one initial state per layer of the stack (better_model.py:181)
    if isinstance(previous_state, tuple):
      h0, c0 = previous_state[0].contiguous(), previous_state[1].contiguous()
    else:
      h0, c0 = previous_state.contiguous(), None
    assert h0.dim() == 3 and h0.shape[0] == self.num_layers, "previous_state must be (num_layers, B, hidden)"
    drop_mask = None
    if self.training and self.rnn_dropout and self.num_layers > 1:
      # nn.GRU/LSTM(dropout=p), training mode: outputs of every layer but the last, independently per step
      # (0 with probability p, else 1 / (1 - p): lr_dropout_forward's Philox mask, seeded from torch's host generator)
      drop_mask = torch.empty((self.num_layers - 1, B, L, self.hidden_size), dtype=torch.float32,
                              device=encoder_hidden_states.device)
      _C.check(_C.lib().lr_dropout_forward(None, None, drop_mask.data_ptr(), drop_mask.numel(),
                                           float(self.rnn_dropout), int(torch.randint(0, 2 ** 62, (1,)).item()),
                                           _C.stream_handle()), "lr_dropout_forward")
    enc = encoder_hidden_states.to(torch.float32).contiguous()
    if teacher_forced is None:
      teacher_forced = [True] * L
    if L > 1 and mode != 2 and all(teacher_forced):
      why = _C.lib().lr_rnn_one_launch_status(mode, B, L, self.hidden_size, self.hidden_size, 1)
      if why != 0:
        from .encoder import _note_step_kernel_fallback
        _note_step_kernel_fallback(self.rnn_type, self.hidden_size, why)   # (one warning per shape and reason)
    if seed is None:
      # drawn from torch's (host) generator, so --seed / torch.manual_seed control the multinomial
      # draws of the loop as they do in the reference (train_better_model.py:63); no device sync
      seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    lp, sampled, h_n, c_n = _AttnDecoderFunction.apply(
        tokens.to(torch.int32).contiguous(), tuple(bool(t) for t in teacher_forced), int(seed), mode,
        _ATT_CODE[self.attention_type], self.attn_hidden_size, enc,
        encoder_lens.to(device=enc.device, dtype=torch.int32).contiguous(), h0, c0, self.output_mask, drop_mask,
        *self._params())
    final_state = (h_n, c_n) if mode == 1 else h_n
    return lp, sampled, final_state

  def forward(self, input_, previous_state, encoder_lens, encoder_hidden_states):
    """One step, reference contract (better_model.py:161-231): input_ (B,), previous_state
    (num_layers, B, Hd) [(h, c) for the LSTM], encoder_lens (B,), encoder_hidden_states (B, T, Hd)
    -> (output_log_probs (B, V), final_state)."""
    lp, _, final_state = self.decode_sequence(input_.unsqueeze(1), previous_state, encoder_lens,
                                              encoder_hidden_states)
    return lp[:, 0], final_state

  def save_best_model(self, error, file_path):
    """better_model.py:237-245."""
    if error < self.best_error:
      self.best_error = error
      folder = os.path.dirname(file_path)
      if folder and not os.path.exists(folder):
        os.makedirs(folder)
      torch.save(self.state_dict(), file_path)
      print("\tSaving best error '{}' to '{}'".format(self.best_error, file_path))
