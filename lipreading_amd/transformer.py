"""Transformer sequence encoder over per-frame features + CTC head.

BUILD-DEFINED: the reference has no transformer encoder — BASELINE.json's
configs[4] names one ("transformer encoder over per-frame conv features (self-attn MFMA path) +
CTC").  The specification is this repo's:

    x  = Linear(frame_dim, d_model)(frames) + sinusoidal positional encoding
    x  = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout=0, activation=relu,
                                 norm_first=False)  x num_layers     (key padding mask = frame_lens)
    log_probs = masked_log_softmax(Linear(d_model, V+1)(x))          (the reference's CTC head,
                                                                      better_model.py:92-93)

i.e. exactly torch.nn.TransformerEncoder, which is also the CPU oracle (there is NO reference
parity for this stage).  Parameters live in real torch modules (same names, shapes and
initialisation as nn.TransformerEncoder; state_dicts interchange) that are never called: the
arithmetic is the HIP path, and since round 5 the whole stack is THREE enqueues (lr_tfm_forward,
lr_tfm_backward_data, lr_tfm_backward_weights: lr_transformer.hip): every projection a product that
reads its operands as they lie in memory (lr_fgemm.hip) with bias / residual / positional table / ReLU
in its epilogue, every weight gradient of the stack (bias gradients included) in one launch.
It drops in wherever VideoEncoder does: forward(frames, frame_lens) -> (log_probs, hidden, None).
"""
import math

import torch
import torch.nn as nn

from . import _C
from .data import BOS, PAD
from .encoder import _ProjLogSoftmaxFunction, _direct_grads, _notify, _ptr_array


_X3, _X_BF16, _DX_BF16, _ATTN_FUSED, _ROWBLOCK = 1, 2, 4, 8, 16   # LR_TFM_* (include/lipreading_hip.h)

# test hook: False keeps the row-wise half of a layer as five launches per direction (lr_fgemm products + LayerNorm)
rowblock_layers = True


class _StackFunction(torch.autograd.Function):
  """input projection + positional table + every encoder layer: lr_tfm_forward / lr_tfm_backward_data /
  lr_tfm_backward_weights (three enqueues; lr_transformer.hip composes them from lr_fgemm.hip products, the fused
  attention and the LayerNorm kernels).  x (B, T, I) fp32 or — the conv frontend's features — bf16."""

  @staticmethod
  def forward(ctx, x, lens, pe, cfg, *weights):
    L = _C.lib()
    mode, Dm, nhead, F, nlayers, eps = cfg
    B, T, I = x.shape
    dev = x.device
    dims = (B, T, I, Dm, nhead, F, nlayers)
    rbytes, wbytes = L.lr_tfm_reserve_bytes(mode, *dims), L.lr_tfm_workspace_bytes(mode, *dims)
    if rbytes == 0:
      raise ValueError("lr_tfm: unsupported shape (d_model %d, heads %d, feed-forward %d, T %d)" % (Dm, nhead, F, T))
    reserve = torch.empty(rbytes, dtype=torch.uint8, device=dev)
    ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)    # (the forward's split-K slabs; then the backward's buffers)
    h = torch.empty((B, T, Dm), dtype=torch.float32, device=dev)
    _C.check(L.lr_tfm_forward(mode, x.data_ptr(), lens.data_ptr(), _ptr_array(weights), pe.data_ptr(), h.data_ptr(),
                              reserve.data_ptr(), rbytes, ws.data_ptr(), wbytes, *dims, float(eps), _C.stream_handle()),
             "lr_tfm_forward")
    if any(ctx.needs_input_grad):
      ctx.save_for_backward(x, lens, reserve, ws, *weights)
      ctx.cfg = cfg
    # (inference — torch.no_grad(), eval loops: nothing is saved, so the reserve (every layer's activations) and the
    # workspace go back to the caching allocator as soon as the head has read h)
    return h

  @staticmethod
  def backward(ctx, dh):
    from . import encoder as _enc
    x, lens, reserve, ws = ctx.saved_tensors[:4]
    weights = ctx.saved_tensors[4:]
    L = _C.lib()
    mode, Dm, nhead, F, nlayers, eps = ctx.cfg
    B, T, I = x.shape
    dims = (B, T, I, Dm, nhead, F, nlayers)
    dh = dh.contiguous()
    dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None   # (bf16 where x is: LR_TFM_DX_BF16)
    _C.check(L.lr_tfm_backward_data(mode, lens.data_ptr(), _ptr_array(weights), dh.data_ptr(), _C.ptr(dx),
                                    reserve.data_ptr(), reserve.numel(), ws.data_ptr(), ws.numel(), *dims,
                                    _C.stream_handle()), "lr_tfm_backward_data")
    direct = _direct_grads(weights)
    grads = [w.grad for w in weights] if direct else [torch.empty_like(w) for w in weights]

    def weight_half():
      _C.check(L.lr_tfm_backward_weights(mode, x.data_ptr(), _ptr_array(grads), 1 if direct else 0, reserve.data_ptr(),
                                         reserve.numel(), ws.data_ptr(), ws.numel(), *dims, _C.stream_handle()),
               "lr_tfm_backward_weights")
    if _enc.overlap_weight_grads and direct and dx is not None:
      # the pixel regime: whatever produced x (the conv frontend) only waits for dx — every weight gradient of the
      # stack goes to the encoder's side stream and runs beside the conv backward (joined by flush_deferred)
      _enc.flush_deferred()
      side = _enc._get_side_stream(x.device)
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):
        weight_half()
        _notify(weights)
        for hook in _enc.encoder_grads_complete_hooks:   # (the head's gradients were written before this backward)
          hook(weights[0])
      _enc._deferred.append((x, dh, reserve, ws, grads, weights))
      return (dx, None, None, None) + (None,) * len(weights)
    weight_half()
    if direct:
      _notify(weights)
      return (dx, None, None, None) + (None,) * len(weights)
    return (dx, None, None, None) + tuple(grads)


class _AttentionFunction(torch.autograd.Function):
  """Multi-head self-attention core: qkv [B,T,3D] (fused projection) -> context [B,T,D]; keys at
  positions >= key_lens[b] are masked."""

  @staticmethod
  def forward(ctx, qkv, key_lens, nhead, fused=False):
    L = _C.lib()
    st = _C.stream_handle()
    B, T, D3 = qkv.shape
    D = D3 // 3
    dh = D // nhead
    qkv = qkv.contiguous()
    dev = qkv.device
    if fused:
      # one launch per (sample, head) problem set on the bf16 matrix cores; nothing T x T is written
      scale = 1.0 / math.sqrt(dh)
      out = torch.empty((B, T, D), dtype=torch.float32, device=dev)
      _C.check(L.lr_attn_fused_forward(qkv.data_ptr(), key_lens.data_ptr(), out.data_ptr(), scale, B, T, nhead, dh, st),
               "lr_attn_fused_forward")
      ctx.save_for_backward(qkv, key_lens)
      ctx.cfg = (nhead, scale, True)
      return out
    probs = torch.empty((B, nhead, T, T), dtype=torch.float32, device=dev)
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D
    # scores[b,h] = Q[b,h] K[b,h]^T, read in place from the fused projection (row pitch 3D)
    _C.check(L.lr_sgemm_batched(0, 1, T, T, dh, 1.0, q, D3, T * D3, dh, k, D3, T * D3, dh, 0.0, probs.data_ptr(), T,
                                nhead * T * T, T * T, B, nhead, st), "lr_sgemm_batched(QK^T)")
    scale = 1.0 / math.sqrt(dh)
    _C.check(L.lr_attn_softmax_forward(probs.data_ptr(), key_lens.data_ptr(), scale, B, nhead, T, st),
             "lr_attn_softmax_forward")
    out = torch.empty((B, T, D), dtype=torch.float32, device=dev)
    _C.check(L.lr_sgemm_batched(0, 0, T, dh, T, 1.0, probs.data_ptr(), T, nhead * T * T, T * T, v, D3, T * D3, dh,
                                0.0, out.data_ptr(), D, T * D, dh, B, nhead, st), "lr_sgemm_batched(PV)")
    ctx.save_for_backward(qkv, probs)
    ctx.cfg = (nhead, scale, False)
    return out

  @staticmethod
  def backward(ctx, dout):
    nhead, scale, fused = ctx.cfg
    L = _C.lib()
    st = _C.stream_handle()
    dout = dout.contiguous()
    if fused:
      qkv, key_lens = ctx.saved_tensors
      B, T, D3 = qkv.shape
      dqkv = torch.empty_like(qkv)
      _C.check(L.lr_attn_fused_backward(qkv.data_ptr(), key_lens.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), scale,
                                        B, T, nhead, D3 // 3 // nhead, st), "lr_attn_fused_backward")
      return dqkv, None, None, None
    qkv, probs = ctx.saved_tensors
    B, T, D3 = qkv.shape
    D = D3 // 3
    dh = D // nhead
    dev = qkv.device
    dqkv = torch.empty_like(qkv)
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D
    dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + 4 * D, dqkv.data_ptr() + 8 * D
    PS, PI = nhead * T * T, T * T
    dP = torch.empty_like(probs)
    # dP = dO V^T ; dV = P^T dO
    _C.check(L.lr_sgemm_batched(0, 1, T, T, dh, 1.0, dout.data_ptr(), D, T * D, dh, v, D3, T * D3, dh, 0.0,
                                dP.data_ptr(), T, PS, PI, B, nhead, st), "lr_sgemm_batched(dP)")
    _C.check(L.lr_sgemm_batched(1, 0, T, dh, T, 1.0, probs.data_ptr(), T, PS, PI, dout.data_ptr(), D, T * D, dh, 0.0,
                                dv, D3, T * D3, dh, B, nhead, st), "lr_sgemm_batched(dV)")
    _C.check(L.lr_attn_softmax_backward(probs.data_ptr(), dP.data_ptr(), scale, B, nhead, T, st),
             "lr_attn_softmax_backward")
    # dQ = dS K ; dK = dS^T Q
    _C.check(L.lr_sgemm_batched(0, 0, T, dh, T, 1.0, dP.data_ptr(), T, PS, PI, k, D3, T * D3, dh, 0.0, dq, D3, T * D3,
                                dh, B, nhead, st), "lr_sgemm_batched(dQ)")
    _C.check(L.lr_sgemm_batched(1, 0, T, dh, T, 1.0, dP.data_ptr(), T, PS, PI, q, D3, T * D3, dh, 0.0, dk, D3, T * D3,
                                dh, B, nhead, st), "lr_sgemm_batched(dK)")
    return dqkv, None, None, None


def sinusoidal_encoding(max_len, d_model):
  pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
  div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
  pe = torch.zeros(max_len, d_model)
  pe[:, 0::2] = torch.sin(pos * div)
  pe[:, 1::2] = torch.cos(pos * div)
  return pe


class TransformerVideoEncoder(nn.Module):
  def __init__(self, frame_dim, d_model=256, nhead=4, num_layers=4, dim_feedforward=1024, enable_ctc=True,
               vocab_size=-1, char2idx=None, max_len=512):
    super().__init__()
    assert d_model % nhead == 0 and d_model % 4 == 0 and (d_model // nhead) % 4 == 0
    # what the one-call stack (lr_tfm_*, lr_transformer.hip) takes; said HERE rather than as "unsupported shape" at the
    # first forward (the per-op path of rounds 1-4, which had no such limits, is gone)
    if dim_feedforward % 4 != 0 or num_layers > 16 or d_model > 1920:
      raise ValueError("TransformerVideoEncoder: the HIP stack needs d_model, the head dimension and dim_feedforward to be "
                       "multiples of 4, at most 16 layers and d_model <= 1920 (got d_model %d, %d heads, feed-forward %d, "
                       "%d layers)" % (d_model, nhead, dim_feedforward, num_layers))
    self.frame_dim, self.d_model, self.nhead, self.num_layers = frame_dim, d_model, nhead, num_layers
    self.enable_ctc = enable_ctc
    self.best_error = 1
    # 'f32': exact fp32 MFMA linears; 'bf16x3' (set by frontend.PixelLipReader): hi/lo split bf16 MFMA
    self.input_projection = 'f32'
    self.input_is_bf16 = False
    # self-attention core: 'f32' (default) = batched fp32 MFMA GEMMs + a softmax kernel (exact fp32, the path the
    # torch oracle is compared with at 2e-4); 'bf16' (set by frontend.PixelLipReader, like the bf16 input
    # projection) = fused QK^T -> masked softmax -> PV on the bf16 matrix cores, one launch per (sample, head) set
    # (lr_attention.hip; T <= 96, head dim 32 / 64)
    self.attention = 'f32'
    self.input_proj = nn.Linear(frame_dim, d_model)
    layer = nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout=0.0, activation='relu',
                                       batch_first=True, norm_first=False)
    # parameter container only (torch's names / shapes / initialisation); never called
    self.layers = nn.TransformerEncoder(layer, num_layers, enable_nested_tensor=False).layers
    self.register_buffer("pe", sinusoidal_encoding(max_len, d_model), persistent=False)
    if enable_ctc:
      assert vocab_size > 0 and char2idx is not None
      self.vocab_size, self.adj_vocab_size, self.char2idx = vocab_size, vocab_size + 1, char2idx
      mask = torch.ones(self.adj_vocab_size)
      mask[char2idx[PAD] + 1] = 0
      mask[char2idx[BOS] + 1] = 0
      self.register_buffer("output_mask", mask, persistent=False)
      self.output_proj = nn.Linear(d_model, self.adj_vocab_size)

  def forward(self, frames, frame_lens, max_len=None, need_final_state=True):
    """frames (B, T, ...) f32, frame_lens (B,) -> (log_probs (B,Tmax,V+1), hidden (B,Tmax,d_model), None)
    [(hidden, None) without the CTC head] — VideoEncoder's contract; there is no recurrent final state
    (`need_final_state` is accepted for that contract and has nothing to switch off)."""
    _C.require_cuda(frames)
    frames = frames.reshape(frames.shape[0], frames.shape[1], -1)
    B, T, I = frames.shape
    assert I == self.frame_dim
    if max_len is None:
      max_len = int(frame_lens.max())
    assert 1 <= max_len <= min(T, self.pe.shape[0])
    # the pixel regime hands the conv frontend's bf16 features over as they are (LR_TFM_X_BF16)
    stored_bf16 = frames.dtype == torch.bfloat16 and self.input_is_bf16
    x = frames[:, :max_len]
    x = x.contiguous() if stored_bf16 else x.to(torch.float32).contiguous()
    lens = frame_lens.to(device=x.device, dtype=torch.int32).contiguous()
    fused = self.attention == 'bf16' and bool(_C.lib().lr_attn_fused_supported(max_len, self.d_model // self.nhead))
    mode = ((_X3 if self.input_projection == 'bf16x3' else 0) | ((_X_BF16 | _DX_BF16) if stored_bf16 else 0) |
            (_ATTN_FUSED if fused else 0))
    F = self.layers[0].linear1.out_features
    if (mode & _X3) and rowblock_layers and _C.lib().lr_tfm_rowblock_supported(B, max_len, self.d_model, F, len(self.layers)):
      mode |= _ROWBLOCK   # out-projection .. LN2 as one launch per layer and direction (lr_tfm_rowblock.hip)
    weights = [self.input_proj.weight, self.input_proj.bias]
    for layer in self.layers:
      at = layer.self_attn
      weights += [at.in_proj_weight, at.in_proj_bias, at.out_proj.weight, at.out_proj.bias, layer.linear1.weight,
                  layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias,
                  layer.norm2.weight, layer.norm2.bias]
    cfg = (mode, self.d_model, self.nhead, self.layers[0].linear1.out_features, len(self.layers),
           float(self.layers[0].norm1.eps))
    h = _StackFunction.apply(x, lens, self.pe, cfg, *weights)
    if self.enable_ctc:
      lp = _ProjLogSoftmaxFunction.apply(h, self.output_proj.weight, self.output_proj.bias, self.output_mask)
      return lp, h, None
    return h, None

  def save_best_model(self, error, file_path):
    import os
    if error < self.best_error:
      self.best_error = error
      folder = os.path.dirname(file_path)
      if folder and not os.path.exists(folder):
        os.makedirs(folder)
      torch.save(self.state_dict(), file_path)
      print("\tSaving best error '{}' to '{}'".format(self.best_error, file_path))
