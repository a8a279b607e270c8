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
arithmetic is the HIP path — projections and the per-(sample, head) QK^T / PV products on the fp32
matrix cores (lr_sgemm, lr_sgemm_batched reading Q/K/V in place out of the fused QKV projection),
LayerNorm(+residual), key-masked softmax, ReLU and the positional add in lr_transformer.hip.
It drops in wherever VideoEncoder does: forward(frames, frame_lens) -> (log_probs, hidden, None).
"""
import math

import torch
import torch.nn as nn

from . import _C
from .data import BOS, PAD
from .encoder import _ProjLogSoftmaxFunction


def _gemm(ta, tb, M, N, K, A, lda, Bm, ldb, C, ldc, bias=None, alpha=1.0, beta=0.0, x3=False):
  """fp32 GEMM on the fp32 matrix cores (lr_sgemm) or, x3, on the bf16 matrix cores with hi/lo split
  operands (lr_xgemm, ~1e-5 relative): the pixel regime's choice, as for the recurrent encoder."""
  L = _C.lib()
  if x3:
    wsb = L.lr_xgemm_workspace_bytes(int(ta), int(tb), M, N, K)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=C.device)
    _C.check(L.lr_xgemm(int(ta), int(tb), M, N, K, alpha, A.data_ptr(), lda, Bm.data_ptr(), ldb, beta, C.data_ptr(),
                        ldc, _C.ptr(bias), 0, 0, ws.data_ptr(), wsb, _C.stream_handle()), "lr_xgemm")
    return
  wsb = L.lr_sgemm_workspace_bytes(M, N, K)
  ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=C.device)
  _C.check(L.lr_sgemm(int(ta), int(tb), M, N, K, alpha, A.data_ptr(), lda, Bm.data_ptr(), ldb, beta, C.data_ptr(),
                      ldc, _C.ptr(bias), 0, 0, ws.data_ptr() if wsb else None, wsb, _C.stream_handle()), "lr_sgemm")


class _LinearFunction(torch.autograd.Function):
  """y[R,N] = x[R,K] W[N,K]^T + b — torch.nn.Linear on the matrix cores."""

  @staticmethod
  def forward(ctx, x, weight, bias, x3=False):
    shape = x.shape
    x2 = x.reshape(-1, shape[-1]).contiguous()
    R, K = x2.shape
    N = weight.shape[0]
    x3 = bool(x3) and R >= 256 and K >= 128   # the split path pays only for real contractions
    y = torch.empty((R, N), dtype=torch.float32, device=x.device)
    _gemm(0, 1, R, N, K, x2, K, weight, K, y, N, bias=bias, x3=x3)
    ctx.save_for_backward(x2, weight)
    ctx.shape = shape
    ctx.x3 = x3
    return y.reshape(shape[:-1] + (N,))

  @staticmethod
  def backward(ctx, dy):
    x2, weight = ctx.saved_tensors
    R, K = x2.shape
    N = weight.shape[0]
    dy2 = dy.reshape(R, N).contiguous()
    dx = torch.empty_like(x2)
    _gemm(0, 0, R, K, N, dy2, N, weight, K, dx, K, x3=ctx.x3)         # dx = dy W
    dW = torch.empty_like(weight)
    _gemm(1, 0, N, K, R, dy2, N, x2, K, dW, K, x3=ctx.x3)             # dW = dy^T x
    ones = torch.ones((R, 1), dtype=torch.float32, device=dy.device)
    db = torch.empty((N,), dtype=torch.float32, device=dy.device)
    _gemm(1, 0, N, 1, R, dy2, N, ones, 1, db, 1)                      # db = dy^T 1
    return dx.reshape(ctx.shape), dW, db, None


class _LayerNormFunction(torch.autograd.Function):
  """y = LayerNorm(x + residual) * gamma + beta."""

  @staticmethod
  def forward(ctx, x, residual, gamma, beta, eps):
    L = _C.lib()
    D = x.shape[-1]
    x = x.contiguous()
    residual = residual.contiguous()
    R = x.numel() // D
    y = torch.empty_like(x)
    stats = torch.empty((R, 2), dtype=torch.float32, device=x.device)
    _C.check(L.lr_layernorm_forward(x.data_ptr(), residual.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                    stats.data_ptr(), R, D, eps, _C.stream_handle()), "lr_layernorm_forward")
    ctx.save_for_backward(x, residual, gamma, stats)
    return y

  @staticmethod
  def backward(ctx, dy):
    x, residual, gamma, stats = ctx.saved_tensors
    L = _C.lib()
    D = x.shape[-1]
    R = x.numel() // D
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    dg = torch.empty_like(gamma)
    db = torch.empty_like(gamma)
    wsb = L.lr_layernorm_workspace_bytes(D)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    _C.check(L.lr_layernorm_backward(x.data_ptr(), residual.data_ptr(), gamma.data_ptr(), stats.data_ptr(),
                                     dy.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb,
                                     0, R, D, _C.stream_handle()), "lr_layernorm_backward")
    return dx, dx, dg, db, None


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


class _ReluFunction(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    x = x.contiguous()
    y = torch.empty_like(x)
    _C.check(_C.lib().lr_relu_forward(x.data_ptr(), y.data_ptr(), x.numel(), _C.stream_handle()), "lr_relu_forward")
    ctx.save_for_backward(y)
    return y

  @staticmethod
  def backward(ctx, dy):
    (y,) = ctx.saved_tensors
    dy = dy.contiguous()
    dx = torch.empty_like(dy)
    _C.check(_C.lib().lr_relu_backward(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), dy.numel(), _C.stream_handle()),
             "lr_relu_backward")
    return dx


class _AddPositionalFunction(torch.autograd.Function):
  """x[b,t,:] + pe[t,:]; the encoding is a constant, the gradient passes through."""

  @staticmethod
  def forward(ctx, x, pe):
    B, T, D = x.shape
    y = x.contiguous().clone()
    _C.check(_C.lib().lr_add_rows(y.data_ptr(), pe.data_ptr(), B, T, D, _C.stream_handle()), "lr_add_rows")
    return y

  @staticmethod
  def backward(ctx, dy):
    return dy, None


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
    x = frames[:, :max_len].to(torch.float32).contiguous()
    lens = frame_lens.to(device=x.device, dtype=torch.int32).contiguous()
    x3 = self.input_projection == 'bf16x3'
    h = _LinearFunction.apply(x, self.input_proj.weight, self.input_proj.bias, x3)
    h = _AddPositionalFunction.apply(h, self.pe[:max_len].contiguous())
    for layer in self.layers:
      at = layer.self_attn
      qkv = _LinearFunction.apply(h, at.in_proj_weight, at.in_proj_bias, x3)
      fused = self.attention == 'bf16' and bool(_C.lib().lr_attn_fused_supported(max_len, self.d_model // self.nhead))
      a = _AttentionFunction.apply(qkv, lens, self.nhead, fused)
      o = _LinearFunction.apply(a, at.out_proj.weight, at.out_proj.bias, x3)
      h = _LayerNormFunction.apply(o, h, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
      f = _ReluFunction.apply(_LinearFunction.apply(h, layer.linear1.weight, layer.linear1.bias, x3))
      f = _LinearFunction.apply(f, layer.linear2.weight, layer.linear2.bias, x3)
      h = _LayerNormFunction.apply(f, h, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
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
