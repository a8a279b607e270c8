"""CTC loss on the MI355X — drop-in for the reference's `src/train/ctc_loss.py:28 ctc_loss`.

Same name, argument meaning and return contract (a differentiable 0-dim tensor, or None where
the reference returns None), but the lattice never leaves HBM: the reference moves the
log-probs to the CPU and calls F.ctc_loss there (ctc_loss.py:85); here per-sample alpha/beta run
as one HIP workgroup per sample (lr_ctc_nll / lr_ctc_grad) and the reference's batch reduction
(length filter, equal-length runs, inf fallback, 'mean' weighting quirk) is evaluated by a
device kernel (lr_ctc_reduce) from the per-sample losses.
"""
import torch

from . import _C

_REDUCTIONS = {"sum": 0, "mean": 1}


class _CTCLossFunction(torch.autograd.Function):
  """forward: nll per sample -> reference reduction.  backward: beta pass + gradient rows."""

  @staticmethod
  def forward(ctx, log_probs, labels_p1, frame_lens, label_lens, reduction, max_label_len):
    L = _C.lib()
    ctx.set_materialize_grads(False)   # status / nll get no gradient: no zero tensors (a fill launch each) for them
    B, T, C = log_probs.shape
    dev = log_probs.device
    ws_bytes = L.lr_ctc_workspace_bytes(B, T, C, max_label_len)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    out = torch.empty(1, dtype=torch.float32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    gw = torch.empty(B, dtype=torch.float32, device=dev)
    st = _C.stream_handle()
    # the recursions and the reference's batch reduction (ctc_loss.py:64-112) in one launch (two for labels > 31)
    _C.check(L.lr_ctc_nll_reduce(log_probs.data_ptr(), log_probs.stride(0), log_probs.stride(1),
                                 labels_p1.data_ptr(), labels_p1.stride(0), frame_lens.data_ptr(),
                                 label_lens.data_ptr(), nll.data_ptr(), ws.data_ptr(), ws_bytes, reduction,
                                 out.data_ptr(), status.data_ptr(), gw.data_ptr(), B, T, C, max_label_len, st),
             "lr_ctc_nll_reduce")
    ctx.save_for_backward(log_probs, labels_p1, frame_lens, label_lens, nll, gw, ws)
    ctx.max_label_len = max_label_len
    ctx.mark_non_differentiable(status, nll)
    return out.reshape(()), status, nll

  @staticmethod
  def backward(ctx, grad_out, _gs, _gn):
    if grad_out is None:
      return None, None, None, None, None, None
    log_probs, labels_p1, frame_lens, label_lens, nll, gw, ws = ctx.saved_tensors
    L = _C.lib()
    B, T, C = log_probs.shape
    st = _C.stream_handle()
    ml = ctx.max_label_len
    # the incoming gradient of the reduced loss multiplies every sample's weight INSIDE the kernel (a device scalar:
    # `gw * grad_out` here was an elementwise launch of its own, ~5 us of a 470 us step)
    go = grad_out.reshape(1)
    if go.dtype != torch.float32:
      go = go.float()
    grad = torch.empty((B, T, C), dtype=torch.float32, device=log_probs.device)
    # grad is addressed with the same (stride_b, stride_t) as log_probs, so give the kernel a
    # contiguous view of the lattice when the input is a transposed view
    lp = log_probs if log_probs.is_contiguous() else log_probs.contiguous()
    _C.check(L.lr_ctc_grad_scaled(lp.data_ptr(), lp.stride(0), lp.stride(1), labels_p1.data_ptr(),
                                  labels_p1.stride(0), frame_lens.data_ptr(), label_lens.data_ptr(),
                                  nll.data_ptr(), gw.data_ptr(), go.data_ptr(), grad.data_ptr(), ws.data_ptr(),
                                  ws.numel(), B, T, C, ml, st), "lr_ctc_grad_scaled")
    return grad, None, None, None, None, None


def ctc_loss_with_status(encoder_outputs, labels, frame_lens, label_lens, reduction):
  """Device-only variant: returns (loss, status, nll) tensors with NO host synchronisation.

  status[0] == 1 where the reference returns None (the loss is then 0 with a zero gradient).
  The training loop uses this to keep a whole step free of device->host round trips."""
  _C.require_cuda(encoder_outputs, labels, frame_lens, label_lens)
  if reduction not in _REDUCTIONS:
    raise ValueError("reduction must be 'mean' or 'sum', got %r" % (reduction,))
  if encoder_outputs.dim() != 3:
    raise ValueError("encoder_outputs must be (batch, seq_len, classes)")
  lp = encoder_outputs
  if lp.dtype != torch.float32:
    lp = lp.float()
  if lp.stride(2) != 1:
    lp = lp.contiguous()
  # req (5)/(3) of the reference docstring: int32 integers, labels moved up by one so that
  # index 0 is the blank (ctc_loss.py:42,80)
  labels_p1 = (labels.to(torch.int32) + 1).contiguous()
  if labels_p1.dim() != 2 or labels_p1.shape[0] != lp.shape[0]:
    raise ValueError("labels must be (batch, max_label_len)")
  fl = frame_lens.to(torch.int32).contiguous()
  ll = label_lens.to(torch.int32).contiguous()
  max_label_len = max(1, min(int(labels_p1.shape[1]), 256))
  return _CTCLossFunction.apply(lp, labels_p1, fl, ll, _REDUCTIONS[reduction], max_label_len)


def prepare_ctc_inputs(chars, frame_lens, char_lens):
  """The train loop's label plumbing (train_better_model.py:31-32 + ctc_loss.py:42,80) in one launch:
  chars (B, Lc) int64 with BOS in column 0, frame_lens / char_lens (B,) int64, all on the device ->
  (labels_p1 (B, Lc-1) int32 = chars[:, 1:] + 1, frame_lens int32, label_lens int32 = char_lens - 1)."""
  _C.require_cuda(chars, frame_lens, char_lens)
  assert chars.dim() == 2 and chars.shape[1] >= 2 and chars.dtype == torch.int64 and chars.stride(1) == 1
  assert frame_lens.dtype == torch.int64 and char_lens.dtype == torch.int64
  B, Lc = chars.shape
  dev = chars.device
  labels_p1 = torch.empty((B, Lc - 1), dtype=torch.int32, device=dev)
  fl = torch.empty(B, dtype=torch.int32, device=dev)
  ll = torch.empty(B, dtype=torch.int32, device=dev)
  _C.check(_C.lib().lr_ctc_prepare_i64(chars.data_ptr(), chars.stride(0), frame_lens.contiguous().data_ptr(),
                                       char_lens.contiguous().data_ptr(), labels_p1.data_ptr(), fl.data_ptr(),
                                       ll.data_ptr(), B, Lc - 1, _C.stream_handle()), "lr_ctc_prepare_i64")
  return labels_p1, fl, ll


def ctc_loss_prepared(encoder_outputs, labels_p1, frame_lens32, label_lens32, reduction):
  """ctc_loss_with_status on the outputs of prepare_ctc_inputs (no conversions, no host synchronisation)."""
  _C.require_cuda(encoder_outputs, labels_p1, frame_lens32, label_lens32)
  if reduction not in _REDUCTIONS:
    raise ValueError("reduction must be 'mean' or 'sum', got %r" % (reduction,))
  lp = encoder_outputs if encoder_outputs.dtype == torch.float32 else encoder_outputs.float()
  if lp.stride(2) != 1:
    lp = lp.contiguous()
  assert labels_p1.dtype == torch.int32 and labels_p1.is_contiguous() and labels_p1.shape[0] == lp.shape[0]
  assert frame_lens32.dtype == torch.int32 and label_lens32.dtype == torch.int32
  max_label_len = max(1, min(int(labels_p1.shape[1]), 256))
  return _CTCLossFunction.apply(lp, labels_p1, frame_lens32, label_lens32, _REDUCTIONS[reduction], max_label_len)


def ctc_loss(encoder_outputs, labels, frame_lens, label_lens, reduction, device=None):
  """Reference signature (src/train/ctc_loss.py:28).

  encoder_outputs (B,T,V') log-probs; labels (B,Lmax) WITHOUT BOS and not yet shifted;
  frame_lens ascending (asserted, :39); returns the loss or None (skip the batch, :49-51,
  :110-112).  `device` is accepted for signature compatibility and ignored: the computation
  stays on the device the log-probs live on.
  """
  assert bool((frame_lens[1:] - frame_lens[:-1] >= 0).all())  # ctc_loss.py:39
  loss, status, _ = ctc_loss_with_status(encoder_outputs, labels, frame_lens, label_lens, reduction)
  if int(status.item()) != 0:   # the one host read the None contract forces
    return None
  return loss
