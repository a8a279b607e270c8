"""Flat parameter storage + fused clip/Adam for the reference's optimisation tail
(`src/train/train_better_model.py:78-80`: clip_grad_norm_ per module, then Adam.step, with
`torch.optim.Adam(params, lr)` rebuilt every epoch at `src/scripts/train.py:280`).

All parameters of a module become views into ONE contiguous fp32 buffer (and their .grad views
into one gradient buffer), so the tail of a step is: one sum-of-squares launch, one Adam launch,
and — under data parallelism — a few large RCCL all-reduces over slices of the gradient buffer
instead of one per tensor.
"""
import torch

from . import _C

_ALIGN = 64  # floats: every parameter starts 256-byte aligned (float4 operand loads need 16 B)


class FlatParameters(object):
  """Re-homes `module`'s parameters (already on the GPU) into one flat buffer."""

  def __init__(self, module):
    self.params = [p for p in module.parameters()]
    assert self.params, "module has no parameters"
    dev = self.params[0].device   # storage only: works on any device (gloo tests run on CPU)
    self.offsets = []
    # the first _ALIGN floats of the buffers belong to no parameter: grad[0:2] are the two WORDS a data-parallel step
    # sums with its gradients — {ranks whose batch was skipped, ranks whose recurrence timed out}
    # (distributed.GradSync, lr_fault_export_f32) — so that the exchange needs no second collective
    off = _ALIGN
    for p in self.params:
      assert p.dtype == torch.float32 and p.device == dev
      self.offsets.append(off)
      off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
    self.numel = off
    self.data = torch.zeros(off, dtype=torch.float32, device=dev)
    self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
    self.words = self.grad[0:2]
    self.first = _ALIGN     # where the parameters start
    for p, o in zip(self.params, self.offsets):
      n = p.numel()
      self.data[o:o + n].copy_(p.data.reshape(-1))
      p.data = self.data[o:o + n].view(p.shape)
    self.attach_grads()

  def attach_grads(self):
    """(Re)point every .grad at its slice of the flat gradient buffer; autograd then
    accumulates in place (two backward passes per step in the reference, :69 and :74)."""
    for p, o in zip(self.params, self.offsets):
      n = p.numel()
      view = self.grad[o:o + n].view(p.shape)
      if p.grad is None or p.grad.data_ptr() != view.data_ptr():
        p.grad = view

  def zero_grad(self, also_zero=None, ctc_inputs=None):
    """opt.zero_grad() (train_better_model.py:67) AND the top of a step for the device-side fault words: one launch
    (lr_step_begin) clears the flat gradient buffer and moves a time-out raised by a one-launch recurrence of the
    PREVIOUS step from `pending` to `total` (include/lipreading_hip.h).  also_zero: a float32[1] device tensor to
    clear in the same launch (FusedAdam's sum-of-squares accumulator).  ctc_inputs = (chars, frame_lens, char_lens)
    int64 on the device: the step's label plumbing (ctc.prepare_ctc_inputs) rides in the same launch; returns its
    (labels_p1, frame_lens32, label_lens32)."""
    if self.grad.is_cuda and ctc_inputs is not None:
      import torch as _t
      chars, frame_lens, char_lens = ctc_inputs
      B, Lc = chars.shape
      labels_p1 = _t.empty((B, Lc - 1), dtype=_t.int32, device=chars.device)
      fl = _t.empty(B, dtype=_t.int32, device=chars.device)
      ll = _t.empty(B, dtype=_t.int32, device=chars.device)
      _C.check(_C.lib().lr_step_begin_ctc(self.grad.data_ptr(), self.numel, _C.ptr(also_zero), chars.data_ptr(),
                                          chars.stride(0), frame_lens.contiguous().data_ptr(),
                                          char_lens.contiguous().data_ptr(), labels_p1.data_ptr(), fl.data_ptr(),
                                          ll.data_ptr(), B, Lc - 1, _C.stream_handle()), "lr_step_begin_ctc")
      self.attach_grads()
      return labels_p1, fl, ll
    if self.grad.is_cuda:
      _C.check(_C.lib().lr_step_begin(self.grad.data_ptr(), self.numel, _C.ptr(also_zero), _C.stream_handle()),
               "lr_step_begin")
    else:
      self.grad.zero_()   # storage-only use (gloo tests on CPU)
      if also_zero is not None:
        also_zero.zero_()
    self.attach_grads()


class FusedAdam(object):
  """Adam over a FlatParameters buffer: lr_sumsq (+ optional all-reduce hook) + lr_adam_step."""

  def __init__(self, flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
    self.flat = flat
    self.lr, self.betas, self.eps = float(lr), betas, float(eps)
    dev = flat.data.device
    self.exp_avg = torch.zeros_like(flat.data)
    self.exp_avg_sq = torch.zeros_like(flat.data)
    self.step_count = torch.zeros(2, dtype=torch.int32, device=dev)   # [updates taken, steps skipped]
    self._sumsq = torch.zeros(2, dtype=torch.float32, device=dev)     # {accumulator, ticket}: lr_clip_adam_step leaves both 0
    self._scratch = torch.zeros(8, dtype=torch.float32, device=dev)   # [0..3] the step's coefficients, [5] the sum of squares

  def zero_grad(self):
    """opt.zero_grad() — and THE TOP OF A STEP for the device-side fault words: the same launch rolls a recurrence
    time-out of the PREVIOUS step from `pending` to `total` (include/lipreading_hip.h).  Call it BEFORE the forward pass
    (train.ctc_step / decoder_step do; `step_begin` is the same call under the name that says so): a loop that runs
    forward -> zero_grad -> backward -> step would clear a time-out raised by its own forward pass before lr_adam_step
    reads it, and update the weights from garbage gradients.  Gradient accumulation over micro-batches: one
    zero_grad() in front of the first micro-batch's forward, none in between."""
    # (the clip's accumulator and ticket are cleared by the same launch — they are clean already, lr_clip_adam_step puts
    # them back to zero itself; this only mends a launch that was torn down half-way)
    self.flat.zero_grad(also_zero=self._sumsq)
    self._early_armed = hasattr(self, "_early_lo")

  step_begin = zero_grad

  def zero_grad_and_prepare_ctc(self, chars, frame_lens, char_lens):
    """zero_grad() with the step's CTC label plumbing (ctc.prepare_ctc_inputs) in the same launch; returns
    (labels_p1, frame_lens32, label_lens32)."""
    self._early_armed = hasattr(self, "_early_lo")
    return self.flat.zero_grad(also_zero=self._sumsq, ctc_inputs=(chars, frame_lens, char_lens))

  def reset(self, lr=None):
    """What re-creating torch.optim.Adam each epoch does (train.py:280): moments and step
    count start over."""
    self.exp_avg.zero_()
    self.exp_avg_sq.zero_()
    self.step_count.zero_()
    if lr is not None:
      self.lr = float(lr)

  def sum_squares_early(self, module):
    """Pixel regime, single process, ONE backward per step: `module` (the encoder) owns the TAIL of the flat buffer, and its
    gradients are complete long before the conv frontend's (its last weight half runs on the side stream beside the conv
    backward): take their share of the clip's sum of squares there, so that the launch in front of Adam only sweeps the
    frontend's 1.3 MB (round 5: 23 -> 6 us at the end of the step).  Not under data parallelism (the clip is taken on the
    all-reduced gradient) and not with several backward passes per step (the early sum would see a partial gradient)."""
    from . import encoder as _enc
    import torch.distributed as _dist
    if _dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1:
      # the clip is taken on the ALL-REDUCED gradient; an early sum would see this rank's share only
      raise RuntimeError("FusedAdam.sum_squares_early is a single-process option (torch.distributed world size > 1)")
    ids = {id(p) for p in module.parameters()}
    idx = [i for i, p in enumerate(self.flat.params) if id(p) in ids]
    if not (idx and idx == list(range(idx[0], len(self.flat.params)))):
      raise ValueError("sum_squares_early: the module must own the tail of the flat buffer")
    if self.flat.offsets[idx[0]] <= self.flat.first:
      # the module owns the WHOLE buffer: nothing would be left for the launch in front of Adam (a zero-sized grid)
      raise ValueError("sum_squares_early: the module owns every parameter; there is no later share to overlap with")
    self._early_lo = self.flat.offsets[idx[0]]
    early_ptrs = {p.data_ptr() for p in module.parameters()}
    self._early_done = False
    self._early_armed = False        # between this optimiser's zero_grad() and its step(): the backward in between is ours
    import weakref
    ref = weakref.ref(self)

    def hook(done=None):
      # `done`: a parameter of the encoder whose gradients are complete (encoder.py passes it); another model's backward
      # between this optimiser's zero_grad() and step() must not trigger the sum over THIS buffer
      me = ref()
      if me is None:
        if hook in _enc.encoder_grads_complete_hooks:
          _enc.encoder_grads_complete_hooks.remove(hook)
        return
      if not me._early_armed or me._early_done:
        return
      if done is not None and done.data_ptr() not in early_ptrs:
        return
      n = me.flat.numel - me._early_lo
      _C.check(_C.lib().lr_sumsq(me.flat.grad.data_ptr() + 4 * me._early_lo, n, me._sumsq.data_ptr(),
                                 _C.stream_handle()), "lr_sumsq")
      me._early_done = True
    _enc.encoder_grads_complete_hooks.append(hook)

  def step(self, grad_norm=None, grad_scale=1.0, skip=None, dist_words=None, world=1):
    """grad_norm: max norm for clip_grad_norm_ (None = no clipping); grad_scale: multiplies the
    gradient first (1/world after an all-reduce sum); skip: int32[1] device flag — non-zero
    leaves parameters, moments and step count untouched (the reference's `continue`).  The
    kernel also reads the device-side fault word: a step whose one-launch recurrence timed out
    (garbage gradients) updates nothing either.  dist_words (data parallel, distributed.GradSync): the two summed words
    at the front of the flat gradient buffer decide instead of `skip` — skipped only if all `world` ranks skipped,
    no update if any rank timed out."""
    from .encoder import flush_deferred
    flush_deferred()   # weight-gradient work still on the side stream (encoder.overlap_weight_grads)
    L = _C.lib()
    f = self.flat
    st = _C.stream_handle()
    o, n = f.first, f.numel - f.first          # (the words in front of the parameters are no gradient)
    ptrs = [t.data_ptr() + 4 * o for t in (f.data, f.grad, self.exp_avg, self.exp_avg_sq)]
    early = getattr(self, "_early_done", False)
    self._early_done = False
    self._early_armed = False
    if grad_norm is not None and float(grad_norm) > 0:
      # sum of squares -> clip coefficient -> Adam in two launches (the first one's last workgroup derives the coefficients)
      assert not (early and dist_words is not None), "sum_squares_early is a single-process option"
      n_sumsq = (self._early_lo - o) if early else n
      _C.check(L.lr_clip_adam_step(*ptrs, n, self._sumsq.data_ptr(), float(grad_norm),
                                   float(grad_scale), self.lr, self.betas[0], self.betas[1], self.eps,
                                   self.step_count.data_ptr(), _C.ptr(skip), self._scratch.data_ptr(),
                                   _C.ptr(dist_words), float(world), n_sumsq, st),
               "lr_clip_adam_step")
      return
    if early:
      self._sumsq.zero_()     # (no clip this step: the early sum is dropped)
    _C.check(L.lr_adam_step(*ptrs, n, None, 0.0, float(grad_scale),
                            self.lr, self.betas[0], self.betas[1], self.eps,
                            self.step_count.data_ptr(), _C.ptr(skip), self._scratch.data_ptr(),
                            _C.ptr(dist_words), float(world), st),
             "lr_adam_step")

  def skipped_steps(self):
    """Steps that updated nothing since the last reset() (a batch the reference `continue`s past, or a step whose
    one-launch recurrence timed out); one host read."""
    return int(self.step_count[1].item())

  def total_norm(self):
    """sqrt of the last sum of squares (valid after a step with grad_norm)."""
    return self._scratch[5:6].sqrt()
