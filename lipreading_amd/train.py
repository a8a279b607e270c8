"""Train / eval loops — the caller contract of `src/train/train_better_model.py` (train :7,
eval :89) on the MI355X path.

`train(encoder, decoding_step, data_loader, opt, device, char2idx, teacher_forcing_ratio,
grad_norm)` keeps the reference's argument meaning.  Differences, all about not stalling the
GPU: the framing asserts (:26-33) run on the host copies the loader hands over; the CTC loss
uses the device-side status flag instead of a `None` return, so a skipped batch costs no
device->host round trip (the optimiser kernel honours the flag); losses are accumulated on the
device and read once per epoch.  `decoding_step=None` runs the encoder+CTC path alone — the
path BASELINE.json's north_star names; with a `CharDecodingStep` the attention decoder loop
(:56-65) runs as one fused enqueue per batch (lipreading_amd/attention_decoder.py).
"""
import contextlib
import os

import torch

from .encoder import VideoEncoder as _VideoEncoder

from . import _C
from .ctc import ctc_loss_prepared, ctc_loss_with_status, prepare_ctc_inputs
from .data import BOS, EOS, PAD
from .optim import FusedAdam


def _check_framing(chars, char_lens, frame_lens, char2idx, use_ctc, in_eval=False):
  """train_better_model.py:26-33.  The loader hands chars / lengths over on the HOST (data.py's
  collate, as the reference's does), so the asserts cost no device round trip; a caller that passes
  device tensors pays one copy here.  The ascending-length requirement is ctc_loss.py:39's, so it is
  only checked when the CTC loss will run."""
  if chars.is_cuda or char_lens.is_cuda or frame_lens.is_cuda:
    chars, char_lens, frame_lens = chars.cpu(), char_lens.cpu(), frame_lens.cpu()
  assert (chars[:, 0].squeeze() == char2idx[BOS]).all()
  assert (chars.gather(1, (char_lens - 1).unsqueeze(dim=1)).squeeze() == char2idx[EOS]).all()
  if use_ctc and not in_eval:
    # (the reference's eval does not assert this, train_better_model.py:100-106: a caption longer than its clip makes
    # ctc_loss skip the sample there, and does so here)
    assert (frame_lens >= char_lens).all()  # otherwise ctc loss will produce inf
  labels = chars[:, 1:]
  label_lens = char_lens - 1
  assert (labels != char2idx[PAD]).sum() == label_lens.sum()
  if use_ctc:
    assert (frame_lens[1:] - frame_lens[:-1] >= 0).all()  # ctc_loss.py:39


class _NLLMeanFunction(torch.autograd.Function):
  """train_better_model.py:62,65 in two launches: sum over the decoder steps of nll_loss(ignore_index=PAD,
  reduction='sum') divided by (labels != PAD).sum().  log_probs (B, L, V); labels = a view of chars[:, 1:]
  whose first L columns are the steps' labels (int64, any row stride)."""

  @staticmethod
  def forward(ctx, log_probs, labels, pad):
    B, L, V = log_probs.shape
    lp = log_probs.contiguous()
    assert labels.dtype == torch.int64 and labels.stride(1) == 1 and labels.shape[0] == B and labels.shape[1] >= L
    out2 = torch.empty(2, dtype=torch.float32, device=lp.device)
    _C.check(_C.lib().lr_nll_mean_forward(lp.data_ptr(), labels.data_ptr(), labels.stride(0), L, int(pad),
                                          out2.data_ptr(), B * L, V, _C.stream_handle()), "lr_nll_mean_forward")
    ctx.save_for_backward(labels, out2)
    ctx.cfg = (B, L, V, int(pad))
    return out2[0]

  @staticmethod
  def backward(ctx, g):
    labels, out2 = ctx.saved_tensors
    B, L, V, pad = ctx.cfg
    g = g.reshape(1).to(torch.float32).contiguous()
    d_lp = torch.empty((B, L, V), dtype=torch.float32, device=out2.device)
    _C.check(_C.lib().lr_nll_mean_backward(labels.data_ptr(), labels.stride(0), L, pad, out2.data_ptr(), g.data_ptr(),
                                           d_lp.data_ptr(), B * L, V, _C.stream_handle()), "lr_nll_mean_backward")
    return d_lp, None, None


def decoder_nll(log_probs, labels, pad):
  """decoder_loss of train_better_model.py:62-65 from the (B, L, V) log-probs of the loop (any V)."""
  return _NLLMeanFunction.apply(log_probs, labels, pad)


def decoder_nll_sum(log_probs, labels, pad):
  """(sum of -log_probs[label] over the non-PAD steps, their count): eval adds the sums over the batches and divides
  once (train_better_model.py:127,138).  No gradient."""
  B, L, V = log_probs.shape
  lp = log_probs.contiguous()
  assert labels.dtype == torch.int64 and labels.stride(1) == 1 and labels.shape[0] == B and labels.shape[1] >= L
  out3 = torch.empty(3, dtype=torch.float32, device=lp.device)
  _C.check(_C.lib().lr_nll_forward3(lp.data_ptr(), labels.data_ptr(), labels.stride(0), L, int(pad), out3.data_ptr(),
                                    B * L, V, _C.stream_handle()), "lr_nll_forward3")
  return out3[2], out3[1]


class _Held(object):
  """`with _Held(syncs):` = every GradSync in `syncs` on hold(): gradients that become ready inside launch no
  collective (a collective is never captured into a hipGraph, and a replay fires no hooks at all); the buckets
  go out when the sync is called after the step."""

  def __init__(self, syncs):
    self._cms = [s.hold() for s in syncs if s is not None and hasattr(s, "hold")]

  def __enter__(self):
    for cm in self._cms:
      cm.__enter__()

  def __exit__(self, *exc):
    for cm in reversed(self._cms):
      cm.__exit__(*exc)
    return False


# what the last train() epoch skipped: {"batches", "skipped", "recurrence_faults"} (None before the first epoch)
last_epoch_stats = None

_ONES = {}


def _one(device):
  """A cached scalar 1 on `device`: the root gradient of loss.backward() (autograd otherwise fills a fresh
  ones_like tensor every step — one more launch)."""
  key = (device.type, device.index)
  t = _ONES.get(key)
  if t is None:
    t = _ONES[key] = torch.ones((), dtype=torch.float32, device=device)
  return t


class StepGraphs(object):
  """hipGraph capture of the optimisation step, cached per batch shape.

  A step is a few hundred short launches whose shapes depend only on (B, Tmax, Lmax): the collate
  function already pads every batch to those, so the launch sequence of a shape is recorded ONCE and
  replayed for every later batch of that shape (inputs are copied into the graph's static buffers; the
  loss / status tensors a replay returns are the graph's static outputs — consume them, e.g. `sum +=
  loss`, before the next step of the same shape).  The first `warmup` steps of a shape run eagerly on a
  side stream (allocator warm-up, as capture requires), the next one is captured, the rest replay.
  Same kernels in the same order on the same data: results are bit-identical to eager launches.

  What is captured: zero_grad -> forward -> loss(es) -> backward, plus clip + Adam when no gradient
  exchange sits between them (a collective is never captured: with `grad_sync` the exchange and the
  optimiser run eagerly after the replay).  Steps whose launch sequence depends on host randomness (a
  decoder loop with sampled inputs, teacher_forcing_ratio < 1) are not graphed."""

  def __init__(self, max_entries=16, warmup=2, enabled=True):
    self.max_entries, self.warmup, self.enabled = max_entries, warmup, enabled
    self._entries = {}     # key -> dict(static=..., count=int, graph=CUDAGraph|None, out=...)
    self._failed = set()   # shapes whose capture failed once: eager launches from then on, no retry
    self.replays = 0
    self.captures = 0

  def reset(self):
    """Drop every captured graph (their launch sequences are stale: e.g. after the recurrence fell back to the step
    kernels); shapes are captured again on their next steps."""
    self._entries.clear()
    self._failed.clear()

  def staging_buffers(self):
    """The static input tensors of the most recently used shape (None before its first step): a loader can write
    the next batch of that shape straight into them (same stream) and hand THEM to the step, which then skips
    its device-to-device staging copy of every input."""
    if not self._entries:
      return None
    return next(reversed(self._entries.values()))["static"]

  def run(self, key, inputs, body, capturable=True):
    """inputs: tuple of device tensors; body(*static_inputs) -> tuple of tensors.  Returns body's
    result for these inputs (from a replay when a graph for `key` exists)."""
    if not (self.enabled and capturable) or key in self._failed:
      return body(*inputs)
    e = self._entries.pop(key, None)
    if e is None:
      if self.max_entries <= 0:
        return body(*inputs)
      while len(self._entries) >= self.max_entries:   # least recently used shape goes (its pool is freed)
        self._entries.pop(next(iter(self._entries)))
      e = dict(static=tuple(torch.empty_like(t) for t in inputs), count=0, graph=None, out=None)
    self._entries[key] = e                            # most recently used = last
    for dst, src in zip(e["static"], inputs):
      # a caller that stages its batches straight into the graph's buffers (staging_buffers) passes them back:
      # nothing to copy
      if dst.data_ptr() != src.data_ptr():
        dst.copy_(src, non_blocking=True)
    if e["graph"] is not None:
      e["graph"].replay()
      self.replays += 1
      return e["out"]
    e["count"] += 1
    if e["count"] <= self.warmup:
      side = torch.cuda.Stream()
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):
        out = body(*e["static"])
      torch.cuda.current_stream().wait_stream(side)
      return out
    try:
      torch.cuda.synchronize()
      graph = torch.cuda.CUDAGraph()
      # thread_local: another thread (RCCL's watchdog) may touch the runtime while this one captures
      with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        out = body(*e["static"])
    except Exception as exc:       # keep training: this shape stays on eager launches
      torch.cuda.synchronize()
      self._entries.pop(key, None)
      self._failed.add(key)
      print("note: hipGraph capture failed for step shape %r (%s: %s); eager launches" % (key, type(exc).__name__, exc))
      return body(*inputs)
    graph.replay()                 # capture records, it does not run
    e["graph"], e["out"] = graph, out
    self.captures += 1
    return out


class RecurrenceWatch(object):
  """Keeps a run alive when the one-launch recurrences cannot run on this device.

  A pair / cluster recurrence needs all of its workgroups resident together (one per compute unit); when they are not
  — compute units held by another process or by a long foreign kernel — its members time out, the device-side fault
  word makes the step a skipped one (include/lipreading_hip.h), and the NEXT step would do the same, a few tenths
  of a second each: a run that skips every batch.  The reference's contract is skip a bad batch and keep training
  (src/train/train_better_model.py:49-50), so after `limit` consecutive faulted steps this switches every recurrence
  to the per-step kernels for the rest of the process (lr_rnn_one_launch_enable(0)), says so once, and drops the
  captured hipGraphs (`graphs`), whose launch sequences are stale.

  No host round trip in the step: after a step, the step's fault flag is exported on the stream and copied to pinned
  memory asynchronously; the copy is looked at once it has landed, a step or two later.  Only when the host has run
  more than `lag` steps ahead of an unobserved flag does it wait for it (the device still has those steps queued, so
  nothing idles: this bounds the host's run-ahead, as a healthy loop's own `.item()` reads would)."""

  def __init__(self, device, limit=3, lag=2):
    self.limit, self.run, self.tripped, self.lag = int(limit), 0, False, int(lag)
    n = self.lag + 2
    self._dev = torch.zeros(n, 2, dtype=torch.int32, device=device)
    self._host = torch.zeros(n, 2, dtype=torch.int32).pin_memory()
    self._pending = []          # (slot, event) of the steps not yet observed, oldest first
    self._step = 0

  def _observe(self, graphs):
    slot, _ = self._pending.pop(0)
    self.run = self.run + 1 if int(self._host[slot, 1]) < 0 else 0
    if self.run < self.limit:
      return False
    self.tripped = True
    del self._pending[:]
    torch.cuda.synchronize()
    _C.lib().lr_rnn_one_launch_enable(0)
    if graphs is not None:
      graphs.reset()
    import warnings
    warnings.warn("lipreading_amd: the one-launch recurrence timed out in %d consecutive steps (its workgroups were "
                  "not resident together: are this GPU's compute units shared?); every recurrence runs on the "
                  "per-step kernels from here on (slower, same results)" % self.run)
    return True

  def after_step(self, graphs=None):
    """Call once per optimisation step, after it was enqueued.  Returns True when the fall-back was just taken."""
    if self.tripped or not _C.lib().lr_rnn_one_launch_enabled():
      return False
    while self._pending:
      _, ev = self._pending[0]
      if not ev.query():
        if len(self._pending) <= self.lag:
          break                 # not landed yet, and the host is not far ahead: look again after the next step
        ev.synchronize()
      if self._observe(graphs):
        return True
    slot = self._step % self._dev.shape[0]
    self._step += 1
    _C.check(_C.lib().lr_fault_export(None, self._dev[slot].data_ptr(), _C.stream_handle()), "lr_fault_export")
    self._host[slot].copy_(self._dev[slot], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    self._pending.append((slot, ev))
    return False


def _roll_faults(device):
  """Outside a training step (eval, greedy decoding) nothing rolls the fault words: do it here, so that a time-out of
  an earlier batch does not mark every later one (lr_step_begin with an empty gradient buffer)."""
  _C.check(_C.lib().lr_step_begin(None, 0, None, _C.stream_handle()), "lr_step_begin")


def _fault_keep(flag2):
  """1 if no one-launch recurrence has timed out since the last roll, else 0 — a device scalar, no host read."""
  _C.check(_C.lib().lr_fault_export(None, flag2.data_ptr(), _C.stream_handle()), "lr_fault_export")
  return (flag2[1] == 0)


def ctc_step(encoder, opt, frames, frame_lens, chars, char_lens, grad_norm=None, max_len=None,
             grad_sync=None, graphs=None):
  """One encoder+CTC optimisation step with no host synchronisation.

  labels = chars[:,1:], label_lens = char_lens-1 (:31-32); encoder (:46); ctc_loss 'mean' (:48);
  zero_grad + backward (:67,:74); clip (:78); opt.step (:80).  `opt` is a FusedAdam; `grad_sync`
  (optional) is called between backward and the optimiser — the data-parallel all-reduce.
  Returns (loss, status) device tensors; status == 1 marks a batch the reference skips."""
  whole = grad_sync is None      # nothing between backward and the optimiser: one graph for the step

  def body(frames, frame_lens, chars, char_lens):
    # labels = chars[:, 1:] + 1, label_lens = char_lens - 1 and the int32 lengths in one launch; the encoder takes
    # the int32 frame lengths as they are
    fused_prep = (chars.dtype == torch.int64 and frame_lens.dtype == torch.int64 and char_lens.dtype == torch.int64
                  and chars.dim() == 2 and chars.stride(1) == 1)
    if fused_prep:   # ... and in the SAME launch as zero_grad (lr_step_begin_ctc)
      labels_p1, frame_lens32, label_lens32 = opt.zero_grad_and_prepare_ctc(chars, frame_lens, char_lens)
    else:
      opt.zero_grad()
    # (the CTC-only step never reads the encoder's final states: they are not extracted)
    log_probs, _, _ = encoder(frames, frame_lens32 if fused_prep else frame_lens, max_len=max_len,
                              need_final_state=False)
    if fused_prep:
      loss, status, _ = ctc_loss_prepared(log_probs, labels_p1, frame_lens32, label_lens32, 'mean')
    else:
      loss, status, _ = ctc_loss_with_status(log_probs, chars[:, 1:], frame_lens, char_lens - 1, 'mean')
    if grad_sync is not None and hasattr(grad_sync, "set_status"):
      grad_sync.set_status(status)   # (the bucket that carries the skip / fault words may leave during backward)
    loss.backward(_one(loss.device))
    if whole:
      opt.step(grad_norm=grad_norm, grad_scale=1.0, skip=status)
    return loss.detach(), status

  dev = frames.device
  inputs = (frames, frame_lens.to(dev), chars.to(dev), char_lens.to(dev))
  if graphs is None:
    loss, status = body(*inputs)
  else:
    # a graph bakes max_len in: it must be the padded length itself
    ml = frames.shape[1] if max_len is None else int(max_len)
    # (the learning rate is a by-value kernel argument: part of what a graph bakes in)
    key = ("ctc", id(encoder), id(opt), opt.lr, tuple(frames.shape), str(frames.dtype), tuple(chars.shape), ml,
           grad_norm, whole)
    with _Held((grad_sync,)):   # (an overlapping GradSync must not exchange from inside a capture)
      loss, status = graphs.run(key, inputs, body, capturable=(ml == frames.shape[1]))
  if not whole:
    scale = grad_sync(status)
    opt.step(grad_norm=grad_norm, grad_scale=scale, skip=status, dist_words=getattr(grad_sync, "dist_words", None),
             world=getattr(grad_sync, "world", 1))
  return loss, status


# test hook / switch: False keeps the CTC branch of a decoder step on the step's own stream
overlap_ctc_branch = True
_ctc_side = None


def _ctc_stream(device):
  global _ctc_side
  if _ctc_side is None or _ctc_side.device != device:
    _ctc_side = torch.cuda.Stream(device=device)
  return _ctc_side


def _step_all(opts, grad_norm, status):
  """The per-module clip + Adam of train_better_model.py:77-80 for opts = (encoder's, decoder's).  When the decoder
  loop's weight half ran on the side stream (attention_decoder.overlap_weight_half), its gradients were complete long
  before the encoder's: the decoder's optimiser goes to that stream as well — sum of squares + Adam are HBM-bound and run
  beside the encoder's weight-gradient products — and the streams are joined behind the encoder's optimiser."""
  from . import attention_decoder as _dec
  from . import encoder as _enc
  side_ok = len(opts) == 2 and _dec.take_split_flag() and _dec.side_stream() is not None
  if not side_ok:
    for o in opts:
      o.step(grad_norm=grad_norm, grad_scale=1.0, skip=status)
    return
  main = torch.cuda.current_stream()
  side = _dec.side_stream()
  with torch.cuda.stream(side):     # (the side stream already holds everything the decoder's gradients depend on)
    opts[1].step(grad_norm=grad_norm, grad_scale=1.0, skip=status)
  opts[0].step(grad_norm=grad_norm, grad_scale=1.0, skip=status)
  main.wait_stream(side)


def decoder_step(encoder, decoding_step, opts, frames, frame_lens, chars, char_lens, flags, seed, pad,
                 grad_norm=None, max_len=None, grad_sync=(None, None), graphs=None):
  """The reference's WHOLE step (train_better_model.py:46-80) for one batch, all on the device:
  encoder (:46) -> CTC 'mean' when the encoder has it (:48) -> decoder loop over len(flags) steps with
  the given teacher-forcing pattern (:54-63) -> NLL / non-PAD count (:65) -> one backward of the sum
  (:69,:74) -> per-module clip + Adam (:77-80).  opts = (encoder FusedAdam, decoder FusedAdam);
  grad_sync = (encoder GradSync | None, decoder GradSync | None).
  Returns (decoder_loss, ctc_loss | None, status | None) device tensors."""
  syncs = tuple(grad_sync)
  use_ctc = encoder.enable_ctc
  L = len(flags)
  whole = syncs[0] is None and syncs[1] is None

  def body(frames, frame_lens_d, chars, char_lens_d):
    labels = chars[:, 1:]
    status, total, ctc = None, 0, None
    # labels + 1, label_lens = char_lens - 1 and the int32 lengths in one launch (as ctc_step does) instead of four
    # ATen conversions — the launch that clears the encoder's gradients
    fused_prep = (use_ctc and chars.dtype == torch.int64 and frame_lens_d.dtype == torch.int64
                  and char_lens_d.dtype == torch.int64 and chars.dim() == 2 and chars.stride(1) == 1)
    if fused_prep:
      labels_p1, frame_lens32, label_lens32 = opts[0].zero_grad_and_prepare_ctc(chars, frame_lens_d, char_lens_d)
    for o in (opts[1:] if fused_prep else opts):
      o.zero_grad()
    ctc_side = None
    if use_ctc:
      # The CTC branch (head, loss — and, because autograd runs a node's backward on the stream of its forward, their
      # backward) on a stream of its own beside the decoder loop: both only read the encoder's states.
      if whole and overlap_ctc_branch and isinstance(encoder, _VideoEncoder):
        ctc_side = _ctc_stream(frames.device)
      log_probs, hidden, state = (encoder(frames, frame_lens32 if fused_prep else frame_lens_d, max_len=max_len,
                                          head_stream=ctc_side) if ctc_side is not None else
                                  encoder(frames, frame_lens32 if fused_prep else frame_lens_d, max_len=max_len))
      with (torch.cuda.stream(ctc_side) if ctc_side is not None else contextlib.nullcontext()):
        if fused_prep:
          ctc, status, _ = ctc_loss_prepared(log_probs, labels_p1, frame_lens32, label_lens32, 'mean')
        else:
          ctc, status, _ = ctc_loss_with_status(log_probs, labels, frame_lens_d, char_lens_d - 1, 'mean')
      total = ctc
    else:
      hidden, state = encoder(frames, frame_lens_d, max_len=max_len)
    log_probs_d, _, _ = decoding_step.decode_sequence(chars[:, :L], state, frame_lens32 if fused_prep else frame_lens_d, hidden,
                                                      teacher_forced=flags, seed=seed)
    decoder_loss = decoder_nll(log_probs_d, labels, pad)
    # one traversal from BOTH roots (the reference's two backward calls, train_better_model.py:70,74, add up to this);
    # no `decoder_loss + ctc` tensor: that sum was an elementwise launch of its own
    for sy in syncs:
      if sy is not None and hasattr(sy, "set_status"):
        sy.set_status(status)
    one = _one(decoder_loss.device)
    if ctc_side is not None:
      torch.cuda.current_stream().wait_stream(ctc_side)     # the loss and the status exist
    if use_ctc:
      torch.autograd.backward([decoder_loss, total], [one, one])
    else:
      decoder_loss.backward(one)
    if ctc_side is not None:
      # (the head's parameter gradients are written in place by its backward, on that stream: no AccumulateGrad node
      # that would make the engine join it)
      torch.cuda.current_stream().wait_stream(ctc_side)
    if whole:
      _step_all(opts, grad_norm, status)
    out = (decoder_loss.detach(),)
    return out + ((ctc.detach(), status) if use_ctc else ())

  dev = frames.device
  inputs = (frames, frame_lens.to(dev), chars.to(dev), char_lens.to(dev))
  if graphs is None:
    out = body(*inputs)
  else:
    ml = frames.shape[1] if max_len is None else int(max_len)
    key = ("dec", id(encoder), id(decoding_step), tuple((id(o), o.lr) for o in opts), tuple(frames.shape),
           str(frames.dtype), tuple(chars.shape), ml, L, grad_norm, whole, use_ctc)
    # only the all-teacher-forced loop has a launch sequence that does not depend on the coins; its
    # sampled tokens are not used by train(), so replaying the captured seed changes nothing
    with _Held(syncs):
      out = graphs.run(key, inputs, body, capturable=(all(flags) and ml == frames.shape[1]))
  decoder_loss = out[0]
  ctc, status = (out[1], out[2]) if use_ctc else (None, None)
  if not whole:
    # per-module clip (:77-79) on the all-reduced gradients; a batch the reference skips (:49-50)
    # updates nothing (with several ranks: only if every rank skipped, GradSync's MIN over ranks)
    for o, sync in zip(opts, syncs):
      scale = sync(status) if sync is not None else 1.0
      o.step(grad_norm=grad_norm, grad_scale=scale, skip=status, dist_words=getattr(sync, "dist_words", None),
             world=getattr(sync, "world", 1))
  return decoder_loss, ctc, status


def _decoder_losses(decoding_step, chars, labels, label_lens_host, frame_lens, hidden, state,
                    teacher_forcing_ratio, pad):
  """train_better_model.py:54-65: the decoder loop and its summed NLL.  The teacher-forcing coin
  of every step (:57) is drawn up front from the same host generator (torch.rand(1) per step), so
  all L steps go out as one enqueue; prev_output's multinomial (:63) is drawn on the device."""
  L = int(label_lens_host.max())
  flags = [bool(torch.rand(1) < teacher_forcing_ratio) for _ in range(L)]
  log_probs, sampled, _ = decoding_step.decode_sequence(chars[:, :L], state, frame_lens, hidden,
                                                        teacher_forced=flags)
  nll, _ = decoder_nll_sum(log_probs, labels, pad)
  return nll, sampled, L


def train(encoder, decoding_step, data_loader, opt, device, char2idx,
          teacher_forcing_ratio=1, grad_norm=None, grad_sync=None, graphs=None):
  """Assumes sequences begin with BOS and end with EOS; data_loader yields
  (frames f32, frame_lens i64, chars i64, char_lens i64) — train_better_model.py:7-86.

  decoding_step=None: encoder+CTC alone, `opt` a FusedAdam over the encoder (`grad_sync`: one
  distributed.GradSync or None).  With a CharDecodingStep `opt` is (encoder FusedAdam, decoder
  FusedAdam) and `grad_sync`, if given, (encoder GradSync, decoder GradSync): the reference clips the two
  modules separately (:77-79) and one Adam over both is the same update as two Adams.  The
  reference runs decoder_loss.backward(retain_graph) and then ctc_loss.backward() (:70,:74): two
  traversals of the encoder graph whose gradients add; here the two losses are summed and the
  encoder is traversed once — the same gradients up to fp32 addition order.

  `graphs` (optional, a StepGraphs): replay each batch shape's step as one hipGraph."""
  use_ctc = encoder.enable_ctc
  pad = char2idx[PAD]
  device = torch.device(device)
  if decoding_step is None:
    assert use_ctc, "without a decoding step the encoder must have enable_ctc=True"
    assert isinstance(opt, FusedAdam), "opt must be lipreading_amd.optim.FusedAdam"
    opts = (opt,)
  else:
    # data parallel: grad_sync = (encoder GradSync, decoder GradSync), one per flat buffer
    syncs = (None, None) if grad_sync is None else tuple(grad_sync)
    assert len(syncs) == 2, "with a decoding step grad_sync is (encoder GradSync, decoder GradSync)"
    opts = tuple(opt)
    assert len(opts) == 2 and all(isinstance(o, FusedAdam) for o in opts), \
        "opt must be (encoder FusedAdam, decoder FusedAdam)"
    decoding_step.train()
  encoder.train()
  ctc_sum = torch.zeros((), dtype=torch.float32, device=device)
  dec_sum = torch.zeros((), dtype=torch.float32, device=device)
  skipped_before = opts[0].skipped_steps() if device.type == "cuda" else 0
  watch = RecurrenceWatch(device) if device.type == "cuda" else None
  flag2 = torch.zeros(2, dtype=torch.int32, device=device) if device.type == "cuda" else None
  for frames, frame_lens, chars, char_lens in data_loader:
    _check_framing(chars, char_lens, frame_lens, char2idx, use_ctc)
    max_len = int(frame_lens.max()) if not frame_lens.is_cuda else None
    label_lens_host = (char_lens - 1).cpu()
    frames, chars = frames.to(device, non_blocking=True), chars.to(device, non_blocking=True)
    frame_lens_d, char_lens_d = frame_lens.to(device, non_blocking=True), char_lens.to(device, non_blocking=True)
    if decoding_step is None:
      loss, _ = ctc_step(encoder, opt, frames, frame_lens_d, chars, char_lens_d, grad_norm=grad_norm,
                         max_len=max_len, grad_sync=grad_sync, graphs=graphs)
      ctc_sum += loss  # a skipped batch contributes 0, as `continue` does at :49-50
      if watch is not None:
        watch.after_step(graphs)
      continue
    # the teacher-forcing coins (:57) and the sampler seed (:63) come from the host generator; drawn here,
    # outside anything that may be captured
    L = int(label_lens_host.max())
    flags = tuple(bool(torch.rand(1) < teacher_forcing_ratio) for _ in range(L))
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    decoder_loss, ctc, status = decoder_step(encoder, decoding_step, opts, frames, frame_lens_d, chars, char_lens_d,
                                             flags, seed, pad, grad_norm=grad_norm, max_len=max_len, grad_sync=syncs,
                                             graphs=graphs)
    # a recurrence time-out anywhere in the step (the decoder loop's included, which runs AFTER the CTC status was
    # written) leaves garbage in decoder_loss: the fault word masks it, with or without a CTC status
    keep = _fault_keep(flag2) if flag2 is not None else 1
    if status is None:
      dec_sum += torch.where(keep, decoder_loss, torch.zeros_like(decoder_loss)) if flag2 is not None else decoder_loss
    else:
      ok = (status.reshape(()) == 0) & keep if flag2 is not None else (status.reshape(()) == 0)
      dec_sum += torch.where(ok, decoder_loss, torch.zeros_like(decoder_loss))
      ctc_sum += ctc
    if watch is not None:
      watch.after_step(graphs)
  avg_ctc_loss = (ctc_sum / len(data_loader)).item()  # :84 divides by len(data_loader)
  avg_decoder_loss = (dec_sum / len(data_loader)).item()
  # batches that updated nothing this epoch — the reference's `continue` (:49-50) and, here, steps whose one-launch
  # recurrence timed out (the device-side fault word made CTC / Adam skip them) — read once per epoch
  global last_epoch_stats
  if device.type == "cuda":
    faults = int(_C.lib().lr_rnn_pair_errors())
    last_epoch_stats = {"batches": len(data_loader), "skipped": opts[0].skipped_steps() - skipped_before,
                        "recurrence_faults": faults}
    if last_epoch_stats["skipped"] or faults:
      print("\tSkipped batches: %(skipped)d of %(batches)d (recurrence time-outs: %(recurrence_faults)d)"
            % last_epoch_stats)
  if decoding_step is not None:
    print(f'\tTraining decoder_loss: {avg_decoder_loss}')
  if use_ctc:
    print(f'\tTraining ctc_loss: {avg_ctc_loss}')
  return avg_decoder_loss, avg_ctc_loss


def eval(encoder, decoding_step, data_loader, device, char2idx):
  """train_better_model.py:89-143: CTC 'sum' per batch (:116) averaged over len(data_loader)
  (:141); decoder loop teacher-forced on every step (:125), NLL summed and divided by the count of
  non-PAD labels (:138), `correct` counts multinomial samples equal to the label (:129-133).
  Returns (decoder_loss, correct, count, ctc_loss) — the reference's tuple plus the CTC average
  it computes and drops."""
  pad = char2idx[PAD]
  use_ctc = encoder.enable_ctc
  encoder.eval()
  if decoding_step is not None:
    decoding_step.eval()
  ctc_sum = torch.zeros((), dtype=torch.float32, device=device)
  dec_sum = torch.zeros((), dtype=torch.float32, device=device)
  correct = torch.zeros((), dtype=torch.int64, device=device)
  count = torch.zeros((), dtype=torch.float32, device=device)
  on_gpu = torch.device(device).type == "cuda"
  flag2 = torch.zeros(2, dtype=torch.int32, device=device) if on_gpu else None
  faulted = torch.zeros((), dtype=torch.int32, device=device)
  with torch.no_grad():
    for frames, frame_lens, chars, char_lens in data_loader:
      _check_framing(chars, char_lens, frame_lens, char2idx, use_ctc, in_eval=True)
      if on_gpu:
        _roll_faults(device)   # a time-out in an earlier batch must not mark this one
      max_len = int(frame_lens.max()) if not frame_lens.is_cuda else None
      label_lens_host = (char_lens - 1).cpu()
      frames, chars = frames.to(device), chars.to(device)
      frame_lens_d = frame_lens.to(device)
      labels = chars[:, 1:]
      keep = 1
      if use_ctc:
        log_probs, hidden, state = encoder(frames, frame_lens_d, max_len=max_len)
        loss, status, _ = ctc_loss_with_status(log_probs, labels, frame_lens_d,
                                               (char_lens - 1).to(device), 'sum')
        ctc_sum += loss
        keep = (status.reshape(()) == 0)  # :117-118 `continue`s past the decoder on a skipped batch
      else:
        hidden, state = encoder(frames, frame_lens_d, max_len=max_len)
      if decoding_step is None:
        if on_gpu:
          faulted += (~_fault_keep(flag2)).int()
        continue
      nll, sampled, L = _decoder_losses(decoding_step, chars, labels, label_lens_host, frame_lens_d,
                                        hidden, state, 2.0, pad)
      mask = labels[:, :L] != pad
      if on_gpu:
        # (a one-launch recurrence that timed out — the encoder's or the decoder loop's — left garbage: the batch counts
        # like one the reference `continue`s past, and is reported below)
        fk = _fault_keep(flag2)
        faulted += (~fk).int()
        keep = keep * fk
      dec_sum += torch.where(keep.bool(), nll, torch.zeros_like(nll)) if torch.is_tensor(keep) else nll
      correct += ((sampled.long() == labels[:, :L]) & mask).sum() * keep
      count += mask.sum().float() * keep
  if on_gpu:
    _roll_faults(device)
    nf = int(faulted.item())
    if nf:
      print("\tEvaluation batches dropped after a recurrence time-out: %d of %d" % (nf, len(data_loader)))
  ctc_avg = (ctc_sum / len(data_loader)).item()
  if decoding_step is None:
    return 0.0, 0, 0, ctc_avg
  count_h = count.item()
  return (dec_sum / count).item() if count_h else float('nan'), int(correct.item()), count_h, ctc_avg


def greedy_cer(encoder, data_loader, device, char2idx):
  """Character error rate of the CTC greedy path (decoder.py:64-73 on decoder.py:182-197
  output): sum of edit distances / sum of reference lengths.  The reference composes these
  only in its dead archived script (archive/train_model.py:351-357)."""
  from .decoder import GreedyDecoder, ctc_labels
  labels = ctc_labels(char2idx)
  dec = GreedyDecoder(labels, blank_index=0)
  inv = {v: k for k, v in char2idx.items()}
  encoder.eval()
  dist, total = 0, 0
  on_gpu = torch.device(device).type == "cuda"
  flag2 = torch.zeros(2, dtype=torch.int32, device=device) if on_gpu else None
  with torch.no_grad():
    for frames, frame_lens, chars, char_lens in data_loader:
      max_len = int(frame_lens.max()) if not frame_lens.is_cuda else None
      if on_gpu:
        # per batch, as eval() does: only THIS batch's time-out counts.  (Round 4 read lr_rnn_pair_errors() here —
        # pending + total, cleared by the read —, so one timed-out TRAINING step of the epoch made the first
        # validation batch look faulted and took the epoch's count away from whoever reads it next.)
        _roll_faults(device)
      log_probs, _, _ = encoder(frames.to(device), frame_lens.to(device), max_len=max_len)
      strings, _ = dec.decode(log_probs, frame_lens.to(device))
      # (decode() has just synchronised: reading the flag costs no wait of its own)
      if on_gpu and not bool(_fault_keep(flag2)):
        # The one-launch recurrence timed out: these strings are garbage, and val_cer drives save_best_model and the
        # annealing — decode this batch again on the per-step kernels instead of scoring it
        inner = getattr(encoder, "encoder", encoder)   # (PixelLipReader wraps the VideoEncoder)
        if hasattr(inner, "recurrence"):
          saved, inner.recurrence = inner.recurrence, 'f32'
          try:
            log_probs, _, _ = encoder(frames.to(device), frame_lens.to(device), max_len=max_len)
            strings, _ = dec.decode(log_probs, frame_lens.to(device))
          finally:
            inner.recurrence = saved
      for b in range(len(strings)):
        ref = ''.join(inv[int(c)] for c in chars[b, 1:int(char_lens[b]) - 1])  # strip BOS/EOS
        hyp = strings[b][0].replace(EOS, '')
        dist += dec.cer(hyp, ref)
        total += len(ref.replace(' ', ''))
  return dist / max(total, 1)
