"""Single-node data parallelism for the encoder+CTC step: one process per GPU, gradients summed
with RCCL all-reduce over xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

The reference lipreader is single-process (src/scripts/train.py:200-203 carries a REVIEW note
asking how to use several GPUs); its only distributed code is the unused vendored audio
trainer.  The path shards by independent samples along the batch: every rank takes a contiguous
slice of the length-sorted batch (so each shard keeps ascending frame_lens, ctc_loss.py:39) and
the one exchange per step is the gradient sum.

Design for xGMI (point-to-point links, a ring all-reduce is bound by ONE ~153 GB/s link and at
these sizes by latency): the gradients already live in one flat buffer
(optim.FlatParameters), so the exchange is a handful of large collectives over contiguous slices
— one bucket per autograd stage (output_proj, then each recurrent layer from last to first) —
each launched on a side HIP stream the moment its stage's gradients have been accumulated, so it
overlaps the remaining backward kernels.  The optimiser waits on the side stream, not the host.
"""
import torch
import torch.distributed as dist


def shard_batch(n_items, rank, world):
  """Contiguous slice [lo, hi) of a length-sorted batch for `rank`: keeps ascending order and
  keeps equal-length runs together as far as the split allows."""
  base, rem = divmod(n_items, world)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


class GradSync(object):
  """Bucketed gradient all-reduce over a FlatParameters buffer.

  buckets: by default one per top-level child module of `module` (registration order), which for
  VideoEncoder is [rnn (all layers), output_proj]; `per_param_groups` can refine that.  Works
  with any torch.distributed backend (gloo on CPU for tests, nccl/RCCL on the GPUs)."""

  def __init__(self, flat, groups=None, process_group=None, overlap=True):
    self.flat = flat
    self.pg = process_group
    self.world = dist.get_world_size(process_group)
    self.cuda = flat.grad.is_cuda
    self.overlap = overlap and self.cuda
    params = flat.params
    if groups is None:
      groups = [list(range(len(params)))]
    self.groups = groups
    self.bounds = []
    for g in groups:
      lo = flat.offsets[g[0]]
      last = g[-1]
      hi = flat.offsets[last + 1] if last + 1 < len(params) else flat.numel
      self.bounds.append((lo, hi))
    self._contiguous = all(self.bounds[i][1] == self.bounds[i + 1][0] for i in range(len(self.bounds) - 1))
    # The bucket that holds the FIRST parameter also carries the two words in front of it (optim.FlatParameters.words:
    # {ranks whose batch was skipped, ranks whose recurrence timed out}, written by lr_fault_export_f32 right before that
    # bucket goes out).  It is the last bucket of a backward — the first layer's gradients come last —, so every
    # recurrence of the step has been enqueued by then, and the sum over the ranks arrives with the gradients: no second
    # collective, no import kernel (round 4: a MIN all-reduce of two ints behind the last bucket, 0.04 ms of a step).
    self._words_bucket = None
    if self.cuda and getattr(flat, "first", 0) >= 2:
      for gi, (lo, hi) in enumerate(self.bounds):
        if lo == flat.first:
          self._words_bucket = gi
          self.bounds[gi] = (0, hi)
    self._order = []             # buckets in the order they went out this round
    self._status = None          # the step's status tensor, announced by set_status() before backward
    self._words_done = False     # the words of this round have been exported (with or without a status)
    self._words_have_status = False
    self.dist_words = None       # after __call__: flat.words when they carry the round's skip / fault sums, else None
    self._owner = {}
    for gi, g in enumerate(groups):
      for pi in g:
        self._owner[pi] = gi
    self._works = []
    self._reset_round()
    self.side = torch.cuda.Stream() if self.overlap else None
    self._hooks = []
    self._direct_hook = None
    self._held = 0
    self._closed = False
    # {status, -(fault pending)} of the step: one MIN all-reduce carries both (see __call__)
    self._pair = torch.zeros(2, dtype=torch.int32, device=flat.grad.device) if self.cuda else None
    if self.overlap:
      for pi, p in enumerate(params):
        self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(pi)))
      # gradients written in place by the HIP backward kernels bypass AccumulateGrad
      from . import encoder as _encoder
      index = {id(p): i for i, p in enumerate(params)}
      per_param = [self._make_hook(i, kind=1) for i in range(len(params))]

      def direct_hook(ready):
        for p in ready:
          i = index.get(id(p))
          if i is not None:
            per_param[i](p)
      self._direct_hook = direct_hook
      _encoder.grad_ready_hooks.append(direct_hook)

  @staticmethod
  def groups_for_encoder(encoder, flat):
    """[each recurrent layer's parameters] + [output_proj]: the order backward finishes them is
    the reverse of this list."""
    index = {id(p): i for i, p in enumerate(flat.params)}
    D = encoder.num_dirs
    groups = []
    for layer in range(encoder.num_layers):
      groups.append([index[id(p)] for p in encoder.rnn.layer_weights(layer, D)])
    if encoder.enable_ctc:
      groups.append([index[id(encoder.output_proj.weight)], index[id(encoder.output_proj.bias)]])
    return groups

  @staticmethod
  def groups_for_pixel_model(model, flat):
    """[the conv frontend's parameters], [every parameter of the encoder]: TWO buckets.  The encoder's bucket is
    complete when the first recurrent layer's weight gradients are — they are the last of the encoder's and run on
    its side stream beside the conv backward — so its all-reduce (21+ MB, the step's large one) goes out when no
    cluster recurrence is left in the step: an RCCL ring kernel never holds compute units while a one-launch
    recurrence, whose workgroups must all be resident together, is being placed, and it still has the whole conv
    backward (1.1 ms at the bench shape) to hide under.  (A bucket per recurrent layer, as groups_for_encoder cuts
    them, puts the upper layers' all-reduces right beside the lower layers' recurrence launches.)"""
    index = {id(p): i for i, p in enumerate(flat.params)}
    conv = [index[id(p)] for p in model.frontend.parameters()]
    enc = [index[id(p)] for p in model.encoder.parameters()]
    assert sorted(conv + enc) == list(range(len(flat.params))) and max(conv) < min(enc)
    return [conv, enc]

  def _make_hook(self, pi, kind=0):
    """kind 0: autograd's post-accumulate hook; kind 1: announced by a HIP backward that wrote the
    gradient in place.  A parameter whose gradient is written in place is announced by BOTH in the same
    backward (torch >= 2.x runs the post-accumulate hook even when the Function returned None), so
    a parameter counts as ready on the first announcement of either kind; the SAME kind arriving twice
    for a parameter is a second backward."""
    def hook(_param):
      if self._held or self._closed:
        return
      gi = self._owner[pi]
      self._count[kind][pi] += 1
      if self._count[kind][pi] > 1:
        if self._launched[gi]:
          # the bucket already holds the SUM over ranks: a gradient accumulated into it now would never
          # be exchanged and the ranks would silently diverge
          raise RuntimeError(
              "GradSync: a gradient of bucket %d arrived after the bucket was all-reduced (a second "
              "backward before sync()); run the extra backward passes under `with sync.hold():` so the "
              "buckets go out when sync() is called, or call sync() between them" % gi)
        return
      if self._ready[pi]:
        return
      self._ready[pi] = True
      self._pending[gi] -= 1
      if self._pending[gi] == 0:
        self._launch(gi)
    return hook

  def _reset_round(self):
    n = len(self.flat.params)
    self._pending = [len(g) for g in self.groups]
    self._launched = [False] * len(self.groups)
    self._order = []
    self._ready = [False] * n
    self._count = ([0] * n, [0] * n)

  def hold(self):
    """Context manager: gradients that become ready inside it launch nothing; every bucket is
    exchanged by the next sync() call instead.  For steps that run more than one backward before
    the optimiser (the reference's decoder_loss.backward(retain_graph) followed by
    ctc_loss.backward(), train_better_model.py:69,74)."""
    sync = self

    class _Hold(object):
      def __enter__(self):
        sync._held += 1

      def __exit__(self, *exc):
        sync._held -= 1
        return False
    return _Hold()

  def close(self):
    """Remove the parameter hooks and the entry in encoder.grad_ready_hooks (which otherwise keep
    this object, its flat buffer and the model alive for the life of the process)."""
    for h in self._hooks:
      h.remove()
    self._hooks = []
    if self._direct_hook is not None:
      from . import encoder as _encoder
      if self._direct_hook in _encoder.grad_ready_hooks:
        _encoder.grad_ready_hooks.remove(self._direct_hook)
      self._direct_hook = None
    self._closed = True

  def set_status(self, status):
    """Announce the step's status tensor (int32[1], device) BEFORE backward: the bucket that carries the words may go
    out from a gradient-ready hook, before __call__ hands the status over.  (train.ctc_step / decoder_step do.)"""
    self._status = status

  def _export_words(self, status):
    from . import _C
    _C.check(_C.lib().lr_fault_export_f32(_C.ptr(status), self.flat.words.data_ptr(), _C.stream_handle()),
             "lr_fault_export_f32")
    self._words_done = True
    self._words_have_status = status is not None

  def _launch(self, gi, status=None):
    if self._launched[gi]:
      return
    self._launched[gi] = True
    self._order.append(gi)
    lo, hi = self.bounds[gi]
    if gi == self._words_bucket and not self._words_done:
      self._export_words(status if status is not None else self._status)
    self._all_reduce(self.flat.grad[lo:hi])

  def _all_reduce(self, buf):
    if self.overlap:
      self.side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(self.side):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
    else:
      self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

  def __call__(self, status=None):
    """Finish the exchange; returns the gradient scale (1/world).  Called between backward and
    the optimiser (train.ctc_step's grad_sync).  On the GPU the step's skip / fault facts travel as two floats in
    front of the gradients (see __init__): afterwards `self.dist_words` holds their sums for FusedAdam.step(dist_words=,
    world=) — the step is skipped only if every rank's batch was skipped (a rank whose own batch was skipped
    contributed zero gradients), nobody updates if any rank's recurrence timed out.  `status` keeps THIS rank's value
    until the optimiser step: lr_adam_step / lr_clip_adam_step then overwrite it (and OR the fault word) with the
    ranks' verdict, so a caller that wants its own rank's code must read it before opt.step(); train.decoder_step's
    second sync re-exports the verdict the encoder's optimiser step wrote, which is why the encoder steps first.
    Where the words could not travel (CPU tensors; the status unknown when their bucket left) `status`
    becomes the MIN over ranks as in rounds 1-4 and dist_words is None."""
    st_in = status if status is not None else self._status
    # did a bucket leave from a gradient-ready hook AFTER the one that carries the words?  (see `folded` below)
    words_left_early = self._words_bucket in self._order and self._order[-1] != self._words_bucket
    if not any(self._launched) and self._contiguous and len(self.groups) > 1:
      # nothing has gone out yet (no-overlap mode — the step was a hipGraph replay, or ran under hold()): the
      # buckets tile one stretch of the flat buffer, so ONE all-reduce carries them all (a collective costs its
      # launch and latency whatever its size: five per step were +0.3 ms on the 2.7 ms pixel step)
      self._launched = [True] * len(self.groups)
      if self._words_bucket is not None and not self._words_done:
        self._export_words(st_in)
      self._all_reduce(self.flat.grad[self.bounds[0][0]:self.bounds[-1][1]])
    else:
      for gi in range(len(self.groups)):
        self._launch(gi, st_in)   # anything backward did not reach
    # the words travelled with the gradients unless there are none (CPU / gloo) or the bucket that carries them left
    # before anybody had announced the status (a caller that uses neither train.ctc_step nor set_status)
    folded = self._words_bucket is not None and self._words_done and (self._words_have_status or status is None)
    # ... and unless another bucket left AFTER the words' bucket: the words are a snapshot taken when their bucket was
    # exported, and a recurrence enqueued later (custom groups whose first-parameter bucket is not the last of backward)
    # could still raise this rank's fault word — lr_clip_adam_step decides from the summed words alone, so such a
    # time-out would go unseen.  The exchange after backward (below) reads the words as they are now.
    if words_left_early:
      folded = False
    word = None
    if not folded:
      word = status
      if self.cuda:
        # the legacy exchange: {status, -(fault pending)} in ONE MIN all-reduce — a rank whose one-launch recurrence timed
        # out (include/lipreading_hip.h, fault words) has put garbage into the gradient sum, so EVERY rank must skip the
        # update (MIN of -1/0), while a batch is skipped as such only if every rank skipped it (MIN of status)
        from . import _C
        word = self._pair
        _C.check(_C.lib().lr_fault_export(_C.ptr(status), word.data_ptr(), _C.stream_handle()), "lr_fault_export")
      if word is not None:
        if self.overlap:
          self.side.wait_stream(torch.cuda.current_stream())
          with torch.cuda.stream(self.side):
            dist.all_reduce(word, op=dist.ReduceOp.MIN, group=self.pg)
        else:
          self._works.append(dist.all_reduce(word, op=dist.ReduceOp.MIN, group=self.pg, async_op=True))
    if self.overlap:
      torch.cuda.current_stream().wait_stream(self.side)
    for w in self._works:
      w.wait()
    self._works = []
    if self.cuda and not folded:
      from . import _C
      _C.check(_C.lib().lr_fault_import(word.data_ptr(), _C.ptr(status), _C.stream_handle()), "lr_fault_import")
    self.dist_words = self.flat.words if folded else None
    self._status, self._words_done, self._words_have_status = None, False, False
    self._reset_round()
    return 1.0 / self.world

  def broadcast_parameters(self, src=0):
    """Start every rank from rank `src`'s weights."""
    dist.broadcast(self.flat.data, src=src, group=self.pg)
