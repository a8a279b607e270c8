"""Train / eval loops — the caller contract of `src/train/train_better_model.py` (train :7,
eval :89) on the MI355X path.

`train(encoder, decoding_step, data_loader, opt, device, char2idx, teacher_forcing_ratio,
grad_norm)` keeps the reference's argument meaning.  Differences, all about not stalling the
GPU: the framing asserts (:26-33) run on the host copies the loader hands over; the CTC loss
uses the device-side status flag instead of a `None` return, so a skipped batch costs no
device->host round trip (the optimiser kernel honours the flag); losses are accumulated on the
device and read once per epoch.  `decoding_step=None` runs the encoder+CTC path alone — the
path BASELINE.json's north_star names; the attention decoder loop (:56-65) is the "next" row.
"""
import torch
import torch.nn.functional as F

from .ctc import ctc_loss_with_status
from .data import BOS, EOS, PAD
from .optim import FusedAdam


def _check_framing(chars, char_lens, frame_lens, char2idx, use_ctc):
  """train_better_model.py:26-33, on host tensors."""
  chars, char_lens, frame_lens = chars.cpu(), char_lens.cpu(), frame_lens.cpu()
  assert (chars[:, 0].squeeze() == char2idx[BOS]).all()
  assert (chars.gather(1, (char_lens - 1).unsqueeze(dim=1)).squeeze() == char2idx[EOS]).all()
  if use_ctc:
    assert (frame_lens >= char_lens).all()  # otherwise ctc loss will produce inf
  labels = chars[:, 1:]
  label_lens = char_lens - 1
  assert (labels != char2idx[PAD]).sum() == label_lens.sum()
  assert (frame_lens[1:] - frame_lens[:-1] >= 0).all()  # ctc_loss.py:39


def ctc_step(encoder, opt, frames, frame_lens, chars, char_lens, grad_norm=None, max_len=None,
             grad_sync=None):
  """One encoder+CTC optimisation step with no host synchronisation.

  labels = chars[:,1:], label_lens = char_lens-1 (:31-32); encoder (:46); ctc_loss 'mean' (:48);
  zero_grad + backward (:67,:74); clip (:78); opt.step (:80).  `opt` is a FusedAdam; `grad_sync`
  (optional) is called between backward and the optimiser — the data-parallel all-reduce.
  Returns (loss, status) device tensors; status == 1 marks a batch the reference skips."""
  labels = chars[:, 1:]
  label_lens = char_lens - 1
  opt.zero_grad()
  log_probs, _, _ = encoder(frames, frame_lens, max_len=max_len)
  loss, status, _ = ctc_loss_with_status(log_probs, labels, frame_lens.to(log_probs.device),
                                         label_lens.to(log_probs.device), 'mean')
  loss.backward()
  scale = 1.0
  if grad_sync is not None:
    scale = grad_sync(status)
  opt.step(grad_norm=grad_norm, grad_scale=scale, skip=status)
  return loss.detach(), status


def train(encoder, decoding_step, data_loader, opt, device, char2idx,
          teacher_forcing_ratio=1, grad_norm=None, grad_sync=None):
  """Assumes sequences begin with BOS and end with EOS; data_loader yields
  (frames f32, frame_lens i64, chars i64, char_lens i64) — train_better_model.py:7-16."""
  use_ctc = encoder.enable_ctc
  if decoding_step is not None:
    raise NotImplementedError("the attention CharDecodingStep loop (train_better_model.py:56-65) "
                              "is the N1 'next' row; pass decoding_step=None for encoder+CTC")
  assert use_ctc, "without a decoding step the encoder must have enable_ctc=True"
  assert isinstance(opt, FusedAdam), "opt must be lipreading_amd.optim.FusedAdam"
  encoder.train()
  ctc_sum = torch.zeros((), dtype=torch.float32, device=device)
  for frames, frame_lens, chars, char_lens in data_loader:
    _check_framing(chars, char_lens, frame_lens, char2idx, use_ctc)
    max_len = int(frame_lens.max()) if not frame_lens.is_cuda else None
    frames, chars = frames.to(device), chars.to(device)
    frame_lens_d, char_lens_d = frame_lens.to(device), char_lens.to(device)
    loss, _ = ctc_step(encoder, opt, frames, frame_lens_d, chars, char_lens_d, grad_norm=grad_norm,
                       max_len=max_len, grad_sync=grad_sync)
    ctc_sum += loss  # a skipped batch contributes 0, as `continue` does at :49-50
  avg_ctc_loss = (ctc_sum / len(data_loader)).item()  # :84 divides by len(data_loader)
  print(f'\tTraining ctc_loss: {avg_ctc_loss}')
  return 0.0, avg_ctc_loss


def eval(encoder, decoding_step, data_loader, device, char2idx):
  """train_better_model.py:89-143 for the encoder+CTC path: CTC 'sum' per batch (:116) averaged
  over len(data_loader) (:141).  Returns (decoder_loss, correct, count, ctc_loss); the first
  three are the reference's tuple and stay 0 without a decoding step."""
  if decoding_step is not None:
    raise NotImplementedError("CharDecodingStep evaluation is the N1 'next' row")
  encoder.eval()
  ctc_sum = torch.zeros((), dtype=torch.float32, device=device)
  with torch.no_grad():
    for frames, frame_lens, chars, char_lens in data_loader:
      _check_framing(chars, char_lens, frame_lens, char2idx, False)
      max_len = int(frame_lens.max()) if not frame_lens.is_cuda else None
      frames, chars = frames.to(device), chars.to(device)
      log_probs, _, _ = encoder(frames, frame_lens.to(device), max_len=max_len)
      loss, _, _ = ctc_loss_with_status(log_probs, chars[:, 1:], frame_lens.to(device),
                                        (char_lens - 1).to(device), 'sum')
      ctc_sum += loss
  return 0.0, 0, 0, (ctc_sum / len(data_loader)).item()


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
  with torch.no_grad():
    for frames, frame_lens, chars, char_lens in data_loader:
      max_len = int(frame_lens.max()) if not frame_lens.is_cuda else None
      log_probs, _, _ = encoder(frames.to(device), frame_lens.to(device), max_len=max_len)
      strings, _ = dec.decode(log_probs, frame_lens.to(device))
      for b in range(len(strings)):
        ref = ''.join(inv[int(c)] for c in chars[b, 1:int(char_lens[b]) - 1])  # strip BOS/EOS
        hyp = strings[b][0].replace(EOS, '')
        dist += dec.cer(hyp, ref)
        total += len(ref.replace(' ', ''))
  return dist / max(total, 1)
