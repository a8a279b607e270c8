#!/usr/bin/env python
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (build container only).

  python tests/golden/make_golden.py            # needs /root/reference; writes next to itself

The reference cannot travel to the GPU box, so the vectors it produces are committed as data.
Nothing here is reference source: the reference modules are imported from /root/reference and
called; this file holds only input generation and the stand-ins listed below.

Pins (see oracle/torch_oracle.py header):
  ctc_cases.npz     reference src/train/ctc_loss.py imported as-is (torch only)        CLEAN
  greedy_cases.npz  reference src/models/lipreader/decoder.py GreedyDecoder; the import needs
                    a `Levenshtein` module, stubbed inert (only wer/cer use it)         CLEAN
  enc_cases.npz     reference VideoEncoder (better_model.py).  Its import needs `spacy`
                    (inert stub) and the unpinned third-party `allennlp.nn.util`; the three
                    functions used are stood in below from allennlp's published definitions.
                    sort_batch_by_length cannot change results (undone at better_model.py:84-89);
                    masked_log_softmax's 1e-45 constant is the only observable part.  SHIMMED
  step_cases.npz    reference VideoEncoder + reference ctc_loss composed as
                    train_better_model.py:31-32,46-48,74-80 (CTC-only step)            SHIMMED
  dec_cases.npz     reference VideoEncoder + CharDecodingStep (five attention types) driven as
                    train_better_model.py:46-74 does at teacher_forcing_ratio = 1; needs the
                    allennlp stand-ins incl. masked_softmax                              SHIMMED
  enc2_cases.npz    as enc_cases for rnn_type='RNN' (tanh)                               SHIMMED
  dec2_cases.npz    as dec_cases for decoders with num_layers 2 / 3 and the tanh RNN     SHIMMED
  lmk_cases.npz     reference face.py _applyPadding / extractFace / getFace, imported with inert
                    dlib / tensorflow / skimage stubs (none on the arithmetic path)     CLEAN
  prn_cases.npz     reference PRN.process + get_landmarks (prnet.py:112-170) composed as
                    generate_dataview.py:58-64, network output replaced by a seeded position map;
                    skimage's estimate_transform stood in by oracle.umeyama_similarity  SHIMMED
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("LIPREADING_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
V = 64          # vocab incl. 4 markers (data_loader.py:33,35)
VP = V + 1      # CTC classes (better_model.py:38)


def install_stubs():
  # -- inert: not on the arithmetic path
  sys.modules.setdefault("spacy", types.ModuleType("spacy"))
  lev = types.ModuleType("Levenshtein")
  lev.distance = lambda a, b: (_ for _ in ()).throw(NotImplementedError("stub"))
  sys.modules.setdefault("Levenshtein", lev)
  # -- allennlp.nn.util stand-in (published definitions, allennlp 0.7-0.9)
  def masked_log_softmax(vector, mask, dim=-1):
    if mask is not None:
      mask = mask.float()
      while mask.dim() < vector.dim():
        mask = mask.unsqueeze(1)
      vector = vector + (mask + 1e-45).log()
    return torch.nn.functional.log_softmax(vector, dim=dim)

  def masked_softmax(vector, mask, dim=-1):
    mask = mask.float()
    while mask.dim() < vector.dim():
      mask = mask.unsqueeze(1)
    result = torch.nn.functional.softmax(vector * mask, dim=dim) * mask
    return result / (result.sum(dim=dim, keepdim=True) + 1e-13)

  def sort_batch_by_length(tensor, sequence_lengths):
    sorted_lens, perm = sequence_lengths.sort(0, descending=True)
    sorted_tensor = tensor.index_select(0, perm)
    index_range = torch.arange(0, len(sequence_lengths), device=sequence_lengths.device)
    _, reverse_mapping = perm.sort(0, descending=False)
    restoration = index_range.index_select(0, reverse_mapping)
    return sorted_tensor, sorted_lens, restoration, perm

  a = types.ModuleType("allennlp")
  a_nn = types.ModuleType("allennlp.nn")
  a_util = types.ModuleType("allennlp.nn.util")
  a_util.masked_log_softmax = masked_log_softmax
  a_util.masked_softmax = masked_softmax
  a_util.sort_batch_by_length = sort_batch_by_length
  a.nn, a_nn.util = a_nn, a_util
  sys.modules.update({"allennlp": a, "allennlp.nn": a_nn, "allennlp.nn.util": a_util})


def rand_log_probs(g, B, T):
  return torch.log_softmax(torch.randn(B, T, VP, generator=g), dim=-1)


def make_labels(g, lens, width=None, lo=4, hi=V):
  """ids uniform in [lo,hi) then EOS=2 as the last label (train_better_model.py:27,31)."""
  width = width or max(max(lens), 1)
  lab = torch.zeros(len(lens), width, dtype=torch.long)
  for i, n in enumerate(lens):
    if n > 0:
      lab[i, :n - 1] = torch.randint(lo, hi, (n - 1,), generator=g)
      lab[i, n - 1] = 2
  return lab


def gen_ctc(ref_ctc):
  g = torch.Generator().manual_seed(123456)  # the reference's default seed (train.py:165)
  cases = {}

  def add(name, lp, labels, frame_lens, label_lens):
    rec = {"lp": lp.numpy(), "labels": labels.numpy(), "frame_lens": np.asarray(frame_lens),
           "label_lens": np.asarray(label_lens)}
    for red in ("mean", "sum"):
      x = lp.clone().requires_grad_(True)
      loss = ref_ctc(x, labels, torch.tensor(frame_lens), torch.tensor(label_lens), red, "cpu")
      if loss is None:
        rec["none_" + red] = np.array(1)
        rec["loss_" + red] = np.array(np.nan, dtype=np.float32)
        rec["grad_" + red] = np.zeros_like(lp.numpy())
      else:
        loss.backward()
        rec["none_" + red] = np.array(0)
        rec["loss_" + red] = loss.detach().numpy().astype(np.float32)
        rec["grad_" + red] = x.grad.numpy()
    for k, v in rec.items():
      cases["%s/%s" % (name, k)] = v

  # equal lengths: one run, plain torch 'mean'
  add("equal", rand_log_probs(g, 4, 20), make_labels(g, [6, 6, 6, 6]), [20] * 4, [6] * 4)
  # run sizes (1,1,2): the SURVEY A4 quirk probe
  add("runs112", rand_log_probs(g, 4, 24), make_labels(g, [5, 7, 6, 8]), [16, 20, 24, 24],
      [5, 7, 6, 8])
  # run sizes (2,3,3)
  add("runs233", rand_log_probs(g, 8, 30), make_labels(g, [4, 9, 7, 7, 10, 3, 12, 6]),
      [18, 18, 25, 25, 25, 30, 30, 30], [4, 9, 7, 7, 10, 3, 12, 6])
  # repeated characters force the blank between equal neighbours
  lab = make_labels(g, [7, 7, 7])
  lab[0, :6] = torch.tensor([10, 10, 10, 11, 11, 10])
  lab[1, :6] = torch.tensor([20, 20, 20, 20, 20, 20])
  add("repeats", rand_log_probs(g, 3, 22), lab, [22] * 3, [7] * 3)
  # a label longer than 256 is dropped (ctc_loss.py:46)
  ll = [5, 300, 6]
  add("toolong", rand_log_probs(g, 3, 14), make_labels(g, ll, width=300), [12, 14, 14], ll)
  # every label too long -> None
  add("alltoolong", rand_log_probs(g, 2, 8), make_labels(g, [257, 260], width=260), [8, 8],
      [257, 260])
  # an inf sample (label longer than its frames) inside a run with finite ones
  ll = [4, 9, 4, 5]
  add("infsample", rand_log_probs(g, 4, 10), make_labels(g, ll), [6, 6, 10, 10], ll)
  # a run that is entirely inf is skipped without advancing prev_change_point (:92 vs :107)
  ll = [3, 8, 8, 4, 5]
  add("infrun", rand_log_probs(g, 5, 12), make_labels(g, ll), [5, 6, 6, 12, 12], ll)
  # first run entirely inf, later runs fine
  ll = [9, 3, 4]
  add("inffirst", rand_log_probs(g, 3, 12), make_labels(g, ll), [4, 10, 12], ll)
  # everything inf -> None
  ll = [9, 9]
  add("allinf", rand_log_probs(g, 2, 5), make_labels(g, ll), [5, 5], ll)
  # the benchmark shape, reduced batch: T=75, L=30+EOS
  add("bench", rand_log_probs(g, 4, 75), make_labels(g, [31] * 4), [75] * 4, [31] * 4)
  # mixed lengths at the benchmark T, more than 64 states for one sample (L=40 -> 81 states)
  ll = [12, 31, 40, 25, 31, 18]
  add("mixed75", rand_log_probs(g, 6, 75), make_labels(g, ll), [45, 45, 60, 60, 75, 75], ll)
  np.savez_compressed(os.path.join(OUT, "ctc_cases.npz"), **cases)
  return cases


def gen_greedy(ref_decoder_mod, char2idx):
  g = torch.Generator().manual_seed(123456)
  inv = {v: k for k, v in char2idx.items()}
  labels = ['_'] + [inv[i] for i in range(len(inv))]
  dec = ref_decoder_mod.GreedyDecoder(labels, blank_index=0)
  cases = {}
  B, T = 6, 40
  lp = rand_log_probs(g, B, T)
  # engineer repeats / blanks / spaces: a,_,a -> "aa"; a,a -> "a"; include the space class (5)
  path = torch.randint(0, VP, (B, T), generator=g)
  path[0, :8] = torch.tensor([40, 40, 0, 40, 0, 0, 5, 5])
  path[1, :6] = torch.tensor([0, 0, 41, 41, 41, 0])
  path[2, :] = 0
  path[3, :] = 45
  # classes 1..4 are the multi-character marker names; PAD/BOS are masked in the live model
  path[path == 1] = 0
  path[path == 2] = 0
  path[path == 3] = 50
  path[path == 4] = 51
  lp.scatter_(2, path.unsqueeze(-1), 1.0)  # make `path` the argmax
  sizes = torch.tensor([40, 33, 40, 17, 1, 25])
  strings, offsets = dec.decode(lp, sizes)
  cases["lp"] = lp.numpy()
  cases["sizes"] = sizes.numpy()
  cases["labels"] = np.array(labels, dtype=object)
  cases["strings"] = np.array([s[0] for s in strings], dtype=object)
  for b in range(B):
    cases["offsets_%d" % b] = offsets[b][0].numpy()
  # no sizes -> full length
  strings2, _ = dec.decode(lp)
  cases["strings_nosizes"] = np.array([s[0] for s in strings2], dtype=object)
  np.savez_compressed(os.path.join(OUT, "greedy_cases.npz"), **cases)


def enc_inputs(g, B, T, lens, scale):
  frames = torch.randn(B, T, 68, 3, generator=g) * scale
  for b, n in enumerate(lens):
    frames[b, n:] = 0  # _collate_fn zero-pads (data_loader.py:137)
  return frames


ENC_CFGS = [("gru_bi", "GRU", True, 1, 12), ("gru_uni", "GRU", False, 1, 8),
            ("lstm_bi", "LSTM", True, 1, 12), ("lstm_uni", "LSTM", False, 1, 8),
            ("gru_bi_l2", "GRU", True, 2, 8), ("lstm_bi_l2", "LSTM", True, 2, 8)]
# rnn_type='RNN' (tanh; better_model.py:9 allows it) — a second file so enc_cases.npz stays byte-identical
ENC2_CFGS = [("rnn_bi", "RNN", True, 1, 12), ("rnn_uni_l2", "RNN", False, 2, 8), ("rnn_bi_l2", "RNN", True, 2, 8)]


def gen_enc(bm, char2idx, cfgs=ENC_CFGS, out="enc_cases.npz"):
  g = torch.Generator().manual_seed(123456)
  cases = {}
  for name, rnn_type, bi, layers, H in cfgs:
    torch.manual_seed(123456)
    enc = bm.VideoEncoder(204, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bi,
                          enable_ctc=True, vocab_size=V, char2idx=char2idx)
    enc.eval()
    for tag, lens in (("eq", [14, 14, 14, 14]), ("mix", [5, 9, 14, 14])):
      frames = enc_inputs(g, 4, 14, lens, 1.0)
      with torch.no_grad():
        lp, hid, fin = enc(frames, torch.tensor(lens))
      key = "%s/%s" % (name, tag)
      cases[key + "/frames"] = frames.numpy()
      cases[key + "/lens"] = np.asarray(lens)
      cases[key + "/log_probs"] = lp.numpy()
      cases[key + "/hidden"] = hid.numpy()
      if isinstance(fin, tuple):
        cases[key + "/h_n"], cases[key + "/c_n"] = fin[0].numpy(), fin[1].numpy()
      else:
        cases[key + "/h_n"] = fin.numpy()
    for k, v in enc.state_dict().items():
      cases["%s/sd/%s" % (name, k)] = v.numpy()
    cases[name + "/cfg"] = np.array([H, layers, int(bi)])
  np.savez_compressed(os.path.join(OUT, out), **cases)


def gen_step(bm, ref_ctc, char2idx):
  """One CTC-only optimisation step: train_better_model.py:31-32,46-48,74,78,80 with
  Adam(lr) as built at train.py:280."""
  g = torch.Generator().manual_seed(123456)
  cases = {}
  for name, rnn_type, H in (("gru", "GRU", 12), ("lstm", "LSTM", 12)):
    torch.manual_seed(123456)
    enc = bm.VideoEncoder(204, H, rnn_type=rnn_type, num_layers=1, bidirectional=True,
                          enable_ctc=True, vocab_size=V, char2idx=char2idx)
    enc.train()
    lens = [12, 12, 16, 20, 20, 20]
    frames = enc_inputs(g, 6, 20, lens, 1.0)
    cl = [6, 8, 7, 9, 5, 10]                       # char_lens incl. BOS and EOS
    chars = torch.zeros(6, max(cl), dtype=torch.long)
    for i, n in enumerate(cl):
      chars[i, 0] = 1
      chars[i, 1:n - 1] = torch.randint(4, V, (n - 2,), generator=g)
      chars[i, n - 1] = 2
    for k, v in enc.state_dict().items():
      cases["%s/sd0/%s" % (name, k)] = v.clone().numpy()
    opt = torch.optim.Adam(enc.parameters(), lr=1e-3)
    labels, label_lens = chars[:, 1:], torch.tensor(cl) - 1
    lp, _, _ = enc(frames, torch.tensor(lens))
    loss = ref_ctc(lp, labels, torch.tensor(lens), label_lens, 'mean', 'cpu')
    opt.zero_grad()
    loss.backward()
    for k, p in enc.named_parameters():
      cases["%s/grad/%s" % (name, k)] = p.grad.clone().numpy()
    total_norm = torch.nn.utils.clip_grad_norm_(enc.parameters(), 5.0)
    opt.step()
    for k, v in enc.state_dict().items():
      cases["%s/sd1/%s" % (name, k)] = v.clone().numpy()
    cases[name + "/frames"] = frames.numpy()
    cases[name + "/lens"] = np.asarray(lens)
    cases[name + "/chars"] = chars.numpy()
    cases[name + "/char_lens"] = np.asarray(cl)
    cases[name + "/loss"] = loss.detach().numpy()
    cases[name + "/total_norm"] = np.asarray(float(total_norm), dtype=np.float32)
    cases[name + "/cfg"] = np.array([H, 1, 1])
  np.savez_compressed(os.path.join(OUT, "step_cases.npz"), **cases)


DEC_CFGS = [("gru_1layernn", "GRU", True, "1_layer_nn", -1), ("lstm_dot", "LSTM", True, "dot", -1),
            ("gru_general", "GRU", False, "general", -1), ("lstm_concat", "LSTM", False, "concat", 10),
            ("gru_none", "GRU", True, "none", -1)]
# decoders with num_layers > 1 (better_model.py:136,147-148) and the tanh RNN; second file, as above
DEC2_CFGS = [("gru_l2_1layernn", "GRU", True, "1_layer_nn", -1, 2), ("lstm_l2_dot", "LSTM", False, "dot", -1, 2),
             ("rnn_l1_general", "RNN", True, "general", -1, 1), ("rnn_l3_none", "RNN", False, "none", -1, 3),
             ("lstm_l3_concat", "LSTM", True, "concat", 6, 3)]


def gen_dec(bm, ref_ctc, char2idx, cfgs=DEC_CFGS, out_file="dec_cases.npz"):
  """Reference VideoEncoder -> reference CharDecodingStep, driven as train_better_model.py:46-65
  does with teacher_forcing_ratio=1 (decoder loss), plus the CTC loss; gradients of
  decoder_loss.backward(retain_graph) followed by ctc_loss.backward() (:69,:74).  SHIMMED."""
  g = torch.Generator().manual_seed(123456)
  cases = {}
  for cfg in cfgs:
    name, rnn_type, bi, attn, ah = cfg[:5]
    layers = cfg[5] if len(cfg) > 5 else 1
    torch.manual_seed(123456)
    H = 8
    enc = bm.VideoEncoder(204, H, rnn_type=rnn_type, num_layers=layers, bidirectional=bi, enable_ctc=True,
                          vocab_size=V, char2idx=char2idx)
    dec = bm.CharDecodingStep(enc, char_dim=12, vocab_size=V, char2idx=char2idx, attention_type=attn,
                              attn_hidden_size=ah)
    enc.train(); dec.train()
    lens = [9, 11, 14, 14]
    frames = enc_inputs(g, 4, 14, lens, 1.0)
    cl = [5, 7, 6, 8]
    chars = torch.zeros(4, max(cl), dtype=torch.long)
    for i, n in enumerate(cl):
      chars[i, 0] = 1
      chars[i, 1:n - 1] = torch.randint(4, V, (n - 2,), generator=g)
      chars[i, n - 1] = 2
    char_lens = torch.tensor(cl)
    labels, label_lens = chars[:, 1:], char_lens - 1
    lp_enc, hid, prev_state = enc(frames, torch.tensor(lens))
    ctc = ref_ctc(lp_enc, labels, torch.tensor(lens), label_lens, 'mean', 'cpu')
    dec_loss, outs = 0, []
    for i in range(int(label_lens.max())):
      out, prev_state = dec(chars[:, i], prev_state, torch.tensor(lens), hid)
      dec_loss = dec_loss + torch.nn.functional.nll_loss(out, labels[:, i], ignore_index=0, reduction='sum')
      outs.append(out)
    dec_loss = dec_loss / (labels != 0).sum()
    dec_loss.backward(retain_graph=True)
    ctc.backward()
    for k, v in enc.state_dict().items():
      cases["%s/enc_sd/%s" % (name, k)] = v.clone().numpy()
    for k, v in dec.state_dict().items():
      cases["%s/dec_sd/%s" % (name, k)] = v.clone().numpy()
    for k, p in enc.named_parameters():
      cases["%s/enc_grad/%s" % (name, k)] = p.grad.clone().numpy()
    for k, p in dec.named_parameters():
      cases["%s/dec_grad/%s" % (name, k)] = (p.grad.clone() if p.grad is not None else torch.zeros_like(p)).numpy()
    cases[name + "/frames"] = frames.numpy()
    cases[name + "/lens"] = np.asarray(lens)
    cases[name + "/chars"] = chars.numpy()
    cases[name + "/char_lens"] = np.asarray(cl)
    cases[name + "/dec_loss"] = dec_loss.detach().numpy()
    cases[name + "/ctc_loss"] = ctc.detach().numpy()
    cases[name + "/dec_log_probs"] = torch.stack(outs, 1).detach().numpy()
    cases[name + "/cfg"] = np.array([H, int(bi), 12, ah] + ([layers] if len(cfg) > 5 else []))
  np.savez_compressed(os.path.join(OUT, out_file), **cases)


def install_face_stubs():
  """What face.py / prnet.py need at IMPORT time and this container lacks.  Inert (never on the
  arithmetic path): dlib, tensorflow(+contrib), skimage.io, skimage.transform.warp (its result only
  feeds the network, which the fixtures replace by a fixed position map).  NOT inert:
  skimage.transform.estimate_transform (scikit-image 0.14.1) — stood in by the oracle's restatement of
  the Umeyama algorithm, which makes every prn_* vector a SHIMMED pin."""
  repo = os.path.dirname(os.path.dirname(OUT))
  if repo not in sys.path:
    sys.path.insert(0, repo)
  from oracle import torch_oracle as O
  class _Inert(types.ModuleType):
    """Any attribute is another inert object (prnet.py names tf.nn.relu / tcl.batch_norm as default
    arguments at import time); calling one is an error."""
    def __getattr__(self, item):
      if item.startswith("__"):
        raise AttributeError(item)
      return _Inert(self.__name__ + "." + item)

    def __call__(self, *a, **k):
      raise NotImplementedError("inert stub %s called" % self.__name__)
  for name in ("dlib", "tensorflow", "tensorflow.contrib", "tensorflow.contrib.layers",
               "tensorflow.contrib.framework", "skimage", "skimage.io"):
    sys.modules.setdefault(name, _Inert(name))
  sys.modules.setdefault("skimage.transform", types.ModuleType("skimage.transform"))

  class _Tform(object):
    def __init__(self, params):
      self.params, self.inverse = params, None

  def estimate_transform(kind, src, dst):
    assert kind == 'similarity'
    return _Tform(O.umeyama_similarity(src, dst))
  sys.modules["skimage.transform"].estimate_transform = estimate_transform
  sys.modules["skimage.transform"].warp = lambda image, inv, output_shape=None: np.zeros(tuple(output_shape) + (3,))


def gen_lmk(face):
  """Reference face.py:_applyPadding (:76-90), extractFace (:103-115) and getFace (:164-175) RUN
  (inert dlib / tensorflow / skimage stubs, see install_face_stubs).  CLEAN pin."""
  rng = np.random.RandomState(123456)
  dims = np.array([[480, 640, 3], [100, 100, 3], [720, 1280, 3], [360, 480, 3], [1080, 1920, 3]])
  rects = np.array([[200, 300, 100, 220], [5, 95, 10, 90], [1000, 1270, 600, 715], [0, 37, 3, 41],
                    [700, 1211, 150, 661]])
  padding = 0.3
  lmk = np.stack([np.stack([rng.uniform(0, d[1], 68), rng.uniform(0, d[0], 68), rng.uniform(-60, 60, 68)], 1)
                  for d in dims])
  lmk = np.round(lmk * 64) / 64     # exact in float32 too: the translation is then bit-exact in either width
  padded, faces, crops = [], [], []
  for d, r, l in zip(dims, rects, lmk):
    p = face._applyPadding(tuple(int(v) for v in d), tuple(int(v) for v in r), padding)
    img = np.zeros(tuple(int(v) for v in d), dtype=np.uint8)
    crop, p2 = face.extractFace(img, tuple(int(v) for v in r), padding=padding)
    assert tuple(p2) == tuple(p)
    padded.append(p)
    crops.append(crop.shape[:2])
    faces.append(face.getFace(l, tuple(int(v) for v in p)))
  cases = {"dims": dims, "rects": rects, "padding": np.array(padding), "padded": np.array(padded),
           "crop_hw": np.array(crops), "lmk": lmk, "face": np.stack(faces)}
  np.savez_compressed(os.path.join(OUT, "lmk_cases.npz"), **cases)


def prn_position_map(seed, rect, res=256):
  """A synthetic network output: what PosPrediction.predict returns (prnet.py:304-309: net output x
  256*1.1, float32) — smooth-ish coordinates inside the 256 x 256 crop plus noise.  tests regenerate
  it from the seed (RandomState is a frozen legacy generator)."""
  rng = np.random.RandomState(seed)
  u, v = np.meshgrid(np.arange(res, dtype=np.float64), np.arange(res, dtype=np.float64))
  pos = np.stack([u + rng.uniform(-20, 20, (res, res)), v + rng.uniform(-20, 20, (res, res)),
                  rng.uniform(-80, 80, (res, res))], -1)
  return pos.astype(np.float32)


def gen_prn(face, prnet):
  """The reference's PRN.process (prnet.py:77-159; crop geometry :112-140, restore :150-156) and
  get_landmarks (:162-170) bodies, composed as generate_dataview.py:58-64 does (_gen_data: unpadded
  rect -> PRNet, padded rect -> getFace).  The PRN object is built without __init__ (no weights, no
  TensorFlow) and its net_forward returns prn_position_map(seed).  SHIMMED pin: estimate_transform."""
  uv = np.loadtxt(os.path.join(REF, "src/models/extern/prnet/Data/uv/uv_kpt_ind.txt")).astype(np.int32)  # 2 x 68 (data)
  cases = {"uv_kpt_ind": uv}
  dims = np.array([[480, 640, 3], [720, 1280, 3], [360, 480, 3], [1080, 1920, 3]])
  rects = np.array([[200, 300, 100, 220], [1000, 1270, 600, 715], [0, 37, 3, 41], [700, 1211, 150, 661]])
  seeds = np.array([11, 12, 13, 14])
  tforms, kpts, faces, padded, sub = [], [], [], [], []
  for d, r, seed in zip(dims, rects, seeds):
    prn = prnet.PRN.__new__(prnet.PRN)
    prn.resolution_inp = prn.resolution_op = 256
    prn.uv_kpt_ind = uv
    cropped = prn_position_map(int(seed), r)
    prn.net_forward = lambda image, _c=cropped: _c
    img = np.zeros(tuple(int(v) for v in d), dtype=np.uint8)
    rect = tuple(int(v) for v in r)
    _, rect_pad = face.extractFace(img, rect, padding=0.3)             # generate_dataview.py:61
    pos, _ = prn.process(img, image_info=rect)                         # :62 via face.detect3dLandmarks
    kpt = prn.get_landmarks(pos)
    tform, _ = __import__("oracle.torch_oracle", fromlist=["x"]).prn_crop_transform(rect)
    tforms.append(tform)
    kpts.append(kpt)
    faces.append(face.getFace(kpt, rect_pad))                          # :64
    padded.append(rect_pad)
    sub.append(pos[::16, ::16])
  cases.update({"dims": dims, "rects": rects, "seeds": seeds, "tform": np.stack(tforms), "kpt": np.stack(kpts),
                "face_lmk": np.stack(faces), "padded": np.array(padded), "pos_sub16": np.stack(sub)})
  np.savez_compressed(os.path.join(OUT, "prn_cases.npz"), **cases)


def main():
  sys.path.insert(0, REF)
  os.environ.setdefault("LIP_READING_WS_PATH", REF)
  install_stubs()
  torch.set_num_threads(1)
  from src.train.ctc_loss import ctc_loss as ref_ctc          # clean import (torch only)
  import src.models.lipreader.better_model as bm              # shimmed import
  import src.models.lipreader.decoder as ref_decoder          # clean apart from Levenshtein
  import src.data.data_loader as dl
  char2idx = dict(dl._markers2Id)
  for ch in dl._labels:                                       # build_vocab fallback (:100-115)
    char2idx[ch] = len(char2idx)
  assert len(char2idx) == V
  gen_ctc(ref_ctc)
  gen_greedy(ref_decoder, char2idx)
  gen_enc(bm, char2idx)
  gen_step(bm, ref_ctc, char2idx)
  gen_dec(bm, ref_ctc, char2idx)
  gen_enc(bm, char2idx, ENC2_CFGS, "enc2_cases.npz")
  gen_dec(bm, ref_ctc, char2idx, DEC2_CFGS, "dec2_cases.npz")
  install_face_stubs()
  import src.models.face.prnet as ref_prnet                   # shimmed import (estimate_transform)
  import src.utils.data.face as ref_face                      # clean apart from the inert stubs
  gen_lmk(ref_face)
  gen_prn(ref_face, ref_prnet)
  for f in sorted(os.listdir(OUT)):
    if f.endswith(".npz"):
      print("%-20s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
  main()
