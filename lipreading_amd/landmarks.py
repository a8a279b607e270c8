"""Landmark step on the device — the arithmetic of `src/utils/data/face.py` and
`src/models/face/prnet.py` that surrounds the two networks (dlib detector, PRNet) between a video
frame and the dataview's `face_lmk_seq` rows:

  `_applyPadding` face.py:76-90 · PRNet crop geometry prnet.py:112-119,136-140 · restore
  prnet.py:150-156 · `get_landmarks` prnet.py:162-170 · `getFace` face.py:164-175,
  composed as `_gen_data` does (generate_dataview.py:58-64: the UNPADDED rect defines the crop, the
  PADDED one the translation).

The reference does this per frame in NumPy inside the offline ETL; here whole clips go through one
launch so the step can sit in front of the encoder on the GPU.  dlib / PRNet inference itself is out
of scope (third-party nets, weights not shipped): `landmark_step` takes the network's position maps.
"""
import torch

from . import _C

_mouth = slice(48, 68)  # face.py:21 (defined, unused by the reference as well)


def apply_padding(dims, rects, padding):
  """Batched `_applyPadding` (face.py:76-90).

  dims (n,2|3) int = (img_h, img_w[, c]); rects (n,4) int = (left,right,top,bottom) ->
  padded rects (n,4) int32 on the device.  Integer arithmetic is the reference's:
  int(padding * box_w) truncates toward zero, then clamps to the image."""
  _C.require_cuda(rects, dims)
  assert dims.shape[-1] in (2, 3) and rects.shape[-1] == 4
  n = rects.shape[0]
  r = rects.to(torch.int32).contiguous()
  d = dims[:, :2].to(torch.int32).contiguous()
  out = torch.empty_like(r)
  _C.check(_C.lib().lr_lmk_apply_padding(r.data_ptr(), d.data_ptr(), out.data_ptr(), n,
                                         float(padding), _C.stream_handle()), "lr_lmk_apply_padding")
  return out


def get_face(lmks, rects):
  """Batched `getFace` (face.py:164-175): x -= left, y -= top, z untouched.

  lmks (n,P,3) float; rects (n,4) int (left,right,top,bottom) -> (n,P,3) float32."""
  _C.require_cuda(lmks, rects)
  assert lmks.dim() == 3 and lmks.shape[2] == 3
  assert rects.dim() == 2 and rects.shape[1] == 4 and rects.shape[0] == lmks.shape[0]
  x = lmks.to(torch.float32).contiguous()
  r = rects.to(torch.int32).contiguous()
  out = torch.empty_like(x)
  _C.check(_C.lib().lr_lmk_translate(x.data_ptr(), r.data_ptr(), out.data_ptr(), x.shape[0],
                                     x.shape[1], _C.stream_handle()), "lr_lmk_translate")
  return out


def crop_transform(rects, resolution=256):
  """PRNet crop geometry (prnet.py:112-119,136-140) for n face rects (left,right,top,bottom):
  -> (tform (n,3,3) float64 — what `estimate_transform('similarity', src_pts, DST_PTS).params` is in
  the reference — and the crop side `size` (n,) int32)."""
  _C.require_cuda(rects)
  assert rects.dim() == 2 and rects.shape[1] == 4
  r = rects.to(torch.int32).contiguous()
  n = r.shape[0]
  tform = torch.empty((n, 3, 3), dtype=torch.float64, device=r.device)
  sizes = torch.empty((n,), dtype=torch.int32, device=r.device)
  _C.check(_C.lib().lr_lmk_crop_transform(r.data_ptr(), tform.data_ptr(), sizes.data_ptr(), n, int(resolution),
                                          _C.stream_handle()), "lr_lmk_crop_transform")
  return tform, sizes


def restore(cropped_pos, tform):
  """prnet.py:150-156: position maps of the crops (n,res,res,3) float32 + tform (n,3,3) float64 ->
  position maps in image coordinates (n,res,res,3) float64."""
  _C.require_cuda(cropped_pos, tform)
  assert cropped_pos.dim() == 4 and cropped_pos.shape[3] == 3 and tform.shape == (cropped_pos.shape[0], 3, 3)
  cp = cropped_pos.to(torch.float32).contiguous()
  tf = tform.to(torch.float64).contiguous()
  pos = torch.empty(cp.shape, dtype=torch.float64, device=cp.device)
  n = cp.shape[0]
  _C.check(_C.lib().lr_lmk_restore(cp.data_ptr(), tf.data_ptr(), pos.data_ptr(), n, cp.shape[1] * cp.shape[2],
                                   _C.stream_handle()), "lr_lmk_restore")
  return pos


def get_landmarks(pos, uv_kpt_ind, rects=None):
  """prnet.py:162-170: kpt = pos[uv_kpt_ind[1], uv_kpt_ind[0], :] for n position maps (n,res,res,3)
  float64 -> (n,K,3) float64; with `rects` also getFace's translation (face.py:164-175)."""
  _C.require_cuda(pos, uv_kpt_ind, rects)
  assert pos.dim() == 4 and pos.shape[1] == pos.shape[2] and pos.shape[3] == 3 and uv_kpt_ind.shape[0] == 2
  p = pos.to(torch.float64).contiguous()
  uv = uv_kpt_ind.to(torch.int32).contiguous()
  r = None if rects is None else rects.to(torch.int32).contiguous()
  n, K = p.shape[0], uv.shape[1]
  out = torch.empty((n, K, 3), dtype=torch.float64, device=p.device)
  _C.check(_C.lib().lr_lmk_gather(p.data_ptr(), uv.data_ptr(), _C.ptr(r), out.data_ptr(), n, p.shape[1], K,
                                  _C.stream_handle()), "lr_lmk_gather")
  return out


def landmark_step(cropped_pos, rects, dims, uv_kpt_ind, padding=0.3, dtype=torch.float64):
  """`_gen_data` (generate_dataview.py:58-64) around the networks, one launch for n frames:
  cropped_pos (n,res,res,3) float32 = PRNet's output for the crop of each frame, rects (n,4) the
  detector's UNPADDED face rects, dims (n,2|3) = (img_h, img_w[, c]) -> (face landmarks (n,K,3)
  relative to the padded rect — a `face_lmk_seq` row per frame; float64 as the reference stores them,
  or float32 as the encoder consumes them — and the padded rects (n,4) int32)."""
  _C.require_cuda(cropped_pos, rects, dims, uv_kpt_ind)
  assert cropped_pos.dim() == 4 and cropped_pos.shape[1] == cropped_pos.shape[2] and cropped_pos.shape[3] == 3
  assert dtype in (torch.float64, torch.float32)
  cp = cropped_pos.to(torch.float32).contiguous()
  r = rects.to(torch.int32).contiguous()
  d = dims[:, :2].to(torch.int32).contiguous()
  uv = uv_kpt_ind.to(torch.int32).contiguous()
  n, K = cp.shape[0], uv.shape[1]
  out = torch.empty((n, K, 3), dtype=dtype, device=cp.device)
  padded = torch.empty((n, 4), dtype=torch.int32, device=cp.device)
  f64 = out.data_ptr() if dtype == torch.float64 else None
  f32 = out.data_ptr() if dtype == torch.float32 else None
  _C.check(_C.lib().lr_lmk_landmarks(cp.data_ptr(), r.data_ptr(), d.data_ptr(), float(padding), uv.data_ptr(), f64, f32,
                                     padded.data_ptr(), n, cp.shape[1], K, _C.stream_handle()), "lr_lmk_landmarks")
  return out, padded


def lip_crop(frames, lmks, size=96, margin=0.3, mouth=_mouth):
  """BUILD-DEFINED (the reference never crops the mouth, face.py:21 is unused):
  frames (n,3,H,W) uint8 + landmarks (n,68,3) in image pixels -> (n,3,size,size) uint8 mouth crops:
  bounding box of landmarks `mouth` -> square window of side max(w,h)*(1+2*margin) about its centre
  -> bilinear resize (half-pixel centres, edge clamping), rounded to nearest."""
  _C.require_cuda(frames, lmks)
  assert frames.dim() == 4 and frames.shape[1] == 3 and frames.dtype == torch.uint8
  assert lmks.dim() == 3 and lmks.shape[2] == 3 and lmks.shape[0] == frames.shape[0]
  f = frames.contiguous()
  l = lmks.to(torch.float32).contiguous()
  n, _, H, W = f.shape
  out = torch.empty((n, 3, size, size), dtype=torch.uint8, device=f.device)
  _C.check(_C.lib().lr_lip_crop_u8(f.data_ptr(), l.data_ptr(), out.data_ptr(), n, H, W, size, l.shape[1],
                                   mouth.start, mouth.stop, float(margin), _C.stream_handle()),
           "lr_lip_crop_u8")
  return out
