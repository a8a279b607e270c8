"""Landmark step on the device — the arithmetic of `src/utils/data/face.py` that touches the
68 3-D landmarks between PRNet and the dataview (`_applyPadding` :76-90, `getFace` :164-175).

The reference does this per frame in NumPy inside the offline ETL (generate_dataview.py:58-76);
here whole clips are translated in one launch so the step can sit in front of the encoder on
the GPU.  dlib / PRNet inference itself is out of scope (third-party nets, weights not shipped).
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
