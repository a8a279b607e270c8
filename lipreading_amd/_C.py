"""ctypes binding of include/lipreading_hip.h.

The product path has no CPU fallback: if the shared object is missing or a kernel reports an
error, the call raises.  Nothing in this module imports from oracle/.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

from . import _build

_lib = None

P = c_void_p  # every device pointer crosses the boundary as void*

# lr_status (include/lipreading_hip.h)
LR_OK, LR_ERR_INVALID_ARG, LR_ERR_WORKSPACE, LR_ERR_LAUNCH, LR_ERR_UNSUPPORTED, LR_ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5

# name -> (restype, argtypes); mirrors include/lipreading_hip.h declaration by declaration.
SIGNATURES = {
    "lr_version": (c_int, []),
    "lr_status_string": (c_char_p, [c_int]),
    "lr_device_count": (c_int, []),
    "lr_collate_pad_f32": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "lr_lmk_apply_padding": (c_int, [P, P, P, c_int, c_float, P]),
    "lr_lmk_translate": (c_int, [P, P, P, c_int, c_int, P]),
    "lr_lmk_crop_transform": (c_int, [P, P, P, c_int, c_int, P]),
    "lr_lmk_restore": (c_int, [P, P, P, c_int, c_int64, P]),
    "lr_lmk_gather": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "lr_lmk_landmarks": (c_int, [P, P, P, c_double, P, P, P, P, c_int, c_int, c_int, P]),
    "lr_lip_crop_u8": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    "lr_sgemm_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "lr_sgemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, P, c_int, P, c_int, c_float,
                          P, c_int, P, c_int, c_int, P, c_size_t, P]),
    "lr_xgemm_workspace_bytes": (c_size_t, [c_int] * 5),
    "lr_xgemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, P, c_int, P, c_int, c_float,
                          P, c_int, P, c_int, c_int, P, c_size_t, P]),
    "lr_sgemm_batched": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, P, c_int, c_int64, c_int64, P, c_int,
                                  c_int64, c_int64, c_float, P, c_int, c_int64, c_int64, c_int, c_int, P]),
    "lr_layernorm_forward": (c_int, [P, P, P, P, P, P, c_int, c_int, c_float, P]),
    "lr_layernorm_workspace_bytes": (c_size_t, [c_int]),
    "lr_layernorm_backward": (c_int, [P, P, P, P, P, P, P, P, P, c_size_t, c_int, c_int, c_int, P]),
    "lr_attn_softmax_forward": (c_int, [P, P, c_float, c_int, c_int, c_int, P]),
    "lr_attn_softmax_backward": (c_int, [P, P, c_float, c_int, c_int, c_int, P]),
    "lr_attn_fused_supported": (c_int, [c_int, c_int]),
    "lr_attn_fused_forward": (c_int, [P, P, P, c_float, c_int, c_int, c_int, c_int, P]),
    "lr_attn_fused_backward": (c_int, [P, P, P, P, c_float, c_int, c_int, c_int, c_int, P]),
    "lr_fgemm": (c_int, [c_int, c_int, c_int, c_int, P, c_int, P]),
    "lr_fgemm_splits": (c_int, [c_int, c_int, c_int]),
    "lr_fgemm_slab_floats": (ctypes.c_longlong, [c_int, c_int, c_int]),
    "lr_tfm_reserve_bytes": (c_size_t, [c_int] * 8),
    "lr_tfm_workspace_bytes": (c_size_t, [c_int] * 8),
    "lr_tfm_rowblock_supported": (c_int, [c_int] * 5),
    "lr_tfm_forward": (c_int, [c_int, P, P, P, P, P, P, c_size_t, P, c_size_t] + [c_int] * 7 + [c_float, P]),
    "lr_tfm_backward_data": (c_int, [c_int, P, P, P, P, P, c_size_t, P, c_size_t] + [c_int] * 7 + [P]),
    "lr_tfm_backward_weights": (c_int, [c_int, P, P, c_int, P, c_size_t, P, c_size_t] + [c_int] * 7 + [P]),
    "lr_rnn_pair_supported": (c_int, [c_int] * 6),
    "lr_rnn_one_launch_status": (c_int, [c_int] * 6),
    "lr_rnn_pass_launches": (c_int, [c_int] * 6),
    "lr_rnn_pair_errors": (c_int, []),
    "lr_fault_words_ptr": (c_void_p, []),
    "lr_fault_export": (c_int, [P, P, P]),
    "lr_fault_import": (c_int, [P, P, P]),
    "lr_fault_export_f32": (c_int, [P, P, P]),
    "lr_step_begin": (c_int, [P, c_int64, P, P]),
    "lr_step_begin_ctc": (c_int, [P, c_int64, P, P, c_int64, P, P, P, P, P, c_int, c_int, P]),
    "lr_rnn_debug_drop_member": (None, [c_int]),
    "lr_rnn_debug_disable_cluster": (None, [c_int]),
    "lr_rnn_debug_tune": (None, [c_int, c_int, c_int]),
    "lr_rnn_one_launch_enable": (None, [c_int]),
    "lr_rnn_one_launch_enabled": (c_int, []),
    "lr_debug_busy": (c_int, [c_int, c_int, c_int, P]),
    "lr_rnn_reserve_bytes": (c_size_t, [c_int] * 6),
    "lr_rnn_workspace_bytes": (c_size_t, [c_int] * 6),
    "lr_rnn_layer_forward": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, c_size_t, c_int, c_int,
                                      c_int, c_int, c_int, P]),
    "lr_rnn_layer_backward": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P,
                                       c_size_t, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int,
                                       P]),
    "lr_rnn_layer_backward_parts": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P,
                                             c_size_t, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int,
                                             c_int, P]),
    "lr_profile_enable": (c_int, [c_int]),
    "lr_profile_read": (c_int, [c_int, P, P]),
    "lr_proj_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "lr_proj_logsoftmax_forward": (c_int, [P, P, P, P, P, P, c_size_t, c_int, c_int, c_int, P]),
    "lr_proj_logsoftmax_backward": (c_int, [P, P, P, P, P, P, P, P, P, c_size_t, c_int, c_int, c_int,
                                             c_int, P]),
    "lr_decoder_reserve_bytes": (c_size_t, [c_int] * 10),
    "lr_decoder_workspace_bytes": (c_size_t, [c_int] * 10),
    "lr_decoder_forward": (c_int, [c_int, c_int, P, P, P, P, P, P, P, P, P, c_uint64, P, P, P, P, P, c_size_t] +
                           [c_int] * 7 + [P]),
    "lr_decoder_backward": (c_int, [c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_size_t, P,
                                     c_size_t, c_int] + [c_int] * 7 + [P]),
    "lr_decoder_backward_parts": (c_int, [c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_size_t, P,
                                           c_size_t, c_int] + [c_int] * 7 + [c_int, P]),
    "lr_decoder_backward_splittable": (c_int, [c_int, c_int]),
    "lr_ctc_prepare_i64": (c_int, [P, c_int64, P, P, P, P, P, c_int, c_int, P]),
    "lr_nll_mean_forward": (c_int, [P, P, c_int64, c_int, c_int, P, c_int, c_int, P]),
    "lr_nll_forward3": (c_int, [P, P, c_int64, c_int, c_int, P, c_int, c_int, P]),
    "lr_nll_mean_backward": (c_int, [P, c_int64, c_int, c_int, P, P, P, c_int, c_int, P]),
    "lr_ctc_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "lr_ctc_nll": (c_int, [P, c_int64, c_int64, P, c_int, P, P, P, P, c_size_t, c_int, c_int,
                            c_int, c_int, P]),
    "lr_ctc_grad": (c_int, [P, c_int64, c_int64, P, c_int, P, P, P, P, P, P, c_size_t, c_int,
                             c_int, c_int, c_int, P]),
    "lr_ctc_grad_scaled": (c_int, [P, c_int64, c_int64, P, c_int, P, P, P, P, P, P, P, c_size_t, c_int,
                                    c_int, c_int, c_int, P]),
    "lr_ctc_nll_reduce": (c_int, [P, c_int64, c_int64, P, c_int, P, P, P, P, c_size_t, c_int, P, P, P, c_int, c_int,
                                  c_int, c_int, P]),
    "lr_ctc_reduce": (c_int, [P, P, P, c_int, P, P, P, c_int, P]),
    "lr_ctc_greedy_decode": (c_int, [P, c_int64, c_int64, P, P, P, P, P, c_int, c_int, c_int,
                                      c_int, P]),
    "lr_clip_to_ndhwc_bf16": (c_int, [P, c_int, P, c_int64, c_int, c_int, P]),
    "lr_conv3d_pack_weights": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "lr_conv3d_pack_weights_multi": (c_int, [c_int] + [P] * 9 + [P]),
    "lr_conv3d_forward": (c_int, [P, P, P, P] + [c_int] * 14 + [P]),
    "lr_conv3d_patch_supported": (c_int, [c_int] * 11),
    "lr_conv3d_pool_fusion_supported": (c_int, [c_int] * 11),
    "lr_conv3d_forward_pooled": (c_int, [P, P, P, P, P] + [c_int] * 14 + [P]),
    "lr_conv3d_dgrad_pooled_supported": (c_int, [c_int] * 10),
    "lr_conv3d_dgrad_pooled": (c_int, [P, P, P, P] + [c_int] * 12 + [P]),
    "lr_unpool_code_bf16": (c_int, [P, P, P, P, P, c_int, P, c_size_t, c_int64, c_int, c_int, c_int, P]),
    "lr_conv3d_wgrad_workspace_bytes": (c_size_t, [c_int] * 5),
    "lr_conv3d_wgrad": (c_int, [P, P, P, P, P, c_size_t] + [c_int] * 15 + [P]),
    "lr_conv3d_wgrad_pooled_supported": (c_int, [c_int] * 12),
    "lr_conv3d_wgrad_pooled_supported_frames": (c_int, [c_int] * 13),
    "lr_conv3d_wgrad_pooled": (c_int, [P, P, P, P, P, P, P, c_size_t] + [c_int] * 16 + [P]),
    "lr_maxpool_hw2_bf16": (c_int, [P, P, c_int64, c_int, c_int, c_int, P]),
    "lr_unpool_workspace_bytes": (c_size_t, [c_int]),
    "lr_unpool_relu_mask_bf16": (c_int, [P, P, P, P, c_int, P, c_size_t, c_int64, c_int, c_int, c_int, P]),
    "lr_bf16_to_f32": (c_int, [P, P, c_int64, P]),
    "lr_f32_to_bf16": (c_int, [P, P, c_int64, P]),
    "lr_dropout_forward": (c_int, [P, P, P, c_int64, c_float, ctypes.c_uint64, P]),
    "lr_mul_f32": (c_int, [P, P, P, c_int64, P]),
    "lr_cat_directions": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "lr_sumsq": (c_int, [P, c_int64, P, P]),
    "lr_adam_step": (c_int, [P, P, P, P, c_int64, P, c_float, c_float, c_float, c_float, c_float,
                              c_float, P, P, P, P, c_float, P]),
    "lr_clip_adam_step": (c_int, [P, P, P, P, c_int64, P, c_float, c_float, c_float, c_float, c_float,
                                   c_float, P, P, P, P, c_float, c_int64, P]),
}


class DecoderParams(ctypes.Structure):
  """lr_decoder_params (include/lipreading_hip.h)."""
  _fields_ = [(n, c_void_p) for n in ("emb", "w_ih", "w_hh", "b_ih", "b_hh", "attn_w1", "attn_b1", "attn_w2",
                                      "attn_b2", "w_c", "b_c", "w_o", "b_o", "out_mask")]


class DecoderGrads(ctypes.Structure):
  """lr_decoder_grads."""
  _fields_ = [(n, c_void_p) for n in ("emb", "w_ih", "w_hh", "b_ih", "b_hh", "attn_w1", "attn_b1", "attn_w2",
                                      "attn_b2", "w_c", "b_c", "w_o", "b_o")] + [("emb_padding_idx", c_int)]


class FgemmJob(ctypes.Structure):
  """lr_fgemm_job (include/lipreading_hip.h)."""
  _fields_ = [(n, c_void_p) for n in ("A", "B", "C", "bias", "addend", "mask", "colsum", "slabs")] + \
             [(n, ctypes.c_int32) for n in ("M", "N", "K", "lda", "ldb", "ldc", "ldadd", "add_period", "ldmask", "flags",
                                            "splits")] + [("alpha", c_float), ("beta", c_float)] + \
             [("b_shift", ctypes.c_int32), ("b_period", ctypes.c_int32)]


DEC_MAX_LAYERS = 8   # LR_DEC_MAX_LAYERS


class DecoderUpper(ctypes.Structure):
  """lr_decoder_upper: layers 1.. of the decoder's RNN stack."""
  _fields_ = [("num_layers", c_int)] + [(n, c_void_p * (DEC_MAX_LAYERS - 1)) for n in ("w_ih", "w_hh", "b_ih", "b_hh")] + \
             [("drop_mask", c_void_p)]


class DecoderUpperGrads(ctypes.Structure):
  """lr_decoder_upper_grads."""
  _fields_ = [(n, c_void_p * (DEC_MAX_LAYERS - 1)) for n in ("w_ih", "w_hh", "b_ih", "b_hh")]


class LipReadingHipError(RuntimeError):
  pass


def lib_path():
  # LIPREADING_HIP_LIB: another build of the same C ABI (A/B timing of kernel variants: tools/build_variant.sh)
  return os.environ.get("LIPREADING_HIP_LIB") or _build.LIB_PATH


def lib():
  """Load (once) and return the C-ABI library.  Raises if it has not been built."""
  global _lib
  if _lib is None:
    path = lib_path()
    if not os.path.exists(path):
      raise LipReadingHipError(
          "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
          "(there is no CPU fallback for the MI355X hot path)" % path)
    # torch bundles its own libamdhip64.so.7; load it FIRST so that this library binds to the
    # same HIP runtime instance that owns torch's streams and allocations.  (Loaded the other
    # way round, /opt/rocm's runtime and torch's coexist and every launch on a torch stream
    # fails with an invalid-handle error.)
    import torch  # noqa: F401
    handle = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
      fn = getattr(handle, name)  # AttributeError if the header and the .so disagree
      fn.restype = restype
      fn.argtypes = argtypes
    _lib = handle
    if torch.cuda.is_available():
      handle.lr_fault_words_ptr()   # allocate the device-side fault words now, outside any stream capture
  return _lib


def check(status, what=""):
  if status != 0:
    msg = lib().lr_status_string(int(status)).decode()
    raise LipReadingHipError("%s failed: %s (%d)" % (what or "lipreading_hip call", msg, status))


def stream_handle():
  """hipStream_t of torch's current stream as an integer (void*)."""
  import torch
  return torch.cuda.current_stream().cuda_stream


def ptr(t):
  """Device pointer of a tensor (None -> NULL)."""
  return None if t is None else t.data_ptr()


def require_cuda(*tensors):
  for t in tensors:
    if t is not None and not t.is_cuda:
      raise LipReadingHipError(
          "lipreading_amd ops run on the MI355X only; got a %s tensor (no CPU fallback)" % t.device)
