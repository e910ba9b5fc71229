"""ctypes binding of libtwingan_hip.so (include/twingan_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a kernel call returns
an error, this module raises.  (``oracle/`` is test infrastructure and is never imported here.)
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint, c_void_p

TG_F32, TG_BF16, TG_F16 = 0, 1, 2
TG_ALGO_DIRECT, TG_ALGO_MFMA = 0, 1
TG_EPI_BIAS, TG_EPI_LRELU = 1, 2
NF_LRELU, NF_PIXNORM, NF_NOSTATS = 1, 2, 4

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtwingan_hip.so')
if os.environ.get('TG_LIB_PATH'):      # kernel A/B runs (tools/ab.sh): an alternative build of the same ABI
  LIB_PATH = os.environ['TG_LIB_PATH']


class TgError(RuntimeError):
  pass


class TgConvDesc(Structure):
  _fields_ = [(k, c_int32) for k in ('n', 'hin', 'win', 'cin', 'hout', 'wout', 'cout', 'kh', 'kw', 'pad_t', 'pad_l',
                                     'dtype', 'algo', 'epilogue')] + [('lrelu_alpha', c_float), ('groups', c_int32)]


_P = c_void_p
_FP = c_void_p      # float* passed as raw address
_D = POINTER(TgConvDesc)

# name -> (restype, argtypes).  Mirrors include/twingan_hip.h one to one.
SIGNATURES = {
    'tg_version': (c_int, []),
    'tg_last_error': (c_char_p, []),
    'tg_last_kernel': (c_char_p, []),
    'tg_set_deterministic': (c_int, [c_int]),
    'tg_get_deterministic': (c_int, []),
    'tg_wgrad_defer': (c_int, [c_int]),
    'tg_wgrad_defer_flush': (c_int, [_P]),
    'tg_channel_sum_ordered': (c_int, [_P, _FP, c_int64, c_int, c_int, _FP, c_size_t, c_int, _P]),
    'tg_sum_ordered': (c_int, [_P, _P, _FP, c_int64, c_float, c_int, _FP, c_size_t, c_int, _P]),
    'tg_pointwise_conv_bwd_weight_ordered': (c_int, [_P, _P, _FP, c_int64, c_int, c_int, c_int, _FP, c_size_t, c_int, _P]),
    'tg_conv2d_fwd': (c_int, [_D, _P, _P, _FP, _P, _P]),
    'tg_conv2d_bwd_data': (c_int, [_D, _P, _P, _P, _P]),
    'tg_conv2d_bwd_data_masked': (c_int, [_D, _P, _P, _P, _P, _P]),
    'tg_conv2d_fwd_masked': (c_int, [_D, _P, _P, _P, _P, _P]),
    'tg_conv2d_bwd_data_unpool_supported': (c_int, [_D]),
    'tg_conv2d_bwd_data_unpool': (c_int, [_D, _P, _P, _P, _P, _P, _P, _P]),
    'tg_conv2d_bwd_data_unpool_act': (c_int, [_D, _P, _P, _P, _P, _P, _P, _P]),
    'tg_conv2d_bwd_weight_workspace': (c_size_t, [_D]),
    'tg_conv2d_bwd_weight': (c_int, [_D, _P, _P, _FP, c_int, _P, c_size_t, _P]),
    'tg_conv2d_bwd_weight2_workspace': (c_size_t, [_D, c_int]),
    'tg_conv2d_bwd_weight2': (c_int, [_D, c_int, _P, _P, _P, _P, _FP, c_int, _P, c_size_t, _P]),
    'tg_conv2d_bwd_weight_bias': (c_int, [_D, _P, _P, _FP, _FP, c_int, _P, c_size_t, _P]),
    'tg_conv2d_bwd_weight2_bias': (c_int, [_D, c_int, _P, _P, _P, _P, _FP, _FP, c_int, c_int, _P, c_size_t, _P]),
    'tg_conv2d_upcat_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'tg_conv2d_upcat_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_uint, c_int, _P]),
    'tg_conv2d_upcat_bwd_data': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_uint, c_int, _P]),
    'tg_transpose16': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'tg_flash_attention_supported': (c_int, [c_int, c_int, c_int]),
    'tg_flash_attention_workspace_bytes': (ctypes.c_int64, [c_int, c_int, c_int, c_int, c_int]),
    'tg_flash_attention_fwd': (c_int, [_P, _P, _P, _P, _FP, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'tg_flash_attention_bwd': (c_int, [_P] * 5 + [_FP] + [_P] * 4 + [c_int, c_int, c_int, c_int, c_int, _P]),
    'tg_flash_attention_bwd_bwd': (c_int, [_P] * 5 + [_FP] + [_P] * 8 + [c_int, c_int, c_int, c_int, c_int, _P]),
    'tg_comm_unique_id_bytes': (c_int, []),
    'tg_comm_unique_id': (c_int, [_P]),
    'tg_comm_init': (c_int, [_P, c_int, c_int, POINTER(c_void_p)]),
    'tg_allreduce': (c_int, [_P, _P, c_int64, c_int, _P]),
    'tg_comm_destroy': (c_int, [_P]),
    'tg_preprocess_images': (c_int, [_P, _P, _P, _FP, _P, c_int, c_int, c_int, _P]),
    'tg_preprocess_images_crop': (c_int, [_P, _P, _P, _P, _FP, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'tg_conv2d_fwd_pool_supported': (c_int, [_D]),
    'tg_conv2d_fwd_pool': (c_int, [_D, _P, _P, _FP, _P, _P, _P]),
    'tg_conv2d_fwd_pool_signs': (c_int, [_D, _P, _P, _FP, _P, _P, _P]),
    'tg_conv2d_fwd_stats_chunks': (c_int, [_D]),
    'tg_conv2d_fwd_stats': (c_int, [_D, _P, _P, _P, _FP, c_int, _P]),
    'tg_conv2d_upcat_fwd_stats_chunks': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'tg_conv2d_upcat_fwd_stats': (c_int, [_P, _P, _P, _P, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_uint,
                                          c_int, _P]),
    'tg_conv2d_upcat_bwd_weight_workspace': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'tg_conv2d_upcat_bwd_weight': (c_int, [_P, _P, _P, _FP, c_int, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_uint, c_int, _P]),
    'tg_conv2d_pack_elems': (c_size_t, [_D, c_int]),
    'tg_conv2d_pack_layout': (c_int, [_D, c_int]),
    'tg_conv2d_pack_weights': (c_int, [_D, _FP, c_int, _P, _P]),
    'tg_pack_table_bytes': (c_size_t, [c_int]),
    'tg_pack_table_fill': (c_int, [_D, _FP, c_int, _P, c_int, _P, POINTER(c_int32)]),
    'tg_conv2d_pack_weights_multi': (c_int, [_P, c_int, c_int, _P]),
    'tg_pointwise_conv_fwd': (c_int, [_P, _FP, _FP, _P, c_int64, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    'tg_pointwise_conv_fwd_masked': (c_int, [_P, _FP, _P, _P, c_int64, c_int, c_int, c_int, c_float, c_int, _P]),
    'tg_pointwise_conv_bwd_weight': (c_int, [_P, _P, _FP, c_int64, c_int, c_int, c_int, c_int, _P]),
    'tg_pointwise_conv_bwd_weight_bias': (c_int, [_P, _P, _FP, _FP, c_int64, c_int, c_int, c_int, c_int, _P]),
    'tg_instance_norm_stats': (c_int, [_P, _FP, _FP, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    'tg_norm_chunks': (c_int, [c_int, c_int, c_int]),
    'tg_instance_norm_partials': (c_int, [_P, _FP, c_int, c_int, c_int, c_int, c_int, _P]),
    'tg_norm_act_fwd_partials': (c_int, [_P, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_int, _P, _P, _FP, c_int, c_int, c_int, c_int,
                                         c_int, c_float, c_float, c_float, c_int, _P]),
    'tg_norm_act_fwd_conv_stats': (c_int, [_P, _FP, c_int, _FP, _FP, _FP, _FP, _FP, _FP, c_int, _P, _P, _FP, c_int, c_int, c_int,
                                           c_int, c_int, c_float, c_float, c_float, c_int, _P]),
    'tg_norm_act_fwd': (c_int, [_P, _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, _P, _FP, c_int, c_int, c_int, c_int, c_int,
                                c_float, c_float, c_int, _P]),
    'tg_norm_act_bwd': (c_int, [_P, _P, _P, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_int, _P, c_int, _FP, _FP, _FP, _FP, _FP,
                                c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, _P]),
    'tg_bias_lrelu_fwd': (c_int, [_P, _FP, _P, c_int64, c_int, c_float, c_int, _P]),
    'tg_lrelu_bwd': (c_int, [_P, _P, _P, c_int64, c_float, c_int, _P]),
    'tg_lrelu_bwd_bias': (c_int, [_P, _P, _P, _FP, c_int64, c_int, c_float, c_int, c_int, _P]),
    'tg_lrelu_pool_bwd': (c_int, [_P, _P, _P, _P, _FP, c_int, c_int, c_int, c_int, c_float, c_int, c_int, _P]),
    'tg_lrelu_pool_bwd_signs': (c_int, [_P, _P, _P, _FP, c_int, c_int, c_int, c_int, c_float, c_int, c_int, _P]),
    'tg_channel_sum': (c_int, [_P, _FP, c_int64, c_int, c_int, c_int, _P]),
    'tg_upsample2x_concat_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint, c_int, _P]),
    'tg_upsample2x_concat_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint, c_int, _P]),
    'tg_pool2x2_fwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    'tg_pool2x2_bwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    'tg_axpby': (c_int, [_P, _P, _P, c_int64, c_float, c_float, c_int, _P]),
    'tg_sample_lerp': (c_int, [_P, _P, _FP, _P, c_int, c_int64, c_int, _P]),
    'tg_sample_scale': (c_int, [_P, _FP, _FP, _P, c_int, c_int64, c_int, _P]),
    'tg_gdrop': (c_int, [_P, _FP, _FP, c_float, c_int, _P, c_int, c_int64, c_int, c_int, _P]),
    'tg_fill_scaled': (c_int, [_P, _FP, c_float, c_int64, c_int, _P]),
    'tg_cast': (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    'tg_mbstd_fwd': (c_int, [_P, _P, _FP, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    'tg_mbstd_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    'tg_mbstd_bwd_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    'tg_small_gemm': (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'tg_sum': (c_int, [_P, _FP, c_int64, c_float, c_int, c_int, _P]),
    'tg_abs_diff_sum': (c_int, [_P, _P, _FP, c_int64, c_float, c_int, c_int, _P]),
    'tg_abs_diff_bwd': (c_int, [_P, _P, _FP, _P, _P, c_int64, c_float, c_int, _P]),
    'tg_fc_fwd': (c_int, [_P, _FP, _FP, _FP, c_int, c_int, c_int, c_int, _P]),
    'tg_fc_bwd': (c_int, [_P, _FP, _FP, _P, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'tg_pred_losses_fwd': (c_int, [_FP, c_int, c_int, _P, c_int, _FP, c_int, _P]),
    'tg_pred_losses_bwd': (c_int, [_FP, c_int, c_int, _P, c_int, _P, c_int, _FP, _P]),
    'tg_sum_scalars': (c_int, [_P, c_int, _FP, _P]),
    'tg_rows_assemble': (c_int, [_P, c_int, _P, c_int, _P]),
    'tg_uniform': (c_int, [_FP, c_int64, ctypes.c_uint64, _P, c_float, c_float, _P]),
    'tg_pred_loss_fwd': (c_int, [_FP, _FP, c_int, c_int, c_float, c_float, c_float, c_int, _P]),
    'tg_pred_loss_bwd': (c_int, [_FP, _FP, _FP, c_int, c_int, c_float, c_float, c_float, _P]),
    'tg_var_from_sums': (c_int, [_FP, _FP, _FP, c_int, c_int64, _P]),
    'tg_sample_sumsq': (c_int, [_P, _FP, c_int, c_int64, c_int, _P]),
    'tg_gp_penalty': (c_int, [_FP, _FP, _FP, c_int, c_float, _P]),
    'tg_adam_step': (c_int, [_FP, _FP, _FP, _FP, _P, c_int64, c_float, _FP, c_float, c_float, c_float, c_float, _P]),
    'tg_adam_tick': (c_int, [_P, _FP, c_float, c_float, c_float, _P]),
    'tg_batched_gemm': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64,
                                c_int64, c_float, c_int, c_int, c_int, _P]),
    'tg_softmax_rows_fwd': (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    'tg_softmax_rows_bwd': (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P]),
    'tg_softmax_rows_bwd_bwd': (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    'tg_tanh_fwd': (c_int, [_P, _P, c_int64, c_int, _P]),
    'tg_tanh_bwd': (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    'tg_mul3': (c_int, [_P, _P, _P, _P, c_float, c_int64, c_int, _P]),
    'tg_scale_dev': (c_int, [_P, _FP, _P, c_int64, c_int, _P]),
    'tg_dot': (c_int, [_P, _P, _FP, _FP, c_int64, c_int, _P]),
    'tg_cosine_distance_fwd': (c_int, [_FP, _FP, _FP, c_int, c_int, c_float, _P]),
    'tg_cosine_distance_bwd': (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_float, _P]),
    'tg_spectral_norm_workspace': (c_size_t, [c_int, c_int]),
    'tg_spectral_norm_fwd': (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, _P, c_size_t, _P]),
    'tg_spectral_norm_bwd': (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, _P, c_size_t, _P]),
    'tg_sn_table_bytes': (c_size_t, [c_int]),
    'tg_sn_table_fill': (c_int, [c_int, _FP, _FP, _FP, _FP, _FP, _FP, _P, c_size_t, c_int, c_int, _P, POINTER(c_int32)]),
    'tg_spectral_norm_fwd_multi': (c_int, [_P, c_int, c_int, c_int, c_int, _P]),
    'tg_sn_assign_u': (c_int, [_P, c_int, _P]),
}

_lib = None


def load():
  """Loads (once) and returns the ctypes handle.  Raises TgError if the library was not built."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise TgError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                  '(make -C twingan_amd/csrc). There is no CPU fallback.' % LIB_PATH)
  lib = ctypes.CDLL(LIB_PATH)
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)       # AttributeError if a declared symbol is missing
    fn.restype = res
    fn.argtypes = args
  _lib = lib
  return lib


# Optional per-launch profiler used by bench.py's roofline pass: when set to a list, every call is
# bracketed by HIP events on torch's current stream (the stream the kernels are enqueued on) and
# (name, tag, flops, bytes, start_event, end_event) is appended.
profiler = None


def call(name, *args, work=None):
  """Calls an int-returning entry point and raises TgError(tg_last_error()) on failure.
  ``work`` = (tag, algorithmic flops, algorithmic bytes) of this launch, for the roofline pass."""
  lib = load()
  if profiler is not None:
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = getattr(lib, name)(*args)
    e1.record()
    if callable(work):
      work = work()
    tag, fl, by = work if work is not None else ('', 0, 0)
    kname = lib.tg_last_kernel().decode() if name.startswith('tg_conv2d') else ''
    profiler.append((name, tag, fl, by, e0, e1, kname))
  else:
    rc = getattr(lib, name)(*args)
  if rc != 0:
    raise TgError('%s failed (%d): %s' % (name, rc, lib.tg_last_error().decode()))
  return rc
