"""ctypes loader of libmonolith_hip.so. There is no CPU fallback: a missing or unloadable
library raises, and every non-zero status from the C ABI raises MonolithHipError."""
import ctypes as C
import os

# torch must load its HIP runtime before libmonolith_hip.so pulls in libamdhip64: with the order
# reversed both end up sharing the system copy and torch then reports "No HIP GPUs are available".
import torch  # noqa: F401  pylint: disable=unused-import

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmonolith_hip.so")


class MonolithHipError(RuntimeError):
  """Raised for any non-zero status of the C ABI (the reference raises
  tf.errors.InvalidArgumentError / ResourceExhaustedError from its Status)."""


class SegCfg(C.Structure):  # mono_segment_cfg
  _fields_ = [("dim", C.c_int32), ("opt_kind", C.c_int32), ("init_acc", C.c_float),
              ("weight_decay", C.c_float), ("l1", C.c_float), ("l2", C.c_float),
              ("beta", C.c_float), ("init_kind", C.c_int32), ("init_a", C.c_float),
              ("init_b", C.c_float), ("momentum", C.c_float), ("epsilon", C.c_float),
              ("averaging_ratio", C.c_float), ("learning_rate", C.c_float),
              ("use_nesterov", C.c_int32), ("beta1", C.c_float), ("beta2", C.c_float),
              ("comp_kind", C.c_int32), ("comp_a", C.c_float), ("comp_b", C.c_float),
              ("stochastic_rounding_float16", C.c_int32)]


class TableCfg(C.Structure):  # mono_table_cfg
  _fields_ = [("n_segments", C.c_int32), ("default_expire_days", C.c_int32),
              ("segments", C.POINTER(SegCfg)), ("initial_capacity", C.c_int64),
              ("n_slot_expire", C.c_int32), ("entry_type", C.c_int32),
              ("slot_ids", C.POINTER(C.c_int32)), ("slot_expire_days", C.POINTER(C.c_int32)),
              ("seed", C.c_uint64), ("enable_feature_eviction", C.c_int32),
              ("feature_evict_every_n_hours", C.c_int32)]


def _load():
  if not os.path.exists(LIB_PATH):
    raise ImportError(
        "monolith_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; "
        "g.build()'` or `make -C monolith_amd/csrc`; there is no CPU fallback." % LIB_PATH)
  lib = C.CDLL(LIB_PATH)
  lib.mono_last_error.restype = C.c_char_p
  lib.mono_fused_reorder_last_plan.restype = C.c_int64
  return lib


lib = _load()


def check(rc):
  if rc != 0:
    raise MonolithHipError((lib.mono_last_error() or b"unknown error").decode())


def vp(x):
  """torch tensor / int / None -> c_void_p."""
  if x is None:
    return C.c_void_p(0)
  if isinstance(x, int):
    return C.c_void_p(x)
  return C.c_void_p(x.data_ptr())


def stream_ptr(device=None):
  import torch
  return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
