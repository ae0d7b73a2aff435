"""Mirror of monolith/native_training/hash_filter_ops.py:29-170 over the C ABI.

A filter "tensor" of the reference (a TF resource handle) is a HashFilter object here."""
import ctypes as C
from typing import List, Optional

import torch

from monolith_amd import _lib
from monolith_amd.entry import SlotOccurrenceThresholdConfig

HASH_FILTER_CAPACITY = 300000000  # hash_filter_ops.py:29
HASH_FILTER_SPLIT_NUM = 7  # hash_filter_ops.py:30


class FilterType(object):  # hash_filter_ops.py:48-51
  SLIDING_HASH_FILTER = 'sliding_hash_filter'
  PROBABILISTIC_FILTER = 'probabilistic_filter'


class HashFilter:
  """Owner of a mono_filter handle (ops/hash_filter_op.cc:109-141)."""

  def __init__(self, kind: str, capacity: int = HASH_FILTER_CAPACITY,
               split_num: int = HASH_FILTER_SPLIT_NUM,
               config: Optional[SlotOccurrenceThresholdConfig] = None, name: str = ""):
    config = config or SlotOccurrenceThresholdConfig()
    st = config.slot_occurrence_thresholds
    n = len(st)
    slots = (C.c_int32 * max(n, 1))(*list(st.keys()))
    thr = (C.c_int32 * max(n, 1))(*list(st.values()))
    self.kind = kind
    self.name = name
    self._h = C.c_void_p()
    _lib.check(_lib.lib.mono_filter_create(
        C.c_int(1 if kind == "sliding" else 0), C.c_int64(capacity), C.c_int(split_num),
        C.c_int(config.default_occurrence_threshold), C.c_int(n), slots, thr, C.byref(self._h)))

  def __del__(self):
    h = getattr(self, "_h", None)
    lib = getattr(_lib, "lib", None) if _lib is not None else None  # None at interpreter exit
    if h and lib is not None:
      lib.mono_filter_destroy(h)
      self._h = None

  @property
  def handle(self):
    return self._h

  def add(self, ids: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """Batch Filter::add (sliding_hash_filter.cc:57-89): old count of every occurrence."""
    ids = ids.contiguous()
    counts = counts.to(torch.int32).contiguous()
    out = torch.empty(ids.numel(), dtype=torch.int32, device=ids.device)
    _lib.check(_lib.lib.mono_filter_add(self._h, _lib.vp(ids), _lib.vp(counts),
                                        C.c_int64(ids.numel()), _lib.vp(out),
                                        _lib.stream_ptr(ids.device)))
    return out

  def get(self, ids: torch.Tensor) -> torch.Tensor:
    ids = ids.contiguous()
    out = torch.empty(ids.numel(), dtype=torch.int32, device=ids.device)
    _lib.check(_lib.lib.mono_filter_get(self._h, _lib.vp(ids), C.c_int64(ids.numel()),
                                        _lib.vp(out), _lib.stream_ptr(ids.device)))
    return out

  def state(self):
    buf = (C.c_int64 * (5 + 1024))()
    _lib.check(_lib.lib.mono_filter_state(self._h, buf, C.c_int64(len(buf)),
                                          _lib.stream_ptr()))
    nb = int(buf[3])
    return {"head": int(buf[0]), "head_increment": int(buf[1]), "failure_count": int(buf[2]),
            "n_blocks": nb, "block_total_size": int(buf[4]),
            "num_elements": [int(buf[5 + i]) for i in range(nb)]}


def create_hash_filter(capacity: int, split_num: int,
                       config: Optional[SlotOccurrenceThresholdConfig] = None,
                       name_suffix: str = "") -> HashFilter:
  """hash_filter_ops.py:54-63."""
  return HashFilter("sliding", capacity, split_num, config, "MonolithHashFilter" + name_suffix)


def create_probabilistic_filter(*args, **kwargs):
  """hash_filter_ops.py:66-74. Out of scope (random admission is unpinnable)."""
  raise NotImplementedError("probabilistic_filter is out of scope of the MI355X hot path")


def create_dummy_hash_filter(name_suffix: str = "0") -> HashFilter:
  """hash_filter_ops.py:77-80."""
  return HashFilter("dummy", name="DummyHashFilter" + name_suffix)


def _create_hash_filter(enable_hash_filter: bool, config=None, name_suffix: str = "",
                        filter_capacity: int = HASH_FILTER_CAPACITY,
                        filter_split_num: int = HASH_FILTER_SPLIT_NUM,
                        filter_equal_probability: bool = False,
                        filter_type=FilterType.SLIDING_HASH_FILTER) -> HashFilter:
  """hash_filter_ops.py:83-103."""
  if enable_hash_filter is True:
    if filter_type == FilterType.SLIDING_HASH_FILTER:
      return create_hash_filter(filter_capacity, filter_split_num, config, name_suffix)
    elif filter_type == FilterType.PROBABILISTIC_FILTER:
      return create_probabilistic_filter(filter_equal_probability, config, name_suffix)
    raise ValueError("Invalid filter type, please investigate and retry!")
  return create_dummy_hash_filter(name_suffix)


def create_hash_filters(ps_num: int, enable_hash_filter: bool, config=None,
                        filter_capacity: int = HASH_FILTER_CAPACITY,
                        filter_split_num: int = HASH_FILTER_SPLIT_NUM,
                        filter_equal_probability: bool = False,
                        filter_type=FilterType.SLIDING_HASH_FILTER) -> List[HashFilter]:
  """hash_filter_ops.py:106-142 (one filter per PS; ps_num == 0 -> one local filter)."""
  if ps_num == 0:
    return [_create_hash_filter(enable_hash_filter, config, "", filter_capacity,
                                filter_split_num, filter_equal_probability, filter_type)]
  return [_create_hash_filter(enable_hash_filter, config, "_" + str(i), filter_capacity,
                              filter_split_num, filter_equal_probability, filter_type)
          for i in range(ps_num)]


class _InterceptGradient(torch.autograd.Function):
  """MonolithHashFilterInterceptGradient: identity forward, filtered backward
  (ops/hash_filter_intercept_gradient_op.cc:25-62)."""

  @staticmethod
  def forward(ctx, filt, ids, embeddings):
    ctx.filt = filt
    ctx.save_for_backward(ids)
    return embeddings.view_as(embeddings)

  @staticmethod
  def backward(ctx, grad):
    (ids,) = ctx.saved_tensors
    return None, None, intercept_gradient_gradient(ctx.filt, ids, grad)


def intercept_gradient_gradient(filter_tensor: HashFilter, ids: torch.Tensor,
                                grad: torch.Tensor) -> torch.Tensor:
  """MonolithHashFilterInterceptGradientGradient as a plain call."""
  ids = ids.contiguous()
  grad = grad.contiguous().float()
  n = ids.numel()
  out = torch.empty_like(grad)
  width = grad.numel() // n if n else 0
  _lib.check(_lib.lib.mono_filter_intercept_gradient(filter_tensor.handle, _lib.vp(ids),
                                                     _lib.vp(grad), C.c_int64(n),
                                                     C.c_int64(width), _lib.vp(out),
                                                     _lib.stream_ptr(ids.device)))
  return out


def intercept_gradient(filter_tensor: HashFilter, ids: torch.Tensor, embeddings: torch.Tensor):
  """If id is supposed to be filtered, the gradient will be intercepted. Outputs the same
  embeddings (hash_filter_ops.py:161-170)."""
  return _InterceptGradient.apply(filter_tensor, ids, embeddings)
