"""Mirror of monolith/native_training/hash_filter_ops.py:29-170 over the C ABI.

A filter "tensor" of the reference (a TF resource handle) is a HashFilter object here."""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
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
    self.split_num = int(split_num)  # as given: SlidingHashFilter::split_num_ is the unclamped value
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


def save_hash_filter(hash_filter: HashFilter, hash_filter_basename: str,
                     enable_hash_filter: bool = False) -> None:
  """hash_filter_ops.py:140-147 -> MonolithHashFilterSave (ops/hash_filter_save_op.cc:41-105):
  one file per block, <basename>-%05d-of-<split_num>; a dummy filter has no splits and writes
  nothing."""
  if enable_hash_filter is not True or hash_filter.kind != "sliding":
    return
  from monolith_amd import checkpoint
  st = hash_filter.state()
  nb, n_cells = st["n_blocks"], st["block_total_size"] + 64
  sliding = {"split_num": hash_filter.split_num, "max_forward_step": 2,
             "max_backward_step": nb - 2, "max_step": 16, "head": st["head"],
             "head_increment": st["head_increment"], "failure_count": st["failure_count"]}
  cells = np.empty(n_cells, np.uint16)
  for b in range(nb):
    ne = C.c_int64()
    _lib.check(_lib.lib.mono_filter_export_split(
        hash_filter.handle, C.c_int(b), C.c_void_p(cells.ctypes.data), C.c_int64(n_cells),
        C.byref(ne), _lib.stream_ptr()))
    meta = {"failure_count": 0, "total_size": st["block_total_size"], "num_elements": ne.value,
            "fill_rate": 1.2, "sliding": sliding}
    checkpoint.write_filter_split(checkpoint.sharded_name(str(hash_filter_basename), b, nb), meta,
                                  cells)


def restore_hash_filter(hash_filter: HashFilter, hash_filter_base_name: str,
                        enable_hash_filter: bool = False) -> None:
  """hash_filter_ops.py:150-157 -> MonolithHashFilterRestore (ops/hash_filter_restore_op.cc:36-110,
  sliding_hash_filter.cc:187-213): split_num / max_forward_step / max_backward_step / max_step of
  the dump must match the filter, head / head_increment / failure_count are taken from it."""
  if enable_hash_filter is not True:
    return
  from monolith_amd import checkpoint
  files = checkpoint.filter_files(str(hash_filter_base_name))
  if hash_filter.kind != "sliding":
    raise _lib.MonolithHipError("restore_hash_filter: dump files for a filter without splits")
  st = hash_filter.state()
  nb, n_cells = st["n_blocks"], st["block_total_size"] + 64
  if len(files) != nb:
    raise ValueError("split_num: %d does't match with : %d read from hash filter checkpoint file."
                     % (nb, len(files)))
  expect = {"split_num": hash_filter.split_num, "max_forward_step": 2, "max_backward_step": nb - 2,
            "max_step": 16}
  last = None
  for b, f in enumerate(files):
    meta, cells = checkpoint.read_filter_split(f)
    for k, v in expect.items():
      if meta["sliding"][k] != v:
        raise ValueError("%s: %d does't match with : %d read from hash filter checkpoint file."
                         % (k, v, meta["sliding"][k]))
    if len(cells) != n_cells:
      raise ValueError("total_size: %d does't match with : %d read from hash filter checkpoint "
                       "file." % (st["block_total_size"], meta["total_size"]))
    cells = np.ascontiguousarray(cells)
    _lib.check(_lib.lib.mono_filter_import_split(
        hash_filter.handle, C.c_int(b), C.c_void_p(cells.ctypes.data), C.c_int64(n_cells),
        C.c_int64(meta["num_elements"]), _lib.stream_ptr()))
    last = meta["sliding"]
  _lib.check(_lib.lib.mono_filter_set_state(
      hash_filter.handle, C.c_int64(last["head"]), C.c_int64(last["head_increment"]),
      C.c_int64(last["failure_count"]), _lib.stream_ptr()))


def save_hash_filters(asset_dir: str, hash_filters: Sequence[HashFilter],
                      enable_hash_filter: bool = False) -> None:
  """HashFilterCheckpointSaverListener.before_save (hash_filter_ops.py:203-226):
  <asset_dir>hash_filter_<ps index>."""
  for ps_idx, f in enumerate(hash_filters):
    save_hash_filter(f, asset_dir + "hash_filter_{}".format(ps_idx), enable_hash_filter)


def restore_hash_filters(asset_dir: str, hash_filters: Sequence[HashFilter],
                         enable_hash_filter: bool = False) -> None:
  """HashFilterCheckpointRestorerListener._restore_from_path_prefix (hash_filter_ops.py:287-303)."""
  for ps_idx, f in enumerate(hash_filters):
    restore_hash_filter(f, asset_dir + "hash_filter_{}".format(ps_idx), enable_hash_filter)


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
