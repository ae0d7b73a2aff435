"""Mirror of monolith/native_training/touched_key_set_ops.py:20-68 over the C ABI: the set of FIDs
an optimizer step touched since the last drain (parameter sync to serving reads and clears it).
The set lives in HBM (csrc/keyset.hip); `steal` hands the keys back as a device tensor."""
import ctypes as C

import torch

from monolith_amd import _lib

# 64 MB
TOUCHED_KEY_SET_CAPACITY = 64 * 1024 * 1024 // (8 * 4)
TOUCHED_KEY_SET_CONCURRENCY_LEVEL = 1024


class TouchedKeySet(object):

  def __init__(self, capacity: int = TOUCHED_KEY_SET_CAPACITY,
               concurrency_level: int = TOUCHED_KEY_SET_CONCURRENCY_LEVEL, name_suffix: str = ""):
    # concurrency_level sizes the reference's lock stripes; the device set needs none
    self._capacity = int(capacity)
    self._concurrency_level = concurrency_level
    self._h = C.c_void_p()
    _lib.check(_lib.lib.mono_keyset_create(C.c_int64(self._capacity), C.byref(self._h)))

  def __del__(self):
    h = getattr(self, "_h", None)
    lib = getattr(_lib, "lib", None) if _lib is not None else None
    if h and lib is not None:
      lib.mono_keyset_destroy(h)
      self._h = None

  def insert(self, ids: torch.Tensor) -> int:
    """MonolithTouchedKeySetInsert: returns total_dropped_num."""
    ids = torch.as_tensor(ids)
    if not ids.is_cuda:
      ids = ids.to(torch.device("cuda", torch.cuda.current_device()))
    ids = ids.to(torch.int64).contiguous().reshape(-1)
    dropped = C.c_int64()
    _lib.check(_lib.lib.mono_keyset_insert(self._h, _lib.vp(ids), C.c_int64(ids.numel()),
                                           C.byref(dropped), _lib.stream_ptr(ids.device)))
    return int(dropped.value)

  def steal(self) -> torch.Tensor:
    """MonolithTouchedKeySetSteal: the keys (order unspecified); the set is empty afterwards."""
    n = self.size()
    dev = torch.device("cuda", torch.cuda.current_device())
    out = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    cnt = C.c_int64()
    _lib.check(_lib.lib.mono_keyset_steal(self._h, _lib.vp(out), C.c_int64(out.numel()),
                                          C.byref(cnt), _lib.stream_ptr(dev)))
    return out[:int(cnt.value)]

  def size(self) -> int:
    out = C.c_int64()
    _lib.check(_lib.lib.mono_keyset_size(self._h, C.byref(out)))
    return int(out.value)

  @property
  def capacity(self) -> int:
    return self._capacity

  @property
  def concurrency_level(self) -> int:
    return self._concurrency_level

  @property
  def handle(self):
    return self._h
