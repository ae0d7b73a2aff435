"""Mirror of monolith/native_training/hash_table_ops.py:50-537 over the C ABI.

Same class / method names, argument order and "a write returns a new table object" chaining
as the reference; tensors are torch CUDA tensors (int64 ids, float32 values) used purely as
device buffers. TF graph / proto / saver-listener plumbing is not reproduced.
"""
import abc
import copy
import os
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from monolith_amd import _lib
from monolith_amd import entry
from monolith_amd import hash_filter_ops

_DEVICE = "cuda"


_WARNED_MULTI_THREAD_DEDUP = False


def _dev():
  return torch.device(_DEVICE, torch.cuda.current_device())


def _ids(x) -> torch.Tensor:
  """`ids = tf.convert_to_tensor(ids, tf.int64)` (hash_table_ops.py:254)."""
  if not isinstance(x, torch.Tensor):
    x = torch.as_tensor(x, dtype=torch.int64)
  elif x.is_cuda and x.dtype == torch.int64 and x.dim() == 1 and x.is_contiguous():
    return x  # (the hot callers: no torch call on the launch path)
  return x.to(device=_dev(), dtype=torch.int64).reshape(-1).contiguous()


def _f32(x) -> torch.Tensor:
  if not isinstance(x, torch.Tensor):
    x = torch.as_tensor(x, dtype=torch.float32)
  elif x.is_cuda and x.dtype == torch.float32 and x.is_contiguous():
    return x
  return x.to(device=_dev(), dtype=torch.float32).contiguous()


def _scalar(x, default=0) -> int:
  if x is None:
    return default
  if isinstance(x, torch.Tensor):
    return int(x.item())
  return int(x)


class BaseHashTable(abc.ABC):
  """hash_table_ops.py:50-128. For the write operation, it will return a new HashTable."""

  @abc.abstractmethod
  def assign(self, ids, values, req_time=None) -> "BaseHashTable":
    pass

  @abc.abstractmethod
  def assign_add(self, ids, values, req_time=None) -> "BaseHashTable":
    pass

  @abc.abstractmethod
  def lookup(self, ids, use_multi_threads=False, enable_dedup=False) -> torch.Tensor:
    pass

  @property
  @abc.abstractmethod
  def dim_size(self):
    pass

  @abc.abstractmethod
  def apply_gradients(self, ids, grads, global_step, use_multi_threads=False, enable_dedup=False,
                      req_time=None) -> "BaseHashTable":
    pass

  @abc.abstractmethod
  def as_op(self):
    pass


class _TableResource:
  """The ref-counted resource of ops/hash_table_op.cc:38-131: owns the mono_table handle and
  holds a reference on its filter (hash_table_op.cc:61-63)."""

  def __init__(self, table_config: entry.EmbeddingHashTableConfig,
               hash_filter: hash_filter_ops.HashFilter):
    segs = []
    _reject_unimplemented_optimizers(table_config)
    for s in table_config.segments:
      o, i = s.opt_config, s.init_config
      sr16 = isinstance(o, entry.StochasticRoundingFloat16OptimizerWrapper)
      if sr16:
        o = o._optimizer
      if not isinstance(o, (entry.SgdOptimizer, entry.AdagradOptimizer, entry.FtrlOptimizer,
                            entry.MomentumOptimizer, entry.RmspropOptimizer,
                            entry.RmspropV2Optimizer, entry.AdadeltaOptimizer,
                            entry.AdamOptimizer, entry.AmsgradOptimizer,
                            entry.MovingAverageOptimizer, entry.BatchSoftmaxOptimizer)):
        raise ValueError("optimizer %r is out of scope of the MI355X hot path" % (o,))
      init_acc = getattr(o, "initial_accumulator_value", None)
      seg = _lib.SegCfg(
          s.dim_size, o.kind, 0.1 if init_acc is None else init_acc,
          getattr(o, "weight_decay_factor", 0.0) or 0.0,
          getattr(o, "l1_regularization", None) or 0.0,
          getattr(o, "l2_regularization", None) or 0.0,
          getattr(o, "beta", None) or 0.0, i.kind, 0.0, 0.0)
      seg.momentum = getattr(o, "momentum", 0.9)
      seg.epsilon = getattr(o, "epsilon", 0.01)
      seg.averaging_ratio = getattr(o, "averaging_ratio", 0.9)
      conf_lr = getattr(o, "learning_rate", None)  # _convert_to_proto keeps plain numbers only
      seg.learning_rate = float(conf_lr) if isinstance(conf_lr, (int, float)) else 0.01
      seg.use_nesterov = int(bool(getattr(o, "use_nesterov", False)))
      seg.beta1 = getattr(o, "beta1", 0.9)
      seg.beta2 = getattr(o, "beta2", 0.99)
      seg.stochastic_rounding_float16 = int(sr16)
      comp = s.comp_config
      seg.comp_kind = getattr(comp, "kind", 0) or 0
      if isinstance(comp, entry.FixedR8Compressor):
        seg.comp_a = comp.fixed_range
      elif isinstance(comp, entry.OneBitCompressor):
        seg.comp_a, seg.comp_b = comp.amplitude, comp.init_scale
      if isinstance(i, entry.ConstantsInitializer):
        seg.init_a = i.constant
      elif isinstance(i, entry.RandomUniformInitializer):
        seg.init_a = -0.05 if i.minval is None else i.minval
        seg.init_b = 0.05 if i.maxval is None else i.maxval
      segs.append(seg)
    arr = (_lib.SegCfg * len(segs))(*segs)
    se = table_config.slot_expire_time_config.slot_expire_times
    n = len(se)
    sl = (C.c_int32 * max(n, 1))(*list(se.keys()))
    dd = (C.c_int32 * max(n, 1))(*list(se.values()))
    cfg = _lib.TableCfg(len(segs), table_config.slot_expire_time_config.default_expire_time, arr,
                        table_config.initial_capacity, n,
                        2 if table_config.entry_type == "serving" else 0, sl, dd, table_config.seed,
                        1 if table_config.enable_feature_eviction else 0,
                        int(table_config.feature_evict_every_n_hours))
    self.serving = table_config.entry_type == "serving"
    self.h = C.c_void_p()
    self.filter = hash_filter
    self.config = table_config
    self.touched_key_set = None  # SetHopscotchHashSet (hash_table_op.cc:90-95)
    _lib.check(_lib.lib.mono_table_create(C.byref(cfg), C.byref(self.h)))
    self.dim = _lib.lib.mono_table_dim(self.h)
    self.row_floats = _lib.lib.mono_table_row_floats(self.h)
    self.slice_size = _lib.lib.mono_table_slice_size(self.h)

  def __del__(self):
    h = getattr(self, "h", None)
    lib = getattr(_lib, "lib", None) if _lib is not None else None  # None at interpreter exit
    if h and lib is not None:
      lib.mono_table_destroy(h)
      self.h = None


_NAME_SET = set()  # HashTableMetadata.name_set (hash_table_ops.py:134-137, :218-223)


def reset_name_scope():
  """A fresh tf.Graph in the reference; clears the shared_name registry."""
  _NAME_SET.clear()


class HashTable(BaseHashTable):
  """It maps a int64 to a float32 embedding (hash_table_ops.py:149-361)."""

  def __init__(self, table: _TableResource = None, shared_name: str = None, dim_size: int = None,
               slot_expire_time_config=None, learning_rate_tensor: Sequence[float] = None,
               saver_parallel: int = -1, extra_restore_names=None):
    self._table = table
    self._dim_size = dim_size
    self._init_table_name = shared_name
    self._check_and_insert_name(shared_name)
    self._slot_expire_time_config = slot_expire_time_config
    self._learning_rate_tensor = learning_rate_tensor
    self._saver_parallel = saver_parallel
    self._extra_restore_names = extra_restore_names or []
    self.export_share_embedding = None

  @classmethod
  def _check_and_insert_name(cls, name):
    if name in _NAME_SET:
      raise ValueError("shared_name {} has already been used.".format(name))
    _NAME_SET.add(name)

  @property
  def table(self):
    return self._table

  @property
  def name(self):
    return self._init_table_name

  @property
  def extra_restore_names(self):
    return self._extra_restore_names

  @property
  def dim_size(self):
    return self._dim_size

  @property
  def learning_rate_tensor(self):
    lr = self._learning_rate_tensor
    return [float(x() if callable(x) else x) for x in lr]

  def _h(self):
    return self._table.h

  def _filter_h(self):
    f = self._table.filter
    return f.handle if f is not None else None

  # ---- BaseHashTable
  def assign(self, ids, values, req_time=None) -> "HashTable":
    ids, values = _ids(ids), _f32(values)
    self._check_values(ids, values)
    _lib.check(_lib.lib.mono_table_assign(self._h(), self._filter_h(), _lib.vp(ids),
                                          _lib.vp(values), C.c_int64(ids.numel()),
                                          C.c_int64(_scalar(req_time)), _lib.stream_ptr()))
    return self._copy_with_new_table(self._table)

  def assign_add(self, ids, values, req_time=None) -> "HashTable":
    ids, values = _ids(ids), _f32(values)
    self._check_values(ids, values)
    _lib.check(_lib.lib.mono_table_assign_add(self._h(), self._filter_h(), _lib.vp(ids),
                                              _lib.vp(values), C.c_int64(ids.numel()),
                                              C.c_int64(_scalar(req_time)), _lib.stream_ptr()))
    return self._copy_with_new_table(self._table)

  def lookup(self, ids, use_multi_threads=False, enable_dedup=False) -> torch.Tensor:
    ids = _ids(ids)
    out = torch.empty((ids.numel(), self._dim_size), dtype=torch.float32, device=ids.device)
    _lib.check(_lib.lib.mono_table_lookup(self._h(), _lib.vp(ids), C.c_int64(ids.numel()),
                                          _lib.vp(out), _lib.stream_ptr()))
    return out

  def lookup_hint(self, ids, ahead=1) -> None:
    """mono_table_lookup_hint (no counterpart in the reference's op set; include/monolith_hip.h):
    `ids` are the ids of the lookup() `ahead` calls from now (1 = the next one) — call it between a
    lookup and its update, when the coming batches are already known, and those lookups only have
    to gather rows. Never changes a result."""
    ids = _ids(ids)
    # (the library reads the ids later, on a stream of its own: keep the tensor — possibly a
    # converted copy — alive until the lookups it can still be queued for have passed)
    keep = self._table.__dict__.setdefault("_hint_keep", [])
    keep.append(ids)
    del keep[:-8]
    _lib.check(_lib.lib.mono_table_lookup_hint(self._h(), _lib.vp(ids), C.c_int64(ids.numel()),
                                               C.c_int(ahead), _lib.stream_ptr()))

  def lookup_entry(self, ids) -> List[bytes]:
    """hash_table_ops.py:278-281 -> HashTableLookupEntryOp: one serialized EntryDump per id
    (without the id field: Save() of the accessor does not set it), b"" for ids not in the table."""
    from monolith_amd import checkpoint
    ids = _ids(ids)
    n = int(ids.numel())
    t = self._table
    if t.serving:  # entry_accessor.cc:85-87
      raise RuntimeError("ServingEntryAccessor doesn't support Save")
    rows = torch.empty((n, t.row_floats), dtype=torch.float32, device=ids.device)
    ts = torch.empty(n, dtype=torch.int32, device=ids.device)
    found = torch.empty(n, dtype=torch.uint8, device=ids.device)
    _lib.check(_lib.lib.mono_table_lookup_entry(t.h, _lib.vp(ids), C.c_int64(n), _lib.vp(rows),
                                                _lib.vp(ts), _lib.vp(found), _lib.stream_ptr()))
    layout = checkpoint.DumpLayout.from_table_config(t.config)
    rows_h, ts_h, found_h = rows.cpu().numpy(), ts.cpu().numpy().view(np.uint32), found.cpu().numpy()
    out = [b""] * n
    for i in np.nonzero(found_h)[0]:
      out[i] = layout._encode_one(0, rows_h[i], int(ts_h[i]))[9:]   # drop `09 <id>`
    return out

  def apply_gradients(self, ids, grads, global_step, use_multi_threads=False, enable_dedup=False,
                      req_time=None) -> "HashTable":
    if use_multi_threads and enable_dedup:
      # ops/hash_table_update_op.cc:114-131 cuts the batch into as many ranges as TensorFlow's
      # Shard() finds worker threads for and dedups inside each range: which duplicates meet
      # depends on the host's thread pool. Here the batch is ONE range (the reference's own result
      # whenever Shard() decides on a single shard); said once, loudly, because a stateful
      # optimizer gives a different row when a key's duplicates straddle two of those ranges.
      global _WARNED_MULTI_THREAD_DEDUP
      if not _WARNED_MULTI_THREAD_DEDUP:
        import warnings
        warnings.warn("apply_gradients(use_multi_threads=True, enable_dedup=True): the reference "
                      "dedups per thread range (machine dependent); monolith_amd dedups over the "
                      "whole batch, i.e. the single-range result", RuntimeWarning, stacklevel=2)
        _WARNED_MULTI_THREAD_DEDUP = True
    ids, grads = _ids(ids), _f32(grads)
    self._check_values(ids, grads)
    lrs = self.learning_rate_tensor
    arr = (C.c_float * len(lrs))(*lrs)
    _lib.check(_lib.lib.mono_table_apply_gradients(
        self._h(), self._filter_h(), _lib.vp(ids), _lib.vp(grads), C.c_int64(ids.numel()), arr,
        C.c_int(len(lrs)), C.c_int64(_scalar(req_time)), C.c_int64(_scalar(global_step)),
        C.c_int(int(bool(enable_dedup))), _lib.stream_ptr()))
    self._record_touched(ids)
    return self._copy_with_new_table(self._table)

  def set_touched_key_set(self, key_set) -> "HashTable":
    """EmbeddingHashTableTfBridge::SetHopscotchHashSet (embedding_hash_table_tf_bridge.cc:
    398-402): from now on every id an optimizer step is applied to is inserted into `key_set`
    (a touched_key_set_ops.TouchedKeySet); None detaches it."""
    self._table.touched_key_set = key_set
    return self

  def _record_touched(self, ids):
    """embedding_hash_table_tf_bridge.cc:303-307: ids_after_filter go into the touched set. An id
    the occurrence filter rejected is not in the table afterwards (the gate only applies to
    unknown ids, :274-275), so "applied" == "contained after the step"."""
    ks = self._table.touched_key_set
    if ks is None or ids.numel() == 0:
      return
    f = self._table.filter
    if f is not None and f.kind == "sliding":
      ids = ids[self.contains(ids)]
    ks.insert(ids)

  def as_op(self):
    return self._table

  def table_update(self, update_op) -> "HashTable":
    return self._copy_with_new_table(self._table)

  def _copy_with_new_table(self, new_table):
    copied = copy.copy(self)
    copied.__dict__["_table"] = new_table
    return copied

  def _check_values(self, ids, values):
    # OP_REQUIRES / ValidateDim (embedding_hash_table_tf_bridge.cc:38-44)
    if values.numel() != ids.numel() * self._dim_size:
      raise _lib.MonolithHipError("The dim doesn't match expectation. %d vs %d" %
                                  (values.numel(), ids.numel() * self._dim_size))

  # ---- extras of the reference class
  def size(self) -> int:
    out = C.c_int64()
    _lib.check(_lib.lib.mono_table_size(self._h(), C.byref(out), _lib.stream_ptr()))
    return int(out.value)

  def contains(self, ids) -> torch.Tensor:
    ids = _ids(ids)
    out = torch.empty(ids.numel(), dtype=torch.uint8, device=ids.device)
    _lib.check(_lib.lib.mono_table_contains(self._h(), _lib.vp(ids), C.c_int64(ids.numel()),
                                            _lib.vp(out), _lib.stream_ptr()))
    return out.bool()

  def clear(self) -> "HashTable":
    _lib.check(_lib.lib.mono_table_clear(self._h(), _lib.stream_ptr()))
    return self._copy_with_new_table(self._table)

  def evict(self, max_update_time: int) -> "HashTable":
    """CuckooEmbeddingHashTable::Evict, driven by the caller instead of the bridge's
    background thread (embedding_hash_table_tf_bridge.cc:69-100)."""
    _lib.check(_lib.lib.mono_table_evict(self._h(), C.c_int64(int(max_update_time)),
                                         _lib.stream_ptr()))
    return self._copy_with_new_table(self._table)

  def max_update_ts(self) -> int:
    out = C.c_int64()
    _lib.check(_lib.lib.mono_table_max_update_ts(self._h(), C.byref(out)))
    return int(out.value)

  def reserve(self, rows: int) -> "HashTable":
    _lib.check(_lib.lib.mono_table_reserve(self._h(), C.c_int64(int(rows)), _lib.stream_ptr()))
    return self

  def save_as_tensor(self, apply_expiry=False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Dump (ids, rows[dim + optimizer state], last_update_ts). The reference emits serialized
    EntryDump protos (hash_table_ops.py:335-361); the payload is the same three fields."""
    n = self.size()
    dev = _dev()
    ids = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    rows = torch.empty((max(n, 1), self._table.row_floats), dtype=torch.float32, device=dev)
    ts = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = C.c_int64()
    _lib.check(_lib.lib.mono_table_export(self._h(), C.c_int64(n), C.c_int(int(apply_expiry)),
                                          _lib.vp(ids), _lib.vp(rows), _lib.vp(ts), C.byref(cnt),
                                          _lib.stream_ptr()))
    k = int(cnt.value)
    return ids[:k], rows[:k], ts[:k]

  def restore_from_tensor(self, ids, rows, ts, clear: bool = True) -> "HashTable":
    """Clear + upsert of full entries (ops/hash_table_restore_op.cc:82-131); clear=False is the
    multi-table restore, which upserts into the tables as they are."""
    if clear:
      self.clear()
    ids = _ids(ids)
    rows = _f32(rows)
    ts = ts.to(device=_dev(), dtype=torch.int32).contiguous()
    _lib.check(_lib.lib.mono_table_import(self._h(), _lib.vp(ids), _lib.vp(rows), _lib.vp(ts),
                                          C.c_int64(ids.numel()), _lib.stream_ptr()))
    return self._copy_with_new_table(self._table)


def _save_restore_methods():
  """save / restore in the reference's on-disk format (hash_table_ops.py:313-333 ->
  ops/hash_table_save_op.cc / hash_table_restore_op.cc), through monolith_amd.checkpoint."""
  from monolith_amd import checkpoint

  def save(self, basename: str, random_sleep_ms: int = 0) -> "HashTable":
    if self._table.serving:  # entry_accessor.cc:85-87
      raise RuntimeError("ServingEntryAccessor doesn't support Save")
    # save-time expiry: rows older than their slot's expire time are skipped (:142-150)
    ids, rows, ts = self.save_as_tensor(apply_expiry=True)
    layout = checkpoint.DumpLayout.from_table_config(self._table.config)
    nshards = checkpoint.pick_nshards(int(ids.numel()), self._saver_parallel)
    checkpoint.write_shards(str(basename), layout, ids.cpu().numpy(), rows.cpu().numpy(),
                            ts.cpu().numpy().astype("uint32"), max(1, nshards))
    return self._copy_with_new_table(self._table)

  def restore(self, basename: str) -> "HashTable":
    layout = checkpoint.DumpLayout.from_table_config(self._table.config)
    ids, rows, ts = checkpoint.read_shards(str(basename), layout)
    if self._table.serving:  # Restore = Encode(dump.num): the optimizer part of the dump is unused
      rows = np.ascontiguousarray(rows[:, :self._table.dim])
    return self.restore_from_tensor(torch.from_numpy(ids), torch.from_numpy(rows),
                                    torch.from_numpy(ts.astype("int64")))

  HashTable.save = save
  HashTable.restore = restore


_save_restore_methods()


def _all_table_tensor_prefix(table: HashTable) -> List[str]:
  """hash_table_ops.py:554-556."""
  all_names = [table.name] + list(table.extra_restore_names)
  return [name.replace(":", "-").replace("/", "-") for name in all_names]


def save_hash_tables(asset_dir: str, tables: Sequence[HashTable]) -> None:
  """What HashTableCheckpointSaverListener.before_save does for every table of the graph
  collection (hash_table_ops.py:573-590): each table is dumped to
  <asset_dir><table name with ':' and '/' replaced by '-'>-%05d-of-%05d. `asset_dir` is used as a
  string prefix, exactly like the reference (it ends with '/')."""
  d = os.path.dirname(asset_dir)
  if d:
    os.makedirs(d, exist_ok=True)
  for t in tables:
    t.save(asset_dir + _all_table_tensor_prefix(t)[0])


def restore_hash_tables(asset_dir: str, tables: Sequence[HashTable]) -> None:
  """HashTableCheckpointRestorerListener._restore_from_path_prefix (hash_table_ops.py:645-679):
  the first of [name] + extra_restore_names that has dump files under the prefix is restored;
  none -> ValueError (model structure changed)."""
  import glob
  for t in tables:
    prefixes = _all_table_tensor_prefix(t)
    for prefix in prefixes:
      if glob.glob(glob.escape(asset_dir + prefix) + "*"):
        t.restore(asset_dir + prefix)
        break
    else:
      raise ValueError("Unable to find table checkpoint in '%s' for table: %r. Maybe the model "
                       "structure has been changed." % (asset_dir, prefixes))


def _host_np(x, dtype) -> np.ndarray:
  """Host metadata (tensor / list / ndarray, possibly a list of 0-d tensors) as a flat contiguous
  numpy array."""
  if isinstance(x, torch.Tensor):
    x = x.detach().cpu().numpy()
  elif isinstance(x, (list, tuple)) and x and isinstance(x[0], torch.Tensor):
    x = [float(v) for v in x]
  return np.ascontiguousarray(x, dtype=dtype).reshape(-1)


def _handles(tables):
  res = [t.table if isinstance(t, HashTable) else t for t in tables]
  return res, (C.c_void_p * len(res))(*[r.h.value for r in res])


def fused_lookup(tables: List, ids, fused_slot_size, num_of_shards: int):
  """A fused operation for lookup (hash_table_ops.py:364-399).

  Returns (embeddings, recv_splits, id_offsets, emb_offsets, embedding_sizes): the flat
  embeddings on the device, the four int32 metadata vectors as CPU tensors."""
  res, hs = _handles(tables)
  K, N = len(res), int(num_of_shards)
  ids = _ids(ids)
  # The caller is usually waiting on a size it just read back, so the time to the first launch
  # is GPU idle time: host metadata stays in numpy (a torch CPU op costs several microseconds),
  # and the output length = sum over (shard, table) cells of ids * dim is computed here so that
  # the kernels go out with ONE call.
  if isinstance(fused_slot_size, torch.Tensor):
    fused_slot_size = fused_slot_size.cpu().numpy()
  fss = np.ascontiguousarray(fused_slot_size, dtype=np.int32).reshape(-1)
  if fss.size != K * N:
    raise _lib.MonolithHipError("len(fused_slot_size) / num_of_shards != len(table_handles)")
  cells = fss.tolist()
  if sum(cells) != ids.numel():
    raise _lib.MonolithHipError("sum(fused_slot_size) != len(ids)")
  meta_buf = np.zeros(N + 3 * K * N, np.int32)
  splits, id_off = meta_buf[:N], meta_buf[N:N + K * N]
  emb_off, emb_sizes = meta_buf[N + K * N:N + 2 * K * N], meta_buf[N + 2 * K * N:]
  dims = [r.dim for r in res]
  total = sum(c * dims[i % K] for i, c in enumerate(cells))
  emb = torch.empty(total, dtype=torch.float32, device=ids.device)
  _lib.check(_lib.lib.mono_tables_fused_lookup(
      hs, C.c_int(K), _lib.vp(ids), C.c_void_p(fss.ctypes.data), C.c_int(N), _lib.vp(emb),
      C.c_void_p(splits.ctypes.data), C.c_void_p(id_off.ctypes.data),
      C.c_void_p(emb_off.ctypes.data), C.c_void_p(emb_sizes.ctypes.data), _lib.stream_ptr()))
  return (emb, torch.from_numpy(splits), torch.from_numpy(id_off), torch.from_numpy(emb_off),
          torch.from_numpy(emb_sizes))


_fused_hint_keep: List = []


def fused_lookup_hint(tables: List, ids, fused_slot_size, num_of_shards: int, ahead: int = 1) -> None:
  """mono_tables_fused_lookup_hint (no counterpart in the reference's op set): announces the
  fused_lookup(tables, ids, fused_slot_size, num_of_shards) that is `ahead` calls away, so that it
  only has to gather rows. Never changes a result; takes effect for one table."""
  res, hs = _handles(tables)
  K, N = len(res), int(num_of_shards)
  ids = _ids(ids)
  if isinstance(fused_slot_size, torch.Tensor):
    fused_slot_size = fused_slot_size.cpu().numpy()
  fss = np.ascontiguousarray(fused_slot_size, dtype=np.int32).reshape(-1)
  if fss.size != K * N:
    raise _lib.MonolithHipError("len(fused_slot_size) / num_of_shards != len(table_handles)")
  # (the library reads the ids later, on a stream of its own)
  _fused_hint_keep.append(ids)
  del _fused_hint_keep[:-8]
  _lib.check(_lib.lib.mono_tables_fused_lookup_hint(
      hs, C.c_int(K), _lib.vp(ids), C.c_void_p(fss.ctypes.data), C.c_int(N), C.c_int(ahead),
      _lib.stream_ptr()))


def fused_apply_gradient(tables: List, ids, fused_slot_size, id_grads, id_offsets, grad_offsets,
                         learning_rate_tensors, learning_rate_lengths, req_time, global_step,
                         num_of_shards: int, enable_grad_accumulation: bool = False):
  """A fused operation for applying gradients (hash_table_ops.py:402-455)."""
  res, hs = _handles(tables)
  K, N = len(res), int(num_of_shards)
  ids = _ids(ids)
  grads = _f32(id_grads).reshape(-1)
  # host metadata in numpy: a torch CPU op costs microseconds each on the launch path
  fss = _host_np(fused_slot_size, np.int32)
  io = _host_np(id_offsets, np.int32)
  go = _host_np(grad_offsets, np.int32)
  lrs = _host_np(learning_rate_tensors, np.float32)
  lrl = _host_np(learning_rate_lengths, np.int32)
  if fss.size != K * N:
    raise _lib.MonolithHipError("len(fused_slot_size) / num_of_shards != len(table_handles)")
  fs = (C.c_void_p * K)(*[(r.filter.handle.value if r.filter is not None else None) for r in res])
  _lib.check(_lib.lib.mono_tables_fused_apply_gradient(
      hs, fs, C.c_int(K), _lib.vp(ids), C.c_void_p(fss.ctypes.data), _lib.vp(grads),
      C.c_void_p(io.ctypes.data), C.c_void_p(go.ctypes.data), C.c_void_p(lrs.ctypes.data),
      C.c_void_p(lrl.ctypes.data), C.c_int64(_scalar(req_time)), C.c_int64(_scalar(global_step)),
      C.c_int(N), C.c_int(int(bool(enable_grad_accumulation))), _lib.stream_ptr()))
  for k, t in enumerate(tables):
    if isinstance(t, HashTable) and t.table.touched_key_set is not None:
      for n in range(N):
        lo, cnt = int(io[n * K + k]), int(fss[n * K + k])
        if cnt:
          t._record_touched(ids[lo:lo + cnt])
  return tables


def extract_slot_from_entry(entry: Sequence[bytes], fid_v2: bool = True) -> np.ndarray:
  """hash_table_ops.py:692-693 -> ExtractSlotFromEntryOp (ops/hash_table/misc_ops.cc:113-142):
  slot of the id inside each serialized EntryDump (slot_id_v2 = bits 48..62, v1 = fid >> 54,
  reader_util.h:33-37); an unparsable / id-less dump reads as id 0."""
  from monolith_amd import checkpoint
  out = np.zeros(len(entry), np.int32)
  for i, b in enumerate(entry):
    b, fid, pos = bytes(b), 0, 0
    try:
      while pos < len(b):  # EntryDump.id = field 1, sfixed64 (embedding_hash_table.proto:45-50)
        tag, pos = checkpoint._read_varint(b, pos)
        wt = tag & 7
        if tag == 0x09:
          fid = int.from_bytes(b[pos:pos + 8], "little")
        if wt == 1:
          pos += 8
        elif wt == 5:
          pos += 4
        elif wt == 0:
          _, pos = checkpoint._read_varint(b, pos)
        elif wt == 2:
          ln, pos = checkpoint._read_varint(b, pos)
          pos += ln
        else:
          break
    except ValueError:  # "Fail to parse entry dump." is only logged there; the id read so far counts
      pass
    out[i] = ((fid >> 48) & 0x7FFF) if fid_v2 else (fid >> 54)
  return out


def _reject_unimplemented_optimizers(table_config):
  """optimizer_factory.cc:50-53,76-78: the reference factory throws for these configs."""
  for s in table_config.segments:
    o = getattr(s.opt_config, "_optimizer", s.opt_config)
    if isinstance(o, entry._NotImplementedOptimizer):
      raise ValueError("optimizer is not implemented yet. %s %r" % (type(o).__name__, o.__dict__))


def hash_table_from_config(config: entry.HashTableConfigInstance,
                           hash_filter: Optional[hash_filter_ops.HashFilter] = None,
                           name_suffix="", sync_client=None, saver_parallel: int = -1) -> HashTable:
  """hash_table_ops.py:458-492."""
  table_config = config.table_config
  dim_size = table_config.dim_size
  hash_table_name = "MonolithHashTable_" + name_suffix
  _reject_unimplemented_optimizers(table_config)
  if hash_filter is None:
    hash_filter = hash_filter_ops.create_dummy_hash_filter(name_suffix=name_suffix)
  if len(config.learning_rate_fns) != len(table_config.segments):
    raise ValueError("Size of learning_rate_fns and size of segments must be equal.")
  table = HashTable(_TableResource(table_config, hash_filter), shared_name=hash_table_name,
                    dim_size=dim_size,
                    slot_expire_time_config=table_config.slot_expire_time_config,
                    learning_rate_tensor=list(config.learning_rate_fns),
                    saver_parallel=saver_parallel, extra_restore_names=config.extra_restore_names)
  # ops/hash_table_op.cc:88-95: a table created with a sync client reports its updated ids to the
  # client's touched-key set. The client itself (gRPC push to serving) is out of scope; any object
  # with a `touched_key_set` attribute, or a TouchedKeySet, is accepted here.
  if sync_client is not None:
    table.set_touched_key_set(getattr(sync_client, "touched_key_set", sync_client))
  return table


_TEST_COUNTER = [0]


def test_hash_table(dim_size, enable_hash_filter=False, name_suffix=None, learning_rate=1.0,
                    occurrence_threshold=0, use_adagrad=False, expire_time=36500,
                    sync_client=None, extra_restore_names=None,
                    filter_capacity=hash_filter_ops.HASH_FILTER_CAPACITY) -> HashTable:
  """Returns a hash table which essentially is a |dim_size| float table with sgd optimizer
  (hash_table_ops.py:495-537)."""
  opt = entry.AdagradOptimizer() if use_adagrad else entry.SgdOptimizer()
  table_config = entry.EmbeddingHashTableConfig(
      segments=[entry.CombineAsSegment(dim_size, entry.ZerosInitializer(), opt,
                                       entry.Fp32Compressor())])
  table_config.slot_expire_time_config.default_expire_time = expire_time
  thr = entry.SlotOccurrenceThresholdConfig(default_occurrence_threshold=occurrence_threshold)
  config = entry.HashTableConfigInstance(table_config, [learning_rate],
                                         extra_restore_names=extra_restore_names)
  hash_filters = hash_filter_ops.create_hash_filters(0, enable_hash_filter, thr,
                                                     filter_capacity=filter_capacity)
  if not name_suffix:
    _TEST_COUNTER[0] += 1
    name_suffix = "test_%d" % _TEST_COUNTER[0]
  return hash_table_from_config(config=config, hash_filter=hash_filters[0],
                                name_suffix=name_suffix, sync_client=sync_client)


def vocab_hash_table(vocab_size: int, dim_size: int, enable_hash_filter=False,
                     learning_rate=1.0) -> HashTable:
  """hash_table_ops.py:540-552."""
  return test_hash_table(dim_size, enable_hash_filter=enable_hash_filter,
                         learning_rate=learning_rate)
