"""Mirror of monolith/native_training/multi_hash_table_ops.py:128-495: several hash tables behind
one object with the "raw" ragged API (ids of all tables in one flat tensor + row splits, values /
gradients in one flat float tensor).

The reference keeps the tables in one C++ resource (MonolithMultiHashTable*); here the object owns
one HBM table per name and the raw calls are the fused multi-table entry points of the C ABI with
a single shard (mono_tables_fused_lookup / mono_tables_fused_apply_gradient): one launch sequence
for all tables, no per-table Python loop on the hot calls.

Not reproduced: to_proto / from_proto (TF graph plumbing) and save / restore — the multi-table dump
(ops/multi_hash_table_save_restore_ops.cc:96-160) is a Snappy-compressed record file for which the
reference ships no fixture; single tables use hash_table_ops.HashTable.save / restore."""
from typing import Dict, List, Optional, Tuple

import torch

from monolith_amd import entry, hash_filter_ops, hash_table_ops
from monolith_amd.embedding_combiners import Ragged
from monolith_amd.multi_type_hash_table import BaseMultiTypeHashTable


def concat_1d_tensors(*args) -> Ragged:  # multi_hash_table_ops.py:102-106
  dev = args[0].device if args else torch.device("cpu")
  values = torch.cat([a.reshape(-1) for a in args]) if args else torch.zeros(0, dtype=torch.int64)
  splits = [0]
  for a in args:
    splits.append(splits[-1] + a.numel())
  return Ragged(values, torch.tensor(splits, dtype=torch.int64, device=dev))


def get_list_from_flat_value(key: Ragged, dims: Tuple[int], flat_value: torch.Tensor):
  """multi_hash_table_ops.py:110-117."""
  rs = key.row_splits.cpu().tolist()
  out, pos = [], 0
  for i, d in enumerate(dims):
    n = rs[i + 1] - rs[i]
    out.append(flat_value[pos:pos + n * d].reshape(-1, d))
    pos += n * d
  return out


def flatten_n_tensors(*args) -> torch.Tensor:  # multi_hash_table_ops.py:121-125
  return torch.cat([a.reshape(-1) for a in args])


class MultiHashTable(BaseMultiTypeHashTable):
  NAME_PREFIX = "MonolithMultiHashTable"

  def __init__(self, tables: List[hash_table_ops.HashTable], table_names: Tuple[str],
               dims: Tuple[int], shared_name: str):
    self._tables = tables
    self._table_names = tuple(table_names)
    self._dims = tuple(dims)
    self._shared_name = shared_name

  @classmethod
  def from_configs(cls, configs: Dict[str, entry.HashTableConfigInstance], hash_filter=None,
                   sync_client=None, name_suffix: str = "", saver_parallel: int = -1):
    """multi_hash_table_ops.py:70-96, :235-276: tables in sorted-name order, one shared filter."""
    table_names = tuple(sorted(configs.keys()))
    if hash_filter is None:
      hash_filter = hash_filter_ops.create_dummy_hash_filter(name_suffix=name_suffix)
    tables = []
    for name in table_names:
      config = configs[name]
      if len(config.learning_rate_fns) != len(config.table_config.segments):
        raise ValueError("Size of learning_rate_fns and size of segments must be equal.")
      tables.append(hash_table_ops.hash_table_from_config(
          config, hash_filter=hash_filter, name_suffix="%s_multi_%s" % (name_suffix, name),
          sync_client=sync_client, saver_parallel=saver_parallel))
    dims = tuple(t.dim_size for t in tables)
    return cls(tables, table_names, dims, "_".join([cls.NAME_PREFIX, name_suffix]))

  @property
  def table_names(self):
    return self._table_names

  @property
  def shared_name(self):
    return self._shared_name

  def _copy(self):
    return MultiHashTable(self._tables, self._table_names, self._dims, self._shared_name)

  def _device(self):
    return torch.device("cuda", torch.cuda.current_device())

  # ---- BaseMultiTypeHashTable
  def assign(self, slot_to_id_and_value, req_time=None, enable_inter_table_parallelism=False):
    ragged_id = self.get_ragged_id({k: v[0] for k, v in slot_to_id_and_value.items()})
    flat_value = self.get_flat_value({k: v[1] for k, v in slot_to_id_and_value.items()})
    return self.raw_assign(ragged_id, flat_value, req_time)

  def assign_add(self, slot_to_id_and_value, req_time=None):
    ragged_id = self.get_ragged_id({k: v[0] for k, v in slot_to_id_and_value.items()})
    flat_value = self.get_flat_value({k: v[1] for k, v in slot_to_id_and_value.items()})
    return self._raw_update("assign_add", ragged_id, flat_value, req_time)

  def lookup(self, slot_to_id: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    ragged_id = self.get_ragged_id(slot_to_id)
    emb = self.get_embeddings(ragged_id, self.raw_lookup(ragged_id))
    return {k: v for k, v in emb.items() if k in slot_to_id}

  def apply_gradients(self, slot_to_id_and_grad, global_step, req_time=None):
    ragged_id = self.get_ragged_id({k: v[0] for k, v in slot_to_id_and_grad.items()})
    flat_grad = self.get_flat_value({k: v[1] for k, v in slot_to_id_and_grad.items()})
    return self.raw_apply_gradients(ragged_id, flat_grad, global_step, req_time)

  def lookup_entry(self, slot_to_id, enable_inter_table_parallelism: bool = False):
    raise NotImplementedError("")  # multi_hash_table_ops.py:362-366: the reference raises too

  # multi_hash_table_ops.py:410-429: declared and left unimplemented by the reference
  def fused_lookup(self, ids, fused_slot_size, num_of_shards: int):
    raise NotImplementedError("")

  def fused_apply_gradient(self, ids, fused_slot_size, id_grads, id_offsets, grad_offsets,
                           global_step, req_time, num_of_shards: int,
                           enable_grad_accumulation: bool = False):
    raise NotImplementedError("")

  def as_op(self, *args, **kwargs):
    return self._tables

  def get_table_dim_sizes(self):
    return self._dims

  # ---- RawMultiTypeHashTable (multi_hash_table_ops.py:438-495)
  def _row_lengths(self, ragged_id: Ragged) -> List[int]:
    rs = ragged_id.row_splits.cpu().tolist()
    if len(rs) != len(self._tables) + 1:
      raise ValueError("ragged id has %d rows, the table set has %d tables" %
                       (len(rs) - 1, len(self._tables)))
    return [rs[i + 1] - rs[i] for i in range(len(self._tables))]

  def raw_lookup(self, ragged_id: Ragged) -> torch.Tensor:
    """MonolithMultiHashTableLookup: flat float32 embeddings, table after table."""
    lens = self._row_lengths(ragged_id)
    emb, _, _, _, _ = hash_table_ops.fused_lookup(self._tables, ragged_id.values, lens, 1)
    return emb

  def raw_apply_gradients(self, ragged_id: Ragged, flat_grad: torch.Tensor, global_step,
                          req_time=None):
    """MonolithMultiHashTableOptimize: one optimizer step per id occurrence, in order."""
    lens = self._row_lengths(ragged_id)
    id_off, grad_off, io, go = [], [], 0, 0
    for n, d in zip(lens, self._dims):
      id_off.append(io)
      grad_off.append(go)
      io += n
      go += n * d
    lrs, lr_lens = [], []
    for t in self._tables:
      lr = t.learning_rate_tensor
      lrs += lr
      lr_lens.append(len(lr))
    hash_table_ops.fused_apply_gradient(self._tables, ragged_id.values, lens, flat_grad, id_off,
                                        grad_off, lrs, lr_lens, 0 if req_time is None else req_time,
                                        global_step, 1)
    return self._copy()

  def _raw_update(self, method: str, ragged_id: Ragged, flat_value: torch.Tensor, req_time):
    lens = self._row_lengths(ragged_id)
    ids = ragged_id.values
    flat_value = flat_value.reshape(-1)
    ip = vp = 0
    for t, n, d in zip(self._tables, lens, self._dims):
      if n:
        getattr(t, method)(ids[ip:ip + n], flat_value[vp:vp + n * d].reshape(n, d), req_time)
      ip += n
      vp += n * d
    return self._copy()

  def raw_assign(self, ragged_id: Ragged, flat_value: torch.Tensor, req_time=None):
    return self._raw_update("assign", ragged_id, flat_value, req_time)

  def get_embeddings(self, ragged_id: Ragged, value: torch.Tensor) -> Dict[str, torch.Tensor]:
    values = get_list_from_flat_value(ragged_id, self._dims, value)
    return dict(zip(self._table_names, values))

  def get_ragged_id(self, slot_to_id: Dict[str, torch.Tensor]) -> Ragged:
    dev = self._device()
    empty = torch.zeros(0, dtype=torch.int64, device=dev)
    return concat_1d_tensors(*[
        torch.as_tensor(slot_to_id[n]).to(device=dev, dtype=torch.int64) if n in slot_to_id
        else empty for n in self._table_names])

  def get_flat_value(self, slot_to_value: Dict[str, torch.Tensor]) -> torch.Tensor:
    dev = self._device()
    empty = torch.zeros(0, dtype=torch.float32, device=dev)
    return flatten_n_tensors(*[
        torch.as_tensor(slot_to_value[n]).to(device=dev, dtype=torch.float32)
        if n in slot_to_value else empty for n in self._table_names])

  def save(self, basename) -> "MultiHashTable":
    """multi_hash_table_ops.py:379-385 -> MultiHashTableSaveOp (ops/multi_hash_table_save_restore_ops.cc:
    99-232): every member table's entries, save-time expiry applied, into the snappy record files
    <basename>-%05d-of-%05d plus the per-table counts in <basename>.meta-%05d-of-%05d."""
    from monolith_amd import checkpoint
    dumps, total = [], 0
    for name, t in zip(self._table_names, self._tables):
      if t._table.serving:
        raise RuntimeError("ServingEntryAccessor doesn't support Save")
      ids, rows, ts = t.save_as_tensor(apply_expiry=True)
      dumps.append((name, checkpoint.DumpLayout.from_table_config(t._table.config),
                    ids.cpu().numpy(), rows.cpu().numpy(), ts.cpu().numpy().astype("uint32")))
      total += int(ids.numel())
    saver_parallel = getattr(self._tables[0], "_saver_parallel", -1) if self._tables else -1
    checkpoint.write_multi_shards(str(basename), dumps,
                                  checkpoint.pick_nshards(total, saver_parallel))  # PickNshards :214-222
    return self._copy()

  def restore(self, basename) -> "MultiHashTable":
    """multi_hash_table_ops.py:387-390 -> MultiHashTableRestoreOp (:232-352): entries are upserted
    into the member tables by table NAME (no clear, unlike the single-table restore op); tables of
    the dump this set does not have are skipped, tables the dump does not have stay as they are."""
    from monolith_amd import checkpoint
    import numpy as np
    layouts = {name: checkpoint.DumpLayout.from_table_config(t._table.config)
               for name, t in zip(self._table_names, self._tables)}
    found = checkpoint.read_multi_shards(str(basename), layouts)
    for name, t in zip(self._table_names, self._tables):
      if name not in found:
        continue
      ids, rows, ts = found[name]
      if t._table.serving:
        rows = np.ascontiguousarray(rows[:, :t._table.dim])
      t.restore_from_tensor(torch.from_numpy(ids), torch.from_numpy(rows),
                            torch.from_numpy(ts.astype("int64")), clear=False)
    return self._copy()
