"""Slot-keyed composition of hash tables — the surface of
monolith/native_training/multi_type_hash_table.py:38-392.

"Write returns a new table object" chaining of the reference (TF control dependencies) is kept as
an API shape: write methods return a shallow copy that shares the device tables; work is
stream-ordered, so nothing else is needed for ordering."""
import abc
import copy
import hashlib
import re
from collections import defaultdict
from typing import Callable, Dict, List, Tuple

import torch

from monolith_amd import entry, hash_table_ops


class BaseMultiTypeHashTable(abc.ABC):  # multi_type_hash_table.py:38-92

  @abc.abstractmethod
  def lookup(self, slot_to_id: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    pass

  @abc.abstractmethod
  def assign(self, slot_to_id_and_value) -> "BaseMultiTypeHashTable":
    pass

  @abc.abstractmethod
  def assign_add(self, slot_to_id_and_value) -> "BaseMultiTypeHashTable":
    pass

  @abc.abstractmethod
  def apply_gradients(self, slot_to_id_and_grad, *args) -> "BaseMultiTypeHashTable":
    pass

  @abc.abstractmethod
  def as_op(self, name=None):
    pass

  @abc.abstractmethod
  def get_table_dim_sizes(self) -> List[int]:
    pass


class MultiTypeHashTable(BaseMultiTypeHashTable):
  """Hash tables of different entry types distinguished by slot; inputs are maps slot -> ids /
  values (multi_type_hash_table.py:96-224). hash_table_factory(name_suffix, config)."""

  def __init__(self, slot_to_config: Dict[str, entry.HashTableConfigInstance],
               hash_table_factory: Callable[[str, entry.HashTableConfigInstance],
                                            hash_table_ops.BaseHashTable]):
    self._slot_to_config = slot_to_config
    self._hash_tables = {}
    self._hash_table_resources = []
    for slot in sorted(self._slot_to_config.keys()):  # the order is part of the fused-op contract
      config = self._slot_to_config[slot]
      self._hash_tables[slot] = hash_table_factory(slot, config)
      self._hash_table_resources.append(self._hash_tables[slot].as_op())

  def _learning_rates(self):
    """Flattened learning rates + per-table lengths for the fused optimize op
    (multi_type_hash_table.py:115-129). The reference builds graph tensors that are evaluated
    every step; here the learning-rate functions are called at every apply."""
    tensors, lengths = [], []
    for slot in sorted(self._slot_to_config.keys()):
      lrs = self._slot_to_config[slot].call_learning_rate_fns()
      tensors.extend(lrs)
      lengths.append(len(lrs))
    return tensors, lengths

  def lookup(self, slot_to_id: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {slot: self._hash_tables[slot].lookup(id) for slot, id in slot_to_id.items()}

  def assign(self, slot_to_id_and_value) -> "MultiTypeHashTable":
    return self._update("assign", slot_to_id_and_value)

  def assign_add(self, slot_to_id_and_value) -> "MultiTypeHashTable":
    return self._update("assign_add", slot_to_id_and_value)

  def apply_gradients(self, slot_to_id_and_grad, *args, **kwargs) -> "MultiTypeHashTable":
    return self._update("apply_gradients", slot_to_id_and_grad, *args, **kwargs)

  def _update(self, method_name: str, slot_to_id_and_tensor, *args, **kwargs):
    updated_tables = dict(self._hash_tables)
    for slot, (id, tensor) in slot_to_id_and_tensor.items():
      updated_tables[slot] = getattr(self._hash_tables[slot], method_name)(id, tensor, *args,
                                                                           **kwargs)
    return self._copy_with_new_tables(updated_tables)

  def as_op(self, name=None):
    return [table.as_op() for table in self._hash_tables.values()]

  def _copy_with_new_tables(self, tables) -> "MultiTypeHashTable":
    copied = copy.copy(self)
    copied.__dict__["_hash_tables"] = tables
    copied.__dict__["_hash_table_resources"] = [
        tables[slot].as_op() for slot in sorted(self._slot_to_config.keys())
    ]
    return copied

  def fused_lookup(self, ids, fused_slot_size, num_of_shards: int):
    """multi_type_hash_table.py:189-192."""
    return hash_table_ops.fused_lookup(self._hash_table_resources, ids, fused_slot_size,
                                       num_of_shards)

  def fused_lookup_hint(self, ids, fused_slot_size, num_of_shards: int, ahead: int = 1) -> None:
    """The fused_lookup `ahead` calls away, announced (hash_table_ops.fused_lookup_hint)."""
    hash_table_ops.fused_lookup_hint(self._hash_table_resources, ids, fused_slot_size,
                                     num_of_shards, ahead)

  def fused_apply_gradient(self, ids, fused_slot_size, id_grads, id_offsets, grad_offsets,
                           global_step, req_time, num_of_shards: int,
                           enable_grad_accumulation: bool = False) -> "MultiTypeHashTable":
    """multi_type_hash_table.py:196-218."""
    lr_tensors, lr_lengths = self._learning_rates()
    hash_table_ops.fused_apply_gradient(self._hash_table_resources, ids, fused_slot_size, id_grads,
                                        id_offsets, grad_offsets, lr_tensors, lr_lengths, req_time,
                                        global_step, num_of_shards, enable_grad_accumulation)
    return copy.copy(self)

  def get_table_dim_sizes(self) -> List[int]:
    return [self._hash_tables[slot].dim_size for slot in sorted(self._slot_to_config.keys())]


def merged_slot_name(slots: List[str]) -> Tuple[str, str]:
  """multi_type_hash_table.py:261-263: merged table name = md5(",".join(slots))."""
  concat = ",".join(slots)
  return concat, hashlib.md5(concat.encode()).hexdigest()


class MergedMultiTypeHashTable(BaseMultiTypeHashTable):
  """A decorator that merges slots which have the same embedding config. The caller must make
  sure ids in different slots are different (multi_type_hash_table.py:240-392)."""

  def __init__(self, slot_to_config: Dict[str, entry.HashTableConfigInstance],
               factory: Callable[[Dict[str, entry.HashTableConfigInstance]],
                                 BaseMultiTypeHashTable]):
    self._slot_to_config = slot_to_config
    self._slot_mapping: Dict[str, str] = {}  # feature/slot -> merged_slot
    deduped_config_to_slots = defaultdict(list)
    for slot in sorted(slot_to_config):
      deduped_config_to_slots[str(slot_to_config[slot])].append(slot)

    merged_slot_to_config = {}
    for key, slots in deduped_config_to_slots.items():
      _, merged_slot = merged_slot_name(slots)
      merged_config = copy.copy(slot_to_config[slots[0]])
      merged_config.extra_restore_names = list(merged_config.extra_restore_names)
      # "fc_slot_*" restores from the dump written under "slot_*"
      old_slots = [slot[3:] if re.match("^fc_slot_[0-9]*$", slot) else slot for slot in slots]
      _, old_merged_slot = merged_slot_name(old_slots)
      if old_merged_slot != merged_slot:
        merged_config.extra_restore_names.append(old_merged_slot)
      merged_slot_to_config[merged_slot] = merged_config
      for slot in slots:
        self._slot_mapping[slot] = merged_slot

    self._merged_slot_to_config = merged_slot_to_config
    self._table = factory(merged_slot_to_config)

  @property
  def slot_mapping(self):
    return self._slot_mapping

  def lookup(self, slot_to_id: Dict[str, torch.Tensor], auxiliary_bundle=None,
             early_reorder_indicies_res_pack=None):
    if auxiliary_bundle is None:
      auxiliary_bundle = {}
    if early_reorder_indicies_res_pack:
      merged_slot_to_id, merged_slot_to_sizes, res_pack = early_reorder_indicies_res_pack
      auxiliary_bundle["merged_slot_to_sizes"] = merged_slot_to_sizes
      merged_slot_to_embedding, auxiliary_bundle = self._table.lookup(
          merged_slot_to_id, auxiliary_bundle, res_pack)
    else:
      merged_slot_to_id, merged_slot_to_sizes = self._get_merged_to_indexed_tensor(slot_to_id)
      auxiliary_bundle["merged_slot_to_sizes"] = merged_slot_to_sizes
      if getattr(self._table, "lookup_takes_bundle", False):
        # the sync-training inner table (DistributedMultiTypeHashTableMpi) threads its exchange
        # metadata through the bundle and returns it
        merged_slot_to_embedding, auxiliary_bundle = self._table.lookup(merged_slot_to_id,
                                                                        auxiliary_bundle)
      else:
        merged_slot_to_embedding = self._table.lookup(merged_slot_to_id)

    merged_slot_to_slots = defaultdict(list)
    for k in sorted(self._slot_mapping.keys()):
      if k in slot_to_id:
        merged_slot_to_slots[self._slot_mapping[k]].append(k)

    merged_slot_to_sizes = auxiliary_bundle.pop("merged_slot_to_sizes")
    slot_to_embedding = {}
    for merged_slot, emb in merged_slot_to_embedding.items():
      sizes = merged_slot_to_sizes[merged_slot]
      slots = merged_slot_to_slots[merged_slot]
      embs = torch.split(emb, sizes, dim=0)
      for slot, e in zip(slots, embs):
        slot_to_embedding[slot] = e
    if auxiliary_bundle:
      return slot_to_embedding, auxiliary_bundle
    return slot_to_embedding

  def assign(self, slot_to_id_and_value) -> "MergedMultiTypeHashTable":
    return self._update(self._table.assign, slot_to_id_and_value)

  def assign_add(self, slot_to_id_and_value) -> "MergedMultiTypeHashTable":
    return self._update(self._table.assign_add, slot_to_id_and_value)

  def apply_gradients(self, slot_to_id_and_grad, *args, **kwargs) -> "MergedMultiTypeHashTable":
    return self._update(self._table.apply_gradients, slot_to_id_and_grad, *args, **kwargs)

  def _update(self, method, slot_to_id_and_tensor, *args, **kwargs):
    slot_to_id = {k: v[0] for k, v in slot_to_id_and_tensor.items()}
    merged_slot_to_id, _ = self._get_merged_to_indexed_tensor(slot_to_id)
    slot_to_tensor = {k: v[1] for k, v in slot_to_id_and_tensor.items()}
    merged_slot_to_tensor, _ = self._get_merged_to_indexed_tensor(slot_to_tensor)
    merged = {slot: (merged_slot_to_id[slot], merged_slot_to_tensor[slot])
              for slot in merged_slot_to_id}
    return self._copy_with_new_table(method(merged, *args, **kwargs))

  def as_op(self, name=None):
    return self._table.as_op(name)

  def _get_merged_to_indexed_tensor(self, slot_to_tensor: Dict[str, torch.Tensor]):
    merged_slot_to_tensors = defaultdict(list)
    merged_slot_to_sizes = defaultdict(list)
    for slot in sorted(slot_to_tensor.keys()):  # sorted: guarantees the merging order
      tensor = torch.as_tensor(slot_to_tensor[slot])
      merged_slot = self._slot_mapping[slot]
      merged_slot_to_sizes[merged_slot].append(int(tensor.shape[0]))
      merged_slot_to_tensors[merged_slot].append(tensor)
    return ({k: torch.cat(v, dim=0) for k, v in merged_slot_to_tensors.items()},
            dict(merged_slot_to_sizes))

  def _copy_with_new_table(self, table: BaseMultiTypeHashTable) -> "MergedMultiTypeHashTable":
    copied = copy.copy(self)
    copied._table = table
    return copied

  def get_table_dim_sizes(self) -> List[int]:
    return [self._merged_slot_to_config[slot].table_config.dim_size
            for slot in sorted(self._merged_slot_to_config.keys())]
