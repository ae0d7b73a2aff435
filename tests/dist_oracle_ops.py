"""Oracle-backed stand-ins for the local (per-rank) ops, used ONLY by the CPU gloo tests to drive
monolith_amd.distributed_ps_sync's exchange logic without a GPU. Same call signatures as
monolith_amd.distribution_ops / multi_type_hash_table.MultiTypeHashTable."""
import os

import numpy as np
import torch

from oracle import cpu


class OracleOps:

  @staticmethod
  def fused_reorder_by_indices(inputs, num_of_shards, dim_sizes):
    rank0_empty = os.environ.get("MONOLITH_SYNC_EMPTY_RANK0_PS_SHARD", "1") == "1"
    rule = 1 if (num_of_shards > 4 and rank0_empty) else 0
    ins = [np.asarray(x, np.int64).reshape(-1) for x in inputs]
    o, s, ss, offs = cpu.fused_reorder_by_indices(ins, num_of_shards, dim_sizes, shard_rule=rule)
    return (torch.from_numpy(o.copy()), torch.from_numpy(s.copy()), torch.from_numpy(ss.copy()),
            [torch.from_numpy(x.copy()) for x in offs])

  @staticmethod
  def fused_gather_embeddings_by_input(fused, offsets, dims):
    outs = cpu.fused_gather_embeddings_by_input(fused.numpy(), [o.numpy() for o in offsets], dims)
    return [torch.from_numpy(o) for o in outs]

  @staticmethod
  def fused_gather_embeddings_by_input_gradient(fused_len, grads, offsets, dims):
    g = cpu.fused_gather_embeddings_by_input_gradient(
        int(fused_len), [np.asarray(x, np.float32) for x in grads], [o.numpy() for o in offsets],
        dims)
    return torch.from_numpy(g)


class OracleMultiTable:
  """slot -> OracleTable; tables ordered by sorted slot name like MultiTypeHashTable."""

  def __init__(self, slot_to_seg, lrs):
    self.slots = sorted(slot_to_seg)
    self.tables = {s: cpu.OracleTable([slot_to_seg[s]]) for s in self.slots}
    self.lrs = [float(lrs[s]) for s in self.slots]

  def get_table_dim_sizes(self):
    return [self.tables[s].dim for s in self.slots]

  def fused_lookup(self, ids, fused_slot_size, num_of_shards):
    emb, splits, id_off, emb_off, sizes = cpu.fused_lookup(
        [self.tables[s] for s in self.slots], np.asarray(ids, np.int64),
        np.asarray(fused_slot_size, np.int32), num_of_shards)
    return (torch.from_numpy(emb), torch.from_numpy(splits), torch.from_numpy(id_off),
            torch.from_numpy(emb_off), torch.from_numpy(sizes))

  def fused_apply_gradient(self, ids, fused_slot_size, id_grads, id_offsets, grad_offsets,
                           global_step, req_time, num_of_shards, enable_grad_accumulation=False):
    cpu.fused_apply_gradient([self.tables[s] for s in self.slots], np.asarray(ids, np.int64),
                             np.asarray(fused_slot_size, np.int32),
                             np.asarray(id_grads, np.float32), np.asarray(id_offsets, np.int32),
                             np.asarray(grad_offsets, np.int32), self.lrs, [1] * len(self.slots),
                             int(req_time), int(global_step), num_of_shards,
                             enable_grad_accumulation)
    return self

  def as_op(self, name=None):
    return self
