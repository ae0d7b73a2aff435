"""Sync-training embedding exchange — the surface of
monolith/native_training/distributed_ps_sync.py:110-545 (DistributedMultiTypeHashTableMpi).

One process per GPU. Rank r owns the rows of shard r (fused_reorder_by_indices' shard rule); a
step is

  lookup:           F1 reorder/dedup -> a2a(sizes int32[K]) + a2a-v(FIDs int64)
                    -> fused multi-table lookup on the owner -> a2a-v(rows fp32)
                    -> F2 gather by the offsets F1 produced
  apply_gradients:  F2 gradient (in-order scatter-add) -> a2a-v(grads fp32, splits swapped)
                    -> fused multi-table optimize on the owner

Pipelining (queue_configs["enable_pipelined_fwda2a"] > 0, the reference's switch,
distributed_ps_sync.py:218-224): apply_gradients(next_slot_to_id=...) runs the FIRST HALF of the
next batch's lookup — reorder, size exchange with its host read-back, FID exchange, and the
owner's slot resolution + update order (fused_lookup_hint) — on a side stream beside this batch's
update; the lookup that follows only gathers rows, exchanges them and runs F2. The reference's
queues run the whole lookup ahead and train on stale rows; here the rows are read after the
update has landed, so the result is the serial one bit for bit (the tests compare them).

The reference's Horovod / BytePS alltoall (distributed_ps_sync.py:170-193,230-283,369-496) is
`torch.distributed.all_to_all_single` here: RCCL over xGMI on the GPUs (backend "nccl"), gloo in
the CPU tests. Message sizes are exchanged explicitly (hvd.alltoall(shard_sizes)); the rows /
grads exchanges need no size message because both sides can compute them (recv_emb_splits =
reshape(sharded_slot_sizes, [N, K]) @ dims, :211-216).

`ops` names the module providing fused_reorder_by_indices / fused_gather_embeddings_by_input /
..._gradient; it defaults to the HIP ops. (The CPU tests inject the oracle there to cover the
exchange logic with gloo; the product path never does.)"""
import copy
import os
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from monolith_amd import multi_type_hash_table


def _world(group=None) -> Tuple[int, int]:
  if dist.is_available() and dist.is_initialized():
    return dist.get_rank(group), dist.get_world_size(group)
  return 0, 1


def _forced_collective() -> bool:
  return (os.environ.get("MONOLITH_AMD_FORCE_COLLECTIVE") == "1" and dist.is_available() and
          dist.is_initialized())


def alltoall(tensor: torch.Tensor, splits, recv_splits=None, group=None):
  """hvd.alltoall(tensor, splits[, recv_splits]) on dim 0. Returns (received, recv_splits).
  `splits` / `recv_splits` are host int sequences (length world)."""
  rank, world = _world(group)
  splits = [int(x) for x in splits]
  if world == 1 and not _forced_collective():
    # (MONOLITH_AMD_FORCE_COLLECTIVE=1 sends even a one-rank exchange through the backend: the
    # way to exercise the RCCL calls on a single-GPU box)
    return tensor, splits
  if tensor.is_cuda and dist.get_backend(group) == "gloo":
    # gloo has no device all-to-all: stage through the host (single-GPU rehearsals only; the
    # production backend is "nccl" = RCCL, which exchanges device buffers over xGMI directly)
    out, recv_splits = alltoall(tensor.cpu(), splits, recv_splits, group)
    return out.to(tensor.device), recv_splits
  if recv_splits is None:
    # size message first: one int64 per peer
    dev = tensor.device
    s = torch.tensor(splits, dtype=torch.int64, device=dev)
    r = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(r, s, group=group)
    recv_splits = [int(x) for x in r.cpu().tolist()]
  else:
    recv_splits = [int(x) for x in recv_splits]
  out = torch.empty((sum(recv_splits),) + tuple(tensor.shape[1:]), dtype=tensor.dtype,
                    device=tensor.device)
  dist.all_to_all_single(out, tensor.contiguous(), output_split_sizes=recv_splits,
                         input_split_sizes=splits, group=group)
  return out, recv_splits


class ForwardPrefetch:
  """The first half of a lookup, done ahead (see the module docstring): what lookup() needs to go
  on from the owner's fused lookup. `event` marks its end on the stream it ran on."""

  def __init__(self, sorted_slot_keys, pack, id_flat_t, id_size_np, sss_np, event):
    self.sorted_slot_keys = sorted_slot_keys
    self.pack = pack
    self.id_flat_t = id_flat_t
    self.id_size_np = id_size_np
    self.sss_np = sss_np
    self.event = event

  def result(self):  # (a caller that only wants the reorder, as from a ReorderHandle)
    return self.pack


class DistributedMultiTypeHashTableMpi(multi_type_hash_table.BaseMultiTypeHashTable):
  """distributed_ps_sync.py:110-545. table_factory(rank) builds this rank's local
  (Merged)MultiTypeHashTable; it must offer fused_lookup / fused_apply_gradient /
  get_table_dim_sizes."""

  lookup_takes_bundle = True

  def __init__(self, shard_num: int,
               table_factory: Callable[[int], multi_type_hash_table.BaseMultiTypeHashTable],
               queue_configs: Optional[Dict[str, int]] = None, group=None, ops=None,
               grad_exchange_bits: Optional[int] = None):
    # grad_exchange_bits: 32 (default) or 16 = the gradient rows cross the fabric as binary16
    # (distributed_ps_sync.py:353-356,381-385: tf.cast(grad_flat, tf.float16) before the exchange,
    # back to float32 behind it). Default from MONOLITH_WITH_BYTEPS_BWD_CAST, the reference's
    # own switch (:44-47).
    if grad_exchange_bits is None:
      grad_exchange_bits = int(os.environ.get("MONOLITH_WITH_BYTEPS_BWD_CAST", "32"))
    assert grad_exchange_bits in (16, 32), "grad_exchange_bits must be 16 or 32"
    self._grad_bits = grad_exchange_bits
    self._shard_num = shard_num
    self._group = group
    rank, world = _world(group)
    assert world == self._shard_num, "world size %d != shard_num %d" % (world, shard_num)
    self._index = rank
    self._table = table_factory(self._index)
    self._output_dims = self._table.get_table_dim_sizes()
    self._dims_np = np.asarray(self._output_dims, dtype=np.int64)
    self._world_size = world
    self._side_stream = None
    self._prefetched = None
    # The reference's pipelining queues. enable_pipelined_fwda2a > 0: the first half of the next
    # lookup runs beside the update (module docstring). The others (embedding_prefetch_capacity,
    # enable_async_optimize, enable_pipelined_bwda2a) trade staleness for overlap in the reference;
    # they are accepted and have no effect: every exchange here is stream-ordered.
    # "announce_owner_lookup" (no reference counterpart) adds the owner's fused_lookup_hint to the
    # first half: fewer kernels in line, more launches — pays when the step is not host-bound.
    self._queue_configs = queue_configs or {}
    self._pipelined_fwd = int(self._queue_configs.get("enable_pipelined_fwda2a", 0)) > 0
    self._announce = int(self._queue_configs.get("announce_owner_lookup", 0)) > 0
    if ops is None:
      from monolith_amd import distribution_ops as ops  # HIP ops; raises without the library
    self._ops = ops

  def _exchange_ids(self, slot_num, pack):
    """Size exchange + FID exchange of a reordered batch: (id_flat_t, id_size_np [N, K], sss_np)."""
    all_fids, shard_sizes, sharded_slot_sizes, _ = pack
    # M: num_of_ids, N: num_of_shards, K: num_of_merged_tables, E: total embedding floats.
    # id_flat_t: [M'], id_size_flat_t: [K*N] (source-shard major)
    # The per-(shard, slot) counts travel first (fixed size, K per peer): their row sums are the
    # FID counts every peer is about to send, so the FID exchange needs no size message of its
    # own (one collective and one host synchronisation fewer per step than hvd.alltoall's).
    # Host metadata stays in numpy from here on: the GPU is idle until the owner's lookup is
    # launched, and every torch CPU op on the way costs microseconds.
    N, K = self._shard_num, slot_num
    shard_sizes_h = np.asarray(shard_sizes).reshape(-1).tolist()
    sss_np = np.ascontiguousarray(sharded_slot_sizes, dtype=np.int32).reshape(N, K)
    single = self._world_size == 1 and not _forced_collective()
    if single:
      id_size_np = sss_np
    else:
      sss = torch.from_numpy(sss_np.reshape(-1))
      id_size_flat_t, _ = alltoall(sss.to(all_fids.device), [K] * N, recv_splits=[K] * N,
                                   group=self._group)
      id_size_np = id_size_flat_t.cpu().numpy().reshape(N, K)
    recv_fid_splits = id_size_np.sum(1).tolist()
    id_flat_t, _ = alltoall(all_fids, shard_sizes_h, recv_splits=recv_fid_splits,
                            group=self._group)
    return id_flat_t, id_size_np, sss_np

  def lookup(self, slot_to_id: Dict[str, torch.Tensor], auxiliary_bundle: Dict,
             early_reorder_indicies_res_pack=None):
    sorted_slot_keys = sorted(slot_to_id.keys())
    slot_num = len(slot_to_id)
    sorted_input = [slot_to_id[k] for k in sorted_slot_keys]
    ahead = None
    if isinstance(early_reorder_indicies_res_pack, ForwardPrefetch):
      ahead = early_reorder_indicies_res_pack
      assert ahead.sorted_slot_keys == sorted_slot_keys, "prefetched lookup of other slots"
      early_reorder_indicies_res_pack = ahead.pack
      if ahead.event is not None:  # (its kernels and collectives ran on the side stream)
        cur = torch.cuda.current_stream(ahead.id_flat_t.device)
        cur.wait_event(ahead.event)
        # (allocated under a side stream, read on this one from here on: the ids the owner looks
        # up, the reorder's id list and its embedding offsets)
        ahead.id_flat_t.record_stream(cur)
        all_fids_, _, _, offs_ = ahead.pack
        for t_ in [all_fids_] + list(offs_):
          if isinstance(t_, torch.Tensor) and t_.is_cuda:
            t_.record_stream(cur)
    if early_reorder_indicies_res_pack:
      if hasattr(early_reorder_indicies_res_pack, "result"):  # a reorder still in flight
        early_reorder_indicies_res_pack = early_reorder_indicies_res_pack.result()
      pack = early_reorder_indicies_res_pack
    else:
      # (a reorder started ahead and never asked for would leave the library waiting for its
      # finish: retire it before this batch's own)
      stale, self._prefetched = self._prefetched, None
      if stale is not None and hasattr(stale, "result"):
        stale.result()
      pack = self._ops.fused_reorder_by_indices(sorted_input, self._shard_num, self._output_dims)
    all_fids, shard_sizes, sharded_slot_sizes, fused_embedding_offsets = pack

    if ahead is not None:
      id_flat_t, id_size_np, sss_np = ahead.id_flat_t, ahead.id_size_np, ahead.sss_np
    else:
      id_flat_t, id_size_np, sss_np = self._exchange_ids(slot_num, pack)
    id_size_flat_h = torch.from_numpy(id_size_np.reshape(-1))
    auxiliary_bundle["shard_sizes"] = shard_sizes
    auxiliary_bundle["fused_embedding_offsets"] = fused_embedding_offsets
    auxiliary_bundle["id_flat_t"] = id_flat_t
    auxiliary_bundle["id_size_flat_t"] = id_size_flat_h

    # fused_embeddings: [E], embedding_splits: [N], id_offsets / emb_offsets: [K*N]
    fused_embeddings, embedding_splits, id_offsets, emb_offsets, _ = \
        self._table.fused_lookup(id_flat_t, id_size_np.reshape(-1), self._shard_num)
    auxiliary_bundle["embedding_splits"] = embedding_splits
    auxiliary_bundle["id_offsets"] = id_offsets
    auxiliary_bundle["emb_offsets"] = emb_offsets
    recv_emb_splits_np = sss_np.astype(np.int64) @ self._dims_np
    recv_emb_splits = torch.from_numpy(recv_emb_splits_np)
    auxiliary_bundle["recv_emb_splits"] = recv_emb_splits

    # recv_embeddings: [E']
    recv_embeddings, _ = alltoall(fused_embeddings, embedding_splits.tolist(),
                                  recv_splits=recv_emb_splits_np.tolist(), group=self._group)
    auxiliary_bundle["recv_embeddings_size"] = int(recv_embeddings.numel())

    outputs = self._ops.fused_gather_embeddings_by_input(recv_embeddings, fused_embedding_offsets,
                                                         self._output_dims)
    slot_to_embedding = {k: outputs[i] for i, k in enumerate(sorted_slot_keys)}
    return slot_to_embedding, auxiliary_bundle

  def prefetch_forward(self, next_slot_to_id: Dict[str, torch.Tensor], handle=None):
    """The first half of the NEXT lookup, now: finish (or run) its reorder, exchange its sizes
    and FIDs, optionally announce the owner's lookup. Call it right after apply_gradients() has
    launched the update — the host waits in here for the reorder's sizes and the size exchange,
    beside that update instead of in the step — and pass the result to lookup(...,
    early_reorder_indicies_res_pack=...). `handle`: a reorder_begin() of the same ids started
    earlier (the earlier the better: it then has the whole backward pass to finish)."""
    keys = sorted(next_slot_to_id.keys())
    first = next_slot_to_id[keys[0]]
    on_gpu = isinstance(first, torch.Tensor) and first.is_cuda
    if on_gpu:
      if self._side_stream is None:
        self._side_stream = torch.cuda.Stream(first.device)
      side = self._side_stream
      if handle is None:
        # the reorder runs here: behind whatever produced the ids on the current stream. (With a
        # handle the side stream must NOT wait for the current one — that is the update just
        # launched; result() orders it behind the reorder's own stream.)
        side.wait_stream(torch.cuda.current_stream(first.device))
        for v in next_slot_to_id.values():
          if isinstance(v, torch.Tensor) and v.is_cuda:
            v.record_stream(side)
      ctx = torch.cuda.stream(side)
    else:  # (host tensors: the CPU tests' oracle ops, no streams)
      import contextlib
      side, ctx = None, contextlib.nullcontext()
    with ctx:
      if handle is None:
        handle = self._ops.fused_reorder_by_indices([next_slot_to_id[k] for k in keys],
                                                    self._shard_num, self._output_dims)
      pack = handle.result() if hasattr(handle, "result") else handle
      id_flat_t, id_size_np, sss_np = self._exchange_ids(len(keys), pack)
      hint = getattr(self._table, "fused_lookup_hint", None) if self._announce else None
      if hint is not None and id_flat_t.is_cuda and id_flat_t.numel():
        hint(id_flat_t, id_size_np.reshape(-1), self._shard_num, 1)
      event = side.record_event() if side is not None else None
    return ForwardPrefetch(keys, pack, id_flat_t, id_size_np, sss_np, event)

  def reorder_begin(self, slot_to_id: Dict[str, torch.Tensor]):
    """Start the reorder of a coming batch (on the current stream, without waiting for it) and
    return the handle to pass to lookup(..., early_reorder_indicies_res_pack=handle): what the
    reference's input pipeline does ahead of the step (distributed_ps_sync.py:313-331). One
    batch ahead at most."""
    sorted_input = [slot_to_id[k] for k in sorted(slot_to_id.keys())]
    return self._ops.fused_reorder_by_indices_begin(sorted_input, self._shard_num,
                                                    self._output_dims)

  def assign(self, slot_to_id_and_value):  # "assign is broken" in the reference (:317-321)
    raise NotImplementedError

  def assign_add(self, slot_to_id_and_value):  # (:323-327)
    raise NotImplementedError

  def apply_gradients(self, slot_to_id_and_grad: Dict[str, Tuple[torch.Tensor, torch.Tensor]],
                      auxiliary_bundle: Dict, global_step, req_time=None, next_slot_to_id=None):
    """Fused update (distributed_ps_sync.py:330-521). Gradient rows of one FID coming from
    different ranks stay separate entries of the owner's id list and are applied as separate
    optimizer steps in source-shard-major order.

    next_slot_to_id: the ids of the NEXT batch, if the caller has them (the reference's input
    pipeline does: early_reorder_indicies_res_pack, :313-331). Their reorder is then issued on a
    side stream at the point of this call where the launches cost the step nothing — behind the
    gradient kernels, ahead of the update's wait — and the handle is kept for
    take_prefetched_reorder()."""
    if req_time is None:
      req_time = 0
    sorted_slot_keys = sorted(slot_to_id_and_grad.keys())
    sorted_grads = [slot_to_id_and_grad[k][1] for k in sorted_slot_keys]

    recv_embeddings_size = auxiliary_bundle.pop("recv_embeddings_size")
    fused_embedding_offsets = auxiliary_bundle.pop("fused_embedding_offsets")
    grad_flat = self._ops.fused_gather_embeddings_by_input_gradient(
        recv_embeddings_size, sorted_grads, fused_embedding_offsets, self._output_dims)

    recv_emb_splits = auxiliary_bundle.pop("recv_emb_splits")
    embedding_splits = auxiliary_bundle.pop("embedding_splits")
    if self._grad_bits == 16:  # to nearest even, as tf.cast; overflow -> inf, like the reference
      grad_flat = grad_flat.to(torch.float16)
    grad_flat_t, _ = alltoall(grad_flat, recv_emb_splits.tolist(),
                              recv_splits=embedding_splits.tolist(), group=self._group)
    if self._grad_bits == 16:
      grad_flat_t = grad_flat_t.to(torch.float32)
    auxiliary_bundle.pop("shard_sizes")
    prefetched = None
    if next_slot_to_id is not None and grad_flat.is_cuda:
      if self._side_stream is None:
        self._side_stream = torch.cuda.Stream(grad_flat.device)
      # the next batch's ids may come out of work on the current stream (an H2D copy, id
      # preprocessing): the side stream starts behind it, and the tensors stay alive for it
      self._side_stream.wait_stream(torch.cuda.current_stream(grad_flat.device))
      for v in next_slot_to_id.values():
        if isinstance(v, torch.Tensor) and v.is_cuda:
          v.record_stream(self._side_stream)
      with torch.cuda.stream(self._side_stream):
        prefetched = self.reorder_begin(next_slot_to_id)

    updated_table = self._table.fused_apply_gradient(
        auxiliary_bundle.pop("id_flat_t"), auxiliary_bundle.pop("id_size_flat_t"), grad_flat_t,
        auxiliary_bundle.pop("id_offsets"), auxiliary_bundle.pop("emb_offsets"), global_step,
        req_time, self._shard_num)
    updated = self._copy_with_new_table(updated_table)
    if next_slot_to_id is not None and self._pipelined_fwd:
      # behind the update's launches: the host wait inside costs the step nothing
      prefetched = updated.prefetch_forward(next_slot_to_id, prefetched)
    updated._prefetched = prefetched
    return updated

  def take_prefetched_reorder(self):
    """The reorder apply_gradients(next_slot_to_id=...) started, for lookup's
    early_reorder_indicies_res_pack (None if there is none)."""
    pack, self._prefetched = self._prefetched, None
    return pack

  def as_op(self, name=None):
    return self._table.as_op(name)

  def get_table_dim_sizes(self) -> List[int]:
    return self._table.get_table_dim_sizes()

  def _copy_with_new_table(self, table):
    copied = copy.copy(self)
    copied._table = table
    return copied
