"""Routing, exchange and pooling ops of the sparse-embedding path — the surface of the reference's
monolith/native_training/distribution_ops.py, executed by libmonolith_hip.so on the current HIP
stream. torch tensors are only buffer carriers; every kernel is reached through the C ABI
(include/monolith_hip.h). There is no CPU fallback: CPU tensors are rejected.

Differentiable ops are torch.autograd.Functions whose backward is the reference's registered
gradient op (the @tf.RegisterGradient functions of distribution_ops.py).
"""
import ctypes as C
import os
from typing import List, Sequence, Tuple

import numpy as np
import torch

from monolith_amd import _lib
from monolith_amd._lib import vp


def _dev(t: torch.Tensor, who: str):
  if not t.is_cuda:
    raise _lib.MonolithHipError("%s: tensors must live on the GPU (no CPU fallback)" % who)
  return t.device


def _ptrs(tensors: Sequence[torch.Tensor]):
  arr = (C.c_void_p * max(1, len(tensors)))()
  for i, t in enumerate(tensors):
    arr[i] = t.data_ptr()
  return arr


def _i32s(xs):
  return (C.c_int32 * max(1, len(xs)))(*[int(x) for x in xs])


def _i64s(xs):
  return (C.c_int64 * max(1, len(xs)))(*[int(x) for x in xs])


def _flat(x: torch.Tensor, dtype) -> torch.Tensor:
  """x as a flat contiguous tensor of `dtype`; x itself when it already is one (the hot callers
  pass such tensors, and every torch call costs microseconds of launch-path host time)."""
  if x.dtype == dtype and x.dim() == 1 and x.is_contiguous():
    return x
  return x.reshape(-1).to(dtype).contiguous()


def _host_i32(carr) -> torch.Tensor:
  """ctypes int32 array -> CPU int32 tensor (own copy)."""
  return torch.from_numpy(np.ctypeslib.as_array(carr).copy())


def _row_ids(id_indices: torch.Tensor) -> torch.Tensor:
  """The reference passes `[n, 1]` (or `[n, k]`) index matrices whose column 0 is the output row
  (reduce_op.cc:37-47); a plain `[n]` vector is accepted as well."""
  if id_indices.dim() == 2:
    id_indices = id_indices[:, 0]
  return id_indices.to(torch.int64).contiguous()


def _batch(id_length) -> int:
  if isinstance(id_length, torch.Tensor):
    return int(id_length.reshape(-1)[0].item())
  if isinstance(id_length, (list, tuple)):
    return int(id_length[0])
  return int(id_length)


# ----------------------------------------------------------------------------- routing
def fused_reorder_by_indices(inputs: List[torch.Tensor], num_of_shards: int,
                             dim_sizes: List[int]):
  """distribution_ops.py:223-272 -> ops/fused_reorder_by_indices.cc:36-112.

  Per-slot dedup (first occurrence keeps its place), ids grouped shard-major / slot-minor,
  and for every input id the float offset of its embedding inside the fused embedding buffer
  the lookup of `output` will produce.

  Returns (output int64[U], shard_sizes int32[N] (host), sharded_slot_sizes int32[N*M] (host),
  fused_embedding_offsets: list of int32 device tensors shaped like the inputs)."""
  return fused_reorder_by_indices_begin(inputs, num_of_shards, dim_sizes).result()


class ReorderHandle:
  """A fused_reorder_by_indices in flight (fused_reorder_by_indices_begin). result() blocks until
  the sizes are on the host, makes the current stream wait for the reorder's stream if they
  differ, and returns what fused_reorder_by_indices returns."""

  def __init__(self, inputs, flat, lens, out_ids, offsets, N, M, stream):
    self._keep = (inputs, flat)  # the id arrays must outlive the kernels
    self._lens, self._out_ids, self._offsets, self._N, self._M = lens, out_ids, offsets, N, M
    self._stream = stream
    self._done = torch.cuda.Event()
    self._done.record(stream)
    self._res = None

  def result(self):
    if self._res is None:
      N, M = self._N, self._M
      shard_sizes = (C.c_int32 * N)()
      slot_sizes = (C.c_int32 * (N * M))()
      n_unique = C.c_int64(0)
      _lib.check(_lib.lib.mono_fused_reorder_finish(C.byref(n_unique), shard_sizes, slot_sizes))
      cur = torch.cuda.current_stream(self._out_ids.device)
      if cur != self._stream:
        cur.wait_event(self._done)
        for t in [self._out_ids] + self._offsets:
          t.record_stream(cur)
      inputs = self._keep[0]
      offs = PlannedOffsets((o if n > 0 and x.dim() == 1 else o[:n].reshape(x.shape))
                            for o, n, x in zip(self._offsets, self._lens, inputs))
      offs.plan_token = int(_lib.lib.mono_fused_reorder_last_plan())
      self._res = (self._out_ids[:n_unique.value], _host_i32(shard_sizes), _host_i32(slot_sizes),
                   offs)
      self._keep = None
    return self._res


def fused_reorder_by_indices_begin(inputs: List[torch.Tensor], num_of_shards: int,
                                   dim_sizes: List[int]) -> ReorderHandle:
  """Launch fused_reorder_by_indices on the current stream without waiting for its sizes; the
  reference computes the reorder ahead of the step in its input pipeline and hands the result
  to lookup as early_reorder_indicies_res_pack (distributed_ps_sync.py:313-331). One reorder
  may be in flight at a time: call result() before the next begin."""
  assert len(inputs) == len(dim_sizes) and len(inputs) > 0
  dev = _dev(inputs[0], "fused_reorder_by_indices")
  M, N = len(inputs), int(num_of_shards)
  rank0_empty_shard = os.environ.get("MONOLITH_SYNC_EMPTY_RANK0_PS_SHARD", "1") == "1"
  shard_rule = 1 if (N > 4 and rank0_empty_shard) else 0
  flat = [_flat(x, torch.int64) for x in inputs]
  lens = [x.numel() for x in flat]
  total = sum(lens)
  out_ids = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
  offsets = [torch.empty(max(n, 1), dtype=torch.int32, device=dev) for n in lens]
  stream = torch.cuda.current_stream(dev)
  _lib.check(_lib.lib.mono_fused_reorder_begin(
      _ptrs(flat), _i64s(lens), M, N, _i32s(dim_sizes), shard_rule, vp(out_ids), _ptrs(offsets),
      C.c_void_p(stream.cuda_stream)))
  return ReorderHandle(inputs, flat, lens, out_ids, offsets, N, M, stream)


class PlannedOffsets(list):
  """fused_embedding_offsets as fused_reorder_by_indices returns them: a list of int32 tensors
  that also carries the token of the summation order the library prepared for their gradient
  (include/monolith_hip.h, mono_fused_gather_embeddings_by_input_gradient_planned). The token
  only helps while the tensors are passed on unmodified; a plain list works everywhere too."""
  plan_token = 0


def fused_gather_embeddings_by_input_gradient(fused_embeddings, grads: List[torch.Tensor],
                                              embedding_offsets: List[torch.Tensor],
                                              embedding_dims: List[int],
                                              plan_token=None) -> torch.Tensor:
  """distribution_ops.py:512-519 -> ops/map_id_to_embedding_op.cc:294-343. `fused_embeddings`
  only supplies the output length (a tensor or an int). Accumulation order is the input order
  (slot by slot, row by row), i.e. the reference's CPU kernel — deterministic."""
  fused_len = int(fused_embeddings.numel()) if isinstance(fused_embeddings, torch.Tensor) \
      else int(fused_embeddings)
  dev = _dev(grads[0], "fused_gather_embeddings_by_input_gradient")
  M = len(grads)
  gs = [g if (g.dtype == torch.float32 and g.is_contiguous()) else g.to(torch.float32).contiguous()
        for g in grads]
  offs = [_flat(o, torch.int32) for o in embedding_offsets]
  lens = [o.numel() for o in offs]
  out = torch.empty(max(fused_len, 1), dtype=torch.float32, device=dev)
  if plan_token is None:
    plan_token = getattr(embedding_offsets, "plan_token", 0)
  _lib.check(_lib.lib.mono_fused_gather_embeddings_by_input_gradient_planned(
      C.c_int64(plan_token), C.c_int64(fused_len), _ptrs(gs), _ptrs(offs), _i64s(lens),
      _i32s(embedding_dims), M, vp(out), _lib.stream_ptr(dev)))
  return out if fused_len == out.numel() else out[:fused_len]


def _fused_gather_forward(fused, dims, offsets):
  dev = _dev(fused, "fused_gather_embeddings_by_input")
  fused_c = _flat(fused, torch.float32)
  offs = [_flat(o, torch.int32) for o in offsets]
  lens = [o.numel() for o in offs]
  outs = [torch.empty((n, d), dtype=torch.float32, device=dev) for n, d in zip(lens, dims)]
  _lib.check(_lib.lib.mono_fused_gather_embeddings_by_input(
      vp(fused_c), C.c_int64(fused_c.numel()), _ptrs(offs), _i64s(lens), _i32s(dims), len(offs),
      _ptrs(outs), _lib.stream_ptr(dev)))
  return fused_c, offs, outs


class _FusedGather(torch.autograd.Function):

  @staticmethod
  def forward(ctx, fused, dims, plan_token, *offsets):
    fused_c, offs, outs = _fused_gather_forward(fused, dims, offsets)
    ctx.dims = list(dims)
    ctx.plan_token = plan_token
    ctx.fused_len = int(fused_c.numel())
    ctx.save_for_backward(*offs)
    return tuple(outs)

  @staticmethod
  def backward(ctx, *grads):
    offs = ctx.saved_tensors
    g = fused_gather_embeddings_by_input_gradient(ctx.fused_len, list(grads), list(offs), ctx.dims,
                                                  ctx.plan_token)
    return (g, None, None) + (None,) * len(offs)


def fused_gather_embeddings_by_input(fused_embeddings: torch.Tensor,
                                     fused_embedding_offsets: List[torch.Tensor],
                                     embedding_dims: List[int]) -> List[torch.Tensor]:
  """distribution_ops.py:505-509 -> ops/map_id_to_embedding_op.cc:239-292: slot m's output row i
  is fused_embeddings[offset_m[i] : offset_m[i] + dim_m]."""
  if not (fused_embeddings.requires_grad and torch.is_grad_enabled()):
    # nothing to differentiate (the sync-training path calls the gradient op itself)
    return _fused_gather_forward(fused_embeddings, list(embedding_dims), fused_embedding_offsets)[2]
  return list(_FusedGather.apply(fused_embeddings, list(embedding_dims),
                                 getattr(fused_embedding_offsets, "plan_token", 0),
                                 *fused_embedding_offsets))


# ----------------------------------------------------------------------------- pooling
def _reduce_call(fn, row_ids, values, batch):
  dev = _dev(values, "reduce")
  v = values.to(torch.float32).contiguous()
  n, dim = v.shape
  out = torch.empty((batch, dim), dtype=torch.float32, device=dev)
  _lib.check(fn(vp(row_ids), vp(v), C.c_int64(n), C.c_int64(dim), C.c_int64(batch), vp(out),
                _lib.stream_ptr(dev)))
  return out


def _reduce_grad_call(fn, row_ids, grads):
  dev = _dev(grads, "reduce gradient")
  g = grads.to(torch.float32).contiguous()
  batch, dim = g.shape
  n = int(row_ids.numel())
  out = torch.empty((n, dim), dtype=torch.float32, device=dev)
  _lib.check(fn(vp(row_ids), vp(g), C.c_int64(n), C.c_int64(dim), C.c_int64(batch), vp(out),
                _lib.stream_ptr(dev)))
  return out


class _ReduceSum(torch.autograd.Function):

  @staticmethod
  def forward(ctx, row_ids, values, batch):
    ctx.save_for_backward(row_ids)
    return _reduce_call(_lib.lib.mono_reduce_sum, row_ids, values, batch)

  @staticmethod
  def backward(ctx, grad):  # distribution_ops.py:577-582
    (row_ids,) = ctx.saved_tensors
    return None, _reduce_grad_call(_lib.lib.mono_reduce_sum_gradient, row_ids, grad), None


class _ReduceMean(torch.autograd.Function):

  @staticmethod
  def forward(ctx, row_ids, values, batch):
    ctx.save_for_backward(row_ids)
    return _reduce_call(_lib.lib.mono_reduce_mean, row_ids, values, batch)

  @staticmethod
  def backward(ctx, grad):  # distribution_ops.py:551-556
    (row_ids,) = ctx.saved_tensors
    return None, _reduce_grad_call(_lib.lib.mono_reduce_mean_gradient, row_ids, grad), None


def reduce_sum(id_indices: torch.Tensor, id_values: torch.Tensor, id_length, name=None):
  """distribution_ops.py:559-574 -> ops/reduce_op.cc:29-51: out[id_length, dim], row
  id_indices[i, 0] accumulates id_values[i] in input order."""
  return _ReduceSum.apply(_row_ids(id_indices), id_values, _batch(id_length))


def reduce_mean(id_indices: torch.Tensor, id_values: torch.Tensor, id_length):
  """distribution_ops.py:522-533 -> ops/reduce_op.cc:55-87 (empty rows stay zero)."""
  return _ReduceMean.apply(_row_ids(id_indices), id_values, _batch(id_length))


class _ReduceSumSplit(torch.autograd.Function):

  @staticmethod
  def forward(ctx, row_ids, values, batch, split_dims):
    dev = _dev(values, "fused_reduce_sum_and_split")
    v = values.to(torch.float32).contiguous()
    n, dim = v.shape
    outs = [torch.empty((batch, d), dtype=torch.float32, device=dev) for d in split_dims]
    _lib.check(_lib.lib.mono_fused_reduce_sum_and_split(
        vp(row_ids), vp(v), C.c_int64(n), C.c_int64(dim), C.c_int64(batch), _i32s(split_dims),
        len(split_dims), _ptrs(outs), _lib.stream_ptr(dev)))
    ctx.save_for_backward(row_ids)
    ctx.split_dims = list(split_dims)
    ctx.batch = batch
    return tuple(outs)

  @staticmethod
  def backward(ctx, *grads):  # distribution_ops.py:662-668
    (row_ids,) = ctx.saved_tensors
    dev = grads[0].device
    gs = [g.to(torch.float32).contiguous() for g in grads]
    n = int(row_ids.numel())
    out = torch.empty((n, sum(ctx.split_dims)), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib.mono_fused_reduce_sum_and_split_gradient(
        vp(row_ids), _ptrs(gs), C.c_int64(n), C.c_int64(ctx.batch), _i32s(ctx.split_dims),
        len(gs), vp(out), _lib.stream_ptr(dev)))
    return None, out, None, None


def fused_reduce_sum_and_split(id_indices: torch.Tensor, id_values: torch.Tensor, id_length,
                               split_dims: List[int], name: str = None):
  """distribution_ops.py:635-659 -> ops/reduce_op.cc:232-277: reduce_sum whose [batch, dim]
  result is written directly as M tensors [batch, split_dims[m]]."""
  return list(_ReduceSumSplit.apply(_row_ids(id_indices), id_values, _batch(id_length),
                                    list(split_dims)))


class _SegmentSum(torch.autograd.Function):

  @staticmethod
  def forward(ctx, n, out_rows, *tensors):
    indices, values = tensors[:n], tensors[n:]
    dev = _dev(values[0], "fused_sorted_segment_sum")
    idx = [i.reshape(-1).to(torch.int32).contiguous() for i in indices]
    vals = [v.to(torch.float32).contiguous() for v in values]
    lens = [int(i.numel()) for i in idx]
    dims = [int(v.shape[1]) for v in vals]
    outs = [torch.empty((r, d), dtype=torch.float32, device=dev) for r, d in zip(out_rows, dims)]
    _lib.check(_lib.lib.mono_fused_sorted_segment_sum(
        _ptrs(idx), _ptrs(vals), _i64s(lens), _i64s(dims), _i64s(out_rows), n, _ptrs(outs),
        _lib.stream_ptr(dev)))
    ctx.save_for_backward(*idx)
    ctx.n = n
    return tuple(outs)

  @staticmethod
  def backward(ctx, *grads):  # distribution_ops.py:624-632: gather_nd(grad, indices)
    idx = ctx.saved_tensors
    gv = [g.index_select(0, i.to(torch.int64)) for g, i in zip(grads, idx)]
    return (None, None) + (None,) * ctx.n + tuple(gv)


def fused_sorted_segment_sum(indices: List[torch.Tensor], values: List[torch.Tensor],
                             shapes: List[Sequence[int]]):
  """distribution_ops.py:608-621 -> ops/reduce_op.cu.cc:53-161: N sorted segment sums in one
  call; output i has shape `shapes[i]` = [rows_i, dim_i]."""
  n = len(indices)
  out_rows = [int(s[0]) for s in shapes]
  return list(_SegmentSum.apply(n, out_rows, *indices, *values))


# ----------------------------------------------------------------------------- layout fusion
def _layout_buffers(plan, embeddings_list, fid_list_row_split, ps_num, version):
  """The embedding buffers a fid_offset's high word indexes: (tensor index, float offset, row
  dim, floats). V1: the tensors themselves; V2: cut per feature (feature_configs.split_...)."""
  from monolith_amd import feature_configs as fc
  if version == 2:
    return fc.split_embeddings_v2(plan, embeddings_list, fid_list_row_split, ps_num)
  return [(i, 0, int(e.shape[1]), int(e.numel())) for i, e in enumerate(embeddings_list)]


def _layout_call(plan, bufs, base_tensors, fid_offset, feature_offset, nfl_offset, batch, outs,
                 grads_mode):
  n_s = len(plan.slices)
  slices = (type(plan.slices[0]) * max(1, n_s))(*plan.slices) if n_s else None
  ptrs = (C.c_void_p * max(1, len(bufs)))(
      *[base_tensors[t].data_ptr() + 4 * off for (t, off, _, _) in bufs])
  dims = _i32s([d for (_, _, d, _) in bufs])
  floats = _i64s([f for (_, _, _, f) in bufs])
  nfl = torch.as_tensor(nfl_offset).cpu().to(torch.int64).tolist()
  h_nfl = (C.c_uint32 * max(1, len(nfl)))(*[int(x) & 0xFFFFFFFF for x in nfl])
  out_ptrs = _ptrs(outs)
  out_floats = _i64s([int(o.numel()) for o in outs])
  dev = fid_offset.device
  args = [slices, n_s, ptrs, dims, floats, len(bufs), vp(fid_offset), C.c_int64(fid_offset.numel()),
          vp(feature_offset), C.c_int64(feature_offset.numel()), h_nfl, len(nfl), C.c_int32(batch),
          out_ptrs, out_floats, len(outs)]
  if grads_mode:
    _lib.check(_lib.lib.mono_embedding_to_layout_grad(*args, 0, _lib.stream_ptr(dev)))
  else:
    _lib.check(_lib.lib.mono_embedding_to_layout(*args, _lib.stream_ptr(dev)))


def fused_embedding_to_layout_grad(nfl_offset, feature_offset, fid_offset, batch_size,
                                   embeddings_list: List[torch.Tensor],
                                   fid_list_row_split: List[torch.Tensor],
                                   layout_tensors_grad: List[torch.Tensor], variant_type: str,
                                   feature_cfgs, ps_num: int, parallel_flag=0,
                                   version: int = 2) -> List[torch.Tensor]:
  """distribution_ops.py:408-436 -> ops/fused_embedding_to_layout.cc:703-976: one gradient tensor
  per embeddings_list entry. MEAN slices scatter the undivided gradient, as the reference does."""
  from monolith_amd import feature_configs as fc
  assert variant_type in {"example", "example_batch", "examplebatch", "instance"}
  plan = fc.build_layout_plan(feature_cfgs)
  dev = _dev(fid_offset, "fused_embedding_to_layout_grad")
  bufs = _layout_buffers(plan, embeddings_list, fid_list_row_split, ps_num, version)
  # one allocation, one view per tensor: the library zero-fills and sums straight into it (its
  # buffers then follow each other in memory; separate tensors cost a scratch image and a copy)
  flat = torch.empty(sum(int(e.numel()) for e in embeddings_list), dtype=torch.float32, device=dev)
  grads, at = [], 0
  for e in embeddings_list:
    grads.append(flat[at:at + e.numel()].view(e.shape))
    at += int(e.numel())
  gouts = [g.to(torch.float32).contiguous() for g in layout_tensors_grad]
  _layout_call(plan, bufs, grads, fid_offset.contiguous(),
               feature_offset.to(torch.int32).contiguous(), nfl_offset, _batch(batch_size), gouts,
               True)
  return grads


class _EmbeddingToLayout(torch.autograd.Function):

  @staticmethod
  def forward(ctx, meta, fid_offset, feature_offset, *embeddings_list):
    plan, fid_list_row_split, nfl_offset, batch, feature_cfgs, ps_num, version = meta
    embs = [e.to(torch.float32).contiguous() for e in embeddings_list]
    bufs = _layout_buffers(plan, embs, fid_list_row_split, ps_num, version)
    dev = fid_offset.device
    outs = [torch.empty([batch if d == -1 else d for d in shape], dtype=torch.float32, device=dev)
            for shape in plan.out_shapes]
    _layout_call(plan, bufs, embs, fid_offset, feature_offset, nfl_offset, batch, outs, False)
    ctx.meta = meta
    ctx.save_for_backward(fid_offset, feature_offset, *embs)
    return tuple(outs)

  @staticmethod
  def backward(ctx, *grads):  # distribution_ops.py:360-405
    plan, fid_list_row_split, nfl_offset, batch, feature_cfgs, ps_num, version = ctx.meta
    fid_offset, feature_offset = ctx.saved_tensors[:2]
    embs = list(ctx.saved_tensors[2:])
    g = fused_embedding_to_layout_grad(nfl_offset, feature_offset, fid_offset, batch, embs,
                                       fid_list_row_split, list(grads), "example_batch",
                                       feature_cfgs, ps_num, version=version)
    return (None, None, None) + tuple(g)


def fused_embedding_to_layout(embeddings_list: List[torch.Tensor],
                              fid_list_row_split: List[torch.Tensor], fid_offset: torch.Tensor,
                              feature_offset: torch.Tensor, nfl_offset, batch_size,
                              variant_type: str, feature_cfgs, ps_num: int, parallel_flag: int = 0,
                              version: int = 2) -> List[torch.Tensor]:
  """distribution_ops.py:313-357 -> ops/fused_embedding_to_layout.cc:403-695.

  fid_offset: uint64 (index1 << 32 | row) as an int64 device tensor; feature_offset: int32 device
  tensor; nfl_offset: uint32 values (bit 31 = shared feature), host side; fid_list_row_split:
  host int64 tensors (V2). Returns one tensor per OutConfig shape, layouts in sorted-name order."""
  from monolith_amd import feature_configs as fc
  assert variant_type in {"example", "example_batch", "examplebatch", "instance"}
  plan = fc.build_layout_plan(feature_cfgs)
  _dev(fid_offset, "fused_embedding_to_layout")
  meta = (plan, fid_list_row_split, nfl_offset, _batch(batch_size), feature_cfgs, ps_num, version)
  return list(_EmbeddingToLayout.apply(meta, fid_offset.contiguous(),
                                       feature_offset.to(torch.int32).contiguous(),
                                       *embeddings_list))


def fused_lookup_to_layout(hash_tables: List, fid_list: List[torch.Tensor],
                           fid_list_row_split: List[torch.Tensor], fid_offset: torch.Tensor,
                           feature_offset: torch.Tensor, nfl_offset, batch_size, variant_type: str,
                           feature_cfgs, ps_num: int, version: int = 2) -> List[torch.Tensor]:
  """Fused form of what PartitionedHashTable.lookup does in two steps (distributed_ps.py:889-964):
  `hash_tables[i].lookup(fid_list[i])` for every (table, ps) list, then fused_embedding_to_layout
  over the results. Same arguments as fused_embedding_to_layout with (hash_tables, fid_list) in
  place of embeddings_list, same outputs bit for bit; no [n_unique, dim] tensor is materialised
  (forward only: training keeps the two-step form, whose embeddings the backward needs)."""
  from monolith_amd import feature_configs as fc
  assert variant_type in {"example", "example_batch", "examplebatch", "instance"}
  plan = fc.build_layout_plan(feature_cfgs)
  dev = _dev(fid_offset, "fused_lookup_to_layout")
  if len(hash_tables) != len(fid_list):
    raise ValueError("one hash table per fid_list entry is required")
  ids = [torch.as_tensor(f, dtype=torch.int64).to(dev).contiguous() for f in fid_list]

  class _Shape:  # what split_embeddings_v2 reads of an embedding buffer
    def __init__(self, rows, dim):
      self.shape, self._n = (rows, dim), rows * dim
    def numel(self):
      return self._n
  shapes = [_Shape(int(i.numel()), int(t.dim_size)) for i, t in zip(ids, hash_tables)]
  bufs = _layout_buffers(plan, shapes, fid_list_row_split, ps_num, version)
  batch = _batch(batch_size)
  outs = [torch.empty([batch if d == -1 else d for d in shape], dtype=torch.float32, device=dev)
          for shape in plan.out_shapes]
  n_s = len(plan.slices)
  slices = (type(plan.slices[0]) * max(1, n_s))(*plan.slices) if n_s else None
  nb = len(bufs)
  tabs = (C.c_void_p * max(1, nb))(*[hash_tables[t].table.h for (t, _, _, _) in bufs])
  lists = (C.c_void_p * max(1, nb))(
      *[ids[t].data_ptr() + 8 * (off // max(1, d)) for (t, off, d, _) in bufs])
  counts = _i64s([f // max(1, d) for (_, _, d, f) in bufs])
  nfl = torch.as_tensor(nfl_offset).cpu().to(torch.int64).tolist()
  h_nfl = (C.c_uint32 * max(1, len(nfl)))(*[int(x) & 0xFFFFFFFF for x in nfl])
  fo = fid_offset.contiguous()
  fe = feature_offset.to(torch.int32).contiguous()
  _lib.check(_lib.lib.mono_lookup_to_layout(
      slices, n_s, tabs, lists, counts, nb, vp(fo), C.c_int64(fo.numel()), vp(fe),
      C.c_int64(fe.numel()), h_nfl, len(nfl), C.c_int32(batch), _ptrs(outs),
      _i64s([int(o.numel()) for o in outs]), len(outs), _lib.stream_ptr(dev)))
  return outs
