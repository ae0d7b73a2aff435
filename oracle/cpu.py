"""numpy/ctypes front-end of the CPU oracle (oracle/oracle.cc).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by monolith_amd/. All arrays are host numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OPT_SGD, OPT_ADAGRAD, OPT_FTRL = 0, 1, 2
OPT_MOMENTUM, OPT_RMSPROP, OPT_RMSPROP_V2, OPT_ADADELTA = 3, 4, 5, 6
OPT_ADAM, OPT_AMSGRAD = 7, 8
OPT_MOVING_AVERAGE = 9
OPT_BATCH_SOFTMAX = 10
INIT_ZEROS, INIT_ONES, INIT_CONSTANT, INIT_UNIFORM = 0, 1, 2, 3


class SegCfg(C.Structure):
  _fields_ = [("dim", C.c_int32), ("opt_kind", C.c_int32), ("init_acc", C.c_float),
              ("weight_decay", C.c_float), ("l1", C.c_float), ("l2", C.c_float),
              ("beta", C.c_float), ("init_kind", C.c_int32), ("init_a", C.c_float),
              ("init_b", C.c_float), ("momentum", C.c_float), ("epsilon", C.c_float),
              ("averaging_ratio", C.c_float), ("learning_rate", C.c_float),
              ("use_nesterov", C.c_int32), ("beta1", C.c_float), ("beta2", C.c_float),
              ("comp_kind", C.c_int32), ("comp_a", C.c_float), ("comp_b", C.c_float),
              ("stochastic_rounding_float16", C.c_int32)]


class TableCfg(C.Structure):
  _fields_ = [("n_segments", C.c_int32), ("default_expire_days", C.c_int32),
              ("segments", C.POINTER(SegCfg)), ("initial_capacity", C.c_int64),
              ("n_slot_expire", C.c_int32), ("entry_type", C.c_int32),
              ("slot_ids", C.POINTER(C.c_int32)), ("slot_expire_days", C.POINTER(C.c_int32)),
              ("seed", C.c_uint64)]


def build(force=False):
  so = os.path.join(_HERE, "liboracle.so")
  srcs = [os.path.join(_HERE, f) for f in ("oracle.cc", "layout_oracle.cc")]
  if force or not os.path.exists(so) or any(
      os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
  return so


def lib():
  global _LIB
  if _LIB is None:
    so = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(so):
      build()
    _LIB = C.CDLL(so)
    _LIB.orc_table_create.restype = C.c_void_p
    _LIB.orc_filter_create.restype = C.c_void_p
    _LIB.orc_table_size.restype = C.c_int64
    _LIB.orc_table_max_update_ts.restype = C.c_int64
    _LIB.orc_table_export.restype = C.c_int64
    _LIB.orc_fused_reorder_by_indices.restype = C.c_int64
  return _LIB


def _p(a):
  return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _i64(a):
  return np.ascontiguousarray(np.asarray(a, dtype=np.int64).reshape(-1))


def _f32(a):
  return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr_array(arrs):
  return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def seg(dim, opt="sgd", init="zeros", init_acc=0.1, weight_decay=0.0, l1=0.0, l2=0.0, beta=0.0,
        init_a=0.0, init_b=0.0, momentum=0.9, epsilon=0.01, averaging_ratio=0.9,
        learning_rate=0.01, use_nesterov=False, beta1=0.9, beta2=0.99, comp="fp32", comp_a=0.0,
        comp_b=1.0, stochastic_rounding_float16=False):
  ok = {"sgd": OPT_SGD, "adagrad": OPT_ADAGRAD, "ftrl": OPT_FTRL, "momentum": OPT_MOMENTUM,
        "rmsprop": OPT_RMSPROP, "rmspropv2": OPT_RMSPROP_V2, "adadelta": OPT_ADADELTA,
        "adam": OPT_ADAM, "amsgrad": OPT_AMSGRAD, "moving_average": OPT_MOVING_AVERAGE,
        "batch_softmax": OPT_BATCH_SOFTMAX}[opt]
  ik = {"zeros": INIT_ZEROS, "ones": INIT_ONES, "constant": INIT_CONSTANT,
        "uniform": INIT_UNIFORM}[init]
  return SegCfg(dim, ok, init_acc, weight_decay, l1, l2, beta, ik, init_a, init_b, momentum,
                epsilon, averaging_ratio, learning_rate, int(bool(use_nesterov)), beta1, beta2,
                {"fp32": 0, "fp16": 1, "fixed_r8": 2, "one_bit": 3}[comp], comp_a, comp_b,
                int(bool(stochastic_rounding_float16)))


def make_table_cfg(segments, initial_capacity=1, default_expire_days=36500, slot_expire=None,
                   seed=0, serving=False):
  """Returns (TableCfg, keepalive)."""
  segs = (SegCfg * len(segments))(*segments)
  slot_expire = slot_expire or {}
  n = len(slot_expire)
  sl = (C.c_int32 * max(n, 1))(*list(slot_expire.keys()))
  dd = (C.c_int32 * max(n, 1))(*list(slot_expire.values()))
  cfg = TableCfg(len(segments), default_expire_days, segs, initial_capacity, n,
                 2 if serving else 0, sl, dd, seed)
  return cfg, (segs, sl, dd)


class OracleFilter:

  def __init__(self, kind="sliding", capacity=300000000, split_num=7, default_threshold=0,
               slot_thresholds=None):
    slot_thresholds = slot_thresholds or {}
    n = len(slot_thresholds)
    sl = (C.c_int32 * max(n, 1))(*list(slot_thresholds.keys()))
    th = (C.c_int32 * max(n, 1))(*list(slot_thresholds.values()))
    self.h = C.c_void_p(lib().orc_filter_create(
        C.c_int(1 if kind == "sliding" else 0), C.c_int64(capacity), C.c_int(split_num),
        C.c_int(default_threshold), C.c_int(n), sl, th))

  def __del__(self):
    if getattr(self, "h", None):
      lib().orc_filter_destroy(self.h)
      self.h = None

  def add(self, ids, counts):
    ids = _i64(ids)
    counts = np.ascontiguousarray(np.asarray(counts, dtype=np.int32))
    out = np.empty(len(ids), np.int32)
    lib().orc_filter_add(self.h, _p(ids), _p(counts), C.c_int64(len(ids)), _p(out))
    return out

  def get(self, ids):
    ids = _i64(ids)
    out = np.empty(len(ids), np.int32)
    lib().orc_filter_get(self.h, _p(ids), C.c_int64(len(ids)), _p(out))
    return out

  def intercept_gradient(self, ids, grad):
    ids = _i64(ids)
    grad = _f32(grad)
    out = np.empty_like(grad)
    width = grad.size // max(len(ids), 1)
    lib().orc_filter_intercept_gradient(self.h, _p(ids), _p(grad), C.c_int64(len(ids)),
                                        C.c_int64(width), _p(out))
    return out

  def state(self):
    out = np.zeros(5 + 256, np.int64)
    lib().orc_filter_state(self.h, _p(out), C.c_int64(len(out)))
    nb = int(out[3])
    return {"head": int(out[0]), "head_increment": int(out[1]), "failure_count": int(out[2]),
            "n_blocks": nb, "block_total_size": int(out[4]),
            "num_elements": out[5:5 + nb].tolist()}


class OracleTable:

  def __init__(self, segments, initial_capacity=1, default_expire_days=36500, slot_expire=None,
               seed=0, serving=False):
    cfg, keep = make_table_cfg(segments, initial_capacity, default_expire_days, slot_expire, seed,
                               serving)
    self.h = C.c_void_p(lib().orc_table_create(C.byref(cfg)))
    self.dim = lib().orc_table_dim(self.h)
    self.row_floats = lib().orc_table_row_floats(self.h)
    self.n_segments = len(segments)

  def __del__(self):
    if getattr(self, "h", None):
      lib().orc_table_destroy(self.h)
      self.h = None

  def lookup(self, ids):
    ids = _i64(ids)
    out = np.empty((len(ids), self.dim), np.float32)
    lib().orc_table_lookup(self.h, _p(ids), C.c_int64(len(ids)), _p(out))
    return out

  def assign(self, ids, values, req_time=0, filt=None):
    ids = _i64(ids)
    values = _f32(values)
    assert values.size == len(ids) * self.dim
    lib().orc_table_assign(self.h, filt.h if filt else None, _p(ids), _p(values),
                           C.c_int64(len(ids)), C.c_int64(req_time))

  def assign_add(self, ids, values, req_time=0, filt=None):
    ids = _i64(ids)
    values = _f32(values)
    assert values.size == len(ids) * self.dim
    lib().orc_table_assign_add(self.h, filt.h if filt else None, _p(ids), _p(values),
                               C.c_int64(len(ids)), C.c_int64(req_time))

  def apply_gradients(self, ids, grads, lrs, req_time=0, global_step=0, enable_dedup=False,
                      filt=None):
    ids = _i64(ids)
    grads = _f32(grads)
    lrs = _f32(lrs).reshape(-1)
    assert grads.size == len(ids) * self.dim and len(lrs) == self.n_segments
    lib().orc_table_apply_gradients(self.h, filt.h if filt else None, _p(ids), _p(grads),
                                    C.c_int64(len(ids)), _p(lrs), C.c_int64(req_time),
                                    C.c_int64(global_step), C.c_int(int(enable_dedup)))

  def size(self):
    return int(lib().orc_table_size(self.h))

  def contains(self, ids):
    ids = _i64(ids)
    out = np.empty(len(ids), np.uint8)
    lib().orc_table_contains(self.h, _p(ids), C.c_int64(len(ids)), _p(out))
    return out

  def clear(self):
    lib().orc_table_clear(self.h)

  def evict(self, max_update_time):
    lib().orc_table_evict(self.h, C.c_int64(max_update_time))

  def max_update_ts(self):
    return int(lib().orc_table_max_update_ts(self.h))

  def export(self, apply_expiry=False):
    n = self.size()
    ids = np.empty(n, np.int64)
    rows = np.empty((n, self.row_floats), np.float32)
    ts = np.empty(n, np.uint32)
    k = lib().orc_table_export(self.h, C.c_int64(n), C.c_int(int(apply_expiry)), _p(ids),
                               _p(rows), _p(ts))
    return ids[:k], rows[:k], ts[:k]

  def import_rows(self, ids, rows, ts):
    ids = _i64(ids)
    rows = _f32(rows)
    ts = np.ascontiguousarray(np.asarray(ts, dtype=np.uint32))
    lib().orc_table_import(self.h, _p(ids), _p(rows), _p(ts), C.c_int64(len(ids)))


def fused_lookup(tables, ids, fused_slot_size, num_of_shards, metadata_only=False):
  K, N = len(tables), num_of_shards
  ids = _i64(ids)
  fss = np.ascontiguousarray(np.asarray(fused_slot_size, np.int32))
  assert len(fss) == K * N
  hs = (C.c_void_p * K)(*[t.h.value for t in tables])
  splits = np.zeros(N, np.int32)
  id_off = np.zeros(K * N, np.int32)
  emb_off = np.zeros(K * N, np.int32)
  emb_sizes = np.zeros(K * N, np.int32)
  lib().orc_tables_fused_lookup(hs, K, _p(ids), _p(fss), N, None, _p(splits), _p(id_off),
                                _p(emb_off), _p(emb_sizes))
  emb = np.zeros(int(splits.sum()), np.float32)
  if not metadata_only:
    lib().orc_tables_fused_lookup(hs, K, _p(ids), _p(fss), N, _p(emb), _p(splits), _p(id_off),
                                  _p(emb_off), _p(emb_sizes))
  return emb, splits, id_off, emb_off, emb_sizes


def fused_apply_gradient(tables, ids, fused_slot_size, id_grads, id_offsets, grad_offsets,
                         learning_rates, learning_rate_lengths, req_time, global_step,
                         num_of_shards, enable_grad_accumulation=False, filters=None):
  K = len(tables)
  ids = _i64(ids)
  fss = np.ascontiguousarray(np.asarray(fused_slot_size, np.int32))
  g = _f32(id_grads).reshape(-1)
  io = np.ascontiguousarray(np.asarray(id_offsets, np.int32))
  go = np.ascontiguousarray(np.asarray(grad_offsets, np.int32))
  lrs = _f32(learning_rates).reshape(-1)
  lrl = np.ascontiguousarray(np.asarray(learning_rate_lengths, np.int32))
  hs = (C.c_void_p * K)(*[t.h.value for t in tables])
  fs = (C.c_void_p * K)(*[(f.h.value if f else None) for f in filters]) if filters else None
  lib().orc_tables_fused_apply_gradient(hs, fs, K, _p(ids), _p(fss), _p(g), _p(io), _p(go),
                                        _p(lrs), _p(lrl), C.c_int64(req_time),
                                        C.c_int64(global_step), num_of_shards,
                                        C.c_int(int(enable_grad_accumulation)))


def fused_reorder_by_indices(inputs, num_of_shards, dim_sizes, shard_rule=0):
  M = len(inputs)
  ins = [_i64(x) for x in inputs]
  lens = np.array([len(x) for x in ins], np.int64)
  dims = np.ascontiguousarray(np.asarray(dim_sizes, np.int32))
  out_ids = np.empty(int(lens.sum()), np.int64)
  shard_sizes = np.zeros(num_of_shards, np.int32)
  sss = np.zeros(num_of_shards * M, np.int32)
  offs = [np.empty(len(x), np.int32) for x in ins]
  k = lib().orc_fused_reorder_by_indices(_ptr_array(ins), _p(lens), M, num_of_shards, _p(dims),
                                         shard_rule, _p(out_ids), _p(shard_sizes), _p(sss),
                                         _ptr_array(offs))
  return out_ids[:k], shard_sizes, sss, offs


def fused_gather_embeddings_by_input(fused, offsets, dims):
  fused = _f32(fused).reshape(-1)
  offs = [np.ascontiguousarray(np.asarray(o, np.int32).reshape(-1)) for o in offsets]
  lens = np.array([len(o) for o in offs], np.int64)
  d = np.ascontiguousarray(np.asarray(dims, np.int32))
  outs = [np.empty((len(o), int(dd)), np.float32) for o, dd in zip(offs, dims)]
  lib().orc_fused_gather_embeddings_by_input(_p(fused), _ptr_array(offs), _p(lens), _p(d),
                                             len(offs), _ptr_array(outs))
  return outs


def fused_gather_embeddings_by_input_gradient(fused_len, grads, offsets, dims):
  gs = [_f32(g) for g in grads]
  offs = [np.ascontiguousarray(np.asarray(o, np.int32).reshape(-1)) for o in offsets]
  lens = np.array([len(o) for o in offs], np.int64)
  d = np.ascontiguousarray(np.asarray(dims, np.int32))
  out = np.empty(fused_len, np.float32)
  lib().orc_fused_gather_embeddings_by_input_gradient(C.c_int64(fused_len), _ptr_array(gs),
                                                      _ptr_array(offs), _p(lens), _p(d),
                                                      len(offs), _p(out))
  return out


def _reduce(fn, row_ids, values, batch):
  row_ids = _i64(row_ids)
  values = _f32(values)
  n = len(row_ids)
  dim = values.size // max(n, 1) if n else (values.shape[1] if values.ndim == 2 else 0)
  out = np.empty((batch, dim), np.float32)
  fn(_p(row_ids), _p(values), C.c_int64(n), C.c_int64(dim), C.c_int64(batch), _p(out))
  return out


def reduce_sum(row_ids, values, batch):
  return _reduce(lib().orc_reduce_sum, row_ids, values, batch)


def reduce_mean(row_ids, values, batch):
  return _reduce(lib().orc_reduce_mean, row_ids, values, batch)


def _reduce_grad(fn, row_ids, grads):
  row_ids = _i64(row_ids)
  grads = _f32(grads)
  batch, dim = grads.shape
  out = np.empty((len(row_ids), dim), np.float32)
  fn(_p(row_ids), _p(grads), C.c_int64(len(row_ids)), C.c_int64(dim), C.c_int64(batch), _p(out))
  return out


def reduce_sum_gradient(row_ids, grads):
  return _reduce_grad(lib().orc_reduce_sum_gradient, row_ids, grads)


def reduce_mean_gradient(row_ids, grads):
  return _reduce_grad(lib().orc_reduce_mean_gradient, row_ids, grads)


def fused_reduce_sum_and_split(row_ids, values, batch, split_dims):
  row_ids = _i64(row_ids)
  values = _f32(values)
  n, dim = values.shape
  sd = np.ascontiguousarray(np.asarray(split_dims, np.int32))
  outs = [np.empty((batch, int(d)), np.float32) for d in split_dims]
  lib().orc_fused_reduce_sum_and_split(_p(row_ids), _p(values), C.c_int64(n), C.c_int64(dim),
                                       C.c_int64(batch), _p(sd), len(outs), _ptr_array(outs))
  return outs


def fused_reduce_sum_and_split_gradient(row_ids, grads, split_dims):
  row_ids = _i64(row_ids)
  gs = [_f32(g) for g in grads]
  batch = gs[0].shape[0]
  sd = np.ascontiguousarray(np.asarray(split_dims, np.int32))
  out = np.empty((len(row_ids), int(sd.sum())), np.float32)
  lib().orc_fused_reduce_sum_and_split_gradient(_p(row_ids), _ptr_array(gs),
                                                C.c_int64(len(row_ids)), C.c_int64(batch), _p(sd),
                                                len(gs), _p(out))
  return out


def fused_sorted_segment_sum(indices, values, out_rows):
  idx = [np.ascontiguousarray(np.asarray(i, np.int32).reshape(-1)) for i in indices]
  vals = [_f32(v) for v in values]
  lens = np.array([len(i) for i in idx], np.int64)
  dims = np.array([v.shape[1] for v in vals], np.int64)
  rows = np.array(out_rows, np.int64)
  outs = [np.empty((int(r), int(d)), np.float32) for r, d in zip(rows, dims)]
  lib().orc_fused_sorted_segment_sum(_ptr_array(idx), _ptr_array(vals), _p(lens), _p(dims),
                                     _p(rows), len(idx), _ptr_array(outs))
  return outs


def first_n(row_ids, values, batch, max_seq):
  row_ids = _i64(row_ids)
  values = _f32(values)
  n, dim = values.shape
  out = np.empty((batch, max_seq, dim), np.float32)
  lib().orc_first_n(_p(row_ids), _p(values), C.c_int64(n), C.c_int64(dim), C.c_int64(batch),
                    C.c_int64(max_seq), _p(out))
  return out


def adagrad(num, norm, grad, lr, wd):
  num, norm, grad = _f32(num).copy(), _f32(norm).copy(), _f32(grad)
  lib().orc_adagrad(_p(num), _p(norm), _p(grad), C.c_int64(num.size), C.c_float(lr),
                    C.c_float(wd))
  return num, norm


class OrcLayoutSlice(C.Structure):  # == mono_layout_slice
  _fields_ = [("feature_idx", C.c_int32), ("start", C.c_int32), ("dim", C.c_int32),
              ("pooling", C.c_int32), ("max_seq", C.c_int32), ("out_index", C.c_int32),
              ("out_stride", C.c_int32), ("out_offset", C.c_int32), ("addn", C.c_int32),
              ("first", C.c_int32)]


def _layout_common(plan_slices, bufs, fid_offset, feature_offset, nfl_offset):
  sl = (OrcLayoutSlice * max(1, len(plan_slices)))(*[
      OrcLayoutSlice(s.feature_idx, s.start, s.dim, s.pooling, s.max_seq, s.out_index,
                     s.out_stride, s.out_offset, s.addn, s.first) for s in plan_slices])
  fo = np.ascontiguousarray(np.asarray(fid_offset).astype(np.uint64))
  feat = np.ascontiguousarray(np.asarray(feature_offset, np.int32))
  nfl = np.ascontiguousarray(np.asarray(nfl_offset).astype(np.uint32))
  dims = np.ascontiguousarray(np.array([d for _, d in bufs], np.int32))
  return sl, fo, feat, nfl, dims


def embedding_to_layout(plan_slices, out_shapes, bufs, fid_offset, feature_offset, nfl_offset,
                        batch):
  """bufs: list of (float32 1-D array, row dim) in fid_offset buffer-index order."""
  sl, fo, feat, nfl, dims = _layout_common(plan_slices, bufs, fid_offset, feature_offset, nfl_offset)
  arrs = [_f32(a).reshape(-1) for a, _ in bufs]
  outs = [np.empty([batch if d == -1 else d for d in shape], np.float32) for shape in out_shapes]
  floats = np.array([o.size for o in outs], np.int64)
  lib().orc_embedding_to_layout(sl, len(plan_slices), _ptr_array(arrs), _p(dims), len(arrs),
                                _p(fo), C.c_int64(len(fo)), _p(feat), C.c_int64(len(feat)),
                                _p(nfl), len(nfl), C.c_int32(batch), _ptr_array(outs), _p(floats),
                                len(outs))
  return outs


def embedding_to_layout_grad(plan_slices, slice_idx_of, max_slice_num, bufs, fid_offset,
                             feature_offset, nfl_offset, batch, out_grads, mean_divides=False):
  """bufs: list of (floats, row dim); returns one flat gradient array per buffer."""
  sl, fo, feat, nfl, dims = _layout_common(plan_slices, bufs, fid_offset, feature_offset, nfl_offset)
  grads = [np.empty(int(n), np.float32) for n, _ in bufs]
  floats = np.array([int(n) for n, _ in bufs], np.int64)
  sidx = np.ascontiguousarray(np.asarray(slice_idx_of, np.int32))
  og = [_f32(g).reshape(-1) for g in out_grads]
  lib().orc_embedding_to_layout_grad(sl, _p(sidx), len(plan_slices), int(max_slice_num),
                                     _ptr_array(grads), _p(dims), _p(floats), len(grads), _p(fo),
                                     C.c_int64(len(fo)), _p(feat), C.c_int64(len(feat)), _p(nfl),
                                     len(nfl), C.c_int32(batch), _ptr_array(og),
                                     C.c_int(int(mean_divides)))
  return grads


def bench_lookup_apply_mt(segments, ids_batches, grads, lr, threads):
  """bench.py's cpu_baseline: (seconds, rows) of lookup + apply_gradients(dedup) over the batches
  with `threads` shard-owning threads (orc_bench_lookup_apply_mt)."""
  cfg, keep = make_table_cfg(segments)
  ids = np.ascontiguousarray(np.stack([_i64(b) for b in ids_batches]))
  g = _f32(grads)
  rows = C.c_int64(0)
  fn = lib().orc_bench_lookup_apply_mt
  fn.restype = C.c_double
  dt = fn(C.byref(cfg), _p(ids), _p(g), C.c_int64(ids.shape[1]), int(ids.shape[0]), C.c_float(lr),
          int(threads), C.byref(rows))
  del keep
  return float(dt), int(rows.value)


# ------------------------------------------------------------------ sharding_sparse_fids (8f-2)
def sharding_sparse_fids(feature_tables, features, ps_num, unique):
  """ShardingSparseFidsV2, sequential restatement (test infrastructure).

  data/kernels/parse_sparse_feature.cc:92-141 (sorted tables / features, feature_in_table_index,
  output_pre_index), :173-211 (FillFidList: shard = uint64 fid % ps_num; unique = hash map per
  (feature, shard), index = first-occurrence rank), :213-288 (CreateOffsetTensor: nfl_offset with
  bit 31 for shared features, feature_offset = running FID count per run),
  parse_sparse_feature.h:1583-1636 (fid_list = features of a table back to back per ps,
  fid_list_row_splits = prefix sums of the per-feature counts).

  feature_tables: {feature name: table name} (every configured feature)
  features: {feature name: ("shared", [fids]) | ("rows", [[fids] per sample])}; missing = absent
  Returns (fid_list [table*ps_num+ps] int64 arrays, row_splits, fid_offset uint64, feature_offset
  int32, nfl_offset uint32)."""
  names = sorted(feature_tables)
  tables = sorted(set(feature_tables.values()))
  tfeat = {t: [n for n in names if feature_tables[n] == t] for t in tables}
  pre, pre_index = {}, 0
  for t in tables:
    pre[t] = pre_index
    pre_index += max(len(tfeat[t]), 1) * ps_num
  lists = {(n, p): [] for n in names for p in range(ps_num)}
  seen = {(n, p): {} for n in names for p in range(ps_num)}
  fid_offset, feature_offset, nfl_offset = [], [], []
  for n in names:
    t = feature_tables[n]
    nfl = len(feature_offset)
    f = features.get(n)
    if f is not None and f[0] == "shared":
      nfl |= 1 << 31
    nfl_offset.append(nfl)
    if f is None:
      continue
    runs = [f[1]] if f[0] == "shared" else f[1]
    for run in runs:
      feature_offset.append(len(fid_offset))
      for fid in run:
        p = (int(fid) & 0xFFFFFFFFFFFFFFFF) % ps_num
        out_index = pre[t] + p * len(tfeat[t]) + tfeat[t].index(n)
        if unique:
          idx = seen[(n, p)].get(fid)
          if idx is None:
            idx = len(lists[(n, p)])
            seen[(n, p)][fid] = idx
            lists[(n, p)].append(fid)
        else:
          idx = len(lists[(n, p)])
          lists[(n, p)].append(fid)
        fid_offset.append((out_index << 32) | idx)
  fid_list, row_splits = [], []
  for t in tables:
    for p in range(ps_num):
      parts = [lists[(n, p)] for n in tfeat[t]]
      fid_list.append(np.array([x for part in parts for x in part], np.int64))
      row_splits.append(np.cumsum([0] + [len(part) for part in parts]).astype(np.int64))
  return (fid_list, row_splits, np.array(fid_offset, np.uint64),
          np.array(feature_offset, np.int32), np.array(nfl_offset, np.uint32))


# ------------------------------------------------------------------ touched-key set (8f-4)
class OracleKeySet:
  """HopscotchHashSet<FID>, sequential (runtime/hopscotch/hopscotch_hash_set.cc:104-193): an
  insert that finds MORE than `capacity` keys drops the whole set first and reports how many;
  GetAndClear returns everything. Test infrastructure."""

  def __init__(self, capacity):
    self.capacity = int(capacity)
    self.keys = set()

  def insert(self, ids):
    dropped = 0
    for k in ids:
      if len(self.keys) > self.capacity:
        dropped += len(self.keys)
        self.keys.clear()
      self.keys.add(int(k))
    return dropped

  def steal(self):
    out = sorted(self.keys)
    self.keys.clear()
    return out
