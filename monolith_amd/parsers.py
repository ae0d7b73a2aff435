"""Mirror of monolith/native_training/data/parsers.py:402-456 (`sharding_sparse_fids`) for FIDs
that are already ragged tensors on the GPU.

The reference op (data/kernels/parse_sparse_feature.cc, parse_sparse_feature.h) walks protobuf
Example / ExampleBatch / Instance messages on the CPU; decoding protobuf is outside the hot path
(SURVEY.md 8f-2), so the input here is what that walk produces: per configured feature the FIDs
of every sample (a Ragged), or one FID list shared by the whole batch (SharedFeature). Everything
after the walk — shard routing, per-(feature, shard) dedup in first-occurrence order, the
fid_list / fid_list_row_split / fid_offset / feature_offset / nfl_offset tensors the fused layout
op consumes — runs on the device (mono_sharding_sparse_fids)."""
import ctypes as C
from typing import Dict, NamedTuple, Union

import numpy as np
import torch

from monolith_amd import _lib
from monolith_amd.embedding_combiners import Ragged
from monolith_amd.feature_configs import FeatureConfigs

SHARED_FLAG = 1 << 31  # parse_sparse_feature.h:163 shard_flag_


class SharedFeature(NamedTuple):
  """A NamedFeatureList of type SHARED: one FID list used by every sample of the batch."""
  values: torch.Tensor


class ShardingPlan(NamedTuple):
  feature_names: list   # sorted
  table_names: list     # sorted
  table_features: list  # per table: its feature names, sorted
  out_index: np.ndarray  # int32[M, ps_num]: FeatureInfo::GetFidOutputIndex
  n_out: int


def build_sharding_plan(feature_cfgs: FeatureConfigs, ps_num: int, version: int = 2) -> ShardingPlan:
  """The constructor of ShardingSparseFidsOp (parse_sparse_feature.cc:92-141): tables and features
  sorted by name, feature_in_table_index by sorted feature name, output_pre_index advancing by
  max(#features, 1) * ps_num per table (V2) or table_index * ps_num (V1)."""
  fcs = feature_cfgs.feature_configs
  names = sorted(fcs.keys())
  tables = sorted({fc.table for fc in fcs.values()})
  tfeat = [[n for n in names if fcs[n].table == t] for t in tables]
  out_index = np.zeros((len(names), ps_num), np.int32)
  pre = 0
  for ti, feats in enumerate(tfeat):
    for fi, n in enumerate(feats):
      for ps in range(ps_num):
        out_index[names.index(n), ps] = (pre + ps * len(feats) + fi) if version == 2 \
            else ti * ps_num + ps
    pre += max(len(feats), 1) * ps_num
  n_out = pre if version == 2 else len(tables) * ps_num
  return ShardingPlan(names, tables, tfeat, out_index, max(n_out, 1))


def sharding_sparse_fids(features: Dict[str, Union[Ragged, SharedFeature]], ps_num: int,
                         feature_cfgs: FeatureConfigs, unique: bool, input_type: str = "ragged",
                         parallel_flag: int = 0, fid_list_ret_list: bool = False, version: int = 2):
  """parsers.py:402-456, version 2 (the fused-layout path). `features` replaces the reference's
  serialized-protobuf `tensor`; names missing from it are features absent from the input (their
  nfl_offset entry points at the next feature's first run, as in CreateOffsetTensor).
  `parallel_flag` selects a CPU threading scheme in the reference and is ignored.

  Returns (fid_list, fid_offset, feature_offset, nfl_offset, batch_size, fid_list_row_split):
  dicts keyed "<table>:<ps>" (lists in that order with fid_list_ret_list), fid_offset int64 on the
  device holding the uint64 bit pattern, feature_offset int32 on the device, nfl_offset uint32
  numpy on the host (the layout op reads it there)."""
  if version != 2:
    raise ValueError("only ShardingSparseFidsV2 is implemented (the fused-layout path)")
  ps_num = 1 if ps_num == 0 else int(ps_num)
  plan = build_sharding_plan(feature_cfgs, ps_num)
  names = plan.feature_names
  M = len(names)
  dev = None
  for f in features.values():
    dev = f.values.device
    break
  if dev is None or dev.type != "cuda":
    raise _lib.MonolithHipError("sharding_sparse_fids: inputs must be CUDA tensors")
  unknown = set(features) - set(names)
  if unknown:
    raise ValueError("features not in feature_cfgs: %r" % sorted(unknown))
  batch_size = 0
  for f in features.values():
    if isinstance(f, Ragged):
      batch_size = f.nrows()
      break
  keep, fid_ptrs, rs_ptrs = [], (C.c_void_p * M)(), (C.c_void_p * M)()
  lens, runs = (C.c_int64 * M)(), (C.c_int64 * M)()
  nfl = np.zeros(M, np.uint32)
  total = total_runs = 0
  for m, n in enumerate(names):
    nfl[m] = total_runs
    f = features.get(n)
    if f is None:
      continue
    vals = f.values.to(torch.int64).contiguous()
    if isinstance(f, SharedFeature):
      rs = torch.tensor([0, vals.numel()], dtype=torch.int64, device=dev)
      nfl[m] |= SHARED_FLAG
    else:
      if f.nrows() != batch_size:
        raise ValueError("feature %s has %d rows, batch is %d" % (n, f.nrows(), batch_size))
      rs = f.row_splits.to(device=dev, dtype=torch.int64).contiguous()
    keep += [vals, rs]
    fid_ptrs[m], rs_ptrs[m] = vals.data_ptr() or None, rs.data_ptr()
    lens[m], runs[m] = vals.numel(), rs.numel() - 1
    total += vals.numel()
    total_runs += rs.numel() - 1
  fid_all = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
  fid_offset = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
  feature_offset = torch.empty(max(total_runs, 1), dtype=torch.int32, device=dev)
  counts = np.zeros(plan.n_out, np.int32)
  oi = np.ascontiguousarray(plan.out_index.reshape(-1))
  _lib.check(_lib.lib.mono_sharding_sparse_fids(
      fid_ptrs, rs_ptrs, lens, runs, C.c_int(M), C.c_int(ps_num), C.c_void_p(oi.ctypes.data),
      C.c_int(plan.n_out), C.c_int(int(bool(unique))), _lib.vp(fid_all), _lib.vp(fid_offset),
      _lib.vp(feature_offset), C.c_void_p(counts.ctypes.data), _lib.stream_ptr(dev)))
  # split the grouped FIDs into the per-(table, ps) tensors + per-feature row splits
  fid_list, row_splits = [], []
  pos = 0
  for ti, feats in enumerate(plan.table_features):
    for ps in range(ps_num):
      c = [int(counts[plan.out_index[names.index(n), ps]]) for n in feats]
      size = sum(c)
      fid_list.append(fid_all[pos:pos + size])
      row_splits.append(torch.tensor(np.concatenate([[0], np.cumsum(c)]), dtype=torch.int64))
      pos += size
  fid_offset, feature_offset = fid_offset[:total], feature_offset[:total_runs]
  if fid_list_ret_list:
    return fid_list, fid_offset, feature_offset, nfl, batch_size, row_splits
  keys = [t + ":" + str(ps) for t in plan.table_names for ps in range(ps_num)]
  return (dict(zip(keys, fid_list)), fid_offset, feature_offset, nfl, batch_size,
          dict(zip(keys, row_splits)))
