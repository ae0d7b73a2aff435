"""FeatureConfigs — the layout description of the fused-layout path
(idl/matrix/proto/example.proto: FeatureConfigs / FeatureConfig / OutConfig / SliceConfig /
TensorShape, used by distribution_ops.fused_embedding_to_layout), as plain dataclasses, plus the
flattening the reference op does in its constructor (ops/fused_embedding_to_layout.cc:246-334) and
in its Layout classes (:67-242): `build_layout_plan`.
"""
import ctypes as C
import dataclasses
from typing import Dict, List


class PoolingType:  # example.proto PoolingType
  SUM = 0
  MEAN = 1
  FIRSTN = 2


class OutType:  # example.proto OutType
  CONCAT = 0
  STACK = 1
  ADDN = 2
  NONE = 3


@dataclasses.dataclass
class FeatureConfig:
  table: str = ""
  slice_dims: List[int] = dataclasses.field(default_factory=list)
  pooling_type: int = PoolingType.SUM
  max_sequence_length: int = 0


@dataclasses.dataclass
class SliceConfig:
  feature_name: str = ""
  start: int = 0
  end: int = 0


@dataclasses.dataclass
class OutConfig:
  slice_configs: List[SliceConfig] = dataclasses.field(default_factory=list)
  out_type: int = OutType.CONCAT
  shape: List[List[int]] = dataclasses.field(default_factory=list)  # TensorShape.dims, -1 = batch


@dataclasses.dataclass
class FeatureConfigs:
  feature_configs: Dict[str, FeatureConfig] = dataclasses.field(default_factory=dict)
  out_configs: Dict[str, OutConfig] = dataclasses.field(default_factory=dict)


def infer_shape(out_conf: OutConfig, out_type: int, max_sequence_length: int = 0):
  """Shape rules the reference's callers use (fused_embedding_to_layout_test.py:36-83)."""
  out_conf.out_type = out_type
  out_conf.shape = []
  dims = [sc.end - sc.start for sc in out_conf.slice_configs]
  if out_type == OutType.NONE:
    for d in dims:
      out_conf.shape.append([-1, max_sequence_length, d] if max_sequence_length > 0 else [-1, d])
  elif out_type == OutType.CONCAT:
    out_conf.shape.append([-1, max_sequence_length, sum(dims)] if max_sequence_length > 0
                          else [-1, sum(dims)])
  elif out_type == OutType.STACK:
    assert len(set(dims)) == 1
    out_conf.shape.append([-1, len(dims), max_sequence_length, dims[0]] if max_sequence_length > 0
                          else [-1, len(dims), dims[0]])
  elif out_type == OutType.ADDN:
    assert len(set(dims)) == 1
    out_conf.shape.append([-1, max_sequence_length, dims[0]] if max_sequence_length > 0
                          else [-1, dims[0]])
  else:
    raise ValueError("out_type error")


class LayoutSlice(C.Structure):  # mono_layout_slice (include/monolith_hip.h)
  _fields_ = [("feature_idx", C.c_int32), ("start", C.c_int32), ("dim", C.c_int32),
              ("pooling", C.c_int32), ("max_seq", C.c_int32), ("out_index", C.c_int32),
              ("out_stride", C.c_int32), ("out_offset", C.c_int32), ("addn", C.c_int32),
              ("first", C.c_int32)]


@dataclasses.dataclass
class LayoutPlan:
  slices: List[LayoutSlice]
  slice_idx_of: List[int]          # index of every slice inside its feature
  max_slice_num: int
  out_shapes: List[List[int]]      # one per output tensor, dim 0 == -1 -> batch
  feature_names: List[str]         # sorted: position = index into nfl_offset
  table_feature_dims: List[List[int]]  # tables by sorted name -> dims of their sorted features


def _prod(xs):
  r = 1
  for x in xs:
    r *= x
  return r


def build_layout_plan(cfgs: FeatureConfigs) -> LayoutPlan:
  """Flattens FeatureConfigs the way the reference op does.

  * feature index = position among the sorted feature names (:261-275,323-330)
  * V2 embedding buffers: tables by sorted name, features by sorted name inside (:276-296)
  * layouts by sorted name (:333); NONE layouts own one tensor per slice (:92-116), the others
    one tensor, slices at offset += 1 (STACK) / += dim (CONCAT) / 0 (ADDN) (:147-191)
  * per-sample address of a slice = GetSlice(row 0) + row * PtrWrapper::offset (:120-237)."""
  fcs = cfgs.feature_configs
  feature_names = sorted(fcs.keys())
  tables: Dict[str, Dict[str, int]] = {}
  for name, fc in fcs.items():
    tables.setdefault(fc.table, {})[name] = sum(fc.slice_dims)
  table_feature_dims = [[tables[t][f] for f in sorted(tables[t])] for t in sorted(tables)]
  start2slice = []
  max_slice_num = 0
  for name in feature_names:
    m, pre = {}, 0
    for i, d in enumerate(fcs[name].slice_dims):
      m[pre] = i
      pre += d
    start2slice.append(m)
    max_slice_num = max(max_slice_num, len(fcs[name].slice_dims))

  slices, slice_idx_of, out_shapes = [], [], []
  for lname in sorted(cfgs.out_configs.keys()):
    oc = cfgs.out_configs[lname]
    if oc.out_type == OutType.NONE:
      if len(oc.slice_configs) != len(oc.shape):
        raise ValueError("NONE layout %s: one shape per slice expected" % lname)
    elif len(oc.shape) != 1:
      raise ValueError("layout %s: exactly one shape expected" % lname)
    offset = 0
    for i, sc in enumerate(oc.slice_configs):
      fc = fcs[sc.feature_name]
      fidx = feature_names.index(sc.feature_name)
      dim = sc.end - sc.start
      firstn = fc.pooling_type == PoolingType.FIRSTN
      if oc.out_type == OutType.NONE:
        shape = oc.shape[i]
        out_index = len(out_shapes)
        out_shapes.append(list(shape))
        if firstn:
          if len(shape) != 3:
            raise ValueError("NONE/FIRSTN slice needs a rank-3 shape")
          stride = shape[1] * shape[2]
        else:
          if len(shape) != 2:
            raise ValueError("NONE slice needs a rank-2 shape")
          stride = shape[1]
        out_offset = 0
      else:
        shape = oc.shape[0]
        out_index = len(out_shapes)  # appended after the loop
        if firstn:
          if oc.out_type == OutType.ADDN:
            raise ValueError("ADDN over FIRSTN slices is not supported (the reference overruns "
                             "its temporary there)")
          if len(shape) == 3:    # &tensor(row, 0, offset)
            stride, out_offset = shape[1] * shape[2], offset
          elif len(shape) == 4:  # &tensor(row, 0, 0, offset)
            stride, out_offset = shape[1] * shape[2] * shape[3], offset
          else:
            raise ValueError("FIRSTN layout needs rank 3 or 4")
        else:
          if len(shape) == 2:    # &mat(row, offset)
            stride, out_offset = shape[1], offset
          elif len(shape) == 3:  # &tensor(row, offset, 0)
            stride, out_offset = shape[1] * shape[2], offset * shape[2]
          else:
            raise ValueError("layout needs rank 2 or 3")
        if oc.out_type == OutType.STACK:
          offset += 1
        elif oc.out_type == OutType.CONCAT:
          offset += dim
      slices.append(LayoutSlice(fidx, sc.start, dim, fc.pooling_type, fc.max_sequence_length,
                                out_index, stride, out_offset,
                                1 if oc.out_type == OutType.ADDN else 0, 1 if i == 0 else 0))
      slice_idx_of.append(start2slice[fidx].get(sc.start, 0))
    if oc.out_type != OutType.NONE:
      out_shapes.append(list(oc.shape[0]))
  return LayoutPlan(slices, slice_idx_of, max_slice_num, out_shapes, feature_names,
                    table_feature_dims)


def split_embeddings_v2(plan: LayoutPlan, embeddings_list, fid_list_row_split, ps_num: int):
  """V2 input convention (:438-476): embeddings_list[table_i * ps_num + ps_i] is one flat buffer
  holding, back to back, the rows of every feature of that table (sorted by name);
  fid_list_row_split gives the per-feature row counts. Returns [(tensor, byte_offset_in_floats,
  dim, floats)] in fid_offset's buffer-index order."""
  out = []
  for table_i, feature_dims in enumerate(plan.table_feature_dims):
    for ps_i in range(ps_num):
      emb_index = table_i * ps_num + ps_i
      split = [int(x) for x in fid_list_row_split[emb_index].tolist()]
      if len(split) != len(feature_dims) + 1:
        raise ValueError("fid_list_row_split[%d] has %d entries, expected %d" %
                         (emb_index, len(split), len(feature_dims) + 1))
      pre_offset, pre_emb = 0, 0
      total = int(embeddings_list[emb_index].numel())
      for feature_i, dim in enumerate(feature_dims):
        offset = split[feature_i + 1]
        cnt = offset - pre_offset
        out.append((emb_index, pre_emb, dim, cnt * dim))
        pre_offset = offset
        pre_emb += cnt * dim
        if pre_emb > total:
          raise ValueError("fid_list_row_split[%d] overruns its embedding buffer" % emb_index)
  return out
