"""Cross-check of the hand-written wire encoders / parsers in monolith_amd/checkpoint.py against
the protobuf runtime. The message definitions are restated from the reference's .proto files
(runtime/hash_table/embedding_hash_table.proto:45-50,108-133, optimizer/optimizer.proto:28-30,
56-57,69-72,216-236) as dynamic descriptors — nothing is generated or copied from the reference."""
import glob
import os

import numpy as np
import pytest

pb = pytest.importorskip("google.protobuf")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

from monolith_amd import checkpoint as ck  # noqa: E402

F = descriptor_pb2.FieldDescriptorProto
CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt")


def _msg(fd, name, fields):
  m = fd.message_type.add()
  m.name = name
  for fname, num, ftype, label, tname in fields:
    f = m.field.add()
    f.name, f.number, f.type, f.label = fname, num, ftype, label
    if tname:
      f.type_name = ".t." + tname
  return m


@pytest.fixture(scope="module")
def classes():
  fd = descriptor_pb2.FileDescriptorProto()
  fd.name, fd.package, fd.syntax = "mono_ckpt_test.proto", "t", "proto2"
  O, R = F.LABEL_OPTIONAL, F.LABEL_REPEATED
  _msg(fd, "AdagradOptimizerDump", [("norm", 1, F.TYPE_FLOAT, R, None)])
  _msg(fd, "SgdOptimizerDump", [])
  _msg(fd, "FtrlOptimizerDump", [("zero", 1, F.TYPE_FLOAT, R, None),
                                 ("norm", 2, F.TYPE_FLOAT, R, None)])
  _msg(fd, "AdadeltaOptimizerDump", [("accum", 1, F.TYPE_FLOAT, R, None),
                                     ("accum_update", 2, F.TYPE_FLOAT, R, None)])
  _msg(fd, "AdamOptimizerDump", [("m", 1, F.TYPE_FLOAT, R, None), ("v", 2, F.TYPE_FLOAT, R, None),
                                 ("beta1_power", 3, F.TYPE_FLOAT, O, None),
                                 ("beta2_power", 4, F.TYPE_FLOAT, O, None)])
  _msg(fd, "AmsgradOptimizerDump", [("m", 1, F.TYPE_FLOAT, R, None), ("v", 2, F.TYPE_FLOAT, R, None),
                                    ("vhat", 3, F.TYPE_FLOAT, R, None),
                                    ("beta1_power", 4, F.TYPE_FLOAT, O, None),
                                    ("beta2_power", 5, F.TYPE_FLOAT, O, None)])
  for n in ("MomentumOptimizerDump", "RmspropOptimizerDump", "RmspropV2OptimizerDump"):
    _msg(fd, n, [("n", 1, F.TYPE_FLOAT, R, None)])
  _msg(fd, "BatchSoftmaxOptimizerDump", [("global_step", 1, F.TYPE_INT64, O, None)])
  _msg(fd, "SingleOptimizerDump", [("adagrad", 1, F.TYPE_MESSAGE, O, "AdagradOptimizerDump"),
                                   ("sgd", 2, F.TYPE_MESSAGE, O, "SgdOptimizerDump"),
                                   ("ftrl", 3, F.TYPE_MESSAGE, O, "FtrlOptimizerDump"),
                                   ("adadelta", 6, F.TYPE_MESSAGE, O, "AdadeltaOptimizerDump"),
                                   ("adam", 7, F.TYPE_MESSAGE, O, "AdamOptimizerDump"),
                                   ("amsgrad", 8, F.TYPE_MESSAGE, O, "AmsgradOptimizerDump"),
                                   ("momentum", 9, F.TYPE_MESSAGE, O, "MomentumOptimizerDump"),
                                   ("rmsprop", 11, F.TYPE_MESSAGE, O, "RmspropOptimizerDump"),
                                   ("rmspropv2", 12, F.TYPE_MESSAGE, O, "RmspropV2OptimizerDump"),
                                   ("batch_softmax", 14, F.TYPE_MESSAGE, O,
                                    "BatchSoftmaxOptimizerDump")])
  _msg(fd, "OptimizerDump", [("dump", 1, F.TYPE_MESSAGE, R, "SingleOptimizerDump")])
  _msg(fd, "EntryDump", [("id", 1, F.TYPE_SFIXED64, O, None), ("num", 2, F.TYPE_FLOAT, R, None),
                         ("opt", 3, F.TYPE_MESSAGE, O, "OptimizerDump"),
                         ("last_update_ts_sec", 4, F.TYPE_INT64, O, None)])
  _msg(fd, "SlidingHashFilterMetaDump",
       [(n, i + 1, F.TYPE_UINT64 if n == "failure_count" else F.TYPE_UINT32, O, None)
        for i, n in enumerate(["split_num", "max_forward_step", "max_backward_step", "max_step",
                               "head", "head_increment", "failure_count"])])
  _msg(fd, "HashFilterSplitMetaDump",
       [("failure_count", 1, F.TYPE_UINT64, O, None), ("total_size", 2, F.TYPE_UINT64, O, None),
        ("num_elements", 3, F.TYPE_UINT64, O, None), ("fill_rate", 4, F.TYPE_DOUBLE, O, None),
        ("sliding_hash_filter_meta", 5, F.TYPE_MESSAGE, O, "SlidingHashFilterMetaDump")])
  _msg(fd, "HashFilterSplitDataDump", [("offset", 1, F.TYPE_UINT32, O, None),
                                       ("data", 2, F.TYPE_UINT32, R, None)])
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("t." + n))
          for n in ("EntryDump", "HashFilterSplitMetaDump", "HashFilterSplitDataDump")}


def _entry_msg(cls, layout, rid, row, ts):
  m = cls["EntryDump"]()
  m.id = int(rid)
  m.num.extend(float(x) for x in row[:layout.dim])
  cur = layout.dim
  m.opt.SetInParent()
  for d, kind in layout.segments:
    if kind == ck.MOVING_AVERAGE:  # Save() returns no entry (moving_average_optimizer.cc:49-52)
      continue
    s = m.opt.dump.add()
    if kind == ck.BATCH_SOFTMAX:
      s.batch_softmax.global_step = int(row[cur:cur + 2].view(np.int64)[0])
      cur += 2
    elif kind == ck.ADAGRAD:
      s.adagrad.norm.extend(float(x) for x in row[cur:cur + d])
      cur += d
    elif kind == ck.FTRL:  # row state order norm | zero (ftrl_optimizer.cc:38-45)
      s.ftrl.zero.extend(float(x) for x in row[cur + d:cur + 2 * d])
      s.ftrl.norm.extend(float(x) for x in row[cur:cur + d])
      cur += 2 * d
    elif kind == ck.ADADELTA:
      s.adadelta.accum.extend(float(x) for x in row[cur:cur + d])
      s.adadelta.accum_update.extend(float(x) for x in row[cur + d:cur + 2 * d])
      cur += 2 * d
    elif kind in (ck.ADAM, ck.AMSGRAD):
      sub = s.adam if kind == ck.ADAM else s.amsgrad
      sub.m.extend(float(x) for x in row[cur:cur + d])
      sub.v.extend(float(x) for x in row[cur + d:cur + 2 * d])
      cur += 2 * d
      if kind == ck.AMSGRAD:
        sub.vhat.extend(float(x) for x in row[cur:cur + d])
        cur += d
      sub.beta1_power, sub.beta2_power = float(row[cur]), float(row[cur + 1])
      cur += 2
    elif kind in (ck.MOMENTUM, ck.RMSPROP, ck.RMSPROP_V2):
      sub = {ck.MOMENTUM: s.momentum, ck.RMSPROP: s.rmsprop, ck.RMSPROP_V2: s.rmspropv2}[kind]
      sub.n.extend(float(x) for x in row[cur:cur + d])
      cur += d
    else:
      s.sgd.SetInParent()
  m.last_update_ts_sec = int(ts)
  return m


def test_entry_dump_encoder_equals_protobuf(classes):
  layout = ck.DumpLayout([(3, ck.ADAGRAD), (2, ck.FTRL), (4, ck.SGD), (1, ck.ADAGRAD),
                          (2, ck.ADADELTA), (3, ck.MOMENTUM), (1, ck.RMSPROP), (2, ck.RMSPROP_V2),
                          (2, ck.ADAM), (3, ck.AMSGRAD)])
  rng = np.random.Generator(np.random.PCG64(11))
  n = 64
  ids = rng.integers(-2**62, 2**62, n)
  rows = rng.normal(0, 1, (n, layout.row_floats)).astype(np.float32)
  ts = rng.choice([0, 1, 127, 128, 1 << 20, 1_700_000_000, 2**32 - 1], n).astype(np.uint32)
  mine = {}
  for grp in layout.encode(ids, rows, ts):
    for rec in grp:
      m = classes["EntryDump"]()
      m.ParseFromString(rec.tobytes())
      mine[m.id] = rec.tobytes()
  assert len(mine) == n
  for i in range(n):
    assert _entry_msg(classes, layout, ids[i], rows[i], ts[i]).SerializeToString() == mine[int(ids[i])]


def test_varint_state_and_dumpless_segments_equal_protobuf(classes):
  """Batch softmax keeps an int64 (varint field, negative values as ten bytes); a moving-average
  segment adds no SingleOptimizerDump: records are built one by one and must still be byte-equal."""
  layout = ck.DumpLayout([(1, ck.BATCH_SOFTMAX), (2, ck.MOVING_AVERAGE), (2, ck.ADAGRAD),
                          (1, ck.BATCH_SOFTMAX)])
  assert layout.irregular and layout.row_floats == 6 + 2 + 2 + 2
  rng = np.random.Generator(np.random.PCG64(12))
  n = 40
  ids = rng.integers(-2**62, 2**62, n)
  rows = rng.normal(0, 1, (n, layout.row_floats)).astype(np.float32)
  steps = rng.choice([0, 1, 127, 128, 300, 2**31, 2**40 + 5, -1, -2**40], (n, 2)).astype(np.int64)
  rows[:, 6:8] = steps[:, :1].copy().view(np.float32)
  rows[:, 10:12] = steps[:, 1:].copy().view(np.float32)
  ts = rng.choice([0, 5, 128, 1_700_000_000], n).astype(np.uint32)
  recs = [r.tobytes() for grp in layout.encode(ids, rows, ts) for r in grp]
  assert len(recs) == n
  by_id = {}
  for r in recs:
    m = classes["EntryDump"]()
    m.ParseFromString(r)
    by_id[m.id] = r
  for i in range(n):
    assert _entry_msg(classes, layout, ids[i], rows[i], ts[i]).SerializeToString() == by_id[int(ids[i])]
  buf = np.frombuffer(b"".join(recs), np.uint8)
  ln = np.array([len(r) for r in recs])
  off = np.concatenate([[0], np.cumsum(ln)[:-1]])
  ids2, rows2, ts2 = layout.decode(buf, off, ln)
  order = np.argsort(ids2)
  o0 = np.argsort(ids)
  assert np.array_equal(ids2[order], ids[o0]) and np.array_equal(ts2[order], ts[o0])
  assert np.array_equal(rows2[order].view(np.uint32), rows[o0].view(np.uint32))


def test_reference_dumps_parse_identically_with_protobuf(classes):
  from tests.test_checkpoint import TABLES
  seen = 0
  for f in sorted(glob.glob(os.path.join(CKPT, "MonolithHashTable_*"))):
    buf, off, ln = ck.read_records(f)
    if not len(off):
      continue
    layout = ck.DumpLayout(TABLES[os.path.basename(f).split("_")[1].split("-")[0]])
    ids, rows, ts = layout.decode(buf, off, ln)
    for i in range(len(off)):
      raw = buf[off[i]:off[i] + ln[i]].tobytes()
      m = classes["EntryDump"]()
      m.ParseFromString(raw)
      assert m.id == ids[i] and m.last_update_ts_sec == ts[i]
      assert _entry_msg(classes, layout, ids[i], rows[i], ts[i]).SerializeToString() == raw
      seen += 1
  assert seen == 30


def test_filter_messages_equal_protobuf(classes):
  sliding = {"split_num": 100, "max_forward_step": 2, "max_backward_step": 98, "max_step": 16,
             "head": 4, "head_increment": 70000, "failure_count": 1 << 40}
  meta = {"failure_count": 0, "total_size": 51428571, "num_elements": 123456, "fill_rate": 1.2,
          "sliding": sliding}
  m = classes["HashFilterSplitMetaDump"]()
  m.failure_count, m.total_size, m.num_elements, m.fill_rate = 0, 51428571, 123456, 1.2
  for k, v in sliding.items():
    setattr(m.sliding_hash_filter_meta, k, v)
  assert m.SerializeToString() == ck.encode_filter_meta(meta)
  assert ck.decode_filter_meta(m.SerializeToString()) == meta
  rng = np.random.Generator(np.random.PCG64(5))
  cells = rng.integers(0, 65536, 3000).astype(np.uint16)
  d = classes["HashFilterSplitDataDump"]()
  d.offset = 30000
  d.data.extend(int(x) for x in cells)
  assert d.SerializeToString() == ck.encode_filter_data(30000, cells)
  off, back = ck.decode_filter_data(d.SerializeToString())
  assert off == 30000 and np.array_equal(back, cells)
