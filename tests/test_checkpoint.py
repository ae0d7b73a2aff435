"""Reference-format table dumps. The golden files under tests/golden/ckpt are the
dump shards the reference itself ships (monolith/native_training/model_export/testdata/
saved_model/ps_3/1622716114/assets): 4 tables x 5 PS x 4 shards written by MonolithHashTableSave.

CPU: framing (lengths, masked crc32c), EntryDump parsing and byte-exact re-encoding.
GPU: restore into the HIP table, lookup, save again -> the same records."""
import collections
import glob
import os

import numpy as np
import pytest

from monolith_amd import checkpoint as ck
from monolith_amd import entry

HERE = os.path.dirname(os.path.abspath(__file__))
CKPT = os.path.join(HERE, "golden", "ckpt")
# table name hash -> segments [(dim, optimizer kind)] (read off the dumps: the demo model's
# bias = dim-1 FTRL, vectors = SGD)
TABLES = {
    "0b9721ec6fc5396c38499b5be394b722": [(1, ck.FTRL), (32, ck.SGD)],
    "3fc25c64637605aa3983374cc61db982": [(16, ck.SGD)],
    "e3997af6324e55640d4611001fa3a15b": [(1, ck.FTRL), (16, ck.SGD)],
    "f6962510869b682fd764009be0e4e9c3": [(32, ck.SGD)],
}


def _bases():
  out = collections.OrderedDict()
  for f in sorted(glob.glob(os.path.join(CKPT, "MonolithHashTable_*"))):
    base = f[:-len("-00000-of-00004")]
    out.setdefault(base, []).append(f)
  return out


def test_crc32c_known_answers():
  """RFC 3720 B.4 check value and TFRecord's mask."""
  from monolith_amd import _lib
  import ctypes as C
  fn = _lib.lib.mono_crc32c
  fn.restype = C.c_uint32
  assert fn(b"123456789", C.c_int64(9)) == 0xE3069283
  assert fn(bytes(32), C.c_int64(32)) == 0x8A9136AA
  assert fn(bytes([0xFF] * 32), C.c_int64(32)) == 0x62A8AB43


def test_reference_dump_files_parse_and_reencode_byte_exact():
  bases = _bases()
  assert len(bases) == 20 and all(len(v) == 4 for v in bases.values())
  n_records = 0
  for base, files in bases.items():
    h = os.path.basename(base).split("_")[1]
    layout = ck.DumpLayout(TABLES[h])
    for f in files:
      buf, off, ln = ck.read_records(f, verify=True)   # raises on any crc / framing error
      if not len(off):
        assert os.path.getsize(f) == 0
        continue
      ids, rows, ts = layout.decode(buf, off, ln)
      n_records += len(ids)
      assert rows.shape == (len(ids), layout.row_floats) and np.all(ts == 0)
      assert np.all(np.isfinite(rows))
      # the generic (field-by-field) parser agrees with the vectorised template path
      for i in range(len(off)):
        rid, row, t = layout._decode_one(buf[off[i]:off[i] + ln[i]])
        assert rid == ids[i] and np.array_equal(row, rows[i]) and t == ts[i]
      # re-encoding + framing reproduces the file byte for byte
      groups = layout.encode(ids, rows, ts)
      assert len(groups) == 1
      assert np.array_equal(ck.frame_records(groups[0]), buf)
      # FTRL state of the reference's dumps: norm > 0 (initial accumulator + g^2)
      if TABLES[h][0][1] == ck.FTRL:
        assert np.all(rows[:, layout.dim] > 0)
  assert n_records == 30


def test_corruption_is_detected(tmp_path):
  f = sorted(glob.glob(os.path.join(CKPT, "*")), key=os.path.getsize)[-1]
  b = bytearray(open(f, "rb").read())
  b[40] ^= 1
  p = tmp_path / "bad-00000-of-00001"
  p.write_bytes(bytes(b))
  with pytest.raises(Exception):
    ck.read_records(str(p), verify=True)
  p.write_bytes(bytes(b[:-3]))
  with pytest.raises(Exception):
    ck.read_records(str(p), verify=False)


def test_write_and_read_shards_roundtrip(tmp_path):
  """Mixed timestamp widths (varint 1..5 bytes), negative ids, Adagrad + FTRL + SGD segments."""
  layout = ck.DumpLayout([(3, ck.ADAGRAD), (2, ck.FTRL), (4, ck.SGD)])
  assert layout.row_floats == 9 + 3 + 4
  rng = np.random.Generator(np.random.PCG64(3))
  n = 5000
  ids = rng.integers(-2**62, 2**62, n)
  rows = rng.normal(0, 1, (n, layout.row_floats)).astype(np.float32)
  ts = rng.choice([0, 5, 127, 128, 70000, 1_700_000_000, 2**32 - 1], n).astype(np.uint32)
  base = str(tmp_path / "sub" / "MonolithHashTable_x")
  ck.write_shards(base, layout, ids, rows, ts, 4)
  assert sorted(os.listdir(tmp_path / "sub")) == ["MonolithHashTable_x-%05d-of-00004" % i
                                                 for i in range(4)]
  i2, r2, t2 = ck.read_shards(base, layout)
  o1, o2 = np.argsort(ids), np.argsort(i2)
  assert np.array_equal(ids[o1], i2[o2]) and np.array_equal(ts[o1], t2[o2])
  assert np.array_equal(rows[o1].view(np.uint32), r2[o2].view(np.uint32))
  os.remove(base + "-00002-of-00004")
  with pytest.raises(ValueError):
    ck.read_shards(base, layout)


def test_multi_table_dump_roundtrip(tmp_path):
  """Multi-table dumps (ops/multi_hash_table_save_restore_ops.cc): snappy record stream + plain
  metadata records, several shards, a table the reader does not know (skipped by its entry count)
  and one the dump does not hold. The reference ships no such file, so the BYTES are
  parity-unpinned: what is checked is the framing restated from TensorFlow's published snappy
  record format (block = 4-byte big-endian compressed length + raw snappy block, blocks of at
  most 256 KiB of input), the metadata message against its proto definition, and the round trip."""
  raw = os.urandom(1000) + b"abc" * 300000  # three blocks
  z = ck.snappy_stream_compress(raw)
  assert ck.snappy_stream_decompress(z) == raw
  n0 = int.from_bytes(z[:4], "big")
  assert 0 < n0 < ck.SNAPPY_BLOCK and z[4] == 0x80 and z[5] == 0x80 and z[6] == 0x10  # varint 262144
  with pytest.raises(ValueError):
    ck.snappy_stream_decompress(z[:-3])
  assert ck.encode_multi_meta("table_a", 12345).hex() == "0a077461626c655f6110b960"
  assert ck.decode_multi_meta(ck.encode_multi_meta("t", 0)) == ("t", 0)
  la = ck.DumpLayout([(3, ck.ADAGRAD)])
  lb = ck.DumpLayout([(1, ck.FTRL), (4, ck.SGD)])
  rng = np.random.Generator(np.random.PCG64(9))
  def mk(layout, n):
    return (rng.integers(-2**62, 2**62, n), rng.normal(0, 1, (n, layout.row_floats)).astype(np.float32),
            rng.choice([0, 127, 128, 1_700_000_000], n).astype(np.uint32))
  a, b, c = mk(la, 3000), mk(lb, 0), mk(lb, 700)
  base = str(tmp_path / "d" / "MonolithMultiHashTable_x")
  ck.write_multi_shards(base, [("a", la) + a, ("b", lb) + b, ("c", lb) + c], 3)
  assert sorted(os.listdir(tmp_path / "d")) == sorted(
      ["MonolithMultiHashTable_x-%05d-of-00003" % i for i in range(3)] +
      ["MonolithMultiHashTable_x.meta-%05d-of-00003" % i for i in range(3)])
  got = ck.read_multi_shards(base, {"a": la, "c": lb, "zzz": la})
  assert sorted(got) == ["a", "c"]
  for name, (ids, rows, ts) in (("a", a), ("c", c)):
    i2, r2, t2 = got[name]
    o1, o2 = np.argsort(ids), np.argsort(i2)
    assert np.array_equal(ids[o1], i2[o2]) and np.array_equal(ts[o1], t2[o2])
    assert np.array_equal(rows[o1].view(np.uint32), r2[o2].view(np.uint32))
  got = ck.read_multi_shards(base, {"c": lb})  # "a" skipped by count, "b" empty
  assert sorted(got) == ["c"] and len(got["c"][0]) == 700


@pytest.mark.gpu
def test_restore_reference_dumps_into_hip_table_and_save_again(tmp_path):
  import torch
  from monolith_amd import hash_table_ops
  dev = torch.device("cuda", 0)
  for k, (base, files) in enumerate(_bases().items()):
    h = os.path.basename(base).split("_")[1]
    segs = []
    for d, kind in TABLES[h]:
      opt = entry.FtrlOptimizer(initial_accumulator_value=0.1, beta=1.0) if kind == ck.FTRL \
          else entry.SgdOptimizer()
      segs.append(entry.CombineAsSegment(d, entry.ZerosInitializer(), opt))
    tc = entry.EmbeddingHashTableConfig(segments=segs)
    t = hash_table_ops.hash_table_from_config(
        entry.HashTableConfigInstance(tc, [0.01] * len(segs)), name_suffix="ckpt_%d" % k)
    layout = ck.DumpLayout(TABLES[h])
    ids, rows, ts = ck.read_shards(base, layout)
    t = t.restore(base)
    assert t.size() == len(ids)
    if len(ids):
      emb = t.lookup(torch.from_numpy(ids).to(dev)).cpu().numpy()
      assert np.array_equal(emb.view(np.uint32), rows[:, :layout.dim].view(np.uint32))
    out = str(tmp_path / os.path.basename(base))
    t.save(out)
    i2, r2, t2 = ck.read_shards(out, layout)
    o1, o2 = np.argsort(ids), np.argsort(i2)
    assert np.array_equal(ids[o1], i2[o2]) and np.array_equal(ts[o1], t2[o2])
    assert np.array_equal(rows[o1].view(np.uint32), r2[o2].view(np.uint32))


# ------------------------------------------------ the reference's own save / restore scenarios
def _ids_t(x):
  import torch
  return torch.tensor(x, dtype=torch.int64, device="cuda:0")


def _f(x):
  import torch
  return torch.tensor(x, dtype=torch.float32, device="cuda:0")


@pytest.mark.gpu
def test_save_restore(tmp_path):
  """hash_table_ops_test.py:262-277 (negative id included) + :468-475 (missing files raise)."""
  from monolith_amd.hash_table_ops import test_hash_table as make
  t = make(1).assign_add(_ids_t([-1, 1]), _f([[1], [2]]))
  base = str(tmp_path / "test_save_restore" / "table")
  t.save(base)
  t2 = make(1, False).restore(base)
  assert t2.lookup(_ids_t([-1, 1])).cpu().tolist() == [[1.0], [2.0]]
  with pytest.raises(Exception):
    make(1).restore(str(tmp_path / "test_restore_not_found" / "table"))


@pytest.mark.gpu
def test_restore_clears_first(tmp_path):
  """ops/hash_table_restore_op.cc:82: rows present before a restore do not survive it."""
  from monolith_amd.hash_table_ops import test_hash_table as make
  base = str(tmp_path / "table")
  make(1).assign(_ids_t([1]), _f([[1]])).save(base)
  t2 = make(1).assign(_ids_t([1, 7]), _f([[5], [7]]))
  t2 = t2.restore(base)
  assert t2.size() == 1
  assert t2.lookup(_ids_t([1, 7])).cpu().tolist() == [[1.0], [0.0]]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["assign_add", "apply_gradients"])
def test_save_restore_with_feature_eviction(tmp_path, mode):
  """hash_table_ops_test.py:294-329 / :331-379: expiry is applied while dumping, against the
  largest req_time the table has seen."""
  from monolith_amd.hash_table_ops import test_hash_table as make
  expire_time = 1
  t = make(dim_size=1, expire_time=expire_time)
  max_ts = 10000000
  sec = expire_time * 24 * 3600
  for fid, ts in ((1, max_ts - sec - 1), (2, max_ts - sec + 1), (3, max_ts)):
    if mode == "assign_add":
      t = t.assign_add(_ids_t([fid]), _f([[fid]]), ts)
    else:
      t = t.apply_gradients(_ids_t([fid]), _f([[fid]]), 0, req_time=ts)
  base = str(tmp_path / mode / "table")
  t.save(base)
  t2 = make(1, False).restore(base)
  sign = 1.0 if mode == "assign_add" else -1.0
  assert t2.lookup(_ids_t([1, 2, 3])).cpu().tolist() == [[0.0], [sign * 2], [sign * 3]]


@pytest.mark.gpu
def test_entry_ttl(tmp_path):
  """hash_table_ops_test.py:381-411: expire_time 0 drops everything at save, 3600 days keeps."""
  from monolith_amd.hash_table_ops import test_hash_table as make
  for days, expect in ((0, [[0.0], [0.0]]), (60 * 60, [[1.0], [2.0]])):
    base = str(tmp_path / ("ttl_%d" % days) / "table")
    make(1, expire_time=days).assign_add(_ids_t([-1, 1]), _f([[1], [2]])).save(base)
    assert make(1).restore(base).lookup(_ids_t([-1, 1])).cpu().tolist() == expect


@pytest.mark.gpu
def test_entry_ttl_by_slots(tmp_path):
  """hash_table_ops_test.py:413-466: slot 1 expires after 0 days, slot 2 after 1 day."""
  from monolith_amd import hash_filter_ops, hash_table_ops as ops
  tc = entry.EmbeddingHashTableConfig(
      segments=[entry.CombineAsSegment(1, entry.ZerosInitializer(), entry.SgdOptimizer())])
  tc.slot_expire_time_config.default_expire_time = 60 * 60
  tc.slot_expire_time_config.slot_expire_times[1] = 0
  tc.slot_expire_time_config.slot_expire_times[2] = 1
  filters = hash_filter_ops.create_hash_filters(0, False)
  config = entry.HashTableConfigInstance(tc, [1.0])
  id_1, id_2 = 1 << 48, 2 << 48
  base, base_new = str(tmp_path / "table"), str(tmp_path / "table_new")
  t = ops.hash_table_from_config(config, hash_filter=filters[0], name_suffix="ttl_slots_a")
  t.assign_add(_ids_t([id_1, id_2]), _f([[1], [2]]), 100).save(base)
  t2 = ops.hash_table_from_config(config, hash_filter=filters[0], name_suffix="ttl_slots_b")
  t2 = t2.restore(base)
  assert t2.lookup(_ids_t([id_1, id_2])).cpu().tolist() == [[0.0], [2.0]]
  t2.save(base_new)
  t3 = ops.test_hash_table(1).restore(base_new)
  assert t3.lookup(_ids_t([id_1, id_2])).cpu().tolist() == [[0.0], [2.0]]


@pytest.mark.gpu
def test_checkpoint_dir_helpers_and_extra_restore_names(tmp_path):
  """The saver / restorer listeners' file naming (hash_table_ops.py:554-556, :573-590, :645-679):
  <asset_dir><table name>, falling back to extra_restore_names
  (hash_table_ops_test.py:279-292); unknown table -> ValueError."""
  from monolith_amd import hash_table_ops as ops
  asset_dir = str(tmp_path / "model.ckpt-10.assets") + "/"
  t1 = ops.test_hash_table(1, name_suffix="old/name:0").assign(_ids_t([1]), _f([[1]]))
  ops.save_hash_tables(asset_dir, [t1])
  files = sorted(os.listdir(asset_dir))
  assert files and all(f.startswith(t1.name.replace(":", "-").replace("/", "-") + "-0")
                       for f in files)
  t2 = ops.test_hash_table(1, name_suffix="renamed", extra_restore_names=[t1.name])
  ops.restore_hash_tables(asset_dir, [t2])
  assert t2.lookup(_ids_t([1])).cpu().tolist() == [[1.0]]
  t3 = ops.test_hash_table(1, name_suffix="never_saved")
  with pytest.raises(ValueError):
    ops.restore_hash_tables(asset_dir, [t3])


# ----------------------------------------------------------------------------- hash-filter dumps
def test_filter_dump_messages_roundtrip_and_known_bytes():
  """HashFilterSplitMetaDump / HashFilterSplitDataDump wire format
  (embedding_hash_table.proto:108-133). No dump file of a filter ships with the reference, so the
  bytes are pinned to the proto definition only ("parity unpinned" against a reference-written
  file): hand-assembled messages, plus python-protobuf's own encoder when it is importable."""
  sliding = {"split_num": 7, "max_forward_step": 2, "max_backward_step": 5, "max_step": 16,
             "head": 3, "head_increment": 300, "failure_count": 0}
  meta = {"failure_count": 0, "total_size": 60, "num_elements": 2, "fill_rate": 1.2,
          "sliding": sliding}
  b = ck.encode_filter_meta(meta)
  import struct
  sl = bytes([0x08, 7, 0x10, 2, 0x18, 5, 0x20, 16, 0x28, 3, 0x30, 0xAC, 0x02, 0x38, 0])
  assert b == bytes([0x08, 0, 0x10, 60, 0x18, 2, 0x21]) + struct.pack("<d", 1.2) + \
      bytes([0x2A, len(sl)]) + sl
  assert ck.decode_filter_meta(b) == meta
  cells = np.array([0, 1, 127, 128, 0x3FFF, 0x4000, 0xFFFF, (0xABC << 4) | 3], np.uint16)
  d = ck.encode_filter_data(20000, cells)
  assert d == bytes([0x08, 0xA0, 0x9C, 0x01, 0x10, 0, 0x10, 1, 0x10, 127, 0x10, 0x80, 0x01,
                     0x10, 0xFF, 0x7F, 0x10, 0x80, 0x80, 0x01, 0x10, 0xFF, 0xFF, 0x03,
                     0x10, 0xC3, 0xD7, 0x02])
  off, back = ck.decode_filter_data(d)
  assert off == 20000 and np.array_equal(back, cells)
  # a packed encoding of the same message parses too (generic path)
  packed = bytes([0x08, 5, 0x12, 3, 1, 0x80, 0x01])
  assert ck.decode_filter_data(packed)[0] == 5
  assert ck.decode_filter_data(packed)[1].tolist() == [1, 128]
  rng = np.random.Generator(np.random.PCG64(1))
  big = rng.integers(0, 65536, 10000).astype(np.uint16)
  off, back = ck.decode_filter_data(ck.encode_filter_data(0, big))
  assert off == 0 and np.array_equal(back, big)


def test_filter_split_file_roundtrip(tmp_path):
  """One block file: meta record + ceil(cells / 10000) data records (hash_filter.cc:25-56)."""
  rng = np.random.Generator(np.random.PCG64(2))
  cells = np.where(rng.random(25064) < 0.3, rng.integers(1, 65536, 25064), 0).astype(np.uint16)
  meta = {"failure_count": 0, "total_size": 25000, "num_elements": 7000, "fill_rate": 1.2,
          "sliding": {"split_num": 7, "max_forward_step": 2, "max_backward_step": 5, "max_step": 16,
                      "head": 6, "head_increment": 13, "failure_count": 4}}
  p = str(tmp_path / "d" / "hash_filter_0-00003-of-00007")
  ck.write_filter_split(p, meta, cells)
  buf, off, ln = ck.read_records(p)
  assert len(off) == 1 + 3
  m2, c2 = ck.read_filter_split(p)
  assert m2 == meta and np.array_equal(c2, cells)


@pytest.mark.gpu
def test_hash_filter_save_restore(tmp_path):
  """hash_filter_ops_test.py:60-103: threshold 3, ids [1, 3 << 17, 1]; save / restore in the
  middle of the sequence leaves the pass pattern of the four runs unchanged, 7 files per dump."""
  import torch
  from monolith_amd import hash_filter_ops as ops
  cfg = entry.SlotOccurrenceThresholdConfig(default_occurrence_threshold=3)
  f = ops.create_hash_filter(100, 7, cfg)
  ids = _ids_t([1, 3 << 17, 1])

  def run():
    emb = torch.zeros(3, 2, device="cuda:0", requires_grad=True)
    ops.intercept_gradient(f, ids, emb).sum().backward()
    return emb.grad.cpu().tolist()

  base0 = str(tmp_path / "f" / "hash_filter_test_0")
  ops.save_hash_filter(f, base0, True)
  assert len(glob.glob(base0 + "*")) == 7
  ops.restore_hash_filter(f, base0, True)
  assert run() == [[0, 0], [0, 0], [0, 0]]
  assert run() == [[0, 0], [0, 0], [1, 1]]
  base1 = str(tmp_path / "f" / "hash_filter_test_1")
  ops.save_hash_filter(f, base1, True)
  assert len(glob.glob(base1 + "*")) == 7
  # counts move on, then the restore brings checkpoint 1 back
  run()
  ops.restore_hash_filter(f, base1, True)
  assert run() == [[1, 1], [0, 0], [1, 1]]
  assert run() == [[1, 1], [1, 1], [1, 1]]
  # into a fresh filter of the same shape: same continuation
  g = ops.create_hash_filter(100, 7, cfg)
  ops.restore_hash_filter(g, base1, True)
  ops.restore_hash_filter(f, base1, True)
  assert g.get(_ids_t([1, 3 << 17, 5])).cpu().tolist() == [4, 2, 0]
  assert f.get(_ids_t([1, 3 << 17, 5])).cpu().tolist() == [4, 2, 0]
  # shape mismatch is rejected like SlidingHashFilter::ValidateData
  with pytest.raises(ValueError):
    ops.restore_hash_filter(ops.create_hash_filter(100, 9, cfg), base1, True)
  # enable_hash_filter False: no-op both ways
  ops.save_hash_filter(f, str(tmp_path / "none" / "x"), False)
  assert not os.path.exists(str(tmp_path / "none"))
  ops.restore_hash_filter(f, str(tmp_path / "none" / "x"), False)


@pytest.mark.gpu
def test_hash_filter_dump_carries_rotation_state(tmp_path):
  """hash_filter_ops_test.py:105-177 shape (capacity 300, 100 splits -> blocks of total_size 3):
  ids fed one per call so rotation falls where the reference puts it; after two passes the dump's
  meta shows head = head_increment = 4 and num_elements 2 in the first four blocks, and a
  restored filter continues identically."""
  import torch
  from monolith_amd import hash_filter_ops as ops
  cfg = entry.SlotOccurrenceThresholdConfig(default_occurrence_threshold=2)
  f = ops.create_hash_filter(300, 100, cfg)
  base0 = str(tmp_path / "hash_filter_test_0")
  ops.save_hash_filter(f, base0, True)
  files = sorted(glob.glob(base0 + "*"))
  assert len(files) == 100
  for p in files:
    m, cells = ck.read_filter_split(p)
    assert m["total_size"] == 3 and m["num_elements"] == 0 and len(cells) == 67
    assert m["sliding"]["split_num"] == 100 and m["sliding"]["head"] == 0
    assert m["sliding"]["head_increment"] == 0
  seq = [1, 1 << 17, 2 << 17, 1]
  one = torch.ones(1, dtype=torch.int32, device="cuda:0")
  old = [[int(f.add(_ids_t([i]), one)[0]) for i in seq] for _ in range(2)]
  assert old == [[0, 0, 0, 1], [2, 1, 1, 3]]
  base1 = str(tmp_path / "hash_filter_test_1")
  ops.save_hash_filter(f, base1, True)
  files = sorted(glob.glob(base1 + "*"))
  for k, p in enumerate(files):
    m, _ = ck.read_filter_split(p)
    assert m["num_elements"] == (2 if k < 4 else 0)
    assert m["sliding"]["head"] == 4 and m["sliding"]["head_increment"] == 4
  g = ops.create_hash_filter(300, 100, cfg)
  ops.restore_hash_filter(g, base1, True)
  assert g.state() == f.state()
  assert [int(g.add(_ids_t([i]), one)[0]) for i in seq] == \
      [int(f.add(_ids_t([i]), one)[0]) for i in seq]
