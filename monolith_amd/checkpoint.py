"""Reference-format table dumps: read / write the files MonolithHashTableSave / Restore produce
(ops/hash_table_save_op.cc:95-165, ops/hash_table_restore_op.cc:60-150).

  file      <basename>-%05d-of-%05d (save_utils / GetShardedFileName), TFRecord framing
  record    serialized EntryDump (runtime/hash_table/embedding_hash_table.proto:45-50):
              1: sfixed64 id            2: repeated float num (NOT packed: one 0x15 tag per value)
              3: OptimizerDump opt      4: int64 last_update_ts_sec (varint)
            OptimizerDump.dump[i] = SingleOptimizerDump, one per segment
            (optimizer.proto:216-236): adagrad = 1 {norm = 1}, sgd = 2 {}, ftrl = 3 {zero = 1, norm = 2}

protobuf is not used: the wire format of these four messages is written / parsed directly, so that
whole tables convert with numpy (records of one table share one byte layout per timestamp-varint
length). Pinned byte for byte by the dump files the reference ships
(tests/golden/ckpt/, from model_export/testdata/saved_model/ps_3/.../assets)."""
import ctypes as C
import glob
import os
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from monolith_amd import _lib, entry

SGD, ADAGRAD, FTRL = entry.OPT_SGD, entry.OPT_ADAGRAD, entry.OPT_FTRL
MOMENTUM, RMSPROP, RMSPROP_V2, ADADELTA = (entry.OPT_MOMENTUM, entry.OPT_RMSPROP,
                                           entry.OPT_RMSPROP_V2, entry.OPT_ADADELTA)
ADAM, AMSGRAD = entry.OPT_ADAM, entry.OPT_AMSGRAD
MOVING_AVERAGE = entry.OPT_MOVING_AVERAGE
BATCH_SOFTMAX = entry.OPT_BATCH_SOFTMAX
# SingleOptimizerDump oneof field (optimizer.proto:216-231) and, per optimizer, the dump's float
# fields in SERIALIZATION order: (field number, state slot of the row | "scalar k").
_DUMP_FIELDS = {
    SGD: (2, []),
    ADAGRAD: (1, [(1, 0)]),            # norm
    FTRL: (3, [(1, 1), (2, 0)]),       # zero = 1 <- slot 1, norm = 2 <- slot 0
    ADADELTA: (6, [(1, 0), (2, 1)]),   # accum, accum_update
    MOMENTUM: (9, [(1, 0)]),           # n
    RMSPROP: (11, [(1, 0)]),           # n
    RMSPROP_V2: (12, [(1, 0)]),        # n
    # m, v (, vhat) per weight, then the row's beta1^t, beta2^t (optional float fields)
    ADAM: (7, [(1, 0), (2, 1), (3, "s0"), (4, "s1")]),
    AMSGRAD: (8, [(1, 0), (2, 1), (3, 2), (4, "s0"), (5, "s1")]),
    # moving_average_optimizer.cc:49-52: Save() returns an OptimizerDump without entries, so the
    # segment contributes NO SingleOptimizerDump (optimizer_combination.cc:65-75 concatenates)
    MOVING_AVERAGE: (None, []),
    # BatchSoftmaxOptimizerDump = 14 {optional int64 global_step = 1}: a varint, see _INT64_PART
    BATCH_SOFTMAX: (14, []),
}
# optimizers whose state is one int64 (two state floats, low word first) dumped as a varint field:
# records of such tables have no fixed layout and are converted one by one
_INT64_PART = {BATCH_SOFTMAX: 1}



def _dump_parts(kind: int, d: int):
  """(oneof field, [(field number, first row offset inside the segment's state, count)],
  state floats of the segment)."""
  field, parts = _DUMP_FIELDS[kind]
  nslots = sum(1 for _, w in parts if not isinstance(w, str))
  out = []
  for fno, where in parts:
    if isinstance(where, str):
      out.append((fno, nslots * d + int(where[1:]), 1))
    else:
      out.append((fno, where * d, d))
  return field, out, sum(c for _, _, c in out) + (2 if kind in _INT64_PART else 0)


def sharded_name(basename: str, idx: int, total: int) -> str:
  return "%s-%05d-of-%05d" % (basename, idx, total)


# ----------------------------------------------------------------------------- framing
def masked_crc32c(data: bytes) -> int:
  fn = _lib.lib.mono_crc32c
  fn.restype = C.c_uint32
  c = fn(data, C.c_int64(len(data)))
  return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def read_records(path: str, verify: bool = True) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
  """(file image uint8, payload offsets, payload lengths); raises on a bad frame / crc."""
  return scan_records(np.fromfile(path, dtype=np.uint8), verify)


def scan_records(buf: np.ndarray, verify: bool = True) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
  """The same over a record stream already in memory (uint8 array)."""
  buf = np.ascontiguousarray(buf, dtype=np.uint8)
  if buf.size == 0:
    return buf, np.zeros(0, np.int64), np.zeros(0, np.int64)
  fn = _lib.lib.mono_tfrecord_scan
  fn.restype = C.c_int64
  p = C.c_void_p(buf.ctypes.data)
  n = fn(p, C.c_int64(buf.size), None, None, C.c_int64(0), C.c_int(int(verify)))
  if n < 0:
    _lib.check(1)
  off = np.empty(n, np.int64)
  ln = np.empty(n, np.int64)
  fn(p, C.c_int64(buf.size), C.c_void_p(off.ctypes.data), C.c_void_p(ln.ctypes.data), C.c_int64(n),
     C.c_int(0))
  return buf, off, ln


def frame_records(payloads: np.ndarray) -> np.ndarray:
  """payloads uint8[n, L] -> framed bytes uint8[n * (L + 16)]."""
  payloads = np.ascontiguousarray(payloads, dtype=np.uint8)
  n, L = payloads.shape
  out = np.empty(n * (L + 16), np.uint8)
  fn = _lib.lib.mono_tfrecord_frame
  fn.restype = C.c_int64
  w = fn(C.c_void_p(payloads.ctypes.data), C.c_int64(n), C.c_int64(L), C.c_void_p(out.ctypes.data))
  assert w == out.size
  return out


# ----------------------------------------------------------------------------- EntryDump
def _varint(v: int) -> bytes:
  v &= (1 << 64) - 1
  out = bytearray()
  while True:
    b = v & 0x7F
    v >>= 7
    if v:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _read_varint(b, pos):
  v, shift = 0, 0
  while True:
    x = int(b[pos])
    pos += 1
    v |= (x & 0x7F) << shift
    if not x & 0x80:
      return v, pos
    shift += 7


def _opt_payload_len(kind: int, dim: int) -> int:
  return 5 * _dump_parts(kind, dim)[2]


def _len_prefixed(tag: int, n: int) -> bytes:
  return bytes([tag]) + _varint(n)


class DumpLayout:
  """Byte layout of the EntryDump records of one table (segments = [(dim, opt kind)])."""

  def __init__(self, segments: Sequence[Tuple[int, int]]):
    self.segments = [(int(d), int(k)) for d, k in segments]
    self.dim = sum(d for d, _ in self.segments)
    self.state = sum(_opt_payload_len(k, d) // 5 for d, k in self.segments)
    self.row_floats = self.dim + self.state
    self.irregular = any(k in _INT64_PART for _, k in self.segments)

  @staticmethod
  def from_table_config(tc: entry.EmbeddingHashTableConfig) -> "DumpLayout":
    return DumpLayout([(s.dim_size, s.opt_config.kind) for s in tc.segments])

  # -- template of one record with a ts varint of `ts_len` bytes
  def _template(self, ts_len: int):
    """(bytes template, offsets of id / floats (row order) / ts)."""
    t = bytearray()
    t += b"\x09" + b"\x00" * 8
    id_off = 1
    float_offs = [0] * self.row_floats
    for i in range(self.dim):
      t += b"\x15"
      float_offs[i] = len(t)
      t += b"\x00" * 4
    # OptimizerDump
    singles = []
    st = self.dim  # row index of the next state float
    for d, k in self.segments:
      body = bytearray()
      marks = []
      field, parts, nfloats = _dump_parts(k, d)
      if field is None:
        continue
      for fno, off, cnt in parts:  # ascending field numbers = serialization order
        for i in range(cnt):
          body.append((fno << 3) | 5)
          marks.append((len(body), st + off + i))
          body += b"\x00" * 4
      st += nfloats
      tag = (field << 3) | 2
      single = _len_prefixed(tag, len(body)) + bytes(body)
      hdr = len(_len_prefixed(tag, len(body)))
      singles.append((single, hdr, marks))
    opt = bytearray()
    opt_marks = []
    for single, hdr, marks in singles:
      pre = _len_prefixed(0x0a, len(single))
      base = len(opt) + len(pre) + hdr
      for off, row_i in marks:
        opt_marks.append((base + off, row_i))
      opt += pre + single
    pre = _len_prefixed(0x1a, len(opt))
    base = len(t) + len(pre)
    for off, row_i in opt_marks:
      float_offs[row_i] = base + off
    t += pre + opt
    t += b"\x20"
    ts_off = len(t)
    t += b"\x00" * ts_len
    return bytes(t), id_off, np.array(float_offs, np.int64), ts_off

  def encode(self, ids: np.ndarray, rows: np.ndarray, ts: np.ndarray) -> List[np.ndarray]:
    """Serialized records, grouped by record length: list of uint8[n_g, L_g] (input order kept
    inside a group)."""
    ids = np.asarray(ids, np.int64)
    rows = np.ascontiguousarray(rows, np.float32).reshape(len(ids), self.row_floats)
    ts = np.asarray(ts).astype(np.uint64)
    if self.irregular:
      recs = [self._encode_one(int(ids[i]), rows[i], int(ts[i])) for i in range(len(ids))]
      lens = np.array([len(r) for r in recs], np.int64)
      return [np.frombuffer(b"".join(recs[i] for i in np.nonzero(lens == L)[0]), np.uint8)
              .reshape(-1, int(L)) for L in np.unique(lens)]
    vlen = np.ones(len(ids), np.int64)
    for k in range(1, 10):
      vlen += (ts >= (1 << (7 * k))).astype(np.int64)
    out = []
    for L in np.unique(vlen):
      sel = np.nonzero(vlen == L)[0]
      tmpl, id_off, f_offs, ts_off = self._template(int(L))
      rec = np.tile(np.frombuffer(tmpl, np.uint8), (len(sel), 1))
      rec[:, id_off:id_off + 8] = ids[sel].view(np.uint8).reshape(-1, 8)
      fb = rows[sel].view(np.uint8).reshape(len(sel), self.row_floats, 4)
      for j in range(4):
        rec[:, f_offs + j] = fb[:, :, j]
      t = ts[sel].copy()
      for j in range(int(L)):
        b = (t & 0x7F).astype(np.uint8)
        t >>= np.uint64(7)
        if j < int(L) - 1:
          b |= 0x80
        rec[:, ts_off + j] = b
      out.append(rec)
    return out

  def decode(self, buf: np.ndarray, off: np.ndarray, ln: np.ndarray):
    """(ids int64[n], rows float32[n, row_floats], ts uint32[n]) of the records; records whose
    bytes follow this table's template are converted with numpy, anything else (missing ts,
    reordered or unknown fields) record by record."""
    n = len(off)
    ids = np.empty(n, np.int64)
    rows = np.zeros((n, self.row_floats), np.float32)
    ts = np.zeros(n, np.uint32)
    done = np.zeros(n, bool)
    for ts_len in (() if self.irregular else range(1, 6)):
      tmpl, id_off, f_offs, ts_off = self._template(ts_len)
      sel = np.nonzero((ln == len(tmpl)) & ~done)[0]
      if not len(sel):
        continue
      rec = buf[off[sel][:, None] + np.arange(len(tmpl))[None, :]]
      mask = np.ones(len(tmpl), bool)
      mask[id_off:id_off + 8] = False
      for j in range(4):
        mask[f_offs + j] = False
      mask[ts_off:ts_off + ts_len] = False
      ok = (rec[:, mask] == np.frombuffer(tmpl, np.uint8)[mask]).all(axis=1)
      tb = rec[:, ts_off:ts_off + ts_len]
      ok &= (tb[:, :-1] & 0x80).all(axis=1) & ((tb[:, -1] & 0x80) == 0)
      good = sel[ok]
      rec = rec[ok]
      ids[good] = rec[:, id_off:id_off + 8].copy().view(np.int64).reshape(-1)
      fb = np.empty((len(good), self.row_floats, 4), np.uint8)
      for j in range(4):
        fb[:, :, j] = rec[:, f_offs + j]
      rows[good] = fb.view(np.float32).reshape(len(good), self.row_floats)
      t = np.zeros(len(good), np.uint64)
      for j in range(ts_len):
        t |= (rec[:, ts_off + j] & 0x7F).astype(np.uint64) << np.uint64(7 * j)
      ts[good] = t.astype(np.uint32)
      done[good] = True
    for i in np.nonzero(~done)[0]:
      ids[i], rows[i], ts[i] = self._decode_one(buf[off[i]:off[i] + ln[i]])
    return ids, rows, ts

  def _encode_one(self, rid: int, row: np.ndarray, ts: int) -> bytes:
    """One EntryDump, field by field (tables with a varint optimizer field)."""
    import struct
    row = np.ascontiguousarray(row, np.float32)
    b = b"\x09" + struct.pack("<q", rid)
    b += b"".join(b"\x15" + row[i:i + 1].tobytes() for i in range(self.dim))
    opt = b""
    st = self.dim
    for d, k in self.segments:
      field, parts, nfloats = _dump_parts(k, d)
      if field is not None:
        body = b""
        if k in _INT64_PART:
          a = int(row[st:st + 2].view(np.int64)[0])
          body += bytes([(_INT64_PART[k] << 3) | 0]) + _varint(a & 0xFFFFFFFFFFFFFFFF)
        for fno, off, cnt in parts:
          body += b"".join(bytes([(fno << 3) | 5]) + row[st + off + i:st + off + i + 1].tobytes()
                           for i in range(cnt))
        single = _len_prefixed((field << 3) | 2, len(body)) + body
        opt += _len_prefixed(0x0a, len(single)) + single
      st += nfloats
    return b + _len_prefixed(0x1a, len(opt)) + opt + b"\x20" + _varint(ts)

  def _decode_one(self, b):
    """Generic EntryDump parse of one record (any field order, packed or unpacked floats)."""
    import struct
    b = bytes(b)
    rid, nums, opt, ts = 0, [], b"", 0

    ints: Dict[int, int] = {}

    def floats(body, want_fields):
      res: Dict[int, List[float]] = {f: [] for f in want_fields}
      ints.clear()
      pos = 0
      while pos < len(body):
        tag, pos = _read_varint(body, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 5:
          v = struct.unpack_from("<f", body, pos)[0]
          pos += 4
          if f in res:
            res[f].append(v)
        elif wt == 2:
          n, pos = _read_varint(body, pos)
          if f in res:
            res[f].extend(struct.unpack_from("<%df" % (n // 4), body, pos))
          pos += n
        elif wt == 0:
          ints[f], pos = _read_varint(body, pos)
        elif wt == 1:
          pos += 8
        else:
          raise ValueError("unsupported wire type %d" % wt)
      return res

    pos = 0
    while pos < len(b):
      tag, pos = _read_varint(b, pos)
      f, wt = tag >> 3, tag & 7
      if f == 1 and wt == 1:
        rid = struct.unpack_from("<q", b, pos)[0]
        pos += 8
      elif f == 2 and wt == 5:
        nums.append(struct.unpack_from("<f", b, pos)[0])
        pos += 4
      elif f == 2 and wt == 2:
        n, pos = _read_varint(b, pos)
        nums.extend(struct.unpack_from("<%df" % (n // 4), b, pos))
        pos += n
      elif f == 3 and wt == 2:
        n, pos = _read_varint(b, pos)
        opt = b[pos:pos + n]
        pos += n
      elif f == 4 and wt == 0:
        ts, pos = _read_varint(b, pos)
      elif wt == 0:
        _, pos = _read_varint(b, pos)
      elif wt == 2:
        n, pos = _read_varint(b, pos)
        pos += n
      elif wt == 5:
        pos += 4
      elif wt == 1:
        pos += 8
      else:
        raise ValueError("unsupported wire type %d" % wt)
    if len(nums) != self.dim:
      raise ValueError("EntryDump of id %d has %d values, the table expects %d" %
                       (rid, len(nums), self.dim))
    row = np.zeros(self.row_floats, np.float32)
    row[:self.dim] = nums
    singles = []
    pos = 0
    while pos < len(opt):
      tag, pos = _read_varint(opt, pos)
      n, pos = _read_varint(opt, pos)
      if tag >> 3 == 1:
        singles.append(opt[pos:pos + n])
      pos += n
    dumping = [(d, k) for d, k in self.segments if _DUMP_FIELDS[k][0] is not None]
    if len(singles) != len(dumping):
      raise ValueError("EntryDump of id %d has %d optimizer dumps, the table has %d dumping "
                       "segments" % (rid, len(singles), len(dumping)))
    st = self.dim
    for (d, k), single in zip(dumping, singles):
      tag, p2 = _read_varint(single, 0) if single else (0, 0)
      n, p2 = _read_varint(single, p2) if single else (0, 0)
      body = single[p2:p2 + n]
      kind_field = tag >> 3
      field, parts, nfloats = _dump_parts(k, d)
      if kind_field != field:
        raise ValueError("segment expects optimizer dump field %d, found field %d" %
                         (field, kind_field))
      r = floats(body, [fno for fno, _, _ in parts])
      if k in _INT64_PART:  # an absent field reads as 0, like the proto2 getter
        a = ints.get(_INT64_PART[k], 0) & 0xFFFFFFFFFFFFFFFF
        row[st:st + 2] = np.array([a], np.uint64).view(np.float32)
      for fno, off, cnt in parts:
        if len(r[fno]) != cnt:
          raise ValueError("optimizer dump field %d has %d values, expected %d" %
                           (fno, len(r[fno]), cnt))
        row[st + off:st + off + cnt] = r[fno]
      st += nfloats
    return rid, row, np.uint32(ts & 0xFFFFFFFF)


# ----------------------------------------------------------------------------- table level
def pick_nshards(size: int, nshards: int = -1) -> int:
  """HashTableSaveOp::PickNshards (hash_table_save_op.cc:116-121)."""
  if nshards >= 0:
    return nshards
  return int(min(4, max(1, size // 1000000)))


def write_shards(basename: str, layout: DumpLayout, ids, rows, ts, nshards: int):
  """Writes <basename>-%05d-of-%05d; rows go to shard floormod(id, nshards) (the reference's
  split follows its cuckoo buckets and is unobservable; restore reads every shard)."""
  os.makedirs(os.path.dirname(os.path.abspath(basename)), exist_ok=True)
  ids = np.asarray(ids, np.int64)
  shard = np.mod(ids, nshards) if nshards > 1 else np.zeros(len(ids), np.int64)
  for i in range(nshards):
    sel = np.nonzero(shard == i)[0]
    name = sharded_name(basename, i, nshards)
    tmp = "%s-tmp-%d" % (name, os.getpid())
    with open(tmp, "wb") as f:
      if len(sel):
        for group in layout.encode(ids[sel], np.asarray(rows)[sel], np.asarray(ts)[sel]):
          frame_records(group).tofile(f)
    os.replace(tmp, name)  # RenameFile(tmp, final), hash_table_save_op.cc:163


def read_shards(basename: str, layout: DumpLayout, verify: bool = True):
  """All records of <basename>-*-of-* (ValidateShardedFiles: a complete, consistent set)."""
  files = sorted(f for f in glob.glob(glob.escape(basename) + "-*")
                 if re.fullmatch(r".*-\d{5}-of-\d{5}", f))
  if not files:
    raise FileNotFoundError("Unable to find the dump files for: " + basename)
  totals = {int(f[-5:]) for f in files}
  if len(totals) != 1 or len(files) != totals.pop():
    raise ValueError("inconsistent shard files for " + basename)
  parts = []
  for f in files:
    buf, off, ln = read_records(f, verify)
    if len(off):
      parts.append(layout.decode(buf, off, ln))
  if not parts:
    return (np.zeros(0, np.int64), np.zeros((0, layout.row_floats), np.float32),
            np.zeros(0, np.uint32))
  return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))


# ============================================================================ multi-table dumps
# ops/multi_hash_table_save_restore_ops.cc:143-213 (save), :264-352 (restore). Per shard i of n:
#   <basename>-%05d-of-%05d        the EntryDumps of ALL member tables, table after table, as a
#                                  TFRecord stream under RecordWriterOptions::SNAPPY_COMPRESSION
#   <basename>.meta-%05d-of-%05d   one plain TFRecord per table: MultiHashTableMetadata
#                                  {1: string table_name, 2: uint64 num_entries}
#                                  (embedding_hash_table.proto:135-138)
# TensorFlow's snappy record stream (third-party: tensorflow/core/lib/io/snappy/
# snappy_outputbuffer.cc at the reference's pinned TensorFlow 2.4.0, absent here) is restated from
# its published form: the framed records are cut into input blocks of at most 256 KiB
# (RecordWriterOptions' snappy input buffer), each written as a 4-byte big-endian length of the
# compressed block followed by the raw snappy block. The reference ships no multi-table dump, so
# the BYTES are parity-unpinned (stated in the tests); the reader accepts any block size.
SNAPPY_BLOCK = 256 * 1024


def _snappy():
  import pyarrow as pa  # raw snappy block codec
  return pa.Codec("snappy")


def snappy_stream_compress(raw: bytes, block: int = SNAPPY_BLOCK) -> bytes:
  codec, out = _snappy(), []
  for lo in range(0, len(raw), block):
    c = codec.compress(raw[lo:lo + block]).to_pybytes()
    out.append(len(c).to_bytes(4, "big"))
    out.append(c)
  return b"".join(out)


def snappy_stream_decompress(data: bytes) -> bytes:
  codec, out, p = _snappy(), [], 0
  while p < len(data):
    if p + 4 > len(data):
      raise ValueError("snappy record stream: truncated block length")
    n = int.from_bytes(data[p:p + 4], "big")
    p += 4
    if n == 0 or p + n > len(data):
      raise ValueError("snappy record stream: truncated block")
    blk = data[p:p + n]
    size, _ = _read_varint(blk, 0)  # a raw snappy block opens with its uncompressed length
    out.append(codec.decompress(blk, decompressed_size=size).to_pybytes())
    p += n
  return b"".join(out)


def sharded_meta_name(basename: str, idx: int, total: int) -> str:
  return "%s.meta-%05d-of-%05d" % (basename, idx, total)


def encode_multi_meta(table_name: str, num_entries: int) -> bytes:
  nb = table_name.encode()
  return b"\x0a" + _varint(len(nb)) + nb + b"\x10" + _varint(int(num_entries))


def decode_multi_meta(b: bytes) -> Tuple[str, int]:
  name, n = "", 0
  for fno, wt, val in _parse_fields(bytes(b)):
    if fno == 1 and wt == 2:
      name = bytes(val).decode()
    elif fno == 2 and wt == 0:
      n = int(val)
  return name, n


def _frame_one(payload: bytes) -> bytes:
  return frame_records(np.frombuffer(payload, np.uint8).reshape(1, -1)).tobytes()


def write_multi_shards(basename: str, tables, nshards: int) -> None:
  """tables: [(name, DumpLayout, ids, rows, ts)] in the table set's order."""
  os.makedirs(os.path.dirname(os.path.abspath(basename)), exist_ok=True)
  nshards = max(1, int(nshards))
  for i in range(nshards):
    data, meta = [], []
    for name, layout, ids, rows, ts in tables:
      ids = np.asarray(ids, np.int64)
      sel = np.nonzero(np.mod(ids, nshards) == i)[0] if nshards > 1 else np.arange(len(ids))
      if len(sel):
        for group in layout.encode(ids[sel], np.asarray(rows)[sel], np.asarray(ts)[sel]):
          data.append(frame_records(group).tobytes())
      meta.append(_frame_one(encode_multi_meta(name, len(sel))))
    for name, payload in ((sharded_name(basename, i, nshards), snappy_stream_compress(b"".join(data))),
                          (sharded_meta_name(basename, i, nshards), b"".join(meta))):
      tmp = "%s-tmp-%d" % (name, os.getpid())
      with open(tmp, "wb") as f:
        f.write(payload)
      os.replace(tmp, name)


def read_multi_shards(basename: str, layouts: Dict[str, "DumpLayout"], verify: bool = True):
  """{table name: (ids, rows, ts)} for the tables of `layouts` found in the dump; tables of the
  dump that `layouts` does not name are skipped record by record, as the reference does."""
  files = sorted(f for f in glob.glob(glob.escape(basename) + "-*")
                 if re.fullmatch(r".*-\d{5}-of-\d{5}", f))
  if not files:
    raise FileNotFoundError("Unable to find the dump files for: " + basename)
  totals = {int(f[-5:]) for f in files}
  if len(totals) != 1 or len(files) != totals.pop():
    raise ValueError("inconsistent shard files for " + basename)
  parts: Dict[str, list] = {}
  for i, f in enumerate(files):
    with open(f, "rb") as fh:
      buf = np.frombuffer(snappy_stream_decompress(fh.read()), np.uint8)
    buf, off, ln = scan_records(buf, verify)
    mbuf, moff, mln = read_records(sharded_meta_name(basename, i, len(files)), verify)
    at = 0
    for o, l in zip(moff.tolist(), mln.tolist()):
      name, n = decode_multi_meta(mbuf[o:o + l].tobytes())
      if at + n > len(off):
        raise ValueError("Couldn't read all of checkpoint shard %d" % i)
      if name in layouts and n:
        parts.setdefault(name, []).append(layouts[name].decode(buf, off[at:at + n], ln[at:at + n]))
      at += n
    if at != len(off):
      raise ValueError("Couldn't read all of checkpoint shard %d" % i)
  return {name: tuple(np.concatenate([p[k] for p in ps]) for k in range(3))
          for name, ps in parts.items()}


# ============================================================================ hash-filter dumps
# ops/hash_filter_save_op.cc:73-105: one file per block ("split") of the sliding filter,
# <basename>-%05d-of-%05d with total = split_num; record 0 = HashFilterSplitMetaDump, then
# HashFilterSplitDataDump records of at most 10000 cells (hash_filter.cc:25-56).
#   HashFilterSplitMetaDump   (embedding_hash_table.proto:122-128)  1: uint64 failure_count
#       2: uint64 total_size  3: uint64 num_elements  4: double fill_rate  5: SlidingHashFilterMetaDump
#   SlidingHashFilterMetaDump (:108-116)  1: split_num 2: max_forward_step 3: max_backward_step
#       4: max_step 5: head 6: head_increment (uint32)  7: uint64 failure_count
#   HashFilterSplitDataDump   (:130-133)  1: uint32 offset  2: repeated uint32 data (proto2: NOT
#       packed, one 0x10 tag per cell)
# proto2 writes every field that was set, zero or not; the reference sets them all.
FILTER_CELLS_PER_RECORD = 10000
_SLIDING_FIELDS = ("split_num", "max_forward_step", "max_backward_step", "max_step", "head",
                   "head_increment", "failure_count")


def _read_varint(b: bytes, p: int) -> Tuple[int, int]:
  v = shift = 0
  while True:
    if p >= len(b):
      raise ValueError("truncated varint")
    c = b[p]
    p += 1
    v |= (c & 0x7F) << shift
    if c < 0x80:
      return v, p
    shift += 7
    if shift > 63:
      raise ValueError("varint too long")


def _parse_fields(b: bytes):
  """Generic proto wire parser: yields (field number, wire type, value)."""
  p = 0
  while p < len(b):
    tag, p = _read_varint(b, p)
    fno, wt = tag >> 3, tag & 7
    if wt == 0:
      v, p = _read_varint(b, p)
    elif wt == 1:
      v, p = b[p:p + 8], p + 8
    elif wt == 5:
      v, p = b[p:p + 4], p + 4
    elif wt == 2:
      n, p = _read_varint(b, p)
      v, p = b[p:p + n], p + n
    else:
      raise ValueError("unsupported wire type %d" % wt)
    if p > len(b):
      raise ValueError("truncated field %d" % fno)
    yield fno, wt, v


def encode_filter_meta(meta: Dict) -> bytes:
  import struct
  sl = b"".join(bytes([(i + 1) << 3]) + _varint(meta["sliding"][k])
                for i, k in enumerate(_SLIDING_FIELDS))
  return (b"\x08" + _varint(meta["failure_count"]) + b"\x10" + _varint(meta["total_size"]) +
          b"\x18" + _varint(meta["num_elements"]) + b"\x21" + struct.pack("<d", meta["fill_rate"]) +
          b"\x2a" + _varint(len(sl)) + sl)


def decode_filter_meta(b: bytes) -> Dict:
  import struct
  meta = {"failure_count": 0, "total_size": 0, "num_elements": 0, "fill_rate": 0.0,
          "sliding": {k: 0 for k in _SLIDING_FIELDS}}
  names = {1: "failure_count", 2: "total_size", 3: "num_elements"}
  for fno, wt, v in _parse_fields(bytes(b)):
    if fno in names and wt == 0:
      meta[names[fno]] = v
    elif fno == 4 and wt == 1:
      meta["fill_rate"] = struct.unpack("<d", v)[0]
    elif fno == 5 and wt == 2:
      for f2, w2, v2 in _parse_fields(v):
        if 1 <= f2 <= 7 and w2 == 0:
          meta["sliding"][_SLIDING_FIELDS[f2 - 1]] = v2
  return meta


def encode_filter_data(offset: int, cells: np.ndarray) -> bytes:
  """One HashFilterSplitDataDump: offset + the cells, each as tag 0x10 + varint."""
  c = np.ascontiguousarray(cells, dtype=np.uint32)
  nb = 1 + (c >= 0x80).astype(np.int64) + (c >= 0x4000).astype(np.int64) + \
      (c >= 0x200000).astype(np.int64) + (c >= 0x10000000).astype(np.int64)
  start = np.zeros(len(c) + 1, np.int64)
  np.cumsum(nb + 1, out=start[1:])
  out = np.empty(int(start[-1]), np.uint8)
  out[start[:-1]] = 0x10
  for k in range(5):
    sel = nb > k
    if not sel.any():
      break
    byte = (c[sel] >> np.uint32(7 * k)) & np.uint32(0x7F)
    byte |= np.where(nb[sel] > k + 1, 0x80, 0).astype(np.uint32)
    out[start[:-1][sel] + 1 + k] = byte.astype(np.uint8)
  return b"\x08" + _varint(offset) + out.tobytes()


def decode_filter_data(b) -> Tuple[int, np.ndarray]:
  a = np.frombuffer(bytes(b), np.uint8)
  if a.size == 0:
    return 0, np.zeros(0, np.uint32)
  ends = np.flatnonzero(a < 0x80)
  if ends.size == 0 or ends[-1] != a.size - 1 or ends.size % 2:
    raise ValueError("corrupt HashFilterSplitDataDump")
  starts = np.concatenate([[0], ends[:-1] + 1])
  pos = np.arange(a.size) - np.repeat(starts, ends - starts + 1)
  if pos.max() > 9:
    raise ValueError("corrupt HashFilterSplitDataDump (varint too long)")
  vals = np.add.reduceat((a & 0x7F).astype(np.uint64) << (7 * pos).astype(np.uint64), starts)
  tags, vals = vals[0::2], vals[1::2]
  if np.any((ends - starts)[0::2] != 0) or not np.all((tags == 0x08) | (tags == 0x10)):
    # packed or unknown fields: take the slow generic parser
    off, data = 0, []
    for fno, wt, v in _parse_fields(bytes(b)):
      if fno == 1 and wt == 0:
        off = v
      elif fno == 2 and wt == 0:
        data.append(v)
      elif fno == 2 and wt == 2:
        p = 0
        while p < len(v):
          x, p = _read_varint(v, p)
          data.append(x)
    return int(off), np.array(data, np.uint32)
  off = vals[tags == 0x08]
  return (int(off[-1]) if off.size else 0), vals[tags == 0x10].astype(np.uint32)


def write_filter_split(path: str, meta: Dict, cells: np.ndarray) -> None:
  """hash_filter.cc:27-56 through ops/hash_filter_save_op.cc:73-105 (tmp file + rename)."""
  recs = [encode_filter_meta(meta)]
  for s in range(0, len(cells), FILTER_CELLS_PER_RECORD):
    recs.append(encode_filter_data(s, cells[s:s + FILTER_CELLS_PER_RECORD]))
  os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
  tmp = "%s-tmp-%d" % (path, int.from_bytes(os.urandom(8), "little"))
  with open(tmp, "wb") as f:
    for r in recs:
      f.write(frame_records(np.frombuffer(r, np.uint8).reshape(1, -1)).tobytes())
  os.replace(tmp, path)


def read_filter_split(path: str, verify: bool = True) -> Tuple[Dict, np.ndarray]:
  """(meta, cells uint16[total_size + 64]) — hash_filter.cc:60-78: cells start zeroed and every
  data record overwrites [offset, offset + len)."""
  buf, off, ln = read_records(path, verify)
  if not len(off):
    raise ValueError("empty hash filter dump: " + path)
  meta = decode_filter_meta(buf[off[0]:off[0] + ln[0]].tobytes())
  cells = np.zeros(int(meta["total_size"]) + 64, np.uint16)
  for i in range(1, len(off)):
    o, data = decode_filter_data(buf[off[i]:off[i] + ln[i]].tobytes())
    if o + len(data) > len(cells):
      raise ValueError("hash filter dump %s: data beyond total_size + 64" % path)
    cells[o:o + len(data)] = data.astype(np.uint16)
  return meta, cells


def filter_files(basename: str) -> List[str]:
  """Complete, consistent shard set of <basename>-*-of-* (file_utils.cc ValidateShardedFiles)."""
  files = sorted(f for f in glob.glob(glob.escape(basename) + "-*")
                 if re.fullmatch(r".*-\d{5}-of-\d{5}", f))
  if not files:
    raise FileNotFoundError("Unable to find the dump files for: " + basename)
  totals = {int(f[-5:]) for f in files}
  if len(totals) != 1 or len(files) != totals.pop():
    raise ValueError("inconsistent shard files for " + basename)
  return files
