"""serving entries (entry_accessor.cc:43-98) and their float compressors
(compressor/float_compressor.cc:42-137).

CPU: the oracle's Decode(Encode(v)) against the reference's unit-test vectors
(float_compressor_test.cc:53-91) and, where oracle/_ref holds the harness built from the
reference's own dependency-free headers (half.hpp, fake_quantizer.h), against those bit for bit.
GPU: a serving table on the HIP path against the oracle on assign / assign_add / restore / lookup
streams; the operations a serving entry refuses."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import cpu
from tests import oracle_backend

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                   "libcompress_ref.so")
KIND = {"fp32": 0, "fp16": 1, "fixed_r8": 2, "one_bit": 3}


def _q(kind, x, a=0.0, b=1.0):
  x = np.ascontiguousarray(x, np.float32)
  out = np.empty_like(x)
  cpu.lib().orc_compress_roundtrip(C.c_int(KIND[kind]), C.c_float(a), C.c_float(b),
                                   C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data),
                                   C.c_int64(x.size))
  return out


def _samples():
  rng = np.random.Generator(np.random.PCG64(4))
  bits = rng.integers(0, 2**32, 2_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
  hf = np.arange(0, 0x7c00, dtype=np.uint16)
  lo = hf.view(np.float16).astype(np.float64)
  hi = (hf + 1).astype(np.uint16).view(np.float16).astype(np.float64)
  mid = ((lo + hi) / 2).astype(np.float32)          # exact ties between neighbouring halves
  near = np.concatenate([np.nextafter(mid, np.float32(0)), np.nextafter(mid, np.float32(1e9))])
  edge = np.array([0.0, -0.0, 65504, 65519.996, 65520, 65536, 1e30, -1e30, np.inf, -np.inf,
                   2.98e-8, 2.9802322e-8, 3e-8, 5.96e-8, 6.1e-5, 6.097555e-5, 1e-45], np.float32)
  small = rng.normal(0, 2.0, 500_000).astype(np.float32)
  return np.concatenate([bits, mid, -mid, near, -near, edge, small])


def test_reference_vectors():
  """float_compressor_test.cc:53-91."""
  x = np.array([0.1, 0.2, 10000.0], np.float32)
  assert np.array_equal(_q("fp32", x), x)
  assert np.abs(_q("fp16", x) - x).max() < 1e-4 + 1e-7
  k = np.float32(5.0) / np.float32(128)
  assert np.abs(_q("fixed_r8", [100.0, 0.0, 3.5], a=5.0) - [5.0, 0.0, 3.5]).max() <= k
  got = _q("fixed_r8", np.array([k * 1.4, k * 1.6, -k * 1.4, -k * 1.6], np.float32), a=5.0)
  assert np.array_equal(got, np.array([k, k * 2, -k, -k * 2], np.float32))
  got = _q("one_bit", [100.0, 0.1, 0.00001, 0.0, -0.00001, -0.1, -100.0], a=1.0, b=1.0)
  assert np.array_equal(got, [1.0, 1.0, 1.0, -1.0, -1.0, -1.0, -1.0])


def test_fp16_equals_ieee_round_to_nearest_even():
  x = _samples()
  with np.errstate(over="ignore", invalid="ignore"):
    exp = x.astype(np.float16).astype(np.float32)
  got = _q("fp16", x)
  nan = np.isnan(x)
  assert np.isnan(got[nan]).all()
  assert np.array_equal(got[~nan].view(np.uint32), exp[~nan].view(np.uint32))


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (reference absent)")
def test_oracle_equals_reference_headers():
  """half_float::half(v) and FakeQuantizer of the reference itself, compiled where they lie."""
  ref = C.CDLL(REF)
  x = _samples()
  out = np.empty_like(x)
  ref.ref_fp16_roundtrip(C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data), C.c_int64(x.size))
  got = _q("fp16", x)
  nan = np.isnan(x)
  assert np.isnan(out[nan]).all() and np.isnan(got[nan]).all()
  assert np.array_equal(got[~nan].view(np.uint32), out[~nan].view(np.uint32))
  for r in (1.0, 5.0, 0.3):
    ref.ref_fixed_r8(C.c_float(r), C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data),
                     C.c_int64(x.size))
    got = _q("fixed_r8", x, a=r)
    assert np.array_equal(got.view(np.uint32), out.view(np.uint32)), r


SERVING = {"serving": True, "segments": [
    {"dim": 3, "comp": "fp16", "opt": "adagrad"}, {"dim": 2, "comp": "fixed_r8", "comp_a": 5.0},
    {"dim": 2, "comp": "one_bit", "comp_a": 0.1}, {"dim": 1, "comp": "fp32", "opt": "ftrl"},
    {"dim": 4, "comp": "fixed_r8", "comp_a": 1.0}]}


def test_serving_oracle_table_semantics():
  t = oracle_backend.make_table(SERVING)
  assert t.row_floats == t.dim == 12              # no optimizer context in a serving entry
  ids = np.array([5, 9, 5], np.int64)
  v = np.full((3, 12), 0.3337, np.float32)
  v[2] = -2.6
  t.assign(ids, v)
  row = t.lookup(np.array([5, 9, 77], np.int64))
  assert np.array_equal(row[2], np.zeros(12, np.float32))
  assert np.array_equal(row[1][:3], np.float32([0.3337] * 3).astype(np.float16).astype(np.float32))
  assert np.array_equal(row[0][5:7], np.float32([-0.1, -0.1])) and row[0][7] == np.float32(-2.6)
  assert np.array_equal(row[0][8:], np.float32([-1.0] * 4))    # clamped to -r
  t.assign_add(np.array([9, 9], np.int64), np.full((2, 12), 0.01, np.float32))
  a = np.float32(0.3337).astype(np.float16).astype(np.float32)
  for _ in range(2):                               # re-encoded after every add
    a = (a + np.float32(0.01)).astype(np.float16).astype(np.float32)
  assert t.lookup(np.array([9], np.int64))[0][0] == a


@pytest.mark.gpu
def test_serving_table_matches_oracle(tmp_path):
  from tests import hip_backend
  o, h = oracle_backend.make_table(SERVING), hip_backend.make_table(SERVING)
  assert h.row_floats == h.dim == 12
  rng = np.random.Generator(np.random.PCG64(31))
  probe = np.arange(-20, 420, dtype=np.int64)
  for step in range(6):
    n = 4000
    ids = rng.zipf(1.4, n).astype(np.int64) % 400 - 10
    if step == 2:
      ids[::2] = 7   # 2000 adds on one entry: walked whole, re-encoded after every add
    v = (rng.normal(0, 1.5, (n, 12)) * rng.choice([1e-3, 1, 40], (n, 1))).astype(np.float32)
    for t in (o, h):
      (t.assign if step % 3 == 0 else t.assign_add)(ids, v, req_time=50 + step)
    assert np.array_equal(o.lookup(probe).view(np.uint32), h.lookup(probe).view(np.uint32)), step
  a, b = o.export(), h.export()
  assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
  # the operations a serving entry refuses (entry_accessor.cc:79-87)
  from monolith_amd import _lib
  with pytest.raises(_lib.MonolithHipError, match="doesn't support Optimize"):
    h.apply_gradients(np.array([1], np.int64), np.zeros((1, 12), np.float32), [0.1] * 5)
  with pytest.raises(RuntimeError, match="doesn't support Save"):
    h.t.save(str(tmp_path / "x"))


@pytest.mark.gpu
def test_serving_table_restores_a_training_dump(tmp_path):
  """The export path: a table trained with optimizers is saved, a serving table with the same
  segments restores it (weights only, through the compressors) and serves lookups."""
  from tests import hip_backend
  train = dict(SERVING, serving=False)
  tr = hip_backend.make_table(train)
  rng = np.random.Generator(np.random.PCG64(32))
  ids = rng.integers(0, 300, 2000).astype(np.int64)
  tr.apply_gradients(ids, rng.normal(0, 1, (2000, 12)).astype(np.float32), [0.05] * 5, req_time=9)
  base = str(tmp_path / "tr")
  tr.t.save(base)
  sv = hip_backend.make_table(SERVING)
  sv.t = sv.t.restore(base)
  o = oracle_backend.make_table(SERVING)
  tid, trow, tts = tr.export()
  o.import_rows(tid, np.ascontiguousarray(trow[:, :12]), tts)
  probe = np.arange(-5, 305, dtype=np.int64)
  assert np.array_equal(o.lookup(probe).view(np.uint32), sv.lookup(probe).view(np.uint32))


# ------------------------------------------------------------------ fp16 stochastic rounding
def test_half_neighbours_and_stochastic_round_against_reference_header():
  """stochastic_rounding.h:27-37. up / down = float2half<round_toward_(neg_)infinity> of the
  reference's half.hpp (oracle/_ref), bit for bit incl. overflow, subnormals and signed zeros;
  the draw itself (a thread-local sequence in the reference) is not pinnable."""
  x = _samples()
  up, down = np.empty_like(x), np.empty_like(x)
  cpu.lib().orc_half_up_down(C.c_void_p(x.ctypes.data), C.c_void_p(up.ctypes.data),
                             C.c_void_p(down.ctypes.data), C.c_int64(x.size))
  fin = np.isfinite(x)
  assert (up[fin] >= x[fin]).all() and (down[fin] <= x[fin]).all()
  with np.errstate(over="ignore", invalid="ignore"):
    exact = x.astype(np.float16).astype(np.float32) == x
  assert np.array_equal(up[exact & fin], x[exact & fin]) and np.array_equal(down[exact & fin], x[exact & fin])
  if os.path.exists(REF):
    ref = C.CDLL(REF)
    ru, rd = np.empty_like(x), np.empty_like(x)
    ref.ref_half_up_down(C.c_void_p(x.ctypes.data), C.c_void_p(ru.ctypes.data),
                         C.c_void_p(rd.ctypes.data), C.c_int64(x.size))
    nan = np.isnan(x)
    assert np.array_equal(ru[~nan].view(np.uint32), up[~nan].view(np.uint32))
    assert np.array_equal(rd[~nan].view(np.uint32), down[~nan].view(np.uint32))
  # the rule: p <= (v - down) / (up - down) -> up; a representable value is returned unchanged
  v = np.float32([0.1, 0.1, 0.1, 0.5, -0.1, 70000.0])
  p = np.float32([0.0, 0.3, 1.0, 0.7, 0.5, 0.5])
  out = np.empty_like(v)
  cpu.lib().orc_stochastic_round(C.c_void_p(v.ctypes.data), C.c_void_p(p.ctypes.data),
                                 C.c_void_p(out.ctypes.data), C.c_int64(v.size))
  h = np.float32(0.1).astype(np.float16).astype(np.float32)   # 0.0999755859375 < 0.1
  hi = np.nextafter(np.float16(h), np.float16(1)).astype(np.float32)
  frac = (np.float32(0.1) - h) / (hi - h)
  assert out[0] == hi and out[1] == (hi if 0.3 <= frac else h) and out[2] == h
  assert out[3] == 0.5 and out[4] in (-h, -hi) and out[5] == 65504.0
  # unbiased on average
  n = 200_000
  v = np.full(n, 0.1, np.float32)
  p = np.random.Generator(np.random.PCG64(3)).random(n, dtype=np.float32)
  out = np.empty_like(v)
  cpu.lib().orc_stochastic_round(C.c_void_p(v.ctypes.data), C.c_void_p(p.ctypes.data),
                                 C.c_void_p(out.ctypes.data), C.c_int64(n))
  assert abs(out.astype(np.float64).mean() - 0.1) < 2e-7 * 50


SR16 = {"seed": 17, "segments": [
    {"dim": 2, "opt": "adagrad", "init": "uniform", "init_a": -0.1, "init_b": 0.1},
    {"dim": 3, "opt": "sgd", "sr16": True, "init": "uniform", "init_a": -0.1, "init_b": 0.1},
    {"dim": 2, "opt": "adam", "init": "uniform", "init_a": -0.1, "init_b": 0.1},
    {"dim": 1, "opt": "ftrl", "sr16": True, "beta": 1.0}]}


def test_rounding_span_runs_to_the_end_of_the_entry():
  """Inside CombinedOptimizer the decorated segment's span keeps the tail of the entry
  (optimizer_combination.cc:58-62): segment 1's rounding also rounds segments 2 and 3 — before
  their own steps; segment 0 is never rounded."""
  t = oracle_backend.make_table(SR16)
  ids = np.array([3], np.int64)
  t.apply_gradients(ids, np.full((1, 8), 0.25, np.float32), [0.01] * 4, global_step=5)
  _, rows, _ = t.export()
  w = rows[0][:8]
  is_half = lambda a: np.array_equal(a.astype(np.float16).astype(np.float32), a)
  assert not is_half(w[:2]) and is_half(w[2:5]) and is_half(w[7:8])
  assert not is_half(w[5:7])      # Adam stepped from the rounded value, result not rounded


@pytest.mark.gpu
@pytest.mark.parametrize("dedup", [False, True])
def test_stochastic_rounding_table_matches_oracle(dedup):
  from tests import hip_backend
  o, h = oracle_backend.make_table(SR16), hip_backend.make_table(SR16)
  rng = np.random.Generator(np.random.PCG64(41 + dedup))
  probe = np.arange(-20, 320, dtype=np.int64)
  for step in range(5):
    n = 3000
    ids = rng.zipf(1.4, n).astype(np.int64) % 300 - 10
    if step == 2:
      ids[::3] = 4   # 1000 steps on one entry, each with its own draws
    g = rng.normal(0, 0.3, (n, 8)).astype(np.float32)
    for t in (o, h):
      t.apply_gradients(ids, g, [0.01, 0.02, 0.03, 0.04], req_time=9 + step, global_step=100 + step,
                        enable_dedup=dedup)
    assert np.array_equal(o.lookup(probe).view(np.uint32), h.lookup(probe).view(np.uint32)), step
  a, b = o.export(), h.export()
  assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
