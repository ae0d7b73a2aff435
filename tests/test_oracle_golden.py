"""Pins the CPU oracle against the reference's own unit-test vectors (CPU only).

Every case cites the reference test it comes from; tests/golden/avx_vectors.json was produced
by the reference's avx_utils.h itself (tests/golden/gen_avx_vectors.py).
"""
import json
import os

import numpy as np
import pytest

from oracle import cpu
from tests import oracle_backend
from tests.scenario import GOLDEN, load_scenarios, run_scenario


@pytest.mark.parametrize("sc", load_scenarios(), ids=lambda s: s["name"])
def test_table_scenarios(sc):
  run_scenario(oracle_backend, sc)


def _f(bits):
  return np.array(bits, np.uint32).view(np.float32)


def test_adagrad_matches_reference_header_bit_exact():
  """Oracle Adagrad == BaselineAdagradOptimize (avx_utils.h:29-38) bit for bit; the AVX form
  (avx_utils.h:96-119) agrees to 1e-6 relative when weight_decay == 0."""
  with open(os.path.join(GOLDEN, "avx_vectors.json")) as f:
    vec = json.load(f)
  for case in vec["adagrad"]:
    num, norm = _f(case["num0"]).copy(), _f(case["norm0"]).copy()
    for st in case["steps"]:
      num, norm = cpu.adagrad(num, norm, _f(st["grad"]), case["lr"], case["wd"])
      assert np.array_equal(num.view(np.uint32), np.array(st["baseline_num"], np.uint32))
      assert np.array_equal(norm.view(np.uint32), np.array(st["baseline_norm"], np.uint32))
      if case["wd"] == 0.0:
        np.testing.assert_allclose(num, _f(st["avx_num"]), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(norm, _f(st["avx_norm"]), rtol=1e-6, atol=1e-9)
  known = vec["adagrad_test_known"]
  np.testing.assert_allclose(known["first"], [-0.07071067, -0.08944272], atol=1e-6)
  np.testing.assert_allclose(known["second"], [-0.128173, -0.155943], atol=1e-6)


def test_reduce_sum_matches_reference_header():
  """In-order pairwise add == ReduceSum (avx_utils.h:88-93,224-235), exact."""
  with open(os.path.join(GOLDEN, "avx_vectors.json")) as f:
    vec = json.load(f)
  for c in vec["reduce_sum"]:
    a, b = _f(c["a"]), _f(c["b"])
    got = cpu.reduce_sum([0, 0], np.stack([a, b]), 1)[0]
    # oracle sums 0 + a + b; (0 + a) is exact, so equality with a + b is exact
    assert np.array_equal(got.view(np.uint32), np.array(c["out"], np.uint32))


# ---------------------------------------------------------------- filter
def test_sliding_filter_simple():
  """sliding_hash_filter_test.cc:23-36: add returns the old count 0..15 and saturates."""
  for key_num in (1, 3, 100):
    f = cpu.OracleFilter("sliding", key_num, 10)
    for i in range(16):
      assert f.add([1], [1])[0] == i
    assert f.add([1], [1])[0] == 15
    assert f.get([1])[0] == 15


def test_sliding_filter_count():
  """sliding_hash_filter_test.cc:38-55: one element after three adds of the same fid."""
  f = cpu.OracleFilter("sliding", 1000000, 10)
  fid = np.array([10000002961562801052], np.uint64).view(np.int64)
  f.add(fid, [1]); f.add(fid, [20]); f.add(fid, [1])
  assert sum(f.state()["num_elements"]) == 1
  rng = np.random.Generator(np.random.PCG64(1))
  f2 = cpu.OracleFilter("sliding", 1000000, 10)
  ids = rng.integers(1, 2**31, 11)
  f2.add(ids, np.full(11, 2))
  assert sum(f2.state()["num_elements"]) == 11


def test_sliding_filter_zero_threshold_skipped():
  """sliding_hash_filter_test.cc:93-104: threshold 0 never filters and never counts."""
  f0 = cpu.OracleFilter("sliding", 1000000, 10, default_threshold=0)
  f1 = cpu.OracleFilter("sliding", 1000000, 10, default_threshold=1)
  g = np.ones((5, 1), np.float32)
  assert np.array_equal(f0.intercept_gradient(np.arange(5), g), g)
  assert np.array_equal(f1.intercept_gradient(np.arange(5) * 2, g), np.zeros_like(g))
  assert sum(f0.state()["num_elements"]) == 0
  assert sum(f1.state()["num_elements"]) == 5


def test_filter_intercept_gradient_basic():
  """hash_filter_ops_test.py:42-58: ids [1, 3<<17, 1], threshold 3, capacity 100, split 7."""
  f = cpu.OracleFilter("sliding", 100, 7, default_threshold=3)
  ids = np.array([1, 3 << 17, 1], np.int64)
  g = np.ones((3, 2), np.float32)
  exp = [[[0, 0], [0, 0], [0, 0]], [[0, 0], [0, 0], [1, 1]], [[1, 1], [0, 0], [1, 1]],
         [[1, 1], [1, 1], [1, 1]]]
  for e in exp:
    assert np.array_equal(f.intercept_gradient(ids, g), np.array(e, np.float32))


def test_filter_rotation_across_blocks():
  """hash_filter_ops_test.py:105-177: capacity 300, split 100 -> total_size 3 per block; after
  two runs head == head_increment == 4 and the first four blocks hold 2 elements each."""
  f = cpu.OracleFilter("sliding", 300, 100, default_threshold=2)
  st = f.state()
  assert st["block_total_size"] == 3 and st["n_blocks"] == 100 and st["head"] == 0
  ids = np.array([1, 1 << 17, 2 << 17, 1], np.int64)
  g = np.ones((4, 2), np.float32)
  r1 = f.intercept_gradient(ids, g)
  assert np.array_equal(r1, np.zeros((4, 2), np.float32))
  r2 = f.intercept_gradient(ids, g)
  assert np.array_equal(r2, np.array([[1, 1], [0, 0], [0, 0], [1, 1]], np.float32))
  st = f.state()
  assert st["head"] == 4 and st["head_increment"] == 4
  assert st["num_elements"][:4] == [2, 2, 2, 2] and sum(st["num_elements"][4:]) == 0
  for _ in range(2):
    assert np.array_equal(f.intercept_gradient(ids, g), g)


def test_dummy_filter_never_filters():
  """hash_filter_ops_test.py test_dummy_hash_filter_basic; dummy_hash_filter.h:30-51."""
  f = cpu.OracleFilter("dummy")
  g = np.ones((3, 2), np.float32)
  assert np.array_equal(f.intercept_gradient([1, 2, 1], g), g)


# ---------------------------------------------------------------- routing
REORDER_CASES = [
    # distribution_ops_fused_test.py:52-119
    ([[0, 1, 2, 2, 3, 5]], 3, None, [0, 3, 1, 2, 5], [2, 1, 2], [2, 1, 2], None),
    ([[0, 1, 2, 2, 3, 5], []], 3, None, [0, 3, 1, 2, 5], [2, 1, 2], [2, 0, 1, 0, 2, 0], None),
    ([[0, 1, 2, 2, 3, 5], [6, 7, 8, 8, 9, 11]], 3, None, [0, 3, 6, 9, 1, 7, 2, 5, 8, 11],
     [4, 2, 4], [2, 2, 1, 1, 2, 2], None),
    ([[], []], 2, None, [], [0, 0], [0, 0, 0, 0], None),
    ([[0, 1, 4, 5], [2, 3, 6, 7]], 2, None, [0, 4, 2, 6, 1, 5, 3, 7], [4, 4], [2, 2, 2, 2], None),
    ([[0, 1, 0], [3, 2, 3], [5, 6, 7]], 2, [1, 2, 3], [0, 2, 6, 1, 3, 5, 7], [3, 4],
     [1, 1, 1, 1, 1, 2], [[0, 6, 0], [7, 1, 7], [9, 3, 12]]),
    ([[2, 3, 1, 2, 7, 2], [5, 8, 4, 4, 5, 11, 6]], 3, [3, 2], [3, 6, 1, 7, 4, 2, 5, 8, 11],
     [2, 3, 4], [1, 1, 2, 1, 1, 3], [[13, 0, 5, 13, 8, 13], [16, 18, 11, 11, 16, 20, 3]]),
]


@pytest.mark.parametrize("case", REORDER_CASES, ids=lambda c: "reorder%d" % len(c[3]))
def test_fused_reorder_by_indices(case):
  inputs, n, dims, out, sizes, sss, offs = case
  dims_ = dims or [1] * len(inputs)
  o, s, ss, of = cpu.fused_reorder_by_indices(inputs, n, dims_)
  assert o.tolist() == out and s.tolist() == sizes and ss.tolist() == sss
  if offs:
    assert [x.tolist() for x in of] == offs


def test_fused_reorder_rank0_empty_rule():
  """distribution_ops.py:255-266: N > 4 and MONOLITH_SYNC_EMPTY_RANK0_PS_SHARD=1 -> shard =
  1 + floormod(id, N-1); floormod is the sign-of-divisor mod."""
  ids = [np.array([-3, -1, 0, 5, 7, 12], np.int64)]
  o, s, ss, of = cpu.fused_reorder_by_indices(ids, 5, [1], shard_rule=1)
  shard = 1 + np.mod(ids[0], 4)
  assert s[0] == 0
  exp = [v for n in range(5) for v in ids[0][shard == n]]
  assert o.tolist() == exp
  o0, s0, _, _ = cpu.fused_reorder_by_indices(ids, 5, [1], shard_rule=0)
  exp0 = [v for n in range(5) for v in ids[0][np.mod(ids[0], 5) == n]]
  assert o0.tolist() == exp0


def _mk(dim, ids, vals):
  t = cpu.OracleTable([cpu.seg(dim, "sgd")])
  t.assign(ids, vals)
  return t


def test_fused_lookup_docstring_and_ops_test():
  """hash_table_ops.py:372-383 docstring example and hash_table_ops_test.py:1086-1109."""
  t0 = _mk(1, [1, 2], [[1], [2]])
  t1 = _mk(2, [3, 4], [[3, 3], [4, 4]])
  emb, splits, id_off, emb_off, sizes = cpu.fused_lookup([t0, t1], [1, 3, 2, 4], [1, 1, 1, 1], 2)
  assert emb.tolist() == [1, 3, 3, 2, 4, 4] and splits.tolist() == [3, 3]
  assert id_off.tolist() == [0, 1, 2, 3] and emb_off.tolist() == [0, 1, 3, 4]
  assert sizes.tolist() == [1, 2, 1, 2]

  tabs = []
  for x, d in enumerate([1, 1, 2]):
    v = np.ones((2, d), np.float32) if x % 2 == 0 else np.zeros((2, d), np.float32)
    tabs.append(_mk(d, [0 + 3 * x, 1 + 3 * x], v))
  emb, splits, id_off, emb_off, sizes = cpu.fused_lookup(tabs, [0, 4, 6, 1, 3, 7], [1] * 6, 2)
  assert emb.tolist() == [1, 0, 1, 1, 1, 0, 1, 1] and splits.tolist() == [4, 4]
  assert id_off.tolist() == [0, 1, 2, 3, 4, 5] and emb_off.tolist() == [0, 1, 2, 4, 5, 6]
  assert sizes.tolist() == [1, 1, 2, 1, 1, 2]


def test_fused_lookup_multi_type():
  """multi_type_hash_table_test.py:156-198 (1 and 2 shards, empty (shard, table) cells)."""
  t0 = _mk(1, [0], [[1]]); t1 = _mk(2, [1], [[2, 2]]); t2 = _mk(2, [2, 3], [[4, 4], [8, 8]])
  emb, splits, id_off, emb_off, sizes = cpu.fused_lookup([t0, t1, t2], [0, 1, 2, 3], [1, 1, 2], 1)
  assert emb.tolist() == [1, 2, 2, 4, 4, 8, 8] and splits.tolist() == [7]
  assert id_off.tolist() == [0, 1, 2] and emb_off.tolist() == [0, 1, 3] and sizes.tolist() == [1, 2, 4]
  emb, splits, id_off, emb_off, sizes = cpu.fused_lookup([t0, t1, t2], [0, 2, 1, 3],
                                                         [1, 0, 1, 0, 1, 1], 2)
  assert emb.tolist() == [1, 4, 4, 2, 2, 8, 8] and splits.tolist() == [3, 4]
  assert id_off.tolist() == [0, 1, 1, 2, 2, 3] and emb_off.tolist() == [0, 1, 1, 3, 3, 5]
  assert sizes.tolist() == [1, 0, 2, 0, 2, 2]


def test_fused_optimize():
  """hash_table_ops_test.py:1111-1153 and multi_type_hash_table_test.py:200-246."""
  t0 = _mk(1, [0, 1], np.ones((2, 1))); t1 = _mk(2, [3, 4], np.zeros((2, 2)))
  ids = [0, 4, 1, 3]
  _, _, id_off, emb_off, _ = cpu.fused_lookup([t0, t1], ids, [1, 1, 1, 1], 2)
  cpu.fused_apply_gradient([t0, t1], ids, [1, 1, 1, 1], [-1, -2, -2, -1, -2, -2], id_off, emb_off,
                           [0.1, 0.1], [1, 1], 0, 0, 2)
  emb, splits, *_ = cpu.fused_lookup([t0, t1], ids, [1, 1, 1, 1], 2)
  np.testing.assert_allclose(emb, [1.1, 0.2, 0.2, 1.1, 0.2, 0.2], rtol=1e-6)
  assert splits.tolist() == [3, 3]

  a = cpu.OracleTable([cpu.seg(1, "sgd")]); b = cpu.OracleTable([cpu.seg(2, "sgd")])
  _, _, id_off, emb_off, _ = cpu.fused_lookup([a, b], [0, 1, 2], [1, 2], 1)
  cpu.fused_apply_gradient([a, b], [0, 1, 2], [1, 2], [2.0, 1.0, 3.0, 2.0, 4.0], id_off, emb_off,
                           [1.0, 1.0], [1, 1], 0, 0, 1)
  emb, *_ = cpu.fused_lookup([a, b], [0, 1, 2], [1, 2], 1)
  assert emb.tolist() == [-2, -1, -3, -2, -4]

  a = cpu.OracleTable([cpu.seg(1, "sgd")]); b = cpu.OracleTable([cpu.seg(2, "sgd")])
  fss = [1, 0, 1, 0]
  _, _, id_off, emb_off, _ = cpu.fused_lookup([a, b], [1, 1], fss, 2)
  cpu.fused_apply_gradient([a, b], [1, 1], fss, [1.0, 2.0], id_off, emb_off, [1.0, 1.0], [1, 1],
                           0, 0, 2)
  emb, splits, id_off, emb_off, sizes = cpu.fused_lookup([a, b], [1, 1], fss, 2)
  assert emb.tolist() == [-3, -3] and splits.tolist() == [1, 1]
  assert id_off.tolist() == [0, 1, 1, 2] and emb_off.tolist() == [0, 1, 1, 2]
  assert sizes.tolist() == [1, 0, 1, 0]


def test_fused_gather_roundtrip():
  """distribution_ops_fused_test.py:113-119 layout: gather by the offsets of the last reorder
  case reproduces per-occurrence rows; gradient accumulates duplicates in order."""
  fused = np.array([1.1, 1.2, 1.3, 2.1, 2.2, 3.1, 3.2, 3.3, 4.1, 4.2, 4.3, 5.1, 5.2, 6.1, 6.2, 6.3,
                    7.1, 7.2, 8.1, 8.2, 9.1, 9.2], np.float32)
  offs = [[13, 0, 5, 13, 8, 13], [16, 18, 11, 11, 16, 20, 3]]
  outs = cpu.fused_gather_embeddings_by_input(fused, offs, [3, 2])
  np.testing.assert_array_equal(outs[0][0], fused[13:16])
  np.testing.assert_array_equal(outs[0][1], fused[0:3])
  np.testing.assert_array_equal(outs[1][6], fused[3:5])
  g = cpu.fused_gather_embeddings_by_input_gradient(len(fused), [np.ones((6, 3)), np.ones((7, 2))],
                                                    offs, [3, 2])
  assert g[13:16].tolist() == [3, 3, 3] and g[0:3].tolist() == [1, 1, 1]
  assert g[11:13].tolist() == [2, 2] and g[16:18].tolist() == [2, 2] and g[3:5].tolist() == [1, 1]


# ---------------------------------------------------------------- pooling
def test_combiners():
  """embedding_combiners_test.py:22-35."""
  out = cpu.reduce_sum([0, 1, 1], [[1.0], [2.0], [3.0]], 2)
  assert out.tolist() == [[1], [5]]
  out = cpu.first_n([0, 1, 1, 2, 2], [[1.0], [2.0], [3.0], [4.0], [5.0]], 3, 2)
  assert out.tolist() == [[[1], [0]], [[2], [3]], [[4], [5]]]
  out = cpu.first_n([0, 0, 0], [[1.0], [2.0], [3.0]], 1, 2)
  assert out.tolist() == [[[1], [2]]]


def test_reduce_mean_and_grads():
  """reduce_op.cc:59-86,128-179: mean = sum * (1/count); empty row -> 0 * inf = NaN."""
  v = np.array([[1, 2], [3, 4], [5, 7]], np.float32)
  out = cpu.reduce_mean([0, 0, 2], v, 3)
  assert out[0].tolist() == [2, 3] and np.isnan(out[1]).all() and out[2].tolist() == [5, 7]
  g = np.array([[1, 1], [2, 2], [4, 4]], np.float32)
  assert cpu.reduce_sum_gradient([0, 0, 2], g).tolist() == [[1, 1], [1, 1], [4, 4]]
  assert cpu.reduce_mean_gradient([0, 0, 2], g).tolist() == [[.5, .5], [.5, .5], [4, 4]]


def test_reduce_sum_and_split():
  """reduce_op.cc:232-320."""
  v = np.arange(12, dtype=np.float32).reshape(4, 3)
  outs = cpu.fused_reduce_sum_and_split([0, 0, 1, 1], v, 2, [1, 2])
  assert outs[0].tolist() == [[3], [15]] and outs[1].tolist() == [[5, 7], [17, 19]]
  g = cpu.fused_reduce_sum_and_split_gradient([0, 0, 1, 1], [np.array([[1.], [2.]]),
                                                            np.array([[3., 4.], [5., 6.]])], [1, 2])
  assert g.tolist() == [[1, 3, 4], [1, 3, 4], [2, 5, 6], [2, 5, 6]]
  outs = cpu.fused_sorted_segment_sum([[0, 0, 2]], [np.array([[1., 1.], [2., 2.], [3., 3.]])], [3])
  assert outs[0].tolist() == [[3, 3], [0, 0], [3, 3]]
