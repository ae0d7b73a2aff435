"""GPU parity of the fused multi-table lookup / optimize ops (MonolithHashTableFusedLookup /
FusedOptimize): reference test vectors through the Python op surface, then a seeded multi-table
multi-shard stream against the oracle, bit-exact."""
import numpy as np
import pytest
import torch

from oracle import cpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hto():
  from monolith_amd import hash_table_ops
  return hash_table_ops


def dev():
  return torch.device("cuda", 0)


def _mk(hto, dim, ids, vals, lr=1.0):
  t = hto.test_hash_table(dim, learning_rate=lr)
  t.assign(torch.tensor(ids, dtype=torch.int64, device=dev()),
           torch.tensor(np.asarray(vals, np.float32).reshape(len(ids), dim), device=dev()))
  return t


def test_fused_lookup_reference_vectors(hto):
  """hash_table_ops.py:372-383 docstring, hash_table_ops_test.py:1086-1109,
  multi_type_hash_table_test.py:156-198."""
  t0 = _mk(hto, 1, [1, 2], [[1], [2]])
  t1 = _mk(hto, 2, [3, 4], [[3, 3], [4, 4]])
  emb, splits, id_off, emb_off, sizes = hto.fused_lookup([t0, t1], [1, 3, 2, 4], [1, 1, 1, 1], 2)
  assert emb.cpu().tolist() == [1, 3, 3, 2, 4, 4] and splits.tolist() == [3, 3]
  assert id_off.tolist() == [0, 1, 2, 3] and emb_off.tolist() == [0, 1, 3, 4]
  assert sizes.tolist() == [1, 2, 1, 2]

  tabs = []
  for x, d in enumerate([1, 1, 2]):
    v = np.ones((2, d), np.float32) if x % 2 == 0 else np.zeros((2, d), np.float32)
    tabs.append(_mk(hto, d, [0 + 3 * x, 1 + 3 * x], v))
  emb, splits, id_off, emb_off, sizes = hto.fused_lookup(tabs, [0, 4, 6, 1, 3, 7], [1] * 6, 2)
  assert emb.cpu().tolist() == [1, 0, 1, 1, 1, 0, 1, 1] and splits.tolist() == [4, 4]
  assert id_off.tolist() == [0, 1, 2, 3, 4, 5] and emb_off.tolist() == [0, 1, 2, 4, 5, 6]
  assert sizes.tolist() == [1, 1, 2, 1, 1, 2]

  t0 = _mk(hto, 1, [0], [[1]]); t1 = _mk(hto, 2, [1], [[2, 2]])
  t2 = _mk(hto, 2, [2, 3], [[4, 4], [8, 8]])
  emb, splits, id_off, emb_off, sizes = hto.fused_lookup([t0, t1, t2], [0, 1, 2, 3], [1, 1, 2], 1)
  assert emb.cpu().tolist() == [1, 2, 2, 4, 4, 8, 8] and splits.tolist() == [7]
  assert id_off.tolist() == [0, 1, 2] and emb_off.tolist() == [0, 1, 3] and sizes.tolist() == [1, 2, 4]
  emb, splits, id_off, emb_off, sizes = hto.fused_lookup([t0, t1, t2], [0, 2, 1, 3],
                                                         [1, 0, 1, 0, 1, 1], 2)
  assert emb.cpu().tolist() == [1, 4, 4, 2, 2, 8, 8] and splits.tolist() == [3, 4]
  assert id_off.tolist() == [0, 1, 1, 2, 2, 3] and emb_off.tolist() == [0, 1, 1, 3, 3, 5]
  assert sizes.tolist() == [1, 0, 2, 0, 2, 2]


def test_fused_optimize_reference_vectors(hto):
  """hash_table_ops_test.py:1111-1153 and multi_type_hash_table_test.py:200-246."""
  t0 = _mk(hto, 1, [0, 1], np.ones((2, 1))); t1 = _mk(hto, 2, [3, 4], np.zeros((2, 2)))
  ids = [0, 4, 1, 3]
  _, _, id_off, emb_off, _ = hto.fused_lookup([t0, t1], ids, [1, 1, 1, 1], 2)
  hto.fused_apply_gradient([t0, t1], ids, [1, 1, 1, 1],
                           torch.tensor([-1, -2, -2, -1, -2, -2], dtype=torch.float32, device=dev()),
                           id_off, emb_off, [0.1, 0.1], [1, 1], 0, 0, 2)
  emb, splits, *_ = hto.fused_lookup([t0, t1], ids, [1, 1, 1, 1], 2)
  np.testing.assert_allclose(emb.cpu().numpy(), [1.1, 0.2, 0.2, 1.1, 0.2, 0.2], rtol=1e-6)
  assert splits.tolist() == [3, 3]

  a = hto.test_hash_table(1); b = hto.test_hash_table(2)
  _, _, id_off, emb_off, _ = hto.fused_lookup([a, b], [0, 1, 2], [1, 2], 1)
  hto.fused_apply_gradient([a, b], [0, 1, 2], [1, 2],
                           torch.tensor([2.0, 1.0, 3.0, 2.0, 4.0], device=dev()), id_off, emb_off,
                           [1.0, 1.0], [1, 1], 0, 0, 1)
  emb, *_ = hto.fused_lookup([a, b], [0, 1, 2], [1, 2], 1)
  assert emb.cpu().tolist() == [-2, -1, -3, -2, -4]

  a = hto.test_hash_table(1); b = hto.test_hash_table(2)
  fss = [1, 0, 1, 0]
  _, _, id_off, emb_off, _ = hto.fused_lookup([a, b], [1, 1], fss, 2)
  hto.fused_apply_gradient([a, b], [1, 1], fss, torch.tensor([1.0, 2.0], device=dev()), id_off,
                           emb_off, [1.0, 1.0], [1, 1], 0, 0, 2)
  emb, splits, id_off, emb_off, sizes = hto.fused_lookup([a, b], [1, 1], fss, 2)
  assert emb.cpu().tolist() == [-3, -3] and splits.tolist() == [1, 1]
  assert id_off.tolist() == [0, 1, 1, 2] and emb_off.tolist() == [0, 1, 1, 2]
  assert sizes.tolist() == [1, 0, 1, 0]


@pytest.mark.parametrize("accum", [False, True])
def test_fused_stream_vs_oracle(hto, accum):
  """K tables x N shards with empty cells and duplicate ids, three optimizers: embeddings,
  metadata and the table state after FusedOptimize are bit-identical to the oracle."""
  from monolith_amd import entry
  rng = np.random.Generator(np.random.PCG64(17 + int(accum)))
  specs = [(4, "sgd", entry.SgdOptimizer()),
           (16, "adagrad", entry.AdagradOptimizer(initial_accumulator_value=0.1)),
           (8, "ftrl", entry.FtrlOptimizer(initial_accumulator_value=0.1, beta=1.0,
                                           l1_regularization=0.01, l2_regularization=0.1))]
  K, N = len(specs), 3
  hip, orc = [], []
  for i, (d, oname, opt) in enumerate(specs):
    tc = entry.EmbeddingHashTableConfig(
        segments=[entry.CombineAsSegment(d, entry.ZerosInitializer(), opt)])
    hip.append(hto.hash_table_from_config(entry.HashTableConfigInstance(tc, [0.05]),
                                          name_suffix="fs_%d_%d" % (i, int(accum))))
    ocfg = dict(init_acc=0.1) if oname == "adagrad" else {}
    if oname == "ftrl":
      ocfg = dict(init_acc=0.1, beta=1.0, l1=0.01, l2=0.1)
    orc.append(cpu.OracleTable([cpu.seg(d, oname, **ocfg)]))
  dims = [s[0] for s in specs]
  for step in range(4):
    fss = rng.integers(0, 400, K * N)
    fss[rng.integers(0, K * N)] = 0
    ids = np.concatenate([(rng.integers(0, 150, int(c)) * 7 + (cell % K)) for cell, c in
                          enumerate(fss)]).astype(np.int64)
    emb, splits, id_off, emb_off, sizes = hto.fused_lookup(hip, ids, fss, N)
    e_emb, e_splits, e_id_off, e_emb_off, e_sizes = cpu.fused_lookup(orc, ids, fss, N)
    assert np.array_equal(emb.cpu().numpy().view(np.uint32), e_emb.view(np.uint32))
    assert splits.tolist() == e_splits.tolist() and id_off.tolist() == e_id_off.tolist()
    assert emb_off.tolist() == e_emb_off.tolist() and sizes.tolist() == e_sizes.tolist()
    g = rng.normal(0, 0.1, len(e_emb)).astype(np.float32)
    lrs = [0.05] * K
    hto.fused_apply_gradient(hip, ids, fss, torch.tensor(g, device=dev()), id_off, emb_off, lrs,
                             [1] * K, 100 + step, step, N, enable_grad_accumulation=accum)
    cpu.fused_apply_gradient(orc, ids, fss, g, e_id_off, e_emb_off, lrs, [1] * K, 100 + step, step,
                             N, enable_grad_accumulation=accum)
  for h, o, d in zip(hip, orc, dims):
    hi, hr, ht = h.save_as_tensor()
    oi, orows, ots = o.export()
    order = np.argsort(hi.cpu().numpy(), kind="stable")
    oorder = np.argsort(oi, kind="stable")
    assert np.array_equal(hi.cpu().numpy()[order], oi[oorder])
    assert np.array_equal(hr.cpu().numpy()[order].view(np.uint32), orows[oorder].view(np.uint32))
    assert np.array_equal(ht.cpu().numpy()[order], ots[oorder])
