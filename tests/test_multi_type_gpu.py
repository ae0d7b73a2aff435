"""GPU: slot-keyed table composition and the sync-training exchange class on the HIP ops —
the reference's multi_type_hash_table_test.py / distributed_ps_sync_test.py vectors."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_n = [0]


def dev():
  return torch.device("cuda", 0)


def _id(x):
  return torch.tensor(x, dtype=torch.int64, device=dev())


def _value(x):
  return torch.tensor(x, dtype=torch.float32, device=dev())


def _cfg(dim, learning_rate=1.0):
  """test_utils.generate_test_hash_table_config (test_utils.py:23-35)."""
  from monolith_amd import entry
  tc = entry.EmbeddingHashTableConfig(segments=[
      entry.CombineAsSegment(dim, entry.ZerosInitializer(), entry.SgdOptimizer(),
                             entry.Fp32Compressor())])
  return entry.HashTableConfigInstance(tc, [learning_rate])


def factory(idx, config):
  from monolith_amd import hash_table_ops
  _n[0] += 1
  return hash_table_ops.hash_table_from_config(config=config, name_suffix="%s_%d" % (idx, _n[0]))


def _multi_type_factory(slot_to_config):
  from monolith_amd import multi_type_hash_table
  return multi_type_hash_table.MultiTypeHashTable(slot_to_config, factory)


def test_multi_type_basic_and_gradients():
  """multi_type_hash_table_test.py:43-154."""
  from monolith_amd import learning_rate_functions, multi_type_hash_table
  t = multi_type_hash_table.MultiTypeHashTable({"slot0": _cfg(1), "slot1": _cfg(2),
                                                "slot2": _cfg(2)}, factory)
  t = t.assign_add({"slot0": (_id([0]), _value([[1]])), "slot1": (_id([1]), _value([[2, 2]])),
                    "slot2": (_id([2, 3]), _value([[4, 4], [8, 8]]))})
  v = t.lookup({"slot0": _id([0]), "slot1": _id([1]), "slot2": _id([2, 3])})
  assert v["slot0"].cpu().tolist() == [[1]] and v["slot1"].cpu().tolist() == [[2, 2]]
  assert v["slot2"].cpu().tolist() == [[4, 4], [8, 8]]

  t = multi_type_hash_table.MultiTypeHashTable({"slot0": _cfg(1), "slot1": _cfg(2)}, factory)
  t = t.apply_gradients({"slot0": (_id([0]), _value(2.0)),
                         "slot1": (_id([1, 2]), _value([[1.0, 3.0], [2.0, 4.0]]))}, 0)
  v = t.lookup({"slot0": _id([0]), "slot1": _id([1, 2])})
  assert v["slot0"].cpu().tolist() == [[-2]] and v["slot1"].cpu().tolist() == [[-1, -3], [-2, -4]]

  learning_rate_functions.set_global_step(0)
  mk = lambda: learning_rate_functions.PolynomialDecay(initial_learning_rate=0.1, decay_steps=10,
                                                       end_learning_rate=1.1)
  t = multi_type_hash_table.MultiTypeHashTable({"slot0": _cfg(1, mk()), "slot1": _cfg(2, mk())},
                                               factory)
  t = t.apply_gradients({"slot0": (_id([0]), _value(2.0)),
                         "slot1": (_id([1, 2]), _value([[1.0, 3.0], [2.0, 4.0]]))}, 0)
  v = t.lookup({"slot0": _id([0]), "slot1": _id([1, 2])})
  np.testing.assert_allclose(v["slot0"].cpu().numpy(), [[-0.2]], rtol=1e-6)
  np.testing.assert_allclose(v["slot1"].cpu().numpy(), [[-0.1, -0.3], [-0.2, -0.4]], rtol=1e-6)

  # ids [2, 2] without dedup -> two SGD steps
  t = multi_type_hash_table.MultiTypeHashTable({"slot0": _cfg(1), "slot1": _cfg(2)}, factory)
  t = t.apply_gradients({"slot0": (_id([1]), _value(3.0)),
                         "slot1": (_id([2, 2]), _value([[1.1, 3.1], [2.2, 4.2]]))}, 0)
  v = t.lookup({"slot0": _id([1]), "slot1": _id([1, 2])})
  np.testing.assert_allclose(v["slot0"].cpu().numpy(), [[-3]], rtol=1e-6)
  np.testing.assert_allclose(v["slot1"].cpu().numpy(), [[0, 0], [-3.3, -7.3]], rtol=1e-6)


def test_multi_type_fused_ops():
  """multi_type_hash_table_test.py:156-246 through MultiTypeHashTable.fused_*."""
  from monolith_amd import multi_type_hash_table
  t = multi_type_hash_table.MultiTypeHashTable({"slot0": _cfg(1), "slot1": _cfg(2),
                                                "slot2": _cfg(2)}, factory)
  t = t.assign_add({"slot0": (_id([0]), _value([[1]])), "slot1": (_id([1]), _value([[2, 2]])),
                    "slot2": (_id([2, 3]), _value([[4, 4], [8, 8]]))})
  emb, splits, id_off, emb_off, sizes = t.fused_lookup(_id([0, 1, 2, 3]), [1, 1, 2], 1)
  assert emb.cpu().tolist() == [1, 2, 2, 4, 4, 8, 8] and splits.tolist() == [7]
  emb, splits, id_off, emb_off, sizes = t.fused_lookup(_id([0, 2, 1, 3]), [1, 0, 1, 0, 1, 1], 2)
  assert emb.cpu().tolist() == [1, 4, 4, 2, 2, 8, 8] and splits.tolist() == [3, 4]
  assert emb_off.tolist() == [0, 1, 1, 3, 3, 5] and sizes.tolist() == [1, 0, 2, 0, 2, 2]

  t = multi_type_hash_table.MultiTypeHashTable({"slot0": _cfg(1), "slot1": _cfg(2)}, factory)
  ids = _id([0, 1, 2])
  _, _, id_off, emb_off, _ = t.fused_lookup(ids, [1, 2], 1)
  t = t.fused_apply_gradient(ids, [1, 2], _value([2.0, 1.0, 3.0, 2.0, 4.0]), id_off, emb_off, 0, 0,
                             1)
  emb, *_ = t.fused_lookup(ids, [1, 2], 1)
  assert emb.cpu().tolist() == [-2, -1, -3, -2, -4]


def test_merged_multi_type():
  """multi_type_hash_table_test.py:255-325: merged slots share one table; md5 names; fc_slot_N
  restore alias."""
  from monolith_amd import multi_type_hash_table
  t = multi_type_hash_table.MergedMultiTypeHashTable(
      {"slot0": _cfg(1), "slot1": _cfg(2), "slot2": _cfg(2)}, _multi_type_factory)
  u = t.assign_add({"slot0": (_id([0]), _value([[1]])), "slot1": (_id([1]), _value([[2, 2]])),
                    "slot2": (_id([1]), _value([[4, 4]]))})
  v = u.lookup({"slot0": _id([0]), "slot1": _id([1]), "slot2": _id([1])})
  assert v["slot0"].cpu().tolist() == [[1]]
  assert v["slot1"].cpu().tolist() == [[6, 6]] and v["slot2"].cpu().tolist() == [[6, 6]]
  u = t.apply_gradients({"slot0": (_id([0]), _value([[-1]])), "slot1": (_id([1]), _value([[1, 1]])),
                         "slot2": (_id([1]), _value([[1, 1]]))}, 0)
  v = u.lookup({"slot0": _id([0]), "slot1": _id([1])})
  assert v["slot0"].cpu().tolist() == [[2]] and v["slot1"].cpu().tolist() == [[4, 4]]

  seen = []

  def call(slot_to_config):
    seen.append(list(slot_to_config.keys()))
    return _multi_type_factory(slot_to_config)

  multi_type_hash_table.MergedMultiTypeHashTable({"slot0": _cfg(1), "slot1": _cfg(1)}, call)
  assert seen[-1] == ["e21904dd414d1780e5fc904866dc69c2"]

  def call2(slot_to_config):
    config = next(iter(slot_to_config.values()))
    assert config.extra_restore_names == [hashlib.md5("slot_0".encode()).hexdigest()]
    return _multi_type_factory(slot_to_config)

  multi_type_hash_table.MergedMultiTypeHashTable({"fc_slot_0": _cfg(1)}, call2)


def test_distributed_table_world1_hip():
  """distributed_ps_sync_test.py:66-104 with one rank: the whole exchange pipeline (reorder ->
  fused lookup -> gather; gather-grad -> fused optimize) on the HIP ops."""
  from monolith_amd import distributed_ps_sync, learning_rate_functions, multi_type_hash_table
  learning_rate_functions.set_global_step(0)
  cfgs = {"1": _cfg(1, 1.0),
          "2": _cfg(2, learning_rate_functions.PolynomialDecay(initial_learning_rate=1.0,
                                                               decay_steps=10,
                                                               end_learning_rate=2.0))}
  table = distributed_ps_sync.DistributedMultiTypeHashTableMpi(
      1, lambda idx: multi_type_hash_table.MultiTypeHashTable(
          cfgs, lambda s, c: factory(s + str(idx), c)))
  ids = {"1": _id([1, 1]), "2": _id([2])}
  emb, bundle = table.lookup(ids, auxiliary_bundle={})
  assert emb["1"].cpu().tolist() == [[0], [0]] and emb["2"].cpu().tolist() == [[0, 0]]
  updated = table.apply_gradients({"1": (ids["1"], _value([[0.5], [0.5]])),
                                   "2": (ids["2"], _value([[0.5, 1.0]]))},
                                  auxiliary_bundle=bundle, global_step=0, req_time=0)
  emb, bundle = updated.lookup(ids, auxiliary_bundle={})
  assert emb["1"].cpu().tolist() == [[-1.0], [-1.0]] and emb["2"].cpu().tolist() == [[-0.5, -1.0]]


def test_distributed_table_random_world1_vs_direct():
  """The exchange pipeline at world 1 equals applying the per-slot deduped, in-order-summed
  gradients directly to identical tables (bit-exact)."""
  from monolith_amd import distributed_ps_sync, entry, hash_table_ops, multi_type_hash_table
  rng = np.random.Generator(np.random.PCG64(31))

  def mk(dim):
    tc = entry.EmbeddingHashTableConfig(segments=[
        entry.CombineAsSegment(dim, entry.ZerosInitializer(),
                               entry.AdagradOptimizer(initial_accumulator_value=0.1))])
    return entry.HashTableConfigInstance(tc, [0.05])

  cfgs = {"a": mk(8), "b": mk(16)}
  table = distributed_ps_sync.DistributedMultiTypeHashTableMpi(
      1, lambda idx: multi_type_hash_table.MultiTypeHashTable(cfgs, factory))
  direct = {k: factory("direct_" + k, c) for k, c in cfgs.items()}
  for step in range(3):
    ids = {"a": rng.integers(0, 500, 4000).astype(np.int64),
           "b": rng.integers(0, 50, 3000).astype(np.int64) * 7}
    grads = {k: rng.normal(0, 0.1, (len(v), cfgs[k].table_config.dim_size)).astype(np.float32)
             for k, v in ids.items()}
    emb, bundle = table.lookup({k: _id(v) for k, v in ids.items()}, auxiliary_bundle={})
    for k in ids:
      exp = direct[k].lookup(_id(ids[k]))
      assert torch.equal(emb[k], exp)
    table = table.apply_gradients({k: (_id(ids[k]), _value(grads[k])) for k in ids},
                                  auxiliary_bundle=bundle, global_step=step, req_time=step)
    for k in ids:
      direct[k].apply_gradients(_id(ids[k]), _value(grads[k]), step, enable_dedup=True,
                                req_time=step)
  for k in cfgs:
    a = table._table._hash_tables[k].save_as_tensor()
    b = direct[k].save_as_tensor()
    oa, ob = torch.argsort(a[0]), torch.argsort(b[0])
    assert torch.equal(a[0][oa], b[0][ob]) and torch.equal(a[1][oa], b[1][ob])
    assert torch.equal(a[2][oa], b[2][ob])
