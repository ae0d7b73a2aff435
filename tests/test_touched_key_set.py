"""Touched-key set. CPU: the oracle against the reference's two test vectors
(touched_key_set_ops_test.py:24-50, hopscotch_hash_set_test.cc). GPU: the HIP set against the
oracle on random streams that overflow inside and between batches, and the hook that records the
ids an optimizer step was applied to (embedding_hash_table_tf_bridge.cc:303-307)."""
import numpy as np
import pytest

from oracle import cpu


def test_oracle_reference_vectors():
  s = cpu.OracleKeySet(1000)
  assert s.insert(range(1000)) == 0
  assert s.steal() == list(range(1000))
  s = cpu.OracleKeySet(1000)
  assert s.insert(range(1005)) == 1001
  assert s.steal() == [1001, 1002, 1003, 1004]
  assert s.steal() == []


@pytest.mark.gpu
def test_reference_vectors_on_gpu():
  import torch
  from monolith_amd.touched_key_set_ops import TouchedKeySet
  s = TouchedKeySet(1000, 1)
  assert s.insert(torch.arange(1000)) == 0
  assert sorted(s.steal().cpu().tolist()) == list(range(1000))
  s = TouchedKeySet(1000, 1)
  assert s.insert(torch.arange(1005)) == 1001
  assert sorted(s.steal().cpu().tolist()) == [1001, 1002, 1003, 1004]
  assert s.steal().numel() == 0 and s.size() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("capacity,vocab,batch", [(50, 400, 300), (1000, 3000, 700), (5, 9, 64),
                                                   (100000, 5000, 4000)])
def test_random_streams_match_oracle(capacity, vocab, batch):
  import torch
  from monolith_amd.touched_key_set_ops import TouchedKeySet
  rng = np.random.Generator(np.random.PCG64(capacity + batch))
  o, h = cpu.OracleKeySet(capacity), TouchedKeySet(capacity)
  for step in range(12):
    ids = rng.integers(-vocab, vocab, batch).astype(np.int64)
    if step == 4:
      ids[::5] = np.iinfo(np.int64).min  # the cell sentinel itself is a legal key
      ids[1::7] = -1
    if step == 7:
      ids = np.zeros(0, np.int64)
    assert h.insert(torch.from_numpy(ids)) == o.insert(ids.tolist()), step
    assert h.size() == len(o.keys)
    if step % 5 == 4:
      assert sorted(h.steal().cpu().tolist()) == o.steal()
  assert sorted(h.steal().cpu().tolist()) == o.steal()


@pytest.mark.gpu
@pytest.mark.timeout(120)
@pytest.mark.parametrize("capacity,batch", [(1000, 6000), (1000, 50000), (3, 5000)])
def test_batches_far_beyond_the_set_size(capacity, batch):
  """A batch with several times more DISTINCT ids than the set has cells (2048 cells at capacity
  1000): one round used to insert all of them before the overflow position was known, filled the
  table and spun for ever. Rounds are now bounded by the free cells; drop positions and counts stay
  the reference's (hopscotch_hash_set.cc:104-166 replayed by the oracle)."""
  import torch
  from monolith_amd.touched_key_set_ops import TouchedKeySet
  rng = np.random.Generator(np.random.PCG64(7 * capacity + batch))
  o, h = cpu.OracleKeySet(capacity), TouchedKeySet(capacity)
  for step in range(4):
    ids = rng.permutation(batch * 4)[:batch].astype(np.int64) - batch  # all distinct
    if step == 2:
      ids[::3] = ids[0]  # duplicates in between
    assert h.insert(torch.from_numpy(ids)) == o.insert(ids.tolist()), step
    assert h.size() == len(o.keys)
  assert sorted(h.steal().cpu().tolist()) == o.steal()


@pytest.mark.gpu
def test_table_records_applied_ids():
  """Ids that pass the occurrence filter and are optimized land in the set; filtered ids and
  ids touched only by assign / lookup do not."""
  import torch
  from monolith_amd import hash_table_ops as ops
  from monolith_amd.touched_key_set_ops import TouchedKeySet
  keys = TouchedKeySet(1000)
  t = ops.test_hash_table(2, enable_hash_filter=True, occurrence_threshold=2,
                          filter_capacity=100000)
  t.set_touched_key_set(keys)
  dev = "cuda:0"
  ids = torch.tensor([5, 6, 5, 7], device=dev)
  g = torch.ones(4, 2, device=dev)
  t = t.assign(torch.tensor([100], device=dev), torch.ones(1, 2, device=dev))
  t.lookup(torch.tensor([100, 5], device=dev))
  assert keys.size() == 0
  t = t.apply_gradients(ids, g, 0)        # nothing admitted yet (threshold 2, counts 0 / 1)
  assert sorted(keys.steal().cpu().tolist()) == []
  t = t.apply_gradients(ids, g, 0)        # 5 reached the threshold on its 3rd occurrence
  assert sorted(keys.steal().cpu().tolist()) == [5]
  t = t.apply_gradients(ids, g, 0)        # 6 and 7 follow; 5 is in the table and bypasses the filter
  assert sorted(keys.steal().cpu().tolist()) == [5, 6, 7]
  # 100: the gated assign counted it once, this step counts it twice -> admitted on the next one
  one = torch.ones(1, 2, device=dev)
  t = t.apply_gradients(torch.tensor([100], device=dev), one, 0)
  assert keys.steal().cpu().tolist() == []
  t = t.apply_gradients(torch.tensor([100], device=dev), one, 0)
  assert keys.steal().cpu().tolist() == [100]
