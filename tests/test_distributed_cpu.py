"""World-size-2 gloo tests of the sync-training exchange (monolith_amd.distributed_ps_sync): the
host-side orchestration (split computation, four all-to-alls, offset bookkeeping) runs for real;
the per-rank local ops are the oracle (no GPU in this container). The same class runs with the
HIP ops and RCCL on the GPUs (bench.py --gpus N, tests/test_distributed_gpu.py)."""
import os
import socket
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _run(rank, world, port, fn, q):
  try:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fn(rank, world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, None))
  except Exception:
    q.put((rank, traceback.format_exc()))


def _spawn(fn, world=2):
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  ps = [ctx.Process(target=_run, args=(r, world, port, fn, q)) for r in range(world)]
  for p in ps:
    p.start()
  res = [q.get(timeout=120) for _ in ps]
  for p in ps:
    p.join(timeout=30)
  for rank, err in res:
    assert err is None, "rank %d failed:\n%s" % (rank, err)


def _table_factory(slot_to_seg, lrs):
  from tests.dist_oracle_ops import OracleMultiTable
  return lambda idx: OracleMultiTable(slot_to_seg, lrs)


def _basic(rank, world):
  """distributed_ps_sync_test.py:66-104: SGD lr 1; every rank sends grads 0.5+0.5 for id 1 and
  [0.5, 1.0] for id 2 -> after one step the rows are -1*world and [-0.5*world, -1*world]."""
  from oracle import cpu
  from monolith_amd import distributed_ps_sync
  from tests.dist_oracle_ops import OracleOps
  segs = {"1": cpu.seg(1, "sgd"), "2": cpu.seg(2, "sgd")}
  table = distributed_ps_sync.DistributedMultiTypeHashTableMpi(
      world, _table_factory(segs, {"1": 1.0, "2": 1.0}), ops=OracleOps)
  slot_to_ids = {"1": torch.tensor([1, 1], dtype=torch.int64),
                 "2": torch.tensor([2], dtype=torch.int64)}
  emb, bundle = table.lookup(slot_to_ids, auxiliary_bundle={})
  assert emb["1"].tolist() == [[0], [0]] and emb["2"].tolist() == [[0, 0]]
  updated = table.apply_gradients(
      {"1": (slot_to_ids["1"], torch.tensor([[0.5], [0.5]])),
       "2": (slot_to_ids["2"], torch.tensor([[0.5, 1.0]]))},
      auxiliary_bundle=bundle, global_step=0, req_time=0)
  emb, bundle = updated.lookup(slot_to_ids, auxiliary_bundle={})
  assert emb["1"].tolist() == [[-1.0 * world]] * 2
  assert emb["2"].tolist() == [[-0.5 * world, -1.0 * world]]


def _batch(rank, step):
  rng = np.random.Generator(np.random.PCG64(1000 * step + rank))
  ids = {"a": rng.integers(-40, 40, 300).astype(np.int64),
         "b": (rng.integers(0, 25, 120) * 3 + 1).astype(np.int64),
         "c": np.zeros(0, np.int64)}
  dims = {"a": 4, "b": 2, "c": 3}
  grads = {k: rng.normal(0, 0.5, (len(v), dims[k])).astype(np.float32) for k, v in ids.items()}
  return ids, grads


def _random_fp16(rank, world):
  """The same stream with the gradient rows exchanged as binary16 (the reference's
  MONOLITH_WITH_BYTEPS_BWD_CAST=16, distributed_ps_sync.py:353-356): every per-rank in-order sum
  is rounded to the nearest half (ties to even) before the owner applies it."""
  _random(rank, world, grad_bits=16)


def _random_pipelined(rank, world):
  """The same stream with the reference's enable_pipelined_fwda2a switched on
  (distributed_ps_sync.py:218-224): the first half of every next lookup — reorder, size exchange,
  FID exchange — runs inside apply_gradients(next_slot_to_id=...), right after the update is
  issued, and lookup() continues from it. Rows are still read after the update, so the
  serial model must hold bit for bit (the reference's queues would serve stale rows here)."""
  _random(rank, world, pipelined=True)


def _random(rank, world, grad_bits=32, pipelined=False):
  """Each rank draws its own batch (duplicates inside a rank and across ranks, an empty slot,
  negative ids -> signed floor-mod routing). Lookup results and the owner-side table state after
  each step equal a serial simulation: per owner shard, contributions applied source-rank-major,
  one optimizer step per (source rank, unique id) on the in-order sum of that rank's duplicates."""
  from oracle import cpu
  from monolith_amd import distributed_ps_sync
  from tests.dist_oracle_ops import OracleOps
  mk = lambda: {"a": cpu.seg(4, "adagrad"), "b": cpu.seg(2, "sgd"), "c": cpu.seg(3, "ftrl")}
  lrs = {"a": 0.1, "b": 0.5, "c": 0.01}
  table = distributed_ps_sync.DistributedMultiTypeHashTableMpi(
      world, _table_factory(mk(), lrs), ops=OracleOps, grad_exchange_bits=grad_bits,
      queue_configs={"enable_pipelined_fwda2a": 1} if pipelined else None)
  # serial model of every owner shard (each rank simulates all of them)
  model = [{k: cpu.OracleTable([v]) for k, v in mk().items()} for _ in range(world)]
  for step in range(3):
    ids, grads = _batch(rank, step)
    slot_to_ids = {k: torch.from_numpy(v) for k, v in ids.items()}
    ahead = table.take_prefetched_reorder()
    if pipelined and step > 0:
      assert isinstance(ahead, distributed_ps_sync.ForwardPrefetch)
    emb, bundle = table.lookup(slot_to_ids, auxiliary_bundle={},
                               early_reorder_indicies_res_pack=ahead)
    for k, v in ids.items():
      exp = np.zeros((len(v), model[0][k].dim), np.float32)
      for n in range(world):
        sel = np.mod(v, world) == n
        if sel.any():
          exp[sel] = model[n][k].lookup(v[sel])
      assert np.array_equal(emb[k].numpy().view(np.uint32), exp.view(np.uint32)), (step, k)
    nxt = {k: torch.from_numpy(v) for k, v in _batch(rank, step + 1)[0].items()} if pipelined else None
    table = table.apply_gradients(
        {k: (slot_to_ids[k], torch.from_numpy(grads[k])) for k in ids}, auxiliary_bundle=bundle,
        global_step=step, req_time=100 + step, next_slot_to_id=nxt)
    for n in range(world):
      for src in range(world):
        s_ids, s_grads = _batch(src, step)
        for k in sorted(s_ids):
          v, g = s_ids[k], s_grads[k]
          sel = np.mod(v, world) == n
          v, g = v[sel], g[sel]
          uniq, first = np.unique(v, return_index=True)
          order = uniq[np.argsort(first)]  # first-occurrence order
          summed = np.zeros((len(order), g.shape[1]), np.float32)
          for i, u in enumerate(order):
            acc = np.zeros(g.shape[1], np.float32)
            for row in g[v == u]:
              acc = acc + row
            summed[i] = acc
          if grad_bits == 16:
            summed = summed.astype(np.float16).astype(np.float32)
          model[n][k].apply_gradients(order, summed, [lrs[k]], req_time=100 + step,
                                      global_step=step)
    mine = table._table.tables
    for k in mine:
      gi, gr, gt = mine[k].export()
      ei, er, et = model[rank][k].export()
      go, eo = np.argsort(gi, kind="stable"), np.argsort(ei, kind="stable")
      assert np.array_equal(gi[go], ei[eo]), (step, k)
      assert np.array_equal(gr[go].view(np.uint32), er[eo].view(np.uint32)), (step, k)
      assert np.array_equal(gt[go], et[eo])


def _merged(rank, world):
  """MergedMultiTypeHashTable over the distributed table: two slots with the same config share
  one merged table (multi_type_hash_table_test.py:255-311), exchanged across 2 ranks."""
  from oracle import cpu
  from monolith_amd import distributed_ps_sync, entry, multi_type_hash_table
  from tests.dist_oracle_ops import OracleMultiTable, OracleOps
  tc = entry.EmbeddingHashTableConfig(
      segments=[entry.CombineAsSegment(2, entry.ZerosInitializer(), entry.SgdOptimizer())])
  cfgs = {"slot0": entry.HashTableConfigInstance(tc, [1.0]),
          "slot1": entry.HashTableConfigInstance(tc, [1.0])}

  def factory(merged_cfgs):
    assert list(merged_cfgs) == ["e21904dd414d1780e5fc904866dc69c2"]
    segs = {k: cpu.seg(2, "sgd") for k in merged_cfgs}
    return distributed_ps_sync.DistributedMultiTypeHashTableMpi(
        world, lambda idx: OracleMultiTable(segs, {k: 1.0 for k in merged_cfgs}), ops=OracleOps)

  table = multi_type_hash_table.MergedMultiTypeHashTable(cfgs, factory)
  ids = {"slot0": torch.tensor([0, 3], dtype=torch.int64),
         "slot1": torch.tensor([5], dtype=torch.int64)}
  emb, bundle = table.lookup(ids, auxiliary_bundle={})
  assert emb["slot0"].shape == (2, 2) and emb["slot1"].shape == (1, 2)
  table = table.apply_gradients(
      {"slot0": (ids["slot0"], torch.tensor([[1.0, 1.0], [2.0, 2.0]])),
       "slot1": (ids["slot1"], torch.tensor([[3.0, 3.0]]))}, auxiliary_bundle=bundle,
      global_step=0, req_time=0)
  emb, _ = table.lookup(ids, auxiliary_bundle={})
  assert emb["slot0"].tolist() == [[-1.0 * world] * 2, [-2.0 * world] * 2]
  assert emb["slot1"].tolist() == [[-3.0 * world] * 2]


@pytest.mark.parametrize("fn", [_basic, _random, _random_fp16, _random_pipelined, _merged],
                         ids=lambda f: f.__name__)
def test_world2_gloo(fn):
  _spawn(fn, 2)


def test_world3_gloo_pipelined():
  """Three ranks (not a power of two: floormod routing of negative ids over an odd modulus), the
  pipelined forward switched on."""
  _spawn(_random_pipelined, 3)


def test_world1_no_process_group():
  """shard_num 1 without torch.distributed initialised: the exchanges are identities."""
  _basic(0, 1)
