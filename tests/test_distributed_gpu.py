"""GPU, world size 2 on ONE device: the sync-training exchange with the real HIP ops and tables
on both ranks. torch.distributed's gloo backend carries the all-to-alls (host-staged: two RCCL
ranks cannot share a GPU); on a multi-GPU node the same code runs over RCCL (bench.py --gpus N).
Checked against a serial oracle model of every owner shard (source-rank-major application)."""
import os
import socket
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _batch(rank, step, shape="small"):
  if shape == "config4":  # BASELINE config 4's shape: one dim-64 Adagrad table, >= 1e5 ids per rank
    rng = np.random.Generator(np.random.PCG64(77000 + 1000 * step + rank))
    slot = rng.integers(0, 26, 120000).astype(np.int64)
    r = (40000 * rng.random(120000) ** 3).astype(np.int64)   # bench.py's power-law skew
    ids = {"criteo": (slot << 48) | r}
    grads = {"criteo": rng.normal(0, 0.1, (120000, 64)).astype(np.float32)}
    return ids, grads
  rng = np.random.Generator(np.random.PCG64(1000 * step + rank))
  ids = {"a": rng.integers(-400, 400, 3000).astype(np.int64),
         "b": (rng.integers(0, 25, 1200) * 3 + 1).astype(np.int64),
         "c": np.zeros(0, np.int64)}
  dims = {"a": 16, "b": 2, "c": 3}
  grads = {k: rng.normal(0, 0.5, (len(v), dims[k])).astype(np.float32) for k, v in ids.items()}
  return ids, grads


def _worker(rank, world, port, q, backend="gloo", shape="small", pipelined=0):
  try:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MONOLITH_SYNC_EMPTY_RANK0_PS_SHARD"] = "1"  # N <= 4: plain floormod either way
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
      os.environ["MONOLITH_AMD_FORCE_COLLECTIVE"] = "1"
      dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
      dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cpu
    from monolith_amd import distributed_ps_sync, entry, hash_table_ops, multi_type_hash_table

    def cfg(dim, opt):
      tc = entry.EmbeddingHashTableConfig(
          segments=[entry.CombineAsSegment(dim, entry.ZerosInitializer(), opt)])
      return tc
    opts = {"a": (16, entry.AdagradOptimizer(initial_accumulator_value=0.1), 0.1),
            "b": (2, entry.SgdOptimizer(), 0.5),
            "c": (3, entry.FtrlOptimizer(initial_accumulator_value=0.1, beta=0.0,
                                         l1_regularization=0.0, l2_regularization=0.0), 0.01)}
    if shape == "config4":
      opts = {"criteo": (64, entry.AdagradOptimizer(initial_accumulator_value=0.1), 0.01)}
    cfgs = {k: entry.HashTableConfigInstance(cfg(d, o), [lr]) for k, (d, o, lr) in opts.items()}
    # pipelined: the reference's enable_pipelined_fwda2a (1), plus the owner's lookup hint (2)
    qc = {"enable_pipelined_fwda2a": 1, "announce_owner_lookup": int(pipelined == 2)} if pipelined else None
    table = distributed_ps_sync.DistributedMultiTypeHashTableMpi(
        world, lambda idx: multi_type_hash_table.MultiTypeHashTable(
            cfgs, lambda slot, c: hash_table_ops.hash_table_from_config(
                c, name_suffix="dg_%s_%d" % (slot, idx))), queue_configs=qc)
    segs = {"a": cpu.seg(16, "adagrad"), "b": cpu.seg(2, "sgd"), "c": cpu.seg(3, "ftrl")}
    if shape == "config4":
      segs = {"criteo": cpu.seg(64, "adagrad")}
    lrs = {k: v[2] for k, v in opts.items()}
    model = [{k: cpu.OracleTable([v]) for k, v in segs.items()} for _ in range(world)]
    n_steps = (4 if pipelined else 3) if shape == "small" else 2
    for step in range(n_steps):
      ids, grads = _batch(rank, step, shape)
      slot_to_ids = {k: torch.from_numpy(v).to(dev) for k, v in ids.items()}
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
        assert np.array_equal(emb[k].cpu().numpy().view(np.uint32), exp.view(np.uint32)), (step, k)
      nxt = None
      if pipelined and step + 1 < n_steps:
        nxt = {k: torch.from_numpy(v).to(dev) for k, v in _batch(rank, step + 1, shape)[0].items()}
      table = table.apply_gradients(
          {k: (slot_to_ids[k], torch.from_numpy(grads[k]).to(dev)) for k in ids},
          auxiliary_bundle=bundle, global_step=step, req_time=100 + step, next_slot_to_id=nxt)
      for n in range(world):
        for src in range(world):
          s_ids, s_grads = _batch(src, step, shape)
          for k in sorted(s_ids):
            v, g = s_ids[k], s_grads[k]
            sel = np.mod(v, world) == n
            v, g = v[sel], g[sel]
            uniq, first, inv = np.unique(v, return_index=True, return_inverse=True)
            by_first = np.argsort(first)
            order = uniq[by_first]
            rank_of = np.empty(len(uniq), np.int64)
            rank_of[by_first] = np.arange(len(uniq))
            # the requesting rank sums duplicates in input order, first term copied (F2 gradient):
            # ufunc.at is unbuffered and walks its operands in order
            summed = np.zeros((len(order), g.shape[1]), np.float32)
            if len(v):
              summed[rank_of[inv[first]]] = g[first]
              rest = np.ones(len(v), bool)
              rest[first] = False
              np.add.at(summed, rank_of[inv[rest]], g[rest])
            model[n][k].apply_gradients(order, summed, [lrs[k]], req_time=100 + step,
                                        global_step=step)
    for k, ht in table._table._hash_tables.items():
      gi, gr, gt = [x.cpu().numpy() for x in ht.save_as_tensor()]
      ei, er, et = model[rank][k].export()
      go, eo = np.argsort(gi, kind="stable"), np.argsort(ei, kind="stable")
      assert np.array_equal(gi[go], ei[eo]), k
      assert np.array_equal(gr[go].view(np.uint32), er[eo].view(np.uint32)), k
      assert np.array_equal(gt[go], et[eo]), k
    if pipelined == 2 and shape == "config4":  # one table: the announced lookups were served as such
      import ctypes as C
      from monolith_amd import _lib
      counts = (C.c_int64 * 4)()
      h = table._table._hash_tables["criteo"].table.h
      _lib.check(_lib.lib.mono_table_path_counts(h, counts))
      assert counts[2] >= 1, list(counts)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, None))
  except Exception:
    q.put((rank, traceback.format_exc()))


def test_world2_exchange_on_hip_ops():
  world = 2
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
  for p in ps:
    p.start()
  res = [q.get(timeout=150) for _ in ps]
  for p in ps:
    p.join(timeout=30)
  for rank, err in res:
    assert err is None, "rank %d failed:\n%s" % (rank, err)


@pytest.mark.parametrize("pipelined", [1, 2], ids=["fwda2a", "fwda2a+hint"])
def test_world2_exchange_pipelined_forward(pipelined):
  """queue_configs["enable_pipelined_fwda2a"] (distributed_ps_sync.py:218-224): the next batch's
  reorder, size exchange and FID exchange (and, variant 2, the owner's lookup hint) run on a side
  stream beside the update; lookup() continues from the ForwardPrefetch. Rows and the owner
  shards' final state equal the serial oracle model bit for bit — nothing is stale."""
  world = 2
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  ps = [ctx.Process(target=_worker, args=(r, world, port, q, "gloo", "small", pipelined))
        for r in range(world)]
  for p in ps:
    p.start()
  res = [q.get(timeout=150) for _ in ps]
  for p in ps:
    p.join(timeout=30)
  for rank, err in res:
    assert err is None, "rank %d failed:\n%s" % (rank, err)


def test_world1_pipelined_forward_over_rccl():
  """The pipelined forward with backend "nccl" (= RCCL) at world size 1, exchanges forced through
  dist.all_to_all_single: the size and FID collectives are issued under the side stream, the
  gradient and row collectives under the main one, as at N > 1. One table (config 4's shape), so
  the owner's lookup hint takes effect."""
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  p = ctx.Process(target=_worker, args=(0, 1, _free_port(), q, "nccl", "config4", 2))
  p.start()
  rank, err = q.get(timeout=250)
  p.join(timeout=30)
  assert err is None, "rccl world-1 pipelined run failed:\n%s" % err


def test_world2_exchange_config4_shape():
  """BASELINE config 4's table (dim 64, Adagrad, bench.py's skewed FIDs, 1.2e5 ids per rank and
  step) through the world-2 exchange: embeddings and the owner shards' final state against the
  serial oracle model, bit for bit. (The N = 8 routing of the same ids, both shard rules, is
  checked against the oracle in tests/test_ops_gpu.py::test_reorder_config4_routing.)"""
  world = 2
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  ps = [ctx.Process(target=_worker, args=(r, world, port, q, "gloo", "config4")) for r in range(world)]
  for p in ps:
    p.start()
  res = [q.get(timeout=280) for _ in ps]
  for p in ps:
    p.join(timeout=30)
  for rank, err in res:
    assert err is None, "rank %d failed:\n%s" % (rank, err)


def test_world1_exchange_over_rccl():
  """The same step with backend "nccl" (= RCCL) at world size 1, the exchange forced through
  dist.all_to_all_single (MONOLITH_AMD_FORCE_COLLECTIVE=1): device tensors, uneven split lists
  and the int64 / fp32 payloads go through RCCL exactly as they do at N > 1 on a multi-GPU node
  (which this pool's one-GPU box cannot host)."""
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  p = ctx.Process(target=_worker, args=(0, 1, _free_port(), q, "nccl"))
  p.start()
  rank, err = q.get(timeout=150)
  p.join(timeout=30)
  assert err is None, "rccl world-1 run failed:\n%s" % err


def test_reorder_one_batch_ahead_on_a_side_stream():
  """fused_reorder_by_indices_begin on a side stream while another batch is looked up and
  updated on the main stream (bench.py's pipelined step): the pack handed to lookup as
  early_reorder_indicies_res_pack gives the rows and the table state of the in-line order."""
  from monolith_amd import distributed_ps_sync, entry, hash_table_ops, multi_type_hash_table
  dev = torch.device("cuda", 0)

  def build(tag):
    tc = entry.EmbeddingHashTableConfig(
        segments=[entry.CombineAsSegment(16, entry.ZerosInitializer(),
                                         entry.AdagradOptimizer(initial_accumulator_value=0.1))],
        initial_capacity=1 << 16, seed=5)
    cfgs = {"s": entry.HashTableConfigInstance(tc, [0.05])}
    fac = lambda idx: multi_type_hash_table.MultiTypeHashTable(
        cfgs, lambda slot, c: hash_table_ops.hash_table_from_config(c, name_suffix="%s_%s" % (tag, slot)))
    return distributed_ps_sync.DistributedMultiTypeHashTableMpi(1, fac)

  rng = np.random.Generator(np.random.PCG64(77))
  batches = [torch.tensor((3000 * rng.random(40000) ** 3).astype(np.int64) * 7919, device=dev)
             for _ in range(5)]
  grads = [torch.tensor(rng.normal(0, 0.1, (40000, 16)).astype(np.float32), device=dev)
           for _ in range(5)]
  outs = {}
  for mode in ("inline", "ahead"):
    t = build(mode)
    res = []
    for i, (b, g) in enumerate(zip(batches, grads)):
      emb, bundle = t.lookup({"s": b}, {}, early_reorder_indicies_res_pack=t.take_prefetched_reorder())
      res.append(emb["s"].clone())
      nxt = {"s": batches[i + 1]} if mode == "ahead" and i + 1 < len(batches) else None
      t = t.apply_gradients({"s": (b, g)}, bundle, global_step=i, req_time=100 + i,
                            next_slot_to_id=nxt)
    torch.cuda.synchronize()
    outs[mode] = [r.cpu().numpy() for r in res]
  for a, b in zip(outs["inline"], outs["ahead"]):
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
