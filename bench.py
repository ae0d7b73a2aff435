"""bench.py — the sparse-embedding hot path on MI355X, BASELINE.json's configurations.

  python bench.py [--gpus N --steps K --warmup W] [--config 2|3|4|5]
                                                   (N > 1: launched by torch.distributed.run)

metric value = (lookups + updates) / s = 2 * n * N / t_step (n = FID occurrences per step per GPU),
inputs resident in HBM when the timed region starts.

  --config 2 (default at N = 1, the headline line): Criteo-shaped stream, 26 single-hot slots,
      1e8-row table, dim 16 fp32, batch 65536, Adagrad; step = lookup + apply_gradients(dedup).
  --config 3: 8 multi-hot slots (bags ~ Poisson(50), ids Zipf 1.1 over 1e7 per slot), batch 16384,
      half SUM half MEAN pooling into one CONCAT layout, FTRL; step = sharding_sparse_fids
      (dedup) -> lookup -> fused_embedding_to_layout -> its gradient -> apply_gradients.
  --config 4 (every N > 1): dim 64 Adagrad, 1.25e8 rows PER GPU (1e9 at N = 8) sharded by
      floormod(fid, N), batch 65536 per rank (weak scaling); step = the sync-training exchange
      of distributed_ps_sync: reorder/dedup -> all-to-all(FIDs) -> fused lookup on the owner ->
      all-to-all(rows) -> gather, then gather-gradient -> all-to-all(grads) -> fused optimize.
  --config 5: config 2 + sliding hash filter (threshold 3, capacity 3e8, 7 blocks), 5 % of every
      step's FIDs from a fresh range, req_time + 1 h per step, half the slots expire after a day,
      evict every 24 steps.
`--backend gloo` rehearses the exchange with several ranks on ONE GPU (not a measurement).
"""
import argparse
import ctypes as C
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E
SLOT_VOCAB = [40000000, 20000000, 10000000, 10000000, 5000000, 5000000, 3000000, 2000000,
              1000000, 1000000, 500000, 500000, 200000, 100000, 100000, 50000, 20000, 10000,
              10000, 5000, 2000, 1000, 500, 100, 50, 10]
SLOT_VOCAB[0] += 100000000 - sum(SLOT_VOCAB)  # 26 slots, exactly 1e8 FIDs


def make_batch(gen, batch, device, skew):
  """fid = slot << 48 | r, r ~ truncated power law over the slot's vocabulary."""
  vocab = torch.tensor(SLOT_VOCAB, dtype=torch.float64, device=device)
  u = torch.rand((batch, len(SLOT_VOCAB)), generator=gen, device=device, dtype=torch.float64)
  r = (vocab * u.pow(skew)).to(torch.int64)
  r = torch.minimum(r, (vocab - 1).to(torch.int64))
  slots = torch.arange(len(SLOT_VOCAB), device=device, dtype=torch.int64) << 48
  return (slots | r).reshape(-1).contiguous()


def populate(table, device, rows_per_slot_cap=None):
  """Pre-populates every FID of every slot (rows U(-0.05, 0.05) via the table initializer,
  Adagrad accumulators 0.1) with zero-valued assign_add in 4M-id chunks."""
  chunk = 1 << 22
  zeros = torch.zeros((chunk, 16), dtype=torch.float32, device=device)
  t0 = time.time()
  for s, v in enumerate(SLOT_VOCAB):
    for lo in range(0, v, chunk):
      n = min(chunk, v - lo)
      ids = (torch.arange(lo, lo + n, device=device, dtype=torch.int64) | (s << 48))
      table.assign_add(ids, zeros[:n], req_time=1)
  torch.cuda.synchronize()
  return time.time() - t0


def cpu_baseline(ids_batches, grad, lr, D=16):
  """The CPU oracle ('port' of the reference algorithm, oracle/oracle.cc) on a bounded sample of
  the same stream: tables holding exactly the FIDs of the sampled batches, then lookup +
  apply_gradients(dedup). Threads own disjoint table shards (no locks: an upper bound for the
  reference's striped-lock table); the thread count is the GPU box's CPU share."""
  from oracle import cpu
  threads = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1))
  ids_np = [b.cpu().numpy() for b in ids_batches]
  g = grad.cpu().numpy()
  seg = cpu.seg(D, "adagrad", "uniform", init_a=-0.05, init_b=0.05)
  dt, rows = cpu.bench_lookup_apply_mt([seg], ids_np, g, lr, threads)
  n = sum(len(b) for b in ids_np)
  dt1, _ = cpu.bench_lookup_apply_mt([seg], ids_np[:1], g, lr, 1)
  return {"value": 2.0 * n / dt, "unit": "lookups+updates/s", "cores": threads, "kind": "port",
          "single_thread_value": 2.0 * len(ids_np[0]) / dt1,
          "sample": "%d batches of %d FIDs (same stream), tables holding only those FIDs (%d rows), "
                    "%d shard-owning threads" % (len(ids_np), len(ids_np[0]), rows, threads)}


_REAL_STDOUT = None


def quiet_stdout():
  """Libraries (RCCL prints a version banner) write to fd 1; the contract is ONE JSON line there.
  Everything else is sent to stderr, the JSON line goes to the saved descriptor."""
  global _REAL_STDOUT
  if _REAL_STDOUT is None:
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def emit(res):
  line = (json.dumps(res) + "\n").encode()
  sys.stdout.flush()
  os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, line)


def main_sharded(args, rank, world, device):
  """N > 1: DistributedMultiTypeHashTableMpi over one merged table sharded by floormod(fid, N)."""
  import torch.distributed as dist
  os.environ["MONOLITH_SYNC_EMPTY_RANK0_PS_SHARD"] = "0"  # plain mod-N: all ranks hold rows
  from monolith_amd import distributed_ps_sync, entry, hash_table_ops, multi_type_hash_table
  global SLOT_VOCAB
  # config 4: dim 64, 1.25e8 rows per GPU (1e9 at N = 8); --sharded alone keeps config 2's shape
  cfg4 = args.config == 4
  D = args.dim or (64 if cfg4 else 16)
  rows = args.rows if args.rows != 1e8 else (1.25e8 * world if cfg4 else 1e8)
  if rows != 1e8:
    scale = rows / 1e8
    SLOT_VOCAB = [max(10, int(v * scale)) for v in SLOT_VOCAB]
  total_rows = sum(SLOT_VOCAB)
  lr = 0.01
  tc = entry.EmbeddingHashTableConfig(
      segments=[entry.CombineAsSegment(D, entry.RandomUniformInitializer(-0.05, 0.05),
                                       entry.AdagradOptimizer(initial_accumulator_value=0.1))],
      initial_capacity=int(total_rows / world * 1.05) + 1024, seed=1234)
  cfgs = {"criteo": entry.HashTableConfigInstance(tc, [lr])}

  def factory(idx):
    return multi_type_hash_table.MultiTypeHashTable(
        cfgs, lambda slot, c: hash_table_ops.hash_table_from_config(
            c, name_suffix="bench_%s_%d" % (slot, idx)))

  # the first half of the next lookup (reorder, size + FID exchange, the owner's slot resolution)
  # beside the update: the reference's enable_pipelined_fwda2a, without its stale rows
  # (N > 1 only: at N = 1 there is no exchange to move and the step is bound by launch work, where
  # the extra launches cost more than they save — 0.92 against 0.81 ms; MONO_BENCH_PIPELINED
  # overrides either way, = 2 adds the owner-side lookup hint)
  pl = os.environ.get("MONO_BENCH_PIPELINED", "1" if world > 1 else "0")
  pipelined = not args.no_prefetch and pl != "0"
  qc = {"enable_pipelined_fwda2a": 1, "announce_owner_lookup": 1 if pl == "2" else 0}
  table = distributed_ps_sync.DistributedMultiTypeHashTableMpi(
      world, factory, queue_configs=qc if pipelined else None)
  local = table._table._hash_tables["criteo"]
  # populate this rank's shard: fid = slot << 48 | r with floormod(fid, N) == rank
  chunk = 1 << 22
  chunk = 1 << (22 if D <= 16 else 20)
  zeros = torch.zeros((chunk, D), dtype=torch.float32, device=device)
  t0 = time.time()
  for s_, v in enumerate(SLOT_VOCAB):
    for lo in range(0, v, chunk):
      m = min(chunk, v - lo)
      ids = (torch.arange(lo, lo + m, device=device, dtype=torch.int64) | (s_ << 48))
      ids = ids[torch.remainder(ids, world) == rank].contiguous()
      if ids.numel():
        local.assign_add(ids, zeros[:ids.numel()], req_time=1)
  torch.cuda.synchronize()
  t_pop = time.time() - t0
  rows_live = torch.tensor([local.size()], dtype=torch.int64, device=device)
  if args.backend == "gloo":
    rows_live = rows_live.cpu()
  if world > 1:
    dist.all_reduce(rows_live)
  assert int(rows_live.item()) == total_rows, (int(rows_live.item()), total_rows)

  gen = torch.Generator(device=device)
  gen.manual_seed(20250117 + rank)
  n_pool = 8
  pool = [make_batch(gen, args.batch, device, args.skew) for _ in range(n_pool)]
  n = pool[0].numel()
  grad = (torch.randn((n, D), generator=gen, device=device, dtype=torch.float32) * 0.1)
  uniq = float(np.mean([torch.unique(b).numel() / n for b in pool]))

  state = {"t": table}
  # The reorder of batch i + 1 is issued on a side stream while step i runs, as the reference's
  # input pipeline does (distributed_ps_sync.py:313-331, early_reorder_indicies_res_pack): every
  # step still does one reorder, one lookup and one update; --no-prefetch keeps them in line.
  side = not args.no_prefetch
  where = os.environ.get("MONO_BENCH_PREFETCH_AT", "lookup")  # experiment: start | lookup | grad
  side_stream = torch.cuda.Stream(device) if side else None

  def step(i):
    ids = {"criteo": pool[i % n_pool]}
    nxt = {"criteo": pool[(i + 1) % n_pool]} if side else None
    pack = state.pop("pack", None) or state["t"].take_prefetched_reorder()
    if side and where == "start":
      pack = (pack if pack is not None else state["t"].reorder_begin(ids)).result()
      with torch.cuda.stream(side_stream):
        state["pack"] = state["t"].reorder_begin(nxt)
    emb, bundle = state["t"].lookup(ids, auxiliary_bundle={}, early_reorder_indicies_res_pack=pack)
    if side and where == "lookup":
      with torch.cuda.stream(side_stream):
        state["pack"] = state["t"].reorder_begin(nxt)
    state["t"] = state["t"].apply_gradients({"criteo": (ids["criteo"], grad)}, bundle,
                                            global_step=i, req_time=10 + i,
                                            next_slot_to_id=nxt if where == "grad" else None)
    if pipelined and where != "grad":  # the reorder began behind the lookup; its sequel here
      state["pack"] = state["t"].prefetch_forward(nxt, state.pop("pack", None))
    return emb

  def barrier():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
      torch.cuda.synchronize()

  for i in range(args.warmup):
    step(i)
  barrier()
  t0 = time.perf_counter()
  for i in range(args.steps):
    step(args.warmup + i)
  barrier()
  dt = time.perf_counter() - t0
  if world > 1:
    tmax = torch.tensor([dt], dtype=torch.float64,
                        device=device if args.backend == "nccl" else "cpu")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
  ms_per_step = dt / args.steps * 1e3

  # per-GPU algorithmic HBM bytes of one step: every occurrence is gathered and
  # its gradient scattered on the requesting rank (2 * (4 + 4D) + 4D out + 4D grad), every
  # per-rank-unique FID is looked up and updated once on its owner (144 + 72 + 268 B)
  per_gpu_bytes = n * (8 + 4 * D * 2 + 8) + uniq * n * ((16 + 8 * D) + (8 + 4 * D) + (12 + 16 * D))
  step_gbs = per_gpu_bytes / (ms_per_step * 1e-3) / 1e9
  # xGMI: a per-rank-unique FID moves 8 B out, 4D B back, 4D B of gradient; (N-1)/N are remote
  fabric_bytes = uniq * n * (8 + 8 * D) * (world - 1) / world
  res = {
      "metric": "FID lookups+updates/sec (node)", "value": 2.0 * n * world / (dt / args.steps),
      "unit": "lookups+updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
      "vs_baseline": None, "dtype": "f32", "data": "synthetic",
      "config": {"workload": "%scriteo-shaped: 26 single-hot slots, %.3g-row table sharded by "
                             "floormod(fid, %d), dim %d fp32, batch %d per GPU, Adagrad, "
                             "distributed_ps_sync lookup + apply_gradients (FID / row / grad "
                             "all-to-all, backend %s)" % ("config 4: " if cfg4 else "", total_rows,
                                                          world, D, args.batch, args.backend),
                 "fids_per_step_per_gpu": n, "unique_ratio_per_rank": uniq, "skew": args.skew,
                 "parallelism": "table sharded over %d GPUs, all-to-all exchange" % world,
                 "reorder": "one batch ahead on a side stream" if side else "in line",
                 "pipelined_forward": ("size + FID exchange of the next batch beside the update"
                                       + (" + owner lookup announced" if pl == "2" else ""))
                                      if pipelined else "off",
                 "populate_s": t_pop, "table_rows": total_rows},
      "roofline": {"bound": "hbm", "kernel": "whole step (per GPU)", "achieved": step_gbs,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS,
                   "traffic": None,
                   "xgmi_GBps_per_gpu": fabric_bytes / (ms_per_step * 1e-3) / 1e9},
      "cpu_baseline": None,
  }
  if rank == 0 and world == 1 and not args.no_cpu:
    res["cpu_baseline"] = cpu_baseline(pool[:2], grad, lr, D)
  if rank == 0:
    emit(res)
  if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()


def _timed_steps(step, args, sync):
  for i in range(args.warmup):
    step(i)
  sync()
  t0 = time.perf_counter()
  for i in range(args.steps):
    step(args.warmup + i)
  sync()
  return (time.perf_counter() - t0) / args.steps


def _phase_ms(phases, reps=6):
  """Average milliseconds of every (name, fn) phase, run in order `reps` times, timed with HIP
  events on the current stream (the stream every kernel of these phases is launched on, or
  joined to before the phase returns)."""
  ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(phases) + 1)] for _ in range(reps)]
  for r in range(reps):
    ev[r][0].record()
    for k, (_, fn) in enumerate(phases):
      fn()
      ev[r][k + 1].record()
  torch.cuda.synchronize()
  out = {}
  for k, (name, _) in enumerate(phases):
    out[name] = float(np.mean([ev[r][k].elapsed_time(ev[r][k + 1]) for r in range(2, reps)]))
  return out


def main_config3(args, device):
  """BASELINE config 3: multi-hot slots, sum / mean pooling into a layout, FTRL."""
  from monolith_amd import distribution_ops, entry, feature_configs as fc, hash_table_ops, parsers
  from monolith_amd.embedding_combiners import Ragged
  S, D, B, lr = 8, 16, args.batch if args.batch != 65536 else 16384, 0.01
  vocab = int(1e7 * (args.rows / 1e8)) if args.rows != 1e8 else 10_000_000
  cfgs = fc.FeatureConfigs()
  concat = fc.OutConfig()
  names = ["slot_%02d" % s_ for s_ in range(S)]
  for s_, name in enumerate(names):
    pool = fc.PoolingType.SUM if s_ % 2 == 0 else fc.PoolingType.MEAN
    cfgs.feature_configs[name] = fc.FeatureConfig("t0", [D], pool, 0)
    concat.slice_configs.append(fc.SliceConfig(name, 0, D))
  fc.infer_shape(concat, fc.OutType.CONCAT)
  cfgs.out_configs = {"concat": concat}
  tc = entry.EmbeddingHashTableConfig(
      segments=[entry.CombineAsSegment(D, entry.RandomUniformInitializer(-0.05, 0.05),
                                       entry.FtrlOptimizer(initial_accumulator_value=0.1, beta=1.0))],
      initial_capacity=int(S * vocab * 1.02), seed=77)
  table = hash_table_ops.hash_table_from_config(entry.HashTableConfigInstance(tc, [lr]),
                                                name_suffix="bench_c3")
  chunk = 1 << 22
  zeros = torch.zeros((chunk, D), dtype=torch.float32, device=device)
  t0 = time.time()
  for s_ in range(S):
    for lo in range(0, vocab, chunk):
      m = min(chunk, vocab - lo)
      ids = torch.arange(lo, lo + m, device=device, dtype=torch.int64) | ((s_ + 1) << 48)
      table.assign_add(ids, zeros[:m], req_time=1)
  torch.cuda.synchronize()
  t_pop = time.time() - t0
  assert table.size() == S * vocab

  gen = torch.Generator(device=device)
  gen.manual_seed(20250117)

  def make_batch():
    feats = {}
    for s_, name in enumerate(names):
      lens = torch.poisson(torch.full((B,), 50.0, device=device), generator=gen).clamp_(1, 200).to(torch.int64)
      rs = torch.zeros(B + 1, dtype=torch.int64, device=device)
      rs[1:] = torch.cumsum(lens, 0)
      tot = int(rs[-1].item())
      u = torch.rand(tot, generator=gen, device=device, dtype=torch.float64)
      # Zipf(1.1) over [1, vocab]: inverse of the truncated continuous power-law CDF
      a = 1.0 - 1.1
      r = ((float(vocab) ** a - 1.0) * u + 1.0).pow(1.0 / a).to(torch.int64).clamp_(1, vocab) - 1
      feats[name] = Ragged(r | ((s_ + 1) << 48), rs)
    return feats

  n_pool = 4
  pool = [make_batch() for _ in range(n_pool)]
  n = int(np.mean([sum(f.values.numel() for f in b.values()) for b in pool]))
  gout = torch.randn((B, S * D), generator=gen, device=device, dtype=torch.float32) * 0.1
  st = {}

  def p_shard(i):
    st["sh"] = parsers.sharding_sparse_fids(pool[i % n_pool], 1, cfgs, True, fid_list_ret_list=True)

  def p_lookup():
    fid_list = st["sh"][0]
    st["emb"] = table.lookup(fid_list[0])

  def p_layout():
    fid_list, fid_offset, feature_offset, nfl, batch, splits = st["sh"]
    st["out"] = distribution_ops.fused_embedding_to_layout(
        [st["emb"].reshape(-1)], splits, fid_offset, feature_offset, nfl, batch, "example_batch",
        cfgs, 1)

  def p_fused():
    fid_list, fid_offset, feature_offset, nfl, batch, splits = st["sh"]
    st["out2"] = distribution_ops.fused_lookup_to_layout(
        [table], fid_list, splits, fid_offset, feature_offset, nfl, batch, "example_batch", cfgs, 1)

  def p_grad():
    fid_list, fid_offset, feature_offset, nfl, batch, splits = st["sh"]
    # (the gradient op reads only the SHAPE of the embedding list: with the fused forward there
    # is no looked-up buffer, an unwritten tensor of that shape stands in)
    emb = torch.empty((fid_list[0].numel() * D,), dtype=torch.float32, device=device)
    st["g"] = distribution_ops.fused_embedding_to_layout_grad(
        nfl, feature_offset, fid_offset, batch, [emb], splits, [gout], "example_batch", cfgs, 1)

  def p_apply(i):
    nonlocal table
    fid_list = st["sh"][0]
    table = table.apply_gradients(fid_list[0], st["g"][0].reshape(-1, D), global_step=i,
                                  req_time=10 + i)

  two_step = not bool(getattr(args, "fused_forward", False))

  # sharding_sparse_fids belongs to the reference's INPUT pipeline (data/parsers.py:402-456: a
  # dataset map function, prefetched): batch i + 1 is sharded on a side stream while batch i's
  # update runs, as the sharded bench does with its reorder. Every batch is still sharded once per
  # step inside the timed region; --no-prefetch keeps it in line.
  ahead = not args.no_prefetch
  side = torch.cuda.Stream(device) if ahead else None

  def shard_ahead(i):
    with torch.cuda.stream(side):
      res = parsers.sharding_sparse_fids(pool[i % n_pool], 1, cfgs, True, fid_list_ret_list=True)
      ev = side.record_event()
    return res, ev

  def tensors_of(x):
    if isinstance(x, torch.Tensor):
      yield x
    elif isinstance(x, (list, tuple)):
      for y in x:
        yield from tensors_of(y)

  def step(i):
    if ahead:
      res, ev = st.pop("next") if "next" in st else shard_ahead(i)
      cur = torch.cuda.current_stream(device)
      cur.wait_event(ev)
      for t_ in tensors_of(res):  # (allocated under the side stream, read on this one)
        if t_.is_cuda:
          t_.record_stream(cur)
      st["sh"] = res
    else:
      p_shard(i)
    if two_step:
      p_lookup(); p_layout()
    else:
      p_fused()
    p_grad(); p_apply(i)
    if ahead:  # (the host waits in here for the counts: beside the update it has just launched)
      st["next"] = shard_ahead(i + 1)

  dt = _timed_steps(step, args, torch.cuda.synchronize)
  st.pop("next", None)
  u_per_step = float(np.mean([torch.cat([torch.unique(f.values) for f in b.values()]).numel()
                              for b in pool]))
  # phases in step order (the update then finds the slot record / prepared order of the step's
  # own lookup); the forward the step does not use is timed on its own afterwards
  fwd_two = [("lookup", p_lookup), ("layout_forward", p_layout)]
  fwd_one = [("fused_lookup_to_layout", p_fused)]
  ph = _phase_ms([("sharding_sparse_fids", lambda: p_shard(0))] + (fwd_two if two_step else fwd_one) +
                 [("layout_grad", p_grad), ("apply_gradients", lambda: p_apply(999))])
  ph.update(_phase_ms(fwd_one if two_step else fwd_two))
  p_lookup(); p_layout(); p_fused()  # (no update in between: the same rows)
  same = all(torch.equal(a, b) for a, b in zip(st["out"], st["out2"]))
  # algorithmic bytes: pooled lookup per occurrence 8 (descriptor) + 4D (row read)
  # + 4D / L (pooled output); per unique FID the lookup moves 8 + 8 + 4D + 4D; the gradient
  # scatter reads 4D per occurrence and writes 4D per unique; FTRL update 8 + 8 + 4D + 2*12D + 4
  u = u_per_step
  alg = {
      "sharding_sparse_fids": n * (8 + 8) + u * 8,
      "lookup": u * (16 + 8 * D),
      "layout_forward": n * (8 + 4 * D) + B * S * D * 4,
      "fused_lookup_to_layout": u * 20 + n * (8 + 4 + 4 * D) + B * S * D * 4,
      "layout_grad": n * (8 + 4 * D) + u * 4 * D,
      "apply_gradients": u * (8 + 4 * D) + u * (8 + 2 * 12 * D + 4),
  }
  off_step = ("fused_lookup_to_layout",) if two_step else ("lookup", "layout_forward")
  step_alg = sum(alg[k] for k in alg if k not in off_step)
  dom = max((k for k in ph if k not in off_step), key=lambda k: ph[k])
  achieved = alg[dom] / (ph[dom] * 1e-3) / 1e9
  res = {
      "metric": "FID lookups+updates/sec (node)", "value": 2.0 * n / dt,
      "unit": "lookups+updates/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
      "dtype": "f32", "data": "synthetic",
      "config": {"workload": "config 3: %d multi-hot slots, bags ~ Poisson(50) in [1, 200], ids "
                             "Zipf(1.1) over %.3g per slot, batch %d, half SUM half MEAN pooling "
                             "into a CONCAT layout, dim %d, FTRL; sharding_sparse_fids%s -> %s -> "
                             "gradient -> apply_gradients" %
                             (S, vocab, B, D,
                              " (of the next batch, on a side stream beside the update, as the "
                              "reference's input pipeline prefetches it)" if ahead else "",
                              "lookup -> fused_embedding_to_layout" if two_step
                              else "fused lookup + pool (fused_lookup_to_layout)"),
                 "fids_per_step": n, "unique_fids_per_step": u, "table_rows": S * vocab,
                 "populate_s": t_pop, "fused_forward_equals_two_step": bool(same)},
      "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                   "algorithmic_bytes": alg[dom], "phase_ms": ph,
                   "phase_algorithmic_GBps": {k: alg[k] / (ph[k] * 1e-3) / 1e9 for k in ph},
                   "phase_note": "phases timed with HIP events on the launch stream outside the "
                                 "timed region; not part of the step: " + " + ".join(off_step) +
                                 (" (the one-pass alternative)" if two_step else
                                  " (the two-step alternative to fused_lookup_to_layout)"),
                   "step_algorithmic_GBps": step_alg / dt / 1e9,
                   "step_frac": step_alg / dt / 1e9 / HBM_PEAK_GBS},
      "cpu_baseline": None,
  }
  if not args.no_cpu:
    res["cpu_baseline"] = cpu_baseline_config3(pool[0], cfgs, names, D, lr, B)
  emit(res)


def cpu_baseline_config3(batch, cfgs, names, D, lr, B, sample=384):
  """The oracle ('port') on the first `sample` samples of one batch: per-feature dedup, table
  lookup, pooling into the layout, its gradient, FTRL update. One thread."""
  from monolith_amd import feature_configs as fc
  from oracle import cpu
  plan = fc.build_layout_plan(cfgs)
  fid_offset, feature_offset, nfl_offset, uniq = [], [], [], []
  n_occ = 0
  for fi, name in enumerate(names):
    rs = batch[name].row_splits[:sample + 1].cpu().numpy()
    vals = batch[name].values[:int(rs[-1])].cpu().numpy()
    n_occ += len(vals)
    nfl_offset.append(len(feature_offset))
    u, inv = np.unique(vals, return_inverse=True)
    first = np.full(len(u), len(vals), np.int64)
    np.minimum.at(first, inv, np.arange(len(vals)))
    order = np.argsort(first, kind="stable")       # first-occurrence order
    rank = np.empty(len(u), np.int64)
    rank[order] = np.arange(len(u))
    uniq.append(u[order])
    feature_offset += rs[:-1].tolist()
    base = len(fid_offset)
    feature_offset[-sample:] = [base + int(x) for x in rs[:-1]]
    fid_offset += ((fi << 32) | rank[inv]).tolist()
  fid_offset = np.array(fid_offset, np.uint64)
  feature_offset = np.array(feature_offset, np.int32)
  nfl_offset = np.array(nfl_offset, np.uint32)
  ids = np.concatenate(uniq)
  seg = cpu.seg(D, "ftrl", "uniform", init_acc=0.1, beta=1.0, init_a=-0.05, init_b=0.05)
  t = cpu.OracleTable([seg], initial_capacity=len(ids) * 2)
  t.assign_add(ids, np.zeros((len(ids), D), np.float32), req_time=1)
  gout = [np.random.default_rng(0).normal(0, 0.1, (sample, len(names) * D)).astype(np.float32)]
  t0 = time.perf_counter()
  reps = 0
  while time.perf_counter() - t0 < 8.0 or reps < 2:
    emb = t.lookup(ids)
    bufs, pos = [], 0
    for u in uniq:
      bufs.append((emb[pos:pos + len(u)].reshape(-1), D))
      pos += len(u)
    cpu.embedding_to_layout(plan.slices, plan.out_shapes, bufs, fid_offset, feature_offset,
                            nfl_offset, sample)
    g = cpu.embedding_to_layout_grad(plan.slices, plan.slice_idx_of, plan.max_slice_num,
                                     [(len(a), d) for a, d in bufs], fid_offset, feature_offset,
                                     nfl_offset, sample, gout)
    t.apply_gradients(ids, np.concatenate(g).reshape(-1, D), [lr], req_time=5 + reps)
    reps += 1
  dt = (time.perf_counter() - t0) / reps
  return {"value": 2.0 * n_occ / dt, "unit": "lookups+updates/s", "cores": 1, "kind": "port",
          "sample": "the first %d samples of one batch (%d FID occurrences, %d unique), tables "
                    "holding only those FIDs; lookup + pooling + gradient + FTRL update, dedup "
                    "excluded, %d repetitions" % (sample, n_occ, len(ids), reps)}


def main_config5(args, device):
  """BASELINE config 5: config 2's stream with a sliding hash filter in front of inserts, 5 % of
  every step's FIDs from a fresh range (every fresh FID recurs in five consecutive steps: the
  filter admits it on its fourth sight, hash_filter.h:157-164), req_time + 1 h per step, every
  second slot expires after a day, expiry eviction twice inside the timed region."""
  from monolith_amd import _lib, entry, hash_filter_ops, hash_table_ops
  global SLOT_VOCAB
  if args.rows != 1e8:
    scale = args.rows / 1e8
    SLOT_VOCAB = [max(10, int(v * scale)) for v in SLOT_VOCAB]
  total_rows = sum(SLOT_VOCAB)
  S = len(SLOT_VOCAB)
  lr, D = 0.01, 16
  tc = entry.EmbeddingHashTableConfig(
      segments=[entry.CombineAsSegment(D, entry.RandomUniformInitializer(-0.05, 0.05),
                                       entry.AdagradOptimizer(initial_accumulator_value=0.1))],
      initial_capacity=int(total_rows * 1.15), seed=1234)
  tc.slot_expire_time_config.default_expire_time = 36500
  tc.slot_expire_time_config.slot_expire_times = {s_: 1 for s_ in range(0, S, 2)}
  thr = entry.SlotOccurrenceThresholdConfig(default_occurrence_threshold=3)
  filt = hash_filter_ops.create_hash_filters(0, True, thr, filter_capacity=300_000_000)[0]
  now0 = 1_700_000_000
  # evictions: after the timed steps number ev_every and 2 * ev_every (both inside the timed region
  # whatever --steps / --warmup are); absolute step indices i1 < i2
  ev_every = max(1, args.steps // 2)
  i1, i2 = args.warmup + ev_every - 1, args.warmup + 2 * ev_every - 1
  # A trained table: populated WITHOUT the filter, last touched "now" — except two classes of rows
  # of the expiring slots (1 in 256 each) that were last touched almost a day ago: class A is a
  # day old at the first eviction, class B only at the second.
  ts_a = now0 - 86400 + 3600
  ts_b = now0 - 86400 + 3600 * (i1 + 2)
  table = hash_table_ops.hash_table_from_config(entry.HashTableConfigInstance(tc, [lr]),
                                                hash_filter=filt, name_suffix="bench_c5")
  chunk = 1 << 22
  zeros = torch.zeros((chunk, D), dtype=torch.float32, device=device)

  def touch(ids, ts):
    _lib.check(_lib.lib.mono_table_assign_add(table.table.h, None, _lib.vp(ids), _lib.vp(zeros),
                                              C.c_int64(ids.numel()), C.c_int64(ts),
                                              _lib.stream_ptr(device)))

  t0 = time.time()
  stale = 0
  for s_, v in enumerate(SLOT_VOCAB):
    for lo in range(0, v, chunk):
      m = min(chunk, v - lo)
      r = torch.arange(lo, lo + m, device=device, dtype=torch.int64)
      touch(r | (s_ << 48), now0)
      if s_ % 2 == 0:
        for cls, ts in ((0, ts_a), (1, ts_b)):
          old = (r[(r & 255) == cls] | (s_ << 48)).contiguous()
          if old.numel():
            touch(old, ts)
            stale += old.numel()
  torch.cuda.synchronize()
  t_pop = time.time() - t0
  rows0 = table.size()

  gen = torch.Generator(device=device)
  gen.manual_seed(20250117)
  n_pool = 8
  pool = [make_batch(gen, args.batch, device, args.skew) for _ in range(n_pool)]
  n = pool[0].numel()
  grad = torch.randn((n, D), generator=gen, device=device, dtype=torch.float32) * 0.1
  n_fresh = n // 20
  recur = 5
  fresh_pos = torch.randperm(n, generator=gen, device=device)[:n_fresh]
  state = {"t": table, "evictions": 0, "evicted_rows": 0, "evict_ms": []}

  # The next lookup is announced one step ahead (a gated update keeps the queued hints: an
  # insertion moves no row, k_hint_fix probes the ids recorded as absent again). One ahead, not
  # two as on config 2: 471 / 509 / 588 us per step with one / none / two.
  ahead = 0 if args.no_hint else args.c5_hint_ahead
  made = {}  # step -> its ids (made `ahead` steps early when the lookups are announced ahead)

  def make_ids(i):
    ids = pool[i % n_pool].clone()
    # 5 % churn: the fresh FID at position p of step i is (1 << 40) + i * n_fresh / recur + p, so it
    # comes back in the next four steps (at another position); its slot is a function of the FID
    x = (1 << 40) + i * (n_fresh // recur) + torch.arange(n_fresh, device=device)
    ids[fresh_pos] = ((x % S) << 48) | x
    return ids

  ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
  # the input pipeline's work (building the id tensors) is not part of the step: every batch of
  # the run exists before the clock starts, as on config 2
  for j in range(args.warmup + args.steps + 8 + ahead + 1):
    made[j] = make_ids(j)
  torch.cuda.synchronize()

  marks = {}  # an event at the start of every step: the period of a step that does not evict

  def step(i):
    ids = made.pop(i)
    now = now0 + 3600 * (i + 1)
    t = state["t"]
    marks[i] = torch.cuda.Event(enable_timing=True)
    marks[i].record()
    t.lookup(ids)
    for a in range(1, ahead + 1):
      t.lookup_hint(made[i + a], a)
    state["t"] = t.apply_gradients(ids, grad, global_step=i, enable_dedup=True, req_time=now)
    if i in (i1, i2):
      before = state["t"].size()  # (reads a counter back: part of the timed region, twice per run)
      ev[0].record()
      state["t"].evict(now)
      ev[1].record()
      after = state["t"].size()
      state["evictions"] += 1
      state["evicted_rows"] += before - after
      state["evict_ms"].append(ev[0].elapsed_time(ev[1]))

  dt = _timed_steps(step, args, torch.cuda.synchronize)
  rows1 = state["t"].size()
  inserted = rows1 - rows0 + state["evicted_rows"]
  periods = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.warmup, args.warmup + args.steps - 1)
             if i + 1 in marks]
  uniq = float(np.mean([torch.unique(b).numel() / n for b in pool]))
  # phases of the steps that follow, each on its own ids (lookup announced as in the timed loop)
  i0 = args.warmup + args.steps
  reps = 8
  pe = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
  for r in range(reps):
    i = i0 + r
    ids = made.pop(i)
    pe[r][0].record()
    state["t"].lookup(ids)
    pe[r][1].record()
    for a in range(1, ahead + 1):
      state["t"].lookup_hint(made[i + a], a)
    state["t"] = state["t"].apply_gradients(ids, grad, global_step=i, enable_dedup=True,
                                            req_time=now0 + 3600 * (i + 1))
    pe[r][2].record()
  torch.cuda.synchronize()
  ph = {"lookup": float(np.mean([pe[r][0].elapsed_time(pe[r][1]) for r in range(2, reps)])),
        "gated apply_gradients": float(np.mean([pe[r][1].elapsed_time(pe[r][2]) for r in range(2, reps)]))}
  # algorithmic bytes: config 2's step + 2 B read / 2 B written per gated (unknown) id and probed
  # filter block (two blocks forward); eviction scan 12 B per slot
  step_bytes = n * 144 + n * (8 + 4 * D) + uniq * n * (8 + 2 * (8 * D) + 4) + n_fresh * 2 * 4
  cap_slots = 1
  while cap_slots * 0.6 < total_rows * 1.15:
    cap_slots *= 2
  evict_ms = float(np.mean(state["evict_ms"])) if state["evict_ms"] else float("nan")
  res = {
      "metric": "FID lookups+updates/sec (node)", "value": 2.0 * n / dt,
      "unit": "lookups+updates/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
      "dtype": "f32", "data": "synthetic",
      "config": {"workload": "config 5: criteo-shaped stream (26 slots, %.3g rows, dim 16, Adagrad, "
                             "batch %d) + sliding hash filter (threshold 3, capacity 3e8, 7 blocks) "
                             "gating inserts, 5 %% fresh FIDs per step (each recurs in %d consecutive "
                             "steps, admitted on its fourth sight), req_time + 1 h per step, every "
                             "second slot expires after 1 day, expiry eviction after timed steps %d "
                             "and %d%s" %
                             (total_rows, args.batch, recur, ev_every, 2 * ev_every,
                              ", the ids of the next %d batches announced by lookup_hint" % ahead
                              if ahead else ""),
                 "fids_per_step": n, "unique_ratio": uniq, "fresh_fids_per_step": n_fresh,
                 "rows_before": rows0, "rows_after": rows1, "stale_rows_planted": stale,
                 "inserted_rows": inserted, "inserted_per_step": inserted / float(args.warmup + args.steps),
                 "evictions": state["evictions"], "evicted_rows": state["evicted_rows"],
                 "filter_state": {k: v for k, v in filt.state().items() if k != "num_elements"},
                 "populate_s": t_pop},
      "roofline": {"bound": "hbm", "kernel": "whole step (lookup + gated apply_gradients)",
                   "achieved": step_bytes / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": step_bytes / dt / 1e9 / HBM_PEAK_GBS, "traffic": None,
                   "algorithmic_bytes": step_bytes, "phase_ms": ph,
                   # (HIP events at the start of every timed step: the median is a step that does
                   # not evict; ms_per_step above averages the two evictions and what follows in)
                   "step_period_ms_median": float(np.median(periods)) if periods else None,
                   "step_period_ms_max": float(np.max(periods)) if periods else None,
                   "k_evict_ms": evict_ms, "k_evict_slots": cap_slots,
                   "k_evict_GBps": cap_slots * 12 / (evict_ms * 1e-3) / 1e9 if evict_ms == evict_ms else None,
                   "k_evict_frac": (cap_slots * 12 / (evict_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                    if evict_ms == evict_ms else None),
                   "step_algorithmic_GBps": step_bytes / dt / 1e9,
                   "step_frac": step_bytes / dt / 1e9 / HBM_PEAK_GBS},
      "cpu_baseline": None,
  }
  if not args.no_cpu:
    res["cpu_baseline"] = cpu_baseline_config5([make_ids(i0 + reps + k) for k in range(2)], pool, grad, lr, D,
                                               now0)
  emit(res)


def cpu_baseline_config5(batches, pool, grad, lr, D, now0):
  """The oracle ('port', one thread) on two batches of the same stream: a table holding exactly the
  known FIDs of those batches, the sliding filter (same threshold / capacity / blocks) in front of
  the inserts; per batch lookup + apply_gradients(dedup, filter)."""
  from oracle import cpu
  ids_np = [b.cpu().numpy() for b in batches]
  g = grad.cpu().numpy()
  known = np.unique(np.concatenate([b.cpu().numpy() for b in pool]))
  seg = cpu.seg(D, "adagrad", "uniform", init_a=-0.05, init_b=0.05)
  t = cpu.OracleTable([seg], initial_capacity=len(known) * 2)
  t.assign_add(known, np.zeros((len(known), D), np.float32), req_time=now0)
  f = cpu.OracleFilter("sliding", 300_000_000, 7, 3)
  t0 = time.perf_counter()
  for k, ids in enumerate(ids_np):
    t.lookup(ids)
    t.apply_gradients(ids, g, [lr], req_time=now0 + 3600 * (k + 1), enable_dedup=True, filt=f)
  dt = time.perf_counter() - t0
  n = sum(len(b) for b in ids_np)
  return {"value": 2.0 * n / dt, "unit": "lookups+updates/s", "cores": 1, "kind": "port",
          "sample": "%d batches of %d FIDs of the same stream (5 %% fresh), table holding the %d known "
                    "FIDs, sliding filter threshold 3; lookup + gated apply_gradients(dedup), no "
                    "eviction" % (len(ids_np), len(ids_np[0]), len(known))}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--batch", type=int, default=65536)
  ap.add_argument("--rows", type=float, default=1e8, help="debug: scale the table down")
  ap.add_argument("--skew", type=float, default=3.0, help="power-law exponent (1 = uniform)")
  ap.add_argument("--no-cpu", action="store_true")
  ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) | gloo (single-GPU rehearsal)")
  ap.add_argument("--sharded", action="store_true",
                  help="run the N > 1 exchange pipeline even at N = 1 (profiling aid)")
  ap.add_argument("--no-prefetch", action="store_true",
                  help="sharded pipeline: reorder each batch in line instead of one batch ahead")
  ap.add_argument("--fused-forward", action="store_true",
                  help="config 3: the fused lookup + pool in the step instead of lookup + "
                       "fused_embedding_to_layout (1.98 against 1.95 ms: the update order prepared "
                       "beside the pooling kernel slows it by what the fusion saves)")
  ap.add_argument("--no-hint", action="store_true",
                  help="config 2: do not announce the next batch's ids (mono_table_lookup_hint) "
                       "between a step's lookup and its update")
  ap.add_argument("--c5-hint-ahead", type=int, default=3,
                  help="config 5: lookups announced ahead (0..3; --no-hint = 0)")
  # (three ahead: 318 against 324 us with two, config 5 0.50 against 0.55 ms with one and 0.61 with
  # two — the hinted chains run 3-5x slower beside the apply kernels than alone and need the room)
  ap.add_argument("--hint-ahead", type=int, default=3, help="config 2: lookups announced ahead (1..3)")
  ap.add_argument("--config", type=int, default=0, help="2 | 3 | 4 | 5 (default: 2, or 4 when N > 1)")
  ap.add_argument("--dim", type=int, default=0, help="config 4 / --sharded: row width (default 64 / 16)")
  args = ap.parse_args()
  if args.config == 0:
    args.config = 4 if int(os.environ.get("WORLD_SIZE", "1")) > 1 else 2
  quiet_stdout()

  rank = int(os.environ.get("RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if args.backend == "gloo":
    local_rank = 0  # rehearsal: all ranks share the one GPU
  torch.cuda.set_device(local_rank)
  device = torch.device("cuda", local_rank)
  force = os.environ.get("MONOLITH_AMD_FORCE_COLLECTIVE") == "1"  # one-rank RCCL rehearsal
  if world > 1 or force:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if args.backend == "nccl":
      dist.init_process_group("nccl", device_id=device)
    else:
      dist.init_process_group("gloo")
    return main_sharded(args, rank, world, device)
  if args.sharded or args.config == 4:
    return main_sharded(args, rank, world, device)
  if args.config == 3:
    return main_config3(args, device)
  if args.config == 5:
    return main_config5(args, device)

  from monolith_amd import _lib, entry, hash_table_ops
  global SLOT_VOCAB
  if args.rows != 1e8:
    scale = args.rows / 1e8
    SLOT_VOCAB = [max(10, int(v * scale)) for v in SLOT_VOCAB]
  total_rows = sum(SLOT_VOCAB)

  lr = 0.01
  tc = entry.EmbeddingHashTableConfig(
      segments=[entry.CombineAsSegment(16, entry.RandomUniformInitializer(-0.05, 0.05),
                                       entry.AdagradOptimizer(initial_accumulator_value=0.1))],
      initial_capacity=int(total_rows * 1.02), seed=1234 + rank)
  table = hash_table_ops.hash_table_from_config(entry.HashTableConfigInstance(tc, [lr]),
                                                name_suffix="bench%d" % rank)
  t_pop = populate(table, device)
  rows_live = table.size()
  assert rows_live == total_rows, (rows_live, total_rows)

  gen = torch.Generator(device=device)
  gen.manual_seed(20250117 + rank)
  n_pool = 8
  pool = [make_batch(gen, args.batch, device, args.skew) for _ in range(n_pool)]
  n = pool[0].numel()
  grad = (torch.randn((n, 16), generator=gen, device=device, dtype=torch.float32) * 0.1)
  uniq = float(np.mean([torch.unique(b).numel() / n for b in pool]))
  out = torch.empty((n, 16), dtype=torch.float32, device=device)
  h = table.table.h
  lrs = (C.c_float * 1)(lr)
  stream = torch.cuda.current_stream(device)
  sp = C.c_void_p(stream.cuda_stream)

  def hint(i):
    if not args.no_hint:
      # the input pipeline knows the next batches while this one trains (the reference prefetches
      # batches the same way): their slots and update orders are resolved beside this step's update
      for ahead in range(1, args.hint_ahead + 1):  # (the table takes the one that extends its queue)
        _lib.check(_lib.lib.mono_table_lookup_hint(h, _lib.vp(pool[(i + ahead) % n_pool]),
                                                   C.c_int64(n), C.c_int(ahead), sp))

  def step(i):
    ids = pool[i % n_pool]
    _lib.check(_lib.lib.mono_table_lookup(h, _lib.vp(ids), C.c_int64(n), _lib.vp(out), sp))
    hint(i)
    _lib.check(_lib.lib.mono_table_apply_gradients(h, None, _lib.vp(ids), _lib.vp(grad),
                                                   C.c_int64(n), lrs, C.c_int(1), C.c_int64(10 + i),
                                                   C.c_int64(i), C.c_int(1), sp))

  def barrier():
    if world > 1:
      torch.distributed.barrier()
    torch.cuda.synchronize()

  for i in range(args.warmup):
    step(i)
  barrier()
  t0 = time.perf_counter()
  for i in range(args.steps):
    step(args.warmup + i)
  barrier()
  dt = time.perf_counter() - t0
  if world > 1:
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
  ms_per_step = dt / args.steps * 1e3

  # ---- per-kernel timing with HIP events on the launch stream (roofline leg, untimed region)
  _lib.check(_lib.lib.mono_table_profile(h, 1))
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  lk, ph = [], []
  ms3 = (C.c_float * 3)()
  for i in range(args.warmup + args.steps, args.warmup + args.steps + 10):  # (the hinted sequence goes on)
    ids = pool[i % n_pool]
    ev0.record(stream)
    _lib.check(_lib.lib.mono_table_lookup(h, _lib.vp(ids), C.c_int64(n), _lib.vp(out), sp))
    ev1.record(stream)
    hint(i)
    _lib.check(_lib.lib.mono_table_apply_gradients(h, None, _lib.vp(ids), _lib.vp(grad),
                                                   C.c_int64(n), lrs, C.c_int(1), C.c_int64(99),
                                                   C.c_int64(i), C.c_int(1), sp))
    _lib.check(_lib.lib.mono_table_last_phase_ms(h, ms3))
    # k_apply alone (events on the auxiliary stream it runs on, beside k_apply_long); should the
    # library report no such launch, the phase time stands in
    ka = C.c_float(0)
    if _lib.lib.mono_table_last_apply_kernel_ms(h, C.byref(ka)) != 0:
      ka = C.c_float(float("nan"))
    torch.cuda.synchronize()
    lk.append(ev0.elapsed_time(ev1))
    ph.append([ms3[0], ms3[1], ms3[2], ka.value])
  _lib.check(_lib.lib.mono_table_profile(h, 0))
  # ---- the drop-in surface alone (mono_table_lookup + mono_table_apply_gradients, the two calls
  # the reference's HashTable has): the same steps with no lookup announced ahead, timed the same
  # way after the queued hints have run out
  no_hint_ms = None
  if not args.no_hint:
    def step_plain(i):
      ids = pool[i % n_pool]
      _lib.check(_lib.lib.mono_table_lookup(h, _lib.vp(ids), C.c_int64(n), _lib.vp(out), sp))
      _lib.check(_lib.lib.mono_table_apply_gradients(h, None, _lib.vp(ids), _lib.vp(grad),
                                                     C.c_int64(n), lrs, C.c_int(1), C.c_int64(10 + i),
                                                     C.c_int64(i), C.c_int(1), sp))
    base_i = args.warmup + args.steps + 10
    for i in range(6):
      step_plain(base_i + i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
      step_plain(base_i + 6 + i)
    torch.cuda.synchronize()
    no_hint_ms = (time.perf_counter() - t1) / args.steps * 1e3
  lk_ms = float(np.mean(lk[2:]))
  find_ms, sort_ms, apply_ms, kapply_ms = [float(x) for x in np.mean(np.array(ph[2:]), axis=0)]
  # occurrences / keys that k_apply_long serves instead (segments longer than 96, table.hip kLongSeg)
  n_long, u_long = [], []
  for b in pool:
    cnt = torch.unique(b, return_counts=True)[1]
    n_long.append(float(cnt[cnt > 96].sum().item()))
    u_long.append(float((cnt > 96).sum().item()))
  n_long, u_long = float(np.mean(n_long)), float(np.mean(u_long))

  # algorithmic bytes (SURVEY.md §8d, D = 16, Adagrad): lookup 144 B / occurrence;
  # update 72 B / occurrence + 268 B / unique FID  (u/n measured on the stream)
  D = 16
  lookup_bytes = n * (8 + 8 + 4 * D + 4 * D)
  update_bytes = n * (8 + 4 * D) + uniq * n * (8 + 2 * (4 * D + 4 * D) + 4)
  # per kernel: resolve (k_claim_pending) = id + handed-over (id, slot) read, slot write; apply = grad rows +
  # row/state read+write + ts (the part of §8d's update figure this kernel moves)
  # (mono_table_lookup = k_probe + k_gather_rows once the lookup -> update pattern is seen: the
  # sort of the update order then runs beside the row gather, and the update's "resolve" is
  # k_verify + a 12-byte read-back, its "sort" the residual wait for the prepared order. With
  # hints the probe, the sort and k_classify of a batch ran two steps earlier beside the updates
  # of the batches before: mono_table_lookup = k_hint_fix + k_gather_rows, and those updates'
  # kernels — k_apply here — are timed with the chains of later batches running beside them)
  kern = {
      "k_lookup": (lk_ms, lookup_bytes),
      "resolve (k_verify / k_claim_pending)": (find_ms, n * (8 + 8)),
      "sort (residual wait when prepared)": (sort_ms, 0),
      "apply phase": (apply_ms, n * 4 * D + uniq * n * (2 * (4 * D + 4 * D) + 4)),
  }
  if kapply_ms == kapply_ms:  # the kernel alone: what rocprofv3 reports as mono::k_apply<2, 1, 3>
    kern["k_apply"] = (kapply_ms, (n - n_long) * 4 * D + (uniq * n - u_long) * (2 * (4 * D + 4 * D) + 4))
  else:
    kern["k_apply"] = kern["apply phase"]
  # the dominant single kernel: k_apply (mono_table_lookup is two kernels, k_probe + a row gather
  # throttled on purpose so that the sort of the update order overlaps it: kernel_ms carries it)
  dom = "k_apply"
  achieved = kern[dom][1] / (kern[dom][0] * 1e-3) / 1e9
  step_gbs = (lookup_bytes + update_bytes) / (ms_per_step * 1e-3) / 1e9
  # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE + WRITE_SIZE,
  # separate rocprofv3 --pmc runs of this same command; calibration note in the summary file)
  traffic, traffic_src = None, None
  try:
    import glob
    pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[0-9]_*_pmc_summary.json")))[-1]
    with open(pmc_file) as f:
      pmc = json.load(f)["kernels"]
    members = (["mono::k_apply<2, 1, 3>"] if kapply_ms == kapply_ms else
               ["mono::k_classify", "mono::k_apply<2, 1, 3>", "mono::k_apply_long<1, 3, 1, 6144>",
                "mono::k_combine<1>"])
    if args.batch == 65536 and args.rows == 1e8 and args.skew == 3.0:
      traffic = sum((pmc[m]["FETCH_SIZE_KB_last_launch"] + pmc[m]["WRITE_SIZE_KB_last_launch"]) *
                    1024.0 for m in members)
      traffic_src = "profiles/%s (%s)" % (os.path.basename(pmc_file), " + ".join(members))
  except (OSError, KeyError, ValueError, IndexError):
    pass

  res = {
      "metric": "FID lookups+updates/sec (node)", "value": 2.0 * n * world / (dt / args.steps),
      "unit": "lookups+updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
      "vs_baseline": None, "dtype": "f32", "data": "synthetic",
      "config": {"workload": "criteo-shaped: 26 single-hot slots, %.3g-row table, dim 16 fp32, "
                             "batch %d, Adagrad, lookup + apply_gradients(dedup)%s" %
                             (total_rows, args.batch,
                              "" if args.no_hint else ", the ids of the next %d batches announced "
                              "by lookup_hint" % args.hint_ahead),
                 "fids_per_step": n, "unique_ratio": uniq, "skew": args.skew,
                 "parallelism": "independent table per GPU" if world > 1 else "1 GPU",
                 "populate_s": t_pop, "table_rows": rows_live},
      "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                   "traffic_source": traffic_src, "algorithmic_bytes": kern[dom][1],
                   "kernel_note": "k_apply = the sub-group apply kernel alone (segments of <= 96 "
                                  "occurrences), timed with HIP events on the stream it runs on; "
                                  "'apply phase' = k_verify's verdict, then k_apply beside "
                                  "k_long_pack -> k_long_chain (hot keys, summed in input order), "
                                  "joined on the caller's stream",
                   "long_segment_occurrences": n_long, "long_segment_keys": u_long,
                   "measured_random_access_ceiling": measured_ceiling(),
                   "kernel_ms": {k: v[0] for k, v in kern.items()},
                   "summation_order": "every key's gradients in input order, the reference's "
                                      "(bridge :241-266), at any number of occurrences; "
                                      "MONO_CHUNK=<rows> is the opt-in chunked sum",
                   "no_hint_ms_per_step": no_hint_ms,
                   "no_hint_value": (2.0 * n / (no_hint_ms * 1e-3)) if no_hint_ms else None,
                   "no_hint_note": "the same step through the reference's own two calls only "
                                   "(mono_table_lookup, mono_table_apply_gradients), measured in this "
                                   "run after the timed region",
                   "step_algorithmic_GBps": step_gbs, "step_frac": step_gbs / HBM_PEAK_GBS},
  }
  if rank == 0 and world == 1 and not args.no_cpu:
    res["cpu_baseline"] = cpu_baseline(pool[:8], grad, lr)
  elif rank == 0:
    res["cpu_baseline"] = None
  if rank == 0:
    emit(res)
  if world > 1:
    torch.distributed.destroy_process_group()


def measured_ceiling():
  """The random-access ceiling of this GPU as scripts/exp_ceiling.hip measured it (committed output:
  profiles/r02_ceiling.txt): independent random reads over a 32 GiB footprint, by access width."""
  path = os.path.join(ROOT, "profiles", "r02_ceiling.txt")
  res = {"source": "profiles/r02_ceiling.txt (scripts/exp_ceiling.hip)"}
  try:
    with open(path) as f:
      for line in f:
        m = re.match(r"read (64|128) B over +32\.00 GiB .* ([0-9.]+) G acc/s +([0-9.]+) GB/s", line)
        if m:
          res["random_%sB_Greq_per_s" % m.group(1)] = float(m.group(2))
          res["random_%sB_GBps" % m.group(1)] = float(m.group(3))
  except OSError:
    return None
  return res


if __name__ == "__main__":
  main()
