"""GPU parity on workloads shaped like BASELINE.json's configs (at sizes the oracle finishes in
seconds) and size-independent properties at config 2's full size.

  config 1  demo CTR entry: dim-1 FTRL bias + dim-16 SGD vector in one multi-segment table
  config 3  multi-hot slots (Zipf 1.1), sum / mean pooling through fused_embedding_to_layout, FTRL
  config 5  streaming: sliding hash filter (threshold 3), 5 % fresh ids per step, req_time
            advancing 1 h / step, per-slot expiry, evict every 24 steps
  config 2  full size (1e8-row table, 1.7 M ids per step): round trips / linearity / idempotence
"""
import numpy as np
import pytest
import torch

from oracle import cpu
from tests import hip_backend, oracle_backend
from tests.test_layout_gpu import assert_gradient_rows, exact_order

pytestmark = pytest.mark.gpu


def dev():
  return torch.device("cuda", 0)


def _same_state(o, h):
  oi, orows, ots = o.export()
  hi, hrows, hts = h.export()
  assert np.array_equal(oi, hi), "key sets differ"
  assert np.array_equal(ots, hts), "timestamps differ"
  assert np.array_equal(orows.view(np.uint32), hrows.view(np.uint32)), "rows differ"


def test_config1_demo_entry_stream():
  """native_training/demo: every slot's entry is a dim-1 FTRL bias + a dim-16 SGD vector; six
  slots, small batches, no dedup (the demo's default) -> bit-exact state after 20 steps."""
  cfg = {"segments": [{"dim": 1, "opt": "ftrl", "init_acc": 0.1, "beta": 1.0, "l1": 0.0, "l2": 0.0},
                      {"dim": 16, "opt": "sgd", "init": "uniform", "init_a": -0.015625,
                       "init_b": 0.015625}], "seed": 3}
  o, h = oracle_backend.make_table(cfg), hip_backend.make_table(cfg)
  rng = np.random.Generator(np.random.PCG64(1))
  for step in range(20):
    slots = np.repeat(np.arange(1, 7, dtype=np.int64), 64)
    ids = (slots << 48) | rng.zipf(1.5, len(slots)).astype(np.int64) % 5000
    e_o, e_h = o.lookup(ids), h.lookup(ids)
    assert np.array_equal(e_o.view(np.uint32), e_h.view(np.uint32))
    g = rng.normal(0, 0.05, (len(ids), 17)).astype(np.float32)
    for t in (o, h):
      t.apply_gradients(ids, g, [0.01, 0.05], req_time=1000 + step, global_step=step)
  _same_state(o, h)


@pytest.mark.parametrize("B,bag,bag_max,vocab", [(256, 12, 60, 3000), (2048, 50, 200, 100000)],
                         ids=["b256_bag12", "b2048_bag50"])
def test_config3_pooling_ftrl_pipeline(B, bag, bag_max, vocab):
  """Multi-hot bags (Poisson lengths, Zipf 1.1 ids) -> per-slot unique ids -> table lookup ->
  fused_embedding_to_layout (half SUM, half MEAN; CONCAT + ADDN layouts) -> layout gradient ->
  FTRL apply_gradients(dedup). Layout tensors, the gradients and the table state after three
  steps are bit-identical between the HIP path and the oracle. The second case has config 3's
  own bag lengths (Poisson(50) in [1, 200]): 0.8 M uses per step, Zipf-head rows with thousands
  of uses each (the layout gradient's long in-order chains)."""
  from monolith_amd import distribution_ops, feature_configs as fc
  from tests import layout_cases as lc
  S, D = 8, 16
  cfgs = lc.make_pooling_configs(S, D)
  plan = fc.build_layout_plan(cfgs)
  names = plan.feature_names
  tcfg = {"segments": [{"dim": D, "opt": "ftrl", "init_acc": 0.1, "beta": 1.0, "init": "uniform",
                        "init_a": -0.05, "init_b": 0.05}], "seed": 5}
  # one table per table name ("t0", "t1"), features of a table share it (ids carry the slot)
  tabs = sorted({cfgs.feature_configs[n].table for n in names})
  o_t = {t: oracle_backend.make_table(tcfg) for t in tabs}
  h_t = {t: hip_backend.make_table(tcfg) for t in tabs}
  rng = np.random.Generator(np.random.PCG64(7))
  for step in range(3):
    # ragged batch
    fid_offset, feature_offset, nfl_offset = [], [], []
    buf_index = {}
    for t_i, t in enumerate(tabs):
      feats = [n for n in names if cfgs.feature_configs[n].table == t]
      for f_i, n in enumerate(feats):
        buf_index[n] = sum(len([m for m in names if cfgs.feature_configs[m].table == tt])
                           for tt in tabs[:t_i]) + f_i
    per_feature_fids, use_count, rows_used = {}, {}, {}
    for n_i, n in enumerate(names):
      nfl_offset.append(len(feature_offset))
      rowmap = {}
      for b in range(B):
        feature_offset.append(len(fid_offset))
        k = int(np.clip(rng.poisson(bag), 1, bag_max))
        fids = ((n_i + 1) << 48) | (rng.zipf(1.1, k).astype(np.int64) % vocab)
        for fid in fids.tolist():
          r = rowmap.setdefault(fid, len(rowmap))
          rows_used.setdefault(n, []).append(r)
          fid_offset.append((buf_index[n] << 32) | r)
      per_feature_fids[n] = np.array(list(rowmap), np.int64)
      use_count[n] = np.bincount(np.array(rows_used[n], np.int64), minlength=len(rowmap))
    fid_offset = np.array(fid_offset, np.uint64)
    feature_offset = np.array(feature_offset, np.int32)
    nfl_offset = np.array(nfl_offset, np.uint32)
    # lookups (unique ids per feature, features of one table back to back = one V2 buffer)
    embs_o, embs_h, splits = [], [], []
    for t in tabs:
      feats = [n for n in names if cfgs.feature_configs[n].table == t]
      ids = np.concatenate([per_feature_fids[n] for n in feats])
      embs_o.append(o_t[t].lookup(ids))
      embs_h.append(h_t[t].lookup(ids))
      assert np.array_equal(embs_o[-1].view(np.uint32), embs_h[-1].view(np.uint32))
      splits.append(np.cumsum([0] + [len(per_feature_fids[n]) for n in feats]).astype(np.int64))
    et = [torch.tensor(e.reshape(-1), device=dev(), requires_grad=True) for e in embs_h]
    outs = distribution_ops.fused_embedding_to_layout(
        et, [torch.tensor(s) for s in splits], torch.tensor(fid_offset.view(np.int64), device=dev()),
        torch.tensor(feature_offset, device=dev()), nfl_offset, B, "example_batch", cfgs, 1)
    inp = {"embeddings_list": [e.reshape(-1) for e in embs_o], "fid_list_row_split": splits,
           "ps_num": 1}
    bufs = lc.oracle_bufs(plan, inp)
    exp = cpu.embedding_to_layout(plan.slices, plan.out_shapes, bufs, fid_offset, feature_offset,
                                  nfl_offset, B)
    for a, e in zip(outs, exp):
      assert np.array_equal(a.detach().cpu().numpy().view(np.uint32), e.view(np.uint32))
    gouts = [rng.normal(0, 0.1, e.shape).astype(np.float32) for e in exp]
    torch.autograd.backward(outs, [torch.tensor(g, device=dev()) for g in gouts])
    eg = cpu.embedding_to_layout_grad(plan.slices, plan.slice_idx_of, plan.max_slice_num,
                                      [(len(a), d) for a, d in bufs], fid_offset, feature_offset,
                                      nfl_offset, B, gouts)
    cut = 0
    for t_i, t in enumerate(tabs):
      feats = [n for n in names if cfgs.feature_configs[n].table == t]
      g_o = np.concatenate([eg[cut + i] for i in range(len(feats))]).reshape(-1, D)
      cut += len(feats)
      g_h = et[t_i].grad.detach().cpu().numpy().reshape(-1, D)
      # addends of a gradient row: the id's uses in the batch, once per layout (CONCAT + ADDN).
      # Every row is bit-exact, the Zipf heads of the large case included; only under
      # MONO_CHUNK=4096 (tests/test_layout_gpu.py::test_layout_gradient_chunked_switch_subprocess
      # runs this case with it) are rows above 4096 addends summed in chunks and bounded instead.
      uses = np.concatenate([use_count[n] for n in feats]) * 2
      assert_gradient_rows(g_h, g_o, uses, (step, t))
      if not exact_order():
        g_h = g_o  # the table update is compared on identical gradients
      ids = np.concatenate([per_feature_fids[n] for n in feats])
      o_t[t].apply_gradients(ids, g_o, [0.01], req_time=50 + step, enable_dedup=True)
      h_t[t].apply_gradients(ids, g_h, [0.01], req_time=50 + step, enable_dedup=True)
  for t in tabs:
    _same_state(o_t[t], h_t[t])


@pytest.mark.parametrize("ahead", [0, 2], ids=["plain", "hinted"])
def test_config5_streaming_filter_eviction(ahead):
  """Online-training stream: sliding filter with occurrence threshold 3, 5 % of every step's ids
  from a fresh range, req_time advancing one hour per step, slots 1-3 expire after one day,
  evict every 24 steps. Ids carry distinct 12-bit filter signatures (alias-free by construction,
  SURVEY.md 8c) and a batch never fills a filter block (no mid-batch rotation). "hinted": the
  next two lookups are announced between every lookup and its update (mono_table_lookup_hint),
  as bench.py does on config 2 — insertions by the gated update and evictions overtake hints all
  the time here, and nothing may change."""
  fcfg = {"kind": "sliding", "capacity": 400000, "split_num": 7, "default_threshold": 3}
  tcfg = {"segments": [{"dim": 8, "opt": "adagrad", "init": "uniform", "init_a": -0.05,
                        "init_b": 0.05}], "seed": 9, "default_expire_days": 36500,
          "slot_expire": {"1": 1, "2": 1, "3": 1}}
  of, hf = oracle_backend.make_filter(fcfg), hip_backend.make_filter(fcfg)
  o, h = oracle_backend.make_table(tcfg), hip_backend.make_table(tcfg, hf)
  rng = np.random.Generator(np.random.PCG64(13))
  n, fresh_base, steps = 600, 300, 60
  uid = {}  # (slot, r) -> unique filter signature: at most 4096 distinct ids in the whole stream
  stream = []
  for step in range(steps + 2):
    slots = rng.integers(1, 7, n).astype(np.int64)
    r = (rng.zipf(1.3, n) % 300).astype(np.int64)
    fresh = rng.random(n) < 0.05
    r[fresh] = fresh_base + np.arange(int(fresh.sum()))
    fresh_base += int(fresh.sum())
    sig = np.array([uid.setdefault((int(s_), int(r_)), len(uid)) for s_, r_ in zip(slots, r)],
                   np.int64)
    assert len(uid) < 4096
    ids = (slots << 48) | (sig << 17) | 1   # signature = bits 17..28
    stream.append((ids, rng.normal(0, 0.1, (n, 8)).astype(np.float32)))
  for step in range(steps):
    ids, g = stream[step]
    now = 1_700_000_000 + 3600 * step
    eo, eh = o.lookup(ids), h.lookup(ids)
    assert np.array_equal(eo.view(np.uint32), eh.view(np.uint32)), step
    for a in range(1, ahead + 1):
      h.lookup_hint(stream[step + a][0], a)
    o.apply_gradients(ids, g, [0.02], req_time=now, global_step=step, enable_dedup=True, filt=of)
    h.apply_gradients(ids, g, [0.02], req_time=now, global_step=step, enable_dedup=True, filt=hf)
    if step % 24 == 23:
      o.evict(now); h.evict(now)
      assert o.size() == h.size()
  _same_state(o, h)
  assert h.size() < steps * n  # the filter kept one-off ids out, eviction removed stale rows
  probe = np.unique(ids)
  assert np.array_equal(of.get(probe), hf.get(torch.as_tensor(probe).cuda()).cpu().numpy())


def test_config2_full_size_properties():
  """1e8-row table (dim 16, Adagrad), 1.7 M ids per step drawn like bench.py: properties that do
  not need the oracle at this size.
    * size == number of distinct ids ever written (torch.unique as the counter)
    * lookup after assign returns the assigned rows bit-exactly; misses return zeros
    * SGD linearity: dedup and non-dedup application agree to 1e-6 relative (fp32)
    * export -> clear -> import round trip restores every probed row bit-exactly
    * evict(now) with everything fresh removes nothing; with everything stale removes all"""
  import bench
  from monolith_amd import entry, hash_table_ops
  d = dev()
  tc = entry.EmbeddingHashTableConfig(
      segments=[entry.CombineAsSegment(16, entry.ZerosInitializer(), entry.SgdOptimizer())],
      initial_capacity=int(1.02e8))
  t = hash_table_ops.hash_table_from_config(entry.HashTableConfigInstance(tc, [0.5]),
                                            name_suffix="full_size_sgd")
  bench.populate(t, d)
  assert t.size() == sum(bench.SLOT_VOCAB) == 100000000
  gen = torch.Generator(device=d)
  gen.manual_seed(5)
  ids = bench.make_batch(gen, 65536, d, 3.0)
  n = ids.numel()
  assert n == 1703936
  uniq = torch.unique(ids)
  # assign / lookup round trip at full size
  vals = torch.randn((uniq.numel(), 16), generator=gen, device=d)
  t.assign(uniq, vals, req_time=100)
  assert torch.equal(t.lookup(uniq), vals)
  assert torch.equal(t.lookup(ids), vals[torch.searchsorted(uniq, ids)])
  miss = torch.arange(5, device=d, dtype=torch.int64) | (900 << 48)
  assert torch.count_nonzero(t.lookup(miss)) == 0
  assert t.size() == 100000000
  # SGD linearity: k sequential steps == one step on the sum (to fp32 rounding)
  g = torch.randn((n, 16), generator=gen, device=d) * 0.1
  t2 = hash_table_ops.hash_table_from_config(entry.HashTableConfigInstance(tc, [0.5]),
                                             name_suffix="full_size_sgd2")
  t2.assign(uniq, vals, req_time=100)
  t.apply_gradients(ids, g, 0, enable_dedup=False, req_time=200)
  t2.apply_gradients(ids, g, 0, enable_dedup=True, req_time=200)
  a, b = t.lookup(uniq), t2.lookup(uniq)
  cnt = torch.bincount(torch.searchsorted(uniq, ids), minlength=uniq.numel()).unsqueeze(1)
  tol = 1e-6 * (vals.abs() + 0.5 * 0.1 * 6 * cnt)  # |w| + lr * sum|g| bound
  assert torch.all((a - b).abs() <= tol + 1e-7)
  # ids with a single occurrence took the identical path: bit-exact
  single = (cnt.squeeze(1) == 1)
  assert torch.equal(a[single], b[single])
  # export -> clear -> import (the small table: 888 k rows)
  ids_d, rows_d, ts_d = t2.save_as_tensor()
  assert ids_d.numel() == uniq.numel() and int(ts_d.max()) == 200
  t2.clear()
  assert t2.size() == 0
  t2.restore_from_tensor(ids_d, rows_d, ts_d)
  assert torch.equal(t2.lookup(uniq), b) and t2.size() == uniq.numel()
  # eviction: nothing is older than the default 36500 days
  t2.evict(200 + 86400 * 100)
  assert t2.size() == uniq.numel()
  t2.evict(200 + 86400 * 36501)
  assert t2.size() == 0 and torch.count_nonzero(t2.lookup(uniq)) == 0
