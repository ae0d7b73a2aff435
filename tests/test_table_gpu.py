"""GPU parity: the HIP table through the C ABI vs the CPU oracle and the reference's vectors."""
import zlib

import os

import numpy as np
import pytest
import torch

from oracle import cpu
from tests import hip_backend, oracle_backend
from tests.scenario import load_scenarios, run_scenario

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sc", load_scenarios(), ids=lambda s: s["name"])
def test_reference_scenarios(sc):
  """The reference's own unit-test vectors, replayed through libmonolith_hip.so."""
  run_scenario(hip_backend, sc)


def _pair(cfg, filt_cfg=None):
  of = oracle_backend.make_filter(filt_cfg) if filt_cfg else None
  hf = hip_backend.make_filter(filt_cfg) if filt_cfg else None
  return oracle_backend.make_table(cfg), hip_backend.make_table(cfg, hf), of, hf


def _assert_same_state(o, h, exact=True):
  oi, orows, ots = o.export()
  hi, hrows, hts = h.export()
  assert np.array_equal(oi, hi), "key sets differ"
  assert np.array_equal(ots, hts), "timestamps differ"
  if exact:
    assert np.array_equal(orows.view(np.uint32), hrows.view(np.uint32)), \
        "rows differ (max abs %g)" % np.abs(orows - hrows).max()
  else:
    np.testing.assert_allclose(hrows, orows, rtol=1e-6, atol=1e-7)
  assert o.size() == h.size()


CFGS = {
    "sgd1": {"segments": [{"dim": 1, "opt": "sgd"}]},
    "sgd3": {"segments": [{"dim": 3, "opt": "sgd", "init": "constant", "init_a": 0.25}]},
    "adagrad16": {"segments": [{"dim": 16, "opt": "adagrad", "init": "uniform", "init_a": -0.05,
                                "init_b": 0.05}], "seed": 7},
    "adagrad16wd": {"segments": [{"dim": 16, "opt": "adagrad", "weight_decay": 0.1,
                                  "init": "ones"}]},
    "ftrl16": {"segments": [{"dim": 16, "opt": "ftrl", "beta": 1.0, "l1": 0.01, "l2": 0.1}]},
    "ftrl6": {"segments": [{"dim": 6, "opt": "ftrl"}]},
    "adagrad64": {"segments": [{"dim": 64, "opt": "adagrad"}]},
    "demo_ftrl1_sgd16": {"segments": [{"dim": 1, "opt": "ftrl", "init_acc": 1e-6, "beta": 1.0},
                                      {"dim": 16, "opt": "sgd", "init": "uniform",
                                       "init_a": -0.1, "init_b": 0.1}]},
    "multi3": {"segments": [{"dim": 2, "opt": "adagrad"}, {"dim": 5, "opt": "ftrl", "l1": 0.001},
                            {"dim": 1, "opt": "sgd", "init": "ones"}]},
}


@pytest.mark.parametrize("name", list(CFGS))
@pytest.mark.parametrize("dedup", [False, True])
def test_random_stream_bit_exact(name, dedup):
  """Seeded id streams with heavy duplication, growth through several rehashes, all write ops:
  table state (keys, rows incl. optimizer state, timestamps) bit-identical to the oracle."""
  cfg = CFGS[name]
  o, h, _, _ = _pair(cfg)
  rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode()) % 1000 + int(dedup)))
  D, nseg = o.dim, len(cfg["segments"])
  for step in range(6):
    n = int(rng.integers(1, 4000))
    vocab = int(rng.choice([8, 300, 20000]))
    ids = (rng.integers(0, vocab, n) * 2654435761 - 5).astype(np.int64)
    if step == 2:
      ids[:4] = [-1, -2, -1, 0]  # the two sentinel-valued keys
    g = rng.normal(0, 0.1, (n, D)).astype(np.float32)
    lrs = rng.uniform(0.001, 0.1, nseg).astype(np.float32)
    for t in (o, h):
      t.apply_gradients(ids, g, lrs, req_time=1000 + step, enable_dedup=dedup)
    q = np.concatenate([ids[:50], rng.integers(-5, 5, 10)])
    assert np.array_equal(o.lookup(q).view(np.uint32), h.lookup(q).view(np.uint32))
    if step % 2 == 1:
      v = rng.normal(0, 1, (n, D)).astype(np.float32)
      for t in (o, h):
        t.assign_add(ids, v, req_time=2000 + step)
      m = max(1, n // 3)
      for t in (o, h):
        t.assign(ids[:m], v[:m], req_time=3000 + step)
    assert np.array_equal(o.contains(q), h.contains(q))
  _assert_same_state(o, h)
  assert o.max_update_ts() == h.max_update_ts()


# (the widths and optimizer templates the chain kernel is instantiated for, beside CFGS')
HOT_CFGS = dict(CFGS, **{
    "sgd16": {"segments": [{"dim": 16, "opt": "sgd", "init": "constant", "init_a": 0.25}]},
    "adagrad8": {"segments": [{"dim": 8, "opt": "adagrad", "init": "uniform", "init_a": -0.05,
                               "init_b": 0.05}], "seed": 3},
    "adagrad32": {"segments": [{"dim": 32, "opt": "adagrad", "init": "uniform", "init_a": -0.05,
                                "init_b": 0.05}], "seed": 5},
})


@pytest.mark.parametrize("name", ["adagrad16", "ftrl6", "multi3", "adagrad64", "ftrl16", "sgd16",
                                  "adagrad8", "adagrad32"])
def test_hot_keys_in_reference_order(name):
  """Keys with thousands of occurrences in a batch (9001 and exactly 4096 here): every sum runs
  in the reference's order at any length (embedding_hash_table_tf_bridge.cc:241-266 sums a key's
  gradients sequentially), so EVERY row is bit-exact — dedup on and off, assign_add — over several
  steps with no re-synchronisation of the hot rows. (Rows of width 8 / 16 / 32 / 64 go through
  k_long_pack + k_long_chain, the others through k_apply_long.)"""
  cfg = HOT_CFGS[name]
  o, h, _, _ = _pair(cfg)
  rng = np.random.Generator(np.random.PCG64(11))
  D, nseg = o.dim, len(cfg["segments"])
  n = 30000
  ids = (rng.integers(0, 5000, n) * 2654435761 + 3).astype(np.int64)
  pos = rng.permutation(n)
  hot, edge = np.int64(77), np.int64(78)
  ids[pos[:9001]] = hot
  ids[pos[9001:9001 + 4096]] = edge
  lrs = np.full(nseg, 0.05, np.float32)
  for step, dedup in enumerate([True, False, True]):
    g = rng.normal(0, 0.1, (n, D)).astype(np.float32)
    for t in (o, h):
      t.apply_gradients(ids, g, lrs, req_time=10 + step, enable_dedup=dedup)
    v = rng.normal(0, 1, (n, D)).astype(np.float32)
    for t in (o, h):
      t.assign_add(ids, v, req_time=20 + step)
    oi, orows, ots = o.export()
    hi, hrows, hts = h.export()
    assert np.array_equal(oi, hi) and np.array_equal(ots, hts)
    assert np.array_equal(orows.view(np.uint32), hrows.view(np.uint32)), (step, dedup)


def test_hot_key_bound_at_config2_shape():
  """The hottest id of BASELINE config 2's stream: ~30 000 occurrences per batch, gradients
  N(0, 0.1), dim-16 Adagrad, dedup on, six steps with NO re-synchronisation. The HIP table's row
  equals the oracle's — the strictly sequential fp32 sum, the reference's order
  (embedding_hash_table_tf_bridge.cc:251-266) — bit for bit after every step, weights and
  accumulator, like every other row. A third table (the oracle on a batch whose hot occurrences
  are collapsed to their float64 sum) records how far that order is from the exact sum; printed
  only — the bar is the reference's result, not the exact one."""
  cfg = CFGS["adagrad16"]
  o, h, _, _ = _pair(cfg)
  x = oracle_backend.make_table(cfg)
  rng = np.random.Generator(np.random.PCG64(2025))
  D, n, hot = 16, 120000, np.int64(424242)
  lrs = np.full(1, 0.01, np.float32)
  err_o_tot = 0.0
  for step in range(6):
    ids = (rng.integers(0, 50000, n) * 2654435761 + 11).astype(np.int64)
    pos = rng.permutation(n)
    ids[pos[:30000]] = hot
    g = rng.normal(0, 0.1, (n, D)).astype(np.float32)
    gx = g.copy()
    hp = np.flatnonzero(ids == hot)
    gx[hp] = 0.0
    gx[hp[0]] = g[hp].astype(np.float64).sum(0).astype(np.float32)
    o.apply_gradients(ids, g, lrs, req_time=50 + step, enable_dedup=True)
    h.apply_gradients(ids, g, lrs, req_time=50 + step, enable_dedup=True)
    x.apply_gradients(ids, gx, lrs, req_time=50 + step, enable_dedup=True)
    oi, orows, ots = o.export()
    hi, hrows, hts = h.export()
    xi, xrows, _ = x.export()
    assert np.array_equal(oi, hi) and np.array_equal(oi, xi) and np.array_equal(ots, hts)
    assert np.array_equal(orows.view(np.uint32), hrows.view(np.uint32)), step
    ro, rx = orows[oi == hot][0], xrows[oi == hot][0]
    for part in (slice(0, D), slice(D, 2 * D)):  # weights, Adagrad accumulator
      err_o_tot += np.abs(ro[part].astype(np.float64) - rx[part]).sum() / np.abs(rx[part]).max()
  print("hot key, 6 steps: HIP == oracle bit for bit; the sequential fp32 order is %.3g away from "
        "the float64-exact row (summed, relative)" % err_o_tot)


def test_empty_and_ragged_inputs():
  o, h, _, _ = _pair(CFGS["adagrad16"])
  e = np.zeros(0, np.int64)
  for t in (o, h):
    t.apply_gradients(e, np.zeros((0, 16), np.float32), [0.1])
    t.assign(e, np.zeros((0, 16), np.float32))
    t.assign_add(e, np.zeros((0, 16), np.float32))
  assert h.lookup(e).shape == (0, 16) and h.size() == 0
  assert np.array_equal(h.lookup([1, 2, 3]), np.zeros((3, 16), np.float32))
  with pytest.raises(Exception):
    h.t.assign([1, 2], np.zeros((2, 5), np.float32))


def test_evict_and_save_time_expiry():
  """cuckoo_embedding_hash_table.cc:220-233 + hash_table_save_op.cc:142-150 on a seeded stream
  with per-slot expire days."""
  cfg = {"segments": [{"dim": 4, "opt": "adagrad"}], "default_expire_days": 10,
         "slot_expire": {"1": 1, "2": 0, "3": 100}}
  o, h, _, _ = _pair(cfg)
  rng = np.random.Generator(np.random.PCG64(5))
  for step in range(5):
    slots = rng.integers(0, 5, 3000)
    ids = ((slots.astype(np.int64) << 48) | rng.integers(0, 2000, 3000)).astype(np.int64)
    g = rng.normal(0, 1, (3000, 4)).astype(np.float32)
    for t in (o, h):
      t.apply_gradients(ids, g, [0.05], req_time=86400 * step * 3, enable_dedup=True)
  oi, _, _ = o.export(apply_expiry=True)
  hi, _, _ = h.export(apply_expiry=True)
  assert np.array_equal(oi, hi) and len(oi) < o.size()
  for now in (86400 * 4, 86400 * 13, 86400 * 40):
    o.evict(now); h.evict(now)
    _assert_same_state(o, h)
  # evicted keys can be re-inserted (tombstones are skipped, not resurrected)
  ids = np.arange(100, dtype=np.int64) | (2 << 48)
  for t in (o, h):
    t.assign_add(ids, np.ones((100, 4), np.float32), req_time=86400 * 41)
  _assert_same_state(o, h)


def test_periodic_eviction_with_injected_clock():
  """enable_feature_eviction / feature_evict_every_n_hours (embedding_hash_table.proto:84-86): the
  bridge evicts with the largest req_time seen so far every N hours of WALL clock
  (embedding_hash_table_tf_bridge.cc:69-100: the first look only notes the time; a later look
  evicts once N hours have passed since the last eviction and notes the time again). The library
  looks at the entry of every write call; with the clock injected (mono_table_set_clock) the
  evictions fall exactly where the rule below puts them, and the table equals the oracle driven
  by the same rule — rows, timestamps, survivors."""
  import ctypes as C
  from monolith_amd import _lib
  cfg = {"segments": [{"dim": 4, "opt": "adagrad"}], "default_expire_days": 2,
         "slot_expire": {"1": 1, "3": 100}, "evict_every_n_hours": 6}
  o, h, _, _ = _pair(cfg)
  rng = np.random.Generator(np.random.PCG64(17))
  wall0, last, n_evict = 1_700_000_000, 0, 0
  count = C.c_int64()
  for step in range(40):
    wall = wall0 + step * 2 * 3600 + (step % 3) * 1000      # the host's clock: ~2 h per call
    req = 86400 * step // 3                                 # the data's clock: 8 h per call
    _lib.check(_lib.lib.mono_table_set_clock(h.t.table.h, C.c_int64(wall), None))
    # the reference's rule, restated: look BEFORE the call is applied
    if last == 0:
      last = wall
    elif wall - last >= 6 * 3600:
      o.evict(o.max_update_ts())
      last = wall
      n_evict += 1
    slots = rng.integers(0, 5, 2000)
    ids = ((slots.astype(np.int64) << 48) | rng.integers(0, 3000, 2000)).astype(np.int64)
    g = rng.normal(0, 1, (2000, 4)).astype(np.float32)
    kind = step % 3
    for t in (o, h):
      if kind == 0:
        t.apply_gradients(ids, g, [0.05], req_time=req, enable_dedup=True)
      elif kind == 1:
        t.assign_add(ids, g, req_time=req)
      else:
        t.assign(ids[:500], g[:500], req_time=req)
    if step % 5 == 4:
      _assert_same_state(o, h)
  _lib.check(_lib.lib.mono_table_set_clock(h.t.table.h, C.c_int64(-1), C.byref(count)))
  assert count.value == n_evict and n_evict >= 10, (count.value, n_evict)
  _assert_same_state(o, h)
  assert o.size() < 15000  # (rows did leave)
  # a table without the flag never evicts by itself
  o2, h2, _, _ = _pair({"segments": [{"dim": 4, "opt": "adagrad"}], "default_expire_days": 0})
  for t in (o2, h2):
    t.assign_add(np.arange(10, dtype=np.int64), np.ones((10, 4), np.float32), req_time=5)
    t.assign_add(np.arange(10, dtype=np.int64), np.ones((10, 4), np.float32), req_time=9 * 86400)
  assert h2.size() == 10 and o2.size() == 10


def test_save_restore_roundtrip():
  """hash_table_restore_op.cc:82-131: clear + upsert of (id, row incl. optimizer state, ts)."""
  o, h, _, _ = _pair(CFGS["ftrl16"])
  rng = np.random.Generator(np.random.PCG64(9))
  ids = rng.integers(-1000, 1000, 5000).astype(np.int64)
  g = rng.normal(0, 1, (5000, 16)).astype(np.float32)
  for t in (o, h):
    t.apply_gradients(ids, g, [0.01], req_time=77, enable_dedup=False)
  dids, drows, dts = h.t.save_as_tensor()
  h2 = hip_backend.make_table(CFGS["ftrl16"])
  h2.t.restore_from_tensor(dids, drows, dts)
  _assert_same_state(o, h2)
  assert h2.max_update_ts() == 77


def test_filter_gate_closed_forms():
  """SURVEY.md §8a closed forms of the hash-filter gate (bridge :167-176, :195-198, :274-291)
  for Optimize (dedup / no dedup), Assign and AssignAdd, on collision-free id streams."""
  fcfg = {"kind": "sliding", "capacity": 200000, "split_num": 7, "default_threshold": 3,
          "slot_thresholds": {"2": 0, "3": 20, "4": 1}}
  for dedup in (False, True):
    o, h, of, hf = _pair(CFGS["sgd3"], fcfg)
    rng = np.random.Generator(np.random.PCG64(11 + dedup))
    for step in range(8):
      slots = rng.integers(0, 6, 600).astype(np.int64)
      k = rng.integers(0, 60, 600).astype(np.int64)
      # distinct 12-bit signatures (bits 17..28) so that no two fids alias in the filter
      ids = (slots << 48) | ((slots * 60 + k) << 17) | k
      g = rng.normal(0, 1, (600, 3)).astype(np.float32)
      o.apply_gradients(ids, g, [0.1], req_time=step, enable_dedup=dedup, filt=of)
      h.apply_gradients(ids, g, [0.1], req_time=step, enable_dedup=dedup, filt=hf)
      if step == 3:
        o.assign_add(ids[:200], g[:200], req_time=step, filt=of)
        h.assign_add(ids[:200], g[:200], req_time=step, filt=hf)
      if step == 5:
        o.assign(ids[:200], g[:200], req_time=step, filt=of)
        h.assign(ids[:200], g[:200], req_time=step, filt=hf)
      _assert_same_state(o, h)
      probe = np.unique(ids)
      assert np.array_equal(of.get(probe), hf.get(torch.as_tensor(probe).cuda()).cpu().numpy())


def test_lookup_slot_cache_hazards():
  """mono_table_lookup remembers the slot of every position for the update that follows
  (table.hip: cache_ids / cache_slots). Everything that could make those slots stale — other
  ids in the update, ids overwritten in place, eviction, clear, growth (rehash), keys missing at
  lookup time — must still give the oracle's result bit for bit."""
  import torch
  from tests import hip_backend, oracle_backend
  cfg = {"segments": [{"dim": 8, "opt": "adagrad", "init": "uniform", "init_a": -0.1, "init_b": 0.1}],
         "seed": 4, "default_expire_days": 1}
  o, h = oracle_backend.make_table(cfg), hip_backend.make_table(cfg)
  rng = np.random.Generator(np.random.PCG64(21))
  day = 86400

  def both(fn):
    fn(o)
    fn(h)

  def check(probe):
    assert np.array_equal(o.lookup(probe).view(np.uint32), h.lookup(probe).view(np.uint32))

  n = 4000
  ids = rng.integers(0, 3000, n).astype(np.int64)
  g = rng.normal(0, 0.2, (n, 8)).astype(np.float32)
  probe = np.arange(-10, 3100, dtype=np.int64)
  # 1. plain forward / backward on the same ids; half of them unknown at lookup time
  both(lambda t: t.lookup(ids))
  both(lambda t: t.apply_gradients(ids, g, [0.05], req_time=10, enable_dedup=True))
  check(probe)
  # 2. update with OTHER ids of the same length
  ids2 = rng.integers(1000, 6000, n).astype(np.int64)
  both(lambda t: t.lookup(ids))
  both(lambda t: t.apply_gradients(ids2, g, [0.05], req_time=11))
  check(np.arange(0, 6100, dtype=np.int64))
  # 3. the id tensor overwritten in place between lookup and update (same device buffer)
  dev_ids = torch.tensor(ids, device="cuda:0")
  h.t.lookup(dev_ids)
  o.lookup(ids)
  dev_ids.copy_(torch.tensor(ids2))
  h.t = h.t.apply_gradients(dev_ids, torch.tensor(g, device="cuda:0"), 0, req_time=12)
  h.lrs = [0.05]
  o.apply_gradients(ids2, g, [0.05], req_time=12)
  # (HipTable.apply_gradients sets lrs before the call; here the previous value 0.05 is in place)
  check(np.arange(0, 6100, dtype=np.int64))
  # 4. eviction between lookup and update: the cached slots may now belong to nobody
  both(lambda t: t.lookup(ids))
  both(lambda t: t.evict(12 + 5 * day))
  assert o.size() == h.size() == 0
  both(lambda t: t.apply_gradients(ids, g, [0.05], req_time=13 + 5 * day))
  check(probe)
  # 5. clear between lookup and update
  both(lambda t: t.lookup(ids))
  both(lambda t: t.clear())
  both(lambda t: t.apply_gradients(ids[:100], g[:100], [0.05], req_time=14 + 5 * day))
  check(probe)
  # 6. growth (rehash) between lookup and update
  both(lambda t: t.lookup(ids))
  big = np.arange(10_000, 210_000, dtype=np.int64)
  both(lambda t: t.assign_add(big, np.ones((len(big), 8), np.float32), req_time=15 + 5 * day))
  both(lambda t: t.apply_gradients(ids, g, [0.05], req_time=16 + 5 * day, enable_dedup=True))
  check(probe)
  oi, orows, ots = o.export()
  hi, hrows, hts = h.export()
  assert np.array_equal(oi, hi) and np.array_equal(ots, hts)
  assert np.array_equal(orows.view(np.uint32), hrows.view(np.uint32))


@pytest.mark.parametrize("name", ["adagrad16", "ftrl16", "adagrad64", "multi3", "sgd3"])
def test_training_loop_prepared_update_order(name):
  """lookup(ids) -> apply_gradients(ids) step after step: from the second step on the lookup
  prepares the update order (table.hip Table::Prep: k_probe, sort + k_classify beside the row
  gather) and the update only verifies it. Every step is the oracle's, bit for bit, across what
  the prepared order has to survive: keys with thousands of occurrences (long segments, split
  chunks), switches between dedup / per-occurrence / assign_add (the classification is redone),
  steps that bring ids the table has never seen (the update falls back to the full path and
  inserts them), a lookup that is not followed by its update, and a repeated update."""
  cfg = CFGS[name]
  o, h, _, _ = _pair(cfg)
  D = sum(sg["dim"] for sg in cfg["segments"])
  nlr = len(cfg["segments"])
  rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))
  n, vocab = 24000, 3000
  lrs = [0.05, 0.02, 0.1][:nlr]
  base = rng.integers(0, vocab, n).astype(np.int64)
  o.assign_add(np.arange(vocab), np.zeros((vocab, D), np.float32), req_time=1)
  h.assign_add(np.arange(vocab), np.zeros((vocab, D), np.float32), req_time=1)
  probe = np.arange(-5, vocab + 4000, dtype=np.int64)
  for step in range(12):
    ids = base.copy()
    ids[rng.integers(0, n, 6000)] = 7            # a hot key: ~6000 occurrences (> one chunk)
    ids[rng.integers(0, n, 300)] = 11            # a long segment that is not split
    if step in (5, 9):                           # ids nobody has seen: the full path inserts them
      ids[rng.integers(0, n, 50)] = vocab + 100 * step + np.arange(50)
    g = rng.normal(0, 0.1, (n, D)).astype(np.float32)
    want, got = o.lookup(ids), h.lookup(ids)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), step
    if step == 3:                                # a second lookup before the update
      assert np.array_equal(o.lookup(probe).view(np.uint32), h.lookup(probe).view(np.uint32))
      o.lookup(ids)
      h.lookup(ids)
    dedup = step % 3 != 1
    if step == 7:
      o.assign_add(ids, g, req_time=100 + step)
      h.assign_add(ids, g, req_time=100 + step)
    else:
      o.apply_gradients(ids, g, lrs, req_time=100 + step, global_step=step, enable_dedup=dedup)
      h.apply_gradients(ids, g, lrs, req_time=100 + step, global_step=step, enable_dedup=dedup)
    if step == 10:                               # the same update once more, no lookup in between
      o.apply_gradients(ids, g, lrs, req_time=200, global_step=step, enable_dedup=True)
      h.apply_gradients(ids, g, lrs, req_time=200, global_step=step, enable_dedup=True)
    # compared through the export (a lookup here would break the lookup -> update pattern that
    # switches the prepared order on). Every row bit for bit, the hot one included: its
    # occurrences are summed in input order (k_long_chain), as the reference sums them.
    oi, orows, ots = o.export()
    hi, hrows, hts = h.export()
    assert np.array_equal(oi, hi) and np.array_equal(ots, hts), step
    assert np.array_equal(orows.view(np.uint32), hrows.view(np.uint32)), step
  _assert_same_state(o, h)
  assert o.max_update_ts() == h.max_update_ts()


@pytest.mark.parametrize("depth", [1, 2, 3])
@pytest.mark.parametrize("name", ["adagrad16", "multi3"])
def test_lookup_hint_never_changes_a_result(name, depth):
  """mono_table_lookup_hint (include/monolith_hip.h; no reference counterpart): the ids of the next
  lookup announced between a lookup and its update. The step loop lookup(t) -> hint(t+1) ->
  update(t) gives the oracle's rows and table state bit for bit — when the hint is right (the
  hinted slot record and update order are used), when the next lookup brings other ids of the same
  size (partly or wholly), another size, when the update in between inserts new keys, when
  eviction / clear / growth come between hint and lookup, when a hint is never consumed, and when
  the update after a hinted lookup is for other ids. depth: how many lookups ahead the loop
  announces (every step announces all of 1..depth; the table takes what extends its queue)."""
  cfg = CFGS[name]
  o, h, _, _ = _pair(cfg)
  D = sum(sg["dim"] for sg in cfg["segments"])
  nlr = len(cfg["segments"])
  rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode()) + 5))
  n, vocab = 20000, 2500
  lrs = [0.05, 0.02, 0.1][:nlr]
  o.assign_add(np.arange(vocab), np.zeros((vocab, D), np.float32), req_time=1)
  h.assign_add(np.arange(vocab), np.zeros((vocab, D), np.float32), req_time=1)

  def batch(step, size=n):
    r = np.random.Generator(np.random.PCG64(1000 + step))
    ids = r.integers(0, vocab, size).astype(np.int64)
    ids[r.integers(0, size, size // 4)] = 7      # a hot key (long segment, split chunks)
    if step in (6, 13):                          # ids nobody has seen: the update inserts them
      ids[r.integers(0, size, 40)] = vocab + 100 * step + np.arange(40)
    return ids

  steps = 20
  for step in range(steps):
    ids = batch(step)
    if step == 4:                                # a lookup of other ids than the hinted ones
      ids = batch(100 + step)
    if step == 5:                                # ... of a few other ids
      ids = ids.copy()
      ids[rng.integers(0, n, 17)] = rng.integers(0, vocab + 50, 17)
    if step == 8:                                # ... of another size
      ids = batch(step, n - 333)
    g = rng.normal(0, 0.1, (len(ids), D)).astype(np.float32)
    want, got = o.lookup(ids), h.lookup(ids)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), step
    for ahead in range(1, depth + 1):
      h.lookup_hint(batch(step + ahead), ahead)
    if step == 10:                               # hints out of turn: ignored, or queued and wrong
      h.lookup_hint(batch(200), depth + 1)
      h.lookup_hint(batch(201), 1)
    upd = ids
    if step == 11:                               # the update after a hinted lookup: other ids
      upd = batch(300)
    dedup = step % 3 != 1
    o.apply_gradients(upd, g, lrs, req_time=100 + step, global_step=step, enable_dedup=dedup)
    h.apply_gradients(upd, g, lrs, req_time=100 + step, global_step=step, enable_dedup=dedup)
    if step == 14:                               # eviction between hint and lookup
      o.evict(50)
      h.evict(50)
    if step == 15:                               # growth (rehash) between hint and lookup
      big = np.arange(50_000, 250_000, dtype=np.int64)
      o.assign_add(big, np.ones((len(big), D), np.float32), req_time=100 + step)
      h.assign_add(big, np.ones((len(big), D), np.float32), req_time=100 + step)
    if step == 17:                               # clear between hint and lookup
      o.clear()
      h.clear()
      o.assign_add(np.arange(vocab), np.zeros((vocab, D), np.float32), req_time=1)
      h.assign_add(np.arange(vocab), np.zeros((vocab, D), np.float32), req_time=1)
    oi, orows, ots = o.export()
    hi, hrows, hts = h.export()
    assert np.array_equal(oi, hi) and np.array_equal(ots, hts), step
    assert np.array_equal(orows.view(np.uint32), hrows.view(np.uint32)), step
  _assert_same_state(o, h)


@pytest.mark.parametrize("depth", [1, 2])
def test_gated_update_keeps_the_hinted_order_under_insertions(depth):
  """Online training behind a hash filter with the next lookups announced ahead: every step the
  gated update inserts ids that the queued hints recorded as absent. The lookup's k_hint_fix
  corrects the slot record (the rows come back), and the hinted order stays in use: the update
  walks the order's tail of then-absent ids by key and finds the inserted ones in the table
  (gated_after_prepared) — rows, key set, timestamps and filter counts bit for bit, every step.
  Ids carry distinct 12-bit filter signatures (alias-free)."""
  fcfg = {"kind": "sliding", "capacity": 400000, "split_num": 7, "default_threshold": 3}
  o, h, of, hf = _pair(CFGS["adagrad16"], fcfg)
  n, base, per_step, steps = 20000, 400, 150, 18

  def fid(k):
    k = np.asarray(k, np.int64)
    return ((1 + k % 5) << 48) | (k << 17) | (k & 0xFFFF)

  def batch(step):
    r = np.random.Generator(np.random.PCG64(500 + step))
    hi_k = base + per_step * (step + 1)      # ids introduced up to and including this step
    assert hi_k < 4096
    k = r.integers(0, hi_k, n)
    k[r.integers(0, n, n // 5)] = 3          # a hot id (long segment beside the gated remainder)
    # ... and two bursts of ids nobody knows yet, this step's and the one before's: on its second
    # sight a burst id is admitted with hundreds of occurrences — a chain in the gated remainder
    # (pack buffer of its own) beside the known hot id's chain
    pos = r.permutation(n)
    k[pos[:600]] = 3900 + step
    if step > 0:
      k[pos[600:1100]] = 3900 + step - 1
    return fid(k)

  known = np.repeat(fid(np.arange(base)), 5)   # (five occurrences: past the gate's threshold)
  o.assign_add(known, np.zeros((len(known), 16), np.float32), req_time=1, filt=of)
  h.assign_add(known, np.zeros((len(known), 16), np.float32), req_time=1, filt=hf)
  assert o.size() == h.size() == base
  rng = np.random.Generator(np.random.PCG64(77))
  inserted_late = 0
  for step in range(steps):
    ids = batch(step)
    g = rng.normal(0, 0.1, (n, 16)).astype(np.float32)
    want, got = o.lookup(ids), h.lookup(ids)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), step
    for ahead in range(1, depth + 1):
      h.lookup_hint(batch(step + ahead), ahead)
    before = o.size()
    dedup = step % 4 != 2
    o.apply_gradients(ids, g, [0.05], req_time=100 + step, global_step=step, enable_dedup=dedup,
                      filt=of)
    h.apply_gradients(ids, g, [0.05], req_time=100 + step, global_step=step, enable_dedup=dedup,
                      filt=hf)
    inserted_late += o.size() - before
    oi, orows, ots = o.export()
    hi, hrows, hts = h.export()
    assert np.array_equal(oi, hi) and np.array_equal(ots, hts), step
    assert np.array_equal(orows.view(np.uint32), hrows.view(np.uint32)), step
  assert inserted_late > 1000    # the stream did insert behind the hints
  import ctypes as C
  from monolith_amd import _lib
  counts = (C.c_int64 * 4)()
  _lib.check(_lib.lib.mono_table_path_counts(h.t.table.h, counts))
  # every update ran on the order its lookup (hinted, from the second step on) had prepared,
  # except the first and those at which the growing table rehashed (they drop the queue; twice
  # here); none tried a prepared order and had to fall back
  assert counts[0] >= steps - 4 and counts[1] == 0, list(counts)
  assert counts[2] >= steps - 5 and counts[3] <= 2, list(counts)
  probe = np.unique(batch(steps - 1))
  assert np.array_equal(of.get(probe), hf.get(torch.as_tensor(probe).cuda()).cpu().numpy())


def test_lookup_entry_and_extract_slot():
  """hash_table_ops_test.py:100-114 (entries of ids 0..2 parse, id 3 -> b""),
  embedding_hash_table_test.h:78-93 (the dump of an SGD dim-1 entry holding 1.0, byte for byte:
  no id field), hash_table_ops_test.py:1190-1194 (slot of 1 << 48 is 1)."""
  import struct
  from monolith_amd import checkpoint as ck, hash_table_ops as hto
  t = hto.test_hash_table(1)
  t = t.assign(torch.tensor([0, 1, 2], device="cuda"), torch.tensor([[0.1], [0.2], [0.3]], device="cuda"))
  es = t.lookup_entry(torch.tensor([0, 1, 2, 3], device="cuda"))
  assert es[3] == b""
  layout = ck.DumpLayout([(1, ck.SGD)])
  for e, v in zip(es[:3], (0.1, 0.2, 0.3)):
    rid, row, ts = layout._decode_one(e)
    assert rid == 0 and ts == 0 and row[0] == np.float32(v)
  t2 = hto.test_hash_table(1)
  t2 = t2.assign(torch.tensor([-10], device="cuda"), torch.tensor([[1.0]], device="cuda"))
  (e,) = t2.lookup_entry(torch.tensor([-10], device="cuda"))
  assert e == b"\x15" + struct.pack("<f", 1.0) + b"\x1a\x04\x0a\x02\x12\x00" + b"\x20\x00"
  # a stateful entry: Adagrad accumulator and timestamp ride along
  t3 = hto.test_hash_table(2, use_adagrad=True, learning_rate=0.1)
  ids = torch.tensor([5, 9], device="cuda")
  t3 = t3.apply_gradients(ids, torch.ones((2, 2), device="cuda"), global_step=0, req_time=1234)
  e5, e7 = t3.lookup_entry(torch.tensor([5, 7], device="cuda"))
  assert e7 == b""
  rid, row, ts = ck.DumpLayout([(2, ck.ADAGRAD)])._decode_one(e5)
  _, rows, tss = t3.save_as_tensor()
  assert ts == 1234 and np.array_equal(row, rows.cpu().numpy()[0])
  entry = b"\x09" + struct.pack("<q", 1 << 48)
  assert hto.extract_slot_from_entry([entry, b"", b"\x15\x00\x00\x80\x3f" + entry]).tolist() == [1, 0, 1]
  assert hto.extract_slot_from_entry([b"\x09" + struct.pack("<q", 3 << 54)], fid_v2=False).tolist() == [3]


def test_apply_kernel_timing_hook():
  """mono_table_profile / mono_table_last_phase_ms / mono_table_last_apply_kernel_ms: the kernel
  timing exists exactly when the write call forked k_apply onto the auxiliary stream (batches of
  more than 96 ids, which may hold a long segment), and fails loudly otherwise."""
  import ctypes as C
  from monolith_amd import _lib, hash_table_ops as hto
  t = hto.test_hash_table(4, use_adagrad=True, learning_rate=0.1)
  h = t.table.h
  ms3, ka = (C.c_float * 3)(), C.c_float(0)
  assert _lib.lib.mono_table_last_phase_ms(h, ms3) != 0          # profiling is off
  _lib.check(_lib.lib.mono_table_profile(h, 1))
  ids = torch.arange(50, device="cuda")
  g = torch.ones((50, 4), device="cuda")
  t = t.apply_gradients(ids, g, global_step=0, enable_dedup=True)
  _lib.check(_lib.lib.mono_table_last_phase_ms(h, ms3))
  assert all(ms3[i] > 0 for i in range(3))
  assert _lib.lib.mono_table_last_apply_kernel_ms(h, C.byref(ka)) != 0   # 50 ids: no fork
  ids = torch.cat([torch.arange(5000, device="cuda"),
                   torch.full((300,), 7, device="cuda", dtype=torch.int64)])
  t = t.apply_gradients(ids, torch.ones((5300, 4), device="cuda"), global_step=1, enable_dedup=True)
  _lib.check(_lib.lib.mono_table_last_apply_kernel_ms(h, C.byref(ka)))
  _lib.check(_lib.lib.mono_table_last_phase_ms(h, ms3))
  assert 0 < ka.value <= ms3[2]
  _lib.check(_lib.lib.mono_table_profile(h, 0))
