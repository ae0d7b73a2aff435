"""debug: the training-loop test, step by step"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import hip_backend, oracle_backend
from tests.test_table_gpu import CFGS
name = "adagrad16"
cfg = CFGS[name]
o, h = oracle_backend.make_table(cfg), hip_backend.make_table(cfg)
D = 16; nlr = 1
rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))
n, vocab = 24000, 3000
lrs = [0.05]
base = rng.integers(0, vocab, n).astype(np.int64)
o.assign_add(np.arange(vocab), np.zeros((vocab, D), np.float32), req_time=1)
h.assign_add(np.arange(vocab), np.zeros((vocab, D), np.float32), req_time=1)
probe = np.arange(-5, vocab + 4000, dtype=np.int64)
def cmp(tag, step):
  wo, wh = o.lookup(probe), h.lookup(probe)
  bad = np.nonzero((wo.view(np.uint32) != wh.view(np.uint32)).any(1))[0]
  print(tag, "step", step, "rows differing", len(bad), probe[bad][:10], "maxabs", np.abs(wo - wh).max(), flush=True)
for step in range(12):
  ids = base.copy()
  ids[rng.integers(0, n, 6000)] = 7
  ids[rng.integers(0, n, 300)] = 11
  if step in (5, 9):
    ids[rng.integers(0, n, 50)] = vocab + 100 * step + np.arange(50)
  g = rng.normal(0, 0.1, (n, D)).astype(np.float32)
  want, got = o.lookup(ids), h.lookup(ids)
  print("step", step, "lookup equal", np.array_equal(want.view(np.uint32), got.view(np.uint32)))
  if step == 3:
    o.lookup(probe); h.lookup(probe); o.lookup(ids); h.lookup(ids)
  dedup = step % 3 != 1
  if step == 7:
    o.assign_add(ids, g, req_time=100 + step); h.assign_add(ids, g, req_time=100 + step)
  else:
    o.apply_gradients(ids, g, lrs, req_time=100 + step, global_step=step, enable_dedup=dedup)
    h.apply_gradients(ids, g, lrs, req_time=100 + step, global_step=step, enable_dedup=dedup)
  cmp("after apply", step)
  if step == 10:
    o.apply_gradients(ids, g, lrs, req_time=200, global_step=step, enable_dedup=True)
    h.apply_gradients(ids, g, lrs, req_time=200, global_step=step, enable_dedup=True)
    cmp("after 2nd apply", step)
  hi, hrows, hts = h.export()
  k = int(np.searchsorted(hi, 7))
  o.import_rows(np.array([7], np.int64), hrows[k:k + 1], hts[k:k + 1])
