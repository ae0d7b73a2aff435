"""Host time per call of the sharded step (N = 1): wraps the ops DistributedMultiTypeHashTableMpi
goes through with perf_counter (no device sync inside), prints ms per step of each and the step."""
import sys, time, collections
import torch
sys.path.insert(0, ".")
import bench
from monolith_amd import distributed_ps_sync as dps, distribution_ops as dops, hash_table_ops as hto
acc = collections.defaultdict(float)
def wrap(mod, name):
  f = getattr(mod, name)
  def g(*a, **k):
    t = time.perf_counter()
    r = f(*a, **k)
    acc[name] += time.perf_counter() - t
    return r
  setattr(mod, name, g)
for n in ("fused_reorder_by_indices", "fused_gather_embeddings_by_input",
          "fused_gather_embeddings_by_input_gradient"):
  wrap(dops, n)
wrap(hto, "fused_lookup"); wrap(hto, "fused_apply_gradient"); wrap(dps, "alltoall")
# second level: the C calls alone (ctypes call -> return), to split wrapper time from library time
from monolith_amd import _lib as L
class Proxy:
  def __init__(self, lib):
    self._lib, self._cache = lib, {}
  def __getattr__(self, name):
    if name not in self._cache:
      f = getattr(self._lib, name)
      def g(*a, _f=f, _n="C:" + name):
        t = time.perf_counter()
        r = _f(*a)
        acc[_n] += time.perf_counter() - t
        return r
      self._cache[name] = g
    return self._cache[name]
L.lib = Proxy(L.lib)
orig_emit = bench.emit
def emit(res):
  steps = res["steps"] + res["warmup"]
  print("ms/step", round(res["ms_per_step"], 4), file=sys.stderr)
  for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("host %-45s %.1f us/step" % (k, v / steps * 1e6), file=sys.stderr)
  orig_emit(res)
bench.emit = emit
sys.argv = ["bench.py"] + (sys.argv[1:] or ["--sharded"]) + ["--steps", "50", "--warmup", "10", "--no-cpu"]
bench.main()
