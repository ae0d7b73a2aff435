"""What a wrong mono_table_lookup_hint costs (config 2's table and batch): the lookup that follows
a hint for OTHER ids re-resolves every position in k_hint_fix. Prints ms per lookup for: no hint,
right hint, wholly wrong hint, 5 % wrong."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from monolith_amd import entry, hash_table_ops
d = torch.device("cuda", 0)
tc = entry.EmbeddingHashTableConfig(
    segments=[entry.CombineAsSegment(16, entry.ZerosInitializer(),
                                     entry.AdagradOptimizer(initial_accumulator_value=0.1))],
    initial_capacity=int(1.02e8))
t = hash_table_ops.hash_table_from_config(entry.HashTableConfigInstance(tc, [0.01]), name_suffix="wh")
bench.populate(t, d)
gen = torch.Generator(device=d); gen.manual_seed(3)
pool = [bench.make_batch(gen, 65536, d, 3.0) for _ in range(6)]
n = pool[0].numel()
g = torch.randn((n, 16), device=d, generator=gen) * 0.1
def run(kind, steps=12):
  global t
  ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]
  for i in range(steps):
    ids = pool[i % 6]
    ev[i][0].record()
    t.lookup(ids)
    ev[i][1].record()
    nxt = pool[(i + 1) % 6]
    if kind == "right":
      t.lookup_hint(nxt, 1)
    elif kind == "wrong":
      t.lookup_hint(pool[(i + 3) % 6], 1)
    elif kind == "5% wrong":
      h = nxt.clone()
      sel = torch.randperm(n, device=d, generator=gen)[: n // 20]
      h[sel] = pool[(i + 3) % 6][sel]
      t.lookup_hint(h, 1)
    t = t.apply_gradients(ids, g, global_step=i, enable_dedup=True, req_time=10 + i)
  torch.cuda.synchronize()
  ms = [e[0].elapsed_time(e[1]) for e in ev[4:]]
  print("%-10s lookup %.3f ms" % (kind, sum(ms) / len(ms)))
for k in ("none", "right", "wrong", "5% wrong"):
  run(k)
