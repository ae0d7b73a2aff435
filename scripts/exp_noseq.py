"""config-2 step WITHOUT dedup (every occurrence its own Adagrad step, the reference's default):
ms per apply_gradients call on a smaller table. usage: exp_noseq.py"""
import ctypes as C, sys, time
import torch
sys.path.insert(0, ".")
import bench
from monolith_amd import _lib, entry, hash_table_ops
dev = torch.device("cuda", 0)
tc = entry.EmbeddingHashTableConfig(
    segments=[entry.CombineAsSegment(16, entry.RandomUniformInitializer(-0.05, 0.05),
                                     entry.AdagradOptimizer(initial_accumulator_value=0.1))],
    initial_capacity=int(1e7), seed=1)
t = hash_table_ops.hash_table_from_config(entry.HashTableConfigInstance(tc, [0.01]), name_suffix="x")
bench.SLOT_VOCAB = [max(10, v // 20) for v in bench.SLOT_VOCAB]
gen = torch.Generator(device=dev); gen.manual_seed(1)
pool = [bench.make_batch(gen, 65536, dev, 3.0) for _ in range(4)]
g = torch.randn((pool[0].numel(), 16), generator=gen, device=dev) * 0.1
for dedup in (True, False):
  for i in range(5):
    t = t.apply_gradients(pool[i % 4], g, global_step=i, enable_dedup=dedup)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(40):
    t = t.apply_gradients(pool[i % 4], g, global_step=i, enable_dedup=dedup)
  torch.cuda.synchronize()
  print("dedup" if dedup else "per-occurrence", "%.1f us / apply" % ((time.perf_counter() - t0) / 40 * 1e6), flush=True)
