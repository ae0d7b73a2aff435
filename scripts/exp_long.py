"""Experiment: time of the long-segment apply kernel for one hot key with K occurrences."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monolith_amd import _lib, entry, hash_table_ops
dev = torch.device("cuda", 0)
tc = entry.EmbeddingHashTableConfig(segments=[entry.CombineAsSegment(16, entry.ZerosInitializer(), entry.AdagradOptimizer())], initial_capacity=1 << 20)
t = hash_table_ops.hash_table_from_config(entry.HashTableConfigInstance(tc, [0.01]), name_suffix="x")
h = t.table.h
_lib.check(_lib.lib.mono_table_profile(h, 1))
lrs = (C.c_float * 1)(0.01)
ms = (C.c_float * 3)()
sp = _lib.stream_ptr()
for K, nkeys in [(1, 1700000), (2, 850000), (8, 212500), (32, 53125), (90, 18888), (30000, 1)]:
  ids = (torch.arange(nkeys, device=dev, dtype=torch.int64) * 7919 + 5).repeat_interleave(K)
  ids = ids[torch.randperm(ids.numel(), device=dev)].contiguous()
  g = torch.randn((ids.numel(), 16), device=dev)
  for dedup in (1, 0):
    r = []
    for it in range(4):
      _lib.check(_lib.lib.mono_table_apply_gradients(h, None, _lib.vp(ids), _lib.vp(g), C.c_int64(ids.numel()), lrs, 1, C.c_int64(1), C.c_int64(0), dedup, sp))
      _lib.check(_lib.lib.mono_table_last_phase_ms(h, ms))
      r.append(ms[2])
    print("K=%d keys=%d n=%d dedup=%d apply_ms=%.3f (us/row of one key %.4f)" % (K, nkeys, ids.numel(), dedup, min(r), min(r) * 1e3 / K))
