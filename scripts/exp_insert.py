"""Timing experiment: fused_reorder_by_indices on the bench id stream (debug knobs via MONO_DBG)."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from monolith_amd import distribution_ops as ops
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
ids = bench.make_batch(gen, 65536, dev, 3.0)
for _ in range(3): ops.fused_reorder_by_indices([ids], 1, [16])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): ops.fused_reorder_by_indices([ids], 1, [16])
torch.cuda.synchronize(); print("reorder us", (time.perf_counter() - t0) / 20 * 1e6)
