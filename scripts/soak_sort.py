"""Soak of the fused sort passes (spin-waits on published tile rows): random sizes up to the
512-tile limit and beyond, many repetitions, result checked against numpy for every size, the
sticky error word read at the end. usage: soak_sort.py [seconds]"""
import ctypes as C, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_ops_gpu import _sort_pairs
from monolith_amd import _lib
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.Generator(np.random.PCG64(99))
t0, sorts, last = time.time(), 0, time.time()
while time.time() - t0 < budget:
  n = int(rng.integers(1, 600 * 8192)) if rng.random() < 0.3 else int(rng.integers(1, 260 * 8192))
  bits = int(rng.integers(1, 33))
  k = rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(np.uint32)
  if rng.random() < 0.5:
    k[rng.integers(0, n, n // 2)] = k[0]
  kd = torch.from_numpy(k.view(np.int32)).cuda()
  ko, vo, ms = _sort_pairs(kd, bits, 25)
  order = np.argsort(k, kind="stable").astype(np.uint32)
  assert np.array_equal(vo, order) and np.array_equal(ko, k[order]), (n, bits)
  sorts += 25
  if time.time() - last > 20:
    print("sorts so far", sorts, flush=True)
    last = time.time()
fn = _lib.lib.mono_debug_sort_error
fn.restype = C.c_int
print("sorts", sorts, "error word", fn())
assert fn() == 0
