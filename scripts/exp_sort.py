"""Sort alone: n pairs, key bits as in the headline step (slots of a 1e8-row table), avg ms."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_ops_gpu import _sort_pairs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1703936
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 28
rng = np.random.Generator(np.random.PCG64(1))
k = rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(np.uint32)
k[rng.integers(0, n, n // 4)] = k[:64][rng.integers(0, 64, n // 4)]
kd = torch.from_numpy(k.view(np.int32)).cuda()
_sort_pairs(kd, bits, 5)
for _ in range(3):
  print("n %d bits %d: %.2f us / sort" % (n, bits, _sort_pairs(kd, bits, 50)[2] * 1e3), flush=True)
